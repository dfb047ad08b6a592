"""MI355X-native ternary / binary quantized matmul path (HIP kernels behind a C ABI).

The directory name carries a hyphen (it names the reference project); import it as
``ternary_vllm_amd`` - the one-file module of that name at the repository root registers this
directory as the package.

  custom_ops_mod   operator surface: zero_out, prevent_folding, pack_fn, unpack_ternary, ternary_matmul, ...
  modules          ScaledTernaryLinear / ScaledBinaryLinear / ... and linear_to_quantized
  layers           packed inference layers (ScaledTernaryMatmul, ...Unpack, ...Tf, ...PackedMM, ScaledTernary)
  utils            quantize_moondream / quantize_text_model, clip_binary_weights
  conversion       torch layer -> packed layer transfer, the packed arena, row sharding
"""
from . import custom_ops_mod
from .custom_ops_mod import (fit_binary_scale, fit_ternary_scale, pack_fn, prevent_folding, ternarize_pack, ternary_matmul,
                             ternary_mm, unpack_ternary, zero_out)
from . import modules, layers, utils, conversion  # noqa: E402
from .modules import (ScaledBinary01Linear, ScaledBinaryLinear, ScaledTernaryLinear, SignSTESat, SmoothlyBinarizedLinear,
                      linear_to_quantized)
from .conversion import PackedArena, shard_rows

__all__ = ["custom_ops_mod", "modules", "layers", "utils", "conversion", "zero_out", "prevent_folding", "pack_fn",
           "unpack_ternary", "ternary_matmul", "ternarize_pack", "fit_ternary_scale", "fit_binary_scale", "ternary_mm",
           "ScaledTernaryLinear", "ScaledBinaryLinear", "ScaledBinary01Linear", "SmoothlyBinarizedLinear", "SignSTESat",
           "linear_to_quantized", "PackedArena", "shard_rows"]
