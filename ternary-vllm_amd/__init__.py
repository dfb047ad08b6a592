"""MI355X-native ternary / binary quantized matmul path (HIP kernels behind a C ABI).

The directory name carries a hyphen (it names the reference project); import it as
``ternary_vllm_amd`` - the one-file module of that name at the repository root registers this
directory as the package.
"""
from . import custom_ops_mod
from .custom_ops_mod import (fit_binary_scale, fit_ternary_scale, pack_fn, prevent_folding, ternarize_pack, ternary_matmul,
                             ternary_mm, unpack_ternary, zero_out)

__all__ = ["custom_ops_mod", "zero_out", "prevent_folding", "pack_fn", "unpack_ternary", "ternary_matmul", "ternarize_pack",
           "fit_ternary_scale", "fit_binary_scale", "ternary_mm"]
