"""Model-level quantization driver: which layers get quantized, and with which options.

Mirror of ``binary/utils.py``.  The reference walks ``model.text_model.named_modules()`` of moondream2
(a remote-code model that cannot be fetched here); the selection rule itself does not depend on that
model and is kept verbatim: every module whose name contains neither ``lm_head`` nor the prefix of a
skipped block has its direct ``nn.Linear`` children converted (``binary/utils.py:14-21``).
"""
from __future__ import annotations

from typing import Iterable, Optional

from torch import nn

from .modules import (ScaledBinary01Linear, ScaledBinaryLinear, ScaledTernaryLinear, SmoothlyBinarizedLinear,
                      linear_to_quantized)


def layers_to_skip(n_blocks: int, start_skip: int = 1, last_skip: int = 0):
    """Name fragments of the transformer blocks left unquantized (``binary/utils.py:14-15``)."""
    return [f"transformer.h.{i}." for i in range(start_skip)] + [f"transformer.h.{n_blocks - 1 - i}." for i in range(last_skip)]


def quantize_text_model(phimodel: nn.Module, start_skip=1, last_skip=0, quantization="binary", scaling="none",
                        neuron_scale="uniform", remove_blocks: Optional[Iterable[int]] = None, verbose: bool = False):
    """``quantize_moondream`` (``binary/utils.py:7-33``) on the text model itself: expects
    ``phimodel.transformer.h`` to be the ``ModuleList`` of decoder blocks.  Returns the names converted."""
    n_blocks = len(phimodel.transformer.h)
    skip = layers_to_skip(n_blocks, start_skip, last_skip)
    converted = []
    for name, mod in list(phimodel.named_modules()):
        if "lm_head" not in name and not any(s in name for s in skip):
            if linear_to_quantized(mod, quantization=quantization, scaling=scaling, neuron_scale=neuron_scale):
                converted.append(name)
                if verbose:
                    print(f"Converted {name} to {quantization}")
    remove = list(remove_blocks or [])
    phimodel.transformer.h = nn.ModuleList([m for i, m in enumerate(phimodel.transformer.h) if i not in remove])
    for i, m in enumerate(phimodel.transformer.h):
        if hasattr(m, "mixer"):
            m.mixer.layer_idx = i  # binary/utils.py:32-33
    return converted


def quantize_moondream(model, start_skip=1, last_skip=0, quantization="binary", scaling="none", neuron_scale="uniform",
                       remove_blocks=None):
    """Same signature as ``binary/utils.py:7``; ``model.text_model`` is the Phi text model."""
    return quantize_text_model(model.text_model, start_skip, last_skip, quantization, scaling, neuron_scale, remove_blocks)


def quantized_size_bits(model: nn.Module) -> int:
    """Size accounting of ``binary/utils.py:41-57``: 1 bit per binary weight, 2 bits per ternary weight,
    16 bits per scale / bias element, 16 bits per remaining dense parameter."""
    n_bits = 0
    for mod in model.modules():
        if isinstance(mod, (ScaledBinaryLinear, ScaledBinary01Linear)):
            n_bits += mod.weights.numel() + 16 * 2 * (mod.bias.numel() if mod.bias is not None else 0)
        elif isinstance(mod, SmoothlyBinarizedLinear):
            n_bits += mod.weights.numel() + 16 * (mod.bias.numel() if mod.bias is not None else 0)
        elif isinstance(mod, ScaledTernaryLinear):
            n_bits += mod.weights.numel() * 2 + 16 * 2 * (mod.bias.numel() if mod.bias is not None else 0)
        elif isinstance(mod, nn.Linear):
            n_bits += mod.weight.numel() * 16 + (mod.bias.numel() if mod.bias is not None else 0) * 16
        elif isinstance(mod, (nn.Embedding, nn.LayerNorm)):
            n_bits += sum(p.numel() for p in mod.parameters()) * 16
    return n_bits


def clip_binary_weights(model: nn.Module):
    """Clip the latent weights of every quantized layer to [-1, 1] (``binary/utils.py:74-77``)."""
    for mod in model.modules():
        if isinstance(mod, (ScaledBinaryLinear, ScaledBinary01Linear, ScaledTernaryLinear, SmoothlyBinarizedLinear)):
            mod.weights.data = mod.weights.data.clip(-1.0, 1.0)
            mod.invalidate_packed()
