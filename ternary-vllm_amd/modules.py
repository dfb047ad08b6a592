"""Quantized Linear layers whose forward runs the packed HIP matmul.

Host-side mirror of the reference's torch layer family (``binary/modules.py``): same class names,
constructor signatures, parameter names (``weights``, ``bias``, ``scale``) and ``maxval`` attribute, so a
state dict written by the reference loads unchanged (``convert.py:75-78`` iterates ``named_parameters``).

What differs is how ``forward`` computes.  The reference re-quantizes all latent weights and runs a dense
``F.linear`` on every call (``binary/modules.py:102-104``); here the latent weights are quantized and packed
to 2 bits once per weight version (``ternarize_pack`` kernel) and every call is one fused
``ternary_matmul`` launch (matmul, ``* scale``, ``+ bias``, clip).  The quantized weight values are exactly
{-1,0,+1}; the reference's straight-through term ``q + w - w.detach()`` leaves its training-time weights up
to one float32 ulp away from that (see ``tests/golden``), which the packed form cannot and does not mimic.

There is no CPU path: parameters and inputs must live on the GPU.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn

from . import custom_ops_mod as ops

__all__ = ["tri_step_ste", "sign_ste", "step_ste", "sign_ste_sat", "step_ste_sat", "SignSTESat", "ScaledBinaryLinear", "ScaledBinary01Linear", "ScaledTernaryLinear",
           "SmoothlyBinarizedLinear", "linear_to_quantized"]


# ---------------------------------------------------------------------------------------------------
# quantizer functions (binary/modules.py:5-42).  Forward value from the HIP kernel; identity gradient.
# ---------------------------------------------------------------------------------------------------

class _QuantizeSTE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, quantization, saturate):
        ctx.saturate = saturate
        if saturate:
            ctx.save_for_backward(x)
        return ops.quantize(x, quantization)

    @staticmethod
    def backward(ctx, g):
        if ctx.saturate:  # the *_sat variants pass the gradient of clip(x, -1, 1)
            (x,) = ctx.saved_tensors
            g = g * ((x >= -1.0) & (x <= 1.0)).to(g.dtype)
        return g, None, None


def tri_step_ste(x: torch.Tensor) -> torch.Tensor:
    """``(x >= 0.5) - (x <= -0.5)`` with a straight-through gradient (``binary/modules.py:21-26``).
    Unlike the reference, works with or without ``requires_grad`` (its no-grad branch raises)."""
    return _QuantizeSTE.apply(x, "ternary", False)


def sign_ste(x: torch.Tensor) -> torch.Tensor:
    """``(x >= 0) * 2 - 1`` (``binary/modules.py:5-10``)."""
    return _QuantizeSTE.apply(x, "binary", False)


def step_ste(x: torch.Tensor) -> torch.Tensor:
    """``(x >= 0)`` (``binary/modules.py:13-18``)."""
    return _QuantizeSTE.apply(x, "binary01", False)


def sign_ste_sat(x: torch.Tensor) -> torch.Tensor:
    """``sign_ste`` whose gradient is that of ``clip(x, -1, 1)`` (``binary/modules.py:29-34``)."""
    return _QuantizeSTE.apply(x, "binary", True)


def step_ste_sat(x: torch.Tensor) -> torch.Tensor:
    """``step_ste`` whose gradient is that of ``clip(x, -1, 1)`` (``binary/modules.py:37-42``)."""
    return _QuantizeSTE.apply(x, "binary01", True)


class SignSTESat(nn.Module):
    """Module form of :func:`sign_ste_sat` (``binary/modules.py:45-51``): ``(x >= 0) * 2 - 1`` forward, the gradient
    of ``clip(x, -1, 1)`` backward.  Works with or without ``requires_grad`` (the reference's no-grad branch raises)."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return sign_ste_sat(x)


# ---------------------------------------------------------------------------------------------------
# fused forward with a plain-torch backward (training is not the accelerated path)
# ---------------------------------------------------------------------------------------------------

class _PackedLinearFn(torch.autograd.Function):
    """y = clip((x @ Q(w)^T) * scale + bias).  Forward: HIP.  Backward: dense torch ops on the unpacked
    weights with the straight-through estimator for the latent weights - kept so that the layers remain
    trainable like the reference's; it is not tuned."""

    @staticmethod
    def forward(ctx, x, weights, scale, bias, pw, maxval):
        y = ops.ternary_matmul(x, pw, scale, bias, maxval)
        ctx.save_for_backward(x, scale, bias if bias is not None else scale.new_empty(0), pw, y)
        ctx.has_bias = bias is not None
        ctx.maxval = maxval
        ctx.k = weights.shape[1]
        return y

    @staticmethod
    def backward(ctx, gy):
        x, scale, bias, pw, y = ctx.saved_tensors
        inside = (y.abs() < ctx.maxval).to(gy.dtype)  # clip passes no gradient where it saturates
        g = gy * inside
        n = pw.shape[0]
        g2 = g.reshape(-1, n)
        x2 = x.reshape(-1, ctx.k)
        q = ops.unpack_ternary(pw, ctx.k, dtype=x.dtype)
        gs = g2 * scale.to(g2.dtype)
        gx = (gs @ q).reshape(x.shape) if ctx.needs_input_grad[0] else None
        gw = (gs.t() @ x2).to(torch.float32) if ctx.needs_input_grad[1] else None  # straight-through
        gscale = None
        if ctx.needs_input_grad[2]:
            acc = (x2 @ q.t())
            gscale = (g2 * acc).sum(dim=0).to(scale.dtype)
        gbias = g2.sum(dim=0).to(bias.dtype) if (ctx.has_bias and ctx.needs_input_grad[3]) else None
        return gx, gw, gscale, gbias, None, None


class _PackedCache:
    """2-bit packed copy of a layer's latent ``weights``, re-packed when they change.

    The copy is keyed on the parameter's storage address, version counter, device and dtype, which follows every
    autograd-visible update (optimizer steps, ``copy_``, ``mul_`` ...).  A write through ``.data`` - what the reference's
    checkpoint loader does, ``p.data[:] = v`` at ``convert.py:78`` - changes neither, so: loading a state dict, switching
    between ``train()`` and ``eval()`` and ``clip_binary_weights`` drop the copy, a training-mode forward with gradients
    enabled re-packs on every call (the reference re-quantizes on every call too, ``binary/modules.py:102``), and code
    that writes ``weights.data`` by hand between two evaluation-mode forwards calls :meth:`invalidate_packed`."""

    quantization = "ternary"

    def _init_packed_cache(self):
        self._pw: Optional[torch.Tensor] = None
        self._pw_key = None

    def invalidate_packed(self) -> None:
        self._pw = None
        self._pw_key = None

    def packed_weights(self) -> torch.Tensor:
        w = self.weights
        key = (w.data_ptr(), w._version, w.device, w.dtype)
        if self._pw is None or self._pw_key != key or (self.training and torch.is_grad_enabled()):
            self._pw = ops.ternarize_pack(w.detach(), 0.5, self.quantization)
            self._pw_key = key
        return self._pw

    def train(self, mode: bool = True):
        self.invalidate_packed()
        return super().train(mode)

    def _load_from_state_dict(self, *args, **kwargs):
        self.invalidate_packed()
        return super()._load_from_state_dict(*args, **kwargs)


class _ScaledQuantLinear(_PackedCache, nn.Module):
    """Shared body of the three scaled layers: latent ``weights`` [N,K], ``scale`` [N], ``bias`` [N] | None."""

    quantization = "ternary"

    def __init__(self, in_features: int, out_features: int, bias: bool = True, max_val=100.0):
        super().__init__()
        self.weights = nn.Parameter(torch.randn(out_features, in_features))
        self.bias = nn.Parameter(torch.randn(out_features)) if bias else None
        self.scale = nn.Parameter(torch.full((out_features,), 1.0))
        self.maxval = abs(max_val)
        self._init_packed_cache()

    @property
    def in_features(self) -> int:
        return self.weights.shape[1]

    @property
    def out_features(self) -> int:
        return self.weights.shape[0]

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        pw = self.packed_weights()
        if torch.is_grad_enabled() and (x.requires_grad or self.weights.requires_grad and self.training):
            return _PackedLinearFn.apply(x, self.weights, self.scale, self.bias, pw, self.maxval)
        return ops.ternary_matmul(x, pw, self.scale, self.bias, self.maxval)

    def extra_repr(self) -> str:
        return f"in_features={self.in_features}, out_features={self.out_features}, bias={self.bias is not None}, maxval={self.maxval}"


class ScaledTernaryLinear(_ScaledQuantLinear):
    """``binary/modules.py:93-109``: weights quantized to {-1,0,+1} at thresholds +-0.5 (inclusive)."""
    quantization = "ternary"


class ScaledBinaryLinear(_ScaledQuantLinear):
    """``binary/modules.py:55-71``: weights quantized to {-1,+1} by sign (0 -> +1)."""
    quantization = "binary"


class ScaledBinary01Linear(_ScaledQuantLinear):
    """``binary/modules.py:74-90``: weights quantized to {0,1} by step (0 -> 1)."""
    quantization = "binary01"


class SmoothlyBinarizedLinear(_PackedCache, nn.Module):
    """``binary/modules.py:112-146``.  Evaluation: weights binarized by sign, no scale, bias inside the
    clip - one fused packed matmul with unit scale.  Training (latent weights clipped to [-1,1], dense
    matmul, ``:140-141``) stays a dense torch matmul: it has no quantized operand to pack."""

    quantization = "binary"

    def __init__(self, in_features: int, out_features: int, bias: bool = True, max_val=100.0):
        super().__init__()
        self.weights = nn.Parameter(torch.randn(out_features, in_features))
        self.bias = nn.Parameter(torch.randn(out_features)) if bias else None
        self.maxval = abs(max_val)
        self._init_packed_cache()
        self._ones = None

    def count_converged_parameters(self, eps):
        return (1.0 - self.weights.abs() < eps).sum()

    def count_binary_parameters(self):
        return self.weights.numel()

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.training:
            w = self.weights.clip(-1.0, 1.0)
            y = nn.functional.linear(x, w.to(x.dtype), bias=self.bias.to(x.dtype) if self.bias is not None else None)
            return y.clip(-self.maxval, self.maxval)
        pw = self.packed_weights()
        if self._ones is None or self._ones.device != pw.device:
            self._ones = torch.ones(pw.shape[0], dtype=torch.float32, device=pw.device)
        return ops.ternary_matmul(x, pw, self._ones, self.bias, self.maxval)


# ---------------------------------------------------------------------------------------------------
# one-time conversion of nn.Linear children (binary/modules.py:189-299)
# ---------------------------------------------------------------------------------------------------

@torch.no_grad()
def linear_to_quantized(model: nn.Module, quantization="binary", scaling="none", neuron_scale="uniform", kmeans_iter=10):
    """Replace every direct ``nn.Linear`` child of ``model`` by its quantized counterpart; returns whether
    anything was converted.  Same options and error behaviour as ``binary/modules.py:189-299``
    (unknown ``neuron_scale`` / ``scaling`` strings raise ``ValueError`` in the binary branches; the ternary
    branch treats every ``neuron_scale`` other than ``"uniform"`` as per-row).  Scale fits and latent
    weights are computed by the HIP fit kernels, so the Linear must already be on the GPU."""
    converted = False
    for name, mod in list(model.named_children()):
        if not isinstance(mod, nn.Linear):
            continue
        has_bias = mod.bias is not None
        if quantization in ("binary", "binary01"):
            cls = ScaledBinaryLinear if quantization == "binary" else ScaledBinary01Linear
            latent, scale = ops.fit_binary_scale(mod.weight, quantization, neuron_scale, scaling)
            q = cls(mod.in_features, mod.out_features, has_bias)
        elif quantization == "smoothBinary":
            q = SmoothlyBinarizedLinear(mod.in_features, mod.out_features, has_bias)
            latent, scale = mod.weight.detach().float(), None
        elif quantization == "ternary":
            latent, scale = ops.fit_ternary_scale(mod.weight, neuron_scale, kmeans_iter)
            q = ScaledTernaryLinear(mod.in_features, mod.out_features, has_bias)
        else:
            continue  # the reference silently ignores unknown quantization names (no else branch)
        q.to(mod.weight.device)
        q.weights.data.copy_(latent)
        if scale is not None:
            q.scale.data.copy_(scale)
        if has_bias:
            q.bias.data = mod.bias.data
        model.add_module(name, q)
        converted = True
    return converted
