"""Packed-inference layers: mirrors of the reference's Keras layer variants.

``impl/tf/moondream/ternary.py`` defines five Dense replacements that share one contract -
``Layer(units, activation=None, clip_val=100.0)``, weights in the order ``[scale, bias, packed-or-kernel]``
(``utils/conversion.py:41,58,77,92``), ``call(x) = act(clip((x @ W^T) * scale + bias, +-clip_val))`` - and
differ in how the ternary weights are stored:

=========================  =========================================  ================================
reference layer            weight storage                              here
=========================  =========================================  ================================
``ScaledTernaryMatmul``    uint8 ``[units, ceil(K/4)(+pad)]`` fused op  fused HIP matmul
``ScaledTernaryUnpack``    same, ``ops.unpack_ternary`` + dense matmul  HIP unpack + dense matmul (torch)
``ScaledTernaryPackedMM``  same bytes, 4 strided dense matmuls          fused HIP matmul (same maths)
``ScaledTernaryTf``        planar 1-D uint8 ``[ceil(units*K/4)]``       HIP planar unpack + dense matmul
``ScaledTernary``          latent float kernel ``[K, units]`` (larq)    ternarize+pack once, fused matmul
=========================  =========================================  ================================

TensorFlow / Keras are not part of this stack; the layers are ``torch.nn.Module`` s with the Keras-style
``build`` / ``call`` / ``set_weights`` / ``get_weights`` methods the reference's conversion code uses.
The unfused variants exist so that fused and unfused results can be compared like the reference does.
"""
from __future__ import annotations

from functools import partial
from typing import Callable, Optional, Sequence, Union

import torch
from torch import nn

from . import custom_ops_mod as ops

Activation = Optional[Union[str, Callable]]


def gelu_tanh(x: torch.Tensor) -> torch.Tensor:
    """tanh-approximate GELU, the activation the reference gives fc1 (``utils/conversion.py:185``)."""
    return nn.functional.gelu(x, approximate="tanh")


def _fusable(act: Activation) -> Optional[str]:
    """Name of the in-kernel activation if ``act`` is one the matmul epilogue implements."""
    if act is None:
        return None
    if isinstance(act, str):
        if act in ("gelu_tanh", "gelu"):
            return "gelu_tanh"
        raise ValueError(f"Unknown activation: {act}")
    if act is gelu_tanh:
        return "gelu_tanh"
    if isinstance(act, partial) and act.func is nn.functional.gelu and act.keywords.get("approximate") == "tanh":
        return "gelu_tanh"
    return None


class _PackedDenseBase(nn.Module):
    def __init__(self, units: int, activation: Activation = None, clip_val: float = 100.0):
        super().__init__()
        self.units = int(units)
        self.act = activation
        self.clip_val = abs(float(clip_val))
        self.register_buffer("scale", torch.ones(self.units))    # initializer="ones"  (ternary.py:258)
        self.register_buffer("bias", torch.zeros(self.units))    # initializer="zeros" (ternary.py:259)
        self.orig_shape = None
        self.built = False

    # Keras-style plumbing ---------------------------------------------------------------------
    def build(self, input_shape: Sequence[int]):
        raise NotImplementedError

    def _post(self, y: torch.Tensor, fused: bool) -> torch.Tensor:
        if self.act is not None and not fused:
            act = gelu_tanh if isinstance(self.act, str) else self.act
            y = act(y)
        return y

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not self.built:
            self.build(x.shape)
        return self.call(x)

    def _epilogue(self, acc: torch.Tensor) -> torch.Tensor:
        y = acc * self.scale.to(acc.dtype)
        y = y + self.bias.to(acc.dtype)
        return y.clip(-self.clip_val, self.clip_val)


class ScaledTernaryMatmul(_PackedDenseBase):
    """``impl/tf/moondream/ternary.py:254-303``: ``ops.ternary_matmul(x, pw, scale, bias, clip)`` then the
    activation.  ``pw`` is allocated with the reference's row length ``K//4 + (4 - K%4 if K%4 else 0)``
    (``:269-270``) and random bytes (``:272-276``); ``set_weights`` replaces it."""

    def build(self, input_shape):
        k = int(input_shape[-1])
        self.orig_shape = (self.units, k)
        dev = self.scale.device
        pw = torch.randint(0, 255, (self.units, ops.reference_row_bytes(k)), dtype=torch.uint8, device=dev)
        self.register_buffer("pw", pw)
        self.built = True

    def set_weights(self, weights):
        scale, bias, pw = weights
        if not self.built:
            raise RuntimeError("build() the layer (call it once) before set_weights, as in Keras")
        dev = self.scale.device
        pw = torch.as_tensor(pw, dtype=torch.uint8, device=dev)
        if pw.dim() != 2 or pw.shape[0] != self.units or pw.shape[1] < ops.packed_row_bytes(self.orig_shape[1]):
            raise ValueError(f"packed weights of shape {tuple(pw.shape)} do not fit a layer of shape {self.orig_shape}")
        self.scale = torch.as_tensor(scale, dtype=torch.float32, device=dev).reshape(self.units).clone()
        self.bias = torch.as_tensor(bias, dtype=torch.float32, device=dev).reshape(self.units).clone()
        self.pw = pw.contiguous().clone()

    def get_weights(self):
        return [self.scale, self.bias, self.pw]

    def call(self, x: torch.Tensor) -> torch.Tensor:
        fused = _fusable(self.act)
        y = ops.ternary_matmul(x, self.pw, self.scale, self.bias, self.clip_val, activation=fused)
        return self._post(y, fused is not None)


class ScaledTernaryPackedMM(ScaledTernaryMatmul):
    """``impl/tf/moondream/ternary.py:349-538``.  Same packed bytes; the reference evaluates
    ``sum_i x[..., i::4] @ plane_i^T`` with four dense matmuls (``:507-530``), which is the product the fused
    kernel computes directly.  ``pack_fn`` is the layer's own packer (``:374-398``)."""

    def pack_fn(self, w: torch.Tensor) -> torch.Tensor:
        if w.dtype != torch.int8:
            w = w.to(torch.int8)
        return ops.pack_fn(w)


class ScaledTernaryUnpack(ScaledTernaryMatmul):
    """``impl/tf/moondream/ternary.py:306-346``: ``w = ops.unpack_ternary(pw, orig_dim)`` then a dense
    ``matmul(x, w, transpose_b=True)``, scale, bias, clip, activation - the unfused twin."""

    def call(self, x: torch.Tensor) -> torch.Tensor:
        w = ops.unpack_ternary(self.pw, x.shape[-1], dtype=x.dtype)
        y = self._epilogue(x @ w.t())
        return self._post(y, False)


class ScaledTernaryMatmulPair(nn.Module):
    """Two built ``ScaledTernaryMatmul`` layers that read the SAME input, run as one packed matmul: ``first`` must be
    linear, ``second`` may carry the fusable activation (the Phi block's Wqkv and fc1 behind one LayerNorm,
    ``impl/tf/moondream/layers.py:334-347``; fc1's GELU folded in at ``utils/conversion.py:185-187``).  The packed rows,
    scales and biases are concatenated once; ``forward`` returns the two layers' outputs as views of one buffer."""

    def __init__(self, first: ScaledTernaryMatmul, second: ScaledTernaryMatmul):
        super().__init__()
        if not (first.built and second.built):
            raise RuntimeError("build() both layers (call them once) before pairing them")
        if first.orig_shape[1] != second.orig_shape[1]:
            raise ValueError(f"the layers read different input widths: {first.orig_shape[1]} and {second.orig_shape[1]}")
        if first.act is not None:
            raise ValueError("the first layer of a pair must have no activation")
        if first.clip_val != second.clip_val:
            raise ValueError("the layers clip at different values")
        self.fused_act = _fusable(second.act)
        self.post_act = None if (second.act is None or self.fused_act is not None) else second.act
        self.k = first.orig_shape[1]
        self.split = first.units
        self.units = first.units + second.units
        self.clip_val = first.clip_val
        rb = ops.packed_row_bytes(self.k)
        self.register_buffer("pw", torch.cat([first.pw[:, :rb], second.pw[:, :rb]]).contiguous())
        self.register_buffer("scale", torch.cat([first.scale, second.scale]).contiguous())
        self.register_buffer("bias", torch.cat([first.bias, second.bias]).contiguous())

    def forward(self, x: torch.Tensor):
        y = ops.ternary_matmul(x, self.pw, self.scale, self.bias, self.clip_val, activation=self.fused_act, act_begin=self.split)
        a, b = y[..., : self.split], y[..., self.split:]
        return a, (b if self.post_act is None else self.post_act(b))


class ScaledTernaryTf(_PackedDenseBase):
    """``impl/tf/moondream/ternary.py:178-250``: the whole weight matrix flattened and packed in the
    planar layout B (``pack_ternary2``), unpacked on every call, dense matmul."""

    pack_format = "b"  # the reference hard-wires layout B (``:188-190``); "a" selects ``pack_ternary``

    def build(self, input_shape):
        k = int(input_shape[-1])
        self.orig_shape = (self.units, k)
        self.tot_size = self.units * k
        rem = self.tot_size % 4
        final = self.tot_size // 4 + ((4 - rem) if rem else 0)  # ternary.py:201-202
        self.register_buffer("pw", torch.randint(0, 255, (final,), dtype=torch.uint8, device=self.scale.device))
        self.built = True

    def pack_fn(self, t: torch.Tensor) -> torch.Tensor:
        return ops.pack_ternary_planar(t.to(torch.int8), self.pack_format)

    def unpack_fn(self, pw: torch.Tensor, dtype=torch.float32) -> torch.Tensor:
        return ops.unpack_ternary_planar(pw, self.pack_format, dtype=dtype)

    def set_weights(self, weights):
        scale, bias, pw = weights
        if not self.built:
            raise RuntimeError("build() the layer (call it once) before set_weights, as in Keras")
        dev = self.scale.device
        pw = torch.as_tensor(pw, dtype=torch.uint8, device=dev).reshape(-1)
        if pw.numel() * 4 < self.tot_size:
            raise ValueError(f"{pw.numel()} packed bytes cannot hold {self.tot_size} weights")
        self.scale = torch.as_tensor(scale, dtype=torch.float32, device=dev).reshape(self.units).clone()
        self.bias = torch.as_tensor(bias, dtype=torch.float32, device=dev).reshape(self.units).clone()
        self.pw = pw.contiguous().clone()

    def get_weights(self):
        return [self.scale, self.bias, self.pw]

    def call(self, x: torch.Tensor) -> torch.Tensor:
        unpacked = self.unpack_fn(self.pw, dtype=x.dtype)
        w = unpacked[: self.tot_size].reshape(self.orig_shape)
        y = self._epilogue(x @ w.t())
        return self._post(y, False)


class ScaledTernary(_PackedDenseBase):
    """``impl/tf/moondream/ternary.py:33-54``: larq ``QuantDense`` holding a latent float kernel ``[K, units]``
    that is ternarized at call time.  Weight order ``[scale, bias, kernel]`` (``utils/conversion.py:41``).
    Here the kernel is ternarized and packed when it is set, and calls run the fused matmul."""

    def build(self, input_shape):
        k = int(input_shape[-1])
        self.orig_shape = (self.units, k)
        self.register_buffer("kernel", torch.empty((k, self.units), device=self.scale.device).uniform_(-1, 1))
        self._repack()
        self.built = True

    def _repack(self):
        self.pw = ops.ternarize_pack(self.kernel.t().contiguous(), 0.5, "ternary")

    def set_weights(self, weights):
        scale, bias, kernel = weights
        if not self.built:
            raise RuntimeError("build() the layer (call it once) before set_weights, as in Keras")
        dev = self.scale.device
        kernel = torch.as_tensor(kernel, dtype=torch.float32, device=dev)
        if tuple(kernel.shape) != (self.orig_shape[1], self.units):
            raise ValueError(f"kernel of shape {tuple(kernel.shape)} does not fit [K={self.orig_shape[1]}, units={self.units}]")
        self.scale = torch.as_tensor(scale, dtype=torch.float32, device=dev).reshape(self.units).clone()
        self.bias = torch.as_tensor(bias, dtype=torch.float32, device=dev).reshape(self.units).clone()
        self.kernel = kernel.clone()
        self._repack()

    def get_weights(self):
        return [self.scale, self.bias, self.kernel]

    def call(self, x: torch.Tensor) -> torch.Tensor:
        fused = _fusable(self.act)
        y = ops.ternary_matmul(x, self.pw, self.scale, self.bias, self.clip_val, activation=fused)
        return self._post(y, fused is not None)
