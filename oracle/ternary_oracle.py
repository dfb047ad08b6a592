"""CPU oracle for the ternary/binary quantized-matmul path (numpy).

TEST INFRASTRUCTURE ONLY.  This file restates, on the CPU, the arithmetic of
the reference's Python layers so that the HIP kernels can be checked against
it.  Nothing in the product package imports it; only tests/, the smoke check
and the CPU-baseline leg of the benchmark do.

Pinning status
--------------
* ternarize / binarize, the scale fits and the dense forward are pinned by
  golden vectors generated from the reference's own torch code
  (``oracle/gen_golden.py`` imports ``/root/reference/binary/modules.py`` and
  ``matrix_mul_tests.py``'s ``ternary_mm``; vectors live in ``tests/golden``).
* pack formats A, B and C are restated from source text
  (``impl/tf/moondream/ternary.py``; TensorFlow is not installed, so the file
  cannot be imported).  They are pinned by the reference's own round-trip
  property (``impl/tf/moondream/ternary_tests.py:17-25``, length 5031) and by
  hand-worked byte examples kept in ``tests/golden/pack_examples.json``.
* the byte layout of the external ``larq_compute_engine`` fork's
  ``ops.pack_fn`` / ``ops.unpack_ternary`` / ``ops.ternary_matmul`` is NOT in the
  reference and no reference test pins it: **parity unpinned** for that layout.
  This build uses format C (row-interleaved), which has the shape the
  reference allocates for those ops (``ternary.py:270``).

Every function cites the reference file:line it follows (paths relative to
``/root/reference``).
"""
from __future__ import annotations

import numpy as np

# --------------------------------------------------------------------------
# quantizers
# --------------------------------------------------------------------------


def ternarize(w: np.ndarray, threshold: float = 0.5) -> np.ndarray:
    """``(w >= t) - (w <= -t)`` with inclusive thresholds, in w's dtype.

    Follows ``binary/modules.py:21-26`` (``tri_step_ste``) and is the torch twin
    of ``ternary_act = SteTern(0.5)`` at ``impl/tf/moondream/ternary.py:25``.
    NaN compares false on both sides and therefore maps to 0.

    The reference adds a straight-through term, ``q + x - x.detach()``.  In
    float32 ``(q + x)`` rounds, so the reference's *training-time* weights can
    be one ulp off +-1 (0.99999994) and +-inf gives NaN; the deployed path
    (``utils/conversion.py:71-72``: quantize, cast to int8, pack) stores exact
    {-1,0,+1}.  This oracle returns the exact ternary value q: it equals the
    reference bit for bit whenever ``q + x`` is exact (inputs on the grid k/64)
    and ``rint`` of it otherwise.
    """
    w = np.asarray(w)
    t = w.dtype.type(threshold)
    return (w >= t).astype(w.dtype) - (w <= -t).astype(w.dtype)


def binarize_sign(w: np.ndarray) -> np.ndarray:
    """``(w >= 0) * 2 - 1`` in {-1,+1}.  ``binary/modules.py:5-10`` (``sign_ste``)."""
    w = np.asarray(w)
    return ((w >= 0).astype(w.dtype) * w.dtype.type(2) - w.dtype.type(1)).astype(w.dtype)


def binarize_step(w: np.ndarray) -> np.ndarray:
    """``(w >= 0)`` in {0,1}.  ``binary/modules.py:13-18`` (``step_ste``)."""
    w = np.asarray(w)
    return (w >= 0).astype(w.dtype)


QUANTIZERS = {"ternary": ternarize, "binary": binarize_sign, "binary01": binarize_step}

# --------------------------------------------------------------------------
# 2-bit pack layouts
# --------------------------------------------------------------------------


def _pad4_1d(t: np.ndarray) -> np.ndarray:
    to_add = (4 - t.shape[-1] % 4) % 4
    if to_add:
        t = np.concatenate([t, np.zeros((to_add,), dtype=t.dtype)], axis=-1)
    return t


def pack_format_a(t: np.ndarray) -> np.ndarray:
    """Planar, sign-bit code.  ``impl/tf/moondream/ternary.py:57-78`` (``pack_ternary``).

    1-D input, zero-padded to a multiple of 4; with S = len/4, byte j holds
    element ``i*S + j`` in bits ``[2i+1:2i]``; code bit0 = (t == +1),
    bit1 = (t == -1).
    """
    t = _pad4_1d(np.asarray(t).reshape(-1))
    s = t.shape[0] // 4
    packed = np.zeros((s,), dtype=np.uint8)
    for i in range(4):
        sl = t[i * s:(i + 1) * s]
        val = (sl == 1).astype(np.uint8) ^ ((sl == -1).astype(np.uint8) << 1)
        packed |= (val << (2 * i)).astype(np.uint8)
    return packed


def unpack_format_a(packed: np.ndarray, dtype=np.float32) -> np.ndarray:
    """``impl/tf/moondream/ternary.py:101-110`` (``unpack_ternary_iterative``):
    plane i = ``((b >> 2i) & 1) - (((b >> 2i) & 3) >> 1)``, planes concatenated.
    Code 0b11 decodes to 0."""
    packed = np.asarray(packed, dtype=np.uint8)
    res = []
    for i in range(4):
        sh = packed >> np.uint8(2 * i)
        pos = sh & np.uint8(1)
        neg = (sh & np.uint8(3)) >> np.uint8(1)
        res.append(pos.astype(dtype) - neg.astype(dtype))
    return np.concatenate(res, axis=-1)


def pack_format_b(t: np.ndarray) -> np.ndarray:
    """Planar, offset code.  ``impl/tf/moondream/ternary.py:114-133`` (``pack_ternary2``).

    Whole tensor flattened, zero-padded to a multiple of 4, planar as format
    A, code = t + 1 (-1 -> 0, 0 -> 1, +1 -> 2)."""
    t = _pad4_1d(np.asarray(t).reshape(-1))
    s = t.shape[0] // 4
    packed = np.zeros((s,), dtype=np.uint8)
    for i in range(4):
        sl = t[i * s:(i + 1) * s]
        val = ((sl + 1).astype(np.int64) & 0xFF).astype(np.uint8)
        packed |= (val << (2 * i)).astype(np.uint8)
    return packed


def unpack_format_b(packed: np.ndarray, dtype=np.float32) -> np.ndarray:
    """``impl/tf/moondream/ternary.py:136-156`` (``unpack_ternary_iterative2``) and
    its early-concat twin ``:159-175``: plane i = ``((b >> 2i) & 3) - 1``.
    Code 0b11 decodes to +2 (the reference does not reject it)."""
    packed = np.asarray(packed, dtype=np.uint8)
    res = [(((packed >> np.uint8(2 * i)) & np.uint8(3)).astype(dtype) - dtype(1)) for i in range(4)]
    return np.concatenate(res, axis=-1)


def packed_row_bytes(k: int) -> int:
    """Bytes this build writes per packed row: ceil(K/4)."""
    return (int(k) + 3) // 4


def reference_row_bytes(k: int) -> int:
    """Bytes the reference *allocates* per packed row: ``K//4 + (4 - K%4 if K%4 else 0)``
    (``impl/tf/moondream/ternary.py:269-270``, ``:363-364``); never smaller than
    ``packed_row_bytes``."""
    k = int(k)
    r = k % 4
    return k // 4 + ((4 - r) if r else 0)


def pack_format_c(t: np.ndarray) -> np.ndarray:
    """Row-interleaved, offset code: the layout of this build's packed weights.

    ``impl/tf/moondream/ternary.py:374-398`` (``ScaledTernaryPackedMM.pack_fn``):
    byte ``[r, j] = sum_i (t[r, 4j+i] + 1) << 2i``; the last axis is zero-padded
    (value 0 -> code 1) to a multiple of 4.  (The reference's own padding line
    ``:379-380`` only works for 1-D input; the intent - pad the last axis with
    zeros - is what is restated here.)
    """
    t = np.asarray(t)
    k = t.shape[-1]
    to_add = (4 - k % 4) % 4
    if to_add:
        t = np.concatenate([t, np.zeros(t.shape[:-1] + (to_add,), dtype=t.dtype)], axis=-1)
    codes = ((t + 1).astype(np.int64) & 0x3).astype(np.uint8)
    codes = codes.reshape(t.shape[:-1] + (t.shape[-1] // 4, 4))
    return (codes[..., 0] | (codes[..., 1] << 2) | (codes[..., 2] << 4) | (codes[..., 3] << 6)).astype(np.uint8)


def unpack_format_c(packed: np.ndarray, orig_dim: int, dtype=np.float32) -> np.ndarray:
    """Inverse of format C: ``w[r, 4j+i] = ((b[r,j] >> 2i) & 3) - 1``, first
    ``orig_dim`` columns.  Decode rule from ``impl/tf/moondream/ternary.py:507-520``;
    call shape ``ops.unpack_ternary(pw, orig_dim)`` from ``:334``.  Code 0b11 -> +2."""
    packed = np.asarray(packed, dtype=np.uint8)
    planes = [((packed >> np.uint8(2 * i)) & np.uint8(3)).astype(np.int8) - np.int8(1) for i in range(4)]
    w = np.stack(planes, axis=-1).reshape(packed.shape[:-1] + (packed.shape[-1] * 4,))
    return w[..., :orig_dim].astype(dtype)


# --------------------------------------------------------------------------
# activations and the fused op
# --------------------------------------------------------------------------


def gelu_tanh(x: np.ndarray) -> np.ndarray:
    """tanh-approximate GELU, ``0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3)))``:
    the activation the reference attaches to fc1 (``utils/conversion.py:185-187``,
    ``keras.activations.gelu(approximate=True)``).  Computed in float64."""
    x64 = np.asarray(x, dtype=np.float64)
    y = 0.5 * x64 * (1.0 + np.tanh(np.sqrt(2.0 / np.pi) * (x64 + 0.044715 * x64 ** 3)))
    return y


ACT_NONE, ACT_GELU_TANH = 0, 1


def ternary_matmul(x, pw, scale, bias, clip_val, k=None, act=ACT_NONE, acc_dtype=np.float64, out_dtype=np.float32, act_begin=0):
    """``clip((x @ unpack(pw)^T) * scale + bias, +-clip)`` then optional activation.

    Order of operations follows ``binary/modules.py:101-109`` (matmul, ``*= scale``,
    ``+= bias``, clip) and the unfused twin ``impl/tf/moondream/ternary.py:334-342``;
    the fused call is ``ops.ternary_matmul(x, pw, scale, bias, clip)`` at ``:281``
    with the activation applied after the op (``:298-299``).

    The contraction is done in ``acc_dtype`` (float64 = error-free reference for
    random data; for inputs on a dyadic grid float32 is already exact); the
    epilogue multiply, add and clip are done in float32 - two roundings, no fused
    multiply-add - exactly like the reference's separate ``*=`` and ``+=``.

    ``act_begin``: only output channels ``>= act_begin`` get the activation (two layers that share
    their input, ``impl/tf/moondream/layers.py:334-347``, stored as one packed matrix).
    """
    x = np.asarray(x)
    kk = x.shape[-1] if k is None else k
    w = unpack_format_c(pw, kk, dtype=acc_dtype)
    acc = x.astype(acc_dtype).reshape(-1, kk) @ w.T
    acc = acc.astype(np.float32)
    y = acc * np.asarray(scale, dtype=np.float32)[None, :]
    if bias is not None:
        y = y + np.asarray(bias, dtype=np.float32)[None, :]
    c = np.float32(abs(clip_val))
    y = np.clip(y, -c, c)
    if act == ACT_GELU_TANH:
        y[:, act_begin:] = gelu_tanh(y[:, act_begin:]).astype(np.float32)
    return y.reshape(x.shape[:-1] + (w.shape[0],)).astype(out_dtype)


def round_to(a, dtype_name):
    """float32 values rounded to nearest-even in "f32" / "f16" / "bf16" and returned as float32 (numpy has no bfloat16)."""
    a = np.asarray(a, dtype=np.float32)
    if dtype_name == "f32":
        return a
    if dtype_name == "f16":
        return a.astype(np.float16).astype(np.float32)
    if dtype_name == "bf16":
        bits = a.view(np.uint32).astype(np.uint64)
        rounded = ((bits + 0x7FFF + ((bits >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
        out = rounded.view(np.float32).copy()
        nan = np.isnan(a)
        out[nan] = np.nan
        return out
    raise ValueError(f"Unknown dtype name: {dtype_name}")


def ternary_matmul_residual(x, pw, scale, bias, clip_val, add1, add2=None, out="f32", **kw):
    """The fused op followed by the decoder block's closing sum, ``attn_outputs + feed_forward_hidden_states + residual``
    (``impl/tf/moondream/layers.py:345-347``), as element-wise additions in the output type: ``(y + add1) + add2`` with the
    product and every sum rounded to ``out`` ("f32": the reference's own float32 arithmetic).  ``kw``: as ``ternary_matmul``."""
    y = round_to(ternary_matmul(x, pw, scale, bias, clip_val, **kw), out)
    y = round_to(y + np.asarray(add1, dtype=np.float32).reshape(y.shape), out)
    if add2 is not None:
        y = round_to(y + np.asarray(add2, dtype=np.float32).reshape(y.shape), out)
    return y


def scaled_linear_forward(x, weights, scale, bias, maxval=100.0, quantization="ternary", acc_dtype=np.float64):
    """Dense forward of the torch layers: quantize the latent weights on every
    call, ``F.linear``, ``*= scale``, ``+= bias``, clip.
    ``binary/modules.py:63-71`` (binary), ``:82-90`` (binary01), ``:101-109`` (ternary)."""
    q = QUANTIZERS[quantization](np.asarray(weights, dtype=np.float32))
    x = np.asarray(x)
    acc = (x.astype(acc_dtype).reshape(-1, x.shape[-1]) @ q.astype(acc_dtype).T).astype(np.float32)
    y = acc * np.asarray(scale, dtype=np.float32)[None, :]
    if bias is not None:
        y = y + np.asarray(bias, dtype=np.float32)[None, :]
    c = np.float32(abs(maxval))
    return np.clip(y, -c, c).reshape(x.shape[:-1] + (q.shape[0],))


def smooth_binary_forward(x, weights, bias, maxval=100.0, training=False, acc_dtype=np.float64):
    """``SmoothlyBinarizedLinear.forward`` (``binary/modules.py:139-146``): evaluation binarizes the latent weights
    by ``(w >= 0) * 2 - 1`` (``:143``; zero and negative zero give +1), training clips them to [-1, 1] (``:141``);
    then ``F.linear(x, w, bias)`` and the clip (``:144-145``).  There is no scale parameter."""
    w = np.asarray(weights, dtype=np.float32)
    w = np.clip(w, -1.0, 1.0) if training else binarize_sign(w)
    x = np.asarray(x)
    y = (x.astype(acc_dtype).reshape(-1, x.shape[-1]) @ w.astype(acc_dtype).T)
    if bias is not None:
        y = y + np.asarray(bias, dtype=acc_dtype)[None, :]
    c = np.float32(abs(maxval))
    return np.clip(y.astype(np.float32), -c, c).reshape(x.shape[:-1] + (w.shape[0],))


def ternary_mm(m: np.ndarray, tm: np.ndarray) -> np.ndarray:
    """Multiplication-free select-and-sum: ``res[:, i] = sum_{tm[i,k]>0} m[:,k] -
    sum_{tm[i,k]<0} m[:,k]``.  ``matrix_mul_tests.py:29-37``.

    The reference allocates ``res`` as ``[M, tm.shape[1]]`` but fills
    ``tm.shape[0]`` columns (identical for its square inputs); the output here is
    ``[M, tm.shape[0]]``.  Sums are taken in float32 in column order like the
    reference's ``sp.sum(dim=1) - sn.sum(dim=1)`` (two separate sums, then one
    subtraction)."""
    m = np.asarray(m, dtype=np.float32)
    tm = np.asarray(tm)
    res = np.zeros((m.shape[0], tm.shape[0]), dtype=np.float32)
    pos, neg = tm > 0, tm < 0
    for i in range(tm.shape[0]):
        res[:, i] = m[:, pos[i]].sum(axis=1, dtype=np.float32) - m[:, neg[i]].sum(axis=1, dtype=np.float32)
    return res


# --------------------------------------------------------------------------
# scale fits (one-time quantization of a dense layer)
# --------------------------------------------------------------------------


def fit_ternary_scale(w: np.ndarray, neuron_scale: str = "independent", kmeans_iter: int = 10):
    """Masked-mean fixed point of ``binary/modules.py:275-293``.

    ``m0 = mean|w|`` per row (one global row when ``neuron_scale == "uniform"``,
    ``:281-282``; any other string is treated as per-row exactly like the
    reference, which special-cases only ``"uniform"``); repeat ``kmeans_iter``
    times ``mask = (|w| - m) > -m/2``, ``m = sum(|w| * mask) / count(mask)``
    (float32 sums, ``:288``).  Returns ``(latent, scale)`` with
    ``latent = clip(w * (1/m), -1, 1)`` (``:290-292``) and ``scale = m``
    broadcast to ``[N]`` (``:293``).  A row whose mask empties divides by zero
    (NaN), as in the reference.
    """
    w = np.asarray(w, dtype=np.float32)
    abs_w = np.abs(w)
    if neuron_scale == "uniform":
        abs_w = abs_w.reshape(1, -1)
    m = abs_w.mean(axis=1, dtype=np.float32)
    with np.errstate(invalid="ignore", divide="ignore"):
        for _ in range(kmeans_iter):
            d = abs_w - m[:, None]
            mask = d > -m[:, None] / np.float32(2.0)
            m = (abs_w * mask).sum(axis=1, dtype=np.float32) / mask.sum(axis=1).astype(np.float32)
        inv = (np.float32(1.0) / m).astype(np.float32)
    latent = np.clip(w * inv[:, None], -1.0, 1.0).astype(np.float32)
    scale = np.broadcast_to(m, (w.shape[0],)).astype(np.float32).copy()
    return latent, scale


def fit_binary_scale(w: np.ndarray, quantization: str = "binary", neuron_scale: str = "uniform", scaling="none"):
    """Binary / binary01 scale fit and latent weights.

    ``binary/modules.py:193-231`` (binary: ``scale = (sum_+ w - sum_- w) / count``)
    and ``:232-266`` (binary01: ``scale = sum_+ w / count``); ``count`` is all
    elements (uniform) or the row length (independent).  ``scaling`` pre-scales
    the latent weights: none / float / "std" (``1/std`` with torch's unbiased
    estimator) / "natural" (``1/scale`` per row) (``:214-226``); latent =
    ``clip(w * s, -1, 1)``.  Unknown strings raise ValueError like the reference.
    """
    w = np.asarray(w, dtype=np.float32)
    n, k = w.shape
    pos = np.where(w > 0, w, np.float32(0))
    neg = np.where(w < 0, w, np.float32(0))
    if quantization not in ("binary", "binary01"):
        raise ValueError(f"Unknown quantization: {quantization}")
    if neuron_scale == "uniform":
        num = pos.sum(dtype=np.float32) - (neg.sum(dtype=np.float32) if quantization == "binary" else np.float32(0))
        scale = np.full((n,), num / np.float32(n * k), dtype=np.float32)
    elif neuron_scale == "independent":
        num = pos.sum(axis=1, dtype=np.float32) - (neg.sum(axis=1, dtype=np.float32) if quantization == "binary" else np.float32(0))
        scale = (num / np.float32(k)).astype(np.float32)
    else:
        raise ValueError(f"Unknown neuron scaling method: {neuron_scale}")
    if scaling is None or scaling == "none":
        s = np.float32(1.0)
    elif isinstance(scaling, float):
        s = np.float32(scaling)
    elif scaling == "std":
        s = np.float32(1.0) / w.std(ddof=1, dtype=np.float64).astype(np.float32)
    elif scaling == "natural":
        s = (np.float32(1.0) / scale)[:, None]
    else:
        raise ValueError(f"Unknown scaling method: {scaling}")
    latent = np.clip(w * s, -1.0, 1.0).astype(np.float32)
    return latent, scale


# --------------------------------------------------------------------------
# plumbing ops of the op library
# --------------------------------------------------------------------------


# ---------------------------------------------------------------------------------------------
# binarized activations: AND + popcount form (larq input_quantizer="ste_sign", impl/tf/moondream/ternary.py:36)
# ---------------------------------------------------------------------------------------------

def sign_bits(x: np.ndarray) -> np.ndarray:
    """Sign bits of the activations, 32 per word, LSB first: bit = (x >= 0), i.e. sign_ste = +1
    (binary/modules.py:5-10); pad bits 0.  x [M, K] -> uint32 [M, ceil(K/32)]."""
    x = np.asarray(x)
    m, k = x.shape
    words = (k + 31) // 32
    bits = np.zeros((m, words * 32), dtype=np.uint8)
    bits[:, :k] = (x >= 0)
    return np.packbits(bits.reshape(m, words, 32), axis=-1, bitorder="little").view(np.uint32).reshape(m, words)


def ternary_bitplanes(pw: np.ndarray, k: int):
    """(pos, neg) bit planes of packed format-C rows: pos = weight +1 (codes 2 and 3), neg = weight -1 (code 0) -
    the two planes the reference's pack_ternary builds per element (impl/tf/moondream/ternary.py:72-75).
    uint8 [N, >= ceil(K/4)] -> two uint32 [N, ceil(K/32)]."""
    pw = np.asarray(pw, dtype=np.uint8)
    n = pw.shape[0]
    codes = np.stack([(pw >> (2 * i)) & 3 for i in range(4)], axis=-1).reshape(n, -1)[:, :k]
    words = (k + 31) // 32
    out = []
    for plane in ((codes >= 2), (codes == 0)):
        bits = np.zeros((n, words * 32), dtype=np.uint8)
        bits[:, :k] = plane
        out.append(np.packbits(bits.reshape(n, words, 32), axis=-1, bitorder="little").view(np.uint32).reshape(n, words))
    return out[0], out[1]


def popcount_matmul(xb: np.ndarray, pos: np.ndarray, neg: np.ndarray, scale, bias, clip, act: int = ACT_NONE) -> np.ndarray:
    """sum_k s t = 2 (popc(xb & pos) - popc(xb & neg)) - (popc(pos) - popc(neg)), then the matmul's float32
    epilogue (binary/modules.py:101-109 order)."""
    def popc(a):
        return np.unpackbits(np.ascontiguousarray(a).view(np.uint8), axis=-1).sum(axis=-1, dtype=np.int64)
    p = popc(xb[:, None, :] & pos[None, :, :])
    q = popc(xb[:, None, :] & neg[None, :, :])
    dot = 2 * (p - q) - (popc(pos) - popc(neg))[None, :]
    y = dot.astype(np.float32) * np.asarray(scale, dtype=np.float32)[None, :]
    if bias is not None:
        y = y + np.asarray(bias, dtype=np.float32)[None, :]
    c = np.float32(abs(clip))
    y = np.clip(y, -c, c)
    if act == ACT_GELU_TANH:
        y = gelu_tanh(y.astype(np.float64)).astype(np.float32)
    return y


def zero_out(x: np.ndarray) -> np.ndarray:
    """Keep flat element 0, zero the rest.  Export at ``custom_ops_mod.py:29``;
    behaviour pinned by ``zero_out_tests.py:9-10`` and
    ``native/tensorflow_custom_ops/python/ops/zero_out_ops_test.py:33-34``."""
    x = np.asarray(x, dtype=np.int32)
    out = np.zeros_like(x)
    if x.size:
        out.reshape(-1)[0] = x.reshape(-1)[0]
    return out


def prevent_folding(constant: np.ndarray, witness) -> np.ndarray:
    """Identity on ``constant`` (uint8); ``witness`` only creates a data dependency.
    Export at ``custom_ops_mod.py:30``; Python twin ``impl/tf/moondream/ternary.py:28-30``."""
    return np.array(constant, dtype=np.uint8, copy=True)
