"""Operator surface of the ternary / binary quantized-matmul path, backed by libternary_hip.so.

Mirror of the reference's op-library loader: ``custom_ops_mod.py:27-30`` loads ``custom_ops.so``
from the directory of the loader module and exposes snake_case callables (``zero_out``,
``prevent_folding``); the packed ops ``pack_fn`` / ``unpack_ternary`` / ``ternary_matmul`` are
what the reference imports from its larq_compute_engine fork
(``impl/tf/moondream/ternary.py:22,281,334``, ``utils/conversion.py:72``).  Same names, same
positional argument order.  TensorFlow is not part of this stack: tensors are torch tensors
that live on an AMD GPU, and every op runs a hand-written HIP kernel through the C ABI
declared in ``include/tvl.h`` - there is no CPU or PyTorch fallback; a missing library or a
CPU tensor is an error.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIBRARY_PATH = os.path.join(_HERE, "libternary_hip.so")

# enums of include/tvl.h
F32, F16, BF16, I8, I32 = 0, 1, 2, 3, 4
Q_TERNARY, Q_SIGN, Q_STEP = 0, 1, 2
ACT_NONE, ACT_GELU_TANH = 0, 1
PLANAR_A, PLANAR_B = 0, 1
ERR_INVALID, ERR_HIP = 1, 2

_DTYPES = {torch.float32: F32, torch.float16: F16, torch.bfloat16: BF16, torch.int8: I8, torch.int32: I32}
_QUANT = {"ternary": Q_TERNARY, "binary": Q_SIGN, "sign": Q_SIGN, "binary01": Q_STEP, "step": Q_STEP}

_c_i64, _c_int, _c_f32, _c_vp, _c_sz = ctypes.c_int64, ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t

# name -> (restype, argtypes); also the list of symbols the library must export
SIGNATURES = {
    "tvl_version": (_c_int, []),
    "tvl_last_error": (ctypes.c_char_p, []),
    "tvl_build_info": (ctypes.c_char_p, []),
    "tvl_quantize_pack": (_c_int, [_c_vp, _c_int, _c_i64, _c_i64, _c_i64, _c_int, _c_f32, _c_vp, _c_i64, _c_vp]),
    "tvl_quantize_pack_batch": (_c_int, [_c_vp, _c_i64, _c_int, _c_int, _c_f32, _c_vp]),
    "tvl_pack_i8": (_c_int, [_c_vp, _c_i64, _c_i64, _c_i64, _c_vp, _c_i64, _c_vp]),
    "tvl_unpack": (_c_int, [_c_vp, _c_i64, _c_i64, _c_i64, _c_vp, _c_int, _c_i64, _c_vp]),
    "tvl_pack_planar_i8": (_c_int, [_c_vp, _c_i64, _c_int, _c_vp, _c_vp]),
    "tvl_unpack_planar": (_c_int, [_c_vp, _c_i64, _c_int, _c_vp, _c_int, _c_vp]),
    "tvl_quantize": (_c_int, [_c_vp, _c_int, _c_i64, _c_int, _c_f32, _c_vp, _c_int, _c_vp]),
    "tvl_ternary_matmul": (_c_int, [_c_vp, _c_int, _c_i64, _c_i64, _c_i64, _c_vp, _c_i64, _c_i64, _c_vp, _c_vp, _c_f32,
                                    _c_int, _c_vp, _c_int, _c_i64, _c_vp, _c_sz, _c_vp]),
    "tvl_ternary_matmul_act_from": (_c_int, [_c_vp, _c_int, _c_i64, _c_i64, _c_i64, _c_vp, _c_i64, _c_i64, _c_vp, _c_vp, _c_f32,
                                             _c_int, _c_i64, _c_vp, _c_int, _c_i64, _c_vp, _c_sz, _c_vp]),
    "tvl_ternary_matmul_residual": (_c_int, [_c_vp, _c_int, _c_i64, _c_i64, _c_i64, _c_vp, _c_i64, _c_i64, _c_vp, _c_vp, _c_f32,
                                             _c_int, _c_i64, _c_vp, _c_i64, _c_vp, _c_i64, _c_vp, _c_int, _c_i64, _c_vp, _c_sz, _c_vp]),
    "tvl_ternary_matmul_route": (_c_int, [_c_int, _c_i64, _c_i64, _c_i64, _c_sz, ctypes.c_char_p, _c_sz]),
    "tvl_ternary_matmul_noalloc": (_c_int, [_c_vp, _c_int, _c_i64, _c_i64, _c_i64, _c_vp, _c_i64, _c_i64, _c_vp, _c_vp, _c_f32,
                                            _c_int, _c_i64, _c_vp, _c_int, _c_i64, _c_vp]),
    "tvl_noalloc_scratch_bytes": (_c_sz, []),
    "tvl_ternary_matmul_workspace_bytes": (_c_sz, [_c_i64, _c_i64, _c_i64]),
    "tvl_ternary_matmul_workspace_bytes_for": (_c_sz, [_c_int, _c_i64, _c_i64, _c_i64]),
    "tvl_sign_pack_bits": (_c_int, [_c_vp, _c_int, _c_i64, _c_i64, _c_i64, _c_vp, _c_i64, _c_vp]),
    "tvl_ternary_bitplanes": (_c_int, [_c_vp, _c_i64, _c_i64, _c_i64, _c_vp, _c_vp, _c_i64, _c_vp]),
    "tvl_popcount_matmul": (_c_int, [_c_vp, _c_i64, _c_i64, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_vp, _c_vp, _c_f32, _c_int,
                                     _c_vp, _c_int, _c_i64, _c_vp]),
    "tvl_ternary_mm_f32": (_c_int, [_c_vp, _c_i64, _c_i64, _c_i64, _c_vp, _c_i64, _c_i64, _c_vp, _c_i64, _c_vp]),
    "tvl_fit_ternary_scale_f32": (_c_int, [_c_vp, _c_i64, _c_i64, _c_i64, _c_int, _c_int, _c_vp, _c_vp, _c_i64, _c_vp,
                                           _c_sz, _c_vp]),
    "tvl_fit_binary_scale_f32": (_c_int, [_c_vp, _c_i64, _c_i64, _c_i64, _c_int, _c_int, _c_f32, _c_int, _c_vp, _c_vp,
                                          _c_i64, _c_vp, _c_sz, _c_vp]),
    "tvl_fit_scale_workspace_bytes": (_c_sz, [_c_i64, _c_i64]),
    "tvl_zero_out_i32": (_c_int, [_c_vp, _c_i64, _c_vp, _c_vp]),
    "tvl_prevent_folding_u8": (_c_int, [_c_vp, _c_i64, _c_vp, _c_vp, _c_vp]),
}


def _load_library() -> ctypes.CDLL:
    if not os.path.exists(LIBRARY_PATH):
        raise ImportError(
            f"{LIBRARY_PATH} not found: build it with `make -C {os.path.join(_HERE, 'csrc')}` "
            "(hipcc --offload-arch=gfx950).  There is no fallback implementation.")
    lib = ctypes.CDLL(LIBRARY_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    return lib


_lib = _load_library()


def library_version() -> int:
    return int(_lib.tvl_version())


def build_info() -> str:
    return _lib.tvl_build_info().decode()


def _check(status: int) -> None:
    if status == 0:
        return
    msg = _lib.tvl_last_error().decode(errors="replace")
    if status == ERR_INVALID:
        raise ValueError(msg)
    raise RuntimeError(msg)


def _gpu(t: torch.Tensor, name: str) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor, got {type(t).__name__}")
    if not t.is_cuda:
        raise ValueError(f"{name}: tensor must live on the GPU (got device {t.device}); this op has no CPU path")
    return t


def _stream(t: torch.Tensor) -> int:
    return int(torch.cuda.current_stream(t.device).cuda_stream)


def _dt(t: torch.Tensor, name: str) -> int:
    try:
        return _DTYPES[t.dtype]
    except KeyError:
        raise ValueError(f"{name}: unsupported dtype {t.dtype}") from None


def _rows(t: torch.Tensor, name: str) -> torch.Tensor:
    """2-D view with unit stride in the last dimension (row stride passed as ld)."""
    if t.dim() != 2:
        raise ValueError(f"{name}: expected a 2-D tensor, got shape {tuple(t.shape)}")
    if t.shape[1] > 1 and t.stride(1) != 1:
        t = t.contiguous()
    if t.shape[0] > 1 and t.stride(0) < t.shape[1]:
        t = t.contiguous()
    return t


def _ld(t: torch.Tensor) -> int:
    return int(t.stride(0)) if t.shape[0] > 1 else int(max(t.shape[1], 1))


# --------------------------------------------------------------------------------------------
# plumbing ops (custom_ops_mod.py:29-30)
# --------------------------------------------------------------------------------------------

def zero_out(x) -> torch.Tensor:
    """Keep the first (flat) element, zero the rest.  ``custom_ops_mod.py:29``; int32 like the
    reference op's tests (``zero_out_tests.py:9``).  Accepts a tensor or a nested list."""
    if not isinstance(x, torch.Tensor):
        x = torch.tensor(x, dtype=torch.int32, device="cuda")
    x = _gpu(x, "zero_out")
    if x.dtype != torch.int32:
        raise ValueError(f"zero_out: expected int32, got {x.dtype}")
    x = x.contiguous()
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _check(_lib.tvl_zero_out_i32(x.data_ptr(), x.numel(), out.data_ptr(), _stream(x)))
    return out


def prevent_folding(constant: torch.Tensor, witness: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Identity on a uint8 ``constant``; ``witness`` is only a fake data dependency
    (``custom_ops_mod.py:30``, Python twin ``impl/tf/moondream/ternary.py:28-30``)."""
    constant = _gpu(constant, "prevent_folding")
    if constant.dtype != torch.uint8:
        raise ValueError(f"prevent_folding: expected uint8, got {constant.dtype}")
    constant = constant.contiguous()
    out = torch.empty_like(constant)
    wptr = witness.data_ptr() if isinstance(witness, torch.Tensor) and witness.is_cuda else None
    with torch.cuda.device(constant.device):
        _check(_lib.tvl_prevent_folding_u8(constant.data_ptr(), constant.numel(), wptr, out.data_ptr(), _stream(constant)))
    return out


# --------------------------------------------------------------------------------------------
# pack / unpack
# --------------------------------------------------------------------------------------------

def packed_row_bytes(k: int) -> int:
    return (int(k) + 3) // 4


def reference_row_bytes(k: int) -> int:
    """Row length the reference allocates for ``pw`` (``impl/tf/moondream/ternary.py:269-270``)."""
    r = int(k) % 4
    return int(k) // 4 + ((4 - r) if r else 0)


def pack_fn(t: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """int8 ``[N, K]`` in {-1,0,+1} -> uint8 ``[N, ceil(K/4)]`` (row-interleaved, code = value+1).
    ``ops.pack_fn`` at ``utils/conversion.py:72``; layout of ``impl/tf/moondream/ternary.py:374-398``."""
    t = _gpu(t, "pack_fn")
    if t.dtype != torch.int8:
        raise ValueError(f"pack_fn: expected int8 values in {{-1,0,1}}, got {t.dtype}")
    t = _rows(t, "pack_fn")
    n, k = t.shape
    if out is None:
        out = torch.empty((n, packed_row_bytes(k)), dtype=torch.uint8, device=t.device)
    out = _check_pw(out, n, k, "pack_fn(out)")
    with torch.cuda.device(t.device):
        _check(_lib.tvl_pack_i8(t.data_ptr(), n, k, _ld(t), out.data_ptr(), _ld(out), _stream(t)))
    return out


def _check_pw(pw: torch.Tensor, n: Optional[int], k: int, name: str) -> torch.Tensor:
    pw = _gpu(pw, name)
    if pw.dtype != torch.uint8 or pw.dim() != 2:
        raise ValueError(f"{name}: packed weights must be a 2-D uint8 tensor, got {pw.dtype} {tuple(pw.shape)}")
    if pw.shape[1] > 1 and pw.stride(1) != 1:
        raise ValueError(f"{name}: packed weights must have unit stride in the last dimension")
    if pw.shape[0] > 1 and pw.stride(0) < pw.shape[1]:
        raise ValueError(f"{name}: packed rows overlap (row stride {pw.stride(0)} < {pw.shape[1]} bytes): pass a dense or row-padded tensor")
    if n is not None and pw.shape[0] != n:
        raise ValueError(f"{name}: packed weights have {pw.shape[0]} rows, expected {n}")
    if pw.shape[1] < packed_row_bytes(k):
        raise ValueError(f"{name}: packed rows hold {pw.shape[1]} bytes but orig_dim={k} needs {packed_row_bytes(k)}")
    return pw


def unpack_ternary(pw: torch.Tensor, orig_dim: int, dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """uint8 ``[N, >=ceil(K/4)]`` -> ``[N, orig_dim]`` holding {-1,0,+1} (code 3 -> +2).
    ``ops.unpack_ternary(pw, orig_dim)`` at ``impl/tf/moondream/ternary.py:334``."""
    orig_dim = int(orig_dim)
    if orig_dim < 0:
        raise ValueError("unpack_ternary: orig_dim must be >= 0")
    pw = _check_pw(pw, None, orig_dim, "unpack_ternary")
    n = pw.shape[0]
    out = torch.empty((n, orig_dim), dtype=dtype, device=pw.device)
    with torch.cuda.device(pw.device):
        _check(_lib.tvl_unpack(pw.data_ptr(), n, orig_dim, _ld(pw), out.data_ptr(), _dt(out, "unpack_ternary"), _ld(out),
                               _stream(pw)))
    return out


def ternarize_pack(w: torch.Tensor, threshold: float = 0.5, quantization: str = "ternary",
                   out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Latent weights ``[N, K]`` (f32/f16/bf16) -> packed uint8 ``[N, ceil(K/4)]`` in one pass:
    quantize (``binary/modules.py:21-26`` ternary, ``:5-10`` sign, ``:13-18`` step) and pack.
    Fuses ``ternary_act(w) -> int8 -> ops.pack_fn`` of ``utils/conversion.py:71-72``."""
    w = _gpu(w, "ternarize_pack")
    if quantization not in _QUANT:
        raise ValueError(f"Unknown quantization: {quantization}")
    w = _rows(w.detach(), "ternarize_pack")
    n, k = w.shape
    if out is None:
        out = torch.empty((n, packed_row_bytes(k)), dtype=torch.uint8, device=w.device)
    out = _check_pw(out, n, k, "ternarize_pack(out)")
    with torch.cuda.device(w.device):
        _check(_lib.tvl_quantize_pack(w.data_ptr(), _dt(w, "ternarize_pack"), n, k, _ld(w), _QUANT[quantization],
                                      float(threshold), out.data_ptr(), _ld(out), _stream(w)))
    return out


class _PackItem(ctypes.Structure):   # tvl_pack_item of include/tvl.h
    _fields_ = [("w", _c_vp), ("n", _c_i64), ("k", _c_i64), ("ld_w", _c_i64), ("pw", _c_vp), ("ld_pw", _c_i64)]


def ternarize_pack_batch(ws, threshold: float = 0.5, quantization: str = "ternary", outs=None):
    """:func:`ternarize_pack` for a list of latent weight matrices (same dtype, same device) in as few launches as
    possible - every layer of a checkpoint into its place in a packed arena (``utils/conversion.py:126-137`` calls the
    per-layer transfer in a loop).  ``outs``: optional list of uint8 ``[N_i, >= ceil(K_i / 4)]`` destinations (views into an
    arena); returns the list of packed tensors."""
    if quantization not in _QUANT:
        raise ValueError(f"Unknown quantization: {quantization}")
    ws = [_rows(_gpu(w, "ternarize_pack_batch").detach(), "ternarize_pack_batch") for w in ws]
    if not ws:
        return []
    dev, dt = ws[0].device, ws[0].dtype
    if any(w.device != dev or w.dtype != dt for w in ws):
        raise ValueError("ternarize_pack_batch: all matrices must share one device and one dtype")
    if outs is None:
        outs = [torch.empty((w.shape[0], packed_row_bytes(w.shape[1])), dtype=torch.uint8, device=dev) for w in ws]
    if len(outs) != len(ws):
        raise ValueError("ternarize_pack_batch: `outs` must have one entry per matrix")
    outs = [_check_pw(o, w.shape[0], w.shape[1], "ternarize_pack_batch(out)") for o, w in zip(outs, ws)]
    items = (_PackItem * len(ws))()
    for it, w, o in zip(items, ws, outs):
        it.w, it.n, it.k, it.ld_w, it.pw, it.ld_pw = w.data_ptr(), w.shape[0], w.shape[1], _ld(w), o.data_ptr(), _ld(o)
    with torch.cuda.device(dev):
        _check(_lib.tvl_quantize_pack_batch(ctypes.cast(items, _c_vp), len(ws), _dt(ws[0], "ternarize_pack_batch"), _QUANT[quantization],
                                            float(threshold), _stream(ws[0])))
    return outs


def quantize(w: torch.Tensor, quantization: str = "ternary", threshold: float = 0.5,
             dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """Dense quantizer value: {-1,0,+1} / {-1,+1} / {0,1} in w's dtype (or int8)."""
    w = _gpu(w, "quantize")
    if quantization not in _QUANT:
        raise ValueError(f"Unknown quantization: {quantization}")
    w = w.detach().contiguous()
    out = torch.empty(w.shape, dtype=dtype or w.dtype, device=w.device)
    with torch.cuda.device(w.device):
        _check(_lib.tvl_quantize(w.data_ptr(), _dt(w, "quantize"), w.numel(), _QUANT[quantization], float(threshold),
                                 out.data_ptr(), _dt(out, "quantize"), _stream(w)))
    return out


def pack_ternary_planar(t: torch.Tensor, fmt: str = "b") -> torch.Tensor:
    """Planar 1-D layouts of the pure-TF layer: ``pack_ternary`` (fmt "a",
    ``impl/tf/moondream/ternary.py:57-78``) and ``pack_ternary2`` (fmt "b", ``:114-133``)."""
    t = _gpu(t, "pack_ternary_planar")
    if t.dtype != torch.int8:
        raise ValueError(f"pack_ternary_planar: expected int8, got {t.dtype}")
    t = t.contiguous().reshape(-1)
    out = torch.empty(((t.numel() + 3) // 4,), dtype=torch.uint8, device=t.device)
    with torch.cuda.device(t.device):
        _check(_lib.tvl_pack_planar_i8(t.data_ptr(), t.numel(), _planar(fmt), out.data_ptr(), _stream(t)))
    return out


def unpack_ternary_planar(pw: torch.Tensor, fmt: str = "b", dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """Inverse of :func:`pack_ternary_planar`: ``4 * len(pw)`` values (the caller slices).
    ``unpack_ternary_iterative`` / ``unpack_ternary_iterative2`` (``ternary.py:101-110``, ``:136-156``)."""
    pw = _gpu(pw, "unpack_ternary_planar")
    if pw.dtype != torch.uint8:
        raise ValueError(f"unpack_ternary_planar: expected uint8, got {pw.dtype}")
    pw = pw.contiguous().reshape(-1)
    out = torch.empty((pw.numel() * 4,), dtype=dtype, device=pw.device)
    with torch.cuda.device(pw.device):
        _check(_lib.tvl_unpack_planar(pw.data_ptr(), pw.numel(), _planar(fmt), out.data_ptr(), _dt(out, "unpack_ternary_planar"),
                                      _stream(pw)))
    return out


def _planar(fmt: str) -> int:
    try:
        return {"a": PLANAR_A, "b": PLANAR_B}[str(fmt).lower()]
    except KeyError:
        raise ValueError(f"Unknown planar format: {fmt}") from None


# --------------------------------------------------------------------------------------------
# the fused op
# --------------------------------------------------------------------------------------------

_ACTS = {None: ACT_NONE, "none": ACT_NONE, "gelu_tanh": ACT_GELU_TANH, "gelu": ACT_GELU_TANH}

# Split-K scratch memory, one grow-only buffer per (device, stream): calls on one stream are ordered, so
# they can share it; calls on different streams must not.  A buffer that is outgrown is kept alive for the life of
# the process (`_retired_workspaces`): a HIP graph captured while it was current has its address baked in, and a
# replay must never write into memory the allocator has meanwhile handed to another tensor.
_matmul_workspaces = {}
_retired_workspaces = []


_workspace_need = {}


def _matmul_workspace(device: torch.device, stream: int, x_dtype: int, m: int, n: int, k: int):
    shape_key = (x_dtype, m, n, k)
    need = _workspace_need.get(shape_key)
    if need is None:
        need = _workspace_need[shape_key] = int(_lib.tvl_ternary_matmul_workspace_bytes_for(x_dtype, m, n, k))
    if need == 0:
        return None, 0
    key = (device.index, stream)
    ws = _matmul_workspaces.get(key)
    if ws is None or ws.numel() < need:
        if ws is not None:
            _retired_workspaces.append(ws)
            need = max(need, 2 * ws.numel())   # geometric growth bounds what the retired buffers add up to
        ws = torch.empty((need,), dtype=torch.uint8, device=device)
        _matmul_workspaces[key] = ws
    return ws.data_ptr(), ws.numel()



def ternary_matmul(x: torch.Tensor, pw: torch.Tensor, scale: torch.Tensor, bias: Optional[torch.Tensor], clip_val: float,
                   activation: Optional[str] = None, out_dtype: Optional[torch.dtype] = None,
                   out: Optional[torch.Tensor] = None, act_begin: int = 0, residual: Optional[torch.Tensor] = None,
                   residual2: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``act(clip((x @ unpack(pw)^T) * scale + bias, +-clip_val))``  (``+ residual + residual2`` when given).

    ``ops.ternary_matmul(x, pw, scale, bias, clip)`` at ``impl/tf/moondream/ternary.py:281``; the
    order of operations is ``binary/modules.py:104-108``.  ``x``: ``[..., K]`` f16 / bf16 / f32;
    ``pw``: uint8 ``[N, >= ceil(K/4)]``; ``scale``, ``bias``: ``[N]``.  The optional activation
    (the reference applies it after the op, ``ternary.py:298-299``) is fused into the kernel.
    ``act_begin``: output channels below it get no activation - two Dense layers on one input (the Phi
    block's Wqkv and fc1, ``impl/tf/moondream/layers.py:334-347``) stored as one packed matrix, one launch.
    ``residual`` / ``residual2``: ``[..., N]`` tensors of the output's dtype added after the activation, ``(y + residual) +
    residual2`` with every sum rounded to the output dtype - the bits of ``y + residual + residual2`` in torch: the decoder
    block's ``attn_outputs + feed_forward_hidden_states + residual`` (``impl/tf/moondream/layers.py:347``) as part of fc2's call.
    """
    x = _gpu(x, "ternary_matmul(x)")
    if activation not in _ACTS:
        raise ValueError(f"Unknown activation: {activation}")
    if x.dim() < 1:
        raise ValueError("ternary_matmul: x must have at least one dimension")
    k = x.shape[-1]
    pw = _check_pw(pw, None, k, "ternary_matmul(pw)")
    n = pw.shape[0]
    lead = x.shape[:-1]
    x2 = _rows(x.reshape(-1, k), "ternary_matmul(x)")
    m = x2.shape[0]
    scale = _vec(scale, n, x.device, "scale")
    bias = _vec(bias, n, x.device, "bias") if bias is not None else None
    ydt = out_dtype or x.dtype
    if out is None:
        y = torch.empty((m, n), dtype=ydt, device=x.device)
    else:
        y = _rows(_gpu(out, "ternary_matmul(out)").reshape(-1, n), "ternary_matmul(out)")
        if y.shape[0] != m or y.data_ptr() != out.data_ptr():
            raise ValueError("ternary_matmul: `out` must be a contiguous [..., N] tensor matching x")
    xdt = _dt(x2, "ternary_matmul(x)")
    act_begin = int(act_begin)
    if act_begin < 0 or act_begin > n:
        raise ValueError(f"ternary_matmul: act_begin={act_begin} outside [0, {n}]")

    adds = []
    for name, r in (("residual", residual), ("residual2", residual2)):
        if r is None:
            continue
        r = _gpu(r, f"ternary_matmul({name})")
        if r.dtype != y.dtype or r.device != x.device or r.shape[-1] != n or r.numel() != m * n:
            raise ValueError(f"ternary_matmul: {name} must be a [..., {n}] {y.dtype} tensor with {m} rows on {x.device}, got {tuple(r.shape)} {r.dtype}")
        adds.append(_rows(r.reshape(-1, n), f"ternary_matmul({name})"))

    def launch():
        stream = _stream(x)
        ws_ptr, ws_bytes = _matmul_workspace(x.device, stream, xdt, m, n, k)
        if adds:
            a1, a2 = adds[0], adds[1] if len(adds) > 1 else None
            _check(_lib.tvl_ternary_matmul_residual(x2.data_ptr(), xdt, m, k, _ld(x2), pw.data_ptr(), n, _ld(pw),
                                                    scale.data_ptr(), bias.data_ptr() if bias is not None else None, float(clip_val),
                                                    _ACTS[activation], act_begin, a1.data_ptr(), _ld(a1),
                                                    a2.data_ptr() if a2 is not None else None, _ld(a2) if a2 is not None else 0,
                                                    y.data_ptr(), _dt(y, "ternary_matmul(out)"), _ld(y), ws_ptr, ws_bytes, stream))
            return
        _check(_lib.tvl_ternary_matmul_act_from(x2.data_ptr(), xdt, m, k, _ld(x2), pw.data_ptr(), n, _ld(pw),
                                                scale.data_ptr(), bias.data_ptr() if bias is not None else None, float(clip_val),
                                                _ACTS[activation], act_begin, y.data_ptr(), _dt(y, "ternary_matmul(out)"), _ld(y),
                                                ws_ptr, ws_bytes, stream))

    if torch.cuda.current_device() == x.device.index:   # skip the device guard (several microseconds) when it is a no-op
        launch()
    else:
        with torch.cuda.device(x.device):
            launch()
    if out is not None:
        return out
    return y if len(lead) == 1 else y.reshape(*lead, n)


def matmul_route(x_dtype: torch.dtype, m: int, n: int, k: int, workspace_bytes: Optional[int] = None) -> str:
    """Names of the kernels ``ternary_matmul`` launches for dense operands of this shape (``tvl_ternary_matmul_route``); the
    workspace defaults to what ``ternary_matmul`` itself provides."""
    xdt = _DTYPES[x_dtype]
    if workspace_bytes is None:
        workspace_bytes = int(_lib.tvl_ternary_matmul_workspace_bytes_for(xdt, m, n, k))
    buf = ctypes.create_string_buffer(512)
    _check(_lib.tvl_ternary_matmul_route(xdt, m, n, k, workspace_bytes, buf, len(buf)))
    return buf.value.decode()


class PreparedMatmul:
    """``ternary_matmul`` on fixed buffers with every check done once: ``p = PreparedMatmul(x, pw, scale, bias, clip)``,
    then ``p()`` re-launches on the current stream reading ``x`` / writing ``p.out`` in place (update ``x`` with
    ``copy_``).  For eager decode loops, where the per-call argument checks cost more than the kernel; a loop that
    can be captured in a HIP graph does not need it."""

    def __init__(self, x: torch.Tensor, pw: torch.Tensor, scale: torch.Tensor, bias: Optional[torch.Tensor], clip_val: float,
                 activation: Optional[str] = None, out_dtype: Optional[torch.dtype] = None, out: Optional[torch.Tensor] = None,
                 act_begin: int = 0):
        self.out = ternary_matmul(x, pw, scale, bias, clip_val, activation, out_dtype, out, act_begin)   # validates, allocates, runs once
        k, n = x.shape[-1], pw.shape[0]
        x2 = x.reshape(-1, k)
        if x2.data_ptr() != x.data_ptr() or (x2.shape[0] > 1 and x2.stride(0) < k) or (k > 1 and x2.stride(1) != 1):
            raise ValueError("PreparedMatmul: x must be a dense [..., K] tensor (it is read in place on every call)")
        y2 = self.out.reshape(-1, n)
        self._scale = _vec(scale, n, x.device, "scale")
        self._bias = _vec(bias, n, x.device, "bias") if bias is not None else None
        self._device = x.device
        self._shape = (_dt(x2, "x"), x2.shape[0], n, k)
        # its own scratch memory, never replaced: safe to capture in a HIP graph on any stream and to replay for as long as
        # this object lives (two PreparedMatmul objects never share scratch, so they may run on different streams)
        need = int(_lib.tvl_ternary_matmul_workspace_bytes_for(*self._shape))
        self._ws = torch.empty((need,), dtype=torch.uint8, device=x.device) if need else None
        self._keep = (x, pw, self.out, self._ws)
        self._args_head = (x2.data_ptr(), _dt(x2, "x"), x2.shape[0], k, _ld(x2), pw.data_ptr(), n, _ld(pw), self._scale.data_ptr(),
                           self._bias.data_ptr() if self._bias is not None else None, float(clip_val), _ACTS[activation],
                           int(act_begin), y2.data_ptr(), _dt(y2, "out"), _ld(y2))

    def __call__(self) -> torch.Tensor:
        stream = int(torch.cuda.current_stream(self._device).cuda_stream)
        ws_ptr, ws_bytes = (self._ws.data_ptr(), self._ws.numel()) if self._ws is not None else (None, 0)
        if torch.cuda.current_device() == self._device.index:
            _check(_lib.tvl_ternary_matmul_act_from(*self._args_head, ws_ptr, ws_bytes, stream))
        else:   # the stream handle belongs to the tensors' device: make it current for the launch
            with torch.cuda.device(self._device):
                _check(_lib.tvl_ternary_matmul_act_from(*self._args_head, ws_ptr, ws_bytes, stream))
        return self.out


def _vec(v: torch.Tensor, n: int, device, name: str) -> torch.Tensor:
    if (isinstance(v, torch.Tensor) and v.dtype == torch.float32 and v.dim() == 1 and v.shape[0] == n and v.device == device
            and v.is_contiguous() and not v.requires_grad):
        return v   # the common case, without building views
    v = _gpu(v, f"ternary_matmul({name})")
    if v.device != device:
        raise ValueError(f"ternary_matmul: {name} is on {v.device}, x on {device}")
    if v.numel() != n:
        raise ValueError(f"ternary_matmul: {name} has {v.numel()} elements, expected {n}")
    v = v.detach().reshape(-1)
    if v.dtype != torch.float32 or not v.is_contiguous():
        v = v.float().contiguous()
    return v


_F32_DENORM_MIN = 1.401298464324817e-45   # smallest positive float32: (w >= it) is (w > 0), (w <= -it) is (w < 0)


_UNIT_SCALES = {}


def _unit_scale(n: int, device: torch.device) -> torch.Tensor:
    """A vector of n ones on `device`, kept between calls (a few sizes at most: every call of ternary_mm would otherwise
    launch a fill kernel of its own)."""
    key = (device.type, device.index, n)
    t = _UNIT_SCALES.get(key)
    if t is None and torch.cuda.is_current_stream_capturing():   # a tensor born inside a capture belongs to that graph: not kept
        return torch.ones((n,), dtype=torch.float32, device=device)
    if t is None:
        if len(_UNIT_SCALES) >= 32:
            _UNIT_SCALES.clear()
        t = _UNIT_SCALES[key] = torch.ones((n,), dtype=torch.float32, device=device)
    return t


def ternary_mm(m: torch.Tensor, tm: torch.Tensor, method: str = "auto") -> torch.Tensor:
    """Multiplication-free select-and-sum with dense weights: ``m @ sign(tm)^T``
    (``matrix_mul_tests.py:29-37``; float32).

    ``method="select"``: one kernel that keeps the reference's two separate sums (columns where ``tm > 0``, columns
    where ``tm < 0``) and subtracts them.  ``method="matmul"``: ``sign(tm)`` packed to 2 bits (one pass), then the
    packed matmul with float32 activations on the matrix cores - the same float32 additions in another order
    (bit-identical whenever the sums are exact): 1024^3 in 41 us instead of 317, 4096^3 in 0.33 ms instead of 5.4.  ``"auto"`` picks by size."""
    m = _rows(_gpu(m, "ternary_mm(m)"), "ternary_mm(m)")
    tm = _rows(_gpu(tm, "ternary_mm(tm)"), "ternary_mm(tm)")
    if m.dtype != torch.float32 or tm.dtype != torch.float32:
        raise ValueError("ternary_mm: float32 only")
    if m.shape[1] != tm.shape[1]:
        raise ValueError(f"ternary_mm: inner dimensions differ ({m.shape[1]} vs {tm.shape[1]})")
    if method not in ("auto", "select", "matmul"):
        raise ValueError(f"Unknown ternary_mm method: {method}")
    if method == "auto":   # measured (HIP-graph replay, us): 64^3 select 18.3 / matmul 9.7, 256^3 68 / 14.8, 1024^3 313 / 32.7
        method = "matmul" if m.shape[0] * tm.shape[0] * m.shape[1] >= (1 << 16) else "select"
    if method == "matmul":
        pw = ternarize_pack(tm, _F32_DENORM_MIN)
        return ternary_matmul(m, pw, _unit_scale(tm.shape[0], m.device), None, float("inf"))
    res = torch.empty((m.shape[0], tm.shape[0]), dtype=torch.float32, device=m.device)
    with torch.cuda.device(m.device):
        _check(_lib.tvl_ternary_mm_f32(m.data_ptr(), m.shape[0], m.shape[1], _ld(m), tm.data_ptr(), tm.shape[0], _ld(tm),
                                       res.data_ptr(), _ld(res), _stream(m)))
    return res


# --------------------------------------------------------------------------------------------
# binarized activations (AND + popcount)
# --------------------------------------------------------------------------------------------

def sign_pack_bits(x: torch.Tensor) -> torch.Tensor:
    """Sign bits of x, 32 per int32 word, LSB first: bit = (x >= 0) (``sign_ste``, ``binary/modules.py:5-10``)."""
    x2 = _rows(_gpu(x, "sign_pack_bits(x)").reshape(-1, x.shape[-1]), "sign_pack_bits(x)")
    m, k = x2.shape
    xb = torch.empty((m, (k + 31) // 32), dtype=torch.int32, device=x.device)
    with torch.cuda.device(x.device):
        _check(_lib.tvl_sign_pack_bits(x2.data_ptr(), _dt(x2, "sign_pack_bits(x)"), m, k, _ld(x2), xb.data_ptr(), _ld(xb), _stream(x)))
    return xb.reshape(*x.shape[:-1], xb.shape[1])


def ternary_bitplanes(pw: torch.Tensor, orig_dim: int):
    """(pos, neg) bit planes (int32 ``[N, ceil(K/32)]``) of packed rows: weight +1 / weight -1."""
    pw = _rows(_gpu(pw, "ternary_bitplanes(pw)"), "ternary_bitplanes(pw)")
    n, k = pw.shape[0], int(orig_dim)
    if pw.dtype != torch.uint8 or pw.shape[1] < (k + 3) // 4:
        raise ValueError(f"ternary_bitplanes: pw must be uint8 [N, >= {(k + 3) // 4}], got {pw.dtype} {tuple(pw.shape)}")
    pos = torch.empty((n, (k + 31) // 32), dtype=torch.int32, device=pw.device)
    neg = torch.empty_like(pos)
    with torch.cuda.device(pw.device):
        _check(_lib.tvl_ternary_bitplanes(pw.data_ptr(), n, k, _ld(pw), pos.data_ptr(), neg.data_ptr(), _ld(pos), _stream(pw)))
    return pos, neg


def popcount_matmul(xb: torch.Tensor, pos: torch.Tensor, neg: torch.Tensor, orig_dim: int, scale: torch.Tensor,
                    bias: Optional[torch.Tensor], clip_val: float, activation: Optional[str] = None,
                    out_dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """``act(clip((sign(x) @ t^T) * scale + bias))`` from bit planes: AND + popcount, integer-exact accumulation."""
    lead = xb.shape[:-1]
    xb2 = _rows(_gpu(xb, "popcount_matmul(xb)").reshape(-1, xb.shape[-1]), "popcount_matmul(xb)")
    pos = _rows(_gpu(pos, "popcount_matmul(pos)"), "popcount_matmul(pos)")
    neg = _rows(_gpu(neg, "popcount_matmul(neg)"), "popcount_matmul(neg)")
    k, words = int(orig_dim), (int(orig_dim) + 31) // 32
    if any(t.dtype != torch.int32 for t in (xb2, pos, neg)) or xb2.shape[1] < words or pos.shape != neg.shape or pos.shape[1] < words:
        raise ValueError("popcount_matmul: xb / pos / neg must be int32 bit planes with >= ceil(K/32) words per row")
    if activation not in _ACTS:
        raise ValueError(f"Unknown activation: {activation}")
    m, n = xb2.shape[0], pos.shape[0]
    scale = _vec(scale, n, xb.device, "scale")
    bias = _vec(bias, n, xb.device, "bias") if bias is not None else None
    y = torch.empty((m, n), dtype=out_dtype, device=xb.device)
    with torch.cuda.device(xb.device):
        _check(_lib.tvl_popcount_matmul(xb2.data_ptr(), m, _ld(xb2), pos.data_ptr(), neg.data_ptr(), n, _ld(pos), k, scale.data_ptr(),
                                        bias.data_ptr() if bias is not None else None, float(clip_val), _ACTS[activation],
                                        y.data_ptr(), _dt(y, "popcount_matmul(out)"), _ld(y), _stream(xb)))
    return y.reshape(*lead, n)


def binarized_ternary_matmul(x: torch.Tensor, pw: torch.Tensor, scale: torch.Tensor, bias: Optional[torch.Tensor], clip_val: float,
                             activation: Optional[str] = None, method: str = "auto", planes=None) -> torch.Tensor:
    """The matmul with sign-binarized activations (larq ``input_quantizer="ste_sign"``, ``ternary.py:36``).

    ``method="popcount"``: bit planes + AND/popcount kernels; ``"mfma"``: sign(x) as +-1 f16 through the fused matrix-core
    matmul - the same integers, so the two agree bit for bit; ``"auto"`` takes popcount for up to 8 rows (it reads the
    same 2 bits per weight and skips the decode) and the matrix cores otherwise (4x the popcount rate per CU).
    ``planes`` = a cached ``ternary_bitplanes(pw, K)`` result."""
    k = x.shape[-1]
    rows = x.numel() // max(1, k)
    if method == "auto":
        method = "popcount" if rows <= 8 else "mfma"
    if method == "popcount":
        pos, neg = planes if planes is not None else ternary_bitplanes(pw, k)
        return popcount_matmul(sign_pack_bits(x), pos, neg, k, scale, bias, clip_val, activation)
    if method != "mfma":
        raise ValueError(f"Unknown method: {method}")
    s = torch.where(x >= 0, 1.0, -1.0).to(torch.float16)
    return ternary_matmul(s, pw, scale, bias, clip_val, activation=activation, out_dtype=torch.float32)


# --------------------------------------------------------------------------------------------
# scale fits
# --------------------------------------------------------------------------------------------

def _workspace(w: torch.Tensor) -> torch.Tensor:
    nbytes = int(_lib.tvl_fit_scale_workspace_bytes(w.shape[0], w.shape[1]))
    return torch.empty((nbytes + 7) // 8, dtype=torch.float64, device=w.device)


def fit_ternary_scale(w: torch.Tensor, neuron_scale: str = "independent", kmeans_iter: int = 10, return_latent: bool = True):
    """Ternary scale by the reference's masked-mean iteration (``binary/modules.py:280-293``).
    Returns ``(latent, scale)``: ``latent = clip(w / m, -1, 1)`` ``[N, K]`` and ``scale = m`` ``[N]``.
    Only ``neuron_scale == "uniform"`` fits one global scale; anything else is per row, like the
    reference (``:281``)."""
    w = _rows(_gpu(w, "fit_ternary_scale").detach(), "fit_ternary_scale")
    if w.dtype != torch.float32:
        w = w.float()
    n, k = w.shape
    scale = torch.empty((n,), dtype=torch.float32, device=w.device)
    latent = torch.empty((n, k), dtype=torch.float32, device=w.device) if return_latent else None
    ws = _workspace(w)
    with torch.cuda.device(w.device):
        _check(_lib.tvl_fit_ternary_scale_f32(w.data_ptr(), n, k, _ld(w), 1 if neuron_scale == "uniform" else 0,
                                              int(kmeans_iter), scale.data_ptr(),
                                              latent.data_ptr() if latent is not None else None, k, ws.data_ptr(),
                                              ws.numel() * 8, _stream(w)))
    return latent, scale


def fit_binary_scale(w: torch.Tensor, quantization: str = "binary", neuron_scale: str = "uniform", scaling="none",
                     return_latent: bool = True):
    """Binary / binary01 scale and latent weights (``binary/modules.py:193-266``)."""
    if quantization not in ("binary", "binary01"):
        raise ValueError(f"Unknown quantization: {quantization}")
    if neuron_scale not in ("uniform", "independent"):
        raise ValueError(f"Unknown neuron scaling method: {neuron_scale}")
    w = _rows(_gpu(w, "fit_binary_scale").detach(), "fit_binary_scale")
    if w.dtype != torch.float32:
        w = w.float()
    natural, prescale = 0, 1.0
    if scaling is None or scaling == "none":
        pass
    elif isinstance(scaling, float):
        prescale = scaling
    elif scaling == "std":
        prescale = 1.0 / float(w.std())  # torch's unbiased estimator, like the reference (modules.py:219-220)
    elif scaling == "natural":
        natural = 1
    else:
        raise ValueError(f"Unknown scaling method: {scaling}")
    n, k = w.shape
    scale = torch.empty((n,), dtype=torch.float32, device=w.device)
    latent = torch.empty((n, k), dtype=torch.float32, device=w.device) if return_latent else None
    ws = _workspace(w)
    with torch.cuda.device(w.device):
        _check(_lib.tvl_fit_binary_scale_f32(w.data_ptr(), n, k, _ld(w), _QUANT[quantization],
                                             1 if neuron_scale == "uniform" else 0, float(prescale), natural,
                                             scale.data_ptr(), latent.data_ptr() if latent is not None else None, k,
                                             ws.data_ptr(), ws.numel() * 8, _stream(w)))
    return latent, scale
