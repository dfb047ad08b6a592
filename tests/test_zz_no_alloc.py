"""tvl_ternary_matmul_noalloc (the entry point for callers that cannot allocate: scratch memory built into the library)
through the raw C ABI, against tvl_ternary_matmul_act_from with a caller-owned workspace: bit for bit the same output,
on two calls in a row with the second smaller than the first (the built-in buffer keeps the first call's contents)."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import ternary_oracle as O

pytestmark = pytest.mark.gpu


def _problem(dev, seed, m, n, k, xdt, grid=True):
    rng = np.random.default_rng(seed)
    x = rng.integers(-64, 65, size=(m, k)).astype(np.float32) / 64.0 if grid else rng.normal(size=(m, k)).astype(np.float32)
    pw = rng.integers(0, 171, size=(n, (k + 3) // 4)).astype(np.uint8)
    scale = rng.uniform(0.5, 1.5, size=(n,)).astype(np.float32)
    bias = rng.normal(size=(n,)).astype(np.float32)
    t = lambda a: torch.from_numpy(a).to(dev)
    return x, pw, scale, bias, t(x).to(xdt), t(pw), t(scale), t(bias)


def _both(ops, dev, m, n, k, xdt, act=0, act_begin=0, seed=0, grid=True):
    lib = ops._lib
    x, pw, scale, bias, xt, pwt, st, bt = _problem(dev, seed + m + n + k, m, n, k, xdt, grid)
    xd = ops._DTYPES[xdt]
    need = int(lib.tvl_ternary_matmul_workspace_bytes_for(xd, m, n, k))
    ws = torch.empty((max(need, 1),), dtype=torch.uint8, device=dev)
    y_ref = torch.full((m, n), float("nan"), dtype=torch.float32, device=dev)
    y_new = torch.full((m, n), float("nan"), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    rc = lib.tvl_ternary_matmul_act_from(xt.data_ptr(), xd, m, k, 0, pwt.data_ptr(), n, 0, st.data_ptr(), bt.data_ptr(), ctypes.c_float(100.0),
                                         act, act_begin, y_ref.data_ptr(), ops.F32, 0, ws.data_ptr() if need else None, need, stream)
    assert rc == 0, lib.tvl_last_error()
    before = torch.cuda.memory_allocated(dev)
    rc = lib.tvl_ternary_matmul_noalloc(xt.data_ptr(), xd, m, k, 0, pwt.data_ptr(), n, 0, st.data_ptr(), bt.data_ptr(), ctypes.c_float(100.0),
                                        act, act_begin, y_new.data_ptr(), ops.F32, 0, stream)
    assert rc == 0, lib.tvl_last_error()
    assert torch.cuda.memory_allocated(dev) == before
    torch.cuda.synchronize()
    assert torch.equal(y_ref, y_new), (m, n, k, xdt)                       # no NaN survives: every element was written
    if grid and act == 0:
        want = O.ternary_matmul(x, pw, scale, bias, 100.0, k=k)
        assert np.array_equal(y_new.cpu().numpy(), want), (m, n, k, xdt)   # and both equal the oracle
    return need


def test_noalloc_matches_workspace_entry_point_twice_in_a_row(ops, dev):
    lib = ops._lib
    limit = int(lib.tvl_noalloc_scratch_bytes())
    assert limit == 32 << 20
    # (first call, then a SMALLER one that uses the same kind of scratch) per user of the scratch memory
    pairs = [
        [(12, 2048, 16384, torch.float16), (8, 512, 8192, torch.float16)],          # few rows, split-K partial sums
        [(300, 1024, 4096, torch.float16), (130, 512, 2048, torch.bfloat16)],       # matrix-core tiles with split-K
        [(200, 768, 1024, torch.float32), (70, 96, 256, torch.float32)],            # float32 activations: three bf16 planes
        [(400, 600, 1000, torch.float16), (120, 300, 500, torch.bfloat16)],         # k % 64 != 0: zero-padded copies
    ]
    used = 0
    for big, small in pairs:
        for (m, n, k, xdt) in (big, small):
            need = _both(ops, dev, m, n, k, xdt)
            assert need <= limit
            used += need > 0
    assert used >= 6, "the shapes above are meant to need scratch memory"
    # random data (sums that depend on the order of addition: any difference in kernel choice or k split would show), one shape per
    # user of the scratch memory
    for (m, n, k, xdt) in [(12, 2048, 16384, torch.float16), (300, 1024, 4096, torch.float16), (200, 768, 1024, torch.float32),
                           (400, 600, 1000, torch.bfloat16), (40, 6144, 2048, torch.float16), (2, 8192, 8192, torch.bfloat16)]:
        _both(ops, dev, m, n, k, xdt, grid=False)
    # the fused activation from a channel on, random data, and a shape that needs no scratch at all
    _both(ops, dev, 12, 1024, 8192, torch.float16, act=ops.ACT_GELU_TANH, act_begin=300, grid=False)
    assert _both(ops, dev, 4096, 4096, 256, torch.float16) == 0
    _both(ops, dev, 1, 512, 1024, torch.float16)


def test_noalloc_leaves_the_devices_free_memory_unchanged(ops, dev):
    """hipMemGetInfo's free bytes (what a hipMalloc / hipFree inside the library would move; torch's own counters only see torch's
    allocator) before and after calls that use the scratch memory, after a warm-up call of each kind."""
    lib = ops._lib
    stream = torch.cuda.current_stream().cuda_stream
    shapes = [(12, 2048, 16384, torch.float16), (300, 1024, 4096, torch.float16), (200, 768, 1024, torch.float32), (400, 600, 1000, torch.float16)]
    probs = []
    for (m, n, k, xdt) in shapes:
        _, _, _, _, xt, pwt, st, bt = _problem(dev, 7 + m, m, n, k, xdt)
        probs.append((m, n, k, ops._DTYPES[xdt], xt, pwt, st, bt, torch.empty((m, n), dtype=torch.float32, device=dev)))

    def run_all():
        for (m, n, k, xd, xt, pwt, st, bt, y) in probs:
            rc = lib.tvl_ternary_matmul_noalloc(xt.data_ptr(), xd, m, k, 0, pwt.data_ptr(), n, 0, st.data_ptr(), bt.data_ptr(), ctypes.c_float(100.0),
                                                0, 0, y.data_ptr(), ops.F32, 0, stream)
            assert rc == 0, lib.tvl_last_error()

    run_all()                                   # warm-up: code objects loaded, the scratch address looked up
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info(dev)[0]
    for _ in range(3):
        run_all()
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info(dev)[0] == free_before


def test_noalloc_rejects_shapes_beyond_the_builtin_buffer(ops, dev):
    lib = ops._lib
    m, n, k = 4096, 8192, 4096                        # float32 activations: the three planes alone are 96 MiB
    assert int(lib.tvl_ternary_matmul_workspace_bytes_for(ops.F32, m, n, k)) > int(lib.tvl_noalloc_scratch_bytes())
    x = torch.zeros((8, 8), device=dev)               # never read: the call must fail before any launch
    rc = lib.tvl_ternary_matmul_noalloc(x.data_ptr(), ops.F32, m, k, 0, x.data_ptr(), n, 0, x.data_ptr(), None, ctypes.c_float(1.0), 0, 0,
                                        x.data_ptr(), ops.F32, 0, torch.cuda.current_stream().cuda_stream)
    assert rc == ops.ERR_INVALID
    msg = lib.tvl_last_error().decode()
    assert "scratch" in msg and str(32 << 20) in msg
    torch.cuda.synchronize()
    # argument errors are reported like the other entry points
    rc = lib.tvl_ternary_matmul_noalloc(None, ops.F16, 4, 64, 0, None, 8, 0, None, None, ctypes.c_float(1.0), 0, 0, None, ops.F16, 0, None)
    assert rc == ops.ERR_INVALID and b"NULL" in lib.tvl_last_error()
