import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def ops():
    """The product's operator module (HIP-backed).  Importing it loads the C-ABI library."""
    import ternary_vllm_amd
    return ternary_vllm_amd.custom_ops_mod


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(scope="session")
def c_oracle():
    """The C restatement of the fused matmul (oracle/ternary_oracle.c) as ``f(x, pw, scale, bias, clip, k=None)`` ->
    float32 [m, n]: the checker for full-size outputs, where the numpy oracle would take minutes.  On the dyadic grid
    every partial sum is exact in float32, so its result does not depend on its summation order."""
    import ctypes

    import numpy as np

    from __graft_entry__ import build_oracle
    lib = ctypes.CDLL(build_oracle())
    fn = lib.tvo_ternary_matmul_dense_f32
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float, ctypes.c_void_p, ctypes.c_int]
    threads = min(16, os.cpu_count() or 1)   # the GPU box reports the whole machine's cores; one GPU's share is 16

    def run(x, pw, scale, bias, clip, k=None):
        x = np.ascontiguousarray(x, dtype=np.float32)
        pw = np.ascontiguousarray(pw, dtype=np.uint8)
        scale = np.ascontiguousarray(scale, dtype=np.float32)
        bias = None if bias is None else np.ascontiguousarray(bias, dtype=np.float32)
        m, kk = x.shape
        k = kk if k is None else k
        assert k == kk and pw.shape[1] >= (k + 3) // 4 and scale.shape == (pw.shape[0],)
        y = np.empty((m, pw.shape[0]), np.float32)
        rc = fn(x.ctypes.data, m, k, pw.ctypes.data, pw.shape[1], pw.shape[0], scale.ctypes.data,
                bias.ctypes.data if bias is not None else None, float(clip), y.ctypes.data, threads)
        assert rc == 0
        return y

    return run
