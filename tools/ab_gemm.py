"""A/B two builds of libternary_hip.so on the same GEMM in ONE process (interleaved rounds, same device).
    python tools/ab_gemm.py libA.so libB.so [M N K] [rounds]
Prints median / min device time per build (HIP events around each launch)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from importlib import import_module  # noqa: E402

libs = [ctypes.CDLL(os.path.abspath(p)) for p in sys.argv[1:3]]
M, N, K = (int(a) for a in sys.argv[3:6]) if len(sys.argv) >= 6 else (4096, 4096, 4096)
rounds = int(sys.argv[6]) if len(sys.argv) >= 7 else 30
vp, i64, f32, sz, ci = ctypes.c_void_p, ctypes.c_int64, ctypes.c_float, ctypes.c_size_t, ctypes.c_int
for lib in libs:
    lib.tvl_ternary_matmul.restype = ci
    lib.tvl_ternary_matmul.argtypes = [vp, ci, i64, i64, i64, vp, i64, i64, vp, vp, f32, ci, vp, ci, i64, vp, sz, vp]
dev = torch.device("cuda:0")
torch.manual_seed(0)
nb = 4
xs = [torch.randn(M, K, device=dev).half() for _ in range(nb)]
pw = torch.randint(0, 171, (N, K // 4), dtype=torch.uint8, device=dev)
scale = torch.rand(N, device=dev) + 0.5
bias = torch.randn(N, device=dev)
ys = [torch.empty((M, N), dtype=torch.float16, device=dev) for _ in range(nb)]
ws = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
stream = torch.cuda.current_stream().cuda_stream


def launch(lib, i):
    rc = lib.tvl_ternary_matmul(xs[i % nb].data_ptr(), 1, M, K, 0, pw.data_ptr(), N, 0, scale.data_ptr(), bias.data_ptr(), 100.0, 0,
                                ys[i % nb].data_ptr(), 1, 0, ws.data_ptr(), ws.numel(), stream)
    assert rc == 0


for lib in libs:
    for i in range(5):
        launch(lib, i)
torch.cuda.synchronize()
times = [[] for _ in libs]
for r in range(rounds):
    for li, lib in enumerate(libs):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        launch(lib, r)
        e.record()
        torch.cuda.synchronize()
        times[li].append(s.elapsed_time(e) * 1e3)
outs = []
for li, lib in enumerate(libs):
    launch(lib, 0)
    torch.cuda.synchronize()
    outs.append(ys[0].clone())
for li, t in enumerate(times):
    t.sort()
    print(f"{sys.argv[1 + li]:40s} median {t[len(t) // 2]:8.1f} us  min {t[0]:8.1f} us  -> {2.0 * M * N * K / t[len(t) // 2] / 1e6:8.1f} TFLOP/s")
print("outputs identical:", bool(torch.equal(outs[0], outs[1])))
