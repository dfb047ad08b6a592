"""A/B several builds of libternary_hip.so on one ternary_matmul shape in ONE process.
    python tools/ab_libs.py M N K libA.so libB.so [libC.so ...]      (TVL_AB_DTYPE=f32|f16|bf16 picks the activation type)
The packed weights and the outputs rotate over buffers totalling > 512 MiB each (more than the Infinity Cache), so every launch
streams them from / to HBM.  Each sample times one sweep over all buffers between two events."""
import ctypes
import os
import sys

import torch

M, N, K = (int(a) for a in sys.argv[1:4])
paths = sys.argv[4:]
libs = [ctypes.CDLL(os.path.abspath(p)) for p in paths]
vp, i64, f32, sz, ci = ctypes.c_void_p, ctypes.c_int64, ctypes.c_float, ctypes.c_size_t, ctypes.c_int
for lib in libs:
    lib.tvl_ternary_matmul.restype = ci
    lib.tvl_ternary_matmul.argtypes = [vp, ci, i64, i64, i64, vp, i64, i64, vp, vp, f32, ci, vp, ci, i64, vp, sz, vp]
dev = torch.device("cuda:0")
torch.manual_seed(0)
nb = max(2, min(64, (640 << 20) // (N * K // 4)))
XD = {"f32": (torch.float32, 0), "f16": (torch.float16, 1), "bf16": (torch.bfloat16, 2)}[os.environ.get("TVL_AB_DTYPE", "f16")]
x = torch.randn(M, K, device=dev).to(XD[0])
pws = [torch.randint(0, 171, (N, K // 4), dtype=torch.uint8, device=dev) for _ in range(nb)]
scale = torch.rand(N, device=dev) + 0.5
bias = torch.randn(N, device=dev)
# outputs rotate too (more than the 256 MiB Infinity Cache in total), so stores cannot be absorbed by a cache; TVL_AB_YBUF=1 pins one buffer
nby = int(os.environ.get("TVL_AB_YBUF", "0")) or max(2, min(64, -(-(640 << 20) // (M * N * 2))))
ys = [[torch.empty((M, N), dtype=torch.float16, device=dev) for _ in range(nby)] for _ in libs]
ws = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
stream = torch.cuda.current_stream().cuda_stream


def launch(li, i):
    rc = libs[li].tvl_ternary_matmul(x.data_ptr(), XD[1], M, K, 0, pws[i % nb].data_ptr(), N, 0, scale.data_ptr(), bias.data_ptr(), 100.0, 0,
                                     ys[li][i % nby].data_ptr(), 1, 0, ws.data_ptr(), ws.numel(), stream)
    assert rc == 0


for li in range(len(libs)):
    for i in range(nb):
        launch(li, i)
torch.cuda.synchronize()
times = [[] for _ in libs]
for r in range(9):
    for li in range(len(libs)):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for i in range(nb):
            launch(li, i)
        e.record()
        torch.cuda.synchronize()
        times[li].append(s.elapsed_time(e) * 1e3 / nb)
byts = M * K * 2 + N * K // 4 + 8 * N + M * N * 2
for li, p in enumerate(paths):
    t = sorted(times[li])
    same = torch.equal(ys[li][(nb - 1) % nby], ys[0][(nb - 1) % nby])
    print(f"{M}x{N}x{K} {p:28s} median {t[len(t)//2]:7.2f} us/call  min {t[0]:7.2f}  -> {byts/t[len(t)//2]/1e3:7.1f} GB/s  same_as_first={same}")
