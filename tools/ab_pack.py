"""A/B builds of the library on quantize+pack (f32 -> 2 bit): python tools/ab_pack.py N K libA.so libB.so ..."""
import ctypes
import os
import sys

import torch

N, K = int(sys.argv[1]), int(sys.argv[2])
paths = sys.argv[3:]
libs = [ctypes.CDLL(os.path.abspath(p)) for p in paths]
vp, i64, f32, ci = ctypes.c_void_p, ctypes.c_int64, ctypes.c_float, ctypes.c_int
for lib in libs:
    lib.tvl_quantize_pack.restype = ci
    lib.tvl_quantize_pack.argtypes = [vp, ci, i64, i64, i64, ci, f32, vp, i64, vp]
dev = torch.device("cuda:0")
nb = max(2, (768 << 20) // (N * K * 4))
ws = [torch.rand(N, K, device=dev) * 2 - 1 for _ in range(nb)]
outs = [torch.empty((N, K // 4), dtype=torch.uint8, device=dev) for _ in libs]
stream = torch.cuda.current_stream().cuda_stream
times = [[] for _ in libs]
for r in range(9):
    for li, lib in enumerate(libs):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(nb):
            assert lib.tvl_quantize_pack(ws[i].data_ptr(), 0, N, K, 0, 0, 0.5, outs[li].data_ptr(), 0, stream) == 0
        b.record(); torch.cuda.synchronize()
        times[li].append(a.elapsed_time(b) / nb * 1e3)
byts = N * K * 4 + N * K // 4
for li, p in enumerate(paths):
    t = sorted(times[li])
    print(f"{N}x{K} {p:24s} median {t[4]:7.2f} us  -> {byts / t[4] / 1e3:7.1f} GB/s  same={torch.equal(outs[li], outs[0])}")
