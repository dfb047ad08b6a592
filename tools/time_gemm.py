"""Time the packed ternary GEMM for a list of shapes: python tools/time_gemm.py M,N,K [M,N,K ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ternary_vllm_amd as tv  # noqa: E402

dev = torch.device("cuda:0")
dt = torch.float16
for spec in sys.argv[1:]:
    M, N, K = (int(a) for a in spec.split(","))
    nb = 4
    xs = [torch.randn(M, K, device=dev).to(dt) for _ in range(nb)]
    pw = tv.ternarize_pack(torch.rand(N, K, device=dev) * 2 - 1)
    scale = torch.rand(N, device=dev) + 0.5
    bias = torch.randn(N, device=dev)
    ys = [torch.empty((M, N), dtype=dt, device=dev) for _ in range(nb)]
    for i in range(5):
        tv.ternary_matmul(xs[i % nb], pw, scale, bias, 100.0, out=ys[i % nb])
    torch.cuda.synchronize()
    n = 30
    st = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
    en = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
    for i in range(n):
        st[i].record()
        tv.ternary_matmul(xs[i % nb], pw, scale, bias, 100.0, out=ys[i % nb])
        en[i].record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in zip(st, en))
    med = ts[n // 2]
    print(f"{M}x{N}x{K}: median {med*1e3:.1f} us  min {ts[0]*1e3:.1f} us  {2.0*M*N*K/med/1e9:.1f} TFLOP/s")
