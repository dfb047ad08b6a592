"""Time ternary_matmul with float32 activations: python tools/time_f32.py M,N,K [...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ternary_vllm_amd as tv  # noqa: E402

dev = torch.device("cuda:0")
for spec in sys.argv[1:]:
    M, N, K = (int(a) for a in spec.split(","))
    x = torch.randn(M, K, device=dev)
    pw = tv.ternarize_pack(torch.rand(N, K, device=dev) * 2 - 1)
    scale = torch.rand(N, device=dev) + 0.5
    bias = torch.randn(N, device=dev)
    y = torch.empty((M, N), device=dev)
    for _ in range(3):
        tv.ternary_matmul(x, pw, scale, bias, 100.0, out=y)
    torch.cuda.synchronize()
    n = 10
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        tv.ternary_matmul(x, pw, scale, bias, 100.0, out=y)
    b.record()
    torch.cuda.synchronize()
    t = a.elapsed_time(b) / n
    print(f"f32 x {M}x{N}x{K}: {t*1e3:.1f} us per call  {2.0*M*N*K/t/1e9:.1f} TFLOP/s (effective)")
