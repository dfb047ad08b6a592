"""Device time of the dense-weight select-and-sum (matrix_mul_tests.py:29-37), both methods: python tools/time_ternary_mm.py"""
import sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ternary_vllm_amd as tv
ops = tv.custom_ops_mod
dev = torch.device("cuda:0")
for M, N, K in ((1024, 1024, 1024), (256, 256, 256), (4096, 4096, 4096)):
    x = torch.randn(M, K, device=dev)
    tm = torch.randn(N, K, device=dev) * (torch.rand(N, K, device=dev) > 0.5)
    for method in ("select", "matmul"):
        for _ in range(3):
            ops.ternary_mm(x, tm, method)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20 if M < 4096 else 3
        a.record()
        for _ in range(reps):
            ops.ternary_mm(x, tm, method)
        b.record()
        torch.cuda.synchronize()
        t = a.elapsed_time(b) / reps
        print(f"ternary_mm {M}x{N}x{K} {method:7s}: {t*1e3:9.1f} us  {2*M*N*K/t/1e9:8.1f} TOPS")
