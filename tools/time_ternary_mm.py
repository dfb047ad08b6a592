"""Device time (HIP-graph replay) of the dense-weight select-and-sum (matrix_mul_tests.py:29-37), both methods, over sizes:
    python tools/time_ternary_mm.py [size ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ternary_vllm_amd as tv  # noqa: E402

ops = tv.custom_ops_mod
dev = torch.device("cuda:0")
sizes = [int(a) for a in sys.argv[1:]] or [64, 128, 256, 512, 1024, 4096]
for S in sizes:
    x = torch.randn(S, S, device=dev)
    tm = torch.randn(S, S, device=dev) * (torch.rand(S, S, device=dev) > 0.5)
    for method in ("select", "matmul"):
        for _ in range(3):
            ops.ternary_mm(x, tm, method)
        torch.cuda.synchronize()
        reps = 20 if S < 4096 else 3
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            ops.ternary_mm(x, tm, method)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                for _ in range(reps):
                    ops.ternary_mm(x, tm, method)
            g.replay()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(side)
            for _ in range(5):
                g.replay()
            b.record(side)
            torch.cuda.synchronize()
        t = a.elapsed_time(b) / (5 * reps)
        print(f"ternary_mm {S}^3 {method:7s}: {t*1e3:9.2f} us  {2*S**3/t/1e9:9.2f} TOPS", flush=True)
