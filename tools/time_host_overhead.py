"""Host cost of one ternary_matmul call from Python (tiny problem, so the GPU is never the limit)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ternary_vllm_amd as tv  # noqa: E402

dev = torch.device("cuda:0")
x = torch.randn(1, 2048, device=dev).half()
pw = tv.ternarize_pack(torch.rand(2048, 2048, device=dev) * 2 - 1)
scale = torch.ones(2048, device=dev); bias = torch.zeros(2048, device=dev)
y = torch.empty((1, 2048), dtype=torch.float16, device=dev)
for label, kw in (("out= given", {"out": y}), ("allocating", {})):
    for _ in range(200):
        tv.ternary_matmul(x, pw, scale, bias, 100.0, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5000
    for _ in range(n):
        tv.ternary_matmul(x, pw, scale, bias, 100.0, **kw)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"{label}: {(t1 - t0) / n * 1e6:.1f} us per call (host issue rate)")

prep = tv.custom_ops_mod.PreparedMatmul(x, pw, scale, bias, 100.0, out=y)
for _ in range(200):
    prep()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5000):
    prep()
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"PreparedMatmul: {(t1 - t0) / 5000 * 1e6:.1f} us per call (host issue rate)")
assert torch.equal(prep(), tv.ternary_matmul(x, pw, scale, bias, 100.0))
