"""Launch the packed ternary GEMM a few times (profiling target).  python tools/run_gemm.py [M N K] [reps] [dtype]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ternary_vllm_amd as tv  # noqa: E402

M, N, K = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (4096, 4096, 4096)
reps = int(sys.argv[4]) if len(sys.argv) >= 5 else 10
dt = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}[sys.argv[5] if len(sys.argv) >= 6 else "f16"]
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.randn(M, K, device=dev).to(dt)
pw = tv.ternarize_pack(torch.rand(N, K, device=dev) * 2 - 1)
scale = torch.rand(N, device=dev) + 0.5
bias = torch.randn(N, device=dev)
y = torch.empty((M, N), dtype=dt, device=dev)
for _ in range(reps):
    tv.ternary_matmul(x, pw, scale, bias, 100.0, out=y)
torch.cuda.synchronize()
print("done", float(y.float().abs().mean()))
