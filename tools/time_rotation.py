"""How the 4096^3 GEMM's time depends on where its operands come from: the number of rotating activation / output
buffers (1: L2 / Infinity-Cache resident ... 16: 1 GiB, always from HBM) and of packed-weight buffers.
    python tools/time_rotation.py [M N K]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ternary_vllm_amd as tv  # noqa: E402

M, N, K = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (4096, 4096, 4096)
dev = torch.device("cuda:0")
torch.manual_seed(0)
PAD = int(os.environ.get("TVL_ROW_PAD", "0"))   # extra elements per activation row (breaks the power-of-two row pitch)
xs = [torch.randn(M, K + PAD, device=dev).half()[:, :K] for _ in range(16)]
ys = [torch.empty(M, N, device=dev, dtype=torch.float16) for _ in range(16)]
pws = [torch.randint(0, 171, (N, K // 4), dtype=torch.uint8, device=dev) for _ in range(16)]
scale, bias = torch.rand(N, device=dev) + 0.5, torch.randn(N, device=dev)
for _ in range(300):
    tv.ternary_matmul(xs[0], pws[0], scale, bias, 100.0, out=ys[0])
print(f"activation row pitch {xs[0].stride(0)} elements")
for nx, ny, nw in ((1, 1, 1), (6, 6, 1), (16, 1, 1), (16, 16, 16)):
    per = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(320):
            tv.ternary_matmul(xs[i % nx], pws[i % nw], scale, bias, 100.0, out=ys[i % ny])
        b.record()
        torch.cuda.synchronize()
        per.append(a.elapsed_time(b) / 320 * 1e3)
    print(f"{M}x{N}x{K}: x buffers {nx:2d}  y buffers {ny:2d}  weight buffers {nw:2d}: {min(per):7.2f} us (best of 3 x 320 launches)", flush=True)
