"""Profiling target: run ternary_matmul `reps` times for each M,N,K given (in order).  Used under
rocprofv3 --kernel-trace; tools/trace_shapes.py turns the trace into per-shape device times."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ternary_vllm_amd as tv  # noqa: E402

reps = int(os.environ.get("REPS", "20"))
dev = torch.device("cuda:0")
for spec in sys.argv[1:]:
    M, N, K = (int(a) for a in spec.split(","))
    nb = max(2, min(8, int(400e6 // max(1, N * K // 4))))  # rotate weights so that they come from HBM, not cache
    pad = int(os.environ.get("ROWPAD", "0"))  # extra bytes per packed row (experiment: non power-of-two row stride)
    pws = [torch.randint(0, 171, (N, K // 4 + pad), dtype=torch.uint8, device=dev)[:, :K // 4] for _ in range(nb)]
    x = torch.randn(M, K, device=dev).half()
    scale = torch.ones(N, device=dev)
    y = torch.empty((M, N), dtype=torch.float16, device=dev)
    torch.cuda.synchronize()
    mark = torch.zeros(1, dtype=torch.int32, device=dev)
    for i in range(reps):
        mark.fill_(i)   # one int32 fill kernel in front of every call: tools/trace_shapes.py splits the trace into calls there
        tv.ternary_matmul(x, pws[i % nb], scale, None, 100.0, out=y)
    torch.cuda.synchronize()
    del pws
print("done")
