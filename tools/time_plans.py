"""Time (HIP-graph replay, device time) and check ternary_matmul on a list of shapes under the tile plan given by the
TVL_FORCE_PLAN environment variable ("kind:bm:bn:splits"; unset = the library's own choice).  One process per plan: the
library reads the variable once.  Prints one JSON line per shape.

    TVL_FORCE_PLAN=1:128:128:4 python tools/time_plans.py 256,4096,4096 1024,4096,4096
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.environ.get("TVL_PKG_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # TVL_PKG_ROOT: a copy of the package with another build of the library (A/B runs)
import ternary_vllm_amd as tv  # noqa: E402

ops = tv.custom_ops_mod
dev = torch.device("cuda:0")
plan = os.environ.get("TVL_FORCE_PLAN", "auto")
xdt = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}[os.environ.get("XDT", "f16")]
for spec in sys.argv[1:]:
    M, N, K = (int(a) for a in spec.split(","))
    nb = max(2, min(16, int(600e6 // max(1, N * K // 4 + M * K * 2))))
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    pws = [torch.randint(0, 171, (N, K // 4), dtype=torch.uint8, device=dev) for _ in range(nb)]
    xs = [(torch.randint(-64, 65, (M, K), generator=g).float() / 64).to(dev).to(xdt) for _ in range(2)] + \
         [torch.randn(M, K, device=dev).to(xdt) for _ in range(nb - 2)]
    ys = [torch.empty((M, N), dtype=xdt, device=dev) for _ in range(nb)]
    scale = (torch.randint(4, 13, (N,), generator=g) / 8.0).float().to(dev)
    bias = (torch.randint(-16, 17, (N,), generator=g) / 8.0).float().to(dev)
    # exact-grid check against a float32 matmul of the unpacked weights (any correct summation order gives these bits)
    got = ops.ternary_matmul(xs[0], pws[0], scale, bias, 100.0, out_dtype=torch.float32)
    ref = torch.clip((xs[0].float() @ ops.unpack_ternary(pws[0], K).t()) * scale + bias, -100.0, 100.0)
    ok = bool(torch.equal(got, ref))
    del got, ref
    call = lambda i: ops.ternary_matmul(xs[i % nb], pws[i % nb], scale, bias, 100.0, out=ys[i % nb])  # noqa: E731
    for i in range(3 * nb):
        call(i)
    torch.cuda.synchronize()
    reps = 4 * nb
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for i in range(reps):
            call(i)
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    best = 1e9
    tot = 0.0
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); graph.replay(); b.record()
        torch.cuda.synchronize()
        t = a.elapsed_time(b) / reps
        best = min(best, t); tot += t
    us = tot / 5 * 1e3
    print(json.dumps({"plan": plan, "m": M, "n": N, "k": K, "us": round(us, 2), "best_us": round(best * 1e3, 2),
                      "TFLOPs": round(2.0 * M * N * K / (us * 1e-6) / 1e12, 1), "exact": ok}), flush=True)
    del pws, xs, ys
