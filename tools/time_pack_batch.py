"""Time the batched quantize+pack over a moondream-sized set of layers (22 blocks x 4 matrices, f32 latent weights) by HIP-graph
replay, next to the same bytes as single large matrices.  python tools/time_pack_batch.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ternary_vllm_amd as tv  # noqa: E402

ops = tv.custom_ops_mod
dev = torch.device("cuda:0")


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / reps)
    return best


shapes = [(6144, 2048), (2048, 2048), (8192, 2048), (2048, 8192)] * 22
lat = [torch.rand(n, k, device=dev) * 2 - 1 for n, k in shapes]
dst = [torch.empty((n, k // 4), dtype=torch.uint8, device=dev) for n, k in shapes]
byts = sum(n * k * 4 + n * k // 4 for n, k in shapes)
t = timed(lambda: ops.ternarize_pack_batch(lat, 0.5, "ternary", outs=dst))
print(f"batch of {len(shapes)} layers: {t*1e3:.1f} us, {byts/t/1e6:.0f} GB/s ({byts/t/1e6/80:.1f} % of 8 TB/s)")
t = timed(lambda: [ops.ternarize_pack(w, 0.5, out=o) for w, o in zip(lat, dst)], reps=1)
print(f"{len(shapes)} single calls:      {t*1e3:.1f} us, {byts/t/1e6:.0f} GB/s")
del lat, dst
for n, k in ((16384, 16384), (32768, 8192), (16384, 4096), (8192, 8192)):
    nb = max(2, (1536 << 20) // (n * k * 4))
    ws = [torch.rand(n, k, device=dev) * 2 - 1 for _ in range(nb)]
    os_ = [torch.empty((n, k // 4), dtype=torch.uint8, device=dev) for _ in range(nb)]
    t = timed(lambda: [ops.ternarize_pack(w, 0.5, out=o) for w, o in zip(ws, os_)], reps=1) / nb
    b1 = n * k * 4 + n * k // 4
    print(f"{n}x{k}: {t*1e3:.1f} us, {b1/t/1e6:.0f} GB/s ({b1/t/1e6/80:.1f} %)")
    del ws, os_
