"""Time quantize+pack (f32 -> 2 bit) for a list of shapes: python tools/time_pack.py N,K [N,K ...]
Batches of launches over rotating buffers (> 256 MiB in total) between one event pair."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ternary_vllm_amd as tv  # noqa: E402

dev = torch.device("cuda:0")
for spec in sys.argv[1:]:
    N, K = (int(a) for a in spec.split(","))
    nb = max(2, (768 << 20) // (N * K * 4))
    ws = [torch.rand(N, K, device=dev) * 2 - 1 for _ in range(nb)]
    pws = [torch.empty((N, (K + 3) // 4), dtype=torch.uint8, device=dev) for _ in range(nb)]
    for i in range(nb):
        tv.ternarize_pack(ws[i], 0.5, out=pws[i])
    torch.cuda.synchronize()
    reps, batch = 7, 2 * nb
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(batch):
            tv.ternarize_pack(ws[i % nb], 0.5, out=pws[i % nb])
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / batch)
    ts.sort()
    byts = N * K * 4 + N * ((K + 3) // 4)
    print(f"{N}x{K}: per launch median {ts[reps//2]*1e3:.1f} us min {ts[0]*1e3:.1f} us -> {byts/ts[reps//2]/1e6:.0f} GB/s ({nb} buffers)")
