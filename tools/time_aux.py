"""Device time of the one-time / helper kernels at N x K (default 4096 x 4096), cache-cold rotating buffers:
    python tools/time_aux.py [N,K]
Prints microseconds per call and algorithmic GB/s (bytes each op must read + write once)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ternary_vllm_amd as tv  # noqa: E402

ops = tv.custom_ops_mod
dev = torch.device("cuda:0")
N, K = (int(a) for a in (sys.argv[1] if len(sys.argv) > 1 else "4096,4096").split(","))
nb = max(2, (640 << 20) // (N * K * 4))
ws = [torch.randn(N, K, device=dev) * 0.05 for _ in range(nb)]
ts8 = [torch.randint(-1, 2, (N, K), dtype=torch.int8, device=dev) for _ in range(nb)]
pws = [ops.pack_fn(t) for t in ts8]


def timeit(name, fn, byts, reps=5):
    for i in range(nb):
        fn(i)
    torch.cuda.synchronize()
    try:   # device time: replay the calls from a graph (the Python wrappers cost more than the short kernels)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(nb):
                fn(i)
        g.replay()
        torch.cuda.synchronize()
        run = g.replay
    except Exception as exc:   # noqa: BLE001
        print(f"  ({name}: graph capture failed, timing eager calls: {exc})")
        run = lambda: [fn(i) for i in range(nb)]  # noqa: E731
    best = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        run()
        b.record()
        torch.cuda.synchronize()
        best.append(a.elapsed_time(b) / nb)
    best.sort()
    t = best[len(best) // 2]
    print(f"{name:44s} {t*1e3:9.1f} us  {byts/t/1e6:8.0f} GB/s", flush=True)


f32, u2 = N * K * 4, N * ((K + 3) // 4)
timeit("fit_ternary_scale independent (+latent)", lambda i: ops.fit_ternary_scale(ws[i], "independent"), 2 * f32)
timeit("fit_ternary_scale independent (scale only)", lambda i: ops.fit_ternary_scale(ws[i], "independent", return_latent=False), f32)
timeit("fit_ternary_scale uniform (+latent)", lambda i: ops.fit_ternary_scale(ws[i], "uniform"), 2 * f32)
timeit("fit_binary_scale uniform (+latent)", lambda i: ops.fit_binary_scale(ws[i], "binary", "uniform"), 2 * f32)
timeit("fit_binary_scale independent (+latent)", lambda i: ops.fit_binary_scale(ws[i], "binary", "independent"), 2 * f32)
timeit("quantize (f32 -> f32 ternary)", lambda i: ops.quantize(ws[i], "ternary"), 2 * f32)
timeit("pack_fn (int8 -> 2 bit)", lambda i: ops.pack_fn(ts8[i]), N * K + u2)
timeit("unpack_ternary -> f32", lambda i: ops.unpack_ternary(pws[i], K, torch.float32), u2 + f32)
timeit("unpack_ternary -> f16", lambda i: ops.unpack_ternary(pws[i], K, torch.float16), u2 + f32 // 2)
flat = [t.reshape(-1) for t in ts8]
timeit("pack_ternary_planar b (int8)", lambda i: ops.pack_ternary_planar(flat[i], "b"), N * K + u2)
pl = [ops.pack_ternary_planar(f, "b") for f in flat]
timeit("unpack_ternary_planar b -> f32", lambda i: ops.unpack_ternary_planar(pl[i], "b"), u2 + f32)
