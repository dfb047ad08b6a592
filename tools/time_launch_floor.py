import torch, sys, os
sys.path.insert(0, os.getcwd())
import ternary_vllm_amd as tv
ops = tv.custom_ops_mod
dev = torch.device("cuda:0")
a = torch.zeros(64, dtype=torch.int32, device=dev)
def bench(fn, reps=50):
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(reps): fn()
        g.replay(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(side)
        for _ in range(10): g.replay()
        e.record(side); torch.cuda.synchronize()
    return s.elapsed_time(e) / (10 * reps) * 1e3
print("tiny kernel (zero_out of 64 ints) per node: %.2f us" % bench(lambda: ops.zero_out(a)))
for (m, n, k) in [(1, 2048, 2048), (1, 4096, 4096), (1, 6144, 2048), (1, 256, 256), (1, 64, 64)]:
    x = torch.randn(m, k, device=dev).half(); pw = torch.randint(0, 171, (n, k // 4), dtype=torch.uint8, device=dev)
    sc = torch.ones(n, device=dev); y = torch.empty(m, n, device=dev, dtype=torch.float16)
    print((m, n, k), "%.2f us" % bench(lambda: ops.ternary_matmul(x, pw, sc, None, 100.0, out=y)))
