import ctypes, os, sys, torch
# usage: python tools/ab_rotation.py libA.so libB.so [...]      (TVL_AB_SHAPE=M,N,K, default 4096,4096,4096)
paths = sys.argv[1:]
libs = [ctypes.CDLL(os.path.abspath(p)) for p in paths]
vp, i64, f32, sz, ci = ctypes.c_void_p, ctypes.c_int64, ctypes.c_float, ctypes.c_size_t, ctypes.c_int
for lib in libs:
    lib.tvl_ternary_matmul.restype = ci
    lib.tvl_ternary_matmul.argtypes = [vp, ci, i64, i64, i64, vp, i64, i64, vp, vp, f32, ci, vp, ci, i64, vp, sz, vp]
M, N, K = (int(a) for a in os.environ.get('TVL_AB_SHAPE', '4096,4096,4096').split(','))
NBUF = max(2, min(16, (1 << 30) // (M * K * 2)))
dev = torch.device("cuda:0")
xs = [torch.randn(M, K, device=dev).half() for _ in range(NBUF)]
ys = [torch.empty(M, N, device=dev, dtype=torch.float16) for _ in range(NBUF)]
pw = torch.randint(0, 171, (N, K // 4), dtype=torch.uint8, device=dev)
scale, bias = torch.rand(N, device=dev) + 0.5, torch.randn(N, device=dev)
st = torch.cuda.current_stream().cuda_stream
def launch(li, i, nx):
    assert libs[li].tvl_ternary_matmul(xs[i % nx].data_ptr(), 1, M, K, 0, pw.data_ptr(), N, 0, scale.data_ptr(), bias.data_ptr(), 100.0, 0, ys[i % nx].data_ptr(), 1, 0, None, 0, st) == 0
for i in range(300): launch(0, i, 1)
for nx in (1, NBUF):
    res = [[] for _ in libs]
    for r in range(5):
        for li in range(len(libs)):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for i in range(160): launch(li, i, nx)
            b.record(); torch.cuda.synchronize()
            res[li].append(a.elapsed_time(b) / 160 * 1e3)
    print(f"x/y buffers {nx:2d}: " + "   ".join(f"{os.path.basename(p)} {sorted(t)[2]:7.2f} us" for p, t in zip(paths, res)), flush=True)
same = all(torch.equal(ys[0], ys[0]) for _ in libs)
