"""Binarized-activation matmul: AND + popcount kernels against the matrix-core path (same integers).
    python tools/time_popcount.py M,N,K [...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ternary_vllm_amd as tv  # noqa: E402

ops = tv.custom_ops_mod
dev = torch.device("cuda:0")


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for spec in sys.argv[1:]:
    M, N, K = (int(a) for a in spec.split(","))
    x = torch.randn(M, K, device=dev).half()
    pw = tv.ternarize_pack(torch.rand(N, K, device=dev) * 2 - 1)
    scale = torch.rand(N, device=dev) + 0.5
    bias = torch.randn(N, device=dev)
    pos, neg = ops.ternary_bitplanes(pw, K)
    xb = ops.sign_pack_bits(x)
    s = torch.where(x >= 0, 1.0, -1.0).half()
    t_pop = timed(lambda: ops.popcount_matmul(xb, pos, neg, K, scale, bias, 100.0))
    t_bits = timed(lambda: ops.sign_pack_bits(x))
    t_mfma = timed(lambda: ops.ternary_matmul(s, pw, scale, bias, 100.0, out_dtype=torch.float32))
    same = torch.equal(ops.popcount_matmul(xb, pos, neg, K, scale, bias, 100.0), ops.ternary_matmul(s, pw, scale, bias, 100.0, out_dtype=torch.float32))
    print(f"{M}x{N}x{K}: popcount {t_pop:8.1f} us (+ sign bits {t_bits:6.1f} us)   matrix cores {t_mfma:8.1f} us   "
          f"{2.0*M*N*K/t_pop/1e6:8.1f} vs {2.0*M*N*K/t_mfma/1e6:8.1f} TOPS   identical={same}")

# HBM-cold variant for few rows: rotate the bit planes through more than the Infinity Cache holds
for spec in sys.argv[1:]:
    M, N, K = (int(a) for a in spec.split(","))
    if M > 8:
        continue
    nb = max(2, min(40, (640 << 20) // (N * K // 4)))
    planes = [(torch.randint(-2**31, 2**31 - 1, (N, K // 32), dtype=torch.int32, device=dev), torch.randint(-2**31, 2**31 - 1, (N, K // 32), dtype=torch.int32, device=dev)) for _ in range(nb)]
    xb = ops.sign_pack_bits(torch.randn(M, K, device=dev))
    scale = torch.ones(N, device=dev)
    state = {"i": 0}

    def call():
        p, q = planes[state["i"] % nb]
        state["i"] += 1
        return ops.popcount_matmul(xb, p, q, K, scale, None, 100.0)
    t = timed(call, n=2 * nb)
    byts = N * K // 4 + M * K // 8 + 4 * N + 4 * M * N
    print(f"{M}x{N}x{K} popcount, planes from HBM ({nb} rotating sets): {t:7.1f} us  {byts/t/1e3:7.1f} GB/s")
