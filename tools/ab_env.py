"""A/B the SAME library under two environments in one process is impossible (the switch is read once), so this runs
the 4096^3 GEMM timing in two child processes alternately, several rounds each, and prints medians.
    python tools/ab_env.py VAR[=VALUE] [M N K]      (VAR unset vs VAR=VALUE, default 1)"""
import os
import subprocess
import sys

var, _, val = sys.argv[1].partition("=")
val = val or "1"
shape = sys.argv[2:5] if len(sys.argv) >= 5 else ["4096", "4096", "4096"]
child = r'''
import os, sys, torch
sys.path.insert(0, os.getcwd())
import ternary_vllm_amd as tv
M, N, K = (int(a) for a in sys.argv[1:4])
dev = torch.device("cuda:0")
xs = [torch.randn(M, K, device=dev).half() for _ in range(4)]
pw = tv.ternarize_pack(torch.rand(N, K, device=dev) * 2 - 1)
scale = torch.rand(N, device=dev) + 0.5; bias = torch.randn(N, device=dev)
ys = [torch.empty((M, N), dtype=torch.float16, device=dev) for _ in range(4)]
for i in range(300): tv.ternary_matmul(xs[i % 4], pw, scale, bias, 100.0, out=ys[i % 4])
torch.cuda.synchronize()
best = []
for r in range(5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(200): tv.ternary_matmul(xs[i % 4], pw, scale, bias, 100.0, out=ys[i % 4])
    b.record(); torch.cuda.synchronize()
    best.append(a.elapsed_time(b) / 200 * 1e3)
best.sort()
print(f"{best[2]:.2f} {best[0]:.2f} {float(ys[0].float().abs().sum()):.6e}")
'''
res = {False: [], True: []}
for rnd in range(3):
    for on in (False, True):
        env = dict(os.environ)
        env.pop(var, None)
        if on:
            env[var] = val
        out = subprocess.run([sys.executable, "-c", child, *shape], env=env, capture_output=True, text=True, timeout=300)
        if out.returncode != 0:
            print(out.stderr[-2000:]); sys.exit(1)
        res[on].append(out.stdout.strip().split())
for on in (False, True):
    meds = sorted(float(r[0]) for r in res[on])
    flops = 2.0 * int(shape[0]) * int(shape[1]) * int(shape[2])
    print(f"{var}={val if on else 'unset':5s}: median of process medians {meds[1]:7.2f} us  best {min(float(r[1]) for r in res[on]):7.2f} us -> "
          f"{flops / meds[1] / 1e6:7.1f} TFLOP/s   checksum {res[on][0][2]}")
