"""Parity of ternary_matmul under a pinned tile plan (TVL_FORCE_PLAN=kind:bm:bn:splits, read once per process) against the CPU
oracle on exact-grid inputs: ragged shapes, every activation / output dtype, bias on / off, the fused activation from a channel
on, over-wide packed rows with code 3.  Prints one line per case and exits non-zero on the first mismatch.

    TVL_FORCE_PLAN=1:128:256:2 python tools/check_plan.py [M,N,K ...]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ternary_vllm_amd as tv  # noqa: E402
from oracle import ternary_oracle as O  # noqa: E402  (checker only)

ops = tv.custom_ops_mod
dev = torch.device("cuda:0")
shapes = [tuple(int(a) for a in s.split(",")) for s in sys.argv[1:]] or [(300, 500, 512), (129, 257, 64), (1000, 130, 1280), (256, 384, 192)]
rng = np.random.default_rng(99)
EXTRA = [int(a) for a in os.environ.get("CHECK_EXTRA", "0,8,16").split(",")]   # bytes appended to every packed row, per case % 3
bad = 0
for (m, n, k) in shapes:
    for case in range(6):
        xdt = [torch.float16, torch.bfloat16, torch.float32][case % 3]
        ydt = [torch.float32, xdt][case % 2]
        has_bias, gelu = case % 2 == 0, case == 4
        act_begin = n // 3 if gelu else 0
        extra = EXTRA[case % 3]
        x = rng.integers(-64, 65, size=(m, k)).astype(np.float32) / 64.0
        pw = rng.integers(0, 256, size=(n, k // 4 + extra)).astype(np.uint8)
        scale = (rng.integers(1, 9, size=(n,)) / 8.0).astype(np.float32) * (0.03125 if gelu else 1.0)
        bias = (rng.integers(-16, 17, size=(n,)) / 8.0).astype(np.float32) if has_bias else None
        want = O.ternary_matmul(x, pw, scale, bias, 1e6, k=k, act=O.ACT_GELU_TANH if gelu else O.ACT_NONE, act_begin=act_begin)
        got = ops.ternary_matmul(torch.from_numpy(x).to(dev).to(xdt), torch.from_numpy(pw).to(dev), torch.from_numpy(scale).to(dev),
                                 torch.from_numpy(bias).to(dev) if has_bias else None, 1e6, activation="gelu_tanh" if gelu else None,
                                 out_dtype=ydt, act_begin=act_begin)
        g = got.float().cpu().numpy()
        w = torch.from_numpy(want).to(ydt).float().numpy()
        ok = (np.max(np.abs(g - w)) <= (1e-5 if ydt == torch.float32 else 2e-2)) if gelu else np.array_equal(g, w)
        print(f"plan {os.environ.get('TVL_FORCE_PLAN', 'auto')}: {m}x{n}x{k} x={xdt} y={ydt} bias={has_bias} gelu={gelu}: {'ok' if ok else 'MISMATCH'}", flush=True)
        bad += not ok
sys.exit(1 if bad else 0)
