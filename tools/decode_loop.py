"""Decode-loop timing with synthetic ternary weights (prefill, then single-token steps):
    python tools/decode_loop.py [--batch 1] [--prompt 745] [--steps 64] [--blocks 24] [--no-graph] [--no-fuse]
Prints one JSON line per configuration."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ternary_vllm_amd as tv  # noqa: E402,F401
from ternary_vllm_amd import decode_harness as dh  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, nargs="+", default=[1])
ap.add_argument("--prompt", type=int, default=745)   # 729 image tokens + a short text prompt (the reference's case)
ap.add_argument("--steps", type=int, default=64)
ap.add_argument("--blocks", type=int, default=24)
ap.add_argument("--no-graph", action="store_true")
ap.add_argument("--no-fuse", action="store_true", help="Wqkv and fc1 as two matmuls instead of one")
args = ap.parse_args()
for b in args.batch:
    cfg = dh.DecoderConfig(blocks=args.blocks, max_len=((args.prompt + args.steps + 2 + 63) // 64) * 64, fuse_qkv_fc1=not args.no_fuse)
    r = dh.run(cfg, b, args.prompt, args.steps, use_graph=not args.no_graph)
    r["fused_qkv_fc1"] = not args.no_fuse
    print(json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items()}), flush=True)
    torch.cuda.empty_cache()
