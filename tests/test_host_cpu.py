"""CPU-only checks: the C-ABI library loads and exports every symbol of include/tvl.h, argument errors
are reported without a GPU, the host-side helpers, and the row-sharded multi-rank path over gloo."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import ternary_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import ternary_vllm_amd as tv
    header = open(os.path.join(ROOT, "include", "tvl.h")).read()
    declared = set(re.findall(r"\b(tvl_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations found in include/tvl.h"
    lib = ctypes.CDLL(tv.custom_ops_mod.LIBRARY_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in tvl.h but not exported"
    assert declared == set(tv.custom_ops_mod.SIGNATURES), "python binding table and header disagree"
    assert tv.custom_ops_mod.library_version() == 100
    out = subprocess.check_output(["nm", "-D", "--defined-only", tv.custom_ops_mod.LIBRARY_PATH], text=True)
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    assert exported == declared, f"unexpected exported symbols: {exported ^ declared}"


def test_argument_errors_without_gpu():
    """Validation happens before any launch: bad arguments give TVL_ERR_INVALID and a message."""
    import ternary_vllm_amd as tv
    lib = tv.custom_ops_mod._lib
    rc = lib.tvl_ternary_matmul(None, 9, 4, 64, 0, None, 8, 0, None, None, 1.0, 0, None, 0, 0, None, 0, None)
    assert rc == 1 and b"dtype" in lib.tvl_last_error()
    rc = lib.tvl_ternary_matmul(None, 1, 4, 64, 0, None, 8, 15, None, None, 1.0, 0, None, 1, 0, None, 0, None)
    assert rc == 1 and b"packed rows hold 15 bytes" in lib.tvl_last_error()
    rc = lib.tvl_ternary_matmul(None, 1, 4, 64, 0, None, 8, 0, None, None, 1.0, 0, None, 1, 0, None, 0, None)
    assert rc == 1 and b"NULL" in lib.tvl_last_error()
    assert lib.tvl_ternary_matmul(None, 1, 0, 64, 0, None, 8, 0, None, None, 1.0, 0, None, 1, 0, None, 0, None) == 0  # m = 0
    assert lib.tvl_unpack(None, 4, 65, 16, None, 0, 0, None) == 1
    assert lib.tvl_quantize_pack(None, 0, 4, 8, 0, 7, 0.5, None, 0, None) == 1 and b"quantizer" in lib.tvl_last_error()
    assert lib.tvl_pack_planar_i8(None, 8, 5, None, None) == 1
    with pytest.raises(ValueError):
        tv.ternary_matmul(torch.zeros(2, 8), torch.zeros(2, 2, dtype=torch.uint8), torch.ones(2), None, 1.0)  # CPU tensors
    with pytest.raises(ValueError):
        tv.pack_fn(torch.zeros(2, 8, dtype=torch.int8))


def test_row_bytes_and_sharding_helpers():
    import ternary_vllm_amd as tv
    ops = tv.custom_ops_mod
    for k in range(1, 40):
        assert ops.packed_row_bytes(k) == O.packed_row_bytes(k)
        assert ops.reference_row_bytes(k) == O.reference_row_bytes(k) >= ops.packed_row_bytes(k)
    for m in (0, 1, 7, 2048, 2049):
        for world in (1, 2, 3, 8):
            spans = [tv.shard_rows(m, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == m
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1
    assert tv.utils.layers_to_skip(24, 1, 1) == ["transformer.h.0.", "transformer.h.23."]  # convert.py:72


def test_checkpoint_layer_selection_follows_the_parent_module_rule():
    """The converter's layer choice against the reference's walk (binary/utils.py:17-21 tests the name of the module whose direct
    children binary/modules.py:191 converts), re-enacted on a module tree with torch alone: Phi-shaped blocks plus a Linear that is a
    DIRECT child of a skipped block and a head."""
    import ternary_vllm_amd as tv
    from torch import nn

    class Block(nn.Module):
        def __init__(self):
            super().__init__()
            self.mixer = nn.ModuleDict({"Wqkv": nn.Linear(4, 4), "out_proj": nn.Linear(4, 4)})
            self.mlp = nn.ModuleDict({"fc1": nn.Linear(4, 4), "fc2": nn.Linear(4, 4)})
            self.gate = nn.Linear(4, 4)   # direct child of the block

    model = nn.Module()
    model.transformer = nn.Module()
    model.transformer.h = nn.ModuleList([Block() for _ in range(4)])
    model.lm_head = nn.ModuleDict({"linear": nn.Linear(4, 4)})
    model.proj = nn.Linear(4, 4)       # direct child of the root
    for start_skip, last_skip in [(1, 0), (1, 1), (0, 0), (2, 1)]:
        skip = [f"transformer.h.{i}." for i in range(start_skip)] + [f"transformer.h.{4 - 1 - i}." for i in range(last_skip)]
        want = []                       # what the reference's loop converts
        for name, mod in model.named_modules():
            if "lm_head" not in name and not any(s in name for s in skip):
                want += [(name + "." if name else "") + cn for cn, child in mod.named_children() if isinstance(child, nn.Linear)]
        keys = []
        for name, mod in model.named_modules():
            if isinstance(mod, nn.Linear):
                keys += [name + ".weights", name + ".scale", name + ".bias"]
        keys += ["transformer.embd.wte.weight", "orphan.weights"]          # not quantized layers: no .scale beside them
        chosen, skipped = tv.conversion.checkpoint_layer_selection(keys, start_skip, last_skip, n_blocks=4)
        assert sorted(chosen) == sorted(want), (start_skip, last_skip)
        assert sorted(chosen + skipped) == sorted(k[:-8] for k in keys if k.endswith(".weights") and k != "orphan.weights")
        if start_skip:
            assert "transformer.h.0.gate" in chosen and "transformer.h.0.mlp.fc1" in skipped   # the reference's quirk, kept
        assert "lm_head.linear" in skipped and "proj" in chosen
    # block count inferred from the keys when not given
    assert tv.conversion.checkpoint_layer_selection(keys, 0, 1)[1] == tv.conversion.checkpoint_layer_selection(keys, 0, 1, n_blocks=4)[1]


def test_c_port_matches_numpy_oracle():
    """The C/OpenMP restatement (CPU baseline of the benchmark) against the numpy oracle."""
    from __graft_entry__ import build_oracle
    lib = ctypes.CDLL(build_oracle())
    rng = np.random.default_rng(5)
    m, n, k = 37, 50, 203
    x = (rng.integers(-64, 65, (m, k)) / 64.0).astype(np.float32)
    w = rng.uniform(-1, 1, (n, k)).astype(np.float32)
    pw = np.zeros((n, (k + 3) // 4), np.uint8)
    lib.tvo_ternarize_pack_f32(w.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(n), ctypes.c_int64(k), ctypes.c_float(0.5),
                               pw.ctypes.data_as(ctypes.c_void_p))
    assert np.array_equal(pw, O.pack_format_c(O.ternarize(w)))
    scale = rng.uniform(0.5, 1.5, n).astype(np.float32)
    bias = rng.normal(size=n).astype(np.float32)
    want = O.ternary_matmul(x, pw, scale, bias, 3.0)
    for fn in (lib.tvo_ternary_matmul_dense_f32, lib.tvo_ternary_matmul_select_f32):
        fn.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                       ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float, ctypes.c_void_p, ctypes.c_int]
        y = np.empty((m, n), np.float32)
        assert fn(x.ctypes.data, m, k, pw.ctypes.data, pw.shape[1], n, scale.ctypes.data, bias.ctypes.data, 3.0, y.ctypes.data, 2) == 0
        assert np.array_equal(y, want)


_WORKER = r"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["TVL_ROOT"])
import ternary_vllm_amd as tv
from ternary_vllm_amd.conversion import ArenaEntry, PackedArena
from oracle import ternary_oracle as O

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
rng = np.random.default_rng(7)                      # same on every rank: the full problem
m, shapes = 11, {"fc1": (24, 16), "fc2": (16, 24)}
x = (rng.integers(-64, 65, (m, 16)) / 64.0).astype(np.float32)
entries, off, voff = [], 0, 0
for name, (n, k) in shapes.items():
    ld = 16
    entries.append(ArenaEntry(name, n, k, ld, off, voff, 100.0, True, "ternary")); off += n * ld; voff += n
arena = None
if rank == 0:                                       # rank 0 owns the packed weights
    arena = PackedArena.empty_like_index(entries, "cpu")
    arena.packed.fill_(0x55)
    for e in entries:
        w = rng.uniform(-1, 1, (e.n, e.k)).astype(np.float32)
        pw = O.pack_format_c(O.ternarize(w))
        arena.packed[e.offset:e.offset + e.n * e.ld].view(e.n, e.ld)[:, :pw.shape[1]] = torch.from_numpy(pw)
        arena.scales[e.vec_offset:e.vec_offset + e.n] = torch.from_numpy(rng.uniform(0.5, 1.5, e.n).astype(np.float32))
        arena.biases[e.vec_offset:e.vec_offset + e.n] = torch.from_numpy(rng.normal(size=e.n).astype(np.float32))
arena = PackedArena.replicate(arena, "cpu", src=0)  # index + one replication for all layers: what bench.py does on N GPUs
assert list(arena.entries) == ["fc1", "fc2"]
b, e_ = tv.shard_rows(m, world, rank)               # rows of x are independent: shard them, no collective
def run(xx):
    pw, s, bi, mv = arena.layer("fc1")
    h = O.ternary_matmul(xx, pw.numpy(), s.numpy(), bi.numpy(), mv, k=16)
    pw, s, bi, mv = arena.layer("fc2")
    return O.ternary_matmul(h, pw.numpy(), s.numpy(), bi.numpy(), mv, k=24)
mine = torch.from_numpy(run(x[b:e_]))
sizes = [tv.shard_rows(m, world, r) for r in range(world)]
parts = [torch.empty((hi - lo, 16), dtype=torch.float32) for lo, hi in sizes]
dist.all_gather(parts, mine) if len({p.shape for p in parts}) == 1 else None
gathered = [None] * world
dist.all_gather_object(gathered, mine.numpy())
got = np.concatenate(gathered, axis=0)
want_all = [None]
if rank == 0:
    want_all = [run(x)]
dist.broadcast_object_list(want_all, src=0)
assert got.shape == (m, 16) and np.array_equal(got, want_all[0]), "sharded result differs from the unsharded one"
dist.barrier()
if rank == 0:
    print("SHARD_OK")
dist.destroy_process_group()
"""


def test_row_sharded_two_ranks_gloo(tmp_path):
    """world_size 2 over gloo on the CPU: arena broadcast from rank 0, rows of x sharded, outputs reassembled
    equal to the single-rank result.  The oracle stands in for the GPU kernels (compute is not under test)."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, TVL_ROOT=ROOT, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29611", str(script)], env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "SHARD_OK" in out.stdout


def _bench(*argv, env=None, timeout=300):
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_gpus_flag_launches_that_many_ranks():
    """`bench.py --gpus 2` started as a plain process spawns two ranks (torch.distributed.run as a child), which replicate
    the packed arena from rank 0, take their row shards and report through rank 0: one JSON line with n_gpus == 2.
    --dry-run: host tensors over gloo, no kernels (the same launch / replicate / shard / barrier + MAX-timing code)."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    out = _bench("--gpus", "2", "--dry-run", "--steps", "4", "--warmup", "1", "--m", "48", "--n", "40", "--k", "100", env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["ranks_reported"] == 2 and r["steps"] == 4 and r["dry_run"] is True
    assert r["row_spans"] == [[0, 48], [48, 96]] and r["replicated_arena_identical"] is True
    assert r["weight_broadcast_bytes"] == 40 * 32 + 8 * 40          # rows of 25 bytes padded to 32, scales and biases
    # a rank count that disagrees with --gpus is refused, not silently reported as something else
    bad = _bench("--gpus", "4", "--dry-run", env=dict(env, RANK="0", WORLD_SIZE="2", LOCAL_RANK="0"))
    assert bad.returncode == 2 and "WORLD_SIZE=2" in bad.stderr
    # a failing rank fails the launcher
    assert _bench("--gpus", "2", "--dry-run", "--steps", "x", env=env).returncode != 0


def test_inline_asm_load_audit_is_part_of_the_build():
    """The matrix-core kernels issue loads from inline asm and complete them with hand-counted waits, which the compiler
    cannot see: `make` audits the device assembly of the shipped objects and refuses to link on a finding.  Here: the
    audit target is up to date with the sources (or rebuilds), and every audited kernel reports zero suspicious copies."""
    csrc = os.path.join(ROOT, "ternary-vllm_amd", "csrc")
    out = subprocess.run(["make", "-C", csrc, "-j8", "audit"], capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    assert "asm audit clean" in out.stdout
    rules = open(os.path.join(csrc, "Makefile")).read()
    assert re.search(r"^\$\(OUT\):.*\$\(AUDITS\)", rules, re.M), "the library must depend on the audits"
    for name, kernel in (("audit_mfma.txt", "gemm_mfma_kernel"), ("audit_mfma16.txt", "gemm_mfma16_kernel"),
                         ("audit_skinny.txt", "gemm_skinny_kernel"), ("audit_stream.txt", "gemm_stream_kernel")):
        text = open(os.path.join(csrc, "build", name)).read()
        assert "TOTAL suspicious copies: 0" in text and kernel in text, name
        assert len(re.findall(r"load destinations \d+ regs", text)) >= 4, f"{name}: too few kernel instances audited"


def test_planner_model_in_python_has_the_librarys_cost_table():
    """tools/planner_model.py restates the tile planner's cost model to score rule changes against measured plan timings
    (profiles/r02_planner_audit.jsonl); its table must be the library's (gemm_mfma.hip: TILE_COSTS), and on the committed audit
    data the model's picks must stay close to the best pinned plan."""
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "ternary-vllm_amd", "csrc", "gemm_mfma.hip")).read()
    body = src[src.index("static const TileCost TILE_COSTS[]"):]
    body = body[:body.index("};")]
    kinds = {"KIND_BIG": 2, "KIND_MID": 1, "KIND_SMALL": 0}
    lib = [(kinds[m.group(1)],) + tuple(float(v) for v in m.group(2).split(","))
           for m in re.finditer(r"\{(KIND_\w+),([^}]*)\}", body)]
    py = open(os.path.join(root, "tools", "planner_model.py")).read()
    tab = py[py.index("TILE_COSTS = ["):]
    tab = tab[:tab.index("]\n")]
    names = {"BIG": 2, "MID": 1, "SMALL": 0}
    mine = [(names[m.group(1)],) + tuple(float(v) for v in m.group(2).split(","))
            for m in re.finditer(r"\((BIG|MID|SMALL),([^)]*)\)", tab)]
    assert lib == mine and len(lib) == 5
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "planner_model.py"), os.path.join(root, "profiles", "r02_planner_audit.jsonl")],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    regret = float(re.search(r"mean regret ([0-9.]+) %", out.stdout).group(1))
    assert regret < 1.5, out.stdout
