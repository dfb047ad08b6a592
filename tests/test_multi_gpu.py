"""The RCCL leg on real hardware, as far as one GPU allows: bench.py as ONE rank under torch.distributed.run (the same code
path the N-rank launches take: nccl process group, arena replication by broadcast, barrier-bracketed timing, rank count by
all-reduce), and PackedArena.replicate on GPU tensors inside a one-rank nccl group.  Two or more ranks need as many GPUs:
the world-size-2 test over gloo in test_host_cpu.py covers the receiving side of the same replicate / shard code."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _json_line(cmd):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", PYTHONUNBUFFERED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, proc.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_as_one_rank_under_the_launcher_uses_rccl():
    """`python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1` (a child process: this one keeps its GPU state) next to the
    plain single-process run of the same command line."""
    flags = ["--gpus", "1", "--no-side", "--no-cpu", "--steps", "50", "--warmup", "5"]
    ranked = _json_line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
                         "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py")] + flags)
    plain = _json_line([sys.executable, os.path.join(ROOT, "bench.py")] + flags)
    assert ranked["n_gpus"] == 1 and ranked["ranks_reported"] == 1 and ranked["scaling"] == "weak"
    assert ranked["weight_broadcast_bytes"] == 4096 * 1024 + 2 * 4096 * 4       # 2-bit weights + float32 scales and biases of 4096 x 4096
    assert ranked["weight_broadcast_ms"] > 0
    assert "weight_broadcast_bytes" not in plain and plain["ranks_reported"] == 1
    assert ranked["metric"] == plain["metric"] and ranked["config"] == plain["config"]
    assert abs(ranked["value"] - plain["value"]) <= 0.10 * plain["value"], (ranked["value"], plain["value"])


def test_arena_replicate_on_gpu_tensors_in_a_one_rank_nccl_group(dev):
    import torch.distributed as dist

    import ternary_vllm_amd as tv
    assert not dist.is_initialized()
    torch.manual_seed(2)
    layers = {f"blk.{i}.fc": (torch.rand(n, k, device=dev) * 2 - 1, torch.rand(n, device=dev) + 0.5, torch.randn(n, device=dev), 100.0, "ternary")
              for i, (n, k) in enumerate([(256, 512), (96, 200), (2048, 2048)])}
    arena = tv.PackedArena.from_layers(layers, dev)
    before = (arena.packed.cpu().clone(), arena.scales.cpu().clone(), arena.biases.cpu().clone())
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1, device_id=dev)
    try:
        out = tv.PackedArena.replicate(arena, dev, src=0)
        torch.cuda.synchronize()
        assert out.packed.is_cuda and list(out.entries) == list(arena.entries)
        assert torch.equal(out.packed.cpu(), before[0]) and torch.equal(out.scales.cpu(), before[1]) and torch.equal(out.biases.cpu(), before[2])
        # the receiving side's buffers (what ranks other than the source allocate), filled by the same collectives
        recv = tv.PackedArena.empty_like_index(list(arena.entries.values()), dev)
        assert recv.packed.numel() == arena.packed.numel() and recv.nbytes() == arena.nbytes()
        for dst, src in ((recv.packed, arena.packed), (recv.scales, arena.scales), (recv.biases, arena.biases)):
            dst.copy_(src)
        recv.broadcast(src=0)
        torch.cuda.synchronize()
        assert torch.equal(recv.packed.cpu(), before[0])
        begin, end = tv.shard_rows(1000, dist.get_world_size(), dist.get_rank())
        assert (begin, end) == (0, 1000)
    finally:
        dist.destroy_process_group()
