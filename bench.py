"""Benchmark of the packed ternary GEMM hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--m 4096 --n 4096 --k 4096] [--sweep] [--no-cpu]

`--gpus N` with N > 1, started as a plain process, launches N ranks itself (`python -m torch.distributed.run
--nproc-per-node N bench.py ...` as a child process, before anything touches a GPU), relays rank 0's JSON line and
exits with the child's status; started by torch.distributed.run (RANK / WORLD_SIZE in the environment) it is one of
the ranks and refuses to run when WORLD_SIZE differs from --gpus.  `--dry-run` replaces the GPU work by host
tensors over gloo: the launch, replication, sharding and timing protocol without a GPU (used by the CPU tests).

One step = one `ternary_matmul` (packed {-1,0,+1} weights x f16 activations, fused scale/bias/clip)
over one synthetic batch resident in HBM: M x K activations against an N x K weight matrix that was
ternarized and packed once, outside the timed region.  Default workload: 4096 x 4096 x 4096 (the
configuration the headline metric is quoted on).  With N > 1 ranks (one per GPU, launched by
torch.distributed.run) rank 0 quantizes + packs, the 2-bit weights are broadcast over RCCL, and each
rank runs the same M-row shard size (weak scaling, no collective on the timed path).

Rank 0 prints ONE JSON line.  `value` = total ternary-GEMM operations (2 M N K per step, all ranks)
per second in GOPS.  `roofline` prices the GEMM kernel against the dense f16 matrix-core peak (this
shape is compute-bound, 1 927 op/byte); `hbm` gives the same launch as a fraction of HBM bandwidth
(the metric's second half) plus the HBM-bound kernels of the path (quantize+pack, skinny-M GEMM).
`cpu_baseline` times a CPU restatement of the reference op on the host's cores (bounded sample); `parity` holds the
errors of the GPU path on this workload against the C restatement / float64 (outside the timed region); `configs` the
ViT-Base/16 (batch 128) and batch-256 decode shapes of the other benchmark configurations.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec
HBM_ACHIEVABLE_GBS = 6300.0  # what a plain 16-byte-per-lane copy reaches on this part (79 % of the spec figure)
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense f16/bf16 matrix-core peak


def gemm_bytes(m, n, k, out_bytes=2):
    """Algorithmic bytes of one launch: x f16 + packed W + scale,bias f32 + y."""
    return m * k * 2 + n * ((k + 3) // 4) + 2 * n * 4 + m * n * out_bytes


def time_launches(fn, steps, warmup, rounds=5):
    """Device time of one fn() in ms: one event pair around `steps` back-to-back launches on the current stream
    (the stream the library launches on), repeated `rounds` times.  Returns (average, best round).  A pair per
    launch would add the event markers' own few microseconds to every sample."""
    for i in range(warmup):
        fn(i)
    torch.cuda.synchronize()
    per = []
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(steps):
            fn(i)
        b.record()
        torch.cuda.synchronize()
        per.append(a.elapsed_time(b) / steps)
    return sum(per) / len(per), min(per)


def time_graph(fn, launches, warmup, rounds=5):
    """As time_launches, but the `launches` calls are captured into one HIP graph and the graph is replayed:
    the device time of kernels that are shorter than the host's cost of launching them (a few-row matmul is
    ~8 us of device time behind ~20 us of Python + ctypes per call).  Falls back to eager timing if the
    capture fails."""
    try:
        for i in range(warmup):
            fn(i)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for i in range(launches):
                fn(i)
        graph.replay()
        torch.cuda.synchronize()
        per = []
        for _ in range(rounds):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            graph.replay()
            b.record()
            torch.cuda.synchronize()
            per.append(a.elapsed_time(b) / launches)
        return sum(per) / len(per), min(per), "hip graph replay"
    except Exception as e:  # noqa: BLE001
        torch.cuda.synchronize()
        a, b = time_launches(fn, launches, warmup, rounds)
        return a, b, f"eager launches (graph capture failed: {type(e).__name__})"


def cpu_baseline(m_full, n, k, seed, budget_s=12.0):
    """Reference-semantics CPU run on the host cores: ScaledTernaryLinear.forward restated with torch CPU
    ops (binary/modules.py:101-109: ternarize on every call, dense float32 F.linear, *scale, +bias, clip)
    on a bounded row sample of the same workload."""
    threads = min(16, os.cpu_count() or 1)  # the GPU box gives one GPU's share of cores (16)
    torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(seed)
    w = torch.rand(n, k, generator=g) * 2 - 1
    scale = torch.rand(n, generator=g) + 0.5
    bias = torch.randn(n, generator=g)

    def fwd(x):
        wq = (w >= 0.5).float() - (w <= -0.5).float()
        y = torch.nn.functional.linear(x, wq)
        y *= scale
        y += bias
        return y.clip(-100.0, 100.0)

    ms = min(512, m_full)
    x = torch.randint(-64, 65, (ms, k), generator=g).float() / 64.0
    fwd(x)  # warm
    t0 = time.perf_counter(); fwd(x); dt = time.perf_counter() - t0
    # grow the sample towards the budget (bounded by the full M)
    ms2 = int(min(m_full, max(ms, ms * (budget_s / 3.0) / max(dt, 1e-6))))
    ms2 = max(64, (ms2 // 64) * 64)
    x = torch.randint(-64, 65, (ms2, k), generator=g).float() / 64.0
    reps, t_total = 0, 0.0
    while t_total < budget_s and reps < 50:
        t0 = time.perf_counter(); fwd(x); t_total += time.perf_counter() - t0
        reps += 1
    gops = 2.0 * ms2 * n * k * reps / t_total / 1e9
    out = {"value": round(gops, 2), "unit": "GOPS", "cores": threads, "kind": "port",
           "sample": f"{reps} x ScaledTernaryLinear.forward semantics in torch CPU float32 (per-call ternarize + dense matmul + "
                     f"scale/bias/clip) on {ms2} of {m_full} rows, N={n}, K={k}"}
    # the multiplication-free select-and-sum form (matrix_mul_tests.py:29-37) as a C/OpenMP port
    try:
        from __graft_entry__ import ORACLE_SO
        lib = ctypes.CDLL(ORACLE_SO)
        fnc = lib.tvo_ternary_matmul_select_f32
        fnc.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float, ctypes.c_void_p, ctypes.c_int]
        from oracle import ternary_oracle as O
        pw = np.ascontiguousarray(O.pack_format_c(O.ternarize(w.numpy())))
        msel = 128
        xs = np.ascontiguousarray(x[:msel].numpy())
        ys = np.empty((msel, n), np.float32)
        sc, bi = scale.numpy(), bias.numpy()
        t0 = time.perf_counter()
        rc = fnc(xs.ctypes.data, msel, k, pw.ctypes.data, pw.shape[1], n, sc.ctypes.data, bi.ctypes.data, 100.0, ys.ctypes.data, threads)
        dt = time.perf_counter() - t0
        if rc == 0:
            out["select_and_sum_c_port_gops"] = round(2.0 * msel * n * k / dt / 1e9, 2)
    except Exception as e:  # the C port is optional colour; the torch form above is the baseline
        out["select_and_sum_c_port_gops"] = f"unavailable: {e}"
    # configuration 1: the reference's ternary_mm loop (matrix_mul_tests.py:29-37) as its torch CPU restatement, 1024 x 1024
    mm_, tm_ = torch.randn(1024, 1024, generator=g), torch.randn(1024, 1024, generator=g)
    tm_[torch.rand(1024, 1024, generator=g) < 0.5] = 0.0
    t0 = time.perf_counter()
    res = torch.zeros(1024, 1024)
    for i in range(1024):
        r = tm_[i]
        res[:, i] = mm_[:, r > 0].sum(dim=1) - mm_[:, r < 0].sum(dim=1)
    out["ternary_mm_1024_loop_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
    return out


def parity_block(ops, dev, pw, scale, bias, x_randn, M, N, K, threads):
    """Errors of the GPU path on the benchmark's own workload, checked with the C restatement (oracle/ternary_oracle.c) outside
    every timed region: (1) dyadic-grid activations (every partial sum exact in float32, so any correct kernel gives the oracle's
    bits): the full M x N product on the GPU, every 8th row compared; (2) the N(0,1) activations of the timed runs: 48 rows
    against a float64 evaluation, next to the error of a float32 CPU evaluation of the same rows."""
    from __graft_entry__ import ORACLE_SO
    lib = ctypes.CDLL(ORACLE_SO)
    fn = lib.tvo_ternary_matmul_dense_f32
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p,
                   ctypes.c_void_p, ctypes.c_float, ctypes.c_void_p, ctypes.c_int]
    pw_h = np.ascontiguousarray(pw.cpu().numpy())
    sc_h, bi_h = np.ascontiguousarray(scale.cpu().numpy()), np.ascontiguousarray(bias.cpu().numpy())

    def cpu_f32(x_rows):
        x_rows = np.ascontiguousarray(x_rows, dtype=np.float32)
        y = np.empty((x_rows.shape[0], N), np.float32)
        rc = fn(x_rows.ctypes.data, x_rows.shape[0], K, pw_h.ctypes.data, pw_h.shape[1], N, sc_h.ctypes.data, bi_h.ctypes.data, 100.0, y.ctypes.data, threads)
        assert rc == 0
        return y

    g = torch.Generator().manual_seed(4321)
    xg = (torch.randint(-64, 65, (M, K), generator=g).float() / 64.0).half()
    yg = ops.ternary_matmul(xg.to(dev), pw, scale, bias, 100.0, out_dtype=torch.float32).cpu().numpy()
    rows = np.arange(0, M, 8)
    grid_err = float(np.abs(yg[rows] - cpu_f32(xg[rows].float().numpy())).max())
    # N(0,1): float64 evaluation of 48 rows spread over the batch
    rr = np.linspace(0, M - 1, 48).astype(np.int64)
    yr = ops.ternary_matmul(x_randn, pw, scale, bias, 100.0, out_dtype=torch.float32)[torch.from_numpy(rr).to(dev)].cpu().numpy()
    xr = x_randn[torch.from_numpy(rr).to(dev)].float().cpu().numpy()
    codes = pw_h[:, :K // 4]
    w64 = np.stack([((codes >> (2 * i)) & 3).astype(np.float64) - 1.0 for i in range(4)], axis=-1).reshape(N, K)
    y64 = np.clip((xr.astype(np.float64) @ w64.T) * sc_h.astype(np.float64) + bi_h.astype(np.float64), -100.0, 100.0)
    return {"grid_max_abs": grid_err, "grid_rows_compared": int(rows.size), "grid_note": "x = k/64: exact partial sums, expected 0",
            "randn_max_abs_vs_f64": float(np.abs(yr - y64).max()), "cpu_f32_max_abs_vs_f64": float(np.abs(cpu_f32(xr) - y64).max()),
            "randn_rows_compared": 48, "randn_output_rms": float(np.sqrt(np.mean(y64 ** 2))),
            "checker": "oracle/ternary_oracle.c tvo_ternary_matmul_dense_f32 (float32) and numpy float64; f16 activations, float32 GPU output"}


def launch_ranks(n):
    """Start n ranks of this script under torch.distributed.run as a CHILD process (this process has not touched a GPU and
    never replaces itself), pass the child's output through, and return its exit status."""
    import socket
    import subprocess
    port = 0
    argv = [{"--m": "--gemm-m", "--n": "--gemm-n", "--k": "--gemm-k"}.get(a, a) for a in sys.argv[1:]]
    if "--master-port" in argv:
        i = argv.index("--master-port")
        port = int(argv[i + 1])
        del argv[i:i + 2]
    if not port:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    if proc.returncode == 0 and len(lines) != 1:
        print(f"bench.py: expected one JSON line from rank 0, got {len(lines)}", file=sys.stderr)
        return 3
    for ln in lines:
        print(ln, flush=True)
    return proc.returncode


def dry_run(args, world, rank):
    """The multi-rank protocol on host tensors over gloo, no GPU and no kernel: rank 0 builds a packed arena (random 2-bit
    codes: quantize + pack is a GPU kernel), PackedArena.replicate hands it to every rank, rows are assigned by shard_rows,
    the timed region is bracketed by barriers and reduced by MAX, and rank 0 prints the JSON line with value 0."""
    import torch.distributed as dist
    import ternary_vllm_amd as tv
    from ternary_vllm_amd.conversion import ArenaEntry, PackedArena
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "RANK" not in os.environ:
        os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_PORT=str(args.master_port or 29533))
    dist.init_process_group("gloo")
    M, N, K = args.m, args.n, args.k
    arena = None
    if rank == 0:
        ld = -(-((K + 3) // 4) // 16) * 16
        g = torch.Generator().manual_seed(1234)
        arena = PackedArena(torch.randint(0, 171, (N * ld,), dtype=torch.uint8, generator=g), torch.rand(N, generator=g) + 0.5,
                            torch.randn(N, generator=g), [ArenaEntry("w", N, K, ld, 0, 0, 100.0, True, "ternary")])
    dist.barrier()
    t0 = time.perf_counter()
    arena = PackedArena.replicate(arena, "cpu", src=0)
    bcast_ms = (time.perf_counter() - t0) * 1e3
    pw, scale, bias, _ = arena.layer("w")
    begin, end = tv.shard_rows(M * world, world, rank)
    sums = [torch.zeros(3, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(sums, torch.tensor([float(pw.long().sum()), float(begin), float(end)], dtype=torch.float64))
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pass                                   # the GPU step would go here
    dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ok = all(float(s[0]) == float(sums[0][0]) for s in sums)
    spans = [(int(s[1]), int(s[2])) for s in sums]
    ok = ok and spans[0][0] == 0 and spans[-1][1] == M * world and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    if rank == 0:
        print(json.dumps({"metric": "dry run: multi-rank protocol only", "value": 0.0, "unit": "GOPS", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": round(float(t.item()) / max(1, args.steps) * 1e3, 6), "dry_run": True,
                          "ranks_reported": len(sums), "row_spans": spans, "replicated_arena_identical": ok,
                          "weight_broadcast_ms": round(bcast_ms, 3), "weight_broadcast_bytes": arena.nbytes(), "scaling": "weak"}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return 0 if ok else 4


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=100)
    # (the long spellings exist because torch.distributed.run's own parser rejects "--m" / "--n" as ambiguous abbreviations)
    ap.add_argument("--m", "--gemm-m", dest="m", type=int, default=4096)
    ap.add_argument("--n", "--gemm-n", dest="n", type=int, default=4096)
    ap.add_argument("--k", "--gemm-k", dest="k", type=int, default=4096)
    ap.add_argument("--sweep", action="store_true", help="also time the M sweep and the moondream layer shapes")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--x-buffers", type=int, default=0, help="rotating activation / output buffers (default: enough to exceed the Infinity Cache)")
    ap.add_argument("--no-side", action="store_true", help="skip the side measurements (pack kernel, few-row shapes, float32, vendor GEMM)")
    ap.add_argument("--dry-run", action="store_true", help="host tensors over gloo, no GPU work: the multi-rank protocol only")
    ap.add_argument("--master-port", type=int, default=0, help="rendezvous port when this process launches the ranks itself")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")

    under_launcher = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not under_launcher:
        sys.exit(launch_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1")) if under_launcher else 1
    rank = int(os.environ.get("RANK", "0")) if under_launcher else 0
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if under_launcher else 0
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}", file=sys.stderr)
        sys.exit(2)
    if args.dry_run:
        sys.exit(dry_run(args, world, rank))

    # One of the ranks of torch.distributed.run (also with a single rank: the same collectives then run on one GPU).
    dist = None
    if under_launcher:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    import ternary_vllm_amd as tv
    from ternary_vllm_amd.conversion import PackedArena
    ops = tv.custom_ops_mod

    M, N, K = args.m, args.n, args.k
    seed = 1234
    g = torch.Generator().manual_seed(seed)
    w_lat = torch.rand(N, K, generator=g) * 2 - 1          # latent weights U(-1,1): ~25/50/25 % -1/0/+1 at threshold 0.5
    scale = (torch.rand(N, generator=g) + 0.5).to(dev)      # U(0.5, 1.5)
    bias = torch.randn(N, generator=g).to(dev)

    # ---- one-time quantize + pack on rank 0; the 2-bit arena is replicated to the other ranks (the path's only exchange) ----
    extra = {}
    arena = PackedArena.from_layers({"w": (w_lat.to(dev), scale, bias, 100.0, "ternary")}, dev) if rank == 0 else None
    if dist is not None:
        torch.cuda.synchronize(); dist.barrier()
        t0 = time.perf_counter()
        arena = PackedArena.replicate(arena, dev, src=0)
        torch.cuda.synchronize()
        extra["weight_broadcast_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
        extra["weight_broadcast_bytes"] = arena.nbytes()
    pw, scale, bias, _ = arena.layer("w")
    # rows of x are independent: the global batch of world * M rows is split evenly, every rank multiplies its own M rows
    row_begin, row_end = tv.shard_rows(M * world, world, rank)
    assert row_end - row_begin == M

    # ---- synthetic activations: rotate over enough buffers that no step finds its x in a cache: the outputs leave
    # through non-temporal stores and do not occupy the 256 MiB Infinity Cache, so the activation buffers alone must
    # exceed it (12 x 32 MiB at 4096 x 4096)
    nbuf = max(6, -(-(400 << 20) // max(1, M * K * 2)))
    nbuf = min(nbuf, 64)
    if args.x_buffers > 0:
        nbuf = args.x_buffers
    gx = torch.Generator().manual_seed(seed + 1 + row_begin)
    xs = [(torch.randn(M, K, generator=gx)).to(dev).half() for _ in range(nbuf)]
    ys = [torch.empty((M, N), dtype=torch.float16, device=dev) for _ in range(nbuf)]

    def step(i):
        ops.ternary_matmul(xs[i % nbuf], pw, scale, bias, 100.0, out=ys[i % nbuf])

    # Set-up, untimed: bring the GPU out of its idle power state (the first milliseconds of matrix-core work
    # after idling run at a lower clock), then the W warm-up steps the caller asked for.
    PREWARM = 300
    for i in range(PREWARM):
        step(i)
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for i in range(args.steps):
        step(i)
    ev1.record()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    event_ms_per_launch = ev0.elapsed_time(ev1) / args.steps   # device time of the timed region itself
    ranks_reported = 1
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        c = torch.ones(1, dtype=torch.int32, device=dev)
        dist.all_reduce(c)
        ranks_reported = int(c.item())
        assert ranks_reported == args.gpus, f"{ranks_reported} ranks took part, --gpus {args.gpus}"

    ops_per_step = 2.0 * M * N * K * world
    value = ops_per_step * args.steps / elapsed / 1e9

    result = None
    if rank == 0:
        # ---- roofline of the dominant kernel, from per-launch device time (events on the launch stream)
        avg_ms = event_ms_per_launch
        flops = 2.0 * M * N * K
        achieved_tf = flops / (avg_ms * 1e-3) / 1e12
        by = gemm_bytes(M, N, K, 2)
        big = (M + 255) // 256 * ((N + 255) // 256) >= 192
        traffic, traffic_note = None, None
        try:   # HBM-side bytes per launch from the committed rocprofv3 --pmc passes: valid for the profiled binary and shape only
            import hashlib
            tpath = next(p_ for p_ in (os.path.join(ROOT, "profiles", f"r0{r_}_traffic.json") for r_ in (3, 2)) if os.path.exists(p_))
            with open(tpath) as f:
                tj = json.load(f)
            with open(ops.LIBRARY_PATH, "rb") as f:
                same_binary = hashlib.sha256(f.read()).hexdigest() == tj.get("library_sha256")
            if (M, N, K) == (4096, 4096, 4096) and same_binary:
                traffic = tj["gemm_4096x4096x4096_f16"]["traffic_bytes"]
            elif (M, N, K) == (4096, 4096, 4096):
                traffic_note = f"profiles/{os.path.basename(tpath)} was collected with another build of the library: not reported"
        except (OSError, KeyError, ValueError, StopIteration):
            pass
        roofline = {"bound": "mfma", "achieved": round(achieved_tf, 2), "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved_tf / MFMA_F16_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_note": traffic_note,
                    "kernel": "tvl::gemm_mfma16_kernel<f16, f16, bias, none> (256x256 tile, v_mfma_f32_16x16x32_f16)" if big else "tvl::gemm_mfma_kernel (small tiles / split-K)",
                    "avg_launch_ms": round(avg_ms, 4), "flops_per_launch": flops,
                    "timing": f"one HIP event pair around the {args.steps} timed launches on the launch stream"}
        hbm = {"gemm_algorithmic_bytes": by, "gemm_GBps": round(by / (avg_ms * 1e-3) / 1e9, 1),
               "gemm_frac_of_hbm_peak": round(by / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
               "note": "4096^3 with f16 activations is matrix-core bound (1927 op/B): its HBM fraction is capped near 16 %"}

        if world == 1 and not args.no_side:   # the side measurements are single-GPU figures; multi-GPU runs report the sharded GEMM only
            # ---- HBM-bound kernel of the path: quantize + pack (f32 in, 2 bits out), cache-cold buffers
            nw = 6  # 6 x 64 MiB of latent weights > 256 MiB Infinity Cache
            wl = [torch.rand(N, K, device=dev) * 2 - 1 for _ in range(nw)]
            pws = [torch.empty_like(pw) for _ in range(nw)]
            p_avg, p_best = time_launches(lambda i: ops.ternarize_pack(wl[i % nw], 0.5, out=pws[i % nw]), 2 * nw, nw)
            pack_bytes = N * K * 4 + N * ((K + 3) // 4)
            hbm["pack_kernel"] = {"kernel": "tvl::quantize_pack_fast_kernel<float, ternary>", "shape": [N, K],
                                  "algorithmic_bytes": pack_bytes, "avg_launch_ms": round(p_avg, 4),
                                  "GBps": round(pack_bytes / (p_avg * 1e-3) / 1e9, 1),
                                  "frac_of_hbm_peak": round(pack_bytes / (p_avg * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                  "frac_of_achievable_6300": round(pack_bytes / (p_avg * 1e-3) / 1e9 / HBM_ACHIEVABLE_GBS, 4)}
            del wl, pws
            # the arena case: every quantized layer of a moondream-sized checkpoint (22 blocks x Wqkv, out_proj, fc1, fc2) in one
            # batched call - the form the conversion path really runs (utils/conversion.py:126-137, one pack per Dense)
            shapes = [(6144, 2048), (2048, 2048), (8192, 2048), (2048, 8192)] * 22
            lat = [torch.rand(n_, k_, device=dev) * 2 - 1 for (n_, k_) in shapes]
            dst = [torch.empty((n_, k_ // 4), dtype=torch.uint8, device=dev) for (n_, k_) in shapes]
            a_avg, _, a_how = time_graph(lambda i: ops.ternarize_pack_batch(lat, 0.5, "ternary", outs=dst), 3, 1, rounds=3)
            a_bytes = sum(n_ * k_ * 4 + n_ * k_ // 4 for (n_, k_) in shapes)
            hbm["pack_arena"] = {"kernel": "tvl::quantize_pack_batch_kernel<float, ternary> (up to 64 matrices per launch)", "layers": len(shapes), "timing": a_how,
                                 "algorithmic_bytes": a_bytes, "avg_call_ms": round(a_avg, 4), "GBps": round(a_bytes / (a_avg * 1e-3) / 1e9, 1),
                                 "frac_of_hbm_peak": round(a_bytes / (a_avg * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                 "frac_of_achievable_6300": round(a_bytes / (a_avg * 1e-3) / 1e9 / HBM_ACHIEVABLE_GBS, 4)}
            del lat, dst
            # one large matrix (1 GiB of latent weights): the kernel's rate without launch ramps and tails
            big_w = torch.rand(16384, 16384, device=dev) * 2 - 1
            big_o = torch.empty((16384, 4096), dtype=torch.uint8, device=dev)
            b_avg, _, b_how = time_graph(lambda i: ops.ternarize_pack(big_w, 0.5, out=big_o), 3, 1, rounds=3)
            b_bytes = 16384 * 16384 * 4 + 16384 * 4096
            hbm["pack_16384"] = {"kernel": "tvl::quantize_pack_fast_kernel<float, ternary>", "shape": [16384, 16384], "timing": b_how,
                                 "algorithmic_bytes": b_bytes, "avg_launch_ms": round(b_avg, 4), "GBps": round(b_bytes / (b_avg * 1e-3) / 1e9, 1),
                                 "frac_of_hbm_peak": round(b_bytes / (b_avg * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                 "frac_of_achievable_6300": round(b_bytes / (b_avg * 1e-3) / 1e9 / HBM_ACHIEVABLE_GBS, 4)}
            del big_w, big_o
            # ---- HBM-bound form of the matmul: few activation rows (decode / small batches), weights streamed once from HBM.
            # 16384 x 16384 (64 MiB packed, 10 rotating copies) is large enough to leave the launch-latency regime.
            for (dm, dn, dk, tag) in ((1, 16384, 16384, "decode_m1_16384"), (1, N, K, "decode_m1"), (16, N, K, "rows16"), (64, N, K, "rows64"),
                                      (16, 16384, 16384, "rows16_16384"), (64, 16384, 16384, "rows64_16384")):
                nd = max(2, min(40, (640 << 20) // (dn * dk // 4)))
                pwd = [torch.randint(0, 171, (dn, dk // 4), dtype=torch.uint8, device=dev) for _ in range(nd)]
                xd = torch.randn(dm, dk, device=dev).half()
                yd = torch.empty((dm, dn), dtype=torch.float16, device=dev)
                sd, bd = torch.ones(dn, device=dev), torch.zeros(dn, device=dev)
                d_avg, d_best, how = time_graph(lambda i: ops.ternary_matmul(xd, pwd[i % nd], sd, bd, 100.0, out=yd), 2 * nd, nd)
                d_bytes = gemm_bytes(dm, dn, dk, 2)
                hbm[tag] = {"shape": [dm, dn, dk], "timing": how, "algorithmic_bytes": d_bytes, "avg_call_ms": round(d_avg, 5),
                            "GBps": round(d_bytes / (d_avg * 1e-3) / 1e9, 1), "frac_of_hbm_peak": round(d_bytes / (d_avg * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
                del pwd

            # the cost of one kernel node of a replayed HIP graph (a 64-element zero_out): the floor under every short launch above
            tiny = torch.zeros(64, dtype=torch.int32, device=dev)
            f_avg, _, f_how = time_graph(lambda i: ops.zero_out(tiny), 50, 5)
            hbm["graph_node_floor"] = {"avg_call_ms": round(f_avg, 5), "timing": f_how,
                                       "note": "one empty-ish kernel per graph node: launch-bound points (decode_m1, rows16, rows64 at 4096^2) sit on this floor plus their own work"}

            # ---- the vendor's dense f16 GEMM (hipBLASLt through torch.matmul) on the same shape, buffers and pre-warm: a yardstick
            wd = ops.unpack_ternary(pw, K, dtype=torch.float16)
            for i in range(PREWARM):
                torch.matmul(xs[i % nbuf], wd.t(), out=ys[i % nbuf])
            v_avg, v_best = time_launches(lambda i: torch.matmul(xs[i % nbuf], wd.t(), out=ys[i % nbuf]), args.steps, args.warmup, rounds=3)
            roofline["vendor_dense_f16_ms"] = round(v_avg, 4)
            roofline["vendor_dense_f16_frac"] = round(flops / (v_avg * 1e-3) / 1e12 / MFMA_F16_PEAK_TFLOPS, 4)
            del wd

            # ---- configuration 1: the reference's ternary_mm micro-benchmark (matrix_mul_tests.py: 1024 x 1024, 50 % zeros; 256 x 256)
            cfg1 = {}
            for size in (256, 1024):
                mm_ = torch.randn(size, size, device=dev)
                tm_ = torch.randn(size, size, device=dev)
                tm_[torch.rand(size, size, device=dev) < 0.5] = 0.0
                t_avg, _, how = time_graph(lambda i: ops.ternary_mm(mm_, tm_), 20, 5)
                cfg1[str(size)] = {"avg_call_ms": round(t_avg, 5), "GOPS": round(2.0 * size ** 3 / (t_avg * 1e-3) / 1e9, 1), "timing": how}
            extra["config1_ternary_mm"] = cfg1

            # ---- BASELINE configurations 3 and 4 under the same clock: the ViT-Base/16 Dense stack at batch 128 (M = 128 x 197 tokens) and
            # the packed matmuls of one batch-256 decode step of the Phi-shaped text model (Wqkv | fc1 as one product, GELU from 6144)
            cfg_rows = []
            for (mm, nn, kk, tag, act, act_begin) in (
                    (128 * 197, 2304, 768, "config3 vit_b16 qkv", None, 0), (128 * 197, 768, 768, "config3 vit_b16 attn_out", None, 0),
                    (128 * 197, 3072, 768, "config3 vit_b16 fc1+gelu", "gelu_tanh", 0), (128 * 197, 768, 3072, "config3 vit_b16 fc2", None, 0),
                    (256, 14336, 2048, "config4 decode_b256 Wqkv|fc1 (gelu from 6144)", "gelu_tanh", 6144),
                    (256, 2048, 2048, "config4 decode_b256 out_proj", None, 0), (256, 2048, 8192, "config4 decode_b256 fc2", None, 0)):
                nb = max(2, min(16, int(600e6 // max(1, nn * kk // 4 + mm * kk * 2))))   # operands rotate through > 256 MiB where they are small
                pwl = [torch.randint(0, 171, (nn, kk // 4), dtype=torch.uint8, device=dev) for _ in range(nb)]
                xl = [torch.randn(mm, kk, device=dev).half() for _ in range(nb)]
                yl = [torch.empty((mm, nn), dtype=torch.float16, device=dev) for _ in range(nb)]
                sc_, bi_ = torch.rand(nn, device=dev) + 0.5, torch.randn(nn, device=dev)
                call = lambda i: ops.ternary_matmul(xl[i % nb], pwl[i % nb], sc_, bi_, 100.0, activation=act, act_begin=act_begin, out=yl[i % nb])  # noqa: E731
                if 2.0 * mm * nn * kk < 2e10:
                    a_ms, _, how = time_graph(call, 2 * nb, nb)
                else:
                    a_ms, _ = time_launches(call, 60, 100, rounds=3)
                    how = "event pair around 60 launches"
                by_ = gemm_bytes(mm, nn, kk, 2)
                cfg_rows.append({"name": tag, "shape": [mm, nn, kk], "us": round(a_ms * 1e3, 2),
                                 "frac_mfma": round(2.0 * mm * nn * kk / (a_ms * 1e-3) / 1e12 / MFMA_F16_PEAK_TFLOPS, 4),
                                 "frac_hbm": round(by_ / (a_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "kernel": ops.matmul_route(torch.float16, mm, nn, kk),
                                 "timing": how})
                del pwl, xl, yl
            extra["configs"] = cfg_rows

            # ---- the same workload with float32 activations (the reference layer's dtype): three exact bf16 planes on
            # the matrix cores, float32 output
            xf = torch.randn(M, K, device=dev)
            yf = torch.empty((M, N), dtype=torch.float32, device=dev)
            f_avg, _ = time_launches(lambda i: ops.ternary_matmul(xf, pw, scale, bias, 100.0, out=yf), 10, 3, rounds=3)
            extra["f32_activations"] = {"shape": [M, N, K], "avg_call_ms": round(f_avg, 4),
                                        "effective_TFLOPs": round(2.0 * M * N * K / (f_avg * 1e-3) / 1e12, 1),
                                        "kernels": "tvl::split_f32_kernel + tvl::gemm_mfma16_kernel<bf16, f32> over 3K"}
            del xf, yf

        if args.sweep and world == 1:
            sweep = []
            shapes = [(mm, 4096, 4096) for mm in (1, 16, 64, 256, 1024, 4096)]
            shapes += [(mm, nn, kk) for mm in (1, 745) for (nn, kk) in ((6144, 2048), (2048, 2048), (8192, 2048), (2048, 8192))]
            shapes += [(128 * 197, nn, kk) for (nn, kk) in ((2304, 768), (768, 768), (3072, 768), (768, 3072))]   # ViT-Base/16, batch 128
            for (mm, nn, kk) in shapes:
                nb = max(2, min(16, int(600e6 // max(1, nn * kk // 4 + mm * kk * 2))))  # rotate > 256 MiB where small
                pwl = [torch.randint(0, 171, (nn, kk // 4), dtype=torch.uint8, device=dev) for _ in range(nb)]
                xl = [torch.randn(mm, kk, device=dev).half() for _ in range(nb)]
                yl = [torch.empty((mm, nn), dtype=torch.float16, device=dev) for _ in range(nb)]
                sc = torch.ones(nn, device=dev); bi = torch.zeros(nn, device=dev)
                call = lambda i: ops.ternary_matmul(xl[i % nb], pwl[i % nb], sc, bi, 100.0, out=yl[i % nb])  # noqa: E731
                if 2.0 * mm * nn * kk < 2e10:   # short kernels: device time through a graph, not host launch rate
                    a, _, _ = time_graph(call, 2 * nb, nb)
                else:
                    a, _ = time_launches(call, max(2 * nb, 100), 300)   # like the headline: out of the idle clock state first
                b_ = gemm_bytes(mm, nn, kk, 2)
                sweep.append({"m": mm, "n": nn, "k": kk, "avg_ms": round(a, 5), "GOPS": round(2.0 * mm * nn * kk / (a * 1e-3) / 1e9, 1),
                              "GBps": round(b_ / (a * 1e-3) / 1e9, 1), "frac_hbm": round(b_ / (a * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                              "frac_mfma": round(2.0 * mm * nn * kk / (a * 1e-3) / 1e12 / MFMA_F16_PEAK_TFLOPS, 4)})
                del pwl, xl, yl
            extra["sweep"] = sweep

        result = {
            "metric": "ternary GEMM GOPS + HBM GB/s (% roofline), 4096x4096x4096, 1 GPU" if world == 1 else
                      "ternary GEMM GOPS + HBM GB/s (% roofline), 4096x4096x4096 per GPU",
            "value": round(value, 1), "unit": "GOPS", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"{M}x{N}x{K} ternary x f16 GEMM, 2-bit packed weights, fused scale/bias/clip, f16 out",
                       "m_per_gpu": M, "n": N, "k": K, "rotating_activation_buffers": nbuf, "sharding": "rows of x (M) across ranks; packed weights replicated by broadcast"},
            "roofline": roofline, "hbm": hbm, "prewarm_launches": PREWARM, "ranks_reported": ranks_reported,
        }
        if not args.no_cpu and world == 1:
            result["cpu_baseline"] = cpu_baseline(M, N, K, seed)
            try:   # the checker's verdict on this very workload, beside the CPU baseline it shares the host leg with
                result["parity"] = parity_block(ops, dev, pw, scale, bias, xs[0], M, N, K, result["cpu_baseline"]["cores"])
            except Exception as e:  # noqa: BLE001
                result["parity"] = {"error": f"{type(e).__name__}: {e}"}
        else:
            result["cpu_baseline"] = None
        result.update(extra)
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
