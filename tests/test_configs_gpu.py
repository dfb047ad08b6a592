"""The benchmark configurations at their own shapes, every output element against the C restatement of the reference op.

ViT-Base/16 Dense stack at batch 128 (M = 128 * 197 = 25 216 rows against the four layer shapes), the batch-256 decode
step of the ternary VLM (M = 256 against Wqkv | fc1 as one packed matrix with the GELU from channel 6144 on, out_proj,
fc2) and the headline 4096^3 shape.  Inputs are on the dyadic grid k/64, where every partial sum is exact in float32,
so the HIP result must equal the CPU restatement bit for bit whatever the summation order; N(0,1) inputs are compared
with the float64 oracle on a row sample."""
import numpy as np
import pytest
import torch

from oracle import ternary_oracle as O

pytestmark = pytest.mark.gpu


def _problem(seed, m, n, k):
    rng = np.random.default_rng(seed)
    x = rng.integers(-64, 65, size=(m, k)).astype(np.float32) / 64.0
    w = rng.uniform(-1, 1, size=(n, k)).astype(np.float32)
    scale = (rng.integers(4, 13, size=(n,)) / 8.0).astype(np.float32)
    bias = (rng.integers(-16, 17, size=(n,)) / 8.0).astype(np.float32)
    return x, w, scale, bias


VIT_B16 = [(2304, 768), (768, 768), (3072, 768), (768, 3072)]    # qkv, attention out, mlp fc1, mlp fc2


@pytest.mark.parametrize("n,k", VIT_B16)
def test_vit_base_batch128_full_output(ops, dev, c_oracle, n, k):
    m = 128 * 197
    x, w, scale, bias = _problem(n + k, m, n, k)
    pw = ops.ternarize_pack(torch.from_numpy(w).to(dev))
    want_pw = O.pack_format_c(O.ternarize(w))
    assert np.array_equal(pw.cpu().numpy(), want_pw)
    want = c_oracle(x, want_pw, scale, bias, 100.0)
    st, bt = torch.from_numpy(scale).to(dev), torch.from_numpy(bias).to(dev)
    for xdt in (torch.float16, torch.bfloat16):
        got = ops.ternary_matmul(torch.from_numpy(x).to(dev).to(xdt), pw, st, bt, 100.0, out_dtype=torch.float32)
        assert np.array_equal(got.cpu().numpy(), want), xdt
        del got
    got16 = ops.ternary_matmul(torch.from_numpy(x).to(dev).half(), pw, st, bt, 100.0)      # f16 out: one rounding of the same value
    assert torch.equal(got16.cpu(), torch.from_numpy(want).half())


DECODE_256 = [(6144 + 8192, 2048, 6144), (2048, 2048, 0), (2048, 8192, 0)]   # Wqkv | fc1 (GELU from 6144), out_proj, fc2


@pytest.mark.parametrize("n,k,act_begin", DECODE_256)
@pytest.mark.parametrize("m", [256, 2048])      # batch 256 per GPU; batch 2048 = the eight-GPU configuration on one device
def test_decode_batch_full_output(ops, dev, c_oracle, m, n, k, act_begin):
    x, w, scale, bias = _problem(m + n + k, m, n, k)
    scale *= 0.125
    pw = ops.ternarize_pack(torch.from_numpy(w).to(dev))
    want_pw = O.pack_format_c(O.ternarize(w))
    assert np.array_equal(pw.cpu().numpy(), want_pw)
    lin = c_oracle(x, want_pw, scale, bias, 100.0)
    st, bt = torch.from_numpy(scale).to(dev), torch.from_numpy(bias).to(dev)
    for xdt in (torch.float16, torch.bfloat16):
        xt = torch.from_numpy(x).to(dev).to(xdt)
        if act_begin:
            got = ops.ternary_matmul(xt, pw, st, bt, 100.0, activation="gelu_tanh", act_begin=act_begin, out_dtype=torch.float32).cpu().numpy()
            assert np.array_equal(got[:, :act_begin], lin[:, :act_begin]), xdt                    # linear channels: bit-exact
            want_act = O.gelu_tanh(lin[:, act_begin:]).astype(np.float32)                         # float64 tanh, rounded once
            assert np.max(np.abs(got[:, act_begin:] - want_act)) <= 1e-5, xdt
        else:
            got = ops.ternary_matmul(xt, pw, st, bt, 100.0, out_dtype=torch.float32).cpu().numpy()
            assert np.array_equal(got, lin), xdt


def test_headline_4096_full_output_and_randn_rows(ops, dev, c_oracle):
    """4096 x 4096 x 4096 f16: all 16.7 M outputs against the C restatement on the exact grid, then the benchmark's own
    data (N(0,1) activations, U(-1,1) latent weights, seed 1234) on a row sample against the float64 oracle, next to
    the error a float32 CPU evaluation of the same rows makes."""
    m = n = k = 4096
    x, w, scale, bias = _problem(4096, m, n, k)
    w[0, :4] = [0.5, -0.5, 0.4999, -0.4999]
    pw = ops.ternarize_pack(torch.from_numpy(w).to(dev))
    want_pw = O.pack_format_c(O.ternarize(w))
    assert np.array_equal(pw.cpu().numpy(), want_pw)
    want = c_oracle(x, want_pw, scale, bias, 100.0)
    st, bt = torch.from_numpy(scale).to(dev), torch.from_numpy(bias).to(dev)
    got = ops.ternary_matmul(torch.from_numpy(x).to(dev).half(), pw, st, bt, 100.0, out_dtype=torch.float32)
    assert np.array_equal(got.cpu().numpy(), want)
    del got, want
    g = torch.Generator().manual_seed(1235)
    xr = torch.randn(m, k, generator=g).half()
    got = ops.ternary_matmul(xr.to(dev), pw, st, bt, 1e6, out_dtype=torch.float32).cpu().numpy()
    rows = np.random.default_rng(1).integers(0, m, size=48)
    xs = xr[rows].float().numpy()
    want64 = O.ternary_matmul(xs, want_pw, scale, bias, 1e6)                       # float64 accumulation
    cpu32 = c_oracle(xs, want_pw, scale, bias, 1e6)                                # float32 accumulation on the CPU
    err_gpu, err_cpu = float(np.max(np.abs(got[rows] - want64))), float(np.max(np.abs(cpu32 - want64)))
    # float32 accumulation error scales with the sum of the magnitudes of the terms (here ~1600 per output, so one ulp of
    # a partial sum is ~1e-5 already): the gate is relative to that sum, 2^-22 = four float32 roundings of it, or twice
    # what the float32 CPU evaluation of the same rows shows, whichever is larger
    mag = np.abs(xs.astype(np.float64)) @ np.abs(O.unpack_format_c(want_pw, k).astype(np.float64)).T * scale + np.abs(bias)
    rel_gpu, rel_cpu = float(np.max(np.abs(got[rows] - want64) / mag)), float(np.max(np.abs(cpu32 - want64) / mag))
    print(f"4096^3 N(0,1) f16: max|gpu - f64| = {err_gpu:.3e} ({rel_gpu:.2e} of sum|terms|), max|cpu f32 - f64| = {err_cpu:.3e} ({rel_cpu:.2e})")
    assert rel_gpu <= max(2.0 * rel_cpu, 2.0 ** -22)


@pytest.mark.parametrize("k,form", [(1024, "256x256"), (768, "256x256 persistent"), (128, "256x256 persistent")])
def test_full_tile_epilogue_clips_after_rounding_like_before_it(ops, dev, c_oracle, k, form):
    """Tiles inside the output take a shorter epilogue on the 256 x 256 kernel: float16 results are clipped on the packed halves
    after the conversion, scale and bias come as vectors (from LDS in the persistent form).  With a clip that binds on most
    outputs, a clip beyond the float16 range, channels whose bias is +-inf or NaN and scales that overflow float32, every
    16-bit output must still be the one rounding of the C restatement's float32 value."""
    for m, n in ((4096, 4096), (16384, 4096), (25216, 3072), (32768, 4096)):     # the first shape the planner gives this form of the kernel
        route = ops.matmul_route(torch.float16, m, n, k)
        if route.startswith("gemm_mfma16_kernel " + form) and ("persistent" in route) == ("persistent" in form):
            break
    else:
        pytest.fail(f"no candidate shape runs as {form} at k = {k}: last route {route}")
    x, w, scale, bias = _problem(m + k, m, n, k)
    scale[5::64] *= 64.0
    scale[7] = 3e38                                        # acc * scale overflows to +-inf before the clip
    bias[3], bias[258], bias[n - 1] = np.inf, -np.inf, np.nan
    want_pw = O.pack_format_c(O.ternarize(w))
    pw = torch.from_numpy(want_pw).to(dev)
    st, bt = torch.from_numpy(scale).to(dev), torch.from_numpy(bias).to(dev)
    xh = torch.from_numpy(x).to(dev).half()
    for clip in (3.0, 70000.0, 1e30):
        with np.errstate(over="ignore", invalid="ignore"):
            want = c_oracle(x, want_pw, scale, bias, clip)
            want_nb = c_oracle(x, want_pw, scale, None, clip)
        assert np.isnan(want[:, n - 1]).all() and np.all(want[:, 3] == clip) and np.all(want[:, 258] == -clip)
        if clip == 3.0:
            assert (np.abs(want[:, 5::64]) == 3.0).mean() > 0.5    # the clip binds
        for odt in (torch.float16, torch.bfloat16):
            got = ops.ternary_matmul(xh, pw, st, bt, clip, out_dtype=odt).float().cpu().numpy()
            assert np.array_equal(got, torch.from_numpy(want).to(odt).float().numpy(), equal_nan=True), (clip, odt)
            got = ops.ternary_matmul(xh, pw, st, None, clip, out_dtype=odt).float().cpu().numpy()
            assert np.array_equal(got, torch.from_numpy(want_nb).to(odt).float().numpy(), equal_nan=True), (clip, odt, "no bias")
    # the activation variant keeps the float32 clip in front of the GELU
    with np.errstate(over="ignore", invalid="ignore"):
        lin = c_oracle(x[:512], want_pw, scale, bias, 3.0)
    got = ops.ternary_matmul(xh, pw, st, bt, 3.0, activation="gelu_tanh", out_dtype=torch.float16)[:512].float().cpu().numpy()
    want_act = torch.from_numpy(O.gelu_tanh(lin).astype(np.float32)).half().float().numpy()
    ok = np.isfinite(want_act)
    assert np.array_equal(np.isnan(got), np.isnan(want_act)) and np.max(np.abs(got[ok] - want_act[ok])) <= 2e-3


def test_persistent_kernel_with_ragged_edges(ops, dev, c_oracle):
    """The persistent 256 x 256 kernel on an output whose last row tile and last channel tile are partial (and whose row length
    keeps the 16-byte alignment of the short epilogue): inner tiles take the short epilogue with scale / bias staged through LDS,
    edge tiles the general one, in one launch; every output against the C restatement."""
    k = 320
    for m, n in ((16400, 4104), (20000, 3080), (25216 + 24, 3072 + 8)):
        route = ops.matmul_route(torch.float16, m, n, k)
        if "256x256 persistent" in route:
            break
    else:
        pytest.fail(f"no candidate shape runs on the persistent kernel: last route {route}")
    x, w, scale, bias = _problem(m + n, m, n, k)
    want_pw = O.pack_format_c(O.ternarize(w))
    pw = torch.from_numpy(want_pw).to(dev)
    st, bt = torch.from_numpy(scale).to(dev), torch.from_numpy(bias).to(dev)
    xh = torch.from_numpy(x).to(dev).half()
    want = c_oracle(x, want_pw, scale, bias, 2.5)
    for odt in (torch.float16, torch.bfloat16, torch.float32):
        got = ops.ternary_matmul(xh, pw, st, bt, 2.5, out_dtype=odt).float().cpu().numpy()
        assert np.array_equal(got, torch.from_numpy(want).to(odt).float().numpy()), odt
    want = c_oracle(x, want_pw, scale, None, 100.0)
    got = ops.ternary_matmul(xh, pw, st, None, 100.0).float().cpu().numpy()
    assert np.array_equal(got, torch.from_numpy(want).half().float().numpy())


def test_rows_beyond_whole_rounds_of_big_tiles_go_to_a_second_launch(ops, dev, c_oracle):
    """4500 x 4096 x 4096: 288 tiles of 256 x 256 on 256 compute units.  The library gives the 4096 rows that fill one round to
    the big tile and the other 404 rows to smaller tiles in a second launch; every output against the C restatement."""
    m, n, k = 4500, 4096, 4096
    x, w, scale, bias = _problem(4500, m, n, k)
    want_pw = O.pack_format_c(O.ternarize(w))
    pw = torch.from_numpy(want_pw).to(dev)
    want = c_oracle(x, want_pw, scale, bias, 100.0)
    st, bt = torch.from_numpy(scale).to(dev), torch.from_numpy(bias).to(dev)
    for xdt in (torch.float16, torch.bfloat16):
        got = ops.ternary_matmul(torch.from_numpy(x).to(dev).to(xdt), pw, st, bt, 100.0, out_dtype=torch.float32)
        assert np.array_equal(got.cpu().numpy(), want), xdt


@pytest.mark.parametrize("method", ["select", "matmul"])
def test_config1_ternary_mm_at_the_files_own_size(ops, dev, method):
    """matrix_mul_tests.py:8-9,16-18: 1024 x 1024 operands, half of the weights zeroed.  Grid activations, so the select-and-sum
    loop's float32 sums are exact and the result must equal the numpy restatement of the reference loop bit for bit."""
    rng = np.random.default_rng(1024)
    m_ = rng.integers(-64, 65, size=(1024, 1024)).astype(np.float32) / 64.0
    tm = rng.normal(size=(1024, 1024)).astype(np.float32)
    tm[rng.random(size=tm.shape) < 0.5] = 0.0
    want = O.ternary_mm(m_, tm)
    got = ops.ternary_mm(torch.from_numpy(m_).to(dev), torch.from_numpy(tm).to(dev), method)
    assert np.array_equal(got.cpu().numpy(), want)
    assert np.array_equal(want, m_ @ np.sign(tm).T)


PLANS = ["2:256:256:1", "1:128:256:1", "1:128:256:-1", "1:128:256:3", "1:128:128:1", "1:128:128:4", "1:64:128:1", "1:64:128:5", "0:64:128:3", "0:32:128:2", "0:128:128:2"]


@pytest.mark.parametrize("plan", PLANS)
def test_every_tile_plan_is_exact_when_pinned(plan):
    """The planner picks a tile and a split-K factor per shape, so a test shape exercises one of them.  Here every tile of the
    matrix-core kernels (256 x 256, 128 x 256 - also in its two-workgroups-per-CU form, splits = -1 -, 128 x 128, the 64- and
    32-row tiles) is pinned in turn, with and without split-K,
    on ragged shapes (edge tiles, odd step counts, float32 activations as three wrapped bf16 planes, the activation from a
    channel on, code 3) in a child process - the library reads TVL_FORCE_PLAN once - and compared with the CPU oracle."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    shapes = ["300,500,512", "129,257,64", "1000,130,1280", "520,2304,192"]
    if plan.startswith("1:128:256:1"):
        shapes.append("9000,1100,128")            # more tiles than compute units: the persistent form of the 128 x 256 tile
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "check_plan.py"), *shapes], env=dict(os.environ, TVL_FORCE_PLAN=plan),
                         capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    assert out.stdout.count(": ok") == 6 * len(shapes)


@pytest.mark.parametrize("splits", ["0", "1", "3"])
def test_shared_activation_rows_kernel_on_small_shapes(splits):
    """The few-row kernel for large layers (8 waves share the activations through LDS, k split into slabs) is chosen by layer size,
    so the ragged small shapes of the other tests never reach it.  Here a child process lifts the size rule
    (TVL_ROWS_SHARED_ANY=1) and pins the k split (TVL_ROWS_SHARED: 0 = the planner's own): 3 .. 64 rows over the three row-block
    variants, channel counts that are not multiples of 16 or 128, 1 .. 16 steps of 256, every dtype pair, bias, the activation
    from a channel on, over-wide packed rows with code 3 (a multiple of 16 bytes wider, so the bf16 cases keep the 16-byte row
    alignment the kernel needs), against the CPU oracle."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    shapes = ["3,16,256", "5,300,1024", "16,130,4096", "17,257,1280", "32,128,2048", "33,200,768", "64,500,2304"]
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "check_plan.py"), *shapes],
                         env=dict(os.environ, TVL_ROWS_SHARED_ANY="1", TVL_ROWS_SHARED=splits, CHECK_EXTRA="0,16,32"), capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    assert out.stdout.count(": ok") == 6 * len(shapes)


@pytest.mark.parametrize("m,n,k,xdt", [(16, 8192, 8192, torch.float16), (24, 4096, 8192, torch.bfloat16), (48, 8192, 4096, torch.float16),
                                       (9, 14336, 2048, torch.bfloat16)])
def test_few_rows_on_large_layers_at_size(ops, dev, c_oracle, m, n, k, xdt):
    """Decode batches on layers of 29 .. 67 million weights (where the shared-activation kernel is the library's own choice):
    grid activations, so the float32 sums are exact and the product must equal the C oracle bit for bit."""
    rng = np.random.default_rng(m * 7 + n)
    x = rng.integers(-64, 65, size=(m, k)).astype(np.float32) / 64.0
    pw = rng.integers(0, 171, size=(n, k // 4)).astype(np.uint8)
    scale = (rng.integers(1, 9, size=(n,)) / 8.0).astype(np.float32)
    bias = (rng.integers(-16, 17, size=(n,)) / 8.0).astype(np.float32)
    want = c_oracle(x, pw, scale, bias, 1e6)
    got = ops.ternary_matmul(torch.from_numpy(x).to(dev).to(xdt), torch.from_numpy(pw).to(dev), torch.from_numpy(scale).to(dev),
                             torch.from_numpy(bias).to(dev), 1e6, out_dtype=torch.float32)
    assert np.array_equal(got.cpu().numpy(), want)


def test_staged_rows_kernel_on_small_shapes():
    """The one-launch few-row kernel that stages all of x in LDS is chosen from 5 rows and a million weights on; a child process
    lifts those rules (TVL_ROWS_STAGED=2) so that ragged small shapes reach it: 3 .. 64 rows, every row-block variant, one and two
    steps per wave, one and two staging phases, one / two / four / eight channel tiles per workgroup, fewer steps than waves, channel counts that are not multiples of 16, every dtype pair (packed rows a multiple
    of 16 bytes wider, code 3), against the CPU oracle."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    shapes = ["3,50,512", "5,300,1024", "9,33,256", "16,130,4096", "16,100,768", "12,257,2304", "17,257,1280", "32,128,2048", "25,40,256",
              "32,128,4096", "20,100,2560", "16,130,8192", "7,300,5120", "16,20000,256", "8,40000,256", "20,5000,2048", "6,70000,256", "24,10000,2048",   # these five: 2 / 4 / 2 / 8 / 4 channel tiles per workgroup
              "33,96,512", "40,100,1024", "48,50,2048", "49,300,1280", "64,2048,2048", "57,9000,768", "64,33,256",
              "40,96,4096", "64,200,3072", "50,5000,2304"]   # four phases: 33 .. 64 rows up to k = 4096   # 33 .. 64 rows: three / four row blocks, the waves split 2 / 4 tiles
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "check_plan.py"), *shapes],
                         env=dict(os.environ, TVL_ROWS_STAGED="2", CHECK_EXTRA="0,16,32"), capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    assert out.stdout.count(": ok") == 6 * len(shapes)


def test_staged_rows_kernel_on_blocks_of_64_rows():
    """65 .. 512 rows on small layers run the staged few-row kernel on blocks of 64 rows (blockIdx.y); a child process lifts the size
    rule (TVL_ROWS_BLOCKS=2) so that ragged shapes reach it: a last block of 1 .. 64 rows, channel counts that are not multiples
    of 32, one and two staging phases, two and four channel tiles per workgroup, every dtype pair, against the CPU oracle."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    shapes = ["65,96,512", "100,300,1024", "128,130,2048", "129,257,1280", "200,2048,2048", "256,33,256", "300,500,768", "512,64,2048",
              "449,2000,1024", "256,4000,256"]
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "check_plan.py"), *shapes],
                         env=dict(os.environ, TVL_ROWS_BLOCKS="2", CHECK_EXTRA="0,16,32"), capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    assert out.stdout.count(": ok") == 6 * len(shapes)


def test_table_lookup_kernel_geometries_on_one_to_four_rows():
    """The one-to-four-row table-lookup kernel in each launch geometry: waves that own channels (short k, many channels), and
    blocks whose four waves split k over two (one row, up to 2048 channels) or four shared channels (k >= 4096, up to 6144
    channels), with ragged channel counts and k that ends inside a wave's range; a child process switches the kernel on
    wherever it can run (TVL_GEMV=2); every dtype pair against the CPU oracle."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    shapes = ["1,2048,8192", "1,1001,4096", "2,2050,4128", "4,5000,4096", "3,6144,8192", "1,6145,4096", "1,300,12288", "2,130,6176",
              "4,2048,2048", "1,17000,4096", "2,16390,8192"]
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "check_plan.py"), *shapes],
                         env=dict(os.environ, TVL_GEMV="2", CHECK_EXTRA="0,8,16"), capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    assert out.stdout.count(": ok") == 6 * len(shapes)
