"""Parity of the HIP operators (called through the C ABI) against the CPU oracle.  Needs an MI355X.

Bars: bit-exact for every integer / byte result (quantize, pack, unpack, zero_out); for the
matmul, bit-exact on dyadic-grid inputs (every partial sum is exact in float32, so the result
does not depend on summation order) and <= 1e-5 absolute against a float64 oracle for N(0,1)
inputs at the small sizes where float32 accumulation error stays below that.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import ternary_oracle as O

pytestmark = pytest.mark.gpu


def _np(t):
    return t.detach().float().cpu().numpy() if t.dtype in (torch.bfloat16, torch.float16) else t.detach().cpu().numpy()


def _grid(rng, shape):
    return rng.integers(-64, 65, size=shape).astype(np.float32) / 64.0


# ------------------------------------------------------------------------------ plumbing ops

def test_zero_out_kats(ops, dev, golden_dir):
    kats = json.load(open(os.path.join(golden_dir, "reference_kats.json")))["zero_out"]
    for case in kats:
        got = ops.zero_out(torch.tensor(case["input"], dtype=torch.int32, device=dev))
        assert got.cpu().tolist() == case["expected"]
    # list input like the reference's own test (zero_out_tests.py:9)
    assert ops.zero_out([[5, 4, 3, 2, 1], [10, 9, 8, 7, 6]]).cpu()[0].tolist() == [5, 0, 0, 0, 0]
    big = torch.arange(7, 7 + 100003, dtype=torch.int32, device=dev)
    z = ops.zero_out(big)
    assert int(z[0]) == 7 and int(z.abs().sum()) == 7
    assert ops.zero_out(torch.empty((0,), dtype=torch.int32, device=dev)).numel() == 0


def test_prevent_folding_is_identity(ops, dev):
    for n in (0, 1, 15, 16, 17, 4096 * 1024 + 3):
        c = torch.randint(0, 256, (n,), dtype=torch.uint8, device=dev)
        w = torch.randn(4, device=dev)
        assert torch.equal(ops.prevent_folding(c, w), c)
    c = torch.randint(0, 256, (33, 7), dtype=torch.uint8, device=dev)
    assert torch.equal(ops.prevent_folding(c[:, 1:], None), c[:, 1:])


# ------------------------------------------------------------------------------ pack / unpack

@pytest.mark.parametrize("n,k", [(1, 1), (1, 4), (3, 5), (8, 16), (7, 33), (64, 64), (5, 5031), (129, 1000), (16, 4096)])
def test_pack_fn_matches_oracle(ops, dev, n, k):
    rng = np.random.default_rng(n * 1000 + k)
    t = rng.integers(-1, 2, size=(n, k)).astype(np.int8)
    got = ops.pack_fn(torch.from_numpy(t).to(dev))
    assert got.shape == (n, (k + 3) // 4) and got.dtype == torch.uint8
    assert np.array_equal(got.cpu().numpy(), O.pack_format_c(t))
    # strided input rows (unit stride inside a row)
    wide = torch.zeros((n, k + 13), dtype=torch.int8, device=dev)
    wide[:, 5:5 + k] = torch.from_numpy(t).to(dev)
    assert np.array_equal(ops.pack_fn(wide[:, 5:5 + k]).cpu().numpy(), O.pack_format_c(t))
    # round trip (impl/tf/moondream/ternary_tests.py:17-25 property)
    back = ops.unpack_ternary(got, k, dtype=torch.int8)
    assert np.array_equal(back.cpu().numpy(), t)


def test_pack_examples_and_reference_buffer(ops, dev, golden_dir):
    ex = json.load(open(os.path.join(golden_dir, "pack_examples.json")))
    for case in ex["format_c"]:
        t = torch.tensor(case["values"], dtype=torch.int8, device=dev)
        assert ops.pack_fn(t).cpu().tolist() == case["bytes"]
    for fmt in ("a", "b"):
        for case in ex[f"format_{fmt}"]:
            t = torch.tensor(case["values"], dtype=torch.int8, device=dev)
            assert ops.pack_ternary_planar(t, fmt).cpu().tolist() == case["bytes"]
    # the reference allocates K//4 + (4 - K%4) bytes per row when K%4 != 0 (ternary.py:269-270):
    # packing into / unpacking from such an over-sized buffer must work
    k = 13
    t = torch.randint(-1, 2, (6, k), dtype=torch.int8, device=dev)
    buf = torch.full((6, ops.reference_row_bytes(k)), 0xAB, dtype=torch.uint8, device=dev)
    ops.pack_fn(t, out=buf)
    assert torch.equal(ops.unpack_ternary(buf, k, dtype=torch.int8), t)
    assert torch.all(buf[:, (k + 3) // 4:] == 0xAB)  # bytes beyond ceil(K/4) are untouched


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16, torch.int8])
def test_unpack_dtypes_and_invalid_code(ops, dev, dtype):
    pw = torch.arange(256, dtype=torch.uint8, device=dev).reshape(8, 32)  # every byte value, code 3 included
    for k in (128, 127, 125, 1):
        got = ops.unpack_ternary(pw, k, dtype=dtype)
        want = O.unpack_format_c(pw.cpu().numpy(), k)
        assert got.dtype == dtype and np.array_equal(_np(got).astype(np.float32), want)


@pytest.mark.parametrize("fmt", ["a", "b"])
def test_planar_formats(ops, dev, fmt, golden_dir):
    n = json.load(open(os.path.join(golden_dir, "reference_kats.json")))["roundtrip_length"]["length"]
    pack = O.pack_format_a if fmt == "a" else O.pack_format_b
    unpack = O.unpack_format_a if fmt == "a" else O.unpack_format_b
    for length in (n, 1, 4, 7, 1 << 16):
        rng = np.random.default_rng(length)
        t = rng.integers(-1, 2, size=(length,)).astype(np.int8)
        got = ops.pack_ternary_planar(torch.from_numpy(t).to(dev), fmt)
        assert np.array_equal(got.cpu().numpy(), pack(t))
        back = ops.unpack_ternary_planar(got, fmt)
        assert np.array_equal(back.cpu().numpy(), unpack(pack(t)))
        assert np.array_equal(back.cpu().numpy()[:length].astype(np.int8), t)
    allb = torch.arange(256, dtype=torch.uint8, device=dev)
    assert np.array_equal(ops.unpack_ternary_planar(allb, fmt).cpu().numpy(), unpack(allb.cpu().numpy()))
    # values outside {-1, 0, +1} pack like the reference's uint8 arithmetic, in the vector kernel (4096) and the scalar one (1000)
    for length in (4096, 1000):
        t = np.random.default_rng(length).integers(-128, 128, size=(length,)).astype(np.int8)
        assert np.array_equal(ops.pack_ternary_planar(torch.from_numpy(t).to(dev), fmt).cpu().numpy(), pack(t))


# ------------------------------------------------------------------------------ quantize + pack

def test_quantizers_on_golden_inputs(ops, dev, golden_dir):
    g = np.load(os.path.join(golden_dir, "quantizers.npz"))
    for key in ("x_grid", "x"):
        x = g[key]
        xt = torch.from_numpy(x).to(dev)
        for name, fn in (("ternary", O.ternarize), ("binary", O.binarize_sign), ("binary01", O.binarize_step)):
            got = ops.quantize(xt, name)
            assert np.array_equal(got.cpu().numpy(), fn(x)), (key, name)
            assert np.array_equal(ops.quantize(xt, name, dtype=torch.int8).cpu().numpy(), fn(x).astype(np.int8))
    # bit-equal to the reference's own output where its straight-through term is exact
    got = ops.quantize(torch.from_numpy(g["x_grid"]).to(dev), "ternary").cpu().numpy()
    assert np.array_equal(got, g["ternary_grid"])


@pytest.mark.parametrize("n,k", [(1, 16), (4, 1024), (3, 1040), (5, 2048 + 16), (2, 4096), (7, 5031), (9, 30), (33, 768)])
@pytest.mark.parametrize("quant", ["ternary", "binary", "binary01"])
def test_ternarize_pack_matches_oracle(ops, dev, n, k, quant):
    rng = np.random.default_rng(n * 7919 + k)
    w = rng.uniform(-1, 1, size=(n, k)).astype(np.float32)
    w.reshape(-1)[:6] = [0.5, -0.5, 0.49999997, -0.49999997, 0.0, -0.0][: min(6, w.size)]
    want = O.pack_format_c(O.QUANTIZERS[quant](w))
    got = ops.ternarize_pack(torch.from_numpy(w).to(dev), 0.5, quant)
    assert np.array_equal(got.cpu().numpy(), want)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_ternarize_pack_half_types(ops, dev, dtype):
    torch.manual_seed(5)
    for n, k in ((6, 2048), (5, 100)):
        w = (torch.rand(n, k, device=dev) * 2 - 1).to(dtype)
        w[0, :4] = torch.tensor([0.5, -0.5, 0.4998, -0.4998]).to(dtype)
        want = O.pack_format_c(O.ternarize(_np(w)))
        assert np.array_equal(ops.ternarize_pack(w).cpu().numpy(), want)


def test_ternarize_pack_threshold_and_strides(ops, dev):
    rng = np.random.default_rng(3)
    w = rng.normal(size=(8, 256)).astype(np.float32)
    for thr in (0.0, 0.25, 1.5):
        want = O.pack_format_c(O.ternarize(w, thr))
        assert np.array_equal(ops.ternarize_pack(torch.from_numpy(w).to(dev), thr).cpu().numpy(), want)
    big = torch.from_numpy(rng.normal(size=(8, 512)).astype(np.float32)).to(dev)
    view = big[:, 128:384]  # row stride 512, 16-byte aligned start: fast path with ld_w != k
    assert np.array_equal(ops.ternarize_pack(view).cpu().numpy(), O.pack_format_c(O.ternarize(view.cpu().numpy())))
    view = big[:, 3:259]    # misaligned start: generic path
    assert np.array_equal(ops.ternarize_pack(view).cpu().numpy(), O.pack_format_c(O.ternarize(view.cpu().numpy())))
    w = torch.tensor([[float("nan"), float("inf"), -float("inf"), 0.5]], device=dev)
    assert ops.ternarize_pack(w).cpu().tolist() == [[0b10001001]]  # nan -> 0, +inf -> +1, -inf -> -1, 0.5 -> +1


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_ternarize_pack_batch_matches_single_calls_and_oracle(ops, dev, dtype):
    """Many matrices in one launch (a checkpoint's layers into an arena): the bytes of the single-matrix calls and of the oracle.
    70 matrices cross the 64-per-launch limit; odd k, a misaligned view and an empty matrix take the per-matrix path; the
    destinations are row-padded views into one buffer whose pad bytes must stay untouched."""
    rng = np.random.default_rng(17)
    shapes = [(int(rng.integers(1, 70)), 16 * int(rng.integers(1, 200))) for _ in range(66)] + [(5, 1030), (3, 7), (0, 64), (4, 2048)]
    ws, outs, views, want = [], [], [], []
    arena = torch.full((sum(n * (-(-((k + 3) // 4) // 16) * 16) for n, k in shapes) + 64,), 0xEE, dtype=torch.uint8, device=dev)
    off = 0
    for i, (n, k) in enumerate(shapes):
        w = rng.uniform(-1, 1, size=(n, k)).astype(np.float32)
        if w.size >= 4:
            w.reshape(-1)[:4] = [0.5, -0.5, 0.4999, -0.4999]
        wt = torch.from_numpy(w).to(dev).to(dtype)
        if i == 7:                                       # a view that starts 1 element into a wider buffer: not 16-byte loadable
            wide = torch.zeros((n, k + 8), dtype=dtype, device=dev)
            wide[:, 1:1 + k] = wt
            wt = wide[:, 1:1 + k]
        ld = -(-((k + 3) // 4) // 16) * 16
        view = arena[off:off + n * ld].view(n, ld)
        off += n * ld
        ws.append(wt); views.append(view)
        want.append(O.pack_format_c(O.ternarize(wt.float().cpu().numpy())) if n else np.zeros((0, (k + 3) // 4), np.uint8))
    got = ops.ternarize_pack_batch(ws, 0.5, "ternary", outs=views)
    for (n, k), g, v, w_, wt in zip(shapes, got, views, want, ws):
        kb = (k + 3) // 4
        assert np.array_equal(v.cpu().numpy()[:, :kb], w_), (n, k)
        assert torch.all(v[:, kb:] == 0xEE)                                      # row padding untouched
        if n:
            assert torch.equal(v[:, :kb], ops.ternarize_pack(wt))               # same bytes as the single call
    assert torch.all(arena[off:] == 0xEE)
    fresh = ops.ternarize_pack_batch(ws[:3], 0.25, "binary")                     # allocating form, another quantizer
    for f, wt in zip(fresh, ws[:3]):
        assert np.array_equal(f.cpu().numpy(), O.pack_format_c(O.binarize_sign(wt.float().cpu().numpy())))
    assert ops.ternarize_pack_batch([]) == []
    with pytest.raises(ValueError):
        ops.ternarize_pack_batch([ws[0], ws[1].to(torch.float64 if dtype != torch.float32 else torch.float16)])
    with pytest.raises(ValueError):
        ops.ternarize_pack_batch(ws[:2], outs=views[:1])


def test_ternarize_pack_full_size(ops, dev):
    """4096 x 4096 (the benchmark's weight matrix): every byte against the oracle."""
    g = torch.Generator(device="cpu").manual_seed(1234)
    w = torch.rand(4096, 4096, generator=g) * 2 - 1
    w[0, :4] = torch.tensor([0.5, -0.5, 0.4999, -0.4999])
    got = ops.ternarize_pack(w.to(dev)).cpu().numpy()
    want = O.pack_format_c(O.ternarize(w.numpy()))
    assert got.shape == (4096, 1024) and np.array_equal(got, want)
    # round trip back to the ternary values
    back = ops.unpack_ternary(torch.from_numpy(got).to(dev), 4096, dtype=torch.int8).cpu().numpy()
    assert np.array_equal(back, O.ternarize(w.numpy()).astype(np.int8))


# ------------------------------------------------------------------------------ the fused matmul

def _case(rng, m, n, k, grid=True, bias=True):
    x = _grid(rng, (m, k)) if grid else rng.normal(size=(m, k)).astype(np.float32)
    t = rng.integers(-1, 2, size=(n, k)).astype(np.int8)
    pw = O.pack_format_c(t)
    scale = rng.uniform(0.5, 1.5, size=(n,)).astype(np.float32)
    b = rng.normal(size=(n,)).astype(np.float32) if bias else None
    return x, pw, scale, b


SHAPES = [(1, 1, 1), (1, 8, 4), (3, 5, 7), (17, 33, 30), (64, 64, 64), (65, 130, 100), (5, 48, 5031),  # generic kernel
          (1, 128, 64), (7, 100, 128), (32, 128, 256), (33, 64, 192), (64, 256, 512), (100, 300, 640),  # MFMA, small tiles
          (129, 129, 64), (200, 512, 1024), (745, 2048, 2048),
          (1, 128, 256), (1, 4096, 4096), (3, 200, 768), (16, 130, 512), (31, 6144, 2048), (32, 256, 1024), (8, 96, 3072),  # few rows:
          (2, 1000, 16384), (5, 129, 8192), (9, 64, 1280), (12, 520, 4352), (16, 2048, 8192), (1, 30, 32768),         # streaming kernels
          (1, 7, 32), (2, 9, 96), (3, 130, 2080), (4, 2050, 2048), (1, 16390, 4096), (4, 16384, 160), (2, 16400, 8192), (3, 16385, 10240)]                 # single-launch decode kernel


@pytest.mark.parametrize("m,n,k", SHAPES)
@pytest.mark.parametrize("xdt", [torch.float16, torch.bfloat16, torch.float32])
def test_matmul_exact_grid(ops, dev, m, n, k, xdt):
    rng = np.random.default_rng(m * 100003 + n * 1009 + k)
    x, pw, scale, b = _case(rng, m, n, k)
    want = O.ternary_matmul(x, pw, scale, b, 100.0)
    xt = torch.from_numpy(x).to(dev).to(xdt)  # k/64 with |k| <= 64 is exact in f16 and bf16
    got = ops.ternary_matmul(xt, torch.from_numpy(pw).to(dev), torch.from_numpy(scale).to(dev), torch.from_numpy(b).to(dev),
                             100.0, out_dtype=torch.float32)
    assert got.shape == (m, n) and got.dtype == torch.float32
    assert np.array_equal(got.cpu().numpy(), want)  # max|delta| = 0: every partial sum is exact
    # output in the activation dtype: one rounding of the same float32 value
    got_lp = ops.ternary_matmul(xt, torch.from_numpy(pw).to(dev), torch.from_numpy(scale).to(dev), torch.from_numpy(b).to(dev), 100.0)
    assert got_lp.dtype == xdt
    assert torch.equal(got_lp.cpu(), torch.from_numpy(want).to(xdt))


@pytest.mark.parametrize("m,n,k", [(300, 500, 1000), (1, 4096, 1000), (16, 700, 2080), (729, 1152, 4304), (40, 90, 1101)])
@pytest.mark.parametrize("xdt", [torch.float16, torch.bfloat16, torch.float32])
def test_matmul_k_not_multiple_of_64_padded_route(ops, dev, m, n, k, xdt):
    """Large shapes with an odd K are zero-padded into the workspace and run on the fast kernels: same bits as the oracle
    (pad weights are code 1 = 0), also with arbitrary bytes - code 3, junk beyond K - in over-wide packed rows."""
    rng = np.random.default_rng(m + n + k)
    x = _grid(rng, (m, k))
    pw = rng.integers(0, 256, size=(n, (k + 3) // 4 + 5)).astype(np.uint8)
    scale = rng.uniform(0.5, 1.5, size=(n,)).astype(np.float32)
    b = rng.normal(size=(n,)).astype(np.float32)
    want = O.ternary_matmul(x, pw, scale, b, 1e6, k=k)
    got = ops.ternary_matmul(torch.from_numpy(x).to(dev).to(xdt), torch.from_numpy(pw).to(dev), torch.from_numpy(scale).to(dev),
                             torch.from_numpy(b).to(dev), 1e6, out_dtype=torch.float32)
    assert np.array_equal(got.cpu().numpy(), want)


@pytest.mark.parametrize("m,n,k", [(9, 40, 52), (130, 257, 256), (300, 512, 512)])
def test_matmul_no_bias_clip_and_batch_dims(ops, dev, m, n, k):
    rng = np.random.default_rng(k)
    x, pw, scale, _ = _case(rng, m, n, k, bias=False)
    for clip in (100.0, 2.5, -2.5):  # the reference stores abs(max_val) (binary/modules.py:99)
        want = O.ternary_matmul(x, pw, scale, None, clip)
        got = ops.ternary_matmul(torch.from_numpy(x).to(dev).half(), torch.from_numpy(pw).to(dev), torch.from_numpy(scale).to(dev),
                                 None, clip, out_dtype=torch.float32)
        assert np.array_equal(got.cpu().numpy(), want)
        assert np.max(np.abs(want)) <= abs(clip)
    x3 = torch.from_numpy(x).to(dev).half().reshape(1, m, k).expand(2, m, k)
    got = ops.ternary_matmul(x3, torch.from_numpy(pw).to(dev), torch.from_numpy(scale).to(dev), None, 100.0, out_dtype=torch.float32)
    assert got.shape == (2, m, n)
    assert np.array_equal(got[1].cpu().numpy(), O.ternary_matmul(x, pw, scale, None, 100.0))


@pytest.mark.parametrize("k", [64, 60])
def test_matmul_invalid_code_and_padding(ops, dev, k):
    """Random bytes (code 3 included, as in the reference's random pw initialiser,
    impl/tf/moondream/ternary.py:272-276) decode as code-1; pad codes beyond K never contribute."""
    rng = np.random.default_rng(11)
    m, n = 40, 96
    x = _grid(rng, (m, k))
    pw = rng.integers(0, 256, size=(n, (k + 3) // 4 + 3)).astype(np.uint8)  # wider rows than needed
    scale = np.ones((n,), np.float32)
    want = O.ternary_matmul(x, pw[:, : (k + 3) // 4], scale, None, 1e4, k=k)
    pwt = torch.from_numpy(pw).to(dev)
    for xdt in (torch.float16, torch.float32):
        got = ops.ternary_matmul(torch.from_numpy(x).to(dev).to(xdt), pwt, torch.from_numpy(scale).to(dev), None, 1e4,
                                 out_dtype=torch.float32)
        assert np.array_equal(got.cpu().numpy(), want)


@pytest.mark.parametrize("m,k", [(1, 256), (4, 1024), (13, 8192), (2, 96)])
def test_matmul_few_rows_invalid_code_and_extremes(ops, dev, m, k):
    """The few-row kernel feeds the raw 2-bit codes to the matrix cores as f16 subnormals: every code value
    (3 included, decoding as +2) at every bit position, and activations at the ends of the f16 range."""
    rng = np.random.default_rng(k + m)
    n = 70
    pw = rng.integers(0, 256, size=(n, k // 4)).astype(np.uint8)
    scale = np.ones((n,), np.float32)
    x = _grid(rng, (m, k))
    x[0, :8] = [65504.0, -65504.0, 2.0 ** -24, -(2.0 ** -24), 2.0 ** -14, 1024.0, -0.5, 0.0]  # f16 max, min subnormal, min normal
    want = O.ternary_matmul(x.astype(np.float64), pw, scale, None, 1e30, acc_dtype=np.float64)
    got = ops.ternary_matmul(torch.from_numpy(x).to(dev).half(), torch.from_numpy(pw).to(dev), torch.from_numpy(scale).to(dev), None, 1e30,
                             out_dtype=torch.float32)
    # row 0 mixes 6.5e4 with 6e-8: float32 accumulation cannot be exact there, the other rows are
    assert np.array_equal(got.cpu().numpy()[1:], want[1:].astype(np.float32))
    assert np.max(np.abs(got.cpu().numpy()[0] - want[0]) / np.maximum(1.0, np.abs(want[0]))) <= 1e-2 * 65504 * 2.0 ** -10


@pytest.mark.parametrize("m,n,k", [(70, 96, 64), (300, 200, 1024), (17, 520, 4096)])
def test_matmul_f32_activations_matrix_core_path(ops, dev, m, n, k):
    """float32 activations run on the matrix cores as three exact bf16 planes (x = hi + mid + lo): full float32
    inputs - 24 significant bits, wide exponent range, signed zeros, subnormals - against the float64 oracle."""
    rng = np.random.default_rng(m + n + k)
    x = (rng.normal(size=(m, k)) * np.exp2(rng.integers(-12, 12, size=(m, k)))).astype(np.float32)
    x[0, :6] = [0.0, -0.0, 1e-40, -1e-40, 3.0e38, 1.17549435e-38]
    _, pw, scale, b = _case(rng, m, n, k)
    want = O.ternary_matmul(x.astype(np.float64), pw, scale, b, 1e38, acc_dtype=np.float64)
    cpu_f32 = O.ternary_matmul(x, pw, scale, b, 1e38, acc_dtype=np.float32)
    got = ops.ternary_matmul(torch.from_numpy(x).to(dev), torch.from_numpy(pw).to(dev), torch.from_numpy(scale).to(dev),
                             torch.from_numpy(b).to(dev), 1e38).cpu().numpy()
    mag = np.abs(x.astype(np.float64)) @ np.abs(O.unpack_format_c(pw, k).astype(np.float64)).T * scale + np.abs(b)  # sum of |terms|
    rel_gpu = np.max(np.abs(got[1:] - want[1:]) / mag[1:])
    rel_cpu = np.max(np.abs(cpu_f32[1:] - want[1:]) / mag[1:])
    print(f"K={k}: max relative-to-sum-of-magnitudes error gpu {rel_gpu:.3e}  cpu float32 {rel_cpu:.3e}")
    assert rel_gpu <= max(2.0 * rel_cpu, 2.0 ** -22)
    assert np.all(np.isfinite(got[0])) or not np.all(np.isfinite(want[0]))


def test_matmul_few_rows_without_workspace(ops, dev):
    """Through the raw C ABI with workspace = NULL: long k ranges need split-K scratch for the streaming kernel,
    so the call must fall back to a kernel that needs none - same bits on the exact grid."""
    import ctypes
    rng = np.random.default_rng(5)
    m, n, k = 3, 200, 16384 + 256
    x, pw, scale, b = _case(rng, m, n, k)
    want = O.ternary_matmul(x, pw, scale, b, 100.0)
    xt, pwt = torch.from_numpy(x).to(dev).half(), torch.from_numpy(pw).to(dev)
    st, bt = torch.from_numpy(scale).to(dev), torch.from_numpy(b).to(dev)
    y = torch.empty((m, n), dtype=torch.float32, device=dev)
    lib = ops._lib
    rc = lib.tvl_ternary_matmul(xt.data_ptr(), ops.F16, m, k, 0, pwt.data_ptr(), n, 0, st.data_ptr(), bt.data_ptr(), ctypes.c_float(100.0),
                                ops.ACT_NONE, y.data_ptr(), ops.F32, 0, None, 0, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, lib.tvl_last_error()
    torch.cuda.synchronize()
    assert np.array_equal(y.cpu().numpy(), want)


def test_matmul_strided_inputs(ops, dev):
    rng = np.random.default_rng(21)
    m, n, k = 70, 130, 128
    x, pw, scale, b = _case(rng, m, n, k)
    want = O.ternary_matmul(x, pw, scale, b, 100.0)
    xw = torch.zeros((m, k + 64), dtype=torch.float16, device=dev)
    xw[:, 8:8 + k] = torch.from_numpy(x).to(dev).half()
    pww = torch.zeros((n, k // 4 + 24), dtype=torch.uint8, device=dev)
    pww[:, 8:8 + k // 4] = torch.from_numpy(pw).to(dev)
    args = (torch.from_numpy(scale).to(dev), torch.from_numpy(b).to(dev), 100.0)
    got = ops.ternary_matmul(xw[:, 8:8 + k], pww[:, 8:8 + k // 4], *args, out_dtype=torch.float32)   # aligned views: MFMA path
    assert np.array_equal(got.cpu().numpy(), want)
    xw[:, 3:3 + k] = torch.from_numpy(x).to(dev).half()
    pww[:, 5:5 + k // 4] = torch.from_numpy(pw).to(dev)
    got = ops.ternary_matmul(xw[:, 3:3 + k], pww[:, 5:5 + k // 4], *args, out_dtype=torch.float32)   # misaligned: generic path
    assert np.array_equal(got.cpu().numpy(), want)


@pytest.mark.parametrize("m,n,k", [(16, 64, 96), (150, 256, 256), (64, 128, 4096), (1, 256, 4096), (7, 192, 2048), (16, 128, 8192)])
def test_matmul_randn_vs_f64(ops, dev, m, n, k):
    """N(0,1) activations: float32 accumulation against the float64 oracle.

    1e-5 absolute holds while the sums stay small (K <= 96 here).  Beyond that no float32
    summation order meets it - the reference's own two CPU paths (BLAS matmul and the
    select-and-sum loop) already differ by 3.4e-5 at K = 256 - so the gate there is "no worse than
    twice the error of a float32 CPU evaluation of the same product against the same oracle"."""
    rng = np.random.default_rng(1234)
    x, pw, scale, b = _case(rng, m, n, k, grid=False)
    want = O.ternary_matmul(x, pw, scale, b, 1e6)
    cpu_f32 = O.ternary_matmul(x, pw, scale, b, 1e6, acc_dtype=np.float32)  # float32 BLAS order
    err_cpu = float(np.max(np.abs(cpu_f32 - want)))
    tol = 1e-5 if k <= 96 else max(1e-5, 2.0 * err_cpu)
    dev_args = (torch.from_numpy(pw).to(dev), torch.from_numpy(scale).to(dev), torch.from_numpy(b).to(dev), 1e6)
    got = ops.ternary_matmul(torch.from_numpy(x).to(dev), *dev_args)
    err = float(np.max(np.abs(got.cpu().numpy() - want)))
    print(f"f32 x: K={k} max|gpu-f64|={err:.3e}  max|cpu_f32-f64|={err_cpu:.3e}")
    assert err <= tol
    # f16 / bf16 activations (matrix-core path when K % 64 == 0): the oracle sees the same rounded inputs
    for xdt in (torch.float16, torch.bfloat16):
        xr = torch.from_numpy(x).to(xdt)
        want_r = O.ternary_matmul(xr.float().numpy(), pw, scale, b, 1e6)
        got_r = ops.ternary_matmul(xr.to(dev), *dev_args, out_dtype=torch.float32)
        err_r = float(np.max(np.abs(got_r.cpu().numpy() - want_r)))
        print(f"{xdt}: K={k} max|gpu-f64|={err_r:.3e}")
        assert err_r <= tol


def test_matmul_gelu_epilogue(ops, dev):
    rng = np.random.default_rng(8)
    for m, n, k in ((20, 50, 36), (140, 256, 128)):
        x, pw, scale, b = _case(rng, m, n, k)
        scale *= 0.1
        want = O.ternary_matmul(x, pw, scale, b, 100.0, act=O.ACT_GELU_TANH)
        got = ops.ternary_matmul(torch.from_numpy(x).to(dev).half(), torch.from_numpy(pw).to(dev), torch.from_numpy(scale).to(dev),
                                 torch.from_numpy(b).to(dev), 100.0, activation="gelu_tanh", out_dtype=torch.float32)
        assert np.max(np.abs(got.cpu().numpy() - want)) <= 1e-5  # float32 tanh vs float64 oracle


@pytest.mark.parametrize("m,n,k,xdt", [
    (1, 192, 256, torch.float16),       # one-launch few-row kernel
    (3, 160, 128, torch.float32),       # the same with float32 activations
    (8, 384, 512, torch.float16),       # weight-stream kernel
    (12, 20480, 2048, torch.float16),   # weight-stream kernel with split-K + reduce
    (24, 256, 512, torch.bfloat16),     # skinny matrix-core kernel
    (300, 384, 256, torch.float16),     # small matrix-core tiles / split-K
    (1024, 1024, 256, torch.float16),   # 256 x 256 tile
    (520, 4608, 128, torch.bfloat16),   # persistent 256 x 256 tile loop
    (300, 384, 256, torch.float32),     # float32 activations on the matrix cores
    (21, 50, 38, torch.float16),        # generic kernel
])
def test_matmul_activation_from_channel(ops, dev, m, n, k, xdt):
    """Two layers on one input stored as one packed matrix (Wqkv | fc1 of the Phi block, layers.py:334-347): channels
    below act_begin stay linear, the rest get tanh-GELU.  Bit-identical to the two separate calls, and equal to the
    oracle; act_begin placed off every tile boundary."""
    rng = np.random.default_rng(77)
    x, pw, scale, b = _case(rng, m, n, k)
    scale *= 0.1
    a0 = n // 3 + 5
    xt = torch.from_numpy(x).to(dev).to(xdt)
    pwt, st, bt = torch.from_numpy(pw).to(dev), torch.from_numpy(scale).to(dev), torch.from_numpy(b).to(dev)
    got = ops.ternary_matmul(xt, pwt, st, bt, 100.0, activation="gelu_tanh", out_dtype=torch.float32, act_begin=a0)
    want = O.ternary_matmul(xt.float().cpu().numpy(), pw, scale, b, 100.0, act=O.ACT_GELU_TANH, act_begin=a0)
    assert np.max(np.abs(got.cpu().numpy() - want)) <= 1e-5
    lin = ops.ternary_matmul(xt, pwt, st, bt, 100.0, out_dtype=torch.float32)
    act = ops.ternary_matmul(xt, pwt, st, bt, 100.0, activation="gelu_tanh", out_dtype=torch.float32)
    assert torch.equal(got[:, :a0], lin[:, :a0]) and torch.equal(got[:, a0:], act[:, a0:])
    # act_begin == n: no channel carries the activation; == 0: all do
    assert torch.equal(ops.ternary_matmul(xt, pwt, st, bt, 100.0, activation="gelu_tanh", out_dtype=torch.float32, act_begin=n), lin)
    assert torch.equal(ops.ternary_matmul(xt, pwt, st, bt, 100.0, activation="gelu_tanh", out_dtype=torch.float32, act_begin=0), act)
    with pytest.raises(ValueError):
        ops.ternary_matmul(xt, pwt, st, bt, 100.0, activation="gelu_tanh", act_begin=n + 1)


def test_matmul_matches_reference_forward_golden(ops, dev, golden_dir):
    """Golden outputs of the reference's ScaledTernaryLinear / ScaledBinary*Linear forward."""
    g = np.load(os.path.join(golden_dir, "scaled_linear_forward.npz"))
    for quant in ("ternary", "binary", "binary01"):
        for bias in (True, False):
            tag = f"{quant}_{'bias' if bias else 'nobias'}"
            pw = ops.ternarize_pack(torch.from_numpy(g[f"{tag}_w"]).to(dev), 0.5, quant)
            b = torch.from_numpy(g[f"{tag}_b"]).to(dev) if bias else None
            for xdt in (torch.float32, torch.float16):
                got = ops.ternary_matmul(torch.from_numpy(g[f"{tag}_x"]).to(dev).to(xdt), pw, torch.from_numpy(g[f"{tag}_scale"]).to(dev),
                                         b, float(g[f"{tag}_maxval"]), out_dtype=torch.float32)
                assert np.array_equal(got.cpu().numpy(), g[f"{tag}_y"]), (tag, xdt)
    for bias in (True, False):  # uniform latent weights: the reference's weights are 1 ulp off +-1
        tag = f"ternary_u_{'bias' if bias else 'nobias'}"
        pw = ops.ternarize_pack(torch.from_numpy(g[f"{tag}_w"]).to(dev))
        b = torch.from_numpy(g[f"{tag}_b"]).to(dev) if bias else None
        got = ops.ternary_matmul(torch.from_numpy(g[f"{tag}_x"]).to(dev), pw, torch.from_numpy(g[f"{tag}_scale"]).to(dev), b,
                                 float(g[f"{tag}_maxval"]))
        assert np.max(np.abs(got.cpu().numpy() - g[f"{tag}_y"])) <= 1e-5
    pw = ops.ternarize_pack(torch.from_numpy(g["ternary_randn_w"]).to(dev))
    got = ops.ternary_matmul(torch.from_numpy(g["ternary_randn_x"]).to(dev), pw, torch.from_numpy(g["ternary_randn_scale"]).to(dev),
                             torch.from_numpy(g["ternary_randn_b"]).to(dev), float(g["ternary_randn_maxval"]))
    assert np.max(np.abs(got.cpu().numpy() - g["ternary_randn_y"])) <= 5e-5  # two float32 summation orders at K=256


def test_matmul_full_size_4096(ops, dev, c_oracle):
    """The benchmark configuration, 4096 x 4096 x 4096 f16 on the exact grid: all 16.7M outputs
    against the C restatement of the reference op (sums are exact on this grid, so any correct
    implementation gives the same bits), plus 64 rows against the numpy oracle."""
    g = torch.Generator(device="cpu").manual_seed(1234)
    m = n = k = 4096
    x = (torch.randint(-64, 65, (m, k), generator=g).float() / 64.0)
    w = torch.rand(n, k, generator=g) * 2 - 1
    scale = torch.rand(n, generator=g) + 0.5
    bias = torch.randn(n, generator=g)
    pw = ops.ternarize_pack(w.to(dev))
    y = ops.ternary_matmul(x.to(dev).half(), pw, scale.to(dev), bias.to(dev), 100.0, out_dtype=torch.float32)
    ref = torch.from_numpy(c_oracle(x.numpy(), pw.cpu().numpy(), scale.numpy(), bias.numpy(), 100.0)).to(dev)   # every output, C restatement
    assert torch.equal(y, ref)
    rows = torch.randint(0, m, (64,), generator=g)
    want = O.ternary_matmul(x[rows].numpy(), pw.cpu().numpy(), scale.numpy(), bias.numpy(), 100.0)
    assert np.array_equal(y[rows.to(dev)].cpu().numpy(), want)
    # linearity in x (size-independent property): f(2x) - bias-part == 2 (f(x) - bias-part) before clipping
    y2 = ops.ternary_matmul((2 * x).to(dev).half(), pw, scale.to(dev), None, 1e9, out_dtype=torch.float32)
    y1 = ops.ternary_matmul(x.to(dev).half(), pw, scale.to(dev), None, 1e9, out_dtype=torch.float32)
    assert torch.equal(y2, 2 * y1)


# ------------------------------------------------------------------------------ binarized activations (popcount)

@pytest.mark.parametrize("m,n,k", [(1, 1, 1), (1, 130, 31), (2, 64, 4096), (3, 200, 1000), (8, 96, 2048), (9, 70, 33), (64, 64, 64),
                                   (100, 300, 1029), (257, 129, 4096)])
def test_popcount_matmul_matches_oracle_and_matrix_core_path(ops, dev, m, n, k):
    rng = np.random.default_rng(m * 7 + n * 3 + k)
    x = rng.normal(size=(m, k)).astype(np.float32)
    x[0, 0] = -0.0
    t = rng.integers(-1, 2, size=(n, k)).astype(np.int8)
    pw = O.pack_format_c(t)
    scale = rng.uniform(0.5, 1.5, size=(n,)).astype(np.float32)
    bias = rng.normal(size=(n,)).astype(np.float32)
    xt, pwt = torch.from_numpy(x).to(dev), torch.from_numpy(pw).to(dev)
    st, bt = torch.from_numpy(scale).to(dev), torch.from_numpy(bias).to(dev)
    xb = ops.sign_pack_bits(xt)
    pos, neg = ops.ternary_bitplanes(pwt, k)
    assert np.array_equal(xb.cpu().numpy().view(np.uint32), O.sign_bits(x))
    wp, wn = O.ternary_bitplanes(pw, k)
    assert np.array_equal(pos.cpu().numpy().view(np.uint32), wp) and np.array_equal(neg.cpu().numpy().view(np.uint32), wn)
    want = O.popcount_matmul(O.sign_bits(x), wp, wn, scale, bias, 100.0)
    got = ops.popcount_matmul(xb, pos, neg, k, st, bt, 100.0)
    assert np.array_equal(got.cpu().numpy(), want)
    for method in ("popcount", "mfma", "auto"):
        y = ops.binarized_ternary_matmul(xt, pwt, st, bt, 100.0, method=method)
        assert np.array_equal(y.cpu().numpy(), want), method
    for xdt in (torch.float16, torch.bfloat16):   # the sign survives the cast
        assert torch.equal(ops.sign_pack_bits(xt.to(xdt)), xb)


def test_popcount_invalid_code_and_gelu(ops, dev):
    rng = np.random.default_rng(3)
    m, n, k = 5, 40, 256
    x = rng.normal(size=(m, k)).astype(np.float32)
    pw = rng.integers(0, 256, size=(n, k // 4 + 2)).astype(np.uint8)     # code 3 present, wider rows than needed
    scale = np.full((n,), 0.05, np.float32)
    wp, wn = O.ternary_bitplanes(pw, k)
    want = O.popcount_matmul(O.sign_bits(x), wp, wn, scale, None, 100.0, act=O.ACT_GELU_TANH)
    pos, neg = ops.ternary_bitplanes(torch.from_numpy(pw).to(dev), k)
    got = ops.popcount_matmul(ops.sign_pack_bits(torch.from_numpy(x).to(dev)), pos, neg, k, torch.from_numpy(scale).to(dev), None, 100.0,
                              activation="gelu_tanh")
    assert np.max(np.abs(got.cpu().numpy() - want)) <= 1e-5


@pytest.mark.parametrize("m,n,k", [(4400, 4200, 192), (5000, 4100, 64), (25216, 768, 128), (2600, 8192, 448)])
@pytest.mark.parametrize("xdt", [torch.float16, torch.bfloat16, torch.float32])
def test_matmul_more_tiles_than_cus(ops, dev, c_oracle, m, n, k, xdt):
    """More 256 x 256 output tiles than compute units: one workgroup walks several tiles as one stream of k steps
    (ragged edges, one to seven steps per tile).  Exact grid, so the C restatement of the reference op gives the
    same bits whatever its summation order; a row sample is checked against the numpy oracle too."""
    g = torch.Generator(device="cpu").manual_seed(m + n + k)
    x = torch.randint(-64, 65, (m, k), generator=g).float() / 64.0
    w = torch.rand(n, k, generator=g) * 2 - 1
    scale = torch.rand(n, generator=g) + 0.5
    bias = torch.randn(n, generator=g)
    pw = ops.ternarize_pack(w.to(dev))
    y = ops.ternary_matmul(x.to(dev).to(xdt), pw, scale.to(dev), bias.to(dev), 100.0, out_dtype=torch.float32)
    ref = torch.from_numpy(c_oracle(x.numpy(), pw.cpu().numpy(), scale.numpy(), bias.numpy(), 100.0)).to(dev)   # every output, C restatement
    assert torch.equal(y, ref)
    rows = torch.randint(0, m, (16,), generator=g)
    want = O.ternary_matmul(x[rows].numpy(), pw.cpu().numpy(), scale.numpy(), bias.numpy(), 100.0)
    assert np.array_equal(y[rows.to(dev)].cpu().numpy(), want)
    y16 = ops.ternary_matmul(x.to(dev).half(), pw, scale.to(dev), bias.to(dev), 100.0)   # 16-byte store path
    assert torch.equal(y16, ref.half())


@pytest.mark.parametrize("m,n,k", [(4096, 4096, 64), (4000, 4096, 128), (4096, 3900, 320), (4096, 4096, 576), (3840, 4096, 640), (4096, 3900, 704), (4096, 4096, 768)])
def test_matmul_one_tile_per_cu_short_k(ops, dev, c_oracle, m, n, k):
    """192 .. 256 tiles of 256 x 256, one per compute unit, with one to twelve k steps (every remainder modulo four from nine on, where the planner picks this tile): the four-tile LDS pipeline's
    prologue, its counted waits and its tail (fewer steps than tiles in flight)."""
    g = torch.Generator(device="cpu").manual_seed(m + n + k)
    x = torch.randint(-64, 65, (m, k), generator=g).float() / 64.0
    w = torch.rand(n, k, generator=g) * 2 - 1
    scale = torch.rand(n, generator=g) + 0.5
    bias = torch.randn(n, generator=g)
    pw = ops.ternarize_pack(w.to(dev))
    ref = torch.from_numpy(c_oracle(x.numpy(), pw.cpu().numpy(), scale.numpy(), bias.numpy(), 100.0)).to(dev)   # every output, C restatement
    for xdt in (torch.float16, torch.bfloat16):
        y = ops.ternary_matmul(x.to(dev).to(xdt), pw, scale.to(dev), bias.to(dev), 100.0, out_dtype=torch.float32)
        assert torch.equal(y, ref)
    rows = torch.randint(0, m, (16,), generator=g)
    want = O.ternary_matmul(x[rows].numpy(), pw.cpu().numpy(), scale.numpy(), bias.numpy(), 100.0)
    assert np.array_equal(ref[rows.to(dev)].cpu().numpy(), want)


@pytest.mark.parametrize("m,n,k", [(300, 512, 256), (2, 64, 64), (20, 100, 52), (4400, 4200, 64), (12, 96, 768), (16, 300, 8192), (3, 4096, 4096),
                                   (24, 256, 512), (40, 512, 1024), (200, 2048, 2048)])
def test_matmul_nan_and_inf_like_torch_clip(ops, dev, m, n, k):
    """NaN activations give NaN outputs (torch.clip keeps NaN, binary/modules.py:107), infinities clip to +-clip_val
    where the weight is +-1 and give NaN where it is 0 (0 * inf in the reference's dense product), rows without them
    are untouched - on every kernel (small tiles, big tile, persistent, generic, the few-row kernels with and without split-K)."""
    rng = np.random.default_rng(m + k)
    x, pw, scale, b = _case(rng, m, n, k)
    x[0, 3] = np.nan
    x[1 % m, 5] = np.inf
    xt = torch.from_numpy(x).to(dev).half()
    wt = torch.from_numpy(O.unpack_format_c(pw, k)).to(dev)
    ref = torch.clip((xt.float() @ wt.t()) * torch.from_numpy(scale).to(dev) + torch.from_numpy(b).to(dev), -100.0, 100.0)
    got = ops.ternary_matmul(xt, torch.from_numpy(pw).to(dev), torch.from_numpy(scale).to(dev), torch.from_numpy(b).to(dev), 100.0,
                             out_dtype=torch.float32)
    assert torch.equal(torch.isnan(got), torch.isnan(ref))
    assert bool(torch.isnan(got[0]).any())
    assert torch.equal(torch.nan_to_num(got, nan=7.0), torch.nan_to_num(ref, nan=7.0))


@pytest.mark.parametrize("method", ["select", "matmul", "auto"])
def test_ternary_mm(ops, dev, golden_dir, method):
    g = np.load(os.path.join(golden_dir, "ternary_mm.npz"))
    for size in (32, 256):
        got = ops.ternary_mm(torch.from_numpy(g[f"m{size}"]).to(dev), torch.from_numpy(g[f"tm{size}"]).to(dev), method)
        assert np.array_equal(got.cpu().numpy(), g[f"res{size}"])
    got = ops.ternary_mm(torch.from_numpy(g["m256r"]).to(dev), torch.from_numpy(g["tm256r"]).to(dev), method)
    assert np.max(np.abs(got.cpu().numpy() - g["res256r"])) <= 1e-4  # float32 order noise at K=256, |sum| ~ 16
    rng = np.random.default_rng(2)
    m, tm = _grid(rng, (37, 101)), rng.normal(size=(19, 101)).astype(np.float32)
    tm[3, :7] = [0.0, -0.0, 1e-45, -1e-45, np.inf, -np.inf, np.nan]     # zeros, denormals, infinities, NaN: sign only
    assert np.array_equal(ops.ternary_mm(torch.from_numpy(m).to(dev), torch.from_numpy(tm).to(dev), method).cpu().numpy(), O.ternary_mm(m, tm))
    with pytest.raises(ValueError):
        ops.ternary_mm(torch.from_numpy(m).to(dev), torch.from_numpy(tm).to(dev), "popcount")


# ------------------------------------------------------------------------------ scale fits

def test_fit_scales_match_reference_golden(ops, dev, golden_dir):
    g = np.load(os.path.join(golden_dir, "linear_to_quantized.npz"))
    w = torch.from_numpy(g["w"]).to(dev)
    for key in g.files:
        if not key.endswith("__scale"):
            continue
        quant, ns, sc = key.split("__")[:3]
        try:
            sc = float(sc)
        except ValueError:
            pass
        if quant == "ternary":
            latent, scale = ops.fit_ternary_scale(w, ns)
        else:
            latent, scale = ops.fit_binary_scale(w, quant, ns, sc)
        np.testing.assert_allclose(scale.cpu().numpy(), g[key], rtol=1e-5, err_msg=key)
        np.testing.assert_allclose(latent.cpu().numpy(), g[key.replace("__scale", "__latent")], rtol=1e-5, atol=1e-6, err_msg=key)


@pytest.mark.parametrize("n,k", [(1, 1), (3, 7), (64, 2048), (300, 1000), (9, 4096), (5, 8192), (3, 12288), (6, 1002)])
def test_fit_ternary_scale_vs_oracle(ops, dev, n, k):
    rng = np.random.default_rng(n + k)
    w = (rng.normal(size=(n, k)) * 0.05).astype(np.float32)
    for ns in ("independent", "uniform"):
        latent, scale = ops.fit_ternary_scale(torch.from_numpy(w).to(dev), ns)
        lo, so = O.fit_ternary_scale(w, ns)
        np.testing.assert_allclose(scale.cpu().numpy(), so, rtol=2e-5)
        np.testing.assert_allclose(latent.cpu().numpy(), lo, rtol=2e-5, atol=1e-6)
    for quant in ("binary", "binary01"):
        for ns, scaling in (("uniform", "none"), ("independent", "natural"), ("independent", 0.5)):
            latent, scale = ops.fit_binary_scale(torch.from_numpy(w).to(dev), quant, ns, scaling)
            lo, so = O.fit_binary_scale(w, quant, ns, scaling)
            np.testing.assert_allclose(scale.cpu().numpy(), so, rtol=2e-5, atol=1e-9)
            np.testing.assert_allclose(latent.cpu().numpy(), lo, rtol=4e-5, atol=1e-6)
    with pytest.raises(ValueError):
        ops.fit_binary_scale(torch.from_numpy(w).to(dev), "binary", "none")


# ------------------------------------------------------------------------------ error behaviour

def test_errors(ops, dev):
    x = torch.zeros((4, 64), dtype=torch.float16, device=dev)
    pw = torch.zeros((8, 16), dtype=torch.uint8, device=dev)
    s = torch.ones(8, device=dev)
    with pytest.raises(ValueError):
        ops.ternary_matmul(x.cpu(), pw, s, None, 1.0)                # CPU tensor: no CPU path
    with pytest.raises(ValueError):
        ops.ternary_matmul(x, pw[:, :15], s, None, 1.0)              # packed rows too short for K
    with pytest.raises(ValueError):
        ops.ternary_matmul(x, pw, s[:7], None, 1.0)                  # scale length
    with pytest.raises(ValueError):
        ops.ternary_matmul(x.to(torch.int32), pw, s, None, 1.0)      # dtype
    with pytest.raises(ValueError):
        ops.ternary_matmul(x, pw, s, None, 1.0, activation="relu")   # unknown option string
    with pytest.raises(ValueError):
        ops.pack_fn(torch.zeros((2, 2), dtype=torch.float32, device=dev))
    with pytest.raises(ValueError):
        ops.unpack_ternary(pw, 65)
    with pytest.raises(ValueError):
        ops.ternarize_pack(torch.zeros((2, 8), device=dev), quantization="quaternary")
    # empty shapes are fine
    assert ops.ternary_matmul(x[:0], pw, s, None, 1.0).shape == (0, 8)
    assert ops.pack_fn(torch.zeros((0, 8), dtype=torch.int8, device=dev)).shape == (0, 2)
    # the process survives all of the above and still computes
    assert float(ops.ternary_matmul(x + 1, pw + 0xAA, s, None, 1e4).sum()) == 4 * 8 * 64


# ------------------------------------------------------------------------------ the product with the block's closing sum

@pytest.mark.parametrize("m,n,k", [(1, 2048, 2048), (2, 2048, 8192), (16, 2048, 2048), (24, 2048, 8192), (64, 512, 4096), (256, 2048, 2048),
                                   (256, 2048, 8192), (300, 1024, 4096), (745, 2048, 2048), (70, 96, 250)])
@pytest.mark.parametrize("dtype", ["f32", "f16", "bf16"])
def test_matmul_residual_matches_oracle_and_the_separate_additions(ops, dev, m, n, k, dtype):
    """tvl_ternary_matmul_residual on every family of kernels (one and two rows, few rows in one launch and with a k split, 128-row
    tiles with and without a k split, an odd k): equal to the oracle twin, and to the product followed by torch's additions."""
    tdt = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[dtype]
    rng = np.random.default_rng(m + n + k)
    x = rng.integers(-64, 65, size=(m, k)).astype(np.float32) / 64.0
    pw = rng.integers(0, 171, size=(n, (k + 3) // 4)).astype(np.uint8)
    if k % 4:
        pw[:, -1] = (pw[:, -1] & ((1 << (2 * (k % 4))) - 1)) | (0x55 & ~((1 << (2 * (k % 4))) - 1))   # pad codes 1
    scale = (rng.integers(4, 13, size=n) / 8.0).astype(np.float32)
    bias = (rng.integers(-16, 17, size=n) / 8.0).astype(np.float32)
    a1 = O.round_to(rng.normal(size=(m, n)).astype(np.float32) * 3, dtype)
    a2 = O.round_to(rng.normal(size=(m, n)).astype(np.float32) * 3, dtype)
    t = lambda a, d=None: torch.from_numpy(a).to(dev) if d is None else torch.from_numpy(a).to(dev).to(d)
    xt, pwt, st, bt, a1t, a2t = t(x, tdt), t(pw), t(scale), t(bias), t(a1, tdt), t(a2, tdt)
    for adds in ((a1, a2), (a1, None)):
        got = ops.ternary_matmul(xt, pwt, st, bt, 100.0, residual=a1t, residual2=a2t if adds[1] is not None else None)
        want = O.ternary_matmul_residual(x, pw, scale, bias, 100.0, adds[0], adds[1], out=dtype, k=k)
        assert got.dtype == tdt
        assert np.array_equal(got.float().cpu().numpy(), want), (m, n, k, dtype)
        plain = ops.ternary_matmul(xt, pwt, st, bt, 100.0) + a1t
        if adds[1] is not None:
            plain = plain + a2t
        assert torch.equal(got, plain)
    # the activation comes before the sums
    got = ops.ternary_matmul(xt, pwt, st, bt, 100.0, activation="gelu_tanh", act_begin=n // 2, residual=a1t)
    assert torch.equal(got, ops.ternary_matmul(xt, pwt, st, bt, 100.0, activation="gelu_tanh", act_begin=n // 2) + a1t)


def test_matmul_residual_argument_errors(ops, dev):
    x = torch.zeros(4, 64, device=dev).half()
    pw = torch.zeros(8, 16, dtype=torch.uint8, device=dev)
    s = torch.ones(8, device=dev)
    with pytest.raises(ValueError):
        ops.ternary_matmul(x, pw, s, None, 1.0, residual=torch.zeros(4, 8, device=dev))             # wrong dtype
    with pytest.raises(ValueError):
        ops.ternary_matmul(x, pw, s, None, 1.0, residual=torch.zeros(4, 9, device=dev).half())      # wrong shape
    with pytest.raises(ValueError):
        ops.ternary_matmul(x, pw, s, None, 1.0, residual=torch.zeros(4, 8).half())                  # not on the GPU
    buf = torch.zeros(4, 8, device=dev).half()
    with pytest.raises(ValueError, match="output buffer"):
        ops.ternary_matmul(x, pw, s, None, 1.0, residual=buf, out=buf)                               # the output as its own addend


def test_matmul_route_names_the_kernels(ops):
    r = ops.matmul_route
    assert "256x256" in r(torch.float16, 4096, 4096, 4096)
    assert "gemv_kernel" in r(torch.float16, 1, 4096, 4096)
    assert "gemm_rows" in r(torch.float16, 16, 4096, 4096)
    assert "split-k" in r(torch.float16, 256, 2048, 8192) and "splitk_reduce_kernel" in r(torch.float16, 256, 2048, 8192)
    assert "gemm_rows_staged_kernel" in r(torch.float16, 256, 2048, 2048)          # blocks of 64 rows on a small layer
    assert "split_f32_kernel" in r(torch.float32, 4096, 4096, 4096)
    assert "pad_rows_kernel" in r(torch.float16, 729, 1152, 4304)
    assert r(torch.float16, 3, 7, 5) == "gemm_generic_kernel"
