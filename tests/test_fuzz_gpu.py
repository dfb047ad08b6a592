"""Randomised parity sweep of the fused matmul against the CPU oracle (needs an MI355X).

Shapes are drawn around the dispatch boundaries of the kernels (row counts 1 / 4 / 16 / 32 / 48 / 64, K multiples of 32 / 64 /
256 and odd K, channel counts around tile edges), with random choice of activation dtype, output dtype, bias, fused
activation, the channel the activation starts at, row strides and over-wide packed rows including code 3.  Inputs are
on the dyadic grid k/64, so every partial sum is exact in float32 and the result must equal the oracle bit for bit
whatever kernel and summation order the library picks; with the GELU the comparison is at 1e-5."""
import os

import numpy as np
import pytest
import torch

from oracle import ternary_oracle as O

pytestmark = pytest.mark.gpu

M_CHOICES = [1, 2, 3, 4, 5, 8, 15, 16, 17, 31, 32, 33, 48, 49, 63, 64, 100, 129, 255, 256, 257, 300, 513]
K_CHOICES = [4, 28, 32, 36, 60, 64, 96, 100, 128, 192, 256, 320, 512, 768, 1000, 1024, 1280, 2048, 2080, 4096]
N_CHOICES = [1, 2, 7, 16, 31, 32, 33, 64, 100, 127, 128, 129, 200, 256, 257, 384, 511, 512, 640, 1000, 2050]


def _draw(rng):
    while True:
        m, n, k = (int(rng.choice(c)) for c in (M_CHOICES, N_CHOICES, K_CHOICES))
        if m * n * k <= 3e8:
            return m, n, k


@pytest.mark.parametrize("seed", range(int(os.environ.get("TVL_FUZZ_SEEDS", "6"))))   # more seeds for a longer soak
def test_matmul_random_configurations(ops, dev, seed):
    rng = np.random.default_rng(9000 + seed)
    for case in range(40):
        m, n, k = _draw(rng)
        xdt = [torch.float16, torch.bfloat16, torch.float32][int(rng.integers(0, 3))]
        ydt = [torch.float32, xdt][int(rng.integers(0, 2))]
        has_bias, gelu = bool(rng.integers(0, 2)), bool(rng.integers(0, 4) == 0)
        act_begin = int(rng.integers(0, n + 1)) if gelu and rng.integers(0, 2) else 0
        code3 = bool(rng.integers(0, 5) == 0)
        extra = int(rng.choice([0, 0, 1, 4, 8]))                 # packed rows wider than needed
        x = rng.integers(-64, 65, size=(m, k)).astype(np.float32) / 64.0
        kb = (k + 3) // 4
        if code3:
            pw = rng.integers(0, 256, size=(n, kb + extra)).astype(np.uint8)
        else:
            pw = np.zeros((n, kb + extra), dtype=np.uint8)
            pw[:, :kb] = O.pack_format_c(rng.integers(-1, 2, size=(n, k)).astype(np.int8))
            pw[:, kb:] = rng.integers(0, 256, size=(n, extra)).astype(np.uint8)
        scale = (rng.integers(1, 9, size=(n,)) / 8.0).astype(np.float32) * (0.125 if gelu else 1.0)
        bias = (rng.integers(-16, 17, size=(n,)) / 8.0).astype(np.float32) if has_bias else None
        clip = float(rng.choice([100.0, 4.0, 1e6]))
        want = O.ternary_matmul(x, pw, scale, bias, clip, k=k, act=O.ACT_GELU_TANH if gelu else O.ACT_NONE, act_begin=act_begin)
        xt = torch.from_numpy(x).to(dev).to(xdt)
        if rng.integers(0, 3) == 0:                              # rows with a stride (multiple of 8 elements keeps fast paths eligible)
            pad = int(rng.choice([8, 24, 3]))
            xt = torch.cat([xt, torch.full((m, pad), 7.0, device=dev, dtype=xdt)], dim=1)[:, :k]
        got = ops.ternary_matmul(xt, torch.from_numpy(pw).to(dev), torch.from_numpy(scale).to(dev),
                                 torch.from_numpy(bias).to(dev) if has_bias else None, clip,
                                 activation="gelu_tanh" if gelu else None, out_dtype=ydt, act_begin=act_begin)
        tag = f"seed {seed} case {case}: m={m} n={n} k={k} x={xdt} y={ydt} bias={has_bias} gelu={gelu}@{act_begin} code3={code3} extra={extra} clip={clip}"
        assert got.shape == (m, n) and got.dtype == ydt, tag
        g = got.float().cpu().numpy()
        w = torch.from_numpy(want).to(ydt).float().numpy()       # the oracle's float32 result rounded once to the output type
        if gelu:
            if ydt == torch.float32:
                assert np.max(np.abs(g - w)) <= 1e-5, tag
            else:   # 16-bit outputs: the float32 values agree to 1e-5, so the rounded ones differ by at most one unit in the last place
                ulp = np.exp2(np.floor(np.log2(np.maximum(np.abs(w), 2.0 ** -14))) - (10 if ydt == torch.float16 else 7))
                assert np.all(np.abs(g - w) <= ulp + 1e-5), tag
        else:
            assert np.array_equal(g, w), tag


@pytest.mark.parametrize("seed", range(8))
def test_matmul_random_large_shapes(ops, dev, c_oracle, seed):
    """Random large shapes (around 200 .. 400 tiles of 256 x 256: one tile per compute unit and the persistent tile loop,
    ragged edges, 9 .. 24 k steps) against the C restatement of the reference op on the same exact-grid inputs (every output), plus
    a row sample against the numpy oracle."""
    g = torch.Generator(device="cpu").manual_seed(777 + seed)
    m = int(torch.randint(3000, 5200, (1,), generator=g))
    n = int(torch.randint(3000, 5200, (1,), generator=g))
    k = 64 * int(torch.randint(9, 25, (1,), generator=g))
    xdt = [torch.float16, torch.bfloat16][seed % 2]
    x = torch.randint(-64, 65, (m, k), generator=g).float() / 64.0
    w = torch.rand(n, k, generator=g) * 2 - 1
    scale = (torch.randint(1, 9, (n,), generator=g) / 8.0).float()
    bias = (torch.randint(-16, 17, (n,), generator=g) / 8.0).float()
    pw = ops.ternarize_pack(w.to(dev))
    y = ops.ternary_matmul(x.to(dev).to(xdt), pw, scale.to(dev), bias.to(dev), 100.0, out_dtype=torch.float32)
    ref = torch.from_numpy(c_oracle(x.numpy(), pw.cpu().numpy(), scale.numpy(), bias.numpy(), 100.0)).to(dev)
    assert torch.equal(y, ref), f"m={m} n={n} k={k} {xdt}"
    rows = torch.randint(0, m, (8,), generator=g)
    want = O.ternary_matmul(x[rows].numpy(), pw.cpu().numpy(), scale.numpy(), bias.numpy(), 100.0)
    assert np.array_equal(y[rows.to(dev)].cpu().numpy(), want)
    # 16-bit outputs with a clip that binds (inner tiles: the short epilogue, clip after the rounding for float16; edge tiles: the
    # general one), with and without bias: one rounding of the reference's float32 value
    del y, ref
    for b in (bias, None):
        y16 = ops.ternary_matmul(x.to(dev).to(xdt), pw, scale.to(dev), None if b is None else b.to(dev), 2.0, out_dtype=xdt)
        ref16 = torch.from_numpy(c_oracle(x.numpy(), pw.cpu().numpy(), scale.numpy(), None if b is None else b.numpy(), 2.0)).to(xdt)
        assert torch.equal(y16.cpu(), ref16), f"m={m} n={n} k={k} {xdt} 16-bit out, bias={b is not None}"
