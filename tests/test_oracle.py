"""The CPU oracle against the golden vectors generated from the reference's torch code,
the reference's literal KATs and its round-trip property.  No GPU."""
import json
import os

import numpy as np
import pytest

from oracle import ternary_oracle as O


def _npz(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def test_quantizers_match_reference(golden_dir):
    g = _npz(golden_dir, "quantizers.npz")
    # on the grid k/64 the reference's straight-through term (q + x - x) is exact: bit equality
    xg = g["x_grid"]
    assert np.array_equal(O.ternarize(xg), g["ternary_grid"])
    assert np.array_equal(O.binarize_sign(xg), g["sign_grid"])
    assert np.array_equal(O.binarize_step(xg), g["step_grid"])
    # elsewhere (q + x) rounds in float32: the reference is within one ulp of q and rounds to q
    x = g["x"]
    fin = np.isfinite(x)
    for mine, ref in ((O.ternarize(x), g["ternary"]), (O.binarize_sign(x), g["sign"]), (O.binarize_step(x), g["step"])):
        assert np.array_equal(mine[fin], np.rint(ref[fin]))
        assert np.max(np.abs(mine[fin] - ref[fin])) <= 1.2e-7
        assert np.all(np.isnan(ref[~fin]))  # q + inf - inf in the reference
    assert O.ternarize(x[~fin]).tolist() == [1.0, -1.0]


def test_threshold_is_inclusive():
    x = np.array([-0.5, 0.5, 0.4999, -0.4999, np.nan], dtype=np.float32)
    assert O.ternarize(x).tolist() == [-1.0, 1.0, 0.0, 0.0, 0.0]


@pytest.mark.parametrize("quant", ["ternary", "binary", "binary01"])
@pytest.mark.parametrize("bias", [True, False])
def test_scaled_linear_forward_exact_grid(golden_dir, quant, bias):
    g = _npz(golden_dir, "scaled_linear_forward.npz")
    tag = f"{quant}_{'bias' if bias else 'nobias'}"
    y = O.scaled_linear_forward(g[f"{tag}_x"], g[f"{tag}_w"], g[f"{tag}_scale"], g[f"{tag}_b"] if bias else None,
                                float(g[f"{tag}_maxval"]), quant)
    # dyadic-grid activations and latent weights: every partial sum is exact and the reference
    # multiplies by exactly {-1,0,+1}, so the oracle must be bit-equal
    assert np.array_equal(y, g[f"{tag}_y"])


@pytest.mark.parametrize("bias", [True, False])
def test_scaled_linear_forward_uniform_latent(golden_dir, bias):
    """Uniform random latent weights: the reference's straight-through term leaves its weights
    up to one ulp off +-1, so its output differs from exact-ternary arithmetic by ~1e-7 * sum|x|."""
    g = _npz(golden_dir, "scaled_linear_forward.npz")
    tag = f"ternary_u_{'bias' if bias else 'nobias'}"
    y = O.scaled_linear_forward(g[f"{tag}_x"], g[f"{tag}_w"], g[f"{tag}_scale"], g[f"{tag}_b"] if bias else None,
                                float(g[f"{tag}_maxval"]), "ternary")
    assert np.max(np.abs(y - g[f"{tag}_y"])) <= 1e-5


def test_scaled_linear_forward_randn(golden_dir):
    g = _npz(golden_dir, "scaled_linear_forward.npz")
    y = O.scaled_linear_forward(g["ternary_randn_x"], g["ternary_randn_w"], g["ternary_randn_scale"],
                                g["ternary_randn_b"], float(g["ternary_randn_maxval"]), "ternary")
    # float64 contraction vs the reference's float32 BLAS: summation-order noise only
    assert np.max(np.abs(y - g["ternary_randn_y"])) < 5e-5


def test_smooth_binary_forward(golden_dir):
    """SmoothlyBinarizedLinear (binary/modules.py:139-146), evaluation and training branches."""
    g = _npz(golden_dir, "smooth_binary.npz")
    for tag in ("bias", "nobias"):      # grid inputs, grid bias: exact
        b = g[f"{tag}_b"] if tag == "bias" else None
        assert np.array_equal(O.smooth_binary_forward(g[f"{tag}_x"], g[f"{tag}_w"], b, float(g[f"{tag}_maxval"])), g[f"{tag}_y_eval"])
        assert np.array_equal(O.smooth_binary_forward(g[f"{tag}_x"], g[f"{tag}_w"], b, float(g[f"{tag}_maxval"]), training=True),
                              g[f"{tag}_y_train"])
        assert np.max(np.abs(g[f"{tag}_y_eval"])) == float(g[f"{tag}_maxval"])      # the clip is exercised
        # the evaluation branch as a packed product: sign codes, unit scale, bias inside the clip
        pw = O.pack_format_c(O.binarize_sign(g[f"{tag}_w"]))
        n = pw.shape[0]
        assert np.array_equal(O.ternary_matmul(g[f"{tag}_x"], pw, np.ones(n, np.float32), b, float(g[f"{tag}_maxval"])), g[f"{tag}_y_eval"])
    assert np.max(np.abs(O.smooth_binary_forward(g["r_x"], g["r_w"], g["r_b"], float(g["r_maxval"])) - g["r_y_eval"])) <= 1e-5
    assert np.max(np.abs(O.smooth_binary_forward(g["r_x"], g["r_w"], g["r_b"], float(g["r_maxval"]), training=True) - g["r_y_train"])) <= 1e-5


def test_saturating_quantizers_have_the_plain_forward_value(golden_dir):
    """sign_ste_sat / step_ste_sat / SignSTESat (binary/modules.py:29-51) differ from sign_ste / step_ste only in the
    gradient; where the straight-through term is exact (|x| <= 1 on the grid) the forward values are those of the plain
    quantizers.  Beyond +-1 the reference adds clip(x) - clip(x).detach() = 0 as well."""
    g = _npz(golden_dir, "quantizers.npz")
    xg = g["x_grid"]
    assert np.array_equal(O.binarize_sign(xg), g["sign_sat_grid"])
    assert np.array_equal(O.binarize_sign(xg), g["sign_sat_module_grid"])
    assert np.array_equal(O.binarize_step(xg), g["step_sat_grid"])


def test_packed_matmul_equals_dense_forward(golden_dir):
    g = _npz(golden_dir, "scaled_linear_forward.npz")
    for tag, quant in (("ternary_bias", "ternary"), ("binary_bias", "binary"), ("binary01_nobias", "binary01")):
        q = O.QUANTIZERS[quant](g[f"{tag}_w"])
        pw = O.pack_format_c(q)
        b = g[f"{tag}_b"] if f"{tag}_b" in g.files else None
        y = O.ternary_matmul(g[f"{tag}_x"], pw, g[f"{tag}_scale"], b, float(g[f"{tag}_maxval"]))
        assert np.array_equal(y, g[f"{tag}_y"])


def test_linear_to_quantized(golden_dir):
    g = _npz(golden_dir, "linear_to_quantized.npz")
    w = g["w"]
    for key in g.files:
        if not key.endswith("__scale"):
            continue
        quant, ns, sc = key.split("__")[:3]
        try:
            sc = float(sc)
        except ValueError:
            pass
        if quant == "ternary":
            latent, scale = O.fit_ternary_scale(w, ns)
        else:
            latent, scale = O.fit_binary_scale(w, quant, ns, sc)
        np.testing.assert_allclose(scale, g[key], rtol=2e-6, atol=0, err_msg=key)
        np.testing.assert_allclose(latent, g[key.replace("__scale", "__latent")], rtol=3e-6, atol=1e-7, err_msg=key)
    # smoothBinary copies the weights and the bias, there is nothing to fit (binary/modules.py:267-274)
    assert np.array_equal(g["smoothBinary__uniform__none__latent"], w) and np.array_equal(g["smoothBinary__uniform__none__bias"], g["b"])


def test_ternary_scale_unknown_neuron_scale_is_per_row():
    rng = np.random.default_rng(0)
    w = rng.normal(size=(5, 33)).astype(np.float32)
    a = O.fit_ternary_scale(w, "independent")
    b = O.fit_ternary_scale(w, "none")
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    with pytest.raises(ValueError):
        O.fit_binary_scale(w, "binary", "none")


def test_ternary_mm(golden_dir):
    g = _npz(golden_dir, "ternary_mm.npz")
    for size in (32, 256):
        res = O.ternary_mm(g[f"m{size}"], g[f"tm{size}"])
        assert np.array_equal(res, g[f"res{size}"])  # exact grid
        assert np.array_equal(res, g[f"m{size}"] @ np.sign(g[f"tm{size}"]).T)
    res = O.ternary_mm(g["m256r"], g["tm256r"])
    assert np.max(np.abs(res - g["res256r"])) < 1e-4


def test_reference_kats(golden_dir):
    k = json.load(open(os.path.join(golden_dir, "reference_kats.json")))
    for case in k["zero_out"]:
        assert O.zero_out(np.array(case["input"])).tolist() == case["expected"]
    # zero_out_tests.py:10 compares with broadcasting against the first row only
    assert np.all(O.zero_out(np.array(k["zero_out"][0]["input"]))[0] == [5, 0, 0, 0, 0])


def test_pack_examples(golden_dir):
    ex = json.load(open(os.path.join(golden_dir, "pack_examples.json")))
    for case in ex["format_a"]:
        assert O.pack_format_a(np.array(case["values"])).tolist() == case["bytes"], case
    for case in ex["format_b"]:
        assert O.pack_format_b(np.array(case["values"])).tolist() == case["bytes"], case
    for case in ex["format_c"]:
        assert O.pack_format_c(np.array(case["values"])).tolist() == case["bytes"], case


@pytest.mark.parametrize("fmt", ["a", "b", "c"])
def test_roundtrip_5031(golden_dir, fmt):
    """impl/tf/moondream/ternary_tests.py:17-25: random ternary vector of length 5031."""
    n = json.load(open(os.path.join(golden_dir, "reference_kats.json")))["roundtrip_length"]["length"]
    rng = np.random.default_rng(1234)
    t = rng.integers(-1, 2, size=(n,)).astype(np.int32)
    if fmt == "a":
        back = O.unpack_format_a(O.pack_format_a(t))[:n]
    elif fmt == "b":
        back = O.unpack_format_b(O.pack_format_b(t))[:n]
    else:
        back = O.unpack_format_c(O.pack_format_c(t[None, :]), n)[0]
    assert np.array_equal(back.astype(np.int32), t)


def test_invalid_code_decodes_like_reference():
    b = np.array([0xFF], dtype=np.uint8)
    assert O.unpack_format_a(b).tolist() == [0, 0, 0, 0]        # pos - neg = 1 - 1
    assert O.unpack_format_b(b).tolist() == [2, 2, 2, 2]        # code - 1
    assert O.unpack_format_c(b[None, :], 4).tolist() == [[2, 2, 2, 2]]


def test_row_bytes():
    assert [O.packed_row_bytes(k) for k in (4, 5, 6, 7, 8)] == [1, 2, 2, 2, 2]
    assert [O.reference_row_bytes(k) for k in (4, 5, 6, 7, 8)] == [1, 4, 3, 2, 2]


def test_gelu_tanh_values():
    x = np.array([-3.0, -1.0, 0.0, 0.5, 1.0, 3.0])
    ref = np.array([-0.0036373920817729943, -0.15880800939172324, 0.0, 0.34571400982514394, 0.8411919906082768,
                    2.996362607918227])
    np.testing.assert_allclose(O.gelu_tanh(x), ref, rtol=1e-12)


def test_popcount_form_equals_dense_sign_matmul():
    """AND + popcount over bit planes == sign(x) @ t^T (integers), for ragged K and every code incl. the invalid 3."""
    rng = np.random.default_rng(77)
    for m, n, k in ((1, 1, 1), (3, 5, 31), (4, 7, 32), (5, 9, 33), (6, 10, 100), (2, 4, 1029)):
        x = rng.normal(size=(m, k)).astype(np.float32)
        x[0, 0] = 0.0
        x[-1, -1] = -0.0                                    # -0.0 >= 0 is true: +1, as in the reference's (x >= 0)
        t = rng.integers(-1, 2, size=(n, k)).astype(np.int8)
        pw = O.pack_format_c(t)
        scale = rng.uniform(0.5, 1.5, size=(n,)).astype(np.float32)
        bias = rng.normal(size=(n,)).astype(np.float32)
        xb = O.sign_bits(x)
        pos, neg = O.ternary_bitplanes(pw, k)
        assert xb.shape == (m, (k + 31) // 32) and pos.shape == (n, (k + 31) // 32)
        got = O.popcount_matmul(xb, pos, neg, scale, bias, 100.0)
        s = np.where(x >= 0, 1.0, -1.0).astype(np.float32)
        want = O.ternary_matmul(s, pw, scale, bias, 100.0, k=k)
        assert np.array_equal(got, want)
    pw = np.array([[0b11100100]], dtype=np.uint8)            # codes 0,1,2,3 -> -1, 0, +1, (+1 in the planes)
    pos, neg = O.ternary_bitplanes(pw, 4)
    assert pos[0, 0] == 0b1100 and neg[0, 0] == 0b0001
