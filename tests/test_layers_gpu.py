"""The host-side mirrors of the reference's layer classes, on the GPU, against the golden outputs of the
reference's own torch layers and against the oracle."""
import os

import numpy as np
import pytest
import torch
from torch import nn

from oracle import ternary_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tv():
    import ternary_vllm_amd
    return ternary_vllm_amd


def _load(layer, g, tag, dev, bias=True):
    with torch.no_grad():
        layer.weights.copy_(torch.from_numpy(g[f"{tag}_w"]))
        layer.scale.copy_(torch.from_numpy(g[f"{tag}_scale"]))
        if bias:
            layer.bias.copy_(torch.from_numpy(g[f"{tag}_b"]))
    return layer.to(dev)


@pytest.mark.parametrize("quant,cls", [("ternary", "ScaledTernaryLinear"), ("binary", "ScaledBinaryLinear"), ("binary01", "ScaledBinary01Linear")])
@pytest.mark.parametrize("bias", [True, False])
def test_scaled_linear_forward_matches_reference_golden(tv, dev, golden_dir, quant, cls, bias):
    g = np.load(os.path.join(golden_dir, "scaled_linear_forward.npz"))
    tag = f"{quant}_{'bias' if bias else 'nobias'}"
    n, k = g[f"{tag}_w"].shape
    layer = _load(getattr(tv, cls)(k, n, bias=bias, max_val=-float(g[f"{tag}_maxval"])), g, tag, dev, bias)  # abs(max_val)
    assert layer.maxval == float(g[f"{tag}_maxval"])
    assert sorted(n_ for n_, _ in layer.named_parameters()) == (["bias", "scale", "weights"] if bias else ["scale", "weights"])
    x = torch.from_numpy(g[f"{tag}_x"]).to(dev)
    with torch.no_grad():
        y = layer(x)
    assert np.array_equal(y.cpu().numpy(), g[f"{tag}_y"])       # bit-equal to the reference forward
    y3 = layer.eval()(x.reshape(1, *x.shape).half())             # leading dims and f16 activations
    assert y3.shape == (1, x.shape[0], n) and y3.dtype == torch.float16
    assert torch.equal(y3[0].cpu(), torch.from_numpy(g[f"{tag}_y"]).half())


def test_repack_follows_weight_updates(tv, dev):
    torch.manual_seed(0)
    layer = tv.ScaledTernaryLinear(64, 32).to(dev)
    x = (torch.randint(-64, 65, (8, 64)).float() / 64).to(dev)
    with torch.no_grad():
        y0 = layer(x)
        pw0 = layer.packed_weights()
        assert layer.packed_weights() is pw0            # cached while the weights are unchanged
        layer.weights.mul_(-1.0)                        # in-place update bumps the version
        y1 = layer(x)
        b = layer.bias
        assert torch.equal(((y1 - b) + (y0 - b)).abs() < 1e-5, torch.ones_like(y0, dtype=torch.bool))  # sign flip
    sd = layer.state_dict()
    assert set(sd) == {"weights", "bias", "scale"}      # checkpoint names of binary/modules.py:96-98


def test_data_writes_need_invalidate_and_loads_do_it(tv, dev):
    """A write through .data (the reference's checkpoint loader, convert.py:78) bumps no version counter: loading a state
    dict, train() / eval() and invalidate_packed() drop the packed copy; a training-mode forward with gradients re-packs."""
    torch.manual_seed(0)
    layer = tv.ScaledTernaryLinear(64, 32).to(dev).eval()
    x = (torch.randint(-64, 65, (8, 64)).float() / 64).to(dev)
    new_w = (torch.randint(-64, 65, (32, 64)).float() / 64).to(dev)
    with torch.no_grad():
        y0 = layer(x)
        want = tv.ScaledTernaryLinear(64, 32).to(dev).eval()
        want.load_state_dict({**layer.state_dict(), "weights": new_w})
        y_new = want(x)
        assert not torch.equal(y0, y_new)
        layer.weights.data[:] = new_w                   # invisible to the cache key
        layer.invalidate_packed()
        assert torch.equal(layer(x), y_new)
        layer.weights.data[:] = -new_w
        layer.load_state_dict(want.state_dict())        # loading drops the packed copy
        assert torch.equal(layer(x), y_new)
        layer.weights.data[:] = -new_w
        layer.train(); layer.eval()                     # so does switching modes
        y_neg = layer(x)
        assert not torch.equal(y_neg, y_new)
    layer.train()
    layer.weights.data[:] = new_w
    assert torch.equal(layer(x).detach(), y_new)        # training forward: re-packed on every call


@pytest.mark.parametrize("tag", ["bias", "nobias", "r"])
def test_smoothly_binarized_linear_matches_reference_golden(tv, dev, golden_dir, tag):
    """binary/modules.py:139-146: evaluation = sign-binarized weights, bias, clip as ONE packed matmul (bit-equal to the
    reference's output on the grid cases); training = latent weights clipped to [-1, 1], dense."""
    g = np.load(os.path.join(golden_dir, "smooth_binary.npz"))
    n, k = g[f"{tag}_w"].shape
    bias = tag != "nobias"
    layer = tv.SmoothlyBinarizedLinear(k, n, bias=bias, max_val=float(g[f"{tag}_maxval"]))
    assert sorted(n_ for n_, _ in layer.named_parameters()) == (["bias", "weights"] if bias else ["weights"])
    with torch.no_grad():
        layer.weights.copy_(torch.from_numpy(g[f"{tag}_w"]))
        if bias:
            layer.bias.copy_(torch.from_numpy(g[f"{tag}_b"]))
    layer = layer.to(dev)
    x = torch.from_numpy(g[f"{tag}_x"]).to(dev)
    with torch.no_grad():
        y_eval = layer.eval()(x).cpu().numpy()
        y_train = layer.train()(x).cpu().numpy()
        y_eval16 = layer.eval()(x.half())
    if tag == "r":
        assert np.max(np.abs(y_eval - g["r_y_eval"])) <= 1e-5 and np.max(np.abs(y_train - g["r_y_train"])) <= 1e-5
    else:
        assert np.array_equal(y_eval, g[f"{tag}_y_eval"])
        assert np.max(np.abs(y_train - g[f"{tag}_y_train"])) <= 1e-5
        assert torch.equal(y_eval16.cpu(), torch.from_numpy(g[f"{tag}_y_eval"]).half())
    want = O.smooth_binary_forward(g[f"{tag}_x"], g[f"{tag}_w"], g[f"{tag}_b"] if bias else None, float(g[f"{tag}_maxval"]))
    assert np.max(np.abs(y_eval - want)) <= (1e-5 if tag == "r" else 0.0)
    assert int(layer.count_converged_parameters(1e-1)) == int(g[f"{tag}_converged_1e-1"])
    assert layer.count_binary_parameters() == n * k


def test_training_forward_backward(tv, dev):
    """The layers stay trainable (straight-through estimator); gradients against a dense torch model."""
    torch.manual_seed(1)
    layer = tv.ScaledTernaryLinear(48, 24).to(dev).train()
    x = torch.randn(10, 48, device=dev, requires_grad=True)
    y = layer(x)
    y.square().sum().backward()
    q = ((layer.weights >= 0.5).float() - (layer.weights <= -0.5).float()).detach()
    x2 = x.detach().clone().requires_grad_(True)
    wl = layer.weights.detach().clone().requires_grad_(True)
    s2 = layer.scale.detach().clone().requires_grad_(True)
    b2 = layer.bias.detach().clone().requires_grad_(True)
    y2 = (nn.functional.linear(x2, q + wl - wl.detach()) * s2 + b2).clip(-100, 100)
    y2.square().sum().backward()
    assert torch.allclose(y, y2, atol=1e-4)
    for a, b in ((x.grad, x2.grad), (layer.weights.grad, wl.grad), (layer.scale.grad, s2.grad), (layer.bias.grad, b2.grad)):
        assert torch.allclose(a, b, rtol=1e-3, atol=1e-3)


def test_quantizer_functions(tv, dev, golden_dir):
    g = np.load(os.path.join(golden_dir, "quantizers.npz"))
    x = torch.from_numpy(g["x_grid"]).to(dev)
    m = tv.modules
    assert np.array_equal(m.tri_step_ste(x).cpu().numpy(), g["ternary_grid"])      # works without requires_grad too
    xr = x.clone().requires_grad_(True)
    out = m.tri_step_ste(xr)
    out.sum().backward()
    assert torch.equal(xr.grad, torch.ones_like(xr))                               # straight-through
    assert np.array_equal(m.sign_ste(x).cpu().numpy(), g["sign_grid"])
    assert np.array_equal(m.step_ste(x).cpu().numpy(), g["step_grid"])
    xs = (x * 2).clone().requires_grad_(True)
    m.sign_ste_sat(xs).sum().backward()
    assert torch.equal(xs.grad, ((xs >= -1) & (xs <= 1)).float())
    # the saturating forms (binary/modules.py:29-51): same forward values, function and module
    assert np.array_equal(m.sign_ste_sat(x).cpu().numpy(), g["sign_sat_grid"])
    assert np.array_equal(m.step_ste_sat(x).cpu().numpy(), g["step_sat_grid"])
    mod = tv.SignSTESat()
    assert np.array_equal(mod(x).cpu().numpy(), g["sign_sat_module_grid"])
    xm = (x * 2).clone().requires_grad_(True)
    mod(xm).sum().backward()
    assert torch.equal(xm.grad, xs.grad)


def test_linear_to_quantized_matches_reference_golden(tv, dev, golden_dir):
    g = np.load(os.path.join(golden_dir, "linear_to_quantized.npz"))
    for key in g.files:
        if not key.endswith("__scale"):
            continue
        quant, ns, sc = key.split("__")[:3]
        try:
            sc = float(sc)
        except ValueError:
            pass
        holder = nn.Sequential(nn.Linear(96, 40)).to(dev)
        with torch.no_grad():
            holder[0].weight.copy_(torch.from_numpy(g["w"]))
            holder[0].bias.copy_(torch.from_numpy(g["b"]))
        assert tv.linear_to_quantized(holder, quantization=quant, scaling=sc, neuron_scale=ns)
        q = holder[0]
        assert type(q).__name__ == {"ternary": "ScaledTernaryLinear", "binary": "ScaledBinaryLinear", "binary01": "ScaledBinary01Linear"}[quant]
        np.testing.assert_allclose(q.scale.detach().cpu().numpy(), g[key], rtol=1e-5, err_msg=key)
        np.testing.assert_allclose(q.weights.detach().cpu().numpy(), g[key.replace("__scale", "__latent")], rtol=1e-5, atol=1e-6, err_msg=key)
        assert np.array_equal(q.bias.detach().cpu().numpy(), g[key.replace("__scale", "__bias")])
    # smoothBinary: weights and bias copied, no scale (binary/modules.py:267-274); the evaluation forward is the packed matmul
    holder = nn.Sequential(nn.Linear(96, 40)).to(dev)
    with torch.no_grad():
        holder[0].weight.copy_(torch.from_numpy(g["w"]))
        holder[0].bias.copy_(torch.from_numpy(g["b"]))
    assert tv.linear_to_quantized(holder, quantization="smoothBinary")
    q = holder[0]
    assert type(q).__name__ == "SmoothlyBinarizedLinear" and not hasattr(q, "scale")
    assert np.array_equal(q.weights.detach().cpu().numpy(), g["smoothBinary__uniform__none__latent"])
    assert np.array_equal(q.bias.detach().cpu().numpy(), g["smoothBinary__uniform__none__bias"])
    xg = (torch.randint(-64, 65, (5, 96)).float() / 64)
    with torch.no_grad():
        got = q.eval()(xg.to(dev)).cpu().numpy()
    assert np.max(np.abs(got - O.smooth_binary_forward(xg.numpy(), g["w"], g["b"]))) <= 1e-5
    holder = nn.Sequential(nn.Linear(8, 4)).to(dev)
    with pytest.raises(ValueError):
        tv.linear_to_quantized(holder, quantization="binary", neuron_scale="none")     # modules.py:211-212
    with pytest.raises(ValueError):
        tv.linear_to_quantized(holder, quantization="binary", scaling="bogus")         # modules.py:223-224
    assert tv.linear_to_quantized(holder, quantization="ternary", neuron_scale="none")  # accepted, per row (convert.py:72-73)
    assert not tv.linear_to_quantized(nn.Sequential(nn.ReLU()), quantization="ternary")


class _Block(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.mixer = nn.Module()
        self.mixer.Wqkv = nn.Linear(d, 3 * d)
        self.mixer.out_proj = nn.Linear(d, d)
        self.mlp = nn.Module()
        self.mlp.fc1 = nn.Linear(d, 4 * d)
        self.mlp.fc2 = nn.Linear(4 * d, d)


class _Phi(nn.Module):
    def __init__(self, d=64, n=4):
        super().__init__()
        self.transformer = nn.Module()
        self.transformer.h = nn.ModuleList([_Block(d) for _ in range(n)])
        self.lm_head = nn.Module()
        self.lm_head.linear = nn.Linear(d, 100)


def test_quantize_text_model_selection_rule(tv, dev):
    """binary/utils.py:14-21: skip the first start_skip and last last_skip blocks and lm_head."""
    phi = _Phi().to(dev)
    names = tv.utils.quantize_text_model(phi, start_skip=1, last_skip=1, quantization="ternary", neuron_scale="independent")
    assert sorted(names) == sorted([f"transformer.h.{i}.{p}" for i in (1, 2) for p in ("mixer", "mlp")])
    assert isinstance(phi.transformer.h[0].mlp.fc1, nn.Linear) and isinstance(phi.transformer.h[3].mlp.fc1, nn.Linear)
    assert isinstance(phi.transformer.h[1].mlp.fc1, tv.ScaledTernaryLinear) and isinstance(phi.lm_head.linear, nn.Linear)
    tv.utils.clip_binary_weights(phi)
    assert float(phi.transformer.h[1].mlp.fc1.weights.detach().abs().max()) <= 1.0
    bits = tv.utils.quantized_size_bits(phi)
    dense = sum(16 * p.numel() for b in (0, 3) for p in phi.transformer.h[b].parameters()) + 16 * (64 * 100 + 100)
    tern = sum(2 * m.weights.numel() + 32 * m.bias.numel() for b in (1, 2) for m in phi.transformer.h[b].modules()
               if isinstance(m, tv.ScaledTernaryLinear))
    assert bits == dense + tern                                  # binary/utils.py:50-55


@pytest.mark.parametrize("variant", ["matmul", "unpack", "tf", "packedmm", "continuous"])
@pytest.mark.parametrize("k", [64, 52])
def test_packed_layer_variants_agree(tv, dev, variant, k):
    """Every storage variant of impl/tf/moondream/ternary.py computes the same function of the same torch layer."""
    rng = np.random.default_rng(k)
    n, m = 40, 9
    src = tv.ScaledTernaryLinear(k, n, max_val=3.0).to(dev)
    with torch.no_grad():
        src.weights.copy_(torch.from_numpy(rng.uniform(-1, 1, (n, k)).astype(np.float32)))
        src.scale.copy_(torch.from_numpy(rng.uniform(0.5, 1.5, (n,)).astype(np.float32)))
    x = torch.from_numpy(rng.integers(-64, 65, (2, m, k)).astype(np.float32) / 64).to(dev)
    layer = tv.conversion.VARIANTS[variant](n, activation=None, clip_val=100.0).to(dev)
    layer(x)                                              # builds with random packed bytes, like Keras
    tv.conversion.transfer_dense_params_torch_to_packed(layer, src)
    assert layer.clip_val == 3.0
    with torch.no_grad():
        want = src(x)
        got = layer(x)
    assert torch.equal(got, want)
    want_np = O.scaled_linear_forward(x.cpu().numpy(), src.weights.detach().cpu().numpy(), src.scale.detach().cpu().numpy(),
                                      src.bias.detach().cpu().numpy(), 3.0)
    assert np.array_equal(got.cpu().numpy(), want_np)
    w = layer.get_weights()
    assert len(w) == 3 and w[0].shape == (n,) and w[1].shape == (n,)     # [scale, bias, packed | kernel]
    if variant in ("matmul", "unpack", "packedmm"):
        assert w[2].shape == (n, (k + 3) // 4) and w[2].dtype == torch.uint8
    # fused vs unfused activation
    layer.act = tv.layers.gelu_tanh
    with torch.no_grad():
        ga = layer(x)
    assert torch.allclose(ga, nn.functional.gelu(want, approximate="tanh"), atol=2e-6)


def test_to_packed_inference_and_arena(tv, dev, tmp_path):
    torch.manual_seed(3)
    phi = _Phi(d=64, n=3).to(dev)
    tv.utils.quantize_text_model(phi, start_skip=0, last_skip=0, quantization="ternary", neuron_scale="independent")
    x = (torch.randint(-64, 65, (5, 64)).float() / 64).to(dev)
    blk = phi.transformer.h[1]
    with torch.no_grad():
        want_fc1 = blk.mlp.fc1(x)
        want_qkv = blk.mixer.Wqkv(x)
    arena = tv.PackedArena.from_model(phi)
    assert set(arena.entries) == {f"transformer.h.{i}.{p}" for i in range(3) for p in ("mixer.Wqkv", "mixer.out_proj", "mlp.fc1", "mlp.fc2")}
    assert torch.equal(arena.matmul("transformer.h.1.mlp.fc1", x), want_fc1)
    assert torch.equal(arena.matmul("transformer.h.1.mixer.Wqkv", x.half(), out_dtype=torch.float32), want_qkv)
    assert arena.packed.numel() == sum(e.n * e.ld for e in arena.entries.values())
    # disk round trip (safetensors, no pickle)
    path = str(tmp_path / "arena.safetensors")
    arena.save(path)
    back = tv.PackedArena.load(path, dev)
    assert torch.equal(back.packed, arena.packed) and torch.equal(back.scales, arena.scales)
    assert torch.equal(back.matmul("transformer.h.1.mlp.fc1", x), want_fc1)
    # from a reference-style state dict (names ...weights / ...scale / ...bias)
    sd = {k_: v.detach().cpu() for k_, v in phi.state_dict().items()}
    arena2 = tv.PackedArena.from_state_dict(sd, dev, select=lambda p: "lm_head" not in p)
    assert torch.equal(arena2.packed, arena.packed)
    names = tv.conversion.to_packed_inference(phi, "matmul")
    assert len(names) == 12 and isinstance(blk.mlp.fc1, tv.layers.ScaledTernaryMatmul)
    with torch.no_grad():
        assert torch.allclose(blk.mlp.fc1(x), nn.functional.gelu(want_fc1, approximate="tanh"), atol=2e-6)  # fc1 carries the GELU
        assert torch.equal(blk.mixer.Wqkv(x), want_qkv)


def test_arena_from_safetensors_checkpoint(tv, dev, tmp_path):
    """A reference-style checkpoint file (parameter names of binary/modules.py:96-98 under the moondream module tree) from
    a local path -> packed arena: the selection rule of binary/utils.py:14-21 by default, every layer's arena bytes equal
    to the oracle's ternarize + format-C pack, scales and biases verbatim (utils/conversion.py:61-77)."""
    from safetensors.torch import save_file
    rng = np.random.default_rng(42)
    d, n_blocks = 48, 4                                  # K = 48 / 192: packed rows of 12 / 48 bytes, padded to 16 / 48
    shapes = {"mixer.Wqkv": (3 * d, d), "mixer.out_proj": (d, d), "mlp.fc1": (4 * d, d), "mlp.fc2": (d, 4 * d)}
    sd = {}
    for i in range(n_blocks):
        for name, (n, k) in shapes.items():
            p = f"text_model.transformer.h.{i}.{name}"
            w = rng.uniform(-1, 1, (n, k)).astype(np.float32)
            w.reshape(-1)[:4] = [0.5, -0.5, 0.4999, -0.4999]
            sd[p + ".weights"] = torch.from_numpy(w)
            sd[p + ".scale"] = torch.from_numpy(rng.uniform(0.5, 1.5, n).astype(np.float32))
            if name != "mixer.out_proj":
                sd[p + ".bias"] = torch.from_numpy(rng.normal(size=n).astype(np.float32))
        sd[f"text_model.transformer.h.{i}.ln.weight"] = torch.ones(d)
    sd["text_model.lm_head.linear.weights"] = torch.randn(100, d)      # quantized-looking, but lm_head is never packed
    sd["text_model.lm_head.linear.scale"] = torch.ones(100)
    sd["text_model.transformer.embd.wte.weight"] = torch.randn(100, d)
    path = str(tmp_path / "model.safetensors")
    save_file(sd, path)

    def check(arena, blocks):
        want_names = [f"text_model.transformer.h.{i}.{name}" for i in blocks for name in shapes]
        assert list(arena.entries) == want_names                      # file order by block index, natural sort
        for p in want_names:
            e = arena.entries[p]
            w = sd[p + ".weights"].numpy()
            pw, scale, bias, maxval = arena.layer(p)
            want_pw = O.pack_format_c(O.ternarize(w))
            assert e.ld % 16 == 0 and e.ld >= want_pw.shape[1] and (e.n, e.k) == w.shape
            got = pw.cpu().numpy()
            assert np.array_equal(got[:, :want_pw.shape[1]], want_pw), p
            assert np.all(got[:, want_pw.shape[1]:] == 0x55)          # row padding: code 1 = weight 0
            assert np.array_equal(scale.cpu().numpy(), sd[p + ".scale"].numpy())
            if p + ".bias" in sd:
                assert np.array_equal(bias.cpu().numpy(), sd[p + ".bias"].numpy())
            else:
                assert bias is None
            assert maxval == 100.0
        return want_names

    arena = tv.PackedArena.from_safetensors(path, dev)                # defaults of quantize_moondream: start_skip=1, last_skip=0
    names = check(arena, [1, 2, 3])
    assert sorted(arena.skipped) == sorted([f"text_model.transformer.h.0.{n_}" for n_ in shapes] + ["text_model.lm_head.linear"])
    check(tv.PackedArena.from_safetensors(path, dev, start_skip=1, last_skip=1), [1, 2])       # convert.py:72
    check(tv.PackedArena.from_safetensors(path, dev, start_skip=0, last_skip=0), [0, 1, 2, 3])
    only = tv.PackedArena.from_safetensors(path, dev, select=lambda p: p.endswith("h.2.mlp.fc2"))
    assert list(only.entries) == ["text_model.transformer.h.2.mlp.fc2"]
    # the arena multiplies like the layer it came from
    p = names[2]
    x = rng.integers(-64, 65, (5, d)).astype(np.float32) / 64
    want = O.scaled_linear_forward(x, sd[p + ".weights"].numpy(), sd[p + ".scale"].numpy(), sd[p + ".bias"].numpy(), 100.0)
    assert np.array_equal(arena.matmul(p, torch.from_numpy(x).to(dev)).cpu().numpy(), want)
    with pytest.raises(ValueError):
        tv.PackedArena.from_safetensors(path, dev, select=lambda p: False)


# ------------------------------------------------------------------------------ decode-loop harness

def test_decode_harness_matches_dense_reference_and_graph_replay(dev):
    """Packed matmuls inside a (small) Phi-style decoder: prefill + steps against the same stack with dense float32
    torch layers; the graph-captured step reproduces the eager step bit for bit; Wqkv | fc1 as one matmul (the
    default) gives the same bits as the two separate matmuls."""
    from ternary_vllm_amd import decode_harness as dh
    cfg = dh.DecoderConfig(hidden=256, heads=4, intermediate=1024, blocks=2, max_len=64)
    cfg_unfused = dh.DecoderConfig(hidden=256, heads=4, intermediate=1024, blocks=2, max_len=64, fuse_qkv_fc1=False)
    torch.manual_seed(3)
    prompt = torch.randn(2, 19, cfg.hidden, device=dev)
    toks = [torch.randn(2, 1, cfg.hidden, device=dev) for _ in range(4)]

    def run(reference, graph, cfg=cfg):
        st = dh.TernaryDecoderStack(cfg, batch=2, device=dev, dtype=torch.float32, seed=11)
        assert len(st.blocks[0].linears()) == (3 if cfg.fuse_qkv_fc1 else 4)
        if reference:
            st.linear = lambda layer, x: layer.reference(x)
            st.fused_calls = False
        outs = [st.prefill(prompt)]
        if graph:
            st.capture_decode(toks[0])
        for t in toks:
            outs.append((st.decode_graph(t) if graph else st.decode_step(t)).clone())
        assert int(st.pos) == 19 + len(toks)
        return outs

    packed, dense, replay = run(False, False), run(True, False), run(False, True)
    for a, b in zip(packed, dense):
        assert a.shape == b.shape
        assert float((a - b).abs().max()) <= 2e-4 * max(1.0, float(b.abs().max()))
    for a, b in zip(packed, replay):
        assert torch.equal(a, b)
    for a, b in zip(packed, run(False, False, cfg_unfused)):
        assert torch.equal(a, b)
    # the block's closing sum inside fc2's call (the default) = the three tensors added by torch
    cfg_sum = dh.DecoderConfig(hidden=256, heads=4, intermediate=1024, blocks=2, max_len=64, fuse_residual=False)
    for a, b in zip(packed, run(False, False, cfg_sum)):
        assert torch.equal(a, b)


def test_decode_harness_at_the_models_dimensions(dev):
    """The harness at moondream2's text-model dimensions (impl/tf/moondream/layers.py:385-405: hidden 2048, 32 heads, intermediate
    8192), two blocks, batch 256, f16: a short prefill and two single-token steps against the same stack with dense float32
    torch layers (same f16 tensors between the layers), and the fused forms (Wqkv | fc1 as one product, the closing sum inside
    fc2's call) against the unfused ones bit for bit."""
    from ternary_vllm_amd import decode_harness as dh
    kw = dict(hidden=2048, heads=32, intermediate=8192, blocks=2, max_len=16)
    torch.manual_seed(5)
    batch = 256
    prompt = torch.randn(batch, 3, 2048, device=dev).half()
    toks = [torch.randn(batch, 1, 2048, device=dev).half() for _ in range(2)]

    def run(reference, **cfg_kw):
        st = dh.TernaryDecoderStack(dh.DecoderConfig(**kw, **cfg_kw), batch=batch, device=dev, dtype=torch.float16, seed=21)
        if reference:
            st.linear = lambda layer, x: layer.reference(x)
            st.fused_calls = False
        outs = [st.prefill(prompt)] + [st.decode_step(t).clone() for t in toks]
        return outs

    packed, dense = run(False), run(True)
    for a, b in zip(packed, dense):
        assert a.shape == b.shape and a.dtype == torch.float16
        # both stacks round to f16 between layers; the packed products differ from the dense float32 ones by summation order only
        err = float((a.float() - b.float()).abs().max())
        assert err <= 4e-3 * max(1.0, float(b.float().abs().max())), err
    plain = run(False, fuse_qkv_fc1=False, fuse_residual=False)
    for a, b in zip(packed, plain):
        assert torch.equal(a, b)


def test_matmul_pair_equals_the_two_layers(tv, dev):
    """Wqkv | fc1 behind one LayerNorm as one packed matmul: same bits as the two layers run separately."""
    L = tv.layers
    torch.manual_seed(9)
    k = 256
    x = torch.randn(3, 5, k, device=dev).half()
    a, b = L.ScaledTernaryMatmul(96).to(dev), L.ScaledTernaryMatmul(160, activation="gelu_tanh").to(dev)
    for layer in (a, b):
        layer(x)   # build
        n = layer.units
        layer.set_weights([torch.rand(n) * 0.1 + 0.05, torch.randn(n) * 0.1, tv.pack_fn(torch.randint(-1, 2, (n, k), dtype=torch.int8, device=dev))])
    pair = L.ScaledTernaryMatmulPair(a, b)
    ya, yb = pair(x)
    assert torch.equal(ya, a(x)) and torch.equal(yb, b(x))
    relu = L.ScaledTernaryMatmul(160, activation=torch.relu).to(dev)     # an activation the kernel does not fuse
    relu(x); relu.set_weights(b.get_weights())
    ya2, yr = L.ScaledTernaryMatmulPair(a, relu)(x)
    assert torch.equal(ya2, ya) and torch.equal(yr, relu(x))
    with pytest.raises(ValueError):
        L.ScaledTernaryMatmulPair(b, a)                                   # the activated layer cannot come first


def test_prepared_matmul_matches_plain_call(dev):
    import ternary_vllm_amd as tv
    ops = tv.custom_ops_mod
    torch.manual_seed(5)
    for m, n, k in ((1, 256, 512), (40, 96, 128)):
        x = torch.randn(m, k, device=dev).half()
        pw = ops.ternarize_pack(torch.rand(n, k, device=dev) * 2 - 1)
        scale, bias = torch.rand(n, device=dev) + 0.5, torch.randn(n, device=dev)
        prep = ops.PreparedMatmul(x, pw, scale, bias, 100.0, activation="gelu_tanh")
        want = ops.ternary_matmul(x, pw, scale, bias, 100.0, activation="gelu_tanh")
        assert torch.equal(prep(), want)
        x.copy_(torch.randn(m, k, device=dev).half())          # new activations in the same buffer
        assert torch.equal(prep(), ops.ternary_matmul(x, pw, scale, bias, 100.0, activation="gelu_tanh"))
    with pytest.raises(ValueError):
        ops.PreparedMatmul(torch.randn(4, 64, device=dev).half()[:, ::2], pw[:, :8], scale, bias, 100.0)
    with pytest.raises(ValueError):     # one packed row expanded to many: rows overlap, the kernels would read past the buffer
        ops.ternary_matmul(x, pw[:1].expand(n, pw.shape[1]), scale, bias, 100.0)


def test_workspace_outgrown_after_graph_capture_stays_valid(dev):
    """A HIP graph captured on a stream keeps the address of that stream's split-K scratch buffer; a later, larger shape
    on the same stream outgrows the buffer.  The old buffer must stay allocated: the replay still gives the right result
    after other tensors have been allocated and written."""
    import ternary_vllm_amd as tv
    ops = tv.custom_ops_mod
    torch.manual_seed(11)
    m, n, k = 12, 512, 16384                       # few rows, long k: split-K through the workspace
    x = (torch.randint(-64, 65, (m, k), device=dev).float() / 64).half()
    pw = torch.randint(0, 171, (n, k // 4), dtype=torch.uint8, device=dev)
    scale, bias = torch.ones(n, device=dev), torch.zeros(n, device=dev)
    assert ops._lib.tvl_ternary_matmul_workspace_bytes_for(ops.F16, m, n, k) > 0
    out = torch.empty((m, n), dtype=torch.float32, device=dev)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        ops.ternary_matmul(x, pw, scale, bias, 1e6, out=out)
        want = out.clone()
        s.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            ops.ternary_matmul(x, pw, scale, bias, 1e6, out=out)
        # a much larger split-K shape on the same stream: the stream's buffer is replaced
        big = ops.ternary_matmul(torch.randn(16, 32768, device=dev).half(), torch.randint(0, 171, (4096, 8192), dtype=torch.uint8, device=dev),
                                 torch.ones(4096, device=dev), None, 1e6)
        junk = [torch.full((1 << 20,), float("nan"), device=dev) for _ in range(8)]   # whatever the allocator hands out next
        out.zero_()
        graph.replay()
        s.synchronize()
    assert torch.equal(out, want) and big.shape == (16, 4096) and len(junk) == 8
