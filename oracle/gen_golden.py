"""Generate the golden vectors under tests/golden/ from the reference's torch code.

TEST INFRASTRUCTURE ONLY.  Runs in the build container only (it needs
/root/reference, which does not exist on the GPU box); the vectors it writes
are committed, the reference is not.

    python oracle/gen_golden.py [--reference /root/reference] [--out tests/golden]

What is taken from the reference, by running it:
  * binary/modules.py: tri_step_ste / sign_ste / step_ste (called on tensors
    that require grad - the no-grad branch of those functions raises
    UnboundLocalError), ScaledTernaryLinear / ScaledBinaryLinear /
    ScaledBinary01Linear forward, SmoothlyBinarizedLinear forward (eval and
    train), the *_sat quantizers and the SignSTESat module, linear_to_quantized
    for all four quantizations and the neuron_scale settings.
  * matrix_mul_tests.py: the ternary_mm function only (the module body runs
    100-iteration timings at import, so the function is cut out with ast and
    executed on its own).
What cannot be run (TensorFlow / larq / the custom larq_compute_engine are
not installed): impl/tf/moondream/ternary.py.  The pack-format byte examples
in pack_examples.json are therefore worked by hand from that file's text.
"""
from __future__ import annotations

import argparse
import ast
import json
import os
import sys

import numpy as np
import torch


def load_ternary_mm(ref_root: str):
    src = open(os.path.join(ref_root, "matrix_mul_tests.py")).read()
    tree = ast.parse(src)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "ternary_mm"]
    assert len(fn) == 1
    mod = ast.Module(body=fn, type_ignores=[])
    ns = {"torch": torch}
    exec(compile(mod, "matrix_mul_tests.py:ternary_mm", "exec"), ns)
    return ns["ternary_mm"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(__file__), "..", "tests", "golden"))
    args = ap.parse_args()
    out = os.path.abspath(args.out)
    os.makedirs(out, exist_ok=True)
    sys.path.insert(0, args.reference)
    from binary import modules as ref  # noqa: E402
    from torch import nn

    torch.manual_seed(1234)
    rng = np.random.default_rng(1234)

    # ---- quantizers, including the inclusive thresholds -------------------
    # The reference's quantizers return q + x - x (straight-through term).  In float32 that is
    # NOT always q: (q + x) rounds, so the result can be 1 ulp off +-1 (0.99999994), and +-inf
    # gives NaN.  On the dyadic grid k/64 the sum is exact and the reference returns q itself.
    edge = np.array([-2.0, -1.0, -0.5000001, -0.5, -0.4999999, -0.4999, -0.25, -0.0, 0.0, 1e-30, 0.25,
                     0.4999, 0.4999999, 0.5, 0.5000001, 1.0, 2.0, np.inf, -np.inf], dtype=np.float32)
    rnd = rng.uniform(-1.0, 1.0, size=(237,)).astype(np.float32)
    q_in = np.concatenate([edge, rnd])
    grid = (np.arange(-128, 129).astype(np.float32) / 64.0)
    t_in = torch.from_numpy(q_in.copy()).requires_grad_(True)
    t_grid = torch.from_numpy(grid.copy()).requires_grad_(True)
    np.savez(os.path.join(out, "quantizers.npz"),
             x=q_in,
             ternary=ref.tri_step_ste(t_in).detach().numpy(),
             sign=ref.sign_ste(t_in).detach().numpy(),
             step=ref.step_ste(t_in).detach().numpy(),
             x_grid=grid,
             ternary_grid=ref.tri_step_ste(t_grid).detach().numpy(),
             sign_grid=ref.sign_ste(t_grid).detach().numpy(),
             step_grid=ref.step_ste(t_grid).detach().numpy(),
             sign_sat_grid=ref.sign_ste_sat(t_grid).detach().numpy(),
             step_sat_grid=ref.step_ste_sat(t_grid).detach().numpy(),
             sign_sat_module_grid=ref.SignSTESat()(t_grid).detach().numpy())

    # ---- dense forward of the three scaled layers --------------------------
    # Latent weights on the grid k/64: the straight-through term is then exact, the reference
    # multiplies by exactly {-1,0,+1} and its output can be compared bit for bit.  The "_u"
    # variants use uniform random latent weights, where the reference's weights are up to one
    # ulp off +-1 (see above) and its output is compared with the 1e-5 tolerance.
    fwd = {}
    for name, cls in (("ternary", ref.ScaledTernaryLinear), ("binary", ref.ScaledBinaryLinear),
                      ("binary01", ref.ScaledBinary01Linear), ("ternary_u", ref.ScaledTernaryLinear)):
        for bias in (True, False):
            torch.manual_seed(1234)
            n, k, m = 24, 52, 7
            layer = cls(k, n, bias=bias, max_val=3.0 if name.startswith("ternary") else 100.0)
            with torch.no_grad():
                if name.endswith("_u"):
                    layer.weights.copy_(torch.from_numpy(rng.uniform(-1, 1, size=(n, k)).astype(np.float32)))
                else:
                    layer.weights.copy_(torch.from_numpy(rng.integers(-64, 65, size=(n, k)).astype(np.float32) / 64.0))
                layer.weights[0, :4] = torch.tensor([0.5, -0.5, 0.4999, -0.4999])
                layer.scale.copy_(torch.from_numpy(rng.uniform(0.5, 1.5, size=(n,)).astype(np.float32)))
            # exact dyadic-grid activations: every partial sum is exact in float32
            x = (rng.integers(-64, 65, size=(m, k)).astype(np.float32) / 64.0)
            y = layer(torch.from_numpy(x.copy())).detach().numpy()
            tag = f"{name}_{'bias' if bias else 'nobias'}"
            fwd[f"{tag}_x"] = x
            fwd[f"{tag}_w"] = layer.weights.detach().numpy().copy()
            fwd[f"{tag}_scale"] = layer.scale.detach().numpy().copy()
            if bias:
                fwd[f"{tag}_b"] = layer.bias.detach().numpy().copy()
            fwd[f"{tag}_maxval"] = np.float32(layer.maxval)
            fwd[f"{tag}_y"] = y
    # one N(0,1) case (summation-order dependent: compared with a tolerance)
    torch.manual_seed(1234)
    layer = ref.ScaledTernaryLinear(256, 256)
    with torch.no_grad():
        layer.weights.copy_(torch.rand(256, 256) * 2 - 1)
        layer.scale.copy_(torch.rand(256) + 0.5)
    x = torch.randn(256, 256)
    fwd["ternary_randn_x"] = x.numpy().copy()
    fwd["ternary_randn_w"] = layer.weights.detach().numpy().copy()
    fwd["ternary_randn_scale"] = layer.scale.detach().numpy().copy()
    fwd["ternary_randn_b"] = layer.bias.detach().numpy().copy()
    fwd["ternary_randn_maxval"] = np.float32(layer.maxval)
    fwd["ternary_randn_y"] = layer(x.clone()).detach().numpy()
    np.savez(os.path.join(out, "scaled_linear_forward.npz"), **fwd)

    # ---- SmoothlyBinarizedLinear (binary/modules.py:112-146) -----------------
    # eval: weights binarized by (w >= 0) * 2 - 1, F.linear with the bias, clip.  Grid inputs, grid bias: every
    # partial sum (bias included, wherever the BLAS adds it) is exact, so the output can be compared bit for bit.
    # The "_r" case has a random-normal bias and N(0,1) inputs (tolerance).  train: latent weights clipped to [-1, 1].
    rng_s = np.random.default_rng(4321)   # its own stream: the fixtures generated after this block stay as they were
    sm = {}
    for tag, bias, grid in (("bias", True, True), ("nobias", False, True), ("r", True, False)):
        torch.manual_seed(77)
        n, k, m = 24, 52, 7
        layer = ref.SmoothlyBinarizedLinear(k, n, bias=bias, max_val=2.0 if grid else 100.0)
        with torch.no_grad():
            if grid:
                layer.weights.copy_(torch.from_numpy(rng_s.integers(-96, 97, size=(n, k)).astype(np.float32) / 64.0))
                layer.weights[0, :4] = torch.tensor([0.0, -0.0, 1e-30, -1e-30])
                if bias:
                    layer.bias.copy_(torch.from_numpy(rng_s.integers(-16, 17, size=(n,)).astype(np.float32) / 8.0))
                x = rng_s.integers(-64, 65, size=(m, k)).astype(np.float32) / 64.0
            else:
                layer.weights.mul_(1.5)
                x = rng_s.normal(size=(m, k)).astype(np.float32)
        sm[f"{tag}_x"] = x
        sm[f"{tag}_w"] = layer.weights.detach().numpy().copy()
        if bias:
            sm[f"{tag}_b"] = layer.bias.detach().numpy().copy()
        sm[f"{tag}_maxval"] = np.float32(layer.maxval)
        sm[f"{tag}_y_eval"] = layer.eval()(torch.from_numpy(x.copy())).detach().numpy()
        sm[f"{tag}_y_train"] = layer.train()(torch.from_numpy(x.copy())).detach().numpy()
        sm[f"{tag}_converged_1e-1"] = np.int64(layer.count_converged_parameters(1e-1))
    np.savez(os.path.join(out, "smooth_binary.npz"), **sm)

    # ---- linear_to_quantized ------------------------------------------------
    fit = {}
    cases = [("ternary", "uniform", "none"), ("ternary", "independent", "none"), ("ternary", "none", "none"),
             ("binary", "uniform", "none"), ("binary", "independent", "none"), ("binary", "independent", "natural"),
             ("binary", "uniform", "std"), ("binary", "uniform", 0.75),
             ("binary01", "uniform", "none"), ("binary01", "independent", "none"),
             ("binary01", "independent", "natural"), ("smoothBinary", "uniform", "none")]
    torch.manual_seed(1234)
    lin = nn.Linear(96, 40)
    with torch.no_grad():
        lin.weight.mul_(3.0)
    fit["w"] = lin.weight.detach().numpy().copy()
    fit["b"] = lin.bias.detach().numpy().copy()
    for quant, ns, sc in cases:
        holder = nn.Sequential(nn.Linear(96, 40))
        holder[0].load_state_dict(lin.state_dict())
        ok = ref.linear_to_quantized(holder, quantization=quant, scaling=sc, neuron_scale=ns)
        assert ok
        q = holder[0]
        tag = f"{quant}__{ns}__{sc}"
        fit[f"{tag}__latent"] = q.weights.detach().numpy().copy()
        if quant == "smoothBinary":     # no scale parameter (binary/modules.py:267-274)
            assert type(q).__name__ == "SmoothlyBinarizedLinear" and not hasattr(q, "scale")
            fit[f"{tag}__bias"] = q.bias.detach().numpy().copy()
            continue
        fit[f"{tag}__scale"] = q.scale.detach().numpy().copy()
        fit[f"{tag}__bias"] = q.bias.detach().numpy().copy()
    np.savez(os.path.join(out, "linear_to_quantized.npz"), **fit)

    # ---- ternary_mm (select-and-sum) ----------------------------------------
    ternary_mm = load_ternary_mm(args.reference)
    mm = {}
    for size in (32, 256):
        torch.manual_seed(1234 + size)
        m_ = torch.from_numpy(rng.integers(-64, 65, size=(size, size)).astype(np.float32) / 64.0)
        tm = torch.randn(size, size)
        tm[torch.rand(size, size) < 0.5] = 0.0  # matrix_mul_tests.py:15-18
        mm[f"m{size}"] = m_.numpy().copy()
        mm[f"tm{size}"] = tm.numpy().copy()
        mm[f"res{size}"] = ternary_mm(m_, tm).numpy().copy()
    torch.manual_seed(99)
    m_ = torch.randn(256, 256)
    tm = torch.randn(256, 256)
    tm[torch.rand(256, 256) < 0.5] = 0.0
    mm["m256r"] = m_.numpy().copy()
    mm["tm256r"] = tm.numpy().copy()
    mm["res256r"] = ternary_mm(m_, tm).numpy().copy()
    np.savez(os.path.join(out, "ternary_mm.npz"), **mm)

    # ---- literal known-answer tests held by the reference's own tests --------
    kats = {
        "zero_out": [
            {"source": "zero_out_tests.py:9-10", "input": [[5, 4, 3, 2, 1], [10, 9, 8, 7, 6]],
             "expected": [[5, 0, 0, 0, 0], [0, 0, 0, 0, 0]]},
            {"source": "native/tensorflow_custom_ops/python/ops/zero_out_ops_test.py:33-34",
             "input": [[1, 2], [3, 4]], "expected": [[1, 0], [0, 0]]},
        ],
        "roundtrip_length": {"source": "impl/tf/moondream/ternary_tests.py:19", "length": 5031},
    }
    with open(os.path.join(out, "reference_kats.json"), "w") as f:
        json.dump(kats, f, indent=1)

    # ---- hand-worked byte examples of the three pack layouts ------------------
    examples = {
        "_note": "Worked by hand from the text of impl/tf/moondream/ternary.py (TensorFlow is not "
                 "installed, the file cannot be run).  'values' is the ternary input, 'bytes' the packed output.",
        "format_a": [
            # S=2: byte j = code(t[j]) | code(t[2+j])<<2 | code(t[4+j])<<4 | code(t[6+j])<<6 ; +1->01, -1->10
            {"source": "ternary.py:57-78", "values": [1, -1, 0, 1, -1, 0, 1, 1], "bytes": [0b01100001, 0b01000110]},
            {"source": "ternary.py:57-78 (length 5 pads to 8)", "values": [1, 1, -1, -1, 1],
             "bytes": [0b00011001, 0b00001001]},
        ],
        "format_b": [
            # code = t+1, planar
            {"source": "ternary.py:114-133", "values": [1, 1, 1, 1], "bytes": [0b10101010]},
            {"source": "ternary.py:114-133", "values": [1, -1, 0, 1, -1, 0, 1, 1], "bytes": [0b10000110, 0b10011000]},
        ],
        "format_c": [
            # byte[r,j] = sum_i (t[r,4j+i]+1) << 2i
            {"source": "ternary.py:374-398", "values": [[1, -1, 0, -1]], "bytes": [[0b00010010]]},
            {"source": "ternary.py:374-398", "values": [[0, 0, 0, 0, 1, 1, 1, 1], [-1, -1, -1, -1, 1, 0, -1, 0]],
             "bytes": [[0b01010101, 0b10101010], [0b00000000, 0b01000110]]},
            {"source": "ternary.py:374-398 (K=5 pads with value 0 -> code 1)", "values": [[1, 1, 1, 1, -1]],
             "bytes": [[0b10101010, 0b01010100]]},
        ],
    }
    with open(os.path.join(out, "pack_examples.json"), "w") as f:
        json.dump(examples, f, indent=1)
    print("golden vectors written to", out)


if __name__ == "__main__":
    main()
