"""Decode-loop harness: the packed matmuls inside a Phi-style text decoder, with synthetic weights.

The reference times its converted moondream2 text model with one prefill over the prompt + image
embeddings and then single-token steps (``test_tflite.py:193-218``: ``predict`` with a growing cache,
tokens per second over the steps after the first).  The checkpoint cannot be fetched here, so this harness
rebuilds the part of that loop the packed path lives in, from the model's published shape
(``impl/tf/moondream/layers.py:385-405``: hidden 2048, 32 heads, intermediate 8192, 24 blocks, partial rotary
0.5, layer-norm eps 1e-5) and block structure (``layers.py:334-347``: one LayerNorm feeding both attention and
MLP, ``out = attn + mlp + residual``): per block four packed matmuls - Wqkv, out_proj, fc1 (+ tanh-GELU in the
matmul epilogue, ``utils/conversion.py:185-187``) and fc2 - with random ternary weights.  Wqkv and fc1 read the
same LayerNorm output, so by default they run as ONE packed matmul over 6144 + 8192 channels whose activation
starts at channel 6144 (``DecoderConfig.fuse_qkv_fc1``): three launches per block instead of four.

A decode step has ~100 short kernels; launched one by one it is bound by the host.  The step is therefore
captured once into a HIP graph (static cache buffers, position kept on the device) and replayed.
"""
from __future__ import annotations

import math
import time
from dataclasses import dataclass
from typing import Dict, List, Optional

import torch

from . import custom_ops_mod as ops


@dataclass
class DecoderConfig:
    hidden: int = 2048
    heads: int = 32
    intermediate: int = 8192
    blocks: int = 24
    partial_rotary: float = 0.5
    rope_theta: float = 10000.0
    eps: float = 1e-5
    clip: float = 100.0
    max_len: int = 1024
    fuse_qkv_fc1: bool = True     # Wqkv | fc1 as one packed matrix, tanh-GELU from channel 3 * hidden on
    fuse_residual: bool = True    # the block's closing attn + mlp + residual as part of fc2's call (same bits as the separate additions)

    @property
    def head_dim(self) -> int:
        return self.hidden // self.heads

    @property
    def rotary_dim(self) -> int:
        return int(self.partial_rotary * self.head_dim)


class PackedLinear:
    """scale / bias / packed weights of one layer (the ``[scale, bias, pw]`` triple of ``utils/conversion.py:77``)."""

    def __init__(self, out_features: int, in_features: int, gen: torch.Generator, device, activation: Optional[str] = None,
                 clip: float = 100.0):
        t = torch.randint(-1, 2, (out_features, in_features), generator=gen, dtype=torch.int8)
        self.ternary = t                                   # kept on the host for reference computations
        self.pw = ops.pack_fn(t.to(device))
        # scale ~ 1/sqrt(fan_in * P(nonzero)) keeps activations O(1) through the stack
        self.scale = (torch.rand(out_features, generator=gen) * 0.5 + 0.75).mul_(1.0 / math.sqrt(in_features * 2.0 / 3.0)).to(device)
        self.bias = (torch.randn(out_features, generator=gen) * 0.02).to(device)
        self.activation = activation
        self.act_begin = 0                                 # first output channel that gets the activation
        self.clip = clip
        self.out_features, self.in_features = out_features, in_features

    @classmethod
    def stacked(cls, linear: "PackedLinear", activated: "PackedLinear") -> "PackedLinear":
        """Two layers on the same input as one: rows, scales and biases concatenated, ``activated``'s activation
        applied from its first channel on."""
        assert linear.in_features == activated.in_features and linear.activation is None and linear.clip == activated.clip
        f = cls.__new__(cls)
        f.ternary = torch.cat([linear.ternary, activated.ternary])
        f.pw = torch.cat([linear.pw, activated.pw]).contiguous()
        f.scale = torch.cat([linear.scale, activated.scale]).contiguous()
        f.bias = torch.cat([linear.bias, activated.bias]).contiguous()
        f.activation, f.act_begin, f.clip = activated.activation, linear.out_features, linear.clip
        f.out_features, f.in_features = linear.out_features + activated.out_features, linear.in_features
        return f

    def __call__(self, x: torch.Tensor, residual: Optional[torch.Tensor] = None, residual2: Optional[torch.Tensor] = None) -> torch.Tensor:
        return ops.ternary_matmul(x, self.pw, self.scale, self.bias, self.clip, activation=self.activation, act_begin=self.act_begin,
                                  residual=residual, residual2=residual2)

    def reference(self, x: torch.Tensor, residual: Optional[torch.Tensor] = None, residual2: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The same layer with dense float32 weights in plain torch (``binary/modules.py:101-109`` order); the optional addends
        are added by the caller's dtype rules (see TernaryDecoderStack._block)."""
        assert residual is None and residual2 is None, "the dense twin returns the product; its caller adds"
        w = self.ternary.to(x.device).float()
        y = x.float() @ w.t()
        y = y * self.scale + self.bias
        y = y.clip(-self.clip, self.clip)
        if self.activation == "gelu_tanh":
            y = torch.cat([y[..., :self.act_begin], torch.nn.functional.gelu(y[..., self.act_begin:], approximate="tanh")], dim=-1)
        return y

    @property
    def packed_bytes(self) -> int:
        return self.pw.numel() + 8 * self.out_features


class Block:
    def __init__(self, cfg: DecoderConfig, gen: torch.Generator, device, dtype):
        h = cfg.hidden
        self.cfg = cfg
        self.ln_w = (1.0 + 0.1 * torch.randn(h, generator=gen)).to(device=device, dtype=dtype)
        self.ln_b = (0.1 * torch.randn(h, generator=gen)).to(device=device, dtype=dtype)
        self.wqkv = PackedLinear(3 * h, h, gen, device, clip=cfg.clip)
        self.out_proj = PackedLinear(h, h, gen, device, clip=cfg.clip)
        self.fc1 = PackedLinear(cfg.intermediate, h, gen, device, activation="gelu_tanh", clip=cfg.clip)
        self.fc2 = PackedLinear(h, cfg.intermediate, gen, device, clip=cfg.clip)
        self.qkv_fc1 = PackedLinear.stacked(self.wqkv, self.fc1) if cfg.fuse_qkv_fc1 else None

    def linears(self) -> List[PackedLinear]:
        if self.qkv_fc1 is not None:
            return [self.qkv_fc1, self.out_proj, self.fc2]
        return [self.wqkv, self.out_proj, self.fc1, self.fc2]


def _rope_tables(cfg: DecoderConfig, device, dtype):
    d = cfg.rotary_dim
    inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, d, 2, dtype=torch.float32) / d))
    t = torch.arange(cfg.max_len, dtype=torch.float32)
    f = torch.outer(t, inv)
    emb = torch.cat([f, f], dim=-1)
    return emb.cos().to(device=device, dtype=dtype), emb.sin().to(device=device, dtype=dtype)


def _rotate_half(x):
    a, b = x[..., : x.shape[-1] // 2], x[..., x.shape[-1] // 2:]
    return torch.cat([-b, a], dim=-1)


class TernaryDecoderStack:
    """``blocks`` Phi-style decoder blocks whose four Dense layers are packed ternary matmuls."""

    def __init__(self, cfg: DecoderConfig, batch: int, device="cuda:0", dtype=torch.float16, seed: int = 1234):
        self.cfg, self.batch, self.device, self.dtype = cfg, batch, torch.device(device), dtype
        gen = torch.Generator().manual_seed(seed)
        self.blocks = [Block(cfg, gen, self.device, dtype) for _ in range(cfg.blocks)]
        self.cos, self.sin = _rope_tables(cfg, self.device, dtype)
        shape = (cfg.blocks, batch, cfg.heads, cfg.max_len, cfg.head_dim)
        self.k_cache = torch.zeros(shape, device=self.device, dtype=dtype)
        self.v_cache = torch.zeros(shape, device=self.device, dtype=dtype)
        self.pos = torch.zeros((), device=self.device, dtype=torch.long)     # tokens already in the cache
        self._graph = None
        self._static_in = None
        self._static_out = None
        self.linear = lambda layer, x: layer(x)    # swapped for PackedLinear.reference in tests
        self.fused_calls = True                    # tests that swap `linear` for the dense twin turn this off too

    # ------------------------------------------------------------------ one block, any number of new tokens
    def _block(self, i: int, blk: Block, h: torch.Tensor, positions: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
        cfg = self.cfg
        b, t, _ = h.shape
        x = torch.nn.functional.layer_norm(h, (cfg.hidden,), blk.ln_w.to(h.dtype), blk.ln_b.to(h.dtype), cfg.eps)
        if blk.qkv_fc1 is not None:
            z = self.linear(blk.qkv_fc1, x).to(h.dtype)
            qkv, fc1_out = z[..., : 3 * cfg.hidden].reshape(b, t, 3, cfg.heads, cfg.head_dim), z[..., 3 * cfg.hidden:]
        else:
            qkv, fc1_out = self.linear(blk.wqkv, x).to(h.dtype).view(b, t, 3, cfg.heads, cfg.head_dim), None
        q, k, v = (qkv[:, :, j].transpose(1, 2) for j in range(3))            # [b, heads, t, hd]
        rd = cfg.rotary_dim
        cos, sin = self.cos[positions].to(h.dtype), self.sin[positions].to(h.dtype)   # [t, rd]
        q = torch.cat([q[..., :rd] * cos + _rotate_half(q[..., :rd]) * sin, q[..., rd:]], dim=-1)
        k = torch.cat([k[..., :rd] * cos + _rotate_half(k[..., :rd]) * sin, k[..., rd:]], dim=-1)
        kc, vc = self.k_cache[i], self.v_cache[i]
        kc.index_copy_(2, positions, k.to(kc.dtype))
        vc.index_copy_(2, positions, v.to(vc.dtype))
        att = torch.nn.functional.scaled_dot_product_attention(q, kc.to(h.dtype), vc.to(h.dtype), attn_mask=mask)
        att = att.transpose(1, 2).reshape(b, t, cfg.hidden)
        attn_out = self.linear(blk.out_proj, att).to(h.dtype)
        if fc1_out is None:
            fc1_out = self.linear(blk.fc1, x).to(h.dtype)
        if self.cfg.fuse_residual and self.fused_calls:
            # layers.py:347 `attn_outputs + feed_forward_hidden_states + residual` = (mlp + attn) + h, inside fc2's call
            return blk.fc2(fc1_out, residual=attn_out, residual2=h)
        mlp = self.linear(blk.fc2, fc1_out).to(h.dtype)
        return attn_out + mlp + h

    def _forward(self, h: torch.Tensor, positions: torch.Tensor) -> torch.Tensor:
        # a new token at position p attends to cache slots <= p (slots beyond the filled part are masked out)
        mask = torch.arange(self.cfg.max_len, device=self.device)[None, :] <= positions[:, None]    # [t, max_len]
        for i, blk in enumerate(self.blocks):
            h = self._block(i, blk, h, positions, mask)
        return h

    # ------------------------------------------------------------------ prefill and decode
    def reset(self):
        self.k_cache.zero_(); self.v_cache.zero_(); self.pos.zero_()

    def prefill(self, emb: torch.Tensor) -> torch.Tensor:
        t = emb.shape[1]
        positions = torch.arange(t, device=self.device)
        out = self._forward(emb, positions)
        self.pos.fill_(t)
        return out

    def decode_step(self, emb: torch.Tensor) -> torch.Tensor:
        """One new token per sequence, launched kernel by kernel."""
        out = self._forward(emb, self.pos.view(1))
        self.pos += 1
        return out

    def capture_decode(self, emb_example: torch.Tensor):
        """Capture one decode step (all blocks) into a HIP graph; afterwards decode_graph(emb) replays it."""
        self._static_in = emb_example.clone()
        pos0 = self.pos.clone()
        s = torch.cuda.Stream(self.device)
        s.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(s):     # warm-up on a side stream: allocations and lazy initialisation happen here
            for _ in range(2):
                self.decode_step(self._static_in)
        torch.cuda.current_stream(self.device).wait_stream(s)
        self.pos.copy_(pos0)
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph):
            self._static_out = self.decode_step(self._static_in)
        self.pos.copy_(pos0)           # the capture itself runs nothing, but keep the position explicit

    def decode_graph(self, emb: torch.Tensor) -> torch.Tensor:
        self._static_in.copy_(emb)
        self._graph.replay()
        return self._static_out

    # ------------------------------------------------------------------ bookkeeping
    def packed_weight_bytes(self) -> int:
        return sum(l.packed_bytes for b in self.blocks for l in b.linears())

    def matmul_flops_per_token(self) -> int:
        return sum(2 * l.out_features * l.in_features for b in self.blocks for l in b.linears())


def run(cfg: DecoderConfig, batch: int, prompt_len: int, steps: int, device="cuda:0", dtype=torch.float16,
        use_graph: bool = True) -> Dict[str, float]:
    """Prefill `prompt_len` synthetic embeddings, then `steps` single-token steps; returns timings."""
    stack = TernaryDecoderStack(cfg, batch, device, dtype)
    gen = torch.Generator().manual_seed(7)
    prompt = torch.randn(batch, prompt_len, cfg.hidden, generator=gen).to(device=device, dtype=dtype)
    tok = torch.randn(batch, 1, cfg.hidden, generator=gen).to(device=device, dtype=dtype)
    stack.prefill(prompt); stack.reset()                       # warm-up (workspaces, kernels)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    stack.prefill(prompt)
    torch.cuda.synchronize()
    prefill_s = time.perf_counter() - t0
    if use_graph:
        stack.capture_decode(tok)
        step = stack.decode_graph
    else:
        step = stack.decode_step
    step(tok)                                                    # like the reference, the first step is not counted
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(tok)
    torch.cuda.synchronize()
    decode_s = time.perf_counter() - t0
    # the packed matmuls of one step on their own (same shapes, one graph): their share of the step
    xs = {l.in_features: torch.randn(batch, l.in_features, device=device).to(dtype) for b in stack.blocks for l in b.linears()}
    for b in stack.blocks:
        for l in b.linears():
            l(xs[l.in_features])
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for b in stack.blocks:
            for l in b.linears():
                l(xs[l.in_features])
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    mm_s = (time.perf_counter() - t0) / 20
    return {
        "matmuls_only_ms_per_step": mm_s * 1e3, "matmuls_only_weight_stream_GBps": stack.packed_weight_bytes() / mm_s / 1e9,
        "batch": batch, "prompt_len": prompt_len, "steps": steps, "blocks": cfg.blocks, "graph": use_graph,
        "prefill_ms": prefill_s * 1e3, "prefill_tokens_per_s": batch * prompt_len / prefill_s,
        "decode_ms_per_step": decode_s / steps * 1e3, "decode_tokens_per_s": batch * steps / decode_s,
        "packed_weight_MB": stack.packed_weight_bytes() / 1e6,
        "weight_stream_GBps": stack.packed_weight_bytes() / (decode_s / steps) / 1e9,
        "prefill_matmul_TFLOPs": stack.matmul_flops_per_token() * batch * prompt_len / prefill_s / 1e12,
    }
