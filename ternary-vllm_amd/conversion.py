"""Weight transfer from the trainable torch layers to the packed inference layers, and the packed arena.

Mirror of the ternary branches of ``transfer_dense_params_torch_to_tf`` (``utils/conversion.py:32-94``):
``ternary_act(w) -> int8 -> pack -> set_weights([scale, bias, packed])`` with ``clip_val = maxval``.  Here the
quantize and pack steps are one HIP kernel (``ternarize_pack``); nothing is computed on the host.

The arena is this build's on-device / on-disk form of a converted checkpoint: every selected layer's packed
weights in ONE contiguous uint8 buffer (rows padded to 16 bytes so each layer is directly usable by the
matrix-core kernel), plus concatenated scales and biases and a small index.  It is what gets broadcast to
the other GPUs when activations are sharded by rows (one collective for all layers).
"""
from __future__ import annotations

import json
from dataclasses import dataclass
from typing import Dict, Iterable, List, Mapping, Optional, Tuple

import torch
from torch import nn

from . import custom_ops_mod as ops
from .layers import (ScaledTernary, ScaledTernaryMatmul, ScaledTernaryPackedMM, ScaledTernaryTf, ScaledTernaryUnpack,
                     gelu_tanh)
from .modules import ScaledBinary01Linear, ScaledBinaryLinear, ScaledTernaryLinear, _ScaledQuantLinear

VARIANTS = {"matmul": ScaledTernaryMatmul, "unpack": ScaledTernaryUnpack, "tf": ScaledTernaryTf,
            "packedmm": ScaledTernaryPackedMM, "continuous": ScaledTernary}


@torch.no_grad()
def transfer_dense_params_torch_to_packed(dense_packed: nn.Module, dense_torch: _ScaledQuantLinear):
    """``transfer_dense_params_torch_to_tf`` (``utils/conversion.py:32-94``) for the packed layer classes.
    ``dense_packed`` must be built (``build((..., K))``)."""
    assert isinstance(dense_torch, _ScaledQuantLinear), "expected a Scaled{Ternary,Binary,Binary01}Linear"
    w, b, scale = dense_torch.weights.detach(), dense_torch.bias, dense_torch.scale.detach()
    b = b.detach() if b is not None else torch.zeros_like(scale)
    quant = dense_torch.quantization
    dense_packed.clip_val = dense_torch.maxval  # conversion.py:39,56,75,90
    if isinstance(dense_packed, ScaledTernary):
        # conversion.py:41: latent weights transposed to the Keras kernel layout [K, units]
        dense_packed.set_weights([scale, b, w.t().contiguous()])
    elif isinstance(dense_packed, ScaledTernaryTf):
        # conversion.py:52-58: quantize, flatten, planar pack
        t = ops.quantize(w, quant, dtype=torch.int8)
        dense_packed.set_weights([scale, b, dense_packed.pack_fn(t.reshape(-1))])
    elif isinstance(dense_packed, ScaledTernaryMatmul):  # also Unpack and PackedMM (subclasses)
        # conversion.py:71-77 / :87-92: quantize + row-interleaved pack, fused into one kernel
        dense_packed.set_weights([scale, b, ops.ternarize_pack(w, 0.5, quant)])
    else:
        raise TypeError(f"unsupported packed layer type {type(dense_packed).__name__}")


@torch.no_grad()
def to_packed_inference(model: nn.Module, variant: str = "matmul", gelu_names: Iterable[str] = ("fc1",)) -> List[str]:
    """Swap every ``Scaled*Linear`` in ``model`` for a packed inference layer of the given variant
    (``torch_moondream_to_keras``, ``utils/conversion.py:175-196``): layers whose attribute name is in
    ``gelu_names`` get the tanh-GELU activation like ``fc1`` (``:185-187``).  Note that the activation then
    becomes part of the layer: remove it from the surrounding block.  Returns the converted names."""
    if variant not in VARIANTS:
        raise ValueError(f"Unknown variant: {variant}")
    done = []
    for parent_name, parent in list(model.named_modules()):
        for name, mod in list(parent.named_children()):
            if isinstance(mod, _ScaledQuantLinear):
                act = gelu_tanh if name in set(gelu_names) else None
                layer = VARIANTS[variant](mod.out_features, activation=act, clip_val=mod.maxval).to(mod.weights.device)
                layer.build((1, mod.in_features))
                transfer_dense_params_torch_to_packed(layer, mod)
                parent.add_module(name, layer)
                done.append(f"{parent_name}.{name}" if parent_name else name)
    return done


# ---------------------------------------------------------------------------------------------------
# packed arena
# ---------------------------------------------------------------------------------------------------

ROW_ALIGN = 16  # bytes: keeps every layer's rows 16-byte aligned inside the arena
PACK_FLUSH_LAYERS = 64        # layers per batched pack call (one launch takes up to 64 matrix descriptors)
PACK_FLUSH_BYTES = 2 << 30    # ... or this many bytes of latent weights waiting on the device, whichever comes first


def _natural_key(name: str):
    """Sort key that orders ``h.2.`` before ``h.10.``."""
    import re
    return [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", name)]


@dataclass
class ArenaEntry:
    name: str
    n: int
    k: int
    ld: int          # packed row length in bytes
    offset: int      # byte offset of the layer's first packed row
    vec_offset: int  # element offset into scales / biases
    maxval: float
    has_bias: bool
    quantization: str


def checkpoint_layer_selection(keys, start_skip: int = 1, last_skip: int = 0, n_blocks: Optional[int] = None, select=None):
    """Which quantized layers of a checkpoint are packed: ``(selected prefixes, skipped prefixes)``, both in natural order.
    A quantized layer is every prefix with ``<prefix>.weights`` and ``<prefix>.scale`` among ``keys``.  The rule is
    ``quantize_moondream``'s (``binary/utils.py:14-21``) evaluated where the reference evaluates it - on the name of the
    module whose children ``linear_to_quantized`` converts (``binary/modules.py:191``), i.e. the layer's parent."""
    import re

    keys = set(keys)
    if n_blocks is None:
        blocks = [int(mo.group(1)) for key in keys for mo in [re.search(r"transformer\.h\.(\d+)\.", key)] if mo]
        n_blocks = max(blocks) + 1 if blocks else 0
    from .utils import layers_to_skip
    skip = layers_to_skip(n_blocks, start_skip, last_skip)

    def parent_rule(prefix):
        parent = prefix.rsplit(".", 1)[0] if "." in prefix else ""
        return "lm_head" not in parent and not any(s_ in parent for s_ in skip)

    rule = select if select is not None else parent_rule
    chosen, skipped = [], []
    for key in sorted(keys, key=_natural_key):
        if not key.endswith(".weights"):
            continue
        prefix = key[: -len(".weights")]
        if prefix + ".scale" not in keys:
            continue
        (chosen if rule(prefix) else skipped).append(prefix)
    return chosen, skipped


class PackedArena:
    """All packed layers of a model in one device buffer.  ``layer(name)`` returns views
    ``(pw [N, ld] uint8, scale [N], bias [N] | None, maxval)`` ready for ``ternary_matmul``."""

    def __init__(self, packed: torch.Tensor, scales: torch.Tensor, biases: torch.Tensor, entries: List[ArenaEntry]):
        self.packed, self.scales, self.biases = packed, scales, biases
        self.entries = {e.name: e for e in entries}
        self.skipped: List[str] = []

    # -- construction -------------------------------------------------------------------------
    @classmethod
    @torch.no_grad()
    def _build(cls, specs, fetch, device) -> "PackedArena":
        """``specs``: list of (name, n, k, maxval, has_bias, quantization); ``fetch(name)`` -> (latent weights [n,k],
        scale [n], bias [n] | None), called once per layer while the arena is filled.  Latent weights wait on the device for the
        batched pack call, which runs every ``PACK_FLUSH_LAYERS`` layers (or ``PACK_FLUSH_BYTES`` of latent weights): the peak is
        the arena plus one such batch, not the whole checkpoint."""
        entries, off, voff = [], 0, 0
        for name, n, k, maxval, has_bias, quant in specs:
            ld = -(-ops.packed_row_bytes(k) // ROW_ALIGN) * ROW_ALIGN
            entries.append(ArenaEntry(name, int(n), int(k), ld, off, voff, float(abs(maxval)), bool(has_bias), quant))
            off += n * ld
            voff += n
        packed = torch.full((off,), 0x55, dtype=torch.uint8, device=device)  # pad bytes = four zeros
        scales = torch.empty((voff,), dtype=torch.float32, device=device)
        biases = torch.zeros((voff,), dtype=torch.float32, device=device)
        pending = {}   # (quantization, dtype) -> (latent weights, arena views): packed together, one launch per batch
        held_layers = held_bytes = 0

        def flush():
            nonlocal held_layers, held_bytes
            for (quant, _), (ws, views) in pending.items():
                ops.ternarize_pack_batch(ws, 0.5, quant, outs=views)
            pending.clear()   # drops the device copies of the latent weights
            held_layers = held_bytes = 0

        for e in entries:
            w, scale, bias = fetch(e.name)
            if tuple(w.shape) != (e.n, e.k) or scale.numel() != e.n or (bias is not None and bias.numel() != e.n):
                raise ValueError(f"layer {e.name}: weights {tuple(w.shape)}, scale {tuple(scale.shape)} do not match [{e.n}, {e.k}]")
            view = packed[e.offset:e.offset + e.n * e.ld].view(e.n, e.ld)
            ws, views = pending.setdefault((e.quantization, w.dtype), ([], []))
            ws.append(w.to(device))
            views.append(view)
            held_layers += 1
            held_bytes += w.numel() * w.element_size()
            scales[e.vec_offset:e.vec_offset + e.n] = scale.to(device, torch.float32).reshape(-1)
            if bias is not None:
                biases[e.vec_offset:e.vec_offset + e.n] = bias.to(device, torch.float32).reshape(-1)
            if held_layers >= PACK_FLUSH_LAYERS or held_bytes >= PACK_FLUSH_BYTES:
                flush()
        flush()
        return cls(packed, scales, biases, entries)

    @classmethod
    def from_layers(cls, layers: Mapping[str, Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor], float, str]],
                    device=None) -> "PackedArena":
        """``layers``: name -> (latent weights [N,K], scale [N], bias [N] | None, maxval, quantization)."""
        specs = [(name, w.shape[0], w.shape[1], maxval, bias is not None, quant) for name, (w, _, bias, maxval, quant) in layers.items()]
        device = device or next(iter(layers.values()))[0].device
        return cls._build(specs, lambda name: layers[name][:3], device)

    @classmethod
    def from_model(cls, model: nn.Module) -> "PackedArena":
        layers = {}
        for name, mod in model.named_modules():
            if isinstance(mod, _ScaledQuantLinear):
                layers[name] = (mod.weights.detach(), mod.scale.detach(), mod.bias.detach() if mod.bias is not None else None,
                                mod.maxval, mod.quantization)
        if not layers:
            raise ValueError("model has no Scaled*Linear layers")
        return cls.from_layers(layers)

    @classmethod
    def from_state_dict(cls, state: Mapping[str, torch.Tensor], device, quantization: str = "ternary", maxval: float = 100.0,
                        select=None) -> "PackedArena":
        """Build from a checkpoint written by the reference's training script: a layer is every prefix
        that has ``<prefix>.weights`` and ``<prefix>.scale`` (parameter names of ``binary/modules.py:96-98``;
        ``<prefix>.bias`` optional).  ``select(prefix) -> bool`` narrows the choice."""
        layers = {}
        for key in state:
            if key.endswith(".weights"):
                p = key[: -len(".weights")]
                if p + ".scale" in state and (select is None or select(p)):
                    layers[p] = (state[key].to(device), state[p + ".scale"].to(device),
                                 state[p + ".bias"].to(device) if p + ".bias" in state else None, maxval, quantization)
        if not layers:
            raise ValueError("no '<prefix>.weights' / '<prefix>.scale' pairs found in the state dict")
        return cls.from_layers(layers, device)

    @classmethod
    def from_safetensors(cls, path: str, device, start_skip: int = 1, last_skip: int = 0, quantization: str = "ternary",
                         maxval: float = 100.0, select=None, n_blocks: Optional[int] = None) -> "PackedArena":
        """Convert a checkpoint written by the reference's training script (``train.py:336-338``) straight from a local
        ``.safetensors`` file: the counterpart of ``convert.py:70-79`` (quantize the selected layers, then fill their
        parameters from the file) followed by the pack call of ``utils/conversion.py:61-77``.

        A quantized layer is every prefix with ``<prefix>.weights`` (latent, [N,K]) and ``<prefix>.scale`` ([N]);
        ``<prefix>.bias`` is optional (``binary/modules.py:96-98``).  Which of them are packed follows
        ``quantize_moondream`` (``binary/utils.py:14-21``), which tests the name of the module whose CHILDREN
        ``linear_to_quantized`` converts (``binary/modules.py:191``): the rule is applied to the layer's PARENT name -
        no ``lm_head`` in it, and none of ``transformer.h.{i}.`` for the first ``start_skip`` / last ``last_skip`` blocks.
        (So a Linear that is a direct child of a skipped block, ``transformer.h.0.<linear>``, is converted, as in the
        reference: its parent ``transformer.h.0`` lacks the trailing dot.  Phi blocks hold their Linears one level
        deeper, under ``mixer`` / ``mlp``.)  ``n_blocks`` is ``len(transformer.h)``; if not given it is taken to be the
        highest block index in the file + 1, which is right unless trailing blocks hold no tensors.  ``select(prefix) ->
        bool`` replaces the rule.  The file is opened with ``safe_open`` (tensors only,
        nothing is executed) and read one layer at a time; prefixes the rule left out are listed in ``arena.skipped``."""
        from safetensors import safe_open
        with safe_open(path, framework="pt", device="cpu") as f:
            keys = set(f.keys())
            chosen, skipped = checkpoint_layer_selection(keys, start_skip, last_skip, n_blocks, select)
            specs = []
            for prefix in chosen:
                key = prefix + ".weights"
                shape = f.get_slice(key).get_shape()
                if len(shape) != 2:
                    raise ValueError(f"{key}: expected a 2-D latent weight matrix, got shape {shape}")
                specs.append((prefix, shape[0], shape[1], maxval, prefix + ".bias" in keys, quantization))
            if not specs:
                raise ValueError(f"{path}: no '<prefix>.weights' / '<prefix>.scale' pairs selected")

            def fetch(prefix):
                return (f.get_tensor(prefix + ".weights"), f.get_tensor(prefix + ".scale"),
                        f.get_tensor(prefix + ".bias") if prefix + ".bias" in keys else None)

            arena = cls._build(specs, fetch, device)
        arena.skipped = skipped
        return arena

    # -- use -------------------------------------------------------------------------------------
    def layer(self, name: str):
        e = self.entries[name]
        pw = self.packed[e.offset:e.offset + e.n * e.ld].view(e.n, e.ld)
        scale = self.scales[e.vec_offset:e.vec_offset + e.n]
        bias = self.biases[e.vec_offset:e.vec_offset + e.n] if e.has_bias else None
        return pw, scale, bias, e.maxval

    def matmul(self, name: str, x: torch.Tensor, activation=None, out_dtype=None) -> torch.Tensor:
        e = self.entries[name]
        if x.shape[-1] != e.k:
            raise ValueError(f"layer {name} expects K={e.k}, got {x.shape[-1]}")
        pw, scale, bias, maxval = self.layer(name)
        return ops.ternary_matmul(x, pw, scale, bias, maxval, activation=activation, out_dtype=out_dtype)

    def nbytes(self) -> int:
        return self.packed.numel() + 4 * (self.scales.numel() + self.biases.numel())

    # -- disk --------------------------------------------------------------------------------------
    def save(self, path: str):
        from safetensors.torch import save_file
        meta = {"index": json.dumps([e.__dict__ for e in self.entries.values()]), "format": "ternary-packed-arena-v1"}
        save_file({"packed": self.packed.cpu(), "scales": self.scales.cpu(), "biases": self.biases.cpu()}, path, metadata=meta)

    @classmethod
    def load(cls, path: str, device) -> "PackedArena":
        from safetensors import safe_open
        with safe_open(path, framework="pt", device="cpu") as f:
            meta = f.metadata()
            if meta.get("format") != "ternary-packed-arena-v1":
                raise ValueError(f"{path} is not a packed arena")
            entries = [ArenaEntry(**d) for d in json.loads(meta["index"])]
            return cls(f.get_tensor("packed").to(device), f.get_tensor("scales").to(device), f.get_tensor("biases").to(device), entries)

    # -- multi-GPU ---------------------------------------------------------------------------------
    def broadcast(self, src: int = 0, group=None):
        """Replicate the arena from rank ``src`` to every rank (three collectives for ALL layers; RCCL
        when the tensors are on GPUs).  Every rank must hold an arena with the same index."""
        import torch.distributed as dist
        for t in (self.packed, self.scales, self.biases):
            dist.broadcast(t, src=src, group=group)

    @classmethod
    def replicate(cls, arena: Optional["PackedArena"], device, src: int = 0, group=None) -> "PackedArena":
        """Every rank ends up with rank ``src``'s arena: the index travels as a small object, then the packed bytes,
        scales and biases as three collectives (RCCL for GPU tensors, gloo for CPU tensors).  ``arena`` is only read on
        rank ``src``; the other ranks pass ``None``.  This is the one exchange step of the row-sharded multi-GPU mode:
        weights are quantized and packed once, on one rank, and 2 bits per weight go over the links."""
        import torch.distributed as dist
        is_src = dist.get_rank(group) == src
        if is_src and arena is None:
            raise ValueError("PackedArena.replicate: the source rank must pass its arena")
        box = [[e.__dict__ for e in arena.entries.values()] if is_src else None]
        dist.broadcast_object_list(box, src=src, group=group)
        entries = [ArenaEntry(**d) for d in box[0]]
        out = arena if is_src else cls.empty_like_index(entries, device)
        out.broadcast(src=src, group=group)
        return out

    @classmethod
    def empty_like_index(cls, entries: List[ArenaEntry], device) -> "PackedArena":
        total = max((e.offset + e.n * e.ld for e in entries), default=0)
        nvec = max((e.vec_offset + e.n for e in entries), default=0)
        return cls(torch.empty((total,), dtype=torch.uint8, device=device), torch.empty((nvec,), dtype=torch.float32, device=device),
                   torch.empty((nvec,), dtype=torch.float32, device=device), entries)


def shard_rows(m: int, world: int, rank: int) -> Tuple[int, int]:
    """Row range ``[begin, end)`` of rank ``rank`` when ``m`` activation rows are split as evenly as
    possible over ``world`` ranks (first ``m % world`` ranks get one row more)."""
    base, rem = divmod(m, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)
