"""Functional fp32 restatement of the reference noise predictor (test oracle).

Everything works on a flat ``state`` dict with the reference's state_dict keys
(``unet.encoders.1.1.attention_1.in_proj.weight`` ...), NCHW fp32 tensors on
the CPU, and the reference's operator order, so results agree with the
reference to fp32 rounding.

Reference lines restated here (paths under /root/reference):
  * Diffusion.forward            src/models/diffusion.py:73-101
  * FinalLayer.forward           src/models/diffusion.py:34-47
  * TimeEmbedding.forward        src/models/embeddings.py:32-45
  * UNet layout / forward        src/models/unet.py:86-175
  * Upsample.forward             src/models/unet.py:31-34
  * ResidualBlock.forward        src/models/blocks.py:44-72
  * AttentionBlock.forward       src/models/blocks.py:114-159
  * SelfAttention.forward        src/models/attention.py:24-45
  * CrossAttention.forward       src/models/attention.py:58-79
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

State = Dict[str, torch.Tensor]

# (kind, *args) per stage entry; mirrors src/models/unet.py:90-146.
ENCODERS = [
    [("conv", 4, 320, 1)],
    [("res", 320, 320), ("attn", 8, 40)],
    [("res", 320, 320), ("attn", 8, 40)],
    [("conv", 320, 320, 2)],
    [("res", 320, 640), ("attn", 8, 80)],
    [("res", 640, 640), ("attn", 8, 80)],
    [("conv", 640, 640, 2)],
    [("res", 640, 1280), ("attn", 8, 160)],
    [("res", 1280, 1280), ("attn", 8, 160)],
    [("conv", 1280, 1280, 2)],
    [("res", 1280, 1280)],
    [("res", 1280, 1280)],
]
BOTTLENECK = [("res", 1280, 1280), ("attn", 8, 160), ("res", 1280, 1280)]
DECODERS = [
    [("res", 2560, 1280)],
    [("res", 2560, 1280)],
    [("res", 2560, 1280), ("up", 1280)],
    [("res", 2560, 1280), ("attn", 8, 160)],
    [("res", 2560, 1280), ("attn", 8, 160)],
    [("res", 1920, 1280), ("attn", 8, 160), ("up", 1280)],
    [("res", 1920, 640), ("attn", 8, 80)],
    [("res", 1280, 640), ("attn", 8, 80)],
    [("res", 960, 640), ("attn", 8, 80), ("up", 640)],
    [("res", 960, 320), ("attn", 8, 40)],
    [("res", 640, 320), ("attn", 8, 40)],
    [("res", 640, 320), ("attn", 8, 40)],
]


# ----------------------------------------------------------------------------
# parameter shapes (so tests can build deterministic states without nn.Modules)
# ----------------------------------------------------------------------------
def residual_block_shapes(cin: int, cout: int, n_time: int = 1280) -> List[Tuple[str, Tuple[int, ...]]]:
    s = [
        ("groupnorm_feature.weight", (cin,)), ("groupnorm_feature.bias", (cin,)),
        ("conv_feature.weight", (cout, cin, 3, 3)), ("conv_feature.bias", (cout,)),
        ("linear_time.weight", (cout, n_time)), ("linear_time.bias", (cout,)),
        ("groupnorm_merged.weight", (cout,)), ("groupnorm_merged.bias", (cout,)),
        ("conv_merged.weight", (cout, cout, 3, 3)), ("conv_merged.bias", (cout,)),
    ]
    if cin != cout:
        s += [("residual_layer.weight", (cout, cin, 1, 1)), ("residual_layer.bias", (cout,))]
    return s


def attention_block_shapes(n_heads: int, n_embed: int, d_context: int = 768):
    c = n_heads * n_embed
    return [
        ("group_norm.weight", (c,)), ("group_norm.bias", (c,)),
        ("conv_input.weight", (c, c, 1, 1)), ("conv_input.bias", (c,)),
        ("layer_norm_1.weight", (c,)), ("layer_norm_1.bias", (c,)),
        ("attention_1.in_proj.weight", (3 * c, c)),
        ("attention_1.out_proj.weight", (c, c)), ("attention_1.out_proj.bias", (c,)),
        ("layer_norm_2.weight", (c,)), ("layer_norm_2.bias", (c,)),
        ("attention_2.q_proj.weight", (c, c)),
        ("attention_2.k_proj.weight", (c, d_context)),
        ("attention_2.v_proj.weight", (c, d_context)),
        ("attention_2.out_proj.weight", (c, c)), ("attention_2.out_proj.bias", (c,)),
        ("layer_norm_3.weight", (c,)), ("layer_norm_3.bias", (c,)),
        ("linear_geglu_1.weight", (8 * c, c)), ("linear_geglu_1.bias", (8 * c,)),
        ("linear_geglu_2.weight", (c, 4 * c)), ("linear_geglu_2.bias", (c,)),
        ("conv_output.weight", (c, c, 1, 1)), ("conv_output.bias", (c,)),
    ]


def _entry_shapes(entry):
    kind = entry[0]
    if kind == "conv":
        return [("weight", (entry[2], entry[1], 3, 3)), ("bias", (entry[2],))]
    if kind == "res":
        return residual_block_shapes(entry[1], entry[2])
    if kind == "attn":
        return attention_block_shapes(entry[1], entry[2])
    if kind == "up":
        return [("conv.weight", (entry[1], entry[1], 3, 3)), ("conv.bias", (entry[1],))]
    raise ValueError(kind)


def diffusion_shapes() -> List[Tuple[str, Tuple[int, ...]]]:
    """(name, shape) for all 654 tensors of the reference ``Diffusion`` state_dict, in order."""
    out = [
        ("time_embedding.linear_1.weight", (1280, 320)), ("time_embedding.linear_1.bias", (1280,)),
        ("time_embedding.linear_2.weight", (1280, 1280)), ("time_embedding.linear_2.bias", (1280,)),
    ]
    for i, stage in enumerate(ENCODERS):
        for j, e in enumerate(stage):
            out += [(f"unet.encoders.{i}.{j}.{n}", s) for n, s in _entry_shapes(e)]
    for j, e in enumerate(BOTTLENECK):
        out += [(f"unet.bottleneck.{j}.{n}", s) for n, s in _entry_shapes(e)]
    for i, stage in enumerate(DECODERS):
        for j, e in enumerate(stage):
            out += [(f"unet.decoders.{i}.{j}.{n}", s) for n, s in _entry_shapes(e)]
    out += [("final.group_norm.weight", (320,)), ("final.group_norm.bias", (320,)),
            ("final.conv.weight", (4, 320, 3, 3)), ("final.conv.bias", (4,))]
    return out


# ----------------------------------------------------------------------------
# blocks
# ----------------------------------------------------------------------------
def _sub(state: State, prefix: str) -> State:
    return state if not prefix else _Prefixed(state, prefix)


class _Prefixed(dict):
    """Read-only view of ``state`` with ``prefix`` prepended to every lookup."""

    def __init__(self, state, prefix):
        super().__init__()
        self._s, self._p = state, prefix

    def __getitem__(self, k):
        return self._s[self._p + k]

    def __contains__(self, k):
        return (self._p + k) in self._s

    def get(self, k, default=None):
        return self._s.get(self._p + k, default)


def time_embedding(w: State, t: torch.Tensor) -> torch.Tensor:
    h = F.linear(t, w["linear_1.weight"], w["linear_1.bias"])
    return F.linear(F.silu(h), w["linear_2.weight"], w["linear_2.bias"])


def residual_block(w: State, x: torch.Tensor, time: torch.Tensor) -> torch.Tensor:
    h = F.group_norm(x, 32, w["groupnorm_feature.weight"], w["groupnorm_feature.bias"], eps=1e-5)
    h = F.conv2d(F.silu(h), w["conv_feature.weight"], w["conv_feature.bias"], padding=1)
    tp = F.linear(F.silu(time), w["linear_time.weight"], w["linear_time.bias"])
    h = h + tp[:, :, None, None]
    h = F.group_norm(h, 32, w["groupnorm_merged.weight"], w["groupnorm_merged.bias"], eps=1e-5)
    h = F.conv2d(F.silu(h), w["conv_merged.weight"], w["conv_merged.bias"], padding=1)
    if "residual_layer.weight" in w:
        x = F.conv2d(x, w["residual_layer.weight"], w["residual_layer.bias"])
    return h + x


def _heads(t: torch.Tensor, n_heads: int) -> torch.Tensor:
    b, s, c = t.shape
    return t.reshape(b, s, n_heads, c // n_heads).transpose(1, 2)


def attention_core(q, k, v, n_heads: int, causal: bool = False) -> torch.Tensor:
    """softmax(q k^T / sqrt(d)) v per head; q (B,Sq,C), k/v (B,Sk,C) -> (B,Sq,C)."""
    b, sq, c = q.shape
    qh, kh, vh = _heads(q, n_heads), _heads(k, n_heads), _heads(v, n_heads)
    wgt = qh @ kh.transpose(-1, -2)
    if causal:
        mask = torch.ones_like(wgt, dtype=torch.bool).triu(1)
        wgt = wgt.masked_fill(mask, -torch.inf)
    wgt = wgt / math.sqrt(c // n_heads)
    out = torch.softmax(wgt, dim=-1) @ vh
    return out.transpose(1, 2).reshape(b, sq, c)


def self_attention(w: State, x: torch.Tensor, n_heads: int, causal: bool = False) -> torch.Tensor:
    q, k, v = F.linear(x, w["in_proj.weight"], w.get("in_proj.bias")).chunk(3, dim=-1)
    o = attention_core(q, k, v, n_heads, causal)
    return F.linear(o, w["out_proj.weight"], w.get("out_proj.bias"))


def cross_attention(w: State, x: torch.Tensor, y: torch.Tensor, n_heads: int) -> torch.Tensor:
    q = F.linear(x, w["q_proj.weight"], w.get("q_proj.bias"))
    k = F.linear(y, w["k_proj.weight"], w.get("k_proj.bias"))
    v = F.linear(y, w["v_proj.weight"], w.get("v_proj.bias"))
    o = attention_core(q, k, v, n_heads)
    return F.linear(o, w["out_proj.weight"], w.get("out_proj.bias"))


def attention_block(w: State, x: torch.Tensor, context: torch.Tensor, n_heads: int) -> torch.Tensor:
    b, c, hh, ww = x.shape
    h = F.group_norm(x, 32, w["group_norm.weight"], w["group_norm.bias"], eps=1e-6)
    h = F.conv2d(h, w["conv_input.weight"], w["conv_input.bias"])
    h = h.reshape(b, c, hh * ww).transpose(1, 2)
    n1 = F.layer_norm(h, (c,), w["layer_norm_1.weight"], w["layer_norm_1.bias"])
    h = h + self_attention(_sub(w, "attention_1."), n1, n_heads)
    n2 = F.layer_norm(h, (c,), w["layer_norm_2.weight"], w["layer_norm_2.bias"])
    h = h + cross_attention(_sub(w, "attention_2."), n2, context, n_heads)
    n3 = F.layer_norm(h, (c,), w["layer_norm_3.weight"], w["layer_norm_3.bias"])
    a, gate = F.linear(n3, w["linear_geglu_1.weight"], w["linear_geglu_1.bias"]).chunk(2, dim=-1)
    h = h + F.linear(a * F.gelu(gate), w["linear_geglu_2.weight"], w["linear_geglu_2.bias"])
    h = h.transpose(1, 2).reshape(b, c, hh, ww)
    return F.conv2d(h, w["conv_output.weight"], w["conv_output.bias"]) + x


def upsample(w: State, x: torch.Tensor) -> torch.Tensor:
    x = F.interpolate(x, scale_factor=2, mode="nearest")
    return F.conv2d(x, w["conv.weight"], w["conv.bias"], padding=1)


def final_layer(w: State, x: torch.Tensor) -> torch.Tensor:
    h = F.group_norm(x, 32, w["group_norm.weight"], w["group_norm.bias"], eps=1e-5)
    return F.conv2d(F.silu(h), w["conv.weight"], w["conv.bias"], padding=1)


def _stage(state: State, prefix: str, entries, x, context, time):
    for j, e in enumerate(entries):
        w = _sub(state, f"{prefix}{j}.")
        if e[0] == "conv":
            x = F.conv2d(x, w["weight"], w["bias"], stride=e[3], padding=1)
        elif e[0] == "res":
            x = residual_block(w, x, time)
        elif e[0] == "attn":
            x = attention_block(w, x, context, e[1])
        else:
            x = upsample(w, x)
    return x


def unet(state: State, x, context, time, prefix: str = "unet.") -> torch.Tensor:
    skips = []
    for i, stage in enumerate(ENCODERS):
        x = _stage(state, f"{prefix}encoders.{i}.", stage, x, context, time)
        skips.append(x)
    x = _stage(state, f"{prefix}bottleneck.", BOTTLENECK, x, context, time)
    for i, stage in enumerate(DECODERS):
        x = torch.cat((x, skips.pop()), dim=1)
        x = _stage(state, f"{prefix}decoders.{i}.", stage, x, context, time)
    return x


@torch.no_grad()
def diffusion(state: State, latent, context, time) -> torch.Tensor:
    """eps = Diffusion(latent (B,4,H,W), context (B,77,768), time (B,320))."""
    t = time_embedding(_sub(state, "time_embedding."), time)
    return final_layer(_sub(state, "final."), unet(state, latent, context, t))


class OracleDiffusion:
    """Callable with the reference's ``model(sample, context, time_emb)`` protocol."""

    def __init__(self, state: State):
        self.state = state

    def __call__(self, sample, context, time_emb):
        return diffusion(self.state, sample, context, time_emb)
