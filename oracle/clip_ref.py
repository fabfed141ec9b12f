"""Functional fp32 restatement of the reference CLIP text encoder (test oracle).

Works on a flat ``state`` dict with the reference's state_dict keys
(``embedding.token_embedding.weight``, ``layers.3.attention.in_proj.weight`` ...) on the CPU, in the
reference's operator order.  Test infrastructure only: nothing in the product path imports it.

Reference lines restated here (paths under /root/reference):
  * CLIPEmbedding.forward   src/models/clip.py:42-59
  * CLIPLayer.forward       src/models/clip.py:91-119   (QuickGELU: x * sigmoid(1.702 x), :114)
  * CLIP.forward            src/models/clip.py:167-194
  * SelfAttention.forward   src/models/attention.py:24-45 with causal_mask=True (:35-37)
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

State = Dict[str, torch.Tensor]


def clip_layer_shapes(n_embed: int) -> List[Tuple[str, Tuple[int, ...]]]:
    e = n_embed
    return [
        ("layernorm_1.weight", (e,)), ("layernorm_1.bias", (e,)),
        ("attention.in_proj.weight", (3 * e, e)), ("attention.in_proj.bias", (3 * e,)),
        ("attention.out_proj.weight", (e, e)), ("attention.out_proj.bias", (e,)),
        ("layernorm_2.weight", (e,)), ("layernorm_2.bias", (e,)),
        ("linear_1.weight", (4 * e, e)), ("linear_1.bias", (4 * e,)),
        ("linear_2.weight", (e, 4 * e)), ("linear_2.bias", (e,)),
    ]


def clip_shapes(n_vocab: int = 49408, n_embed: int = 768, n_tokens: int = 77, n_layers: int = 12):
    # state_dict order: a module's own parameters come before those of its children
    s = [("embedding.position_embedding", (n_tokens, n_embed)), ("embedding.token_embedding.weight", (n_vocab, n_embed))]
    for i in range(n_layers):
        s += [(f"layers.{i}.{n}", shp) for n, shp in clip_layer_shapes(n_embed)]
    s += [("layernorm.weight", (n_embed,)), ("layernorm.bias", (n_embed,))]
    return s


def causal_self_attention(state: State, prefix: str, x: torch.Tensor, n_heads: int) -> torch.Tensor:
    b, s, e = x.shape
    d = e // n_heads
    q, k, v = F.linear(x, state[prefix + "in_proj.weight"], state[prefix + "in_proj.bias"]).chunk(3, dim=-1)
    q, k, v = (t.view(b, s, n_heads, d).transpose(1, 2) for t in (q, k, v))
    w = q @ k.transpose(-1, -2)
    w = w.masked_fill(torch.ones_like(w, dtype=torch.bool).triu(1), -torch.inf)
    w = w / math.sqrt(d)
    w = F.softmax(w, dim=-1)
    o = (w @ v).transpose(1, 2).reshape(b, s, e)
    return F.linear(o, state[prefix + "out_proj.weight"], state[prefix + "out_proj.bias"])


def clip_layer(state: State, prefix: str, x: torch.Tensor, n_heads: int) -> torch.Tensor:
    e = x.shape[-1]
    residue = x
    x = F.layer_norm(x, (e,), state[prefix + "layernorm_1.weight"], state[prefix + "layernorm_1.bias"])
    x = causal_self_attention(state, prefix + "attention.", x, n_heads) + residue
    residue = x
    x = F.layer_norm(x, (e,), state[prefix + "layernorm_2.weight"], state[prefix + "layernorm_2.bias"])
    x = F.linear(x, state[prefix + "linear_1.weight"], state[prefix + "linear_1.bias"])
    x = x * torch.sigmoid(1.702 * x)
    x = F.linear(x, state[prefix + "linear_2.weight"], state[prefix + "linear_2.bias"])
    return x + residue


def clip_embedding(state: State, tokens: torch.Tensor) -> torch.Tensor:
    return F.embedding(tokens, state["embedding.token_embedding.weight"]) + state["embedding.position_embedding"]


def clip(state: State, tokens: torch.Tensor, n_heads: int = 12) -> torch.Tensor:
    n_layers = 1 + max(int(k.split(".")[1]) for k in state if k.startswith("layers."))
    x = clip_embedding(state, tokens.long())
    for i in range(n_layers):
        x = clip_layer(state, f"layers.{i}.", x, n_heads)
    e = x.shape[-1]
    return F.layer_norm(x, (e,), state["layernorm.weight"], state["layernorm.bias"])
