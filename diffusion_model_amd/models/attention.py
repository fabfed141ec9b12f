"""Self- and cross-attention (reference: src/models/attention.py:10-79)."""
import torch
from torch import nn

from .. import runtime as rt
from ._common import engine_for


class SelfAttention(nn.Module):
    """Multi-head self-attention with a fused 3*d input projection; x (B, S, d_embed)."""

    def __init__(self, n_heads, d_embed, in_proj_bias=True, out_proj_bias=True):
        super().__init__()
        self.in_proj = nn.Linear(d_embed, 3 * d_embed, bias=in_proj_bias)
        self.out_proj = nn.Linear(d_embed, d_embed, bias=out_proj_bias)
        self.n_heads = n_heads
        self.d_head = d_embed // n_heads

    def forward(self, x, causal_mask=False):
        eng = engine_for(self, x)
        pk = rt.cached_pack(self, eng, rt.pack_self_attention)
        b, s, c = x.shape
        eng.reset()
        rows = eng.to_rows(x)
        out = rt.run_self_attention(eng, pk, rows, b, s, causal=bool(causal_mask))
        return eng.from_rows(out, (b, s, c)).to(x.dtype)


class CrossAttention(nn.Module):
    """Queries from x (B, S, d_embed); keys/values from y (B, S_kv, d_cross)."""

    def __init__(self, n_heads, d_embed, d_cross, in_proj_bias=True, out_proj_bias=True):
        super().__init__()
        self.q_proj = nn.Linear(d_embed, d_embed, bias=in_proj_bias)
        self.k_proj = nn.Linear(d_cross, d_embed, bias=in_proj_bias)
        self.v_proj = nn.Linear(d_cross, d_embed, bias=in_proj_bias)
        self.out_proj = nn.Linear(d_embed, d_embed, bias=out_proj_bias)
        self.n_heads = n_heads
        self.d_head = d_embed // n_heads

    def forward(self, x, y):
        eng = engine_for(self, x)
        pk = rt.cached_pack(self, eng, rt.pack_cross_attention)
        b, s, c = x.shape
        sk = y.shape[1]
        eng.reset()
        kv = rt.context_kv(eng, pk, eng.to_rows(y), b, sk)
        out = rt.run_cross_attention(eng, pk, eng.to_rows(x), kv, b, s, sk)
        return eng.from_rows(out, (b, s, c)).to(x.dtype)
