"""CLIP text encoder (reference: src/models/clip.py:15-202): producer of the cross-attention context."""
import torch
from torch import nn

from .. import runtime as rt
from ._common import engine_for
from .attention import SelfAttention


class CLIPEmbedding(nn.Module):
    """Token embedding + learned positional embedding; tokens (B, S) -> (B, S, n_embed)."""

    def __init__(self, n_vocab: int = 49408, n_embed: int = 768, n_tokens: int = 77):
        super().__init__()
        self.token_embedding = nn.Embedding(n_vocab, n_embed)
        self.position_embedding = nn.Parameter(torch.zeros(n_tokens, n_embed))

    def forward(self, tokens: torch.LongTensor) -> torch.Tensor:
        eng = engine_for(self, tokens)
        _check_tokens(tokens, self.token_embedding.num_embeddings)
        eng.reset()
        b, s = tokens.shape
        rows = eng.embed_tokens(tokens, rt._f32(eng, self.token_embedding.weight), rt._f32(eng, self.position_embedding))
        return eng.from_rows(rows, (b, s, self.token_embedding.embedding_dim)).to(self.position_embedding.dtype)


class CLIPLayer(nn.Module):
    """LN -> causal self-attention -> + ; LN -> Linear(4x) -> QuickGELU -> Linear -> + ; x (B, S, n_embed)."""

    def __init__(self, n_heads: int, n_embed: int):
        super().__init__()
        self.layernorm_1 = nn.LayerNorm(n_embed)
        self.attention = SelfAttention(n_heads, n_embed)
        self.layernorm_2 = nn.LayerNorm(n_embed)
        self.linear_1 = nn.Linear(n_embed, 4 * n_embed)
        self.linear_2 = nn.Linear(4 * n_embed, n_embed)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        eng = engine_for(self, x)
        pk = rt.cached_pack(self, eng, rt.pack_clip_layer)
        b, s, c = x.shape
        eng.reset()
        rows = eng.to_rows(x)
        if rows.data_ptr() == x.data_ptr():  # the layer writes its result over its input rows
            rows = rows.clone()
        out = rt.run_clip_layer(eng, pk, rows, b, s)
        return eng.from_rows(out, (b, s, c)).to(x.dtype)


class CLIP(nn.Module):
    """12-layer, 12-head, 768-wide text transformer; tokens (B, 77) -> (B, 77, 768)."""

    def __init__(self, n_vocab: int = 49408, n_embed: int = 768, n_tokens: int = 77, n_layers: int = 12,
                 n_heads: int = 12):
        super().__init__()
        self.embedding = CLIPEmbedding(n_vocab, n_embed, n_tokens)
        self.layers = nn.ModuleList([CLIPLayer(n_heads, n_embed) for _ in range(n_layers)])
        self.layernorm = nn.LayerNorm(n_embed)

    def forward(self, tokens: torch.LongTensor) -> torch.Tensor:
        eng = engine_for(self, tokens)
        tokens = tokens.type(torch.long)
        _check_tokens(tokens, self.embedding.token_embedding.num_embeddings)
        pk = rt.cached_pack(self, eng, rt.pack_clip)
        eng.reset()
        b, s = tokens.shape
        out = rt.run_clip(eng, pk, tokens)
        return eng.from_rows(out, (b, s, self.layernorm.normalized_shape[0])).to(self.layernorm.weight.dtype)

    def encode_text(self, text_tokens: torch.LongTensor) -> torch.Tensor:
        return self.forward(text_tokens)


def _check_tokens(tokens: torch.Tensor, n_vocab: int) -> None:
    """nn.Embedding raises IndexError for ids outside the table; keep that behaviour."""
    if tokens.dim() != 2:
        raise RuntimeError(f"tokens must be (batch, sequence), got {tuple(tokens.shape)}")
    lo, hi = int(tokens.min()), int(tokens.max())
    if lo < 0 or hi >= n_vocab:
        raise IndexError(f"token id out of range [0, {n_vocab}): min {lo}, max {hi}")
