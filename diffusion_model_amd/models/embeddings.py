"""Time-embedding MLP (reference: src/models/embeddings.py:13-45)."""
import torch
from torch import nn

from .. import runtime as rt
from ._common import engine_for


class TimeEmbedding(nn.Module):
    """Linear(n, 4n) -> SiLU -> Linear(4n, 4n); input (B, n_embed), output (B, 4*n_embed)."""

    def __init__(self, n_embed: int = 320):
        super().__init__()
        self.linear_1 = nn.Linear(n_embed, 4 * n_embed)
        self.linear_2 = nn.Linear(4 * n_embed, 4 * n_embed)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        eng = engine_for(self, x)
        pk = rt.cached_pack(self, eng, rt.pack_time_embedding)
        eng.reset()
        t = eng.linear_rows(x.float().contiguous(), pk.w1, pk.b1, silu_in=False)
        out = torch.empty(x.shape[0], pk.w2.shape[0], dtype=torch.float32, device=x.device)
        eng.linear_rows(t, pk.w2, pk.b2, silu_in=True, out=out)
        return out.to(x.dtype)
