"""U-Net building blocks (reference: src/models/blocks.py:13-159)."""
import torch
from torch import nn

from .. import runtime as rt
from ._common import engine_for
from .attention import CrossAttention, SelfAttention


class ResidualBlock(nn.Module):
    """GN+SiLU -> conv3x3 -> + time projection -> GN+SiLU -> conv3x3 -> + (identity | conv1x1)(input).

    feature (B, C_in, H, W), time (B, n_time) -> (B, C_out, H, W).
    """

    def __init__(self, in_channels: int, out_channels: int, n_time: int = 1280):
        super().__init__()
        self.groupnorm_feature = nn.GroupNorm(32, in_channels)
        self.conv_feature = nn.Conv2d(in_channels, out_channels, kernel_size=3, padding=1)
        self.linear_time = nn.Linear(n_time, out_channels)
        self.groupnorm_merged = nn.GroupNorm(32, out_channels)
        self.conv_merged = nn.Conv2d(out_channels, out_channels, kernel_size=3, padding=1)
        if in_channels == out_channels:
            self.residual_layer = nn.Identity()
        else:
            self.residual_layer = nn.Conv2d(in_channels, out_channels, kernel_size=1, padding=0)

    def forward(self, feature: torch.Tensor, time: torch.Tensor) -> torch.Tensor:
        eng = engine_for(self, feature)
        pk = rt.cached_pack(self, eng, rt.pack_residual_block)
        b, c, h, w = feature.shape
        eng.reset()
        tproj = eng.linear_rows(time.float().contiguous(), pk.lt_w, pk.lt_b, silu_in=True)
        x = eng.nchw_to_rows(feature)
        out = rt.run_residual_block(eng, pk, x, b, h, w, tproj)
        return eng.rows_to_nchw(out, b, pk.cout, h, w).to(feature.dtype)


class AttentionBlock(nn.Module):
    """GN -> conv1x1 -> [LN, self-attn, +] -> [LN, cross-attn(context), +] -> [LN, GEGLU FFN, +] -> conv1x1 -> + input.

    x (B, n_heads*n_embed, H, W), context (B, S, d_context) -> same shape as x.
    """

    def __init__(self, n_heads: int, n_embed: int, d_context: int = 768):
        super().__init__()
        channels = n_heads * n_embed
        self.group_norm = nn.GroupNorm(32, channels, eps=1e-6)
        self.conv_input = nn.Conv2d(channels, channels, kernel_size=1, padding=0)
        self.layer_norm_1 = nn.LayerNorm(channels)
        self.attention_1 = SelfAttention(n_heads, channels, in_proj_bias=False)
        self.layer_norm_2 = nn.LayerNorm(channels)
        self.attention_2 = CrossAttention(n_heads, channels, d_context, in_proj_bias=False)
        self.layer_norm_3 = nn.LayerNorm(channels)
        self.linear_geglu_1 = nn.Linear(channels, 4 * channels * 2)
        self.linear_geglu_2 = nn.Linear(4 * channels, channels)
        self.conv_output = nn.Conv2d(channels, channels, kernel_size=1, padding=0)

    def forward(self, x: torch.Tensor, context: torch.Tensor) -> torch.Tensor:
        eng = engine_for(self, x)
        pk = rt.cached_pack(self, eng, rt.pack_attention_block)
        b, c, h, w = x.shape
        sk = context.shape[1]
        eng.reset()
        kv = rt.context_kv(eng, pk.ca, eng.to_rows(context), b, sk)
        rows = eng.nchw_to_rows(x)
        out = rt.run_attention_block(eng, pk, rows, b, h, w, kv, sk)
        return eng.rows_to_nchw(out, b, c, h, w).to(x.dtype)
