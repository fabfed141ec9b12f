"""VAE decoder and encoder for the latent <-> image ends of the pipeline.

The reference has no decoder at all (scripts/model_loader.py:9 imports a module that does not exist;
scripts/inference.py:325-332 is commented out) and its ``Encoder`` (src/models/encoder.py:103-242) cannot run: it puts
the U-Net's ``ResidualBlock`` - whose forward needs a time argument - into an ``nn.Sequential``.  Both classes here are
therefore this build's own design, PARITY UNPINNED: the Stable-Diffusion-1.5 autoencoder laid out as the mirror of
src/models/encoder.py:107-152 (residual blocks 512 / 256 / 128 without a time input, one single-head attention block at
latent resolution, GroupNorm(32) + SiLU, nearest x2 upsampling where the encoder strides, the 0.18215 latent scaling).
``Encoder`` keeps the reference's constructor, ``forward(x, noise)`` / ``encode_without_sampling(x)`` signatures,
``scaling_factor`` and its asymmetric (0, 1, 0, 1) padding in front of the stride-2 convolutions.
They are checked against ``oracle/vae_ref.py`` (the same design stated in plain torch), not against the reference.
"""
import torch
from torch import nn

from .. import _lib, runtime as rt
from .._lib import check
from ._common import engine_for
from .attention import SelfAttention
from .unet import Upsample

SCALING_FACTOR = 0.18215


class VAEResidualBlock(nn.Module):
    """GN(32) -> SiLU -> conv3x3 -> GN(32) -> SiLU -> conv3x3, plus the input (1x1 conv when channels change)."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        # parameter names of the reference's ResidualBlock (src/models/blocks.py:32-42), so that a state_dict of the
        # reference's Encoder loads into this one; its linear_time (the time projection its Encoder can never feed)
        # has no counterpart here and is dropped on load
        self.groupnorm_feature = nn.GroupNorm(32, in_channels)
        self.conv_feature = nn.Conv2d(in_channels, out_channels, kernel_size=3, padding=1)
        self.groupnorm_merged = nn.GroupNorm(32, out_channels)
        self.conv_merged = nn.Conv2d(out_channels, out_channels, kernel_size=3, padding=1)
        self.residual_layer = (nn.Identity() if in_channels == out_channels
                               else nn.Conv2d(in_channels, out_channels, kernel_size=1, padding=0))

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        for k in [k for k in state_dict if k.startswith(prefix + "linear_time.")]:
            state_dict.pop(k)
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return _run_single(self, x)


class VAEAttentionBlock(nn.Module):
    """GN(32) -> single-head self-attention over the H*W positions -> + input (mirror of encoder.py:20-69)."""

    def __init__(self, channels: int):
        super().__init__()
        self.groupnorm = nn.GroupNorm(32, channels)
        self.attention = SelfAttention(1, channels)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        # the reference's block holds an nn.MultiheadAttention (src/models/encoder.py:36-40), whose fused input
        # projection is a bare parameter pair: attention.in_proj_weight / in_proj_bias
        for old, new in (("attention.in_proj_weight", "attention.in_proj.weight"),
                         ("attention.in_proj_bias", "attention.in_proj.bias")):
            if prefix + old in state_dict:
                state_dict[prefix + new] = state_dict.pop(prefix + old)
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return _run_single(self, x)


def _run_single(layer: nn.Module, x: torch.Tensor) -> torch.Tensor:
    eng = engine_for(layer, x)
    ops = rt.cached_pack(layer, eng, lambda m, e: rt.pack_vae_layers([m], e))
    b, c, h, w = x.shape
    eng.reset()
    rows, c2, h2, w2 = rt.run_vae_layers(eng, ops, eng.nchw_to_rows(x, cpad=rt._rup(c, 8)), b, h, w)
    return eng.rows_to_nchw(rows, b, c2, h2, w2).to(x.dtype)


class Decoder(nn.Module):
    """latent (B, 4, h, w) -> image (B, 3, 8h, 8w) in about [-1, 1]."""

    def __init__(self):
        super().__init__()
        R, A = VAEResidualBlock, VAEAttentionBlock
        self.layers = nn.Sequential(
            nn.Conv2d(4, 4, kernel_size=1, padding=0),
            nn.Conv2d(4, 512, kernel_size=3, padding=1),
            R(512, 512), A(512), R(512, 512), R(512, 512), R(512, 512), R(512, 512),
            Upsample(512), R(512, 512), R(512, 512), R(512, 512),
            Upsample(512), R(512, 256), R(256, 256), R(256, 256),
            Upsample(256), R(256, 128), R(128, 128), R(128, 128),
            nn.GroupNorm(32, 128), nn.SiLU(),
            nn.Conv2d(128, 3, kernel_size=3, padding=1))
        self.scaling_factor = SCALING_FACTOR

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        eng = engine_for(self, x)
        ops = rt.cached_pack(self, eng, lambda m, e: rt.pack_vae_layers(list(m.layers), e))
        b, c, h, w = x.shape
        eng.reset()
        # x / 0.18215 as a true fp32 division, before anything is rounded to the engine dtype
        xin = x.float().contiguous()
        scaled = torch.empty_like(xin)
        check(_lib.lib().dm_rescale(xin.data_ptr(), scaled.data_ptr(), xin.numel(), _lib.DM_F32, 0.0,
                                    float(self.scaling_factor), 1.0, 0.0, 0, 0.0, 0.0, eng.stream), "rescale")
        rows = eng.nchw_to_rows(scaled, cpad=8)
        out = torch.empty(b, 4, 8 * h, 8 * w, dtype=torch.float32, device=x.device)  # fourth plane: zero weights
        rt.run_vae_layers(eng, ops, rows, b, h, w, nchw_out=out)
        return out[:, :3].contiguous().to(x.dtype)


class Encoder(nn.Module):
    """image (B, 3, H, W) in [-1, 1], noise (B, 4, H/8, W/8) -> latent (B, 4, H/8, W/8) (reference: encoder.py:103-212)."""

    def __init__(self):
        super().__init__()
        R, A = VAEResidualBlock, VAEAttentionBlock
        self.layers = nn.Sequential(
            nn.Conv2d(3, 128, kernel_size=3, padding=1),
            R(128, 128), R(128, 128),
            nn.Conv2d(128, 128, kernel_size=3, stride=2, padding=0),
            R(128, 256), R(256, 256),
            nn.Conv2d(256, 256, kernel_size=3, stride=2, padding=0),
            R(256, 512), R(512, 512),
            nn.Conv2d(512, 512, kernel_size=3, stride=2, padding=0),
            R(512, 512), R(512, 512), R(512, 512),
            A(512),
            R(512, 512),
            nn.GroupNorm(32, 512), nn.SiLU(),
            nn.Conv2d(512, 8, kernel_size=3, padding=1),
            nn.Conv2d(8, 8, kernel_size=1, padding=0))
        self.scaling_factor = SCALING_FACTOR

    def _moments(self, x: torch.Tensor) -> torch.Tensor:
        eng = engine_for(self, x)
        ops = rt.cached_pack(self, eng, lambda m, e: rt.pack_vae_layers(list(m.layers), e))
        b, c, h, w = x.shape
        if h % 8 or w % 8:
            raise RuntimeError("Encoder: image height and width must be multiples of 8")
        eng.reset()
        rows, c2, h2, w2 = rt.run_vae_layers(eng, ops, eng.nchw_to_rows(x, cpad=8), b, h, w)
        return eng.rows_to_nchw(rows, b, c2, h2, w2)  # fp32 (B, 8, H/8, W/8)

    def forward(self, x: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
        m = self._moments(x)
        b, _, h, w = m.shape
        nz = noise.float().contiguous()
        if tuple(nz.shape) != (b, 4, h, w):
            raise RuntimeError(f"Encoder: noise must be {(b, 4, h, w)}, got {tuple(nz.shape)}")
        out = torch.empty(b, 4, h, w, dtype=torch.float32, device=x.device)
        check(_lib.lib().dm_vae_sample(m.data_ptr(), nz.data_ptr(), out.data_ptr(), b, h * w, float(self.scaling_factor),
                                       torch.cuda.current_stream(x.device).cuda_stream), "vae_sample")
        return out.to(x.dtype)

    def encode_without_sampling(self, x: torch.Tensor):
        m = self._moments(x)
        mean, log_variance = torch.chunk(m, 2, dim=1)
        return (mean * self.scaling_factor).to(x.dtype), torch.clamp(log_variance, -30, 20).to(x.dtype)
