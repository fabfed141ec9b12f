"""U-Net noise predictor body (reference: src/models/unet.py:14-175)."""
import torch
from torch import nn

from .. import runtime as rt
from ._common import engine_for
from .blocks import AttentionBlock, ResidualBlock


class Upsample(nn.Module):
    """Nearest-neighbour x2 followed by conv3x3; the kernel folds the upsampling into the conv's gather."""

    def __init__(self, channels: int):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, kernel_size=3, padding=1)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        eng = engine_for(self, x)
        lp = rt.cached_pack(self, eng, lambda m, e: rt.pack_conv_layer(m.conv, e, up=True))
        b, c, h, w = x.shape
        eng.reset()
        out, ho, wo = rt.run_conv_layer(eng, lp, eng.nchw_to_rows(x), b, h, w)
        return eng.rows_to_nchw(out, b, lp.cout, ho, wo).to(x.dtype)


class SwitchSequential(nn.Sequential):
    """Sequential that routes (x, context) to AttentionBlocks, (x, time) to ResidualBlocks and x to the rest."""

    def forward(self, x: torch.Tensor, context: torch.Tensor, time: torch.Tensor) -> torch.Tensor:
        for layer in self:
            if isinstance(layer, AttentionBlock):
                x = layer(x, context)
            elif isinstance(layer, ResidualBlock):
                x = layer(x, time)
            elif isinstance(layer, nn.Conv2d):
                x = _conv2d_forward(layer, x)
            else:
                x = layer(x)
        return x


def _conv2d_forward(conv: nn.Conv2d, x: torch.Tensor) -> torch.Tensor:
    eng = engine_for(conv, x)
    lp = rt.cached_pack(conv, eng, lambda m, e: rt._pack_stage([m], e)[0])
    b, c, h, w = x.shape
    eng.reset()
    out, ho, wo = rt.run_conv_layer(eng, lp, eng.nchw_to_rows(x, cpad=rt._rup(c, 8)), b, h, w)
    return eng.rows_to_nchw(out, b, lp.cout, ho, wo).to(x.dtype)


class UNet(nn.Module):
    """12 encoder stages, a 3-block bottleneck and 12 decoder stages with LIFO skip connections.

    latent (B, 4, H, W), context (B, S, 768), time (B, 1280) -> (B, 320, H, W); H and W divisible by 8.
    """

    def __init__(self):
        super().__init__()
        S, R, A = SwitchSequential, ResidualBlock, AttentionBlock
        self.encoders = nn.ModuleList([
            S(nn.Conv2d(4, 320, kernel_size=3, padding=1)),
            S(R(320, 320), A(8, 40)),
            S(R(320, 320), A(8, 40)),
            S(nn.Conv2d(320, 320, kernel_size=3, stride=2, padding=1)),
            S(R(320, 640), A(8, 80)),
            S(R(640, 640), A(8, 80)),
            S(nn.Conv2d(640, 640, kernel_size=3, stride=2, padding=1)),
            S(R(640, 1280), A(8, 160)),
            S(R(1280, 1280), A(8, 160)),
            S(nn.Conv2d(1280, 1280, kernel_size=3, stride=2, padding=1)),
            S(R(1280, 1280)),
            S(R(1280, 1280)),
        ])
        self.bottleneck = S(R(1280, 1280), A(8, 160), R(1280, 1280))
        self.decoders = nn.ModuleList([
            S(R(2560, 1280)),
            S(R(2560, 1280)),
            S(R(2560, 1280), Upsample(1280)),
            S(R(2560, 1280), A(8, 160)),
            S(R(2560, 1280), A(8, 160)),
            S(R(1920, 1280), A(8, 160), Upsample(1280)),
            S(R(1920, 640), A(8, 80)),
            S(R(1280, 640), A(8, 80)),
            S(R(960, 640), A(8, 80), Upsample(640)),
            S(R(960, 320), A(8, 40)),
            S(R(640, 320), A(8, 40)),
            S(R(640, 320), A(8, 40)),
        ])

    def forward(self, x: torch.Tensor, context: torch.Tensor, time: torch.Tensor) -> torch.Tensor:
        eng = engine_for(self, x)
        pk = rt.cached_pack(self, eng, rt.pack_unet)
        b, c, h, w = x.shape
        eng.reset()
        kvs, sk = rt.precompute_context(eng, pk, context)
        tproj = eng.linear_rows(time.float().contiguous(), pk.lt_w, pk.lt_b, silu_in=True)
        rows = eng.nchw_to_rows(x, cpad=rt._rup(c, 8))
        out, ho, wo = rt.run_unet(eng, pk, rows, b, h, w, tproj, kvs, sk)
        return eng.rows_to_nchw(out, b, out.shape[1], ho, wo).to(x.dtype)
