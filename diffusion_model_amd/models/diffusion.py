"""Noise predictor eps(x_t, context, t) (reference: src/models/diffusion.py:15-101)."""
import weakref

import torch
from torch import nn

from .. import runtime as rt
from ._common import engine_for
from .embeddings import TimeEmbedding
from .unet import UNet


class FinalLayer(nn.Module):
    """GroupNorm(32) + SiLU + conv3x3 from U-Net features to latent channels."""

    def __init__(self, in_channels: int = 320, out_channels: int = 4):
        super().__init__()
        self.group_norm = nn.GroupNorm(32, in_channels)
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=3, padding=1)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        eng = engine_for(self, x)
        pk = rt.cached_pack(self, eng, rt.pack_final_layer)
        b, c, h, w = x.shape
        eng.reset()
        a = eng.groupnorm(eng.nchw_to_rows(x), b, h * w, pk.gn_w, pk.gn_b, pk.gn_eps, silu=True)
        out = torch.empty(b, pk.cout, h, w, dtype=torch.float32, device=x.device)
        eng.gemm(a, pk.conv, nchw_out=out, rows_per_batch=h * w,
                 conv=dict(batch=b, hin=h, win=w, hout=h, wout=w, cin_real=c))
        return out.to(x.dtype)


class Diffusion(nn.Module):
    """TimeEmbedding(320) -> UNet -> FinalLayer(320, 4).

    forward(latent (B,4,H,W), context (B,S,768), time (B,320)) -> predicted noise (B,4,H,W).
    Parameters in float32 select the exact-fp32 MFMA kernels, ``.bfloat16()`` the bf16 MFMA kernels
    (fp32 accumulation and statistics).  Runs on the HIP device only.
    """

    def __init__(self):
        super().__init__()
        self.time_embedding = TimeEmbedding(320)
        self.unet = UNet()
        self.final = FinalLayer(320, 4)

    def hip_pack(self, device=None):
        """(engine, packed weights) for the current parameters; packs on first use."""
        p = next(self.parameters())
        eng = rt.Engine.get(device or p.device, p.dtype)
        return eng, rt.cached_pack(self, eng, rt.pack_diffusion)

    def forward(self, latent: torch.Tensor, context: torch.Tensor, time: torch.Tensor) -> torch.Tensor:
        eng = engine_for(self, latent)
        pk = rt.cached_pack(self, eng, rt.pack_diffusion)
        eng.reset()
        kvs, sk = self._context_kv(eng, pk, context)
        out = torch.empty(latent.shape[0], pk.final.cout, latent.shape[2], latent.shape[3], dtype=torch.float32,
                          device=latent.device)
        rt.diffusion_forward_into(eng, pk, latent.float().contiguous(), time.float().contiguous(), kvs, sk, out)
        return out.to(latent.dtype)

    def _context_kv(self, eng, pk, context: torch.Tensor):
        """Cross-attention K / V of all 16 AttentionBlocks for ``context``.  They depend on neither the latent nor
        the timestep, so a loop that passes the SAME tensor object every step (and has not written to it: torch's
        version counter) gets them from the previous call instead of 32 projection launches."""
        c = self.__dict__.get("_dm_ctx_cache")
        key = (id(eng), id(pk), context._version, tuple(context.shape), context.dtype, context.data_ptr())
        if c is not None and c["ref"]() is context and c["key"] == key:
            return c["kvs"], context.shape[1]
        reuse = c["kvs"] if c is not None and c["key"][:2] == key[:2] and c["key"][3] == key[3] else None
        kvs, sk = rt.precompute_context(eng, pk.unet, context, persistent=reuse is None, out=reuse)
        self.__dict__["_dm_ctx_cache"] = dict(ref=weakref.ref(context), key=key, kvs=kvs)
        return kvs, sk
