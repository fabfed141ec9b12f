"""Execution of the noise predictor on the HIP engine: weight packs and block runners.

The ``nn.Module`` classes in ``diffusion_model_amd.models`` only hold parameters (so state_dicts stay
interchangeable with the reference); their ``forward`` methods call the functions below, which
issue the kernel launches.  Mapping to the reference's forward passes:

  run_residual_block  <- ResidualBlock.forward   (src/models/blocks.py:44-72)
  run_attention_block <- AttentionBlock.forward  (src/models/blocks.py:114-159)
  run_unet            <- UNet.forward            (src/models/unet.py:148-175)
  diffusion_forward   <- Diffusion.forward       (src/models/diffusion.py:73-101)
  run_clip            <- CLIP.forward            (src/models/clip.py:167-194), CLIPLayer.forward (:91-119)
"""
from __future__ import annotations

from types import SimpleNamespace as NS
from typing import List, Optional

import torch

from .engine import Engine, _rup


# ----------------------------------------------------------------------------------------------
# packing
# ----------------------------------------------------------------------------------------------
def _f32(eng: Engine, p: torch.Tensor) -> torch.Tensor:
    return p.detach().to(device=eng.device, dtype=torch.float32).contiguous()


def module_dtype(module: torch.nn.Module) -> torch.dtype:
    for p in module.parameters():
        return p.dtype
    raise RuntimeError("module has no parameters")


def _signature(module: torch.nn.Module):
    return tuple((p.data_ptr(), p._version) for p in module.parameters())


def cached_pack(module: torch.nn.Module, eng: Engine, builder):
    """Pack once per (engine, parameter version); repack if any parameter was modified in place or moved."""
    cache = module.__dict__.setdefault("_dm_pack_cache", {})
    sig = (_signature(module), eng.split_bf16)  # the arithmetic mode decides the weight image
    hit = cache.get(id(eng))
    if hit is None or hit[0] != sig:
        hit = (sig, builder(module, eng))
        cache[id(eng)] = hit
    return hit[1]


def pack_residual_block(m, eng: Engine) -> NS:
    cin, cout = m.conv_feature.in_channels, m.conv_feature.out_channels
    # the 1x1 residual_layer rides in the second convolution as extra k (one launch, no skip tensor) when the
    # channel counts are whole k-tiles
    fold = (eng.fuse_skip and isinstance(m.residual_layer, torch.nn.Conv2d) and cin % eng.bk == 0
            and cout % eng.bk == 0)
    return NS(kind="res", cin=cin, cout=cout,
              gn1_w=_f32(eng, m.groupnorm_feature.weight), gn1_b=_f32(eng, m.groupnorm_feature.bias),
              gn1_eps=m.groupnorm_feature.eps,
              conv1=eng.pack_conv3x3(m.conv_feature, same_size=True),
              lt_w=_f32(eng, m.linear_time.weight), lt_b=_f32(eng, m.linear_time.bias),
              gn2_w=_f32(eng, m.groupnorm_merged.weight), gn2_b=_f32(eng, m.groupnorm_merged.bias),
              gn2_eps=m.groupnorm_merged.eps,
              conv2=eng.pack_conv3x3(m.conv_merged, skip=m.residual_layer if fold else None, same_size=True),
              res=None if (fold or isinstance(m.residual_layer, torch.nn.Identity)) else eng.pack_conv1x1(m.residual_layer),
              fold_skip=fold, t_off=0)


def pack_self_attention(m, eng: Engine, ln=None) -> NS:
    """``ln``: the LayerNorm in front of the attention (AttentionBlock only); folded into ``in_proj``."""
    return NS(heads=m.n_heads, d=m.d_head, in_proj=eng.pack_linear(m.in_proj, ln=ln), out_proj=eng.pack_linear(m.out_proj))


def pack_cross_attention(m, eng: Engine, ln=None) -> NS:
    return NS(heads=m.n_heads, d=m.d_head, q_proj=eng.pack_linear(m.q_proj, ln=ln), k_proj=eng.pack_linear(m.k_proj),
              v_proj=eng.pack_linear(m.v_proj), out_proj=eng.pack_linear(m.out_proj))


def pack_attention_block(m, eng: Engine) -> NS:
    c = m.conv_input.in_channels
    tail = eng.fuse_tail and c % eng.bk == 0
    return NS(kind="attn", c=c,
              gn_w=_f32(eng, m.group_norm.weight), gn_b=_f32(eng, m.group_norm.bias), gn_eps=m.group_norm.eps,
              conv_in=eng.pack_conv1x1(m.conv_input),
              ln1_w=_f32(eng, m.layer_norm_1.weight), ln1_b=_f32(eng, m.layer_norm_1.bias), ln1_eps=m.layer_norm_1.eps,
              fold=eng.fuse_ln,  # LayerNorms folded into in_proj / q_proj / linear_geglu_1
              sa=pack_self_attention(m.attention_1, eng, ln=m.layer_norm_1 if eng.fuse_ln else None),
              ln2_w=_f32(eng, m.layer_norm_2.weight), ln2_b=_f32(eng, m.layer_norm_2.bias), ln2_eps=m.layer_norm_2.eps,
              ca=pack_cross_attention(m.attention_2, eng, ln=m.layer_norm_2 if eng.fuse_ln else None),
              ln3_w=_f32(eng, m.layer_norm_3.weight), ln3_b=_f32(eng, m.layer_norm_3.bias), ln3_eps=m.layer_norm_3.eps,
              geglu1=eng.pack_linear(m.linear_geglu_1, geglu=True, ln=m.layer_norm_3 if eng.fuse_ln else None),
              geglu2=None if tail else eng.pack_linear(m.linear_geglu_2),
              conv_out=None if tail else eng.pack_conv1x1(m.conv_output),
              # linear_geglu_2 -> += residue -> conv_output has no non-linearity in between: one GEMM over the row
              # [gelu product | residue] with pre-multiplied weights
              tail=eng.pack_chain(m.linear_geglu_2, m.conv_output) if tail else None)


def pack_conv_layer(conv: torch.nn.Conv2d, eng: Engine, up: bool = False) -> NS:
    return NS(kind="up" if up else "conv", cin=conv.in_channels, cout=conv.out_channels, stride=conv.stride[0],
              w=eng.pack_conv3x3(conv, same_size=not up and conv.stride[0] == 1))


def pack_final_layer(m, eng: Engine) -> NS:
    return NS(gn_w=_f32(eng, m.group_norm.weight), gn_b=_f32(eng, m.group_norm.bias), gn_eps=m.group_norm.eps,
              cin=m.conv.in_channels, cout=m.conv.out_channels, conv=eng.pack_conv3x3(m.conv, same_size=True))


def pack_time_embedding(m, eng: Engine) -> NS:
    return NS(w1=_f32(eng, m.linear_1.weight), b1=_f32(eng, m.linear_1.bias),
              w2=_f32(eng, m.linear_2.weight), b2=_f32(eng, m.linear_2.bias))


def _pack_stage(stage, eng: Engine) -> List[NS]:
    from .models.blocks import AttentionBlock, ResidualBlock
    from .models.unet import Upsample
    out = []
    for layer in stage:
        if isinstance(layer, AttentionBlock):
            out.append(pack_attention_block(layer, eng))
        elif isinstance(layer, ResidualBlock):
            out.append(pack_residual_block(layer, eng))
        elif isinstance(layer, Upsample):
            out.append(pack_conv_layer(layer.conv, eng, up=True))
        elif isinstance(layer, torch.nn.Conv2d):
            if layer.kernel_size != (3, 3) or layer.padding != (1, 1):
                raise RuntimeError("only 3x3 / padding-1 convolutions are supported inside the U-Net")
            out.append(pack_conv_layer(layer, eng))
        else:
            raise RuntimeError(f"unsupported layer in SwitchSequential: {type(layer).__name__}")
    return out


def _stage_cout(layers: List[NS], cin: int) -> int:
    c = cin
    for lp in layers:
        c = lp.cout if lp.kind in ("res", "conv", "up") else c
    return c


def pack_unet(m, eng: Engine) -> NS:
    pk = NS(encoders=[_pack_stage(s, eng) for s in m.encoders],
            bottleneck=_pack_stage(m.bottleneck, eng),
            decoders=[_pack_stage(s, eng) for s in m.decoders])
    # all ResidualBlock time projections as one matrix: one launch per forward instead of 22
    res = [lp for st in pk.encoders + [pk.bottleneck] + pk.decoders for lp in st if lp.kind == "res"]
    off = 0
    for lp in res:
        lp.t_off = off
        off += lp.cout
    pk.lt_w = torch.cat([lp.lt_w for lp in res], 0).contiguous()
    pk.lt_b = torch.cat([lp.lt_b for lp in res], 0).contiguous()
    pk.attn = [lp for st in pk.encoders + [pk.bottleneck] + pk.decoders for lp in st if lp.kind == "attn"]
    pk.cin = pk.encoders[0][0].cin
    return pk


def pack_diffusion(m, eng: Engine) -> NS:
    return NS(te=pack_time_embedding(m.time_embedding, eng), unet=pack_unet(m.unet, eng),
              final=pack_final_layer(m.final, eng))


# ----------------------------------------------------------------------------------------------
# block runners (all tensors are engine-dtype [rows, C] views unless noted)
# ----------------------------------------------------------------------------------------------
def run_residual_block(eng: Engine, pk: NS, x, batch: int, h: int, w: int, tproj, out=None):
    """tproj: fp32 (batch, cout) view holding linear_time(silu(time))."""
    hw = h * w
    if out is None:
        out = eng.new(batch * hw, pk.cout)
    mark = eng.arena.mark()
    geom = dict(batch=batch, hin=h, win=w, hout=h, wout=w)
    # Both GroupNorm + SiLU pairs go to the convolution behind them as `gn_in`: in fp32 mode the convolution
    # normalises its A fragments itself (no launch, no normalised tensor), otherwise the GroupNorm kernel runs first.
    gn1 = dict(batch=batch, hw=hw, gamma=pk.gn1_w, beta=pk.gn1_b, eps=pk.gn1_eps, silu=True)
    gn2 = dict(batch=batch, hw=hw, gamma=pk.gn2_w, beta=pk.gn2_b, eps=pk.gn2_eps, silu=True)
    a = eng.gemm(x, pk.conv1, rowbias=tproj, rows_per_batch=hw, conv=dict(geom, cin_real=pk.cin), gn_batch=batch, gn_in=gn1)
    # the block output is group-normalised by whatever follows (ResidualBlock, AttentionBlock or FinalLayer)
    if pk.fold_skip:
        eng.gemm(a, pk.conv2, out=out, conv=dict(geom, cin_real=pk.cout, skip=x), gn_batch=batch, gn_in=gn2)
    else:
        skip = x if pk.res is None else eng.gemm(x, pk.res)
        eng.gemm(a, pk.conv2, out=out, residual=skip, conv=dict(geom, cin_real=pk.cout), gn_batch=batch, gn_in=gn2)
    eng.arena.release(mark)
    return out


def context_kv(eng: Engine, ca: NS, ctx_rows, batch: int, sk: int, persistent=False, out=None):
    """K and V rows of the conditioning sequence for one cross-attention.

    ``out`` = (k, v) buffers from an earlier call to refill in place (addresses baked into a captured graph).
    """
    k = eng.gemm(ctx_rows, ca.k_proj, out=None if out is None else out[0], persistent=persistent)
    v = eng.gemm(ctx_rows, ca.v_proj, out=None if out is None else out[1], persistent=persistent)
    return k, v


def run_self_attention(eng: Engine, sa: NS, x, batch: int, s: int, out=None, residual=None, causal=False,
                       ln_in=None, ln_out=False):
    """``ln_in``: row sums of ``x`` when the LayerNorm in front is folded into ``in_proj``; ``ln_out``: accumulate
    the row sums of the result for the LayerNorm that follows."""
    c = sa.heads * sa.d
    qkv = eng.gemm(x, sa.in_proj, ln_in=ln_in)
    o = eng.attention(qkv[:, 0:c], qkv[:, c:2 * c], qkv[:, 2 * c:3 * c], batch, sa.heads, s, s, sa.d, causal=causal)
    return eng.gemm(o, sa.out_proj, out=out, residual=residual, ln_out=ln_out)


def run_cross_attention(eng: Engine, ca: NS, x, kv, batch: int, s: int, sk: int, out=None, residual=None,
                        ln_in=None, ln_out=False):
    q = eng.gemm(x, ca.q_proj, ln_in=ln_in)
    o = eng.attention(q, kv[0], kv[1], batch, ca.heads, s, sk, ca.d)
    return eng.gemm(o, ca.out_proj, out=out, residual=residual, ln_out=ln_out)


def run_attention_block(eng: Engine, pk: NS, x, batch: int, h: int, w: int, kv, sk: int, out=None):
    s = h * w
    if out is None:
        out = eng.new(batch * s, pk.c)
    mark = eng.arena.mark()
    gn = dict(batch=batch, hw=s, gamma=pk.gn_w, beta=pk.gn_b, eps=pk.gn_eps, silu=False)
    if pk.fold:
        # The three LayerNorms never run as kernels: each GEMM that writes the residual stream also accumulates
        # the row sums, and the GEMM behind the LayerNorm (weights pre-scaled by gamma) applies mean / rstd.
        t = eng.gemm(x, pk.conv_in, ln_out=True, gn_in=gn)
        t = run_self_attention(eng, pk.sa, t, batch, s, residual=t, ln_in=eng.ln_stats(t), ln_out=True)
        t = run_cross_attention(eng, pk.ca, t, kv, batch, s, sk, residual=t, ln_in=eng.ln_stats(t), ln_out=True)
        if pk.tail is not None and eng.ffn_ok(batch * s, pk.c, batch):
            # the whole feed-forward half in one launch: the 4C-wide gated product stays on the CU
            eng.ffn_geglu(t, pk.geglu1, pk.tail, out, residual=x, ln_in=eng.ln_stats(t), gn_batch=batch)
            eng.arena.release(mark)
            return out
        a = eng.gemm(t, pk.geglu1, ln_in=eng.ln_stats(t))
    else:
        t = eng.gemm(x, pk.conv_in, gn_in=gn)
        a = eng.layernorm(t, pk.ln1_w, pk.ln1_b, pk.ln1_eps)
        t = run_self_attention(eng, pk.sa, a, batch, s, residual=t)
        a = eng.layernorm(t, pk.ln2_w, pk.ln2_b, pk.ln2_eps)
        t = run_cross_attention(eng, pk.ca, a, kv, batch, s, sk, residual=t)
        a = eng.layernorm(t, pk.ln3_w, pk.ln3_b, pk.ln3_eps)
        a = eng.gemm(a, pk.geglu1)
    if pk.tail is not None:
        eng.gemm(a, pk.tail, a2=t, out=out, residual=x, gn_batch=batch)
    else:
        t = eng.gemm(a, pk.geglu2, residual=t)
        eng.gemm(t, pk.conv_out, out=out, residual=x, gn_batch=batch)
    eng.arena.release(mark)
    return out


def _conv_out_hw(lp: NS, h: int, w: int):
    if lp.kind == "up":
        return 2 * h, 2 * w
    return (h + 2 - 3) // lp.stride + 1, (w + 2 - 3) // lp.stride + 1


def run_conv_layer(eng: Engine, lp: NS, x, batch: int, h: int, w: int, out=None):
    ho, wo = _conv_out_hw(lp, h, w)
    geom = dict(batch=batch, hin=h, win=w, hout=ho, wout=wo, stride=lp.stride, up=int(lp.kind == "up"),
                cin_real=lp.cin)
    return eng.gemm(x, lp.w, out=out, conv=geom, gn_batch=batch), ho, wo


def _run_stage(eng: Engine, layers: List[NS], x, batch, h, w, tproj, kvs, sk, out):
    """Run one SwitchSequential; the last layer writes into ``out`` (e.g. a concat-buffer slice)."""
    if out is None:  # allocate the result before the scope mark so it survives the release below
        c2, h2, w2 = _stage_shape(layers, x.shape[1], h, w)
        out = eng.new(batch * h2 * w2, c2)
    mark = eng.arena.mark()
    for i, lp in enumerate(layers):
        dst = out if i == len(layers) - 1 else None
        if lp.kind == "res":
            x = run_residual_block(eng, lp, x, batch, h, w, tproj[:, lp.t_off:lp.t_off + lp.cout], out=dst)
        elif lp.kind == "attn":
            x = run_attention_block(eng, lp, x, batch, h, w, kvs[id(lp)], sk, out=dst)
        else:
            x, h, w = run_conv_layer(eng, lp, x, batch, h, w, out=dst)
    eng.arena.release(mark)
    return x, h, w


def _stage_shape(layers: List[NS], c: int, h: int, w: int):
    for lp in layers:
        if lp.kind in ("conv", "up"):
            h, w = _conv_out_hw(lp, h, w)
        c = lp.cout if lp.kind != "attn" else c
    return c, h, w


def run_unet(eng: Engine, pk: NS, x, batch: int, h: int, w: int, tproj, kvs, sk: int):
    """x: [batch*h*w, cin_padded] rows.  Returns ([batch*h*w, 320] rows, h, w).

    The reference's ``torch.cat((x, skip))`` never happens: for every decoder stage one buffer
    [rows, C_x + C_skip] is laid out up front; the encoder stage that produces the skip writes it
    into the right-hand channel slice and the preceding decoder stage writes x into the left one.
    """
    n = len(pk.encoders)
    # shapes of the encoder outputs
    shapes = []
    c, hh, ww = pk.cin, h, w
    for st in pk.encoders:
        c, hh, ww = _stage_shape(st, c, hh, ww)
        shapes.append((c, hh, ww))
    # decoder inputs: x channels come from the bottleneck / previous decoder stage
    cx, hx, wx = _stage_shape(pk.bottleneck, *shapes[-1])
    cat = []
    for i, st in enumerate(pk.decoders):
        cs, hs, ws = shapes[n - 1 - i]
        if (hs, ws) != (hx, wx):
            raise RuntimeError(f"skip connection size mismatch at decoder {i}: {(hx, wx)} vs {(hs, ws)}; "
                               "latent height/width must be divisible by 8")
        cat.append((eng.new(batch * hs * ws, cx + cs), cx, cs))
        eng.gn_register(cat[-1][0], batch)  # both halves are written by GEMMs; the stage's first GroupNorm reads it whole
        cx, hx, wx = _stage_shape(st, cx + cs, hs, ws)

    hh, ww = h, w
    for j, st in enumerate(pk.encoders):
        buf, cxj, csj = cat[n - 1 - j]
        x, hh, ww = _run_stage(eng, st, x, batch, hh, ww, tproj, kvs, sk, out=buf[:, cxj:cxj + csj])
    buf0, cx0, _ = cat[0]
    x, hh, ww = _run_stage(eng, pk.bottleneck, x, batch, hh, ww, tproj, kvs, sk, out=buf0[:, 0:cx0])
    for i, st in enumerate(pk.decoders):
        buf = cat[i][0]
        if i + 1 < len(pk.decoders):
            nbuf, ncx, _ = cat[i + 1]
            dst = nbuf[:, 0:ncx]
        else:
            dst = None
        x, hh, ww = _run_stage(eng, st, buf, batch, hh, ww, tproj, kvs, sk, out=dst)
    return x, hh, ww


def precompute_context(eng: Engine, pk_unet: NS, context: torch.Tensor, persistent=False, out=None):
    """Cross-attention K / V of every AttentionBlock for a fixed conditioning tensor (B, S, 768).

    These 32 projections do not depend on the latent or the timestep, so a sampling loop computes
    them once (``persistent=True`` keeps them outside the per-forward arena; ``out`` refills the
    buffers of an earlier call).
    """
    batch, sk, _ = context.shape
    mark = eng.arena.mark()
    rows = eng.to_rows(context)
    kvs = {}
    for lp in pk_unet.attn:
        m2 = eng.arena.mark()
        kvs[id(lp)] = context_kv(eng, lp.ca, rows, batch, sk, persistent, None if out is None else out[id(lp)])
        if persistent or out is not None:
            eng.arena.release(m2)
    if persistent or out is not None:
        eng.arena.release(mark)
    return kvs, sk


def time_projections(eng: Engine, pk: NS, time_f32: torch.Tensor, out: Optional[torch.Tensor] = None):
    """(rows, 320) timestep features -> (rows, sum of C_out) fp32: TimeEmbedding followed by every ResidualBlock's
    ``linear_time(silu(.))``, all blocks side by side.  Rows are independent, so a sampling loop whose batch shares
    one timestep per step computes the table for all its steps once instead of three launches per step."""
    t = eng.linear_rows(time_f32, pk.te.w1, pk.te.b1, silu_in=False)
    t = eng.linear_rows(t, pk.te.w2, pk.te.b2, silu_in=True)
    return eng.linear_rows(t, pk.unet.lt_w, pk.unet.lt_b, silu_in=True, out=out)


def diffusion_forward_into(eng: Engine, pk: NS, latent_f32: torch.Tensor, time_f32: Optional[torch.Tensor], kvs, sk: int,
                           eps_out: torch.Tensor, tproj: Optional[torch.Tensor] = None):
    """Launch-only body (no torch allocation): latent (B,4,H,W) fp32, time (B,320) fp32 -> eps_out (B,4,H,W) fp32.
    ``tproj`` (B, sum of C_out; row stride 0 allowed) replaces the time path when the caller already has it."""
    batch, cin, h, w = latent_f32.shape
    hw = h * w
    if tproj is None:
        tproj = time_projections(eng, pk, time_f32)
    x = eng.nchw_to_rows(latent_f32, cpad=_rup(cin, 8))
    x, ho, wo = run_unet(eng, pk.unet, x, batch, h, w, tproj, kvs, sk)
    eng.gemm(x, pk.final.conv, nchw_out=eps_out, rows_per_batch=hw,
             conv=dict(batch=batch, hin=h, win=w, hout=h, wout=w, cin_real=pk.final.cin),
             gn_in=dict(batch=batch, hw=hw, gamma=pk.final.gn_w, beta=pk.final.gn_b, eps=pk.final.gn_eps, silu=True))
    return eps_out


# ----------------------------------------------------------------------------------------------
# CLIP text encoder (producer of the cross-attention context)
# ----------------------------------------------------------------------------------------------
def pack_clip_layer(m, eng: Engine) -> NS:
    return NS(ln1_w=_f32(eng, m.layernorm_1.weight), ln1_b=_f32(eng, m.layernorm_1.bias), ln1_eps=m.layernorm_1.eps,
              sa=pack_self_attention(m.attention, eng),
              ln2_w=_f32(eng, m.layernorm_2.weight), ln2_b=_f32(eng, m.layernorm_2.bias), ln2_eps=m.layernorm_2.eps,
              fc1=eng.pack_linear(m.linear_1), fc2=eng.pack_linear(m.linear_2))


def pack_clip(m, eng: Engine) -> NS:
    return NS(table=_f32(eng, m.embedding.token_embedding.weight), pos=_f32(eng, m.embedding.position_embedding),
              layers=[pack_clip_layer(l, eng) for l in m.layers],
              ln_w=_f32(eng, m.layernorm.weight), ln_b=_f32(eng, m.layernorm.bias), ln_eps=m.layernorm.eps)


def run_clip_layer(eng: Engine, pk: NS, x, batch: int, s: int):
    """Pre-LN transformer layer: x + attn(LN(x)) with the causal mask, then + fc2(quickgelu(fc1(LN(.))))."""
    mark = eng.arena.mark()
    a = eng.layernorm(x, pk.ln1_w, pk.ln1_b, pk.ln1_eps)
    t = run_self_attention(eng, pk.sa, a, batch, s, residual=x, causal=True)
    a = eng.layernorm(t, pk.ln2_w, pk.ln2_b, pk.ln2_eps)
    a = eng.gemm(a, pk.fc1, act="quickgelu")
    eng.gemm(a, pk.fc2, out=x, residual=t)  # x is dead by now: reuse it for the layer output
    eng.arena.release(mark)
    return x


def run_clip(eng: Engine, pk: NS, tokens: torch.Tensor):
    """tokens (B, S) int64 -> engine-dtype rows [B*S, E] (embedding, layers, final LayerNorm)."""
    batch, s = tokens.shape
    x = eng.embed_tokens(tokens, pk.table, pk.pos)
    for lp in pk.layers:
        x = run_clip_layer(eng, lp, x, batch, s)
    return eng.layernorm(x, pk.ln_w, pk.ln_b, pk.ln_eps)


# ----------------------------------------------------------------------------------------------
# VAE decoder / encoder (this build's own design - the reference has no runnable autoencoder; see models/vae.py)
# ----------------------------------------------------------------------------------------------
def pack_vae_residual(m, eng: Engine) -> NS:
    cin, cout = m.conv_feature.in_channels, m.conv_feature.out_channels
    fold = (eng.fuse_skip and isinstance(m.residual_layer, torch.nn.Conv2d) and cin % eng.bk == 0 and cout % eng.bk == 0)
    return NS(kind="vres", cin=cin, cout=cout,
              gn1_w=_f32(eng, m.groupnorm_feature.weight), gn1_b=_f32(eng, m.groupnorm_feature.bias), gn1_eps=m.groupnorm_feature.eps,
              conv1=eng.pack_conv3x3(m.conv_feature, same_size=True),
              gn2_w=_f32(eng, m.groupnorm_merged.weight), gn2_b=_f32(eng, m.groupnorm_merged.bias), gn2_eps=m.groupnorm_merged.eps,
              conv2=eng.pack_conv3x3(m.conv_merged, skip=m.residual_layer if fold else None, same_size=True),
              res=None if (fold or isinstance(m.residual_layer, torch.nn.Identity)) else eng.pack_conv1x1(m.residual_layer),
              fold_skip=fold)


def _pad_rows(t: torch.Tensor, n: int) -> torch.Tensor:
    out = torch.zeros((n,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    out[:t.shape[0]] = t
    return out


def pack_vae_layers(layers, eng: Engine) -> List[NS]:
    """Sequential layers of the autoencoder -> launch descriptions.  A GroupNorm directly followed by SiLU becomes one
    fused launch; channel counts that are not multiples of 8 (3, 4) are zero-padded to 8 on the k side, output widths
    that are not multiples of 4 (the 3 image planes) to 4 on the n side."""
    from .models.unet import Upsample
    from .models.vae import VAEAttentionBlock, VAEResidualBlock
    ops: List[NS] = []
    layers = list(layers)
    i = 0
    while i < len(layers):
        m = layers[i]
        if isinstance(m, VAEResidualBlock):
            ops.append(pack_vae_residual(m, eng))
        elif isinstance(m, VAEAttentionBlock):
            c = m.groupnorm.num_channels
            ops.append(NS(kind="vattn", c=c, gn_w=_f32(eng, m.groupnorm.weight), gn_b=_f32(eng, m.groupnorm.bias),
                          gn_eps=m.groupnorm.eps, in_proj=eng.pack_linear(m.attention.in_proj),
                          out_proj=eng.pack_linear(m.attention.out_proj)))
        elif isinstance(m, Upsample):
            ops.append(pack_conv_layer(m.conv, eng, up=True))
        elif isinstance(m, torch.nn.Conv2d):
            cin, cout = m.in_channels, m.out_channels
            if m.kernel_size == (1, 1):
                # 1x1 convolution on rows of width rup(cin, 8); its output feeds a layer that expects rup(cout, 8)
                kp, npd = _rup(cin, 8), _rup(cout, 8)
                w = torch.zeros(npd, kp, dtype=torch.float32, device=eng.device)
                w[:cout, :cin] = m.weight.detach().reshape(cout, cin).to(eng.device, torch.float32)
                b = torch.zeros(npd, dtype=torch.float32, device=eng.device)
                if m.bias is not None:
                    b[:cout] = m.bias.detach().to(eng.device, torch.float32)
                ops.append(NS(kind="conv1", cin=cin, cout=cout, cols=npd, w=eng.pack_matrix(w, b)))
            elif m.kernel_size == (3, 3):
                conv = m
                npd = _rup(cout, 8) if cout > 4 else 4
                if npd != cout:  # zero output rows keep the padded columns exactly zero (bias included)
                    # a parameter holder built without initialisation: the pack path must not draw from the global RNG
                    conv = torch.nn.utils.skip_init(torch.nn.Conv2d, cin, npd, 3, stride=m.stride, padding=1,
                                                    device=m.weight.device, dtype=m.weight.dtype)
                    with torch.no_grad():
                        conv.weight.copy_(_pad_rows(m.weight.detach(), npd))
                        if m.bias is not None:
                            conv.bias.copy_(_pad_rows(m.bias.detach(), npd))
                        else:
                            conv.bias.zero_()
                asym = m.stride == (2, 2) and m.padding == (0, 0)
                if not asym and m.padding != (1, 1):
                    raise RuntimeError("VAE convolutions are 3x3 with padding 1, or stride 2 with the (0,1,0,1) padding")
                ops.append(NS(kind="conv", cin=cin, cout=cout, cols=npd, stride=m.stride[0], pad_hi_only=int(asym),
                              w=eng.pack_conv3x3(conv, same_size=m.stride == (1, 1))))
            else:
                raise RuntimeError("unsupported convolution in the autoencoder")
        elif isinstance(m, torch.nn.GroupNorm):
            silu = i + 1 < len(layers) and isinstance(layers[i + 1], torch.nn.SiLU)
            ops.append(NS(kind="gn", c=m.num_channels, w=_f32(eng, m.weight), b=_f32(eng, m.bias), eps=m.eps, silu=silu,
                          groups=m.num_groups))
            i += int(silu)
        else:
            raise RuntimeError(f"unsupported layer in the autoencoder: {type(m).__name__}")
        i += 1
    return ops


def run_vae_residual(eng: Engine, pk: NS, x, batch: int, h: int, w: int):
    hw = h * w
    out = eng.new(batch * hw, pk.cout)
    mark = eng.arena.mark()
    geom = dict(batch=batch, hin=h, win=w, hout=h, wout=w)
    a = eng.groupnorm(x, batch, hw, pk.gn1_w, pk.gn1_b, pk.gn1_eps, silu=True)
    a = eng.gemm(a, pk.conv1, conv=dict(geom, cin_real=pk.cin), gn_batch=batch)
    a = eng.groupnorm(a, batch, hw, pk.gn2_w, pk.gn2_b, pk.gn2_eps, silu=True)
    if pk.fold_skip:
        eng.gemm(a, pk.conv2, out=out, conv=dict(geom, cin_real=pk.cout, skip=x), gn_batch=batch)
    else:
        skip = x if pk.res is None else eng.gemm(x, pk.res)
        eng.gemm(a, pk.conv2, out=out, residual=skip, conv=dict(geom, cin_real=pk.cout), gn_batch=batch)
    eng.arena.release(mark)
    return out


def run_vae_attention(eng: Engine, pk: NS, x, batch: int, s: int):
    """Single head of width C (512): wider than the flash kernel's LDS tiles allow, so the scores are materialised
    per sample - Q K^T and P V run on the GEMM kernel with K / V^T packed on the device as its weight operand."""
    c = pk.c
    if s % eng.epc:
        # the score matrix [s, s] is a GEMM output and a softmax operand: its rows are accessed as 16-byte pieces
        raise RuntimeError(f"autoencoder attention over {s} positions: the latent height x width must be a multiple of "
                           f"{eng.epc} for {'bf16' if eng.esz == 2 else 'fp32'} (image sides that are multiples of "
                           f"{32 if eng.esz == 2 else 16} always are)")
    out = eng.new(batch * s, c)
    mark = eng.arena.mark()
    qkv = eng.gemm(x, pk.in_proj, gn_in=dict(batch=batch, hw=s, gamma=pk.gn_w, beta=pk.gn_b, eps=pk.gn_eps, silu=False))
    o = eng.new(batch * s, c)
    for i in range(batch):
        m2 = eng.arena.mark()
        rows = slice(i * s, (i + 1) * s)
        q, k, v = qkv[rows, 0:c], qkv[rows, c:2 * c], qkv[rows, 2 * c:3 * c]
        scores = eng.gemm(q, eng.pack_rows(k), bias=False)
        eng.softmax_rows(scores, 1.0 / (c ** 0.5))
        eng.gemm(scores, eng.pack_rows(v, transposed=True), out=o[rows], bias=False)
        eng.arena.release(m2)
    eng.gemm(o, pk.out_proj, out=out, residual=x, gn_batch=batch)
    eng.arena.release(mark)
    return out


def run_vae_layers(eng: Engine, ops: List[NS], x, batch: int, h: int, w: int, nchw_out: Optional[torch.Tensor] = None):
    """x: engine-dtype rows [batch*h*w, C (padded to 8)].  Returns (rows, channels, h, w); with ``nchw_out`` the last
    convolution writes fp32 (B, C, H, W) planes there instead."""
    c = x.shape[1]
    for idx, op in enumerate(ops):
        last = idx == len(ops) - 1
        if op.kind == "vres":
            x, c = run_vae_residual(eng, op, x, batch, h, w), op.cout
        elif op.kind == "vattn":
            x = run_vae_attention(eng, op, x, batch, h * w)
        elif op.kind == "gn":
            x = eng.groupnorm(x, batch, h * w, op.w, op.b, op.eps, silu=op.silu, groups=op.groups)
        elif op.kind == "conv1":
            x, c = eng.gemm(x, op.w, gn_batch=0), op.cout
        elif op.kind in ("conv", "up"):
            if op.kind == "up":
                ho, wo = 2 * h, 2 * w
                geom = dict(batch=batch, hin=h, win=w, hout=ho, wout=wo, up=1, cin_real=op.cin)
            else:
                ho, wo = ((h - 2) // 2 + 1, (w - 2) // 2 + 1) if op.pad_hi_only else ((h - 1) // op.stride + 1, (w - 1) // op.stride + 1)
                geom = dict(batch=batch, hin=h, win=w, hout=ho, wout=wo, stride=op.stride, cin_real=op.cin,
                            pad_hi_only=op.pad_hi_only)
            if last and nchw_out is not None:
                eng.gemm(x, op.w, nchw_out=nchw_out, rows_per_batch=ho * wo, conv=geom)
                x = None
            else:
                x = eng.gemm(x, op.w, conv=geom, gn_batch=batch)
            h, w, c = ho, wo, op.cout
    return x, c, h, w
