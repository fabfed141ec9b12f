"""Functional fp32 statement of the VAE decoder and encoder (test oracle) - PARITY UNPINNED.

The reference has NO decoder (scripts/model_loader.py:9 imports a module that does not exist,
scripts/inference.py:325-332 is commented out) and its Encoder cannot run (src/models/encoder.py:112-141 puts the
U-Net's ResidualBlock, whose forward needs a time argument, into an nn.Sequential; measured: TypeError).  There are
therefore no reference outputs to pin these functions against.  What they state is the design the build implements:
the Stable-Diffusion-1.5 autoencoder laid out as the mirror of src/models/encoder.py:107-152 -

  * residual blocks 512 / 256 / 128 WITHOUT a time input: GN(32) -> SiLU -> conv3x3 -> GN(32) -> SiLU -> conv3x3,
    plus the input (through a 1x1 convolution when the channel count changes)           [encoder.py:112-141]
  * one single-head self-attention block over the 64x64 (latent-resolution) map: GN(32) -> fused q/k/v projection
    -> softmax(q k^T / sqrt(C)) v -> output projection -> + input                       [encoder.py:20-69]
  * nearest x2 upsampling followed by conv3x3 where the encoder has its stride-2 convolutions
  * the latent scaling 0.18215: the encoder multiplies by it (encoder.py:153-155,209-210), the decoder divides first
  * encoder: asymmetric (0, 1, 0, 1) zero padding in front of the stride-2 convolutions (encoder.py:179-185),
    mean / clamped log-variance / reparameterisation as in encoder.py:191-212.

Test infrastructure only; the GPU tests compare the HIP path with these functions on identical weights.
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

State = Dict[str, torch.Tensor]
SCALING = 0.18215

# decoder layout: entries of Decoder.layers, in order
DECODER = ([("conv", 4, 4, 1), ("conv", 4, 512, 3), ("res", 512, 512), ("attn", 512)]
           + [("res", 512, 512)] * 4
           + [("up", 512)] + [("res", 512, 512)] * 3
           + [("up", 512), ("res", 512, 256), ("res", 256, 256), ("res", 256, 256)]
           + [("up", 256), ("res", 256, 128), ("res", 128, 128), ("res", 128, 128)]
           + [("gn", 128), ("silu",), ("conv", 128, 3, 3)])

# encoder layout (src/models/encoder.py:107-152, residual blocks without time)
ENCODER = ([("conv", 3, 128, 3), ("res", 128, 128), ("res", 128, 128), ("down", 128),
            ("res", 128, 256), ("res", 256, 256), ("down", 256),
            ("res", 256, 512), ("res", 512, 512), ("down", 512)]
           + [("res", 512, 512)] * 3 + [("attn", 512), ("res", 512, 512), ("gn", 512), ("silu",),
                                        ("conv", 512, 8, 3), ("conv", 8, 8, 1)])


def _entry_shapes(e) -> List[Tuple[str, Tuple[int, ...]]]:
    kind = e[0]
    if kind == "conv":
        _, cin, cout, k = e
        return [("weight", (cout, cin, k, k)), ("bias", (cout,))]
    if kind in ("up", "down"):
        c = e[1]
        return [("conv.weight", (c, c, 3, 3)), ("conv.bias", (c,))] if kind == "up" else \
               [("weight", (c, c, 3, 3)), ("bias", (c,))]
    if kind == "res":
        _, cin, cout = e
        s = [("groupnorm_feature.weight", (cin,)), ("groupnorm_feature.bias", (cin,)),
             ("conv_feature.weight", (cout, cin, 3, 3)), ("conv_feature.bias", (cout,)),
             ("groupnorm_merged.weight", (cout,)), ("groupnorm_merged.bias", (cout,)),
             ("conv_merged.weight", (cout, cout, 3, 3)), ("conv_merged.bias", (cout,))]
        if cin != cout:
            s += [("residual_layer.weight", (cout, cin, 1, 1)), ("residual_layer.bias", (cout,))]
        return s
    if kind == "attn":
        c = e[1]
        return [("groupnorm.weight", (c,)), ("groupnorm.bias", (c,)),
                ("attention.in_proj.weight", (3 * c, c)), ("attention.in_proj.bias", (3 * c,)),
                ("attention.out_proj.weight", (c, c)), ("attention.out_proj.bias", (c,))]
    if kind == "gn":
        return [("weight", (e[1],)), ("bias", (e[1],))]
    return []


def shapes(layout) -> List[Tuple[str, Tuple[int, ...]]]:
    out = []
    for i, e in enumerate(layout):
        out += [(f"layers.{i}.{n}", s) for n, s in _entry_shapes(e)]
    return out


def decoder_shapes():
    return shapes(DECODER)


def encoder_shapes():
    return shapes(ENCODER)


def _sub(state: State, prefix: str) -> State:
    n = len(prefix)
    return {k[n:]: v for k, v in state.items() if k.startswith(prefix)}


def residual_block(w: State, x: torch.Tensor) -> torch.Tensor:
    h = F.group_norm(x, 32, w["groupnorm_feature.weight"], w["groupnorm_feature.bias"], eps=1e-5)
    h = F.conv2d(F.silu(h), w["conv_feature.weight"], w["conv_feature.bias"], padding=1)
    h = F.group_norm(h, 32, w["groupnorm_merged.weight"], w["groupnorm_merged.bias"], eps=1e-5)
    h = F.conv2d(F.silu(h), w["conv_merged.weight"], w["conv_merged.bias"], padding=1)
    if "residual_layer.weight" in w:
        x = F.conv2d(x, w["residual_layer.weight"], w["residual_layer.bias"])
    return h + x


def attention_block(w: State, x: torch.Tensor) -> torch.Tensor:
    b, c, hh, ww = x.shape
    h = F.group_norm(x, 32, w["groupnorm.weight"], w["groupnorm.bias"], eps=1e-5)
    h = h.reshape(b, c, hh * ww).transpose(1, 2)
    q, k, v = F.linear(h, w["attention.in_proj.weight"], w["attention.in_proj.bias"]).chunk(3, dim=-1)
    wgt = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(c), dim=-1)
    h = F.linear(wgt @ v, w["attention.out_proj.weight"], w["attention.out_proj.bias"])
    return h.transpose(1, 2).reshape(b, c, hh, ww) + x


def _run(state: State, layout, x: torch.Tensor) -> torch.Tensor:
    for i, e in enumerate(layout):
        w = _sub(state, f"layers.{i}.")
        kind = e[0]
        if kind == "conv":
            x = F.conv2d(x, w["weight"], w["bias"], padding=e[3] // 2)
        elif kind == "down":
            x = F.conv2d(F.pad(x, (0, 1, 0, 1)), w["weight"], w["bias"], stride=2)
        elif kind == "up":
            x = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), w["conv.weight"], w["conv.bias"], padding=1)
        elif kind == "res":
            x = residual_block(w, x)
        elif kind == "attn":
            x = attention_block(w, x)
        elif kind == "gn":
            x = F.group_norm(x, 32, w["weight"], w["bias"], eps=1e-5)
        elif kind == "silu":
            x = F.silu(x)
    return x


def decode(state: State, latent: torch.Tensor) -> torch.Tensor:
    """latent (B, 4, h, w) -> image (B, 3, 8h, 8w)."""
    return _run(state, DECODER, latent / SCALING)


def encode(state: State, image: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
    """image (B, 3, H, W), noise (B, 4, H/8, W/8) -> latent (B, 4, H/8, W/8)."""
    x = _run(state, ENCODER, image)
    mean, log_variance = torch.chunk(x, 2, dim=1)
    log_variance = torch.clamp(log_variance, -30, 20)
    std = log_variance.exp().sqrt()
    return (mean + std * noise) * SCALING
