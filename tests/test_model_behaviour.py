"""The behaviour the reference's own model tests ask for (tests/test_models.py there: TimeEmbedding :26-68, Upsample
:190-217, SwitchSequential :220-258, Diffusion :342-409), checked on the HIP front-ends.  Those tests assert shapes,
finiteness, routing and a parameter-count bound, no numbers; the numeric parity of the same modules is pinned by the
golden vectors in test_hip_model.py.  Gradient-flow tests have no counterpart: this package is the sampling path and
its forward passes run under no_grad on hand-written kernels."""
import json
import os

import pytest
import torch

from conftest import GOLDEN
from diffusion_model_amd.models import (AttentionBlock, Diffusion, FinalLayer, ResidualBlock, SwitchSequential,
                                        TimeEmbedding, UNet, Upsample)
from diffusion_model_amd.weights import deterministic_fill_


def _rand(shape, seed, dev, scale=1.0):
    return (torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale).to(dev)


# ---------------------------------------------------------------- structure (no device needed)
def test_time_embedding_layout():
    m = TimeEmbedding(n_embed=320)
    assert (m.linear_1.in_features, m.linear_1.out_features) == (320, 1280)
    assert (m.linear_2.in_features, m.linear_2.out_features) == (1280, 1280)


def test_upsample_layout():
    m = Upsample(channels=320)
    assert m.conv.in_channels == 320 and m.conv.out_channels == 320


def test_diffusion_layout_and_parameter_count():
    with torch.device("meta"):
        m = Diffusion()
    assert isinstance(m.time_embedding, TimeEmbedding) and isinstance(m.unet, UNet) and isinstance(m.final, FinalLayer)
    assert len(m.unet.encoders) == 12 and len(m.unet.decoders) == 12
    total = sum(p.numel() for p in m.parameters())
    assert 1_000_000 < total < 2_000_000_000
    assert total == 859_520_964  # what the reference's Diffusion() holds


def test_reference_encoder_state_dict_loads():
    """A state_dict with the reference Encoder's names and shapes (recorded from the reference by
    tools/gen_golden.py encoder_keys) loads into this Encoder with strict=True: its ResidualBlock parameter names are
    kept, nn.MultiheadAttention's fused in_proj_weight / in_proj_bias map onto SelfAttention.in_proj, and linear_time
    (which the reference's Encoder can never use) is dropped."""
    from diffusion_model_amd.models import Encoder
    with open(os.path.join(GOLDEN, "encoder_state_keys.json")) as f:
        ref = json.load(f)
    enc = Encoder()
    mine = {k: tuple(v.shape) for k, v in enc.state_dict().items()}
    renamed = {k.replace("attention.in_proj_weight", "attention.in_proj.weight")
                .replace("attention.in_proj_bias", "attention.in_proj.bias"): tuple(v)
               for k, v in ref.items() if ".linear_time." not in k}
    assert renamed == mine
    state = {k: torch.full(v, 0.25) for k, v in ref.items()}
    res = enc.load_state_dict(state, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    assert float(enc.layers[13].attention.in_proj.weight.detach()[5, 7]) == 0.25
    assert float(enc.layers[1].conv_feature.bias.detach()[3]) == 0.25


# ---------------------------------------------------------------- behaviour on the device
@pytest.mark.gpu
def test_time_embedding_shapes_and_distinct_timesteps(dev):
    m = deterministic_fill_(TimeEmbedding(320), prefix="time_embedding.").to(dev).eval()
    for batch in (1, 2, 8, 16):
        out = m(_rand((batch, 320), batch, dev))
        assert out.shape == (batch, 1280) and torch.isfinite(out).all()
    a, b = m(_rand((1, 320), 100, dev)), m(_rand((1, 320), 101, dev))
    assert not torch.allclose(a, b)
    # rows are independent: a batch equals its single-row runs
    x = _rand((4, 320), 7, dev)
    full = m(x)
    for i in range(4):
        assert torch.allclose(full[i:i + 1], m(x[i:i + 1].contiguous()), rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("channels", [64, 128, 320, 640, 1280])
def test_upsample_doubles_the_plane_and_keeps_channels(dev, channels):
    m = deterministic_fill_(Upsample(channels), prefix="up.").to(dev).eval()
    out = m(_rand((2, channels, 8, 8), channels, dev))
    assert out.shape == (2, channels, 16, 16) and torch.isfinite(out).all()


@pytest.mark.gpu
def test_switch_sequential_routing(dev):
    x, ctx, time = _rand((2, 320, 16, 16), 1, dev), _rand((2, 77, 768), 2, dev), _rand((2, 1280), 3, dev)
    res = deterministic_fill_(ResidualBlock(320, 320), prefix="res.").to(dev).eval()
    conv = deterministic_fill_(torch.nn.Conv2d(320, 320, kernel_size=3, padding=1), prefix="conv.").to(dev)
    attn = deterministic_fill_(AttentionBlock(8, 40), prefix="attn.").to(dev).eval()
    # (x, time) to the ResidualBlock: same as calling it directly, and the time input matters
    out = SwitchSequential(res)(x, ctx, time)
    assert out.shape == (2, 320, 16, 16) and torch.equal(out, res(x, time))
    assert not torch.allclose(out, SwitchSequential(res)(x, ctx, time + 1.0))
    # x alone to a plain convolution
    stem = deterministic_fill_(torch.nn.Conv2d(4, 320, kernel_size=3, padding=1), prefix="stem.").to(dev)
    assert SwitchSequential(stem)(_rand((2, 4, 64, 64), 4, dev), ctx, time).shape == (2, 320, 64, 64)
    # (x, context) to the AttentionBlock
    out = SwitchSequential(attn)(x, ctx, time)
    assert torch.equal(out, attn(x, ctx)) and not torch.allclose(out, SwitchSequential(attn)(x, ctx * 0.5, time))
    # mixed layers run in order
    mixed = SwitchSequential(conv, res, attn)(x, ctx, time)
    assert mixed.shape == (2, 320, 16, 16) and torch.isfinite(mixed).all()
    assert torch.allclose(mixed, attn(res(SwitchSequential(conv)(x, ctx, time), time), ctx), rtol=1e-5, atol=1e-6)


@pytest.fixture(scope="module")
def diffusion(dev):
    with torch.device("meta"):
        m = Diffusion()
    return deterministic_fill_(m.to_empty(device=dev), seed=0).eval()


@pytest.mark.gpu
@pytest.mark.parametrize("batch", [1, 2, 4])
def test_diffusion_forward_shapes(dev, diffusion, batch):
    lat, ctx, time = _rand((batch, 4, 32, 32), 10, dev), _rand((batch, 77, 768), 11, dev), _rand((batch, 320), 12, dev)
    out = diffusion(lat, ctx, time)
    assert out.shape == (batch, 4, 32, 32) and out.dtype == lat.dtype and torch.isfinite(out).all()


@pytest.mark.gpu
def test_diffusion_output_depends_on_the_timestep_features(dev, diffusion):
    lat, ctx = _rand((1, 4, 32, 32), 20, dev), _rand((1, 77, 768), 21, dev)
    a, b = diffusion(lat, ctx, _rand((1, 320), 22, dev)), diffusion(lat, ctx, _rand((1, 320), 23, dev))
    assert not torch.allclose(a, b)
    assert torch.equal(a, diffusion(lat, ctx, _rand((1, 320), 22, dev)))  # and on nothing hidden
