"""CLIP text encoder (producer of the cross-attention context): oracle against the vectors recorded from the
reference (CPU), and the HIP path against both (GPU).  fp32 mode is asserted at max-rel <= 1e-4; bf16 mode
at <= 3e-2 (12 layers of bf16 storage on a LayerNorm-ed output of unit scale).
"""
import numpy as np
import pytest
import torch

from conftest import load_golden, max_rel
from diffusion_model_amd.weights import deterministic_fill_, fill_state, test_input as tin
from oracle import clip_ref


def test_oracle_reproduces_reference_vectors():
    g = load_golden("clip")
    state = fill_state(clip_ref.clip_layer_shapes(768), seed=0)
    out = clip_ref.clip_layer(state, "", tin("clip.layer.x", (2, 50, 768)), 12)
    assert max_rel(out, g["layer_2x50"]) < 1e-5
    state = fill_state(clip_ref.clip_shapes(1000, 256, 33, 3), seed=0)
    out = clip_ref.clip(state, torch.as_tensor(g["small_tokens"]), 4)
    assert max_rel(out, g["small_out"]) < 1e-5


def test_oracle_full_size():
    g = load_golden("clip")
    state = fill_state(clip_ref.clip_shapes(), seed=0)
    out = clip_ref.clip(state, torch.as_tensor(g["full_tokens"]), 12)
    assert max_rel(out, g["full_out"]) < 1e-5


def test_causal_mask_really_masks():
    """Changing a later token must not change earlier positions (and must change its own)."""
    state = fill_state(clip_ref.clip_shapes(1000, 256, 33, 3), seed=0)
    tok = torch.arange(33).remainder(1000)[None]
    a = clip_ref.clip(state, tok, 4)
    tok2 = tok.clone(); tok2[0, 20] = 7
    b = clip_ref.clip(state, tok2, 4)
    assert torch.equal(a[:, :20], b[:, :20]) and not torch.equal(a[:, 20], b[:, 20])


def test_front_end_needs_the_device():
    from diffusion_model_amd.models import CLIP
    m = CLIP(n_vocab=100, n_embed=64, n_tokens=8, n_layers=1, n_heads=2)
    assert [k for k, _ in m.state_dict().items()] == [k for k, _ in clip_ref.clip_shapes(100, 64, 8, 1)]
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 8, dtype=torch.long))  # CPU tensors: there is no CPU path


# ------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-4), (torch.bfloat16, 3e-2)], ids=["f32", "bf16"])
def test_hip_clip_layer_and_small_stack(dev, dtype, tol):
    from diffusion_model_amd.models import CLIP, CLIPLayer
    g = load_golden("clip")
    layer = deterministic_fill_(CLIPLayer(12, 768)).to(dev).to(dtype)
    x = tin("clip.layer.x", (2, 50, 768))
    out = layer(x.to(dev).to(dtype))
    assert out.shape == (2, 50, 768) and out.dtype == dtype
    assert max_rel(out.float(), g["layer_2x50"]) < tol
    small = deterministic_fill_(CLIP(n_vocab=1000, n_embed=256, n_tokens=33, n_layers=3, n_heads=4)).to(dev).to(dtype)
    out = small(torch.as_tensor(g["small_tokens"]).to(dev))
    assert max_rel(out.float(), g["small_out"]) < tol
    with pytest.raises(IndexError):
        small(torch.full((1, 33), 1000, dtype=torch.long, device=dev))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-4), (torch.bfloat16, 3e-2)], ids=["f32", "bf16"])
def test_hip_clip_full_size(dev, dtype, tol):
    from diffusion_model_amd.models import CLIP
    g = load_golden("clip")
    with torch.device("meta"):
        m = CLIP()
    m = deterministic_fill_(m.to_empty(device=dev)).to(dtype).eval()
    tok = torch.as_tensor(g["full_tokens"]).to(dev)
    out = m.encode_text(tok)
    assert out.shape == (1, 77, 768)
    assert max_rel(out.float(), g["full_out"]) < tol
    # batch of prompts against the oracle on this host; causal masking: a changed late token leaves
    # earlier positions bitwise unchanged
    tok3 = torch.randint(0, 49408, (3, 77), generator=torch.Generator().manual_seed(11))
    out3 = m(tok3.to(dev))
    if dtype == torch.float32:
        state = fill_state(clip_ref.clip_shapes(), seed=0)
        assert max_rel(out3, clip_ref.clip(state, tok3, 12)) < tol
    tok4 = tok3.clone(); tok4[:, 60] = 5
    out4 = m(tok4.to(dev))
    assert torch.equal(out3[:, :60], out4[:, :60]) and not torch.equal(out3[:, 60], out4[:, 60])
