"""Image range / layout helpers (rescale, move_channel, the uint8 post-processing chain): oracle against the
vectors recorded from the reference's scripts/utils.py (CPU), and the HIP kernels against both (GPU).
Integer output and separately rounded fp32 steps: the bar is bit-exact.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden
from diffusion_model_amd.weights import test_input as tin
from oracle import image_ref

U8_CASES = (("2x3x16x20", (2, 3, 16, 20)), ("1x3x5x7", (1, 3, 5, 7)), ("1x1x6x6", (1, 1, 6, 6)))


def _x255():
    return (tin("post.x255", (1, 3, 9, 11)).abs() * 97.0).clamp(0, 255).floor()


def test_oracle_reproduces_reference_vectors():
    g = load_golden("postprocess")
    for tag, shape in U8_CASES:
        out = image_ref.images_to_uint8(tin("post." + tag, shape) * 1.5)
        assert out.dtype == torch.uint8 and np.array_equal(out.numpy(), g["u8_" + tag]), tag
        assert out.min() == 0 and out.max() == 255  # both clamp edges are hit
    x = tin("post.noclamp", (2, 3, 8, 8)) * 1.5
    assert np.array_equal(image_ref.rescale(x, (-1, 1), (0, 255)).numpy(), g["rescale_noclamp"])
    assert np.array_equal(image_ref.rescale(_x255(), (0, 255), (-1, 1)).numpy(), g["rescale_255_to_pm1"])
    xb = (tin("post.bf16", (1, 3, 8, 12)) * 1.5).bfloat16()
    assert np.array_equal(image_ref.images_to_uint8(xb).numpy(), g["u8_bf16_1x3x8x12"])


def _picture():
    from PIL import Image
    return Image.fromarray((tin("post.pic", (29, 37, 3)).abs() * 120.0).clamp(0, 255).to(torch.uint8).numpy())


def test_oracle_prepare_image_tensor():
    g = load_golden("postprocess")
    for dt, tag in ((torch.float32, "f32"), (torch.bfloat16, "bf16")):
        out = image_ref.prepare_image_tensor(_picture(), 24, 16, dt)
        assert out.shape == (3, 16, 24) and out.dtype == dt
        assert np.array_equal(out.float().numpy(), g["prepare_" + tag]), tag


def test_move_channel_views_and_errors():
    from diffusion_model_amd.pipeline import move_channel
    x = torch.arange(2 * 3 * 4 * 5).reshape(2, 3, 4, 5)
    last = move_channel(x, to="last")
    assert last.shape == (2, 4, 5, 3) and torch.equal(move_channel(last, to="first"), x)
    assert torch.equal(last, image_ref.move_channel(x, "last"))
    assert move_channel(x[0], to="last").shape == (4, 5, 3) and move_channel(x[0, 0], to="last").shape == (4, 5)
    with pytest.raises(ValueError):
        move_channel(x, to="middle")


def test_cpu_tensors_are_rejected():
    from diffusion_model_amd.pipeline import images_to_uint8, rescale
    with pytest.raises(RuntimeError):
        rescale(torch.zeros(4), (-1, 1), (0, 255))
    with pytest.raises(RuntimeError):
        images_to_uint8(torch.zeros(1, 3, 4, 4))


@pytest.mark.gpu
def test_hip_uint8_chain_bit_exact(dev):
    from diffusion_model_amd.pipeline import images_to_uint8
    g = load_golden("postprocess")
    for tag, shape in U8_CASES:  # 4-pixel RGB path, generic path with a ragged pixel count, one channel
        out = images_to_uint8((tin("post." + tag, shape) * 1.5).to(dev))
        assert out.dtype == torch.uint8 and out.shape == (shape[0], shape[2], shape[3], shape[1])
        assert np.array_equal(out.cpu().numpy(), g["u8_" + tag]), tag
    xb = (tin("post.bf16", (1, 3, 8, 12)) * 1.5).bfloat16()
    assert np.array_equal(images_to_uint8(xb.to(dev)).cpu().numpy(), g["u8_bf16_1x3x8x12"])
    # decoder-sized batch (768x768 RGB) against the oracle, plus a non-contiguous input
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(2, 3, 768, 768, generator=gen) * 0.8
    assert torch.equal(images_to_uint8(x.to(dev)).cpu(), image_ref.images_to_uint8(x))
    xt = torch.randn(2, 40, 24, 3, generator=gen).permute(0, 3, 1, 2)
    assert torch.equal(images_to_uint8(xt.to(dev)).cpu(), image_ref.images_to_uint8(xt))


@pytest.mark.gpu
def test_hip_rescale_bit_exact(dev):
    from diffusion_model_amd.pipeline import rescale
    g = load_golden("postprocess")
    x = tin("post.noclamp", (2, 3, 8, 8)) * 1.5
    assert np.array_equal(rescale(x.to(dev), (-1, 1), (0, 255)).cpu().numpy(), g["rescale_noclamp"])
    assert np.array_equal(rescale(_x255().to(dev), (0, 255), (-1, 1)).cpu().numpy(), g["rescale_255_to_pm1"])
    gen = torch.Generator().manual_seed(4)
    x = torch.randn(3, 1001, generator=gen) * 300.0
    for old, new, clamp in (((0, 255), (-1, 1), False), ((-1, 1), (0, 255), True), ((-3.5, 9), (2, 2.25), True)):
        assert torch.equal(rescale(x.to(dev), old, new, clamp).cpu(), image_ref.rescale(x, old, new, clamp)), (old, new)
    xb = x.bfloat16()
    assert torch.equal(rescale(xb.to(dev), (0, 255), (-1, 1)).cpu(), image_ref.rescale(xb, (0, 255), (-1, 1)))
    with pytest.raises(ValueError):
        rescale(x.to(dev), (1, 1), (0, 255))


@pytest.mark.gpu
def test_hip_prepare_and_save_images(dev, tmp_path):
    from PIL import Image
    from diffusion_model_amd.pipeline import images_to_uint8, prepare_image_tensor, save_images
    g = load_golden("postprocess")
    for dt, tag in ((torch.float32, "f32"), (torch.bfloat16, "bf16")):
        out = prepare_image_tensor(_picture(), 24, 16, dtype=dt, device=dev)
        assert out.shape == (3, 16, 24) and out.dtype == dt and out.is_contiguous()
        assert np.array_equal(out.float().cpu().numpy(), g["prepare_" + tag]), tag
    path = str(tmp_path / "pic.png")
    _picture().save(path)
    from_path = prepare_image_tensor(path, 24, 16, device=dev)
    assert np.array_equal(from_path.cpu().numpy(), g["prepare_f32"])
    with pytest.raises(RuntimeError):
        prepare_image_tensor(_picture(), 24, 16, device="cpu")
    # round trip through the PNG writer: the files hold exactly the uint8 pixels of the post-processing chain
    batch = torch.stack([from_path, -from_path])
    save_images(batch, str(tmp_path / "out"), prefix="t")
    want = images_to_uint8(batch).cpu().numpy()
    for i in range(2):
        got = np.array(Image.open(str(tmp_path / "out" / f"t_{i:04d}.png")))
        assert np.array_equal(got, want[i]), i
