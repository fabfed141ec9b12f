"""dm_groupnorm_noalloc (the entry point that allocates nothing) against dm_groupnorm_stats + dm_groupnorm_apply.

Both run the same two kernels with the same slab count; only the home of the slab partials differs (the library's
fixed device array instead of a caller workspace), so outputs must be bitwise equal.  Two calls in a row, the second
on a smaller input: leftovers of the first call in the shared scratch must not leak into the second.
"""
import pytest
import torch

from diffusion_model_amd import _lib
from diffusion_model_amd._lib import check

pytestmark = pytest.mark.gpu


def _both(dev, batch, hw, c, dtype, silu, seed):
    lib = _lib.lib()
    dm = _lib.DM_F32 if dtype == torch.float32 else _lib.DM_BF16
    gen = torch.Generator().manual_seed(seed)
    x = (torch.randn(batch * hw, c, generator=gen) * 2 + 0.5).to(dev).to(dtype)
    gamma = (1 + 0.1 * torch.randn(c, generator=gen)).to(dev)
    beta = (0.1 * torch.randn(c, generator=gen)).to(dev)
    s = torch.cuda.current_stream(dev).cuda_stream
    nsplit = lib.dm_groupnorm_default_nsplit(batch, hw)
    part = torch.empty(batch * nsplit * 32 * 2, dtype=torch.float32, device=dev)
    want = torch.empty_like(x)
    check(lib.dm_groupnorm_stats(x.data_ptr(), c, batch, hw, c, 32, dm, part.data_ptr(), nsplit, s), "stats")
    check(lib.dm_groupnorm_apply(x.data_ptr(), c, want.data_ptr(), c, batch, hw, c, 32, dm, part.data_ptr(), nsplit,
                                 gamma.data_ptr(), beta.data_ptr(), 1e-5, int(silu), s), "apply")
    got = torch.full_like(x, float("nan"))
    # free device memory as the driver sees it (hipMemGetInfo): a hipMalloc inside the library would show here, which
    # torch's own allocator statistics cannot see
    torch.cuda.synchronize(dev)
    before, free_before = torch.cuda.memory_allocated(dev), torch.cuda.mem_get_info(dev)[0]
    check(lib.dm_groupnorm_noalloc(x.data_ptr(), c, got.data_ptr(), c, batch, hw, c, 32, dm, gamma.data_ptr(),
                                   beta.data_ptr(), 1e-5, int(silu), s), "noalloc")
    torch.cuda.synchronize(dev)
    assert torch.cuda.memory_allocated(dev) == before
    assert torch.cuda.mem_get_info(dev)[0] >= free_before
    # also against plain torch math (fp32 statistics), so that "equal" cannot mean "equally wrong"
    xf = x.float().view(batch, hw, 32, c // 32)
    mean = xf.mean(dim=(1, 3), keepdim=True)
    var = xf.var(dim=(1, 3), keepdim=True, unbiased=False)
    ref = ((xf - mean) / torch.sqrt(var + 1e-5)).view(batch * hw, c) * gamma + beta
    if silu:
        ref = ref * torch.sigmoid(ref)
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    assert float((got.float() - ref).abs().max() / ref.abs().max()) < tol
    return got, want


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_no_alloc_groupnorm_matches_workspace_form(dev, dtype):
    lib = _lib.lib()
    check(lib.dm_init(), "init")
    big, big_want = _both(dev, 4, 64 * 64, 320, dtype, True, 1)
    assert torch.equal(big, big_want)
    small, small_want = _both(dev, 1, 8 * 8, 1280, dtype, False, 2)  # second call, smaller input, same scratch
    assert torch.equal(small, small_want)
    even, even_want = _both(dev, 3, 24 * 16, 640, dtype, True, 3)
    assert torch.equal(even, even_want)
    # rows per sample that the slab count does not divide: the last slab of every sample is a partial one
    for hw in (100, 25 * 25):
        assert hw % lib.dm_groupnorm_default_nsplit(2, hw) != 0
        ragged, ragged_want = _both(dev, 2, hw, 320, dtype, True, 4)
        assert torch.equal(ragged, ragged_want)


def test_no_alloc_groupnorm_rejects_oversize(dev):
    lib = _lib.lib()
    x = torch.zeros(16, 64, device=dev)
    g = torch.ones(64, device=dev)
    # 4096 samples x 8 slabs x 32 groups x 2 floats = 2^21 floats > the 2^18-float scratch
    rc = lib.dm_groupnorm_noalloc(x.data_ptr(), 64, x.data_ptr(), 64, 4096, 4096, 64, 32, _lib.DM_F32, g.data_ptr(),
                                  g.data_ptr(), 1e-5, 0, 0)
    assert rc == 5
