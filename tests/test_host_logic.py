"""CPU-only checks: the C-ABI library loads and exports what include/dm_hip.h declares, host-side
schedule tables match the reference's golden vectors, and the front-ends refuse to compute on the CPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden
from diffusion_model_amd import _lib
from diffusion_model_amd.engine import Arena
from diffusion_model_amd.models import Diffusion, ResidualBlock, TimeEmbedding
from diffusion_model_amd.sampling import (KEulerAncestralSampler, KEulerSampler, KLMSSampler, get_alphas_cumprod,
                                          get_sigmas_karras, sigma_to_timestep, timestep_to_sigma)
from diffusion_model_amd.schedulers import DDIMScheduler, DDPMScheduler
from diffusion_model_amd.weights import fill_value, unit_uniform
from oracle import sched_ref, unet_ref


# ------------------------------------------------------------------------------------------- C ABI
def declared_symbols():
    text = open(os.path.join(ROOT, "include", "dm_hip.h")).read()
    return sorted(set(re.findall(r"^\s*(?:int|long|void)\s+(dm_\w+)\s*\(", text, flags=re.M)))


def test_library_exports_every_declared_symbol():
    names = declared_symbols()
    assert len(names) >= 25
    handle = _lib.lib()
    for n in names:
        assert hasattr(handle, n), f"{n} declared in include/dm_hip.h but not exported"
    assert sorted(_lib.EXPORTED_SYMBOLS) == names, "ctypes prototypes and header are out of sync"
    header = open(os.path.join(ROOT, "include", "dm_hip.h")).read()
    assert handle.dm_abi_version() == int(re.search(r"#define DM_ABI_VERSION (\d+)", header).group(1))


def test_pack_order_is_a_permutation_with_lane_contiguous_columns():
    handle = _lib.lib()
    for n in (320, 4, 960, 1280):
        npad = (n + 127) // 128 * 128
        arr = (C.c_int * npad)()
        assert handle.dm_pack_weight_order(n, npad, 0, arr) == 0
        order = np.array(arr)
        assert sorted(order[order >= 0].tolist()) == list(range(n))
        # lane q of a 64-row group owns columns 16q .. 16q+15: rows {16j + 4q + r} -> 64g + 16q + 4j + r
        g = order[:64].reshape(4, 4, 4)  # [j][q][r]
        for q in range(4):
            cols = sorted(g[:, q, :].reshape(-1).tolist())
            want = [c for c in range(16 * q, 16 * q + 16) if c < n]
            assert [c for c in cols if c >= 0] == want
    arr = (C.c_int * 2560)()
    assert handle.dm_pack_weight_order(2560, 2560, 1, arr) == 0
    order = np.array(arr).reshape(-1, 4, 4, 4)  # [g][j][q][r]
    assert np.array_equal(order[:, 2:], order[:, :2] + 1280)  # gate rows sit two fragments after their value rows
    assert sorted(order.reshape(-1).tolist()) == list(range(2560))
    assert handle.dm_pack_weight_order(100, 100, 0, arr) != 0  # npad must be a multiple of 64


def test_gemm_plan_returns_supported_tiles():
    handle = _lib.lib()
    t, s = C.c_int(), C.c_int()
    for dt in (_lib.DM_F32, _lib.DM_BF16):
        for m, n, k in ((4096, 320, 2880), (64, 1280, 11520), (65536, 320, 320), (77, 320, 768), (1024, 5120, 640)):
            for epi in (_lib.EPI_PLAIN, _lib.EPI_GEGLU):
                handle.dm_gemm_plan(dt, m, n, k, epi, C.byref(t), C.byref(s))
                assert t.value in _lib.TILE_DIMS and 1 <= s.value <= 32
                assert epi == _lib.EPI_PLAIN or s.value == 1
                assert handle.dm_gemm_workspace_bytes(m, n, s.value) == (0 if s.value == 1 else s.value * m * n * 4)


def test_null_arguments_are_rejected_without_a_device():
    handle = _lib.lib()
    assert handle.dm_gemm(None, None) == 1
    assert handle.dm_ddpm_step(None, None, None, None, 16, None, None, 1, None) == 1
    assert handle.dm_attention(None, 0, None, 0, None, 0, None, 0, 1, 8, 64, 64, 40, 0, 0, None) == 1
    assert handle.dm_groupnorm_stats(None, 0, 1, 64, 320, 32, 0, None, 1, None) == 1
    # the entry points added later reject null pointers / empty problems the same way, before touching a device
    assert handle.dm_groupnorm_noalloc(None, 0, None, 0, 1, 64, 320, 32, 0, None, None, 1e-5, 1, None) == 1
    assert handle.dm_softmax_rows(None, 0, 4, 64, 1.0, 0, None) == 1
    assert handle.dm_pack_rows(None, 0, 64, 64, 0, None, 64, 64, 0, None) == 1
    assert handle.dm_vae_sample(None, None, None, 1, 64, 0.18215, None) == 1
    assert handle.dm_ksampler_input(None, None, 16, 2, None, None, None) == 1
    assert handle.dm_ksampler_step(None, None, None, None, 16, None, None, 7.5, 1, None) == 1
    assert handle.dm_groupnorm_default_nsplit(1, 4096) >= 1


# ------------------------------------------------------------------------------------------- schedules
def test_scheduler_tables_match_reference_goldens():
    g = load_golden("schedules")
    for T in (5, 50, 1000):
        for sched in ("linear", "cosine"):
            s = DDPMScheduler(num_train_timesteps=T, beta_schedule=sched)
            for k in ("betas", "alphas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
                      "sqrt_one_minus_alphas_cumprod", "sqrt_recip_alphas", "posterior_variance"):
                assert np.array_equal(getattr(s, k).numpy(), g[f"{sched}_{T}_{k}"]), (sched, T, k)
            assert len(s) == T
    with pytest.raises(ValueError):
        DDPMScheduler(beta_schedule="quadratic")
    s = DDPMScheduler()
    assert np.array_equal(s._get_timestep_embedding(torch.as_tensor(g["temb_t"]), 320).numpy(), g["temb"])
    assert s._get_timestep_embedding(torch.tensor([5]), 321).shape == (1, 321)
    assert np.array_equal(DDIMScheduler().timesteps.numpy(), g["ddim_timesteps_1000_50"])
    assert "DDPMScheduler" in repr(s) and "DDIMScheduler" in repr(DDIMScheduler())


def test_step_coefficients_are_the_reference_scalars():
    for T in (5, 50, 1000):
        s = DDPMScheduler(num_train_timesteps=T)
        tab = sched_ref.make_tables(T)
        coef = s.step_coefficients()
        assert coef.shape == (T, 5)
        for t in (0, 1, T // 2, T - 1):
            acp = tab["alphas_cumprod"][t]
            prev = tab["alphas_cumprod_prev"][t] if t > 0 else torch.tensor(1.0)
            want = [(1 - acp).sqrt(), acp.sqrt(), (prev.sqrt() * tab["betas"][t]) / (1 - acp),
                    (tab["alphas"][t].sqrt() * (1 - prev)) / (1 - acp),
                    tab["posterior_variance"][t].sqrt() if t > 0 else torch.tensor(0.0)]
            assert [float(c) for c in coef[t]] == [float(w) for w in want], (T, t)
    d = DDIMScheduler()
    assert float(d._get_variance(500, 480)) == float(sched_ref.ddim_variance(sched_ref.make_tables(1000), 500, 480))


def test_ksampler_tables_match_reference_goldens():
    g = load_golden("ksamplers")
    for n in (50, 20):
        for cls in (KEulerSampler, KEulerAncestralSampler, KLMSSampler):
            k = cls(n_inference_steps=n)
            assert np.array_equal(k.sigmas, g[f"sigmas_{n}"]) and np.array_equal(k.timesteps, g[f"timesteps_{n}"])
            assert k.initial_scale == g[f"sigmas_{n}"].max() and len(k) == n and k.step_count == 0
    k = KEulerSampler(50)
    assert np.array_equal(np.array([k.get_input_scale(i) for i in range(50)]), g["input_scale_50"])
    k.set_strength(0.6)
    assert k.step_count == 20 and len(k.timesteps) == 30 and k.initial_scale == k.sigmas[20]
    k.reset()
    assert k.step_count == 0
    lms = KLMSSampler(50)
    sig, _ = sched_ref.k_sigmas(50)
    assert lms._coefficients(3, 4) == sched_ref.k_lms_coeffs(sig, 3, 4)
    assert np.array_equal(get_alphas_cumprod(), sched_ref.k_alphas_cumprod())
    assert sigma_to_timestep(timestep_to_sigma(123)) == 123
    ks = get_sigmas_karras(10)
    assert ks.shape == (11,) and ks[-1] == 0 and abs(ks[0] - 10.0) < 1e-9 and abs(ks[-2] - 0.1) < 1e-9


# ------------------------------------------------------------------------------------------- front-ends
def test_state_dict_layout_and_constructor_contract():
    with torch.device("meta"):
        m = Diffusion()
    names = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    assert names == unet_ref.diffusion_shapes()
    assert len(list(m.buffers())) == 0
    assert m.time_embedding.linear_1.in_features == 320 and m.time_embedding.linear_2.out_features == 1280
    assert len(m.unet.encoders) == 12 and len(m.unet.decoders) == 12 and len(m.unet.bottleneck) == 3
    assert isinstance(ResidualBlock(64, 64).residual_layer, torch.nn.Identity)
    assert isinstance(ResidualBlock(64, 128).residual_layer, torch.nn.Conv2d)
    assert m.unet.encoders[1][1].group_norm.eps == 1e-6 and m.unet.encoders[1][0].groupnorm_feature.eps == 1e-5
    assert m.unet.encoders[1][1].attention_1.in_proj.bias is None and m.unet.encoders[1][1].attention_2.k_proj.bias is None


def test_cpu_inputs_fail_loudly():
    s = DDPMScheduler(num_train_timesteps=10)
    x = torch.zeros(1, 4, 8, 8)
    for call in (lambda: s.step(x, 3, x), lambda: s.add_noise(x, x, torch.tensor([1])),
                 lambda: s.sample(None, (1, 4, 8, 8), device="cpu"),
                 lambda: DDIMScheduler().step(x, 20, x),
                 lambda: KEulerSampler(10).step(x, x),
                 lambda: ResidualBlock(64, 64)(torch.zeros(1, 64, 8, 8), torch.zeros(1, 1280)),
                 lambda: TimeEmbedding(320)(torch.zeros(1, 320))):
        with pytest.raises(RuntimeError):
            call()


def test_missing_library_is_an_error(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", os.path.join(ROOT, "does_not_exist.so"))
    with pytest.raises(RuntimeError, match="not built"):
        _lib.lib()


# ------------------------------------------------------------------------------------------- helpers
def test_deterministic_fill_known_answers():
    """Pins the name-keyed weight generator: fixtures and both sides of every parity test depend on it."""
    v = unit_uniform("unet.encoders.0.0.weight", 4, seed=0)
    assert v.dtype == torch.float32 and float(v.abs().max()) < 1.0
    again = unit_uniform("unet.encoders.0.0.weight", 4, seed=0)
    assert torch.equal(v, again)
    assert not torch.equal(v, unit_uniform("unet.encoders.0.0.weight", 4, seed=1))
    assert not torch.equal(v, unit_uniform("unet.encoders.0.0.bias", 4, seed=0))
    big = unit_uniform("x", 1 << 16)
    assert abs(float(big.mean())) < 0.02 and abs(float(big.var()) - 1 / 3) < 0.02
    w = fill_value("a.weight", (8, 16))
    assert float(w.abs().max()) <= 0.25
    assert abs(float(fill_value("n.weight", (64,)).mean()) - 1.0) < 0.1
    g = load_golden("model")  # the committed reference outputs were produced with exactly this generator
    assert g["eps_32"].shape == (1, 4, 32, 32)


def test_arena_scoping():
    grown = []
    a = Arena("cpu", nbytes=4096, on_grow=lambda: grown.append(1))
    x = a.alloc(4, 8, torch.float32)
    m = a.mark()
    y = a.alloc(16, 8, torch.bfloat16)
    assert y.data_ptr() == x.data_ptr() + 256  # every allocation is rounded up to 256 bytes
    a.release(m)
    z = a.alloc(2, 2, torch.float32)
    assert z.data_ptr() == y.data_ptr()
    big = a.alloc(1024, 1024, torch.float32)  # grows; earlier views stay valid
    assert big.numel() == 1 << 20 and x.shape == (4, 8)
    assert grown == [1]  # holders of captured graphs learn that the addresses moved
    a.frozen = True
    with pytest.raises(RuntimeError):
        a.alloc(1 << 14, 1 << 12, torch.float32)
    a.reset()
    assert a.mark() == 0 and a.peak > 0


def test_shipped_tile_table_is_well_formed():
    """The committed tile table parses, names known tiles and covers both arithmetic types of the benchmark
    configurations (it fixes tile and work-list choices, hence the fp32 summation order, for every process)."""
    import ast
    import json
    import os

    from diffusion_model_amd import _lib

    path = os.path.join(os.path.dirname(_lib.__file__), "tune_table.json")
    with open(path) as f:
        table = json.load(f)
    assert set(table) == {"torch.float32", "torch.bfloat16"}
    for dtype, rows in table.items():
        assert len(rows) > 100
        for key, (tile, split) in rows.items():
            k = ast.literal_eval(key)
            # 13 fields, plus (fp32) the marker of the launches that normalise their A operand themselves or of the
            # opt-in split-bf16 product modes, or (bf16) of stride-1 convolution weights with the chunk-major k axis
            assert isinstance(k, tuple) and (len(k) == 13 or (len(k) == 14 and (
                (k[13] in ("gn_in", "w_split", "w_split3") and "float32" in dtype)
                or (k[13] == "chunk_major" and "bfloat16" in dtype and k[3] == 1 and k[4] == 1 and k[5] == 0))))
            assert k[0] > 0 and k[1] > 0 and k[2] > 0
            assert tile in _lib.TILE_DIMS
            assert split != 0
            if tile == _lib.TILE_D256x256_S2:
                assert dtype == "torch.bfloat16" and split <= 1
