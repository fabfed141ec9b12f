"""Kernel-level parity: every HIP kernel, called through the C ABI, against fp32 torch math on the CPU.

fp32 mode must agree to rounding (exact-f32 MFMA); bf16 mode is compared against the fp32
result computed from bf16-rounded inputs with a tolerance stated per test.
"""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import max_rel
from diffusion_model_amd import _lib
from diffusion_model_amd.engine import Engine
from diffusion_model_amd.weights import test_input as tin

pytestmark = pytest.mark.gpu

F32_TOL = 2e-5
BF16_TOL = 2e-2
DTYPES = [torch.float32, torch.bfloat16]


def tol(dt):
    return F32_TOL if dt == torch.float32 else BF16_TOL


def rnd(x, dt):
    """Round to the engine's storage dtype (what the kernel actually sees) and return fp32."""
    return x.to(dt).float()


def wide_refuses(eng, tile, stride=1, upx=0, cin=64):
    """The 256x256 ping-pong tile exists for bf16 on whole k-tiles and stride-1 same-size convolutions only; the
    library refuses it elsewhere (DM_ERR_ARG), which test_wide_tile_is_refused_where_unsupported checks."""
    return tile == _lib.TILE_D256x256_S2 and (eng.dtype != torch.bfloat16 or stride != 1 or upx or cin % 64)


def up(eng, x2d):
    """CPU fp32 [rows, C] -> engine-dtype device rows."""
    return x2d.to(eng.device).to(eng.dtype).contiguous()


@pytest.fixture(params=DTYPES, ids=["f32", "bf16"])
def eng(request, dev):
    e = Engine.get(dev, request.param)
    e.reset()
    e.tile_override = None
    return e


# ------------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("m,n,k", [(64, 320, 320), (77, 640, 768), (200, 1280, 640), (1024, 320, 1280), (130, 4, 64)])
@pytest.mark.parametrize("tile", (_lib.TILE_AUTO,) + _lib.ALL_TILES)
def test_gemm_dense(eng, m, n, k, tile):
    if wide_refuses(eng, tile):
        pytest.skip("wide tile: bf16 only")
    a = tin("g.a", (m, k)); w = tin("g.w", (n, k), scale=1 / math.sqrt(k)); b = tin("g.b", (n,)); r = tin("g.r", (m, n))
    lin = torch.nn.Linear(k, n)
    with torch.no_grad():
        lin.weight.copy_(w); lin.bias.copy_(b)
    pw = eng.pack_linear(lin)
    eng.tile_override = None if tile == _lib.TILE_AUTO else (tile, 1)
    # rows of C and R are accessed as 16-byte pieces: n = 4 lives in buffers with a leading dimension of 8
    ld = (n + 7) // 8 * 8
    cbuf = torch.zeros(m, ld, dtype=eng.dtype, device=eng.device)
    rbuf = torch.zeros(m, ld, dtype=eng.dtype, device=eng.device); rbuf[:, :n] = up(eng, r)
    out = eng.gemm(up(eng, a), pw, out=cbuf[:, :n], residual=rbuf[:, :n])
    ref = rnd(a, eng.dtype) @ rnd(w, eng.dtype).t() + b + rnd(r, eng.dtype)
    assert max_rel(out.float(), ref) < tol(eng.dtype)
    assert float(cbuf[:, n:].abs().max() if ld > n else 0.0) == 0.0


@pytest.mark.parametrize("splitk", [2, 3, 5])
def test_gemm_splitk_and_strides(eng, splitk):
    m, n, k = 96, 640, 2560
    a = tin("s.a", (m, k)); w = tin("s.w", (n, k), scale=1 / math.sqrt(k)); b = tin("s.b", (n,))
    rb = tin("s.rb", (3, n)); r = tin("s.r", (m, n))
    lin = torch.nn.Linear(k, n)
    with torch.no_grad():
        lin.weight.copy_(w); lin.bias.copy_(b)
    pw = eng.pack_linear(lin)
    # strided views: A inside a wider buffer, C and R as channel slices of wider buffers
    abuf = torch.zeros(m, k + 64, dtype=eng.dtype, device=eng.device); abuf[:, 32:32 + k] = up(eng, a)
    cbuf = torch.zeros(m, n + 128, dtype=eng.dtype, device=eng.device)
    rbuf = torch.zeros(m, n + 32, dtype=eng.dtype, device=eng.device); rbuf[:, 32:] = up(eng, r)
    eng.tile_override = (_lib.TILE_64x64, splitk)
    eng.gemm(abuf[:, 32:32 + k], pw, out=cbuf[:, 64:64 + n], rowbias=rb.to(eng.device), rows_per_batch=32,
             residual=rbuf[:, 32:])
    ref = rnd(a, eng.dtype) @ rnd(w, eng.dtype).t() + b + rb.repeat_interleave(32, 0) + rnd(r, eng.dtype)
    assert max_rel(cbuf[:, 64:64 + n].float(), ref) < tol(eng.dtype)
    assert float(cbuf[:, :64].abs().max()) == 0.0 and float(cbuf[:, 64 + n:].abs().max()) == 0.0


DMA_TILES = tuple(t for t in _lib.ALL_TILES if t >= _lib.TILE_D128x128_S2)


@pytest.mark.parametrize("tile", DMA_TILES)
@pytest.mark.parametrize("blocks", [1, 7, 64, 280, 300])  # 280, 300: whole multiples of the tile count (segment-major dealing)
def test_gemm_streamk(eng, tile, blocks):
    """Work-list form: `blocks` blocks share the (tile, k-tile) list; tiles cut between blocks are summed by
    the last block to arrive.  Two launches in a row: the arrival counters must come back to zero, and the
    result must not depend on arrival order (bitwise equal)."""
    if wide_refuses(eng, tile):
        pytest.skip("wide tile: bf16 only")
    m, n, k = 200, 640, 1280
    a = tin("k.a", (m, k)); w = tin("k.w", (n, k), scale=1 / math.sqrt(k)); b = tin("k.b", (n,))
    rb = tin("k.rb", (2, n)); r = tin("k.r", (m, n))
    lin = torch.nn.Linear(k, n)
    with torch.no_grad():
        lin.weight.copy_(w); lin.bias.copy_(b)
    pw = eng.pack_linear(lin)
    eng.tile_override = (tile, -blocks)
    ad, rd = up(eng, a), up(eng, r)
    outs = [eng.gemm(ad, pw, rowbias=rb.to(eng.device), rows_per_batch=100, residual=rd).clone() for _ in range(2)]
    ref = rnd(a, eng.dtype) @ rnd(w, eng.dtype).t() + b + rb.repeat_interleave(100, 0) + rnd(r, eng.dtype)
    assert max_rel(outs[0].float(), ref) < tol(eng.dtype)
    assert torch.equal(outs[0], outs[1])
    assert int(eng.ws_sk[:4096].max()) == 0


@pytest.mark.parametrize("blocks", [3, 50, 256])
def test_gemm_streamk_geglu_and_conv(eng, blocks):
    m, c = 150, 320
    a = tin("kg.a", (m, c)); w = tin("kg.w", (8 * c, c), scale=1 / math.sqrt(c)); b = tin("kg.b", (8 * c,))
    lin = torch.nn.Linear(c, 8 * c)
    with torch.no_grad():
        lin.weight.copy_(w); lin.bias.copy_(b)
    pw = eng.pack_linear(lin, geglu=True)
    eng.tile_override = (_lib.TILE_D64x128_S3, -blocks)
    out = eng.gemm(up(eng, a), pw)
    y = rnd(a, eng.dtype) @ rnd(w, eng.dtype).t() + b
    v, g = y.chunk(2, dim=-1)
    assert max_rel(out.float(), v * F.gelu(g)) < tol(eng.dtype)
    bsz = 2
    for cin, cout, h, wd, stride, upx in ((320, 320, 8, 8, 1, 0), (64, 128, 9, 7, 2, 0), (128, 64, 6, 5, 1, 1), (4, 320, 8, 8, 1, 0)):
        x = tin("kc.x", (bsz, cin, h, wd)); wt = tin("kc.w", (cout, cin, 3, 3), scale=1 / math.sqrt(9 * cin)); bias = tin("kc.b", (cout,))
        conv = torch.nn.Conv2d(cin, cout, 3, stride=stride, padding=1)
        with torch.no_grad():
            conv.weight.copy_(wt); conv.bias.copy_(bias)
        pc = eng.pack_conv3x3(conv)
        rows = eng.nchw_to_rows(x.to(eng.device), cpad=pc.conv_cin)
        xin = F.interpolate(rnd(x, eng.dtype), scale_factor=2, mode="nearest") if upx else rnd(x, eng.dtype)
        ref = F.conv2d(xin, rnd(wt, eng.dtype), bias, stride=stride, padding=1)
        ho, wo = ref.shape[2], ref.shape[3]
        for tile in (_lib.TILE_D128x64_S2, _lib.TILE_D32x64_S4):
            eng.tile_override = (tile, -blocks)
            out = eng.gemm(rows, pc, conv=dict(batch=bsz, hin=h, win=wd, hout=ho, wout=wo, stride=stride, up=upx, cin_real=cin))
            assert max_rel(eng.rows_to_nchw(out, bsz, cout, ho, wo), ref) < tol(eng.dtype), (cin, cout, stride, upx, tile)
    eng.tile_override = None
    assert int(eng.ws_sk[:4096].max()) == 0


@pytest.mark.parametrize("tile,sk", [(_lib.TILE_D128x128_S2, 1), (_lib.TILE_D128x128_S2, -50), (_lib.TILE_D256x128_S3, -100),
                                     (_lib.TILE_D64x128_S3, -37), (_lib.TILE_D64x64_S4, 1), (_lib.TILE_D32x64_S4, -9),
                                     (_lib.TILE_D256x64_S2, 1), (_lib.TILE_E128x128, -11), (_lib.TILE_E128x64, 1),
                                     (_lib.TILE_U64x64, 1), (_lib.TILE_U64x64, -29),
                                     (_lib.TILE_D256x256_S2, 1), (_lib.TILE_D256x256_S2, -23)])
@pytest.mark.parametrize("pointwise", [False, True])
def test_conv3x3_chunk_major_k_order(eng, tile, sk, pointwise):
    """Weights packed for stride-1 same-size use (k axis ordered channel chunk, then tap; DmGemmDesc.conv_chunk_major)
    against torch's convolution (blocks.py:60,70 are such convolutions) and against the tap-major image of the same
    weights; odd image sizes and three samples, so tiles cross image rows and samples and every border tap is masked
    somewhere.  Any other geometry is refused for such weights."""
    if eng.dtype != torch.bfloat16:
        conv = torch.nn.Conv2d(64, 64, 3, padding=1)
        assert not eng.pack_conv3x3(conv, same_size=True).chunk_major  # the fp32 engine keeps the tap-major image
        pytest.skip("chunk-major k axis: bf16 only")
    bsz, h, wd, cin, cout, c2 = 3, 9, 7, 192, 320, 64
    m = bsz * h * wd
    x = tin("cm.x", (bsz, cin, h, wd)); wt = tin("cm.w", (cout, cin, 3, 3), scale=1 / math.sqrt(9 * cin)); bias = tin("cm.b", (cout,))
    conv = torch.nn.Conv2d(cin, cout, 3, padding=1)
    with torch.no_grad():
        conv.weight.copy_(wt); conv.bias.copy_(bias)
    ref = F.conv2d(rnd(x, eng.dtype), rnd(wt, eng.dtype), bias, padding=1)
    skip, extra = None, {}
    if pointwise:
        skip = torch.nn.Conv2d(c2, cout, 1)
        ws = tin("cm.sw", (cout, c2, 1, 1), scale=1 / math.sqrt(c2)); bs = tin("cm.sb", (cout,)); s_in = tin("cm.s", (m, c2))
        with torch.no_grad():
            skip.weight.copy_(ws); skip.bias.copy_(bs)
        ref = ref + F.conv2d(rnd(s_in, eng.dtype).reshape(bsz, h, wd, c2).permute(0, 3, 1, 2), rnd(ws, eng.dtype), bs)
        extra = dict(skip=up(eng, s_in))
    rows = eng.nchw_to_rows(x.to(eng.device), cpad=cin)
    geom = dict(batch=bsz, hin=h, win=wd, hout=h, wout=wd, cin_real=cin, **extra)
    pw = eng.pack_conv3x3(conv, skip=skip, same_size=True)
    pw_tap = eng.pack_conv3x3(conv, skip=skip)
    assert pw.chunk_major and not pw_tap.chunk_major
    assert not torch.equal(pw.w, pw_tap.w)
    eng.tile_override = (tile, sk)
    try:
        out = eng.gemm(rows, pw, conv=geom)
        out_tap = eng.gemm(rows, pw_tap, conv=geom)
        with pytest.raises(RuntimeError):
            eng.gemm(rows, pw, conv=dict(geom, hout=(h + 1) // 2, wout=(wd + 1) // 2, stride=2))
    finally:
        eng.tile_override = None
    assert max_rel(eng.rows_to_nchw(out, bsz, cout, h, wd), ref) < tol(eng.dtype)
    # the same products summed in another order
    assert max_rel(out.float(), out_tap.float()) < (2e-6 if eng.dtype == torch.float32 else 1e-2)
    assert int(eng.ws_sk[:4096].max()) == 0


@pytest.mark.parametrize("geom", [(4, 8, 8), (2, 16, 16), (1, 32, 32), (1, 64, 64), (3, 16, 8)])
@pytest.mark.parametrize("sk", [1, -7, -50])
@pytest.mark.parametrize("pointwise", [False, True])
def test_conv3x3_halo_tile(eng, geom, sk, pointwise):
    """The tile that keeps a chunk's activation and its one-pixel border in LDS (DM_TILE_H128x128, weights in its own unit
    order) against torch's convolution (blocks.py:60,70): every image width it takes, tiles of several whole images, of
    several rows and of part of an image, work lists that cut a chunk anywhere, the pointwise operand behind the chunks."""
    if eng.dtype != torch.bfloat16:
        pytest.skip("halo tile: bf16 only")
    bsz, h, wd = geom
    cin, cout, c2 = 192, 320, 64
    m = bsz * h * wd
    x = tin("hl.x", (bsz, cin, h, wd)); wt = tin("hl.w", (cout, cin, 3, 3), scale=1 / math.sqrt(9 * cin)); bias = tin("hl.b", (cout,))
    conv = torch.nn.Conv2d(cin, cout, 3, padding=1)
    with torch.no_grad():
        conv.weight.copy_(wt); conv.bias.copy_(bias)
    ref = F.conv2d(rnd(x, eng.dtype), rnd(wt, eng.dtype), bias, padding=1)
    skip, extra = None, {}
    if pointwise:
        skip = torch.nn.Conv2d(c2, cout, 1)
        ws = tin("hl.sw", (cout, c2, 1, 1), scale=1 / math.sqrt(c2)); bs = tin("hl.sb", (cout,)); s_in = tin("hl.s", (m, c2))
        with torch.no_grad():
            skip.weight.copy_(ws); skip.bias.copy_(bs)
        ref = ref + F.conv2d(rnd(s_in, eng.dtype).reshape(bsz, h, wd, c2).permute(0, 3, 1, 2), rnd(ws, eng.dtype), bs)
        extra = dict(skip=up(eng, s_in))
    rows = eng.nchw_to_rows(x.to(eng.device), cpad=cin)
    g = dict(batch=bsz, hin=h, win=wd, hout=h, wout=wd, cin_real=cin, **extra)
    pw = eng.pack_conv3x3(conv, skip=skip, same_size=True)
    assert pw.w_halo is not None
    supported = bool(eng.lib.dm_conv_halo_supported(m, h, wd, cin, rows.stride(0)))
    assert supported == (m % 128 == 0 and wd <= 32)
    eng.tile_override = (_lib.TILE_H128x128, sk)
    try:
        if not supported:
            with pytest.raises(RuntimeError):
                eng.gemm(rows, pw, conv=g)
            return
        out = eng.gemm(rows, pw, conv=g)
        out2 = eng.gemm(rows, pw, conv=g)
    finally:
        eng.tile_override = None
    assert max_rel(eng.rows_to_nchw(out, bsz, cout, h, wd), ref) < tol(eng.dtype)
    assert torch.equal(out, out2)
    assert int(eng.ws_sk[:4096].max()) == 0


def test_conv3x3_halo_choice_falls_back(eng):
    """A table entry that names the halo tile is keyed by (m, n, k), not by image shape: a convolution with the same key on
    images the tile does not take, or on weights packed without the second image, runs on the ring tile instead."""
    if eng.dtype != torch.bfloat16:
        pytest.skip("halo tile: bf16 only")
    cin, cout = 64, 64
    conv = torch.nn.Conv2d(cin, cout, 3, padding=1)
    pw = eng.pack_conv3x3(conv, same_size=True)
    saved = dict(eng.tuned)
    try:
        for (bsz, h, wd) in ((1, 2, 128), (1, 16, 16)):  # 128 pixels wide: not taken; 16 x 16: taken
            m = bsz * h * wd
            x = torch.randn(bsz, cin, h, wd)
            rows = eng.nchw_to_rows(x.to(eng.device), cpad=cin)
            g = dict(batch=bsz, hin=h, win=wd, hout=h, wout=wd, cin_real=cin)
            key = (m, cout, 9 * cin, 1, 1, 0, _lib.EPI_PLAIN, False, False, False, False, False, 0, "chunk_major")
            eng.tuned[key] = (_lib.TILE_H128x128, 1)
            out = eng.gemm(rows, pw, conv=g)
            ref = F.conv2d(rnd(x, eng.dtype), rnd(conv.weight.detach(), eng.dtype), conv.bias.detach(), padding=1)
            assert max_rel(eng.rows_to_nchw(out, bsz, cout, h, wd), ref) < tol(eng.dtype)
            keep, pw.w_halo = pw.w_halo, None
            try:
                out = eng.gemm(rows, pw, conv=g)
            finally:
                pw.w_halo = keep
            assert max_rel(eng.rows_to_nchw(out, bsz, cout, h, wd), ref) < tol(eng.dtype)
    finally:
        eng.tuned.clear()
        eng.tuned.update(saved)


def test_conv3x3_chunk_major_needs_whole_k_tiles(eng):
    """Channel counts that are not whole k-tiles keep the tap-major image whatever the caller promises; the library
    refuses the chunk-major flag for fp32 problems."""
    if eng.dtype == torch.float32:
        conv = torch.nn.Conv2d(64, 64, 3, padding=1)
        pw = eng.pack_conv3x3(conv, same_size=True)
        pw.chunk_major = True  # force the flag into the descriptor
        x = torch.zeros(2 * 8 * 8, 64, device=eng.device)
        with pytest.raises(RuntimeError):
            eng.gemm(x, pw, conv=dict(batch=2, hin=8, win=8, hout=8, wout=8, cin_real=64))
    conv = torch.nn.Conv2d(4, 320, 3, padding=1)
    assert not eng.pack_conv3x3(conv, same_size=True).chunk_major
    conv = torch.nn.Conv2d(48, 64, 3, padding=1)
    assert not eng.pack_conv3x3(conv, same_size=True).chunk_major


@pytest.mark.parametrize("tile,sk", [(_lib.TILE_D128x128_S2, 1), (_lib.TILE_D256x128_S3, -100), (_lib.TILE_D64x64_S2, -37),
                                     (_lib.TILE_D128x64_S2, 1), (_lib.TILE_64x64, 1), (_lib.TILE_D32x64_S4, -9),
                                     (_lib.TILE_D256x256_S2, 1), (_lib.TILE_D256x256_S2, -23)])
def test_gemm_fused_groupnorm_statistics(eng, tile, sk):
    """The GEMM accumulates per-channel sums of what it stores; GroupNorm then needs no statistics pass.
    Two producers fill the two channel slices of one buffer (the skip-concat case)."""
    if wide_refuses(eng, tile):
        pytest.skip("wide tile: bf16 only")
    b, hw, k, n1, n2 = 3, 64, 640, 320, 160
    m = b * hw
    a = tin("fs.a", (m, k)); w1 = tin("fs.w1", (n1, k), scale=1 / math.sqrt(k)); w2 = tin("fs.w2", (n2, k), scale=1 / math.sqrt(k))
    b1 = tin("fs.b1", (n1,)) + 0.5; b2 = tin("fs.b2", (n2,)) - 1.0
    gamma = tin("fs.g", (n1 + n2,)) + 1; beta = tin("fs.be", (n1 + n2,))
    lins = []
    for w, bb in ((w1, b1), (w2, b2)):
        lin = torch.nn.Linear(k, w.shape[0])
        with torch.no_grad():
            lin.weight.copy_(w); lin.bias.copy_(bb)
        lins.append(eng.pack_linear(lin))
    buf = eng.new(m, n1 + n2)
    eng.gn_register(buf, b)
    eng.tile_override = (tile, sk)
    ad = up(eng, a)
    eng.gemm(ad, lins[0], out=buf[:, :n1], gn_batch=b)
    eng.gemm(ad, lins[1], out=buf[:, n1:], gn_batch=b)
    eng.tile_override = None
    assert eng._gn_lookup(buf, b) is not None, "statistics were not fused"
    out = eng.groupnorm(buf, b, hw, gamma.to(eng.device), beta.to(eng.device), 1e-5, True)
    y = torch.cat([rnd(a, eng.dtype) @ rnd(w1, eng.dtype).t() + b1, rnd(a, eng.dtype) @ rnd(w2, eng.dtype).t() + b2], 1)
    y = rnd(y, eng.dtype).reshape(b, hw, n1 + n2).permute(0, 2, 1)
    ref = F.silu(F.group_norm(y, 32, gamma, beta, eps=1e-5)).permute(0, 2, 1).reshape(m, n1 + n2)
    assert max_rel(out.float(), ref) < (2e-5 if eng.dtype == torch.float32 else 2e-2)
    # the right-hand slice is also normalised on its own (an encoder output read by the next encoder stage): the same
    # launch accumulated a second partition of its columns into groups of 160 / 32 channels
    assert eng._gn_lookup(buf[:, n1:], b) is not None, "the slice's own statistics were not fused"
    out2 = eng.groupnorm(buf[:, n1:], b, hw, gamma[n1:].contiguous().to(eng.device), beta[n1:].contiguous().to(eng.device), 1e-5, False)
    ref2 = F.group_norm(y[:, n1:], 32, gamma[n1:], beta[n1:], eps=1e-5).permute(0, 2, 1).reshape(m, n2)
    assert max_rel(out2.float(), ref2) < (2e-5 if eng.dtype == torch.float32 else 2e-2)


GN_TILES = (_lib.TILE_D128x128_S2, _lib.TILE_D64x128_S3, _lib.TILE_D64x64_S4, _lib.TILE_D32x64_S4, _lib.TILE_D128x64_S2,
            _lib.TILE_D64x64_S2)


@pytest.mark.parametrize("tile", GN_TILES)
@pytest.mark.parametrize("sk", [1, -7, -60])
@pytest.mark.parametrize("kind", ["conv", "conv+pointwise", "dense"])
def test_gemm_groupnorm_on_a_fragments(dev, tile, sk, kind):
    """fp32: GroupNorm (+ SiLU) of the operand applied inside the consuming launch (blocks.py:58-59,68-69,128) from the
    statistics the producing launch accumulated; padding taps stay zero, the appended pointwise operand stays raw.
    Three samples, so work-list segments cross sample boundaries and the scale / shift table is rebuilt."""
    eng = Engine.get(dev, torch.float32)
    eng.reset()
    b, h, w, k0, c, n = 3, 16, 8, 64, 320, 192
    hw, m = h * w, b * h * w
    a0 = tin("gnf.a0", (m, k0)); w0 = tin("gnf.w0", (c, k0), scale=1 / math.sqrt(k0)); b0 = tin("gnf.b0", (c,)) + 0.3
    gamma = tin("gnf.g", (c,)) + 1; beta = tin("gnf.be", (c,)) * 0.5
    lin0 = torch.nn.Linear(k0, c)
    with torch.no_grad():
        lin0.weight.copy_(w0); lin0.bias.copy_(b0)
    x = eng.gemm(up(eng, a0), eng.pack_linear(lin0), gn_batch=b)  # producer: writes x and its per-channel sums
    assert eng._gn_lookup(x, b) is not None
    x_ref = a0 @ w0.t() + b0
    xn = F.group_norm(x_ref.reshape(b, hw, c).permute(0, 2, 1), 32, gamma, beta, eps=1e-5)  # (b, c, hw)
    gn = dict(batch=b, hw=hw, gamma=gamma.to(dev), beta=beta.to(dev), eps=1e-5, silu=kind != "dense")
    eng.tile_override = (tile, sk)

    def no_fallback(*_):
        raise AssertionError("the fragment-side form did not run (fell back to the GroupNorm kernel)")

    eng._gn_unfused = no_fallback
    max_n, eng.fuse_gn_conv_max_n = eng.fuse_gn_conv_max_n, 1 << 20  # the engine's default keeps wide convolutions unfused
    try:
        if kind == "dense":
            w1 = tin("gnf.w1", (n, c), scale=1 / math.sqrt(c)); b1 = tin("gnf.b1", (n,))
            lin = torch.nn.Linear(c, n)
            with torch.no_grad():
                lin.weight.copy_(w1); lin.bias.copy_(b1)
            out = eng.gemm(x, eng.pack_linear(lin), gn_in=gn)
            ref = xn.permute(0, 2, 1).reshape(m, c) @ w1.t() + b1
        else:
            conv = torch.nn.Conv2d(c, n, 3, padding=1)
            w1 = tin("gnf.cw", (n, c, 3, 3), scale=1 / math.sqrt(9 * c)); b1 = tin("gnf.cb", (n,))
            with torch.no_grad():
                conv.weight.copy_(w1); conv.bias.copy_(b1)
            img = F.silu(xn).reshape(b, c, h, w)
            ref = F.conv2d(img, w1, b1, padding=1)
            geom = dict(batch=b, hin=h, win=w, hout=h, wout=w, cin_real=c)
            if kind == "conv+pointwise":
                c2 = 64
                skip = torch.nn.Conv2d(c2, n, 1)
                ws = tin("gnf.sw", (n, c2, 1, 1), scale=1 / math.sqrt(c2)); bs = tin("gnf.sb", (n,))
                with torch.no_grad():
                    skip.weight.copy_(ws); skip.bias.copy_(bs)
                s_in = tin("gnf.s", (m, c2))
                ref = ref + F.conv2d(s_in.reshape(b, h, w, c2).permute(0, 3, 1, 2), ws, bs)
                # same_size: what the model passes for these convolutions; the fp32 engine keeps the tap-major image
                pw = eng.pack_conv3x3(conv, skip=skip, same_size=True)
                out = eng.gemm(x, pw, conv=dict(geom, skip=up(eng, s_in)), gn_in=gn)
            else:
                pw = eng.pack_conv3x3(conv, same_size=True)
                out = eng.gemm(x, pw, conv=geom, gn_in=gn)
            assert not pw.chunk_major
            ref = ref.permute(0, 2, 3, 1).reshape(m, n)
    finally:
        eng.tile_override = None
        eng.fuse_gn_conv_max_n = max_n
        del eng._gn_unfused
    assert max_rel(out.float().cpu(), ref) < 2e-5


SPLIT_TILES = (_lib.TILE_D128x128_S2, _lib.TILE_D64x128_S3, _lib.TILE_D64x64_S4, _lib.TILE_D32x64_S4, _lib.TILE_D256x64_S2,
               _lib.TILE_D128x64_S2, _lib.TILE_D64x64_S2)


@pytest.mark.parametrize("tile", SPLIT_TILES)
@pytest.mark.parametrize("sk", [1, -9, -70])
@pytest.mark.parametrize("kind", ["dense", "dense+second", "conv", "conv+pointwise"])
@pytest.mark.parametrize("terms", [2, 3])
def test_gemm_split_bf16_weight_mode(dev, tile, sk, kind, terms):
    """The fp32 engine's opt-in product mode (Engine.split_bf16): weights packed as bf16 (hi | lo) pairs, activations
    split in registers, three bf16 MFMA products per fp32 product.  Not the reference's arithmetic: about 16 bits of
    each operand, so the bound here is 3e-5 of the output range instead of the 2e-5 of the exact-fp32 kernels.  With
    three terms per operand and six products the result is held to 3e-6: what that form drops is the size of an fp32
    rounding, and the exact-fp32 kernel itself is 0.6-1.4e-6 from an fp64 product on these shapes."""
    eng = Engine.get(dev, torch.float32)
    eng.reset()
    prev, eng.split_bf16 = eng.split_bf16, terms
    eng.tile_override = (tile, sk)
    try:
        if kind.startswith("dense"):
            m, k, n = 200, 640, 320
            a = tin("sp.a", (m, k)); w = tin("sp.w", (n, k), scale=1 / math.sqrt(k)); b = tin("sp.b", (n,))
            r = tin("sp.r", (m, n))
            lin = torch.nn.Linear(k, n)
            with torch.no_grad():
                lin.weight.copy_(w); lin.bias.copy_(b)
            pw = eng.pack_linear(lin)
            assert pw.split == terms
            if kind == "dense+second":
                pw.skip_cin = 128  # the last 128 columns of the logical A row live in a second tensor
                out = eng.gemm(up(eng, a[:, :k - 128]), pw, a2=up(eng, a[:, k - 128:]), residual=up(eng, r))
            else:
                out = eng.gemm(up(eng, a), pw, residual=up(eng, r))
            ref = (a.double() @ w.double().t() + b.double() + r.double()).float()
        else:
            bsz, h, wd, c, n = 2, 9, 8, 64, 128
            m = bsz * h * wd
            a = tin("sp.ca", (m, c)); w = tin("sp.cw", (n, c, 3, 3), scale=1 / math.sqrt(9 * c)); b = tin("sp.cb", (n,))
            conv = torch.nn.Conv2d(c, n, 3, padding=1)
            with torch.no_grad():
                conv.weight.copy_(w); conv.bias.copy_(b)
            ref = F.conv2d(a.double().reshape(bsz, h, wd, c).permute(0, 3, 1, 2), w.double(), b.double(), padding=1)
            geom = dict(batch=bsz, hin=h, win=wd, hout=h, wout=wd, cin_real=c)
            if kind == "conv+pointwise":
                c2 = 32
                ws = tin("sp.sw", (n, c2, 1, 1), scale=1 / math.sqrt(c2)); s_in = tin("sp.s", (m, c2))
                skip = torch.nn.Conv2d(c2, n, 1, bias=False)
                with torch.no_grad():
                    skip.weight.copy_(ws)
                ref = ref + F.conv2d(s_in.double().reshape(bsz, h, wd, c2).permute(0, 3, 1, 2), ws.double())
                out = eng.gemm(up(eng, a), eng.pack_conv3x3(conv, skip=skip), conv=dict(geom, skip=up(eng, s_in)))
            else:
                out = eng.gemm(up(eng, a), eng.pack_conv3x3(conv), conv=geom)
            ref = ref.permute(0, 2, 3, 1).reshape(m, n).float()
    finally:
        eng.tile_override = None
        eng.split_bf16 = prev
    assert max_rel(out.float().cpu(), ref) < (3e-5 if terms == 2 else 3e-6)


@pytest.mark.parametrize("geglu", [False, True])
@pytest.mark.parametrize("tile,sk", [(_lib.TILE_AUTO, 1), (_lib.TILE_D128x128_S2, -40), (_lib.TILE_D64x64_S2, 1),
                                     (_lib.TILE_D256x256_S2, 1), (_lib.TILE_D256x256_S2, -40)])
def test_gemm_folded_layernorm(eng, geglu, tile, sk):
    """LayerNorm between two GEMMs never runs as a kernel: the first GEMM accumulates row sums of what it stores,
    the second (weights pre-scaled by gamma, bias + beta W^T) applies mean / rstd in its epilogue."""
    if wide_refuses(eng, tile):
        pytest.skip("wide tile: bf16 only")
    m, k0, c = 150, 256, 320
    n = 8 * c if geglu else 3 * c
    a = tin("fl.a", (m, k0)); w0 = tin("fl.w0", (c, k0), scale=1 / math.sqrt(k0)); b0 = tin("fl.b0", (c,)) + 0.7
    r = tin("fl.r", (m, c)) * 2
    gamma = tin("fl.g", (c,)) + 1; beta = tin("fl.be", (c,))
    w1 = tin("fl.w1", (n, c), scale=1 / math.sqrt(c)); b1 = tin("fl.b1", (n,))
    l0 = torch.nn.Linear(k0, c); l1 = torch.nn.Linear(c, n); ln = torch.nn.LayerNorm(c, eps=1e-5)
    with torch.no_grad():
        l0.weight.copy_(w0); l0.bias.copy_(b0); l1.weight.copy_(w1); l1.bias.copy_(b1)
        ln.weight.copy_(gamma); ln.bias.copy_(beta)
    p0 = eng.pack_linear(l0); p1 = eng.pack_linear(l1, geglu=geglu, ln=ln)
    eng.tile_override = None if tile == _lib.TILE_AUTO else (tile, sk)
    t = eng.gemm(up(eng, a), p0, residual=up(eng, r), ln_out=True)
    out = eng.gemm(t, p1, ln_in=eng.ln_stats(t))
    eng.tile_override = None
    tt = rnd(rnd(a, eng.dtype) @ rnd(w0, eng.dtype).t() + b0 + rnd(r, eng.dtype), eng.dtype)
    assert max_rel(t.float(), tt) < tol(eng.dtype)
    y = F.linear(F.layer_norm(tt, (c,), gamma, beta, 1e-5), w1, b1)
    if geglu:
        v, g = y.chunk(2, dim=-1)
        y = v * F.gelu(g)
    assert max_rel(out.float(), y) < (5e-5 if eng.dtype == torch.float32 else 3e-2)
    with pytest.raises(RuntimeError):
        eng.gemm(t, p1)  # weights packed with a LayerNorm are useless without the row statistics


def test_gemm_geglu(eng):
    m, c = 150, 320
    a = tin("gg.a", (m, c)); w = tin("gg.w", (8 * c, c), scale=1 / math.sqrt(c)); b = tin("gg.b", (8 * c,))
    lin = torch.nn.Linear(c, 8 * c)
    with torch.no_grad():
        lin.weight.copy_(w); lin.bias.copy_(b)
    pw = eng.pack_linear(lin, geglu=True)
    out = eng.gemm(up(eng, a), pw)
    y = rnd(a, eng.dtype) @ rnd(w, eng.dtype).t() + b
    v, g = y.chunk(2, dim=-1)
    assert out.shape == (m, 4 * c)
    assert max_rel(out.float(), v * F.gelu(g)) < tol(eng.dtype)


@pytest.mark.parametrize("tile,sk", [(_lib.TILE_AUTO, 1)] + [(t, sk) for t in DMA_TILES for sk in (1, -19)])
def test_gemm_second_operand_and_chain(eng, tile, sk):
    if wide_refuses(eng, tile):
        pytest.skip("wide tile: bf16 only")
    """second(first(a) + r) + x as one GEMM over the row [a | r] (linear_geglu_2 -> += residue -> conv_output):
    the k axis is read from two tensors, the seam on a k-tile boundary; work-list splits cross it."""
    m, c = 200, 128
    a = tin("ch.a", (m, 4 * c)); r = tin("ch.r", (m, c)); x = tin("ch.x", (m, c))
    first = torch.nn.Linear(4 * c, c); second = torch.nn.Conv2d(c, c, 1)
    with torch.no_grad():
        first.weight.copy_(tin("ch.w1", (c, 4 * c), scale=1 / math.sqrt(4 * c))); first.bias.copy_(tin("ch.b1", (c,)))
        second.weight.copy_(tin("ch.w2", (c, c, 1, 1), scale=1 / math.sqrt(c))); second.bias.copy_(tin("ch.b2", (c,)))
    pw = eng.pack_chain(first, second)
    assert pw.k == 5 * c and pw.skip_cin == c
    w2 = second.weight.detach().reshape(c, c).double()
    ref = ((rnd(a, eng.dtype).double() @ first.weight.detach().double().T + first.bias.detach().double()
            + rnd(r, eng.dtype).double()) @ w2.T + second.bias.detach().double() + rnd(x, eng.dtype).double())
    dev = eng.device
    eng.tile_override = None if tile == _lib.TILE_AUTO else (tile, sk)
    out = eng.gemm(a.to(dev).to(eng.dtype), pw, a2=r.to(dev).to(eng.dtype), residual=x.to(dev).to(eng.dtype))
    eng.tile_override = None
    assert max_rel(out.float(), ref) < tol(eng.dtype)
    with pytest.raises(RuntimeError):  # weights packed for two operands need both
        eng.gemm(a.to(dev).to(eng.dtype), pw)


@pytest.mark.parametrize("cin,cout,h,w,stride,upx", [(320, 320, 8, 8, 1, 0), (64, 128, 9, 7, 1, 0), (320, 320, 16, 16, 2, 0),
                                                      (64, 64, 7, 10, 2, 0), (128, 64, 6, 5, 1, 1), (4, 320, 8, 8, 1, 0)])
def test_conv3x3(eng, cin, cout, h, w, stride, upx):
    b = 2
    x = tin("c.x", (b, cin, h, w)); wt = tin("c.w", (cout, cin, 3, 3), scale=1 / math.sqrt(9 * cin)); bias = tin("c.b", (cout,))
    conv = torch.nn.Conv2d(cin, cout, 3, stride=stride, padding=1)
    with torch.no_grad():
        conv.weight.copy_(wt); conv.bias.copy_(bias)
    pw = eng.pack_conv3x3(conv)
    rows = eng.nchw_to_rows(x.to(eng.device), cpad=pw.conv_cin)
    xin = rnd(x, eng.dtype)
    if upx:
        xin = F.interpolate(xin, scale_factor=2, mode="nearest")
    ref = F.conv2d(xin, rnd(wt, eng.dtype), bias, stride=stride, padding=1)
    ho, wo = ref.shape[2], ref.shape[3]
    out = eng.gemm(rows, pw, conv=dict(batch=b, hin=h, win=w, hout=ho, wout=wo, stride=stride, up=upx, cin_real=cin))
    got = eng.rows_to_nchw(out, b, cout, ho, wo)
    assert max_rel(got, ref) < tol(eng.dtype)


@pytest.mark.parametrize("tile", _lib.ALL_TILES)
@pytest.mark.parametrize("sk", [1, 3])
def test_conv3x3_every_tile(eng, tile, sk):
    b = 2
    for cin, cout, h, w, stride, upx in ((320, 320, 8, 8, 1, 0), (64, 128, 9, 7, 2, 0), (128, 64, 6, 5, 1, 1), (4, 320, 8, 8, 1, 0)):
        x = tin("ct.x", (b, cin, h, w)); wt = tin("ct.w", (cout, cin, 3, 3), scale=1 / math.sqrt(9 * cin)); bias = tin("ct.b", (cout,))
        conv = torch.nn.Conv2d(cin, cout, 3, stride=stride, padding=1)
        with torch.no_grad():
            conv.weight.copy_(wt); conv.bias.copy_(bias)
        if wide_refuses(eng, tile, stride, upx, cin) or (tile == _lib.TILE_D256x256_S2 and sk > 1):
            continue  # (no separate split-K reduction launch for the wide tile either)
        pw = eng.pack_conv3x3(conv)
        rows = eng.nchw_to_rows(x.to(eng.device), cpad=pw.conv_cin)
        xin = F.interpolate(rnd(x, eng.dtype), scale_factor=2, mode="nearest") if upx else rnd(x, eng.dtype)
        ref = F.conv2d(xin, rnd(wt, eng.dtype), bias, stride=stride, padding=1)
        ho, wo = ref.shape[2], ref.shape[3]
        eng.tile_override = (tile, sk)
        out = eng.gemm(rows, pw, conv=dict(batch=b, hin=h, win=w, hout=ho, wout=wo, stride=stride, up=upx, cin_real=cin))
        eng.tile_override = None
        assert max_rel(eng.rows_to_nchw(out, b, cout, ho, wo), ref) < tol(eng.dtype), (cin, cout, stride, upx)


@pytest.mark.parametrize("tile,sk", [(_lib.TILE_AUTO, 1)] + [(t, sk) for t in DMA_TILES for sk in (1, -24, -61)])
def test_conv3x3_with_pointwise_term(eng, tile, sk):
    """conv3x3(a) + conv1x1(x) in one launch (the residual_layer of a ResidualBlock riding in its second conv):
    k runs over the nine taps and then over the second tensor's channels; work-list splits may start or end
    inside either part."""
    if wide_refuses(eng, tile):
        pytest.skip("wide tile: bf16 only")
    b, h, w = 2, 8, 6
    for cin, cin2, cout in ((64, 128, 64), (128, 64, 192)):
        a = tin("cp.a", (b, cin, h, w)); x = tin("cp.x", (b, cin2, h, w))
        wt = tin("cp.w", (cout, cin, 3, 3), scale=1 / math.sqrt(9 * cin)); bias = tin("cp.b", (cout,))
        w1 = tin("cp.w1", (cout, cin2, 1, 1), scale=1 / math.sqrt(cin2)); b1 = tin("cp.b1", (cout,))
        conv = torch.nn.Conv2d(cin, cout, 3, padding=1); skip = torch.nn.Conv2d(cin2, cout, 1)
        with torch.no_grad():
            conv.weight.copy_(wt); conv.bias.copy_(bias); skip.weight.copy_(w1); skip.bias.copy_(b1)
        pw = eng.pack_conv3x3(conv, skip=skip)
        assert pw.skip_cin == cin2 and pw.k == 9 * cin + cin2
        arows = eng.nchw_to_rows(a.to(eng.device)); xrows = eng.nchw_to_rows(x.to(eng.device))
        ref = (F.conv2d(rnd(a, eng.dtype), rnd(wt, eng.dtype), bias, padding=1)
               + F.conv2d(rnd(x, eng.dtype), rnd(w1, eng.dtype), b1))
        eng.tile_override = None if tile == _lib.TILE_AUTO else (tile, sk)
        out = eng.gemm(arows, pw, conv=dict(batch=b, hin=h, win=w, hout=h, wout=w, cin_real=cin, skip=xrows))
        eng.tile_override = None
        assert max_rel(eng.rows_to_nchw(out, b, cout, h, w), ref) < tol(eng.dtype), (cin, cin2, cout)
    # the descriptor is refused where the term cannot be honoured: missing input, staged tiles, strided conv
    with pytest.raises(RuntimeError):
        eng.gemm(arows, pw, conv=dict(batch=b, hin=h, win=w, hout=h, wout=w, cin_real=cin))
    eng.tile_override = (_lib.TILE_64x64, 1)
    with pytest.raises(RuntimeError):
        eng.gemm(arows, pw, conv=dict(batch=b, hin=h, win=w, hout=h, wout=w, cin_real=cin, skip=xrows))
    eng.tile_override = None


def test_conv3x3_nchw_epilogue_and_rowbias(eng):
    b, cin, cout, h, w = 2, 320, 4, 8, 8
    x = tin("n.x", (b, cin, h, w)); wt = tin("n.w", (cout, cin, 3, 3), scale=0.02); bias = tin("n.b", (cout,))
    conv = torch.nn.Conv2d(cin, cout, 3, padding=1)
    with torch.no_grad():
        conv.weight.copy_(wt); conv.bias.copy_(bias)
    pw = eng.pack_conv3x3(conv)
    rows = eng.nchw_to_rows(x.to(eng.device))
    out = torch.empty(b, cout, h, w, device=eng.device)
    eng.gemm(rows, pw, nchw_out=out, rows_per_batch=h * w, conv=dict(batch=b, hin=h, win=w, hout=h, wout=w, cin_real=cin))
    ref = F.conv2d(rnd(x, eng.dtype), rnd(wt, eng.dtype), bias, padding=1)
    assert max_rel(out, ref) < tol(eng.dtype)


def test_gemm_rejects_bad_arguments(eng):
    lin = torch.nn.Linear(320, 320)
    pw = eng.pack_linear(lin)
    with pytest.raises(RuntimeError):
        eng.gemm(torch.zeros(8, 64, dtype=eng.dtype, device=eng.device), pw)  # K mismatch
    a = torch.zeros(8, 321, dtype=eng.dtype, device=eng.device)[:, 1:]          # misaligned view
    with pytest.raises(RuntimeError):
        eng.gemm(a, pw)


# ------------------------------------------------------------------------------------------- norms
@pytest.mark.parametrize("c,hw,b", [(320, 64, 2), (640, 256, 1), (960, 100, 2), (1920, 64, 1), (2560, 16, 3), (64, 37, 2)])
@pytest.mark.parametrize("silu", [True, False])
def test_groupnorm(eng, c, hw, b, silu):
    x = tin("gn.x", (b, c, hw, 1)) * 2 + 0.5
    gamma = tin("gn.g", (c,)) + 1; beta = tin("gn.b", (c,))
    rows = eng.nchw_to_rows(x.to(eng.device))
    out = eng.groupnorm(rows, b, hw, gamma.to(eng.device), beta.to(eng.device), 1e-6, silu)
    ref = F.group_norm(rnd(x, eng.dtype), 32, gamma, beta, eps=1e-6)
    if silu:
        ref = F.silu(ref)
    got = eng.rows_to_nchw(out, b, c, hw, 1)
    assert max_rel(got, ref) < (1e-5 if eng.dtype == torch.float32 else 1e-2)


@pytest.mark.parametrize("c,rows", [(320, 130), (640, 64), (1280, 33), (768, 77)])
def test_layernorm(eng, c, rows):
    x = tin("ln.x", (rows, c)) * 3 - 1
    gamma = tin("ln.g", (c,)) + 1; beta = tin("ln.b", (c,))
    out = eng.layernorm(up(eng, x), gamma.to(eng.device), beta.to(eng.device))
    ref = F.layer_norm(rnd(x, eng.dtype), (c,), gamma, beta)
    assert max_rel(out.float(), ref) < (1e-5 if eng.dtype == torch.float32 else 1e-2)


# ------------------------------------------------------------------------------------------- attention
def ref_attention(q, k, v, heads):
    b, sq, c = q.shape
    d = c // heads
    qh, kh, vh = (t.reshape(b, -1, heads, d).transpose(1, 2) for t in (q, k, v))
    w = torch.softmax(qh @ kh.transpose(-1, -2) / math.sqrt(d), dim=-1)
    return (w @ vh).transpose(1, 2).reshape(b, sq, c)


@pytest.mark.parametrize("d,sq,sk", [(40, 256, 256), (40, 64, 77), (80, 100, 100), (80, 256, 77), (160, 64, 64),
                                     (160, 16, 16), (160, 200, 77), (64, 77, 77)])
def test_attention(eng, d, sq, sk):
    b, heads = 2, 8
    c = heads * d
    q = tin("at.q", (b, sq, c)); k = tin("at.k", (b, sk, c)); v = tin("at.v", (b, sk, c))
    # one query row whose scores spike on a late key: forces the online-softmax rescale path
    k[:, sk - 3] *= 6.0
    qr, kr, vr = (up(eng, t.reshape(-1, c)) for t in (q, k, v))
    out = eng.attention(qr, kr, vr, b, heads, sq, sk, d)
    ref = ref_attention(rnd(q, eng.dtype), rnd(k, eng.dtype), rnd(v, eng.dtype), heads)
    assert max_rel(out.float().reshape(b, sq, c), ref) < (2e-5 if eng.dtype == torch.float32 else 2e-2)


def test_attention_many_query_tiles_long_key_loop(eng):
    """d = 40 with at least 4096 64-row query tiles and 1024+ keys: the shape class for which the bf16 launcher takes 64
    query rows per wave (blocks.py:26-31 at the 64 x 64 level from batch 8 up).  Ragged query and key tails, a late spike
    that forces the online-softmax rescale."""
    b, heads, d, sq, sk = 2, 8, 40, 16390, 1030
    c = heads * d
    q = tin("atl.q", (b, sq, c)); k = tin("atl.k", (b, sk, c)); v = tin("atl.v", (b, sk, c))
    k[:, sk - 3] *= 6.0
    qr, kr, vr = (up(eng, t.reshape(-1, c)) for t in (q, k, v))
    out = eng.attention(qr, kr, vr, b, heads, sq, sk, d)
    ref = ref_attention(rnd(q, eng.dtype), rnd(k, eng.dtype), rnd(v, eng.dtype), heads)
    assert max_rel(out.float().reshape(b, sq, c), ref) < (2e-5 if eng.dtype == torch.float32 else 2e-2)


# ------------------------------------------------------------------------------------------- time path
@pytest.mark.parametrize("d,sq,sk", [(40, 256, 256), (40, 64, 77), (80, 100, 100), (80, 256, 77), (160, 320, 300), (64, 130, 45)])
@pytest.mark.parametrize("causal", [False, True])
def test_attention_split_bf16_mode(dev, d, sq, sk, causal):
    """fp32 attention with both products as split-bf16 products (the engine's opt-in mode): masked key tails, causal
    masks, every head dimension.  Bound: 1e-4 of the output range (measured 1-3.5e-5; the exact-fp32 kernel is held to
    2e-5): the inputs here give scores of up to +-20, and a score that is off by 2^-16 of its size moves its
    probability by that much."""
    if causal and sq != sk:
        pytest.skip("causal attention is square")
    eng = Engine.get(dev, torch.float32)
    eng.reset()
    heads, b = 2, 2
    q = tin("as.q", (b * sq, heads * d)) * 2; k = tin("as.k", (b * sk, heads * d)) * 2; v = tin("as.v", (b * sk, heads * d))
    prev, eng.split_bf16 = eng.split_bf16, True
    try:
        out = eng.attention(up(eng, q), up(eng, k), up(eng, v), b, heads, sq, sk, d, causal=causal)
    finally:
        eng.split_bf16 = prev
    qq = q.double().reshape(b, sq, heads, d).transpose(1, 2); kk = k.double().reshape(b, sk, heads, d).transpose(1, 2)
    vv = v.double().reshape(b, sk, heads, d).transpose(1, 2)
    w = qq @ kk.transpose(-1, -2) / math.sqrt(d)
    if causal:
        w = w.masked_fill(torch.ones(sq, sk, dtype=torch.bool).triu(1), float("-inf"))
    ref = (torch.softmax(w, -1) @ vv).transpose(1, 2).reshape(b * sq, heads * d).float()
    assert max_rel(out.float().cpu(), ref) < 1e-4


def test_linear_rows(dev):
    eng = Engine.get(dev, torch.float32)
    eng.arena.reset()
    for batch in (1, 3, 16):
        x = tin("lr.x", (batch, 1280)); w = tin("lr.w", (645, 1280), scale=0.03); b = tin("lr.b", (645,))
        out = eng.linear_rows(x.to(dev), w.to(dev), b.to(dev), silu_in=True)
        assert max_rel(out, F.linear(F.silu(x), w, b)) < 1e-5
        out = eng.linear_rows(x.to(dev), w.to(dev), None, silu_in=False)
        assert max_rel(out, F.linear(x, w)) < 1e-5


def test_layout_roundtrip(eng):
    x = tin("lay.x", (3, 100, 9, 5))
    rows = eng.nchw_to_rows(x.to(eng.device), cpad=104)
    assert rows.shape == (3 * 45, 104)
    assert float(rows[:, 100:].float().abs().max()) == 0.0
    back = eng.rows_to_nchw(rows, 3, 100, 9, 5)
    assert torch.equal(back.cpu(), rnd(x, eng.dtype))


# ------------------------------------------------------------------------------------------- 256x256 ping-pong tile
WIDE = _lib.TILE_D256x256_S2


@pytest.mark.parametrize("sk", [1, -5, -64, -300])
def test_wide_tile_multi_tile_gemm_and_conv(dev, sk):
    """Several 256x256 tiles in both directions, ragged edges in M and N, 1 to 45 k-tiles (the prologue, steady
    state and tail of its counted waits), residual / per-sample bias epilogue, work-list splits mid-tile."""
    eng = Engine.get(dev, torch.bfloat16)
    eng.reset()
    for m, n, k in ((700, 640, 64), (513, 320, 128), (300, 960, 640), (1030, 1280, 1920)):
        a = tin("w.a", (m, k)); w = tin("w.w", (n, k), scale=1 / math.sqrt(k)); b = tin("w.b", (n,))
        rb = tin("w.rb", (2, n)); r = tin("w.r", (m, n))
        lin = torch.nn.Linear(k, n)
        with torch.no_grad():
            lin.weight.copy_(w); lin.bias.copy_(b)
        pw = eng.pack_linear(lin)
        half = (m + 1) // 2
        eng.tile_override = (WIDE, sk)
        out = eng.gemm(up(eng, a), pw, residual=up(eng, r))
        eng.tile_override = None
        ref = rnd(a, eng.dtype) @ rnd(w, eng.dtype).t() + b + rnd(r, eng.dtype)
        assert max_rel(out.float(), ref) < BF16_TOL, (m, n, k)
    bsz, cin, cout, h, w_ = 3, 320, 640, 16, 16
    x = tin("wc.x", (bsz, cin, h, w_)); wt = tin("wc.w", (cout, cin, 3, 3), scale=1 / math.sqrt(9 * cin)); bias = tin("wc.b", (cout,))
    rb = tin("wc.rb", (bsz, cout))
    conv = torch.nn.Conv2d(cin, cout, 3, padding=1)
    with torch.no_grad():
        conv.weight.copy_(wt); conv.bias.copy_(bias)
    pw = eng.pack_conv3x3(conv)
    rows = eng.nchw_to_rows(x.to(dev))
    eng.tile_override = (WIDE, sk)
    out = eng.gemm(rows, pw, rowbias=rb.to(dev), rows_per_batch=h * w_,
                   conv=dict(batch=bsz, hin=h, win=w_, hout=h, wout=w_, cin_real=cin))
    eng.tile_override = None
    ref = F.conv2d(rnd(x, eng.dtype), rnd(wt, eng.dtype), bias, padding=1) + rb[:, :, None, None]
    assert max_rel(eng.rows_to_nchw(out, bsz, cout, h, w_), ref) < BF16_TOL
    assert int(eng.ws_sk[:4096].max()) == 0


def test_wide_tile_is_refused_where_unsupported(dev):
    lin = torch.nn.Linear(320, 320)
    for dt in (torch.float32, torch.bfloat16):
        eng = Engine.get(dev, dt)
        eng.reset()
        if dt == torch.float32:
            pw = eng.pack_linear(lin)
            eng.tile_override = (WIDE, 1)
            with pytest.raises(RuntimeError):
                eng.gemm(torch.zeros(64, 320, device=dev), pw)
        else:
            conv = torch.nn.Conv2d(64, 64, 3, stride=2, padding=1)
            pw = eng.pack_conv3x3(conv)
            eng.tile_override = (WIDE, 1)
            with pytest.raises(RuntimeError):
                eng.gemm(torch.zeros(2 * 64, 64, device=dev, dtype=dt), pw,
                         conv=dict(batch=2, hin=8, win=8, hout=4, wout=4, stride=2, cin_real=64))
        eng.tile_override = None


# ------------------------------------------------------------------------------------------- row-panel GEMM (bf16)
PANEL = tuple(_lib.PANEL_TILES)


def _bf16_engine(dev):
    e = Engine.get(dev, torch.bfloat16)
    e.reset()
    e.tile_override = None
    return e


@pytest.mark.parametrize("m,n,k", [(200, 320, 320), (1000, 960, 320), (260, 640, 640), (512, 1920, 640), (64, 64, 320)])
@pytest.mark.parametrize("tile", PANEL)
@pytest.mark.parametrize("ns", [1, 2, 5])
def test_gemm_panel_dense(dev, m, n, k, tile, ns):
    """Row-panel tiles (A rows stationary in registers, all column tiles per block, `ns` column slices): plain epilogue with
    bias and residual, ragged last row panel, strided operands."""
    eng = _bf16_engine(dev)
    if tile == _lib.TILE_P128 and k != 320:
        pytest.skip("P128 exists for K = 320")
    if ns > n // 64:
        pytest.skip("more column slices than column tiles")
    a = tin("pn.a", (m, k)); w = tin("pn.w", (n, k), scale=1 / math.sqrt(k)); b = tin("pn.b", (n,)); r = tin("pn.r", (m, n))
    lin = torch.nn.Linear(k, n)
    with torch.no_grad():
        lin.weight.copy_(w); lin.bias.copy_(b)
    pw = eng.pack_linear(lin)
    abuf = torch.zeros(m, k + 64, dtype=eng.dtype, device=dev); abuf[:, 32:32 + k] = up(eng, a)
    cbuf = torch.zeros(m, n + 128, dtype=eng.dtype, device=dev)
    rbuf = torch.zeros(m, n + 32, dtype=eng.dtype, device=dev); rbuf[:, 32:] = up(eng, r)
    eng.tile_override = (tile, ns)
    try:
        eng.gemm(abuf[:, 32:32 + k], pw, out=cbuf[:, 64:64 + n], residual=rbuf[:, 32:])
    finally:
        eng.tile_override = None
    ref = rnd(a, eng.dtype) @ rnd(w, eng.dtype).t() + b + rnd(r, eng.dtype)
    assert max_rel(cbuf[:, 64:64 + n].float(), ref) < BF16_TOL
    assert float(cbuf[:, :64].abs().max()) == 0.0 and float(cbuf[:, 64 + n:].abs().max()) == 0.0


@pytest.mark.parametrize("geglu", [False, True])
@pytest.mark.parametrize("tile", PANEL)
@pytest.mark.parametrize("ns", [1, 3])
@pytest.mark.parametrize("c", [320, 640])
def test_gemm_panel_folded_layernorm(dev, geglu, tile, ns, c):
    """The LayerNorm fold on the panel tiles: the producer's row sums come out of registers (one plain store per row
    when a block owns whole rows, atomics when the columns are sliced), the consumer applies mean / rstd."""
    eng = _bf16_engine(dev)
    if tile == _lib.TILE_P128 and c != 320:
        pytest.skip("P128 exists for K = 320")
    m, k0 = 300, c
    n = 8 * c if geglu else 3 * c
    a = tin("pl.a", (m, k0)); w0 = tin("pl.w0", (c, k0), scale=1 / math.sqrt(k0)); b0 = tin("pl.b0", (c,)) + 0.7
    r = tin("pl.r", (m, c)) * 2
    gamma = tin("pl.g", (c,)) + 1; beta = tin("pl.be", (c,))
    w1 = tin("pl.w1", (n, c), scale=1 / math.sqrt(c)); b1 = tin("pl.b1", (n,))
    l0 = torch.nn.Linear(k0, c); l1 = torch.nn.Linear(c, n); ln = torch.nn.LayerNorm(c, eps=1e-5)
    with torch.no_grad():
        l0.weight.copy_(w0); l0.bias.copy_(b0); l1.weight.copy_(w1); l1.bias.copy_(b1)
        ln.weight.copy_(gamma); ln.bias.copy_(beta)
    p0 = eng.pack_linear(l0); p1 = eng.pack_linear(l1, geglu=geglu, ln=ln)
    eng.tile_override = (tile, ns)
    try:
        t = eng.gemm(up(eng, a), p0, residual=up(eng, r), ln_out=True)
        sums = eng.ln_stats(t).clone()
        out = eng.gemm(t, p1, ln_in=eng.ln_stats(t))
    finally:
        eng.tile_override = None
    tt = rnd(rnd(a, eng.dtype) @ rnd(w0, eng.dtype).t() + b0 + rnd(r, eng.dtype), eng.dtype)
    assert max_rel(t.float(), tt) < BF16_TOL
    # the row sums are those of the values as stored
    ts = t.float().cpu().double()
    assert max_rel(sums[:, 0].cpu(), ts.sum(1)) < 1e-5 and max_rel(sums[:, 1].cpu(), (ts * ts).sum(1)) < 1e-5
    y = F.linear(F.layer_norm(tt, (c,), gamma, beta, 1e-5), w1, b1)
    if geglu:
        v, g = y.chunk(2, dim=-1)
        y = v * F.gelu(g)
    assert max_rel(out.float(), y) < 3e-2


@pytest.mark.parametrize("tile", PANEL)
@pytest.mark.parametrize("ns", [1, 2])
def test_gemm_panel_groupnorm_statistics(dev, tile, ns):
    """Per-channel sums of the stored output accumulated by the panel tiles feed the GroupNorm that follows."""
    eng = _bf16_engine(dev)
    batch, hw, c = 3, 256, 320
    m = batch * hw
    a = tin("ps.a", (m, c)); w = tin("ps.w", (c, c), scale=1 / math.sqrt(c)); b = tin("ps.b", (c,))
    gamma = tin("ps.g", (c,)) + 1; beta = tin("ps.be", (c,))
    conv = torch.nn.Conv2d(c, c, 1)
    with torch.no_grad():
        conv.weight.copy_(w.reshape(c, c, 1, 1)); conv.bias.copy_(b)
    pw = eng.pack_conv1x1(conv)
    eng.tile_override = (tile, ns)
    try:
        t = eng.gemm(up(eng, a), pw, gn_batch=batch)
        assert eng._gn_lookup(t, batch) is not None  # the statistics were accumulated by the launch
        y = eng.groupnorm(t, batch, hw, gamma.to(dev), beta.to(dev), 1e-5, silu=True)
    finally:
        eng.tile_override = None
    tt = t.float().cpu()
    ref = F.silu(F.group_norm(tt.reshape(batch, hw, c).permute(0, 2, 1), 32, gamma, beta, 1e-5)).permute(0, 2, 1).reshape(m, c)
    assert max_rel(y.float(), ref) < BF16_TOL


def test_gemm_panel_refused_where_unsupported(dev):
    """The panel tiles exist for bf16, dense, K = 320 / 640 and N a multiple of 64; everything else is DM_ERR_ARG."""
    for dt, k, n in ((torch.float32, 320, 320), (torch.bfloat16, 256, 320), (torch.bfloat16, 320, 96)):
        eng = Engine.get(dev, dt)
        eng.reset()
        lin = torch.nn.Linear(k, n)
        pw = eng.pack_linear(lin)
        eng.tile_override = (_lib.TILE_P128, 1)
        try:
            with pytest.raises(RuntimeError):
                eng.gemm(torch.zeros(128, k, dtype=dt, device=dev), pw)
        finally:
            eng.tile_override = None


@pytest.mark.parametrize("tile", PANEL)
@pytest.mark.parametrize("ns", [1, 2])
@pytest.mark.parametrize("c", [320, 640])
def test_gemm_panel_groupnorm_of_the_operand(dev, tile, ns, c):
    """bf16: the GroupNorm in front of an AttentionBlock's input projection (blocks.py:128-129, no activation) applied by the
    row-panel launch to the rows it keeps in registers, from the per-group sums the producing launch accumulated; the
    normalised tensor is never written.  Three samples, so blocks of different samples use different statistics."""
    eng = _bf16_engine(dev)
    if tile == _lib.TILE_P128 and c != 320:
        pytest.skip("P128 exists for K = 320")
    b, hw, k0, n = 3, 256, 64, c
    m = b * hw
    a0 = tin("pg.a0", (m, k0)) * (1 + torch.arange(m)[:, None] // hw); w0 = tin("pg.w0", (c, k0), scale=1 / math.sqrt(k0))
    b0 = tin("pg.b0", (c,)) + 0.3
    gamma = tin("pg.g", (c,)) + 1; beta = tin("pg.be", (c,)) * 0.5
    lin0 = torch.nn.Linear(k0, c); lin1 = torch.nn.Linear(c, n)
    w1 = tin("pg.w1", (n, c), scale=1 / math.sqrt(c)); b1 = tin("pg.b1", (n,))
    with torch.no_grad():
        lin0.weight.copy_(w0); lin0.bias.copy_(b0); lin1.weight.copy_(w1); lin1.bias.copy_(b1)
    x = eng.gemm(up(eng, a0), eng.pack_linear(lin0), gn_batch=b)  # producer: writes x and its per-group sums
    assert eng._gn_lookup(x, b) is not None
    gn = dict(batch=b, hw=hw, gamma=gamma.to(dev), beta=beta.to(dev), eps=1e-6, silu=False)

    def no_fallback(*_):
        raise AssertionError("the panel launch did not normalise its operand (fell back to the GroupNorm kernel)")

    eng._gn_unfused = no_fallback
    eng.tile_override = (tile, ns)
    try:
        out = eng.gemm(x, eng.pack_linear(lin1), gn_in=gn, ln_out=True)
    finally:
        eng.tile_override = None
        del eng._gn_unfused
    xs = x.float().cpu()
    xn = F.group_norm(xs.reshape(b, hw, c).permute(0, 2, 1), 32, gamma, beta, eps=1e-6).permute(0, 2, 1).reshape(m, c)
    ref = rnd(xn, eng.dtype) @ rnd(w1, eng.dtype).t() + b1
    assert max_rel(out.float(), ref) < BF16_TOL


# ------------------------------------------------------------------------------------------- fused feed-forward
@pytest.mark.parametrize("batch,hw", [(1, 128), (3, 256), (2, 1024)])
@pytest.mark.parametrize("with_x", [True, False])
def test_ffn_geglu_one_launch(dev, batch, hw, with_x):
    """The feed-forward half of an AttentionBlock (blocks.py:137-153: layernorm_3, linear_geglu_1, GELU gate, linear_geglu_2,
    residue, conv_output, long residue) as one launch against fp32 torch math on the CPU and against the two-launch form of
    the same packed operands; the per-group sums of the output feed a GroupNorm.  Several samples, so blocks of different
    samples add into different statistics rows; the output lands in a channel slice of a wider buffer."""
    eng = _bf16_engine(dev)
    c, m = 320, batch * hw
    k0 = 64
    a0 = tin("ff.a0", (m, k0)); w0 = tin("ff.w0", (c, k0), scale=1 / math.sqrt(k0)); b0 = tin("ff.b0", (c,)) + 0.4
    x = tin("ff.x", (m, c)) * (1 + torch.arange(m)[:, None] // hw)
    gamma = tin("ff.g", (c,)) + 1; beta = tin("ff.be", (c,)) * 0.5
    w1 = tin("ff.w1", (8 * c, c), scale=1 / math.sqrt(c)); b1 = tin("ff.b1", (8 * c,))
    w2 = tin("ff.w2", (c, 4 * c), scale=1 / math.sqrt(4 * c)); b2 = tin("ff.b2", (c,))
    wc = tin("ff.wc", (c, c), scale=1 / math.sqrt(c)); bc = tin("ff.bc", (c,))
    gn_g = tin("ff.gg", (c,)) + 1; gn_b = tin("ff.gb", (c,))
    l0 = torch.nn.Linear(k0, c); l1 = torch.nn.Linear(c, 8 * c); l2 = torch.nn.Linear(4 * c, c)
    ln = torch.nn.LayerNorm(c, eps=1e-5); co = torch.nn.Conv2d(c, c, 1)
    with torch.no_grad():
        l0.weight.copy_(w0); l0.bias.copy_(b0); l1.weight.copy_(w1); l1.bias.copy_(b1); l2.weight.copy_(w2); l2.bias.copy_(b2)
        ln.weight.copy_(gamma); ln.bias.copy_(beta); co.weight.copy_(wc.reshape(c, c, 1, 1)); co.bias.copy_(bc)
    p0 = eng.pack_linear(l0); p1 = eng.pack_linear(l1, geglu=True, ln=ln); p2 = eng.pack_chain(l2, co)
    assert eng.lib.dm_ffn_geglu_supported(eng.dm, m, c, hw, 1)
    t = eng.gemm(up(eng, a0), p0, ln_out=True)  # the residual stream and its row sums
    xd = up(eng, x) if with_x else None
    wide = eng.new(m, c + 64)
    wide.zero_()
    out = wide[:, 32:32 + c]
    eng.ffn_geglu(t, p1, p2, out, residual=xd, ln_in=eng.ln_stats(t), gn_batch=batch)
    assert eng._gn_lookup(out, batch) is not None, "the launch did not accumulate the statistics of its output"
    y = eng.groupnorm(out, batch, hw, gn_g.to(dev), gn_b.to(dev), 1e-5, silu=True)
    assert float(wide[:, :32].abs().max()) == 0.0 and float(wide[:, 32 + c:].abs().max()) == 0.0
    # two-launch form on the same operands
    g2 = eng.gemm(t, p1, ln_in=eng.ln_stats(t))
    ref2 = eng.gemm(g2, p2, a2=t, residual=xd)
    assert max_rel(out.float(), ref2.float()) < 1e-2
    # fp32 math of the reference's operator chain on the stored residual stream
    ts = t.float().cpu()
    v, g = F.linear(F.layer_norm(ts, (c,), gamma, beta, 1e-5), w1, b1).chunk(2, dim=-1)
    ref = F.linear(F.linear(v * F.gelu(g), w2, b2) + ts, wc, bc)
    if with_x:
        ref = ref + rnd(x, eng.dtype)
    assert max_rel(out.float(), ref) < 3e-2
    os_ = out.float().cpu()
    gref = F.silu(F.group_norm(os_.reshape(batch, hw, c).permute(0, 2, 1), 32, gn_g, gn_b, 1e-5)).permute(0, 2, 1).reshape(m, c)
    assert max_rel(y.float(), gref) < BF16_TOL


def test_ffn_geglu_refused_where_unsupported(dev):
    """One launch exists for bf16, C = 320 and whole blocks of 128 rows inside one sample; everything else reports so and the
    block runner keeps the two-launch form."""
    eng = _bf16_engine(dev)
    big = 16 * 4096
    assert eng.ffn_ok(big, 320, 16)
    assert not eng.ffn_ok(big, 640, 16) and not eng.ffn_ok(big + 64, 320, 1) and not eng.ffn_ok(342 * 192, 320, 342)
    assert not eng.ffn_ok(4096, 320, 1)  # too few rows to pay: the block runner keeps two launches
    assert not Engine.get(dev, torch.float32).ffn_ok(big, 320, 16)
    d = _lib.FfnDesc()
    assert _lib.lib().dm_ffn_geglu(d, None) != 0
