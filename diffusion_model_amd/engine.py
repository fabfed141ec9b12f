"""Host-side driver of the HIP kernels: memory arena, weight packing and per-operator launch wrappers.

Everything here is plumbing around ``libdm_hip.so``: torch tensors are used only as device
memory (``data_ptr``) and for one-time weight re-layout; all arithmetic of the hot path runs in
the hand-written kernels.  Activations are 2-D row-major views ``[rows, C]`` (row = (batch, y, x)
or (batch, token)) with an arbitrary row stride, which lets producers write straight into the
channel slices of the U-Net's skip-concatenation buffers.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib
from ._lib import GemmDesc, check

_DM = {torch.float32: _lib.DM_F32, torch.bfloat16: _lib.DM_BF16}


def _rup(x: int, m: int) -> int:
    return (x + m - 1) // m * m


class Arena:
    """Bump allocator over one device buffer; deterministic addresses so launches can be graph-captured."""

    def __init__(self, device, nbytes: int = 256 << 20, on_grow=None):
        self.device = device
        self.buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        self.off = 0
        self.peak = 0
        self.frozen = False  # set while a hipGraph capture is in progress (no regrowth allowed)
        self._old = []       # previous buffers kept alive while views into them may still be in use
        self._on_grow = on_grow  # called after the buffer was replaced (addresses handed out so far are stale)

    def reset(self):
        self.off = 0
        if self._old:
            # kernels on the current stream may still read the retired buffers: tell the caching allocator
            if self.buf.is_cuda:
                cur = torch.cuda.current_stream(self.buf.device)
                for b in self._old:
                    b.record_stream(cur)
            self._old.clear()

    def mark(self) -> int:
        return self.off

    def release(self, mark: int):
        self.off = mark

    def alloc(self, rows: int, cols: int, dtype: torch.dtype) -> torch.Tensor:
        esz = 4 if dtype in (torch.float32, torch.int32) else 2
        nbytes = _rup(rows * cols * esz, 256)
        if self.off + nbytes > self.buf.numel():
            if self.frozen:
                raise RuntimeError("arena exhausted during graph capture; run one eager step first")
            self._old.append(self.buf)
            self.buf = torch.empty(max(2 * self.buf.numel(), self.off + nbytes), dtype=torch.uint8, device=self.device)
            # earlier views stay valid in the old buffer; continue in the new one at the same offset
            if self._on_grow is not None:
                self._on_grow()
        t = self.buf[self.off:self.off + rows * cols * esz].view(dtype).view(rows, cols)
        self.off += nbytes
        self.peak = max(self.peak, self.off)
        return t


class PackedWeight:
    """Weight matrix in the GEMM kernel's layout plus its fp32 bias."""
    __slots__ = ("w", "n", "k", "npad", "kpad", "bias", "geglu", "conv_cin", "colsum", "ln_eps", "skip_cin", "split",
                 "chunk_major", "w_halo")

    def __init__(self, w, n, k, npad, kpad, bias, geglu=False, conv_cin=0):
        self.w, self.n, self.k, self.npad, self.kpad = w, n, k, npad, kpad
        self.bias, self.geglu, self.conv_cin = bias, geglu, conv_cin
        self.colsum = None  # set when a LayerNorm is folded in: sum over k of every packed row (fp32)
        self.ln_eps = 0.0
        self.skip_cin = 0  # conv3x3 packed with a pointwise term over a second tensor: its channel count
        self.split = 0  # fp32 weights stored as 2 or 3 bf16 terms per value (Engine.split_bf16); 0 = plain image
        self.chunk_major = False  # conv3x3: k axis ordered (channel chunk, tap) instead of (tap, channel)
        self.w_halo = None  # conv3x3, chunk-major: second image in the unit order of the halo tile (_lib.TILE_H128x128)


class Engine:
    """One per (device, compute dtype).  ``dtype`` float32 = exact-fp32 MFMA mode, bfloat16 = bf16 MFMA mode."""

    _instances: Dict[Tuple[str, torch.dtype], "Engine"] = {}

    @classmethod
    def get(cls, device, dtype: torch.dtype) -> "Engine":
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError(
                "diffusion_model_amd runs on the HIP device only (tensor on %s); there is no CPU path" % device)
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        if dtype not in _DM:
            raise RuntimeError(f"unsupported compute dtype {dtype}; use float32 or bfloat16")
        if torch.cuda.current_device() != device.index:
            # kernel launches and per-device kernel attributes go to the CURRENT device: make it the tensors' one
            # (it stays current afterwards, as after torch.cuda.set_device)
            torch.cuda.set_device(device)
        key = (str(device), dtype)
        if key not in cls._instances:
            cls._instances[key] = Engine(device, dtype)
        return cls._instances[key]

    def __init__(self, device, dtype):
        self.lib = _lib.lib()
        self.device, self.dtype = device, dtype
        self.dm = _DM[dtype]
        self.esz = 4 if dtype == torch.float32 else 2
        self.epc = 16 // self.esz
        self.bk = 8 * self.epc
        # Bumped whenever one of the engine-owned buffers (arena, split-K slabs, work-list workspace, statistics)
        # is replaced by a larger one.  A captured hipGraph bakes those addresses in: whoever holds such a graph
        # records the generation at capture and re-captures when it has moved.
        self.generation = 0
        self.arena = Arena(device, on_grow=self._bump_generation)
        self.ws = torch.empty(64 << 20, dtype=torch.uint8, device=device)  # split-K slabs
        # work-list (stream-K) GEMMs: arrival counters + accumulator slabs; the counters must read zero
        # at every launch (the kernel leaves them zero), so this buffer is never shared with the slabs above
        self.ws_sk = torch.zeros(160 << 20, dtype=torch.uint8, device=device)
        # GroupNorm statistics accumulated by the producing GEMM: per-group fp64 (sum, sumsq), [batch][2][32], and the
        # per-row (sum, sumsq) of tensors a folded LayerNorm reads.  Regions are handed out bump-style per forward and
        # the used prefix is zeroed by reset().
        self.stats_buf = torch.zeros(8 << 20, dtype=torch.float64, device=device)
        self._stats_off = 0    # elements handed out since the last reset
        self._stats_dirty = 0  # high-water mark of elements that may be non-zero
        self._gn_shared: List[dict] = []      # buffers with several producers (skip-concat buffers)
        self._gn_auto: Dict[int, dict] = {}   # data_ptr -> record of a tensor written by one GEMM
        self._ln_stats: Dict[int, torch.Tensor] = {}  # data_ptr -> [rows][2] fp64 row sums (folded LayerNorm)
        self.fuse_ln = os.environ.get("DM_FUSE_LN", "1") == "1"
        self.tune_early = os.environ.get("DM_TUNE_EARLY", "1") != "0"
        self.fuse_skip = os.environ.get("DM_FUSE_SKIP", "1") != "0"  # residual_layer folded into the second conv
        self.fuse_tail = os.environ.get("DM_FUSE_TAIL", "1") != "0"  # linear_geglu_2 + residue + conv_output as one GEMM
        self.fuse_gn_stats = os.environ.get("DM_FUSE_GN_STATS", "1") == "1"
        # bf16, C = 320: linear_geglu_1 -> GELU gate -> [linear_geglu_2 + residue -> conv_output] as one launch
        self.fuse_ffn = os.environ.get("DM_FUSE_FFN", "1") != "0"
        self.ffn_min_rows = int(os.environ.get("DM_FFN_MIN_ROWS", str(256 * 128)))
        # bf16, stride-1 same-size 3x3 convolutions: k axis of the packed weights chunk-major
        # (pack_conv3x3(same_size=True)).  It cuts the convolutions' L2 misses (the nine taps' re-reads of the activation
        # then hit) and is worth 1 % of their time.  The exact-fp32 loop moves an eighth of the operand bytes per MFMA
        # cycle and would only pay the per-request address arithmetic (measured 5 % slower at batch 1): bf16 only.
        self.conv_chunk_major = os.environ.get("DM_CONV_CHUNK_MAJOR", "1") != "0" and dtype == torch.bfloat16
        # ... and a second image of those weights for the tile that keeps the activation of a chunk resident in LDS
        self.conv_halo = os.environ.get("DM_CONV_HALO", "1") != "0" and self.conv_chunk_major
        # fp32: GroupNorm (+ SiLU) applied to the A fragments of the convolution / projection that reads it
        self.fuse_gn_apply = os.environ.get("DM_FUSE_GN_APPLY", "1") != "0"
        # ... for convolutions only up to this many output channels.  The map is redone by every block that reads the
        # element (9 taps x N/64 column tiles); with SiLU it holds two transcendental instructions, and the 320-channel
        # 64 x 64 convolution measured 93 us as GroupNorm kernel + convolution against 114 us with the map inside.  The
        # projections behind an AttentionBlock's GroupNorm (no SiLU, no taps) gain 1.1-1.2x and are always taken.
        self.fuse_gn_conv_max_n = int(os.environ.get("DM_FUSE_GN_CONV_MAX_N", "64"))
        # Opt-in arithmetic modes of the fp32 engine (DM_FP32_SPLIT=2, also written 1; off by default: the default is the reference's
        # arithmetic, exact-fp32 MFMA).  Weights are packed as two bf16 terms per value and the GEMM / conv kernels
        # form a_hi w_hi + a_lo w_hi + a_hi w_lo on the bf16 MFMA with the fp32 activations split in registers: about
        # 16 bits of each operand instead of 24, fp32 accumulation, 16x the matrix rate per product.  Tensors, layouts,
        # attention, norms and the scheduler stay fp32.
        # DM_FP32_SPLIT=3: three terms per operand and six products - 24 bits of each operand, what is dropped is the
        # size of an fp32 rounding; GEMM / conv only (attention stays on the exact-fp32 MFMA in that form).
        mode = os.environ.get("DM_FP32_SPLIT", "0") if dtype == torch.float32 else "0"
        self.split_bf16 = {"1": 2, "2": 2, "3": 3}.get(mode, 0)  # 0 = off; True counts as 2
        self.profile: Optional[dict] = None  # kernel class -> [events...] when profiling
        self._order_cache: Dict[Tuple[int, int, bool], torch.Tensor] = {}
        self.tile_override: Optional[Tuple[int, int]] = None
        self.autotune = os.environ.get("DM_AUTOTUNE", "0") == "1"  # the sampling loop switches it on
        self.tuned: Dict[tuple, Tuple[int, int]] = {}
        # Tile table.  The package ships one (tune_table.json next to this file: winners measured on MI355X for the
        # benchmark configurations), so that tile and work-list choices - which fix the fp32 summation order - are the
        # same in every process and no launch of a covered shape is ever timed on the user's stream.  Shapes missing
        # from it are timed once by a sampling loop's first eager step (StepLoop) and otherwise planned analytically.
        # DM_TUNE_TABLE=<file> replaces the shipped table ("0" disables it); DM_TUNE_CACHE=<file> is read on top and
        # extended with every newly timed shape.
        self._tune_cache = os.environ.get("DM_TUNE_CACHE")
        table = os.environ.get("DM_TUNE_TABLE", os.path.join(os.path.dirname(os.path.abspath(__file__)), "tune_table.json"))
        for path in (table, self._tune_cache):
            if path and path != "0" and os.path.exists(path):
                self._load_tuned(path)

    def _load_tuned(self, path: str):
        import ast
        import json
        with open(path) as f:
            for k, v in json.load(f).get(str(self.dtype), {}).items():
                self.tuned[ast.literal_eval(k)] = tuple(v)

    # ------------------------------------------------------------------ utilities
    def _bump_generation(self):
        self.generation += 1

    def _retire(self, buf: torch.Tensor):
        """A workspace is being replaced: launches already queued on the current stream may still use it."""
        buf.record_stream(torch.cuda.current_stream(self.device))
        self.generation += 1

    @property
    def stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    def _prof(self, klass: str, work: float, tag: str = ""):
        """Context manager timing one launch group with HIP events when profiling is on."""
        return _Timed(self, klass, work, tag) if self.profile is not None else _NULLCTX

    def new(self, rows, cols, dtype=None, persistent=False) -> torch.Tensor:
        dtype = dtype or self.dtype
        if persistent:
            return torch.empty(rows, cols, dtype=dtype, device=self.device)
        t = self.arena.alloc(rows, cols, dtype)
        self._gn_auto.pop(t.data_ptr(), None)  # whatever lived at this address before is gone
        return t

    def reset(self):
        """Start of a forward: recycle the activation arena and clear the statistics region used so far."""
        self.arena.reset()
        self._gn_shared.clear()
        self._gn_auto.clear()
        self._ln_stats.clear()
        if self._stats_dirty:
            check(self.lib.dm_zero(self.stats_buf.data_ptr(), self._stats_dirty * 8, self.stream), "zero")
        self._stats_off = 0

    # ------------------------------------------------------------------ fused GroupNorm statistics
    def _stats_alloc(self, batch: int, c: int) -> torch.Tensor:
        n = _rup(batch * 2 * c, 32)
        if self._stats_off + n > self.stats_buf.numel():
            if self.arena.frozen:
                raise RuntimeError("statistics buffer exhausted during graph capture; run one eager step first")
            # Views handed out earlier in this forward keep living in the old buffer; allocation continues at the SAME
            # offset in the new one (all zeros), so the high-water mark that reset() clears - and that a captured
            # memset node bakes in - still covers everything a whole forward of this shape touches.
            self._retire(self.stats_buf)
            self.stats_buf = torch.zeros(max(2 * self.stats_buf.numel(), self._stats_off + n), dtype=torch.float64,
                                         device=self.device)
        t = self.stats_buf[self._stats_off:self._stats_off + batch * 2 * c].view(batch, 2, c)
        self._stats_off += n
        self._stats_dirty = max(self._stats_dirty, self._stats_off)
        return t

    GN_GROUPS = 32  # every GroupNorm of the U-Net and of the autoencoder has 32 groups; others take the two-launch form

    def _gn_record(self, t: torch.Tensor, batch: int) -> dict:
        """Bookkeeping of one tensor whose per-group sums are accumulated by the GEMM(s) that write it: the kernels reduce
        their column sums to the groups of the GroupNorm that reads the tensor, [batch][2][groups] fp64."""
        cols = t.shape[1]
        return dict(base=t.data_ptr(), ld=t.stride(0), cols=cols, rows=t.shape[0], batch=batch, groups=self.GN_GROUPS,
                    cpg=cols // self.GN_GROUPS, stats=self._stats_alloc(batch, self.GN_GROUPS),
                    valid=cols % self.GN_GROUPS == 0, parts=0)

    def gn_register(self, buf: torch.Tensor, batch: int):
        """Announce a buffer whose channel slices are written by several GEMMs and then group-normalised whole."""
        if not self.fuse_gn_stats:
            return
        self._gn_shared.append(self._gn_record(buf, batch))

    def _gn_targets(self, out: torch.Tensor, batch: int):
        """[(record, channel offset)] that a GEMM writing `out` accumulates its statistics into: the tensor itself, and
        the shared buffer it is a channel slice of (if any) - the two GroupNorms that may read it partition its
        channels into different groups."""
        p, ld = out.data_ptr(), out.stride(0)
        rec = self._gn_record(out, batch)
        self._gn_auto[p] = rec
        targets = [(rec, 0)]
        for sh in self._gn_shared:
            if sh["ld"] == ld and sh["base"] <= p < sh["base"] + ld * self.esz and sh["rows"] == out.shape[0]:
                targets.append((sh, (p - sh["base"]) // self.esz))
                break
        return targets

    def _gn_lookup(self, x: torch.Tensor, batch: int, groups: int = 32):
        """Statistics of `x` (the input of a GroupNorm with `groups` groups) if the launches that wrote it accumulated
        all of them: the tensor's own record, or the shared buffer's when `x` is such a buffer as a whole."""
        p = x.data_ptr()
        cands = [self._gn_auto.get(p)] + [r for r in self._gn_shared if r["base"] == p]
        for rec in cands:
            if (rec is not None and rec["valid"] and rec["groups"] == groups and rec["ld"] == x.stride(0)
                    and rec["cols"] == x.shape[1] and rec["rows"] == x.shape[0] and rec["batch"] == batch
                    and rec["parts"] == rec["cols"]):
                return dict(stats_ptr=rec["stats"].data_ptr(), ld=rec["groups"], rec=rec)
        return None

    # ------------------------------------------------------------------ weight packing
    def _order(self, n: int, npad: int, geglu: bool) -> torch.Tensor:
        key = (n, npad, geglu)
        if key not in self._order_cache:
            arr = (C.c_int * npad)()
            check(self.lib.dm_pack_weight_order(n, npad, int(geglu), arr), "pack_weight_order")
            self._order_cache[key] = torch.tensor(list(arr), dtype=torch.long, device=self.device)
        return self._order_cache[key]

    def pack_matrix(self, w2d: torch.Tensor, bias: Optional[torch.Tensor], geglu=False, conv_cin=0,
                    rowsum=False) -> PackedWeight:
        n, k = w2d.shape
        npad, kpad = _rup(n, 256), _rup(k, self.bk)  # widest block tile is 256 columns
        order = self._order(n, npad, geglu)
        wp = torch.zeros(npad, kpad, dtype=self.dtype, device=self.device)
        valid = order >= 0
        wp[valid, :k] = w2d.detach().to(self.device)[order[valid]].to(self.dtype)
        b = None if bias is None else bias.detach().to(device=self.device, dtype=torch.float32).contiguous()
        pw = PackedWeight(None, n, k, npad, kpad, b, geglu, conv_cin)
        if self.split_bf16:
            nkt = kpad // self.bk
            # k order inside a k-tile: lane group q of the kernel holds k = 4q .. 4q+3 and 16+4q .. 16+4q+3 (its two
            # conflict-free chunks of the fp32 activation row), so those eight terms sit together in the weight row
            perm = torch.tensor([4 * q + e if e < 4 else 16 + 4 * q + e - 4 for q in range(4) for e in range(8)],
                                device=wp.device)
            wp = wp.view(npad, nkt, self.bk)[:, :, perm].reshape(npad, kpad)
            hi = wp.to(torch.bfloat16)
            r1 = wp - hi.float()
            if self.split_bf16 == 3:
                # three terms: per k-tile two 128-byte rows, [hi | mid] and [lo | zeros], planes side by side per 64-row group
                mid = r1.to(torch.bfloat16)
                lo = (r1 - mid.float()).to(torch.bfloat16)
                seen = hi.float() + mid.float() + lo.float()
                p0 = torch.cat([hi.view(npad, nkt, self.bk), mid.view(npad, nkt, self.bk)], dim=2)
                p1 = torch.cat([lo.view(npad, nkt, self.bk), torch.zeros_like(lo).view(npad, nkt, self.bk)], dim=2)
                img = torch.stack([p0, p1], dim=2)  # [npad][nkt][2][64 bf16]
                pw.w = img.view(npad // 64, 64, nkt, 2, 2 * self.bk).permute(0, 2, 3, 1, 4).contiguous()
                pw.split = 3
            else:
                # two terms: per k-tile of 32 values 32 hi terms, then 32 lo terms - 128 bytes, as in the fp32 image
                lo = r1.to(torch.bfloat16)
                seen = hi.float() + lo.float()
                img = torch.cat([hi.view(npad, nkt, self.bk), lo.view(npad, nkt, self.bk)], dim=2)
                pw.w = img.view(npad // 64, 64, nkt, 2 * self.bk).permute(0, 2, 1, 3).contiguous()
                pw.split = 2
            if rowsum:
                pw.colsum = seen.sum(dim=1).contiguous()
            return pw
        if rowsum:  # per packed row, of the values the MFMA really sees (after rounding to the engine dtype)
            pw.colsum = wp.float().sum(dim=1).contiguous()
        # blocked image [npad/64][kpad/bk][64][bk]: the k-tile of a 64-row group is one contiguous 8 KiB piece
        pw.w = wp.view(npad // 64, 64, kpad // self.bk, self.bk).permute(0, 2, 1, 3).contiguous()
        return pw

    def pack_linear(self, lin: torch.nn.Linear, geglu=False, ln: Optional[torch.nn.LayerNorm] = None) -> PackedWeight:
        """``ln``: the LayerNorm whose output is this Linear's only input.  Its scale goes into the weights
        (W' = W diag(gamma)), its shift into the bias (beta W^T + b); the per-row mean / rstd are applied by the
        GEMM epilogue from row sums that the producer of the LayerNorm's input accumulated."""
        if ln is None:
            return self.pack_matrix(lin.weight, lin.bias, geglu)
        w = lin.weight.detach().to(self.device, torch.float32)
        gamma = ln.weight.detach().to(self.device, torch.float32)
        beta = ln.bias.detach().to(self.device, torch.float32)
        bias = w @ beta
        if lin.bias is not None:
            bias = bias + lin.bias.detach().to(self.device, torch.float32)
        pw = self.pack_matrix(w * gamma[None, :], bias, geglu, rowsum=True)
        pw.ln_eps = float(ln.eps)
        return pw

    def pack_chain(self, first: torch.nn.Linear, second: torch.nn.Conv2d) -> PackedWeight:
        """Weights of ``second(first(a) + r)`` (a Linear, a residual add, a 1x1 convolution: all linear) as ONE
        matrix over the concatenated row [a | r]: [W2 W1 | W2], bias W2 b1 + b2.  The product is formed in fp64."""
        w1 = first.weight.detach().to(self.device, torch.float64)
        w2 = second.weight.detach().to(self.device, torch.float64).reshape(second.out_channels, second.in_channels)
        if w2.shape[1] != w1.shape[0] or w1.shape[1] % self.bk or w2.shape[1] % self.bk:
            raise RuntimeError("chain needs matching widths that are whole k-tiles")
        bias = torch.zeros(w2.shape[0], dtype=torch.float64, device=self.device)
        if first.bias is not None:
            bias = bias + w2 @ first.bias.detach().to(self.device, torch.float64)
        if second.bias is not None:
            bias = bias + second.bias.detach().to(self.device, torch.float64)
        pw = self.pack_matrix(torch.cat([w2 @ w1, w2], dim=1).float(), bias.float())
        pw.skip_cin = w2.shape[1]
        return pw

    def pack_conv1x1(self, conv: torch.nn.Conv2d) -> PackedWeight:
        return self.pack_matrix(conv.weight.reshape(conv.out_channels, conv.in_channels), conv.bias)

    def pack_conv3x3(self, conv: torch.nn.Conv2d, skip: Optional[torch.nn.Conv2d] = None,
                     same_size: bool = False) -> PackedWeight:
        """``skip``: a 1x1 convolution over a second tensor whose result is added to this convolution's output
        (ResidualBlock.residual_layer); its weights are appended to the k axis and its bias folded in, so the
        sum is formed inside one launch (``gemm(..., conv=dict(..., skip=x))``).
        ``same_size``: the caller only ever runs these weights as a stride-1, padding-1 convolution whose output has the
        input's size (no fused upsampling).  The k axis is then laid out chunk-major (the nine taps of one k-tile of
        channels follow each other) when the channel count allows it, which keeps the taps' re-reads of the activation
        in L2; ``gemm`` refuses such weights for any other geometry."""
        w = conv.weight.detach().permute(0, 2, 3, 1)  # (Cout, 3, 3, Cin): k = tap*Cin + c
        cin = conv.in_channels
        cpad = _rup(cin, 8)
        if cpad != cin:
            w = torch.nn.functional.pad(w, (0, cpad - cin))
        chunk_major = bool(same_size and self.conv_chunk_major and cpad % self.bk == 0 and not self.split_bf16)
        if chunk_major:
            # (Cout, tap, chunk, bk) -> (Cout, chunk, tap, bk): k-tile index = chunk * 9 + tap
            w = w.reshape(conv.out_channels, 9, cpad // self.bk, self.bk).permute(0, 2, 1, 3)
        w2d = w.reshape(conv.out_channels, 9 * cpad)
        # the halo tile reads the same weights in its own k order: per chunk (half of 32 channels, tap, 32 values)
        w2d_halo = None
        if chunk_major and self.conv_halo:
            wt = conv.weight.detach().permute(0, 2, 3, 1)
            if cpad != cin:
                wt = torch.nn.functional.pad(wt, (0, cpad - cin))
            w2d_halo = wt.reshape(conv.out_channels, 9, cpad // self.bk, 2, self.bk // 2).permute(0, 2, 3, 1, 4) \
                         .reshape(conv.out_channels, 9 * cpad)
        if skip is None:
            pw = self.pack_matrix(w2d, conv.bias, conv_cin=cpad)
            pw.chunk_major = chunk_major
            if w2d_halo is not None:
                pw.w_halo = self.pack_matrix(w2d_halo, None, conv_cin=cpad).w
            return pw
        cin2 = skip.in_channels
        if cpad % self.bk or cin2 % self.bk or skip.out_channels != conv.out_channels:
            raise RuntimeError("pointwise term needs channel counts that are multiples of the k-tile")
        w2d = torch.cat([w2d, skip.weight.detach().reshape(skip.out_channels, cin2)], dim=1)
        bias = conv.bias.detach().float()
        if skip.bias is not None:
            bias = bias + skip.bias.detach().float()
        pw = self.pack_matrix(w2d, bias, conv_cin=cpad)
        pw.skip_cin = cin2
        pw.chunk_major = chunk_major
        if w2d_halo is not None:
            w2d_halo = torch.cat([w2d_halo, skip.weight.detach().reshape(skip.out_channels, cin2)], dim=1)
            pw.w_halo = self.pack_matrix(w2d_halo, None, conv_cin=cpad).w
        return pw

    # ------------------------------------------------------------------ operators
    def gemm(self, a: torch.Tensor, pw: PackedWeight, out: Optional[torch.Tensor] = None, *, bias=True,
             rowbias: Optional[torch.Tensor] = None, rows_per_batch: int = 0, residual: Optional[torch.Tensor] = None,
             conv: Optional[dict] = None, nchw_out: Optional[torch.Tensor] = None, persistent=False,
             gn_batch: int = 0, act: Optional[str] = None, ln_in: Optional[torch.Tensor] = None,
             ln_out: bool = False, a2: Optional[torch.Tensor] = None, gn_in: Optional[dict] = None) -> torch.Tensor:
        """``gn_batch`` > 0: a GroupNorm over ``gn_batch`` samples reads the output next; its statistics are
        accumulated by this launch where the chosen tile allows it.
        ``gn_in`` = dict(batch, hw, gamma, beta, eps, silu[, groups]): ``a`` is the INPUT of a GroupNorm (+ SiLU) whose
        output is this launch's operand.  In fp32 mode, when the producer of ``a`` accumulated its statistics and the
        tile allows it, the normalisation is applied to the A fragments inside the launch and the normalised tensor
        is never written; otherwise the GroupNorm kernel runs first.
        ``ln_out``: a LayerNorm folded into a later GEMM reads the output next; per-row sums are accumulated
        here (fetch them with ``ln_stats(out)``).  ``ln_in``: such sums for the rows of ``a``; ``pw`` must have
        been packed with that LayerNorm (``pack_linear(..., ln=...)``)."""
        d = GemmDesc()
        gn_rec = None
        if gn_in is not None:
            c_gn = a.shape[1]
            geom_ok = conv is None or (conv.get("stride", 1) == 1 and not conv.get("up", 0)
                                       and not conv.get("pad_hi_only", 0) and conv["hin"] == conv["hout"]
                                       and conv["win"] == conv["wout"] and conv["hout"] * conv["wout"] == gn_in["hw"])
            f32_form = (self.dtype == torch.float32 and not pw.split and geom_ok and ln_in is None and a2 is None
                        and (conv is None or pw.n <= self.fuse_gn_conv_max_n)
                        and (conv is not None) == bool(gn_in["silu"]) and c_gn % 32 == 0 and c_gn <= 1280)
            # bf16: the row-panel tiles normalise the rows they keep in registers (affine only, i.e. the projection
            # behind an AttentionBlock's GroupNorm); whether the launch gets such a tile is known once it is chosen
            panel_form = (self.dtype == torch.bfloat16 and conv is None and not gn_in["silu"] and ln_in is None
                          and a2 is None and c_gn in (320, 640))
            if self.fuse_gn_apply and (f32_form or panel_form):
                gn_rec = self._gn_lookup(a, gn_in["batch"], gn_in.get("groups", 32))
            if gn_rec is None:
                a = self._gn_unfused(a, gn_in)
        if conv is not None:
            m = conv["batch"] * conv["hout"] * conv["wout"]
            d.conv, d.hin, d.win, d.hout, d.wout = 1, conv["hin"], conv["win"], conv["hout"], conv["wout"]
            d.cin, d.stride, d.up = pw.conv_cin, conv.get("stride", 1), conv.get("up", 0)
            d.pad_hi_only = int(conv.get("pad_hi_only", 0))
            if a.shape[1] != pw.conv_cin:
                raise RuntimeError("conv input channel mismatch")
            if pw.chunk_major:
                if (d.stride != 1 or d.up or d.pad_hi_only or d.hin != d.hout or d.win != d.wout):
                    raise RuntimeError("weights packed with same_size=True run stride-1, same-size convolutions only")
                d.conv_chunk_major = 1
            skip = conv.get("skip")
            if (skip is None) != (pw.skip_cin == 0):
                raise RuntimeError("weights packed with a pointwise term need its input (conv['skip']) and vice versa")
            if skip is not None:
                if skip.shape[0] != m or skip.shape[1] != pw.skip_cin:
                    raise RuntimeError("pointwise input shape mismatch")
                d.a2, d.lda2, d.cin2 = skip.data_ptr(), skip.stride(0), pw.skip_cin
            klass = "conv3x3"
        else:
            m = a.shape[0]
            if a2 is not None:
                # the last pw.skip_cin columns of the logical A row live in a second tensor
                if pw.skip_cin == 0 or a2.shape != (m, pw.skip_cin) or a.shape[1] + pw.skip_cin != pw.k:
                    raise RuntimeError("second GEMM operand does not match the packed weights")
                d.a2, d.lda2, d.cin2 = a2.data_ptr(), a2.stride(0), pw.skip_cin
            elif a.shape[1] != pw.k:
                raise RuntimeError(f"GEMM inner dimension mismatch: {a.shape[1]} vs {pw.k}")
            klass = "gemm"
        n_out = pw.n // 2 if pw.geglu else pw.n
        if nchw_out is not None:
            d.c, d.ldc, d.epi = nchw_out.data_ptr(), 0, _lib.EPI_NCHW_F32
            out = nchw_out
        else:
            if out is None:
                out = self.new(m, n_out, persistent=persistent)
            d.c, d.ldc = out.data_ptr(), out.stride(0)
            d.epi = _lib.EPI_GEGLU if pw.geglu else _lib.EPI_PLAIN
            if act is not None:
                if act != "quickgelu" or pw.geglu or residual is not None:
                    raise RuntimeError(f"unsupported fused activation {act!r}")
                d.epi = _lib.EPI_QUICKGELU
        d.a, d.lda = a.data_ptr(), a.stride(0)
        d.w = pw.w.data_ptr()
        d.bias = pw.bias.data_ptr() if (bias and pw.bias is not None) else None
        if rowbias is not None:
            d.rowbias, d.ld_rowbias = rowbias.data_ptr(), rowbias.stride(0)
        d.rows_per_batch = rows_per_batch
        if residual is not None:
            d.r, d.ldr = residual.data_ptr(), residual.stride(0)
        d.m, d.n, d.k, d.npad, d.kpad, d.dtype = m, pw.n, pw.k, pw.npad, pw.kpad, self.dm
        d.w_split = int(pw.split)
        tile, sk = C.c_int(0), C.c_int(1)
        if self.tile_override is not None:
            tile.value, sk.value = self.tile_override
            if d.epi != _lib.EPI_PLAIN and sk.value > 1 and tile.value not in _lib.PANEL_TILES:
                sk.value = 1
        else:
            no_legacy = ln_in is not None or ln_out
            # operands that change what a launch does beside the product (fused statistics, a second k-axis
            # operand) are part of the key: a tile chosen for one use is never silently applied to another
            key = (m, pw.n, pw.k, d.conv, d.stride, d.up, d.epi, residual is not None, rowbias is not None, no_legacy,
                   gn_batch > 0 and self.fuse_gn_stats, bool(d.a2), d.pad_hi_only)
            if pw.split:
                key = key + ("w_split" if pw.split == 2 else "w_split3",)  # other kernels, other winners
            if pw.chunk_major:
                key = key + ("chunk_major",)  # another memory access pattern, other winners
            if gn_rec is not None:
                if self.dtype == torch.float32:
                    # the fragment-side GroupNorm changes the LDS footprint and the loop: its own entry
                    key = key + ("gn_in",)
                self._bind_gn(d, gn_rec, gn_in)
            best = self.tuned.get(key)
            if best is None and self.autotune and not self.arena.frozen and self.profile is None:
                self._tune_pw = pw
                best = self._tune(d, key, legacy=not no_legacy)
            if best is not None:
                tile.value, sk.value = best
            else:
                self.lib.dm_gemm_plan(self.dm, m, pw.n, pw.k, d.epi, C.byref(tile), C.byref(sk))
        panel = tile.value in _lib.PANEL_TILES  # row-panel form: sk counts column slices, no workspace
        if (ln_in is not None or ln_out) and sk.value > 1 and not panel:
            sk.value = 1  # the separate split-K reduction launch does not carry the LayerNorm epilogue
        if tile.value == _lib.TILE_H128x128 and self.tile_override is None and (
                pw.w_halo is None or not self.lib.dm_conv_halo_supported(m, d.hin, d.win, d.cin, d.lda)):
            # the table keys a convolution by (m, n, k), not by image shape: another geometry with the same key (or weights
            # packed without the second image) runs on the ring tile of the same size and split
            tile.value = _lib.TILE_D128x128_S2
        d.tile, d.splitk = tile.value, sk.value
        self._bind_conv_image(d, pw)
        if gn_rec is not None:
            if self.tile_override is not None:
                self._bind_gn(d, gn_rec, gn_in)
            if tile.value in _lib.PANEL_TILES:
                fused_ok = self.lib.dm_gemm_panel_supported(C.byref(d), tile.value, max(sk.value, 1))
            else:
                fused_ok = self.dtype == torch.float32 and self.lib.dm_gemm_gn_supported(
                    self.dm, tile.value, d.conv, a.shape[1], d.gn_groups, gn_in["hw"], sk.value)
            if not fused_ok:
                # the chosen tile has no fragment-side form for this shape: normalise with the stand-alone kernel
                a = self._gn_unfused(a, gn_in)
                d.a, d.lda = a.data_ptr(), a.stride(0)
                d.gn_sums = None
                d.rows_per_batch = rows_per_batch
                gn_rec = None
            else:
                self._gn_auto.pop(a.data_ptr(), None)
        if ln_in is not None:
            if pw.colsum is None:
                raise RuntimeError("ln_in needs weights packed with their LayerNorm")
            d.ln_in, d.ln_colsum, d.ln_eps = ln_in.data_ptr(), pw.colsum.data_ptr(), pw.ln_eps
        elif pw.colsum is not None:
            raise RuntimeError("weights packed with a LayerNorm need the row statistics of their input (ln_in)")
        if ln_out:
            st = self._stats_alloc(m, 1).view(m, 2)
            self._ln_stats[out.data_ptr()] = st
            d.ln_out = st.data_ptr()
        if sk.value > 1 and not panel:
            self._bind_workspace(d, m, pw.n, sk.value)
        elif sk.value < 0:
            self._bind_streamk(d, tile.value, m, pw.n, -sk.value)
        if gn_batch > 0 and self.fuse_gn_stats and nchw_out is None and not pw.geglu:
            hw = m // gn_batch
            ok = (m % gn_batch == 0 and (rows_per_batch in (0, hw)) and tile.value != _lib.TILE_AUTO
                  and self.lib.dm_gemm_stats_supported(tile.value, hw, sk.value))
            self._bind_out_stats(d, out, gn_batch, hw, n_out, ok)
        with self._prof(klass, 2.0 * m * pw.n * (pw.k if conv is None else 9 * conv["cin_real"] + pw.skip_cin),
                        f"m={m} n={pw.n} k={pw.k} tile={tile.value} sk={sk.value} epi={d.epi}"
                        + (f" s={d.stride} up={d.up}" if conv is not None else "")):
            check(self.lib.dm_gemm(C.byref(d), self.stream), klass)
        return out

    def _bind_out_stats(self, d, out: torch.Tensor, gn_batch: int, hw: int, n_out: int, ok: bool):
        """A GroupNorm over ``gn_batch`` samples reads ``out`` next: point the launch descriptor ``d`` at the per-group
        sums it accumulates (the tensor's own, and those of the concatenation buffer it is a slice of), or - when the
        launch cannot produce them (``ok`` false) - mark the records incomplete so that a statistics pass runs."""
        targets = self._gn_targets(out, gn_batch)
        live = [(rec, off) for rec, off in targets if ok and rec["valid"]]
        for rec, off in targets:
            if (rec, off) in live:
                rec["parts"] += n_out
            else:
                rec["valid"] = False
        if live:
            d.rows_per_batch = hw
            (r1, o1) = live[0]
            d.stats, d.ld_stats, d.stats_cpg, d.stats_ch0 = r1["stats"].data_ptr(), r1["groups"], r1["cpg"], o1
            if len(live) > 1:
                (r2, o2) = live[1]
                d.stats2, d.ld_stats2, d.stats2_cpg, d.stats2_ch0 = r2["stats"].data_ptr(), r2["groups"], r2["cpg"], o2

    def ffn_ok(self, m: int, c: int, batch: int) -> bool:
        """Whether the feed-forward half of an AttentionBlock over ``m`` rows of width ``c`` runs as one launch."""
        # A block streams both weight images (2.6 MB) for its 128 rows whatever the row count: it pays once every CU has a
        # block (measured at 64 x 64, bf16: 0.5x the two-launch form at batch 1, 0.8x at 4, 1.1x at 8, 1.4x at 16)
        return bool(self.fuse_ffn and batch > 0 and m % batch == 0 and m >= self.ffn_min_rows
                    and self.lib.dm_ffn_geglu_supported(self.dm, m, c, m // batch, 1))

    def ffn_geglu(self, t: torch.Tensor, pw1: PackedWeight, pw2: PackedWeight, out: torch.Tensor, *,
                  residual: Optional[torch.Tensor], ln_in: torch.Tensor, gn_batch: int = 0) -> torch.Tensor:
        """``conv_output(linear_geglu_2(v * gelu(g)) + t) + residual`` with ``[v | g] = linear_geglu_1(LN(t))`` as one
        launch (dm_ffn_geglu): ``pw1`` = linear_geglu_1 packed with the GEGLU interleave and its LayerNorm
        (``pack_linear(..., geglu=True, ln=...)``), ``pw2`` = ``pack_chain(linear_geglu_2, conv_output)``, ``ln_in`` the row
        sums of ``t``.  The 4C-wide product never reaches memory."""
        m, c = t.shape
        if not (pw1.geglu and pw1.colsum is not None and pw1.k == c and pw1.n == 8 * c and pw2.skip_cin == c
                and pw2.k == 5 * c and pw2.n == c and out.shape == (m, c)):
            raise RuntimeError("feed-forward operands do not match (linear_geglu_1 with its LayerNorm, chained tail)")
        d = _lib.FfnDesc()
        d.a, d.lda = t.data_ptr(), t.stride(0)
        d.w1, d.bias1 = pw1.w.data_ptr(), pw1.bias.data_ptr()
        d.ln_in, d.ln_colsum, d.ln_eps = ln_in.data_ptr(), pw1.colsum.data_ptr(), pw1.ln_eps
        d.w2, d.bias2, d.npad2 = pw2.w.data_ptr(), pw2.bias.data_ptr(), pw2.npad
        if residual is not None:
            d.r, d.ldr = residual.data_ptr(), residual.stride(0)
        d.c, d.ldc = out.data_ptr(), out.stride(0)
        d.m, d.c_model, d.dtype = m, c, self.dm
        if gn_batch > 0 and self.fuse_gn_stats:
            hw = m // gn_batch
            ok = m % gn_batch == 0 and bool(self.lib.dm_ffn_geglu_supported(self.dm, m, c, hw, 1))
            self._bind_out_stats(d, out, gn_batch, hw, c, ok)
        with self._prof("gemm", 2.0 * m * (pw1.n * pw1.k + pw2.n * pw2.k), f"m={m} c={c} ffn (geglu_1 + tail)"):
            check(self.lib.dm_ffn_geglu(C.byref(d), self.stream), "ffn_geglu")
        return out

    @staticmethod
    def _bind_gn(d: GemmDesc, rec: dict, gn_in: dict):
        """Descriptor fields of the fragment-side GroupNorm: the producer's sums and the affine parameters."""
        d.gn_sums, d.ld_gn = rec["stats_ptr"], rec["ld"]
        d.gn_gamma, d.gn_beta = gn_in["gamma"].data_ptr(), gn_in["beta"].data_ptr()
        d.gn_eps, d.gn_silu, d.gn_groups = gn_in["eps"], int(gn_in["silu"]), gn_in.get("groups", 32)
        d.rows_per_batch = gn_in["hw"]

    def _gn_unfused(self, a: torch.Tensor, gn_in: dict) -> torch.Tensor:
        return self.groupnorm(a, gn_in["batch"], gn_in["hw"], gn_in["gamma"], gn_in["beta"], gn_in["eps"],
                              silu=bool(gn_in["silu"]), groups=gn_in.get("groups", 32))

    def _bind_workspace(self, d: GemmDesc, m: int, n: int, sk: int):
        need = self.lib.dm_gemm_workspace_bytes(m, n, sk)
        if need > self.ws.numel():
            if self.arena.frozen:
                raise RuntimeError("split-K workspace too small during graph capture")
            self._retire(self.ws)
            self.ws = torch.empty(need * 2, dtype=torch.uint8, device=self.device)
        d.workspace, d.workspace_bytes = self.ws.data_ptr(), self.ws.numel()

    def _bind_streamk(self, d: GemmDesc, tile: int, m: int, n: int, blocks: int):
        need = self.lib.dm_gemm_streamk_workspace_bytes(tile, m, n, blocks)
        if need < 0:
            raise RuntimeError("the work-list form needs a direct-to-LDS tile and at most 2M output tiles")
        if need > self.ws_sk.numel():
            if self.arena.frozen:
                raise RuntimeError("stream-K workspace too small during graph capture")
            self._retire(self.ws_sk)
            self.ws_sk = torch.zeros(need * 2, dtype=torch.uint8, device=self.device)
        d.workspace, d.workspace_bytes = self.ws_sk.data_ptr(), self.ws_sk.numel()

    # Candidate (tile, split-K) pairs are timed once per distinct problem on the real operands (the
    # launch is idempotent: it only rewrites the output) and the winner is cached for the process.
    _TUNE_TILES = (_lib.TILE_D256x256_S2, _lib.TILE_D256x128_S3, _lib.TILE_D128x128_S2, _lib.TILE_D256x64_S2, _lib.TILE_D128x64_S2,
                   _lib.TILE_D64x128_S3, _lib.TILE_D64x64_S4, _lib.TILE_D64x64_S2, _lib.TILE_D32x64_S4,
                   _lib.TILE_E128x128, _lib.TILE_E128x64, _lib.TILE_H128x128) + _lib.PAIR_TILES
    _EARLY_REFILL = (_lib.TILE_E128x128, _lib.TILE_E128x64)  # measured to win only on some bf16 shapes
    _TUNE_SK = (1, 2, 3, 4, 6, 8, 12, 16, 24, 32)
    _TUNE_STREAMK = (256, 512, 768, 1024)  # blocks sharing the (tile, k-tile) work list
    _TUNE_STREAMK_MULT = (2, 3, 4, 6, 8, 12, 16, 24)  # ... and whole multiples of the tile count
    _TUNE_PANEL_SLICES = (1, 2, 3, 4, 5, 8, 10, 16, 20)  # column slices (grid.y) of the row-panel tiles
    _TUNE_LEGACY_SK = False  # split-K with a separate reduction launch: measured slower overall than the work-list
                             # form once GroupNorm / LayerNorm statistics ride in the GEMM epilogue

    def _time_candidate(self, d: GemmDesc, reps: int = int(os.environ.get("DM_TUNE_REPS", "6"))) -> Optional[float]:
        """Device time (ms) of one launch of ``d``.  The launches are captured into a hipGraph and the replay is
        timed, so host launch latency (which exceeds the run time of the small problems) is not measured;
        on the legacy default stream, where capture is not allowed, back-to-back launches are timed."""
        stream = self.stream
        if self.lib.dm_gemm(C.byref(d), stream) != 0:
            return None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        graph = C.c_void_p()
        if stream != 0 and getattr(self, "_tune_graph", True) and self.lib.dm_graph_begin(stream) == 0:
            for _ in range(reps):
                self.lib.dm_gemm(C.byref(d), stream)
            if self.lib.dm_graph_end(stream, C.byref(graph)) != 0:
                graph = C.c_void_p()
        if graph:
            self.lib.dm_graph_launch(graph, stream)
            e0.record()
            self.lib.dm_graph_launch(graph, stream)
            e1.record()
            e1.synchronize()
            self.lib.dm_graph_destroy(graph)
        else:
            e0.record()
            for _ in range(reps):
                self.lib.dm_gemm(C.byref(d), stream)
            e1.record()
            e1.synchronize()
        return e0.elapsed_time(e1) / reps

    def ln_stats(self, t: torch.Tensor) -> torch.Tensor:
        """Row sums accumulated by the GEMM that wrote ``t`` (``ln_out=True``)."""
        return self._ln_stats[t.data_ptr()]

    def _bind_conv_image(self, d: GemmDesc, pw: PackedWeight):
        """The halo tile reads its own image of a convolution's weights; every other tile the ordinary one."""
        if d.tile == _lib.TILE_H128x128:
            if pw.w_halo is None:
                raise RuntimeError("the halo tile needs convolution weights packed with same_size=True on a bf16 engine")
            d.w, d.conv_chunk_major = pw.w_halo.data_ptr(), 2
        elif d.conv:
            d.w, d.conv_chunk_major = pw.w.data_ptr(), int(pw.chunk_major)

    def _bind_choice(self, d: GemmDesc, tile: int, sk: int):
        d.tile, d.splitk = tile, sk
        if d.conv and getattr(self, "_tune_pw", None) is not None:
            self._bind_conv_image(d, self._tune_pw)
        if tile in _lib.PANEL_TILES:
            return
        if sk > 1:
            self._bind_workspace(d, d.m, d.n, sk)
        elif sk < 0:
            self._bind_streamk(d, tile, d.m, d.n, -sk)

    def _tune(self, d: GemmDesc, key, legacy: bool = True):
        nkt = d.kpad // self.bk
        best, best_t = None, float("inf")
        seen = []  # (time, (tile, split)) of every candidate that ran
        for tile in self._TUNE_TILES:
            bm, bn = _lib.TILE_DIMS[tile]
            if d.npad < _rup(d.n, bn) or (tile in (_lib.TILE_D256x128_S3, _lib.TILE_D256x256_S2) and d.m < 2048):
                continue
            if tile == _lib.TILE_D256x256_S2 and self.dtype != torch.bfloat16:
                continue
            if tile == _lib.TILE_H128x128:
                pw_t = getattr(self, "_tune_pw", None)
                if (pw_t is None or pw_t.w_halo is None or not d.conv
                        or not self.lib.dm_conv_halo_supported(d.m, d.hin, d.win, d.cin, d.lda)):
                    continue
            if tile in self._EARLY_REFILL and (self.dtype != torch.bfloat16 or not self.tune_early):
                continue
            blocks = ((d.m + bm - 1) // bm) * ((d.n + bn - 1) // bn)
            for sk in self._TUNE_SK:
                if sk > 1 and not (self._TUNE_LEGACY_SK and legacy):
                    break
                if sk > 1 and (d.epi != _lib.EPI_PLAIN or nkt // sk < 2 or blocks * sk > 8192
                               or d.m * d.n * sk * 4 > (256 << 20)):
                    break
                d.tile, d.splitk = tile, sk
                if d.conv:
                    self._bind_conv_image(d, self._tune_pw)
                if sk > 1:
                    self._bind_workspace(d, d.m, d.n, sk)
                t = self._time_candidate(d)
                if t is not None:
                    seen.append((t, (tile, sk)))
                if t is not None and t < best_t:
                    best, best_t = (tile, sk), t
                if blocks * sk >= 2048 and sk > 1:  # already many rounds of blocks; more slices only add slab traffic
                    break
            grids = set(self._TUNE_STREAMK) if self._TUNE_STREAMK else set()
            if self._TUNE_STREAMK:
                grids.update(blocks * s for s in self._TUNE_STREAMK_MULT if blocks * s <= 2048 and nkt // s >= 2)
            for grid in sorted(grids):
                if grid >= 2 * blocks * nkt:
                    break
                d.tile, d.splitk = tile, -grid
                if d.conv:
                    self._bind_conv_image(d, self._tune_pw)
                if self.lib.dm_gemm_streamk_workspace_bytes(tile, d.m, d.n, grid) < 0:
                    break
                self._bind_streamk(d, tile, d.m, d.n, grid)
                t = self._time_candidate(d)
                if t is not None:
                    seen.append((t, (tile, -grid)))
                if t is not None and t < best_t:
                    best, best_t = (tile, -grid), t
        # row-panel form (bf16, K = 320 / 640): a block keeps its rows in registers and walks all column tiles, or the
        # slice of them that one of `ns` column slices holds
        d.workspace, d.workspace_bytes = None, 0
        for tile, bm in _lib.PANEL_TILES.items():
            panels = (d.m + bm - 1) // bm
            for ns in self._TUNE_PANEL_SLICES:
                if ns > 1 and panels * ns > 4096:
                    break
                if not self.lib.dm_gemm_panel_supported(C.byref(d), tile, ns):
                    continue
                d.tile, d.splitk = tile, ns
                t = self._time_candidate(d)
                if t is not None:
                    seen.append((t, (tile, ns)))
                if t is not None and t < best_t:
                    best, best_t = (tile, ns), t
        # The sweep times every candidate once; candidates within a few percent of each other change places from run to
        # run.  The three fastest are timed again, interleaved, three more times each, and the smallest median wins.
        # A work list whose block count is a whole multiple of the tile count cuts every tile at the same k positions, so
        # the blocks that hold the same k range of neighbouring row tiles (same XCD, same time) read the same weight
        # bytes and all but one hit in L2; any other count shifts the cuts from tile to tile and every block fetches its
        # weights itself (measured on the 1920 -> 640 conv at 32 x 32, batch 1: 873 MB of L2 misses with 512 blocks,
        # 287 MB with 480 = 6 x 80 tiles, 3 % slower).  The fastest such candidate joins the finalists and wins when it is
        # within 3 % of the fastest.
        def aligned(c):
            tile, sk = c
            if tile in _lib.PANEL_TILES:
                return False
            bm, bn = _lib.TILE_DIMS[tile]
            tiles = ((d.m + bm - 1) // bm) * ((d.n + bn - 1) // bn)
            return sk == 1 or (sk < 0 and (-sk) % tiles == 0)

        ordered = [c for _, c in sorted(seen)]
        finalists = ordered[:3]
        extra = next((c for c in ordered if aligned(c)), None)
        if extra is not None and extra not in finalists:
            finalists.append(extra)
        if len(finalists) > 1:
            times = {c: [] for c in finalists}
            for _ in range(3):
                for c in finalists:
                    self._bind_choice(d, *c)
                    t = self._time_candidate(d)
                    if t is not None:
                        times[c].append(t)
            ranked = sorted((sorted(v)[len(v) // 2], c) for c, v in times.items() if v)
            if ranked:
                best = ranked[0][1]
                near = [c for t, c in ranked if t <= 1.03 * ranked[0][0] and aligned(c)]
                if near:
                    best = near[0]
        if best is not None:
            self.tuned[key] = best
            if self._tune_cache:
                self._save_tuned()
        return best

    def _save_tuned(self):
        import json
        data = {}
        if os.path.exists(self._tune_cache):
            with open(self._tune_cache) as f:
                data = json.load(f)
        data[str(self.dtype)] = {repr(k): list(v) for k, v in self.tuned.items()}
        tmp = self._tune_cache + ".tmp"
        with open(tmp, "w") as f:
            json.dump(data, f)
        os.replace(tmp, self._tune_cache)

    def groupnorm(self, x: torch.Tensor, batch: int, hw: int, gamma, beta, eps: float, silu: bool,
                  out: Optional[torch.Tensor] = None, groups: int = 32) -> torch.Tensor:
        c = x.shape[1]
        rec = self._gn_lookup(x, batch, groups)
        if out is None:
            out = self.new(batch * hw, c)
        if rec is not None:
            self._gn_auto.pop(x.data_ptr(), None)
            with self._prof("groupnorm", float(batch * hw * c * self.esz * 2), f"b={batch} hw={hw} c={c} fused-stats"):
                check(self.lib.dm_groupnorm_apply_sums(x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0), batch,
                                                       hw, c, groups, self.dm, rec["stats_ptr"], rec["ld"],
                                                       gamma.data_ptr(), beta.data_ptr(), eps, int(silu), self.stream),
                      "groupnorm_apply_sums")
            return out
        nsplit = self.lib.dm_groupnorm_default_nsplit(batch, hw)
        part = self.arena.alloc(batch * nsplit, groups * 2, torch.float32)
        with self._prof("groupnorm", float(batch * hw * c * self.esz * 2), f"b={batch} hw={hw} c={c}"):
            check(self.lib.dm_groupnorm_stats(x.data_ptr(), x.stride(0), batch, hw, c, groups, self.dm,
                                              part.data_ptr(), nsplit, self.stream), "groupnorm_stats")
            check(self.lib.dm_groupnorm_apply(x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0), batch, hw, c,
                                              groups, self.dm, part.data_ptr(), nsplit, gamma.data_ptr(),
                                              beta.data_ptr(), eps, int(silu), self.stream), "groupnorm_apply")
        return out

    def pack_rows(self, x: torch.Tensor, transposed: bool = False) -> PackedWeight:
        """An activation matrix as the weight operand of a GEMM: ``x`` [n, k] (or, ``transposed``, [k, n]) re-laid on
        the device into the packed layout (arena memory: valid for the current forward)."""
        rows, cols = x.shape
        n, k = (cols, rows) if transposed else (rows, cols)
        npad, kpad = _rup(n, 256), _rup(k, self.bk)
        w = self.new(npad, kpad)
        check(self.lib.dm_pack_rows(x.data_ptr(), x.stride(0), n, k, int(transposed), w.data_ptr(), npad, kpad, self.dm,
                                    self.stream), "pack_rows")
        return PackedWeight(w, n, k, npad, kpad, None)

    def softmax_rows(self, x: torch.Tensor, scale: float) -> torch.Tensor:
        rows, cols = x.shape
        with self._prof("softmax", float(rows * cols * self.esz * 2), f"rows={rows} cols={cols}"):
            check(self.lib.dm_softmax_rows(x.data_ptr(), x.stride(0), rows, cols, float(scale), self.dm, self.stream),
                  "softmax_rows")
        return x

    def layernorm(self, x: torch.Tensor, gamma, beta, eps: float = 1e-5, out=None) -> torch.Tensor:
        rows, c = x.shape
        if out is None:
            out = self.new(rows, c)
        with self._prof("layernorm", float(rows * c * self.esz * 2), f"rows={rows} c={c}"):
            check(self.lib.dm_layernorm(x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0), rows, c, self.dm,
                                        gamma.data_ptr(), beta.data_ptr(), eps, self.stream), "layernorm")
        return out

    def attention(self, q, k, v, batch, heads, sq, sk, d, out=None, causal=False) -> torch.Tensor:
        if out is None:
            out = self.new(batch * sq, heads * d)
        with self._prof("attention", 4.0 * batch * heads * sq * sk * d, f"b={batch} sq={sq} sk={sk} d={d}"):
            if self.split_bf16 and self.split_bf16 != 3:  # the two-term product mode covers both attention products too
                check(self.lib.dm_attention_split_f32(q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0), v.data_ptr(),
                                                      v.stride(0), out.data_ptr(), out.stride(0), batch, heads, sq, sk,
                                                      d, int(causal), self.stream), "attention")
            else:
                check(self.lib.dm_attention(q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0), v.data_ptr(),
                                            v.stride(0), out.data_ptr(), out.stride(0), batch, heads, sq, sk, d,
                                            int(causal), self.dm, self.stream), "attention")
        return out

    def embed_tokens(self, tokens: torch.Tensor, table: torch.Tensor, pos: torch.Tensor) -> torch.Tensor:
        """tokens (B, S) int64, table fp32 (V, E), pos fp32 (S, E) -> engine-dtype rows [B*S, E]."""
        b, s = tokens.shape
        v, e = table.shape
        if pos.shape != (s, e):
            raise RuntimeError(f"sequence of {s} tokens does not match the {tuple(pos.shape)} position table")
        out = self.new(b * s, e)
        tok = tokens.to(device=self.device, dtype=torch.long).contiguous()
        check(self.lib.dm_embed_tokens(tok.data_ptr(), table.data_ptr(), pos.data_ptr(), out.data_ptr(), out.stride(0),
                                       b * s, s, e, v, self.dm, self.stream), "embed_tokens")
        return out

    def linear_rows(self, x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], silu_in: bool,
                    out: Optional[torch.Tensor] = None) -> torch.Tensor:
        batch, k = x.shape
        n = w.shape[0]
        if out is None:
            out = self.arena.alloc(batch, n, torch.float32)
        with self._prof("time_linear", float(n * k * 4)):
            check(self.lib.dm_linear_rows(x.data_ptr(), x.stride(0), w.data_ptr(), b.data_ptr() if b is not None else None,
                                          out.data_ptr(), out.stride(0), batch, n, k, int(silu_in), self.stream),
                  "linear_rows")
        return out

    def nchw_to_rows(self, x: torch.Tensor, cpad: Optional[int] = None, out=None) -> torch.Tensor:
        b, c, h, w = x.shape
        cpad = cpad or c
        x = x.contiguous()
        if x.dtype != torch.float32:
            x = x.float()
        if out is None:
            out = self.new(b * h * w, cpad)
        check(self.lib.dm_nchw_to_rows(x.data_ptr(), b, c, h * w, out.data_ptr(), out.stride(0), cpad, self.dm,
                                       self.stream), "nchw_to_rows")
        return out

    def rows_to_nchw(self, rows: torch.Tensor, b: int, c: int, h: int, w: int, out=None) -> torch.Tensor:
        if out is None:
            out = torch.empty(b, c, h, w, dtype=torch.float32, device=self.device)
        check(self.lib.dm_rows_to_nchw(rows.data_ptr(), rows.stride(0), b, c, h * w, out.data_ptr(), self.dm,
                                       self.stream), "rows_to_nchw")
        return out

    def to_rows(self, x: torch.Tensor) -> torch.Tensor:
        """(B, S, C) fp32/bf16 tensor -> engine-dtype [B*S, C] rows (arena)."""
        b, s, c = x.shape
        x = x.contiguous()
        if x.dtype == self.dtype:
            return x.view(b * s, c)
        out = self.new(b * s, c)
        if x.dtype != torch.float32:
            x = x.float()
        if self.dtype == torch.float32:
            return x.view(b * s, c)
        check(self.lib.dm_convert(x.data_ptr(), _lib.DM_F32, out.data_ptr(), self.dm, b * s * c, self.stream), "convert")
        return out

    def from_rows(self, rows: torch.Tensor, shape) -> torch.Tensor:
        """engine-dtype contiguous rows -> new fp32 torch tensor of ``shape``."""
        out = torch.empty(shape, dtype=torch.float32, device=self.device)
        n = out.numel()
        if self.dtype == torch.float32:
            out.view(-1).copy_(rows.reshape(-1))
        else:
            check(self.lib.dm_convert(rows.data_ptr(), self.dm, out.data_ptr(), _lib.DM_F32, n, self.stream), "convert")
        return out

    # ------------------------------------------------------------------ profiling
    def start_profile(self):
        self.profile = {}

    def stop_profile(self) -> Dict[str, dict]:
        """Synchronise and return {class: {"ms": total, "launches": n, "work": sum, "detail": [(tag, ms, work)]}}."""
        prof, self.profile = self.profile, None
        out = {}
        ms = C.c_float()
        for klass, items in prof.items():
            tot, work, detail = 0.0, 0.0, []
            for e0, e1, w, tag in items:
                check(self.lib.dm_event_elapsed_ms(e0, e1, C.byref(ms)), "event_elapsed")
                tot += ms.value
                work += w
                detail.append((tag, ms.value, w))
                self.lib.dm_event_destroy(e0)
                self.lib.dm_event_destroy(e1)
            out[klass] = {"ms": tot, "launches": len(items), "work": work, "detail": detail}
        return out


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


_NULLCTX = _Null()


class _Timed:
    def __init__(self, eng: Engine, klass: str, work: float, tag: str = ""):
        self.eng, self.klass, self.work, self.tag = eng, klass, work, tag

    def __enter__(self):
        self.e0, self.e1 = C.c_void_p(), C.c_void_p()
        check(self.eng.lib.dm_event_create(C.byref(self.e0)), "event_create")
        check(self.eng.lib.dm_event_create(C.byref(self.e1)), "event_create")
        check(self.eng.lib.dm_event_record(self.e0, self.eng.stream), "event_record")
        return self

    def __exit__(self, *a):
        check(self.eng.lib.dm_event_record(self.e1, self.eng.stream), "event_record")
        self.eng.profile.setdefault(self.klass, []).append((self.e0, self.e1, self.work, self.tag))
        return False
