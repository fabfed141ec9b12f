// LDS-tiled MFMA GEMM / implicit-GEMM conv3x3 for gfx950.
//
//   C[m][n] = epilogue( sum_k A[m][k] * W[n][k] )
//
// A is the activation matrix (rows = pixels/tokens, channel-contiguous).  In conv mode the
// k axis is (tap, cin) and row m = (b, y, x) gathers its 3x3 neighbourhood on the fly from an
// NHWC image (zero padding 1; stride 1 or 2; optional nearest x2 upsampling of the input), so
// no im2col buffer ever exists.  W is a pre-packed weight matrix of Npad x Kpad elements, zero padded,
// rows permuted inside every group of 64 so that each lane ends up owning 16 consecutive output
// columns (see pack order below), and stored blocked as [Npad/64][Kpad/BK][64 rows][128 bytes]: the
// k-tile of a 64-row group is one contiguous 8 KiB piece and the group's whole K panel one contiguous
// stream, so weight-bound shapes (few rows of A) read HBM in long bursts instead of 128-byte pieces
// that are a full weight row apart.
//
// Block tile BM x BN x (128 bytes of k); 16x16 MFMA fragments; each wave owns a
// (16*FM) x 64 sub-tile.  LDS tiles are 128-byte rows with an XOR swizzle of the 16-byte chunk
// index (chunk ^ ((row >> 1) & 7)) that makes the fragment reads (ds_read_b128) conflict-free.
//
// Two main loops share the epilogue:
//   * "direct" (gemm_dma_kernel): operands go global -> LDS with global_load_lds_dwordx4 into a
//     ring of STAGES tiles; the LDS image is lane-linear, so the swizzle is applied to the per-lane
//     SOURCE address; padding / out-of-range lanes read a 16-byte zero buffer; counted
//     s_waitcnt vmcnt(N) + one raw s_barrier per k-tile keep STAGES-1 tiles in flight.
//   * "staged" (gemm_reg_kernel): global -> registers -> LDS, two LDS buffers, one barrier per k-tile.
//
// Packed row order: packed row p = 64*g + 16*j + 4*q + r holds weight row
//   plain : n = 64*g + 16*q + 4*j + r
//   GEGLU : value half (j < 2) / gate half (j >= 2) of output column 32*g + 8*q + 4*(j&1) + r
// so that after the MFMA lane (q = lane>>4) holds columns 64*g + 16*q .. +15 of row m = lane&15
// (GEGLU: 8 value columns and their 8 gates).
#include <cstdlib>

#include "common.h"
#include "dm_hip.h"
#include "panel.h"

namespace {

struct GemmArgs {
    const char* A;
    long lda;  // elements
    const char* W;
    char* C;
    long ldc;
    const float* bias;
    const float* rowbias;
    long ld_rowbias;
    int rows_per_batch;
    const char* R;
    long ldr;
    int M, N, K, Kpad;
    int conv, Hin, Win, Hout, Wout, Cin, stride, up;
    int padlo;  // zero padding above / left of the image: 1 (padding 1 all round) or 0 (padding only below / right)
    const char* A2;  // second operand appended to the k axis, Cin2 elements per row: conv: a pointwise term (tap 9,
                     // the pixel itself); dense: columns K-Cin2 .. K-1 of A come from this tensor instead
    long lda2;
    int Cin2;
    int epi;
    int splitk;
    float* partial;
    const double* ln_in;     // per-row (sum, sumsq) of A: the LayerNorm that precedes this GEMM is folded in
    const float* ln_colsum;  // sum over k of each packed weight row (packed order)
    float ln_eps;
    double* ln_out;          // per-row (sum, sumsq) of the stored output, atomically accumulated
    DmGroupStats stats, stats2;  // per-group (sum, sumsq) of the stored output for the GroupNorm(s) that read it (common.h)
    int streamk;    // stream-K: number of blocks sharing the (tile, k-tile) work list; 0 = one tile per block
    int* counters;  // stream-K: one arrival counter per output tile, zero on entry, left zero
    float* slabs;   // stream-K: 2 accumulator slabs per block
    // GroupNorm (+ SiLU) of the A operand applied to the fragments in registers (fp32, GN kernels only)
    const double* gn_sums;  // per-group (sum, sumsq) of A, [batch][2][ld_gn]
    long ld_gn;
    const float* gn_gamma;
    const float* gn_beta;
    float gn_eps;
    int gn_groups;
    int gn_c;       // channels that are normalised: Cin (conv) or K - Cin2 (dense)
    int korder;     // conv: 1 = chunk-major k axis (k-tile = chunk * 9 + tap), see DmGemmDesc.conv_chunk_major
};

// Chunk-major convolution: (tap, channel offset) of k-tile kt, and the step to the next k-tile.  Tap 9 is the appended
// pointwise operand, whose channel offset simply runs on.
__device__ __forceinline__ void chunk_major_at(int kt, int cin, int bk, int& tap, int& c0) {
    const int nine = 9 * (cin / bk);
    if (kt < nine) {
        const int chunk = kt / 9;
        tap = kt - 9 * chunk;
        c0 = chunk * bk;
    } else {
        tap = 9;
        c0 = (kt - nine) * bk;
    }
}
__device__ __forceinline__ void chunk_major_next(int cin, int bk, int& tap, int& c0) {
    if (tap >= 9) {
        c0 += bk;
    } else if (++tap == 9) {
        c0 += bk;
        if (c0 < cin) tap = 0;
        else c0 = 0;
    }
}

// source of every lane whose 16 bytes must read as zero in the direct-to-LDS loop
__device__ __attribute__((aligned(16))) uint32_t g_zero16[4];

// Diagnostic build only (python build.py --timeline -> libdm_hip_tl.so, read with tools/timeline_gemm.py): thread 0 of
// every block of the generic kernel stamps the 100 MHz wall clock at the phase boundaries of its first two work-list
// segments.  The shipped library compiles none of this.
#ifdef DM_TIMELINE
constexpr int TL_BLOCKS = 4096, TL_SLOTS = 16;
__device__ unsigned long long g_timeline[TL_BLOCKS * TL_SLOTS];
#define DM_TL(seg, i)                                                                            \
    do {                                                                                          \
        if (threadIdx.x == 0 && blockIdx.x < TL_BLOCKS && (seg) < 2)                              \
            g_timeline[blockIdx.x * TL_SLOTS + (seg) * 8 + (i)] = wall_clock64();                 \
    } while (0)
#define DM_TL_CYCLES_BEGIN() tl_cycles = clock64()
#define DM_TL_CYCLES_END(seg)                                                                     \
    do {                                                                                          \
        if (threadIdx.x == 0 && blockIdx.x < TL_BLOCKS && (seg) < 2)                              \
            g_timeline[blockIdx.x * TL_SLOTS + (seg) * 8 + 7] = clock64() - tl_cycles;            \
    } while (0)
// Knock-out switches of the diagnostic build (dm_timeline_variant): bit 1 = no DMA issue inside the k loop, bit 2 = no
// waits / barrier after the first k-tile, bit 8 = no epilogue at all, bit 16 = chunk-major convolution: only tap 0 of a
// chunk reads the activation (the other taps' requests all read one 16-byte line), bit 32 = every weight request reads
// that one line.  Results are garbage; only durations are of interest.
__device__ int g_tl_variant;
#define DM_TL_ON(bit) (!(tl_variant & (bit)))
#else
#define DM_TL(seg, i) do { } while (0)
#define DM_TL_ON(bit) true
#define DM_TL_CYCLES_BEGIN() do { } while (0)
#define DM_TL_CYCLES_END(seg) do { } while (0)
#endif

__device__ __forceinline__ int swz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// Byte offset of packed weight row `row`, k-tile `kt`, in the blocked image [Npad/64][nkt][64][128 B];
// from one k-tile to the next a row's piece moves by W_KTILE_STEP bytes.
constexpr int W_KTILE_STEP = 64 * 128;
__device__ __forceinline__ long w_block_offset(int row, int nkt, int kt) {
    return (((long)(row >> 6) * nkt + kt) * 64 + (row & 63)) * 128;
}

// XCD-aware tile id: blocks b, b+8, ... share an XCD (and its L2); give each XCD a contiguous run
// of logical tiles so that neighbours sharing A rows / W columns hit the same L2.  Bijective for any nblk.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// Per-thread coordinates of the A rows it stages (fixed over the k loop).
template <int A_IT> struct ARows {
    long off[A_IT];  // dense: element offset of the row; conv: pixel index of image b's origin
    int y[A_IT], x[A_IT];
    bool ok[A_IT];
};

template <int A_IT, int ROWS_PER_IT>
__device__ __forceinline__ void init_rows(ARows<A_IT>& r, const GemmArgs& p, int m0, int row0, int bm) {
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        const int row = it * ROWS_PER_IT + row0;
        const int m = m0 + row;
        r.ok[it] = (row < bm) && (m < p.M);
        r.y[it] = r.x[it] = 0;
        r.off[it] = 0;
        if (r.ok[it]) {
            if (p.conv) {
                const int hw = p.Hout * p.Wout;
                const int b = m / hw;
                const int pix = m - b * hw;
                r.y[it] = pix / p.Wout;
                r.x[it] = pix - r.y[it] * p.Wout;
                r.off[it] = (long)b * p.Hin * p.Win;
            } else {
                r.off[it] = (long)m * p.lda;
                r.y[it] = m;  // dense: the row index itself (second-operand addressing)
            }
        }
    }
}

// Element offset (and validity) of the 16-byte piece of A for staged row `it`, conv tap `tap`, channel `kc`.
template <int A_IT>
__device__ __forceinline__ bool conv_src(const ARows<A_IT>& r, const GemmArgs& p, int it, int tap, int kc, long& e) {
    const int dy = tap / 3 - p.padlo, dx = tap - (tap / 3) * 3 - p.padlo;
    int iy, ix;
    bool ok = r.ok[it] && tap < 9;
    if (p.up) {
        const int uy = r.y[it] + dy, ux = r.x[it] + dx;
        ok = ok && (unsigned)uy < (unsigned)p.Hout && (unsigned)ux < (unsigned)p.Wout;
        iy = uy >> 1;
        ix = ux >> 1;
    } else {
        iy = r.y[it] * p.stride + dy;
        ix = r.x[it] * p.stride + dx;
        ok = ok && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
    }
    e = (r.off[it] + (long)iy * p.Win + ix) * p.lda + kc;
    return ok;
}

// One k-tile of MFMAs for this wave from the LDS tile at (sa, sw).
template <typename T, int FM>
__device__ __forceinline__ void mma_tile(f32x4 (&acc)[FM][4], const char* sa, const char* sw, int arow0, int wrow0, int fr,
                                         int fq) {
    if constexpr (sizeof(T) == 2 && FM == 4) {
        // bf16: every fragment of the k-tile is requested before the first MFMA, so that only the first
        // k-step's LDS latency is exposed; the second k-step's reads land behind the first one's MFMAs
        // (LDS returns in order, so the waits in between are counted ones).
        u32x4 af[2][FM], wf[2][4];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int j = 0; j < 4; ++j) wf[ks][j] = *reinterpret_cast<const u32x4*>(sw + swz(wrow0 + j * 16 + fr, ks * 4 + fq));
#pragma unroll
            for (int i = 0; i < FM; ++i) af[ks][i] = *reinterpret_cast<const u32x4*>(sa + swz(arow0 + i * 16 + fr, ks * 4 + fq));
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the reads ahead of the MFMAs (the scheduler would sink them)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) Mma16<T>::mma(acc[i][j], wf[ks][j], af[ks][i]);
        return;
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        u32x4 af[FM], wf[4];
#pragma unroll
        for (int i = 0; i < FM; ++i) af[i] = *reinterpret_cast<const u32x4*>(sa + swz(arow0 + i * 16 + fr, ks * 4 + fq));
#pragma unroll
        for (int j = 0; j < 4; ++j) wf[j] = *reinterpret_cast<const u32x4*>(sw + swz(wrow0 + j * 16 + fr, ks * 4 + fq));
        if constexpr (sizeof(T) == 4) {
            // 16x16x4 f32 MFMA has a 40-cycle dependent latency against a 32-cycle issue: walk all
            // accumulators for one k element before touching any of them again
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(wf[j][e]),
                                                                        __uint_as_float(af[i][e]), acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) Mma16<T>::mma(acc[i][j], wf[j], af[i]);
        }
    }
}

// The same k-tile in two steps, for the loop that refills an LDS buffer as soon as its contents sit in
// registers: all fragment reads of the tile, then (after the caller's barrier and refill) all MFMAs.
template <typename T, int FM>
__device__ __forceinline__ void read_tile_frags(u32x4 (&af)[2][FM], u32x4 (&wf)[2][4], const char* sa, const char* sw,
                                                int arow0, int wrow0, int fr, int fq) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int j = 0; j < 4; ++j) wf[ks][j] = *reinterpret_cast<const u32x4*>(sw + swz(wrow0 + j * 16 + fr, ks * 4 + fq));
#pragma unroll
        for (int i = 0; i < FM; ++i) af[ks][i] = *reinterpret_cast<const u32x4*>(sa + swz(arow0 + i * 16 + fr, ks * 4 + fq));
    }
}
template <typename T, int FM>
__device__ __forceinline__ void mfma_tile_frags(f32x4 (&acc)[FM][4], const u32x4 (&af)[2][FM], const u32x4 (&wf)[2][4]) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        if constexpr (sizeof(T) == 4) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(wf[ks][j][e]),
                                                                        __uint_as_float(af[ks][i][e]), acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) Mma16<T>::mma(acc[i][j], wf[ks][j], af[ks][i]);
        }
    }
}

// ---- GroupNorm of the A operand inside the GEMM (fp32) ----------------------------------------------
// The tensor a GroupNorm reads has exactly one consumer here (the convolution or 1x1 projection behind it), and
// its per-channel sums already exist (the launch that wrote it accumulated them).  Instead of a launch that
// reads the tensor, normalises it and writes it back, the consumer normalises its A fragments on their way from
// LDS to the MFMA: y = x * scale[c] + shift[c] (+ SiLU), scale = rstd * gamma, shift = beta - mean * scale.  With the
// exact-fp32 MFMA (32 cycles for 4 k values of a 16 x 16 tile) a lane has one A register per four MFMAs, so the
// extra VALU work hides behind the matrix pipe; with bf16 operands it would not (eight values per register, 16-cycle
// MFMAs), which is why only the fp32 kernels have this form.
//
// Scale / shift of the sample's C channels live in LDS behind the operand ring: scale[C] | shift[C] | scratch.
constexpr int GN_FUSE_MAX_C = 1280;
constexpr int GN_FUSE_SCRATCH = 2 * 64 * 4;  // group mean / rstd
__host__ __device__ constexpr int gn_fuse_lds_bytes(int c) { return 2 * c * 4 + GN_FUSE_SCRATCH; }

// Same arithmetic as gn_apply_kernel (norm.hip): mean and variance of a group from the producer's per-group fp64 sums.
template <int NT>
__device__ __forceinline__ void gn_build_table(const GemmArgs& p, int b, float* tab) {
    const int c = p.gn_c, groups = p.gn_groups, cpg = c / groups;
    float* t_scale = tab;
    float* t_shift = tab + c;
    float* g_mean = tab + 2 * c;  // [64]
    float* g_rstd = g_mean + 64;
    const int tid = threadIdx.x;
    // affine parameters of this thread's channels, requested now: their latency overlaps the statistics fold
    constexpr int CH = (GN_FUSE_MAX_C + NT - 1) / NT;
    float gam[CH], bet[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k) {
        const int ch = tid + k * NT;
        gam[k] = ch < c ? p.gn_gamma[ch] : 0.f;
        bet[k] = ch < c ? p.gn_beta[ch] : 0.f;
    }
    const double* src = p.gn_sums + (long)b * 2 * p.ld_gn;
    for (int grp = tid; grp < groups; grp += NT)
        group_mean_rstd(src[grp], src[p.ld_gn + grp], (double)p.rows_per_batch * cpg, p.gn_eps, g_mean[grp], g_rstd[grp]);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < CH; ++k) {
        const int ch = tid + k * NT;
        if (ch < c) {
            const int grp = ch / cpg;
            const float sc = g_rstd[grp] * gam[k];
            t_scale[ch] = sc;
            t_shift[ch] = fmaf(-g_mean[grp], sc, bet[k]);
        }
    }
    __syncthreads();
}

// One k-tile of fp32 MFMAs with the A fragments normalised on the way.  ts / tt: scale / shift of the k-tile's first
// channel; vmask[i] bit `bit`: row i of this lane reads real data in this k-tile (a padding tap or a row past M must
// stay zero AFTER the affine map); raw: the k-tile belongs to the appended operand, which is not normalised.
// A lane's register e of a fragment feeds only the MFMAs of k element e, so the map runs one element ahead of the
// matrix pipe: while the 4*FM MFMAs of element e execute (32 cycles each), the VALU maps element e+1 (seven
// instructions per value, two of them transcendental); only the first element of a k-tile is mapped with the pipe idle.
// The map of one value as seven single-instruction stages, so that the caller can place them between MFMAs.
struct GnVal {
    float v, u;
};
template <bool SILU>
__device__ __forceinline__ void gn_op(int stage, GnVal& x, float a, float s, float t, bool live) {
    switch (stage) {
        case 0: x.v = fmaf(a, s, t); break;
        case 1: if (SILU) x.u = -1.4426950408889634f * x.v; break;
        case 2: if (SILU) x.u = __builtin_amdgcn_exp2f(x.u); break;
        case 3: if (SILU) x.u = 1.0f + x.u; break;
        case 4: if (SILU) x.u = __builtin_amdgcn_rcpf(x.u); break;
        case 5: if (SILU) x.v = x.v * x.u; break;
        default: x.v = live ? x.v : 0.f; break;
    }
}
template <int FM, bool SILU>
__device__ __forceinline__ void mma_tile_gn(f32x4 (&acc)[FM][4], const char* sa, const char* sw, int arow0, int wrow0,
                                            int fr, int fq, const f32x4 (&s4)[2], const f32x4 (&t4)[2],
                                            const uint32_t (&vmask)[FM], int bit, bool raw) {
    // LDS returns in order: the first k-step's A fragments are requested first, so the first element's map starts
    // while the other fragments are still on their way (the scale / shift chunks were requested before the k-tile's
    // barrier by the caller)
    u32x4 af[2][FM], wf[2][4];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int i = 0; i < FM; ++i) af[ks][i] = *reinterpret_cast<const u32x4*>(sa + swz(arow0 + i * 16 + fr, ks * 4 + fq));
#pragma unroll
        for (int j = 0; j < 4; ++j) wf[ks][j] = *reinterpret_cast<const u32x4*>(sw + swz(wrow0 + j * 16 + fr, ks * 4 + fq));
        if (ks == 0) __builtin_amdgcn_sched_barrier(0);
    }
    if (raw) {
        mfma_tile_frags<float, FM>(acc, af, wf);
        return;
    }
    bool live[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) live[i] = (vmask[i] >> bit) & 1u;
    GnVal cur[FM], nxt[FM];
#pragma unroll
    for (int st = 0; st < 7; ++st)
#pragma unroll
        for (int i = 0; i < FM; ++i) gn_op<SILU>(st, cur[i], __uint_as_float(af[0][i][0]), s4[0][0], t4[0][0], live[i]);
#pragma unroll
    for (int step = 0; step < 8; ++step) {
        const int ks = step >> 2, e = step & 3;
        const int ks1 = ((step + 1) >> 2) & 1, e1 = (step + 1) & 3;
        __builtin_amdgcn_sched_barrier(0);
        // the order below is the order of issue: one MFMA, two instructions of the next element's map, ...
#pragma unroll
        for (int q = 0; q < 4 * FM; ++q) {
            const int i = q >> 2, j = q & 3;
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(wf[ks][j][e]), cur[i].v, acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int op = 2 * q + r;  // stage-major over the FM values, so that neighbouring instructions are independent
                if (step < 7 && op < 7 * FM)
                    gn_op<SILU>(op / FM, nxt[op % FM], __uint_as_float(af[ks1][op % FM][e1]), s4[ks1][e1], t4[ks1][e1],
                                live[op % FM]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < FM; ++i) cur[i] = nxt[i];
    }
}

// ---- fp32 activations x split-bf16 weights on the bf16 MFMA (opt-in mode, DmGemmDesc.w_split) --------------------
// The weight image of this mode holds, per packed row and k-tile of 32 values, 32 bf16 "hi" terms followed by 32 bf16
// "lo" terms (w ~ hi + lo, lo = bf16(w - hi)): the same 128 bytes per row and k-tile as the fp32 image, so staging is
// unchanged.  The activations stay fp32 in memory and LDS; lane group q takes the eight k values 4q .. 4q+3 and
// 16+4q .. 16+4q+3 of the k-tile - chunks q and 4+q of its row, the conflict-free read pattern of the exact-fp32 loop
// (chunks 2q, 2q+1 cost 20 % more LDS cycles in bank conflicts) - and the weight terms are stored in that k order inside
// every k-tile.  It splits them the same way in registers and issues three 16x16x32 MFMAs per output fragment:
// a_hi.w_hi + a_lo.w_hi + a_hi.w_lo with fp32 accumulation (the lo.lo term, 2^-16 of a product, is dropped).
// One k-tile is then 12 bf16 MFMAs per 16 x 64 wave tile (192 cycles) instead of 32 exact-fp32 ones (1024 cycles).
template <int FM>
__device__ __forceinline__ void mma_tile_split(f32x4 (&acc)[FM][4], const char* sa, const char* sw, int arow0, int wrow0,
                                               int fr, int fq) {
    u32x4 a0[FM], a1[FM], wh[4], wl[4];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        a0[i] = *reinterpret_cast<const u32x4*>(sa + swz(arow0 + i * 16 + fr, fq));
        a1[i] = *reinterpret_cast<const u32x4*>(sa + swz(arow0 + i * 16 + fr, 4 + fq));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        wh[j] = *reinterpret_cast<const u32x4*>(sw + swz(wrow0 + j * 16 + fr, fq));
        wl[j] = *reinterpret_cast<const u32x4*>(sw + swz(wrow0 + j * 16 + fr, 4 + fq));
    }
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        float f[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            f[e] = __uint_as_float(a0[i][e]);
            f[4 + e] = __uint_as_float(a1[i][e]);
        }
        u32x4 ah, al;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t h = pack_bf16x2(f[2 * q], f[2 * q + 1]);
            ah[q] = h;
            al[q] = pack_bf16x2(f[2 * q] - __uint_as_float(h << 16), f[2 * q + 1] - __uint_as_float(h & 0xffff0000u));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            Mma16<bf16_t>::mma(acc[i][j], wh[j], ah);
            Mma16<bf16_t>::mma(acc[i][j], wh[j], al);
            Mma16<bf16_t>::mma(acc[i][j], wl[j], ah);
        }
    }
}

// Three bf16 terms per operand (x ~ hi + mid + lo: 24 bits) and the six products down to 2^-16 of the leading one:
// hi hi + hi mid + mid hi + hi lo + mid mid + lo hi; what is dropped (mid lo, lo mid, lo lo) is 2^-24 of a product, the
// size of an fp32 rounding.  The weight image of this form has TWO 128-byte rows per packed row and k-tile: plane 0 =
// 32 hi terms | 32 mid terms, plane 1 = 32 lo terms | 64 bytes of zeros; in LDS plane 1 follows plane 0 (BN rows each).
template <int FM, int BN>
__device__ __forceinline__ void mma_tile_split3(f32x4 (&acc)[FM][4], const char* sa, const char* sw, int arow0, int wrow0,
                                                int fr, int fq) {
    u32x4 a0[FM], a1[FM], wh[4], wm[4], wl[4];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        a0[i] = *reinterpret_cast<const u32x4*>(sa + swz(arow0 + i * 16 + fr, fq));
        a1[i] = *reinterpret_cast<const u32x4*>(sa + swz(arow0 + i * 16 + fr, 4 + fq));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        wh[j] = *reinterpret_cast<const u32x4*>(sw + swz(wrow0 + j * 16 + fr, fq));
        wm[j] = *reinterpret_cast<const u32x4*>(sw + swz(wrow0 + j * 16 + fr, 4 + fq));
        wl[j] = *reinterpret_cast<const u32x4*>(sw + BN * 128 + swz(wrow0 + j * 16 + fr, fq));
    }
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        float f[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            f[e] = __uint_as_float(a0[i][e]);
            f[4 + e] = __uint_as_float(a1[i][e]);
        }
        u32x4 ah, am, al;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t h = pack_bf16x2(f[2 * q], f[2 * q + 1]);
            const float r0 = f[2 * q] - __uint_as_float(h << 16), r1 = f[2 * q + 1] - __uint_as_float(h & 0xffff0000u);
            const uint32_t m = pack_bf16x2(r0, r1);
            ah[q] = h;
            am[q] = m;
            al[q] = pack_bf16x2(r0 - __uint_as_float(m << 16), r1 - __uint_as_float(m & 0xffff0000u));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // smallest terms first
            Mma16<bf16_t>::mma(acc[i][j], wl[j], ah);
            Mma16<bf16_t>::mma(acc[i][j], wh[j], al);
            Mma16<bf16_t>::mma(acc[i][j], wm[j], am);
            Mma16<bf16_t>::mma(acc[i][j], wm[j], ah);
            Mma16<bf16_t>::mma(acc[i][j], wh[j], am);
            Mma16<bf16_t>::mma(acc[i][j], wh[j], ah);
        }
    }
}

// mean and 1/sqrt(var + eps) of the rows this lane will finish, from the per-row (sum, sumsq) that the producer
// of A accumulated.  The difference of the two moments is taken in fp64 (cancellation), the rest in fp32.
template <int FM>
__device__ __forceinline__ void ln_row_stats(const GemmArgs& p, int mrow0, int ngroup0, int fr, int fq, float (&mu)[FM],
                                             float (&rs)[FM], f32x4 (&cs)[4]) {
    const float inv_k = 1.0f / (float)p.K;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        cs[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.ln_in) cs[j] = *reinterpret_cast<const f32x4*>(p.ln_colsum + ngroup0 + 16 * j + 4 * fq);
    }
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int m = mrow0 + i * 16 + fr;
        mu[i] = 0.f;
        rs[i] = 0.f;
        if (p.ln_in && m < p.M) {
            const double mean = p.ln_in[2L * m] * (double)inv_k;
            const float var = fmaxf((float)(p.ln_in[2L * m + 1] * (double)inv_k - mean * mean), 0.f);
            mu[i] = (float)mean;
            rs[i] = rsqrtf(var + p.ln_eps);
        }
    }
}

// The column-only epilogue terms of a lane (its 16 bias values; GEGLU: 8 value biases in b[0..1], their 8 gate biases
// in b[2..3]).  A caller that runs the epilogue in several passes fetches them once and hands them in.
__device__ __forceinline__ void load_bias(const GemmArgs& p, int ngroup0, int fq, f32x4 (&b)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.epi == DM_EPI_GEGLU) {
        const int half = p.N >> 1;
        const int nb = (ngroup0 >> 1) + fq * 8;
        if (nb >= half) return;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            b[j] = *reinterpret_cast<const f32x4*>(p.bias + nb + 4 * j);
            b[2 + j] = *reinterpret_cast<const f32x4*>(p.bias + half + nb + 4 * j);
        }
    } else if (p.bias) {
        const int ncol0 = ngroup0 + fq * 16;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (ncol0 + 4 * j < p.N) b[j] = *reinterpret_cast<const f32x4*>(p.bias + ncol0 + 4 * j);
    }
}

// Epilogue: lane holds, for row m = mrow0 + 16*i + fr, columns ncol0 + 4*j + r (16 consecutive).
// Column-only terms (bias) are fetched once per lane, outside the row loop; bf16 rows are written as
// 16-byte pieces.
template <typename T, int FM>
__device__ __forceinline__ void epilogue(const GemmArgs& p, f32x4 (&acc)[FM][4], int mrow0, int ngroup0, int fr, int fq,
                                         int kslice, const float (&ln_mu)[FM], const float (&ln_rs)[FM],
                                         const f32x4 (&ln_cs)[4], const f32x4* pre_bias = nullptr) {
    const int ncol0 = ngroup0 + fq * 16;
    if (p.splitk > 1) {
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int m = mrow0 + i * 16 + fr;
            if (m >= p.M) continue;
            float* dst = p.partial + ((long)kslice * p.M + m) * p.N;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = ncol0 + 4 * j;
                if (n < p.N) *reinterpret_cast<f32x4*>(dst + n) = acc[i][j];
            }
        }
        return;
    }
    if (p.ln_in) {
        // A was the raw input of a LayerNorm whose scale is folded into W and whose shift is folded into the
        // bias: LN(a) W^T = rstd * (a W'^T - mean * colsum(W')) + (beta W^T + b).  mean / rstd of this lane's
        // rows were fetched before the main loop (ln_row_stats), so no load latency is exposed here.
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = ln_rs[i] * (acc[i][j][r] - ln_mu[i] * ln_cs[j][r]);
    }
    if (p.epi == DM_EPI_GEGLU) {
        // lane holds 8 value columns nb..nb+7 (j = 0, 1) and their gates (j = 2, 3)
        const int half = p.N >> 1;
        const int nb = (ngroup0 >> 1) + fq * 8;
        if (nb >= half) return;  // N/2 is a multiple of 8 (checked on the host)
        f32x4 bv[2], bg[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            bv[j] = pre_bias ? pre_bias[j] : *reinterpret_cast<const f32x4*>(p.bias + nb + 4 * j);
            bg[j] = pre_bias ? pre_bias[2 + j] : *reinterpret_cast<const f32x4*>(p.bias + half + nb + 4 * j);
        }
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int m = mrow0 + i * 16 + fr;
            if (m >= p.M) continue;
            float o[8];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = acc[i][j][r] + bv[j][r];
                    const float g = acc[i][j + 2][r] + bg[j][r];
                    if constexpr (sizeof(T) == 4) o[4 * j + r] = v * gelu_erf_f(g);
                    else o[4 * j + r] = v * gelu_erf_fast_f(g);
                }
            T* dst = reinterpret_cast<T*>(p.C) + (long)m * p.ldc + nb;
            if constexpr (sizeof(T) == 4) {
                *reinterpret_cast<f32x4*>(dst) = f32x4{o[0], o[1], o[2], o[3]};
                *reinterpret_cast<f32x4*>(dst + 4) = f32x4{o[4], o[5], o[6], o[7]};
            } else {
                *reinterpret_cast<u32x4*>(dst) = u32x4{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]),
                                                       pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7])};
            }
        }
        return;
    }
    f32x4 b4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        b4[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (pre_bias) b4[j] = pre_bias[j];
        else if (p.bias && ncol0 + 4 * j < p.N) b4[j] = *reinterpret_cast<const f32x4*>(p.bias + ncol0 + 4 * j);
    }
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int m = mrow0 + i * 16 + fr;
        if (m >= p.M) continue;
        const int bidx = p.rowbias ? m / p.rows_per_batch : 0;
        float row_s = 0.f, row_q = 0.f;  // sums of this row's stored values (for a following LayerNorm)
        // two 8-column halves; N is a multiple of 4, so a half is all-valid, half-valid or absent
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int n = ncol0 + 8 * h;
            if (n >= p.N) continue;
            const bool full = n + 8 <= p.N;
            float o[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) o[q] = acc[i][2 * h + (q >> 2)][q & 3] + b4[2 * h + (q >> 2)][q & 3];
            if (p.rowbias) {
                const float* rb = p.rowbias + (long)bidx * p.ld_rowbias + n;
                const f32x4 r0 = *reinterpret_cast<const f32x4*>(rb);
                const f32x4 r1 = full ? *reinterpret_cast<const f32x4*>(rb + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    o[r] += r0[r];
                    o[4 + r] += r1[r];
                }
            }
            if (p.R) {
                const T* rp = reinterpret_cast<const T*>(p.R) + (long)m * p.ldr + n;
                if constexpr (sizeof(T) == 4) {
                    const f32x4 r0 = *reinterpret_cast<const f32x4*>(rp);
                    const f32x4 r1 = full ? *reinterpret_cast<const f32x4*>(rp + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        o[r] += r0[r];
                        o[4 + r] += r1[r];
                    }
                } else {
                    u32x4 rr = {0u, 0u, 0u, 0u};
                    if (full) {
                        rr = *reinterpret_cast<const u32x4*>(rp);
                    } else {
                        const u32x2 r2 = *reinterpret_cast<const u32x2*>(rp);
                        rr[0] = r2[0];
                        rr[1] = r2[1];
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        o[2 * q] += __uint_as_float(rr[q] << 16);
                        o[2 * q + 1] += __uint_as_float(rr[q] & 0xffff0000u);
                    }
                }
            }
            if (p.epi == DM_EPI_QUICKGELU) {
                // x * sigmoid(1.702 x)  (src/models/clip.py:114)
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    if constexpr (sizeof(T) == 4) o[q] = o[q] * (1.0f / (1.0f + __expf(-1.702f * o[q])));
                    else o[q] = o[q] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.4554670f * o[q]));
                }
            }
            if (p.ln_out) {
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (q < 4 || full) {
                        float v = o[q];
                        if constexpr (sizeof(T) == 2) v = bf16_to_f32(f32_to_bf16(v));
                        row_s += v;
                        row_q = fmaf(v, v, row_q);
                    }
            }
            if (p.stats.sums) {
                // keep the value as stored (after rounding) in the accumulator for the statistics pass below
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    float v = o[q];
                    if constexpr (sizeof(T) == 2) v = bf16_to_f32(f32_to_bf16(v));
                    acc[i][2 * h + (q >> 2)][q & 3] = v;
                }
            }
            if (p.epi == DM_EPI_NCHW_F32) {
                const int hw = p.rows_per_batch;
                const int b = m / hw, pix = m - b * hw;
                float* dst = reinterpret_cast<float*>(p.C);
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (q < 4 || full) dst[((long)b * p.N + n + q) * hw + pix] = o[q];
            } else {
                T* dst = reinterpret_cast<T*>(p.C) + (long)m * p.ldc + n;
                if constexpr (sizeof(T) == 4) {
                    *reinterpret_cast<f32x4*>(dst) = f32x4{o[0], o[1], o[2], o[3]};
                    if (full) *reinterpret_cast<f32x4*>(dst + 4) = f32x4{o[4], o[5], o[6], o[7]};
                } else if (full) {
                    *reinterpret_cast<u32x4*>(dst) = u32x4{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]),
                                                           pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7])};
                } else {
                    *reinterpret_cast<u32x2*>(dst) = u32x2{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
                }
            }
        }
        if (p.ln_out) {
            // the four lanes that share this row (one per 16-column group) add up, then one of them publishes
            row_s += __shfl_xor(row_s, 16);
            row_q += __shfl_xor(row_q, 16);
            row_s += __shfl_xor(row_s, 32);
            row_q += __shfl_xor(row_q, 32);
            if (fq == 0) {
                atomicAdd(p.ln_out + 2L * m, (double)row_s);
                atomicAdd(p.ln_out + 2L * m + 1, (double)row_q);
            }
        }
    }
    if (p.stats.sums && mrow0 < p.M) {
        // GroupNorm statistics of the tensor just written: per-channel sum and sum of squares over this
        // wave's rows, which all belong to one sample and are all valid (checked on the host).  Four columns
        // at a time (few live registers): reduce-scatter over lane bits 0-1, all-reduce over bits 2-3, then
        // the lanes with fr>>2 == j own column 4*j + (fr&3) = fr, i.e. lane l owns column ngroup0 + l; those are
        // then folded to groups and added to memory by group_stats_add.
        float own_s = 0.f, own_q = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v[2][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[0][r] = v[1][r] = 0.f;
#pragma unroll
            for (int i = 0; i < FM; ++i)
                if (mrow0 + i * 16 < p.M) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[0][r] += acc[i][j][r];
                        v[1][r] = fmaf(acc[i][j][r], acc[i][j][r], v[1][r]);
                    }
                }
#pragma unroll
            for (int w = 0; w < 2; ++w) {
#pragma unroll
                for (int half = 2; half >= 1; half >>= 1) {
                    const bool upper = (fr & half) != 0;
#pragma unroll
                    for (int q = 0; q < half; ++q) {
                        const float keep = upper ? v[w][q + half] : v[w][q];
                        const float send = upper ? v[w][q] : v[w][q + half];
                        v[w][q] = keep + __shfl_xor(send, half);
                    }
                }
                v[w][0] += __shfl_xor(v[w][0], 4);
                v[w][0] += __shfl_xor(v[w][0], 8);
            }
            if ((fr >> 2) == j) {
                own_s = v[0][0];
                own_q = v[1][0];
            }
        }
        if (ncol0 + fr >= p.N) own_s = own_q = 0.f;  // padding columns of the tile
        const long b = mrow0 / p.rows_per_batch;
        const int lane = fq * 16 + fr;
        group_stats_add(p.stats, b, ngroup0, lane, own_s, own_q);
        if (p.stats2.sums) group_stats_add(p.stats2, b, ngroup0, lane, own_s, own_q);
    }
}

// Wait until at most min(young, MAXY) k-tiles of LOADS LDS-DMA instructions each are still outstanding
// (loads retire in order, so everything older has landed).  The count must be an immediate: one compare per
// candidate, the steady state (young >= MAXY) taking the first.
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int LOADS, int MAXY> __device__ __forceinline__ void wait_young(int young) {
    if constexpr (MAXY == 0) {
        wait_vm<0>();
    } else {
        if (young >= MAXY) wait_vm<MAXY * LOADS>();
        else wait_young<LOADS, MAXY - 1>(young);
    }
}

// ------------------------------------------------------------------------------------------------
// direct-to-LDS main loop
// ------------------------------------------------------------------------------------------------
template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, int STAGES, bool CONV, bool EARLY = false, bool GN = false,
          int SPLIT = 0, int KU = 1, bool KORD = false, bool HALO = false>
__device__ __forceinline__ void gemm_dma_body(const GemmArgs& p) {
    static_assert(!KORD || (CONV && sizeof(T) == 2 && !GN && SPLIT == 0), "chunk-major k axis: bf16 convolutions");
    // HALO: the activation of a 64-channel chunk sits in LDS once, as the tile's pixels plus a one-pixel border (two
    // half-chunks of 32 channels, 96-byte pixel pitch: 16 consecutive pixels are then read by ds_read_b128 at full rate
    // from any starting pixel, which 64-, 80- and 128-byte pitches are not), and the nine taps read it at nine offsets; only the weights are
    // requested per k-tile.  k axis of the weights: per chunk the 18 (half, tap) units in the order half-major, two units
    // per k-tile, so that half 0 is dead after k-tile 4 of a chunk and is refilled with the next chunk's while half 1 is read.
    static_assert(!HALO || (KORD && BM == 128 && BN == 128 && WAVES_M == 2 && WAVES_N == 2 && STAGES == 2 && KU == 1 && !EARLY),
                  "activation halo in LDS: 128 x 128 tile, four waves, two weight stages");
    // KU = 2: two k-tiles per barrier.  The four LDS stages form two halves of two k-tiles each; a block waits for the pair
    // it is about to multiply (the only requests outstanding), passes ONE barrier, requests the next pair into the other half
    // and issues the MFMAs of both tiles: half the waits, barriers and request bursts per MFMA, a whole pair of lead time.
    static_assert(KU == 1 || (KU == 2 && STAGES == 4 && !EARLY && !GN && SPLIT == 0), "pairs of k-tiles: plain ring loop, four stages");
    // SPLIT: 0 = the operands' own arithmetic, 2 / 3 = fp32 problems with two / three bf16 terms per operand
    static_assert(SPLIT == 0 || ((SPLIT == 2 || SPLIT == 3) && sizeof(T) == 4 && !EARLY && !GN),
                  "split-bf16 weights go with the plain fp32 ring loop");
    constexpr int WPL = SPLIT == 3 ? 2 : 1;  // 128-byte rows per packed weight row and k-tile
    static_assert(!EARLY || STAGES == 2, "the early-refill loop runs on two LDS buffers");
    static_assert(!GN || (sizeof(T) == 4 && !EARLY), "fragment-side GroupNorm exists for the fp32 ring loop only");
    constexpr int NT = WAVES_M * WAVES_N * 64;
    constexpr int FM = BM / (16 * WAVES_M);
    static_assert(BN == 64 * WAVES_N, "each wave owns 64 output columns");
    static_assert(BM % (16 * WAVES_M) == 0 && (BM * 8) % NT == 0 && (BN * 8) % NT == 0, "bad tile");
    static_assert((NT / 8) % 16 == 0, "per-thread swizzle must not depend on the staging iteration");
    constexpr int EPC = ElemTraits<T>::EPC;
    constexpr int BK = 8 * EPC;
    constexpr int A_IT = BM * 8 / NT;
    constexpr int W_IT = WPL * BN * 8 / NT;
    constexpr int LOADS = A_IT + W_IT;           // LDS-DMA instructions per thread per k-tile
    constexpr int STAGE_BYTES = (BM + WPL * BN) * 128;
    static_assert(STAGES >= 2 && STAGES <= 8 && (STAGES - 2) * LOADS <= 63, "ring depth / vmcnt field");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: LDS destinations become SALU arithmetic
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
    const int nkt = p.Kpad / BK;
    const int fr = lane & 15, fq = lane >> 4;
    // weight-heavy shapes (N > M) walk M fastest so the blocks sharing a W tile sit in one XCD's L2;
    // activation-heavy shapes walk N fastest so the A tile is the shared one
    const bool m_fast = (p.conv ? 9 * p.N : p.N) > p.M;  // unique W bytes vs unique A bytes
    const char* zsrc = reinterpret_cast<const char*>(g_zero16);
    const int lds_wave = wave * 1024;  // wave-uniform part of every destination
    // This thread always lands on LDS slot (row = it*NT/8 + tid/8, physical chunk tid&7); the logical
    // chunk stored there is phys ^ ((row>>1)&7) = (tid&7) ^ ((tid>>4)&7), the same for all its rows.
    const int lchunk = (tid & 7) ^ ((tid >> 4) & 7);

    // Work list.  One tile per block: tile = blockIdx.x, k range = slice blockIdx.y of splitk.
    // Stream-K: the (tile, k-tile) pairs, tile-major, are cut into gridDim.x equal runs; a block walks
    // its run segment by segment (at most one tile entered late and one left early).
    long u = 0, u_end = 1;
    int sk_g = 0;
    [[maybe_unused]] float* gn_tab = reinterpret_cast<float*>(smem + STAGES * STAGE_BYTES);
    [[maybe_unused]] int gn_b = -1;  // sample whose scale / shift table sits in LDS
    [[maybe_unused]] int tl_seg = -1;
    [[maybe_unused]] unsigned long long tl_cycles = 0;  // shader-clock count at the start of the k loop (diagnostic build)
#ifdef DM_TIMELINE
    const int tl_variant = __builtin_amdgcn_readfirstlane(g_tl_variant);
#endif
    if (p.streamk) {
        sk_g = xcd_remap(blockIdx.x, gridDim.x);
        const long U = (long)ntm * ntn * nkt;
        u = sk_g * U / gridDim.x;
        u_end = (sk_g + 1) * U / gridDim.x;
    }

    while (u < u_end) {
        int tile_id, kt0, kt1;
        if (p.streamk) {
            tile_id = (int)(u / nkt);
            kt0 = (int)(u - (long)tile_id * nkt);
            kt1 = (int)min((long)nkt, kt0 + (u_end - u));
            u += kt1 - kt0;
            // the previous segment's LDS reads (and ticket word) are done before the ring is refilled
            __syncthreads();
        } else {
            tile_id = xcd_remap(blockIdx.x, ntm * ntn);
            kt0 = 0;
            kt1 = nkt;
            if (p.splitk > 1) {
                const int per = (nkt + p.splitk - 1) / p.splitk;
                kt0 = blockIdx.y * per;
                kt1 = min(nkt, kt0 + per);
            }
            u = u_end;
        }
        const int tile_m = m_fast ? tile_id % ntm : tile_id / ntn;
        const int tile_n = m_fast ? tile_id / ntm : tile_id - tile_m * ntn;
        const int m0 = tile_m * BM, n0 = tile_n * BN;
        ++tl_seg;
        DM_TL(tl_seg, 0);  // segment entered

        // fragment-side GroupNorm: per fragment row of this lane, which taps read real data
        [[maybe_unused]] uint32_t gn_vmask[FM];
        if constexpr (GN) {
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int m = m0 + wm * 16 * FM + i * 16 + fr;
                uint32_t bits = 0;
                if (m < p.M) {
                    if (CONV) {
                        const int pix = m % (p.Hout * p.Wout);
                        const int y = pix / p.Wout, x = pix - y * p.Wout;
#pragma unroll
                        for (int t9 = 0; t9 < 9; ++t9) {
                            const int iy = y + t9 / 3 - 1, ix = x + t9 % 3 - 1;
                            if ((unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win) bits |= 1u << t9;
                        }
                    } else {
                        bits = 1u;
                    }
                }
                gn_vmask[i] = bits;
            }
        }
        ARows<A_IT> rows;
        init_rows<A_IT, NT / 8>(rows, p, m0, tid >> 3, BM);
        // Chunk-major convolution (stride 1, same size): the input pixel of (output row m, tap) is row m + dy*Win + dx of
        // the image matrix, so a request is a wave-uniform base plus this thread's fixed row offset; what differs per row
        // is whether the tap falls outside the image (bit t of a_mask; bit 9 = the appended pointwise operand).
        constexpr bool kord = KORD;  // chunk-major k axis (DmGemmDesc.conv_chunk_major): its own kernels, no run-time test
        [[maybe_unused]] uint32_t a_mask[A_IT];
        if (CONV) {
#pragma unroll
            for (int it = 0; it < A_IT; ++it) {
                uint32_t bits = 0;
                if (kord && rows.ok[it]) {
#pragma unroll
                    for (int t9 = 0; t9 < 9; ++t9) {
                        const int iy = rows.y[it] + t9 / 3 - 1, ix = rows.x[it] + t9 % 3 - 1;
                        if ((unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win) bits |= 1u << t9;
                    }
                    if (p.A2) bits |= 1u << 9;
                }
                a_mask[it] = bits;
            }
        }
        // Running source pointers, advanced after every k-tile.  A row (or, in a convolution, a tap of a
        // row) that must read as zero points at the 16-byte zero buffer with step 0, so the steady state
        // costs one 64-bit add per DMA and no select.
        const char* w_ptr[W_IT];
#pragma unroll
        for (int it = 0; it < W_IT; ++it) {
            // staged row vr of the W tile: plane vr / BN (one plane unless the image has three terms), weight row vr % BN
            const int vr = it * (NT / 8) + (tid >> 3);
            const int plane = WPL > 1 ? vr / BN : 0, wrow = n0 + vr - plane * BN;
            w_ptr[it] = p.W + ((((long)(wrow >> 6) * nkt + kt0) * WPL + plane) * 64 + (wrow & 63)) * 128 + lchunk * 16;
        }
        // ---- HALO: where this thread's pieces of a half-chunk image come from, and where its fragment rows sit in it
        constexpr int HALO_BYTES = 20480;   // 204 pixel slots of 96 bytes and the tail of the last wave's request
        constexpr int HALO_BASE = 2 * 16384;  // behind the two weight stages
        [[maybe_unused]] uint32_t hoff[6];  // byte offset from p.A of the piece's 16 bytes at channel 0; ~0 = reads as zero
        [[maybe_unused]] uint32_t lc[FM];   // byte offset of fragment row i of this lane inside a half-chunk image (centre tap)
        [[maybe_unused]] int h_pitch = 0, h_pieces = 0;
        if constexpr (HALO) {
            const int Wd = p.Win, HW = p.Hin * Wd;
            const int R = min(BM / Wd, p.Hin), RW = R * Wd;  // image rows and pixels of one group of the tile
            h_pitch = Wd + 2;
            const int SG = (R + 2) * h_pitch;                // slots of one group
            h_pieces = (BM / RW) * SG * 6;                   // 16-byte pieces of a half-chunk image (a slot's last two are padding)
#pragma unroll
            for (int it = 0; it < 6; ++it) {
                const int piece = it * NT + tid;
                const int slot = piece / 6, ch = piece - slot * 6;
                const int g = slot / SG, rs = slot - g * SG;
                const int ry = rs / h_pitch, rx = rs - ry * h_pitch;
                const int p0 = m0 + g * RW, b = p0 / HW;
                const int y = (p0 - b * HW) / Wd + ry - 1, x = rx - 1;
                const bool ok = piece < h_pieces && ch < 4 && (unsigned)x < (unsigned)Wd && (unsigned)y < (unsigned)p.Hin;
                hoff[it] = ok ? (uint32_t)((((long)b * HW + (long)y * Wd + x) * p.lda + ch * 8) * 2) : 0xffffffffu;
            }
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int px = wm * 16 * FM + i * 16 + fr;
                const int g = px / RW, q = px - g * RW;
                const int y = q / Wd, x = q - y * Wd;
                lc[i] = (uint32_t)((g * SG + (y + 1) * h_pitch + x + 1) * 96 + fq * 16);
            }
        }
        int kc = kt0 * BK + lchunk * EPC, tap = 0;
        if (CONV) {
            tap = min(kc / p.Cin, 9);  // tap 9: the appended pointwise operand (only with A2)
            kc -= tap * p.Cin;
        }
        // Fast paths: convolution with Cin a multiple of the k-tile (the tap is the same for every thread of
        // a k-tile and changes only every Cin/BK tiles); dense with no padded k tail.
        const bool fast = CONV ? (p.Cin % BK == 0) : (p.Kpad == p.K);
        const char* a_ptr[A_IT];
        int a_step[A_IT];
        int c0 = 0;  // channel offset of the next k-tile inside its tap
        auto set_tap = [&]() {
            if (tap >= 9) {
                // appended pointwise operand: the output pixel's own row of the second tensor (the launcher
                // admits it only for stride-1, same-size convolutions, where output and input pixels coincide)
#pragma unroll
                for (int it = 0; it < A_IT; ++it) {
                    const bool ok = rows.ok[it] && p.A2 != nullptr;
                    const long pix = rows.off[it] + (long)rows.y[it] * p.Win + rows.x[it];
                    a_ptr[it] = ok ? p.A2 + (pix * p.lda2 + c0 + lchunk * EPC) * sizeof(T) : zsrc;
                    a_step[it] = ok ? BK * (int)sizeof(T) : 0;
                }
                return;
            }
#pragma unroll
            for (int it = 0; it < A_IT; ++it) {
                long e;
                const bool ok = conv_src<A_IT>(rows, p, it, tap, lchunk * EPC, e);
                a_ptr[it] = ok ? p.A + (e + c0) * sizeof(T) : zsrc;
                a_step[it] = ok ? BK * (int)sizeof(T) : 0;
            }
        };
        // dense with a second operand: from k-tile kt_second on, A's columns come from A2 (same rows)
        const int kt_second = (!CONV && p.A2) ? (p.K - p.Cin2) / BK : 0x7fffffff;
        auto second_operand = [&](int kt) {
#pragma unroll
            for (int it = 0; it < A_IT; ++it)
                if (rows.ok[it])
                    a_ptr[it] = p.A2 + ((long)rows.y[it] * p.lda2 + (long)(kt - kt_second) * BK + lchunk * EPC) * sizeof(T);
        };
        if (fast) {
            if (CONV) {
                if (kord) {
                    chunk_major_at(kt0, p.Cin, BK, tap, c0);
                } else {
                    c0 = kc - lchunk * EPC;
                    set_tap();
                }
            } else {
#pragma unroll
                for (int it = 0; it < A_IT; ++it) {
                    a_ptr[it] = rows.ok[it] ? p.A + (rows.off[it] + (long)kt0 * BK + lchunk * EPC) * sizeof(T) : zsrc;
                    a_step[it] = rows.ok[it] ? BK * (int)sizeof(T) : 0;
                }
                if (kt0 >= kt_second) second_operand(kt0);
            }
        }

        // one k-tile of DMAs; called for kt < kt1 only
        auto issue = [&](int kt, int stage) {
            char* sa = smem + stage * STAGE_BYTES + lds_wave;
            char* sw = sa + BM * 128;
            if (CONV && kord) {
                const bool second = tap >= 9;
                const long ld = second ? p.lda2 : p.lda;
                const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
                const char* base = (second ? p.A2 + ((long)m0 * ld + c0) * sizeof(T)
                                           : p.A + ((long)(m0 + dy * p.Win + dx) * ld + c0) * sizeof(T)) +
                                   ((long)(tid >> 3) * ld + lchunk * EPC) * sizeof(T);
                const long row_step = (long)(NT / 8) * ld * sizeof(T);
#pragma unroll
                for (int it = 0; it < A_IT; ++it) {
                    const bool good = ((a_mask[it] >> tap) & 1u) && (DM_TL_ON(16) || tap == 0);
                    const char* src = good ? base + it * row_step : zsrc;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)(sa + it * (NT * 16)), 16, 0, 0);
                }
                chunk_major_next(p.Cin, BK, tap, c0);
            } else if (fast) {
                if (!CONV && kt == kt_second) second_operand(kt);  // wave-uniform, once per run
#pragma unroll
                for (int it = 0; it < A_IT; ++it) {
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)a_ptr[it],
                                                     (__attribute__((address_space(3))) void*)(sa + it * (NT * 16)), 16, 0, 0);
                    a_ptr[it] += a_step[it];
                }
                if (CONV) {
                    c0 += BK;
                    if (c0 == p.Cin && tap < 9) {
                        c0 = 0;
                        ++tap;
                        set_tap();
                    }
                }
            } else if (CONV) {
#pragma unroll
                for (int it = 0; it < A_IT; ++it) {
                    long e;
                    const bool ok = conv_src<A_IT>(rows, p, it, tap, kc, e);
                    const char* src = ok ? p.A + e * sizeof(T) : zsrc;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)(sa + it * (NT * 16)), 16, 0, 0);
                }
                kc += BK;
                while (kc >= p.Cin) {
                    kc -= p.Cin;
                    ++tap;
                }
            } else {
                const int k = kt * BK + lchunk * EPC;
#pragma unroll
                for (int it = 0; it < A_IT; ++it) {
                    const bool ok = rows.ok[it] && k < p.K;
                    const char* src = ok ? p.A + (rows.off[it] + k) * sizeof(T) : zsrc;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)(sa + it * (NT * 16)), 16, 0, 0);
                }
            }
#pragma unroll
            for (int it = 0; it < W_IT; ++it) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(DM_TL_ON(32) ? w_ptr[it] : zsrc),
                                                 (__attribute__((address_space(3))) void*)(sw + it * (NT * 16)), 16, 0, 0);
                w_ptr[it] += WPL * W_KTILE_STEP;
            }
        };

        f32x4 acc[FM][4];
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        float ln_mu[FM], ln_rs[FM];
        f32x4 ln_cs[4];
        ln_row_stats<FM>(p, m0 + wm * 16 * FM, n0 + wn * 64, fr, fq, ln_mu, ln_rs, ln_cs);

        if constexpr (HALO) {
            const int nchunk = p.Cin / BK, nine = 9 * nchunk;
            // one half-chunk image: at most six requests per thread, whole waves past the image's end request nothing
            auto issue_halo = [&](int h, int c) {
                const char* base = p.A + ((long)c * BK + h * 32) * sizeof(T);
                char* dst = smem + HALO_BASE + h * HALO_BYTES + lds_wave;
#pragma unroll
                for (int it = 0; it < 6; ++it) {
                    if (it * NT + wave * 64 >= h_pieces) continue;  // wave-uniform
                    const char* src = hoff[it] != 0xffffffffu ? base + hoff[it] : zsrc;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)(dst + it * (NT * 16)), 16, 0, 0);
                }
            };
            // the weights of k-tile kt into weight stage ws; behind the chunks, the k-tile of the pointwise operand as a plain
            // 128-row A tile into the (then idle) image area of the same parity
            auto issue_w = [&](int kt, int ws) {
                if (kt >= nine) {
                    const char* base = p.A2 + ((long)m0 * p.lda2 + (long)(kt - nine) * BK) * sizeof(T) +
                                       ((long)(tid >> 3) * p.lda2 + lchunk * EPC) * sizeof(T);
                    const long row_step = (long)(NT / 8) * p.lda2 * sizeof(T);
                    char* sa = smem + HALO_BASE + ((kt - nine) & 1) * HALO_BYTES + lds_wave;
#pragma unroll
                    for (int it = 0; it < A_IT; ++it) {
                        const char* src = ((a_mask[it] >> 9) & 1u) ? base + it * row_step : zsrc;
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                         (__attribute__((address_space(3))) void*)(sa + it * (NT * 16)), 16, 0, 0);
                    }
                }
                char* sw = smem + ws * 16384 + lds_wave;
#pragma unroll
                for (int it = 0; it < W_IT; ++it) {
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)w_ptr[it],
                                                     (__attribute__((address_space(3))) void*)(sw + it * (NT * 16)), 16, 0, 0);
                    w_ptr[it] += W_KTILE_STEP;
                }
            };
            int c = 0, j = 0;  // chunk and k-tile inside it of the k-tile about to be consumed
            if (kt0 < nine) {
                c = kt0 / 9;
                j = kt0 - 9 * c;
                if (j <= 4) issue_halo(0, c);
                else if (j > 5 && c + 1 < nchunk) issue_halo(0, c + 1);
                issue_halo(1, c);
            }
            if (kt0 < kt1) issue_w(kt0, 0);
            int stage = 0;
            for (int kt = kt0; kt < kt1; ++kt) {
                wait_vm<0>();
                __builtin_amdgcn_s_barrier();
                if (kt + 1 < kt1) issue_w(kt + 1, stage ^ 1);
                const char* sw = smem + stage * 16384;
                if (kt < nine) {
                    // half 1 was last read by k-tile 8 of the previous chunk, half 0 by k-tile 4 of this one
                    if (j == 0 && kt != kt0) issue_halo(1, c);
                    if (j == 5 && c + 1 < nchunk && (c + 1) * 9 < kt1) issue_halo(0, c + 1);
                    u32x4 af[2][FM], wf[2][4];
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        const int u = 2 * j + ks, h = u >= 9 ? 1 : 0, t = u - 9 * h;
                        const int dy = t / 3, dx = t - 3 * dy;
                        const char* hb = smem + HALO_BASE + h * HALO_BYTES + ((dy - 1) * h_pitch + dx - 1) * 96;
#pragma unroll
                        for (int jn = 0; jn < 4; ++jn)
                            wf[ks][jn] = *reinterpret_cast<const u32x4*>(sw + swz(wn * 64 + jn * 16 + fr, ks * 4 + fq));
#pragma unroll
                        for (int i = 0; i < FM; ++i) af[ks][i] = *reinterpret_cast<const u32x4*>(hb + lc[i]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    mfma_tile_frags<T, FM>(acc, af, wf);
                    if (++j == 9) {
                        j = 0;
                        ++c;
                    }
                } else {
                    const char* sa = smem + HALO_BASE + ((kt - nine) & 1) * HALO_BYTES;
                    mma_tile<T, FM>(acc, sa, sw, wm * 16 * FM, wn * 64, fr, fq);
                }
                stage ^= 1;
            }
        } else if constexpr (KU == 2) {
            if (kt0 < kt1) issue(kt0, 0);
            if (kt0 + 1 < kt1) issue(kt0 + 1, 1);
            int stage = 0;  // first stage of the half that holds the current pair: 0 or 2
            for (int kt = kt0; kt < kt1; kt += 2) {
                wait_vm<0>();                  // the current pair has landed: nothing else was requested
                __builtin_amdgcn_s_barrier();  // ... every wave's share is visible, and all reads of the other half are done
                const int ns = stage ^ 2;
                if (kt + 2 < kt1) issue(kt + 2, ns);
                if (kt + 3 < kt1) issue(kt + 3, ns + 1);
                const char* sa = smem + stage * STAGE_BYTES;
                mma_tile<T, FM>(acc, sa, sa + BM * 128, wm * 16 * FM, wn * 64, fr, fq);
                if (kt + 1 < kt1) {
                    sa += STAGE_BYTES;
                    mma_tile<T, FM>(acc, sa, sa + BM * 128, wm * 16 * FM, wn * 64, fr, fq);
                }
                stage = ns;
            }
        } else if constexpr (EARLY) {
            // Two LDS buffers, two k-tiles in flight: a tile's fragments are all read into registers first, a
            // second barrier then frees its buffer, and the tile two steps ahead is requested into it before
            // the MFMAs start.  Every request so has almost two iterations to land instead of one.
            if (kt0 < kt1) issue(kt0, 0);
            if (kt0 + 1 < kt1) issue(kt0 + 1, 1);
            int stage = 0;
            for (int kt = kt0; kt < kt1; ++kt) {
                wait_young<LOADS, 1>(kt1 - 1 - kt);  // tile kt has landed; tile kt+1 may still be in flight
                __builtin_amdgcn_s_barrier();        // ... and every wave's share of it is visible
                const char* sa = smem + stage * STAGE_BYTES;
                u32x4 af[2][FM], wf[2][4];
                read_tile_frags<T, FM>(af, wf, sa, sa + BM * 128, wm * 16 * FM, wn * 64, fr, fq);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();        // all waves hold tile kt in registers: its buffer is free
                if (kt + 2 < kt1) issue(kt + 2, stage);
                __builtin_amdgcn_sched_barrier(0);
                mfma_tile_frags<T, FM>(acc, af, wf);
                stage ^= 1;
            }
        } else {
        // prologue: up to STAGES-1 tiles in flight
        DM_TL(tl_seg, 1);  // pointers set up, first request about to go out
#pragma unroll
        for (int s = 0; s < STAGES - 1; ++s)
            if (kt0 + s < kt1) issue(kt0 + s, s);
        if constexpr (GN) {
            // The sample's scale / shift table, behind the ring, built while the first k-tiles are in flight; rebuilt when
            // a work-list segment enters another sample (all waves passed the barrier at the top of the segment, so
            // nobody still reads the old one).
            const int b = m0 / p.rows_per_batch;
            if (b != gn_b) {
                gn_build_table<NT>(p, b, gn_tab);
                gn_b = b;
            }
        }

        // channel offset and tap of the k-tile about to be CONSUMED (the issue side runs STAGES-1 tiles ahead)
        [[maybe_unused]] int gn_tap = 0, gn_c0 = 0;
        if constexpr (GN) {
            if (CONV) {
                gn_tap = min(kt0 * BK / p.Cin, 9);
                gn_c0 = kt0 * BK - gn_tap * p.Cin;
            } else {
                gn_c0 = kt0 * BK;
            }
        }
        int stage = 0;
        for (int kt = kt0; kt < kt1; ++kt) {
            // fragment-side GroupNorm: this lane's scale / shift chunks of the k-tile, requested ahead of the wait
            [[maybe_unused]] f32x4 gn_s4[2], gn_t4[2];
            [[maybe_unused]] bool gn_raw = false;
            if constexpr (GN) {
                gn_raw = CONV ? gn_tap >= 9 : kt >= kt_second;
                const float* ts = gn_tab + (gn_raw ? 0 : gn_c0);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    gn_s4[ks] = *reinterpret_cast<const f32x4*>(ts + (ks * 4 + fq) * 4);
                    gn_t4[ks] = *reinterpret_cast<const f32x4*>(ts + p.gn_c + (ks * 4 + fq) * 4);
                }
            }
            // tile kt has landed once only the younger tiles in flight (at most STAGES-2, fewer at the end
            // of the run) are outstanding; the barrier makes every wave's share visible and retires all
            // reads of the stage refilled below
            if (DM_TL_ON(2) || kt == kt0) {
                wait_young<LOADS, STAGES - 2>(kt1 - 1 - kt);
                __builtin_amdgcn_s_barrier();
            }
            if (kt == kt0) {
                DM_TL(tl_seg, 2);  // first k-tile landed
                DM_TL_CYCLES_BEGIN();
            }
            int nstage = stage + STAGES - 1;
            if (nstage >= STAGES) nstage -= STAGES;
            const char* sa = smem + stage * STAGE_BYTES;
            if (DM_TL_ON(1) && kt + STAGES - 1 < kt1) issue(kt + STAGES - 1, nstage);
            if constexpr (GN) {
                mma_tile_gn<FM, CONV>(acc, sa, sa + BM * 128, wm * 16 * FM, wn * 64, fr, fq, gn_s4, gn_t4, gn_vmask,
                                      CONV ? gn_tap : 0, gn_raw);
                gn_c0 += BK;
                if (CONV && gn_c0 == p.Cin && gn_tap < 9) {
                    gn_c0 = 0;
                    ++gn_tap;
                }
            } else if constexpr (SPLIT == 3) {
                mma_tile_split3<FM, BN>(acc, sa, sa + BM * 128, wm * 16 * FM, wn * 64, fr, fq);
            } else if constexpr (SPLIT == 2) {
                mma_tile_split<FM>(acc, sa, sa + BM * 128, wm * 16 * FM, wn * 64, fr, fq);
            } else {
                mma_tile<T, FM>(acc, sa, sa + BM * 128, wm * 16 * FM, wn * 64, fr, fq);
            }
            if (++stage == STAGES) stage = 0;
        }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        DM_TL(tl_seg, 3);  // k loop done
        DM_TL_CYCLES_END(tl_seg);

        if (p.streamk && !(kt0 == 0 && kt1 == nkt)) {
            // Partial tile.  Every block that holds a k-run of this tile stores its accumulators, in
            // register order, write-through (sc1: the bytes are at memory scope once vmcnt drains, no
            // release fence needed) and then draws a ticket; the block that draws the last one sums the
            // slabs in block order (so the result does not depend on arrival order) and runs the epilogue.
            const long ut0 = (long)tile_id * nkt;
            const long U = (long)ntm * ntn * nkt;
            const int G = gridDim.x;
            const int g_first = (int)(((ut0 + 1) * G + U - 1) / U) - 1;    // block whose run holds unit ut0
            const int g_last = (int)(((ut0 + nkt) * G + U - 1) / U) - 1;  // ... and the tile's last unit
            constexpr long SLAB = (long)BM * BN;                           // floats
            {
                float* slab = p.slabs + (2L * sk_g + (kt0 == 0 ? 1 : 0)) * SLAB;
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(slab, 0, (int)(SLAB * 4), 0x00020000);
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), rs,
                                                               ((i * 4 + j) * NT + tid) * 16, 0, 16);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            int* ticket = reinterpret_cast<int*>(smem);
            if (tid == 0)
                *ticket = __hip_atomic_fetch_add(p.counters + tile_id, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            const bool last = *ticket == g_last - g_first;
            DM_TL(tl_seg, 4);  // slab stored, ticket drawn
            if (!last) continue;
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(p.counters + tile_id, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
            // A deep split leaves a dozen slabs per tile; read one after the other, each costs an exposed memory latency
            // and the sum lasts as long as the k loop.  The LDS ring is idle now: as many slabs as fit are requested into
            // it at once by the DMA path (no registers), then added in block order.  Every thread reads back exactly
            // the 16-byte pieces its own DMA lanes wrote, so draining vmcnt is the only synchronisation.
            constexpr int SLAB_BYTES = BM * BN * 4;
            constexpr int SLABS_IN_LDS = STAGES * STAGE_BYTES / SLAB_BYTES;
            if constexpr (SLABS_IN_LDS >= 2) {
                for (int c0s = g_first; c0s <= g_last; c0s += SLABS_IN_LDS) {
#pragma unroll
                    for (int u = 0; u < SLABS_IN_LDS; ++u) {
                        const int c = c0s + u;
                        if (c <= g_last) {  // block-uniform
                            const char* src = reinterpret_cast<const char*>(p.slabs + (2L * c + (c == g_first ? 1 : 0)) * SLAB) + tid * 16;
#pragma unroll
                            for (int e = 0; e < FM * 4; ++e)
                                __builtin_amdgcn_global_load_lds(
                                    (const __attribute__((address_space(1))) void*)(src + e * (NT * 16)),
                                    (__attribute__((address_space(3))) void*)(smem + u * SLAB_BYTES + e * (NT * 16) + lds_wave), 16, 0, 0);
                        }
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                    for (int u = 0; u < SLABS_IN_LDS; ++u) {
                        const int c = c0s + u;
                        if (c <= g_last) {
#pragma unroll
                            for (int i = 0; i < FM; ++i)
#pragma unroll
                                for (int j = 0; j < 4; ++j) {
                                    const f32x4 v = *reinterpret_cast<const f32x4*>(smem + u * SLAB_BYTES + ((i * 4 + j) * NT + tid) * 16);
                                    acc[i][j] = c == g_first ? v : acc[i][j] + v;
                                }
                        }
                    }
                    // the reads above are complete (their values were added) before the next round's DMA is issued
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
            } else {
                for (int c = g_first; c <= g_last; ++c) {
                    const float* slab = p.slabs + (2L * c + (c == g_first ? 1 : 0)) * SLAB;
#pragma unroll
                    for (int i = 0; i < FM; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const f32x4 v = *reinterpret_cast<const f32x4*>(slab + ((i * 4 + j) * NT + tid) * 4);
                            acc[i][j] = c == g_first ? v : acc[i][j] + v;
                        }
                }
            }
        }
        DM_TL(tl_seg, 5);  // (last holder of a split tile: slabs summed) epilogue starts
        if (DM_TL_ON(8)) epilogue<T, FM>(p, acc, m0 + wm * 16 * FM, n0 + wn * 64, fr, fq, blockIdx.y, ln_mu, ln_rs, ln_cs);
        DM_TL(tl_seg, 6);  // epilogue issued
    }
}
// Waves per SIMD that the LDS ring lets a CU hold (capped at 4): the register budget the compiler must
// respect, so that a rarely taken path (the partial-tile sum) can never cost the main loop its occupancy.
constexpr int dma_min_waves(int bm, int bn, int waves, int stages) {
    const int blocks = 163840 / (stages * (bm + bn) * 128);
    const int w = (blocks < 8 ? blocks : 8) * waves / 4;
    return w < 1 ? 1 : (w > 4 ? 4 : w);
}
template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, int STAGES, bool EARLY = false>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64, dma_min_waves(BM, BN, WAVES_M* WAVES_N, STAGES)) void gemm_dma_kernel(
    GemmArgs p) {
    gemm_dma_body<T, BM, BN, WAVES_M, WAVES_N, STAGES, false, EARLY>(p);
}
template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, int STAGES, bool EARLY = false>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64, dma_min_waves(BM, BN, WAVES_M* WAVES_N, STAGES)) void conv3x3_igemm_kernel(
    GemmArgs p) {
    gemm_dma_body<T, BM, BN, WAVES_M, WAVES_N, STAGES, true, EARLY>(p);
}
// bf16, weights with the chunk-major k axis (stride 1, same size)
template <int BM, int BN, int WAVES_M, int WAVES_N, int STAGES, bool EARLY = false>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64, dma_min_waves(BM, BN, WAVES_M* WAVES_N, STAGES)) void conv3x3_chunk_kernel(
    GemmArgs p) {
    gemm_dma_body<bf16_t, BM, BN, WAVES_M, WAVES_N, STAGES, true, EARLY, false, 0, 1, true>(p);
}

// fp32 activations, split-bf16 weight image (mma_tile_split)
// two k-tiles per barrier (four stages)
template <typename T, int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64, dma_min_waves(BM, BN, WAVES_M* WAVES_N, 4)) void gemm_pair_kernel(GemmArgs p) {
    gemm_dma_body<T, BM, BN, WAVES_M, WAVES_N, 4, false, false, false, 0, 2>(p);
}
template <typename T, int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64, dma_min_waves(BM, BN, WAVES_M* WAVES_N, 4)) void conv3x3_pair_kernel(GemmArgs p) {
    gemm_dma_body<T, BM, BN, WAVES_M, WAVES_N, 4, true, false, false, 0, 2>(p);
}
// bf16, the chunk's activation resident in LDS with its border (weights in the half-major unit order)
__global__ __launch_bounds__(256, 2) void conv3x3_halo_kernel(GemmArgs p) {
    gemm_dma_body<bf16_t, 128, 128, 2, 2, 2, true, false, false, 0, 1, true, true>(p);
}
template <int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64, dma_min_waves(BM, BN, WAVES_M* WAVES_N, 4)) void conv3x3_chunk_pair_kernel(GemmArgs p) {
    gemm_dma_body<bf16_t, BM, BN, WAVES_M, WAVES_N, 4, true, false, false, 0, 2, true>(p);
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int STAGES>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64, dma_min_waves(BM, BN, WAVES_M* WAVES_N, STAGES)) void gemm_split_kernel(
    GemmArgs p) {
    gemm_dma_body<float, BM, BN, WAVES_M, WAVES_N, STAGES, false, false, false, 2>(p);
}
template <int BM, int BN, int WAVES_M, int WAVES_N, int STAGES>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64, dma_min_waves(BM, BN, WAVES_M* WAVES_N, STAGES)) void conv3x3_split_kernel(
    GemmArgs p) {
    gemm_dma_body<float, BM, BN, WAVES_M, WAVES_N, STAGES, true, false, false, 2>(p);
}
// three terms per operand: the W tile takes twice the LDS
template <int BM, int BN, int WAVES_M, int WAVES_N, int STAGES>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64, dma_min_waves(BM, 2 * BN, WAVES_M* WAVES_N, STAGES)) void gemm_split3_kernel(
    GemmArgs p) {
    gemm_dma_body<float, BM, BN, WAVES_M, WAVES_N, STAGES, false, false, false, 3>(p);
}
template <int BM, int BN, int WAVES_M, int WAVES_N, int STAGES>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64, dma_min_waves(BM, 2 * BN, WAVES_M* WAVES_N, STAGES)) void conv3x3_split3_kernel(
    GemmArgs p) {
    gemm_dma_body<float, BM, BN, WAVES_M, WAVES_N, STAGES, true, false, false, 3>(p);
}

// fp32 with the GroupNorm of the A operand applied to the fragments (conv: + SiLU; dense: affine only)
// The table behind the ring lowers the blocks a CU holds, and the map keeps a k-tile's fragments and the scale / shift
// chunks live: half the waves per SIMD of the plain kernel, i.e. twice its register budget.
constexpr int gn_min_waves(int bm, int bn, int waves, int stages) {
    const int w = dma_min_waves(bm, bn, waves, stages) / 2;
    return w < 1 ? 1 : w;
}
template <int BM, int BN, int WAVES_M, int WAVES_N, int STAGES>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64, gn_min_waves(BM, BN, WAVES_M* WAVES_N, STAGES)) void gemm_gn_kernel(
    GemmArgs p) {
    gemm_dma_body<float, BM, BN, WAVES_M, WAVES_N, STAGES, false, false, true>(p);
}
template <int BM, int BN, int WAVES_M, int WAVES_N, int STAGES>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64, gn_min_waves(BM, BN, WAVES_M* WAVES_N, STAGES)) void conv3x3_gn_kernel(
    GemmArgs p) {
    gemm_dma_body<float, BM, BN, WAVES_M, WAVES_N, STAGES, true, false, true>(p);
}

// ------------------------------------------------------------------------------------------------
// 256 x 256 block tile, 8 waves (2 x 4), 128 x 64 outputs per wave, two waves per SIMD taking turns
// ------------------------------------------------------------------------------------------------
// Why this tile exists: a 128 x 128 tile pulls (128 + 128) x 128 B through the CU's vector-memory path for
// every 512 MFMA cycles (64 B per clock, that path's whole width); here it is 512 x 128 B per 2048 cycles, half
// of that.  The k-tile (128 bytes of k) lives in LDS as four "halves" of 128 rows:
//   A0 / A1: for both wave rows wm, fragments i = 4*ih .. 4*ih+3 of the wave's 128 output rows
//            (LDS row = wm*64 + (i & 3)*16 + r  <->  tile row wm*128 + i*16 + r)
//   W0 / W1: for all four wave columns wn, fragments j = 2*jh, 2*jh+1 of the wave's 64-column group
//            (LDS row = wn*32 + (j & 1)*16 + r  <->  packed weight row wn*64 + j*16 + r)
// and a wave walks its output tile in four quadrants (ih, jh) = (0,0) (0,1) (1,1) (1,0): after the first quadrant's
// fragment reads W0 and A0 are dead, after the second W1, after the third A1, so every half is refilled with the
// k-tile TWO steps ahead soon after its last read (W0 in the second phase, A0 in the third, W1 and A1 in the fourth,
// which has no reads: the request-issue cost is spread evenly over the phases) - more than one k-tile of latency
// cover per request, inside two stages (128 KiB).  All fragments of a k-tile stay in registers (A half 32 + W 32 registers besides the 128
// accumulators), so the fourth quadrant reads nothing.
// Schedule: a phase is [R: refill requests, fragment reads, counted vmcnt] barrier [M: 16 MFMAs] barrier.  Wave row 1
// runs one barrier behind wave row 0, so on every SIMD one wave is in its M slot while the other is in its R slot.
// Hazards (slot numbers count barriers; wave row 0 reads phase p in slot 2p, wave row 1 in slot 2p+1):
//   RAW  a half read in phase p+1 is waited for (own DMA pieces, counted vmcnt) at the end of R(p) by every wave,
//        i.e. before barriers 2p / 2p+1, and first read in slot 2p+2;
//   WAR  a half last read in phase p is refilled in R(p+1) or later, i.e. after barrier 2p+1, which both wave rows
//        reach only with those reads returned (lgkmcnt(0) closes every R).
// Per thread and k-tile the requests are issued in the order W0 A0 W1 A1 (two 16-byte pieces each); the counted waits
// below follow from that order.
template <int OFF> __device__ __forceinline__ void lds_read_b128(u32x4& dst, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}

template <typename T, int IH, int JH>
__device__ __forceinline__ void wide_quadrant(f32x4 (&acc)[8][4], const u32x4 (&afr)[4][2], const u32x4 (&wfr)[2][2][2]) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int ii = 0; ii < 4; ++ii)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) Mma16<T>::mma(acc[4 * IH + ii][2 * JH + jj], wfr[JH][jj][ks], afr[ii][ks]);
}

template <typename T, bool CONV>
__device__ __forceinline__ void gemm_wide_body(const GemmArgs& p) {
    constexpr int BM = 256, BN = 256, NT = 512, FM = 8;
    constexpr int EPC = ElemTraits<T>::EPC;
    constexpr int BK = 8 * EPC;
    constexpr int HALF = 128 * 128;   // bytes: 128 LDS rows of 128 B
    constexpr int STAGE = 4 * HALF;   // A0 A1 W0 W1

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
    const int nkt = p.Kpad / BK;
    const int fr = lane & 15, fq = lane >> 4;
    const bool m_fast = (p.conv ? 9 * p.N : p.N) > p.M;
    const char* zsrc = reinterpret_cast<const char*>(g_zero16);
    const int lds_wave = wave * 1024;
    const int lchunk = (tid & 7) ^ ((tid >> 4) & 7);  // logical 16-byte chunk of this thread's DMA slots (see gemm_dma_body)
    // fragment read addresses inside a stage: one register per k-step and operand, everything else is an immediate
    const uint32_t lds0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
    uint32_t a_rd[2], w_rd[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int ch = ((ks * 4 + fq) ^ ((fr >> 1) & 7)) << 4;
        a_rd[ks] = lds0 + (wm * 64 + fr) * 128 + ch;
        w_rd[ks] = lds0 + 2 * HALF + (wn * 32 + fr) * 128 + ch;
    }
    // DMA slots of this thread in every half: LDS rows lr = (tid >> 3) + 64 * it, it = 0, 1
    const int r16 = (tid >> 3) & 15;
    const int a_sub = ((tid >> 7) & 3) * 16 + r16;             // + it*128 + h*64: tile row
    const int w_sub = (tid >> 8) * 64 + ((tid >> 7) & 1) * 16 + r16;  // + it*128 + h*32: packed weight row of the tile

    long u = 0, u_end = 1;
    int sk_g = 0;
    if (p.streamk) {
        sk_g = xcd_remap(blockIdx.x, gridDim.x);
        const long U = (long)ntm * ntn * nkt;
        u = sk_g * U / gridDim.x;
        u_end = (sk_g + 1) * U / gridDim.x;
    }

    while (u < u_end) {
        int tile_id, kt0, kt1;
        if (p.streamk) {
            tile_id = (int)(u / nkt);
            kt0 = (int)(u - (long)tile_id * nkt);
            kt1 = (int)min((long)nkt, kt0 + (u_end - u));
            u += kt1 - kt0;
            __syncthreads();
        } else {
            tile_id = xcd_remap(blockIdx.x, ntm * ntn);
            kt0 = 0;
            kt1 = nkt;
            u = u_end;
        }
        const int tile_m = m_fast ? tile_id % ntm : tile_id / ntn;
        const int tile_n = m_fast ? tile_id / ntm : tile_id - tile_m * ntn;
        const int m0 = tile_m * BM, n0 = tile_n * BN;
        const int nt = kt1 - kt0;

        // ---- source addressing.  Persistent per-thread state is two 32-bit offsets and the tap masks; everything
        // else about a request (tile origin, k-tile, tap, wave-row / half) is wave-uniform and lives in SGPRs.
        // A row r = (h, it) of this thread is tile row it*128 + h*64 + a_sub.  In a stride-1, same-size convolution the
        // input pixel of (output row m, tap) is simply row m + dy*Win + dx of the image matrix, so dense and conv share
        // "row-linear" addressing; what differs per row is only whether a tap falls outside the image.
        const uint32_t a_voff = (uint32_t)(((long)a_sub * p.lda + lchunk * EPC) * sizeof(T));
        const uint32_t a2_voff = (uint32_t)(((long)a_sub * p.lda2 + lchunk * EPC) * sizeof(T));
        const uint32_t w_voff = (uint32_t)(((long)(tid >> 8) * nkt * 64 + (((tid >> 7) & 1) * 16 + r16)) * 128 + lchunk * 16);
        // amask[h]: bits 0..9 for it = 0, bits 16..25 for it = 1; bit t = tap t of that row reads real data
        // (dense: bit 0 = row inside M; conv: taps 0..8 inside the image, bit 9 = the appended pointwise operand)
        uint32_t amask[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            amask[h] = 0;
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int m = m0 + it * 128 + h * 64 + a_sub;
                uint32_t bits = 0;
                if (m < p.M) {
                    if (CONV) {
                        const int hw = p.Hout * p.Wout;
                        const int pix = m % hw;
                        const int y = pix / p.Wout, x = pix - y * p.Wout;
#pragma unroll
                        for (int t9 = 0; t9 < 9; ++t9) {
                            const int iy = y + t9 / 3 - 1, ix = x + t9 % 3 - 1;
                            if ((unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win) bits |= 1u << t9;
                        }
                        if (p.A2) bits |= 1u << 9;
                    } else {
                        bits = 1u;
                    }
                }
                amask[h] |= bits << (16 * it);
            }
        }
        int tap[2], c0[2];  // conv: tap and channel offset of the next k-tile to request, per A half
        if (CONV && p.korder) {
            chunk_major_at(kt0, p.Cin, BK, tap[0], c0[0]);
            tap[1] = tap[0];
            c0[1] = c0[0];
        } else if (CONV) {
            const int kc = kt0 * BK;
            tap[0] = tap[1] = min(kc / p.Cin, 9);
            c0[0] = c0[1] = kc - tap[0] * p.Cin;
        } else {
            tap[0] = tap[1] = c0[0] = c0[1] = 0;
        }
        const int kt_second = (!CONV && p.A2) ? (p.K - p.Cin2) / BK : 0x7fffffff;
        const long w_tile_base = (long)(n0 >> 6) * nkt * W_KTILE_STEP;
        // requests of one half of data k-tile `kt` into stage `st` (both wave-uniform)
        auto issue_a = [&](auto h_, int kt, int st) {
            constexpr int h = decltype(h_)::value;
            char* dst = smem + st * STAGE + h * HALF + lds_wave;
            // wave-uniform part of the source address
            const char* base;
            uint32_t voff = a_voff;
            int bit = 0;
            if (CONV) {
                bit = tap[h];
                if (tap[h] >= 9) {
                    base = p.A2 + ((long)(m0 + h * 64) * p.lda2 + c0[h]) * sizeof(T);
                    voff = a2_voff;
                } else {
                    const int dy = tap[h] / 3 - 1, dx = tap[h] - (tap[h] / 3) * 3 - 1;
                    base = p.A + ((long)(m0 + h * 64 + dy * p.Win + dx) * p.lda + c0[h]) * sizeof(T);
                }
            } else if (kt >= kt_second) {
                base = p.A2 + ((long)(m0 + h * 64) * p.lda2 + (long)(kt - kt_second) * BK) * sizeof(T);
                voff = a2_voff;
            } else {
                base = p.A + ((long)(m0 + h * 64) * p.lda + (long)kt * BK) * sizeof(T);
            }
            const long row_step = 128L * ((CONV ? tap[h] >= 9 : kt >= kt_second) ? p.lda2 : p.lda) * sizeof(T);
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const bool good = (amask[h] >> (16 * it + bit)) & 1u;
                const char* src = good ? base + it * row_step + voff : zsrc;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(dst + it * 8192), 16, 0, 0);
            }
            if (CONV && p.korder) {
                chunk_major_next(p.Cin, BK, tap[h], c0[h]);
            } else if (CONV) {
                c0[h] += BK;
                if (c0[h] == p.Cin && tap[h] < 9) {
                    c0[h] = 0;
                    ++tap[h];
                }
            }
        };
        auto issue_w = [&](auto h_, int kt, int st) {
            constexpr int h = decltype(h_)::value;
            char* dst = smem + st * STAGE + (2 + h) * HALF + lds_wave;
            const char* base = p.W + w_tile_base + (long)kt * W_KTILE_STEP + h * 4096;
#pragma unroll
            for (int it = 0; it < 2; ++it)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(base + (long)it * 2 * nkt * W_KTILE_STEP + w_voff),
                    (__attribute__((address_space(3))) void*)(dst + it * 8192), 16, 0, 0);
        };
        constexpr std::integral_constant<int, 0> H0{};
        constexpr std::integral_constant<int, 1> H1{};

        f32x4 acc[FM][4];
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        // ---- prologue: k-tiles kt0 and kt0+1 into stages 0 and 1
        issue_w(H0, kt0, 0); issue_a(H0, kt0, 0); issue_w(H1, kt0, 0); issue_a(H1, kt0, 0);
        if (nt > 1) { issue_w(H0, kt0 + 1, 1); issue_a(H0, kt0 + 1, 1); issue_w(H1, kt0 + 1, 1); issue_a(H1, kt0 + 1, 1); }
        if (nt > 1) wait_vm<12>(); else wait_vm<4>();  // W0, A0 of the first k-tile have landed
        __builtin_amdgcn_s_barrier();
        if (wm == 1) __builtin_amdgcn_s_barrier();  // wave row 1 runs one barrier behind from here on

        u32x4 afr[4][2], wfr[2][2][2];
        for (int t = 0; t < nt; ++t) {
            const int st = t & 1;
            const bool more1 = t + 1 < nt, more2 = t + 2 < nt;
            const uint32_t a0 = a_rd[0] + st * STAGE, a1 = a_rd[1] + st * STAGE;
            const uint32_t w0 = w_rd[0] + st * STAGE, w1 = w_rd[1] + st * STAGE;
            // ---- quadrant (0, 0): reads W0 and A0
            lds_read_b128<0>(wfr[0][0][0], w0); lds_read_b128<0>(wfr[0][0][1], w1);
            lds_read_b128<2048>(wfr[0][1][0], w0); lds_read_b128<2048>(wfr[0][1][1], w1);
            lds_read_b128<0>(afr[0][0], a0); lds_read_b128<0>(afr[0][1], a1);
            lds_read_b128<2048>(afr[1][0], a0); lds_read_b128<2048>(afr[1][1], a1);
            lds_read_b128<4096>(afr[2][0], a0); lds_read_b128<4096>(afr[2][1], a1);
            lds_read_b128<6144>(afr[3][0], a0); lds_read_b128<6144>(afr[3][1], a1);
            if (more1) wait_vm<10>(); else wait_vm<2>();  // W1 of this k-tile (younger: A1, the whole next k-tile)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_setprio(1);
            wide_quadrant<T, 0, 0>(acc, afr, wfr);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            // ---- quadrant (0, 1): W0 is refilled with k-tile t+2; reads W1
            if (more2) issue_w(H0, kt0 + t + 2, st);
            lds_read_b128<HALF>(wfr[1][0][0], w0); lds_read_b128<HALF>(wfr[1][0][1], w1);
            lds_read_b128<HALF + 2048>(wfr[1][1][0], w0); lds_read_b128<HALF + 2048>(wfr[1][1][1], w1);
            if (more2) wait_vm<10>(); else if (more1) wait_vm<8>(); else wait_vm<0>();  // A1 of this k-tile
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_setprio(1);
            wide_quadrant<T, 0, 1>(acc, afr, wfr);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            // ---- quadrant (1, 1): A0 is refilled; reads A1
            if (more2) issue_a(H0, kt0 + t + 2, st);
            lds_read_b128<HALF>(afr[0][0], a0); lds_read_b128<HALF>(afr[0][1], a1);
            lds_read_b128<HALF + 2048>(afr[1][0], a0); lds_read_b128<HALF + 2048>(afr[1][1], a1);
            lds_read_b128<HALF + 4096>(afr[2][0], a0); lds_read_b128<HALF + 4096>(afr[2][1], a1);
            lds_read_b128<HALF + 6144>(afr[3][0], a0); lds_read_b128<HALF + 6144>(afr[3][1], a1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_setprio(1);
            wide_quadrant<T, 1, 1>(acc, afr, wfr);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            // ---- quadrant (1, 0): W1 and A1 are refilled; nothing to read; W0, A0 of the next k-tile must have landed
            if (more2) { issue_w(H1, kt0 + t + 2, st); issue_a(H1, kt0 + t + 2, st); }
            if (more1) { if (more2) wait_vm<12>(); else wait_vm<4>(); }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_setprio(1);
            wide_quadrant<T, 1, 0>(acc, afr, wfr);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
        }
        if (wm == 0) __builtin_amdgcn_s_barrier();  // both wave rows have now passed the same number of barriers
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

        if (p.streamk && !(kt0 == 0 && kt1 == nkt)) {
            // partial tile: same slab / ticket protocol as gemm_dma_body
            const long ut0 = (long)tile_id * nkt;
            const long U = (long)ntm * ntn * nkt;
            const int G = gridDim.x;
            const int g_first = (int)(((ut0 + 1) * G + U - 1) / U) - 1;
            const int g_last = (int)(((ut0 + nkt) * G + U - 1) / U) - 1;
            constexpr long SLAB = (long)BM * BN;
            {
                float* slab = p.slabs + (2L * sk_g + (kt0 == 0 ? 1 : 0)) * SLAB;
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(slab, 0, (int)(SLAB * 4), 0x00020000);
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), rs,
                                                               ((i * 4 + j) * NT + tid) * 16, 0, 16);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            int* ticket = reinterpret_cast<int*>(smem);
            if (tid == 0)
                *ticket = __hip_atomic_fetch_add(p.counters + tile_id, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            const bool last = *ticket == g_last - g_first;
            if (!last) continue;
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(p.counters + tile_id, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
            for (int c = g_first; c <= g_last; ++c) {
                const float* slab = p.slabs + (2L * c + (c == g_first ? 1 : 0)) * SLAB;
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x4 v = *reinterpret_cast<const f32x4*>(slab + ((i * 4 + j) * NT + tid) * 4);
                        acc[i][j] = c == g_first ? v : acc[i][j] + v;
                    }
            }
        }
        // The wave's 128 rows in four passes of 32, fenced from each other: with 128 accumulator registers live, an
        // epilogue over more fragment rows at once (its loads hoisted for latency) spills by the hundred.
        f32x4 pre_b[4];
        load_bias(p, n0 + wn * 64, fq, pre_b);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float ln_mu[2], ln_rs[2];
            f32x4 ln_cs[4];
            const int mrow = m0 + wm * 128 + q * 32;
            ln_row_stats<2>(p, mrow, n0 + wn * 64, fr, fq, ln_mu, ln_rs, ln_cs);
            epilogue<T, 2>(p, reinterpret_cast<f32x4(&)[2][4]>(acc[2 * q]), mrow, n0 + wn * 64, fr, fq, 0, ln_mu, ln_rs, ln_cs,
                           pre_b);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <typename T> __global__ __launch_bounds__(512, 2) void gemm_wide_kernel(GemmArgs p) { gemm_wide_body<T, false>(p); }
template <typename T> __global__ __launch_bounds__(512, 2) void conv3x3_wide_kernel(GemmArgs p) { gemm_wide_body<T, true>(p); }

// ------------------------------------------------------------------------------------------------
// register-staged main loop
// ------------------------------------------------------------------------------------------------
template <typename T, int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64) void gemm_reg_kernel(GemmArgs p) {
    constexpr int NT = WAVES_M * WAVES_N * 64;
    constexpr int FM = BM / (16 * WAVES_M);
    static_assert(BN == 64 * WAVES_N, "each wave owns 64 output columns");
    static_assert(FM >= 1 && BM % (16 * WAVES_M) == 0, "bad tile");
    constexpr int EPC = ElemTraits<T>::EPC;
    constexpr int BK = 8 * EPC;
    constexpr int A_IT = (BM * 8 + NT - 1) / NT;
    constexpr int W_IT = (BN * 8 + NT - 1) / NT;
    constexpr int TILE_BYTES = (BM + BN) * 128;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
    const int bid = xcd_remap(blockIdx.x, ntm * ntn);
    // weight-heavy shapes (N > M) walk M fastest so the blocks sharing a W tile sit in one XCD's L2;
    // activation-heavy shapes walk N fastest so the A tile is the shared one
    const bool m_fast = (p.conv ? 9 * p.N : p.N) > p.M;  // unique W bytes vs unique A bytes
    const int tile_m = m_fast ? bid % ntm : bid / ntn;
    const int tile_n = m_fast ? bid / ntm : bid - tile_m * ntn;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int nkt = p.Kpad / BK;
    int kt0 = 0, kt1 = nkt;
    if (p.splitk > 1) {
        const int per = (nkt + p.splitk - 1) / p.splitk;
        kt0 = blockIdx.y * per;
        kt1 = min(nkt, kt0 + per);
    }

    const int chunk = tid & 7;
    ARows<A_IT> rows;
    init_rows<A_IT, NT / 8>(rows, p, m0, tid >> 3, BM);
    const char* w_ptr[W_IT];
#pragma unroll
    for (int it = 0; it < W_IT; ++it)
        w_ptr[it] = p.W + w_block_offset(n0 + it * (NT / 8) + (tid >> 3), nkt, 0) + chunk * 16;
    int kc = kt0 * BK + chunk * EPC, tap = 0;
    if (p.conv) {
        tap = kc / p.Cin;
        kc -= tap * p.Cin;
    }

    u32x4 ra[A_IT], rw[W_IT];
    auto load_tile = [&](int kt) {
        if (p.conv) {
#pragma unroll
            for (int it = 0; it < A_IT; ++it) {
                long e;
                const bool ok = conv_src<A_IT>(rows, p, it, tap, kc, e);
                ra[it] = ldg16(p.A + e * sizeof(T), ok);
            }
            kc += BK;
            while (kc >= p.Cin) {
                kc -= p.Cin;
                ++tap;
            }
        } else {
            const int k = kt * BK + chunk * EPC;
#pragma unroll
            for (int it = 0; it < A_IT; ++it)
                ra[it] = ldg16(p.A + (rows.off[it] + k) * sizeof(T), rows.ok[it] && k < p.K);
        }
#pragma unroll
        for (int it = 0; it < W_IT; ++it) {
            const int row = it * (NT / 8) + (tid >> 3);
            rw[it] = ldg16(w_ptr[it] + (long)kt * W_KTILE_STEP, row < BN);
        }
    };
    auto store_tile = [&](int buf) {
        char* sa = smem + buf * TILE_BYTES;
        char* sw = sa + BM * 128;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int row = it * (NT / 8) + (tid >> 3);
            if (row < BM) *reinterpret_cast<u32x4*>(sa + swz(row, chunk)) = ra[it];
        }
#pragma unroll
        for (int it = 0; it < W_IT; ++it) {
            const int row = it * (NT / 8) + (tid >> 3);
            if (row < BN) *reinterpret_cast<u32x4*>(sw + swz(row, chunk)) = rw[it];
        }
    };

    f32x4 acc[FM][4];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fq = lane >> 4;

    if (kt0 < kt1) {
        load_tile(kt0);
        store_tile(0);
    }
    __syncthreads();
    for (int kt = kt0; kt < kt1; ++kt) {
        const int cur = (kt - kt0) & 1;
        const bool more = kt + 1 < kt1;
        if (more) load_tile(kt + 1);
        const char* sa = smem + cur * TILE_BYTES;
        mma_tile<T, FM>(acc, sa, sa + BM * 128, wm * 16 * FM, wn * 64, fr, fq);
        if (more) store_tile(cur ^ 1);
        __syncthreads();
    }
    float ln_mu[FM], ln_rs[FM];
    f32x4 ln_cs[4];
    ln_row_stats<FM>(p, m0 + wm * 16 * FM, n0 + wn * 64, fr, fq, ln_mu, ln_rs, ln_cs);
    epilogue<T, FM>(p, acc, m0 + wm * 16 * FM, n0 + wn * 64, fr, fq, blockIdx.y, ln_mu, ln_rs, ln_cs);
}

// Sum split-K partial slabs in slab order (deterministic) and apply the plain epilogue.
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmArgs p) {
    const long n4 = p.N >> 2;
    const long total = (long)p.M * n4;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long m = idx / n4;
        const int n = (int)(idx - m * n4) << 2;
        f32x4 s = *reinterpret_cast<const f32x4*>(p.partial + m * p.N + n);
        for (int z = 1; z < p.splitk; ++z) s += *reinterpret_cast<const f32x4*>(p.partial + ((long)z * p.M + m) * p.N + n);
        float o[4] = {s[0], s[1], s[2], s[3]};
        if (p.bias)
            for (int r = 0; r < 4; ++r) o[r] += p.bias[n + r];
        if (p.rowbias) {
            const long b = m / p.rows_per_batch;
            for (int r = 0; r < 4; ++r) o[r] += p.rowbias[b * p.ld_rowbias + n + r];
        }
        if (p.R) {
            const T* rp = reinterpret_cast<const T*>(p.R) + m * p.ldr + n;
            for (int r = 0; r < 4; ++r) o[r] += ElemTraits<T>::load(rp + r);
        }
        T* dst = reinterpret_cast<T*>(p.C) + m * p.ldc + n;
        for (int r = 0; r < 4; ++r) ElemTraits<T>::store(dst + r, o[r]);
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
struct TileInfo {
    int bm, bn, threads, lds;
};

// lds_attr: the dynamic-LDS ceiling registered for the kernel when launches of it differ in their LDS size
template <typename K> int launch_kernel(K kfn, DmLdsAttr& attr_done, const TileInfo& t, const GemmArgs& a, hipStream_t stream,
                                        int lds_attr = 0) {
    int lds = t.lds;
#ifdef DM_TIMELINE
    // diagnostic build: DM_TL_LDS_PAD=<bytes> of unused LDS per block (fewer co-resident blocks per CU, same code)
    static const int pad = getenv("DM_TL_LDS_PAD") ? atoi(getenv("DM_TL_LDS_PAD")) : 0;
    if (pad > 0 && lds + pad <= 160 * 1024) {
        lds += pad;
        if (!lds_attr || lds_attr < lds) lds_attr = lds;
    }
#endif
    if (const int rc = dm_set_lds_attr(reinterpret_cast<const void*>(kfn), attr_done, lds_attr ? lds_attr : lds)) return rc;
    const int ntm = (a.M + t.bm - 1) / t.bm, ntn = (a.N + t.bn - 1) / t.bn;
    dim3 grid(a.streamk ? a.streamk : ntm * ntn, a.splitk > 1 ? a.splitk : 1, 1);
    hipLaunchKernelGGL(kfn, grid, dim3(t.threads), lds, stream, a);
    return dm_check_launch();
}

template <typename T, int BM, int BN, int WM, int WN> int launch_reg(const GemmArgs& a, hipStream_t s) {
    static DmLdsAttr done;
    if (a.streamk) return DM_ERR_ARG;  // the work-list form exists for the direct-to-LDS tiles only
    const TileInfo t{BM, BN, WM * WN * 64, 2 * (BM + BN) * 128};
    return launch_kernel(gemm_reg_kernel<T, BM, BN, WM, WN>, done, t, a, s);
}
template <typename T, int BM, int BN, int WM, int WN, int ST, bool EARLY = false>
int launch_dma(const GemmArgs& a, hipStream_t s) {
    static DmLdsAttr done, done_conv;
    const TileInfo t{BM, BN, WM * WN * 64, ST * (BM + BN) * 128};
    if constexpr (sizeof(T) == 2) {
        static DmLdsAttr done_chunk;
        if (a.conv && a.korder == 1) return launch_kernel(conv3x3_chunk_kernel<BM, BN, WM, WN, ST, EARLY>, done_chunk, t, a, s);
    }
    if (a.korder) return DM_ERR_ARG;
    if (a.conv) return launch_kernel(conv3x3_igemm_kernel<T, BM, BN, WM, WN, ST, EARLY>, done_conv, t, a, s);
    return launch_kernel(gemm_dma_kernel<T, BM, BN, WM, WN, ST, EARLY>, done, t, a, s);
}

template <typename T, int BM, int BN, int WM, int WN> int launch_pair(const GemmArgs& a, hipStream_t s) {
    static DmLdsAttr done, done_conv;
    const TileInfo t{BM, BN, WM * WN * 64, 4 * (BM + BN) * 128};
    if constexpr (sizeof(T) == 2) {
        static DmLdsAttr done_chunk;
        if (a.conv && a.korder == 1) return launch_kernel(conv3x3_chunk_pair_kernel<BM, BN, WM, WN>, done_chunk, t, a, s);
    }
    if (a.korder) return DM_ERR_ARG;
    if (a.conv) return launch_kernel(conv3x3_pair_kernel<T, BM, BN, WM, WN>, done_conv, t, a, s);
    return launch_kernel(gemm_pair_kernel<T, BM, BN, WM, WN>, done, t, a, s);
}

template <typename T> int launch_halo(const GemmArgs& a, hipStream_t s) {
    if constexpr (sizeof(T) == 2) {
        static DmLdsAttr done;
        const TileInfo t{128, 128, 256, 2 * 16384 + 2 * 20480};
        if (!a.conv || a.korder != 2 || a.splitk > 1) return DM_ERR_ARG;
        return launch_kernel(conv3x3_halo_kernel, done, t, a, s);
    }
    return DM_ERR_ARG;
}

// fp32 with the GroupNorm of A applied to the fragments: the scale / shift table sits in LDS behind the ring
template <int BM, int BN, int WM, int WN, int ST> int launch_gn(const GemmArgs& a, hipStream_t s) {
    static DmLdsAttr done, done_conv;
    const int ring = ST * (BM + BN) * 128;
    const TileInfo t{BM, BN, WM * WN * 64, ring + gn_fuse_lds_bytes(a.gn_c)};
    const int ceiling = ring + gn_fuse_lds_bytes(GN_FUSE_MAX_C);
    if (a.conv) return launch_kernel(conv3x3_gn_kernel<BM, BN, WM, WN, ST>, done_conv, t, a, s, ceiling);
    return launch_kernel(gemm_gn_kernel<BM, BN, WM, WN, ST>, done, t, a, s, ceiling);
}
bool gn_tile(int cfg) {
    switch (cfg) {
        case DM_TILE_D128x128_S2: case DM_TILE_D64x128_S3: case DM_TILE_D64x64_S4: case DM_TILE_D32x64_S4:
        case DM_TILE_D128x64_S2: case DM_TILE_D64x64_S2:
            return true;
        default: return false;
    }
}
int launch_gn_typed(const GemmArgs& a, int cfg, hipStream_t s) {
    switch (cfg) {
        case DM_TILE_D128x128_S2: return launch_gn<128, 128, 2, 2, 2>(a, s);
        case DM_TILE_D64x128_S3: return launch_gn<64, 128, 2, 2, 3>(a, s);
        case DM_TILE_D64x64_S4: return launch_gn<64, 64, 4, 1, 4>(a, s);
        case DM_TILE_D32x64_S4: return launch_gn<32, 64, 2, 1, 4>(a, s);
        case DM_TILE_D128x64_S2: return launch_gn<128, 64, 4, 1, 2>(a, s);
        case DM_TILE_D64x64_S2: return launch_gn<64, 64, 4, 1, 2>(a, s);
        default: return DM_ERR_ARG;
    }
}

template <int BM, int BN, int WM, int WN, int ST> int launch_split(const GemmArgs& a, int terms, hipStream_t s) {
    static DmLdsAttr done, done_conv, done3, done3_conv;
    if (terms == 3) {
        const TileInfo t3{BM, BN, WM * WN * 64, ST * (BM + 2 * BN) * 128};
        if (a.conv) return launch_kernel(conv3x3_split3_kernel<BM, BN, WM, WN, ST>, done3_conv, t3, a, s);
        return launch_kernel(gemm_split3_kernel<BM, BN, WM, WN, ST>, done3, t3, a, s);
    }
    const TileInfo t{BM, BN, WM * WN * 64, ST * (BM + BN) * 128};
    if (a.conv) return launch_kernel(conv3x3_split_kernel<BM, BN, WM, WN, ST>, done_conv, t, a, s);
    return launch_kernel(gemm_split_kernel<BM, BN, WM, WN, ST>, done, t, a, s);
}
bool split_tile(int cfg) {
    switch (cfg) {
        case DM_TILE_D128x128_S2: case DM_TILE_D64x128_S3: case DM_TILE_D64x64_S4: case DM_TILE_D32x64_S4:
        case DM_TILE_D256x64_S2: case DM_TILE_D128x64_S2: case DM_TILE_D64x64_S2:
            return true;
        default: return false;
    }
}
int launch_split_typed(const GemmArgs& a, int cfg, int terms, hipStream_t s) {
    switch (cfg) {
        case DM_TILE_D128x128_S2: return launch_split<128, 128, 2, 2, 2>(a, terms, s);
        case DM_TILE_D64x128_S3: return launch_split<64, 128, 2, 2, 3>(a, terms, s);
        case DM_TILE_D64x64_S4: return launch_split<64, 64, 4, 1, 4>(a, terms, s);
        case DM_TILE_D32x64_S4: return launch_split<32, 64, 2, 1, 4>(a, terms, s);
        case DM_TILE_D256x64_S2: return launch_split<256, 64, 4, 1, 2>(a, terms, s);
        case DM_TILE_D128x64_S2: return launch_split<128, 64, 4, 1, 2>(a, terms, s);
        case DM_TILE_D64x64_S2: return launch_split<64, 64, 4, 1, 2>(a, terms, s);
        default: return DM_ERR_ARG;
    }
}

// 256 x 256 ping-pong tile: bf16 only (exact-fp32 MFMA is 16x slower per operand byte: operand traffic is not its limit)
template <typename T> int launch_wide(const GemmArgs& a, hipStream_t s) {
    if constexpr (sizeof(T) == 2) {
        static DmLdsAttr done, done_conv;
        const TileInfo t{256, 256, 512, 2 * 4 * 128 * 128};
        if (a.conv) return launch_kernel(conv3x3_wide_kernel<T>, done_conv, t, a, s);
        return launch_kernel(gemm_wide_kernel<T>, done, t, a, s);
    } else {
        return DM_ERR_ARG;
    }
}

template <typename T> int launch_typed(const GemmArgs& a, int cfg, hipStream_t s) {
    switch (cfg) {
        case DM_TILE_128x128: return launch_reg<T, 128, 128, 2, 2>(a, s);
        case DM_TILE_64x128: return launch_reg<T, 64, 128, 2, 2>(a, s);
        case DM_TILE_64x64: return launch_reg<T, 64, 64, 4, 1>(a, s);
        case DM_TILE_256x128: return launch_reg<T, 256, 128, 4, 2>(a, s);
        case DM_TILE_32x64: return launch_reg<T, 32, 64, 2, 1>(a, s);
        case DM_TILE_D128x128_S2: return launch_dma<T, 128, 128, 2, 2, 2>(a, s);
        case DM_TILE_D128x128_S3: return launch_dma<T, 128, 128, 2, 2, 3>(a, s);
        case DM_TILE_D256x128_S3: return launch_dma<T, 256, 128, 4, 2, 3>(a, s);
        case DM_TILE_D64x128_S3: return launch_dma<T, 64, 128, 2, 2, 3>(a, s);
        case DM_TILE_D64x128_S4: return launch_dma<T, 64, 128, 2, 2, 4>(a, s);
        case DM_TILE_D64x64_S4: return launch_dma<T, 64, 64, 4, 1, 4>(a, s);
        case DM_TILE_D32x64_S4: return launch_dma<T, 32, 64, 2, 1, 4>(a, s);
        case DM_TILE_D256x64_S3: return launch_dma<T, 256, 64, 4, 1, 3>(a, s);
        case DM_TILE_D128x64_S4: return launch_dma<T, 128, 64, 4, 1, 4>(a, s);
        case DM_TILE_D256x64_S2: return launch_dma<T, 256, 64, 4, 1, 2>(a, s);
        case DM_TILE_D128x64_S2: return launch_dma<T, 128, 64, 4, 1, 2>(a, s);
        case DM_TILE_D64x64_S2: return launch_dma<T, 64, 64, 4, 1, 2>(a, s);
        case DM_TILE_D256x256_S2: return launch_wide<T>(a, s);
        case DM_TILE_E128x128: return launch_dma<T, 128, 128, 2, 2, 2, true>(a, s);
        case DM_TILE_E128x64: return launch_dma<T, 128, 64, 4, 1, 2, true>(a, s);
        case DM_TILE_E64x64: return launch_dma<T, 64, 64, 4, 1, 2, true>(a, s);
        case DM_TILE_E64x128: return launch_dma<T, 64, 128, 2, 2, 2, true>(a, s);
        case DM_TILE_U64x64: return launch_pair<T, 64, 64, 4, 1>(a, s);
        case DM_TILE_H128x128: return launch_halo<T>(a, s);
        default: return DM_ERR_ARG;
    }
}

// rows of the output tile owned by one wave
int tile_wave_rows(int cfg) {
    switch (cfg) {
        case DM_TILE_64x64: case DM_TILE_32x64: case DM_TILE_D64x64_S4: case DM_TILE_D64x64_S2: case DM_TILE_D32x64_S4:
        case DM_TILE_E64x64: case DM_TILE_U64x64:
            return 16;
        case DM_TILE_64x128: case DM_TILE_D64x128_S3: case DM_TILE_D64x128_S4: case DM_TILE_D128x64_S4:
        case DM_TILE_D128x64_S2: case DM_TILE_E128x64: case DM_TILE_E64x128:
            return 32;
        case DM_TILE_D256x256_S2: return 32;  // its epilogue handles a wave's 128 rows in four passes of 32
        default: return 64;
    }
}

int tile_dims(int cfg, int* bm, int* bn) {
    switch (cfg) {
        case DM_TILE_128x128: case DM_TILE_D128x128_S2: case DM_TILE_D128x128_S3: case DM_TILE_E128x128:
        case DM_TILE_H128x128:
            *bm = 128; *bn = 128; return 0;
        case DM_TILE_64x128: case DM_TILE_D64x128_S3: case DM_TILE_D64x128_S4: case DM_TILE_E64x128:
            *bm = 64; *bn = 128; return 0;
        case DM_TILE_64x64: case DM_TILE_D64x64_S4: case DM_TILE_U64x64: *bm = 64; *bn = 64; return 0;
        case DM_TILE_256x128: case DM_TILE_D256x128_S3: *bm = 256; *bn = 128; return 0;
        case DM_TILE_32x64: case DM_TILE_D32x64_S4: *bm = 32; *bn = 64; return 0;
        case DM_TILE_D256x64_S3: *bm = 256; *bn = 64; return 0;
        case DM_TILE_D128x64_S4: case DM_TILE_D128x64_S2: case DM_TILE_E128x64:
            *bm = 128; *bn = 64; return 0;
        case DM_TILE_D256x64_S2: *bm = 256; *bn = 64; return 0;
        case DM_TILE_D64x64_S2: case DM_TILE_E64x64: *bm = 64; *bn = 64; return 0;
        case DM_TILE_D256x256_S2: *bm = 256; *bn = 256; return 0;
        default: return 1;
    }
}

}  // namespace

// Tile / split-K choice when the caller passes tile == DM_TILE_AUTO.
//
// Cost model in microseconds.  A CU holds up to `occ` blocks of a tile (LDS-limited); a block alone on
// a CU runs its MFMAs at the CU rate scaled by the SIMDs it occupies, blocks sharing a CU split the
// rate; every k-tile iteration also has a latency floor (barrier + LDS round trip).  Split-K adds the
// slab write/read and one more launch.  Rates are sustained figures measured per tile on MI355X.
extern "C" void dm_gemm_plan(int dtype, int M, int N, int K, int epi, int* tile, int* splitk) {
    struct Cand { int tile, waves, occ; double cu_tf_bf16, cu_tf_f32; };
    static const Cand cands[] = {
        {DM_TILE_D256x128_S3, 8, 1, 3.40, 0.48},
        {DM_TILE_D128x128_S2, 4, 2, 3.15, 0.48},
        {DM_TILE_D64x128_S3, 4, 2, 2.40, 0.47},
        {DM_TILE_D64x64_S4, 4, 2, 1.70, 0.44},
        {DM_TILE_D32x64_S4, 2, 3, 1.30, 0.42},
    };
    const int bk = dtype == DM_F32 ? 32 : 64;
    const int nkt = (K + bk - 1) / bk;
    const double t_lat = 0.30;  // floor per k-tile iteration of one block
    static const int sks[] = {1, 2, 3, 4, 6, 8, 12, 16, 24, 32};
    double best_cost = 1e30;
    int best = DM_TILE_D64x64_S4, best_sk = 1;
    for (const Cand& c : cands) {
        if (dtype == DM_F32 && c.tile == DM_TILE_D256x128_S3) continue;
        int bm, bn;
        tile_dims(c.tile, &bm, &bn);
        const double tiles = (double)((M + bm - 1) / bm) * ((N + bn - 1) / bn);
        const double cu_rate = (dtype == DM_F32 ? c.cu_tf_f32 : c.cu_tf_bf16) * 1e6;  // flop / us
        const double flops_tile = 2.0 * bm * bn * bk;
        const double t_alone = flops_tile / (cu_rate * (c.waves >= 4 ? 1.0 : c.waves / 4.0));
        for (int sk : sks) {
            if (sk > 1 && (epi != DM_EPI_PLAIN || nkt / sk < 4)) break;
            const double blocks = tiles * sk;
            const long per_cu = (long)((blocks + 255.0) / 256.0);
            const long conc = per_cu < c.occ ? per_cu : c.occ;
            const long rounds = (per_cu + c.occ - 1) / c.occ;
            double t_iter = conc * flops_tile / cu_rate;
            if (t_iter < t_alone) t_iter = t_alone;
            if (t_iter < t_lat) t_iter = t_lat;
            const double kslice = (double)((nkt + sk - 1) / sk);
            double cost = rounds * (kslice + 2.0) * t_iter + 3.0;
            if (sk > 1) cost += 2.5 + (double)sk * M * N * 8.0 / 4.0e6;
            if (cost < best_cost) {
                best_cost = cost;
                best = c.tile;
                best_sk = sk;
            }
        }
    }
    *tile = best;
    *splitk = best_sk;
}

extern "C" int dm_pack_weight_order(int n, int npad, int geglu, int* order) {
    if (!order || n <= 0 || npad % 64 || npad < n || (geglu && n % 64)) return DM_ERR_ARG;
    for (int p = 0; p < npad; ++p) {
        const int g = p >> 6, j = (p >> 4) & 3, q = (p >> 2) & 3, r = p & 3;
        int src, ok;
        if (!geglu) {
            src = 64 * g + 16 * q + 4 * j + r;
            ok = src < n;
        } else {
            const int half = n >> 1;
            const int col = 32 * g + 8 * q + 4 * (j & 1) + r;
            src = col + (j >= 2 ? half : 0);
            ok = col < half;
        }
        order[p] = ok ? src : -1;
    }
    return DM_OK;
}

namespace {
// One thread per 16-byte chunk of the packed image: packed row p, k-chunk kc.
template <typename T>
__global__ __launch_bounds__(256) void pack_rows_kernel(const T* __restrict__ src, long ld, int n, int k, int transposed,
                                                        T* __restrict__ dst, int npad, int kpad) {
    constexpr int EPC = ElemTraits<T>::EPC;
    constexpr int BK = 8 * EPC;
    const int kchunks = kpad / EPC, nkt = kpad / BK;
    const long total = (long)npad * kchunks;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int p = (int)(idx / kchunks), kc = (int)(idx - (long)p * kchunks);
        const int g = p >> 6, j = (p >> 4) & 3, q = (p >> 2) & 3, r = p & 3;
        const int row = 64 * g + 16 * q + 4 * j + r;  // dm_pack_weight_order, plain form
        const int k0 = kc * EPC;
        T v[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            const int kk = k0 + e;
            T x = 0;
            if (row < n && kk < k) x = transposed ? src[(long)kk * ld + row] : src[(long)row * ld + kk];
            v[e] = x;
        }
        const int kt = k0 / BK, kin = k0 - kt * BK;
        T* out = dst + ((((long)g * nkt + kt) * 64 + (p & 63)) * BK + kin);
        *reinterpret_cast<u32x4*>(out) = *reinterpret_cast<const u32x4*>(v);
    }
}
}  // namespace

extern "C" int dm_pack_rows(const void* src, long ld, int n, int k, int transposed, void* dst, int npad, int kpad,
                            int dtype, hipStream_t stream) {
    if (!src || !dst || n <= 0 || k <= 0 || npad % 64 || npad < n) return DM_ERR_ARG;
    const int esz = dtype == DM_F32 ? 4 : (dtype == DM_BF16 ? 2 : 0);
    if (!esz) return DM_ERR_ARG;
    const int bk = 8 * (16 / esz);
    if (kpad % bk || kpad < k) return DM_ERR_ARG;
    if (!dm_aligned16(dst)) return DM_ERR_ALIGN;
    const long total = (long)npad * (kpad / (16 / esz));
    long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (dtype == DM_F32)
        hipLaunchKernelGGL(pack_rows_kernel<float>, dim3((int)blocks), dim3(256), 0, stream, static_cast<const float*>(src),
                           ld, n, k, transposed, static_cast<float*>(dst), npad, kpad);
    else
        hipLaunchKernelGGL(pack_rows_kernel<bf16_t>, dim3((int)blocks), dim3(256), 0, stream,
                           static_cast<const bf16_t*>(src), ld, n, k, transposed, static_cast<bf16_t*>(dst), npad, kpad);
    return dm_check_launch();
}

extern "C" long dm_gemm_workspace_bytes(int M, int N, int splitk) {
    return splitk > 1 ? (long)splitk * M * N * (long)sizeof(float) : 0;
}

namespace {
// Arrival counters live in a FIXED area at the head of the work-list workspace, whatever the launch: launches of
// different shapes share one workspace, and a counter area sized per launch would let one launch's accumulator
// slabs land where a later, larger launch expects counters that read zero.
constexpr long SK_COUNTER_BYTES = 8L << 20;  // 2M output tiles
long streamk_counter_bytes(int tile, int M, int N) {
    int bm, bn;
    if (tile_dims(tile, &bm, &bn)) return -1;
    const long tiles = (long)((M + bm - 1) / bm) * ((N + bn - 1) / bn);
    return tiles * 4 > SK_COUNTER_BYTES ? -1 : SK_COUNTER_BYTES;
}
}  // namespace

extern "C" int dm_gemm_panel_supported(const DmGemmDesc* d, int tile, int nsplit) {
    return d ? dm_panel_supported(d, tile, nsplit) : 0;
}

extern "C" int dm_gemm_stats_supported(int tile, int rows_per_batch, int splitk) {
    int bm, bn;
    if (const int pr = dm_panel_rows(tile)) return rows_per_batch > 0 && rows_per_batch % (pr / 4) == 0;  // splitk = column slices
    if (tile_dims(tile, &bm, &bn) || rows_per_batch <= 0 || splitk > 1) return 0;
    return rows_per_batch % tile_wave_rows(tile) == 0;
}

extern "C" int dm_gemm_gn_supported(int dtype, int tile, int conv, int c, int groups, int rows_per_batch, int splitk) {
    int bm, bn;
    (void)conv;
    if (dtype != DM_F32 || !gn_tile(tile) || tile_dims(tile, &bm, &bn) || splitk > 1) return 0;
    if (c <= 0 || c > GN_FUSE_MAX_C || c % 32 || groups <= 0 || groups > 64 || c % groups) return 0;
    return rows_per_batch > 0 && rows_per_batch % bm == 0;
}

extern "C" long dm_gemm_streamk_workspace_bytes(int tile, int M, int N, int blocks) {
    int bm, bn;
    if (tile_dims(tile, &bm, &bn) || blocks <= 0) return -1;
    const long counters = streamk_counter_bytes(tile, M, N);
    if (counters < 0) return -1;  // more output tiles than the counter area holds
    return counters + 2L * blocks * bm * bn * (long)sizeof(float);
}

// Geometry the halo tile takes: a 128-row tile is G groups of R whole image rows inside one image (R = min(128 / W, H)),
// every group plus its one-pixel border fits the LDS image (204 slots: images up to 32 pixels wide), 32-bit source offsets
// reach the whole tensor.
extern "C" int dm_conv_halo_supported(int m, int h, int w, int cin, long lda) {
    if (m <= 0 || h <= 0 || w <= 0 || w > 32 || 128 % w || m % 128 || cin <= 0 || cin % 64) return 0;
    const int r = 128 / w < h ? 128 / w : h;
    const int rw = r * w;
    if (h % r || 128 % rw || m % (h * w)) return 0;
    if ((128 / rw) * (r + 2) * (w + 2) > 204) return 0;
    return (long)m * lda * 2 < (1L << 32);
}

extern "C" int dm_gemm(const DmGemmDesc* d, hipStream_t stream) {
    if (!d || !d->a || !d->w || !d->c) return DM_ERR_ARG;
    if (d->m <= 0 || d->n <= 0 || d->k <= 0) return DM_ERR_ARG;
    const int esz = d->dtype == DM_F32 ? 4 : (d->dtype == DM_BF16 ? 2 : 0);
    if (!esz) return DM_ERR_ARG;
    const int epc = 16 / esz, bk = 8 * epc;
    if (d->n % 4 || d->kpad % bk || d->kpad < d->k) return DM_ERR_ARG;
    if (!dm_aligned16(d->a) || !dm_aligned16(d->w) || (d->lda % epc)) return DM_ERR_ALIGN;
    // output / residual rows are accessed as 16-byte pieces
    if (d->epi != DM_EPI_NCHW_F32 && (!dm_aligned16(d->c) || d->ldc % epc)) return DM_ERR_ALIGN;
    if (d->r && (d->ldr % epc || !dm_aligned16(d->r))) return DM_ERR_ALIGN;
    if (d->bias && !dm_aligned16(d->bias)) return DM_ERR_ALIGN;
    if (d->rowbias && (!dm_aligned16(d->rowbias) || d->ld_rowbias % 4)) return DM_ERR_ALIGN;
    if (d->conv) {
        if (d->cin % epc || d->k != 9 * d->cin + (d->a2 ? d->cin2 : 0)) return DM_ERR_ARG;
        if (d->a2) {
            // appended pointwise operand: direct-to-LDS tiles on their aligned fast path, same-size convolution
            if (d->cin2 <= 0 || d->cin % bk || d->cin2 % bk || d->stride != 1 || d->up || d->lda2 % epc ||
                !dm_aligned16(d->a2))
                return DM_ERR_ARG;
        }
        if (d->stride != 1 && d->stride != 2) return DM_ERR_ARG;
        if (d->up && (d->stride != 1 || d->hout != 2 * d->hin || d->wout != 2 * d->win)) return DM_ERR_ARG;
        const int pad2 = d->pad_hi_only ? 1 : 2;  // total zero padding per axis
        if (d->pad_hi_only && (d->up || d->a2)) return DM_ERR_ARG;
        if (!d->up && (d->hout != (d->hin + pad2 - 3) / d->stride + 1 || d->wout != (d->win + pad2 - 3) / d->stride + 1))
            return DM_ERR_ARG;
        if (d->m % (d->hout * d->wout)) return DM_ERR_ARG;
    } else if (d->k % epc) {
        return DM_ERR_ARG;
    } else if (d->a2) {
        // dense second operand: whole k-tiles on both sides of the seam, no padded k tail
        if (d->cin2 <= 0 || d->cin2 >= d->k || d->cin2 % bk || (d->k - d->cin2) % bk || d->k != d->kpad ||
            d->lda2 % epc || !dm_aligned16(d->a2))
            return DM_ERR_ARG;
    }
    if (d->epi < DM_EPI_PLAIN || d->epi > DM_EPI_QUICKGELU) return DM_ERR_ARG;
    if (d->epi == DM_EPI_GEGLU && (!d->bias || d->n % 64 || d->rowbias || d->r)) return DM_ERR_ARG;
    if (d->epi == DM_EPI_QUICKGELU && d->r) return DM_ERR_ARG;  // activation of a sum with a residual is not a thing here
    if ((d->rowbias || d->epi == DM_EPI_NCHW_F32 || d->stats) && d->rows_per_batch <= 0) return DM_ERR_ARG;
    if (d->stats && (d->epi != DM_EPI_PLAIN || !dm_group_stats_ok(d->stats, d->ld_stats, d->stats_cpg, d->stats_ch0, d->n)))
        return DM_ERR_ARG;
    if (d->stats2 && (!d->stats || !dm_group_stats_ok(d->stats2, d->ld_stats2, d->stats2_cpg, d->stats2_ch0, d->n)))
        return DM_ERR_ARG;

    int tile = d->tile, splitk = d->splitk;
    if (tile == DM_TILE_AUTO) dm_gemm_plan(d->dtype, d->m, d->n, d->k, d->epi, &tile, &splitk);
    if (dm_panel_rows(tile)) {
        // row-panel form: `splitk` counts column slices; its own shape rules (panel.hip)
        if (d->ln_in && (!d->ln_colsum || !dm_aligned16(d->ln_colsum) || (reinterpret_cast<uintptr_t>(d->ln_in) & 7u)))
            return DM_ERR_ARG;
        if (d->ln_out && (d->epi != DM_EPI_PLAIN || (reinterpret_cast<uintptr_t>(d->ln_out) & 7u))) return DM_ERR_ARG;
        return dm_panel_launch(d, tile, splitk < 1 ? 1 : splitk, stream);
    }
    int bm, bn;
    if (tile_dims(tile, &bm, &bn)) return DM_ERR_ARG;
    if (d->a2 && tile < DM_TILE_D128x128_S2) return DM_ERR_ARG;  // only the direct-to-LDS loop reads a second operand
    if (d->conv_chunk_major) {
        // chunk-major k axis: bf16, direct-to-LDS loops only, row-linear addressing (stride 1, same size, padding 1)
        if (!d->conv || d->dtype != DM_BF16 || d->gn_sums || tile < DM_TILE_D128x128_S2 || d->w_split || d->cin % bk ||
            d->stride != 1 || d->up || d->pad_hi_only || d->hin != d->hout || d->win != d->wout)
            return DM_ERR_ARG;
        if (d->conv_chunk_major != 1 && d->conv_chunk_major != 2) return DM_ERR_ARG;
        if ((d->conv_chunk_major == 2) != (tile == DM_TILE_H128x128)) return DM_ERR_ARG;  // the halo tile and its weight order go together
        if (tile == DM_TILE_H128x128 && !dm_conv_halo_supported(d->m, d->hin, d->win, d->cin, d->lda)) return DM_ERR_ARG;
    } else if (tile == DM_TILE_H128x128) {
        return DM_ERR_ARG;
    }
    if (tile == DM_TILE_D256x256_S2) {
        // bf16, whole k-tiles only (conv: channels a multiple of the k-tile; dense: no padded k tail), no legacy split-K
        // convolutions: stride 1, same size (input pixel = output row + tap offset)
        if (d->dtype != DM_BF16 || splitk > 1 || (d->conv ? d->cin % bk != 0 : d->kpad != d->k)) return DM_ERR_ARG;
        if (d->conv && (d->stride != 1 || d->up || d->pad_hi_only || d->hin != d->hout || d->win != d->wout)) return DM_ERR_ARG;
    }
    // folded / produced LayerNorm statistics live in the epilogue: not with the separate split-K reduction
    if ((d->ln_in || d->ln_out) && splitk > 1) return DM_ERR_ARG;
    if (d->ln_in && (!d->ln_colsum || d->conv || !dm_aligned16(d->ln_colsum) || (reinterpret_cast<uintptr_t>(d->ln_in) & 7u)))
        return DM_ERR_ARG;
    if (d->ln_out && (d->epi != DM_EPI_PLAIN || (reinterpret_cast<uintptr_t>(d->ln_out) & 7u))) return DM_ERR_ARG;
    // fused statistics: a wave's rows must lie inside one sample, and the separate split-K reduction
    // launch does not produce them
    if (d->stats && (d->rows_per_batch % tile_wave_rows(tile) || d->m % d->rows_per_batch || splitk > 1)) return DM_ERR_ARG;
    if (d->w_split) {
        // split-bf16 weight image: fp32 problems on the plain direct-to-LDS ring loop
        if (d->dtype != DM_F32 || !split_tile(tile) || d->gn_sums || (d->w_split != 2 && d->w_split != 3)) return DM_ERR_ARG;
    }
    int gn_c = 0;
    if (d->gn_sums) {
        // GroupNorm of A inside the launch: every tile lies in one sample, whole k-tiles of normalised channels
        gn_c = d->conv ? d->cin : (d->a2 ? d->k - d->cin2 : d->k);
        if (!dm_gemm_gn_supported(d->dtype, tile, d->conv, gn_c, d->gn_groups, d->rows_per_batch, splitk)) return DM_ERR_ARG;
        if (!d->gn_gamma || !d->gn_beta || (reinterpret_cast<uintptr_t>(d->gn_sums) & 7u) || d->ld_gn < d->gn_groups ||
            d->m % d->rows_per_batch || d->ln_in)
            return DM_ERR_ARG;
        if (d->conv ? (d->stride != 1 || d->up || d->pad_hi_only || d->hin != d->hout || d->win != d->wout || !d->gn_silu ||
                       d->rows_per_batch != d->hout * d->wout)
                    : (d->gn_silu || d->kpad != d->k))
            return DM_ERR_ARG;
    }
    int streamk = 0;
    if (splitk < 0) {
        // work-list form: -splitk blocks share the (tile, k-tile) list; never more blocks than list entries
        const long units = (long)((d->m + bm - 1) / bm) * ((d->n + bn - 1) / bn) * (d->kpad / bk);
        streamk = (long)-splitk < units ? -splitk : (int)units;
        splitk = 1;
        const long need = dm_gemm_streamk_workspace_bytes(tile, d->m, d->n, streamk);
        if (!d->workspace || (reinterpret_cast<uintptr_t>(d->workspace) & 255u) || need < 0 || d->workspace_bytes < need)
            return DM_ERR_ARG;
    }
    if (splitk < 1) splitk = 1;
    if (splitk > 1 && (d->epi != DM_EPI_PLAIN || !d->workspace ||
                       d->workspace_bytes < dm_gemm_workspace_bytes(d->m, d->n, splitk)))
        return DM_ERR_ARG;
    // packed weights must cover every tile row that will be read
    if (d->npad % 64 || d->npad < ((d->n + bn - 1) / bn) * bn) return DM_ERR_ARG;  // 64-row groups of the blocked image

    GemmArgs a;
    a.A = static_cast<const char*>(d->a);
    a.lda = d->lda;
    a.W = static_cast<const char*>(d->w);
    a.C = static_cast<char*>(d->c);
    a.ldc = d->ldc;
    a.bias = d->bias;
    a.rowbias = d->rowbias;
    a.ld_rowbias = d->ld_rowbias;
    a.rows_per_batch = d->rows_per_batch > 0 ? d->rows_per_batch : 1;
    a.R = static_cast<const char*>(d->r);
    a.ldr = d->ldr;
    a.M = d->m; a.N = d->n; a.K = d->k; a.Kpad = d->kpad;
    a.conv = d->conv; a.Hin = d->hin; a.Win = d->win; a.Hout = d->hout; a.Wout = d->wout;
    a.Cin = d->cin; a.stride = d->stride; a.up = d->up;
    a.padlo = d->pad_hi_only ? 0 : 1;
    a.A2 = static_cast<const char*>(d->a2); a.lda2 = d->lda2; a.Cin2 = d->cin2;
    a.epi = d->epi;
    a.splitk = splitk;
    a.partial = static_cast<float*>(d->workspace);
    a.stats = dm_group_stats(d->stats, d->ld_stats, d->stats_cpg, d->stats_ch0, d->n);
    a.stats2 = dm_group_stats(d->stats2, d->ld_stats2, d->stats2_cpg, d->stats2_ch0, d->n);
    a.ln_in = d->ln_in;
    a.ln_colsum = d->ln_colsum;
    a.ln_eps = d->ln_eps;
    a.ln_out = d->ln_out;
    a.streamk = streamk;
    a.counters = static_cast<int*>(d->workspace);
    a.slabs = streamk ? reinterpret_cast<float*>(static_cast<char*>(d->workspace) + streamk_counter_bytes(tile, d->m, d->n))
                      : nullptr;
    a.gn_sums = d->gn_sums;
    a.ld_gn = d->ld_gn;
    a.gn_gamma = d->gn_gamma;
    a.gn_beta = d->gn_beta;
    a.gn_eps = d->gn_eps;
    a.gn_groups = d->gn_groups;
    a.gn_c = gn_c;
    a.korder = d->conv_chunk_major;

    int rc = d->gn_sums ? launch_gn_typed(a, tile, stream)
             : d->w_split ? launch_split_typed(a, tile, d->w_split, stream)
                          : (d->dtype == DM_F32 ? launch_typed<float>(a, tile, stream) : launch_typed<bf16_t>(a, tile, stream));
    if (rc != DM_OK) return rc;
    if (splitk > 1) {
        const long total = (long)a.M * (a.N >> 2);
        int blocks = (int)((total + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        if (d->dtype == DM_F32)
            hipLaunchKernelGGL(splitk_reduce_kernel<float>, dim3(blocks), dim3(256), 0, stream, a);
        else
            hipLaunchKernelGGL(splitk_reduce_kernel<bf16_t>, dim3(blocks), dim3(256), 0, stream, a);
        rc = dm_check_launch();
    }
    return rc;
}

#ifdef DM_TIMELINE
extern "C" int dm_timeline_variant(int bits) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_tl_variant), &bits, sizeof(int)) == hipSuccess ? DM_OK : DM_ERR_LAUNCH;
}
// Diagnostic build: copies the phase stamps of the last launch to the host and clears them.
extern "C" int dm_timeline_read(unsigned long long* host, int blocks) {
    if (!host || blocks <= 0 || blocks > TL_BLOCKS) return DM_ERR_ARG;
    if (hipDeviceSynchronize() != hipSuccess) return DM_ERR_LAUNCH;
    const size_t bytes = (size_t)blocks * TL_SLOTS * sizeof(unsigned long long);
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_timeline), bytes) != hipSuccess) return DM_ERR_LAUNCH;
    void* dev = nullptr;
    if (hipGetSymbolAddress(&dev, HIP_SYMBOL(g_timeline)) != hipSuccess) return DM_ERR_LAUNCH;
    return hipMemset(dev, 0, sizeof(unsigned long long) * TL_BLOCKS * TL_SLOTS) == hipSuccess ? DM_OK : DM_ERR_LAUNCH;
}
#endif
