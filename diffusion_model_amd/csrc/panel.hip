// Row-panel GEMM for short contractions on gfx950 (bf16 operands, fp32 accumulation).
//
//   C[m][n] = epilogue( sum_k A[m][k] * W[n][k] ),  K = 320 or 640 (whole k-tiles of 64), N a multiple of 64
//
// The attention projections, the GEGLU projection and the 1x1 convolutions at the 64 x 64 and 32 x 32 levels of the
// U-Net have K = C = 320 / 640: five or ten k-tiles.  A kernel that gives every block one output tile pays its fixed
// costs (pointer set-up, the first operand latency, a drained pipeline, an epilogue that nothing overlaps) once per
// five k-tiles and fetches the activation rows once per column tile.  Here a block owns 64 * FM ROWS and walks ALL
// column tiles of them (or the slice of column tiles that grid.y assigns to it):
//   * the block's A rows are read from memory once, straight into registers in MFMA fragment layout, and stay there
//     for the whole launch (wave w holds rows 16*FM*w .. 16*FM*(w+1) for all K: FM * K/32 fragments of 4 registers);
//   * the weights stream through an LDS ring of 8 KiB slots (one k-tile of one 64-column group) by LDS-DMA.  In the
//     blocked weight image [N/64][K/64][64 rows][128 B] the slots a block needs are ONE contiguous byte stream, so
//     the stream never stops at a column-tile boundary: while the waves run the epilogue of column tile j, the first
//     k-tiles of tile j+1 are already landing;
//   * per k-tile a wave reads only W fragments from LDS (each feeds FM MFMAs): half the LDS bytes of a tile that
//     also reads A from LDS, and no A bytes through the vector-memory path after the prologue;
//   * two blocks fit a CU (registers: 3 waves per SIMD for 2 x 5 waves; LDS: 48-64 KiB each), so one block's epilogue
//     (bias, GELU, rounding, stores) overlaps the other's MFMA phase without any hand-built schedule;
//   * a block sees whole output rows: the row sums a following (folded) LayerNorm needs are accumulated in registers
//     across the column tiles and written once, without atomics.
// Every wave requests its share of each slot (two LDS-DMA instructions per wave and slot) and counts its own requests with
// s_waitcnt vmcnt(N): STAGES-1 slots stay in flight, one raw s_barrier per k-tile publishes slot t and retires the reads of slot
// t-1.  vmcnt retires in issue order over loads, stores, atomics and LDS-DMA alike and the count assumes that only slot requests
// are younger than the slot waited for, so behind an epilogue (whose stores are younger too) a wait is stricter than needed for a
// few k-tiles: it also waits for those stores.  That errs on the safe side, never the other way.
#include "common.h"
#include "dm_hip.h"
#include "panel.h"

namespace {

struct PanelArgs {
    const char* A;
    long lda;  // elements
    const char* W;
    char* C;
    long ldc;
    const float* bias;
    const char* R;
    long ldr;
    int M, N;
    int epi;
    int ntn, per_split;      // column tiles of 64 in all / per grid.y slice
    const double* ln_in;     // per-row (sum, sumsq) of A: the LayerNorm in front is folded into W / bias
    const float* ln_colsum;  // sum over k of each packed weight row
    float ln_eps;
    double* ln_out;          // per-row (sum, sumsq) of the stored output
    int ln_out_atomic;       // grid.y > 1: several blocks contribute to a row
    DmGroupStats stats, stats2;  // per-group (sum, sumsq) of the stored output (common.h)
    int rows_per_batch;
    // GroupNorm (no activation) of the A operand applied to the block's rows as they are loaded: A is the INPUT of the
    // GroupNorm, gn_sums its per-group (sum, sumsq) [batch][2][ld_gn] from the launch that wrote it
    const double* gn_sums;
    long ld_gn;
    const float* gn_gamma;
    const float* gn_beta;
    float gn_eps;
    int gn_groups;
};

__device__ __forceinline__ int swz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// at most min(young, MAXY) slot requests of LOADS instructions each may still be outstanding
template <int LOADS, int MAXY> __device__ __forceinline__ void wait_young(int young) {
    if constexpr (MAXY == 0) {
        wait_vm<0>();
    } else {
        if (young >= MAXY) wait_vm<MAXY * LOADS>();
        else wait_young<LOADS, MAXY - 1>(young);
    }
}

#ifndef PANEL_STAGES
#define PANEL_STAGES 6
#endif
constexpr int SLOT = 64 * 128;  // bytes: one k-tile (64 bf16) of one 64-row group of the weight image

template <int FM, int KT, int STAGES>
__global__ __launch_bounds__(256, 2) void gemm_panel_kernel(PanelArgs p) {
    static_assert(FM * KT <= 10, "the A fragments (FM * KT * 8 registers) must leave room for two waves per SIMD");
    static_assert(STAGES >= 3 && (STAGES - 2) * 2 <= 63, "ring depth / vmcnt field");
    constexpr int BM = 64 * FM;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int m0 = blockIdx.x * BM;
    const int jt0 = blockIdx.y * p.per_split;
    const int jt1 = min(p.ntn, jt0 + p.per_split);
    const int T = (jt1 - jt0) * KT;  // slots this block consumes
    if (T <= 0) return;
    const int mrow0 = m0 + wave * 16 * FM;

    // ---- weight stream: slot t of this block = bytes [t * SLOT, (t + 1) * SLOT) behind the first column tile's image.
    // Thread tid lands on LDS row it*32 + tid/8, physical chunk tid&7 of the slot; the XOR swizzle that makes the
    // fragment reads conflict-free is applied to the SOURCE chunk (the DMA writes LDS lane-linearly).
    const int lchunk = (tid & 7) ^ ((tid >> 4) & 7);
    const char* wsrc = p.W + (long)jt0 * KT * SLOT + (tid >> 3) * 128 + lchunk * 16;
    char* const ldst = smem + wave * 1024;
    auto issue = [&](int stage) {
#pragma unroll
        for (int it = 0; it < 2; ++it)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + it * 4096),
                                             (__attribute__((address_space(3))) void*)(ldst + stage * SLOT + it * 4096), 16, 0, 0);
        wsrc += SLOT;
    };
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (s < T) issue(s);

    // ---- column-only epilogue terms of all this block's column tiles, staged in LDS behind the ring in the order the
    // epilogue reads them: per column tile 64 bias floats [fq][16] (GEGLU: 8 value biases, then their 8 gate biases)
    // and 64 column sums [fq][j][4] of the folded LayerNorm.  A global load inside the epilogue would expose its whole
    // latency once per column tile, and the compiler drains the LDS-DMA ring in front of it.
    float* const tab = reinterpret_cast<float*>(smem + STAGES * SLOT);
    {
        const int ncols = (jt1 - jt0) * 64;
        const int half = p.N >> 1;
        for (int idx = tid; idx < ncols; idx += 256) {
            const int jl = idx >> 6, q = (idx >> 4) & 3, e = idx & 15;
            const int n0t = (jt0 + jl) * 64;
            float bv = 0.f, cv = 0.f;
            if (p.bias) {
                if (p.epi == DM_EPI_GEGLU) bv = p.bias[(e < 8 ? 0 : half) + (n0t >> 1) + q * 8 + (e & 7)];
                else bv = p.bias[n0t + q * 16 + e];
            }
            if (p.ln_in) cv = p.ln_colsum[n0t + 16 * (e >> 2) + 4 * q + (e & 3)];
            tab[jl * 128 + q * 16 + e] = bv;
            tab[jl * 128 + 64 + q * 16 + e] = cv;
        }
    }

    // ---- GroupNorm of the A operand: per-channel scale = rstd * gamma and shift = beta - mean * scale of the block's sample
    // (all rows of a block lie in one sample: checked on the host), in LDS behind the column-term table
    float* const gn_tab = tab + (jt1 - jt0) * 128;  // scale[K] | shift[K] | group mean[64] | group rstd[64]
    if (p.gn_sums) {
        constexpr int KC = KT * 64;
        float* g_mean = gn_tab + 2 * KC;
        float* g_rstd = g_mean + 64;
        const int cpg = KC / p.gn_groups;
        const double* src = p.gn_sums + (long)(m0 / p.rows_per_batch) * 2 * p.ld_gn;
        if (tid < p.gn_groups)
            group_mean_rstd(src[tid], src[p.ld_gn + tid], (double)p.rows_per_batch * cpg, p.gn_eps, g_mean[tid], g_rstd[tid]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        for (int c = tid; c < KC; c += 256) {
            const int grp = c / cpg;
            const float sc = g_rstd[grp] * p.gn_gamma[c];
            gn_tab[c] = sc;
            gn_tab[KC + c] = fmaf(-g_mean[grp], sc, p.gn_beta[c]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }

    // ---- the block's A rows, once, into registers: fragment (i, ks) of lane (fr, fq) = row mrow0 + 16 i + fr,
    // k = 32 ks + 8 fq .. + 8 (16 bytes)
    u32x4 afr[FM][2 * KT];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int m = mrow0 + i * 16 + fr;
        const bool ok = m < p.M;
        const char* ap = p.A + ((long)(ok ? m : 0) * p.lda) * 2 + fq * 16;
#pragma unroll
        for (int ks = 0; ks < 2 * KT; ++ks) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(ap + ks * 64);
            afr[i][ks] = ok ? v : u32x4{0u, 0u, 0u, 0u};
        }
    }
    if (p.gn_sums) {
        // the map runs once per element of A: the rows stay in registers for every column tile
#pragma unroll
        for (int ks = 0; ks < 2 * KT; ++ks) {
            const float* sc = gn_tab + ks * 32 + fq * 8;
            const f32x4 s0 = *reinterpret_cast<const f32x4*>(sc), s1 = *reinterpret_cast<const f32x4*>(sc + 4);
            const f32x4 t0 = *reinterpret_cast<const f32x4*>(sc + KT * 64), t1 = *reinterpret_cast<const f32x4*>(sc + KT * 64 + 4);
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                float f[8];
                chunk_to_f32<bf16_t>(afr[i][ks], f);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    f[e] = fmaf(f[e], s0[e], t0[e]);
                    f[4 + e] = fmaf(f[4 + e], s1[e], t1[e]);
                }
                afr[i][ks] = f32_to_chunk<bf16_t>(f);
            }
        }
    }
    // folded LayerNorm in front: mean and 1/sqrt(var + eps) of this lane's rows (difference of moments in fp64)
    float ln_mu[FM], ln_rs[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int m = mrow0 + i * 16 + fr;
        ln_mu[i] = 0.f;
        ln_rs[i] = 0.f;
        if (p.ln_in && m < p.M) {
            const double inv_k = 1.0 / (double)(KT * 64);
            const double mean = p.ln_in[2L * m] * inv_k;
            const float var = fmaxf((float)(p.ln_in[2L * m + 1] * inv_k - mean * mean), 0.f);
            ln_mu[i] = (float)mean;
            ln_rs[i] = rsqrtf(var + p.ln_eps);
        }
    }
    float row_s[FM], row_q[FM];  // sums over the stored row (for a following LayerNorm), this lane's columns
#pragma unroll
    for (int i = 0; i < FM; ++i) row_s[i] = row_q[i] = 0.f;

    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the table writes above; the first barrier below publishes them
    int stage = 0, t = 0;
    for (int jt = jt0; jt < jt1; ++jt) {
        f32x4 acc[FM][4];
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
        for (int kt = 0; kt < KT; ++kt, ++t) {
            // slot t has landed once only the younger requests (at most STAGES-2 slots) are outstanding; the barrier makes
            // every wave's share visible and retires all reads of the slot refilled below
            wait_young<2, STAGES - 2>(T - 1 - t);
            __builtin_amdgcn_s_barrier();
            if (t + STAGES - 1 < T) {
                int ns = stage + STAGES - 1;
                if (ns >= STAGES) ns -= STAGES;
                issue(ns);
            }
            const char* sw = smem + stage * SLOT;
            u32x4 wf[2][4];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < 4; ++j) wf[ks][j] = *reinterpret_cast<const u32x4*>(sw + swz(j * 16 + fr, ks * 4 + fq));
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) Mma16<bf16_t>::mma(acc[i][j], wf[ks][j], afr[i][2 * kt + ks]);
            if (++stage == STAGES) stage = 0;
        }

        // ---- epilogue of column tile jt: lane holds rows mrow0 + 16 i + fr, columns n0 + 16 fq + 4 j + r
        const int n0 = jt * 64;
        const float* const tb = tab + (jt - jt0) * 128 + fq * 16;  // this lane's 16 bias values; + 64: its 16 column sums
        if (p.ln_in) {
            // LN(a) W^T = rstd * (a W'^T - mean * colsum(W')) + (beta W^T + b); colsum is in packed row order
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 cs = *reinterpret_cast<const f32x4*>(tb + 64 + 4 * j);
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][j][r] = ln_rs[i] * (acc[i][j][r] - ln_mu[i] * cs[r]);
            }
        }
        if (p.epi == DM_EPI_GEGLU) {
            // lane holds 8 value columns nb .. nb+7 (j = 0, 1) and their gates (j = 2, 3)
            const int nb = (n0 >> 1) + fq * 8;
            f32x4 bv[2], bg[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                bv[j] = *reinterpret_cast<const f32x4*>(tb + 4 * j);
                bg[j] = *reinterpret_cast<const f32x4*>(tb + 8 + 4 * j);
            }
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int m = mrow0 + i * 16 + fr;
                if (m >= p.M) continue;
                float o[8];
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 4; r += 2) {
                        // two values per packed instruction
                        const f32x2 g = gelu_erf_fast_f2(f32x2{acc[i][j + 2][r], acc[i][j + 2][r + 1]} + f32x2{bg[j][r], bg[j][r + 1]});
                        const f32x2 v = (f32x2{acc[i][j][r], acc[i][j][r + 1]} + f32x2{bv[j][r], bv[j][r + 1]}) * g;
                        o[4 * j + r] = v[0];
                        o[4 * j + r + 1] = v[1];
                    }
                bf16_t* dst = reinterpret_cast<bf16_t*>(p.C) + (long)m * p.ldc + nb;
                *reinterpret_cast<u32x4*>(dst) = u32x4{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]),
                                                       pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7])};
            }
            continue;
        }
        const int ncol0 = n0 + fq * 16;
        f32x4 b4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            b4[j] = *reinterpret_cast<const f32x4*>(tb + 4 * j);
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int m = mrow0 + i * 16 + fr;
            if (m >= p.M) continue;
            float o[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) o[q] = acc[i][q >> 2][q & 3] + b4[q >> 2][q & 3];
            if (p.R) {
                const bf16_t* rp = reinterpret_cast<const bf16_t*>(p.R) + (long)m * p.ldr + ncol0;
                const u32x4 r0 = *reinterpret_cast<const u32x4*>(rp);
                const u32x4 r1 = *reinterpret_cast<const u32x4*>(rp + 8);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    o[2 * q] += __uint_as_float(r0[q] << 16);
                    o[2 * q + 1] += __uint_as_float(r0[q] & 0xffff0000u);
                    o[8 + 2 * q] += __uint_as_float(r1[q] << 16);
                    o[8 + 2 * q + 1] += __uint_as_float(r1[q] & 0xffff0000u);
                }
            }
            u32x4 pk[2];
#pragma unroll
            for (int q = 0; q < 8; ++q) pk[q >> 2][q & 3] = pack_bf16x2(o[2 * q], o[2 * q + 1]);
            if (p.ln_out || p.stats.sums) {
                // statistics are taken of the values as stored (after rounding)
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float v0 = __uint_as_float(pk[q >> 2][q & 3] << 16);
                    const float v1 = __uint_as_float(pk[q >> 2][q & 3] & 0xffff0000u);
                    if (p.ln_out) {
                        row_s[i] += v0 + v1;
                        row_q[i] = fmaf(v1, v1, fmaf(v0, v0, row_q[i]));
                    }
                    acc[i][(2 * q) >> 2][(2 * q) & 3] = v0;
                    acc[i][(2 * q + 1) >> 2][(2 * q + 1) & 3] = v1;
                }
            }
            bf16_t* dst = reinterpret_cast<bf16_t*>(p.C) + (long)m * p.ldc + ncol0;
            *reinterpret_cast<u32x4*>(dst) = pk[0];
            *reinterpret_cast<u32x4*>(dst + 8) = pk[1];
        }
        if (p.stats.sums && mrow0 < p.M) {
            // GroupNorm statistics of the tensor just written: per-channel sums over this wave's rows (all of one sample and
            // all valid: checked on the host).  Four columns at a time: reduce-scatter over lane bits 0-1, all-reduce over
            // bits 2-3, then the lanes with fr>>2 == j own column 4 j + (fr & 3) = fr, i.e. lane l owns column n0 + l;
            // group_stats_add folds them to groups and adds the groups' shares to memory.
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
                    for (int hf = 2; hf >= 1; hf >>= 1) {
                        const bool upper = (fr & hf) != 0;
#pragma unroll
                        for (int q = 0; q < hf; ++q) {
                            const float keep = upper ? v[w][q + hf] : v[w][q];
                            const float send = upper ? v[w][q] : v[w][q + hf];
                            v[w][q] = keep + __shfl_xor(send, hf);
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
            const long b = mrow0 / p.rows_per_batch;
            group_stats_add(p.stats, b, n0, lane, own_s, own_q);
            if (p.stats2.sums) group_stats_add(p.stats2, b, n0, lane, own_s, own_q);
        }
    }
    if (p.ln_out) {
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            // the four lanes that share a row (one per 16-column group) add up, then one of them publishes
            float s = row_s[i], q = row_q[i];
            s += __shfl_xor(s, 16);
            q += __shfl_xor(q, 16);
            s += __shfl_xor(s, 32);
            q += __shfl_xor(q, 32);
            const int m = mrow0 + i * 16 + fr;
            if (fq == 0 && m < p.M) {
                if (p.ln_out_atomic) {
                    atomicAdd(p.ln_out + 2L * m, (double)s);
                    atomicAdd(p.ln_out + 2L * m + 1, (double)q);
                } else {
                    // this block wrote the whole row: the sums are final (the caller's buffer holds zeros)
                    p.ln_out[2L * m] = (double)s;
                    p.ln_out[2L * m + 1] = (double)q;
                }
            }
        }
    }
}

template <int FM, int KT, int STAGES> int launch(const PanelArgs& a, int nsplit, hipStream_t stream) {
    static DmLdsAttr attr;
    auto kfn = gemm_panel_kernel<FM, KT, STAGES>;
    const int lds = STAGES * SLOT + a.per_split * 512 + (a.gn_sums ? (2 * KT * 64 + 128) * 4 : 0);  // ring + column terms + GroupNorm table
    if (lds > 160 * 1024) return DM_ERR_ARG;
    if (const int rc = dm_set_lds_attr(reinterpret_cast<const void*>(kfn), attr, 160 * 1024)) return rc;
    const dim3 grid((a.M + 64 * FM - 1) / (64 * FM), nsplit, 1);
    hipLaunchKernelGGL(kfn, grid, dim3(256), lds, stream, a);
    return dm_check_launch();
}

}  // namespace

int dm_panel_rows(int tile) {
    switch (tile) {
        case DM_TILE_P64: return 64;
        case DM_TILE_P128: return 128;
        default: return 0;
    }
}

int dm_panel_supported(const DmGemmDesc* d, int tile, int nsplit) {
    const int bm = dm_panel_rows(tile);
    if (!bm || d->dtype != DM_BF16 || d->conv || d->a2 || d->rowbias || d->w_split) return 0;
    if (d->gn_sums) {
        // GroupNorm of A inside the launch: affine only (no SiLU), every block inside one sample
        if (d->gn_silu || d->ln_in || !d->gn_gamma || !d->gn_beta || (reinterpret_cast<uintptr_t>(d->gn_sums) & 7u) ||
            d->gn_groups <= 0 || d->gn_groups > 64 || d->k % d->gn_groups || d->ld_gn < d->gn_groups ||
            d->rows_per_batch <= 0 || d->rows_per_batch % bm || d->m % d->rows_per_batch)
            return 0;
    }
    if (d->k != d->kpad || (d->k != 320 && d->k != 640)) return 0;
    if (tile == DM_TILE_P128 && d->k != 320) return 0;  // 32 rows per wave: the fragments of K = 640 do not fit the registers
    if (d->n % 64 || (d->epi != DM_EPI_PLAIN && d->epi != DM_EPI_GEGLU)) return 0;
    if (d->epi == DM_EPI_GEGLU && (!d->bias || d->r || d->ln_out || d->stats)) return 0;
    const int ntn = d->n / 64;
    if (nsplit < 1 || nsplit > ntn || d->npad < d->n) return 0;
    // fused statistics: a wave's rows lie inside one sample
    if (d->stats && (d->rows_per_batch <= 0 || d->rows_per_batch % (bm / 4) || d->m % d->rows_per_batch)) return 0;
    return 1;
}

int dm_panel_launch(const DmGemmDesc* d, int tile, int nsplit, hipStream_t stream) {
    if (!dm_panel_supported(d, tile, nsplit)) return DM_ERR_ARG;
    PanelArgs a;
    a.A = static_cast<const char*>(d->a);
    a.lda = d->lda;
    a.W = static_cast<const char*>(d->w);
    a.C = static_cast<char*>(d->c);
    a.ldc = d->ldc;
    a.bias = d->bias;
    a.R = static_cast<const char*>(d->r);
    a.ldr = d->ldr;
    a.M = d->m;
    a.N = d->n;
    a.epi = d->epi;
    a.ntn = d->n / 64;
    a.per_split = (a.ntn + nsplit - 1) / nsplit;
    nsplit = (a.ntn + a.per_split - 1) / a.per_split;  // no empty slices
    a.ln_in = d->ln_in;
    a.ln_colsum = d->ln_colsum;
    a.ln_eps = d->ln_eps;
    a.ln_out = d->ln_out;
    a.ln_out_atomic = nsplit > 1;
    a.stats = dm_group_stats(d->stats, d->ld_stats, d->stats_cpg, d->stats_ch0, d->n);
    a.stats2 = dm_group_stats(d->stats2, d->ld_stats2, d->stats2_cpg, d->stats2_ch0, d->n);
    a.rows_per_batch = d->rows_per_batch > 0 ? d->rows_per_batch : 1;
    a.gn_sums = d->gn_sums;
    a.ld_gn = d->ld_gn;
    a.gn_gamma = d->gn_gamma;
    a.gn_beta = d->gn_beta;
    a.gn_eps = d->gn_eps;
    a.gn_groups = d->gn_groups;
    const bool k5 = d->k == 320;
    switch (tile) {
        case DM_TILE_P64: return k5 ? launch<1, 5, PANEL_STAGES>(a, nsplit, stream) : launch<1, 10, PANEL_STAGES>(a, nsplit, stream);
        case DM_TILE_P128: return launch<2, 5, PANEL_STAGES>(a, nsplit, stream);
        default: return DM_ERR_ARG;
    }
}
