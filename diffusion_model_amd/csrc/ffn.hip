// Feed-forward half of an AttentionBlock as ONE launch on gfx950 (bf16 operands, fp32 accumulation):
//
//   out = conv_output( linear_geglu_2( v * gelu(g) ) + t ) + x,   [v | g] = linear_geglu_1( LayerNorm_3(t) )
//   (/root/reference/src/models/blocks.py:137-153)
//
// Both weight images are the ones the two-launch form uses: W1 = linear_geglu_1 with the LayerNorm folded in and value /
// gate rows interleaved, W2 = [W_c W_2 | W_c] over the row [v * gelu(g) | t] (conv_output multiplied into linear_geglu_2,
// its residue term appended), bias W_c b_2 + b_c.  In the two-launch form the 4C-wide product v * gelu(g) makes a round
// trip through memory (168 MB written and read again per block at batch 16, C = 320), and both launches are bound by that
// traffic.  Here the product never leaves the CU:
//   * a block owns 128 rows; wave (pr, h) of its 8 waves holds the rows 32 pr .. 32 pr + 31 of t in registers for the whole
//     launch, as MFMA fragments (C = 320: 2 x 10 fragments), and accumulates HALF of the block's 128 x C output tile for them:
//     the 16-column tiles 2h, 2h + 1 of every 64-column group (2 x 10 accumulator tiles);
//   * the weights stream through LDS in 8 KiB slots (one k-tile of one 64-row group of an image) by LDS-DMA, in the order of
//     use (SlotOrder below).  Every slot is four fragment reads and eight MFMAs per wave, whatever it belongs to, and each
//     fragment read feeds two MFMAs.  The stream is cut into PHASES of KT = 5 slots - the k-tiles of one column tile of the
//     first product, or the C / 64 row groups of one k-tile of the second - and the ring is two phases long.  A phase passes
//     two barriers (slots 0-2, slots 3-4) instead of one per slot: a wave can issue one instruction every four cycles, two
//     waves share a SIMD, and at eight MFMAs per barrier the waits, scalar bookkeeping and slot requests around them, not
//     the matrix pipe, set the pace;
//   * of a 64-row slot of W1 (16 value columns and their 16 gates per 16-row tile pair) wave half h takes value tile h and
//     gate tile 2 + h, so that every lane holds values and gates of the same four product columns; bias, folded LayerNorm,
//     erf-GELU and the product happen in registers and give, per lane, four bf16 products of columns 32 jt + 8 fq + 4 h + r.
//     This epilogue is cut into four quarters that run one per slot behind the MFMAs of the NEXT phase, on accumulators those
//     MFMAs do not touch, so the VALU work overlaps the matrix pipe instead of waiting for it (the second product of a pair of
//     column tiles therefore runs one pair late);
//   * the second product needs all eight products 32 jt + 8 fq + 0..7 of a lane group as its k operand: the two halves of a
//     wave pair swap their four through LDS (8 bytes per lane and row tile), which is the only exchange between waves;
//   * the main loop contains no vector-memory instruction other than the slot requests, so the counted s_waitcnt vmcnt(N) on
//     them is exact and nothing the block stores ever delays an operand; it contains no branch either (slot addresses come
//     from a constant table by scalar loads one phase ahead).
// The epilogue adds bias and the long residual x, stores bf16, and reduces the per-column sums of the stored values over the
// block's rows (in-wave butterfly, then a fixed-order sum over the four row groups through LDS) to the per-group statistics
// the next GroupNorm reads.
#include <type_traits>

#include "common.h"
#include "dm_hip.h"

namespace {

template <int I, int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

struct FfnArgs {
    const char* A;
    long lda;
    const char* W1;
    const float* bias1;
    const double* ln_in;
    const float* ln_colsum;
    float ln_eps;
    const char* W2;
    const float* bias2;
    const char* R;
    long ldr;
    char* C;
    long ldc;
    int M;
    DmGroupStats stats, stats2;
    int rows_per_batch;
};

constexpr int SLOT = 64 * 128;  // bytes: one k-tile (64 bf16) of one 64-row group of a weight image

__device__ __forceinline__ int swz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Order of the weight slots, in phases of KT: W1 of pair 0 (column tiles 0 and 1: two phases); for q = 1 .. NP-1 the two
// phases of W1 of pair q (column tiles 2q, 2q + 1), then the phase of k-tile q - 1 of W2 (one slot per 64-row group); the
// phase of k-tile NP - 1 of W2; the KT phases of the k-tiles of W2 that multiply t.  An entry is the slot's index inside its
// blocked image [rows / 64][k-tiles][64][128 B]; bit 30 marks the second image.
template <int KT> struct SlotOrder {
    static constexpr int NP = 4 * KT, G2 = KT, NKT2 = 5 * KT, T = NP * 3 * KT + KT * G2;
    struct Tab {
        int v[T + 8];  // padded: the entries of a phase's requests are read as one window of KT, also near the end
    };
    static constexpr Tab make() {
        Tab t{};
        int n = 0;
        for (int q = 0; q < NP; ++q) {
            for (int r = 0; r < 2 * KT; ++r) t.v[n++] = q * 2 * KT + r;
            if (q > 0)
                for (int g = 0; g < G2; ++g) t.v[n++] = (1 << 30) | (g * NKT2 + q - 1);
        }
        for (int g = 0; g < G2; ++g) t.v[n++] = (1 << 30) | (g * NKT2 + NP - 1);
        for (int kt = 0; kt < KT; ++kt)
            for (int g = 0; g < G2; ++g) t.v[n++] = (1 << 30) | (g * NKT2 + 4 * KT + kt);
        return t;
    }
    __device__ static const int* table();
};
__device__ const SlotOrder<5>::Tab g_slot_order_5 = SlotOrder<5>::make();
template <> __device__ __forceinline__ const int* SlotOrder<5>::table() { return g_slot_order_5.v; }

template <int KT>
__global__ __launch_bounds__(512, 2) void ffn_geglu_kernel(FfnArgs p) {
    static_assert(KT == 5, "the two barrier groups of a phase are slots 0-2 and 3-4");
    constexpr int C = KT * 64;            // width of the residual stream
    constexpr int NJT = 8 * KT;           // 64-row column tiles of W1 (8C packed rows: 4C values interleaved with 4C gates)
    constexpr int NP = NJT / 2;           // pairs of them = k-tiles of the product part of W2
    constexpr int G2 = KT;                // 64-column groups of the output
    constexpr int T = SlotOrder<KT>::T;   // slots in all
    constexpr int NPH = T / KT;           // phases
    constexpr int HALF = KT * SLOT;       // bytes of one half of the ring = one phase
    using True = std::true_type;
    using False = std::false_type;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pr = wave & 3, h = wave >> 2;
    const int fr = lane & 15, fq = lane >> 4;
    const int m0 = blockIdx.x * 128;
    const int mrow0 = m0 + pr * 32;

    // ---- weight stream.  Thread tid lands on LDS row tid / 8, physical chunk tid & 7 of a slot; the XOR swizzle that makes
    // the fragment reads conflict-free is applied to the SOURCE chunk (the DMA writes LDS lane-linearly).
    const int my_off = (tid >> 3) * 128 + (((tid & 7) ^ ((tid >> 4) & 7)) << 4);
    char* const ldst = smem + wave * 1024;
    const int* const order = SlotOrder<KT>::table();
    auto issue = [&](int entry, int lds_off) __attribute__((always_inline)) {
        const char* base = (entry & (1 << 30)) ? p.W2 : p.W1;
        base += (long)(entry & 0xffffff) * SLOT;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (unsigned)my_off),
                                         (__attribute__((address_space(3))) void*)(ldst + lds_off), 16, 0, 0);
    };
    // phase 0 whole and the first barrier group of phase 1
#pragma unroll
    for (int s = 0; s < KT + 3; ++s) issue(order[s], s * SLOT);

    // ---- column terms of the first product, behind the ring, in the order its epilogue reads them: per column tile 64 bias
    // floats [fq][8 value biases | their 8 gate biases] and the 64 column sums [fq][j][4] of the folded LayerNorm
    float* const tab = reinterpret_cast<float*>(smem + 2 * HALF);
    for (int idx = tid; idx < NJT * 64; idx += 512) {
        const int jl = idx >> 6, q = (idx >> 4) & 3, e = idx & 15;
        const int n0t = jl * 64;
        tab[jl * 128 + q * 16 + e] = p.bias1[(e < 8 ? 0 : NJT * 32) + (n0t >> 1) + q * 8 + (e & 7)];
        tab[jl * 128 + 64 + q * 16 + e] = p.ln_colsum[n0t + 16 * (e >> 2) + 4 * q + (e & 3)];
    }
    // exchange buffers of the product halves, one per pair parity: [row group][column-tile parity][row tile][lane][16 B], wave
    // half h owns bytes 8h .. 8h + 7
    constexpr int PBUF = 4 * 2 * 2 * 64 * 16;
    char* const pbuf = reinterpret_cast<char*>(tab + NJT * 128);

    // ---- the wave's rows of t, once, into registers: fragment (i, ks) of lane (fr, fq) = row mrow0 + 16 i + fr, k = 32 ks + 8 fq .. + 8
    u32x4 afr[2][2 * KT];
    float ln_mu[2], ln_rs[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = mrow0 + i * 16 + fr;
        const char* ap = p.A + ((long)m * p.lda) * 2 + fq * 16;
#pragma unroll
        for (int ks = 0; ks < 2 * KT; ++ks) afr[i][ks] = *reinterpret_cast<const u32x4*>(ap + ks * 64);
        // folded LayerNorm: mean and 1 / sqrt(var + eps) of the row (difference of moments in fp64)
        const double inv_k = 1.0 / (double)C;
        const double mean = p.ln_in[2L * m] * inv_k;
        const float var = fmaxf((float)(p.ln_in[2L * m + 1] * inv_k - mean * mean), 0.f);
        ln_mu[i] = (float)mean;
        ln_rs[i] = rsqrtf(var + p.ln_eps);
    }

    f32x4 oacc[G2][2][2];  // [64-column group][16-column tile 2h + jj][row tile]
#pragma unroll
    for (int g = 0; g < G2; ++g)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int i = 0; i < 2; ++i) oacc[g][jj][i] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- one phase.  Its slots sit in half (ph & 1) of the ring.  Requests in flight, in order, when its first barrier
    // group starts: the phase's five slots and slots 0-2 of the next phase; slots 0-2 have landed once at most Y0 = 5 requests
    // are outstanding.  Behind that barrier every wave is done with the previous phase, whose slots 3-4 (other half) are
    // refilled with slots 3-4 of the next phase; behind the second barrier (slots 3-4 landed: again at most 5 outstanding)
    // slots 0-2 of this half are refilled with those of the phase after next.  Each barrier follows a wait for the wave's
    // own LDS reads and publishes LDS writes made before it.  e[]: table entries of the five slots this phase requests,
    // loaded one phase ahead.
    int ph = 0;
    int e[KT];
#pragma unroll
    for (int i = 0; i < KT; ++i) e[i] = order[KT + 3 + i];
    // Fragment addresses.  swz(16 j + fr, 4 ks + fq) = 2048 j + frag0 for k-step 0 and the same with bit 6 flipped for k-step 1
    // (the swizzle term (row >> 1) & 7 does not depend on j): one register per lane, two adds per phase, everything else an
    // immediate offset of the read.
    const int frag0 = fr * 128 + ((fq ^ ((fr >> 1) & 7)) << 4);
    auto rd = [&](const char* s0, const char* s1, u32x4 (&dst)[2][2], auto dj_, int off) __attribute__((always_inline)) {
        constexpr int dj = decltype(dj_)::value * 2048;
        dst[0][0] = *reinterpret_cast<const u32x4*>(s0 + off);
        dst[0][1] = *reinterpret_cast<const u32x4*>(s0 + off + dj);
        dst[1][0] = *reinterpret_cast<const u32x4*>(s1 + off);
        dst[1][1] = *reinterpret_cast<const u32x4*>(s1 + off + dj);
    };
    // j0, j0 + dj: the wave's two 16-row tiles in every slot of the phase; mm(k_, fragments): the MFMAs of slot k; side(k_):
    // VALU work placed behind them
    // ... first(): LDS reads that need what other waves wrote during the previous phase (issued behind the first barrier)
    auto run_phase = [&](auto y0_, auto y1_, auto r0_, auto r1_, int j0, auto dj_, auto&& first, auto&& mm, auto&& side) __attribute__((always_inline)) {
        const int hb = (ph & 1) * HALF, ob = HALF - hb;
        const char* const s0 = smem + hb + j0 * 2048 + frag0;
        const char* const s1 = smem + hb + j0 * 2048 + (frag0 ^ 64);
        u32x4 wa[2][2], wb[2][2];
        wait_vm<decltype(y0_)::value>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if constexpr (decltype(r0_)::value) {
            issue(e[0], ob + 3 * SLOT);
            issue(e[1], ob + 4 * SLOT);
        }
        first();
        rd(s0, s1, wa, dj_, 0);
        rd(s0, s1, wb, dj_, SLOT);
        mm(std::integral_constant<int, 0>{}, wa);
        side(std::integral_constant<int, 0>{});
        rd(s0, s1, wa, dj_, 2 * SLOT);
        mm(std::integral_constant<int, 1>{}, wb);
        side(std::integral_constant<int, 1>{});
        mm(std::integral_constant<int, 2>{}, wa);
        side(std::integral_constant<int, 2>{});
        wait_vm<decltype(y1_)::value>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if constexpr (decltype(r1_)::value) {
            issue(e[2], hb);
            issue(e[3], hb + SLOT);
            issue(e[4], hb + 2 * SLOT);
        }
        rd(s0, s1, wa, dj_, 3 * SLOT);
        rd(s0, s1, wb, dj_, 4 * SLOT);
        mm(std::integral_constant<int, 3>{}, wa);
        side(std::integral_constant<int, 3>{});
        mm(std::integral_constant<int, 4>{}, wb);
        side(std::integral_constant<int, 4>{});
        ++ph;
        if constexpr (decltype(r1_)::value) {
            const int nb = KT * ph + KT + 3;
#pragma unroll
            for (int i = 0; i < KT; ++i) e[i] = order[nb + i];
        }
    };
    auto no_first = []() __attribute__((always_inline)) {};
    auto steady = [&](int j0, auto dj_, auto&& first, auto&& mm, auto&& side) __attribute__((always_inline)) {
        run_phase(std::integral_constant<int, 5>{}, std::integral_constant<int, 5>{}, True{}, True{}, j0, dj_, first, mm, side);
    };
    const int v0 = h;      // slots of W1: value tile h and gate tile 2 + h of this wave half
    const int c0 = 2 * h;  // slots of W2: the two 16-column tiles 2 h, 2 h + 1 of this wave half
    constexpr std::integral_constant<int, 2> DJ_W1{};
    constexpr std::integral_constant<int, 1> DJ_W2{};

    // One quarter of a column tile's GEGLU epilogue: row tile u / 2, columns 2 (u % 2) + {0, 1} of the lane's four.  Lane
    // holds, for rows mrow0 + 16 i + fr, the values of product columns 32 jt + 8 fq + 4 h + r (acc[0]) and their gates (acc[1]).
    float o[4];
    auto gelu_unit = [&](const f32x4 (&acc)[2][2], int jt, int par, int buf, auto u_) __attribute__((always_inline)) {
        constexpr int u = decltype(u_)::value;
        constexpr int i = u >> 1, r = (u & 1) * 2;
        const float* const tb = tab + jt * 128 + fq * 16 + 4 * h + r;
        const f32x2 bv = *reinterpret_cast<const f32x2*>(tb), bg = *reinterpret_cast<const f32x2*>(tb + 8);
        const f32x2 cv = *reinterpret_cast<const f32x2*>(tb + 64), cg = *reinterpret_cast<const f32x2*>(tb + 64 + 8);
        // LN(a) W^T = rstd * (a W'^T - mean * colsum(W')) + (beta W^T + b); two values per packed instruction
        const f32x2 mu = {ln_mu[i], ln_mu[i]}, rs = {ln_rs[i], ln_rs[i]};
        const f32x2 gv = rs * (f32x2{acc[1][i][r], acc[1][i][r + 1]} - mu * cg) + bg;
        const f32x2 vv = rs * (f32x2{acc[0][i][r], acc[0][i][r + 1]} - mu * cv) + bv;
        const f32x2 pv = vv * gelu_erf_fast_f2(gv);
        o[r] = pv[0];
        o[r + 1] = pv[1];
        if constexpr ((u & 1) != 0)
            *reinterpret_cast<u32x2*>(pbuf + buf * PBUF + ((((pr * 2 + par) * 2 + i) * 64 + lane) << 4) + 8 * h) =
                u32x2{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
    };
    auto zero4 = [](f32x4 (&acc)[2][2]) __attribute__((always_inline)) {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[jj][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    auto nothing = [](auto) __attribute__((always_inline)) {};
    // the KT slots of one column tile of W1 into `acc`
    auto phase_w1 = [&](f32x4 (&acc)[2][2], auto&& side) __attribute__((always_inline)) {
        zero4(acc);
        steady(v0, DJ_W1, no_first, [&](auto k_, const u32x4 (&w)[2][2]) __attribute__((always_inline)) {
            constexpr int k = decltype(k_)::value;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    Mma16<bf16_t>::mma(acc[0][i], w[ks][0], afr[i][2 * k + ks]);
                    Mma16<bf16_t>::mma(acc[1][i], w[ks][1], afr[i][2 * k + ks]);
                }
        }, side);
    };
    // the G2 slots of one k-tile of W2 (product columns of one pair of column tiles): `buf` = exchange buffer of that pair
    auto phase_w2 = [&](int buf, auto&& side) __attribute__((always_inline)) {
        // [k-step = column-tile parity][row tile]: the lane group's eight products, both halves.  The partner half wrote its
        // last piece during the previous phase, behind that phase's second barrier: read behind this phase's first one.
        u32x4 pf[2][2];
        auto first = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    pf[ks][i] = *reinterpret_cast<const u32x4*>(pbuf + buf * PBUF + ((((pr * 2 + ks) * 2 + i) * 64 + lane) << 4));
        };
        steady(c0, DJ_W2, first, [&](auto g_, const u32x4 (&w)[2][2]) __attribute__((always_inline)) {
            constexpr int g = decltype(g_)::value;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    Mma16<bf16_t>::mma(oacc[g][0][i], w[ks][0], pf[ks][i]);
                    Mma16<bf16_t>::mma(oacc[g][1][i], w[ks][1], pf[ks][i]);
                }
        }, side);
    };
    f32x4 acc_a[2][2], acc_b[2][2];  // first product of the even / odd column tile of a pair: [value | gate][row tile]

    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the table writes above; the first barrier publishes them
    // pair 0 has no earlier pair whose second product could cover the epilogue of its odd tile
    phase_w1(acc_a, nothing);
    phase_w1(acc_b, [&](auto k_) __attribute__((always_inline)) {
        if constexpr (decltype(k_)::value < 4) gelu_unit(acc_a, 0, 0, 0, k_);
    });
    static_for<0, 4>([&](auto u_) __attribute__((always_inline)) { gelu_unit(acc_b, 1, 1, 0, u_); });
    // Pair q: column tile 2q (A), column tile 2q + 1 (B) with the epilogue of A behind its MFMAs, then the k-tile q - 1 of
    // W2 (C) with the epilogue of B behind its MFMAs; the products of pair q reach the second product one pair later.
    for (int q = 1; q < NP; ++q) {
        phase_w1(acc_a, nothing);
        phase_w1(acc_b, [&](auto k_) __attribute__((always_inline)) {
            if constexpr (decltype(k_)::value < 4) gelu_unit(acc_a, 2 * q, 0, q & 1, k_);
        });
        phase_w2((q - 1) & 1, [&](auto g_) __attribute__((always_inline)) {
            if constexpr (decltype(g_)::value < 4) gelu_unit(acc_b, 2 * q + 1, 1, q & 1, g_);
        });
    }
    phase_w2((NP - 1) & 1, nothing);
    // ---- the columns of W2 that multiply t itself (conv_output of the residue): KT phases, the last two with the ring draining
    static_assert(NPH == 3 * NP + KT, "phase count");
    static_for<0, KT>([&](auto kt_) __attribute__((always_inline)) {
        constexpr int kt = decltype(kt_)::value;
        auto mm = [&](auto g_, const u32x4 (&w)[2][2]) __attribute__((always_inline)) {
            constexpr int g = decltype(g_)::value;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    Mma16<bf16_t>::mma(oacc[g][0][i], w[ks][0], afr[i][2 * kt + ks]);
                    Mma16<bf16_t>::mma(oacc[g][1][i], w[ks][1], afr[i][2 * kt + ks]);
                }
        };
        if constexpr (kt + 2 < KT) steady(c0, DJ_W2, no_first, mm, nothing);
        else if constexpr (kt + 2 == KT)  // last phase but one: nothing left to request behind its second barrier
            run_phase(std::integral_constant<int, 5>{}, std::integral_constant<int, 5>{}, True{}, False{}, c0, DJ_W2, no_first, mm, nothing);
        else  // last phase: only its own slots 3-4 are younger than slots 0-2
            run_phase(std::integral_constant<int, 2>{}, std::integral_constant<int, 0>{}, False{}, False{}, c0, DJ_W2, no_first, mm, nothing);
    });

    // ---- epilogue: lane holds rows mrow0 + 16 i + fr, columns 64 g + 16 fq + 8 h + 4 jj + r
    __syncthreads();  // every wave is done with the ring: its first bytes become the column-sum exchange buffer
    float* const colbuf = reinterpret_cast<float*>(smem);  // [row group][C][2]
    const bool want_stats = p.stats.sums != nullptr;
#pragma unroll
    for (int g = 0; g < G2; ++g) {
        const int col = g * 64 + fq * 16 + 8 * h;
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bias2 + col), b1 = *reinterpret_cast<const f32x4*>(p.bias2 + col + 4);
        float s8[8], q8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) s8[e] = q8[e] = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const long m = mrow0 + i * 16 + fr;
            float o[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                o[r] = oacc[g][0][i][r] + b0[r];
                o[4 + r] = oacc[g][1][i][r] + b1[r];
            }
            if (p.R) {
                const u32x4 rv = *reinterpret_cast<const u32x4*>(p.R + (m * p.ldr + col) * 2);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[2 * e] += __uint_as_float(rv[e] << 16);
                    o[2 * e + 1] += __uint_as_float(rv[e] & 0xffff0000u);
                }
            }
            u32x4 pk;
#pragma unroll
            for (int e = 0; e < 4; ++e) pk[e] = pack_bf16x2(o[2 * e], o[2 * e + 1]);
            *reinterpret_cast<u32x4*>(p.C + (m * p.ldc + col) * 2) = pk;
            if (want_stats) {
                // statistics are taken of the values as stored (after rounding)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v0 = __uint_as_float(pk[e] << 16), v1 = __uint_as_float(pk[e] & 0xffff0000u);
                    s8[2 * e] += v0;
                    s8[2 * e + 1] += v1;
                    q8[2 * e] = fmaf(v0, v0, q8[2 * e]);
                    q8[2 * e + 1] = fmaf(v1, v1, q8[2 * e + 1]);
                }
            }
        }
        if (want_stats) {
            // sum over the 16 rows of the lane group: reduce-scatter over lane bits 3, 2, 1 (eight columns -> one), then
            // bit 0; lane fr ends up with column col + (fr >> 1)
#pragma unroll
            for (int w2 = 0; w2 < 2; ++w2) {
                float(&v)[8] = w2 ? q8 : s8;
#pragma unroll
                for (int hf = 4; hf >= 1; hf >>= 1) {
                    const bool upper = (fr & (2 * hf)) != 0;
#pragma unroll
                    for (int e = 0; e < hf; ++e) {
                        const float keep = upper ? v[e + hf] : v[e];
                        const float send = upper ? v[e] : v[e + hf];
                        v[e] = keep + __shfl_xor(send, 2 * hf);
                    }
                }
                v[0] += __shfl_xor(v[0], 1);
            }
            if ((fr & 1) == 0) {
                float* dst = colbuf + ((pr * C + col + (fr >> 1)) << 1);
                dst[0] = s8[0];
                dst[1] = q8[0];
            }
        }
    }
    if (want_stats) {
        __syncthreads();
        // wave g sums group g's 64 columns over the four row groups, in row order, and adds the groups' shares to memory
        if (wave < G2) {
            const int col = wave * 64 + lane;
            float s = 0.f, qq = 0.f;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                s += colbuf[(r4 * C + col) << 1];
                qq += colbuf[((r4 * C + col) << 1) + 1];
            }
            const long b = m0 / p.rows_per_batch;
            group_stats_add(p.stats, b, wave * 64, lane, s, qq);
            if (p.stats2.sums) group_stats_add(p.stats2, b, wave * 64, lane, s, qq);
        }
    }
}

template <int KT> int launch(const FfnArgs& a, hipStream_t stream) {
    static DmLdsAttr attr;
    auto kfn = ffn_geglu_kernel<KT>;
    constexpr int LDS = 2 * KT * SLOT + 8 * KT * 128 * 4 + 2 * 4 * 2 * 2 * 64 * 16;  // ring + column terms + two product exchange buffers
    static_assert(LDS <= 160 * 1024 && 4 * KT * 64 * 2 * 4 <= 2 * KT * SLOT, "LDS budget");
    if (const int rc = dm_set_lds_attr(reinterpret_cast<const void*>(kfn), attr, LDS)) return rc;
    hipLaunchKernelGGL(kfn, dim3(a.M / 128), dim3(512), LDS, stream, a);
    return dm_check_launch();
}

}  // namespace

extern "C" int dm_ffn_geglu_supported(int dtype, int m, int c, int rows_per_batch, int want_stats) {
    if (dtype != DM_BF16 || c != 320 || m <= 0 || m % 128) return 0;
    // fused statistics: a block's 128 rows lie inside one sample
    if (want_stats && (rows_per_batch <= 0 || rows_per_batch % 128 || m % rows_per_batch)) return 0;
    return 1;
}

extern "C" int dm_ffn_geglu(const DmFfnDesc* d, hipStream_t stream) {
    if (!d || !d->a || !d->w1 || !d->w2 || !d->c || !d->bias1 || !d->bias2 || !d->ln_in || !d->ln_colsum) return DM_ERR_ARG;
    if (!dm_ffn_geglu_supported(d->dtype, d->m, d->c_model, d->rows_per_batch, d->stats != nullptr)) return DM_ERR_ARG;
    // second image: [npad2 / 64][5 C / 64][64][128 B] with at least C rows
    if (d->npad2 % 64 || d->npad2 < d->c_model) return DM_ERR_ARG;
    if (!dm_aligned16(d->a) || !dm_aligned16(d->w1) || !dm_aligned16(d->w2) || !dm_aligned16(d->c) || !dm_aligned16(d->bias1) ||
        !dm_aligned16(d->bias2) || !dm_aligned16(d->ln_colsum) || (reinterpret_cast<uintptr_t>(d->ln_in) & 7u) || d->lda % 8 ||
        d->ldc % 8)
        return DM_ERR_ALIGN;
    if (d->r && (!dm_aligned16(d->r) || d->ldr % 8)) return DM_ERR_ALIGN;
    if (d->stats && !dm_group_stats_ok(d->stats, d->ld_stats, d->stats_cpg, d->stats_ch0, d->c_model)) return DM_ERR_ARG;
    if (d->stats2 && (!d->stats || !dm_group_stats_ok(d->stats2, d->ld_stats2, d->stats2_cpg, d->stats2_ch0, d->c_model)))
        return DM_ERR_ARG;
    FfnArgs a;
    a.A = static_cast<const char*>(d->a);
    a.lda = d->lda;
    a.W1 = static_cast<const char*>(d->w1);
    a.bias1 = d->bias1;
    a.ln_in = d->ln_in;
    a.ln_colsum = d->ln_colsum;
    a.ln_eps = d->ln_eps;
    a.W2 = static_cast<const char*>(d->w2);
    a.bias2 = d->bias2;
    a.R = static_cast<const char*>(d->r);
    a.ldr = d->ldr;
    a.C = static_cast<char*>(d->c);
    a.ldc = d->ldc;
    a.M = d->m;
    a.stats = dm_group_stats(d->stats, d->ld_stats, d->stats_cpg, d->stats_ch0, d->c_model);
    a.stats2 = dm_group_stats(d->stats2, d->ld_stats2, d->stats2_cpg, d->stats2_ch0, d->c_model);
    a.rows_per_batch = d->rows_per_batch > 0 ? d->rows_per_batch : d->m;
    return launch<5>(a, stream);
}
