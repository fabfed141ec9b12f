// Flash-style attention forward for gfx950: softmax(Q K^T / sqrt(d)) V per (batch, head) without
// ever writing the score matrix.  Non-causal; key count need not be a multiple of the tile.
//
// Work split: block = 4 waves = 64*QT query rows of one (batch, head); each wave owns QT tiles of
// 16 query rows and keeps their (pre-scaled) Q fragments and the running (max, sum, O) in registers.
// K and V tiles (64 bf16 / 32 fp32 keys) are streamed global -> LDS by global_load_lds_dwordx4 into a
// two-stage ring: tile t+1 is in flight while tile t is consumed, one barrier per tile.  That DMA
// writes LDS lane-linearly, so both tile images are contiguous and any permutation is applied to the
// per-lane SOURCE address:
//   K image [k-step][key][64 B], 16-byte chunk c of a row stored at c ^ ksw(key)
//   V image [key][VCH chunks]: the row's D values, then (head dims that leave spare rows in the last
//           16-row tile of O^T) a chunk of ones, then zero chunks up to the pitch
// Rows / keys outside the problem read a 16-byte zero buffer.
// A 128-bit LDS read is served in four cycles, and the lanes sharing a cycle are NOT 16 consecutive
// ones: measured on MI355X they are {0-3, 12-15, 20-23, 24-27}, {4-7, 8-11, 16-19, 28-31} and the same
// two sets + 32.  With lane = 16*chunk + row, ksw() = (key >> 1) & 3 makes those sets hit 16 distinct
// 16-byte slots of the 256-byte bank line.
//
// The score tile is computed transposed, S^T = K Q^T (K is the MFMA row operand), so that after the
// MFMA every lane holds scores of ONE query row (column l&15) for keys 4*(l>>4)+r: the row max/sum
// need only two cross-lane steps, and the probabilities are already laid out as the column
// operand of the second product O^T = V^T P^T.  Its row operand V^T[d][8 keys] is read straight out
// of the row-major V tile: bf16 by ds_read_b64_tr_b16 (a 4-key x 16-column block per 16 lanes,
// delivered column-major), fp32 by one ds_read_b32 per key (the f32 MFMA takes one value per lane);
// the fp32 row pitch is an odd number of chunks so that the four key rows read together fall in
// different banks.  No transposed copy of V is ever made.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "dm_hip.h"

namespace {

struct AttnArgs {
    const char* q; long ldq;
    const char* k; long ldk;
    const char* v; long ldv;
    char* o; long ldo;
    int heads, sq, sk;
    int causal;        // 1: key j is visible to query i only when j <= i (attention.py:35-37)
    int nqblk, nblk;   // query blocks per (batch, head); blocks in the grid
    float scale_log2;  // log2(e) / sqrt(d)
    int split;         // fp32 only: both products as split-bf16 products on the bf16 MFMA (opt-in mode)
};

__device__ __attribute__((aligned(16))) uint32_t g_attn_zero16[4];
// 16 bytes of 1.0 (bf16 / fp32): source of the extra V^T row whose product with P^T is the row sum
__device__ __attribute__((aligned(16))) uint32_t g_attn_ones_bf16[4] = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
__device__ __attribute__((aligned(16))) uint32_t g_attn_ones_f32[4] = {0x3F800000u, 0x3F800000u, 0x3F800000u, 0x3F800000u};

#define DM_GLDS16(src, dst)                                                                         \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src),          \
                                     (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)

// Eight fp32 values (two 16-byte chunks) as two bf16 terms each: x ~ hi + lo, lo = bf16(x - hi).
__device__ __forceinline__ void split8(const u32x4& x0, const u32x4& x1, u32x4& hi, u32x4& lo) {
    float f[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        f[e] = __uint_as_float(x0[e]);
        f[4 + e] = __uint_as_float(x1[e]);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t h = pack_bf16x2(f[2 * q], f[2 * q + 1]);
        hi[q] = h;
        lo[q] = pack_bf16x2(f[2 * q] - __uint_as_float(h << 16), f[2 * q + 1] - __uint_as_float(h & 0xffff0000u));
    }
}
// acc += a b with both fp32 operands given as split pairs: a_hi b_hi + a_lo b_hi + a_hi b_lo (fp32 accumulation)
__device__ __forceinline__ void mma_split(f32x4& acc, const u32x4& ah, const u32x4& al, const u32x4& bh, const u32x4& bl) {
    Mma16<bf16_t>::mma(acc, ah, bh);
    Mma16<bf16_t>::mma(acc, al, bh);
    Mma16<bf16_t>::mma(acc, ah, bl);
}

template <int I, int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// Two transposed LDS reads (4 keys x 16 columns each, delivered column-major) at immediate offsets from
// one address register.  Written as asm so that the compiler does not order them behind the LDS-DMA of the
// NEXT tile (it drains vmcnt before an LDS read it knows about while such a DMA is in flight); the caller
// waits with lgkmcnt(0) before using the results.
template <int OFF0, int OFF1> __device__ __forceinline__ void tr_read_pair(u32x2& a, u32x2& b, uint32_t lds_addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"
                 : "=&v"(a), "=&v"(b)
                 : "v"(lds_addr), "i"(OFF0), "i"(OFF1));
}

// DSPLIT (problems with fewer query tiles than CUs): the four waves of a block share ONE tile of 16 query rows and
// split the head dimension instead.  Each wave multiplies every fourth k-step of K Q^T, the partial score tiles are
// summed through LDS (two small buffers, alternating per key tile), every wave forms the same probabilities and
// multiplies them with every fourth 16-column tile of V.  Four times the blocks, a quarter of the MFMA chain per
// wave and key tile: the deep U-Net levels at batch 1 are bound by that chain, not by throughput.
// SPLIT (fp32 tensors, opt-in arithmetic mode): K, Q, V and the probabilities are split into two bf16 terms in
// registers and both products run as three 16x16x32 bf16 MFMAs per fragment pair (hi hi + lo hi + hi lo, fp32
// accumulation) instead of eight exact-fp32 ones.  A 32-wide k-step takes, per lane group, the group's chunk of two
// consecutive 16-float k-steps (K Q^T) or its four keys of both 16-key blocks of the tile (V^T P^T): the same
// permutation of k on both operands, so the LDS images and reads are those of the fp32 path.
template <typename T, int D, int QT, bool DSPLIT = false, bool SPLIT = false>
__global__ __launch_bounds__(256) void attn_kernel(AttnArgs p) {
    static_assert(!DSPLIT || QT == 1, "the head-dimension split works on one query tile per block");
    static_assert(!SPLIT || (sizeof(T) == 4 && !DSPLIT), "split-bf16 products exist for the plain fp32 form");
    constexpr int EPC = ElemTraits<T>::EPC;
    constexpr int ESZ = sizeof(T);
    constexpr int NKS = (D * ESZ + 63) / 64;   // 64-byte k-steps over the head dim
    constexpr int DCH = D / EPC;               // valid 16-byte chunks per Q/K row
    constexpr int DT = (D + 15) / 16;          // 16-row tiles of V^T / O^T
    constexpr int KT = 8 * EPC;                // keys per tile
    constexpr int NKT = KT / 16;
    constexpr int K_SLAB = KT * 64;            // bytes of one k-step slab
    constexpr int K_BYTES = NKS * K_SLAB;
    constexpr int VREAL = D / EPC;             // 16-byte chunks of real values per V row
    constexpr int VCH0 = DT * 16 / EPC;        // chunks covering every column an O^T tile touches
    // pitch: fp32 an odd number of chunks; bf16 an odd multiple of two chunks (the eight key rows that one
    // 32-lane half reads transposed then start in eight different 32-byte bank groups)
    constexpr int VCH = ESZ == 4 ? (VCH0 | 1) : VCH0 + (6 - VCH0 % 4) % 4;
    constexpr int VPITCH = VCH * 16;
    constexpr int V_BYTES = KT * VPITCH;
    constexpr int K_PIECES = K_BYTES / 1024;   // 1 KiB = one wave-instruction of the DMA
    constexpr int V_PIECES = (V_BYTES + 1023) / 1024;
    constexpr int STAGE = K_BYTES + V_PIECES * 1024;
    constexpr int KP = (K_PIECES + 3) / 4;     // pieces per wave (upper bound)
    constexpr int VP = (V_PIECES + 3) / 4;
    // When D leaves spare rows in the last 16-row tile of O^T, column D of the V tile is fed with ones so
    // that the second product also delivers sum_k P[q][k] (in O^T row D) and the VALU never adds the row sum.
    constexpr bool ONES = !DSPLIT && (D % 16) != 0;
    // fp32, head dim = 8 mod 16 (d = 40): the last 64-byte k-step of K Q^T holds 8 real values and 8 of padding, and
    // with the usual layout (lane group qp owns chunk qp) each of its four 16x16x4 MFMAs multiplies half padding.
    // Here lane group qp owns the two values 2*qp, 2*qp + 1 of that step instead (the same permutation of k on both
    // operands), so two MFMAs cover it: 20 instead of 24 per 16 keys.
    constexpr bool HALF_LAST = sizeof(T) == 4 && !DSPLIT && !SPLIT && (D % 16) == 8;
    constexpr int P2 = (NKS + 1) / 2;  // SPLIT: 32-wide k-steps of K Q^T (pairs of 16-float k-steps)
    constexpr int PBUF = 4 * NKT * 64 * 16;  // DSPLIT: one buffer of partial score tiles (4 waves x NKT x 64 lanes x 16 B)
    auto ksw = [](int row) { return (row >> 1) & 3; };
    static_assert(D % EPC == 0, "head dim must be a whole number of 16-byte chunks");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, qp = lane >> 4;
    // Blocks i, i+8, ... run on one XCD (one L2).  Hand each XCD a contiguous run of (batch, head,
    // query block) ids, query block fastest, so that every block streaming one head's K / V^T sits behind
    // the same L2 at about the same time and those tiles are fetched from HBM once.
    int logical;
    {
        const int bid = blockIdx.x, nq8 = p.nblk >> 3, r8 = p.nblk & 7, x = bid & 7;
        logical = (x < r8 ? x * (nq8 + 1) : r8 * (nq8 + 1) + (x - r8) * nq8) + (bid >> 3);
    }
    const int bh = logical / p.nqblk;
    const int h = bh % p.heads, b = bh / p.heads;
    const int q0 = DSPLIT ? (logical - bh * p.nqblk) * 16 : (logical - bh * p.nqblk) * (64 * QT) + wave * (16 * QT);
    const int wv = __builtin_amdgcn_readfirstlane(wave);  // scalar copy for wave-uniform branches
    const char* zsrc = reinterpret_cast<const char*>(g_attn_zero16);

    // ---- loader coordinates (fixed over the key loop)
    long k_off[KP];   // byte offset of this lane's 16 B inside K for tile 0
    int k_row[KP];    // key row inside the tile (validity against sk), -1 = never valid
    const char* v_src[VP];  // this lane's 16 B of V for tile 0 (or the constant zero / ones buffer)
    long v_adv[VP];         // bytes to advance per tile
    int v_key[VP];          // key row inside the tile (validity against sk), -1 = constant source
#pragma unroll
    for (int j = 0; j < KP; ++j) {
        const int pc = wave + 4 * j;
        const int ks = pc / (KT / 16), row = (pc % (KT / 16)) * 16 + (lane >> 2);
        const int ch = ks * 4 + ((lane & 3) ^ ksw(row));
        k_row[j] = (pc < K_PIECES && ch < DCH) ? row : -1;
        k_off[j] = (((long)b * p.sk + row) * p.ldk + (long)h * D + ch * EPC) * ESZ;
    }
#pragma unroll
    for (int j = 0; j < VP; ++j) {
        const int pc = wave + 4 * j;
        const int slot = pc * 64 + lane;  // 16-byte slot of the V image
        const int key = slot / VCH, ch = slot - key * VCH;
        v_src[j] = zsrc;
        v_adv[j] = 0;
        v_key[j] = -1;
        if (pc < V_PIECES && key < KT) {
            if (ch < VREAL) {
                v_src[j] = p.v + (((long)b * p.sk + key) * p.ldv + (long)h * D + ch * EPC) * ESZ;
                v_adv[j] = (long)KT * p.ldv * ESZ;
                v_key[j] = key;
            } else if (ONES && ch == VREAL) {
                v_src[j] = reinterpret_cast<const char*>(ESZ == 2 ? g_attn_ones_bf16 : g_attn_ones_f32);
            }
        }
    }
    auto issue = [&](int t, int stage) {
        char* sk_ = smem + stage * STAGE;
        char* sv_ = sk_ + K_BYTES;
#pragma unroll
        for (int j = 0; j < KP; ++j) {
            const int pc = wave + 4 * j;
            if (pc < K_PIECES) {  // wave-uniform
                const bool ok = k_row[j] >= 0 && t * KT + k_row[j] < p.sk;
                const char* src = ok ? p.k + k_off[j] + (long)t * KT * p.ldk * ESZ : zsrc;
                DM_GLDS16(src, sk_ + pc * 1024);
            }
        }
#pragma unroll
        for (int j = 0; j < VP; ++j) {
            const int pc = wave + 4 * j;
            if (pc < V_PIECES) {
                const bool gone = v_key[j] >= 0 && t * KT + v_key[j] >= p.sk;  // key row past the end
                DM_GLDS16(gone ? zsrc : v_src[j] + (long)t * v_adv[j], sv_ + pc * 1024);
            }
        }
    };

    // key (inside the tile) that row i of score block kt stands for
    auto key_of = [](int kt, int i) { return 16 * kt + i; };

    const int ntiles = (p.sk + KT - 1) / KT;
    issue(0, 0);

    // ---- Q fragments (column operand of S^T), pre-multiplied by log2(e)/sqrt(d)
    u32x4 qf[QT][NKS];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int row = q0 + qt * 16 + c;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int ch = ks * 4 + qp;
            const long e = ((long)b * p.sq + row) * p.ldq + (long)h * D + ch * EPC;
            u32x4 raw;
            if (HALF_LAST && ks == NKS - 1) {
                const long e2 = ((long)b * p.sq + row) * p.ldq + (long)h * D + ks * 16 + 2 * qp;
                u32x2 two = {0u, 0u};
                if (row < p.sq) two = *reinterpret_cast<const u32x2*>(p.q + e2 * ESZ);
                raw = u32x4{two[0], two[1], 0u, 0u};
            } else {
                raw = ldg16(p.q + e * ESZ, row < p.sq && ch < DCH);
            }
            float f[EPC];
            chunk_to_f32<T>(raw, f);
#pragma unroll
            for (int i = 0; i < EPC; ++i) f[i] *= p.scale_log2;
            qf[qt][ks] = f32_to_chunk<T>(f);
        }
    }

    [[maybe_unused]] u32x4 qh[QT][P2], ql[QT][P2];
    if constexpr (SPLIT) {
        const u32x4 zero = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
            for (int s2 = 0; s2 < P2; ++s2)
                split8(qf[qt][2 * s2], 2 * s2 + 1 < NKS ? qf[qt][2 * s2 + 1] : zero, qh[qt][s2], ql[qt][s2]);
    }

    f32x4 oacc[DT][QT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) oacc[dt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
    // Scores are kept relative to a per-row REFERENCE m_ref that only moves when a tile's maximum
    // exceeds it by more than 2^8: the MFMA chain starts from -m_ref, so S^T arrives already shifted and
    // the common path is just exp2 + convert.  Any reference gives the same softmax after the final
    // division; staying within 2^8 of the true maximum keeps exp2 far from overflow.
    float m_ref[QT], l_run[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        m_ref[qt] = 0.f;
        l_run[qt] = 0.f;
    }

    for (int t = 0; t < ntiles; ++t) {
        // tile t has landed (it is the only DMA outstanding); the barrier publishes every wave's share
        // and retires all reads of the other stage, which is refilled right away
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (t + 1 < ntiles) issue(t + 1, (t + 1) & 1);
        const char* s_k = smem + (t & 1) * STAGE;
        const char* s_v = s_k + K_BYTES;

        // ---- S^T - m_ref = K Q^T - m_ref  (log2 units); K fragments are fetched one k-step ahead
        f32x4 sacc[NKT][QT];
        auto load_k = [&](int ks, u32x4 (&dst)[NKT]) {
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
                const int row = key_of(kt, c);
                if (HALF_LAST && ks == NKS - 1) {
                    const u32x2 two = *reinterpret_cast<const u32x2*>(s_k + ks * K_SLAB + row * 64 +
                                                                      (((qp >> 1) ^ ksw(row)) << 4) + (qp & 1) * 8);
                    dst[kt] = u32x4{two[0], two[1], 0u, 0u};
                } else {
                    dst[kt] = *reinterpret_cast<const u32x4*>(s_k + ks * K_SLAB + row * 64 + ((qp ^ ksw(row)) << 4));
                }
            }
        };
        // one k-step of K Q^T for one (key block, query tile)
        auto mma_ks = [&](f32x4& acc, const u32x4& kfrag, const u32x4& qfrag, int ks) {
            if constexpr (HALF_LAST) {
                if (ks == NKS - 1) {
#pragma unroll
                    for (int e = 0; e < 2; ++e)
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(kfrag[e]), __uint_as_float(qfrag[e]), acc, 0, 0, 0);
                    return;
                }
            }
            Mma16<T>::mma(acc, kfrag, qfrag);
        };
        if constexpr (DSPLIT) {
            const float start = wv == 0 ? -m_ref[0] : 0.f;  // the shift is added once, by wave 0's partial
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) sacc[kt][0] = f32x4{start, start, start, start};
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                if ((ks & 3) == wv) {
                    u32x4 kfs[NKT];
                    load_k(ks, kfs);
#pragma unroll
                    for (int kt = 0; kt < NKT; ++kt) Mma16<T>::mma(sacc[kt][0], kfs[kt], qf[0][ks]);
                }
            }
            f32x4* part = reinterpret_cast<f32x4*>(smem + 2 * STAGE + (t & 1) * PBUF);
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) part[(wv * NKT + kt) * 64 + lane] = sacc[kt][0];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
                f32x4 sum = part[kt * 64 + lane];
#pragma unroll
                for (int w = 1; w < 4; ++w) sum += part[(w * NKT + kt) * 64 + lane];
                sacc[kt][0] = sum;
            }
        } else if constexpr (SPLIT) {
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) sacc[kt][qt] = f32x4{-m_ref[qt], -m_ref[qt], -m_ref[qt], -m_ref[qt]};
#pragma unroll
            for (int s2 = 0; s2 < P2; ++s2) {
                u32x4 k0[NKT], k1[NKT];
                load_k(2 * s2, k0);
                if (2 * s2 + 1 < NKS) {
                    load_k(2 * s2 + 1, k1);
                } else {
#pragma unroll
                    for (int kt = 0; kt < NKT; ++kt) k1[kt] = u32x4{0u, 0u, 0u, 0u};
                }
#pragma unroll
                for (int kt = 0; kt < NKT; ++kt) {
                    u32x4 kh, kl;
                    split8(k0[kt], k1[kt], kh, kl);
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt) mma_split(sacc[kt][qt], kh, kl, qh[qt][s2], ql[qt][s2]);
                }
            }
        } else {
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) sacc[kt][qt] = f32x4{-m_ref[qt], -m_ref[qt], -m_ref[qt], -m_ref[qt]};
            u32x4 kf[2][NKT];
            load_k(0, kf[0]);
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                if (ks + 1 < NKS) load_k(ks + 1, kf[(ks + 1) & 1]);
#pragma unroll
                for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt) mma_ks(sacc[kt][qt], kf[ks & 1][kt], qf[qt][ks], ks);
            }
        }
        // ---- softmax numerators; lane holds keys t*KT + key_of(kt, 4*qp + r) of query row c
        // Masks, only on the tiles that need them (wave-uniform test): keys past the end of the sequence
        // and, when causal, keys after the query position.  The key index is made opaque inside the branch
        // so that none of this arithmetic is hoisted into the common path.
        const bool tail = (t + 1) * KT > p.sk;
        if (tail || (p.causal && (t + 1) * KT > q0 + 1)) {
            int kbase = t * KT + 4 * qp;
            asm volatile("" : "+v"(kbase));
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                int limit = p.sk;
                if (p.causal) limit = min(limit, q0 + qt * 16 + c + 1);
#pragma unroll
                for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (kbase + 16 * kt + r >= limit) sacc[kt][qt][r] = -INFINITY;
            }
        }
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            float mloc = sacc[0][qt][0];
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) mloc = fmaxf(mloc, sacc[kt][qt][r]);
            // wave-uniform: the first tile fixes the reference exactly, later ones only when some row
            // outgrew it by 2^8
            if (t == 0 || __builtin_amdgcn_ballot_w64(mloc > 8.f) != 0) {
                mloc = fmaxf(mloc, __shfl_xor(mloc, 16));
                mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
                const float delta = t == 0 ? mloc : fmaxf(mloc, 0.f);
                m_ref[qt] += delta;
                const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
                for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sacc[kt][qt][r] -= delta;
                l_run[qt] *= alpha;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) oacc[dt][qt] *= alpha;
            }
            float psum = 0.f;
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pv = __builtin_amdgcn_exp2f(sacc[kt][qt][r]);
                    sacc[kt][qt][r] = pv;
                    if constexpr (!ONES) psum += pv;
                }
            if constexpr (!ONES) l_run[qt] += psum;
        }
        // ---- O^T += V^T P^T
        if constexpr (sizeof(T) == 2) {
            // The lane's eight probabilities of k-step kk are keys kk*32 + 4*qp + r (score block 2*kk) and
            // kk*32 + 16 + 4*qp + r (block 2*kk + 1); V^T[d = dt*16 + c] for the same keys comes from two
            // transposed reads: lane 4*q + pp of a 16-lane group addresses key row q, columns 4*pp .. 4*pp+3.
            const uint32_t v_addr = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) const char*)s_v) +
                                    (4 * qp + (c >> 2)) * VPITCH + 8 * (c & 3);
            static_for<0, NKT / 2>([&](auto kk_) {
                constexpr int kk = decltype(kk_)::value;
                u32x2 vlo[DT], vhi[DT];
                static_for<0, DT>([&](auto dt_) {
                    constexpr int dt = decltype(dt_)::value;
                    if (!DSPLIT || (dt & 3) == wv)
                        tr_read_pair<kk * 32 * VPITCH + dt * 32, (kk * 32 + 16) * VPITCH + dt * 32>(vlo[dt], vhi[dt], v_addr);
                });
                u32x4 pb[QT];
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) {
                    const f32x4 a = sacc[2 * kk][qt], d2 = sacc[2 * kk + 1][qt];
                    pb[qt] = u32x4{pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]), pack_bf16x2(d2[0], d2[1]),
                                   pack_bf16x2(d2[2], d2[3])};
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    if (DSPLIT && (dt & 3) != wv) continue;
                    const u32x4 va = u32x4{vlo[dt][0], vlo[dt][1], vhi[dt][0], vhi[dt][1]};
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt) Mma16<T>::mma(oacc[dt][qt], va, pb[qt]);
                }
            });
        } else if constexpr (SPLIT) {
            static_assert(!SPLIT || NKT == 2, "a tile of 32 keys is one 32-wide k-step of the second product");
            u32x4 ph[QT], pl[QT];
#pragma unroll
            for (int qt = 0; qt < QT; ++qt)
                split8(__builtin_bit_cast(u32x4, sacc[0][qt]), __builtin_bit_cast(u32x4, sacc[NKT - 1][qt]), ph[qt], pl[qt]);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                // keys 4*qp + e and 16 + 4*qp + e, e = 0..3, of column dt*16 + c
                u32x4 v0, v1, vh, vl;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v0[e] = *reinterpret_cast<const uint32_t*>(s_v + (4 * qp + e) * VPITCH + (dt * 16 + c) * 4);
                    v1[e] = *reinterpret_cast<const uint32_t*>(s_v + (16 + 4 * qp + e) * VPITCH + (dt * 16 + c) * 4);
                }
                split8(v0, v1, vh, vl);
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) mma_split(oacc[dt][qt], vh, vl, ph[qt], pl[qt]);
            }
        } else {
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    if (DSPLIT && (dt & 3) != wv) continue;
                    // keys 16*kt + 4*qp + e, e = 0..3, of column dt*16 + c: the e-th f32 MFMA of the chunk
                    u32x4 va;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        va[e] = *reinterpret_cast<const uint32_t*>(s_v + (16 * kt + 4 * qp + e) * VPITCH + (dt * 16 + c) * 4);
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt) {
                        const f32x4 pr = sacc[kt][qt];
                        const u32x4 pb = u32x4{__float_as_uint(pr[0]), __float_as_uint(pr[1]), __float_as_uint(pr[2]),
                                               __float_as_uint(pr[3])};
                        Mma16<T>::mma(oacc[dt][qt], va, pb);
                    }
                }
            }
        }
    }

    // ---- normalise and store: lane holds O[row c][d = dt*16 + 4*qp + r]
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        float l;
        if constexpr (ONES) {
            // O^T row D (the ones row) sits in tile DT-1 at lane group (D % 16) / 4, element D % 4
            l = __shfl(oacc[DT - 1][qt][D % 4], ((D % 16) / 4) * 16 + c);
        } else {
            l = l_run[qt];
            l += __shfl_xor(l, 16);
            l += __shfl_xor(l, 32);
        }
        const float inv = 1.0f / l;
        const int row = q0 + qt * 16 + c;
        if (row >= p.sq) continue;
        T* orow = reinterpret_cast<T*>(p.o) + ((long)b * p.sq + row) * p.ldo + (long)h * D;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int d0 = dt * 16 + 4 * qp;
            if (d0 >= D || (DSPLIT && (dt & 3) != wv)) continue;
            const f32x4 v = oacc[dt][qt] * inv;
            if constexpr (sizeof(T) == 4) {
                *reinterpret_cast<f32x4*>(orow + d0) = v;
            } else {
                *reinterpret_cast<u32x2*>(orow + d0) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
            }
        }
    }
}

template <typename T, int D, int QT, bool DSPLIT = false, bool SPLIT = false>
int attn_launch_qt(const AttnArgs& a, int batch, hipStream_t s) {
    constexpr int EPC = ElemTraits<T>::EPC;
    constexpr int NKS = (D * (int)sizeof(T) + 63) / 64, DT = (D + 15) / 16, KT = 8 * EPC;
    constexpr int VCH0 = DT * 16 / EPC;  // V image: see the kernel
    constexpr int VCH = sizeof(T) == 4 ? (VCH0 | 1) : VCH0 + (6 - VCH0 % 4) % 4;
    constexpr int LDS = 2 * (NKS * KT * 64 + (KT * VCH * 16 + 1023) / 1024 * 1024) + (DSPLIT ? 2 * 4 * (KT / 16) * 64 * 16 : 0);
    static DmLdsAttr attr_done;
    auto kfn = attn_kernel<T, D, QT, DSPLIT, SPLIT>;
    if (const int rc = dm_set_lds_attr(reinterpret_cast<const void*>(kfn), attr_done, LDS)) return rc;
    AttnArgs args = a;
    args.nqblk = DSPLIT ? (a.sq + 15) / 16 : (a.sq + 64 * QT - 1) / (64 * QT);
    const long nblk = (long)args.nqblk * a.heads * batch;
    if (nblk > 0x7fffffffL) return DM_ERR_ARG;
    args.nblk = (int)nblk;
    hipLaunchKernelGGL(kfn, dim3(args.nblk), dim3(256), LDS, s, args);
    return dm_check_launch();
}

// Rows per block (64*QT): as many as keep >= 2 blocks per CU; wide heads cap QT by registers.
template <typename T, int D> int attn_launch(const AttnArgs& a, int batch, hipStream_t s) {
    const long per64 = (long)((a.sq + 63) / 64) * a.heads * batch;
    static const int forced = getenv("DM_ATTN_QT") ? atoi(getenv("DM_ATTN_QT")) : 0;  // tuning knob
    if constexpr (sizeof(T) == 4) {
        // opt-in split-bf16 products: 16 rows per wave (32 measured slower: 1.58 -> 1.73 ms per forward at batch 1); the
        // small wide-head problems keep the exact-fp32 head-dimension split below
        if (a.split && !(D >= 160 && per64 < 128)) return attn_launch_qt<T, D, 1, false, true>(a, batch, s);
    }
    if (forced == 1) return attn_launch_qt<T, D, 1>(a, batch, s);
    if (forced == 2) return attn_launch_qt<T, D, 2>(a, batch, s);
    if constexpr (D <= 80) {
        if (forced == 4) return attn_launch_qt<T, D, 4>(a, batch, s);
    }
    // Measured on MI355X inside whole U-Net forwards (bf16 at batch 1 / 4 / 16, fp32 at batch 1, all four levels).
    // fp32: 16 rows per wave wins or ties everywhere.  bf16: a 16-row wave reads 14 KB of K / V fragments from LDS per
    // key tile for 14 MFMAs, and with four or five such waves per SIMD the LDS pipe, not the MFMA or exp rate, sets
    // the pace; 32 rows per wave halve those bytes per row and win once there are enough blocks to keep every CU
    // supplied (S = 4096, d = 40 at batch 16: 623 -> 480 us; S = 256, d = 160: 24.0 -> 20.1 us), and lose below
    // that (batch 1, S = 4096: 61.8 -> 65.2 us).  64 rows per wave win only for d = 40 with a long key loop and at least
    // 4096 64-row tiles (S = 4096 at batch 8 / 16 / 32: 296 -> 280, 463 -> 442 inside the U-Net, 1046 -> 1003 us; batch 4: a
    // tie; batch 2, d = 80 and the 77-key cross-attention: 6-15 % slower).
    if constexpr (sizeof(T) == 2) {
        if constexpr (D == 40) {
            if (forced == 0 && per64 >= 4096 && a.sk >= 1024) return attn_launch_qt<T, D, 4>(a, batch, s);
        }
        if (forced == 0 && (per64 >= 1024 || (per64 >= 512 && D >= 160))) return attn_launch_qt<T, D, 2>(a, batch, s);
    }
    // Far fewer 64-row blocks than CUs and a long MFMA chain per key tile (fp32, wide heads): split the head
    // dimension over the waves instead (16 rows per block).  Measured at batch 1 fp32: d = 160, S = 256: 32.8 -> 23.1 us
    // (77 keys: 16.9 -> 13.0 us); d <= 80 and every bf16 shape are faster unsplit.
    if constexpr (sizeof(T) == 4 && D >= 160) {
        static const int dsplit = getenv("DM_ATTN_DSPLIT") ? atoi(getenv("DM_ATTN_DSPLIT")) : -1;  // tuning knob: 0 / 1 force
        if (dsplit == 1 || (dsplit < 0 && per64 < 128)) return attn_launch_qt<T, D, 1, true>(a, batch, s);
    }
    return attn_launch_qt<T, D, 1>(a, batch, s);
}

template <typename T> int attn_dispatch(const AttnArgs& a, int batch, int d, hipStream_t s) {
    switch (d) {
        case 40: return attn_launch<T, 40>(a, batch, s);
        case 64: return attn_launch<T, 64>(a, batch, s);
        case 80: return attn_launch<T, 80>(a, batch, s);
        case 160: return attn_launch<T, 160>(a, batch, s);
        default: return DM_ERR_HEAD_DIM;
    }
}

}  // namespace

static int attention_entry(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, void* o, long ldo,
                           int batch, int heads, int sq, int sk, int d, int causal, int dtype, int split,
                           hipStream_t stream) {
    if (!q || !k || !v || !o || batch <= 0 || heads <= 0 || sq <= 0 || sk <= 0) return DM_ERR_ARG;
    if (split && dtype != DM_F32) return DM_ERR_ARG;
    if (causal && sq != sk) return DM_ERR_ARG;
    if (dtype != DM_F32 && dtype != DM_BF16) return DM_ERR_ARG;
    const int epc = dtype == DM_F32 ? 4 : 8;
    if (ldq % epc || ldk % epc || ldv % epc || ldo % 4 || d % epc) return DM_ERR_ALIGN;
    if (!dm_aligned16(q) || !dm_aligned16(k) || !dm_aligned16(v) || !dm_aligned16(o)) return DM_ERR_ALIGN;
    AttnArgs a;
    a.q = static_cast<const char*>(q); a.ldq = ldq;
    a.k = static_cast<const char*>(k); a.ldk = ldk;
    a.v = static_cast<const char*>(v); a.ldv = ldv;
    a.o = static_cast<char*>(o); a.ldo = ldo;
    a.heads = heads; a.sq = sq; a.sk = sk;
    a.causal = causal ? 1 : 0;
    a.scale_log2 = 1.4426950408889634f / sqrtf((float)d);
    a.split = split;
    return dtype == DM_F32 ? attn_dispatch<float>(a, batch, d, stream) : attn_dispatch<bf16_t>(a, batch, d, stream);
}

extern "C" int dm_attention(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, void* o,
                            long ldo, int batch, int heads, int sq, int sk, int d, int causal, int dtype,
                            hipStream_t stream) {
    return attention_entry(q, ldq, k, ldk, v, ldv, o, ldo, batch, heads, sq, sk, d, causal, dtype, 0, stream);
}

extern "C" int dm_attention_split_f32(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, void* o,
                                      long ldo, int batch, int heads, int sq, int sk, int d, int causal,
                                      hipStream_t stream) {
    return attention_entry(q, ldq, k, ldk, v, ldv, o, ldo, batch, heads, sq, sk, d, causal, DM_F32, 1, stream);
}
