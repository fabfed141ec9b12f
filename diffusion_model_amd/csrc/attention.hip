// Flash-style attention forward for gfx950: softmax(Q K^T / sqrt(d)) V per (batch, head) without
// ever writing the score matrix.  Non-causal; key count need not be a multiple of the tile.
//
// Work split: block = 4 waves = 64*QT query rows of one (batch, head); each wave owns QT tiles of
// 16 query rows and keeps their Q fragments and the running (max, sum, O) in registers.  K and
// V^T tiles (128 bytes of keys: 64 bf16 / 32 fp32) are staged through LDS and shared by the 4 waves.
//
// The score tile is computed transposed, S^T = K Q^T (K is the MFMA row operand), so that after the
// MFMA every lane holds scores of ONE query row (column l&15) for keys 4*(l>>4)+r: the row max/sum
// need only two cross-lane steps, and the probabilities are already laid out as the column
// operand of the second product O^T = V^T P^T, whose row operand comes from the V^T tile (keys
// contiguous).  V^T is produced once per projection by dm_transpose_v.
#include "common.h"
#include "dm_hip.h"

namespace {

struct AttnArgs {
    const char* q; long ldq;
    const char* k; long ldk;
    const char* vt; long ldv;
    char* o; long ldo;
    int heads, sq, sk;
    float scale_log2;  // log2(e) / sqrt(d)
};

template <typename T, int D, int QT>
__global__ __launch_bounds__(256) void attn_kernel(AttnArgs p) {
    constexpr int EPC = ElemTraits<T>::EPC;
    constexpr int ESZ = sizeof(T);
    constexpr int NKS = (D * ESZ + 63) / 64;   // 64-byte k-steps over the head dim
    constexpr int DCH = D / EPC;               // valid 16-byte chunks per Q/K row
    constexpr int DT = (D + 15) / 16;          // 16-row tiles of V^T / O^T
    constexpr int KT = 8 * EPC;                // keys per tile
    constexpr int NKT = KT / 16;
    constexpr int KROW = NKS * 64 + 16;        // padded LDS row strides (bytes)
    constexpr int VROW = 128 + 16;
    static_assert(D % EPC == 0, "head dim must be a whole number of 16-byte chunks");

    __shared__ __attribute__((aligned(16))) char s_k[KT * KROW];
    __shared__ __attribute__((aligned(16))) char s_v[DT * 16 * VROW];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, qp = lane >> 4;
    const int h = blockIdx.y, b = blockIdx.z;
    const int q0 = blockIdx.x * (64 * QT) + wave * (16 * QT);

    // Q fragments (column operand of S^T): lane supplies Q[row c of tile qt][chunk ks*4+qp]
    u32x4 qf[QT][NKS];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int row = q0 + qt * 16 + c;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int ch = ks * 4 + qp;
            const long e = ((long)b * p.sq + row) * p.ldq + (long)h * D + ch * EPC;
            qf[qt][ks] = ldg16(p.q + e * ESZ, row < p.sq && ch < DCH);
        }
    }

    f32x4 oacc[DT][QT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) oacc[dt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run[QT], l_run[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        m_run[qt] = -1e30f;
        l_run[qt] = 0.f;
    }

    const int ntiles = (p.sk + KT - 1) / KT;
    for (int t = 0; t < ntiles; ++t) {
        __syncthreads();
        // ---- stage K tile [KT][NKS*4 chunks] and V^T tile [DT*16][8 chunks]
        for (int idx = tid; idx < KT * NKS * 4; idx += 256) {
            const int row = idx / (NKS * 4), ch = idx - row * (NKS * 4);
            const int key = t * KT + row;
            const long e = ((long)b * p.sk + key) * p.ldk + (long)h * D + ch * EPC;
            *reinterpret_cast<u32x4*>(s_k + row * KROW + ch * 16) = ldg16(p.k + e * ESZ, key < p.sk && ch < DCH);
        }
        for (int idx = tid; idx < DT * 16 * 8; idx += 256) {
            const int row = idx >> 3, ch = idx & 7;
            const long e = (((long)b * p.heads + h) * D + row) * p.ldv + (long)t * KT + ch * EPC;
            *reinterpret_cast<u32x4*>(s_v + row * VROW + ch * 16) = ldg16(p.vt + e * ESZ, row < D);
        }
        __syncthreads();

        // ---- S^T = K Q^T
        f32x4 sacc[NKT][QT];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) sacc[kt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
                const u32x4 kf = *reinterpret_cast<const u32x4*>(s_k + (kt * 16 + c) * KROW + (ks * 4 + qp) * 16);
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) Mma16<T>::mma(sacc[kt][qt], kf, qf[qt][ks]);
            }
        }
        // ---- online softmax; lane holds keys t*KT + kt*16 + 4*qp + r of query row c
        const bool tail = (t + 1) * KT > p.sk;
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            float mloc = -1e30f;
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float s = sacc[kt][qt][r] * p.scale_log2;
                    if (tail && t * KT + kt * 16 + 4 * qp + r >= p.sk) s = -INFINITY;
                    sacc[kt][qt][r] = s;
                    mloc = fmaxf(mloc, s);
                }
            mloc = fmaxf(mloc, __shfl_xor(mloc, 16));
            mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
            const float mnew = fmaxf(m_run[qt], mloc);
            const float alpha = __builtin_amdgcn_exp2f(m_run[qt] - mnew);
            m_run[qt] = mnew;
            float psum = 0.f;
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pv = __builtin_amdgcn_exp2f(sacc[kt][qt][r] - mnew);
                    sacc[kt][qt][r] = pv;
                    psum += pv;
                }
            l_run[qt] = l_run[qt] * alpha + psum;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) oacc[dt][qt] *= alpha;
        }
        // ---- O^T += V^T P^T
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int kk = 0; kk < NKT / 2; ++kk) {
                u32x4 pb[QT];
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) {
                    const f32x4 a = sacc[2 * kk][qt], d2 = sacc[2 * kk + 1][qt];
                    pb[qt] = u32x4{pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]), pack_bf16x2(d2[0], d2[1]),
                                   pack_bf16x2(d2[2], d2[3])};
                }
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const char* vp = s_v + (dt * 16 + c) * VROW + kk * 64 + qp * 8;
                    const u32x2 lo = *reinterpret_cast<const u32x2*>(vp);
                    const u32x2 hi = *reinterpret_cast<const u32x2*>(vp + 32);
                    const u32x4 va = u32x4{lo[0], lo[1], hi[0], hi[1]};
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt) Mma16<T>::mma(oacc[dt][qt], va, pb[qt]);
                }
            }
        } else {
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const u32x4 va = *reinterpret_cast<const u32x4*>(s_v + (dt * 16 + c) * VROW + kt * 64 + qp * 16);
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
        float l = l_run[qt];
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        const float inv = 1.0f / l;
        const int row = q0 + qt * 16 + c;
        if (row >= p.sq) continue;
        T* orow = reinterpret_cast<T*>(p.o) + ((long)b * p.sq + row) * p.ldo + (long)h * D;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int d0 = dt * 16 + 4 * qp;
            if (d0 >= D) continue;
            const f32x4 v = oacc[dt][qt] * inv;
            if constexpr (sizeof(T) == 4) {
                *reinterpret_cast<f32x4*>(orow + d0) = v;
            } else {
                *reinterpret_cast<u32x2*>(orow + d0) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
            }
        }
    }
}

// V [rows (b, s)][h*d + j] -> V^T [b][h][j][ldv]; keys in [s, ldv) are written as zero.
template <typename T>
__global__ __launch_bounds__(256) void transpose_v_kernel(const T* __restrict__ src, long ld, int s, int heads, int d,
                                                          T* __restrict__ vt, long ldv) {
    extern __shared__ __attribute__((aligned(16))) char smem_t[];
    T* tile = reinterpret_cast<T*>(smem_t);  // [64][d + 1]
    const int k0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
    const int tid = threadIdx.x;
    const int stride = d + 1;
    for (int idx = tid; idx < 64 * d; idx += 256) {
        const int r = idx / d, j = idx - r * d;
        const int key = k0 + r;
        T v = 0;
        if (key < s) v = src[((long)b * s + key) * ld + (long)h * d + j];
        tile[r * stride + j] = v;
    }
    __syncthreads();
    for (int idx = tid; idx < 64 * d; idx += 256) {
        const int j = idx >> 6, r = idx & 63;
        vt[(((long)b * heads + h) * d + j) * ldv + k0 + r] = tile[r * stride + j];
    }
}

template <typename T, int D> int attn_launch(const AttnArgs& a, int batch, hipStream_t s) {
    if (a.sq > 64) {
        dim3 grid((a.sq + 127) / 128, a.heads, batch);
        hipLaunchKernelGGL((attn_kernel<T, D, 2>), grid, dim3(256), 0, s, a);
    } else {
        dim3 grid((a.sq + 63) / 64, a.heads, batch);
        hipLaunchKernelGGL((attn_kernel<T, D, 1>), grid, dim3(256), 0, s, a);
    }
    return dm_check_launch();
}

template <typename T> int attn_dispatch(const AttnArgs& a, int batch, int d, hipStream_t s) {
    switch (d) {
        case 40: return attn_launch<T, 40>(a, batch, s);
        case 64: return attn_launch<T, 64>(a, batch, s);
        case 80: return attn_launch<T, 80>(a, batch, s);
        case 160: return attn_launch<T, 160>(a, batch, s);
        default: return DM_ERR_ARG;
    }
}

}  // namespace

extern "C" int dm_attention(const void* q, long ldq, const void* k, long ldk, const void* vt, long ldv, void* o,
                            long ldo, int batch, int heads, int sq, int sk, int d, int dtype, hipStream_t stream) {
    if (!q || !k || !vt || !o || batch <= 0 || heads <= 0 || sq <= 0 || sk <= 0) return DM_ERR_ARG;
    if (dtype != DM_F32 && dtype != DM_BF16) return DM_ERR_ARG;
    const int epc = dtype == DM_F32 ? 4 : 8;
    if (ldv % 64 || ldv < sk || ldv < ((sk + 8 * epc - 1) / (8 * epc)) * 8 * epc) return DM_ERR_ARG;
    if (ldq % epc || ldk % epc || ldo % 4 || d % epc) return DM_ERR_ALIGN;
    if (!dm_aligned16(q) || !dm_aligned16(k) || !dm_aligned16(vt) || !dm_aligned16(o)) return DM_ERR_ALIGN;
    AttnArgs a;
    a.q = static_cast<const char*>(q); a.ldq = ldq;
    a.k = static_cast<const char*>(k); a.ldk = ldk;
    a.vt = static_cast<const char*>(vt); a.ldv = ldv;
    a.o = static_cast<char*>(o); a.ldo = ldo;
    a.heads = heads; a.sq = sq; a.sk = sk;
    a.scale_log2 = 1.4426950408889634f / sqrtf((float)d);
    return dtype == DM_F32 ? attn_dispatch<float>(a, batch, d, stream) : attn_dispatch<bf16_t>(a, batch, d, stream);
}

extern "C" int dm_transpose_v(const void* src, long ld, int batch, int s, int heads, int d, void* vt, long ldv,
                              int dtype, hipStream_t stream) {
    if (!src || !vt || batch <= 0 || s <= 0 || heads <= 0 || d <= 0 || ldv % 64 || ldv < s) return DM_ERR_ARG;
    dim3 grid((int)(ldv / 64), heads, batch);
    if (dtype == DM_F32)
        hipLaunchKernelGGL(transpose_v_kernel<float>, grid, dim3(256), 64 * (d + 1) * 4, stream,
                           static_cast<const float*>(src), ld, s, heads, d, static_cast<float*>(vt), ldv);
    else if (dtype == DM_BF16)
        hipLaunchKernelGGL(transpose_v_kernel<bf16_t>, grid, dim3(256), 64 * (d + 1) * 2, stream,
                           static_cast<const bf16_t*>(src), ld, s, heads, d, static_cast<bf16_t*>(vt), ldv);
    else
        return DM_ERR_ARG;
    return dm_check_launch();
}
