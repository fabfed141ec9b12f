// GroupNorm(+SiLU) and LayerNorm over channel-contiguous rows; HBM-bound, 16-byte accesses.
//
// GroupNorm statistics span (H*W) x (C/groups) elements per (sample, group), i.e. a strided
// slice of every row.  Pass 1 cuts each sample's rows into `nsplit` slabs; a block sweeps its
// slab with fully coalesced row reads, every thread owning fixed channel chunks so it can keep
// per-channel (sum, sumsq) in registers, and reduces them to per-group partials in a fixed
// order (bitwise reproducible).  Pass 2 folds the slab partials, then normalises + affine
// (+ SiLU) with the same coalesced sweep.  The second read of x is served by the 256 MiB
// Infinity Cache for every tensor on this path.
#include <cstdlib>

#include "common.h"
#include "dm_hip.h"

namespace {

constexpr int GN_THREADS = 256;
constexpr int GN_MAX_SLOTS = 3;   // chunks per thread per row
constexpr int GN_UNROLL = 4;      // rows in flight per thread
constexpr int GN_MAX_C = 4096;    // widest tensor: 3 chunk slots of 256 threads (bf16: 8 channels per chunk)

struct GnGeom {
    int nchunks, slots, tpr, rpi;  // chunks per row, chunk slots per thread, threads per row, rows per iteration
};

template <typename T> __host__ __device__ inline GnGeom gn_geom(int c) {
    GnGeom g;
    g.nchunks = c / ElemTraits<T>::EPC;
    g.slots = (g.nchunks + GN_THREADS - 1) / GN_THREADS;
    g.tpr = (g.nchunks + g.slots - 1) / g.slots;
    g.rpi = GN_THREADS / g.tpr;
    return g;
}

template <typename T>
__global__ __launch_bounds__(GN_THREADS) void gn_stats_kernel(const T* __restrict__ x, long ldx, int hw, int c,
                                                               int groups, float* __restrict__ partial, int nsplit) {
    constexpr int EPC = ElemTraits<T>::EPC;
    extern __shared__ __attribute__((aligned(16))) float red[];  // [rows per iteration][c][2]
    const GnGeom g = gn_geom<T>(c);
    const int split = blockIdx.x, b = blockIdx.y;
    const int rows_per = (hw + nsplit - 1) / nsplit;
    const int r0 = split * rows_per, r1 = min(hw, r0 + rows_per);
    const int tid = threadIdx.x;
    const int lir = tid % g.tpr, rl = tid / g.tpr;
    const bool active = rl < g.rpi;

    float s[GN_MAX_SLOTS][EPC], q[GN_MAX_SLOTS][EPC];
#pragma unroll
    for (int k = 0; k < GN_MAX_SLOTS; ++k)
#pragma unroll
        for (int e = 0; e < EPC; ++e) s[k][e] = q[k][e] = 0.f;

    if (active) {
        const T* base = x + ((long)b * hw) * ldx;
        // GN_UNROLL rows of loads are issued before any is consumed: a thread streams only 16 bytes
        // per row, so memory-level parallelism has to come from rows in flight
        for (int row = r0 + rl; row < r1; row += GN_UNROLL * g.rpi) {
            u32x4 v[GN_UNROLL][GN_MAX_SLOTS];
#pragma unroll
            for (int u = 0; u < GN_UNROLL; ++u) {
                const int rr = row + u * g.rpi;
#pragma unroll
                for (int k = 0; k < GN_MAX_SLOTS; ++k) {
                    const int chunk = lir + k * g.tpr;
                    v[u][k] = ldg16(base + (long)rr * ldx + chunk * EPC, rr < r1 && k < g.slots && chunk < g.nchunks);
                }
            }
#pragma unroll
            for (int u = 0; u < GN_UNROLL; ++u)
#pragma unroll
                for (int k = 0; k < GN_MAX_SLOTS; ++k) {
                    float f[EPC];
                    chunk_to_f32<T>(v[u][k], f);  // predicated-off loads are zeros and add nothing
#pragma unroll
                    for (int e = 0; e < EPC; ++e) {
                        s[k][e] += f[e];
                        q[k][e] = fmaf(f[e], f[e], q[k][e]);
                    }
                }
        }
    }
    // red[rl][channel][2]
    if (active) {
#pragma unroll
        for (int k = 0; k < GN_MAX_SLOTS; ++k) {
            const int chunk = lir + k * g.tpr;
            if (k < g.slots && chunk < g.nchunks) {
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    const int ch = chunk * EPC + e;
                    red[(rl * c + ch) * 2] = s[k][e];
                    red[(rl * c + ch) * 2 + 1] = q[k][e];
                }
            }
        }
    }
    __syncthreads();
    for (int ch = tid; ch < c; ch += GN_THREADS) {
        float a = red[ch * 2], d = red[ch * 2 + 1];
        for (int r = 1; r < g.rpi; ++r) {
            a += red[(r * c + ch) * 2];
            d += red[(r * c + ch) * 2 + 1];
        }
        red[ch * 2] = a;
        red[ch * 2 + 1] = d;
    }
    __syncthreads();
    const int cpg = c / groups;
    for (int grp = tid; grp < groups; grp += GN_THREADS) {
        float a = 0.f, d = 0.f;
        for (int ch = grp * cpg; ch < (grp + 1) * cpg; ++ch) {
            a += red[ch * 2];
            d += red[ch * 2 + 1];
        }
        float* dst = partial + (((long)b * nsplit + split) * groups + grp) * 2;
        dst[0] = a;
        dst[1] = d;
    }
}

template <typename T>
__global__ __launch_bounds__(GN_THREADS) void gn_apply_kernel(const T* __restrict__ x, long ldx, T* __restrict__ y,
                                                               long ldy, int hw, int c, int groups,
                                                               const float* __restrict__ partial, int nsplit,
                                                               const double* __restrict__ sums, long ld_sums,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float eps, int silu,
                                                               int rows_per_block) {
    constexpr int EPC = ElemTraits<T>::EPC;
    __shared__ float g_mean[64], g_rstd[64];
    __shared__ double sub[64][8][2];
    const int b = blockIdx.y, tid = threadIdx.x;
    const int cpg = c / groups;
    const GnGeom g = gn_geom<T>(c);
    const int lir = tid % g.tpr, rl = tid / g.tpr;
    // A thread owns the same channel chunks on every row.  Their affine parameters are requested first, so that their
    // latency overlaps the statistics below.
    float gam[GN_MAX_SLOTS][EPC], bet[GN_MAX_SLOTS][EPC];
#pragma unroll
    for (int k = 0; k < GN_MAX_SLOTS; ++k) {
        const int chunk = lir + k * g.tpr;
        const bool ok = rl < g.rpi && k < g.slots && chunk < g.nchunks;
#pragma unroll
        for (int e4 = 0; e4 < EPC; e4 += 4) {
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            const f32x4 gv = ok ? *reinterpret_cast<const f32x4*>(gamma + chunk * EPC + e4) : zero;
            const f32x4 bv = ok ? *reinterpret_cast<const f32x4*>(beta + chunk * EPC + e4) : zero;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                gam[k][e4 + e] = gv[e];
                bet[k][e4 + e] = bv[e];
            }
        }
    }
    if (sums) {
        // per-group sums accumulated by the kernel that wrote x: one thread per group, two 8-byte reads each - no fold
        const double* src = sums + (long)b * 2 * ld_sums;
        for (int grp = tid; grp < groups; grp += GN_THREADS)
            group_mean_rstd(src[grp], src[ld_sums + grp], (double)hw * cpg, eps, g_mean[grp], g_rstd[grp]);
    } else {
        // fold the slab partials: 8 threads per group each sum a strided subset in fp64 (fixed order),
        // then the group's first thread adds the 8 sub-sums in order
        for (int item = tid; item < groups * 8; item += GN_THREADS) {
            const int grp = item >> 3, part = item & 7;
            double a = 0.0, d = 0.0;
            for (int sp = part; sp < nsplit; sp += 8) {
                const float* src = partial + (((long)b * nsplit + sp) * groups + grp) * 2;
                a += (double)src[0];
                d += (double)src[1];
            }
            sub[grp][part][0] = a;
            sub[grp][part][1] = d;
        }
        __syncthreads();
        for (int grp = tid; grp < groups; grp += GN_THREADS) {
            double a = 0.0, d = 0.0;
            for (int part = 0; part < 8; ++part) {
                a += sub[grp][part][0];
                d += sub[grp][part][1];
            }
            group_mean_rstd(a, d, (double)hw * cpg, eps, g_mean[grp], g_rstd[grp]);
        }
    }
    __syncthreads();
    if (rl >= g.rpi) return;
    // y = x * scale + shift with scale = rstd * gamma, shift = beta - mean * scale, kept in registers
    float sc[GN_MAX_SLOTS][EPC], sh[GN_MAX_SLOTS][EPC];
#pragma unroll
    for (int k = 0; k < GN_MAX_SLOTS; ++k) {
        const int chunk = lir + k * g.tpr;
        const bool ok = k < g.slots && chunk < g.nchunks;
        // one division per chunk; the group index then follows the channel index
        int grp = ok ? (chunk * EPC) / cpg : 0;
        int rem = ok ? chunk * EPC - grp * cpg : 0;
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            sc[k][e] = g_rstd[grp] * gam[k][e];
            sh[k][e] = fmaf(-g_mean[grp], sc[k][e], bet[k][e]);
            if (ok && ++rem == cpg) {
                rem = 0;
                ++grp;
            }
        }
    }
    const int r0 = blockIdx.x * rows_per_block, r1 = min(hw, r0 + rows_per_block);
    const T* xb = x + ((long)b * hw) * ldx;
    T* yb = y + ((long)b * hw) * ldy;
    for (int row = r0 + rl; row < r1; row += GN_UNROLL * g.rpi) {
        u32x4 v[GN_UNROLL][GN_MAX_SLOTS];
#pragma unroll
        for (int u = 0; u < GN_UNROLL; ++u) {
            const int rr = row + u * g.rpi;
#pragma unroll
            for (int k = 0; k < GN_MAX_SLOTS; ++k) {
                const int chunk = lir + k * g.tpr;
                v[u][k] = ldg16(xb + (long)rr * ldx + chunk * EPC, rr < r1 && k < g.slots && chunk < g.nchunks);
            }
        }
#pragma unroll
        for (int u = 0; u < GN_UNROLL; ++u) {
            const int rr = row + u * g.rpi;
#pragma unroll
            for (int k = 0; k < GN_MAX_SLOTS; ++k) {
                const int chunk = lir + k * g.tpr;
                if (rr < r1 && k < g.slots && chunk < g.nchunks) {
                    float f[EPC];
                    chunk_to_f32<T>(v[u][k], f);
#pragma unroll
                    for (int e = 0; e < EPC; ++e) {
                        const float o = fmaf(f[e], sc[k][e], sh[k][e]);
                        if constexpr (sizeof(T) == 4) {
                            f[e] = silu ? silu_f(o) : o;
                        } else {
                            // bf16 output: reciprocal instead of the IEEE division sequence
                            f[e] = silu ? o * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * o)) : o;
                        }
                    }
                    *reinterpret_cast<u32x4*>(yb + (long)rr * ldy + chunk * EPC) = f32_to_chunk<T>(f);
                }
            }
        }
    }
}

// One wave per row; the row lives in registers between the mean and the centred-variance pass.
template <typename T, int SLOTS>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* __restrict__ x, long ldx, T* __restrict__ y, long ldy,
                                                        int rows, int c, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps) {
    constexpr int EPC = ElemTraits<T>::EPC;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nchunks = c / EPC;
    float f[SLOTS][EPC];
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < SLOTS; ++k) {
        const int chunk = lane + k * 64;
        if (chunk < nchunks) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(x + (long)row * ldx + chunk * EPC);
            chunk_to_f32<T>(v, f[k]);
#pragma unroll
            for (int e = 0; e < EPC; ++e) sum += f[k][e];
        } else {
#pragma unroll
            for (int e = 0; e < EPC; ++e) f[k][e] = 0.f;
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) sum += __shfl_xor(sum, o);
    const float mean = sum / (float)c;
    float var = 0.f;
#pragma unroll
    for (int k = 0; k < SLOTS; ++k) {
        const int chunk = lane + k * 64;
        if (chunk < nchunks) {
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                const float d = f[k][e] - mean;
                var = fmaf(d, d, var);
            }
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) var += __shfl_xor(var, o);
    const float rstd = 1.0f / sqrtf(var / (float)c + eps);
#pragma unroll
    for (int k = 0; k < SLOTS; ++k) {
        const int chunk = lane + k * 64;
        if (chunk < nchunks) {
            float o[EPC];
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                const int ch = chunk * EPC + e;
                o[e] = (f[k][e] - mean) * rstd * gamma[ch] + beta[ch];
            }
            *reinterpret_cast<u32x4*>(y + (long)row * ldy + chunk * EPC) = f32_to_chunk<T>(o);
        }
    }
}

// Row softmax in place: x[r][:] = softmax(scale * x[r][:]).  One block per row; three sweeps of the row (maximum, sum of
// exponentials, normalised values) - the row (tens of KB) stays in L2 between them.
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_kernel(T* __restrict__ x, long ld, int cols, float scale) {
    constexpr int EPC = ElemTraits<T>::EPC;
    __shared__ float red[4];
    T* row = x + (long)blockIdx.x * ld;
    const int tid = threadIdx.x, nch = cols / EPC;
    auto block_reduce = [&](float v, bool is_max) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const float w = __shfl_xor(v, o);
            v = is_max ? fmaxf(v, w) : v + w;
        }
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = v;
        __syncthreads();
        return is_max ? fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) : (red[0] + red[1]) + (red[2] + red[3]);
    };
    float m = -INFINITY;
    for (int c = tid; c < nch; c += 256) {
        float f[EPC];
        chunk_to_f32<T>(*reinterpret_cast<const u32x4*>(row + c * EPC), f);
#pragma unroll
        for (int e = 0; e < EPC; ++e) m = fmaxf(m, f[e]);
    }
    m = block_reduce(m, true);
    float sum = 0.f;
    for (int c = tid; c < nch; c += 256) {
        float f[EPC];
        chunk_to_f32<T>(*reinterpret_cast<const u32x4*>(row + c * EPC), f);
#pragma unroll
        for (int e = 0; e < EPC; ++e) sum += expf(scale * (f[e] - m));
    }
    sum = block_reduce(sum, false);
    const float inv = 1.0f / sum;
    for (int c = tid; c < nch; c += 256) {
        float f[EPC];
        chunk_to_f32<T>(*reinterpret_cast<const u32x4*>(row + c * EPC), f);
#pragma unroll
        for (int e = 0; e < EPC; ++e) f[e] = expf(scale * (f[e] - m)) * inv;
        *reinterpret_cast<u32x4*>(row + c * EPC) = f32_to_chunk<T>(f);
    }
}

template <typename T> int gn_stats_launch(const void* x, long ldx, int batch, int hw, int c, int groups, float* partial,
                                          int nsplit, hipStream_t s) {
    const GnGeom g = gn_geom<T>(c);
    if (g.slots > GN_MAX_SLOTS) return DM_ERR_ARG;
    hipLaunchKernelGGL(gn_stats_kernel<T>, dim3(nsplit, batch), dim3(GN_THREADS), (size_t)g.rpi * c * 2 * sizeof(float),
                       s, static_cast<const T*>(x), ldx, hw, c, groups, partial, nsplit);
    return dm_check_launch();
}

template <typename T>
int gn_apply_launch(const void* x, long ldx, void* y, long ldy, int batch, int hw, int c, int groups,
                    const float* partial, int nsplit, const double* sums, long ld_sums, const float* gamma,
                    const float* beta, float eps, int silu, hipStream_t s) {
    const GnGeom g = gn_geom<T>(c);
    if (g.slots > GN_MAX_SLOTS || c > GN_MAX_C) return DM_ERR_ARG;
    // enough blocks to fill the chip, at least rpi rows each
    static const int total_blocks = getenv("DM_GN_APPLY_BLOCKS") ? atoi(getenv("DM_GN_APPLY_BLOCKS")) : 1024;  // tuning knob
    int want = (total_blocks + batch - 1) / batch;
    int rows_per_block = (hw + want - 1) / want;
    if (rows_per_block < g.rpi) rows_per_block = g.rpi;
    const int nb = (hw + rows_per_block - 1) / rows_per_block;
    hipLaunchKernelGGL(gn_apply_kernel<T>, dim3(nb, batch), dim3(GN_THREADS), 0, s,
                       static_cast<const T*>(x), ldx, static_cast<T*>(y), ldy, hw, c, groups, partial, nsplit, sums,
                       ld_sums, gamma, beta, eps, silu, rows_per_block);
    return dm_check_launch();
}

template <typename T>
int ln_launch(const void* x, long ldx, void* y, long ldy, int rows, int c, const float* gamma, const float* beta,
              float eps, hipStream_t s) {
    constexpr int EPC = ElemTraits<T>::EPC;
    const int slots = (c / EPC + 63) / 64;
    const dim3 grid((rows + 3) / 4), block(256);
    const T* xp = static_cast<const T*>(x);
    T* yp = static_cast<T*>(y);
#define LN_CASE(S)                                                                                                  \
    case S:                                                                                                         \
        hipLaunchKernelGGL((layernorm_kernel<T, S>), grid, block, 0, s, xp, ldx, yp, ldy, rows, c, gamma, beta, eps); \
        break;
    switch (slots) {
        LN_CASE(1) LN_CASE(2) LN_CASE(3) LN_CASE(4) LN_CASE(5) LN_CASE(6) LN_CASE(7) LN_CASE(8)
        default: return DM_ERR_ARG;
    }
#undef LN_CASE
    return dm_check_launch();
}

// ---- fixed scratch for the allocation-free GroupNorm entry point -------------------------------------------
// Slab partials [batch][nsplit][groups][2] of dm_groupnorm_noalloc live in this file-scope device array instead
// of a caller-provided workspace.  Every device the library runs on holds its own copy (1 MiB) until the process
// ends.  In host code the array's name is only a stand-in for the symbol: the device address is resolved once per
// device with hipGetSymbolAddress and kept in g_gn_scratch_ptr; kernels only ever see that pointer.
constexpr long GN_SCRATCH_FLOATS = 1L << 18;
static __device__ __align__(256) float scratch[GN_SCRATCH_FLOATS];
float* g_gn_scratch_ptr[64] = {};

int gn_scratch_for_current_device(float** out) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return DM_ERR_LAUNCH;
    if (!g_gn_scratch_ptr[dev]) {
        float* ptr = nullptr;
        const hipError_t e = hipGetSymbolAddress(reinterpret_cast<void**>(&ptr), HIP_SYMBOL(scratch));
        if (e != hipSuccess || !ptr) {
            fprintf(stderr, "libdm_hip: hipGetSymbolAddress(scratch): %s\n", hipGetErrorName(e));
            return DM_ERR_LAUNCH;
        }
        g_gn_scratch_ptr[dev] = ptr;
    }
    *out = g_gn_scratch_ptr[dev];
    return DM_OK;
}

bool norm_args_ok(const void* x, long ldx, const void* y, long ldy, int c, int dtype) {
    const int epc = dtype == DM_F32 ? 4 : 8;
    return x && (dtype == DM_F32 || dtype == DM_BF16) && c > 0 && c % epc == 0 && ldx % epc == 0 && dm_aligned16(x) &&
           (!y || (ldy % epc == 0 && dm_aligned16(y)));
}

}  // namespace

extern "C" int dm_groupnorm_stats(const void* x, long ldx, int batch, int hw, int c, int groups, int dtype,
                                  float* partial, int nsplit, hipStream_t stream) {
    if (!norm_args_ok(x, ldx, nullptr, 0, c, dtype) || !partial || batch <= 0 || hw <= 0 || groups <= 0 ||
        groups > 64 || c % groups || nsplit <= 0 || nsplit > hw)
        return DM_ERR_ARG;
    return dtype == DM_F32 ? gn_stats_launch<float>(x, ldx, batch, hw, c, groups, partial, nsplit, stream)
                           : gn_stats_launch<bf16_t>(x, ldx, batch, hw, c, groups, partial, nsplit, stream);
}

extern "C" int dm_groupnorm_apply(const void* x, long ldx, void* y, long ldy, int batch, int hw, int c, int groups,
                                  int dtype, const float* partial, int nsplit, const float* gamma, const float* beta,
                                  float eps, int silu, hipStream_t stream) {
    if (!norm_args_ok(x, ldx, y, ldy, c, dtype) || !y || !partial || !gamma || !beta || batch <= 0 || hw <= 0 ||
        groups <= 0 || groups > 64 || c % groups || nsplit <= 0)
        return DM_ERR_ARG;
    return dtype == DM_F32
               ? gn_apply_launch<float>(x, ldx, y, ldy, batch, hw, c, groups, partial, nsplit, nullptr, 0, gamma, beta, eps, silu, stream)
               : gn_apply_launch<bf16_t>(x, ldx, y, ldy, batch, hw, c, groups, partial, nsplit, nullptr, 0, gamma, beta, eps, silu, stream);
}

extern "C" int dm_groupnorm_apply_sums(const void* x, long ldx, void* y, long ldy, int batch, int hw, int c, int groups,
                                       int dtype, const double* sums, long ld_sums, const float* gamma,
                                       const float* beta, float eps, int silu, hipStream_t stream) {
    if (!norm_args_ok(x, ldx, y, ldy, c, dtype) || !y || !sums || ld_sums < groups || !gamma || !beta || batch <= 0 ||
        hw <= 0 || groups <= 0 || groups > 64 || c % groups || (reinterpret_cast<uintptr_t>(sums) & 7u))
        return DM_ERR_ARG;
    return dtype == DM_F32
               ? gn_apply_launch<float>(x, ldx, y, ldy, batch, hw, c, groups, nullptr, 0, sums, ld_sums, gamma, beta, eps, silu, stream)
               : gn_apply_launch<bf16_t>(x, ldx, y, ldy, batch, hw, c, groups, nullptr, 0, sums, ld_sums, gamma, beta, eps, silu, stream);
}

extern "C" int dm_groupnorm_default_nsplit(int batch, int hw) {
    // slabs of >= 16 rows, about 1024 blocks per launch over the batch, at most 128 slabs per sample
    if (batch <= 0 || hw <= 0) return 0;
    int per_sample = 1024 / batch;
    if (per_sample < 8) per_sample = 8;
    int n = hw / 16;
    if (n > per_sample) n = per_sample;
    if (n > 128) n = 128;
    return n < 1 ? 1 : n;
}

extern "C" int dm_init(void) {
    float* p = nullptr;
    return gn_scratch_for_current_device(&p);
}

extern "C" int dm_groupnorm_noalloc(const void* x, long ldx, void* y, long ldy, int batch, int hw, int c, int groups,
                                    int dtype, const float* gamma, const float* beta, float eps, int silu,
                                    hipStream_t stream) {
    if (!norm_args_ok(x, ldx, y, ldy, c, dtype) || !y || !gamma || !beta || batch <= 0 || hw <= 0 || groups <= 0 ||
        groups > 64 || c % groups)
        return DM_ERR_ARG;
    const int nsplit = dm_groupnorm_default_nsplit(batch, hw);
    if ((long)batch * nsplit * groups * 2 > GN_SCRATCH_FLOATS) return DM_ERR_TOO_LARGE;
    float* partial = nullptr;
    if (const int rc = gn_scratch_for_current_device(&partial)) return rc;
    // every (sample, slab, group) entry the second kernel reads is written by the first: nothing to clear
    int rc = dm_groupnorm_stats(x, ldx, batch, hw, c, groups, dtype, partial, nsplit, stream);
    if (rc != DM_OK) return rc;
    return dm_groupnorm_apply(x, ldx, y, ldy, batch, hw, c, groups, dtype, partial, nsplit, gamma, beta, eps, silu,
                              stream);
}

extern "C" int dm_softmax_rows(void* x, long ld, int rows, int cols, float scale, int dtype, hipStream_t stream) {
    if (!norm_args_ok(x, ld, nullptr, 0, cols, dtype) || rows <= 0 || !(scale > 0.f)) return DM_ERR_ARG;
    if (dtype == DM_F32)
        hipLaunchKernelGGL(softmax_rows_kernel<float>, dim3(rows), dim3(256), 0, stream, static_cast<float*>(x), ld, cols, scale);
    else
        hipLaunchKernelGGL(softmax_rows_kernel<bf16_t>, dim3(rows), dim3(256), 0, stream, static_cast<bf16_t*>(x), ld, cols,
                           scale);
    return dm_check_launch();
}

extern "C" int dm_layernorm(const void* x, long ldx, void* y, long ldy, int rows, int c, int dtype, const float* gamma,
                            const float* beta, float eps, hipStream_t stream) {
    if (!norm_args_ok(x, ldx, y, ldy, c, dtype) || !y || !gamma || !beta || rows <= 0) return DM_ERR_ARG;
    return dtype == DM_F32 ? ln_launch<float>(x, ldx, y, ldy, rows, c, gamma, beta, eps, stream)
                           : ln_launch<bf16_t>(x, ldx, y, ldy, rows, c, gamma, beta, eps, stream);
}
