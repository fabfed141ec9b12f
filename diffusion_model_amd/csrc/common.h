// Shared device helpers for the gfx950 (MI355X, CDNA4) kernels of the denoising hot path.
//
// Conventions used by every kernel in this directory:
//   * wavefront = 64 lanes; block = 256 threads (4 waves) unless stated otherwise
//   * activations are row-major [rows][ld] with the channel dimension contiguous
//     ("NHWC": row = (batch, y, x)); every tensor argument carries its own leading dimension
//   * two storage types: float (fp32 mode, exact f32 MFMA) and bf16 (bf16 mode, f32 accumulate)
//   * a "chunk" is 16 bytes of the contraction dimension: 4 floats or 8 bf16
#pragma once
#include <cstdio>
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

typedef uint16_t bf16_t;  // storage type of a bf16 element

#define DM_WAVE 64

// ---- error codes returned through the C ABI -------------------------------------------------
#define DM_OK 0
#define DM_ERR_ARG 1      // invalid argument / unsupported shape
#define DM_ERR_LAUNCH 2   // HIP reported a launch error
#define DM_ERR_ALIGN 3    // pointer or leading dimension not 16-byte aligned
#define DM_ERR_HEAD_DIM 4 // attention head dimension without a kernel instantiation
#define DM_ERR_TOO_LARGE 5 // problem exceeds the fixed scratch of an allocation-free entry point

// ---- scalar conversions -----------------------------------------------------------------------
__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    __bf16 h = (__bf16)f;  // v_cvt_pk_bf16_f32: round-to-nearest-even, NaN preserved
    return __builtin_bit_cast(bf16_t, h);
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    // one v_cvt_pk_bf16_f32 for the pair (a scalar conversion per half costs two plus an OR)
    typedef float dm_f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 dm_bf16x2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(dm_f32x2{lo, hi}, dm_bf16x2));
}

template <typename T> struct ElemTraits;
template <> struct ElemTraits<float> {
    static constexpr int EPC = 4;  // elements per 16-byte chunk
    __device__ static __forceinline__ float load(const float* p) { return *p; }
    __device__ static __forceinline__ void store(float* p, float v) { *p = v; }
};
template <> struct ElemTraits<bf16_t> {
    static constexpr int EPC = 8;
    __device__ static __forceinline__ float load(const bf16_t* p) { return bf16_to_f32(*p); }
    __device__ static __forceinline__ void store(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

// Expand one 16-byte chunk to floats / pack floats back into a chunk.
template <typename T> __device__ __forceinline__ void chunk_to_f32(const u32x4& c, float* out);
template <> __device__ __forceinline__ void chunk_to_f32<float>(const u32x4& c, float* out) {
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = __uint_as_float(c[i]);
}
template <> __device__ __forceinline__ void chunk_to_f32<bf16_t>(const u32x4& c, float* out) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        out[2 * i] = __uint_as_float(c[i] << 16);
        out[2 * i + 1] = __uint_as_float(c[i] & 0xffff0000u);
    }
}
template <typename T> __device__ __forceinline__ u32x4 f32_to_chunk(const float* in);
template <> __device__ __forceinline__ u32x4 f32_to_chunk<float>(const float* in) {
    u32x4 c;
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] = __float_as_uint(in[i]);
    return c;
}
template <> __device__ __forceinline__ u32x4 f32_to_chunk<bf16_t>(const float* in) {
    u32x4 c;
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] = pack_bf16x2(in[2 * i], in[2 * i + 1]);
    return c;
}

// ---- 16x16 MFMA over one chunk per lane -----------------------------------------------------
// D[i][j] += sum_k A[i][k] * B[k][j] for a 16x16 output tile.  Lane l supplies, as 16 bytes,
// A[i = l&15][chunk l>>4] and B[chunk l>>4][j = l&15]; the four lane groups together cover a
// "k-step" of 64 bytes of the contraction dimension (32 bf16 or 16 floats).
// Result layout: lane l holds D[i = 4*(l>>4) + r][j = l&15] in element r.
//   bf16 : one v_mfma_f32_16x16x32_bf16
//   float: four v_mfma_f32_16x16x4_f32, the e-th using element e of each lane's chunk (a
//          permutation of k that is identical on both operands, so the sum is unchanged)
template <typename T> struct Mma16;
template <> struct Mma16<bf16_t> {
    __device__ static __forceinline__ void mma(f32x4& acc, const u32x4& a, const u32x4& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b),
                                                      acc, 0, 0, 0);
    }
};
template <> struct Mma16<float> {
    __device__ static __forceinline__ void mma(f32x4& acc, const u32x4& a, const u32x4& b) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[e]), __uint_as_float(b[e]), acc, 0, 0, 0);
    }
};

// ---- activations ----------------------------------------------------------------------------------
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// GELU(erf) for outputs that are rounded to bf16: erf by Abramowitz-Stegun 7.1.25 (three terms, |error| <= 2.5e-5: the
// result is off by at most 1.3e-5 |x|, against a bf16 half-ulp of 2e-3 |result|), one rcp + one exp2 and eleven
// instructions in all instead of the branchy libm erff.  The GEGLU epilogue evaluates 84 million of these per forward
// at batch 16 and is bound by VALU issue.  With w = |x| sqrt(log2(e) / 2): exp(-x^2 / 2) = exp2(-w^2), and the
// rational argument 1 / (1 + p |x| / sqrt 2) becomes 1 / (1 + p' w).
__device__ __forceinline__ float gelu_erf_fast_f(float x) {
    const float w = fabsf(x) * 0.84932180028801904272f;                      // sqrt(log2(e) / 2)
    const float t = __builtin_amdgcn_rcpf(fmaf(0.39168235197889934f, w, 1.0f));  // 0.47047 / sqrt(log2 e)
    float poly = fmaf(0.7478556f, t, -0.0958798f);
    poly = fmaf(poly, t, 0.3480242f);
    poly *= t;
    const float e = __builtin_amdgcn_exp2f(-(w * w));
    const float erf_abs = fmaf(-poly, e, 1.0f);  // erf(|x| / sqrt 2)
    const float h = 0.5f * x;
    return fmaf(fabsf(h), erf_abs, h);           // 0.5 x (1 + sign(x) erf_abs)
}

// The same for two values at once, written on 2-vectors so that the multiply-adds compile to packed fp32 instructions
// (v_pk_mul_f32 / v_pk_fma_f32: two lanes of arithmetic per issue slot); only the reciprocal and the exponential stay
// scalar.  Bit-identical to two calls of gelu_erf_fast_f.
__device__ __forceinline__ f32x2 gelu_erf_fast_f2(f32x2 x) {
    const f32x2 ax = __builtin_elementwise_abs(x);
    const f32x2 w = ax * 0.84932180028801904272f;
    const f32x2 d = __builtin_elementwise_fma(f32x2{0.39168235197889934f, 0.39168235197889934f}, w, f32x2{1.0f, 1.0f});
    const f32x2 t = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    f32x2 poly = __builtin_elementwise_fma(f32x2{0.7478556f, 0.7478556f}, t, f32x2{-0.0958798f, -0.0958798f});
    poly = __builtin_elementwise_fma(poly, t, f32x2{0.3480242f, 0.3480242f});
    poly = poly * t;
    const f32x2 nw2 = -(w * w);
    const f32x2 e = {__builtin_amdgcn_exp2f(nw2[0]), __builtin_amdgcn_exp2f(nw2[1])};
    const f32x2 erf_abs = __builtin_elementwise_fma(-poly, e, f32x2{1.0f, 1.0f});
    const f32x2 h = x * 0.5f;
    return __builtin_elementwise_fma(__builtin_elementwise_abs(h), erf_abs, h);
}

// ---- GroupNorm statistics accumulated by the kernel that writes a tensor -----------------------------------------
// A GroupNorm only ever needs per-GROUP sums, so the writer reduces its per-column sums to groups before it touches
// memory: lane l of a wave holds (s, q) = (sum, sum of squares) over the wave's rows of output column col0 + l; the
// columns of one group are consecutive lanes, a segmented inclusive scan over the lanes (no differences of prefix sums,
// so nothing cancels) leaves every group's share in the last lane of its segment, and only those lanes issue the fp64
// atomics: 2 x (groups touched) per wave instead of 2 x 64.  The group of normalised channel c is c / cpg, evaluated as
// (c * magic) >> 32 with magic = ceil(2^32 / cpg) (exact while c * cpg < 2^32).  stats: [batch][2][ld] fp64.
struct DmGroupStats {
    double* sums;   // NULL = no target
    long ld;        // doubles between the (sum | sumsq) rows: at least the number of groups
    int ch0;        // channel index, inside the normalised tensor, of this launch's output column 0
    int cpg;        // channels per group of the GroupNorm that will read the tensor
    uint32_t magic;
    int nch;        // ch0 + N: channels past it are padding columns of the tile and carry zeros
};
__device__ __forceinline__ void group_stats_add(const DmGroupStats& t, long b, int col0, int lane, float s, float q) {
    const int c = t.ch0 + col0 + lane;
    const int g = (int)__umulhi((uint32_t)c, t.magic);
    const int seg0 = g * t.cpg - (t.ch0 + col0);  // first lane of this lane's group (negative: the group started in an earlier tile)
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const float su = __shfl_up(s, d), qu = __shfl_up(q, d);
        if (lane >= d && lane - d >= seg0) {
            s += su;
            q += qu;
        }
    }
    const bool last = lane == 63 || (int)__umulhi((uint32_t)(c + 1), t.magic) != g;
    if (last && g * t.cpg < t.nch) {
        double* dst = t.sums + (b * 2) * t.ld + g;
        atomicAdd(dst, (double)s);
        atomicAdd(dst + t.ld, (double)q);
    }
}
// mean and 1 / sqrt(var + eps) of one group from its fp64 sums over n values (difference of moments in fp64)
__device__ __forceinline__ void group_mean_rstd(double sum, double sumsq, double n, float eps, float& mean_out, float& rstd_out) {
    const double mean = sum / n;
    double var = sumsq / n - mean * mean;
    if (var < 0.0) var = 0.0;
    mean_out = (float)mean;
    rstd_out = (float)(1.0 / sqrt(var + (double)eps));
}

// 16-byte global load with a zero default (predicated-off lanes read nothing).
__device__ __forceinline__ u32x4 ldg16(const void* p, bool pred) {
    u32x4 z = {0u, 0u, 0u, 0u};
    return pred ? *reinterpret_cast<const u32x4*>(p) : z;
}

// Status of the launch just made.  hipGetLastError also returns (and clears) what EARLIER runtime calls of this
// host thread left behind; hipErrorNotReady from an event / stream query by the caller's framework is such a
// leftover and never the result of a launch.  Any other code is reported with its name on stderr.
static inline int dm_check_launch() {
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess || e == hipErrorNotReady) return DM_OK;
    fprintf(stderr, "libdm_hip: %s (%s)\n", hipGetErrorName(e), hipGetErrorString(e));
    return DM_ERR_LAUNCH;
}
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-device setting: remember, per kernel, on which devices
// it has been made (one bit per device ordinal; setting it twice is harmless, so no lock is needed).
struct DmLdsAttr {
    unsigned long long devices = 0;
};
static inline int dm_set_lds_attr(const void* kfn, DmLdsAttr& st, int lds_bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return DM_ERR_LAUNCH;
    if ((st.devices >> dev) & 1ull) return DM_OK;
    const hipError_t e = hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) {
        fprintf(stderr, "libdm_hip: hipFuncSetAttribute(%d bytes of LDS): %s\n", lds_bytes, hipGetErrorName(e));
        return DM_ERR_LAUNCH;
    }
    st.devices |= 1ull << dev;
    return DM_OK;
}
// host side of DmGroupStats: argument check and set-up
static inline bool dm_group_stats_ok(const double* sums, long ld, int cpg, int ch0, int n) {
    return sums && !(reinterpret_cast<uintptr_t>(sums) & 7u) && cpg > 0 && ch0 >= 0 && (long)(ch0 + n) * cpg < (1L << 31) &&
           ld >= (ch0 + n + cpg - 1) / cpg;
}
static inline DmGroupStats dm_group_stats(double* sums, long ld, int cpg, int ch0, int n) {
    DmGroupStats t;
    t.sums = sums;
    t.ld = ld;
    t.ch0 = ch0;
    t.cpg = cpg > 0 ? cpg : 1;
    t.magic = (uint32_t)((0x100000000ULL + (uint64_t)t.cpg - 1) / (uint64_t)t.cpg);
    t.nch = ch0 + n;
    return t;
}
static inline bool dm_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
