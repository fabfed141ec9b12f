// Bandwidth-bound pieces of the sampling loop: the fused reverse-process update, the small fp32
// Linears of the time path, layout/dtype conversion at the model boundary, and the hipGraph /
// event plumbing exported through the C ABI.
#include "common.h"
#include "dm_hip.h"

namespace {

constexpr int EW_THREADS = 256;
inline int ew_blocks(long n_items) {
    long b = (n_items + EW_THREADS - 1) / EW_THREADS;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (int)b;
}

// ---- reverse-process update ------------------------------------------------------------------------
// Each arithmetic step is a separately rounded fp32 operation (no FMA contraction), in the order
// the reference's tensor expression evaluates them, so the result is bitwise what fp32 torch gives.
// `out` may alias `x` (the sampling loop updates the latent in place): neither carries __restrict__.
template <bool SECOND_IS_EPS>
__global__ __launch_bounds__(EW_THREADS) void ddpm_step_kernel(const float* x, const float* __restrict__ eps,
                                                                const float* __restrict__ noise, float* out,
                                                                long n, const float* __restrict__ coef,
                                                                const int* __restrict__ step_index, int clip) {
    long shift = 0;
    if (step_index) {
        const int s = *step_index;
        coef += 5 * s;
        shift = (long)s * n;
    }
    const float c_eps = coef[0], c_div = coef[1], c_x0 = coef[2], c_xt = coef[3], c_nz = coef[4];
    const bool add_noise = noise != nullptr && c_nz != 0.0f;
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const f32x4 xv = reinterpret_cast<const f32x4*>(x)[i];
        const f32x4 ev = reinterpret_cast<const f32x4*>(eps)[i];
        f32x4 nv = {0.f, 0.f, 0.f, 0.f};
        if (add_noise) nv = reinterpret_cast<const f32x4*>(noise + shift)[i];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float x0 = __fdiv_rn(__fsub_rn(xv[e], __fmul_rn(c_eps, ev[e])), c_div);
            if (clip) x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
            float prev = __fadd_rn(__fmul_rn(c_x0, x0), __fmul_rn(c_xt, SECOND_IS_EPS ? ev[e] : xv[e]));
            if (add_noise) prev = __fadd_rn(prev, __fmul_rn(c_nz, nv[e]));
            o[e] = prev;
        }
        reinterpret_cast<f32x4*>(out)[i] = o;
    }
    // tail (n not a multiple of 4)
    for (long i = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float x0 = __fdiv_rn(__fsub_rn(x[i], __fmul_rn(c_eps, eps[i])), c_div);
        if (clip) x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
        float prev = __fadd_rn(__fmul_rn(c_x0, x0), __fmul_rn(c_xt, SECOND_IS_EPS ? eps[i] : x[i]));
        if (add_noise) prev = __fadd_rn(prev, __fmul_rn(c_nz, noise[shift + i]));
        out[i] = prev;
    }
}

__global__ __launch_bounds__(EW_THREADS) void scale_add_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                                float* __restrict__ out, long per_sample,
                                                                const float* __restrict__ ta, const float* __restrict__ tc,
                                                                const long long* __restrict__ t, float sign_c) {
    const int b = blockIdx.y;
    const float a = ta[t[b]], c = sign_c * tc[t[b]];
    const long base = (long)b * per_sample;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < per_sample; i += (long)gridDim.x * blockDim.x)
        out[base + i] = __fadd_rn(__fmul_rn(a, x[base + i]), __fmul_rn(c, y[base + i]));
}

__global__ __launch_bounds__(EW_THREADS) void axpbypcz_kernel(const float* __restrict__ x, float alpha,
                                                               const float* __restrict__ y, float beta,
                                                               const float* __restrict__ z, float gamma,
                                                               float* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float v = __fmul_rn(alpha, x[i]);
        if (y) v = __fadd_rn(v, __fmul_rn(beta, y[i]));
        if (z) v = __fadd_rn(v, __fmul_rn(gamma, z[i]));
        out[i] = v;
    }
}

// ---- K-sampler loop body (scripts/inference.py:296-318 with src/sampling/k_euler*.py), per-step scalars in device
// tables indexed by a device-side step counter so that one captured graph serves every step.  Operation order and
// rounding are those of the eager path (dm_axpbypcz launches), hence of the reference's tensor expressions.
// out[c][i] = scale[s] * x[i] for c < copies: the model input, duplicated for classifier-free guidance
__global__ __launch_bounds__(EW_THREADS) void ksampler_input_kernel(const float* __restrict__ x, float* __restrict__ out, long n,
                                                                     int copies, const float* __restrict__ scale,
                                                                     const int* __restrict__ step_index) {
    const float sc = scale[*step_index];
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float v = __fmul_rn(sc, x[i]);
        for (int c = 0; c < copies; ++c) out[(long)c * n + i] = v;
    }
}
// guided = cfg * (cond - uncond) + uncond (eps = [cond | uncond], when do_cfg); out = x + coef[s][0] * guided
// (+ coef[s][1] * noise[s] when that coefficient is non-zero).  `out` may alias `x`.
__global__ __launch_bounds__(EW_THREADS) void ksampler_step_kernel(const float* x, const float* __restrict__ eps,
                                                                    const float* __restrict__ noise, float* out, long n,
                                                                    const float* __restrict__ coef,
                                                                    const int* __restrict__ step_index, float cfg_scale,
                                                                    int do_cfg) {
    const int s = *step_index;
    const float b = coef[2 * s], c = coef[2 * s + 1];
    const bool add_noise = noise != nullptr && c != 0.0f;
    const float* nz = noise ? noise + (long)s * n : nullptr;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float o = eps[i];
        if (do_cfg) {
            const float un = eps[n + i];
            const float diff = __fadd_rn(__fmul_rn(1.0f, o), __fmul_rn(-1.0f, un));
            o = __fadd_rn(__fmul_rn(cfg_scale, diff), __fmul_rn(1.0f, un));
        }
        float v = __fadd_rn(__fmul_rn(1.0f, x[i]), __fmul_rn(b, o));
        if (add_noise) v = __fadd_rn(v, __fmul_rn(c, nz[i]));
        out[i] = v;
    }
}

// Linear multistep update (src/sampling/k_lms.py:113-137) inside the captured loop.  The guided model output o_s goes
// into slot s % ring of a history ring; out = c[s][0] o_s + c[s][1] o_(s-1) + ... over the order[s] newest outputs,
// accumulated in the order of the launch-by-launch path (alpha * x, then 1 * result + c * o per older output).  As in
// the reference, the incoming latents do not enter the result.
__global__ __launch_bounds__(EW_THREADS) void klms_step_kernel(const float* __restrict__ eps, float* hist, float* __restrict__ out,
                                                               long n, const float* __restrict__ coef,
                                                               const int* __restrict__ order, int max_order, int ring,
                                                               const int* __restrict__ step_index, float cfg_scale,
                                                               int do_cfg) {
    const int s = *step_index;
    const int ord = order[s];
    const float* c = coef + (long)s * max_order;
    float* cur = hist + (long)(s % ring) * n;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float o = eps[i];
        if (do_cfg) {
            const float un = eps[n + i];
            const float diff = __fadd_rn(__fmul_rn(1.0f, o), __fmul_rn(-1.0f, un));
            o = __fadd_rn(__fmul_rn(cfg_scale, diff), __fmul_rn(1.0f, un));
        }
        cur[i] = o;
        float v = __fmul_rn(c[0], o);
        for (int j = 1; j < ord; ++j)
            v = __fadd_rn(__fmul_rn(1.0f, v), __fmul_rn(c[j], hist[(long)((s - j) % ring) * n + i]));
        out[i] = v;
    }
}

// ---- time path: y[b][n] = sum_k f(x[b][k]) w[n][k] + bias[n]; one wave per output column ----------
constexpr int LR_BATCH = 8;
__global__ __launch_bounds__(256) void linear_rows_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ y, long ldy,
                                                          int batch, int n, int k, int silu_in) {
    const int lane = threadIdx.x & 63;
    const int col = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (col >= n) return;
    const float* wr = w + (long)col * k;
    for (int b0 = 0; b0 < batch; b0 += LR_BATCH) {
        float acc[LR_BATCH];
#pragma unroll
        for (int i = 0; i < LR_BATCH; ++i) acc[i] = 0.f;
        for (int kk = lane * 4; kk < k; kk += 256) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(wr + kk);
#pragma unroll
            for (int i = 0; i < LR_BATCH; ++i) {
                if (b0 + i < batch) {
                    f32x4 xv = *reinterpret_cast<const f32x4*>(x + (long)(b0 + i) * ldx + kk);
                    if (silu_in) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) xv[e] = silu_f(xv[e]);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i] = fmaf(xv[e], wv[e], acc[i]);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < LR_BATCH; ++i) {
            float v = acc[i];
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
            if (lane == 0 && b0 + i < batch) y[(long)(b0 + i) * ldy + col] = v + (bias ? bias[col] : 0.f);
        }
    }
}

__global__ __launch_bounds__(EW_THREADS) void timestep_embedding_kernel(const long long* __restrict__ t, int batch, int dim,
                                                                        float log_max_period, float* __restrict__ out) {
    const int half = dim >> 1;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < batch * dim; idx += gridDim.x * blockDim.x) {
        const int b = idx / dim, j = idx - b * dim;
        float v = 0.f;
        if (j < 2 * half) {
            const int i = j < half ? j : j - half;
            const float freq = expf(__fdiv_rn(__fmul_rn(-log_max_period, (float)i), (float)half));
            const float arg = __fmul_rn((float)t[b], freq);
            v = j < half ? cosf(arg) : sinf(arg);
        }
        out[idx] = v;
    }
}

// ---- layout conversion: fp32 NCHW  <->  [rows = (b, pixel)][channels] in the engine's dtype ----------
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_rows_kernel(const float* __restrict__ src, int c, int hw, T* __restrict__ dst,
                                                           long ld, int cpad) {
    __shared__ float tile[64][65];
    const int p0 = blockIdx.x * 64, c0 = blockIdx.y * 64, b = blockIdx.z;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {  // i: channel within tile, tx: pixel
        const int ch = c0 + i, p = p0 + tx;
        tile[i][tx] = (ch < c && p < hw) ? src[((long)b * c + ch) * hw + p] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {  // i: pixel within tile, tx: channel
        const int ch = c0 + tx, p = p0 + i;
        if (ch < cpad && p < hw) ElemTraits<T>::store(dst + ((long)b * hw + p) * ld + ch, tile[tx][i]);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void rows_to_nchw_kernel(const T* __restrict__ src, long ld, int c, int hw,
                                                           float* __restrict__ dst) {
    __shared__ float tile[64][65];
    const int p0 = blockIdx.x * 64, c0 = blockIdx.y * 64, b = blockIdx.z;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {  // i: pixel, tx: channel
        const int ch = c0 + tx, p = p0 + i;
        tile[i][tx] = (ch < c && p < hw) ? ElemTraits<T>::load(src + ((long)b * hw + p) * ld + ch) : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {  // i: channel, tx: pixel
        const int ch = c0 + i, p = p0 + tx;
        if (ch < c && p < hw) dst[((long)b * c + ch) * hw + p] = tile[tx][i];
    }
}

__global__ __launch_bounds__(EW_THREADS) void f32_to_bf16_kernel(const float* __restrict__ s, bf16_t* __restrict__ d, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        d[i] = f32_to_bf16(s[i]);
}
__global__ __launch_bounds__(EW_THREADS) void bf16_to_f32_kernel(const bf16_t* __restrict__ s, float* __restrict__ d, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        d[i] = bf16_to_f32(s[i]);
}

__global__ __launch_bounds__(EW_THREADS) void gather_row_kernel(const float* __restrict__ table, long ld,
                                                                 const int* __restrict__ index, float* __restrict__ dst,
                                                                 long ld_dst, int batch, int n) {
    const float* row = table + (long)(*index) * ld;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < batch * n; idx += gridDim.x * blockDim.x) {
        const int b = idx / n, j = idx - b * n;
        dst[(long)b * ld_dst + j] = row[j];
    }
}

// out[(b, s)][:] = table[token[b][s]][:] + pos[s][:], stored in the engine dtype (fp32 or bf16).
// The sum is formed in fp32 from the fp32 tables, as nn.Embedding + the broadcast add do.
template <typename T>
__global__ __launch_bounds__(EW_THREADS) void embed_tokens_kernel(const long long* __restrict__ tokens,
                                                                   const float* __restrict__ table,
                                                                   const float* __restrict__ pos, T* __restrict__ out,
                                                                   long ld, int rows, int seq, int dim, int vocab) {
    const int dim4 = dim >> 2;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < (long)rows * dim4;
         idx += (long)gridDim.x * blockDim.x) {
        const int row = (int)(idx / dim4), j = (int)(idx - (long)row * dim4) << 2;
        long long tok = tokens[row];
        tok = tok < 0 ? 0 : (tok >= vocab ? vocab - 1 : tok);  // the host rejects out-of-range ids; never fault here
        const f32x4 e = *reinterpret_cast<const f32x4*>(table + tok * dim + j);
        const f32x4 q = *reinterpret_cast<const f32x4*>(pos + (long)(row % seq) * dim + j);
        T* dst = out + (long)row * ld + j;
#pragma unroll
        for (int r = 0; r < 4; ++r) ElemTraits<T>::store(dst + r, e[r] + q[r]);
    }
}

// ---- image range / layout (scripts/utils.py:41-53 rescale; scripts/inference.py:337-343 uint8 chain) ----
// t = (x - old_min) / old_span; t = t * new_span + new_min; optional clamp to [lo, hi].  Every step is a
// separately rounded operation in the tensor's own dtype (a bf16 tensor rounds to bf16 after each one), the
// division is a true division, so the result is bitwise what the torch expression gives.
struct RescaleArgs {
    float old_min, old_span, new_span, new_min, lo, hi;
    int clamp;
};
template <typename T> __device__ __forceinline__ float round_as(float v) { return v; }
template <> __device__ __forceinline__ float round_as<bf16_t>(float v) { return bf16_to_f32(f32_to_bf16(v)); }
template <typename T> __device__ __forceinline__ float rescale_one(float v, const RescaleArgs& a) {
    float t = round_as<T>(__fsub_rn(v, a.old_min));
    t = round_as<T>(__fdiv_rn(t, a.old_span));
    t = round_as<T>(__fmul_rn(t, a.new_span));
    t = round_as<T>(__fadd_rn(t, a.new_min));
    if (a.clamp) t = fminf(fmaxf(t, a.lo), a.hi);
    return t;
}

template <typename T>
__global__ __launch_bounds__(EW_THREADS) void rescale_kernel(const T* __restrict__ x, T* __restrict__ y, long n,
                                                              RescaleArgs a) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        ElemTraits<T>::store(y + i, rescale_one<T>(ElemTraits<T>::load(x + i), a));
}

// (B, C, H, W) -> rescaled, clamped, truncated to uint8, stored (B, H, W, C).  One thread per pixel; a wave
// reads 64 consecutive pixels of each channel plane and writes 64*C consecutive bytes.
template <typename T>
__global__ __launch_bounds__(EW_THREADS) void images_to_u8_kernel(const T* __restrict__ x, int c, long hw,
                                                                   unsigned char* __restrict__ out, RescaleArgs a) {
    const int b = blockIdx.y;
    const T* src = x + (long)b * c * hw;
    unsigned char* dst = out + (long)b * c * hw;
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < hw; p += (long)gridDim.x * blockDim.x)
        for (int ch = 0; ch < c; ++ch)
            dst[p * c + ch] = (unsigned char)(int)rescale_one<T>(ElemTraits<T>::load(src + (long)ch * hw + p), a);
}

// Three-channel fp32 images with H*W a multiple of 4: a thread converts 4 pixels (three 16-byte loads,
// one 12-byte store).
__global__ __launch_bounds__(EW_THREADS) void images_to_u8_rgb4_kernel(const float* __restrict__ x, long hw,
                                                                        unsigned char* __restrict__ out, RescaleArgs a) {
    const int b = blockIdx.y;
    const float* src = x + (long)b * 3 * hw;
    uint32_t* dst = reinterpret_cast<uint32_t*>(out + (long)b * 3 * hw);
    const long q = hw >> 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < q; i += (long)gridDim.x * blockDim.x) {
        f32x4 v[3];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) v[ch] = reinterpret_cast<const f32x4*>(src + (long)ch * hw)[i];
        uint32_t bytes[12];
#pragma unroll
        for (int px = 0; px < 4; ++px)
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) bytes[px * 3 + ch] = (uint32_t)(int)rescale_one<float>(v[ch][px], a);
#pragma unroll
        for (int w = 0; w < 3; ++w)
            dst[i * 3 + w] = bytes[4 * w] | (bytes[4 * w + 1] << 8) | (bytes[4 * w + 2] << 16) | (bytes[4 * w + 3] << 24);
    }
}

// (H, W, C) or (B, H, W, C) uint8 pixels -> rescaled floating-point planes (B, C, H, W): the tensor conversion,
// rescale and move_channel(to="first") of prepare_image_tensor (scripts/utils.py:203-212) in one pass.
template <typename T>
__global__ __launch_bounds__(EW_THREADS) void u8_to_images_kernel(const unsigned char* __restrict__ x, int c, long hw,
                                                                   T* __restrict__ out, RescaleArgs a) {
    const int b = blockIdx.y;
    const unsigned char* src = x + (long)b * c * hw;
    T* dst = out + (long)b * c * hw;
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < hw; p += (long)gridDim.x * blockDim.x)
        for (int ch = 0; ch < c; ++ch)
            ElemTraits<T>::store(dst + (long)ch * hw + p, rescale_one<T>((float)src[p * c + ch], a));
}

// VAE encoder tail (src/models/encoder.py:191-212): moments (B, 8, hw) -> (mean + sqrt(exp(clamp(logvar, -30, 20))) * noise) * s
__global__ __launch_bounds__(EW_THREADS) void vae_sample_kernel(const float* __restrict__ moments, const float* __restrict__ noise,
                                                                 float* __restrict__ out, int batch, long hw, float scaling) {
    const long per = 4 * hw;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < batch * per; i += (long)gridDim.x * blockDim.x) {
        const long b = i / per, j = i - b * per;
        const float mean = moments[b * 2 * per + j];
        const float lv = fminf(fmaxf(moments[b * 2 * per + per + j], -30.0f), 20.0f);
        out[i] = __fmul_rn(__fadd_rn(mean, __fmul_rn(sqrtf(expf(lv)), noise[i])), scaling);
    }
}

// Single-lane vector load/store; every other launch on the stream is ordered after it.
__global__ void advance_counter_kernel(int* counter, int delta) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const int v = *reinterpret_cast<volatile int*>(counter);
        *reinterpret_cast<volatile int*>(counter) = v + delta;
    }
}

}  // namespace

extern "C" int dm_abi_version(void) { return DM_ABI_VERSION; }

static int reverse_step_launch(bool second_is_eps, const float* x, const float* eps, const float* noise, float* out,
                               long n, const float* coef, const int* step_index, int clip, hipStream_t stream) {
    if (!x || !eps || !out || !coef || n <= 0) return DM_ERR_ARG;
    if (!dm_aligned16(x) || !dm_aligned16(eps) || !dm_aligned16(out) || (noise && !dm_aligned16(noise)) ||
        (step_index && noise && (n & 3)))
        return DM_ERR_ALIGN;
    const dim3 grid(ew_blocks((n + 3) / 4)), block(EW_THREADS);
    if (second_is_eps)
        hipLaunchKernelGGL(ddpm_step_kernel<true>, grid, block, 0, stream, x, eps, noise, out, n, coef, step_index, clip);
    else
        hipLaunchKernelGGL(ddpm_step_kernel<false>, grid, block, 0, stream, x, eps, noise, out, n, coef, step_index, clip);
    return dm_check_launch();
}

extern "C" int dm_ddpm_step(const float* x, const float* eps, const float* noise, float* out, long n, const float* coef,
                            const int* step_index, int clip, hipStream_t stream) {
    return reverse_step_launch(false, x, eps, noise, out, n, coef, step_index, clip, stream);
}

extern "C" int dm_ddim_step(const float* x, const float* eps, const float* noise, float* out, long n, const float* coef,
                            const int* step_index, int clip, hipStream_t stream) {
    return reverse_step_launch(true, x, eps, noise, out, n, coef, step_index, clip, stream);
}

extern "C" int dm_scale_add(const float* x, const float* y, float* out, int batch, long per_sample, const float* ta,
                            const float* tc, const long long* t, float sign_c, hipStream_t stream) {
    if (!x || !y || !out || !ta || !tc || !t || batch <= 0 || per_sample <= 0) return DM_ERR_ARG;
    hipLaunchKernelGGL(scale_add_kernel, dim3(ew_blocks(per_sample), batch), dim3(EW_THREADS), 0, stream, x, y, out,
                       per_sample, ta, tc, t, sign_c);
    return dm_check_launch();
}

extern "C" int dm_axpbypcz(const float* x, float alpha, const float* y, float beta, const float* z, float gamma,
                           float* out, long n, hipStream_t stream) {
    if (!x || !out || n <= 0) return DM_ERR_ARG;
    hipLaunchKernelGGL(axpbypcz_kernel, dim3(ew_blocks(n)), dim3(EW_THREADS), 0, stream, x, alpha, y, beta, z, gamma, out,
                       n);
    return dm_check_launch();
}

extern "C" int dm_ksampler_input(const float* x, float* out, long n, int copies, const float* scale_table,
                                 const int* step_index, hipStream_t stream) {
    if (!x || !out || !scale_table || !step_index || n <= 0 || copies < 1) return DM_ERR_ARG;
    hipLaunchKernelGGL(ksampler_input_kernel, dim3(ew_blocks(n)), dim3(EW_THREADS), 0, stream, x, out, n, copies,
                       scale_table, step_index);
    return dm_check_launch();
}

extern "C" int dm_ksampler_step(const float* x, const float* eps, const float* noise, float* out, long n,
                                const float* coef, const int* step_index, float cfg_scale, int do_cfg,
                                hipStream_t stream) {
    if (!x || !eps || !out || !coef || !step_index || n <= 0) return DM_ERR_ARG;
    hipLaunchKernelGGL(ksampler_step_kernel, dim3(ew_blocks(n)), dim3(EW_THREADS), 0, stream, x, eps, noise, out, n, coef,
                       step_index, cfg_scale, do_cfg);
    return dm_check_launch();
}

extern "C" int dm_klms_step(const float* eps, float* history, float* out, long n, const float* coef, const int* order,
                            int max_order, int ring, const int* step_index, float cfg_scale, int do_cfg,
                            hipStream_t stream) {
    if (!eps || !history || !out || !coef || !order || !step_index || n <= 0 || max_order < 1 || ring < max_order)
        return DM_ERR_ARG;
    hipLaunchKernelGGL(klms_step_kernel, dim3(ew_blocks(n)), dim3(EW_THREADS), 0, stream, eps, history, out, n, coef, order,
                       max_order, ring, step_index, cfg_scale, do_cfg);
    return dm_check_launch();
}

extern "C" int dm_linear_rows(const float* x, long ldx, const float* w, const float* bias, float* y, long ldy, int batch,
                              int n, int k, int silu_in, hipStream_t stream) {
    if (!x || !w || !y || batch <= 0 || n <= 0 || k <= 0 || (k & 3) || (ldx & 3)) return DM_ERR_ARG;
    if (!dm_aligned16(x) || !dm_aligned16(w)) return DM_ERR_ALIGN;
    hipLaunchKernelGGL(linear_rows_kernel, dim3((n + 3) / 4), dim3(256), 0, stream, x, ldx, w, bias, y, ldy, batch, n, k,
                       silu_in);
    return dm_check_launch();
}

extern "C" int dm_timestep_embedding(const long long* timesteps, int batch, int dim, float max_period, float* out,
                                     hipStream_t stream) {
    if (!timesteps || !out || batch <= 0 || dim <= 0 || max_period <= 0.f) return DM_ERR_ARG;
    hipLaunchKernelGGL(timestep_embedding_kernel, dim3(ew_blocks((long)batch * dim)), dim3(EW_THREADS), 0, stream,
                       timesteps, batch, dim, logf(max_period), out);
    return dm_check_launch();
}

extern "C" int dm_nchw_to_rows(const float* src, int batch, int c, int hw, void* dst, long ld, int cpad, int dtype,
                               hipStream_t stream) {
    if (!src || !dst || batch <= 0 || c <= 0 || hw <= 0 || cpad < c || ld < cpad) return DM_ERR_ARG;
    dim3 grid((hw + 63) / 64, (cpad + 63) / 64, batch);
    if (dtype == DM_F32)
        hipLaunchKernelGGL(nchw_to_rows_kernel<float>, grid, dim3(256), 0, stream, src, c, hw, static_cast<float*>(dst), ld,
                           cpad);
    else if (dtype == DM_BF16)
        hipLaunchKernelGGL(nchw_to_rows_kernel<bf16_t>, grid, dim3(256), 0, stream, src, c, hw, static_cast<bf16_t*>(dst),
                           ld, cpad);
    else
        return DM_ERR_ARG;
    return dm_check_launch();
}

extern "C" int dm_rows_to_nchw(const void* src, long ld, int batch, int c, int hw, float* dst, int dtype,
                               hipStream_t stream) {
    if (!src || !dst || batch <= 0 || c <= 0 || hw <= 0 || ld < c) return DM_ERR_ARG;
    dim3 grid((hw + 63) / 64, (c + 63) / 64, batch);
    if (dtype == DM_F32)
        hipLaunchKernelGGL(rows_to_nchw_kernel<float>, grid, dim3(256), 0, stream, static_cast<const float*>(src), ld, c, hw,
                           dst);
    else if (dtype == DM_BF16)
        hipLaunchKernelGGL(rows_to_nchw_kernel<bf16_t>, grid, dim3(256), 0, stream, static_cast<const bf16_t*>(src), ld, c,
                           hw, dst);
    else
        return DM_ERR_ARG;
    return dm_check_launch();
}

extern "C" int dm_convert(const void* src, int src_dtype, void* dst, int dst_dtype, long n, hipStream_t stream) {
    if (!src || !dst || n <= 0) return DM_ERR_ARG;
    if (src_dtype == DM_F32 && dst_dtype == DM_BF16)
        hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(ew_blocks(n)), dim3(EW_THREADS), 0, stream,
                           static_cast<const float*>(src), static_cast<bf16_t*>(dst), n);
    else if (src_dtype == DM_BF16 && dst_dtype == DM_F32)
        hipLaunchKernelGGL(bf16_to_f32_kernel, dim3(ew_blocks(n)), dim3(EW_THREADS), 0, stream,
                           static_cast<const bf16_t*>(src), static_cast<float*>(dst), n);
    else
        return DM_ERR_ARG;
    return dm_check_launch();
}

extern "C" int dm_gather_row(const float* table, long ld, const int* index, float* dst, long ld_dst, int batch, int n,
                             hipStream_t stream) {
    if (!table || !index || !dst || batch <= 0 || n <= 0) return DM_ERR_ARG;
    hipLaunchKernelGGL(gather_row_kernel, dim3(ew_blocks((long)batch * n)), dim3(EW_THREADS), 0, stream, table, ld, index,
                       dst, ld_dst, batch, n);
    return dm_check_launch();
}

extern "C" int dm_embed_tokens(const long long* tokens, const float* table, const float* pos, void* out, long ld,
                               int rows, int seq, int dim, int vocab, int dtype, hipStream_t stream) {
    if (!tokens || !table || !pos || !out || rows <= 0 || seq <= 0 || dim <= 0 || dim % 4 || vocab <= 0 || ld < dim)
        return DM_ERR_ARG;
    if (!dm_aligned16(table) || !dm_aligned16(pos)) return DM_ERR_ALIGN;
    const dim3 grid(ew_blocks((long)rows * (dim / 4))), block(EW_THREADS);
    if (dtype == DM_F32)
        hipLaunchKernelGGL(embed_tokens_kernel<float>, grid, block, 0, stream, tokens, table, pos,
                           static_cast<float*>(out), ld, rows, seq, dim, vocab);
    else if (dtype == DM_BF16)
        hipLaunchKernelGGL(embed_tokens_kernel<bf16_t>, grid, block, 0, stream, tokens, table, pos,
                           static_cast<bf16_t*>(out), ld, rows, seq, dim, vocab);
    else
        return DM_ERR_ARG;
    return dm_check_launch();
}

extern "C" int dm_rescale(const void* x, void* y, long n, int dtype, float old_min, float old_span, float new_span,
                          float new_min, int clamp, float lo, float hi, hipStream_t stream) {
    if (!x || !y || n <= 0 || old_span == 0.0f) return DM_ERR_ARG;
    const RescaleArgs a{old_min, old_span, new_span, new_min, lo, hi, clamp};
    if (dtype == DM_F32)
        hipLaunchKernelGGL(rescale_kernel<float>, dim3(ew_blocks(n)), dim3(EW_THREADS), 0, stream,
                           static_cast<const float*>(x), static_cast<float*>(y), n, a);
    else if (dtype == DM_BF16)
        hipLaunchKernelGGL(rescale_kernel<bf16_t>, dim3(ew_blocks(n)), dim3(EW_THREADS), 0, stream,
                           static_cast<const bf16_t*>(x), static_cast<bf16_t*>(y), n, a);
    else
        return DM_ERR_ARG;
    return dm_check_launch();
}

extern "C" int dm_images_to_u8(const void* x, int dtype, int batch, int c, long hw, float old_min, float old_span,
                               float new_span, float new_min, float lo, float hi, unsigned char* out,
                               hipStream_t stream) {
    if (!x || !out || batch <= 0 || batch > 65535 || c <= 0 || hw <= 0 || old_span == 0.0f) return DM_ERR_ARG;
    if (!(lo >= 0.0f && hi <= 255.0f && lo <= hi)) return DM_ERR_ARG;  // the clamp is what keeps the uint8 cast defined
    const RescaleArgs a{old_min, old_span, new_span, new_min, lo, hi, 1};
    if (dtype == DM_F32 && c == 3 && hw % 4 == 0 && dm_aligned16(x) && (reinterpret_cast<uintptr_t>(out) & 3u) == 0) {
        hipLaunchKernelGGL(images_to_u8_rgb4_kernel, dim3(ew_blocks(hw / 4), batch), dim3(EW_THREADS), 0, stream,
                           static_cast<const float*>(x), hw, out, a);
    } else if (dtype == DM_F32) {
        hipLaunchKernelGGL(images_to_u8_kernel<float>, dim3(ew_blocks(hw), batch), dim3(EW_THREADS), 0, stream,
                           static_cast<const float*>(x), c, hw, out, a);
    } else if (dtype == DM_BF16) {
        hipLaunchKernelGGL(images_to_u8_kernel<bf16_t>, dim3(ew_blocks(hw), batch), dim3(EW_THREADS), 0, stream,
                           static_cast<const bf16_t*>(x), c, hw, out, a);
    } else {
        return DM_ERR_ARG;
    }
    return dm_check_launch();
}

extern "C" int dm_u8_to_images(const unsigned char* x, int batch, int c, long hw, float old_min, float old_span,
                               float new_span, float new_min, void* out, int dtype, hipStream_t stream) {
    if (!x || !out || batch <= 0 || batch > 65535 || c <= 0 || hw <= 0 || old_span == 0.0f) return DM_ERR_ARG;
    const RescaleArgs a{old_min, old_span, new_span, new_min, 0.f, 0.f, 0};
    if (dtype == DM_F32)
        hipLaunchKernelGGL(u8_to_images_kernel<float>, dim3(ew_blocks(hw), batch), dim3(EW_THREADS), 0, stream, x, c, hw,
                           static_cast<float*>(out), a);
    else if (dtype == DM_BF16)
        hipLaunchKernelGGL(u8_to_images_kernel<bf16_t>, dim3(ew_blocks(hw), batch), dim3(EW_THREADS), 0, stream, x, c, hw,
                           static_cast<bf16_t*>(out), a);
    else
        return DM_ERR_ARG;
    return dm_check_launch();
}

extern "C" int dm_vae_sample(const float* moments, const float* noise, float* out, int batch, long hw, float scaling,
                             hipStream_t stream) {
    if (!moments || !noise || !out || batch <= 0 || hw <= 0) return DM_ERR_ARG;
    hipLaunchKernelGGL(vae_sample_kernel, dim3(ew_blocks((long)batch * 4 * hw)), dim3(EW_THREADS), 0, stream, moments, noise,
                       out, batch, hw, scaling);
    return dm_check_launch();
}

extern "C" int dm_zero(void* ptr, long bytes, hipStream_t stream) {
    if (!ptr || bytes <= 0) return DM_ERR_ARG;
    return hipMemsetAsync(ptr, 0, (size_t)bytes, stream) == hipSuccess ? DM_OK : DM_ERR_LAUNCH;
}

extern "C" int dm_advance_counter(int* counter, int delta, hipStream_t stream) {
    if (!counter) return DM_ERR_ARG;
    hipLaunchKernelGGL(advance_counter_kernel, dim3(1), dim3(64), 0, stream, counter, delta);
    return dm_check_launch();
}

// ---- hipGraph capture / replay -----------------------------------------------------------------------
extern "C" int dm_graph_begin(hipStream_t stream) {
    return hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal) == hipSuccess ? DM_OK : DM_ERR_LAUNCH;
}

extern "C" int dm_graph_end(hipStream_t stream, void** graph_exec_out) {
    if (!graph_exec_out) return DM_ERR_ARG;
    hipGraph_t graph = nullptr;
    if (hipStreamEndCapture(stream, &graph) != hipSuccess || !graph) return DM_ERR_LAUNCH;
    hipGraphExec_t exec = nullptr;
    const hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    hipGraphDestroy(graph);
    if (e != hipSuccess) return DM_ERR_LAUNCH;
    *graph_exec_out = exec;
    return DM_OK;
}

extern "C" int dm_graph_launch(void* graph_exec, hipStream_t stream) {
    if (!graph_exec) return DM_ERR_ARG;
    return hipGraphLaunch(static_cast<hipGraphExec_t>(graph_exec), stream) == hipSuccess ? DM_OK : DM_ERR_LAUNCH;
}

extern "C" int dm_graph_destroy(void* graph_exec) {
    if (!graph_exec) return DM_ERR_ARG;
    return hipGraphExecDestroy(static_cast<hipGraphExec_t>(graph_exec)) == hipSuccess ? DM_OK : DM_ERR_LAUNCH;
}

// ---- events ------------------------------------------------------------------------------------------
extern "C" int dm_event_create(void** ev) {
    if (!ev) return DM_ERR_ARG;
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return DM_ERR_LAUNCH;
    *ev = e;
    return DM_OK;
}
extern "C" int dm_event_record(void* ev, hipStream_t stream) {
    return hipEventRecord(static_cast<hipEvent_t>(ev), stream) == hipSuccess ? DM_OK : DM_ERR_LAUNCH;
}
extern "C" int dm_event_elapsed_ms(void* start, void* stop, float* ms) {
    if (!ms) return DM_ERR_ARG;
    if (hipEventSynchronize(static_cast<hipEvent_t>(stop)) != hipSuccess) return DM_ERR_LAUNCH;
    return hipEventElapsedTime(ms, static_cast<hipEvent_t>(start), static_cast<hipEvent_t>(stop)) == hipSuccess
               ? DM_OK
               : DM_ERR_LAUNCH;
}
extern "C" int dm_event_destroy(void* ev) {
    return hipEventDestroy(static_cast<hipEvent_t>(ev)) == hipSuccess ? DM_OK : DM_ERR_LAUNCH;
}
