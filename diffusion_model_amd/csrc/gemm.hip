// LDS-tiled MFMA GEMM / implicit-GEMM conv3x3 for gfx950.
//
//   C[m][n] = epilogue( sum_k A[m][k] * W[n][k] )
//
// A is the activation matrix (rows = pixels/tokens, channel-contiguous).  In conv mode the
// k axis is (tap, cin) and row m = (b, y, x) gathers its 3x3 neighbourhood on the fly from an
// NHWC image (zero padding 1; stride 1 or 2; optional nearest x2 upsampling of the input), so
// no im2col buffer ever exists.  W is a pre-packed weight matrix [Npad][Kpad], k-contiguous,
// zero padded, rows permuted inside every group of 64 so that each lane ends up owning 16
// consecutive output columns (see pack order below).
//
// Block tile BM x BN x (128 bytes of k); 16x16 MFMA fragments; each wave owns a
// (16*FM) x 64 sub-tile.  Operands go global -> registers -> LDS (XOR-swizzled 128-byte rows,
// conflict-free ds_read_b128) with the next k-tile's global loads in flight during the
// current tile's MFMAs; two LDS buffers, one barrier per k-tile.
//
// Packed row order: packed row p = 64*g + 16*j + 4*q + r holds weight row
//   plain : n = 64*g + 16*q + 4*j + r
//   GEGLU : value half (j < 2) / gate half (j >= 2) of output column 32*g + 8*q + 4*(j&1) + r
// so that after the MFMA lane (q = lane>>4) holds columns 64*g + 16*q .. +15 of row m = lane&15
// (GEGLU: 8 value columns and their 8 gates).
#include "common.h"
#include "dm_hip.h"

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
    int epi;
    int splitk;
    float* partial;
};

__device__ __forceinline__ int swz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <typename T, int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64) void gemm_kernel(GemmArgs p) {
    constexpr int NT = WAVES_M * WAVES_N * 64;
    constexpr int FM = BM / (16 * WAVES_M);
    constexpr int FN = 4;
    static_assert(BN == 64 * WAVES_N, "each wave owns 64 output columns");
    static_assert(FM >= 1 && BM % (16 * WAVES_M) == 0, "bad tile");
    constexpr int EPC = ElemTraits<T>::EPC;
    constexpr int BK = 8 * EPC;  // elements in one 128-byte LDS row
    constexpr int A_IT = (BM * 8 + NT - 1) / NT;
    constexpr int W_IT = (BN * 8 + NT - 1) / NT;
    constexpr int TILE_BYTES = (BM + BN) * 128;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WAVES_N;
    const int wn = wave % WAVES_N;

    // ---- tile assignment: give every XCD (blocks b, b+8, ... share one) a contiguous run of
    // logical tiles so neighbouring tiles that share A rows / W columns hit the same L2.
    const int ntm = (p.M + BM - 1) / BM;
    const int ntn = (p.N + BN - 1) / BN;
    const int nblk = ntm * ntn;
    int bid = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7, x = bid & 7;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
    }
    const int tile_m = bid / ntn;
    const int tile_n = bid - tile_m * ntn;
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;

    // ---- split-K range
    const int nkt = p.Kpad / BK;
    int kt0 = 0, kt1 = nkt;
    if (p.splitk > 1) {
        const int per = (nkt + p.splitk - 1) / p.splitk;
        kt0 = blockIdx.y * per;
        kt1 = min(nkt, kt0 + per);
    }

    // ---- per-thread load coordinates (fixed over the k loop)
    const int chunk = tid & 7;
    long a_off[A_IT];   // dense: element offset of the row; conv: pixel index of image b's origin
    int a_y[A_IT], a_x[A_IT];
    bool a_ok[A_IT];
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        const int row = (it * NT + tid) >> 3;
        const int m = m0 + row;
        a_ok[it] = (row < BM) && (m < p.M);
        a_y[it] = a_x[it] = 0;
        a_off[it] = 0;
        if (a_ok[it]) {
            if (p.conv) {
                const int hw = p.Hout * p.Wout;
                const int b = m / hw;
                const int pix = m - b * hw;
                a_y[it] = pix / p.Wout;
                a_x[it] = pix - a_y[it] * p.Wout;
                a_off[it] = (long)b * p.Hin * p.Win;
            } else {
                a_off[it] = (long)m * p.lda;
            }
        }
    }
    const char* w_ptr[W_IT];
#pragma unroll
    for (int it = 0; it < W_IT; ++it) {
        const int row = (it * NT + tid) >> 3;
        w_ptr[it] = p.W + ((long)(n0 + row) * p.Kpad + chunk * EPC) * sizeof(T);
    }
    // conv k-state of this thread's chunk: k = tap * Cin + c
    int kc = kt0 * BK + chunk * EPC;
    int tap = 0;
    if (p.conv) {
        tap = kc / p.Cin;
        kc -= tap * p.Cin;
    }

    u32x4 ra[A_IT], rw[W_IT];
    auto load_tile = [&](int kt) {
        if (p.conv) {
            const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
#pragma unroll
            for (int it = 0; it < A_IT; ++it) {
                int iy, ix;
                bool ok = a_ok[it] && tap < 9;
                if (p.up) {
                    const int uy = a_y[it] + dy, ux = a_x[it] + dx;
                    ok = ok && (unsigned)uy < (unsigned)p.Hout && (unsigned)ux < (unsigned)p.Wout;
                    iy = uy >> 1;
                    ix = ux >> 1;
                } else {
                    iy = a_y[it] * p.stride + dy;
                    ix = a_x[it] * p.stride + dx;
                    ok = ok && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
                }
                const long e = (a_off[it] + (long)iy * p.Win + ix) * p.lda + kc;
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
                ra[it] = ldg16(p.A + (a_off[it] + k) * sizeof(T), a_ok[it] && k < p.K);
        }
#pragma unroll
        for (int it = 0; it < W_IT; ++it) {
            const int row = (it * NT + tid) >> 3;
            rw[it] = ldg16(w_ptr[it] + (long)kt * BK * sizeof(T), row < BN);
        }
    };
    auto store_tile = [&](int buf) {
        char* sa = smem + buf * TILE_BYTES;
        char* sw = sa + BM * 128;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int row = (it * NT + tid) >> 3;
            if (row < BM) *reinterpret_cast<u32x4*>(sa + swz(row, chunk)) = ra[it];
        }
#pragma unroll
        for (int it = 0; it < W_IT; ++it) {
            const int row = (it * NT + tid) >> 3;
            if (row < BN) *reinterpret_cast<u32x4*>(sw + swz(row, chunk)) = rw[it];
        }
    };

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

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
        const char* sw = sa + BM * 128;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            u32x4 af[FM], wf[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i)
                af[i] = *reinterpret_cast<const u32x4*>(sa + swz(wm * 16 * FM + i * 16 + fr, ks * 4 + fq));
#pragma unroll
            for (int j = 0; j < FN; ++j)
                wf[j] = *reinterpret_cast<const u32x4*>(sw + swz(wn * 64 + j * 16 + fr, ks * 4 + fq));
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) Mma16<T>::mma(acc[i][j], wf[j], af[i]);
        }
        if (more) store_tile(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: lane holds, for row m = .. + fr, columns ncol0 + 4*j + r (16 consecutive)
    const int ncol0 = n0 + wn * 64 + fq * 16;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int m = m0 + wm * 16 * FM + i * 16 + fr;
        if (m >= p.M) continue;
        if (p.splitk > 1) {
            float* dst = p.partial + ((long)blockIdx.y * p.M + m) * p.N;
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int n = ncol0 + 4 * j;
                if (n < p.N) *reinterpret_cast<f32x4*>(dst + n) = acc[i][j];
            }
            continue;
        }
        if (p.epi == DM_EPI_GEGLU) {
            const int half = p.N >> 1;
            const int nb = ((n0 + wn * 64) >> 1) + fq * 8;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = nb + 4 * j;
                if (n >= half) continue;
                float o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = acc[i][j][r] + p.bias[n + r];
                    const float g = acc[i][j + 2][r] + p.bias[half + n + r];
                    o[r] = v * gelu_erf_f(g);
                }
                T* dst = reinterpret_cast<T*>(p.C) + (long)m * p.ldc + n;
                if constexpr (sizeof(T) == 4) {
                    *reinterpret_cast<f32x4*>(dst) = f32x4{o[0], o[1], o[2], o[3]};
                } else {
                    *reinterpret_cast<u32x2*>(dst) = u32x2{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
                }
            }
            continue;
        }
        const int bidx = p.rowbias ? m / p.rows_per_batch : 0;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int n = ncol0 + 4 * j;
            if (n >= p.N) continue;
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = acc[i][j][r];
            if (p.bias) {
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] += p.bias[n + r];
            }
            if (p.rowbias) {
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] += p.rowbias[(long)bidx * p.ld_rowbias + n + r];
            }
            if (p.R) {
                const T* rp = reinterpret_cast<const T*>(p.R) + (long)m * p.ldr + n;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] += ElemTraits<T>::load(rp + r);
            }
            if (p.epi == DM_EPI_NCHW_F32) {
                const int hw = p.rows_per_batch;
                const int b = m / hw, pix = m - b * hw;
                float* dst = reinterpret_cast<float*>(p.C);
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[((long)b * p.N + n + r) * hw + pix] = o[r];
            } else {
                T* dst = reinterpret_cast<T*>(p.C) + (long)m * p.ldc + n;
                if constexpr (sizeof(T) == 4) {
                    *reinterpret_cast<f32x4*>(dst) = f32x4{o[0], o[1], o[2], o[3]};
                } else {
                    *reinterpret_cast<u32x2*>(dst) = u32x2{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
                }
            }
        }
    }
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

template <typename T, int BM, int BN, int WAVES_M, int WAVES_N>
int launch_cfg(const GemmArgs& a, hipStream_t stream) {
    constexpr int NT = WAVES_M * WAVES_N * 64;
    constexpr int LDS = 2 * (BM + BN) * 128;
    static bool attr_done = false;
    auto kfn = gemm_kernel<T, BM, BN, WAVES_M, WAVES_N>;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) !=
            hipSuccess)
            return DM_ERR_LAUNCH;
        attr_done = true;
    }
    const int ntm = (a.M + BM - 1) / BM, ntn = (a.N + BN - 1) / BN;
    dim3 grid(ntm * ntn, a.splitk > 1 ? a.splitk : 1, 1);
    hipLaunchKernelGGL(kfn, grid, dim3(NT), LDS, stream, a);
    return dm_check_launch();
}

template <typename T> int launch_typed(const GemmArgs& a, int cfg, hipStream_t s) {
    switch (cfg) {
        case DM_TILE_128x128: return launch_cfg<T, 128, 128, 2, 2>(a, s);
        case DM_TILE_64x128: return launch_cfg<T, 64, 128, 2, 2>(a, s);
        case DM_TILE_64x64: return launch_cfg<T, 64, 64, 4, 1>(a, s);
        case DM_TILE_256x128: return launch_cfg<T, 256, 128, 4, 2>(a, s);
        case DM_TILE_32x64: return launch_cfg<T, 32, 64, 2, 1>(a, s);
        default: return DM_ERR_ARG;
    }
}

int tile_dims(int cfg, int* bm, int* bn) {
    switch (cfg) {
        case DM_TILE_128x128: *bm = 128; *bn = 128; return 0;
        case DM_TILE_64x128: *bm = 64; *bn = 128; return 0;
        case DM_TILE_64x64: *bm = 64; *bn = 64; return 0;
        case DM_TILE_256x128: *bm = 256; *bn = 128; return 0;
        case DM_TILE_32x64: *bm = 32; *bn = 64; return 0;
        default: return 1;
    }
}

}  // namespace

// Tile / split-K choice when the caller passes tile == DM_TILE_AUTO.  Aim for at least ~2 blocks
// per CU (256 CUs) with the largest tile that still gets there; below that, split K.
extern "C" void dm_gemm_plan(int dtype, int M, int N, int K, int epi, int* tile, int* splitk) {
    // Crude cost model (microseconds): blocks are dealt to 256 CUs in rounds; a block's time is
    // its MFMA work over a per-tile efficiency; split-K adds slab traffic and one more launch.
    const int bk = dtype == DM_F32 ? 32 : 64;
    const int nkt = (K + bk - 1) / bk;
    const int cands[4] = {DM_TILE_128x128, DM_TILE_64x128, DM_TILE_64x64, DM_TILE_32x64};
    const double eff_bf16[4] = {1.0, 0.8, 0.55, 0.4};
    const double eff_f32[4] = {1.0, 0.97, 0.92, 0.85};
    const double cu_flops_per_us = dtype == DM_F32 ? 0.55e6 : 5.0e6;  // sustained per CU
    const int sks[8] = {1, 2, 3, 4, 6, 8, 12, 16};
    double best_cost = 1e30;
    int best = DM_TILE_64x64, best_sk = 1;
    for (int c = 0; c < 4; ++c) {
        int bm, bn;
        tile_dims(cands[c], &bm, &bn);
        const double tiles = (double)((M + bm - 1) / bm) * ((N + bn - 1) / bn);
        const double eff = dtype == DM_F32 ? eff_f32[c] : eff_bf16[c];
        for (int s = 0; s < 8; ++s) {
            const int sk = sks[s];
            if (sk > 1 && (epi != DM_EPI_PLAIN || nkt / sk < 4)) break;
            const double blocks = tiles * sk;
            double rounds = blocks / 256.0;
            rounds = rounds <= 1.0 ? 1.0 : (double)((long)(rounds * 2.0 + 0.999)) / 2.0;  // half-round steps
            const double kslice = (double)((nkt + sk - 1) / sk) * bk;
            double cost = rounds * (2.0 * bm * bn * kslice) / (eff * cu_flops_per_us) + 2.0;
            if (sk > 1) cost += 2.0 + (double)sk * M * N * 8.0 / 3.0e6;
            if (cost < best_cost) {
                best_cost = cost;
                best = cands[c];
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

extern "C" long dm_gemm_workspace_bytes(int M, int N, int splitk) {
    return splitk > 1 ? (long)splitk * M * N * (long)sizeof(float) : 0;
}

extern "C" int dm_gemm(const DmGemmDesc* d, hipStream_t stream) {
    if (!d || !d->a || !d->w || !d->c) return DM_ERR_ARG;
    if (d->m <= 0 || d->n <= 0 || d->k <= 0) return DM_ERR_ARG;
    const int esz = d->dtype == DM_F32 ? 4 : (d->dtype == DM_BF16 ? 2 : 0);
    if (!esz) return DM_ERR_ARG;
    const int epc = 16 / esz, bk = 8 * epc;
    if (d->n % 4 || d->kpad % bk || d->kpad < d->k) return DM_ERR_ARG;
    if (!dm_aligned16(d->a) || !dm_aligned16(d->w) || (d->lda % epc)) return DM_ERR_ALIGN;
    if (d->epi != DM_EPI_NCHW_F32 && (!dm_aligned16(d->c) || d->ldc % 4)) return DM_ERR_ALIGN;
    if (d->r && (d->ldr % 4)) return DM_ERR_ALIGN;
    if (d->conv) {
        if (d->cin % epc || d->k != 9 * d->cin) return DM_ERR_ARG;
        if (d->stride != 1 && d->stride != 2) return DM_ERR_ARG;
        if (d->up && (d->stride != 1 || d->hout != 2 * d->hin || d->wout != 2 * d->win)) return DM_ERR_ARG;
        if (!d->up && (d->hout != (d->hin + 2 - 3) / d->stride + 1 || d->wout != (d->win + 2 - 3) / d->stride + 1))
            return DM_ERR_ARG;
        if (d->m % (d->hout * d->wout)) return DM_ERR_ARG;
    } else if (d->k % epc) {
        return DM_ERR_ARG;
    }
    if (d->epi == DM_EPI_GEGLU && (!d->bias || d->n % 64 || d->rowbias || d->r)) return DM_ERR_ARG;
    if ((d->rowbias || d->epi == DM_EPI_NCHW_F32) && d->rows_per_batch <= 0) return DM_ERR_ARG;

    int tile = d->tile, splitk = d->splitk;
    if (tile == DM_TILE_AUTO) dm_gemm_plan(d->dtype, d->m, d->n, d->k, d->epi, &tile, &splitk);
    if (splitk < 1) splitk = 1;
    if (splitk > 1 && (d->epi != DM_EPI_PLAIN || !d->workspace ||
                       d->workspace_bytes < dm_gemm_workspace_bytes(d->m, d->n, splitk)))
        return DM_ERR_ARG;
    int bm, bn;
    if (tile_dims(tile, &bm, &bn)) return DM_ERR_ARG;
    // packed weights must cover every tile row that will be read
    if (d->npad < ((d->n + bn - 1) / bn) * bn) return DM_ERR_ARG;

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
    a.epi = d->epi;
    a.splitk = splitk;
    a.partial = static_cast<float*>(d->workspace);

    int rc = d->dtype == DM_F32 ? launch_typed<float>(a, tile, stream) : launch_typed<bf16_t>(a, tile, stream);
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
