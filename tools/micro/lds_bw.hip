// LDS read bandwidth of one CU for the fragment-read pattern of the GEMM ring loop (ds_read_b128, 16 rows of 128 B per
// 16 lanes, XOR-swizzled chunks), with 4 or 8 waves per CU reading at once and nothing else running; and the same with the
// k-tile's 32 bf16 MFMAs between the bursts, as the loop has them.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/lds_bw tools/micro/lds_bw.hip && tools/micro/lds_bw
#include <hip/hip_runtime.h>

#include <cstdio>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ int swz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <bool MFMA>
__global__ __launch_bounds__(256, 2) void lds_kernel(unsigned* sink, long long* cycles, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // 32 KB: A tile 128 rows, W tile 128 rows
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4, wm = wave >> 1, wn = wave & 1;
    for (int i = tid; i < 8192; i += 256) reinterpret_cast<unsigned*>(smem)[i] = i * 2654435761u;
    __syncthreads();
    const char* sa = smem;
    const char* sw = smem + 128 * 128;
    f32x4 acc[4][4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 x = {0, 0, 0, 0};
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        u32x4 af[2][4], wf[2][4];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int j = 0; j < 4; ++j) wf[ks][j] = *reinterpret_cast<const u32x4*>(sw + swz(wn * 64 + j * 16 + fr, ks * 4 + fq));
#pragma unroll
            for (int i = 0; i < 4; ++i) af[ks][i] = *reinterpret_cast<const u32x4*>(sa + swz(wm * 64 + i * 16 + fr, ks * 4 + fq));
        }
        __builtin_amdgcn_sched_barrier(0);
        if (MFMA) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[ks][j]),
                                                                            __builtin_bit_cast(bf16x8, af[ks][i]), acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i) x ^= af[ks][i] ^ wf[ks][i];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    const long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
    if ((x[0] ^ x[1] ^ x[2] ^ x[3]) == 0x12345u || s == 1.2345f) sink[0] = 1;
    if (tid == 0) cycles[blockIdx.x] = t1 - t0;
}

template <bool MFMA> void run(const char* label, int blocks_per_cu, unsigned* sink, long long* cyc, int cus) {
    const int iters = 2000, blocks = cus * blocks_per_cu;
    hipFuncSetAttribute(reinterpret_cast<const void*>(lds_kernel<MFMA>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    // 64 KB of dynamic LDS per block: two blocks per CU at most, as in the ring loop
    hipLaunchKernelGGL((lds_kernel<MFMA>), dim3(blocks), dim3(256), 65536, 0, sink, cyc, iters);
    hipLaunchKernelGGL((lds_kernel<MFMA>), dim3(blocks), dim3(256), 65536, 0, sink, cyc, iters);
    hipDeviceSynchronize();
    long long* h = new long long[blocks];
    hipMemcpy(h, cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
    double sum = 0;
    for (int i = 0; i < blocks; ++i) sum += (double)h[i];
    const double per_iter = sum / blocks / iters;
    // per iteration and block: 4 waves x 16 reads x 1 KB = 64 KB; clock64 counts shader cycles
    printf("%-34s %d block(s)/CU: %8.1f cycles per k-tile and block = %5.1f B/clk of fragment reads per CU\n", label, blocks_per_cu,
           per_iter, blocks_per_cu * 65536.0 / per_iter);
    delete[] h;
}

int main() {
    unsigned* sink;
    long long* cyc;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    hipMalloc(&sink, 64);
    hipMalloc(&cyc, cus * 2 * sizeof(long long));
    for (int b : {1, 2}) {
        run<false>("16 ds_read_b128 per wave", b, sink, cyc, cus);
        run<true>("16 ds_read_b128 + 32 MFMA per wave", b, sink, cyc, cus);
    }
    printf("(a k-tile's 32 MFMAs are 512 cycles of the matrix pipe per wave; 64 KB of fragments per block are 256 cycles at 256 B/clk)\n");
    hipFree(sink);
    hipFree(cyc);
    return 0;
}
