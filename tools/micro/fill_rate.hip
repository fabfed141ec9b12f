// Rate at which a CU can fill LDS from L2: LDS-DMA (global_load_lds_dwordx4) against register staging
// (global_load_dwordx4 + ds_write_b128), for the shape of the GEMM ring loop: 256 threads, 32 KB per step, two blocks per
// CU, one barrier per step.  The source is a 2 MB buffer shared by every block, so all reads are L2 hits.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/fill_rate tools/micro/fill_rate.hip && tools/micro/fill_rate
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int NT = 256, PIECES = 8, STEP_BYTES = NT * 16 * PIECES;  // 32 KB per step
constexpr int SRC_BYTES = 2 << 20;

template <int MODE, int STAGES>
__global__ __launch_bounds__(NT, 2) void fill_kernel(const char* src, unsigned* sink, int steps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned off = (blockIdx.x * 40960u + tid * 16u) & (SRC_BYTES - 1);
    unsigned acc = 0;
    for (int s = 0; s < steps; ++s) {
        char* stage = smem + (s % STAGES) * STEP_BYTES;
        if (MODE == 0) {
#pragma unroll
            for (int p = 0; p < PIECES; ++p)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + ((off + p * 4096u) & (SRC_BYTES - 1))),
                                                 (__attribute__((address_space(3))) void*)(stage + p * 4096 + wave * 1024), 16, 0, 0);
        } else {
            u32x4 r[PIECES];
#pragma unroll
            for (int p = 0; p < PIECES; ++p) r[p] = *reinterpret_cast<const u32x4*>(src + ((off + p * 4096u) & (SRC_BYTES - 1)));
#pragma unroll
            for (int p = 0; p < PIECES; ++p) *reinterpret_cast<u32x4*>(stage + p * 4096 + tid * 16) = r[p];
        }
        off = (off + STEP_BYTES) & (SRC_BYTES - 1);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        acc += *reinterpret_cast<const unsigned*>(stage + ((tid * 68) & (STEP_BYTES - 4)));  // keep the fill observable
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE, int STAGES> double run(const char* src, unsigned* sink, int blocks, int steps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipFuncSetAttribute(reinterpret_cast<const void*>(fill_kernel<MODE, STAGES>), hipFuncAttributeMaxDynamicSharedMemorySize,
                        STAGES * STEP_BYTES);
    hipLaunchKernelGGL((fill_kernel<MODE, STAGES>), dim3(blocks), dim3(NT), STAGES * STEP_BYTES, 0, src, sink, steps);
    hipEventRecord(e0);
    hipLaunchKernelGGL((fill_kernel<MODE, STAGES>), dim3(blocks), dim3(NT), STAGES * STEP_BYTES, 0, src, sink, steps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    char* src;
    unsigned* sink;
    hipMalloc(&src, SRC_BYTES);
    hipMalloc(&sink, 64);
    hipMemset(src, 1, SRC_BYTES);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount, steps = 4000;
    for (int per_cu : {1, 2}) {
        const int blocks = cus * per_cu;
        const double bytes = (double)blocks * steps * STEP_BYTES;
        const double t0 = run<0, 2>(src, sink, blocks, steps), t1 = run<1, 2>(src, sink, blocks, steps);
        printf("%d block(s) per CU, 32 KB per step and block, one barrier per step (no step in flight behind the barrier):\n", per_cu);
        printf("  LDS-DMA          %8.1f us  %6.1f GB/s per CU  %5.1f TB/s chip\n", t0 * 1e3, bytes / (t0 * 1e-3) / cus / 1e9, bytes / (t0 * 1e-3) / 1e12);
        printf("  register staged  %8.1f us  %6.1f GB/s per CU  %5.1f TB/s chip\n", t1 * 1e3, bytes / (t1 * 1e-3) / cus / 1e9, bytes / (t1 * 1e-3) / 1e12);
    }
    hipFree(src);
    hipFree(sink);
    return 0;
}
