// Which lanes does ds_read_b128 serve in the same LDS cycle?  Only lanes 0 and j are active; they read the
// same banks at different addresses (conflict if served together) or different banks (never a conflict).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(int i0, int j, int same_bank, int iters, long long* out, uint32_t* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 16384 / 4; i += blockDim.x) reinterpret_cast<uint32_t*>(smem)[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    u32x4 acc = {0, 0, 0, 0};
    int a[8];
    for (int q = 0; q < 8; ++q) a[q] = (lane == j ? (same_bank ? 256 : 16) : 0) + q * 1024;
    __syncthreads();
    const long long c0 = clock64();
    if (lane == i0 || lane == j) {
        for (int it = 0; it < iters; ++it) {
            u32x4 v0, v1, v2, v3, v4, v5, v6, v7;
            asm volatile(
                "ds_read_b128 %0, %8\n ds_read_b128 %1, %9\n ds_read_b128 %2, %10\n ds_read_b128 %3, %11\n"
                "ds_read_b128 %4, %12\n ds_read_b128 %5, %13\n ds_read_b128 %6, %14\n ds_read_b128 %7, %15\n"
                "s_waitcnt lgkmcnt(0)\n"
                : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7)
                : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7])
                : "memory");
            acc ^= v0 ^ v1 ^ v2 ^ v3 ^ v4 ^ v5 ^ v6 ^ v7;
        }
    }
    __syncthreads();
    const long long c1 = clock64();
    if (threadIdx.x == 0) out[0] = c1 - c0;
    sink[threadIdx.x] = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
}

int main() {
    long long* d; uint32_t* s;
    (void)hipMalloc(&d, 16); (void)hipMalloc(&s, 4 * 1024);
    const int waves = 16, iters = 1000;
    for (int i0 = 0; i0 < 64; i0 += 20) {
        printf("lane %d shares an LDS cycle with:", i0);
        for (int j = 0; j < 64; ++j) {
            if (j == i0) continue;
            double t[2];
            for (int sb = 0; sb < 2; ++sb) {
                hipLaunchKernelGGL(probe, dim3(1), dim3(64 * waves), 16384, 0, i0, j, sb, iters, d, s);
                (void)hipDeviceSynchronize();
                long long h; (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
                t[sb] = (double)h / (iters * 8.0 * waves);
            }
            if (t[1] > t[0] * 1.15) printf(" %d", j);
        }
        printf("\n");
    }
    return 0;
}
