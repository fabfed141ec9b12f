// Cycle cost of ds_read_b128 address patterns (one wave, dependent-free stream of reads).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ int addr_of(int pat, int lane, int j) {
    const int c = lane & 15, qp = lane >> 4;
    switch (pat) {
        case 0: return lane * 16 + j * 1024;                                   // linear
        case 1: {                                                               // K image, permuted keys, 64-B rows
            const int kt = j & 3;
            const int row = 32 * (kt >> 1) + 8 * (c >> 2) + 4 * (kt & 1) + (c & 3);
            return row * 64 + ((qp ^ ((row >> 3) & 3)) << 4);
        }
        case 2: {                                                               // V^T image, 128-B rows
            const int row = (j % 3) * 16 + c;
            return row * 128 + ((((j & 1) * 4 + qp) ^ ((row >> 1) & 7)) << 4);
        }
        case 3: {                                                               // K image, identity keys
            const int row = (j & 3) * 16 + c;
            return row * 64 + ((qp ^ ((row >> 2) & 3)) << 4);
        }
        case 4: {                                                               // 64-B rows, no swizzle
            const int row = (j & 3) * 16 + c;
            return row * 64 + (qp << 4);
        }
        case 5: {  // K image 64-B rows, lanes: row = lane>>2 fast chunk
            const int row = (j & 3) * 16 + (lane >> 2);
            return row * 64 + ((lane & 3) << 4);
        }
        case 6: {  // 80-byte pixel pitch, 16 consecutive pixels per 16 lanes, chunk qp (activation halo image)
            return ((j & 3) * 16 + c + (j >> 2)) * 80 + (qp << 4);
        }
        // 64-byte pixel slots, 16 consecutive slots per 16 lanes starting at slot s0 = pat & 3 (patterns 8 ..): which swizzle
        // or pitch keeps ds_read_b128 at full rate?
        case 8: case 9: case 10: case 11: {  // XOR with slot & 3
            const int slot = (j & 3) * 16 + c + (pat & 3);
            return slot * 64 + ((qp ^ (slot & 3)) << 4);
        }
        case 12: case 13: case 14: case 15: {  // XOR with (slot >> 1) & 3
            const int slot = (j & 3) * 16 + c + (pat & 3);
            return slot * 64 + ((qp ^ ((slot >> 1) & 3)) << 4);
        }
        case 16: case 17: case 18: case 19: {  // pitch 96
            const int slot = (j & 3) * 16 + c + (pat & 3);
            return slot * 96 + (qp << 4);
        }
        case 20: case 21: case 22: case 23: {  // pitch 112
            const int slot = (j & 3) * 16 + c + (pat & 3);
            return slot * 112 + (qp << 4);
        }
        case 24: case 25: case 26: case 27: {  // pitch 144
            const int slot = (j & 3) * 16 + c + (pat & 3);
            return slot * 144 + (qp << 4);
        }
        case 28: case 29: case 30: case 31: {  // 128-byte rows, ring swizzle, odd starts
            const int row = (j & 3) * 16 + c + (pat & 3);
            return row * 128 + (((qp + 4 * (j >> 2)) ^ ((row >> 1) & 7)) << 4);
        }
        case 7: {  // 128-byte rows with the GEMM ring's XOR swizzle (reference point)
            const int row = (j & 3) * 16 + c;
            return row * 128 + (((qp + 4 * (j >> 2)) ^ ((row >> 1) & 7)) << 4);
        }
    }
    return 0;
}

__global__ void probe(int pat, int iters, long long* out, uint32_t* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 16384 / 4; i += blockDim.x) reinterpret_cast<uint32_t*>(smem)[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    int a[8];
    for (int j = 0; j < 8; ++j) a[j] = addr_of(pat, lane, j);
    u32x4 acc = {0, 0, 0, 0};
    __syncthreads();
    const long long t0 = wall_clock64();
    const long long c0 = clock64();
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
    __syncthreads();
    const long long c1 = clock64();
    const long long t1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = t1 - t0; }
    sink[threadIdx.x] = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
}

int main() {
    long long* d; uint32_t* s;
    hipMalloc(&d, 16); hipMalloc(&s, 4 * 1024);
    for (int waves = 16; waves <= 16; waves *= 4)
        for (int pat = 0; pat < 32; ++pat) {
            hipLaunchKernelGGL(probe, dim3(1), dim3(64 * waves), 16384, 0, pat, 2000, d, s);
            hipDeviceSynchronize();
            long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
            printf("waves %d pattern %d: %.2f clk per ds_read_b128 per wave; %.2f clk per wave-instruction for the CU\n", waves, pat, (double)h[0] / (2000.0 * 8), (double)h[0] / (2000.0 * 8 * waves));
        }
    return 0;
}
