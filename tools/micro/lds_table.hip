// ds_read_b128 throughput for host-specified per-lane byte addresses (16 waves on one CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <functional>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const int* addr, int iters, long long* out, uint32_t* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 32768 / 4; i += blockDim.x) reinterpret_cast<uint32_t*>(smem)[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    u32x4 acc = {0, 0, 0, 0};
    int a[8];
    for (int q = 0; q < 8; ++q) a[q] = addr[lane] + q * 4096;
    __syncthreads();
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
    if (threadIdx.x == 0) out[0] = c1 - c0;
    sink[threadIdx.x] = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
}

int main() {
    long long* d; uint32_t* s; int* da;
    (void)hipMalloc(&d, 16); (void)hipMalloc(&s, 4 * 1024); (void)hipMalloc(&da, 256);
    const int waves = 16, iters = 1000;
    auto run = [&](const char* label, std::function<int(int)> f) {
        int h[64];
        for (int i = 0; i < 64; ++i) h[i] = f(i);
        (void)hipMemcpy(da, h, 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64 * waves), 32768, 0, da, iters, d, s);
        (void)hipDeviceSynchronize();
        long long t; (void)hipMemcpy(&t, d, 8, hipMemcpyDeviceToHost);
        printf("%-58s %.2f clk\n", label, (double)t / (iters * 8.0 * waves));
    };
    run("linear i*16", [](int i) { return i * 16; });
    run("D1 (i&15)*16 + (i>>4)*1024  [same slot across 16-groups]", [](int i) { return (i & 15) * 16 + (i >> 4) * 1024; });
    run("D2 (i>>2)*16 + (i&3)*1024   [4 consecutive lanes same slot]", [](int i) { return (i >> 2) * 16 + (i & 3) * 1024; });
    run("D3 (i>>1)*16 + (i&1)*1024   [lane pairs same slot]", [](int i) { return (i >> 1) * 16 + (i & 1) * 1024; });
    run("D4 lanes i,i+8 same slot", [](int i) { return ((i & 7) + 8 * (i >> 4)) * 16 + ((i >> 3) & 1) * 1024; });
    run("D5 lanes i,i+16 same slot", [](int i) { return ((i & 15) + 16 * (i >> 5)) * 16 + ((i >> 4) & 1) * 2048; });
    run("D6 lanes i,i+32 same slot", [](int i) { return (i & 31) * 16 + (i >> 5) * 2048; });
    run("D7 lanes i,i+4 same slot", [](int i) { return ((i & 3) + 4 * (i >> 3)) * 16 + ((i >> 2) & 1) * 1024; });
    for (int a = 0; a < 4; ++a) {
        char lab[96];
        snprintf(lab, sizeof lab, "K 64B rows: row c, slot qp ^ ((c>>%d)&3)", a);
        run(lab, [a](int i) { int c = i & 15, qp = i >> 4; return c * 64 + ((qp ^ ((c >> a) & 3)) << 4); });
    }
    run("K 64B rows no swizzle", [](int i) { int c = i & 15, qp = i >> 4; return c * 64 + (qp << 4); });
    run("K 64B rows: slot qp ^ (c&3) ^ (c>>2)", [](int i) { int c = i & 15, qp = i >> 4; return c * 64 + ((qp ^ (c & 3) ^ (c >> 2)) << 4); });
    run("V 128B rows: slot qp ^ ((c>>1)&7)", [](int i) { int c = i & 15, qp = i >> 4; return c * 128 + ((qp ^ ((c >> 1) & 7)) << 4); });
    run("V 128B rows no swizzle", [](int i) { int c = i & 15, qp = i >> 4; return c * 128 + (qp << 4); });
    run("rows of 80B (pad), c*80 + qp*16", [](int i) { int c = i & 15, qp = i >> 4; return c * 80 + (qp << 4); });
    run("rows of 144B (pad), c*144 + qp*16", [](int i) { int c = i & 15, qp = i >> 4; return c * 144 + (qp << 4); });
    run("rows of 272B (pad), c*272 + qp*16", [](int i) { int c = i & 15, qp = i >> 4; return c * 272 + (qp << 4); });
    return 0;
}
