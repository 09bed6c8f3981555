// What does a scattered one-byte store cost on gfx950, and does the XCD's L2 combine such stores before they go to memory?
// N waves store one byte per lane and round to pseudo-random addresses inside a region of R bytes.
//   mode "any":  every workgroup writes anywhere in the region                 (the region is shared by all eight L2s)
//   mode "xcd":  workgroup b writes only into slice b % 8 of the region          (round-robin placement: one L2 per slice)
// 64 M stores in all; time and stores per second for R = 1 MiB ... 256 MiB.
//   hipcc --offload-arch=gfx950 -O3 -o byte_scatter byte_scatter.hip && ./byte_scatter
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ __launch_bounds__(256) void k(uint8_t *buf, uint64_t region, int rounds, int by_xcd) {
    uint32_t x = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    const uint64_t slice = by_xcd ? region / 8 : region;
    uint8_t *base = buf + (by_xcd ? (blockIdx.x & 7u) * slice : 0);
    for (int i = 0; i < rounds; i++) {
        x ^= x << 13, x ^= x >> 17, x ^= x << 5;  // xorshift32
        base[(uint64_t)x % slice] = (uint8_t)i;
    }
}

int main() {
    uint8_t *buf;
    const uint64_t cap = 256ull << 20;
    hipMalloc(&buf, cap);
    hipMemset(buf, 0, cap);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int wgs = 2048, rounds = 128;  // 2048 x 256 lanes x 128 = 64 M stores
    for (int by_xcd = 0; by_xcd < 2; by_xcd++)
        for (uint64_t mb : {1, 2, 4, 8, 16, 32, 64, 128, 256}) {
            const uint64_t region = mb << 20;
            hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, 0, buf, region, rounds, by_xcd);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, 0, buf, region, rounds, by_xcd);
            hipEventRecord(e1, 0);
            hipDeviceSynchronize();
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            const double n = (double)wgs * 256 * rounds;
            printf("%s region %4llu MiB: %7.3f ms for %.0f M byte stores -> %6.1f G stores/s\n", by_xcd ? "xcd" : "any", (unsigned long long)mb, ms, n / 1e6,
                   n / ms / 1e6);
        }
    return 0;
}
