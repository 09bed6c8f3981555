// How many cycles does a wave64 VALU instruction cost a gfx950 SIMD when the SIMD is full of waves?
// Every workgroup is 512 threads = 8 waves, 2 per SIMD; the grid puts 4 of them on every CU (8 waves per SIMD).  Each wave
// runs a loop of 8 x 32 INDEPENDENT instructions of one kind (eight accumulators); lane 0 of every wave stamps s_memtime
// around the loop.  cycles per instruction and SIMD = elapsed cycles / (8 waves x instructions per wave).
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define REP8(X) X X X X X X X X
#define REP32(X) REP8(X) REP8(X) REP8(X) REP8(X)

template <int KIND>
__global__ __launch_bounds__(512) void k(unsigned long long *out, unsigned *sink, int iters) {
    unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    unsigned long long q0 = a0, q1 = a1;
    const unsigned b = blockIdx.x | 1u;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) { REP32(asm volatile("v_add_u32 %0, %0, %8\nv_add_u32 %1, %1, %8\nv_add_u32 %2, %2, %8\nv_add_u32 %3, %3, %8\nv_add_u32 %4, %4, %8\nv_add_u32 %5, %5, %8\nv_add_u32 %6, %6, %8\nv_add_u32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
        if (KIND == 1) { REP32(asm volatile("v_min_u32 %0, %0, %8\nv_min_u32 %1, %1, %8\nv_min_u32 %2, %2, %8\nv_min_u32 %3, %3, %8\nv_min_u32 %4, %4, %8\nv_min_u32 %5, %5, %8\nv_min_u32 %6, %6, %8\nv_min_u32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
        if (KIND == 2) { REP32(asm volatile("v_lshlrev_b64 %0, 1, %0\nv_lshlrev_b64 %1, 1, %1\nv_lshlrev_b64 %0, 1, %0\nv_lshlrev_b64 %1, 1, %1\nv_lshlrev_b64 %0, 1, %0\nv_lshlrev_b64 %1, 1, %1\nv_lshlrev_b64 %0, 1, %0\nv_lshlrev_b64 %1, 1, %1" : "+v"(q0), "+v"(q1));) }
        if (KIND == 3) { REP32(asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\nv_add_u32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\nv_add_u32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\nv_add_u32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\nv_add_u32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\nv_add_u32_dpp %5, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\nv_add_u32_dpp %6, %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\nv_add_u32_dpp %7, %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (KIND == 4) { REP32(asm volatile("v_min_u32_sdwa %0, %0, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1\nv_min_u32_sdwa %1, %1, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1\nv_min_u32_sdwa %2, %2, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1\nv_min_u32_sdwa %3, %3, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1\nv_min_u32_sdwa %4, %4, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1\nv_min_u32_sdwa %5, %5, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1\nv_min_u32_sdwa %6, %6, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1\nv_min_u32_sdwa %7, %7, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
        if (KIND == 5) { REP32(asm volatile("v_fma_f32 %0, %0, %8, %8\nv_fma_f32 %1, %1, %8, %8\nv_fma_f32 %2, %2, %8, %8\nv_fma_f32 %3, %3, %8, %8\nv_fma_f32 %4, %4, %8, %8\nv_fma_f32 %5, %5, %8, %8\nv_fma_f32 %6, %6, %8, %8\nv_fma_f32 %7, %7, %8, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (unsigned)q0 + (unsigned)q1 == 0x12345u) sink[0] = 1;
}

template <int KIND>
void run(const char *name, int wgs_per_cu) {
    const int cus = 256, wgs = cus * wgs_per_cu, iters = 2000, per_iter = 256;
    unsigned long long *d;
    unsigned *sink;
    hipMalloc(&d, wgs * 8 * 8);
    hipMalloc(&sink, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(wgs), dim3(512), 0, 0, d, sink, iters);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<KIND>, dim3(wgs), dim3(512), 0, 0, d, sink, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(wgs * 8);
    hipMemcpy(h.data(), d, wgs * 64, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2];
    const int waves_per_simd = wgs_per_cu * 2;
    printf("%-28s %d waves per SIMD: %.0f ticks for %d instructions per wave -> %.2f ticks per instruction and SIMD; kernel %.3f ms -> %.2f G ticks per s, "
           "%.2f ns per instruction and SIMD\n", name, waves_per_simd, med, iters * per_iter, med / ((double)iters * per_iter * waves_per_simd), ms, med / ms / 1e6,
           ms * 1e6 / ((double)iters * per_iter * waves_per_simd));
    hipFree(d);
    hipFree(sink);
}

int main() {
    for (int w : {4, 1}) {
        run<0>("v_add_u32", w);
        run<1>("v_min_u32", w);
        run<2>("v_lshlrev_b64", w);
        run<3>("v_add_u32_dpp row_shr:1", w);
        run<4>("v_min_u32_sdwa", w);
        run<5>("v_fma_f32", w);
    }
    return 0;
}
