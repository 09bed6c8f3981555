// What does a DEPENDENT instruction cost a lone wave on a gfx950 SIMD -- the price list of the spine's walker, whose halving
// search is one long dependent chain that crosses between the vector and the scalar unit several times per halving.
// One workgroup of one wave, alone on the chip; every pattern is a chain in which each step needs the result of the one
// before; lane 0 stamps s_memtime around N repetitions.  Reported: shader cycles per repetition of the pattern.
//   hipcc --offload-arch=gfx950 -O3 -o dep_latency dep_latency.hip && ./dep_latency
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP4(X) X X X X
#define REP16(X) REP4(X) REP4(X) REP4(X) REP4(X)

__global__ __launch_bounds__(64) void k(int pattern, unsigned long long *out, unsigned *sink) {
    __shared__ unsigned table[256];
    for (unsigned i = threadIdx.x; i < 256; i += 64) table[i] = ((i + 1) & 255u) * 4u;  // a ring of LDS addresses
    __syncthreads();
    unsigned a = threadIdx.x + 1, b = 1, c = threadIdx.x * 4;
    __builtin_amdgcn_s_setprio(3);
    const int N = 256;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < N; it++) {
        switch (pattern) {
            case 0:  // vector -> vector
                REP16(asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));)
                break;
            case 1:  // scalar -> scalar
                REP16(asm volatile("s_add_u32 s20, s20, 1" ::: "s20", "scc");)
                break;
            case 2:  // vector -> readlane -> vector (a value through an SGPR)
                REP16(asm volatile("v_readlane_b32 s20, %0, 0\n v_add_u32 %0, s20, %0" : "+v"(a) :: "s20");)
                break;
            case 3:  // vector -> compare -> scalar find-first -> vector
                REP16(asm volatile("v_cmp_lt_u32 vcc, %1, %0\n s_ff1_i32_b64 s20, vcc\n v_add_u32 %0, s20, %0" : "+v"(a) : "v"(b) : "vcc", "s20");)
                break;
            case 4:  // vector -> compare -> s_and -> find-first -> readlane with that lane -> vector (the search step of the walker)
                REP16(asm volatile("v_cmp_lt_u32 vcc, %1, %0\n s_and_b64 s[22:23], vcc, exec\n s_ff1_i32_b64 s20, s[22:23]\n s_nop 3\n v_readlane_b32 s21, %0, s20\n v_add_u32 %0, s21, %0"
                                   : "+v"(a) : "v"(b) : "vcc", "s20", "s21", "s22", "s23");)
                break;
            case 5:  // dependent LDS read
                REP16(asm volatile("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)" : "+v"(c));)
                break;
            case 6:  // scalar index -> vector address -> LDS read -> readlane -> scalar (fetch a row chosen by a search)
                REP16(asm volatile("v_readlane_b32 s20, %0, 0\n s_and_b32 s20, s20, 0x3fc\n v_mov_b32 %0, s20\n ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)" : "+v"(c) :: "s20");)
                break;
            case 7:  // six readlanes of one register, six subtractions with them, min3 min3 min: the walker's record test up to the compare
                REP16(asm volatile("v_readlane_b32 s20, %0, 0\n v_readlane_b32 s21, %0, 1\n v_readlane_b32 s22, %0, 2\n v_readlane_b32 s23, %0, 3\n v_readlane_b32 s24, %0, 4\n v_readlane_b32 s25, %0, 5\n"
                                   "v_sub_u32 v40, %1, s20\n v_sub_u32 v41, %1, s21\n v_sub_u32 v42, %1, s22\n v_sub_u32 v43, %1, s23\n v_sub_u32 v44, %1, s24\n v_sub_u32 v45, %1, s25\n"
                                   "v_min3_i32 v40, v40, v41, v42\n v_min3_i32 v43, v43, v44, v45\n v_min_i32 %0, v40, v43"
                                   : "+v"(a) : "v"(b) : "s20", "s21", "s22", "s23", "s24", "s25", "v40", "v41", "v42", "v43", "v44", "v45");)
                break;
            case 8:  // DPP move -> vector
                REP16(asm volatile("s_nop 1\n v_mov_b32_dpp %0, %0 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));)
                break;
            case 9:  // packed subtract -> or3 -> compare -> vcc -> s_and -> ff1 (inside-the-record search up to the index)
                REP16(asm volatile("v_pk_sub_u16 v40, %1, %0 clamp\n v_or3_b32 v40, v40, %1, %1\n v_cmp_eq_u32 vcc, 0, v40\n s_and_b32 s20, vcc_lo, 0xffff\n s_ff1_i32_b32 s20, s20\n v_add_u32 %0, s20, %0"
                                   : "+v"(a) : "v"(b) : "vcc", "s20", "v40");)
                break;
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[pattern] = t1 - t0;
    sink[threadIdx.x] = a + b + c;
}

int main() {
    unsigned long long *d_out;
    unsigned *d_sink;
    hipMalloc(&d_out, 64 * 8);
    hipMalloc(&d_sink, 64 * 4);
    const char *names[] = {"v_add -> v_add", "s_add -> s_add", "v_readlane -> v_add (through an SGPR)", "v_cmp -> s_ff1 -> v_add",
                           "v_cmp -> s_and -> s_ff1 -> v_readlane(lane) -> v_add", "ds_read_b32 -> (address of the next)",
                           "v_readlane -> s_and -> v_mov -> ds_read_b32", "6 v_readlane -> 6 v_sub -> min3, min3, min", "dpp row_shl -> v_add",
                           "v_pk_sub clamp -> v_or3 -> v_cmp -> s_and -> s_ff1 -> v_add"};
    for (int rep = 0; rep < 2; rep++)
        for (int p = 0; p < 10; p++) {
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, p, d_out, d_sink);
            hipDeviceSynchronize();
            unsigned long long t;
            hipMemcpy(&t, d_out + p, 8, hipMemcpyDeviceToHost);
            if (rep == 1) printf("%-62s %7.1f cycles per step of the chain\n", names[p], (double)t / (256.0 * 16.0));
        }
    return 0;
}
