// In which order does one ds_add_rtn_u32 serve the lanes of a wave that name the SAME LDS address?
//
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/lds_atomic_order profiles/tools/micro/lds_atomic_order.hip && /tmp/lds_atomic_order
//
// Every wave draws 64 keys from a small alphabet (1 .. 512 different keys, some lanes switched off), adds 1 to the LDS counter of
// its key with one returning atomic, and compares what came back with the number of LOWER active lanes that hold the same key
// (counted with ballots).  A wave that always gets "rank in lane order" back can rank a batch of events with one LDS instruction;
// the product (k_scatter) still verifies the order it gets and does not depend on this program's answer.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                                   \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) {                                                                    \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));              \
            exit(1);                                                                               \
        }                                                                                          \
    } while (0)

constexpr uint32_t KEYS = 512;

__device__ __forceinline__ uint32_t lcg(uint32_t &s) {
    s = s * 1664525u + 1013904223u;
    return s >> 8;
}

__global__ __launch_bounds__(256) void k_order(uint32_t iters, uint32_t alphabet, uint32_t mask_mode,
                                               unsigned long long *__restrict__ result) {
    __shared__ uint32_t cnt[4][KEYS];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    uint32_t seed = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    unsigned long long wrong_asc = 0, wrong_desc = 0, wrong_half = 0, ops = 0;
    for (uint32_t it = 0; it < iters; it++) {
        for (uint32_t c = lane; c < KEYS; c += 64) cnt[wave][c] = 0;
        __builtin_amdgcn_wave_barrier();
        const uint32_t key = lcg(seed) % alphabet;
        bool on = true;
        if (mask_mode == 1) on = (lcg(seed) & 3u) != 0;       // a quarter of the lanes off, at random
        else if (mask_mode == 2) on = lane < (lcg(seed) & 63u) + 1u;  // per-lane random prefix: ragged
        uint32_t got = 0xFFFFFFFFu;
        if (on) got = atomicAdd(&cnt[wave][key], 1u);
        __builtin_amdgcn_wave_barrier();
        // reference: lanes below me / above me with my key, by ballots over the key's nine bits
        uint64_t m = __ballot(on);
        for (uint32_t b = 0; b < 9; b++) {
            const bool bit = (key >> b) & 1u;
            const uint64_t bb = __ballot(on && bit);
            m &= bit ? bb : ~bb;
        }
        const uint32_t below = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        const uint32_t group = (uint32_t)__popcll(m);
        // a third hypothesis: lanes 32..63 are served before lanes 0..31 or the like -- rank inside the half, halves swapped
        const uint64_t lo = m & 0xFFFFFFFFull, hi = m >> 32;
        const uint32_t half_rank = lane < 32 ? (uint32_t)__popcll(hi) + (uint32_t)__popcll(lo & ((1ull << lane) - 1ull))
                                             : (uint32_t)__popcll(hi & ((1ull << (lane - 32)) - 1ull));
        if (on) {
            ops++;
            wrong_asc += got != below;
            wrong_desc += got != group - 1u - below;
            wrong_half += got != half_rank;
        }
    }
    atomicAdd(&result[0], ops);
    atomicAdd(&result[1], wrong_asc);
    atomicAdd(&result[2], wrong_desc);
    atomicAdd(&result[3], wrong_half);
}

int main() {
    unsigned long long *d, h[4];
    CHECK(hipMalloc(&d, 32));
    const uint32_t alphabets[] = {1, 2, 3, 8, 10, 32, 64, 256, 512};
    for (uint32_t mode = 0; mode < 3; mode++) {
        for (uint32_t a : alphabets) {
            CHECK(hipMemset(d, 0, 32));
            k_order<<<256 * 8, 256>>>(2000, a, mode, d);  // eight workgroups per CU: every SIMD holds eight waves using the LDS
            CHECK(hipDeviceSynchronize());
            CHECK(hipMemcpy(h, d, 32, hipMemcpyDeviceToHost));
            printf("mask mode %u  alphabet %3u  %12llu atomics  not ascending-lane order: %llu   not descending: %llu   not halves-swapped: %llu\n",
                   mode, a, h[0], h[1], h[2], h[3]);
        }
    }
    return 0;
}
