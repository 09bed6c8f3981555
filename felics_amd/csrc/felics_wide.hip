// felics_wide.hip -- front end of the encode path for 16-bit samples (gfx950).
//
// With u16 samples a context Delta = H - L goes up to 131 070 and the estimator keeps 15 counters
// per context (traits.rs:35-43), so the per-tile context histograms of the 8-bit pipeline do not
// apply.  Here the out-of-range EVENTS of the whole batch are ordered by (plane, context) with one
// stable radix sort (rocPRIM), which turns every context's events into one contiguous CHAIN in
// raster order; one wave then replays the estimator (parameter_selection.rs:49-85) along each
// chain, 64 events at a time.  The result is the same k_map (k of every out-of-range pixel, raster
// order) the 8-bit pipeline produces, and lengths / bitscan / pack of felics_kernels.hip take over.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/types/double_buffer.hpp>

#include "felics_device.h"
#include "felics_kernels.h"

namespace felics {

// interleaved RGB16 -> three i32 planes Y, Co, Cg (color_transform.rs:11-17; compression.rs:346-356
// widens to i32 first, so Co / Cg of 16-bit samples need 18 bits).
__global__ void k_rgb16_to_planes(const uint16_t *__restrict__ rgb, int32_t *__restrict__ planes, uint32_t npix,
                                  uint32_t nimg) {
    const uint64_t total = (uint64_t)npix * nimg;
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total;
         g += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t img = (uint32_t)(g / npix);
        const uint32_t i = (uint32_t)(g - (uint64_t)img * npix);
        const uint16_t *s = rgb + g * 3;
        const int r = s[0], gr = s[1], b = s[2];
        const int co = r - b;
        const int t = b + co / 2;
        const int cg = gr - t;
        const int yv = t + cg / 2;
        int32_t *o = planes + (uint64_t)img * 3 * npix;
        o[i] = yv;
        o[(uint64_t)npix + i] = co;
        o[2ull * npix + i] = cg;
    }
}

// One sort record per sample.  Samples that are not events (in range, or one of the two raw
// pixels) get the largest context of their plane, so they end up behind the plane's chains.
template <typename T>
__global__ __launch_bounds__(256) void k_wide_keys(const T *__restrict__ planes, uint32_t *__restrict__ keys,
                                                   uint32_t *__restrict__ vals, uint32_t *__restrict__ e_of, uint32_t W,
                                                   uint32_t npix) {
    const uint32_t plane = blockIdx.y;
    const T *pl = planes + (uint64_t)plane * npix;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += gridDim.x * blockDim.x) {
        uint32_t key = (plane << WIDE_CTX_BITS) | WIDE_NO_EVENT, e = 0;
        if (i >= 2) {
            const uint32_t y = i / W, x = i - y * W;
            const PixelClass pc = classify(pl, i, x, y, W);
            if (pc.cls != CLS_IN) {
                key = (plane << WIDE_CTX_BITS) | pc.ctx;
                e = pc.val;
            }
        }
        const uint32_t g = plane * npix + i;  // < 2^32: the host bounds the batch
        keys[g] = key;
        vals[g] = g;
        e_of[g] = e;
    }
}

// Position j starts a chain if it is an event and the key before it differs.
__global__ __launch_bounds__(256) void k_wide_heads(const uint32_t *__restrict__ keys, uint32_t n,
                                                    uint32_t *__restrict__ heads, uint32_t *__restrict__ nheads) {
    const uint32_t nwaves_total = (n + 63u) / 64u;
    for (uint32_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; w < nwaves_total; w += (gridDim.x * blockDim.x) >> 6) {
        const uint32_t j = w * 64u + lane_id();
        bool head = false;
        if (j < n) {
            const uint32_t k = keys[j];
            head = (k & WIDE_NO_EVENT) != WIDE_NO_EVENT && (j == 0 || keys[j - 1] != k);
        }
        const uint64_t m = __ballot(head);
        if (m == 0) continue;
        uint32_t base = 0;
        if (lane_id() == 0) base = atomicAdd(nheads, (uint32_t)__popcll(m));
        base = readlane(base, 0);
        if (head) heads[base + mbcnt(m)] = j;
    }
}

constexpr int WIDE_NK = 15;             // K_VALUES = 0..=14 (traits.rs:36)
constexpr uint32_t WIDE_HALVE = 1024;   // COUNT_SCALING (traits.rs:40)

// One wave per chain (persistent grid).  Lane l of a step holds event j + l of the chain.  For every
// Rice parameter the lanes' code lengths are prefix-summed, which gives each lane the counters as
// they were before its event (-> its k: smallest counter, ties to the largest k, parameter_selection.rs
// :71-85) and after it.  The counters are halved after the first event that lifts their minimum above
// 1024 (:58-68); that minimum never decreases along the block, so the event is found with one ballot,
// the counters are halved there and the rest of the block is scanned again from the halved state.
__global__ __launch_bounds__(256) void k_wide_chains(const uint32_t *__restrict__ keys, const uint32_t *__restrict__ vals,
                                                     const uint32_t *__restrict__ e_of, uint32_t n,
                                                     const uint32_t *__restrict__ heads,
                                                     const uint32_t *__restrict__ nheads, uint8_t *__restrict__ k_map) {
    const uint32_t lane = lane_id();
    const uint32_t nchains = *nheads;
    const uint32_t wave0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t c = wave0; c < nchains; c += nwaves) {
        uint32_t j = (uint32_t)__builtin_amdgcn_readfirstlane((int)heads[c]);  // wave-uniform: chain position and counters stay scalar
        const uint32_t key = keys[j];
        uint32_t S[WIDE_NK];
#pragma unroll
        for (int k = 0; k < WIDE_NK; k++) S[k] = 0;
        for (;;) {
            const uint32_t idx = j + lane;
            const bool valid = idx < n && keys[idx] == key;
            const uint64_t vm = __ballot(valid);  // a prefix of the lanes
            const uint32_t nvalid = (uint32_t)__popcll(vm);
            if (nvalid == 0) break;
            const uint32_t pix = valid ? vals[idx] : 0u;
            const uint32_t e = valid ? e_of[pix] : 0u;
            uint32_t base = 0;  // first event of the block not yet resolved
            while (base < nvalid) {
                const bool live = valid && lane >= base;
                uint32_t best_k = 0, best = 0xFFFFFFFFu, after_min = 0xFFFFFFFFu;
                uint32_t P[WIDE_NK];
#pragma unroll
                for (int k = 0; k < WIDE_NK; k++) {
                    const uint32_t len = live ? (e >> k) + 1u + (uint32_t)k : 0u;  // rice_coding.rs:40-46
                    const uint32_t inc = wave_incl_scan(len);
                    P[k] = inc;
                    const uint32_t before = S[k] + inc - len;
                    if (before <= best) {
                        best = before;
                        best_k = (uint32_t)k;
                    }
                    after_min = min(after_min, S[k] + inc);
                }
                const uint64_t hm = __ballot(live && after_min > WIDE_HALVE);
                const uint32_t f = hm ? (uint32_t)__builtin_ctzll(hm) : nvalid - 1u;  // last event served by this scan
                if (live && lane <= f) k_map[pix] = (uint8_t)best_k;
#pragma unroll
                for (int k = 0; k < WIDE_NK; k++) {
                    const uint32_t s = S[k] + readlane(P[k], f);
                    S[k] = hm ? s >> 1 : s;
                }
                base = f + 1u;
            }
            if (nvalid < 64u) break;
            j += 64u;
        }
    }
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------

static inline uint32_t cdiv(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

void launch_rgb16_to_planes(hipStream_t s, const uint16_t *rgb, int32_t *planes, uint32_t npix, uint32_t nimg) {
    const uint64_t total = (uint64_t)npix * nimg;
    const uint32_t blocks = (uint32_t)std::min<uint64_t>(cdiv(total, 256), 256u * 32u);
    FELICS_LAUNCH(k_rgb16_to_planes, dim3(blocks), dim3(256), s, rgb, planes, npix, nimg);
}

template <typename T>
void launch_wide_keys(hipStream_t s, const T *planes, uint32_t *keys, uint32_t *vals, uint32_t *e_of, const Geometry &g) {
    const uint32_t bx = std::min<uint32_t>(cdiv(g.npix, 256), 4096u);
    FELICS_LAUNCH((k_wide_keys<T>), dim3(bx, g.nplanes), dim3(256), s, planes, keys, vals, e_of, g.W, g.npix);
}
template void launch_wide_keys<uint16_t>(hipStream_t, const uint16_t *, uint32_t *, uint32_t *, uint32_t *,
                                         const Geometry &);
template void launch_wide_keys<int32_t>(hipStream_t, const int32_t *, uint32_t *, uint32_t *, uint32_t *, const Geometry &);

size_t wide_sort_temp_bytes(size_t n, uint32_t key_bits) {
    size_t bytes = 0;
    rocprim::double_buffer<uint32_t> k(nullptr, nullptr), v(nullptr, nullptr);
    if (rocprim::radix_sort_pairs(nullptr, bytes, k, v, n, 0u, key_bits) != hipSuccess) return 0;
    return bytes;
}

hipError_t wide_sort(hipStream_t s, void *temp, size_t temp_bytes, uint32_t *keys_a, uint32_t *keys_b, uint32_t *vals_a,
                     uint32_t *vals_b, size_t n, uint32_t key_bits, uint32_t **sorted_keys, uint32_t **sorted_vals) {
    rocprim::double_buffer<uint32_t> k(keys_a, keys_b), v(vals_a, vals_b);
    const hipError_t e = rocprim::radix_sort_pairs(temp, temp_bytes, k, v, n, 0u, key_bits, s);
    *sorted_keys = k.current();
    *sorted_vals = v.current();
    return e;
}

void launch_wide_heads(hipStream_t s, const uint32_t *keys, uint32_t n, uint32_t *heads, uint32_t *nheads) {
    const uint32_t blocks = std::min<uint32_t>(cdiv(n, 256), 256u * 16u);
    FELICS_LAUNCH(k_wide_heads, dim3(blocks), dim3(256), s, keys, n, heads, nheads);
}

void launch_wide_chains(hipStream_t s, const uint32_t *keys, const uint32_t *vals, const uint32_t *e_of, uint32_t n,
                        const uint32_t *heads, const uint32_t *nheads, uint8_t *k_map) {
    // persistent: 8 workgroups of 4 waves per CU share the chains
    FELICS_LAUNCH(k_wide_chains, dim3(256u * 8u), dim3(256), s, keys, vals, e_of, n, heads, nheads, k_map);
}

}  // namespace felics
