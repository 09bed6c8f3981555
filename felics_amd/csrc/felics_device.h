// felics_device.h -- device-side helpers shared by the kernel translation units
// (felics_kernels.hip: 8-bit pipeline and the code builder; felics_wide.hip: 16-bit front end).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace felics {

// ------------------------------------------------------------------------------------------
// wave helpers
// ------------------------------------------------------------------------------------------

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

__device__ __forceinline__ uint64_t lanemask_lt() {
    return (1ull << lane_id()) - 1ull;
}

// Inclusive prefix sum over the 64 lanes of a wave, DPP only (no LDS):
// 4 row_shr steps inside each row of 16, then row_bcast:15 / row_bcast:31 across rows.
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);  // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);  // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);  // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);  // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false); // row_bcast:15 -> rows 1,3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false); // row_bcast:31 -> rows 2,3
    return v;
}

__device__ __forceinline__ uint32_t readlane(uint32_t v, uint32_t l) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l);
}

// number of set bits of m below this lane
__device__ __forceinline__ uint32_t mbcnt(uint64_t m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// ------------------------------------------------------------------------------------------
// per-pixel classification shared by hist / scatter / lengths / pack
// ------------------------------------------------------------------------------------------

enum : uint32_t { CLS_IN = 0, CLS_BELOW = 1, CLS_ABOVE = 2 };

struct PixelClass {
    uint32_t cls;  // CLS_*
    uint32_t ctx;  // H - L
    uint32_t val;  // p-L (in range), L-p-1 (below), p-H-1 (above)
};

// Pixel p against its two neighbours' values (compression.rs:124-145).
__device__ __forceinline__ PixelClass classify_values(int p, int v1, int v2) {
    const int H = max(v1, v2), L = min(v1, v2);
    const int ctx = H - L, d = p - L;  // in range: 0 <= d <= ctx
    PixelClass r;
    r.ctx = (uint32_t)ctx;
    r.cls = d < 0 ? CLS_BELOW : (d > ctx ? CLS_ABOVE : CLS_IN);
    // L - p - 1 = ~d ; p - H - 1 = d - ctx - 1
    r.val = (uint32_t)(d < 0 ? ~d : (d > ctx ? d - ctx - 1 : d));
    return r;
}

// The two already-coded neighbours of pixel i = y*W + x, i >= 2 (misc.rs:6-24).
template <typename T>
__device__ __forceinline__ PixelClass classify(const T *__restrict__ pl, uint32_t i, uint32_t x,
                                               uint32_t y, uint32_t W) {
    uint32_t a, b;
    if (x > 0 && y > 0) {
        a = i - 1;
        b = i - W;
    } else if (y == 0) {  // first row, x >= 2 because i >= 2
        a = i - 1;
        b = i - 2;
    } else if (y >= 2) {  // first column
        a = i - W;
        b = i - 2 * W;
    } else {  // pixel (0,1); W >= 2 because i >= 2
        a = i - W;
        b = i - W + 1;
    }
    return classify_values((int)pl[i], (int)pl[a], (int)pl[b]);
}

// Interior pixel (x > 0, y > 0): left and above, no case analysis.
template <typename T>
__device__ __forceinline__ PixelClass classify_interior(const T *__restrict__ pl, uint32_t i, uint32_t W) {
    return classify_values((int)pl[i], (int)pl[i - 1], (int)pl[i - W]);
}

// Four consecutive samples with one (possibly unaligned) 4- or 8-byte load.
__device__ __forceinline__ void load4(const uint8_t *__restrict__ p, int (&v)[4]) {
    uint32_t w;
    __builtin_memcpy(&w, p, 4);
    v[0] = (int)(w & 0xFFu); v[1] = (int)((w >> 8) & 0xFFu); v[2] = (int)((w >> 16) & 0xFFu); v[3] = (int)(w >> 24);
}
__device__ __forceinline__ void load4(const int16_t *__restrict__ p, int (&v)[4]) {
    uint32_t w[2];
    __builtin_memcpy(w, p, 8);
    v[0] = (int)(int16_t)w[0]; v[1] = (int)w[0] >> 16; v[2] = (int)(int16_t)w[1]; v[3] = (int)w[1] >> 16;
}

// 256 consecutive pixels of one image row below the first (y > 0), four per lane: left and above from two wide loads.
// A wave covers 256 consecutive pixels (lane l: first + 4l ..), so the sample left of a lane's first pixel
// is the last sample of the lane before it (one DPP wave shift); lane 0 fetches its own -- and when the span starts in
// the first column that "left" sample is the first-column rule's second neighbour instead (two rows up; above-right in
// row 1: misc.rs:14-23), which is all that rule changes, the two neighbours being interchangeable (classify_values).
// Split in two so that a caller can have the next trip's loads in flight while it works on this one.
__device__ __forceinline__ uint32_t span_left_index(uint32_t first, uint32_t x, uint32_t y, uint32_t W) {
    return x > 0 ? first - 1 : (y >= 2 ? first - 2 * W : first - W + 1);
}
// (the samples stay packed as loaded -- one dword per four u8 samples, two per four i16 -- so that a ring of prefetched
// trips costs three / five registers per entry; they are unpacked when the trip is classified)
template <typename T>
struct Interior4 {
    static constexpr uint32_t NW = sizeof(T);  // dwords per four samples
    uint32_t cur[NW], up[NW];
    int left_lane0;
};

__device__ __forceinline__ void unpack4(const uint32_t (&w)[1], int (&v)[4], uint8_t) {
    v[0] = (int)(w[0] & 0xFFu); v[1] = (int)((w[0] >> 8) & 0xFFu); v[2] = (int)((w[0] >> 16) & 0xFFu); v[3] = (int)(w[0] >> 24);
}
__device__ __forceinline__ void unpack4(const uint32_t (&w)[2], int (&v)[4], int16_t) {
    v[0] = (int)(int16_t)w[0]; v[1] = (int)w[0] >> 16; v[2] = (int)(int16_t)w[1]; v[3] = (int)w[1] >> 16;
}

template <typename T>
__device__ __forceinline__ void load_interior4(const T *__restrict__ pl, uint32_t first, uint32_t W, uint32_t left_index,
                                               Interior4<T> &v) {
    const uint32_t i = first + 4 * lane_id();
    __builtin_memcpy(v.cur, pl + i, 4 * sizeof(T));      // (possibly unaligned) 4- or 8-byte loads
    __builtin_memcpy(v.up, pl + i - W, 4 * sizeof(T));
    v.left_lane0 = (int)pl[left_index];  // span_left_index(); same address in every lane: one scalar-like access
}

template <typename T>
__device__ __forceinline__ void classify_loaded4(const Interior4<T> &v, PixelClass (&pc)[4]) {
    int cur[4], up[4];
    unpack4(v.cur, cur, T());
    unpack4(v.up, up, T());
    int left0 = __builtin_amdgcn_update_dpp(0, cur[3], 0x138, 0xF, 0xF, false);  // wave_shr:1
    if (lane_id() == 0) left0 = v.left_lane0;
    pc[0] = classify_values(cur[0], left0, up[0]);
#pragma unroll
    for (int j = 1; j < 4; j++) pc[j] = classify_values(cur[j], cur[j - 1], up[j]);
}

template <typename T>
__device__ __forceinline__ void classify_interior4(const T *__restrict__ pl, uint32_t first, uint32_t W, uint32_t left_index,
                                                   PixelClass (&pc)[4]) {
    const uint32_t i = first + 4 * lane_id();
    int cur[4], up[4];
    load4(pl + i, cur);
    load4(pl + i - W, up);
    int left0 = __builtin_amdgcn_update_dpp(0, cur[3], 0x138, 0xF, 0xF, false);  // wave_shr:1
    if (lane_id() == 0) left0 = (int)pl[left_index];
    pc[0] = classify_values(cur[0], left0, up[0]);
#pragma unroll
    for (int j = 1; j < 4; j++) pc[j] = classify_values(cur[j], cur[j - 1], up[j]);
}

// (x, y) of linear index i; advance() moves forward by `step` pixels without dividing again.
struct Coord {
    uint32_t x, y;
    __device__ __forceinline__ void set(uint32_t i, uint32_t W) {
        y = i / W;
        x = i - y * W;
    }
    __device__ __forceinline__ void advance(uint32_t step, uint32_t W) {
        x += step;
        while (x >= W) {
            x -= W;
            y++;
        }
    }
};

}  // namespace felics
