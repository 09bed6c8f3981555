// felics_codes.h -- device code shared by the kernel translation units that build and pack codes
// (felics_kernels.hip):
// Rice lengths and block sums of the estimator, the phased-in / Rice code builders, the MSB-first
// bit writers and the look-back status words.
#pragma once

#include "felics_device.h"
#include "felics_kernels.h"

namespace felics {

// sum of the packed elements of w, each shifted right by K (K = 0..5)
template <int K>
__device__ __forceinline__ uint32_t packed_shift_sum(uint32_t w, uint8_t) {  // 4 x u8
    if (K == 0) return __builtin_amdgcn_sad_u8(w, 0u, 0u);
    return __builtin_amdgcn_sad_u8((w >> K) & (0x01010101u * (0xFFu >> K)), 0u, 0u);
}
template <int K>
__device__ __forceinline__ uint32_t packed_shift_sum(uint32_t w, uint16_t) {  // 2 x u16
    const uint32_t x = (w >> K) & (0x00010001u * (0xFFFFu >> K));
    return (x & 0xFFFFu) + (x >> 16);
}

template <typename ET>
__device__ __forceinline__ void add_block_sums(uint32_t w, uint32_t &b01, uint32_t &b23, uint32_t &b45) {
    b01 += packed_shift_sum<0>(w, ET()) | (packed_shift_sum<1>(w, ET()) << 16);
    b23 += packed_shift_sum<2>(w, ET()) | (packed_shift_sum<3>(w, ET()) << 16);
    b45 += packed_shift_sum<4>(w, ET()) | (packed_shift_sum<5>(w, ET()) << 16);
}

// Rice lengths of one event for k = 0..5 (rice_coding.rs:56-58), two 16-bit fields per dword
// (64 * 511 < 2^16, so a wave's prefix sums cannot carry between fields).
__device__ __forceinline__ void packed_lengths(uint32_t e, uint32_t &l01, uint32_t &l23, uint32_t &l45) {
    l01 = (e + 1u) | (((e >> 1) + 2u) << 16);
    l23 = ((e >> 2) + 3u) | (((e >> 3) + 4u) << 16);
    l45 = ((e >> 4) + 5u) | (((e >> 5) + 6u) << 16);
}


// The estimator's six counters as KEYS: key_k = S_k << 3 | (7 - k).  The smallest key names the smallest counter and,
// among equal counters, the largest k (`<=` at parameter_selection.rs:79); min(S) > 1024 <=> min key >= 1025 << 3.
struct EstKeys {
    uint32_t k0, k1, k2, k3, k4, k5, m;  // m = the smallest key (kept up to date by set / step)
    __device__ __forceinline__ void set(uint32_t s0, uint32_t s1, uint32_t s2, uint32_t s3, uint32_t s4, uint32_t s5) {
        k0 = (s0 << 3) | 7u; k1 = (s1 << 3) | 6u; k2 = (s2 << 3) | 5u;
        k3 = (s3 << 3) | 4u; k4 = (s4 << 3) | 3u; k5 = (s5 << 3) | 2u;
        m = min_key();
    }
    __device__ __forceinline__ uint32_t min_key() const { return min(min(min(k0, k1), k2), min(min(k3, k4), k5)); }
    // one event: returns 7 - k (k = get_k before the update), then update + halving
    __device__ __forceinline__ uint32_t step(uint32_t e) {
        const uint32_t r = m & 7u;
        const uint32_t e8 = e << 3;
        k0 += e8 + 8u;
        k1 += ((e8 >> 1) & ~7u) + 16u;
        k2 += ((e8 >> 2) & ~7u) + 24u;
        k3 += ((e8 >> 3) & ~7u) + 32u;
        k4 += ((e8 >> 4) & ~7u) + 40u;
        k5 += ((e8 >> 5) & ~7u) + 48u;
        m = min_key();
        if (m >= (1025u << 3)) {  // x /= 2 on every counter (parameter_selection.rs:62)
            k0 = ((k0 >> 1) & ~7u) | 7u; k1 = ((k1 >> 1) & ~7u) | 6u; k2 = ((k2 >> 1) & ~7u) | 5u;
            k3 = ((k3 >> 1) & ~7u) | 4u; k4 = ((k4 >> 1) & ~7u) | 3u; k5 = ((k5 >> 1) & ~7u) | 2u;
            m = min_key();
        }
        return r;
    }
};


// Phased-in code of v in [0, n) (phase_in_coding.rs:23-84): r = v + 2^m (mod n);
// r < right_p -> r in m bits, else r + right_p in m + 1 bits.  n - left_p = 2^m, so no division.
__device__ __forceinline__ void phase_in(uint32_t n, uint32_t v, uint32_t &bits, uint32_t &nbits) {
    const uint32_t m = 31u - (uint32_t)__clz((int)n);
    const uint32_t right_p = (2u << m) - n;
    uint32_t r = v + (1u << m);
    if (r >= n) r -= n;
    if (r < right_p) {
        bits = r;
        nbits = m;
    } else {
        bits = r + right_p;
        nbits = m + 1;
    }
}

// Bits one pixel emits (compression.rs:130-145): flag + phased-in, or flag + Rice(k).
__device__ __forceinline__ uint32_t code_length(const PixelClass &pc, uint32_t k) {
    if (pc.cls == CLS_IN) {
        uint32_t b, nb;
        phase_in(pc.ctx + 1, pc.val, b, nb);
        return 1 + nb;
    }
    return 2 + (pc.val >> k) + 1 + k;
}


__device__ __forceinline__ int sample_at(const uint32_t *w, uint32_t j, uint8_t) {
    return (int)((w[j >> 2] >> (8u * (j & 3u))) & 0xFFu);
}
__device__ __forceinline__ int sample_at(const uint32_t *w, uint32_t j, int16_t) {
    return (int)(int16_t)(w[j >> 1] >> (16u * (j & 1u)));
}
__device__ __forceinline__ int sample_at(const uint32_t *w, uint32_t j, uint16_t) {
    return (int)((w[j >> 1] >> (16u * (j & 1u))) & 0xFFFFu);
}
__device__ __forceinline__ int sample_at(const uint32_t *w, uint32_t j, int32_t) { return (int)w[j]; }


struct LaneBits {
    uint32_t *win;       // LDS window
    uint32_t win_words;  // its size
    uint64_t win_word0;  // absolute word index of win[0]
    uint64_t cur_word;   // absolute word being filled
    uint64_t acc;        // bits of cur_word in the top half, overflow below
    uint32_t fill;       // bits used in the top half (< 32 between calls)

    __device__ __forceinline__ void begin(uint64_t bitpos) {
        cur_word = bitpos >> 5;
        fill = (uint32_t)(bitpos & 31);
        acc = 0;
    }
    __device__ __forceinline__ void emit(uint32_t w) {
        if (w) {
            const uint64_t rel = cur_word - win_word0;
            if (rel < win_words) atomicOr(&win[rel], w);  // also false when cur_word < win_word0
        }
    }
    // append the low n bits of v (v < 2^n, 1 <= n <= 32), most significant first
    __device__ __forceinline__ void put(uint32_t v, uint32_t n) {
        acc |= (uint64_t)v << (64u - fill - n);
        fill += n;
        if (fill >= 32) {
            emit((uint32_t)(acc >> 32));
            acc <<= 32;
            fill -= 32;
            cur_word++;
        }
    }
    __device__ __forceinline__ void put_ones(uint32_t q) {  // write_unary0's run of ones
        if (q >= 32) {
            put(0xFFFFFFFFu, 32);
            q -= 32;
            // Every further whole word of the run is all ones and leaves acc / fill as they are: only the
            // words inside the window are touched (16-bit samples: a run can be 2^17 bits long).
            const uint64_t n = q >> 5;
            if (n) {
                const uint64_t lo = cur_word > win_word0 ? cur_word : win_word0;
                const uint64_t hi = cur_word + n < win_word0 + win_words ? cur_word + n : win_word0 + win_words;
                for (uint64_t w = lo; w < hi; w++) atomicOr(&win[w - win_word0], 0xFFFFFFFFu);
                cur_word += n;
                q &= 31u;
            }
        }
        if (q) put((1u << q) - 1u, q);
    }
    __device__ __forceinline__ void finish() {
        if (fill) emit((uint32_t)(acc >> 32));
    }
};

// One pixel's code.  Both kinds of code are built without branching -- `1` + phased-in
// (compression.rs:131-134), or `00` below / `01` above (compression.rs:35-42), unary quotient, 0, k-bit
// remainder -- and one of them is appended; only a Rice code longer than 32 bits takes a branch.
template <typename BW>
__device__ __forceinline__ void put_pixel(BW &bw, const PixelClass &pc, uint32_t k) {
    uint32_t b, nb;
    phase_in(pc.ctx + 1, pc.val, b, nb);
    const bool in_range = pc.cls == CLS_IN;
    const uint32_t flag = pc.cls == CLS_ABOVE ? 1u : 0u;
    const uint32_t q = pc.val >> k, rem = pc.val & ((1u << k) - 1u);
    const uint32_t n_rice = q + k + 3;
    const uint32_t n = in_range ? nb + 1 : n_rice;
    if (n <= 32) {
        const uint32_t sh = q & 31u;  // (q <= 29 whenever the Rice code is the one used)
        const uint32_t rice = (((flag << sh) | ((1u << sh) - 1u)) << (k + 1)) | rem;  // 0, flag, q ones, 0, rem
        bw.put(in_range ? (1u << nb) | b : rice, n);
    } else {
        bw.put(flag, 2);
        bw.put_ones(q);
        bw.put(rem, k + 1);
    }
}


__device__ __forceinline__ uint32_t *plane_words(const PlaneOut &po, uint32_t plane, uint64_t &limit_words) {
    const uint32_t img = plane / po.planes_per_image, c = plane - img * po.planes_per_image;
    if (c == 0) {
        limit_words = po.slot_stride >> 2;
        return reinterpret_cast<uint32_t *>(po.out + (uint64_t)img * po.slot_stride);
    }
    limit_words = po.plane_slot >> 2;
    return reinterpret_cast<uint32_t *>(po.scratch + ((uint64_t)img * (po.planes_per_image - 1) + c - 1) * po.plane_slot);
}


constexpr uint32_t ST_AGGREGATE = 1, ST_PREFIX = 2;
constexpr uint32_t ST_VALUE_BITS = 44;        // bits of a plane fit: < 2^32 pixels x < 2^10 bits
constexpr uint32_t ST_EPOCH_MASK = 0x3FFFFu;  // 18 bits of the lane's epoch (status is cleared when they wrap)
constexpr uint32_t LOOKBACK_SPIN_LIMIT = 1u << 19;  // polls of >= 1 us each: gives up after about a second

__device__ __forceinline__ uint64_t status_word(uint32_t epoch, uint32_t state, uint64_t value) {
    return ((uint64_t)(((epoch & ST_EPOCH_MASK) << 2) | state) << ST_VALUE_BITS) | value;
}


}  // namespace felics
