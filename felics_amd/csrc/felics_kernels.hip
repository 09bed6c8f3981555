// felics_kernels.hip -- CDNA4 (gfx950) kernels of the FELICS encode path.
//
// What the reference does per pixel in one serial loop (src/compression.rs:117-146)
// is split here into data-parallel stages over a whole batch of planes:
//
//   planes   RGB -> Y/Co/Cg planes                (compression.rs:346-356, color_transform.rs:11-17)
//   hist     classify every pixel against its two neighbours (misc.rs:6-24,
//            compression.rs:124-145), count out-of-range EVENTS per (tile, context)
//   offsets  scan the counts: every context's events form one CHAIN, stored contiguously
//   scatter  stable partition of the events by context, raster order kept
//   resolve  replay the Rice-parameter estimator along every chain
//            (parameter_selection.rs:49-85): the only sequential dependency
//   lengths  code length of every pixel -> bits per tile
//   bitscan  exclusive scan of tile bits -> bit offset of every tile in its stream
//   pack     build the codes (rice_coding.rs:26-38, phase_in_coding.rs:59-84,
//            compression.rs:29-45) and pack them MSB-first (bitstream-io BigEndian)
//
// Integer work only: no MFMA.  Wave = 64 lanes everywhere.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "felics_kernels.h"

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

// ------------------------------------------------------------------------------------------
// per-pixel classification shared by hist / scatter / lengths / pack
// ------------------------------------------------------------------------------------------

enum : uint32_t { CLS_IN = 0, CLS_BELOW = 1, CLS_ABOVE = 2 };

struct PixelClass {
    uint32_t cls;  // CLS_*
    uint32_t ctx;  // H - L
    uint32_t val;  // p-L (in range), L-p-1 (below), p-H-1 (above)
};

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
    int p = (int)pl[i], v1 = (int)pl[a], v2 = (int)pl[b];
    int H = max(v1, v2), L = min(v1, v2);
    PixelClass r;
    r.ctx = (uint32_t)(H - L);
    if (p < L) {
        r.cls = CLS_BELOW;
        r.val = (uint32_t)(L - p - 1);
    } else if (p > H) {
        r.cls = CLS_ABOVE;
        r.val = (uint32_t)(p - H - 1);
    } else {
        r.cls = CLS_IN;
        r.val = (uint32_t)(p - L);
    }
    return r;
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

// ------------------------------------------------------------------------------------------
// planes: interleaved RGB8 -> three int16 planes Y, Co, Cg (color_transform.rs:11-17).
// `/ 2` on int truncates toward zero exactly like Rust's.
// ------------------------------------------------------------------------------------------

__global__ void k_rgb8_to_planes(const uint8_t *__restrict__ rgb, int16_t *__restrict__ planes,
                                 uint32_t npix, uint32_t nimg) {
    uint64_t total = (uint64_t)npix * nimg;
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total;
         g += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t img = (uint32_t)(g / npix);
        uint32_t i = (uint32_t)(g - (uint64_t)img * npix);
        const uint8_t *s = rgb + g * 3;
        int r = s[0], gr = s[1], b = s[2];
        int co = r - b;
        int t = b + co / 2;
        int cg = gr - t;
        int yv = t + cg / 2;
        int16_t *o = planes + (uint64_t)img * 3 * npix;
        o[i] = (int16_t)yv;
        o[(uint64_t)npix + i] = (int16_t)co;
        o[2ull * npix + i] = (int16_t)cg;
    }
}

// ------------------------------------------------------------------------------------------
// hist: one wave per tile of SORT_TILE pixels; LDS histogram of event contexts.
// counts[(plane*ntiles + tile)*NCTX + ctx]
// ------------------------------------------------------------------------------------------

template <typename T>
__global__ __launch_bounds__(256) void k_hist(const T *__restrict__ planes, uint32_t *__restrict__ counts,
                                              uint32_t W, uint32_t npix, uint32_t ntiles) {
    __shared__ uint32_t hist[4][NCTX];
    const uint32_t wave = threadIdx.x >> 6, lane = lane_id();
    const uint32_t tile = blockIdx.x * 4 + wave;
    const uint32_t plane = blockIdx.y;
    for (uint32_t c = lane; c < NCTX; c += 64) hist[wave][c] = 0;
    __builtin_amdgcn_wave_barrier();
    if (tile < ntiles) {
        const T *pl = planes + (uint64_t)plane * npix;
        const uint32_t begin = tile * SORT_TILE;
        const uint32_t end = min(begin + SORT_TILE, npix);
        Coord xy;
        xy.set(begin + lane, W);
        for (uint32_t i = begin + lane; i < end; i += 64) {
            if (i >= 2) {
                PixelClass pc = classify(pl, i, xy.x, xy.y, W);
                if (pc.cls != CLS_IN) atomicAdd(&hist[wave][pc.ctx], 1u);
            }
            xy.advance(64, W);
        }
        __builtin_amdgcn_wave_barrier();
        uint32_t *dst = counts + ((uint64_t)plane * ntiles + tile) * NCTX;
        for (uint32_t c = lane; c < NCTX; c += 64) dst[c] = hist[wave][c];
    }
}

// ------------------------------------------------------------------------------------------
// offsets: per (plane, ctx) exclusive scan over tiles (in place), chain length out.
// One thread per (plane, ctx); lanes run over ctx so every step is a coalesced row access.
// ------------------------------------------------------------------------------------------

__global__ void k_tile_offsets(uint32_t *__restrict__ counts, uint32_t *__restrict__ chain_len,
                               uint32_t nplanes, uint32_t ntiles) {
    uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= nplanes * NCTX) return;
    uint32_t plane = g / NCTX, c = g - plane * NCTX;
    uint32_t *col = counts + (uint64_t)plane * ntiles * NCTX + c;
    uint32_t run = 0;
    for (uint32_t t = 0; t < ntiles; t++) {
        uint32_t v = col[(uint64_t)t * NCTX];
        col[(uint64_t)t * NCTX] = run;
        run += v;
    }
    chain_len[g] = run;
}

// Exclusive scan of chain_len over all (plane, ctx) -> chain_base; single block.
__global__ __launch_bounds__(1024) void k_chain_bases(const uint32_t *__restrict__ chain_len,
                                                      uint32_t *__restrict__ chain_base, uint32_t n,
                                                      uint32_t *__restrict__ total_events) {
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry;
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n; base += 1024) {
        uint32_t i = base + threadIdx.x;
        uint32_t v = i < n ? chain_len[i] : 0;
        uint32_t inc = wave_incl_scan(v);
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        uint32_t woff = 0;
        for (uint32_t w = 0; w < wave; w++) woff += wsum[w];
        uint32_t c = carry;
        if (i < n) chain_base[i] = c + woff + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = c + woff + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total_events = carry;
}

// ------------------------------------------------------------------------------------------
// scatter: stable partition of events by context.  One wave per tile walks its pixels in
// raster order, 64 at a time; lanes that hold the same context rank themselves with a ballot.
// sorted_e[slot] = value to Rice-code; slot_of[plane*npix + i] = slot (events only).
// ------------------------------------------------------------------------------------------

template <typename T, typename ET>
__global__ __launch_bounds__(256) void k_scatter(const T *__restrict__ planes,
                                                 const uint32_t *__restrict__ tile_off,
                                                 const uint32_t *__restrict__ chain_base,
                                                 ET *__restrict__ sorted_e, uint32_t *__restrict__ slot_of,
                                                 uint32_t W, uint32_t npix, uint32_t ntiles) {
    __shared__ uint32_t runs[4][NCTX];
    const uint32_t wave = threadIdx.x >> 6, lane = lane_id();
    const uint32_t tile = blockIdx.x * 4 + wave;
    const uint32_t plane = blockIdx.y;
    if (tile >= ntiles) return;
    uint32_t *run = runs[wave];
    {
        const uint32_t *off = tile_off + ((uint64_t)plane * ntiles + tile) * NCTX;
        const uint32_t *cb = chain_base + (uint64_t)plane * NCTX;
        for (uint32_t c = lane; c < NCTX; c += 64) run[c] = off[c] + cb[c];
    }
    __builtin_amdgcn_wave_barrier();
    const T *pl = planes + (uint64_t)plane * npix;
    uint32_t *slots = slot_of + (uint64_t)plane * npix;
    const uint32_t begin = tile * SORT_TILE;
    const uint32_t end = min(begin + SORT_TILE, npix);
    const uint64_t lt = lanemask_lt();
    Coord xy;
    xy.set(begin + lane, W);
    for (uint32_t row = begin; row < end; row += 64) {
        const uint32_t i = row + lane;
        bool ev = false;
        uint32_t c = 0, e = 0;
        if (i < end && i >= 2) {
            PixelClass pc = classify(pl, i, xy.x, xy.y, W);
            ev = pc.cls != CLS_IN;
            c = pc.ctx;
            e = pc.val;
        }
        xy.advance(64, W);
        uint64_t pending = __ballot(ev);
        uint32_t dest = 0;
        while (pending) {
            const uint32_t src = (uint32_t)__ffsll((long long)pending) - 1u;
            const uint32_t cc = readlane(c, src);
            const bool mine = ev && c == cc;
            const uint64_t m = __ballot(mine);
            const uint32_t basev = run[cc];
            if (mine) dest = basev + (uint32_t)__popcll(m & lt);
            __builtin_amdgcn_wave_barrier();
            if (lane == src) run[cc] = basev + (uint32_t)__popcll(m);
            __builtin_amdgcn_wave_barrier();
            pending &= ~m;
        }
        if (ev) {
            sorted_e[dest] = (ET)e;
            slots[i] = dest;
        }
    }
}

// ------------------------------------------------------------------------------------------
// resolve: replay KEstimator (parameter_selection.rs:49-85) along one chain per wave.
//
// State S[k] = accumulated Rice lengths for k = 0..5 (traits.rs:26).  For 64 consecutive events
// the wave prefix-sums the six length vectors; while no halving happens the state seen by lane t
// is S + P_excl(t).  `min(S + P_incl(t)) > 1024` (parameter_selection.rs:58-63) is monotone in t
// because lengths are positive, so the first lane f where it holds is the next halving:
// lanes <= f are final, S <- (S + P_incl(f)) >> 1, and later lanes continue from there.
// get_k ties go to the LARGEST k (`<=` at parameter_selection.rs:79).
// ------------------------------------------------------------------------------------------

template <typename ET>
__global__ __launch_bounds__(256) void k_resolve(const ET *__restrict__ sorted_e, uint8_t *__restrict__ k_sorted,
                                                 const uint32_t *__restrict__ chain_base,
                                                 const uint32_t *__restrict__ chain_len, uint32_t nchains) {
    const uint32_t chain = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (chain >= nchains) return;
    const uint32_t n = chain_len[chain];
    if (n == 0) return;
    const uint32_t lane = lane_id();
    const ET *ev = sorted_e + chain_base[chain];
    uint8_t *kout = k_sorted + chain_base[chain];

    uint32_t S0 = 0, S1 = 0, S2 = 0, S3 = 0, S4 = 0, S5 = 0;  // wave-uniform
    uint32_t e_next = lane < n ? (uint32_t)ev[lane] : 0;
    for (uint32_t g = 0; g < n; g += 64) {
        const bool valid = g + lane < n;
        const uint32_t e = e_next;
        if (g + 64 < n) e_next = (g + 64 + lane < n) ? (uint32_t)ev[g + 64 + lane] : 0;
        // Rice lengths (rice_coding.rs:56-58), two 16-bit sums per dword: 64 * 512 < 2^15.
        const uint32_t l0 = valid ? e + 1 : 0, l1 = valid ? (e >> 1) + 2 : 0, l2 = valid ? (e >> 2) + 3 : 0;
        const uint32_t l3 = valid ? (e >> 3) + 4 : 0, l4 = valid ? (e >> 4) + 5 : 0, l5 = valid ? (e >> 5) + 6 : 0;
        const uint32_t p01 = wave_incl_scan(l0 | (l1 << 16));
        const uint32_t p23 = wave_incl_scan(l2 | (l3 << 16));
        const uint32_t p45 = wave_incl_scan(l4 | (l5 << 16));
        const uint32_t P0 = p01 & 0xFFFF, P1 = p01 >> 16, P2 = p23 & 0xFFFF, P3 = p23 >> 16;
        const uint32_t P4 = p45 & 0xFFFF, P5 = p45 >> 16;
        uint32_t kk = 0;
        uint32_t lo = 0;
        while (true) {
            const uint32_t T0 = S0 + P0, T1 = S1 + P1, T2 = S2 + P2, T3 = S3 + P3, T4 = S4 + P4, T5 = S5 + P5;
            const uint32_t mn = min(min(min(T0, T1), min(T2, T3)), min(T4, T5));
            // state BEFORE this lane's event -> its k (get_k precedes update, compression.rs:127,139)
            const uint32_t X0 = T0 - l0, X1 = T1 - l1, X2 = T2 - l2, X3 = T3 - l3, X4 = T4 - l4, X5 = T5 - l5;
            uint32_t key = min(min(min((X0 << 3) | 7u, (X1 << 3) | 6u), min((X2 << 3) | 5u, (X3 << 3) | 4u)),
                               min((X4 << 3) | 3u, (X5 << 3) | 2u));
            const uint32_t cand = 7u - (key & 7u);
            const bool live = valid && lane >= lo;
            const uint64_t hm = __ballot(live && mn > 1024u);
            if (hm == 0) {
                if (live) kk = cand;
                S0 += readlane(P0, 63); S1 += readlane(P1, 63); S2 += readlane(P2, 63);
                S3 += readlane(P3, 63); S4 += readlane(P4, 63); S5 += readlane(P5, 63);
                break;
            }
            const uint32_t f = (uint32_t)__ffsll((long long)hm) - 1u;
            if (live && lane <= f) kk = cand;
            const uint32_t f0 = readlane(P0, f), f1 = readlane(P1, f), f2 = readlane(P2, f);
            const uint32_t f3 = readlane(P3, f), f4 = readlane(P4, f), f5 = readlane(P5, f);
            // S <- ((S + P(f)) >> 1) - P(f): later lanes add their own P(t) >= P(f) back (mod 2^32)
            S0 = ((S0 + f0) >> 1) - f0; S1 = ((S1 + f1) >> 1) - f1; S2 = ((S2 + f2) >> 1) - f2;
            S3 = ((S3 + f3) >> 1) - f3; S4 = ((S4 + f4) >> 1) - f4; S5 = ((S5 + f5) >> 1) - f5;
            lo = f + 1;
            if (lo >= 64) {
                S0 += readlane(P0, 63); S1 += readlane(P1, 63); S2 += readlane(P2, 63);
                S3 += readlane(P3, 63); S4 += readlane(P4, 63); S5 += readlane(P5, 63);
                break;
            }
        }
        if (valid) kout[g + lane] = (uint8_t)kk;
    }
}

// ------------------------------------------------------------------------------------------
// code construction shared by lengths / pack
// ------------------------------------------------------------------------------------------

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

// ------------------------------------------------------------------------------------------
// lengths: bits per PACK_TILE pixels.  Thread t owns PACK_PER_THREAD consecutive pixels.
// Plane 0 of an image also carries the 112 header bits; pixels 0 and 1 are raw 32-bit values
// (compression.rs:105-106).
// ------------------------------------------------------------------------------------------

template <typename T>
__device__ __forceinline__ uint32_t thread_bits(const T *__restrict__ pl, const uint32_t *__restrict__ slots,
                                                const uint8_t *__restrict__ k_sorted, uint32_t first,
                                                uint32_t end, uint32_t W) {
    uint32_t bits = 0;
    if (first >= end) return 0;
    Coord xy;
    xy.set(first, W);
    for (uint32_t i = first; i < end; i++) {
        if (i < 2) {
            bits += 32;
        } else {
            PixelClass pc = classify(pl, i, xy.x, xy.y, W);
            uint32_t k = 0;
            if (pc.cls != CLS_IN) k = k_sorted[slots[i]];
            bits += code_length(pc, k);
        }
        xy.advance(1, W);
    }
    return bits;
}

template <typename T>
__global__ __launch_bounds__(PACK_THREADS) void k_lengths(const T *__restrict__ planes,
                                                          const uint32_t *__restrict__ slot_of,
                                                          const uint8_t *__restrict__ k_sorted,
                                                          uint32_t *__restrict__ tile_bits, uint32_t W,
                                                          uint32_t npix, uint32_t ntiles, uint32_t planes_per_image) {
    __shared__ uint32_t wsum[PACK_THREADS / 64];
    const uint32_t tile = blockIdx.x, plane = blockIdx.y;
    const T *pl = planes + (uint64_t)plane * npix;
    const uint32_t *slots = slot_of + (uint64_t)plane * npix;
    const uint32_t first = tile * PACK_TILE + threadIdx.x * PACK_PER_THREAD;
    const uint32_t end = min(min(first + PACK_PER_THREAD, (tile + 1) * PACK_TILE), npix);
    uint32_t bits = thread_bits(pl, slots, k_sorted, first, end, W);
    if (npix == 1 && first == 0) bits += 32;  // 1x1: second raw value is a literal 0 (compression.rs:99-103)
    if (tile == 0 && threadIdx.x == 0 && (plane % planes_per_image) == 0) bits += 8 * 14;  // header
    uint32_t inc = wave_incl_scan(bits);
    if (lane_id() == 63) wsum[threadIdx.x >> 6] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (uint32_t w = 0; w < PACK_THREADS / 64; w++) tot += wsum[w];
        tile_bits[(uint64_t)plane * ntiles + tile] = tot;
    }
}

// ------------------------------------------------------------------------------------------
// bitscan: one block per image; exclusive scan of its planes' tile bits (planes are
// concatenated with no alignment, compression.rs:365-367).  Also the stream's byte size.
// ------------------------------------------------------------------------------------------

__global__ __launch_bounds__(1024) void k_bitscan(const uint32_t *__restrict__ tile_bits,
                                                  uint64_t *__restrict__ tile_bitoff,
                                                  uint64_t *__restrict__ image_bytes, uint32_t tiles_per_image) {
    __shared__ uint64_t wsum[16];
    __shared__ uint64_t carry;
    const uint32_t img = blockIdx.x;
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
    const uint32_t *src = tile_bits + (uint64_t)img * tiles_per_image;
    uint64_t *dst = tile_bitoff + (uint64_t)img * tiles_per_image;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < tiles_per_image; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < tiles_per_image ? src[i] : 0;
        // tile totals are < 2^32 but a wave of them may not be: scan low/high halves apart
        const uint32_t lo = wave_incl_scan(v & 0xFFFFu), hi = wave_incl_scan(v >> 16);
        const uint64_t inc = (uint64_t)lo + ((uint64_t)hi << 16);
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        uint64_t woff = 0;
        for (uint32_t w = 0; w < wave; w++) woff += wsum[w];
        const uint64_t c = carry;
        if (i < tiles_per_image) dst[i] = c + woff + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = c + woff + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) image_bytes[img] = (carry + 7) >> 3;  // byte_align (compression.rs:279)
}

// Stream placement: offsets[i] = sum of 16-byte-rounded sizes before i; one thread (n is small).
__global__ void k_place_streams(const uint64_t *__restrict__ image_bytes, uint64_t *__restrict__ image_off,
                                uint32_t nimg) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        uint64_t off = 0;
        for (uint32_t i = 0; i < nimg; i++) {
            image_off[i] = off;
            off += (image_bytes[i] + 15) & ~15ull;
        }
        image_off[nimg] = off;
    }
}

// Zero exactly the words the streams will occupy (pack ORs its tile-boundary words in).
__global__ void k_zero_streams(uint32_t *__restrict__ out, const uint64_t *__restrict__ image_off,
                               uint32_t nimg) {
    const uint64_t words = image_off[nimg] >> 2;
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < words;
         g += (uint64_t)gridDim.x * blockDim.x)
        out[g] = 0;
}

// ------------------------------------------------------------------------------------------
// pack: codes -> bits.  Bit b of a stream lives in word b >> 5 at position 31 - (b & 31);
// words are stored byte-swapped so the bytes come out MSB-first (bitstream-io BigEndian).
// Each thread strings the codes of its PACK_PER_THREAD pixels together in a 64-bit register and
// ORs finished 32-bit words into an LDS window; the window is then streamed out coalesced.
// Only a tile's first and last word can be shared with a neighbour tile: those are OR-ed
// atomically into the zeroed output, everything between is a plain store.
// ------------------------------------------------------------------------------------------

struct LaneBits {
    uint32_t *win;       // LDS window
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
            if (rel < PACK_WIN_WORDS) atomicOr(&win[rel], w);  // also false when cur_word < win_word0
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
        while (q >= 32) {
            put(0xFFFFFFFFu, 32);
            q -= 32;
        }
        if (q) put((1u << q) - 1u, q);
    }
    __device__ __forceinline__ void finish() {
        if (fill) emit((uint32_t)(acc >> 32));
    }
};

__device__ __forceinline__ void put_pixel(LaneBits &bw, const PixelClass &pc, uint32_t k) {
    if (pc.cls == CLS_IN) {  // `1` + phased-in (compression.rs:131-134)
        uint32_t b, nb;
        phase_in(pc.ctx + 1, pc.val, b, nb);
        bw.put((1u << nb) | b, nb + 1);
        return;
    }
    // `00` below / `01` above (compression.rs:35-42), unary quotient, 0, k-bit remainder
    const uint32_t flag = pc.cls == CLS_ABOVE ? 1u : 0u;
    const uint32_t q = pc.val >> k, rem = pc.val & ((1u << k) - 1u);
    if (q + k + 3 <= 32) {
        const uint32_t ones = q ? ((1u << q) - 1u) : 0u;  // q <= 29 here
        bw.put((flag << (q + 1 + k)) | (ones << (k + 1)) | rem, q + k + 3);
    } else {
        bw.put(flag, 2);
        bw.put_ones(q);
        bw.put(rem, k + 1);
    }
}

template <typename T>
__global__ __launch_bounds__(PACK_THREADS) void k_pack(const T *__restrict__ planes,
                                                       const uint32_t *__restrict__ slot_of,
                                                       const uint8_t *__restrict__ k_sorted,
                                                       const uint64_t *__restrict__ tile_bitoff,
                                                       const uint32_t *__restrict__ tile_bits,
                                                       const uint64_t *__restrict__ image_off,
                                                       uint8_t *__restrict__ out, uint32_t W, uint32_t H,
                                                       uint32_t npix, uint32_t ntiles, uint32_t planes_per_image,
                                                       uint32_t color, uint32_t depth) {
    __shared__ uint32_t win[PACK_WIN_WORDS];
    __shared__ uint32_t wsum[PACK_THREADS / 64];
    const uint32_t tile = blockIdx.x, plane = blockIdx.y;
    const uint32_t img = plane / planes_per_image;
    const bool first_plane = (plane % planes_per_image) == 0;
    const T *pl = planes + (uint64_t)plane * npix;
    const uint32_t *slots = slot_of + (uint64_t)plane * npix;
    const uint32_t first = tile * PACK_TILE + threadIdx.x * PACK_PER_THREAD;
    const uint32_t end = min(min(first + PACK_PER_THREAD, (tile + 1) * PACK_TILE), npix);
    const bool has_header = tile == 0 && threadIdx.x == 0 && first_plane;

    // this thread's bit offset inside the tile
    uint32_t bits = thread_bits(pl, slots, k_sorted, first, end, W);
    if (npix == 1 && first == 0) bits += 32;
    if (has_header) bits += 8 * 14;
    const uint32_t inc = wave_incl_scan(bits);
    if (lane_id() == 63) wsum[threadIdx.x >> 6] = inc;
    __syncthreads();
    uint32_t woff = 0;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) woff += wsum[w];
    const uint64_t tile_lo = tile_bitoff[(uint64_t)plane * ntiles + tile];  // bit offset in the image stream
    const uint64_t tile_hi = tile_lo + tile_bits[(uint64_t)plane * ntiles + tile];
    const uint64_t my_lo = tile_lo + woff + inc - bits;

    uint32_t *out_words = reinterpret_cast<uint32_t *>(out + image_off[img]);
    const uint64_t first_word = tile_lo >> 5, last_word = (tile_hi - 1) >> 5;

    for (uint64_t w0 = first_word; w0 <= last_word; w0 += PACK_WIN_WORDS) {
        for (uint32_t j = threadIdx.x; j < PACK_WIN_WORDS; j += PACK_THREADS) win[j] = 0;
        __syncthreads();
        // skip threads whose bits lie wholly outside this window
        if (bits != 0 && ((my_lo + bits - 1) >> 5) >= w0 && (my_lo >> 5) < w0 + PACK_WIN_WORDS) {
            LaneBits bw;
            bw.win = win;
            bw.win_word0 = w0;
            bw.begin(my_lo);
            if (has_header) {  // write_header, format.rs:51-61
                bw.put(0x464C4353u, 32);  // "FLCS"
                bw.put((color << 8) | depth, 16);
                bw.put(W, 32);
                bw.put(H, 32);
            }
            Coord xy;
            xy.set(first, W);
            for (uint32_t i = first; i < end; i++) {
                if (i < 2) {
                    bw.put((uint32_t)(int32_t)pl[i], 32);  // write_signed(32, p)
                    if (npix == 1) bw.put(0u, 32);
                } else {
                    PixelClass pc = classify(pl, i, xy.x, xy.y, W);
                    uint32_t k = 0;
                    if (pc.cls != CLS_IN) k = k_sorted[slots[i]];
                    put_pixel(bw, pc, k);
                }
                xy.advance(1, W);
            }
            bw.finish();
        }
        __syncthreads();
        for (uint32_t j = threadIdx.x; j < PACK_WIN_WORDS; j += PACK_THREADS) {
            const uint64_t aw = w0 + j;
            if (aw > last_word) break;
            const uint32_t v = __builtin_bswap32(win[j]);
            if (aw == first_word || aw == last_word) {
                if (v) atomicOr(&out_words[aw], v);
            } else {
                out_words[aw] = v;
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// launchers (host side of this translation unit)
// ------------------------------------------------------------------------------------------

static inline uint32_t cdiv(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

void launch_rgb8_to_planes(hipStream_t s, const uint8_t *rgb, int16_t *planes, uint32_t npix, uint32_t nimg) {
    uint64_t total = (uint64_t)npix * nimg;
    uint32_t blocks = (uint32_t)std::min<uint64_t>((total + 255) / 256, 256u * 32u);
    if (blocks == 0) return;
    hipLaunchKernelGGL(k_rgb8_to_planes, dim3(blocks), dim3(256), 0, s, rgb, planes, npix, nimg);
}

template <typename T>
void launch_hist(hipStream_t s, const T *planes, uint32_t *counts, const Geometry &g) {
    hipLaunchKernelGGL((k_hist<T>), dim3(cdiv(g.sort_tiles, 4), g.nplanes), dim3(256), 0, s, planes, counts, g.W,
                       g.npix, g.sort_tiles);
}
template void launch_hist<uint8_t>(hipStream_t, const uint8_t *, uint32_t *, const Geometry &);
template void launch_hist<int16_t>(hipStream_t, const int16_t *, uint32_t *, const Geometry &);

void launch_offsets(hipStream_t s, uint32_t *counts, uint32_t *chain_len, uint32_t *chain_base,
                    uint32_t *total_events, const Geometry &g) {
    const uint32_t nchains = g.nplanes * NCTX;
    hipLaunchKernelGGL(k_tile_offsets, dim3(cdiv(nchains, 256)), dim3(256), 0, s, counts, chain_len, g.nplanes,
                       g.sort_tiles);
    hipLaunchKernelGGL(k_chain_bases, dim3(1), dim3(1024), 0, s, chain_len, chain_base, nchains, total_events);
}

template <typename T, typename ET>
void launch_scatter(hipStream_t s, const T *planes, const uint32_t *tile_off, const uint32_t *chain_base,
                    ET *sorted_e, uint32_t *slot_of, const Geometry &g) {
    hipLaunchKernelGGL((k_scatter<T, ET>), dim3(cdiv(g.sort_tiles, 4), g.nplanes), dim3(256), 0, s, planes,
                       tile_off, chain_base, sorted_e, slot_of, g.W, g.npix, g.sort_tiles);
}
template void launch_scatter<uint8_t, uint8_t>(hipStream_t, const uint8_t *, const uint32_t *, const uint32_t *,
                                               uint8_t *, uint32_t *, const Geometry &);
template void launch_scatter<int16_t, uint16_t>(hipStream_t, const int16_t *, const uint32_t *, const uint32_t *,
                                                uint16_t *, uint32_t *, const Geometry &);

template <typename ET>
void launch_resolve(hipStream_t s, const ET *sorted_e, uint8_t *k_sorted, const uint32_t *chain_base,
                    const uint32_t *chain_len, const Geometry &g) {
    const uint32_t nchains = g.nplanes * NCTX;
    hipLaunchKernelGGL((k_resolve<ET>), dim3(cdiv(nchains, 4)), dim3(256), 0, s, sorted_e, k_sorted, chain_base,
                       chain_len, nchains);
}
template void launch_resolve<uint8_t>(hipStream_t, const uint8_t *, uint8_t *, const uint32_t *, const uint32_t *,
                                      const Geometry &);
template void launch_resolve<uint16_t>(hipStream_t, const uint16_t *, uint8_t *, const uint32_t *,
                                       const uint32_t *, const Geometry &);

template <typename T>
void launch_lengths(hipStream_t s, const T *planes, const uint32_t *slot_of, const uint8_t *k_sorted,
                    uint32_t *tile_bits, const Geometry &g) {
    hipLaunchKernelGGL((k_lengths<T>), dim3(g.pack_tiles, g.nplanes), dim3(PACK_THREADS), 0, s, planes, slot_of,
                       k_sorted, tile_bits, g.W, g.npix, g.pack_tiles, g.planes_per_image);
}
template void launch_lengths<uint8_t>(hipStream_t, const uint8_t *, const uint32_t *, const uint8_t *, uint32_t *,
                                      const Geometry &);
template void launch_lengths<int16_t>(hipStream_t, const int16_t *, const uint32_t *, const uint8_t *, uint32_t *,
                                      const Geometry &);

void launch_bitscan(hipStream_t s, const uint32_t *tile_bits, uint64_t *tile_bitoff, uint64_t *image_bytes,
                    uint64_t *image_off, const Geometry &g) {
    hipLaunchKernelGGL(k_bitscan, dim3(g.nimages), dim3(1024), 0, s, tile_bits, tile_bitoff, image_bytes,
                       g.pack_tiles * g.planes_per_image);
    hipLaunchKernelGGL(k_place_streams, dim3(1), dim3(64), 0, s, image_bytes, image_off, g.nimages);
}

void launch_zero_streams(hipStream_t s, uint32_t *out, const uint64_t *image_off, const Geometry &g) {
    hipLaunchKernelGGL(k_zero_streams, dim3(256 * 8), dim3(256), 0, s, out, image_off, g.nimages);
}

template <typename T>
void launch_pack(hipStream_t s, const T *planes, const uint32_t *slot_of, const uint8_t *k_sorted,
                 const uint64_t *tile_bitoff, const uint32_t *tile_bits, const uint64_t *image_off, uint8_t *out,
                 const Geometry &g) {
    hipLaunchKernelGGL((k_pack<T>), dim3(g.pack_tiles, g.nplanes), dim3(PACK_THREADS), 0, s, planes, slot_of,
                       k_sorted, tile_bitoff, tile_bits, image_off, out, g.W, g.H, g.npix, g.pack_tiles,
                       g.planes_per_image, g.color, g.depth);
}
template void launch_pack<uint8_t>(hipStream_t, const uint8_t *, const uint32_t *, const uint8_t *,
                                   const uint64_t *, const uint32_t *, const uint64_t *, uint8_t *,
                                   const Geometry &);
template void launch_pack<int16_t>(hipStream_t, const int16_t *, const uint32_t *, const uint8_t *,
                                   const uint64_t *, const uint32_t *, const uint64_t *, uint8_t *,
                                   const Geometry &);

}  // namespace felics
