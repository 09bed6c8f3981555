// felics_wide.hip -- front end of the encode path for 16-bit samples (gfx950).
//
// With u16 samples a context Delta = H - L goes up to 131 070 and the estimator keeps 15 counters
// per context (traits.rs:35-43), so the per-tile context histograms of the 8-bit pipeline do not
// apply.  Here the out-of-range EVENTS of the whole batch are compacted into 64-bit records and ordered by
// (plane, context) with a stable two-pass radix sort of our own, which turns every context's events into
// one contiguous CHAIN in raster order; one wave then replays the estimator (parameter_selection.rs:49-85) along each
// chain, 64 events at a time.  The result is the same k_map (k of every out-of-range pixel, raster
// order) the 8-bit pipeline produces, and lengths / bitscan / pack of felics_kernels.hip take over.
#include <hip/hip_runtime.h>
#include <stdint.h>

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

// ------------------------------------------------------------------------------------------
// Events, compacted and sorted.  One 64-bit record per out-of-range sample:
//     bits 0..17   context Delta = H - L            (<= 131 070)
//     bits 18..34  the value that gets Rice-coded   (<= 131 069)
//     bits 35..63  the sample's index in its plane  (< 2^29: the host bounds the image size)
// The records are written plane by plane in raster order (count -> scan -> emit), then sorted by
// context inside every plane with a stable LSD radix sort of two 9-bit digits (histogram per tile of
// WT records -> scan per plane -> scatter), so that every (plane, context) chain is contiguous and in
// raster order.  Written from scratch for this path: no library sort.
// ------------------------------------------------------------------------------------------

constexpr uint32_t WT = 4096;       // samples per tile (count / emit), records per tile (sort passes)
constexpr uint32_t WTHREADS = 256;  // threads per tile: 16 samples / records each
constexpr uint32_t WDIG = 512;      // 9-bit digits
constexpr uint32_t REC_CTX_BITS = 18, REC_E_BITS = 17;

__device__ __forceinline__ uint64_t make_rec(uint32_t ctx, uint32_t e, uint32_t pix) {
    return (uint64_t)ctx | ((uint64_t)e << REC_CTX_BITS) | ((uint64_t)pix << (REC_CTX_BITS + REC_E_BITS));
}
__device__ __forceinline__ uint32_t rec_ctx(uint64_t r) { return (uint32_t)r & ((1u << REC_CTX_BITS) - 1u); }
__device__ __forceinline__ uint32_t rec_e(uint64_t r) { return (uint32_t)(r >> REC_CTX_BITS) & ((1u << REC_E_BITS) - 1u); }
__device__ __forceinline__ uint32_t rec_pix(uint64_t r) { return (uint32_t)(r >> (REC_CTX_BITS + REC_E_BITS)); }

// exclusive prefix of v over the WTHREADS threads of a workgroup (wsum: 4 words of LDS); *total = sum
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *wsum, uint32_t *total) {
    const uint32_t inc = wave_incl_scan(v);
    const uint32_t wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane_id() == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t off = 0, tot = 0;
    for (uint32_t w = 0; w < WTHREADS / 64; w++) {
        if (w < wave) off += wsum[w];
        tot += wsum[w];
    }
    *total = tot;
    return off + inc - v;
}

// thread t of tile (blockIdx.x, plane blockIdx.y) owns the 16 consecutive samples first .. first + 15: f(j, i, pc) for
// every event among them.  Samples strictly inside the image (left and above neighbours, misc.rs:6-24) come from two
// vector loads; a group that touches the first row, the first column or the end of the plane takes the general rule.
template <typename T, typename F>
__device__ __forceinline__ void for_my_events(const T *__restrict__ pl, uint32_t W, uint32_t npix, F &&f) {
    const uint32_t first = blockIdx.x * WT + threadIdx.x * 16u;
    if (first >= npix) return;
    Coord xy;
    xy.set(first, W);
    if (xy.y > 0 && xy.x > 0 && xy.x + 16u <= W && first + 16u <= npix) {
        T cur[16], up[16];
        __builtin_memcpy(cur, pl + first, sizeof cur);
        __builtin_memcpy(up, pl + first - W, sizeof up);
        int left = (int)pl[first - 1];
#pragma unroll
        for (uint32_t j = 0; j < 16u; j++) {
            const PixelClass pc = classify_values((int)cur[j], left, (int)up[j]);
            if (pc.cls != CLS_IN) f(j, first + j, pc);
            left = (int)cur[j];
        }
        return;
    }
    for (uint32_t j = 0; j < 16u && first + j < npix; j++) {
        const uint32_t i = first + j;
        if (i >= 2) {
            const PixelClass pc = classify(pl, i, xy.x, xy.y, W);
            if (pc.cls != CLS_IN) f(j, i, pc);
        }
        xy.advance(1, W);
    }
}

template <typename T>
__global__ __launch_bounds__(WTHREADS) void k_wide_count(const T *__restrict__ planes, uint32_t W, uint32_t npix,
                                                         uint32_t *__restrict__ tile_cnt) {
    __shared__ uint32_t wsum[WTHREADS / 64];
    const T *pl = planes + (uint64_t)blockIdx.y * npix;
    uint32_t n = 0;
    for_my_events(pl, W, npix, [&](uint32_t, uint32_t, const PixelClass &) { n++; });
    uint32_t total;
    (void)block_excl_scan(n, wsum, &total);
    if (threadIdx.x == 0) tile_cnt[blockIdx.y * gridDim.x + blockIdx.x] = total;
}

// One workgroup: exclusive scan of the tile counts (plane-major) in place; per plane its first record, its
// number of sort tiles and the first of them.  meta (u32): [0] records in all, [1] sort tiles in all, then
// plane_ev0[P + 1] at WMETA_EV0, stile_first[P + 1] behind it.
constexpr uint32_t WMETA_EV0 = 4;
__global__ __launch_bounds__(1024) void k_wide_plan(uint32_t *__restrict__ tile_cnt, uint32_t ntiles, uint32_t nplanes,
                                                    uint32_t *__restrict__ meta) {
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry;
    uint32_t *plane_ev0 = meta + WMETA_EV0, *stile_first = plane_ev0 + nplanes + 1;
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const uint32_t n = ntiles * nplanes;
    for (uint32_t base = 0; base < n; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < n ? tile_cnt[i] : 0;
        const uint32_t inc = wave_incl_scan(v);
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        uint32_t woff = 0;
        for (uint32_t w = 0; w < wave; w++) woff += wsum[w];
        const uint32_t c = carry;
        if (i < n) {
            tile_cnt[i] = c + woff + inc - v;
            if (i % ntiles == 0) plane_ev0[i / ntiles] = c + woff + inc - v;
        }
        __syncthreads();
        if (threadIdx.x == 1023) carry = c + woff + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        plane_ev0[nplanes] = carry;
        meta[0] = carry;
        uint32_t st = 0;
        for (uint32_t p = 0; p < nplanes; p++) {  // (a few hundred planes at most)
            stile_first[p] = st;
            st += (plane_ev0[p + 1] - plane_ev0[p] + WT - 1) / WT;
        }
        stile_first[nplanes] = st;
        meta[1] = st;
    }
}

template <typename T>
__global__ __launch_bounds__(WTHREADS) void k_wide_emit(const T *__restrict__ planes, uint32_t W, uint32_t npix,
                                                        const uint32_t *__restrict__ tile_base, uint64_t *__restrict__ recs) {
    __shared__ uint32_t wsum[WTHREADS / 64];
    const T *pl = planes + (uint64_t)blockIdx.y * npix;
    // classify once: the thread's records wait in registers for their place (raster order: thread by thread)
    uint64_t mine[16];
    uint32_t n = 0;
#pragma unroll
    for (uint32_t j = 0; j < 16u; j++) mine[j] = ~0ull;
    for_my_events(pl, W, npix, [&](uint32_t j, uint32_t i, const PixelClass &pc) {
        // (j is a compile-time constant on the fast path; the general path is rare enough for a select chain)
#pragma unroll
        for (uint32_t q = 0; q < 16u; q++)
            if (q == j) mine[q] = make_rec(pc.ctx, pc.val, i);
        n++;
    });
    // the tile's records are lined up in LDS and leave from there, neighbouring lanes writing neighbouring records (straight
    // from the registers a store touched ~40 lines: a thread's events are ~10 records away from its neighbour's)
    __shared__ uint64_t lined[WT];
    uint32_t total;
    uint32_t at = block_excl_scan(n, wsum, &total);
#pragma unroll
    for (uint32_t j = 0; j < 16u; j++)
        if (mine[j] != ~0ull) lined[at++] = mine[j];
    __syncthreads();
    uint64_t *out = recs + tile_base[blockIdx.y * gridDim.x + blockIdx.x];
    for (uint32_t i = threadIdx.x; i < total; i += WTHREADS) out[i] = lined[i];
}

// Which plane a sort tile belongs to, and its record range.
struct SortTile {
    uint32_t plane, t, ntile, begin, end;
};
__device__ __forceinline__ bool sort_tile(const uint32_t *__restrict__ meta, uint32_t nplanes, uint32_t tile, SortTile &st) {
    const uint32_t *plane_ev0 = meta + WMETA_EV0, *stile_first = plane_ev0 + nplanes + 1;
    if (tile >= meta[1]) return false;
    uint32_t lo = 0, hi = nplanes;  // last plane whose first tile is <= tile (planes without events share their successor's)
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (stile_first[mid] <= tile) lo = mid; else hi = mid;
    }
    st.plane = lo;
    st.t = tile - stile_first[lo];
    st.ntile = stile_first[lo + 1] - stile_first[lo];
    st.begin = plane_ev0[lo] + st.t * WT;
    st.end = min(st.begin + WT, plane_ev0[lo + 1]);
    return true;
}

// One count per record into an LDS histogram.  The high digit of the contexts is the same for almost every record of a tile
// (64 lanes adding to ONE LDS word take 64 turns): the lanes that share the first lane's digit are counted with one add.
__device__ __forceinline__ void count_digit(uint32_t *h, uint32_t d, bool valid) {
    const uint64_t vm = __ballot(valid);
    if (vm == 0) return;
    const uint32_t d0 = readlane(d, (uint32_t)__builtin_ctzll(vm));
    const uint64_t same = __ballot(valid && d == d0);
    if (__popcll(same) >= 8) {
        if (lane_id() == (uint32_t)__builtin_ctzll(same)) atomicAdd(&h[d0], (uint32_t)__popcll(same));
        valid = valid && d != d0;
    }
    if (valid) atomicAdd(&h[d], 1u);
}

// histogram of one digit per sort tile: hist[(WDIG * stile_first[plane]) + digit * ntile + t].  A workgroup counts
// HIST_TILES consecutive tiles: their counts of a digit are neighbours in the matrix and leave with one store, and the
// digit's total of the plane gets one atomic for all of them (one tile per workgroup meant 512 scattered dword stores
// and up to 512 atomics per 4096 records: the kernel waited for its address unit, not for the records).
constexpr uint32_t HIST_TILES = 4;
__global__ __launch_bounds__(WTHREADS) void k_wsort_hist(const uint64_t *__restrict__ recs, const uint32_t *__restrict__ meta,
                                                         uint32_t nplanes, uint32_t shift, uint32_t *__restrict__ hist,
                                                         uint32_t *__restrict__ dig_tot) {
    __shared__ uint32_t h[HIST_TILES][WDIG];
    SortTile st[HIST_TILES];
    bool have[HIST_TILES];
    for (uint32_t d = threadIdx.x; d < HIST_TILES * WDIG; d += WTHREADS) (&h[0][0])[d] = 0;
    __syncthreads();
#pragma unroll
    for (uint32_t i = 0; i < HIST_TILES; i++) {
        have[i] = sort_tile(meta, nplanes, blockIdx.x * HIST_TILES + i, st[i]);
        if (!have[i]) continue;
        uint64_t r[WT / WTHREADS];
#pragma unroll
        for (uint32_t k = 0; k < WT / WTHREADS; k++) {
            const uint32_t j = st[i].begin + k * WTHREADS + threadIdx.x;
            r[k] = j < st[i].end ? recs[j] : ~0ull;
        }
#pragma unroll
        for (uint32_t k = 0; k < WT / WTHREADS; k++)
            count_digit(h[i], (rec_ctx(r[k]) >> shift) & (WDIG - 1u), st[i].begin + k * WTHREADS + threadIdx.x < st[i].end);
    }
    __syncthreads();
    if (!have[0]) return;
    const uint32_t *stile_first = meta + WMETA_EV0 + nplanes + 1;
    const bool one_plane = have[HIST_TILES - 1] && st[HIST_TILES - 1].plane == st[0].plane;
    for (uint32_t d = threadIdx.x; d < WDIG; d += WTHREADS) {
        if (one_plane) {  // (the common case: four neighbouring counts, one total)
            uint32_t *dst = hist + (uint64_t)WDIG * stile_first[st[0].plane] + (uint64_t)d * st[0].ntile + st[0].t;
            uint32_t v[HIST_TILES], sum = 0;
#pragma unroll
            for (uint32_t i = 0; i < HIST_TILES; i++) v[i] = h[i][d], sum += v[i];
            __builtin_memcpy(dst, v, sizeof v);
            if (sum) atomicAdd(&dig_tot[st[0].plane * WDIG + d], sum);  // per (plane, digit): lets the scan start anywhere
            continue;
        }
#pragma unroll
        for (uint32_t i = 0; i < HIST_TILES; i++) {
            if (!have[i]) continue;
            hist[(uint64_t)WDIG * stile_first[st[i].plane] + (uint64_t)d * st[i].ntile + st[i].t] = h[i][d];
            if (h[i][d]) atomicAdd(&dig_tot[st[i].plane * WDIG + d], h[i][d]);
        }
    }
}

// per plane: exclusive scan of its histogram matrix in (digit, tile) order, on top of the plane's first record.
// A workgroup takes WSCAN_DIGITS consecutive digits of one plane; where they start comes from the per-digit totals.
constexpr uint32_t WSCAN_DIGITS = 16;
__global__ __launch_bounds__(1024) void k_wsort_scan(uint32_t *__restrict__ hist, const uint32_t *__restrict__ meta, uint32_t nplanes,
                                                     const uint32_t *__restrict__ dig_tot) {
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry;
    const uint32_t *plane_ev0 = meta + WMETA_EV0, *stile_first = plane_ev0 + nplanes + 1;
    const uint32_t plane = blockIdx.y, d0 = blockIdx.x * WSCAN_DIGITS;
    const uint32_t ntile = stile_first[plane + 1] - stile_first[plane];
    if (ntile == 0) return;
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
    {  // records of this plane with a smaller digit
        const uint32_t v = threadIdx.x < d0 ? dig_tot[plane * WDIG + threadIdx.x] : 0u;  // d0 <= 512 < 1024 threads
        const uint32_t inc = wave_incl_scan(v);
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t c = plane_ev0[plane];
            for (uint32_t w = 0; w < 16; w++) c += wsum[w];
            carry = c;
        }
        __syncthreads();
    }
    uint32_t *m = hist + (uint64_t)WDIG * stile_first[plane] + (uint64_t)d0 * ntile;
    const uint32_t n = WSCAN_DIGITS * ntile;
    for (uint32_t base = 0; base < n; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < n ? m[i] : 0;
        const uint32_t inc = wave_incl_scan(v);
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        uint32_t woff = 0;
        for (uint32_t w = 0; w < wave; w++) woff += wsum[w];
        const uint32_t c = carry;
        if (i < n) m[i] = c + woff + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = c + woff + inc;
        __syncthreads();
    }
}

// Stable scatter of one digit.  Wave w of a tile owns records [w * 1024, (w + 1) * 1024) of it, 64 at a time in
// order; the lanes that share a digit rank themselves with one ballot per digit bit.  The tile is put in digit order in
// LDS first and written from there, so that neighbouring lanes write neighbouring records: with 512 digits a tile has runs
// of ~8 records per digit, and written straight from the ranking every record was a partial line of its own (the first
// pass took three times as long as the second, which has a handful of digits).
// (REORDER = false: written straight from the ranking -- the second pass, whose digits are the high context bits: 0.35 ms
// against 0.60 through LDS, which then only costs occupancy.)
template <bool REORDER>
__global__ __launch_bounds__(WTHREADS) void k_wsort_scatter(const uint64_t *__restrict__ src, uint64_t *__restrict__ dst,
                                                            const uint32_t *__restrict__ meta, uint32_t nplanes, uint32_t shift,
                                                            const uint32_t *__restrict__ hist) {
    __shared__ uint32_t run[WTHREADS / 64][WDIG];  // per wave and digit: count, then the place (in the tile / in dst) of the wave's next record of it
    __shared__ uint32_t gbase[REORDER ? WDIG : 1];  // where the tile's records of a digit go, minus their place in the tile
    __shared__ uint64_t sorted[REORDER ? WT : 1];   // the tile in digit order
    __shared__ uint32_t wsum[WTHREADS / 64];
    static_assert(WDIG == 2 * WTHREADS, "a thread scans two digits");
    SortTile st;
    if (!sort_tile(meta, nplanes, blockIdx.x, st)) return;
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
    for (uint32_t d = threadIdx.x; d < WDIG * (WTHREADS / 64); d += WTHREADS) (&run[0][0])[d] = 0;
    __syncthreads();
    // this wave's records, batch b = records w * 1024 + b * 64 + lane
    uint64_t rec[16];
    const uint32_t wfirst = st.begin + wave * 1024u;
#pragma unroll
    for (uint32_t b = 0; b < 16; b++) {
        const uint32_t j = wfirst + b * 64u + lane;
        rec[b] = j < st.end ? src[j] : ~0ull;
    }
#pragma unroll
    for (uint32_t b = 0; b < 16; b++) count_digit(run[wave], (rec_ctx(rec[b]) >> shift) & (WDIG - 1u), wfirst + b * 64u + lane < st.end);
    __syncthreads();
    const uint32_t *stile_first = meta + WMETA_EV0 + nplanes + 1;
    const uint32_t *off = hist + (uint64_t)WDIG * stile_first[st.plane];
    if (!REORDER) {  // counts -> where each wave's records of each digit go
        for (uint32_t d = threadIdx.x; d < WDIG; d += WTHREADS) {
            uint32_t at = off[(uint64_t)d * st.ntile + st.t];
            for (uint32_t w = 0; w < WTHREADS / 64; w++) {
                const uint32_t n = run[w][d];
                run[w][d] = at;
                at += n;
            }
        }
    } else {  // counts -> the place in the tile of each wave's records of each digit (digit-major, then wave), and where the digit goes
        const uint32_t d0 = 2u * threadIdx.x;
        uint32_t c[2] = {0u, 0u};
#pragma unroll
        for (uint32_t i = 0; i < 2; i++)
            for (uint32_t w = 0; w < WTHREADS / 64; w++) c[i] += run[w][d0 + i];
        uint32_t total;
        uint32_t at = block_excl_scan(c[0] + c[1], wsum, &total);
#pragma unroll
        for (uint32_t i = 0; i < 2; i++) {
            gbase[d0 + i] = off[(uint64_t)(d0 + i) * st.ntile + st.t] - at;
            for (uint32_t w = 0; w < WTHREADS / 64; w++) {
                const uint32_t n = run[w][d0 + i];
                run[w][d0 + i] = at;
                at += n;
            }
        }
    }
    __syncthreads();
    uint32_t *myrun = run[wave];
#pragma unroll
    for (uint32_t b = 0; b < 16; b++) {
        const bool ev = wfirst + b * 64u + lane < st.end;
        if (__ballot(ev) == 0) break;
        const uint32_t d = (rec_ctx(rec[b]) >> shift) & (WDIG - 1u);
        const uint64_t ev_mask = __ballot(ev);
        uint32_t m_lo = (uint32_t)ev_mask, m_hi = (uint32_t)(ev_mask >> 32);
#pragma unroll
        for (uint32_t bit = 0; bit < 9; bit++) {
            const uint32_t t = (uint32_t)((int32_t)(d << (31 - bit)) >> 31);  // all ones if the bit is set
            const uint64_t bb = __ballot(ev && t != 0);
            m_lo &= ~((uint32_t)bb ^ t);
            m_hi &= ~((uint32_t)(bb >> 32) ^ t);
        }
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi(m_hi, __builtin_amdgcn_mbcnt_lo(m_lo, 0u));
        const uint32_t group = (uint32_t)__popc(m_lo) + (uint32_t)__popc(m_hi);
        uint32_t place = 0;
        if (ev) place = myrun[d] + rank;
        __builtin_amdgcn_wave_barrier();
        if (ev && rank == 0) myrun[d] = place + group;
        __builtin_amdgcn_wave_barrier();
        if (ev) {
            if (REORDER) sorted[place] = rec[b]; else dst[place] = rec[b];
        }
    }
    if (!REORDER) return;
    __syncthreads();
    const uint32_t n = st.end - st.begin;
#pragma unroll
    for (uint32_t k = 0; k < WT / WTHREADS; k++) {
        const uint32_t place = k * WTHREADS + threadIdx.x;
        if (place < n) {
            const uint64_t r = sorted[place];
            dst[gbase[(rec_ctx(r) >> shift) & (WDIG - 1u)] + place] = r;
        }
    }
}

// chain heads of the sorted records: first record of a plane, or a context different from the one before.
// heads[h] = record index | plane << 32.  A workgroup takes HEAD_TILES consecutive sort tiles, a wave 1024 consecutive
// records of each; the workgroup reserves the places of all its heads with ONE atomic (one counter takes ~90 atomics per
// microsecond: a batch of 16 4K frames has 45 000 waves with a head, and one atomic per such wave was 0.34 ms of waiting).
// So the list is in (plane, context) order except for the order in which the workgroups arrive: neighbours in the list are
// chains of neighbouring contexts, i.e. of similar length (k_wide_chains_quad relies on that for balance, not for correctness).
constexpr uint32_t HEAD_TILES = 4;
__global__ __launch_bounds__(WTHREADS) void k_wide_heads(const uint64_t *__restrict__ recs, const uint32_t *__restrict__ meta,
                                                         uint32_t nplanes, uint64_t *__restrict__ heads, uint32_t *__restrict__ nheads) {
    __shared__ uint64_t masks[HEAD_TILES][WTHREADS / 64][16];  // per tile, wave and batch of 64 records: which are heads
    __shared__ uint32_t counts[HEAD_TILES][WTHREADS / 64];
    __shared__ uint32_t wg_base;
    const uint32_t lane = lane_id(), wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t *lo32 = (const uint32_t *)recs;  // the context is in the low dword of a record
    constexpr uint32_t CTX_MASK = (1u << REC_CTX_BITS) - 1u;
    for (uint32_t t = 0; t < HEAD_TILES; t++) {
        SortTile st;
        uint32_t count = 0;
        uint64_t hm[16];
#pragma unroll
        for (uint32_t b = 0; b < 16; b++) hm[b] = 0;
        if (sort_tile(meta, nplanes, blockIdx.x * HEAD_TILES + t, st) && st.begin + wave * 1024u < st.end) {
            const uint32_t plane_first = meta[WMETA_EV0 + st.plane];
            const uint32_t wfirst = st.begin + wave * 1024u;
            uint32_t cur[16], prev[16];
#pragma unroll
            for (uint32_t b = 0; b < 16; b++) {  // (all loads before the first ballot)
                const uint32_t jc = min(wfirst + b * 64u + lane, st.end - 1u);
                cur[b] = lo32[2ull * jc] & CTX_MASK;
                prev[b] = lo32[2ull * max(jc, plane_first + 1u) - 2ull] & CTX_MASK;  // (a plane's first record is a head by its index)
            }
#pragma unroll
            for (uint32_t b = 0; b < 16; b++) {
                const uint32_t j = wfirst + b * 64u + lane;
                hm[b] = __ballot(j < st.end && (j == plane_first || cur[b] != prev[b]));
                count += (uint32_t)__popcll(hm[b]);
            }
        }
        if (lane < 16u) {
            uint64_t m = 0;
#pragma unroll
            for (uint32_t b = 0; b < 16; b++) m = lane == b ? hm[b] : m;
            masks[t][wave][lane] = m;
        }
        if (lane == 0) counts[t][wave] = count;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (uint32_t i = 0; i < HEAD_TILES * (WTHREADS / 64); i++) total += (&counts[0][0])[i];
        wg_base = total ? atomicAdd(nheads, total) : 0u;
    }
    __syncthreads();
    uint32_t at = wg_base;
    for (uint32_t t = 0; t < HEAD_TILES; t++) {
        for (uint32_t w = 0; w < WTHREADS / 64; w++) {
            const uint32_t c = counts[t][w];
            if (w == wave && c != 0) {  // (this wave's heads of this tile)
                SortTile st;
                (void)sort_tile(meta, nplanes, blockIdx.x * HEAD_TILES + t, st);
                const uint32_t wfirst = st.begin + wave * 1024u;
                uint32_t o = at;
                for (uint32_t b = 0; b < 16; b++) {
                    const uint64_t m = masks[t][wave][b];
                    if ((m >> lane) & 1ull) heads[o + mbcnt(m)] = (uint64_t)(wfirst + b * 64u + lane) | ((uint64_t)st.plane << 32);
                    o += (uint32_t)__popcll(m);
                }
            }
            at += c;
        }
    }
}

constexpr int WIDE_NK = 15;             // K_VALUES = 0..=14 (traits.rs:36)
constexpr uint32_t WIDE_HALVE = 1024;   // COUNT_SCALING (traits.rs:40)

// Four lanes per chain, sixteen chains per wave.  A batch of 4K frames has tens of thousands of chains of a few thousand
// events each; replaying them event by event, the fifteen counters of a chain spread over four lanes (lane s of the four
// holds k = s, s + 4, s + 8, s + 12; the sixteenth slot is a dummy that never wins), costs ~30 instructions per event and
// sixteen chains against the ~500 per 64 events and ONE chain of the wave-wide form below, with no prefix sum in the
// dependency chain.  What it costs is the latency of the longest chain, so a chain is replayed for `limit` events at most
// and the rest of it -- position and counters -- handed to the wave-wide kernel (long_heads / long_state / nlong).
// k of an event = the k of the smallest key S[k] << 4 | (14 - k) (smallest counter, ties to the largest k:
// parameter_selection.rs:71-85), reduced over the four lanes with two quad permutes; all counters are halved after an
// event that lifts the smallest above 1024 (:58-68).
// Memory: the wave works through its chains in chunks of 32 records per chain.  A chunk is loaded a whole chunk ahead
// (four 16-byte loads per lane), parked in LDS when it is needed, and read from there four records per round; the k bytes of
// a chunk (lane s of the four keeps event s of every round) wait in LDS too and leave with eight stores right before
// the next chunk's loads are issued.  So the only wait for memory is the one for a chunk loaded a chunk's time earlier, with
// no younger store in front of it: vmcnt counts loads and stores in order on gfx9, and with the loads two rounds ahead and
// a k store per event the kernel ran at the latency of a load per two rounds (0.23 us per event; 0.5 us with the stores).
// What is left (profiles/r03/quad_stamps.txt): 41 % of a wave's time in the rounds, 44 % in front of the address unit with
// the eight scattered byte stores and the four loads behind them (a scattered byte costs the chip ~11 ps, the microbenchmark
// profiles/tools/micro/byte_scatter.hip: 87 M of them per step are 1.0 ms); spreading the stores over the rounds was slower.
constexpr uint32_t QUAD_ROUND = 4;                                   // events per round: one per lane of the four
constexpr uint32_t QUAD_CHUNK = 32;                                  // records per chain and chunk
constexpr uint32_t QUAD_ROUNDS = QUAD_CHUNK / QUAD_ROUND;
constexpr uint32_t QUAD_STRIDE = QUAD_CHUNK * 2 + 4;                 // dwords per chain in LDS: + 4 spreads the chains over the banks
constexpr uint32_t QUAD_DUMMY = 1u << 27;                            // the sixteenth counter: above every real one, and S << 4 still fits

__device__ __forceinline__ uint4 load_rec_pair(const uint64_t *p) {
    uint4 v;
    __builtin_memcpy(&v, p, 16);
    return v;
}
__device__ __forceinline__ uint32_t quad_min(uint32_t v) {
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)v, 0xB1, 0xF, 0xF, false));  // quad_perm:[1,0,3,2]
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)v, 0x4E, 0xF, 0xF, false));  // quad_perm:[2,3,0,1]
    return v;
}

#ifdef FELICS_QUAD_STAMPS
// Diagnostic build only: s_memtime ticks of every wave, summed by phase: [0] wait for the chunk + parking it in LDS,
// [1] the k stores of the chunk before, [2] issuing the next chunk's loads, [3] the rounds, [4] chunks, [5] set-up per group.
__device__ unsigned long long g_quad_stamps[8];
#define QSTAMP(i)                                                      \
    do {                                                               \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();  \
        qst[i] += now_ - qlast;                                        \
        qlast = now_;                                                  \
    } while (0)
extern "C" __attribute__((visibility("default"))) int felics_debug_quad_stamps(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_quad_stamps), sizeof(g_quad_stamps)) != hipSuccess) return -1;
    if (reset) {
        static unsigned long long z[8] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_quad_stamps), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#else
#define QSTAMP(i)
#endif

__global__ __launch_bounds__(64) void k_wide_chains_quad(const uint64_t *__restrict__ recs, const uint32_t *__restrict__ meta,
                                                        uint32_t nplanes, uint32_t npix, const uint64_t *__restrict__ heads,
                                                        const uint32_t *__restrict__ nheads, uint8_t *__restrict__ k_map,
                                                        uint32_t limit, uint64_t *__restrict__ long_heads,
                                                        uint32_t *__restrict__ long_state, uint32_t *__restrict__ nlong) {
    constexpr uint32_t CTX_MASK = (1u << REC_CTX_BITS) - 1u, E_MASK = (1u << REC_E_BITS) - 1u;
    __shared__ __attribute__((aligned(16))) uint32_t recbuf[16 * QUAD_STRIDE];
    __shared__ uint2 kbuf[QUAD_ROUNDS][64];  // {where, k} of every lane's event of every round of the chunk
    const uint32_t lane = lane_id(), sub = lane & 3u, chain = lane >> 2;
    const uint32_t nchains = *nheads;
    const uint32_t nrecs = meta[0];         // (the record buffer is readable for 64 records past this)
    const uint32_t spare = nplanes * npix;  // (k_map is STAGE_PAD bytes longer than the planes)
    const uint32_t floor3 = sub == 3u ? QUAD_DUMMY : 0u;
    uint32_t add[4], low[4];  // per event S[i] += (e >> k_i) + 1 + k_i; key_i = S[i] << 4 | 14 - k_i
#pragma unroll
    for (uint32_t i = 0; i < 4; i++) {
        add[i] = 1u + sub + 4u * i;
        low[i] = 14u - min(sub + 4u * i, 14u);
    }
    uint32_t *const mine = recbuf + chain * QUAD_STRIDE;
#ifdef FELICS_QUAD_STAMPS
    unsigned long long qst[8] = {}, qlast = __builtin_amdgcn_s_memtime();
#endif
    for (uint32_t g = blockIdx.x; g * 16u < nchains; g += gridDim.x) {
        const uint32_t c = g * 16u + chain;
        const bool have = c < nchains;
        const uint64_t hd = have ? heads[c] : 0ull;
        const uint32_t j0 = (uint32_t)hd, plane = (uint32_t)(hd >> 32);
        const uint32_t plane_end = have ? meta[WMETA_EV0 + plane + 1] : 0u;
        const uint32_t stop = (uint32_t)min((uint64_t)plane_end, (uint64_t)j0 + limit);  // the chain's events here: j0 .. stop - 1 at most
        const uint32_t ctx = (uint32_t)recs[min(j0, nrecs)] & CTX_MASK;
        const uint32_t kbase = plane * npix;  // (a pass has fewer than 2^32 samples)
        uint32_t S[4] = {0u, 0u, 0u, floor3};
        uint32_t mkey = 0;  // smallest key of the chain's counters: all zero -> k = 14
        uint32_t j = j0;    // first record of the round in hand
        bool alive = have;
        // one event of the chain: counters, smallest key, halving
        auto event = [&](uint32_t lo_q, uint32_t hi_q) {
            const uint32_t e = __builtin_amdgcn_alignbit(hi_q, lo_q, REC_CTX_BITS) & E_MASK;
            uint32_t t = e >> sub;
#pragma unroll
            for (uint32_t i = 0; i < 4; i++) {
                S[i] += t + add[i];  // rice_coding.rs:40-46
                t >>= 4;
            }
            mkey = quad_min(min(min((S[0] << 4) | low[0], (S[1] << 4) | low[1]), min((S[2] << 4) | low[2], (S[3] << 4) | low[3])));
            const bool halve = mkey >= ((WIDE_HALVE + 1u) << 4);
            if (__any(halve)) {
                const uint32_t h = halve ? 1u : 0u;
#pragma unroll
                for (uint32_t i = 0; i < 4; i++) S[i] >>= h;
                S[3] = max(S[3], floor3);
                mkey = quad_min(min(min((S[0] << 4) | low[0], (S[1] << 4) | low[1]), min((S[2] << 4) | low[2], (S[3] << 4) | low[3])));
            }
        };
        // the lane's eighth of a chunk: records at .. at + 7 of its chain
        uint4 R[4];
        auto fetch = [&](uint32_t at) {
            const uint64_t *p = recs + min(at + 8u * sub, nrecs);
#pragma unroll
            for (uint32_t i = 0; i < 4; i++) R[i] = load_rec_pair(p + 2u * i);
        };
        fetch(j);
        bool pending = false;  // a chunk's k bytes wait in kbuf
        QSTAMP(5);
        while (__any(alive)) {
#pragma unroll
            for (uint32_t i = 0; i < 4; i++) *(uint4 *)(mine + 16u * sub + 4u * i) = R[i];  // (the wait for the chunk: nothing younger in front)
            QSTAMP(0);
            if (pending) {
#pragma unroll
                for (uint32_t r = 0; r < QUAD_ROUNDS; r++) {
                    const uint2 w = kbuf[r][lane];
#ifdef FELICS_DIAG_NO_K  // timing only (wrong output): the walk without its k stores
                    if (w.y == 0xFFu)
#endif
                    k_map[w.x] = (uint8_t)w.y;
                }
            }
            QSTAMP(1);
            fetch(j + QUAD_CHUNK);
            QSTAMP(2);
            for (uint32_t r = 0; r < QUAD_ROUNDS; r++) {
                const uint4 r0 = *(const uint4 *)(mine + 8u * r), r1 = *(const uint4 *)(mine + 8u * r + 4u);
                const uint32_t my_hi = mine[8u * r + 2u * sub + 1u];
                const uint32_t lo[QUAD_ROUND] = {r0.x, r0.z, r1.x, r1.z}, hi[QUAD_ROUND] = {r0.y, r0.w, r1.y, r1.w};
                uint32_t my_at = spare, my_k = 0;
                // the records are sorted by context: the round lies inside the chain if its last record does
                const bool full = alive && j + (QUAD_ROUND - 1u) < stop && (lo[QUAD_ROUND - 1u] & CTX_MASK) == ctx;
                if (__all(full || !alive)) {
                    if (alive) {
#pragma unroll
                        for (uint32_t q = 0; q < QUAD_ROUND; q++) {
                            my_k = sub == q ? mkey : my_k;
                            event(lo[q], hi[q]);
                        }
                        my_at = kbase + (my_hi >> (REC_CTX_BITS + REC_E_BITS - 32u));
                    }
                } else {  // a chain of the wave ends in this round
#pragma unroll
                    for (uint32_t q = 0; q < QUAD_ROUND; q++) {
                        alive = alive && j + q < stop && (lo[q] & CTX_MASK) == ctx;
                        if (alive) {
                            if (sub == q) my_k = mkey, my_at = kbase + (my_hi >> (REC_CTX_BITS + REC_E_BITS - 32u));
                            event(lo[q], hi[q]);
                        }
                    }
                }
                kbuf[r][lane] = make_uint2(my_at, 14u - (my_k & 15u));
                j += QUAD_ROUND;
            }
            pending = true;
            QSTAMP(3);
#ifdef FELICS_QUAD_STAMPS
            qst[4]++;
#endif
        }
        if (pending) {
#pragma unroll
            for (uint32_t r = 0; r < QUAD_ROUNDS; r++) {
                const uint2 w = kbuf[r][lane];
#ifdef FELICS_DIAG_NO_K
                if (w.y == 0xFFu)
#endif
                k_map[w.x] = (uint8_t)w.y;
            }
        }
        // a chain that goes on behind its last event here: the records are sorted by context inside the plane, so it does if
        // the record at `stop` is of this plane and this context
        const bool more = have && stop < plane_end && ((uint32_t)recs[min(stop, nrecs)] & CTX_MASK) == ctx;
        uint32_t at = 0;
        if (more && sub == 0u) at = atomicAdd(nlong, 1u);
        at = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)at, 0x00, 0xF, 0xF, false);  // quad_perm:[0,0,0,0]: the four lanes' slot
        if (more) {
            if (sub == 0u) long_heads[at] = (uint64_t)stop | ((uint64_t)plane << 32);
#pragma unroll
            for (uint32_t i = 0; i < 4; i++)
                if (sub + 4u * i < (uint32_t)WIDE_NK) long_state[(uint64_t)at * 16u + sub + 4u * i] = S[i];
        }
        QSTAMP(5);
    }
#ifdef FELICS_QUAD_STAMPS
    if (lane == 0)
        for (int i = 0; i < 6; i++) atomicAdd(&g_quad_stamps[i], qst[i]);
#endif
}

// One wave per chain (persistent grid).  Lane l of a step holds event j + l of the chain.  For every
// Rice parameter the lanes' code lengths are prefix-summed, which gives each lane the counters as
// they were before its event (-> its k: smallest counter, ties to the largest k, parameter_selection.rs
// :71-85) and after it.  The counters are halved after the first event that lifts their minimum above
// 1024 (:58-68); that minimum never decreases along the block, so the event is found with one ballot and
// the counters are halved there in place (the prefix sums stay valid for the lanes behind it).
// The next block's records are in flight while this one is resolved.
__global__ __launch_bounds__(256) void k_wide_chains(const uint64_t *__restrict__ recs, const uint32_t *__restrict__ meta,
                                                     uint32_t nplanes, uint32_t npix, const uint64_t *__restrict__ heads,
                                                     const uint32_t *__restrict__ nheads, const uint32_t *__restrict__ resume,
                                                     uint8_t *__restrict__ k_map) {
    const uint32_t lane = lane_id();
    const uint32_t nchains = *nheads;
    const uint32_t wave0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t c = wave0; c < nchains; c += nwaves) {
        const uint64_t hd = heads[c];
        uint32_t j = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)hd);  // wave-uniform: chain position and counters stay scalar
        const uint32_t plane = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(hd >> 32));
        const uint32_t plane_end = meta[WMETA_EV0 + plane + 1];
        uint8_t *kp = k_map + (uint64_t)plane * npix;
        const uint32_t ctx = rec_ctx(recs[j]);
        uint32_t S[WIDE_NK];  // (resume: the counters k_wide_chains_quad left, 16 words per chain)
#pragma unroll
        for (int k = 0; k < WIDE_NK; k++)
            S[k] = resume ? (uint32_t)__builtin_amdgcn_readfirstlane((int)resume[(uint64_t)c * 16u + (uint32_t)k]) : 0u;
        // Records are streamed two blocks ahead with unconditional loads (index clamped to the plane's last record) and k
        // leaves with an unconditional store (lanes without an event write a spare byte behind the planes): no branch around
        // a memory operation, so the wait for a block's records is a counted one that leaves the younger operations -- the
        // two prefetches and the scattered k bytes of the block before -- in flight.  (With the loads and the store under
        // conditions the compiler waited for everything, the k bytes included, once per block: 16 us per 64 events.)
        const uint32_t last_rec = plane_end - 1u;
        uint8_t *const spare = k_map + (uint64_t)nplanes * npix;  // (the buffer is STAGE_PAD bytes longer than the planes)
        uint64_t cur = __builtin_nontemporal_load(&recs[min(j + lane, last_rec)]);
        uint64_t nx1 = __builtin_nontemporal_load(&recs[min(j + 64u + lane, last_rec)]);
        for (;;) {
            const bool valid = j + lane < plane_end && rec_ctx(cur) == ctx;
            const uint64_t vm = __ballot(valid);  // a prefix of the lanes
            const uint32_t nvalid = (uint32_t)__popcll(vm);
            if (nvalid == 0) break;
            const uint64_t mine = cur;
            const uint64_t nx2 = __builtin_nontemporal_load(&recs[min(j + 128u + lane, last_rec)]);  // two blocks ahead
            const uint32_t pix = rec_pix(mine);
            const uint32_t e = valid ? rec_e(mine) : 0u;
            // One set of prefix sums per block; a halving at lane f turns the counters into ((S + P(f)) >> 1) - P(f), to which the
            // lanes behind f add their own P(t) >= P(f) again (mod 2^32) -- no second scan.
            // The counters before a lane's event are S + (P - len): as keys ((P - len) << 4 | 14 - k) + (S << 4), whose minimum over k
            // names the smallest counter with ties to the largest k -- one add per counter and round instead of add, subtract,
            // compare and two selects (the part in brackets does not change when the counters are halved).
            uint32_t P[WIDE_NK], keyB[WIDE_NK];
#pragma unroll
            for (int k = 0; k < WIDE_NK; k++) {
                const uint32_t len = valid ? (e >> k) + 1u + (uint32_t)k : 0u;  // rice_coding.rs:40-46
                P[k] = wave_incl_scan(len);
                keyB[k] = ((P[k] - len) << 4) | (uint32_t)(14 - k);
            }
            uint32_t base = 0;  // first event of the block not yet resolved
            uint32_t my_k = 0;
            while (true) {
                uint32_t key_min = 0xFFFFFFFFu, after_min = 0xFFFFFFFFu;
#pragma unroll
                for (int k = 0; k < WIDE_NK; k++) {
                    key_min = min(key_min, (S[k] << 4) + keyB[k]);  // (mod 2^32: S may have wrapped, S + P - len has not)
                    after_min = min(after_min, S[k] + P[k]);
                }
                const uint32_t best_k = 14u - (key_min & 15u);
                const bool live = valid && lane >= base;
                const uint64_t hm = __ballot(live && after_min > WIDE_HALVE);
                const uint32_t f = hm ? (uint32_t)__builtin_ctzll(hm) : nvalid - 1u;  // last event served by these counters
                if (live && lane <= f) my_k = best_k;
                if (!hm) {
#pragma unroll
                    for (int k = 0; k < WIDE_NK; k++) S[k] += readlane(P[k], nvalid - 1u);
                    break;
                }
#pragma unroll
                for (int k = 0; k < WIDE_NK; k++) {
                    const uint32_t pf = readlane(P[k], f);
                    S[k] = ((S[k] + pf) >> 1) - pf;
                }
                base = f + 1u;
                if (base >= nvalid) {  // the halving fell on the block's last event: carry the counters over as they are
#pragma unroll
                    for (int k = 0; k < WIDE_NK; k++) S[k] += readlane(P[k], nvalid - 1u);
                    break;
                }
            }
            *(valid ? kp + pix : spare) = (uint8_t)my_k;
            if (nvalid < 64u) break;
            j += 64u;
            cur = nx1;
            nx1 = nx2;
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

WideSizes wide_sizes(const Geometry &g) {
    WideSizes z;
    const uint64_t nsamples = (uint64_t)g.nplanes * g.npix;
    z.px_tiles = cdiv(g.npix, WT);
    z.max_sort_tiles = (uint32_t)(nsamples / WT + g.nplanes);
    z.tile_cnt_bytes = (size_t)z.px_tiles * g.nplanes * 4;
    z.meta_bytes = (size_t)(WMETA_EV0 + 2 * (g.nplanes + 1)) * 4;
    z.rec_bytes = (size_t)nsamples * 8 + 512;  // (k_wide_chains_quad reads up to 39 records past the last)
    z.hist_bytes = (size_t)z.max_sort_tiles * WDIG * 4;
    z.heads_bytes = (size_t)nsamples * 8 + 64;
    z.digtot_bytes = (size_t)g.nplanes * WDIG * 4;
    return z;
}

template <typename T>
void launch_wide_events(hipStream_t s, const T *planes, uint32_t *tile_cnt, uint32_t *meta, uint64_t *recs, const Geometry &g) {
    const WideSizes z = wide_sizes(g);
    FELICS_LAUNCH((k_wide_count<T>), dim3(z.px_tiles, g.nplanes), dim3(WTHREADS), s, planes, g.W, g.npix, tile_cnt);
    FELICS_LAUNCH(k_wide_plan, dim3(1), dim3(1024), s, tile_cnt, z.px_tiles, g.nplanes, meta);
    FELICS_LAUNCH((k_wide_emit<T>), dim3(z.px_tiles, g.nplanes), dim3(WTHREADS), s, planes, g.W, g.npix, tile_cnt, recs);
}
template void launch_wide_events<uint16_t>(hipStream_t, const uint16_t *, uint32_t *, uint32_t *, uint64_t *, const Geometry &);
template void launch_wide_events<int32_t>(hipStream_t, const int32_t *, uint32_t *, uint32_t *, uint64_t *, const Geometry &);

// two stable passes (low, then high nine context bits): the sorted records end up in recs_a again
void launch_wide_sort(hipStream_t s, uint64_t *recs_a, uint64_t *recs_b, const uint32_t *meta, uint32_t *hist, uint32_t *dig_tot,
                      const Geometry &g) {
    const WideSizes z = wide_sizes(g);
    uint64_t *src = recs_a, *dst = recs_b;
    for (uint32_t shift = 0; shift < REC_CTX_BITS; shift += 9) {
        (void)hipMemsetAsync(dig_tot, 0, z.digtot_bytes, s);
        FELICS_LAUNCH(k_wsort_hist, dim3(cdiv(z.max_sort_tiles, HIST_TILES)), dim3(WTHREADS), s, src, meta, g.nplanes, shift, hist, dig_tot);
        FELICS_LAUNCH(k_wsort_scan, dim3(WDIG / WSCAN_DIGITS, g.nplanes), dim3(1024), s, hist, meta, g.nplanes, dig_tot);
        if (shift == 0)
            FELICS_LAUNCH(k_wsort_scatter<true>, dim3(z.max_sort_tiles), dim3(WTHREADS), s, src, dst, meta, g.nplanes, shift, hist);
        else
            FELICS_LAUNCH(k_wsort_scatter<false>, dim3(z.max_sort_tiles), dim3(WTHREADS), s, src, dst, meta, g.nplanes, shift, hist);
        std::swap(src, dst);
    }
}

void launch_wide_chains(hipStream_t s, const uint64_t *recs, const uint32_t *meta, uint64_t *heads, uint32_t *counters,
                        uint8_t *k_map, const Geometry &g, uint32_t lane_limit, uint64_t *long_heads, uint32_t *long_state) {
    const WideSizes z = wide_sizes(g);
    uint32_t *nheads = counters, *nlong = counters + 1;
    FELICS_LAUNCH(k_wide_heads, dim3(cdiv(z.max_sort_tiles, HEAD_TILES)), dim3(WTHREADS), s, recs, meta, g.nplanes, heads, nheads);
    if (lane_limit != 0) {
        // four lanes per chain up to lane_limit events of it (persistent: eight waves per SIMD at most), the wave-wide kernel
        // for what is left of longer chains
        FELICS_LAUNCH(k_wide_chains_quad, dim3(256u * 32u), dim3(64), s, recs, meta, g.nplanes, g.npix, heads, nheads, k_map, lane_limit,
                      long_heads, long_state, nlong);
        FELICS_LAUNCH(k_wide_chains, dim3(256u * 8u), dim3(256), s, recs, meta, g.nplanes, g.npix, long_heads, nlong,
                      (const uint32_t *)long_state, k_map);
        return;
    }
    // persistent: 8 workgroups of 4 waves per CU share the chains
    FELICS_LAUNCH(k_wide_chains, dim3(256u * 8u), dim3(256), s, recs, meta, g.nplanes, g.npix, heads, nheads,
                  (const uint32_t *)nullptr, k_map);
}

// How many events of a chain one lane replays before the wave-wide kernel takes over (0: the wave-wide kernel does everything).
// The four-lane form needs ~0.3 us per event of the longest chain whatever the batch; the wave-wide kernel gets through a batch
// at ~2 ns per 64 events.  So the four-lane form pays when the batch is large enough for the chip to be busy with whole chains
// (profiles/r03/wide_chains_sweep.txt).
uint32_t wide_lane_limit(const Geometry &g) {
    const uint64_t nsamples = (uint64_t)g.nplanes * g.npix;
    if (nsamples < WIDE_LANE_MIN_SAMPLES) return 0;
    // (a chain costs the four-lane kernel ~0.3 us per event however long the others are: with a limit of 16 000 events a batch
    // of height maps, whose smallest contexts hold chains of 100 000 events, took 5.4 ms against 4.3 for the wave-wide kernel
    // alone, a natural-like batch 7.2 against 6.9 -- and 5.9 with the limit at 2 000-4 000; profiles/r03/content16.txt)
    return (uint32_t)std::min<uint64_t>(std::max<uint64_t>(nsamples / 32768u, WIDE_LANE_LIMIT_MIN), WIDE_LANE_LIMIT_MAX);
}
// chains longer than the limit: fewer than nsamples / limit of them (+ one per plane for the rounding)
size_t wide_long_capacity(const Geometry &g, uint32_t lane_limit) {
    return lane_limit ? (size_t)((uint64_t)g.nplanes * g.npix / lane_limit + g.nplanes + 1) : 0;
}

}  // namespace felics
