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
//            (parameter_selection.rs:49-85): spine (sequential, per chain) + assign (parallel)
//   pack     build the codes (rice_coding.rs:26-38, phase_in_coding.rs:59-84,
//            compression.rs:29-45) and pack them MSB-first (bitstream-io BigEndian).
//            Single pass (k_pack_g: code lengths, tile offsets by decoupled look-back, packing),
//            or two passes for exact placement / 16-bit samples / as a fallback:
//   lengths    code length of every pixel -> bits per tile
//   bitscan    exclusive scan of tile bits -> bit offset of every tile in its stream
//   pack       the codes again, packed at those offsets
//
// Integer work only: no MFMA.  Wave = 64 lanes everywhere.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "felics_device.h"
#include "felics_kernels.h"
#include "felics_codes.h"

namespace felics {

thread_local LaunchTiming g_launch_timing;

// ------------------------------------------------------------------------------------------
// planes: interleaved RGB8 -> three int16 planes Y, Co, Cg (color_transform.rs:11-17).
// `/ 2` on int truncates toward zero exactly like Rust's.
// ------------------------------------------------------------------------------------------

__device__ __forceinline__ void ycocg(int r, int gr, int b, int &yv, int &co, int &cg) {
    co = r - b;
    const int t = b + co / 2;
    cg = gr - t;
    yv = t + cg / 2;
}

// Four pixels per thread: 12 bytes in (three dwords), four samples out per plane (one 8-byte store each).
// A group never straddles two images (it starts at a multiple of four inside its image; the last pixels
// of an image whose size is not a multiple of four take the single-pixel path).
__global__ __launch_bounds__(256) void k_rgb8_to_planes(const uint8_t *__restrict__ rgb, int16_t *__restrict__ planes,
                                                        uint32_t npix, uint32_t nimg) {
    const uint32_t groups = (npix + 3) / 4;  // per image
    const uint64_t total = (uint64_t)groups * nimg;
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total;
         g += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t img = (uint32_t)(g / groups);
        const uint32_t i = (uint32_t)(g - (uint64_t)img * groups) * 4;
        const uint8_t *src = rgb + ((uint64_t)img * npix + i) * 3;
        int16_t *o = planes + (uint64_t)img * 3 * npix;
        if (i + 4 <= npix) {
            uint32_t w[3];
            __builtin_memcpy(w, src, 12);
            int yv[4], co[4], cg[4];
            ycocg((int)(w[0] & 0xFFu), (int)((w[0] >> 8) & 0xFFu), (int)((w[0] >> 16) & 0xFFu), yv[0], co[0], cg[0]);
            ycocg((int)(w[0] >> 24), (int)(w[1] & 0xFFu), (int)((w[1] >> 8) & 0xFFu), yv[1], co[1], cg[1]);
            ycocg((int)((w[1] >> 16) & 0xFFu), (int)(w[1] >> 24), (int)(w[2] & 0xFFu), yv[2], co[2], cg[2]);
            ycocg((int)((w[2] >> 8) & 0xFFu), (int)((w[2] >> 16) & 0xFFu), (int)(w[2] >> 24), yv[3], co[3], cg[3]);
            auto put4 = [&](int16_t *dst, const int (&v)[4]) {
                const uint32_t p[2] = {(uint32_t)(v[0] & 0xFFFF) | ((uint32_t)v[1] << 16),
                                       (uint32_t)(v[2] & 0xFFFF) | ((uint32_t)v[3] << 16)};
                __builtin_memcpy(dst, p, 8);
            };
            put4(o + i, yv);
            put4(o + (uint64_t)npix + i, co);
            put4(o + 2ull * npix + i, cg);
        } else {
            for (uint32_t j = i; j < npix; j++) {
                int yv, co, cg;
                ycocg(src[(j - i) * 3], src[(j - i) * 3 + 1], src[(j - i) * 3 + 2], yv, co, cg);
                o[j] = (int16_t)yv;
                o[(uint64_t)npix + j] = (int16_t)co;
                o[2ull * npix + j] = (int16_t)cg;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// hist: one wave per tile of SORT_TILE pixels; LDS histogram of event contexts.
// counts[(plane*ntiles + tile)*nctx + ctx]
// ------------------------------------------------------------------------------------------

template <typename T>
__global__ __launch_bounds__(256) void k_hist(const T *__restrict__ planes, uint32_t *__restrict__ counts,
                                              uint32_t W, uint32_t npix, uint32_t ntiles) {
    // COPIES histograms per wave, lane l counts in copy l % COPIES: a smooth frame has ten contexts that matter, so the 64
    // lanes of an LDS add hit a handful of addresses -- with one copy four fifths of the kernel's LDS cycles were conflicts
    // (SQ_LDS_BANK_CONFLICT 84 M of SQ_LDS_IDX_ACTIVE 103 M cycles per step).
    constexpr uint32_t NC = nctx_of<T>();
    constexpr uint32_t COPIES = 4096 / NC / 4;  // 16 KB of LDS per workgroup either way: 4 (u8) or 2 (i16)
    __shared__ uint32_t hist[4][NC * COPIES];  // [wave][ctx * COPIES + copy]: the copies of a context lie in different banks
    // (wave-uniform, and said so: the tile, its bounds and the trip bookkeeping then live in scalar registers instead of
    // vector registers under exec masks -- without it three quarters of this kernel's instructions were mask handling)
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = lane_id();
    const uint32_t tile = blockIdx.x * 4 + wave;
    const uint32_t plane = blockIdx.y;
    for (uint32_t c = lane; c < NC * COPIES; c += 64) hist[wave][c] = 0;
    uint32_t *my_hist = hist[wave] + lane % COPIES;
    __builtin_amdgcn_wave_barrier();
    if (tile < ntiles) {
        const T *pl = planes + (uint64_t)plane * npix;
        const uint32_t begin = tile * SORT_TILE;
        const uint32_t end = min(begin + SORT_TILE, npix);
        // 256 pixels per trip, four per lane.  (x0, y0) is the trip's first pixel, tracked in scalar registers; a trip whose
        // 256 pixels lie inside one image row with y > 0 -- nearly all of them -- takes the neighbour rule's interior case
        // without any per-pixel case analysis, from two wide loads per lane.  The loads of the next HIST_AHEAD trips are in
        // flight while a trip is counted: a wave walks its tile in 16 trips, and with one trip in flight the kernel was a
        // chain of 16 memory round trips per tile.
        constexpr uint32_t AHEAD = 4;
        auto is_interior = [&](uint32_t r, uint32_t x, uint32_t y) { return y > 0 && x + 256 <= W && r + 256 <= end; };
        Interior4<T> ring[AHEAD];
        bool have[AHEAD];
        uint32_t ri = begin, yi = begin / W, xi = begin - yi * W;  // the next trip to issue
        auto issue = [&](Interior4<T> &slot, bool &h) {
            h = ri < end && is_interior(ri, xi, yi);
            if (h) load_interior4(pl, ri, W, span_left_index(ri, xi, yi, W), slot);
            ri += 256;
            xi += 256;
            if (xi >= W) {  // (once per image row: scalar division)
                const uint32_t q = xi / W;
                yi += q;
                xi -= q * W;
            }
        };
#pragma unroll
        for (uint32_t d = 0; d < AHEAD; d++) issue(ring[d], have[d]);
        for (uint32_t r0 = begin; r0 < end;) {
#pragma unroll
            for (uint32_t d = 0; d < AHEAD; d++) {
                if (r0 < end) {
                    const bool interior = have[d];
                    const Interior4<T> now = ring[d];
                    issue(ring[d], have[d]);  // in flight while this trip (and the next AHEAD - 1) are counted
                    if (interior) {
                        // lane l takes pixels r0 + 4l .. + 3 (counting does not care which lane sees which pixel)
                        PixelClass pc[4];
                        classify_loaded4(now, pc);
#pragma unroll
                        for (uint32_t u = 0; u < 4; u++)
                            if (pc[u].cls != CLS_IN) atomicAdd(&my_hist[pc[u].ctx * COPIES], 1u);
                    } else {  // (a trip that crosses a row end, or lies in the first row: the general neighbour rule)
                        Coord xy;
                        xy.set(r0 + lane, W);
#pragma unroll
                        for (uint32_t u = 0; u < 4; u++) {
                            const uint32_t i = r0 + u * 64 + lane;
                            if (i < end && i >= 2) {
                                const PixelClass pc = classify(pl, i, xy.x, xy.y, W);
                                if (pc.cls != CLS_IN) atomicAdd(&my_hist[pc.ctx * COPIES], 1u);
                            }
                            xy.advance(64, W);
                        }
                    }
                    r0 += 256;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        uint32_t *dst = counts + ((uint64_t)plane * ntiles + tile) * NC;
        for (uint32_t c = lane; c < NC; c += 64) {
            uint32_t n = 0;
#pragma unroll
            for (uint32_t q = 0; q < COPIES; q++) n += hist[wave][c * COPIES + q];
            dst[c] = n;
        }
    }
}

// ------------------------------------------------------------------------------------------
// offsets: per (plane, ctx) exclusive scan over tiles (in place), chain length out.
// A workgroup takes 32 contexts of a plane; the tiles are cut into 8 segments, thread (seg, ctx) sums its
// segment, the segment totals are exchanged through LDS, and the segment is walked again writing the
// running offsets.  Lanes run over ctx, so every step is a coalesced 128-byte access; eight loads are in
// flight per thread.
// ------------------------------------------------------------------------------------------

constexpr uint32_t OFF_SEGS = 8, OFF_CTX = 32;

__global__ __launch_bounds__(OFF_SEGS *OFF_CTX) void k_tile_offsets(uint32_t *__restrict__ counts,
                                                                    uint32_t *__restrict__ chain_len, uint32_t ntiles,
                                                                    uint32_t nctx) {
    __shared__ uint32_t seg_sum[OFF_SEGS][OFF_CTX];
    const uint32_t cl = threadIdx.x % OFF_CTX, seg = threadIdx.x / OFF_CTX;
    const uint32_t plane = blockIdx.y, c = blockIdx.x * OFF_CTX + cl;
    const uint32_t per = (ntiles + OFF_SEGS - 1) / OFF_SEGS;
    const uint32_t t0 = min(seg * per, ntiles), t1 = min(t0 + per, ntiles);
    uint32_t *col = counts + (uint64_t)plane * ntiles * nctx + c;
    uint32_t sum = 0;
    uint32_t t = t0;
    for (; t + 8 <= t1; t += 8) {
        uint32_t v[8];
#pragma unroll
        for (uint32_t u = 0; u < 8; u++) v[u] = col[(uint64_t)(t + u) * nctx];
#pragma unroll
        for (uint32_t u = 0; u < 8; u++) sum += v[u];
    }
    for (; t < t1; t++) sum += col[(uint64_t)t * nctx];
    seg_sum[seg][cl] = sum;
    __syncthreads();
    uint32_t run = 0;
    for (uint32_t q = 0; q < seg; q++) run += seg_sum[q][cl];
    if (seg == OFF_SEGS - 1) chain_len[plane * nctx + c] = run + sum;
    t = t0;
    for (; t + 8 <= t1; t += 8) {  // eight independent loads in flight, then the running sum
        uint32_t v[8];
#pragma unroll
        for (uint32_t u = 0; u < 8; u++) v[u] = col[(uint64_t)(t + u) * nctx];
#pragma unroll
        for (uint32_t u = 0; u < 8; u++) {
            col[(uint64_t)(t + u) * nctx] = run;
            run += v[u];
        }
    }
    for (; t < t1; t++) {
        const uint32_t v = col[(uint64_t)t * nctx];
        col[(uint64_t)t * nctx] = run;
        run += v;
    }
}

// Exclusive scan of the chain lengths, each rounded up to a whole 64-event block, over all
// (plane, ctx) -> chain_base; single block.  total_slots = end of the last chain.
// Thread t owns a contiguous run of chains: it sums them (all loads of the run in flight together), the run totals are
// scanned across the block once, and the run is walked again writing the offsets -- two memory round trips and one
// barrier phase.  256 threads, not 1024: a workgroup of sixteen waves waits for a CU with four free wave slots on every
// SIMD at once, which a GPU full of pack and scatter workgroups does not offer for a long time (0.17 ms per launch).
constexpr uint32_t CHAIN_BASES_THREADS = 256;
__global__ __launch_bounds__(CHAIN_BASES_THREADS) void k_chain_bases(const uint32_t *__restrict__ chain_len,
                                                      uint32_t *__restrict__ chain_base, uint32_t n,
                                                      uint32_t *__restrict__ total_slots) {
    __shared__ uint32_t wsum[CHAIN_BASES_THREADS / 64];
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
    const uint32_t per = (n + CHAIN_BASES_THREADS - 1u) / CHAIN_BASES_THREADS;
    const uint32_t i0 = min(threadIdx.x * per, n), i1 = min(i0 + per, n);
    uint32_t sum = 0;
    uint32_t i = i0;
    for (; i + 8 <= i1; i += 8) {
        uint32_t v[8];
#pragma unroll
        for (uint32_t u = 0; u < 8; u++) v[u] = chain_len[i + u];
#pragma unroll
        for (uint32_t u = 0; u < 8; u++) sum += (v[u] + 63u) & ~63u;
    }
    for (; i < i1; i++) sum += (chain_len[i] + 63u) & ~63u;
    const uint32_t inc = wave_incl_scan(sum);
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t run = inc - sum;
    for (uint32_t w = 0; w < wave; w++) run += wsum[w];
    for (i = i0; i < i1; i++) {
        chain_base[i] = run;
        run += (chain_len[i] + 63u) & ~63u;
    }
    if (threadIdx.x == CHAIN_BASES_THREADS - 1) *total_slots = run;  // (threads past the last chain carry the total along)
}

// Zero the padding events at the end of every chain's last block (value 0 adds nothing to a
// block's sum of e >> k, and nothing after a chain's last real event is ever used) and mark the
// padding slots as belonging to no pixel.
template <typename ET>
__global__ void k_zero_padding(ET *__restrict__ sorted_e, uint32_t *__restrict__ pix_of,
                               const uint32_t *__restrict__ chain_base, const uint32_t *__restrict__ chain_len,
                               uint32_t nchains) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nchains) return;
    const uint32_t n = chain_len[c], base = chain_base[c];
    for (uint32_t i = n; i < ((n + 63u) & ~63u); i++) {
        sorted_e[base + i] = 0;
        if (pix_of) pix_of[base + i] = 0xFFFFFFFFu;  // (null: in-tile offsets, read by run and never in the padding)
    }
}

// ------------------------------------------------------------------------------------------
// scatter: stable partition of events by context.
// sorted_e[slot] = value to Rice-code; pix_of[slot] = the pixel the event came from: plane*npix + i (32 bits) for
// k_k_to_pixels (two-pass pack), or -- REL, what k_pack_g reads -- the pixel's offset in its sort tile (16 bits, the same buffer).
//
// Two kernels.  k_scatter (the default, round 4) sorts a tile's events in LDS and writes every context's run of the tile as
// one contiguous piece; it ranks with one returning LDS atomic per batch and CHECKS the order it got.  k_scatter_ballot (rounds
// 1-3) ranks with ballots and stores every batch of 64 events straight to the chains, in raster order: the fallback a context
// moves to if the check ever fails (felics_api.cpp: note_scatter_order_violation), and what FELICS_SCATTER_BALLOT=1 selects.
//
// k_scatter_ballot: one wave per tile walks its pixels in raster order, 64 events at a time; lanes that hold the same context
// rank themselves with a ballot.
// ------------------------------------------------------------------------------------------

template <typename T, typename ET, bool REL>
__global__ __launch_bounds__(256) void k_scatter_ballot(const T *__restrict__ planes,
                                                 const uint32_t *__restrict__ tile_off,
                                                 const uint32_t *__restrict__ chain_base,
                                                 ET *__restrict__ sorted_e, uint32_t *__restrict__ pix_of,
                                                 uint32_t W, uint32_t npix, uint32_t ntiles, uint32_t tile_begin,
                                                 uint32_t tile_end, uint32_t nplanes) {
    constexpr uint32_t RING = 1024;  // a round of two trips adds at most 512 events to fewer than 64 left over
    static_assert((RING & (RING - 1)) == 0, "the ring is indexed with a mask");
    static_assert(SORT_TILE <= (1u << 13), "ring records keep the pixel's offset in its tile in 13 bits");
    __shared__ uint32_t runs[4][nctx_of<T>()];
    __shared__ uint32_t rings[4][RING];
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = lane_id();  // (uniform: see k_hist)
    // Workgroup -> (plane, four tiles), XCD-aware: workgroups are dealt round-robin over the eight XCDs (MI355X_MICROARCH.md,
    // workgroup dispatch: blocks b and b + 8 share one), and every XCD has an L2 of its own that does not merge its partial
    // lines with another XCD's.  Neighbouring tiles of a plane append to the same cache lines of every chain, so all tiles of
    // plane p go to the XCD p % 8 (in tile order: workgroup b = 8 i + x takes item i of XCD x's list of planes x, x + 8, ...) --
    // the XCD whose spine and pack workgroups read the chains of plane p later (their grids are plane-minor with 64 planes).
    // Placement only: nothing depends on it for correctness.
    const uint32_t wg_tiles = (tile_end - tile_begin + 3u) / 4u;
    const uint32_t item = blockIdx.x >> 3, xcd = blockIdx.x & 7u;
    const uint32_t plane = xcd + 8u * (item / wg_tiles);
    if (plane >= nplanes) return;
    const uint32_t tile = tile_begin + (item % wg_tiles) * 4 + wave;
    if (tile >= tile_end) return;
    uint32_t *run = runs[wave];
    {
        constexpr uint32_t NC = nctx_of<T>();
        const uint32_t *off = tile_off + ((uint64_t)plane * ntiles + tile) * NC;
        const uint32_t *cb = chain_base + (uint64_t)plane * NC;
        for (uint32_t c = lane; c < NC; c += 64) run[c] = off[c] + cb[c];
    }
    __builtin_amdgcn_wave_barrier();
    const T *pl = planes + (uint64_t)plane * npix;
    const uint32_t plane_first = plane * npix;
    const uint32_t begin = tile * SORT_TILE;
    const uint32_t end = min(begin + SORT_TILE, npix);
    // Four rows per trip: their twelve loads are in flight together (one row at a time the kernel waits
    // for memory once per row).  (x0, y0) is the trip's first pixel (scalar); trips inside one image row
    // with x > 0, y > 0 skip the neighbour rule's case analysis.
    uint32_t *ring = rings[wave];
    uint32_t qhead = 0, qtail = 0;  // ring positions (wave-uniform)
    // ranks and stores the next n (<= 64) events of the ring
    auto drain = [&](uint32_t n) {
        const bool ev = lane < n;
        const uint32_t rec = ring[(qhead + lane) & (RING - 1u)];
        const uint32_t c = rec >> 22, e = (rec >> 13) & 0x1FFu, off = rec & 0x1FFFu;
        // Rank the lanes that share a context with ballots only: every lane learns how many earlier lanes
        // hold its context (rank) and how many hold it in all (group).  Contexts are matched bit by bit:
        // after one ballot per context bit every lane holds the mask of the lanes whose context equals its
        // own -- a fixed cost, however many different contexts the 64 events hold.  Most batches only hold
        // contexts below 32 and get away with five of the nine bits.
        const uint64_t ev_mask = __ballot(ev);
        uint32_t m_lo = (uint32_t)ev_mask, m_hi = (uint32_t)(ev_mask >> 32);
        auto match_bit = [&](uint32_t b) {
            const uint32_t t = (uint32_t)((int32_t)(c << (31 - b)) >> 31);  // all ones if bit b of c is set
            const uint64_t bb = __ballot(ev && t != 0);
            m_lo &= ~((uint32_t)bb ^ t);
            m_hi &= ~((uint32_t)(bb >> 32) ^ t);
        };
#pragma unroll
        for (uint32_t b = 0; b < 3; b++) match_bit(b);
        if (__ballot(ev && c >= 8u) != 0) {  // (a smooth frame's batches hold contexts 0 .. 7 only: three bits do)
            match_bit(3);
            match_bit(4);
            if (__ballot(ev && c >= 32u) != 0) {  // contexts are < 256 (gray) / < 512 (Y, Co, Cg)
                constexpr uint32_t CTX_BITS = nctx_of<T>() == 256 ? 8 : 9;
#pragma unroll
                for (uint32_t b = 5; b < CTX_BITS; b++) match_bit(b);
            }
        }
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi(m_hi, __builtin_amdgcn_mbcnt_lo(m_lo, 0u));
        const uint32_t group = (uint32_t)__popc(m_lo) + (uint32_t)__popc(m_hi);
        const bool leader = ev && rank == 0;  // first lane of its context in this batch
        // one LDS read per lane (same context -> same address -> broadcast), one write per leader
        uint32_t dest = 0;
        if (ev) dest = run[c] + rank;
        __builtin_amdgcn_wave_barrier();
        if (leader) run[c] = dest + group;  // the leader has rank 0: dest is the context's running offset
        __builtin_amdgcn_wave_barrier();
        if (ev) {
            sorted_e[dest] = (ET)e;
            if (REL)  // the pack stage knows its tile: two bytes per event instead of four
                reinterpret_cast<uint16_t *>(pix_of)[dest] = (uint16_t)off;
            else
                pix_of[dest] = plane_first + begin + off;
        }
    };
    // Trips go in ROUNDS of PAIR: a round's loads are issued together at the top of the round before it, its events are
    // compacted into the ring trip by trip and ranked / stored at its end.  The stores make the compiler wait for EVERYTHING
    // a wave has in flight wherever it uses a loaded value (gfx9 counts loads and stores in one in-order counter and the
    // number of batches stored is data-dependent), so a load is waited for one "wait interval" after it was issued whatever
    // the depth of the prefetch: with one trip per interval (round 3: three trips of prefetch, one trip per wait) that was
    // less than a memory round trip under load, and every trip stood for the rest of it; a round of two trips is longer
    // than the round trip.
    constexpr uint32_t PAIR = 2;
    static_assert(RING >= PAIR * 256 + 64, "the ring holds a round's events behind a partial batch");
    auto is_interior = [&](uint32_t r, uint32_t x, uint32_t y) { return y > 0 && x + 256 <= W && r + 256 <= end; };  // (a span from the first column included)
    Interior4<T> pre[PAIR];
    bool have[PAIR];
    uint32_t ri = begin, yi = begin / W, xi = begin - yi * W;  // the next trip to issue
    auto issue = [&](Interior4<T> &slot, bool &h) {
        h = ri < end && is_interior(ri, xi, yi);
        if (h) load_interior4(pl, ri, W, span_left_index(ri, xi, yi, W), slot);
        ri += 256;
        xi += 256;
        if (xi >= W) {  // (once per image row: scalar division)
            const uint32_t q = xi / W;
            yi += q;
            xi -= q * W;
        }
    };
    // The events of a trip are compacted into a per-wave ring in LDS, raster order kept, and ranked / stored 64 at a time:
    // every ballot and every store then works on 64 events instead of the ~35 % of a row's lanes that hold one.
    auto trip = [&](const Interior4<T> &now, bool interior, uint32_t row0) {
        if (interior) {
            // lane l takes pixels row0 + 4l .. + 3 (two wide loads instead of twelve byte loads); a prefix
            // sum of the lanes' event counts keeps the ring in raster order
            const uint32_t off0 = row0 - begin + 4 * lane;
            PixelClass pc[4];
            classify_loaded4(now, pc);
            uint32_t nev = 0;
#pragma unroll
            for (uint32_t j = 0; j < 4; j++) nev += pc[j].cls != CLS_IN ? 1u : 0u;
            const uint32_t incl = wave_incl_scan(nev);
            uint32_t pos = qtail + incl - nev;
#pragma unroll
            for (uint32_t j = 0; j < 4; j++) {
                if (pc[j].cls != CLS_IN) {
                    ring[pos & (RING - 1u)] = (pc[j].ctx << 22) | (pc[j].val << 13) | (off0 + j);
                    pos++;
                }
            }
            qtail += readlane(incl, 63);
        } else {
            bool evs[4];
            uint32_t cs[4], es[4];
            Coord xy;
            xy.set(row0 + lane, W);
#pragma unroll
            for (uint32_t u = 0; u < 4; u++) {
                const uint32_t i = row0 + u * 64 + lane;
                evs[u] = false;
                cs[u] = 0;
                es[u] = 0;
                if (i < end && i >= 2) {
                    const PixelClass pc = classify(pl, i, xy.x, xy.y, W);
                    evs[u] = pc.cls != CLS_IN;
                    cs[u] = pc.ctx;
                    es[u] = pc.val;
                }
                xy.advance(64, W);
            }
#pragma unroll
            for (uint32_t u = 0; u < 4; u++) {  // row by row, lane by lane
                const uint64_t m = __ballot(evs[u]);
                if (m == 0) continue;
                if (evs[u]) ring[(qtail + mbcnt(m)) & (RING - 1u)] = (cs[u] << 22) | (es[u] << 13) | (row0 - begin + u * 64 + lane);
                qtail += (uint32_t)__popcll(m);
            }
        }
    };
#pragma unroll
    for (uint32_t d = 0; d < PAIR; d++) issue(pre[d], have[d]);
    for (uint32_t row0 = begin; row0 < end;) {
        Interior4<T> now[PAIR];
        bool inter[PAIR];
#pragma unroll
        for (uint32_t d = 0; d < PAIR; d++) {
            now[d] = pre[d];
            inter[d] = have[d];
        }
#pragma unroll
        for (uint32_t d = 0; d < PAIR; d++) issue(pre[d], have[d]);  // the next round's loads: in flight during this whole round
#pragma unroll
        for (uint32_t d = 0; d < PAIR; d++) {
            if (row0 < end) {
                trip(now[d], inter[d], row0);
                row0 += 256;
            }
        }
        __builtin_amdgcn_wave_barrier();
        while (qtail - qhead >= 64u) {
            drain(64u);
            qhead += 64u;
        }
    }
    if (qtail != qhead) drain(qtail - qhead);
}

// ------------------------------------------------------------------------------------------
// k_scatter: one workgroup per tile, a quarter of the tile per wave.
//
//   1. every wave classifies its 1024 pixels (all of their loads in flight together: there is no store in this kernel before
//      its last step, so nothing makes the compiler wait for more than the load it needs), a trip of 256 at a time: the trip's
//      events are compacted, raster order kept, into a staging buffer in LDS and read back, 64 per batch, into registers;
//   2. it ranks its events within (wave, context), batch by batch as they arrive: ONE returning LDS add on the counter of the
//      event's context ranks the 64 events of a batch (the lanes that name the same address are served in ascending lane order -- measured over 2 x 10^10
//      atomics, profiles/tools/micro/lds_atomic_order.hip, and not documented anywhere, hence step 5's check), against nine
//      ballots and the mask arithmetic around them in k_scatter_ballot; the ranks stay in registers, the counters end as the
//      wave's event count per context;
//   3. thread c turns the four waves' counts of context c into the tile's local layout -- contexts in ascending order, within a
//      context wave 0's events, then wave 1's ... -- i.e. a start per (wave, context), and into the distance between a
//      context's place in that layout and its place in the chain (tile_off + chain_base, as before);
//   4. every wave moves its events to their places (start of its context + rank);
//   5. the workgroup writes the sorted tile out, 256 consecutive events per trip: a context's run is one contiguous piece of
//      its chain, so the 64 lanes of a store touch the two or three cache lines its runs lie in instead of one or two per
//      event context (10-20 lines per instruction on smooth content, 64 on noise: the store path was what this kernel waited
//      for).  Each event is compared with its successor in the sorted tile: (context, pixel offset) must ascend strictly.  That
//      is exactly "stable partition": the set of events of a context is fixed by the counts, and ascending pixel offsets are
//      the one raster order of that set.  A violation raises *order_flag; the host then redoes the batch with k_scatter_ballot.
//
// Record: context << 22 | value << 13 | pixel offset in the tile (9 + 9 + 13 bits).
// ------------------------------------------------------------------------------------------

#ifdef FELICS_SCATTER_STAMPS  // diagnostic build (profiles/tools/scatter_stamps.py): s_memtime of wave 0 between the steps
__device__ unsigned long long g_scatter_stamps[256][16];
#define SSTAMP(i)                                                   \
    do {                                                            \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        st_acc[i] = now_ - st_last;                                 \
        st_last = now_;                                             \
    } while (0)
extern "C" __attribute__((visibility("default"))) int felics_debug_scatter_stamps(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_scatter_stamps), sizeof(g_scatter_stamps)) != hipSuccess) return -1;
    if (reset) {
        static unsigned long long z[256 * 16] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_scatter_stamps), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#else
#define SSTAMP(i)
#endif

// (six workgroups per CU is what the LDS allows -- five for Y / Co / Cg planes -- and the registers are held to that)
template <typename T, typename ET, bool REL>
__attribute__((amdgpu_waves_per_eu(sizeof(T) == 1 ? 6 : 5))) __global__ __launch_bounds__(256) void k_scatter(const T *__restrict__ planes, const uint32_t *__restrict__ tile_off,
                                                 const uint32_t *__restrict__ chain_base, ET *__restrict__ sorted_e,
                                                 uint32_t *__restrict__ pix_of, uint32_t W, uint32_t npix, uint32_t ntiles,
                                                 uint32_t tile_begin, uint32_t tile_end, uint32_t nplanes,
                                                 uint32_t *__restrict__ order_flag, uint32_t test_violation) {
    constexpr uint32_t NC = nctx_of<T>();
    constexpr uint32_t QUARTER = SORT_TILE / 4, TRIPS = QUARTER / 256;
    constexpr uint32_t PER = NC / 256;  // contexts per thread in step 3
    constexpr uint32_t KEY = 0xFFC01FFFu;  // context and pixel offset of a record
    static_assert(SORT_TILE % 1024 == 0 && SORT_TILE <= (1u << 13), "four whole trips per wave; 13 bits of pixel offset");
    static_assert(NC % 256 == 0 && NC <= 512, "a thread takes NC / 256 contexts; 9 bits of context");
    __shared__ uint32_t srt[SORT_TILE + 1];   // the tile's events in chain order (+ a sentinel behind the last)
    __shared__ uint32_t stages[4][256];       // per wave: the events of one trip in raster order, on their way into registers
    __shared__ uint32_t cnt[4][NC];           // per wave and context: count, then cursor into srt
    __shared__ uint32_t gdst[NC];             // chain slot of a context's run minus the run's place in srt
    __shared__ uint32_t wsum[4];
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = lane_id();  // (uniform: see k_hist)
    const uint32_t tid = threadIdx.x;
#ifdef FELICS_SCATTER_STAMPS
    unsigned long long st_last = __builtin_amdgcn_s_memtime(), st_acc[10] = {};
#endif
    // Workgroup -> (plane, tile), XCD-aware: workgroups are dealt round-robin over the eight XCDs (MI355X_MICROARCH.md,
    // workgroup dispatch: blocks b and b + 8 share one), and every XCD has an L2 of its own that does not merge its partial
    // lines with another XCD's.  Neighbouring tiles of a plane append to the same cache lines of every chain, so all tiles of
    // plane p go to the XCD p % 8 (in tile order: workgroup b = 8 i + x takes item i of XCD x's list of planes x, x + 8, ...) --
    // the XCD whose spine and pack workgroups read the chains of plane p later (their grids are plane-minor with 64 planes).
    // Placement only: nothing depends on it for correctness.
    const uint32_t wg_tiles = tile_end - tile_begin;
    const uint32_t item = blockIdx.x >> 3, xcd = blockIdx.x & 7u;
    const uint32_t plane = xcd + 8u * (item / wg_tiles);
    if (plane >= nplanes) return;  // (the whole workgroup)
    const uint32_t tile = tile_begin + item % wg_tiles;
    // where the chains continue for this tile: needed in step 3, asked for now
    uint32_t runpos[PER];
    {
        const uint32_t *off = tile_off + ((uint64_t)plane * ntiles + tile) * NC + tid * PER;
        const uint32_t *cb = chain_base + (uint64_t)plane * NC + tid * PER;
#pragma unroll
        for (uint32_t u = 0; u < PER; u++) runpos[u] = off[u] + cb[u];
    }
    uint32_t *my_cnt = cnt[wave];
    for (uint32_t c = lane; c < NC; c += 64) my_cnt[c] = 0;
    const T *pl = planes + (uint64_t)plane * npix;
    const uint32_t plane_first = plane * npix;
    const uint32_t begin = tile * SORT_TILE;
    const uint32_t qbegin = min(begin + wave * QUARTER, npix);
    const uint32_t end = min(qbegin + QUARTER, npix);  // of this wave's quarter
    uint32_t *stage = stages[wave];
    // ---- 1. classify, compact, rank.  The events of a trip are compacted, raster order kept, into the wave's staging buffer and
    // read back 64 at a time into registers: event 64 u + lane of trip d lives in slot (d, u) of this lane -- static slots under
    // wave-uniform guards -- together with its rank within (wave, context), which ONE returning LDS add per 64 events hands out
    // (the counter of a context ends as the wave's number of events in it).  The batches of a trip go in pairs: their LDS reads
    // are in flight together, and so are their atomics.
    constexpr uint32_t BPT = 4, SLOTS = TRIPS * BPT;  // up to 256 events per trip
    uint32_t rec[SLOTS], rk[SLOTS], nd[TRIPS];
    auto is_interior = [&](uint32_t r, uint32_t x, uint32_t y) { return y > 0 && x + 256 <= W && r + 256 <= end; };  // (a span from the first column included)
    Interior4<T> pre[TRIPS];
    bool have[TRIPS];
    {
        uint32_t ri = qbegin, yi = qbegin / W, xi = qbegin - yi * W;
#pragma unroll
        for (uint32_t d = 0; d < TRIPS; d++) {
            have[d] = ri < end && is_interior(ri, xi, yi);
            if (have[d]) load_interior4(pl, ri, W, span_left_index(ri, xi, yi, W), pre[d]);
            ri += 256;
            xi += 256;
            if (xi >= W) {  // (once per image row: scalar division)
                const uint32_t q = xi / W;
                yi += q;
                xi -= q * W;
            }
        }
    }
    SSTAMP(0);
#pragma unroll
    for (uint32_t d = 0; d < TRIPS; d++) {
        const uint32_t row0 = qbegin + d * 256;
        {   // (a trip past the quarter's end takes the general path with every lane switched off: no guard around the trip, so
            // that the slots are plain assignments and not values merged across a branch -- those cost a register copy each)
            uint32_t n = 0;  // events of this trip (wave-uniform)
            if (have[d]) {
                // lane l takes pixels row0 + 4l .. + 3 (two wide loads instead of twelve byte loads); a prefix
                // sum of the lanes' event counts keeps the raster order
                const uint32_t off0 = row0 - begin + 4 * lane;
                PixelClass pc[4];
                classify_loaded4(pre[d], pc);
                uint32_t nev = 0;
#pragma unroll
                for (uint32_t j = 0; j < 4; j++) nev += pc[j].cls != CLS_IN ? 1u : 0u;
                const uint32_t incl = wave_incl_scan(nev);
                uint32_t pos = incl - nev;
#pragma unroll
                for (uint32_t j = 0; j < 4; j++) {
                    if (pc[j].cls != CLS_IN) {
                        stage[pos] = (pc[j].ctx << 22) | (pc[j].val << 13) | (off0 + j);
                        pos++;
                    }
                }
                n = readlane(incl, 63);
            } else {  // (a trip that crosses a row end, lies in the first row or ends the plane: the general neighbour rule)
                bool evs[4];
                uint32_t cs[4], es[4];
                Coord xy;
                xy.set(row0 + lane, W);
#pragma unroll
                for (uint32_t u = 0; u < 4; u++) {
                    const uint32_t i = row0 + u * 64 + lane;
                    evs[u] = false;
                    cs[u] = 0;
                    es[u] = 0;
                    if (i < end && i >= 2) {
                        const PixelClass pc = classify(pl, i, xy.x, xy.y, W);
                        evs[u] = pc.cls != CLS_IN;
                        cs[u] = pc.ctx;
                        es[u] = pc.val;
                    }
                    xy.advance(64, W);
                }
#pragma unroll
                for (uint32_t u = 0; u < 4; u++) {  // row by row, lane by lane
                    const uint64_t m = __ballot(evs[u]);
                    if (m == 0) continue;
                    if (evs[u]) stage[n + mbcnt(m)] = (cs[u] << 22) | (es[u] << 13) | (row0 - begin + u * 64 + lane);
                    n += (uint32_t)__popcll(m);
                }
            }
            nd[d] = n;
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (uint32_t u = 0; u < BPT; u++) rec[d * BPT + u] = stage[u * 64 + lane];  // (past n: whatever the buffer held, not used)
#pragma unroll
            for (uint32_t u = 0; u < BPT; u++) {
                uint32_t r = 0;
                if (u * 64 + lane < n) r = atomicAdd(&my_cnt[rec[d * BPT + u] >> 22], 1u);
                rk[d * BPT + u] = r;
            }
            __builtin_amdgcn_wave_barrier();  // (the next trip writes the staging buffer again)
        }
    }
    SSTAMP(1);
    SSTAMP(2);
    __syncthreads();
    SSTAMP(3);
    // ---- 3. the tile's layout: thread t takes contexts t * PER ..
    {
        uint32_t n[4][PER], tot = 0;
#pragma unroll
        for (uint32_t u = 0; u < PER; u++) {
#pragma unroll
            for (uint32_t w = 0; w < 4; w++) {
                n[w][u] = cnt[w][tid * PER + u];
                tot += n[w][u];
            }
        }
        const uint32_t incl = wave_incl_scan(tot);
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint32_t at = incl - tot;  // events of the tile in contexts below this thread's
        for (uint32_t w = 0; w < wave; w++) at += wsum[w];
#pragma unroll
        for (uint32_t u = 0; u < PER; u++) {
            const uint32_t c = tid * PER + u;
            gdst[c] = runpos[u] - at;
#pragma unroll
            for (uint32_t w = 0; w < 4; w++) {
                cnt[w][c] = at;
                at += n[w][u];
            }
        }
    }
    const uint32_t total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    if (tid == 0) srt[total] = 0xFFFFFFFFu;  // larger than any record's key: the last event has a successor to be compared with
    __syncthreads();
    SSTAMP(4);
    // ---- 4. place: where the context's events of this wave start + the event's rank among them.  (Lane masks only, no guards: the
    // LDS reads of eight slots are in flight together, then their writes -- under a guard per pair of slots this step was eight
    // LDS round trips one after the other.)
#pragma unroll
    for (uint32_t q0 = 0; q0 < SLOTS; q0 += 8) {  // (eight slots at a time: sixteen would cost the registers of a sixth workgroup per CU)
        uint32_t at[8];
#pragma unroll
        for (uint32_t q = 0; q < 8; q++) at[q] = my_cnt[(rec[q0 + q] >> 22) & (NC - 1u)];  // (masked: an unused slot holds anything)
#pragma unroll
        for (uint32_t q = 0; q < 8; q++)
            if (((q0 + q) % BPT) * 64 + lane < nd[(q0 + q) / BPT]) srt[at[q] + rk[q0 + q]] = rec[q0 + q];
    }
    SSTAMP(5);
    __syncthreads();
    SSTAMP(6);
    // ---- 5. out, checked
    uint32_t bad = test_violation;
    for (uint32_t j = tid; j < total; j += 256) {
        const uint32_t rec = srt[j], nxt = srt[j + 1];
        const uint32_t dst = gdst[rec >> 22] + j;
        sorted_e[dst] = (ET)((rec >> 13) & 0x1FFu);
        if (REL)  // the pack stage knows its tile: two bytes per event instead of four
            reinterpret_cast<uint16_t *>(pix_of)[dst] = (uint16_t)(rec & 0x1FFFu);
        else
            pix_of[dst] = plane_first + begin + (rec & 0x1FFFu);
        bad |= (nxt & KEY) <= (rec & KEY) ? 1u : 0u;
    }
    if (__ballot(bad != 0) != 0 && lane == 0) atomicOr(order_flag, 1u);
#ifdef FELICS_SCATTER_STAMPS
    SSTAMP(7);
    if (tid == 0) {
        unsigned long long *slot = g_scatter_stamps[(tile * 7u + plane) & 255u];
        for (int i = 0; i < 8; i++) atomicAdd(&slot[i], st_acc[i]);
        atomicAdd(&slot[15], 1ull);
    }
#endif
}

// ------------------------------------------------------------------------------------------
// k_front (round 5): the ONE classification of a pixel.  k_scatter's steps 1-4 as they are (classify, compact, rank with a
// returning LDS add, layout, place into the tile's sorted order in LDS); what changes is where the sorted tile goes: not to
// the chains (which needed every tile's counts first: a histogram pass and a scan over the tiles) but to the tile's own
// place, slot (plane * ntiles + tile) * cap + s -- contexts ascending, every context's run starting on a multiple of REC
// slots (felics_kernels.h: tile-local layout) -- with the run table {first record, events} per context and the slots in use.
// The chain of a context is then the sequence of its runs over the tiles (felics_chain.hip).
//   mode & FRONT_SAFE_RANK: ranks from ballots instead of the returning add (the context's fallback once the order check
//   of step 5 has failed: no assumption about the LDS); mode & FRONT_TEST_VIOLATION: report a violation whatever the order.
// ------------------------------------------------------------------------------------------
template <typename T, typename ET>
__attribute__((amdgpu_waves_per_eu(sizeof(T) == 1 ? 6 : 5))) __global__ __launch_bounds__(256) void k_front(
    const T *__restrict__ planes, ET *__restrict__ ev, uint16_t *__restrict__ pix, uint32_t *__restrict__ runtab,
    uint32_t *__restrict__ tile_slots, uint32_t W, uint32_t npix, uint32_t ntiles, uint32_t tile_begin, uint32_t tile_end,
    uint32_t nplanes, uint32_t cap, uint32_t *__restrict__ flags, uint32_t mode) {
    constexpr uint32_t NC = nctx_of<T>();
    constexpr uint32_t QUARTER = SORT_TILE / 4, TRIPS = QUARTER / 256;
    constexpr uint32_t PER = NC / 256;  // contexts per thread in step 3
    constexpr uint32_t KEY = 0xFFC01FFFu;  // context and pixel offset of a record
    static_assert(SORT_TILE % 1024 == 0 && SORT_TILE <= (1u << 13), "four whole trips per wave; 13 bits of pixel offset");
    static_assert(NC % 256 == 0 && NC <= 512, "a thread takes NC / 256 contexts; 9 bits of context");
    __shared__ uint32_t srt[SORT_TILE + 1];   // the tile's events, contexts ascending, raster order inside (+ a sentinel behind the last)
    __shared__ uint32_t stages[4][256];       // per wave: the events of one trip in raster order, on their way into registers
    __shared__ uint32_t cnt[4][NC];           // per wave and context: count, then cursor into srt
    __shared__ uint32_t gdst[NC];             // a context's run: its place among the tile's slots minus its place in srt
    __shared__ uint32_t wsum[4];
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = lane_id();  // (uniform: see k_hist)
    const uint32_t tid = threadIdx.x;
    // Workgroup -> (plane, tile), XCD-aware (as k_scatter): all tiles of plane p on the XCD p % 8, in tile order -- the XCD whose
    // k_enum, spine and pack workgroups read plane p's tiles later.  Placement only.
    const uint32_t wg_tiles = tile_end - tile_begin;
    const uint32_t item = blockIdx.x >> 3, xcd = blockIdx.x & 7u;
    const uint32_t plane = xcd + 8u * (item / wg_tiles);
    if (plane >= nplanes) return;  // (the whole workgroup)
    const uint32_t tile = tile_begin + item % wg_tiles;
    uint32_t *my_cnt = cnt[wave];
    for (uint32_t c = lane; c < NC; c += 64) my_cnt[c] = 0;
    const T *pl = planes + (uint64_t)plane * npix;
    const uint32_t begin = tile * SORT_TILE;
    const uint32_t qbegin = min(begin + wave * QUARTER, npix);
    const uint32_t end = min(qbegin + QUARTER, npix);  // of this wave's quarter
    uint32_t *stage = stages[wave];
    const bool safe_rank = (mode & FRONT_SAFE_RANK) != 0;
    // ---- 1. classify, compact, rank (k_scatter's step: static register slots, one returning LDS add per 64 events)
    constexpr uint32_t BPT = 4, SLOTS = TRIPS * BPT;  // up to 256 events per trip
    uint32_t rec[SLOTS], rk[SLOTS], nd[TRIPS];
    auto is_interior = [&](uint32_t r, uint32_t x, uint32_t y) { return y > 0 && x + 256 <= W && r + 256 <= end; };  // (a span from the first column included)
    Interior4<T> pre[TRIPS];
    bool have[TRIPS];
    {
        uint32_t ri = qbegin, yi = qbegin / W, xi = qbegin - yi * W;
#pragma unroll
        for (uint32_t d = 0; d < TRIPS; d++) {
            have[d] = ri < end && is_interior(ri, xi, yi);
            if (have[d]) load_interior4(pl, ri, W, span_left_index(ri, xi, yi, W), pre[d]);
            ri += 256;
            xi += 256;
            if (xi >= W) {  // (once per image row: scalar division)
                const uint32_t q = xi / W;
                yi += q;
                xi -= q * W;
            }
        }
    }
#pragma unroll
    for (uint32_t d = 0; d < TRIPS; d++) {
        const uint32_t row0 = qbegin + d * 256;
        uint32_t n = 0;  // events of this trip (wave-uniform)
        if (have[d]) {
            const uint32_t off0 = row0 - begin + 4 * lane;
            PixelClass pc[4];
            classify_loaded4(pre[d], pc);
            uint32_t nev = 0;
#pragma unroll
            for (uint32_t j = 0; j < 4; j++) nev += pc[j].cls != CLS_IN ? 1u : 0u;
            const uint32_t incl = wave_incl_scan(nev);
            uint32_t pos = incl - nev;
#pragma unroll
            for (uint32_t j = 0; j < 4; j++) {
                if (pc[j].cls != CLS_IN) {
                    stage[pos] = (pc[j].ctx << 22) | (pc[j].val << 13) | (off0 + j);
                    pos++;
                }
            }
            n = readlane(incl, 63);
        } else {  // (a trip that crosses a row end, lies in the first row or ends the plane: the general neighbour rule)
            bool evs[4];
            uint32_t cs[4], es[4];
            Coord xy;
            xy.set(row0 + lane, W);
#pragma unroll
            for (uint32_t u = 0; u < 4; u++) {
                const uint32_t i = row0 + u * 64 + lane;
                evs[u] = false;
                cs[u] = 0;
                es[u] = 0;
                if (i < end && i >= 2) {
                    const PixelClass pc = classify(pl, i, xy.x, xy.y, W);
                    evs[u] = pc.cls != CLS_IN;
                    cs[u] = pc.ctx;
                    es[u] = pc.val;
                }
                xy.advance(64, W);
            }
#pragma unroll
            for (uint32_t u = 0; u < 4; u++) {  // row by row, lane by lane
                const uint64_t m = __ballot(evs[u]);
                if (m == 0) continue;
                if (evs[u]) stage[n + mbcnt(m)] = (cs[u] << 22) | (es[u] << 13) | (row0 - begin + u * 64 + lane);
                n += (uint32_t)__popcll(m);
            }
        }
        nd[d] = n;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (uint32_t u = 0; u < BPT; u++) rec[d * BPT + u] = stage[u * 64 + lane];  // (past n: whatever the buffer held, not used)
        if (!safe_rank) {
#pragma unroll
            for (uint32_t u = 0; u < BPT; u++) {
                uint32_t r = 0;
                if (u * 64 + lane < n) r = atomicAdd(&my_cnt[rec[d * BPT + u] >> 22], 1u);
                rk[d * BPT + u] = r;
            }
        } else {
            // ranks from ballots (k_scatter_ballot's way): every lane learns the lanes that hold its context, one ballot per
            // context bit; its rank = the context's count so far + the lanes in front of it, the first lane of a context adds
            // the batch's share to the count
#pragma unroll
            for (uint32_t u = 0; u < BPT; u++) {
                const bool e = u * 64 + lane < n;
                const uint32_t c = (rec[d * BPT + u] >> 22) & (NC - 1u);
                const uint64_t ev_mask = __ballot(e);
                uint32_t m_lo = (uint32_t)ev_mask, m_hi = (uint32_t)(ev_mask >> 32);
                constexpr uint32_t CTX_BITS = NC == 256 ? 8 : 9;
#pragma unroll
                for (uint32_t b = 0; b < CTX_BITS; b++) {
                    const uint32_t t = (uint32_t)((int32_t)(c << (31 - b)) >> 31);  // all ones if bit b of c is set
                    const uint64_t bb = __ballot(e && t != 0);
                    m_lo &= ~((uint32_t)bb ^ t);
                    m_hi &= ~((uint32_t)(bb >> 32) ^ t);
                }
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi(m_hi, __builtin_amdgcn_mbcnt_lo(m_lo, 0u));
                const uint32_t group = (uint32_t)__popc(m_lo) + (uint32_t)__popc(m_hi);
                uint32_t r = 0;
                if (e) r = my_cnt[c] + rank;
                __builtin_amdgcn_wave_barrier();
                if (e && rank == 0) my_cnt[c] = r + group;
                __builtin_amdgcn_wave_barrier();
                rk[d * BPT + u] = r;
            }
        }
        __builtin_amdgcn_wave_barrier();  // (the next trip writes the staging buffer again)
    }
    __syncthreads();
    // ---- 3. the tile's layout: thread t takes contexts t * PER ..; two running sums in one register: the events in front (low
    // half: the context's place in srt) and the slots in front (high half: every run rounded up to whole records)
    const uint64_t pt = (uint64_t)plane * ntiles + tile;
    uint32_t padpos[PER], nev_c[PER];
    {
        uint32_t n[4][PER], tot = 0;
#pragma unroll
        for (uint32_t u = 0; u < PER; u++) {
            nev_c[u] = 0;
#pragma unroll
            for (uint32_t w = 0; w < 4; w++) {
                n[w][u] = cnt[w][tid * PER + u];
                nev_c[u] += n[w][u];
            }
            tot += nev_c[u] | (((nev_c[u] + REC - 1u) & ~(REC - 1u)) << 16);
        }
        const uint32_t incl = wave_incl_scan(tot);  // (no carry between the halves: at most 4096 events)
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint32_t at = incl - tot;
        for (uint32_t w = 0; w < wave; w++) at += wsum[w];
#pragma unroll
        for (uint32_t u = 0; u < PER; u++) {
            const uint32_t c = tid * PER + u;
            uint32_t cs = at & 0xFFFFu;
            const uint32_t ps = at >> 16;
            padpos[u] = ps;
            gdst[c] = ps - cs;
#pragma unroll
            for (uint32_t w = 0; w < 4; w++) {
                cnt[w][c] = cs;
                cs += n[w][u];
            }
            at += nev_c[u] | (((nev_c[u] + REC - 1u) & ~(REC - 1u)) << 16);
        }
    }
    const uint32_t both = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    const uint32_t total = both & 0xFFFFu, slots = both >> 16;
    if (tid == 0) srt[total] = 0xFFFFFFFFu;  // larger than any record's key: the last event has a successor to be compared with
    const bool fits = slots <= cap;  // (if not: nothing of the tile is written, the host redoes the batch with the worst-case cap)
    if (fits) {
        // the run table (coalesced: a thread's PER contexts are neighbours) and the padding slots of this thread's runs
        uint32_t *rt = runtab + pt * NC + tid * PER;
#pragma unroll
        for (uint32_t u = 0; u < PER; u++) rt[u] = (padpos[u] / REC) | (nev_c[u] << 16);
        uint16_t *px = pix + pt * cap;
#pragma unroll
        for (uint32_t u = 0; u < PER; u++)
            for (uint32_t i = nev_c[u]; i < ((nev_c[u] + REC - 1u) & ~(REC - 1u)); i++) px[padpos[u] + i] = 0xFFFFu;
        if (tid == 0) tile_slots[pt] = slots;
    } else {  // (an empty run table: the chain stage finds nothing of this tile)
        uint32_t *rt = runtab + pt * NC + tid * PER;
#pragma unroll
        for (uint32_t u = 0; u < PER; u++) rt[u] = 0;
        if (tid == 0) {
            tile_slots[pt] = 0;
            atomicOr(flags, TL_FLAG_OVERFLOW);
        }
    }
    __syncthreads();
    // ---- 4. place: where the context's events of this wave start + the event's rank among them
#pragma unroll
    for (uint32_t q0 = 0; q0 < SLOTS; q0 += 8) {
        uint32_t at[8];
#pragma unroll
        for (uint32_t q = 0; q < 8; q++) at[q] = my_cnt[(rec[q0 + q] >> 22) & (NC - 1u)];  // (masked: an unused slot holds anything)
#pragma unroll
        for (uint32_t q = 0; q < 8; q++)
            if (((q0 + q) % BPT) * 64 + lane < nd[(q0 + q) / BPT]) srt[at[q] + rk[q0 + q]] = rec[q0 + q];
    }
    __syncthreads();
    // ---- 5. out, checked: (context, pixel offset) must ascend strictly -- exactly "stable partition by context"
    uint32_t bad = mode & FRONT_TEST_VIOLATION;
    if (fits) {
        ET *evo = ev + pt * cap;
        uint16_t *pxo = pix + pt * cap;
        for (uint32_t j = tid; j < total; j += 256) {
            const uint32_t r = srt[j], nxt = srt[j + 1];
            const uint32_t dst = gdst[r >> 22] + j;
            evo[dst] = (ET)((r >> 13) & 0x1FFu);
            pxo[dst] = (uint16_t)(r & 0x1FFFu);
            bad |= (nxt & KEY) <= (r & KEY) ? 1u : 0u;
        }
    }
    if (__ballot(bad != 0) != 0 && lane == 0) atomicOr(flags, TL_FLAG_ORDER);
}

// ------------------------------------------------------------------------------------------
// resolve: replay KEstimator (parameter_selection.rs:49-85) along every chain.
//
// State S[k] = accumulated Rice lengths for k = 0..5 (traits.rs:26).  While no halving happens the
// state seen by event t is S + P_excl(t), P = prefix sums of the six length vectors.
// `min(S + P_incl(t)) > 1024` (parameter_selection.rs:58-63) is monotone in t because lengths are
// positive, so inside a run of events the first t where it holds is the next halving:
// S <- (S + P_incl(t)) >> 1, and the following events continue from there.
// get_k ties go to the LARGEST k (`<=` at parameter_selection.rs:79).
//
// Two kernels.  k_spine walks one chain per wave and only finds the state at the start of every
// 64-event block: the one sequential dependency of the codec, so it is written for latency --
// 64 blocks are fetched at a time, each lane sums one block, and a block whose end state still has
// a counter <= 1024 is stepped over with one vector add; only a block that contains a halving is searched.
// k_assign then gives every event its k, one wave per block, all blocks in parallel.
// ------------------------------------------------------------------------------------------

constexpr uint32_t SPINE_BATCH = 64;  // blocks fetched per step: lane j holds block j

// LDS of the one-wave walk (k_spine, and the short chains of k_spine2)
template <typename ET>
struct SpineSingleLDS {
    static constexpr uint32_t DW = 64 * sizeof(ET) / 4;  // dwords per block
    uint32_t stage[SPINE_BATCH * DW];       // the batch's events
    uint32_t bsum[(SPINE_BATCH + 1) * 8];   // [block][k]: sum of the block's lengths for k = 0..5
    uint32_t rec[SPINE_BATCH * 8];          // [block][k]: state at the start of the block
};

template <typename ET>
__device__ __forceinline__ void spine_single(SpineSingleLDS<ET> &sh, const ET *__restrict__ sorted_e, uint32_t *__restrict__ block_state,
                                              const uint32_t *__restrict__ chain_base,
                                              const uint32_t *__restrict__ chain_len, uint32_t nchains,
                                              const uint32_t *__restrict__ tile_off, uint32_t ntiles, uint32_t t_end,
                                              uint32_t *__restrict__ chain_prog, uint32_t *__restrict__ block_tag,
                                              uint2 *__restrict__ partial, uint32_t stamp) {
    constexpr uint32_t DW = 64 * sizeof(ET) / 4;  // dwords per block
    uint32_t (&stage)[SPINE_BATCH * DW] = sh.stage;
    uint32_t (&bsum)[(SPINE_BATCH + 1) * 8] = sh.bsum;
    uint32_t (&rec)[SPINE_BATCH * 8] = sh.rec;
    // Workgroup w -> (context w / nplanes, plane w % nplanes): the long chains (small contexts) of all
    // planes start first and land on different XCDs (workgroups are dealt round-robin over the XCDs).
    if (blockIdx.x >= nchains) return;
    constexpr uint32_t NC = nctx_of<ET>();
    const uint32_t nplanes = nchains / NC;
    const uint32_t ctx = blockIdx.x / nplanes, plane = blockIdx.x % nplanes;
    const uint32_t chain = plane * NC + ctx;
    const uint32_t n = chain_len[chain];
    if (n == 0) return;
    const uint32_t lane = lane_id();
    const uint32_t l7 = lane & 7u;
    // The kernel is launched once per slice of tiles, as soon as that slice's events have been
    // scattered: it resumes every chain at chain_prog and stops at the last whole block whose events
    // all come from tiles < t_end (the final launch, t_end = ntiles, also takes the partial block).
    const bool final_slice = t_end >= ntiles;
    const uint32_t avail = final_slice ? n : tile_off[((uint64_t)plane * ntiles + t_end) * NC + ctx];  // events in place
    const uint32_t nblocks = final_slice ? (n + 63u) >> 6 : avail >> 6;
    uint32_t *prog = chain_prog + (uint64_t)chain * 8;  // [0] next block, [1..6] state
    const uint32_t first_block = prog[0];
    const uint32_t base = chain_base[chain];  // multiple of 64
    const uint4 *src = reinterpret_cast<const uint4 *>(sorted_e + base);  // block b = DW/4 uint4 at b*DW/4
    uint4 *states = reinterpret_cast<uint4 *>(block_state) + (uint64_t)(base >> 6) * 2;
    uint32_t *tags = block_tag + (base >> 6);  // per block: (epoch, slice) of the launch that resolved it
    uint32_t Sv = l7 < 6 ? prog[1 + l7] : 0u;
    // The state must have landed before the walk starts: a load still pending on entry makes the compiler
    // wait for *all* memory operations inside the walk loop, i.e. for the next batch's prefetch as well.
    asm volatile("; state in %0" : "+v"(Sv));
    if (first_block < nblocks) {
    __builtin_amdgcn_s_setprio(3);  // a chain is one long dependent instruction stream: never make it wait for issue

    uint4 buf[DW / 4];
    if (first_block + lane < nblocks) {
#pragma unroll
        for (uint32_t q = 0; q < DW / 4; q++) buf[q] = src[(uint64_t)(first_block + lane) * (DW / 4) + q];
    }
    if (lane < 8) bsum[SPINE_BATCH * 8 + lane] = 0;  // read (and ignored) by the look-ahead of the last block
    // The state lives in a VGPR: lane l holds S[l & 7] (entries 6, 7 unused).  Stepping over a block
    // without a halving is then one LDS read, one add and one compare for all six counters.
    for (uint32_t bb = first_block; bb < nblocks; bb += SPINE_BATCH) {
        const uint32_t nb = min(SPINE_BATCH, nblocks - bb);
        // lane j: sums of block bb + j, constant part 64 * (1 + k) included; events copied to LDS
        if (lane < nb) {
            uint32_t B01 = 64u * (1u | (2u << 16)), B23 = 64u * (3u | (4u << 16)), B45 = 64u * (5u | (6u << 16));
#pragma unroll
            for (uint32_t q = 0; q < DW / 4; q++) {
                add_block_sums<ET>(buf[q].x, B01, B23, B45);
                add_block_sums<ET>(buf[q].y, B01, B23, B45);
                add_block_sums<ET>(buf[q].z, B01, B23, B45);
                add_block_sums<ET>(buf[q].w, B01, B23, B45);
                reinterpret_cast<uint4 *>(stage)[lane * (DW / 4) + q] = buf[q];
            }
            reinterpret_cast<uint4 *>(bsum)[lane * 2] = make_uint4(B01 & 0xFFFFu, B01 >> 16, B23 & 0xFFFFu, B23 >> 16);
            reinterpret_cast<uint4 *>(bsum)[lane * 2 + 1] = make_uint4(B45 & 0xFFFFu, B45 >> 16, 0u, 0u);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();  // (one wave: its own LDS traffic in order)
        // prefetch the next batch while this one is walked
        if (bb + SPINE_BATCH + lane < nblocks) {
#pragma unroll
            for (uint32_t q = 0; q < DW / 4; q++) buf[q] = src[(uint64_t)(bb + SPINE_BATCH + lane) * (DW / 4) + q];
        }
        uint32_t Bv = bsum[l7];
        for (uint32_t j = 0; j < nb; j++) {
            const uint32_t Bnext = bsum[(j + 1) * 8 + l7];  // look-ahead: independent of the state
            rec[j * 8 + l7] = Sv;                            // lanes l and l + 8 store the same value
            const uint32_t Ev = Sv + Bv;
            const uint32_t over = (uint32_t)__ballot(Ev > 1024u) & 0x3Fu;
            if (over != 0x3Fu) {  // some counter still <= 1024 at the end of the block: no halving inside
                Sv = Ev;
                Bv = Bnext;
                continue;
            }
            // a halving happens inside this block: find it with the block's per-event prefix sums
            const uint32_t e = (uint32_t) reinterpret_cast<const ET *>(stage)[j * 64 + lane];
            uint32_t l01, l23, l45;
            packed_lengths(e, l01, l23, l45);
            const uint32_t p01 = wave_incl_scan(l01), p23 = wave_incl_scan(l23), p45 = wave_incl_scan(l45);
            const uint32_t P0 = p01 & 0xFFFFu, P1 = p01 >> 16, P2 = p23 & 0xFFFFu, P3 = p23 >> 16;
            const uint32_t P4 = p45 & 0xFFFFu, P5 = p45 >> 16;
            uint32_t lo = 0;
            while (true) {
                uint32_t S0 = readlane(Sv, 0), S1 = readlane(Sv, 1), S2 = readlane(Sv, 2);
                uint32_t S3 = readlane(Sv, 3), S4 = readlane(Sv, 4), S5 = readlane(Sv, 5);
                const uint32_t mn = min(min(min(S0 + P0, S1 + P1), min(S2 + P2, S3 + P3)), min(S4 + P4, S5 + P5));
                const uint64_t hm = __ballot(lane >= lo && mn > 1024u);
                // (the block-level test said a halving exists, and after a halving the loop is only
                // re-entered when the end state says there is another: hm is never empty here -- but a wave
                // that spins forever on a broken invariant takes the whole GPU with it, so it is checked)
                if (hm == 0) break;
                const uint32_t f = (uint32_t)__ffsll((long long)hm) - 1u;
                // S <- ((S + P(f)) >> 1) - P(f): later lanes add their own P(t) >= P(f) back (mod 2^32).
                // Done on the state vector itself: lane l picks P_{l & 7}(f) out of the three packed scans.
                const uint32_t q01 = readlane(p01, f), q23 = readlane(p23, f), q45 = readlane(p45, f);
                const uint32_t qv = l7 < 2 ? q01 : l7 < 4 ? q23 : l7 < 6 ? q45 : 0u;
                const uint32_t Pf = (qv >> ((l7 & 1u) << 4)) & 0xFFFFu;
                Sv = ((Sv + Pf) >> 1) - Pf;
                lo = f + 1;
                // state at the end of the block if nothing else happens (block sums = prefix sums at
                // lane 63); another round only if that still has all six counters above 1024
                if (lo >= 64 || ((uint32_t)__ballot(Sv + Bv > 1024u) & 0x3Fu) != 0x3Fu) break;
            }
            Sv += Bv;
            Bv = Bnext;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();  // (one wave: its own LDS traffic in order)
        if (lane < nb) {
            const uint4 hi = reinterpret_cast<const uint4 *>(rec)[lane * 2 + 1];
            states[(uint64_t)(bb + lane) * 2] = reinterpret_cast<const uint4 *>(rec)[lane * 2];
            states[(uint64_t)(bb + lane) * 2 + 1] = make_uint4(hi.x, hi.y, 0u, 0u);
            tags[bb + lane] = stamp;  // resolved in this launch (a block is resolved exactly once)
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();  // (one wave: its own LDS traffic in order)
    }
    if (lane < 6) prog[1 + lane] = Sv;
    if (lane == 6) prog[0] = nblocks;
    }
    // Events that are in place but do not fill a block yet: publish the block's start state and, in this
    // launch's own list (never overwritten by a later launch), which block it is and how many of its
    // events exist, so k_assign can serve them now.  The block itself is resolved by a later launch.
    if (!final_slice && (avail & 63u) != 0) {
        const uint32_t s0 = readlane(Sv, 0), s1 = readlane(Sv, 1), s2 = readlane(Sv, 2);
        const uint32_t s3 = readlane(Sv, 3), s4 = readlane(Sv, 4), s5 = readlane(Sv, 5);
        if (lane == 0) {
            states[(uint64_t)nblocks * 2] = make_uint4(s0, s1, s2, s3);
            states[(uint64_t)nblocks * 2 + 1] = make_uint4(s4, s5, 0u, 0u);
            partial[chain] = make_uint2((base >> 6) + nblocks, avail & 63u);
        }
    }
}


// ------------------------------------------------------------------------------------------
// k_spine2: the same walk with HELPER waves.  In k_spine the walker spends two thirds of a halving period building what
// it needs to locate the halving: the block's per-event prefix sums (lengths, three packed DPP scans) and, once per batch,
// the block sums.  None of that depends on the estimator's state, so three more waves of the workgroup produce it ahead of
// the walker -- for EVERY block, although only one block in three holds a halving: they have nothing else to do -- and
// hand it over through LDS, SP_BATCH blocks at a time, double-buffered, one workgroup barrier per batch:
//   helpers, batch t:  pref[t & 1][j][lane] = packed inclusive prefix sums of block j's six length vectors,
//                      bsum[t & 1][j][k]    = the block's sums; and the block-start states the walker left for batch t - 2
//                      (rec) go out to block_state / block_tag;
//   walker,  batch t - 1: per block one LDS read, one add, one compare (as before); in a block with a halving three LDS
//                      reads of its prefix sums and a search that compares them, still packed, with packed thresholds
//                      theta_k = P_k(last halving) + max(1025 - S_k, 0): min(S + P) > 1024  <=>  P_k >= theta_k for all k.
// Chains with fewer than SP_SMALL blocks to walk in this launch take the one-wave path (waves 1-3 leave at once).
// ------------------------------------------------------------------------------------------

#ifdef FELICS_SPINE_STAMPS
// Diagnostic build only: s_memtime ticks of the walker of ONE chain (context 1 of plane 0, the longest of an S1 frame),
// summed by phase: [0] waiting at the batch barrier, [1] block steps without a halving, [2] prefix-sum reads + first
// threshold round up to the ballot, [3] the rest of the halving rounds, [4] halvings, [5] blocks, [6] batches, [7] total.
__device__ unsigned long long g_spine_stamps[8];
#define SSTAMP(i)                                                      \
    do {                                                               \
        if (stamped) {                                                 \
            const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
            st_acc[i] += now_ - st_last;                               \
            st_last = now_;                                            \
        }                                                              \
    } while (0)
#define SCOUNT(i) do { if (stamped) st_acc[i]++; } while (0)
extern "C" __attribute__((visibility("default"))) int felics_debug_spine_stamps(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_spine_stamps), sizeof(g_spine_stamps)) != hipSuccess) return -1;
    if (reset) {
        static unsigned long long z[8] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_spine_stamps), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#else
#define SSTAMP(i)
#define SCOUNT(i)
#endif

// Blocks per hand-over (the kernel's template parameter SP_BATCH).  Sixteen for batches (round 4: 64 frames per blocking call 3.70 ->
// 3.54 ms, the queued step 2.78 -> 2.75; 24 and 32 are faster alone and slower in the queue: LDS), eight for a few planes, where the
// walker's first wait counts (one 4K frame 2.50 against 2.63 ms with sixteen).
constexpr uint32_t SP_BATCH_FEW_PLANES = 8, SP_BATCH_MANY_PLANES = 16, SP_MANY_PLANES = 16;
#ifndef FELICS_SP_HELPERS
#define FELICS_SP_HELPERS 3
#endif
constexpr uint32_t SP_HELPERS = FELICS_SP_HELPERS;
constexpr uint32_t SP_SMALL = 24;   // blocks

typedef unsigned short pk_u16 __attribute__((ext_vector_type(2)));
// both 16-bit halves of p >= the halves of theta
__device__ __forceinline__ bool pk_all_ge(uint32_t p, uint32_t theta) {
    const pk_u16 a = __builtin_bit_cast(pk_u16, p), b = __builtin_bit_cast(pk_u16, theta);
    const pk_u16 m = __builtin_elementwise_max(a, b);
    return __builtin_bit_cast(uint32_t, m) == p;
}

template <typename ET, uint32_t SP_BATCH>
__global__ __launch_bounds__(64 * (1 + SP_HELPERS)) void k_spine2(const ET *__restrict__ sorted_e, uint32_t *__restrict__ block_state,
                                                const uint32_t *__restrict__ chain_base,
                                                const uint32_t *__restrict__ chain_len, uint32_t nchains,
                                                const uint32_t *__restrict__ tile_off, uint32_t ntiles, uint32_t t_end,
                                                uint32_t *__restrict__ chain_prog, uint32_t *__restrict__ block_tag,
                                                uint2 *__restrict__ partial, uint32_t stamp) {
    struct Multi {
        uint32_t pref[2][SP_BATCH][3][64];       // [buffer][block][register][lane]
        uint32_t bsum[2][(SP_BATCH + 1) * 8];    // [buffer][block][k]
    };
    union Shared {
        SpineSingleLDS<ET> single;
        Multi multi;
    };
    __shared__ Shared sh;
    if (blockIdx.x >= nchains) return;
    constexpr uint32_t NC = nctx_of<ET>();
    const uint32_t nplanes = nchains / NC;
    const uint32_t ctx = blockIdx.x / nplanes, plane = blockIdx.x % nplanes;
    const uint32_t chain = plane * NC + ctx;
    // (wave-uniform values loaded from memory, and said so: the walk's loop control then runs on the scalar unit)
    const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)chain_len[chain]);
    if (n == 0) return;
    const uint32_t lane = lane_id();
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t l7 = lane & 7u;
    const bool final_slice = t_end >= ntiles;
    const uint32_t avail = final_slice ? n : (uint32_t)__builtin_amdgcn_readfirstlane((int)tile_off[((uint64_t)plane * ntiles + t_end) * NC + ctx]);  // events in place
    const uint32_t nblocks = final_slice ? (n + 63u) >> 6 : avail >> 6;
    uint32_t *prog = chain_prog + (uint64_t)chain * 8;  // [0] next block, [1..6] state
    const uint32_t first_block = (uint32_t)__builtin_amdgcn_readfirstlane((int)prog[0]);
    if (first_block >= nblocks || nblocks - first_block < SP_SMALL) {
        if (wave == 0)
            spine_single<ET>(sh.single, sorted_e, block_state, chain_base, chain_len, nchains, tile_off, ntiles, t_end, chain_prog, block_tag,
                             partial, stamp);
        return;
    }
    const uint32_t base = (uint32_t)__builtin_amdgcn_readfirstlane((int)chain_base[chain]);  // multiple of 64
    const ET *ev_src = sorted_e + base;
    uint4 *states = reinterpret_cast<uint4 *>(block_state) + (uint64_t)(base >> 6) * 2;
    uint32_t *tags = block_tag + (base >> 6);
    const uint32_t nbatches = (nblocks - first_block + SP_BATCH - 1) / SP_BATCH;
    Multi &m = sh.multi;

    if (wave != 0) {
        // ---- helpers: helper h takes blocks h, h + 3, ... of every batch; the events of batch t + 1 are loaded while batch t is summed
        const uint32_t h = wave - 1;
        constexpr uint32_t PER = (SP_BATCH + SP_HELPERS - 1) / SP_HELPERS;
        uint32_t ev[PER], nxt[PER];
        auto load_batch = [&](uint32_t t, uint32_t (&dst)[PER]) {
#pragma unroll
            for (uint32_t u = 0; u < PER; u++) {
                const uint32_t j = h + u * SP_HELPERS, b = first_block + t * SP_BATCH + j;
                dst[u] = 0;
                if (t < nbatches && j < SP_BATCH && b < nblocks) dst[u] = (uint32_t)ev_src[(uint64_t)b * 64 + lane];
            }
        };
        load_batch(0, ev);
        for (uint32_t t = 0; t < nbatches + 1; t++) {
            load_batch(t + 1, nxt);
            const uint32_t buf = t & 1u;
            if (t < nbatches) {
#pragma unroll
                for (uint32_t u = 0; u < PER; u++) {
                    const uint32_t j = h + u * SP_HELPERS;
                    if (j < SP_BATCH) {  // (blocks past the chain's end: their sums are never used)
                        uint32_t l01, l23, l45;
                        packed_lengths(ev[u], l01, l23, l45);
                        const uint32_t p01 = wave_incl_scan(l01), p23 = wave_incl_scan(l23), p45 = wave_incl_scan(l45);
                        m.pref[buf][j][0][lane] = p01;
                        m.pref[buf][j][1][lane] = p23;
                        m.pref[buf][j][2][lane] = p45;
                        const uint32_t s01 = readlane(p01, 63), s23 = readlane(p23, 63), s45 = readlane(p45, 63);
                        if (lane < 8) {
                            const uint32_t pair = lane < 2 ? s01 : lane < 4 ? s23 : s45;
                            m.bsum[buf][j * 8 + lane] = lane < 6 ? (pair >> ((lane & 1u) << 4)) & 0xFFFFu : 0u;
                        }
                    }
                }
            }
#pragma unroll
            for (uint32_t u = 0; u < PER; u++) ev[u] = nxt[u];
            __syncthreads();
        }
        return;
    }

    // ---- the walker
    uint32_t Sv = l7 < 6 ? prog[1 + l7] : 0u;  // lane l holds S[l & 7] (entries 6, 7: zero)
    asm volatile("; state in %0" : "+v"(Sv));
    __builtin_amdgcn_s_setprio(3);  // a chain is one long dependent instruction stream: never make it wait for issue
    const uint32_t sh16 = (l7 & 1u) << 4;
#ifdef FELICS_SPINE_STAMPS
    const bool stamped = ctx == 1 && plane == 0;
    unsigned long long st_acc[8] = {}, st_last = __builtin_amdgcn_s_memtime();
    const unsigned long long st_begin = st_last;
#endif
    __syncthreads();  // iteration 0: the helpers' first batch
    SSTAMP(0);
    // The block-start states go out eight blocks at a time: the state vector is replicated in every group of eight lanes
    // (lane l holds S[l & 7]), so the lanes 8 u .. 8 u + 7 keep a copy of it at the start of the group's block u -- one
    // select per block, no memory operation in the walk -- and every eight blocks one coalesced 256-byte store carries
    // eight records (block_state: 8 words per block), one more their tags.  Nothing in the walk waits for these stores.
    uint32_t *state_words = reinterpret_cast<uint32_t *>(states);
    for (uint32_t t = 1; t <= nbatches; t++) {
        const uint32_t buf = (t - 1) & 1u;
        const uint32_t bb = first_block + (t - 1) * SP_BATCH;
        const uint32_t nb = min(SP_BATCH, nblocks - bb);
        const uint32_t *bs = m.bsum[buf];
        uint32_t Bv = bs[l7];
        for (uint32_t j0 = 0; j0 < nb; j0 += 8) {
            uint32_t held = 0;
#pragma unroll
            for (uint32_t u = 0; u < 8; u++) {
                const uint32_t j = j0 + u;
                if (j < nb) {
                    const uint32_t Bnext = bs[(j + 1) * 8 + l7];  // look-ahead: independent of the state (row SP_BATCH: never used)
                    held = (lane >> 3) == u ? Sv : held;
                    const uint32_t Ev = Sv + Bv;
                    const uint32_t over = (uint32_t)__ballot(Ev > 1024u) & 0x3Fu;
                    SCOUNT(5);
                    if (over != 0x3Fu) {  // some counter still <= 1024 at the end of the block: no halving inside
                        Sv = Ev;
                        Bv = Bnext;
                        SSTAMP(1);
                    } else {
                        SSTAMP(1);
                        // a halving happens inside this block
                        const uint32_t p01 = m.pref[buf][j][0][lane], p23 = m.pref[buf][j][1][lane], p45 = m.pref[buf][j][2][lane];
                        uint32_t basev = 0;          // P_k at the block's last halving so far (state-vector layout, like Sv)
                        uint64_t live = ~0ull;       // lanes behind the last halving
                        while (true) {
                            // theta_k = base_k + max(1025 - S_k, 0), two to a register: lane 0 -> k = 0, 1; lane 2 -> 2, 3; lane 4 -> 4, 5
                            const uint32_t theta = basev + (uint32_t)max(1025 - (int)Sv, 0);
                            const uint32_t up = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)theta, 0x101, 0xF, 0xF, true);  // row_shl:1
                            const uint32_t th2 = theta | (up << 16);
                            const uint32_t T01 = readlane(th2, 0), T23 = readlane(th2, 2), T45 = readlane(th2, 4);
                            const uint64_t hm = __ballot(pk_all_ge(p01, T01)) & __ballot(pk_all_ge(p23, T23)) & __ballot(pk_all_ge(p45, T45)) & live;
                            // (the block-level test said a halving exists, and the loop is only re-entered when the end state says
                            // there is another: hm is never empty here -- but a wave that spins on a broken invariant takes the GPU
                            // with it)
                            if (hm == 0) break;
                            SSTAMP(2);
                            SCOUNT(4);
                            const uint32_t f = (uint32_t)__builtin_ctzll(hm);
                            const uint32_t q01 = readlane(p01, f), q23 = readlane(p23, f), q45 = readlane(p45, f);
                            const uint32_t qv = l7 < 2 ? q01 : l7 < 4 ? q23 : q45;
                            const uint32_t Pf = l7 < 6 ? (qv >> sh16) & 0xFFFFu : 0u;
                            Sv = (Sv + Pf - basev) >> 1;  // x /= 2 on every counter (parameter_selection.rs:62)
                            basev = Pf;
                            if (f == 63) break;
                            live = ~0ull << (f + 1);
                            // another halving in this block only if the end state still has all six counters above 1024
                            if (((uint32_t)__ballot(Sv + Bv - basev > 1024u) & 0x3Fu) != 0x3Fu) break;
                        }
                        Sv += Bv - basev;
                        Bv = Bnext;
                        SSTAMP(3);
                    }
                }
            }
            // eight records (fewer at the chain's end): lane l -> word l & 7 of block j0 + l / 8
            const uint32_t nrec = min(8u, nb - j0);
            if ((lane >> 3) < nrec) state_words[(uint64_t)(bb + j0) * 8 + lane] = l7 < 6 ? held : 0u;
            if (lane < nrec) tags[bb + j0 + lane] = stamp;  // resolved in this launch (a block is resolved exactly once)
        }
        SCOUNT(6);
        __syncthreads();
        SSTAMP(0);
    }
#ifdef FELICS_SPINE_STAMPS
    if (stamped && lane == 0) {
        for (int i = 0; i < 7; i++) atomicAdd(&g_spine_stamps[i], st_acc[i]);
        atomicAdd(&g_spine_stamps[7], __builtin_amdgcn_s_memtime() - st_begin);
    }
#endif
    if (lane < 6) prog[1 + lane] = Sv;
    if (lane == 6) prog[0] = nblocks;
    // Events that are in place but do not fill a block yet: publish the block's start state and which block it is (k_spine)
    if (!final_slice && (avail & 63u) != 0) {
        const uint32_t s0 = readlane(Sv, 0), s1 = readlane(Sv, 1), s2 = readlane(Sv, 2);
        const uint32_t s3 = readlane(Sv, 3), s4 = readlane(Sv, 4), s5 = readlane(Sv, 5);
        if (lane == 0) {
            states[(uint64_t)nblocks * 2] = make_uint4(s0, s1, s2, s3);
            states[(uint64_t)nblocks * 2 + 1] = make_uint4(s4, s5, 0u, 0u);
            partial[chain] = make_uint2((base >> 6) + nblocks, avail & 63u);
        }
    }
}

constexpr uint32_t TAG_SLICE_BITS = 5;  // a block's tag = epoch << 5 | slice of the spine launch that resolved it

// ------------------------------------------------------------------------------------------
// k of every event, in CHAIN order (k_assign_serial): one LANE per 64-event block.
//
// A lane loads its block's start state (k_spine) and its 64 events and replays the estimator event by event
// (parameter_selection.rs:49-85): k = argmin of the six counters, ties to the largest k (`<=` at :79), taken BEFORE the
// update (compression.rs:127,139); update; halve when the minimum exceeds 1024.  No cross-lane operation: 64 lanes = 64
// independent blocks, ~30 instructions per event instead of the ~290 lane-instructions per event of the wave-wide
// prefix-sum form of rounds 1-2, and every block is computed exactly once.  k leaves as one byte per event slot,
// 64 consecutive bytes per lane (k_sorted[slot]): the pack stage gathers it through the runs of its tile
// (k_pack_g), so nothing is scattered to pixel order in HBM.
//
// The six counters are kept as KEYS: key_k = S_k << 3 | (7 - k).  The smallest key names the smallest counter and,
// among equal counters, the largest k; min(S) > 1024 <=> min key >= 1025 << 3.
// The kernel runs once per slice behind that slice's spine launch and serves what it published, like k_assign: the
// blocks tagged with this launch's stamp (thread = block) and, per chain, the block whose events are in place but
// which is not full yet (thread = chain, behind the block threads).
// ------------------------------------------------------------------------------------------

// replays block gb from its start state and stores the k of its 64 events (one 16-byte store per 16 events)
template <typename ET>
__device__ __forceinline__ void replay_block(const ET *__restrict__ sorted_e, const uint32_t *__restrict__ block_state,
                                             uint8_t *__restrict__ k_sorted, uint32_t gb) {
    const uint4 sa = reinterpret_cast<const uint4 *>(block_state)[(uint64_t)gb * 2];
    const uint4 sb = reinterpret_cast<const uint4 *>(block_state)[(uint64_t)gb * 2 + 1];
    constexpr uint32_t EPW = 4 / sizeof(ET);       // events per dword
    constexpr uint32_t NW = 64 / EPW;              // dwords per block
    const uint4 *src = reinterpret_cast<const uint4 *>(sorted_e + (uint64_t)gb * 64);
    uint32_t w[NW];
#pragma unroll
    for (uint32_t q = 0; q < NW / 4; q++) {
        const uint4 v = src[q];
        w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
    }
    EstKeys est;
    est.set(sa.x, sa.y, sa.z, sa.w, sb.x, sb.y);
    uint4 *dst = reinterpret_cast<uint4 *>(k_sorted + (uint64_t)gb * 64);
#pragma unroll
    for (uint32_t c = 0; c < 4; c++) {  // sixteen events -> four k dwords -> one store
        uint32_t kw[4];
#pragma unroll
        for (uint32_t d = 0; d < 4; d++) {
            uint32_t kk = 0;
#pragma unroll
            for (uint32_t b = 0; b < 4; b++) {
                const uint32_t i = c * 16 + d * 4 + b;  // event index in the block
                const uint32_t word = w[i / EPW], sh = (i % EPW) * 8u * sizeof(ET);
                const uint32_t e = sizeof(ET) == 1 ? (word >> sh) & 0xFFu : (word >> sh) & 0xFFFFu;
                kk |= est.step(e) << (8u * b);
            }
            kw[d] = kk ^ 0x07070707u;  // 7 - (7 - k) in every byte
        }
        dst[c] = make_uint4(kw[0], kw[1], kw[2], kw[3]);
    }
}

// Persistent: a fixed grid strides over the blocks of the pass (their number is only known on the device); a wave looks at
// 64 consecutive tags at a time -- the blocks a spine launch resolved are long runs of consecutive blocks of the long
// chains, so a wave's 64 lanes are nearly always all busy or all idle.
template <typename ET>
__global__ __launch_bounds__(256) void k_assign_serial(const ET *__restrict__ sorted_e, const uint32_t *__restrict__ block_state,
                                                       uint8_t *__restrict__ k_sorted, const uint32_t *__restrict__ total_slots,
                                                       const uint32_t *__restrict__ block_tag, const uint2 *__restrict__ partial,
                                                       uint32_t nchains, uint32_t stamp) {
    const uint32_t nblocks = *total_slots >> 6;
    const uint32_t stride = gridDim.x * 256;
    for (uint32_t b0 = blockIdx.x * 256; b0 < nblocks; b0 += stride) {
        const uint32_t gb = b0 + threadIdx.x;
        if (gb < nblocks && block_tag[gb] == stamp) replay_block<ET>(sorted_e, block_state, k_sorted, gb);
    }
    // per chain: the block whose events are in place but which is not full yet
    for (uint32_t chain = blockIdx.x * 256 + threadIdx.x; chain < nchains; chain += stride) {
        const uint2 entry = partial[chain];  // {block, events in place}; y == 0: no such block after this slice
        if (entry.y == 0) continue;
        // Replay what is there.  Stores go out as whole dwords, so up to three slots behind the last event in place receive
        // a k computed from whatever those slots hold; they belong to later tiles, whose pack launches only run after a
        // later launch of this kernel has replayed the whole block.  (A separate code path with a run-time bound: the main
        // path above is fully unrolled.)
        const uint32_t nvalid = (entry.y + 3u) & ~3u;
        const uint32_t gb = entry.x;
        const uint4 sa = reinterpret_cast<const uint4 *>(block_state)[(uint64_t)gb * 2];
        const uint4 sb = reinterpret_cast<const uint4 *>(block_state)[(uint64_t)gb * 2 + 1];
        EstKeys est;
        est.set(sa.x, sa.y, sa.z, sa.w, sb.x, sb.y);
        const ET *src = sorted_e + (uint64_t)gb * 64;
        uint32_t *dst = reinterpret_cast<uint32_t *>(k_sorted + (uint64_t)gb * 64);
        for (uint32_t i = 0; i < nvalid; i += 4) {
            uint32_t kk = 0;
#pragma unroll
            for (uint32_t b = 0; b < 4; b++) kk |= est.step((uint32_t)src[i + b]) << (8u * b);
            dst[i >> 2] = kk ^ 0x07070707u;
        }
    }
}

// Two-pass pack only (exact placement after a slot overflow, FELICS_TWO_PASS, or after a look-back gave up): k from chain
// order to a byte per pixel, k_map[pix_of[slot]] = k_sorted[slot], once every chain has been replayed.  pix_of holds
// plane * npix + i here (k_scatter<.., false>), 0xFFFFFFFF in the padding slots of a chain's last block.
__global__ __launch_bounds__(256) void k_k_to_pixels(const uint8_t *__restrict__ k_sorted, const uint32_t *__restrict__ pix_of,
                                                     uint8_t *__restrict__ k_map, const uint32_t *__restrict__ total_slots) {
    const uint32_t n = *total_slots;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const uint32_t pix = pix_of[i];
        if (pix != 0xFFFFFFFFu) k_map[pix] = k_sorted[i];
    }
}

// ------------------------------------------------------------------------------------------
// code construction shared by lengths / pack
// ------------------------------------------------------------------------------------------

// ------------------------------------------------------------------------------------------
// Tile staging shared by lengths / pack.  A tile is PACK_TILE consecutive pixels of one plane;
// thread t owns the PACK_PER_THREAD consecutive pixels first = tile*PACK_TILE + t*PACK_PER_THREAD.
// The tile's pixels, the span one row above it and its k bytes are copied into LDS with wide
// coalesced loads; each thread then pulls its 16 pixels into registers with 16-byte LDS reads.
// Interior pixels (x > 0, y > 0: left and above neighbours) are served from those registers; the
// few pixels on the first row / first column take the general neighbour rule from global memory.
// ------------------------------------------------------------------------------------------

constexpr uint32_t STAGE_LEAD = 16;  // elements kept in front of the tile (the left neighbours), keeps 16-B alignment

template <typename T>
struct TileLDS {
    alignas(16) T cur[STAGE_LEAD + PACK_TILE];  // cur[STAGE_LEAD + j] = pixel tile_first + j
    alignas(16) T up[PACK_TILE + 16];            // up[j] = pixel tile_first - W + j
    alignas(16) uint8_t kq[PACK_TILE];           // k of pixel tile_first + j
    alignas(16) uint32_t dump[4];                // where stage_pixels puts the chunks it does not want
};

// lds[j] = g[first + j] for j in [0, count), indices outside [0, limit) skipped; 16-byte copies
// when the global side is 16-byte aligned.
template <typename U>
__device__ __forceinline__ void stage_span(U *lds, const U *__restrict__ g, int64_t first, uint32_t count,
                                           uint32_t limit) {
    constexpr uint32_t EPC = 16 / sizeof(U);  // elements per 16-byte chunk
    const bool aligned = ((reinterpret_cast<uintptr_t>(g) + (uint64_t)first * sizeof(U)) & 15u) == 0;
    const uint32_t nchunks = (count + EPC - 1) / EPC;
    for (uint32_t c = threadIdx.x; c < nchunks; c += PACK_THREADS) {
        const int64_t g0 = first + (int64_t)c * EPC;
        if (aligned && g0 >= 0 && g0 + EPC <= (int64_t)limit) {
            reinterpret_cast<uint4 *>(lds)[c] = *reinterpret_cast<const uint4 *>(g + g0);
        } else {
#pragma unroll
            for (uint32_t e = 0; e < EPC; e++) {
                const int64_t gi = g0 + e;
                if (gi >= 0 && gi < (int64_t)limit && c * EPC + e < count) lds[c * EPC + e] = g[gi];
            }
        }
    }
}

// cur and up of one tile.  All 16-byte loads of both spans are issued before the first of them is waited for (one
// memory round trip instead of one per span and trip); chunks at the ends of the plane, or of a span that is not
// 16-byte aligned in memory, go element by element afterwards.
template <typename T>
__device__ __forceinline__ void stage_pixels(TileLDS<T> &t, const T *__restrict__ pl, uint32_t tile_first, uint32_t W, uint32_t npix) {
    constexpr uint32_t EPC = 16 / sizeof(T);  // elements per 16-byte chunk
    constexpr uint32_t NA = (STAGE_LEAD + PACK_TILE) / EPC, NB = (PACK_TILE + 16) / EPC;
    constexpr uint32_t KA = (NA + PACK_THREADS - 1) / PACK_THREADS, KB = (NB + PACK_THREADS - 1) / PACK_THREADS;
    static_assert((STAGE_LEAD + PACK_TILE) % EPC == 0 && (PACK_TILE + 16) % EPC == 0, "whole chunks");
    const int64_t fa = (int64_t)tile_first - STAGE_LEAD, fb = (int64_t)tile_first - W;
    const bool ala = ((reinterpret_cast<uintptr_t>(pl) + (uint64_t)fa * sizeof(T)) & 15u) == 0;
    const bool alb = ((reinterpret_cast<uintptr_t>(pl) + (uint64_t)fb * sizeof(T)) & 15u) == 0;
    // what a chunk that is not loaded this way reads instead: the 16 aligned bytes the plane starts in (the planes of a
    // batch lie in one allocation, so these exist)
    const uint4 *safe = reinterpret_cast<const uint4 *>(pl - (reinterpret_cast<uintptr_t>(pl) & 15u) / sizeof(T));
    auto slowly = [&](T *lds, int64_t first, uint32_t c) {
        for (uint32_t e = 0; e < EPC; e++) {
            const int64_t gi = first + (int64_t)c * EPC + e;
            if (gi >= 0 && gi < (int64_t)npix) lds[c * EPC + e] = pl[gi];
        }
    };
    // The whole rounds of both spans (every thread one chunk of each per round: one round for byte samples, two for 16-bit
    // ones), without a branch (a branch would be a wait per load; every load is stored, the unwanted ones into a dump slot:
    // a load that is only used under a condition is moved under that condition by the compiler, and then waited for there).
    constexpr uint32_t FA = NA / PACK_THREADS, FB = NB / PACK_THREADS;
    static_assert(FA == FB && KA <= FA + 1 && KB <= FB + 1, "whole rounds, then at most one partial round");
    uint4 va[FA], vb[FB];
    bool wa[FA], wb[FB];
#pragma unroll
    for (uint32_t k = 0; k < FA; k++) {
        const uint32_t c = threadIdx.x + k * PACK_THREADS;
        const int64_t ga = fa + (int64_t)c * EPC, gb = fb + (int64_t)c * EPC;
        wa[k] = ala && ga >= 0 && ga + EPC <= (int64_t)npix;
        wb[k] = alb && gb >= 0 && gb + EPC <= (int64_t)npix;
        va[k] = *(wa[k] ? reinterpret_cast<const uint4 *>(pl + ga) : safe);
        vb[k] = *(wb[k] ? reinterpret_cast<const uint4 *>(pl + gb) : safe);
    }
    uint4 *dump = reinterpret_cast<uint4 *>(t.dump);
#pragma unroll
    for (uint32_t k = 0; k < FA; k++) {
        const uint32_t c = threadIdx.x + k * PACK_THREADS;
        *(wa[k] ? reinterpret_cast<uint4 *>(t.cur) + c : dump) = va[k];
        *(wb[k] ? reinterpret_cast<uint4 *>(t.up) + c : dump) = vb[k];
    }
#pragma unroll
    for (uint32_t k = 0; k < FA; k++) {
        const uint32_t c = threadIdx.x + k * PACK_THREADS;
        if (!wa[k]) slowly(t.cur, fa, c);
        if (!wb[k]) slowly(t.up, fb, c);
    }
    // the chunks beyond (the elements in front of / behind the tile: one or two chunks of each span): a few threads
    const uint32_t c1 = threadIdx.x + FA * PACK_THREADS;
    if (c1 < NA) {
        const int64_t g1 = fa + (int64_t)c1 * EPC;
        if (ala && g1 >= 0 && g1 + EPC <= (int64_t)npix) reinterpret_cast<uint4 *>(t.cur)[c1] = *reinterpret_cast<const uint4 *>(pl + g1);
        else slowly(t.cur, fa, c1);
    }
    if (c1 < NB) {
        const int64_t g1 = fb + (int64_t)c1 * EPC;
        if (alb && g1 >= 0 && g1 + EPC <= (int64_t)npix) reinterpret_cast<uint4 *>(t.up)[c1] = *reinterpret_cast<const uint4 *>(pl + g1);
        else slowly(t.up, fb, c1);
    }
}

template <typename T>
__device__ __forceinline__ void stage_tile(TileLDS<T> &t, const T *__restrict__ pl, const uint8_t *__restrict__ kpl,
                                           uint32_t tile_first, uint32_t W, uint32_t npix) {
    stage_span<T>(t.cur, pl, (int64_t)tile_first - STAGE_LEAD, STAGE_LEAD + PACK_TILE, npix);
    stage_span<T>(t.up, pl, (int64_t)tile_first - W, PACK_TILE + 16, npix);
    stage_span<uint8_t>(t.kq, kpl, (int64_t)tile_first, PACK_TILE, npix);
}

// Calls raw(i, value) for pixels 0 and 1 of the plane (stored as 32-bit values,
// compression.rs:105-106) and f(pc, k) for every other pixel of this thread's group, in raster order.
// The neighbour rule (misc.rs:6-24) is applied from registers: the left neighbours come from the
// group itself, the row above from `up`; only the second neighbour of a first-column pixel
// (two rows up) is fetched from global memory, once per image row.
template <typename T, typename FR, typename F>
__device__ __forceinline__ void walk_group(const TileLDS<T> &t, const uint8_t *kq, const T *__restrict__ pl, uint32_t first,
                                           uint32_t end, uint32_t W, FR &&raw, F &&f) {
    if (first >= end) return;
    constexpr uint32_t NW = PACK_PER_THREAD * sizeof(T) / 4;  // dwords holding the group's pixels
    const uint32_t off = threadIdx.x * PACK_PER_THREAD;
    uint32_t cw[NW], uw[NW], kw[4];
#pragma unroll
    for (uint32_t q = 0; q < NW / 4; q++) {
        const uint4 a = reinterpret_cast<const uint4 *>(t.cur + STAGE_LEAD + off)[q];
        const uint4 b = reinterpret_cast<const uint4 *>(t.up + off)[q];
        cw[4 * q] = a.x; cw[4 * q + 1] = a.y; cw[4 * q + 2] = a.z; cw[4 * q + 3] = a.w;
        uw[4 * q] = b.x; uw[4 * q + 1] = b.y; uw[4 * q + 2] = b.z; uw[4 * q + 3] = b.w;
    }
    {
        const uint4 c = *reinterpret_cast<const uint4 *>(kq + off);
        kw[0] = c.x; kw[1] = c.y; kw[2] = c.z; kw[3] = c.w;
    }
    int left = (int)t.cur[STAGE_LEAD + off - 1], left2 = (int)t.cur[STAGE_LEAD + off - 2];
    for (uint32_t i = first; i < min(end, 2u); i++) raw(i, (uint32_t)(int)pl[i]);
    Coord xy;
    xy.set(first, W);
    // Four pixels per trip; the register arrays are shifted down after each trip so that every
    // index below is a compile-time constant while the loop itself stays rolled (code size).
    constexpr uint32_t D = sizeof(T);  // dwords per four pixels
    uint32_t up_tail = (uint32_t)(int)t.up[off + PACK_PER_THREAD];  // above-right of the group's last pixel
#pragma nounroll
    for (uint32_t g = 0; g < PACK_PER_THREAD; g += 4) {
        const uint32_t un = NW > D ? uw[NW > D ? D : 0] : up_tail;  // dword after this trip's `up` samples
#pragma unroll
        for (uint32_t j = 0; j < 4; j++) {
            const uint32_t i = first + g + j;
            const int p = sample_at(cw, j, T());
            if (i < end && i >= 2) {
                const uint32_t k = (kw[0] >> (8u * j)) & 0xFFu;
                const int above = sample_at(uw, j, T());
                // the neighbour rule (misc.rs:6-24) with selects instead of branches: interior = left and above; first row =
                // the two pixels to the left; first column = above and two rows up (above-right for pixel (0,1)).  Only the
                // two-rows-up sample needs a branch: it is the one value that is not in registers.
                const bool row0 = xy.y == 0, col0 = xy.x == 0 && !row0;
                const int v1 = col0 ? above : left;
                int v2 = row0 ? left2 : above;
                if (col0) v2 = xy.y >= 2 ? (int)pl[i - 2 * W] : (j < 3 ? sample_at(uw, (j + 1) & 3u, T()) : sample_at(&un, 0, T()));
                const int H = max(v1, v2), L = min(v1, v2);
                const int d = p - L, ctx = H - L;  // in range: 0 <= d <= ctx
                const bool below = d < 0, over = d > ctx;
                PixelClass pc;
                pc.ctx = (uint32_t)ctx;
                pc.cls = (below ? (uint32_t)CLS_BELOW : 0u) | (over ? (uint32_t)CLS_ABOVE : 0u);
                // L - p - 1 = ~d ; p - H - 1 = d - ctx - 1 ; p - L = d: all three computed, two selects
                uint32_t val = below ? (uint32_t)~d : (uint32_t)d;
                val = over ? (uint32_t)(d - ctx - 1) : val;
                pc.val = val;
                f(pc, k);
            }
            left2 = left;
            left = p;
            if (++xy.x == W) {
                xy.x = 0;
                xy.y++;
            }
        }
#pragma unroll
        for (uint32_t q = 0; q + D < NW; q++) {
            cw[q] = cw[q + D];
            uw[q] = uw[q + D];
        }
        // once the real samples are used up, the next dword of `up` is the tail element
#pragma unroll
        for (uint32_t q = NW - D; q < NW; q++) uw[q] = up_tail;
        kw[0] = kw[1];
        kw[1] = kw[2];
        kw[2] = kw[3];
    }
}

// ------------------------------------------------------------------------------------------
// lengths: bits of every 16-pixel group (group_bits: u16 for 8-bit samples, u32 for 16-bit ones,
// whose codes reach 2^17 bits) and of every tile (tile_bits).
// Plane 0 of an image also carries the 112 header bits.
// ------------------------------------------------------------------------------------------

template <typename T>
__global__ __launch_bounds__(PACK_THREADS) void k_lengths(const T *__restrict__ planes, const uint8_t *__restrict__ k_map,
                                                          group_bits_t<T> *__restrict__ group_bits,
                                                          uint32_t *__restrict__ tile_bits, uint32_t W, uint32_t npix,
                                                          uint32_t ntiles, uint32_t planes_per_image,
                                                          uint32_t tile_begin) {
    __shared__ TileLDS<T> tl;
    __shared__ uint32_t wsum[PACK_THREADS / 64];
    const uint32_t tile = tile_begin + blockIdx.x, plane = blockIdx.y;
    const T *pl = planes + (uint64_t)plane * npix;
    const uint32_t tile_first = tile * PACK_TILE;
    stage_tile(tl, pl, k_map + (uint64_t)plane * npix, tile_first, W, npix);
    __syncthreads();
    const uint32_t first = tile_first + threadIdx.x * PACK_PER_THREAD;
    const uint32_t end = min(tile_first + PACK_TILE, npix);
    uint32_t bits = 0;
    walk_group(tl, tl.kq, pl, first, end, W, [&](uint32_t, uint32_t) { bits += 32u; },
               [&](const PixelClass &pc, uint32_t k) { bits += code_length(pc, k); });
    if (npix == 1 && first == 0) bits += 32;  // 1x1: second raw value is a literal 0 (compression.rs:99-103)
    if (tile == 0 && threadIdx.x == 0 && (plane % planes_per_image) == 0) bits += 8 * 14;  // header
    group_bits[((uint64_t)plane * ntiles + tile) * PACK_THREADS + threadIdx.x] = (group_bits_t<T>)bits;
    const uint32_t inc = wave_incl_scan(bits);
    if (lane_id() == 63) wsum[threadIdx.x >> 6] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (uint32_t w = 0; w < PACK_THREADS / 64; w++) tot += wsum[w];
        tile_bits[(uint64_t)plane * ntiles + tile] = tot;
    }
}

// ------------------------------------------------------------------------------------------
// bitscan: bit offset of every tile inside its plane, slice by slice.  One block per plane scans the
// tiles [t0, t1) of its plane on top of the plane's running total (plane_carry).  When the last slice
// is done k_finish_sizes turns the plane totals into each plane's offset inside its image stream
// (planes are concatenated with no alignment, compression.rs:365-367) and the stream's byte size.
// ------------------------------------------------------------------------------------------

__global__ __launch_bounds__(1024) void k_bitscan_slice(const uint32_t *__restrict__ tile_bits,
                                                        uint64_t *__restrict__ tile_bitoff,
                                                        uint64_t *__restrict__ plane_carry, uint32_t ntiles,
                                                        uint32_t t0, uint32_t t1) {
    __shared__ uint64_t wsum[16];
    __shared__ uint64_t carry;
    const uint32_t plane = blockIdx.x;
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
    const uint32_t *src = tile_bits + (uint64_t)plane * ntiles;
    uint64_t *dst = tile_bitoff + (uint64_t)plane * ntiles;
    if (threadIdx.x == 0) carry = plane_carry[plane];
    __syncthreads();
    for (uint32_t base = t0; base < t1; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < t1 ? src[i] : 0;
        // tile totals are < 2^32 but a wave of them may not be: scan low/high halves apart
        const uint32_t lo = wave_incl_scan(v & 0xFFFFu), hi = wave_incl_scan(v >> 16);
        const uint64_t inc = (uint64_t)lo + ((uint64_t)hi << 16);
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        uint64_t woff = 0;
        for (uint32_t w = 0; w < wave; w++) woff += wsum[w];
        const uint64_t c = carry;
        if (i < t1) dst[i] = c + woff + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = c + woff + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) plane_carry[plane] = carry;
}

__global__ void k_finish_sizes(const uint64_t *__restrict__ plane_carry, uint64_t *__restrict__ plane_base,
                               uint64_t *__restrict__ image_bytes, uint32_t nimages, uint32_t planes_per_image) {
    const uint32_t img = blockIdx.x * blockDim.x + threadIdx.x;
    if (img >= nimages) return;
    uint64_t bits = 0;
    for (uint32_t c = 0; c < planes_per_image; c++) {
        plane_base[img * planes_per_image + c] = bits;
        bits += plane_carry[img * planes_per_image + c];
    }
    image_bytes[img] = (bits + 7) >> 3;  // byte_align (compression.rs:279)
}

// Where a tile's bits live: stream base (fixed slot per image, or exact placement) and bit range.
struct Placement {
    const uint64_t *image_off;  // exact placement (slot_stride == 0): byte offset of every stream
    uint64_t slot_stride;       // fixed slots: stream i starts at i * slot_stride bytes
};

__device__ __forceinline__ uint32_t *stream_words(uint8_t *out, const Placement &pl, uint32_t img, uint64_t &limit_words) {
    if (pl.slot_stride) {
        limit_words = pl.slot_stride >> 2;
        return reinterpret_cast<uint32_t *>(out + (uint64_t)img * pl.slot_stride);
    }
    limit_words = ~0ull;
    return reinterpret_cast<uint32_t *>(out + pl.image_off[img]);
}

// pack ORs a tile's last word (and its first word when the previous tile ends inside it) into the
// output: zero the last word of every tile of the range.  A tile that lies inside one word shared
// with its predecessor leaves that word alone (the predecessor zeroed it, and may already have packed).
__global__ void k_zero_edges(uint8_t *__restrict__ out, Placement place, const uint64_t *__restrict__ tile_bitoff,
                             const uint32_t *__restrict__ tile_bits, const uint64_t *__restrict__ plane_base,
                             uint32_t ntiles, uint32_t t0, uint32_t t1, uint32_t planes_per_image) {
    const uint32_t tile = t0 + blockIdx.x * blockDim.x + threadIdx.x, plane = blockIdx.y;
    if (tile >= t1) return;
    const uint64_t lo = plane_base[plane] + tile_bitoff[(uint64_t)plane * ntiles + tile];
    const uint64_t hi = lo + tile_bits[(uint64_t)plane * ntiles + tile];
    const uint64_t first_word = lo >> 5, last_word = (hi - 1) >> 5;
    if (first_word == last_word && (lo & 31u) != 0) return;
    uint64_t limit;
    uint32_t *words = stream_words(out, place, plane / planes_per_image, limit);
    if (last_word < limit) words[last_word] = 0;
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

template <typename T>
__global__ __launch_bounds__(PACK_THREADS) void k_pack(const T *__restrict__ planes, const uint8_t *__restrict__ k_map,
                                                       const group_bits_t<T> *__restrict__ group_bits,
                                                       const uint64_t *__restrict__ tile_bitoff,
                                                       const uint32_t *__restrict__ tile_bits,
                                                       const uint64_t *__restrict__ plane_base, Placement place,
                                                       uint8_t *__restrict__ out, uint32_t W, uint32_t H,
                                                       uint32_t npix, uint32_t ntiles, uint32_t planes_per_image,
                                                       uint32_t color, uint32_t depth, uint32_t tile_begin) {
    __shared__ TileLDS<T> tl;
    __shared__ uint32_t win[PACK_WIN_WORDS];
    __shared__ uint32_t wsum[PACK_THREADS / 64];
    const uint32_t tile = tile_begin + blockIdx.x, plane = blockIdx.y;
    const uint32_t img = plane / planes_per_image;
    const bool first_plane = (plane % planes_per_image) == 0;
    const T *pl = planes + (uint64_t)plane * npix;
    const uint32_t tile_first = tile * PACK_TILE;
    const uint32_t first = tile_first + threadIdx.x * PACK_PER_THREAD;
    const uint32_t end = min(tile_first + PACK_TILE, npix);
    const bool has_header = tile == 0 && threadIdx.x == 0 && first_plane;

    stage_tile(tl, pl, k_map + (uint64_t)plane * npix, tile_first, W, npix);
    // this thread's bit offset inside the tile: scan of the group sizes k_lengths left behind
    const uint32_t bits = group_bits[((uint64_t)plane * ntiles + tile) * PACK_THREADS + threadIdx.x];
    const uint32_t inc = wave_incl_scan(bits);
    if (lane_id() == 63) wsum[threadIdx.x >> 6] = inc;
    __syncthreads();
    uint32_t woff = 0;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) woff += wsum[w];
    const uint64_t tile_lo = plane_base[plane] + tile_bitoff[(uint64_t)plane * ntiles + tile];  // bit offset in the image stream
    const uint64_t tile_hi = tile_lo + tile_bits[(uint64_t)plane * ntiles + tile];
    const uint64_t my_lo = tile_lo + woff + inc - bits;

    uint64_t limit_words;  // a stream that outgrows its slot is cut here (the host then re-packs with exact placement)
    uint32_t *out_words = stream_words(out, place, img, limit_words);
    const uint64_t first_word = tile_lo >> 5, last_word = (tile_hi - 1) >> 5;
    const bool first_shared = (tile_lo & 31u) != 0;  // the previous tile ends inside our first word

    for (uint64_t w0 = first_word; w0 <= last_word; w0 += PACK_WIN_WORDS) {
        for (uint32_t j = threadIdx.x; j < PACK_WIN_WORDS; j += PACK_THREADS) win[j] = 0;
        __syncthreads();
        // skip threads whose bits lie wholly outside this window
        if (bits != 0 && ((my_lo + bits - 1) >> 5) >= w0 && (my_lo >> 5) < w0 + PACK_WIN_WORDS) {
            LaneBits bw;
            bw.win = win;
            bw.win_words = PACK_WIN_WORDS;
            bw.win_word0 = w0;
            bw.begin(my_lo);
            if (has_header) {  // write_header, format.rs:51-61
                bw.put(0x464C4353u, 32);  // "FLCS"
                bw.put((color << 8) | depth, 16);
                bw.put(W, 32);
                bw.put(H, 32);
            }
            walk_group(tl, tl.kq, pl, first, end, W,
                       [&](uint32_t, uint32_t rv) {
                           bw.put(rv, 32);  // write_signed(32, p): sign-extended sample
                           if (npix == 1) bw.put(0u, 32);
                       },
                       [&](const PixelClass &pc, uint32_t k) { put_pixel(bw, pc, k); });
            bw.finish();
        }
        __syncthreads();
        for (uint32_t j = threadIdx.x; j < PACK_WIN_WORDS; j += PACK_THREADS) {
            const uint64_t aw = w0 + j;
            if (aw > last_word || aw >= limit_words) break;
            const uint32_t v = __builtin_bswap32(win[j]);
            if ((aw == first_word && first_shared) || aw == last_word) {
                if (v) atomicOr(&out_words[aw], v);  // zeroed beforehand (k_zero_edges / k_zero_streams)
            } else {
                out_words[aw] = v;
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// pack, single pass (8-bit samples, fixed output slots): lengths, bit offsets and packing of a tile in one kernel.
//
// Phase A: every thread builds the codes of its 16 pixels -- left-aligned in a register each, with their lengths -- and
// their total.  The thread totals are scanned inside the workgroup; the tile's offset in its plane comes from a decoupled
// look-back over the tiles before it: a tile first publishes its total (AGGREGATE), then wave 0 reads the status words of
// up to 64 predecessors at a time, adds aggregates until it meets a tile that already knows its inclusive PREFIX, and
// publishes its own.  A status word is one 64-bit value {epoch, state, bits}, written and read with agent-scope atomics, so
// it needs no ordering with any other memory; stale words of earlier submissions carry another epoch.  A tile only waits
// for tiles of smaller index in its plane, which have been dispatched (workgroup index) or are running (ticket); the wait is
// bounded all the same and reports through `error`.
// Phase B: every thread ORs its sixteen codes into the tile's LDS bit window at their final bit positions (two LDS
// atomics per code: a code of up to 32 bits touches two words), and the window is streamed out.  The two words a tile may
// share with its neighbours go to edge_first / edge_last instead of the output; k_join_edges merges them when all tiles
// are done, so the output needs no zeroing.
// (Rounds 1-3 strung a thread's codes together in a private LDS buffer first -- a 64-bit window per thread, one store per
// pixel -- and shifted that string into place afterwards: 14 vector instructions per pixel for the append and 11 for the
// merge, against 5 here; the codes wait in registers for the tile's offset instead.)
// ------------------------------------------------------------------------------------------

constexpr uint32_t FUSED_WIN_WORDS = PACK_TILE * 16 / 32;  // LDS bit window: 16 bits per pixel of a tile in one pass (more bits: more passes)

struct FusedArgs {
    uint64_t *status;
    uint64_t *tile_bitoff;
    uint32_t *tile_bits;
    uint64_t *plane_carry;
    uint32_t *edge_first, *edge_last, *error;
    PlaneOut po;
    uint32_t W, H, npix, ntiles, color, depth, epoch;
    // Tiles are handed out by a ticket counter (zeroed before the launch) in (tile, plane) order, or -- null -- by
    // blockIdx: with tickets a tile only ever waits for tiles held by workgroups that are already running, whatever else
    // shares the GPU -- also another pack kernel whose workgroups spin in their own look-back (with blockIdx two such
    // kernels can hold each other's predecessors out of the CUs: the XCDs dispatch their shares of a grid independently).
    uint32_t *ticket;
    uint32_t nplanes;
};
#ifdef FELICS_PACK_STAMPS
__device__ unsigned long long g_pack_stamps[256][16];
#define PSTAMP(i)                                                                              \
    do {                                                                                       \
        if (threadIdx.x == 0) {                                                                \
            const unsigned long long now_ = __builtin_amdgcn_s_memtime();                      \
            fl.t_acc[i] = now_ - fl.t_last;                                                    \
            fl.t_last = now_;                                                                  \
        }                                                                                      \
    } while (0)
#else
#define PSTAMP(i)
#endif
struct FusedLDS {
#ifdef FELICS_PACK_STAMPS
    unsigned long long t_last, t_acc[12];
#endif
    uint32_t win[FUSED_WIN_WORDS + 2];  // (+ 1: the second word of a code that starts in the window's last word)
    uint32_t wsum[PACK_THREADS / 64];
    uint64_t tile_lo_sh;
    uint32_t ticket_sh;
};

// Whether this thread's 16-pixel group takes the branch-free path (group_codes), and what that path needs from outside the
// tile's LDS image: the position of the group's first-column pixel (PACK_PER_THREAD: none) and that pixel's second neighbour
// (two rows up, or above-right in row 1: misc.rs:14-23).  Computed early by the kernels, so that the one global load is
// long back when the group is coded.
struct GroupGeom {
    bool fast;
    uint32_t j0;
    int special;
};
template <typename T>
__device__ __forceinline__ GroupGeom group_geometry(const T *__restrict__ pl, uint32_t tile, uint32_t W, uint32_t npix) {
    GroupGeom gg{false, PACK_PER_THREAD, 0};
    const uint32_t first = tile * PACK_TILE + threadIdx.x * PACK_PER_THREAD;
    const uint32_t end = min((tile + 1) * PACK_TILE, npix);
    // the whole group below the first image row and inside the plane, at most one first-column pixel in it
    if (W >= PACK_PER_THREAD && first + PACK_PER_THREAD <= end && first >= W) {
        gg.fast = true;
        Coord xy;
        xy.set(first, W);
        if (xy.x == 0) gg.j0 = 0;
        else if (xy.x + PACK_PER_THREAD > W) gg.j0 = W - xy.x;
        if (gg.j0 < PACK_PER_THREAD) {
            const uint32_t i0 = first + gg.j0;
            gg.special = (int)pl[i0 >= 2 * W ? i0 - 2 * W : i0 - W + 1];
        }
    }
    return gg;
}

// ---- Instruction forms.  profiles/r04/valu_rate.txt: a gfx950 SIMD issues 32-bit add / sub / and / or / xor / lshr / ashr,
// v_bitop3_b32 and every 16-bit VOP2 instruction (min, max, add, shifts) in 1.0 ns, everything else -- v_min_u32,
// v_lshlrev_b32, v_cndmask, compares, v_bfe, SDWA / DPP / VOP3 forms, 64-bit shifts -- in 1.7 ns.  The compiler prices them
// alike and turns sign masks back into compare + select, so the code builder names the cheap forms itself.
template <uint32_t TABLE>
__device__ __forceinline__ uint32_t bitop3(uint32_t a, uint32_t b, uint32_t c) {
    return __builtin_amdgcn_bitop3_b32(a, b, c, TABLE);  // bit i of the result = TABLE[a_i << 2 | b_i << 1 | c_i]
}
constexpr uint32_t BT_SEL = 0xE4;      // c ? a : b  =  (a & c) | (b & ~c)
constexpr uint32_t BT_OR_ANDN = 0xF4;  // a | (b & ~c)
constexpr uint32_t BT_ANDN = 0x30;     // a & ~b
__device__ __forceinline__ uint32_t min_u16(uint32_t a, uint32_t b) {  // operands < 2^16
    uint32_t r;
    asm("v_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t max_u16(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_max_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t twice(uint32_t a) {  // a + a as an add (the compiler would make it a left shift)
    uint32_t r;
    asm("v_add_u32 %0, %1, %1" : "=v"(r) : "v"(a));
    return r;
}
__device__ __forceinline__ uint32_t vgpr_const(uint32_t v) {  // a constant kept in a vector register: the shifted operand of v_lshrrev_b32_e32
    uint32_t r;
    asm("v_mov_b32 %0, %1" : "=v"(r) : "s"(v));
    return r;
}

// Sample j of a group held in packed registers, as an unsigned 16-bit value in an order-preserving offset: u8 samples as
// they are, i16 samples (Y / Co / Cg planes) with the sign bit flipped -- the codes depend on differences only.
__device__ __forceinline__ uint32_t field_at(const uint32_t *w, uint32_t j, uint8_t) {
    const uint32_t x = w[j >> 2];
    switch (j & 3u) {
        case 0: return x & 0xFFu;
        case 1: {
            uint32_t r;
            asm("v_lshrrev_b16 %0, 8, %1" : "=v"(r) : "v"(x));  // (low half >> 8, upper half cleared: one cheap instruction)
            return r;
        }
        case 2: return (x >> 16) & 0xFFu;
        default: return x >> 24;
    }
}
__device__ __forceinline__ uint32_t field_at(const uint32_t *w, uint32_t j, int16_t) {
    const uint32_t x = w[j >> 1] ^ 0x80008000u;
    return (j & 1u) ? x >> 16 : x & 0xFFFFu;
}
__device__ __forceinline__ uint32_t field_of(int v, uint8_t) { return (uint32_t)v & 0xFFu; }
__device__ __forceinline__ uint32_t field_of(int v, int16_t) { return ((uint32_t)v ^ 0x8000u) & 0xFFFFu; }

// One pixel's code, left-aligned in 32 bits, and its length (compression.rs:124-145): p against its two neighbours a, b
// (unordered; all three in field_at's form), k = the Rice parameter of the pixel's context (used if p is out of range).
//   in range (L <= p <= H): `1`, then p - L phased-in on n = H - L + 1 values (phase_in_coding.rs:59-84): r = (p - L + P) mod n,
//       P = 2^m = the largest power of two <= n; r < 2 P - n: r in m bits, else r + 2 P - n in m + 1 bits;
//   below / above: `00` / `01`, then L - p - 1 / p - H - 1 Rice-coded: q ones, `0`, k low bits (rice_coding.rs:26-38).
// Both are built and one is kept.  A Rice code longer than 32 bits comes out as garbage with len > 32: the caller redoes
// such a group the general way.  K31 = 0x80000000, K7F = 0x7FFFFFFF in vector registers (vgpr_const).
struct PixelCode {
    uint32_t c32, len;
};
__device__ __forceinline__ PixelCode code_pixel(uint32_t p, uint32_t a, uint32_t b, uint32_t k, uint32_t K31, uint32_t K7F) {
    const uint32_t L = min_u16(a, b), H = max_u16(a, b);
    const uint32_t ctx = H - L;
    const int d = (int)(p - L);              // in range: 0 <= d <= ctx
    const int below = d >> 31;               // all ones: p < L
    const int o = (int)p - (int)H - 1;       // >= 0: p > H
    const int not_above = o >> 31;
    const uint32_t val = bitop3<BT_SEL>((uint32_t)(d ^ below), (uint32_t)o, (uint32_t)not_above);  // ~d = L - p - 1 | d | p - H - 1
    // phased-in
    const uint32_t n = ctx + 1u;
    const uint32_t z = (uint32_t)__builtin_clz(n);  // n >= 1
    const uint32_t P = K31 >> z;
    const uint32_t r0 = (uint32_t)d + P, r1 = r0 - n;
    const uint32_t r = bitop3<BT_SEL>(r0, r1, (uint32_t)((int)r1 >> 31));  // r0 mod n
    const uint32_t P2 = twice(P), right_p = P2 - n;
    const int is_short = (int)(r - right_p) >> 31;
    const uint32_t code_in = r + bitop3<BT_SEL>(P, P2 + right_p, (uint32_t)is_short);  // `1` in front of m or m + 1 bits
    const uint32_t len_in = (33u - z) + (uint32_t)is_short;                             // m + 1 or m + 2
    // Rice
    const uint32_t q = val >> k;
    const uint32_t ones = K7F >> (31u - q);                                             // q ones (q <= 31)
    const uint32_t head = bitop3<BT_OR_ANDN>(ones, ones + 1u, (uint32_t)not_above);    // `0` / `1` (above) in front of them
    const uint32_t rice = bitop3<BT_OR_ANDN>(head << (k + 1u), val, ~0u << k);          // then `0` and the k low bits of val
    const uint32_t len_rice = q + k + 3u;
    const uint32_t in_range = bitop3<BT_ANDN>((uint32_t)not_above, (uint32_t)below, 0u);
    const uint32_t code = bitop3<BT_SEL>(code_in, rice, in_range);
    PixelCode pc;
    pc.len = bitop3<BT_SEL>(len_in, len_rice, in_range);
    pc.c32 = code << (32u - pc.len);
    return pc;
}

// The samples a thread's 16-pixel group needs on the branch-free path, straight from memory into registers (the group, the
// span one row above it, the sample in front of it): coalesced 16-byte loads, a wave reads 1 KB of a row.  The loads are
// issued at the top of the kernel and first used after the gather of k: their latency hides behind it.
template <typename T>
struct GroupSamples {
    static constexpr uint32_t NW = PACK_PER_THREAD * sizeof(T) / 4;  // dwords holding 16 samples
    uint32_t cw[NW], uw[NW];
    int before;
};
template <typename T>
__device__ __forceinline__ void load_group(const T *__restrict__ pl, uint32_t first, uint32_t W, GroupSamples<T> &g) {
    __builtin_memcpy(g.cw, pl + first, PACK_PER_THREAD * sizeof(T));      // (unaligned when W or the plane's base is odd: the hardware takes it)
    __builtin_memcpy(g.uw, pl + first - W, PACK_PER_THREAD * sizeof(T));
    g.before = (int)pl[first - 1];
}

// The codes of a thread's 16 pixels WITHOUT a branch (the common case): every pixel below the first image row, the group
// inside the plane.  Same codes as classify + put_pixel, which stay as the general path (first row, the plane's first
// two samples and its ragged end, codes longer than 32 bits, images narrower than 16 pixels).
//   * neighbours: left and above (misc.rs:6-24, interior case); the left neighbour of pixel j is pixel j - 1 of the group.
//   * a first-column pixel (j0) takes above and two rows up (above-right in row 1) instead; the pair is unordered (H = max,
//     L = min), so that rule only replaces the LEFT sample of that one pixel by `special`, which the caller fetched.  One
//     thread in 240 has such a pixel: its code is built a second time where a wave holds such a thread.
// kq = the tile's k bytes in LDS.  Returns the total length in bits (exact also when a code is longer than 32 bits: only
// that code's c32 is garbage then); longest = the longest code's length.
template <typename T>
__device__ __forceinline__ uint32_t group_codes(const GroupSamples<T> &g, const uint8_t *kq, const T *__restrict__ pl, uint32_t first,
                                                uint32_t W, uint32_t j0, int special, uint32_t (&c32)[PACK_PER_THREAD],
                                                uint32_t (&len)[PACK_PER_THREAD], uint32_t &longest) {
    const uint32_t off = threadIdx.x * PACK_PER_THREAD;
    uint32_t kw[4];
    {
        const uint4 c = *reinterpret_cast<const uint4 *>(kq + off);
        kw[0] = c.x; kw[1] = c.y; kw[2] = c.z; kw[3] = c.w;
    }
    const uint32_t K31 = vgpr_const(0x80000000u), K7F = vgpr_const(0x7FFFFFFFu);
    uint32_t left = field_of(g.before, T());  // the sample in front of the group
    longest = 0;
#pragma unroll
    for (uint32_t j = 0; j < PACK_PER_THREAD; j++) {
        const uint32_t p = field_at(g.cw, j, T());
        const PixelCode pc = code_pixel(p, left, field_at(g.uw, j, T()), field_at(kw, j, uint8_t()), K31, K7F);
        c32[j] = pc.c32;
        len[j] = pc.len;
        longest = max_u16(longest, pc.len);
        left = p;
    }
    if (__ballot(j0 < PACK_PER_THREAD) != 0) {  // (wave-uniform: a quarter of the waves of a 4K plane)
        if (j0 < PACK_PER_THREAD) {
            const uint32_t i0 = first + j0;
            const PixelCode pc = code_pixel(field_of((int)pl[i0], T()), field_of(special, T()), field_of((int)pl[i0 - W], T()),
                                            (uint32_t)kq[off + j0], K31, K7F);
#pragma unroll
            for (uint32_t j = 0; j < PACK_PER_THREAD; j++) {
                c32[j] = j == j0 ? pc.c32 : c32[j];
                len[j] = j == j0 ? pc.len : len[j];
            }
            longest = max_u16(longest, pc.len);
        }
    }
    uint32_t total = 0;
#pragma unroll
    for (uint32_t j = 0; j < PACK_PER_THREAD; j++) total += len[j];
    return total;
}

// The general path of a 16-pixel group (first image row, the plane's first two samples and its ragged end, codes longer than
// 32 bits, images narrower than 16 pixels): the reference's loop as it stands -- neighbour rule (classify: misc.rs:6-24), code
// lengths / codes (code_length / put_pixel) -- pixel by pixel from global memory, once to count the bits and later once more
// to build the codes straight into the tile's bit window.  Functions of their own, not inlined: inside the kernel their
// address arithmetic was hoisted in front of the branch and their registers pushed the common path's sixteen codes into
// scratch memory.  (kq arrives as a generic pointer into LDS; these paths are rare.)
struct GeneralGroup {
    uint32_t tile_first, first, end, W, H, npix, color, depth, has_header;
};
template <typename T, typename FR, typename F>
__device__ __forceinline__ void walk_group_global(const T *__restrict__ pl, const uint8_t *kq, const GeneralGroup &g, FR &&raw, F &&f) {
    Coord xy;
    xy.set(g.first, g.W);
    for (uint32_t i = g.first; i < min(g.end, g.first + PACK_PER_THREAD); i++) {
        if (i < 2)
            raw(i, (uint32_t)(int)pl[i]);  // stored as 32-bit values (compression.rs:105-106)
        else
            f(classify(pl, i, xy.x, xy.y, g.W), (uint32_t)kq[i - g.tile_first]);
        xy.advance(1, g.W);
    }
}
template <typename T>
__device__ __noinline__ uint32_t general_group_bits(const uint8_t *kq, const T *pl, const GeneralGroup g) {
    uint32_t bits = 0;
    if (g.first < g.end) {
        const uint32_t npix = g.npix;
        if (g.has_header) bits += 8u * 14u;
        walk_group_global(pl, kq, g, [&](uint32_t, uint32_t) { bits += npix == 1 ? 64u : 32u; },
                          [&](const PixelClass &pc, uint32_t k) { bits += code_length(pc, k); });
    }
    return bits;
}
// (the window's word 0 is stream word win_word0; bit 0 of this group is stream bit my_lo)
template <typename T>
__device__ __noinline__ void general_group_place(const uint8_t *kq, const T *pl, const GeneralGroup g, uint32_t *win,
                                                 uint32_t win_words, uint64_t win_word0, uint64_t my_lo) {
    LaneBits bw;
    bw.win = win;
    bw.win_words = win_words;
    bw.win_word0 = win_word0;
    bw.begin(my_lo);
    if (g.has_header) {  // write_header, format.rs:51-61
        bw.put(0x464C4353u, 32);  // "FLCS"
        bw.put((g.color << 8) | g.depth, 16);
        bw.put(g.W, 32);
        bw.put(g.H, 32);
    }
    const uint32_t npix = g.npix;
    walk_group_global(pl, kq, g,
                      [&](uint32_t, uint32_t rv) {
                          bw.put(rv, 32);  // write_signed(32, p): sign-extended sample
                          if (npix == 1) bw.put(0u, 32);
                      },
                      [&](const PixelClass &pc, uint32_t k) { put_pixel(bw, pc, k); });
    bw.finish();
}

// this workgroup's (tile offset in the launch, plane)
__device__ __forceinline__ void take_ticket(const FusedArgs &fa, FusedLDS &fl, uint32_t &x, uint32_t &plane) {
    uint32_t t;
    if (fa.ticket) {
        if (threadIdx.x == 0) fl.ticket_sh = atomicAdd(fa.ticket, 1u);
        __syncthreads();
        t = fl.ticket_sh;
    } else {
        // No counter (launch_pack_g with a null ticket: one-dimensional grid): the workgroup index, in the same (tile, plane)
        // order.  Only for a pack kernel that has the look-back to itself (the lanes share the tail stream): it relies on
        // workgroups being started in index order; a look-back that waits in vain still gives up and reports through `error`
        // (the context then switches to tickets: felics_api.cpp, note_lookback_failure).
        t = blockIdx.x;
    }
    x = t / fa.nplanes;
    plane = t - x * fa.nplanes;
}

// The tile's offset in its plane: decoupled look-back by wave 0 (see the comment above).  Publishes the tile's inclusive
// prefix, leaves the exclusive one in fl.tile_lo_sh (far beyond any slot if the wait was given up: every store of the tile is
// then dropped) and, for the last tile of a plane, the plane's size.  The tile's AGGREGATE has been published before.
__device__ __forceinline__ void look_back(const FusedArgs &fa, FusedLDS &fl, uint32_t tile, uint32_t plane, uint32_t tile_total) {
    const uint32_t lane = lane_id(), epoch = fa.epoch, ntiles = fa.ntiles;
    uint64_t *status = fa.status;
    uint64_t excl = 0;
    int64_t look = (int64_t)tile - 1;  // tile examined by lane 0
    uint32_t spins = 0;
    bool failed = false;
    while (look >= 0) {
        const int64_t idx = look - (int64_t)lane;
        uint32_t state = ST_PREFIX;  // in front of tile 0: prefix 0
        uint64_t value = 0;
        if (idx >= 0) {
            const uint64_t sw = __hip_atomic_load(status + (uint64_t)plane * ntiles + (uint64_t)idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t tag = (uint32_t)(sw >> ST_VALUE_BITS);
            state = (tag >> 2) == (epoch & ST_EPOCH_MASK) ? (tag & 3u) : 0u;
            value = sw & ((1ull << ST_VALUE_BITS) - 1ull);
        }
        const uint64_t pm = __ballot(state == ST_PREFIX), vm = __ballot(state != 0);
        const uint32_t fp = pm ? (uint32_t)__builtin_ctzll(pm) : 64u;  // nearest tile that knows its prefix
        const uint64_t need = fp >= 63u ? ~0ull : ((2ull << fp) - 1ull);  // lanes 0..fp must have published
        if ((vm & need) != need) {
            if (++spins > LOOKBACK_SPIN_LIMIT) {
                failed = true;
                break;
            }
            __builtin_amdgcn_s_sleep(2);
            continue;
        }
        // aggregates of the lanes in front of fp (tile totals, < 2^22 each) and the prefix at fp
        const uint32_t agg = wave_incl_scan(lane < fp ? (uint32_t)value : 0u);
        excl += readlane(agg, 63);
        if (fp < 64u) {
            excl += ((uint64_t)readlane((uint32_t)(value >> 32), fp) << 32) | readlane((uint32_t)value, fp);
            break;
        }
        look -= 64;
    }
    if (failed) {
        if (lane == 0) atomicOr(fa.error, 1u);
        excl = ~0ull >> 8;
    }
    if (lane == 0) {
        const uint64_t incl = failed ? 0ull : excl + tile_total;
        __hip_atomic_store(status + (uint64_t)plane * ntiles + tile, status_word(epoch, ST_PREFIX, incl), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        fl.tile_lo_sh = excl;
        if (!failed) {
            fa.tile_bitoff[(uint64_t)plane * ntiles + tile] = excl;
            fa.tile_bits[(uint64_t)plane * ntiles + tile] = tile_total;
            if (tile + 1 == ntiles) {
                fa.plane_carry[plane] = incl;
                if (plane % fa.po.planes_per_image != 0 && incl > fa.po.plane_slot * 8u) atomicOr(fa.error, 2u);  // the plane outgrew its scratch slot
            }
        }
    }
}

// The single-pass pack of ONE tile by a workgroup (the body of k_pack_g): see the comment above.
// gsm = this thread's samples (valid where gg.fast); kq = the tile's k bytes in LDS and fl.win all zero, with a barrier behind both.
template <typename T>
__device__ __forceinline__ void pack_tile_fused(const GroupSamples<T> &gsm, const uint8_t *kq, FusedLDS &fl, const T *__restrict__ planes,
                                                const FusedArgs &fa, uint32_t tile, uint32_t plane, const GroupGeom &gg) {
    uint32_t (&win)[FUSED_WIN_WORDS + 2] = fl.win;
    uint32_t (&wsum)[PACK_THREADS / 64] = fl.wsum;
    const PlaneOut &po = fa.po;
    const uint32_t W = fa.W, H = fa.H, npix = fa.npix, ntiles = fa.ntiles;
    const bool first_plane = plane % po.planes_per_image == 0;
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
    const T *pl = planes + (uint64_t)plane * npix;
    const uint32_t tile_first = tile * PACK_TILE;
    const uint32_t first = tile_first + threadIdx.x * PACK_PER_THREAD;
    const uint32_t end = min(tile_first + PACK_TILE, npix);
    const bool has_header = tile == 0 && threadIdx.x == 0 && first_plane;
    PSTAMP(4);

    // ---- phase A: this thread's codes and their total length
    // The branch-free form where it applies: the whole group below the first image row and inside the plane, at most one
    // first-column pixel in it (whose second neighbour -- two rows up, or above-right in row 1 -- comes from global memory).
    // (The general path is two function calls, placed where none of the common path's codes is in a register: the count in
    // front of group_codes, the placement behind the common path's.)
    const GeneralGroup general{tile_first, first, end, W, H, npix, fa.color, fa.depth, has_header ? 1u : 0u};
    uint32_t bits = 0;
    if (!gg.fast) bits = general_group_bits<T>(kq, pl, general);  // count now, build the codes straight into the window later
    uint32_t c32[PACK_PER_THREAD], len[PACK_PER_THREAD];
    bool in_registers = false;
    if (gg.fast) {
        uint32_t longest;
        bits = group_codes<T>(gsm, kq, pl, first, W, gg.j0, gg.special, c32, len, longest);
        in_registers = longest <= 32u;  // (a longer code: the lengths stand, the codes are built again the general way)
    }
    const uint32_t inc = wave_incl_scan(bits);
    if (lane == 63) wsum[wave] = inc;
    PSTAMP(5);
    __syncthreads();
    PSTAMP(6);
    uint32_t woff = 0, tile_total = 0;
    for (uint32_t w = 0; w < PACK_THREADS / 64; w++) {
        if (w < wave) woff += wsum[w];
        tile_total += wsum[w];
    }
    if (threadIdx.x == 0)
        __hip_atomic_store(fa.status + (uint64_t)plane * ntiles + tile, status_word(fa.epoch, ST_AGGREGATE, tile_total), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t my_rel = woff + inc - bits;  // this thread's first bit, from the tile's first bit

    // The output words of the tile: word r (from the stream word the tile starts in) = window bits 32 r - s .. 32 r - s + 31,
    // s = the tile's bit position in that word, known after the look-back.  Plain stores for the words the tile has to itself;
    // its first and last word, where they are shared with its neighbours, go to the edge arrays (k_join_edges).
    auto flush_window = [&](uint32_t wb, uint32_t s, uint64_t tile_lo) {
        const uint64_t tile_hi = tile_lo + tile_total;
        uint64_t limit_words;  // a stream that outgrows its slot is cut (the host re-packs)
        uint32_t *out_words = plane_words(po, plane, limit_words);
        const uint64_t first_word = tile_lo >> 5, last_word = (tile_hi - 1) >> 5;
        const bool first_shared = (tile_lo & 31u) != 0, last_shared = (tile_hi & 31u) != 0;
        const uint32_t nwords = (uint32_t)(last_word - first_word) + 1u;  // words the tile touches
        uint32_t *out_rel = out_words + first_word;                         // (only dereferenced below limit_words)
        const uint32_t limit_rel = limit_words > first_word ? (uint32_t)std::min<uint64_t>(limit_words - first_word, 0xFFFFFFFFull) : 0u;
        const uint32_t lo = first_shared ? 1u : 0u;
        const uint32_t hi = min(nwords - (last_shared ? 1u : 0u), limit_rel);
        const uint32_t span = hi > lo ? hi - lo : 0u;
        auto word = [&](uint32_t j) { return __builtin_amdgcn_alignbit(j ? win[j - 1] : 0u, win[j], s); };  // (s = 0: win[j])
#pragma unroll
        for (uint32_t u = 0; u < FUSED_WIN_WORDS / PACK_THREADS; u++) {
            const uint32_t j = threadIdx.x + u * PACK_THREADS, r = wb + j;  // word r of the tile
            if (r - lo < span) out_rel[r] = __builtin_bswap32(word(j));
        }
        if (threadIdx.x == 0) {
            if (wb == 0 && first_shared) fa.edge_first[(uint64_t)plane * ntiles + tile] = word(0);  // merged with the previous tile's last word later
            const uint32_t rl = nwords - 1u;
            if (last_shared && !(rl == 0 && first_shared) && rl >= wb && rl - wb < FUSED_WIN_WORDS) fa.edge_last[(uint64_t)plane * ntiles + tile] = word(rl - wb);
        }
    };
    if (tile_total <= (FUSED_WIN_WORDS - 1u) * 32u) {
        // ---- phase B, the common case: the whole tile in one window, placed from the tile's first bit -- which needs nothing
        // from other tiles, so the look-back comes behind it, when the tiles in front have long published.  Code j at bit `at`:
        // its upper part into word at >> 5, what is left of it into the next word (zero if the code ends in the first one; an
        // LDS OR of zero is cheaper than a branch around it).
        if (in_registers) {
            uint32_t at = my_rel;
            char *wbytes = reinterpret_cast<char *>(win);
#pragma unroll
            for (uint32_t j = 0; j < PACK_PER_THREAD; j++) {
                const uint32_t hi = c32[j] >> (at & 31u), lo = __builtin_amdgcn_alignbit(c32[j], 0u, at & 31u);
                uint32_t *w2 = reinterpret_cast<uint32_t *>(wbytes + ((at >> 3) & ~3u));
                atomicOr(w2, hi);
                atomicOr(w2 + 1, lo);
                at += len[j];
            }
        }
        if (!in_registers && bits != 0) general_group_place<T>(kq, pl, general, win, FUSED_WIN_WORDS, 0, my_rel);
        PSTAMP(8);
        if (wave == 0) look_back(fa, fl, tile, plane, tile_total);
        __syncthreads();
        PSTAMP(7);
        if (tile_total == 0) return;
        const uint64_t tile_lo = fl.tile_lo_sh;
        flush_window(0, (uint32_t)(tile_lo & 31u), tile_lo);
        PSTAMP(1);
    } else {
        // more than 16 bits per pixel (no image content does that; a tile of the first rows of a noisy 16 x N image can): several
        // windows at their final alignment, every thread builds its codes again, the general way, into each window its bits touch
        if (wave == 0) look_back(fa, fl, tile, plane, tile_total);
        __syncthreads();
        const uint64_t tile_lo = fl.tile_lo_sh, my_lo = tile_lo + my_rel;
        const uint64_t first_word = tile_lo >> 5;
        const uint32_t nwords = (uint32_t)(((tile_lo + tile_total - 1) >> 5) - first_word) + 1u;
        const uint32_t my_first = (uint32_t)((my_lo >> 5) - first_word), my_last = (uint32_t)(((my_lo + bits - 1) >> 5) - first_word);
#pragma nounroll
        for (uint32_t wb = 0; wb < nwords; wb += FUSED_WIN_WORDS) {
            if (wb != 0) {  // (the first window arrives cleared)
                __syncthreads();
                for (uint32_t j = threadIdx.x; j < FUSED_WIN_WORDS + 2; j += PACK_THREADS) win[j] = 0;
                __syncthreads();
            }
            if (bits != 0 && my_last >= wb && my_first < wb + FUSED_WIN_WORDS)
                general_group_place<T>(kq, pl, general, win, FUSED_WIN_WORDS, first_word + wb, my_lo);
            __syncthreads();
            flush_window(wb, 0u, tile_lo);
        }
    }
#ifdef FELICS_PACK_STAMPS
    if (threadIdx.x == 0) {
        unsigned long long *slot = g_pack_stamps[(tile * 7u + plane) & 255u];
        for (int i = 0; i < 12; i++) atomicAdd(&slot[i], fl.t_acc[i]);
        atomicAdd(&slot[15], 1ull);
    }
#endif
}

#ifdef FELICS_PACK_STAMPS
extern "C" __attribute__((visibility("default"))) int felics_debug_pack_stamps(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pack_stamps), sizeof(g_pack_stamps)) != hipSuccess) return -1;
    if (reset) {
        static unsigned long long z[256 * 16] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_pack_stamps), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

// ------------------------------------------------------------------------------------------
// pack with k gathered from chain order (k_pack_g): a workgroup takes one tile (sort tile = pack tile).  k of every event
// lies in k_sorted[slot] (k_assign_serial); the events of a sort tile in one context are one run of slots of that context's
// chain (tile_off[t][c] .. tile_off[t + 1][c]), and pix_of[slot] is the event's pixel as an offset into the tile.  The
// workgroup reads its runs -- coalesced: a run is contiguous -- and drops k into the LDS array the pack stage indexes by
// pixel.  Nothing else of the estimator is left in this kernel.
// Memory round trips of a tile, in order: {the run table, the thread's own pixels, straight into registers} -> {k and pixel
// offsets of the runs} -> (codes, placement) -> {the status words of the tiles in front}.  Rounds 1-3 staged the pixels and
// the run table through LDS behind a barrier of their own and waited for the look-back in front of the placement.
// ------------------------------------------------------------------------------------------

struct GSources {
    const uint8_t *k_sorted;
    const uint16_t *pix_of;  // offset of the event's pixel in its sort tile
    const uint32_t *tile_off, *chain_base, *chain_len;
    uint32_t sort_ntiles;
};

template <typename T>
__attribute__((amdgpu_waves_per_eu(sizeof(T) == 1 ? 7 : 6))) __global__ __launch_bounds__(PACK_THREADS) void k_pack_g(const T *__restrict__ planes, GSources gs, FusedArgs fa,
                                                                                               uint32_t sort_tile_begin, uint32_t pack_tile_end) {
    __shared__ alignas(16) uint8_t kq[PACK_TILE];  // k of pixel tile_first + j (event pixels only: the others hold what was there)
    __shared__ FusedLDS fl;
    static_assert(SORT_TILE == PACK_TILE, "one workgroup = one sort tile = one pack tile (one look-back per workgroup)");
    uint32_t x, plane;
#ifdef FELICS_PACK_STAMPS
    if (threadIdx.x == 0) {
        for (int i = 0; i < 12; i++) fl.t_acc[i] = 0;
        fl.t_last = __builtin_amdgcn_s_memtime();
    }
#endif
    take_ticket(fa, fl, x, plane);
    PSTAMP(0);
    const uint32_t st = sort_tile_begin + x;
    const uint32_t lane = lane_id();
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    constexpr uint32_t NWV = PACK_THREADS / 64;
    constexpr uint32_t NC = nctx_of<T>();
    constexpr uint32_t CPT = NC / PACK_THREADS;  // contexts per thread
    static_assert(NC == CPT * PACK_THREADS, "every thread takes CPT contexts");
    const T *pl = planes + (uint64_t)plane * fa.npix;
    // ---- round trip 1: the runs of this thread's contexts and its pixels.  Wave w takes contexts w, w + 4, w + 8, ... (lane l:
    // context 4 l + w, + 256 for the second half of a Y / Co / Cg table): the contexts that matter in a smooth frame are the
    // first ten, and this way every wave gets its share of them.
    uint32_t run_a[CPT], run_n[CPT];
    {
        const uint32_t *off0 = gs.tile_off + ((uint64_t)plane * gs.sort_ntiles + st) * NC;
        const bool last_tile = st + 1 == gs.sort_ntiles;
        const uint32_t *off1 = last_tile ? gs.chain_len + (uint64_t)plane * NC : off0 + NC;
        const uint32_t *cb = gs.chain_base + (uint64_t)plane * NC;
        uint32_t ra[CPT], rb[CPT], rbase[CPT];
#pragma unroll
        for (uint32_t h = 0; h < CPT; h++) {
            const uint32_t c = h * PACK_THREADS + lane * NWV + wave;
            ra[h] = off0[c];
            rb[h] = off1[c];
            rbase[h] = cb[c];
        }
#pragma unroll
        for (uint32_t h = 0; h < CPT; h++) {
            run_a[h] = ra[h] + rbase[h];  // first slot of the tile's run in context c
            run_n[h] = rb[h] - ra[h];     // its events
        }
    }
    const GroupGeom gg = group_geometry<T>(pl, st, fa.W, fa.npix);
    GroupSamples<T> gsm;
    if (gg.fast) load_group(pl, st * PACK_TILE + threadIdx.x * PACK_PER_THREAD, fa.W, gsm);
    for (uint32_t j = threadIdx.x; j < FUSED_WIN_WORDS + 2; j += PACK_THREADS) fl.win[j] = 0;  // the bit window (barrier: behind the gather)
    PSTAMP(9);
    // ---- round trip 2: k and pixel offsets through the runs.  Runs of up to GATHER_SHORT events are read one run per lane,
    // all of them at once; longer runs 64 events per wave-load, GATHER_CHUNKS such loads (of any of the wave's runs) in flight
    // together.
    constexpr uint32_t GATHER_SHORT = 8, GATHER_CHUNKS = 8;
#pragma unroll
    for (uint32_t h = 0; h < CPT; h++) {
        const uint32_t a = run_a[h], n = run_n[h];
        // (Loads and LDS stores are not predicated lane by lane: a lane past the end of its run / chunk takes the last element
        // again -- the same k goes to the same pixel twice -- so a whole group of loads runs under ONE condition; lane-wise
        // predicates were four scalar instructions per load, 140 M scalar instructions per step in this stage.)
        const bool is_short = n != 0 && n <= GATHER_SHORT;
        if (is_short) {  // short runs: lane = run
            uint32_t kv[GATHER_SHORT], pv[GATHER_SHORT];
#pragma unroll
            for (uint32_t u = 0; u < GATHER_SHORT; u++) {
                const uint64_t at = (uint64_t)a + min(u, n - 1u);
                kv[u] = gs.k_sorted[at];
                pv[u] = gs.pix_of[at];
            }
#pragma unroll
            for (uint32_t u = 0; u < GATHER_SHORT; u++) kq[pv[u]] = (uint8_t)kv[u];
        }
        uint64_t longs = __ballot(n > GATHER_SHORT);
        uint32_t off = 0;  // events of the first run of `longs` already taken
        while (longs) {
            uint32_t kv[GATHER_CHUNKS], pv[GATHER_CHUNKS], live = 0;
#pragma unroll
            for (uint32_t u = 0; u < GATHER_CHUNKS; u++) {  // (wave-uniform: scalar registers and scalar branches)
                if (longs != 0) {
                    const uint32_t b = (uint32_t)__builtin_ctzll(longs);
                    const uint32_t A = readlane(a, b), N = readlane(n, b);
                    const uint64_t at = (uint64_t)A + min(off + lane, N - 1u);
                    kv[u] = gs.k_sorted[at];
                    pv[u] = gs.pix_of[at];
                    live = u + 1;
                    off += 64;
                    if (off >= N) {
                        longs &= longs - 1;
                        off = 0;
                    }
                }
            }
#pragma unroll
            for (uint32_t u = 0; u < GATHER_CHUNKS; u++)
                if (u < live) kq[pv[u]] = (uint8_t)kv[u];
        }
    }
    PSTAMP(2);
    __syncthreads();
    PSTAMP(3);
    if (st < pack_tile_end) pack_tile_fused<T>(gsm, kq, fl, planes, fa, st, plane, gg);
}

// ------------------------------------------------------------------------------------------
// k_pack_t (round 5): the single-pass pack on the tile-local layout.  The tile's k bytes lie where the front kernel put the
// tile's events -- kq[slot], pix[slot] = the event's pixel, slots [0, tile_slots) of the tile -- so the gather is one
// contiguous read: four slots per thread and round, k dropped into the LDS array the pack stage indexes by pixel (padding
// slots, pix = 0xFFFF, into a dump byte behind it).  No run table, no chains in this kernel.
// ------------------------------------------------------------------------------------------
struct TSources {
    const uint8_t *kq;
    const uint16_t *pix;
    const uint32_t *tile_slots;
    uint32_t cap, sort_ntiles;
};

template <typename T>
__attribute__((amdgpu_waves_per_eu(sizeof(T) == 1 ? 7 : 6))) __global__ __launch_bounds__(PACK_THREADS) void k_pack_t(const T *__restrict__ planes, TSources ts, FusedArgs fa,
                                                                                               uint32_t sort_tile_begin, uint32_t pack_tile_end) {
    __shared__ alignas(16) uint8_t kq[PACK_TILE + 16];  // k of pixel tile_first + j (event pixels only: the others hold what was there); [PACK_TILE]: dump
    __shared__ FusedLDS fl;
    static_assert(SORT_TILE == PACK_TILE, "one workgroup = one sort tile = one pack tile (one look-back per workgroup)");
    uint32_t x, plane;
#ifdef FELICS_PACK_STAMPS
    if (threadIdx.x == 0) {
        for (int i = 0; i < 12; i++) fl.t_acc[i] = 0;
        fl.t_last = __builtin_amdgcn_s_memtime();
    }
#endif
    take_ticket(fa, fl, x, plane);
    PSTAMP(0);
    const uint32_t st = sort_tile_begin + x;
    const T *pl = planes + (uint64_t)plane * fa.npix;
    // ---- round trip 1: the tile's slots in use and the thread's pixels
    const uint64_t pt = (uint64_t)plane * ts.sort_ntiles + st;
    const uint32_t ns = (uint32_t)__builtin_amdgcn_readfirstlane((int)ts.tile_slots[pt]);  // a multiple of REC
    const GroupGeom gg = group_geometry<T>(pl, st, fa.W, fa.npix);
    GroupSamples<T> gsm;
    if (gg.fast) load_group(pl, st * PACK_TILE + threadIdx.x * PACK_PER_THREAD, fa.W, gsm);
    for (uint32_t j = threadIdx.x; j < FUSED_WIN_WORDS + 2; j += PACK_THREADS) fl.win[j] = 0;  // the bit window (barrier: behind the gather)
    PSTAMP(9);
    // ---- round trip 2: k and pixel offsets of the tile's slots.  (A thread past the end takes the last four slots again -- the
    // same k goes to the same pixels twice -- so the loads of a round run under no lane-wise condition.)
    constexpr uint32_t GR = 3;  // rounds in flight together: 3072 slots (a 4K frame's tile has ~2400 in use)
    const uint8_t *ksrc = ts.kq + pt * ts.cap;
    const uint16_t *psrc = ts.pix + pt * ts.cap;
    for (uint32_t s0 = 0; s0 < ns; s0 += GR * 4 * PACK_THREADS) {
        uint32_t kv[GR];
        uint2 pv[GR];
#pragma unroll
        for (uint32_t u = 0; u < GR; u++) {
            const uint32_t s = min(s0 + u * 4 * PACK_THREADS + threadIdx.x * 4, ns - 4u);
            kv[u] = *reinterpret_cast<const uint32_t *>(ksrc + s);
            pv[u] = *reinterpret_cast<const uint2 *>(psrc + s);
        }
#pragma unroll
        for (uint32_t u = 0; u < GR; u++) {
            kq[min(pv[u].x & 0xFFFFu, PACK_TILE)] = (uint8_t)kv[u];
            kq[min(pv[u].x >> 16, PACK_TILE)] = (uint8_t)(kv[u] >> 8);
            kq[min(pv[u].y & 0xFFFFu, PACK_TILE)] = (uint8_t)(kv[u] >> 16);
            kq[min(pv[u].y >> 16, PACK_TILE)] = (uint8_t)(kv[u] >> 24);
        }
    }
    PSTAMP(2);
    __syncthreads();
    PSTAMP(3);
    if (st < pack_tile_end) pack_tile_fused<T>(gsm, kq, fl, planes, fa, st, plane, gg);
}

// Words shared by two tiles (and the last, partly filled word of a plane): OR of the two halves.
__global__ void k_join_edges(const uint64_t *__restrict__ tile_bitoff, const uint32_t *__restrict__ tile_bits,
                             const uint32_t *__restrict__ edge_first, const uint32_t *__restrict__ edge_last, PlaneOut po,
                             uint32_t ntiles) {
    const uint32_t tile = blockIdx.x * blockDim.x + threadIdx.x, plane = blockIdx.y;
    if (tile >= ntiles) return;
    const uint64_t at = (uint64_t)plane * ntiles + tile;
    const uint64_t lo = tile_bitoff[at], hi = lo + tile_bits[at];
    if (hi == lo) return;
    uint64_t limit_words;
    uint32_t *out_words = plane_words(po, plane, limit_words);
    const uint64_t first_word = lo >> 5, last_word = (hi - 1) >> 5;
    const bool first_shared = (lo & 31u) != 0, last_shared = (hi & 31u) != 0;
    if (first_shared && first_word < limit_words) {
        // the tile before ends inside this word; its half is its edge_last unless it lies inside the word
        // altogether (then it is the plane's tiny last tile and has no successor, i.e. cannot be `tile - 1`)
        out_words[first_word] = __builtin_bswap32(edge_first[at] | edge_last[at - 1]);
    }
    if (last_shared && tile + 1 == ntiles && !(first_shared && first_word == last_word) && last_word < limit_words)
        out_words[last_word] = __builtin_bswap32(edge_last[at]);
}

// 32 bits of a plane's own bit string (nbits long, zero beyond) starting at bit p (may be negative)
__device__ __forceinline__ uint32_t plane_bits_at(const uint32_t *__restrict__ src, uint64_t nbits, int64_t p) {
    const int64_t w0 = p >> 5;  // floor
    const uint32_t sh = (uint32_t)(p & 31);
    auto word = [&](int64_t i) -> uint32_t {
        return (i >= 0 && (uint64_t)i * 32u < nbits) ? __builtin_bswap32(src[i]) : 0u;
    };
    const uint32_t a = word(w0);
    if (sh == 0) return a;
    return (a << sh) | (word(w0 + 1) >> (32u - sh));
}

// RGB: moves planes 1.. of every image from their scratch slots to their place behind plane 0.
// plane_base[p] = bit offset of plane p in its image's stream, plane_carry[p] = its bits (k_finish_sizes).
// One thread per output word: it ORs what every plane contributes to that word (the first word also
// keeps plane 0's last bits, already in place).
__global__ __launch_bounds__(256) void k_concat_planes(const uint64_t *__restrict__ plane_base,
                                                       const uint64_t *__restrict__ plane_carry, PlaneOut po) {
    const uint32_t img = blockIdx.y, ppi = po.planes_per_image;
    const uint64_t *base = plane_base + (uint64_t)img * ppi, *bits = plane_carry + (uint64_t)img * ppi;
    const uint64_t begin_bit = base[1], end_bit = base[ppi - 1] + bits[ppi - 1];
    if (end_bit == begin_bit) return;
    uint32_t *dst = reinterpret_cast<uint32_t *>(po.out + (uint64_t)img * po.slot_stride);
    const uint64_t first_word = begin_bit >> 5, last_word = (end_bit - 1) >> 5, limit_words = po.slot_stride >> 2;
    for (uint64_t w = first_word + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w <= last_word && w < limit_words;
         w += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t v = (w == first_word && (begin_bit & 31u) != 0) ? __builtin_bswap32(dst[w]) : 0u;
        for (uint32_t c = 1; c < ppi; c++) {
            uint64_t lim;
            const uint32_t *src = plane_words(po, img * ppi + c, lim);
            v |= plane_bits_at(src, bits[c], (int64_t)(w * 32u) - (int64_t)base[c]);
        }
        dst[w] = __builtin_bswap32(v);
    }
}

// ------------------------------------------------------------------------------------------
// launchers (host side of this translation unit)
// ------------------------------------------------------------------------------------------

static inline uint32_t cdiv(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

void launch_rgb8_to_planes(hipStream_t s, const uint8_t *rgb, int16_t *planes, uint32_t npix, uint32_t nimg) {
    uint64_t total = (uint64_t)((npix + 3) / 4) * nimg;  // four pixels per thread
    uint32_t blocks = (uint32_t)std::min<uint64_t>((total + 255) / 256, 256u * 32u);
    if (blocks == 0) return;
    FELICS_LAUNCH(k_rgb8_to_planes, dim3(blocks), dim3(256), s, rgb, planes, npix, nimg);
}

template <typename T>
void launch_hist(hipStream_t s, const T *planes, uint32_t *counts, const Geometry &g) {
    FELICS_LAUNCH((k_hist<T>), dim3(cdiv(g.sort_tiles, 4), g.nplanes), dim3(256), s, planes, counts, g.W,
                       g.npix, g.sort_tiles);
}
template void launch_hist<uint8_t>(hipStream_t, const uint8_t *, uint32_t *, const Geometry &);
template void launch_hist<int16_t>(hipStream_t, const int16_t *, uint32_t *, const Geometry &);

void launch_offsets(hipStream_t s, uint32_t *counts, uint32_t *chain_len, uint32_t *chain_base,
                    uint32_t *total_events, const Geometry &g) {
    const uint32_t nchains = g.nplanes * g.nctx;
    FELICS_LAUNCH(k_tile_offsets, dim3(g.nctx / OFF_CTX, g.nplanes), dim3(OFF_SEGS * OFF_CTX), s, counts, chain_len,
                       g.sort_tiles, g.nctx);
    FELICS_LAUNCH(k_chain_bases, dim3(1), dim3(CHAIN_BASES_THREADS), s, chain_len, chain_base, nchains, total_events);
}

template <typename T, typename ET>
void launch_scatter(hipStream_t s, const T *planes, const uint32_t *tile_off, const uint32_t *chain_base,
                    ET *sorted_e, uint32_t *pix_of, bool in_tile_offsets, const Geometry &g, uint32_t tile_begin, uint32_t tile_end,
                    uint32_t *order_flag, bool by_ballot, bool test_violation) {
    if (tile_end <= tile_begin) return;
    // the planes are dealt to the XCDs by the kernels (see there)
    if (by_ballot) {  // one workgroup per four tiles of a plane
        const dim3 grid(8u * cdiv(g.nplanes, 8) * cdiv(tile_end - tile_begin, 4));
        if (in_tile_offsets)
            FELICS_LAUNCH((k_scatter_ballot<T, ET, true>), grid, dim3(256), s, planes, tile_off, chain_base, sorted_e, pix_of, g.W,
                          g.npix, g.sort_tiles, tile_begin, tile_end, g.nplanes);
        else
            FELICS_LAUNCH((k_scatter_ballot<T, ET, false>), grid, dim3(256), s, planes, tile_off, chain_base, sorted_e, pix_of, g.W,
                          g.npix, g.sort_tiles, tile_begin, tile_end, g.nplanes);
        return;
    }
    const dim3 grid(8u * cdiv(g.nplanes, 8) * (tile_end - tile_begin));  // one workgroup per tile
    if (in_tile_offsets)
        FELICS_LAUNCH((k_scatter<T, ET, true>), grid, dim3(256), s, planes, tile_off, chain_base, sorted_e, pix_of, g.W, g.npix,
                      g.sort_tiles, tile_begin, tile_end, g.nplanes, order_flag, test_violation ? 1u : 0u);
    else
        FELICS_LAUNCH((k_scatter<T, ET, false>), grid, dim3(256), s, planes, tile_off, chain_base, sorted_e, pix_of, g.W, g.npix,
                      g.sort_tiles, tile_begin, tile_end, g.nplanes, order_flag, test_violation ? 1u : 0u);
}
template void launch_scatter<uint8_t, uint8_t>(hipStream_t, const uint8_t *, const uint32_t *, const uint32_t *,
                                               uint8_t *, uint32_t *, bool, const Geometry &, uint32_t, uint32_t, uint32_t *, bool, bool);
template void launch_scatter<int16_t, uint16_t>(hipStream_t, const int16_t *, const uint32_t *, const uint32_t *,
                                                uint16_t *, uint32_t *, bool, const Geometry &, uint32_t, uint32_t, uint32_t *, bool, bool);

template <typename ET>
void launch_zero_padding(hipStream_t s, ET *sorted_e, uint32_t *pix_of, const uint32_t *chain_base,
                         const uint32_t *chain_len, const Geometry &g) {
    const uint32_t nchains = g.nplanes * g.nctx;
    FELICS_LAUNCH((k_zero_padding<ET>), dim3(cdiv(nchains, 256)), dim3(256), s, sorted_e, pix_of, chain_base,
                       chain_len, nchains);
}
template void launch_zero_padding<uint8_t>(hipStream_t, uint8_t *, uint32_t *, const uint32_t *, const uint32_t *,
                                           const Geometry &);
template void launch_zero_padding<uint16_t>(hipStream_t, uint16_t *, uint32_t *, const uint32_t *, const uint32_t *,
                                            const Geometry &);

template <typename ET>
void launch_spine(hipStream_t s, const ET *sorted_e, uint32_t *block_state, const uint32_t *chain_base,
                  const uint32_t *chain_len, const uint32_t *tile_off, uint32_t t_end, uint32_t *chain_prog,
                  uint32_t *block_tag, uint32_t *partial, uint32_t epoch, uint32_t slice, const Geometry &g) {
    const uint32_t nchains = g.nplanes * g.nctx;
    uint2 *part = reinterpret_cast<uint2 *>(partial) + (uint64_t)(slice - 1) * nchains;
    const uint32_t stamp = (epoch << TAG_SLICE_BITS) | slice;
    if (g.nplanes >= SP_MANY_PLANES)
        FELICS_LAUNCH((k_spine2<ET, SP_BATCH_MANY_PLANES>), dim3(nchains), dim3(64 * (1 + SP_HELPERS)), s, sorted_e, block_state, chain_base,
                      chain_len, nchains, tile_off, g.sort_tiles, t_end, chain_prog, block_tag, part, stamp);
    else
        FELICS_LAUNCH((k_spine2<ET, SP_BATCH_FEW_PLANES>), dim3(nchains), dim3(64 * (1 + SP_HELPERS)), s, sorted_e, block_state, chain_base,
                      chain_len, nchains, tile_off, g.sort_tiles, t_end, chain_prog, block_tag, part, stamp);
}
template void launch_spine<uint8_t>(hipStream_t, const uint8_t *, uint32_t *, const uint32_t *, const uint32_t *,
                                    const uint32_t *, uint32_t, uint32_t *, uint32_t *, uint32_t *, uint32_t, uint32_t,
                                    const Geometry &);
template void launch_spine<uint16_t>(hipStream_t, const uint16_t *, uint32_t *, const uint32_t *, const uint32_t *,
                                     const uint32_t *, uint32_t, uint32_t *, uint32_t *, uint32_t *, uint32_t, uint32_t,
                                     const Geometry &);

template <typename T>
void launch_lengths(hipStream_t s, const T *planes, const uint8_t *k_map, group_bits_t<T> *group_bits,
                    uint32_t *tile_bits, const Geometry &g, uint32_t t0, uint32_t t1) {
    if (t1 <= t0) return;
    FELICS_LAUNCH((k_lengths<T>), dim3(t1 - t0, g.nplanes), dim3(PACK_THREADS), s, planes, k_map, group_bits,
                       tile_bits, g.W, g.npix, g.pack_tiles, g.planes_per_image, t0);
}
template void launch_lengths<uint8_t>(hipStream_t, const uint8_t *, const uint8_t *, uint16_t *, uint32_t *,
                                      const Geometry &, uint32_t, uint32_t);
template void launch_lengths<int16_t>(hipStream_t, const int16_t *, const uint8_t *, uint16_t *, uint32_t *,
                                      const Geometry &, uint32_t, uint32_t);
template void launch_lengths<uint16_t>(hipStream_t, const uint16_t *, const uint8_t *, uint32_t *, uint32_t *,
                                       const Geometry &, uint32_t, uint32_t);
template void launch_lengths<int32_t>(hipStream_t, const int32_t *, const uint8_t *, uint32_t *, uint32_t *,
                                      const Geometry &, uint32_t, uint32_t);

void launch_bitscan_slice(hipStream_t s, const uint32_t *tile_bits, uint64_t *tile_bitoff, uint64_t *plane_carry,
                          const Geometry &g, uint32_t t0, uint32_t t1) {
    if (t1 <= t0) return;
    FELICS_LAUNCH(k_bitscan_slice, dim3(g.nplanes), dim3(1024), s, tile_bits, tile_bitoff, plane_carry,
                       g.pack_tiles, t0, t1);
}

void launch_finish_sizes(hipStream_t s, const uint64_t *plane_carry, uint64_t *plane_base, uint64_t *image_bytes,
                         const Geometry &g) {
    FELICS_LAUNCH(k_finish_sizes, dim3(cdiv(g.nimages, 64)), dim3(64), s, plane_carry, plane_base, image_bytes,
                       g.nimages, g.planes_per_image);
}

void launch_place_streams(hipStream_t s, const uint64_t *image_bytes, uint64_t *image_off, const Geometry &g) {
    FELICS_LAUNCH(k_place_streams, dim3(1), dim3(64), s, image_bytes, image_off, g.nimages);
}

void launch_zero_streams(hipStream_t s, uint32_t *out, const uint64_t *image_off, const Geometry &g) {
    FELICS_LAUNCH(k_zero_streams, dim3(256 * 8), dim3(256), s, out, image_off, g.nimages);
}

void launch_zero_edges(hipStream_t s, uint8_t *out, const uint64_t *image_off, uint64_t slot_stride,
                       const uint64_t *tile_bitoff, const uint32_t *tile_bits, const uint64_t *plane_base,
                       const Geometry &g, uint32_t t0, uint32_t t1) {
    if (t1 <= t0) return;
    Placement pl{image_off, slot_stride};
    FELICS_LAUNCH(k_zero_edges, dim3(cdiv(t1 - t0, 256), g.nplanes), dim3(256), s, out, pl, tile_bitoff, tile_bits,
                       plane_base, g.pack_tiles, t0, t1, g.planes_per_image);
}

template <typename T>
void launch_pack(hipStream_t s, const T *planes, const uint8_t *k_map, const group_bits_t<T> *group_bits,
                 const uint64_t *tile_bitoff, const uint32_t *tile_bits, const uint64_t *plane_base,
                 const uint64_t *image_off, uint64_t slot_stride, uint8_t *out, const Geometry &g, uint32_t t0,
                 uint32_t t1) {
    if (t1 <= t0) return;
    Placement pl{image_off, slot_stride};
    FELICS_LAUNCH((k_pack<T>), dim3(t1 - t0, g.nplanes), dim3(PACK_THREADS), s, planes, k_map, group_bits,
                       tile_bitoff, tile_bits, plane_base, pl, out, g.W, g.H, g.npix, g.pack_tiles, g.planes_per_image,
                       g.color, g.depth, t0);
}
template void launch_pack<uint8_t>(hipStream_t, const uint8_t *, const uint8_t *, const uint16_t *, const uint64_t *,
                                   const uint32_t *, const uint64_t *, const uint64_t *, uint64_t, uint8_t *,
                                   const Geometry &, uint32_t, uint32_t);
template void launch_pack<int16_t>(hipStream_t, const int16_t *, const uint8_t *, const uint16_t *, const uint64_t *,
                                   const uint32_t *, const uint64_t *, const uint64_t *, uint64_t, uint8_t *,
                                   const Geometry &, uint32_t, uint32_t);
template void launch_pack<uint16_t>(hipStream_t, const uint16_t *, const uint8_t *, const uint32_t *, const uint64_t *,
                                    const uint32_t *, const uint64_t *, const uint64_t *, uint64_t, uint8_t *,
                                    const Geometry &, uint32_t, uint32_t);
template void launch_pack<int32_t>(hipStream_t, const int32_t *, const uint8_t *, const uint32_t *, const uint64_t *,
                                   const uint32_t *, const uint64_t *, const uint64_t *, uint64_t, uint8_t *,
                                   const Geometry &, uint32_t, uint32_t);

template <typename ET>
void launch_assign_serial(hipStream_t s, const ET *sorted_e, uint8_t *k_sorted, const uint32_t *block_state,
                          const uint32_t *total_slots, const uint32_t *block_tag, const uint32_t *partial, uint32_t epoch,
                          uint32_t slice, const Geometry &g) {
    const uint32_t nchains = g.nplanes * g.nctx;
    // persistent: eight workgroups of four waves per CU stride over the blocks (fewer if there cannot be that many blocks)
    const uint32_t wgs = std::min<uint32_t>(std::max(cdiv(max_event_blocks(g), 256), cdiv(nchains, 256)), 256u * 8u);
    FELICS_LAUNCH((k_assign_serial<ET>), dim3(wgs), dim3(256), s, sorted_e, block_state, k_sorted, total_slots, block_tag,
                  reinterpret_cast<const uint2 *>(partial) + (uint64_t)(slice - 1) * nchains, nchains,
                  (epoch << TAG_SLICE_BITS) | slice);
}
template void launch_assign_serial<uint8_t>(hipStream_t, const uint8_t *, uint8_t *, const uint32_t *, const uint32_t *,
                                            const uint32_t *, const uint32_t *, uint32_t, uint32_t, const Geometry &);
template void launch_assign_serial<uint16_t>(hipStream_t, const uint16_t *, uint8_t *, const uint32_t *, const uint32_t *,
                                             const uint32_t *, const uint32_t *, uint32_t, uint32_t, const Geometry &);

void launch_k_to_pixels(hipStream_t s, const uint8_t *k_sorted, const uint32_t *pix_of, uint8_t *k_map, const uint32_t *total_slots,
                         const Geometry &g) {
    const uint32_t wgs = std::min<uint32_t>(cdiv(max_event_slots(g), 256 * 8), 256u * 8u);
    FELICS_LAUNCH(k_k_to_pixels, dim3(std::max(wgs, 1u)), dim3(256), s, k_sorted, pix_of, k_map, total_slots);
}

template <typename T>
void launch_pack_g(hipStream_t s, const T *planes, const uint8_t *k_sorted, const uint32_t *pix_of, const uint32_t *tile_off,
                   const uint32_t *chain_base, const uint32_t *chain_len, uint64_t *status, uint64_t *tile_bitoff,
                   uint32_t *tile_bits, uint64_t *plane_carry, uint32_t *edge_first, uint32_t *edge_last, uint32_t *error,
                   const PackTarget &to, const Geometry &g, uint32_t st0, uint32_t st1, uint32_t epoch, uint32_t *ticket) {
    if (st1 <= st0) return;
    const FusedArgs fa{status, tile_bitoff, tile_bits, plane_carry, edge_first, edge_last, error,
                       PlaneOut{to.out, to.slot_stride, to.scratch, to.plane_slot, g.planes_per_image},
                       g.W, g.H, g.npix, g.pack_tiles, g.color, g.depth, epoch, ticket, g.nplanes};
    const GSources gs{k_sorted, reinterpret_cast<const uint16_t *>(pix_of), tile_off, chain_base, chain_len, g.sort_tiles};
    // (the kernel takes its tile from the ticket, or from blockIdx.x of this one-dimensional grid: never from blockIdx.y)
    FELICS_LAUNCH((k_pack_g<T>), dim3((st1 - st0) * g.nplanes), dim3(PACK_THREADS), s, planes, gs, fa, st0, g.pack_tiles);
}
template void launch_pack_g<uint8_t>(hipStream_t, const uint8_t *, const uint8_t *, const uint32_t *, const uint32_t *,
                                     const uint32_t *, const uint32_t *, uint64_t *, uint64_t *, uint32_t *, uint64_t *,
                                     uint32_t *, uint32_t *, uint32_t *, const PackTarget &, const Geometry &, uint32_t,
                                     uint32_t, uint32_t, uint32_t *);
template void launch_pack_g<int16_t>(hipStream_t, const int16_t *, const uint8_t *, const uint32_t *, const uint32_t *,
                                     const uint32_t *, const uint32_t *, uint64_t *, uint64_t *, uint32_t *, uint64_t *,
                                     uint32_t *, uint32_t *, uint32_t *, const PackTarget &, const Geometry &, uint32_t,
                                     uint32_t, uint32_t, uint32_t *);

template <typename T>
void launch_pack_t(hipStream_t s, const T *planes, const uint8_t *kq, const uint16_t *pix, const uint32_t *tile_slots, uint32_t cap,
                   uint64_t *status, uint64_t *tile_bitoff, uint32_t *tile_bits, uint64_t *plane_carry, uint32_t *edge_first,
                   uint32_t *edge_last, uint32_t *error, const PackTarget &to, const Geometry &g, uint32_t st0, uint32_t st1, uint32_t epoch,
                   uint32_t *ticket) {
    if (st1 <= st0) return;
    const FusedArgs fa{status, tile_bitoff, tile_bits, plane_carry, edge_first, edge_last, error,
                       PlaneOut{to.out, to.slot_stride, to.scratch, to.plane_slot, g.planes_per_image},
                       g.W, g.H, g.npix, g.pack_tiles, g.color, g.depth, epoch, ticket, g.nplanes};
    const TSources ts{kq, pix, tile_slots, cap, g.sort_tiles};
    // (the kernel takes its tile from the ticket, or from blockIdx.x of this one-dimensional grid: never from blockIdx.y)
    FELICS_LAUNCH((k_pack_t<T>), dim3((st1 - st0) * g.nplanes), dim3(PACK_THREADS), s, planes, ts, fa, st0, g.pack_tiles);
}
template void launch_pack_t<uint8_t>(hipStream_t, const uint8_t *, const uint8_t *, const uint16_t *, const uint32_t *, uint32_t, uint64_t *,
                                     uint64_t *, uint32_t *, uint64_t *, uint32_t *, uint32_t *, uint32_t *, const PackTarget &,
                                     const Geometry &, uint32_t, uint32_t, uint32_t, uint32_t *);
template void launch_pack_t<int16_t>(hipStream_t, const int16_t *, const uint8_t *, const uint16_t *, const uint32_t *, uint32_t, uint64_t *,
                                     uint64_t *, uint32_t *, uint64_t *, uint32_t *, uint32_t *, uint32_t *, const PackTarget &,
                                     const Geometry &, uint32_t, uint32_t, uint32_t, uint32_t *);

template <typename T, typename ET>
void launch_front(hipStream_t s, const T *planes, const TileLocal<ET> &tl, const Geometry &g, uint32_t tile_begin, uint32_t tile_end,
                  uint32_t *flags, uint32_t mode) {
    if (tile_end <= tile_begin) return;
    const dim3 grid(8u * cdiv(g.nplanes, 8) * (tile_end - tile_begin));  // one workgroup per tile, the planes dealt to the XCDs by the kernel
    FELICS_LAUNCH((k_front<T, ET>), grid, dim3(256), s, planes, tl.ev, tl.pix, tl.runtab, tl.tile_slots, g.W, g.npix, g.sort_tiles, tile_begin,
                  tile_end, g.nplanes, tl.cap, flags, mode);
}
template void launch_front<uint8_t, uint8_t>(hipStream_t, const uint8_t *, const TileLocal<uint8_t> &, const Geometry &, uint32_t, uint32_t,
                                             uint32_t *, uint32_t);
template void launch_front<int16_t, uint16_t>(hipStream_t, const int16_t *, const TileLocal<uint16_t> &, const Geometry &, uint32_t, uint32_t,
                                              uint32_t *, uint32_t);

void launch_join_edges_tiles(hipStream_t s, const uint64_t *tile_bitoff, const uint32_t *tile_bits, const uint32_t *edge_first,
                             const uint32_t *edge_last, const PackTarget &to, const Geometry &g, uint32_t ntiles) {
    const PlaneOut po{to.out, to.slot_stride, to.scratch, to.plane_slot, g.planes_per_image};
    FELICS_LAUNCH(k_join_edges, dim3(cdiv(ntiles, 256), g.nplanes), dim3(256), s, tile_bitoff, tile_bits,
                       edge_first, edge_last, po, ntiles);
}

void launch_join_edges(hipStream_t s, const uint64_t *tile_bitoff, const uint32_t *tile_bits, const uint32_t *edge_first,
                       const uint32_t *edge_last, const PackTarget &to, const Geometry &g) {
    launch_join_edges_tiles(s, tile_bitoff, tile_bits, edge_first, edge_last, to, g, g.pack_tiles);
}

void launch_concat_planes(hipStream_t s, const uint64_t *plane_base, const uint64_t *plane_carry, const PackTarget &to,
                          const Geometry &g) {
    if (g.planes_per_image < 2) return;
    const PlaneOut po{to.out, to.slot_stride, to.scratch, to.plane_slot, g.planes_per_image};
    // the planes behind plane 0 hold at most plane_slot bytes each: enough threads for that many words
    const uint64_t words = (to.plane_slot >> 2) * (g.planes_per_image - 1);
    const uint32_t bx = (uint32_t)std::min<uint64_t>(cdiv(words, 256 * 4), 2048u);
    FELICS_LAUNCH(k_concat_planes, dim3(std::max(bx, 1u), g.nimages), dim3(256), s, plane_base, plane_carry, po);
}

}  // namespace felics
