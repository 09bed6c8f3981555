// felics_kernels.hip -- CDNA4 (gfx950) kernels of the FELICS encode path.
//
// What the reference does per pixel in one serial loop (src/compression.rs:117-146)
// is split here into data-parallel stages over a whole batch of planes:
//
//   planes   RGB -> Y/Co/Cg planes                (compression.rs:346-356, color_transform.rs:11-17)
//   front    classify every pixel against its two neighbours ONCE (misc.rs:6-24, compression.rs:124-145) and sort a tile's
//            out-of-range EVENTS by context, raster order kept, into the tile's own slots (tile-local layout, felics_kernels.h)
//   chains   (felics_chain.hip) replay the Rice-parameter estimator along every context's chain of events
//            (parameter_selection.rs:49-85): records of the chains, spine (sequential, per chain), k per record (parallel)
//   pack     build the codes (rice_coding.rs:26-38, phase_in_coding.rs:59-84,
//            compression.rs:29-45) and pack them MSB-first (bitstream-io BigEndian).
//            Single pass (k_pack_t: code lengths, tile offsets by decoupled look-back, packing),
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
#include <type_traits>

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

// ------------------------------------------------------------------------------------------
// k_front: the ONE classification of a pixel, and the sort of a tile's events by context.  One workgroup per tile of
// SORT_TILE pixels, a quarter of the tile per wave.
//
//   1. every wave classifies its 1024 pixels a trip of 256 at a time (two trips' loads in flight; there is no store in this
//      kernel before its last steps, so nothing makes the compiler wait for more than the load it needs).  Lane l takes pixels
//      l, l + 64, l + 128, l + 192 of a trip, so a trip's raster order is (sub-row, lane); every pixel keeps a static slot in
//      this lane's registers: its record, all ones if the pixel is no event;
//   2. ONE returning LDS add per sub-row, on the counter of each event's context, ranks the sub-row's events within (wave,
//      context) in raster order: the lanes that name the same address are served in ascending lane order -- measured over
//      2 x 10^10 atomics, profiles/tools/micro/lds_atomic_order.hip, and not documented anywhere, hence step 5's check.  A pixel
//      that is no event adds to a counter of its own lane (no exec mask around the atomic).  The ranks stay in registers, the
//      counters end as the wave's event count per context;
//   3. thread c turns the four waves' counts of context c into the tile's layout -- contexts in ascending order, within a
//      context wave 0's events, then wave 1's ..., every context's run starting on a multiple of REC slots -- and writes the
//      run table entry {first record, events} of (tile, c), the slots in use and the padding slots (pix = 0xFFFF);
//   4. every wave moves its events to their places in the tile's sorted order in LDS (start of its context + rank);
//   5. the workgroup writes the sorted tile out, 256 consecutive events per trip, to the tile's own place: slot
//      (plane * ntiles + tile) * cap + s (felics_kernels.h: tile-local layout) -- no histogram pass, no offsets from other
//      tiles.  Each event is compared with its successor: (context, pixel offset) must ascend strictly.  That is exactly
//      "stable partition": the set of events of a context is fixed by the counts, and ascending pixel offsets are the one
//      raster order of that set.  A violation raises TL_FLAG_ORDER; the host then redoes the batch with ranks from ballots.
//   mode & FRONT_SAFE_RANK: ranks from ballots instead of the returning add (a context's fallback once the order check has
//   failed: no assumption about the LDS); mode & FRONT_TEST_VIOLATION: report a violation whatever the order (tests).
//
// Record in LDS: context << 22 | value << 13 | above << 12 | pixel offset in the tile (9 + 9 + 1 + 12 bits; above = the sample lies
// above its neighbours, the second flag bit of its code: compression.rs:139-144).  pix[slot] = the low 13 bits.
// The chain of a context is the sequence of its runs over the tiles (felics_chain.hip).
// (six workgroups per CU is what the LDS allows -- five for Y / Co / Cg planes -- and the registers are held to that)
// ------------------------------------------------------------------------------------------
#ifndef FELICS_FRONT_WAVES
#define FELICS_FRONT_WAVES 6  // (gray planes; what the LDS allows.  A/B builds: profiles/tools/variant.sh)
#endif
template <typename T, typename ET>
__attribute__((amdgpu_waves_per_eu(sizeof(T) == 1 ? FELICS_FRONT_WAVES : 5))) __global__ __launch_bounds__(256) void k_front(
    const T *__restrict__ planes, ET *__restrict__ ev, uint16_t *__restrict__ pix, uint32_t *__restrict__ runtab,
    uint32_t *__restrict__ tile_slots, uint32_t W, uint32_t npix, uint32_t ntiles, uint32_t tile_begin, uint32_t tile_end,
    uint32_t nplanes, uint32_t cap, uint32_t *__restrict__ flags, uint32_t mode) {
    constexpr uint32_t NC = nctx_of<T>();
    constexpr uint32_t QUARTER = SORT_TILE / 4, TRIPS = QUARTER / 256;
    constexpr uint32_t PER = NC / 256;  // contexts per thread in step 3
    constexpr uint32_t KEY = 0xFFC00FFFu;  // context and pixel offset of a record
    static_assert(SORT_TILE % 1024 == 0 && SORT_TILE <= (1u << 12), "four whole trips per wave; 12 bits of pixel offset");
    static_assert(NC % 256 == 0 && NC <= 512, "a thread takes NC / 256 contexts; 9 bits of context");
    __shared__ uint32_t srt[SORT_TILE + 1 + 256];  // the tile's events, contexts ascending, raster order inside, a sentinel behind the last; [.. + 1 + thread]: where slots without an event are "placed"
    __shared__ uint32_t cnt[4][NC + 64];      // per wave and context: count, then cursor into srt; [NC + lane]: what slots without an event count on
    __shared__ uint32_t gdst[NC];             // a context's run: its place among the tile's slots minus its place in srt
    __shared__ uint32_t wsum[4];
    // (wave-uniform, and said so: the tile, its bounds and the trip bookkeeping then live in scalar registers instead of vector
    // registers under exec masks)
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = lane_id();
    const uint32_t tid = threadIdx.x;
    // Workgroup -> (plane, tile), XCD-aware: workgroups are dealt round-robin over the eight XCDs (MI355X_MICROARCH.md, workgroup
    // dispatch: blocks b and b + 8 share one), every XCD with an L2 of its own.  All tiles of plane p go to the XCD p % 8, in tile
    // order (workgroup b = 8 i + x takes item i of XCD x's list of planes x, x + 8, ...): the XCD whose k_enum, spine and pack
    // workgroups read plane p's tiles later, and where the neighbouring tiles' entries of a run-table row meet in one L2.
    // Placement only: nothing depends on it for correctness.
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
    const bool safe_rank = (mode & FRONT_SAFE_RANK) != 0;
    // ---- 1, 2. classify and rank, no compaction: a trip is 256 consecutive pixels, lane l takes pixels l, l + 64, l + 128, l + 192
    // of it -- so that the raster order of a trip's pixels is (sub-row, lane) and ONE returning LDS add per sub-row ranks its events
    // within (wave, context) in raster order (ascending lanes).  Every pixel keeps a static slot of this lane: its record
    // (context << 22 | value << 13 | offset in the tile; all ones where the pixel is no event) and its rank.  A slot without an event
    // adds to a counter of its own lane (no exec mask around the atomic).  (Round 4 compacted a trip's events through LDS first so
    // that every atomic ranked 64 events: a scan, four masked LDS writes and four reads per trip for half as many atomics.)
    constexpr uint32_t SLOTS = TRIPS * 4;
    uint32_t rec[SLOTS], rk[SLOTS];
    auto is_interior = [&](uint32_t r, uint32_t x, uint32_t y) { return y > 0 && x + 256 <= W && r + 256 <= end; };  // (a span from the first column included)
    struct Trip {  // the samples of an interior trip as loaded: four of the row, four of the row above, the one in front of the span
        uint32_t cur[4], up[4], left0;
    };
    auto load_trip = [&](uint32_t r, uint32_t left_index, Trip &t) {
#pragma unroll
        for (uint32_t j = 0; j < 4; j++) {
            t.cur[j] = field_of((int)pl[r + 64 * j + lane], T());
            t.up[j] = field_of((int)pl[r + 64 * j + lane - W], T());
        }
        t.left0 = field_of((int)pl[left_index], T());  // (the same address in every lane)
    };
    bool have[TRIPS];
    uint32_t leftidx[TRIPS];
    {
        uint32_t ri = qbegin, yi = qbegin / W, xi = qbegin - yi * W;
#pragma unroll
        for (uint32_t d = 0; d < TRIPS; d++) {
            have[d] = ri < end && is_interior(ri, xi, yi);
            leftidx[d] = have[d] ? span_left_index(ri, xi, yi, W) : 0u;
            ri += 256;
            xi += 256;
            if (xi >= W) {  // (once per image row: scalar division)
                const uint32_t q = xi / W;
                yi += q;
                xi -= q * W;
            }
        }
    }
    const uint32_t cnt_at = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint32_t *)my_cnt;  // LDS address of the wave's counters
    const uint32_t dummy_at = (NC + lane) * 4u;
    Trip pre[2];  // two trips in flight
    if (have[0]) load_trip(qbegin, leftidx[0], pre[0]);
#pragma unroll
    for (uint32_t d = 0; d < TRIPS; d++) {
        const uint32_t row0 = qbegin + d * 256;
        if (d + 1 < TRIPS && have[d + 1]) load_trip(row0 + 256, leftidx[d + 1], pre[(d + 1) & 1u]);
        if (have[d]) {
            const Trip &t = pre[d & 1u];
#pragma unroll
            for (uint32_t j = 0; j < 4; j++) {
                // the left neighbour: the lane before; lane 0 takes the last lane of the sub-row before (the sample in front of the
                // span for the first sub-row: the first-column rule's second neighbour if the span starts in column 0, misc.rs:14-23)
                uint32_t edge = t.left0;
                if (j > 0) edge = readlane(t.cur[j - 1], 63);
                const uint32_t a = (uint32_t)__builtin_amdgcn_update_dpp((int)edge, (int)t.cur[j], 0x138, 0xF, 0xF, false);  // wave_shr:1
                const uint32_t p = t.cur[j], b = t.up[j];
                // compression.rs:124-145 on unsigned 16-bit fields, selects as sign masks (the 1.0 ns instruction class, as in code_pixel)
                const uint32_t L = min_u16(a, b), H = max_u16(a, b);
                const uint32_t ctx = H - L;
                const int dd = (int)(p - L);         // in range: 0 <= dd <= ctx
                const int below = dd >> 31;          // all ones: p < L
                const int o = (int)(p - H) - 1;      // >= 0: p > H
                const int not_above = o >> 31;
                const uint32_t val = bitop3<BT_SEL>((uint32_t)(dd ^ below), (uint32_t)o, (uint32_t)not_above);  // L - p - 1 | p - L | p - H - 1
                const uint32_t in_range = bitop3<BT_ANDN>((uint32_t)not_above, (uint32_t)below, 0u);               // all ones: no event
                const uint32_t off = (row0 - begin + 64 * j + lane) | bitop3<BT_ANDN>(0x1000u, (uint32_t)not_above, 0u);  // | above << 12
                rec[d * 4 + j] = ((ctx << 22) | (val << 13) | off) | in_range;
                // rank: the counter of the event's context, or this lane's own
                const uint32_t at = cnt_at + bitop3<BT_SEL>(dummy_at, ctx << 2, in_range);
                if (!safe_rank)
                    rk[d * 4 + j] = __hip_atomic_fetch_add((__attribute__((address_space(3))) uint32_t *)(uintptr_t)at, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        } else {  // (a trip that crosses a row end, lies in the first row or ends the plane: the general neighbour rule)
            Coord xy;
            xy.set(row0 + lane, W);
#pragma unroll
            for (uint32_t j = 0; j < 4; j++) {
                const uint32_t i = row0 + j * 64 + lane;
                uint32_t r = 0xFFFFFFFFu;
                if (i < end && i >= 2) {
                    const PixelClass pc = classify(pl, i, xy.x, xy.y, W);
                    if (pc.cls != CLS_IN) r = (pc.ctx << 22) | (pc.val << 13) | (pc.cls == CLS_ABOVE ? 0x1000u : 0u) | (i - begin);
                }
                xy.advance(64, W);
                rec[d * 4 + j] = r;
                const uint32_t at = cnt_at + ((int)r < 0 ? dummy_at : (r >> 22) << 2);
                if (!safe_rank)
                    rk[d * 4 + j] = __hip_atomic_fetch_add((__attribute__((address_space(3))) uint32_t *)(uintptr_t)at, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        if (safe_rank) {
            // ranks from ballots: every lane learns the lanes that hold its context, one ballot per context bit; its rank = the
            // context's count so far + the lanes in front of it, the first lane of a context adds the sub-row's share to the count
#pragma unroll
            for (uint32_t j = 0; j < 4; j++) {
                const bool e = (int)rec[d * 4 + j] >= 0;
                const uint32_t c = (rec[d * 4 + j] >> 22) & (NC - 1u);
                const uint64_t ev_mask = __ballot(e);
                uint32_t m_lo = (uint32_t)ev_mask, m_hi = (uint32_t)(ev_mask >> 32);
                constexpr uint32_t CTX_BITS = NC == 256 ? 8 : 9;
#pragma unroll
                for (uint32_t bt = 0; bt < CTX_BITS; bt++) {
                    const uint32_t tt = (uint32_t)((int32_t)(c << (31 - bt)) >> 31);  // all ones if bit bt of c is set
                    const uint64_t bb = __ballot(e && tt != 0);
                    m_lo &= ~((uint32_t)bb ^ tt);
                    m_hi &= ~((uint32_t)(bb >> 32) ^ tt);
                }
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi(m_hi, __builtin_amdgcn_mbcnt_lo(m_lo, 0u));
                const uint32_t group = (uint32_t)__popc(m_lo) + (uint32_t)__popc(m_hi);
                uint32_t r = 0;
                if (e) r = my_cnt[c] + rank;
                __builtin_amdgcn_wave_barrier();
                if (e && rank == 0) my_cnt[c] = r + group;
                __builtin_amdgcn_wave_barrier();
                rk[d * 4 + j] = r;
            }
        }
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
        // the run table, context-major (a chain reads its row of tiles; the tiles of a plane run in order on one XCD, so the
        // neighbouring tiles' entries of a row meet in that XCD's L2) and the padding slots of this thread's runs
        uint32_t *rt = runtab + ((uint64_t)plane * NC + tid * PER) * ntiles + tile;
#pragma unroll
        for (uint32_t u = 0; u < PER; u++) rt[(uint64_t)u * ntiles] = (padpos[u] / REC) | (nev_c[u] << 16);
        uint16_t *px = pix + pt * cap;
#pragma unroll
        for (uint32_t u = 0; u < PER; u++)
            for (uint32_t i = nev_c[u]; i < ((nev_c[u] + REC - 1u) & ~(REC - 1u)); i++) px[padpos[u] + i] = 0xFFFFu;
        if (tid == 0) tile_slots[pt] = slots;
    } else {  // (an empty run table: the chain stage finds nothing of this tile)
        uint32_t *rt = runtab + ((uint64_t)plane * NC + tid * PER) * ntiles + tile;
#pragma unroll
        for (uint32_t u = 0; u < PER; u++) rt[(uint64_t)u * ntiles] = 0;
        if (tid == 0) {
            tile_slots[pt] = 0;
            atomicOr(flags, TL_FLAG_OVERFLOW);
        }
    }
    __syncthreads();
    // ---- 4. place: where the context's events of this wave start + the event's rank among them (a slot without an event: a
    // place of this thread's own behind the tile)
#pragma unroll
    for (uint32_t q0 = 0; q0 < SLOTS; q0 += 8) {
        uint32_t at[8];
#pragma unroll
        for (uint32_t q = 0; q < 8; q++) at[q] = my_cnt[(rec[q0 + q] >> 22) & (NC - 1u)];  // (masked: a slot without an event holds all ones)
#pragma unroll
        for (uint32_t q = 0; q < 8; q++) {
            const uint32_t none = (uint32_t)((int)rec[q0 + q] >> 31);
            srt[bitop3<BT_SEL>(SORT_TILE + 1u + tid, at[q] + rk[q0 + q], none)] = rec[q0 + q];
        }
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

// The code of a pixel IN RANGE, left-aligned in 32 bits, and its length (compression.rs:129-133): p against its two neighbours a, b
// (unordered; all three in field_at's form): `1`, then p - L phased-in on n = H - L + 1 values (phase_in_coding.rs:59-84):
// r = (p - L + P) mod n, P = 2^m = the largest power of two <= n; r < 2 P - n: r in m bits, else r + 2 P - n in m + 1 bits.
// Built for every pixel (garbage, and harmless, where the pixel is an event: the caller keeps the event's word instead).
// K31 = 0x80000000 in a vector register (vgpr_const).
struct PixelCode {
    uint32_t c32, len;
};
__device__ __forceinline__ PixelCode code_in_range(uint32_t p, uint32_t a, uint32_t b, uint32_t K31) {
    const uint32_t L = min_u16(a, b), H = max_u16(a, b);
    const uint32_t ctx = H - L;
    const uint32_t d = p - L;  // in range: 0 <= d <= ctx
    const uint32_t n = ctx + 1u;
    const uint32_t z = (uint32_t)__builtin_clz(n);  // n >= 1
    const uint32_t P = K31 >> z;
    const uint32_t r0 = d + P, r1 = r0 - n;
    const uint32_t r = bitop3<BT_SEL>(r0, r1, (uint32_t)((int)r1 >> 31));  // r0 mod n
    const uint32_t P2 = twice(P), right_p = P2 - n;
    const int is_short = (int)(r - right_p) >> 31;
    const uint32_t code_in = r + bitop3<BT_SEL>(P, P2 + right_p, (uint32_t)is_short);  // `1` in front of m or m + 1 bits
    PixelCode pc;
    pc.len = (33u - z) + (uint32_t)is_short;                                            // m + 1 or m + 2
    pc.c32 = code_in << ((32u - pc.len) & 31u);
    return pc;
}

// The WORD of an event (k_pack_t builds it where it gathers the tile's events, once per event instead of once per pixel): its code
// `00` / `01` (below / above), then the value Rice-coded -- q ones, `0`, k low bits (rice_coding.rs:26-38) -- left-aligned, with k
// and the length beside it:   code << (32 - len) | k << 6 | len   for len = q + k + 3 <= RICE_WORD_MAX_LEN (the code's bits end above
// bit 9), else   len << 9 | k << 6 | 63: such a group of pixels is coded the general way (it needs k and the exact length).
// A pixel that is no event has the word 0.   K7F = 0x7FFFFFFF in a vector register.
constexpr uint32_t RICE_WORD_MAX_LEN = 23, RICE_WORD_LONG = 63;
constexpr uint32_t WORD_PATH_MAX_SLOTS = 2048;  // k_pack_t builds the events' codes per event up to this many slots in use of a tile (of 4096 pixels), per pixel beyond
__device__ __forceinline__ uint32_t rice_word(uint32_t val, uint32_t k, uint32_t above /* 0 or 1 */, uint32_t K7F) {
    const uint32_t q = val >> k;
    const uint32_t len = q + k + 3u;
    const uint32_t ones = K7F >> ((31u - q) & 31u);               // q ones (q <= 20 wherever the code is used)
    const uint32_t head = bitop3<BT_OR_ANDN>(ones, ones + 1u, above - 1u);   // `0` / `1` (above) in front of them
    const uint32_t rice = bitop3<BT_OR_ANDN>(head << (k + 1u), val, ~0u << k);  // then `0` and the k low bits of val
    const uint32_t c32 = rice << ((32u - len) & 31u);
    return len <= RICE_WORD_MAX_LEN ? (c32 | (k << 6) | len) : ((len << 9) | (k << 6) | RICE_WORD_LONG);
}
__device__ __forceinline__ uint32_t word_k(uint32_t w) { return (w >> 6) & 7u; }
// Where the word of pixel j of the tile lies in LDS: [quad of the thread's sixteen pixels][thread][pixel of the quad] -- a thread's 16-byte
// read of a quad is then 16 bytes from its neighbour's (pixel-major, a wave's reads were 64 bytes apart: sixteen lanes on the same banks,
// and a tile without a single event took twice as long as before).  PACK_TILE and beyond (the dump word): as they are.
__device__ __forceinline__ uint32_t word_index(uint32_t j) {
    return j >= PACK_TILE ? j : (((j >> 2) & 3u) << 10) | ((j >> 4) << 2) | (j & 3u);
}
static_assert(PACK_TILE == 4096 && PACK_PER_THREAD == 16, "word_index: 256 threads x 4 quads x 4 pixels");

// One pixel's code, left-aligned in 32 bits, and its length (compression.rs:124-145): p against its two neighbours a, b
// (unordered; all three in field_at's form), k = the Rice parameter of the pixel's context (used if p is out of range).
//   in range (L <= p <= H): `1`, then p - L phased-in on n = H - L + 1 values (phase_in_coding.rs:59-84): r = (p - L + P) mod n,
//       P = 2^m = the largest power of two <= n; r < 2 P - n: r in m bits, else r + 2 P - n in m + 1 bits;
//   below / above: `00` / `01`, then L - p - 1 / p - H - 1 Rice-coded: q ones, `0`, k low bits (rice_coding.rs:26-38).
// Both are built and one is kept.  A Rice code longer than 32 bits comes out as garbage with len > 32: the caller redoes
// such a group the general way.  K31 = 0x80000000, K7F = 0x7FFFFFFF in vector registers (vgpr_const).
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
// two samples and its ragged end, Rice codes too long for a word, images narrower than 16 pixels).
//   * an EVENT's code comes ready-made from its word (words[pixel]: rice_word, built by the gather); a pixel in range (word 0)
//     takes the phased-in code of p - L, built here for all sixteen pixels and kept where the word is 0;
//   * neighbours: left and above (misc.rs:6-24, interior case); the left neighbour of pixel j is pixel j - 1 of the group.
//   * a first-column pixel (j0) takes above and two rows up (above-right in row 1) instead; the pair is unordered (H = max,
//     L = min), so that rule only replaces the LEFT sample of that one pixel by `special`, which the caller fetched.  One
//     thread in 240 has such a pixel: its code is built a second time where a wave holds such a thread.
// Returns the total length in bits (exact also when a Rice code did not fit its word: the word then carries the length);
// longest = the longest code's length field (RICE_WORD_LONG: one did not fit).
template <typename T>
__device__ __forceinline__ uint32_t group_codes_w(const GroupSamples<T> &g, const uint32_t *words, const T *__restrict__ pl, uint32_t first,
                                                uint32_t W, uint32_t j0, int special, uint32_t (&c32)[PACK_PER_THREAD],
                                                uint32_t (&len)[PACK_PER_THREAD], uint32_t &longest) {
    const uint32_t off = threadIdx.x * PACK_PER_THREAD;
    const uint32_t K31 = vgpr_const(0x80000000u);
    uint32_t left = field_of(g.before, T());  // the sample in front of the group
    longest = 0;
#pragma unroll
    for (uint32_t u = 0; u < PACK_PER_THREAD / 4; u++) {  // (four words at a time: sixteen of them at once cost twelve more registers)
        const uint4 c = *reinterpret_cast<const uint4 *>(words + u * (PACK_TILE / 4) + threadIdx.x * 4);  // word_index(off + 4 u ..)
        const uint32_t wq[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
        for (uint32_t i = 0; i < 4; i++) {
            const uint32_t j = 4 * u + i;
            const uint32_t p = field_at(g.cw, j, T());
            const PixelCode pc = code_in_range(p, left, field_at(g.uw, j, T()), K31);
            const uint32_t lf = wq[i] & 63u;
            const uint32_t in_range = (uint32_t)((int)(lf - 1u) >> 31);  // all ones: the word is 0, no event
            c32[j] = bitop3<BT_SEL>(pc.c32, wq[i] & ~0x1FFu, in_range);
            len[j] = bitop3<BT_SEL>(pc.len, lf, in_range);
            longest = max_u16(longest, len[j]);
            left = p;
        }
    }
    if (__ballot(j0 < PACK_PER_THREAD) != 0) {  // (wave-uniform: a quarter of the waves of a 4K plane)
        if (j0 < PACK_PER_THREAD && (words[word_index(off + j0)] & 63u) == 0u) {  // (an event's word does not depend on who its neighbours are here)
            const uint32_t i0 = first + j0;
            const PixelCode pc = code_in_range(field_of((int)pl[i0], T()), field_of(special, T()), field_of((int)pl[i0 - W], T()), K31);
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
    if (longest == RICE_WORD_LONG) {  // (rare: full-scale spikes) a word that carries its code's length instead of the code
        for (uint32_t j = 0; j < PACK_PER_THREAD; j++) {
            const uint32_t w = words[word_index(off + j)];
            if ((w & 63u) == RICE_WORD_LONG) total += (w >> 9) - RICE_WORD_LONG;
        }
    }
    return total;
}

// group_codes for a tile where most pixels are events (k_pack_t: the tile's slots outnumber WORD_PATH_MAX_SLOTS): k per pixel in LDS,
// both codes built for every pixel and one kept.  Every pixel below the first image row, the group
// inside the plane.  Same codes as classify + put_pixel, which stay as the general path (first row, the plane's first
// two samples and its ragged end, codes longer than 32 bits, images narrower than 16 pixels).
//   * neighbours: left and above (misc.rs:6-24, interior case); the left neighbour of pixel j is pixel j - 1 of the group.
//   * a first-column pixel (j0) takes above and two rows up (above-right in row 1) instead; the pair is unordered (H = max,
//     L = min), so that rule only replaces the LEFT sample of that one pixel by `special`, which the caller fetched.  One
//     thread in 240 has such a pixel: its code is built a second time where a wave holds such a thread.
// kq = the tile's k bytes in LDS.  Returns the total length in bits (exact also when a code is longer than 32 bits: only
// that code's c32 is garbage then); longest = the longest code's length.
template <typename T>
__device__ __forceinline__ uint32_t group_codes_k(const GroupSamples<T> &g, const uint8_t *kq, const T *__restrict__ pl, uint32_t first,
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
// scratch memory.  (words arrives as a generic pointer into LDS: k of an event is in its word; these paths are rare.)
struct GeneralGroup {
    uint32_t tile_first, first, end, W, H, npix, color, depth, has_header;
    uint32_t by_word;  // k of pixel i: word_k(words[i]) (k_pack_t's word path), or byte i of the same LDS array (its k path)
};
template <typename T, typename FR, typename F>
__device__ __forceinline__ void walk_group_global(const T *__restrict__ pl, const uint32_t *words, const GeneralGroup &g, FR &&raw, F &&f) {
    Coord xy;
    xy.set(g.first, g.W);
    for (uint32_t i = g.first; i < min(g.end, g.first + PACK_PER_THREAD); i++) {
        if (i < 2)
            raw(i, (uint32_t)(int)pl[i]);  // stored as 32-bit values (compression.rs:105-106)
        else
            f(classify(pl, i, xy.x, xy.y, g.W),
              g.by_word ? word_k(words[word_index(i - g.tile_first)]) : (uint32_t)reinterpret_cast<const uint8_t *>(words)[i - g.tile_first]);
        xy.advance(1, g.W);
    }
}
template <typename T>
__device__ __noinline__ uint32_t general_group_bits(const uint32_t *words, const T *pl, const GeneralGroup g) {
    uint32_t bits = 0;
    if (g.first < g.end) {
        const uint32_t npix = g.npix;
        if (g.has_header) bits += 8u * 14u;
        walk_group_global(pl, words, g, [&](uint32_t, uint32_t) { bits += npix == 1 ? 64u : 32u; },
                          [&](const PixelClass &pc, uint32_t k) { bits += code_length(pc, k); });
    }
    return bits;
}
// (the window's word 0 is stream word win_word0; bit 0 of this group is stream bit my_lo)
template <typename T>
__device__ __noinline__ void general_group_place(const uint32_t *words, const T *pl, const GeneralGroup g, uint32_t *win,
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
    walk_group_global(pl, words, g,
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
        // No counter (launch_pack_t with a null ticket: one-dimensional grid): the workgroup index, in the same (tile, plane)
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

// The single-pass pack of ONE tile by a workgroup (the body of k_pack_t): see the comment above.
// gsm = this thread's samples (valid where gg.fast); words = the tile's event words in LDS (BY_WORD: 0 where a pixel is no event; else
// the array holds k of pixel j in byte j) and fl.win all zero, with a barrier behind both.
template <typename T, bool BY_WORD>
__device__ __forceinline__ void pack_tile_fused(const GroupSamples<T> &gsm, const uint32_t *words, FusedLDS &fl, const T *__restrict__ planes,
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
    const GeneralGroup general{tile_first, first, end, W, H, npix, fa.color, fa.depth, has_header ? 1u : 0u, BY_WORD ? 1u : 0u};
    uint32_t bits = 0;
    if (!gg.fast) bits = general_group_bits<T>(words, pl, general);  // count now, build the codes straight into the window later
    uint32_t c32[PACK_PER_THREAD], len[PACK_PER_THREAD];
    bool in_registers = false;
    if (gg.fast) {
        uint32_t longest;
        if (BY_WORD) {
            bits = group_codes_w<T>(gsm, words, pl, first, W, gg.j0, gg.special, c32, len, longest);
            in_registers = longest != RICE_WORD_LONG;  // (a Rice code too long for its word: the lengths stand, the codes are built again the general way)
        } else {
            bits = group_codes_k<T>(gsm, reinterpret_cast<const uint8_t *>(words), pl, first, W, gg.j0, gg.special, c32, len, longest);
            in_registers = longest <= 32u;  // (a longer code: the lengths stand, the codes are built again the general way)
        }
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
        if (!in_registers && bits != 0) general_group_place<T>(words, pl, general, win, FUSED_WIN_WORDS, 0, my_rel);
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
                general_group_place<T>(words, pl, general, win, FUSED_WIN_WORDS, first_word + wb, my_lo);
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
// k_pack_t (round 5): the single-pass pack on the tile-local layout.  A tile's events lie where the front kernel put them --
// ev[slot] the value, pix[slot] the pixel's offset and the above flag, kq[slot] the k that k_assign3 left, slots [0, tile_slots) of
// the tile -- so the gather is one contiguous read: four slots per thread and round, and each event's CODE is built right there,
// once per event (rice_word), and dropped into the LDS array the code phase indexes by pixel (padding slots, pix = 0xFFFF, into
// a dump word behind it).  The code phase then builds the phased-in code of every pixel and keeps the event's word where there is
// one: the Rice code is no longer built for every pixel (of which one in seven is an event).  No run table, no chains here.
// ------------------------------------------------------------------------------------------
struct TSources {
    const uint8_t *kq;
    const uint16_t *pix;
    const void *ev;  // u8 (gray planes) / u16 (Y / Co / Cg planes) per slot
    const uint32_t *tile_slots;
    uint32_t cap, sort_ntiles;
};

#ifndef FELICS_PACK_WAVES
#define FELICS_PACK_WAVES 6  // (what the LDS allows: 24.6 KB per workgroup of four waves; A/B builds: profiles/tools/variant.sh)
#endif
template <typename T>
__attribute__((amdgpu_waves_per_eu(FELICS_PACK_WAVES))) __global__ __launch_bounds__(PACK_THREADS) void k_pack_t(const T *__restrict__ planes, TSources ts, FusedArgs fa,
                                                                                               uint32_t sort_tile_begin, uint32_t pack_tile_end) {
    using ET = typename std::conditional<sizeof(T) == 1, uint8_t, uint16_t>::type;
    __shared__ alignas(16) uint32_t words[PACK_TILE + 4];  // the word of the event at pixel tile_first + j, 0 where there is none; [PACK_TILE]: dump
    __shared__ FusedLDS fl;
    static_assert(SORT_TILE == PACK_TILE, "one workgroup = one sort tile = one pack tile (one look-back per workgroup)");
    static_assert(PACK_TILE == 4 * 4 * PACK_THREADS, "the clearing of words: four 16-byte stores per thread");
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
    const uint8_t *ksrc = ts.kq + pt * ts.cap;
    const uint16_t *psrc = ts.pix + pt * ts.cap;
    constexpr uint32_t GR = 3;  // rounds in flight together: 3072 slots
    if (ns <= WORD_PATH_MAX_SLOTS) {
        // ---- round trip 2, few events (smooth and natural content: a 4K S1 tile has ~600 slots in use): every event's code built
        // here, once per event.  Only the waves whose slots exist do that (wave w of round u: slots u * 1024 + w * 256 ..).
#pragma unroll
        for (uint32_t u = 0; u < 4; u++) reinterpret_cast<uint4 *>(words)[threadIdx.x + u * PACK_THREADS] = make_uint4(0u, 0u, 0u, 0u);
        PSTAMP(9);
        const ET *esrc = reinterpret_cast<const ET *>(ts.ev) + pt * ts.cap;
        const uint32_t K7F = vgpr_const(0x7FFFFFFFu);
        constexpr uint32_t WR = (WORD_PATH_MAX_SLOTS + 4 * PACK_THREADS - 1) / (4 * PACK_THREADS);  // rounds at most
        const uint32_t wave_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x & ~63u)) * 4u;
        uint32_t kv[WR];
        uint2 pv[WR];
        uint32_t evv[WR][sizeof(ET) == 1 ? 1 : 2];
#pragma unroll
        for (uint32_t u = 0; u < WR; u++) {  // (a thread past the end takes the tile's last four slots: no lane-wise condition on a load)
            const uint32_t s = min(u * 4 * PACK_THREADS + threadIdx.x * 4, max(ns, 4u) - 4u);
            kv[u] = *reinterpret_cast<const uint32_t *>(ksrc + s);
            pv[u] = *reinterpret_cast<const uint2 *>(psrc + s);
            __builtin_memcpy(evv[u], esrc + s, 4 * sizeof(ET));
        }
        __syncthreads();  // the words are cleared before the first of them is written
#pragma unroll
        for (uint32_t u = 0; u < WR; u++) {
            if (u * 4 * PACK_THREADS + wave_first < ns) {  // (wave-uniform)
#pragma unroll
                for (uint32_t i = 0; i < 4; i++) {
                    const uint32_t pw = (i & 2u) ? pv[u].y : pv[u].x;
                    const uint32_t p = (i & 1u) ? pw >> 16 : pw & 0xFFFFu;
                    const uint32_t k = (kv[u] >> (8u * i)) & 7u;
                    const uint32_t e = sizeof(ET) == 1 ? (evv[u][0] >> (8u * i)) & 0xFFu
                                                       : (evv[u][sizeof(ET) == 1 ? 0 : (i >> 1)] >> (16u * (i & 1u))) & 0xFFFFu;
                    // (a padding slot, pix = 0xFFFF: whatever its k and value make goes to the dump word)
                    words[word_index(min(p & 0xEFFFu, PACK_TILE))] = rice_word(e, k, (p >> 12) & 1u, K7F);
                }
            }
        }
        PSTAMP(2);
        __syncthreads();
        PSTAMP(3);
        if (st < pack_tile_end) pack_tile_fused<T, true>(gsm, words, fl, planes, fa, st, plane, gg);
    } else {
        // ---- round trip 2, many events (noise, texture): k of every event into byte `pixel` of the array; both codes of every pixel
        // are then built in the code phase (there are more slots than pixels to build a code for).  (A thread past the end takes
        // the last four slots again -- the same k goes to the same pixels twice -- so the loads run under no lane-wise condition.)
        uint8_t *kq = reinterpret_cast<uint8_t *>(words);
        PSTAMP(9);
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
            for (uint32_t u = 0; u < GR; u++) {  // (pix: offset | above << 12; padding 0xFFFF -> the dump byte)
                kq[min(pv[u].x & 0xEFFFu, PACK_TILE)] = (uint8_t)kv[u];
                kq[min((pv[u].x >> 16) & 0xEFFFu, PACK_TILE)] = (uint8_t)(kv[u] >> 8);
                kq[min(pv[u].y & 0xEFFFu, PACK_TILE)] = (uint8_t)(kv[u] >> 16);
                kq[min((pv[u].y >> 16) & 0xEFFFu, PACK_TILE)] = (uint8_t)(kv[u] >> 24);
            }
        }
        PSTAMP(2);
        __syncthreads();
        PSTAMP(3);
        if (st < pack_tile_end) pack_tile_fused<T, false>(gsm, words, fl, planes, fa, st, plane, gg);
    }
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

template <typename T>
void launch_pack_t(hipStream_t s, const T *planes, const uint8_t *kq, const uint16_t *pix, const void *ev, const uint32_t *tile_slots, uint32_t cap,
                   uint64_t *status, uint64_t *tile_bitoff, uint32_t *tile_bits, uint64_t *plane_carry, uint32_t *edge_first,
                   uint32_t *edge_last, uint32_t *error, const PackTarget &to, const Geometry &g, uint32_t st0, uint32_t st1, uint32_t epoch,
                   uint32_t *ticket) {
    if (st1 <= st0) return;
    const FusedArgs fa{status, tile_bitoff, tile_bits, plane_carry, edge_first, edge_last, error,
                       PlaneOut{to.out, to.slot_stride, to.scratch, to.plane_slot, g.planes_per_image},
                       g.W, g.H, g.npix, g.pack_tiles, g.color, g.depth, epoch, ticket, g.nplanes};
    const TSources ts{kq, pix, ev, tile_slots, cap, g.sort_tiles};
    // (the kernel takes its tile from the ticket, or from blockIdx.x of this one-dimensional grid: never from blockIdx.y)
    FELICS_LAUNCH((k_pack_t<T>), dim3((st1 - st0) * g.nplanes), dim3(PACK_THREADS), s, planes, ts, fa, st0, g.pack_tiles);
}
template void launch_pack_t<uint8_t>(hipStream_t, const uint8_t *, const uint8_t *, const uint16_t *, const void *, const uint32_t *, uint32_t, uint64_t *,
                                     uint64_t *, uint32_t *, uint64_t *, uint32_t *, uint32_t *, uint32_t *, const PackTarget &,
                                     const Geometry &, uint32_t, uint32_t, uint32_t, uint32_t *);
template void launch_pack_t<int16_t>(hipStream_t, const int16_t *, const uint8_t *, const uint16_t *, const void *, const uint32_t *, uint32_t, uint64_t *,
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
