// felics_chain.hip -- the chain stage of the 8-bit pipeline in tile-local layout (felics_kernels.h): the replay of
// KEstimator (parameter_selection.rs:24-85) along every context's chain of events.
//
//   k_enum     per slice of tiles: the records (16-event pieces of a tile's run of one context) of every chain, in chain
//              order, from the front kernel's run table
//   k_spine3   the one sequential part: per chain, the estimator's state at the start of every record.  A WALKER wave
//              jumps from halving to halving (one search over a window of 64 records, one inside the record that holds
//              the halving); a HELPER wave prepares the windows ahead of it -- a lane per record, no cross-lane work in
//              its loop -- and writes the records' start states out behind it
//   k_assign3  k of every event: one lane replays one record from its start state
//
// State S[k] = accumulated Rice lengths for k = 0..5 (traits.rs:26).  While no halving happens the state seen by event t
// is S + P_excl(t), P = prefix sums of the six length vectors (rice_coding.rs:56-58).  `min(S + P_incl(t)) > 1024`
// (parameter_selection.rs:58-63) is monotone in t because lengths are positive, so the first t where it holds is the next
// halving: S <- (S + P_incl(t)) >> 1.  get_k ties go to the LARGEST k (`<=` at parameter_selection.rs:79).
//
// Integer work only: no MFMA.  Wave = 64 lanes everywhere.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "felics_device.h"
#include "felics_kernels.h"
#include "felics_codes.h"

namespace felics {

static inline uint32_t cdiv_u(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

// ------------------------------------------------------------------------------------------
// k_enum: one workgroup per (chain, slice).  The chain of context c of plane p in this slice = the runs (tile, c) of the
// slice's tiles in tile order, each cut into records of REC events; the chain's row of the run table is contiguous
// (context-major).  All threads add up the row's records; one atomic per non-empty chain gives the chain its place in the
// slice's region; then, 64 tiles (a "round") per wave at a time, the records are written 64 CONSECUTIVE records per store
// instruction: lane = record, which finds its tile by bisection over the round's scanned record counts in LDS.  (A lane per
// tile writing its own records one by one was one scattered 8-byte request per record: 0.7 ms per 64-frame step; one wave
// per chain walked the hot chains' thousands of 64-record batches one after the other: 0.45 ms.)
// All of plane p's workgroups on the XCD p % 8 whose front workgroups wrote that plane's table (placement only).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_enum(const uint32_t *__restrict__ runtab, uint2 *__restrict__ desc, uint2 *__restrict__ chain_seg,
                                              uint32_t *__restrict__ slice_nrec, uint32_t ntiles, uint32_t t0, uint32_t t1, uint32_t nplanes,
                                              uint32_t nctx, uint32_t cap_rec) {
    constexpr uint32_t CHUNK_ROUNDS = 64;  // rounds whose totals are scanned together: 4096 tiles
    __shared__ uint32_t sh_before[4][65 + 63], sh_first[4][64], sh_events[4][64];  // per wave (before: padded, the bisection may look one step past 64)
    __shared__ uint32_t round_total[CHUNK_ROUNDS], wave_sum[4], sh_base;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = lane_id();
    const uint32_t item = blockIdx.x >> 3, xcd = blockIdx.x & 7u;
    const uint32_t plane = xcd + 8u * (item / nctx);
    if (plane >= nplanes) return;
    const uint32_t ctx = item % nctx;
    const uint32_t chain = plane * nctx + ctx;
    const uint32_t *row = runtab + (uint64_t)chain * ntiles;  // the chain's runs, tile by tile
    // ---- the chain's records in this slice
    uint32_t total = 0;
    for (uint32_t t = t0 + threadIdx.x; t < t1; t += 256) total += ((row[t] >> 16) + REC - 1) / REC;
    total = wave_incl_scan(total);
    if (lane == 63) wave_sum[wave] = total;
    __syncthreads();
    total = wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
    if (total == 0) {
        if (threadIdx.x == 0) chain_seg[chain] = make_uint2(0u, 0u);
        return;
    }
    if (threadIdx.x == 0) {
        const uint32_t base = atomicAdd(slice_nrec, total);
        chain_seg[chain] = make_uint2(base, total);
        sh_base = base;
    }
    __syncthreads();
    uint32_t carry = sh_base;  // records in front of the chunk
    uint32_t *before = sh_before[wave];  // [i]: records of the round's tiles in front of tile i; [64]: all of them
    uint32_t *first = sh_first[wave];    // [i]: the run's first record over the whole sub-batch
    uint32_t *events = sh_events[wave];  // [i]: its events
    for (uint32_t c0 = t0; c0 < t1; c0 += CHUNK_ROUNDS * 64) {
        // the rounds of this chunk: wave w takes rounds w, w + 4, ...; their totals first, scanned by wave 0
        const uint32_t nrounds = min(CHUNK_ROUNDS, (t1 - c0 + 63u) / 64u);
        for (uint32_t r = wave; r < nrounds; r += 4) {
            const uint32_t t = c0 + r * 64 + lane;
            const uint32_t nr = t < t1 ? ((row[t] >> 16) + REC - 1) / REC : 0u;
            const uint32_t incl = wave_incl_scan(nr);
            if (lane == 63) round_total[r] = incl;
        }
        __syncthreads();
        uint32_t rt = 0;
        if (wave == 0) {
            rt = lane < nrounds ? round_total[lane] : 0u;
            const uint32_t incl = wave_incl_scan(rt);
            round_total[lane] = incl - rt;  // now: records of the chunk in front of round `lane`
            if (lane == 63) wave_sum[0] = incl;
        }
        __syncthreads();
        for (uint32_t r = wave; r < nrounds; r += 4) {
            const uint32_t t = c0 + r * 64 + lane;
            const uint32_t e = t < t1 ? row[t] : 0u;
            const uint32_t n = e >> 16, nr = (n + REC - 1) / REC;
            const uint32_t incl = wave_incl_scan(nr);
            const uint32_t total_r = readlane(incl, 63);
            if (total_r == 0) continue;
            const uint32_t run = carry + round_total[r];
            __builtin_amdgcn_wave_barrier();  // (the previous round's reads are done)
            before[lane] = incl - nr;
            if (lane == 0) before[64] = total_r;
            first[lane] = (plane * ntiles + t) * cap_rec + (e & 0xFFFFu);
            events[lane] = n;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            for (uint32_t i0 = 0; i0 < total_r; i0 += 64) {
                const uint32_t i = i0 + lane;
                // the last tile with before[tile] <= i: six halvings of [0, 64)
                uint32_t lo = 0;
#pragma unroll
                for (uint32_t step = 32; step != 0; step >>= 1)
                    if (before[lo + step] <= i) lo += step;
                const uint32_t j = i - before[lo];
                if (i < total_r) desc[run + i] = make_uint2(first[lo] + j, min(REC, events[lo] - j * REC));
            }
        }
        carry += wave_sum[0];
        __syncthreads();  // (round_total and wave_sum are written again)
    }
}

void launch_enum(hipStream_t s, const uint32_t *runtab, const ChainSlice &cs, const Geometry &g, uint32_t tile_begin, uint32_t tile_end,
                 uint32_t cap) {
    if (tile_end <= tile_begin) return;
    const dim3 grid(8u * cdiv_u(g.nplanes, 8) * g.nctx);
    FELICS_LAUNCH(k_enum, grid, dim3(256), s, runtab, cs.desc, cs.chain_seg, cs.nrec, g.sort_tiles, tile_begin, tile_end, g.nplanes, g.nctx,
                  cap / REC);
}

// ------------------------------------------------------------------------------------------
// k_spine3.
//
// A window = 64 consecutive records of the chain (up to 1024 events).  For window w the helper leaves in LDS
//   pref[w & 1][j][t]   the packed inclusive prefix sums of record j's six length vectors at its event t (three dwords, two
//                        16-bit fields each: 16 * 511 < 2^16); events past the record's last one carry on from it and are
//                        never looked at: a search that reaches this record finds its event among the real ones, because
//                        the record's total (below) is what made the search come here;
//   cumT[w % 3][j + 1][k]  the window's cumulative sum of counter k through record j (32 bits; row 0 = zeros; the rows
//                        behind the chain's last record repeat it).
// The walker holds the state as a vector (lane l: S[l & 7]) and, per window, the six cumulative sums of record `lane` in six
// registers.  With base_k = the window's cumulative sum at the last halving, min_k(S_k + cum_k - base_k) > 1024 <=>
// cum_k >= theta_k = base_k + max(1025 - S_k, 0) for all k: ONE ballot over the 64 records finds the record f of the next
// halving, one over the 16 events of f (their prefix sums against theta - cum(f - 1), as packed 16-bit compares) the event;
// the state follows, the state minus the new base is left in lastD[f] for the helper.  The start state of a record is
//   lastD[last record in front of it with a halving] (or the window's carry-in) + cum(record - 1),
// which the helper computes a hand-over later, lane = record, and stores with the record's place (state16).
// Hand-over: one workgroup barrier per window; the helper produces window w + 1 and finishes window w - 1 while the walker
// walks window w.  Chains with fewer than SP3_MULTI_MIN windows to walk in this launch are done by one wave, step by step.
// ------------------------------------------------------------------------------------------

#ifdef FELICS_SPINE_STAMPS
// Diagnostic build only (profiles/tools/spine_stamps.py): s_memtime ticks of the two waves of ONE chain (context 1 of plane 0, the
// longest of an S1 frame): [0] walker waiting at the hand-over barrier, [1] walker walking, [2] helper working, [3] helper waiting,
// [4] halvings, [5] windows, [7] the walker's whole life.
__device__ unsigned long long g_spine_stamps[8];
extern "C" __attribute__((visibility("default"))) int felics_debug_spine_stamps(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_spine_stamps), sizeof(g_spine_stamps)) != hipSuccess) return -1;
    if (reset) {
        static unsigned long long z[8] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_spine_stamps), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#define SP3_STAMP(i)                                                        \
    do {                                                                    \
        if (stamped) {                                                      \
            const unsigned long long now_ = __builtin_amdgcn_s_memtime();   \
            st_acc[i] += now_ - st_last;                                    \
            st_last = now_;                                                 \
        }                                                                   \
    } while (0)
#else
#define SP3_STAMP(i)
#endif

constexpr uint32_t SP3_ROW = REC * 3 + 1;   // dwords per record in pref (odd: the helper's 64 rows start in different banks)
constexpr uint32_t SP3_CROW = 9;            // dwords per row of cumT (odd, likewise)
constexpr uint32_t SP3_MULTI_MIN = 3;       // windows

struct Spine3LDS {
    uint32_t pref[2][64 * SP3_ROW];
    uint32_t cumT[3][66 * SP3_CROW];
    uint32_t lastD[2][65 * 8];  // row 0: the state at the window's start; row j + 1: state - base behind the last halving in record j
    uint32_t hmask[2][2];       // records of the window with a halving
    uint32_t grec[3][64];       // the records' places (desc.x)
};

typedef unsigned short pk_u16 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) uint32_t lds_u32;
// both 16-bit halves of p >= the halves of theta
__device__ __forceinline__ bool pk_all_ge(uint32_t p, uint32_t theta) {
    const pk_u16 a = __builtin_bit_cast(pk_u16, p), b = __builtin_bit_cast(pk_u16, theta);
    const pk_u16 m = __builtin_elementwise_max(a, b);
    return __builtin_bit_cast(uint32_t, m) == p;
}

// a record's events as loaded: 16 bytes (u8) or 32 (u16)
template <typename ET>
struct RecEvents {
    static constexpr uint32_t NW = REC * sizeof(ET) / 4;
    uint32_t w[NW];
};
template <typename ET>
__device__ __forceinline__ void load_record(const ET *__restrict__ ev, uint32_t rec, RecEvents<ET> &r) {
    const uint4 *src = reinterpret_cast<const uint4 *>(ev + (uint64_t)rec * REC);
#pragma unroll
    for (uint32_t q = 0; q < RecEvents<ET>::NW / 4; q++) {
        const uint4 v = src[q];
        r.w[4 * q] = v.x; r.w[4 * q + 1] = v.y; r.w[4 * q + 2] = v.z; r.w[4 * q + 3] = v.w;
    }
}
template <typename ET>
__device__ __forceinline__ uint32_t record_event(const RecEvents<ET> &r, uint32_t t) {
    constexpr uint32_t EPW = 4 / sizeof(ET);
    const uint32_t word = r.w[t / EPW], sh = (t % EPW) * 8u * sizeof(ET);
    return sizeof(ET) == 1 ? (word >> sh) & 0xFFu : (word >> sh) & 0xFFFFu;
}

// Event t of a record twice in one register (low and high half): one v_perm of the dword that holds it.
template <typename ET>
__device__ __forceinline__ uint32_t record_event_twice(const RecEvents<ET> &r, uint32_t t) {
    constexpr uint32_t EPW = 4 / sizeof(ET);
    const uint32_t word = r.w[t / EPW], i = t % EPW;
    if (sizeof(ET) == 1) return __builtin_amdgcn_perm(0u, word, 0x0c000c00u + i * 0x00010001u);  // bytes {i, 0, i, 0}
    return __builtin_amdgcn_perm(0u, word, i == 0 ? 0x01000100u : 0x03020302u);                  // halfword i twice
}

// helper, window w: prefix sums of lane's record (its events in `e`, `n` of them; n = 0 behind the chain's last record) and
// the window's cumulative sums.  The Rice lengths of an event for k = 0..5, (e >> k) + 1 + k (rice_coding.rs:56-58), two to a
// register: three packed 16-bit shifts of the event held twice; the constant parts ride along in the running sums' adds.
template <typename ET>
__device__ __forceinline__ void spine3_produce(Spine3LDS &sh, uint32_t w, const RecEvents<ET> &e, uint32_t n, uint32_t rec) {
    const uint32_t lane = lane_id();
    uint32_t *prow = sh.pref[w & 1u] + lane * SP3_ROW;
    uint32_t a01 = 0, a23 = 0, a45 = 0;
    const pk_u16 s01 = {0, 1}, s23 = {2, 3}, s45 = {4, 5};
#pragma unroll
    for (uint32_t t = 0; t < REC; t++) {
        const pk_u16 e2 = __builtin_bit_cast(pk_u16, record_event_twice(e, t));
        a01 += __builtin_bit_cast(uint32_t, (pk_u16)(e2 >> s01)) + 0x00020001u;
        a23 += __builtin_bit_cast(uint32_t, (pk_u16)(e2 >> s23)) + 0x00040003u;
        a45 += __builtin_bit_cast(uint32_t, (pk_u16)(e2 >> s45)) + 0x00060005u;
        prow[t * 3] = a01;
        prow[t * 3 + 1] = a23;
        prow[t * 3 + 2] = a45;
    }
    // the record's totals: the prefix sums at its last real event
    uint32_t t01 = 0, t23 = 0, t45 = 0;
    if (n != 0) {
        t01 = prow[(n - 1) * 3];
        t23 = prow[(n - 1) * 3 + 1];
        t45 = prow[(n - 1) * 3 + 2];
    }
    uint32_t *crow = sh.cumT[w % 3u] + (lane + 1) * SP3_CROW;
    crow[0] = wave_incl_scan(t01 & 0xFFFFu);
    crow[1] = wave_incl_scan(t01 >> 16);
    crow[2] = wave_incl_scan(t23 & 0xFFFFu);
    crow[3] = wave_incl_scan(t23 >> 16);
    crow[4] = wave_incl_scan(t45 & 0xFFFFu);
    crow[5] = wave_incl_scan(t45 >> 16);
    sh.grec[w % 3u][lane] = rec;
}

// walker, window w: Sv = the state at the window's start on entry, at its end on return (lane l: S[l & 7]; entries 6, 7 hold
// nothing that is read).  Returns false if an invariant of the search broke (never seen: it would mean the helper's sums and
// prefix sums disagree).
//
// Written for the length of its dependent instruction chain -- every halving of a chain goes through this loop, one after the
// other, and a lone wave pays 7-10 cycles per dependent instruction, more where a value crosses between the vector and the
// scalar unit (profiles/r05/spine_stamps.txt) -- so:
//   * one crossing per search: the six compares of a record are six subtractions and a signed three-way minimum on the vector
//     unit, ONE compare makes the mask; inside the record three saturating packed subtractions, an OR, one compare;
//   * what does not need the record's prefix sums runs while they are on their way from LDS: the thresholds inside the record
//     (theta_k minus the cumulative sum in front of the record: every lane computes them for ITS record from the exclusive
//     sums it holds, the record f's are read from lane f), and what the previous halving leaves behind for the helper.
__device__ __forceinline__ bool spine3_walk(Spine3LDS &sh, uint32_t w, uint32_t &Sv) {
    const uint32_t lane = lane_id(), l7 = lane & 7u;
    // lane l takes counter l & 7 out of the packed prefix sums at the halving: a mask per register (its own 16-bit field of its
    // own register, nothing of the others), a shift
    const uint32_t sh16 = (l7 & 1u) << 4, field = 0xFFFFu << sh16;
    const uint32_t M01 = l7 < 2 ? field : 0u, M23 = (l7 & 6u) == 2 ? field : 0u, M45 = (l7 & 6u) == 4 ? field : 0u;
    const uint32_t *cT = sh.cumT[w % 3u];
    const uint32_t *pf = sh.pref[w & 1u] + (lane & (REC - 1u)) * 3;
    const uint32_t *cTv = cT + l7;  // row r, this lane's counter (words 6 .. 8 of a row: zero)
    uint32_t *lD = sh.lastD[w & 1u] + l7;
    // the cumulative sums through record `lane` (c) and in front of it (x)
    const uint32_t *cin = cT + (lane + 1) * SP3_CROW, *cex = cT + lane * SP3_CROW;
    const uint32_t c0 = cin[0], c1 = cin[1], c2 = cin[2], c3 = cin[3], c4 = cin[4], c5 = cin[5];
    const uint32_t x0 = cex[0], x1 = cex[1], x2 = cex[2], x3 = cex[3], x4 = cex[4], x5 = cex[5];
    const uint32_t totalv = cTv[64 * SP3_CROW];  // the window's sums (state-vector layout)
    if (lane < 8) lD[0] = Sv;  // row 0: the carry-in (base 0)
    // The recurrence, arranged for depth (a dependent step costs a lone wave ~20 cycles whatever the unit:
    // profiles/r05/dep_latency.txt).  With u = S - base (kept instead of S and base) and, at a halving, nb = the cumulative
    // sums there: S' = (u + nb) >> 1, and the next thresholds theta' = nb + max(1025 - S', 0) = max((nb + 2051 - u) >> 1, nb)
    // (signed, arithmetic shift): three steps behind the prefix sums at the halving instead of six.
    const uint32_t pf_at = (uint32_t)(uintptr_t)(const lds_u32 *)pf, cTv_at = (uint32_t)(uintptr_t)(const lds_u32 *)cTv;  // LDS addresses
    uint32_t uv = Sv;       // S - base
    uint32_t basev = 0;
    uint64_t hm = 0;
    uint32_t lost = 0;  // a search inside a record found nothing: the helper's sums and prefix sums disagree (reported, not acted on)
    // (bounded: a window holds at most 1024 halvings -- one per event -- and a wave that spins on a broken invariant takes
    // the GPU with it)
    int guard = 64 * REC;  // (goes negative when the bound is reached: no record "reaches" anything then, the loop ends)
    uint32_t t01, t23, t45;
    // the records whose six cumulative sums have all reached theta: the first one holds the next halving.  Behind the compare
    // (in the shadow of its way to the scalar unit and back): the thresholds inside every lane's own record -- theta_k minus the
    // cumulative sum in front of the record, two to a register -- of which record f's will be read from lane f.
    auto reached = [&](uint32_t theta) {
        const uint32_t T0 = readlane(theta, 0), T1 = readlane(theta, 1), T2 = readlane(theta, 2);
        const uint32_t T3 = readlane(theta, 3), T4 = readlane(theta, 4), T5 = readlane(theta, 5);
        const int d012 = min(min((int)(c0 - T0), (int)(c1 - T1)), (int)(c2 - T2));
        const int d345 = min(min((int)(c3 - T3), (int)(c4 - T4)), (int)(c5 - T5));
        const uint64_t r = __ballot(min(min(d012, d345), guard) >= 0);
        __builtin_amdgcn_sched_barrier(0);
        const uint32_t a0 = __builtin_elementwise_sub_sat(T0, x0), a1 = __builtin_elementwise_sub_sat(T1, x1);
        const uint32_t a2 = __builtin_elementwise_sub_sat(T2, x2), a3 = __builtin_elementwise_sub_sat(T3, x3);
        const uint32_t a4 = __builtin_elementwise_sub_sat(T4, x4), a5 = __builtin_elementwise_sub_sat(T5, x5);
        t01 = a0 | (a1 << 16);
        t23 = a2 | (a3 << 16);
        t45 = a4 | (a5 << 16);
        return r;
    };
    uint64_t q = reached((uint32_t)max(1025 - (int)Sv, 0));  // theta_k = base_k + max(1025 - S_k, 0), base = 0
    uint32_t pend_row = 0, pend_val = 0;  // what the last halving leaves for the helper: stored while the next one's reads are in flight
    bool pending = false;
    while (q != 0) {
        const uint32_t f = (uint32_t)__builtin_ctzll(q);
        uint32_t prow_at, crow_at;  // (one multiply-add each, on the vector unit: a scalar multiply and a vector add are two steps)
        asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(prow_at) : "s"(f), "v"(SP3_ROW * 4u), "v"(pf_at));
        asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(crow_at) : "s"(f), "v"(SP3_CROW * 4u), "v"(cTv_at));
        const lds_u32 *prow = (const lds_u32 *)(uintptr_t)prow_at;
        const uint32_t p01 = prow[0], p23 = prow[1], p45 = prow[2];
        const uint32_t cprev = *(const lds_u32 *)(uintptr_t)crow_at;  // through record f - 1
        __builtin_amdgcn_sched_barrier(0);  // (the reads are issued; everything below up to the first use of p01 runs in their shadow)
        const uint32_t T01 = readlane(t01, f), T23 = readlane(t23, f), T45 = readlane(t45, f);  // the thresholds inside record f
        if (pending && lane < 8) lD[pend_row] = pend_val;
        hm |= 1ull << f;
        guard--;
        const uint32_t u2051 = 2051u - uv;
        __builtin_amdgcn_sched_barrier(0);
        const uint32_t B1 = cprev + u2051, ucp = cprev + uv;  // (beside the search: they need cprev only)
        const pk_u16 z01 = __builtin_elementwise_sub_sat(__builtin_bit_cast(pk_u16, T01), __builtin_bit_cast(pk_u16, p01));
        const pk_u16 z23 = __builtin_elementwise_sub_sat(__builtin_bit_cast(pk_u16, T23), __builtin_bit_cast(pk_u16, p23));
        const pk_u16 z45 = __builtin_elementwise_sub_sat(__builtin_bit_cast(pk_u16, T45), __builtin_bit_cast(pk_u16, p45));
        const uint32_t z = __builtin_bit_cast(uint32_t, z01) | __builtin_bit_cast(uint32_t, z23) | __builtin_bit_cast(uint32_t, z45);
        // the record's events at which all six prefix sums have reached their thresholds (lanes 16 .. 63 repeat lanes 0 .. 15: the
        // lowest set bit is the first such event)
        const uint32_t m = (uint32_t)__ballot(z == 0);
        lost |= ((m & 0xFFFFu) - 1u) >> 31;  // m == 0
        const uint32_t ts = (uint32_t)__builtin_ctz(m) & 31u;
        const uint32_t q01 = readlane(p01, ts), q23 = readlane(p23, ts), q45 = readlane(p45, ts);
        const uint32_t Pf = ((q01 & M01) | (q23 & M23) | (q45 & M45)) >> sh16;
        const uint32_t nb = cprev + Pf;   // the window's cumulative sums at the halving
        const uint32_t theta = (uint32_t)max((int)(B1 + Pf) >> 1, (int)nb);
        Sv = (ucp + Pf) >> 1;             // x /= 2 on every counter (parameter_selection.rs:62)
        uv = Sv - nb;
        basev = nb;
        pend_row = (uint32_t)__builtin_amdgcn_readfirstlane((int)((f + 1) * 8));
        pend_val = uv;
        pending = true;
        q = reached(theta);
    }
    if (pending && lane < 8) lD[pend_row] = pend_val;
    const bool ok = lost == 0 && guard >= 0;
    Sv += totalv - basev;
#ifdef FELICS_SPINE_STAMPS
    if (lane == 0) atomicAdd(&g_spine_stamps[6], (unsigned long long)(64 * REC - guard));  // halvings of ALL chains
#endif
    if (lane == 0) {
        sh.hmask[w & 1u][0] = (uint32_t)hm;
        sh.hmask[w & 1u][1] = (uint32_t)(hm >> 32);
    }
    return ok;
}

// helper, window w (after it was walked): the start state of lane's record, stored AT the record's place (its slot group over
// the whole sub-batch: k_assign3 takes the records in tile order)
__device__ __forceinline__ void spine3_finish(Spine3LDS &sh, uint32_t w, uint32_t nvalid /* records of the window */,
                                              uint4 *__restrict__ state16 /* [record's place] */) {
    const uint32_t lane = lane_id();
    const uint32_t *cT = sh.cumT[w % 3u] + lane * SP3_CROW;  // row lane = through record lane - 1
    const uint64_t hm = ((uint64_t)sh.hmask[w & 1u][1] << 32) | sh.hmask[w & 1u][0];
    const uint64_t below = hm & lanemask_lt();
    const uint32_t row = below ? 64u - (uint32_t)__builtin_clzll(below) : 0u;  // record r -> row r + 1
    const uint32_t *lD = sh.lastD[w & 1u] + row * 8;
    const uint32_t s0 = lD[0] + cT[0], s1 = lD[1] + cT[1], s2 = lD[2] + cT[2];
    const uint32_t s3 = lD[3] + cT[3], s4 = lD[4] + cT[4], s5 = lD[5] + cT[5];
    const uint32_t place = sh.grec[w % 3u][lane];
    if (lane < nvalid) state16[place] = make_uint4((s0 & 0xFFFFu) | (s1 << 16), (s2 & 0xFFFFu) | (s3 << 16), (s4 & 0xFFFFu) | (s5 << 16), place);
}

template <typename ET>
__global__ __launch_bounds__(128) void k_spine3(const ET *__restrict__ ev, const uint2 *__restrict__ desc, const uint2 *__restrict__ chain_seg,
                                                uint32_t *__restrict__ chain_state, uint4 *__restrict__ state16, uint32_t nchains,
                                                uint32_t *__restrict__ flags) {
    __shared__ Spine3LDS sh;
    if (blockIdx.x >= nchains) return;
    constexpr uint32_t NC = nctx_of<ET>();
    // Workgroup w -> (context w / nplanes, plane w % nplanes): the long chains (small contexts) of all planes start first
    const uint32_t nplanes = nchains / NC;
    const uint32_t ctx = blockIdx.x / nplanes, plane = blockIdx.x % nplanes;
    const uint32_t chain = plane * NC + ctx;
    const uint2 seg = chain_seg[chain];
    const uint32_t nrec = (uint32_t)__builtin_amdgcn_readfirstlane((int)seg.y);
    if (nrec == 0) return;
    const uint32_t rec0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)seg.x);
    const uint32_t lane = lane_id(), l7 = lane & 7u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t nwin = (nrec + 63u) >> 6;
    const uint2 *dsc = desc + rec0;
    uint32_t *cstate = chain_state + (uint64_t)chain * 8;
    // cumT: row 0 of every buffer is zeros (the window's sums in front of its first record), and so are the words behind the
    // six counters in every row (the walker reads a row as a state vector of eight); nobody writes them again
    if (wave == 0) {
        for (uint32_t i = lane; i < 3 * 66; i += 64) {
            uint32_t *row = sh.cumT[i / 66] + (i % 66) * SP3_CROW;
            row[6] = 0;
            row[7] = 0;
            row[8] = 0;
            if (i % 66 == 0) {
#pragma unroll
                for (uint32_t k = 0; k < 6; k++) row[k] = 0;
            }
        }
    }
    auto fetch_desc = [&](uint32_t w) {
        const uint32_t r = w * 64 + lane;
        return r < nrec ? dsc[r] : make_uint2(0u, 0u);  // (behind the last record: no events; record 0 of the batch is there to be read)
    };

    if (nwin < SP3_MULTI_MIN) {
        // ---- one wave, window by window
        if (wave != 0) return;
        uint32_t Sv = l7 < 6 ? cstate[l7] : 0u;
        bool ok = true;
        for (uint32_t w = 0; w < nwin; w++) {
            const uint2 d = fetch_desc(w);
            RecEvents<ET> e;
            load_record(ev, d.x, e);
            __builtin_amdgcn_wave_barrier();  // (the window's buffers are free: the same wave has finished with them)
            spine3_produce<ET>(sh, 0, e, d.y, d.x);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            ok = spine3_walk(sh, 0, Sv) && ok;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            spine3_finish(sh, 0, min(64u, nrec - w * 64), state16);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        if (lane < 6) cstate[lane] = Sv;
        if (!ok && lane == 0) atomicOr(flags, TL_FLAG_SPINE);
        return;
    }

    // ---- walker (wave 0) and helper (wave 1); iteration `it`: the helper finishes window it - 2 and produces window it, the
    // walker walks window it - 1
    if (wave == 0) {
        uint32_t Sv = l7 < 6 ? cstate[l7] : 0u;
        asm volatile("; state in %0" : "+v"(Sv));
        __builtin_amdgcn_s_setprio(3);  // a chain is one long dependent instruction stream: never make it wait for issue
        bool ok = true;
#ifdef FELICS_SPINE_STAMPS
        const bool stamped = ctx == 1 && plane == 0;
        unsigned long long st_acc[4] = {}, st_last = __builtin_amdgcn_s_memtime();
        const unsigned long long st_begin = st_last;
#endif
        for (uint32_t it = 0; it < nwin + 2; it++) {
            if (it >= 1 && it <= nwin) ok = spine3_walk(sh, it - 1, Sv) && ok;
            SP3_STAMP(1);
            __syncthreads();
            SP3_STAMP(0);
        }
#ifdef FELICS_SPINE_STAMPS
        if (stamped && lane == 0) {
            atomicAdd(&g_spine_stamps[0], st_acc[0]);
            atomicAdd(&g_spine_stamps[1], st_acc[1]);
            atomicAdd(&g_spine_stamps[5], (unsigned long long)nwin);
            atomicAdd(&g_spine_stamps[7], __builtin_amdgcn_s_memtime() - st_begin);
        }
#endif
        if (lane < 6) cstate[lane] = Sv;
        if (!ok && lane == 0) atomicOr(flags, TL_FLAG_SPINE);
    } else {
        __builtin_amdgcn_s_setprio(2);  // the walker waits for this wave at every hand-over: in front of the other kernels' waves on its SIMD
        // The events of a window are asked for three hand-overs before they are needed, their places (desc) five: with the GPU full
        // of other kernels a load takes microseconds, and a helper that waits for its events holds up the walker (one window
        // ahead, the helper of the longest chain spent 7 250 cycles per window beside a full batch and 2 500 alone:
        // profiles/r05/spine_stamps.txt).
        uint2 d0 = fetch_desc(0), d1 = fetch_desc(1), d2 = fetch_desc(2), d3 = fetch_desc(3), d4 = fetch_desc(4);
        RecEvents<ET> e0, e1, e2, e3;
        load_record(ev, d0.x, e0);
        load_record(ev, d1.x, e1);
        load_record(ev, d2.x, e2);
#ifdef FELICS_SPINE_STAMPS
        const bool stamped = ctx == 1 && plane == 0;
        unsigned long long st_acc[4] = {}, st_last = __builtin_amdgcn_s_memtime();
#endif
        for (uint32_t it = 0; it < nwin + 2; it++) {
            const uint2 d5 = fetch_desc(it + 5);  // (windows past the last: no records, nothing read)
            load_record(ev, d3.x, e3);
            if (it >= 2) spine3_finish(sh, it - 2, min(64u, nrec - (it - 2) * 64), state16);
            if (it < nwin) spine3_produce<ET>(sh, it, e0, d0.y, d0.x);
            e0 = e1;
            e1 = e2;
            e2 = e3;
            d0 = d1;
            d1 = d2;
            d2 = d3;
            d3 = d4;
            d4 = d5;
            SP3_STAMP(2);
            __syncthreads();
            SP3_STAMP(3);
        }
#ifdef FELICS_SPINE_STAMPS
        if (stamped && lane == 0) {
            atomicAdd(&g_spine_stamps[2], st_acc[2]);
            atomicAdd(&g_spine_stamps[3], st_acc[3]);
        }
#endif
    }
}

template <typename ET>
void launch_spine3(hipStream_t s, const ET *ev, const ChainSlice &cs, uint32_t *chain_state, uint32_t *flags, const Geometry &g) {
    const uint32_t nchains = g.nplanes * g.nctx;
    FELICS_LAUNCH((k_spine3<ET>), dim3(nchains), dim3(128), s, ev, cs.desc, cs.chain_seg, chain_state, cs.state16, nchains, flags);
}
template void launch_spine3<uint8_t>(hipStream_t, const uint8_t *, const ChainSlice &, uint32_t *, uint32_t *, const Geometry &);
template void launch_spine3<uint16_t>(hipStream_t, const uint16_t *, const ChainSlice &, uint32_t *, uint32_t *, const Geometry &);

// ------------------------------------------------------------------------------------------
// k_assign3: k of every event, one LANE per record, records in TILE order.
//
// A lane loads its record's start state (k_spine3 left it at the record's place: state16[slot / REC]) and its 16 events and
// replays the estimator event by event (parameter_selection.rs:49-85): k = argmin of the six counters, ties to the largest k
// (`<=` at :79), taken BEFORE the update (compression.rs:127,139); update; halve when the minimum exceeds 1024.  No
// cross-lane operation.  The record's 16 k bytes leave as one 16-byte store into the tile's k bytes (kq[slot]), which the
// pack stage reads back contiguously.  Slots behind a run's last event are replayed like events: their k is never read
// (pix = 0xFFFF there) and the state they leave goes nowhere (the next record has its own start state).
// Every 16 slots in use of a tile are one record of one run (runs start on multiples of REC), so the records of a tile
// are simply its slot groups [0, tile_slots / REC): a wave's loads and stores cover consecutive groups of a tile, whatever
// the chains look like (in chain order, content with short runs -- noise: one record per tile and context -- had every
// lane on a cache line of its own: 1.45 ms per 64 noise frames against 0.24 on S1).  A workgroup takes ASSIGN_TILES
// consecutive (plane, tile) items of the slice and spreads their groups over its lanes.
// ------------------------------------------------------------------------------------------
constexpr uint32_t ASSIGN_TILES = 8;
template <typename ET>
__global__ __launch_bounds__(256) void k_assign3(const ET *__restrict__ ev, const uint4 *__restrict__ state16,
                                                 const uint32_t *__restrict__ tile_slots, uint8_t *__restrict__ kq, uint32_t tile_begin,
                                                 uint32_t slice_tiles, uint32_t sort_tiles, uint32_t tile_groups, uint32_t nitems) {
    uint32_t first[ASSIGN_TILES + 1], base[ASSIGN_TILES];  // (uniform: scalar registers)
    first[0] = 0;
#pragma unroll
    for (uint32_t j = 0; j < ASSIGN_TILES; j++) {
        const uint32_t item = blockIdx.x * ASSIGN_TILES + j;
        uint32_t n = 0;
        base[j] = 0;
        if (item < nitems) {
            const uint32_t plane = item / slice_tiles, tile = tile_begin + item % slice_tiles;
            const uint32_t pt = plane * sort_tiles + tile;
            n = min(tile_slots[pt] / REC, tile_groups);
            base[j] = pt * tile_groups;
        }
        first[j + 1] = first[j] + n;
    }
    for (uint32_t i = threadIdx.x; i < first[ASSIGN_TILES]; i += 256) {
        uint32_t rec = base[0] + i;
#pragma unroll
        for (uint32_t j = 1; j < ASSIGN_TILES; j++) rec = i >= first[j] ? base[j] + (i - first[j]) : rec;
        const uint4 st = state16[rec];
        RecEvents<ET> e;
        load_record(ev, rec, e);
        EstKeys est;
        est.set(st.x & 0xFFFFu, st.x >> 16, st.y & 0xFFFFu, st.y >> 16, st.z & 0xFFFFu, st.z >> 16);
        uint32_t kw[4];
#pragma unroll
        for (uint32_t d = 0; d < 4; d++) {
            uint32_t kk = 0;
#pragma unroll
            for (uint32_t b = 0; b < 4; b++) kk |= est.step(record_event(e, d * 4 + b)) << (8u * b);
            kw[d] = kk ^ 0x07070707u;  // 7 - (7 - k) in every byte
        }
        *reinterpret_cast<uint4 *>(kq + (uint64_t)rec * REC) = make_uint4(kw[0], kw[1], kw[2], kw[3]);
    }
}

template <typename ET>
void launch_assign3(hipStream_t s, const TileLocal<ET> &tl, const uint4 *state16, const Geometry &g, uint32_t tile_begin, uint32_t tile_end) {
    if (tile_end <= tile_begin || g.nplanes == 0) return;
    const uint32_t slice_tiles = tile_end - tile_begin, nitems = slice_tiles * g.nplanes;
    FELICS_LAUNCH((k_assign3<ET>), dim3(cdiv_u(nitems, ASSIGN_TILES)), dim3(256), s, tl.ev, state16, tl.tile_slots, tl.kq, tile_begin, slice_tiles,
                  g.sort_tiles, tl.cap / REC, nitems);
}
template void launch_assign3<uint8_t>(hipStream_t, const TileLocal<uint8_t> &, const uint4 *, const Geometry &, uint32_t, uint32_t);
template void launch_assign3<uint16_t>(hipStream_t, const TileLocal<uint16_t> &, const uint4 *, const Geometry &, uint32_t, uint32_t);

// Two-pass pack only (exact placement after a slot overflow, FELICS_TWO_PASS, or after a look-back gave up): k from the
// tiles' slots to a byte per pixel, k_map[plane * npix + tile * SORT_TILE + pix[slot]] = kq[slot].
__global__ __launch_bounds__(256) void k_k_to_pixels_tl(const uint8_t *__restrict__ kq, const uint16_t *__restrict__ pix,
                                                        const uint32_t *__restrict__ tile_slots, uint32_t cap, uint8_t *__restrict__ k_map,
                                                        uint32_t npix, uint32_t ntiles) {
    const uint32_t tile = blockIdx.x, plane = blockIdx.y;
    const uint64_t pt = (uint64_t)plane * ntiles + tile;
    const uint32_t ns = tile_slots[pt];
    uint8_t *dst = k_map + (uint64_t)plane * npix + (uint64_t)tile * SORT_TILE;
    for (uint32_t s = threadIdx.x; s < ns; s += 256) {
        const uint32_t p = pix[pt * cap + s];
        if (p != 0xFFFFu) dst[p & 0xFFFu] = kq[pt * cap + s];  // (bit 12: the above flag, for the single-pass pack)
    }
}

void launch_k_to_pixels_tl(hipStream_t s, const uint8_t *kq, const uint16_t *pix, const uint32_t *tile_slots, uint32_t cap, uint8_t *k_map,
                           const Geometry &g) {
    if (g.sort_tiles == 0 || g.nplanes == 0) return;
    FELICS_LAUNCH(k_k_to_pixels_tl, dim3(g.sort_tiles, g.nplanes), dim3(256), s, kq, pix, tile_slots, cap, k_map, g.npix, g.sort_tiles);
}

}  // namespace felics
