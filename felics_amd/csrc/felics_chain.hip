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
// k_enum: one wave per (chain, slice).  The chain of context c of plane p in this slice = the runs (tile, c) of the
// slice's tiles in tile order, each cut into records of REC events.  Lane = tile (64 per round): records per tile from the
// run table's column c, a wave scan for their places, one atomic per non-empty chain for its place in the slice's region.
// Workgroup b -> four neighbouring contexts of one plane (their table entries share cache lines), all of plane p's
// workgroups on the XCD p % 8 whose front workgroups wrote that plane's table (placement only).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_enum(const uint32_t *__restrict__ runtab, uint2 *__restrict__ desc, uint2 *__restrict__ chain_seg,
                                              uint32_t *__restrict__ slice_nrec, uint32_t ntiles, uint32_t t0, uint32_t t1, uint32_t nplanes,
                                              uint32_t nctx, uint32_t cap_rec) {
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = lane_id();
    const uint32_t gpp = nctx / 4;  // workgroups per plane
    const uint32_t item = blockIdx.x >> 3, xcd = blockIdx.x & 7u;
    const uint32_t plane = xcd + 8u * (item / gpp);
    if (plane >= nplanes) return;
    const uint32_t ctx = (item % gpp) * 4 + wave;
    const uint32_t chain = plane * nctx + ctx;
    const uint32_t *col = runtab + (uint64_t)plane * ntiles * nctx + ctx;
    uint32_t total = 0;
    for (uint32_t tb = t0; tb < t1; tb += 256) {  // four rounds of loads in flight
        uint32_t e[4];
#pragma unroll
        for (uint32_t u = 0; u < 4; u++) {
            const uint32_t t = tb + u * 64 + lane;
            e[u] = t < t1 ? col[(uint64_t)t * nctx] : 0u;
        }
#pragma unroll
        for (uint32_t u = 0; u < 4; u++) total += ((e[u] >> 16) + REC - 1) / REC;
    }
    total = readlane(wave_incl_scan(total), 63);
    if (total == 0) {
        if (lane == 0) chain_seg[chain] = make_uint2(0u, 0u);
        return;
    }
    uint32_t base = 0;
    if (lane == 0) {
        base = atomicAdd(slice_nrec, total);
        chain_seg[chain] = make_uint2(base, total);
    }
    base = readlane(base, 0);
    uint32_t run = base;
    for (uint32_t tb = t0; tb < t1; tb += 64) {
        const uint32_t t = tb + lane;
        const uint32_t e = t < t1 ? col[(uint64_t)t * nctx] : 0u;
        const uint32_t n = e >> 16, nr = (n + REC - 1) / REC;
        const uint32_t incl = wave_incl_scan(nr);
        const uint32_t at = run + incl - nr;
        const uint32_t r0 = (plane * ntiles + t) * cap_rec + (e & 0xFFFFu);  // first record of the run, over the whole sub-batch
        for (uint32_t j = 0; j < nr; j++) desc[at + j] = make_uint2(r0 + j, min(REC, n - j * REC));
        run += readlane(incl, 63);
    }
}

void launch_enum(hipStream_t s, const uint32_t *runtab, const ChainSlice &cs, const Geometry &g, uint32_t tile_begin, uint32_t tile_end,
                 uint32_t cap) {
    if (tile_end <= tile_begin) return;
    const dim3 grid(8u * cdiv_u(g.nplanes, 8) * (g.nctx / 4));
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

// helper, window w: prefix sums of lane's record (its events in `e`, `n` of them; n = 0 behind the chain's last record) and
// the window's cumulative sums
template <typename ET>
__device__ __forceinline__ void spine3_produce(Spine3LDS &sh, uint32_t w, const RecEvents<ET> &e, uint32_t n, uint32_t rec) {
    const uint32_t lane = lane_id();
    uint32_t *prow = sh.pref[w & 1u] + lane * SP3_ROW;
    uint32_t a01 = 0, a23 = 0, a45 = 0;
#pragma unroll
    for (uint32_t t = 0; t < REC; t++) {
        uint32_t l01, l23, l45;
        packed_lengths(record_event(e, t), l01, l23, l45);
        a01 += l01;
        a23 += l23;
        a45 += l45;
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

// walker, window w: Sv = the state at the window's start on entry, at its end on return.  Returns false if an invariant of
// the search broke (never seen: it would mean the helper's sums and prefix sums disagree).
__device__ __forceinline__ bool spine3_walk(Spine3LDS &sh, uint32_t w, uint32_t &Sv) {
    const uint32_t lane = lane_id(), l7 = lane & 7u;
    const uint32_t sh16 = (l7 & 1u) << 4;
    const uint32_t *cT = sh.cumT[w % 3u];
    const uint32_t *pf = sh.pref[w & 1u];
    uint32_t *lD = sh.lastD[w & 1u];
    const uint32_t c0 = cT[(lane + 1) * SP3_CROW], c1 = cT[(lane + 1) * SP3_CROW + 1], c2 = cT[(lane + 1) * SP3_CROW + 2];
    const uint32_t c3 = cT[(lane + 1) * SP3_CROW + 3], c4 = cT[(lane + 1) * SP3_CROW + 4], c5 = cT[(lane + 1) * SP3_CROW + 5];
    const uint32_t totalv = l7 < 6 ? cT[64 * SP3_CROW + l7] : 0u;  // the window's sums (state-vector layout)
    if (lane < 8) lD[lane] = Sv;  // row 0: the carry-in (base 0)
    uint32_t basev = 0;
    uint64_t hm = 0;
    bool ok = true;
    // (bounded: a window holds at most 1024 halvings -- one per event -- and a wave that spins on a broken invariant takes
    // the GPU with it)
    for (uint32_t guard = 0; guard < 64 * REC + 1; guard++) {
        // theta_k = base_k + max(1025 - S_k, 0)
        const uint32_t theta = basev + (uint32_t)max(1025 - (int)Sv, 0);
        const uint32_t T0 = readlane(theta, 0), T1 = readlane(theta, 1), T2 = readlane(theta, 2);
        const uint32_t T3 = readlane(theta, 3), T4 = readlane(theta, 4), T5 = readlane(theta, 5);
        const uint64_t q = __ballot(c0 >= T0 && c1 >= T1 && c2 >= T2 && c3 >= T3 && c4 >= T4 && c5 >= T5);
        if (q == 0) break;  // no further halving in this window
        const uint32_t f = (uint32_t)__builtin_ctzll(q);
        const uint32_t *prow = pf + f * SP3_ROW + (lane & (REC - 1u)) * 3;
        const uint32_t p01 = prow[0], p23 = prow[1], p45 = prow[2];
        const uint32_t cprev = l7 < 6 ? cT[f * SP3_CROW + l7] : 0u;  // through record f - 1
        // thresholds inside the record, two to a register: lane 0 -> k = 0, 1; lane 2 -> 2, 3; lane 4 -> 4, 5
        const uint32_t thp = (uint32_t)max((int)(theta - cprev), 0);
        const uint32_t up = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)thp, 0x101, 0xF, 0xF, true);  // row_shl:1
        const uint32_t th2 = thp | (up << 16);
        const uint32_t T01 = readlane(th2, 0), T23 = readlane(th2, 2), T45 = readlane(th2, 4);
        const uint64_t m = __ballot(pk_all_ge(p01, T01) && pk_all_ge(p23, T23) && pk_all_ge(p45, T45)) & 0xFFFFull;
        if (m == 0) {
            ok = false;
            break;
        }
        const uint32_t ts = (uint32_t)__builtin_ctzll(m);
        const uint32_t q01 = readlane(p01, ts), q23 = readlane(p23, ts), q45 = readlane(p45, ts);
        const uint32_t qv = l7 < 2 ? q01 : l7 < 4 ? q23 : q45;
        const uint32_t Pf = l7 < 6 ? (qv >> sh16) & 0xFFFFu : 0u;
        const uint32_t nb = cprev + Pf;   // the window's cumulative sums at the halving
        Sv = (Sv + nb - basev) >> 1;      // x /= 2 on every counter (parameter_selection.rs:62)
        basev = nb;
        if (lane < 8) lD[(f + 1) * 8 + lane] = Sv - basev;
        hm |= 1ull << f;
    }
    Sv += totalv - basev;
    if (lane == 0) {
        sh.hmask[w & 1u][0] = (uint32_t)hm;
        sh.hmask[w & 1u][1] = (uint32_t)(hm >> 32);
    }
    return ok;
}

// helper, window w (after it was walked): the start state of lane's record, stored with the record's place
__device__ __forceinline__ void spine3_finish(Spine3LDS &sh, uint32_t w, uint32_t nvalid /* records of the window */,
                                              uint4 *__restrict__ out /* of the window's first record */) {
    const uint32_t lane = lane_id();
    const uint32_t *cT = sh.cumT[w % 3u] + lane * SP3_CROW;  // row lane = through record lane - 1
    const uint64_t hm = ((uint64_t)sh.hmask[w & 1u][1] << 32) | sh.hmask[w & 1u][0];
    const uint64_t below = hm & lanemask_lt();
    const uint32_t row = below ? 64u - (uint32_t)__builtin_clzll(below) : 0u;  // record r -> row r + 1
    const uint32_t *lD = sh.lastD[w & 1u] + row * 8;
    const uint32_t s0 = lD[0] + cT[0], s1 = lD[1] + cT[1], s2 = lD[2] + cT[2];
    const uint32_t s3 = lD[3] + cT[3], s4 = lD[4] + cT[4], s5 = lD[5] + cT[5];
    if (lane < nvalid)
        out[lane] = make_uint4((s0 & 0xFFFFu) | (s1 << 16), (s2 & 0xFFFFu) | (s3 << 16), (s4 & 0xFFFFu) | (s5 << 16), sh.grec[w % 3u][lane]);
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
    uint4 *out = state16 + rec0;
    uint32_t *cstate = chain_state + (uint64_t)chain * 8;
    // row 0 of every cumT buffer: zeros (the window's sums in front of its first record)
    if (wave == 0 && lane < 3 * SP3_CROW) sh.cumT[lane / SP3_CROW][lane % SP3_CROW] = 0;
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
            spine3_finish(sh, 0, min(64u, nrec - w * 64), out + (uint64_t)w * 64);
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
        for (uint32_t it = 0; it < nwin + 2; it++) {
            if (it >= 1 && it <= nwin) ok = spine3_walk(sh, it - 1, Sv) && ok;
            __syncthreads();
        }
        if (lane < 6) cstate[lane] = Sv;
        if (!ok && lane == 0) atomicOr(flags, TL_FLAG_SPINE);
    } else {
        uint2 dcur = fetch_desc(0), dnxt = fetch_desc(1);
        RecEvents<ET> ecur, enxt;
        load_record(ev, dcur.x, ecur);
        for (uint32_t it = 0; it < nwin + 2; it++) {
            const uint2 dnn = fetch_desc(it + 2);  // (windows past the last: no records, nothing read)
            load_record(ev, dnxt.x, enxt);
            if (it >= 2) spine3_finish(sh, it - 2, min(64u, nrec - (it - 2) * 64), out + (uint64_t)(it - 2) * 64);
            if (it < nwin) spine3_produce<ET>(sh, it, ecur, dcur.y, dcur.x);
            ecur = enxt;
            dcur = dnxt;
            dnxt = dnn;
            __syncthreads();
        }
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
// k_assign3: k of every event, one LANE per record.
//
// A lane loads its record's start state (k_spine3) and its 16 events and replays the estimator event by event
// (parameter_selection.rs:49-85): k = argmin of the six counters, ties to the largest k (`<=` at :79), taken BEFORE the
// update (compression.rs:127,139); update; halve when the minimum exceeds 1024.  No cross-lane operation.  The record's 16
// k bytes leave as one 16-byte store into the tile's k bytes (kq[slot]), which the pack stage reads back contiguously.
// Slots behind a run's last event are replayed like events: their k is never read (pix = 0xFFFF there) and the state they
// leave goes nowhere (the next record has its own start state).
// Persistent: a fixed grid strides over the slice's records (their number is only known on the device).
// ------------------------------------------------------------------------------------------
template <typename ET>
__global__ __launch_bounds__(256) void k_assign3(const ET *__restrict__ ev, const uint4 *__restrict__ state16,
                                                 const uint32_t *__restrict__ slice_nrec, uint8_t *__restrict__ kq) {
    const uint32_t n = *slice_nrec;
    const uint32_t stride = gridDim.x * 256;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const uint4 st = state16[i];
        RecEvents<ET> e;
        load_record(ev, st.w, e);
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
        *reinterpret_cast<uint4 *>(kq + (uint64_t)st.w * REC) = make_uint4(kw[0], kw[1], kw[2], kw[3]);
    }
}

template <typename ET>
void launch_assign3(hipStream_t s, const ET *ev, const ChainSlice &cs, uint8_t *kq, const Geometry &g) {
    // persistent: eight workgroups of four waves per CU stride over the records (fewer if there cannot be that many)
    const uint64_t max_rec = (uint64_t)g.nplanes * g.sort_tiles * (tile_cap_max(g.nctx, g.npix) / REC);
    const uint32_t wgs = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(cdiv_u(max_rec, 256), 1u), 256u * 8u);
    FELICS_LAUNCH((k_assign3<ET>), dim3(wgs), dim3(256), s, ev, cs.state16, cs.nrec, kq);
}
template void launch_assign3<uint8_t>(hipStream_t, const uint8_t *, const ChainSlice &, uint8_t *, const Geometry &);
template void launch_assign3<uint16_t>(hipStream_t, const uint16_t *, const ChainSlice &, uint8_t *, const Geometry &);

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
        if (p != 0xFFFFu) dst[p] = kq[pt * cap + s];
    }
}

void launch_k_to_pixels_tl(hipStream_t s, const uint8_t *kq, const uint16_t *pix, const uint32_t *tile_slots, uint32_t cap, uint8_t *k_map,
                           const Geometry &g) {
    if (g.sort_tiles == 0 || g.nplanes == 0) return;
    FELICS_LAUNCH(k_k_to_pixels_tl, dim3(g.sort_tiles, g.nplanes), dim3(256), s, kq, pix, tile_slots, cap, k_map, g.npix, g.sort_tiles);
}

}  // namespace felics
