// felics_api.cpp -- host side of libfelics: the C ABI of include/felics.h on top of the
// gfx950 kernels.  One context = one GPU + four HIP streams + a grow-only workspace in HBM.
//
// Encode is GPU-only by design: there is no CPU encode path in this library.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <type_traits>
#include <string>
#include <new>
#include <vector>

#include "../../include/felics.h"
#include "felics_kernels.h"

using namespace felics;

namespace {

enum Stage { ST_PLANES = 0, ST_HIST, ST_OFFSETS, ST_SCATTER, ST_SPINE, ST_ASSIGN, ST_LENGTHS, ST_BITSCAN, ST_ZERO, ST_PACK,
             ST_WIDE_KEYS, ST_WIDE_SORT, ST_WIDE_CHAINS, ST_COUNT };
const char *kStageNames[ST_COUNT] = {"planes", "hist", "offsets", "scatter", "spine", "assign", "lengths", "bitscan", "zero", "pack",
                                     "wide_keys", "wide_sort", "wide_chains"};
static_assert(ST_COUNT <= FELICS_MAX_STAGES, "felics.h promises at most FELICS_MAX_STAGES stages");

constexpr int SLICES = 12;              // at most; a submission uses lane.nslices of them
constexpr int EV_PAIRS = SLICES + 2;     // launches of one stage per sub-batch that can be timed
constexpr int MAX_LANES = 4;            // upper bound of the submissions in flight (felics_submit_batch_device), each with streams and workspace of its own
constexpr int DEFAULT_LANES = 2;        // what a context uses unless FELICS_LANES says otherwise (measured round 3: 2 lanes x 4 slices 3.03-3.06 ms per step,
                                        // 3 lanes x 3 slices 2.97-3.16, 4 lanes 3.18-3.47: the kernels are issue-bound, so more of them side by side gain nothing)
int lanes_from_env() {
    if (const char *e = getenv("FELICS_LANES")) return std::max(1, std::min(atoi(e), MAX_LANES));
    return DEFAULT_LANES;
}

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

// One pipeline lane: HIP streams, stage events and a workspace in HBM of its own.  A submission (or one pass of
// a huge one) runs on one lane; felics_submit_batch_device hands the lanes out in turn, so that the GPU starts
// on the next batch while it finishes the last pack slices of this one.
struct Lane {
    hipStream_t stream = nullptr;      // spine slices
    hipStream_t front = nullptr;       // planes, hist, offsets, scatter slices
    hipStream_t kstream = nullptr;     // assign slices (k of the events)
    hipStream_t tail = nullptr;        // lengths, bit scan, pack slices
    hipEvent_t slice_done[SLICES] = {};
    hipEvent_t spine_done[SLICES] = {};
    hipEvent_t assign_done[SLICES] = {};
    hipEvent_t ev[ST_COUNT][EV_PAIRS][2] = {};  // profiling: one start/stop pair per launch of a stage
    int ev_used[ST_COUNT] = {};                  // pairs used by the current sub-batch
    hipEvent_t sized = nullptr;       // stream sizes have landed in h_sizes
    hipEvent_t span_begin = nullptr, span_end = nullptr;  // profiling: in front of the sub-batch's first kernel / behind its last byte
    uint64_t *h_sizes = nullptr;      // pinned: image_bytes[n] followed by image_off[n + 1]
    size_t h_sizes_cap = 0;
    // 8-bit samples, tile-local layout (felics_kernels.h): ev / pix_of / k_sorted = the tiles' slots (event value, pixel, k), counts = the
    // run table, tile_slots, desc / block_state = the records of the chains in chain order (place + events, start state), partial =
    // the chains' record ranges per slice, chain_prog = the chains' running state
    DevBuf planes, counts, chain_prog, scalars, evs, pix_of, k_map, k_sorted, block_state, group_bits, tile_slots, desc,
        tile_bits, tile_bitoff, plane_sums, image_bytes, image_off, partial, status, edge_first, edge_last, pscratch;
    DevBuf wrecs[2], wtile_cnt, wmeta, whist, wdigtot, heads, wlong;  // 16-bit samples: event records (sort double buffer), tile counts, plane ranges, digit histograms, chain heads
    uint32_t epoch = 0;               // sub-batches this lane has run: block tags are (epoch, slice)
    // the submission in flight on this lane (felics_submit_batch_device .. felics_wait_batch)
    bool pending = false;
    bool finished = false;            // it took the synchronous path: results are in r_off / r_len / r_rc
    size_t p_n = 0;
    const void *p_pixels = nullptr;
    uint32_t p_w = 0, p_h = 0;
    int p_color = 0, p_depth = 0;
    uint8_t *p_out = nullptr;
    size_t p_cap = 0;
    uint64_t p_slot = 0;
    std::vector<uint64_t> r_off, r_len;
    int r_rc = 0;
    // the sub-batch in flight
    int nslices = SLICES;             // slices its tiles are cut into (see felics_ctx::slices_*)
    bool m_tickets = false;           // the sub-batch's pack kernels took their tiles by ticket (what a look-back failure escalates from)
    bool m_fused = false;             // ... and were the single-pass kernels at all
    bool queued = false;              // this sub-batch came through felics_submit_batch_device (other submissions share the GPU with it)
    Geometry g;
    size_t first_image = 0;
    const void *d_planes = nullptr;
};

}  // namespace

struct felics_ctx {
    int device = -1;
    int next_lane = 0;          // lane of the next felics_submit_batch_device
    int nlanes = DEFAULT_LANES; // lanes in use (FELICS_LANES)
    // Slices per sub-batch: the stages follow each other slice by slice, so more slices let assign / pack start earlier behind the
    // spine -- and every slice costs a launch, a hand-over per stage and a resume of every chain.  Round 5, blocking calls
    // (profiles/r05/experiments.txt): 64 S1 frames 2 / 3 / 4 / 6 / 8 slices 2.95 / 2.68 / 2.75 / 2.79 / 2.76 ms, noise 4.19 / 4.37 /
    // 4.52 / 4.96 / 5.37, one 4K frame 1.99 / 1.93 / 1.95 / 2.04 / 2.15 (round 4's pipeline wanted 6).
    int slices_blocking = 3;    // FELICS_SLICES
    int slices_queued = 2;      // (round 5, tile-local pipeline: 1 slice 3.05, 2 2.54, 3 2.90, 4 2.86, 6 2.82 ms per step with two lanes; round 3 measured 2-4 lanes x 1-6 slices within 3 % of each other: profiles/r03/experiments.txt;
                                // round 4, with k_scatter: 2 slices 2.94, 3 2.82-2.89, 4 2.78-2.80, 6 2.89-2.91, 8 2.96 ms; three lanes 3.06)
    // k_pack_g takes its tiles from the workgroup index while the lanes share the tail stream: one pack kernel then has the
    // look-back to itself.  With a tail stream per lane (FELICS_OWN_TAILS=1), and after a look-back has given up once, tiles are
    // handed out by a ticket counter instead: a tile then only ever waits for tiles held by workgroups that are already running,
    // whatever else shares the GPU.  The counter is one memory-side atomic per tile on one address -- 130 000 per step at the
    // ~88 per microsecond one address sustains (MI355X_MICROARCH.md, dequeue) -- measured 1.59 against 1.26 ms of pack launches per step.
    bool pack_tickets = false;
    bool two_pass = false;      // FELICS_TWO_PASS=1, or a look-back gave up with ticketed tiles as well: lengths + pack kernels
    bool own_tails = false;     // FELICS_OWN_TAILS=1: a tail stream per lane (pack kernels of two submissions side by side, tiles by ticket)
    bool serial = false;        // FELICS_SERIAL=1 (profiling tools: every kernel alone): all stages of a lane on one stream
    bool test_timeout = false;  // FELICS_TEST_TIMEOUT=1: every wait for the GPU reports a time-out (tests of the failed state)
    bool test_lookback = false; // FELICS_TEST_LOOKBACK_FAIL=1: pretend the first single-pass submission gave up (tests)
    // The front kernel ranks a tile's events with returning LDS atomics and CHECKS the order it produced (felics_kernels.hip,
    // k_front); a context whose check fails once ranks with ballots from then on (FELICS_SCATTER=ballot starts that way: tests).
    bool scatter_ballot = false;
    // Slots per tile of the tile-local layout: the default covers anything but adversarial content; a tile that needs more
    // raises TL_FLAG_OVERFLOW, the batch is redone with the worst case and the context keeps to it (FELICS_TEST_TILE_CAP=1
    // starts with a cap so small that the first batch overflows: tests).
    bool cap_max = false;
    bool test_tile_cap = false;
    bool test_scatter_order = false; // FELICS_TEST_SCATTER_ORDER=1: k_scatter reports a violation whatever it produced (tests)
    bool poison = false;        // FELICS_POISON=1: overwrite the workspace before every sub-batch (tests)
    bool trace = false;         // FELICS_TRACE=1: synchronise and report after every stage (debugging)
    int timeout_s = 120;        // FELICS_TIMEOUT_S: give up waiting for a submission after this long
    // A wait for the GPU timed out: kernels of this context may still be running (or never return), so nothing
    // of it may be reused or freed.  Every later call fails with FELICS_E_HIP; the caller should exit (or run
    // further work in a fresh process).
    bool failed = false;
    felics_stats stats = {};
    Lane lanes[MAX_LANES];
    std::string err;
    bool profiling = false;
    float stage_ms[ST_COUNT] = {};
    float span_ms = 0.f;        // profiling: first kernel -> sizes on the host, of the last submission collected
    int stage_launches[ST_COUNT] = {};
    DevBuf in, out;  // staging of the host-pointer entry point: the batch's frames, the chunks' output slots
    hipStream_t copy_in = nullptr, copy_out = nullptr;  // felics_compress_batch: frames to the device / streams back, beside the kernels
    std::vector<hipEvent_t> h2d_done;                   // a chunk's frames have arrived (one per chunk of a host-buffer batch; grown on demand)
    hipEvent_t wait_before_submit = nullptr;            // the next sub-batch's first kernel waits for this event (set around one submit)
    DevBuf own;      // encode_device's own output when the caller gives none (the host entry point's fall-back for a chunk whose streams outgrew their slots)
    DevBuf dec_meta, dec_planes;  // GPU decoder: offsets | lens | status of a batch; Y / Co / Cg planes of RGB streams
    DevBuf dec_lane_table;        // gray streams decoded 64 to a wave: the estimator rows that do not fit in LDS (3 KB per stream, zeroed per call)
    DevBuf dec_table;             // 16-bit streams: estimator tables in HBM (8.4 MB per stream of a pass), zeroed once, rows tagged with an epoch
    uint32_t dec_epoch = 0;       // last epoch handed out (three per call: one per plane)
};

namespace {

int hip_fail(felics_ctx *ctx, hipError_t e, const char *what) {
    char buf[256];
    snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    if (ctx) ctx->err = buf;
    return FELICS_E_HIP;
}

#define HIP_TRY(ctx, call)                                          \
    do {                                                            \
        hipError_t e__ = (call);                                    \
        if (e__ != hipSuccess) return hip_fail(ctx, e__, #call);    \
    } while (0)

// Wait for an event, but not forever: a kernel that does not return must surface as an error
// (FELICS_E_HIP, "timed out"), not as a caller that hangs.
int wait_event(felics_ctx *ctx, hipEvent_t ev, const char *what) {
    const auto t0 = std::chrono::steady_clock::now();
    if (ctx->test_timeout) {  // FELICS_TEST_TIMEOUT=1 (tests): behave as if the GPU did not answer in time
        ctx->err = std::string(what) + ": timed out waiting for the GPU; the context is unusable from here on";
        ctx->failed = true;
        ctx->stats.failed = 1;
        return FELICS_E_HIP;
    }
    for (uint32_t spins = 0;; spins++) {
        const hipError_t e = hipEventQuery(ev);
        if (e == hipSuccess) return FELICS_OK;
        if (e != hipErrorNotReady) return hip_fail(ctx, e, what);
        if (spins > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));
        if ((spins & 1023) == 1023 &&
            std::chrono::steady_clock::now() - t0 > std::chrono::seconds(ctx->timeout_s)) {
            ctx->err = std::string(what) + ": timed out waiting for the GPU; the context is unusable from here on";
            ctx->failed = true;
            ctx->stats.failed = 1;
            return FELICS_E_HIP;
        }
    }
}

// Waits for everything a lane has queued.  The tail stream is shared by the lanes (the single-pass pack
// kernels of two submissions must not run side by side), so this also waits for the other lane's packs:
// used on the synchronous, fallback and error paths only.
int sync_lane(felics_ctx *ctx, Lane &l) {
    if (l.front) HIP_TRY(ctx, hipStreamSynchronize(l.front));
    if (l.stream) HIP_TRY(ctx, hipStreamSynchronize(l.stream));
    if (l.kstream) HIP_TRY(ctx, hipStreamSynchronize(l.kstream));
    if (l.tail) HIP_TRY(ctx, hipStreamSynchronize(l.tail));
    return FELICS_OK;
}

// Grow-only device buffer.  Callers only grow a lane's buffer while that lane is idle.
int reserve(felics_ctx *ctx, DevBuf &b, size_t bytes) {
    if (bytes <= b.cap) return FELICS_OK;
    if (b.p) {
        HIP_TRY(ctx, hipFree(b.p));  // hipFree waits for the device
        b.p = nullptr;
        b.cap = 0;
    }
    size_t want = bytes + bytes / 8 + 256;  // a little slack so near-equal batches do not realloc
    HIP_TRY(ctx, hipMalloc(&b.p, want));
    b.cap = want;
    return FELICS_OK;
}

// served[] is compared against an epoch: a fresh buffer must not match by accident
int reserve_zeroed(felics_ctx *ctx, DevBuf &b, size_t bytes) {
    if (bytes <= b.cap) return FELICS_OK;
    int rc = reserve(ctx, b, bytes);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemset(b.p, 0, b.cap));
    return FELICS_OK;
}

void release(DevBuf &b) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

// Brackets one launch (or a few back-to-back launches) of a stage with HIP events on the stream it runs
// on; a stage's time is the sum over its launches of a sub-batch.
struct StageTimer {
    felics_ctx *ctx;
    Lane &lane;
    int st;
    hipStream_t stream;
    int slot = -1;
    bool exact;  // one kernel launch inside: record that kernel's own begin / end (see LaunchTiming)
    StageTimer(felics_ctx *c, Lane &l, int s, hipStream_t on, bool single_kernel = false)
        : ctx(c), lane(l), st(s), stream(on), exact(single_kernel) {
        if (ctx->profiling && lane.ev_used[st] < EV_PAIRS) {
            slot = lane.ev_used[st]++;
            if (exact)
                g_launch_timing = LaunchTiming{lane.ev[st][slot][0], lane.ev[st][slot][1]};
            else
                (void)hipEventRecord(lane.ev[st][slot][0], stream);
        }
    }
    ~StageTimer() {
        if (slot >= 0) {
            if (!exact) {
                (void)hipEventRecord(lane.ev[st][slot][1], stream);
            } else if (g_launch_timing.start) {  // nothing was launched: give the pair back
                g_launch_timing = LaunchTiming{};
                lane.ev_used[st]--;
            }
        }
        if (ctx->trace) {  // FELICS_TRACE: wait for the stage and say so (locating a kernel that does not return)
            hipError_t e = hipStreamSynchronize(stream);
            fprintf(stderr, "[felics] %s done (%s)\n", kStageNames[st], hipGetErrorString(e));
        }
    }
};

int check_args(uint32_t w, uint32_t h, int color, int depth) {
    if (color != FELICS_COLOR_GRAY && color != FELICS_COLOR_RGB) return FELICS_E_INVALID_COLOR_TYPE;
    if (depth != FELICS_DEPTH_8 && depth != FELICS_DEPTH_16) return FELICS_E_INVALID_PIXEL_DEPTH;
    // compress_channel unwraps width.checked_mul(height) (compression.rs:86): reported, not a panic
    if ((uint64_t)w * h > 0xFFFFFFFFull) return FELICS_E_INVALID_DIMENSIONS;
    return FELICS_OK;
}

void header_bytes(uint8_t *o, uint32_t w, uint32_t h, int color, int depth) {
    memcpy(o, "FLCS", 4);
    o[4] = (uint8_t)color;
    o[5] = (uint8_t)depth;
    for (int i = 0; i < 4; i++) {
        o[6 + i] = (uint8_t)(w >> (24 - 8 * i));
        o[10 + i] = (uint8_t)(h >> (24 - 8 * i));
    }
}

// Everything one sub-batch of 8-bit frames needs, queued without waiting for the host (tile-local layout, felics_kernels.h):
//   front stream : the front kernel (classify + sort a tile's events, once) slice by slice
//   spine stream : behind every front slice the records of its chains (k_enum) and the spine launch that walks them
//   k stream     : behind every spine launch the k of the slice's events (k_assign3)
//   tail stream  : behind every k launch -- when every stream has a fixed slot in the output -- the packed bits of that
//                  slice's tiles (k_pack_t: code lengths, tile offsets by look-back, packing in one kernel); RGB planes 1, 2
//                  go to scratch slots and are moved behind plane 0 at the end (the offset of planes 1 and 2 needs the size
//                  of the planes before them).
// The stream sizes are copied to the lane's pinned buffer and `sized` is recorded behind them.
// slot_stride == 0: no packing here (the caller places the streams exactly once it has the sizes).
template <typename T, typename ET>
int run_lane(felics_ctx *ctx, Lane &l, uint8_t *d_out, uint64_t slot_stride) {
    const Geometry &g = l.g;
    const int ns = l.nslices;
    const size_t nsamples = (size_t)g.nplanes * g.npix;
    int rc = 0;
    // Fixed slots: code lengths, tile offsets and packing in one kernel per slice (k_pack_t; one such kernel at a time unless the
    // tiles are handed out by ticket: the tiles of two of them waiting for each other's queued predecessors could hold all
    // workgroup slots, so the lanes share the tail stream).  Otherwise (exact placement, FELICS_TWO_PASS, after a look-back gave
    // up twice): k to a byte per pixel once every chain is replayed, then the lengths / bit scan / pack kernels over all tiles.
    const bool fused = slot_stride != 0 && !ctx->two_pass;
    const uint32_t cap = ctx->cap_max ? tile_cap_max(g.nctx, g.npix) : ctx->test_tile_cap ? std::min(4u * REC, tile_cap_max(g.nctx, g.npix)) : tile_cap_default(g.nctx, g.npix);
    const size_t ptiles = (size_t)g.nplanes * g.sort_tiles;
    const size_t slots = ptiles * cap, recs = slots / REC;
    const size_t nchains = (size_t)g.nplanes * g.nctx;
    if ((rc = reserve(ctx, l.evs, slots * sizeof(ET) + STAGE_PAD)) != 0) return rc;
    if ((rc = reserve(ctx, l.pix_of, slots * 2 + STAGE_PAD)) != 0) return rc;
    if ((rc = reserve(ctx, l.k_sorted, slots + STAGE_PAD)) != 0) return rc;
    if ((rc = reserve(ctx, l.counts, ptiles * g.nctx * 4)) != 0) return rc;        // the run table
    if ((rc = reserve(ctx, l.tile_slots, ptiles * 4)) != 0) return rc;
    if ((rc = reserve(ctx, l.desc, recs * 8 + 64)) != 0) return rc;
    if ((rc = reserve(ctx, l.block_state, recs * 16 + 64)) != 0) return rc;         // state16
    if ((rc = reserve(ctx, l.partial, (size_t)SLICES * nchains * 8)) != 0) return rc;  // chain_seg per slice
    if ((rc = reserve(ctx, l.chain_prog, nchains * 32)) != 0) return rc;            // chain_state
    if ((rc = reserve(ctx, l.scalars, 64 + 4 * (SLICES + 2) + 4 * SLICES)) != 0) return rc;
    if (!fused && (rc = reserve(ctx, l.k_map, nsamples + STAGE_PAD)) != 0) return rc;
    if (!fused && (rc = reserve(ctx, l.group_bits, (size_t)g.nplanes * g.pack_tiles * PACK_THREADS * 2)) != 0) return rc;
    if ((rc = reserve(ctx, l.tile_bits, (size_t)g.nplanes * g.pack_tiles * 4)) != 0) return rc;
    if ((rc = reserve(ctx, l.tile_bitoff, (size_t)g.nplanes * g.pack_tiles * 8)) != 0) return rc;
    if ((rc = reserve(ctx, l.plane_sums, (size_t)g.nplanes * 16)) != 0) return rc;  // carry[nplanes], base[nplanes]
    if ((rc = reserve(ctx, l.image_bytes, (size_t)g.nimages * 8)) != 0) return rc;
    if ((rc = reserve(ctx, l.image_off, (size_t)(g.nimages + 1) * 8)) != 0) return rc;
    if ((rc = reserve_zeroed(ctx, l.status, (size_t)g.nplanes * g.pack_tiles * 8)) != 0) return rc;
    if ((rc = reserve(ctx, l.edge_first, (size_t)g.nplanes * g.pack_tiles * 4)) != 0) return rc;
    if ((rc = reserve(ctx, l.edge_last, (size_t)g.nplanes * g.pack_tiles * 4)) != 0) return rc;
    const size_t hs = (size_t)g.nimages * 2 + 1;
    if (hs > l.h_sizes_cap) {
        if (l.h_sizes) HIP_TRY(ctx, hipHostFree(l.h_sizes));
        l.h_sizes = nullptr;
        HIP_TRY(ctx, hipHostMalloc((void **)&l.h_sizes, hs * 8 + 64, hipHostMallocDefault));
        l.h_sizes_cap = hs;
    }
    hipStream_t s = l.stream, f = l.front, ks = l.kstream, tl = l.tail;
    if (ctx->serial) f = ks = tl = s;  // FELICS_SERIAL (profiling: every kernel alone): one stream, same order of launches
    const T *d_planes = (const T *)l.d_planes;
    auto *chain_state = (uint32_t *)l.chain_prog.p;
    auto *plane_carry = (uint64_t *)l.plane_sums.p;
    auto *plane_base = plane_carry + g.nplanes;
    if (++l.epoch >= 0x03FFFFFFu) l.epoch = 1;
    if ((l.epoch & 0x3FFFFu) == 0) HIP_TRY(ctx, hipMemsetAsync(l.status.p, 0, l.status.cap, f));  // look-back tags: 18 epoch bits
    const uint32_t epoch = l.epoch;
    PackTarget target{d_out, slot_stride, nullptr, 0};
    if (fused && g.planes_per_image > 1) {
        target.plane_slot = ((uint64_t)g.npix + g.npix / 4 + 64 + 15) & ~15ull;
        if ((rc = reserve(ctx, l.pscratch, (size_t)(target.plane_slot * g.nimages * (g.planes_per_image - 1)))) != 0) return rc;
        target.scratch = (uint8_t *)l.pscratch.p;
    }
    uint32_t *d_error = (uint32_t *)l.scalars.p + 8;  // look-back watchdog of the single-pass pack
    uint32_t *d_flags = (uint32_t *)l.scalars.p + 9;  // TL_FLAG_*: the front kernel's order check and tile overflow, the spine's self-check (read back together with d_error)
    uint32_t *d_tickets = (uint32_t *)l.scalars.p + 16;  // one per pack launch of this sub-batch: tiles are handed out in order
    uint32_t *d_nrec = (uint32_t *)l.scalars.p + 16 + SLICES + 2;  // records per slice
    const TileLocal<ET> tloc{(ET *)l.evs.p, (uint16_t *)l.pix_of.p, (uint8_t *)l.k_sorted.p, (uint32_t *)l.counts.p, (uint32_t *)l.tile_slots.p, cap};
    l.m_tickets = ctx->pack_tickets;
    l.m_fused = fused;

    uint32_t bounds[SLICES + 1];  // slice boundaries in sort tiles (= pack tiles)
    for (int q = 0; q <= ns; q++) bounds[q] = (uint32_t)((uint64_t)g.sort_tiles * q / ns);
    // the records of slice q live in their own region of desc: as many as its tiles can hold
    auto slice_of = [&](int q) {
        const size_t r0 = (size_t)bounds[q] * g.nplanes * (cap / REC);
        return ChainSlice{(uint2 *)l.desc.p + r0, (uint2 *)l.partial.p + (size_t)q * nchains, d_nrec + q, (uint4 *)l.block_state.p};
    };
    // ---- front stream
    if (ctx->poison) {  // FELICS_POISON: every intermediate buffer starts as garbage, as on a fresh context
        DevBuf *bufs[] = {&l.evs, &l.pix_of, &l.k_sorted, &l.counts, &l.tile_slots, &l.desc, &l.block_state, &l.partial, &l.k_map,
                          &l.group_bits, &l.tile_bits, &l.tile_bitoff, &l.edge_first, &l.edge_last, &l.pscratch};
        for (DevBuf *b : bufs)
            if (b->p) HIP_TRY(ctx, hipMemsetAsync(b->p, 0xA5, b->cap, f));
    }
    HIP_TRY(ctx, hipMemsetAsync(chain_state, 0, nchains * 32, f));
    HIP_TRY(ctx, hipMemsetAsync(d_flags, 0, 4, f));
    HIP_TRY(ctx, hipMemsetAsync(d_nrec, 0, 4 * SLICES, f));
    // (FELICS_TEST_SCATTER_ORDER: the atomically ranked kernel reports a violation whatever it produced; the ballot-ranked form is
    // the remedy and is checked for real)
    const uint32_t front_mode = ctx->scatter_ballot ? FRONT_SAFE_RANK : ctx->test_scatter_order ? FRONT_TEST_VIOLATION : 0u;
    if (!ctx->scatter_ballot) ctx->stats.sorted_event_sorts++;
    for (int q = 0; q < ns; q++) {
        if (bounds[q + 1] != bounds[q]) {
            {
                StageTimer t(ctx, l, ST_SCATTER, f, true);
                launch_front<T, ET>(f, d_planes, tloc, g, bounds[q], bounds[q + 1], d_flags, front_mode);
            }
            // the slice's records in chain order: here, not on the spine stream, so that it runs beside the previous slice's walk
            StageTimer t(ctx, l, ST_OFFSETS, f, true);
            launch_enum(f, tloc.runtab, slice_of(q), g, bounds[q], bounds[q + 1], cap);
        }
        HIP_TRY(ctx, hipEventRecord(l.slice_done[q], f));
    }
    // ---- spine stream: the walk along every chain
    for (int q = 0; q < ns; q++) {
        HIP_TRY(ctx, hipStreamWaitEvent(s, l.slice_done[q], 0));
        if (bounds[q + 1] != bounds[q]) {
            StageTimer t(ctx, l, ST_SPINE, s, true);
            launch_spine3<ET>(s, tloc.ev, slice_of(q), chain_state, d_flags, g);
        }
        HIP_TRY(ctx, hipEventRecord(l.spine_done[q], s));
    }
    // ---- k stream: behind every spine launch, k of the slice's events
    for (int q = 0; q < ns; q++) {
        HIP_TRY(ctx, hipStreamWaitEvent(ks, l.spine_done[q], 0));
        if (bounds[q + 1] != bounds[q]) {
            StageTimer t(ctx, l, ST_ASSIGN, ks, true);
            launch_assign3<ET>(ks, tloc, (const uint4 *)l.block_state.p, g, bounds[q], bounds[q + 1]);
        }
        HIP_TRY(ctx, hipEventRecord(l.assign_done[q], ks));
    }
    // ---- tail stream
    HIP_TRY(ctx, hipMemsetAsync(plane_carry, 0, (size_t)g.nplanes * 16, tl));
    HIP_TRY(ctx, hipMemsetAsync(d_error, 0, 4, tl));
    HIP_TRY(ctx, hipMemsetAsync(d_tickets, 0, 4 * (SLICES + 2), tl));
    if (fused) {
        // behind every k launch: the packed bits of that slice's tiles
        for (int q = 0; q < ns; q++) {
            HIP_TRY(ctx, hipStreamWaitEvent(tl, l.assign_done[q], 0));
            if (bounds[q + 1] == bounds[q]) continue;
            StageTimer t(ctx, l, ST_PACK, tl, true);
            launch_pack_t<T>(tl, d_planes, tloc.kq, tloc.pix, tloc.ev, tloc.tile_slots, cap, (uint64_t *)l.status.p, (uint64_t *)l.tile_bitoff.p,
                             (uint32_t *)l.tile_bits.p, plane_carry, (uint32_t *)l.edge_first.p, (uint32_t *)l.edge_last.p, d_error, target, g,
                             bounds[q], bounds[q + 1], epoch, ctx->pack_tickets ? d_tickets + q : nullptr);
        }
        StageTimer t(ctx, l, ST_ZERO, tl);
        launch_finish_sizes(tl, plane_carry, plane_base, (uint64_t *)l.image_bytes.p, g);
        launch_join_edges(tl, (const uint64_t *)l.tile_bitoff.p, (const uint32_t *)l.tile_bits.p,
                          (const uint32_t *)l.edge_first.p, (const uint32_t *)l.edge_last.p, target, g);
        launch_concat_planes(tl, plane_base, plane_carry, target, g);
    } else {
        HIP_TRY(ctx, hipStreamWaitEvent(tl, l.assign_done[ns - 1], 0));
        {
            StageTimer t(ctx, l, ST_ASSIGN, tl, true);
            launch_k_to_pixels_tl(tl, tloc.kq, tloc.pix, tloc.tile_slots, cap, (uint8_t *)l.k_map.p, g);
        }
        {
            StageTimer t(ctx, l, ST_LENGTHS, tl, true);
            launch_lengths<T>(tl, d_planes, (const uint8_t *)l.k_map.p, (uint16_t *)l.group_bits.p,
                              (uint32_t *)l.tile_bits.p, g, 0, g.pack_tiles);
        }
        {
            StageTimer t(ctx, l, ST_BITSCAN, tl);
            launch_bitscan_slice(tl, (const uint32_t *)l.tile_bits.p, (uint64_t *)l.tile_bitoff.p, plane_carry, g, 0, g.pack_tiles);
            launch_finish_sizes(tl, plane_carry, plane_base, (uint64_t *)l.image_bytes.p, g);
        }
        if (slot_stride != 0) {
            {
                StageTimer t(ctx, l, ST_ZERO, tl);
                launch_zero_edges(tl, d_out, nullptr, slot_stride, (const uint64_t *)l.tile_bitoff.p,
                                  (const uint32_t *)l.tile_bits.p, plane_base, g, 0, g.pack_tiles);
            }
            StageTimer t(ctx, l, ST_PACK, tl, true);
            launch_pack<T>(tl, d_planes, (const uint8_t *)l.k_map.p, (const uint16_t *)l.group_bits.p,
                           (const uint64_t *)l.tile_bitoff.p, (const uint32_t *)l.tile_bits.p, plane_base, nullptr,
                           slot_stride, d_out, g, 0, g.pack_tiles);
        }
    }
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(l.h_sizes, l.image_bytes.p, (size_t)g.nimages * 8, hipMemcpyDeviceToHost, tl));
    l.h_sizes[g.nimages] = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&l.h_sizes[g.nimages], d_error, 8, hipMemcpyDeviceToHost, tl));  // d_error | d_flags << 32
    HIP_TRY(ctx, hipEventRecord(l.sized, tl));
    if (ctx->profiling) HIP_TRY(ctx, hipEventRecord(l.span_end, tl));
    return FELICS_OK;
}

// 16-bit samples (T = u16 gray planes, i32 Y/Co/Cg planes): everything on the lane's main stream.
//   keys -> stable sort by (plane, context) -> chain heads -> estimator replay per chain (k_map)
//   -> lengths, bit scan, sizes -> pack (fixed slots) ; same contract as run_lane towards the caller.
template <typename T>
int run_wide(felics_ctx *ctx, Lane &l, uint8_t *d_out, uint64_t slot_stride) {
    const Geometry &g = l.g;
    const size_t nsamples = (size_t)g.nplanes * g.npix;
    const WideSizes z = wide_sizes(g);
    int rc;
    for (int i = 0; i < 2; i++)
        if ((rc = reserve(ctx, l.wrecs[i], z.rec_bytes)) != 0) return rc;
    if ((rc = reserve(ctx, l.wtile_cnt, z.tile_cnt_bytes)) != 0) return rc;
    if ((rc = reserve(ctx, l.wmeta, z.meta_bytes)) != 0) return rc;
    if ((rc = reserve(ctx, l.whist, z.hist_bytes)) != 0) return rc;
    if ((rc = reserve(ctx, l.wdigtot, z.digtot_bytes)) != 0) return rc;
    if ((rc = reserve(ctx, l.heads, z.heads_bytes)) != 0) return rc;
    if ((rc = reserve(ctx, l.scalars, 64)) != 0) return rc;
    uint32_t lane_limit = wide_lane_limit(g);
    if (const char *e = getenv("FELICS_WIDE_LANE")) lane_limit = (uint32_t)std::max(0, atoi(e));  // tests, A/B: 0 = wave-wide only
    const size_t nlong = wide_long_capacity(g, lane_limit);
    if ((rc = reserve(ctx, l.wlong, nlong * (8 + 64) + 64)) != 0) return rc;
    if ((rc = reserve(ctx, l.k_map, nsamples + STAGE_PAD)) != 0) return rc;
    if ((rc = reserve(ctx, l.group_bits, (size_t)g.nplanes * g.pack_tiles * PACK_THREADS * sizeof(group_bits_t<T>))) != 0) return rc;
    if ((rc = reserve(ctx, l.tile_bits, (size_t)g.nplanes * g.pack_tiles * 4)) != 0) return rc;
    if ((rc = reserve(ctx, l.tile_bitoff, (size_t)g.nplanes * g.pack_tiles * 8)) != 0) return rc;
    if ((rc = reserve(ctx, l.plane_sums, (size_t)g.nplanes * 16)) != 0) return rc;
    if ((rc = reserve(ctx, l.image_bytes, (size_t)g.nimages * 8)) != 0) return rc;
    if ((rc = reserve(ctx, l.image_off, (size_t)(g.nimages + 1) * 8)) != 0) return rc;
    const size_t hs = (size_t)g.nimages * 2 + 1;
    if (hs > l.h_sizes_cap) {
        if (l.h_sizes) HIP_TRY(ctx, hipHostFree(l.h_sizes));
        l.h_sizes = nullptr;
        HIP_TRY(ctx, hipHostMalloc((void **)&l.h_sizes, hs * 8 + 64, hipHostMallocDefault));
        l.h_sizes_cap = hs;
    }
    hipStream_t s = l.stream;
    const T *d_planes = (const T *)l.d_planes;
    auto *plane_carry = (uint64_t *)l.plane_sums.p;
    auto *plane_base = plane_carry + g.nplanes;
    auto *nheads = (uint32_t *)l.scalars.p;
    if (ctx->poison) {
        DevBuf *bufs[] = {&l.wrecs[0], &l.wrecs[1], &l.wtile_cnt, &l.wmeta, &l.whist, &l.heads, &l.k_map,
                          &l.group_bits, &l.tile_bits, &l.tile_bitoff};
        for (DevBuf *b : bufs) HIP_TRY(ctx, hipMemsetAsync(b->p, 0xA5, b->cap, s));
    }
    {
        StageTimer t(ctx, l, ST_WIDE_KEYS, s);
        launch_wide_events<T>(s, d_planes, (uint32_t *)l.wtile_cnt.p, (uint32_t *)l.wmeta.p, (uint64_t *)l.wrecs[0].p, g);
    }
    {
        StageTimer t(ctx, l, ST_WIDE_SORT, s);
        launch_wide_sort(s, (uint64_t *)l.wrecs[0].p, (uint64_t *)l.wrecs[1].p, (const uint32_t *)l.wmeta.p, (uint32_t *)l.whist.p,
                         (uint32_t *)l.wdigtot.p, g);
    }
    {
        StageTimer t(ctx, l, ST_WIDE_CHAINS, s);
        HIP_TRY(ctx, hipMemsetAsync(nheads, 0, 8, s));
        launch_wide_chains(s, (const uint64_t *)l.wrecs[0].p, (const uint32_t *)l.wmeta.p, (uint64_t *)l.heads.p, nheads,
                           (uint8_t *)l.k_map.p, g, lane_limit, (uint64_t *)l.wlong.p, (uint32_t *)((uint64_t *)l.wlong.p + nlong));
    }
    HIP_TRY(ctx, hipMemsetAsync(plane_carry, 0, (size_t)g.nplanes * 16, s));
    {
        StageTimer t(ctx, l, ST_LENGTHS, s, true);
        launch_lengths<T>(s, d_planes, (const uint8_t *)l.k_map.p, (group_bits_t<T> *)l.group_bits.p,
                          (uint32_t *)l.tile_bits.p, g, 0, g.pack_tiles);
    }
    {
        StageTimer t(ctx, l, ST_BITSCAN, s);
        launch_bitscan_slice(s, (const uint32_t *)l.tile_bits.p, (uint64_t *)l.tile_bitoff.p, plane_carry, g, 0,
                             g.pack_tiles);
        launch_finish_sizes(s, plane_carry, plane_base, (uint64_t *)l.image_bytes.p, g);
    }
    if (slot_stride != 0) {
        {
            StageTimer t(ctx, l, ST_ZERO, s);
            launch_zero_edges(s, d_out, nullptr, slot_stride, (const uint64_t *)l.tile_bitoff.p,
                              (const uint32_t *)l.tile_bits.p, plane_base, g, 0, g.pack_tiles);
        }
        {
            StageTimer t(ctx, l, ST_PACK, s, true);
            launch_pack<T>(s, d_planes, (const uint8_t *)l.k_map.p, (const group_bits_t<T> *)l.group_bits.p,
                           (const uint64_t *)l.tile_bitoff.p, (const uint32_t *)l.tile_bits.p, plane_base, nullptr,
                           slot_stride, d_out, g, 0, g.pack_tiles);
        }
    }
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(l.h_sizes, l.image_bytes.p, (size_t)g.nimages * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipEventRecord(l.sized, s));
    if (ctx->profiling) HIP_TRY(ctx, hipEventRecord(l.span_end, s));
    return FELICS_OK;
}

// Exact placement: streams back to back at image_off (computed on the device from the sizes), every
// byte of them zeroed, all tiles packed.  Used when the streams do not get fixed slots, and to redo a
// sub-batch in which a stream outgrew its slot.
template <typename T>
int pack_exact(felics_ctx *ctx, Lane &l, uint8_t *d_out) {
    const Geometry &g = l.g;
    hipStream_t s = l.tail;
    auto *plane_base = (uint64_t *)l.plane_sums.p + g.nplanes;
    {
        StageTimer t(ctx, l, ST_ZERO, s);
        launch_zero_streams(s, (uint32_t *)d_out, (const uint64_t *)l.image_off.p, g);
    }
    {
        StageTimer t(ctx, l, ST_PACK, s, true);
        launch_pack<T>(s, (const T *)l.d_planes, (const uint8_t *)l.k_map.p, (const group_bits_t<T> *)l.group_bits.p,
                       (const uint64_t *)l.tile_bitoff.p, (const uint32_t *)l.tile_bits.p, plane_base,
                       (const uint64_t *)l.image_off.p, 0, d_out, g, 0, g.pack_tiles);
    }
    HIP_TRY(ctx, hipGetLastError());
    return FELICS_OK;
}

void collect_timing(felics_ctx *ctx, Lane &l) {
    if (!ctx->profiling) return;
    ctx->span_ms = 0.f;
    (void)hipEventElapsedTime(&ctx->span_ms, l.span_begin, l.span_end);
    for (int i = 0; i < ST_COUNT; i++) {
        ctx->stage_ms[i] = 0.f;  // sum of the launches' durations (launches overlap: the sum can exceed wall time)
        ctx->stage_launches[i] = 0;
        for (int k = 0; k < l.ev_used[i]; k++) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, l.ev[i][k][0], l.ev[i][k][1]) == hipSuccess) {
                ctx->stage_ms[i] += ms;
                ctx->stage_launches[i]++;
            }
        }
    }
}

// images per pass so that slots / chain bases (8-bit) or sample indices and sort keys (16-bit) stay below 2^32
size_t max_images_per_pass(uint64_t npix, uint32_t planes, int depth) {
    const uint64_t per_image = npix * planes;
    if (per_image == 0) return SIZE_MAX;
    if (const char *e = getenv("FELICS_TEST_PASS_IMAGES"))  // tests: several passes without a 100 GB batch
        return (size_t)std::max(1, atoi(e));
    if (depth == FELICS_DEPTH_16) {
        // ~30 bytes of workspace per sample: keep a pass near 2^30 samples -- and near 2^16 planes: the 16-bit front end scans
        // tiles x planes counts in one workgroup and searches the plane table per sort tile (a batch of many tiny frames)
        constexpr uint64_t WIDE_MAX_PLANES = 1u << 16;
        return (size_t)std::max<uint64_t>(1, std::min<uint64_t>(0x40000000ull / per_image, WIDE_MAX_PLANES / planes));
    }
    // the records of a pass are numbered with 32 bits: tiles x records per tile (worst case)
    const uint64_t tiles = (npix + SORT_TILE - 1) / SORT_TILE;
    const uint64_t rec_per_image = tiles * planes * (tile_cap_max(NCTX, (uint32_t)std::min<uint64_t>(npix, SORT_TILE)) / REC);
    return (size_t)std::max<uint64_t>(1, std::min(0xE0000000ull / per_image, 0xE0000000ull / rec_per_image));
}

// Queues one sub-batch (cnt frames starting at frame `first` of d_pixels) on a lane: geometry, colour
// transform, and everything run_lane / run_wide enqueue.  Returns without waiting.
int launch_sub_batch(felics_ctx *ctx, Lane &l, size_t first, size_t cnt, const void *d_pixels, uint32_t w, uint32_t h,
                     int color, int depth, uint8_t *lane_out, uint64_t slot, int nslices, bool queued = false) {
    l.nslices = std::max(1, std::min(nslices, SLICES));
    l.queued = queued;
    ctx->stats.submissions++;
    const uint32_t planes = color == FELICS_COLOR_RGB ? 3 : 1;
    const uint64_t npix = (uint64_t)w * h;
    const bool wide = depth == FELICS_DEPTH_16;
    const size_t frame_bytes = (size_t)npix * planes * (wide ? 2 : 1);
    int rc;
    Geometry &g = l.g;
    g.W = w;
    g.H = h;
    g.npix = (uint32_t)npix;
    g.nimages = (uint32_t)cnt;
    g.planes_per_image = planes;
    g.nplanes = (uint32_t)(cnt * planes);
    g.sort_tiles = (uint32_t)((npix + SORT_TILE - 1) / SORT_TILE);
    g.pack_tiles = (uint32_t)((npix + PACK_TILE - 1) / PACK_TILE);
    g.color = (uint32_t)color;
    g.depth = (uint32_t)depth;
    g.nctx = planes == 3 ? nctx_of<int16_t>() : nctx_of<uint8_t>();  // (16-bit samples: run_wide has tables of its own)
    l.first_image = first;
    const uint8_t *src = (const uint8_t *)d_pixels + first * frame_bytes;
    l.d_planes = src;
    if (ctx->wait_before_submit)  // (felics_compress_batch: the frames are still on their way)
        HIP_TRY(ctx, hipStreamWaitEvent(wide || ctx->serial ? l.stream : l.front, ctx->wait_before_submit, 0));
    if (ctx->profiling)  // on the stream the sub-batch's first kernel runs on
        HIP_TRY(ctx, hipEventRecord(l.span_begin, wide || ctx->serial ? l.stream : l.front));
    if (planes == 3) {
        if ((rc = reserve(ctx, l.planes, (size_t)g.nplanes * npix * (wide ? 4 : 2) + STAGE_PAD)) != 0) return rc;
        hipStream_t fs = wide || ctx->serial ? l.stream : l.front;
        StageTimer t(ctx, l, ST_PLANES, fs, true);
        if (wide)
            launch_rgb16_to_planes(fs, (const uint16_t *)src, (int32_t *)l.planes.p, g.npix, g.nimages);
        else
            launch_rgb8_to_planes(fs, src, (int16_t *)l.planes.p, g.npix, g.nimages);
        l.d_planes = l.planes.p;
    }
    if (wide) return planes == 3 ? run_wide<int32_t>(ctx, l, lane_out, slot) : run_wide<uint16_t>(ctx, l, lane_out, slot);
    return planes == 3 ? run_lane<int16_t, uint16_t>(ctx, l, lane_out, slot) : run_lane<uint8_t, uint8_t>(ctx, l, lane_out, slot);
}

// What the sizes that came back say about a sub-batch packed into fixed slots.
struct SlotOutcome {
    bool lookback_failed = false;  // a tile of the single-pass pack gave up waiting for the tiles before it
    bool overflow = false;         // a stream outgrew its slot, or an RGB plane its scratch slot
    bool order_violation = false;  // the front kernel's check of its own output failed: nothing of this sub-batch is to be used
    bool tile_overflow = false;    // a tile's events did not fit its slots (tile_cap_default): nothing of this sub-batch is to be used
    bool spine_error = false;      // the spine's search lost its invariant (never seen): an internal error, reported as such
    bool redo() const { return lookback_failed || order_violation || tile_overflow; }
};

SlotOutcome read_sizes(felics_ctx *ctx, Lane &l, bool wide, uint64_t slot, uint64_t *offsets, uint64_t *lens) {
    SlotOutcome o;
    const uint32_t err = (uint32_t)l.h_sizes[l.g.nimages], flags = (uint32_t)(l.h_sizes[l.g.nimages] >> 32);
    if (!wide) {
        if ((err & 1u) != 0 || (ctx->test_lookback && l.m_fused)) o.lookback_failed = true;
        if ((err & 2u) != 0) o.overflow = true;
        if ((flags & TL_FLAG_ORDER) != 0) o.order_violation = true;
        if ((flags & TL_FLAG_OVERFLOW) != 0) o.tile_overflow = true;
        if ((flags & TL_FLAG_SPINE) != 0) o.spine_error = true;
    }
    for (size_t i = 0; i < l.g.nimages; i++) {
        lens[l.first_image + i] = l.h_sizes[i];
        offsets[l.first_image + i] = (uint64_t)(l.first_image + i) * slot;
        if (slot != 0 && l.h_sizes[i] > slot) o.overflow = true;
    }
    return o;
}

// A tile of the single-pass pack gave up waiting for the tiles before it.  With tiles taken from the workgroup index that can
// be this context's own doing (a predecessor's workgroup not started yet: XCDs dispatch their shares of a grid independently
// and the other lane's kernels share them), so the first remedy is the ticket counter -- same kernel, a tile then only waits
// for workgroups that are running.  If a ticketed pack gives up as well, something else holds the GPU for a second at a time:
// the context packs with the two-pass kernels from then on.
void note_lookback_failure(felics_ctx *ctx, const Lane &l) {
    ctx->stats.lookback_fallbacks++;
    if (!l.m_tickets) {  // (what the failed sub-batch itself ran with: two queued submissions that fail together both get here)
        if (!ctx->pack_tickets) ctx->stats.ticket_retries++;
        ctx->pack_tickets = true;
        ctx->err = "a tile gave up waiting for its predecessors: this context now hands its pack tiles out by ticket";
    } else {
        ctx->two_pass = true;
        ctx->stats.two_pass = 1;
        ctx->err = "a tile gave up waiting for its predecessors: this context now packs with the two-pass kernels (slower)";
    }
}

// k_front ranks a batch of events with one returning LDS atomic and relies on the lanes that name one address being served in
// lane order -- which this hardware does (profiles/tools/micro/lds_atomic_order.hip) and no document promises; so the kernel
// checks the order of what it wrote, and a context whose check fails once ranks with ballots from then on.
void note_scatter_order_violation(felics_ctx *ctx) {
    ctx->stats.scatter_fallbacks++;
    ctx->scatter_ballot = true;
    ctx->err = "the front kernel's order check failed: this context now ranks events with ballots";
}

// A tile's events did not fit the slots a tile gets by default: the worst case from now on (more memory, same kernels).
void note_tile_overflow(felics_ctx *ctx) {
    ctx->stats.tile_overflows++;
    ctx->cap_max = true;
    ctx->test_tile_cap = false;
    ctx->err = "a tile's events outgrew its slots: this context now sizes its tiles for the worst case";
}

int spine_failure(felics_ctx *ctx) {
    ctx->err = "internal error: the spine's halving search lost its invariant";
    return FELICS_E_HIP;
}

// Encode `n` same-shape frames resident in device memory into d_out (device), on one lane, and wait.
// If d_out is NULL the context's own output buffer is used (and grown).
int encode_device(felics_ctx *ctx, Lane &l, size_t n, const void *d_pixels, uint32_t w, uint32_t h, int color, int depth,
                  uint8_t *d_out, size_t d_out_cap, uint64_t *offsets, uint64_t *lens, uint8_t **used_out,
                  bool start_exact = false) {
    const uint32_t planes = color == FELICS_COLOR_RGB ? 3 : 1;
    const uint64_t npix = (uint64_t)w * h;
    const bool wide = depth == FELICS_DEPTH_16;
    if (npix * planes >= 0xE0000000ull) return FELICS_E_UNSUPPORTED;
    if (wide && npix > WIDE_MAX_PLANE_PIXELS) return FELICS_E_UNSUPPORTED;  // an event record keeps the sample index in 29 bits
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    for (int i = 0; i < ST_COUNT; i++) l.ev_used[i] = 0;
    const bool own_out = d_out == nullptr;

    if (npix == 0) {
        // (0,_) | (_,0): header + two zero i32 per plane (compression.rs:94-98); nothing to compute
        const size_t sz = 14 + 8 * planes;
        const size_t stride = (sz + 15) & ~(size_t)15;
        const size_t need = stride * n;
        if (own_out) {
            int rc = reserve(ctx, ctx->own, need);
            if (rc) return rc;
            d_out = (uint8_t *)ctx->own.p;
            d_out_cap = ctx->own.cap;
        }
        if (need > d_out_cap) {
            if (n) lens[0] = need;
            return FELICS_E_BUFFER_TOO_SMALL;
        }
        std::vector<uint8_t> tmp(need, 0);
        for (size_t i = 0; i < n; i++) {
            header_bytes(tmp.data() + i * stride, w, h, color, depth);
            offsets[i] = i * stride;
            lens[i] = sz;
        }
        if (need) HIP_TRY(ctx, hipMemcpy(d_out, tmp.data(), need, hipMemcpyHostToDevice));
        if (used_out) *used_out = d_out;
        return FELICS_OK;
    }

    const size_t frame_bytes = (size_t)npix * planes * (wide ? 2 : 1);
    const size_t per_pass = max_images_per_pass(npix, planes, depth);
    int rc;
    // Placement.  Preferred: every stream gets a fixed slot (stream i at i * slot), so packing needs no
    // size from the host and follows the spine slice by slice.  If a stream outgrows its slot, or the
    // caller's buffer is too small for sensible slots, the streams are placed back to back instead
    // (exact sizes first, then one pack pass).
    uint64_t slot = 0;
    if (own_out) {
        slot = ((uint64_t)frame_bytes + frame_bytes / 4 + 64 + 15) & ~15ull;
        if ((rc = reserve(ctx, ctx->own, (size_t)(slot * n))) != 0) return rc;
        d_out = (uint8_t *)ctx->own.p;
        d_out_cap = ctx->own.cap;
    } else {
        slot = (d_out_cap / n) & ~15ull;
        if (slot < 64 || slot < frame_bytes / 4) slot = 0;
    }
    if (start_exact) slot = 0;

    // (at most: ranks from ballots, worst-case tiles, tickets, two-pass, exact placement, and the run that succeeds; a loop that
    // runs out without one is reported, not passed off as a result)
    bool complete = false;
    for (int attempt = 0; attempt < 7 && !complete; attempt++) {
        size_t done = 0;
        uint64_t out_base = 0;  // exact placement: where the next pass's streams start
        SlotOutcome outcome;
        // passes of up to per_pass frames (one pass unless the batch is huge)
        while (done < n && !outcome.overflow && !outcome.redo()) {
            const size_t cnt = std::min(per_pass, n - done);
            const size_t first = done + cnt;
            if ((rc = launch_sub_batch(ctx, l, done, cnt, d_pixels, w, h, color, depth, d_out + done * slot, slot, ctx->slices_blocking)) != 0) {
                (void)sync_lane(ctx, l);
                return rc;
            }
            if ((rc = wait_event(ctx, l.sized, "stream sizes")) != 0) return rc;
            outcome = read_sizes(ctx, l, wide, slot, offsets, lens);
            if (outcome.spine_error) {
                (void)sync_lane(ctx, l);
                return spine_failure(ctx);
            }
            if (slot == 0 && !outcome.redo()) {
                // exact placement of this pass: back to back, 16-byte aligned, in image order
                uint64_t need = out_base;
                for (size_t i = done; i < first; i++) {
                    offsets[i] = need;
                    need += (lens[i] + 15) & ~15ull;
                }
                if (own_out) {
                    if (done != 0) return FELICS_E_UNSUPPORTED;  // the host entry points submit one pass at a time
                    if ((rc = reserve(ctx, ctx->own, (size_t)need)) != 0) return rc;  // waits for the device
                    d_out = (uint8_t *)ctx->own.p;
                    d_out_cap = ctx->own.cap;
                }
                if (need > d_out_cap) {
                    (void)sync_lane(ctx, l);
                    lens[0] = need;  // capacity needed so far (a lower bound if more passes would follow)
                    return FELICS_E_BUFFER_TOO_SMALL;
                }
                launch_place_streams(l.tail, (const uint64_t *)l.image_bytes.p, (uint64_t *)l.image_off.p, l.g);
                uint8_t *lane_out = d_out + offsets[l.first_image];
                if (wide)
                    rc = planes == 3 ? pack_exact<int32_t>(ctx, l, lane_out) : pack_exact<uint16_t>(ctx, l, lane_out);
                else
                    rc = planes == 3 ? pack_exact<int16_t>(ctx, l, lane_out) : pack_exact<uint8_t>(ctx, l, lane_out);
                if (rc) {
                    (void)sync_lane(ctx, l);
                    return rc;
                }
                out_base = need;
            }
            if ((rc = sync_lane(ctx, l)) != 0) return rc;
            done = first;
        }
        if (outcome.redo()) {
            if ((rc = sync_lane(ctx, l)) != 0) return rc;
            if (outcome.order_violation)
                note_scatter_order_violation(ctx);
            else if (outcome.tile_overflow)
                note_tile_overflow(ctx);
            else
                note_lookback_failure(ctx, l);
            continue;
        }
        if (!outcome.overflow) {
            complete = true;
            break;
        }
        ctx->stats.slot_overflows++;
        slot = 0;  // a stream outgrew its slot: do the batch again with exact placement
    }
    if (!complete) {
        ctx->err = "internal error: the batch was redone with every remedy and still did not complete";
        return FELICS_E_HIP;
    }
    collect_timing(ctx, l);
    if (used_out) *used_out = d_out;
    return FELICS_OK;
}

bool any_pending(const felics_ctx *ctx) {
    for (const Lane &l : ctx->lanes)
        if (l.pending) return true;
    return false;
}

}  // namespace

// --------------------------------------------------------------------------------------------
// C ABI
// --------------------------------------------------------------------------------------------

extern "C" {

int felics_ctx_create(int device, felics_ctx **out) {
    if (!out) return FELICS_E_INVALID_ARGUMENT;
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0 || device < 0 || device >= count) return FELICS_E_HIP;
    felics_ctx *ctx = new (std::nothrow) felics_ctx();
    if (!ctx) return FELICS_E_IO;
    ctx->device = device;
    // The lanes' streams want hardware queues of their own; ROCm's default is 4 per process and the caller's stream
    // takes one.
    // (A library does not touch the process environment: felics_amd/api.py, bench.py and the command lines ask for the
    // hardware queues -- GPU_MAX_HW_QUEUES -- before the runtime starts.)
    ctx->nlanes = lanes_from_env();
    ctx->poison = getenv("FELICS_POISON") != nullptr;
    ctx->two_pass = getenv("FELICS_TWO_PASS") != nullptr;
    ctx->test_lookback = getenv("FELICS_TEST_LOOKBACK_FAIL") != nullptr;
    if (const char *e = getenv("FELICS_SCATTER")) ctx->scatter_ballot = !strcmp(e, "ballot");
    ctx->test_tile_cap = getenv("FELICS_TEST_TILE_CAP") != nullptr;
    ctx->test_scatter_order = getenv("FELICS_TEST_SCATTER_ORDER") != nullptr;
    ctx->pack_tickets = ctx->own_tails = getenv("FELICS_OWN_TAILS") != nullptr;
    ctx->serial = getenv("FELICS_SERIAL") != nullptr;
    ctx->test_timeout = getenv("FELICS_TEST_TIMEOUT") != nullptr;
    if (const char *e = getenv("FELICS_SLICES")) ctx->slices_blocking = std::max(1, std::min(atoi(e), SLICES));
    if (const char *e = getenv("FELICS_SLICES_QUEUED")) ctx->slices_queued = std::max(1, std::min(atoi(e), SLICES));  // (tuning sweeps: profiles/tools/sweep_queue.sh)
    ctx->trace = getenv("FELICS_TRACE") != nullptr;
    if (const char *e = getenv("FELICS_TIMEOUT_S")) ctx->timeout_s = std::max(1, atoi(e));
    bool ok = hipSetDevice(device) == hipSuccess;
    // Oldest work first: the spine (the one sequential chain) and the tail, which finishes the submission that is
    // furthest along, go before the front (classification and event sort of the submission that has just started).  Measured
    // with two submissions in flight: 4.11 / 4.13 ms per step against 4.24 / 4.19 with the front preferred (round 1's choice)
    // and 4.12 / 4.17 with only the tail preferred; round 5, all eight combinations of high / low for spine, front and tail:
    // 2.53-2.65 ms, the differences inside the run-to-run spread (profiles/r05/experiments.txt); blocking calls do not care.
    int prio_low = 0, prio_high = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_low, &prio_high);  // numerically: low >= high
    int prio_spine = prio_high, prio_front = prio_low, prio_tail = prio_high;
    for (int li = 0; li < ctx->nlanes; li++) {
        Lane &l = ctx->lanes[li];
        ok = ok && hipStreamCreateWithPriority(&l.stream, hipStreamNonBlocking, prio_spine) == hipSuccess;
        ok = ok && hipStreamCreateWithPriority(&l.front, hipStreamNonBlocking, prio_front) == hipSuccess;
        ok = ok && hipStreamCreateWithPriority(&l.kstream, hipStreamNonBlocking, prio_tail) == hipSuccess;
        // One tail stream for all lanes: the pack kernels of two submissions run one after the other (measured faster:
        // 4.6 vs 4.8 ms per step).  FELICS_OWN_TAILS=1 gives every lane its own; that is safe since the pack kernels hand
        // out their tiles by ticket (FusedArgs::ticket), it just is not faster.
        if (&l == &ctx->lanes[0] || ctx->own_tails)
            ok = ok && hipStreamCreateWithPriority(&l.tail, hipStreamNonBlocking, prio_tail) == hipSuccess;
        else
            l.tail = ctx->lanes[0].tail;
        for (int q = 0; q < SLICES && ok; q++) {
            ok = hipEventCreateWithFlags(&l.slice_done[q], hipEventDisableTiming) == hipSuccess;
            ok = ok && hipEventCreateWithFlags(&l.spine_done[q], hipEventDisableTiming) == hipSuccess;
            ok = ok && hipEventCreateWithFlags(&l.assign_done[q], hipEventDisableTiming) == hipSuccess;
        }
        ok = ok && hipEventCreateWithFlags(&l.sized, hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventCreate(&l.span_begin) == hipSuccess && hipEventCreate(&l.span_end) == hipSuccess;
        for (int i = 0; i < ST_COUNT && ok; i++)
            for (int k = 0; k < EV_PAIRS && ok; k++)
                for (int j = 0; j < 2 && ok; j++) ok = hipEventCreate(&l.ev[i][k][j]) == hipSuccess;
    }
    if (!ok) {
        felics_ctx_destroy(ctx);
        return FELICS_E_HIP;
    }
    *out = ctx;
    return FELICS_OK;
}

void felics_ctx_destroy(felics_ctx *ctx) {
    if (!ctx) return;
    if (ctx->failed) {  // kernels may still hold the streams and the workspace: leave everything to process exit
        delete ctx;
        return;
    }
    (void)hipSetDevice(ctx->device);
    // everything queued by any lane first (the lanes share the tail stream), then the teardown
    for (Lane &l : ctx->lanes) {
        if (l.front) (void)hipStreamSynchronize(l.front);
        if (l.stream) (void)hipStreamSynchronize(l.stream);
        if (l.kstream) (void)hipStreamSynchronize(l.kstream);
    }
    for (Lane &l : ctx->lanes)
        if (l.tail) (void)hipStreamSynchronize(l.tail);
    for (Lane &l : ctx->lanes) {
        DevBuf *bufs[] = {&l.planes, &l.counts, &l.chain_prog, &l.scalars, &l.evs, &l.pix_of, &l.k_map, &l.k_sorted, &l.tile_slots, &l.desc,
                          &l.block_state, &l.group_bits, &l.tile_bits, &l.tile_bitoff, &l.plane_sums, &l.image_bytes, &l.image_off,
                          &l.partial, &l.status, &l.edge_first, &l.edge_last, &l.pscratch, &l.wrecs[0], &l.wrecs[1], &l.wtile_cnt, &l.wmeta, &l.whist, &l.wdigtot, &l.heads, &l.wlong};
        for (DevBuf *b : bufs) release(*b);
        if (l.h_sizes) (void)hipHostFree(l.h_sizes);
        for (int i = 0; i < ST_COUNT; i++)
            for (int k = 0; k < EV_PAIRS; k++)
                for (int j = 0; j < 2; j++)
                    if (l.ev[i][k][j]) (void)hipEventDestroy(l.ev[i][k][j]);
        if (l.sized) (void)hipEventDestroy(l.sized);
        if (l.span_begin) (void)hipEventDestroy(l.span_begin);
        if (l.span_end) (void)hipEventDestroy(l.span_end);
        for (int q = 0; q < SLICES; q++) {
            if (l.slice_done[q]) (void)hipEventDestroy(l.slice_done[q]);
            if (l.spine_done[q]) (void)hipEventDestroy(l.spine_done[q]);
            if (l.assign_done[q]) (void)hipEventDestroy(l.assign_done[q]);
        }
        if (l.front) (void)hipStreamDestroy(l.front);
        if (l.kstream) (void)hipStreamDestroy(l.kstream);
        if (l.stream) (void)hipStreamDestroy(l.stream);
        if (l.tail && (&l == &ctx->lanes[0] || l.tail != ctx->lanes[0].tail)) (void)hipStreamDestroy(l.tail);
    }
    release(ctx->in);
    if (ctx->copy_in) (void)hipStreamSynchronize(ctx->copy_in), (void)hipStreamDestroy(ctx->copy_in);
    if (ctx->copy_out) (void)hipStreamSynchronize(ctx->copy_out), (void)hipStreamDestroy(ctx->copy_out);
    for (hipEvent_t ev : ctx->h2d_done)
        if (ev) (void)hipEventDestroy(ev);
    release(ctx->out);
    release(ctx->own);
    release(ctx->dec_meta);
    release(ctx->dec_planes);
    release(ctx->dec_table);
    release(ctx->dec_lane_table);
    delete ctx;
}

size_t felics_max_compressed_size(uint32_t w, uint32_t h, int color, int depth) {
    const uint64_t planes = color == FELICS_COLOR_RGB ? 3 : 1;
    const uint64_t emax = depth == FELICS_DEPTH_8 ? (color ? 509u : 254u) : (color ? 131069u : 65534u);
    const uint64_t px = (uint64_t)w * h;
    const uint64_t bits = planes * 64u + planes * px * (3u + emax);
    return (size_t)(14u + (bits + 7u) / 8u);
}

int felics_compress_batch_device(felics_ctx *ctx, size_t n, const void *d_pixels, uint32_t w, uint32_t h, int color,
                                 int depth, void *d_out, size_t d_out_cap, uint64_t *offsets, uint64_t *lens) {
    if (!ctx || !offsets || !lens || !d_out || (!d_pixels && n && (uint64_t)w * h)) return FELICS_E_INVALID_ARGUMENT;
    if (ctx->failed) return FELICS_E_HIP;
    int rc = check_args(w, h, color, depth);
    if (rc) return rc;
    if (n == 0) return FELICS_OK;
    if (any_pending(ctx)) return FELICS_E_INVALID_ARGUMENT;  // felics_wait_batch first
    return encode_device(ctx, ctx->lanes[0], n, d_pixels, w, h, color, depth, (uint8_t *)d_out, d_out_cap, offsets, lens,
                         nullptr);
}

int felics_submit_batch_device(felics_ctx *ctx, size_t n, const void *d_pixels, uint32_t w, uint32_t h, int color,
                               int depth, void *d_out, size_t d_out_cap, int *ticket) {
    if (!ctx || !ticket || !d_out || n == 0 || (!d_pixels && (uint64_t)w * h)) return FELICS_E_INVALID_ARGUMENT;
    if (ctx->failed) return FELICS_E_HIP;
    int rc = check_args(w, h, color, depth);
    if (rc) return rc;
    const int L = ctx->next_lane;
    Lane &l = ctx->lanes[L];
    if (l.pending) return FELICS_E_INVALID_ARGUMENT;  // MAX_LANES submissions are in flight: wait for the oldest
    l.p_n = n;
    l.p_pixels = d_pixels;
    l.p_w = w;
    l.p_h = h;
    l.p_color = color;
    l.p_depth = depth;
    l.p_out = (uint8_t *)d_out;
    l.p_cap = d_out_cap;
    l.finished = false;
    const uint32_t planes = color == FELICS_COLOR_RGB ? 3 : 1;
    const uint64_t npix = (uint64_t)w * h;
    const size_t frame_bytes = (size_t)npix * planes * (depth == FELICS_DEPTH_16 ? 2 : 1);
    uint64_t slot = (d_out_cap / n) & ~15ull;
    if (slot < 64 || slot < frame_bytes / 4) slot = 0;
    if (npix == 0 || npix * planes >= 0xE0000000ull || n > max_images_per_pass(npix, planes, depth) || slot == 0) {
        // not the plain case (fixed slots, one pass): do it now, hand the result over at the wait
        l.r_off.assign(n, 0);
        l.r_len.assign(n, 0);
        l.r_rc = encode_device(ctx, l, n, d_pixels, w, h, color, depth, l.p_out, d_out_cap, l.r_off.data(), l.r_len.data(),
                               nullptr);
        l.finished = true;
    } else {
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        for (int i = 0; i < ST_COUNT; i++) l.ev_used[i] = 0;
        l.p_slot = slot;
        if ((rc = launch_sub_batch(ctx, l, 0, n, d_pixels, w, h, color, depth, l.p_out, slot, ctx->slices_queued, true)) != 0) {
            (void)sync_lane(ctx, l);
            return rc;
        }
    }
    l.pending = true;
    *ticket = L;
    ctx->next_lane = (L + 1) % ctx->nlanes;
    return FELICS_OK;
}

int felics_wait_batch(felics_ctx *ctx, int ticket, uint64_t *offsets, uint64_t *lens) {
    if (!ctx || ticket < 0 || ticket >= ctx->nlanes || !offsets || !lens) return FELICS_E_INVALID_ARGUMENT;
    if (ctx->failed) return FELICS_E_HIP;
    Lane &l = ctx->lanes[ticket];
    if (!l.pending) return FELICS_E_INVALID_ARGUMENT;
    if (!l.finished) {
        // the lane stays marked busy until its kernels are known to have finished: after a timeout nothing may
        // reuse or free its workspace
        const int wrc = wait_event(ctx, l.sized, "stream sizes");
        if (wrc) return wrc;
    }
    l.pending = false;
    if (l.finished) {
        for (size_t i = 0; i < l.p_n; i++) {
            offsets[i] = l.r_off[i];
            lens[i] = l.r_len[i];
        }
        return l.r_rc;
    }
    int rc;
    const SlotOutcome o = read_sizes(ctx, l, l.p_depth == FELICS_DEPTH_16, l.p_slot, offsets, lens);
    if (!o.redo() && !o.overflow && !o.spine_error) {
        collect_timing(ctx, l);
        return FELICS_OK;
    }
    // the rare cases: pack again on this lane, synchronously (two-pass kernels / exact placement)
    if ((rc = sync_lane(ctx, l)) != 0) return rc;
    if (o.spine_error) return spine_failure(ctx);
    if (o.order_violation) {
        note_scatter_order_violation(ctx);
    } else if (o.tile_overflow) {
        note_tile_overflow(ctx);
    } else if (o.lookback_failed) {
        note_lookback_failure(ctx, l);
    } else {
        ctx->stats.slot_overflows++;
    }
    return encode_device(ctx, l, l.p_n, l.p_pixels, l.p_w, l.p_h, l.p_color, l.p_depth, l.p_out, l.p_cap, offsets, lens,
                         nullptr, o.overflow && !o.redo());
}

// The reference's own call shape: images in host memory in, .felics bytes in host memory out (compression.rs:255-282, :322-371;
// cfelics.rs:24-31).  The batch goes through the submission queue in CHUNKS: the frames of chunk c + 1 are copied to the device
// on a copy stream of its own while chunk c is encoded and the streams of chunk c - 1 are copied back on a third stream, so the
// link is busy in both directions under the kernels (measured, 64 4K gray8 frames from and to page-locked memory: 13.8 ms per
// batch; with the copies on the lanes' own streams 15.6).  (The copies are hipMemcpyAsync from / to the caller's pointers: at the
// link's rate, and asynchronous, if that memory is page-locked -- hipHostMalloc, hipHostRegister, a pinned torch tensor -- and
// through the runtime's staging otherwise.)  A chunk whose streams outgrow their slots and the room the slots leave for exact
// placement is encoded once more, blocking, into a buffer that grows.
int felics_compress_batch(felics_ctx *ctx, size_t n, const void *const *pixels, uint32_t w, uint32_t h, int color,
                          int depth, uint8_t *const *outs, const size_t *caps, size_t *lens) {
    if (!ctx || (n && (!pixels || !outs || !caps || !lens))) return FELICS_E_INVALID_ARGUMENT;
    if (ctx->failed) return FELICS_E_HIP;
    int rc = check_args(w, h, color, depth);
    if (rc) return rc;
    if (n == 0) return FELICS_OK;
    if (any_pending(ctx)) return FELICS_E_INVALID_ARGUMENT;  // felics_wait_batch first
    const uint32_t planes = color == FELICS_COLOR_RGB ? 3 : 1;
    const size_t frame_bytes = (size_t)w * h * planes * (depth == FELICS_DEPTH_16 ? 2 : 1);
    for (size_t i = 0; i < n && frame_bytes; i++)
        if (!pixels[i]) return FELICS_E_INVALID_ARGUMENT;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!ctx->copy_in) {
        HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->copy_in, hipStreamNonBlocking));
        HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->copy_out, hipStreamNonBlocking));
    }
    const size_t per_pass = max_images_per_pass((uint64_t)w * h, planes, depth);
    // chunks: eight per batch (the first chunk's way in and the last one's way out are what the kernels cannot cover), none larger
    // than a pass; a stream's slot as encode_device sizes it
    const size_t chunk = std::max<size_t>(1, std::min(per_pass, (n + 7) / 8));
    const uint64_t slot = ((uint64_t)frame_bytes + frame_bytes / 4 + 64 + 15) & ~15ull;
    if ((rc = reserve(ctx, ctx->in, frame_bytes * n + 64)) != 0) return rc;
    if ((rc = reserve(ctx, ctx->out, (size_t)(slot * n) + 64)) != 0) return rc;
    struct Flying {
        int ticket;
        size_t first, cnt;
    };
    std::vector<Flying> flying;
    std::vector<uint64_t> offs(chunk), sizes(chunk);
    int result = FELICS_OK;
    auto land = [&](const Flying &f) -> int {  // wait for a chunk and start its streams on their way to the caller
        int r = felics_wait_batch(ctx, f.ticket, offs.data(), sizes.data());
        const uint8_t *from = (const uint8_t *)ctx->out.p + f.first * slot;
        hipStream_t cs = ctx->copy_out;
        if (r == FELICS_E_BUFFER_TOO_SMALL) {
            // The chunk's streams outgrew their slots AND the room the slots leave for exact placement (16-bit noise: a code can be
            // 2^17 bits): once more, blocking, into a buffer of the library's own that grows to what the streams need.
            Lane &l = ctx->lanes[f.ticket];
            uint8_t *d_own = nullptr;
            r = encode_device(ctx, l, f.cnt, (const uint8_t *)ctx->in.p + f.first * frame_bytes, w, h, color, depth, nullptr, 0, offs.data(),
                              sizes.data(), &d_own);
            from = d_own;
            cs = l.stream;  // (copied out before anything else may touch ctx->own: synchronised below)
        }
        if (r) return r;
        for (size_t i = 0; i < f.cnt; i++) {
            lens[f.first + i] = (size_t)sizes[i];
            if (sizes[i] > caps[f.first + i] || !outs[f.first + i]) {
                result = FELICS_E_BUFFER_TOO_SMALL;  // lens[] still reports every size needed
                continue;
            }
            HIP_TRY(ctx, hipMemcpyAsync(outs[f.first + i], from + offs[i], (size_t)sizes[i], hipMemcpyDeviceToHost, cs));
        }
        if (from != (const uint8_t *)ctx->out.p + f.first * slot) HIP_TRY(ctx, hipStreamSynchronize(cs));
        return FELICS_OK;
    };
    auto drain = [&](int r) {  // an error: nothing of this context may be left in flight behind the caller's back
        for (const Flying &f : flying) (void)felics_wait_batch(ctx, f.ticket, offs.data(), sizes.data());
        (void)hipStreamSynchronize(ctx->copy_in);
        (void)hipStreamSynchronize(ctx->copy_out);
        return r;
    };
    // All frames are put on their way at once, chunk by chunk with an event behind each chunk: the copy stream then runs back to
    // back at the link's rate whatever the host is waiting for (with a chunk's copies queued only when its turn came, the stream
    // stood idle while the host waited for an older chunk's kernels: 35 GB/s instead of the link's ~50).
    // All frames are put on their way at once, chunk by chunk with an event behind each chunk: the copy stream then runs back to
    // back at the link's rate whatever the host is waiting for (with a chunk's copies queued only when its turn came, the stream
    // stood idle while the host waited for an older chunk's kernels: 35 GB/s instead of the link's ~50).
    // (Eight chunks of a 64-frame batch: a chunk's kernels take ~2 ms whatever its size -- the chain of a single frame -- so with two
    // chunks in flight sixteen chunks are 16 ms of kernels, four leave the first and the last chunk's 2.5 ms of copying uncovered;
    // a short last chunk changed nothing: profiles/r05/experiments.txt.)
    std::vector<size_t> starts;
    for (size_t first = 0; first < n; first += chunk) starts.push_back(first);
    const size_t nchunks = starts.size();
    starts.push_back(n);
    while (ctx->h2d_done.size() < nchunks) {
        hipEvent_t ev = nullptr;
        HIP_TRY(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        ctx->h2d_done.push_back(ev);
    }
    for (size_t c = 0; c < nchunks; c++) {
        const size_t first = starts[c], cnt = starts[c + 1] - first;
        for (size_t i = 0; i < cnt && frame_bytes; i++) {
            const hipError_t e = hipMemcpyAsync((uint8_t *)ctx->in.p + (first + i) * frame_bytes, pixels[first + i], frame_bytes,
                                                hipMemcpyHostToDevice, ctx->copy_in);
            if (e != hipSuccess) return drain(hip_fail(ctx, e, "copying frames to the device"));
        }
        if (hipEventRecord(ctx->h2d_done[c], ctx->copy_in) != hipSuccess) return drain(hip_fail(ctx, hipGetLastError(), "hipEventRecord"));
    }
    for (size_t c = 0; c < nchunks; c++) {
        const size_t first = starts[c], cnt = starts[c + 1] - first;
        if ((int)flying.size() == ctx->nlanes) {  // every lane is busy: the oldest chunk first (its lane is the next to be used)
            rc = land(flying.front());
            flying.erase(flying.begin());
            if (rc) return drain(rc);
        }
        ctx->wait_before_submit = ctx->h2d_done[c];  // the chunk's first kernel waits for its frames (launch_sub_batch)
        int ticket = -1;
        rc = felics_submit_batch_device(ctx, cnt, (const uint8_t *)ctx->in.p + first * frame_bytes, w, h, color, depth,
                                        (uint8_t *)ctx->out.p + first * slot, (size_t)(slot * cnt), &ticket);
        ctx->wait_before_submit = nullptr;
        if (rc) return drain(rc);
        flying.push_back(Flying{ticket, first, cnt});
    }
    while (!flying.empty()) {
        rc = land(flying.front());
        flying.erase(flying.begin());
        if (rc) return drain(rc);
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->copy_out));  // the streams have landed
    return result;
}

int felics_compress(felics_ctx *ctx, const void *pixels, uint32_t w, uint32_t h, int color, int depth, uint8_t *out,
                    size_t cap, size_t *out_len) {
    if (!out_len) return FELICS_E_INVALID_ARGUMENT;
    const void *px[1] = {pixels};
    uint8_t *outs[1] = {out};
    size_t caps[1] = {cap};
    size_t lens[1] = {0};
    if (!pixels && (uint64_t)w * h != 0) return FELICS_E_INVALID_ARGUMENT;
    static const uint8_t dummy = 0;
    if (!pixels) px[0] = &dummy;
    int rc = felics_compress_batch(ctx, 1, px, w, h, color, depth, outs, caps, lens);
    *out_len = lens[0];
    return rc;
}

int felics_decompress_batch_device(felics_ctx *ctx, size_t n, const void *d_streams, const uint64_t *offsets,
                                   const uint64_t *lens, void *d_pixels, size_t d_pixels_cap, felics_header *hdr_out,
                                   int *status) {
    if (!ctx || (n && (!d_streams || !offsets || !lens || !status))) return FELICS_E_INVALID_ARGUMENT;
    if (ctx->failed) return FELICS_E_HIP;
    if (n == 0) return FELICS_OK;
    if (any_pending(ctx)) return FELICS_E_INVALID_ARGUMENT;  // felics_wait_batch first
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    Lane &l = ctx->lanes[0];
    // the shape every stream must have: header of stream 0
    uint8_t h0[FELICS_HEADER_BYTES] = {0};
    const size_t hl = (size_t)std::min<uint64_t>(lens[0], FELICS_HEADER_BYTES);
    if (hl) HIP_TRY(ctx, hipMemcpy(h0, (const uint8_t *)d_streams + offsets[0], hl, hipMemcpyDeviceToHost));
    // every stream gets a status on every path out of here (felics.h): a call that ends before the streams are decoded
    // reports its own error for all of them
    auto fail_all = [&](int code) {
        for (size_t i = 0; i < n; i++) status[i] = code;
        return code;
    };
    felics_header hdr;
    int rc = felics_read_header(h0, hl, &hdr);
    if (rc) return fail_all(rc);  // stream 0 names the shape: without it nothing is decoded
    if (hdr_out) *hdr_out = hdr;
    const uint32_t planes = hdr.color_type == FELICS_COLOR_RGB ? 3 : 1;
    const size_t bps = hdr.pixel_depth == FELICS_DEPTH_16 ? 2 : 1;
    const uint64_t npix = (uint64_t)hdr.width * hdr.height;
    if (npix > 0xFFFFFFFFull) return fail_all(FELICS_E_INVALID_DIMENSIONS);
    const uint64_t frame_bytes = npix * planes * bps;
    if (frame_bytes * n > d_pixels_cap) return fail_all(FELICS_E_BUFFER_TOO_SMALL);
    if (frame_bytes && !d_pixels) return fail_all(FELICS_E_INVALID_ARGUMENT);
    // a stream of this shape is never longer than this: a caller's length beyond it is not a stream (and not a size to allocate)
    const uint64_t max_len = felics_max_compressed_size(hdr.width, hdr.height, hdr.color_type, hdr.pixel_depth);
    if (bps == 2 && decode16_lds_bytes(hdr.width) <= DECODE_LDS_LIMIT) {
        // 16-bit streams on the device: passes of at most DEC16_PASS streams (a stream's estimator table is 8.4 MB of HBM)
        // (and of at most a quarter of the free HBM; an allocation that fails all the same halves the pass)
        constexpr size_t DEC16_PASS = 1024;
        size_t per = std::min(n, DEC16_PASS);
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && ctx->dec_table.cap < decode16_table_bytes((uint32_t)per))
            per = std::max<size_t>(1, std::min(per, (free_b / 4 + ctx->dec_table.cap) / decode16_table_bytes(1)));
        for (;;) {
            const size_t table_bytes = decode16_table_bytes((uint32_t)per);
            if (table_bytes > ctx->dec_table.cap) ctx->dec_epoch = 0;  // a fresh (zeroed) buffer: epochs start over
            if ((rc = reserve_zeroed(ctx, ctx->dec_table, table_bytes)) == 0) break;
            (void)hipGetLastError();
            if (per == 1) return fail_all(rc);
            per = (per + 1) / 2;
        }
        if ((rc = reserve(ctx, ctx->dec_meta, per * 8 * 2 + per * 4)) != 0) return fail_all(rc);
        int32_t *d_planes32 = nullptr;
        if (planes == 3) {
            if ((rc = reserve(ctx, ctx->dec_planes, (size_t)(npix * 3 * 4 * per) + 64)) != 0) return fail_all(rc);
            d_planes32 = (int32_t *)ctx->dec_planes.p;
        }
        hipStream_t s = l.stream;
        for (size_t i = 0; i < n; i++) status[i] = FELICS_E_HIP;  // until the kernel's own word arrives
        int first_rc = FELICS_OK;
        for (size_t first = 0; first < n; first += per) {
            const size_t cnt = std::min(per, n - first);
            if (ctx->dec_epoch > 0xFFFFFFF0u) {  // epochs used up: clear the tables, start over
                HIP_TRY(ctx, hipMemsetAsync(ctx->dec_table.p, 0, ctx->dec_table.cap, s));
                ctx->dec_epoch = 0;
            }
            const uint32_t epoch0 = ctx->dec_epoch + 1;
            ctx->dec_epoch += 3;
            uint64_t *d_off = (uint64_t *)ctx->dec_meta.p, *d_len = d_off + cnt;
            int *d_status = (int *)(d_len + cnt);
            HIP_TRY(ctx, hipMemcpyAsync(d_off, offsets + first, cnt * 8, hipMemcpyHostToDevice, s));
            HIP_TRY(ctx, hipMemcpyAsync(d_len, lens + first, cnt * 8, hipMemcpyHostToDevice, s));
            HIP_TRY(ctx, hipMemsetAsync(d_status, 0xFF, cnt * 4, s));
            HIP_TRY(ctx, launch_decode16(s, (const uint8_t *)d_streams, d_off, d_len, (uint32_t)cnt, hdr.width, hdr.height, hdr.color_type,
                                         (uint16_t *)d_pixels + first * (frame_bytes / 2), d_planes32, (uint32_t *)ctx->dec_table.p, epoch0,
                                         d_status));
            HIP_TRY(ctx, hipMemcpyAsync(status + first, d_status, cnt * 4, hipMemcpyDeviceToHost, s));
            HIP_TRY(ctx, hipStreamSynchronize(s));
            for (size_t i = first; i < first + cnt && !first_rc; i++)
                if (status[i]) first_rc = status[i];
        }
        return first_rc;
    }
    if (bps == 2 || decode8_lds_bytes(hdr.width, hdr.color_type) > DECODE_LDS_LIMIT) {
        // host decoder, stream by stream
        std::vector<uint8_t> sbuf, pbuf;
        try {
            pbuf.resize((size_t)frame_bytes);
            sbuf.reserve((size_t)std::min<uint64_t>(max_len, 1ull << 32));
        } catch (const std::bad_alloc &) {
            return fail_all(FELICS_E_IO);
        }
        int first_rc = FELICS_OK;
        for (size_t i = 0; i < n; i++) {
            if (lens[i] > max_len) {  // (checked before anything is sized by it)
                status[i] = FELICS_E_INVALID_VALUE;
                if (!first_rc) first_rc = FELICS_E_INVALID_VALUE;
                continue;
            }
            try {
                sbuf.resize((size_t)lens[i]);
            } catch (const std::bad_alloc &) {
                for (size_t k = i; k < n; k++) status[k] = FELICS_E_IO;
                return first_rc ? first_rc : FELICS_E_IO;
            }
            if (lens[i] && hipMemcpy(sbuf.data(), (const uint8_t *)d_streams + offsets[i], (size_t)lens[i], hipMemcpyDeviceToHost) != hipSuccess) {
                for (size_t k = i; k < n; k++) status[k] = FELICS_E_HIP;
                return hip_fail(ctx, hipGetLastError(), "copying a stream to the host decoder");
            }
            felics_header hi;
            int r = felics_read_header(sbuf.data(), sbuf.size(), &hi);
            if (!r && (hi.width != hdr.width || hi.height != hdr.height || hi.color_type != hdr.color_type || hi.pixel_depth != hdr.pixel_depth))
                r = FELICS_E_INVALID_DIMENSIONS;
            if (!r) r = felics_decompress(sbuf.data(), sbuf.size(), pbuf.data(), pbuf.size(), nullptr);
            if (!r && frame_bytes && hipMemcpy((uint8_t *)d_pixels + i * frame_bytes, pbuf.data(), (size_t)frame_bytes, hipMemcpyHostToDevice) != hipSuccess) {
                for (size_t k = i; k < n; k++) status[k] = FELICS_E_HIP;
                return hip_fail(ctx, hipGetLastError(), "copying decoded pixels to the device");
            }
            status[i] = r;
            if (r && !first_rc) first_rc = r;
        }
        return first_rc;
    }
    // offsets | lens | status on the device
    const size_t meta = n * 8 * 2 + n * 4;
    if ((rc = reserve(ctx, ctx->dec_meta, meta)) != 0) return fail_all(rc);
    uint64_t *d_off = (uint64_t *)ctx->dec_meta.p, *d_len = d_off + n;
    int *d_status = (int *)(d_len + n);
    int16_t *d_planes = nullptr;
    if (planes == 3) {
        if ((rc = reserve(ctx, ctx->dec_planes, (size_t)(npix * 3 * 2 * n) + 64)) != 0) return fail_all(rc);
        d_planes = (int16_t *)ctx->dec_planes.p;
    }
    hipStream_t s = l.stream;
    for (size_t i = 0; i < n; i++) status[i] = FELICS_E_HIP;  // until the kernel's own word arrives
    HIP_TRY(ctx, hipMemcpyAsync(d_off, offsets, n * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(d_len, lens, n * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemsetAsync(d_status, 0xFF, n * 4, s));
    // hundreds of streams and more: 64 streams per wave (lane = stream); fewer: one wave per stream
    // (FELICS_TEST_DECODE_LANES=1 / =0 force one form whatever the batch: tests)
    bool by_lane = hdr.width >= 8 && n >= (planes == 3 ? DECODE8_LANES_MIN_STREAMS_RGB : DECODE8_LANES_MIN_STREAMS);
    if (const char *e = getenv("FELICS_TEST_DECODE_LANES")) by_lane = hdr.width >= 8 && atoi(e) != 0;
    if (by_lane) {
        const size_t tb = decode8_lanes_table_bytes((uint32_t)n, hdr.color_type);
        if ((rc = reserve(ctx, ctx->dec_lane_table, tb)) != 0) return fail_all(rc);
        HIP_TRY(ctx, hipMemsetAsync(ctx->dec_lane_table.p, 0, tb, s));
        HIP_TRY(ctx, launch_decode8_lanes(s, (const uint8_t *)d_streams, d_off, d_len, (uint32_t)n, hdr.width, hdr.height, hdr.color_type,
                                          (uint8_t *)d_pixels, d_planes, (uint32_t *)ctx->dec_lane_table.p, d_status));
    } else {
        HIP_TRY(ctx, launch_decode8(s, (const uint8_t *)d_streams, d_off, d_len, (uint32_t)n, hdr.width, hdr.height, hdr.color_type,
                                    (uint8_t *)d_pixels, d_planes, d_status));
    }
    HIP_TRY(ctx, hipMemcpyAsync(status, d_status, n * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    for (size_t i = 0; i < n; i++)
        if (status[i]) return status[i];
    return FELICS_OK;
}

int felics_write_header(const felics_header *hdr, uint8_t *out, size_t cap) {
    if (!hdr || !out) return FELICS_E_INVALID_ARGUMENT;
    if (cap < FELICS_HEADER_BYTES) return FELICS_E_BUFFER_TOO_SMALL;
    if (hdr->color_type > 1) return FELICS_E_INVALID_COLOR_TYPE;
    if (hdr->pixel_depth > 1) return FELICS_E_INVALID_PIXEL_DEPTH;
    header_bytes(out, hdr->width, hdr->height, hdr->color_type, hdr->pixel_depth);
    return FELICS_OK;
}

const char *felics_strerror(int code) {
    switch (code) {
        case FELICS_OK: return "ok";
        case FELICS_E_IO: return "I/O error (truncated stream or allocation failure)";
        case FELICS_E_INVALID_VALUE: return "a decoded value does not fit the image bit depth";
        case FELICS_E_VALUE_OVERFLOW: return "arithmetic overflow while decoding";
        case FELICS_E_INVALID_DIMENSIONS: return "invalid channel dimensions";
        case FELICS_E_INVALID_COLOR_TYPE: return "invalid color type";
        case FELICS_E_INVALID_PIXEL_DEPTH: return "invalid pixel depth";
        case FELICS_E_INVALID_SIGNATURE: return "not a felics file (bad signature)";
        case FELICS_E_BUFFER_TOO_SMALL: return "output buffer too small";
        case FELICS_E_HIP: return "no usable HIP device or HIP runtime error";
        case FELICS_E_UNSUPPORTED: return "not supported by the GPU encoder in this build";
        case FELICS_E_INVALID_ARGUMENT: return "invalid argument";
        default: return "unknown error";
    }
}

const char *felics_last_error(const felics_ctx *ctx) { return ctx ? ctx->err.c_str() : ""; }

int felics_set_profiling(felics_ctx *ctx, int enabled) {
    if (!ctx) return FELICS_E_INVALID_ARGUMENT;
    ctx->profiling = enabled != 0;
    return FELICS_OK;
}

int felics_get_stats(const felics_ctx *ctx, felics_stats *out) {
    if (!ctx || !out) return FELICS_E_INVALID_ARGUMENT;
    *out = ctx->stats;
    out->two_pass = ctx->two_pass ? 1 : 0;
    out->failed = ctx->failed ? 1 : 0;
    return FELICS_OK;
}

int felics_stage_count(void) { return ST_COUNT; }

int felics_lane_count(void) { return lanes_from_env(); }

int felics_ctx_lane_count(const felics_ctx *ctx) { return ctx ? ctx->nlanes : FELICS_E_INVALID_ARGUMENT; }

int felics_get_stage_launches(const felics_ctx *ctx, int *launches, int cap) {
    if (!ctx || !launches) return FELICS_E_INVALID_ARGUMENT;
    int n = cap < ST_COUNT ? cap : (int)ST_COUNT;
    for (int i = 0; i < n; i++) launches[i] = ctx->stage_launches[i];
    return n;
}

const char *felics_stage_name(int stage) { return stage >= 0 && stage < ST_COUNT ? kStageNames[stage] : ""; }

int felics_get_span_ms(const felics_ctx *ctx, float *ms) {
    if (!ctx || !ms) return FELICS_E_INVALID_ARGUMENT;
    *ms = ctx->span_ms;
    return FELICS_OK;
}

int felics_get_stage_ms(const felics_ctx *ctx, float *ms, int cap) {
    if (!ctx || !ms) return FELICS_E_INVALID_ARGUMENT;
    int n = cap < ST_COUNT ? cap : (int)ST_COUNT;
    for (int i = 0; i < n; i++) ms[i] = ctx->stage_ms[i];
    return n;
}

}  // extern "C"
