// felics_stripe.hip -- the encode path of 8-bit frames as ONE persistent kernel (gfx950).
//
// The multi-kernel pipeline of felics_kernels.hip keeps the events of a whole batch in HBM between its stages
// (partitioned events, their pixels, k per pixel: ~19 bytes of HBM traffic per pixel).  Here a workgroup takes a
// TILE of consecutive pixels of one plane through every stage without leaving the CU's LDS, so HBM sees the
// pixels once on the way in and the packed bits on the way out:
//
//   A  load the tile (+ the row above) into LDS; classify every pixel against its two neighbours
//      (misc.rs:6-24, compression.rs:124-145); stable partition of the out-of-range EVENTS by context in LDS
//   B  the one sequential dependency of the codec: the Rice-parameter estimator (parameter_selection.rs:49-85)
//      runs along every context's events in raster order, so tile t needs the estimator's state after tile
//      t - 1.  The table (512 contexts x 6 counters) of a plane lives in global memory; a tile waits for its
//      predecessor's token, replays its own events on top of the rows it needs, writes them back and passes
//      the token on.  Inside the tile a context's events are cut into blocks of 16: one wave finds the state
//      at every block start (lane = block: prefix sums of the blocks' length sums, halvings located by
//      ballot + an in-block search), then every lane replays one block serially and leaves k per event.
//   C  codes (rice_coding.rs:26-38, phase_in_coding.rs:59-84, compression.rs:29-45) strung together per
//      thread, tile offsets by decoupled look-back, bits shifted into an LDS window and streamed out
//      MSB-first -- the single-pass pack of felics_kernels.hip with the classification kept in registers.
//
// Tiles are handed out by a ticket counter in (tile, plane) order, so every tile a workgroup waits for (the
// tile before it in its plane) was taken earlier by a workgroup that is running: no assumption about
// dispatch order or residency.  Cross-workgroup data (table rows, token, look-back status words) moves with
// agent-scope relaxed atomics only (sc1 accesses), rows drained (s_waitcnt vmcnt(0)) before the token is
// stored -- the hand-off form of cdna_hip_programming.md, Guideline 16 R1.
//
// Integer work only: no MFMA.  Wave = 64 lanes.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "felics_codes.h"
#include "felics_device.h"
#include "felics_kernels.h"

namespace felics {

namespace {

constexpr uint32_t SW = STRIPE_THREADS / 64;  // waves per workgroup
constexpr uint32_t EVB = 16;                  // events per block
constexpr uint32_t RING = 512;                // ring entries per wave: the events of 32 lanes x 16 pixels
constexpr uint32_t S_LOCAL_WORDS = 8;         // private bit string of a thread: 256 bits (more: slow path)
constexpr uint32_t S_WIN_WORDS = 4096;        // LDS bit window: 16 KiB
constexpr uint32_t REC_NONE = 3u;             // rec = cls | ctx << 2 | val << 11; cls 3 = no code (pixels 0, 1; past the end)

// LDS carve-up (bytes from the dynamic base).  The region `u` is used three times over: per-wave context
// counters + rings while the events are partitioned, block tables + block sums + block states while the
// estimator runs, private bit strings + bit window while the codes are packed.
struct Layout {
    uint32_t lead;  // samples kept in front of the tile in `span` (>= W, multiple of 16)
    uint32_t span, se, sp, kq, u, tabs, misc, total;
};

template <typename T>
__host__ __device__ inline Layout layout_for(uint32_t W) {
    using C = StripeCfg<T>;
    Layout L;
    L.lead = (W + 15u) & ~15u;
    if (L.lead < 16u) L.lead = 16u;
    uint32_t at = 0;
    L.span = at;
    at += (L.lead + C::TILE + 16u) * (uint32_t)sizeof(T);
    at = (at + 15u) & ~15u;
    L.se = at;
    at += (C::TILE + 2048u) * (uint32_t)sizeof(typename C::ET);
    L.sp = at;
    at += (C::TILE + 2048u) * 2u;
    L.kq = at;
    at += C::TILE;
    L.u = at;
    at += 48u * 1024u;
    L.tabs = at;
    at += 3u * NCTX * 2u;
    L.misc = at;
    at += 256u;
    L.total = at;
    return L;
}

template <typename T>
constexpr uint32_t max_blocks() {
    return StripeCfg<T>::TILE / EVB + NCTX;
}

// misc words
enum : uint32_t { M_TICKET = 0, M_NB = 1, M_TILE_LO = 2 /* u64: 2, 3 */, M_WSUM = 8 /* 16 */, M_WSUM2 = 24 /* 16 */, M_POLL_FAIL = 40 };

__device__ __forceinline__ uint64_t ld_agent(const uint64_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(uint64_t *p, uint64_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// inclusive prefix sums inside each row of 16 lanes (the first four steps of wave_incl_scan)
__device__ __forceinline__ uint32_t row_incl_scan(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);
    return v;
}

__device__ __forceinline__ uint32_t min6(const uint32_t (&v)[6]) {
    return min(min(min(v[0], v[1]), min(v[2], v[3])), min(v[4], v[5]));
}

// block-start states in LDS: six counters of 21 bits (a counter stays below 2^17: the smallest grows by at
// least one per event and is halved above 1024, the others grow at most 32 times as fast)
__device__ __forceinline__ void pack_state(const uint32_t (&S)[6], uint64_t &a, uint64_t &b) {
    a = (uint64_t)S[0] | ((uint64_t)S[1] << 21) | ((uint64_t)S[2] << 42);
    b = (uint64_t)S[3] | ((uint64_t)S[4] << 21) | ((uint64_t)S[5] << 42);
}
__device__ __forceinline__ void unpack_state(uint64_t a, uint64_t b, uint32_t (&S)[6]) {
    S[0] = (uint32_t)a & 0x1FFFFFu;
    S[1] = (uint32_t)(a >> 21) & 0x1FFFFFu;
    S[2] = (uint32_t)(a >> 42) & 0x1FFFFFu;
    S[3] = (uint32_t)b & 0x1FFFFFu;
    S[4] = (uint32_t)(b >> 21) & 0x1FFFFFu;
    S[5] = (uint32_t)(b >> 42) & 0x1FFFFFu;
}

__device__ __forceinline__ uint32_t event_at(const uint32_t *w, uint32_t t, uint8_t) { return (w[t >> 2] >> (8u * (t & 3u))) & 0xFFu; }
__device__ __forceinline__ uint32_t event_at(const uint32_t *w, uint32_t t, uint16_t) { return (w[t >> 1] >> (16u * (t & 1u))) & 0xFFFFu; }

// One block of up to 16 events of one context replayed serially by ONE lane (every lane of the wave has a block
// of its own): k of every event (get_k precedes update, compression.rs:127,139; ties go to the largest k,
// parameter_selection.rs:79) stored at the event's pixel, the six counters advanced and halved
// (parameter_selection.rs:49-63).  S is the state before the block's first event on entry, after its last on exit.
template <typename ET>
__device__ __forceinline__ void replay_block(uint32_t (&S)[6], const ET *se, const uint16_t *sp, uint8_t *kq, uint32_t ev0,
                                             uint32_t n) {
    constexpr uint32_t EW = EVB * sizeof(ET) / 4;
    uint32_t ew[EW], pw[EVB / 2];
    if (n != 0) {
        const uint32_t *es = reinterpret_cast<const uint32_t *>(se + ev0);  // ev0 is a multiple of 4 events
        const uint32_t *ps = reinterpret_cast<const uint32_t *>(sp + ev0);
#pragma unroll
        for (uint32_t q = 0; q < EW; q++) ew[q] = es[q];
#pragma unroll
        for (uint32_t q = 0; q < EVB / 2; q++) pw[q] = ps[q];
    }
#pragma unroll
    for (uint32_t t = 0; t < EVB; t++) {
        if (t < n) {
            const uint32_t e = event_at(ew, t, ET());
            const uint32_t pix = (pw[t >> 1] >> (16u * (t & 1u))) & 0xFFFFu;
            const uint32_t key = min(min(min((S[0] << 3) | 7u, (S[1] << 3) | 6u), min((S[2] << 3) | 5u, (S[3] << 3) | 4u)),
                                     min((S[4] << 3) | 3u, (S[5] << 3) | 2u));
            kq[pix] = (uint8_t)(7u - (key & 7u));
            S[0] += e + 1u;
            S[1] += (e >> 1) + 2u;
            S[2] += (e >> 2) + 3u;
            S[3] += (e >> 3) + 4u;
            S[4] += (e >> 4) + 5u;
            S[5] += (e >> 5) + 6u;
            const uint32_t h = min6(S) > 1024u ? 1u : 0u;
#pragma unroll
            for (uint32_t k = 0; k < 6; k++) S[k] >>= h;
        }
    }
}

// The state at the start of every block of one context with two or more blocks, by the whole wave.
// Lane j holds the six length sums of block j; with the state S before the first block, the state at the end of
// block j is S + (inclusive prefix of the sums) as long as no halving happens.  `min > 1024` is monotone along the
// events, so the first block whose end state has all six counters above 1024 contains the next halving: it is
// searched event by event (16 lanes, prefix sums of the events' six lengths), the state is halved there
// (several times if need be), and the blocks behind it continue from the corrected base.
template <typename ET>
__device__ __forceinline__ void walk_context(uint32_t (&S)[6], const ET *se, uint32_t seg0, uint32_t len, uint32_t b0,
                                             const uint32_t *bsum, uint64_t *bstate) {
    const uint32_t lane = lane_id();
    const uint32_t nb = (len + EVB - 1) / EVB;
    for (uint32_t pass = 0; pass < nb; pass += 64) {
        const uint32_t j = pass + lane;
        const bool have = j < nb;
        uint32_t B[6] = {0u, 0u, 0u, 0u, 0u, 0u};
        if (have) {
            const uint32_t b01 = bsum[(b0 + j) * 3], b23 = bsum[(b0 + j) * 3 + 1], b45 = bsum[(b0 + j) * 3 + 2];
            B[0] = b01 & 0xFFFFu; B[1] = b01 >> 16; B[2] = b23 & 0xFFFFu; B[3] = b23 >> 16; B[4] = b45 & 0xFFFFu; B[5] = b45 >> 16;
        }
        uint32_t I[6], base[6];
#pragma unroll
        for (uint32_t k = 0; k < 6; k++) {
            I[k] = wave_incl_scan(B[k]);
            base[k] = S[k];
        }
        uint32_t jlo = 0;
        while (true) {
            uint32_t X[6];  // state at the start of this lane's block, if no halving lies between jlo and it
#pragma unroll
            for (uint32_t k = 0; k < 6; k++) X[k] = base[k] + I[k] - B[k];
            uint32_t E[6];
#pragma unroll
            for (uint32_t k = 0; k < 6; k++) E[k] = X[k] + B[k];
            const uint64_t over = __ballot(have && lane >= jlo && min6(E) > 1024u);
            const uint32_t js = over ? (uint32_t)__builtin_ctzll(over) : 64u;  // first block with a halving inside
            if (have && lane >= jlo && lane <= js) {
                uint64_t a, b;
                pack_state(X, a, b);
                bstate[(b0 + j) * 2] = a;
                bstate[(b0 + j) * 2 + 1] = b;
            }
            if (!over) break;
            uint32_t Sb[6];
#pragma unroll
            for (uint32_t k = 0; k < 6; k++) Sb[k] = readlane(X[k], js);
            // the events of block js, one per lane (16 lanes)
            const uint32_t bj = pass + js;
            const uint32_t nvalid = min(EVB, len - bj * EVB);
            uint32_t l01 = 0, l23 = 0, l45 = 0;
            if (lane < nvalid) packed_lengths((uint32_t)se[seg0 + bj * EVB + lane], l01, l23, l45);
            const uint32_t p01 = row_incl_scan(l01), p23 = row_incl_scan(l23), p45 = row_incl_scan(l45);
            const uint32_t P[6] = {p01 & 0xFFFFu, p01 >> 16, p23 & 0xFFFFu, p23 >> 16, p45 & 0xFFFFu, p45 >> 16};
            uint32_t lo = 0;
            while (true) {
                uint32_t V[6];
#pragma unroll
                for (uint32_t k = 0; k < 6; k++) V[k] = Sb[k] + P[k];
                const uint64_t hm = __ballot(lane >= lo && lane < nvalid && min6(V) > 1024u);
                if (!hm) break;
                const uint32_t f = (uint32_t)__builtin_ctzll(hm);
                // S <- ((S + P(f)) >> 1) - P(f): the lanes behind f add their own P(t) >= P(f) back (mod 2^32)
#pragma unroll
                for (uint32_t k = 0; k < 6; k++) {
                    const uint32_t pf = readlane(P[k], f);
                    Sb[k] = ((Sb[k] + pf) >> 1) - pf;
                }
                lo = f + 1;
            }
            // state at the end of block js = Sb + (its sums); the blocks behind it continue from a base that
            // makes base + I(j) their end state again
#pragma unroll
            for (uint32_t k = 0; k < 6; k++) base[k] = Sb[k] + readlane(B[k], js) - readlane(I[k], js);
            jlo = js + 1;
            if (jlo >= 64) break;
        }
#pragma unroll
        for (uint32_t k = 0; k < 6; k++) S[k] = base[k] + readlane(I[k], 63);
    }
}

// thread-private bit string, MSB-first; word w of thread t at buf[w * STRIPE_THREADS + t]
struct StripeBits {
    uint32_t *buf;
    uint64_t acc;
    uint32_t fill, word, total;

    __device__ __forceinline__ void begin(uint32_t *b) {
        buf = b;
        acc = 0;
        fill = word = total = 0;
    }
    __device__ __forceinline__ void put(uint32_t v, uint32_t n) {
        acc |= (uint64_t)v << (64u - fill - n);
        fill += n;
        total += n;
        if (fill >= 32) {
            if (word < S_LOCAL_WORDS) buf[word * STRIPE_THREADS] = (uint32_t)(acc >> 32);
            acc <<= 32;
            fill -= 32;
            word++;
        }
    }
    __device__ __forceinline__ void put_ones(uint32_t q) {
        while (q >= 32 && word < S_LOCAL_WORDS) {
            put(0xFFFFFFFFu, 32);
            q -= 32;
        }
        if (q >= 32) {  // past the buffer: only the count matters
            total += q & ~31u;
            word += q >> 5;
            q &= 31u;
        }
        if (q) put((1u << q) - 1u, q);
    }
    __device__ __forceinline__ void finish() {
        if (fill && word < S_LOCAL_WORDS) buf[word * STRIPE_THREADS] = (uint32_t)(acc >> 32);
    }
};

// The pack stage classifies its pixels again from `span` (keeping sixteen classifications in registers across
// the estimator stage costs more than recomputing them).  Calls raw(i, value) for pixels 0 and 1 of the plane
// and f(pc, k) for every other pixel of the thread's group [first, first + PPT), in raster order; four pixels
// per trip, the register arrays shifted down after each trip so that the loop stays rolled.
template <typename T, uint32_t PPT, typename FR, typename F>
__device__ __forceinline__ void stripe_walk(const uint8_t *smem_span, uint32_t lead, uint32_t off, const T *__restrict__ pl,
                                            uint32_t first, uint32_t end, uint32_t W, const uint8_t *kq, FR &&raw, F &&f) {
    constexpr uint32_t NW = PPT * sizeof(T) / 4;  // 4
    constexpr uint32_t D = sizeof(T);             // dwords per four pixels
    const T *span = reinterpret_cast<const T *>(smem_span);
    uint32_t cw[NW], uw[NW + 1], kw[PPT / 4];
    {
        const uint4 c4 = *reinterpret_cast<const uint4 *>(span + lead + off);
        cw[0] = c4.x; cw[1] = c4.y; cw[2] = c4.z; cw[3] = c4.w;
        const uint32_t ub = (lead + off - W) * (uint32_t)sizeof(T);  // the row above: any byte alignment
        const uint32_t *ua = reinterpret_cast<const uint32_t *>(smem_span + (ub & ~3u));
        uint32_t d[NW + 2];
#pragma unroll
        for (uint32_t q = 0; q < NW + 2; q++) d[q] = ua[q];
#pragma unroll
        for (uint32_t q = 0; q < NW + 1; q++) uw[q] = __builtin_amdgcn_alignbyte(d[q + 1], d[q], ub & 3u);
        const uint32_t *kp = reinterpret_cast<const uint32_t *>(kq + off);
#pragma unroll
        for (uint32_t q = 0; q < PPT / 4; q++) kw[q] = kp[q];
    }
    int left = (int)span[lead + off - 1], left2 = (int)span[lead + off - 2];
    for (uint32_t i = first; i < min(end, 2u); i++) raw(i, (uint32_t)(int)span[lead + off + (i - first)]);
    Coord xy;
    xy.set(first, W);
#pragma nounroll
    for (uint32_t g = 0; g < PPT; g += 4) {
#pragma unroll
        for (uint32_t j = 0; j < 4; j++) {
            const uint32_t i = first + g + j;
            const int p = sample_at(cw, j, T());
            if (i < end && i >= 2) {
                const uint32_t k = (kw[0] >> (8u * j)) & 7u;
                const int above = sample_at(uw, j, T());
                int v1 = left, v2 = above;  // interior: left and above (misc.rs:6-24)
                if (xy.y == 0) {
                    v2 = left2;  // first row: the two pixels to the left
                } else if (xy.x == 0) {
                    v1 = above;  // first column: above and two rows up, or above-right for pixel (0,1)
                    v2 = xy.y >= 2 ? (int)pl[i - 2 * W] : sample_at(uw, j + 1, T());
                }
                const int Hh = max(v1, v2), Ll = min(v1, v2);
                PixelClass pc;
                pc.ctx = (uint32_t)(Hh - Ll);
                pc.cls = p < Ll ? CLS_BELOW : (p > Hh ? CLS_ABOVE : CLS_IN);
                pc.val = p < Ll ? (uint32_t)(Ll - p - 1) : (p > Hh ? (uint32_t)(p - Hh - 1) : (uint32_t)(p - Ll));
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
        for (uint32_t q = 0; q + D < NW; q++) cw[q] = cw[q + D];
#pragma unroll
        for (uint32_t q = 0; q + D < NW + 1; q++) uw[q] = uw[q + D];
#pragma unroll
        for (uint32_t q = 0; q + 1 < PPT / 4; q++) kw[q] = kw[q + 1];
    }
}

}  // namespace

template <typename T>
__global__ __launch_bounds__(STRIPE_THREADS) void k_stripe(StripeArgs a) {
    using C = StripeCfg<T>;
    using ET = typename C::ET;
    constexpr uint32_t TILE = C::TILE, PPT = C::PPT;
    constexpr uint32_t NW = PPT * sizeof(T) / 4;  // dwords holding a thread's pixels (4)
    static_assert(NW == 4, "a thread's pixels are one 16-byte LDS read");
    static_assert(TILE <= (1u << 14), "ring records keep the pixel's offset in its tile in 14 bits");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const Layout L = layout_for<T>(a.W);
    T *span = reinterpret_cast<T *>(smem + L.span);
    ET *se = reinterpret_cast<ET *>(smem + L.se);
    uint16_t *sp = reinterpret_cast<uint16_t *>(smem + L.sp);
    uint8_t *kq = smem + L.kq;
    // region u, three lives
    uint16_t *cnt16 = reinterpret_cast<uint16_t *>(smem + L.u);                      // [SW][NCTX]
    uint32_t *rings = reinterpret_cast<uint32_t *>(smem + L.u + SW * NCTX * 2);      // [SW][RING]
    constexpr uint32_t MAXB = max_blocks<T>();
    uint32_t *bsum = reinterpret_cast<uint32_t *>(smem + L.u);                       // [MAXB][3]
    uint64_t *bstate = reinterpret_cast<uint64_t *>(smem + L.u + MAXB * 12);         // [MAXB][2]
    uint16_t *blk_ev0 = reinterpret_cast<uint16_t *>(smem + L.u + MAXB * 28);        // [MAXB]
    uint8_t *blk_n = smem + L.u + MAXB * 30;                                         // [MAXB]: events | 0x80 = only block of its context
    static_assert(MAXB * 31 <= 48 * 1024, "block tables fit region u");
    uint32_t *lbuf = reinterpret_cast<uint32_t *>(smem + L.u);                       // [S_LOCAL_WORDS][STRIPE_THREADS]
    uint32_t *win = reinterpret_cast<uint32_t *>(smem + L.u + S_LOCAL_WORDS * STRIPE_THREADS * 4);  // [S_WIN_WORDS]
    static_assert(S_LOCAL_WORDS * STRIPE_THREADS * 4 + S_WIN_WORDS * 4 <= 48 * 1024, "bit strings fit region u");
    uint16_t *seg_start = reinterpret_cast<uint16_t *>(smem + L.tabs);
    uint16_t *seg_len = seg_start + NCTX;
    uint16_t *first_blk = seg_len + NCTX;
    uint32_t *misc = reinterpret_cast<uint32_t *>(smem + L.misc);

    const uint32_t tid = threadIdx.x, lane = lane_id();
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const uint32_t total_tickets = a.ntiles * a.nplanes;
    const uint32_t W = a.W, npix = a.npix;
    uint32_t *done = a.ctl + STRIPE_CTL_DONE;

    // FELICS_STRIPE_STAMPS (debugging aid): thread 0 leaves the 100 MHz wall clock at the phase boundaries of every tile
    auto stamp = [&](uint32_t ticket_, uint32_t slot) {
        if (a.stamps && tid == 0) a.stamps[(uint64_t)ticket_ * STRIPE_STAMPS + slot] = wall_clock64();
    };
    auto wstamp = [&](uint32_t ticket_, uint32_t slot) {  // per wave: 16 + wave * 4 + slot
        if (a.stamps && lane == 0) a.stamps[(uint64_t)ticket_ * STRIPE_STAMPS + 16 + wave * 4 + slot] = wall_clock64();
    };
    while (true) {
        if (tid == 0) misc[M_TICKET] = atomicAdd(&a.ctl[STRIPE_CTL_TICKET], 1u);
        __syncthreads();
        const uint32_t ticket = misc[M_TICKET];
        if (ticket >= total_tickets) break;
        stamp(ticket, 0);
        const uint32_t tile = ticket / a.nplanes, plane = ticket - tile * a.nplanes;
        const T *pl = reinterpret_cast<const T *>(a.planes) + (uint64_t)plane * npix;
        const uint32_t tile_first = tile * TILE;
        const uint32_t end = min(tile_first + TILE, npix);
        const uint32_t off = tid * PPT, first = tile_first + off;

        // ---------------- A1: the tile and the samples in front of it -> span; clear the counters
        {
            constexpr uint32_t EPC = 16 / sizeof(T);
            const int64_t g_first = (int64_t)tile_first - (int64_t)L.lead;
            const uint32_t count = L.lead + TILE + 16u;
            const bool aligned = ((reinterpret_cast<uintptr_t>(pl) + (uint64_t)g_first * sizeof(T)) & 15u) == 0;
            for (uint32_t c = tid; c < count / EPC; c += STRIPE_THREADS) {
                const int64_t g0 = g_first + (int64_t)c * EPC;
                if (g0 >= (int64_t)end) continue;  // nothing behind the tile is needed
                if (aligned && g0 >= 0 && g0 + EPC <= (int64_t)npix) {
                    reinterpret_cast<uint4 *>(span)[c] = *reinterpret_cast<const uint4 *>(pl + g0);
                } else {
#pragma unroll
                    for (uint32_t e = 0; e < EPC; e++) {
                        const int64_t gi = g0 + e;
                        if (gi >= 0 && gi < (int64_t)npix) span[c * EPC + e] = pl[gi];
                    }
                }
            }
            uint32_t *cz = reinterpret_cast<uint32_t *>(cnt16);
            for (uint32_t i = tid; i < SW * NCTX / 2; i += STRIPE_THREADS) cz[i] = 0;
        }
        __syncthreads();
        stamp(ticket, 6);

        // ---------------- A2: classify this thread's pixels; count the events per (wave, context)
        uint32_t rec[PPT];
        uint32_t nev = 0;
#pragma unroll
        for (uint32_t j = 0; j < PPT; j++) rec[j] = REC_NONE;
        if (first < end) {
            uint32_t cw[NW], uw[NW + 1];
            {
                const uint4 c4 = *reinterpret_cast<const uint4 *>(span + L.lead + off);
                cw[0] = c4.x; cw[1] = c4.y; cw[2] = c4.z; cw[3] = c4.w;
                // the row above starts W samples earlier: any byte alignment
                const uint32_t ub = (L.lead + off - W) * (uint32_t)sizeof(T);
                const uint32_t *ua = reinterpret_cast<const uint32_t *>(smem + L.span + (ub & ~3u));
                uint32_t d[NW + 2];
#pragma unroll
                for (uint32_t q = 0; q < NW + 2; q++) d[q] = ua[q];
#pragma unroll
                for (uint32_t q = 0; q < NW + 1; q++) uw[q] = __builtin_amdgcn_alignbyte(d[q + 1], d[q], ub & 3u);
            }
            int left = (int)span[L.lead + off - 1], left2 = (int)span[L.lead + off - 2];
            Coord xy;
            xy.set(first, W);
            uint32_t *cnt32 = reinterpret_cast<uint32_t *>(cnt16 + wave * NCTX);
#pragma unroll
            for (uint32_t j = 0; j < PPT; j++) {
                const uint32_t i = first + j;
                const int p = sample_at(cw, j, T());
                if (i < end && i >= 2) {
                    const int above = sample_at(uw, j, T());
                    int v1 = left, v2 = above;  // interior: left and above (misc.rs:6-24)
                    if (xy.y == 0) {
                        v2 = left2;  // first row: the two pixels to the left
                    } else if (xy.x == 0) {
                        v1 = above;  // first column: above and two rows up, or above-right for pixel (0,1)
                        v2 = xy.y >= 2 ? (int)pl[i - 2 * W] : sample_at(uw, j + 1, T());
                    }
                    const int Hh = max(v1, v2), Ll = min(v1, v2);
                    const uint32_t ctx = (uint32_t)(Hh - Ll);
                    const uint32_t cls = p < Ll ? CLS_BELOW : (p > Hh ? CLS_ABOVE : CLS_IN);
                    const uint32_t val = p < Ll ? (uint32_t)(Ll - p - 1) : (p > Hh ? (uint32_t)(p - Hh - 1) : (uint32_t)(p - Ll));
                    rec[j] = cls | (ctx << 2) | (val << 11);
                    if (cls != CLS_IN) {
                        nev++;
                        atomicAdd(&cnt32[ctx >> 1], 1u << (16u * (ctx & 1u)));
                    }
                }
                left2 = left;
                left = p;
                if (++xy.x == W) {
                    xy.x = 0;
                    xy.y++;
                }
            }
        }
        __syncthreads();
        stamp(ticket, 7);

        // ---------------- A3: per context: events per wave -> running offsets; segment starts (multiples of 4
        // events, so a block's events are whole dwords); blocks per context -> first block
        {
            uint32_t tot = 0;
            if (tid < NCTX)
                for (uint32_t w = 0; w < SW; w++) tot += cnt16[w * NCTX + tid];
            const uint32_t padded = (tot + 3u) & ~3u, nbk = (tot + EVB - 1) / EVB;
            const uint32_t ie = wave_incl_scan(padded), ib = wave_incl_scan(nbk);
            if (tid < NCTX && lane == 63) {
                misc[M_WSUM + wave] = ie;
                misc[M_WSUM2 + wave] = ib;
            }
            __syncthreads();
            if (tid < NCTX) {
                uint32_t oe = 0, ob = 0;
                for (uint32_t w = 0; w < wave; w++) {
                    oe += misc[M_WSUM + w];
                    ob += misc[M_WSUM2 + w];
                }
                const uint32_t s0 = oe + ie - padded, b0 = ob + ib - nbk;
                seg_start[tid] = (uint16_t)s0;
                seg_len[tid] = (uint16_t)tot;
                first_blk[tid] = (uint16_t)b0;
                if (tid == NCTX - 1) misc[M_NB] = b0 + nbk;
                uint32_t run = s0;
                for (uint32_t w = 0; w < SW; w++) {
                    const uint32_t v = cnt16[w * NCTX + tid];
                    cnt16[w * NCTX + tid] = (uint16_t)run;
                    run += v;
                }
            }
        }
        __syncthreads();
        stamp(ticket, 8);

        // ---------------- A4: stable partition of the events by context.  The wave's events go through a ring in
        // raster order (lane-major: a lane owns consecutive pixels), 32 lanes at a time; 64 ring entries at a time
        // are ranked among the lanes that share a context with one ballot per context bit.
        {
            uint32_t *ring = rings + wave * RING;
            uint16_t *run = cnt16 + wave * NCTX;
            for (uint32_t half = 0; half < 2; half++) {
                const bool mine = (lane >> 5) == half;
                const uint32_t mycnt = mine ? nev : 0u;
                const uint32_t incl = wave_incl_scan(mycnt);
                const uint32_t nring = readlane(incl, 63);
                if (nring == 0) continue;
                if (mycnt) {
                    uint32_t pos = incl - mycnt;
#pragma unroll
                    for (uint32_t j = 0; j < PPT; j++) {
                        const uint32_t r = rec[j], cls = r & 3u;
                        if (cls == CLS_BELOW || cls == CLS_ABOVE)
                            ring[pos++] = ((r >> 2) & 0x1FFu) << 23 | ((r >> 11) & 0x1FFu) << 14 | (off + j);
                    }
                }
                __builtin_amdgcn_wave_barrier();
                for (uint32_t h = 0; h < nring; h += 64) {
                    const bool ev = h + lane < nring;
                    const uint32_t entry = ring[(h + lane) & (RING - 1u)];
                    const uint32_t c = entry >> 23, e = (entry >> 14) & 0x1FFu, pixoff = entry & 0x3FFFu;
                    const uint64_t ev_mask = __ballot(ev);
                    uint32_t m_lo = (uint32_t)ev_mask, m_hi = (uint32_t)(ev_mask >> 32);
                    auto match_bit = [&](uint32_t b) {
                        const uint32_t t = (uint32_t)((int32_t)(c << (31 - b)) >> 31);  // all ones if bit b of c is set
                        const uint64_t bb = __ballot(ev && t != 0);
                        m_lo &= ~((uint32_t)bb ^ t);
                        m_hi &= ~((uint32_t)(bb >> 32) ^ t);
                    };
#pragma unroll
                    for (uint32_t b = 0; b < 5; b++) match_bit(b);
                    if (__ballot(ev && c >= 32u) != 0) {
#pragma unroll
                        for (uint32_t b = 5; b < 9; b++) match_bit(b);
                    }
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi(m_hi, __builtin_amdgcn_mbcnt_lo(m_lo, 0u));
                    const uint32_t group = (uint32_t)__popc(m_lo) + (uint32_t)__popc(m_hi);
                    const bool leader = ev && rank == 0;
                    uint32_t dest = 0;
                    if (ev) dest = (uint32_t)run[c] + rank;
                    __builtin_amdgcn_wave_barrier();
                    if (leader) run[c] = (uint16_t)(dest + group);
                    __builtin_amdgcn_wave_barrier();
                    if (ev) {
                        se[dest] = (ET)e;
                        sp[dest] = (uint16_t)pixoff;
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        __syncthreads();

        stamp(ticket, 1);
        // ---------------- B1: block table and the six length sums of every block (lane = block)
        const uint32_t NB = misc[M_NB];
        for (uint32_t b = tid; b < NB; b += STRIPE_THREADS) {
            // the context of block b: the last one whose first block is <= b
            uint32_t lo = 0, hi = NCTX;
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if ((uint32_t)first_blk[mid] <= b) lo = mid; else hi = mid;
            }
            const uint32_t c = lo, j = b - first_blk[c], len = seg_len[c];
            const uint32_t ev0 = (uint32_t)seg_start[c] + j * EVB, n = min(EVB, len - j * EVB);
            blk_ev0[b] = (uint16_t)ev0;
            blk_n[b] = (uint8_t)(n | (len <= EVB ? 0x80u : 0u));
            constexpr uint32_t EW = EVB * sizeof(ET) / 4, PER = 4 / sizeof(ET);  // dwords per block, events per dword
            const uint32_t *es = reinterpret_cast<const uint32_t *>(se + ev0);
            uint32_t B01 = n * (1u | (2u << 16)), B23 = n * (3u | (4u << 16)), B45 = n * (5u | (6u << 16));
#pragma unroll
            for (uint32_t q = 0; q < EW; q++) {
                const uint32_t keep = n > q * PER ? min(PER, n - q * PER) : 0u;  // events of this dword that exist
                const uint32_t mask = keep == PER ? 0xFFFFFFFFu : ((1u << (keep * 8u * (uint32_t)sizeof(ET))) - 1u);
                add_block_sums<ET>(es[q] & mask, B01, B23, B45);
            }
            bsum[b * 3] = B01;
            bsum[b * 3 + 1] = B23;
            bsum[b * 3 + 2] = B45;
        }
        // ---------------- B2: wait for the tile before this one in the plane
        if (wave == 0 && tile != 0) {
            uint32_t spins = 0;
            bool failed = false;
            while (true) {
                const uint32_t seen = __hip_atomic_load(&done[plane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (seen >= tile) break;
                if (++spins > LOOKBACK_SPIN_LIMIT) {
                    failed = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            if (failed && lane == 0) atomicOr(&a.ctl[STRIPE_CTL_ERROR], 1u);
        }
        __syncthreads();
        stamp(ticket, 2);

        // ---------------- B3: the estimator along this tile's events.  Wave w takes the contexts w, w + 16, ...:
        // a context whose events fit one block is replayed by one lane straight from its table row (32 such
        // contexts side by side); a longer one first gets the state at each of its blocks' starts.
        {
            uint64_t *rows = a.table + (uint64_t)plane * NCTX * 3;
            const uint32_t c = wave + SW * lane;  // lanes 0..31
            uint32_t len = 0;
            if (lane < NCTX / SW) len = seg_len[c];
            uint32_t S[6] = {0u, 0u, 0u, 0u, 0u, 0u};
            if (len != 0) {
                const uint64_t r0 = ld_agent(rows + c * 3), r1 = ld_agent(rows + c * 3 + 1), r2 = ld_agent(rows + c * 3 + 2);
                S[0] = (uint32_t)r0; S[1] = (uint32_t)(r0 >> 32); S[2] = (uint32_t)r1; S[3] = (uint32_t)(r1 >> 32);
                S[4] = (uint32_t)r2; S[5] = (uint32_t)(r2 >> 32);
            }
            uint64_t longer = __ballot(len > EVB);
            wstamp(ticket, 0);
            if (len != 0 && len <= EVB) {
                replay_block<ET>(S, se, sp, kq, seg_start[c], len);
                st_agent(rows + c * 3, (uint64_t)S[0] | ((uint64_t)S[1] << 32));
                st_agent(rows + c * 3 + 1, (uint64_t)S[2] | ((uint64_t)S[3] << 32));
                st_agent(rows + c * 3 + 2, (uint64_t)S[4] | ((uint64_t)S[5] << 32));
            }
            wstamp(ticket, 1);
            __builtin_amdgcn_s_setprio(3);  // a walk is one long dependent instruction stream
            while (longer) {
                const uint32_t l = (uint32_t)__builtin_ctzll(longer);
                longer &= longer - 1;
                const uint32_t cc = wave + SW * l;
                uint32_t Sc[6];
#pragma unroll
                for (uint32_t k = 0; k < 6; k++) Sc[k] = readlane(S[k], l);
                walk_context<ET>(Sc, se, seg_start[cc], seg_len[cc], first_blk[cc], bsum, bstate);
                if (lane < 3)
                    st_agent(rows + cc * 3 + lane, lane == 0 ? ((uint64_t)Sc[0] | ((uint64_t)Sc[1] << 32))
                                                   : lane == 1 ? ((uint64_t)Sc[2] | ((uint64_t)Sc[3] << 32))
                                                               : ((uint64_t)Sc[4] | ((uint64_t)Sc[5] << 32)));
            }
            __builtin_amdgcn_s_setprio(0);
            wstamp(ticket, 2);
        }
        // the rows must have left this CU before the token says so
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wstamp(ticket, 3);
        __syncthreads();
        if (tid == 0) __hip_atomic_store(&done[plane], tile + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        stamp(ticket, 3);

        // ---------------- B4: k of every event of the longer contexts, one block per lane
        for (uint32_t b = tid; b < NB; b += STRIPE_THREADS) {
            const uint32_t nflag = blk_n[b];
            if (nflag & 0x80u) continue;  // replayed in B3
            uint32_t S[6];
            unpack_state(bstate[b * 2], bstate[b * 2 + 1], S);
            replay_block<ET>(S, se, sp, kq, blk_ev0[b], nflag);
        }
        __syncthreads();
        stamp(ticket, 4);

        // ---------------- C1: this thread's bit string
        const bool first_plane = plane % a.po.planes_per_image == 0;
        const bool has_header = tile == 0 && tid == 0 && first_plane;
        auto emit_all = [&](auto &bw) {
            if (has_header) {  // write_header, format.rs:51-61
                bw.put(0x464C4353u, 32);  // "FLCS"
                bw.put((a.color << 8) | a.depth, 16);
                bw.put(W, 32);
                bw.put(a.H, 32);
            }
            stripe_walk<T, PPT>(smem + L.span, L.lead, off, pl, first, end, W, kq,
                                [&](uint32_t, uint32_t rv) {
                                    bw.put(rv, 32);  // write_signed(32, p): compression.rs:99-106
                                    if (npix == 1) bw.put(0u, 32);
                                },
                                [&](const PixelClass &pc, uint32_t k) { put_pixel(bw, pc, k); });
        };
        StripeBits lb;
        lb.begin(lbuf + tid);
        if (first < end) {
            emit_all(lb);
            lb.finish();
        }
        const uint32_t bits = lb.total;
        const uint32_t inc = wave_incl_scan(bits);
        if (lane == 63) misc[M_WSUM + wave] = inc;
        __syncthreads();
        stamp(ticket, 9);
        uint32_t woff = 0, tile_total = 0;
        for (uint32_t w = 0; w < SW; w++) {
            const uint32_t v = misc[M_WSUM + w];
            if (w < wave) woff += v;
            tile_total += v;
        }

        // ---------------- C2: offset of the tile in its plane: decoupled look-back (wave 0); the others clear the window
        uint64_t *my_status = a.status + (uint64_t)plane * a.ntiles + tile;
        for (uint32_t j = tid; j < S_WIN_WORDS; j += STRIPE_THREADS) win[j] = 0;
        if (wave == 0) {
            if (lane == 0)
                __hip_atomic_store(my_status, status_word(a.epoch, ST_AGGREGATE, tile_total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            uint64_t excl = 0;
            int64_t look = (int64_t)tile - 1;
            uint32_t spins = 0;
            bool failed = false;
            while (look >= 0) {
                const int64_t idx = look - (int64_t)lane;
                uint32_t state = ST_PREFIX;  // in front of tile 0: prefix 0
                uint64_t value = 0;
                if (idx >= 0) {
                    const uint64_t sw = __hip_atomic_load(a.status + (uint64_t)plane * a.ntiles + (uint64_t)idx, __ATOMIC_RELAXED,
                                                          __HIP_MEMORY_SCOPE_AGENT);
                    const uint32_t tag = (uint32_t)(sw >> ST_VALUE_BITS);
                    state = (tag >> 2) == (a.epoch & ST_EPOCH_MASK) ? (tag & 3u) : 0u;
                    value = sw & ((1ull << ST_VALUE_BITS) - 1ull);
                }
                const uint64_t pm = __ballot(state == ST_PREFIX), vm = __ballot(state != 0);
                const uint32_t fp = pm ? (uint32_t)__builtin_ctzll(pm) : 64u;
                const uint64_t need = fp >= 63u ? ~0ull : ((2ull << fp) - 1ull);
                if ((vm & need) != need) {
                    if (++spins > LOOKBACK_SPIN_LIMIT) {
                        failed = true;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                    continue;
                }
                const uint32_t agg = wave_incl_scan(lane < fp ? (uint32_t)value : 0u);
                excl += readlane(agg, 63);
                if (fp < 64u) {
                    excl += ((uint64_t)readlane((uint32_t)(value >> 32), fp) << 32) | readlane((uint32_t)value, fp);
                    break;
                }
                look -= 64;
            }
            if (failed) {
                if (lane == 0) atomicOr(&a.ctl[STRIPE_CTL_ERROR], 1u);
                excl = ~0ull >> 8;  // far beyond any slot: every store of this tile is dropped
            }
            if (lane == 0) {
                const uint64_t incl = failed ? 0ull : excl + tile_total;
                __hip_atomic_store(my_status, status_word(a.epoch, ST_PREFIX, incl), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                misc[M_TILE_LO] = (uint32_t)excl;
                misc[M_TILE_LO + 1] = (uint32_t)(excl >> 32);
                if (!failed) {
                    a.tile_bitoff[(uint64_t)plane * a.ntiles + tile] = excl;
                    a.tile_bits[(uint64_t)plane * a.ntiles + tile] = tile_total;
                    if (tile + 1 == a.ntiles) {
                        a.plane_carry[plane] = incl;
                        if (!first_plane && incl > a.po.plane_slot * 8u) atomicOr(&a.ctl[STRIPE_CTL_ERROR], 2u);  // the plane outgrew its scratch slot
                    }
                }
            }
        }
        __syncthreads();
        stamp(ticket, 10);
        const uint64_t tile_lo = (uint64_t)misc[M_TILE_LO] | ((uint64_t)misc[M_TILE_LO + 1] << 32), tile_hi = tile_lo + tile_total;
        const uint64_t my_lo = tile_lo + woff + inc - bits;
        uint64_t limit_words;  // a stream that outgrows its slot is cut (the host re-packs)
        uint32_t *out_words = plane_words(a.po, plane, limit_words);
        if (tile_total != 0) {
            const uint64_t first_word = tile_lo >> 5, last_word = (tile_hi - 1) >> 5;
            const bool first_shared = (tile_lo & 31u) != 0, last_shared = (tile_hi & 31u) != 0;
            const bool overflowed = bits > S_LOCAL_WORDS * 32u;
            // ---------------- C3: window by window
            for (uint64_t w0 = first_word; w0 <= last_word; w0 += S_WIN_WORDS) {
                if (w0 != first_word) {  // (the first window was cleared above)
                    __syncthreads();
                    for (uint32_t j = tid; j < S_WIN_WORDS; j += STRIPE_THREADS) win[j] = 0;
                    __syncthreads();
                }
                if (bits != 0 && ((my_lo + bits - 1) >> 5) >= w0 && (my_lo >> 5) < w0 + S_WIN_WORDS) {
                    if (!overflowed) {
                        const uint32_t shift = (uint32_t)(my_lo & 31u), nsrc = (bits + 31u) >> 5;
                        const uint64_t dw0 = my_lo >> 5;
                        uint32_t prev = 0;
                        for (uint32_t sidx = 0; sidx <= nsrc; sidx++) {
                            const uint32_t cur = sidx < nsrc ? lbuf[sidx * STRIPE_THREADS + tid] : 0u;
                            const uint32_t v = shift ? (prev << (32u - shift)) | (cur >> shift) : cur;
                            prev = cur;
                            const uint64_t rel = dw0 + sidx - w0;
                            if (v != 0 && rel < S_WIN_WORDS) atomicOr(&win[rel], v);
                        }
                    } else {  // more bits than the private buffer holds: build the codes again, straight into the window
                        LaneBits bw;
                        bw.win = win;
                        bw.win_words = S_WIN_WORDS;
                        bw.win_word0 = w0;
                        bw.begin(my_lo);
                        emit_all(bw);
                        bw.finish();
                    }
                }
                __syncthreads();
                for (uint32_t j = tid; j < S_WIN_WORDS; j += STRIPE_THREADS) {
                    const uint64_t aw = w0 + j;
                    if (aw > last_word) break;
                    const uint32_t v = win[j];
                    if (aw == first_word && first_shared) {
                        a.edge_first[(uint64_t)plane * a.ntiles + tile] = v;  // merged with the previous tile's last word later
                    } else if (aw == last_word && last_shared) {
                        a.edge_last[(uint64_t)plane * a.ntiles + tile] = v;
                    } else if (aw < limit_words) {
                        out_words[aw] = __builtin_bswap32(v);
                    }
                }
            }
        }
        __syncthreads();  // LDS is reused by the next tile
        stamp(ticket, 5);
    }
}

template <typename T>
uint32_t stripe_lds_bytes(uint32_t W) {
    return layout_for<T>(W).total;
}
template uint32_t stripe_lds_bytes<uint8_t>(uint32_t);
template uint32_t stripe_lds_bytes<int16_t>(uint32_t);

template <typename T>
hipError_t launch_stripe(hipStream_t s, const StripeArgs &a, uint32_t max_workgroups) {
    const uint32_t lds = stripe_lds_bytes<T>(a.W);
    if (lds > STRIPE_LDS_LIMIT) return hipErrorInvalidValue;
    static thread_local uint32_t granted = 0;  // per instantiation
    if (lds > granted) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_stripe<T>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)STRIPE_LDS_LIMIT);
        if (e != hipSuccess) return e;
        granted = STRIPE_LDS_LIMIT;
    }
    const uint64_t tickets = (uint64_t)a.ntiles * a.nplanes;
    const uint32_t grid = (uint32_t)std::min<uint64_t>(tickets, max_workgroups);
    if (grid == 0) return hipSuccess;
    const LaunchTiming lt = g_launch_timing;
    g_launch_timing = LaunchTiming{};
    if (lt.start)
        hipExtLaunchKernelGGL((k_stripe<T>), dim3(grid), dim3(STRIPE_THREADS), lds, s, lt.start, lt.stop, 0, a);
    else
        hipLaunchKernelGGL((k_stripe<T>), dim3(grid), dim3(STRIPE_THREADS), lds, s, a);
    return hipGetLastError();
}
template hipError_t launch_stripe<uint8_t>(hipStream_t, const StripeArgs &, uint32_t);
template hipError_t launch_stripe<int16_t>(hipStream_t, const StripeArgs &, uint32_t);

}  // namespace felics
