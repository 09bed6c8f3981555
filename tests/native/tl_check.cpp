// tl_check.cpp -- development check of the tile-local pipeline's stages, one by one, against a plain CPU model
// (runs on the GPU box: `make -C tests/native && felics_amd/_build/tl_check`).  It drives the kernels through the internal
// launch interface (felics_kernels.h), not the C ABI: what it pins is every intermediate array -- the front kernel's sorted
// tiles and run table, k_enum's record lists, the spine's start states, k of every event -- so that a parity failure of
// the whole encoder can be traced to a stage.  The parity tests proper are tests/test_gpu_parity.py (through the C ABI).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../felics_amd/csrc/felics_kernels.h"

using namespace felics;

#define CK(call)                                                                                  \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
            exit(2);                                                                              \
        }                                                                                         \
    } while (0)

static uint64_t splitmix(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

struct Event {
    uint32_t ctx, val, off;  // off: offset in the tile | above << 12
};

// CPU model of one plane: events per tile, sorted by context (stable)
template <typename T>
static void model_plane(const T *p, uint32_t W, uint32_t H, std::vector<std::vector<Event>> &tiles) {
    const uint32_t n = W * H, nt = (n + SORT_TILE - 1) / SORT_TILE;
    tiles.assign(nt, {});
    for (uint32_t i = 2; i < n; i++) {
        const uint32_t x = i % W, y = i / W;
        uint32_t a, b;
        if (x > 0 && y > 0) { a = i - 1; b = i - W; }
        else if (y == 0) { a = i - 1; b = i - 2; }
        else if (y >= 2) { a = i - W; b = i - 2 * W; }
        else { a = i - W; b = i - W + 1; }
        const int v1 = p[a], v2 = p[b], px = p[i];
        const int Hh = v1 > v2 ? v1 : v2, L = v1 < v2 ? v1 : v2;
        if (px >= L && px <= Hh) continue;
        Event e;
        e.ctx = (uint32_t)(Hh - L);
        e.val = px < L ? (uint32_t)(L - px - 1) : (uint32_t)(px - Hh - 1);
        e.off = (i % SORT_TILE) | (px > Hh ? 0x1000u : 0u);
        tiles[i / SORT_TILE].push_back(e);
    }
}

struct Est {
    uint32_t s[6] = {0, 0, 0, 0, 0, 0};
    uint32_t get_k() const {
        uint32_t best = 0;
        for (uint32_t k = 1; k < 6; k++)
            if (s[k] <= s[best]) best = k;
        return best;
    }
    void update(uint32_t e) {
        uint32_t mn = 0xFFFFFFFFu;
        for (uint32_t k = 0; k < 6; k++) {
            s[k] += (e >> k) + 1 + k;
            if (s[k] < mn) mn = s[k];
        }
        if (mn > 1024)
            for (uint32_t k = 0; k < 6; k++) s[k] /= 2;
    }
};

template <typename T, typename ET>
static int run_case(const char *name, const std::vector<T> &planes_h, uint32_t W, uint32_t H, uint32_t nplanes, int nslices, uint32_t mode,
                    bool verbose) {
    constexpr uint32_t NC = sizeof(T) == 1 ? 256 : 512;
    Geometry g{};
    g.W = W; g.H = H; g.npix = W * H; g.nimages = nplanes; g.planes_per_image = 1; g.nplanes = nplanes;
    g.sort_tiles = (g.npix + SORT_TILE - 1) / SORT_TILE; g.pack_tiles = g.sort_tiles; g.color = 0; g.depth = 0; g.nctx = NC;
    const uint32_t cap = tile_cap_default(NC, g.npix);
    const size_t ptiles = (size_t)nplanes * g.sort_tiles, slots = ptiles * cap, recs = slots / REC, nchains = (size_t)nplanes * NC;
    T *d_planes;
    ET *d_ev;
    uint16_t *d_pix;
    uint8_t *d_kq;
    uint32_t *d_runtab, *d_tslots, *d_flags, *d_nrec, *d_cstate;
    uint2 *d_desc, *d_seg;
    uint4 *d_state;
    CK(hipMalloc(&d_planes, planes_h.size() * sizeof(T) + 64));
    CK(hipMemcpy(d_planes, planes_h.data(), planes_h.size() * sizeof(T), hipMemcpyHostToDevice));
    CK(hipMalloc(&d_ev, slots * sizeof(ET) + 64));
    CK(hipMalloc(&d_pix, slots * 2 + 64));
    CK(hipMalloc(&d_kq, slots + 64));
    CK(hipMalloc(&d_runtab, ptiles * NC * 4));
    CK(hipMalloc(&d_tslots, ptiles * 4));
    CK(hipMalloc(&d_flags, 64));
    CK(hipMalloc(&d_nrec, 64 * 4));
    CK(hipMalloc(&d_cstate, nchains * 32));
    CK(hipMalloc(&d_desc, recs * 8 + 64));
    CK(hipMalloc(&d_seg, (size_t)nslices * nchains * 8));
    CK(hipMalloc(&d_state, recs * 16 + 64));
    CK(hipMemset(d_ev, 0xA5, slots * sizeof(ET)));
    CK(hipMemset(d_pix, 0xA5, slots * 2));
    CK(hipMemset(d_kq, 0xA5, slots));
    CK(hipMemset(d_runtab, 0xA5, ptiles * NC * 4));
    CK(hipMemset(d_desc, 0xA5, recs * 8));
    CK(hipMemset(d_state, 0xA5, recs * 16));
    CK(hipMemset(d_seg, 0xA5, (size_t)nslices * nchains * 8));
    CK(hipMemset(d_flags, 0, 64));
    CK(hipMemset(d_nrec, 0, 64 * 4));
    CK(hipMemset(d_cstate, 0, nchains * 32));
    TileLocal<ET> tl{d_ev, d_pix, d_kq, d_runtab, d_tslots, cap};
    std::vector<uint32_t> bounds(nslices + 1);
    for (int q = 0; q <= nslices; q++) bounds[q] = (uint32_t)((uint64_t)g.sort_tiles * q / nslices);
    auto slice_of = [&](int q) {
        const size_t r0 = (size_t)bounds[q] * nplanes * (cap / REC);
        return ChainSlice{d_desc + r0, d_seg + (size_t)q * nchains, d_nrec + q, d_state};
    };
    hipStream_t s = nullptr;
    for (int q = 0; q < nslices; q++) {
        launch_front<T, ET>(s, d_planes, tl, g, bounds[q], bounds[q + 1], d_flags, mode);
        if (bounds[q + 1] == bounds[q]) continue;
        launch_enum(s, d_runtab, slice_of(q), g, bounds[q], bounds[q + 1], cap);
        launch_spine3<ET>(s, d_ev, slice_of(q), d_cstate, d_flags, g);
        launch_assign3<ET>(s, tl, d_state, g, bounds[q], bounds[q + 1]);
    }
    CK(hipDeviceSynchronize());
    std::vector<ET> ev(slots);
    std::vector<uint16_t> pix(slots);
    std::vector<uint8_t> kq(slots);
    std::vector<uint32_t> runtab(ptiles * NC), tslots(ptiles), nrec(64), flags(16);
    std::vector<uint2> desc(recs), seg((size_t)nslices * nchains);
    std::vector<uint4> state(recs);
    CK(hipMemcpy(ev.data(), d_ev, slots * sizeof(ET), hipMemcpyDeviceToHost));
    CK(hipMemcpy(pix.data(), d_pix, slots * 2, hipMemcpyDeviceToHost));
    CK(hipMemcpy(kq.data(), d_kq, slots, hipMemcpyDeviceToHost));
    CK(hipMemcpy(runtab.data(), d_runtab, ptiles * NC * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(tslots.data(), d_tslots, ptiles * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(nrec.data(), d_nrec, 64 * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(flags.data(), d_flags, 64, hipMemcpyDeviceToHost));
    CK(hipMemcpy(desc.data(), d_desc, recs * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(seg.data(), d_seg, (size_t)nslices * nchains * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(state.data(), d_state, recs * 16, hipMemcpyDeviceToHost));
    int bad = 0;
    auto fail = [&](const char *what, size_t a, size_t b, long got, long want) {
        if (bad < 12) printf("  [%s] %s at (%zu, %zu): got %ld want %ld\n", name, what, a, b, got, want);
        bad++;
    };
    if (flags[0] != ((mode & FRONT_TEST_VIOLATION) ? TL_FLAG_ORDER : 0u)) fail("flags", 0, 0, flags[0], 0);
    std::vector<uint32_t> nrec_want(nslices, 0);
    for (uint32_t pl = 0; pl < nplanes; pl++) {
        std::vector<std::vector<Event>> tiles;
        model_plane<T>(planes_h.data() + (size_t)pl * g.npix, W, H, tiles);
        // ---- front: layout of every tile
        std::vector<std::vector<uint32_t>> start(g.sort_tiles, std::vector<uint32_t>(NC, 0)), cnt(g.sort_tiles, std::vector<uint32_t>(NC, 0));
        for (uint32_t t = 0; t < g.sort_tiles; t++) {
            const size_t pt = (size_t)pl * g.sort_tiles + t;
            for (const Event &e : tiles[t]) cnt[t][e.ctx]++;
            uint32_t ps = 0;
            for (uint32_t c = 0; c < NC; c++) {
                start[t][c] = ps;
                const uint32_t want = (ps / REC) | (cnt[t][c] << 16);
                const size_t ri = ((size_t)pl * NC + c) * g.sort_tiles + t;
                if (runtab[ri] != want) fail("runtab", pt, c, runtab[ri], want);
                ps += (cnt[t][c] + REC - 1) / REC * REC;
            }
            if (tslots[pt] != ps) fail("tile_slots", pt, 0, tslots[pt], ps);
            if (ps > cap) { printf("  [%s] tile needs %u slots > cap %u\n", name, ps, cap); return 1; }
            std::vector<uint32_t> fill(NC, 0);
            for (const Event &e : tiles[t]) {
                const size_t sl = pt * cap + start[t][e.ctx] + fill[e.ctx]++;
                if (ev[sl] != (ET)e.val) fail("ev", pt, sl - pt * cap, ev[sl], e.val);
                if (pix[sl] != e.off) fail("pix", pt, sl - pt * cap, pix[sl], e.off);
            }
            for (uint32_t c = 0; c < NC; c++)
                for (uint32_t i = cnt[t][c]; i < (cnt[t][c] + REC - 1) / REC * REC; i++)
                    if (pix[pt * cap + start[t][c] + i] != 0xFFFFu) fail("pad", pt, start[t][c] + i, pix[pt * cap + start[t][c] + i], 0xFFFF);
        }
        // ---- chains: records, states, k
        for (uint32_t c = 0; c < NC; c++) {
            const size_t chain = (size_t)pl * NC + c;
            Est est;
            for (int q = 0; q < nslices; q++) {
                if (bounds[q + 1] == bounds[q]) continue;
                const size_t r0 = (size_t)bounds[q] * nplanes * (cap / REC);
                const uint2 sg = seg[(size_t)q * nchains + chain];
                uint32_t want_n = 0;
                for (uint32_t t = bounds[q]; t < bounds[q + 1]; t++) want_n += (cnt[t][c] + REC - 1) / REC;
                if (sg.y != want_n) { fail("chain_seg.n", chain, q, sg.y, want_n); continue; }
                nrec_want[q] += want_n;
                uint32_t r = 0;
                for (uint32_t t = bounds[q]; t < bounds[q + 1]; t++) {
                    const size_t pt = (size_t)pl * g.sort_tiles + t;
                    for (uint32_t j = 0; j < (cnt[t][c] + REC - 1) / REC; j++, r++) {
                        const size_t ri = r0 + sg.x + r;
                        const uint32_t want_rec = (uint32_t)(pt * (cap / REC) + start[t][c] / REC + j);
                        const uint32_t want_cnt = cnt[t][c] - j * REC < REC ? cnt[t][c] - j * REC : REC;
                        if (desc[ri].x != want_rec) fail("desc.rec", chain, ri, desc[ri].x, want_rec);
                        if (desc[ri].y != want_cnt) fail("desc.cnt", chain, ri, desc[ri].y, want_cnt);
                        const uint4 st = state[want_rec];
                        const uint32_t got[6] = {st.x & 0xFFFFu, st.x >> 16, st.y & 0xFFFFu, st.y >> 16, st.z & 0xFFFFu, st.z >> 16};
                        for (int k = 0; k < 6; k++)
                            if (got[k] != est.s[k]) fail("state", chain, ri * 8 + k, got[k], est.s[k]);
                        if (st.w != want_rec) fail("state.rec", chain, ri, st.w, want_rec);
                        for (uint32_t i = 0; i < want_cnt; i++) {
                            const size_t sl = (size_t)want_rec * REC + i;
                            const uint32_t k = est.get_k();
                            if (kq[sl] != k) fail("k", chain, sl, kq[sl], k);
                            est.update((uint32_t)ev[sl]);
                        }
                    }
                }
            }
        }
    }
    for (int q = 0; q < nslices; q++)
        if (nrec[q] != nrec_want[q]) fail("nrec", q, 0, nrec[q], nrec_want[q]);
    if (verbose || bad) printf("[%s] %ux%u x %u planes, %d slices, mode %u: %s (%d mismatches), records %u\n", name, W, H, nplanes, nslices, mode,
                               bad ? "FAILED" : "ok", bad, nrec_want[0]);
    hipFree(d_planes); hipFree(d_ev); hipFree(d_pix); hipFree(d_kq); hipFree(d_runtab); hipFree(d_tslots); hipFree(d_flags);
    hipFree(d_nrec); hipFree(d_cstate); hipFree(d_desc); hipFree(d_seg); hipFree(d_state);
    return bad ? 1 : 0;
}

static std::vector<uint8_t> make_gray(uint32_t W, uint32_t H, uint32_t nplanes, int kind, uint64_t seed) {
    std::vector<uint8_t> v((size_t)W * H * nplanes);
    for (uint32_t p = 0; p < nplanes; p++)
        for (uint32_t y = 0; y < H; y++)
            for (uint32_t x = 0; x < W; x++) {
                const uint64_t h = splitmix(seed ^ ((uint64_t)p << 40) ^ ((uint64_t)y << 20) ^ x);
                int val;
                if (kind == 0) {  // S1-like: smooth + 3 bits of noise
                    auto tri = [](uint64_t t, uint64_t P) { const uint64_t a = (t % P) * 510 / P; return (int)(a <= 255 ? a : 510 - a); };
                    val = ((tri(3 * x + 17 * p, 1531) + tri(5 * y, 1187)) >> 1) + (int)(h & 7) - 3;
                } else if (kind == 1) {  // noise
                    val = (int)(h & 0xFF);
                } else if (kind == 2) {  // flat
                    val = 128;
                } else {  // spikes behind quiet data: long codes, sparse contexts
                    val = (h % 97 == 0) ? (int)(h >> 8 & 0xFF) : 40 + (int)(h & 1);
                }
                v[((size_t)p * H + y) * W + x] = (uint8_t)(val < 0 ? 0 : val > 255 ? 255 : val);
            }
    return v;
}

static std::vector<int16_t> make_i16(uint32_t W, uint32_t H, uint32_t nplanes, int kind, uint64_t seed) {
    std::vector<int16_t> v((size_t)W * H * nplanes);
    for (size_t i = 0; i < v.size(); i++) {
        const uint64_t h = splitmix(seed ^ i);
        const int base = kind == 0 ? (int)((i % W) / 7) - 100 + (int)(h & 7) : (int)(h % 511) - 255;
        v[i] = (int16_t)(base < -255 ? -255 : base > 255 ? 255 : base);
    }
    return v;
}

int main(int argc, char **argv) {
    const bool big = argc > 1 && !strcmp(argv[1], "big");
    int bad = 0;
    struct { uint32_t W, H, P; int kind, slices; uint32_t mode; } cases[] = {
        {64, 64, 1, 0, 1, 0},    {640, 360, 2, 0, 1, 0},   {640, 360, 2, 1, 1, 0},  {640, 360, 1, 2, 1, 0},   {640, 360, 2, 3, 2, 0},
        {1920, 1080, 3, 0, 4, 0}, {1920, 1080, 2, 1, 4, 0}, {1000, 700, 2, 0, 3, FRONT_SAFE_RANK}, {1000, 700, 2, 1, 6, FRONT_SAFE_RANK},
        {17, 9, 3, 1, 1, 0},     {3, 2, 1, 1, 1, 0},       {4099, 3, 2, 1, 2, 0},   {1, 300, 1, 1, 1, 0},     {640, 360, 1, 0, 1, FRONT_TEST_VIOLATION},
    };
    for (auto &c : cases) {
        char name[64];
        snprintf(name, sizeof name, "gray kind %d", c.kind);
        bad += run_case<uint8_t, uint8_t>(name, make_gray(c.W, c.H, c.P, c.kind, 0xFE11C5), c.W, c.H, c.P, c.slices, c.mode, true);
    }
    bad += run_case<int16_t, uint16_t>("i16 smooth", make_i16(640, 360, 3, 0, 7), 640, 360, 3, 2, 0, true);
    bad += run_case<int16_t, uint16_t>("i16 noise", make_i16(640, 360, 3, 1, 9), 640, 360, 3, 3, 0, true);
    bad += run_case<int16_t, uint16_t>("i16 noise safe", make_i16(333, 77, 2, 1, 11), 333, 77, 2, 1, FRONT_SAFE_RANK, true);
    if (big) {
        bad += run_case<uint8_t, uint8_t>("4K S1", make_gray(3840, 2160, 2, 0, 0xFE11C5), 3840, 2160, 2, 4, 0, true);
        bad += run_case<uint8_t, uint8_t>("4K noise", make_gray(3840, 2160, 2, 1, 0xFE11C5), 3840, 2160, 2, 6, 0, true);
    }
    printf(bad ? "tl_check: FAILED\n" : "tl_check: all stages match the CPU model\n");
    return bad ? 1 : 0;
}
