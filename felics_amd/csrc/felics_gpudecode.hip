// felics_gpudecode.hip -- GPU decoder for 8-bit streams (gfx950): decompress_channel of the reference
// (src/compression.rs:151-248, decode_intensity :48-61) for a whole batch of streams at once.
//
// Decoding is bit-serial per stream: the context of a pixel (its two neighbours, misc.rs:6-24) and the
// Rice parameter (the estimator's state, parameter_selection.rs:49-85) depend on everything decoded
// before it, and the planes of an RGB image share one bit stream (compression.rs:385-400: plane c + 1
// starts at the bit plane c ended on).  The only parallelism the format offers is ACROSS streams, so:
// one wave per stream, the estimator table (512 rows of six counters, 12 KiB) and the two image rows the
// neighbour rule looks at in LDS, the stream pulled in 256 bytes at a time with coalesced loads, finished
// rows stored coalesced.  A batch fills the chip from about a thousand streams up; a single stream runs at
// the speed of one wave's instruction stream: ~480 ns per pixel, whether that stream is vector code
// (round 2's first form: 112 MPix/s for 64 streams) or, as now, scalar code (134 MPix/s) -- a lone wave
// issues an instruction of either kind about every ten cycles and pays more for every taken branch.
//
// Valid streams decode to exactly the pixels the host decoder (felics_decode.cpp) produces.  Corrupt
// streams end with an error status (the code can differ from the host decoder's where both a range and a
// length check would fire), never with an out-of-bounds access: every read of the stream is bounded by its
// length and every pixel by the image size.
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>

#include "../../include/felics.h"
#include "felics_device.h"
#include "felics_kernels.h"

namespace felics {

// samples of an LDS row: whole 64-sample blocks (a block of the row above is read with one vector load)
__host__ __device__ static inline uint32_t decode8_row_stride(uint32_t W) { return (W + 63u) & ~63u; }

namespace {

__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ int unii(int v) { return __builtin_amdgcn_readfirstlane(v); }

// MSB-first bit reader over [base, base + len) in global memory (bitstream-io BitReader<_, BigEndian>).
// Everything about the position is wave-uniform and lives in scalar registers; the stream itself sits in two vector
// registers (lane j = dword j of a 256-byte chunk, big-endian order restored; the chunk after it already asked for),
// and the next dword is picked out of them with v_readlane: no LDS and no memory wait on the per-pixel path.
struct ScalarBits {
    const uint32_t *al;    // aligned-down dword pointer of the stream's first byte
    uint64_t total_dw;     // dwords from `al` that hold stream bytes
    uint64_t chunk0;       // dword index of cur's lane 0
    uint32_t cur, nxt;     // VECTOR: this chunk and the next one (zeros past the end)
    uint32_t cpos;         // next dword of `cur` to consume, 0..64
    uint64_t acc;          // unread bits, left-aligned
    uint32_t navail;       // valid bits in acc
    uint64_t end_bit;      // bits from `al` to the end of the stream

    // (whole aligned dwords: the first and the last one may hold up to three bytes that are not the stream's -- never
    // interpreted: `skew` bytes are dropped at the start, end_bit bounds the end -- and an aligned word that holds a stream
    // byte cannot leave the page that byte is in)
    __device__ __forceinline__ uint32_t fetch(uint64_t first) const {
        const uint64_t i = first + lane_id();
        return i < total_dw ? __builtin_bswap32(al[i]) : 0u;
    }
    __device__ __forceinline__ void init(const uint8_t *p, uint64_t n) {
        const uint32_t skew = (uint32_t)(reinterpret_cast<uintptr_t>(p) & 3u);
        al = reinterpret_cast<const uint32_t *>(p - skew);
        total_dw = (skew + n + 3u) >> 2;
        chunk0 = 0;
        cur = fetch(0);
        nxt = fetch(64);
        cpos = 0;
        acc = 0;
        navail = 0;
        end_bit = (skew + n) * 8u;
        refill();
        if (skew) {  // the first dword starts before the stream: drop those bytes
            acc <<= 8u * skew;
            navail -= 8u * skew;
        }
    }
    // at least 33 valid bits in acc afterwards (zeros past the end of the stream)
    __device__ __forceinline__ void refill() {
        if (navail <= 32u) {
            if (cpos == 64u) {
                cur = nxt;
                chunk0 += 64u;
                nxt = fetch(chunk0 + 64u);
                cpos = 0;
            }
            const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)cur, (int)cpos);
            cpos++;
            acc |= (uint64_t)w << (32u - navail);
            navail += 32u;
        }
    }
    // the next n <= 32 bits; the caller has refilled (n <= navail)
    __device__ __forceinline__ uint32_t take(uint32_t n) {
        const uint32_t v = n ? (uint32_t)(acc >> (64u - n)) : 0u;
        acc <<= n;
        navail -= n;
        return v;
    }
    __device__ __forceinline__ uint32_t get(uint32_t n) {
        refill();
        return take(n);
    }
    // Bits past the end of the stream read as zeros, and whether any were handed out is worked out from the position when
    // somebody asks (at the end of every row, and wherever decoding stops): DecompressionError::IoError.  Nothing loops on
    // stream content without a bound: a run of ones ends at the first padding zero.
    __device__ __forceinline__ bool failed() const { return (chunk0 + cpos) * 32u - navail > end_bit; }
    // ones before the first zero, the zero consumed (read_unary0)
    __device__ __forceinline__ uint64_t unary0() {
        uint64_t q = 0;
        while (true) {
            refill();
            const uint32_t top = (uint32_t)(acc >> 32);
            const uint32_t ones = top == 0xFFFFFFFFu ? 32u : (uint32_t)__builtin_clz(~top);
            if (ones == 32u) {
                q += 32u;
                take(32u);
                if (failed()) return q;
                continue;
            }
            take(ones + 1u);
            return q + ones;
        }
    }
};

}  // namespace

// One wave per stream.  status[i] = FELICS_OK or an error code.  Gray: u8 pixels straight to `pixels`;
// RGB: the three planes as int16 to `planes` (image i at i * 3 * npix), converted by k_ycocg8_to_rgb.
// LDS (dynamic): table (256 or 512) x 6 u32 | rows 2 x rstride i16.
//
// The decode loop is one pixel after the other, and a lone wave retires a dependent vector instruction every ~10
// cycles: the per-pixel path is therefore written so that the compiler keeps it in SCALAR registers and instructions
// (every value that comes out of LDS or a vector register passes through readfirstlane / readlane, nothing depends on
// the lane id).  The vector side only moves data in bulk: the stream 256 bytes at a time, the row above 64 samples at
// a time (one register, read with v_readlane), the decoded row 64 samples at a time (collected with v_writelane,
// stored to LDS for the next row and to global memory coalesced).
__global__ __launch_bounds__(64) void k_decode8(const uint8_t *__restrict__ streams, const uint64_t *__restrict__ offsets,
                                                const uint64_t *__restrict__ lens, uint32_t W, uint32_t H, uint32_t color,
                                                uint8_t *__restrict__ pixels, int16_t *__restrict__ planes,
                                                int *__restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint32_t *table = reinterpret_cast<uint32_t *>(smem);
    const uint32_t nctx = color ? nctx_of<int16_t>() : nctx_of<uint8_t>();  // gray: contexts 0..255, half the table
    int16_t *rows = reinterpret_cast<int16_t *>(smem + nctx * 6 * 4);
    const uint32_t rstride = decode8_row_stride(W);
    const uint32_t img = blockIdx.x, lane = lane_id();
    const uint8_t *s = streams + offsets[img];
    const uint64_t slen = lens[img];
    const uint64_t npix = (uint64_t)W * H;
    const uint32_t nplanes = color ? 3u : 1u;
    // header (format.rs:63-84) must be the one the caller announced
    int rc = FELICS_OK;
    if (slen < FELICS_HEADER_BYTES) {
        rc = FELICS_E_IO;
    } else {
        const uint32_t w = ((uint32_t)s[6] << 24) | ((uint32_t)s[7] << 16) | ((uint32_t)s[8] << 8) | s[9];
        const uint32_t h = ((uint32_t)s[10] << 24) | ((uint32_t)s[11] << 16) | ((uint32_t)s[12] << 8) | s[13];
        if (s[0] != 'F' || s[1] != 'L' || s[2] != 'C' || s[3] != 'S') rc = FELICS_E_INVALID_SIGNATURE;
        else if (s[4] > 1) rc = FELICS_E_INVALID_COLOR_TYPE;
        else if (s[5] > 1) rc = FELICS_E_INVALID_PIXEL_DEPTH;
        else if (s[4] != color || s[5] != 0 || w != W || h != H) rc = FELICS_E_INVALID_DIMENSIONS;
    }
    rc = unii(rc);
    if (rc != FELICS_OK) {
        if (lane == 0) status[img] = rc;
        return;
    }
    ScalarBits br;
    br.init(s + FELICS_HEADER_BYTES, slen - FELICS_HEADER_BYTES);
    for (uint32_t c = 0; c < nplanes && rc == FELICS_OK; c++) {
        const int32_t p0 = (int32_t)br.get(32), p1 = (int32_t)br.get(32);  // compression.rs:166-167
        if (br.failed()) {
            rc = FELICS_E_IO;
            break;
        }
        if (npix == 0) continue;
        for (uint32_t i = lane; i < nctx * 6; i += 64) table[i] = 0;  // KEstimator::new
        __builtin_amdgcn_wave_barrier();
        int16_t *outp = planes ? planes + ((uint64_t)img * nplanes + c) * npix : nullptr;
        uint8_t *outg = planes ? nullptr : pixels + (uint64_t)img * npix;
        const int lo_ok = color ? -255 : 0, hi_ok = 255;  // what a sample of this plane can be (Y 0..255, Co / Cg -255..255)
        // rows: cur = the row being decoded, prev = the one above; both in LDS, written 64 samples at a time
        uint32_t x = 0, y = 0;
        int16_t *cur = rows, *prev = rows + rstride;
        int upv = 0;   // VECTOR: prev[xb + lane] for the 64-sample block xb the walk stands in
        int rowv = 0;  // VECTOR: the samples of this block decoded so far
        int left = 0, left2 = 0;
        int first_col2 = 0;  // cur[0] as it was before this row: the sample two rows up (first-column rule)
        for (uint64_t i = 0; i < npix; i++) {
            const uint32_t xl = x & 63u;
            if (xl == 0) {
                if (y > 0) upv = (int)prev[x + lane];  // (rows are padded to whole blocks)
                // second neighbour of the row's first pixel (misc.rs:14-23): two rows up -- what cur[0] still holds --
                // or, in row 1, above-right
                if (x == 0 && y > 0) first_col2 = y >= 2 ? unii((int)cur[0]) : (W > 1 ? __builtin_amdgcn_readlane(upv, 1) : 0);
            }
            int pv;
            if (i < 2) {
                pv = i == 0 ? p0 : p1;
            } else {
                const int above = __builtin_amdgcn_readlane(upv, (int)xl);
                // misc.rs:6-24 with selects: interior = left and above; first row = the two to the left; first column =
                // above and first_col2
                const bool row0 = y == 0, col0 = x == 0 && !row0;
                const int v1 = col0 ? above : left;
                const int v2 = col0 ? first_col2 : (row0 ? left2 : above);
                const int hi = max(v1, v2), lo = min(v1, v2);
                const uint32_t ctx = (uint32_t)(hi - lo);  // <= 510 because every stored sample is in range
                br.refill();  // >= 33 bits: an in-range code has at most 11, the two flags of the other kind 2
                if (br.take(1)) {  // in range: phased-in code of p - L (phase_in_coding.rs:86-112)
                    const uint32_t n = ctx + 1;
                    const uint32_t m = 31u - (uint32_t)__builtin_clz(n);
                    const uint32_t right_p = (2u << m) - n, left_p = n - (1u << m);
                    uint32_t r = br.take(m);
                    const uint32_t longer = r >= right_p ? 1u : 0u;  // the code has one more bit
                    const uint32_t r2 = (r - right_p) * 2u + right_p + br.take(longer);
                    r = longer ? r2 : r;
                    uint32_t rot = r + left_p;  // rotate_left: (r + left_p) mod n, r < n
                    rot = rot >= n ? rot - n : rot;
                    pv = lo + (int)rot;
                } else {
                    const bool above_flag = br.take(1) != 0;
                    const uint64_t *row = reinterpret_cast<const uint64_t *>(table + ctx * 6);  // 24-byte rows: 8-byte aligned
                    const uint64_t r01 = row[0], r23 = row[1], r45 = row[2];                 // one LDS round trip for the row
                    uint32_t S[6] = {uni((uint32_t)r01), uni((uint32_t)(r01 >> 32)), uni((uint32_t)r23),
                                     uni((uint32_t)(r23 >> 32)), uni((uint32_t)r45), uni((uint32_t)(r45 >> 32))};
                    // get_k: smallest counter, ties to the largest k (parameter_selection.rs:71-85)
                    const uint32_t key = min(min(min((S[0] << 3) | 7u, (S[1] << 3) | 6u), min((S[2] << 3) | 5u, (S[3] << 3) | 4u)),
                                             min((S[4] << 3) | 3u, (S[5] << 3) | 2u));
                    const uint32_t k = 7u - (key & 7u);
                    const uint64_t q = br.unary0();
                    const uint64_t e64 = (q << k) + br.get(k);
                    if (e64 > 1024u) {  // no sample of an 8-bit plane is that far from its neighbours
                        rc = e64 > 0xFFFFFFFFull ? FELICS_E_VALUE_OVERFLOW : FELICS_E_INVALID_VALUE;
                        break;
                    }
                    const uint32_t e = (uint32_t)e64;
                    uint32_t mn = 0xFFFFFFFFu;
#pragma unroll
                    for (uint32_t kk = 0; kk < 6; kk++) {
                        S[kk] += (e >> kk) + 1u + kk;
                        mn = min(mn, S[kk]);
                    }
                    const uint32_t hsh = mn > 1024u ? 1u : 0u;
                    uint64_t *wrow = reinterpret_cast<uint64_t *>(table + ctx * 6);  // (every lane: same address, same value)
                    wrow[0] = (uint64_t)(S[0] >> hsh) | ((uint64_t)(S[1] >> hsh) << 32);
                    wrow[1] = (uint64_t)(S[2] >> hsh) | ((uint64_t)(S[3] >> hsh) << 32);
                    wrow[2] = (uint64_t)(S[4] >> hsh) | ((uint64_t)(S[5] >> hsh) << 32);
                    pv = above_flag ? hi + (int)e + 1 : lo - (int)e - 1;
                }
            }
            if (pv < lo_ok || pv > hi_ok) {  // try_into::<u8>() / the estimator's context bound would fail
                rc = FELICS_E_INVALID_VALUE;
                break;
            }
            rowv = lane == xl ? pv : rowv;  // (off the serial path: nothing reads rowv before the block is complete)
            left2 = left;
            left = pv;
            const bool row_end = x + 1 == W;
            if (xl == 63u || row_end) {  // a block of the row is complete: to LDS (the next row's `above`) and to the output
                const uint32_t xb = x & ~63u;
                if (xb + lane <= x) {
                    cur[xb + lane] = (int16_t)rowv;
                    if (outg)
                        outg[(uint64_t)y * W + xb + lane] = (uint8_t)rowv;
                    else
                        outp[(uint64_t)y * W + xb + lane] = (int16_t)rowv;
                }
            }
            if (row_end) {
                if (br.failed()) {
                    rc = FELICS_E_IO;
                    break;
                }
                __builtin_amdgcn_wave_barrier();
                x = 0;
                y++;
                int16_t *t = cur;
                cur = prev;
                prev = t;
            } else {
                x++;
            }
        }
    }
    if (br.failed()) rc = FELICS_E_IO;  // (whatever else stopped the decoding: it was decoding padding)
    if (lane == 0) status[img] = rc;
}

// ------------------------------------------------------------------------------------------
// Sixty-four streams per wave, LANE = stream (8-bit gray, batches of hundreds of streams and more).
//
// The streams of a call have one shape, so all of them are at the same pixel at the same time: the walk over (x, y) and the
// neighbour rule's case analysis are wave-uniform, and only what depends on a stream's content differs between the lanes --
// the bit reader (a 64-bit window, the position and one prefetched dword per lane), the two neighbours, the estimator row.
// Per pixel every lane executes both kinds of code under its own flag; a wave decodes 64 pixels in about the time
// k_decode8 decodes one, at the price of memory operations that are gathers (64 addresses per instruction):
//   * the stream: one dword per lane every ~8 pixels, asked for a whole dword ahead;
//   * the row above: read back from the stream's own OUTPUT image, four samples per lane at a time (a CU sees its own
//     stores; 64 rows do not fit in LDS: 3840 x 64 bytes);
//   * the estimator table: 256 contexts x six 16-bit counters = 3 KB per stream (a counter of an 8-bit gray plane stays
//     below 2^16: the smallest grows by >= 6 per event, the largest by <= 255, so a halving period adds at most 44 000 on
//     top of the halved rest).  The first DEC8L_HOT contexts of every stream live in LDS, one dword per lane and pair of
//     counters (lane-interleaved: no bank conflict whatever the contexts), the others in a zeroed table in HBM: smooth and
//     natural content keeps to the LDS rows, noise pays an L2 round trip per event.
// Error behaviour as in k_decode8: every read of a stream is bounded by its length, a lane that has failed keeps walking
// (on zeros, its samples clamped into range so that contexts stay inside the table) and reports its first error.
// Needs W >= 8 (the read-back of the row above looks four samples ahead of a row's end).
// ------------------------------------------------------------------------------------------

constexpr uint32_t DEC8L_HOT = 32;                 // contexts per stream in LDS: 64 x 32 x 12 B = 24 KB per wave
constexpr uint32_t DEC8L_TABLE_DW = 256 * 3;       // dwords per stream in HBM: 256 contexts x three pairs of u16 counters
constexpr uint32_t DEC8L_TABLE_DW_RGB = 512 * 3;   // per plane of an RGB stream (contexts 0 .. 510)

namespace {

// MSB-first bit reader of ONE LANE over [base, base + len) (bitstream-io BitReader<_, BigEndian>)
struct LaneReader {
    const uint32_t *al;   // aligned-down dword pointer of the stream's first byte
    uint32_t total_dw;    // dwords from `al` that hold stream bytes
    uint32_t pos;         // dwords moved into acc so far
    uint32_t nxt;         // dword `pos` as it lies in memory (zero past the end): asked for when dword pos - 1 was taken and first
                          // LOOKED AT when it is taken itself (the byte swap at the load would be a wait for the load)
    uint64_t acc;         // unread bits, left-aligned
    uint32_t navail;      // valid bits in acc
    uint64_t end_bit;     // bits from `al` to the end of the stream

    __device__ __forceinline__ uint32_t fetch(uint32_t i) const { return i < total_dw ? al[i] : 0u; }
    __device__ __forceinline__ void init(const uint8_t *p, uint64_t n) {
        const uint32_t skew = (uint32_t)(reinterpret_cast<uintptr_t>(p) & 3u);
        al = reinterpret_cast<const uint32_t *>(p - skew);
        total_dw = (uint32_t)std::min<uint64_t>((skew + n + 3u) >> 2, 0xFFFFFFFFull);
        pos = 0;
        nxt = fetch(0);
        acc = 0;
        navail = 0;
        end_bit = (skew + n) * 8u;
        refill();
        if (skew) {  // the first dword starts before the stream: drop those bytes
            acc <<= 8u * skew;
            navail -= 8u * skew;
        }
    }
    // at least 33 valid bits in acc afterwards (zeros past the end of the stream)
    __device__ __forceinline__ void refill() {
        if (navail <= 32u) {
            acc |= (uint64_t)__builtin_bswap32(nxt) << (32u - navail);
            navail += 32u;
            pos++;
            nxt = fetch(pos);
        }
    }
    __device__ __forceinline__ uint32_t take(uint32_t n) {  // the next n <= 32 bits; the caller has refilled (n <= navail)
        const uint32_t v = n ? (uint32_t)(acc >> (64u - n)) : 0u;
        acc <<= n;
        navail -= n;
        return v;
    }
    __device__ __forceinline__ uint32_t get(uint32_t n) {
        refill();
        return take(n);
    }
    __device__ __forceinline__ bool failed() const { return (uint64_t)pos * 32u - navail > end_bit; }
    __device__ __forceinline__ uint64_t unary0() {  // ones before the first zero, the zero consumed (read_unary0)
        uint64_t q = 0;
        while (true) {
            refill();
            const uint32_t top = (uint32_t)(acc >> 32);
            const uint32_t ones = top == 0xFFFFFFFFu ? 32u : (uint32_t)__builtin_clz(~top);
            if (ones == 32u) {
                q += 32u;
                take(32u);
                if (failed()) return q;
                continue;
            }
            take(ones + 1u);
            return q + ones;
        }
    }
};

// four consecutive samples of a stream's own output plane (one unaligned load / store; the plane is this wave's to write and to read)
template <typename ST>
struct Four;
template <>
struct Four<uint8_t> {
    uint32_t v;
    __device__ __forceinline__ void clear() { v = 0; }
    __device__ __forceinline__ void load(const uint8_t *p) { __builtin_memcpy(&v, p, 4); }
    __device__ __forceinline__ void store(uint8_t *p) const { __builtin_memcpy(p, &v, 4); }
    __device__ __forceinline__ int get(uint32_t j) const { return (int)((v >> (8u * j)) & 0xFFu); }
    __device__ __forceinline__ void set(uint32_t j, int s) { v |= (uint32_t)s << (8u * j); }  // (0 <= s <= 255, the field still zero)
};
template <>
struct Four<int16_t> {
    uint32_t lo, hi;  // (two named dwords: an array indexed by the sample's number went to scratch memory)
    __device__ __forceinline__ void clear() { lo = hi = 0; }
    __device__ __forceinline__ void load(const int16_t *p) {
        uint32_t v[2];
        __builtin_memcpy(v, p, 8);
        lo = v[0];
        hi = v[1];
    }
    __device__ __forceinline__ void store(int16_t *p) const {
        const uint32_t v[2] = {lo, hi};
        __builtin_memcpy(p, v, 8);
    }
    __device__ __forceinline__ int get(uint32_t j) const { return (int)(int16_t)(((j & 2u) ? hi : lo) >> (16u * (j & 1u))); }
    __device__ __forceinline__ void set(uint32_t j, int s) {  // (the field still zero)
        const uint32_t f = ((uint32_t)s & 0xFFFFu) << (16u * (j & 1u));
        lo |= (j & 2u) ? 0u : f;
        hi |= (j & 2u) ? f : 0u;
    }
};

}  // namespace

// RGB = false: gray8 streams, u8 frames straight to `out_base`.  RGB = true: the three planes of an RGB8 stream, one after the other from
// the same bit reader (compression.rs:385-400), as int16 planes (image i at i * 3 * npix; k_ycocg8_to_rgb converts them): samples
// -255 .. 255, contexts 0 .. 510, a zeroed table of its own per plane.  A counter still fits 16 bits: the Rice operand is at most 1024
// (larger is the corrupt-stream exit), so counter 0 gains at most 1025 per event while counter 5 gains at least 38, i.e. at most 27.7 K
// before the smallest counter passes 1024 and the row is halved -- below 56 K with the halved rest on top.
template <bool RGB>
__global__ __launch_bounds__(64) void k_decode8_lanes(const uint8_t *__restrict__ streams, const uint64_t *__restrict__ offsets,
                                                      const uint64_t *__restrict__ lens, uint32_t n, uint32_t W, uint32_t H,
                                                      void *out_base, uint32_t *table, int *__restrict__ status) {
    using ST = typename std::conditional<RGB, int16_t, uint8_t>::type;
    constexpr uint32_t NP = RGB ? 3u : 1u;
    constexpr uint32_t TABLE_DW = RGB ? DEC8L_TABLE_DW_RGB : DEC8L_TABLE_DW;  // per plane
    constexpr int LO_OK = RGB ? -255 : 0, HI_OK = 255;
    __shared__ uint32_t hot[DEC8L_HOT * 3 * 64];  // [context][pair of counters][lane]
    const uint32_t lane = lane_id();
    const uint32_t img = blockIdx.x * 64 + lane;
    if (img >= n) return;
    const uint8_t *s = streams + offsets[img];
    const uint64_t slen = lens[img];
    const uint64_t npix = (uint64_t)W * H;
    // header (format.rs:63-84) must be the one the caller announced
    int rc = FELICS_OK;
    if (slen < FELICS_HEADER_BYTES) {
        rc = FELICS_E_IO;
    } else {
        const uint32_t w = ((uint32_t)s[6] << 24) | ((uint32_t)s[7] << 16) | ((uint32_t)s[8] << 8) | s[9];
        const uint32_t h = ((uint32_t)s[10] << 24) | ((uint32_t)s[11] << 16) | ((uint32_t)s[12] << 8) | s[13];
        if (s[0] != 'F' || s[1] != 'L' || s[2] != 'C' || s[3] != 'S') rc = FELICS_E_INVALID_SIGNATURE;
        else if (s[4] > 1) rc = FELICS_E_INVALID_COLOR_TYPE;
        else if (s[5] > 1) rc = FELICS_E_INVALID_PIXEL_DEPTH;
        else if (s[4] != (RGB ? 1 : 0) || s[5] != 0 || w != W || h != H) rc = FELICS_E_INVALID_DIMENSIONS;
    }
    if (rc != FELICS_OK) {  // nothing of this stream is decoded (its lane leaves; the others go on)
        status[img] = rc;
        return;
    }
    LaneReader br;
    br.init(s + FELICS_HEADER_BYTES, slen - FELICS_HEADER_BYTES);
    uint32_t *myhot = hot + lane;
    for (uint32_t plane = 0; plane < NP; plane++) {
    for (uint32_t i = 0; i < DEC8L_HOT * 3; i++) myhot[i * 64] = 0;  // KEstimator::new (the HBM rows arrive zeroed); a lane's own column
    const int32_t p0 = (int32_t)br.get(32), p1 = (int32_t)br.get(32);  // compression.rs:166-167
    if (br.failed() && rc == FELICS_OK) rc = FELICS_E_IO;
    if (npix == 0) continue;
    uint32_t *tab = table + ((uint64_t)img * NP + plane) * TABLE_DW;
    ST *out = reinterpret_cast<ST *>(out_base) + ((uint64_t)img * NP + plane) * npix;
    // (x, y) and everything derived from them alone is wave-uniform: every stream has the same shape
    int left = 0, left2 = 0;
    Four<ST> up4, up4_next, out4;
    up4.clear();
    up4_next.clear();
    out4.clear();
    uint32_t out_of_range = 0;  // gray: OR of every sample as decoded (above 255 if one did not fit); RGB: nonzero if one was outside LO_OK .. HI_OK
    int first_col2 = 0;
    for (uint32_t y = 0; y < H; y++) {
      ST *row = out + (uint64_t)y * W;  // this row of the stream's plane, and the one above it
      const ST *prow = row - W;
      if (y > 0) {
          up4.load(prow);                   // row above, samples 0 .. 3 (later groups are asked for four samples ahead)
          if (4 < W) up4_next.load(prow + 4);
          // second neighbour of a row's first pixel (misc.rs:14-23): two rows up, or above-right in row 1
          first_col2 = y >= 2 ? (int)prow[-(int64_t)W] : (W > 1 ? up4.get(1) : 0);
      }
      for (uint32_t x = 0; x < W; x++) {
        const uint32_t xs = x & 3u;
        if (xs == 0 && y > 0 && x != 0) {
            up4 = up4_next;
            if (x + 4 < W) up4_next.load(prow + x + 4);
        }
        int pv;
        if (y == 0 && x < 2) {
            pv = x == 0 ? p0 : p1;
        } else {
            const int above = up4.get(xs);
            const bool row0 = y == 0, col0 = x == 0 && !row0;
            const int v1 = col0 ? above : left;
            const int v2 = col0 ? first_col2 : (row0 ? left2 : above);
            const int hi = max(v1, v2), lo = min(v1, v2);
            const uint32_t ctx = (uint32_t)(hi - lo);  // <= 255 (510): every sample kept is in range
            // The context's row for every lane, whether its pixel turns out to be an event or not (no divergence, and the LDS
            // round trip runs beside the arithmetic below): three pairs of 16-bit counters; hot contexts from LDS.
            const bool is_hot = ctx < DEC8L_HOT;
            const uint32_t hrow = min(ctx, DEC8L_HOT - 1u) * 3u * 64u;
            uint32_t w01 = myhot[hrow], w23 = myhot[hrow + 64], w45 = myhot[hrow + 128];
            br.refill();  // >= 33 valid bits: both kinds of code are read off the top 32 of them, then consumed in one go
            const uint32_t top = (uint32_t)(br.acc >> 32);
            const bool in_range = (top >> 31) != 0;
            // -- in range: `1`, then the phased-in code of p - L in m or m + 1 bits (phase_in_coding.rs:86-112), m <= 8 (context 255: n = 256)
            const uint32_t nn = ctx + 1;
            const uint32_t m = 31u - (uint32_t)__builtin_clz(nn);
            const uint32_t right_p = (2u << m) - nn, left_p = nn - (1u << m);
            const uint32_t t1 = top << 1;
            uint32_t r = (t1 >> 1) >> (31u - m);               // the m bits behind the flag
            const uint32_t extra = (t1 >> (31u - m)) & 1u;      // the bit behind them
            const uint32_t longer = r >= right_p ? 1u : 0u;     // the code has one more bit
            r = longer ? (r - right_p) * 2u + right_p + extra : r;
            uint32_t rot = r + left_p;                          // rotate_left: (r + left_p) mod n, r < n
            rot = rot >= nn ? rot - nn : rot;
            const int pv_in = lo + (int)rot;
            const uint32_t bits_in = 1u + m + longer;
            // -- out of range: `0`, above / below flag, q ones, `0`, k bits -- off the same 32 bits when it fits in them
            const bool above_flag = ((top >> 30) & 1u) != 0;
            if (!in_range && !is_hot) {  // (noise: a cold context's row comes from the stream's table in HBM)
                w01 = tab[ctx * 3 + 0];
                w23 = tab[ctx * 3 + 1];
                w45 = tab[ctx * 3 + 2];
            }
            uint32_t S[6] = {w01 & 0xFFFFu, w01 >> 16, w23 & 0xFFFFu, w23 >> 16, w45 & 0xFFFFu, w45 >> 16};
            // get_k: smallest counter, ties to the largest k (parameter_selection.rs:71-85)
            const uint32_t key = min(min(min((S[0] << 3) | 7u, (S[1] << 3) | 6u), min((S[2] << 3) | 5u, (S[3] << 3) | 4u)),
                                     min((S[4] << 3) | 3u, (S[5] << 3) | 2u));
            const uint32_t k = 7u - (key & 7u);
            const uint32_t t2 = top << 2;                                  // 30 bits of the stream, two zeros behind them
            const uint32_t ones = (uint32_t)__builtin_clz(~t2);            // (<= 30: ~t2 ends in ones)
            const bool fits = ones + 1u + k <= 30u;                        // unary part, its zero and the k bits lie inside the 30
            uint32_t e = (ones << k) + (((t2 << (ones & 31u)) << 1 >> 1) >> (31u - k));  // k bits behind the zero (k <= 5)
            uint32_t nbits = in_range ? bits_in : 3u + ones + k;
            if (!in_range && !fits) {
                // a long code (or the end of the stream): the general reader, bit field by bit field
                br.take(2);
                const uint64_t q = br.unary0();
                const uint64_t e64 = (q << k) + br.get(k);
                e = (uint32_t)e64;
                if (e64 > 1024u) {  // no sample of an 8-bit plane is that far from its neighbours
                    if (rc == FELICS_OK) rc = e64 > 0xFFFFFFFFull ? FELICS_E_VALUE_OVERFLOW : FELICS_E_INVALID_VALUE;
                    e = 0;
                }
                nbits = 0;
            }
            br.acc <<= nbits;  // (nbits <= 32 < the valid bits)
            br.navail -= nbits;
            if (!in_range) {
                // update (parameter_selection.rs:49-68): add the six Rice lengths, halve when the smallest passes 1024
                uint32_t mn = 0xFFFFFFFFu;
#pragma unroll
                for (uint32_t kk = 0; kk < 6; kk++) {
                    S[kk] += (e >> kk) + 1u + kk;
                    mn = min(mn, S[kk]);
                }
                const uint32_t hsh = mn > 1024u ? 1u : 0u;
                w01 = (S[0] >> hsh) | ((S[1] >> hsh) << 16);
                w23 = (S[2] >> hsh) | ((S[3] >> hsh) << 16);
                w45 = (S[4] >> hsh) | ((S[5] >> hsh) << 16);
                if (is_hot) {
                    myhot[hrow] = w01;
                    myhot[hrow + 64] = w23;
                    myhot[hrow + 128] = w45;
                } else {
                    tab[ctx * 3 + 0] = w01;
                    tab[ctx * 3 + 1] = w23;
                    tab[ctx * 3 + 2] = w45;
                }
            }
            pv = in_range ? pv_in : (above_flag ? hi + (int)e + 1 : lo - (int)e - 1);
        }
        // try_into::<u8>() (for RGB: the estimator's context bound) would fail on anything outside LO_OK .. HI_OK: remembered and
        // reported at the end of the row; the sample is cut into the range so that a failed stream's contexts stay inside the table
        if (RGB) {
            out_of_range |= (uint32_t)(pv - LO_OK) > (uint32_t)(HI_OK - LO_OK) ? 1u : 0u;
            pv = min(max(pv, LO_OK), HI_OK);
        } else {
            out_of_range |= (uint32_t)pv;  // (two instructions per pixel: above 255 if one did not fit eight bits, negative ones included)
            pv &= 255;
        }
        out4.set(xs, pv);
        left2 = left;
        left = pv;
        if (xs == 3u) {  // four samples complete: one (unaligned) store to the stream's plane
            out4.store(row + (x - 3u));
            out4.clear();
        } else if (x + 1 == W) {  // the last one to three samples of a row
            for (uint32_t j = 0; j <= xs; j++) row[(x - xs) + j] = (ST)out4.get(j);
            out4.clear();
        }
      }
      if (rc == FELICS_OK) rc = br.failed() ? FELICS_E_IO : ((RGB ? out_of_range != 0 : out_of_range > 255u) ? FELICS_E_INVALID_VALUE : FELICS_OK);
    }
    }
    if (br.failed() && rc == FELICS_OK) rc = FELICS_E_IO;  // (whatever else: it was decoding padding)
    status[img] = rc;
}

// ycocg_to_rgb (color_transform.rs:20-26) on the decoded planes, range-checked like try_into::<u8>()
__global__ __launch_bounds__(256) void k_ycocg8_to_rgb(const int16_t *__restrict__ planes, uint8_t *__restrict__ pixels,
                                                       uint32_t npix, int *__restrict__ status) {
    const uint32_t img = blockIdx.y;
    if (status[img] != FELICS_OK) return;
    const int16_t *pl = planes + (uint64_t)img * 3 * npix;
    uint8_t *dst = pixels + (uint64_t)img * 3 * npix;
    bool bad = false;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += gridDim.x * blockDim.x) {
        const int yv = pl[i], co = pl[(uint64_t)npix + i], cg = pl[2ull * npix + i];
        const int t = yv - cg / 2;  // `/` truncates toward zero like Rust's
        const int g = cg + t, b = t - co / 2, r = b + co;
        if ((r | g | b) < 0 || r > 255 || g > 255 || b > 255) bad = true;
        dst[(uint64_t)i * 3] = (uint8_t)r;
        dst[(uint64_t)i * 3 + 1] = (uint8_t)g;
        dst[(uint64_t)i * 3 + 2] = (uint8_t)b;
    }
    if (bad) atomicCAS(&status[img], FELICS_OK, FELICS_E_INVALID_VALUE);
}

// ------------------------------------------------------------------------------------------
// 16-bit streams (traits.rs:35-43: fifteen Rice parameters, contexts 0 .. 131 070).
//
// The same walk, one wave per stream, with the estimator table where it fits: in HBM, 131 071 rows of sixteen words per
// stream (fifteen counters + the EPOCH the row was last written in: a row of another epoch reads as zeros, so the table
// is never cleared between planes, streams or calls), behind a direct-mapped write-back cache of DEC16_SLOTS rows in LDS
// (a plane uses a few thousand contexts, a few dozen of them for most of its events).  A row lives one counter per lane
// (lanes 0 .. 14): get_k and the update's minimum are DPP reductions over one row of sixteen lanes, everything else about
// a pixel is scalar code as in k_decode8.  Gray: u16 pixels straight out; RGB: Y / Co / Cg as int32 planes + k_ycocg16_to_rgb.
// ------------------------------------------------------------------------------------------

constexpr uint32_t DEC16_SLOTS = 512;                 // cached rows: 32 KB of LDS
constexpr uint32_t DEC16_ROW = 16;                    // words per row (15 counters, epoch)
constexpr uint32_t DEC16_CONTEXTS = 2u * 65535u + 1u;  // MAX_CONTEXT + 1 (traits.rs:38)

// minimum over lanes 0 .. 15 (one DPP row), left in lane 15, returned wave-uniform
__device__ __forceinline__ uint32_t row16_min(uint32_t v) {
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)v, 0x111, 0xF, 0xF, false));  // row_shr:1
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)v, 0x112, 0xF, 0xF, false));  // row_shr:2
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)v, 0x114, 0xF, 0xF, false));  // row_shr:4
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)v, 0x118, 0xF, 0xF, false));  // row_shr:8
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 15);
}

__global__ __launch_bounds__(64) void k_decode16(const uint8_t *__restrict__ streams, const uint64_t *__restrict__ offsets,
                                                 const uint64_t *__restrict__ lens, uint32_t W, uint32_t H, uint32_t color,
                                                 uint16_t *__restrict__ pixels, int32_t *__restrict__ planes,
                                                 uint32_t *__restrict__ gtable, uint32_t epoch0, int *__restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint32_t *crow = reinterpret_cast<uint32_t *>(smem);                       // [DEC16_SLOTS][DEC16_ROW] cached rows
    uint32_t *ctag = crow + DEC16_SLOTS * DEC16_ROW;                            // [DEC16_SLOTS] context held (or ~0)
    int32_t *rows = reinterpret_cast<int32_t *>(ctag + DEC16_SLOTS);
    const uint32_t rstride = decode8_row_stride(W);
    const uint32_t img = blockIdx.x, lane = lane_id();
    const uint8_t *s = streams + offsets[img];
    const uint64_t slen = lens[img];
    const uint64_t npix = (uint64_t)W * H;
    const uint32_t nplanes = color ? 3u : 1u;
    uint32_t *table = gtable + (uint64_t)img * DEC16_CONTEXTS * DEC16_ROW;
    int rc = FELICS_OK;
    if (slen < FELICS_HEADER_BYTES) {
        rc = FELICS_E_IO;
    } else {
        const uint32_t w = ((uint32_t)s[6] << 24) | ((uint32_t)s[7] << 16) | ((uint32_t)s[8] << 8) | s[9];
        const uint32_t h = ((uint32_t)s[10] << 24) | ((uint32_t)s[11] << 16) | ((uint32_t)s[12] << 8) | s[13];
        if (s[0] != 'F' || s[1] != 'L' || s[2] != 'C' || s[3] != 'S') rc = FELICS_E_INVALID_SIGNATURE;
        else if (s[4] > 1) rc = FELICS_E_INVALID_COLOR_TYPE;
        else if (s[5] > 1) rc = FELICS_E_INVALID_PIXEL_DEPTH;
        else if (s[4] != color || s[5] != 1 || w != W || h != H) rc = FELICS_E_INVALID_DIMENSIONS;
    }
    rc = unii(rc);
    if (rc != FELICS_OK) {
        if (lane == 0) status[img] = rc;
        return;
    }
    ScalarBits br;
    br.init(s + FELICS_HEADER_BYTES, slen - FELICS_HEADER_BYTES);
    for (uint32_t c = 0; c < nplanes && rc == FELICS_OK; c++) {
        const int32_t p0 = (int32_t)br.get(32), p1 = (int32_t)br.get(32);  // compression.rs:166-167
        if (br.failed()) {
            rc = FELICS_E_IO;
            break;
        }
        if (npix == 0) continue;
        const uint32_t epoch = epoch0 + c;  // KEstimator::new: rows of other epochs read as zeros
        for (uint32_t i = lane; i < DEC16_SLOTS; i += 64) ctag[i] = 0xFFFFFFFFu;  // (nothing to write back: the last plane's rows are dead)
        __builtin_amdgcn_wave_barrier();
        int32_t *outp = planes ? planes + ((uint64_t)img * nplanes + c) * npix : nullptr;
        uint16_t *outg = planes ? nullptr : pixels + (uint64_t)img * npix;
        const int lo_ok = (color && c > 0) ? -65535 : 0, hi_ok = 65535;  // Y 0..65535, Co / Cg -65535..65535
        uint32_t x = 0, y = 0;
        int32_t *cur = rows, *prev = rows + rstride;
        int upv = 0;   // VECTOR: prev[xb + lane] for the 64-sample block xb the walk stands in
        int rowv = 0;  // VECTOR: the samples of this block decoded so far
        int left = 0, left2 = 0;
        int first_col2 = 0;
        for (uint64_t i = 0; i < npix; i++) {
            const uint32_t xl = x & 63u;
            if (xl == 0) {
                if (y > 0) upv = (int)prev[x + lane];
                if (x == 0 && y > 0) first_col2 = y >= 2 ? unii((int)cur[0]) : (W > 1 ? __builtin_amdgcn_readlane(upv, 1) : 0);
            }
            int pv;
            if (i < 2) {
                pv = i == 0 ? p0 : p1;
            } else {
                const int above = __builtin_amdgcn_readlane(upv, (int)xl);
                const bool row0 = y == 0, col0 = x == 0 && !row0;
                const int v1 = col0 ? above : left;
                const int v2 = col0 ? first_col2 : (row0 ? left2 : above);
                const int hi = max(v1, v2), lo = min(v1, v2);
                const uint32_t ctx = (uint32_t)(hi - lo);  // <= 131 070 because every stored sample is in range
                br.refill();  // >= 33 bits: an in-range code has at most 18, the two flags of the other kind 2
                if (br.take(1)) {  // in range: phased-in code of p - L (phase_in_coding.rs:86-112)
                    const uint32_t n = ctx + 1;
                    const uint32_t m = 31u - (uint32_t)__builtin_clz(n);
                    const uint32_t right_p = (2u << m) - n, left_p = n - (1u << m);
                    uint32_t r = br.take(m);
                    const uint32_t longer = r >= right_p ? 1u : 0u;
                    const uint32_t r2 = (r - right_p) * 2u + right_p + br.take(longer);
                    r = longer ? r2 : r;
                    uint32_t rot = r + left_p;
                    rot = rot >= n ? rot - n : rot;
                    pv = lo + (int)rot;
                } else {
                    const bool above_flag = br.take(1) != 0;
                    // the context's row: lane k < 15 holds counter k
                    const uint32_t slot = ctx & (DEC16_SLOTS - 1u);
                    const uint32_t held = uni(ctag[slot]);
                    uint32_t S;
                    if (held == ctx) {
                        S = crow[slot * DEC16_ROW + (lane & 15u)];
                    } else {
                        if (held != 0xFFFFFFFFu && lane < DEC16_ROW)  // write the row this slot held back (its epoch word with it)
                            table[(uint64_t)held * DEC16_ROW + lane] = crow[slot * DEC16_ROW + lane];
                        // (read from L2, not from this CU's L1: the row may be one this wave wrote back a while ago; that store
                        // has been acknowledged by now -- every load's wait covers the stores before it)
                        uint32_t g = lane < DEC16_ROW ? __hip_atomic_load(&table[(uint64_t)ctx * DEC16_ROW + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
                        const uint32_t row_epoch = (uint32_t)__builtin_amdgcn_readlane((int)g, 15);
                        g = row_epoch == epoch ? g : 0u;  // a row of another plane / call: fresh
                        S = (uint32_t)__shfl((int)g, (int)(lane & 15u));  // every row of sixteen lanes holds the counters
                        if (lane == 0) ctag[slot] = ctx;
                    }
                    // get_k: smallest counter, ties to the largest k (parameter_selection.rs:71-85)
                    const uint32_t l15 = lane & 15u;
                    const uint32_t key = l15 < 15u ? (S << 4) | (15u - l15) : 0xFFFFFFFFu;
                    const uint32_t k = 15u - (row16_min(key) & 15u);
                    const uint64_t q = br.unary0();
                    const uint64_t e64 = (q << k) + br.get(k);
                    if (e64 > 262144u) {  // no sample of a 16-bit plane is that far from its neighbours
                        rc = e64 > 0xFFFFFFFFull ? FELICS_E_VALUE_OVERFLOW : FELICS_E_INVALID_VALUE;
                        break;
                    }
                    const uint32_t e = (uint32_t)e64;
                    uint32_t S2 = S + (e >> l15) + 1u + l15;                       // update (rice_coding.rs:56-58 lengths)
                    const uint32_t mn = row16_min(l15 < 15u ? S2 : 0xFFFFFFFFu);
                    S2 = mn > 1024u ? S2 >> 1 : S2;                                // x /= 2 on every counter
                    if (lane < DEC16_ROW) crow[slot * DEC16_ROW + lane] = lane < 15u ? S2 : epoch;
                    pv = above_flag ? hi + (int)e + 1 : lo - (int)e - 1;
                }
            }
            if (pv < lo_ok || pv > hi_ok) {
                rc = FELICS_E_INVALID_VALUE;
                break;
            }
            rowv = lane == xl ? pv : rowv;
            left2 = left;
            left = pv;
            const bool row_end = x + 1 == W;
            if (xl == 63u || row_end) {
                const uint32_t xb = x & ~63u;
                if (xb + lane <= x) {
                    cur[xb + lane] = rowv;
                    if (outg)
                        outg[(uint64_t)y * W + xb + lane] = (uint16_t)rowv;
                    else
                        outp[(uint64_t)y * W + xb + lane] = rowv;
                }
            }
            if (row_end) {
                if (br.failed()) {
                    rc = FELICS_E_IO;
                    break;
                }
                __builtin_amdgcn_wave_barrier();
                x = 0;
                y++;
                int32_t *t = cur;
                cur = prev;
                prev = t;
            } else {
                x++;
            }
        }
    }
    if (br.failed()) rc = FELICS_E_IO;
    if (lane == 0) status[img] = rc;
}

// ycocg_to_rgb (color_transform.rs:20-26) on the decoded 16-bit planes, range-checked like try_into::<u16>()
__global__ __launch_bounds__(256) void k_ycocg16_to_rgb(const int32_t *__restrict__ planes, uint16_t *__restrict__ pixels,
                                                        uint32_t npix, int *__restrict__ status) {
    const uint32_t img = blockIdx.y;
    if (status[img] != FELICS_OK) return;
    const int32_t *pl = planes + (uint64_t)img * 3 * npix;
    uint16_t *dst = pixels + (uint64_t)img * 3 * npix;
    bool bad = false;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += gridDim.x * blockDim.x) {
        const int yv = pl[i], co = pl[(uint64_t)npix + i], cg = pl[2ull * npix + i];
        const int t = yv - cg / 2;  // `/` truncates toward zero like Rust's
        const int g = cg + t, b = t - co / 2, r = b + co;
        if ((r | g | b) < 0 || r > 65535 || g > 65535 || b > 65535) bad = true;
        dst[(uint64_t)i * 3] = (uint16_t)r;
        dst[(uint64_t)i * 3 + 1] = (uint16_t)g;
        dst[(uint64_t)i * 3 + 2] = (uint16_t)b;
    }
    if (bad) atomicCAS(&status[img], FELICS_OK, FELICS_E_INVALID_VALUE);
}

uint32_t decode16_lds_bytes(uint32_t W) {
    return DEC16_SLOTS * DEC16_ROW * 4 + DEC16_SLOTS * 4 + 2u * decode8_row_stride(W) * 4u;
}
size_t decode16_table_bytes(uint32_t n) { return (size_t)n * DEC16_CONTEXTS * DEC16_ROW * 4; }

hipError_t launch_decode16(hipStream_t s, const uint8_t *streams, const uint64_t *offsets, const uint64_t *lens, uint32_t n,
                           uint32_t W, uint32_t H, uint32_t color, uint16_t *pixels, int32_t *planes, uint32_t *table,
                           uint32_t epoch0, int *status) {
    if (n == 0) return hipSuccess;
    const uint32_t lds = decode16_lds_bytes(W);
    if (lds > 64u * 1024u) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_decode16),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)DECODE_LDS_LIMIT);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_decode16, dim3(n), dim3(64), lds, s, streams, offsets, lens, W, H, color, pixels, planes, table, epoch0,
                       status);
    if (color) {
        const uint64_t npix = (uint64_t)W * H;
        const uint32_t bx = (uint32_t)std::min<uint64_t>((npix + 255) / 256, 1024u);
        if (bx) hipLaunchKernelGGL(k_ycocg16_to_rgb, dim3(bx, n), dim3(256), 0, s, planes, pixels, (uint32_t)npix, status);
    }
    return hipGetLastError();
}

size_t decode8_lanes_table_bytes(uint32_t n, uint32_t color) { return (size_t)n * (color ? 3u * DEC8L_TABLE_DW_RGB : DEC8L_TABLE_DW) * 4; }

// lane = stream (gray8 or RGB8, W >= 8); `table` = decode8_lanes_table_bytes(n, color) bytes, all zero; RGB: `planes` takes the int16
// planes (n * 3 * W * H), `pixels` the converted frames
hipError_t launch_decode8_lanes(hipStream_t s, const uint8_t *streams, const uint64_t *offsets, const uint64_t *lens, uint32_t n,
                                uint32_t W, uint32_t H, uint32_t color, uint8_t *pixels, int16_t *planes, uint32_t *table, int *status) {
    if (n == 0) return hipSuccess;
    if (!color) {
        hipLaunchKernelGGL(k_decode8_lanes<false>, dim3((n + 63) / 64), dim3(64), 0, s, streams, offsets, lens, n, W, H, (void *)pixels, table, status);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_decode8_lanes<true>, dim3((n + 63) / 64), dim3(64), 0, s, streams, offsets, lens, n, W, H, (void *)planes, table, status);
    const uint64_t npix = (uint64_t)W * H;
    const uint32_t bx = (uint32_t)std::min<uint64_t>((npix + 255) / 256, 1024u);
    if (bx) hipLaunchKernelGGL(k_ycocg8_to_rgb, dim3(bx, n), dim3(256), 0, s, planes, pixels, (uint32_t)npix, status);
    return hipGetLastError();
}

uint32_t decode8_lds_bytes(uint32_t W, uint32_t color) {
    return (color ? nctx_of<int16_t>() : nctx_of<uint8_t>()) * 6 * 4 + 2u * decode8_row_stride(W) * 2u;
}

hipError_t launch_decode8(hipStream_t s, const uint8_t *streams, const uint64_t *offsets, const uint64_t *lens, uint32_t n,
                          uint32_t W, uint32_t H, uint32_t color, uint8_t *pixels, int16_t *planes, int *status) {
    if (n == 0) return hipSuccess;
    const uint32_t lds = decode8_lds_bytes(W, color);
    if (lds > 64u * 1024u) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_decode8),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)DECODE_LDS_LIMIT);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_decode8, dim3(n), dim3(64), lds, s, streams, offsets, lens, W, H, color, pixels, planes, status);
    if (color) {
        const uint64_t npix = (uint64_t)W * H;
        const uint32_t bx = (uint32_t)std::min<uint64_t>((npix + 255) / 256, 1024u);
        if (bx) hipLaunchKernelGGL(k_ycocg8_to_rgb, dim3(bx, n), dim3(256), 0, s, planes, pixels, (uint32_t)npix, status);
    }
    return hipGetLastError();
}

}  // namespace felics
