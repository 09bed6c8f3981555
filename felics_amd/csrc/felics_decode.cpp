// felics_decode.cpp -- host decoder of libfelics (felics_read_header, felics_decompress).
//
// Mirrors decompress_image / decompress_channel (src/compression.rs:151-248, :284-314,
// :373-441) of the reference.  The entropy decoder is bit-serial per plane (each pixel's context
// depends on pixels decoded just before it), so it runs on the host; SURVEY.md §8(f) ranks a GPU
// decoder after the encode path.  Where the reference panics on a corrupt stream (context above
// MAX_CONTEXT, parameter_selection.rs:72; overflowing quotient, rice_coding.rs:49) this returns
// FELICS_E_INVALID_VALUE / FELICS_E_VALUE_OVERFLOW instead.
#include <cstdint>
#include <cstring>
#include <memory>
#include <new>
#include <vector>

#include "../../include/felics.h"

namespace {

// MSB-first bit reader over a byte range (bitstream-io BitReader<_, BigEndian> semantics).
class BitReader {
  public:
    BitReader(const uint8_t *p, size_t n) : p_(p), end_(p + n) {}

    bool failed() const { return failed_; }

    uint32_t bit() {
        if (have_ == 0 && !refill()) return 0;
        have_--;
        return (uint32_t)(window_ >> have_) & 1u;
    }

    // up to 32 bits, most significant first
    uint32_t bits(unsigned n) {
        uint64_t v = 0;
        while (n) {
            if (have_ == 0 && !refill()) return 0;
            const unsigned take = n < have_ ? n : have_;  // <= 32
            have_ -= take;
            v = (v << take) | ((window_ >> have_) & ((1ull << take) - 1ull));
            n -= take;
        }
        return (uint32_t)v;
    }

    // read_unary0: number of one-bits before the first zero-bit
    uint32_t unary0() {
        uint32_t q = 0;
        for (;;) {
            if (have_ == 0 && !refill()) return q;
            // count leading ones of the `have_` unread bits
            uint64_t unread = window_ << (64 - have_);
            unsigned ones = unread == ~0ull ? 64 : (unsigned)__builtin_clzll(~unread);
            if (ones >= have_) {
                q += have_;
                have_ = 0;
                continue;
            }
            q += ones;
            have_ -= ones + 1;
            return q;
        }
    }

  private:
    bool refill() {
        if (p_ == end_) {
            failed_ = true;
            return false;
        }
        window_ = 0;
        have_ = 0;
        while (p_ != end_ && have_ <= 48) {
            window_ = (window_ << 8) | *p_++;
            have_ += 8;
        }
        return true;
    }

    const uint8_t *p_, *end_;
    uint64_t window_ = 0;  // low `have_` bits are unread, MSB of them first
    unsigned have_ = 0;
    bool failed_ = false;
};

struct Options {  // traits.rs:25-43
    uint32_t max_context;
    unsigned nk;  // k in 0..nk-1
};

// KEstimator (parameter_selection.rs:24-85) with a flat table.
class Estimator {
  public:
    Estimator(const Options &o) : nk_(o.nk), table_((size_t)(o.max_context + 1) * o.nk, 0u) {}

    unsigned get_k(uint32_t ctx) const {
        const uint32_t *row = &table_[(size_t)ctx * nk_];
        uint32_t best = row[0];
        unsigned k = 0;
        for (unsigned i = 1; i < nk_; i++)
            if (row[i] <= best) {  // ties: last wins
                best = row[i];
                k = i;
            }
        return k;
    }

    void update(uint32_t ctx, uint32_t v) {
        uint32_t *row = &table_[(size_t)ctx * nk_];
        uint32_t mn = 0xFFFFFFFFu;
        for (unsigned i = 0; i < nk_; i++) {
            row[i] += (v >> i) + 1 + i;
            if (row[i] < mn) mn = row[i];
        }
        if (mn > 1024)
            for (unsigned i = 0; i < nk_; i++) row[i] >>= 1;
    }

  private:
    unsigned nk_;
    std::vector<uint32_t> table_;
};

// decompress_channel (compression.rs:151-248)
int decode_plane(BitReader &br, uint32_t W, uint32_t H, const Options &opt, std::vector<int32_t> &out) {
    const int32_t p0 = (int32_t)br.bits(32);
    const int32_t p1 = (int32_t)br.bits(32);
    if (br.failed()) return FELICS_E_IO;
    out.clear();
    if (W == 0 || H == 0) return FELICS_OK;
    if (W == 1 && H == 1) {
        out.push_back(p0);
        return FELICS_OK;
    }
    const uint64_t total = (uint64_t)W * H;
    if (total > 0xFFFFFFFFull) return FELICS_E_INVALID_DIMENSIONS;
    try {
        out.assign((size_t)total, 0);
    } catch (const std::bad_alloc &) {
        return FELICS_E_INVALID_DIMENSIONS;
    }
    out[0] = p0;
    out[1] = p1;
    Estimator est(opt);
    uint32_t x = 2 % W, y = 2 / W;
    for (size_t i = 2; i < (size_t)total; i++) {
        size_t a, b;  // misc.rs:6-24
        if (x > 0 && y > 0) {
            a = i - 1;
            b = i - W;
        } else if (y == 0) {
            a = i - 1;
            b = i - 2;
        } else if (y >= 2) {
            a = i - W;
            b = i - 2 * (size_t)W;
        } else {
            a = i - W;
            b = i - W + 1;
        }
        const int64_t v1 = out[a], v2 = out[b];
        const int64_t hi = v1 > v2 ? v1 : v2, lo = v1 < v2 ? v1 : v2;
        if (hi - lo > (int64_t)opt.max_context) return FELICS_E_INVALID_VALUE;
        const uint32_t ctx = (uint32_t)(hi - lo);
        int64_t pv;
        if (br.bit()) {  // in range: phased-in code of p - L in [0, ctx]
            const uint32_t n = ctx + 1;
            const unsigned m = 31u - (unsigned)__builtin_clz(n);
            const uint32_t right_p = (2u << m) - n, left_p = n - (1u << m);
            uint32_t r = br.bits(m);
            if (r >= right_p) r = (r - right_p) * 2 + right_p + br.bit();
            pv = lo + (int64_t)(((uint64_t)r + left_p) % n);  // rotate_left, phase_in_coding.rs:50-52
        } else {
            const bool above = br.bit() != 0;
            const unsigned k = est.get_k(ctx);
            const uint64_t q = br.unary0();
            const uint64_t e = (q << k) + br.bits(k);
            if (br.failed()) return FELICS_E_IO;
            if (e > 0xFFFFFFFFull) return FELICS_E_VALUE_OVERFLOW;
            est.update(ctx, (uint32_t)e);
            if (e > 0x7FFFFFFFull) return FELICS_E_INVALID_VALUE;
            pv = above ? hi + (int64_t)e + 1 : lo - (int64_t)e - 1;
        }
        if (br.failed()) return FELICS_E_IO;
        if (pv > INT32_MAX || pv < INT32_MIN) return FELICS_E_VALUE_OVERFLOW;
        out[i] = (int32_t)pv;
        if (++x == W) {
            x = 0;
            y++;
        }
    }
    return FELICS_OK;
}

template <typename S>
int store_pixels(const std::vector<int32_t> (&ch)[3], unsigned planes, S *dst, int32_t maxv) {
    const size_t n = ch[0].size();
    for (size_t i = 0; i < n; i++) {
        int32_t v[3];
        if (planes == 1) {
            v[0] = ch[0][i];
        } else {  // ycocg_to_rgb, color_transform.rs:20-26 (`/` truncates toward zero)
            const int32_t yv = ch[0][i], co = ch[1][i], cg = ch[2][i];
            const int32_t t = yv - cg / 2;
            v[1] = cg + t;
            v[2] = t - co / 2;
            v[0] = v[2] + co;
        }
        for (unsigned c = 0; c < planes; c++) {
            if (v[c] < 0 || v[c] > maxv) return FELICS_E_INVALID_VALUE;  // try_into::<T>() fails
            dst[i * planes + c] = (S)v[c];
        }
    }
    return FELICS_OK;
}

}  // namespace

extern "C" {

int felics_read_header(const uint8_t *in, size_t len, felics_header *hdr) {
    if (!hdr || (!in && len)) return FELICS_E_INVALID_ARGUMENT;
    // format.rs:63-84: each read_exact fails with an IoError at end of input
    if (len < 4) return FELICS_E_IO;
    if (memcmp(in, "FLCS", 4) != 0) return FELICS_E_INVALID_SIGNATURE;
    if (len < 5) return FELICS_E_IO;
    if (in[4] > 1) return FELICS_E_INVALID_COLOR_TYPE;
    if (len < 6) return FELICS_E_IO;
    if (in[5] > 1) return FELICS_E_INVALID_PIXEL_DEPTH;
    if (len < FELICS_HEADER_BYTES) return FELICS_E_IO;
    hdr->color_type = in[4];
    hdr->pixel_depth = in[5];
    hdr->width = ((uint32_t)in[6] << 24) | ((uint32_t)in[7] << 16) | ((uint32_t)in[8] << 8) | in[9];
    hdr->height = ((uint32_t)in[10] << 24) | ((uint32_t)in[11] << 16) | ((uint32_t)in[12] << 8) | in[13];
    return FELICS_OK;
}

// decompress_with_header (traits.rs:53-56; impls compression.rs:284-314, :373-409): `in` is the bit stream that
// follows the 14 header bytes.  Nothing is allocated before the header's claims have been checked against
// the caller's buffer and against the stream itself (a pixel costs at least one bit).
int felics_decompress_with_header(const uint8_t *in, size_t len, const felics_header *hdr_in, void *pixels,
                                  size_t pixels_cap) {
    if (!hdr_in || (!in && len)) return FELICS_E_INVALID_ARGUMENT;
    const felics_header hdr = *hdr_in;
    if (hdr.color_type > 1) return FELICS_E_INVALID_COLOR_TYPE;
    if (hdr.pixel_depth > 1) return FELICS_E_INVALID_PIXEL_DEPTH;
    const unsigned planes = hdr.color_type == FELICS_COLOR_RGB ? 3 : 1;
    const size_t bps = hdr.pixel_depth == FELICS_DEPTH_8 ? 1 : 2;
    const uint64_t npix = (uint64_t)hdr.width * hdr.height;
    if (npix > 0xFFFFFFFFull) return FELICS_E_INVALID_DIMENSIONS;
    if (npix * planes * bps > pixels_cap) return FELICS_E_BUFFER_TOO_SMALL;
    if (npix && !pixels) return FELICS_E_INVALID_ARGUMENT;
    // every plane starts with two 32-bit values; every further pixel takes at least one flag bit
    if (len < 8ull * planes) return FELICS_E_IO;
    if (npix > 2 && (npix - 2) * planes > (uint64_t)(len - 8ull * planes) * 8ull) return FELICS_E_IO;
    const Options opt = hdr.pixel_depth == FELICS_DEPTH_8 ? Options{255u * 2u, 6} : Options{65535u * 2u, 15};
    BitReader br(in, len);
    std::vector<int32_t> ch[3];
    try {
        for (unsigned c = 0; c < planes; c++) {
            const int rc = decode_plane(br, hdr.width, hdr.height, opt, ch[c]);
            if (rc) return rc;
        }
    } catch (const std::bad_alloc &) {
        return FELICS_E_INVALID_DIMENSIONS;
    }
    if (ch[0].empty()) return FELICS_OK;
    return bps == 1 ? store_pixels(ch, planes, (uint8_t *)pixels, 255) : store_pixels(ch, planes, (uint16_t *)pixels, 65535);
}

int felics_decompress(const uint8_t *in, size_t len, void *pixels, size_t pixels_cap, felics_header *hdr_out) {
    felics_header hdr;
    const int rc = felics_read_header(in, len, &hdr);
    if (rc) return rc;
    if (hdr_out) *hdr_out = hdr;
    return felics_decompress_with_header(in + FELICS_HEADER_BYTES, len - FELICS_HEADER_BYTES, &hdr, pixels, pixels_cap);
}

}  // extern "C"
