/*
 * felics_oracle.c -- CPU restatement of the reference FELICS codec (plain C99).
 *
 * TEST INFRASTRUCTURE ONLY (see felics_oracle.h).  The product path never calls
 * into this file.
 *
 * Each function cites the reference file:line it restates (paths relative to
 * /root/reference/).  The per-pixel loop keeps the reference's structure: one
 * iteration per pixel in raster order, one estimator query per pixel, bits
 * pushed through a big-endian bit writer.
 *
 * Third-party behaviour restated here (not present under /root/reference):
 *   bitstream-io 2.4.2 (Cargo.lock:264-265) BitWriter/BitReader<_, BigEndian>:
 *     - bits fill every byte most-significant bit first;
 *     - write(n, v) emits the low n bits of v, most significant of them first,
 *       n == 0 emits nothing;
 *     - write_unary0(q) emits q one-bits and then a zero-bit;
 *     - write_signed(32, v) emits the 32-bit two's complement of v, MSB first,
 *       at whatever bit position the stream is in;
 *     - byte_align() pads the pending byte with zero bits.
 *   byteorder 1.5.0: u32 big-endian in the header (format.rs:58-59).
 */
#include "felics_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* bit sink / source (bitstream-io BigEndian semantics, see header)    */
/* ------------------------------------------------------------------ */

typedef struct {
    uint8_t *buf;   /* NULL => text mode */
    size_t cap;
    size_t pos;     /* bytes completed */
    uint64_t acc;   /* pending bits, right-aligned */
    unsigned nacc;  /* number of pending bits (<8 after every call) */
    int overflow;
    uint64_t total_bits;
    /* text mode (KAT helpers) */
    char *text;
    size_t text_cap;
    size_t text_len;
    int mock_order;
} bitsink;

static void sink_init(bitsink *s, uint8_t *buf, size_t cap) {
    memset(s, 0, sizeof(*s));
    s->buf = buf;
    s->cap = cap;
}

static void sink_push_byte(bitsink *s, uint8_t b) {
    if (s->pos < s->cap)
        s->buf[s->pos] = b;
    else
        s->overflow = 1;
    s->pos++;
}

static void sink_text_bit(bitsink *s, int bit) {
    if (s->text_len + 1 < s->text_cap) s->text[s->text_len] = bit ? '1' : '0';
    s->text_len++;
}

/* BitWrite::write_bit */
static void sink_bit(bitsink *s, int bit) {
    s->total_bits++;
    if (s->text) {
        sink_text_bit(s, bit);
        return;
    }
    s->acc = (s->acc << 1) | (uint64_t)(bit & 1);
    if (++s->nacc == 8) {
        sink_push_byte(s, (uint8_t)s->acc);
        s->acc = 0;
        s->nacc = 0;
    }
}

/* BitWrite::write(bits, value), bits <= 32 */
static void sink_write(bitsink *s, unsigned bits, uint32_t value) {
    if (bits == 0) return;
    if (s->text) {
        s->total_bits += bits;
        if (s->mock_order) { /* bitwrite_mock.rs:30-41: value % 2 first */
            for (unsigned i = 0; i < bits; i++) sink_text_bit(s, (value >> i) & 1);
        } else {
            for (unsigned i = bits; i-- > 0;) sink_text_bit(s, (value >> i) & 1);
        }
        return;
    }
    s->total_bits += bits;
    if (bits < 32) value &= (1u << bits) - 1u;
    s->acc = (s->acc << bits) | value;
    s->nacc += bits;
    while (s->nacc >= 8) {
        s->nacc -= 8;
        sink_push_byte(s, (uint8_t)(s->acc >> s->nacc));
    }
    s->acc &= (1ull << s->nacc) - 1ull;
}

/* BitWrite::write_unary0 */
static void sink_unary0(bitsink *s, uint32_t q) {
    while (q >= 32) {
        sink_write(s, 32, 0xFFFFFFFFu);
        q -= 32;
    }
    if (q) sink_write(s, q, (1u << q) - 1u);
    sink_bit(s, 0);
}

/* BitWrite::write_signed(32, v) */
static void sink_signed32(bitsink *s, int32_t v) { sink_write(s, 32, (uint32_t)v); }

/* BitWrite::byte_align */
static void sink_align(bitsink *s) {
    while (s->nacc != 0) sink_bit(s, 0);
}

typedef struct {
    const uint8_t *buf;
    size_t len;
    size_t pos;     /* next byte */
    uint32_t cur;   /* current byte */
    unsigned left;  /* unread bits in cur */
    int eof;
} bitsrc;

static void src_init(bitsrc *r, const uint8_t *buf, size_t len) {
    r->buf = buf;
    r->len = len;
    r->pos = 0;
    r->cur = 0;
    r->left = 0;
    r->eof = 0;
}

static int src_bit(bitsrc *r) {
    if (r->left == 0) {
        if (r->pos >= r->len) {
            r->eof = 1;
            return 0;
        }
        r->cur = r->buf[r->pos++];
        r->left = 8;
    }
    r->left--;
    return (int)((r->cur >> r->left) & 1u);
}

static uint32_t src_read(bitsrc *r, unsigned bits) {
    uint32_t v = 0;
    for (unsigned i = 0; i < bits; i++) v = (v << 1) | (uint32_t)src_bit(r);
    return v;
}

/* BitRead::read_unary0: count ones up to the first zero */
static uint32_t src_unary0(bitsrc *r) {
    uint32_t q = 0;
    while (!r->eof && src_bit(r)) q++;
    return q;
}

/* ------------------------------------------------------------------ */
/* Rice code (src/coding/rice_coding.rs:19-58)                          */
/* ------------------------------------------------------------------ */

/* rice_coding.rs:26-38 */
static void rice_encode(bitsink *s, unsigned k, uint32_t number) {
    uint32_t quotient = number >> k;
    uint32_t remainder = number & ((1u << k) - 1u);
    sink_unary0(s, quotient);
    sink_write(s, k, remainder);
}

/* rice_coding.rs:42-51; the reference unwraps a checked_mul (panic), here an error */
static int rice_decode(bitsrc *r, unsigned k, uint32_t *out) {
    uint32_t q = src_unary0(r);
    uint32_t rem = src_read(r, k);
    if (r->eof) return FO_E_IO;
    uint64_t v = ((uint64_t)q << k) + rem;
    if (((uint64_t)q << k) > 0xFFFFFFFFull || v > 0xFFFFFFFFull) return FO_E_VALUE_OVERFLOW;
    *out = (uint32_t)v;
    return FO_OK;
}

/* rice_coding.rs:56-58 */
uint32_t fo_rice_code_length(unsigned k, uint32_t v) { return (v >> k) + 1u + k; }

/* ------------------------------------------------------------------ */
/* Phased-in code (src/coding/phase_in_coding.rs:23-112)                */
/* ------------------------------------------------------------------ */

typedef struct {
    uint32_t n, m, left_p, right_p;
} phasein;

static unsigned ilog2_u32(uint32_t v) {
    unsigned m = 0;
    while (v >>= 1) m++;
    return m;
}

/* phase_in_coding.rs:23-36; n == 0 and n >= 2^31 panic in the reference */
static int phasein_new(phasein *c, uint32_t n) {
    if (n == 0 || n >= 0x80000000u) return -1;
    c->n = n;
    c->m = ilog2_u32(n);
    c->left_p = n - (1u << c->m);
    c->right_p = (1u << (c->m + 1)) - n;
    return 0;
}

/* phase_in_coding.rs:59-84 */
static void phasein_encode(bitsink *s, const phasein *c, uint32_t number) {
    uint32_t rot = (number + c->n - c->left_p) % c->n; /* rotate_right, :45-47 */
    if (rot < c->right_p) {
        sink_write(s, c->m, rot);
    } else {
        uint32_t pair = (rot - c->right_p) / 2;
        uint32_t last = (rot - c->right_p) % 2;
        sink_write(s, c->m, pair + c->right_p);
        sink_bit(s, (int)last);
    }
}

/* phase_in_coding.rs:90-112 */
static uint32_t phasein_decode(bitsrc *r, const phasein *c) {
    uint32_t first = src_read(r, c->m);
    uint32_t number;
    if (first < c->right_p) {
        number = first;
    } else {
        number = (first - c->right_p) * 2 + c->right_p;
        if (src_bit(r)) number += 1;
    }
    /* rotate_left (:50-52); 64-bit so a corrupt stream cannot wrap */
    return (uint32_t)(((uint64_t)number + c->left_p) % c->n);
}

/* ------------------------------------------------------------------ */
/* Neighbour rule (src/compression/misc.rs:6-24)                        */
/* ------------------------------------------------------------------ */

int fo_nearest_neighbours(size_t i, size_t width, size_t *a, size_t *b) {
    size_t x = i % width, y = i / width;
    if (x > 0 && y > 0) {
        *a = i - 1;
        *b = i - width;
        return 1;
    } else if (y == 0) {
        if (x >= 2) {
            *a = i - 1;
            *b = i - 2;
            return 1;
        }
        return 0;
    } else if (y >= 2) {
        *a = i - width;
        *b = i - 2 * width;
        return 1;
    } else if (x + 1 < width) {
        *a = i - width;
        *b = i - width + 1;
        return 1;
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* Rice parameter estimator (src/compression/parameter_selection.rs)    */
/* ------------------------------------------------------------------ */

struct fo_kest {
    uint32_t max_context;
    uint8_t k_values[32];
    size_t nk;
    int64_t halve_at; /* <0 = None */
    uint32_t *map;    /* (max_context+1) x nk */
};

/* parameter_selection.rs:24-41 */
fo_kest *fo_kest_new(uint32_t max_context, const uint8_t *k_values, size_t nk, int64_t halve_at) {
    if (nk == 0 || nk > 32) return NULL; /* empty list panics in the reference */
    fo_kest *e = (fo_kest *)malloc(sizeof(*e));
    if (!e) return NULL;
    e->max_context = max_context;
    memcpy(e->k_values, k_values, nk);
    e->nk = nk;
    e->halve_at = halve_at;
    e->map = (uint32_t *)calloc(((size_t)max_context + 1) * nk, sizeof(uint32_t));
    if (!e->map) {
        free(e);
        return NULL;
    }
    return e;
}

void fo_kest_free(fo_kest *e) {
    if (e) {
        free(e->map);
        free(e);
    }
}

/* parameter_selection.rs:49-64 */
void fo_kest_update(fo_kest *e, uint32_t context, uint32_t encoded) {
    uint32_t *row = e->map + (size_t)context * e->nk;
    for (size_t i = 0; i < e->nk; i++) row[i] += fo_rice_code_length(e->k_values[i], encoded);
    if (e->halve_at >= 0) {
        uint32_t mn = row[0];
        for (size_t i = 1; i < e->nk; i++)
            if (row[i] < mn) mn = row[i];
        if ((int64_t)mn > e->halve_at)
            for (size_t i = 0; i < e->nk; i++) row[i] /= 2;
    }
}

/* parameter_selection.rs:71-85 -- `<=` makes ties go to the LAST index */
unsigned fo_kest_get_k(const fo_kest *e, uint32_t context) {
    const uint32_t *row = e->map + (size_t)context * e->nk;
    uint32_t smallest = 0xFFFFFFFFu;
    size_t best = 0;
    for (size_t i = 0; i < e->nk; i++) {
        if (row[i] <= smallest) {
            best = i;
            smallest = row[i];
        }
    }
    return e->k_values[best];
}

void fo_kest_row(const fo_kest *e, uint32_t context, uint32_t *row_out) {
    memcpy(row_out, e->map + (size_t)context * e->nk, e->nk * sizeof(uint32_t));
}

/* ------------------------------------------------------------------ */
/* Colour transform (src/compression/color_transform.rs:11-26)          */
/* C `/` on ints truncates toward zero exactly like Rust's.             */
/* ------------------------------------------------------------------ */

void fo_rgb_to_ycocg(int32_t r, int32_t g, int32_t b, int32_t *y, int32_t *co, int32_t *cg) {
    int32_t c_o = r - b;
    int32_t t = b + c_o / 2;
    int32_t c_g = g - t;
    *y = t + c_g / 2;
    *co = c_o;
    *cg = c_g;
}

void fo_ycocg_to_rgb(int32_t y, int32_t co, int32_t cg, int32_t *r, int32_t *g, int32_t *b) {
    int32_t t = y - cg / 2;
    *g = cg + t;
    *b = t - co / 2;
    *r = *b + co;
}

/* ------------------------------------------------------------------ */
/* Intensity constants (src/compression/traits.rs:25-43)                */
/* ------------------------------------------------------------------ */

typedef struct {
    uint32_t max_context;
    uint8_t k_values[16];
    size_t nk;
    int64_t halve_at;
} coding_options;

static coding_options options_for_depth(int depth) {
    coding_options o;
    memset(&o, 0, sizeof(o));
    if (depth == 0) { /* u8: traits.rs:25-33 */
        o.nk = 6;
        o.max_context = 255u * 2u;
    } else { /* u16: traits.rs:35-43 */
        o.nk = 15;
        o.max_context = 65535u * 2u;
    }
    for (size_t i = 0; i < o.nk; i++) o.k_values[i] = (uint8_t)i;
    o.halve_at = 1024;
    return o;
}

/* ------------------------------------------------------------------ */
/* Channel codec (src/compression.rs:29-45, 76-148, 151-248)            */
/* ------------------------------------------------------------------ */

typedef struct {
    uint8_t *cls;
    uint32_t *ctx;
    uint8_t *k;
    uint32_t *val;
    uint32_t *nbits;
} trace_out;

enum { CLS_IN = 0, CLS_BELOW = 1, CLS_ABOVE = 2, CLS_RAW = 3 };

/* compression.rs:29-45 */
static void encode_intensity(bitsink *s, int cls) {
    if (cls == CLS_IN) {
        sink_bit(s, 1);
    } else if (cls == CLS_ABOVE) {
        sink_bit(s, 0);
        sink_bit(s, 1);
    } else {
        sink_bit(s, 0);
        sink_bit(s, 0);
    }
}

/* compression.rs:76-148 */
static int compress_channel(const int32_t *channel, uint32_t width, uint32_t height,
                            const coding_options *opt, bitsink *s, trace_out *tr) {
    uint64_t total64 = (uint64_t)width * height;
    if (total64 > 0xFFFFFFFFull) return FO_E_INVALID_DIMENSIONS; /* checked_mul().unwrap() panics */
    size_t total = (size_t)total64;

    if (width == 0 || height == 0) { /* :94-98 */
        sink_signed32(s, 0);
        sink_signed32(s, 0);
        return FO_OK;
    }
    if (width == 1 && height == 1) { /* :99-103 */
        sink_signed32(s, channel[0]);
        sink_signed32(s, 0);
        if (tr) { tr->cls[0] = CLS_RAW; tr->ctx[0] = 0; tr->k[0] = 0; tr->val[0] = (uint32_t)channel[0]; tr->nbits[0] = 64; }
        return FO_OK;
    }
    sink_signed32(s, channel[0]); /* :105-106 */
    sink_signed32(s, channel[1]);
    if (tr) {
        for (int j = 0; j < 2; j++) {
            tr->cls[j] = CLS_RAW; tr->ctx[j] = 0; tr->k[j] = 0;
            tr->val[j] = (uint32_t)channel[j]; tr->nbits[j] = 32;
        }
    }

    fo_kest *est = fo_kest_new(opt->max_context, opt->k_values, opt->nk, opt->halve_at); /* :110 */
    if (!est) return FO_E_IO;

    for (size_t i = 2; i < total; i++) { /* :117-146 */
        size_t a = 0, b = 0;
        fo_nearest_neighbours(i, width, &a, &b);
        int32_t p = channel[i], v1 = channel[a], v2 = channel[b];
        int32_t h = v1 > v2 ? v1 : v2;
        int32_t l = v1 < v2 ? v1 : v2;
        uint32_t context = (uint32_t)(h - l);
        unsigned k = fo_kest_get_k(est, context);
        uint64_t before = s->total_bits;
        int cls;
        uint32_t to_encode;
        if (p >= l && p <= h) { /* :130-134 */
            cls = CLS_IN;
            encode_intensity(s, cls);
            to_encode = (uint32_t)(p - l);
            phasein c = {1, 0, 0, 1};
            phasein_new(&c, context + 1);
            phasein_encode(s, &c, to_encode);
        } else if (p < l) { /* :135-139 */
            cls = CLS_BELOW;
            encode_intensity(s, cls);
            to_encode = (uint32_t)(l - p - 1);
            rice_encode(s, k, to_encode);
            fo_kest_update(est, context, to_encode);
        } else { /* :140-145 */
            cls = CLS_ABOVE;
            encode_intensity(s, cls);
            to_encode = (uint32_t)(p - h - 1);
            rice_encode(s, k, to_encode);
            fo_kest_update(est, context, to_encode);
        }
        if (tr) {
            tr->cls[i] = (uint8_t)cls;
            tr->ctx[i] = context;
            tr->k[i] = (uint8_t)k;
            tr->val[i] = to_encode;
            tr->nbits[i] = (uint32_t)(s->total_bits - before);
        }
    }
    fo_kest_free(est);
    return FO_OK;
}

/* compression.rs:151-248. The reference panics on a context above max_context
 * (parameter_selection.rs:72) or an overflowing h-l; both are reported as
 * InvalidValue here so a corrupt stream cannot abort the test process. */
static int decompress_channel(uint32_t width, uint32_t height, const coding_options *opt,
                              bitsrc *r, int32_t **out, size_t *out_len) {
    int32_t p1 = (int32_t)src_read(r, 32);
    int32_t p2 = (int32_t)src_read(r, 32);
    if (r->eof) return FO_E_IO;
    *out = NULL;
    *out_len = 0;
    if (width == 0 || height == 0) return FO_OK;
    if (width == 1 && height == 1) {
        *out = (int32_t *)malloc(sizeof(int32_t));
        if (!*out) return FO_E_IO;
        (*out)[0] = p1;
        *out_len = 1;
        return FO_OK;
    }
    uint64_t total64 = (uint64_t)width * height;
    if (total64 > 0xFFFFFFFFull) return FO_E_INVALID_DIMENSIONS;
    size_t total = (size_t)total64;
    int32_t *buf = (int32_t *)calloc(total, sizeof(int32_t));
    if (!buf) return FO_E_INVALID_DIMENSIONS;
    buf[0] = p1;
    buf[1] = p2;
    fo_kest *est = fo_kest_new(opt->max_context, opt->k_values, opt->nk, opt->halve_at);
    if (!est) { free(buf); return FO_E_IO; }
    int rc = FO_OK;
    for (size_t i = 2; i < total; i++) {
        size_t a = 0, b = 0;
        fo_nearest_neighbours(i, width, &a, &b);
        int32_t v1 = buf[a], v2 = buf[b];
        int32_t h = v1 > v2 ? v1 : v2;
        int32_t l = v1 < v2 ? v1 : v2;
        int64_t ctx64 = (int64_t)h - (int64_t)l;
        if (ctx64 > (int64_t)opt->max_context) { rc = FO_E_INVALID_VALUE; break; }
        uint32_t context = (uint32_t)ctx64;
        unsigned k = fo_kest_get_k(est, context);
        int in_range = src_bit(r); /* decode_intensity, :48-61 */
        int64_t pv;
        if (in_range) {
            phasein c = {1, 0, 0, 1};
            phasein_new(&c, context + 1);
            uint32_t d = phasein_decode(r, &c);
            pv = (int64_t)d + l;
        } else {
            int above = src_bit(r);
            uint32_t enc = 0;
            rc = rice_decode(r, k, &enc);
            if (rc != FO_OK) break;
            fo_kest_update(est, context, enc);
            if (enc > 0x7FFFFFFFu) { rc = FO_E_INVALID_VALUE; break; }
            pv = above ? (int64_t)enc + h + 1 : (int64_t)l - enc - 1;
        }
        if (r->eof) { rc = FO_E_IO; break; }
        if (pv > 0x7FFFFFFFll || pv < -0x80000000ll) { rc = FO_E_VALUE_OVERFLOW; break; }
        buf[i] = (int32_t)pv;
    }
    fo_kest_free(est);
    if (rc != FO_OK) {
        free(buf);
        return rc;
    }
    *out = buf;
    *out_len = total;
    return FO_OK;
}

/* ------------------------------------------------------------------ */
/* Header (src/compression/format.rs:51-84)                             */
/* ------------------------------------------------------------------ */

static void put_be32(uint8_t *p, uint32_t v) {
    p[0] = (uint8_t)(v >> 24);
    p[1] = (uint8_t)(v >> 16);
    p[2] = (uint8_t)(v >> 8);
    p[3] = (uint8_t)v;
}

static uint32_t get_be32(const uint8_t *p) {
    return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
}

int fo_read_header(const uint8_t *in, size_t len, fo_header *hdr) {
    if (len < 4) return FO_E_IO;
    if (memcmp(in, "FLCS", 4) != 0) return FO_E_INVALID_SIGNATURE;
    if (len < 5) return FO_E_IO;
    if (in[4] > 1) return FO_E_INVALID_COLOR_TYPE;
    if (len < 6) return FO_E_IO;
    if (in[5] > 1) return FO_E_INVALID_PIXEL_DEPTH;
    if (len < 14) return FO_E_IO;
    hdr->color_type = in[4];
    hdr->pixel_depth = in[5];
    hdr->width = get_be32(in + 6);
    hdr->height = get_be32(in + 10);
    return FO_OK;
}

/* ------------------------------------------------------------------ */
/* Image level (src/compression.rs:255-282, 322-371, 284-314, 373-441)  */
/* ------------------------------------------------------------------ */

size_t fo_max_compressed_size(uint32_t w, uint32_t h, int color, int depth) {
    /* worst code: 2 flag bits + unary(emax) + terminator at k = 0 */
    uint64_t planes = color ? 3 : 1;
    uint64_t emax = depth == 0 ? (color ? 509u : 254u) : (color ? 131069u : 65534u);
    uint64_t px = (uint64_t)w * h;
    uint64_t bits = planes * 64u + planes * px * (3u + emax);
    return (size_t)(14u + (bits + 7u) / 8u);
}

static int32_t load_sample(const void *pixels, size_t idx, int depth) {
    return depth == 0 ? (int32_t)((const uint8_t *)pixels)[idx] : (int32_t)((const uint16_t *)pixels)[idx];
}

int fo_compress(const void *pixels, uint32_t w, uint32_t h, int color, int depth, uint8_t *out,
                size_t cap, size_t *out_len) {
    if (color != 0 && color != 1) return FO_E_INVALID_COLOR_TYPE;
    if (depth != 0 && depth != 1) return FO_E_INVALID_PIXEL_DEPTH;
    if (cap < 14) return FO_E_BUFFER_TOO_SMALL;
    uint64_t np64 = (uint64_t)w * h;
    if (np64 > 0xFFFFFFFFull) return FO_E_INVALID_DIMENSIONS;
    size_t np = (size_t)np64;

    /* write_header, format.rs:51-61 */
    memcpy(out, "FLCS", 4);
    out[4] = (uint8_t)color;
    out[5] = (uint8_t)depth;
    put_be32(out + 6, w);
    put_be32(out + 10, h);

    bitsink s;
    sink_init(&s, out + 14, cap - 14);
    coding_options opt = options_for_depth(depth);
    int rc = FO_OK;

    if (color == 0) { /* compression.rs:276-280 */
        int32_t *channel = (int32_t *)malloc((np ? np : 1) * sizeof(int32_t));
        if (!channel) return FO_E_IO;
        for (size_t i = 0; i < np; i++) channel[i] = load_sample(pixels, i, depth);
        rc = compress_channel(channel, w, h, &opt, &s, NULL);
        free(channel);
    } else { /* compression.rs:337-369 */
        int32_t *y = (int32_t *)malloc((np ? np : 1) * 3 * sizeof(int32_t));
        if (!y) return FO_E_IO;
        int32_t *co = y + np, *cg = y + 2 * np;
        for (size_t i = 0; i < np; i++) {
            fo_rgb_to_ycocg(load_sample(pixels, 3 * i, depth), load_sample(pixels, 3 * i + 1, depth),
                            load_sample(pixels, 3 * i + 2, depth), &y[i], &co[i], &cg[i]);
        }
        rc = compress_channel(y, w, h, &opt, &s, NULL);
        if (rc == FO_OK) rc = compress_channel(co, w, h, &opt, &s, NULL);
        if (rc == FO_OK) rc = compress_channel(cg, w, h, &opt, &s, NULL);
        free(y);
    }
    if (rc != FO_OK) return rc;
    sink_align(&s);
    if (out_len) *out_len = 14 + s.pos;
    return s.overflow ? FO_E_BUFFER_TOO_SMALL : FO_OK;
}

int fo_decompress(const uint8_t *in, size_t len, void *pixels, size_t pixels_cap, fo_header *hdr_out) {
    fo_header hdr;
    int rc = fo_read_header(in, len, &hdr);
    if (rc != FO_OK) return rc;
    if (hdr_out) *hdr_out = hdr;
    coding_options opt = options_for_depth(hdr.pixel_depth);
    bitsrc r;
    src_init(&r, in + 14, len - 14);
    size_t planes = hdr.color_type ? 3 : 1;
    size_t bps = hdr.pixel_depth ? 2 : 1;
    int32_t *ch[3] = {NULL, NULL, NULL};
    size_t n = 0;
    for (size_t c = 0; c < planes; c++) {
        rc = decompress_channel(hdr.width, hdr.height, &opt, &r, &ch[c], &n);
        if (rc != FO_OK) goto done;
    }
    if (n * planes * bps > pixels_cap) {
        rc = FO_E_BUFFER_TOO_SMALL;
        goto done;
    }
    int32_t maxv = hdr.pixel_depth ? 65535 : 255;
    for (size_t i = 0; i < n; i++) {
        int32_t v[3];
        if (planes == 1) {
            v[0] = ch[0][i];
        } else {
            fo_ycocg_to_rgb(ch[0][i], ch[1][i], ch[2][i], &v[0], &v[1], &v[2]);
        }
        for (size_t c = 0; c < planes; c++) {
            if (v[c] < 0 || v[c] > maxv) { /* try_into fails -> InvalidValue */
                rc = FO_E_INVALID_VALUE;
                goto done;
            }
            if (bps == 1)
                ((uint8_t *)pixels)[i * planes + c] = (uint8_t)v[c];
            else
                ((uint16_t *)pixels)[i * planes + c] = (uint16_t)v[c];
        }
    }
done:
    for (size_t c = 0; c < 3; c++) free(ch[c]);
    return rc;
}

int fo_trace_channel(const int32_t *channel, uint32_t w, uint32_t h, int depth, uint8_t *cls,
                     uint32_t *ctx, uint8_t *k, uint32_t *val, uint32_t *nbits) {
    coding_options opt = options_for_depth(depth);
    bitsink s;
    sink_init(&s, NULL, 0); /* counting only: cap 0 => no byte is ever stored */
    trace_out tr = {cls, ctx, k, val, nbits};
    return compress_channel(channel, w, h, &opt, &s, &tr);
}

/* ------------------------------------------------------------------ */
/* KAT helpers                                                          */
/* ------------------------------------------------------------------ */

static void text_sink(bitsink *s, char *out, size_t cap, int mock) {
    memset(s, 0, sizeof(*s));
    s->text = out;
    s->text_cap = cap;
    s->mock_order = mock;
}

int fo_rice_encode_text(unsigned k, uint32_t v, int mock_order, char *out, size_t cap) {
    if (k > 31) return -1; /* RiceCoder::new(32) panics, rice_coding.rs:20 */
    bitsink s;
    text_sink(&s, out, cap, mock_order);
    rice_encode(&s, k, v);
    if (s.text_len >= cap) return -2;
    out[s.text_len] = 0;
    return (int)s.text_len;
}

int fo_phasein_params(uint32_t n, uint32_t *m, uint32_t *left_p, uint32_t *right_p) {
    phasein c;
    if (phasein_new(&c, n) != 0) return -1;
    *m = c.m;
    *left_p = c.left_p;
    *right_p = c.right_p;
    return 0;
}

int fo_phasein_encode_text(uint32_t n, uint32_t v, int mock_order, char *out, size_t cap) {
    phasein c;
    if (phasein_new(&c, n) != 0) return -1;
    if (v >= n) return -1; /* assert!(number < self.n), phase_in_coding.rs:63 */
    bitsink s;
    text_sink(&s, out, cap, mock_order);
    phasein_encode(&s, &c, v);
    if (s.text_len >= cap) return -2;
    out[s.text_len] = 0;
    return (int)s.text_len;
}

int fo_rice_roundtrip(unsigned k, const uint32_t *vals, size_t n) {
    size_t bits = 0;
    for (size_t i = 0; i < n; i++) bits += fo_rice_code_length(k, vals[i]);
    size_t cap = bits / 8 + 8;
    uint8_t *buf = (uint8_t *)malloc(cap);
    if (!buf) return -1;
    bitsink s;
    sink_init(&s, buf, cap);
    for (size_t i = 0; i < n; i++) rice_encode(&s, k, vals[i]);
    sink_align(&s);
    int ok = !s.overflow && s.total_bits == ((bits + 7) / 8) * 8;
    bitsrc r;
    src_init(&r, buf, s.pos);
    for (size_t i = 0; ok && i < n; i++) {
        uint32_t v = 0;
        if (rice_decode(&r, k, &v) != FO_OK || v != vals[i]) ok = 0;
    }
    free(buf);
    return ok ? 0 : -1;
}

int fo_phasein_roundtrip(uint32_t domain, const uint32_t *vals, size_t n) {
    phasein c;
    if (phasein_new(&c, domain) != 0) return -1;
    size_t cap = n * 5 + 8;
    uint8_t *buf = (uint8_t *)malloc(cap);
    if (!buf) return -1;
    bitsink s;
    sink_init(&s, buf, cap);
    for (size_t i = 0; i < n; i++) phasein_encode(&s, &c, vals[i]);
    sink_align(&s);
    int ok = !s.overflow;
    bitsrc r;
    src_init(&r, buf, s.pos);
    for (size_t i = 0; ok && i < n; i++)
        if (phasein_decode(&r, &c) != vals[i] || r.eof) ok = 0;
    free(buf);
    return ok ? 0 : -1;
}
