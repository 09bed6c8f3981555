/*
 * felics_oracle.h -- CPU restatement of the reference FELICS codec.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the timed CPU baseline.
 *
 * Parity status: the reference is Rust and cannot be built in this pipeline
 * (no rustc/cargo, crates not vendored).  The restatement is pinned by every
 * known-answer test the reference's own unit tests hold (tests/test_oracle_kat.py)
 * and by the round-trip property; byte order at the bitstream-io boundary rests
 * on the documented semantics of bitstream-io 2.4.2 BitWriter<_, BigEndian>
 * (Cargo.lock:264-265) because the reference holds no golden .felics file.
 *
 * All file:line citations are relative to /root/reference/.
 */
#ifndef FELICS_ORACLE_H
#define FELICS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* error codes: same numbering as include/felics.h */
#define FO_OK 0
#define FO_E_IO (-1)
#define FO_E_INVALID_VALUE (-2)
#define FO_E_VALUE_OVERFLOW (-3)
#define FO_E_INVALID_DIMENSIONS (-4)
#define FO_E_INVALID_COLOR_TYPE (-5)
#define FO_E_INVALID_PIXEL_DEPTH (-6)
#define FO_E_INVALID_SIGNATURE (-7)
#define FO_E_BUFFER_TOO_SMALL (-8)

typedef struct {
    uint8_t color_type;  /* 0 gray, 1 rgb      (src/compression/format.rs:9-12) */
    uint8_t pixel_depth; /* 0 = 8 bit, 1 = 16  (src/compression/format.rs:28-31) */
    uint32_t width;
    uint32_t height;
} fo_header;

/* ---- whole-image codec (src/compression.rs:250-441) ---- */
size_t fo_max_compressed_size(uint32_t w, uint32_t h, int color, int depth);
int fo_compress(const void *pixels, uint32_t w, uint32_t h, int color, int depth,
                uint8_t *out, size_t cap, size_t *out_len);
int fo_read_header(const uint8_t *in, size_t len, fo_header *hdr);
int fo_decompress(const uint8_t *in, size_t len, void *pixels, size_t pixels_cap,
                  fo_header *hdr);

/* ---- per-pixel trace of one channel (debug aid for kernel parity) ----
 * cls: 0 in-range, 1 below, 2 above, 3 = raw (first two pixels)
 * ctx: H-L, k: Rice parameter the estimator returned before the update,
 * val: value handed to the coder (p-L, L-p-1 or p-H-1), nbits: bits emitted. */
int fo_trace_channel(const int32_t *channel, uint32_t w, uint32_t h, int depth,
                     uint8_t *cls, uint32_t *ctx, uint8_t *k, uint32_t *val,
                     uint32_t *nbits);

/* ---- unit-level hooks used by the KAT tests ---- */
/* bits as '0'/'1' text; mock_order=1 reproduces BitWriterMock (multi-bit
 * fields LSB first, src/coding/bitwrite_mock.rs:30-41), 0 = real MSB-first */
int fo_rice_encode_text(unsigned k, uint32_t v, int mock_order, char *out, size_t cap);
uint32_t fo_rice_code_length(unsigned k, uint32_t v);
int fo_phasein_params(uint32_t n, uint32_t *m, uint32_t *left_p, uint32_t *right_p);
int fo_phasein_encode_text(uint32_t n, uint32_t v, int mock_order, char *out, size_t cap);
int fo_nearest_neighbours(size_t i, size_t width, size_t *a, size_t *b); /* 1 = Some */
void fo_rgb_to_ycocg(int32_t r, int32_t g, int32_t b, int32_t *y, int32_t *co, int32_t *cg);
void fo_ycocg_to_rgb(int32_t y, int32_t co, int32_t cg, int32_t *r, int32_t *g, int32_t *b);

typedef struct fo_kest fo_kest;
fo_kest *fo_kest_new(uint32_t max_context, const uint8_t *k_values, size_t nk,
                     int64_t halve_at /* <0 = None */);
void fo_kest_free(fo_kest *e);
void fo_kest_update(fo_kest *e, uint32_t context, uint32_t encoded);
unsigned fo_kest_get_k(const fo_kest *e, uint32_t context);
void fo_kest_row(const fo_kest *e, uint32_t context, uint32_t *row_out);

/* bit-level round trip through the real-order writer/reader (rice_coding.rs:90-107,
 * phase_in_coding.rs:229-252): encode n values then decode them back */
int fo_rice_roundtrip(unsigned k, const uint32_t *vals, size_t n);
int fo_phasein_roundtrip(uint32_t domain, const uint32_t *vals, size_t n);

#ifdef __cplusplus
}
#endif
#endif
