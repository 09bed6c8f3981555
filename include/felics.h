/*
 * felics.h -- C ABI of libfelics: MI355X-native FELICS lossless image codec.
 *
 * Drop-in boundary for the encode hot path of visanalexandru/felics.  The
 * reference has no FFI layer of its own: its boundary is the public Rust
 * surface in src/compression.rs / src/compression/traits.rs and the
 * cfelics/dfelics command lines.  Every entry point below names the reference
 * interface it replaces (paths relative to the reference repository); the Rust
 * `extern "C"` block a maintainer would add is in INTEGRATION.md.
 *
 * Conventions
 *   - plain C types only; no exceptions or aborts cross this boundary;
 *   - pixels are row-major, native-endian u8 (depth 0) or u16 (depth 1),
 *     interleaved RGBRGB... when color == FELICS_COLOR_RGB -- the layout of
 *     `ImageBuffer::as_raw()` (compression.rs:276, :338);
 *   - the caller owns every buffer; the library keeps no pointer after return;
 *   - a context is bound to one GPU (it owns a few HIP streams) and is not thread-safe:
 *     use one context per host thread / per GPU (the reference is single
 *     threaded and re-entrant on distinct images);
 *   - ENCODE RUNS ON THE GPU ONLY.  There is no CPU encode fallback: without a
 *     usable HIP device felics_ctx_create() returns FELICS_E_HIP.
 *   - every function returning int returns FELICS_OK (0) or a negative code.
 */
#ifndef FELICS_H
#define FELICS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* error.rs:4-19 DecompressionError variants, then the codes this ABI adds */
#define FELICS_OK 0
#define FELICS_E_IO (-1)                  /* DecompressionError::IoError (truncated stream) */
#define FELICS_E_INVALID_VALUE (-2)       /* ::InvalidValue   */
#define FELICS_E_VALUE_OVERFLOW (-3)      /* ::ValueOverflow  */
#define FELICS_E_INVALID_DIMENSIONS (-4)  /* ::InvalidDimensions */
#define FELICS_E_INVALID_COLOR_TYPE (-5)  /* ::InvalidColorType  */
#define FELICS_E_INVALID_PIXEL_DEPTH (-6) /* ::InvalidPixelDepth */
#define FELICS_E_INVALID_SIGNATURE (-7)   /* ::InvalidSignature  */
#define FELICS_E_BUFFER_TOO_SMALL (-8)    /* caller's output buffer cannot hold the result */
#define FELICS_E_HIP (-9)                 /* no device / HIP runtime error (see felics_last_error) */
#define FELICS_E_UNSUPPORTED (-10)        /* valid request this build cannot run on the GPU (one image of >= 3.7 G samples; a 16-bit image of > 2^29 pixels) */
#define FELICS_E_INVALID_ARGUMENT (-11)   /* NULL pointer, bad enum value */

/* format.rs:8-12 ColorType, format.rs:27-31 PixelDepth (wire values) */
#define FELICS_COLOR_GRAY 0
#define FELICS_COLOR_RGB 1
#define FELICS_DEPTH_8 0
#define FELICS_DEPTH_16 1

/* format.rs:44-49 `pub struct Header` */
typedef struct felics_header {
    uint8_t color_type;
    uint8_t pixel_depth;
    uint32_t width;
    uint32_t height;
} felics_header;

#define FELICS_HEADER_BYTES 14 /* "FLCS", u8 colour, u8 depth, u32 BE width, u32 BE height */

typedef struct felics_ctx felics_ctx;

/* Binds a context to HIP device `device` (>= 0).  Replaces nothing in the
 * reference (it has no device); it is the handle the Rust wrapper would keep
 * in a `struct Encoder`. */
int felics_ctx_create(int device, felics_ctx **out);
void felics_ctx_destroy(felics_ctx *ctx);

/* Upper bound of the .felics size of a w x h image (worst case of the code:
 * 2 flag bits + unary(emax) + terminator at k = 0 for every pixel). */
size_t felics_max_compressed_size(uint32_t w, uint32_t h, int color, int depth);

/* Replaces `<ImageBuffer<Luma<T>|Rgb<T>, Vec<T>> as CompressDecompress>::compress`
 * (compression.rs:255-282, :322-371) and `compress_image` (:412-418): writes the
 * WHOLE file (14-byte header + bit stream, zero padded to a byte) to `out`.
 * A Rust `compress<W: Write>(&self, to: W)` is: call this into a Vec<u8>, then
 * `to.write_all(&buf[..out_len])`.  On FELICS_E_BUFFER_TOO_SMALL *out_len holds
 * the size needed. */
int felics_compress(felics_ctx *ctx, const void *pixels, uint32_t w, uint32_t h, int color,
                    int depth, uint8_t *out, size_t cap, size_t *out_len);

/* n images of one shape in one submission (BASELINE config 3/5: a batch of
 * frames on one GPU).  pixels[i], outs[i], caps[i], lens[i] per image.  Images
 * are independent streams, exactly n calls of felics_compress. */
int felics_compress_batch(felics_ctx *ctx, size_t n, const void *const *pixels, uint32_t w,
                          uint32_t h, int color, int depth, uint8_t *const *outs,
                          const size_t *caps, size_t *lens);

/* Same, with input frames and output already in DEVICE memory (what a capture
 * or decode pipeline that lives on the GPU calls; what bench.py times).
 *   d_pixels : n frames back to back (w*h*channels samples each)
 *   d_out    : device buffer of d_out_cap bytes; stream i is written at
 *              offsets[i] (16-byte aligned, ascending) with lens[i] bytes
 *   offsets, lens : HOST arrays of n entries, filled on return
 * On FELICS_E_BUFFER_TOO_SMALL lens[0] holds the capacity needed.
 * The library works on HIP streams of its own (created non-blocking): the frames must be COMPLETE in memory when the
 * call is made -- synchronise the stream that produced them first (hipStreamSynchronize / an event the host has
 * waited for); the same holds for felics_submit_batch_device and for the streams handed to
 * felics_decompress_batch_device.  What the library wrote is complete when the blocking call / felics_wait_batch returns. */
int felics_compress_batch_device(felics_ctx *ctx, size_t n, const void *d_pixels, uint32_t w,
                                 uint32_t h, int color, int depth, void *d_out,
                                 size_t d_out_cap, uint64_t *offsets, uint64_t *lens);

/* The same submission in two halves, for callers that encode batch after batch: felics_submit_batch_device
 * queues the work and returns, felics_wait_batch(ticket) blocks until that batch is complete and fills
 * offsets / lens.  Up to felics_lane_count() submissions can be in flight, each with its own d_out;
 * tickets must be waited for in the order they were handed out.  While the GPU is finishing one batch (its
 * last pack slices) it already classifies, scatters and replays the estimator of the next one, which hides
 * the latency-bound head and tail of a batch.  The synchronous entry points refuse to run
 * (FELICS_E_INVALID_ARGUMENT) while a ticket is outstanding.  The reference has no counterpart (it is one
 * blocking call per image); a Rust wrapper would expose this as a two-deep pipeline over `compress`. */
int felics_submit_batch_device(felics_ctx *ctx, size_t n, const void *d_pixels, uint32_t w,
                               uint32_t h, int color, int depth, void *d_out, size_t d_out_cap,
                               int *ticket);
int felics_wait_batch(felics_ctx *ctx, int ticket, uint64_t *offsets, uint64_t *lens);

/* Replaces `read_header` (format.rs:63-84). */
int felics_read_header(const uint8_t *in, size_t len, felics_header *hdr);
/* Replaces `write_header` (format.rs:51-61): writes FELICS_HEADER_BYTES bytes. */
int felics_write_header(const felics_header *hdr, uint8_t *out, size_t cap);

/* Replaces `decompress_image` / `CompressDecompress::decompress`
 * (compression.rs:420-441, traits.rs:57-64).  Host (CPU) implementation: the
 * entropy decoder is bit-serial per plane.  `pixels` receives
 * width*height*channels samples of the depth the header states. */
int felics_decompress(const uint8_t *in, size_t len, void *pixels, size_t pixels_cap,
                      felics_header *hdr);

/* Replaces `CompressDecompress::decompress_with_header(from, &Header)` (traits.rs:53-56; compression.rs:284-314,
 * :373-409): the caller has read (or knows) the header; `in` points at the bit stream BEHIND the 14 header bytes.
 * The header's claims are checked against pixels_cap and against the stream length before anything is allocated. */
int felics_decompress_with_header(const uint8_t *in, size_t len, const felics_header *hdr, void *pixels,
                                  size_t pixels_cap);

/* GPU decoder: n streams of ONE shape resident in device memory (stream i at d_streams + offsets[i], lens[i] bytes:
 * exactly what felics_compress_batch_device leaves behind) decoded into d_pixels (n frames back to back, the layout
 * the encoder takes).  offsets / lens / status are HOST arrays of n entries; status[i] is FELICS_OK or the
 * DecompressionError code of stream i; the function returns the first non-zero status (or a HIP / argument
 * error).  *hdr (optional) receives the header all streams must share; it is read from stream 0.
 * Replaces n calls of `decompress_image` (compression.rs:420-441).  The format is bit-serial per stream (and the
 * planes of an RGB image share one bit stream), so the only parallelism is across streams: one wave per stream
 * for 8-bit and for 16-bit data (k_decode16: a 8.4 MB estimator table per stream in device memory, rows tagged with an epoch).
 * status[] is written for all n streams on every return (a call that ends before decoding -- bad header of stream 0, buffer too
 * small, a HIP error -- puts its own code in every entry).  The kernel loads a stream as whole ALIGNED 32-bit words: it may
 * touch up to three bytes on either side of a stream, always inside an aligned word that also holds a byte of the stream,
 * hence never outside the page the stream lies in; those bytes are never interpreted. */
int felics_decompress_batch_device(felics_ctx *ctx, size_t n, const void *d_streams, const uint64_t *offsets,
                                   const uint64_t *lens, void *d_pixels, size_t d_pixels_cap, felics_header *hdr,
                                   int *status);

/* Text for a code above; for FELICS_E_HIP felics_last_error(ctx) has the HIP message. */
const char *felics_strerror(int code);
const char *felics_last_error(const felics_ctx *ctx);

/* What has happened to a context that the return codes do not say: how often it had to redo a batch, and
 * whether it is on a slower path or unusable.  (A look-back / hand-off that gives up -- e.g. another process
 * holding the GPU for a second -- moves the context to the two-pass kernels for good; felics_last_error says so.) */
typedef struct felics_stats {
    uint64_t submissions;        /* sub-batches queued so far */
    uint64_t ticket_retries;     /* 0 or 1: a look-back gave up and the context now hands its pack tiles out by ticket (first remedy) */
    uint64_t slot_overflows;     /* batches redone with exact placement: a stream outgrew its fixed slot */
    uint64_t lookback_fallbacks; /* batches redone because a tile gave up waiting for its predecessors */
    int two_pass;                /* 1: a ticketed look-back gave up as well: the context packs with the two-pass kernels from now on (slower) */
    int failed;                  /* 1: a wait for the GPU timed out; every further call returns FELICS_E_HIP */
    uint64_t scatter_fallbacks;  /* sub-batches redone because the front kernel's check of its own event order failed (with submissions in flight each
                                    reports its own, so 0 .. lanes); the context ranks events with ballots from then on (slower, no assumption about the LDS) */
    uint64_t sorted_event_sorts; /* sub-batches of 8-bit samples whose events were ranked with returning LDS atomics (the default), not with ballots */
    uint64_t tile_overflows;     /* sub-batches redone because a tile's events outgrew the slots a tile gets by default; the context sizes its tiles for the
                                    worst case from then on (more memory, same kernels) */
} felics_stats;
int felics_get_stats(const felics_ctx *ctx, felics_stats *out);

/* ---- measurement hooks (bench.py) ----
 * With profiling on, every kernel launch of the next submission gets a start / stop HIP event on the
 * stream it runs on (the kernel's own begin / end for single-kernel stages, event records around the
 * launches of the few stages that are several small kernels).  A submission launches most kernels once per slice of the image (the stages
 * follow each other slice by slice on several streams); felics_get_stage_ms gives, per stage, the SUM
 * of its launches' durations in milliseconds (launches of different stages overlap, so the stages add
 * up to more than the wall time), felics_get_stage_launches how many launches that was. */
#define FELICS_MAX_STAGES 16
int felics_set_profiling(felics_ctx *ctx, int enabled);
int felics_stage_count(void);
const char *felics_stage_name(int stage);
int felics_get_stage_ms(const felics_ctx *ctx, float *ms, int cap);
int felics_get_stage_launches(const felics_ctx *ctx, int *launches, int cap);
/* The same submission from its first kernel to its last byte (stream sizes on the host): one HIP event in front of the first
 * launch, one behind the size copy, on the streams they run on -- BASELINE.md section 2's per-step span. */
int felics_get_span_ms(const felics_ctx *ctx, float *ms);
/* Submissions that can be in flight at a time (felics_submit_batch_device): felics_ctx_lane_count for an existing context
 * (fixed when it was created), felics_lane_count for a context created now (2 unless FELICS_LANES says otherwise). */
int felics_lane_count(void);
int felics_ctx_lane_count(const felics_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* FELICS_H */
