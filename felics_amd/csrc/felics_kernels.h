// felics_kernels.h -- launch interface between the host pipeline (felics_api.cpp)
// and the gfx950 kernels (felics_kernels.hip).  Not part of the public C ABI.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include <algorithm>

namespace felics {

// Contexts (H - L of a pixel's two neighbours, traits.rs:28): 0..255 for u8 samples, 0..510 for the Y/Co/Cg planes of
// RGB8 (table padded to 512).  NCTX sizes what is shared by both (upper bounds); the run table, the chain tables and the
// spine's grid use the sample type's own count, nctx_of<T>() = Geometry::nctx.
constexpr uint32_t NCTX = 512;
template <typename T>
constexpr uint32_t nctx_of() {
    return sizeof(T) == 1 ? 256u : 512u;  // T = sample type (u8 / i16) or event type (u8 / u16)
}
// pixels one workgroup sorts by context in the front stage.  Equal to the pack tile: the pack stage reads back k for exactly
// its own tile's events (k_pack_t), one look-back per workgroup.
constexpr uint32_t SORT_TILE = 4096;
// pack stage: 256 threads x 16 consecutive pixels
constexpr uint32_t PACK_THREADS = 256;
constexpr uint32_t PACK_PER_THREAD = 16;
constexpr uint32_t PACK_TILE = PACK_THREADS * PACK_PER_THREAD;
// LDS bit window of the pack stage: 2048 words = 8 KiB = 16 bits per pixel of a tile (more bits: more windows)
constexpr uint32_t PACK_WIN_WORDS = 2048;

// Sample type of a plane -> type of the per-group bit counts k_lengths hands to k_pack.
// 8-bit samples (u8 gray, i16 Y/Co/Cg): a 16-pixel group is at most 16 * 513 bits.  16-bit samples
// (u16 gray, i32 Y/Co/Cg): one code can be 2^17 bits long.
template <typename T> struct GroupBits { using type = uint16_t; };
template <> struct GroupBits<uint16_t> { using type = uint32_t; };
template <> struct GroupBits<int32_t> { using type = uint32_t; };
template <typename T> using group_bits_t = typename GroupBits<T>::type;

// Profiling: the host pipeline sets g_launch_timing around ONE launch to have that kernel's own begin and
// end recorded in the two events (hipExtLaunchKernel: the dispatch's timestamps, what rocprofv3 reports),
// instead of bracketing the launch with event records, which also measure the launch's wait for free
// compute resources.  The launcher that consumes it clears it.
struct LaunchTiming {
    hipEvent_t start = nullptr, stop = nullptr;
};
extern thread_local LaunchTiming g_launch_timing;

#define FELICS_LAUNCH(KERNEL, GRID, BLOCK, STREAM, ...)                                                    \
    do {                                                                                                   \
        const ::felics::LaunchTiming lt_ = ::felics::g_launch_timing;                                      \
        ::felics::g_launch_timing = ::felics::LaunchTiming{};                                              \
        if (lt_.start)                                                                                     \
            hipExtLaunchKernelGGL(KERNEL, GRID, BLOCK, 0, STREAM, lt_.start, lt_.stop, 0, __VA_ARGS__);   \
        else                                                                                               \
            hipLaunchKernelGGL(KERNEL, GRID, BLOCK, 0, STREAM, __VA_ARGS__);                               \
    } while (0)

struct Geometry {
    uint32_t W, H;
    uint32_t npix;              // W*H, pixels per plane
    uint32_t nimages;
    uint32_t planes_per_image;  // 1 gray, 3 rgb
    uint32_t nplanes;           // nimages * planes_per_image
    uint32_t sort_tiles;        // ceil(npix / SORT_TILE)
    uint32_t pack_tiles;        // ceil(npix / PACK_TILE)
    uint32_t color, depth;      // header fields
    uint32_t nctx;              // contexts per plane: nctx_of<sample type>()
};

void launch_rgb8_to_planes(hipStream_t s, const uint8_t *rgb, int16_t *planes, uint32_t npix, uint32_t nimg);

// ------------------------------------------------------------------------------------------
// The 8-bit pipeline in TILE-LOCAL layout (round 5).  A pixel is classified once: the front kernel sorts the events of
// a tile (SORT_TILE pixels) by context in LDS and writes them as ONE contiguous piece at a fixed place,
//     slot (plane * sort_tiles + tile) * cap + s,
// contexts ascending, raster order inside a context, every context's run starting on a multiple of REC slots (the slots
// between a run's end and the next multiple are padding: pix = 0xFFFF).  A RECORD is REC = 16 consecutive slots of one
// run: the unit of the chain stage -- the spine leaves the estimator's state at the start of every record, the assign
// kernel replays a record per lane -- and of the k bytes the pack kernel reads back, contiguous per tile.  No histogram
// pass, no tile offsets, no chain bases, no global scatter: the chain of a context is the sequence of its runs over the
// tiles, listed per slice by k_enum (record descriptors in chain order).
// ------------------------------------------------------------------------------------------
constexpr uint32_t REC = 16;  // events per record
// Slots per tile.  Worst case SORT_TILE + (REC - 1) * nctx (every context one event over a multiple of REC); the default
// covers anything but adversarial content (uniform noise: 216 runs of a 4096-pixel tile, ~4400 slots); a tile that needs
// more raises TL_FLAG_OVERFLOW and the host redoes the batch with the worst case (felics_api.cpp).
// (npix: pixels per plane -- a plane smaller than a tile needs less)
inline uint32_t tile_cap_max(uint32_t nctx, uint32_t npix) {
    const uint32_t px = std::min(npix, SORT_TILE);
    return (px + (REC - 1) * std::min(nctx, px) + REC - 1) / REC * REC;
}
inline uint32_t tile_cap_default(uint32_t nctx, uint32_t npix) { return std::min(nctx == 256 ? 6144u : 8192u, tile_cap_max(nctx, npix)); }
constexpr uint32_t TL_FLAG_ORDER = 1u, TL_FLAG_OVERFLOW = 2u, TL_FLAG_SPINE = 4u;
constexpr uint32_t FRONT_TEST_VIOLATION = 1u, FRONT_SAFE_RANK = 2u;  // k_front's `mode` bits

template <typename ET>
struct TileLocal {
    ET *ev;                // [slot] value to Rice-code
    uint16_t *pix;         // [slot] the event's pixel: offset in its tile (12 bits) | above << 12 (the sample lies above its neighbours); 0xFFFF in padding slots
    uint8_t *kq;           // [slot] k of the event (k_assign3)
    uint32_t *runtab;      // [(plane * nctx + c) * sort_tiles + tile] = first record of the tile's run of c (in the tile) | events << 16
    uint32_t *tile_slots;  // [plane * sort_tiles + tile] slots in use (a multiple of REC)
    uint32_t cap;          // slots per tile
};
// classify + sort the tiles [tile_begin, tile_end) of every plane (compression.rs:124-145, misc.rs:6-24)
template <typename T, typename ET>
void launch_front(hipStream_t s, const T *planes, const TileLocal<ET> &tl, const Geometry &g, uint32_t tile_begin, uint32_t tile_end,
                  uint32_t *flags, uint32_t mode);

// The chain stage of one slice: records [0, *nrec) of the slice's region of desc.
struct ChainSlice {
    uint2 *desc;           // [rec] {record's first slot / REC (over the whole sub-batch), events in it}: chain order
    uint2 *chain_seg;      // [chain] {first record, records} of the chain in this slice
    uint32_t *nrec;        // records of the slice (device counter, zeroed per sub-batch)
    uint4 *state16;        // [slot / REC, over the whole sub-batch: the same array for every slice] {S0 | S1 << 16, S2 | S3 << 16, S4 | S5 << 16, slot / REC}: the estimator's state at the record's first event
};
void launch_enum(hipStream_t s, const uint32_t *runtab, const ChainSlice &cs, const Geometry &g, uint32_t tile_begin, uint32_t tile_end,
                 uint32_t cap);
// chain_state: 8 words per chain (zeroed per sub-batch): the state behind the chain's last event so far
template <typename ET>
void launch_spine3(hipStream_t s, const ET *ev, const ChainSlice &cs, uint32_t *chain_state, uint32_t *flags, const Geometry &g);
// k of the events of the tiles [tile_begin, tile_end) of every plane, from the states k_spine3 left
template <typename ET>
void launch_assign3(hipStream_t s, const TileLocal<ET> &tl, const uint4 *state16, const Geometry &g, uint32_t tile_begin, uint32_t tile_end);
// two-pass pack: k from the tiles' slots to a byte per pixel
void launch_k_to_pixels_tl(hipStream_t s, const uint8_t *kq, const uint16_t *pix, const uint32_t *tile_slots, uint32_t cap, uint8_t *k_map,
                           const Geometry &g);

// k_map / plane / slot buffers are read in whole 16-byte chunks by the tile staging
constexpr size_t STAGE_PAD = 64;

// lengths / bit scan / pack work on a range [t0, t1) of every plane's PACK tiles, so they can follow the
// spine slice by slice.  tile_bitoff is relative to the plane; plane_base (zero for gray, set by
// launch_finish_sizes for the later planes of an RGB image) makes it relative to the image stream.
template <typename T>
void launch_lengths(hipStream_t s, const T *planes, const uint8_t *k_map, group_bits_t<T> *group_bits,
                    uint32_t *tile_bits, const Geometry &g, uint32_t t0, uint32_t t1);

void launch_bitscan_slice(hipStream_t s, const uint32_t *tile_bits, uint64_t *tile_bitoff, uint64_t *plane_carry,
                          const Geometry &g, uint32_t t0, uint32_t t1);

void launch_finish_sizes(hipStream_t s, const uint64_t *plane_carry, uint64_t *plane_base, uint64_t *image_bytes,
                         const Geometry &g);

// exact placement (streams back to back, 16-byte aligned) and zeroing of exactly those bytes
void launch_place_streams(hipStream_t s, const uint64_t *image_bytes, uint64_t *image_off, const Geometry &g);
void launch_zero_streams(hipStream_t s, uint32_t *out, const uint64_t *image_off, const Geometry &g);

// Placement of the streams in `out`: slot_stride != 0 -> stream i at i * slot_stride (bytes), writes
// beyond the slot are dropped; slot_stride == 0 -> stream i at image_off[i].
void launch_zero_edges(hipStream_t s, uint8_t *out, const uint64_t *image_off, uint64_t slot_stride,
                       const uint64_t *tile_bitoff, const uint32_t *tile_bits, const uint64_t *plane_base,
                       const Geometry &g, uint32_t t0, uint32_t t1);

template <typename T>
void launch_pack(hipStream_t s, const T *planes, const uint8_t *k_map, const group_bits_t<T> *group_bits,
                 const uint64_t *tile_bitoff, const uint32_t *tile_bits, const uint64_t *plane_base,
                 const uint64_t *image_off, uint64_t slot_stride, uint8_t *out, const Geometry &g, uint32_t t0,
                 uint32_t t1);

// Single-pass pack for 8-bit frames with fixed output slots (image i's stream at out + i * slot_stride):
// code lengths, tile offsets (decoupled look-back through `status`, one u64 per tile, zero-initialised
// once, `epoch` distinguishes submissions) and packing in one kernel per slice of tiles.  Writes
// tile_bitoff / tile_bits / plane_carry like the lengths + bitscan kernels.  The words two tiles share are
// left in edge_first / edge_last; launch_join_edges stores them once every tile is done.  Planes 1, 2 of
// an RGB image are packed into scratch slots (plane c of image i at scratch + (2 i + c - 1) * plane_slot)
// and moved behind plane 0 by launch_concat_planes after launch_finish_sizes.
// *error: bit 0 = a look-back gave up waiting, bit 1 = a plane outgrew its scratch slot.
struct PackTarget {
    uint8_t *out;
    uint64_t slot_stride;
    uint8_t *scratch;
    uint64_t plane_slot;
};
// the single-pass pack on the tile-local layout: the events' codes built from the tile's own slots (kq / pix / ev / tile_slots of TileLocal)
template <typename T>
void launch_pack_t(hipStream_t s, const T *planes, const uint8_t *kq, const uint16_t *pix, const void *ev, const uint32_t *tile_slots, uint32_t cap,
                   uint64_t *status, uint64_t *tile_bitoff, uint32_t *tile_bits, uint64_t *plane_carry, uint32_t *edge_first,
                   uint32_t *edge_last, uint32_t *error, const PackTarget &to, const Geometry &g, uint32_t st0, uint32_t st1, uint32_t epoch,
                   uint32_t *ticket);
void launch_join_edges(hipStream_t s, const uint64_t *tile_bitoff, const uint32_t *tile_bits, const uint32_t *edge_first,
                       const uint32_t *edge_last, const PackTarget &to, const Geometry &g);
void launch_concat_planes(hipStream_t s, const uint64_t *plane_base, const uint64_t *plane_carry, const PackTarget &to,
                          const Geometry &g);

// Where the single-pass pack puts a plane's bits.  Plane 0 of an image goes to the image's slot of the
// output (header first); planes 1, 2 of an RGB image are packed as bit strings of their own into scratch
// slots and moved behind plane 0 by k_concat_planes once every size is known (compression.rs:365-367:
// the planes of an image follow each other without alignment).
struct PlaneOut {
    uint8_t *out;
    uint64_t slot_stride;  // image i's stream starts at out + i * slot_stride
    uint8_t *scratch;
    uint64_t plane_slot;   // plane c >= 1 of image i at scratch + (i * (planes_per_image - 1) + c - 1) * plane_slot
    uint32_t planes_per_image;
};


// join_edges for a given tile count
void launch_join_edges_tiles(hipStream_t s, const uint64_t *tile_bitoff, const uint32_t *tile_bits, const uint32_t *edge_first,
                             const uint32_t *edge_last, const PackTarget &to, const Geometry &g, uint32_t ntiles);

constexpr uint32_t DECODE_LDS_LIMIT = 160u * 1024u;  // LDS of one CU (MI355X_MICROARCH.md): what a decoder workgroup may ask for
// ---- GPU decoder for 8-bit streams (felics_gpudecode.hip): one wave per stream.  status[i] = FELICS_OK or an error
// code; gray pixels go straight to `pixels`, RGB through int16 planes (image i at i * 3 * npix) + a conversion kernel.
uint32_t decode8_lds_bytes(uint32_t W, uint32_t color);
// 16-bit streams: the estimator tables live in HBM (decode16_table_bytes(n): 8.4 MB per stream, zero-initialised ONCE: rows carry
// the epoch they were written in; a call uses epochs epoch0 .. epoch0 + 2, never 0 and never reused on the same buffer)
uint32_t decode16_lds_bytes(uint32_t W);
size_t decode16_table_bytes(uint32_t n);
hipError_t launch_decode16(hipStream_t s, const uint8_t *streams, const uint64_t *offsets, const uint64_t *lens, uint32_t n,
                           uint32_t W, uint32_t H, uint32_t color, uint16_t *pixels, int32_t *planes, uint32_t *table,
                           uint32_t epoch0, int *status);
// The same for large batches of 8-bit streams, gray or RGB (felics_gpudecode.hip, k_decode8_lanes): 64 streams per wave, lane = stream;
// needs W >= 8 and a zeroed table of decode8_lanes_table_bytes(n, color) bytes (3 KB per gray stream, 18 KB per RGB stream: the
// estimator rows that do not live in LDS); RGB: `planes` as for launch_decode8.
constexpr uint32_t DECODE8_LANES_MIN_STREAMS = 1536;  // measured (profiles/r04/decode_scaling.txt): one wave per stream saturates at ~2.3 GPix/s from ~1000
                                                       // streams, a lane decodes 1.47 MPix/s whatever the batch: the forms cross at ~1500 streams
constexpr uint32_t DECODE8_LANES_MIN_STREAMS_RGB = 2048;  // RGB8 (profiles/r05/decode_scaling_rgb.txt, 1080p frames): 0.88 against 0.94 GPix/s at 2048 streams, 0.97 / 1.88 at 4096
size_t decode8_lanes_table_bytes(uint32_t n, uint32_t color);
hipError_t launch_decode8_lanes(hipStream_t s, const uint8_t *streams, const uint64_t *offsets, const uint64_t *lens, uint32_t n,
                                uint32_t W, uint32_t H, uint32_t color, uint8_t *pixels, int16_t *planes, uint32_t *table, int *status);
hipError_t launch_decode8(hipStream_t s, const uint8_t *streams, const uint64_t *offsets, const uint64_t *lens, uint32_t n,
                          uint32_t W, uint32_t H, uint32_t color, uint8_t *pixels, int16_t *planes, int *status);

// ---- 16-bit samples (felics_wide.hip): contexts 0..131070 and 15 Rice parameters (traits.rs:35-43).
// The events of a batch are compacted into 64-bit records {context, Rice operand, sample index in its plane},
// ordered by (plane, context) with a stable radix sort (two 9-bit passes per plane; the records are written plane by
// plane) and every context's chain is replayed by one wave; lengths / pack are the kernels above on u16 / i32 planes.
constexpr uint32_t WIDE_MAX_PLANE_PIXELS = 1u << 29;  // the sample index in a record has 29 bits

void launch_rgb16_to_planes(hipStream_t s, const uint16_t *rgb, int32_t *planes, uint32_t npix, uint32_t nimg);

struct WideSizes {
    uint32_t px_tiles, max_sort_tiles;
    size_t tile_cnt_bytes, meta_bytes, rec_bytes, hist_bytes, heads_bytes, digtot_bytes;
};
WideSizes wide_sizes(const Geometry &g);

// count -> scan -> emit: the records of all events, plane by plane in raster order; meta = totals and per-plane ranges
template <typename T>
void launch_wide_events(hipStream_t s, const T *planes, uint32_t *tile_cnt, uint32_t *meta, uint64_t *recs, const Geometry &g);
// stable sort by context inside every plane; the result is in recs_a again
void launch_wide_sort(hipStream_t s, uint64_t *recs_a, uint64_t *recs_b, const uint32_t *meta, uint32_t *hist, uint32_t *dig_tot,
                      const Geometry &g);
// chain heads and the replay of the estimator along every chain: k_map[plane * npix + i] = k.  counters = {heads, chains
// handed over}: two words, zero beforehand.  lane_limit != 0: four lanes per chain for its first lane_limit events
// (k_wide_chains_quad), the wave-per-chain kernel for the rest (long_heads: 8 bytes, long_state: 64 bytes per chain handed
// over, wide_long_capacity() of them); lane_limit == 0: the wave-per-chain kernel alone.
constexpr uint32_t WIDE_LANE_LIMIT_MIN = 2048, WIDE_LANE_LIMIT_MAX = 4096;
constexpr uint64_t WIDE_LANE_MIN_SAMPLES = 96u << 20;  // (measured: 8 4K planes 1.28 against 1.49 ms for the wave-wide form, 16: 2.74 against 2.31, 32: 5.40 against 3.60)
uint32_t wide_lane_limit(const Geometry &g);
size_t wide_long_capacity(const Geometry &g, uint32_t lane_limit);
void launch_wide_chains(hipStream_t s, const uint64_t *recs, const uint32_t *meta, uint64_t *heads, uint32_t *counters,
                        uint8_t *k_map, const Geometry &g, uint32_t lane_limit, uint64_t *long_heads, uint32_t *long_state);

}  // namespace felics
