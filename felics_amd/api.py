"""Host-side mirror of the reference's public surface, bound to libfelics over its C ABI.

Reference items mirrored (paths relative to the reference repository):
  ColorType, PixelDepth, Header, read_header, write_header   src/compression/format.rs:8-84
  DecompressionError                                          src/compression/error.rs:4-19
  CompressDecompress::{compress, decompress}                 src/compression/traits.rs:47-65
  compress_image, decompress_image                            src/compression.rs:412-441

An image is a numpy array: (H, W) for Luma, (H, W, 3) for Rgb; dtype uint8 or uint16 -- the
same four types the trait is implemented for (compression.rs:250, :317).  `to` / `from_` are
binary file objects, standing in for `W: Write` / `R: Read`.

Encode runs on the GPU through the C ABI; this module has no other encode path and raises
FelicsError if the HIP device or libfelics.so is missing.
"""
import ctypes as C
import enum
import importlib.util
import io
import os

import numpy as np

from . import build as _build


class ColorType(enum.IntEnum):  # format.rs:8-12
    Gray = 0
    Rgb = 1


class PixelDepth(enum.IntEnum):  # format.rs:27-31
    Eight = 0
    Sixteen = 1


class FelicsError(RuntimeError):
    """Any non-zero code of the C ABI."""

    def __init__(self, code, detail=""):
        self.code = code
        msg = lib().felics_strerror(code).decode()
        super().__init__("felics error %d: %s%s" % (code, msg, (" (" + detail + ")") if detail else ""))


class DecompressionError(FelicsError):
    """error.rs:4-19; `.kind` is the variant name."""

    KINDS = {-1: "IoError", -2: "InvalidValue", -3: "ValueOverflow", -4: "InvalidDimensions",
             -5: "InvalidColorType", -6: "InvalidPixelDepth", -7: "InvalidSignature"}

    def __init__(self, code):
        super().__init__(code)
        self.kind = self.KINDS.get(code, "Other")


class _CHeader(C.Structure):
    _fields_ = [("color_type", C.c_uint8), ("pixel_depth", C.c_uint8),
                ("width", C.c_uint32), ("height", C.c_uint32)]


class _CStats(C.Structure):
    _fields_ = [("submissions", C.c_uint64), ("ticket_retries", C.c_uint64), ("slot_overflows", C.c_uint64), ("lookback_fallbacks", C.c_uint64),
                ("two_pass", C.c_int), ("failed", C.c_int), ("scatter_fallbacks", C.c_uint64), ("sorted_event_sorts", C.c_uint64),
                ("tile_overflows", C.c_uint64)]


class Header:  # format.rs:44-49
    def __init__(self, color_type, pixel_depth, width, height):
        self.color_type = ColorType(color_type)
        self.pixel_depth = PixelDepth(pixel_depth)
        self.width = int(width)
        self.height = int(height)

    def __eq__(self, other):
        return (self.color_type, self.pixel_depth, self.width, self.height) == (
            other.color_type, other.pixel_depth, other.width, other.height)

    def __repr__(self):
        return "Header(%s, %s, %d, %d)" % (self.color_type.name, self.pixel_depth.name, self.width, self.height)


EXPORTS = [
    "felics_ctx_create", "felics_ctx_destroy", "felics_max_compressed_size", "felics_compress",
    "felics_compress_batch", "felics_compress_batch_device", "felics_submit_batch_device", "felics_wait_batch",
    "felics_read_header", "felics_write_header",
    "felics_decompress", "felics_strerror", "felics_last_error", "felics_set_profiling",
    "felics_stage_count", "felics_stage_name", "felics_get_stage_ms", "felics_get_stage_launches",
    "felics_lane_count", "felics_ctx_lane_count", "felics_get_span_ms", "felics_decompress_with_header", "felics_get_stats", "felics_decompress_batch_device",
]

_lib = None


def _share_hip_runtime():
    """One HIP runtime per process.  A PyTorch wheel bundles its own libamdhip64.so (same soname as
    the system one); if libfelics pulled in the system copy first and torch its own later, the second
    runtime would find no GPU.  Loading torch's copy first (without importing torch) lets both
    resolve to the same file."""
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    path = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(path):
        try:
            C.CDLL(path, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    """The loaded libfelics.so with argument types declared."""
    global _lib
    if _lib is not None:
        return _lib
    # libfelics overlaps kernels on four HIP streams; give them hardware queues of their own
    # (ROCm default: 4 per process).  Read by the HIP runtime when it initialises.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    _share_hip_runtime()
    L = C.CDLL(_build.ensure_lib())
    vp, sz = C.c_void_p, C.c_size_t
    L.felics_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.felics_ctx_destroy.argtypes = [vp]
    L.felics_ctx_destroy.restype = None
    L.felics_max_compressed_size.argtypes = [C.c_uint32, C.c_uint32, C.c_int, C.c_int]
    L.felics_max_compressed_size.restype = sz
    L.felics_compress.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_int, C.c_int, vp, sz, C.POINTER(sz)]
    L.felics_compress_batch.argtypes = [vp, sz, C.POINTER(vp), C.c_uint32, C.c_uint32, C.c_int, C.c_int,
                                        C.POINTER(vp), C.POINTER(sz), C.POINTER(sz)]
    L.felics_compress_batch_device.argtypes = [vp, sz, vp, C.c_uint32, C.c_uint32, C.c_int, C.c_int, vp, sz,
                                               C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.felics_submit_batch_device.argtypes = [vp, sz, vp, C.c_uint32, C.c_uint32, C.c_int, C.c_int, vp, sz, C.POINTER(C.c_int)]
    L.felics_wait_batch.argtypes = [vp, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.felics_read_header.argtypes = [vp, sz, C.POINTER(_CHeader)]
    L.felics_write_header.argtypes = [C.POINTER(_CHeader), vp, sz]
    L.felics_decompress.argtypes = [vp, sz, vp, sz, C.POINTER(_CHeader)]
    L.felics_decompress_with_header.argtypes = [vp, sz, C.POINTER(_CHeader), vp, sz]
    L.felics_get_stats.argtypes = [vp, C.POINTER(_CStats)]
    L.felics_decompress_batch_device.argtypes = [vp, sz, vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), vp, sz,
                                                 C.POINTER(_CHeader), C.POINTER(C.c_int)]
    L.felics_strerror.argtypes = [C.c_int]
    L.felics_strerror.restype = C.c_char_p
    L.felics_last_error.argtypes = [vp]
    L.felics_last_error.restype = C.c_char_p
    L.felics_set_profiling.argtypes = [vp, C.c_int]
    L.felics_stage_name.argtypes = [C.c_int]
    L.felics_stage_name.restype = C.c_char_p
    L.felics_get_stage_ms.argtypes = [vp, C.POINTER(C.c_float), C.c_int]
    L.felics_get_span_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.felics_get_stage_launches.argtypes = [vp, C.POINTER(C.c_int), C.c_int]
    L.felics_ctx_lane_count.argtypes = [vp]
    _lib = L
    return L


def _describe(image):
    image = np.ascontiguousarray(image)
    if image.dtype == np.uint8:
        depth = PixelDepth.Eight
    elif image.dtype == np.uint16:
        depth = PixelDepth.Sixteen
    else:
        raise TypeError("Unsupported image format: %s" % image.dtype)  # cfelics.rs:69-72
    if image.ndim == 2:
        color = ColorType.Gray
    elif image.ndim == 3 and image.shape[2] == 3:
        color = ColorType.Rgb
    else:
        raise TypeError("Unsupported image format: shape %s" % (image.shape,))
    return image, image.shape[1], image.shape[0], color, depth


class Encoder:
    """One GPU context (felics_ctx): one device, one HIP stream, a reusable workspace in HBM."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        rc = lib().felics_ctx_create(device, C.byref(self._h))
        if rc != 0:
            self._h = None
            raise FelicsError(rc, "device %d" % device)
        self.device = device

    def close(self):
        if self._h:
            lib().felics_ctx_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _raise(self, rc):
        raise FelicsError(rc, lib().felics_last_error(self._h).decode())

    def compress(self, image):
        """The whole .felics file of one image as bytes."""
        return self.compress_batch([image])[0]

    def compress_batch(self, images):
        """Files of a list of same-shaped images (one submission: felics_compress_batch)."""
        if not images:
            return []
        descr = [_describe(im) for im in images]
        first = descr[0]
        for d in descr:
            if d[1:] != first[1:]:
                raise ValueError("a batch holds images of one shape and type")
        _, w, h, color, depth = first
        n = len(descr)
        caps = [14 + 8 * 3 + d[0].nbytes + d[0].nbytes // 2 + 64 for d in descr]
        while True:
            outs = [np.empty(c, dtype=np.uint8) for c in caps]
            px = (C.c_void_p * n)(*[d[0].ctypes.data if d[0].size else None for d in descr])
            op = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
            cp = (C.c_size_t * n)(*caps)
            lens = (C.c_size_t * n)()
            rc = lib().felics_compress_batch(self._h, n, px, w, h, int(color), int(depth), op, cp, lens)
            if rc == -8:  # grow to the sizes the library reported and submit again
                grown = [max(c, int(l)) for c, l in zip(caps, lens)]
                if grown == caps:  # (nothing to grow by: not a size of ours -- do not spin)
                    self._raise(rc)
                caps = grown
                continue
            if rc != 0:
                self._raise(rc)
            return [outs[i][: lens[i]].tobytes() for i in range(n)]

    def compress_batch_host(self, pixel_ptrs, n, w, h, color, depth, out_ptrs, caps):
        """felics_compress_batch on raw HOST pointers (lists of n addresses: frames in, buffers of caps[i] bytes out): the
        reference's call shape without Python objects in the way -- the copies run at the link's rate when the memory behind the
        pointers is page-locked (a pinned torch tensor).  Returns the streams' lengths (numpy)."""
        px = (C.c_void_p * n)(*[int(p) for p in pixel_ptrs])
        op = (C.c_void_p * n)(*[int(p) for p in out_ptrs])
        cp = (C.c_size_t * n)(*[int(c) for c in caps])
        lens = (C.c_size_t * n)()
        rc = lib().felics_compress_batch(self._h, n, px, w, h, int(color), int(depth), op, cp, lens)
        if rc == -8:
            raise FelicsError(rc, "a stream needs up to %d bytes" % max(lens))
        if rc != 0:
            self._raise(rc)
        return np.array(lens[:], dtype=np.uint64)

    def compress_batch_device(self, d_pixels, n, w, h, color, depth, d_out, d_out_cap):
        """Frames and streams in device memory (raw pointers). Returns (offsets, lens) numpy arrays."""
        offs = np.zeros(n, dtype=np.uint64)
        lens = np.zeros(n, dtype=np.uint64)
        rc = lib().felics_compress_batch_device(
            self._h, n, d_pixels, w, h, int(color), int(depth), d_out, d_out_cap,
            offs.ctypes.data_as(C.POINTER(C.c_uint64)), lens.ctypes.data_as(C.POINTER(C.c_uint64)))
        if rc == -8:
            raise FelicsError(rc, "need %d bytes" % int(lens[0]))
        if rc != 0:
            self._raise(rc)
        return offs, lens

    def submit_batch_device(self, d_pixels, n, w, h, color, depth, d_out, d_out_cap):
        """Queues a batch and returns a ticket; up to two can be in flight (felics_submit_batch_device)."""
        ticket = C.c_int(-1)
        rc = lib().felics_submit_batch_device(self._h, n, d_pixels, w, h, int(color), int(depth), d_out, d_out_cap,
                                              C.byref(ticket))
        if rc != 0:
            self._raise(rc)
        return ticket.value, n

    def wait_batch(self, submission):
        """Blocks until the batch of `submission` (from submit_batch_device) is complete: (offsets, lens)."""
        ticket, n = submission
        offs = np.zeros(n, dtype=np.uint64)
        lens = np.zeros(n, dtype=np.uint64)
        rc = lib().felics_wait_batch(self._h, ticket, offs.ctypes.data_as(C.POINTER(C.c_uint64)),
                                     lens.ctypes.data_as(C.POINTER(C.c_uint64)))
        if rc == -8:
            raise FelicsError(rc, "need %d bytes" % int(lens[0]))
        if rc != 0:
            self._raise(rc)
        return offs, lens

    def decompress_batch_device(self, d_streams, offsets, lens, d_pixels, d_pixels_cap):
        """GPU decoder (felics_decompress_batch_device): streams and pixels in device memory (raw pointers).
        Returns (Header, status array); raises DecompressionError with the first failing stream's code."""
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        lens = np.ascontiguousarray(lens, dtype=np.uint64)
        n = len(offsets)
        status = np.zeros(n, dtype=np.int32)
        ch = _CHeader()
        rc = lib().felics_decompress_batch_device(
            self._h, n, d_streams, offsets.ctypes.data_as(C.POINTER(C.c_uint64)), lens.ctypes.data_as(C.POINTER(C.c_uint64)),
            d_pixels, d_pixels_cap, C.byref(ch), status.ctypes.data_as(C.POINTER(C.c_int)))
        if rc in DecompressionError.KINDS:
            err = DecompressionError(rc)
            err.status = status
            raise err
        if rc != 0:
            self._raise(rc)
        return Header(ch.color_type, ch.pixel_depth, ch.width, ch.height), status

    def lane_count(self):
        """felics_ctx_lane_count: submissions this context can have in flight (fixed when it was created)."""
        return int(lib().felics_ctx_lane_count(self._h))

    def stats(self):
        """felics_get_stats: batches redone (slot overflow, look-back fallback), slow-path / failed flags."""
        st = _CStats()
        lib().felics_get_stats(self._h, C.byref(st))
        return {k: int(getattr(st, k)) for k, _ in _CStats._fields_}

    def set_profiling(self, on):
        lib().felics_set_profiling(self._h, int(bool(on)))

    def stage_ms(self):
        """Per stage: sum of the durations (ms) of its launches in the last submission."""
        n = lib().felics_stage_count()
        buf = (C.c_float * n)()
        lib().felics_get_stage_ms(self._h, buf, n)
        return {lib().felics_stage_name(i).decode(): float(buf[i]) for i in range(n)}

    def span_ms(self):
        """First kernel -> last byte of the last submission collected (profiling on), in ms."""
        v = C.c_float(0)
        lib().felics_get_span_ms(self._h, C.byref(v))
        return float(v.value)

    def stage_launches(self):
        n = lib().felics_stage_count()
        buf = (C.c_int * n)()
        lib().felics_get_stage_launches(self._h, buf, n)
        return {lib().felics_stage_name(i).decode(): int(buf[i]) for i in range(n)}


_default = {}


def default_encoder(device=0):
    if device not in _default:
        _default[device] = Encoder(device)
    return _default[device]


# ---- the reference's free functions -------------------------------------------------------

def write_header(header, to):
    """format.rs:51-61."""
    ch = _CHeader(int(header.color_type), int(header.pixel_depth), header.width, header.height)
    buf = (C.c_uint8 * 14)()
    rc = lib().felics_write_header(C.byref(ch), buf, 14)
    if rc != 0:
        raise FelicsError(rc)
    to.write(bytes(buf))


def read_header(from_):
    """format.rs:63-84; consumes up to 14 bytes of `from_`."""
    data = from_.read(14)
    ch = _CHeader()
    arr = np.frombuffer(data, dtype=np.uint8)
    rc = lib().felics_read_header(arr.ctypes.data if len(arr) else None, len(arr), C.byref(ch))
    if rc != 0:
        raise DecompressionError(rc)
    return Header(ch.color_type, ch.pixel_depth, ch.width, ch.height)


def compress_image(to, image, encoder=None):
    """compression.rs:412-418: writes the .felics stream of `image` to `to`."""
    enc = encoder or default_encoder()
    to.write(enc.compress(image))


def compress(image, to, encoder=None):
    """CompressDecompress::compress (traits.rs:48-50)."""
    compress_image(to, image, encoder)


def decompress_image(from_):
    """compression.rs:420-441: returns the image as (H,W) / (H,W,3) uint8 / uint16."""
    data = from_.read() if hasattr(from_, "read") else bytes(from_)
    arr = np.frombuffer(data, dtype=np.uint8)
    ch = _CHeader()
    rc = lib().felics_read_header(arr.ctypes.data if len(arr) else None, len(arr), C.byref(ch))
    if rc != 0:
        raise DecompressionError(rc)
    planes = 3 if ch.color_type else 1
    dt = np.uint16 if ch.pixel_depth else np.uint8
    # a corrupt header must not make us allocate before the stream proves it holds that many pixels
    if ch.width * ch.height * planes > max(len(arr), 1) * 8 + 2 * planes:  # a pixel costs at least one bit
        raise DecompressionError(-1)
    shape = (ch.height, ch.width, 3) if planes == 3 else (ch.height, ch.width)
    out = np.zeros(shape, dtype=dt)
    rc = lib().felics_decompress(arr.ctypes.data, len(arr), out.ctypes.data if out.size else None,
                                 out.nbytes, None)
    if rc != 0:
        raise DecompressionError(rc)
    return out


def decompress(from_):
    """CompressDecompress::decompress (traits.rs:57-64)."""
    return decompress_image(from_)


def decompress_with_header(from_, header):
    """CompressDecompress::decompress_with_header (traits.rs:53-56): `from_` is positioned behind the header."""
    data = from_.read() if hasattr(from_, "read") else bytes(from_)
    arr = np.frombuffer(data, dtype=np.uint8)
    planes = 3 if int(header.color_type) else 1
    dt = np.uint16 if int(header.pixel_depth) else np.uint8
    if header.width * header.height * planes > max(len(arr), 1) * 8 + 2 * planes:  # a pixel costs at least one bit
        raise DecompressionError(-1)
    shape = (header.height, header.width, 3) if planes == 3 else (header.height, header.width)
    out = np.zeros(shape, dtype=dt)
    ch = _CHeader(int(header.color_type), int(header.pixel_depth), header.width, header.height)
    rc = lib().felics_decompress_with_header(arr.ctypes.data if len(arr) else None, len(arr), C.byref(ch),
                                             out.ctypes.data if out.size else None, out.nbytes)
    if rc != 0:
        raise DecompressionError(rc)
    return out


def decompress_bytes(data):
    return decompress_image(io.BytesIO(data))
