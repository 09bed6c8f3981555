"""ctypes view of oracle/_build/libfelics_oracle.so (the CPU checker).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "_build", "libfelics_oracle.so")

ERRORS = {
    0: "OK", -1: "IO", -2: "INVALID_VALUE", -3: "VALUE_OVERFLOW", -4: "INVALID_DIMENSIONS",
    -5: "INVALID_COLOR_TYPE", -6: "INVALID_PIXEL_DEPTH", -7: "INVALID_SIGNATURE",
    -8: "BUFFER_TOO_SMALL",
}


class Header(C.Structure):
    _fields_ = [("color_type", C.c_uint8), ("pixel_depth", C.c_uint8),
                ("width", C.c_uint32), ("height", C.c_uint32)]


class OracleError(RuntimeError):
    def __init__(self, code):
        super().__init__("oracle error %d (%s)" % (code, ERRORS.get(code, "?")))
        self.code = code


def build():
    src = os.path.join(ORACLE_DIR, "felics_oracle.c")
    if (not os.path.exists(LIB_PATH)) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR], stdout=subprocess.DEVNULL)
    return LIB_PATH


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        u8p, u32p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32)
        lib.fo_max_compressed_size.restype = C.c_size_t
        lib.fo_max_compressed_size.argtypes = [C.c_uint32, C.c_uint32, C.c_int, C.c_int]
        lib.fo_compress.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_int,
                                    C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        lib.fo_read_header.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(Header)]
        lib.fo_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(Header)]
        lib.fo_trace_channel.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.fo_rice_encode_text.argtypes = [C.c_uint, C.c_uint32, C.c_int, C.c_char_p, C.c_size_t]
        lib.fo_rice_code_length.restype = C.c_uint32
        lib.fo_rice_code_length.argtypes = [C.c_uint, C.c_uint32]
        lib.fo_phasein_params.argtypes = [C.c_uint32, u32p, u32p, u32p]
        lib.fo_phasein_encode_text.argtypes = [C.c_uint32, C.c_uint32, C.c_int, C.c_char_p, C.c_size_t]
        lib.fo_nearest_neighbours.argtypes = [C.c_size_t, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        i32p = C.POINTER(C.c_int32)
        lib.fo_rgb_to_ycocg.argtypes = [C.c_int32] * 3 + [i32p] * 3
        lib.fo_ycocg_to_rgb.argtypes = [C.c_int32] * 3 + [i32p] * 3
        lib.fo_kest_new.restype = C.c_void_p
        lib.fo_kest_new.argtypes = [C.c_uint32, u8p, C.c_size_t, C.c_int64]
        lib.fo_kest_free.argtypes = [C.c_void_p]
        lib.fo_kest_update.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        lib.fo_kest_get_k.argtypes = [C.c_void_p, C.c_uint32]
        lib.fo_kest_get_k.restype = C.c_uint
        lib.fo_kest_row.argtypes = [C.c_void_p, C.c_uint32, u32p]
        lib.fo_rice_roundtrip.argtypes = [C.c_uint, C.c_void_p, C.c_size_t]
        lib.fo_phasein_roundtrip.argtypes = [C.c_uint32, C.c_void_p, C.c_size_t]

    # ---- whole image ----
    @staticmethod
    def _describe(img):
        img = np.ascontiguousarray(img)
        if img.dtype == np.uint8:
            depth = 0
        elif img.dtype == np.uint16:
            depth = 1
        else:
            raise TypeError(img.dtype)
        if img.ndim == 2:
            color = 0
        elif img.ndim == 3 and img.shape[2] == 3:
            color = 1
        else:
            raise ValueError(img.shape)
        return img, img.shape[1], img.shape[0], color, depth

    def compress(self, img):
        """img: (H,W) or (H,W,3) uint8/uint16 -> bytes of the whole .felics file."""
        img, w, h, color, depth = self._describe(img)
        cap = 14 + 64 + img.nbytes * 2 + 1024
        while True:
            out = np.empty(cap, dtype=np.uint8)
            n = C.c_size_t(0)
            rc = self.lib.fo_compress(img.ctypes.data, w, h, color, depth, out.ctypes.data, cap, C.byref(n))
            if rc == -8:
                cap = n.value + 16
                continue
            if rc != 0:
                raise OracleError(rc)
            return out[: n.value].tobytes()

    def read_header(self, data):
        hdr = Header()
        buf = np.frombuffer(data, dtype=np.uint8)
        rc = self.lib.fo_read_header(buf.ctypes.data if len(buf) else None, len(buf), C.byref(hdr))
        if rc != 0:
            raise OracleError(rc)
        return hdr

    def decompress(self, data):
        hdr = self.read_header(data)
        buf = np.frombuffer(data, dtype=np.uint8)
        planes = 3 if hdr.color_type else 1
        dt = np.uint16 if hdr.pixel_depth else np.uint8
        shape = (hdr.height, hdr.width, 3) if planes == 3 else (hdr.height, hdr.width)
        if hdr.width * hdr.height > (1 << 31):
            raise OracleError(-4)
        out = np.zeros(shape, dtype=dt)
        rc = self.lib.fo_decompress(buf.ctypes.data, len(buf), out.ctypes.data, max(out.nbytes, 1), None)
        if rc != 0:
            raise OracleError(rc)
        return out

    def trace_channel(self, channel, w, h, depth):
        ch = np.ascontiguousarray(channel, dtype=np.int32).reshape(-1)
        n = max(w * h, 2)
        cls = np.zeros(n, np.uint8)
        ctx = np.zeros(n, np.uint32)
        k = np.zeros(n, np.uint8)
        val = np.zeros(n, np.uint32)
        nbits = np.zeros(n, np.uint32)
        rc = self.lib.fo_trace_channel(ch.ctypes.data, w, h, depth, cls.ctypes.data, ctx.ctypes.data,
                                       k.ctypes.data, val.ctypes.data, nbits.ctypes.data)
        if rc != 0:
            raise OracleError(rc)
        return dict(cls=cls[: w * h], ctx=ctx[: w * h], k=k[: w * h], val=val[: w * h], nbits=nbits[: w * h])

    # ---- unit hooks ----
    def rice_text(self, k, v, mock=False):
        buf = C.create_string_buffer(1 << 18)
        n = self.lib.fo_rice_encode_text(k, v, int(mock), buf, len(buf))
        if n < 0:
            raise ValueError("rice(%d,%d) rejected" % (k, v))
        return buf.value.decode()

    def rice_len(self, k, v):
        return self.lib.fo_rice_code_length(k, v)

    def phasein_params(self, n):
        m, l, r = C.c_uint32(), C.c_uint32(), C.c_uint32()
        if self.lib.fo_phasein_params(n, C.byref(m), C.byref(l), C.byref(r)) != 0:
            raise ValueError("phase-in n=%d rejected" % n)
        return m.value, l.value, r.value

    def phasein_text(self, n, v, mock=False):
        buf = C.create_string_buffer(128)
        rc = self.lib.fo_phasein_encode_text(n, v, int(mock), buf, len(buf))
        if rc < 0:
            raise ValueError("phase-in(%d,%d) rejected" % (n, v))
        return buf.value.decode()

    def neighbours(self, i, width):
        a, b = C.c_size_t(), C.c_size_t()
        if self.lib.fo_nearest_neighbours(i, width, C.byref(a), C.byref(b)):
            return a.value, b.value
        return None

    def rgb_to_ycocg(self, r, g, b):
        y, co, cg = C.c_int32(), C.c_int32(), C.c_int32()
        self.lib.fo_rgb_to_ycocg(r, g, b, C.byref(y), C.byref(co), C.byref(cg))
        return y.value, co.value, cg.value

    def ycocg_to_rgb(self, y, co, cg):
        r, g, b = C.c_int32(), C.c_int32(), C.c_int32()
        self.lib.fo_ycocg_to_rgb(y, co, cg, C.byref(r), C.byref(g), C.byref(b))
        return r.value, g.value, b.value

    def estimator(self, max_context, k_values, halve_at):
        return Estimator(self.lib, max_context, k_values, halve_at)

    def rice_roundtrip(self, k, vals):
        v = np.ascontiguousarray(vals, dtype=np.uint32)
        return self.lib.fo_rice_roundtrip(k, v.ctypes.data, len(v)) == 0

    def phasein_roundtrip(self, n, vals):
        v = np.ascontiguousarray(vals, dtype=np.uint32)
        return self.lib.fo_phasein_roundtrip(n, v.ctypes.data, len(v)) == 0


class Estimator:
    def __init__(self, lib, max_context, k_values, halve_at):
        self.lib = lib
        self.nk = len(k_values)
        arr = (C.c_uint8 * self.nk)(*k_values)
        self.h = lib.fo_kest_new(max_context, arr, self.nk, -1 if halve_at is None else halve_at)
        if not self.h:
            raise ValueError("estimator rejected (empty k list?)")

    def update(self, ctx, v):
        self.lib.fo_kest_update(self.h, ctx, v)

    def get_k(self, ctx):
        return self.lib.fo_kest_get_k(self.h, ctx)

    def row(self, ctx):
        out = (C.c_uint32 * self.nk)()
        self.lib.fo_kest_row(self.h, ctx, out)
        return list(out)

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.fo_kest_free(self.h)
            self.h = None


_cached = None


def load():
    global _cached
    if _cached is None:
        _cached = Oracle(C.CDLL(build()))
    return _cached
