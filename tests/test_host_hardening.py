"""Host-side parsers of untrusted input under AddressSanitizer + UBSan (CPU build, `make asan`).

A mutated corpus -- truncated and bit-flipped .felics streams, TIFFs with a zero RowsPerStrip, huge tag
counts, strips that overlap or point outside the file, forged headers that claim gigapixel images -- goes
through imageio::read_image, felics_read_header, felics_decompress and felics_decompress_with_header in
the sanitizer build (felics_amd/csrc/host_fuzz.cpp).  Every input must be accepted or rejected with an
error code: no crash, no sanitizer report, no allocation sized by a forged header."""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
CSRC = os.path.join(ROOT, "felics_amd", "csrc")
FUZZ = os.path.join(ROOT, "felics_amd", "_build", "asan", "host_fuzz")


def _tiff(w, h, rps, strip_offsets, strip_counts=None, extra=(), data=b"", bits=8, le=True):
    """A minimal baseline TIFF with exactly the tag values given (valid or not)."""
    e = "<" if le else ">"
    nstr = len(strip_offsets)
    ents = [(256, 4, 1, w), (257, 4, 1, h), (258, 3, 1, bits), (259, 3, 1, 1), (262, 3, 1, 1), (277, 3, 1, 1),
            (278, 4, 1, rps)] + list(extra)
    body = b""
    base = 8 + len(data)
    ifd_at = base
    nent = len(ents) + 1 + (1 if strip_counts is not None else 0)
    after = ifd_at + 2 + nent * 12 + 4
    arrays = b""

    def arr(vals):
        nonlocal arrays
        off = after + len(arrays)
        arrays += b"".join(struct.pack(e + "I", v & 0xFFFFFFFF) for v in vals)
        return off

    if nstr == 1:
        ents.append((273, 4, 1, strip_offsets[0]))
    else:
        ents.append((273, 4, nstr, arr(strip_offsets)))
    if strip_counts is not None:
        ents.append((279, 4, len(strip_counts), strip_counts[0] if len(strip_counts) == 1 else arr(strip_counts)))
    ents.sort()
    body += struct.pack(e + "H", len(ents))
    for tag, typ, cnt, val in ents:
        body += struct.pack(e + "HHI", tag, typ, cnt & 0xFFFFFFFF)
        body += struct.pack(e + "HH", val & 0xFFFF, 0) if (typ == 3 and cnt == 1) else struct.pack(e + "I", val & 0xFFFFFFFF)
    body += struct.pack(e + "I", 0)
    head = (b"II" if le else b"MM") + struct.pack(e + "HI", 42, ifd_at)
    return head + data + body + arrays


def _corpus(dst):
    rng = np.random.default_rng(1234)
    n = 0

    def put(name, blob):
        nonlocal n
        with open(os.path.join(dst, "%04d_%s" % (n, name)), "wb") as f:
            f.write(blob)
        n += 1

    streams = sorted(f for f in os.listdir(GOLDEN) if f.endswith(".felics"))
    tiffs = sorted(f for f in os.listdir(GOLDEN) if f.endswith((".tiff", ".tif")))
    for name in streams[:4]:
        blob = open(os.path.join(GOLDEN, name), "rb").read()
        put(name, blob)
        for cut in (0, 3, 4, 5, 6, 13, 14, 15, 21, 22, 23, 30, len(blob) // 2, len(blob) - 1):
            put("cut%d_%s" % (cut, name), blob[:cut])
        for _ in range(25):  # bit flips: header fields and stream bits
            b = bytearray(blob)
            for _ in range(int(rng.integers(1, 4))):
                pos = int(rng.integers(0, min(len(b), 14 if rng.random() < 0.3 else len(b))))
                b[pos] ^= 1 << int(rng.integers(0, 8))
            put("flip_%s" % name, bytes(b))
        # forged dimensions on a short stream: must be refused before anything is allocated
        for w, h in ((0xFFFFFFFF, 0xFFFFFFFF), (65536, 65536), (1 << 20, 1 << 11), (0, 7), (1, 1), (2, 1)):
            put("dims_%s" % name, blob[:6] + struct.pack(">II", w, h) + blob[14:200])
        put("ones_%s" % name, blob[:22] + b"\xff" * 4096)  # endless unary run
        put("zeros_%s" % name, blob[:22] + b"\x00" * 4096)
    for name in tiffs[:3]:
        blob = open(os.path.join(GOLDEN, name), "rb").read()
        put(name, blob)
        for cut in (0, 2, 7, 8, 9, 100, len(blob) // 2, len(blob) - 1):
            put("cut%d_%s" % (cut, name), blob[:cut])
        for _ in range(25):
            b = bytearray(blob)
            pos = int(rng.integers(max(0, len(b) - 400), len(b)))  # the IFD of these files sits at the end
            b[pos] ^= 1 << int(rng.integers(0, 8))
            put("flip_%s" % name, bytes(b))
            b = bytearray(blob)
            b[int(rng.integers(0, 16))] ^= 1 << int(rng.integers(0, 8))
            put("fliphead_%s" % name, bytes(b))
    px = bytes(range(64))
    put("ok.tiff", _tiff(8, 8, 8, [8], data=px))
    put("rps0.tiff", _tiff(8, 8, 0, [8], data=px))                           # RowsPerStrip = 0 (was a division by zero)
    put("rps0_mm.tiff", _tiff(8, 8, 0, [8], data=px, le=False))
    put("rps_huge.tiff", _tiff(8, 8, 0xFFFFFFFF, [8], data=px))
    put("strips_overlap.tiff", _tiff(8, 8, 2, [8, 8, 8, 8], data=px))         # every strip at the same place
    put("strips_outside.tiff", _tiff(8, 8, 2, [8, 0x7FFFFFF0, 0xFFFFFFF0, 8], data=px))
    put("strips_short.tiff", _tiff(8, 8, 1, [8, 16], data=px))               # table shorter than the strip count
    put("dims_huge.tiff", _tiff(0xFFFFFFFF, 0xFFFFFFFF, 1, [8], data=px))
    put("dims_big.tiff", _tiff(1 << 16, 1 << 16, 1 << 16, [8], data=px))
    put("count_huge.tiff", _tiff(8, 8, 8, [8], data=px, extra=[(270, 2, 0xFFFFFFFF, 8)]))
    put("count_big_array.tiff", _tiff(8, 8, 8, [8] * 3, strip_counts=[1 << 30] * 3, data=px))
    put("bits_7.tiff", _tiff(8, 8, 8, [8], data=px, bits=7))
    put("bits_16_short.tiff", _tiff(8, 8, 8, [8], data=px, bits=16))
    put("ifd_loop.tiff", b"II*\x00\x08\x00\x00\x00" + b"\xff\xff")
    put("pnm_ok.pgm", b"P5\n4 4\n255\n" + bytes(16))
    put("pnm_short.pgm", b"P5\n4 4\n255\n" + bytes(15))
    put("pnm_huge.ppm", b"P6\n4294967295 4294967295\n65535\n")
    put("pnm_overflow.pgm", b"P5\n99999999999 1\n255\n")
    put("empty", b"")
    # PNG and compressed TIFF (zlib / LZW / PackBits paths of image_io.cpp): valid files, cuts, flips anywhere
    import io as _io

    from PIL import Image

    smooth = (np.add.outer(np.arange(40), np.arange(53)) * 5 % 256).astype(np.uint8)
    rgb = np.stack([smooth, smooth[::-1], 255 - smooth], -1).copy()
    seeds = []
    for arr, kw in ((smooth, {}), (rgb, {}), (smooth.astype(np.uint16) * 257, {})):
        b = _io.BytesIO()
        Image.fromarray(arr).save(b, format="PNG", **kw)
        seeds.append(("s.png", b.getvalue()))
    b = _io.BytesIO()
    Image.fromarray(rgb).quantize(colors=20).save(b, format="PNG")
    seeds.append(("pal.png", b.getvalue()))
    for comp in ("tiff_lzw", "tiff_adobe_deflate", "packbits"):
        for arr in (smooth, rgb):
            b = _io.BytesIO()
            Image.fromarray(arr).save(b, format="TIFF", compression=comp)
            seeds.append((comp + ".tiff", b.getvalue()))
    for name, blob in seeds:
        put(name, blob)
        for cut in (9, 20, 33, 34, 50, len(blob) // 2, len(blob) - 5, len(blob) - 1):
            put("cut%d_%s" % (cut, name), blob[:cut])
        for _ in range(30):
            bb = bytearray(blob)
            for _ in range(int(rng.integers(1, 3))):
                bb[int(rng.integers(0, len(bb)))] ^= 1 << int(rng.integers(0, 8))
            put("flip_" + name, bytes(bb))
    # PNGs whose chunks are intact (CRCs right) but whose content is not: bad filter bytes, too little / too much
    # data, flips inside the deflate stream, palette indices without a palette entry, absurd sizes
    import zlib as _z

    def chunk(t, b):
        return struct.pack(">I", len(b)) + t + b + struct.pack(">I", _z.crc32(t + b))

    def png(w, h, depth, ctype, raw, interlace=0, plte=None, z=None):
        body = _z.compress(raw) if z is None else z
        return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, interlace)) +
                (chunk(b"PLTE", plte) if plte is not None else b"") + chunk(b"IDAT", body) + chunk(b"IEND", b""))

    rows = b"".join(b"\x00" + bytes(range(16)) for _ in range(8))
    put("v_ok.png", png(16, 8, 8, 0, rows))
    put("v_filter9.png", png(16, 8, 8, 0, rows.replace(b"\x00\x00\x01", b"\x09\x00\x01", 1)))
    put("v_short.png", png(16, 8, 8, 0, rows[:-20]))
    put("v_long.png", png(16, 8, 8, 0, rows + bytes(500)))
    put("v_pal_noentry.png", png(16, 8, 8, 3, rows, plte=bytes(9)))
    put("v_pal_missing.png", png(16, 8, 8, 3, rows))
    put("v_depth3.png", png(16, 8, 3, 0, rows))
    put("v_huge.png", png(1 << 30, 1 << 30, 16, 6, rows))
    put("v_wide.png", png(1 << 31, 1, 8, 0, rows))
    put("v_zero.png", png(0, 8, 8, 0, rows))
    put("v_a7.png", png(16, 8, 8, 0, rows, interlace=1))
    put("v_a7_small.png", png(3, 2, 8, 0, b"\x00\x01" * 20, interlace=1))
    zgood = _z.compress(rows)
    for _ in range(40):
        zb = bytearray(zgood)
        zb[int(rng.integers(0, len(zb)))] ^= 1 << int(rng.integers(0, 8))
        put("v_zflip.png", png(16, 8, 8, 0, b"", z=bytes(zb)))
    for _ in range(20):
        rb = bytearray(rows)
        rb[int(rng.integers(0, 8)) * 17] = int(rng.integers(0, 256))  # a filter byte
        put("v_filt.png", png(16, 8, int(rng.choice([8, 16])), int(rng.choice([0, 2, 4, 6])), bytes(rb)))
    png = seeds[0][1]
    put("png_dims.png", png[:16] + struct.pack(">II", 1 << 28, 1 << 28) + png[24:])  # (CRC now wrong: refused there)
    return n


@pytest.fixture(scope="module")
def fuzz_binary():
    r = subprocess.run(["make", "-C", CSRC, "asan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return FUZZ


def test_mutated_corpus_under_sanitizers(tmp_path, fuzz_binary):
    d = tmp_path / "corpus"
    d.mkdir()
    n = _corpus(str(d))
    assert n > 700
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:allocator_may_return_null=1:max_allocation_size_mb=2048",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([fuzz_binary, str(d)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-6000:])
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-6000:]
    assert "host_fuzz: %d files" % n in r.stdout


def test_forged_header_is_refused_before_allocation():
    """A 22-byte file that claims 65535 x 65535 pixels: refused from the length alone (the advisory finding:
    decode_plane used to allocate W*H*4 bytes first)."""
    import felics_amd
    from felics_amd import api

    blob = b"FLCS\x00\x00" + struct.pack(">II", 65535, 65535) + bytes(8)
    out = np.zeros(16, np.uint8)
    rc = api.lib().felics_decompress(np.frombuffer(blob, np.uint8).ctypes.data, len(blob), out.ctypes.data, 1 << 40, None)
    assert rc == -1  # FELICS_E_IO: the stream cannot hold that many pixels
    rc = api.lib().felics_decompress(np.frombuffer(blob, np.uint8).ctypes.data, len(blob), out.ctypes.data, 16, None)
    assert rc == -8  # FELICS_E_BUFFER_TOO_SMALL comes first when the caller's buffer is the smaller bound
    with pytest.raises(felics_amd.DecompressionError):
        api.decompress_bytes(blob)


def test_decompress_with_header_matches_decompress(oracle):
    """traits.rs:53-56: decompress_with_header(from, &Header) == decompress(from) with the header read first."""
    import io

    import felics_amd
    from felics_amd import api

    for name in sorted(f for f in os.listdir(GOLDEN) if f.endswith(".felics")):
        blob = open(os.path.join(GOLDEN, name), "rb").read()
        hdr = felics_amd.read_header(io.BytesIO(blob[:14]))
        a = api.decompress_with_header(io.BytesIO(blob[14:]), hdr)
        b = felics_amd.decompress_image(io.BytesIO(blob))
        assert a.shape == b.shape and a.dtype == b.dtype and (a == b).all()
        assert (a == oracle.decompress(blob)).all()
    # a header that disagrees with the stream: error, not a crash
    hdr = felics_amd.Header(0, 0, 4000, 4000)
    with pytest.raises(felics_amd.DecompressionError):
        api.decompress_with_header(io.BytesIO(b"\x00" * 64), hdr)
