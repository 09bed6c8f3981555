"""CPU-side checks of the product: the C-ABI library loads and exports every symbol include/felics.h
declares, the host decoder and header functions agree with the oracle and the committed fixtures,
the command lines behave like the reference's.  No GPU compute is attempted here."""
import ctypes as C
import io
import json
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
BUILD = os.path.join(ROOT, "felics_amd", "_build")


@pytest.fixture(scope="module")
def api():
    from felics_amd import api as a

    a.lib()
    return a


def test_library_exports_every_declared_symbol(api):
    header = open(os.path.join(ROOT, "include", "felics.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(felics_[a-z_]+)\s*\(", header))
    assert len(declared) >= 15
    lib = C.CDLL(os.path.join(BUILD, "libfelics.so"))
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    assert declared == set(api.EXPORTS)


def test_no_gpu_means_loud_failure(api):
    """Without a HIP device the encoder must refuse; there is no CPU encode path to fall back to."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import felics_amd

    with pytest.raises(felics_amd.FelicsError) as ei:
        felics_amd.Encoder(0)
    assert ei.value.code == -9
    with pytest.raises(felics_amd.FelicsError):
        felics_amd.compress_image(io.BytesIO(), np.zeros((4, 4), np.uint8))


def test_strerror_and_sizes(api):
    L = api.lib()
    for code in range(0, -12, -1):
        assert L.felics_strerror(code)
    assert L.felics_max_compressed_size(0, 5, 0, 0) == 14 + 8
    assert L.felics_max_compressed_size(3, 2, 0, 0) == 14 + (64 + 6 * 257 + 7) // 8
    assert L.felics_max_compressed_size(3, 2, 1, 0) == 14 + (3 * 64 + 18 * 512 + 7) // 8
    assert L.felics_stage_count() <= 16 and L.felics_lane_count() >= 1


def test_header_roundtrip_and_errors(api):
    """format.rs:51-84."""
    import felics_amd as F

    hdr = F.Header(F.ColorType.Rgb, F.PixelDepth.Sixteen, 0x01020304, 7)
    buf = io.BytesIO()
    F.write_header(hdr, buf)
    raw = buf.getvalue()
    assert raw == b"FLCS\x01\x01\x01\x02\x03\x04\x00\x00\x00\x07"
    assert F.read_header(io.BytesIO(raw)) == hdr
    for bad, kind in ((b"FLCX" + raw[4:], "InvalidSignature"), (raw[:4] + b"\x02" + raw[5:], "InvalidColorType"),
                      (raw[:5] + b"\x02" + raw[6:], "InvalidPixelDepth"), (raw[:10], "IoError"), (b"", "IoError")):
        with pytest.raises(F.DecompressionError) as ei:
            F.read_header(io.BytesIO(bad))
        assert ei.value.kind == kind


def test_decoder_matches_oracle_streams(api, oracle):
    """felics_decompress (the product's host decoder) on streams produced by the oracle encoder."""
    import felics_amd as F

    rng = np.random.default_rng(5)
    shapes = [(2, 1), (1, 2), (1, 1), (4, 7), (40, 100), (274, 124), (8, 1447), (1, 44), (100, 1), (0, 3), (3, 0)]
    for h, w in shapes:
        for dt in (np.uint8, np.uint16):
            for shape in ((h, w), (h, w, 3)):
                img = rng.integers(0, np.iinfo(dt).max + 1, size=shape, dtype=dt)
                out = F.decompress_image(io.BytesIO(oracle.compress(img)))
                assert out.dtype == dt and out.shape == img.shape and (out == img).all()


def test_decoder_on_golden_fixtures(api):
    from PIL import Image

    import felics_amd as F

    pins = json.load(open(os.path.join(GOLDEN, "pins.json")))
    for name in pins["files"]:
        img = np.array(Image.open(os.path.join(GOLDEN, name)))
        out = F.decompress_image(open(os.path.join(GOLDEN, name + ".felics"), "rb"))
        assert out.dtype == img.dtype and (out == img).all(), name


def test_decoder_rejects_corrupt_streams(api, oracle):
    import felics_amd as F

    good = oracle.compress(np.arange(64, dtype=np.uint8).reshape(8, 8))
    with pytest.raises(F.DecompressionError) as ei:
        F.decompress_image(io.BytesIO(good[:-3]))
    assert ei.value.kind == "IoError"
    # first pixel far outside 8 bits: a context above MAX_CONTEXT or a value that does not fit u8
    bad = good[:14] + (70000).to_bytes(4, "big") + good[18:]
    with pytest.raises(F.DecompressionError) as ei:
        F.decompress_image(io.BytesIO(bad))
    assert ei.value.kind in ("InvalidValue", "ValueOverflow", "IoError")


# ---- command lines -------------------------------------------------------------------------

def _run(tool, *args):
    return subprocess.run([os.path.join(BUILD, tool), *args], capture_output=True, text=True)


def test_dfelics_writes_tiff_and_pnm(tmp_path):
    """src/bin/dfelics.rs: the output format follows the output extension."""
    from PIL import Image

    for name in ("6.3.09.tiff", "house.tiff", "aerial.tiff"):
        src = np.array(Image.open(os.path.join(GOLDEN, name)))
        for ext in ("tiff", "ppm" if src.ndim == 3 else "pgm"):
            out = str(tmp_path / (name + "." + ext))
            r = _run("dfelics", "-i", os.path.join(GOLDEN, name + ".felics"), "--output", out)
            assert r.returncode == 0, r.stdout + r.stderr
            back = np.array(Image.open(out))
            assert back.shape == src.shape and (back == src).all(), (name, ext)


def test_cli_messages_and_exit_codes(tmp_path):
    r = _run("dfelics", "-i", str(tmp_path / "missing.felics"), "-o", str(tmp_path / "x.tiff"))
    assert r.returncode == 1 and r.stdout.startswith("Cannot open input file:")  # dfelics.rs:29
    bad = tmp_path / "bad.felics"
    bad.write_bytes(b"NOPE" + bytes(20))
    r = _run("dfelics", "-i", str(bad), "-o", str(tmp_path / "x.tiff"))
    assert r.returncode == 1 and r.stdout.strip() == "Error while decompressing the image: InvalidSignature"
    r = _run("dfelics", "-i", os.path.join(GOLDEN, "6.3.09.tiff.felics"), "-o", str(tmp_path / "x.jpg"))
    assert r.returncode == 1 and r.stdout.startswith("Cannot save image:")  # dfelics.rs:55
    r = _run("cfelics", "-i", str(tmp_path / "missing.tiff"), "-o", str(tmp_path / "x.felics"))
    assert r.returncode == 1 and r.stdout.startswith("Cannot open file:")  # cfelics.rs:39
    junk = tmp_path / "junk.tiff"
    junk.write_bytes(b"II*\x00" + bytes(4))
    r = _run("cfelics", "-i", str(junk), "-o", str(tmp_path / "x.felics"))
    assert r.returncode == 1 and r.stdout.startswith("Cannot decode image:")  # cfelics.rs:47
    for tool in ("cfelics", "dfelics"):
        assert _run(tool, "--help").returncode == 0
        assert _run(tool, "-V").stdout.startswith(tool)
        assert _run(tool).returncode == 2  # clap: missing required arguments
        assert _run(tool, "--bogus").returncode == 2


def test_cfelics_without_gpu_refuses(tmp_path):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = _run("cfelics", "-i", os.path.join(GOLDEN, "6.3.09.tiff"), "-o", str(tmp_path / "x.felics"))
    assert r.returncode == 1
    lines = r.stdout.strip().splitlines()
    assert lines[0] == "Compressing 8-bit grayscale image..."  # cfelics.rs:54
    assert lines[1].startswith("Cannot compress image:")  # cfelics.rs:76
    assert not (tmp_path / "x.felics").exists()


def test_synthetic_generators_agree():
    """numpy and torch restatements of the BASELINE.md generator are identical."""
    from felics_amd import synth, synth_torch

    for kind in ("S1", "S2", "S3"):
        a = synth.gray8(257, 33, 3, kind)
        b = synth_torch.gray8(257, 33, 3, kind, device="cpu").numpy()
        assert (a == b).all()
    assert (synth.rgb8(130, 17, 2) == synth_torch.rgb8(130, 17, 2, device="cpu").numpy()).all()
    # spot values straight from the formula
    assert synth.gray8(8, 1, 0, "S3")[0, 0] == 128
