"""GPU decoder (felics_decompress_batch_device, src/compression.rs:151-248 of the reference): its pixels against the
oracle's decoder and the original images -- the reference's own round-trip strategy (compression.rs:456-558,
tests/compress.rs) with the streams produced by the GPU encoder and by the oracle, plus the committed golden
streams, truncated / corrupted streams (error.rs:4-19) and a full-size batch."""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def enc():
    import felics_amd

    e = felics_amd.Encoder(0)
    yield e
    e.close()


def _decode_batch(enc, streams, shape, dtype):
    """streams: list of bytes of one shape -> list of arrays decoded by the GPU decoder."""
    import torch

    n = len(streams)
    offs, blob, at = [], bytearray(), 0
    for s in streams:
        offs.append(at)
        blob += s
        pad = (-len(blob)) % 16
        blob += bytes(pad)
        at = len(blob)
    lens = [len(s) for s in streams]
    d_in = torch.from_numpy(np.frombuffer(bytes(blob) + bytes(16), dtype=np.uint8).copy()).cuda()
    per = int(np.prod(shape)) * np.dtype(dtype).itemsize
    d_px = torch.zeros(max(per * n, 16), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    hdr, status = enc.decompress_batch_device(d_in.data_ptr(), offs, lens, d_px.data_ptr(), per * n)
    assert (status == 0).all()
    host = d_px.cpu().numpy()
    return hdr, [host[i * per:(i + 1) * per].view(dtype).reshape(shape) for i in range(n)]


DIMS = [(2, 1), (1, 2), (1, 1), (4, 7), (100, 40), (124, 274), (1447, 8), (44, 1), (1, 100), (680, 480), (3, 3), (0, 5), (5, 0)]


def test_roundtrip_reference_shapes(enc, oracle):
    """compression.rs:500-558: random u8 / u16 gray and rgb at the reference's shapes (16-bit streams on the device too:
    k_decode16, the estimator table in HBM behind an LDS cache)."""
    rng = np.random.default_rng(5)
    for w, h in DIMS:
        for shape, dt, mx in (((h, w), np.uint8, 256), ((h, w, 3), np.uint8, 256), ((h, w), np.uint16, 65536), ((h, w, 3), np.uint16, 65536)):
            imgs = [rng.integers(0, mx, size=shape).astype(dt) for _ in range(3)]
            smooth = (np.add.outer(np.arange(h), np.arange(w)) // 2 % mx).astype(dt)
            imgs.append(smooth if len(shape) == 2 else np.stack([smooth, smooth[::-1], (mx - 1) - smooth], -1).copy())
            streams = [oracle.compress(im) for im in imgs]
            hdr, back = _decode_batch(enc, streams, shape, dt)
            assert (hdr.width, hdr.height) == (w, h)
            for im, b, s in zip(imgs, back, streams):
                assert (b == im).all(), (w, h, shape, dt)
                assert (b == oracle.decompress(s)).all()


def test_every_small_shape(enc, oracle):
    """compression.rs:544-558: every w, h below 12, gray and rgb, one batch per shape."""
    rng = np.random.default_rng(6)
    for w in range(1, 12):
        for h in range(1, 12):
            for shape in ((h, w), (h, w, 3)):
                imgs = [rng.integers(0, 256, size=shape, dtype=np.uint8) for _ in range(2)]
                _, back = _decode_batch(enc, [oracle.compress(im) for im in imgs], shape, np.uint8)
                assert all((b == im).all() for b, im in zip(back, imgs)), (w, h, shape)


def test_extreme_content_and_golden(enc, oracle):
    from PIL import Image

    rng = np.random.default_rng(7)
    h, w = 96, 257
    flat = np.full((h, w), 7, np.uint8)
    spikes = flat.copy()
    spikes[rng.integers(0, h, 200), rng.integers(0, w, 200)] = 255  # long unary runs at k = 0
    checker = ((np.indices((h, w)).sum(0) & 1) * 255).astype(np.uint8)
    ramp = (np.arange(w)[None, :] + np.arange(h)[:, None]).astype(np.uint8)
    imgs = [flat, spikes, checker, ramp]
    _, back = _decode_batch(enc, [oracle.compress(im) for im in imgs], (h, w), np.uint8)
    assert all((b == im).all() for b, im in zip(back, imgs))
    rgb = [np.stack([im, im[::-1], 255 - im], -1).copy() for im in imgs]
    _, back = _decode_batch(enc, [oracle.compress(im) for im in rgb], (h, w, 3), np.uint8)
    assert all((b == im).all() for b, im in zip(back, rgb))
    for p in sorted(glob.glob(os.path.join(GOLDEN, "*.felics"))):
        img = np.array(Image.open(p[:-len(".felics")]))
        _, back = _decode_batch(enc, [open(p, "rb").read()], img.shape, img.dtype)
        assert (back[0] == img).all(), p


def test_corrupt_streams_give_error_codes(enc, oracle):
    """error.rs:4-19: truncated streams, bad signature / colour / depth, streams of another shape, bit flips.
    Every stream of the batch gets a status; nothing hangs or faults."""
    import felics_amd
    import torch

    rng = np.random.default_rng(8)
    img = rng.integers(0, 256, size=(60, 70), dtype=np.uint8)
    good = oracle.compress(img)
    other = oracle.compress(rng.integers(0, 256, size=(61, 70), dtype=np.uint8))
    bad = [good[: len(good) // 2], good[:20], b"XLCS" + good[4:], good[:4] + b"\x07" + good[5:], good[:5] + b"\x09" + good[6:], other,
           good[:30] + b"\xff" * (len(good) - 30), good[:30] + bytes(len(good) - 30)]
    for _ in range(12):
        b = bytearray(good)
        b[int(rng.integers(14, len(b)))] ^= 1 << int(rng.integers(0, 8))
        bad.append(bytes(b))
    streams = [good] + bad + [good]
    offs, blob = [], bytearray()
    for s in streams:
        offs.append(len(blob))
        blob += s + bytes((-len(s)) % 16)
    d_in = torch.from_numpy(np.frombuffer(bytes(blob) + bytes(16), dtype=np.uint8).copy()).cuda()
    d_px = torch.zeros(img.size * len(streams), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    with pytest.raises(felics_amd.DecompressionError) as ei:
        enc.decompress_batch_device(d_in.data_ptr(), offs, [len(s) for s in streams], d_px.data_ptr(), d_px.numel())
    status = ei.value.status
    assert status[0] == 0 and status[-1] == 0
    host = d_px.cpu().numpy()
    assert (host[: img.size].reshape(img.shape) == img).all() and (host[-img.size:].reshape(img.shape) == img).all()
    assert status[1] == -1 and status[2] == -1          # truncated: IoError
    assert status[3] == -7 and status[4] == -5 and status[5] == -6 and status[6] == -4
    for i, s in enumerate(streams[1:-1], start=1):
        # a flipped bit may still decode to some image; if the host decoder rejects the stream the GPU decoder must too
        try:
            want = oracle.decompress(s)
            host_ok = want.shape == img.shape
        except Exception:
            host_ok = False
        if not host_ok:
            assert status[i] != 0, i
        elif status[i] == 0:
            assert (host[i * img.size:(i + 1) * img.size].reshape(img.shape) == want).all(), i


def test_batch_from_the_gpu_encoder(enc, oracle):
    """Encode on the GPU, decode on the GPU, without the streams leaving HBM: 24 1080p gray frames and 12 RGB frames."""
    import torch
    from felics_amd import synth_torch

    for rgb in (False, True):
        n, w, h = (12, 1280, 720) if rgb else (24, 1920, 1080)
        ch = 3 if rgb else 1
        frames = torch.stack([synth_torch.rgb8(w, h, f) if rgb else synth_torch.gray8(w, h, f, "S1" if f % 5 else "S2") for f in range(n)])
        cap = int(n * w * h * ch * 1.3) + (1 << 20)
        d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        offs, lens = enc.compress_batch_device(frames.data_ptr(), n, w, h, int(rgb), 0, d_out.data_ptr(), cap)
        d_px = torch.zeros_like(frames)
        hdr, status = enc.decompress_batch_device(d_out.data_ptr(), offs, lens, d_px.data_ptr(), d_px.numel())
        assert (status == 0).all() and (hdr.width, hdr.height, int(hdr.color_type)) == (w, h, int(rgb))
        assert bool((d_px == frames).all())
        host = d_out[int(offs[3]): int(offs[3] + lens[3])].cpu().numpy().tobytes()
        assert (oracle.decompress(host) == frames[3].cpu().numpy()).all()


def test_baseline_batch_decodes(enc):
    """BASELINE config 3's streams (64 synthetic 4K gray frames) decoded back on the GPU: the size-independent
    round-trip property at full size."""
    import torch
    from felics_amd import synth_torch

    n, w, h = 64, 3840, 2160
    frames = torch.stack([synth_torch.gray8(w, h, f, "S1") for f in range(n)])
    cap = int(n * w * h * 1.25) + (1 << 20)
    d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    offs, lens = enc.compress_batch_device(frames.data_ptr(), n, w, h, 0, 0, d_out.data_ptr(), cap)
    d_px = torch.zeros_like(frames)
    _, status = enc.decompress_batch_device(d_out.data_ptr(), offs, lens, d_px.data_ptr(), d_px.numel())
    assert (status == 0).all() and bool((d_px == frames).all())



def test_sixteen_bit_streams_on_the_device(enc, oracle):
    """16-bit streams never leave the GPU: natural 16-bit files (the golden ones, 256 x 256 and 1081 x 1081), synthetic gray16 and
    rgb16 frames with more contexts than the LDS cache has rows (evictions, write-backs, reloads), extreme content (65 534-bit
    codes), two calls in a row on the same tables (epochs), corrupt streams -- against the original pixels and the host decoder."""
    import felics_amd
    import torch
    from PIL import Image
    from felics_amd import synth

    rng = np.random.default_rng(12)
    for name in ("aerial.tiff", "man.tiff", "heightmap.tiff"):
        img = np.array(Image.open(os.path.join(GOLDEN, name)))
        assert img.dtype == np.uint16
        stream = open(os.path.join(GOLDEN, name + ".felics"), "rb").read()
        for _ in range(2):  # the second call finds the first one's rows in the tables: another epoch
            _, back = _decode_batch(enc, [stream, stream], img.shape, np.uint16)
            assert (back[0] == img).all() and (back[1] == img).all(), name
    frames = [synth.gray16(640, 360, f) for f in range(5)] + [rng.integers(0, 65536, size=(360, 640), dtype=np.uint16)]
    _, back = _decode_batch(enc, [oracle.compress(f) for f in frames], (360, 640), np.uint16)
    assert all((b == f).all() for b, f in zip(back, frames))
    rgb = [np.stack([f, np.roll(f, 3, axis=1), 65535 - f], -1).copy() for f in frames[:3]] + [rng.integers(0, 65536, size=(360, 640, 3), dtype=np.uint16)]
    _, back = _decode_batch(enc, [oracle.compress(f) for f in rgb], (360, 640, 3), np.uint16)
    assert all((b == f).all() for b, f in zip(back, rgb))
    quiet = np.full((40, 300), 9, np.uint16)
    spikes = quiet.copy()
    spikes[rng.integers(0, 40, 60), rng.integers(0, 300, 60)] = 65535  # unary runs of 65 525 ones at k = 0
    _, back = _decode_batch(enc, [oracle.compress(quiet), oracle.compress(spikes)], (40, 300), np.uint16)
    assert (back[0] == quiet).all() and (back[1] == spikes).all()
    # corrupt 16-bit streams: a status for every stream, the good ones intact
    img = frames[0][:50, :70].copy()
    good = oracle.compress(img)
    bad = [good[: len(good) // 2], good[:4] + b"\x07" + good[5:], good[:30] + b"\xff" * (len(good) - 30)]
    for _ in range(8):
        b = bytearray(good)
        b[int(rng.integers(14, len(b)))] ^= 1 << int(rng.integers(0, 8))
        bad.append(bytes(b))
    streams = [good] + bad + [good]
    offs, blob = [], bytearray()
    for s_ in streams:
        offs.append(len(blob))
        blob += s_ + bytes((-len(s_)) % 16)
    d_in = torch.from_numpy(np.frombuffer(bytes(blob) + bytes(16), dtype=np.uint8).copy()).cuda()
    d_px = torch.zeros(img.size * 2 * len(streams), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    with pytest.raises(felics_amd.DecompressionError) as ei:
        enc.decompress_batch_device(d_in.data_ptr(), offs, [len(s_) for s_ in streams], d_px.data_ptr(), d_px.numel())
    status = ei.value.status
    host = d_px.cpu().numpy().view(np.uint16)
    assert status[0] == 0 and status[-1] == 0 and status[1] == -1 and status[2] == -5
    assert (host[: img.size].reshape(img.shape) == img).all() and (host[-img.size:].reshape(img.shape) == img).all()
    for i, s_ in enumerate(streams[1:-1], start=1):
        try:
            want = oracle.decompress(s_)
            host_ok = want.shape == img.shape and want.dtype == np.uint16
        except Exception:
            host_ok = False
        if not host_ok:
            assert status[i] != 0, i
        elif status[i] == 0:
            assert (host[i * img.size:(i + 1) * img.size].reshape(img.shape) == want).all(), i


def _forced(value):
    """context manager: FELICS_TEST_DECODE_LANES for the calls inside (read per call by the library)"""
    import contextlib

    @contextlib.contextmanager
    def cm():
        os.environ["FELICS_TEST_DECODE_LANES"] = value
        try:
            yield
        finally:
            del os.environ["FELICS_TEST_DECODE_LANES"]
    return cm()


def test_lane_decoder_shapes_and_content(enc, oracle):
    """k_decode8_lanes (64 gray streams per wave, lane = stream), forced on small batches: the reference's shapes with W >= 8,
    odd widths (the read-back of the row above is four samples wide, rows end anywhere), extreme content (flat, checker,
    ramps, spikes: codes of hundreds of bits), noise (every context, the estimator rows in HBM) -- against the original
    pixels and the host decoder's."""
    from felics_amd import synth

    rng = np.random.default_rng(21)
    with _forced("1"):
        for w, h in [(8, 1), (9, 3), (10, 7), (11, 2), (63, 5), (64, 4), (65, 9), (100, 40), (124, 274), (1447, 8), (680, 480)]:
            imgs = [rng.integers(0, 256, size=(h, w)).astype(np.uint8) for _ in range(3)]
            imgs.append((np.add.outer(np.arange(h), np.arange(w)) // 2 % 256).astype(np.uint8))
            imgs.append(np.full((h, w), 200, np.uint8))
            imgs.append(((np.add.outer(np.arange(h), np.arange(w)) & 1) * 255).astype(np.uint8))
            spikes = rng.integers(100, 104, size=(h, w)).astype(np.uint8)
            spikes[rng.random((h, w)) < 0.02] = 255
            spikes[rng.random((h, w)) < 0.02] = 0
            imgs.append(spikes)
            imgs.append(synth.gray8(w, h, 3, "S1"))
            streams = [oracle.compress(im) for im in imgs]
            hdr, back = _decode_batch(enc, streams, (h, w), np.uint8)
            assert (hdr.width, hdr.height) == (w, h)
            for b, im, s in zip(back, imgs, streams):
                assert (b == im).all(), (w, h)
                assert (b == oracle.decompress(s)).all()
        # more streams than one wave holds, a ragged last wave
        imgs = [synth.gray8(96, 33, f, "S2" if f % 3 == 0 else "S1") for f in range(130)]
        _, back = _decode_batch(enc, [oracle.compress(im) for im in imgs], (33, 96), np.uint8)
        assert all((b == im).all() for b, im in zip(back, imgs))


def test_lane_decoder_corrupt_streams(enc, oracle):
    """The corrupt-stream list of test_corrupt_streams_give_error_codes through the lane decoder: every stream gets a status,
    the good ones beside the bad ones in the same wave decode intact, nothing faults."""
    import felics_amd
    import torch

    rng = np.random.default_rng(9)
    img = rng.integers(0, 256, size=(60, 70), dtype=np.uint8)
    good = oracle.compress(img)
    other = oracle.compress(rng.integers(0, 256, size=(61, 70), dtype=np.uint8))
    bad = [good[: len(good) // 2], good[:20], b"XLCS" + good[4:], good[:4] + b"\x07" + good[5:], good[:5] + b"\x09" + good[6:], other,
           good[:30] + b"\xff" * (len(good) - 30), good[:30] + bytes(len(good) - 30)]
    for _ in range(40):
        b = bytearray(good)
        b[int(rng.integers(14, len(b)))] ^= 1 << int(rng.integers(0, 8))
        bad.append(bytes(b))
    streams = [good] + bad + [good]
    offs, blob = [], bytearray()
    for s in streams:
        offs.append(len(blob))
        blob += s + bytes((-len(s)) % 16)
    d_in = torch.from_numpy(np.frombuffer(bytes(blob) + bytes(16), dtype=np.uint8).copy()).cuda()
    d_px = torch.zeros(img.size * len(streams), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    with _forced("1"), pytest.raises(felics_amd.DecompressionError) as ei:
        enc.decompress_batch_device(d_in.data_ptr(), offs, [len(s) for s in streams], d_px.data_ptr(), d_px.numel())
    status = ei.value.status
    assert status[0] == 0 and status[-1] == 0
    host = d_px.cpu().numpy()
    assert (host[: img.size].reshape(img.shape) == img).all() and (host[-img.size:].reshape(img.shape) == img).all()
    assert status[1] == -1 and status[2] == -1          # truncated: IoError
    assert status[3] == -7 and status[4] == -5 and status[5] == -6 and status[6] == -4
    for i, s in enumerate(streams[1:-1], start=1):
        try:
            want = oracle.decompress(s)
            host_ok = want.shape == img.shape
        except Exception:
            host_ok = False
        if not host_ok:
            assert status[i] != 0, i
        elif status[i] == 0:
            assert (host[i * img.size:(i + 1) * img.size].reshape(img.shape) == want).all(), i


def test_lane_decoder_rgb(enc, oracle):
    """The lane form on RGB8 streams (three planes from one bit reader per lane, int16 Y / Co / Cg planes, contexts up to 510, a table
    per plane): shapes with W >= 8 and odd widths, noise (every context; Co / Cg at their extremes), flat, checker, primaries with
    spikes, synthetic frames, a ragged last wave -- against the original pixels and the host decoder's; then the corrupt-stream list:
    every stream gets a status, good streams beside bad ones in the same wave decode intact."""
    import felics_amd
    import torch
    from felics_amd import synth

    rng = np.random.default_rng(77)
    with _forced("1"):
        for w, h in [(8, 1), (9, 3), (11, 2), (63, 5), (64, 4), (65, 9), (100, 40), (124, 74), (1447, 8), (320, 240)]:
            imgs = [rng.integers(0, 256, size=(h, w, 3)).astype(np.uint8) for _ in range(3)]
            g = (np.add.outer(np.arange(h), np.arange(w)) // 2 % 256).astype(np.uint8)
            imgs.append(np.stack([g, g[::-1], 255 - g], -1).copy())
            imgs.append(np.full((h, w, 3), (255, 0, 255), np.uint8))                                  # Co, Cg at their extremes
            chk = ((np.add.outer(np.arange(h), np.arange(w)) & 1) * 255).astype(np.uint8)
            imgs.append(np.stack([chk, 255 - chk, chk], -1).copy())                                   # full-scale swings in every plane
            spikes = rng.integers(100, 104, size=(h, w, 3)).astype(np.uint8)
            spikes[rng.random((h, w)) < 0.02] = (255, 0, 0)
            spikes[rng.random((h, w)) < 0.02] = (0, 255, 255)
            imgs.append(spikes)
            imgs.append(synth.rgb8(w, h, 2))
            streams = [oracle.compress(im) for im in imgs]
            hdr, back = _decode_batch(enc, streams, (h, w, 3), np.uint8)
            assert (hdr.width, hdr.height, hdr.color_type) == (w, h, 1)
            for b, im, s in zip(back, imgs, streams):
                assert (b == im).all(), (w, h)
                assert (b == oracle.decompress(s)).all()
        imgs = [synth.rgb8(96, 33, f) if f % 3 else rng.integers(0, 256, size=(33, 96, 3)).astype(np.uint8) for f in range(130)]
        _, back = _decode_batch(enc, [oracle.compress(im) for im in imgs], (33, 96, 3), np.uint8)
        assert all((b == im).all() for b, im in zip(back, imgs))
        # corrupt streams
        img = rng.integers(0, 256, size=(40, 50, 3), dtype=np.uint8)
        good = oracle.compress(img)
        gray = oracle.compress(img[:, :, 0].copy())
        bad = [good[: len(good) // 2], good[:20], b"XLCS" + good[4:], good[:4] + b"\x07" + good[5:], good[:5] + b"\x09" + good[6:], gray,
               good[:30] + b"\xff" * (len(good) - 30), good[:30] + bytes(len(good) - 30)]
        for _ in range(40):
            b = bytearray(good)
            b[int(rng.integers(14, len(b)))] ^= 1 << int(rng.integers(0, 8))
            bad.append(bytes(b))
        streams = [good] + bad + [good]
        offs, blob = [], bytearray()
        for s in streams:
            offs.append(len(blob))
            blob += s + bytes((-len(s)) % 16)
        d_in = torch.from_numpy(np.frombuffer(bytes(blob) + bytes(16), dtype=np.uint8).copy()).cuda()
        d_px = torch.zeros(img.size * len(streams), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        with pytest.raises(felics_amd.DecompressionError) as ei:
            enc.decompress_batch_device(d_in.data_ptr(), offs, [len(s) for s in streams], d_px.data_ptr(), d_px.numel())
        status = ei.value.status
        assert status[0] == 0 and status[-1] == 0
        host = d_px.cpu().numpy()
        assert (host[: img.size].reshape(img.shape) == img).all() and (host[-img.size:].reshape(img.shape) == img).all()
        assert status[1] == -1 and status[2] == -1          # truncated: IoError
        assert status[3] == -7 and status[4] == -5 and status[5] == -6 and status[6] == -4
        for i, s in enumerate(streams[1:-1], start=1):
            try:
                want = oracle.decompress(s)
                host_ok = want.shape == img.shape
            except Exception:
                host_ok = False
            if not host_ok:
                assert status[i] != 0, i
            elif status[i] == 0:
                assert (host[i * img.size:(i + 1) * img.size].reshape(img.shape) == want).all(), i


def test_three_hundred_mixed_streams(enc, oracle):
    """Both forms of the device decoder (one wave per stream; 64 streams per wave, which the library picks from 1536 gray / 2048 RGB
    streams up) on 300 streams of one shape and mixed content -- synthetic S1 / S2 / S3 frames, ramps with spikes, crops of the golden natural images -- encoded on the
    GPU, decoded back on the GPU and compared with the frames; some of the streams also through the oracle's decoder."""
    import torch
    from felics_amd import synth

    w, h = 200, 96
    rng = np.random.default_rng(33)
    nat = []
    from PIL import Image

    for path in sorted(glob.glob(os.path.join(GOLDEN, "*.tif*"))):
        if path.endswith(".felics"):
            continue
        im = np.array(Image.open(path))
        if im.dtype == np.uint8 and im.ndim == 2 and im.shape[0] >= h and im.shape[1] >= w:
            nat.append(im)
    assert nat
    frames = []
    for f in range(300):
        kind = f % 5
        if kind == 0:
            frames.append(synth.gray8(w, h, f, "S1"))
        elif kind == 1:
            frames.append(synth.gray8(w, h, f, "S2"))
        elif kind == 2:
            frames.append(synth.gray8(w, h, f, "S3") if f % 2 else (np.add.outer(np.arange(h), np.arange(w)) * (f % 7 + 1) // 3 % 256).astype(np.uint8))
        elif kind == 3 and nat:
            im = nat[f % len(nat)]
            y0, x0 = int(rng.integers(0, im.shape[0] - h + 1)), int(rng.integers(0, im.shape[1] - w + 1))
            frames.append(np.ascontiguousarray(im[y0:y0 + h, x0:x0 + w]))
        else:
            sp = rng.integers(60, 64, size=(h, w)).astype(np.uint8)
            sp[rng.random((h, w)) < 0.01] = 255
            frames.append(sp)
    d_in = torch.from_numpy(np.stack(frames)).cuda()
    cap = len(frames) * (w * h * 2 + 64)
    d_out = torch.zeros(cap, dtype=torch.uint8, device="cuda")
    offs, lens = enc.compress_batch_device(d_in.data_ptr(), len(frames), w, h, 0, 0, d_out.data_ptr(), cap)
    for form in ("0", "1"):
        d_px = torch.zeros_like(d_in)
        with _forced(form):
            hdr, status = enc.decompress_batch_device(d_out.data_ptr(), offs, lens, d_px.data_ptr(), d_px.numel())
        assert (status == 0).all() and (hdr.width, hdr.height) == (w, h)
        assert bool((d_px == d_in).all()), form
    # the library's own choice for a batch of thousands of streams (the 300, referenced seven times over)
    offs7, lens7 = np.tile(np.asarray(offs, dtype=np.uint64), 7), np.tile(np.asarray(lens, dtype=np.uint64), 7)
    d_px7 = torch.zeros((7,) + tuple(d_in.shape), dtype=torch.uint8, device="cuda")
    hdr, status = enc.decompress_batch_device(d_out.data_ptr(), offs7, lens7, d_px7.data_ptr(), d_px7.numel())
    assert (status == 0).all() and all(bool((d_px7[r] == d_in).all()) for r in range(7))
    host = d_out.cpu().numpy()
    for i in (0, 1, 63, 64, 255, 299):
        assert (oracle.decompress(host[int(offs[i]): int(offs[i] + lens[i])].tobytes()) == frames[i]).all(), i
    # RGB8: 150 frames through both forms, then 2 100 streams (the 150, fourteen times over) through the library's own choice
    wc, hc = 96, 40
    rgb = [synth.rgb8(wc, hc, f) if f % 3 else rng.integers(0, 256, size=(hc, wc, 3)).astype(np.uint8) for f in range(150)]
    d_rgb = torch.from_numpy(np.stack(rgb)).cuda()
    capc = len(rgb) * (wc * hc * 3 * 2 + 64)
    d_outc = torch.zeros(capc, dtype=torch.uint8, device="cuda")
    offc, lenc = enc.compress_batch_device(d_rgb.data_ptr(), len(rgb), wc, hc, 1, 0, d_outc.data_ptr(), capc)
    for form in ("0", "1"):
        d_pxc = torch.zeros_like(d_rgb)
        with _forced(form):
            hdr, status = enc.decompress_batch_device(d_outc.data_ptr(), offc, lenc, d_pxc.data_ptr(), d_pxc.numel())
        assert (status == 0).all() and (hdr.width, hdr.height, hdr.color_type) == (wc, hc, 1)
        assert bool((d_pxc == d_rgb).all()), form
    off14, len14 = np.tile(np.asarray(offc, dtype=np.uint64), 14), np.tile(np.asarray(lenc, dtype=np.uint64), 14)
    d_px14 = torch.zeros((14,) + tuple(d_rgb.shape), dtype=torch.uint8, device="cuda")
    hdr, status = enc.decompress_batch_device(d_outc.data_ptr(), off14, len14, d_px14.data_ptr(), d_px14.numel())
    assert (status == 0).all() and all(bool((d_px14[r] == d_rgb).all()) for r in range(14))
