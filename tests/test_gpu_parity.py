"""GPU parity: the HIP encode path (through the C ABI) against the CPU oracle, bit for bit.

Mirrors the reference's own tests (src/compression.rs:456-558, tests/compress.rs) with a seeded RNG,
then adds the committed golden fixtures, the benchmark's synthetic frames and full-size
round trips.  Integer/byte work: the bar is exact equality of the whole .felics file."""
import glob
import hashlib
import io
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def enc():
    import felics_amd

    # FELICS_POISON: the library overwrites its whole workspace with garbage before every submission, so a
    # kernel that consumes a value before the stage that produces it has run cannot pass by finding the
    # previous submission's (identical) value still in place.
    os.environ["FELICS_POISON"] = "1"
    e = felics_amd.Encoder(0)  # raises if the HIP device / libfelics.so is missing: no fallback
    del os.environ["FELICS_POISON"]
    yield e
    e.close()


def _check(enc, oracle, img, what=""):
    import felics_amd

    got = enc.compress(img)
    want = oracle.compress(img)
    if got != want:
        n = min(len(got), len(want))
        diff = next((i for i in range(n) if got[i] != want[i]), n)
        raise AssertionError("%s shape %s: GPU stream differs from oracle at byte %d (sizes %d vs %d)"
                             % (what, img.shape, diff, len(got), len(want)))
    back = felics_amd.decompress_image(io.BytesIO(got))
    assert back.shape == img.shape and (back == img).all()


def test_compression_zero_width(enc, oracle):
    """compression.rs:456-463."""
    _check(enc, oracle, np.zeros((3, 0), np.uint8))
    _check(enc, oracle, np.zeros((0, 5), np.uint8))
    _check(enc, oracle, np.zeros((0, 5, 3), np.uint8))


DIMS = [(2, 1), (1, 2), (1, 1), (4, 7), (100, 40), (124, 274), (1447, 8), (44, 1), (1, 100), (680, 480)]


def test_compression_decompression_grayscale(enc, oracle):
    """compression.rs:500-530: random u8 and u16 images."""
    rng = np.random.default_rng(21)
    for w, h in DIMS:
        _check(enc, oracle, rng.integers(0, 256, size=(h, w), dtype=np.uint8), "random gray8")
        _check(enc, oracle, rng.integers(0, 65536, size=(h, w), dtype=np.uint16), "random gray16")


def test_compression_decompression_rgb(enc, oracle):
    rng = np.random.default_rng(22)
    for w, h in DIMS:
        _check(enc, oracle, rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8), "random rgb8")
        _check(enc, oracle, rng.integers(0, 65536, size=(h, w, 3), dtype=np.uint16), "random rgb16")


def test_compression_decompression_intensive(enc, oracle):
    """compression.rs:544-558: every w, h below 20."""
    rng = np.random.default_rng(23)
    for w in range(20):
        for h in range(20):
            _check(enc, oracle, rng.integers(0, 256, size=(h, w), dtype=np.uint8))
            _check(enc, oracle, rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8))
            _check(enc, oracle, rng.integers(0, 65536, size=(h, w), dtype=np.uint16))
            _check(enc, oracle, rng.integers(0, 65536, size=(h, w, 3), dtype=np.uint16))


def test_smooth_and_extreme_content(enc, oracle):
    """Long unary runs (k = 0 with large errors), flat areas (zero-bit phased-in codes), ramps."""
    rng = np.random.default_rng(24)
    h, w = 96, 257
    flat = np.full((h, w), 7, np.uint8)
    checker = ((np.indices((h, w)).sum(0) & 1) * 255).astype(np.uint8)
    spikes = flat.copy()
    spikes[rng.integers(0, h, 200), rng.integers(0, w, 200)] = 255
    ramp = (np.arange(w)[None, :] + np.arange(h)[:, None]).astype(np.uint8)
    stripes = np.where((np.arange(w)[None, :] // 3) % 2 == 0, 0, 255).astype(np.uint8) * np.ones((h, 1), np.uint8)
    for name, img in (("flat", flat), ("checker", checker), ("spikes", spikes), ("ramp", ramp), ("stripes", stripes)):
        _check(enc, oracle, img, name)
        _check(enc, oracle, np.stack([img, img[::-1], 255 - img], axis=-1).copy(), name + " rgb")


def test_golden_fixtures(enc, oracle):
    from PIL import Image

    pins = json.load(open(os.path.join(GOLDEN, "pins.json")))
    n = 0
    for name, meta in pins["files"].items():
        img = np.array(Image.open(os.path.join(GOLDEN, name)))
        assert str(img.dtype) == meta["dtype"]
        got = enc.compress(img)
        assert got == open(os.path.join(GOLDEN, name + ".felics"), "rb").read(), name
        assert hashlib.sha256(got).hexdigest() == meta["sha256"]
        n += 1
    assert n >= 8
    img = np.array(pins["hand_vector"]["pixels"], dtype=np.uint8)
    assert enc.compress(img).hex() == pins["hand_vector"]["hex"]


def test_synthetic_frames(enc, oracle):
    from felics_amd import synth


    for kind in ("S1", "S2", "S3"):
        for (w, h) in ((64, 48), (333, 77), (1024, 256), (1920, 1080)):
            _check(enc, oracle, synth.gray8(w, h, 1, kind), kind)
    for (w, h) in ((64, 48), (333, 77), (1280, 720)):
        _check(enc, oracle, synth.rgb8(w, h, 2), "rgb S1")


def test_batch_equals_single(enc, oracle):
    from felics_amd import synth

    frames = [synth.gray8(640, 480, f, "S1") for f in range(5)] + [synth.gray8(640, 480, 9, "S2")]
    got = enc.compress_batch(frames)
    assert got == [oracle.compress(f) for f in frames]
    rgb = [synth.rgb8(320, 200, f) for f in range(4)]
    assert enc.compress_batch(rgb) == [oracle.compress(f) for f in rgb]


def test_4k_frame(enc, oracle):
    """BASELINE config 2 and 4 at full size, compared with the oracle and round-tripped."""
    from felics_amd import synth

    _check(enc, oracle, synth.gray8(3840, 2160, 0, "S1"), "4K S1")
    _check(enc, oracle, synth.gray8(3840, 2160, 0, "S2"), "4K S2")
    _check(enc, oracle, synth.rgb8(3840, 2160, 0), "4K rgb")


def test_baseline_batch_round_trip(enc, oracle):
    """BASELINE config 3 at full size (64 synthetic 4K frames in one submission): every stream decodes back to
    its frame (size-independent property), four of them are also byte-compared with the oracle, and the sizes add
    up to what the oracle's say for those four."""
    import felics_amd
    from felics_amd import synth

    frames = [synth.gray8(3840, 2160, f, "S1") for f in range(64)]
    streams = enc.compress_batch(frames)
    assert len(streams) == 64
    for i in (0, 21, 42, 63):
        assert streams[i] == oracle.compress(frames[i]), i
    for i, (f, st) in enumerate(zip(frames, streams)):
        hdr = felics_amd.read_header(io.BytesIO(st[:14]))
        assert (hdr.width, hdr.height) == (3840, 2160)
        back = felics_amd.decompress_image(io.BytesIO(st))
        assert back.dtype == np.uint8 and (back == f).all(), i


def test_config5_rgb_batch_share(enc, oracle):
    """BASELINE config 5, one GPU's share: 64 synthetic 4K RGB8 frames in ONE felics_compress_batch_device
    submission (frames generated in HBM, streams left in HBM).  Four streams are byte-compared with the oracle,
    all 64 are decoded back to their frames, and every stream carries the right header."""
    import io
    from concurrent.futures import ThreadPoolExecutor

    import felics_amd
    import torch
    from felics_amd import synth_torch

    n, w, h = 64, 3840, 2160
    frames = torch.empty((n, h, w, 3), dtype=torch.uint8, device="cuda")
    for f in range(n):
        frames[f] = synth_torch.rgb8(w, h, f)
    cap = int(n * w * h * 3 * 1.25) + (1 << 20)
    d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    offs, lens = enc.compress_batch_device(frames.data_ptr(), n, w, h, 1, 0, d_out.data_ptr(), cap)
    assert all(int(o) % 16 == 0 for o in offs) and list(offs) == sorted(offs)
    host = d_out[: int(offs[-1] + lens[-1])].cpu().numpy()
    streams = [host[int(offs[i]): int(offs[i] + lens[i])].tobytes() for i in range(n)]
    for i in (0, 22, 41, 63):
        assert streams[i] == oracle.compress(frames[i].cpu().numpy()), i

    def back(i):
        hdr = felics_amd.read_header(io.BytesIO(streams[i][:14]))
        assert (hdr.width, hdr.height, int(hdr.color_type), int(hdr.pixel_depth)) == (w, h, 1, 0)
        img = felics_amd.decompress_image(io.BytesIO(streams[i]))  # ctypes releases the GIL inside the decoder
        return bool((torch.from_numpy(img) == frames[i].cpu()).all())

    with ThreadPoolExecutor(max_workers=12) as ex:
        assert all(ex.map(back, range(n)))


def test_device_resident_batch(enc, oracle):
    """felics_compress_batch_device with torch holding the HBM buffers (plumbing only)."""
    import torch
    from felics_amd import synth

    frames = [synth.gray8(1024, 768, f, "S1") for f in range(8)]
    d_in = torch.from_numpy(np.stack(frames)).cuda()
    d_out = torch.zeros(8 * 1024 * 768 * 2, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    offs, lens = enc.compress_batch_device(d_in.data_ptr(), 8, 1024, 768, 0, 0, d_out.data_ptr(), d_out.numel())
    host = d_out.cpu().numpy()
    for i, f in enumerate(frames):
        assert offs[i] % 16 == 0
        assert host[int(offs[i]): int(offs[i] + lens[i])].tobytes() == oracle.compress(f)


def test_sixteen_bit_content(enc, oracle):
    """u16 samples: contexts up to 131 070, 15 Rice parameters, codes of up to 2^17 bits."""
    from felics_amd import synth

    rng = np.random.default_rng(31)
    h, w = 120, 333
    flat = np.full((h, w), 1000, np.uint16)
    # a quiet image drives k to 0, then a few full-scale spikes cost ~65 000-bit unary runs each
    quiet = (1000 + rng.integers(0, 2, size=(h, w))).astype(np.uint16)
    spikes = quiet.copy()
    spikes[rng.integers(2, h, 40), rng.integers(0, w, 40)] = 65535
    spikes[rng.integers(2, h, 40), rng.integers(0, w, 40)] = 0
    ramp = ((np.arange(w)[None, :] * 197 + np.arange(h)[:, None] * 31) & 0xFFFF).astype(np.uint16)
    checker = ((np.indices((h, w)).sum(0) & 1) * 65535).astype(np.uint16)
    twelve = (rng.normal(2048, 30, size=(h, w)).clip(0, 4095)).astype(np.uint16)  # 12-bit sensor in a u16 container
    for name, img in (("flat", flat), ("quiet", quiet), ("spikes", spikes), ("ramp", ramp), ("checker", checker),
                      ("twelve", twelve)):
        _check(enc, oracle, img, name + " gray16")
        _check(enc, oracle, np.stack([img, img[::-1], 65535 - img], axis=-1).copy(), name + " rgb16")
    for (wd, ht) in ((64, 48), (333, 77), (1920, 1080)):
        _check(enc, oracle, synth.gray16(wd, ht, 1), "synthetic gray16")
    frames = [synth.gray16(640, 480, f) for f in range(5)]
    assert enc.compress_batch(frames) == [oracle.compress(f) for f in frames]
    rgb = [np.stack([synth.gray16(320, 200, f), synth.gray16(320, 200, f + 7), synth.gray16(320, 200, f + 9)], axis=-1).copy()
           for f in range(3)]
    assert enc.compress_batch(rgb) == [oracle.compress(f) for f in rgb]


def test_sixteen_bit_exact_placement(enc, oracle):
    """16-bit and RGB through the device entry point with a buffer too small for fixed slots: the streams are
    placed back to back (two-pass pack, exact placement)."""
    import torch
    from felics_amd import synth

    for frames, color, depth in (([synth.gray16(512, 384, f) for f in range(5)], 0, 1),
                                 ([synth.rgb8(512, 384, f) for f in range(5)], 1, 0)):
        flat = np.random.default_rng(3).integers(0, 3, size=frames[0].shape).astype(frames[0].dtype)
        frames = frames[:2] + [flat] + frames[2:]  # one small stream among big ones
        want = [oracle.compress(f) for f in frames]
        cap = sum((len(x) + 15) // 16 * 16 for x in want) + 64
        d_in = torch.from_numpy(np.stack(frames).view(np.uint8)).cuda()
        d_out = torch.zeros(cap, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        offs, lens = enc.compress_batch_device(d_in.data_ptr(), len(frames), 512, 384, color, depth, d_out.data_ptr(), cap)
        host = d_out.cpu().numpy()
        assert list(offs) == sorted(offs) and all(int(o) % 16 == 0 for o in offs)
        for i, x in enumerate(want):
            assert host[int(offs[i]): int(offs[i] + lens[i])].tobytes() == x, i


def test_sixteen_bit_4k(enc, oracle):
    from felics_amd import synth

    _check(enc, oracle, synth.gray16(3840, 2160, 0), "4K gray16")


def test_pack_variants(oracle):
    """The single-pass pack is the default for gray frames; FELICS_TWO_PASS selects the lengths + pack kernels,
    and a look-back that gives up makes the context fall back to them.  All must produce the oracle's bytes."""
    import felics_amd
    from felics_amd import synth

    frames = [synth.gray8(1920, 1080, f, "S1") for f in range(16)] + [synth.gray8(1920, 1080, 3, "S2")]
    want = [oracle.compress(f) for f in frames]
    # default: k per record from k_assign3 into the tile's own slots, read back by the single-pass pack (k_pack_t), tiles by workgroup
    # index; a look-back that gives up switches to tiles by ticket (what FELICS_OWN_TAILS starts with), a second one to the two-pass
    # kernels (k to a byte per pixel, lengths, bit scan, pack); FELICS_SERIAL / FELICS_TRACE are the profiling / debugging aids
    for env in ({}, {"FELICS_TWO_PASS": "1"}, {"FELICS_TEST_LOOKBACK_FAIL": "1"}, {"FELICS_OWN_TAILS": "1"},
                {"FELICS_OWN_TAILS": "1", "FELICS_TEST_LOOKBACK_FAIL": "1"}, {"FELICS_LANES": "1"}, {"FELICS_LANES": "4", "FELICS_SLICES": "12"},
                {"FELICS_SERIAL": "1", "FELICS_SLICES": "1"}, {"FELICS_TRACE": "1", "FELICS_TIMEOUT_S": "30"},
                {"FELICS_SLICES_QUEUED": "5"},  # (the host entry point queues its chunks: slices per queued submission, the sweep tools' knob)
                # the front kernel's ranks: from ballots from the start; the default's order check failing once (-> ballots), also
                # under the two-pass pack; a tile that outgrows its slots (-> the batch again with worst-case tiles)
                {"FELICS_SCATTER": "ballot"}, {"FELICS_TEST_SCATTER_ORDER": "1"}, {"FELICS_SCATTER": "ballot", "FELICS_TWO_PASS": "1"},
                {"FELICS_TEST_SCATTER_ORDER": "1", "FELICS_TWO_PASS": "1"}, {"FELICS_TEST_TILE_CAP": "1"},
                {"FELICS_TEST_TILE_CAP": "1", "FELICS_TWO_PASS": "1"}):
        os.environ.update(env)
        os.environ["FELICS_POISON"] = "1"
        try:
            e = felics_amd.Encoder(0)
        finally:
            for k in list(env) + ["FELICS_POISON"]:
                del os.environ[k]
        try:
            assert e.compress_batch(frames) == want, env
            assert e.compress_batch(frames[:3]) == want[:3], env  # the same context again (after a fallback)
            rgb = [synth.rgb8(640, 360, f) for f in range(3)]
            assert e.compress_batch(rgb) == [oracle.compress(f) for f in rgb], env
            st = e.stats()
            if "FELICS_TEST_LOOKBACK_FAIL" in env:  # tickets first (unless the context started with them), then two passes
                # (the host entry point queues its chunks two deep: the chunk in flight beside the first failure reports one as well)
                assert st["two_pass"] == 1 and st["lookback_fallbacks"] == (2 if "FELICS_OWN_TAILS" in env else 3), (env, st)
                assert st["ticket_retries"] == (0 if "FELICS_OWN_TAILS" in env else 1), (env, st)
            else:
                assert st["two_pass"] == (1 if "FELICS_TWO_PASS" in env else 0) and st["lookback_fallbacks"] == 0, (env, st)
            # (the host entry point queues its chunks two deep: the chunk in flight beside the first failure reports one as well)
            assert st["scatter_fallbacks"] in ((1, 2) if "FELICS_TEST_SCATTER_ORDER" in env else (0,)), (env, st)
            assert st["tile_overflows"] in ((1, 2) if "FELICS_TEST_TILE_CAP" in env else (0,)), (env, st)
        finally:
            e.close()


def test_lookback_failure_with_two_submissions_in_flight(oracle):
    """Two queued submissions, both launched without tickets, whose look-backs both give up (FELICS_TEST_LOOKBACK_FAIL): the first
    wait turns tickets on and -- its redo failing with tickets as well -- moves the context to the two-pass kernels; the second
    wait finds a sub-batch that ran WITHOUT tickets, so it counts a fallback but does not escalate on its own account
    (felics_api.cpp: note_lookback_failure looks at what the failed sub-batch was launched with).  Streams equal the oracle's."""
    import torch
    import felics_amd
    from felics_amd import synth

    frames = [synth.gray8(800, 600, f, "S1") for f in range(4)]
    want = [oracle.compress(f) for f in frames]
    os.environ["FELICS_TEST_LOOKBACK_FAIL"] = "1"
    try:
        e = felics_amd.Encoder(0)
    finally:
        del os.environ["FELICS_TEST_LOOKBACK_FAIL"]
    try:
        cap = 800 * 600 * 4 * 2
        d_in = torch.from_numpy(np.stack(frames)).cuda()
        outs = [torch.zeros(cap, dtype=torch.uint8, device="cuda") for _ in range(2)]
        torch.cuda.synchronize()
        tickets = [e.submit_batch_device(d_in.data_ptr(), 4, 800, 600, 0, 0, outs[i].data_ptr(), cap) for i in range(2)]
        for i, t in enumerate(tickets):
            offs, lens = e.wait_batch(t)
            host = outs[i].cpu().numpy()
            for j in range(4):
                assert host[int(offs[j]): int(offs[j] + lens[j])].tobytes() == want[j], (i, j)
        st = e.stats()
        assert st["two_pass"] == 1 and st["ticket_retries"] == 1 and st["lookback_fallbacks"] == 3, st
        assert e.compress_batch(frames) == want  # the context, now on the two-pass kernels
    finally:
        e.close()


def test_queued_submissions_of_changing_content(oracle):
    """Queued submissions whose content changes from one to the next (flat, noise, smooth): the tile-local pipeline classifies a
    pixel once whatever the content, every sub-batch's events are ranked by the front kernel's returning LDS atomics (its order
    check never fires), no batch is redone."""
    import torch
    import felics_amd
    from felics_amd import synth

    flat = [np.full((700, 1000), 7 + f, dtype=np.uint8) for f in range(3)]      # 1 bit per pixel
    noise = [synth.gray8(1000, 700, f, "S2") for f in range(3)]
    smooth = [synth.gray8(1000, 700, f, "S1") for f in range(3)]
    e = felics_amd.Encoder(0)
    try:
        assert e.compress_batch(flat) == [oracle.compress(f) for f in flat]
        seq = (flat, flat, noise, noise, flat, noise, smooth)
        cap = 1000 * 700 * 3 * 2
        d_outs = [torch.zeros(cap, dtype=torch.uint8, device="cuda") for _ in range(2)]
        d_ins = [torch.from_numpy(np.stack(frames)).cuda() for frames in seq]
        torch.cuda.synchronize()
        inflight = []

        def finish(item):
            i, t = item
            offs, lens = e.wait_batch(t)
            host = d_outs[i % 2].cpu().numpy()
            for j, f in enumerate(seq[i]):
                assert host[int(offs[j]): int(offs[j] + lens[j])].tobytes() == oracle.compress(f), (i, j)

        for i, frames in enumerate(seq):  # two submissions in flight
            if len(inflight) == 2:
                finish(inflight.pop(0))
            inflight.append((i, e.submit_batch_device(d_ins[i].data_ptr(), len(frames), 1000, 700, 0, 0, d_outs[i % 2].data_ptr(), cap)))
        while inflight:
            finish(inflight.pop(0))
        st = e.stats()
        assert st["sorted_event_sorts"] == st["submissions"], st
        assert st["scatter_fallbacks"] == 0 and st["lookback_fallbacks"] == 0 and st["tile_overflows"] == 0 and st["slot_overflows"] == 0, st
    finally:
        e.close()


def test_sixteen_bit_lane_replay(oracle):
    """16-bit samples: a large batch replays every chain on four lanes (k_wide_chains_quad) up to a limit of events and hands
    the rest of a longer chain -- position and counters -- to the wave-per-chain kernel.  FELICS_WIDE_LANE forces the
    limit on small inputs: hand-over after 1, 7, 64 events and never, against the wave-per-chain kernel alone (0)."""
    import felics_amd
    from felics_amd import synth

    rng = np.random.default_rng(5)
    frames = [synth.gray16(640, 360, f) for f in range(3)]
    frames.append(rng.integers(0, 65536, size=(97, 131), dtype=np.uint16))                 # noise: k = 14 territory
    frames.append((np.arange(200 * 300).reshape(200, 300) % 7).astype(np.uint16))          # few contexts, long chains
    flat = np.full((64, 64), 40000, dtype=np.uint16)
    flat[::9, ::5] = 3                                                                      # full-scale spikes
    frames.append(flat)
    rgb = [np.stack([synth.gray16(320, 200, f), synth.gray16(320, 200, f + 9), synth.gray16(320, 200, f + 5) // 3], -1) for f in range(2)]
    want = [oracle.compress(f) for f in frames]
    want_rgb = [oracle.compress(f) for f in rgb]
    for limit in ("0", "1", "7", "64", "1000000"):
        os.environ["FELICS_WIDE_LANE"] = limit
        os.environ["FELICS_POISON"] = "1"
        try:
            e = felics_amd.Encoder(0)
            try:
                for f, w in zip(frames, want):
                    assert e.compress_batch([f]) == [w], (limit, f.shape)
                same = [frames[0], frames[1], frames[2]]
                assert e.compress_batch(same) == want[:3], limit
                assert e.compress_batch(rgb) == want_rgb, limit
            finally:
                e.close()
        finally:
            del os.environ["FELICS_WIDE_LANE"], os.environ["FELICS_POISON"]
    # the limit the library chooses by itself: thirteen 4K frames are enough for the four-lane form
    big = [synth.gray16(3840, 2160, f % 5) for f in range(13)]
    e = felics_amd.Encoder(0)
    try:
        assert e.compress_batch(big) == [oracle.compress(f) for f in big]
    finally:
        e.close()


def test_random_shapes_and_contents(enc, oracle):
    """A few seconds of random geometry (1 pixel wide to 5000), content (noise, ramps, flat with full-scale
    spikes, constant), sample type and batch size, every stream compared with the oracle."""
    import time

    rng = np.random.default_rng(77)

    def content(h, w, c, dt, kind):
        mx = 255 if dt == np.uint8 else 65535
        shape = (h, w) if c == 1 else (h, w, 3)
        if kind == 0:
            return rng.integers(0, mx + 1, size=shape).astype(dt)
        if kind == 1:
            base = np.add.outer(np.arange(h) * 3, np.arange(w) * 2) % (mx + 1)
            img = base if c == 1 else np.stack([base, base[::-1], (base * 7) % (mx + 1)], -1)
            return (img + rng.integers(0, 4, size=shape)).clip(0, mx).astype(dt)
        if kind == 2:
            img = np.full(shape, mx // 3, dtype=np.int64) + rng.integers(0, 2, size=shape)
            n = max(1, h * w // 50)
            img[rng.integers(0, h, n), rng.integers(0, w, n)] = rng.choice([0, mx], size=(n,) if c == 1 else (n, 3))
            return img.astype(dt)
        return np.full(shape, rng.integers(0, mx + 1), dtype=dt)

    t0, batches = time.time(), 0
    while time.time() - t0 < 8.0:
        w = int(rng.choice([rng.integers(1, 40), rng.integers(1, 700), rng.integers(1000, 5000)]))
        h = int(rng.choice([rng.integers(1, 40), rng.integers(1, 300)]))
        c, dt = int(rng.choice([1, 3])), rng.choice([np.uint8, np.uint8, np.uint16])
        frames = [content(h, w, c, dt, int(rng.integers(0, 4))) for _ in range(int(rng.integers(1, 10)))]
        assert enc.compress_batch(frames) == [oracle.compress(f) for f in frames], (w, h, c, dt, len(frames))
        batches += 1
    assert batches > 20


def test_sixteen_bit_lane_replay_random_shapes(oracle):
    """The four-lane chain kernel and its hand-over on random geometry: 16-bit gray and RGB frames from 1 pixel wide to a few
    thousand, noise / ramps / spikes / constant, batches of 1-6 frames, with the chain limit forced to 3 events, then to 50
    (FELICS_WIDE_LANE; the library itself only takes this path for batches of ~100 M samples)."""
    import time
    import felics_amd

    rng = np.random.default_rng(1234)

    def content(h, w, c, kind):
        shape = (h, w) if c == 1 else (h, w, 3)
        if kind == 0:
            return rng.integers(0, 65536, size=shape).astype(np.uint16)
        if kind == 1:
            base = np.add.outer(np.arange(h) * 5, np.arange(w) * 3) % 65536
            img = base if c == 1 else np.stack([base, base[::-1], (base * 7) % 65536], -1)
            return (img + rng.integers(0, 9, size=shape)).clip(0, 65535).astype(np.uint16)
        if kind == 2:
            img = np.full(shape, 21000, dtype=np.int64) + rng.integers(0, 3, size=shape)
            n = max(1, h * w // 40)
            img[rng.integers(0, h, n), rng.integers(0, w, n)] = rng.choice([0, 65535], size=(n,) if c == 1 else (n, 3))
            return img.astype(np.uint16)
        return np.full(shape, rng.integers(0, 65536), dtype=np.uint16)

    for limit in ("3", "50"):
        os.environ["FELICS_WIDE_LANE"] = limit
        try:
            e = felics_amd.Encoder(0)
            try:
                t0, batches = time.time(), 0
                while time.time() - t0 < 4.0:
                    w = int(rng.choice([rng.integers(1, 40), rng.integers(1, 700), rng.integers(1000, 3000)]))
                    h = int(rng.choice([rng.integers(1, 40), rng.integers(1, 200)]))
                    c = int(rng.choice([1, 3]))
                    frames = [content(h, w, c, int(rng.integers(0, 4))) for _ in range(int(rng.integers(1, 7)))]
                    assert e.compress_batch(frames) == [oracle.compress(f) for f in frames], (limit, w, h, c, len(frames))
                    batches += 1
                assert batches > 8
            finally:
                e.close()
        finally:
            del os.environ["FELICS_WIDE_LANE"]


def test_several_passes(oracle):
    """A batch larger than one pass holds (FELICS_TEST_PASS_IMAGES caps the pass instead of a 400-frame 4K
    batch): host-pointer and device entry points, gray / RGB / 16-bit."""
    import felics_amd
    import torch
    from felics_amd import synth

    os.environ["FELICS_TEST_PASS_IMAGES"] = "3"
    try:
        with felics_amd.Encoder(0) as e:
            for frames in ([synth.gray8(640, 480, f, "S1") for f in range(10)], [synth.rgb8(320, 240, f) for f in range(7)],
                           [synth.gray16(320, 240, f) for f in range(8)]):
                assert e.compress_batch(frames) == [oracle.compress(f) for f in frames]
            frames = [synth.gray8(1024, 512, f, "S1") for f in range(11)]
            d_in = torch.from_numpy(np.stack(frames)).cuda()
            d_out = torch.zeros(11 * 1024 * 512 * 2, dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            offs, lens = e.compress_batch_device(d_in.data_ptr(), 11, 1024, 512, 0, 0, d_out.data_ptr(), d_out.numel())
            host = d_out.cpu().numpy()
            for i, f in enumerate(frames):
                assert host[int(offs[i]): int(offs[i] + lens[i])].tobytes() == oracle.compress(f)
    finally:
        del os.environ["FELICS_TEST_PASS_IMAGES"]


def test_two_submissions_in_flight(enc, oracle):
    """felics_submit_batch_device / felics_wait_batch: batch i + 1 is queued before batch i is waited for (gray,
    RGB with a slot overflow in one batch, 16-bit; different shapes back to back), every stream checked."""
    import felics_amd
    import torch
    from felics_amd import synth

    rng = np.random.default_rng(9)
    batches = []
    for b in range(6):
        if b % 3 == 0:
            frames, color, depth = [synth.gray8(1280, 720, 10 * b + f, "S1") for f in range(6)], 0, 0
        elif b % 3 == 1:
            frames, color, depth = [synth.rgb8(640, 360, 10 * b + f) for f in range(4)], 1, 0
            if b == 4:  # noise does not fit the slots below: exact placement on the wait
                frames[2] = rng.integers(0, 256, size=frames[2].shape, dtype=np.uint8)
        else:
            frames, color, depth = [synth.gray16(800, 600, 10 * b + f) for f in range(3)], 0, 1
        batches.append((frames, color, depth))
    d_in = [torch.from_numpy(np.stack(fr).view(np.uint8)).cuda() for fr, _, _ in batches]
    caps = [int(sum(f.nbytes for f in fr) * 1.1) + 4096 for fr, _, _ in batches]
    d_out = [torch.zeros(c, dtype=torch.uint8, device="cuda") for c in caps]
    torch.cuda.synchronize()

    def submit(b):
        fr, color, depth = batches[b]
        h, w = fr[0].shape[:2]
        return enc.submit_batch_device(d_in[b].data_ptr(), len(fr), w, h, color, depth, d_out[b].data_ptr(), caps[b])

    def check(b, offs, lens):
        host = d_out[b].cpu().numpy()
        for i, f in enumerate(batches[b][0]):
            assert host[int(offs[i]): int(offs[i] + lens[i])].tobytes() == oracle.compress(f), (b, i)

    pending = submit(0)
    for b in range(1, len(batches)):
        nxt = submit(b)
        with pytest.raises(felics_amd.FelicsError):  # the synchronous entry points wait their turn
            enc.compress(batches[0][0][0])
        check(b - 1, *enc.wait_batch(pending))
        pending = nxt
    check(len(batches) - 1, *enc.wait_batch(pending))
    with pytest.raises(felics_amd.FelicsError):
        enc.wait_batch(pending)  # nothing is outstanding any more
    assert enc.compress(batches[0][0][0]) == oracle.compress(batches[0][0][0])


def test_timeout_leaves_a_failed_context(oracle):
    """A wait for the GPU that times out (forced here) must not leave a context that looks idle while its kernels may
    still run: the call fails with FELICS_E_HIP, so does every later call, the lane stays busy, stats say `failed`,
    and destroying the context does not touch the device objects.  A fresh context works."""
    import felics_amd
    import torch
    from felics_amd import synth

    frames = [synth.gray8(640, 480, f, "S1") for f in range(3)]
    os.environ["FELICS_TEST_TIMEOUT"] = "1"
    try:
        e = felics_amd.Encoder(0)
    finally:
        del os.environ["FELICS_TEST_TIMEOUT"]
    with pytest.raises(felics_amd.FelicsError) as ei:
        e.compress_batch(frames)
    assert ei.value.code == -9 and "timed out" in str(ei.value)
    assert e.stats()["failed"] == 1
    with pytest.raises(felics_amd.FelicsError) as ei:
        e.compress(frames[0])
    assert ei.value.code == -9
    d_in = torch.from_numpy(np.stack(frames)).cuda()
    d_out = torch.zeros(3 * 640 * 480 * 2, dtype=torch.uint8, device="cuda")
    with pytest.raises(felics_amd.FelicsError):
        e.submit_batch_device(d_in.data_ptr(), 3, 640, 480, 0, 0, d_out.data_ptr(), d_out.numel())
    torch.cuda.synchronize()  # (the kernels of the "timed-out" call do finish here; the context cannot know that)
    e.close()
    with felics_amd.Encoder(0) as e2:
        assert e2.compress_batch(frames) == [oracle.compress(f) for f in frames]


def test_errors(enc):
    import felics_amd

    with pytest.raises(TypeError):
        enc.compress(np.zeros((4, 4), np.float32))


def test_cfelics_dfelics_cli(tmp_path):
    """src/bin/cfelics.rs / dfelics.rs as a user runs them: TIFF -> .felics (GPU) -> TIFF."""
    import subprocess

    from PIL import Image

    build = os.path.join(os.path.dirname(os.path.dirname(__file__)), "felics_amd", "_build")
    for name, line in (("6.3.09.tiff", "Compressing 8-bit grayscale image..."), ("house.tiff", "Compressing 8-bit rgb image..."),
                       ("lena_color_256.tif", "Compressing 8-bit rgb image...")):
        out = str(tmp_path / (name + ".felics"))
        r = subprocess.run([os.path.join(build, "cfelics"), "-i", os.path.join(GOLDEN, name), "-o", out],
                           capture_output=True, text=True)
        assert r.returncode == 0 and r.stdout.strip() == line, r.stdout + r.stderr
        assert open(out, "rb").read() == open(os.path.join(GOLDEN, name + ".felics"), "rb").read()
        back = str(tmp_path / (name + ".back.tiff"))
        r = subprocess.run([os.path.join(build, "dfelics"), "-i", out, "-o", back], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        assert (np.array(Image.open(back)) == np.array(Image.open(os.path.join(GOLDEN, name)))).all()
    r = subprocess.run([os.path.join(build, "cfelics"), "-i", os.path.join(GOLDEN, "aerial.tiff"), "-o", str(tmp_path / "a.felics")],
                       capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip() == "Compressing 16-bit grayscale image...", r.stdout + r.stderr
    assert open(str(tmp_path / "a.felics"), "rb").read() == open(os.path.join(GOLDEN, "aerial.tiff.felics"), "rb").read()


def test_cfelics_reads_png_and_compressed_tiff(tmp_path, oracle):
    """cfelics.rs:36-44 takes whatever `image` decodes: PNG and LZW / Deflate / PackBits TIFF inputs give the stream of
    the same pixels; dfelics writes PNG by extension (dfelics.rs:45-52); the corpus script runs end to end."""
    import subprocess
    import sys

    from PIL import Image

    build = os.path.join(os.path.dirname(os.path.dirname(__file__)), "felics_amd", "_build")
    for name in ("6.3.09.tiff", "house.tiff", "aerial.tiff"):
        img = np.array(Image.open(os.path.join(GOLDEN, name)))
        want = open(os.path.join(GOLDEN, name + ".felics"), "rb").read()
        variants = {"v.png": {}, "lzw.tiff": {"compression": "tiff_lzw"}, "zip.tiff": {"compression": "tiff_adobe_deflate"},
                    "pb.tiff": {"compression": "packbits"}}
        for fn, kw in variants.items():
            src = str(tmp_path / fn)
            Image.fromarray(img).save(src, **kw)
            out = str(tmp_path / (fn + ".fel"))
            r = subprocess.run([os.path.join(build, "cfelics"), "-i", src, "-o", out], capture_output=True, text=True)
            assert r.returncode == 0, r.stdout + r.stderr
            assert open(out, "rb").read() == want, (name, fn)
        back = str(tmp_path / "back.png")
        r = subprocess.run([os.path.join(build, "dfelics"), "-i", out, "-o", back], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        assert (np.array(Image.open(back)) == img).all()
    rgba = str(tmp_path / "rgba.png")
    Image.fromarray(np.zeros((4, 4, 4), np.uint8)).save(rgba)
    r = subprocess.run([os.path.join(build, "cfelics"), "-i", rgba, "-o", str(tmp_path / "x.fel")], capture_output=True, text=True)
    assert r.returncode == 1 and r.stdout.strip() == "Unsupported image format: Rgba8"  # cfelics.rs:70
    root = os.path.dirname(os.path.dirname(__file__))
    r = subprocess.run([sys.executable, os.path.join(root, "bench", "corpus.py"), "--out", str(tmp_path / "corpus")],
                       capture_output=True, text=True, timeout=500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "Compression times:" in r.stdout and '"round_trip": "ok"' in r.stdout


def test_odd_geometry(enc, oracle):
    """Widths that are not multiples of 16 (unaligned rows in the tile staging), tiles that end mid-row,
    very narrow and very wide images."""
    rng = np.random.default_rng(31)
    for w, h in ((17, 300), (4097, 3), (4095, 5), (3, 3000), (2, 5000), (1, 5000), (5000, 1), (5000, 2), (255, 257), (1023, 65)):
        _check(enc, oracle, rng.integers(0, 256, size=(h, w), dtype=np.uint8), "odd gray")
        smooth = (np.add.outer(np.arange(h), np.arange(w)) // 3 % 256).astype(np.uint8)
        _check(enc, oracle, smooth, "odd smooth")
        _check(enc, oracle, rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8), "odd rgb")


def test_batch_sizes_and_lanes(enc, oracle):
    """Batches below, at and above the sub-batch pipeline's thresholds give the same streams."""
    from felics_amd import synth

    frames = [synth.gray8(200, 100, f, "S1" if f % 3 else "S2") for f in range(37)]
    want = [oracle.compress(f) for f in frames]
    for n in (1, 7, 8, 16, 25, 37):
        assert enc.compress_batch(frames[:n]) == want[:n], n


def test_fresh_context_large_batch(oracle):
    """A new context (nothing left over from earlier submissions) on inputs big enough for every slice and
    lane of the pipeline to be in flight at once."""
    import felics_amd
    from felics_amd import synth

    frames = [synth.gray8(3840, 2160, f, "S1") for f in range(10)]
    want = [hashlib.sha256(oracle.compress(f)).hexdigest() for f in frames]
    for _ in range(2):
        with felics_amd.Encoder(0) as e:
            got = e.compress_batch(frames)
        assert [hashlib.sha256(g).hexdigest() for g in got] == want
    with felics_amd.Encoder(0) as e:
        img = synth.rgb8(3840, 2160, 3)
        assert e.compress(img) == oracle.compress(img)


def test_slot_overflow_falls_back_to_exact_placement(enc, oracle):
    """felics_compress_batch_device gives every stream a fixed slot of capacity / n bytes.  A stream that
    does not fit its slot makes the library lay the streams out back to back instead; if even that does
    not fit, FELICS_E_BUFFER_TOO_SMALL reports the capacity needed."""
    import torch

    import felics_amd
    from felics_amd import synth

    w, h = 512, 384
    frames = [synth.gray8(w, h, 0, "S3"), synth.gray8(w, h, 1, "S2"), synth.gray8(w, h, 2, "S3"), synth.gray8(w, h, 3, "S1")]
    want = [oracle.compress(f) for f in frames]
    assert len(want[1]) > w * h  # the noise frame expands
    d_in = torch.from_numpy(np.stack(frames)).cuda()
    cap = 4 * ((len(want[1]) - 4096) // 16 * 16)  # slot a little smaller than the noise stream
    assert sum((len(x) + 15) // 16 * 16 for x in want) <= cap
    d_out = torch.zeros(cap, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    offs, lens = enc.compress_batch_device(d_in.data_ptr(), 4, w, h, 0, 0, d_out.data_ptr(), cap)
    host = d_out.cpu().numpy()
    assert all(int(o) % 16 == 0 for o in offs) and list(offs) == sorted(offs)
    assert offs[1] != cap // 4 // 16 * 16  # not the slot layout any more
    for i in range(4):
        assert host[int(offs[i]): int(offs[i] + lens[i])].tobytes() == want[i], i
    # capacity below the exact need: an error that says how much is needed
    small = sum((len(x) + 15) // 16 * 16 for x in want) - 1024
    with pytest.raises(felics_amd.FelicsError) as ei:
        enc.compress_batch_device(d_in.data_ptr(), 4, w, h, 0, 0, d_out.data_ptr(), small)
    assert ei.value.code == -8 and "need" in str(ei.value)
    # rgb through the same fallback
    rgb = [synth.rgb8(256, 200, f) for f in range(3)]
    rgb[1] = np.random.default_rng(5).integers(0, 256, size=(200, 256, 3), dtype=np.uint8)
    want = [oracle.compress(f) for f in rgb]
    d_in = torch.from_numpy(np.stack(rgb)).cuda()
    cap = 3 * ((len(want[1]) - 2048) // 16 * 16)
    d_out = torch.zeros(cap, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    offs, lens = enc.compress_batch_device(d_in.data_ptr(), 3, 256, 200, 1, 0, d_out.data_ptr(), cap)
    host = d_out.cpu().numpy()
    for i in range(3):
        assert host[int(offs[i]): int(offs[i] + lens[i])].tobytes() == want[i], i


def test_teardown_after_every_lane_was_used(oracle):
    """Destroying a context whose lanes have all run (they share the tail stream), with shared and with private tail
    streams and for every lane count, and destroying a failed context.  Round 2's records hold an abort inside
    felics_ctx_destroy (gpurun_out/r2b/t2.log): lane 0's loop iteration destroyed the shared tail stream and lane 1's
    iteration then synchronised the same handle; DESIGN.md 5.1.  Run once: a teardown that aborts takes the process."""
    import felics_amd
    import torch
    from felics_amd import synth

    frames = [synth.gray8(960, 540, f, "S1") for f in range(4)]
    want = [oracle.compress(f) for f in frames]
    d_in = torch.from_numpy(np.stack(frames)).cuda()
    for env in ({}, {"FELICS_LANES": "1"}, {"FELICS_LANES": "3"}, {"FELICS_LANES": "4"}, {"FELICS_OWN_TAILS": "1"},
                {"FELICS_LANES": "4", "FELICS_OWN_TAILS": "1"}):
        os.environ.update(env)
        try:
            e = felics_amd.Encoder(0)
        finally:
            for k in env:
                del os.environ[k]
        depth_q = e.lane_count()  # (the context's own count: the environment of a moment ago no longer applies)
        assert depth_q == int(env.get("FELICS_LANES", 2))
        outs = [torch.zeros(4 * 960 * 540 * 2, dtype=torch.uint8, device="cuda") for _ in range(depth_q)]
        subs = [e.submit_batch_device(d_in.data_ptr(), 4, 960, 540, 0, 0, o.data_ptr(), o.numel()) for o in outs]  # every lane
        for sub, o in zip(subs, outs):
            offs, lens = e.wait_batch(sub)
            host = o.cpu().numpy()
            assert [host[int(a): int(a + n)].tobytes() for a, n in zip(offs, lens)] == want, env
        last = e.submit_batch_device(d_in.data_ptr(), 4, 960, 540, 0, 0, outs[0].data_ptr(), outs[0].numel())
        e.wait_batch(last)
        e.close()  # the teardown under test
    # a context that failed while a lane was busy: nothing of it is touched, the process goes on
    os.environ["FELICS_TEST_TIMEOUT"] = "1"
    try:
        e = felics_amd.Encoder(0)
    finally:
        del os.environ["FELICS_TEST_TIMEOUT"]
    out = torch.zeros(4 * 960 * 540 * 2, dtype=torch.uint8, device="cuda")
    sub = e.submit_batch_device(d_in.data_ptr(), 4, 960, 540, 0, 0, out.data_ptr(), out.numel())
    with pytest.raises(felics_amd.FelicsError):
        e.wait_batch(sub)
    torch.cuda.synchronize()
    e.close()
    with felics_amd.Encoder(0) as e2:
        assert e2.compress_batch(frames) == want


def _golden_images():
    from PIL import Image

    out = {}
    for p in sorted(glob.glob(os.path.join(GOLDEN, "*.tif")) + glob.glob(os.path.join(GOLDEN, "*.tiff"))):
        out[os.path.basename(p)] = np.array(Image.open(p))
    return out


def _mosaic(tiles, rows, cols):
    """rows x cols mosaic of same-shape tiles, every other one mirrored (so that the seams are not all alike)."""
    out = []
    for r in range(rows):
        row = []
        for c in range(cols):
            t = tiles[(r * cols + c) % len(tiles)]
            if (r + c) % 2:
                t = t[:, ::-1]
            if r % 2:
                t = t[::-1]
            row.append(t)
        out.append(np.concatenate(row, axis=1))
    return np.ascontiguousarray(np.concatenate(out, axis=0))


def test_natural_image_mosaics(enc, oracle):
    """The reference's integration test round-trips its whole image suite (tests/compress.rs:73-103); the GPU box only has
    the eight 256 x 256 golden files, so natural content larger than 16 tiles is built from them: 2048 x 2048 mosaics
    (gray8, gray16, RGB8; every other tile mirrored), one 1500 x 1100 mosaic with ragged geometry, and a 64-frame batch of
    shifted 1024 x 768 crops -- all byte-compared with the oracle."""
    imgs = _golden_images()
    gray8 = [a for a in imgs.values() if a.ndim == 2 and a.dtype == np.uint8 and a.shape == (256, 256)]
    gray16 = [a for a in imgs.values() if a.ndim == 2 and a.dtype == np.uint16 and a.shape == (256, 256)]
    rgb8 = [a for a in imgs.values() if a.ndim == 3 and a.dtype == np.uint8 and a.shape[:2] == (256, 256)]
    assert gray8 and rgb8, sorted(imgs)
    big8 = _mosaic(gray8, 8, 8)
    _check(enc, oracle, big8, "gray8 mosaic")
    _check(enc, oracle, _mosaic(rgb8, 8, 8), "rgb8 mosaic")
    if gray16:
        _check(enc, oracle, _mosaic(gray16, 8, 8), "gray16 mosaic")
    else:  # no 16-bit golden of that size: the 8-bit content scaled, with natural-looking low bits
        wide = (big8.astype(np.uint16) << 8) | np.roll(big8, 7, axis=1).astype(np.uint16)
        _check(enc, oracle, wide, "gray16 mosaic (scaled gray8 content)")
    _check(enc, oracle, np.ascontiguousarray(_mosaic(gray8, 6, 6)[:1100, :1500]), "ragged gray8 mosaic")
    crops = [np.ascontiguousarray(big8[7 * i: 7 * i + 768, 13 * i: 13 * i + 1024]) for i in range(64)]
    got = enc.compress_batch(crops)
    for i in (0, 1, 31, 63):
        assert got[i] == oracle.compress(crops[i]), i
    import felics_amd

    for i in range(64):
        assert (felics_amd.decompress_image(io.BytesIO(got[i])) == crops[i]).all(), i


def test_suite_originals_full_size(enc, oracle):
    """Originals of the reference's integration corpus at full size through the HIP encoder (tests/compress.rs:73-103 round-trips
    every file of image-suite/): the four 1024 x 1024 gray8 files, gray16 at 1000 x 1000 and 1024 x 1024, three 1024 x 1024 RGB8
    files, the little-endian multi-strip 512 x 512 RGB8 file and a 512 x 512 file of each kind (tests/golden/suite/, data;
    make_golden.py).  Each is encoded singly and compared with the committed oracle stream and its pinned sha256; the files of
    one shape then go through ONE batch call (mixed content side by side in one submission); everything is decoded back."""
    import felics_amd
    from PIL import Image

    pins = json.load(open(os.path.join(GOLDEN, "pins.json")))["suite"]
    assert len(pins) >= 13
    imgs, want = {}, {}
    for name, meta in sorted(pins.items()):
        path = os.path.join(GOLDEN, "suite", name)
        img = np.array(Image.open(path))
        assert list(img.shape) == meta["shape"] and str(img.dtype) == meta["dtype"], name
        imgs[name] = np.ascontiguousarray(img)
        want[name] = open(path + ".felics", "rb").read()
        assert hashlib.sha256(want[name]).hexdigest() == meta["sha256"] and len(want[name]) == meta["size"], name
        got = enc.compress(imgs[name])
        assert got == want[name], name
        assert (felics_amd.decompress_image(io.BytesIO(got)) == imgs[name]).all(), name
    groups = {}
    for name, img in imgs.items():
        groups.setdefault((img.shape, str(img.dtype)), []).append(name)
    assert max(len(v) for v in groups.values()) >= 4  # the 1024 x 1024 gray8 files in one call
    for key, names in groups.items():
        if len(names) > 1:
            got = enc.compress_batch([imgs[n] for n in names] + [imgs[names[0]][::-1].copy()])
            assert got[:-1] == [want[n] for n in names], key
            assert got[-1] == oracle.compress(imgs[names[0]][::-1].copy()), key


def test_many_tiny_sixteen_bit_frames(enc, oracle):
    """A batch of 70 000 tiny 16-bit frames: the 16-bit path takes at most 2^16 planes per pass (its front end scans tiles x
    planes counts in one workgroup), so this batch goes through two passes; a sample is compared with the oracle and decoded back."""
    import felics_amd

    rng = np.random.default_rng(11)
    base = [rng.integers(0, 65536, size=(5, 7), dtype=np.uint16) for _ in range(257)]
    frames = [base[i % 257] for i in range(70000)]
    got = enc.compress_batch(frames)
    want = [oracle.compress(f) for f in base]
    assert all(got[i] == want[i % 257] for i in range(70000))
    for i in range(0, 70000, 4999):
        assert (felics_amd.decompress_image(io.BytesIO(got[i])) == frames[i]).all(), i
