"""Randomised parity run of the GPU encoder against the oracle (tests/oracle_lib): shapes from 1x1 up, widths around the kernels'
trip and tile sizes, flat / noise / gradient / synthetic content, gray8 and RGB8, batches of 1-5 frames, blocking calls and queued
submissions; three contexts: the default, one that ranks events with ballots (FELICS_SCATTER=ballot), one whose first batch overflows its
tiles and which sizes them for the worst case from then on (FELICS_TEST_TILE_CAP).  Runs ON THE GPU BOX from the repository root:  python3 profiles/tools/fuzz_encode.py [seconds] [seed]
Test infrastructure: the oracle is the checker here, as in tests/."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import felics_amd
from felics_amd import synth
from tests import oracle_lib

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
oracle = oracle_lib.load()
encs = {}
MODES = {"default": {}, "ballot": {"FELICS_SCATTER": "ballot"}, "worst-case tiles": {"FELICS_TEST_TILE_CAP": "1"}}
for mode, env in MODES.items():
    os.environ.update(env)
    encs[mode] = felics_amd.Encoder(0)
    for k in env:
        del os.environ[k]
special_w = [1, 2, 3, 4, 5, 63, 64, 65, 127, 255, 256, 257, 511, 512, 513, 1023, 1024, 1025, 4095, 4096, 4097]


def content(kind, h, w, c, f):
    shape = (h, w, 3) if c else (h, w)
    if kind == 0:
        return np.full(shape, rng.integers(0, 256), dtype=np.uint8)
    if kind == 1:
        return rng.integers(0, 256, size=shape, dtype=np.uint8)
    if kind == 2:
        return rng.integers(0, 1 << rng.integers(1, 8), size=shape, dtype=np.uint8)
    if kind == 3:
        y, x = np.mgrid[0:h, 0:w]
        g = ((x * rng.integers(1, 5) + y * rng.integers(1, 5)) // rng.integers(1, 9) + rng.integers(0, 3, size=(h, w))) % 256
        g = g.astype(np.uint8)
        return np.stack([g, (g // 2).astype(np.uint8), (255 - g).astype(np.uint8)], axis=-1) if c else g
    return synth.rgb8(w, h, f) if c else synth.gray8(w, h, f, "S1")


t0 = time.time()
cases = frames_done = 0
by_mode = {m: 0 for m in encs}
while time.time() - t0 < budget:
    w = int(rng.choice(special_w)) if rng.random() < 0.4 else int(rng.integers(1, 900))
    h = int(rng.integers(1, 40)) if w > 1500 else int(rng.integers(1, 300))
    if rng.random() < 0.05:
        w, h = int(rng.integers(1500, 4000)), int(rng.integers(500, 2200))
    c = int(rng.random() < 0.3)
    n = int(rng.integers(1, 6)) if w * h < 500000 else 1
    kind = int(rng.integers(0, 5))
    frames = [content(kind, h, w, c, f) for f in range(n)]
    want = [oracle.compress(f) for f in frames]
    mode = list(MODES)[int(rng.integers(0, 3))]
    enc = encs[mode]
    if rng.random() < 0.5:
        got = enc.compress_batch(frames)
    else:
        d_in = torch.from_numpy(np.stack(frames)).cuda()
        cap = int(sum(len(x) for x in want) * 2 + 4096 * n)
        d_out = torch.zeros(cap, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        offs, lens = enc.wait_batch(enc.submit_batch_device(d_in.data_ptr(), n, w, h, c, 0, d_out.data_ptr(), cap))
        host = d_out.cpu().numpy()
        got = [host[int(offs[i]): int(offs[i] + lens[i])].tobytes() for i in range(n)]
    if got != want:
        print("MISMATCH: w=%d h=%d rgb=%d n=%d content=%d mode=%s" % (w, h, c, n, kind, mode), flush=True)
        sys.exit(1)
    cases += 1
    frames_done += n
    by_mode[mode] += 1
st = {m: e.stats() for m, e in encs.items()}
print("%d cases (%d frames) in %.0f s, all streams equal to the oracle's; by mode %s" % (cases, frames_done, time.time() - t0, by_mode))
for m, s in st.items():
    print("  %-16s submissions %d, ranked by LDS atomics %d, order-check fallbacks %d, tile overflows %d, look-back fallbacks %d, slot overflows %d"
          % (m, s["submissions"], s["sorted_event_sorts"], s["scatter_fallbacks"], s["tile_overflows"], s["lookback_fallbacks"], s["slot_overflows"]))
for e in encs.values():
    e.close()
