"""GPU decoder throughput by batch size, both forms (one wave per stream / 64 streams per wave): runs ON THE GPU BOX.

    python3 profiles/tools/decode_scale.py [--kind S1] [--sizes 64,256,1024,4096] [--rgb]

The streams are the GPU encoder's of 64 synthetic 4K gray8 frames, referenced repeatedly for the larger batches (the decoder
reads them through offsets / lens; every decoded image has a buffer of its own).  Prints GPix/s per form and batch size, the
host decoder's rate on 1 and 16 cores, and the batch size from which each GPU form beats 16 host cores (linear interpolation).
"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np, torch
import felics_amd
from felics_amd import synth_torch, api

ap = argparse.ArgumentParser()
ap.add_argument("--kind", default="S1")
ap.add_argument("--sizes", default="64,256,1024,4096")
ap.add_argument("--rgb", action="store_true", help="RGB8 streams of 1920x1080 frames (a 4K RGB batch of 4096 would need 300 GB for frames + planes)")
a = ap.parse_args()
W, H, F = (1920, 1080, 64) if a.rgb else (3840, 2160, 64)
C = 3 if a.rgb else 1
dev = torch.device("cuda", 0)
if a.rgb:
    from felics_amd import synth
    frames = torch.stack([torch.from_numpy(synth.rgb8(W, H, f)) for f in range(F)]).to(dev)
else:
    frames = torch.stack([synth_torch.gray8(W, H, f, a.kind, device=dev) for f in range(F)])
d_out = torch.empty(int(F * W * H * C * 1.25) + (1 << 20), dtype=torch.uint8, device=dev)
enc = felics_amd.Encoder(0)
torch.cuda.synchronize()  # (the frames are torch's work on torch's stream; the library runs on streams of its own)
offs, lens = enc.compress_batch_device(frames.data_ptr(), F, W, H, 1 if a.rgb else 0, 0, d_out.data_ptr(), d_out.numel())
host = d_out[: int(offs[-1] + lens[-1])].cpu().numpy()
sample = [host[int(offs[i]): int(offs[i] + lens[i])].tobytes() for i in range(16)]
from concurrent.futures import ThreadPoolExecutor
t = time.perf_counter(); api.decompress_bytes(sample[0]); one = time.perf_counter() - t
t = time.perf_counter()
with ThreadPoolExecutor(max_workers=16) as ex:
    list(ex.map(api.decompress_bytes, sample))
many = time.perf_counter() - t
host1, host16 = W * H / one / 1e9, 16 * W * H / many / 1e9
print("host decoder (felics_decompress, C++): %.3f GPix/s on 1 core, %.3f on 16" % (host1, host16))
res = {}
for n in [int(x) for x in a.sizes.split(",")]:
    o = np.array([offs[i % F] for i in range(n)], dtype=np.uint64)
    l = np.array([lens[i % F] for i in range(n)], dtype=np.uint64)
    d_px = torch.empty((n, H, W, C) if a.rgb else (n, H, W), dtype=torch.uint8, device=dev)
    for form in ("0", "1"):
        os.environ["FELICS_TEST_DECODE_LANES"] = form
        best = None
        for rep in range(2):
            torch.cuda.synchronize(); t = time.perf_counter()
            _, st = enc.decompress_batch_device(d_out.data_ptr(), o, l, d_px.data_ptr(), d_px.numel())
            torch.cuda.synchronize(); dt = time.perf_counter() - t
            best = dt if best is None else min(best, dt)
        assert (st == 0).all()
        ok = bool((d_px[n - 1] == frames[(n - 1) % F]).all()) and bool((d_px[0] == frames[0]).all())
        res[(form, n)] = n * W * H / best / 1e9
        print("%5d streams  %-22s %.3f GPix/s  (%.3f s)  pixels %s" % (n, "one wave per stream" if form == "0" else "64 streams per wave", res[(form, n)], best, "ok" if ok else "WRONG"), flush=True)
    del d_px
del os.environ["FELICS_TEST_DECODE_LANES"]
for form, name in (("0", "one wave per stream"), ("1", "64 streams per wave")):
    pts = sorted((n, r) for (f, n), r in res.items() if f == form)
    be = None
    for (n0, r0), (n1, r1) in zip(pts, pts[1:]):
        if r0 < host16 <= r1:
            be = n0 + (host16 - r0) / (r1 - r0) * (n1 - n0)
    print("%s beats 16 host cores from ~%s streams" % (name, "%d" % be if be else ("the smallest batch" if pts and pts[0][1] >= host16 else "more than %d" % pts[-1][0])))
enc.close()
