"""Where a k_scatter workgroup's time goes, step by step (s_memtime stamps of thread 0, summed per tile).

Needs a library built with the stamps compiled in (they are not in the product build):
    profiles/tools/variant.sh sstamps -DFELICS_SCATTER_STAMPS    # in the build container
    python3 profiles/tools/scatter_stamps.py [S1|S2|S3] [frames]  # on the GPU box, from the repository root
"""
import os, sys, ctypes, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("FELICS_LIB_PATH", os.path.join(ROOT, "felics_amd", "_variants", "sstamps", "libfelics.so"))
import numpy as np, torch
import felics_amd
from felics_amd import synth_torch
kind = sys.argv[1] if len(sys.argv) > 1 else "S1"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
W, H = 3840, 2160
frames = torch.stack([synth_torch.gray8(W, H, f, kind) for f in range(n)])
d_out = torch.empty(int(n * W * H * 1.4) + (1 << 20), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
enc = felics_amd.Encoder(0)
lib = ctypes.CDLL(os.environ["FELICS_LIB_PATH"])
buf = (ctypes.c_ulonglong * (256 * 16))()
for _ in range(3):
    enc.compress_batch_device(frames.data_ptr(), n, W, H, 0, 0, d_out.data_ptr(), d_out.numel())
lib.felics_debug_scatter_stamps(buf, 1)
t = time.time()
R = 5
for _ in range(R):
    enc.compress_batch_device(frames.data_ptr(), n, W, H, 0, 0, d_out.data_ptr(), d_out.numel())
dt = (time.time() - t) / R
lib.felics_debug_scatter_stamps(buf, 0)
a = np.array(list(buf), dtype=np.float64).reshape(256, 16)
v = list(a.sum(0))
cnt = v[15]
names = ["chain positions asked for, counters zeroed, pixel loads issued", "wait for the pixels, classify, compact, rank (4 trips)", "-", "barrier",
         "layout (scan, starts) + 2 barriers", "place (start + rank)", "barrier", "out (stores, check)"]
print("%s: blocking call %.3f ms; %d tiles stamped; s_memtime ticks per tile (thread 0):" % (kind, dt * 1e3, cnt))
tot = sum(v[:8])
for i, nm in enumerate(names):
    print("  %-66s %9.0f  %5.1f %%" % (nm, v[i] / cnt, 100.0 * v[i] / tot))
print("  total %.0f ticks per tile" % (tot / cnt))
enc.close()
