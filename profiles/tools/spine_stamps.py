"""Where the two waves of the longest chain's spine workgroup (context 1 of plane 0 of an S1 frame: k_spine3's walker and
helper) spend their time.

Needs a library built with the stamps compiled in (they are not in the product build):
    profiles/tools/variant.sh sstamps -DFELICS_SPINE_STAMPS
    python profiles/tools/spine_stamps.py [frames]        # on the GPU box, from the repository root
"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("FELICS_LIB_PATH", os.path.join(ROOT, "felics_amd", "_variants", "sstamps", "libfelics.so"))
os.environ.setdefault("FELICS_SLICES", "1")
import numpy as np, torch
import felics_amd
from felics_amd import synth_torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
W, H = 3840, 2160
frames = torch.stack([synth_torch.gray8(W, H, f, "S1") for f in range(n)])
d_out = torch.empty(int(n * W * H * 1.25) + (1 << 20), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
enc = felics_amd.Encoder(0)
lib = ctypes.CDLL(os.environ["FELICS_LIB_PATH"])
buf = (ctypes.c_ulonglong * 8)()
for _ in range(3):
    enc.compress_batch_device(frames.data_ptr(), n, W, H, 0, 0, d_out.data_ptr(), d_out.numel())
lib.felics_debug_spine_stamps(buf, 1)
R = 5
import time
t = time.time()
for _ in range(R):
    enc.compress_batch_device(frames.data_ptr(), n, W, H, 0, 0, d_out.data_ptr(), d_out.numel())
dt = (time.time() - t) / R
lib.felics_debug_spine_stamps(buf, 0)
v = [x / R for x in buf]
win, total = v[5], v[7]
print("%d frame(s), blocking call %.3f ms; chain (plane 0, context 1): %.0f windows of 64 records, walker's life %.0f ticks (shader cycles) = %.0f per window"
      % (n, dt * 1e3, win, total, total / max(win, 1)))
for i, nm in enumerate(["walker: waiting at the hand-over barrier", "walker: walking", "helper: finishing + producing windows", "helper: waiting at the barrier"]):
    print("  %-42s %10.0f ticks  %5.1f %%  %7.1f per window" % (nm, v[i], 100.0 * v[i] / total, v[i] / max(win, 1)))
print("  halvings walked by all chains of the call: %.0f" % v[6])
enc.close()
