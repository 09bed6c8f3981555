"""Where a pack workgroup's time goes, phase by phase (s_memtime stamps of thread 0, summed per tile; k_pack_g).

Needs a library built with the stamps compiled in (they are not in the product build):
    profiles/tools/variant.sh pstamps -DFELICS_PACK_STAMPS       # in the build container
    python3 profiles/tools/pack_stamps.py 64                     # on the GPU box, from the repository root
"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("FELICS_LIB_PATH", os.path.join(ROOT, "felics_amd", "_variants", "pstamps", "libfelics.so"))
import numpy as np, torch
import felics_amd
from felics_amd import synth_torch, api
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
W, H = 3840, 2160
frames = torch.stack([synth_torch.gray8(W, H, f, "S1") for f in range(n)])
d_out = torch.empty(int(n * W * H * 1.25) + (1 << 20), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
enc = felics_amd.Encoder(0)
lib = ctypes.CDLL(os.environ["FELICS_LIB_PATH"])
buf = (ctypes.c_ulonglong * (256 * 16))()
import time
for _ in range(3):
    enc.compress_batch_device(frames.data_ptr(), n, W, H, 0, 0, d_out.data_ptr(), d_out.numel())
lib.felics_debug_pack_stamps(buf, 1)
t = time.time()
R = 5
for _ in range(R):
    enc.compress_batch_device(frames.data_ptr(), n, W, H, 0, 0, d_out.data_ptr(), d_out.numel())
dt = (time.time() - t) / R
lib.felics_debug_pack_stamps(buf, 0)
a = np.array(list(buf), dtype=np.float64).reshape(256, 16); v = list(a.sum(0))
cnt = v[15]
names = ["tile (ticket / workgroup index)", "window out (shift, stores)", "gather k through the runs (waits for round trip 1)", "wait for the other waves", "-", "phase A (sixteen codes) + scan", "sync", "look-back + sync", "phase B (codes into the window)", "run table + pixel loads issued, window cleared", "-"]
print("blocking call %.3f ms; %d tiles stamped; s_memtime ticks per tile (thread 0):" % (dt * 1e3, cnt))
tot = sum(v[:11])
for i, nm in enumerate(names):
    if nm != "-":
        print("  %-50s %9.0f  %5.1f %%" % (nm, v[i] / cnt, 100.0 * v[i] / tot))
print("  total %.0f ticks per tile" % (tot / cnt))
enc.close()
