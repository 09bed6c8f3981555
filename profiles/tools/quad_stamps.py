"""Where a wave of k_wide_chains_quad spends its time (s_memtime stamps of every wave, summed per phase).

Needs a library built with the stamps compiled in (they are not in the product build):
    make -C felics_amd/csrc lib OUT=../../scratch/qstamps CXXFLAGS="-O3 -std=c++17 -fPIC -DFELICS_QUAD_STAMPS"
    python3 profiles/tools/quad_stamps.py        # on the GPU box, from the repository root
"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("FELICS_LIB_PATH", os.path.join(ROOT, "scratch", "qstamps", "libfelics.so"))
import numpy as np, torch
import felics_amd
from felics_amd import synth
n, W, H = 16, 3840, 2160
base = [torch.from_numpy(synth.gray16(W, H, i).view(np.int16)) for i in range(4)]
frames = torch.stack([base[i % 4] for i in range(n)]).cuda()
d_out = torch.empty(n * W * H * 3, dtype=torch.uint8, device="cuda")
enc = felics_amd.Encoder(0)
lib = ctypes.CDLL(os.environ["FELICS_LIB_PATH"])
buf = (ctypes.c_ulonglong * 8)()
for _ in range(2):
    enc.compress_batch_device(frames.data_ptr(), n, W, H, 0, 1, d_out.data_ptr(), d_out.numel())
lib.felics_debug_quad_stamps(buf, 1)
enc.compress_batch_device(frames.data_ptr(), n, W, H, 0, 1, d_out.data_ptr(), d_out.numel())
lib.felics_debug_quad_stamps(buf, 0)
v = list(buf)
names = ["wait for the chunk + park in LDS", "k stores of the chunk before", "issue next chunk's loads", "rounds", "chunks", "set-up / hand-over per group"]
tot = v[0] + v[1] + v[2] + v[3] + v[5]
print("chunks %d; ticks per chunk (s_memtime, 100 MHz):" % v[4])
for i in (0, 1, 2, 3, 5):
    print("  %-36s %8.2f  %5.1f %%" % (names[i], v[i] / max(v[4], 1), 100.0 * v[i] / tot))
