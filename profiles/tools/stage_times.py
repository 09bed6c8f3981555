"""usage: python scratch/stage_times.py [lib.so ...]  -- stage times (serial stream) of variants, outputs unchecked"""
import os, sys, json
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import subprocess
if len(sys.argv) > 2:
    for p in sys.argv[1:]:
        subprocess.run([sys.executable, __file__, p])
    sys.exit(0)
import felics_amd.build as b
if len(sys.argv) > 1 and sys.argv[1] != "default":
    path = os.path.abspath(sys.argv[1]); b.LIB = path; b.ensure_lib = lambda: path
import torch, felics_amd
from felics_amd import synth_torch
W, H, F = 3840, 2160, 64
dev = torch.device("cuda", 0)
frames = torch.empty((F, H, W), dtype=torch.uint8, device=dev)
for i in range(F):
    frames[i] = synth_torch.gray8(W, H, i, "S1", device=dev)
d_out = torch.empty(int(F * W * H * 1.25) + (1 << 20), dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
enc = felics_amd.Encoder(0)
import time
for _ in range(2):
    enc.compress_batch_device(frames.data_ptr(), F, W, H, 0, 0, d_out.data_ptr(), d_out.numel())
enc.set_profiling(True)
acc = {}; n = 5
t0 = time.perf_counter()
for _ in range(n):
    enc.compress_batch_device(frames.data_ptr(), F, W, H, 0, 0, d_out.data_ptr(), d_out.numel())
    for k, v in enc.stage_ms().items():
        acc[k] = acc.get(k, 0) + v / n
dt = (time.perf_counter() - t0) / n * 1e3
print(sys.argv[1] if len(sys.argv) > 1 else "default", "serial" if os.environ.get("FELICS_SERIAL") else "overlap", "step %.3f ms" % dt,
      {k: round(v, 3) for k, v in acc.items() if v > 0})
