"""felics_compress_batch end to end: 64 4K gray8 frames in page-locked host memory in, 64 streams in page-locked host memory out.
Runs ON THE GPU BOX:  python3 profiles/tools/e2e_host.py [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import felics_amd
from felics_amd import synth_torch
W, H, F = 3840, 2160, 64
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda", 0)
frames = torch.stack([synth_torch.gray8(W, H, f, "S1", device=dev) for f in range(F)])
torch.cuda.synchronize()
h_in = frames.cpu().pin_memory()
npix = W * H
slot_h = int(npix * 1.25) + 4096
h_out = torch.empty((F, slot_h), dtype=torch.uint8).pin_memory()
h_out.zero_()
in_ptrs = [h_in[i].data_ptr() for i in range(F)]
out_ptrs = [h_out[i].data_ptr() for i in range(F)]
enc = felics_amd.Encoder(0)
for _ in range(2):
    lens = enc.compress_batch_host(in_ptrs, F, W, H, 0, 0, out_ptrs, [slot_h] * F)
ts = []
for _ in range(reps):
    t = time.perf_counter()
    lens = enc.compress_batch_host(in_ptrs, F, W, H, 0, 0, out_ptrs, [slot_h] * F)
    ts.append(time.perf_counter() - t)
ts.sort()
out_bytes = int(np.asarray(lens).sum())
print("host buffers in and out, %d frames: median %.2f ms, best %.2f (%.1f GB/s in, %.1f GB/s out at the median)" % (F, ts[len(ts) // 2] * 1e3, ts[0] * 1e3, F * npix / ts[len(ts) // 2] / 1e9, out_bytes / ts[len(ts) // 2] / 1e9))
enc.close()
