"""usage: python profiles/tools/stage_times_kind.py <S1|S2|S3> -- per-stage times of 64 4K gray8 frames of that content (FELICS_SERIAL=1 FELICS_SLICES=1 for every kernel alone)"""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, felics_amd, time
from felics_amd import synth_torch
kind = sys.argv[1] if len(sys.argv) > 1 else "S1"
W, H, F = 3840, 2160, 64
dev = torch.device("cuda", 0)
frames = torch.stack([synth_torch.gray8(W, H, i, kind, device=dev) for i in range(F)])
d_out = torch.empty(int(F * W * H * 1.4) + (1 << 20), dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
enc = felics_amd.Encoder(0)
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
print(kind, "serial" if os.environ.get("FELICS_SERIAL") else "overlap", "step %.3f ms" % dt, {k: round(v, 3) for k, v in acc.items() if v > 0}, enc.stats())
