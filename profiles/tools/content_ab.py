# ms per queued step and per blocking call of 64 4K gray8 frames of different content (the library FELICS_LIB_PATH points at, or the tree's)
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import felics_amd
from felics_amd import synth_torch
W, H, n = 3840, 2160, 64
dev = "cuda"
def natural(f):
    g = torch.Generator(device=dev); g.manual_seed(1000 + f)
    x = torch.arange(W, device=dev)[None, :].float(); y = torch.arange(H, device=dev)[:, None].float()
    base = 110 + 60 * torch.sin(x / 173.0 + f) * torch.cos(y / 211.0) + 25 * torch.sin((x + 2 * y) / 37.0)
    edges = 40 * ((torch.floor(x / 97.0) + torch.floor(y / 131.0)) % 2)               # blocks with sharp borders
    amp = 1.0 + 14.0 * (0.5 + 0.5 * torch.sin(x / 61.0) * torch.sin(y / 47.0)) ** 4   # texture: noise of varying strength
    noise = torch.randn((H, W), device=dev, generator=g) * amp
    return (base + edges + noise).clamp(0, 255).to(torch.uint8)
kinds = {"S1": lambda f: synth_torch.gray8(W, H, f, "S1", device=dev), "S2 (noise)": lambda f: synth_torch.gray8(W, H, f, "S2", device=dev),
         "S3 (flat)": lambda f: synth_torch.gray8(W, H, f, "S3", device=dev), "natural-like": natural}
enc = felics_amd.Encoder(0)
for name, gen in kinds.items():
    frames = torch.stack([gen(f) for f in range(n)])
    cap = int(n * W * H * 1.4) + (1 << 20)
    d_out = torch.empty(cap, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    for _ in range(2):
        offs, lens = enc.compress_batch_device(frames.data_ptr(), n, W, H, 0, 0, d_out.data_ptr(), cap)
    t = time.perf_counter(); R = 6
    for _ in range(R):
        offs, lens = enc.compress_batch_device(frames.data_ptr(), n, W, H, 0, 0, d_out.data_ptr(), cap)
    dt = (time.perf_counter() - t) / R
    q = enc.lane_count()
    outs = [d_out] + [torch.empty(cap, dtype=torch.uint8, device=dev) for _ in range(q - 1)]
    for i in range(q):
        enc.wait_batch(enc.submit_batch_device(frames.data_ptr(), n, W, H, 0, 0, outs[i].data_ptr(), cap))
    torch.cuda.synchronize(); t = time.perf_counter(); K = 16; fl = []
    for i in range(K):
        if len(fl) == q:
            enc.wait_batch(fl.pop(0))
        fl.append(enc.submit_batch_device(frames.data_ptr(), n, W, H, 0, 0, outs[i % q].data_ptr(), cap))
    while fl:
        enc.wait_batch(fl.pop(0))
    torch.cuda.synchronize(); dq = (time.perf_counter() - t) / K
    print("%-14s %.3f ms per queued step (%d MPix/s), %.3f ms per blocking call, %.2f bits/pixel" % (name, dq * 1e3, n * W * H / dq / 1e6, dt * 1e3, 8.0 * float(sum(lens)) / (n * W * H)), flush=True)
    del frames, d_out, outs
enc.close()
