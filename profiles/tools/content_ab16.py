# blocking-call time of 16 4K gray16 frames of different content, with the four-lane chain kernel limited to different
# numbers of events per chain (FELICS_WIDE_LANE; 0 = wave-per-chain kernel only), for the library FELICS_LIB_PATH points at
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import felics_amd
from felics_amd import synth
W, H, n = 3840, 2160, 16
dev = "cuda"
def natural16(f):
    g = torch.Generator(device=dev); g.manual_seed(2000 + f)
    x = torch.arange(W, device=dev)[None, :].float(); y = torch.arange(H, device=dev)[:, None].float()
    base = 28000 + 15000 * torch.sin(x / 173.0 + f) * torch.cos(y / 211.0) + 6000 * torch.sin((x + 2 * y) / 37.0)
    edges = 9000 * ((torch.floor(x / 97.0) + torch.floor(y / 131.0)) % 2)
    amp = 2.0 + 600.0 * (0.5 + 0.5 * torch.sin(x / 61.0) * torch.sin(y / 47.0)) ** 4   # from nearly clean to strongly textured
    noise = torch.randn((H, W), device=dev, generator=g) * amp
    return (base + edges + noise).clamp(0, 65535).to(torch.int32).to(torch.int16)   # (bit pattern of the u16 value)
def smooth16(f):  # a height map: almost no noise, long chains in the smallest contexts
    x = torch.arange(W, device=dev)[None, :].float(); y = torch.arange(H, device=dev)[:, None].float()
    v = 30000 + 20000 * torch.sin(x / 400.0 + f) * torch.cos(y / 300.0) + 3.0 * torch.sin(x * 0.9) * torch.cos(y * 1.1)
    return v.clamp(0, 65535).to(torch.int32).to(torch.int16)
def s1(f):
    return torch.from_numpy(synth.gray16(W, H, f % 4).view(np.int16)).to(dev)
def noise16(f):
    g = torch.Generator(device=dev); g.manual_seed(3000 + f)
    return torch.randint(0, 65536, (H, W), device=dev, generator=g).to(torch.int32).to(torch.int16)
kinds = {"S1 (bench)": s1, "natural-like": natural16, "smooth height map": smooth16, "noise": noise16}
enc = felics_amd.Encoder(0)
for name, gen in kinds.items():
    frames = torch.stack([gen(f) for f in range(n)])
    cap = int(n * W * H * 3.2) + (1 << 20)
    d_out = torch.empty(cap, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    for lim in ("default", "0", "2048", "4096", "8192", "16384"):
        if lim == "default":
            os.environ.pop("FELICS_WIDE_LANE", None)
        else:
            os.environ["FELICS_WIDE_LANE"] = lim
        for _ in range(2):
            offs, lens = enc.compress_batch_device(frames.data_ptr(), n, W, H, 0, 1, d_out.data_ptr(), cap)
        t = time.perf_counter(); R = 4
        for _ in range(R):
            offs, lens = enc.compress_batch_device(frames.data_ptr(), n, W, H, 0, 1, d_out.data_ptr(), cap)
        dt = (time.perf_counter() - t) / R
        print("%-18s limit %-8s %.3f ms per 16 frames  %.2f bits/pixel" % (name, lim, dt * 1e3, 8.0 * float(sum(lens)) / (n * W * H)), flush=True)
    del frames, d_out
enc.close()
