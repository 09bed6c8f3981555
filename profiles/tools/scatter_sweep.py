"""The two event sorts against content: 64 4K gray8 frames of a smooth surface plus noise of growing strength, ms per queued
step and per blocking call with FELICS_SCATTER=ballot and =sorted (felics_api.cpp: scatter_mode).  Runs ON THE GPU BOX."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import felics_amd
W, H, n = 3840, 2160, 64
dev = "cuda"
def frame(f, amp):
    g = torch.Generator(device=dev); g.manual_seed(1000 + f)
    x = torch.arange(W, device=dev)[None, :].float(); y = torch.arange(H, device=dev)[:, None].float()
    base = 128 + 60 * torch.sin(x / 173.0 + f) * torch.cos(y / 211.0) + 25 * torch.sin((x + 2 * y) / 37.0)
    return (base + torch.randn((H, W), device=dev, generator=g) * amp).clamp(0, 255).to(torch.uint8)
encs = {}
for mode in ("ballot", "sorted"):
    os.environ["FELICS_SCATTER"] = mode
    encs[mode] = felics_amd.Encoder(0)
del os.environ["FELICS_SCATTER"]
cap = int(n * W * H * 1.4) + (1 << 20)
q = encs["ballot"].lane_count()
outs = [torch.empty(cap, dtype=torch.uint8, device=dev) for _ in range(q)]
for amp in (0.0, 0.5, 1.0, 2.0, 3.0, 4.0, 6.0, 8.0, 12.0, 20.0, 40.0, 80.0):
    frames = torch.stack([frame(f, amp) for f in range(n)])
    torch.cuda.synchronize()
    line = "noise sigma %5.1f" % amp
    for rnd in range(2):
        for mode, enc in encs.items():
            sub = lambda i: enc.submit_batch_device(frames.data_ptr(), n, W, H, 0, 0, outs[i % q].data_ptr(), cap)
            for i in range(q):
                offs, lens = enc.wait_batch(sub(i))
            torch.cuda.synchronize(); t = time.perf_counter(); K = 12; fl = []
            for i in range(K):
                if len(fl) == q:
                    enc.wait_batch(fl.pop(0))
                fl.append(sub(i))
            while fl:
                enc.wait_batch(fl.pop(0))
            torch.cuda.synchronize(); dq = (time.perf_counter() - t) / K
            t = time.perf_counter()
            for _ in range(3):
                enc.compress_batch_device(frames.data_ptr(), n, W, H, 0, 0, outs[0].data_ptr(), cap)
            db = (time.perf_counter() - t) / 3
            if rnd == 0 and mode == "ballot":
                line += "  %.2f bits/pixel " % (8.0 * float(sum(lens)) / (n * W * H))
            line += "  %s %.3f / %.3f" % (mode, dq * 1e3, db * 1e3)
    print(line + "   (queued / blocking ms, two rounds)", flush=True)
    del frames
for e in encs.values():
    e.close()
