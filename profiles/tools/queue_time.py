"""What-if timing: ms per queued step of builds whose OUTPUT IS NOT CHECKED (diagnostic builds that stub a stage out or do it
twice, profiles/tools/variant.sh ... -DFELICS_EXP_*), the same two-deep submission queue as bench.py.  Runs ON THE GPU BOX.

    python3 profiles/tools/queue_time.py [--kind S1] [--steps 40] [--rounds 2] name ...      (names under felics_amd/_variants/, or "tree")
"""
import argparse, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def one(name, kind, steps):
    import felics_amd.build as b
    if name != "tree":
        path = os.path.join(ROOT, "felics_amd", "_variants", name, "libfelics.so")
        b.LIB = path
        b.ensure_lib = lambda: path
    import torch, felics_amd
    from felics_amd import synth_torch
    W, H, F = 3840, 2160, 64
    dev = torch.device("cuda", 0)
    frames = torch.empty((F, H, W), dtype=torch.uint8, device=dev)
    for i in range(F):
        frames[i] = synth_torch.gray8(W, H, i, kind, device=dev)
    enc = felics_amd.Encoder(0)
    q = enc.lane_count()
    outs = [torch.empty(int(F * W * H * 1.25) + (1 << 20), dtype=torch.uint8, device=dev) for _ in range(q)]
    torch.cuda.synchronize()
    sub = lambda i: enc.submit_batch_device(frames.data_ptr(), F, W, H, 0, 0, outs[i % q].data_ptr(), outs[i % q].numel())
    for i in range(q):
        enc.wait_batch(sub(i))
    res = []
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fl = []
        for i in range(steps):
            if len(fl) == q:
                enc.wait_batch(fl.pop(0))
            fl.append(sub(i))
        while fl:
            enc.wait_batch(fl.pop(0))
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / steps * 1e3)
    t0 = time.perf_counter()
    for _ in range(5):
        enc.compress_batch_device(frames.data_ptr(), F, W, H, 0, 0, outs[0].data_ptr(), outs[0].numel())
    blocking = (time.perf_counter() - t0) / 5 * 1e3
    print("%-24s %s queued ms/step %s   blocking %.3f" % (name, kind, " ".join("%.3f" % r for r in res), blocking), flush=True)
    enc.close()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--kind", default="S1")
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--one", default=None)
    ap.add_argument("names", nargs="*")
    a = ap.parse_args()
    if a.one:
        one(a.one, a.kind, a.steps)
    else:
        for _ in range(a.rounds):
            for n in a.names:
                subprocess.run([sys.executable, __file__, "--one", n, "--kind", a.kind, "--steps", str(a.steps)])
