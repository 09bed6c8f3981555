#!/usr/bin/env python3
"""bench.py -- encode throughput of the FELICS GPU path on synthetic 4K 8-bit grayscale frames.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the whole encode path (felics_compress_batch_device) over one batch of
FRAMES synthetic S1 frames that are already resident in HBM; the .felics
streams end up in HBM too.  Frames are independent streams, so ranks shard them with no data-path
collective (weak scaling: every rank encodes its own FRAMES frames); torch.distributed is used for
the barrier and the max-over-ranks time only.

Rank 0 prints one JSON line: MPix/s over all ranks, the HBM-read roofline of the dominant kernel
(HIP events on the library's own stream, averaged over the timed steps) and the CPU baseline (the
oracle -- a C port of the reference algorithm, the Rust reference cannot be built here -- timed on a
bounded sample of the same frames on this box's host cores).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# libfelics runs four HIP streams; ask the ROCm runtime for enough hardware queues before torch
# initialises it (see felics_amd/api.py)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

W4K, H4K = 3840, 2160
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=3, choices=[2, 3, 4, 5],
                    help="BASELINE.json config: 2 = one 4K gray8 frame, 3 = 64 gray8 frames (headline, default), "
                         "4 = one 4K RGB8 frame, 5 = 64 RGB8 frames per GPU (one GPU's share of the 512-frame job)")
    ap.add_argument("--frames", type=int, default=None, help="frames per rank per step (default: what --config says)")
    ap.add_argument("--kind", default="S1", choices=["S1", "S2", "S3"])
    ap.add_argument("--rgb", action="store_true", help="RGB8 frames (config 4/5) instead of gray8")
    ap.add_argument("--depth16", action="store_true", help="16-bit grayscale frames (side measurement, not the headline)")
    ap.add_argument("--width", type=int, default=W4K)
    ap.add_argument("--height", type=int, default=H4K)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline leg (0 = skip)")
    ap.add_argument("--check-frames", type=int, default=2, help="frames byte-compared with the oracle before timing")
    ap.add_argument("--synchronous", action="store_true", help="time the blocking entry point (one batch at a time)")
    ap.add_argument("--no-blocking-extra", action="store_true",
                    help="skip the three extra blocking-call steps after the timed region (rocprofv3 runs: keeps the kernel averages those of the timed steps)")
    ap.add_argument("--no-side-configs", action="store_true", help="skip the short config 2 / config 4 legs of the default run")
    ap.add_argument("--no-stage-timing", action="store_true", help="leave the library's per-launch event timing off during the timed steps (roofline is then null)")
    ap.add_argument("--no-decode-leg", action="store_true", help="skip the decode measurement (GPU decoder next to the host decoder) of the default run")
    ap.add_argument("--lanes", type=int, default=None, help="submissions in flight (sets FELICS_LANES; default: the library's 2, and 4 with --depth16, "
                    "whose kernels wait for memory more than for the ALUs: 3.95 against 4.36 ms per 16 frames)")
    args = ap.parse_args()
    if args.lanes is None and args.depth16 and "FELICS_LANES" not in os.environ:
        args.lanes = 4
    if args.lanes is not None:
        os.environ["FELICS_LANES"] = str(args.lanes)
    if args.config in (4, 5):
        args.rgb = True
    if args.frames is None:
        args.frames = 1 if args.config in (2, 4) else 64
    return args


def self_launch(args):
    """`python bench.py --gpus N` outside torchrun: start N ranks as a CHILD process (this parent has made no
    GPU call yet) and pass its output and exit code on.  A run never reports fewer GPUs than it was asked for."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def cpu_baseline(frames_np, budget_s):
    """Oracle (C port of the reference's serial algorithm) on host cores: 1 core, then all cores."""
    from concurrent.futures import ThreadPoolExecutor

    from tests import oracle_lib

    oracle = oracle_lib.load()
    px = frames_np[0].shape[0] * frames_np[0].shape[1]
    # one core (the reference is single-threaded, src/compression.rs:117-146)
    t0 = time.perf_counter()
    done = 0
    for f in frames_np:
        oracle.compress(f)
        done += 1
        if time.perf_counter() - t0 > budget_s * 0.45:
            break
    t1 = time.perf_counter() - t0
    single = {"value": done * px / t1 / 1e6, "unit": "MPix/s", "cores": 1, "kind": "port",
              "sample": "%d of the batch's %dx%d frames, oracle/felics_oracle.c -O3, 1 thread" % (done, frames_np[0].shape[1], frames_np[0].shape[0])}
    # all cores, one image per thread (ctypes releases the GIL inside the C call)
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(cores, 16)  # the CPU share that goes with one GPU of the box
    per_frame = t1 / done
    n = max(cores, min(len(frames_np) * 8, int(budget_s * 0.45 / per_frame) * cores))
    n = max(cores, (n // cores) * cores)
    work = [frames_np[i % len(frames_np)] for i in range(n)]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:
        list(ex.map(oracle.compress, work))
    t2 = time.perf_counter() - t0
    multi = {"value": n * px / t2 / 1e6, "unit": "MPix/s", "cores": cores, "kind": "port",
             "sample": "%d frame encodes over %d threads, one image per thread" % (n, cores)}
    return single, multi


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    import numpy as np
    import torch
    import felics_amd
    from felics_amd import synth_torch

    from felics_amd import dist as fdist

    rank, world, local = fdist.env_rank()
    if world != args.gpus:  # never print n_gpus different from what was asked for
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if torch.cuda.device_count() < world or local >= torch.cuda.device_count():
        raise SystemExit("--gpus %d but this node shows %d GPU(s)" % (args.gpus, torch.cuda.device_count()))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    group = fdist.Group("nccl", dev)  # RCCL; barrier + MAX of the time only, no data-path collective

    W, H, F = args.width, args.height, args.frames
    channels = 3 if args.rgb else 1
    npix = W * H
    # this rank's shard of the job (weak scaling: F frames per rank), generated straight into HBM
    first_frame, _ = fdist.shard_range(F * world, rank, world)
    sample_bytes = 2 if args.depth16 else 1
    if args.depth16:
        if args.rgb:
            raise SystemExit("--depth16 is grayscale only")
        from felics_amd import synth

        base = [torch.from_numpy(synth.gray16(W, H, first_frame + i).view(np.int16)) for i in range(min(F, 4))]
        frames = torch.stack([base[i % len(base)] for i in range(F)]).to(dev)  # int16 storage of the u16 samples
    else:
        frames = torch.empty((F, H, W, channels) if args.rgb else (F, H, W), dtype=torch.uint8, device=dev)
        for i in range(F):
            f = first_frame + i
            frames[i] = synth_torch.rgb8(W, H, f, device=dev) if args.rgb else synth_torch.gray8(W, H, f, args.kind, device=dev)
    out_cap = int(F * npix * channels * sample_bytes * 1.25) + (1 << 20)
    enc = felics_amd.Encoder(local)
    depth_q = max(1, enc.lane_count())  # batches that can be in flight on this context
    d_outs = [torch.empty(out_cap, dtype=torch.uint8, device=dev) for _ in range(depth_q)]
    d_out = d_outs[0]
    torch.cuda.synchronize()
    color = 1 if args.rgb else 0

    depth = 1 if args.depth16 else 0

    def step():  # synchronous entry point
        return enc.compress_batch_device(frames.data_ptr(), F, W, H, color, depth, d_out.data_ptr(), d_out.numel())

    def submit(i):  # felics_submit_batch_device: returns once the batch is queued
        o = d_outs[i % depth_q]
        return enc.submit_batch_device(frames.data_ptr(), F, W, H, color, depth, o.data_ptr(), o.numel())

    # ---- parity first: byte-compare a few streams with the oracle, checksum the rest ----
    offs, lens = step()
    host = d_out[: int(offs[-1] + lens[-1])].cpu().numpy()
    from tests import oracle_lib

    oracle = oracle_lib.load()
    checked = 0
    for i in range(min(args.check_frames, F)):
        want = oracle.compress(frames[i].cpu().numpy().view(np.uint16) if args.depth16 else frames[i].cpu().numpy())
        got = host[int(offs[i]): int(offs[i] + lens[i])].tobytes()
        if got != want:
            raise SystemExit("rank %d frame %d: GPU stream differs from the oracle" % (rank, i))
        checked += 1

    def stream_digests(buf):  # one digest per stream, in order: a reordered or shifted stream cannot pass
        h = buf[: int(offs[-1] + lens[-1])].cpu().numpy()
        return [hashlib.blake2b(h[int(offs[i]): int(offs[i] + lens[i])].tobytes(), digest_size=16).hexdigest() for i in range(F)]

    ref_digests = stream_digests(d_out)
    total_bytes = int(lens.sum())

    for _ in range(args.warmup):
        step()
    if not args.synchronous:  # every submission slot allocates its workspace on first use: not inside the timed region
        subs = [submit(i) for i in range(depth_q)]
        for sub in subs:
            enc.wait_batch(sub)
    enc.set_profiling(not args.no_stage_timing)
    stage_acc = {}
    spans = []  # per timed step: HIP-event span from the step's first kernel to its last byte

    def finish(sub):  # felics_wait_batch: the batch is complete in HBM when this returns
        enc.wait_batch(sub)
        for k, v in enc.stage_ms().items():
            stage_acc[k] = stage_acc.get(k, 0.0) + v
        spans.append(enc.span_ms())

    # The K timed steps go through the two-deep submission queue a streaming caller would use: step i + 1 is
    # queued before step i is waited for, so the GPU classifies / scatters the next batch while it packs the
    # last slices of this one.  Every step runs in full and is complete inside the timed region.
    host_submit_s = 0.0  # host time inside felics_submit_batch_device (queueing a step's launches)
    stats_before = enc.stats()
    group.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if args.synchronous:
        for _ in range(args.steps):
            step()
            for k, v in enc.stage_ms().items():
                stage_acc[k] = stage_acc.get(k, 0.0) + v
            spans.append(enc.span_ms())
    else:
        inflight = []
        for i in range(args.steps):
            if len(inflight) == depth_q:
                finish(inflight.pop(0))
            ts = time.perf_counter()
            inflight.append(submit(i))
            host_submit_s += time.perf_counter() - ts
        while inflight:
            finish(inflight.pop(0))
    torch.cuda.synchronize()
    group.barrier()
    elapsed = time.perf_counter() - t0
    enc.set_profiling(False)
    stats_after = enc.stats()
    # what the timed steps had to redo, if anything (felics_stats: each is a whole batch done again)
    redone = {k: stats_after[k] - stats_before[k] for k in ("scatter_fallbacks", "lookback_fallbacks", "slot_overflows", "tile_overflows")}
    # every rank's own figures (skew between ranks, and which device each one ran on), gathered before the MAX
    per_rank = group.gather_objects({"rank": rank, "local_rank": local, "device": torch.cuda.get_device_name(local),
                                     "device_index": local, "ms_per_step": round(elapsed / max(args.steps, 1) * 1e3, 4),
                                     "frames": F, "first_frame": first_frame})
    elapsed = group.max_over_ranks(elapsed)

    # the timed steps must have produced the same bytes as the checked one (both output buffers)
    for o in (d_outs[: max(1, min(depth_q, args.steps))] if not args.synchronous else d_outs[:1]):
        if stream_digests(o) != ref_digests:
            raise SystemExit("rank %d: output changed between steps" % rank)
    # for the record: a few steps through the blocking entry point (one batch at a time)
    sync_ms = None
    if not args.synchronous and not args.no_blocking_extra and rank == 0:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(3):
            step()
        sync_ms = (time.perf_counter() - t1) / 3 * 1e3

    # Configs 2 and 4 (one 4K gray8 / RGB8 frame per call) ride along with the default run so that they are
    # driver-timed too: a short stream of single-frame submissions through the queue, and blocking calls.
    side = None
    if rank == 0 and args.config == 3 and not args.no_side_configs and not args.depth16 and not args.rgb and (W, H) == (W4K, H4K):
        side = {}
        for name, rgbf in (("config2_one_4k_gray8_frame", False), ("config4_one_4k_rgb8_frame", True)):
            fr = (synth_torch.rgb8(W, H, 0, device=dev) if rgbf else frames[0]).contiguous()
            ch = 3 if rgbf else 1
            outs = [torch.empty(int(npix * ch * 1.25) + (1 << 20), dtype=torch.uint8, device=dev) for _ in range(depth_q)]
            o1, l1 = enc.compress_batch_device(fr.data_ptr(), 1, W, H, int(rgbf), 0, outs[0].data_ptr(), outs[0].numel())
            got = outs[0][int(o1[0]): int(o1[0] + l1[0])].cpu().numpy().tobytes()
            if got != oracle.compress(fr.cpu().numpy()):
                raise SystemExit("%s: GPU stream differs from the oracle" % name)
            reps = 20
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(reps):
                enc.compress_batch_device(fr.data_ptr(), 1, W, H, int(rgbf), 0, outs[0].data_ptr(), outs[0].numel())
            blocking = (time.perf_counter() - t1) / reps
            t1 = time.perf_counter()
            q = []
            for i in range(reps):
                if len(q) == depth_q:
                    enc.wait_batch(q.pop(0))
                q.append(enc.submit_batch_device(fr.data_ptr(), 1, W, H, int(rgbf), 0, outs[i % depth_q].data_ptr(), outs[i % depth_q].numel()))
            while q:
                enc.wait_batch(q.pop(0))
            queued = (time.perf_counter() - t1) / reps
            side[name] = {"ms_per_frame_blocking_call": round(blocking * 1e3, 3), "MPix_s_blocking_call": round(npix / blocking / 1e6, 1),
                          "ms_per_frame_queued": round(queued * 1e3, 3), "MPix_s_queued": round(npix / queued / 1e6, 1),
                          "frac_of_hbm_peak_queued": round(npix * ch / queued / 1e9 / HBM_PEAK_GBS, 5),
                          "calls": reps, "byte_compared_with_oracle": True}

    # End to end with HOST buffers, the reference's own call shape (compress(&self, W): compression.rs:255-282; SURVEY.md section 8(d)
    # asks for this figure next to the HBM-resident one; it is never `value`): the 64 frames in page-locked host memory in, the 64
    # streams in page-locked host memory out, through felics_compress_batch -- chunks on the submission queue, frames going in and
    # streams coming back on copy streams of their own under the kernels.
    if side is not None:
        h_in = frames.cpu().pin_memory()
        slot_h = int(npix * 1.25) + 4096
        h_out = torch.empty((F, slot_h), dtype=torch.uint8).pin_memory()
        in_ptrs = [h_in[i].data_ptr() for i in range(F)]
        out_ptrs = [h_out[i].data_ptr() for i in range(F)]
        lens_h = enc.compress_batch_host(in_ptrs, F, W, H, 0, 0, out_ptrs, [slot_h] * F)  # (also sizes the staging buffers)
        for i in (0, 1, F - 1):
            if h_out[i][: int(lens_h[i])].numpy().tobytes() != oracle.compress(h_in[i].numpy()):
                raise SystemExit("end-to-end leg: stream %d differs from the oracle" % i)
        reps_h = 5
        t1 = time.perf_counter()
        for _ in range(reps_h):
            lens_h = enc.compress_batch_host(in_ptrs, F, W, H, 0, 0, out_ptrs, [slot_h] * F)
        e2e = (time.perf_counter() - t1) / reps_h
        out_bytes = int(lens_h.sum())
        side["end_to_end_host_64_frames"] = {
            "ms_per_batch": round(e2e * 1e3, 3), "MPix_s": round(F * npix / e2e / 1e6, 1),
            "h2d_GBs": round(F * npix / e2e / 1e9, 2), "d2h_GBs": round(out_bytes / e2e / 1e9, 2),
            "note": "felics_compress_batch: %d frames from page-locked host memory, %d stream bytes back to page-locked host memory; "
                    "the link carries %.0f MB in and %.0f MB out per batch (PCIe-inclusive: never `value`)" % (F, out_bytes, F * npix / 1e6, out_bytes / 1e6),
            "streams_byte_compared_with_oracle": 3}
        del h_in, h_out

    # Config 5's per-GPU share (64 4K RGB8 frames per submission) rides along too: three steps through the queue.
    if side is not None:
        F5 = 64
        rgbf = torch.empty((F5, H, W, 3), dtype=torch.uint8, device=dev)
        for i in range(F5):
            rgbf[i] = synth_torch.rgb8(W, H, first_frame + i, device=dev)
        outs5 = [torch.empty(int(F5 * npix * 3 * 1.25) + (1 << 20), dtype=torch.uint8, device=dev) for _ in range(depth_q)]
        o5, l5 = enc.compress_batch_device(rgbf.data_ptr(), F5, W, H, 1, 0, outs5[0].data_ptr(), outs5[0].numel())
        got = outs5[0][int(o5[3]): int(o5[3] + l5[3])].cpu().numpy().tobytes()
        if got != oracle.compress(rgbf[3].cpu().numpy()):
            raise SystemExit("config 5: GPU stream differs from the oracle")
        for i in range(depth_q):  # every lane's workspace for this shape
            enc.wait_batch(enc.submit_batch_device(rgbf.data_ptr(), F5, W, H, 1, 0, outs5[i].data_ptr(), outs5[i].numel()))
        reps5 = 4
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        q = []
        for i in range(reps5):
            if len(q) == depth_q:
                enc.wait_batch(q.pop(0))
            q.append(enc.submit_batch_device(rgbf.data_ptr(), F5, W, H, 1, 0, outs5[i % depth_q].data_ptr(), outs5[i % depth_q].numel()))
        while q:
            enc.wait_batch(q.pop(0))
        t5 = (time.perf_counter() - t1) / reps5
        side["config5_share_64_4k_rgb8_frames"] = {"ms_per_step_queued": round(t5 * 1e3, 3), "MPix_s": round(F5 * npix / t5 / 1e6, 1),
                                                   "frac_of_hbm_peak": round(F5 * npix * 3 / t5 / 1e9 / HBM_PEAK_GBS, 5), "steps": reps5,
                                                   "byte_compared_with_oracle": True,
                                                   "note": "one GPU's share of config 5 (512 frames over 8 GPUs); --config 5 --gpus N times the sharded job"}
        del rgbf, outs5
        # 16-bit samples (traits.rs:35-43: half of the trait's surface) ride along as well: 16 4K gray16 frames per submission on a
        # context of its own with four submissions in flight (the 16-bit kernels wait for memory more than for the ALUs:
        # profiles/r03/experiments.txt), one stream byte-compared with the oracle, three timed steps.
        from felics_amd import synth

        F16 = 16
        base16 = [torch.from_numpy(synth.gray16(W, H, first_frame + i).view(np.int16)) for i in range(4)]
        fr16 = torch.stack([base16[i % 4] for i in range(F16)]).to(dev)
        prev_lanes = os.environ.get("FELICS_LANES")  # (the user's, or --lanes': put back afterwards)
        os.environ["FELICS_LANES"] = "4"
        try:
            enc16 = felics_amd.Encoder(local)
        finally:
            if prev_lanes is None:
                del os.environ["FELICS_LANES"]
            else:
                os.environ["FELICS_LANES"] = prev_lanes
        q16 = enc16.lane_count()
        outs16 = [torch.empty(int(F16 * npix * 2 * 1.25) + (1 << 20), dtype=torch.uint8, device=dev) for _ in range(q16)]
        o16, l16 = enc16.compress_batch_device(fr16.data_ptr(), F16, W, H, 0, 1, outs16[0].data_ptr(), outs16[0].numel())
        got = outs16[0][int(o16[1]): int(o16[1] + l16[1])].cpu().numpy().tobytes()
        if got != oracle.compress(fr16[1].cpu().numpy().view(np.uint16)):
            raise SystemExit("16-bit leg: GPU stream differs from the oracle")
        for i in range(q16):  # every lane's workspace
            enc16.wait_batch(enc16.submit_batch_device(fr16.data_ptr(), F16, W, H, 0, 1, outs16[i].data_ptr(), outs16[i].numel()))
        reps16 = 8
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        q = []
        for i in range(reps16):
            if len(q) == q16:
                enc16.wait_batch(q.pop(0))
            q.append(enc16.submit_batch_device(fr16.data_ptr(), F16, W, H, 0, 1, outs16[i % q16].data_ptr(), outs16[i % q16].numel()))
        while q:
            enc16.wait_batch(q.pop(0))
        t16 = (time.perf_counter() - t1) / reps16
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(3):
            enc16.compress_batch_device(fr16.data_ptr(), F16, W, H, 0, 1, outs16[0].data_ptr(), outs16[0].numel())
        b16 = (time.perf_counter() - t1) / 3
        side["gray16_16_4k_frames"] = {"ms_per_step_queued": round(t16 * 1e3, 3), "ms_per_blocking_call": round(b16 * 1e3, 3),
                                       "MPix_s": round(F16 * npix / t16 / 1e6, 1), "frac_of_hbm_peak": round(F16 * npix * 2 / t16 / 1e9 / HBM_PEAK_GBS, 5),
                                       "submissions_in_flight": q16, "steps": reps16, "dtype": "u16", "byte_compared_with_oracle": True,
                                       "note": "16 synthetic 4K gray16 frames per step (2 B per pixel against the HBM-read roofline); not the headline"}
        enc16.close()
        del fr16, outs16
    # Decode, for the record: the batch's streams through the GPU decoder (one wave per stream: the
    # format is bit-serial per stream) and a sample of them through the host decoder on this box's cores.
    decode = None
    if rank == 0 and args.config == 3 and not args.no_decode_leg and not args.depth16 and not args.rgb:
        from concurrent.futures import ThreadPoolExecutor

        from felics_amd import api as fapi2

        offs0, lens0 = step()
        d_px = torch.empty_like(frames)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        _, st = enc.decompress_batch_device(d_out.data_ptr(), offs0, lens0, d_px.data_ptr(), d_px.numel())
        torch.cuda.synchronize()
        gpu_s = time.perf_counter() - t1
        if not bool((d_px == frames).all()):
            raise SystemExit("GPU decoder: pixels differ from the frames that were encoded")
        del d_px
        hostbuf = d_out[: int(offs0[-1] + lens0[-1])].cpu().numpy()
        sample = [hostbuf[int(offs0[i]): int(offs0[i] + lens0[i])].tobytes() for i in range(min(F, 16))]
        t1 = time.perf_counter()
        fapi2.decompress_bytes(sample[0])
        one = time.perf_counter() - t1
        cores = min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
        t1 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=cores) as ex:
            list(ex.map(fapi2.decompress_bytes, sample))
        many = time.perf_counter() - t1
        # a batch of thousands of streams (the 64, referenced 64 times over: every decoded image has a buffer of its own): the
        # library switches to 64 streams per wave (lane = stream), whose rate per stream does not depend on the batch
        big = None
        try:
            NB = 4096
            o_big = np.array([offs0[i % F] for i in range(NB)], dtype=np.uint64)
            l_big = np.array([lens0[i % F] for i in range(NB)], dtype=np.uint64)
            d_big = torch.empty((NB, H, W), dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            _, stb = enc.decompress_batch_device(d_out.data_ptr(), o_big, l_big, d_big.data_ptr(), d_big.numel())
            torch.cuda.synchronize()
            big_s = time.perf_counter() - t1
            if not (stb == 0).all() or not bool((d_big[NB - 1] == frames[(NB - 1) % F]).all()) or not bool((d_big[F + 1] == frames[1]).all()):
                raise SystemExit("GPU decoder (64 streams per wave): pixels differ from the frames that were encoded")
            del d_big
            big = {"streams": NB, "gpu_MPix_s": round(NB * npix / big_s / 1e6, 1), "gpu_seconds_per_batch": round(big_s, 3),
                   "MPix_s_per_stream": round(npix / big_s / 1e6, 3), "form": "64 streams per wave (lane = stream)"}
        except (torch.OutOfMemoryError, felics_amd.FelicsError):  # no room for 34 GB of frames (or for the library's tables): no large-batch figure
            big = None
        # the same form on RGB8 streams (1920x1080 frames: 4096 4K RGB frames and their int16 planes would not fit): three planes
        # per lane from one bit reader
        big_rgb = None
        try:
            from felics_amd import synth_torch as st2

            WR, HR, FR, NR = 1920, 1080, 16, 4096
            fr = torch.stack([st2.rgb8(WR, HR, f, device=dev) for f in range(FR)])
            capr = int(FR * WR * HR * 3 * 1.25) + (1 << 20)
            d_or = torch.empty(capr, dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()  # (the frames are torch's work on torch's stream; the library runs on streams of its own)
            offr, lenr = enc.compress_batch_device(fr.data_ptr(), FR, WR, HR, 1, 0, d_or.data_ptr(), capr)
            o_r = np.array([offr[i % FR] for i in range(NR)], dtype=np.uint64)
            l_r = np.array([lenr[i % FR] for i in range(NR)], dtype=np.uint64)
            d_r = torch.empty((NR, HR, WR, 3), dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            _, str_ = enc.decompress_batch_device(d_or.data_ptr(), o_r, l_r, d_r.data_ptr(), d_r.numel())
            torch.cuda.synchronize()
            rgb_s = time.perf_counter() - t1
            if not (str_ == 0).all() or not bool((d_r[NR - 1] == fr[(NR - 1) % FR]).all()) or not bool((d_r[FR + 1] == fr[1]).all()):
                raise SystemExit("GPU decoder (64 RGB streams per wave): pixels differ from the frames that were encoded (statuses nonzero: %d; frames "
                                 "differing among the first 128: %s)" % (int((np.asarray(str_) != 0).sum()),
                                                                          [i for i in range(128) if not bool((d_r[i] == fr[i % FR]).all())]))
            del d_r, d_or, fr
            big_rgb = {"streams": NR, "frame": "%dx%d RGB8" % (WR, HR), "gpu_MPix_s": round(NR * WR * HR / rgb_s / 1e6, 1),
                       "gpu_MSamples_s": round(3 * NR * WR * HR / rgb_s / 1e6, 1), "gpu_seconds_per_batch": round(rgb_s, 3),
                       "form": "64 streams per wave (lane = stream), three planes per lane"}
        except (torch.OutOfMemoryError, felics_amd.FelicsError):
            big_rgb = None
        decode = {"gpu_MPix_s": round(F * npix / gpu_s / 1e6, 1), "gpu_seconds_per_batch": round(gpu_s, 3), "streams": F,
                  "gpu_note": "felics_decompress_batch_device: one wave per stream, %d streams = %d waves on 256 CUs" % (F, F),
                  "large_batch": big, "large_batch_rgb": big_rgb,
                  "host_MPix_s_1_core": round(npix / one / 1e6, 1),
                  "host_MPix_s_%d_cores" % cores: round(len(sample) * npix / many / 1e6, 1),
                  "host_sample": "%d of the batch's streams, felics_decompress (C++)" % len(sample), "pixels_checked": True}
        if big:  # a lane decodes at one rate whatever the batch: the batch from which the device beats this box's host cores
            decode["break_even_streams"] = int(len(sample) * npix / many / 1e6 / big["MPix_s_per_stream"]) + 1
            decode["break_even_note"] = ("streams per call from which 64 streams per wave beat %d host cores (host rate / rate per stream); "
                                         "one wave per stream: ~540, saturating at ~2.3 GPix/s (profiles/r04/decode_scaling.txt)" % cores)

    if rank == 0:
        steps = max(args.steps, 1)
        ms_per_step = elapsed / steps * 1e3
        value = world * F * npix * steps / elapsed / 1e6  # MPix/s, whole job
        stage_ms = {k: v / steps for k, v in stage_acc.items()}
        # The dominant kernel: largest summed launch duration per step, no kernel left out.  (That can be the spine, whose
        # launches are a dependency chain of a few hundred long-lived waves running underneath the others: its duration is a
        # latency, not a share of the machine.  roofline.by_kernel lists every stage the same way, so the largest kernel that
        # fills the GPU can be read next to it.)
        cand = stage_ms
        dom = max(cand, key=cand.get) if cand else None
        alg_bytes = F * npix * channels * sample_bytes  # 1 B/pixel/channel read (2 for 16-bit), SURVEY.md §8(d)
        # A step launches most kernels once per slice of the images; stage_ms[k] is the sum of the
        # durations of kernel k's launches in one step (HIP events on the stream each launch runs on).
        launches = max(1, enc.stage_launches().get(dom, 1)) if dom else 1
        roofline = None

        def kernel_of(stage):  # the kernel a stage name stands for on the path that ran
            if stage == "pack":
                if args.depth16:
                    return "k_pack"
                return "k_pack" if os.environ.get("FELICS_TWO_PASS") else "k_pack_t"
            if stage == "offsets" and not args.depth16:
                return "k_enum"
            if stage == "spine":
                return "k_spine3"
            if stage == "zero":
                return "k_finish_sizes+k_join_edges"
            if stage == "assign" and not args.depth16:
                return "k_assign3"
            if stage == "scatter" and not args.depth16:
                return "k_front"
            return "k_" + stage

        if dom and stage_ms[dom] > 0:
            per_launch_ms = stage_ms[dom] / launches
            per_launch_bytes = alg_bytes / launches
            achieved = per_launch_bytes / (per_launch_ms * 1e-3) / 1e9
            # PMC figures (FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU passes of profiles/tools/collect.sh) are a
            # property of a build on this exact workload: used only if the file was made from the sources
            # this library was built from, otherwise null.
            traffic = valu = None
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            headline = not args.rgb and not args.depth16 and args.kind == "S1" and F == 64 and (W, H) == (W4K, H4K)
            if os.path.exists(tpath) and headline:
                from felics_amd import build as fbuild

                tj = json.load(open(tpath))
                if tj.get("_source_sha256") == fbuild.source_hash():
                    t = tj.get(kernel_of(dom), {}).get("hbm_bytes_per_step")
                    traffic = int(t / launches) if t else None
                    insts = tj.get("_valu_wave_insts_per_step")
                    if insts:  # wave64 VALU instructions per step against 1024 SIMDs x 2.4 GHz / 2 cycles per instruction
                        # the chip's sustained rate of wave64 integer VALU instructions, measured (profiles/tools/micro/valu_rate.hip,
                        # profiles/r03/valu_rate.txt: 1.71-1.73 ns per instruction and SIMD with every SIMD issuing), not a data-sheet figure
                        # the step's vector instructions priced by class (profiles/r04/valu_rate.txt: 1.0 ns per wave64 instruction and SIMD for
                        # add / sub / logic / shift-right / bitop3 / 16-bit VOP2, 1.7 ns for the rest; profiles/r04/opcodes.txt: each kernel's
                        # share of cheap instructions from the shipped code object) -- an additive figure: mixed streams cost more
                        # (valu_rate_mixed_streams.txt), so it is a lower bound of the issue time
                        weighted_ns = tj.get("_valu_weighted_ns_per_step")
                        # Two ceilings, both reported: the guide's issue rate (a wave64 VALU instruction every 2 cycles per SIMD:
                        # 1024 SIMDs x 2.4 GHz / 2 = 1.23 T/s -- what packed / dual-issued fp32 reaches) and the rate integer
                        # instructions were measured at on this chip (4 cycles: 0.595 T/s for the 1.7 ns class)
                        valu = {"wave_insts_per_step": int(insts), "issue_peak_per_s": 0.595e12, "issue_peak": "measured, profiles/r04/valu_rate.txt (1.7 ns class)",
                                "frac_of_issue_peak": round(insts / 0.595e12 / (ms_per_step * 1e-3), 4),
                                "datasheet_issue_peak_per_s": 1.2288e12,
                                "frac_of_datasheet_issue_peak": round(insts / 1.2288e12 / (ms_per_step * 1e-3), 4),
                                "weighted_issue_ms_per_step": None if not weighted_ns else round(weighted_ns / 1024 / 1e6, 4),
                                "weighted_frac_of_step": None if not weighted_ns else round(weighted_ns / 1024 / 1e6 / ms_per_step, 4)}
            roofline = {"bound": "hbm", "kernel": kernel_of(dom), "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                        "algorithmic_bytes_per_launch": int(per_launch_bytes), "avg_launch_ms": round(per_launch_ms, 4),
                        "launches_per_step": launches, "valu": valu}
            by_kernel = {}
            for k, v in stage_ms.items():
                n = max(1, enc.stage_launches().get(k, 1))
                if v > 0 and k in ("zero", "offsets"):
                    # bookkeeping over counts and tile words, not over pixels: the algorithmic pixel bytes do not apply
                    by_kernel[kernel_of(k)] = {"ms_per_step_sum_of_launches": round(v, 4), "launches_per_step": n,
                                               "achieved_GBs": None, "frac": None, "note": "reads no pixels"}
                elif v > 0:
                    g = alg_bytes / n / (v / n * 1e-3) / 1e9
                    by_kernel[kernel_of(k)] = {"ms_per_step_sum_of_launches": round(v, 4), "launches_per_step": n,
                                               "achieved_GBs": round(g, 2), "frac": round(g / HBM_PEAK_GBS, 5)}
            roofline["by_kernel"] = by_kernel
            # Two readings of "dominant".  The largest SUM of launch durations is the spine: a latency chain of a few hundred waves that
            # runs underneath the others, with as many of its launches side by side as there are submissions in flight -- so its
            # launches can sum to more than ms_per_step.  The largest kernel that FILLS the GPU is the one to read a share of the
            # machine from.
            roofline["latency_chain_kernel"] = {"kernel": kernel_of("spine"), "concurrent_launches": depth_q,
                                                "note": "launches of %d submissions overlap: the sum of launches per step can exceed ms_per_step" % depth_q,
                                                **by_kernel.get(kernel_of("spine"), {})}
            filling = {k: v for k, v in by_kernel.items() if k != kernel_of("spine")}
            if filling:
                big = max(filling, key=lambda k: filling[k]["ms_per_step_sum_of_launches"])
                roofline["largest_gpu_filling_kernel"] = {"kernel": big, **filling[big]}
        # the same figure for every stage (HIP-event brackets; on the low-priority streams they include the wait
        # for free compute resources, which rocprof's kernel begin / end timestamps do not)
        per_stage = {}
        for k, v in stage_ms.items():
            n = max(1, enc.stage_launches().get(k, 1))
            if v > 0:
                per_stage[kernel_of(k)] = {"avg_launch_ms": round(v / n, 4), "launches_per_step": n,
                                       "achieved_GBs": None if k in ("zero", "offsets") else round(alg_bytes / n / (v / n * 1e-3) / 1e9, 2)}
        pipeline_gbs = alg_bytes / (ms_per_step * 1e-3) / 1e9
        cpu1 = cpum = None
        if args.cpu_seconds > 0:
            sample = [frames[i].cpu().numpy().view(np.uint16) if args.depth16 else frames[i].cpu().numpy()
                      for i in range(min(F, 16))]
            cpu1, cpum = cpu_baseline(sample, args.cpu_seconds)
            cpu1["value"] = round(cpu1["value"], 2)
            cpum["value"] = round(cpum["value"], 2)
        line = {
            "metric": "encode MPix/s on 4K 8-bit grayscale batch (bit-exact); % HBM-read roofline",
            "value": round(value, 1), "unit": "MPix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            # BASELINE.md section 2's protocol next to the pipelined mean: per step, HIP events from the step's first kernel to
            # its stream sizes on the host; median over the timed steps.  With several steps in flight a step's span covers
            # the time it shares the GPU with its neighbours, so the median span exceeds ms_per_step.
            "median_ms_per_step_event_span": round(sorted(spans)[len(spans) // 2], 3) if spans and not args.no_stage_timing else None,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u16" if args.depth16 else "u8", "data": "synthetic",
            "config": {"workload": "batch of %d synthetic %s %dx%d %d-bit %s frames per GPU, resident in HBM"
                                   % (F, "S1-RGB" if args.rgb else args.kind, W, H, 16 if args.depth16 else 8,
                                      "RGB" if args.rgb else "grayscale"),
                       "content": {"S1": "S1 natural-like synthetic (BASELINE.md section 2): smooth ramps + 3 bits of noise; the figure holds for "
                                         "this content only (bits_per_pixel under `parity`); other contents, measured per round: "
                                         "profiles/r05/content_sensitivity.txt",
                                   "S2": "S2 uniform noise (worst case)", "S3": "S3 flat (best case)"}[args.kind],
                       "baseline_config": args.config, "frames_per_gpu": F, "width": W, "height": H, "channels": channels,
                       "sharding": "frames split across ranks, no collective"},
            "roofline": roofline,
            "cpu_baseline": cpu1,
            "cpu_baseline_all_cores": cpum,
            "other_configs": side,
            "decode": decode,
            "pipeline": {"achieved_GBs": round(pipeline_gbs, 2), "frac_of_hbm_peak": round(pipeline_gbs / HBM_PEAK_GBS, 5),
                         "stage_ms_sum_of_launches": {k: round(v, 4) for k, v in stage_ms.items()},
                         "per_stage": per_stage,
                         "submission": "blocking calls" if args.synchronous else "%d batches in flight (submit ahead, wait in order)" % depth_q,
                         "ms_per_step_blocking_calls": None if sync_ms is None else round(sync_ms, 3),
                         "host_ms_per_submit": round(host_submit_s / steps * 1e3, 3),
                         "batches_redone_in_timed_steps": redone,
                         "note": "the stages follow each other slice by slice on HIP streams of their own; launches overlap, so the sums exceed ms_per_step"},
            "per_rank": per_rank,
            "parity": {"frames_byte_compared_with_oracle": checked, "streams_digest_checked_after_timed_steps": F,
                       "compressed_bytes_per_step_rank0": total_bytes,
                       "bits_per_pixel": round(total_bytes * 8 / (F * npix), 4)},
        }
        print(json.dumps(line), flush=True)
    enc.close()
    group.close()


if __name__ == "__main__":
    main()
