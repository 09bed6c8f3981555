"""The multi-GPU path is "shard the frames, no collective": world_size-2 gloo run on the CPU of
what bench.py does around the encoder (shard ranges, barrier, MAX of the time, gathered results),
with the oracle standing in for the GPU encoder so the streams themselves are checked too."""
import hashlib
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions():
    from felics_amd.dist import shard_range

    for total in (0, 1, 7, 64, 512, 513):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


WORKER = textwrap.dedent("""
    import hashlib, sys, time
    sys.path.insert(0, %r)
    from felics_amd import dist as fdist, synth
    from tests import oracle_lib
    g = fdist.Group("gloo")
    oracle = oracle_lib.load()
    total = 10
    first, last = fdist.shard_range(total, g.rank, g.world)
    g.barrier()
    t0 = time.perf_counter()
    digests = {}
    for f in range(first, last):
        digests[f] = hashlib.sha256(oracle.compress(synth.gray8(96, 64, f, "S1"))).hexdigest()
    time.sleep(0.05 * (g.rank + 1))
    g.barrier()
    mine = time.perf_counter() - t0
    worst = g.max_over_ranks(mine)
    frames = g.sum_over_ranks(last - first)
    everything = g.gather_objects(digests)
    assert worst >= mine - 1e-9 and worst >= 0.1
    assert frames == total
    merged = {}
    for d in everything:
        assert not (set(d) & set(merged))
        merged.update(d)
    assert sorted(merged) == list(range(total))
    if g.rank == 0:
        print("DIGEST", hashlib.sha256("".join(merged[i] for i in range(total)).encode()).hexdigest())
    g.close()
""")


def test_two_ranks_gloo(tmp_path):
    from felics_amd import synth
    from tests import oracle_lib

    oracle = oracle_lib.load()
    want = hashlib.sha256("".join(
        hashlib.sha256(oracle.compress(synth.gray8(96, 64, f, "S1"))).hexdigest() for f in range(10)).encode()).hexdigest()
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                       capture_output=True, text=True, timeout=240, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert ("DIGEST " + want) in r.stdout


def test_bench_multi_gpu_command_cannot_degrade():
    """`python bench.py --gpus 2` outside torchrun starts two ranks itself (as a child process) or fails; it
    never prints a line for fewer GPUs than asked.  No GPU here: the ranks must fail and the command with them."""
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--cpu-seconds", "0"], capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    import torch

    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        assert r.returncode == 0 and '"n_gpus": 2' in r.stdout
    else:
        assert r.returncode != 0, r.stdout[-2000:]
        assert '"n_gpus"' not in r.stdout
    # a rank count that disagrees with --gpus is refused as well
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1"], capture_output=True,
                       text=True, timeout=300, cwd=ROOT, env=env2)
    assert r.returncode != 0 and '"n_gpus"' not in r.stdout
