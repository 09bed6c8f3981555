#!/usr/bin/env python3
"""Same-box A/B of builds of libfelics.so (profiles/tools/variant.sh): runs ON THE GPU BOX from the repository root.

    python3 profiles/tools/ab.py [--rounds 2] [--steps 30] [--serial] [--bench-args "..."] name[:ENV=V,ENV=V] ...

Every name is a directory under felics_amd/_variants/ ("tree" = the tree's own build); the runs alternate round by round, each in
a process of its own.  Per run: ms per queued step, ms per blocking call, the sums of the stages' launches; with --serial also
every stage alone (FELICS_SERIAL=1 FELICS_SLICES=1, blocking calls).  Outputs are digest-checked by bench.py as always.
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(name, env_extra, bench_args, serial):
    env = dict(os.environ)
    if name != "tree":
        env["FELICS_LIB_PATH"] = os.path.join(ROOT, "felics_amd", "_variants", name, "libfelics.so")
    env.update(env_extra)
    args = list(bench_args)
    if serial:
        env.update({"FELICS_SERIAL": "1", "FELICS_SLICES": "1"})
        args += ["--synchronous"]
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-seconds", "0", "--no-side-configs", "--no-decode-leg"] + args,
                       env=env, capture_output=True, text=True)
    if p.returncode != 0:
        return {"error": (p.stderr or p.stdout).strip().splitlines()[-1:]}
    d = json.loads(p.stdout.strip().splitlines()[-1])
    st = {k: round(v, 3) for k, v in d["pipeline"]["stage_ms_sum_of_launches"].items() if v > 0}
    return {"ms_per_step": d["ms_per_step"], "blocking": d["pipeline"]["ms_per_step_blocking_calls"], "stages": st}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--serial", action="store_true", help="also time every stage alone")
    ap.add_argument("--bench-args", default="")
    ap.add_argument("names", nargs="+")
    a = ap.parse_args()
    variants = []
    for spec in a.names:
        name, _, envs = spec.partition(":")
        variants.append((spec, name, dict(e.split("=", 1) for e in envs.split(",") if e)))
    bench_args = ["--steps", str(a.steps), "--warmup", "2"] + a.bench_args.split()
    for r in range(a.rounds):
        for spec, name, env in variants:
            print("%-40s queued %s" % (spec, json.dumps(run(name, env, bench_args, False))), flush=True)
            if a.serial and r == 0:
                print("%-40s alone  %s" % (spec, json.dumps(run(name, env, ["--steps", "4", "--warmup", "1"] + a.bench_args.split(), True))), flush=True)


if __name__ == "__main__":
    main()
