"""Turns the rocprofv3 output directories of profiles/tools/collect.sh into the small files kept under profiles/."""
import collections
import csv
import glob
import json
import sys

out = sys.argv[1]


def short(name):
    return name.split("felics::")[1].split("<")[0].split("(")[0] if "felics::" in name else None


# kernel_stats.csv: calls / total / average duration of the felics:: kernels
rows = []
for f in glob.glob(out + "/kt/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "felics::" in r["Name"]:
            rows.append(r)
if rows:
    with open(out + "/kernel_stats.csv", "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows)

# traffic.json: bytes per step per kernel.  The profiled command runs the step twice (one checked step, one
# timed step), so the sums are halved.  FETCH_SIZE / WRITE_SIZE are in KB.
steps_in_command = 2
traffic = collections.defaultdict(lambda: {"fetch_bytes_per_step": 0, "write_bytes_per_step": 0})
for kind, key in (("fetch", "fetch_bytes_per_step"), ("write", "write_bytes_per_step")):
    for f in glob.glob(out + "/" + kind + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k:
                traffic[k][key] += float(r["Counter_Value"]) * 1024 / steps_in_command
res = {}
alias = {"k_pack_fused": "k_pack"}  # bench.py names stages, not kernels
for k, v in sorted(traffic.items()):
    v = {a: int(b) for a, b in v.items()}
    v["hbm_bytes_per_step"] = v["fetch_bytes_per_step"] + v["write_bytes_per_step"]
    res[alias.get(k, k)] = v
res["_note"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `bench.py --steps 1 --warmup 0` "
                "(64 S1 4K gray frames), KB counters x 1024, halved because that command runs the step twice. Raw "
                "counters: on gfx950 FETCH_SIZE can read half the bytes of wide (16 B/lane) coalesced loads "
                "(MI355X_MICROARCH.md, HBM); these kernels mix 1-, 4- and 16-byte accesses and were not calibrated. "
                "k_pack = k_pack_fused (the single-pass pack).")
if traffic:
    json.dump(res, open(out + "/traffic.json", "w"), indent=1)
print("wrote", out)
