"""Turns the rocprofv3 output directories of profiles/tools/collect.sh into the small files kept under profiles/.

traffic.json: per kernel, per step (64 S1 4K gray frames): FETCH_SIZE and WRITE_SIZE (KB counters x 1024), the
correction MI355X_MICROARCH.md (HBM) prescribes, and the SQ instruction counters.
  * FETCH_SIZE reads exactly half the bytes of wide (16 B per lane) coalesced loads on gfx950.  Kernels whose reads are such
    loads get fetch x 2 ("x2: uint4 loads"); kernels that read one dword or less per lane are exact ("x1"), which
    round 1 confirmed on this workload (k_hist read the 530.8 MB of input once with dword loads and reported 0.53 GB; k_spine's
    uint4 loads of 0.29 GB of events reported 0.146 GB).  k_pack_k mixes uint4 staging of the pixels with byte / dword reads
    of the events: its figure is the raw counter plus the input size once more (the half the counter misses).
  * WRITE_SIZE is exact for full-line stores; partial-line stores (scatter) count the written sectors.
"""
import collections
import csv
import glob
import json
import os
import sys

out = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


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
        fh.write("kernel,calls,total_ns,average_ns,percent\n")
        for r in rows:
            fh.write("%s,%s,%s,%s,%s\n" % (short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]))

steps_in_command = 2  # the profiled command runs one checked and one timed step
counters = collections.defaultdict(lambda: collections.defaultdict(float))
for kind in ("fetch", "write", "sq", "sq2"):
    for f in glob.glob(out + "/" + kind + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k:
                counters[k][r["Counter_Name"]] += float(r["Counter_Value"]) / steps_in_command

# which FETCH_SIZE rule applies to which kernel (see the module docstring)
X2 = {"k_spine": "x2: uint4 loads of the event blocks", "k_pack_fused": "x2: uint4 staging of pixels and k",
      "k_pack": "x2: uint4 staging", "k_lengths": "x2: uint4 staging", "k_concat_planes": "x1",
      "k_pack_k": "raw + input bytes: the uint4 staging of the pixels (once the input size) is its only wide load; events are read as bytes / dwords",
      "k_pack_g": "raw + input bytes: the uint4 staging of the pixels (once the input size) is its only wide load; k, offsets and the run table are read as bytes / words / dwords",
      "k_assign_serial": "x2: uint4 loads of the events and the block states"}
res = {}
total = 0
valu_total = 0
for k, c in sorted(counters.items()):
    fetch = c.get("FETCH_SIZE", 0.0) * 1024
    write = c.get("WRITE_SIZE", 0.0) * 1024
    rule = X2.get(k, "x1: dword-or-narrower loads per lane")
    fetch_c = fetch * (2 if rule.startswith("x2") else 1)
    if rule.startswith("raw + input bytes"):
        fetch_c = fetch + 64 * 3840 * 2160  # the workload's input, read once with 16-byte loads and counted at half
    e = {"fetch_bytes_raw": int(fetch), "fetch_rule": rule, "fetch_bytes_per_step": int(fetch_c), "write_bytes_per_step": int(write),
         "hbm_bytes_per_step": int(fetch_c + write)}
    for name in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_LDS",
                 "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM",
                 "SQ_INSTS_BRANCH"):
        if name in c:
            e[name] = int(c[name])
    res[k] = e
    total += e["hbm_bytes_per_step"]
    valu_total += c.get("SQ_INSTS_VALU", 0.0)
if counters:
    from felics_amd import build as fbuild

    res["_total_hbm_bytes_per_step"] = int(total)
    res["_valu_wave_insts_per_step"] = int(valu_total)
    res["_source_sha256"] = fbuild.source_hash()
    res["_workload"] = "64 synthetic S1 3840x2160 gray8 frames per step (bench.py default), blocking call"
    res["_note"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_* (two groups) in four separate counter-only passes of `bench.py --steps 1 --warmup 0 "
                    "--synchronous ...` (one checked + one timed step: sums halved).  bench.py uses this file only while "
                    "_source_sha256 equals the hash of the native sources it runs (felics_amd/build.py: source_hash).")
    json.dump(res, open(out + "/traffic.json", "w"), indent=1)
print("wrote", out)
