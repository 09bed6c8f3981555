"""Turns the rocprofv3 output directories of profiles/tools/collect.sh into the small files kept under profiles/.

traffic.json: per kernel, per step (64 S1 4K gray frames): FETCH_SIZE and WRITE_SIZE (KB counters x 1024), the
correction MI355X_MICROARCH.md (HBM) prescribes, and the SQ instruction counters.
  * FETCH_SIZE reads exactly half the bytes of wide (16 B per lane) coalesced loads on gfx950.  Kernels whose reads are such
    loads get fetch x 2 ("x2: uint4 loads").  k_front reads the input with unaligned 4- / 8-byte loads per lane, several trips
    in flight: the counter under-reports those too (round 4's k_hist, whose only reads were the 530.8 MB of input, reported
    ~ 334 MB).  k_front reads nothing but the input from HBM -- every pixel once, the row above it from cache -- so its fetch
    is taken as max(raw, input bytes) ("input: ..."), the factor input / raw kept as _fetch_calibration_factor.  k_pack_t reads
    its pixels with 16-byte loads per lane (counted at half) next to dword / 8-byte reads of k and pixel offsets: raw + the
    input size.
  * WRITE_SIZE is exact for full-line stores; partial-line stores (scatter) count the written sectors.
"""
import collections
import csv
import glob
import json
import os
import sys

out = sys.argv[1]
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
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
INPUT_BYTES = 64 * 3840 * 2160
X2 = {"k_spine3": "x2: uint4 loads of the records' events (the 8-byte descriptors are a tenth of them)",
      "k_pack": "x2: uint4 staging", "k_lengths": "x2: uint4 staging", "k_concat_planes": "x1",
      "k_pack_t": "raw + input bytes: the 16-byte loads of the pixels (twice the input size: the group and the row above it, counted at half) are its only wide loads; k and pixel offsets are read as dwords / 8 bytes",
      "k_assign3": "x2: uint4 loads of the records' states and events",
      "k_front": "input: max(raw, input bytes): the kernel reads every pixel exactly once (unaligned 4-byte loads, counted low) and nothing else"}
res = {}
total = 0
valu_total = 0
valu_weighted_ns = 0.0
front_raw = counters.get("k_front", {}).get("FETCH_SIZE", 0.0) * 1024
calib = INPUT_BYTES / front_raw if front_raw else 1.0
try:  # each kernel's share of 1.0 ns instructions, from the library this run used (profiles/tools/opcodes.py)
    import opcodes
    from felics_amd import build as fbuild_

    shares = {}
    for name, ctr in opcodes.histograms(fbuild_.ensure_lib()).items():
        kshort = short(name)
        if kshort and ("unsigned char" in name or "<" not in name):
            shares[kshort] = opcodes.cheap_share(ctr)[1]
except Exception as e:  # (no llvm-objdump on the path: the weighted figure is left out)
    shares = None
    print("opcode shares unavailable:", e)
for k, c in sorted(counters.items()):
    fetch = c.get("FETCH_SIZE", 0.0) * 1024
    write = c.get("WRITE_SIZE", 0.0) * 1024
    rule = X2.get(k, "x1: dword-or-narrower loads per lane")
    fetch_c = fetch * (2 if rule.startswith("x2") else 1)
    if rule.startswith("input"):
        fetch_c = max(fetch, INPUT_BYTES)
    if rule.startswith("raw + input bytes"):
        fetch_c = fetch + INPUT_BYTES  # two spans of the input's size read with 16-byte loads, each counted at half
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
    if shares is not None and "SQ_INSTS_VALU" in c:
        sh = shares.get(k, 0.0)
        e["valu_share_1ns_class"] = round(sh, 3)
        valu_weighted_ns += c["SQ_INSTS_VALU"] * (sh * 1.0 + (1.0 - sh) * 1.72)
if counters:
    from felics_amd import build as fbuild

    res["_total_hbm_bytes_per_step"] = int(total)
    res["_valu_wave_insts_per_step"] = int(valu_total)
    if shares is not None:
        res["_valu_weighted_ns_per_step"] = int(valu_weighted_ns)  # SIMD-nanoseconds: / 1024 SIMDs = the step's additive issue time
    res["_fetch_calibration_factor"] = round(calib, 4)
    res["_source_sha256"] = fbuild.source_hash()
    res["_workload"] = "64 synthetic S1 3840x2160 gray8 frames per step (bench.py default), blocking call"
    res["_note"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_* (two groups) in four separate counter-only passes of `bench.py --steps 1 --warmup 0 "
                    "--synchronous ...` (one checked + one timed step: sums halved).  bench.py uses this file only while "
                    "_source_sha256 equals the hash of the native sources it runs (felics_amd/build.py: source_hash).")
    json.dump(res, open(out + "/traffic.json", "w"), indent=1)
print("wrote", out)
