#!/bin/bash
# Runs ON THE GPU BOX from the repository root:  bash profiles/tools/kstats.sh <tag> <bench.py arguments...>
# rocprofv3 --kernel-trace --stats of `python3 bench.py <arguments>` -> gpurun_out/<tag>/kernel_stats.csv
# (felics:: kernels only: name, calls, total ns, average ns, %).
set -eo pipefail
tag=$1
shift
R=$(pwd)
O=$R/gpurun_out/$tag
mkdir -p "$O"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt" -- python3 "$R/bench.py" "$@" > "$O/kt.log" 2>&1 || { tail -20 "$O/kt.log"; exit 1; }
cd "$R"
f=$(find "$O/kt" -name '*kernel_stats.csv' | head -1)
if [ -z "$f" ]; then echo "no kernel_stats.csv"; tail -5 "$O/kt.log"; exit 1; fi
python3 - "$f" "$O/kernel_stats.csv" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "felics" in r["Name"]]
with open(sys.argv[2], "w") as o:
    o.write("kernel,calls,total_ns,average_ns,percent\n")
    for r in rows:
        name = r["Name"].split("felics::")[1].split("(")[0]
        o.write("%s,%s,%s,%s,%s\n" % (name, r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]))
print(open(sys.argv[2]).read())
PY
tail -1 "$O/kt.log" | cut -c1-300
rm -rf "$O/kt"
