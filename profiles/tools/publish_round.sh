#!/bin/bash
# Here (not on the GPU box):  bash profiles/tools/publish_round.sh <tag> <round dir>    e.g. r5/f3 r05
# Copies what tools/final_round.sh left under gpurun_out/<tag>/ into profiles/<round dir>/ (and traffic.json to profiles/), every file
# with the source hash of the build it was taken from.
set -eo pipefail
O=gpurun_out/$1; R=profiles/$2; H=$(cat $O/source_sha256.txt)
cp $O/kernel_stats.csv $R/final_kernel_stats.csv; sed -i "1i # source_sha256 $H" $R/final_kernel_stats.csv
cp $O/s1_alone/kernel_stats.csv $R/s1_kernel_stats_alone.csv
cp $O/s2_alone/kernel_stats.csv $R/s2_kernel_stats_alone.csv
cp $O/s2_queue/kernel_stats.csv $R/s2_kernel_stats.csv
cp $O/content_sensitivity.txt $R/content_sensitivity.txt
sed -i "1i # source_sha256 $H   (python3 profiles/tools/content_ab.py: 64 4K gray8 frames per step)" $R/content_sensitivity.txt
cp $O/traffic.json profiles/traffic.json; cp $O/traffic.json $R/traffic.json
(echo "# source_sha256 $H   (profiles/tools/timeline.sh: queued steps under rocprofv3 --kernel-trace; two lanes x two slices)"; head -48 $O/tl/timeline.txt) > $R/timeline_queued.txt
python3 profiles/tools/opcodes.py > $R/opcodes.txt 2>&1
echo "published $H"
