#!/bin/bash
# kernel-trace + stats of bench.py (headline only) -> gpurun_out/prof_<tag>/ ; usage: bash tools/prof_bench.sh <tag> [bench args]
set -e
TAG=${1:-x}; shift || true
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary "$@" > $OUT/bench.json 2> $OUT/bench.err
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/kernel_stats.csv")))
for r in rows[:22]:
    print("%-60s calls %6s avg_us %9.1f total_ms %9.2f pct %s" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6, r["Percentage"]))
PY
