#!/bin/bash
# Regenerates the measurement artefacts kept under profiles/ (run on the GPU box through gpurun):
#   kernel-trace stats (default 2 pipelines), bench line under the profiler, PMC passes (1 pipeline),
#   plain bench line, matcher latencies.  Outputs land in gpurun_out/refresh/.
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/refresh
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-match > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_handles2.csv
echo "stats done"
bash $R/tools/pmc_profile.sh > $OUT/pmc.log 2>&1
cd $R
python3 tools/pmc_summarize.py gpurun_out/pmc > $OUT/pmc_summary.json
echo "pmc done"
python3 bench.py > $OUT/bench_final.json 2> $OUT/bench_final.err
echo "bench done"
python3 tools/bench_matchers.py > $OUT/matcher_latency.json 2> $OUT/matcher_latency.err
cat $OUT/bench_final.json
cat $OUT/matcher_latency.json
