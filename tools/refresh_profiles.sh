#!/bin/bash
# Regenerates the measurement artefacts kept under profiles/ (run on the GPU box through gpurun):
#   kernel-trace stats for the default three pipelines and for one pipeline alone, PMC passes (1 pipeline),
#   the plain bench line, host-path rates.  Outputs land in gpurun_out/refresh/.
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/refresh
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats2 -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > $OUT/bench_under_rocprof_handles3.json 2> $OUT/stats2.err
cp $(find $OUT/stats2 -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_handles3.csv
echo "stats (3 pipelines) done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats1 -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --handles 1 --frames-per-gpu 64 > $OUT/bench_under_rocprof_handles1.json 2> $OUT/stats1.err
cp $(find $OUT/stats1 -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_handles1.csv
echo "stats (1 pipeline) done"
bash $R/tools/pmc_profile.sh > $OUT/pmc.log 2>&1
cd $R
python3 tools/pmc_summarize.py gpurun_out/pmc > $OUT/pmc_summary.json
echo "pmc done"
python3 bench.py > $OUT/bench_final.json 2> $OUT/bench_final.err
echo "bench done"
python3 tools/bench_host_path.py > $OUT/host_path.json 2> $OUT/host_path.err
python3 tools/bench_host_path.py --pinned > $OUT/host_path_pinned.json 2>> $OUT/host_path.err
python3 tools/bench_matchers.py > $OUT/matcher_latency.json 2> $OUT/matcher_latency.err
python3 tools/bench_track_th.py > $OUT/track_th.json 2> $OUT/track_th.err
timeout -k 5 100 ./tools/micro/valu_rate2 > $OUT/issue_rates.txt 2>&1 || true
timeout -k 5 100 ./tools/micro/valu_rate3 >> $OUT/issue_rates.txt 2>&1 || true
timeout -k 5 100 ./tools/micro/valu_rate4 >> $OUT/issue_rates.txt 2>&1 || true
rm -rf $OUT/stats2 $OUT/stats1
cat $OUT/bench_final.json
cat $OUT/host_path.json $OUT/host_path_pinned.json
