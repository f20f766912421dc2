#!/bin/bash
# Single-frame latency of the host entry and stand-alone stage times for several builds on one GPU box:
#   bash tools/ab_single.sh "<EXTRA A>|<EXTRA B>|..."
IFS='|' read -r -a VARS <<< "$1"; shift
R=$GRAFT_REPO_ROOT
for i in 1 2; do
  for v in "${VARS[@]}"; do
    make -s -B -C $R/orb_slam2_comment_amd/csrc EXTRA="$v" > /dev/null 2>&1 || echo "build failed: $v"
    echo "[$v] single frame: $(python3 $R/tools/bench_host_path.py --frames 1 --reps 300 2>/dev/null | tail -1)"
    python3 $R/bench.py --no-cpu-baseline --no-secondary --handles 1 --frames-per-gpu 64 --min-time 0.5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('[$v] 64 frames:', d['value'], d['roofline']['alone']['stage_us'])"
  done
done
make -s -B -C $R/orb_slam2_comment_amd/csrc > /dev/null 2>&1
