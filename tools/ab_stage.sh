#!/bin/bash
# Stand-alone stage times of ONE 64-frame pipeline for several builds on one GPU box:
#   bash tools/ab_stage.sh "<EXTRA A>|<EXTRA B>|..."   -> per variant the HIP-event stage times (us), two rounds
IFS='|' read -r -a VARS <<< "$1"; shift
R=$GRAFT_REPO_ROOT
for i in 1 2; do
  for v in "${VARS[@]}"; do
    make -s -B -C $R/orb_slam2_comment_amd/csrc EXTRA="$v" > /dev/null 2>&1 || echo "build failed: $v"
    python3 $R/bench.py --no-cpu-baseline --no-secondary --handles 1 --frames-per-gpu 64 --min-time 0.5 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('[$v]', d['value'], d['roofline']['alone']['stage_us'], 'match', d['roofline']['stage_us']['match'])"
  done
done
make -s -B -C $R/orb_slam2_comment_amd/csrc > /dev/null 2>&1
