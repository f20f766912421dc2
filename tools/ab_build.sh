#!/bin/bash
# A/B of two builds of liborbhip.so on the GPU box: bash tools/ab_build.sh "<EXTRA flags of variant B>" [bench args]
# runs bench.py (headline only) as A, B, A, B; prints value / ms_per_step / stage_us of each run.
FL="$1"; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
run() {
  python3 $R/bench.py --no-cpu-baseline --no-secondary "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$TAG', d['value'], d['ms_per_step'], d['roofline']['stage_us'])"
}
for i in 1 2; do
  make -s -B -C $R/orb_slam2_comment_amd/csrc > /dev/null 2>&1; TAG=A run "$@"
  make -s -B -C $R/orb_slam2_comment_amd/csrc EXTRA="$FL" > /dev/null 2>&1; TAG=B run "$@"
done
make -s -B -C $R/orb_slam2_comment_amd/csrc > /dev/null 2>&1
