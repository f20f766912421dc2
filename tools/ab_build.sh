#!/bin/bash
# A/B/C... of several builds of liborbhip.so on ONE GPU box (devices differ by several per cent, so variants are only
# comparable inside one call): bash tools/ab_build.sh "<EXTRA flags A>|<EXTRA flags B>|..." [bench args]
# runs bench.py (headline only) for every variant, twice, interleaved; prints value / ms_per_step / stand-alone stage_us.
# The product build (no EXTRA) is restored at the end.
IFS='|' read -r -a VARS <<< "$1"; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
run() {
  python3 $R/bench.py --no-cpu-baseline --no-secondary "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$TAG', d['value'], d['ms_per_step'], d['roofline']['alone']['stage_us'])"
}
for i in 1 2; do
  for v in "${VARS[@]}"; do
    make -s -B -C $R/orb_slam2_comment_amd/csrc EXTRA="$v" > /dev/null 2>&1 || echo "build failed: $v"
    TAG="[$v]" run "$@"
  done
done
make -s -B -C $R/orb_slam2_comment_amd/csrc > /dev/null 2>&1
