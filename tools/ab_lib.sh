#!/bin/bash
# A/B of prebuilt libraries on ONE GPU box (boxes differ by several per cent): bash tools/ab_lib.sh "<a.so>|<b.so>|..." [mode]
# copies each library over the product path in turn, two interleaved rounds; mode "stage" (default) prints the stand-alone
# stage times of one 64-frame pipeline, mode "headline" the default bench headline.  The last library listed stays installed.
IFS='|' read -r -a LIBS <<< "$1"; MODE=${2:-stage}
R=$GRAFT_REPO_ROOT
for i in 1 2; do
  for l in "${LIBS[@]}"; do
    cp "$R/$l" $R/orb_slam2_comment_amd/liborbhip.so
    if [ "$MODE" = stage ]; then
      python3 $R/bench.py --no-cpu-baseline --no-secondary --handles 1 --frames-per-gpu 64 --min-time 0.5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('[$l]', d['value'], d['roofline']['alone']['stage_us'], 'match', d['roofline']['stage_us']['match'])"
    else
      python3 $R/bench.py --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('[$l]', d['value'], d['ms_per_step'], d['roofline']['alone']['stage_us'])"
    fi
  done
done
