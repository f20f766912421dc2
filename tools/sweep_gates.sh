#!/bin/bash
# headline under several stage-gate configurations, one GPU box: bash tools/sweep_gates.sh "0|1|2|3|0:2|" [handles]
IFS='|' read -r -a GS <<< "$1"; H=${2:-3}
R=$GRAFT_REPO_ROOT
for i in 1 2; do for g in "${GS[@]}" ""; do
python3 $R/bench.py --no-cpu-baseline --no-secondary --handles $H --gates "$g" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('handles $H gates [$g]', d['value'], d['ms_per_step'])"
done; done
