#!/bin/bash
# per-GPU batches of the N = 2 / 4 / 8 runs of configs[3] on one GPU: pipelines x gates.  bash tools/sweep_b64.sh "128 256"
R=$GRAFT_REPO_ROOT
for B in $1; do for h in 2 3; do for g in 0 ""; do
python3 $R/bench.py --no-cpu-baseline --no-secondary --frames-per-gpu $B --handles $h --gates "$g" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('B=$B handles $h gates [$g]', d['value'], d['ms_per_step'])"
done; done; done
