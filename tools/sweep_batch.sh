#!/bin/bash
# headline value for (handles, frames per step) combinations; usage: bash tools/sweep_batch.sh "1:128 2:128 1:256 2:256"
R=$GRAFT_REPO_ROOT
for hf in $1; do
  h=${hf%%:*}; f=${hf##*:}
  python3 $R/bench.py --no-cpu-baseline --no-secondary --handles $h --frames-per-gpu $f 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('handles $h frames $f ->', d['value'], 'f/s', d['ms_per_step'], 'ms', d['roofline']['stage_us'])"
done
