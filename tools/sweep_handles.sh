R=$GRAFT_REPO_ROOT
for h in 2 3 4; do for g in 0 ""; do
python3 $R/bench.py --no-cpu-baseline --no-secondary --handles $h --gates "$g" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('handles $h gates [$g]', d['value'], d['ms_per_step'])"
done; done
