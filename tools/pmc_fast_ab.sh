#!/bin/bash
# Development: SQ counters of the extractor kernels for one FAST variant through tools/fast_ab.py
# (separate rocprofv3 --pmc passes, kernel-trace only).  usage: bash tools/pmc_fast_ab.sh <variant> <tag>
set -e
V=${1:-0}; TAG=${2:-v$V}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmcab_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$R/tools/fast_ab.py --variants $V --rounds 1 --calls 3"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/sq1 -- python3 $ARGS > $OUT/sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq2 -- python3 $ARGS > $OUT/sq2.log 2>&1
cd $R
python3 tools/pmc_summarize.py $OUT > $OUT/summary.json
python3 - <<PY
import json
d=json.load(open("$OUT/summary.json"))
for k,v in d.items():
    if "fast" in k or "octree" in k or "describe" in k:
        w=v.get("SQ_WAVES",1)
        print(k, {c: round(v[c]/w,1) for c in ("SQ_INSTS_VALU","SQ_INSTS_SALU","SQ_INSTS_LDS","SQ_WAVE_CYCLES","SQ_WAIT_ANY","SQ_ACTIVE_INST_ANY","SQ_WAIT_INST_ANY") if c in v}, "waves", w,
              {c: round(v[c]) for c in ("SQ_LDS_BANK_CONFLICT","SQ_LDS_IDX_ACTIVE","SQ_ACTIVE_INST_VALU","SQ_ACTIVE_INST_LDS","SQ_ACTIVE_INST_SCA","SQ_BUSY_CYCLES","GRBM_GUI_ACTIVE") if c in v})
PY
