#!/bin/bash
# Collect PMC counters for the bench workload in separate rocprofv3 passes (MI355X_MICROARCH.md,
# "rocprofv3 PMC slots": FETCH_SIZE and WRITE_SIZE cannot share a pass).  Run on the GPU box:
#   bash tools/pmc_profile.sh            -> gpurun_out/pmc/<pass>/...csv
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc
rm -rf $OUT; mkdir -p $OUT
ARGS="$R/bench.py --steps 4 --warmup 2 --min-time 0 --no-cpu-baseline --no-secondary --handles 1 --frames-per-gpu 64"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $ARGS > $OUT/fetch.log 2>&1
echo "fetch pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $ARGS > $OUT/write.log 2>&1
echo "write pass done"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/sq1 -- python3 $ARGS > $OUT/sq1.log 2>&1
echo "sq1 pass done"
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq2 -- python3 $ARGS > $OUT/sq2.log 2>&1
echo "sq2 pass done"
find $OUT -name "*counter_collection.csv" | head
