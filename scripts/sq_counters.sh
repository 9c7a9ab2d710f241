#!/bin/bash
# SQ / GRBM / TCC counter passes over the bench command for the factorisation's kernels (round 3: the tile-DAG kernel is
# the dominant one): MFMA busy, issue stalls, LDS activity and bank conflicts, L2 hit rate.  Counters only (no trace
# domains besides the kernel trace the counter collection implies).
# usage (on the GPU box, from the repo root): bash scripts/sq_counters.sh rNN [bench args...]
R=${1:-r03}; shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/sq_$R
mkdir -p $OUT
ARGS="--metric-only --steps 4 --warmup 2 $*"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE \
    --output-format csv -d $OUT/p1 -- python3 $ROOT/bench.py $ARGS > $OUT/p1.line 2> $OUT/p1.err; echo "p1 rc=$?"
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE \
    --output-format csv -d $OUT/p2 -- python3 $ROOT/bench.py $ARGS > $OUT/p2.line 2> $OUT/p2.err; echo "p2 rc=$?"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum \
    --output-format csv -d $OUT/p3 -- python3 $ROOT/bench.py $ARGS > $OUT/p3.line 2> $OUT/p3.err; echo "p3 rc=$?"
cd $ROOT
python3 scripts/counter_summary.py $OUT/summary.json $OUT/p1 $OUT/p2 $OUT/p3 > $OUT/summary.txt 2>&1; echo "summary rc=$?"
find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete
cat $OUT/summary.txt | head -60
