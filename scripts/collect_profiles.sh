#!/bin/bash
# rocprofv3 evidence for the round: kernel stats of the bench command, PMC traffic passes, a kernel timeline.
# usage (on the GPU box, from the repo root): bash scripts/collect_profiles.sh rNN
R=${1:-r02}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --metric-only > $OUT/line_under_rocprof.json 2> $OUT/stats.err
echo "stats rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py --metric-only --steps 4 --warmup 2 > /dev/null 2> $OUT/pmc_fetch.err
echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py --metric-only --steps 4 --warmup 2 > /dev/null 2> $OUT/pmc_write.err
echo "write rc=$?"
cd $ROOT
python3 scripts/pmc_summary.py $OUT/pmc_summary.json $OUT/pmc_fetch $OUT/pmc_write > /dev/null 2>&1; echo "summary rc=$?"
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/kernel_stats.csv 2>/dev/null; head -12 $OUT/kernel_stats.csv
# keep the merge small: drop the raw per-dispatch traces
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete
cat $OUT/line_under_rocprof.json | cut -c1-400
# SQ / GRBM / TCC counter passes (MFMA busy, stalls, LDS conflicts, L2 hit rate) over the same command
bash scripts/sq_counters.sh $R > $OUT/sq.log 2>&1; cp gpurun_out/sq_$R/summary.json $OUT/sq_summary.json; cp gpurun_out/sq_$R/summary.txt $OUT/sq_summary.txt
