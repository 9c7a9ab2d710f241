#!/bin/bash
# Regenerates the evidence files quoted in DESIGN.md sections 4/5 (run on the GPU box from the repo root).
ROOT=$(pwd); O=$ROOT/gpurun_out/final; mkdir -p $O
python3 scripts/tail_stamps.py 2048 > $O/stamps.txt 2>&1
python3 scripts/tail_stamps.py 4096 >> $O/stamps.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/tl -- python3 $ROOT/scripts/one_eval.py 8192 > $O/tl.log 2>&1
cd $ROOT
python3 scripts/timeline2.py $O/tl 60 > $O/timeline.txt 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 3000 $O/bench_default.json
