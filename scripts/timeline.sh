#!/bin/bash
# kernel timeline of one replayed evaluation: bash scripts/timeline.sh N [options...]  (GPU box, repo root)
ROOT=$(pwd); O=$ROOT/gpurun_out/tl_$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/tl -- python3 $ROOT/scripts/one_eval.py "$@" > $O/tl.log 2>&1
cd $ROOT
python3 scripts/timeline2.py $O/tl 60 > $O/timeline.txt 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
cat $O/timeline.txt
