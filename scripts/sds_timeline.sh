#!/bin/bash
# kernel timeline of one SDS proposal at N (GPU box, repo root): bash scripts/sds_timeline.sh N
ROOT=$(pwd); O=$ROOT/gpurun_out/sdstl_$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/tl -- python3 $ROOT/scripts/sds_time.py $1 > $O/tl.log 2>&1
cd $ROOT
python3 scripts/sds_timeline.py $O/tl > $O/timeline.txt 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
cat $O/timeline.txt
