#!/bin/bash
# usage: loop_test.sh DIR RUNS -- runs the launch-chain schedule test RUNS times in fresh processes, counts crashes
cd $1
bad=0
for i in $(seq 1 $2); do
  timeout -k 10 120 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "large_size or options_agree or variants_agree or knobs" > /tmp/loop.txt 2>&1
  rc=$?
  if [ $rc -ne 0 ]; then bad=$((bad+1)); echo "  run $i rc=$rc: $(grep -m1 -i "Segmentation\|Aborted\|failed\|Error" /tmp/loop.txt | cut -c1-100)"; fi
done
echo "$1: $bad bad runs of $2"
