#!/bin/bash
# usage: two_rank_repeat.sh N RUNS [bench args] -- two ranks on one card (gloo), counts runs ending in the spin limit
n=$1; runs=$2; shift 2
fails=0
for i in $(seq 1 $runs); do
  GPMC_DIST_BACKEND=gloo timeout -k 10 150 python bench.py --gpus 2 --n $n --steps 8 --warmup 2 --metric-only "$@" > gpurun_out/rep.json 2> gpurun_out/rep.err
  if [ $? -ne 0 ]; then fails=$((fails+1)); cp gpurun_out/rep.err gpurun_out/rep_fail_$n.err; fi
done
echo "N=$n two ranks on one card: $fails failures in $runs runs"
