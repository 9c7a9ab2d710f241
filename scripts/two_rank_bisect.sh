#!/bin/bash
# two ranks on one card, several option sets, N runs each: counts the runs that end in the tile-DAG spin limit
for opts in "" "tail_prefetch_t=0" "tail_workers=480" "outer_nb=384" "tail_acc16=0"; do
  fails=0
  for i in 1 2 3 4; do
    GPMC_BENCH_OPTS="$opts" GPMC_DIST_BACKEND=gloo timeout -k 10 120 python bench.py --gpus 2 --steps 8 --warmup 2 --metric-only > gpurun_out/bis.json 2> gpurun_out/bis.err
    rc=$?; if [ $rc -ne 0 ]; then fails=$((fails+1)); fi
  done
  echo "opts='$opts' failures=$fails of 4"
done
echo "single process, 300 steps:"
timeout -k 10 200 python bench.py --steps 300 --metric-only > gpurun_out/single300.json 2> gpurun_out/single300.err; echo "rc=$? timeouts=$(grep -c 1001 gpurun_out/single300.err)"
