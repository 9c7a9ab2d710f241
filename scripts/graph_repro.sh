#!/bin/bash
# run every ingredient set of scripts/graph_repro.hip in its own process; one line per set
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for f in 256 257 0 1 129 3 7 9 11 15 31 63 127 191; do
  timeout -k 5 120 scripts/graph_repro $f 96 > gpurun_out/repro_$f.txt 2>&1
  rc=$?
  echo "flags=$f rc=$rc :: $(grep -v amdgpu.ids gpurun_out/repro_$f.txt | tail -1)"
done
