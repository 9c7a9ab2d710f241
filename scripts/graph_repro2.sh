#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export LD_LIBRARY_PATH=$PWD/gaussianprocess-mcmc_amd/gpmc:$LD_LIBRARY_PATH
echo "--- standalone repro, richer ingredients"
for f in 575 1599 3647 3711; do
  timeout -k 5 200 scripts/graph_repro $f 64 > gpurun_out/repro_$f.txt 2>&1; rc=$?
  echo "flags=$f rc=$rc :: $(grep -v amdgpu.ids gpurun_out/repro_$f.txt | tail -1)"
done
echo "--- libgpmc churn from plain C++ (no torch)"
run() { tag=$1; shift; timeout -k 5 200 scripts/churn_native "$@" > gpurun_out/churn_$tag.txt 2>&1; rc=$?; echo "churn $tag [$*] rc=$rc :: $(grep -v amdgpu.ids gpurun_out/churn_$tag.txt | tail -1)"; }
run m3_8192 3 32 8192
run m3_4096 3 32 4096
run m3_8192_nola 3 32 8192 lookahead=0
run m3_2048_la 3 32 2048 lookahead=1
run m3_8192_nopair 3 32 8192 panel_pair=0
run m3_8192_noprio 3 32 8192 panel_prio=0
run m0_8192 0 32 8192
run m8_8192 8 32 8192
