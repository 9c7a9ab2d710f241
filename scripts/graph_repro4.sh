#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
TL=$(python3 -c "import torch,os;print(os.path.join(os.path.dirname(torch.__file__),'lib'))")
G=$PWD/gaussianprocess-mcmc_amd/gpmc
run() { tag=$1; shift; timeout -k 5 300 "$@" > gpurun_out/r4_$tag.txt 2>&1; rc=$?; echo "$tag rc=$rc :: $(grep 'hip runtime' gpurun_out/r4_$tag.txt | head -1) :: $(grep -v amdgpu.ids gpurun_out/r4_$tag.txt | tail -1)"; }
echo "--- plain C++ preloaded with torch's bundled runtime (7.0)"
LD_LIBRARY_PATH=$G LD_PRELOAD=$TL/libamdhip64.so run cpp_rt70_m3 scripts/churn_native 3 32 8192
LD_PRELOAD=$TL/libamdhip64.so run repro_rt70_3711 scripts/graph_repro 3711 64
echo "--- python with /opt/rocm's 7.2 runtime preloaded"
LD_PRELOAD=/opt/rocm/lib/libamdhip64.so.7 run py_rt72_m3 python3 scripts/graph_stress.py 3 32 8192
echo "--- python as shipped (7.0 bundled)"
run py_rt70_m3 python3 scripts/graph_stress.py 3 32 8192
run py_rt70_m3_nola python3 scripts/graph_stress.py 3 32 8192 0
