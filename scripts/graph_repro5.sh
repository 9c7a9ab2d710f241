#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
TL=$(python3 -c "import torch,os;print(os.path.join(os.path.dirname(torch.__file__),'lib'))")
G=$PWD/gaussianprocess-mcmc_amd/gpmc
export LD_LIBRARY_PATH=$G
timeout -k 5 600 /opt/rocm/bin/rocgdb -batch -ex "set pagination off" -ex "set exec-wrapper env LD_PRELOAD=$TL/libamdhip64.so" \
   -ex "catch throw" -ex run -ex "bt 40" -ex "info sharedlibrary amdhip" -ex continue -ex "bt 40" --args scripts/churn_native 3 32 8192 > gpurun_out/r5_gdb.txt 2>&1
echo "gdb rc=$?"
grep -v "^\[New Thread\|^\[Thread\|amdgpu.ids" gpurun_out/r5_gdb.txt | tail -80
