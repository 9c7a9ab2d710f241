#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
TL=$(python3 -c "import torch,os;print(os.path.join(os.path.dirname(torch.__file__),'lib'))")
G=$PWD/gaussianprocess-mcmc_amd/gpmc
echo "torch lib dir: $TL"
python3 - <<'PY'
import ctypes
for p in ('/opt/rocm/lib/libamdhip64.so.7',):
    L=ctypes.CDLL(p); v=ctypes.c_int(); L.hipRuntimeGetVersion(ctypes.byref(v)); print(p, v.value)
PY
python3 - <<'PY'
import torch, ctypes, os
L=ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__),'lib','libamdhip64.so')); v=ctypes.c_int(); L.hipRuntimeGetVersion(ctypes.byref(v)); print('torch bundled', v.value, torch.version.hip)
PY
run() { tag=$1; shift; timeout -k 5 300 "$@" > gpurun_out/r3_$tag.txt 2>&1; rc=$?; echo "$tag rc=$rc :: $(grep -v amdgpu.ids gpurun_out/r3_$tag.txt | tail -1)"; }
echo "--- python (torch's bundled runtime), round-1 script"
run py_m3 python3 scripts/graph_stress.py 3 32 8192
run py_m0 python3 scripts/graph_stress.py 0 32 8192
echo "--- plain C++ on /opt/rocm runtime"
LD_LIBRARY_PATH=$G run cpp_rocm72_m3 scripts/churn_native 3 32 8192
echo "--- plain C++ forced onto torch's bundled runtime"
LD_LIBRARY_PATH=$TL:$G run cpp_torchrt_m3 scripts/churn_native 3 32 8192
LD_LIBRARY_PATH=$TL:$G run cpp_torchrt_m0 scripts/churn_native 0 32 8192
LD_LIBRARY_PATH=$TL:$G run cpp_torchrt_m8 scripts/churn_native 8 32 8192
LD_LIBRARY_PATH=$TL run repro_torchrt_63 scripts/graph_repro 63 64
LD_LIBRARY_PATH=$TL run repro_torchrt_3711 scripts/graph_repro 3711 64
LD_LIBRARY_PATH=$TL run repro_torchrt_1 scripts/graph_repro 1 64
