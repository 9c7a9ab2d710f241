#!/bin/bash
cd "$(dirname "$0")/.."
r() { timeout -k 5 200 python scripts/one_eval.py "$@" 2>&1 | grep -v amdgpu.ids | tail -1; }
for n in 8192 9216 10240 11264 12288 14336 16384 20480 24576; do r $n; done
r 12288 dag_workers=160
r 16384 dag_workers=128
r 20480 dag_workers=128
r 20480 dag_workers=96
