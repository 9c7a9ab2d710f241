#!/bin/bash
cd "$(dirname "$0")/.."
r() { timeout -k 5 200 python scripts/one_eval.py "$@" 2>&1 | grep -v amdgpu.ids | tail -1; }
for n in 1024 2048 4096 6144 8192 12288; do r $n; r $n tail_warm=0; done
