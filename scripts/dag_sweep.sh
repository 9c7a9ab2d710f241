#!/bin/bash
cd "$(dirname "$0")/.."
r() { timeout -k 5 120 python scripts/one_eval.py "$@" 2>&1 | grep -v amdgpu.ids | tail -1; }
for tm in 4096 5120 6144; do
for tw in 448 480; do
r 8192 tail_m=$tm tail_workers=$tw
done
done
r 8192 tail_m=5120 tail_workers=480 dag_workers=192
r 8192 tail_m=5632 tail_workers=480
r 7168 tail_m=5120 tail_workers=480
r 7168 tail_m=8192 tail_workers=480 lookahead=0
for tm in 4096 6144; do r 12288 tail_m=$tm tail_workers=480; r 16384 tail_m=$tm tail_workers=480; done
