#!/bin/bash
cd "$(dirname "$0")/.."
r() { timeout -k 5 120 python scripts/one_eval.py "$@" 2>&1 | grep -v amdgpu.ids | tail -1; }
timeout -k 10 300 python scripts/tail_check.py 2>&1 | grep -v amdgpu.ids | grep jitchol | tail -3
r 2048
r 2048 tail_workers=256
r 4096
r 4096 tail_workers=256
for nb in 256 512; do
for wk in 32 64 96 128 192 480; do
r 8192 outer_nb=$nb dag_workers=$wk
done
done
r 6144 outer_nb=512 dag_workers=64
r 6144 outer_nb=256 dag_workers=64
r 16384 outer_nb=256 dag_workers=64
r 16384 outer_nb=512 dag_workers=64
r 16384 outer_nb=512 dag_workers=128
