#!/bin/bash
cd "$(dirname "$0")/.."
r() { timeout -k 5 120 python scripts/one_eval.py "$@" 2>&1 | grep -v amdgpu.ids | tail -1; }
timeout -k 10 300 python scripts/tail_check.py 2>&1 | grep -v amdgpu.ids | grep jitchol
r 1024
r 2048
r 3072
r 4096
r 6144
r 8192
r 8192 outer_nb=512 dag_workers=192
r 16384
timeout -k 10 200 python scripts/tail_stamps.py 2048 2>&1 | grep -v amdgpu.ids | grep -v "column period\|whole chain"
