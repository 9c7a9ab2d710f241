#!/bin/bash
cd "$(dirname "$0")/.."
r() { timeout -k 5 120 python scripts/one_eval.py "$@" 2>&1 | grep -v amdgpu.ids | tail -1; }
for tw in 192 256 320 384 416 448 480 512 640; do r 6144 tail_workers=$tw; done
for tw in 384 448 480 512; do r 8192 tail_workers=$tw; done
