#!/bin/bash
cd "$(dirname "$0")/.."
r() { timeout -k 5 120 python scripts/one_eval.py "$@" 2>&1 | grep -v amdgpu.ids | tail -1; }
r 2048
r 3072 tail_m=4096
r 3072 tail_m=2048
r 4096 lookahead=0 tail_m=4096
r 4096 tail_m=2048
r 4096 tail_m=3072 la_min_m=3072
r 5120 lookahead=0 tail_m=4096
r 5120 tail_m=2048
r 5120 tail_m=4096 la_min_m=4096
r 6144 tail_m=2048
r 6144 tail_m=3072 la_min_m=3072
r 6144 tail_m=4096 la_min_m=4096
r 8192 tail_m=2048
r 8192 tail_m=3072 la_min_m=3072
r 8192 tail_m=4096 la_min_m=4096
r 8192 tail_m=2048
r 8192 tail_m=3072 la_min_m=3072
r 8192 tail_m=4096 la_min_m=4096
