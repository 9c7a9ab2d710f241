#!/bin/bash
cd "$(dirname "$0")/.."
r() { timeout -k 5 200 python scripts/one_eval.py "$@" 2>&1 | grep -v amdgpu.ids | tail -1; }
r 8192
r 8192 tail_m=5120
r 8192 tail_m=7168
r 8192 tail_whole_m=8192
r 8192 dag_workers=96
r 8192 dag_workers=192
r 9216
r 9216 tail_m=7168
r 10240
r 10240 tail_m=7168
r 12288
r 12288 tail_m=7168
r 16384
r 16384 tail_m=7168
r 16384 f32
r 8192 f32
r 8192 f32 tail_m_f32=5120
