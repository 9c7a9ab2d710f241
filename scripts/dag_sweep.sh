#!/bin/bash
cd "$(dirname "$0")/.."
r() { timeout -k 5 200 python scripts/one_eval.py "$@" 2>&1 | grep -v amdgpu.ids | tail -1; }
for n in 6144 8192 16384; do
r $n f32 tail_m_f32=0 tail_whole_m_f32=0
r $n f32
r $n f32 tail_m_f32=2048
r $n f32 tail_m_f32=3072
done
