#!/bin/bash
cd "$(dirname "$0")/.."
r() { timeout -k 5 200 python scripts/one_eval.py "$@" 2>&1 | grep -v amdgpu.ids | tail -1; }
for i in 1 2; do for n in 4096 8192 12288; do r $n; r $n tail_roles_by_index=1; done; done
