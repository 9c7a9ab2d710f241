#!/bin/bash
cd "$(dirname "$0")/.."
r() { timeout -k 5 200 python scripts/one_eval.py "$@" 2>&1 | grep -v amdgpu.ids | tail -1; }
r 8192
r 8192 outer_nb=768
r 8192 outer_nb=768 dag_workers=256
r 8192 outer_nb=768 dag_workers=320
r 8192 outer_nb=768 dag_workers=192
r 8192 outer_nb=1152
r 8192 outer_nb=1152 dag_workers=320
r 8192
