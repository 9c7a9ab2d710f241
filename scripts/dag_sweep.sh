#!/bin/bash
cd "$(dirname "$0")/.."
r() { timeout -k 5 120 python scripts/one_eval.py "$@" 2>&1 | grep -v amdgpu.ids | tail -1; }
r 8192
r 8192 dag_wprio=1
r 8192 dag_wprio=2
r 8192 dag_wprio=2 dag_workers=96
r 8192 dag_wprio=2 dag_workers=64
r 8192 dag_wprio=2 outer_nb=512 dag_workers=192
r 8192 dag_wprio=2 outer_nb=512 dag_workers=128
r 8192 dag_wprio=2 outer_nb=512 dag_workers=128 dag_hist=0
r 16384 dag_wprio=2
