#!/bin/bash
cd "$(dirname "$0")/.."
r() { timeout -k 5 120 python scripts/one_eval.py "$@" 2>&1 | grep -v amdgpu.ids | tail -1; }
for nb in 256 384 512; do
for wk in 160 192 224 256 320; do
r 8192 outer_nb=$nb dag_workers=$wk
done
done
for wk in 128 192 256; do r 6144 dag_workers=$wk; done
for wk in 128 192 256; do r 12288 dag_workers=$wk; done
for wk in 192 256; do r 16384 outer_nb=512 dag_workers=$wk; done
for wk in 192 256; do r 16384 outer_nb=256 dag_workers=$wk; done
