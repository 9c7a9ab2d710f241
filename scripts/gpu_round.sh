#!/bin/bash
# one GPU-box pass: tests, timing repeatability, bench (1 rank + 2-rank rehearsal over gloo on one card)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_gpu.txt
timeout -k 10 200 python scripts/stability.py > gpurun_out/stability.txt 2>&1; echo "stability rc=$?"; grep -v amdgpu gpurun_out/stability.txt | cut -c1-140
timeout -k 10 400 python bench.py > gpurun_out/bench_r02b.json 2> gpurun_out/bench_r02b.err; echo "bench rc=$?"; cut -c1-1200 gpurun_out/bench_r02b.json; tail -3 gpurun_out/bench_r02b.err
for n in 12288 16384; do timeout -k 5 120 python scripts/one_eval.py $n 2>&1 | tail -1; timeout -k 5 120 python scripts/one_eval.py $n outer_nb=512 2>&1 | tail -1; done
