#!/bin/bash
# one GPU-box pass: tests, timing repeatability, bench (1 rank + 2-rank rehearsal over gloo on one card)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_gpu.txt
timeout -k 10 200 python scripts/stability.py > gpurun_out/stability.txt 2>&1; echo "stability rc=$?"; cat gpurun_out/stability.txt | cut -c1-120
timeout -k 10 300 python bench.py --no-cpu-ref > gpurun_out/bench_r02a.json 2> gpurun_out/bench_r02a.err; echo "bench rc=$?"; cat gpurun_out/bench_r02a.json; tail -3 gpurun_out/bench_r02a.err
GPMC_DIST_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 10 --no-cpu > gpurun_out/bench_r02a_2rank.json 2> gpurun_out/bench_r02a_2rank.err; echo "bench2 rc=$?"; cat gpurun_out/bench_r02a_2rank.json; tail -3 gpurun_out/bench_r02a_2rank.err
