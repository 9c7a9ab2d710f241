#!/bin/bash
cd "$(dirname "$0")/.."
r() { timeout -k 5 200 python "$@" 2>&1 | grep -v "amdgpu.ids\|^\[W\|Gloo" | tail -${TAILN:-1}; }
echo "== sizes, f64 (scripts/one_eval.py)"
for n in 1024 2048 3072 4096 5120 6144 8192 12288 16384; do r scripts/one_eval.py $n; done
echo "== round-1 schedule for comparison (tail_m=0 la_dag=0)"
for n in 2048 4096 8192; do r scripts/one_eval.py $n tail_m=0 la_dag=0; done
echo "== config 5 shape"
TAILN=4 r scripts/config5.py
echo "== small chains"
TAILN=4 r scripts/small_chain.py 200 455 1024 2048
echo "== full bench extras"
timeout -k 10 300 python bench.py --metric-only --full > gpurun_out/bench_full.json 2>/dev/null; python3 -c "
import json; d=json.load(open('gpurun_out/bench_full.json')); print(d['value'], d['extra'])"
echo "== 2 ranks on one card (gloo rehearsal of the self-spawned launch)"
GPMC_DIST_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 10 --no-cpu 2>gpurun_out/two_ranks.err | python3 -c "
import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(d['n_gpus'], d['value'], d['ms_per_step'])"
echo "== N=4096 bench line (config 4's per-GPU workload)"
timeout -k 10 300 python bench.py --n 4096 --metric-only 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
echo "== N=2048 bench line (config 2)"
timeout -k 10 300 python bench.py --n 2048 --metric-only 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
