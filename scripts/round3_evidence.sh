#!/bin/bash
# Round-3 evidence (GPU box, repo root): profiles of the bench command, timeline, chain stamps, size sweep, bench lines.
R=r03; O=gpurun_out/evidence; mkdir -p $O
bash scripts/collect_profiles.sh $R > $O/collect.log 2>&1
bash scripts/timeline.sh 8192 > $O/timeline_n8192.txt 2>&1
python3 scripts/tail_stamps.py 2048 > $O/stamps.txt 2>&1; python3 scripts/tail_stamps.py 4096 >> $O/stamps.txt 2>&1
python3 scripts/optsweep.py 1024,2048,3072,4096,5120,6144,7168,8192,12288,16384 > $O/sizes.txt 2>&1
python3 scripts/config5.py > $O/config5.txt 2>&1
python3 scripts/sds_time.py 8192 > $O/sds_time.txt 2>&1
python3 scripts/small_chain.py 200 455 1024 2048 > $O/small_chain.txt 2>&1
GPMC_DIST_BACKEND=gloo timeout -k 10 300 python3 bench.py --gpus 2 --steps 10 --metric-only > $O/two_ranks.json 2> $O/two_ranks.err
timeout -k 10 300 python3 bench.py --metric-only --full > $O/bench_full.json 2> $O/bench_full.err
tail -c 600 $O/bench_full.json
