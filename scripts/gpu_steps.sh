#!/bin/bash
# Run GPU steps in order on the box; an ordinary failure (assert, non-zero exit) is logged and the next step runs,
# but a step that TIMED OUT or was KILLED (124 / 137 / 143) ends the call -- no GPU step is started after one.
# usage: bash scripts/gpu_steps.sh "<timeout_s>|<name>|<command>" ...
mkdir -p gpurun_out
for spec in "$@"; do
    t=${spec%%|*}; rest=${spec#*|}; name=${rest%%|*}; cmd=${rest#*|}
    echo "=== $name (limit ${t}s): $cmd" | tee -a gpurun_out/steps.log
    s0=$(date +%s)
    timeout -k 10 $t bash -c "$cmd" > gpurun_out/$name.out 2> gpurun_out/$name.err
    rc=$?
    echo "=== $name rc=$rc $(( $(date +%s) - s0 ))s" | tee -a gpurun_out/steps.log
    tail -3 gpurun_out/$name.out
    if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -eq 143 ]; then echo "step $name timed out / was killed: stopping" | tee -a gpurun_out/steps.log; exit $rc; fi
done
exit 0
