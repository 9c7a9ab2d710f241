#!/bin/bash
cd "$(dirname "$0")/.."
TL=$(python3 -c "import torch,os;print(os.path.join(os.path.dirname(torch.__file__),'lib'))")
for sp in 4000 400; do
echo "=== /opt/rocm runtime"; scripts/graph_launch_cost $sp 2>&1 | grep -v amdgpu.ids
echo "=== torch's bundled runtime"; LD_PRELOAD=$TL/libamdhip64.so scripts/graph_launch_cost $sp 2>&1 | grep -v amdgpu.ids
done
