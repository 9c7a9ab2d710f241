#!/bin/bash
# same-box A/B of library builds (development): bash scripts/ab_libs.sh v0 v1 ...   (libgpmc_<tag>.so beside libgpmc.so)
for rep in 1 2; do
for t in "$@"; do
  export GPMC_LIB=$(pwd)/gaussianprocess-mcmc_amd/gpmc/libgpmc_$t.so
  echo "== $t (rep $rep)"
  python3 scripts/tail_stamps.py 2048 2>/dev/null | grep -E "TRSM|factor|stage\+store|load L|of \"stage"
  python3 scripts/optsweep.py 2048,4096,8192 2>/dev/null
done
done
