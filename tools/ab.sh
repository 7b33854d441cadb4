#!/bin/bash
# Kernel A/B on one box: the bench's timed step alone (tools/score_only.py) for several builds of the product library,
# alternating so that clock drift hits them alike.   bash tools/ab.sh "cfg3 cfg3 cfg2" variants/base.so ransac.jl_amd/libransac_hip.so
WLS=$1; shift
for wl in $WLS; do
  for lib in "$@"; do
    r=$(WL=$wl RH_LIB_PATH=$PWD/$lib timeout -k 10 400 python tools/score_only.py 2>/dev/null | grep ms_per_step)
    echo "$wl $(basename $lib) $r"
  done
done
