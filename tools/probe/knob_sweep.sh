#!/bin/bash
# same-box A/B of the step bench under single tuning knobs (each run ~5 s of GPU time)
run() { echo -n "$1: "; env $1 python bench.py --steps 24 --warmup 6 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2))"; }
for K in "X=0" "X=1" "ET_CONV_NARROW_K=64" "ET_CONV_NARROW_K=256" "ET_CONV_TAP_INNER=0" "ET_EW_VPT=1" "ET_EW_VPT=2" "ET_EW_VPT=4" "ET_WGRAD_GROUP=4" "ET_WGRAD_GROUP=16" "ET_WGRAD_XCD=1" "ET_CONV_BIG_MINFILL=45" "ET_FUSE_BN_BWD=0" "X=2"; do run "$K"; done
