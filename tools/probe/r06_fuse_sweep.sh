#!/bin/bash
# BatchNorm-backward reduce pass inside the producing dgrad's epilogue: which dgrads carry it (autograd.FUSE_BN_BWD_K: 1 = the 1x1 layers, 5 = + the 3x3 layers on 128-row
# row-shift tiles (default), 7 = + the 3x3 layers on the 256 x 256 tiles), at the final build
OUT=gpurun_out/${TAG:-r06fuse}; mkdir -p $OUT
line() { python -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d['kernel_ms_by_family']; print('$1', round(d['ms_per_step'],2), k['main_stream'])"; }
for i in 1 2 3; do for K in 5 7 1; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-teacher-alone --set autograd.FUSE_BN_BWD_K=$K 2>/dev/null | line "FUSE_BN_BWD_K=$K" | tee -a $OUT/ab.txt
done; done
