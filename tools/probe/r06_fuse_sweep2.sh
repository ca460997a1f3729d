#!/bin/bash
OUT=gpurun_out/${TAG:-r06fuse2}; mkdir -p $OUT
line() { python -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d['kernel_ms_by_family']; print('$1', round(d['ms_per_step'],2))"; }
for i in 1 2 3; do for K in 5 1 0; do
  timeout 600 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-teacher-alone --set autograd.FUSE_BN_BWD_K=$K 2>/dev/null | line "100 steps FUSE_BN_BWD_K=$K" | tee -a $OUT/ab.txt
done; done
for i in 1 2; do for K in 5 1 0; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-teacher-alone --set autograd.FUSE_BN_BWD_K=$K 2>/dev/null | line "20 steps FUSE_BN_BWD_K=$K" | tee -a $OUT/ab.txt
done; done
