#!/bin/bash
# does the sign of the pp buffer-form A/B depend on the box or on the schedule constants?  the same library with the earlier constants (teacher start p2, FUSE_BN_BWD_K = 5)
set -u
OUT=gpurun_out/${TAG:-r06ppbuf5}; mkdir -p $OUT
export ET_HIP_LIB=$PWD/tools/probe/libet_ppbuf.so
line() { python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$1', round(d['ms_per_step'],2))"; }
for i in 1 2 3; do for L in 0 1; do
  ET_PP_BUF=$L timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-teacher-alone --teacher-after p2 --set autograd.FUSE_BN_BWD_K=5 2>/dev/null | line "p2,K=5 pp_buf=$L" | tee -a $OUT/ab.txt
done; done
for i in 1 2; do for L in 0 1; do
  ET_PP_BUF=$L timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-teacher-alone 2>/dev/null | line "final constants pp_buf=$L" | tee -a $OUT/ab.txt
done; done
