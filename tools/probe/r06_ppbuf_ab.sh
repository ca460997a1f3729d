#!/bin/bash
# conv_gemm_pp_kernel on buffer descriptors (tools/probe/pp_buffer_dma.patch on the current conv.hip): ET_PP_BUF=1 both operands, =2 the WEIGHT pieces only
set -u
OUT=gpurun_out/${TAG:-r06ppbuf2}; mkdir -p $OUT
export ET_HIP_LIB=$PWD/tools/probe/libet_ppbuf.so
true
line() { python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$1', round(d['ms_per_step'],2))"; }
for S in 20 100; do for i in 1 2 3; do for L in 0 2 1; do
  ET_PP_BUF=$L timeout 600 python bench.py --steps $S --warmup 5 --no-cpu-baseline --no-teacher-alone 2>/dev/null | line "pp_buf=$L steps=$S" | tee -a $OUT/ab.txt
done; done; done
