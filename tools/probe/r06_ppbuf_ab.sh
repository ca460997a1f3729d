#!/bin/bash
# conv_gemm_pp_kernel on buffer descriptors (tools/probe/pp_buffer_dma.patch on the current conv.hip), re-measured now that the row-sharing kernels run that form
set -u
OUT=gpurun_out/${TAG:-r06ppbuf}; mkdir -p $OUT
export ET_HIP_LIB=$PWD/tools/probe/libet_ppbuf.so
ET_PP_BUF=1 timeout 900 python -m pytest tests/test_conv.py tests/test_fullsize.py -x -q -m gpu 2>&1 | tail -2 | tee $OUT/tests.txt
line() { python -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d['kernel_ms_by_family']; print('$1', round(d['ms_per_step'],2), k['main_stream'])"; }
for S in 20 100; do for i in 1 2 3; do for L in 0 1; do
  ET_PP_BUF=$L timeout 600 python bench.py --steps $S --warmup 5 --no-cpu-baseline --no-teacher-alone 2>/dev/null | line "pp_buf=$L steps=$S" | tee -a $OUT/ab.txt
done; done; done
