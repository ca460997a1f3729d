#!/bin/bash
# the LDS-ring fix (lgkmcnt(0) in front of the slot-reuse barriers): old library vs new, then the buffer-descriptor pieces on top of the new one
set -u
OUT=gpurun_out/${TAG:-r06lgkm}; mkdir -p $OUT
line() { python -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d['kernel_ms_by_family']; print('$1', round(d['ms_per_step'],2), k['main_stream'])"; }
for i in 1 2 3; do
  for L in base new newbuf; do
    unset ET_HIP_LIB; unset ET_CONV_BUF_DMA
    [ $L = base ] && export ET_HIP_LIB=$PWD/tools/probe/libet_base.so
    [ $L = newbuf ] && export ET_CONV_BUF_DMA=1
    timeout 600 python bench.py --steps ${STEPS:-20} --warmup 5 --no-cpu-baseline --no-teacher-alone 2>/dev/null | line $L | tee -a $OUT/ab.txt
  done
done
unset ET_HIP_LIB; unset ET_CONV_BUF_DMA
