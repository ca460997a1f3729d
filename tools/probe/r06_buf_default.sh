#!/bin/bash
# evidence for making the buffer-descriptor pieces the default (on top of the lgkmcnt(0) ring fix): the GPU tier with the arm on, the
# linearity test that found the mismatches in 12 fresh processes, 100-step and 20-step A/B of the step
set -u
OUT=gpurun_out/${TAG:-r06bufdef}; mkdir -p $OUT
export ET_CONV_BUF_DMA=1
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 | tee $OUT/tests_buf_on.txt
for i in $(seq 12); do timeout 300 python -m pytest "tests/test_fullsize.py::test_conv_adjoint_and_linearity_full_size" -x -q -m gpu 2>&1 | tail -1; done | tee $OUT/flake_buf_on.txt
unset ET_CONV_BUF_DMA
line() { python -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d['kernel_ms_by_family']; print('$1', round(d['ms_per_step'],2), k['main_stream'])"; }
for S in 100 20; do for i in 1 2 3; do for L in flat buf; do
  if [ $L = buf ]; then export ET_CONV_BUF_DMA=1; else export ET_CONV_BUF_DMA=0; fi
  timeout 600 python bench.py --steps $S --warmup 5 --no-cpu-baseline --no-teacher-alone 2>/dev/null | line "$L$S" | tee -a $OUT/ab.txt
done; done; done
unset ET_CONV_BUF_DMA
