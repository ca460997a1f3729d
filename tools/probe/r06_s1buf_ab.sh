#!/bin/bash
# conv1x1_stream_kernel on buffer descriptors: GPU tests, then the step against the library of the previous commit (tools/probe/libet_base.so)
set -u
OUT=gpurun_out/${TAG:-r06s1buf}; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_conv.py tests/test_conv_fuzz.py tests/test_fullsize.py tests/test_bn_sharded.py tests/test_model.py tests/test_step_fullsize.py -x -q -m gpu 2>&1 | tail -2 | tee $OUT/tests.txt
line() { python -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d['kernel_ms_by_family']; print('$1', round(d['ms_per_step'],2), k['main_stream'])"; }
for S in 20 100; do for i in 1 2 3; do for L in base new; do
  unset ET_HIP_LIB; [ $L = base ] && export ET_HIP_LIB=$PWD/tools/probe/libet_base.so
  timeout 600 python bench.py --steps $S --warmup 5 --no-cpu-baseline --no-teacher-alone 2>/dev/null | line "$L steps=$S" | tee -a $OUT/ab.txt
done; done; done
unset ET_HIP_LIB
