#!/bin/bash
# hardware bf16 conversion (v_cvt_pk_bf16_f32) vs the bit-twiddling form: GPU tier, conv / BN microbench, the step, on one box
OUT=gpurun_out/cvt; mkdir -p $OUT
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; tail -2 $OUT/pytest_gpu.log
for L in old new; do
  if [ $L = old ]; then export ET_HIP_LIB=$PWD/tools/probe/libet_oldcvt.so; else unset ET_HIP_LIB; fi
  echo "== $L"
  MB_REF=0 timeout 600 python tools/microbench.py conv > $OUT/mb_conv_$L.log 2>&1; tail -1 $OUT/mb_conv_$L.log | cut -c1-200
  timeout 600 python tools/microbench.py bn > $OUT/mb_bn_$L.log 2>&1; tail -1 $OUT/mb_bn_$L.log | cut -c1-300
done
for L in old new old new; do
  if [ $L = old ]; then export ET_HIP_LIB=$PWD/tools/probe/libet_oldcvt.so; else unset ET_HIP_LIB; fi
  timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_$L.json 2> $OUT/bench_$L.err; echo "== $L"; cut -c1-200 $OUT/bench_$L.json
done
