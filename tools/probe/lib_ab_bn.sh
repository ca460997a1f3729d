#!/bin/bash
# same-box A/B of the in-tree library against tools/probe/libet_prev.so: BN tests, BN microbench, the step
# usage: cp efficientteacher_amd/libet_hip.so tools/probe/libet_prev.so BEFORE rebuilding with the change under test, then gpurun this script
OUT=gpurun_out/lib_ab_bn; mkdir -p $OUT
timeout 900 python -m pytest tests/test_norm_spatial.py tests/test_fuzz_misc.py tests/test_model.py -x -q -m gpu > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log
for L in prev new; do
  if [ $L = prev ]; then export ET_HIP_LIB=$PWD/tools/probe/libet_prev.so; else unset ET_HIP_LIB; fi
  timeout 600 python tools/microbench.py bn > $OUT/mb_bn_$L.log 2>&1; echo "== $L"; tail -1 $OUT/mb_bn_$L.log | cut -c1-300
done
for L in prev new prev new; do
  if [ $L = prev ]; then export ET_HIP_LIB=$PWD/tools/probe/libet_prev.so; else unset ET_HIP_LIB; fi
  timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_$L.json 2> $OUT/bench_$L.err; echo "== $L"; cut -c1-200 $OUT/bench_$L.json
done
