#!/bin/bash
# same-box A/B of the in-tree library against tools/probe/libet_prev.so: loss / step tests, then the step
# usage: cp efficientteacher_amd/libet_hip.so tools/probe/libet_prev.so BEFORE rebuilding with the change under test, then gpurun this script
OUT=gpurun_out/lib_ab_loss; mkdir -p $OUT
timeout 900 python -m pytest tests/test_loss.py tests/test_ota.py tests/test_ssod_step.py tests/test_step_fullsize.py tests/test_fuzz_misc.py -x -q -m gpu > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log
for L in prev new prev new; do
  if [ $L = prev ]; then export ET_HIP_LIB=$PWD/tools/probe/libet_prev.so; else unset ET_HIP_LIB; fi
  timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_$L.json 2> $OUT/bench_$L.err; echo "== $L"; cut -c1-200 $OUT/bench_$L.json
done
