#!/bin/bash
# same-box A/B of the streaming 1x1 kernel: parity tests, 1x1 layers isolated (rotating buffers), the step
OUT=gpurun_out/s1x1; mkdir -p $OUT
timeout 900 python -m pytest tests/test_conv.py -x -q -m gpu -k "streaming or instantiation or adversarial" > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log
for S in 0 1; do
  ET_CONV_S1X1=$S MB_ROTATE=4 MB_K=1 MB_REF=0 timeout 600 python tools/microbench.py conv > $OUT/mb_s$S.log 2>&1; echo "S1X1=$S"; python tools/probe/show1x1.py $OUT/mb_s$S.log
done
for S in ${BENCH_S:-0 1 0 1}; do
  ET_CONV_S1X1=$S timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_s$S.json 2> $OUT/bench_s$S.err; echo "S1X1=$S"; cut -c1-200 $OUT/bench_s$S.json
done
