#!/bin/bash
# same-box A/B of the stream-K form of the ping-pong gather-GEMM: parity tests, per-layer microbench, step
OUT=gpurun_out/sk_ab; mkdir -p $OUT
timeout 900 python -m pytest tests/test_conv.py -x -q -m gpu -k "stream_k or instantiation" > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log
for SK in ${SKS:-0 1}; do
  ET_CONV_SK=$SK MB_REF=0 MB_ONLY=pp timeout 600 python tools/microbench.py conv > $OUT/mb_sk$SK.log 2>&1; echo "SK=$SK"; tail -1 $OUT/mb_sk$SK.log | cut -c1-200
done
for SK in ${BENCH_SKS:-0 1 0 1}; do
  ET_CONV_SK=$SK timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_sk$SK.json 2> $OUT/bench_sk$SK.err; echo "SK=$SK"; cut -c1-200 $OUT/bench_sk$SK.json
done
