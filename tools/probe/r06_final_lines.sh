#!/bin/bash
# the committed bench lines of the final build: bf16 (driver form), fp16, default (40 steps) form
OUT=gpurun_out/${TAG:-r06lines}; mkdir -p $OUT
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; cut -c1-300 $OUT/bench.json
timeout 900 python bench.py --steps 20 --warmup 5 --dtype fp16 > $OUT/bench_fp16.json 2> $OUT/bench_fp16.err; cut -c1-300 $OUT/bench_fp16.json
timeout 900 python bench.py --no-cpu-baseline > $OUT/bench_default40.json 2> $OUT/bench_default40.err; cut -c1-300 $OUT/bench_default40.json
rocm-smi --showpower --showclocks 2>/dev/null | head -14 > $OUT/smi.txt
