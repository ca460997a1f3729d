#!/bin/bash
OUT=gpurun_out/${TAG:-r06tstart2}; mkdir -p $OUT
line() { python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$1', round(d['ms_per_step'],2))"; }
for i in 1 2 3; do for A in p2 p3; do
  timeout 600 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-teacher-alone --teacher-after $A 2>/dev/null | line "100 steps teacher_after=$A" | tee -a $OUT/ab.txt
done; done
for i in 1 2; do for A in p2 p3; do
  timeout 600 python bench.py --steps 20 --warmup 5 --dtype fp16 --no-cpu-baseline --no-teacher-alone --teacher-after $A 2>/dev/null | line "fp16 20 steps teacher_after=$A" | tee -a $OUT/ab.txt
done; done
