#!/bin/bash
# where should the teacher stream start at the final build?  (p2 was tuned in r03 / r04 at 55-58 ms per step)
OUT=gpurun_out/${TAG:-r06tstart}; mkdir -p $OUT
line() { python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$1', round(d['ms_per_step'],2))"; }
for i in 1 2 3; do for A in p2 start p1 p3 p4; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-teacher-alone --teacher-after $A 2>/dev/null | line "teacher_after=$A" | tee -a $OUT/ab.txt
done; done
