#!/bin/bash
# weight-gradient group size (ops.WGRAD_QUEUE.group: layers of identical geometry per grouped launch; 8 since r03, when the launches ran on the main stream)
OUT=gpurun_out/${TAG:-r06wgroup}; mkdir -p $OUT
line() { python -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d['kernel_ms_by_family']; print('$1', round(d['ms_per_step'],2), k['wgrad_stream'])"; }
for i in 1 2 3; do for G in 8 4 16 2; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-teacher-alone --set ops.WGRAD_QUEUE.group=$G 2>/dev/null | line "group=$G" | tee -a $OUT/ab.txt
done; done
