#!/bin/bash
# the r05 tree (git archive f481a0a, built here with its own build()) against the final r06 tree on ONE box, alternating processes
OUT=gpurun_out/${TAG:-r06vsr05}; mkdir -p $OUT
line() { python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$1', round(d['ms_per_step'],2), round(d['value'],1), 'images/s')"; }
for S in 20 100; do for i in 1 2 3; do
  (cd _r05_tree && timeout 600 python bench.py --steps $S --warmup 5 --no-cpu-baseline 2>/dev/null) | line "r05 tree, $S steps:" | tee -a $OUT/ab.txt
  timeout 600 python bench.py --steps $S --warmup 5 --no-cpu-baseline --no-teacher-alone 2>/dev/null | line "r06 tree, $S steps:" | tee -a $OUT/ab.txt
done; done
