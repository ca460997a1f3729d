#!/bin/bash
# the other workloads and the one-rank data-parallel path at the final build
OUT=gpurun_out/${TAG:-r06other}; mkdir -p $OUT
for W in v5s-sup v8-sup v8-ssod; do
  timeout 900 python bench.py --workload $W --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$W', round(d['value'],1), d['unit'], round(d['ms_per_step'],2), 'ms')" | tee -a $OUT/other.txt
done
timeout 900 python bench.py --force-dp --per-rank 16 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('force-dp 16+16', round(d['ms_per_step'],2), d['config']['step_graph'])" | tee -a $OUT/other.txt
timeout 900 python bench.py --force-dp --per-rank 16 --no-graph --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('force-dp 16+16 eager', round(d['ms_per_step'],2))" | tee -a $OUT/other.txt
