#!/bin/bash
# is the slow kind of box slow on the HOST side?  the captured step graph (2.5 ms of host per replay) against eager issue (15-23 ms)
OUT=gpurun_out/${TAG:-r06graph}; mkdir -p $OUT
line() { python -c "import sys,json; d=json.loads(sys.stdin.readline()); c=d['config']; print('$1', round(d['ms_per_step'],2), 'host empty-queue', round(c['host_enqueue_ms_empty_queue'],1), 'graph', c['step_graph']['enabled'], c['step_graph']['error'])"; }
for i in 1 2 3; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-teacher-alone 2>/dev/null | line "eager" | tee -a $OUT/ab.txt
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-teacher-alone --graph 2>/dev/null | line "graph" | tee -a $OUT/ab.txt
done
nproc; grep -m1 "model name" /proc/cpuinfo; rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4
