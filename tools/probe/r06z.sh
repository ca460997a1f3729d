set -u
OUT=gpurun_out/r06z; mkdir -p $OUT
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; cut -c1-300 $OUT/bench.json
timeout 900 python bench.py --steps 20 --warmup 5 --dtype fp16 > $OUT/bench_fp16.json 2> $OUT/bench_fp16.err; cut -c1-300 $OUT/bench_fp16.json
for W in v5s-sup v8-sup v8-ssod; do timeout 900 python bench.py --workload $W --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$W', round(d['value'],1), d['unit'], round(d['ms_per_step'],2), 'ms')" | tee -a $OUT/other_workloads.txt; done
timeout 1500 python tools/dp_sweep.py --gpus 1 --chunks 48 --channels 0 --steps 12 --warmup 4 --wire fp32,bf16 --out $OUT/dp_sweep_1rank.json > $OUT/dp_sweep1.log 2>&1; tail -4 $OUT/dp_sweep1.log | cut -c1-300
timeout 600 python bench.py --steps 20 --warmup 5 --force-dp --per-rank 16 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('force-dp 16+16', round(d['ms_per_step'],2), d['config'].get('step_graph'), d['config'].get('grad_allreduce'))" | cut -c1-600 | tee $OUT/force_dp.txt
