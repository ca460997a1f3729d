mkdir -p gpurun_out/r05c
OUT=gpurun_out/r05c
timeout 1500 python -m pytest tests/test_fp16.py tests/test_conv.py tests/test_norm_spatial.py tests/test_input_path.py tests/test_optim_ema.py tests/test_fuzz_misc.py tests/test_abi.py -x -q -m gpu > $OUT/pytest_a.log 2>&1; tail -4 $OUT/pytest_a.log | cut -c1-300
timeout 1500 python -m pytest tests/test_step_fullsize.py -x -q -m gpu -s -k "fp16 or fp32" > $OUT/pytest_b.log 2>&1; grep -E "PARITY|passed|failed|Error" $OUT/pytest_b.log | cut -c1-900 | tail -8
for D in bf16 fp16 bf16 fp16; do timeout 600 python bench.py --steps 20 --warmup 8 --no-cpu-baseline --dtype $D 2>$OUT/bench_$D.err | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('$D', round(d['ms_per_step'],2), d['dtype'], 'frac', round(d['roofline']['frac'],4), d['roofline']['kernel'], 'scale', d['config'].get('loss_scale_after_the_timed_region'), d['kernel_ms_by_family']['main_stream'])" | tee -a $OUT/dtype_ab.txt; done
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; python -c "
import json; d=json.loads(open('$OUT/bench.json').readline()); print(d['ms_per_step'], json.dumps(d['parity_check'])[:1500]); print(json.dumps(d['cpu_baseline'])[:300])"
