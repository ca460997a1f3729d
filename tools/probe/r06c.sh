set -u
OUT=gpurun_out/r06c; mkdir -p $OUT
line() { python -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d['kernel_ms_by_family']; r=d['roofline']; print('$1', round(d['ms_per_step'],2), 'frac', round(r['frac'],4), 'conv_tf', round(r['all_conv_kernels']['tflops'],1), k['main_stream'])"; }
timeout 900 python -m pytest tests/test_bn_sharded.py tests/test_norm_spatial.py tests/test_ssod_step.py -x -q -m gpu > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
for i in 1 2 3; do
for N in 1700 3200 13000; do
  timeout 600 python tools/probe/bn_sharded_ab.py $N --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | line "adds<=$N" | tee -a $OUT/ab_shard.txt
done
done
for i in 1 2 3; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --float-inputs 2>/dev/null | line "float-inputs" | tee -a $OUT/ab_inputs.txt
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | line "uint8-inputs" | tee -a $OUT/ab_inputs.txt
done
