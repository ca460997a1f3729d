set -u
OUT=gpurun_out/${TAG:-r06j}; mkdir -p $OUT
line() { python -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d['kernel_ms_by_family']; r=d['roofline']; print('$1', round(d['ms_per_step'],2), 'frac', round(r['frac'],4), 'conv_tf', round(r['all_conv_kernels']['tflops'],1), k['main_stream'], k['teacher_stream'].get('gather_gemm_teacher'))"; }
for K in ${KNOB}=1 ${KNOB}=0; do echo "--- tests $K"; env $K timeout 900 python -m pytest tests/test_conv.py tests/test_conv_fuzz.py -x -q -m gpu 2>&1 | tail -2; done | tee $OUT/tests.txt
for M in hot rot; do for K in ${KNOB}=0 ${KNOB}=1 ${KNOB}=0 ${KNOB}=1; do echo "--- $M $K"; env $K ISO_MODE=$M timeout 300 python tools/probe/iso_conv.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        if '${KMATCH:-pprs}' in d['kernel']: print('   ', d['shape'], d['dir'], d['us_mean'], 'us', d['tflops'], 'TF')
"; done; done | tee $OUT/iso_ab.txt
for i in 1 2 3; do for K in ${KNOB}=0 ${KNOB}=1; do env $K timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | line "$K" | tee -a $OUT/ab_step.txt; done; done
