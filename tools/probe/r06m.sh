set -u
OUT=gpurun_out/${TAG:-r06m}; mkdir -p $OUT
line() { python -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d['kernel_ms_by_family']; r=d['roofline']; print('$1', round(d['ms_per_step'],2), 'frac', round(r['frac'],4), 'conv_tf', round(r['all_conv_kernels']['tflops'],1), k['main_stream'], k['teacher_stream'].get('gather_gemm_teacher'))"; }
for i in 1 2 3; do for K in "ET_PPRS_BUF=0 ET_PP_BUF=0" "ET_PPRS_BUF=1 ET_PP_BUF=0" "ET_PPRS_BUF=0 ET_PP_BUF=1" "ET_PPRS_BUF=1 ET_PP_BUF=1"; do env $K timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline ${EXTRA:-} 2>/dev/null | line "$K" | tee -a $OUT/ab_step.txt; done; done
