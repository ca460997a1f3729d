set -u
OUT=gpurun_out/r06o; mkdir -p $OUT; export TMPDIR=/tmp
for K in 0 1 0 1; do
  D=$OUT/prof_$K; rm -rf $D
  (cd /tmp && ET_PPRS_BUF=1 ET_PP_BUF=$K timeout 900 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$D -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-teacher-alone --no-overlap > $GRAFT_REPO_ROOT/$OUT/bench_$K.json 2> $GRAFT_REPO_ROOT/$OUT/prof_$K.err)
  T=$(find $D -name "*kernel_trace.csv" | head -1)
  echo "=== ET_PP_BUF=$K  ms_per_step $(python -c "import json;print(round(json.load(open('$OUT/bench_$K.json'))['ms_per_step'],2))")"
  python tools/trace_kernel_gaps.py $T conv_gemm_pp_kernel,conv_gemm_pprs,conv_gemm_rs,conv_gemm_glds,conv1x1_stream,conv_wgrad,bn_act,rows_reduce
  find $D -name "*.db" -delete; find $D -name "*kernel_trace.csv" -delete
done | tee $OUT/gaps.txt
