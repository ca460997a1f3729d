set -u
OUT=gpurun_out/r06r; mkdir -p $OUT
for i in 1 2; do for K in "ET_PPRS_BUF=0 ET_PP_BUF=0" "ET_PPRS_BUF=1 ET_PP_BUF=0" "ET_PPRS_BUF=1 ET_PP_BUF=1"; do
  env $K timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-teacher-alone 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$K  200 steps: ms_per_step', round(d['ms_per_step'],2))"
done; done | tee $OUT/long_ab.txt
rocm-smi --showmaxpower 2>/dev/null | grep -i -E "max|power" | head -5 | tee -a $OUT/long_ab.txt
