set -u
OUT=gpurun_out/r06x; mkdir -p $OUT
for i in 1 2 3; do for K in ET_RS_BUF=0 ET_RS_BUF=1; do
  env $K timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-teacher-alone 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$K  100 steps: ms_per_step', round(d['ms_per_step'],2))"
done; done | tee $OUT/rs_buf_long.txt
