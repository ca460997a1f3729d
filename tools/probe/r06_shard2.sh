set -u
OUT=gpurun_out/r06shard2; mkdir -p $OUT
line() { python -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d['kernel_ms_by_family']; print('$1', round(d['ms_per_step'],2), k['main_stream'])"; }
for S in 20 100; do for i in 1 2 3; do for N in 3200 13000; do
  timeout 600 python tools/probe/bn_sharded_ab.py $N --steps $S --warmup 5 --no-cpu-baseline --no-teacher-alone 2>/dev/null | line "steps=$S adds<=$N" | tee -a $OUT/ab.txt
done; done; done
