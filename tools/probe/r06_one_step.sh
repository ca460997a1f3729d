set -u
OUT=gpurun_out/r06step; mkdir -p $OUT; export TMPDIR=/tmp
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-teacher-alone > $GRAFT_REPO_ROOT/$OUT/bench.json 2> $GRAFT_REPO_ROOT/$OUT/prof.err)
T=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
python tools/trace_one_step.py $T > $OUT/one_step.txt 2>&1
find $OUT/prof -name "*.db" -delete; find $OUT/prof -name "*kernel_trace.csv" -delete
head -3 $OUT/one_step.txt
