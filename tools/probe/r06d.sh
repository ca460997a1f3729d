set -u
OUT=gpurun_out/r06d; mkdir -p $OUT; export TMPDIR=/tmp
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-teacher-alone --no-overlap > $GRAFT_REPO_ROOT/$OUT/bench.json 2> $GRAFT_REPO_ROOT/$OUT/prof.err)
T=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
python tools/trace_by_grid.py $T bn_act,rows_reduce > $OUT/bn_by_grid.txt
python tools/trace_by_grid.py $T > $OUT/all_by_grid.txt
head -3 $T > $OUT/trace_head.txt
find $OUT/prof -name "*.db" -delete; find $OUT/prof -name "*kernel_trace.csv" -delete
head -70 $OUT/bn_by_grid.txt | cut -c1-200
