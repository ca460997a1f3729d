set -u
OUT=gpurun_out/r06p; mkdir -p $OUT; export TMPDIR=/tmp
for K in 0 1 0 1; do
  D=$OUT/pmc_$K; rm -rf $D
  (cd /tmp && ET_PPRS_BUF=1 ET_PP_BUF=$K timeout 900 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$D -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-teacher-alone --no-overlap > /dev/null 2> $GRAFT_REPO_ROOT/$OUT/pmc_$K.err)
  echo "=== ET_PP_BUF=$K"
  python tools/pmc_by_grid.py $D --match conv_gemm,conv_wgrad_rs,bn_act_fwd | python -c "
import sys, json, collections
acc = collections.defaultdict(lambda: [0.0, 0.0, 0])
for l in sys.stdin:
    d = json.loads(l); k = d['kernel'].split('<')[0]
    acc[k][0] += d['GRBM_GUI_ACTIVE'] / 8 * d['launches']; acc[k][1] += d['us'] * d['launches']; acc[k][2] += d['launches']
for k, (cyc, us, n) in sorted(acc.items()): print('  %-28s x%5d  mean %7.2f us  clock %.3f GHz  cycles/launch %.0f' % (k, n, us / n, cyc / us / 1e3, cyc / n))
"
  find $D -name "*.db" -delete; find $D -name "*.csv" -delete
done | tee $OUT/clocks.txt
