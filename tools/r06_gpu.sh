#!/bin/bash
# Round-6 GPU sections (one gpurun call runs a list of them).  Usage: tools/r06_gpu.sh <tag> <section>...
# Everything lands in gpurun_out/<tag>/ (merged back by gpurun); nothing here reads /root/reference.
set -u
TAG=${1:-r06}; shift || true
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
run() { echo "=== $1 ($(date +%T))" | tee -a $OUT/log.txt; }
line() { python -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d['kernel_ms_by_family']; r=d['roofline']; print('$1', round(d['ms_per_step'],2), 'frac', round(r['frac'],4), 'conv_tf', round(r['all_conv_kernels']['tflops'],1), k['main_stream'], k['teacher_stream'])"; }
for s in "$@"; do
case $s in
env) run env; { nproc; free -g | head -2; python -c "import torchvision; print('torchvision', torchvision.__version__)" 2>&1 | tail -1; rocm-smi --showclocks 2>/dev/null | head -12; } > $OUT/env.txt 2>&1; cat $OUT/env.txt ;;
t_new) run t_new; timeout 1500 python -m pytest tests/test_step_benchbatch.py tests/test_nms_torchvision.py tests/test_abi.py -q -m gpu -s -rs > $OUT/pytest_new.log 2>&1; grep -E 'PARITY|passed|failed|SKIP|Error' $OUT/pytest_new.log | cut -c1-1500 ;;
t_conv) run t_conv; timeout 1200 python -m pytest tests/test_conv.py tests/test_conv_fuzz.py tests/test_norm_spatial.py -x -q -m gpu > $OUT/pytest_conv.log 2>&1; tail -5 $OUT/pytest_conv.log ;;
t_all) run t_all; timeout 2400 python -m pytest tests -x -q -m gpu -rs > $OUT/pytest_gpu.log 2>&1; tail -8 $OUT/pytest_gpu.log ;;
smoke) run smoke; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -3 $OUT/smoke.log ;;
bench) run bench; timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; cut -c1-400 $OUT/bench.json; tail -3 $OUT/bench.err ;;
bench_quick) run bench_quick; timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2> $OUT/bench_quick.err | tee $OUT/bench_quick.json | line quick ;;
launches) run launches; timeout 900 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-teacher-alone --dump-launches $OUT/launches_one_step.json > $OUT/bench_launches.json 2> $OUT/launches.err; python tools/launch_table.py $OUT/launches_one_step.json > $OUT/launch_table.txt 2>&1; tail -16 $OUT/launch_table.txt ;;
ab_env) run ab_env; for K in ${AB_ENV:-"X=0"}; do env $K timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | line "$K" | tee -a $OUT/ab_env.txt; done ;;
mb_k1) run mb_k1; for K in ${MB_K1_ENV:-"X=0"}; do echo "--- $K" | tee -a $OUT/mb_k1.txt; env $K MB_REF=0 MB_K=1 MB_ROTATE=3 MB_B=${MB_B:-64} timeout 600 python tools/microbench.py conv 2>&1 | python -c "
import sys, json, os
B=int(os.environ.get('MB_B','64'))
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); px=B*(d['h']//d['s'])**2
        byt=px*(d['cin']+d['cout'])*2
        full = ' | full-dgrad %6.1f us %5.2f TB/s teacher-fwd %6.1f us' % (d['dgrad_full_ms']*1e3, (byt + 2*px*d['cin']*2)/d['dgrad_full_ms']/1e9, d['fwd_teacher_ms']*1e3) if 'dgrad_full_ms' in d else ''
        print('%4d->%4d @%3d x%2d  fwd %6.1f us %5.2f TB/s %4.0f TF | dgrad %6.1f us %5.2f TB/s | wgrad %6.1f us %5.2f TB/s' % (d['cin'], d['cout'], d['h'], d['count'], d['fwd_ms']*1e3, byt/d['fwd_ms']/1e9, d['fwd_tf'], d['dgrad_ms']*1e3, byt/d['dgrad_ms']/1e9, d['wgrad_ms']*1e3, byt/d['wgrad_ms']/1e9) + full, d['fwd_kernel'][:48])
    elif l.startswith('SUMMARY'): print(l.strip()[:200])
" | tee -a $OUT/mb_k1.txt; done ;;
dp_sweep1) run dp_sweep1; timeout 1500 python tools/dp_sweep.py --gpus 1 --chunks 48 --channels 0,8 --steps 12 --warmup 4 --out $OUT/dp_sweep_1rank.json > $OUT/dp_sweep1.log 2>&1; tail -6 $OUT/dp_sweep1.log | cut -c1-400 ;;
t_par) run t_par; timeout 1200 python -m pytest tests/test_parallel.py -x -q -m gpu -s > $OUT/pytest_par.log 2>&1; tail -4 $OUT/pytest_par.log ;;
mb_k3w) run mb_k3w; for K in ${MB_K3_ENV:-"X=0"}; do echo "--- $K" | tee -a $OUT/mb_k3w.txt; env $K MB_REF=0 MB_K=3 timeout 600 python tools/microbench.py conv 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print('%4d->%4d s%d @%3d x%2d  fwd %6.1f us %4.0f TF | dgrad %6.1f us %4.0f TF | wgrad %6.1f us %4.0f TF' % (d['cin'], d['cout'], d['s'], d['h'], d['count'], d['fwd_ms']*1e3, d['fwd_tf'], d['dgrad_ms']*1e3, d['dgrad_tf'], d['wgrad_ms']*1e3, d['wgrad_tf']))
" | tee -a $OUT/mb_k3w.txt; done ;;
prof) run prof; (cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-teacher-alone > $GRAFT_REPO_ROOT/$OUT/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$OUT/prof.err); find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv; find $OUT/prof -name "*kernel_trace.csv" | head -1 | xargs -I{} python tools/trace_streams.py {} > $OUT/trace_streams.txt 2>&1; cat $OUT/trace_streams.txt; find $OUT/prof -name "*.db" -delete; find $OUT/prof -name "*kernel_trace.csv" -size +20M -delete; head -8 $OUT/kernel_stats.csv | cut -c1-160 ;;
pmc) run pmc; for C in FETCH_SIZE WRITE_SIZE; do (cd /tmp && timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_$C -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-teacher-alone > $GRAFT_REPO_ROOT/$OUT/bench_pmc_$C.json 2> $GRAFT_REPO_ROOT/$OUT/pmc_$C.err); done; python tools/pmc_summarize.py $OUT > $OUT/pmc_bench_summary.csv; find $OUT -name "*.db" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete; timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-teacher-alone --dump-launches $OUT/launches_for_pmc.json > /dev/null 2>&1; python tools/pmc_to_traffic.py $OUT/pmc_bench_summary.csv $OUT/pmc_traffic.json $OUT/launches_for_pmc.json ;;
pmc_mfma) run pmc_mfma; SETS=("SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES"); for C in "${SETS[@]}"; do rm -rf $OUT/pmc_mfma; (cd /tmp && timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_mfma -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-teacher-alone > $GRAFT_REPO_ROOT/$OUT/bench_pmc_mfma.json 2> $GRAFT_REPO_ROOT/$OUT/pmc_mfma.err); if find $OUT/pmc_mfma -name "*counter_collection.csv" -size +1k | grep -q .; then echo "counters: $C" | tee $OUT/pmc_mfma_counters.txt; break; else echo "pmc set failed: $C" | tee -a $OUT/log.txt; tail -3 $OUT/pmc_mfma.err; fi; done; [ -f $OUT/launches_one_step.json ] || timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-teacher-alone --dump-launches $OUT/launches_one_step.json > /dev/null 2>&1; python tools/pmc_mfma.py $OUT/pmc_mfma $OUT/pmc_mfma.json $OUT/launches_one_step.json | tee $OUT/pmc_mfma.txt; python tools/pmc_summarize.py $OUT/pmc_mfma > $OUT/pmc_mfma_summary.csv; find $OUT -name "*.db" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete ;;
pmc_list) run pmc_list; (cd /tmp && rocprofv3 -L 2>&1 | grep -iE "MFMA|SQ_BUSY_CU|GRBM_GUI|SQ_WAVE_CYCLES|SQ_WAIT_INST" | head -40) > $OUT/pmc_list.txt; head -40 $OUT/pmc_list.txt ;;
t_sel) run t_sel; timeout 1800 python -m pytest ${T_SEL:-tests/test_conv.py tests/test_ssod_step.py tests/test_parallel.py tests/test_nms.py tests/test_abi.py} -x -q -m gpu -s -rs > $OUT/pytest_sel.log 2>&1; grep -E "side-stream|passed|failed|SKIP|Error" $OUT/pytest_sel.log | cut -c1-600 | tail -12 ;;
iso) run iso; for M in hot rot nbr; do ISO_MODE=$M timeout 600 python tools/probe/iso_conv.py > $OUT/iso_$M.jsonl 2> $OUT/iso_$M.err; done; python - <<'PY' $OUT
import json, sys, collections
out = sys.argv[1]
rows = collections.defaultdict(dict)
for m in ("hot", "rot", "nbr"):
    for l in open(f"{out}/iso_{m}.jsonl"):
        if l.startswith("{"):
            d = json.loads(l); rows[(tuple(d["shape"]), d["dir"], d["kernel"][:40])][m] = d
for k, v in rows.items():
    print(k[0], k[1], k[2], " | ".join(f"{m} {v[m]['us_mean']:7.1f} us {v[m]['tflops']:6.0f} TF" for m in ("hot", "rot", "nbr") if m in v))
PY
;;
iso_pmc) run iso_pmc; SETS=("GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CU_CYCLES" "${TCC_SET:-TCC_HIT_sum TCC_MISS_sum}" "FETCH_SIZE"); for M in hot rot nbr; do i=0; DIRS=""; for C in "${SETS[@]}"; do i=$((i+1)); D=$OUT/isopmc_${M}_$i; rm -rf $D; (cd /tmp && ISO_MODE=$M ISO_ITERS=12 timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$D -o iso -- python $GRAFT_REPO_ROOT/tools/probe/iso_conv.py > /dev/null 2> $GRAFT_REPO_ROOT/$OUT/isopmc_${M}_$i.err); DIRS="$DIRS $D"; done; python tools/pmc_by_grid.py $DIRS --match conv_gemm > $OUT/isopmc_$M.jsonl; done; find $OUT -name "*.db" -delete; find $OUT -path "*isopmc_*" -name "*.csv" -delete; head -4 $OUT/isopmc_hot.jsonl | cut -c1-600 ;;
instep_pmc) run instep_pmc; SETS=("GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CU_CYCLES" "${TCC_SET:-TCC_HIT_sum TCC_MISS_sum}" "FETCH_SIZE"); i=0; DIRS=""; for C in "${SETS[@]}"; do i=$((i+1)); D=$OUT/instep_$i; rm -rf $D; (cd /tmp && timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$D -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-overlap --no-cpu-baseline --no-teacher-alone > $GRAFT_REPO_ROOT/$OUT/instep_bench_$i.json 2> $GRAFT_REPO_ROOT/$OUT/instep_$i.err); DIRS="$DIRS $D"; done; python tools/pmc_by_grid.py $DIRS --match conv_gemm_pprs,conv_gemm_pp_,conv_gemm_rs > $OUT/instep_pmc.jsonl; find $OUT -name "*.db" -delete; find $OUT -path "*instep_*" -name "*.csv" -delete; head -6 $OUT/instep_pmc.jsonl | cut -c1-600 ;;
tcc_list) run tcc_list; (cd /tmp && rocprofv3 -L 2>&1 | grep -iE "TCC_HIT|TCC_MISS|TCC_REQ|TCC_EA_RDREQ|MALL|TCP_TCC" | head -60) > $OUT/tcc_list.txt; head -30 $OUT/tcc_list.txt | cut -c1-200 ;;
*) echo "unknown section $s" ;;
esac
done
echo "=== done ($(date +%T))" | tee -a $OUT/log.txt
