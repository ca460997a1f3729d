#!/bin/bash
# One gpurun call: parity tests + A/B microbenchmarks + step bench.  Usage: tools/gpu_round.sh <tag> [sections...]
# Everything lands in gpurun_out/<tag>/ (merged back by gpurun); nothing here reads /root/reference.
set -u
TAG=${1:-r02}; shift || true
SECTIONS=${*:-"conv_tests mb_pp mb_lock host bench"}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
run() { echo "=== $1 ($(date +%T))" | tee -a $OUT/log.txt; }
for s in $SECTIONS; do
case $s in
conv_tests) run conv_tests; timeout 900 python -m pytest tests/test_conv.py -x -q -m gpu > $OUT/pytest_conv.log 2>&1; tail -3 $OUT/pytest_conv.log ;;
ab_swp) run ab_swp; for R in ${AB_SWP_MB:-0 7 0 7}; do echo "--- ET_CONV_SWP=$R"; ET_CONV_SWP=$R MB_REF=0 MB_WGRAD=0 MB_ROTATE=4 timeout 600 python tools/microbench.py conv 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print(d['cin'], d['cout'], d['k'], d['s'], d['h'], 'x%d' % d['count'], 'fwd %.1f us %.0f TF  dgrad %.1f us %.0f TF' % (d['fwd_ms'] * 1e3, d['fwd_tf'], d['dgrad_ms'] * 1e3, d['dgrad_tf']), d['fwd_kernel'][:40])
    elif l.startswith('SUMMARY'): print(l.strip()[:200])
" | tee -a $OUT/ab_swp_mb.txt; done; for R in ${AB_SWP_STEP:-0 7 0 7 1 2 4}; do ET_CONV_SWP=$R timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('ET_CONV_SWP=$R', round(d['ms_per_step'],2), round(d['roofline']['frac'],4), d['roofline']['kernel'], round(d['roofline']['all_conv_kernels']['tflops'],1), d['kernel_ms_by_family']['main_stream'], d['kernel_ms_by_family']['teacher_stream'])" | tee -a $OUT/ab_swp_step.txt; done ;;
ab_swp_abl) run ab_swp_abl; for R in ${AB_SWP_ABL:-0: 7: 7:swpabl1 0: 7: 7:swpabl1}; do M=${R%%:*}; L=${R#*:}; if [ -n "$L" ]; then export ET_HIP_LIB=$PWD/tools/probe/libet_$L.so; else unset ET_HIP_LIB; fi; echo "--- ET_CONV_SWP=$M lib=$L" | tee -a $OUT/ab_swp_abl.txt; ET_CONV_SWP=$M MB_REF=0 MB_WGRAD=0 MB_ROTATE=4 timeout 600 python tools/microbench.py conv 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print(d['cin'], d['cout'], d['k'], d['s'], d['h'], 'x%d' % d['count'], 'fwd %.1f us %.0f TF  dgrad %.1f us %.0f TF' % (d['fwd_ms'] * 1e3, d['fwd_tf'], d['dgrad_ms'] * 1e3, d['dgrad_tf']), d['fwd_kernel'][:40])
    elif l.startswith('SUMMARY'): print(l.strip()[:200])
" | tee -a $OUT/ab_swp_abl.txt; done; unset ET_HIP_LIB ;;
ab_fin) run ab_fin; for R in ${AB_FIN:-base: new:0 new:2048 new:8192 base: new:0 new:2048 new:8192}; do L=${R%%:*}; F=${R#*:}; if [ $L = base ]; then export ET_HIP_LIB=$PWD/tools/probe/libet_base.so; else unset ET_HIP_LIB; fi; if [ -n "$F" ]; then export ET_BN_FIN_SMALL=$F; else unset ET_BN_FIN_SMALL; fi; timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d['kernel_ms_by_family']; print('$R', round(d['ms_per_step'],2), round(d['roofline']['frac'],4), round(d['roofline']['all_conv_kernels']['tflops'],1), k['main_stream'], k['teacher_stream'])" | tee -a $OUT/ab_fin.txt; done; unset ET_HIP_LIB ET_BN_FIN_SMALL ;;
ab_wgroup) run ab_wgroup; for K in ${AB_WGROUP:-8 4 2 1 8 4 2 1}; do ET_WGRAD_GROUP=$K timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d['kernel_ms_by_family']; print('ET_WGRAD_GROUP=$K', round(d['ms_per_step'],2), round(d['roofline']['frac'],4), round(d['roofline']['all_conv_kernels']['tflops'],1), k['main_stream'], k['teacher_stream'])" | tee -a $OUT/ab_wgroup.txt; done ;;
mb_wgrad_ident) run mb_wgrad_ident; for L in base new base new; do if [ $L = base ]; then export ET_HIP_LIB=$PWD/tools/probe/libet_base.so; else unset ET_HIP_LIB; fi; echo "--- $L" | tee -a $OUT/mb_wgrad_ident.txt; MB_REF=0 MB_K=1 timeout 600 python tools/microbench.py conv 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print(d['cin'], d['cout'], d['k'], d['s'], d['h'], 'x%d' % d['count'], 'wgrad %.1f us %.0f TF' % (d['wgrad_ms'] * 1e3, d['wgrad_tf']))
    elif l.startswith('SUMMARY'): print(l.strip()[:160])
" | tee -a $OUT/mb_wgrad_ident.txt; done; unset ET_HIP_LIB ;;
ab_s2slice) run ab_s2slice; for K in ${AB_S2SLICE:-0 49152 24576 98304 0 49152 24576 98304}; do ET_DGRAD_S2_SLICE_KB=$K timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d['kernel_ms_by_family']; print('ET_DGRAD_S2_SLICE_KB=$K', round(d['ms_per_step'],2), round(d['roofline']['frac'],4), round(d['roofline']['all_conv_kernels']['tflops'],1), k['main_stream'], k['teacher_stream'])" | tee -a $OUT/ab_s2slice.txt; done; for K in 0 49152 0 49152; do echo "--- slice $K" | tee -a $OUT/ab_s2slice.txt; ET_DGRAD_S2_SLICE_KB=$K MB_REF=0 MB_WGRAD=0 MB_K=3 MB_ROTATE=4 timeout 600 python tools/microbench.py conv 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        if d['s'] == 2: print(d['cin'], d['cout'], d['k'], d['s'], d['h'], 'x%d' % d['count'], 'fwd %.1f us  dgrad %.1f us %.0f TF' % (d['fwd_ms'] * 1e3, d['dgrad_ms'] * 1e3, d['dgrad_tf']))
" | tee -a $OUT/ab_s2slice.txt; done ;;
ab_bnrev) run ab_bnrev; for K in ${AB_BNREV:-0 1 3 7 0 1 3 7}; do ET_BN_REVERSE=$K timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d['kernel_ms_by_family']; print('ET_BN_REVERSE=$K', round(d['ms_per_step'],2), round(d['roofline']['frac'],4), round(d['roofline']['all_conv_kernels']['tflops'],1), k['main_stream'], k['teacher_stream'])" | tee -a $OUT/ab_bnrev.txt; done ;;
ab_minfill) run ab_minfill; for K in ${AB_MINFILL:-0 45 80 0 45 80}; do ET_CONV_BIG_MINFILL=$K timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d['kernel_ms_by_family']; print('ET_CONV_BIG_MINFILL=$K', round(d['ms_per_step'],2), round(d['roofline']['frac'],4), d['roofline']['kernel'], round(d['roofline']['all_conv_kernels']['tflops'],1), k['main_stream'], k['teacher_stream'])" | tee -a $OUT/ab_minfill.txt; done ;;
ab_fusek) run ab_fusek; for K in ${AB_FUSEK:-3 1 2 0 3 1 2 0}; do ET_FUSE_BN_BWD_K=$K timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d['kernel_ms_by_family']; print('ET_FUSE_BN_BWD_K=$K', round(d['ms_per_step'],2), round(d['roofline']['frac'],4), round(d['roofline']['all_conv_kernels']['tflops'],1), k['main_stream'], k['teacher_stream'])" | tee -a $OUT/ab_fusek.txt; done ;;
ab_env) run ab_env; for K in ${AB_ENV:-"X=0"}; do env $K timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d['kernel_ms_by_family']; print('$K', round(d['ms_per_step'],2), round(d['roofline']['frac'],4), round(d['roofline']['all_conv_kernels']['tflops'],1), k['main_stream'], k['teacher_stream'])" | tee -a $OUT/ab_env.txt; done ;;
ab_libs) run ab_libs; for L in ${AB_LIBS:-base new base new}; do if [ $L = new ]; then unset ET_HIP_LIB; else export ET_HIP_LIB=$PWD/tools/probe/libet_$L.so; fi; timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d['kernel_ms_by_family']; print('$L', round(d['ms_per_step'],2), round(d['roofline']['frac'],4), round(d['roofline']['all_conv_kernels']['tflops'],1), k['main_stream'], k['teacher_stream'])" | tee -a $OUT/ab_libs.txt; done; unset ET_HIP_LIB ;;
all_tests) run all_tests; timeout 2400 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; tail -5 $OUT/pytest_gpu.log ;;
mb_pp) run mb_pp; MB_REF=0 MB_ONLY=${MB_ONLY:-pp} timeout 600 python tools/microbench.py conv > $OUT/mb_pp1.log 2>&1; tail -1 $OUT/mb_pp1.log ;;
mb_lock) run mb_lock; ET_CONV_PP=0 MB_REF=0 MB_ONLY=${MB_ONLY:-"256, 256"} timeout 600 python tools/microbench.py conv > $OUT/mb_pp0.log 2>&1; tail -1 $OUT/mb_pp0.log ;;
mb_stem) run mb_stem; MB_REF=0 MB_ONLY=stem timeout 600 python tools/microbench.py conv > $OUT/mb_stem.log 2>&1; grep -v "^$" $OUT/mb_stem.log | tail -3 | cut -c1-400; ET_CONV_STEM=0 MB_REF=0 MB_ONLY="4, 2, false" timeout 600 python tools/microbench.py conv > $OUT/mb_stem_off.log 2>&1; grep -v "^$" $OUT/mb_stem_off.log | tail -3 | cut -c1-400 ;;
bench_nostem) run bench_nostem; ET_CONV_STEM=0 timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_nostem.json 2> $OUT/bench_nostem.err; cut -c1-300 $OUT/bench_nostem.json ;;
mb_fill) run mb_fill; for F in 0 45 80; do echo "MINFILL $F" >> $OUT/mb_fill.log; ET_CONV_BIG_MINFILL=$F MB_B=32 MB_REF=0 timeout 600 python tools/microbench.py conv 2>&1 | grep -E '"k": 3|"k": 1' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l)
    if d['h']<=40 and d['cout']>=256: print(d['cin'],d['cout'],d['k'],d['s'],d['h'],d['fwd_kernel'][:32],round(d['fwd_ms']*1e3,1),round(d['dgrad_ms']*1e3,1))
" >> $OUT/mb_fill.log; done; cat $OUT/mb_fill.log ;;
bench_fill) run bench_fill; for F in 45 80; do ET_CONV_BIG_MINFILL=$F timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_fill$F.json 2> $OUT/bench_fill$F.err; cut -c1-200 $OUT/bench_fill$F.json; done ;;
mb_all) run mb_all; MB_REF=0 timeout 900 python tools/microbench.py conv > $OUT/mb_all.log 2>&1; tail -1 $OUT/mb_all.log ;;
mb_bn) run mb_bn; timeout 600 python tools/microbench.py bn > $OUT/mb_bn.log 2>&1; tail -1 $OUT/mb_bn.log ;;
host) run host; timeout 600 python tools/host_bound.py > $OUT/host_bound.log 2>&1; tail -1 $OUT/host_bound.log ;;
bench) run bench; timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; cut -c1-400 $OUT/bench.json; tail -3 $OUT/bench.err ;;
bench_lock) run bench_lock; ET_CONV_PP=0 timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_pp0.json 2> $OUT/bench_pp0.err; cut -c1-300 $OUT/bench_pp0.json ;;
bench_nostream) run bench_nostream; ET_WGRAD_STREAM=0 timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_nostream.json 2> $OUT/bench_nostream.err; cut -c1-300 $OUT/bench_nostream.json ;;
bench_nofuse) run bench_nofuse; ET_FUSE_BN_BWD=0 timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_nofuse.json 2> $OUT/bench_nofuse.err; cut -c1-300 $OUT/bench_nofuse.json ;;
bench_graph) run bench_graph; timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --graph > $OUT/bench_graph.json 2> $OUT/bench_graph.err; cut -c1-300 $OUT/bench_graph.json ;;
bench_host) run bench_host; timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --host-inputs > $OUT/bench_host_inputs.json 2> $OUT/bench_host.err; cut -c1-300 $OUT/bench_host_inputs.json ;;
bench_quick) run bench_quick; timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_quick.json 2> $OUT/bench_quick.err; cut -c1-300 $OUT/bench_quick.json ;;
bench_v5s) run bench_v5s; timeout 900 python bench.py --workload v5s-sup --steps 20 --warmup 5 > $OUT/bench_v5s.json 2> $OUT/bench_v5s.err; cut -c1-500 $OUT/bench_v5s.json; tail -3 $OUT/bench_v5s.err ;;
bench_v8) run bench_v8; timeout 900 python bench.py --workload v8-sup --steps 20 --warmup 5 > $OUT/bench_v8.json 2> $OUT/bench_v8.err; cut -c1-500 $OUT/bench_v8.json; tail -3 $OUT/bench_v8.err ;;
bench_dp1) run bench_dp1; timeout 900 python bench.py --force-dp --no-graph --per-rank 16 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_dp1_eager.json 2> $OUT/bench_dp1_eager.err; cut -c1-300 $OUT/bench_dp1_eager.json; tail -3 $OUT/bench_dp1_eager.err ;;
bench_dp1g) run bench_dp1g; timeout 900 python bench.py --force-dp --graph --per-rank 16 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_dp1_graph.json 2> $OUT/bench_dp1_graph.err; cut -c1-300 $OUT/bench_dp1_graph.json; tail -3 $OUT/bench_dp1_graph.err ;;
bench16) run bench16; timeout 900 python bench.py --per-rank 16 --no-graph --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench16_eager.json 2> $OUT/bench16_eager.err; cut -c1-300 $OUT/bench16_eager.json; timeout 900 python bench.py --per-rank 16 --graph --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench16_graph.json 2> $OUT/bench16_graph.err; cut -c1-300 $OUT/bench16_graph.json ;;
t_par) run t_par; timeout 900 python -m pytest tests/test_parallel.py tests/test_step_fullsize.py tests/test_input_path.py tests/test_labelmatch.py -x -q -m gpu -s > $OUT/pytest_sel.log 2>&1; grep -E 'PARITY|dp\+graph|passed|failed' $OUT/pytest_sel.log | cut -c1-600 ;;
ab_bn) run ab_bn; for L in base new base new; do if [ $L = base ]; then export ET_HIP_LIB=$PWD/tools/probe/libet_base.so; else unset ET_HIP_LIB; fi; timeout 300 python tools/microbench.py bn 2>&1 | tail -1 | cut -c1-200; timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$L', round(d['ms_per_step'],2), d['kernel_ms_by_family']['main_stream'])"; done; unset ET_HIP_LIB ;;
knobs) run knobs; for K in "X=0" "ET_WGRAD_STREAM=1" "ET_WGRAD_STREAM=1 ET_WGRAD_GROUP=1" "ET_WGRAD_STREAM=1 ET_WGRAD_GROUP=4" "X=0"; do env $K timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$K', round(d['ms_per_step'],2), d['kernel_ms_by_family']['main_stream'], d['kernel_ms_by_family']['teacher_stream_ms'])"; done ;;
bench_v8s) run bench_v8s; timeout 900 python bench.py --workload v8-ssod --steps 20 --warmup 5 > $OUT/bench_v8_ssod.json 2> $OUT/bench_v8_ssod.err; cut -c1-500 $OUT/bench_v8_ssod.json; tail -3 $OUT/bench_v8_ssod.err ;;
knobs2) run knobs2; for K in "X=0" "ET_FUSE_BN_BWD=0" "ET_CONV_BIG_MINFILL=45" "ET_WGRAD_GROUP=16" "X=0"; do env $K timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$K', round(d['ms_per_step'],2), d['kernel_ms_by_family']['main_stream'], d['kernel_ms_by_family']['teacher_stream_ms'])"; done ;;
knobs16) run knobs16; for K in "X=0" "ET_CONV_BIG_MINFILL=45" "ET_CONV_BIG_MINFILL=80" "X=0" "ET_CONV_BIG_MINFILL=45"; do env $K timeout 600 python bench.py --per-rank 16 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$K', round(d['ms_per_step'],2), d['kernel_ms_by_family']['main_stream'], d['kernel_ms_by_family']['teacher_stream_ms'], d['config']['step_graph']['enabled'])"; done ;;
first16) run first16; for K in "--graph" "--graph" "--no-graph"; do timeout 600 python bench.py --per-rank 16 $K --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$K', round(d['ms_per_step'],2), d['config']['step_graph'])"; done ;;
ab_fork) run ab_fork; for K in "X=0" "ET_GRAD_FORK=0" "X=0" "ET_GRAD_FORK=0"; do env $K timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d['kernel_ms_by_family']; print('$K', round(d['ms_per_step'],2), k['main_stream'], k['aten_and_gaps_ms'])"; done ;;
ab_lib) run ab_lib; for L in base new base new; do if [ $L = base ]; then export ET_HIP_LIB=$PWD/tools/probe/libet_base.so; else unset ET_HIP_LIB; fi; MB_REF=0 timeout 600 python tools/microbench.py conv 2>&1 | tail -1 | cut -c1-230; timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d['kernel_ms_by_family']; print('$L', round(d['ms_per_step'],2), k['main_stream'], k['teacher_stream_ms'], round(d['roofline']['frac'],4))"; done; unset ET_HIP_LIB ;;
launches) run launches; timeout 900 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --dump-launches $OUT/launches_one_step.json > $OUT/bench_launches.json 2> $OUT/launches.err; python tools/launch_table.py $OUT/launches_one_step.json > $OUT/launch_table.txt 2>&1; tail -25 $OUT/launch_table.txt ;;
ab_teacher) run ab_teacher; for K in ${AB_TEACHER:-"X=0" "ET_TEACHER_AFTER=p2" "ET_TEACHER_AFTER=p1" "X=0" "ET_TEACHER_AFTER=p2" "ET_TEACHER_AFTER=p1" "X=0" "ET_TEACHER_AFTER=p2"}; do env $K timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d['kernel_ms_by_family']; print('$K', round(d['ms_per_step'],2), round(d['roofline']['frac'],4), round(d['roofline']['all_conv_kernels']['tflops']), k['main_stream'], k['teacher_stream_ms'])"; done ;;
graph_ab) run graph_ab; for K in "ET_TEACHER_AFTER=p2 --per-rank 32" "ET_TEACHER_AFTER=start --per-rank 32" "ET_TEACHER_AFTER=p2 --per-rank 16" "ET_TEACHER_AFTER=start --per-rank 16" "ET_TEACHER_AFTER=p2 --per-rank 24"; do E=${K%% *}; A=${K#* }; env $E timeout 600 python bench.py --graph $A --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$K', round(d['ms_per_step'],2), d['config']['step_graph'])"; done ;;
graph_rep) run graph_rep; for K in 32 32 32 16 16 16 32 16; do timeout 600 python bench.py --graph --per-rank $K --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$K', round(d['ms_per_step'],2), d['config']['step_graph'])" | tee -a $OUT/graph_rep.txt; done ;;
t_graph) run t_graph; timeout 900 python -m pytest tests/test_ssod_step.py -m gpu -x -q -k graph 2>&1 | tail -5 ;;
ab_rs) run ab_rs; for R in 0 1 0 1; do echo "--- ET_CONV_RS=$R"; ET_CONV_RS=$R MB_REF=0 MB_K=3 MB_ROTATE=4 timeout 600 python tools/microbench.py conv 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        if d['s'] == 1: print(d['cin'], d['cout'], d['h'], 'fwd %.1f us %.0f TF  dgrad %.1f us %.0f TF' % (d['fwd_ms'] * 1e3, d['fwd_tf'], d['dgrad_ms'] * 1e3, d['dgrad_tf']), d['fwd_kernel'][:34])
    elif l.startswith('SUMMARY'): print(l.strip()[:200])
" | tee -a $OUT/ab_rs_mb.txt; done; for R in 0 1 0 1; do ET_CONV_RS=$R timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('ET_CONV_RS=$R', round(d['ms_per_step'],2), round(d['roofline']['frac'],4), d['roofline']['all_conv_kernels']['tflops'], d['kernel_ms_by_family']['main_stream'], d['kernel_ms_by_family']['teacher_stream_ms'])" | tee -a $OUT/ab_rs_step.txt; done ;;
mb_rs) run mb_rs; for R in 0 1 0 1; do echo "--- ET_CONV_RS=$R"; ET_CONV_RS=$R MB_REF=0 MB_K=3 MB_ROTATE=4 timeout 600 python tools/microbench.py conv 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        if d['s'] == 1 and d['cin'] < 256: print(d['cin'], d['cout'], d['h'], 'fwd %.1f us %.0f TF  dgrad %.1f us %.0f TF' % (d['fwd_ms'] * 1e3, d['fwd_tf'], d['dgrad_ms'] * 1e3, d['dgrad_tf']), d['fwd_kernel'][:34])
" | tee -a $OUT/mb_rs.txt; done ;;
mb_rs_abl) run mb_rs_abl; for L in ${RS_LIBS:-base new base new}; do if [ $L = new ]; then unset ET_HIP_LIB; export ET_CONV_RS=1; elif [ $L = base ]; then unset ET_HIP_LIB; export ET_CONV_RS=0; else export ET_HIP_LIB=$PWD/tools/probe/libet_$L.so; export ET_CONV_RS=1; fi; echo "--- $L"; MB_REF=0 MB_K=3 MB_ROTATE=4 timeout 600 python tools/microbench.py conv 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        if d['s'] == 1 and d['cin'] < 256: print(d['cin'], d['cout'], d['h'], 'fwd %.1f us %.0f TF  dgrad %.1f us %.0f TF' % (d['fwd_ms'] * 1e3, d['fwd_tf'], d['dgrad_ms'] * 1e3, d['dgrad_tf']), d['fwd_kernel'][:34])
" | tee -a $OUT/mb_rs_abl.txt; done ;;
ab_rs_step) run ab_rs_step; for R in 0 1 0 1 0 1; do ET_CONV_RS=$R timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('ET_CONV_RS=$R', round(d['ms_per_step'],2), round(d['roofline']['frac'],4), round(d['roofline']['all_conv_kernels']['tflops'],1), d['kernel_ms_by_family']['main_stream'], d['kernel_ms_by_family']['teacher_stream_ms'])" | tee -a $OUT/ab_rs_step.txt; done ;;
mb_pp_abl) run mb_pp_abl; for L in base rs8 base rs8; do if [ $L = base ]; then unset ET_HIP_LIB; else export ET_HIP_LIB=$PWD/tools/probe/libet_$L.so; fi; echo "--- $L"; MB_REF=0 MB_K=3 MB_ROTATE=4 timeout 600 python tools/microbench.py conv 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        if d['s'] == 1 and d['cin'] >= 256: print(d['cin'], d['cout'], d['h'], 'fwd %.1f us %.0f TF  dgrad %.1f us %.0f TF' % (d['fwd_ms'] * 1e3, d['fwd_tf'], d['dgrad_ms'] * 1e3, d['dgrad_tf']), d['fwd_kernel'][:34])
" | tee -a $OUT/mb_pp_abl.txt; done ;;
ab_pprs) run ab_pprs; for R in 0 1 0 1; do echo "--- ET_CONV_PPRS=$R"; ET_CONV_PPRS=$R MB_REF=0 MB_K=3 MB_ROTATE=4 timeout 600 python tools/microbench.py conv 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        if d['s'] == 1 and d['cin'] >= 256: print(d['cin'], d['cout'], d['h'], 'fwd %.1f us %.0f TF  dgrad %.1f us %.0f TF' % (d['fwd_ms'] * 1e3, d['fwd_tf'], d['dgrad_ms'] * 1e3, d['dgrad_tf']), d['fwd_kernel'][:34])
" | tee -a $OUT/ab_pprs_mb.txt; done; for R in 0 1 0 1 0 1; do ET_CONV_PPRS=$R timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('ET_CONV_PPRS=$R', round(d['ms_per_step'],2), d['roofline']['kernel'], round(d['roofline']['frac'],4), round(d['roofline']['all_conv_kernels']['tflops'],1), d['kernel_ms_by_family']['main_stream'], d['kernel_ms_by_family']['teacher_stream_ms'])" | tee -a $OUT/ab_pprs_step.txt; done ;;
ab_wrs) run ab_wrs; for R in ${WRS_MB:-0 3 0 3}; do echo "--- ET_WGRAD_RS=$R"; ET_WGRAD_RS=$R MB_REF=0 MB_K=3 timeout 600 python tools/microbench.py conv 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        if d['s'] == 1: print(d['cin'], d['cout'], d['h'], 'wgrad %.1f us %.0f TF' % (d['wgrad_ms'] * 1e3, d['wgrad_tf']))
" | tee -a $OUT/ab_wrs_mb.txt; done; for R in ${WRS_STEP:-0 3 1 2 0 3}; do ET_WGRAD_RS=$R timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('ET_WGRAD_RS=$R', round(d['ms_per_step'],2), round(d['roofline']['frac'],4), round(d['roofline']['all_conv_kernels']['tflops'],1), d['kernel_ms_by_family']['main_stream'], d['kernel_ms_by_family']['teacher_stream_ms'])" | tee -a $OUT/ab_wrs_step.txt; done ;;
ab_wlib) run ab_wlib; for L in base new base new; do if [ $L = base ]; then export ET_HIP_LIB=$PWD/tools/probe/libet_base.so; else unset ET_HIP_LIB; fi; echo "--- $L"; MB_REF=0 MB_K=3 timeout 600 python tools/microbench.py conv 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        if d['s'] == 1: print(d['cin'], d['cout'], d['h'], 'wgrad %.1f us %.0f TF' % (d['wgrad_ms'] * 1e3, d['wgrad_tf']))
" | tee -a $OUT/ab_wlib_mb.txt; done; for L in base new base new; do if [ $L = base ]; then export ET_HIP_LIB=$PWD/tools/probe/libet_base.so; else unset ET_HIP_LIB; fi; timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$L', round(d['ms_per_step'],2), round(d['roofline']['frac'],4), round(d['roofline']['all_conv_kernels']['tflops'],1), d['kernel_ms_by_family']['main_stream'], d['kernel_ms_by_family']['teacher_stream_ms'])" | tee -a $OUT/ab_wlib_step.txt; done ;;
mb_k1) run mb_k1; MB_REF=0 MB_K=1 MB_ROTATE=3 timeout 600 python tools/microbench.py conv 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); B=64; px=B*(d['h']//d['s'])**2
        byt=px*(d['cin']+d['cout'])*2
        print('%4d->%4d @%3d x%2d  fwd %6.1f us %5.2f TB/s %4.0f TF | dgrad %6.1f us %5.2f TB/s | wgrad %6.1f us %5.2f TB/s' % (d['cin'], d['cout'], d['h'], d['count'], d['fwd_ms']*1e3, byt/d['fwd_ms']/1e9, d['fwd_tf'], d['dgrad_ms']*1e3, byt/d['dgrad_ms']/1e9, d['wgrad_ms']*1e3, byt/d['wgrad_ms']/1e9), d['fwd_kernel'][22:60])
" | tee $OUT/mb_k1.txt ;;
graph_seq) run graph_seq; for C in "--force-dp --graph --per-rank 16" "--graph" "--graph" "--force-dp --per-rank 16" "--graph" "" "--graph"; do echo "--- bench.py $C"; ET_BENCH_STEP_TIMES=1 timeout 600 python bench.py $C --steps 20 --warmup 5 --no-cpu-baseline 2> $OUT/gs.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['ms_per_step'],2), d['config']['step_graph'].get('replay_probe'))"; grep "step enqueue" $OUT/gs.err; done 2>&1 | tee $OUT/graph_seq.txt ;;
smoke) run smoke; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -3 $OUT/smoke.log ;;
prof) run prof; (cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$OUT/prof.err); find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv; find $OUT/prof -name "*kernel_trace.csv" | head -1 | xargs -I{} python tools/trace_streams.py {} > $OUT/trace_streams.txt 2>&1; cat $OUT/trace_streams.txt; find $OUT/prof -name "*.db" -delete; find $OUT/prof -name "*kernel_trace.csv" -size +20M -delete; head -8 $OUT/kernel_stats.csv | cut -c1-160 ;;
pmc) run pmc; for C in FETCH_SIZE WRITE_SIZE; do (cd /tmp && timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_$C -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/bench_pmc_$C.json 2> $GRAFT_REPO_ROOT/$OUT/pmc_$C.err); done; python tools/pmc_summarize.py $OUT > $OUT/pmc_bench_summary.csv; find $OUT -name "*.db" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete; head -12 $OUT/pmc_bench_summary.csv ;;
*) echo "unknown section $s" ;;
esac
done
echo "=== done ($(date +%T))" | tee -a $OUT/log.txt
