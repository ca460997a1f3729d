set -u
OUT=gpurun_out/${TAG:-r06lib}; mkdir -p $OUT
line() { python -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d['kernel_ms_by_family']; print('$1', round(d['ms_per_step'],2), k['main_stream'])"; }
timeout 900 python -m pytest ${TESTS:-tests/test_loss.py tests/test_ota.py tests/test_ssod_step.py tests/test_step_fullsize.py} -x -q -m gpu 2>&1 | tail -2 | tee $OUT/tests.txt
for i in 1 2 3; do for L in base new; do if [ $L = new ]; then unset ET_HIP_LIB; else export ET_HIP_LIB=$PWD/tools/probe/libet_base.so; fi; timeout 600 python bench.py --steps ${STEPS:-20} --warmup 5 --no-cpu-baseline --no-teacher-alone 2>/dev/null | line $L | tee -a $OUT/ab.txt; done; done; unset ET_HIP_LIB
