#!/bin/bash
# the main stream joins the teacher stream in front of the unsupervised loss instead of right behind the student's forward
OUT=gpurun_out/${TAG:-r06join}; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_ssod_step.py tests/test_step_fullsize.py tests/test_adapters.py tests/test_step_benchbatch.py -x -q -m gpu 2>&1 | tail -2 | tee $OUT/tests.txt
line() { python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$1', round(d['ms_per_step'],2))"; }
for S in 20 100; do for i in 1 2 3; do for E in False True; do
  timeout 600 python bench.py --steps $S --warmup 5 --no-cpu-baseline --no-teacher-alone --set trainer.ssod_trainer.SSODTrainer.join_teacher_late=$E 2>/dev/null | line "late=$E steps=$S" | tee -a $OUT/ab.txt
done; done; done
for i in 1 2; do for E in False True; do
  timeout 600 python bench.py --steps 20 --warmup 5 --dtype fp16 --no-cpu-baseline --no-teacher-alone --set trainer.ssod_trainer.SSODTrainer.join_teacher_late=$E 2>/dev/null | line "fp16 late=$E" | tee -a $OUT/ab.txt
done; done
