#!/bin/bash
OUT=gpurun_out/${TAG:-r06cos3}; mkdir -p $OUT
show() { python -c "import sys,json; d=json.loads(sys.stdin.readline()); c=d['parity_check']['fp16']['conv_grad_cosine_vs_fp32_oracle']; print('$1', 'min', round(c['min'],5), 'median', round(c['median'],5), c['worst_tensor'])"; }
ET_CONV_BUF_DMA=0 timeout 900 python bench.py --steps 5 --warmup 2 --no-teacher-alone --teacher-after p2 --set autograd.FUSE_BN_BWD_K=5 --set trainer.ssod_trainer.SSODTrainer.join_teacher_late=False 2>/dev/null | show "all four back (BUF=0, p2, K=5, early join)" | tee -a $OUT/cos.txt
timeout 900 python bench.py --steps 5 --warmup 2 --no-teacher-alone --teacher-after p2 2>/dev/null | show "p2 only" | tee -a $OUT/cos.txt
timeout 900 python bench.py --steps 5 --warmup 2 --no-teacher-alone --no-overlap 2>/dev/null | show "no overlap" | tee -a $OUT/cos.txt
