#!/bin/bash
# is parity_check.fp16.conv_grad_cosine_vs_fp32_oracle sensitive to 1-ulp changes of the BatchNorm partial sums?  (the persistent-grid size regroups the partial rows)
OUT=gpurun_out/${TAG:-r06cos6}; mkdir -p $OUT
show() { python -c "import sys,json; d=json.loads(sys.stdin.readline()); c=d['parity_check']['fp16']['conv_grad_cosine_vs_fp32_oracle']; print('$1', 'min', round(c['min'],5), 'median', round(c['median'],5), c['worst_tensor'])"; }
for W in 300 200 150; do
  ET_CONV_S1_WGS=$W timeout 900 python bench.py --steps 5 --warmup 2 --no-teacher-alone 2>/dev/null | show "new library, ET_CONV_S1_WGS=$W" | tee -a $OUT/cos.txt
  ET_CONV_S1_WGS=$W ET_HIP_LIB=$GRAFT_REPO_ROOT/_old/efficientteacher_amd/libet_hip.so timeout 900 python bench.py --steps 5 --warmup 2 --no-teacher-alone 2>/dev/null | show "old library, ET_CONV_S1_WGS=$W" | tee -a $OUT/cos.txt
done
