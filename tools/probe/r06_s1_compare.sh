#!/bin/bash
OUT=gpurun_out/${TAG:-r06s1cmp}; mkdir -p $OUT
ET_HIP_LIB=$GRAFT_REPO_ROOT/_old/efficientteacher_amd/libet_hip.so python tools/probe/s1_compare.py /tmp/old.pt > $OUT/names.txt 2>&1
ET_CONV_BUF_DMA=0 python tools/probe/s1_compare.py /tmp/new_flat.pt > /dev/null 2>&1
python tools/probe/s1_compare.py /tmp/new_buf.pt > /dev/null 2>&1
python - <<'PY' | tee $OUT/cmp.txt
import torch
a=torch.load('/tmp/old.pt'); b=torch.load('/tmp/new_flat.pt'); c=torch.load('/tmp/new_buf.pt')
for k in a:
    ea = torch.equal(a[k], b[k]); ec = torch.equal(a[k], c[k])
    if k.endswith('dw'):
        da=(a[k]-b[k]).abs().max().item()/max(a[k].abs().max().item(),1e-9); dc=(a[k]-c[k]).abs().max().item()/max(a[k].abs().max().item(),1e-9)
        print(k, 'old vs new-flat rel', f'{da:.2e}', 'old vs new-buf rel', f'{dc:.2e}')
    else:
        print(k, 'old==new-flat', ea, 'old==new-buf', ec, '' if (ea and ec) else 'max abs %.4g %.4g' % ((a[k].float()-b[k].float()).abs().max().item(), (a[k].float()-c[k].float()).abs().max().item()))
PY
head -12 $OUT/names.txt
