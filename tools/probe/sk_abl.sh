#!/bin/bash
# where the stream-K kernel's extra time goes: ablation builds (50 no partial stores, 51 no partial loads / waits, 52 neither) and worker counts
OUT=gpurun_out/sk_abl; mkdir -p $OUT
show() { grep '^{' $1 | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l)
    if (r['cin'],r['cout'],r['k'],r['h']) in ((256,256,3,40),(512,512,3,20),(512,256,1,40),(512,1024,3,40),(1024,512,1,20)): print('   ',r['cin'],r['cout'],r['k'],r['s'],r['h'],r['fwd_kernel'][10:16],'fwd %.3f dgrad %.3f'%(r['fwd_ms'],r['dgrad_ms']))
"; }
echo "== SK=0"; ET_CONV_SK=0 MB_REF=0 MB_ONLY=pp timeout 300 python tools/microbench.py conv > $OUT/sk0.log 2>&1; show $OUT/sk0.log
echo "== SK=2"; ET_CONV_SK=2 MB_REF=0 MB_ONLY=pp timeout 300 python tools/microbench.py conv > $OUT/sk2.log 2>&1; show $OUT/sk2.log
for A in 50 51 52; do
  echo "== SK=2 ablation $A"; ET_HIP_LIB=$PWD/tools/probe/libet_abl$A.so ET_CONV_SK=2 MB_REF=0 MB_ONLY=pp timeout 300 python tools/microbench.py conv > $OUT/abl$A.log 2>&1; show $OUT/abl$A.log
done
for W in 200 128 64; do
  echo "== SK=2 workers $W"; ET_CONV_SK_WORKERS=$W ET_CONV_SK=2 MB_REF=0 MB_ONLY=pp timeout 300 python tools/microbench.py conv > $OUT/w$W.log 2>&1; show $OUT/w$W.log
done
