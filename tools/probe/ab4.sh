mb() { MB_REF=0 timeout 900 python tools/microbench.py conv 2>&1 | grep '"cin"' | python -c "
import sys,json
tot={}
for l in sys.stdin:
    d=json.loads(l)
    k=d['fwd_kernel'][:60]
    if 'glds' not in k: continue
    a=tot.setdefault(k,[0.0,0.0]); a[0]+=d['fwd_ms']*d['count']; a[1]+=d['dgrad_ms']*d['count']
for k,v in sorted(tot.items()): print('   ',k, round(v[0],3), round(v[1],3))
print('    total', round(sum(v[0] for v in tot.values()),3), round(sum(v[1] for v in tot.values()),3))
"; }
ET_HIP_LIB=tools/probe/libet_abl40.so python -m pytest tests/test_conv.py -x -q -m gpu 2>&1 | tail -1
echo "default:"; mb
echo "ablate 40 (pieces between the k-steps):"; ET_HIP_LIB=tools/probe/libet_abl40.so mb
