for NK in 256 128 64; do echo "NARROW_K $NK"; ET_CONV_NARROW_K=$NK MB_REF=0 timeout 600 python tools/microbench.py conv 2>&1 | grep '"k": 1' | python -c "
import sys,json
tf=td=0
for l in sys.stdin:
    d=json.loads(l)
    if d['cin']*1<=256:
        print(d['cin'],d['cout'],d['h'],'x%d'%d['count'],d['fwd_kernel'][38:62],round(d['fwd_ms']*1e3,1),round(d['dgrad_ms']*1e3,1))
        tf+=d['fwd_ms']*d['count']; td+=d['dgrad_ms']*d['count']
print('total fwd',round(tf,3),'dgrad',round(td,3))
"; done
