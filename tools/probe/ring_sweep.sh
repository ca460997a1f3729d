# 3x3 layers with < 256 output channels under forced LDS ring shapes (ET_CONV_RING=<rows><kvec><depth>), B=64
for R in 0 12883 12843 12844; do echo "RING $R"; ET_CONV_RING=$R MB_REF=0 timeout 600 python tools/microbench.py conv 2>&1 | grep '"k": 3' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l)
    if d['cout']<=128 and d['cin']>=64: print('  ',d['cin'],d['cout'],d['s'],d['h'],'x%d'%d['count'],d['fwd_kernel'][38:62],round(d['fwd_ms']*1e3,1),round(d['dgrad_ms']*1e3,1))
"; done
