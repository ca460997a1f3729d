import sys,json
for l in open(sys.argv[1]):
    if not l.startswith('{'): continue
    r=json.loads(l)
    if r['k']!=1: continue
    B=64; h=r['h']; by=B*h*h*(r['cin']+r['cout'])*2
    print(r['cin'],r['cout'],h,'x%d'%r['count'],r.get('fwd_kernel','')[-24:],'fwd %.1f us %.2f TB/s  dgrad %.1f us'%(r['fwd_ms']*1e3, by/r['fwd_ms']/1e9, r['dgrad_ms']*1e3))
