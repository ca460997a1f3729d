set -u
OUT=gpurun_out/${TAG:-r06l}; mkdir -p $OUT
for B in 64 32; do for K in ${KNOB}=0 ${KNOB}=1 ${KNOB}=0 ${KNOB}=1; do echo "--- B=$B $K"; env $K MB_REF=0 MB_ROTATE=3 MB_B=$B MB_FULL=1 MB_ONLY="${MB_ONLY}" timeout 600 python tools/microbench.py conv 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print('%4d->%4d k%d s%d @%3d  fwd %6.1f us | dgrad %6.1f us | full-dgrad %6.1f | teacher-fwd %6.1f | %s' % (d['cin'], d['cout'], d['k'], d['s'], d['h'], d['fwd_ms']*1e3, d['dgrad_ms']*1e3, d.get('dgrad_full_ms',0)*1e3, d.get('fwd_teacher_ms',0)*1e3, d['fwd_kernel'][:40]))
    elif l.startswith('SUMMARY'): print(l.strip()[:160])
"; done; done | tee $OUT/mb_ab.txt
