mb() { MB_REF=0 MB_ONLY=pp timeout 600 python tools/microbench.py conv 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().split('SUMMARY ')[1]); print(round(d['total_fwd_ms'],3), round(d['total_dgrad_ms'],3))"; }
python -m pytest tests/test_conv.py -x -q -m gpu 2>&1 | tail -1
echo -n "default: "; mb
for n in 35 29; do echo -n "ablate $n: "; ET_HIP_LIB=tools/probe/libet_abl$n.so mb; done
echo -n "default: "; mb
