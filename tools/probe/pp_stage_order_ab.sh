# ping-pong gather-GEMM: where the LDS-DMA pieces of a phase are issued (same box, isolated layers + step)
mb() { MB_REF=0 MB_ONLY=pp timeout 600 python tools/microbench.py conv 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().split('SUMMARY ')[1]); print(round(d['total_fwd_ms'],3), round(d['total_dgrad_ms'],3))"; }
echo -n "default (after MFMA 1,4): "; mb
for n in 29 31 32 33; do echo -n "ablate $n: "; ET_HIP_LIB=tools/probe/libet_abl$n.so mb; done
echo -n "default again: "; mb
