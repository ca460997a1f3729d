set -u
OUT=gpurun_out/r06q; mkdir -p $OUT
for K in 0 1 0 1; do
  ( while true; do rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|junction" | tr '\n' ' '; echo; sleep 0.2; done ) > $OUT/smi_$K.txt 2>/dev/null &
  SMI=$!
  ET_PPRS_BUF=1 ET_PP_BUF=$K timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-teacher-alone --no-overlap 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('ET_PP_BUF=$K ms_per_step', round(d['ms_per_step'],2))"
  kill $SMI; wait $SMI 2>/dev/null
  python - <<PY
import re
P, S = [], []
for l in open("$OUT/smi_$K.txt"):
    m = re.search(r"Power[^:]*:\s*([0-9.]+)", l); s = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", l)
    if m: P.append(float(m.group(1)))
    if s: S.append(float(s.group(1)))
P = [p for p in P if p > 300]
print("   samples", len(P), "power W mean %.0f max %.0f" % (sum(P)/max(len(P),1), max(P or [0])), "| sclk MHz mean %.0f min %.0f max %.0f" % (sum(S)/max(len(S),1), min(S or [0]), max(S or [0])))
PY
done | tee $OUT/power.txt
head -3 $OUT/smi_0.txt | cut -c1-300
