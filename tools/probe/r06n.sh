set -u
OUT=gpurun_out/r06n; mkdir -p $OUT
for K in 0 1 0 1; do ET_PPRS_BUF=1 ET_PP_BUF=$K timeout 600 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-teacher-alone --no-overlap --dump-launches $OUT/l${K}.json > /dev/null 2>&1; python tools/launch_table.py $OUT/l${K}.json > $OUT/table_$K.txt 2>&1; done
python - <<'PY'
import json, collections
def load(p):
    d = json.load(open(p)); rows = d if isinstance(d, list) else d.get("launches", d)
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in rows:
        k = (r.get("kernel", r.get("tag", "?"))[:40], tuple(r.get("shape") or ()))
        agg[k][0] += r["ms"]; agg[k][1] += 1
    return agg
a, b = load("gpurun_out/r06n/l0.json"), load("gpurun_out/r06n/l1.json")
tot0 = tot1 = 0
for k in sorted(a, key=lambda k: -a[k][0]):
    if "conv_gemm_pp_" not in k[0] and "parity" not in k[0]: continue
    x, y = a[k][0], b.get(k, [0, 0])[0]
    tot0 += x; tot1 += y
    print(f"{k[0]:40s} {str(k[1]):40s} x{a[k][1]:3d}  flat {x*1e3:8.1f} us  buf {y*1e3:8.1f} us  {100*(y/x-1):+5.1f} %")
print("total pp ms", tot0, tot1)
PY
