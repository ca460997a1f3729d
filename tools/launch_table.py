"""Per-shape table of the conv launches of ONE instrumented step (HIP events per launch).

  python bench.py --steps 12 --warmup 4 --no-cpu-baseline --dump-launches launches.json
  python tools/launch_table.py launches.json

Groups the launches by (kernel, op, shape), prints time, TFLOP/s and -- for the 256x256-tile kernels -- how full the last
residency round of the grid is (tiles / CUs), then totals by family (student / teacher = batch 64 / 32 at the bench's
default 32 + 32 images).  DESIGN.md section 8 quotes this table.
"""
import json
import math
import sys
from collections import defaultdict


def main(path, n_cu=256):
    rows = json.load(open(path))
    agg = defaultdict(lambda: [0, 0.0, 0.0, 0.0])
    fam = defaultdict(lambda: [0, 0.0, 0.0, 0.0])
    batches = sorted({r["shape"][1] for r in rows if r["shape"]})
    student_n = batches[-1] if batches else 0
    for r in rows:
        sh = tuple(r["shape"]) if r["shape"] else None
        a = agg[(r["kernel"], sh)]
        a[0] += r["launches"]; a[1] += r["ms"]; a[2] += r["flops"]; a[3] += r["bytes"]
        if sh:
            op, N, IH, IW, Cin, Cout, k, st = sh
            key = (op, f"k{k}", "student" if N == student_n else "teacher", "256-tile" if "pp_kernel" in r["kernel"] or "256, 256" in r["kernel"] else "128-tile")
        else:
            key = ("wgrad",)
        f = fam[key]
        f[0] += r["launches"]; f[1] += r["ms"]; f[2] += r["flops"]; f[3] += r["bytes"]
    print(f"{sum(v[1] for v in agg.values()):.2f} ms of conv launches in the step (both streams, overlap counted twice)\n")
    for (kern, sh), v in sorted(agg.items(), key=lambda x: -x[1][1])[:48]:
        extra = ""
        if sh and ("pp_kernel" in kern or "256, 256" in kern):
            op, N, IH, IW, Cin, Cout, k, st = sh
            M = N * IH * IW if (op == "dgrad" or st == 1) else N * (IH // st) * (IW // st)
            tiles = math.ceil(M / 256) * math.ceil((Cout if op == "fwd" else Cin) / 256)
            extra = f"  tiles {tiles} = {tiles / n_cu:.2f} rounds, last-round fill {tiles / n_cu / math.ceil(tiles / n_cu):.2f}"
        tb = f"  {v[3] / v[1] / 1e9:5.2f} TB/s alg." if v[3] else ""
        print(f"{v[1]:7.3f} ms  x{v[0]:3d}  {v[2] / v[1] / 1e9:6.0f} TFLOP/s{tb}  {kern[:46]:46s} {sh}{extra}")
    print()
    for key, v in sorted(fam.items(), key=lambda x: -x[1][1]):
        print(f"{v[1]:7.2f} ms  x{v[0]:4d}  {v[2] / v[1] / 1e9:6.0f} TFLOP/s  {v[3] / v[1] / 1e9:6.2f} TB/s algorithmic  {key}")


if __name__ == "__main__":
    main(sys.argv[1])
