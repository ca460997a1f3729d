"""rocprofv3 --kernel-trace CSV -> per (kernel, grid size, workgroup size): launches, mean / min / max duration, total ms.
usage: python tools/trace_by_grid.py <kernel_trace.csv> [substr,substr]   (launches of one kernel with different grids are different layers)"""
import csv
import sys
from collections import defaultdict


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


def main():
    match = sys.argv[2].split(",") if len(sys.argv) > 2 else None
    acc = defaultdict(list)
    rows = list(csv.DictReader(open(sys.argv[1])))
    if not rows:
        sys.exit("empty trace")
    gk = next((k for k in ("Grid_Size", "Grid_Size_X", "Grid") if k in rows[0]), None)
    wk = next((k for k in ("Workgroup_Size", "Workgroup_Size_X") if k in rows[0]), None)
    for r in rows:
        k = short(r.get("Kernel_Name", ""))
        if not k or (match and not any(m in k for m in match)):
            continue
        try:
            ns = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        except (KeyError, ValueError):
            continue
        acc[(k, r.get(gk, "?") if gk else "?", r.get(wk, "?") if wk else "?")].append(ns)
    tot = sum(sum(v) for v in acc.values())
    print(f"{tot / 1e6:.2f} ms of matching kernel time; columns: total ms, share, launches, mean / min / max us, kernel, grid, workgroup")
    for (k, g, w), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        print(f"{sum(v) / 1e6:9.3f} {sum(v) / tot:6.1%} {len(v):6d} {sum(v) / len(v) / 1e3:9.1f} {min(v) / 1e3:9.1f} {max(v) / 1e3:9.1f}  {k[:80]}  grid {g} wg {w}")


if __name__ == "__main__":
    main()
