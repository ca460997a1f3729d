"""Probe: which launches surround the small device-to-device copies / fills of a step?  Reads a rocprofv3 kernel trace (csv) and
prints, for every occurrence of the named runtime kernels, the (previous kernel, next kernel) pair on the same queue, aggregated.
    python tools/probe/copy_neighbors.py <kernel_trace.csv> [name-substring ...]
"""
import collections
import csv
import sys


def main(path, subs):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    by_q = collections.defaultdict(list)
    for r in rows:
        by_q[r.get("Queue_Id", "0")].append(r)
    pairs = collections.Counter()
    for q, rs in by_q.items():
        for i, r in enumerate(rs):
            n = r["Kernel_Name"]
            if any(s in n for s in subs):
                prev = rs[i - 1]["Kernel_Name"][:60] if i else "-"
                nxt = rs[i + 1]["Kernel_Name"][:60] if i + 1 < len(rs) else "-"
                pairs[(n[:40], prev, nxt)] += 1
    for (n, p, x), c in pairs.most_common(40):
        print(f"{c:6d}  {n:40s} after {p:60s} before {x}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:] or ["copyBuffer", "fillBuffer", "FillFunctor", "direct_copy"])
