"""Idle gaps on the main stream of the SSOD step, from a rocprofv3 --kernel-trace CSV of bench.py.
usage: python tools/trace_gaps.py <kernel_trace.csv>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
byq = collections.defaultdict(list)
for r in rows:
    byq[(r["Queue_Id"], r.get("Stream_Id", "0"))].append(r)
main = max(byq.values(), key=len)
# steady state: the last third of the main-stream kernels
main = main[len(main) * 2 // 3:]
t0, t1 = int(main[0]["Start_Timestamp"]), int(main[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in main)
gaps = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(main, main[1:])]
pos = [g for g in gaps if g > 0]
print(f"main-stream kernels {len(main)}  span {(t1 - t0) / 1e6:.2f} ms  busy {busy / 1e6:.2f} ms  idle {(t1 - t0 - busy) / 1e6:.2f} ms "
      f"({100 * (t1 - t0 - busy) / (t1 - t0):.1f} %)")
hist = collections.Counter(min(int(g / 1000), 50) for g in pos)
print("gap histogram (us: count):", sorted(hist.items())[:20])
big = sorted(((g, a["Kernel_Name"][:50], b["Kernel_Name"][:50]) for g, a, b in zip(gaps, main, main[1:]) if g > 20000), reverse=True)[:15]
for g, a, b in big:
    print(f"  {g / 1e3:8.1f} us between {a}  ->  {b}")
