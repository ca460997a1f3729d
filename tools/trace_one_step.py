"""One steady-state step of the main queue from a rocprofv3 --kernel-trace CSV, in launch order: offset, duration, gap before, kernel
(short), and which other queues were busy at that moment.  usage: python tools/trace_one_step.py <kernel_trace.csv> [min_gap_us]
The step boundary is the pack_input kernel (first launch of a step)."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
byq = defaultdict(list)
for r in rows:
    byq[r["Queue_Id"]].append(r)
main_q = max(byq, key=lambda q: len(byq[q]))
main = byq[main_q]
others = [r for q, v in byq.items() if q != main_q for r in v]
starts = [i for i, r in enumerate(main) if "pack_input" in r["Kernel_Name"]]
# steps begin with 3 pack launches close together: take the first of each cluster
firsts = [i for k, i in enumerate(starts) if k == 0 or main[i]["s"] - main[starts[k - 1]]["s"] > 5_000_000]
a, b = firsts[-3], firsts[-2]
step = main[a:b]
t0 = step[0]["s"]
print(f"step of {len(step)} main-queue kernels, span {(main[b]['s'] - t0) / 1e6:.2f} ms")
prev_e = t0
small_run, small_t = [], 0
for r in step:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")[:58]
    busy = sum(1 for o in others if o["s"] < r["e"] and o["e"] > r["s"])
    d, g = (r["e"] - r["s"]) / 1e3, (r["s"] - prev_e) / 1e3
    print(f"{(r['s'] - t0) / 1e3:9.1f} us  {d:8.1f} us  gap {g:6.1f}  others {busy:2d}  {n}")
    prev_e = max(prev_e, r["e"])
