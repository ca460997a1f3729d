"""per kernel name (substring match): launches, mean duration, mean gap BEFORE the launch and mean gap AFTER it on the same queue, from a
rocprofv3 --kernel-trace CSV.   usage: python tools/trace_kernel_gaps.py <kernel_trace.csv> substr[,substr]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
byq = collections.defaultdict(list)
for r in rows:
    byq[r["Queue_Id"]].append(r)
main = max(byq.values(), key=len)
main = main[len(main) // 3:]
subs = sys.argv[2].split(",") if len(sys.argv) > 2 else [""]
acc = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
for a, r, b in zip(main, main[1:], main[2:]):
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")
    key = next((s for s in subs if s in n), None)
    if key is None:
        key = "(other)"
    e = acc[key]
    e[0] += 1
    e[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    e[2] += max(0, int(r["Start_Timestamp"]) - int(a["End_Timestamp"]))
    e[3] += max(0, int(b["Start_Timestamp"]) - int(r["End_Timestamp"]))
t0, t1 = int(main[0]["Start_Timestamp"]), int(main[-1]["End_Timestamp"])
print(f"kernels {len(main)}  span {(t1 - t0) / 1e6:.2f} ms")
for k, (n, d, gb, ga) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:30s} x{n:6d}  mean {d / n / 1e3:8.2f} us  gap before {gb / n / 1e3:6.2f} us  gap after {ga / n / 1e3:6.2f} us   total {d / 1e6:8.2f} ms")
