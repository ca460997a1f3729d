"""Per-stream occupancy of the SSOD step from a rocprofv3 --kernel-trace CSV of bench.py: for the steady-state part of
the trace, per (queue/stream): kernels, busy time, union-of-all-streams busy time and idle gaps of the whole GPU.
usage: python tools/trace_streams.py <kernel_trace.csv> [steps_in_trace]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 13
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
# steady state: the last 60 % of the trace by time
t_lo = rows[0]["s"] + int(0.4 * (rows[-1]["e"] - rows[0]["s"]))
rows = [r for r in rows if r["s"] >= t_lo]
span = rows[-1]["e"] - rows[0]["s"]
byq = collections.defaultdict(list)
for r in rows:
    byq[(r.get("Queue_Id", "?"), r.get("Stream_Id", "?"))].append(r)
print(f"steady-state window {span / 1e6:.2f} ms, {len(rows)} kernels")
for q, rs in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    busy = sum(r["e"] - r["s"] for r in rs)
    print(f"  queue/stream {q}: {len(rs):6d} kernels  busy {busy / 1e6:8.2f} ms ({100 * busy / span:5.1f} % of the window)")
# union over all streams
ev = sorted([(r["s"], 1) for r in rows] + [(r["e"], -1) for r in rows])
depth, last, busy_any, busy2 = 0, ev[0][0], 0, 0
for t, d in ev:
    if depth >= 1:
        busy_any += t - last
    if depth >= 2:
        busy2 += t - last
    depth += d
    last = t
print(f"GPU busy (any stream) {busy_any / 1e6:.2f} ms = {100 * busy_any / span:.1f} %; >= 2 kernels in flight {100 * busy2 / span:.1f} %; idle {100 * (span - busy_any) / span:.1f} %")
# top kernels by total time in the window
agg = collections.defaultdict(lambda: [0, 0])
for r in rows:
    a = agg[r["Kernel_Name"][:70]]
    a[0] += r["e"] - r["s"]; a[1] += 1
tot = sum(a[0] for a in agg.values())
for k, (t, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:22]:
    print(f"  {100 * t / tot:5.1f} %  {t / 1e6:8.2f} ms  {n:6d} x {t / n / 1e3:8.1f} us  {k}")
