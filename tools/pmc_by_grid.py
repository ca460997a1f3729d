"""Per (kernel, grid size): mean counters and mean duration per launch from one or more rocprofv3 --pmc passes of the SAME command.
usage: python tools/pmc_by_grid.py <dir> [<dir> ...] [--match substr,substr]  -> JSON lines
Each counter_collection.csv row carries Grid_Size, Dispatch_Id and (with --kernel-trace) the start / end timestamps; launches of one
kernel that differ in their grid are different layers of the step (400 / 200 / 100 tiles ...), which the per-kernel means of
tools/pmc_summarize.py average away."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


def main():
    dirs = [a for a in sys.argv[1:] if not a.startswith("--")]
    match = None
    for i, a in enumerate(sys.argv):
        if a == "--match":
            match = sys.argv[i + 1].split(",")
    dirs = [d for d in dirs if not (match and d == ",".join(match))]
    acc = defaultdict(lambda: dict(c=defaultdict(float), n=defaultdict(int), ns=0.0, nn=0, seen=set()))
    for d in dirs:
        dur = {}
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                try:
                    dur[r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                except (KeyError, ValueError):
                    pass
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r.get("Kernel_Name", ""))
                if not k or (match and not any(m in k for m in match)):
                    continue
                key = (k, r.get("Grid_Size", "?"))
                e = acc[key]
                e["c"][r["Counter_Name"]] += float(r["Counter_Value"]); e["n"][r["Counter_Name"]] += 1
                did = (d, r.get("Dispatch_Id"))
                if did not in e["seen"]:
                    e["seen"].add(did)
                    ns = dur.get(r.get("Dispatch_Id"))
                    if ns is None:
                        try:
                            ns = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                        except (KeyError, ValueError):
                            ns = None
                    if ns:
                        e["ns"] += ns; e["nn"] += 1
    for (k, grid), e in sorted(acc.items(), key=lambda kv: -kv[1]["ns"]):
        out = dict(kernel=k, grid=grid, launches=max(e["n"].values()), us=round(e["ns"] / max(e["nn"], 1) / 1e3, 2))
        for c in sorted(e["c"]):
            out[c] = round(e["c"][c] / e["n"][c], 1)
        ga = out.get("GRBM_GUI_ACTIVE")
        if ga and out["us"]:
            out["clock_ghz"] = round(ga / 8.0 / (out["us"] * 1e3), 3)    # GRBM_GUI_ACTIVE sums the 8 XCDs (tools/pmc_mfma.py): cycles per ns
        h, m = out.get("TCC_HIT_sum", out.get("TCC_HIT")), out.get("TCC_MISS_sum", out.get("TCC_MISS"))
        if h is not None and m is not None and h + m > 0:
            out["l2_hit_rate"] = round(h / (h + m), 4)
        print(json.dumps(out))


if __name__ == "__main__":
    main()
