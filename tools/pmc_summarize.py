"""Summarise rocprofv3 --pmc passes: mean counter value per launch for EVERY kernel of the run.
usage: python tools/pmc_summarize.py <dir with *counter_collection.csv> > summary.csv
(the FETCH_SIZE and WRITE_SIZE passes are separate rocprofv3 runs of the same command: MI355X_MICROARCH.md, HBM section)"""
import csv
import glob
import os
import sys
from collections import defaultdict

acc = defaultdict(lambda: [0.0, 0])
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if not k:
            continue
        k = k.split("(")[0].replace("void ", "").strip()
        a = acc[(k, r["Counter_Name"])]
        a[0] += float(r["Counter_Value"]); a[1] += 1
w = csv.writer(sys.stdout)
w.writerow(["kernel", "counter", "mean_per_launch", "launches"])
for (k, c), (v, n) in sorted(acc.items()):
    w.writerow([k, c, f"{v / n:.1f}", n])
