"""profiles/<pmc summary>.csv (tools/pmc_summarize.py over `rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE -- python bench.py`)
-> profiles/pmc_traffic.json: HBM/fabric bytes per launch per kernel, FETCH_SIZE doubled (gfx950 reports half of a
wide coalesced read, MI355X_MICROARCH.md "HBM"), both counters in KiB.
usage: python tools/pmc_to_traffic.py profiles/r01_pmc_bench_summary.csv profiles/pmc_traffic.json"""
import csv
import json
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
out = {}
for r in rows:
    k = r["kernel"].replace("void ", "")
    d = out.setdefault(k, {})
    d[r["counter"]] = float(r["mean_per_launch"])
    d["launches"] = int(r["launches"])
res = {}
for k, d in out.items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        res[k] = dict(bytes_per_launch=(2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0,
                      fetch_kib_raw=d["FETCH_SIZE"], write_kib=d["WRITE_SIZE"], launches=d["launches"])
# the swapped-order kernels come in two forms (last template argument: lean / general epilogue) that the library reports under ONE
# family name (et_conv2d_kernel_name): add the launch-weighted family entry bench.py looks up
fam = {}
for k, d in res.items():
    if "_swp_kernel<" not in k:
        continue
    head, _, last = k[:-1].rpartition(", ")
    name = (head + ">") if head else k[:k.index("<")]
    f = fam.setdefault(name, dict(b=0.0, n=0, forms=[]))
    f["b"] += d["bytes_per_launch"] * d["launches"]; f["n"] += d["launches"]; f["forms"].append(k)
for name, f in fam.items():
    res[name] = dict(bytes_per_launch=f["b"] / max(f["n"], 1), launches=f["n"], forms=f["forms"])
json.dump(dict(source=sys.argv[1], note="FETCH_SIZE doubled per the gfx950 calibration note", kernels=res),
          open(sys.argv[2], "w"), indent=1)
print(json.dumps(res, indent=1)[:600])
