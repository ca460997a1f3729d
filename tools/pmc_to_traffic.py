"""profiles/<pmc summary>.csv (tools/pmc_summarize.py over `rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE -- python bench.py`)
-> profiles/pmc_traffic.json: HBM/fabric bytes per launch for EVERY kernel of the step, the per-step total by kernel family, and --
for the conv kernels -- the algorithmic bytes per launch beside the measured ones (from a `bench.py --dump-launches` file of the
same build), so that the wasted-traffic ratio can be read off per kernel.
FETCH_SIZE is doubled (gfx950 reports half of a wide coalesced read, MI355X_MICROARCH.md "HBM"); both counters are in KiB.
usage: python tools/pmc_to_traffic.py <summary.csv> <out.json> [launches_one_step.json]
The number of steps the profiled command ran is taken from the launch count of a once-per-step kernel (nms_greedy_kernel for the
SSOD workloads, sgd_kernel / 3 otherwise)."""
import csv
import json
import sys

FAMILY = (("conv1x1_stream", "conv_1x1_stream"), ("conv_wgrad", "conv_wgrad"), ("conv_stem", "conv_gather_gemm"), ("conv_gemm", "conv_gather_gemm"),
          ("bn_act_fwd", "bn_fwd"), ("bn_act_bwd", "bn_bwd"), ("rows_reduce_finalize", "bn_finalize"), ("act_bwd", "bn_bwd"),
          ("nms_", "nms_loss_pl"), ("loss_", "nms_loss_pl"), ("select_targets", "nms_loss_pl"), ("pseudo_label", "nms_loss_pl"),
          ("detect_decode", "nms_loss_pl"), ("ema_", "optimizer_ema"), ("sgd_", "optimizer_ema"), ("weight_transpose", "optimizer_ema"),
          ("cast_", "optimizer_ema"), ("maxpool", "pool_upsample_pack"), ("upsample", "pool_upsample_pack"), ("pack_input", "pool_upsample_pack"))


def family(k):
    for pat, fam in FAMILY:
        if pat in k:
            return fam
    return "torch_and_other"


rows = list(csv.DictReader(open(sys.argv[1])))
out = {}
for r in rows:
    d = out.setdefault(r["kernel"], {})
    d[r["counter"]] = float(r["mean_per_launch"])
    d["launches"] = int(r["launches"])
steps = None
for k, d in out.items():
    if k.startswith("nms_greedy_kernel"):
        steps = d["launches"]
if steps is None:
    steps = max(1, sum(d["launches"] for k, d in out.items() if k.startswith("sgd_kernel")) // 3)
alg = {}
if len(sys.argv) > 3:
    for r in json.load(open(sys.argv[3])):
        a = alg.setdefault(r["kernel"], [0.0, 0])
        a[0] += r["bytes"]; a[1] += r["launches"]
res, fam = {}, {}
for k, d in out.items():
    if "FETCH_SIZE" not in d or "WRITE_SIZE" not in d:
        continue
    b = (2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0
    e = dict(bytes_per_launch=b, fetch_kib_raw=d["FETCH_SIZE"], write_kib=d["WRITE_SIZE"], launches=d["launches"],
             launches_per_step=d["launches"] / steps, family=family(k))
    a = alg.get(k)
    if a and a[1] and a[0] > 0:
        e["algorithmic_bytes_per_launch"] = a[0] / a[1]
        e["traffic_over_algorithmic"] = b / (a[0] / a[1])
    res[k] = e
    f = fam.setdefault(e["family"], 0.0)
    fam[e["family"]] = f + b * d["launches"] / steps
total = sum(fam.values())
json.dump(dict(source=sys.argv[1], note="FETCH_SIZE doubled per the gfx950 calibration note; bytes = (2 * FETCH_SIZE + WRITE_SIZE) KiB",
               steps_profiled=steps, step=dict(bytes_per_step=total, by_family={k: v for k, v in sorted(fam.items(), key=lambda kv: -kv[1])}),
               kernels=res), open(sys.argv[2], "w"), indent=1)
print(f"{steps} steps; {total / 1e9:.1f} GB per step:", {k: round(v / 1e9, 2) for k, v in sorted(fam.items(), key=lambda kv: -kv[1])})
