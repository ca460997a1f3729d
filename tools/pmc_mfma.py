"""MFMA utilisation from hardware counters: `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES ... --kernel-trace --
python bench.py` -> profiles/pmc_mfma.json (bench.py puts `roofline.mfma_busy_frac` of the dominant kernel into its line from it).

    python tools/pmc_mfma.py <dir with *counter_collection.csv [+ *kernel_trace.csv]> <out.json> [launches_one_step.json]

Per kernel (mean per launch): the raw counters, the kernel's duration in that profiled run, and

  mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x kernel cycles)
      kernel cycles = GRBM_GUI_ACTIVE of the dispatch when collected (the chip clocks to its power budget: 1.9-2.0 GHz under dense
      MFMA, MI355X_MICROARCH.md "DVFS give-back"), else duration x the effective clock of the kernels that did collect it, else
      duration x 2.4 GHz (then `clock_source` says "max clock": a LOWER bound of the fraction).
      The counter counts matrix-pipe busy cycles summed over the SIMDs (= 32 x the number of v_mfma_f32_32x32x16_bf16 issued,
      MI355X_MICROARCH.md "Per-instruction cycle constants"): `mfma_cycles_expected` = algorithmic FLOPs / 32768 x 32 from the launch
      dump is printed beside it as the calibration of that unit (ratio ~1.0 + padding / zero-tile work).
  cu_busy_frac   = SQ_BUSY_CU_CYCLES / (256 CUs x kernel cycles)        (gfx94x MfmaUtil's denominator family; quad-cycle units are
      detected by the same ratio test: a value in (0.2, 0.3] x is rescaled by 4 and flagged)
  wait_inst_frac = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES, active_frac = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES   (both quad-cycles)
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

N_CU, N_SIMD, MAX_HZ = 256, 4, 2.4e9


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


def main():
    src, out_path = sys.argv[1], sys.argv[2]
    dump = json.load(open(sys.argv[3])) if len(sys.argv) > 3 else []
    dur = {}                                   # dispatch id -> ns (kernel trace of the SAME run)
    for f in glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            try:
                dur[r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
            except (KeyError, ValueError):
                pass
    per = defaultdict(lambda: dict(n=defaultdict(int), v=defaultdict(float), ns=0.0, ns_n=0, seen=set()))
    files = glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        sys.exit(f"no *counter_collection.csv under {src}")
    for f in files:
        for r in csv.DictReader(open(f)):
            k = short(r.get("Kernel_Name", ""))
            if not k:
                continue
            e = per[k]
            c = r["Counter_Name"]
            e["v"][c] += float(r["Counter_Value"]); e["n"][c] += 1
            did = r.get("Dispatch_Id")
            if did is not None and did not in e["seen"]:
                e["seen"].add(did)
                ns = None
                if "Start_Timestamp" in r and "End_Timestamp" in r:
                    try:
                        ns = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                    except ValueError:
                        ns = None
                if ns is None or ns <= 0:
                    ns = dur.get(did)
                if ns:
                    e["ns"] += ns; e["ns_n"] += 1
    alg = defaultdict(lambda: [0.0, 0])
    for r in dump:
        a = alg[r["kernel"]]
        a[0] += r["flops"]; a[1] += r["launches"]
    # effective clock from the kernels that have GRBM_GUI_ACTIVE: cycles per launch / seconds per launch; rocprofv3 may report the
    # counter summed over the 8 XCDs (or 32 SEs): pick the divisor that lands the clock inside the part's range
    def mean(e, c):
        return e["v"][c] / e["n"][c] if e["n"].get(c) else None
    gui_div, clocks = None, []
    for k, e in per.items():
        g = mean(e, "GRBM_GUI_ACTIVE")
        if g and e["ns_n"] and e["ns"] / e["ns_n"] > 20e3:       # kernels longer than 20 us only
            hz = g / (e["ns"] / e["ns_n"] * 1e-9)
            clocks.append(hz)
    if clocks:
        med = sorted(clocks)[len(clocks) // 2]
        for d in (1, 8, 32, 256):
            if 1.0e9 <= med / d <= 2.7e9:
                gui_div = d
                break
    eff_hz = (sorted(clocks)[len(clocks) // 2] / gui_div) if gui_div else None
    res = {}
    for k, e in per.items():
        o = dict(launches=max(e["n"].values()), avg_us=(e["ns"] / e["ns_n"] / 1e3) if e["ns_n"] else None)
        for c in e["v"]:
            o[c] = mean(e, c)
        cyc, src_c = None, None
        if gui_div and o.get("GRBM_GUI_ACTIVE"):
            cyc, src_c = o["GRBM_GUI_ACTIVE"] / gui_div, f"GRBM_GUI_ACTIVE / {gui_div}"
        elif o["avg_us"] and eff_hz:
            cyc, src_c = o["avg_us"] * 1e-6 * eff_hz, f"duration x {eff_hz / 1e9:.2f} GHz (median effective clock of this run)"
        elif o["avg_us"]:
            cyc, src_c = o["avg_us"] * 1e-6 * MAX_HZ, "duration x 2.4 GHz max clock (lower bound of the fractions)"
        o["kernel_cycles"], o["clock_source"] = cyc, src_c
        if cyc and o.get("SQ_VALU_MFMA_BUSY_CYCLES") is not None:
            o["mfma_busy_frac"] = o["SQ_VALU_MFMA_BUSY_CYCLES"] / (N_SIMD * N_CU * cyc)
        if cyc and o.get("SQ_BUSY_CU_CYCLES") is not None:
            f_ = o["SQ_BUSY_CU_CYCLES"] / (N_CU * cyc)
            if 0.2 < f_ <= 0.3:                                 # quad-cycle units
                f_, o["cu_busy_unit"] = f_ * 4, "quad-cycles (x4)"
            o["cu_busy_frac"] = f_
            if o.get("mfma_busy_frac") is not None and f_ > 0:
                # matrix-pipe busy share of the time a CU is busy with this kernel: independent of the clock estimate and of the grid's
                # tail / ramp (what the main loop itself achieves; the in-loop s_memtime stamps of NOTEBOOK.md read 62-64 % for pprs)
                o["mfma_busy_of_cu_busy"] = o["mfma_busy_frac"] / f_
        if o.get("SQ_WAVE_CYCLES"):
            for c, n in (("SQ_WAIT_INST_ANY", "wait_inst_frac"), ("SQ_ACTIVE_INST_ANY", "active_frac"), ("SQ_WAIT_ANY", "wait_any_frac")):
                if o.get(c) is not None:
                    o[n] = o[c] / o["SQ_WAVE_CYCLES"]
        a = alg.get(k)
        if a and a[1] and a[0] > 0:
            o["algorithmic_gflop_per_launch"] = a[0] / a[1] / 1e9
            o["mfma_cycles_expected"] = a[0] / a[1] / 32768.0 * 32.0
            if o.get("SQ_VALU_MFMA_BUSY_CYCLES"):
                o["mfma_cycles_counted_over_expected"] = o["SQ_VALU_MFMA_BUSY_CYCLES"] / o["mfma_cycles_expected"]
        res[k] = o
    top = sorted((k for k in res if res[k]["avg_us"]), key=lambda k: -(res[k]["avg_us"] * res[k]["launches"]))
    json.dump(dict(source=src, effective_clock_ghz=eff_hz / 1e9 if eff_hz else None, gui_active_divisor=gui_div,
                   note="mean per launch; mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles); see tools/pmc_mfma.py",
                   top_by_time=top[:12], kernels=res), open(out_path, "w"), indent=1)
    print(f"effective clock {eff_hz / 1e9 if eff_hz else float('nan'):.2f} GHz (GRBM_GUI_ACTIVE / {gui_div})")
    print(f"{'kernel':58s} {'launches':>8s} {'avg us':>8s} {'mfma_busy':>9s} {'cu_busy':>8s} {'wait_inst':>9s} {'cnt/exp':>8s}")
    for k in top[:14]:
        o = res[k]
        f = lambda v: f"{v:8.3f}" if v is not None else "       -"
        print(f"{k[:58]:58s} {o['launches']:8d} {o['avg_us']:8.1f} {f(o.get('mfma_busy_frac'))} {f(o.get('cu_busy_frac'))} {f(o.get('wait_inst_frac'))} {f(o.get('mfma_cycles_counted_over_expected'))}")


if __name__ == "__main__":
    main()
