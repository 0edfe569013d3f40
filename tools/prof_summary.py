#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (gpurun_out/prof_*/{stats,pmc_*}) into the text summary that is
committed under profiles/.  Usage: tools/prof_summary.py gpurun_out/prof_r1 > profiles/r1_xxx.txt"""
import collections
import csv
import glob
import re
import sys


def short(name):
    m = re.search(r"pass_kernel<(\d+), (\d+), (\d+)(?:, (\d+))?>", name)
    if m:
        kb, data, mode = map(int, m.groups()[:3])
        th = ",Theta-in" if m.group(4) and int(m.group(4)) else ""
        return f"pass_kernel<K={16*kb},{['BIN','F64','F64M'][data]},{'HWLT'[mode]}{th}>"
    m = re.search(r"(\w+_kernel|__amd_rocclr_\w+)", name)
    return m.group(1) if m else name[:40]


def main(root):
    print(f"# rocprofv3 summary of {root}")
    for f in glob.glob(f"{root}/stats/*/*_kernel_stats.csv"):
        print("\n## --kernel-trace --stats (per kernel)\n")
        print("%-40s %6s %14s %12s %7s %12s %12s" % ("kernel", "calls", "total_ns", "avg_ns", "pct", "min_ns", "max_ns"))
        for r in csv.DictReader(open(f)):
            print("%-40s %6s %14s %12.0f %7s %12s %12s" % (short(r["Name"]), r["Calls"], r["TotalDurationNs"],
                                                          float(r["AverageNs"]), r["Percentage"][:6], r["MinNs"], r["MaxNs"]))
    for d in sorted(glob.glob(f"{root}/pmc_*")):
        for f in glob.glob(f"{d}/*/*_counter_collection.csv"):
            agg = collections.defaultdict(lambda: collections.defaultdict(list))
            dur = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            print(f"\n## --pmc pass {d.split('/')[-1]} (mean per dispatch)\n")
            for k in sorted(agg):
                if not k.startswith("pass_kernel") and "update" not in k and "reduce" not in k:
                    continue
                parts = ["%s=%.6g" % (c, sum(v) / len(v)) for c, v in sorted(agg[k].items())]
                print("%-40s n=%-3d dur_ns=%-10.0f %s" % (k, len(dur[k]) // max(1, len(agg[k])), sum(dur[k]) / len(dur[k]), " ".join(parts)))


def derived(root):
    """MFMA utilisation, clock and HBM rate of the pass kernels from the PMC passes (each pass is its own
    run, so each quantity uses the duration of the run it was collected in)."""
    vals = collections.defaultdict(dict)
    for d in sorted(glob.glob(f"{root}/pmc_*")):
        for f in glob.glob(f"{d}/*/*_counter_collection.csv"):
            acc, dur = collections.defaultdict(lambda: collections.defaultdict(list)), collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                if not k.startswith("pass_kernel"):
                    continue
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            for k in acc:
                t = sum(dur[k]) / len(dur[k]) * 1e-9
                for cname, v in acc[k].items():
                    vals[k][cname] = (sum(v) / len(v), t)
    print("\n## derived (per launch)\n")
    for k in sorted(vals):
        v = vals[k]
        out = [k]
        if "GRBM_GUI_ACTIVE" in v:
            cyc, t = v["GRBM_GUI_ACTIVE"][0] / 8.0, v["GRBM_GUI_ACTIVE"][1]
            out.append("clock %.2f GHz" % (cyc / t / 1e9))
            if "SQ_VALU_MFMA_BUSY_CYCLES" in v:
                out.append("MFMA busy %.1f %% of SIMD cycles (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs))"
                           % (100.0 * v["SQ_VALU_MFMA_BUSY_CYCLES"][0] / (cyc * 1024)))
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            b = (2.0 * v["FETCH_SIZE"][0] + v["WRITE_SIZE"][0]) * 1024.0
            out.append("HBM traffic %.3f GB = %.2f TB/s ((2 x FETCH_SIZE + WRITE_SIZE) KiB over the fetch run's duration)"
                       % (b / 1e9, b / v["FETCH_SIZE"][1] / 1e12))
        print("  " + "; ".join(out))


def sidecar(root, out_json):
    """Machine-readable extract for bench.py's roofline.traffic: per-launch means of the dominant kernel."""
    import json
    rec = {}
    for d in sorted(glob.glob(f"{root}/pmc_*")):
        for f in glob.glob(f"{d}/*/*_counter_collection.csv"):
            acc = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                if k.startswith("pass_kernel") and k.endswith(",H>"):
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
                    rec["kernel"] = k
            for c, v in acc.items():
                rec[c] = sum(v) / len(v)
    for f in glob.glob(f"{root}/stats/*/*_kernel_stats.csv"):
        for r in csv.DictReader(open(f)):
            if short(r["Name"]) == rec.get("kernel"):
                rec["avg_ns"] = float(r["AverageNs"])
    json.dump(rec, open(out_json, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main(sys.argv[1].rstrip("/"))
    derived(sys.argv[1].rstrip("/"))
    if len(sys.argv) > 2:
        sidecar(sys.argv[1].rstrip("/"), sys.argv[2])
