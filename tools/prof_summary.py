#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (gpurun_out/prof_*/{stats,pmc_*}) into the text summary that is
committed under profiles/.  Usage: tools/prof_summary.py gpurun_out/prof_r1 > profiles/r1_xxx.txt"""
import collections
import csv
import glob
import re
import sys


def short(name):
    m = re.search(r"pass_kernel<(\d+), (\d+), (\d+)(?:, (\d+))?(?:, (true|false))?(?:, (true|false))?(?:, (true|false))?>", name)
    if m:
        kb, data, mode = map(int, m.groups()[:3])
        th = ",Theta-in" if m.group(4) and int(m.group(4)) else ""
        tiny = ",tiny-eps" if m.group(5) == "true" else ""
        rag = ",ragged-K" if m.group(6) == "true" else ""
        full = ",two-state" if m.group(7) == "true" else ""
        return f"pass_kernel<K={16*kb},{['BIN','F64','F64M'][data]},{'HWLT'[mode]}{th}{tiny}{rag}{full}>"
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
    """Per launch of every pass kernel, from the PMC passes (each pass is its own run, so each quantity uses the
    duration of the run it was collected in): clock, MFMA-busy share of all SIMD cycles, what a 16x16 tile costs
    (SIMD cycles, of which MFMA; vector instructions besides the MFMAs; scalar / LDS / branch instructions), where the
    waves' time goes (parked at a wait or barrier / waiting to issue / issuing) and the HBM traffic."""
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
        get = lambda c: v[c][0] if c in v else None   # noqa: E731
        out = [k]
        m = re.search(r"K=(\d+),\w+,([HWLT])", k)
        per_tile = None
        if m:
            K, mode = int(m.group(1)), m.group(2)
            per_tile = {"H": 3 * K // 4, "W": K // 2, "L": K // 4, "T": K // 4}[mode]
        if "GRBM_GUI_ACTIVE" in v:
            cyc, t = v["GRBM_GUI_ACTIVE"][0] / 8.0, v["GRBM_GUI_ACTIVE"][1]
            out.append("clock %.2f GHz" % (cyc / t / 1e9))
            if "SQ_VALU_MFMA_BUSY_CYCLES" in v:
                out.append("MFMA busy %.1f %% of SIMD cycles (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs))"
                           % (100.0 * v["SQ_VALU_MFMA_BUSY_CYCLES"][0] / (cyc * 1024)))
            if per_tile and get("SQ_INSTS_MFMA"):
                tiles = get("SQ_INSTS_MFMA") / per_tile
                line = "per 16x16 tile: %.0f SIMD cycles, %d of them MFMA (%d x 64)" % (cyc * 1024 / tiles, 64 * per_tile, per_tile)
                if get("SQ_INSTS_VALU"):
                    line += ", %.1f vector instructions besides the MFMAs" % ((get("SQ_INSTS_VALU") - get("SQ_INSTS_MFMA")) / tiles)
                for cname, label in (("SQ_INSTS_SALU", "scalar"), ("SQ_INSTS_LDS", "LDS"), ("SQ_INSTS_VMEM", "VMEM"), ("SQ_INSTS_BRANCH", "branch")):
                    if get(cname):
                        line += ", %.1f %s" % (get(cname) / tiles, label)
                out.append(line)
        if get("SQ_WAVE_CYCLES"):
            wc = get("SQ_WAVE_CYCLES")
            parts = []
            for cname, label in (("SQ_WAIT_ANY", "parked at a wait / barrier"), ("SQ_WAIT_INST_ANY", "waiting to issue")):
                if get(cname):
                    parts.append("%.1f %% %s" % (100.0 * get(cname) / wc, label))
            if get("SQ_ACTIVE_INST_ANY"):
                parts.append("%.1f %% issuing" % (100.0 * get("SQ_ACTIVE_INST_ANY") / wc))
            if parts:
                out.append("wave time: " + ", ".join(parts) + " (of SQ_WAVE_CYCLES)")
        if get("SQ_LDS_BANK_CONFLICT") is not None and get("SQ_LDS_IDX_ACTIVE"):
            out.append("LDS bank conflicts %.2f %% of LDS cycles" % (100.0 * get("SQ_LDS_BANK_CONFLICT") / get("SQ_LDS_IDX_ACTIVE")))
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            b = (2.0 * v["FETCH_SIZE"][0] + v["WRITE_SIZE"][0]) * 1024.0
            out.append("HBM traffic %.3f GB = %.2f TB/s ((2 x FETCH_SIZE + WRITE_SIZE) KiB over the fetch run's duration)"
                       % (b / 1e9, b / v["FETCH_SIZE"][1] / 1e12))
        print("  " + ";\n      ".join(out))


def sidecar(root, out_json):
    """Machine-readable extract for bench.py's roofline.traffic: per-launch means of the dominant kernel."""
    import json
    rec = {}
    for d in sorted(glob.glob(f"{root}/pmc_*")):
        for f in glob.glob(f"{d}/*/*_counter_collection.csv"):
            acc = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                if k.startswith("pass_kernel") and ",H" in k.split("<", 1)[1]:
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
