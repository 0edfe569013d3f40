#!/usr/bin/env python3
"""Print per-kernel averages from rocprofv3 kernel_stats.csv files under the given directories."""
import csv, glob, re, sys
for d in sys.argv[1:]:
    for f in glob.glob(f"{d}/*/*_kernel_stats.csv"):
        print(d)
        tot = 0
        rows = list(csv.DictReader(open(f)))
        for r in rows:
            n = r["Name"]; m = re.search(r"pass_kernel<(\d+), (\d+), (\d+)>", n)
            s = f"pass<{m.group(1)},{m.group(2)},{m.group(3)}>" if m else re.search(r"(\w+_kernel|__amd\w+)", n).group(1)
            print("  %-28s calls=%-4s avg_us=%9.1f total_ms=%8.2f pct=%s" % (s, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"][:5]))
