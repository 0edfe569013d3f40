#!/usr/bin/env python3
"""Print the key numbers of the JSON line found in each given bench log."""
import json, sys
for f in sys.argv[1:]:
    for line in open(f):
        if line.startswith("{"):
            d = json.loads(line); r = d["roofline"]
            print("%-50s it/s %8.1f ms/step %7.3f hpass %.3f wpass %.3f frac %.3f iterfrac %.3f nll %.12f" % (
                f.split("/")[-1], d["value"], d["ms_per_step"], r["hpass_ms"], r["wpass_ms"], r["frac"], r["iteration_frac"], d["final_nll_per_entry"]))
