#!/usr/bin/env python3
"""The two speeds of the single-launch path on configs[0]: many launches of the same fit, each timed on
the host and instrumented on the device (NBMF_SMALL_DEBUG=1: workgroup 0's per-phase wall clock over iterations 8..62
and the shader clock it ran at).  Prints every launch's us/iteration, its phase breakdown and clock, then the mean
breakdown of the fast and of the slow launches side by side.  usage: bimodal_probe.py [launches] [iterations] [idle_ms]"""
import os, re, subprocess, sys, tempfile, time
HERE = os.path.dirname(os.path.abspath(__file__))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, os.path.dirname(HERE))
    import numpy as np
    from nbmf_mm_amd import _hip, _dist
    launches, iters, idle_ms = int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
    m, n, k = 100, 500, 6
    X = (np.random.default_rng(0).random((m, n)) < 0.25).astype(np.float64)
    W, H = _dist.global_init(m, n, k, random_state=0)
    with _hip.Context(m, n, k) as ctx:
        ctx.set_hyper(1.2, 1.2)
        ctx.upload(X)
        for i in range(launches):
            ctx.set_factors(W, H)
            if idle_ms > 0:
                time.sleep(idle_ms * 1e-3)
            t0 = time.perf_counter()
            losses, nit = ctx.run(iters, 0.0)
            dt = time.perf_counter() - t0
            sys.stderr.write("[probe] launch %d: %.3f us/iteration at t=%.3f s\n" % (i, 1e6 * dt / nit, t0))
            sys.stderr.flush()
    sys.exit(0)
launches = int(sys.argv[1]) if len(sys.argv) > 1 else 60
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
idle_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
env = dict(os.environ, NBMF_SMALL_DEBUG="1")
p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(launches), str(iters), str(idle_ms)], env=env,
                   capture_output=True, text=True)
recs, cur = [], {}
for line in p.stderr.splitlines():
    m = re.search(r"from the first instruction to the last: ([0-9.]+) ms", line)
    if m:
        cur["dev_ms"] = float(m.group(1))
    m = re.search(r"shader clock over iterations 8..62: (\d+) MHz", line)
    if m:
        cur["mhz"] = float(m.group(1))
    m = re.search(r"us per iteration part \(mean of \d+\): (.*)", line)
    if m:
        cur["parts"] = [(a.strip(), float(b)) for a, b in re.findall(r"([A-Za-z+ ]+?) ([0-9.]+)(?: \||$)", m.group(1))]
    m = re.search(r"\[probe\] launch (\d+): ([0-9.]+) us/iteration at t=([0-9.]+) s", line)
    if m:
        cur["us"] = float(m.group(2))
        cur["t"] = float(m.group(3))
        recs.append(cur)
        cur = {}
if not recs:
    print(p.stderr[-3000:])
    sys.exit(1)
for i, r in enumerate(recs):
    print("launch %2d  host %6.2f us/it = %7.3f ms | device %7.3f ms | %5.0f MHz  %s" % (
        i, r["us"], r["us"] * iters * 1e-3, r.get("dev_ms", 0), r.get("mhz", 0), "  ".join("%s %.2f" % ab for ab in r.get("parts", []))))
us = sorted(r["us"] for r in recs)
cut = 0.5 * (us[0] + us[-1])
fast, slow = [r for r in recs if r["us"] <= cut], [r for r in recs if r["us"] > cut]
print("\nfast: %d launches, mean %.2f us/it;  slow (> %.2f): %d launches, mean %.2f us/it" % (
    len(fast), sum(r["us"] for r in fast) / max(1, len(fast)), cut, len(slow), sum(r["us"] for r in slow) / max(1, len(slow))))
if slow:
    print("slow launches started at t = " + ", ".join("%.2f s" % (r["t"] - recs[0]["t"]) for r in slow) + " (after the first launch)")
for name, grp in (("fast", fast), ("slow", slow)):
    grp = [r for r in grp if "parts" in r]
    if not grp:
        continue
    keys = [a for a, _ in grp[0]["parts"]]
    means = [sum(dict(r["parts"])[k] for r in grp) / len(grp) for k in keys]
    print("%s: device %.3f ms of host %.3f ms | %5.0f MHz | " % (name, sum(r.get("dev_ms", 0) for r in grp) / len(grp),
          sum(r["us"] for r in grp) / len(grp) * iters * 1e-3, sum(r.get("mhz", 0) for r in grp) / len(grp)) + " | ".join("%s %.2f" % (k, v) for k, v in zip(keys, means)))
