#!/usr/bin/env python3
"""Instruction mix of a kernel's hottest loop, from the ISA `make` keeps (build/nbmf_hip.isa.s).

usage: tools/loop_mix.py <mangled-substring> [isa-file]
  e.g. tools/loop_mix.py pass_kernelILi4ELi1ELi0ELi0ELb0EE      (K = 64, DATA_F64, MODE_H)
The hottest loop is taken to be the backward branch whose body holds the most MFMAs; prints the counts per class for
that body and, divided by the tiles a trip covers (MFMAs / mfma-per-tile if given as 3rd argument), per tile."""
import re
import sys


def kernel_text(path, pat):
    on, out = False, []
    for line in open(path):
        if not on and re.match(r"^_Z\w*" + re.escape(pat) + r"\w*:", line):
            on = True
        if on:
            out.append(line.rstrip("\n"))
            if line.startswith(".Lfunc_end"):
                break
    return out


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_barrier") or op.startswith("s_setprio"):
        return "wait/misc"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def main():
    pat = sys.argv[1]
    path = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].isdigit() else "build/nbmf_hip.isa.s"
    per_tile = int(sys.argv[-1]) if sys.argv[-1].isdigit() else 0
    text = kernel_text(path, pat)
    if not text:
        sys.exit(f"no kernel matching {pat} in {path}")
    ins, labels = [], {}
    for line in text:
        t = line.strip()
        m = re.match(r"^(\.LBB\w+):", t)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        if not t or t.startswith((".", ";", "//")) or t.endswith(":"):
            continue
        ins.append(t.split(";")[0].strip())
    best = None
    for i, t in enumerate(ins):
        m = re.match(r"^s_cbranch\w*\s+(\.LBB\w+)|^s_branch\s+(\.LBB\w+)", t)
        if not m:
            continue
        tgt = labels.get(m.group(1) or m.group(2))
        if tgt is None or tgt > i:
            continue
        body = ins[tgt:i + 1]
        n = sum(1 for b in body if b.startswith("v_mfma"))
        if best is None or n > best[0]:
            best = (n, tgt, i)
    n, a, b = best
    counts = {}
    for t in ins[a:b + 1]:
        c = classify(t.split()[0])
        counts[c] = counts.get(c, 0) + 1
    print(f"{text[0]}  loop body: {b - a + 1} instructions, {n} MFMAs")
    tiles = n / per_tile if per_tile else 1
    for k in sorted(counts):
        print(f"  {k:10s} {counts[k]:5d}" + (f"   per tile {counts[k] / tiles:7.1f}" if per_tile else ""))
    # The loop as runs (round 5, tools/microbench4.hip: beside f64 MFMAs a 64-bit or VOP3 vector instruction costs ~4.1
    # cycles, v_rcp_f64 16.3, a plain 32-bit VOP2 2-3, the FIRST vector instruction behind an MFMA ~9.4 on top; scalar
    # instructions and a lone LDS read cost nothing): MFMAs, vector runs and the issue cycles that model gives a trip.
    seq, cyc, runs = [], 0.0, 0
    for t in ins[a:b + 1]:
        op = t.split()[0]
        c = classify(op)
        if c == "mfma":
            k, cost = "M", 64.6
        elif c == "valu":
            k = "v"
            cost = 16.3 if op.startswith(("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64")) else (
                2.5 if op in ("v_and_b32_e32", "v_or_b32_e32", "v_xor_b32_e32", "v_lshrrev_b32_e32", "v_lshlrev_b32_e32", "v_mov_b32_e32", "v_add_u32_e32", "v_sub_u32_e32") else 4.1)
            if not seq or seq[-1][0] != "v":
                cost += 9.4
                runs += 1
        else:
            k, cost = {"lds": "L", "vmem": "G", "smem": "S", "salu": "s", "branch": "b"}.get(c, "w"), 0.0
            if k in "sw":
                continue   # (free beside MFMAs, and they do not break a vector run)
        cyc += cost
        if seq and seq[-1][0] == k:
            seq[-1][1] += 1
        else:
            seq.append([k, 1])
    print("  runs: " + " ".join(f"{k}{n}" if n > 1 else k for k, n in seq))
    print(f"  vector runs per trip: {runs}; modelled issue cycles per trip: {cyc:.0f}" + (f" = {cyc / tiles:.0f} per tile" if per_tile else ""))
    scratch = sum(1 for t in ins if t.startswith("scratch_"))
    print(f"  scratch instructions in the whole kernel: {scratch}")


if __name__ == "__main__":
    main()
