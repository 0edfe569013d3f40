#!/usr/bin/env python3
"""gfx950: an MFMA must not read a VGPR within two wait states of a vector instruction writing it.  hipcc inserts the
wait states for instructions it knows, but not behind INLINE ASSEMBLY (the hand-written selects of nbmf_hip.hip).  This
walks build/nbmf_hip.s (`make -C nbmf_mm_amd/csrc asm`) and reports every MFMA whose A/B/C operand was written by an
inline-assembly instruction fewer than NEED wait states earlier; with --sgpr also every vector instruction that reads an
SGPR fewer than NEED wait states behind an inline-assembly compare that wrote it.  Exit status 1 if any."""
import re, sys
NEED = 2
args = [a for a in sys.argv[1:] if not a.startswith("--")]
path = args[0] if args else "build/nbmf_hip.s"
reg = re.compile(r"v\[(\d+):(\d+)\]|v(\d+)")
def regs(tok):
    out = set()
    for m in reg.finditer(tok):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out
sreg = re.compile(r"s\[(\d+):(\d+)\]|\bs(\d+)\b")
def sregs(tok):
    out = set()
    for m in sreg.finditer(tok):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out
bad = total = sbad = 0
kernel, in_asm, clock, last_asm_write, last_asm_swrite = None, False, 0, {}, {}
for line in open(path):
    s = line.strip()
    m = re.match(r"^(_Z\w+):", s)
    if m:
        kernel, clock, last_asm_write, last_asm_swrite = m.group(1), 0, {}, {}
        continue
    if s.startswith(";;#ASMSTART"):
        in_asm = True; continue
    if s.startswith(";;#ASMEND"):
        in_asm = False; continue
    if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
        continue
    op, _, rest = s.partition(" ")
    if op.endswith(":"):
        continue
    if op == "s_nop":
        clock += int(rest.strip()) + 1
        continue
    if op.startswith("v_mfma"):
        total += 1
        ops = [t.strip() for t in rest.split(",")]
        used = set().union(*[regs(t) for t in ops[1:4]])
        for r in used:
            if r in last_asm_write and clock - last_asm_write[r] - 1 < NEED:
                bad += 1
                print(f"{kernel}: {s}   <- v{r} written by inline asm {clock - last_asm_write[r] - 1} wait state(s) earlier")
                break
    if op.startswith("v_") and "--sgpr" in sys.argv:
        # (LLVM's table for gfx940: a vector instruction reading an SGPR needs two wait states behind a vector write of it)
        srcs = rest.split(",")[1:] if not op.startswith("v_cmp") else rest.split(",")[1:]
        for r in set().union(*[sregs(t) for t in srcs]) if srcs else ():
            if r in last_asm_swrite and clock - last_asm_swrite[r] - 1 < NEED:
                sbad += 1
                print(f"{kernel}: {s}   <- s{r} written by inline asm {clock - last_asm_swrite[r] - 1} wait state(s) earlier")
                break
    if in_asm and op.startswith("v_cmp"):
        for r in sregs(rest.split(",")[0]):
            last_asm_swrite[r] = clock
    if in_asm and op.startswith("v_") and not op.startswith("v_cmp"):
        dst = rest.split(",")[0]
        for r in regs(dst):
            last_asm_write[r] = clock
    elif not in_asm:
        # a known instruction overwriting the register: the compiler handles its hazards itself
        if op.startswith(("v_", "ds_read", "global_load", "buffer_load")):
            for r in regs(rest.split(",")[0]):
                last_asm_write.pop(r, None)
    if op.startswith(("s_cbranch", "s_branch", "s_barrier", "s_endpgm", "s_setpc")):
        pass   # (branches: the conservative view keeps the last writes; a taken branch only adds wait states)
    clock += 1
print(f"{total} MFMA instructions checked, {bad} too close to an inline-assembly write" + (f"; {sbad} vector reads of an SGPR too close to an inline-assembly compare" if "--sgpr" in sys.argv else ""))
sys.exit(1 if bad or sbad else 0)
