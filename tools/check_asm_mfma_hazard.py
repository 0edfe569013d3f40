#!/usr/bin/env python3
"""Static hazard check of the compiled kernels (build/nbmf_hip.s, `make -C nbmf_mm_amd/csrc asm`).

gfx950 needs wait states that hipcc inserts only behind instructions it knows -- not behind INLINE ASSEMBLY:
  (1) an MFMA must not read a VGPR fewer than NEED (2) wait states behind a vector instruction that wrote it;
  (2) --sgpr: a vector instruction must not read an SGPR fewer than NEED wait states behind a vector compare that
      wrote it (LLVM's hazard table for gfx940, VALUWriteSGPRVALURead).
Every inline-assembly vector write is tracked through the kernel's CONTROL-FLOW GRAPH (basic blocks, branch
targets, fall-through, loop back-edges: a write at the bottom of a loop body is still fresh at the top of the next
trip), by a forward data-flow pass whose merge keeps the smallest distance over all predecessors.  A wait state is one
issued instruction, or N+1 for `s_nop N`.  A register overwritten by an instruction the compiler knows is dropped (its
hazards are the compiler's business).  Exit status 1 if anything is too close.

Part of the build: `make` runs this on every library it produces and removes the library on failure."""
import re
import sys

NEED = 2
_VREG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")
_SREG = re.compile(r"\bs\[(\d+):(\d+)\]|\bs(\d+)\b")
_AREG = re.compile(r"\ba\[(\d+):(\d+)\]|\ba(\d+)\b")


def _regs(rx, tok, tag):
    out = set()
    for m in rx.finditer(tok):
        if m.group(1):
            out.update((tag, r) for r in range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add((tag, int(m.group(3))))
    return out


def vregs(tok):
    return _regs(_VREG, tok, "v") | _regs(_AREG, tok, "a")


def sregs(tok):
    out = _regs(_SREG, tok, "s")
    if re.search(r"\bvcc\b", tok):
        out |= {("s", "vcc")}
    return out


class Inst:
    __slots__ = ("op", "rest", "asm", "text", "line")

    def __init__(self, op, rest, asm, text, line):
        self.op, self.rest, self.asm, self.text, self.line = op, rest, asm, text, line


def parse(path):
    """-> {kernel: (blocks, labels)}; blocks = list of lists of Inst, labels = {name: block index}."""
    kernels = {}
    kernel, blocks, labels, cur, in_asm = None, None, None, None, False

    def start_block():
        nonlocal cur
        if cur is None or cur:
            cur = []
            blocks.append(cur)

    for ln, line in enumerate(open(path), 1):
        s = line.split("//")[0].strip()
        m = re.match(r"^(_Z\w+|[A-Za-z_]\w*):\s*(;.*)?$", s)
        if m and not s.startswith(".L") and (m.group(1).startswith("_Z") or kernel is None):
            kernel, blocks, labels, cur, in_asm = m.group(1), [], {}, None, False
            kernels[kernel] = (blocks, labels)
            start_block()
            continue
        if kernel is None:
            continue
        if s.startswith(".Lfunc_end"):
            kernel = None
            continue
        m = re.match(r"^(\.L\w+):", s)
        if m:
            start_block()
            labels[m.group(1)] = len(blocks) - 1
            continue
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        s = s.split(";")[0].strip()
        if not s or s.startswith("."):
            continue
        op, _, rest = s.partition(" ")
        cur.append(Inst(op, rest.strip(), in_asm, s, ln))
        if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc")):
            start_block()
    return kernels


def successors(blocks, labels, i):
    b = blocks[i]
    nxt = [i + 1] if i + 1 < len(blocks) else []
    if not b:
        return nxt
    last = b[-1]
    if last.op.startswith(("s_endpgm", "s_setpc", "s_swappc")):
        return []
    if last.op.startswith(("s_branch", "s_cbranch")):
        tgt = labels.get(last.rest.split(",")[-1].strip())
        out = [] if tgt is None else [tgt]
        return out if last.op.startswith("s_branch") else out + nxt
    return nxt


def dests(inst):
    """(vector registers, scalar registers) written: the first operand -- two for the SDWA/VOP3 compares that name
    an SGPR destination; plain v_cmp writes vcc."""
    first = inst.rest.split(",")[0]
    if inst.op.startswith("v_cmp"):
        s = sregs(first) if ("_e64" in inst.op or "_sdwa" in inst.op) else {("s", "vcc")}
        return set(), s
    if inst.op.startswith(("v_readlane", "v_readfirstlane")):
        return set(), sregs(first)
    if inst.op.startswith(("v_", "ds_read", "global_load", "buffer_load", "flat_load", "scratch_load")) and "_lds_" not in inst.op:
        return vregs(first), set()
    return set(), set()


def sources(inst):
    ops = [t.strip() for t in inst.rest.split(",")]
    return ops[1:] if len(ops) > 1 else []


def transfer(state, inst, check_sgpr, report):
    """state: {reg: wait states since the inline-assembly write}; returns the state behind `inst`."""
    if inst.op.startswith("v_mfma"):
        used = set().union(*[vregs(t) for t in sources(inst)[:3]]) if sources(inst) else set()
        for r in sorted(used, key=str):
            if r in state:
                report("mfma", inst, r, state[r])
                break
    if check_sgpr and inst.op.startswith("v_"):
        used = set().union(*[sregs(t) for t in sources(inst)]) if sources(inst) else set()
        for r in sorted(used, key=str):
            if r in state:
                report("sgpr", inst, r, state[r])
                break
    cost = int(inst.rest.split()[0], 0) + 1 if inst.op == "s_nop" else 1
    out = {r: d + cost for r, d in state.items() if d + cost < NEED}
    vw, sw = dests(inst)
    for r in vw | sw:
        out.pop(r, None)                     # overwritten: whoever wrote it before no longer matters
    if inst.asm:
        for r in vw:
            out[r] = 0
        if check_sgpr:
            for r in sw:
                out[r] = 0
    return out


def check(path, check_sgpr=False, verbose=True):
    n_mfma = 0
    found = {}
    for kernel, (blocks, labels) in parse(path).items():
        n_mfma += sum(1 for b in blocks for i in b if i.op.startswith("v_mfma"))
        succ = [successors(blocks, labels, i) for i in range(len(blocks))]
        entry = [None] * len(blocks)         # None = not reached yet
        entry[0] = {}
        work = [0]

        def report(kind, inst, reg, dist, kernel=kernel):
            found.setdefault((kernel, inst.line, kind), (inst, reg, dist))

        while work:
            i = work.pop()
            st = dict(entry[i])
            for inst in blocks[i]:
                st = transfer(st, inst, check_sgpr, report)
            for j in succ[i]:
                if entry[j] is None:
                    entry[j] = dict(st)
                    work.append(j)
                else:
                    merged = dict(entry[j])
                    for r, d in st.items():
                        if r not in merged or d < merged[r]:
                            merged[r] = d
                    if merged != entry[j]:
                        entry[j] = merged
                        work.append(j)
    bad = sum(1 for k in found if k[2] == "mfma")
    sbad = sum(1 for k in found if k[2] == "sgpr")
    if verbose:
        for (kernel, line, kind), (inst, reg, dist) in sorted(found.items(), key=lambda kv: (kv[0][0], kv[0][1])):
            print(f"{kernel}: line {line}: {inst.text}   <- {reg[0]}{reg[1]} written by inline asm {dist} wait state(s) earlier")
        print(f"{n_mfma} MFMA instructions checked, {bad} too close to an inline-assembly write"
              + (f"; {sbad} vector reads of an SGPR too close to an inline-assembly compare" if check_sgpr else ""))
    return n_mfma, bad, sbad


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    _, bad, sbad = check(args[0] if args else "build/nbmf_hip.s", "--sgpr" in sys.argv)
    sys.exit(1 if bad or sbad else 0)
