#!/usr/bin/env python3
"""Compare two ISA files of the library kernel by kernel (tools/isa_diff.py old.s new.s): a refactoring that is meant
to change no code must leave every kernel's instruction stream identical.  Symbols that differ per build (the
__hip_cuid_* symbol), comment lines and directive-only differences are ignored; printed: kernels whose bodies differ,
kernels only in one file, and per differing kernel the size of the diff."""
import difflib
import re
import sys


def kernels(path):
    out, name, body = {}, None, []
    for line in open(path):
        m = re.match(r"^(_Z\w+|\w+):\s*(;.*)?$", line)
        if m and not line.startswith(".L"):
            if name:
                out[name] = body
            name, body = m.group(1), []
            continue
        if name is None:
            continue
        t = line.split(";")[0].rstrip()
        if not t.strip() or t.strip().startswith(".") and not t.strip().startswith(".amdhsa"):
            if t.strip().startswith(".Lfunc_end"):
                out[name] = body
                name = None
            continue
        if "__hip_cuid" in t:
            continue
        body.append(re.sub(r"\.LBB\d+_", ".LBB_", t.strip()))   # (block labels carry the function's ordinal: not a difference)
    if name:
        out[name] = body
    return out


def canonical(name):
    """pass_kernel gained a sixth template parameter in round 5 (RAG); its RAG = false instantiations are the old kernels."""
    return re.sub(r"(pass_kernelILi\dELi\dELi\dELi\dELb[01])ELb0(EEEv)", r"\1\2", name)


def main(a, b):
    ka, kb = {canonical(k): v for k, v in kernels(a).items()}, {canonical(k): v for k, v in kernels(b).items()}
    only_a, only_b = sorted(set(ka) - set(kb)), sorted(set(kb) - set(ka))
    differ = []
    for k in sorted(set(ka) & set(kb)):
        if ka[k] != kb[k]:
            d = [x for x in difflib.unified_diff(ka[k], kb[k], lineterm="", n=0) if x[:1] in "+-" and x[:3] not in ("+++", "---")]
            differ.append((k, len(d), len(ka[k]), len(kb[k])))
    print(f"{len(ka)} / {len(kb)} functions; identical: {len(set(ka) & set(kb)) - len(differ)}; differing: {len(differ)}; only in old: {len(only_a)}; only in new: {len(only_b)}")
    for k, n, la, lb in differ:
        print(f"  DIFFERS {k}: {n} changed lines ({la} -> {lb} instructions)")
    for k in only_a:
        print(f"  only in old: {k}")
    for k in only_b:
        print(f"  only in new: {k}")
    return 1 if differ or only_a or only_b else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1], sys.argv[2]))
