"""Static check of the compiled kernels: wait states hipcc cannot insert behind inline assembly.

gfx950 needs two wait states between a vector instruction writing a VGPR and an MFMA reading it (and between a vector
compare writing an SGPR and a vector instruction reading it); hipcc inserts them for instructions it knows, not behind
inline assembly.  A violation once showed up as run-to-run differences of W in the single-launch path.  Three layers:
the selects are now instructions the compiler knows (``sel64`` via ``inverse_ballot``: no inline-assembly vector write
feeds an MFMA any more); `make` runs tools/check_asm_mfma_hazard.py on every library it builds and REMOVES a library
that fails; and these tests hold the checker itself to crafted violations -- straight-line and across a loop back-edge."""
import importlib.util
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHECKER = os.path.join(ROOT, "tools", "check_asm_mfma_hazard.py")
HAVE_HIPCC = shutil.which("hipcc") is not None or os.path.exists("/opt/rocm/bin/hipcc")


def _checker():
    spec = importlib.util.spec_from_file_location("check_asm_mfma_hazard", CHECKER)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("name,mfma_bad,sgpr_bad", [
    ("ok_straight", 0, 0),            # compare ... two instructions ... select; select, s_nop 1, MFMA
    ("bad_straight", 1, 0),           # inline-asm select, one instruction, MFMA
    ("bad_backedge", 1, 0),           # inline-asm select at the bottom of a loop, MFMA at its top: only visible along the back-edge
    ("bad_sgpr_backedge", 0, 1),      # inline-asm compare at the bottom of a loop, select at its top
])
def test_checker_on_crafted_isa(name, mfma_bad, sgpr_bad):
    path = os.path.join(ROOT, "tests", "isa", name + ".s")
    _, bad, sbad = _checker().check(path, check_sgpr=True, verbose=False)
    assert (bad, sbad) == (mfma_bad, sgpr_bad)
    rc = subprocess.run([sys.executable, CHECKER, path, "--sgpr"], capture_output=True, text=True).returncode
    assert rc == (1 if mfma_bad or sgpr_bad else 0)


def test_straight_line_scan_would_miss_the_back_edge():
    """The round-2 checker walked the file top to bottom; the loop-carried case is exactly what it could not see:
    in bad_backedge.s the MFMA at the loop head comes BEFORE the inline-assembly write in file order."""
    lines = open(os.path.join(ROOT, "tests", "isa", "bad_backedge.s")).read().splitlines()
    first_mfma = next(i for i, s in enumerate(lines) if "v_mfma" in s)
    first_asm = next(i for i, s in enumerate(lines) if "ASMSTART" in s)
    assert first_mfma < first_asm


@pytest.mark.skipif(not HAVE_HIPCC, reason="needs hipcc")
def test_the_shipped_library_passes_and_a_seeded_violation_fails_the_build(tmp_path):
    csrc = os.path.join(ROOT, "nbmf_mm_amd", "csrc")
    # the real build: `make` itself runs the checker (a failing library would have been removed)
    build = subprocess.run(["make", "-C", csrc], capture_output=True, text=True)
    assert build.returncode == 0, (build.stdout + build.stderr)[-2000:]
    assert os.path.exists(os.path.join(ROOT, "nbmf_mm_amd", "libnbmf_hip.so"))
    # the ISA under its fixed name: left there by whichever build made the library (plain `make`, _hip._autobuild under
    # a temporary library name), or made now from the same sources and flags
    mk = subprocess.run(["make", "-C", csrc, "isa"], capture_output=True, text=True)
    assert mk.returncode == 0, (mk.stdout + mk.stderr)[-2000:]
    isa = os.path.join(ROOT, "build", "nbmf_hip.isa.s")
    n, bad, sbad = _checker().check(isa, check_sgpr=True, verbose=False)
    assert n > 2000 and (bad, sbad) == (0, 0)
    # two more properties of the ISA the sources rely on:
    text = open(isa).read()
    import re
    # (1) the sweeps have NO static LDS, so their dynamic region starts at address 0 and the general path's table is
    #     gathered with an immediate offset (pass_kernel: LTAB_BYTE; it also returns at once if this ever changed)
    sizes = re.findall(r"\.amdhsa_kernel (\S*pass_kernel\S*)\n(?:.*\n)*?\s*\.amdhsa_group_segment_fixed_size (\d+)", text)
    assert len(sizes) > 60 and all(int(b) == 0 for _, b in sizes), [s_ for s_ in sizes if int(s_[1])][:3]
    # (2) the peak self-test's MFMAs are the VGPR form (bench.py's roofline.peak_measured would read ~8 % low otherwise)
    body = text[text.index("mfma_peak_kernelEPdidd:"):]
    body = body[:body.index(".Lfunc_end")]
    assert body.count("v_mfma_f64_16x16x4_f64 v[") >= 8 and "a[" not in body and "accvgpr" not in body
    # (3) NO kernel of the library spills: every .amdhsa_kernel has a private segment of 0 bytes and no scratch_ instruction.
    #     (Round 4 shipped pass_kernel<1, F64, L> with 144 bytes of scratch per lane -- compiled for three workgroups per CU,
    #     168 registers -- and two small_fit_kernel variants with 36: nobody looked.  pass_wgs_per_cu() in
    #     nbmf_pass_kernel.inc and the launch bounds of the other kernels are held to this here.)
    priv = re.findall(r"\.amdhsa_kernel (\S+)\n(?:.*\n)*?\s*\.amdhsa_private_segment_fixed_size (\d+)", text)
    assert len(priv) > 140
    spilling = [(name, int(b)) for name, b in priv if int(b) != 0]
    assert not spilling, spilling
    assert not re.search(r"^\s*scratch_(load|store)", text, re.M)
    # the same sources with one deliberately unprotected MFMA compiled in: make must fail and leave no library behind
    out = str(tmp_path / "libseed.so")
    flags = "-O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-value -Wno-unused-result -DNBMF_HAZARD_SEED=1"
    seeded = subprocess.run(["make", "-C", csrc, f"OUT={out}", f"CXXFLAGS={flags}"], capture_output=True, text=True)
    assert seeded.returncode != 0
    assert "hazard_seed_kernel" in seeded.stdout and "ISA hazard check FAILED" in seeded.stdout
    assert not os.path.exists(out)
