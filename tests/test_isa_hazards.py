"""Static check of the compiled kernels: no MFMA reads a register right behind an inline-assembly write.

gfx950 needs two wait states between a vector instruction writing a VGPR and an MFMA reading it; hipcc inserts them
for instructions it knows, not behind the hand-written selects (inline assembly).  A violation showed up as run-to-run
differences of W in the single-launch path, so the ISA is checked on every build (tools/check_asm_mfma_hazard.py)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_no_mfma_reads_a_fresh_inline_asm_result():
    os.makedirs(os.path.join(ROOT, "build"), exist_ok=True)
    build = subprocess.run(["make", "-C", os.path.join(ROOT, "nbmf_mm_amd", "csrc"), "asm"], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-2000:]
    check = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_asm_mfma_hazard.py"),
                            os.path.join(ROOT, "build", "nbmf_hip.s"), "--sgpr"], capture_output=True, text=True)
    assert check.returncode == 0, check.stdout[-3000:]
    assert "MFMA instructions checked, 0 too close" in check.stdout and "; 0 vector reads" in check.stdout
