"""Build-time guard for the hazards the compiler does not cover for this code (tools/isa_hazards.py): a VALU write to the
data registers of a 16-byte buffer store right behind it (MI355X: lanes 12..15 of the second register reach memory with the
new value; LLVM pads it only for stores without an SGPR soffset), a DPP instruction within 5 wait states of an EXEC write made
inside inline assembly, and any scratch traffic in a sweep kernel.  No GPU: the gfx950 ISA is generated here by
`make -C epic_amd/csrc asm` and scanned; fixtures hold the two sequences that did corrupt results, which the scanner
must flag."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCAN = os.path.join(ROOT, "tools", "isa_hazards.py")
FIX = os.path.join(ROOT, "tests", "golden", "isa")


def scan(*files):
    r = subprocess.run([sys.executable, SCAN, *files], capture_output=True, text=True)
    return r.returncode, r.stdout


def test_scanner_flags_the_sequences_that_corrupted_results():
    rc, out = scan(os.path.join(FIX, "store_data_hazard.s"))
    assert rc == 1 and "store-data" in out and "v_pk_fma_f32 v[18:19]" in out
    rc, out = scan(os.path.join(FIX, "exec_dpp_hazard.s"))
    assert rc == 1 and "exec-dpp" in out
    rc, out = scan(os.path.join(FIX, "clean.s"))
    assert rc == 0, out


@pytest.mark.timeout(900)
def test_generated_isa_of_the_sweep_kernels_is_clean():
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc here")
    csrc = os.path.join(ROOT, "epic_amd", "csrc")
    subprocess.run(["make", "-s", "-C", csrc, "asm"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    asm = os.path.join(csrc, "build", "asm")
    files = [os.path.join(asm, f) for f in ("kernels_2d-hip-amdgcn-amd-amdhsa-gfx950.s", "kernels_3d-hip-amdgcn-amd-amdhsa-gfx950.s")]
    assert all(os.path.exists(f) for f in files)
    rc, out = scan(*files)
    assert rc == 0, out[-4000:]
    assert "kernels scanned, 0 findings" in out and not out.startswith("0 kernels")
