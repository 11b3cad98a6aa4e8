"""Host half of the library and the checker under AddressSanitizer + UndefinedBehaviorSanitizer (the
reference has no sanitizer runs; GPU ASan is not available on the pool, so the device half is covered by parity tests)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "epic_amd", "csrc")


@pytest.mark.skipif(shutil.which("g++") is None or shutil.which("gcc") is None, reason="needs gcc/g++")
def test_cpu_exports_and_checker_are_clean_under_asan_ubsan(tmp_path):
    flags = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined",
             "-ffp-contract=off"]
    exe = str(tmp_path / "sanitize_driver")
    oracle_o = str(tmp_path / "oracle.o")
    subprocess.run(["gcc", "-std=c11", *flags, "-c", os.path.join(ROOT, "oracle", "harmonic_oracle.c"), "-o", oracle_o],
                   check=True)
    srcs = [os.path.join(CSRC, f) for f in ("harmonic_cpu.cpp", "harmonic_path_cpu.cpp", "harmonic_legacy_cpu.cpp",
                                            "abi_checks.cpp")]
    subprocess.run(["g++", "-std=c++17", *flags, "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "sanitize", "driver.cpp"), *srcs, oracle_o, "-lm", "-o", exe], check=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    env.pop("LD_PRELOAD", None)
    run = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=300)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "sanitize driver: ok" in run.stdout
    assert "AddressSanitizer" not in run.stderr and "runtime error" not in run.stderr, run.stderr
