"""The library's HOST driver under the sanitizers with fault injection (CPU container; nothing here needs or touches a GPU).

epic_amd/csrc/driver_*.hip + driver_config.cpp (one file, harmonic_gpu.hip, until round 5) -- the registry of contexts, the
device-state lifecycle with its ~20 allocation sites, the driver loops, the multi-device mode with its issuing threads -- are
compiled UNCHANGED with g++ against a fake HIP runtime
(tests/fake_hip/: malloc-backed memory, streams that execute at once, no-op kernels, "fail the n-th call") and driven by
tests/fake_hip/driver.cpp:

* ASan + UBSan: every fallible runtime call of fifteen call sequences fails in turn (~4 400 runs); after each, the return code
  must be the reference's for what failed, nothing may be left behind (device / pinned memory, streams, events), the struct's
  d_* must be null, and the same sequence must run cleanly afterwards.  The reference's own unwind leaks on several of these
  paths (/root/reference/libepic/src/harmonic/harmonic_model_gpu.cu:50-55, harmonic_utilities_gpu.cu:81-135).
* TSan: the scenarios in which one host thread per slab issues the launches (struct Crew: hand-over by generation counter,
  condition variables, per-thread results), with a sample of failing calls.

Test infrastructure only: none of it is compiled into, linked with or loaded by libepic.so.
"""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "epic_amd", "csrc")
FAKE = os.path.join(ROOT, "tests", "fake_hip")

pytestmark = pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
DRIVER_UNITS = ["driver_registry.hip", "driver_plan.hip", "driver_enqueue.hip", "driver_multi.hip", "driver_loop.hip", "driver_ext.hip"]


def build(tmp_path, sanitize):
    flags = ["-std=c++17", "-O1", "-g", "-fsanitize=" + sanitize, "-fno-omit-frame-pointer", "-ffp-contract=off",
             "-I", FAKE, "-I", os.path.join(ROOT, "include")]
    if "undefined" in sanitize:
        flags.append("-fno-sanitize-recover=undefined")
    exe = str(tmp_path / ("fault_driver_" + sanitize.split(",")[0]))
    srcs = [os.path.join(CSRC, f) for f in ("harmonic_cpu.cpp", "harmonic_path_cpu.cpp", "harmonic_legacy_cpu.cpp", "abi_checks.cpp", "driver_config.cpp")]
    # the driver's .hip files hold host code only (the kernels live in kernels_*.hip, whose launchers the fake replaces)
    driver = [os.path.join(CSRC, f) for f in DRIVER_UNITS]
    subprocess.run(["g++", *flags, "-x", "c++", *driver, os.path.join(FAKE, "fake_hip.cpp"),
                    os.path.join(FAKE, "driver.cpp"), *srcs, "-lpthread", "-o", exe], check=True)
    return exe


def test_the_environment_is_read_in_one_place_only():
    """Round 5: every EPIC_HIP_* knob is parsed once per context into struct Config (epic_amd/csrc/driver_config.cpp); no other
    translation unit of the library calls getenv."""
    offenders = []
    for name in sorted(os.listdir(CSRC)):
        if not name.endswith((".hip", ".cpp", ".h")) or name == "driver_config.cpp":
            continue
        for i, line in enumerate(open(os.path.join(CSRC, name)), 1):
            code = line.split("//")[0]
            if "getenv" in code:
                offenders.append("%s:%d" % (name, i))
    assert not offenders, offenders
    assert sorted(f for f in os.listdir(CSRC) if f.startswith("driver_") and f.endswith(".hip")) == sorted(DRIVER_UNITS)


def test_every_failing_runtime_call_unwinds_cleanly_under_asan_ubsan(tmp_path):
    exe = build(tmp_path, "address,undefined")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    env.pop("LD_PRELOAD", None)
    run = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=1500)
    tail = run.stdout[-3000:] + run.stderr[-3000:]
    assert run.returncode == 0, tail
    assert "fault driver: ok" in run.stdout
    for bad in ("AddressSanitizer", "LeakSanitizer", "runtime error", "EXPECT failed"):
        assert bad not in run.stderr, tail
    walked = [l for l in run.stdout.splitlines() if l.startswith("walked ")]
    assert walked and int(walked[0].split()[1]) > 3000, walked


def test_distinct_devices_keep_the_runtimes_device_rules(tmp_path):
    """FOUR fake devices with the device rules of the real runtime enforced (tests/fake_hip/fake_hip.cpp, "device affinity"): a kernel
    goes into a stream of the CURRENT device and touches only that device's memory or an enabled peer's, an event is recorded into a
    stream of its own device, hipMemcpyPeerAsync names the devices that own the buffers -- with every slab on a device of its own, device
    lists out of order and with repeats, 2-D rows and 3-D planes, fused tol pairs, devices that cannot reach each other (staged halos),
    the caller's current device not 0 (and restored afterwards), clean runs and a sample of failing calls.  No session of this project has
    had two GPUs; this is the closest the multi-device mode's device bookkeeping comes to a node before the first real one."""
    exe = build(tmp_path, "address,undefined")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    env.pop("LD_PRELOAD", None)
    run = subprocess.run([exe, "devices"], capture_output=True, text=True, env=env, timeout=1500)
    tail = run.stdout[-3000:] + run.stderr[-3000:]
    assert run.returncode == 0, tail
    assert "fault driver: ok" in run.stdout
    for bad in ("AddressSanitizer", "LeakSanitizer", "runtime error", "EXPECT failed"):
        assert bad not in run.stderr, tail
    assert run.stdout.count("fallible runtime calls") == 11, tail          # eleven device scenarios ran


def test_the_device_model_catches_a_driver_that_forgets_a_device(tmp_path):
    """The check above is only worth something if it can fail: the same program built from a MUTATED copy of driver_multi.hip -- the
    uploads select the first slab's device instead of each slab's own -- must report violations of the device rules."""
    src = open(os.path.join(CSRC, "driver_multi.hip")).read()
    good = "if (hipSetDevice(sl.dev) != hipSuccess) return EPIC_ERROR_DEVICE_MALLOC;"
    assert src.count(good) == 2
    mutated = tmp_path / "driver_multi_mutated.hip"
    mutated.write_text(src.replace(good, "if (hipSetDevice(c->slabs[0].dev) != hipSuccess) return EPIC_ERROR_DEVICE_MALLOC;"))
    exe = str(tmp_path / "fault_driver_mutated")
    units = [str(mutated) if f == "driver_multi.hip" else os.path.join(CSRC, f) for f in DRIVER_UNITS]
    srcs = [os.path.join(CSRC, f) for f in ("harmonic_cpu.cpp", "harmonic_path_cpu.cpp", "harmonic_legacy_cpu.cpp", "abi_checks.cpp", "driver_config.cpp")]
    subprocess.run(["g++", "-std=c++17", "-O1", "-ffp-contract=off", "-I", FAKE, "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-x", "c++", *units,
                    os.path.join(FAKE, "fake_hip.cpp"), os.path.join(FAKE, "driver.cpp"), *srcs, "-lpthread", "-o", exe], check=True)
    run = subprocess.run([exe, "devices"], capture_output=True, text=True, timeout=1500)
    assert run.returncode != 0 and "fault driver: ok" not in run.stdout
    assert "violations of the device rules" in run.stderr and "while device 0 is current" in run.stderr, run.stderr[-2000:]


def test_issuing_threads_are_clean_under_tsan(tmp_path):
    exe = build(tmp_path, "thread")
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0:second_deadlock_stack=1")
    env.pop("LD_PRELOAD", None)
    for mode in ("threads", "devices"):     # one device named several times; four fake devices under the device rules (issuing threads in both)
        run = subprocess.run([exe, mode], capture_output=True, text=True, env=env, timeout=1500)
        tail = run.stdout[-3000:] + run.stderr[-3000:]
        assert run.returncode == 0, (mode, tail)
        assert "fault driver: ok" in run.stdout, (mode, tail)
        assert "ThreadSanitizer" not in run.stderr and "EXPECT failed" not in run.stderr, (mode, tail)
