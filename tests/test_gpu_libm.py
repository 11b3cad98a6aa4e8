"""The device's expf / logf against the host libm over their WHOLE input ranges (GPU box only).

The reference's arithmetic (harmonic_cpu.cpp:65-70) is libm's expf / logf; epic_amd/csrc/cell_update.h restates them in
f64 on the device.  test_gpu_parity.py spot-checks the restatement through ctypes; this file checks every input."""
import ctypes as ct
import os

import numpy as np
import pytest

import _oracle as O
from epic_amd import epic_harmonic as eh

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]

E = eh._epic


@pytest.mark.parametrize("form", [0, 2], ids=["plain", "kept-addend"])
def test_device_libm_replica_exhaustive(form):
    """(form 2: the same routines as the red-black kernels instantiate them, epic_hip.h: epic_hip_eval_math.)  Every input the sweeps can hand to the device's expf / logf -- ALL 1 120 927 745 floats in [-104, -0] for exp (below
    that the result is 0 on both sides, checked on a sample) and ALL floats in [1, 6] for log -- gives the bits of the
    host libm (the arithmetic harmonic_cpu.cpp:65-70 runs on).  The comparison runs in the checker
    (oracle_libm_mismatches, OpenMP), chunk by chunk."""
    import torch

    dev = torch.device("cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    threads = max(1, min(16, len(os.sched_getaffinity(0))))
    chunk = 1 << 26

    def sweep_range(which, first_bits, last_bits):
        total_bad, checked = 0, 0
        lo = first_bits
        while lo <= last_bits:
            n = min(chunk, last_bits - lo + 1)
            bits = torch.arange(lo, lo + n, dtype=torch.int64, device=dev)
            bits = torch.where(bits >= 2 ** 31, bits - 2 ** 32, bits).to(torch.int32)
            d_in = bits.view(torch.float32)
            d_out = torch.empty_like(d_in)
            assert E.epic_hip_eval_math(d_in.data_ptr(), d_out.data_ptr(), n, which | form, s) == 0
            torch.cuda.synchronize()
            got = d_out.cpu().numpy()
            first_bad = ct.c_size_t(0)
            bad = O.oracle().oracle_libm_mismatches(which, lo, n, got.ctypes.data, ct.byref(first_bad), threads)
            assert bad == 0, "%s: %d mismatches in chunk at bits 0x%08x, first at input bits 0x%08x" % (
                ("expf", "logf")[which], bad, lo, lo + first_bad.value)
            total_bad += bad
            checked += n
            lo += n
            del bits, d_in, d_out
        return checked

    n_log = sweep_range(1, int(np.float32(1.0).view(np.uint32)), int(np.float32(6.0).view(np.uint32)))
    n_exp = sweep_range(0, 0x80000000, int(np.float32(-104.0).view(np.uint32)))
    assert n_exp == 1120927745 and n_log == 20971521
    # beyond -104 the terms vanish on both sides (the sweeps meet such arguments next to obstacles: u = -1e6)
    x = -np.geomspace(104.0, 3.0e6, 100000).astype(np.float32)
    d_in = torch.from_numpy(x).to(dev)
    d_out = torch.empty_like(d_in)
    assert E.epic_hip_eval_math(d_in.data_ptr(), d_out.data_ptr(), x.size, form, s) == 0
    torch.cuda.synchronize()
    assert not d_out.cpu().numpy().any()
    # +0.0 is what a tie produces (w - mx with w == mx); the range above starts at -0.0
    z = torch.zeros(64, dtype=torch.float32, device=dev)
    o = torch.empty_like(z)
    assert E.epic_hip_eval_math(z.data_ptr(), o.data_ptr(), 64, form, s) == 0
    torch.cuda.synchronize()
    assert o.cpu().numpy().tolist() == [1.0] * 64
