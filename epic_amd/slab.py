"""Row-slab decomposition of one 2-D relaxation over several GPUs, one process per GPU.

Not in the reference (it has no multi-GPU code); BASELINE.json asks for it: the global grid is cut
into `world` contiguous row slabs, each rank keeps its slab (+ one ghost row per interior side) resident in HBM and,
per Jacobi sweep, exchanges exactly one boundary row with each neighbour -- the only data the 5-point stencil needs
from the other side.  No other collective sits on the data path; the convergence test is one MAX all-reduce of a
single float every `stagger` sweeps.

Per sweep and rank:
    second stream  : sweep the first and the last owned row (2 x 1 row), then isend those two fresh rows and irecv the
                     neighbours' rows into the ghost rows of the OUTPUT buffer (torch.distributed P2P = RCCL send/recv
                     over xGMI; one pitch x 4 B message each way per neighbour)
    compute stream : sweep every other owned row, concurrently with the above
    join           : the compute stream waits for the exchange; swap buffers
The sweeps themselves are the library's raw operator (include/epic_hip.h: epic_hip_sweep_2d) on torch-owned device
memory and torch's streams: PyTorch is plumbing here (memory, streams, process group), the arithmetic is the HIP kernel.

The sweep backend is injected so that the decomposition / exchange logic can be exercised on CPU tensors with the
gloo backend (tests/test_slab_gloo.py passes the checker's row sweep).  There is no CPU fallback: the default
backend needs the HIP library and a GPU.
"""
import ctypes as ct
import os

import numpy as np

import torch
import torch.distributed as dist

from .synthetic import DEFAULT_SEED, _GOLD, _mix64


def partition_rows(rows, world):
    """Contiguous, near-equal row ranges [(lo, hi), ...] covering [0, rows)."""
    base, rem = divmod(rows, world)
    out, lo = [], 0
    for r in range(world):
        hi = lo + base + (1 if r < rem else 0)
        out.append((lo, hi))
        lo = hi
    return out


def synthetic_rows(m, row_lo, row_hi, seed=DEFAULT_SEED, density=0.05):
    """Rows [row_lo, row_hi) of epic_amd.synthetic.synthetic_grid(m) without building the whole grid."""
    rows, cols = int(m[0]), int(m[1])
    thresh = np.uint64(int(density * 9007199254740992.0))
    n = (row_hi - row_lo) * cols
    locked = np.empty(n, dtype=np.uint32)
    u = np.full(n, -1e6, dtype=np.float32)
    chunk_rows = max(1, (1 << 24) // cols)
    with np.errstate(over="ignore"):
        for r0 in range(row_lo, row_hi, chunk_rows):
            r1 = min(row_hi, r0 + chunk_rows)
            idx = np.arange(r0 * cols, r1 * cols, dtype=np.uint64)
            h = _mix64(np.uint64(seed) ^ (idx * _GOLD))
            obstacle = (h >> np.uint64(11)) < thresh
            rr = idx // np.uint64(cols)
            cc = idx % np.uint64(cols)
            border = (rr == 0) | (rr == np.uint64(rows - 1)) | (cc == 0) | (cc == np.uint64(cols - 1))
            locked[(r0 - row_lo) * cols:(r1 - row_lo) * cols] = obstacle | border
    gr, gc = rows // 2, cols // 2
    if row_lo <= gr < row_hi:
        u[(gr - row_lo) * cols + gc] = 0.0
        locked[(gr - row_lo) * cols + gc] = 1
    return u, locked


class HipBackend:
    """Sweeps through libepic.so's raw operators on the current torch stream."""

    def __init__(self, rows_per_task=0, math="precise"):
        from . import epic_harmonic as eh

        self.E = eh._epic
        if self.E.epic_hip_device_count() < 1:
            raise RuntimeError("epic_amd.slab: no HIP device -- the slab solver has no CPU path")
        self.rows_per_task = int(rows_per_task) or (10 if math == "tol" else 16)   # the tol kernel's row loop runs in trips of 10
        self.rows_per_pair = int(os.environ.get("EPIC_HIP_FUSED_ROWS", "0"))   # task height of the fused double sweep; 0 = the library's rule for the slab's size
        modes = {"precise": 0, "fast": 1, "traffic": 2, "tol": 4}
        if math not in modes:
            raise ValueError("epic_amd.slab: unknown math mode %r (one of %s)" % (math, ", ".join(sorted(modes))))
        self.math = modes[math]
        self.maskf = None   # the masks in the fused passes' layout, made by pack_mask

    def pitch_for(self, cols):
        return int(self.E.epic_hip_pitch_for_cols(cols))

    def mask_words(self, rows, pitch):
        return int(self.E.epic_hip_mask_words_2d(rows, pitch))

    def pack_mask(self, locked_i32, rows, cols, pitch, ghost_top, ghost_bottom, maskw):
        rc = self.E.epic_hip_pack_mask_2d(locked_i32.data_ptr(), rows, cols, pitch, int(ghost_top), int(ghost_bottom),
                                          maskw.data_ptr(), torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            raise RuntimeError("epic_hip_pack_mask_2d failed: %d" % rc)
        # the same masks cut for the fused passes' lane mapping (include/epic_hip.h: epic_hip_fuse_masks_2d)
        words = int(self.E.epic_hip_mask_words_fused_2d(rows, pitch))
        if self.maskf is None or self.maskf.numel() != words or self.maskf.device != maskw.device:
            self.maskf = torch.empty(words, dtype=torch.int32, device=maskw.device)
        rc = self.E.epic_hip_fuse_masks_2d(maskw.data_ptr(), rows, pitch, self.maskf.data_ptr(), torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            raise RuntimeError("epic_hip_fuse_masks_2d failed: %d" % rc)

    def sweep(self, src, dst, maskw, rows, pitch, row_begin, row_end, delta_bits):
        if row_end <= row_begin:
            return
        rc = self.E.epic_hip_sweep_2d(src.data_ptr(), dst.data_ptr(), maskw.data_ptr(), rows, pitch, row_begin, row_end,
                                      self.rows_per_task, self.math,
                                      delta_bits.data_ptr() if delta_bits is not None else None,
                                      torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            raise RuntimeError("epic_hip_sweep_2d failed: %d" % rc)

    @property
    def pairs(self):
        """Whether sweep2() exists for this arithmetic (the fused double sweep: tol math)."""
        return self.math == 4 and os.environ.get("EPIC_HIP_NO_FUSE") is None

    def sweep2(self, src, dst, maskw, rows, pitch):
        """Two Jacobi sweeps of all local rows in one pass (src -> dst), bit-identical to two sweep() calls."""
        rc = self.E.epic_hip_sweep2_2d(src.data_ptr(), dst.data_ptr(), maskw.data_ptr(), rows, pitch, self.rows_per_pair,
                                       self.math, torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            raise RuntimeError("epic_hip_sweep2_2d failed: %d" % rc)

    def sweeps(self, a, b, maskw, rows, pitch, n):
        """n plain Jacobi sweeps of all local rows in ONE library call (pairs as fused passes with the tol math), the field
        starting in `a`; returns 1 if the result is in `b` (an odd number of buffer changes), 0 if it is back in `a`."""
        flips = ct.c_int(0)
        rc = self.E.epic_hip_sweeps_2d(a.data_ptr(), b.data_ptr(), maskw.data_ptr(),
                                       self.maskf.data_ptr() if self.maskf is not None else None, rows, pitch, int(n),
                                       self.rows_per_task, self.rows_per_pair, self.math, ct.byref(flips),
                                       torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            raise RuntimeError("epic_hip_sweeps_2d failed: %d" % rc)
        return flips.value

    def sweep_rb(self, u, maskw, rows, pitch, row_begin, row_end, parity, delta_bits):
        """One in-place red-black half-sweep of local rows [row_begin, row_end): cells with (row + col + parity) odd."""
        if row_end <= row_begin:
            return
        rc = self.E.epic_hip_sweep_rb_2d(u.data_ptr(), maskw.data_ptr(), rows, pitch, row_begin, row_end,
                                         self.rows_per_task, self.math, int(parity) & 1,
                                         delta_bits.data_ptr() if delta_bits is not None else None,
                                         torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            raise RuntimeError("epic_hip_sweep_rb_2d failed: %d" % rc)


class SlabSolver:
    """One rank's share of a row-slab-decomposed 2-D relaxation.

    halo = G ghost rows per interior side.  After an exchange all ghost rows are exact; the outermost one cannot be
    updated locally (nothing above it), so with every sweep one more ghost row goes stale from the outside in, and after
    G sweeps the staleness would reach the owned rows -- that is when the next exchange happens.  So the ranks trade
    G rows every G sweeps instead of 1 row every sweep: the same bytes in G times fewer, G times larger messages (the
    exchange is latency-bound: 32 KiB per row at 8192 columns), for 2 G extra rows of arithmetic per sweep.  The ghost
    rows are swept with their true masks, so every owned cell sees exactly the values a single-domain sweep would."""

    def __init__(self, grid, rank, world, device, stagger=100, epsilon=1e-6, rows_per_task=0, math="precise",
                 backend=None, group=None, halo=8, scheme="jacobi"):
        self.grid = (int(grid[0]), int(grid[1]))
        self.rank, self.world, self.device = rank, world, torch.device(device)
        self.stagger, self.epsilon = int(stagger), float(epsilon)
        self.group = group
        if scheme not in ("jacobi", "redblack"):
            raise ValueError("scheme must be 'jacobi' or 'redblack'")
        # "redblack" = the reference's own iteration (harmonic_cpu.cpp:46-51): one iteration updates, in place, the cells
        # with (global row + column + iteration) odd; with the precise math the distributed result -- field, iteration
        # count, delta -- is harmonic_complete_cpu's, bit for bit.  Ghost rows age exactly as in the Jacobi scheme
        # (a row's update reads the rows next to it as they were one iteration ago), so the same G rows are traded
        # every G iterations.
        self.redblack = scheme == "redblack"
        # EPIC_HIP_JACOBI_CHECKS=reference, as the library reads it (epic_amd/csrc/driver_loop.hip: run_block): every check iteration of a Jacobi run is the
        # reference's red-black half-sweep of that iteration's colour, in place -- the state after a check is the reference's own
        self.jacobi_ref_checks = os.environ.get("EPIC_HIP_JACOBI_CHECKS") in ("reference", "1")
        self.backend = backend if backend is not None else HipBackend(rows_per_task, math)
        parts = partition_rows(self.grid[0], world)
        self.lo, self.hi = parts[rank]
        if self.hi - self.lo < 2:
            raise ValueError("every slab needs at least 2 rows")
        self.halo = max(1, min(int(halo), min(h - l for l, h in parts) // 2)) if world > 1 else 0
        self.g_top = self.halo if rank > 0 else 0
        self.g_bot = self.halo if rank < world - 1 else 0
        self.ghost_top, self.ghost_bottom = self.g_top > 0, self.g_bot > 0
        self.rows = (self.hi - self.lo) + self.g_top + self.g_bot
        self.cols = self.grid[1]
        self.pitch = self.backend.pitch_for(self.cols)
        self.first = self.g_top                      # first owned local row
        self.last = self.rows - 1 - self.g_bot       # last owned local row
        self.since = 0                               # sweeps since the ghost rows were last exchanged
        self.cuda = self.device.type == "cuda"
        self.buf = [torch.full((self.rows, self.pitch), -1e6, dtype=torch.float32, device=self.device) for _ in range(2)]
        self.cur = 0
        self.maskw = torch.zeros(self.backend.mask_words(self.rows, self.pitch), dtype=torch.int32, device=self.device)
        self.delta_bits = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.iteration = 0
        self.delta = self.epsilon + 1.0
        self.free_cells = 0
        if self.cuda:
            self.comm_stream = torch.cuda.Stream(device=self.device)
            self.ev_boundary = torch.cuda.Event()
            self.ev_comm = torch.cuda.Event()

    # ---- data --------------------------------------------------------------------------------------------
    def load_rows(self, u_rows, locked_rows):
        """u_rows / locked_rows: this rank's LOCAL rows including the ghost rows, shape (rows, cols); the ghost rows
        hold the neighbours' true values and masks.  Returns the number of unlocked owned cells."""
        u_rows = np.ascontiguousarray(u_rows, dtype=np.float32).reshape(self.rows, self.cols)
        locked_rows = np.ascontiguousarray(locked_rows, dtype=np.uint32).reshape(self.rows, self.cols)
        for b in self.buf:
            b.fill_(-1e6)
            b[:, :self.cols] = torch.from_numpy(u_rows).to(self.device)
        lk = torch.from_numpy(locked_rows.astype(np.int32)).to(self.device)
        self.locked_rows = lk  # kept for set_cells(): the packed mask format is the backend's business
        # only the outermost ghost row is pinned (it has nothing above it to be computed from)
        self.backend.pack_mask(lk, self.rows, self.cols, self.pitch, self.ghost_top, self.ghost_bottom, self.maskw)
        if self.cuda:
            torch.cuda.synchronize(self.device)
        self.cur = 0
        self.iteration = 0
        self.since = 0
        owned = locked_rows[self.first:self.last + 1]
        interior = np.ones_like(owned, dtype=bool)
        interior[:, 0] = interior[:, -1] = False
        if self.rank == 0:
            interior[0] = False
        if self.rank == self.world - 1:
            interior[-1] = False
        self.free_cells = int(((owned == 0) & interior).sum())
        return self.free_cells

    def load_synthetic(self, seed=DEFAULT_SEED, density=0.05, ramp=0.0):
        """ramp > 0: the developed-like start of epic_amd.synthetic.ramp_rows instead of the all -1e6 one (timing legs only)."""
        u, lk = synthetic_rows(self.grid, self.lo - self.g_top, self.hi + self.g_bot, seed, density)
        if ramp > 0.0:
            from .synthetic import ramp_rows

            ramp_rows(self.grid, self.lo - self.g_top, self.hi + self.g_bot, u, lk, ramp)
        return self.load_rows(u, lk)

    def set_cells(self, v, types):
        """The slab form of harmonic_utilities_set_cells_2d_gpu (harmonic_utilities_gpu.cu:38-138): `v` holds k GLOBAL
        (x = column, y = row) pairs, `types` k cell types (0 goal: u = 0, locked; 1 obstacle: u = -1e6, locked; 2 free:
        u = -1e6, unlocked; border cells stay locked).  Every rank gets the whole list and applies the edits that fall
        into its local rows -- owned AND ghost rows, so that neighbours agree without an exchange -- to both ping-pong
        buffers, then repacks its mask.  Returns the number of edits this rank OWNS."""
        v = np.asarray(v, dtype=np.int64).reshape(-1, 2)
        types = np.asarray(types, dtype=np.int64).reshape(-1)
        if v.shape[0] != types.shape[0]:
            raise ValueError("set_cells: one type per (x, y) pair")
        top = self.lo - self.g_top                      # global row of local row 0
        x, y = v[:, 0], v[:, 1]
        ok = (types >= 0) & (types <= 2) & (x >= 0) & (x < self.cols) & (y >= 0) & (y < self.grid[0])
        here = ok & (y >= top) & (y < top + self.rows)
        owned = int((ok & (y >= self.lo) & (y < self.hi)).sum())
        if not here.any():
            return owned
        # later edits of the same cell win, as in a sequential loop: keep the last occurrence of each cell
        xs, ys, ts = x[here], y[here] - top, types[here]
        key = ys * self.cols + xs
        _, last = np.unique(key[::-1], return_index=True)
        keep = key.size - 1 - last
        xs, ys, ts = xs[keep], ys[keep], ts[keep]
        gy = ys + top
        border = (xs == 0) | (xs == self.cols - 1) | (gy == 0) | (gy == self.grid[0] - 1)
        val = torch.from_numpy(np.where(ts == 0, 0.0, -1e6).astype(np.float32)).to(self.device)
        lock = torch.from_numpy(((ts != 2) | border).astype(np.int32)).to(self.device)
        iy = torch.from_numpy(ys).to(self.device)
        ix = torch.from_numpy(xs).to(self.device)
        for b in self.buf:
            b[iy, ix] = val
        self.locked_rows[iy, ix] = lock
        self.backend.pack_mask(self.locked_rows, self.rows, self.cols, self.pitch, self.ghost_top, self.ghost_bottom,
                               self.maskw)
        if self.cuda:
            torch.cuda.synchronize(self.device)
        return owned

    def owned(self):
        """This rank's owned rows of the current field, (hi - lo, cols), on the host."""
        return self.buf[self.cur][self.first:self.last + 1, :self.cols].cpu().numpy()

    # ---- sweeps ------------------------------------------------------------------------------------------
    def _bands(self, dst):
        """(send band, ghost band, peer) per neighbour: my outermost `halo` owned rows go out, the neighbour's come in."""
        out = []
        if self.ghost_top:
            out.append((dst[self.first:self.first + self.halo], dst[0:self.g_top], self.rank - 1))
        if self.ghost_bottom:
            out.append((dst[self.last + 1 - self.halo:self.last + 1], dst[self.rows - self.g_bot:self.rows], self.rank + 1))
        return out

    def _exchange_staged(self, dst):
        """Device tensors over a host-only process group (gloo): the rows are staged through host memory.  Only for
        exercising the multi-process path on a box with a single GPU; on a multi-GPU node the group is RCCL."""
        torch.cuda.current_stream().synchronize()
        ops, recvs = [], []
        for send, ghost, peer in self._bands(dst):
            buf = torch.empty(ghost.shape, dtype=torch.float32)
            ops.append(dist.P2POp(dist.isend, send.cpu(), peer, self.group))
            ops.append(dist.P2POp(dist.irecv, buf, peer, self.group))
            recvs.append((ghost, buf))
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            w.wait()
        for ghost, buf in recvs:
            ghost.copy_(buf)
        return []

    def _exchange(self, dst):
        if self.cuda and self.world > 1 and dist.get_backend(self.group) == "gloo":
            return self._exchange_staged(dst)
        ops = []
        for send, ghost, peer in self._bands(dst):
            ops.append(dist.P2POp(dist.isend, send, peer, self.group))
            ops.append(dist.P2POp(dist.irecv, ghost, peer, self.group))
        return dist.batch_isend_irecv(ops) if ops else []

    def sweep(self, check=False):
        """One iteration of the whole (distributed) grid: a Jacobi sweep, or a red-black half-sweep.  With check=True the local max |du| of the OWNED rows lands
        in self.delta_bits (float bits); combine across ranks with reduce_delta()."""
        src = self.buf[self.cur]
        dst = src if self.redblack else self.buf[self.cur ^ 1]
        be, d = self.backend, (self.delta_bits if check else None)
        if check:
            self.delta_bits.zero_()
        parity = (self.iteration + self.lo - self.g_top) & 1   # local row 0 is global row lo - g_top

        def rows(lo, hi, delta=None):
            if self.redblack:
                be.sweep_rb(src, self.maskw, self.rows, self.pitch, lo, hi, parity, delta)
            else:
                be.sweep(src, dst, self.maskw, self.rows, self.pitch, lo, hi, delta)

        if self.world == 1 or self.since + 1 < self.halo:
            # no exchange due: ghost rows (still exact deep enough) are swept like any other row
            if check and (self.ghost_top or self.ghost_bottom):
                rows(0, self.first)
                rows(self.first, self.last + 1, d)
                rows(self.last + 1, self.rows)
            else:
                rows(0, self.rows, d)
            self.since += 1
        else:
            # exchange due after this sweep: the outermost `halo` owned rows of each side first, on the second stream,
            # then their exchange -- concurrent with the interior sweep; the ghost rows are not swept, they are replaced
            g = self.halo
            top_hi = self.first + g if self.ghost_top else self.first
            bot_lo = self.last + 1 - g if self.ghost_bottom else self.last + 1

            def boundary_bands():
                rows(self.first, top_hi, d)
                rows(bot_lo, self.last + 1, d)

            if self.cuda:
                pr = getattr(self, "_probe", None)   # probe_exchange(): timing events around the two concurrent halves
                self.ev_boundary.record()          # everything the previous sweep wrote is visible after this point
                with torch.cuda.stream(self.comm_stream):
                    self.comm_stream.wait_event(self.ev_boundary)
                    if pr:
                        pr["cp0"].record()
                    boundary_bands()
                # the interior sweep is ENQUEUED before the exchange is issued: issuing send / recv costs host time (a group of
                # P2P operations; with gloo it blocks until the rows have gone through the host), and the interior must already be
                # in its queue by then (round 4: the first probe of this path showed the interior starting 300 us late, behind a
                # host that was still inside the exchange call)
                if pr:
                    pr["int0"].record()
                rows(top_hi, bot_lo, d)
                if pr:
                    pr["int1"].record()
                with torch.cuda.stream(self.comm_stream):
                    for w in self._exchange(dst):
                        w.wait()                   # stream-ordered for NCCL/RCCL: makes comm_stream wait, not the host
                    self.ev_comm.record()
                    if pr:
                        pr["cp1"].record()
                torch.cuda.current_stream().wait_event(self.ev_comm)
            else:
                boundary_bands()
                works = self._exchange(dst)
                rows(top_hi, bot_lo, d)
                for w in works:
                    w.wait()
            self.since = 0
        if not self.redblack:
            self.cur ^= 1
        self.iteration += 1

    def probe_exchange(self):
        """One exchange iteration under timing events (GPU, world > 1): the boundary bands + the halo send / recv on the second
        stream against the interior sweep on the compute stream, in microseconds from the earlier of the two starts -- the
        N > 1 bench line carries it per rank, so that the first run on real xGMI says whether the overlap the design rests on
        happened.  Advances the iteration count up to and including the next exchange; returns None where there is nothing to
        probe."""
        if not self.cuda or self.world == 1:
            return None
        while self.since + 1 < self.halo:       # the plain iterations before the one that ends with an exchange
            self.sweep(False)
        self._probe = {k: torch.cuda.Event(enable_timing=True) for k in ("cp0", "cp1", "int0", "int1")}
        try:
            self.sweep(False)
            torch.cuda.synchronize()
            p = self._probe
            i0, i1, c1 = p["cp0"].elapsed_time(p["int0"]), p["cp0"].elapsed_time(p["int1"]), p["cp0"].elapsed_time(p["cp1"])
        finally:
            self._probe = None
        base = min(0.0, i0)
        is_, ie, cs, ce = (i0 - base) * 1e3, (i1 - base) * 1e3, -base * 1e3, (c1 - base) * 1e3
        return {"interior_us": [round(is_, 1), round(ie, 1)], "bands_and_exchange_us": [round(cs, 1), round(ce, 1)],
                "overlap_us": round(max(0.0, min(ie, ce) - max(is_, cs)), 1), "exchange_hidden": bool(ce <= ie),
                "halo_rows": self.halo, "bytes_each_way": int(self.halo * self.pitch * 4)}

    PAIR_HEIGHTS = (20, 23, 26, 29, 32, 35, 38, 40, 41, 43, 46, 49, 52, 58, 64, 80, 96, 128)

    def tune_pairs(self):
        """Measure the task height of the fused double sweep on this slab, as the library does for its own grids
        (epic_amd/csrc/driver_plan.hip: tune_fused_rows -- the time of a pass depends on the height in a way no rule predicts):
        every candidate runs three times from the current buffer into the other one, which the next real pass overwrites.
        Slabs of at least 4 Mcell on a GPU; results do not depend on the height.  Returns the height in use (0: the rule)."""
        be = self.backend
        if (not self.cuda or self.redblack or not getattr(be, "pairs", False) or os.environ.get("EPIC_HIP_FUSED_ROWS")
                or os.environ.get("EPIC_HIP_TUNE", "1")[:1] == "0" or self.rows * self.pitch < (1 << 22)):
            return be.rows_per_pair
        src, dst = self.buf[self.cur], self.buf[self.cur ^ 1]

        def pair():
            # through sweeps(n = 2), i.e. WITH the fused mask layout: the instantiation of the pass that advance() runs
            # (sweep2() has no d_maskf and takes the funnel-shift instantiation, which ranks the heights differently)
            assert be.sweeps(src, dst, self.maskw, self.rows, self.pitch, 2) == 1

        def timed(height):
            be.rows_per_pair = height
            pair()          # warm
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            pair()
            pair()
            e1.record()
            e1.synchronize()
            return e0.elapsed_time(e1)

        rule_ms = timed(0)
        best, best_ms = 0, rule_ms
        for height in self.PAIR_HEIGHTS:
            if height >= self.rows:
                continue
            ms = timed(height)
            if ms < best_ms:
                best, best_ms = height, ms
        be.rows_per_pair = best if best_ms < 0.99 * rule_ms else 0
        return be.rows_per_pair

    def can_pair(self):
        """Two plain Jacobi iterations as ONE pass (backend.sweep2: the fused double sweep of the tol math, 4 B of HBM
        traffic per cell-update instead of 8): possible when neither of them ends with an exchange -- a pass leaves two
        more ghost rows stale."""
        return (not self.redblack and getattr(self.backend, "pairs", False)
                and (self.world == 1 or self.since + 2 < self.halo))

    def sweep_pair(self):
        self.backend.sweep2(self.buf[self.cur], self.buf[self.cur ^ 1], self.maskw, self.rows, self.pitch)
        self.since += 2
        self.cur ^= 1
        self.iteration += 2

    def advance(self, budget):
        """The next iteration -- or the next two as one pass where that is possible and neither is a check iteration --
        of at most `budget`; returns (iterations done, whether the last one was a check)."""
        check = self.iteration % self.stagger == 0
        if not check and not self.redblack and hasattr(self.backend, "sweeps") and os.environ.get("EPIC_SLAB_ONE_BY_ONE") is None:
            # the whole stretch of plain iterations up to the next check, or to the iteration that ends with an exchange, in ONE
            # call into the library: the interpreter's per-call cost is paid once per stretch, not once per launch
            n = min(int(budget), self.stagger - self.iteration % self.stagger)
            if self.world > 1:
                n = min(n, self.halo - 1 - self.since)
            if n >= 2:
                flips = self.backend.sweeps(self.buf[self.cur], self.buf[self.cur ^ 1], self.maskw, self.rows, self.pitch, n)
                self.cur ^= flips & 1
                self.since += n
                self.iteration += n
                return n, False
        if not check and budget >= 2 and (self.iteration + 1) % self.stagger != 0 and self.can_pair():
            self.sweep_pair()
            return 2, False
        if check and not self.redblack and self.jacobi_ref_checks:
            self.redblack = True
            try:
                self.sweep(True)
            finally:
                self.redblack = False
            return 1, True
        self.sweep(check)
        return 1, check

    def reduce_delta(self):
        """Global max |du| of the last check sweep (one MAX all-reduce of one float)."""
        t = self.delta_bits.view(torch.float32).clone()
        if self.world > 1:
            if self.cuda and dist.get_backend(self.group) == "gloo":
                t = t.cpu()
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        self.delta = float(t.item())
        return self.delta

    def step(self):
        """One pass of the reference's driver loop over `stagger` iterations (harmonic_gpu.cu:266-290): a check
        sweep when iteration % stagger == 0, plain sweeps otherwise.  Returns True if the check sweep converged."""
        converged = False
        left = self.stagger
        while left > 0:
            done, check = self.advance(left)
            left -= done
            if check:
                converged = self.reduce_delta() < self.epsilon
        return converged

    def timed_step(self):
        """step() bracketed by events on the compute stream; returns device milliseconds."""
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        self.step()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1)

    def solve(self, max_sweeps=None):
        """Relax until a check sweep finds delta < epsilon with iteration >= max(grid) (the reference's exit rule).
        Jacobi hands over to the reference's red-black half-sweeps at the first check with delta < 1 that is not below the
        previous check's delta, exactly as harmonic_execute_gpu does (epic_amd/csrc/driver_loop.hip, "Jacobi handover":
        Jacobi's two colour chains can stagnate one ulp apart and never meet the absolute test).  delta is the all-reduced
        value, so every rank takes the decision at the same iteration."""
        self.iteration = 0
        floor = max(self.grid)
        result = False
        last_check, handed_over = -1.0, False
        # tol math: the loop finishes with the reference's own iteration (precise red-black) from the first check with
        # delta < 10 epsilon (100 epsilon for epsilon <= 1e-5), and only a check of that phase may end it -- harmonic_execute_gpu's "Finish" rule, same float
        # arithmetic for the limit; EPIC_HIP_TOL_FINISH=0 switches it off
        import numpy as np

        be = self.backend
        # (the switch is honoured for relaxations to stagnation only, epsilon <= 1e-5, as in harmonic_execute_gpu)
        finish_off = os.environ.get("EPIC_HIP_TOL_FINISH", "1")[:1] == "0" and np.float32(self.epsilon) <= np.float32(1e-5)
        finish_wanted = getattr(be, "math", 0) == 4 and not finish_off
        factor = np.float32(100.0) if np.float32(self.epsilon) <= np.float32(1e-5) else np.float32(10.0)   # as harmonic_execute_gpu
        finish_below = float(factor * np.float32(self.epsilon))
        finishing, math0, redblack0 = False, getattr(be, "math", 0), self.redblack
        try:
            while not result or self.iteration < floor:
                # (an iteration count that must not be passed -- max_sweeps -- bounds a pair as well)
                _, check = self.advance(2 if max_sweeps is None else max_sweeps - self.iteration)
                result = (self.reduce_delta() < self.epsilon) if check else False
                if check:
                    # (not at the first check of a run that has not moved yet: delta == 0 exactly -- harmonic_execute_gpu's rule, round 6)
                    if finish_wanted and not finishing and self.delta < finish_below and not (self.delta == 0.0 and last_check < 0.0):
                        # (at the callers' epsilons this check keeps its verdict: harmonic_execute_gpu's rule, round 6)
                        finishing, result = True, bool(result and np.float32(self.epsilon) > np.float32(1e-5))
                        be.math = 0
                        self.redblack = True
                    elif not self.redblack and not result and self.delta < 1.0 and 0.0 <= last_check <= self.delta:
                        self.redblack = handed_over = True
                    last_check = self.delta
                if max_sweeps is not None and self.iteration >= max_sweeps:
                    break
        finally:
            if finishing:
                be.math, self.redblack = math0, redblack0
            elif handed_over:
                self.redblack = False
        return self.iteration
