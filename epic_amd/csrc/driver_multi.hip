#include "driver.h"

namespace epic_drv {

// ---------------------------------------------------------------------------------------------------------
// several devices in one process (EPIC_HIP_DEVICES): SURVEY.md section 8(e) -- the ABI is a plain in-process C call, so
// the callers that only ever make that call (the ROS plugin: harmonic_complete_gpu, src/epic_nav_core_plugin.cpp:256)
// get a whole node by setting one environment variable.  Not in the reference (it has no multi-GPU code).
//
// The grid is cut along its slowest axis into one slab per listed device: rows of a 2-D grid, planes of a 3-D one ("units"
// below).  Every interior side of a slab carries G = `halo` ghost units that are swept like owned ones, with their true
// masks; the outermost ghost unit has nothing beyond it to be computed from, so with every iteration one more ghost unit goes
// stale from the outside in, and after G iterations the neighbours trade their G outermost owned units -- the bytes of one
// unit per iteration in G times fewer, G times larger copies.  Every owned cell sees exactly the values a single-domain
// iteration would give it: fields, max |du| (ghost units are swept but kept out of the convergence test) and iteration
// counts are bit-identical to the single-device path for any number of slabs and any G (tests/test_gpu_multi_device.py runs
// the parity suite with EPIC_HIP_DEVICES=0,0,0,0).  Red-black: the colour of a local unit is the colour of its GLOBAL unit
// (parity shifted by the slab's first global unit).
//
// Transport (the product choice): the copy engines, device to device -- hipMemcpyPeerAsync on the RECEIVER's second stream
// behind the sender's event, peer access enabled where the fabric offers it (xGMI on an MI355X node); where it cannot be
// enabled the library says so once on stderr and stages the units through pinned host memory itself (two copies and an
// event; EPIC_HIP_NO_PEER=1 forces that path).  RCCL is not linked: its send / recv pairs would bring a communicator per
// process and a kernel per message into a library whose callers (one C call from a ROS node) have neither a launcher nor
// ranks, for messages -- G rows of 32 KiB -- that a copy engine moves without occupying a CU; the one-process-per-GPU form
// over RCCL lives beside the library (epic_amd/slab.py, what bench.py --gpus N runs).
//
// Issue: one host thread per slab (struct Crew).  A single thread issuing N launches of ~3.5 us each is as slow as a
// 1024-row slab's 14 us sweep at N = 8; the crew's threads each own one device (hipSetDevice once) and are handed whole
// stretches of iterations -- everything up to the next exchange -- in one hand-over.  EPIC_HIP_THREADS=0: the caller's thread
// issues everything (A/B, debugging).
//
// Activity tracking works per slab (Track, one per slab): a slab's sweep is one list-driven launch over its local tiles;
// after an exchange the tiles that hold or read the rewritten ghost units are woken for the next launch.
// ---------------------------------------------------------------------------------------------------------
// One host thread per slab.  run(f) has every thread call f(k) for its slab k and returns the first error once all are
// back; the threads spin for a SHORT, BOUNDED while between hand-overs -- spin_us microseconds (EPIC_HIP_SPIN_US, default 20:
// a relaxation hands over every few tens of microseconds) -- and sleep on a condition variable otherwise: an idle Harmonic
// costs its host process nothing, and a busy one at most spin_us per hand-over and thread (until round 5 the spin was 20 000
// pause instructions, ~1 ms: eight slabs kept eight of a ROS node's cores busy between calls).  EPIC_HIP_SPIN_US=0: no spinning.
struct Crew {
    int spin_us = 20;
    // spin until `done()` or until spin_us have passed (the clock is looked at every 32 pauses)
    template <class F> void spin_for(F done) const
    {
        if (spin_us <= 0 || done()) return;
        const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(spin_us);
        for (;;) {
            for (int i = 0; i < 32; i++) {
                if (done()) return;
                __builtin_ia32_pause();
            }
            if (std::chrono::steady_clock::now() >= until) return;
        }
    }
    std::vector<std::thread> threads;
    std::mutex mu;
    std::condition_variable cv_go, cv_done;
    std::atomic<unsigned> generation{0};
    std::atomic<int> pending{0};
    std::function<hipError_t(int)> job;
    std::vector<hipError_t> result;
    bool quit = false;

    void start(const std::vector<int> &devices)
    {
        result.assign(devices.size(), hipSuccess);
        for (size_t k = 0; k < devices.size(); k++)
            threads.emplace_back([this, k, dev = devices[k]] {
                (void)hipSetDevice(dev);
                unsigned seen = 0;
                for (;;) {
                    spin_for([&] { return generation.load(std::memory_order_acquire) != seen; });
                    if (generation.load(std::memory_order_acquire) == seen) {
                        std::unique_lock<std::mutex> lk(mu);
                        cv_go.wait(lk, [&] { return generation.load(std::memory_order_acquire) != seen; });
                    }
                    seen = generation.load(std::memory_order_acquire);
                    if (quit) return;
                    result[k] = job((int)k);
                    if (pending.fetch_sub(1, std::memory_order_acq_rel) == 1) {
                        std::lock_guard<std::mutex> lk(mu);
                        cv_done.notify_all();
                    }
                }
            });
    }
    hipError_t run(std::function<hipError_t(int)> f)
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            job = std::move(f);
            pending.store((int)threads.size(), std::memory_order_release);
            generation.fetch_add(1, std::memory_order_acq_rel);
        }
        cv_go.notify_all();
        spin_for([&] { return pending.load(std::memory_order_acquire) == 0; });
        if (pending.load(std::memory_order_acquire) != 0) {
            std::unique_lock<std::mutex> lk(mu);
            cv_done.wait(lk, [&] { return pending.load(std::memory_order_acquire) == 0; });
        }
        for (hipError_t e : result)
            if (e != hipSuccess) return e;
        return hipSuccess;
    }
    void stop()
    {
        if (threads.empty()) return;
        {
            std::lock_guard<std::mutex> lk(mu);
            quit = true;
            generation.fetch_add(1, std::memory_order_acq_rel);
        }
        cv_go.notify_all();
        for (auto &t : threads) t.join();
        threads.clear();
    }
};

// f(k) for every slab k: by the crew's threads, or one after the other on the caller's thread (with its device restored)
hipError_t for_each_slab(Ctx *c, const std::function<hipError_t(int)> &f)
{
    if (c->crew && !c->crew->threads.empty()) return c->crew->run(f);
    DeviceGuard g;
    for (int k = 0; k < (int)c->slabs.size(); k++) {
        hipError_t e = hipSetDevice(c->slabs[k].dev);
        if (e == hipSuccess) e = f(k);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

void multi_free_u(Ctx *c)
{
    for (auto &sl : c->slabs) {
        (void)hipSetDevice(sl.dev);
        if (sl.stream) (void)hipStreamSynchronize(sl.stream);
        if (sl.comm) (void)hipStreamSynchronize(sl.comm);
        for (float *&b : sl.buf) { if (b) (void)hipFree(b); b = nullptr; }
    }
}
void multi_free_mask(Ctx *c)
{
    for (auto &sl : c->slabs) {
        (void)hipSetDevice(sl.dev);
        if (sl.stream) (void)hipStreamSynchronize(sl.stream);
        if (sl.maskw) (void)hipFree(sl.maskw);
        sl.maskw = nullptr;
    }
}
void multi_free_delta(Ctx *c)
{
    for (auto &sl : c->slabs) {
        (void)hipSetDevice(sl.dev);
        if (sl.stream) (void)hipStreamSynchronize(sl.stream);
        if (sl.d_delta) (void)hipFree(sl.d_delta);
        sl.d_delta = nullptr;
    }
}
bool multi_holds_anything(const Ctx *c)
{
    for (const auto &sl : c->slabs)
        if (sl.buf[0] || sl.maskw || sl.d_delta) return true;
    return false;
}
void multi_destroy(Ctx *c)  // the crew, streams, events, pinned words, work lists; the slabs themselves
{
    DeviceGuard g;
    if (c->crew) { c->crew->stop(); delete c->crew; c->crew = nullptr; }
    multi_free_u(c); multi_free_mask(c); multi_free_delta(c);
    for (auto &sl : c->slabs) {
        (void)hipSetDevice(sl.dev);
        sl.trk.release();
        sl.trk_f.release();
        if (sl.stream) (void)hipStreamDestroy(sl.stream);
        if (sl.comm) (void)hipStreamDestroy(sl.comm);
        for (hipEvent_t e : {sl.ev_prev, sl.ev_band, sl.ev_comm, sl.ev_stage}) if (e) (void)hipEventDestroy(e);
        if (sl.h_delta) (void)hipHostFree(sl.h_delta);
        for (float *&b : sl.bounce) { if (b) (void)hipHostFree(b); b = nullptr; }
    }
    c->slabs.clear();
}

// Decide the mode for the dimensions now in *c and, in multi-device mode, lay the slabs out (no device memory yet).
// Single-device mode when fewer than two devices are listed or the grid is too small to cut.
bool multi_plan(Ctx *c)
{
    const int want = (int)c->devices.size();
    const int units = c->n == 2 ? c->rows : c->m[0];
    const bool multi = want >= 2 && (c->n == 2 || c->n == 3) && units >= 4 * want;
    c->plan_failed = false;
    if (!multi) {
        if (!c->slabs.empty() && !multi_holds_anything(c)) multi_destroy(c);
        return false;
    }
    if ((int)c->slabs.size() == want && c->slabs.back().hi == units && c->slab_n == c->n) return true;  // already laid out for these dimensions
    if (!c->slabs.empty()) multi_destroy(c);
    c->tuned_rows[0] = c->tuned_rows[1] = c->tuned_rows[2] = 0;   // measured on the first slab of the OLD layout
    DeviceGuard g;
    const int base = units / want, rem = units % want;
    // ghost depth G = iterations between two exchanges: an exchange costs a fixed few tens of microseconds while a sweep of
    // a short slab takes ~15, and 2 G extra rows per slab are cheap -- 8 from 4096 rows per device up, 16 from 2048, 32 below;
    // a plane of a 3-D grid is a whole sweep's worth of rows: 2 planes
    const int want_halo = c->halo_env > 0 ? c->halo_env : c->n == 3 ? 2 : base >= 4096 ? 8 : base >= 2048 ? 16 : 32;
    const int halo = std::max(1, std::min(want_halo, base / 2));
    const bool no_peer = c->cfg.no_peer;
    int lo = 0;
    c->slabs.resize(want);
    for (int k = 0; k < want; k++) {
        Ctx::Slab &sl = c->slabs[k];
        sl.dev = c->devices[k];
        sl.lo = lo;
        sl.hi = lo + base + (k < rem ? 1 : 0);
        lo = sl.hi;
        sl.g_top = k > 0 ? halo : 0;
        sl.g_bot = k < want - 1 ? halo : 0;
        sl.rows = (sl.hi - sl.lo) + sl.g_top + sl.g_bot;
        bool ok = hipSetDevice(sl.dev) == hipSuccess &&
                  hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking) == hipSuccess &&
                  hipStreamCreateWithFlags(&sl.comm, hipStreamNonBlocking) == hipSuccess &&
                  hipEventCreateWithFlags(&sl.ev_prev, hipEventDisableTiming) == hipSuccess &&
                  hipEventCreateWithFlags(&sl.ev_band, hipEventDisableTiming) == hipSuccess &&
                  hipEventCreateWithFlags(&sl.ev_comm, hipEventDisableTiming) == hipSuccess &&
                  hipEventCreateWithFlags(&sl.ev_stage, hipEventDisableTiming) == hipSuccess &&
                  hipHostMalloc((void **)&sl.h_delta, 64, hipHostMallocPortable) == hipSuccess;
        if (!ok) {
            (void)hipGetLastError();
            multi_destroy(c);
            c->plan_failed = true;
            return false;
        }
        // the seam between slab k - 1 and this one: direct copies where the fabric allows them, checked in both directions
        sl.peer_up = true;
        if (k > 0) {
            Ctx::Slab &up = c->slabs[k - 1];
            bool direct = !no_peer;
            if (direct && up.dev != sl.dev) {
                auto enable = [](int from, int to) {   // `from` may address `to`'s memory
                    int can = 0;
                    if (hipSetDevice(from) != hipSuccess || hipDeviceCanAccessPeer(&can, from, to) != hipSuccess || !can) {
                        (void)hipGetLastError();
                        return false;
                    }
                    const hipError_t e = hipDeviceEnablePeerAccess(to, 0);
                    (void)hipGetLastError();
                    return e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled;
                };
                direct = enable(sl.dev, up.dev) && enable(up.dev, sl.dev);
                (void)hipSetDevice(sl.dev);
                if (!direct)
                    fprintf(stderr, "Warning[epic_hip]: no peer access between devices %d and %d: halo units are staged through pinned host memory.\n",
                            up.dev, sl.dev);
            }
            sl.peer_up = direct;
            if (!direct) {   // one pinned buffer per direction across this seam
                const size_t bytes = (size_t)halo * (c->n == 2 ? 1 : c->m[1]) * c->pitch * sizeof(float);
                if (hipHostMalloc((void **)&sl.bounce[0], bytes, hipHostMallocPortable) != hipSuccess ||
                    hipHostMalloc((void **)&sl.bounce[1], bytes, hipHostMallocPortable) != hipSuccess) {
                    (void)hipGetLastError();
                    multi_destroy(c);
                    c->plan_failed = true;
                    return false;
                }
            }
        }
    }
    c->halo = halo;
    c->since = 0;
    c->slab_n = c->n;
    if (c->cfg.threads) {
        c->crew = new Crew();
        c->crew->spin_us = c->cfg.spin_us;
        c->crew->start(c->devices);
    }
    resolve_tracking(c);
    return true;
}

bool multi_ready(const Ctx *c)
{
    if (c->slabs.empty()) return false;
    for (const auto &sl : c->slabs)
        if (!sl.buf[0] || !sl.buf[1] || !sl.maskw) return false;
    return true;
}

void multi_sync(Ctx *c)
{
    for (auto &sl : c->slabs) {
        (void)hipSetDevice(sl.dev);
        (void)hipStreamSynchronize(sl.comm);
        (void)hipStreamSynchronize(sl.stream);
    }
}

// geometry of a unit (a row of a 2-D grid, a plane of a 3-D one) on the device and in the caller's arrays
size_t unit_floats(const Ctx *c) { return (size_t)(c->n == 2 ? 1 : c->m[1]) * c->pitch; }
size_t unit_rows(const Ctx *c) { return (size_t)(c->n == 2 ? 1 : c->m[1]); }

int multi_upload_u(Harmonic *h, Ctx *c, const char *fn)
{
    DeviceGuard g;
    for (auto &sl : c->slabs) {
        if (hipSetDevice(sl.dev) != hipSuccess) return EPIC_ERROR_DEVICE_MALLOC;
        for (int b = 0; b < 2; b++)
            if (epic_hip::launch_fill(sl.buf[b], (size_t)sl.rows * unit_floats(c), -1e6f, sl.stream) != hipSuccess) {
                report(fn, "Failed to initialise device-side memory for the potential values.");
                return EPIC_ERROR_KERNEL_EXECUTION;
            }
        if (hipStreamSynchronize(sl.stream) != hipSuccess) return EPIC_ERROR_DEVICE_SYNCHRONIZE;
        if (hipMemcpy2D(sl.buf[0], (size_t)c->pitch * sizeof(float), h->u + (size_t)sl.top() * unit_rows(c) * c->cols,
                        (size_t)c->cols * sizeof(float), (size_t)c->cols * sizeof(float), (size_t)sl.rows * unit_rows(c),
                        hipMemcpyHostToDevice) != hipSuccess) {
            report(fn, "Failed to copy memory from host to device for the potential values.");
            return EPIC_ERROR_MEMCPY_TO_DEVICE;
        }
        sl.trk.force = 2;
        sl.trk_f.force = std::max(sl.trk_f.force, 1);
    }
    c->cur = 0;
    c->since = 0;
    h->d_u = c->slabs[0].buf[0];
    return EPIC_SUCCESS;
}

size_t slab_mask_words(const Ctx *c, const Ctx::Slab &sl)
{
    return c->n == 2 ? Ctx::mask_words_both_2d(sl.rows, c->pitch) : epic_hip::mask_words_3d(sl.rows, c->m[1], c->pitch);
}

int multi_upload_locked(Harmonic *h, Ctx *c, const char *fn)
{
    DeviceGuard g;
    for (auto &sl : c->slabs) {
        if (hipSetDevice(sl.dev) != hipSuccess) return EPIC_ERROR_DEVICE_MALLOC;
        const size_t cells = (size_t)sl.rows * unit_rows(c) * c->cols;
        uint32_t *tmp = nullptr;
        if (hipMalloc((void **)&tmp, cells * sizeof(uint32_t)) != hipSuccess) {
            (void)hipGetLastError();
            report(fn, "Failed to allocate device-side staging memory for the locked cells.");
            return EPIC_ERROR_DEVICE_MALLOC;
        }
        int rc = EPIC_SUCCESS;
        if (hipMemcpy(tmp, h->locked + (size_t)sl.top() * unit_rows(c) * c->cols, cells * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess) {
            report(fn, "Failed to copy memory from host to device for the locked cells.");
            rc = EPIC_ERROR_MEMCPY_TO_DEVICE;
        } else {
            // the outermost unit of a local grid is locked either way: the grid's own border, or a ghost unit that has nothing
            // beyond it to be computed from (2-D: the ghost flags; 3-D: the packer locks the faces of the grid it is given)
            hipError_t e = c->n == 2 ? epic_hip::launch_pack_mask_2d(tmp, sl.rows, c->cols, c->pitch, sl.g_top > 0, sl.g_bot > 0, sl.maskw, sl.stream)
                                     : epic_hip::launch_pack_mask_3d(tmp, sl.rows, c->m[1], c->m[2], c->pitch, sl.maskw, sl.stream);
            if (e == hipSuccess && c->n == 2) e = epic_hip::launch_fuse_masks_2d(sl.maskw, sl.rows, c->pitch, c->maskf(sl), sl.stream);
            if (e != hipSuccess || hipStreamSynchronize(sl.stream) != hipSuccess) {
                report(fn, "Failed to execute the 'pack mask' kernel.");
                rc = EPIC_ERROR_KERNEL_EXECUTION;
            }
        }
        (void)hipFree(tmp);
        if (rc != EPIC_SUCCESS) return rc;
        sl.trk.force = 2;
        sl.trk_f.force = std::max(sl.trk_f.force, 1);
    }
    // (`since` is left alone: the mask does not refresh the ghost units of u -- multi_upload_u does, and resets the countdown)
    return EPIC_SUCCESS;
}

// `n` units from slab `src` (starting at its local unit `sunit`) into slab `dst` (at `dunit`), on dst's second stream, behind
// src's event `after`.  Direct where the seam allows it; otherwise through the seam's pinned buffer `bounce`: device -> host on
// src's second stream, host -> device on dst's, an event in between.
hipError_t multi_copy_units(const Ctx *c, Ctx::Slab &dst, float *dbuf, int dunit, Ctx::Slab &src, const float *sbuf, int sunit, int n,
                            hipEvent_t after, bool direct, float *bounce)
{
    const size_t bytes = (size_t)n * unit_floats(c) * sizeof(float);
    float *d = dbuf + (size_t)dunit * unit_floats(c);
    const float *sp = sbuf + (size_t)sunit * unit_floats(c);
    hipError_t e;
    if (direct) {
        if ((e = hipSetDevice(dst.dev)) != hipSuccess || (e = hipStreamWaitEvent(dst.comm, after, 0)) != hipSuccess) return e;
        return dst.dev == src.dev ? hipMemcpyAsync(d, sp, bytes, hipMemcpyDeviceToDevice, dst.comm)
                                  : hipMemcpyPeerAsync(d, dst.dev, sp, src.dev, bytes, dst.comm);
    }
    if ((e = hipSetDevice(src.dev)) != hipSuccess || (e = hipStreamWaitEvent(src.comm, after, 0)) != hipSuccess ||
        (e = hipMemcpyAsync(bounce, sp, bytes, hipMemcpyDeviceToHost, src.comm)) != hipSuccess ||
        (e = hipEventRecord(src.ev_stage, src.comm)) != hipSuccess)
        return e;
    if ((e = hipSetDevice(dst.dev)) != hipSuccess || (e = hipStreamWaitEvent(dst.comm, src.ev_stage, 0)) != hipSuccess) return e;
    return hipMemcpyAsync(d, bounce, bytes, hipMemcpyHostToDevice, dst.comm);
}

// units [lo, hi) of one slab, one launch; check units [clo, chi) count for max |du| when d != nullptr
hipError_t slab_launch(Ctx *c, Ctx::Slab &sl, int lo, int hi, unsigned *d, int clo, int chi, unsigned iteration, hipStream_t st,
                       const epic_hip::Activity *act)
{
    float *src = sl.buf[c->cur], *dst = c->redblack ? src : sl.buf[c->cur ^ 1];
    const int parity = c->redblack ? (int)((iteration + (unsigned)sl.top()) & 1u) : -1;
    if (c->n == 2)
        return epic_hip::launch_sweep_2d(src, dst, sl.maskw, sl.rows, c->pitch, lo, hi, auto_rows_per_task(c), c->math, parity, d, st, act, clo, chi);
    return epic_hip::launch_sweep_3d(src, dst, sl.maskw, sl.rows, c->m[1], c->pitch, lo, hi, c->math, parity, d, st, act, clo, chi, &c->cfg.launch);
}

size_t slab_tiles(const Ctx *c, const Ctx::Slab &sl, int rpt)
{
    return c->n == 2 ? epic_hip::sweep_2d_tiles(sl.rows, c->pitch, rpt) : epic_hip::sweep_3d_tiles(sl.rows, c->m[1], c->pitch);
}

// Iterations [first, first + count) of the whole grid, enqueued on every slab's streams.  check_first: the first of them is a
// check iteration (its max |du| lands in the slabs' delta words).  Between two exchanges a slab needs nothing from the others:
// each crew thread gets the whole stretch at once; the iteration that ends with an exchange takes three hand-overs (sweeps
// and band events, then every slab pulling its two halos, then the joins).
hipError_t multi_run(Ctx *c, unsigned count, unsigned first, bool check_first)
{
    const int G = c->halo;
    const bool tracked = c->track;
    // pairs of plain iterations as one fused pass, as on one device: Jacobi and red-black with the tol math, red-black with the
    // precise / fast math (2-D grids from 4 Mcell up, no work lists)
    const bool fuse_rb = !tracked && c->redblack && c->n == 2 && c->math != 4 && !c->cfg.no_fuse && (long long)c->rows * c->pitch >= (1ll << 22);
    const bool fuse = !tracked && (fuses_tol(c) || fuse_rb);
    const int rpt_track = c->n == 2 ? auto_rows_per_task(c) : 32;
    note_iterations(c, first, count);
    if (tracked) {   // (as enqueue_sweep: fused passes have run since these lists were made -> one iteration over every tile)
        if (c->last_lists == 2)
            for (auto &sl : c->slabs) sl.trk.force = std::max(sl.trk.force, 1);
        c->last_lists = 1;
    } else {
        c->last_lists = 0;
    }
    unsigned done = 0;
    while (done < count) {
        // a stretch without exchange: iterations that keep `since` below G - 1 at their start
        const unsigned calm = (unsigned)std::max(0, G - 1 - c->since);
        const unsigned n_calm = std::min(count - done, calm);
        if (n_calm > 0) {
            const unsigned it0 = first + done;
            const bool chk = check_first && done == 0;
            const int cur0 = c->cur;
            if (fuse && n_calm >= 2) tune_fused_rows(c, fuse_rb ? 2 : c->redblack ? 1 : 0, it0);
            const int fused_rpt = !fuse ? 0 : fuse_rb ? fused_rows_per_task(c) : jacobi_fused_rows_per_task(c);
            hipError_t e = for_each_slab(c, [&, it0, chk, cur0, fused_rpt, n_calm](int k) -> hipError_t {
                Ctx::Slab &sl = c->slabs[k];
                int cur = cur0;
                hipError_t e = hipSuccess;
                for (unsigned i = 0; i < n_calm && e == hipSuccess;) {
                    const bool check = chk && i == 0;
                    if (fuse && !check && n_calm - i >= 2) {   // two more ghost units go stale: n_calm leaves room for them
                        const int parity = c->redblack ? (int)((it0 + i + (unsigned)sl.top()) & 1u) : -1;
                        e = fuse_rb ? epic_hip::launch_rb_fused_2d(sl.buf[cur], sl.buf[cur ^ 1], sl.maskw, sl.rows, c->pitch, fused_rpt, c->math,
                                                                   parity, sl.stream, c->maskf(sl))
                                    : epic_hip::launch_jacobi_fused_2d(sl.buf[cur], sl.buf[cur ^ 1], sl.maskw, sl.rows, c->pitch, fused_rpt,
                                                                       c->math, sl.stream, parity, c->maskf(sl));
                        cur ^= 1;
                        i += 2;
                        continue;
                    }
                    if (check) e = hipMemsetAsync(sl.d_delta, 0, sizeof(unsigned), sl.stream);
                    epic_hip::Activity act = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
                    if (tracked) act = sl.trk.next(slab_tiles(c, sl, rpt_track), rpt_track, sl.stream, nullptr);
                    if (e == hipSuccess) {
                        float *src = sl.buf[cur], *dst = c->redblack ? src : sl.buf[cur ^ 1];
                        const int parity = c->redblack ? (int)((it0 + i + (unsigned)sl.top()) & 1u) : -1;
                        unsigned *d = check ? sl.d_delta : nullptr;
                        // ghost units (still exact deep enough) are swept like any other; they do not count for max |du|
                        e = c->n == 2 ? epic_hip::launch_sweep_2d(src, dst, sl.maskw, sl.rows, c->pitch, 0, sl.rows, auto_rows_per_task(c), c->math,
                                                                  parity, d, sl.stream, &act, sl.first(), sl.last() + 1)
                                      : epic_hip::launch_sweep_3d(src, dst, sl.maskw, sl.rows, c->m[1], c->pitch, 0, sl.rows, c->math, parity, d,
                                                                  sl.stream, &act, sl.first(), sl.last() + 1, &c->cfg.launch);
                    }
                    if (e == hipSuccess && act.list_out) sl.trk.advance();
                    if (!c->redblack) cur ^= 1;   // (a red-black half-sweep is in place; the fused passes above are in -> out either way)
                    i++;
                }
                return e;
            });
            if (e != hipSuccess) return e;
            // host-side bookkeeping of the stretch (the same walk as the threads made)
            for (unsigned i = 0; i < n_calm;) {
                const bool check = chk && i == 0;
                if (fuse && !check && n_calm - i >= 2) { c->cur ^= 1; c->since += 2; c->work_full += tracked ? 0.0 : 2.0; i += 2; continue; }
                if (!c->redblack) c->cur ^= 1;
                c->since++;
                if (!tracked) c->work_full += 1.0;
                i++;
            }
            // (a fused pair never starts with since == G - 2: it would leave the last ghost unit stale before the exchange)
            done += n_calm;
            if (done >= count) break;
        }
        // the iteration that ends with an exchange
        const unsigned it = first + done;
        const bool check = check_first && done == 0;
        const int cur = c->cur;
        // 1. sweeps.  Untracked: the outermost G owned units of each interior side first, on the second stream, so that they can
        //    travel while the interior is swept (the ghost units are not swept: the exchange replaces them).  Tracked: one
        //    list-driven launch of the whole slab (a launch with work lists covers the whole domain), the copies behind it.
        hipError_t e = for_each_slab(c, [&, it, check, cur](int k) -> hipError_t {
            Ctx::Slab &sl = c->slabs[k];
            hipError_t e = hipSuccess;
            auto fail = [&](hipError_t x) { if (e == hipSuccess && x != hipSuccess) e = x; return x != hipSuccess; };
            unsigned *d = check ? sl.d_delta : nullptr;
            if (check && fail(hipMemsetAsync(sl.d_delta, 0, sizeof(unsigned), sl.stream))) return e;
            (void)cur;
            if (tracked) {
                epic_hip::Activity act = sl.trk.next(slab_tiles(c, sl, rpt_track), rpt_track, sl.stream, nullptr);
                if (fail(slab_launch(c, sl, 0, sl.rows, d, sl.first(), sl.last() + 1, it, sl.stream, &act))) return e;
                if (act.list_out) sl.trk.advance();
                if (fail(hipEventRecord(sl.ev_band, sl.stream))) return e;
                return e;
            }
            const int top_hi = sl.g_top ? sl.first() + G : sl.first();
            const int bot_lo = sl.g_bot ? sl.last() + 1 - G : sl.last() + 1;
            const Ctx::Probe *pr = c->probe.empty() ? nullptr : &c->probe[k];
            if (fail(hipEventRecord(sl.ev_prev, sl.stream)) || fail(hipStreamWaitEvent(sl.comm, sl.ev_prev, 0))) return e;
            if (pr && fail(hipEventRecord(pr->cp0, sl.comm))) return e;    // second stream: bands, then the incoming copies (step 2)
            if (sl.g_top && fail(slab_launch(c, sl, sl.first(), top_hi, d, -1, -1, it, sl.comm, nullptr))) return e;
            if (sl.g_bot && fail(slab_launch(c, sl, bot_lo, sl.last() + 1, d, -1, -1, it, sl.comm, nullptr))) return e;
            if (fail(hipEventRecord(sl.ev_band, sl.comm))) return e;
            if (pr && fail(hipEventRecord(pr->int0, sl.stream))) return e;
            if (fail(slab_launch(c, sl, top_hi, bot_lo, d, -1, -1, it, sl.stream, nullptr))) return e;
            if (pr && fail(hipEventRecord(pr->int1, sl.stream))) return e;
            return e;
        });
        if (e != hipSuccess) return e;
        // 2. every slab pulls its two halos (the neighbours' band events exist now) and records "my second stream is done"
        e = for_each_slab(c, [&, cur](int k) -> hipError_t {
            Ctx::Slab &sl = c->slabs[k];
            const int out = c->redblack ? cur : cur ^ 1;
            // (with work lists the whole slab was swept in one launch, ghost units included: the copies land behind it)
            hipError_t e = tracked ? hipStreamWaitEvent(sl.comm, sl.ev_band, 0) : hipSuccess;
            if (e == hipSuccess && k > 0) {   // the upper neighbour's last G owned units -> my top ghost units
                Ctx::Slab &up = c->slabs[k - 1];
                e = multi_copy_units(c, sl, sl.buf[out], 0, up, up.buf[out], up.last() + 1 - G, G, up.ev_band, sl.peer_up, sl.bounce[0]);
            }
            if (e == hipSuccess && k + 1 < (int)c->slabs.size()) {   // the lower neighbour's first G owned units -> my bottom ghost units
                Ctx::Slab &dn = c->slabs[k + 1];
                e = multi_copy_units(c, sl, sl.buf[out], sl.rows - G, dn, dn.buf[out], dn.first(), G, dn.ev_band, dn.peer_up, dn.bounce[1]);
            }
            if (e == hipSuccess) e = hipSetDevice(sl.dev);
            if (e == hipSuccess) e = hipEventRecord(sl.ev_comm, sl.comm);
            if (e == hipSuccess && !c->probe.empty() && !tracked) e = hipEventRecord(c->probe[k].cp1, sl.comm);
            return e;
        });
        if (e != hipSuccess) return e;
        // 3. a slab's next iteration starts when its own bands and incoming copies are done AND the neighbours have read the
        //    units they copy out of it; with work lists, the tiles that hold or read the rewritten ghost units are woken
        e = for_each_slab(c, [&](int k) -> hipError_t {
            Ctx::Slab &sl = c->slabs[k];
            hipError_t e = hipStreamWaitEvent(sl.stream, sl.ev_comm, 0);
            if (e == hipSuccess && k > 0) e = hipStreamWaitEvent(sl.stream, c->slabs[k - 1].ev_comm, 0);
            if (e == hipSuccess && k + 1 < (int)c->slabs.size()) e = hipStreamWaitEvent(sl.stream, c->slabs[k + 1].ev_comm, 0);
            if (e == hipSuccess && tracked && sl.trk.tiles && sl.trk.force == 0) {
                const epic_hip::Activity next = sl.trk.upcoming();
                const int per_unit = c->n == 2 ? 0 : (int)(sl.trk.tiles / (size_t)sl.rows);   // 3-D: tiles per plane
                auto wake_units = [&](int lo, int hi) {   // tiles that hold units [lo, hi)
                    lo = std::max(lo, 0);
                    hi = std::min(hi, sl.rows);
                    if (hi <= lo) return hipSuccess;
                    const int nstrips = c->pitch / 256;
                    const int t_lo = c->n == 2 ? (lo / sl.trk.rpt) * nstrips : lo * per_unit;
                    const int t_hi = c->n == 2 ? ((hi - 1) / sl.trk.rpt + 1) * nstrips : hi * per_unit;
                    return epic_hip::launch_wake_tile_range(&next, sl.trk.tiles, t_lo, t_hi, sl.stream);
                };
                if (sl.g_top) e = wake_units(0, G + 1);
                if (e == hipSuccess && sl.g_bot) e = wake_units(sl.rows - G - 1, sl.rows);
            }
            return e;
        });
        if (e != hipSuccess) return e;
        if (!c->redblack) c->cur ^= 1;
        c->since = 0;
        if (!tracked) c->work_full += 1.0;
        done++;
    }
    return hipSuccess;
}

// Tracked PAIRS on the slabs (round 6): `npairs` pairs of iterations from iteration `first`, every pair ONE list-driven fused pass per
// slab over its local rows (ghost rows included; rb_fused2d_kernel / tol_fused2d_tracked_kernel with the slab's own lists of that tiling:
// Slab::trk_f); check_last: the second iteration of the last pair is a check (the slabs' delta words, owned rows only).  A pass makes two
// more ghost rows stale, so the ghost rows are traded whenever the next pass would run out of exact ones (every halo / 2 passes), as a
// step of its own behind the passes: the neighbours' outermost owned rows are exact at any time, so an exchange may come early.  After
// an exchange the tiles that hold or read the rewritten rows -- two rows deep for two iterations -- are woken for the next pass.
// Until round 6 tracked relaxations on slabs ran single list-driven sweeps: the plugin's default 8192^2 relaxation on two slabs took
// 17 % longer than on one device.  Same iterations, same bits (tests/test_gpu_multi_device.py).
hipError_t multi_run_pairs(Ctx *c, unsigned npairs, unsigned first, bool check_last)
{
    const int G = c->halo;
    if (c->n != 2 || G < 2) return hipErrorInvalidValue;
    note_iterations(c, first, 2 * npairs);
    const bool tol = c->math == 4;
    const int rpt = rb_pairs_rows_per_task(c);
    const int nstrips_f = (c->pitch + 247) / 248;
    if (c->last_lists != 2)
        for (auto &sl : c->slabs) sl.trk_f.force = std::max(sl.trk_f.force, 1);   // something else has touched the field since
    c->last_lists = 2;
    // every slab's newest rows are in buf[c->cur]; its band event is recorded behind its latest pass
    auto exchange = [&]() -> hipError_t {
        const int cur = c->cur;
        hipError_t e = for_each_slab(c, [&, cur](int k) -> hipError_t {
            Ctx::Slab &sl = c->slabs[k];
            hipError_t e = hipStreamWaitEvent(sl.comm, sl.ev_band, 0);
            if (e == hipSuccess && k > 0) {
                Ctx::Slab &up = c->slabs[k - 1];
                e = multi_copy_units(c, sl, sl.buf[cur], 0, up, up.buf[cur], up.last() + 1 - G, G, up.ev_band, sl.peer_up, sl.bounce[0]);
            }
            if (e == hipSuccess && k + 1 < (int)c->slabs.size()) {
                Ctx::Slab &dn = c->slabs[k + 1];
                e = multi_copy_units(c, sl, sl.buf[cur], sl.rows - G, dn, dn.buf[cur], dn.first(), G, dn.ev_band, dn.peer_up, dn.bounce[1]);
            }
            if (e == hipSuccess) e = hipSetDevice(sl.dev);
            if (e == hipSuccess) e = hipEventRecord(sl.ev_comm, sl.comm);
            return e;
        });
        if (e != hipSuccess) return e;
        e = for_each_slab(c, [&](int k) -> hipError_t {
            Ctx::Slab &sl = c->slabs[k];
            hipError_t e = hipStreamWaitEvent(sl.stream, sl.ev_comm, 0);
            if (e == hipSuccess && k > 0) e = hipStreamWaitEvent(sl.stream, c->slabs[k - 1].ev_comm, 0);
            if (e == hipSuccess && k + 1 < (int)c->slabs.size()) e = hipStreamWaitEvent(sl.stream, c->slabs[k + 1].ev_comm, 0);
            if (e == hipSuccess && sl.trk_f.tiles && sl.trk_f.force == 0) {
                const epic_hip::Activity next = sl.trk_f.upcoming();
                auto wake_rows = [&](int lo, int hi) {   // tiles of the fused tiling that hold rows [lo, hi)
                    lo = std::max(lo, 0);
                    hi = std::min(hi, sl.rows);
                    if (hi <= lo) return hipSuccess;
                    return epic_hip::launch_wake_tile_range(&next, sl.trk_f.tiles, (lo / sl.trk_f.rpt) * nstrips_f, ((hi - 1) / sl.trk_f.rpt + 1) * nstrips_f, sl.stream);
                };
                if (sl.g_top) e = wake_rows(0, G + 2);
                if (e == hipSuccess && sl.g_bot) e = wake_rows(sl.rows - G - 2, sl.rows);
            }
            return e;
        });
        c->since = 0;
        return e;
    };
    // (an exchange needs every slab's band event behind its latest launch: single sweeps of an earlier call record it only in their
    //  exchange iteration, so the first exchange of this call records it itself)
    auto record_bands = [&]() -> hipError_t {
        return for_each_slab(c, [&](int k) -> hipError_t { return hipEventRecord(c->slabs[k].ev_band, c->slabs[k].stream); });
    };
    unsigned p = 0;
    bool bands_recorded = false;
    while (p < npairs) {
        if (c->since + 2 > G) {
            hipError_t e = bands_recorded ? hipSuccess : record_bands();
            if (e == hipSuccess) e = exchange();
            if (e != hipSuccess) return e;
        }
        const unsigned n = std::min<unsigned>(npairs - p, (unsigned)((G - c->since) / 2));
        const unsigned it0 = first + 2 * p;
        const int cur0 = c->cur;
        const bool last_stretch = p + n == npairs;
        hipError_t e = for_each_slab(c, [&, n, it0, cur0, last_stretch](int k) -> hipError_t {
            Ctx::Slab &sl = c->slabs[k];
            const size_t tiles = epic_hip::rb_fused_2d_tiles(sl.rows, c->pitch, rpt);
            int cur = cur0;
            hipError_t e = hipSuccess;
            for (unsigned i = 0; i < n && e == hipSuccess; i++) {
                const bool check = check_last && last_stretch && i + 1 == n;
                if (check) e = hipMemsetAsync(sl.d_delta, 0, sizeof(unsigned), sl.stream);
                if (e != hipSuccess) break;
                epic_hip::Activity act = sl.trk_f.next(tiles, rpt, sl.stream, nullptr);
                const int parity = (int)((it0 + 2 * i + (unsigned)sl.top()) & 1u);
                e = tol ? epic_hip::launch_jacobi_fused_2d(sl.buf[cur], sl.buf[cur ^ 1], sl.maskw, sl.rows, c->pitch, rpt, c->math, sl.stream,
                                                           c->redblack ? parity : -1, c->maskf(sl), act.list_out ? &act : nullptr,
                                                           check ? sl.d_delta : nullptr, sl.first(), sl.last() + 1)
                        : epic_hip::launch_rb_fused_2d(sl.buf[cur], sl.buf[cur ^ 1], sl.maskw, sl.rows, c->pitch, rpt, c->math, parity, sl.stream,
                                                       c->maskf(sl), act.list_out ? &act : nullptr, check ? sl.d_delta : nullptr, sl.first(), sl.last() + 1);
                if (e == hipSuccess && act.list_out) sl.trk_f.advance();
                cur ^= 1;
            }
            if (e == hipSuccess) e = hipEventRecord(sl.ev_band, sl.stream);   // (what the next exchange waits for)
            return e;
        });
        if (e != hipSuccess) return e;
        bands_recorded = true;
        if (n & 1u) c->cur ^= 1;
        c->since += 2 * (int)n;
        p += n;
    }
    // multi_run's invariant -- at least one exact ghost row at the start of every iteration, since <= G - 1 -- holds on return: whatever
    // comes next may be a single sweep (an odd count, a check of its own, a batch without the lists).  (First version of round 6 returned
    // with since == G after G / 2 passes: the next single sweep then read ghost rows that were all stale -- found by
    // tests/test_gpu_multi_device.py::test_tracked_pairs_on_slabs_maps_equal_the_reference, umass at eps = 1e-3 on three slabs.)
    if (c->since >= G) return exchange();
    return hipSuccess;
}

hipError_t multi_sweep(Ctx *c, bool check, unsigned iteration) { return multi_run(c, 1, iteration, check); }

int multi_read_delta(Harmonic *h, Ctx *c, const char *fn)
{
    DeviceGuard g;
    for (auto &sl : c->slabs) {
        if (hipSetDevice(sl.dev) != hipSuccess ||
            hipStreamSynchronize(sl.comm) != hipSuccess ||  // the boundary bands of a check iteration ran there
            hipMemcpyAsync(sl.h_delta, sl.d_delta, sizeof(float), hipMemcpyDeviceToHost, sl.stream) != hipSuccess) {
            report(fn, "Failed to copy memory from device to host for the max delta.");
            return EPIC_ERROR_MEMCPY_TO_HOST;
        }
    }
    float d = 0.0f;
    for (auto &sl : c->slabs) {
        if (hipSetDevice(sl.dev) != hipSuccess || hipStreamSynchronize(sl.stream) != hipSuccess) {
            report(fn, "Failed to synchronize the device after the 'update and check' kernel.");
            return EPIC_ERROR_DEVICE_SYNCHRONIZE;
        }
        d = std::max(d, *sl.h_delta);  // the host-side max of the per-device words
    }
    h->delta = d;
    return EPIC_SUCCESS;
}

int multi_get_values(Harmonic *h, Ctx *c, const char *fn)
{
    DeviceGuard g;
    multi_sync(c);
    for (auto &sl : c->slabs) {
        if (hipSetDevice(sl.dev) != hipSuccess ||
            hipMemcpy2D(h->u + (size_t)sl.lo * unit_rows(c) * c->cols, (size_t)c->cols * sizeof(float),
                        sl.buf[c->cur] + (size_t)sl.first() * unit_floats(c), (size_t)c->pitch * sizeof(float),
                        (size_t)c->cols * sizeof(float), (size_t)(sl.hi - sl.lo) * unit_rows(c), hipMemcpyDeviceToHost) != hipSuccess) {
            report(fn, "Failed to copy memory from device to host for the potential values.");
            return EPIC_ERROR_MEMCPY_TO_HOST;
        }
    }
    return EPIC_SUCCESS;
}

// harmonic_utilities_set_cells_2d_gpu on the slabs: every slab applies the edits that fall into its local rows -- owned
// AND ghost rows, so that neighbours agree without an exchange.
int multi_set_cells(Ctx *c, unsigned k, const unsigned *v, const unsigned *types, const char *fn)
{
    DeviceGuard g;
    multi_sync(c);
    int rc = EPIC_SUCCESS;
    for (auto &sl : c->slabs) {
        unsigned *d_v = nullptr, *d_types = nullptr;
        sl.trk.force = 2;
        sl.trk_f.force = std::max(sl.trk_f.force, 1);
        if (hipSetDevice(sl.dev) != hipSuccess || hipMalloc((void **)&d_v, 2 * (size_t)k * sizeof(unsigned)) != hipSuccess ||
            hipMalloc((void **)&d_types, (size_t)k * sizeof(unsigned)) != hipSuccess) {
            (void)hipGetLastError();
            report(fn, "Failed to allocate device-side memory for the cell locations and types.");
            rc = EPIC_ERROR_DEVICE_MALLOC;
        } else if (hipMemcpyAsync(d_v, v, 2 * (size_t)k * sizeof(unsigned), hipMemcpyHostToDevice, sl.stream) != hipSuccess ||
                   hipMemcpyAsync(d_types, types, (size_t)k * sizeof(unsigned), hipMemcpyHostToDevice, sl.stream) != hipSuccess) {
            report(fn, "Failed to copy memory from host to device for the cell locations and types.");
            rc = EPIC_ERROR_MEMCPY_TO_DEVICE;
        } else if (epic_hip::launch_set_cells_2d(sl.buf[c->cur], sl.maskw, sl.rows, c->cols, c->pitch, k, d_v, d_types, sl.stream,
                                                 sl.top(), c->rows, sl.g_top > 0, sl.g_bot > 0) != hipSuccess ||
                   epic_hip::launch_fuse_masks_2d(sl.maskw, sl.rows, c->pitch, c->maskf(sl), sl.stream) != hipSuccess) {
            report(fn, "Failed to execute the 'set cells' kernel.");
            rc = EPIC_ERROR_KERNEL_EXECUTION;
        }
        if (hipStreamSynchronize(sl.stream) != hipSuccess && rc == EPIC_SUCCESS) rc = EPIC_ERROR_DEVICE_SYNCHRONIZE;
        if (d_v) (void)hipFree(d_v);
        if (d_types) (void)hipFree(d_types);
        if (rc != EPIC_SUCCESS) break;
    }
    return rc;
}

bool multi_has_threads(const Ctx *c) { return c->crew && !c->crew->threads.empty(); }

}  // namespace epic_drv

