#include "driver.h"

using namespace epic_drv;

// ---------------------------------------------------------------------------------------------------------
// extension entry points (include/epic_hip.h)
// ---------------------------------------------------------------------------------------------------------
extern "C" {

const char *epic_hip_version(void) { return "epic-hip 0.2.0 gfx950"; }

int epic_hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int epic_hip_update_n_gpu(Harmonic *harmonic, unsigned int sweeps, int check_last)
{
    static const char *fn = "epic_hip_update_n_gpu";
    Ctx *c = harmonic ? find_ctx(harmonic) : nullptr;
    if (!harmonic || !ready(harmonic, c) || (check_last && !has_delta(c))) {
        report(fn, "Invalid data.");
        return EPIC_ERROR_INVALID_DATA;
    }
    if (const int frc = flush_pending(harmonic, c, fn)) return frc;
    if (sweeps == 0) return EPIC_SUCCESS;
    // one block of harmonic_execute_gpu's loop (driver_loop.hip: run_block -- tiles with the check folded in, tracked pairs, fused
    // passes, a captured graph), decided from the context's state as that loop decides it
    const bool check = check_last != 0;
    const unsigned plain = sweeps - (check ? 1u : 0u);
    const int rc = run_block(harmonic, c, plain, harmonic->currentIteration, check, fn, -1);
    if (rc != EPIC_SUCCESS) return rc;
    harmonic->currentIteration += sweeps;
    return check && harmonic->delta < harmonic->epsilon ? EPIC_SUCCESS_AND_CONVERGED : EPIC_SUCCESS;
}

int epic_hip_timed_sweeps_gpu(Harmonic *harmonic, unsigned int sweeps, unsigned int check_every, float *elapsed_ms)
{
    static const char *fn = "epic_hip_timed_sweeps_gpu";
    Ctx *c = harmonic ? find_ctx(harmonic) : nullptr;
    if (!harmonic || !elapsed_ms || !ready(harmonic, c) || (check_every && !has_delta(c))) {
        report(fn, "Invalid data.");
        return EPIC_ERROR_INVALID_DATA;
    }
    if (const int frc = flush_pending(harmonic, c, fn)) return frc;
    if (c->multi()) {
        // several devices: no single stream sees the whole batch, so the batch is bracketed by host clocks around
        // "every stream of every slab idle" (the batches this is used for run for milliseconds to seconds)
        { DeviceGuard g; multi_sync(c); }
        const auto t0 = std::chrono::steady_clock::now();
        int rc = EPIC_SUCCESS;
        bool checked = false;
        unsigned done = 0;
        while (done < sweeps && rc == EPIC_SUCCESS) {
            const unsigned it = harmonic->currentIteration;
            if (check_every && it % check_every == 0) {
                if (enqueue_check_sweep(c, it) != hipSuccess) rc = EPIC_ERROR_KERNEL_EXECUTION;
                checked = true;
                harmonic->currentIteration++;
                done++;
            } else {
                unsigned run = sweeps - done;
                if (check_every) run = std::min(run, check_every - it % check_every);
                if (enqueue_plain_run(c, run, it) != hipSuccess) rc = EPIC_ERROR_KERNEL_EXECUTION;
                harmonic->currentIteration += run;
                done += run;
            }
        }
        if (rc != EPIC_SUCCESS) report(fn, "Failed to execute the 'update' kernel.");
        harmonic->d_u = current_u(c);
        if (rc == EPIC_SUCCESS && checked) rc = read_delta(harmonic, c, fn);
        { DeviceGuard g; multi_sync(c); }
        *elapsed_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
        return rc;
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
        (void)hipGetLastError();
        if (e0) (void)hipEventDestroy(e0);
        return EPIC_ERROR_DEVICE_MALLOC;
    }
    int rc = EPIC_SUCCESS;
    (void)hipEventRecord(e0, c->stream);
    bool checked = false;
    unsigned done = 0;
    while (done < sweeps && rc == EPIC_SUCCESS) {
        const unsigned it = harmonic->currentIteration;
        if (check_every && it % check_every == 0) {
            if (enqueue_check_sweep(c, it) != hipSuccess) rc = EPIC_ERROR_KERNEL_EXECUTION;
            checked = true;
            harmonic->currentIteration++;
            done++;
        } else {
            unsigned run = sweeps - done;
            if (check_every) run = std::min(run, check_every - it % check_every);
            if (enqueue_plain_run(c, run, it) != hipSuccess) rc = EPIC_ERROR_KERNEL_EXECUTION;
            harmonic->currentIteration += run;
            done += run;
        }
    }
    if (rc != EPIC_SUCCESS) report(fn, "Failed to execute the 'update' kernel.");
    (void)hipEventRecord(e1, c->stream);
    harmonic->d_u = current_u(c);
    if (rc == EPIC_SUCCESS) {
        if (checked) rc = read_delta(harmonic, c, fn);  // the most recent check sweep's delta
        if (hipEventSynchronize(e1) != hipSuccess) rc = EPIC_ERROR_DEVICE_SYNCHRONIZE;
        else (void)hipEventElapsedTime(elapsed_ms, e0, e1);
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}

int epic_hip_iterations_per_pass(Harmonic *harmonic)
{
    Ctx *c = harmonic ? find_ctx(harmonic) : nullptr;
    if (!c) return 0;
    if (fuses_jacobi(c) || fuses_rb_tol(c)) return 2;
    return fuses_rb_precise(c) ? 2 : 1;
}

int epic_hip_tile_iterations(Harmonic *harmonic)
{
    Ctx *c = harmonic ? find_ctx(harmonic) : nullptr;
    return c ? tile_plan(c).halo : 0;
}

unsigned int epic_hip_finish_iteration(Harmonic *harmonic)
{
    Ctx *c = harmonic ? find_ctx(harmonic) : nullptr;
    return c ? c->finish_from : 0u;
}

int epic_hip_fused_rows_per_task(Harmonic *harmonic)
{
    Ctx *c = harmonic ? find_ctx(harmonic) : nullptr;
    if (!c || epic_hip_iterations_per_pass(harmonic) != 2) return 0;
    return (fuses_jacobi(c) || fuses_rb_tol(c)) ? jacobi_fused_rows_per_task(c) : fused_rows_per_task(c);
}

int epic_hip_set_rows_per_task(Harmonic *harmonic, unsigned int rows_per_task)
{
    Ctx *c = harmonic ? find_ctx(harmonic) : nullptr;
    if (!c || rows_per_task > 65536) return EPIC_ERROR_INVALID_DATA;
    if (const int frc = flush_pending(harmonic, c, "epic_hip_set_rows_per_task")) return frc;
    c->rows_per_task = (int)rows_per_task;
    c->tuned_rows[0] = c->tuned_rows[1] = c->tuned_rows[2] = 0;   // back to "not measured": a height of 0 here means automatic again
    force_all(c);
    return EPIC_SUCCESS;
}

int epic_hip_set_math_mode(Harmonic *harmonic, int mode)
{
    Ctx *c = harmonic ? find_ctx(harmonic) : nullptr;
    if (!c || mode < 0 || mode > 4 || mode == 3) return EPIC_ERROR_INVALID_DATA;  // 2 = traffic-only diagnostic (2-D), 4 = tol; 3 was round 1's df32
    if (const int frc = flush_pending(harmonic, c, "epic_hip_set_math_mode")) return frc;   // iterations counted so far run in the mode they were asked for in
    if (c->math != mode) c->tuned_rows[0] = c->tuned_rows[1] = c->tuned_rows[2] = 0;   // (kind 2 serves precise and fast: measured per arithmetic)
    c->math = mode;
    force_all(c);
    return EPIC_SUCCESS;
}

int epic_hip_set_scheme(Harmonic *harmonic, int scheme)
{
    Ctx *c = harmonic ? find_ctx(harmonic) : nullptr;
    if (!c || (scheme != 0 && scheme != 1)) return EPIC_ERROR_INVALID_DATA;
    if (const int frc = flush_pending(harmonic, c, "epic_hip_set_scheme")) return frc;
    c->redblack = scheme == 1;
    force_all(c);
    return EPIC_SUCCESS;
}

int epic_hip_set_activity_tracking(Harmonic *harmonic, int on)
{
    Ctx *c = harmonic ? find_ctx(harmonic) : nullptr;
    if (!c || on < 0 || on > 2) return EPIC_ERROR_INVALID_DATA;
    if (const int frc = flush_pending(harmonic, c, "epic_hip_set_activity_tracking")) return frc;
    c->track_mode = on;
    resolve_tracking(c);
    force_all(c);
    return EPIC_SUCCESS;
}

// Streamlines on the resident field.  starts: n_paths (x, y) pairs; k / rc: n_paths entries; paths: n_paths rows of
// 2 * maxLength floats (row i holds 2 * k[i] values).  All host pointers.
int epic_hip_compute_paths_2d_gpu(Harmonic *harmonic, unsigned int n_paths, const float *starts, float stepSize,
                                  float cdPrecision, unsigned int maxLength, unsigned int *k, int *rc_out, float *paths)
{
    static const char *fn = "epic_hip_compute_paths_2d_gpu";
    Ctx *c = harmonic ? find_ctx(harmonic) : nullptr;
    if (!harmonic || !ready(harmonic, c) || c->n != 2 || n_paths == 0 || !starts || !k || !rc_out || !paths) {
        report(fn, "Invalid data.");
        return EPIC_ERROR_INVALID_DATA;
    }
    if (c->multi()) {  // the field lives on several devices: read it back and walk it with harmonic_compute_path_2d_cpu
        report(fn, "Invalid data (not available in multi-device mode: use harmonic_get_potential_values_gpu and the CPU walk).");
        return EPIC_ERROR_INVALID_DATA;
    }
    if (const int frc = flush_pending(harmonic, c, fn)) return frc;   // the walk reads the field of every iteration asked for
    // the host walk stops at size() < 2u * maxLength values (unsigned product, harmonic_path_cpu.cpp:185)
    const unsigned max_points = (2u * maxLength) / 2u;
    const size_t row = 2 * (size_t)max_points;
    float *d_starts = nullptr, *d_pts = nullptr;
    unsigned *d_k = nullptr;
    int *d_rc = nullptr;
    int rc = EPIC_SUCCESS;
    if (hipMalloc((void **)&d_starts, 2 * (size_t)n_paths * sizeof(float)) != hipSuccess ||
        hipMalloc((void **)&d_pts, std::max<size_t>(row * n_paths, 1) * sizeof(float)) != hipSuccess ||
        hipMalloc((void **)&d_k, (size_t)n_paths * sizeof(unsigned)) != hipSuccess ||
        hipMalloc((void **)&d_rc, (size_t)n_paths * sizeof(int)) != hipSuccess) {
        (void)hipGetLastError();
        report(fn, "Failed to allocate device-side memory for the paths.");
        rc = EPIC_ERROR_DEVICE_MALLOC;
    } else if (hipMemcpyAsync(d_starts, starts, 2 * (size_t)n_paths * sizeof(float), hipMemcpyHostToDevice, c->stream) !=
               hipSuccess) {
        report(fn, "Failed to copy memory from host to device for the start points.");
        rc = EPIC_ERROR_MEMCPY_TO_DEVICE;
    } else if (epic_hip::launch_follow_paths_2d(c->buf[c->cur], c->maskw, c->rows, c->cols, c->pitch, n_paths, d_starts,
                                                stepSize, cdPrecision, max_points, d_pts, d_k, d_rc, c->stream) != hipSuccess) {
        report(fn, "Failed to execute the 'follow paths' kernel.");
        rc = EPIC_ERROR_KERNEL_EXECUTION;
    } else if (hipMemcpyAsync(k, d_k, (size_t)n_paths * sizeof(unsigned), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
               hipMemcpyAsync(rc_out, d_rc, (size_t)n_paths * sizeof(int), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
               hipStreamSynchronize(c->stream) != hipSuccess) {
        report(fn, "Failed to copy memory from device to host for the path lengths.");
        rc = EPIC_ERROR_MEMCPY_TO_HOST;
    } else {
        for (unsigned i = 0; i < n_paths; i++)  // only the way-points that exist come back
            if (k[i] > 0 && hipMemcpyAsync(paths + i * row, d_pts + i * row, 2 * (size_t)k[i] * sizeof(float),
                                           hipMemcpyDeviceToHost, c->stream) != hipSuccess)
                rc = EPIC_ERROR_MEMCPY_TO_HOST;
        if (hipStreamSynchronize(c->stream) != hipSuccess) rc = EPIC_ERROR_MEMCPY_TO_HOST;
        if (rc != EPIC_SUCCESS) report(fn, "Failed to copy memory from device to host for the paths.");
    }
    if (rc != EPIC_SUCCESS) (void)hipStreamSynchronize(c->stream);
    for (void *p : {(void *)d_starts, (void *)d_pts, (void *)d_k, (void *)d_rc})
        if (p) (void)hipFree(p);
    return rc;
}

// harmonic_compute_path_2d_cpu's contract (harmonic_path_cpu.cpp:154-221) on the resident field: *path must be null on
// entry and receives a new[] array of 2 * *k floats that harmonic_free_path_cpu (or delete[]) releases.
int epic_hip_compute_path_2d_gpu(Harmonic *harmonic, float x, float y, float stepSize, float cdPrecision,
                                 unsigned int maxLength, unsigned int *k, float **path)
{
    static const char *fn = "epic_hip_compute_path_2d_gpu";
    if (!harmonic || !k || !path || *path != nullptr) {
        report(fn, "Invalid data.");
        return EPIC_ERROR_INVALID_DATA;
    }
    const size_t row = 2 * (size_t)((2u * maxLength) / 2u);
    std::vector<float> pts(std::max<size_t>(row, 1));
    const float start[2] = {x, y};
    unsigned n = 0;
    int walk = EPIC_SUCCESS;
    int rc = epic_hip_compute_paths_2d_gpu(harmonic, 1, start, stepSize, cdPrecision, maxLength, &n, &walk, pts.data());
    if (rc != EPIC_SUCCESS) return rc;
    if (walk != EPIC_SUCCESS) {
        report(fn, walk == EPIC_ERROR_INVALID_LOCATION ? "Invalid location."
                   : walk == EPIC_ERROR_INVALID_GRADIENT ? "Could not compute gradient."
                                                         : "Could not compute a valid path.");
        return walk;
    }
    *k = n;
    *path = new float[2 * (size_t)n];
    std::copy(pts.begin(), pts.begin() + 2 * (size_t)n, *path);
    return EPIC_SUCCESS;
}

int epic_hip_activity_stats(Harmonic *harmonic, unsigned long long *active_tiles, unsigned long long *tiles)
{
    return epic_hip_activity_stats2(harmonic, active_tiles, nullptr, tiles);
}

int epic_hip_activity_stats2(Harmonic *harmonic, unsigned long long *active_tiles, unsigned long long *due_tiles,
                             unsigned long long *tiles)
{
    Ctx *c = harmonic ? find_ctx(harmonic) : nullptr;
    if (!c || !active_tiles || !tiles) return EPIC_ERROR_INVALID_DATA;
    if (const int frc = flush_pending(harmonic, c, "epic_hip_activity_stats")) return frc;
    *active_tiles = *tiles = 0;
    if (due_tiles) *due_tiles = 0;
    if (!c->track) return EPIC_SUCCESS;
    // the counter sets the next launches will consume were filled by the latest ones: the tiles they woke (summed over the
    // slabs in multi-device mode; a forced iteration runs every tile)
    if (!c->multi() && hipStreamSynchronize(c->stream) != hipSuccess) return EPIC_ERROR_DEVICE_SYNCHRONIZE;
    unsigned long long due = 0, all = 0;
    if (!epic_drv::due_tiles(c, &due, &all, true)) return EPIC_SUCCESS;
    *active_tiles = due;
    if (due_tiles) *due_tiles = due;
    *tiles = all;
    return EPIC_SUCCESS;
}

int epic_hip_work_done(Harmonic *harmonic, double *grid_iterations, int reset)
{
    Ctx *c = harmonic ? find_ctx(harmonic) : nullptr;
    if (!c || !grid_iterations) return EPIC_ERROR_INVALID_DATA;
    if (const int frc = flush_pending(harmonic, c, "epic_hip_work_done")) return frc;
    if (c->multi()) { DeviceGuard g; multi_sync(c); }
    fold_listed_work(c);
    *grid_iterations = c->work_full;
    if (reset) c->work_full = 0.0;
    return EPIC_SUCCESS;
}

int epic_hip_eval_math(const float *d_in, float *d_out, size_t n, int which, void *stream)
{
    if (!d_in || !d_out) return EPIC_ERROR_INVALID_DATA;
    return epic_hip::launch_eval_math(d_in, d_out, n, which, (hipStream_t)stream) == hipSuccess ? EPIC_SUCCESS
                                                                                                  : EPIC_ERROR_KERNEL_EXECUTION;
}

int epic_hip_get_layout(Harmonic *harmonic, unsigned int *pitch, size_t *u_bytes, size_t *mask_bytes)
{
    Ctx *c = harmonic ? find_ctx(harmonic) : nullptr;
    if (!c || c->pitch == 0) return EPIC_ERROR_INVALID_DATA;
    if (pitch) *pitch = (unsigned)c->pitch;
    if (u_bytes) *u_bytes = c->u_bytes();
    if (mask_bytes) *mask_bytes = c->mask_bytes();
    if (c->multi()) {  // per-device state, ghost rows included, summed
        size_t ub = 0, mb = 0;
        for (const auto &sl : c->slabs) {
            ub += (size_t)sl.rows * unit_floats(c) * sizeof(float);
            mb += sizeof(uint32_t) * slab_mask_words(c, sl);
        }
        if (u_bytes) *u_bytes = ub;
        if (mask_bytes) *mask_bytes = mb;
    }
    return EPIC_SUCCESS;
}

// Which device holds which rows (multi-device mode: one entry per slab; otherwise one entry, the whole grid on the current
// device).  Returns the number of slabs; fills at most `cap` entries of each non-null array.
// What the multi-device mode decided and how one exchange iteration actually ran, as one JSON object in `buf` (the first run on
// real hardware cannot be rehearsed, so it reports on itself): per seam the two devices, what hipDeviceCanAccessPeer says in
// both directions, whether peer access was enabled (transport "peer" / "staged" / "same-device"), link type and hop count where
// the runtime reports them; then ONE untracked exchange iteration with timing events -- per slab the interior sweep on the
// compute stream and the boundary bands + incoming halo copies on the second stream, in microseconds from the moment the
// earlier of the two starts: overlap_us = the time both were running, copies_hidden = the copies ended before the interior sweep
// did.  Advances currentIteration by up to `halo`
// iterations (a whole stretch up to and including an exchange).  Returns the bytes written (0: not in multi-device mode).
int epic_hip_multi_report(Harmonic *harmonic, char *buf, size_t cap)
{
    Ctx *c = harmonic ? find_ctx(harmonic) : nullptr;
    if (!c || !buf || cap < 64 || !c->multi() || !ready(harmonic, c)) return 0;
    if (flush_pending(harmonic, c, "epic_hip_multi_report") != EPIC_SUCCESS) return 0;   // (the probe below runs an iteration of its own)
    std::string out = "{";
    auto add = [&](const char *fmt, auto... a) { char t[256]; snprintf(t, sizeof t, fmt, a...); out += t; };
    add("\"slabs\": %d, \"halo\": %d, \"issuing_threads\": %s, \"seams\": [", (int)c->slabs.size(), c->halo,
        multi_has_threads(c) ? "true" : "false");
    DeviceGuard g;
    for (size_t k = 1; k < c->slabs.size(); k++) {
        const Ctx::Slab &up = c->slabs[k - 1], &sl = c->slabs[k];
        int can_du = -1, can_ud = -1;
        if (up.dev != sl.dev) {
            if (hipDeviceCanAccessPeer(&can_du, sl.dev, up.dev) != hipSuccess) { (void)hipGetLastError(); can_du = -1; }
            if (hipDeviceCanAccessPeer(&can_ud, up.dev, sl.dev) != hipSuccess) { (void)hipGetLastError(); can_ud = -1; }
        }
        uint32_t link = 0, hops = 0;
        const bool have_link = up.dev != sl.dev && hipExtGetLinkTypeAndHopCount(up.dev, sl.dev, &link, &hops) == hipSuccess;
        if (!have_link) (void)hipGetLastError();
        add("%s{\"upper_device\": %d, \"lower_device\": %d, \"can_access_peer\": [%d, %d], \"transport\": \"%s\", \"link_type\": %s, \"hops\": %s}",
            k > 1 ? ", " : "", up.dev, sl.dev, can_ud, can_du, up.dev == sl.dev ? "same-device" : sl.peer_up ? "peer" : "staged",
            have_link ? std::to_string(link).c_str() : "null", have_link ? std::to_string(hops).c_str() : "null");
    }
    out += "], \"link_type_legend\": \"hipExtGetLinkTypeAndHopCount: 1 HyperTransport, 2 QPI, 3 PCIe, 4 InfiniBand, 5 xGMI\", \"exchange\": [";
    // one stretch ending with an exchange, untracked, with the probe armed
    const bool track0 = c->track;
    c->track = false;
    multi_sync(c);
    c->probe.assign(c->slabs.size(), Ctx::Probe{});
    bool ok = true;
    for (size_t k = 0; k < c->slabs.size() && ok; k++) {
        ok = hipSetDevice(c->slabs[k].dev) == hipSuccess && hipEventCreate(&c->probe[k].int0) == hipSuccess && hipEventCreate(&c->probe[k].int1) == hipSuccess &&
             hipEventCreate(&c->probe[k].cp0) == hipSuccess && hipEventCreate(&c->probe[k].cp1) == hipSuccess;
    }
    const unsigned n = (unsigned)std::max(1, c->halo - c->since);
    if (ok) ok = multi_run(c, n, harmonic->currentIteration, false) == hipSuccess;
    if (ok) {
        harmonic->currentIteration += n;
        harmonic->d_u = current_u(c);
        multi_sync(c);
        for (size_t k = 0; k < c->slabs.size(); k++) {
            const Ctx::Probe &p = c->probe[k];
            float i0 = 0, i1 = 0, c1 = 0;
            if (hipSetDevice(c->slabs[k].dev) != hipSuccess || hipEventElapsedTime(&i0, p.cp0, p.int0) != hipSuccess ||
                hipEventElapsedTime(&i1, p.cp0, p.int1) != hipSuccess || hipEventElapsedTime(&c1, p.cp0, p.cp1) != hipSuccess) {
                (void)hipGetLastError();
                add("%s{\"slab\": %d, \"error\": \"no timing\"}", k ? ", " : "", (int)k);
                continue;
            }
            // both intervals from the earlier of the two starts (the two streams of a slab start independently)
            const float base = std::min(0.0f, i0), is = (i0 - base) * 1e3f, ie = (i1 - base) * 1e3f, cs = (0.0f - base) * 1e3f, ce = (c1 - base) * 1e3f;
            const float both = std::max(0.0f, std::min(ie, ce) - std::max(is, cs));
            add("%s{\"slab\": %d, \"device\": %d, \"interior_us\": [%.1f, %.1f], \"bands_and_copies_us\": [%.1f, %.1f], \"overlap_us\": %.1f, \"copies_hidden\": %s}",
                k ? ", " : "", (int)k, c->slabs[k].dev, is, ie, cs, ce, both, ce <= ie ? "true" : "false");
        }
    }
    for (auto &p : c->probe)
        for (hipEvent_t e : {p.int0, p.int1, p.cp0, p.cp1})
            if (e) (void)hipEventDestroy(e);
    c->probe.clear();
    c->track = track0;
    force_all(c);
    out += ok ? "]}" : "], \"error\": \"the probed exchange failed\"}";
    if (out.size() + 1 > cap) return 0;
    memcpy(buf, out.c_str(), out.size() + 1);
    return (int)out.size();
}

int epic_hip_device_layout(Harmonic *harmonic, int cap, int *devices, unsigned int *row_begin, unsigned int *row_end,
                           unsigned int *ghost_rows)
{
    Ctx *c = harmonic ? find_ctx(harmonic) : nullptr;
    if (!c || c->pitch == 0) return 0;
    if (!c->multi()) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (cap > 0) {
            if (devices) devices[0] = dev;
            if (row_begin) row_begin[0] = 0;
            if (row_end) row_end[0] = (unsigned)c->rows;
            if (ghost_rows) ghost_rows[0] = 0;
        }
        return 1;
    }
    for (int k = 0; k < (int)c->slabs.size() && k < cap; k++) {
        if (devices) devices[k] = c->slabs[k].dev;
        if (row_begin) row_begin[k] = (unsigned)c->slabs[k].lo;
        if (row_end) row_end[k] = (unsigned)c->slabs[k].hi;
        if (ghost_rows) ghost_rows[k] = (unsigned)c->halo;
    }
    return (int)c->slabs.size();
}

size_t epic_hip_mask_words_2d(unsigned int rows, unsigned int pitch) { return epic_hip::mask_words_2d((int)rows, (int)pitch); }
unsigned int epic_hip_pitch_for_cols(unsigned int cols) { return (unsigned)epic_hip::pitch_for_cols((int)cols); }

int epic_hip_pack_mask_2d(const uint32_t *d_locked, unsigned int rows, unsigned int cols, unsigned int pitch,
                          int ghost_top, int ghost_bottom, uint32_t *d_maskw, void *stream)
{
    if (!d_locked || !d_maskw || rows < 3 || cols < 3 || pitch < cols || pitch % 256 != 0) return EPIC_ERROR_INVALID_DATA;
    return epic_hip::launch_pack_mask_2d(d_locked, (int)rows, (int)cols, (int)pitch, ghost_top, ghost_bottom, d_maskw,
                                         (hipStream_t)stream) == hipSuccess
               ? EPIC_SUCCESS
               : EPIC_ERROR_KERNEL_EXECUTION;
}

int epic_hip_sweep_2d(const float *d_in, float *d_out, const uint32_t *d_maskw, unsigned int rows, unsigned int pitch,
                      unsigned int row_begin, unsigned int row_end, unsigned int rows_per_task, int math_mode,
                      uint32_t *d_delta_bits, void *stream)
{
    if (!d_in || !d_out || !d_maskw || d_in == d_out || rows < 3 || pitch % 256 != 0 || pitch == 0 || row_end > rows ||
        row_begin > row_end)
        return EPIC_ERROR_INVALID_DATA;
    if (rows_per_task == 0) rows_per_task = 32;
    if (math_mode < 0 || math_mode > 4 || math_mode == 3) return EPIC_ERROR_INVALID_DATA;
    return epic_hip::launch_sweep_2d(d_in, d_out, d_maskw, (int)rows, (int)pitch, (int)row_begin, (int)row_end,
                                     (int)rows_per_task, math_mode, -1, d_delta_bits, (hipStream_t)stream) == hipSuccess
               ? EPIC_SUCCESS
               : EPIC_ERROR_KERNEL_EXECUTION;
}

int epic_hip_sweep2_2d(const float *d_in, float *d_out, const uint32_t *d_maskw, unsigned int rows, unsigned int pitch,
                       unsigned int rows_per_task, int math_mode, void *stream)
{
    if (!d_in || !d_out || !d_maskw || d_in == d_out || rows < 3 || pitch % 256 != 0 || pitch == 0) return EPIC_ERROR_INVALID_DATA;
    if (math_mode != 4) return EPIC_ERROR_INVALID_DATA;  // the fused pass exists for the tol arithmetic
    if (rows_per_task == 0) rows_per_task = (unsigned)epic_hip::jacobi_fused_auto_rows((int)rows, (int)pitch);
    return epic_hip::launch_jacobi_fused_2d(d_in, d_out, d_maskw, (int)rows, (int)pitch, (int)rows_per_task, math_mode,
                                            (hipStream_t)stream) == hipSuccess
               ? EPIC_SUCCESS
               : EPIC_ERROR_KERNEL_EXECUTION;
}

size_t epic_hip_mask_words_fused_2d(unsigned int rows, unsigned int pitch) { return epic_hip::mask_words_fused_2d((int)rows, (int)pitch); }

int epic_hip_fuse_masks_2d(const uint32_t *d_maskw, unsigned int rows, unsigned int pitch, uint32_t *d_maskf, void *stream)
{
    if (!d_maskw || !d_maskf || rows < 3 || pitch == 0 || pitch % 256 != 0) return EPIC_ERROR_INVALID_DATA;
    return epic_hip::launch_fuse_masks_2d(d_maskw, (int)rows, (int)pitch, d_maskf, (hipStream_t)stream) == hipSuccess
               ? EPIC_SUCCESS
               : EPIC_ERROR_KERNEL_EXECUTION;
}

int epic_hip_sweeps_2d(float *d_a, float *d_b, const uint32_t *d_maskw, const uint32_t *d_maskf, unsigned int rows,
                       unsigned int pitch, unsigned int n, unsigned int rows_per_task, unsigned int rows_per_pair, int math_mode,
                       int *flips, void *stream)
{
    if (!d_a || !d_b || d_a == d_b || !d_maskw || rows < 3 || pitch == 0 || pitch % 256 != 0) return EPIC_ERROR_INVALID_DATA;
    if (math_mode < 0 || math_mode > 4 || math_mode == 3) return EPIC_ERROR_INVALID_DATA;
    if (rows_per_task == 0) rows_per_task = 32;
    if (rows_per_pair == 0) rows_per_pair = (unsigned)epic_hip::jacobi_fused_auto_rows((int)rows, (int)pitch);
    const bool pairs = math_mode == 4 && !process_config().no_fuse;   // (no context here: the process-wide knobs)
    float *buf[2] = {d_a, d_b};
    int cur = 0;
    for (unsigned i = 0; i < n;) {
        hipError_t e;
        if (pairs && n - i >= 2) {
            e = epic_hip::launch_jacobi_fused_2d(buf[cur], buf[cur ^ 1], d_maskw, (int)rows, (int)pitch, (int)rows_per_pair, math_mode,
                                                 (hipStream_t)stream, -1, d_maskf);
            i += 2;
        } else {
            e = epic_hip::launch_sweep_2d(buf[cur], buf[cur ^ 1], d_maskw, (int)rows, (int)pitch, 0, (int)rows, (int)rows_per_task,
                                          math_mode, -1, nullptr, (hipStream_t)stream);
            i += 1;
        }
        if (e != hipSuccess) return EPIC_ERROR_KERNEL_EXECUTION;
        cur ^= 1;
    }
    if (flips) *flips = cur;
    return EPIC_SUCCESS;
}

int epic_hip_sweep_rb_2d(float *d_u, const uint32_t *d_maskw, unsigned int rows, unsigned int pitch, unsigned int row_begin,
                         unsigned int row_end, unsigned int rows_per_task, int math_mode, int parity,
                         uint32_t *d_delta_bits, void *stream)
{
    if (!d_u || !d_maskw || rows < 3 || pitch % 256 != 0 || pitch == 0 || row_end > rows || row_begin > row_end ||
        (parity != 0 && parity != 1))
        return EPIC_ERROR_INVALID_DATA;
    if (rows_per_task == 0) rows_per_task = 32;
    if (math_mode < 0 || math_mode > 4 || math_mode == 3) return EPIC_ERROR_INVALID_DATA;
    return epic_hip::launch_sweep_2d(d_u, d_u, d_maskw, (int)rows, (int)pitch, (int)row_begin, (int)row_end,
                                     (int)rows_per_task, math_mode, parity, d_delta_bits, (hipStream_t)stream) == hipSuccess
               ? EPIC_SUCCESS
               : EPIC_ERROR_KERNEL_EXECUTION;
}

// What a context was configured with and which path it is on: one JSON object in buf -- "config": every EPIC_HIP_* knob as read
// when the context was created (driver_config.h), "state": dimensions, the modes in force now (the setters change them), "path":
// the kernel family a batch of plain iterations takes, the tile plan, task heights, halo and transport per seam.  Returns the
// bytes written (0: no context for this Harmonic, or buf too small).
int epic_hip_config_dump(Harmonic *harmonic, char *buf, size_t cap)
{
    Ctx *c = harmonic ? find_ctx(harmonic) : nullptr;
    if (!c || !buf || cap < 64) return 0;
    std::string out = "{\"config\": " + c->cfg.json();
    auto add = [&](const char *fmt, auto... a) { char t[512]; snprintf(t, sizeof t, fmt, a...); out += t; };
    add(", \"state\": {\"n\": %d, \"m\": [%d, %d, %d, %d], \"rows\": %d, \"cols\": %d, \"pitch\": %d, \"math\": %d, \"scheme\": \"%s\", "
        "\"track_mode\": %d, \"tracking\": %s, \"rows_per_task\": %d, \"graphs_broken\": %s, \"slabs\": %d}",
        c->n, c->m[0], c->m[1], c->m[2], c->m[3], c->rows, c->cols, c->pitch, c->math, c->redblack ? "redblack" : "jacobi", c->track_mode,
        c->track ? "true" : "false", c->rows_per_task, c->graphs_broken ? "true" : "false", (int)c->slabs.size());
    add(", \"path\": {\"plain_batch\": \"%s\"", c->pitch > 0 ? plain_batch_path(c) : "no dimensions yet");
    if (c->pitch > 0 && c->n == 2) {
        const epic_hip::TilePlan tp = c->multi() ? epic_hip::TilePlan{0, 0, 0, 0, 0} : tile_plan(c);
        add(", \"tile_plan\": {\"iterations_per_launch\": %d, \"tile_rows\": %d, \"tile_cols\": %d, \"tiles\": [%d, %d]}", tp.halo, tp.tile_rows,
            tp.tile_cols, tp.tiles_r, tp.tiles_c);
        add(", \"sweep_rows_per_task\": %d, \"fused_rows_per_task\": %d, \"tracked_pair_rows\": %d", auto_rows_per_task(c),
            c->math == 4 ? jacobi_fused_rows_per_task(c) : fused_rows_per_task(c), rb_pairs_rows_per_task(c));
    }
    if (c->multi()) {
        add(", \"halo\": %d, \"issuing_threads\": %s, \"spin_us\": %d, \"seams\": [", c->halo, multi_has_threads(c) ? "true" : "false", c->cfg.spin_us);
        for (size_t k = 1; k < c->slabs.size(); k++)
            add("%s{\"devices\": [%d, %d], \"transport\": \"%s\"}", k > 1 ? ", " : "", c->slabs[k - 1].dev, c->slabs[k].dev,
                c->slabs[k - 1].dev == c->slabs[k].dev ? "same-device" : c->slabs[k].peer_up ? "peer" : "staged");
        out += "]";
    }
    out += "}}";
    if (out.size() + 1 > cap) return 0;
    memcpy(buf, out.c_str(), out.size() + 1);
    return (int)out.size();
}

// Re-reads the environment: for this Harmonic's context (its Config and the mode fields derived from it -- math, scheme, tracking,
// task height; the device list and halo depth are taken over when no field or mask is resident, i.e. they shape the NEXT
// initialisation) AND the process-wide knobs; with NULL, the process-wide knobs only (what the raw operators and the 2-D launchers'
// EPIC_HIP_FLAGS / EPIC_HIP_LIST_WAVES go by).  The library reads the environment ONCE per context; a caller (a test, bench.py) that changes
// a variable on a live context says so with this call.
int epic_hip_config_reload(Harmonic *harmonic)
{
    if (!harmonic) {
        reload_process_config();
        return EPIC_SUCCESS;
    }
    Ctx *c = find_ctx(harmonic);
    if (!c) return EPIC_ERROR_INVALID_DATA;
    if (const int frc = flush_pending(harmonic, c, "epic_hip_config_reload")) return frc;   // iterations counted so far run under the knobs they were asked for under
    const Config before = c->cfg;
    c->cfg = Config::from_env();
    // a mode set through epic_hip_set_* stays unless ITS variable changed
    if (c->cfg.math != before.math) { c->math = c->cfg.math; c->tuned_rows[0] = c->tuned_rows[1] = c->tuned_rows[2] = 0; }
    if (c->cfg.redblack != before.redblack) c->redblack = c->cfg.redblack;
    if (c->cfg.track_mode != before.track_mode) c->track_mode = c->cfg.track_mode;
    if (c->cfg.rows_per_task != before.rows_per_task) { c->rows_per_task = c->cfg.rows_per_task; c->tuned_rows[0] = c->tuned_rows[1] = c->tuned_rows[2] = 0; }
    if (c->cfg.fused_rows != before.fused_rows) c->tuned_rows[0] = c->tuned_rows[1] = c->tuned_rows[2] = 0;
    // The device list, the halo depth and the transport belong to the slab layout: they are taken over now if no field or mask is
    // resident (the next initialisation lays the slabs out from them), otherwise the layout in force stays until the caller has
    // uninitialised both and initialises again (config_dump shows the layout in force under "state").
    if (!c->buf[0] && !c->maskw && !multi_holds_anything(c)) {
        const int math = c->math, track_mode = c->track_mode, rpt = c->rows_per_task;
        const bool rb = c->redblack;
        if (c->multi()) multi_destroy(c);
        apply_config(c);   // devices (validated) and halo; the mode fields are put back: only a variable that CHANGED moves them (above)
        c->math = math; c->track_mode = track_mode; c->rows_per_task = rpt; c->redblack = rb;
    }
    // EPIC_HIP_FLAGS and EPIC_HIP_LIST_WAVES are read by the 2-D launchers from the process-wide knobs (kernels_2d.hip: speed only,
    // never results): a reload on behalf of one context refreshes those too, so that what config_dump prints is what the launches use
    reload_process_config();
    if (c->pitch > 0) resolve_tracking(c);
    drop_graphs(c);           // captured sequences were made under the old knobs
    force_all(c);
    return EPIC_SUCCESS;
}

}  // extern "C"

