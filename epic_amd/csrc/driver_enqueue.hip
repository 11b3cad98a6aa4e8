// driver_enqueue.hip -- iterations enqueued on a context's stream(s): single sweeps, batches of plain iterations (fused passes, LDS
// tiles, captured hipGraphs), tracked pairs; the readback of max |du|; work-list bookkeeping.
#include "driver.h"

namespace epic_drv {

// Iterations [first, first + count) are about to be enqueued.  With work lists and the red-black scheme the lists in force were made for
// the colour of iteration seq_next; another colour (the caller has renumbered its iterations) makes them void: every tile runs.
static void expect_iteration(Ctx *c, unsigned first)
{
    if (c->track && c->redblack && c->seq_valid && ((first ^ c->seq_next) & 1u)) {
        force_all(c);
        c->seq_valid = false;
    }
}
void note_iterations(Ctx *c, unsigned first, unsigned count)
{
    expect_iteration(c, first);
    c->seq_next = first + count;
    c->seq_valid = true;
}

hipError_t enqueue_sweep(Ctx *c, bool check, unsigned iteration)
{
    if (c->n == 4) return hipSuccess;   // the reference's empty n == 4 branch: nothing is swept, the caller counts
    if (c->multi()) return multi_sweep(c, check, iteration);
    note_iterations(c, iteration, 1);
    hipError_t e;
    if (check) {
        e = hipMemsetAsync(c->d_delta, 0, sizeof(unsigned), c->stream);
        if (e != hipSuccess) return e;
    }
    const float *in = c->buf[c->cur];
    float *out = c->buf[c->cur ^ 1];
    // wake lists of this iteration (2-D only); (re)allocated when the tiling changes
    epic_hip::Activity act = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (c->track) {
        const int rpt = c->n == 2 ? auto_rows_per_task(c) : 32;   // the 3-D kernel has a fixed task shape
        const size_t tiles = c->n == 2 ? epic_hip::sweep_2d_tiles(c->rows, c->pitch, rpt)
                                       : epic_hip::sweep_3d_tiles(c->m[0], c->m[1], c->pitch);
        if (tiles != c->trk.tiles || rpt != c->trk.rpt) {
            drop_graphs(c);       // captured sequences hold the old lists (never reached during a capture: force > 0)
            fold_listed_work(c);  // the sum kept in the old block
        }
        if (c->last_lists == 2) c->trk.force = std::max(c->trk.force, 1);   // fused passes have run since these lists were made
        act = c->trk.next(tiles, rpt, c->stream, nullptr);
        c->last_lists = act.list_out ? 1 : 0;
    }
    if (!act.list_in) c->work_full += 1.0;  // every tile runs (untracked, or a forced iteration of a tracked run)
    auto advance = [&](hipError_t e) {
        if (e == hipSuccess && act.list_out) c->trk.advance();
        return e;
    };
    if (c->redblack) {
        float *inout = c->buf[c->cur];
        if (c->n == 2)
            return advance(epic_hip::launch_sweep_2d(inout, inout, c->maskw, c->rows, c->pitch, 0, c->rows,
                                                     auto_rows_per_task(c), c->math, (int)(iteration & 1u),
                                                     check ? c->d_delta : nullptr, c->stream, &act));
        return advance(epic_hip::launch_sweep_3d(inout, inout, c->maskw, c->m[0], c->m[1], c->pitch, 0, c->m[0], c->math,
                                                 (int)(iteration & 1u), check ? c->d_delta : nullptr, c->stream, &act, -1, -1, &c->cfg.launch));
    }
    if (c->n == 2)
        e = advance(epic_hip::launch_sweep_2d(in, out, c->maskw, c->rows, c->pitch, 0, c->rows, auto_rows_per_task(c),
                                              c->math, -1, check ? c->d_delta : nullptr, c->stream, &act));
    else
        e = advance(epic_hip::launch_sweep_3d(in, out, c->maskw, c->m[0], c->m[1], c->pitch, 0, c->m[0], c->math, -1,
                                              check ? c->d_delta : nullptr, c->stream, &act, -1, -1, &c->cfg.launch));
    if (e == hipSuccess) c->cur ^= 1;
    return e;
}

// A check iteration as the context's scheme has it: the scheme's own sweep, or -- EPIC_HIP_JACOBI_CHECKS=reference on a context that runs Jacobi
// sweeps (driver_loop.hip: run_block says why) -- the reference's half-sweep of that iteration's colour, in place in the current buffer.
hipError_t enqueue_check_sweep(Ctx *c, unsigned iteration)
{
    if (!jacobi_reference_checks(c)) return enqueue_sweep(c, true, iteration);
    c->redblack = true;
    force_all(c);   // the lists in force were made by Jacobi sweeps
    const hipError_t e = enqueue_sweep(c, true, iteration);
    c->redblack = false;
    force_all(c);
    return e;
}

// Adds what the device has summed for the list-driven launches (in tiles) to the host's count and clears it; waits for
// the stream.
void fold_listed_work(Ctx *c)
{
    auto fold = [&](Track &t, hipStream_t stream, double share) {
        if (!t.wake || t.tiles == 0) return;
        unsigned long long n = 0;
        if (hipStreamSynchronize(stream) == hipSuccess && hipMemcpy(&n, t.total(), sizeof n, hipMemcpyDeviceToHost) == hipSuccess &&
            hipMemset(t.total(), 0, sizeof n) == hipSuccess)
            c->work_full += share * (double)n / (double)t.tiles;
        else
            (void)hipGetLastError();
    };
    if (!c->multi()) {
        fold(c->trk, c->stream, 1.0);
        fold(c->trk_f, c->stream, 2.0);   // a listed tile of a fused pass is recomputed twice
        return;
    }
    DeviceGuard g;
    const int units = c->n == 2 ? c->rows : c->m[0];
    for (auto &sl : c->slabs)   // a slab's lists cover its ghost rows too: weighted by its share of the grid
        if (hipSetDevice(sl.dev) == hipSuccess) {
            fold(sl.trk, sl.stream, (double)sl.rows / (double)units);
            fold(sl.trk_f, sl.stream, 2.0 * (double)sl.rows / (double)units);   // a listed tile of a fused pass is recomputed twice
        }
}

void drop_graphs(Ctx *c)
{
    for (auto &g : c->graphs) (void)hipGraphExecDestroy(g.second.exec);
    c->graphs.clear();
}

// check_last (tile path only, tile_checks()): one more iteration after the `count` plain ones, a check, in the same launches.
hipError_t enqueue_plain_run(Ctx *c, unsigned count, unsigned first, bool check_last)
{
    if (c->n == 4) return check_last ? hipErrorInvalidValue : hipSuccess;
    if (c->multi()) return check_last ? hipErrorInvalidValue : multi_run(c, count, first, false);
    const epic_hip::TilePlan tp = tile_plan(c);
    if (check_last && !tile_checks(c, tp)) return hipErrorInvalidValue;
    if (tp.halo > 0 && (count >= 2 || check_last)) {   // (a single plain iteration is cheaper as the plain sweep: no ghost rings to load)
        const unsigned total = count + (check_last ? 1u : 0u);
        for (unsigned i = 0; i < total;) {
            const unsigned k = std::min<unsigned>(total - i, (unsigned)tp.halo);
            hipError_t e = epic_hip::launch_tile_2d(c->buf[c->cur], c->buf[c->cur ^ 1], c->maskw, c->rows, c->pitch, tp, (int)k, c->math,
                                                    c->redblack ? (int)((first + i) & 1u) : -1, nullptr, c->stream,
                                                    check_last && i + k == total ? c->h_tile_delta : nullptr);
            if (e != hipSuccess) return e;
            c->cur ^= 1;
            c->work_full += (double)k;
            i += k;
        }
        if (check_last) c->tile_delta_n = tp.tiles_r * tp.tiles_c;   // read_tile_delta reads exactly what this launch wrote
        return hipSuccess;
    }
    const bool no_fuse = c->cfg.no_fuse;
    // (the fused passes have their own 248-column tiling and no work lists: they are used when tracking is off -- or
    //  bypassed for the batch, harmonic_execute_gpu; rb_fused2d_kernel for the precise / fast arithmetic, the RB instance of
    //  the tol pass for tol)
    const bool fuse = fuses_rb_precise(c);
    unsigned i = 0;
    // Jacobi, tol math, 2-D: two consecutive plain iterations run as one pass as well (kernels_2d.hip,
    // jacobi_fused2d_kernel: 4 B of HBM traffic per cell-update instead of 8, bit-identical to two sweeps).
    // EPIC_HIP_FUSE_MIN_CELLS: grids below it keep the single sweeps (default 4 Mcell; the tests set 0).
    if (!no_fuse && fuses_jacobi(c)) {
        if (count >= 2) tune_fused_rows(c, 0, first);
        while (i < count) {
            if (count - i >= 2) {
                hipError_t e = epic_hip::launch_jacobi_fused_2d(c->buf[c->cur], c->buf[c->cur ^ 1], c->maskw, c->rows, c->pitch,
                                                                jacobi_fused_rows_per_task(c), c->math, c->stream, -1, c->maskf());
                if (e != hipSuccess) return e;
                c->cur ^= 1;
                c->work_full += 2.0;
                i += 2;
            } else {
                hipError_t e = enqueue_sweep(c, false, first + i);
                if (e != hipSuccess) return e;
                i++;
            }
        }
        return hipSuccess;
    }
    if (!no_fuse && fuses_rb_tol(c) && count - i >= 2) tune_fused_rows(c, 1, first);
    if (fuse && count - i >= 2) tune_fused_rows(c, 2, first);
    while (!no_fuse && fuses_rb_tol(c) && count - i >= 2) {
        hipError_t e = epic_hip::launch_jacobi_fused_2d(c->buf[c->cur], c->buf[c->cur ^ 1], c->maskw, c->rows, c->pitch,
                                                        jacobi_fused_rows_per_task(c), c->math, c->stream, (int)((first + i) & 1u),
                                                        c->maskf());
        if (e != hipSuccess) return e;
        c->cur ^= 1;
        c->work_full += 2.0;
        i += 2;
    }
    while (fuse && count - i >= 2) {
        hipError_t e = epic_hip::launch_rb_fused_2d(c->buf[c->cur], c->buf[c->cur ^ 1], c->maskw, c->rows, c->pitch,
                                                    fused_rows_per_task(c), c->math, (int)((first + i) & 1u), c->stream, c->maskf());
        if (e != hipSuccess) return e;
        c->cur ^= 1;
        c->work_full += 2.0;
        i += 2;
    }
    for (; i < count; i++) {
        hipError_t e = enqueue_sweep(c, false, first + i);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// `npairs` pairs of iterations starting at iteration `first`; check_last: the second iteration of the last pair is a check (the
// device delta word is zeroed and filled).  bypass: without the lists (every tile; the untracked pass's own task height).
hipError_t enqueue_rb_pairs_tracked(Ctx *c, unsigned npairs, unsigned first, bool check_last, bool bypass)
{
    note_iterations(c, first, 2 * npairs);
    const bool tol = c->math == 4;
    const int rpt = !bypass ? rb_pairs_rows_per_task(c) : tol ? jacobi_fused_rows_per_task(c) : fused_rows_per_task(c);
    const size_t tiles = epic_hip::rb_fused_2d_tiles(c->rows, c->pitch, rpt);
    for (unsigned p = 0; p < npairs; ++p) {
        const bool check = check_last && p + 1 == npairs;
        hipError_t e;
        if (check && (e = hipMemsetAsync(c->d_delta, 0, sizeof(unsigned), c->stream)) != hipSuccess) return e;
        epic_hip::Activity act = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        if (!bypass) {
            if (tiles != c->trk_f.tiles || rpt != c->trk_f.rpt) fold_listed_work(c);   // the sum kept in the old block
            if (c->last_lists != 2) c->trk_f.force = std::max(c->trk_f.force, 1);     // something else has touched the field since
            act = c->trk_f.next(tiles, rpt, c->stream, nullptr);
        }
        const int parity = (int)((first + 2 * p) & 1u);
        e = tol ? epic_hip::launch_jacobi_fused_2d(c->buf[c->cur], c->buf[c->cur ^ 1], c->maskw, c->rows, c->pitch, rpt, c->math, c->stream,
                                                   c->redblack ? parity : -1, c->maskf(), act.list_out ? &act : nullptr, check ? c->d_delta : nullptr)
                : epic_hip::launch_rb_fused_2d(c->buf[c->cur], c->buf[c->cur ^ 1], c->maskw, c->rows, c->pitch, rpt, c->math, parity, c->stream,
                                               c->maskf(), act.list_out ? &act : nullptr, check ? c->d_delta : nullptr);
        if (e != hipSuccess) return e;
        if (!act.list_in) c->work_full += 2.0;   // every tile ran
        if (act.list_out) { c->trk_f.advance(); c->last_lists = 2; }
        else c->last_lists = 0;
        c->cur ^= 1;
    }
    return hipSuccess;
}

hipError_t enqueue_plain_batch(Ctx *c, unsigned count, unsigned first, bool check_last)
{
    if (c->n == 4) return enqueue_plain_run(c, count, first, check_last);
    expect_iteration(c, first);   // (before the decision to replay a captured sequence: a forced iteration is never replayed)
    const bool small = (long long)c->rows * c->pitch <= (1ll << 22);
    const bool no_graph = c->cfg.no_graph;
    // a captured sequence bakes in the work-list buffers and list mode: run eagerly until the forced iterations are over
    if (!small || count < 8 || no_graph || c->multi() || (c->track && (c->trk.force > 0 || c->trk.tiles == 0)))
        return enqueue_plain_run(c, count, first, check_last);
    // (the fused-pass switches are read per batch -- EPIC_HIP_NO_FUSE, EPIC_HIP_FUSE_MIN_CELLS, EPIC_HIP_FUSED_ROWS --, so they
    // belong to the key: 0 = single sweeps, otherwise the task height of the pass)
    const epic_hip::TilePlan tp = tile_plan(c);
    if (tp.halo > 0 && count + (check_last ? 1u : 0u) <= (unsigned)tp.halo) return enqueue_plain_run(c, count, first, check_last);   // one launch: nothing to replay
    const int fuse_cfg = tp.halo > 0 ? -((tp.halo * 1024 + tp.tile_rows) * 4 + tp.tile_cols / 64) : fuses_tol(c) ? jacobi_fused_rows_per_task(c) : 0;
    const auto key = std::make_tuple(2u * count + (check_last ? 1u : 0u), c->cur + 2 * (c->track ? 1 + c->trk.phase : 0), (int)(first & 1u), c->math,
                                     (int)c->redblack, auto_rows_per_task(c), fuse_cfg);
    if (c->graphs_broken) return enqueue_plain_run(c, count, first, check_last);
    auto it = c->graphs.find(key);
    if (it == c->graphs.end()) {
        // Capture is an optimisation: whatever goes wrong in it (begin, a launch during capture, end, instantiate), the
        // state is put back as it was, the error is cleared, the context stops trying and the batch runs eagerly.
        const int cur0 = c->cur, phase0 = c->trk.phase, force0 = c->trk.force;
        const double work0 = c->work_full;
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        hipError_t e = hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal);
        if (e == hipSuccess) {
            e = enqueue_plain_run(c, count, first, check_last);  // (fused passes included)
            hipError_t e2 = hipStreamEndCapture(c->stream, &graph);
            if (e == hipSuccess) e = e2;
        }
        const int cur_flip = c->cur ^ cur0;  // (a fused pass advances two iterations and changes buffers once)
        c->cur = cur0;  // nothing has run yet
        c->trk.phase = phase0;
        c->trk.force = force0;
        const double work = c->work_full - work0;
        c->work_full = work0;
        if (e == hipSuccess) e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        if (graph) (void)hipGraphDestroy(graph);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            c->graphs_broken = true;
            return enqueue_plain_run(c, count, first, check_last);
        }
        if (c->graphs.size() >= 16) drop_graphs(c);
        it = c->graphs.emplace(key, Ctx::Replay{exec, cur_flip, work, check_last ? c->tile_delta_n : 0}).first;
    }
    hipError_t e = hipGraphLaunch(it->second.exec, c->stream);
    if (e != hipSuccess) {  // nothing was enqueued: run the batch eagerly instead, and stop replaying
        (void)hipGetLastError();
        c->graphs_broken = true;
        return enqueue_plain_run(c, count, first, check_last);
    }
    c->seq_next = first + count + (check_last ? 1u : 0u);   // (the replayed launches did not pass through note_iterations)
    c->seq_valid = true;
    c->cur ^= it->second.cur_flip;
    c->work_full += it->second.work;
    if (check_last) c->tile_delta_n = it->second.tile_delta_n;
    if (c->track) c->trk.phase = (int)((c->trk.phase + count) % 6);
    return hipSuccess;
}

// The pipelined form of the small-grid path needs a third buffer of u (padding columns seeded like the two others) and two
// events; both are made on first use and go with the potential values / the context.  EPIC_HIP_TILE_PIPELINE=0: the plain form.
bool tiles_pipeline_ready(Ctx *c)
{
    if (!c->cfg.tile_pipeline) return false;
    for (hipEvent_t &ev : c->ev_blk)
        if (!ev && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            ev = nullptr;
            return false;
        }
    if (!c->spare) {
        if (hipMalloc((void **)&c->spare, c->u_bytes()) != hipSuccess) {
            (void)hipGetLastError();
            c->spare = nullptr;
            return false;
        }
        if (epic_hip::launch_fill(c->spare, (size_t)c->rows * c->pitch, -1e6f, c->stream) != hipSuccess) {
            (void)hipGetLastError();
            free_spare(c);
            return false;
        }
    }
    return true;
}

// Run-ahead at a check (Ctx::ahead): the first block of plain iterations after the check that has just been enqueued, as ONE tile
// launch from the current buffer into the spare one.  Nothing of the context's state changes until the block is adopted.
hipError_t enqueue_ahead(Ctx *c, unsigned first)
{
    c->ahead.live = false;
    const epic_hip::TilePlan tp = tile_plan(c);
    if (tp.halo <= 0 || !c->spare) return hipErrorInvalidValue;
    hipError_t e = epic_hip::launch_tile_2d(c->buf[c->cur], c->spare, c->maskw, c->rows, c->pitch, tp, tp.halo, c->math,
                                            c->redblack ? (int)(first & 1u) : -1, nullptr, c->stream, nullptr);
    if (e != hipSuccess) return e;
    c->ahead.live = true;
    c->ahead.first = first;
    c->ahead.count = (unsigned)tp.halo;
    return hipSuccess;
}

// max |du| of a check iteration that ran as the last step of a tile launch (enqueue_plain_run, check_last): wait for the
// stream -- or, with `after`, for that event only (work enqueued behind the check is not waited for) --, take the maximum over the
// tiles' words in pinned memory.  The wait for an event polls for a bounded while first: the blocks in question run for ~10 us, an
// interrupt-driven wake-up alone takes longer than that.
int read_tile_delta(Harmonic *h, Ctx *c, const char *fn, hipEvent_t after)
{
    hipError_t e = hipSuccess;
    if (after) {
        const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(200);
        do e = hipEventQuery(after);
        while (e == hipErrorNotReady && std::chrono::steady_clock::now() < until);
        if (e == hipErrorNotReady) { (void)hipGetLastError(); e = hipEventSynchronize(after); }
    } else {
        e = hipStreamSynchronize(c->stream);
    }
    if (e != hipSuccess) {
        report(fn, "Failed to synchronize the device after the 'update and check' kernel.");
        return EPIC_ERROR_DEVICE_SYNCHRONIZE;
    }
    float d = 0.0f;
    for (int t = 0; t < c->tile_delta_n; ++t) d = std::max(d, c->h_tile_delta[t]);   // the plan of the launch that wrote them
    h->delta = d;
    return EPIC_SUCCESS;
}

int read_delta(Harmonic *h, Ctx *c, const char *fn)
{
    if (c->n == 4) return EPIC_SUCCESS;   // nothing was swept: delta stays what it was (as harmonic_update_and_check_cpu leaves it)
    if (c->multi()) return multi_read_delta(h, c, fn);
    if (hipMemcpyAsync(c->h_delta, c->d_delta, sizeof(float), hipMemcpyDeviceToHost, c->stream) != hipSuccess) {
        report(fn, "Failed to copy memory from device to host for the max delta.");
        return EPIC_ERROR_MEMCPY_TO_HOST;
    }
    if (hipStreamSynchronize(c->stream) != hipSuccess) {
        report(fn, "Failed to synchronize the device after the 'update and check' kernel.");
        return EPIC_ERROR_DEVICE_SYNCHRONIZE;
    }
    h->delta = *c->h_delta;
    return EPIC_SUCCESS;
}

void force_all(Ctx *c)
{
    c->trk.force = 2;
    c->trk_f.force = std::max(c->trk_f.force, 1);   // (one pass of two iterations, in -> out, rewrites every tile of the other buffer)
    c->last_lists = 0;
    for (auto &sl : c->slabs) {
        sl.trk.force = 2;
        sl.trk_f.force = std::max(sl.trk_f.force, 1);
    }
}

// tiles listed for the next iteration / tiles in all, summed over the domains (false: no lists in use).  forced_runs_all: a
// domain whose next iteration is forced counts as all its tiles (what WILL run); otherwise the count is what the latest
// iteration listed (what its successor NEEDS: a forced iteration still lists the tiles it changed).
bool due_tiles(Ctx *c, unsigned long long *due, unsigned long long *tiles, bool forced_runs_all)
{
    *due = *tiles = 0;
    auto one = [&](const Track &t) {
        if (t.tiles == 0) return false;
        uint32_t counts[Ctx::kL * Ctx::kCS];
        if (hipMemcpy(counts, t.counter(t.phase % 3), sizeof(counts), hipMemcpyDeviceToHost) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        unsigned long long d = 0;
        for (size_t i = 0; i < Ctx::kL; i++) d += counts[i * Ctx::kCS];
        *due += (forced_runs_all && t.force > 0) ? t.tiles : d;
        *tiles += t.tiles;
        return true;
    };
    if (!c->multi()) return one(c->last_lists == 2 ? c->trk_f : c->trk);
    DeviceGuard g;
    for (auto &sl : c->slabs)
        if (hipSetDevice(sl.dev) != hipSuccess || hipStreamSynchronize(sl.stream) != hipSuccess || !one(c->last_lists == 2 ? sl.trk_f : sl.trk)) return false;
    return true;
}

}  // namespace epic_drv

