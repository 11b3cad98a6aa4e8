// driver_loop.hip -- the solver drivers of the C-ABI (reference: libepic/src/harmonic/harmonic_gpu.cu:168-201, :226-304, :327-415;
// harmonic_utilities_gpu.cu:66-138): one iteration, one checked iteration, the whole relaxation with the reference's exit rule, and the
// two rules this library adds to that loop (Jacobi handover, the tol mode's finishing iterations); sparse edits on the resident state.
#include "driver.h"

using namespace epic_drv;

namespace epic_drv {

// ---------------------------------------------------------------------------------------------------------
// One block of iterations on the kernel family the context's state calls for -- the ONE place that picks it, for
// harmonic_execute_gpu's loop, for the deferred iterations of harmonic_update_gpu and for epic_hip_update_n_gpu alike:
//   * several devices: the slabs' stretches between exchanges (multi_run), the check as a sweep of its own;
//   * large 2-D grids with work lists (the library's defaults at the benchmark's size): pairs of iterations as list-driven fused
//     passes, the check as the second iteration of the last pair (an odd count starts with one plain half-sweep);
//   * small 2-D grids: several iterations per launch on LDS tiles, the check folded into the last launch;
//   * everything else: enqueue_plain_batch (fused pairs without lists, a captured graph of single sweeps, single sweeps) and the
//     check as a sweep of its own.
// Same iterations in the same order whichever it takes: the fields are bit-identical (tests/test_gpu_node_flow.py).
// ---------------------------------------------------------------------------------------------------------
int run_block(Harmonic *h, Ctx *c, unsigned plain, unsigned first, bool check, const char *fn, int bypass, bool run_ahead)
{
    if (plain == 0 && !check) return EPIC_SUCCESS;
    if (check && jacobi_reference_checks(c)) {
        // EPIC_HIP_JACOBI_CHECKS=reference (opt-in; round 6).  A Jacobi sweep advances two interleaved red-black chains: the cells of the colour the
        // reference updates at iteration k, and the others -- a chain the reference never computes, one half-sweep behind.
        // A relaxation stopped at a coarse epsilon ends with that second chain up to epsilon / |u| from the reference's field.  With this knob
        // the CHECK iteration is the reference's own half-sweep, in place: after the Jacobi sweeps 1 .. k-1 the other colour holds the first chain's
        // values of sweep k-1, so the half-sweep k leaves exactly the reference's state after k half-sweeps -- both colours on the reference's chain,
        // delta over the reference's colour -- and the next Jacobi sweep goes on from it (the colour it skipped recomputes to what it holds).
        // Bit-identical to harmonic_complete_cpu with the precise arithmetic, iteration count included; measured with the checkers first
        // (tests/jacobi_halfcheck_study.py), here as the plain iterations on whatever family runs them and one half-sweep of the single-sweep kernel.
        if (plain > 0) {
            const int rc = run_block(h, c, plain, first, false, fn, bypass, false);
            if (rc != EPIC_SUCCESS) return rc;
        }
        if (enqueue_check_sweep(c, first + plain) != hipSuccess) {
            report(fn, "Failed to execute the 'Jacobi update and check' kernel.");
            return EPIC_ERROR_KERNEL_EXECUTION;
        }
        h->d_u = current_u(c);
        return read_delta(h, c, fn);
    }
    hipError_t e = hipSuccess;
    bool tiled_check = false;
    hipEvent_t after_check = nullptr;   // run_ahead: the event between the check's launch and the block enqueued ahead of the caller
    const unsigned total = plain + (check ? 1u : 0u);
    if (c->n == 4) {
        // the reference's empty n == 4 branch: nothing is swept, the caller counts
    } else if (c->multi() && total >= 2 && rb_pairs_tracked_multi(c) && (!check || has_delta(c))) {
        // the slabs of the multi-device mode with work lists: pairs as on one device (multi_run_pairs); without the lists -- nearly every
        // tile due -- the untracked stretches, whose exchanges overlap the interior sweeps, and the check as a sweep of its own
        unsigned done = 0;
        if (total & 1u) {
            e = multi_run(c, 1, first, false);
            done = 1;
        }
        if (e == hipSuccess) {
            if (bypass < 0 || check) rb_pairs_choose_rows(c);
            else if (c->pair_rows == 0) c->pair_rows = 16;
            const bool by = bypass < 0 ? bypass_lists_for_batch(c, true) : bypass != 0;
            if (by) {
                c->track = false;
                const unsigned rest = total - done - (check ? 1u : 0u);
                if (rest > 0) e = multi_run(c, rest, first + done, false);
                if (e == hipSuccess && check) e = multi_sweep(c, true, first + done + rest);
                c->track = true;
                force_all(c);
            } else {
                e = multi_run_pairs(c, (total - done) / 2, first + done, check);
            }
        }
    } else if (c->multi()) {
        if (plain > 0) e = multi_run(c, plain, first, false);
        if (e == hipSuccess && check) e = multi_sweep(c, true, first + plain);
    } else if (total >= 2 && rb_pairs_tracked(c) && (!check || has_delta(c))) {
        unsigned done = 0;
        if (total & 1u) {
            e = enqueue_sweep(c, false, first);
            done = 1;
        }
        if (e == hipSuccess) {
            // (the task height of the tracked pass is reconsidered where a check has just been read or is about to be: it reads the
            //  list counters back, which a block of two deferred iterations cannot afford)
            if (bypass < 0 || check) rb_pairs_choose_rows(c);
            else if (c->pair_rows == 0) c->pair_rows = 16;
            const bool by = bypass < 0 ? bypass_lists_for_batch(c, true) : bypass != 0;
            if (by) tune_fused_rows(c, c->math != 4 ? 2 : c->redblack ? 1 : 0, first);   // the untracked pass's task height, measured once per grid
            e = enqueue_rb_pairs_tracked(c, (total - done) / 2, first + done, check, by);
        }
    } else if (check && plain > 0 && tile_checks(c, tile_plan(c))) {
        e = enqueue_plain_batch(c, plain, first, true);
        tiled_check = true;
        // Run-ahead (Ctx::ahead): the caller's next block behind the check, before the wait for the check's result.  An optimisation:
        // whatever fails in it is cleared and the check goes on as without it.
        if (e == hipSuccess && run_ahead && tiles_pipeline_ready(c)) {
            if (hipEventRecord(c->ev_blk[0], c->stream) == hipSuccess) {
                after_check = c->ev_blk[0];
                if (enqueue_ahead(c, first + total) != hipSuccess) (void)hipGetLastError();
            } else {
                (void)hipGetLastError();
            }
        }
    } else {
        if (plain > 0) {
            // Work lists pay while a good part of the tiles is at rest; in the phase in which nearly every tile is due (the
            // middle third of a relaxation from scratch) a list-driven sweep costs more than the plain one -- 12 us + 115 us
            // x share against 97 us per iteration of the fused pass at 8192^2, tol math.  The check that has just completed
            // counted the tiles due next: above the switch share this batch runs without lists, and the next two
            // iterations rebuild them (force = 2, as after an upload).  2-D, one device, tracking in its automatic mode;
            // measured on 8192^2 with every arithmetic and scheme (seconds to eps = 1e-6, never / 0.8): tol Jacobi 2.91 / 2.75,
            // tol red-black 2.62 / 2.52, precise Jacobi 3.78 / 3.69, precise red-black 3.01 / 2.62.  Fields and iteration
            // counts do not depend on it.
            const bool by = bypass < 0 ? bypass_lists_for_batch(c) : (bypass != 0 && c->track);
            if (by) c->track = false;
            e = enqueue_plain_batch(c, plain, first);
            if (by) { c->track = true; force_all(c); }
        }
        if (e == hipSuccess && check) e = enqueue_sweep(c, true, first + plain);
    }
    if (e != hipSuccess) {
        report(fn, check && plain == 0 ? "Failed to execute the 'Jacobi update and check' kernel." : "Failed to execute the 'Jacobi update' kernel.");
        return EPIC_ERROR_KERNEL_EXECUTION;
    }
    h->d_u = current_u(c);
    if (!check) return EPIC_SUCCESS;
    return tiled_check ? read_tile_delta(h, c, fn, after_check) : read_delta(h, c, fn);
}

// ---------------------------------------------------------------------------------------------------------
// harmonic_update_gpu counts (round 6).  The navigation node drives the solver one call per iteration -- one
// harmonic_update_and_check_gpu and steps_per_update - 1 harmonic_update_gpu calls per tick
// (src/epic_navigation_node_harmonic.cpp:165-189; 50 / 100 steps at 10 / 30 Hz) -- and one launch per call would keep it away from
// every kernel that advances more than one iteration per pass (the LDS tiles of the maps it really runs on: 10 iterations per
// launch; the fused pairs of large grids).  So a plain update only COUNTS: currentIteration advances, the iteration joins the
// pending run, and the run is enqueued as ONE block when it has reached the size the kernel family in use is built for
// (defer_cap) and at every point of the boundary at which the caller can see the field or change what the iterations act on:
// harmonic_update_and_check_gpu (the pending run and the check are one block), get_potential_values, update_model, set_cells,
// execute / complete, (un)initialize_*, every epic_hip_* entry point that reads or changes the context (flush_pending).  At most
// cap - 1 iterations are ever outstanding, so the device is never more than one launch behind the caller.  A launch that fails
// surfaces at the call that enqueues it, with the code the reference gives a failed update (EPIC_ERROR_KERNEL_EXECUTION).
// The reference's own update returns after cudaDeviceSynchronize (harmonic_gpu.cu:343-346); none of its callers reads d_u
// itself (they all go through harmonic_get_potential_values_gpu), and d_u has never been directly readable here (pitched rows,
// a private stream).  EPIC_HIP_DEFER=0: one launch per call, as before.
// ---------------------------------------------------------------------------------------------------------
unsigned defer_cap(const Harmonic *h, const Ctx *c)
{
    if (!c->cfg.defer || c->n == 4) return 1;
    unsigned cap = 1;
    if (c->multi()) cap = (unsigned)std::max(1, rb_pairs_tracked_multi(c) ? c->halo / 2 * 2 : c->halo);   // a stretch between two exchanges: one hand-over to the issuing threads
    else if (rb_pairs_tracked(c)) cap = 2;                               // list-driven fused pairs
    else {
        const epic_hip::TilePlan tp = tile_plan(c);
        if (tp.halo > 0) cap = (unsigned)tp.halo;                        // LDS tiles: iterations per launch
        else if (fuses_jacobi(c) || fuses_rb_tol(c) || fuses_rb_precise(c)) cap = 2;   // fused pairs without lists
        else if (c->n == 2 && (long long)c->rows * c->pitch <= (1ll << 22) && !c->cfg.no_graph && !c->graphs_broken && !c->track) cap = 16;   // launch-bound single sweeps: a captured graph
    }
    // never past the caller's own checking period: a block that long would be two of the solver loop's
    const unsigned stagger = h->numIterationsToStaggerCheck ? h->numIterationsToStaggerCheck : 100u;
    return std::max(1u, std::min(cap, stagger));
}

int flush_pending(Harmonic *h, Ctx *c, const char *fn)
{
    if (!c) return EPIC_SUCCESS;
    c->ahead.live = false;   // an ordering point: whatever was enqueued ahead of the caller is not what the caller did next
    if (c->pending == 0) return EPIC_SUCCESS;
    const unsigned n = c->pending, first = c->pending_first;
    c->pending = 0;
    if (!h || !ready(h, c)) return EPIC_SUCCESS;   // part of the device state is gone already (the caller is tearing it down): nothing to advance
    return run_block(h, c, n, first, false, fn, c->defer_bypass ? 1 : 0);
}

}  // namespace epic_drv

namespace epic {
extern "C" {

// ---------------------------------------------------------------------------------------------------------
// solver drivers (reference: libepic/src/harmonic/harmonic_gpu.cu)
// ---------------------------------------------------------------------------------------------------------

int harmonic_update_gpu(Harmonic *harmonic, unsigned int numThreads)  // harmonic_gpu.cu:327-350
{
    static const char *fn = "harmonic_update_gpu";
    (void)numThreads;
    Ctx *c = harmonic ? find_ctx(harmonic) : nullptr;
    if (!harmonic || !ready(harmonic, c)) {
        report(fn, "Invalid data.");
        return EPIC_ERROR_INVALID_DATA;
    }
    // (a caller that has set currentIteration itself since the last call starts a new run: the colour of a red-black iteration is its number's parity)
    if (c->pending > 0 && c->pending_first + c->pending != harmonic->currentIteration) {
        const int rc = flush_pending(harmonic, c, fn);
        if (rc != EPIC_SUCCESS) return rc;
    }
    if (c->pending == 0) {
        c->pending_first = harmonic->currentIteration;
        c->pending_cap = defer_cap(harmonic, c);   // (nothing the cap depends on changes while iterations are pending: every such change flushes)
    }
    c->pending++;
    c->tick_plain++;
    harmonic->currentIteration++;
    if (c->ahead.live) {
        if (c->pending_first != c->ahead.first) {
            c->ahead.live = false;   // the caller went elsewhere
        } else if (c->pending == c->ahead.count) {
            // exactly the block that was enqueued behind the latest check: its result becomes the current buffer, nothing is launched
            std::swap(c->spare, c->buf[c->cur]);
            drop_graphs(c);          // (captured sequences hold buffer addresses)
            c->work_full += (double)c->ahead.count;
            c->ahead.live = false;
            c->pending = 0;
            harmonic->d_u = current_u(c);
            return EPIC_SUCCESS;
        } else {
            return EPIC_SUCCESS;     // still inside that block
        }
    }
    if (c->pending >= c->pending_cap) return flush_pending(harmonic, c, fn);
    return EPIC_SUCCESS;
}

// decide: also settle whether the deferred blocks up to the next check run with or without the work lists (one read-back of the
// list counters, tracked 2-D grids only); harmonic_execute_gpu's own loop decides per block and passes false.
static int update_and_check(Harmonic *harmonic, bool decide)
{
    static const char *fn = "harmonic_update_and_check_gpu";
    Ctx *c = harmonic ? find_ctx(harmonic) : nullptr;
    if (!harmonic || !ready(harmonic, c) || !has_delta(c) || harmonic->d_delta == nullptr) {
        report(fn, "Invalid data.");
        return EPIC_ERROR_INVALID_DATA;
    }
    // Run-ahead: a caller that followed its previous check with at least one block of plain updates (the navigation node's tick) is
    // expected to do so again; harmonic_execute_gpu's own loop (decide == false) pipelines whole blocks itself.
    const bool run_ahead = decide && c->cfg.defer && c->pending_cap > 1 && c->tick_plain >= c->pending_cap;
    c->tick_plain = 0;
    c->ahead.live = false;
    // the pending run and this check are one block -- unless the caller has moved currentIteration in between
    unsigned plain = 0, first = harmonic->currentIteration;
    if (c->pending > 0 && c->pending_first + c->pending == harmonic->currentIteration) {
        plain = c->pending;
        first = c->pending_first;
        c->pending = 0;
    } else if (c->pending > 0) {
        const int rc = flush_pending(harmonic, c, fn);
        if (rc != EPIC_SUCCESS) return rc;
    }
    const int rc = run_block(harmonic, c, plain, first, true, fn, c->defer_bypass ? 1 : 0, run_ahead);
    if (rc != EPIC_SUCCESS) return rc;
    harmonic->currentIteration++;
    if (decide) c->defer_bypass = c->cfg.defer && bypass_lists_for_batch(c, rb_pairs_tracked(c) || rb_pairs_tracked_multi(c));
    return harmonic->delta < harmonic->epsilon ? EPIC_SUCCESS_AND_CONVERGED : EPIC_SUCCESS;
}

int harmonic_update_and_check_gpu(Harmonic *harmonic, unsigned int numThreads)  // harmonic_gpu.cu:353-415
{
    (void)numThreads;
    return update_and_check(harmonic, true);
}

int harmonic_execute_gpu(Harmonic *harmonic, unsigned int numThreads)  // harmonic_gpu.cu:226-304
{
    static const char *fn = "harmonic_execute_gpu";
    if (harmonic == nullptr || harmonic->m == nullptr || harmonic->u == nullptr || harmonic->locked == nullptr ||
        harmonic->epsilon <= 0.0 || harmonic->d_m == nullptr || harmonic->d_u == nullptr ||
        harmonic->d_locked == nullptr) {
        report(fn, "Invalid data.");
        return EPIC_ERROR_INVALID_DATA;
    }
    if (numThreads % 32 != 0) {
        report(fn, "Must specficy a number of threads divisible by 32 (the number of threads in a warp).");
        return EPIC_ERROR_INVALID_CUDA_PARAM;
    }
    if (harmonic->numIterationsToStaggerCheck == 0) {  // the reference divides by zero here (harmonic_gpu.cu:268)
        report(fn, "Invalid data (numIterationsToStaggerCheck must be positive).");
        return EPIC_ERROR_INVALID_DATA;
    }
    Ctx *c = find_ctx(harmonic);
    if (!ready(harmonic, c)) {
        report(fn, "Invalid data.");
        return EPIC_ERROR_INVALID_DATA;
    }
    if (c->n == 4) {   // no sweep ever changes delta: the reference's loop (harmonic_gpu.cu:266-290) would never return
        report(fn, "Invalid data (n = 4 is a counting no-op: there is nothing to relax).");
        return EPIC_ERROR_INVALID_DATA;
    }
    {
        const int frc = flush_pending(harmonic, c, fn);   // iterations counted by harmonic_update_gpu: the relaxation starts from their field
        if (frc != EPIC_SUCCESS) return frc;
    }

    harmonic->currentIteration = 0;
    fold_listed_work(c);
    c->work_full = 0.0;  // epic_hip_work_done counts from here
    c->finish_from = 0;
    int result = harmonic_initialize_gpu(harmonic, numThreads);
    if (result != EPIC_SUCCESS) {
        report(fn, "Failed to initialize GPU variables.");
        return result;
    }

    unsigned int mMax = 0;
    for (unsigned int i = 0; i < harmonic->n; i++) mMax = std::max(mMax, harmonic->m[i]);
    harmonic->delta = harmonic->epsilon + 1.0f;

    // The reference's loop (harmonic_gpu.cu:266-290): a sweep with currentIteration % stagger == 0 is a check
    // sweep; a plain sweep resets "converged"; exit right after a converged check with currentIteration >= mMax.
    // The plain sweeps between two checks need no host decision, so they are enqueued back to back.
    //
    // Jacobi handover.  A Jacobi iteration is two interleaved red-black chains: the cells of one colour at even iterations
    // and of the other colour at odd ones never meet the rest.  In f32 the two chains may stagnate one unit in the last
    // place apart; every cell then flips between them for ever and max |du| never falls below eps, where the reference's
    // red-black iteration from the same state stops (first seen on the nav_core plugin's SECOND makePlan, whose start is
    // the first goal's converged field: tests/test_gpu_plugin_replay.py).  This loop's contract is "until the test fires",
    // so at the first check with delta < 1 that is not below the previous check's delta it continues with the reference's
    // in-place half-sweeps (what EPIC_HIP_SCHEME=redblack runs from the start), which end as the reference ends.  delta < 1
    // keeps the rule away from the phase in which the front still moves (delta ~1e6 for many checks in a row); a handover
    // that comes early costs time, never correctness.  oracle_jacobi_complete / oracle_tol_complete state the same rule.
    struct Handover {
        Ctx *c;
        bool done = false;
        float last_check = -1.0f;  // no check yet
        // the deltas of the latest checks (a ring), for the plateau test of the tol mode's hand-over
        enum { kWindow = 32 };
        float recent[kWindow];
        int seen = 0;
        void note(float d) { recent[seen++ % kWindow] = d; }
        float window_ago() const { return seen >= kWindow ? recent[seen % kWindow] : -1.0f; }   // the delta kWindow checks before the next one
        ~Handover() { if (done) { c->redblack = false; force_all(c); } }
    } handover{c};   // (the context outlives this function: it still holds the field and the mask)
    // Finish (tol math only; EPIC_HIP_TOL_FINISH=0 switches it off).  Where a converged f32 field ends inside the iteration's
    // dead band is decided by the last few per cent of the iterations, and the parity bar is on the reference's end point:
    // so at the first check with delta < 10 epsilon (100 epsilon when epsilon <= 1e-5: below) this loop leaves the tol arithmetic and continues with THE REFERENCE'S OWN
    // ITERATION -- red-black half-sweeps with the bit-exact expf / logf, what the library runs by default from the start --
    // and only a check of that phase may end it.  Measured with the checker (oracle_tol_complete states the same rule) on the
    // reference's maps: umass.png 1.6e-5 -> 1.4e-6 from harmonic_complete_cpu's field, after 86 101 + 8 101 iterations against the
    // reference's 94 401; maze 1.4e-6 -> 5.6e-7 (52 001 + 3 501 against 52 101); basic 3.3e-6 -> 2.3e-7 (19 601 + 4 301 against
    // 23 801).  What it costs: the finishing phase starts from a field that already looks converged and walks the dead band on its
    // own, so its iterations come ON TOP of the tol phase's -- 8 % more iterations than the reference on maze, 22 % on the 8192^2
    // benchmark grid (45 001 + 9 800, where 82 % of the iterations run at the tol kernels' speed) -- and with them a tol Jacobi
    // relaxation is no faster to a converged field than the bit-exact default (round 3: 2.64 s against 2.51 s at 8192^2; tol
    // red-black: 2.20 s).  tol is a kernel-throughput mode; the time-to-solution numbers are in the bench line (relax*).
    struct Finish {
        Ctx *c;
        int math0;
        bool redblack0, on = false;
        ~Finish() { if (on) { c->math = math0; c->redblack = redblack0; force_all(c); } }
    } finish{c, c->math, c->redblack};
    // The switch only exists for relaxations to STAGNATION (epsilon <= 1e-5, where the tol iteration alone also comes to rest and
    // the finishing phase moves the end point by ~1e-5).  At the epsilons the reference's callers use (1e-3: the ROS plugin and
    // node; 1e-2: the python default) the loop stops while the field still moves, the iteration at which it stops decides the
    // field, and the tol iteration alone stops elsewhere than the reference -- basic.png at 1e-3: 7 001 iterations instead of
    // 8 701, 5.6e-2 (relative) away.  There the finishing phase is what makes the stop the reference's, and it stays on.
    const bool finish_off_asked = c->cfg.tol_finish == 0;   // EPIC_HIP_TOL_FINISH=0
    const bool finish_off = finish_off_asked && harmonic->epsilon <= kTolFinishOptionalBelow;
    if (c->math == 4 && finish_off_asked && !finish_off) {
        static std::atomic<bool> said{false};
        if (!said.exchange(true))
            fprintf(stderr, "Warning[epic_hip]: EPIC_HIP_TOL_FINISH=0 ignored for epsilon > 1e-5 (the relaxation stops before stagnation; the finishing iterations decide where).\n");
    }
    const bool finish_wanted = c->math == 4 && !finish_off;
    // 10 at the epsilons the callers use (round 3's rule, verified there on every map of the reference), 100 for relaxations to
    // stagnation (epsilon <= 1e-5).  Round 4 swept the factor over all thirteen maps x {1e-2, 1e-3, 1e-6} x both schemes
    // (tools/finish_study_gpu.py): twelve maps are inside the bar for ANY factor; maps/trivial.png -- an almost
    // empty 1024^2 room whose delta crosses epsilon in single ulps over tens of thousands of iterations -- is outside it for
    // 10 (at 1e-6: 475 401 instead of 503 201 iterations, 4.2e-3), 30, 50, 100 and 300 at one epsilon or the other, by chance
    // rather than by trend.  This rule is inside the bar on all 78 cases; on that map that is one good draw, and only the
    // bit-exact default reproduces the reference there.
    float finish_factor = harmonic->epsilon <= kTolFinishOptionalBelow ? 100.0f : 10.0f;
    if (c->cfg.tol_finish_factor >= 1.0f) finish_factor = c->cfg.tol_finish_factor;   // EPIC_HIP_TOL_FINISH_FACTOR: study knob (tools/finish_study_gpu.py; include/epic_hip.h)
    const float finish_below = finish_factor * harmonic->epsilon;
    const unsigned stagger = harmonic->numIterationsToStaggerCheck;
    result = EPIC_SUCCESS;
    // what follows every check iteration (result and harmonic->delta are the check's)
    auto after_check = [&] {
        // (not at the FIRST check of a run that has not moved yet: a red-black iteration 0 whose colour has no cell next to a goal reports
        //  delta = 0 exactly -- the 512^3 benchmark grid does --, and handing over there ran the whole relaxation in the reference's
        //  arithmetic: correct, and not what the mode is for; round 6)
        if (finish_wanted && !finish.on && harmonic->delta < finish_below && !(harmonic->delta == 0.0f && handover.seen == 0)) {
            // A delta that has fallen by less than 0.3 % per check over the last 32 checks is a plateau on which the stop is decided
            // by single ulps of single cells: maps/trivial.png falls 0.12-0.4 % per 100 iterations, in steps of one ulp of |u| < 8
            // (at delta = 1e-5 = 21 ulp one step is 5 %, once in ~40 checks -- hence a window, not two successive checks).  There
            // an arithmetic that is not the reference's bit for bit cannot promise the reference's stop, whatever the hand-over
            // factor; the reference's other maps fall 0.8 % per check and faster.
            const float ago = handover.window_ago();
            if (ago > 0.0f && harmonic->delta > 0.908f * ago)   // 0.997^32
                fprintf(stderr, "Warning[epic_hip]: tol math on a slowly converging map (delta %.3e, %.3e %d checks earlier): the stop is decided by "
                                "single ulps here and the tol arithmetic cannot promise the 1e-5 parity bar; the default (precise) math reproduces the reference.\n",
                        (double)harmonic->delta, (double)ago, (int)Handover::kWindow);
            finish.on = true;
            c->finish_from = harmonic->currentIteration;
            c->math = 0;          // precise
            c->redblack = true;   // the reference's half-sweeps, colour by currentIteration
            force_all(c);
            // Relaxations to stagnation (epsilon <= 1e-5): only a check of the finishing phase may end the loop -- there the finishing
            // iterations decide the end point.  At the callers' epsilons THIS check keeps its verdict (round 6; found by the campaign of
            // tests/tol_campaign.py on maps that converge within a few checks: delta falls from above 10 epsilon to below epsilon between
            // two checks, the reference stops here, and 100 more iterations at an epsilon at which the field still moves ended up to
            // 7e-4 away).  The tol delta of an iteration is the reference's to an ulp or two of |u|: as good a judge of "below epsilon"
            // as a finishing phase's delta would be 100 iterations later.  oracle_tol_complete and SlabSolver.solve state the same rule.
            if (!(harmonic->epsilon > kTolFinishOptionalBelow)) result = EPIC_SUCCESS;
        } else if (!c->redblack && result == EPIC_SUCCESS && harmonic->delta < 1.0f && handover.last_check >= 0.0f &&
                   harmonic->delta >= handover.last_check) {
            c->redblack = true;
            force_all(c);
            handover.done = true;
        }
        handover.last_check = harmonic->delta;
        handover.note(harmonic->delta);
    };
    while (result != EPIC_SUCCESS_AND_CONVERGED || harmonic->currentIteration < mMax) {
        if (harmonic->currentIteration % stagger == 0) {
            result = update_and_check(harmonic, false);
            if (result != EPIC_SUCCESS && result != EPIC_SUCCESS_AND_CONVERGED) {
                report(fn, "Failed to perform the Jacobi update and check step.");
                return result;
            }
            after_check();
        } else if (rb_pairs_tracked(c) && has_delta(c)) {
            // Tracked red-black with the precise math on a large grid (the library's defaults at the benchmark's size): the plain
            // iterations up to the next check AND that check as PAIRS, each one list-driven fused pass; the check is the second
            // iteration of the last pair (an odd count starts with one plain half-sweep).  Same iterations, same order, same bits.
            const unsigned batch = stagger - harmonic->currentIteration % stagger;
            result = run_block(harmonic, c, batch, harmonic->currentIteration, true, fn, -1);
            if (result != EPIC_SUCCESS) return result;
            harmonic->currentIteration += batch + 1;
            result = harmonic->delta < harmonic->epsilon ? EPIC_SUCCESS_AND_CONVERGED : EPIC_SUCCESS;
            after_check();
        } else if (tile_checks(c, tile_plan(c)) && tiles_pipeline_ready(c)) {
            // Small grids, pipelined (round 4).  A block = the plain iterations up to the next check and that check, as tile launches.
            // The host does not wait for a check's result before enqueueing the NEXT block: it enqueues it from the state the
            // check refers to, into the two buffers that state is not in (three buffers rotate), and only then waits for the
            // check (an event; the per-tile maxima are in pinned memory).  If the check ends the loop -- or changes the mode
            // (Jacobi handover, the tol mode's finishing phase) -- the block enqueued ahead is let run and ignored: the state the
            // check refers to is intact.  The GPU never waits for the host between blocks (that wait was 12 % of a map's
            // relaxation: profiles/r04_experiments.txt); iterations, checks and results are those of the plain loop.
            struct Blk { float *final_buf; unsigned it_end, steps; int slot, ntiles; };
            float *bufs[3] = {c->buf[0], c->buf[1], c->spare};
            int slot = 0;
            const epic_hip::TilePlan tp = tile_plan(c);   // ONE plan for every block of this stretch (the mode cannot change inside it)
            // (EPIC_HIP_JACOBI_CHECKS=reference: the check is a launch of its own -- ONE step, the reference's colour of that iteration; run_block says why)
            const bool ref_checks = jacobi_reference_checks(c);
            auto enqueue_block = [&](float *in, unsigned first, Blk *out) -> hipError_t {
                const unsigned total = (stagger - first % stagger) + 1;   // the plain iterations and the check
                float *o1 = nullptr, *o2 = nullptr;
                for (float *b : bufs)
                    if (b != in) (o1 ? o2 : o1) = b;
                float *src = in;
                for (unsigned i = 0; i < total;) {
                    unsigned k = std::min<unsigned>(total - i, (unsigned)tp.halo);
                    if (ref_checks && i + k == total && k > 1) k--;
                    const bool half_sweep = c->redblack || (ref_checks && i + k == total);
                    float *dst = src == o1 ? o2 : o1;
                    hipError_t e = epic_hip::launch_tile_2d(src, dst, c->maskw, c->rows, c->pitch, tp, (int)k, c->math,
                                                            half_sweep ? (int)((first + i) & 1u) : -1, nullptr, c->stream,
                                                            i + k == total ? c->h_tile_delta + (size_t)slot * kTileDeltaCap : nullptr);
                    if (e != hipSuccess) return e;
                    src = dst;
                    i += k;
                }
                hipError_t e = hipEventRecord(c->ev_blk[slot], c->stream);
                *out = Blk{src, first + total, total, slot, tp.tiles_r * tp.tiles_c};
                slot ^= 1;
                return e;
            };
            auto adopt = [&](const Blk &b) {   // the state after block b becomes the context's current buffer
                if (b.final_buf == c->spare) { std::swap(c->spare, c->buf[c->cur]); drop_graphs(c); }   // (captured sequences hold addresses)
                else c->cur = b.final_buf == c->buf[0] ? 0 : 1;
                harmonic->d_u = current_u(c);
            };
            Blk prev, next, verified;      // verified: the latest block whose check has been read -- what currentIteration describes
            bool have_verified = false;
            // every error return below leaves the context in the state the iteration count describes: the blocks in flight are
            // waited for (best effort) and the last verified block is adopted
            auto bail = [&](int code) {
                (void)hipStreamSynchronize(c->stream);
                if (have_verified) adopt(verified);
                return code;
            };
            hipError_t pe = enqueue_block(c->buf[c->cur], harmonic->currentIteration, &prev);
            bool leave = false;
            while (pe == hipSuccess && !leave) {
                pe = enqueue_block(prev.final_buf, prev.it_end, &next);   // ahead of prev's check
                if (pe != hipSuccess) break;
                if (hipEventSynchronize(c->ev_blk[prev.slot]) != hipSuccess) {
                    report(fn, "Failed to synchronize the device after the 'update and check' kernel.");
                    return bail(EPIC_ERROR_DEVICE_SYNCHRONIZE);
                }
                float d = 0.0f;
                const float *tile_max = c->h_tile_delta + (size_t)prev.slot * kTileDeltaCap;
                for (int t = 0; t < prev.ntiles; ++t) d = std::max(d, tile_max[t]);   // exactly the words that block's check wrote
                harmonic->delta = d;
                harmonic->currentIteration = prev.it_end;
                c->work_full += (double)prev.steps;
                verified = prev;
                have_verified = true;
                result = d < harmonic->epsilon ? EPIC_SUCCESS_AND_CONVERGED : EPIC_SUCCESS;
                const int math0 = c->math;
                const bool rb0 = c->redblack;
                after_check();
                const bool stop = result == EPIC_SUCCESS_AND_CONVERGED && harmonic->currentIteration >= mMax;
                const bool changed = c->math != math0 || c->redblack != rb0;
                if (stop || changed) {
                    // the block enqueued ahead ran (or runs) in a mode, or past an end, that the check has just ruled out
                    if (hipStreamSynchronize(c->stream) != hipSuccess) return bail(EPIC_ERROR_DEVICE_SYNCHRONIZE);
                    adopt(prev);
                    leave = true;   // the outer loop ends (stop) or goes on from here in the new mode
                } else {
                    prev = next;
                }
            }
            if (pe != hipSuccess) {
                report(fn, "Failed to perform the Jacobi update step.");
                return bail(EPIC_ERROR_KERNEL_EXECUTION);
            }
        } else if (tile_checks(c, tile_plan(c))) {
            // Small grids (kernels_tile2d.hip): the plain iterations up to the next check AND that check are one sequence of tile
            // launches (one captured graph); the check is the last step of the last launch and leaves its max |du| per tile in
            // pinned memory.  Same iterations in the same order as the branches above and below run them.
            const unsigned batch = stagger - harmonic->currentIteration % stagger;
            result = run_block(harmonic, c, batch, harmonic->currentIteration, true, fn, -1);
            if (result != EPIC_SUCCESS) return result;
            harmonic->currentIteration += batch + 1;
            result = harmonic->delta < harmonic->epsilon ? EPIC_SUCCESS_AND_CONVERGED : EPIC_SUCCESS;
            after_check();
        } else {
            // every plain iteration returns SUCCESS (which clears a previous CONVERGED), so the ones up to the next
            // check need no host decision in between: enqueue them as one batch (with or without the work lists: run_block)
            const unsigned batch = stagger - harmonic->currentIteration % stagger;
            result = run_block(harmonic, c, batch, harmonic->currentIteration, false, fn, -1);
            if (result != EPIC_SUCCESS) return result;
            harmonic->currentIteration += batch;
            result = EPIC_SUCCESS;
        }
    }

    result = harmonic_get_potential_values_gpu(harmonic);
    if (result != EPIC_SUCCESS) {
        report(fn, "Failed to get all the potential values.");
        return result;
    }
    result = harmonic_uninitialize_gpu(harmonic);
    if (result != EPIC_SUCCESS) {
        report(fn, "Failed to uninitialize GPU variables.");
        return result;
    }
    return EPIC_SUCCESS;
}

int harmonic_complete_gpu(Harmonic *harmonic, unsigned int numThreads)  // harmonic_gpu.cu:168-201
{
    int result = harmonic_initialize_dimension_size_gpu(harmonic);
    if (result != EPIC_SUCCESS) return result;
    result = harmonic_initialize_potential_values_gpu(harmonic);
    if (result != EPIC_SUCCESS) return result;
    result = harmonic_initialize_locked_gpu(harmonic);
    if (result != EPIC_SUCCESS) return result;

    result = harmonic_execute_gpu(harmonic, numThreads);
    if (result != EPIC_SUCCESS) return result;

    result = EPIC_SUCCESS;
    if (harmonic_uninitialize_dimension_size_gpu(harmonic) != EPIC_SUCCESS) result = EPIC_ERROR_DEVICE_FREE;
    if (harmonic_uninitialize_potential_values_gpu(harmonic) != EPIC_SUCCESS) result = EPIC_ERROR_DEVICE_FREE;
    if (harmonic_uninitialize_locked_gpu(harmonic) != EPIC_SUCCESS) result = EPIC_ERROR_DEVICE_FREE;
    return result;
}

// ---------------------------------------------------------------------------------------------------------
// sparse edits on the resident state (reference: libepic/src/harmonic/harmonic_utilities_gpu.cu:66-138)
// ---------------------------------------------------------------------------------------------------------

int harmonic_utilities_set_cells_2d_gpu(Harmonic *harmonic, unsigned int numThreads, unsigned int k, unsigned int *v,
                                        unsigned int *types)
{
    static const char *fn = "harmonic_utilities_set_cells_2d_gpu";
    (void)numThreads;
    if (harmonic == nullptr || harmonic->n == 0 || harmonic->m == nullptr || harmonic->u == nullptr ||
        harmonic->locked == nullptr || k == 0 || v == nullptr || types == nullptr) {
        report(fn, "Invalid data.");
        return EPIC_ERROR_INVALID_DATA;
    }
    Ctx *c = find_ctx(harmonic);
    if (!ready(harmonic, c) || c->n != 2) {
        report(fn, "Invalid data.");
        return EPIC_ERROR_INVALID_DATA;
    }
    {
        const int frc = flush_pending(harmonic, c, fn);   // the iterations counted so far act on the cells as they were
        if (frc != EPIC_SUCCESS) return frc;
    }
    if (c->multi()) return multi_set_cells(c, k, v, types, fn);
    unsigned *d_v = nullptr, *d_types = nullptr;
    int rc = EPIC_SUCCESS;
    force_all(c);  // cells and mask bits change under the work lists
    if (hipMalloc((void **)&d_v, 2 * (size_t)k * sizeof(unsigned)) != hipSuccess ||
        hipMalloc((void **)&d_types, (size_t)k * sizeof(unsigned)) != hipSuccess) {
        (void)hipGetLastError();
        report(fn, "Failed to allocate device-side memory for the cell locations and types.");
        rc = EPIC_ERROR_DEVICE_MALLOC;
    } else if (hipMemcpyAsync(d_v, v, 2 * (size_t)k * sizeof(unsigned), hipMemcpyHostToDevice, c->stream) != hipSuccess ||
               hipMemcpyAsync(d_types, types, (size_t)k * sizeof(unsigned), hipMemcpyHostToDevice, c->stream) != hipSuccess) {
        report(fn, "Failed to copy memory from host to device for the cell locations and types.");
        rc = EPIC_ERROR_MEMCPY_TO_DEVICE;
    } else if (epic_hip::launch_set_cells_2d(c->buf[c->cur], c->maskw, c->rows, c->cols, c->pitch, k, d_v, d_types,
                                             c->stream) != hipSuccess ||
               epic_hip::launch_fuse_masks_2d(c->maskw, c->rows, c->pitch, c->maskf(), c->stream) != hipSuccess) {
        report(fn, "Failed to execute the 'set cells' kernel.");
        rc = EPIC_ERROR_KERNEL_EXECUTION;
    }
    if (hipStreamSynchronize(c->stream) != hipSuccess && rc == EPIC_SUCCESS) {
        report(fn, "Failed to synchronize the device after 'set cells' kernel.");
        rc = EPIC_ERROR_DEVICE_SYNCHRONIZE;
    }
    if (d_v) (void)hipFree(d_v);       // freed on every path (the reference leaks them on errors)
    if (d_types) (void)hipFree(d_types);
    return rc;
}

}  // extern "C"
}  // namespace epic

