// driver.h -- the host driver of the GPU half of the libepic C-ABI (HIP runtime on MI355X), shared declarations.
//
// Replaces libepic/src/harmonic/harmonic_gpu.cu:156-434 (solver drivers), harmonic_model_gpu.cu:34-204 (device-state lifecycle)
// and harmonic_utilities_gpu.cu:66-138 (sparse edits) with the same exported names, validation, return codes and
// "Error[<function>]: <text>" stderr lines, over a different device design:
//
//  * device state lives in a library-side context keyed by the caller's Harmonic* (the 80-byte struct has no room for a second
//    ping-pong buffer, a stream or pinned readback memory).  The struct's d_* fields are still set non-null / nulled exactly
//    where the reference does, because callers and the library itself null-test them (harmonic_gpu.cu:208, :232-235;
//    harmonic_model_gpu.cu:174-176);
//  * u is kept pitched (row length padded to 256 floats) in two buffers; d_u points at the current one;
//  * locked is kept bit-packed (d_locked points at the packed words);
//  * sweeps are enqueued on one non-blocking stream; only the check sweeps, the readbacks and the edits synchronise (the
//    reference synchronises the whole device after every kernel).
//
// Translation units (round 5; one 2.9 kLoC file before):
//   driver_config.cpp    struct Config: EVERY EPIC_HIP_* environment knob, parsed once when a context is created (and again only
//                        on epic_hip_config_reload); nothing else in the library calls getenv
//   driver_registry.hip  contexts keyed by Harmonic*, dimensions, uploads; the lifecycle entry points (initialize / uninitialize /
//                        update_model / get_potential_values)
//   driver_plan.hip      which kernel family runs a batch and with what tiling (task heights, fused passes, LDS tiles, tracked
//                        pairs, list bypass); the tuner of the fused passes' task height
//   driver_enqueue.hip   iterations enqueued: single sweeps, batches (fused passes, tiles, captured graphs), tracked pairs; the
//                        readback of max |du|; work-list bookkeeping
//   driver_multi.hip     several devices in one process (EPIC_HIP_DEVICES): slabs, halo copies, one issuing thread per slab
//   driver_loop.hip      harmonic_update_gpu / _and_check_gpu / execute / complete with the reference's exit rule, the Jacobi
//                        handover and the tol mode's finishing iterations; set_cells
//   driver_ext.hip       the extension entry points of include/epic_hip.h (batches, timing, mode selection, reports, raw operators)
// All of it is host code: tests/test_host_driver_faults.py compiles these files unchanged with g++ against a fake HIP runtime.
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <map>
#include <string>
#include <thread>
#include <vector>
#include <mutex>
#include <tuple>
#include <unordered_map>

#include "../../include/epic/epic_abi.h"
#include "../../include/epic_hip.h"
#include "driver_config.h"
#include "kernels.h"

namespace epic_drv __attribute__((visibility("hidden"))) {

using epic::Harmonic;

struct DeviceGuard {  // the caller's current device is restored whatever happens in between
    int prev = -1;
    DeviceGuard() { if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); prev = -1; } }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
struct Crew;  // one host thread per slab (multi-device mode), below

// Work lists of one domain (the whole grid, or one slab of it).  Every iteration lists the tiles its successor has to
// recompute; tiles whose inputs did not change are never touched (bit-identical results, see kernels_2d.hip).  One device
// block `wake` holds 3 counter sets (L words each, L = kWakeListCount), the running sum of listed tiles, the queued marks of
// both directions (tiles words each) and both directions' L lists (ceil(tiles / L) words each); `phase` (mod 6) says which
// direction (phase & 1) and which counter set (phase % 3) the next launch consumes.  force > 0: the next `force` iterations
// run every tile.
struct Track {
    uint32_t *wake = nullptr;
    int phase = 0, rpt = 0, force = 2;
    size_t tiles = 0;
    static constexpr size_t kL = epic_hip::kWakeListCount, kCS = epic_hip::kWakeCounterStride;
    uint32_t *counter(int set) const { return wake + kL * kCS * set; }
    // two words behind the counters: the running sum (64 bits) of the tiles handed to list-driven launches
    unsigned long long *total() const { return reinterpret_cast<unsigned long long *>(wake + 3 * kL * kCS); }
    uint32_t *queued(int i) const { return wake + 3 * kL * kCS + 2 + (size_t)i * tiles; }
    uint32_t *list(int i) const { return wake + 3 * kL * kCS + 2 + 2 * tiles + (size_t)i * kL * epic_hip::sweep_2d_list_cap(tiles); }
    static size_t words(size_t tiles) { return 3 * kL * kCS + 2 + 2 * tiles + 2 * kL * epic_hip::sweep_2d_list_cap(tiles); }
    static size_t zeroed_words(size_t tiles) { return 3 * kL * kCS + 2 + 2 * tiles; }  // counters, sum and marks; lists need no init
    void release()
    {
        if (wake) (void)hipFree(wake);
        wake = nullptr;
        tiles = 0;
    }
    // The lists of the next launch of this domain over `tiles_now` tiles of `rpt_now` rows (re-allocated, on `stream`'s device,
    // when the tiling has changed: *changed says so).  All null when the block cannot be had: the launch then runs untracked.
    epic_hip::Activity next(size_t tiles_now, int rpt_now, hipStream_t stream, bool *changed)
    {
        epic_hip::Activity act = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        if (tiles_now != tiles || rpt_now != rpt) {
            if (changed) *changed = true;
            release();
            if (hipMalloc((void **)&wake, words(tiles_now) * sizeof(uint32_t)) == hipSuccess &&
                hipMemsetAsync(wake, 0, zeroed_words(tiles_now) * sizeof(uint32_t), stream) == hipSuccess) {
                tiles = tiles_now;
                rpt = rpt_now;
                phase = 0;
                force = 2;
            } else {
                (void)hipGetLastError();
                release();
            }
        }
        if (tiles) {
            const int li = phase & 1, ci = phase % 3;
            act.list_in = force > 0 ? nullptr : list(li);
            act.count_in = counter(ci);
            act.list_out = list(li ^ 1);
            act.count_out = counter((ci + 1) % 3);
            act.count_zero = counter((ci + 2) % 3);
            act.queued_in = queued(li);
            act.queued_out = queued(li ^ 1);
            act.total = total();
        }
        return act;
    }
    void advance()   // after a successful launch with lists
    {
        phase = (phase + 1) % 6;
        if (force > 0) force--;
    }
    // the lists the NEXT launch will consume, in the *_out fields (for launch_wake_tile_range)
    epic_hip::Activity upcoming() const
    {
        epic_hip::Activity act = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        if (tiles) {
            act.list_out = list(phase & 1);
            act.count_out = counter(phase % 3);
            act.queued_out = queued(phase & 1);
        }
        return act;
    }
};

struct Ctx {
    int n = 0;
    int m[4] = {0, 0, 0, 0};   // as given by the caller
    int rows = 0;           // 2-D: m[0];            3-D: m[0] * m[1] (rows of length m[2]);   4-D: m[0] * m[1] * m[2] (held, never swept)
    int cols = 0;           // last dimension
    int pitch = 0;          // floats per row on the device
    float *buf[2] = {nullptr, nullptr};
    int cur = 0;
    uint32_t *maskw = nullptr;
    unsigned *d_m = nullptr;
    unsigned *d_delta = nullptr;   // float bits of max |du|
    float *h_delta = nullptr;      // pinned
    // small grids (kernels_tile2d.hip): max |du| per tile of a check iteration, written by the kernel straight into pinned host
    // memory -- no zeroing, no atomics on one word, no copy: the check costs the wait for the stream and nothing else
    float *h_tile_delta = nullptr;   // 2 x kTileDeltaCap floats: two blocks of iterations may be in flight (tiles_pipelined)
    int tile_delta_n = 0;            // words the latest check launch of enqueue_plain_run wrote there (that launch's tiles_r x tiles_c)
    // small grids, harmonic_execute_gpu: a THIRD buffer of u and two events, so that the block of iterations after a check can be
    // enqueued before the check's result is known without destroying the state that check refers to (tiles_pipelined)
    float *spare = nullptr;
    hipEvent_t ev_blk[2] = {nullptr, nullptr};
    hipStream_t stream = nullptr;
    int rows_per_task = 0;         // 0 = automatic
    // Task height of the fused passes, measured on this grid (tune_fused_rows): [0] two Jacobi iterations (tol), [1] two
    // red-black iterations (tol), [2] two red-black iterations (precise / fast).  0 = not measured yet, -1 = not to be measured.
    int tuned_rows[3] = {0, 0, 0};
    unsigned finish_from = 0;      // harmonic_execute_gpu, tol math: first iteration of the finishing phase of the latest call (0: none)
    int math = 0;                  // 0 = precise (default), 1 = fast, 2 = traffic, 4 = tol; EPIC_HIP_MATH / epic_hip_set_math_mode
    // Launch-bound grids replay the plain sweeps between two checks from a captured hipGraph; key = (count, starting
    // buffer, starting parity, math, scheme, rows_per_task, fused-pass configuration) -- everything a captured launch
    // sequence depends on.
    struct Replay { hipGraphExec_t exec; int cur_flip; double work; int tile_delta_n; };  // cur_flip: whether the sequence ends in the other buffer; work: what it adds to work_full; tile_delta_n: per-tile maxima its check writes
    std::map<std::tuple<unsigned, int, int, int, int, int, int>, Replay> graphs;
    bool graphs_broken = false;    // a capture / instantiate / launch failed once: batches run eagerly from then on
    // Activity tracking (struct Track above): one set of work lists for the grid, or one per slab in multi-device mode.
    int track_mode = 2;            // 0 off, 1 on, 2 automatic (on for grids above 4 Mcell): EPIC_HIP_TRACK / epic_hip_set_activity_tracking
    bool track = false;            // the mode resolved for the current dimensions (resolve_tracking)
    Track trk;
    // Red-black, precise / fast math, 2-D, one device: tracked relaxations run PAIRS of iterations as list-driven fused passes
    // (rb_fused2d_kernel<.., TRACK>), whose tiles are not the plain sweep's: their own lists.  last_lists says whose lists
    // describe the field as it is: 0 nobody's (the next tracked launch of either kind runs every tile), 1 the plain sweep's,
    // 2 the fused pass's.
    Track trk_f;
    int last_lists = 0;
    // The work lists of a red-black run say which tiles the NEXT colour has to recompute.  A caller may renumber its iterations
    // (currentIteration is its field): if the iteration about to run has the colour of the one that ran last, those lists are for
    // the wrong colour -- that iteration changes nothing, lists nothing, and its successor would find every tile at rest although the
    // cells it reads have moved.  seq_next = the iteration number the lists in force expect next (note_iterations, driver_enqueue.hip);
    // found by the script fuzz of round 6 (caller-set iteration numbers under EPIC_HIP_TRACK=1).
    unsigned seq_next = 0;
    bool seq_valid = false;
    int pair_rows = 0;             // task height of the tracked pass in use (0: not chosen yet); see rb_pairs_choose_rows
    static constexpr size_t kL = Track::kL, kCS = Track::kCS;
    // Work accounting (epic_hip_work_done): whole-grid iterations' worth of cells recomputed since the last reset.  Launches
    // that run every tile count 1 (a fused pass 2) on the host; list-driven launches add their tile counts on the device.
    double work_full = 0.0;
    bool redblack = true;          // scheme: true = the reference's in-place red-black half-sweeps (default: with the precise math that is
                                   // harmonic_complete_cpu bit for bit), false = Jacobi ping-pong (EPIC_HIP_SCHEME=jacobi / epic_hip_set_scheme)
    // Multi-device mode (EPIC_HIP_DEVICES=0,1,...): the grid is cut along its slowest axis -- rows of a 2-D grid, planes of a
    // 3-D one: "units" -- into one slab per listed device, every interior side carries `halo` ghost units that are swept like
    // owned ones and traded every `halo` iterations (see the "several devices in one process" section below).  buf / maskw /
    // d_delta / stream / trk above then stay unused.
    struct Slab {
        int dev = 0;                 // HIP device ordinal (the list may name a device more than once)
        int lo = 0, hi = 0;          // owned global units [lo, hi)
        int g_top = 0, g_bot = 0;    // ghost units above / below
        int rows = 0;                // local units, ghosts included
        float *buf[2] = {nullptr, nullptr};
        uint32_t *maskw = nullptr;
        unsigned *d_delta = nullptr;
        float *h_delta = nullptr;    // pinned
        hipStream_t stream = nullptr, comm = nullptr;   // sweeps / boundary bands + halo copies
        hipEvent_t ev_prev = nullptr, ev_band = nullptr, ev_comm = nullptr, ev_stage = nullptr;
        Track trk;                   // this slab's work lists (single sweeps)
        Track trk_f;                 // ... and those of the fused passes' tiling (tracked pairs on slabs: multi_run_pairs; Ctx::last_lists says whose are current)
        bool peer_up = true;         // the seam to the slab above: direct device-to-device copies (else through `bounce`)
        float *bounce[2] = {nullptr, nullptr};   // pinned staging of that seam: [0] downwards (into this slab), [1] upwards
        int first() const { return g_top; }              // first owned local row
        int last() const { return rows - 1 - g_bot; }    // last owned local row
        int top() const { return lo - g_top; }           // global row of local row 0
    };
    std::vector<Slab> slabs;
    // epic_hip_multi_report: timing events of ONE exchange iteration per slab (interior sweep on the compute stream; boundary
    // bands + halo copies on the second stream), recorded only while a probe is armed
    struct Probe { hipEvent_t int0 = nullptr, int1 = nullptr, cp0 = nullptr, cp1 = nullptr; };
    std::vector<Probe> probe;      // empty: not armed
    bool plan_failed = false;      // multi_plan: the slabs of a usable device list could not be given their streams / events / pinned buffers
    int slab_n = 0;                // dimensionality the slabs were laid out for
    Crew *crew = nullptr;          // one issuing thread per slab (null: the caller's thread issues everything)
    std::vector<int> devices;      // EPIC_HIP_DEVICES as given (validated); fewer than 2 entries: single-device mode
    int halo = 8, since = 0;       // ghost rows per interior side in use; iterations since the last exchange
    int halo_env = 0;              // EPIC_HIP_HALO (0 = not given: chosen by slab height in multi_plan)
    Config cfg;                    // every EPIC_HIP_* knob as the environment had it when this context was created (driver_config.h)
    // Deferred iterations (driver_loop.hip, "harmonic_update_gpu counts"): plain iterations the caller has asked for one call at a
    // time and that are not enqueued yet -- iterations [pending_first, pending_first + pending).  They are enqueued as ONE block (LDS
    // tiles, fused pairs, a captured graph: whatever a block of harmonic_execute_gpu takes) when `defer_cap` of them have been
    // counted and at every ordering point of the boundary (flush_pending).  defer_bypass: whether those blocks run without the
    // work lists, decided at the latest check (bypass_lists_for_batch reads counters back: not once per block of two iterations).
    unsigned pending = 0, pending_first = 0, pending_cap = 1;
    bool defer_bypass = false;
    // Run-ahead at the check (small grids on the tile path; driver_loop.hip): the caller's thread has to wait for a check's result,
    // and while it does -- and until its next plain updates have been counted and enqueued -- the device would sit idle, 40 % of a
    // tick on the reference's maps.  A caller that followed its previous check with plain updates is expected to do so again: the
    // first block after the check (`count` iterations from iteration `first`, ONE tile launch from the current buffer into
    // Ctx::spare) is enqueued behind the check before the wait.  When the caller has then counted exactly those iterations the
    // block is adopted (the buffers change roles, nothing is launched); anything else -- a read-back, an edit, another check, a
    // renumbered iteration -- discards it: the state the check refers to was never touched.
    struct Ahead { bool live = false; unsigned first = 0, count = 0; } ahead;
    unsigned tick_plain = 0;       // plain updates counted since the latest harmonic_update_and_check_gpu
    bool multi() const { return !slabs.empty(); }
    size_t u_bytes() const { return (size_t)rows * pitch * sizeof(float); }
    // 2-D: the lane masks are kept twice in one block -- the standard layout, and behind it the same bits cut for the fused
    // passes' lane -> column mapping (kernels.h: fused layout; derived on the device after every upload and edit)
    static size_t mask_words_both_2d(int rows, int pitch) { return epic_hip::mask_words_2d(rows, pitch) + epic_hip::mask_words_fused_2d(rows, pitch); }
    size_t mask_bytes() const
    {
        if (n == 4) return 64;   // (nothing is ever swept: d_locked only has to be a live allocation)
        return sizeof(uint32_t) * (n == 2 ? mask_words_both_2d(rows, pitch) : epic_hip::mask_words_3d(m[0], m[1], pitch));
    }
    uint32_t *maskf() const { return n == 2 && maskw ? maskw + epic_hip::mask_words_2d(rows, pitch) : nullptr; }
    uint32_t *maskf(const Slab &sl) const { return sl.maskw ? sl.maskw + epic_hip::mask_words_2d(sl.rows, pitch) : nullptr; }
};

constexpr float kTolFinishOptionalBelow = 1e-5f;   // EPIC_HIP_TOL_FINISH=0 is honoured for epsilon <= this (harmonic_execute_gpu)
constexpr size_t kTileDeltaCap = 4096;   // tiles of a launch whose check may go through Ctx::h_tile_delta

// ---- driver_registry.hip ------------------------------------------------------------------------------------------------
void report(const char *fn, const char *msg);   // "Error[<fn>]: <msg>" on stderr, the reference's convention
void free_spare(Ctx *c);
Ctx *find_ctx(Harmonic *h);
bool dims_from(const Harmonic *h, Ctx *c);
bool dims_into_ctx(const Harmonic *h, Ctx *c);
bool same_dims(const Harmonic *h, const Ctx *c);
Ctx *get_ctx(Harmonic *h, bool create);
void drop_ctx_if_empty(Harmonic *h);
bool ready(const Harmonic *h, const Ctx *c);
float *current_u(const Ctx *c);
bool has_delta(const Ctx *c);
int upload_u(Harmonic *h, Ctx *c, const char *fn);
int upload_locked(Harmonic *h, Ctx *c, const char *fn);
void apply_config(Ctx *c);                      // the mode fields of a context (math, scheme, tracking, task height, halo, devices) from c->cfg

// ---- driver_plan.hip ----------------------------------------------------------------------------------------------------
void resolve_tracking(Ctx *c);
int auto_rows_per_task(const Ctx *c);
int fused_rows_per_task(const Ctx *c);
long long fuse_from_cells(const Ctx *c);       // untracked fused pairs from this many cells (rows x pitch) on
long long tile_up_to_cells(const Ctx *c);      // LDS tiles up to this many cells (rows x cols)
long long tracked_pairs_from_cells(const Ctx *c);
bool fuses_tol(const Ctx *c);
bool fuses_jacobi(const Ctx *c);
bool fuses_rb_tol(const Ctx *c);
bool fuses_rb_precise(const Ctx *c);            // red-black, precise / fast math: pairs of plain iterations as rb_fused2d_kernel
int jacobi_fused_rows_per_task(const Ctx *c);
void tune_fused_rows(Ctx *c, int kind, unsigned iteration);
epic_hip::TilePlan tile_plan(const Ctx *c);
bool tile_checks(const Ctx *c, const epic_hip::TilePlan &tp);
bool rb_pairs_tracked(const Ctx *c);
bool rb_pairs_tracked_multi(const Ctx *c);      // the same on the slabs of the multi-device mode (2-D, at least two ghost rows)
int rb_pairs_rows_per_task(const Ctx *c);
void rb_pairs_choose_rows(Ctx *c);
bool bypass_lists_for_batch(Ctx *c, bool pairs = false);
const char *plain_batch_path(const Ctx *c);     // the kernel family a batch of plain iterations takes now (epic_hip_config_dump)

// ---- driver_enqueue.hip -------------------------------------------------------------------------------------------------
void note_iterations(Ctx *c, unsigned first, unsigned count);   // iterations [first, first + count) are about to be enqueued (see Ctx::seq_next)
hipError_t enqueue_sweep(Ctx *c, bool check, unsigned iteration);
hipError_t enqueue_check_sweep(Ctx *c, unsigned iteration);   // a check iteration as the context's scheme and EPIC_HIP_JACOBI_CHECKS have it
void fold_listed_work(Ctx *c);
void drop_graphs(Ctx *c);  // captured launch sequences hold the buffer addresses: drop them whenever a buffer goes away
hipError_t enqueue_plain_run(Ctx *c, unsigned count, unsigned first, bool check_last = false);
hipError_t enqueue_rb_pairs_tracked(Ctx *c, unsigned npairs, unsigned first, bool check_last, bool bypass);
hipError_t enqueue_plain_batch(Ctx *c, unsigned count, unsigned first, bool check_last = false);
bool tiles_pipeline_ready(Ctx *c);
hipError_t enqueue_ahead(Ctx *c, unsigned first);   // one tile launch of plain iterations from the current buffer into Ctx::spare (Ctx::ahead)
int read_tile_delta(Harmonic *h, Ctx *c, const char *fn, hipEvent_t after = nullptr);   // after: the event behind the check launch (else: the whole stream)
int read_delta(Harmonic *h, Ctx *c, const char *fn);
void force_all(Ctx *c);    // the next two iterations run every tile (after any change of values, masks, mode or tiling)
bool due_tiles(Ctx *c, unsigned long long *due, unsigned long long *tiles, bool forced_runs_all);

// ---- driver_loop.hip ----------------------------------------------------------------------------------------------------
// `plain` unchecked iterations from iteration `first` and then -- check -- one check iteration whose max |du| is read back into
// h->delta, on the kernel family the context's state calls for.  bypass: -1 decide here (reads the list counters back), 0 / 1 given.
// Does not touch h->currentIteration (the callers count); an EPIC_* code.
int run_block(Harmonic *h, Ctx *c, unsigned plain, unsigned first, bool check, const char *fn, int bypass, bool run_ahead = false);
// EPIC_HIP_JACOBI_CHECKS=reference on a context that runs the Jacobi scheme right now: its check iterations are the reference's half-sweeps (run_block)
inline bool jacobi_reference_checks(const Ctx *c) { return c->cfg.jacobi_ref_checks && !c->redblack && c->n != 4; }
unsigned defer_cap(const Harmonic *h, const Ctx *c);   // deferred iterations that make a block worth enqueueing (1: no deferral)
int flush_pending(Harmonic *h, Ctx *c, const char *fn);   // c may be null; an EPIC_* code (the failure of a deferred launch surfaces here)

// ---- driver_multi.hip (EPIC_HIP_DEVICES: see that file) -------------------------------------------------------------------
bool multi_plan(Ctx *c);
void multi_destroy(Ctx *c);
bool multi_holds_anything(const Ctx *c);
bool multi_ready(const Ctx *c);
void multi_sync(Ctx *c);
void multi_free_u(Ctx *c);
void multi_free_mask(Ctx *c);
void multi_free_delta(Ctx *c);
size_t unit_floats(const Ctx *c);
size_t unit_rows(const Ctx *c);
size_t slab_mask_words(const Ctx *c, const Ctx::Slab &sl);
bool multi_has_threads(const Ctx *c);
hipError_t multi_sweep(Ctx *c, bool check, unsigned iteration);
hipError_t multi_run(Ctx *c, unsigned count, unsigned first, bool check_first);
hipError_t multi_run_pairs(Ctx *c, unsigned npairs, unsigned first, bool check_last);   // list-driven fused passes on every slab (rb_pairs_tracked_multi)
int multi_read_delta(Harmonic *h, Ctx *c, const char *fn);
int multi_upload_u(Harmonic *h, Ctx *c, const char *fn);
int multi_upload_locked(Harmonic *h, Ctx *c, const char *fn);
int multi_get_values(Harmonic *h, Ctx *c, const char *fn);
int multi_set_cells(Ctx *c, unsigned k, const unsigned *v, const unsigned *types, const char *fn);

}  // namespace epic_drv
