// harmonic_gpu.hip -- host drivers of the GPU half of the libepic C-ABI, HIP runtime on MI355X.
//
// Replaces libepic/src/harmonic/harmonic_gpu.cu:156-434 (solver drivers), harmonic_model_gpu.cu:34-204
// (device-state lifecycle) and harmonic_utilities_gpu.cu:66-138 (sparse edits) with the same exported names,
// validation, return codes and "Error[<function>]: <text>" stderr lines, over a different device design:
//
//  * device state lives in a library-side context keyed by the caller's Harmonic* (the 80-byte struct has no
//    room for a second ping-pong buffer, a stream or pinned readback memory).  The struct's d_* fields are
//    still set non-null / nulled exactly where the reference does, because callers and the library itself
//    null-test them (harmonic_gpu.cu:208, :232-235; harmonic_model_gpu.cu:174-176);
//  * u is kept pitched (row length padded to 256 floats) in two buffers; d_u points at the current one;
//  * locked is kept bit-packed (d_locked points at the packed words);
//  * sweeps are enqueued on one non-blocking stream; only the check sweeps, the readbacks and the edits
//    synchronise (the reference synchronises the whole device after every kernel).
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
#include "kernels.h"

using epic::Harmonic;

namespace {

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
        Track trk;                   // this slab's work lists
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
constexpr long long kTileMaxCellsDefault = 3ll << 20;   // EPIC_HIP_TILE_MAX_CELLS: grids up to this many cells take the tile path
constexpr size_t kTileDeltaCap = 4096;   // tiles of a launch whose check may go through Ctx::h_tile_delta

std::mutex g_mu;
std::unordered_map<Harmonic *, Ctx *> g_ctx;

void resolve_tracking(Ctx *c);
void force_all(Ctx *c);    // the next two iterations run every tile (after any change of values, masks, mode or tiling)
bool due_tiles(Ctx *c, unsigned long long *due, unsigned long long *tiles, bool forced_runs_all);
void fold_listed_work(Ctx *c);
void drop_graphs(Ctx *c);  // captured launch sequences hold the buffer addresses: drop them whenever a buffer goes away
bool multi_plan(Ctx *c);   // multi-device mode (EPIC_HIP_DEVICES): see "several devices in one process" below
void multi_destroy(Ctx *c);
bool multi_holds_anything(const Ctx *c);
bool multi_ready(const Ctx *c);

void report(const char *fn, const char *msg) { fprintf(stderr, "Error[%s]: %s\n", fn, msg); }

void free_spare(Ctx *c)   // the third u buffer of the small-grid path goes wherever the two others go
{
    if (c->spare) (void)hipFree(c->spare);
    c->spare = nullptr;
}

Ctx *find_ctx(Harmonic *h)
{
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_ctx.find(h);
    return it == g_ctx.end() ? nullptr : it->second;
}

bool dims_from(const Harmonic *h, Ctx *c)
{
    // n = 4: the reference holds the state on the device and sweeps NOTHING (harmonic_gpu.cu:156-162, :327-336 -- the n == 4
    // branches are empty, currentIteration still counts); the same here: state resident, every update a counting no-op.
    if (h->n != 2 && h->n != 3 && h->n != 4) return false;
    for (unsigned i = 0; i < h->n; i++)
        if (h->m[i] < (h->n == 4 ? 1u : 3u) || h->m[i] > (1u << 30)) return false;
    c->n = (int)h->n;
    for (unsigned i = 0; i < 4; i++) c->m[i] = i < h->n ? (int)h->m[i] : 0;
    c->cols = c->m[c->n - 1];
    long long rows = 1;
    for (int i = 0; i + 1 < c->n; i++) {
        rows *= c->m[i];
        if (rows > 0x7fffffffLL) return false;
    }
    c->rows = (int)rows;
    c->pitch = epic_hip::pitch_for_cols(c->cols);
    resolve_tracking(c);
    return true;
}

// dims_from() on the context that owns device state: also (re)decides single- or multi-device mode.  Changing dimensions
// while one kind of state is still resident is refused by the callers (same_dims), so the layout never changes under
// live buffers.
bool dims_into_ctx(const Harmonic *h, Ctx *c)
{
    const int rows0 = c->rows, cols0 = c->cols, n0 = c->n;
    if (!dims_from(h, c)) return false;
    // the measured task heights belong to ONE grid (the context survives a re-initialisation with other dimensions)
    if (c->rows != rows0 || c->cols != cols0 || c->n != n0) c->tuned_rows[0] = c->tuned_rows[1] = c->tuned_rows[2] = c->pair_rows = 0;
    if (!c->devices.empty()) multi_plan(c);
    return true;
}

bool same_dims(const Harmonic *h, const Ctx *c)
{
    if ((int)h->n != c->n) return false;
    for (unsigned i = 0; i < h->n; i++)
        if ((int)h->m[i] != c->m[i]) return false;
    return true;
}

// Create (or fetch) the context of this Harmonic; sets up the stream and the pinned readback word.
Ctx *get_ctx(Harmonic *h, bool create)
{
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_ctx.find(h);
    if (it != g_ctx.end()) {
        Ctx *c = it->second;
        // A Harmonic whose fields are all null but which we still track was freed and re-created by the
        // caller without uninitialize: drop the stale device state.
        if (h->d_m == nullptr && h->d_u == nullptr && h->d_locked == nullptr && h->d_delta == nullptr &&
            (c->buf[0] || c->maskw || c->d_m || c->d_delta || multi_holds_anything(c))) {
            drop_graphs(c);
            if (c->multi()) multi_destroy(c);
            for (float *&b : c->buf) { if (b) (void)hipFree(b); b = nullptr; }
            free_spare(c);
            if (c->maskw) (void)hipFree(c->maskw);
            if (c->d_m) (void)hipFree(c->d_m);
            if (c->d_delta) (void)hipFree(c->d_delta);
            c->maskw = nullptr; c->d_m = nullptr; c->d_delta = nullptr;
        }
        return c;
    }
    if (!create) return nullptr;
    Ctx *c = new Ctx();
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        (void)hipGetLastError();
        delete c;
        return nullptr;
    }
    if (hipHostMalloc((void **)&c->h_delta, 64, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipStreamDestroy(c->stream);
        delete c;
        return nullptr;
    }
    if (hipHostMalloc((void **)&c->h_tile_delta, 2 * kTileDeltaCap * sizeof(float), hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();   // (the tile path then checks with the plain sweep)
        c->h_tile_delta = nullptr;
    }
    const char *e = getenv("EPIC_HIP_ROWS_PER_TASK");
    if (e) c->rows_per_task = atoi(e);
    e = getenv("EPIC_HIP_MATH");
    if (e && strcmp(e, "fast") == 0) c->math = 1;
    if (e && strcmp(e, "tol") == 0) c->math = 4;
    e = getenv("EPIC_HIP_SCHEME");
    if (e && strcmp(e, "redblack") == 0) c->redblack = true;
    if (e && strcmp(e, "jacobi") == 0) c->redblack = false;
    e = getenv("EPIC_HIP_TRACK");
    if (e && (strcmp(e, "0") == 0 || strcmp(e, "1") == 0)) c->track_mode = atoi(e);
    e = getenv("EPIC_HIP_HALO");
    if (e && atoi(e) >= 1) c->halo_env = atoi(e);
    e = getenv("EPIC_HIP_DEVICES");
    if (e && *e) {  // "0,1,2,3"; a device may be named more than once ("0,0,0,0": four slabs on one GPU)
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess) { (void)hipGetLastError(); ndev = 0; }
        std::vector<int> devs;
        bool ok = true;
        for (const char *p = e; *p;) {
            char *end = nullptr;
            const long d = strtol(p, &end, 10);
            if (end == p || d < 0 || d >= ndev) { ok = false; break; }
            devs.push_back((int)d);
            p = (*end == ',') ? end + 1 : end;
            if (*end && *end != ',') { ok = false; break; }
        }
        if (ok && devs.size() <= 64) c->devices = devs;
        else fprintf(stderr, "Warning[epic_hip]: EPIC_HIP_DEVICES=%s ignored (%d device(s) visible)\n", e, ndev);
    }
    g_ctx[h] = c;
    return c;
}

void drop_ctx_if_empty(Harmonic *h)
{
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_ctx.find(h);
    if (it == g_ctx.end()) return;
    Ctx *c = it->second;
    if (c->buf[0] || c->maskw || c->d_m || c->d_delta || multi_holds_anything(c)) return;
    drop_graphs(c);
    if (c->multi()) multi_destroy(c);
    c->trk.release();
    c->trk_f.release();
    if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
    if (c->h_delta) (void)hipHostFree(c->h_delta);
    if (c->h_tile_delta) (void)hipHostFree(c->h_tile_delta);
    free_spare(c);
    for (hipEvent_t &e : c->ev_blk) { if (e) (void)hipEventDestroy(e); e = nullptr; }
    delete c;
    g_ctx.erase(it);
}

// Tracking pays where a sweep is long enough to hide the list handling: grids above 4 Mcell.  The ROS maps (0.1-1 Mcell)
// are launch-bound either way (3.2-4 us per sweep without, 5-6.7 us with lists, measured), so "automatic" leaves them alone.
void resolve_tracking(Ctx *c)
{
    c->track = c->n != 4 && (c->track_mode == 1 || (c->track_mode == 2 && (long long)c->rows * c->pitch > (1ll << 22)));
}

int auto_rows_per_task(const Ctx *c)
{
    if (c->rows_per_task > 0) return c->rows_per_task;
    const long long nstrips = (c->pitch + 255) / 256;
    // The kernel is VALU-bound (precise math): a full sweep of 8192^2 takes the same time with 8 to 32 rows per task
    // (2 extra halo rows per task are cheap).  With activity tracking the tile is also the unit of skipping, and a sweep
    // with few listed tiles costs the march of its longest task (one wave alone: ~1.3 us per row), so shorter tasks help
    // the tail of a relaxation, while very short ones pay the per-task prologue too often (profiles/r01_experiments.txt).
    // Aim at >= 32768 wave-tasks (8 rows per task at 8192^2).
    // Small grids (the ROS maps are 0.1-1 Mcell) cannot fill the chip at all: there one row per wave is best
    // (310 x 940: 4.3 us per sweep at 1 row per task vs 11.4 us at 8, both measured).
    long long r = (long long)c->rows * nstrips / 32768;
    // tol: the row loop runs in trips of 10 rows (kernels_2d.hip), anything else goes through its slower ragged loop.  One
    // trip per task wherever that still gives every wave slot of the chip a task (256 CUs x 24 waves); measured, us per
    // sweep at 1 / 2 / 4 / 8 / 10 / 20 rows per task (profiles/r02_rows_per_task_tol.txt): 4096^2 45 / 39 / 34 / 36 / 31 / 33,
    // 6144 x 8192 112 / 98 / 92 / 90 / 77 / 77, 8192^2 147 / 128 / 122 / 117 / 99 / 110 (30, 40, 60: slower still);
    // 2048^2 and below keep the rule for small grids (10.8 / 10.7 / 10.5 / 12.7 / 11.1 / 14.7 at 2048^2).
    constexpr int trip = epic_hip::kTolTripRows;  // 10
    if (c->math == 4 && (long long)c->rows / trip * nstrips >= 6144) return trip;
    if (r >= 8) r = r / 8 * 8;
    else if (r >= 4) r = 4;
    else if (r >= 2) r = 2;
    else r = 1;
    return (int)std::min<long long>(64, r);
}

bool ready(const Harmonic *h, const Ctx *c)
{
    if (c && c->multi()) return multi_ready(c) && h->d_u && h->d_locked;
    return c && c->buf[0] && c->buf[1] && c->maskw && h->d_u && h->d_locked;
}

float *current_u(const Ctx *c) { return c->multi() ? c->slabs[0].buf[c->cur] : c->buf[c->cur]; }
bool has_delta(const Ctx *c) { return c->multi() ? c->slabs[0].d_delta != nullptr : c->d_delta != nullptr; }

// One iteration, enqueued: a Jacobi sweep (buffers swap) or, in the red-black scheme (2-D), the reference's half-sweep
// of the colour selected by `iteration` in place.  check != 0 also zeroes and fills the device delta word.
hipError_t multi_sweep(Ctx *c, bool check, unsigned iteration);

hipError_t enqueue_sweep(Ctx *c, bool check, unsigned iteration)
{
    if (c->n == 4) return hipSuccess;   // the reference's empty n == 4 branch: nothing is swept, the caller counts
    if (c->multi()) return multi_sweep(c, check, iteration);
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
                                                 (int)(iteration & 1u), check ? c->d_delta : nullptr, c->stream, &act));
    }
    if (c->n == 2)
        e = advance(epic_hip::launch_sweep_2d(in, out, c->maskw, c->rows, c->pitch, 0, c->rows, auto_rows_per_task(c),
                                              c->math, -1, check ? c->d_delta : nullptr, c->stream, &act));
    else
        e = advance(epic_hip::launch_sweep_3d(in, out, c->maskw, c->m[0], c->m[1], c->pitch, 0, c->m[0], c->math, -1,
                                              check ? c->d_delta : nullptr, c->stream, &act));
    if (e == hipSuccess) c->cur ^= 1;
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
        if (hipSetDevice(sl.dev) == hipSuccess) fold(sl.trk, sl.stream, (double)sl.rows / (double)units);
}

void drop_graphs(Ctx *c)
{
    for (auto &g : c->graphs) (void)hipGraphExecDestroy(g.second.exec);
    c->graphs.clear();
}

// `count` plain (unchecked) iterations starting at iteration number `first`.  Kernels of a few microseconds are
// bound by the host's launch rate (~4 us each); for those the sequence is captured once into a hipGraph and replayed
// (inter-kernel gap ~1.5 us, MI355X_MICROARCH.md "boundary" row).  Large grids keep the plain launches.
// Red-black, 2-D, large grids: two consecutive plain iterations run as ONE fused pass (kernels_2d.hip,
// rb_fused2d_kernel: 4 B of HBM traffic per cell-update instead of 16); an odd iteration left over is an in-place
// half-sweep.  Small grids keep the half-sweeps (a fused task recomputes 2 extra rows, too much at 1-2 rows per task).
int fused_rows_per_task(const Ctx *c)
{
    if (c->rows_per_task > 0) return std::max(c->rows_per_task, 4);
    if (c->tuned_rows[2] > 0) return c->tuned_rows[2];   // (several devices: measured on the first slab)
    const long long nstrips = (c->pitch + 247) / 248;
    long long r = (long long)c->rows * nstrips / 8192 / 8 * 8;
    return (int)std::min<long long>(64, std::max<long long>(16, r));
}

// Red-black with the precise / fast math: from how many cells (rows x pitch) on two plain iterations run as one fused pass
// (rb_fused2d_kernel).  EPIC_HIP_FUSE_MIN_CELLS overrides (the tests set 0).
long long rb_fuse_min_cells()
{
    const char *e = getenv("EPIC_HIP_FUSE_MIN_CELLS");
    return e ? atoll(e) : (1ll << 22);
}

// Whether two consecutive plain Jacobi iterations run as one fused pass in the context's current configuration.
// EPIC_HIP_FUSE_MIN_CELLS: grids below it keep the single sweeps (default 4 Mcell: below, a sweep is launch-bound and the
// fused pass's extra rows cost more than the second launch; read per batch, not cached -- the tests switch it).
bool fuses_tol(const Ctx *c)   // either scheme
{
    if (getenv("EPIC_HIP_NO_FUSE") != nullptr) return false;
    const char *e = getenv("EPIC_HIP_FUSE_MIN_CELLS");
    const long long min_cells = e ? atoll(e) : (1ll << 22);
    return c->n == 2 && !c->track && c->math == 4 && (long long)c->rows * c->pitch >= min_cells;
}
bool fuses_jacobi(const Ctx *c) { return !c->redblack && fuses_tol(c); }
// red-black, tol math: both colours in one pass (rb_tol_fused2d_kernel); one device only
bool fuses_rb_tol(const Ctx *c) { return c->redblack && fuses_tol(c); }
// (on several devices a pass leaves two more ghost units stale, so neither of its two iterations may be one that ends with an
// exchange: multi_run fuses inside the stretches between exchanges only)

// Rows per task of the fused Jacobi pass (kernels.h: jacobi_fused_auto_rows), per device in multi-device mode.
int jacobi_fused_rows_per_task(const Ctx *c)
{
    const char *e = getenv("EPIC_HIP_FUSED_ROWS");  // experiment / test knob
    if (e && atoi(e) > 0) return atoi(e);
    const int tuned = c->tuned_rows[c->redblack ? 1 : 0];
    if (tuned > 0) return tuned;   // (several devices: measured on the first slab)
    const long long rows = c->multi() ? c->rows / (long long)c->slabs.size() : c->rows;
    return epic_hip::jacobi_fused_auto_rows((int)rows, c->pitch);
}

// The time of a fused pass depends on its task height in a way no rule of ours predicts: at 8192 x 8192 (tol Jacobi, one
// box, us per launch) 38 rows 182.2, 39: 179.5, 40: 167.9, 41: 169.6, 42: 172.9, 44: 178.6, 46: 170.1, 64: 184.4, against 178.6 for
// the 23 of jacobi_fused_auto_rows (profiles/r03_experiments.txt item 9) -- rounds of resident waves, the XCD bands and the
// memory channels all have a say.  So the height is MEASURED, once per grid and kind of pass, the first time such a pass
// is about to run on a grid of at least 4 Mcell: every candidate runs three times from the current buffer into the other
// one (which the next real pass overwrites anyway; nothing else is touched) between two events, ~10 ms in all.  Results do
// not depend on the height (tests/test_gpu_tol.py, test_gpu_parity.py sweep it).  EPIC_HIP_TUNE=0: the rules only.
void tune_fused_rows(Ctx *c, int kind, unsigned iteration)
{
    if (c->tuned_rows[kind] != 0) return;
    // not on the field of the first iterations (all cells at the initial value: the passes run up to 15 % faster on it and rank the
    // heights differently): the rule serves until the front has crossed a good part of the grid
    if (iteration < (unsigned)std::min(c->rows, c->cols) / 2) return;
    c->tuned_rows[kind] = -1;
    const char *t = getenv("EPIC_HIP_TUNE");
    if ((t && t[0] == '0') || c->n != 2 || c->rows_per_task > 0 || getenv("EPIC_HIP_FUSED_ROWS") != nullptr) return;
    // several devices: the first slab stands for all (they are of one size within a row), on its own device and stream
    const bool multi = c->multi();
    DeviceGuard restore_device;
    if (multi && hipSetDevice(c->slabs[0].dev) != hipSuccess) { (void)hipGetLastError(); return; }
    const int rows = multi ? c->slabs[0].rows : c->rows;
    const hipStream_t stream = multi ? c->slabs[0].stream : c->stream;
    const float *in = multi ? c->slabs[0].buf[c->cur] : c->buf[c->cur];
    float *out = multi ? c->slabs[0].buf[c->cur ^ 1] : c->buf[c->cur ^ 1];
    const uint32_t *maskw = multi ? c->slabs[0].maskw : c->maskw, *maskf = multi ? c->maskf(c->slabs[0]) : c->maskf();
    if ((long long)rows * c->pitch < (1ll << 22)) return;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) { c->tuned_rows[kind] = 0; return; }
    const int dflt = kind == 2 ? fused_rows_per_task(c) : jacobi_fused_rows_per_task(c);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess) return;
    if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); return; }
    auto launch = [&](int height) -> hipError_t {
        if (kind == 2) return epic_hip::launch_rb_fused_2d(in, out, maskw, rows, c->pitch, height, c->math, 0, stream, maskf);
        return epic_hip::launch_jacobi_fused_2d(in, out, maskw, rows, c->pitch, height, c->math, stream, kind == 1 ? 0 : -1, maskf);
    };
    auto timed = [&](int height, float *ms) -> bool {
        if (launch(height) != hipSuccess) return false;   // warm
        if (hipEventRecord(e0, stream) != hipSuccess) return false;
        for (int i = 0; i < 2; ++i)
            if (launch(height) != hipSuccess) return false;
        return hipEventRecord(e1, stream) == hipSuccess && hipEventSynchronize(e1) == hipSuccess &&
               hipEventElapsedTime(ms, e0, e1) == hipSuccess;
    };
    static const int kCandidates[] = {20, 23, 26, 29, 32, 35, 38, 40, 41, 43, 46, 49, 52, 58, 64, 80, 96, 128};   // (the tall ones: grids of many rounds, 32768^2)
    const bool say = getenv("EPIC_HIP_TUNE_DEBUG") != nullptr;   // the table on stderr
    float best_ms = 0.0f, dflt_ms = 0.0f;
    int best = 0;
    bool ok = timed(dflt, &dflt_ms);
    if (say && ok) fprintf(stderr, "[epic_hip tune] kind %d, %d x %d%s: rule %d rows %.1f us", kind, rows, c->cols, multi ? " (first slab)" : "", dflt, dflt_ms * 500.0f);
    for (int r : kCandidates) {
        if (!ok) break;
        if (r == dflt || r > rows) continue;
        float ms = 0.0f;
        ok = timed(r, &ms);
        if (say && ok) fprintf(stderr, ", %d: %.1f", r, ms * 500.0f);
        if (ok && (best == 0 || ms < best_ms)) { best = r; best_ms = ms; }
    }
    if (say) fprintf(stderr, "\n");
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (!ok) { (void)hipGetLastError(); return; }
    c->tuned_rows[kind] = (best > 0 && best_ms < 0.99f * dflt_ms) ? best : dflt;   // a clear win only
}

hipError_t multi_run(Ctx *c, unsigned count, unsigned first, bool check_first);

// Small 2-D grids (one device, no work lists): the plain iterations between two checks run several per launch on tiles that
// stay in LDS (kernels_tile2d.hip) -- the reference's maps are launch-bound, 2.5-3 us per half-sweep whatever it computes.
// halo == 0: not for this context.  Knobs (read per batch; the tests switch them): EPIC_HIP_TILE=0 off; EPIC_HIP_TILE_HALO = ghost
// rings = iterations per launch, EPIC_HIP_TILE_WIDTH = 64 | 128 (default for both: a cost model of the launch, below);
// EPIC_HIP_TILE_ROWS = owned rows per tile; EPIC_HIP_TILE_MAX_CELLS (default 3 Mcell; from 4 Mcell up the fused passes take over).
// Where it pays (tools/tile_probe.py, tools/time_maps.py; profiles/r04_experiments.txt items 1f, 1i): 2.0-2.4 x on the maps up to
// 0.3 Mcell (one narrow tile per CU), 1.1-1.3 x on the 1-2.5 Mcell fixtures (wide tiles).
epic_hip::TilePlan tile_plan(const Ctx *c)
{
    epic_hip::TilePlan none = {0, 0, 0, 0, 0};
    if (c->n != 2 || c->multi() || c->track || c->math == 2) return none;
    if (fuses_tol(c)) return none;   // (a fused pass asked for on a small grid: EPIC_HIP_FUSE_MIN_CELLS, the tests)
    const char *e = getenv("EPIC_HIP_TILE");
    if (e && e[0] == '0') return none;
    e = getenv("EPIC_HIP_TILE_MAX_CELLS");
    const long long max_cells = e ? atoll(e) : kTileMaxCellsDefault;
    if ((long long)c->rows * c->cols > max_cells) return none;
    const char *rows_env = getenv("EPIC_HIP_TILE_ROWS");
    const int want_rows = rows_env ? atoi(rows_env) : 0;
    e = getenv("EPIC_HIP_TILE_WIDTH");   // 64 | 128: one width only (experiments, tests)
    const int only_width = e ? atoi(e) : 0;
    e = getenv("EPIC_HIP_TILE_HALO");
    const int only_halo = e && atoi(e) > 0 ? atoi(e) : 0;
    // Which LDS tile (64 or 128 columns wide), and how many ghost rings = iterations per launch.  Measured (tools/tile_probe.py,
    // profiles/r04_experiments.txt items 1d, 1i): a launch costs ~2.7 us + 0.06 us per 1000 LDS cells to load and store, plus per
    // iteration the larger of a latency chain -- 0.84 us + 0.055 us per pass and SIMD: the barrier, three LDS round trips, one
    // cell update's dependent instructions -- and the issue of the passes themselves (0.21 us per pass and SIMD with the
    // bit-exact arithmetic -- 67 % of full VALU issue --, 0.17 us with tol), as long as every tile has a CU of its own (+60 % as
    // soon as one CU takes two).
    // More rings amortise the launch but leave fewer owned cells per tile; wide tiles waste fewer cells on rings (1.3-1.5 per
    // owned cell instead of 2.3) but give each CU 3-6 times the passes.  The model is evaluated for both widths and a few depths.
    epic_hip::TilePlan best = none;
    double best_cost = 0.0;
    for (int width : {epic_hip::kTile2dCols, epic_hip::kTile2dWideCols}) {
        if (only_width && width != only_width) continue;
        const int max_rows = epic_hip::tile_2d_max_rows(c->math, c->redblack, width);
        if (max_rows == 0) continue;
        for (int halo : {8, 10, 12, 14, 16}) {
            if (only_halo && halo != only_halo && halo != 8) continue;
            const int h = only_halo ? only_halo : halo;
            const epic_hip::TilePlan p = epic_hip::tile_2d_plan(c->rows, c->cols, h, want_rows, width, max_rows);
            if (p.halo == 0) continue;
            const int s_r = p.tile_rows + 2 * h;
            const double units = (c->redblack ? s_r / 2.0 : (double)s_r) * (width / 64), per_simd = units / 4.0;
            const double rounds = std::max(1.0, (double)p.tiles_r * p.tiles_c / 256.0);
            const double step = std::max(0.84 + 0.055 * per_simd, 0.2 + (c->math == 4 ? 0.17 : 0.21) * per_simd);
            const double cost = (rounds > 1.0 ? 1.6 * rounds / 2.0 + 0.2 : 1.0) * ((2.7 + 0.00006 * s_r * width) / h + step);
            if (best.halo == 0 || cost < best_cost) { best = p; best_cost = cost; }
            if (only_halo) break;
        }
    }
    return best;
}

// Whether a check iteration may run as the LAST step of a tile launch (its max |du| per tile into Ctx::h_tile_delta).
bool tile_checks(const Ctx *c, const epic_hip::TilePlan &tp)
{
    return tp.halo > 0 && c->h_tile_delta != nullptr && (size_t)tp.tiles_r * tp.tiles_c <= kTileDeltaCap;
}

// check_last (tile path only, tile_checks()): one more iteration after the `count` plain ones, a check, in the same launches.
hipError_t enqueue_plain_run(Ctx *c, unsigned count, unsigned first, bool check_last = false)
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
    const bool no_fuse = getenv("EPIC_HIP_NO_FUSE") != nullptr;   // (read per call: the tests switch it)
    // (the fused passes have their own 248-column tiling and no work lists: they are used when tracking is off -- or
    //  bypassed for the batch, harmonic_execute_gpu; rb_fused2d_kernel for the precise / fast arithmetic, the RB instance of
    //  the tol pass for tol)
    const bool fuse = c->redblack && c->n == 2 && !no_fuse && !c->track && !c->multi() && c->math != 4 && (long long)c->rows * c->pitch >= rb_fuse_min_cells();
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

// Tracked red-black relaxations with the precise / fast math on one device (2-D, from 4 Mcell up -- EPIC_HIP_FUSE_MIN_CELLS): the
// iterations between two checks AND the check run as pairs, each one list-driven fused pass (kernels_2d.hip: rb_fused2d_kernel
// with TRACK; the check is the second iteration of the last pair).  Against in-place half-sweeps with lists: half the launches
// (10-12 us each with next to nothing due, a quarter of all iterations of the 8192^2 relaxation), and a listed tile moves through
// HBM once for two iterations instead of twice.  EPIC_HIP_TRACK_PAIRS=0: the half-sweeps, as before round 4.
bool rb_pairs_tracked(const Ctx *c)   // (the name is round 4's first form: red-black, precise; the tol passes -- both schemes -- followed)
{
    if (!c->track || c->n != 2 || c->multi() || c->math == 2) return false;
    if (!c->redblack && c->math != 4) return false;   // precise Jacobi has no fused pass
    if (getenv("EPIC_HIP_NO_FUSE") != nullptr) return false;
    const char *e = getenv("EPIC_HIP_TRACK_PAIRS");
    if (e && e[0] == '0') return false;
    e = getenv("EPIC_HIP_FUSE_MIN_CELLS");
    return (long long)c->rows * c->pitch >= (e ? atoll(e) : (1ll << 22));
}
// rows per task of the tracked pass: the unit of skipping, and every task recomputes the first colour of one row above and one
// below its chunk (2 / rows extra arithmetic).  EPIC_HIP_TRACK_PAIR_ROWS overrides.
int rb_pairs_rows_per_task(const Ctx *c)
{
    const char *e = getenv("EPIC_HIP_TRACK_PAIR_ROWS");
    if (e && atoi(e) > 0) return std::max(atoi(e), 2);
    if (c->rows_per_task > 0) return std::max(c->rows_per_task, 2);
    return c->pair_rows > 0 ? c->pair_rows : 16;
}
// Two task heights.  While a good part of the grid is active the tall tasks (16 rows: 2 extra first-colour rows per task, 12 %)
// are right.  With next to nothing due a pass costs the march of ONE task by a wave that is alone on its SIMD -- ~1.4 us per
// row, 26 us per pass at 16 rows (measured: profiles/r04_experiments.txt) -- and the tail of a relaxation is thousands of such
// passes: there 4 rows per task make the pass three times shorter, and recomputing 1.5 x the few cells that are due costs
// nothing.  Changing the height means new lists: the next pass runs every tile once (~0.2 ms), so the decision is taken at a
// check, from the share of tiles that check listed, with a wide hysteresis.
void rb_pairs_choose_rows(Ctx *c)
{
    if (getenv("EPIC_HIP_TRACK_PAIR_ROWS") != nullptr || c->rows_per_task > 0) return;
    if (c->pair_rows == 0) c->pair_rows = 16;
    if (c->last_lists != 2) return;
    unsigned long long due = 0, tiles = 0;
    if (!due_tiles(c, &due, &tiles, false) || tiles == 0) return;
    const double share = (double)due / (double)tiles;
    if (c->pair_rows == 16 && share < 0.04) c->pair_rows = 4;
    else if (c->pair_rows == 4 && share > 0.20) c->pair_rows = 16;
}
// `npairs` pairs of iterations starting at iteration `first`; check_last: the second iteration of the last pair is a check (the
// device delta word is zeroed and filled).  bypass: without the lists (every tile; the untracked pass's own task height).
hipError_t enqueue_rb_pairs_tracked(Ctx *c, unsigned npairs, unsigned first, bool check_last, bool bypass)
{
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

hipError_t enqueue_plain_batch(Ctx *c, unsigned count, unsigned first, bool check_last = false)
{
    if (c->n == 4) return enqueue_plain_run(c, count, first, check_last);
    const bool small = (long long)c->rows * c->pitch <= (1ll << 22);
    const bool no_graph = getenv("EPIC_HIP_NO_GRAPH") != nullptr;  // (read per batch: the tests switch it)
    // a captured sequence bakes in the work-list buffers and list mode: run eagerly until the forced iterations are over
    if (!small || count < 8 || no_graph || c->multi() || (c->track && (c->trk.force > 0 || c->trk.tiles == 0)))
        return enqueue_plain_run(c, count, first, check_last);
    // (the fused-pass switches are read per batch -- EPIC_HIP_NO_FUSE, EPIC_HIP_FUSE_MIN_CELLS, EPIC_HIP_FUSED_ROWS --, so they
    // belong to the key: 0 = single sweeps, otherwise the task height of the pass)
    const epic_hip::TilePlan tp = tile_plan(c);
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
    const char *e = getenv("EPIC_HIP_TILE_PIPELINE");
    if (e && e[0] == '0') return false;
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

// max |du| of a check iteration that ran as the last step of a tile launch (enqueue_plain_run, check_last): wait for the
// stream, take the maximum over the tiles' words in pinned memory
int read_tile_delta(Harmonic *h, Ctx *c, const char *fn)
{
    if (hipStreamSynchronize(c->stream) != hipSuccess) {
        report(fn, "Failed to synchronize the device after the 'update and check' kernel.");
        return EPIC_ERROR_DEVICE_SYNCHRONIZE;
    }
    float d = 0.0f;
    for (int t = 0; t < c->tile_delta_n; ++t) d = std::max(d, c->h_tile_delta[t]);   // the plan of the launch that wrote them
    h->delta = d;
    return EPIC_SUCCESS;
}

int multi_read_delta(Harmonic *h, Ctx *c, const char *fn);

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

int multi_upload_u(Harmonic *h, Ctx *c, const char *fn);
int multi_upload_locked(Harmonic *h, Ctx *c, const char *fn);

int upload_u(Harmonic *h, Ctx *c, const char *fn)
{
    if (c->multi()) return multi_upload_u(h, c, fn);
    force_all(c);  // new values: no tile may be left out on the strength of the old work lists
    // padding columns hold the obstacle seed; both buffers, so that whichever is read first is complete
    if (c->pitch != c->cols) {
        for (int b = 0; b < 2; b++)
            if (epic_hip::launch_fill(c->buf[b], (size_t)c->rows * c->pitch, -1e6f, c->stream) != hipSuccess) {
                report(fn, "Failed to initialise device-side memory for the potential values.");
                return EPIC_ERROR_KERNEL_EXECUTION;
            }
        if (hipStreamSynchronize(c->stream) != hipSuccess) return EPIC_ERROR_DEVICE_SYNCHRONIZE;
    }
    c->cur = 0;
    if (hipMemcpy2D(c->buf[0], (size_t)c->pitch * sizeof(float), h->u, (size_t)c->cols * sizeof(float),
                    (size_t)c->cols * sizeof(float), (size_t)c->rows, hipMemcpyHostToDevice) != hipSuccess) {
        report(fn, "Failed to copy memory from host to device for the potential values.");
        return EPIC_ERROR_MEMCPY_TO_DEVICE;
    }
    h->d_u = c->buf[0];
    return EPIC_SUCCESS;
}

int upload_locked(Harmonic *h, Ctx *c, const char *fn)
{
    if (c->multi()) return multi_upload_locked(h, c, fn);
    force_all(c);
    if (c->n == 4) return EPIC_SUCCESS;   // never read: no lane masks to derive
    const size_t cells = (size_t)c->rows * c->cols;
    uint32_t *tmp = nullptr;
    if (hipMalloc((void **)&tmp, cells * sizeof(uint32_t)) != hipSuccess) {
        (void)hipGetLastError();
        report(fn, "Failed to allocate device-side staging memory for the locked cells.");
        return EPIC_ERROR_DEVICE_MALLOC;
    }
    int rc = EPIC_SUCCESS;
    if (hipMemcpy(tmp, h->locked, cells * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess) {
        report(fn, "Failed to copy memory from host to device for the locked cells.");
        rc = EPIC_ERROR_MEMCPY_TO_DEVICE;
    } else {
        hipError_t e = c->n == 2
                           ? epic_hip::launch_pack_mask_2d(tmp, c->rows, c->cols, c->pitch, 0, 0, c->maskw, c->stream)
                           : epic_hip::launch_pack_mask_3d(tmp, c->m[0], c->m[1], c->m[2], c->pitch, c->maskw, c->stream);
        if (e == hipSuccess && c->n == 2) e = epic_hip::launch_fuse_masks_2d(c->maskw, c->rows, c->pitch, c->maskf(), c->stream);
        if (e != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) {
            report(fn, "Failed to execute the 'pack mask' kernel.");
            rc = EPIC_ERROR_KERNEL_EXECUTION;
        }
    }
    (void)hipFree(tmp);
    return rc;
}


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
// back; the threads spin for a short while between hand-overs (a relaxation hands over every few tens of microseconds) and
// sleep on a condition variable otherwise.
struct Crew {
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
                    for (int spin = 0; spin < 20000 && generation.load(std::memory_order_acquire) == seen; spin++)
                        __builtin_ia32_pause();
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
        for (int spin = 0; spin < 20000 && pending.load(std::memory_order_acquire) != 0; spin++) __builtin_ia32_pause();
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
    const bool no_peer = getenv("EPIC_HIP_NO_PEER") != nullptr;
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
    const char *te = getenv("EPIC_HIP_THREADS");
    if (!(te && atoi(te) == 0)) {
        c->crew = new Crew();
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
    return epic_hip::launch_sweep_3d(src, dst, sl.maskw, sl.rows, c->m[1], c->pitch, lo, hi, c->math, parity, d, st, act, clo, chi);
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
    const bool fuse_rb = !tracked && c->redblack && c->n == 2 && c->math != 4 && getenv("EPIC_HIP_NO_FUSE") == nullptr &&
                         (long long)c->rows * c->pitch >= (1ll << 22);
    const bool fuse = !tracked && (fuses_tol(c) || fuse_rb);
    const int rpt_track = c->n == 2 ? auto_rows_per_task(c) : 32;
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
                                                                  sl.stream, &act, sl.first(), sl.last() + 1);
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

void force_all(Ctx *c)
{
    c->trk.force = 2;
    c->trk_f.force = std::max(c->trk_f.force, 1);   // (one pass of two iterations, in -> out, rewrites every tile of the other buffer)
    c->last_lists = 0;
    for (auto &sl : c->slabs) sl.trk.force = 2;
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
        if (hipSetDevice(sl.dev) != hipSuccess || hipStreamSynchronize(sl.stream) != hipSuccess || !one(sl.trk)) return false;
    return true;
}

}  // namespace

namespace epic {
extern "C" {

// ---------------------------------------------------------------------------------------------------------
// device-state lifecycle (reference: libepic/src/harmonic/harmonic_model_gpu.cu)
// ---------------------------------------------------------------------------------------------------------

int harmonic_initialize_dimension_size_gpu(Harmonic *harmonic)  // harmonic_model_gpu.cu:34-59
{
    static const char *fn = "harmonic_initialize_dimension_size_gpu";
    if (harmonic == nullptr || harmonic->n == 0 || harmonic->m == nullptr) {
        report(fn, "Invalid input.");
        return EPIC_ERROR_INVALID_DATA;
    }
    Ctx *c = get_ctx(harmonic, true);
    if (c == nullptr) {
        report(fn, "Failed to allocate device-side memory for the dimension size.");
        return EPIC_ERROR_DEVICE_MALLOC;
    }
    if (c->d_m) { (void)hipFree(c->d_m); c->d_m = nullptr; }  // re-initialise without uninitialise (harmonic.py:67-71 then harmonic_gpu.cu:172)
    if (hipMalloc((void **)&c->d_m, harmonic->n * sizeof(unsigned)) != hipSuccess) {
        (void)hipGetLastError();
        report(fn, "Failed to allocate device-side memory for the dimension size.");
        drop_ctx_if_empty(harmonic);
        return EPIC_ERROR_DEVICE_MALLOC;
    }
    if (hipMemcpy(c->d_m, harmonic->m, harmonic->n * sizeof(unsigned), hipMemcpyHostToDevice) != hipSuccess) {
        report(fn, "Failed to copy memory from host to device for the dimension size.");
        (void)hipFree(c->d_m);  // (the reference leaks it here, harmonic_model_gpu.cu:50-55)
        c->d_m = nullptr;
        harmonic->d_m = nullptr;
        drop_ctx_if_empty(harmonic);
        return EPIC_ERROR_MEMCPY_TO_DEVICE;
    }
    harmonic->d_m = c->d_m;
    return EPIC_SUCCESS;
}

int harmonic_uninitialize_dimension_size_gpu(Harmonic *harmonic)  // harmonic_model_gpu.cu:62-75
{
    if (harmonic == nullptr) return EPIC_ERROR_INVALID_DATA;
    int rc = EPIC_SUCCESS;
    Ctx *c = find_ctx(harmonic);
    if (c && c->d_m) {
        if (hipFree(c->d_m) != hipSuccess) {
            report("harmonic_uninitialize_dimension_size_gpu", "Failed to free device-side memory for the dimension size.");
            rc = EPIC_ERROR_DEVICE_FREE;
        }
        c->d_m = nullptr;
    }
    harmonic->d_m = nullptr;
    drop_ctx_if_empty(harmonic);
    return rc;
}

int harmonic_initialize_potential_values_gpu(Harmonic *harmonic)  // harmonic_model_gpu.cu:78-110
{
    static const char *fn = "harmonic_initialize_potential_values_gpu";
    if (harmonic == nullptr || harmonic->n == 0 || harmonic->m == nullptr || harmonic->u == nullptr) {
        report(fn, "Invalid input.");
        return EPIC_ERROR_INVALID_DATA;
    }
    Ctx probe;
    if (!dims_from(harmonic, &probe)) {
        report(fn, "Invalid input (n = 2 and n = 3 need every m[i] >= 3; n = 4 is held but never swept; other n are not supported).");
        return EPIC_ERROR_INVALID_DATA;
    }
    Ctx *c = get_ctx(harmonic, true);
    if (c == nullptr) {
        report(fn, "Failed to allocate device-side memory for the potential values.");
        return EPIC_ERROR_DEVICE_MALLOC;
    }
    if ((c->maskw || (c->multi() && c->slabs[0].maskw)) && !same_dims(harmonic, c)) {
        report(fn, "Invalid input (dimensions differ from the locked cells already on the device).");
        return EPIC_ERROR_INVALID_DATA;
    }
    drop_graphs(c);
    for (float *&b : c->buf) { if (b) (void)hipFree(b); b = nullptr; }
    free_spare(c);
    if (c->multi()) { DeviceGuard g; multi_free_u(c); }
    dims_into_ctx(harmonic, c);
    if (c->plan_failed) {   // (EPIC_HIP_DEVICES: not silently on one device instead -- the caller asked for the node)
        report(fn, "Failed to create the streams, events and staging buffers of the device slabs.");
        harmonic->d_u = nullptr;
        drop_ctx_if_empty(harmonic);
        return EPIC_ERROR_DEVICE_MALLOC;
    }
    if (c->multi()) {  // one pair of buffers per slab, each on its device
        DeviceGuard g;
        for (auto &sl : c->slabs)
            for (int b = 0; b < 2; b++)
                if (hipSetDevice(sl.dev) != hipSuccess ||
                    hipMalloc((void **)&sl.buf[b], (size_t)sl.rows * unit_floats(c) * sizeof(float)) != hipSuccess) {
                    (void)hipGetLastError();
                    report(fn, "Failed to allocate device-side memory for the potential values.");
                    multi_free_u(c);
                    harmonic->d_u = nullptr;
                    drop_ctx_if_empty(harmonic);
                    return EPIC_ERROR_DEVICE_MALLOC;
                }
        return upload_u(harmonic, c, fn);
    }
    for (int b = 0; b < 2; b++) {
        if (hipMalloc((void **)&c->buf[b], c->u_bytes()) != hipSuccess) {
            (void)hipGetLastError();
            report(fn, "Failed to allocate device-side memory for the potential values.");
            for (float *&bb : c->buf) { if (bb) (void)hipFree(bb); bb = nullptr; }
            harmonic->d_u = nullptr;
            drop_ctx_if_empty(harmonic);
            return EPIC_ERROR_DEVICE_MALLOC;
        }
    }
    return upload_u(harmonic, c, fn);
}

int harmonic_uninitialize_potential_values_gpu(Harmonic *harmonic)  // harmonic_model_gpu.cu:113-126
{
    if (harmonic == nullptr) return EPIC_ERROR_INVALID_DATA;
    int rc = EPIC_SUCCESS;
    Ctx *c = find_ctx(harmonic);
    if (c) {
        if (c->stream) (void)hipStreamSynchronize(c->stream);
        drop_graphs(c);
        if (c->multi()) { DeviceGuard g; multi_free_u(c); }
        for (float *&b : c->buf) {
            if (b && hipFree(b) != hipSuccess) {
                report("harmonic_uninitialize_potential_values_gpu", "Failed to free device-side memory for the potential values.");
                rc = EPIC_ERROR_DEVICE_FREE;
            }
            b = nullptr;
        }
        free_spare(c);
    }
    harmonic->d_u = nullptr;
    drop_ctx_if_empty(harmonic);
    return rc;
}

int harmonic_initialize_locked_gpu(Harmonic *harmonic)  // harmonic_model_gpu.cu:129-161
{
    static const char *fn = "harmonic_initialize_locked_gpu";
    if (harmonic == nullptr || harmonic->n == 0 || harmonic->m == nullptr || harmonic->locked == nullptr) {
        report(fn, "Invalid input.");
        return EPIC_ERROR_INVALID_DATA;
    }
    Ctx probe;
    if (!dims_from(harmonic, &probe)) {
        report(fn, "Invalid input (n = 2 and n = 3 need every m[i] >= 3; n = 4 is held but never swept; other n are not supported).");
        return EPIC_ERROR_INVALID_DATA;
    }
    Ctx *c = get_ctx(harmonic, true);
    if (c == nullptr) {
        report(fn, "Failed to allocate device-side memory for the locked cells.");
        return EPIC_ERROR_DEVICE_MALLOC;
    }
    if ((c->buf[0] || (c->multi() && c->slabs[0].buf[0])) && !same_dims(harmonic, c)) {
        report(fn, "Invalid input (dimensions differ from the potential values already on the device).");
        return EPIC_ERROR_INVALID_DATA;
    }
    drop_graphs(c);
    if (c->maskw) { (void)hipFree(c->maskw); c->maskw = nullptr; }
    if (c->multi()) { DeviceGuard g; multi_free_mask(c); }
    dims_into_ctx(harmonic, c);
    if (c->plan_failed) {
        report(fn, "Failed to create the streams, events and staging buffers of the device slabs.");
        harmonic->d_locked = nullptr;
        drop_ctx_if_empty(harmonic);
        return EPIC_ERROR_DEVICE_MALLOC;
    }
    if (c->multi()) {
        {
            DeviceGuard g;
            for (auto &sl : c->slabs)
                if (hipSetDevice(sl.dev) != hipSuccess ||
                    hipMalloc((void **)&sl.maskw, sizeof(uint32_t) * slab_mask_words(c, sl)) != hipSuccess) {
                    (void)hipGetLastError();
                    report(fn, "Failed to allocate device-side memory for the locked cells.");
                    multi_free_mask(c);
                    harmonic->d_locked = nullptr;
                    drop_ctx_if_empty(harmonic);
                    return EPIC_ERROR_DEVICE_MALLOC;
                }
        }
        int rc = upload_locked(harmonic, c, fn);
        if (rc == EPIC_SUCCESS) harmonic->d_locked = c->slabs[0].maskw;
        return rc;
    }
    if (hipMalloc((void **)&c->maskw, c->mask_bytes()) != hipSuccess) {
        (void)hipGetLastError();
        report(fn, "Failed to allocate device-side memory for the locked cells.");
        harmonic->d_locked = nullptr;
        drop_ctx_if_empty(harmonic);
        return EPIC_ERROR_DEVICE_MALLOC;
    }
    int rc = upload_locked(harmonic, c, fn);
    if (rc == EPIC_SUCCESS) harmonic->d_locked = c->maskw;
    return rc;
}

int harmonic_uninitialize_locked_gpu(Harmonic *harmonic)  // harmonic_model_gpu.cu:164-177
{
    if (harmonic == nullptr) return EPIC_ERROR_INVALID_DATA;
    int rc = EPIC_SUCCESS;
    Ctx *c = find_ctx(harmonic);
    if (c && c->multi()) { DeviceGuard g; multi_free_mask(c); }
    if (c && c->maskw) {
        if (c->stream) (void)hipStreamSynchronize(c->stream);
        drop_graphs(c);
        if (hipFree(c->maskw) != hipSuccess) {
            report("harmonic_uninitialize_locked_gpu", "Failed to free device-side memory for the locked cells.");
            rc = EPIC_ERROR_DEVICE_FREE;
        }
        c->maskw = nullptr;
    }
    harmonic->d_locked = nullptr;
    drop_ctx_if_empty(harmonic);
    return rc;
}

int harmonic_update_model_gpu(Harmonic *harmonic)  // harmonic_model_gpu.cu:172-204
{
    static const char *fn = "harmonic_update_model_gpu";
    if (harmonic == nullptr || harmonic->n == 0 || harmonic->m == nullptr || harmonic->u == nullptr ||
        harmonic->d_u == nullptr || harmonic->locked == nullptr || harmonic->d_locked == nullptr) {
        report(fn, "Invalid data.");
        return EPIC_ERROR_INVALID_DATA;
    }
    Ctx *c = find_ctx(harmonic);
    if (!ready(harmonic, c) || !same_dims(harmonic, c)) {
        report(fn, "Invalid data.");
        return EPIC_ERROR_INVALID_DATA;
    }
    if (c->multi()) { DeviceGuard g; multi_sync(c); }
    if (hipStreamSynchronize(c->stream) != hipSuccess) return EPIC_ERROR_DEVICE_SYNCHRONIZE;
    int rc = upload_u(harmonic, c, fn);
    if (rc != EPIC_SUCCESS) return rc;
    return upload_locked(harmonic, c, fn);
}

// ---------------------------------------------------------------------------------------------------------
// solver drivers (reference: libepic/src/harmonic/harmonic_gpu.cu)
// ---------------------------------------------------------------------------------------------------------

int harmonic_initialize_gpu(Harmonic *harmonic, unsigned int numThreads)  // harmonic_gpu.cu:204-223
{
    static const char *fn = "harmonic_initialize_gpu";
    (void)numThreads;
    if (harmonic == nullptr || harmonic->n == 0 || harmonic->m == nullptr || harmonic->d_delta != nullptr) {
        report(fn, "Invalid input.");
        return EPIC_ERROR_INVALID_DATA;
    }
    Ctx *c = get_ctx(harmonic, true);
    if (c == nullptr) {
        report(fn, "Failed to allocate device-side memory for delta.");
        return EPIC_ERROR_DEVICE_MALLOC;
    }
    if (c->d_delta) { (void)hipFree(c->d_delta); c->d_delta = nullptr; }
    if (c->multi()) {  // one delta word per slab; the host takes the maximum
        DeviceGuard g;
        multi_free_delta(c);
        for (auto &sl : c->slabs)
            if (hipSetDevice(sl.dev) != hipSuccess || hipMalloc((void **)&sl.d_delta, 64) != hipSuccess) {
                (void)hipGetLastError();
                report(fn, "Failed to allocate device-side memory for delta.");
                multi_free_delta(c);
                drop_ctx_if_empty(harmonic);
                return EPIC_ERROR_DEVICE_MALLOC;
            }
        harmonic->d_delta = reinterpret_cast<float *>(c->slabs[0].d_delta);
        return EPIC_SUCCESS;
    }
    if (hipMalloc((void **)&c->d_delta, 64) != hipSuccess) {
        (void)hipGetLastError();
        report(fn, "Failed to allocate device-side memory for delta.");
        drop_ctx_if_empty(harmonic);
        return EPIC_ERROR_DEVICE_MALLOC;
    }
    harmonic->d_delta = reinterpret_cast<float *>(c->d_delta);
    return EPIC_SUCCESS;
}

int harmonic_uninitialize_gpu(Harmonic *harmonic)  // harmonic_gpu.cu:307-324
{
    if (harmonic == nullptr) return EPIC_ERROR_INVALID_DATA;
    int rc = EPIC_SUCCESS;
    Ctx *c = find_ctx(harmonic);
    if (c && c->multi()) { DeviceGuard g; multi_free_delta(c); }
    if (c && c->d_delta) {
        if (c->stream) (void)hipStreamSynchronize(c->stream);
        if (hipFree(c->d_delta) != hipSuccess) {
            report("harmonic_uninitialize_gpu", "Failed to free device-side memory for delta.");
            rc = EPIC_ERROR_DEVICE_FREE;
        }
        c->d_delta = nullptr;
    }
    harmonic->d_delta = nullptr;
    drop_ctx_if_empty(harmonic);
    return rc;
}

int harmonic_update_gpu(Harmonic *harmonic, unsigned int numThreads)  // harmonic_gpu.cu:327-350
{
    static const char *fn = "harmonic_update_gpu";
    (void)numThreads;
    Ctx *c = harmonic ? find_ctx(harmonic) : nullptr;
    if (!harmonic || !ready(harmonic, c)) {
        report(fn, "Invalid data.");
        return EPIC_ERROR_INVALID_DATA;
    }
    if (enqueue_sweep(c, false, harmonic->currentIteration) != hipSuccess) {
        report(fn, "Failed to execute the 'Jacobi update' kernel.");
        return EPIC_ERROR_KERNEL_EXECUTION;
    }
    harmonic->d_u = current_u(c);
    harmonic->currentIteration++;
    return EPIC_SUCCESS;
}

int harmonic_update_and_check_gpu(Harmonic *harmonic, unsigned int numThreads)  // harmonic_gpu.cu:353-415
{
    static const char *fn = "harmonic_update_and_check_gpu";
    (void)numThreads;
    Ctx *c = harmonic ? find_ctx(harmonic) : nullptr;
    if (!harmonic || !ready(harmonic, c) || !has_delta(c) || harmonic->d_delta == nullptr) {
        report(fn, "Invalid data.");
        return EPIC_ERROR_INVALID_DATA;
    }
    if (enqueue_sweep(c, true, harmonic->currentIteration) != hipSuccess) {
        report(fn, "Failed to execute the 'Jacobi update and check' kernel.");
        return EPIC_ERROR_KERNEL_EXECUTION;
    }
    harmonic->d_u = current_u(c);
    int rc = read_delta(harmonic, c, fn);
    if (rc != EPIC_SUCCESS) return rc;
    harmonic->currentIteration++;
    return harmonic->delta < harmonic->epsilon ? EPIC_SUCCESS_AND_CONVERGED : EPIC_SUCCESS;
}

int harmonic_get_potential_values_gpu(Harmonic *harmonic)  // harmonic_gpu.cu:418-434
{
    static const char *fn = "harmonic_get_potential_values_gpu";
    Ctx *c = harmonic ? find_ctx(harmonic) : nullptr;
    if (!harmonic || harmonic->u == nullptr || !c || !(c->buf[0] || (c->multi() && c->slabs[0].buf[0])) || harmonic->d_u == nullptr) {
        report(fn, "Invalid data.");
        return EPIC_ERROR_INVALID_DATA;
    }
    if (c->multi()) return multi_get_values(harmonic, c, fn);
    if (hipStreamSynchronize(c->stream) != hipSuccess) {
        report(fn, "Failed to synchronize the device before reading the potential values.");
        return EPIC_ERROR_DEVICE_SYNCHRONIZE;
    }
    if (hipMemcpy2D(harmonic->u, (size_t)c->cols * sizeof(float), c->buf[c->cur], (size_t)c->pitch * sizeof(float),
                    (size_t)c->cols * sizeof(float), (size_t)c->rows, hipMemcpyDeviceToHost) != hipSuccess) {
        report(fn, "Failed to copy memory from device to host for the potential values.");
        return EPIC_ERROR_MEMCPY_TO_HOST;
    }
    return EPIC_SUCCESS;
}

// harmonic_execute_gpu: should the plain batch that follows a check run without the work lists?  (see the call site)
static bool bypass_lists_for_batch(Ctx *c, bool pairs = false)
{
    // (a forced iteration runs every tile but still lists the tiles it changed)
    if (!c->track || c->track_mode != 2 || c->n != 2) return false;
    if (pairs && c->last_lists != 2) return false;   // no lists of the fused tiling yet: the next pass runs every tile and makes them
    const char *e = getenv("EPIC_HIP_TRACK_SWITCH");   // share of due tiles above which lists are bypassed (tests: 0 / 2)
    // The break-even share is where a list-driven iteration costs what an iteration of the untracked path costs -- and that path
    // differs: fused pairs for everything but precise Jacobi, and a red-black pair recomputes each cell once where two list-driven
    // half-sweeps move the whole field twice.  Measured on whole 8192^2 relaxations, same box (tools/exp_track_switch.sh,
    // profiles/r03_experiments.txt item 10), seconds at 0.4 / 0.5 / 0.6 / 0.7 / 0.8 / 0.9:
    //   tol red-black      2.02 / 2.02 / 2.11 / 2.14 / 2.24 / 2.45        precise red-black   2.57 / 2.51 / 2.49 / 2.49 / 2.57 / 2.75
    //   tol Jacobi           -  / 2.53 / 2.49 / 2.48 / 2.51 / 2.61        precise Jacobi        -  / 4.08 / 3.94 / 3.82 / 3.74 / 3.73
    const bool tol = c->math == 4;
    // (tracked PAIRS, round 4: a list-driven fused pass costs what the untracked one costs plus the lists and the shorter tasks'
    //  extra rows -- the lists only lose where nearly every tile is due)
    const double rule = pairs ? 0.85 : c->redblack ? (tol ? 0.45 : 0.6) : (tol ? 0.7 : 0.85);
    const double limit = e ? atof(e) : rule;
    // the counter sets the next launches would consume were filled by the check iteration that has just been read back
    unsigned long long due = 0, tiles = 0;
    if (!due_tiles(c, &due, &tiles, false) || tiles == 0) return false;
    return (double)due > limit * (double)tiles;
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
    const char *fin_env = getenv("EPIC_HIP_TOL_FINISH");
    const bool finish_off = fin_env && fin_env[0] == '0' && harmonic->epsilon <= kTolFinishOptionalBelow;
    if (c->math == 4 && fin_env && fin_env[0] == '0' && !finish_off) {
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
    if (const char *e = getenv("EPIC_HIP_TOL_FINISH_FACTOR")) {   // study knob (tools/finish_study_gpu.py)
        const float v = (float)atof(e);
        if (v >= 1.0f && v <= 1e9f) finish_factor = v;
    }
    const float finish_below = finish_factor * harmonic->epsilon;
    const unsigned stagger = harmonic->numIterationsToStaggerCheck;
    result = EPIC_SUCCESS;
    // what follows every check iteration (result and harmonic->delta are the check's)
    auto after_check = [&] {
        if (finish_wanted && !finish.on && harmonic->delta < finish_below) {
            // A delta that has fallen by less than 0.3 % over the last check interval is a plateau on which the stop is decided by
            // single ulps of single cells (maps/trivial.png: 0.15 % per 100 iterations): there an arithmetic that is not the
            // reference's bit for bit cannot promise the reference's stop, whatever the hand-over factor.
            if (handover.last_check > 0.0f && harmonic->delta > 0.997f * handover.last_check) {
                fprintf(stderr, "Warning[epic_hip]: tol math on a slowly converging map (delta %.3e after %.3e one check earlier): the stop is decided by "
                                "single ulps here and the tol arithmetic cannot promise the 1e-5 parity bar; the default (precise) math reproduces the reference.\n",
                        (double)harmonic->delta, (double)handover.last_check);
            }
            finish.on = true;
            c->finish_from = harmonic->currentIteration;
            c->math = 0;          // precise
            c->redblack = true;   // the reference's half-sweeps, colour by currentIteration
            force_all(c);
            result = EPIC_SUCCESS;   // only a check of the finishing phase may end the loop
        } else if (!c->redblack && result == EPIC_SUCCESS && harmonic->delta < 1.0f && handover.last_check >= 0.0f &&
                   harmonic->delta >= handover.last_check) {
            c->redblack = true;
            force_all(c);
            handover.done = true;
        }
        handover.last_check = harmonic->delta;
    };
    while (result != EPIC_SUCCESS_AND_CONVERGED || harmonic->currentIteration < mMax) {
        if (harmonic->currentIteration % stagger == 0) {
            result = harmonic_update_and_check_gpu(harmonic, numThreads);
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
            const unsigned total = batch + 1;
            unsigned done = 0;
            hipError_t pe = hipSuccess;
            if (total & 1u) {
                pe = enqueue_sweep(c, false, harmonic->currentIteration);
                done = 1;
            }
            if (pe == hipSuccess) {
                rb_pairs_choose_rows(c);
                const bool bypass = bypass_lists_for_batch(c, true);
                if (bypass) tune_fused_rows(c, c->math != 4 ? 2 : c->redblack ? 1 : 0, harmonic->currentIteration);   // the untracked pass's task height, measured once per grid
                pe = enqueue_rb_pairs_tracked(c, (total - done) / 2, harmonic->currentIteration + done, true, bypass);
            }
            if (pe != hipSuccess) {
                report(fn, "Failed to perform the Jacobi update step.");
                return EPIC_ERROR_KERNEL_EXECUTION;
            }
            harmonic->d_u = current_u(c);
            harmonic->currentIteration += batch;
            result = read_delta(harmonic, c, fn);
            if (result != EPIC_SUCCESS) return result;
            harmonic->currentIteration++;
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
            auto enqueue_block = [&](float *in, unsigned first, Blk *out) -> hipError_t {
                const unsigned total = (stagger - first % stagger) + 1;   // the plain iterations and the check
                float *o1 = nullptr, *o2 = nullptr;
                for (float *b : bufs)
                    if (b != in) (o1 ? o2 : o1) = b;
                float *src = in;
                for (unsigned i = 0; i < total;) {
                    const unsigned k = std::min<unsigned>(total - i, (unsigned)tp.halo);
                    float *dst = src == o1 ? o2 : o1;
                    hipError_t e = epic_hip::launch_tile_2d(src, dst, c->maskw, c->rows, c->pitch, tp, (int)k, c->math,
                                                            c->redblack ? (int)((first + i) & 1u) : -1, nullptr, c->stream,
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
            if (enqueue_plain_batch(c, batch, harmonic->currentIteration, true) != hipSuccess) {
                report(fn, "Failed to perform the Jacobi update step.");
                return EPIC_ERROR_KERNEL_EXECUTION;
            }
            harmonic->d_u = current_u(c);
            harmonic->currentIteration += batch;
            result = read_tile_delta(harmonic, c, fn);
            if (result != EPIC_SUCCESS) return result;
            harmonic->currentIteration++;
            result = harmonic->delta < harmonic->epsilon ? EPIC_SUCCESS_AND_CONVERGED : EPIC_SUCCESS;
            after_check();
        } else {
            // every plain iteration returns SUCCESS (which clears a previous CONVERGED), so the ones up to the next
            // check need no host decision in between: enqueue them as one batch
            const unsigned batch = stagger - harmonic->currentIteration % stagger;
            // Work lists pay while a good part of the tiles is at rest; in the phase in which nearly every tile is due (the
            // middle third of a relaxation from scratch) a list-driven sweep costs more than the plain one -- 12 us + 115 us
            // x share against 97 us per iteration of the fused pass at 8192^2, tol math.  The check that has just completed
            // counted the tiles due next: above the switch share this batch runs without lists, and the next two
            // iterations rebuild them (force = 2, as after an upload).  2-D, one device, tracking in its automatic mode;
            // measured on 8192^2 with every arithmetic and scheme (seconds to eps = 1e-6, never / 0.8): tol Jacobi 2.91 / 2.75,
            // tol red-black 2.62 / 2.52, precise Jacobi 3.78 / 3.69, precise red-black 3.01 / 2.62.  Fields and iteration
            // counts do not depend on it.
            const bool bypass = bypass_lists_for_batch(c);
            if (bypass) c->track = false;
            const hipError_t be = enqueue_plain_batch(c, batch, harmonic->currentIteration);
            if (bypass) { c->track = true; force_all(c); }
            if (be != hipSuccess) {
                report(fn, "Failed to perform the Jacobi update step.");
                return EPIC_ERROR_KERNEL_EXECUTION;
            }
            harmonic->d_u = current_u(c);
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
    const unsigned plain = sweeps - ((check_last && sweeps > 0) ? 1u : 0u);
    const bool tiled_check = plain < sweeps && plain > 0 && tile_checks(c, tile_plan(c));   // the check rides the last tile launch
    if (enqueue_plain_batch(c, plain, harmonic->currentIteration, tiled_check) != hipSuccess ||
        (plain < sweeps && !tiled_check && enqueue_sweep(c, true, harmonic->currentIteration + plain) != hipSuccess)) {
        report(fn, "Failed to execute the 'Jacobi update' kernel.");
        return EPIC_ERROR_KERNEL_EXECUTION;
    }
    harmonic->currentIteration += plain;
    harmonic->d_u = current_u(c);
    if (check_last && sweeps > 0) {
        int rc = tiled_check ? read_tile_delta(harmonic, c, fn) : read_delta(harmonic, c, fn);
        if (rc != EPIC_SUCCESS) return rc;
        harmonic->currentIteration++;
        return harmonic->delta < harmonic->epsilon ? EPIC_SUCCESS_AND_CONVERGED : EPIC_SUCCESS;
    }
    return EPIC_SUCCESS;
}

int epic_hip_timed_sweeps_gpu(Harmonic *harmonic, unsigned int sweeps, unsigned int check_every, float *elapsed_ms)
{
    static const char *fn = "epic_hip_timed_sweeps_gpu";
    Ctx *c = harmonic ? find_ctx(harmonic) : nullptr;
    if (!harmonic || !elapsed_ms || !ready(harmonic, c) || (check_every && !has_delta(c))) {
        report(fn, "Invalid data.");
        return EPIC_ERROR_INVALID_DATA;
    }
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
                if (enqueue_sweep(c, true, it) != hipSuccess) rc = EPIC_ERROR_KERNEL_EXECUTION;
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
            if (enqueue_sweep(c, true, it) != hipSuccess) rc = EPIC_ERROR_KERNEL_EXECUTION;
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
    const bool no_fuse = getenv("EPIC_HIP_NO_FUSE") != nullptr;   // (read per call: the tests switch it)
    const bool rb_fused = c->redblack && c->n == 2 && !no_fuse && !c->track && c->math != 4 &&
                          (long long)c->rows * c->pitch >= rb_fuse_min_cells();
    return rb_fused ? 2 : 1;
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
    c->rows_per_task = (int)rows_per_task;
    c->tuned_rows[0] = c->tuned_rows[1] = c->tuned_rows[2] = 0;   // back to "not measured": a height of 0 here means automatic again
    force_all(c);
    return EPIC_SUCCESS;
}

int epic_hip_set_math_mode(Harmonic *harmonic, int mode)
{
    Ctx *c = harmonic ? find_ctx(harmonic) : nullptr;
    if (!c || mode < 0 || mode > 4 || mode == 3) return EPIC_ERROR_INVALID_DATA;  // 2 = traffic-only diagnostic (2-D), 4 = tol; 3 was round 1's df32
    if (c->math != mode) c->tuned_rows[0] = c->tuned_rows[1] = c->tuned_rows[2] = 0;   // (kind 2 serves precise and fast: measured per arithmetic)
    c->math = mode;
    force_all(c);
    return EPIC_SUCCESS;
}

int epic_hip_set_scheme(Harmonic *harmonic, int scheme)
{
    Ctx *c = harmonic ? find_ctx(harmonic) : nullptr;
    if (!c || (scheme != 0 && scheme != 1)) return EPIC_ERROR_INVALID_DATA;
    c->redblack = scheme == 1;
    force_all(c);
    return EPIC_SUCCESS;
}

int epic_hip_set_activity_tracking(Harmonic *harmonic, int on)
{
    Ctx *c = harmonic ? find_ctx(harmonic) : nullptr;
    if (!c || on < 0 || on > 2) return EPIC_ERROR_INVALID_DATA;
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
    *active_tiles = *tiles = 0;
    if (due_tiles) *due_tiles = 0;
    if (!c->track) return EPIC_SUCCESS;
    // the counter sets the next launches will consume were filled by the latest ones: the tiles they woke (summed over the
    // slabs in multi-device mode; a forced iteration runs every tile)
    if (!c->multi() && hipStreamSynchronize(c->stream) != hipSuccess) return EPIC_ERROR_DEVICE_SYNCHRONIZE;
    unsigned long long due = 0, all = 0;
    if (!::due_tiles(c, &due, &all, true)) return EPIC_SUCCESS;
    *active_tiles = due;
    if (due_tiles) *due_tiles = due;
    *tiles = all;
    return EPIC_SUCCESS;
}

int epic_hip_work_done(Harmonic *harmonic, double *grid_iterations, int reset)
{
    Ctx *c = harmonic ? find_ctx(harmonic) : nullptr;
    if (!c || !grid_iterations) return EPIC_ERROR_INVALID_DATA;
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
    std::string out = "{";
    auto add = [&](const char *fmt, auto... a) { char t[256]; snprintf(t, sizeof t, fmt, a...); out += t; };
    add("\"slabs\": %d, \"halo\": %d, \"issuing_threads\": %s, \"seams\": [", (int)c->slabs.size(), c->halo,
        c->crew && !c->crew->threads.empty() ? "true" : "false");
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
    const bool pairs = math_mode == 4 && getenv("EPIC_HIP_NO_FUSE") == nullptr;
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

}  // extern "C"
