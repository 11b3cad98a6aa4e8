// driver_plan.hip -- which kernel family runs a batch of iterations, and with what tiling: the planner (task heights, fused passes,
// LDS tiles for small grids, tracked pairs, when work lists are bypassed) and the tuner that measures the fused passes' task height.
// Every decision is a function of the context (dimensions, mode) and of its Config (driver.h): no environment is read here.
#include "driver.h"

namespace epic_drv {

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

// Which kernel family serves a 2-D grid WITHOUT work lists is a matter of its size, and the sizes at which the families cross differ
// by arithmetic and scheme.  Measured in round 6 (tools/size_curve.py, profiles/r06_size_curve.txt: us per iteration of a block of
// harmonic_execute_gpu's loop, every family forced on every size):
//                              LDS tiles win up to         single sweeps in between        fused pairs win from
//   precise / fast, red-black        ~3 Mcell            3 - 5.5 Mcell (2048^2: 10.1 us           ~5.5 Mcell
//                                                         against the pass's 12.1)
//   precise / fast, Jacobi          ~0.6 Mcell           everything above (no fused pass)             --
//   tol, red-black                  ~2 Mcell                        --                             ~2 Mcell   (1900^2: 5.9 us against 10.5)
//   tol, Jacobi                     ~1.5 Mcell                      --                             ~1.5 Mcell (1419 x 1735: 6.4 against 10.3)
// Until round 6 one pair of numbers served all four (tiles <= 3 Mcell, fused >= 4 Mcell), which left 3-4 Mcell on single sweeps for the
// tol arithmetic (1.8 x slower than its fused pass there) and put 1.5-3 Mcell tol grids on tiles that the fused pass beats by 1.4-1.6 x.
// EPIC_HIP_FUSE_MIN_CELLS / EPIC_HIP_TILE_MAX_CELLS override (the tests set 0 / huge values).
long long fuse_from_cells(const Ctx *c)
{
    if (c->cfg.fuse_min_cells >= 0) return c->cfg.fuse_min_cells;
    if (c->math == 4) return c->redblack ? (2ll << 20) : (3ll << 19);
    return 11ll << 19;   // 5.5 Mcell
}
long long tile_up_to_cells(const Ctx *c)
{
    if (c->cfg.tile_max_cells >= 0) return c->cfg.tile_max_cells;
    if (c->math != 4 && !c->redblack) return 5ll << 17;   // 0.625 Mcell
    return 3ll << 20;   // (the tol grids above fuse_from_cells never get here: tile_plan asks fuses_tol first)
}
// Tracked pairs: wherever the work lists are on by themselves (grids above 4 Mcell) -- list-driven single sweeps are the alternative there,
// not untracked ones.
long long tracked_pairs_from_cells(const Ctx *c) { return c->cfg.fuse_min_cells >= 0 ? c->cfg.fuse_min_cells : (1ll << 22); }

// Whether two consecutive plain Jacobi iterations run as one fused pass in the context's current configuration.
// EPIC_HIP_FUSE_MIN_CELLS: grids below it keep the single sweeps (default: fuse_from_cells above; below, a sweep is launch-bound and the
// fused pass's extra rows cost more than the second launch; read per batch, not cached -- the tests switch it).
bool fuses_tol(const Ctx *c)   // either scheme
{
    if (c->cfg.no_fuse) return false;
    return c->n == 2 && !c->track && c->math == 4 && (long long)c->rows * c->pitch >= fuse_from_cells(c);
}
bool fuses_jacobi(const Ctx *c) { return !c->redblack && fuses_tol(c); }
// red-black with the precise / fast math: two plain iterations as one rb_fused2d_kernel pass (no work lists; from fuse_min_cells up)
bool fuses_rb_precise(const Ctx *c)
{
    return c->redblack && c->n == 2 && !c->cfg.no_fuse && !c->track && c->math != 4 && (long long)c->rows * c->pitch >= fuse_from_cells(c);
}
// red-black, tol math: both colours in one pass (rb_tol_fused2d_kernel); one device only
bool fuses_rb_tol(const Ctx *c) { return c->redblack && fuses_tol(c); }
// (on several devices a pass leaves two more ghost units stale, so neither of its two iterations may be one that ends with an
// exchange: multi_run fuses inside the stretches between exchanges only)

// Rows per task of the fused Jacobi pass (kernels.h: jacobi_fused_auto_rows), per device in multi-device mode.
int jacobi_fused_rows_per_task(const Ctx *c)
{
    if (c->cfg.fused_rows > 0) return c->cfg.fused_rows;   // EPIC_HIP_FUSED_ROWS: experiment / test knob
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
// (Round 5 found the largest part of it for the tol passes -- the CU with the most blocks sets the time, ceil(blocks / CUs) of them: the
// steps of 7-10 % sit where one block more than a multiple of the CU count appears -- and the launcher now cuts the rows into as many
// chunks as fit that count (kernels_2d.hip: tighten_chunks), so that a height only chooses the number of blocks per CU: the measurement
// stays, its table is flat within 2 % around the best where it was a saw.)
void tune_fused_rows(Ctx *c, int kind, unsigned iteration)
{
    if (c->tuned_rows[kind] != 0) return;
    // not on the field of the first iterations (all cells at the initial value: the passes run up to 15 % faster on it and rank the
    // heights differently): the rule serves until the front has crossed a good part of the grid
    if (iteration < (unsigned)std::min(c->rows, c->cols) / 2) return;
    c->tuned_rows[kind] = -1;
    if (!c->cfg.tune || c->n != 2 || c->rows_per_task > 0 || c->cfg.fused_rows > 0) return;
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
    const bool say = c->cfg.tune_debug;   // EPIC_HIP_TUNE_DEBUG: the table on stderr
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

// Small 2-D grids (one device, no work lists): the plain iterations between two checks run several per launch on tiles that
// stay in LDS (kernels_tile2d.hip) -- the reference's maps are launch-bound, 2.5-3 us per half-sweep whatever it computes.
// halo == 0: not for this context.  Knobs (read per batch; the tests switch them): EPIC_HIP_TILE=0 off; EPIC_HIP_TILE_HALO = ghost
// rings = iterations per launch, EPIC_HIP_TILE_WIDTH = 64 | 128 (default for both: a cost model of the launch, below);
// EPIC_HIP_TILE_ROWS = owned rows per tile; EPIC_HIP_TILE_MAX_CELLS (default: tile_up_to_cells above).
// Where it pays (tools/tile_probe.py, tools/time_maps.py; profiles/r04_experiments.txt items 1f, 1i): 2.0-2.4 x on the maps up to
// 0.3 Mcell (one narrow tile per CU), 1.1-1.3 x on the 1-2.5 Mcell fixtures (wide tiles).
epic_hip::TilePlan tile_plan(const Ctx *c)
{
    epic_hip::TilePlan none = {0, 0, 0, 0, 0};
    if (c->n != 2 || c->multi() || c->track || c->math == 2) return none;
    if (fuses_tol(c)) return none;   // (a fused pass asked for on a small grid: EPIC_HIP_FUSE_MIN_CELLS, the tests)
    if (!c->cfg.tile) return none;
    if ((long long)c->rows * c->cols > tile_up_to_cells(c)) return none;
    const int want_rows = c->cfg.tile_rows;
    const int only_width = c->cfg.tile_width;   // 64 | 128: one width only (experiments, tests)
    const int only_halo = c->cfg.tile_halo;
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

// Tracked red-black relaxations with the precise / fast math on one device (2-D, from 4 Mcell up -- EPIC_HIP_FUSE_MIN_CELLS): the
// iterations between two checks AND the check run as pairs, each one list-driven fused pass (kernels_2d.hip: rb_fused2d_kernel
// with TRACK; the check is the second iteration of the last pair).  Against in-place half-sweeps with lists: half the launches
// (10-12 us each with next to nothing due, a quarter of all iterations of the 8192^2 relaxation), and a listed tile moves through
// HBM once for two iterations instead of twice.  EPIC_HIP_TRACK_PAIRS=0: the half-sweeps, as before round 4.
bool rb_pairs_tracked(const Ctx *c)   // (the name is round 4's first form: red-black, precise; the tol passes -- both schemes -- followed)
{
    if (!c->track || c->n != 2 || c->multi() || c->math == 2) return false;
    if (!c->redblack && c->math != 4) return false;   // precise Jacobi has no fused pass
    if (c->cfg.no_fuse || !c->cfg.track_pairs) return false;
    return (long long)c->rows * c->pitch >= tracked_pairs_from_cells(c);
}

// The same on the slabs of the multi-device mode (round 6; driver_multi.hip: multi_run_pairs): every slab runs the pass over its local
// rows with lists of its own, the ghost rows are traded after every halo / 2 passes.  A pass makes two ghost rows stale: at least two.
bool rb_pairs_tracked_multi(const Ctx *c)
{
    if (!c->track || c->n != 2 || !c->multi() || c->halo < 2 || c->math == 2) return false;
    if (!c->redblack && c->math != 4) return false;
    if (c->cfg.no_fuse || !c->cfg.track_pairs) return false;
    return (long long)c->rows * c->pitch >= tracked_pairs_from_cells(c);
}

// rows per task of the tracked pass: the unit of skipping, and every task recomputes the first colour of one row above and one
// below its chunk (2 / rows extra arithmetic).  EPIC_HIP_TRACK_PAIR_ROWS overrides.
int rb_pairs_rows_per_task(const Ctx *c)
{
    if (c->cfg.track_pair_rows > 0) return std::max(c->cfg.track_pair_rows, 2);
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
    if (c->cfg.track_pair_rows > 0 || c->rows_per_task > 0) return;
    if (c->pair_rows == 0) c->pair_rows = 16;
    if (c->last_lists != 2) return;
    unsigned long long due = 0, tiles = 0;
    if (!due_tiles(c, &due, &tiles, false) || tiles == 0) return;
    const double share = (double)due / (double)tiles;
    if (c->pair_rows == 16 && share < 0.04) c->pair_rows = 4;
    else if (c->pair_rows == 4 && share > 0.20) c->pair_rows = 16;
}

// harmonic_execute_gpu: should the plain batch that follows a check run without the work lists?  (see the call site)
bool bypass_lists_for_batch(Ctx *c, bool pairs)
{
    // (a forced iteration runs every tile but still lists the tiles it changed)
    if (!c->track || c->track_mode != 2 || c->n != 2) return false;
    if (pairs && c->last_lists != 2) return false;   // no lists of the fused tiling yet: the next pass runs every tile and makes them
    const double given = c->cfg.track_switch;   // EPIC_HIP_TRACK_SWITCH: share of due tiles above which lists are bypassed (tests: 0 / 2)
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
    const double limit = given >= 0.0 ? given : rule;
    // the counter sets the next launches would consume were filled by the check iteration that has just been read back
    unsigned long long due = 0, tiles = 0;
    if (!due_tiles(c, &due, &tiles, false) || tiles == 0) return false;
    return (double)due > limit * (double)tiles;
}

// The kernel family a batch of plain iterations of harmonic_execute_gpu takes in the context's present state -- ONE place that
// says which path a context is on (epic_hip_config_dump; the decisions themselves are the functions above, this only names them).
const char *plain_batch_path(const Ctx *c)
{
    if (c->n == 4) return "none (n = 4: a counting no-op)";
    if (c->multi()) {
        if (rb_pairs_tracked_multi(c)) return c->math == 4 ? "slabs: tracked pairs of list-driven fused tol passes per slab" : "slabs: tracked pairs of list-driven fused red-black passes per slab";
        if (c->track) return "slabs: list-driven single sweeps per slab";
        if (fuses_tol(c)) return c->redblack ? "slabs: fused tol red-black pairs between exchanges" : "slabs: fused tol Jacobi pairs between exchanges";
        if (c->redblack && c->n == 2 && c->math != 4 && !c->cfg.no_fuse && (long long)c->rows * c->pitch >= (1ll << 22)) return "slabs: fused red-black pairs between exchanges";
        return c->n == 3 ? "slabs of planes: single 3-D sweeps" : "slabs: single sweeps";
    }
    if (rb_pairs_tracked(c)) return c->math == 4 ? "tracked pairs of list-driven fused tol passes" : "tracked pairs of list-driven fused red-black passes";
    const epic_hip::TilePlan tp = tile_plan(c);
    if (tp.halo > 0) return tile_checks(c, tp) ? (c->cfg.tile_pipeline ? "LDS tiles, several iterations per launch, check folded in, pipelined blocks" : "LDS tiles, several iterations per launch, check folded in")
                                               : "LDS tiles, several iterations per launch";
    if (fuses_jacobi(c)) return "fused tol Jacobi pairs (jacobi_fused2d_kernel)";
    if (fuses_rb_tol(c)) return "fused tol red-black pairs (rb_tol_fused2d_kernel)";
    if (fuses_rb_precise(c)) return "fused red-black pairs (rb_fused2d_kernel)";
    if (c->track) return c->n == 3 ? "list-driven single 3-D sweeps" : "list-driven single sweeps";
    if (c->n == 3) return c->math == 4 && c->cfg.launch.pair3d ? "single 3-D sweeps, two planes per wave (sweep3d_pair_kernel)" : "single 3-D sweeps (sweep3d_kernel)";
    const bool graphs = (long long)c->rows * c->pitch <= (1ll << 22) && !c->cfg.no_graph && !c->graphs_broken;
    return graphs ? "single sweeps replayed from a captured hipGraph" : "single sweeps";
}

}  // namespace epic_drv

