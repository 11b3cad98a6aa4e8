/*
 * epic_hip.h -- extension entry points of the MI355X-native libepic.so (plain C ABI).
 *
 * The reference ABI (include/epic/epic_abi.h) has no notion of streams, batches of sweeps, timing or
 * more than one GPU.  These additions sit beside it for callers that want them (bench.py, the slab
 * decomposition driver epic_amd/slab.py, a ROS node that wants "k sweeps" in one call); nothing in the
 * reference ABI depends on them.  All pointers named d_* are device pointers on the current HIP device;
 * `stream` is a hipStream_t passed as void* (NULL = the default stream).
 */
#ifndef EPIC_HIP_H
#define EPIC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
namespace epic { struct Harmonic; }
typedef epic::Harmonic EpicHarmonicT;
extern "C" {
#else
struct Harmonic;
typedef struct Harmonic EpicHarmonicT;
#endif

/* "epic-hip <semver> gfx950" */
const char *epic_hip_version(void);
/* Number of HIP devices visible (0 on a GPU-less host; never fails). */
int epic_hip_device_count(void);

/* ---- batches on an initialised Harmonic (device state created by harmonic_initialize_*_gpu) ----------
 * Enqueue `sweeps` iterations of the selected scheme without a host round-trip; currentIteration advances by `sweeps`.
 * If check_last != 0 the last sweep also reduces max|du| and the call synchronises, stores it in
 * harmonic->delta and returns EPIC_SUCCESS_AND_CONVERGED when delta < epsilon (needs harmonic_initialize_gpu).
 * Otherwise it returns after enqueueing (ordering points: any *_and_check, get_potential_values, update_model,
 * set_cells, uninitialize).  Replaces a loop over harmonic_update_gpu (libepic/src/harmonic/harmonic_gpu.cu:327-350),
 * e.g. the navigation node's update(num_steps) (src/epic_navigation_node_harmonic.cpp:165-189). */
int epic_hip_update_n_gpu(EpicHarmonicT *harmonic, unsigned int sweeps, int check_last);

/* Same, bracketed by HIP events on the library's own stream; *elapsed_ms = device time of the batch.
 * Synchronises.  check_every = 0: no checks; k > 0: every k-th sweep (counted by currentIteration % k == 0,
 * the reference's rule, harmonic_gpu.cu:268) is a check sweep. */
int epic_hip_timed_sweeps_gpu(EpicHarmonicT *harmonic, unsigned int sweeps, unsigned int check_every,
                              float *elapsed_ms);

/* How many iterations one kernel launch of a batch of PLAIN (unchecked) iterations advances in the current configuration:
 * 2 where pairs of iterations run as one fused pass over the data -- Jacobi with the tol math (two sweeps, 4 B of HBM
 * traffic per cell-update instead of 8) and red-black with any math (both colours), 2-D grids from the size at which the pass beats
 * the other kernel families (tol Jacobi 1.5 Mcell, tol red-black 2 Mcell, precise red-black 5.5 Mcell; EPIC_HIP_FUSE_MIN_CELLS) with activity tracking off (per device in multi-device mode, between exchanges); results are bit-identical to single iterations, an odd iteration and
 * every check iteration run singly -- 1 otherwise, 0 without device state.  EPIC_HIP_NO_FUSE=1 switches the fusion off. */
int epic_hip_iterations_per_pass(EpicHarmonicT *harmonic);

/* Small 2-D grids (at most 3 Mcell -- less where another family is faster: EPIC_HIP_TILE_MAX_CELLS --, one device, activity tracking off -- the maps the reference's callers relax): the plain
 * iterations between two checks run SEVERAL PER LAUNCH on tiles that stay in LDS with that many ghost rings
 * (epic_amd/csrc/kernels_tile2d.hip); bit-identical to single iterations.  Returns the iterations one such launch advances
 * (8, 10, 12, 14 or 16, chosen per grid by a cost model of the launch; 1 to 27 when EPIC_HIP_TILE_HALO fixes it), 0 where the path
 * is not used.  EPIC_HIP_TILE=0 switches it off; EPIC_HIP_TILE_HALO / _ROWS / _WIDTH / _MAX_CELLS tune it. */
int epic_hip_tile_iterations(EpicHarmonicT *harmonic);

/* Rows per task of that fused pass in the current configuration (0: no fused pass).  On grids of at least 4 Mcell on one
 * device the height is measured on the grid itself the first time a pair of plain iterations is enqueued (a few candidates,
 * three launches each, ~10 ms, once per grid and kind of pass; results do not depend on it): before that the call returns
 * the rule's value.  EPIC_HIP_TUNE=0 keeps the rule; EPIC_HIP_FUSED_ROWS / epic_hip_set_rows_per_task fix the height. */
int epic_hip_fused_rows_per_task(EpicHarmonicT *harmonic);

/* tol math, harmonic_execute_gpu / harmonic_complete_gpu: the loop leaves the tol arithmetic at the first check with
 * delta < 10 epsilon (100 epsilon for epsilon <= 1e-5) and FINISHES with the reference's own iteration (red-black half-sweeps, bit-exact expf / logf).  For
 * relaxations to stagnation (epsilon <= 1e-5) only a check of that phase may end the loop -- the converged field is then the end point of the
 * reference's iteration from a state within a few 1e-5 of it: maps/umass.png 1.4e-6 from harmonic_complete_cpu's field instead of 1.6e-5.  At the
 * callers' epsilons (> 1e-5) the hand-over check keeps its own verdict (round 6: on maps that converge within a few checks the reference stops at that
 * very check), and a first check that reports delta = 0 exactly (a run that has not moved yet) does not hand over.  Measured against the reference on
 * 720 generated cases (2-D and 3-D): tests/tol_campaign.py, DESIGN.md section 2.
 * EPIC_HIP_TOL_FINISH=0 in the environment keeps the tol iteration to the end (honoured for epsilon <= 1e-5 only: above, the
 * finishing phase is what makes the stop the reference's).  EPIC_HIP_TOL_FINISH_FACTOR=f replaces the 10 / 100 (a study knob:
 * tools/finish_study_gpu.py swept it over every map of the reference).
 * WHAT tol CANNOT PROMISE: on maps whose delta crosses epsilon in steps of single ulps (the reference's maps/trivial.png, an
 * almost empty 1024^2 room: delta falls 0.15 % per 100 iterations) the iteration at which the loop stops is decided by single
 * ulps of single cells; no hand-over factor is safe there (4e-3 from the reference for factors 10, 50, 300; inside the bar for
 * 20 and 100 -- by chance), and only the default precise math reproduces the reference.  The library says so once on stderr
 * when a tol relaxation reaches its hand-over on such a plateau (delta down by less than 0.3 % per check over the last 32 checks).
 * Returns the iteration number at which the finishing phase of the latest call began (0: it had none). */
unsigned int epic_hip_finish_iteration(EpicHarmonicT *harmonic);

/* Tuning knob: rows marched by one wave in the 2-D kernel (0 = automatic).  Also EPIC_HIP_ROWS_PER_TASK. */
int epic_hip_set_rows_per_task(EpicHarmonicT *harmonic, unsigned int rows_per_task);

/* Arithmetic of the sweep kernels (also EPIC_HIP_MATH=precise|tol|fast in the environment at initialisation):
 *   0 precise (default)  exp/log bit-identical to the host libm's expf/logf, evaluated in f64 -- the bit-exact parity mode;
 *   4 tol                one exp-class split e^u = q 2^n per CELL (packed f32 polynomial), shared by the cells it is a
 *                        neighbour of, one table-driven log per cell (256 intervals per binade, exact f32 reduction,
 *                        2^-33 accurate); every rounding stage of the reference kept.  2-D and 3-D,
 *                        Jacobi and red-black.  Jacobi stops by the reference's own test.  harmonic_execute_gpu /
 *                        harmonic_complete_gpu finish a tol relaxation with the reference's own iteration
 *                        (epic_hip_finish_iteration, above): converged fields within 1e-5 max(1, |u|) of the reference's
 *                        on every map and grid under test, 1.4e-6 on the ill-conditioned umass.png (the tol iteration
 *                        alone: 1.6e-5 there).  ~1.3x faster per sweep than precise, ~1.75x where pairs of
 *                        iterations run as one fused pass (epic_hip_iterations_per_pass); the benchmarked mode;
 *   1 fast               v_exp_f32 / v_log_f32: biased, ~1e-4 relative drift on ill-conditioned maps; no parity claim;
 *   2 traffic            diagnostic: same loads/stores, trivial arithmetic (2-D only).
 * (3 was round 1's df32 mode, removed: EPIC_ERROR_INVALID_DATA.) */
int epic_hip_set_math_mode(EpicHarmonicT *harmonic, int mode);

/* Iteration scheme: 1 = the reference's red-black Gauss-Seidel, in place (DEFAULT: one iteration = one colour,
 * libepic/src/harmonic/harmonic_cpu.cpp:46-51, :89-102; with the default precise math every half-sweep, the iteration count
 * and the converged field are bit-identical to harmonic_complete_cpu), 0 = Jacobi ping-pong (one iteration recomputes
 * every unlocked cell: what BASELINE.json's metric names and bench.py times; twice the arithmetic for the same answer).
 * Also EPIC_HIP_SCHEME=jacobi|redblack in the environment at initialisation.
 * EPIC_HIP_JACOBI_CHECKS=reference (environment, opt-in): every CHECK iteration of a Jacobi run is the reference's half-sweep of that
 * iteration's colour, in place -- the state after a check, and so the field a relaxation ends with, is the reference's own (with the
 * precise math bit for bit, iteration count included); INTEGRATION.md section 6. */
int epic_hip_set_scheme(EpicHarmonicT *harmonic, int scheme);

/* Activity tracking: 0 off, 1 on, 2 automatic (default: on for grids above 4 Mcell; also
 * EPIC_HIP_TRACK=0|1 in the environment at initialisation).  A tile (2-D: rows_per_task x 256 cells; 3-D: 32 x 256 cells
 * of one plane) is recomputed in an
 * iteration only if the previous iteration changed a value it reads (one of its own cells or the adjacent row / column of
 * an edge neighbour); otherwise the update would reproduce the values already held, so fields, delta and iteration counts
 * are bit-identical with tracking on or off.  Every iteration builds the work lists of its successor, which then runs as
 * a fixed-size launch of persistent waves over exactly those tiles.  Any upload, set_cells or mode change makes the next
 * two iterations run every tile.  bench.py times the kernel with tracking off. */
int epic_hip_set_activity_tracking(EpicHarmonicT *harmonic, int on);

/* Streamline extraction on the device-resident field (the path step that follows the relaxation).
 * The reference copies the whole field to the host for every request (src/epic_navigation_node_harmonic.cpp:621-626) and
 * walks it with harmonic_compute_path_2d_cpu (libepic/src/harmonic/harmonic_path_cpu.cpp:154-221); these walk it in HBM,
 * one lane per start point, and return only the way-points -- bit-identical to the host walk on the same field.
 *   starts  n_paths (x, y) pairs in cell units;  k / rc  n_paths entries (rc: EPIC_SUCCESS, EPIC_ERROR_INVALID_LOCATION,
 *   EPIC_ERROR_INVALID_GRADIENT or EPIC_ERROR_INVALID_PATH per path, k = 0 unless EPIC_SUCCESS);
 *   paths   n_paths rows of 2 * maxLength floats, row i holding k[i] (x, y) pairs.  All pointers are host pointers.
 * The device mask treats the grid border as locked (harmonic.h:35-37 makes that a precondition of the solver). */
int epic_hip_compute_paths_2d_gpu(EpicHarmonicT *harmonic, unsigned int n_paths, const float *starts, float stepSize,
                                  float cdPrecision, unsigned int maxLength, unsigned int *k, int *rc, float *paths);
/* One path, with harmonic_compute_path_2d_cpu's contract: *path must be NULL, receives a new[] array of 2 * *k floats
 * (release with harmonic_free_path_cpu), the return value is the walk's code. */
int epic_hip_compute_path_2d_gpu(EpicHarmonicT *harmonic, float x, float y, float stepSize, float cdPrecision,
                                 unsigned int maxLength, unsigned int *k, float **path);

/* Diagnostic: the number of tiles the next iteration will recompute and the number of tiles (both 0 when tracking is
 * off).  Synchronises the stream and copies the list counters to the host. */
int epic_hip_activity_stats(EpicHarmonicT *harmonic, unsigned long long *active_tiles, unsigned long long *tiles);
/* The same with a separate output for the tiles due in the next iteration (with work lists both outputs report that
 * number: the tiles a launch changed are the tiles it lists, apart from woken neighbours); due_tiles may be NULL. */
int epic_hip_activity_stats2(EpicHarmonicT *harmonic, unsigned long long *active_tiles, unsigned long long *due_tiles,
                             unsigned long long *tiles);

/* Work accounting: how many whole-grid iterations' worth of tiles the kernels have actually RUN since the last reset --
 * an iteration that runs every tile counts 1 (a fused pass over two iterations 2), a list-driven iteration of a tracked
 * run counts (tiles listed) / (tiles).  harmonic_execute_gpu resets it when it starts; reset != 0 resets it after reading.
 * currentIteration minus this number is what activity tracking skipped.  Synchronises the stream. */
int epic_hip_work_done(EpicHarmonicT *harmonic, double *grid_iterations, int reset);

/* Test hook: d_out[i] = (which & 1) ? ln(d_in[i]) : exp(d_in[i]) with the precise device routines; which & 2: in the form the red-black
 * kernels use (the first Horner addend kept in vector registers, cell_update.h: MathTab::consts) -- same operations, same bits. */
int epic_hip_eval_math(const float *d_in, float *d_out, size_t n, int which, void *stream);

/* ---- several GPUs behind the unchanged ABI ------------------------------------------------------------------------
 * EPIC_HIP_DEVICES=0,1,2,3 in the environment when a Harmonic's device state is created makes every harmonic_*_gpu entry
 * point work on ONE grid cut along its slowest axis into one slab per listed device (rows in 2-D, planes in 3-D), in this process: harmonic_complete_gpu(&h, 1024) --
 * the ROS plugin's only call, /root/reference/src/epic_nav_core_plugin.cpp:256 -- then uses the whole node.  The reference
 * has nothing like it (libepic/src/harmonic/harmonic_gpu.cu:168-201 drives one device).  A device may be listed more than
 * once ("0,0,0,0": four slabs on one GPU).  EPIC_HIP_HALO=G (default by slab height: 8 from 4096 rows per device up, 16 from
 * 2048, 32 below): ghost units per interior side, traded every G
 * iterations with hipMemcpyPeerAsync -- or, where peer access between two listed devices cannot be enabled (reported once on
 * stderr; EPIC_HIP_NO_PEER=1 forces it), through pinned host memory; results are bit-identical to the single-device path for
 * every list and every G.  One host thread per slab issues its launches (EPIC_HIP_THREADS=0: the calling thread does; between hand-overs a
 * thread spins for at most EPIC_HIP_SPIN_US microseconds, default 20, then sleeps: an idle Harmonic costs its host process nothing);
 * activity tracking works per slab.  Grids with fewer than 4 units per listed device and an unusable list fall back to one
 * device; path requests walk on the host in this mode.
 * epic_hip_device_layout: which device holds which units (rows of a 2-D grid, planes of a 3-D one) -- returns the number of slabs (1 in single-device mode) and
 * fills at most `cap` entries of each non-NULL array (owned rows [row_begin, row_end); ghost rows per interior side). */
int epic_hip_device_layout(EpicHarmonicT *harmonic, int cap, int *devices, unsigned int *row_begin, unsigned int *row_end,
                           unsigned int *ghost_rows);

/* Multi-device mode reporting on itself (the first run on real hardware cannot be rehearsed): one JSON object in buf -- per seam
 * the two devices, hipDeviceCanAccessPeer in both directions, the transport chosen (peer / staged / same-device), link type and
 * hops where the runtime reports them; then ONE exchange iteration under timing events: per slab the interior sweep (compute
 * stream) and the boundary bands + incoming halo copies (second stream), in microseconds from the earlier of the two starts,
 * the time both were running (overlap_us) and whether the copies ended before the interior did (copies_hidden).  Runs the iterations up to and including the next exchange (currentIteration advances).  Returns the
 * number of bytes written, 0 when the Harmonic is not in multi-device mode or buf is too small. */
int epic_hip_multi_report(EpicHarmonicT *harmonic, char *buf, size_t cap);

/* ---- configuration ---------------------------------------------------------------------------------------------------
 * The library reads its environment (the EPIC_HIP_* knobs of INTEGRATION.md section 6) ONCE per Harmonic, when the library-side
 * context of that Harmonic is created (the first harmonic_initialize_*_gpu), into one struct (epic_amd/csrc/driver_config.h);
 * nothing is read again behind the caller's back.  Two classes: PRODUCT knobs (EPIC_HIP_MATH, _SCHEME, _TRACK, _DEVICES with _HALO / _NO_PEER /
 * _THREADS / _SPIN_US, _TOL_FINISH, _DEFER) are always honoured; STUDY knobs (thresholds, task heights, tile plans, launch flags, the tuner: they
 * select a code path, never a result) are read only under EPIC_HIP_STUDY=1 -- one that is set without it is ignored and named once on stderr.
 * epic_hip_config_dump: one JSON object in buf -- "config": every knob as read then; "state": dimensions and the modes in force now
 * (epic_hip_set_* change them); "path": the kernel family a batch of plain iterations takes (LDS tiles / fused pairs / tracked
 * pairs / list-driven sweeps / single sweeps, replayed from a hipGraph or not), the tile plan, task heights, and in multi-device
 * mode the halo depth and the transport of every seam.  Returns the bytes written; 0 without a context or when buf is too small.
 * epic_hip_config_reload: re-reads the environment for this Harmonic's context (a knob that changed is applied; modes set through
 * epic_hip_set_* stay unless their variable changed; EPIC_HIP_DEVICES / EPIC_HIP_HALO are taken over when neither a field nor a
 * mask is resident, i.e. they shape the NEXT initialisation -- "state" of config_dump shows the slab layout in force) -- for a caller
 * that changes a variable on a live context (tests, bench.py).  It also refreshes the process-wide knobs: EPIC_HIP_FLAGS and
 * EPIC_HIP_LIST_WAVES are read by the 2-D launchers from those (speed only, never results), for every context of the process.
 * NULL: the process-wide knobs only (what the raw operators below go by). */
int epic_hip_config_dump(EpicHarmonicT *harmonic, char *buf, size_t cap);
int epic_hip_config_reload(EpicHarmonicT *harmonic);

/* Geometry of the device-resident state: pitch in floats, bytes of one u buffer, bytes of the packed mask. */
int epic_hip_get_layout(EpicHarmonicT *harmonic, unsigned int *pitch, size_t *u_bytes, size_t *mask_bytes);

/* ---- raw operators on caller-owned device memory (slab decomposition / multi-process drivers) ----------
 * 2-D grid of `rows` x `pitch` floats (pitch % 256 == 0, rows include any ghost rows); the mask is the private
 * bit layout produced by epic_hip_pack_mask_2d (epic_hip_mask_words_2d(rows, pitch) uint32 words, 1 bit per cell).
 * epic_hip_sweep_2d sweeps rows [row_begin, row_end) from d_in to d_out; if d_delta_bits != NULL the max |du|
 * of those rows is atomically max-ed into it as float bits (zero it first).  Asynchronous on `stream`. */
size_t epic_hip_mask_words_2d(unsigned int rows, unsigned int pitch);
unsigned int epic_hip_pitch_for_cols(unsigned int cols);
int epic_hip_pack_mask_2d(const uint32_t *d_locked, unsigned int rows, unsigned int cols, unsigned int pitch,
                          int ghost_top, int ghost_bottom, uint32_t *d_maskw, void *stream);
int epic_hip_sweep_2d(const float *d_in, float *d_out, const uint32_t *d_maskw, unsigned int rows,
                      unsigned int pitch, unsigned int row_begin, unsigned int row_end, unsigned int rows_per_task,
                      int math_mode, uint32_t *d_delta_bits, void *stream);
/* TWO Jacobi sweeps of the whole local grid in one pass (tol math only: math_mode 4): d_out receives what two calls of
 * epic_hip_sweep_2d over rows [0, rows) -- d_in -> tmp -> d_out -- would leave there, bit for bit, with the field moved
 * through HBM once.  No delta (check iterations run singly).  In a slab, two more ghost rows go stale.
 * rows_per_task = 0: the library's choice for this size. */
int epic_hip_sweep2_2d(const float *d_in, float *d_out, const uint32_t *d_maskw, unsigned int rows, unsigned int pitch,
                       unsigned int rows_per_task, int math_mode, void *stream);
/* The masks a second time, cut for the fused passes' lane -> column mapping (strips of 248 columns): epic_hip_sweeps_2d takes
 * them as d_maskf and then skips a funnel shift of two mask words per row and wave (NULL: it shifts, as epic_hip_sweep2_2d,
 * which has no such parameter, always does -- a different instantiation of the pass: time and tune through the call you run).
 * epic_hip_mask_words_fused_2d(rows, pitch) uint32 words; derive after every epic_hip_pack_mask_2d. */
size_t epic_hip_mask_words_fused_2d(unsigned int rows, unsigned int pitch);
int epic_hip_fuse_masks_2d(const uint32_t *d_maskw, unsigned int rows, unsigned int pitch, uint32_t *d_maskf, void *stream);
/* `n` plain Jacobi sweeps of the whole local grid in ONE call (a driver in an interpreted language pays its per-call cost
 * once per stretch between two halo exchanges instead of once per launch): the field starts in d_a, the sweeps ping-pong
 * between d_a and d_b, *flips receives the PARITY of the number of buffer changes (1: the result is in d_b, 0: in d_a).  tol math (math_mode 4):
 * pairs of sweeps run as fused passes (rows_per_pair = 0: the library's choice), a last odd sweep singly; other modes: n single
 * sweeps.  Bit-identical to n calls of epic_hip_sweep_2d. */
int epic_hip_sweeps_2d(float *d_a, float *d_b, const uint32_t *d_maskw, const uint32_t *d_maskf, unsigned int rows,
                       unsigned int pitch, unsigned int n, unsigned int rows_per_task, unsigned int rows_per_pair, int math_mode,
                       int *flips, void *stream);
/* The same for the reference's red-black scheme: one in-place half-sweep of rows [row_begin, row_end) of d_u, updating
 * the unlocked cells with (local row + column + parity) odd (libepic/src/harmonic/harmonic_cpu.cpp:46-51 with
 * parity = currentIteration; a slab whose local row 0 is global row `top` passes (currentIteration + top) & 1). */
int epic_hip_sweep_rb_2d(float *d_u, const uint32_t *d_maskw, unsigned int rows, unsigned int pitch, unsigned int row_begin,
                         unsigned int row_end, unsigned int rows_per_task, int math_mode, int parity,
                         uint32_t *d_delta_bits, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* EPIC_HIP_H */
