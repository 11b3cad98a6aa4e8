// driver_config.h -- every EPIC_HIP_* environment knob of the library in ONE struct.
//
// A Config is filled from the environment when a Harmonic's library-side context is created (driver_registry.hip: get_ctx) and
// again only when the caller asks (epic_hip_config_reload); nothing else in the library calls getenv.  epic_hip_config_dump prints
// the Config of a context together with what the planner decided from it.  The knobs are documented in INTEGRATION.md section 6;
// a value of -1 / 0 below means "not given: the library's rule".  Plain C++ (no HIP): driver_config.cpp is compiled by the host
// compiler.
#pragma once
#include <stddef.h>

#include <string>
#include <vector>

namespace epic_hip {
// What the kernel launchers (kernels_2d.hip, kernels_3d.hip, kernels.h) take from the environment.
struct LaunchKnobs {
    int flags = 7;             // EPIC_HIP_FLAGS: bit 0 = alternate march direction (2-D sweeps), bit 2 = the tol fused passes cut their rows into as many
                               // chunks as fit the last round of blocks (kernels_2d.hip: tighten_chunks); speed only, never results
    size_t list_waves = 0;     // EPIC_HIP_LIST_WAVES: persistent waves of a list-driven launch (0: what the chip holds)
    bool pair3d = true;        // EPIC_HIP_3D_PAIR=0: the one-plane-per-wave 3-D kernel for every launch
    int pair3d_rows = 0;       // EPIC_HIP_3D_PAIR_ROWS: x1-rows per task of the two-plane kernel (0: the rule)
    bool march_x0 = false;     // EPIC_HIP_3D_MARCH=x0: the two-plane kernel marches from plane to plane
};
// The process-wide knobs: the environment as it was at the first use (or at the latest epic_hip_config_reload(NULL)); what a
// launcher gets when its caller passes no knobs of its own (the raw operators of include/epic_hip.h).
const LaunchKnobs &process_launch_knobs();
}  // namespace epic_hip

namespace epic_drv {

struct Config {
    bool study = false;              // EPIC_HIP_STUDY=1: the STUDY knobs below are read at all (driver_config.cpp: two classes of variables)
    // ---- product knobs: always honoured ----
    // mode of a new context (also settable per context: epic_hip_set_*)
    int math = 0;                    // EPIC_HIP_MATH: 0 precise (default), 1 fast, 4 tol
    bool redblack = true;            // EPIC_HIP_SCHEME: redblack (default) | jacobi
    bool jacobi_ref_checks = false;  // EPIC_HIP_JACOBI_CHECKS=reference: every CHECK iteration of a Jacobi run is the reference's red-black half-sweep (driver_loop.hip: run_block)
    int track_mode = 2;              // EPIC_HIP_TRACK: 0 | 1 (2: automatic, above 4 Mcell)
    int rows_per_task = 0;           // EPIC_HIP_ROWS_PER_TASK (0: automatic)
    // several devices in one process
    std::string devices_text;        // EPIC_HIP_DEVICES as given ("" : one device)
    std::vector<int> devices;        //   parsed ordinals (validated against the device count by the registry); empty if malformed
    bool devices_malformed = false;
    int halo = 0;                    // EPIC_HIP_HALO (0: by slab height)
    bool no_peer = false;            // EPIC_HIP_NO_PEER: ghost units through pinned host memory
    bool threads = true;             // EPIC_HIP_THREADS=0: the calling thread issues every slab's launches
    int spin_us = 20;                // EPIC_HIP_SPIN_US: how long an issuing thread (and the caller waiting for them) spins before it sleeps
    // which kernel family runs a batch
    bool no_fuse = false;            // EPIC_HIP_NO_FUSE
    bool no_graph = false;           // EPIC_HIP_NO_GRAPH
    long long fuse_min_cells = -1;   // EPIC_HIP_FUSE_MIN_CELLS (-1: by arithmetic and scheme, driver_plan.hip: fuse_from_cells)
    int fused_rows = 0;              // EPIC_HIP_FUSED_ROWS (0: measured / the rule)
    bool tune = true;                // EPIC_HIP_TUNE=0: the rules only
    bool tune_debug = false;         // EPIC_HIP_TUNE_DEBUG
    bool tile = true;                // EPIC_HIP_TILE=0
    long long tile_max_cells = -1;   // EPIC_HIP_TILE_MAX_CELLS (-1: by arithmetic and scheme, driver_plan.hip: tile_up_to_cells)
    int tile_rows = 0, tile_width = 0, tile_halo = 0;   // EPIC_HIP_TILE_ROWS / _WIDTH / _HALO (0: the cost model)
    bool tile_pipeline = true;       // EPIC_HIP_TILE_PIPELINE=0
    bool defer = true;               // EPIC_HIP_DEFER=0: harmonic_update_gpu launches one single iteration per call (as before round 6) instead of counting
    bool track_pairs = true;         // EPIC_HIP_TRACK_PAIRS=0
    int track_pair_rows = 0;         // EPIC_HIP_TRACK_PAIR_ROWS (0: 16, then 4 in the tail)
    double track_switch = -1.0;      // EPIC_HIP_TRACK_SWITCH (< 0: by configuration)
    // the tol mode's finishing iterations
    int tol_finish = -1;             // EPIC_HIP_TOL_FINISH: 0 off (honoured for epsilon <= 1e-5), -1 not given
    float tol_finish_factor = 0.0f;  // EPIC_HIP_TOL_FINISH_FACTOR (0: 10 / 100)
    epic_hip::LaunchKnobs launch;    // EPIC_HIP_FLAGS, _LIST_WAVES, _3D_PAIR, _3D_PAIR_ROWS, _3D_MARCH

    static Config from_env();        // the one place that reads the environment
    std::string json() const;        // every field, as one JSON object
};

// The process-wide Config: the environment as it was at the first use, or at the latest reload_process_config()
// (epic_hip_config_reload(NULL)) -- what the entry points without a context (the raw operators of include/epic_hip.h) go by.
const Config &process_config();
void reload_process_config();

}  // namespace epic_drv
