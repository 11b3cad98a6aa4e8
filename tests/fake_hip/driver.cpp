// Fault-injection driver for the library's HOST driver (epic_amd/csrc/driver_*.hip compiled against the fake HIP runtime
// of this directory; tests/test_host_driver_faults.py builds and runs it under ASan + UBSan and under TSan).
//
// For every scenario -- the plugin's one call, the navigation node's fine-grained flow, 3-D, the tile path and the plain
// sweeps, forced work lists, the tol mode's loop rules, several slabs with and without issuing threads and staged halos --
// the call sequence is first run cleanly to COUNT the fallible runtime calls it makes, then once per n with the n-th call
// failing.  After each run the callers' own clean-up is made (the four uninitialize calls, as libepic/python/epic/
// harmonic.py:88-92 and the node's uninitAlg do whatever happened) and the following must hold:
//   * the return code is a code of the reference (error_codes.h:31-46), and it is an error code unless the library
//     has a fallback for what failed (graph capture -> eager launches, a failed timing event -> the rule's task height, ...);
//   * nothing is left behind: no device or pinned allocation, stream or event is live, nothing was freed twice or copied
//     outside an allocation, and the struct's d_* pointers are null (the reference leaks on several of these paths:
//     harmonic_model_gpu.cu:50-55, harmonic_utilities_gpu.cu:81-135);
//   * the same sequence runs cleanly afterwards.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <functional>
#include <map>
#include <set>
#include <string>
#include <vector>

#include <epic/epic_abi.h>
#include <epic_hip.h>

#include "../../epic_amd/csrc/kernels.h"   // tile_2d_plan (host logic of the small-grid path)

using namespace epic;

static int failures = 0;
#define EXPECT(cond, ...)                                                                     \
    do {                                                                                      \
        if (!(cond)) {                                                                        \
            fprintf(stderr, "EXPECT failed: %s (line %d) ", #cond, __LINE__);                 \
            fprintf(stderr, __VA_ARGS__);                                                     \
            fprintf(stderr, "\n");                                                            \
            failures++;                                                                       \
        }                                                                                     \
    } while (0)

struct Grid {
    std::vector<unsigned> m;
    std::vector<float> u;
    std::vector<unsigned> lk;
    Harmonic h;
    Grid(std::vector<unsigned> dims, float eps, unsigned stagger) : m(dims)
    {
        size_t cells = 1;
        for (unsigned d : m) cells *= d;
        u.assign(cells, -1e6f);
        lk.assign(cells, 0u);
        unsigned s = 12345u;
        for (size_t i = 0; i < cells; i++) {
            s = s * 1664525u + 1013904223u;
            size_t rem = i;
            bool border = false;
            for (size_t d = m.size(); d-- > 0;) { const unsigned x = rem % m[d]; rem /= m[d]; border |= x == 0 || x == m[d] - 1; }
            if (border || (s >> 24) < 12) lk[i] = 1;
        }
        size_t goal = 0;
        for (unsigned d : m) goal = goal * d + d / 2;
        lk[goal] = 1;
        u[goal] = 0.0f;
        memset(&h, 0, sizeof h);
        h.n = (unsigned)m.size();
        h.m = m.data();
        h.u = u.data();
        h.locked = lk.data();
        h.epsilon = eps;
        h.numIterationsToStaggerCheck = stagger;
    }
};

static void cleanup(Harmonic *h)   // what every caller does after a failure (and python's solve() always)
{
    (void)harmonic_uninitialize_dimension_size_gpu(h);
    (void)harmonic_uninitialize_potential_values_gpu(h);
    (void)harmonic_uninitialize_locked_gpu(h);
    (void)harmonic_uninitialize_gpu(h);
}

// ---- the call sequences -----------------------------------------------------------------------------------------
static int seq_plugin(Grid &g) { return harmonic_complete_gpu(&g.h, 1024); }   // src/epic_nav_core_plugin.cpp:256

static int seq_python(Grid &g)   // libepic/python/epic/harmonic.py:67-76: initialize x 3, then complete (which re-initialises)
{
    int rc = harmonic_initialize_dimension_size_gpu(&g.h);
    rc += harmonic_initialize_potential_values_gpu(&g.h);
    rc += harmonic_initialize_locked_gpu(&g.h);
    if (rc != 0) return rc > 13 ? 13 : rc;
    return harmonic_complete_gpu(&g.h, 1024);
}

static int seq_node(Grid &g)   // src/epic_navigation_node_harmonic.cpp: initAlg, update(k), setCells, srvComputePath, update_model
{
    int rc = harmonic_initialize_dimension_size_gpu(&g.h);
    if (rc) return rc;
    if ((rc = harmonic_initialize_potential_values_gpu(&g.h))) return rc;
    if ((rc = harmonic_initialize_locked_gpu(&g.h))) return rc;
    if ((rc = harmonic_initialize_gpu(&g.h, 1024))) return rc;
    for (int round = 0; round < 2; round++) {
        rc = harmonic_update_and_check_gpu(&g.h, 1024);
        if (rc != EPIC_SUCCESS && rc != EPIC_SUCCESS_AND_CONVERGED) return rc;
        for (int i = 0; i < 3; i++)
            if ((rc = harmonic_update_gpu(&g.h, 1024))) return rc;
        if (g.h.n == 2) {
            unsigned v[4] = {3, 4, 5, 2}, types[2] = {EPIC_CELL_TYPE_OBSTACLE, EPIC_CELL_TYPE_GOAL};
            if ((rc = harmonic_utilities_set_cells_2d_cpu(&g.h, 2, v, types))) return rc;
            if ((rc = harmonic_utilities_set_cells_2d_gpu(&g.h, 1024, 2, v, types))) return rc;
        }
        rc = epic_hip_update_n_gpu(&g.h, 12, 1);
        if (rc != EPIC_SUCCESS && rc != EPIC_SUCCESS_AND_CONVERGED) return rc;
        if ((rc = harmonic_get_potential_values_gpu(&g.h))) return rc;
    }
    if ((rc = harmonic_update_model_gpu(&g.h))) return rc;
    if ((rc = epic_hip_update_n_gpu(&g.h, 5, 0))) return rc;
    if ((rc = harmonic_uninitialize_gpu(&g.h))) return rc;
    if ((rc = harmonic_uninitialize_dimension_size_gpu(&g.h))) return rc;
    if ((rc = harmonic_uninitialize_potential_values_gpu(&g.h))) return rc;
    return harmonic_uninitialize_locked_gpu(&g.h);
}

// The node's ticks with enough plain updates per tick for whole deferred blocks (driver_loop.hip: harmonic_update_gpu counts): blocks
// enqueued at the cap, the block enqueued AHEAD of the caller behind a check and adopted, a read-back and an edit in the middle of a
// block (the block ahead is discarded), a renumbered iteration, and a teardown with iterations pending.
static int seq_node_ticks(Grid &g)
{
    int rc = harmonic_initialize_dimension_size_gpu(&g.h);
    if (rc) return rc;
    if ((rc = harmonic_initialize_potential_values_gpu(&g.h))) return rc;
    if ((rc = harmonic_initialize_locked_gpu(&g.h))) return rc;
    if ((rc = harmonic_initialize_gpu(&g.h, 1024))) return rc;
    for (int tick = 0; tick < 4; tick++) {
        rc = harmonic_update_and_check_gpu(&g.h, 1024);
        if (rc != EPIC_SUCCESS && rc != EPIC_SUCCESS_AND_CONVERGED) return rc;
        for (int i = 0; i < (tick == 2 ? 5 : 39); i++)   // (tick 2 stops inside the block that was enqueued ahead)
            if ((rc = harmonic_update_gpu(&g.h, 1024))) return rc;
        if (tick == 2) {
            if ((rc = harmonic_get_potential_values_gpu(&g.h))) return rc;
            if (g.h.n == 2) {
                unsigned v[2] = {3, 4}, types[1] = {EPIC_CELL_TYPE_OBSTACLE};
                if ((rc = harmonic_utilities_set_cells_2d_cpu(&g.h, 1, v, types))) return rc;
                if ((rc = harmonic_utilities_set_cells_2d_gpu(&g.h, 1024, 1, v, types))) return rc;
            }
        }
    }
    g.h.currentIteration = 7;   // the caller renumbers with iterations pending
    for (int i = 0; i < 3; i++)
        if ((rc = harmonic_update_gpu(&g.h, 1024))) return rc;
    if ((rc = harmonic_uninitialize_dimension_size_gpu(&g.h))) return rc;   // (the pending iterations surface here)
    if ((rc = harmonic_get_potential_values_gpu(&g.h))) return rc;
    if ((rc = harmonic_uninitialize_gpu(&g.h))) return rc;
    if ((rc = harmonic_uninitialize_potential_values_gpu(&g.h))) return rc;
    return harmonic_uninitialize_locked_gpu(&g.h);
}

struct Scenario {
    const char *name;
    std::vector<unsigned> dims;
    float eps;
    unsigned stagger;
    std::function<int(Grid &)> run;
    std::map<std::string, std::string> env;
};

static void set_env(const std::map<std::string, std::string> &env, bool on)
{
    static const char *all[] = {"EPIC_HIP_DEVICES", "EPIC_HIP_THREADS", "EPIC_HIP_NO_PEER", "EPIC_HIP_TRACK", "EPIC_HIP_TILE", "EPIC_HIP_MATH",
                                "EPIC_HIP_SCHEME", "EPIC_HIP_HALO", "EPIC_HIP_FUSE_MIN_CELLS", "EPIC_HIP_TUNE", "EPIC_HIP_DEFER", "EPIC_HIP_JACOBI_CHECKS", "FAKE_NO_PEER_CAPABLE", "FAKE_CURRENT_DEVICE"};
    for (const char *k : all) unsetenv(k);
    setenv("EPIC_HIP_STUDY", "1", 1);   // the scenarios steer the kernel plan with study knobs (epic_amd/csrc/driver_config.cpp)
    if (on)
        for (auto &kv : env) setenv(kv.first.c_str(), kv.second.c_str(), 1);
}

static bool clean(const char *what, const Scenario &sc, long n, const Harmonic &h)
{
    const int before = failures;
    for (int kind = 0; kind < 5; kind++)
        EXPECT(fake_hip_live(kind) == 0, "%s, call %ld failing (%s): %ld live objects of kind %d after %s", sc.name, n, fake_hip_failed_call(),
               fake_hip_live(kind), kind, what);
    EXPECT(fake_hip_misuse() == 0, "%s, call %ld failing (%s): %ld misuses of the runtime", sc.name, n, fake_hip_failed_call(), fake_hip_misuse());
    EXPECT(fake_hip_affinity() == 0, "%s, call %ld failing (%s): %ld violations of the device rules, the first: %s", sc.name, n, fake_hip_failed_call(),
           fake_hip_affinity(), fake_hip_first_affinity());
    EXPECT(!h.d_m && !h.d_u && !h.d_locked && !h.d_delta, "%s, call %ld failing (%s): d_* not null after %s", sc.name, n, fake_hip_failed_call(), what);
    return failures == before;
}

int main(int argc, char **argv)
{
    const bool threads_only = argc > 1 && strcmp(argv[1], "threads") == 0;   // the TSan build: the scenarios with issuing threads, no walk
    // "devices": FOUR fake devices with the real runtime's device rules enforced (fake_hip.cpp: device affinity) -- every slab on a
    // device of its own, lists out of order and with repeats, with and without peer access, the caller's current device not 0.
    // No session of this project has had two GPUs: this is the multi-device mode's device bookkeeping under test.
    const bool devices_mode = argc > 1 && strcmp(argv[1], "devices") == 0;
    fake_hip_set_devices(devices_mode ? 4 : 1);
    std::vector<Scenario> device_scenarios = {
        {"plugin, 2-D, four devices, issuing threads", {64, 40}, 1e-3f, 10, seq_plugin, {{"EPIC_HIP_DEVICES", "0,1,2,3"}, {"EPIC_HIP_HALO", "4"}}},
        {"plugin, 2-D, four devices, caller's thread", {64, 40}, 1e-3f, 10, seq_plugin, {{"EPIC_HIP_DEVICES", "0,1,2,3"}, {"EPIC_HIP_THREADS", "0"}, {"EPIC_HIP_HALO", "3"}}},
        {"node, 2-D, devices 2,0,3 (out of order), work lists, issuing threads", {48, 300}, 1e-3f, 4, seq_node, {{"EPIC_HIP_DEVICES", "2,0,3"}, {"EPIC_HIP_TRACK", "1"}}},
        {"node, 2-D, devices 1,1,3 (one device twice), caller's thread", {48, 40}, 1e-3f, 4, seq_node, {{"EPIC_HIP_DEVICES", "1,1,3"}, {"EPIC_HIP_THREADS", "0"}}},
        {"plugin, 3-D, planes on devices 3,1", {12, 6, 7}, 1e-3f, 5, seq_plugin, {{"EPIC_HIP_DEVICES", "3,1"}, {"EPIC_HIP_THREADS", "0"}}},
        {"plugin, 2-D, tol fused pairs on devices 0,1,2", {72, 300}, 1e-3f, 10, seq_plugin,
         {{"EPIC_HIP_DEVICES", "0,1,2"}, {"EPIC_HIP_MATH", "tol"}, {"EPIC_HIP_SCHEME", "jacobi"}, {"EPIC_HIP_FUSE_MIN_CELLS", "0"}, {"EPIC_HIP_HALO", "6"}}},
        {"plugin, 2-D, four devices that CANNOT reach each other (staged halos)", {64, 40}, 1e-3f, 10, seq_plugin,
         {{"EPIC_HIP_DEVICES", "0,1,2,3"}, {"EPIC_HIP_THREADS", "0"}, {"EPIC_HIP_HALO", "2"}, {"FAKE_NO_PEER_CAPABLE", "1"}}},
        {"plugin, 2-D, devices 0,2 with staging forced, issuing threads", {64, 40}, 1e-3f, 10, seq_plugin, {{"EPIC_HIP_DEVICES", "0,2"}, {"EPIC_HIP_NO_PEER", "1"}}},
        {"plugin, 2-D, ONE device, the caller's current device is 2", {20, 30}, 1e-3f, 10, seq_plugin, {{"FAKE_CURRENT_DEVICE", "2"}}},
        {"plugin, 2-D, devices 1,3, tracked pairs per slab", {60, 300}, 1e-3f, 10, seq_plugin, {{"EPIC_HIP_DEVICES", "1,3"}, {"EPIC_HIP_TRACK", "1"}, {"EPIC_HIP_FUSE_MIN_CELLS", "0"}, {"EPIC_HIP_HALO", "4"}}},
        {"node, 2-D, slabs on 0,1 while the caller's current device is 3", {32, 20}, 1e-3f, 4, seq_node, {{"EPIC_HIP_DEVICES", "0,1"}, {"FAKE_CURRENT_DEVICE", "3"}}},
    };
    std::vector<Scenario> scenarios = {
        {"plugin, 2-D, tiles", {20, 30}, 1e-3f, 10, seq_plugin, {}},
        {"plugin, 2-D, plain sweeps", {20, 30}, 1e-3f, 10, seq_plugin, {{"EPIC_HIP_TILE", "0"}}},
        {"python solve, 2-D", {12, 40}, 1e-2f, 7, seq_python, {}},
        {"plugin, 3-D", {6, 7, 9}, 1e-3f, 5, seq_plugin, {}},
        {"node, 2-D", {16, 18}, 1e-3f, 4, seq_node, {}},
        {"node ticks, 2-D, deferred blocks on tiles with run-ahead", {40, 70}, 1e-9f, 100, seq_node_ticks, {}},
        {"node ticks, 2-D, one launch per call (EPIC_HIP_DEFER=0)", {40, 70}, 1e-9f, 100, seq_node_ticks, {{"EPIC_HIP_DEFER", "0"}}},
        {"node ticks, 2-D, deferred pairs with work lists", {40, 300}, 1e-9f, 100, seq_node_ticks, {{"EPIC_HIP_TRACK", "1"}, {"EPIC_HIP_FUSE_MIN_CELLS", "0"}}},
        {"node ticks, 2-D, deferred fused tol pairs", {24, 300}, 1e-9f, 100, seq_node_ticks, {{"EPIC_HIP_MATH", "tol"}, {"EPIC_HIP_SCHEME", "jacobi"}, {"EPIC_HIP_FUSE_MIN_CELLS", "0"}, {"EPIC_HIP_TILE", "0"}}},
        {"node ticks, 3-D", {5, 6, 7}, 1e-9f, 100, seq_node_ticks, {}},
        {"node ticks, 2-D, two slabs, deferred stretches", {32, 20}, 1e-9f, 100, seq_node_ticks, {{"EPIC_HIP_DEVICES", "0,0"}, {"EPIC_HIP_THREADS", "0"}}},
        {"node, 3-D", {5, 6, 7}, 1e-3f, 4, seq_node, {}},
        {"plugin, 2-D, work lists", {40, 300}, 1e-3f, 10, seq_plugin, {{"EPIC_HIP_TRACK", "1"}}},
        {"plugin, 2-D, tracked pairs of fused passes", {40, 300}, 1e-3f, 10, seq_plugin, {{"EPIC_HIP_TRACK", "1"}, {"EPIC_HIP_FUSE_MIN_CELLS", "0"}}},
        {"plugin, 2-D, tracked pairs, tol red-black", {40, 300}, 1e-3f, 10, seq_plugin, {{"EPIC_HIP_TRACK", "1"}, {"EPIC_HIP_FUSE_MIN_CELLS", "0"}, {"EPIC_HIP_MATH", "tol"}}},
        {"plugin, 2-D, tracked pairs, tol Jacobi", {40, 300}, 1e-3f, 10, seq_plugin, {{"EPIC_HIP_TRACK", "1"}, {"EPIC_HIP_FUSE_MIN_CELLS", "0"}, {"EPIC_HIP_MATH", "tol"}, {"EPIC_HIP_SCHEME", "jacobi"}}},
        {"plugin, 2-D, tracked pairs, odd count", {40, 300}, 1e-3f, 7, seq_plugin, {{"EPIC_HIP_TRACK", "1"}, {"EPIC_HIP_FUSE_MIN_CELLS", "0"}}},
        {"plugin, 2-D, tol Jacobi (handover and finish rules)", {20, 30}, 1e-3f, 10, seq_plugin, {{"EPIC_HIP_MATH", "tol"}, {"EPIC_HIP_SCHEME", "jacobi"}}},
        {"plugin, 2-D, tol fused pairs", {24, 300}, 1e-3f, 10, seq_plugin, {{"EPIC_HIP_MATH", "tol"}, {"EPIC_HIP_SCHEME", "jacobi"}, {"EPIC_HIP_FUSE_MIN_CELLS", "0"}}},
        {"plugin, 2-D, Jacobi on tiles, the reference's half-sweep at every check", {20, 30}, 1e-3f, 10, seq_plugin, {{"EPIC_HIP_SCHEME", "jacobi"}, {"EPIC_HIP_JACOBI_CHECKS", "reference"}}},
        {"node, 2-D, tol Jacobi pairs with work lists, reference checks", {40, 300}, 1e-3f, 4, seq_node,
         {{"EPIC_HIP_MATH", "tol"}, {"EPIC_HIP_SCHEME", "jacobi"}, {"EPIC_HIP_JACOBI_CHECKS", "reference"}, {"EPIC_HIP_TRACK", "1"}, {"EPIC_HIP_FUSE_MIN_CELLS", "0"}}},
        {"plugin, 2-D, two slabs, Jacobi with reference checks", {32, 40}, 1e-3f, 10, seq_plugin,
         {{"EPIC_HIP_DEVICES", "0,0"}, {"EPIC_HIP_THREADS", "0"}, {"EPIC_HIP_SCHEME", "jacobi"}, {"EPIC_HIP_JACOBI_CHECKS", "reference"}, {"EPIC_HIP_HALO", "3"}}},
        {"plugin, 2-D, three slabs, caller's thread", {48, 40}, 1e-3f, 10, seq_plugin, {{"EPIC_HIP_DEVICES", "0,0,0"}, {"EPIC_HIP_THREADS", "0"}, {"EPIC_HIP_HALO", "3"}}},
        {"plugin, 2-D, three slabs, staged halos", {48, 40}, 1e-3f, 10, seq_plugin, {{"EPIC_HIP_DEVICES", "0,0,0"}, {"EPIC_HIP_THREADS", "0"}, {"EPIC_HIP_NO_PEER", "1"}, {"EPIC_HIP_HALO", "2"}}},
        {"node, 2-D, two slabs, caller's thread", {32, 20}, 1e-3f, 4, seq_node, {{"EPIC_HIP_DEVICES", "0,0"}, {"EPIC_HIP_THREADS", "0"}}},
        {"plugin, 3-D, two slabs of planes", {12, 6, 7}, 1e-3f, 5, seq_plugin, {{"EPIC_HIP_DEVICES", "0,0"}, {"EPIC_HIP_THREADS", "0"}}},
        {"plugin, 2-D, four slabs, issuing threads", {64, 40}, 1e-3f, 10, seq_plugin, {{"EPIC_HIP_DEVICES", "0,0,0,0"}, {"EPIC_HIP_HALO", "4"}}},
        {"node, 2-D, three slabs, issuing threads, work lists", {48, 300}, 1e-3f, 4, seq_node, {{"EPIC_HIP_DEVICES", "0,0,0"}, {"EPIC_HIP_TRACK", "1"}}},
        {"plugin, 2-D, three slabs, tracked pairs per slab, caller's thread", {60, 300}, 1e-3f, 10, seq_plugin, {{"EPIC_HIP_DEVICES", "0,0,0"}, {"EPIC_HIP_THREADS", "0"}, {"EPIC_HIP_TRACK", "1"}, {"EPIC_HIP_FUSE_MIN_CELLS", "0"}, {"EPIC_HIP_HALO", "4"}}},
        {"plugin, 2-D, two slabs, tracked tol pairs, staged halos, odd count", {40, 300}, 1e-3f, 7, seq_plugin, {{"EPIC_HIP_DEVICES", "0,0"}, {"EPIC_HIP_THREADS", "0"}, {"EPIC_HIP_TRACK", "1"}, {"EPIC_HIP_FUSE_MIN_CELLS", "0"}, {"EPIC_HIP_MATH", "tol"}, {"EPIC_HIP_NO_PEER", "1"}, {"EPIC_HIP_HALO", "3"}}},
        {"node ticks, 2-D, two slabs, tracked pairs, issuing threads", {48, 300}, 1e-9f, 100, seq_node_ticks, {{"EPIC_HIP_DEVICES", "0,0"}, {"EPIC_HIP_TRACK", "1"}, {"EPIC_HIP_FUSE_MIN_CELLS", "0"}, {"EPIC_HIP_HALO", "4"}}},
    };
    // host logic of the small-grid path: whatever the grid and the ring depth, the tiles cover the grid, fit the LDS tile and
    // leave at least two owned rows and columns; a plan that cannot be made says so (halo == 0)
    if (!threads_only) {
        long plans = 0;
        for (int rows : {3, 4, 17, 64, 256, 310, 482, 1000, 1024})
            for (int cols : {3, 5, 48, 49, 256, 482, 940, 1024})
                for (int halo : {1, 2, 5, 8, 12, 14, 27, 28, 40})
                    for (int want : {0, 2, 7, 20, 64}) {
                        const epic_hip::TilePlan p = epic_hip::tile_2d_plan(rows, cols, halo, want);
                        if (p.halo == 0) { EXPECT(2 * halo >= epic_hip::kTile2dCols - 8, "no plan for %d x %d, halo %d", rows, cols, halo); continue; }
                        plans++;
                        EXPECT(p.halo == halo && p.tile_cols == epic_hip::kTile2dCols - 2 * halo && p.tile_cols >= 8, "columns: %d x %d halo %d", rows, cols, halo);
                        EXPECT(p.tile_rows >= 2 && p.tile_rows % 2 == 0 && p.tile_rows + 2 * halo <= epic_hip::kTile2dMaxRows, "rows: %d x %d halo %d -> %d", rows, cols, halo, p.tile_rows);
                        EXPECT((long)p.tiles_r * p.tile_rows >= rows && (long)(p.tiles_r - 1) * p.tile_rows < rows, "row cover: %d x %d halo %d", rows, cols, halo);
                        EXPECT((long)p.tiles_c * p.tile_cols >= cols && (long)(p.tiles_c - 1) * p.tile_cols < cols, "column cover: %d x %d halo %d", rows, cols, halo);
                    }
        printf("tile plans checked: %ld\n", plans);
    }
    long walked = 0, tolerated = 0;
    std::map<std::string, std::set<int>> codes;   // failing call -> return codes seen
    if (devices_mode) scenarios = device_scenarios;
    for (const Scenario &sc : scenarios) {
        const bool threaded = sc.env.count("EPIC_HIP_DEVICES") && !sc.env.count("EPIC_HIP_THREADS");
        if (threads_only && !threaded) continue;
        set_env(sc.env, true);
        fake_hip_set_peer_capable(sc.env.count("FAKE_NO_PEER_CAPABLE") ? 0 : 1);
        const int caller_device = sc.env.count("FAKE_CURRENT_DEVICE") ? atoi(sc.env.at("FAKE_CURRENT_DEVICE").c_str()) : 0;
        (void)hipSetDevice(caller_device);
        long total;
        {
            Grid g(sc.dims, sc.eps, sc.stagger);
            fake_hip_fail_at(0);
            const int rc = sc.run(g);
            total = fake_hip_calls();
            EXPECT(rc == 0, "%s: clean run returned %d", sc.name, rc);
            cleanup(&g.h);
            clean("the clean run", sc, 0, g.h);
            int now = -1;
            (void)hipGetDevice(&now);
            EXPECT(now == caller_device, "%s: the caller's current device was %d and is %d after the calls", sc.name, caller_device, now);
        }
        printf("%-60s %5ld fallible runtime calls\n", sc.name, total);
        // with issuing threads the order of the calls is not deterministic: walk a sample there (every call is still some n)
        // (the node-ticks scenarios of round 6 make several hundred calls each, most of them the same launch in another tick: every third)
        const bool long_script = strncmp(sc.name, "node ticks", 10) == 0;
        const long stride = threads_only || devices_mode ? (total > 40 ? total / 40 : 1) : (threaded || long_script ? 3 : 1);
        for (long n = 1; n <= total; n += stride) {
            Grid g(sc.dims, sc.eps, sc.stagger);
            fake_hip_fail_at(n);
            const int rc = sc.run(g);
            const bool fired = fake_hip_failed() != 0;
            const std::string what = fake_hip_failed_call();
            cleanup(&g.h);
            EXPECT(rc >= 0 && rc <= 13, "%s, call %ld (%s): return code %d is not one of the reference's", sc.name, n, what.c_str(), rc);
            if (fired) {
                codes[what].insert(rc);
                if (rc == 0) tolerated++;
            } else {
                EXPECT(rc == 0, "%s, call %ld: nothing failed, yet the sequence returned %d", sc.name, n, rc);
            }
            walked++;
            if (!clean("the unwind", sc, n, g.h)) break;
            // and the library is as good as new: the same sequence on a fresh struct at the SAME address
            fake_hip_fail_at(0);
            Grid g2(sc.dims, sc.eps, sc.stagger);
            g.h = g2.h;
            const int rc2 = sc.run(g);
            EXPECT(rc2 == 0, "%s, after call %ld (%s) had failed: the next run returned %d", sc.name, n, what.c_str(), rc2);
            cleanup(&g.h);
            if (!clean("the run after the unwind", sc, n, g.h)) break;
            int now = -1;
            (void)hipGetDevice(&now);
            EXPECT(now == caller_device, "%s, call %ld (%s): the caller's current device was %d and is %d", sc.name, n, what.c_str(), caller_device, now);
        }
        set_env(sc.env, false);
        (void)hipSetDevice(0);
    }
    printf("walked %ld failing calls (%ld tolerated by a fallback)\n", walked, tolerated);
    for (auto &kv : codes) {
        printf("  %-28s ->", kv.first.c_str());
        for (int c : kv.second) printf(" %d", c);
        printf("\n");
    }
    // what failed must surface as the reference's code for it (error_codes.h:31-46; 0 where the library has a fallback)
    const std::map<std::string, std::set<int>> allowed = {
        {"hipMalloc", {0, EPIC_ERROR_DEVICE_MALLOC}}, {"hipHostMalloc", {0, EPIC_ERROR_DEVICE_MALLOC}},
        {"hipStreamCreateWithFlags", {EPIC_ERROR_DEVICE_MALLOC}},
        {"hipEventCreateWithFlags", {0, EPIC_ERROR_DEVICE_MALLOC}},   // (0: the events of the pipelined small-grid loop -- the plain loop serves)
        {"hipEventCreate", {0}},   // (only the timing of candidate task heights uses it: the rule's height serves)
        {"hipMemcpy", {0, EPIC_ERROR_MEMCPY_TO_DEVICE, EPIC_ERROR_MEMCPY_TO_HOST}},   // (0: reading the list counters is advisory -- which tiling, bypass or not)
        {"hipMemcpy2D", {EPIC_ERROR_MEMCPY_TO_DEVICE, EPIC_ERROR_MEMCPY_TO_HOST}},
        {"hipMemcpyAsync", {EPIC_ERROR_MEMCPY_TO_DEVICE, EPIC_ERROR_MEMCPY_TO_HOST, EPIC_ERROR_KERNEL_EXECUTION}},
        {"hipMemcpyPeerAsync", {EPIC_ERROR_KERNEL_EXECUTION}},
        {"hipMemsetAsync", {0, EPIC_ERROR_KERNEL_EXECUTION}}, {"hipMemset", {0}},
        {"hipStreamSynchronize", {0, EPIC_ERROR_MEMCPY_TO_HOST, EPIC_ERROR_KERNEL_EXECUTION, EPIC_ERROR_DEVICE_SYNCHRONIZE}},
        {"hipStreamWaitEvent", {EPIC_ERROR_KERNEL_EXECUTION}},
        {"hipEventRecord", {0, EPIC_ERROR_KERNEL_EXECUTION}},   // (0: the event in front of the block enqueued ahead of the caller -- the check goes on without that block)
        {"hipEventSynchronize", {0, EPIC_ERROR_DEVICE_SYNCHRONIZE}}, {"hipEventQuery", {EPIC_ERROR_DEVICE_SYNCHRONIZE}},
    };
    for (auto &kv : codes) {
        auto it = allowed.find(kv.first);
        for (int c : kv.second) {
            if (kv.first == "launch_fill" && c == 0) continue;   // (seeding the third buffer of the pipelined small-grid loop: the plain loop serves)
            if (kv.first == "launch_tile_2d" && c == 0) continue;   // (the block enqueued AHEAD of the caller behind a check, Ctx::ahead: the check goes on without it)
            if (kv.first.rfind("launch_", 0) == 0) EXPECT(c == EPIC_ERROR_KERNEL_EXECUTION, "%s failing gave code %d", kv.first.c_str(), c);
            else if (it == allowed.end()) EXPECT(false, "no expectation for %s (code %d)", kv.first.c_str(), c);
            else EXPECT(it->second.count(c) == 1, "%s failing gave code %d", kv.first.c_str(), c);
        }
    }
    if (failures == 0) printf("fault driver: ok\n");
    return failures == 0 ? 0 : 1;
}
