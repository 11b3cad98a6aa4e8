// driver_config.cpp -- the ONE place where the library reads its environment (driver_config.h).
#include "driver_config.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <memory>
#include <mutex>

namespace epic_drv {

namespace {
// Two classes of variables.  PRODUCT knobs -- arithmetic, scheme, work lists, the device list and its transport,
// the tol mode's finishing iterations, deferred updates -- are always honoured.  STUDY knobs -- everything that only selects a code path or
// a tiling and never changes a result: thresholds, task heights, tile plans, launch flags, the tuner -- are what the tests, the fuzz
// campaigns, bench.py's A/B legs and tools/ steer the library with; a drop-in library loaded into somebody else's process should not change
// its kernel plan because a variable of that name happens to be set, so they are read only when EPIC_HIP_STUDY=1 says the caller means
// it.  A study knob that is set without it is ignored, and the library says so once on stderr.
bool g_study = false;
std::mutex g_ignored_mu;
bool g_ignored_said = false;
const char *env(const char *name) { return getenv(name); }
const char *senv(const char *name)
{
    const char *v = getenv(name);
    if (v && !g_study) {
        std::lock_guard<std::mutex> lk(g_ignored_mu);
        if (!g_ignored_said) {
            g_ignored_said = true;
            fprintf(stderr, "Warning[epic_hip]: %s is a study knob and is ignored without EPIC_HIP_STUDY=1 (INTEGRATION.md section 6); further ones are not reported.\n", name);
        }
        return nullptr;
    }
    return v;
}
bool given(const char *name) { return env(name) != nullptr; }
bool sgiven(const char *name) { return senv(name) != nullptr; }
bool is_zero(const char *name) { const char *e = env(name); return e && e[0] == '0'; }   // "=0" switches a default-on feature off
bool sis_zero(const char *name) { const char *e = senv(name); return e && e[0] == '0'; }
int sint_of(const char *name, int dflt) { const char *e = senv(name); return e ? atoi(e) : dflt; }
}  // namespace

Config Config::from_env()
{
    Config c;
    {
        const char *st = env("EPIC_HIP_STUDY");
        c.study = g_study = st != nullptr && st[0] != '0';
    }
    const char *e = env("EPIC_HIP_MATH");
    if (e && strcmp(e, "fast") == 0) c.math = 1;
    if (e && strcmp(e, "tol") == 0) c.math = 4;
    e = env("EPIC_HIP_SCHEME");
    if (e && strcmp(e, "redblack") == 0) c.redblack = true;
    if (e && strcmp(e, "jacobi") == 0) c.redblack = false;
    e = env("EPIC_HIP_JACOBI_CHECKS");
    c.jacobi_ref_checks = e && (strcmp(e, "reference") == 0 || strcmp(e, "1") == 0);
    e = env("EPIC_HIP_TRACK");
    if (e && (strcmp(e, "0") == 0 || strcmp(e, "1") == 0)) c.track_mode = atoi(e);
    c.rows_per_task = sint_of("EPIC_HIP_ROWS_PER_TASK", 0);
    e = env("EPIC_HIP_HALO");
    if (e && atoi(e) >= 1) c.halo = atoi(e);
    e = env("EPIC_HIP_DEVICES");
    if (e && *e) {  // "0,1,2,3"; a device may be named more than once ("0,0,0,0": four slabs on one GPU)
        c.devices_text = e;
        for (const char *p = e; *p;) {
            char *end = nullptr;
            const long d = strtol(p, &end, 10);
            if (end == p || d < 0 || d > 1023) { c.devices_malformed = true; break; }
            c.devices.push_back((int)d);
            p = (*end == ',') ? end + 1 : end;
            if (*end && *end != ',') { c.devices_malformed = true; break; }
        }
        if (c.devices.size() > 64) c.devices_malformed = true;
        if (c.devices_malformed) c.devices.clear();
    }
    c.no_peer = given("EPIC_HIP_NO_PEER");
    e = env("EPIC_HIP_THREADS");
    c.threads = !(e && atoi(e) == 0);
    e = env("EPIC_HIP_SPIN_US");
    if (e && atoi(e) >= 0) c.spin_us = atoi(e) > 100000 ? 100000 : atoi(e);
    c.no_fuse = sgiven("EPIC_HIP_NO_FUSE");
    c.no_graph = sgiven("EPIC_HIP_NO_GRAPH");
    e = senv("EPIC_HIP_FUSE_MIN_CELLS");
    if (e && atoll(e) >= 0) c.fuse_min_cells = atoll(e);
    e = senv("EPIC_HIP_FUSED_ROWS");
    if (e && atoi(e) > 0) c.fused_rows = atoi(e);
    c.tune = !sis_zero("EPIC_HIP_TUNE");
    c.tune_debug = sgiven("EPIC_HIP_TUNE_DEBUG");
    c.tile = !sis_zero("EPIC_HIP_TILE");
    e = senv("EPIC_HIP_TILE_MAX_CELLS");
    if (e && atoll(e) >= 0) c.tile_max_cells = atoll(e);
    c.tile_rows = sint_of("EPIC_HIP_TILE_ROWS", 0);
    c.tile_width = sint_of("EPIC_HIP_TILE_WIDTH", 0);
    e = senv("EPIC_HIP_TILE_HALO");
    if (e && atoi(e) > 0) c.tile_halo = atoi(e);
    c.tile_pipeline = !sis_zero("EPIC_HIP_TILE_PIPELINE");
    c.defer = !is_zero("EPIC_HIP_DEFER");
    c.track_pairs = !sis_zero("EPIC_HIP_TRACK_PAIRS");
    e = senv("EPIC_HIP_TRACK_PAIR_ROWS");
    if (e && atoi(e) > 0) c.track_pair_rows = atoi(e);
    e = senv("EPIC_HIP_TRACK_SWITCH");
    if (e) c.track_switch = atof(e);
    e = env("EPIC_HIP_TOL_FINISH");
    if (e) c.tol_finish = e[0] == '0' ? 0 : 1;
    e = senv("EPIC_HIP_TOL_FINISH_FACTOR");
    if (e) {
        const float v = (float)atof(e);
        if (v >= 1.0f && v <= 1e9f) c.tol_finish_factor = v;
    }
    c.launch.flags = sint_of("EPIC_HIP_FLAGS", 7);
    e = senv("EPIC_HIP_LIST_WAVES");
    if (e && atol(e) >= 4) c.launch.list_waves = (size_t)atol(e);
    c.launch.pair3d = !sis_zero("EPIC_HIP_3D_PAIR");
    e = senv("EPIC_HIP_3D_PAIR_ROWS");
    if (e && atoi(e) > 0) c.launch.pair3d_rows = atoi(e);
    e = senv("EPIC_HIP_3D_MARCH");
    c.launch.march_x0 = e && e[0] == 'x' && e[1] == '0';
    return c;
}

std::string Config::json() const
{
    // EPIC_HIP_DEVICES is the caller's text: clipped and escaped, so that whatever the environment holds this stays one JSON object
    std::string dev;
    for (char ch : devices_text.substr(0, 256)) {
        if (ch == '"' || ch == '\\') { dev += '\\'; dev += ch; }
        else if ((unsigned char)ch < 0x20) dev += ' ';
        else dev += ch;
    }
    char t[1536];
    std::string out = "{";
    snprintf(t, sizeof t, "\"study\": %s, \"math\": %d, \"scheme\": \"%s\", \"jacobi_checks\": \"%s\", \"track_mode\": %d, \"rows_per_task\": %d, \"devices\": \"",
             study ? "true" : "false", math, redblack ? "redblack" : "jacobi", jacobi_ref_checks ? "reference" : "jacobi", track_mode, rows_per_task);
    out += t;
    out += dev;
    snprintf(t, sizeof t,
             "\", \"halo\": %d, \"no_peer\": %s, "
             "\"threads\": %s, \"spin_us\": %d, \"no_fuse\": %s, \"no_graph\": %s, \"fuse_min_cells\": %lld, \"fused_rows\": %d, \"tune\": %s, "
             "\"tile\": %s, \"tile_max_cells\": %lld, \"tile_rows\": %d, \"tile_width\": %d, \"tile_halo\": %d, \"tile_pipeline\": %s, \"defer\": %s, "
             "\"track_pairs\": %s, \"track_pair_rows\": %d, \"track_switch\": %g, \"tol_finish\": %d, \"tol_finish_factor\": %g, "
             "\"flags\": %d, \"list_waves\": %zu, \"pair3d\": %s, \"pair3d_rows\": %d, \"march_x0\": %s}",
             halo, no_peer ? "true" : "false",
             threads ? "true" : "false", spin_us, no_fuse ? "true" : "false", no_graph ? "true" : "false", fuse_min_cells, fused_rows,
             tune ? "true" : "false", tile ? "true" : "false", tile_max_cells, tile_rows, tile_width, tile_halo, tile_pipeline ? "true" : "false",
             defer ? "true" : "false",
             track_pairs ? "true" : "false", track_pair_rows, track_switch, tol_finish, (double)tol_finish_factor, launch.flags, launch.list_waves,
             launch.pair3d ? "true" : "false", launch.pair3d_rows, launch.march_x0 ? "true" : "false");
    out += t;
    return out;
}

namespace {
std::mutex g_process_mu;
Config *g_process = nullptr;                       // the current one
std::vector<std::unique_ptr<Config>> g_process_all;   // every one ever made: a reference handed out earlier stays valid (reloads are a test facility)
}  // namespace

void reload_process_config()
{
    std::unique_ptr<Config> fresh(new Config(Config::from_env()));
    std::lock_guard<std::mutex> lk(g_process_mu);
    g_process = fresh.get();
    g_process_all.push_back(std::move(fresh));
}

const Config &process_config()
{
    {
        std::lock_guard<std::mutex> lk(g_process_mu);
        if (g_process) return *g_process;
    }
    reload_process_config();
    std::lock_guard<std::mutex> lk(g_process_mu);
    return *g_process;
}

}  // namespace epic_drv

namespace epic_hip {
const LaunchKnobs &process_launch_knobs() { return epic_drv::process_config().launch; }
}  // namespace epic_hip
