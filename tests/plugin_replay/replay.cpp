// replay.cpp -- the call sequences of the reference's two C++ callers, replayed against libepic.so on a device.
//
// Test infrastructure (tests/test_gpu_plugin_replay.py builds and runs it).  This translation unit includes ONLY the
// header paths the reference's callers include (<epic/harmonic/...>, <epic/constants.h>, <epic/error_codes.h>), uses
// the reference's names with their C++ reference parameters (float *&path, unsigned int &k), allocates with new[] and
// releases paths with delete[] as the callers do, and links with -lepic: what a maintainer gets who rebuilds the ROS
// package against this library.  ROS itself is not in the image, so the message plumbing around the calls is not
// replayed -- the library calls, their order, arguments and ownership are:
//   plugin  src/epic_nav_core_plugin.cpp:234-338  makePlan: setGoal edits on the host arrays (:341-366) ->
//           harmonic_complete_gpu(&harmonic, 1024) -> (fallback harmonic_complete_cpu) -> harmonic_compute_path_2d_cpu
//           with step 0.05, precision 0.5, max_length m0 m1 / step -> delete [] raw_plan
//   node    src/epic_navigation_node_harmonic.cpp:208-244 initAlg (zeroed arrays, boundaries, initialize x 4),
//           :357-380 setCells (CPU arrays, then GPU), :165-206 update(num_steps) = update_and_check + plain updates,
//           :614-674 srvComputePath = harmonic_get_potential_values_gpu + harmonic_compute_path_2d_cpu -> delete []
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <epic/constants.h>
#include <epic/error_codes.h>
#include <epic/harmonic/harmonic.h>
#include <epic/harmonic/harmonic_cpu.h>
#include <epic/harmonic/harmonic_gpu.h>
#include <epic/harmonic/harmonic_model_gpu.h>
#include <epic/harmonic/harmonic_path_cpu.h>
#include <epic/harmonic/harmonic_utilities_cpu.h>
#include <epic/harmonic/harmonic_utilities_gpu.h>

using namespace epic;

#define NUM_THREADS_GPU 1024

struct Input {
    unsigned rows = 0, cols = 0;
    float epsilon = 1e-3f;                        // the callers' own: src/epic_nav_core_plugin.cpp:61,85, src/epic_navigation_node_harmonic.cpp:64
    std::vector<unsigned char> occupied;          // rows x cols, 1 = obstacle
    std::vector<unsigned> goals;                  // (x, y) pairs
    std::vector<float> starts;                    // x, y, step, precision
};

static bool read_input(const char *path, Input &in)
{
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    unsigned hdr[5];   // rows, cols, goals, starts, epsilon (float bits)
    bool ok = fread(hdr, 4, 5, f) == 5;
    if (ok) {
        in.rows = hdr[0]; in.cols = hdr[1];
        memcpy(&in.epsilon, &hdr[4], 4);
        in.occupied.resize((size_t)in.rows * in.cols);
        in.goals.resize(2 * (size_t)hdr[2]);
        in.starts.resize(4 * (size_t)hdr[3]);
        ok = fread(in.occupied.data(), 1, in.occupied.size(), f) == in.occupied.size() &&
             fread(in.goals.data(), 4, in.goals.size(), f) == in.goals.size() &&
             fread(in.starts.data(), 4, in.starts.size(), f) == in.starts.size();
    }
    fclose(f);
    return ok;
}

static void write_path(FILE *out, int rc, unsigned k, const float *path)
{
    fwrite(&rc, 4, 1, out);
    fwrite(&k, 4, 1, out);
    if (rc == EPIC_SUCCESS && k > 0) fwrite(path, 4, 2 * (size_t)k, out);
}

// ---- the navigation node ---------------------------------------------------------------------------------------
struct Node {
    Harmonic harmonic;
    bool gpu = false;
    Node() { memset(&harmonic, 0, sizeof(harmonic)); }

    void setBoundariesAsObstacles()  // epic_navigation_node_harmonic.cpp:283-306
    {
        for (unsigned y = 0; y < harmonic.m[0]; y++) {
            harmonic.u[y * harmonic.m[1] + 0] = EPIC_LOG_SPACE_OBSTACLE;
            harmonic.locked[y * harmonic.m[1] + 0] = 1;
            harmonic.u[y * harmonic.m[1] + harmonic.m[1] - 1] = EPIC_LOG_SPACE_OBSTACLE;
            harmonic.locked[y * harmonic.m[1] + harmonic.m[1] - 1] = 1;
        }
        for (unsigned x = 0; x < harmonic.m[1]; x++) {
            harmonic.u[x] = EPIC_LOG_SPACE_OBSTACLE;
            harmonic.locked[x] = 1;
            harmonic.u[(harmonic.m[0] - 1) * harmonic.m[1] + x] = EPIC_LOG_SPACE_OBSTACLE;
            harmonic.locked[(harmonic.m[0] - 1) * harmonic.m[1] + x] = 1;
        }
    }
    bool initAlg(unsigned w, unsigned h, float epsilon)
    {
        harmonic.n = 2;
        harmonic.m = new unsigned int[2];
        harmonic.m[0] = h;
        harmonic.m[1] = w;
        harmonic.u = new float[w * h];
        harmonic.locked = new unsigned int[w * h];
        for (unsigned i = 0; i < w * h; i++) { harmonic.u[i] = 0.0f; harmonic.locked[i] = 0; }
        setBoundariesAsObstacles();
        harmonic.epsilon = epsilon;
        harmonic.numIterationsToStaggerCheck = 100;
        int result = harmonic_initialize_dimension_size_gpu(&harmonic);
        result += harmonic_initialize_potential_values_gpu(&harmonic);
        result += harmonic_initialize_locked_gpu(&harmonic);
        result += harmonic_initialize_gpu(&harmonic, NUM_THREADS_GPU);
        gpu = result == EPIC_SUCCESS;
        return gpu;
    }
    bool setCells(std::vector<unsigned int> &v, std::vector<unsigned int> &types)
    {
        int result = harmonic_utilities_set_cells_2d_cpu(&harmonic, types.size(), &v[0], &types[0]);
        if (result != EPIC_SUCCESS) return false;
        if (gpu) {
            result = harmonic_utilities_set_cells_2d_gpu(&harmonic, NUM_THREADS_GPU, types.size(), &v[0], &types[0]);
            if (result != EPIC_SUCCESS) return false;
        }
        return true;
    }
    int update(unsigned num_steps)  // returns the code of the check step
    {
        int result = harmonic_update_and_check_gpu(&harmonic, NUM_THREADS_GPU);
        if (result == EPIC_SUCCESS) {
            for (unsigned i = 0; i < num_steps - 1; i++)
                if (harmonic_update_gpu(&harmonic, NUM_THREADS_GPU) != EPIC_SUCCESS) return -1;
        }
        return result;
    }
    int computePath(float x, float y, float step, float precision, unsigned max_length, unsigned &k, float *&raw_path)
    {
        if (gpu && harmonic_get_potential_values_gpu(&harmonic) != EPIC_SUCCESS) return -1;
        return harmonic_compute_path_2d_cpu(&harmonic, x, y, step, precision, max_length, k, raw_path);
    }
    void uninitAlg()
    {
        harmonic_uninitialize_dimension_size_gpu(&harmonic);
        harmonic_uninitialize_potential_values_gpu(&harmonic);
        harmonic_uninitialize_locked_gpu(&harmonic);
        harmonic_uninitialize_gpu(&harmonic);
        delete[] harmonic.m; delete[] harmonic.u; delete[] harmonic.locked;
        harmonic.m = nullptr; harmonic.u = nullptr; harmonic.locked = nullptr; harmonic.n = 0;
    }
};

static int run_node(const Input &in, FILE *out)
{
    Node node;
    if (!node.initAlg(in.cols, in.rows, in.epsilon)) { fprintf(stderr, "replay: initAlg failed (no GPU?)\n"); return 3; }
    // the /map callback (epic_navigation_node_harmonic.cpp:383-422): every interior cell becomes an obstacle or a free cell
    std::vector<unsigned int> v, types;
    for (unsigned y = 1; y + 1 < in.rows; y++)
        for (unsigned x = 1; x + 1 < in.cols; x++) {
            v.push_back(x); v.push_back(y);
            types.push_back(in.occupied[(size_t)y * in.cols + x] ? EPIC_CELL_TYPE_OBSTACLE : EPIC_CELL_TYPE_FREE);
        }
    if (!node.setCells(v, types)) return 4;
    // the add-goals service (:425-470): goal cells
    v.clear(); types.clear();
    for (size_t i = 0; i + 1 < in.goals.size(); i += 2) { v.push_back(in.goals[i]); v.push_back(in.goals[i + 1]); types.push_back(EPIC_CELL_TYPE_GOAL); }
    if (!node.setCells(v, types)) return 4;
    // the main loop (epic_navigation_node_main.cpp:72-81): update(steps) until the check step reports convergence
    const unsigned mMax = in.rows > in.cols ? in.rows : in.cols;
    int result = EPIC_SUCCESS;
    unsigned calls = 0;
    while (result != EPIC_SUCCESS_AND_CONVERGED || node.harmonic.currentIteration < mMax) {
        result = node.update(100);
        if (result != EPIC_SUCCESS && result != EPIC_SUCCESS_AND_CONVERGED) return 5;
        if (++calls > 100000) return 6;
    }
    fprintf(stderr, "replay node: converged after %u iterations, delta %.3e\n", node.harmonic.currentIteration, node.harmonic.delta);
    unsigned it = node.harmonic.currentIteration;
    fwrite(&it, 4, 1, out);
    for (size_t i = 0; i + 3 < in.starts.size(); i += 4) {
        unsigned int k = 0;
        float *raw_path = nullptr;
        int rc = node.computePath(in.starts[i], in.starts[i + 1], in.starts[i + 2], in.starts[i + 3], 1000000, k, raw_path);
        write_path(out, rc, k, raw_path);
        if (raw_path != nullptr) { delete[] raw_path; raw_path = nullptr; }   // as the callers do (:642-644, :670)
    }
    node.uninitAlg();
    return 0;
}

// ---- the nav_core plugin ---------------------------------------------------------------------------------------
static void set_goal(Harmonic &harmonic, unsigned x_goal, unsigned y_goal)   // epic_nav_core_plugin.cpp:341-366
{
    for (unsigned y = 1; y < harmonic.m[0] - 1; y++)
        for (unsigned x = 1; x < harmonic.m[1] - 1; x++)
            if (harmonic.u[y * harmonic.m[1] + x] == EPIC_LOG_SPACE_GOAL) {
                harmonic.u[y * harmonic.m[1] + x] = EPIC_LOG_SPACE_FREE;
                harmonic.locked[y * harmonic.m[1] + x] = 0;
            }
    harmonic.u[y_goal * harmonic.m[1] + x_goal] = EPIC_LOG_SPACE_GOAL;
    harmonic.locked[y_goal * harmonic.m[1] + x_goal] = 1;
}

static void plugin_grid(const Input &in, Harmonic &harmonic)   // epic_nav_core_plugin.cpp:140-187
{
    memset(&harmonic, 0, sizeof(harmonic));
    harmonic.n = 2;
    harmonic.m = new unsigned int[2];
    harmonic.m[0] = in.rows;
    harmonic.m[1] = in.cols;
    harmonic.u = new float[(size_t)in.rows * in.cols];
    harmonic.locked = new unsigned int[(size_t)in.rows * in.cols];
    for (unsigned y = 0; y < in.rows; y++)
        for (unsigned x = 0; x < in.cols; x++) {
            const bool border = y == 0 || x == 0 || y == in.rows - 1 || x == in.cols - 1;
            const bool obst = border || in.occupied[(size_t)y * in.cols + x];
            harmonic.u[y * in.cols + x] = obst ? EPIC_LOG_SPACE_OBSTACLE : EPIC_LOG_SPACE_FREE;
            harmonic.locked[y * in.cols + x] = obst ? 1 : 0;
        }
    harmonic.epsilon = in.epsilon;
    harmonic.delta = 0.0f;
    harmonic.numIterationsToStaggerCheck = 100;
}

static int run_plugin(const Input &in, FILE *out)
{
    if (in.goals.size() < 2 || in.starts.size() < 4) return 2;
    Harmonic harmonic, twin;
    plugin_grid(in, harmonic);
    plugin_grid(in, twin);
    int rc_all = 0;
    for (size_t g = 0; g + 1 < in.goals.size(); g += 2) {   // makePlan once per goal: the state carries over between calls
        set_goal(harmonic, in.goals[g], in.goals[g + 1]);
        set_goal(twin, in.goals[g], in.goals[g + 1]);
        int result = harmonic_complete_gpu(&harmonic, NUM_THREADS_GPU);
        if (result != EPIC_SUCCESS) { fprintf(stderr, "replay plugin: harmonic_complete_gpu returned %d\n", result); return 3; }
        if (harmonic.d_m || harmonic.d_u || harmonic.d_locked || harmonic.d_delta) return 7;   // complete_gpu leaves nothing behind
        if (harmonic_complete_cpu(&twin) != EPIC_SUCCESS) return 8;                              // the plugin's fallback path
        const size_t cells = (size_t)in.rows * in.cols;
        unsigned same = memcmp(harmonic.u, twin.u, cells * sizeof(float)) == 0 && harmonic.currentIteration == twin.currentIteration;
        unsigned it = harmonic.currentIteration;
        fwrite(&it, 4, 1, out);
        fwrite(&same, 4, 1, out);
        const size_t s = (g / 2) % (in.starts.size() / 4) * 4;
        float step_size = 0.05f, cd_precision = 0.5f;
        unsigned int max_length = harmonic.m[0] * harmonic.m[1] / step_size;
        unsigned int k = 0, k2 = 0;
        float *raw_plan = nullptr, *raw_twin = nullptr;
        result = harmonic_compute_path_2d_cpu(&harmonic, in.starts[s], in.starts[s + 1], step_size, cd_precision, max_length, k, raw_plan);
        int r2 = harmonic_compute_path_2d_cpu(&twin, in.starts[s], in.starts[s + 1], step_size, cd_precision, max_length, k2, raw_twin);
        if (result != r2 || k != k2 || (result == EPIC_SUCCESS && memcmp(raw_plan, raw_twin, 2 * (size_t)k * sizeof(float)) != 0)) rc_all = 9;
        write_path(out, result, k, raw_plan);
        if (raw_plan != nullptr) { delete[] raw_plan; raw_plan = nullptr; }
        if (raw_twin != nullptr) { delete[] raw_twin; raw_twin = nullptr; }
    }
    for (Harmonic *h : {&harmonic, &twin}) { delete[] h->m; delete[] h->u; delete[] h->locked; }
    return rc_all;
}

int main(int argc, char **argv)
{
    if (argc != 4) { fprintf(stderr, "usage: replay plugin|node <input.bin> <output.bin>\n"); return 2; }
    Input in;
    if (!read_input(argv[2], in)) { fprintf(stderr, "replay: cannot read %s\n", argv[2]); return 2; }
    FILE *out = fopen(argv[3], "wb");
    if (!out) return 2;
    const int rc = strcmp(argv[1], "node") == 0 ? run_node(in, out) : run_plugin(in, out);
    fclose(out);
    return rc;
}
