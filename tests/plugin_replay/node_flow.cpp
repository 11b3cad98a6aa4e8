// node_flow.cpp -- the navigation node's update loop against libepic.so, timed (bench.py's `node_flow` leg, tests/test_gpu_node_flow.py).
//
// The reference's second caller drives the solver through the fine-grained API: every tick is one
// harmonic_update_and_check_gpu and, if that returned EPIC_SUCCESS, num_steps - 1 calls of harmonic_update_gpu
// (src/epic_navigation_node_harmonic.cpp:165-189; num_steps = 50 at 10 Hz by default, 100 at 30 Hz in
// launch/epic_navigation_node_umass.launch:11-12).  A Python loop of ctypes calls would time the interpreter (~1 us per call against
// ~1.5 us per iteration on the reference's maps), so the loop is this translation unit: it includes only the reference's header
// paths, calls only exported harmonic_* functions and is linked with -lepic, like replay.cpp.  Test / bench infrastructure: not part
// of libepic.so.
//
//   g++ -std=c++11 -O2 -shared -fPIC -I include tests/plugin_replay/node_flow.cpp -L epic_amd/lib -lepic -o tests/plugin_replay/libnodeflow.so
#include <chrono>

#include <epic/constants.h>
#include <epic/error_codes.h>
#include <epic/harmonic/harmonic.h>
#include <epic/harmonic/harmonic_gpu.h>

using namespace epic;

extern "C" {

// `iterations` iterations in ticks of num_steps (the last tick is cut short when num_steps does not divide iterations).  Like the node,
// a tick whose check reports convergence skips its plain updates -- unless keep_going, which runs them regardless (a fixed iteration
// count, comparable with harmonic_execute_gpu's loop).  Returns the first failing call's code (0: none); *seconds: wall time of the loop
// plus the final harmonic_get_potential_values_gpu (the node's srvComputePath; it is also what makes the device finish);
// *iterations_done: harmonic->currentIteration's advance; *converged_ticks: ticks whose check returned EPIC_SUCCESS_AND_CONVERGED.
int node_flow_run(Harmonic *harmonic, unsigned int iterations, unsigned int num_steps, unsigned int num_gpu_threads, int keep_going,
                  double *seconds, unsigned int *iterations_done, unsigned int *converged_ticks)
{
    if (harmonic == nullptr || num_steps == 0) return EPIC_ERROR_INVALID_DATA;
    const unsigned int start = harmonic->currentIteration;
    unsigned int conv = 0;
    int rc = EPIC_SUCCESS;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned int done = 0; done < iterations && rc == EPIC_SUCCESS;) {
        const unsigned int steps = iterations - done < num_steps ? iterations - done : num_steps;
        // EpicNavigationNodeHarmonic::update(num_steps), the GPU branch
        int result = harmonic_update_and_check_gpu(harmonic, num_gpu_threads);
        unsigned int ran = 1;
        if (result == EPIC_SUCCESS || (result == EPIC_SUCCESS_AND_CONVERGED && keep_going)) {
            ran = steps;
            if (result == EPIC_SUCCESS_AND_CONVERGED) conv++;
            for (unsigned int i = 0; i < steps - 1; i++) {
                if (harmonic_update_gpu(harmonic, num_gpu_threads) != EPIC_SUCCESS) {
                    rc = EPIC_ERROR_KERNEL_EXECUTION;
                    break;
                }
            }
        } else if (result == EPIC_SUCCESS_AND_CONVERGED) {
            conv++;
        } else {
            rc = result;
        }
        done += ran;
    }
    if (rc == EPIC_SUCCESS) rc = harmonic_get_potential_values_gpu(harmonic);
    if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (iterations_done) *iterations_done = harmonic->currentIteration - start;
    if (converged_ticks) *converged_ticks = conv;
    return rc;
}

// harmonic_execute_gpu under the same clock (it ends with its own harmonic_get_potential_values_gpu and harmonic_uninitialize_gpu)
int node_flow_execute(Harmonic *harmonic, unsigned int num_gpu_threads, double *seconds)
{
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = harmonic_execute_gpu(harmonic, num_gpu_threads);
    if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}

}  // extern "C"
