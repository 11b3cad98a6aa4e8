/*
 * epic_abi.h -- the drop-in C-ABI of the MI355X-native libepic.so.
 *
 * One header declares the whole boundary; the per-topic headers at the reference's include
 * paths (epic/harmonic/harmonic_gpu.h, ...) forward here, so sources written against the
 * reference (src/epic_nav_core_plugin.cpp, src/epic_navigation_node_harmonic.cpp) compile
 * unchanged.  Usable from C and C++: the reference's C++ reference parameters (`float &`)
 * are pointers at the ABI level, which is what ctypes / cgo / JNI bind
 * (libepic/python/epic/epic_harmonic.py:95-102).
 *
 * Every entry point cites the reference declaration it replaces (paths relative to the
 * reference repository root).  Return codes: epic/error_codes.h.  All functions print
 * "Error[<function>]: <text>\n" on stderr and return a code; none throws.
 */
#ifndef EPIC_ABI_H
#define EPIC_ABI_H

#ifdef __cplusplus
#define EPIC_OUT(T) T &
namespace epic {
extern "C" {
#else
#define EPIC_OUT(T) T *
#endif

/* ---- cell types and log-space seeds: libepic/include/epic/constants.h:34-43 ---- */
#define EPIC_FLT_MAX 1e+300
#define EPIC_FLT_MIN (-EPIC_FLT_MAX)
#define EPIC_CELL_TYPE_GOAL 0
#define EPIC_CELL_TYPE_OBSTACLE 1
#define EPIC_CELL_TYPE_FREE 2
#define EPIC_LOG_SPACE_GOAL 0.0
#define EPIC_LOG_SPACE_OBSTACLE -1e6
#define EPIC_LOG_SPACE_FREE -1e6

/* ---- return codes: libepic/include/epic/error_codes.h:31-46 ---- */
#define EPIC_SUCCESS 0
#define EPIC_SUCCESS_AND_CONVERGED 1
#define EPIC_ERROR_INVALID_DATA 2
#define EPIC_ERROR_INVALID_CUDA_PARAM 3
#define EPIC_ERROR_DEVICE_MALLOC 4
#define EPIC_ERROR_MEMCPY_TO_DEVICE 5
#define EPIC_ERROR_MEMCPY_TO_HOST 6
#define EPIC_ERROR_DEVICE_FREE 7
#define EPIC_ERROR_KERNEL_EXECUTION 8
#define EPIC_ERROR_DEVICE_SYNCHRONIZE 9
#define EPIC_ERROR_INVALID_LOCATION 10
#define EPIC_ERROR_INVALID_CELL_TYPE 11
#define EPIC_ERROR_INVALID_GRADIENT 12
#define EPIC_ERROR_INVALID_PATH 13

/*
 * The solver state carrier: libepic/include/epic/harmonic/harmonic.h:44-64.
 * 80 bytes on x86-64; offsets n 0 | m 8 | u 16 | locked 24 | epsilon 32 | delta 36 |
 * numIterationsToStaggerCheck 40 | currentIteration 44 | d_m 48 | d_u 56 | d_locked 64 |
 * d_delta 72 (static_asserts in epic_amd/csrc/abi_checks.cpp).
 *
 * m, u, locked are caller-owned host arrays (m[0] = rows/Y, m[1] = cols/X for n = 2; the last
 * dimension is contiguous).  u is float32 log-space potential, locked is one uint32 per cell.
 * Border cells are assumed locked.  The d_* fields are owned by this library: non-null
 * between initialize_* and uninitialize_*, opaque otherwise (here d_u points at the current
 * pitched ping-pong buffer and d_locked at a bit-packed mask).
 */
typedef struct Harmonic {
    unsigned int n;
    unsigned int *m;
    float *u;
    unsigned int *locked;
    float epsilon;
    float delta;
    unsigned int numIterationsToStaggerCheck;
    unsigned int currentIteration;
    unsigned int *d_m;
    float *d_u;
    unsigned int *d_locked;
    float *d_delta;
} Harmonic;

/* ---- CPU solver (red-black Gauss-Seidel), libepic/include/epic/harmonic/harmonic_cpu.h:40-56 ---- */
int harmonic_complete_cpu(Harmonic *harmonic);         /* harmonic_cpu.h:40 */
int harmonic_update_cpu(Harmonic *harmonic);           /* harmonic_cpu.h:47 */
int harmonic_update_and_check_cpu(Harmonic *harmonic); /* harmonic_cpu.h:56 */

/* ---- GPU solver, libepic/include/epic/harmonic/harmonic_gpu.h:39-86.
 * numThreads is kept for ABI compatibility: it must be a multiple of 32
 * (else EPIC_ERROR_INVALID_CUDA_PARAM, as harmonic_gpu.cu:240-244) and is otherwise a hint.
 * One "iteration" is what it is in the reference: one red-black half-sweep, the colour chosen by currentIteration
 * (harmonic_cpu.cpp:46-51); with the library's defaults every iteration, the iteration count and the converged field are
 * harmonic_complete_cpu's bit for bit.  EPIC_HIP_SCHEME=jacobi / epic_hip_set_scheme make an iteration a full Jacobi sweep
 * of every unlocked cell instead (include/epic_hip.h). ---- */
int harmonic_complete_gpu(Harmonic *harmonic, unsigned int numThreads);         /* harmonic_gpu.h:39 */
int harmonic_initialize_gpu(Harmonic *harmonic, unsigned int numThreads);       /* harmonic_gpu.h:47 */
int harmonic_execute_gpu(Harmonic *harmonic, unsigned int numThreads);          /* harmonic_gpu.h:55 */
int harmonic_uninitialize_gpu(Harmonic *harmonic);                              /* harmonic_gpu.h:62 */
int harmonic_update_gpu(Harmonic *harmonic, unsigned int numThreads);           /* harmonic_gpu.h:70 */
int harmonic_update_and_check_gpu(Harmonic *harmonic, unsigned int numThreads); /* harmonic_gpu.h:79 */
int harmonic_get_potential_values_gpu(Harmonic *harmonic);                      /* harmonic_gpu.h:86 */

/* ---- device-state lifecycle, libepic/include/epic/harmonic/harmonic_model_gpu.h:38-80 ---- */
int harmonic_initialize_dimension_size_gpu(Harmonic *harmonic);     /* harmonic_model_gpu.h:38 */
int harmonic_uninitialize_dimension_size_gpu(Harmonic *harmonic);   /* harmonic_model_gpu.h:45 */
int harmonic_initialize_potential_values_gpu(Harmonic *harmonic);   /* harmonic_model_gpu.h:52 */
int harmonic_uninitialize_potential_values_gpu(Harmonic *harmonic); /* harmonic_model_gpu.h:59 */
int harmonic_initialize_locked_gpu(Harmonic *harmonic);             /* harmonic_model_gpu.h:66 */
int harmonic_uninitialize_locked_gpu(Harmonic *harmonic);           /* harmonic_model_gpu.h:73 */
int harmonic_update_model_gpu(Harmonic *harmonic);                  /* harmonic_model_gpu.h:80 */

/* ---- sparse cell edits; v = k (x, y) pairs, types = k EPIC_CELL_TYPE_* ----
 * harmonic_utilities_cpu.h:41-42 and harmonic_utilities_gpu.h:42-43 */
int harmonic_utilities_set_cells_2d_cpu(Harmonic *harmonic, unsigned int k, unsigned int *v, unsigned int *types);
int harmonic_utilities_set_cells_2d_gpu(Harmonic *harmonic, unsigned int numThreads, unsigned int k,
                                        unsigned int *v, unsigned int *types);

/* ---- streamline extraction on the host field, libepic/include/epic/harmonic/harmonic_path_cpu.h:42-82.
 * (x, y) are "float pixel" coordinates (x along m[1]).  *path must be NULL on entry; on success it points at
 * 2*k floats allocated with new[] -- C++ callers release it with delete[] (src/epic_nav_core_plugin.cpp:303-305),
 * everyone else with harmonic_free_path_cpu. ---- */
int harmonic_compute_potential_2d_cpu(Harmonic *harmonic, float x, float y, EPIC_OUT(float) potential); /* :42 */
int harmonic_compute_gradient_2d_cpu(Harmonic *harmonic, float x, float y, float cdPrecision,
                                     EPIC_OUT(float) partialX, EPIC_OUT(float) partialY);               /* :56 */
int harmonic_compute_path_2d_cpu(Harmonic *harmonic, float x, float y, float stepSize, float cdPrecision,
                                 unsigned int maxLength, EPIC_OUT(unsigned int) k, EPIC_OUT(float *) path); /* :73 */
int harmonic_free_path_cpu(EPIC_OUT(float *) path);                                                      /* :82 */

/* ---- legacy linear-space SOR and its path follower (the paper's comparison baseline; CPU only, never accelerated):
 * libepic/include/epic/harmonic/harmonic_legacy_cpu.h:44-77, harmonic_legacy_path_cpu.h:43-90 ---- */
int harmonic_legacy_sor_2d_float_cpu(unsigned int w, unsigned int h, float epsilon, float omega, unsigned int *locked,
                                     float *u, EPIC_OUT(unsigned int) iter);
int harmonic_legacy_sor_2d_double_cpu(unsigned int w, unsigned int h, double epsilon, double omega,
                                      unsigned int *locked, double *u, EPIC_OUT(unsigned int) iter);
int harmonic_legacy_sor_2d_long_double_cpu(unsigned int w, unsigned int h, long double epsilon, long double omega,
                                           unsigned int *locked, long double *u, EPIC_OUT(unsigned int) iter);
int harmonic_legacy_compute_potential_2d_cpu(unsigned int w, unsigned int h, unsigned int *locked, double *u, double x,
                                             double y, EPIC_OUT(double) potential);
int harmonic_legacy_compute_gradient_2d_cpu(unsigned int w, unsigned int h, unsigned int *locked, double *u, double x,
                                            double y, double cdPrecision, EPIC_OUT(double) partialX,
                                            EPIC_OUT(double) partialY);
int harmonic_legacy_compute_path_2d_cpu(unsigned int w, unsigned int h, unsigned int *locked, double *u, double x,
                                        double y, double stepSize, double cdPrecision, unsigned int maxLength,
                                        int flipped, EPIC_OUT(unsigned int) k, EPIC_OUT(double *) path);
int harmonic_legacy_free_path_cpu(EPIC_OUT(double *) path);

#ifdef __cplusplus
} /* extern "C" */
} /* namespace epic */
#endif

#endif /* EPIC_ABI_H */
