// ubench_grid_barrier.hip -- what would a PERSISTENT tile kernel pay per exchange of ghost rings?  (round 6)
//
// The small-grid path (kernels_tile2d.hip) advances K iterations per launch on LDS tiles with K ghost rings and pays ~2.7 us per launch
// besides the iterations themselves (profiles/r04_experiments.txt item 1d).  The one design not yet measured keeps every tile's workgroup
// resident for the whole relaxation and replaces the launch boundary by a synchronisation inside the kernel.  This program measures that
// synchronisation alone, on 256 workgroups of 256 threads (one per CU, as the reference's maps give), in the two forms such a kernel could use:
//   grid      every workgroup arrives at ONE counter (atomicAdd, agent scope) and spins until all have arrived: a grid-wide barrier;
//   neighbour every workgroup publishes its step in a flag of its own (release store) and spins on the flags of its four neighbours in a
//             16 x 16 arrangement (acquire loads): the point-to-point form -- all a tile needs is its neighbours' rings.
// Between two synchronisations every workgroup writes and reads 4 KiB of "ring" data through global memory, as the tiles would.
// For comparison: back-to-back launches of an empty kernel of the same shape, eagerly and from a captured graph.
//
// Every spin is BOUNDED (1 << 22 polls, then the workgroup raises an error flag and every workgroup leaves): nothing here can hang the GPU.
//
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_grid_barrier.hip -o tools/ubench_grid_barrier && tools/ubench_grid_barrier
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <chrono>
#include <vector>

#define CHECK(x)                                                                       \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
            exit(1);                                                                   \
        }                                                                              \
    } while (0)

constexpr int kBlocks = 256, kThreads = 256, kSide = 16, kRingWords = 1024;   // 4 KiB of ring data per workgroup and step
constexpr unsigned kMaxPolls = 1u << 22;

struct Args {
    unsigned *counter;     // grid form
    unsigned *flags;       // neighbour form: one word per workgroup, 64 bytes apart
    unsigned *error;
    float *rings;          // kBlocks x 2 x kRingWords (double-buffered by step parity)
    float *sink;
    int steps;
    int mode;              // 0 grid, 1 neighbour, 2 no synchronisation (the work alone)
    int lean;              // 1: rings and flags live in UNCACHED device memory (hipDeviceMallocUncached): no L2 write-back / invalidate around the
                           //    exchange -- the stores are waited for (s_waitcnt) and the flag is a relaxed store; the consumer's loads bypass the caches
};

__global__ __launch_bounds__(kThreads) void persistent_kernel(Args a)
{
    const int b = blockIdx.x, t = threadIdx.x;
    const int bi = b / kSide, bj = b % kSide;
    const int nb[4] = {((bi + kSide - 1) % kSide) * kSide + bj, ((bi + 1) % kSide) * kSide + bj, bi * kSide + (bj + kSide - 1) % kSide,
                       bi * kSide + (bj + 1) % kSide};
    __shared__ int bail;
    if (t == 0) bail = 0;
    __syncthreads();
    float acc = 0.0f;
    for (int s = 1; s <= a.steps; s++) {
        // publish this step's ring (what a tile's owned boundary cells would be)
        float *mine = a.rings + ((size_t)b * 2 + (s & 1)) * kRingWords;
        if (a.lean) {
            for (int i = t; i < kRingWords; i += kThreads) __builtin_nontemporal_store(acc + (float)(s + i), mine + i);
            __builtin_amdgcn_s_waitcnt(0);   // this thread's stores have left for the fabric
        } else {
            for (int i = t; i < kRingWords; i += kThreads) mine[i] = acc + (float)(s + i);
            __threadfence();
        }
        __syncthreads();
        if (t == 0 && a.mode != 2) {
            unsigned polls = 0;
            const int order_ld = a.lean ? __ATOMIC_RELAXED : __ATOMIC_ACQUIRE;
            if (a.mode == 0) {
                __hip_atomic_fetch_add(a.counter, 1u, a.lean ? __ATOMIC_RELAXED : __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned want = (unsigned)s * kBlocks;
                while (__hip_atomic_load(a.counter, order_ld, __HIP_MEMORY_SCOPE_AGENT) < want) {
                    if (++polls > kMaxPolls || __hip_atomic_load(a.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { bail = 1; break; }
                }
            } else {
                __hip_atomic_store(a.flags + b * 16, (unsigned)s, a.lean ? __ATOMIC_RELAXED : __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                for (int k = 0; k < 4 && !bail; k++)
                    while (__hip_atomic_load(a.flags + nb[k] * 16, order_ld, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)s) {
                        if (++polls > kMaxPolls || __hip_atomic_load(a.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { bail = 1; break; }
                    }
            }
            if (bail) __hip_atomic_store(a.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (bail) break;
        // read the neighbours' rings of this step
        for (int k = 0; k < 4; k++) {
            const float *theirs = a.rings + ((size_t)nb[k] * 2 + (s & 1)) * kRingWords;
            for (int i = t; i < kRingWords / 4; i += kThreads) acc += __builtin_nontemporal_load(theirs + i * 4);
        }
    }
    if (acc == 12345.678f) a.sink[b * kThreads + t] = acc;
}

__global__ __launch_bounds__(kThreads) void step_kernel(Args a, int s)
{
    const int b = blockIdx.x, t = threadIdx.x;
    const int bi = b / kSide, bj = b % kSide;
    const int nb[4] = {((bi + kSide - 1) % kSide) * kSide + bj, ((bi + 1) % kSide) * kSide + bj, bi * kSide + (bj + kSide - 1) % kSide,
                       bi * kSide + (bj + 1) % kSide};
    float acc = 0.0f;
    for (int k = 0; k < 4; k++) {
        const float *theirs = a.rings + ((size_t)nb[k] * 2 + ((s - 1) & 1)) * kRingWords;
        for (int i = t; i < kRingWords / 4; i += kThreads) acc += theirs[i * 4];
    }
    float *mine = a.rings + ((size_t)b * 2 + (s & 1)) * kRingWords;
    for (int i = t; i < kRingWords; i += kThreads) mine[i] = acc + (float)(s + i);
}

int main()
{
    Args a;
    CHECK(hipMalloc((void **)&a.counter, 256));
    CHECK(hipMalloc((void **)&a.flags, kBlocks * 64));
    CHECK(hipMalloc((void **)&a.error, 256));
    CHECK(hipMalloc((void **)&a.rings, (size_t)kBlocks * 2 * kRingWords * sizeof(float)));
    CHECK(hipMalloc((void **)&a.sink, (size_t)kBlocks * kThreads * sizeof(float)));
    hipStream_t st;
    CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    int cus = 0;
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    printf("%d CUs; %d workgroups x %d threads, %d B of ring data per workgroup and step\n", cus, kBlocks, kThreads, kRingWords * 4);
    const int steps = 4000;
    a.steps = steps;
    a.lean = 0;
    // the lean variant's memory: uncached (fine-grained) device memory
    unsigned *u_counter = nullptr, *u_flags = nullptr;
    float *u_rings = nullptr;
    const bool have_uncached = hipExtMallocWithFlags((void **)&u_counter, 256, hipDeviceMallocUncached) == hipSuccess &&
                               hipExtMallocWithFlags((void **)&u_flags, kBlocks * 64, hipDeviceMallocUncached) == hipSuccess &&
                               hipExtMallocWithFlags((void **)&u_rings, (size_t)kBlocks * 2 * kRingWords * sizeof(float), hipDeviceMallocUncached) == hipSuccess;
    if (!have_uncached) { (void)hipGetLastError(); printf("(no uncached device memory here: the lean variant is skipped)\n"); }
    const char *names[3] = {"persistent kernel, grid-wide barrier (one counter)", "persistent kernel, four neighbour flags", "persistent kernel, no synchronisation (work only)"};
    for (int run = 0; run < (have_uncached ? 10 : 5); run++) {
        const int mode = (const int[]){2, 0, 1, 0, 1}[run % 5];
        a.mode = mode;
        a.lean = run >= 5;
        if (run == 5) { a.counter = u_counter; a.flags = u_flags; a.rings = u_rings; }
        CHECK(hipMemsetAsync(a.counter, 0, 256, st));
        CHECK(hipMemsetAsync(a.flags, 0, kBlocks * 64, st));
        CHECK(hipMemsetAsync(a.error, 0, 256, st));
        CHECK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(persistent_kernel, dim3(kBlocks), dim3(kThreads), 0, st, a);
        CHECK(hipGetLastError());
        CHECK(hipEventRecord(e1, st));
        CHECK(hipEventSynchronize(e1));
        float ms = 0.0f;
        unsigned err = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        CHECK(hipMemcpy(&err, a.error, 4, hipMemcpyDeviceToHost));
        printf("%-58s %s %8.3f us per step%s\n", names[mode], a.lean ? "[uncached memory, relaxed flags]" : "[release / acquire]            ", ms * 1e3 / steps,
               err ? "   (BAILED OUT: a spin ran into its bound)" : "");
    }
    // the launch boundary it would replace: one kernel per step, eagerly and from a captured graph
    for (int graph = 0; graph < 2; graph++) {
        const int n = 1000;
        hipGraphExec_t exec = nullptr;
        if (graph) {
            hipGraph_t g = nullptr;
            CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            for (int s = 1; s <= 100; s++) hipLaunchKernelGGL(step_kernel, dim3(kBlocks), dim3(kThreads), 0, st, a, s);
            CHECK(hipStreamEndCapture(st, &g));
            CHECK(hipGraphInstantiate(&exec, g, nullptr, nullptr, 0));
            CHECK(hipGraphDestroy(g));
            CHECK(hipGraphLaunch(exec, st));
            CHECK(hipStreamSynchronize(st));
        }
        for (int rep = 0; rep < 2; rep++) {
            CHECK(hipEventRecord(e0, st));
            if (graph)
                for (int k = 0; k < n / 100; k++) CHECK(hipGraphLaunch(exec, st));
            else
                for (int s = 1; s <= n; s++) hipLaunchKernelGGL(step_kernel, dim3(kBlocks), dim3(kThreads), 0, st, a, s);
            CHECK(hipEventRecord(e1, st));
            CHECK(hipEventSynchronize(e1));
            float ms = 0.0f;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("%-58s %8.3f us per step\n", graph ? "one kernel per step, replayed from a graph of 100" : "one kernel per step, launched eagerly", ms * 1e3 / n);
        }
        if (exec) CHECK(hipGraphExecDestroy(exec));
    }
    return 0;
}
