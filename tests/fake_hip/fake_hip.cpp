// fake_hip.cpp -- a malloc-backed stand-in for the HIP runtime and for the kernel launchers, TEST INFRASTRUCTURE ONLY.
//
// tests/test_host_driver_faults.py compiles epic_amd/csrc/driver_*.hip (the library's HOST driver, unchanged) against
// tests/fake_hip/hip/hip_runtime.h and links it with this file under -fsanitize=address,undefined (and, for the issuing
// threads of the multi-device mode, -fsanitize=thread).  What the fake provides:
//   * memory: hipMalloc / hipHostMalloc are malloc + a registry; copies are memcpy with the device side checked against the
//     registry (and by ASan against the allocation); streams execute at once; events and graphs are registered objects;
//   * fault injection: fake_hip_fail_at(n) makes the n-th fallible call fail -- allocations, copies, memsets, stream / event
//     creation, synchronisations, kernel launches --, once;
//   * kernels: the launchers of epic_amd/csrc/kernels.h as no-ops that honour the injection (a check iteration then reports
//     max |du| = 0, so the "until converged" loops end after max(m) iterations);
//   * accounting: live objects per kind and a misuse counter (free / destroy of something not live, copies outside an
//     allocation), which the driver program asserts on after every unwind;
//   * DEVICE AFFINITY (round 5): every device allocation, stream and event belongs to the device that was current when it was
//     made, peer access is enabled per ordered pair of devices, and the rules the real runtime enforces on a multi-GPU node are
//     checked on every call -- a kernel goes into a stream of the CURRENT device and may only touch memory of that device or of
//     a peer it has enabled; an event is recorded into a stream of ITS device; hipMemcpyPeerAsync names the devices that really
//     own the two buffers; elapsed time is taken between events of one device.  No session of this project has had two GPUs:
//     this is where the multi-device mode's device bookkeeping is held to account (driver.cpp, mode "devices").
// Stream capture is refused (hipErrorNotSupported), which sends the library down its eager fallback.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <initializer_list>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <utility>

#include "../../epic_amd/csrc/kernels.h"

namespace {
struct State {
    std::mutex mu;
    std::map<char *, size_t> dev, host;
    std::map<char *, int> dev_owner;                 // device allocation -> its device
    std::map<void *, int> stream_dev, event_dev;     // stream / event -> its device
    std::set<std::pair<int, int>> peers;             // (from, to): kernels and copies of `from` may address memory of `to`
    bool peer_capable = true;                        // what hipDeviceCanAccessPeer says about two different devices
    long affinity = 0;                               // violations of the device rules
    std::string first_affinity;
    std::set<void *> streams, events, graphs;
    long calls = 0, fail_at = 0, misuse = 0;
    bool failed = false;
    const char *failed_name = "";
    int devices = 1;
};
State g;
thread_local int t_dev = 0;
thread_local hipError_t t_last = hipSuccess;

hipError_t fail(hipError_t e)
{
    t_last = e;
    return e;
}
// one fallible call: true = this is the one that fails
bool trip(const char *name)
{
    std::lock_guard<std::mutex> lk(g.mu);
    g.calls++;
    if (g.fail_at > 0 && !g.failed && g.calls == g.fail_at) {
        g.failed = true;
        g.failed_name = name;
        return true;
    }
    return false;
}
// [p, p + bytes) inside a live allocation of `m`
bool inside(std::map<char *, size_t> &m, const void *p, size_t bytes)
{
    auto it = m.upper_bound((char *)const_cast<void *>(p));
    if (it == m.begin()) return false;
    --it;
    return (char *)p >= it->first && (char *)p + bytes <= it->first + it->second;
}
void check_device(const void *p, size_t bytes)
{
    std::lock_guard<std::mutex> lk(g.mu);
    if (!inside(g.dev, p, bytes)) g.misuse++;
}
// ---- device affinity ---------------------------------------------------------------------------------------------------
void affinity_fail(const std::string &what)   // (g.mu held)
{
    if (g.affinity++ == 0) g.first_affinity = what;
}
int owner_of(const void *p)   // (g.mu held) device of the allocation that holds p; -1: not device memory (host, pinned, null)
{
    auto it = g.dev.upper_bound((char *)const_cast<void *>(p));
    if (it == g.dev.begin()) return -1;
    --it;
    if ((char *)p >= it->first + it->second) return -1;
    return g.dev_owner[it->first];
}
int device_of_stream(hipStream_t s)   // (g.mu held) the null stream is the current device's
{
    if (s == nullptr) return t_dev;
    auto it = g.stream_dev.find(s);
    return it == g.stream_dev.end() ? t_dev : it->second;
}
bool reachable(int from, const void *p)   // (g.mu held)
{
    const int o = owner_of(p);
    return o < 0 || o == from || g.peers.count({from, o}) != 0;
}
// a kernel (or an asynchronous memset) on stream s touching the listed buffers
void check_launch(const char *name, hipStream_t s, std::initializer_list<const void *> ptrs, bool must_be_current = true)
{
    std::lock_guard<std::mutex> lk(g.mu);
    const int sd = device_of_stream(s);
    if (must_be_current && sd != t_dev)
        affinity_fail(std::string(name) + ": launched into a stream of device " + std::to_string(sd) + " while device " + std::to_string(t_dev) + " is current");
    for (const void *p : ptrs)
        if (p && !reachable(sd, p))
            affinity_fail(std::string(name) + ": stream of device " + std::to_string(sd) + " touches memory of device " + std::to_string(owner_of(p)) + " without peer access");
}
void check_side(const void *dst, const void *src, size_t bytes, hipMemcpyKind kind)
{
    if (bytes == 0) return;
    if (kind == hipMemcpyHostToDevice || kind == hipMemcpyDeviceToDevice) check_device(dst, bytes);
    if (kind == hipMemcpyDeviceToHost || kind == hipMemcpyDeviceToDevice) check_device(src, bytes);
}
template <class T> hipError_t make(T **out, std::set<void *> &reg, const char *name)
{
    if (trip(name)) return fail(hipErrorOutOfMemory);
    *out = (T *)malloc(16);
    std::lock_guard<std::mutex> lk(g.mu);
    reg.insert(*out);
    return hipSuccess;
}
hipError_t unmake(void *p, std::set<void *> &reg)
{
    std::lock_guard<std::mutex> lk(g.mu);
    if (!reg.erase(p)) { g.misuse++; return fail(hipErrorInvalidValue); }
    free(p);
    return hipSuccess;
}
}  // namespace

extern "C" {
void fake_hip_fail_at(long n)
{
    std::lock_guard<std::mutex> lk(g.mu);
    g.calls = 0;
    g.fail_at = n;
    g.failed = false;
    g.failed_name = "";
}
long fake_hip_calls(void) { std::lock_guard<std::mutex> lk(g.mu); return g.calls; }
int fake_hip_failed(void) { std::lock_guard<std::mutex> lk(g.mu); return g.failed; }
const char *fake_hip_failed_call(void) { std::lock_guard<std::mutex> lk(g.mu); return g.failed_name; }
long fake_hip_live(int kind)
{
    std::lock_guard<std::mutex> lk(g.mu);
    return kind == 0 ? (long)g.dev.size() : kind == 1 ? (long)g.host.size() : kind == 2 ? (long)g.streams.size()
           : kind == 3 ? (long)g.events.size() : (long)g.graphs.size();
}
long fake_hip_misuse(void) { std::lock_guard<std::mutex> lk(g.mu); return g.misuse; }
void fake_hip_set_devices(int n) { std::lock_guard<std::mutex> lk(g.mu); g.devices = n; g.peers.clear(); }
void fake_hip_set_peer_capable(int on) { std::lock_guard<std::mutex> lk(g.mu); g.peer_capable = on != 0; g.peers.clear(); }
long fake_hip_affinity(void) { std::lock_guard<std::mutex> lk(g.mu); return g.affinity; }
const char *fake_hip_first_affinity(void) { std::lock_guard<std::mutex> lk(g.mu); return g.first_affinity.c_str(); }

hipError_t hipGetLastError(void) { const hipError_t e = t_last; t_last = hipSuccess; return e; }
hipError_t hipGetDeviceCount(int *n) { std::lock_guard<std::mutex> lk(g.mu); *n = g.devices; return hipSuccess; }
hipError_t hipGetDevice(int *dev) { *dev = t_dev; return hipSuccess; }
hipError_t hipSetDevice(int dev)
{
    { std::lock_guard<std::mutex> lk(g.mu); if (dev < 0 || dev >= g.devices) return fail(hipErrorInvalidDevice); }
    t_dev = dev;
    return hipSuccess;
}
hipError_t hipDeviceGetAttribute(int *value, hipDeviceAttribute_t, int) { *value = 256; return hipSuccess; }
hipError_t hipDeviceCanAccessPeer(int *can, int dev, int peer)
{
    std::lock_guard<std::mutex> lk(g.mu);
    if (dev < 0 || dev >= g.devices || peer < 0 || peer >= g.devices) return fail(hipErrorInvalidDevice);
    *can = dev != peer && g.peer_capable;
    return hipSuccess;
}
hipError_t hipDeviceEnablePeerAccess(int peer, unsigned)   // the CURRENT device may address `peer`'s memory from now on
{
    std::lock_guard<std::mutex> lk(g.mu);
    if (peer < 0 || peer >= g.devices || peer == t_dev || !g.peer_capable) return fail(hipErrorInvalidDevice);
    if (!g.peers.insert({t_dev, peer}).second) return fail(hipErrorPeerAccessAlreadyEnabled);
    return hipSuccess;
}
hipError_t hipExtGetLinkTypeAndHopCount(int, int, uint32_t *linktype, uint32_t *hops) { *linktype = 5; *hops = 1; return hipSuccess; }

hipError_t hipMalloc(void **p, size_t bytes)
{
    *p = nullptr;
    if (trip("hipMalloc")) return fail(hipErrorOutOfMemory);
    char *q = (char *)malloc(bytes ? bytes : 1);
    if (!q) return fail(hipErrorOutOfMemory);
    memset(q, 0xA5, bytes);   // device memory is not zeroed
    std::lock_guard<std::mutex> lk(g.mu);
    g.dev[q] = bytes;
    g.dev_owner[q] = t_dev;
    *p = q;
    return hipSuccess;
}
hipError_t hipFree(void *p)
{
    if (!p) return hipSuccess;
    std::lock_guard<std::mutex> lk(g.mu);
    g.dev_owner.erase((char *)p);
    if (!g.dev.erase((char *)p)) { g.misuse++; return fail(hipErrorInvalidValue); }
    free(p);
    return hipSuccess;
}
hipError_t hipHostMalloc(void **p, size_t bytes, unsigned)
{
    *p = nullptr;
    if (trip("hipHostMalloc")) return fail(hipErrorOutOfMemory);
    char *q = (char *)malloc(bytes ? bytes : 1);
    if (!q) return fail(hipErrorOutOfMemory);
    memset(q, 0x5A, bytes);
    std::lock_guard<std::mutex> lk(g.mu);
    g.host[q] = bytes;
    *p = q;
    return hipSuccess;
}
hipError_t hipHostFree(void *p)
{
    if (!p) return hipSuccess;
    std::lock_guard<std::mutex> lk(g.mu);
    if (!g.host.erase((char *)p)) { g.misuse++; return fail(hipErrorInvalidValue); }
    free(p);
    return hipSuccess;
}
hipError_t hipMemcpy(void *dst, const void *src, size_t bytes, hipMemcpyKind kind)
{
    if (trip("hipMemcpy")) return fail(hipErrorUnknown);
    check_side(dst, src, bytes, kind);
    memmove(dst, src, bytes);
    return hipSuccess;
}
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t bytes, hipMemcpyKind kind, hipStream_t s)
{
    if (trip("hipMemcpyAsync")) return fail(hipErrorUnknown);
    check_side(dst, src, bytes, kind);
    // (a copy may be issued into a stream of another device than the current one; what it touches must be that stream's device's,
    //  or a peer's it has enabled: copies between two devices go through hipMemcpyPeerAsync in this library)
    check_launch("hipMemcpyAsync", s, {kind == hipMemcpyHostToDevice || kind == hipMemcpyDeviceToDevice ? dst : nullptr,
                                       kind == hipMemcpyDeviceToHost || kind == hipMemcpyDeviceToDevice ? src : nullptr}, false);
    memmove(dst, src, bytes);
    return hipSuccess;
}
hipError_t hipMemcpy2D(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height, hipMemcpyKind kind)
{
    if (trip("hipMemcpy2D")) return fail(hipErrorUnknown);
    if (width > dpitch || width > spitch) { std::lock_guard<std::mutex> lk(g.mu); g.misuse++; return fail(hipErrorInvalidValue); }
    for (size_t r = 0; r < height; r++) {
        check_side((char *)dst + r * dpitch, (const char *)src + r * spitch, width, kind);
        memmove((char *)dst + r * dpitch, (const char *)src + r * spitch, width);
    }
    return hipSuccess;
}
hipError_t hipMemcpyPeerAsync(void *dst, int ddev, const void *src, int sdev, size_t bytes, hipStream_t)
{
    if (trip("hipMemcpyPeerAsync")) return fail(hipErrorUnknown);
    check_side(dst, src, bytes, hipMemcpyDeviceToDevice);
    {
        std::lock_guard<std::mutex> lk(g.mu);
        if (bytes && (owner_of(dst) != ddev || owner_of(src) != sdev))
            affinity_fail("hipMemcpyPeerAsync: named devices " + std::to_string(ddev) + " <- " + std::to_string(sdev) + " but the buffers belong to " +
                          std::to_string(owner_of(dst)) + " <- " + std::to_string(owner_of(src)));
    }
    memmove(dst, src, bytes);
    return hipSuccess;
}
hipError_t hipMemset(void *p, int v, size_t bytes)
{
    if (trip("hipMemset")) return fail(hipErrorUnknown);
    check_device(p, bytes);
    check_launch("hipMemset", nullptr, {p});
    memset(p, v, bytes);
    return hipSuccess;
}
hipError_t hipMemsetAsync(void *p, int v, size_t bytes, hipStream_t s)
{
    if (trip("hipMemsetAsync")) return fail(hipErrorUnknown);
    check_device(p, bytes);
    check_launch("hipMemsetAsync", s, {p}, false);
    memset(p, v, bytes);
    return hipSuccess;
}
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned)
{
    const hipError_t e = make(s, g.streams, "hipStreamCreateWithFlags");
    if (e == hipSuccess) { std::lock_guard<std::mutex> lk(g.mu); g.stream_dev[*s] = t_dev; }
    return e;
}
hipError_t hipStreamDestroy(hipStream_t s) { { std::lock_guard<std::mutex> lk(g.mu); g.stream_dev.erase(s); } return unmake(s, g.streams); }
hipError_t hipStreamSynchronize(hipStream_t) { return trip("hipStreamSynchronize") ? fail(hipErrorUnknown) : hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return trip("hipStreamWaitEvent") ? fail(hipErrorUnknown) : hipSuccess; }
hipError_t hipStreamIsCapturing(hipStream_t, hipStreamCaptureStatus *status) { *status = hipStreamCaptureStatusNone; return hipSuccess; }
hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return fail(hipErrorNotSupported); }
hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t *gr) { *gr = nullptr; return fail(hipErrorNotSupported); }
hipError_t hipGraphInstantiate(hipGraphExec_t *e, hipGraph_t, hipGraphNode_t *, char *, size_t) { *e = nullptr; return fail(hipErrorNotSupported); }
hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }
hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return fail(hipErrorNotSupported); }
static hipError_t make_event(hipEvent_t *e, const char *name)
{
    const hipError_t r = make(e, g.events, name);
    if (r == hipSuccess) { std::lock_guard<std::mutex> lk(g.mu); g.event_dev[*e] = t_dev; }
    return r;
}
hipError_t hipEventCreate(hipEvent_t *e) { return make_event(e, "hipEventCreate"); }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return make_event(e, "hipEventCreateWithFlags"); }
hipError_t hipEventDestroy(hipEvent_t e) { { std::lock_guard<std::mutex> lk(g.mu); g.event_dev.erase(e); } return unmake(e, g.events); }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s)
{
    if (trip("hipEventRecord")) return fail(hipErrorUnknown);
    std::lock_guard<std::mutex> lk(g.mu);
    auto it = g.event_dev.find(e);
    if (it != g.event_dev.end() && it->second != device_of_stream(s))   // the real runtime: hipErrorInvalidHandle
        affinity_fail("hipEventRecord: an event of device " + std::to_string(it->second) + " recorded into a stream of device " + std::to_string(device_of_stream(s)));
    return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t) { return trip("hipEventSynchronize") ? fail(hipErrorUnknown) : hipSuccess; }
hipError_t hipEventQuery(hipEvent_t) { return trip("hipEventQuery") ? fail(hipErrorUnknown) : hipSuccess; }   // (streams execute at once: never hipErrorNotReady)
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b)
{
    *ms = 1.0f;
    std::lock_guard<std::mutex> lk(g.mu);
    if (g.event_dev.count(a) && g.event_dev.count(b) && g.event_dev[a] != g.event_dev[b])
        affinity_fail("hipEventElapsedTime: events of devices " + std::to_string(g.event_dev[a]) + " and " + std::to_string(g.event_dev[b]));
    return hipSuccess;
}
}  // extern "C"

// ---- the kernel launchers of epic_amd/csrc/kernels.h: no-ops that can be made to fail --------------------------------
namespace epic_hip {
#define FAKE_LAUNCH(name) do { if (trip(name)) return fail(hipErrorUnknown); } while (0)
int resident_blocks_of(const void *) { return 2048; }
hipError_t launch_sweep_2d(const float *in, float *out, const uint32_t *maskw, int rows, int pitch, int, int, int, int, int, unsigned *delta, hipStream_t s,
                           const Activity *act, int, int)
{
    FAKE_LAUNCH("launch_sweep_2d");
    check_launch("launch_sweep_2d", s, {in, out, maskw, delta, act ? act->list_out : nullptr, act ? act->count_in : nullptr});
    check_device(in, (size_t)rows * pitch * 4);
    check_device(out, (size_t)rows * pitch * 4);
    return hipSuccess;
}
hipError_t launch_wake_tile_range(const Activity *next, size_t, int, int, hipStream_t s)
{
    FAKE_LAUNCH("launch_wake_tile_range");
    check_launch("launch_wake_tile_range", s, {next ? next->list_out : nullptr, next ? next->count_out : nullptr});
    return hipSuccess;
}
hipError_t launch_rb_fused_2d(const float *in, float *out, const uint32_t *maskw, int, int, int, int, int, hipStream_t s, const uint32_t *maskf, const Activity *act,
                              unsigned *delta, int, int)
{
    FAKE_LAUNCH("launch_rb_fused_2d");
    check_launch("launch_rb_fused_2d", s, {in, out, maskw, maskf, delta, act ? act->list_out : nullptr});
    return hipSuccess;
}
hipError_t launch_jacobi_fused_2d(const float *in, float *out, const uint32_t *maskw, int, int, int, int, hipStream_t s, int, const uint32_t *maskf, const Activity *act,
                                  unsigned *delta, int, int)
{
    FAKE_LAUNCH("launch_jacobi_fused_2d");
    check_launch("launch_jacobi_fused_2d", s, {in, out, maskw, maskf, delta, act ? act->list_out : nullptr});
    return hipSuccess;
}
hipError_t launch_fuse_masks_2d(const uint32_t *maskw, int, int, uint32_t *maskf, hipStream_t s)
{
    FAKE_LAUNCH("launch_fuse_masks_2d");
    check_launch("launch_fuse_masks_2d", s, {maskw, maskf});
    return hipSuccess;
}
hipError_t launch_eval_math(const float *, float *, size_t, int, hipStream_t) { FAKE_LAUNCH("launch_eval_math"); return hipSuccess; }
hipError_t launch_pack_mask_2d(const uint32_t *locked, int rows, int cols, int, int, int, uint32_t *maskw, hipStream_t s)
{
    FAKE_LAUNCH("launch_pack_mask_2d");
    check_launch("launch_pack_mask_2d", s, {locked, maskw});
    check_device(locked, (size_t)rows * cols * 4);
    return hipSuccess;
}
hipError_t launch_fill(float *p, size_t n, float v, hipStream_t s)
{
    FAKE_LAUNCH("launch_fill");
    check_launch("launch_fill", s, {p});
    check_device(p, n * 4);
    for (size_t i = 0; i < n; i++) p[i] = v;
    return hipSuccess;
}
hipError_t launch_set_cells_2d(float *u, uint32_t *maskw, int, int, int, unsigned, const unsigned *v, const unsigned *types, hipStream_t s, int, int, int, int)
{
    FAKE_LAUNCH("launch_set_cells_2d");
    check_launch("launch_set_cells_2d", s, {u, maskw, v, types});
    return hipSuccess;
}
hipError_t launch_tile_2d(const float *in, float *out, const uint32_t *maskw, int rows, int pitch, const TilePlan &plan, int, int, int, unsigned *delta, hipStream_t s,
                          float *tile_delta)
{
    FAKE_LAUNCH("launch_tile_2d");
    check_launch("launch_tile_2d", s, {in, out, maskw, delta});
    check_device(in, (size_t)rows * pitch * 4);
    check_device(out, (size_t)rows * pitch * 4);
    if (tile_delta)
        for (int t = 0; t < plan.tiles_r * plan.tiles_c; t++) tile_delta[t] = 0.0f;
    return hipSuccess;
}
hipError_t launch_follow_paths_2d(const float *, const uint32_t *, int, int, int, unsigned n_paths, const float *, float, float, unsigned, float *, unsigned *d_k,
                                  int *d_rc, hipStream_t)
{
    FAKE_LAUNCH("launch_follow_paths_2d");
    for (unsigned i = 0; i < n_paths; i++) { d_k[i] = 0; d_rc[i] = 12; }
    return hipSuccess;
}
hipError_t launch_sweep_3d(const float *in, float *out, const uint32_t *maskw, int m0, int m1, int pitch, int, int, int, int, unsigned *delta, hipStream_t s,
                           const Activity *act, int, int, const LaunchKnobs *)
{
    FAKE_LAUNCH("launch_sweep_3d");
    check_launch("launch_sweep_3d", s, {in, out, maskw, delta, act ? act->list_out : nullptr});
    check_device(in, (size_t)m0 * m1 * pitch * 4);
    check_device(out, (size_t)m0 * m1 * pitch * 4);
    return hipSuccess;
}
hipError_t launch_pack_mask_3d(const uint32_t *locked, int, int, int, int, uint32_t *maskw, hipStream_t s)
{
    FAKE_LAUNCH("launch_pack_mask_3d");
    check_launch("launch_pack_mask_3d", s, {locked, maskw});
    return hipSuccess;
}
}  // namespace epic_hip
