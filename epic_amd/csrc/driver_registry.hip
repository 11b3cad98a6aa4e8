// driver_registry.hip -- the registry of library-side contexts (one per caller's Harmonic*), their dimensions and uploads, and the
// device-state lifecycle entry points of the C-ABI (reference: libepic/src/harmonic/harmonic_model_gpu.cu:34-204, harmonic_gpu.cu:204-223,
// :307-324, :418-434).  See driver.h for the design of the host driver as a whole.
#include "driver.h"

namespace epic_drv {

std::mutex g_mu;
std::unordered_map<Harmonic *, Ctx *> g_ctx;

void report(const char *fn, const char *msg) { fprintf(stderr, "Error[%s]: %s\n", fn, msg); }

void free_spare(Ctx *c)   // the third u buffer of the small-grid path goes wherever the two others go
{
    if (c->spare) (void)hipFree(c->spare);
    c->spare = nullptr;
}

Ctx *find_ctx(Harmonic *h)
{
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_ctx.find(h);
    return it == g_ctx.end() ? nullptr : it->second;
}

bool dims_from(const Harmonic *h, Ctx *c)
{
    // n = 4: the reference holds the state on the device and sweeps NOTHING (harmonic_gpu.cu:156-162, :327-336 -- the n == 4
    // branches are empty, currentIteration still counts); the same here: state resident, every update a counting no-op.
    if (h->n != 2 && h->n != 3 && h->n != 4) return false;
    for (unsigned i = 0; i < h->n; i++)
        if (h->m[i] < (h->n == 4 ? 1u : 3u) || h->m[i] > (1u << 30)) return false;
    c->n = (int)h->n;
    for (unsigned i = 0; i < 4; i++) c->m[i] = i < h->n ? (int)h->m[i] : 0;
    c->cols = c->m[c->n - 1];
    long long rows = 1;
    for (int i = 0; i + 1 < c->n; i++) {
        rows *= c->m[i];
        if (rows > 0x7fffffffLL) return false;
    }
    c->rows = (int)rows;
    c->pitch = epic_hip::pitch_for_cols(c->cols);
    resolve_tracking(c);
    return true;
}

// dims_from() on the context that owns device state: also (re)decides single- or multi-device mode.  Changing dimensions
// while one kind of state is still resident is refused by the callers (same_dims), so the layout never changes under
// live buffers.
bool dims_into_ctx(const Harmonic *h, Ctx *c)
{
    const int rows0 = c->rows, cols0 = c->cols, n0 = c->n;
    if (!dims_from(h, c)) return false;
    // the measured task heights belong to ONE grid (the context survives a re-initialisation with other dimensions)
    if (c->rows != rows0 || c->cols != cols0 || c->n != n0) c->tuned_rows[0] = c->tuned_rows[1] = c->tuned_rows[2] = c->pair_rows = 0;
    if (!c->devices.empty()) multi_plan(c);
    return true;
}

bool same_dims(const Harmonic *h, const Ctx *c)
{
    if ((int)h->n != c->n) return false;
    for (unsigned i = 0; i < h->n; i++)
        if ((int)h->m[i] != c->m[i]) return false;
    return true;
}

// Create (or fetch) the context of this Harmonic; sets up the stream and the pinned readback word.
Ctx *get_ctx(Harmonic *h, bool create)
{
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_ctx.find(h);
    if (it != g_ctx.end()) {
        Ctx *c = it->second;
        // A Harmonic whose fields are all null but which we still track was freed and re-created by the
        // caller without uninitialize: drop the stale device state.
        if (h->d_m == nullptr && h->d_u == nullptr && h->d_locked == nullptr && h->d_delta == nullptr &&
            (c->buf[0] || c->maskw || c->d_m || c->d_delta || multi_holds_anything(c))) {
            drop_graphs(c);
            if (c->multi()) multi_destroy(c);
            for (float *&b : c->buf) { if (b) (void)hipFree(b); b = nullptr; }
            free_spare(c);
            if (c->maskw) (void)hipFree(c->maskw);
            if (c->d_m) (void)hipFree(c->d_m);
            if (c->d_delta) (void)hipFree(c->d_delta);
            c->maskw = nullptr; c->d_m = nullptr; c->d_delta = nullptr;
            // logically a NEW context: nothing pending, the environment read again, nothing measured
            c->pending = 0;
            c->ahead.live = false;
            c->tick_plain = 0;
            c->defer_bypass = false;
            c->cfg = Config::from_env();
            apply_config(c);
            c->tuned_rows[0] = c->tuned_rows[1] = c->tuned_rows[2] = c->pair_rows = 0;
            c->graphs_broken = false;
        }
        return c;
    }
    if (!create) return nullptr;
    Ctx *c = new Ctx();
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        (void)hipGetLastError();
        delete c;
        return nullptr;
    }
    if (hipHostMalloc((void **)&c->h_delta, 64, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipStreamDestroy(c->stream);
        delete c;
        return nullptr;
    }
    if (hipHostMalloc((void **)&c->h_tile_delta, 2 * kTileDeltaCap * sizeof(float), hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();   // (the tile path then checks with the plain sweep)
        c->h_tile_delta = nullptr;
    }
    c->cfg = Config::from_env();   // the environment, once per context (driver_config.h)
    apply_config(c);
    g_ctx[h] = c;
    return c;
}

// The mode fields of a context from its Config: at creation, and on epic_hip_config_reload (the setters of include/epic_hip.h
// change the same fields per context afterwards).  The device list is validated against what the runtime shows.
void apply_config(Ctx *c)
{
    const Config &cfg = c->cfg;
    c->rows_per_task = cfg.rows_per_task;
    c->math = cfg.math;
    c->redblack = cfg.redblack;
    c->track_mode = cfg.track_mode;
    c->halo_env = cfg.halo;
    c->devices.clear();
    if (!cfg.devices_text.empty()) {  // "0,1,2,3"; a device may be named more than once ("0,0,0,0": four slabs on one GPU)
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess) { (void)hipGetLastError(); ndev = 0; }
        bool ok = !cfg.devices_malformed && !cfg.devices.empty();
        for (int d : cfg.devices)
            if (d >= ndev) ok = false;
        if (ok) c->devices = cfg.devices;
        else fprintf(stderr, "Warning[epic_hip]: EPIC_HIP_DEVICES=%s ignored (%d device(s) visible)\n", cfg.devices_text.c_str(), ndev);
    }
}

void drop_ctx_if_empty(Harmonic *h)
{
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_ctx.find(h);
    if (it == g_ctx.end()) return;
    Ctx *c = it->second;
    if (c->buf[0] || c->maskw || c->d_m || c->d_delta || multi_holds_anything(c)) return;
    drop_graphs(c);
    if (c->multi()) multi_destroy(c);
    c->trk.release();
    c->trk_f.release();
    if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
    if (c->h_delta) (void)hipHostFree(c->h_delta);
    if (c->h_tile_delta) (void)hipHostFree(c->h_tile_delta);
    free_spare(c);
    for (hipEvent_t &e : c->ev_blk) { if (e) (void)hipEventDestroy(e); e = nullptr; }
    delete c;
    g_ctx.erase(it);
}

bool ready(const Harmonic *h, const Ctx *c)
{
    if (c && c->multi()) return multi_ready(c) && h->d_u && h->d_locked;
    return c && c->buf[0] && c->buf[1] && c->maskw && h->d_u && h->d_locked;
}

float *current_u(const Ctx *c) { return c->multi() ? c->slabs[0].buf[c->cur] : c->buf[c->cur]; }
bool has_delta(const Ctx *c) { return c->multi() ? c->slabs[0].d_delta != nullptr : c->d_delta != nullptr; }

int upload_u(Harmonic *h, Ctx *c, const char *fn)
{
    if (c->multi()) return multi_upload_u(h, c, fn);
    force_all(c);  // new values: no tile may be left out on the strength of the old work lists
    // padding columns hold the obstacle seed; both buffers, so that whichever is read first is complete
    if (c->pitch != c->cols) {
        for (int b = 0; b < 2; b++)
            if (epic_hip::launch_fill(c->buf[b], (size_t)c->rows * c->pitch, -1e6f, c->stream) != hipSuccess) {
                report(fn, "Failed to initialise device-side memory for the potential values.");
                return EPIC_ERROR_KERNEL_EXECUTION;
            }
        if (hipStreamSynchronize(c->stream) != hipSuccess) return EPIC_ERROR_DEVICE_SYNCHRONIZE;
    }
    c->cur = 0;
    if (hipMemcpy2D(c->buf[0], (size_t)c->pitch * sizeof(float), h->u, (size_t)c->cols * sizeof(float),
                    (size_t)c->cols * sizeof(float), (size_t)c->rows, hipMemcpyHostToDevice) != hipSuccess) {
        report(fn, "Failed to copy memory from host to device for the potential values.");
        return EPIC_ERROR_MEMCPY_TO_DEVICE;
    }
    h->d_u = c->buf[0];
    return EPIC_SUCCESS;
}

int upload_locked(Harmonic *h, Ctx *c, const char *fn)
{
    if (c->multi()) return multi_upload_locked(h, c, fn);
    force_all(c);
    if (c->n == 4) return EPIC_SUCCESS;   // never read: no lane masks to derive
    const size_t cells = (size_t)c->rows * c->cols;
    uint32_t *tmp = nullptr;
    if (hipMalloc((void **)&tmp, cells * sizeof(uint32_t)) != hipSuccess) {
        (void)hipGetLastError();
        report(fn, "Failed to allocate device-side staging memory for the locked cells.");
        return EPIC_ERROR_DEVICE_MALLOC;
    }
    int rc = EPIC_SUCCESS;
    if (hipMemcpy(tmp, h->locked, cells * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess) {
        report(fn, "Failed to copy memory from host to device for the locked cells.");
        rc = EPIC_ERROR_MEMCPY_TO_DEVICE;
    } else {
        hipError_t e = c->n == 2
                           ? epic_hip::launch_pack_mask_2d(tmp, c->rows, c->cols, c->pitch, 0, 0, c->maskw, c->stream)
                           : epic_hip::launch_pack_mask_3d(tmp, c->m[0], c->m[1], c->m[2], c->pitch, c->maskw, c->stream);
        if (e == hipSuccess && c->n == 2) e = epic_hip::launch_fuse_masks_2d(c->maskw, c->rows, c->pitch, c->maskf(), c->stream);
        if (e != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) {
            report(fn, "Failed to execute the 'pack mask' kernel.");
            rc = EPIC_ERROR_KERNEL_EXECUTION;
        }
    }
    (void)hipFree(tmp);
    return rc;
}

}  // namespace epic_drv

using namespace epic_drv;

namespace epic {
extern "C" {

// ---------------------------------------------------------------------------------------------------------
// device-state lifecycle (reference: libepic/src/harmonic/harmonic_model_gpu.cu)
// ---------------------------------------------------------------------------------------------------------

int harmonic_initialize_dimension_size_gpu(Harmonic *harmonic)  // harmonic_model_gpu.cu:34-59
{
    static const char *fn = "harmonic_initialize_dimension_size_gpu";
    if (harmonic == nullptr || harmonic->n == 0 || harmonic->m == nullptr) {
        report(fn, "Invalid input.");
        return EPIC_ERROR_INVALID_DATA;
    }
    Ctx *c = get_ctx(harmonic, true);
    if (c == nullptr) {
        report(fn, "Failed to allocate device-side memory for the dimension size.");
        return EPIC_ERROR_DEVICE_MALLOC;
    }
    {
        const int frc = flush_pending(harmonic, c, fn);   // (iterations counted by harmonic_update_gpu: driver_loop.hip)
        if (frc != EPIC_SUCCESS) return frc;
    }
    if (c->d_m) { (void)hipFree(c->d_m); c->d_m = nullptr; }  // re-initialise without uninitialise (harmonic.py:67-71 then harmonic_gpu.cu:172)
    if (hipMalloc((void **)&c->d_m, harmonic->n * sizeof(unsigned)) != hipSuccess) {
        (void)hipGetLastError();
        report(fn, "Failed to allocate device-side memory for the dimension size.");
        drop_ctx_if_empty(harmonic);
        return EPIC_ERROR_DEVICE_MALLOC;
    }
    if (hipMemcpy(c->d_m, harmonic->m, harmonic->n * sizeof(unsigned), hipMemcpyHostToDevice) != hipSuccess) {
        report(fn, "Failed to copy memory from host to device for the dimension size.");
        (void)hipFree(c->d_m);  // (the reference leaks it here, harmonic_model_gpu.cu:50-55)
        c->d_m = nullptr;
        harmonic->d_m = nullptr;
        drop_ctx_if_empty(harmonic);
        return EPIC_ERROR_MEMCPY_TO_DEVICE;
    }
    harmonic->d_m = c->d_m;
    return EPIC_SUCCESS;
}

int harmonic_uninitialize_dimension_size_gpu(Harmonic *harmonic)  // harmonic_model_gpu.cu:62-75
{
    if (harmonic == nullptr) return EPIC_ERROR_INVALID_DATA;
    Ctx *c = find_ctx(harmonic);
    int rc = flush_pending(harmonic, c, "harmonic_uninitialize_dimension_size_gpu");   // (the field lives on: the iterations counted so far belong to it)
    if (c && c->d_m) {
        if (hipFree(c->d_m) != hipSuccess) {
            report("harmonic_uninitialize_dimension_size_gpu", "Failed to free device-side memory for the dimension size.");
            rc = EPIC_ERROR_DEVICE_FREE;
        }
        c->d_m = nullptr;
    }
    harmonic->d_m = nullptr;
    drop_ctx_if_empty(harmonic);
    return rc;
}

int harmonic_initialize_potential_values_gpu(Harmonic *harmonic)  // harmonic_model_gpu.cu:78-110
{
    static const char *fn = "harmonic_initialize_potential_values_gpu";
    if (harmonic == nullptr || harmonic->n == 0 || harmonic->m == nullptr || harmonic->u == nullptr) {
        report(fn, "Invalid input.");
        return EPIC_ERROR_INVALID_DATA;
    }
    Ctx probe;
    if (!dims_from(harmonic, &probe)) {
        report(fn, "Invalid input (n = 2 and n = 3 need every m[i] >= 3; n = 4 is held but never swept; other n are not supported).");
        return EPIC_ERROR_INVALID_DATA;
    }
    Ctx *c = get_ctx(harmonic, true);
    if (c == nullptr) {
        report(fn, "Failed to allocate device-side memory for the potential values.");
        return EPIC_ERROR_DEVICE_MALLOC;
    }
    if ((c->maskw || (c->multi() && c->slabs[0].maskw)) && !same_dims(harmonic, c)) {
        report(fn, "Invalid input (dimensions differ from the locked cells already on the device).");
        return EPIC_ERROR_INVALID_DATA;
    }
    c->pending = 0;   // iterations counted on a field that is being replaced
    c->ahead.live = false;
    drop_graphs(c);
    for (float *&b : c->buf) { if (b) (void)hipFree(b); b = nullptr; }
    free_spare(c);
    if (c->multi()) { DeviceGuard g; multi_free_u(c); }
    dims_into_ctx(harmonic, c);
    if (c->plan_failed) {   // (EPIC_HIP_DEVICES: not silently on one device instead -- the caller asked for the node)
        report(fn, "Failed to create the streams, events and staging buffers of the device slabs.");
        harmonic->d_u = nullptr;
        drop_ctx_if_empty(harmonic);
        return EPIC_ERROR_DEVICE_MALLOC;
    }
    if (c->multi()) {  // one pair of buffers per slab, each on its device
        DeviceGuard g;
        for (auto &sl : c->slabs)
            for (int b = 0; b < 2; b++)
                if (hipSetDevice(sl.dev) != hipSuccess ||
                    hipMalloc((void **)&sl.buf[b], (size_t)sl.rows * unit_floats(c) * sizeof(float)) != hipSuccess) {
                    (void)hipGetLastError();
                    report(fn, "Failed to allocate device-side memory for the potential values.");
                    multi_free_u(c);
                    harmonic->d_u = nullptr;
                    drop_ctx_if_empty(harmonic);
                    return EPIC_ERROR_DEVICE_MALLOC;
                }
        return upload_u(harmonic, c, fn);
    }
    for (int b = 0; b < 2; b++) {
        if (hipMalloc((void **)&c->buf[b], c->u_bytes()) != hipSuccess) {
            (void)hipGetLastError();
            report(fn, "Failed to allocate device-side memory for the potential values.");
            for (float *&bb : c->buf) { if (bb) (void)hipFree(bb); bb = nullptr; }
            harmonic->d_u = nullptr;
            drop_ctx_if_empty(harmonic);
            return EPIC_ERROR_DEVICE_MALLOC;
        }
    }
    return upload_u(harmonic, c, fn);
}

int harmonic_uninitialize_potential_values_gpu(Harmonic *harmonic)  // harmonic_model_gpu.cu:113-126
{
    if (harmonic == nullptr) return EPIC_ERROR_INVALID_DATA;
    int rc = EPIC_SUCCESS;
    Ctx *c = find_ctx(harmonic);
    if (c) {
        c->pending = 0;   // iterations counted on a field that is being dropped
        c->ahead.live = false;
        if (c->stream) (void)hipStreamSynchronize(c->stream);
        drop_graphs(c);
        if (c->multi()) { DeviceGuard g; multi_free_u(c); }
        for (float *&b : c->buf) {
            if (b && hipFree(b) != hipSuccess) {
                report("harmonic_uninitialize_potential_values_gpu", "Failed to free device-side memory for the potential values.");
                rc = EPIC_ERROR_DEVICE_FREE;
            }
            b = nullptr;
        }
        free_spare(c);
    }
    harmonic->d_u = nullptr;
    drop_ctx_if_empty(harmonic);
    return rc;
}

int harmonic_initialize_locked_gpu(Harmonic *harmonic)  // harmonic_model_gpu.cu:129-161
{
    static const char *fn = "harmonic_initialize_locked_gpu";
    if (harmonic == nullptr || harmonic->n == 0 || harmonic->m == nullptr || harmonic->locked == nullptr) {
        report(fn, "Invalid input.");
        return EPIC_ERROR_INVALID_DATA;
    }
    Ctx probe;
    if (!dims_from(harmonic, &probe)) {
        report(fn, "Invalid input (n = 2 and n = 3 need every m[i] >= 3; n = 4 is held but never swept; other n are not supported).");
        return EPIC_ERROR_INVALID_DATA;
    }
    Ctx *c = get_ctx(harmonic, true);
    if (c == nullptr) {
        report(fn, "Failed to allocate device-side memory for the locked cells.");
        return EPIC_ERROR_DEVICE_MALLOC;
    }
    if ((c->buf[0] || (c->multi() && c->slabs[0].buf[0])) && !same_dims(harmonic, c)) {
        report(fn, "Invalid input (dimensions differ from the potential values already on the device).");
        return EPIC_ERROR_INVALID_DATA;
    }
    {
        const int frc = flush_pending(harmonic, c, fn);   // the iterations counted so far act under the mask as it was
        if (frc != EPIC_SUCCESS) return frc;
    }
    drop_graphs(c);
    if (c->maskw) { (void)hipFree(c->maskw); c->maskw = nullptr; }
    if (c->multi()) { DeviceGuard g; multi_free_mask(c); }
    dims_into_ctx(harmonic, c);
    if (c->plan_failed) {
        report(fn, "Failed to create the streams, events and staging buffers of the device slabs.");
        harmonic->d_locked = nullptr;
        drop_ctx_if_empty(harmonic);
        return EPIC_ERROR_DEVICE_MALLOC;
    }
    if (c->multi()) {
        {
            DeviceGuard g;
            for (auto &sl : c->slabs)
                if (hipSetDevice(sl.dev) != hipSuccess ||
                    hipMalloc((void **)&sl.maskw, sizeof(uint32_t) * slab_mask_words(c, sl)) != hipSuccess) {
                    (void)hipGetLastError();
                    report(fn, "Failed to allocate device-side memory for the locked cells.");
                    multi_free_mask(c);
                    harmonic->d_locked = nullptr;
                    drop_ctx_if_empty(harmonic);
                    return EPIC_ERROR_DEVICE_MALLOC;
                }
        }
        int rc = upload_locked(harmonic, c, fn);
        if (rc == EPIC_SUCCESS) harmonic->d_locked = c->slabs[0].maskw;
        return rc;
    }
    if (hipMalloc((void **)&c->maskw, c->mask_bytes()) != hipSuccess) {
        (void)hipGetLastError();
        report(fn, "Failed to allocate device-side memory for the locked cells.");
        harmonic->d_locked = nullptr;
        drop_ctx_if_empty(harmonic);
        return EPIC_ERROR_DEVICE_MALLOC;
    }
    int rc = upload_locked(harmonic, c, fn);
    if (rc == EPIC_SUCCESS) harmonic->d_locked = c->maskw;
    return rc;
}

int harmonic_uninitialize_locked_gpu(Harmonic *harmonic)  // harmonic_model_gpu.cu:164-177
{
    if (harmonic == nullptr) return EPIC_ERROR_INVALID_DATA;
    Ctx *c = find_ctx(harmonic);
    int rc = flush_pending(harmonic, c, "harmonic_uninitialize_locked_gpu");   // (the field can still be read back afterwards)
    if (c && c->multi()) { DeviceGuard g; multi_free_mask(c); }
    if (c && c->maskw) {
        if (c->stream) (void)hipStreamSynchronize(c->stream);
        drop_graphs(c);
        if (hipFree(c->maskw) != hipSuccess) {
            report("harmonic_uninitialize_locked_gpu", "Failed to free device-side memory for the locked cells.");
            rc = EPIC_ERROR_DEVICE_FREE;
        }
        c->maskw = nullptr;
    }
    harmonic->d_locked = nullptr;
    drop_ctx_if_empty(harmonic);
    return rc;
}

int harmonic_update_model_gpu(Harmonic *harmonic)  // harmonic_model_gpu.cu:172-204
{
    static const char *fn = "harmonic_update_model_gpu";
    if (harmonic == nullptr || harmonic->n == 0 || harmonic->m == nullptr || harmonic->u == nullptr ||
        harmonic->d_u == nullptr || harmonic->locked == nullptr || harmonic->d_locked == nullptr) {
        report(fn, "Invalid data.");
        return EPIC_ERROR_INVALID_DATA;
    }
    Ctx *c = find_ctx(harmonic);
    if (!ready(harmonic, c) || !same_dims(harmonic, c)) {
        report(fn, "Invalid data.");
        return EPIC_ERROR_INVALID_DATA;
    }
    c->pending = 0;   // field and mask are both replaced: iterations counted on the old ones have nothing left to show
    c->ahead.live = false;
    if (c->multi()) { DeviceGuard g; multi_sync(c); }
    if (hipStreamSynchronize(c->stream) != hipSuccess) return EPIC_ERROR_DEVICE_SYNCHRONIZE;
    int rc = upload_u(harmonic, c, fn);
    if (rc != EPIC_SUCCESS) return rc;
    return upload_locked(harmonic, c, fn);
}

int harmonic_initialize_gpu(Harmonic *harmonic, unsigned int numThreads)  // harmonic_gpu.cu:204-223
{
    static const char *fn = "harmonic_initialize_gpu";
    (void)numThreads;
    if (harmonic == nullptr || harmonic->n == 0 || harmonic->m == nullptr || harmonic->d_delta != nullptr) {
        report(fn, "Invalid input.");
        return EPIC_ERROR_INVALID_DATA;
    }
    Ctx *c = get_ctx(harmonic, true);
    if (c == nullptr) {
        report(fn, "Failed to allocate device-side memory for delta.");
        return EPIC_ERROR_DEVICE_MALLOC;
    }
    if (c->d_delta) { (void)hipFree(c->d_delta); c->d_delta = nullptr; }
    if (c->multi()) {  // one delta word per slab; the host takes the maximum
        DeviceGuard g;
        multi_free_delta(c);
        for (auto &sl : c->slabs)
            if (hipSetDevice(sl.dev) != hipSuccess || hipMalloc((void **)&sl.d_delta, 64) != hipSuccess) {
                (void)hipGetLastError();
                report(fn, "Failed to allocate device-side memory for delta.");
                multi_free_delta(c);
                drop_ctx_if_empty(harmonic);
                return EPIC_ERROR_DEVICE_MALLOC;
            }
        harmonic->d_delta = reinterpret_cast<float *>(c->slabs[0].d_delta);
        return EPIC_SUCCESS;
    }
    if (hipMalloc((void **)&c->d_delta, 64) != hipSuccess) {
        (void)hipGetLastError();
        report(fn, "Failed to allocate device-side memory for delta.");
        drop_ctx_if_empty(harmonic);
        return EPIC_ERROR_DEVICE_MALLOC;
    }
    harmonic->d_delta = reinterpret_cast<float *>(c->d_delta);
    return EPIC_SUCCESS;
}

int harmonic_uninitialize_gpu(Harmonic *harmonic)  // harmonic_gpu.cu:307-324
{
    if (harmonic == nullptr) return EPIC_ERROR_INVALID_DATA;
    Ctx *c = find_ctx(harmonic);
    int rc = flush_pending(harmonic, c, "harmonic_uninitialize_gpu");
    if (c && c->multi()) { DeviceGuard g; multi_free_delta(c); }
    if (c && c->d_delta) {
        if (c->stream) (void)hipStreamSynchronize(c->stream);
        if (hipFree(c->d_delta) != hipSuccess) {
            report("harmonic_uninitialize_gpu", "Failed to free device-side memory for delta.");
            rc = EPIC_ERROR_DEVICE_FREE;
        }
        c->d_delta = nullptr;
    }
    harmonic->d_delta = nullptr;
    drop_ctx_if_empty(harmonic);
    return rc;
}

int harmonic_get_potential_values_gpu(Harmonic *harmonic)  // harmonic_gpu.cu:418-434
{
    static const char *fn = "harmonic_get_potential_values_gpu";
    Ctx *c = harmonic ? find_ctx(harmonic) : nullptr;
    if (!harmonic || harmonic->u == nullptr || !c || !(c->buf[0] || (c->multi() && c->slabs[0].buf[0])) || harmonic->d_u == nullptr) {
        report(fn, "Invalid data.");
        return EPIC_ERROR_INVALID_DATA;
    }
    {
        const int frc = flush_pending(harmonic, c, fn);   // the caller sees the field of every iteration it has asked for
        if (frc != EPIC_SUCCESS) return frc;
    }
    if (c->multi()) return multi_get_values(harmonic, c, fn);
    if (hipStreamSynchronize(c->stream) != hipSuccess) {
        report(fn, "Failed to synchronize the device before reading the potential values.");
        return EPIC_ERROR_DEVICE_SYNCHRONIZE;
    }
    if (hipMemcpy2D(harmonic->u, (size_t)c->cols * sizeof(float), c->buf[c->cur], (size_t)c->pitch * sizeof(float),
                    (size_t)c->cols * sizeof(float), (size_t)c->rows, hipMemcpyDeviceToHost) != hipSuccess) {
        report(fn, "Failed to copy memory from device to host for the potential values.");
        return EPIC_ERROR_MEMCPY_TO_HOST;
    }
    return EPIC_SUCCESS;
}

}  // extern "C"
}  // namespace epic

