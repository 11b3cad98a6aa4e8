// A FAKE HIP runtime header -- TEST INFRASTRUCTURE ONLY (tests/test_host_driver_faults.py).
//
// The 2.4 kLoC host driver of the library (epic_amd/csrc/driver_*.hip: registry, lifecycle, unwind paths, the issuing
// threads of the multi-device mode) is compiled against THIS header with g++ and the sanitizers, in the CPU container, so that
// every allocation / copy / launch site can be made to fail in turn (fake_hip.cpp: "fail the n-th call") and the unwind checked
// for leaks, double frees and stale pointers.  Memory is malloc-backed, streams execute at once, kernels are no-ops.  Nothing
// of this directory is compiled into, linked with or loaded by libepic.so.
#pragma once
#include <stddef.h>
#include <stdint.h>

#define __host__
#define __device__
#define __global__

typedef enum hipError_t {
    hipSuccess = 0,
    hipErrorInvalidValue = 1,
    hipErrorOutOfMemory = 2,
    hipErrorInvalidDevice = 101,
    hipErrorNotReady = 600,
    hipErrorPeerAccessAlreadyEnabled = 704,
    hipErrorNotSupported = 801,
    hipErrorUnknown = 999,
} hipError_t;

typedef struct fakeHipStream *hipStream_t;
typedef struct fakeHipEvent *hipEvent_t;
typedef struct fakeHipGraph *hipGraph_t;
typedef struct fakeHipGraphExec *hipGraphExec_t;
typedef struct fakeHipGraphNode *hipGraphNode_t;

typedef enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 } hipMemcpyKind;
typedef enum hipStreamCaptureStatus { hipStreamCaptureStatusNone = 0, hipStreamCaptureStatusActive = 1 } hipStreamCaptureStatus;
typedef enum hipStreamCaptureMode { hipStreamCaptureModeGlobal = 0, hipStreamCaptureModeThreadLocal = 1 } hipStreamCaptureMode;
typedef enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 16 } hipDeviceAttribute_t;

#define hipStreamNonBlocking 1u
#define hipEventDisableTiming 2u
#define hipHostMallocDefault 0u
#define hipHostMallocPortable 1u
#define hipHostMallocMapped 2u

#ifdef __cplusplus
extern "C" {
#endif
hipError_t hipGetLastError(void);
hipError_t hipGetDeviceCount(int *n);
hipError_t hipGetDevice(int *dev);
hipError_t hipSetDevice(int dev);
hipError_t hipDeviceGetAttribute(int *value, hipDeviceAttribute_t attr, int dev);
hipError_t hipDeviceCanAccessPeer(int *can, int dev, int peer);
hipError_t hipDeviceEnablePeerAccess(int peer, unsigned flags);
hipError_t hipExtGetLinkTypeAndHopCount(int d1, int d2, uint32_t *linktype, uint32_t *hops);
hipError_t hipMalloc(void **p, size_t bytes);
hipError_t hipFree(void *p);
hipError_t hipHostMalloc(void **p, size_t bytes, unsigned flags);
hipError_t hipHostFree(void *p);
hipError_t hipMemcpy(void *dst, const void *src, size_t bytes, hipMemcpyKind kind);
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t bytes, hipMemcpyKind kind, hipStream_t s);
hipError_t hipMemcpy2D(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height, hipMemcpyKind kind);
hipError_t hipMemcpyPeerAsync(void *dst, int ddev, const void *src, int sdev, size_t bytes, hipStream_t s);
hipError_t hipMemset(void *p, int v, size_t bytes);
hipError_t hipMemsetAsync(void *p, int v, size_t bytes, hipStream_t s);
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags);
hipError_t hipStreamIsCapturing(hipStream_t s, hipStreamCaptureStatus *status);
hipError_t hipStreamBeginCapture(hipStream_t s, hipStreamCaptureMode mode);
hipError_t hipStreamEndCapture(hipStream_t s, hipGraph_t *g);
hipError_t hipGraphInstantiate(hipGraphExec_t *e, hipGraph_t g, hipGraphNode_t *err_node, char *log, size_t log_bytes);
hipError_t hipGraphDestroy(hipGraph_t g);
hipError_t hipGraphExecDestroy(hipGraphExec_t e);
hipError_t hipGraphLaunch(hipGraphExec_t e, hipStream_t s);
hipError_t hipEventCreate(hipEvent_t *e);
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventQuery(hipEvent_t e);
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b);

// ---- controls of the fake (tests/fake_hip/fake_hip.cpp) ----
void fake_hip_fail_at(long n);            // the n-th fallible call from now fails (0: none); the counter restarts
long fake_hip_calls(void);                // fallible calls since the last fake_hip_fail_at()
int fake_hip_failed(void);                // whether the armed failure has fired
const char *fake_hip_failed_call(void);   // its name
long fake_hip_live(int kind);             // live objects: 0 device allocations, 1 pinned host allocations, 2 streams, 3 events, 4 graphs
long fake_hip_misuse(void);               // frees / destroys of things that were not live, copies into unknown device memory
void fake_hip_set_devices(int n);          // devices the runtime shows (also forgets every enabled peer access)
void fake_hip_set_peer_capable(int on);    // what hipDeviceCanAccessPeer says about two different devices (default 1)
long fake_hip_affinity(void);              // violations of the device rules: kernels / events / copies on the wrong device, memory of another
                                           // device touched without peer access, hipMemcpyPeerAsync naming the wrong devices
const char *fake_hip_first_affinity(void); // the first one, in words
#ifdef __cplusplus
}
#endif
