// spock_rt_hip.cpp -- HIP implementation of spock_rt.hpp (the only one the product ships).
#include <hip/hip_runtime.h>

#include "spock_rt.hpp"

namespace spock {
namespace rt {
#define RT(x) return (int)(x)
int set_device(int dev) { RT(hipSetDevice(dev)); }
int get_device(int *dev) { RT(hipGetDevice(dev)); }
int stream_create(void **s) { hipStream_t st; hipError_t e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking); *s = (void *)st; RT(e); }
int stream_destroy(void *s) { RT(hipStreamDestroy((hipStream_t)s)); }
int dmalloc(void **p, size_t bytes) { RT(hipMalloc(p, bytes ? bytes : 8)); }
int dfree(void *p) { RT(hipFree(p)); }
int dzero(void *p, size_t bytes, void *s) { RT(hipMemsetAsync(p, 0, bytes, (hipStream_t)s)); }
int h2d(void *dst, const void *src, size_t bytes, void *s) { RT(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)s)); }
int d2h(void *dst, const void *src, size_t bytes, void *s) { RT(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)s)); }
int d2d(void *dst, const void *src, size_t bytes, void *s) { RT(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)s)); }
int sync(void *s) { RT(hipStreamSynchronize((hipStream_t)s)); }
int event_create(void **e) { hipEvent_t ev; hipError_t r = hipEventCreate(&ev); *e = (void *)ev; RT(r); }
int event_destroy(void *e) { RT(hipEventDestroy((hipEvent_t)e)); }
int event_record(void *e, void *s) { RT(hipEventRecord((hipEvent_t)e, (hipStream_t)s)); }
int stream_wait_event(void *s, void *e) { RT(hipStreamWaitEvent((hipStream_t)s, (hipEvent_t)e, 0)); }
int event_elapsed(float *ms, void *e0, void *e1) { RT(hipEventElapsedTime(ms, (hipEvent_t)e0, (hipEvent_t)e1)); }
bool graphs_supported() { return true; }
bool timing_supported() { return true; }
int capture_begin(void *s) { RT(hipStreamBeginCapture((hipStream_t)s, hipStreamCaptureModeThreadLocal)); }
int capture_end(void *s, void **exec) {
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture((hipStream_t)s, &g);
    if (e != hipSuccess) return (int)e;
    hipGraphExec_t ge = nullptr;
    e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphDestroy(g);
    *exec = (void *)ge;
    RT(e);
}
int graph_launch(void *exec, void *s) { RT(hipGraphLaunch((hipGraphExec_t)exec, (hipStream_t)s)); }
int graph_destroy(void *exec) { RT(hipGraphExecDestroy((hipGraphExec_t)exec)); }
const char *err(int code) { return hipGetErrorString((hipError_t)code); }
}  // namespace rt
}  // namespace spock
