// spock_rt.hpp -- the few device-runtime services the driver (spock_abi.cpp) needs.  The product links
// spock_rt_hip.cpp (HIP runtime on the handle's stream); tests/emu links a host-memory stand-in so that
// the driver logic can be exercised on a machine without a GPU.  Every call returns 0 on success.
#pragma once
#include <cstddef>

namespace spock {
namespace rt {
int set_device(int dev);
int get_device(int *dev);
int stream_create(void **s);
int stream_destroy(void *s);
int dmalloc(void **p, size_t bytes);
int dfree(void *p);
int dzero(void *p, size_t bytes, void *s);
int h2d(void *dst, const void *src, size_t bytes, void *s);
int d2h(void *dst, const void *src, size_t bytes, void *s);
int d2d(void *dst, const void *src, size_t bytes, void *s);
int sync(void *s);
int event_create(void **e);
int event_destroy(void *e);
int event_record(void *e, void *s);
int stream_wait_event(void *s, void *e);   // later work of stream s waits for the work recorded in e (also inside a capture: fork / join)
int event_elapsed(float *ms, void *e0, void *e1);
bool timing_supported();   // events measure device time (false in the host emulation)
// stream capture of a launch sequence into an executable graph
bool graphs_supported();
int capture_begin(void *s);
int capture_end(void *s, void **exec);
int graph_launch(void *exec, void *s);
int graph_destroy(void *exec);
const char *err(int code);
}  // namespace rt
}  // namespace spock
