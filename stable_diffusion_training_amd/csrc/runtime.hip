// Error plumbing and device probe of libsdtrain_hip.so (no global mutable state besides the thread-local message).
#include <stdarg.h>
#include <stdio.h>

#include "sdt_common.h"

static thread_local char g_err[512] = "";

void sdt_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" {

const char* sdt_last_error(void) { return g_err; }
int sdt_abi_version(void) { return 5; }
int sdt_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}


// ---- step-graph hand-off events: a captured step marks "this gradient bucket is complete" with an event-record NODE
// (hipEventRecordExternal), and the communication stream outside the graph waits on it before the bucket's all-reduce.
int sdt_event_create(void** event) {
  hipEvent_t ev;
  if (!event) { sdt_set_error("sdt_event_create: null out pointer"); return SDT_ERR_INVALID_ARG; }
  if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
    sdt_set_error("sdt_event_create: %s", hipGetErrorString(hipGetLastError()));
    return SDT_ERR_LAUNCH;
  }
  *event = (void*)ev;
  return SDT_OK;
}
int sdt_event_destroy(void* event) {
  if (event && hipEventDestroy((hipEvent_t)event) != hipSuccess) { (void)hipGetLastError(); return SDT_ERR_LAUNCH; }
  return SDT_OK;
}
int sdt_event_record(void* event, int external, hipStream_t stream) {
  hipEvent_t ev = (hipEvent_t)event;
  hipError_t e;
  if (external) {
    // add the event-record node by hand: it depends on everything the capturing stream has enqueued so far, and whatever
    // the stream enqueues next depends on it (hipEventRecordWithFlags(.., hipEventRecordExternal) is refused by this runtime)
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    unsigned long long id = 0;
    hipGraph_t graph = nullptr;
    const hipGraphNode_t* deps = nullptr;
    size_t ndeps = 0;
    e = hipStreamGetCaptureInfo_v2(stream, &st, &id, &graph, &deps, &ndeps);
    if (e == hipSuccess && st == hipStreamCaptureStatusActive) {
      hipGraphNode_t node;
      e = hipGraphAddEventRecordNode(&node, graph, deps, ndeps, ev);
      if (e == hipSuccess) e = hipStreamUpdateCaptureDependencies(stream, &node, 1, hipStreamSetCaptureDependencies);
      if (e != hipSuccess) { sdt_set_error("sdt_event_record (graph node): %s", hipGetErrorString(e)); (void)hipGetLastError(); return SDT_ERR_LAUNCH; }
      return SDT_OK;
    }
    if (e != hipSuccess) { sdt_set_error("sdt_event_record (capture info): %s", hipGetErrorString(e)); (void)hipGetLastError(); return SDT_ERR_LAUNCH; }
  }
  e = hipEventRecord(ev, stream);
  if (e != hipSuccess) { sdt_set_error("sdt_event_record: %s", hipGetErrorString(e)); (void)hipGetLastError(); return SDT_ERR_LAUNCH; }
  return SDT_OK;
}
int sdt_stream_wait_event(hipStream_t stream, void* event) {
  hipError_t e = hipStreamWaitEvent(stream, (hipEvent_t)event, 0);
  if (e != hipSuccess) { sdt_set_error("sdt_stream_wait_event: %s", hipGetErrorString(e)); (void)hipGetLastError(); return SDT_ERR_LAUNCH; }
  return SDT_OK;
}
// The reverse hand-off: a captured step waits, at a point INSIDE the graph, for work the host enqueues on another stream between
// two launches of that graph (the all-gather of the weight mirrors, recorded into `event` before the next launch).  On a capturing
// stream this adds an event-wait NODE (each launch waits for the event's most recent record at launch time); otherwise it is a
// plain hipStreamWaitEvent.  An event that has never been recorded is complete.
int sdt_stream_wait_event_external(hipStream_t stream, void* event) {
  hipEvent_t ev = (hipEvent_t)event;
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  unsigned long long id = 0;
  hipGraph_t graph = nullptr;
  const hipGraphNode_t* deps = nullptr;
  size_t ndeps = 0;
  hipError_t e = hipStreamGetCaptureInfo_v2(stream, &st, &id, &graph, &deps, &ndeps);
  if (e != hipSuccess) { sdt_set_error("sdt_stream_wait_event_external (capture info): %s", hipGetErrorString(e)); (void)hipGetLastError(); return SDT_ERR_LAUNCH; }
  if (st == hipStreamCaptureStatusActive) {
    hipGraphNode_t node;
    e = hipGraphAddEventWaitNode(&node, graph, deps, ndeps, ev);
    if (e == hipSuccess) e = hipStreamUpdateCaptureDependencies(stream, &node, 1, hipStreamSetCaptureDependencies);
    if (e != hipSuccess) { sdt_set_error("sdt_stream_wait_event_external (graph node): %s", hipGetErrorString(e)); (void)hipGetLastError(); return SDT_ERR_LAUNCH; }
    return SDT_OK;
  }
  return sdt_stream_wait_event(stream, event);
}

}  // extern "C"
