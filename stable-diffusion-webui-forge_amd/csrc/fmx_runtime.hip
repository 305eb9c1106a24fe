// Runtime glue of libfmx: last-error string, device query, HIP-graph capture helpers, HIP-event timing.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "fmx_common.hpp"

static thread_local char g_err[512] = "";

int fmx_set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

#define HIP_TRY(expr, what)                                                                   \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess) return fmx_set_error((int)_e, "%s: %s", what, hipGetErrorString(_e)); \
  } while (0)

extern "C" int fmx_abi_version(void) { return FMX_ABI_VERSION; }
extern "C" const char* fmx_last_error(void) { return g_err; }

extern "C" int fmx_device_info(int* cu_count, int* wave_size, char* arch, int arch_len) {
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev), "hipGetDevice");
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, dev), "hipGetDeviceProperties");
  if (cu_count) *cu_count = prop.multiProcessorCount;
  if (wave_size) *wave_size = prop.warpSize;
  if (arch && arch_len > 0) {
    strncpy(arch, prop.gcnArchName, (size_t)arch_len - 1);
    arch[arch_len - 1] = 0;
  }
  return FMX_OK;
}

extern "C" int fmx_graph_begin(void* stream) {
  HIP_TRY(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal), "hipStreamBeginCapture");
  return FMX_OK;
}

extern "C" int fmx_graph_end(void* stream, void** graph_exec_out) {
  FMX_REQUIRE(graph_exec_out, "graph_end: null out");
  hipGraph_t graph = nullptr;
  HIP_TRY(hipStreamEndCapture((hipStream_t)stream, &graph), "hipStreamEndCapture");
  hipGraphExec_t exec = nullptr;
  hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (e != hipSuccess) return fmx_set_error((int)e, "hipGraphInstantiate: %s", hipGetErrorString(e));
  *graph_exec_out = (void*)exec;
  return FMX_OK;
}

extern "C" int fmx_graph_launch(void* graph_exec, void* stream) {
  FMX_REQUIRE(graph_exec, "graph_launch: null graph");
  HIP_TRY(hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream), "hipGraphLaunch");
  return FMX_OK;
}

extern "C" int fmx_graph_destroy(void* graph_exec) {
  if (graph_exec) HIP_TRY(hipGraphExecDestroy((hipGraphExec_t)graph_exec), "hipGraphExecDestroy");
  return FMX_OK;
}

extern "C" int fmx_event_create(void** ev_out) {
  FMX_REQUIRE(ev_out, "event_create: null out");
  hipEvent_t ev;
  HIP_TRY(hipEventCreate(&ev), "hipEventCreate");
  *ev_out = (void*)ev;
  return FMX_OK;
}

extern "C" int fmx_event_record(void* ev, void* stream) {
  HIP_TRY(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream), "hipEventRecord");
  return FMX_OK;
}

extern "C" int fmx_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms_out) {
  FMX_REQUIRE(ms_out, "event_elapsed: null out");
  HIP_TRY(hipEventSynchronize((hipEvent_t)ev_stop), "hipEventSynchronize");
  HIP_TRY(hipEventElapsedTime(ms_out, (hipEvent_t)ev_start, (hipEvent_t)ev_stop), "hipEventElapsedTime");
  return FMX_OK;
}

extern "C" int fmx_event_destroy(void* ev) {
  if (ev) HIP_TRY(hipEventDestroy((hipEvent_t)ev), "hipEventDestroy");
  return FMX_OK;
}
