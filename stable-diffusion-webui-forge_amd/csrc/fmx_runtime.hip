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

// ---- development knobs: honoured only under FMX_ALLOW_KNOBS=1, and recorded -------------------------------------------------------------------
#include <stdlib.h>

#include <mutex>
#include <string>

static std::mutex g_knob_mu;
static std::string g_knobs_active, g_knobs_ignored;

static void knob_note(std::string& list, const char* name, const char* value) {
  const std::string item = std::string(name) + "=" + value;
  if (("," + list + ",").find("," + item + ",") != std::string::npos) return;
  if (!list.empty()) list += ",";
  list += item;
}

const char* fmx_knob(const char* name) {
  const char* v = getenv(name);
  if (!v) return nullptr;
  const char* a = getenv("FMX_ALLOW_KNOBS");
  const bool allow = a && strcmp(a, "1") == 0;   // exactly "1", the predicate forge_amd/_lib.py knob() uses
  std::lock_guard<std::mutex> lk(g_knob_mu);
  knob_note(allow ? g_knobs_active : g_knobs_ignored, name, v);
  return allow ? v : nullptr;
}

extern "C" int fmx_active_knobs(char* buf, int buf_len, int ignored) {
  FMX_REQUIRE(buf && buf_len > 0, "active_knobs: null buffer");
  std::lock_guard<std::mutex> lk(g_knob_mu);
  const std::string& s = ignored ? g_knobs_ignored : g_knobs_active;
  strncpy(buf, s.c_str(), (size_t)buf_len - 1);
  buf[buf_len - 1] = 0;
  return FMX_OK;
}

// FMX_SRC_HASH: csrc/src_hash.py over the kernel sources at build time (csrc/Makefile); bench.py compares it with the hash a committed PMC summary
// was taken on, so a stale .so beside newer sources is not credited with their measurements
#ifndef FMX_SRC_HASH
#define FMX_SRC_HASH "unknown"
#endif
#define FMX_STR2(x) #x
#define FMX_STR(x) FMX_STR2(x)
extern "C" const char* fmx_build_info(void) { return "src=" FMX_SRC_HASH " abi=" FMX_STR(FMX_ABI_VERSION) " arch=gfx950"; }

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
