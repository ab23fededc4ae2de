// Library plumbing: error text, device info, hipGraph capture, weight-free helpers.
#include <cstring>
#include <string>

#include "ds_common.h"

namespace ds {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
const char* get_error() { return g_err; }
}  // namespace ds

struct ds_graph {
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
};

extern "C" {

int ds_version(void) { return 4; }

const char* ds_last_error(void) { return ds::get_error(); }

int ds_device_info(int* cu_count, int* lds_bytes_per_cu, char* arch_name, int arch_name_len) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return ds::hip_fail(e, "hipGetDevice");
  hipDeviceProp_t p;
  e = hipGetDeviceProperties(&p, dev);
  if (e != hipSuccess) return ds::hip_fail(e, "hipGetDeviceProperties");
  if (cu_count) *cu_count = p.multiProcessorCount;
  if (lds_bytes_per_cu) *lds_bytes_per_cu = (int)p.maxSharedMemoryPerMultiProcessor;
  if (arch_name && arch_name_len > 0) {
    strncpy(arch_name, p.gcnArchName, arch_name_len - 1);
    arch_name[arch_name_len - 1] = 0;
  }
  return DS_OK;
}

int ds_graph_begin_capture(void* stream) {
  hipError_t e = hipStreamBeginCapture(ds::as_stream(stream), hipStreamCaptureModeThreadLocal);
  if (e != hipSuccess) return ds::hip_fail(e, "hipStreamBeginCapture");
  return DS_OK;
}

int ds_graph_end_capture(void* stream, ds_graph** out_graph, int* node_count) {
  DS_REQUIRE(out_graph != nullptr, DS_ERR_NULL, "ds_graph_end_capture: out_graph is NULL");
  hipGraph_t g = nullptr;
  hipError_t e = hipStreamEndCapture(ds::as_stream(stream), &g);
  if (e != hipSuccess) return ds::hip_fail(e, "hipStreamEndCapture");
  if (node_count) {
    size_t n = 0;
    (void)hipGraphGetNodes(g, nullptr, &n);
    *node_count = (int)n;
  }
  hipGraphExec_t x = nullptr;
  e = hipGraphInstantiate(&x, g, nullptr, nullptr, 0);
  if (e != hipSuccess) {
    (void)hipGraphDestroy(g);
    return ds::hip_fail(e, "hipGraphInstantiate");
  }
  ds_graph* h = new ds_graph;
  h->graph = g;
  h->exec = x;
  *out_graph = h;
  return DS_OK;
}

int ds_graph_launch(ds_graph* g, void* stream) {
  DS_REQUIRE(g != nullptr && g->exec != nullptr, DS_ERR_NULL, "ds_graph_launch: graph is NULL");
  hipError_t e = hipGraphLaunch(g->exec, ds::as_stream(stream));
  if (e != hipSuccess) return ds::hip_fail(e, "hipGraphLaunch");
  return DS_OK;
}

int ds_graph_destroy(ds_graph* g) {
  if (!g) return DS_OK;
  if (g->exec) (void)hipGraphExecDestroy(g->exec);
  if (g->graph) (void)hipGraphDestroy(g->graph);
  delete g;
  return DS_OK;
}

}  // extern "C"
