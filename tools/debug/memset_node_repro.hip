// Do hipGraph MEMSET nodes between kernel nodes replay reliably on this HIP version?  ("No memset nodes";
// the cause of the wrong replays of round 3's 3-D loop was not established.)
//
// The pattern of that loop, reduced: one captured stream holds K rounds of
//     memset(slot[0..S), 0)  ->  k_max: atomicMax(slot[s], data_r[s*n..])  ->  k_use: out_r[s] = slot[s] (+ check value)
// on the SAME slots every round, as the network zeroes one amax arena per evaluation.  The graph is replayed R times; after every
// replay all K x S results are compared with the host's.  Two builds of the capture: the memset as hipMemsetD32Async (a MEMSET
// node) and as a one-line fill kernel (what the library does since round 3).  The tool prints the node / edge structure of both
// graphs and every mismatch with the round and replay it occurred in.
//   hipcc --offload-arch=gfx950 -O2 tools/debug/memset_node_repro.hip -o tools/bin/memset_node_repro && tools/bin/memset_node_repro
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void k_fill(unsigned* p, unsigned v, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = v; }
__global__ void k_max(unsigned* slot, const unsigned* data, int n) {            // grid (blocks, S): slot[s] = max over data[s*n ..]
  const unsigned* d = data + (size_t)blockIdx.y * n;
  unsigned m = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) m = d[i] > m ? d[i] : m;
  for (int o = 32; o > 0; o >>= 1) { unsigned t = __shfl_xor(m, o, 64); m = t > m ? t : m; }
  if ((threadIdx.x & 63) == 0) atomicMax(slot + blockIdx.y, m);
}
__global__ void k_use(unsigned* out, const unsigned* slot, int S) { int s = blockIdx.x * blockDim.x + threadIdx.x; if (s < S) out[s] = slot[s]; }

static const char* node_type(hipGraphNodeType t) {
  switch (t) {
    case hipGraphNodeTypeKernel: return "kernel";
    case hipGraphNodeTypeMemset: return "memset";
    case hipGraphNodeTypeMemcpy: return "memcpy";
    case hipGraphNodeTypeEmpty: return "empty";
    default: return "other";
  }
}

int main(int argc, char** argv) {
  const int K = argc > 1 ? atoi(argv[1]) : 40, S = 64, n = 1 << 16, R = argc > 2 ? atoi(argv[2]) : 12;
  std::vector<unsigned> h((size_t)K * S * n);
  unsigned seed = 12345u;
  for (auto& v : h) { seed = seed * 1664525u + 1013904223u; v = seed >> 9; }
  // make the maxima DEcrease from round to round: a slot that is not zeroed in time keeps the previous round's larger value
  std::vector<unsigned> want((size_t)K * S);
  for (int r = 0; r < K; ++r)
    for (int s = 0; s < S; ++s) {
      unsigned* d = &h[((size_t)r * S + s) * n];
      unsigned m = 0;
      for (int i = 0; i < n; ++i) { d[i] = d[i] >> (r / 4); m = d[i] > m ? d[i] : m; }
      want[(size_t)r * S + s] = m;
    }
  unsigned *data, *slot, *out;
  CK(hipMalloc(&data, h.size() * 4)); CK(hipMalloc(&slot, S * 4)); CK(hipMalloc(&out, (size_t)K * S * 4));
  CK(hipMemcpy(data, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  int total_bad = 0;
  for (int variant = 0; variant < 2; ++variant) {
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int r = 0; r < K; ++r) {
      if (variant == 0) CK(hipMemsetD32Async((hipDeviceptr_t)slot, 0, S, st));
      else hipLaunchKernelGGL(k_fill, dim3(1), dim3(64), 0, st, slot, 0u, S);
      hipLaunchKernelGGL(k_max, dim3(8, S), dim3(256), 0, st, slot, data + (size_t)r * S * n, n);
      hipLaunchKernelGGL(k_use, dim3(1), dim3(64), 0, st, out + (size_t)r * S, slot, S);
    }
    hipGraph_t g;
    CK(hipStreamEndCapture(st, &g));
    size_t nn = 0, ne = 0;
    CK(hipGraphGetNodes(g, nullptr, &nn));
    std::vector<hipGraphNode_t> nodes(nn);
    CK(hipGraphGetNodes(g, nodes.data(), &nn));
    CK(hipGraphGetEdges(g, nullptr, nullptr, &ne));
    std::vector<hipGraphNode_t> from(ne), to(ne);
    CK(hipGraphGetEdges(g, from.data(), to.data(), &ne));
    int counts[5] = {0, 0, 0, 0, 0};
    std::vector<int> indeg(nn, 0), outdeg(nn, 0);
    for (size_t i = 0; i < nn; ++i) {
      hipGraphNodeType t;
      CK(hipGraphNodeGetType(nodes[i], &t));
      counts[t == hipGraphNodeTypeKernel ? 0 : t == hipGraphNodeTypeMemset ? 1 : t == hipGraphNodeTypeMemcpy ? 2 : t == hipGraphNodeTypeEmpty ? 3 : 4]++;
    }
    for (size_t e = 0; e < ne; ++e)
      for (size_t i = 0; i < nn; ++i) { if (nodes[i] == from[e]) outdeg[i]++; if (nodes[i] == to[e]) indeg[i]++; }
    int chain = 1;
    for (size_t i = 0; i < nn; ++i) if (indeg[i] > 1 || outdeg[i] > 1) chain = 0;
    printf("variant %d (%s): %zu nodes (%d kernel, %d memset, %d other), %zu edges, %s\n", variant,
           variant == 0 ? "hipMemsetD32Async -> MEMSET nodes" : "fill kernel", nn, counts[0], counts[1], counts[2] + counts[3] + counts[4], ne,
           chain && ne + 1 == nn ? "one linear chain (every node has exactly its stream predecessor)" : "NOT a linear chain");
    if (nn <= 12)
      for (size_t e = 0; e < ne; ++e) {
        hipGraphNodeType a, b;
        hipGraphNodeGetType(from[e], &a); hipGraphNodeGetType(to[e], &b);
        printf("    edge %s -> %s\n", node_type(a), node_type(b));
      }
    hipGraphExec_t ex;
    CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    std::vector<unsigned> got((size_t)K * S);
    int bad_variant = 0;
    for (int rep = 0; rep < R; ++rep) {
      CK(hipMemsetAsync(out, 0xff, (size_t)K * S * 4, st));
      CK(hipMemsetAsync(slot, 0x7f, S * 4, st));             // garbage in the slots before each replay: the graph's own zeroing must remove it
      CK(hipGraphLaunch(ex, st));
      CK(hipStreamSynchronize(st));
      CK(hipMemcpy(got.data(), out, got.size() * 4, hipMemcpyDeviceToHost));
      int bad = 0, first_r = -1, first_s = -1;
      for (int r = 0; r < K; ++r)
        for (int s = 0; s < S; ++s)
          if (got[(size_t)r * S + s] != want[(size_t)r * S + s]) { if (!bad) { first_r = r; first_s = s; } ++bad; }
      if (bad) printf("  replay %2d: %d of %d results wrong; first at round %d slot %d: got %u, want %u (previous round's maximum: %u)\n", rep, bad, K * S,
                      first_r, first_s, got[(size_t)first_r * S + first_s], want[(size_t)first_r * S + first_s],
                      first_r ? want[(size_t)(first_r - 1) * S + first_s] : 0u);
      bad_variant += bad;
    }
    printf("  %s: %d wrong results over %d replays of %d rounds x %d slots\n", bad_variant ? "WRONG REPLAYS" : "every replay correct", bad_variant, R, K, S);
    total_bad += variant == 1 ? bad_variant : 0;
    hipGraphExecDestroy(ex);
    hipGraphDestroy(g);
  }
  return total_bad ? 1 : 0;              // only the fill-kernel form has to be right: it is the one the library uses
}
