// Per-sample max |x| of a tensor, as float bits: the input side of the fp16x3 kernels' activation exponent
// (ds_conv_epilogue.h: act_scale).  The convolution / attention epilogues leave this number for their own outputs
// (out_amax); these entry points serve tensors that come from elsewhere -- the sampler's c_in * x, user fields
// concatenated by PUNetGCond (punetg.py:719-735), an extra_residual module's output, slice copies of volumes.
// HBM-bound: one read pass, 4 B/elt.
#include "ds_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int NT = 256;

__device__ __forceinline__ float abs_max4(f32x4 v) {
  return fmaxf(fmaxf(__builtin_fabsf(v.x), __builtin_fabsf(v.y)), fmaxf(__builtin_fabsf(v.z), __builtin_fabsf(v.w)));
}

__device__ __forceinline__ void block_commit(unsigned* slot, float m) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(slot, __builtin_bit_cast(unsigned, m));   // non-negative floats order like their bits
}

// grid (blocks per row, rows); fmaxf drops NaNs, so a row of NaNs reports 0 (the consuming kernel then runs unscaled
// and the NaNs propagate through its arithmetic as they would anyway)
__global__ __launch_bounds__(NT) void k_absmax_rows(unsigned* __restrict__ out, const float* __restrict__ x, size_t n_per_row,
                                                    size_t row_stride) {
  const float* row = x + (size_t)blockIdx.y * row_stride;
  float m = 0.f;
  if (((reinterpret_cast<uintptr_t>(row) | (n_per_row * 4)) & 15u) == 0) {
    const f32x4* r4 = reinterpret_cast<const f32x4*>(row);
    const size_t n4 = n_per_row / 4;
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n4; i += (size_t)gridDim.x * NT) {
      const f32x4 v = r4[i];
      m = fmaxf(m, fmaxf(fmaxf(__builtin_fabsf(v.x), __builtin_fabsf(v.y)), fmaxf(__builtin_fabsf(v.z), __builtin_fabsf(v.w))));
    }
  } else {
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n_per_row; i += (size_t)gridDim.x * NT) m = fmaxf(m, __builtin_fabsf(row[i]));
  }
  block_commit(out + blockIdx.y, m);
}

__global__ void k_fill_u32(unsigned* __restrict__ p, unsigned value, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = value;
}

__global__ void k_amax_merge(unsigned* __restrict__ out, const unsigned* __restrict__ a, const unsigned* __restrict__ b, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    unsigned u = out[i];
    const unsigned v = a[i], w = b ? b[i] : 0u;
    u = u > v ? u : v;
    out[i] = u > w ? u : w;
  }
}

// out[b] = max(out[b], m) with m = max_c s[b, c]; *flag |= 1 when one exponent per sample cannot serve the input layer:
// under it an element carries an absolute error of 2^-38 m (ds_conv_epilogue.h), i.e. channel c contributes an error of
// ~2^-36 m wmax[c] to an output of scale P = max_c' s[b, c'] wmax[c'] (wmax[c]: the layer's largest |weight| on input channel c).
// That stays below fp32's own accumulation noise unless m wmax[c] > 2^gap P for a channel that carries data: small channels
// with ordinary weights (a 1e-8 field next to c_in x) are harmless, the same field with compensating 1e8 weights is not.
// Without wmax: any non-zero channel more than `gap` binades below m.
__global__ void k_amax_channels(unsigned* __restrict__ out, unsigned* __restrict__ flag, const unsigned* __restrict__ s,
                                const float* __restrict__ wmax, int B, int C, int gap) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  unsigned m = 0;
  double P = 0.0;
  for (int c = 0; c < C; ++c) {
    const unsigned v = s[(size_t)b * C + c];
    m = v > m ? v : m;
    if (wmax) { const double p = (double)__builtin_bit_cast(float, v) * (double)wmax[c]; P = p > P ? p : P; }
  }
  const unsigned u = out[b];
  out[b] = u > m ? u : m;
  const int em = (int)((m >> 23) & 0xffu);
  if (em == 0 || em == 255 || !flag) return;
  bool bad = false;
  const double md = (double)__builtin_bit_cast(float, m), lim = ldexp(P, gap);
  for (int c = 0; c < C; ++c) {
    const unsigned v = s[(size_t)b * C + c];
    if (v == 0) continue;
    if (wmax ? (md * (double)wmax[c] > lim) : ((int)((v >> 23) & 0xffu) < em - gap)) bad = true;
  }
  if (bad) atomicOr(flag, 1u);
}

// One launch for a network input of moderate size (C <= 64): block b reduces sample b's C planes (per-channel maxima in LDS, no
// global atomics), applies ds_absmax_channels' criterion, zeroes column b of the forward's amax arena [arena_rows][B] and writes
// the sample's maximum into arena row `out_row` -- in place of a fill, a reduction and a combine launch per evaluation.
constexpr int NTI = 1024;   // one workgroup has to pull a whole sample: 16 waves with four 16-byte loads in flight each
__global__ __launch_bounds__(NTI) void k_input_amax(unsigned* __restrict__ arena, int arena_rows, int out_row, unsigned* __restrict__ flag,
                                                    const float* __restrict__ x, const float* __restrict__ wmax, int B, int C,
                                                    size_t HW, int gap) {
  __shared__ unsigned cmax[64];
  const int b = blockIdx.x;
  for (int r = threadIdx.x; r < arena_rows; r += NTI)
    if (r != out_row) arena[(size_t)r * B + b] = 0u;
  if (threadIdx.x < 64) cmax[threadIdx.x] = 0u;
  __syncthreads();
  for (int c = 0; c < C; ++c) {
    const float* row = x + ((size_t)b * C + c) * HW;
    float v = 0.f;
    if (((reinterpret_cast<uintptr_t>(row) | (HW * 4)) & 15u) == 0) {
      const f32x4* r4 = reinterpret_cast<const f32x4*>(row);
      const size_t n4 = HW / 4;
      size_t i = threadIdx.x;
      for (; i + 3 * NTI < n4; i += 4 * NTI) {
        const f32x4 t0 = r4[i], t1 = r4[i + NTI], t2 = r4[i + 2 * NTI], t3 = r4[i + 3 * NTI];
        v = fmaxf(v, fmaxf(fmaxf(abs_max4(t0), abs_max4(t1)), fmaxf(abs_max4(t2), abs_max4(t3))));
      }
      for (; i < n4; i += NTI) v = fmaxf(v, abs_max4(r4[i]));
    } else {
      for (size_t i = threadIdx.x; i < HW; i += NTI) v = fmaxf(v, __builtin_fabsf(row[i]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(&cmax[c], __builtin_bit_cast(unsigned, v));      // LDS: the block's sixteen waves
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  unsigned m = 0;
  double P = 0.0;
  for (int c = 0; c < C; ++c) {
    const unsigned v = cmax[c];
    m = v > m ? v : m;
    if (wmax) { const double p = (double)__builtin_bit_cast(float, v) * (double)wmax[c]; P = p > P ? p : P; }
  }
  arena[(size_t)out_row * B + b] = m;
  const int em = (int)((m >> 23) & 0xffu);
  if (em == 0 || em == 255 || !flag) return;
  bool bad = false;
  const double md = (double)__builtin_bit_cast(float, m), lim = ldexp(P, gap);
  for (int c = 0; c < C; ++c) {
    const unsigned v = cmax[c];
    if (v == 0) continue;
    if (wmax ? (md * (double)wmax[c] > lim) : ((int)((v >> 23) & 0xffu) < em - gap)) bad = true;
  }
  if (bad) atomicOr(flag, 1u);
}

}  // namespace

extern "C" {

int ds_fill_u32(unsigned* p, unsigned value, size_t n, void* stream) {
  DS_REQUIRE(p || n == 0, DS_ERR_NULL, "ds_fill_u32: NULL pointer");
  if (n == 0) return DS_OK;
  // a kernel, not hipMemsetD32Async: inside a captured graph the memset NODE of this HIP version is not ordered reliably against
  // its neighbouring kernel nodes (observed on MI355X / ROCm 7.2: replays of a graph with memset nodes between kernels read slots
  // zeroed too late or not at all; the same sequence with this kernel replays bit-identically)
  DS_REQUIRE(n < ((size_t)1 << 31), DS_ERR_SHAPE, "ds_fill_u32: n too large");
  hipLaunchKernelGGL(k_fill_u32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ds::as_stream(stream), p, value, n);
  DS_CHECK_LAUNCH("ds_fill_u32");
  return DS_OK;
}

int ds_absmax_rows(unsigned* out, const float* x, int rows, size_t n_per_row, size_t row_stride, void* stream) {
  DS_REQUIRE(out && x, DS_ERR_NULL, "ds_absmax_rows: NULL pointer");
  DS_REQUIRE(rows >= 0 && rows < 65536, DS_ERR_SHAPE, "ds_absmax_rows: rows=%d", rows);
  DS_REQUIRE(row_stride >= n_per_row, DS_ERR_SHAPE, "ds_absmax_rows: row_stride < n_per_row");
  if (rows == 0 || n_per_row == 0) return DS_OK;
  // enough workgroups to pull the HBM rate on large rows, one per row on small ones
  size_t per = (n_per_row + (size_t)NT * 16 - 1) / ((size_t)NT * 16);
  const size_t want = (size_t)2048 / (size_t)rows + 1;
  if (per > want) per = want;
  hipLaunchKernelGGL(k_absmax_rows, dim3((unsigned)per, (unsigned)rows), dim3(NT), 0, ds::as_stream(stream), out, x, n_per_row, row_stride);
  DS_CHECK_LAUNCH("ds_absmax_rows");
  return DS_OK;
}

int ds_absmax_channels(unsigned* out, unsigned* flag, unsigned* scratch, const float* x, const float* wmax, int B, int C, size_t HW,
                       int gap, void* stream) {
  DS_REQUIRE(out && scratch && x, DS_ERR_NULL, "ds_absmax_channels: NULL pointer");
  DS_REQUIRE(B >= 0 && C > 0 && (long long)B * C < 65536, DS_ERR_SHAPE, "ds_absmax_channels: B=%d C=%d", B, C);
  if (B == 0 || HW == 0) return DS_OK;
  const int rc = ds_absmax_rows(scratch, x, B * C, HW, HW, stream);
  if (rc != DS_OK) return rc;
  hipLaunchKernelGGL(k_amax_channels, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, ds::as_stream(stream), out, flag, scratch, wmax, B, C, gap);
  DS_CHECK_LAUNCH("ds_absmax_channels");
  return DS_OK;
}

int ds_input_amax(unsigned* arena, int arena_rows, int out_row, unsigned* flag, const float* x, const float* wmax, int B, int C,
                  size_t HW, int gap, void* stream) {
  DS_REQUIRE(arena && x, DS_ERR_NULL, "ds_input_amax: NULL pointer");
  DS_REQUIRE(B >= 0 && B < 65536 && C > 0 && C <= 64 && arena_rows > 0 && out_row >= 0 && out_row < arena_rows, DS_ERR_SHAPE,
             "ds_input_amax: B=%d C=%d rows=%d out_row=%d", B, C, arena_rows, out_row);
  if (B == 0) return DS_OK;
  hipLaunchKernelGGL(k_input_amax, dim3((unsigned)B), dim3(NTI), 0, ds::as_stream(stream), arena, arena_rows, out_row, flag, x, wmax, B, C,
                     HW, gap);
  DS_CHECK_LAUNCH("ds_input_amax");
  return DS_OK;
}

int ds_amax_merge(unsigned* out, const unsigned* a, const unsigned* b, int n, void* stream) {
  DS_REQUIRE(out && a, DS_ERR_NULL, "ds_amax_merge: NULL pointer");
  if (n <= 0) return DS_OK;
  hipLaunchKernelGGL(k_amax_merge, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ds::as_stream(stream), out, a, b, n);
  DS_CHECK_LAUNCH("ds_amax_merge");
  return DS_OK;
}

}  // extern "C"
