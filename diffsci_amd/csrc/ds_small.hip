// Small per-evaluation pieces of the score network: the noise-level embedding and its MLPs.
// These touch a few KB per launch (the embedding depends only on sigma, which is shared by
// the whole batch during sampling -- schedulers.py:254), so they are written for clarity and
// exactness, not for a roofline: fp32 FMA-free dot products reduced across a 64-lane wave.
#include "ds_common.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// one wave per output column n; the wave walks all rows m
__global__ __launch_bounds__(256) void k_linear(float* y, const float* __restrict__ x, const float* __restrict__ w,
                                                const float* __restrict__ b, int M, int K, int N, int act) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const float* wr = w + (size_t)n * K;
  const float bn = b ? b[n] : 0.f;
  for (int m = blockIdx.y; m < M; m += gridDim.y) {
    const float* xr = x + (size_t)m * K;
    float acc = 0.f;
    for (int k = lane; k < K; k += 64) acc += xr[k] * wr[k];
    acc = wave_sum(acc);
    if (lane == 0) {
      float v = acc + bn;
      if (act == 1) v = v / (1.0f + expf(-v));
      else if (act == 2) v = fmaxf(v, 0.f);
      y[(size_t)m * N + n] = v;
    }
  }
}

__global__ void k_fourier(float* out, const float* __restrict__ t, const float* __restrict__ W,
                          const float* __restrict__ add, int add_rows, int M, int half) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * half) return;
  const int m = i / half, j = i - m * half;
  // 2*math.pi*x*W (commonlayers.py:186): the python double 2*pi enters the fp32 multiply as 6.2831855f
  const float proj = (t[m] * 6.283185307179586f) * W[j];
  float s = (float)sin((double)proj);
  float c = (float)cos((double)proj);
  if (add) {
    const float* ar = add + (size_t)(add_rows == 1 ? 0 : m) * 2 * half;
    s = s + ar[j];
    c = c + ar[half + j];
  }
  out[(size_t)m * 2 * half + j] = s;
  out[(size_t)m * 2 * half + half + j] = c;
}

}  // namespace

extern "C" {

int ds_linear(float* y, const float* x, const float* w, const float* b, int M, int K, int N, int act, void* stream) {
  DS_REQUIRE(y && x && w, DS_ERR_NULL, "ds_linear: NULL pointer");
  DS_REQUIRE(M >= 0 && K > 0 && N > 0, DS_ERR_SHAPE, "ds_linear: bad shape M=%d K=%d N=%d", M, K, N);
  DS_REQUIRE(act >= 0 && act <= 2, DS_ERR_UNSUPPORTED, "ds_linear: act %d", act);
  if (M == 0) return DS_OK;
  dim3 g((N + 3) / 4, M < 64 ? M : 64);
  hipLaunchKernelGGL(k_linear, g, dim3(256), 0, ds::as_stream(stream), y, x, w, b, M, K, N, act);
  DS_CHECK_LAUNCH("ds_linear");
  return DS_OK;
}

int ds_fourier_features(float* out, const float* t, const float* W, const float* add, int add_rows, int M, int half,
                        void* stream) {
  DS_REQUIRE(out && t && W, DS_ERR_NULL, "ds_fourier_features: NULL pointer");
  DS_REQUIRE(M >= 0 && half > 0, DS_ERR_SHAPE, "ds_fourier_features: bad shape M=%d half=%d", M, half);
  DS_REQUIRE(add == nullptr || add_rows == 1 || add_rows == M, DS_ERR_SHAPE,
             "ds_fourier_features: add_rows=%d must be 1 or M=%d", add_rows, M);
  if (M == 0) return DS_OK;
  const int total = M * half;
  hipLaunchKernelGGL(k_fourier, dim3((total + 255) / 256), dim3(256), 0, ds::as_stream(stream), out, t, W, add,
                     add_rows, M, half);
  DS_CHECK_LAUNCH("ds_fourier_features");
  return DS_OK;
}

}  // extern "C"
