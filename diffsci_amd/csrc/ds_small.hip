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

// ConvolutionalFourierProjection (commonlayers.py:229-255): xc[b,d,p] = sum_c x[b,c,p] * (2*pi*W[c,d]); out = cat[sin(xc), cos(xc)]
// along channels.  One thread per (b, d, p), p fastest: coalesced reads of the C input planes and of both output planes.
__global__ __launch_bounds__(256) void k_fourier_channels(float* __restrict__ out, const float* __restrict__ x,
                                                          const float* __restrict__ W, int C, int D, size_t HW, size_t total) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < total; i += stride) {
    const size_t p = i % HW;
    const size_t bd = i / HW;
    const int d = (int)(bd % D);
    const size_t b = bd / D;
    const float* xb = x + b * C * HW + p;
    float acc = 0.f;
    for (int c = 0; c < C; ++c) acc = acc + xb[(size_t)c * HW] * (W[(size_t)c * D + d] * 6.283185307179586f);
    float sn, cs;
    sincosf(acc, &sn, &cs);
    float* ob = out + b * 2 * D * HW + p;
    ob[(size_t)d * HW] = sn;
    ob[(size_t)(D + d) * HW] = cs;
  }
}

}  // namespace

extern "C" {

int ds_fourier_channels(float* out, const float* x, const float* W, int B, int C, int D, size_t HW, void* stream) {
  DS_REQUIRE(out && x && W, DS_ERR_NULL, "ds_fourier_channels: NULL pointer");
  DS_REQUIRE(B >= 0 && C > 0 && D > 0 && HW > 0, DS_ERR_SHAPE, "ds_fourier_channels: bad shape B=%d C=%d D=%d", B, C, D);
  const size_t total = (size_t)B * D * HW;
  if (total == 0) return DS_OK;
  size_t g = (total + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(k_fourier_channels, dim3((unsigned)g), dim3(256), 0, ds::as_stream(stream), out, x, W, C, D, HW, total);
  DS_CHECK_LAUNCH("ds_fourier_channels");
  return DS_OK;
}

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

// ---------------------------------------------------------------------------------------------
// Generic single-head attention for sequence lengths the MFMA kernels do not take (L not a multiple
// of 32: odd or tiny bottlenecks such as 4x4 or 5x7).  One wave per (sample, query), exact fp32 FMA
// chains, softmax in fp32 -- a correctness path: these shapes are a few thousand MACs per query.
//   qkv [B, 3E, L] channel-major, out [B, E, L]   (same contract as ds_attention)
namespace {

__global__ __launch_bounds__(64) void k_attn_generic(float* out, const float* __restrict__ qkv, int E, int L, float scale) {
  extern __shared__ float sm[];            // [E] scaled query, then [L] probabilities
  float* qs = sm;
  float* ps = sm + E;
  const int q = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
  const float* Qt = qkv + (size_t)b * 3 * E * L;
  const float* Kt = Qt + (size_t)E * L;
  const float* Vt = Kt + (size_t)E * L;
  for (int d = lane; d < E; d += 64) qs[d] = Qt[(size_t)d * L + q] * scale;
  __syncthreads();
  float mx = -INFINITY;
  for (int k = lane; k < L; k += 64) {
    float s = 0.f;
    for (int d = 0; d < E; ++d) s = fmaf(qs[d], Kt[(size_t)d * L + k], s);
    ps[k] = s;
    mx = fmaxf(mx, s);
  }
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  float sum = 0.f;
  for (int k = lane; k < L; k += 64) {
    const float p = expf(ps[k] - mx);
    ps[k] = p;
    sum += p;
  }
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
  __syncthreads();
  const float inv = 1.0f / sum;
  for (int d = lane; d < E; d += 64) {
    const float* v = Vt + (size_t)d * L;
    float acc = 0.f;
    for (int k = 0; k < L; ++k) acc = fmaf(ps[k], v[k], acc);
    out[((size_t)b * E + d) * L + q] = acc * inv;
  }
}

}  // namespace

extern "C" int ds_attention_generic(float* out, const float* qkv, int B, int E, int L, void* stream) {
  DS_REQUIRE(out && qkv, DS_ERR_NULL, "ds_attention_generic: NULL pointer");
  DS_REQUIRE(B >= 0 && E > 0 && L > 0, DS_ERR_SHAPE, "ds_attention_generic: bad shape B=%d E=%d L=%d", B, E, L);
  DS_REQUIRE(B < 65536, DS_ERR_SHAPE, "ds_attention_generic: B=%d exceeds grid.y", B);
  const size_t lds = (size_t)(E + L) * sizeof(float);
  DS_REQUIRE(lds <= 64 * 1024, DS_ERR_UNSUPPORTED, "ds_attention_generic: E + L = %d is too large for the generic path", E + L);
  if (B == 0) return DS_OK;
  const float scale = (float)sqrt(1.0 / (double)E);
  hipLaunchKernelGGL(k_attn_generic, dim3(L, B), dim3(64), lds, ds::as_stream(stream), out, qkv, E, L, scale);
  DS_CHECK_LAUNCH("ds_attention_generic");
  return DS_OK;
}

// ---- cosine attention (attention.py:300-372): queries and keys are L2-normalised per token before the product ----
namespace {

// x: [B, Ctot, L] channel-major; channels [c0, c0 + C) of every token are divided by (their L2 norm + eps) and
// multiplied by `gain`.  One thread per token: loads are coalesced along L; the C values are read twice (L2 hits).
__global__ __launch_bounds__(256) void k_token_l2norm(float* x, int Ctot, int c0, int C, int L, float eps, float gain) {
  const int l = blockIdx.x * 256 + threadIdx.x;
  if (l >= L) return;
  float* p = x + ((size_t)blockIdx.y * Ctot + c0) * L + l;
  float ss = 0.f;
  for (int c = 0; c < C; ++c) {
    const float v = p[(size_t)c * L];
    ss += v * v;
  }
  const float den = sqrtf(ss) + eps;
  for (int c = 0; c < C; ++c) {
    float v = p[(size_t)c * L] / den;
    p[(size_t)c * L] = v * gain;
  }
}

}  // namespace

extern "C" int ds_token_l2_normalize(float* x, int B, int Ctot, int c0, int C, int L, float eps, float gain, void* stream) {
  DS_REQUIRE(x, DS_ERR_NULL, "ds_token_l2_normalize: NULL pointer");
  DS_REQUIRE(B >= 0 && Ctot > 0 && c0 >= 0 && C > 0 && c0 + C <= Ctot && L > 0, DS_ERR_SHAPE,
             "ds_token_l2_normalize: bad shape B=%d Ctot=%d c0=%d C=%d L=%d", B, Ctot, c0, C, L);
  DS_REQUIRE(B < 65536, DS_ERR_SHAPE, "ds_token_l2_normalize: B must stay below 65536");
  if (B == 0) return DS_OK;
  hipLaunchKernelGGL(k_token_l2norm, dim3((unsigned)((L + 255) / 256), (unsigned)B), dim3(256), 0, ds::as_stream(stream), x,
                     Ctot, c0, C, L, eps, gain);
  DS_CHECK_LAUNCH("ds_token_l2_normalize");
  return DS_OK;
}
