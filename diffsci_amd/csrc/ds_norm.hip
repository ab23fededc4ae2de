// Per-(sample, channel) instance normalisation fused with SiLU -- the two norms of
// ResnetBlockC (commonlayers.py:824, 829): GroupNorm(C, C) and GroupRMSNorm(C, C).
//
// HBM-bound: 8 bytes per element (one read, one write).  A plane (H*W floats, contiguous in
// NCHW) is held entirely in registers between the statistics pass and the apply pass:
//   - planes of <= 1024 floats: one 64-lane wave per plane, reductions by DPP/shuffle only;
//   - planes up to 64 Ki floats: one workgroup per plane (256 or 1024 threads, <= 16 float4 per
//     lane), wave shuffles + one LDS exchange per statistic;
//   - anything else (H*W not a multiple of 4, or larger): a two-read fallback.
// Statistics are two-pass in registers (mean, then centred second moment), fp32.
#include "ds_common.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ float silu(float v) { return v / (1.0f + __expf(-v)); }

// exact-ish exp: use expf (ocml, <= 1 ulp) rather than the fast intrinsic
__device__ __forceinline__ float silu_precise(float v) { return v / (1.0f + expf(-v)); }

template <int KIND>
__device__ __forceinline__ float apply_one(float v, float mean, float scale_or_denom, float w, float b) {
  float y;
  if (KIND == 0) {
    y = (v - mean) * scale_or_denom * w + b;   // scale_or_denom = rstd
  } else {
    y = v / scale_or_denom * w + b;            // scale_or_denom = sqrt(mean(x^2)+eps), commonlayers.py:377-383
  }
  return silu_precise(y);
}

// ---- one wave per plane ---------------------------------------------------------------
template <int KIND, int VPT>
__global__ __launch_bounds__(256) void k_inorm_wave(float* out, const float* x, const float* __restrict__ w,
                                                    const float* __restrict__ b, int planes, int C, int hw4,
                                                    float inv_hw, float eps) {
  const int lane = threadIdx.x & 63;
  const int plane = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (plane >= planes) return;
  const float4* src = reinterpret_cast<const float4*>(x) + (size_t)plane * hw4;
  float4 v[VPT];
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    int idx = lane + 64 * i;
    v[i] = idx < hw4 ? src[idx] : make_float4(0, 0, 0, 0);
  }
  float mean = 0.f, sod;
  if (KIND == 0) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPT; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    mean = wave_sum(s) * inv_hw;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
      if (lane + 64 * i < hw4) {
        float a0 = v[i].x - mean, a1 = v[i].y - mean, a2 = v[i].z - mean, a3 = v[i].w - mean;
        q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
      }
    }
    float var = wave_sum(q) * inv_hw;
    sod = 1.0f / sqrtf(var + eps);
  } else {
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPT; ++i) q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
    sod = sqrtf(wave_sum(q) * inv_hw + eps);
  }
  const int c = plane % C;
  const float wc = w ? w[c] : 1.0f, bc = b ? b[c] : 0.0f;
  float4* dst = reinterpret_cast<float4*>(out) + (size_t)plane * hw4;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    int idx = lane + 64 * i;
    if (idx < hw4) {
      float4 o;
      o.x = apply_one<KIND>(v[i].x, mean, sod, wc, bc);
      o.y = apply_one<KIND>(v[i].y, mean, sod, wc, bc);
      o.z = apply_one<KIND>(v[i].z, mean, sod, wc, bc);
      o.w = apply_one<KIND>(v[i].w, mean, sod, wc, bc);
      dst[idx] = o;
    }
  }
}

// ---- one workgroup per plane ----------------------------------------------------------
template <int THREADS>
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int wid = threadIdx.x >> 6;
  __syncthreads();  // protect red[] reuse
  if ((threadIdx.x & 63) == 0) red[wid] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < THREADS / 64; ++i) t += red[i];
  return t;
}

template <int KIND, int THREADS, int VPT>
__global__ __launch_bounds__(THREADS) void k_inorm_block(float* out, const float* x, const float* __restrict__ w,
                                                         const float* __restrict__ b, int C, int hw4, float inv_hw,
                                                         float eps) {
  __shared__ float red[THREADS / 64];
  const int plane = blockIdx.x;
  const float4* src = reinterpret_cast<const float4*>(x) + (size_t)plane * hw4;
  float4 v[VPT];
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    int idx = threadIdx.x + THREADS * i;
    v[i] = idx < hw4 ? src[idx] : make_float4(0, 0, 0, 0);
  }
  float mean = 0.f, sod;
  if (KIND == 0) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPT; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    mean = block_sum<THREADS>(s, red) * inv_hw;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
      if (threadIdx.x + THREADS * i < hw4) {
        float a0 = v[i].x - mean, a1 = v[i].y - mean, a2 = v[i].z - mean, a3 = v[i].w - mean;
        q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
      }
    }
    float var = block_sum<THREADS>(q, red) * inv_hw;
    sod = 1.0f / sqrtf(var + eps);
  } else {
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPT; ++i) q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
    sod = sqrtf(block_sum<THREADS>(q, red) * inv_hw + eps);
  }
  const int c = plane % C;
  const float wc = w ? w[c] : 1.0f, bc = b ? b[c] : 0.0f;
  float4* dst = reinterpret_cast<float4*>(out) + (size_t)plane * hw4;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    int idx = threadIdx.x + THREADS * i;
    if (idx < hw4) {
      float4 o;
      o.x = apply_one<KIND>(v[i].x, mean, sod, wc, bc);
      o.y = apply_one<KIND>(v[i].y, mean, sod, wc, bc);
      o.z = apply_one<KIND>(v[i].z, mean, sod, wc, bc);
      o.w = apply_one<KIND>(v[i].w, mean, sod, wc, bc);
      dst[idx] = o;
    }
  }
}

// ---- fallback: any plane size, data re-read from memory -------------------------------
template <int KIND>
__global__ __launch_bounds__(256) void k_inorm_generic(float* out, const float* x, const float* __restrict__ w,
                                                       const float* __restrict__ b, int C, int hw, float inv_hw,
                                                       float eps) {
  __shared__ float red[4];
  const int plane = blockIdx.x;
  const float* src = x + (size_t)plane * hw;
  float mean = 0.f, sod;
  if (KIND == 0) {
    float s = 0.f;
    for (int i = threadIdx.x; i < hw; i += 256) s += src[i];
    mean = block_sum<256>(s, red) * inv_hw;
    float q = 0.f;
    for (int i = threadIdx.x; i < hw; i += 256) {
      float a = src[i] - mean;
      q += a * a;
    }
    sod = 1.0f / sqrtf(block_sum<256>(q, red) * inv_hw + eps);
  } else {
    float q = 0.f;
    for (int i = threadIdx.x; i < hw; i += 256) q += src[i] * src[i];
    sod = sqrtf(block_sum<256>(q, red) * inv_hw + eps);
  }
  const int c = plane % C;
  const float wc = w ? w[c] : 1.0f, bc = b ? b[c] : 0.0f;
  float* dst = out + (size_t)plane * hw;
  // out may alias x: each element is read (above and here) only by the thread that writes it
  // after the statistics are complete (block_sum ends with a barrier-ordered LDS read).
  __syncthreads();
  for (int i = threadIdx.x; i < hw; i += 256) dst[i] = apply_one<KIND>(src[i], mean, sod, wc, bc);
}

template <int KIND>
int launch_inorm(float* out, const float* x, const float* w, const float* b, int B, int C, int HW, float eps,
                 hipStream_t s) {
  const int planes = B * C;
  const float inv = 1.0f / (float)HW;
  const bool vec = (HW % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15u) == 0;
  const int hw4 = HW / 4;
  if (vec && hw4 <= 256) {
    dim3 g((planes + 3) / 4), t(256);
    if (hw4 <= 64) hipLaunchKernelGGL((k_inorm_wave<KIND, 1>), g, t, 0, s, out, x, w, b, planes, C, hw4, inv, eps);
    else if (hw4 <= 128) hipLaunchKernelGGL((k_inorm_wave<KIND, 2>), g, t, 0, s, out, x, w, b, planes, C, hw4, inv, eps);
    else hipLaunchKernelGGL((k_inorm_wave<KIND, 4>), g, t, 0, s, out, x, w, b, planes, C, hw4, inv, eps);
  } else if (vec && hw4 <= 4096) {
    dim3 g(planes), t(256);
    if (hw4 <= 512) hipLaunchKernelGGL((k_inorm_block<KIND, 256, 2>), g, t, 0, s, out, x, w, b, C, hw4, inv, eps);
    else if (hw4 <= 1024) hipLaunchKernelGGL((k_inorm_block<KIND, 256, 4>), g, t, 0, s, out, x, w, b, C, hw4, inv, eps);
    else if (hw4 <= 2048) hipLaunchKernelGGL((k_inorm_block<KIND, 256, 8>), g, t, 0, s, out, x, w, b, C, hw4, inv, eps);
    else hipLaunchKernelGGL((k_inorm_block<KIND, 256, 16>), g, t, 0, s, out, x, w, b, C, hw4, inv, eps);
  } else if (vec && hw4 <= 16384) {
    dim3 g(planes), t(1024);
    if (hw4 <= 8192) hipLaunchKernelGGL((k_inorm_block<KIND, 1024, 8>), g, t, 0, s, out, x, w, b, C, hw4, inv, eps);
    else hipLaunchKernelGGL((k_inorm_block<KIND, 1024, 16>), g, t, 0, s, out, x, w, b, C, hw4, inv, eps);
  } else {
    hipLaunchKernelGGL((k_inorm_generic<KIND>), dim3(planes), dim3(256), 0, s, out, x, w, b, C, HW, inv, eps);
  }
  DS_CHECK_LAUNCH("ds_inorm_silu");
  return DS_OK;
}

// kinds without a reduction: 2 = no normalisation (torch.nn.Identity: SiLU only), 3 = GroupPixNorm(C, C)
// (commonlayers.py:387-440 with one channel per group: x / sqrt(x^2 + eps) * w[c] + b[c]), then SiLU
template <int KIND>
__global__ __launch_bounds__(256) void k_pointwise_silu(float* out, const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ b, int C, int HW, float eps) {
  const int plane = blockIdx.y;
  const int c = plane % C;
  const float wc = (KIND == 3 && w) ? w[c] : 1.0f, bc = (KIND == 3 && b) ? b[c] : 0.0f;
  const float* xp = x + (size_t)plane * HW;
  float* op = out + (size_t)plane * HW;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
    float v = xp[i];
    if (KIND == 3) {
      v = v / sqrtf(v * v + eps);
      v = v * wc + bc;
    }
    op[i] = v / (1.0f + expf(-v));
  }
}

// ---- norm + SiLU written as the consuming convolution's pre-split fp16 hi / lo images (ds_conv2d_h3_img) ----------------------
// images [B][chunk][piece 2][half 2][H+2][W+2] vectors of 8 channels (c = 16 chunk + 8 half + 0..7), zero border, zero channels
// past C: the same 8 bytes per element as ds_inorm_silu moves, but the convolution then stages its patches by LDS-DMA and no
// workgroup splits anything (with four channel tiles and the halo every element used to be split 5.3 times).
// One workgroup of 8 waves per 8-channel group: wave k normalises channel k exactly as k_inorm_wave does (same loads, same
// summation order, same activation: bit-identical values), parks its plane in LDS, then all threads gather the 8 channels of a
// pixel, split and store two vectors.
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split2_img(float a, float b, unsigned& hi, unsigned& lo) {      // ds_h3_common.h's split2
  f16x2_t h = {(_Float16)a, (_Float16)b};
  unsigned hp = __builtin_bit_cast(unsigned, h);
  asm volatile("" : "+v"(hp));
  const f16x2_t hq = __builtin_bit_cast(f16x2_t, hp);
  f16x2_t l = {(_Float16)(a - (float)hq[0]), (_Float16)(b - (float)hq[1])};
  hi = hp;
  lo = __builtin_bit_cast(unsigned, l);
}

template <int KIND, int VPT>
__global__ __launch_bounds__(512) void k_inorm_images(u32x4_t* __restrict__ img, const float* __restrict__ x,
                                                     const float* __restrict__ w, const float* __restrict__ b, int C, int nchunk,
                                                     int H, int W, int hw4, float inv_hw, float eps) {
  __shared__ __attribute__((aligned(16))) float act[8][1024];
  const int lane = threadIdx.x & 63, k = threadIdx.x >> 6;
  const int g = blockIdx.x;
  const int h = g & 1, chunk = (g >> 1) % nchunk, bb = (g >> 1) / nchunk;
  const int c = 16 * chunk + 8 * h + k;
  float4 v[VPT];
  if (c < C) {
    const size_t plane = (size_t)bb * C + c;
    const float4* src = reinterpret_cast<const float4*>(x) + plane * hw4;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
      int idx = lane + 64 * i;
      v[i] = idx < hw4 ? src[idx] : make_float4(0, 0, 0, 0);
    }
    float mean = 0.f, sod;
    if (KIND == 0) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < VPT; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
      mean = wave_sum(s) * inv_hw;
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < VPT; ++i) {
        if (lane + 64 * i < hw4) {
          float a0 = v[i].x - mean, a1 = v[i].y - mean, a2 = v[i].z - mean, a3 = v[i].w - mean;
          q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
        }
      }
      float var = wave_sum(q) * inv_hw;
      sod = 1.0f / sqrtf(var + eps);
    } else {
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < VPT; ++i) q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
      sod = sqrtf(wave_sum(q) * inv_hw + eps);
    }
    const float wc = w ? w[c] : 1.0f, bc = b ? b[c] : 0.0f;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {                              // activated in place: the plane stays in registers
      v[i].x = apply_one<KIND>(v[i].x, mean, sod, wc, bc);
      v[i].y = apply_one<KIND>(v[i].y, mean, sod, wc, bc);
      v[i].z = apply_one<KIND>(v[i].z, mean, sod, wc, bc);
      v[i].w = apply_one<KIND>(v[i].w, mean, sod, wc, bc);
    }
  } else {
#pragma unroll
    for (int i = 0; i < VPT; ++i) v[i] = make_float4(0, 0, 0, 0);
  }
  const int Hp = H + 2, Wp = W + 2, HW = 4 * hw4;
  u32x4_t* hi_img = img + ((((size_t)bb * nchunk + chunk) * 2 + 0) * 2 + h) * Hp * Wp;
  u32x4_t* lo_img = img + ((((size_t)bb * nchunk + chunk) * 2 + 1) * 2 + h) * Hp * Wp;
  // slabs of 1024 pixels (= 256 float4 = registers 4 s .. 4 s + 3 of every lane) through LDS: gather the 8 channels of a
  // pixel, split, store two vectors
  constexpr int NSLAB = (VPT + 3) / 4;
#pragma unroll
  for (int sl = 0; sl < NSLAB; ++sl) {
    if (sl > 0) __syncthreads();
#pragma unroll
    for (int i = 4 * sl; i < 4 * sl + 4 && i < VPT; ++i) {
      const int idx = lane + 64 * i;
      if (idx < hw4) reinterpret_cast<float4*>(act[k])[idx - 256 * sl] = v[i];
    }
    __syncthreads();
    for (int j = threadIdx.x; j < 1024; j += 512) {
      const int i = 1024 * sl + j;
      if (i < HW) {
        const int y = i / W, xx = i - y * W;
        u32x4_t qh, ql;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          unsigned ph, pl;
          split2_img(act[2 * t][j], act[2 * t + 1][j], ph, pl);
          qh[t] = ph; ql[t] = pl;
        }
        const size_t o = (size_t)(y + 1) * Wp + xx + 1;
        hi_img[o] = qh;
        lo_img[o] = ql;
      }
    }
  }
  const int nb = 2 * Wp + 2 * H;                                   // the zero border (the convolution's padding)
  const u32x4_t z = {0u, 0u, 0u, 0u};
  for (int j = threadIdx.x; j < nb; j += 512) {
    size_t o;
    if (j < Wp) o = j;
    else if (j < 2 * Wp) o = (size_t)(Hp - 1) * Wp + (j - Wp);
    else { const int r = j - 2 * Wp; o = (size_t)(1 + (r >> 1)) * Wp + ((r & 1) ? Wp - 1 : 0); }
    hi_img[o] = z;
    lo_img[o] = z;
  }
}

}  // namespace

extern "C" int ds_inorm_silu(float* out, const float* x, const float* w, const float* b, int B, int C, int HW,
                             float eps, int kind, void* stream) {
  DS_REQUIRE(out && x, DS_ERR_NULL, "ds_inorm_silu: NULL pointer");
  DS_REQUIRE(B >= 0 && C > 0 && HW > 0, DS_ERR_SHAPE, "ds_inorm_silu: bad shape B=%d C=%d HW=%d", B, C, HW);
  DS_REQUIRE(kind >= 0 && kind <= 3, DS_ERR_UNSUPPORTED,
             "ds_inorm_silu: kind must be 0 (GroupLN), 1 (GroupRMS), 2 (none) or 3 (GroupPix)");
  DS_REQUIRE((long long)B * C < (1ll << 31), DS_ERR_SHAPE, "ds_inorm_silu: too many planes");
  if (B == 0) return DS_OK;
  hipStream_t s = ds::as_stream(stream);
  if (kind >= 2) {
    const int planes = B * C;
    DS_REQUIRE(planes < 65536, DS_ERR_SHAPE, "ds_inorm_silu: kind %d supports B*C < 65536 planes", kind);
    dim3 g((unsigned)((HW + 1023) / 1024 > 64 ? 64 : (HW + 1023) / 1024), (unsigned)planes);
    if (kind == 2) hipLaunchKernelGGL((k_pointwise_silu<2>), g, dim3(256), 0, s, out, x, w, b, C, HW, eps);
    else hipLaunchKernelGGL((k_pointwise_silu<3>), g, dim3(256), 0, s, out, x, w, b, C, HW, eps);
    DS_CHECK_LAUNCH("ds_inorm_silu");
    return DS_OK;
  }
  return kind == 0 ? launch_inorm<0>(out, x, w, b, B, C, HW, eps, s) : launch_inorm<1>(out, x, w, b, B, C, HW, eps, s);
}

extern "C" int ds_inorm_silu_images_supported(int H, int W) {
  const long long hw = (long long)H * W;
  return H > 0 && W > 0 && hw % 4 == 0 && hw <= 4096;
}

extern "C" int ds_inorm_silu_images(void* images, const float* x, const float* w, const float* b, int B, int C, int H, int W,
                                    float eps, int kind, void* stream) {
  DS_REQUIRE(images && x, DS_ERR_NULL, "ds_inorm_silu_images: NULL pointer");
  DS_REQUIRE(B >= 0 && C > 0 && H > 0 && W > 0, DS_ERR_SHAPE, "ds_inorm_silu_images: bad shape");
  DS_REQUIRE(kind == 0 || kind == 1, DS_ERR_UNSUPPORTED, "ds_inorm_silu_images: kind must be 0 (GroupLN) or 1 (GroupRMS)");
  DS_REQUIRE(ds_inorm_silu_images_supported(H, W), DS_ERR_UNSUPPORTED,
             "ds_inorm_silu_images: planes of H*W <= 4096 floats, H*W a multiple of 4 (got %d x %d)", H, W);
  DS_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(images)) & 15u) == 0, DS_ERR_SHAPE,
             "ds_inorm_silu_images: x and images must be 16-byte aligned");
  if (B == 0) return DS_OK;
  const int nchunk = (C + 15) / 16, hw4 = H * W / 4;
  DS_REQUIRE((long long)B * nchunk * 2 < (1ll << 31), DS_ERR_SHAPE, "ds_inorm_silu_images: too many channel groups");
  const float inv = 1.0f / (float)(H * W);
  hipStream_t s = ds::as_stream(stream);
  const dim3 g((unsigned)(B * nchunk * 2)), t(512);
  u32x4_t* img = reinterpret_cast<u32x4_t*>(images);
#define DS_LI(K) \
  do { \
    if (hw4 <= 64) hipLaunchKernelGGL((k_inorm_images<K, 1>), g, t, 0, s, img, x, w, b, C, nchunk, H, W, hw4, inv, eps); \
    else if (hw4 <= 128) hipLaunchKernelGGL((k_inorm_images<K, 2>), g, t, 0, s, img, x, w, b, C, nchunk, H, W, hw4, inv, eps); \
    else if (hw4 <= 256) hipLaunchKernelGGL((k_inorm_images<K, 4>), g, t, 0, s, img, x, w, b, C, nchunk, H, W, hw4, inv, eps); \
    else if (hw4 <= 512) hipLaunchKernelGGL((k_inorm_images<K, 8>), g, t, 0, s, img, x, w, b, C, nchunk, H, W, hw4, inv, eps); \
    else hipLaunchKernelGGL((k_inorm_images<K, 16>), g, t, 0, s, img, x, w, b, C, nchunk, H, W, hw4, inv, eps); \
  } while (0)
  if (kind == 0) DS_LI(0); else DS_LI(1);
#undef DS_LI
  DS_CHECK_LAUNCH("ds_inorm_silu_images");
  return DS_OK;
}

