// ADM's normalisations: GroupNorm(num_groups = 1) and GroupRMSNorm(1, C) reduce over the whole
// (C, H, W) volume of a sample (adm.py:385-406) -- 32 MiB per sample at BASELINE config 3 -- so
// the statistics are a cross-workgroup reduction:
//   ds_gnorm1_stats   phase 1: grid (chunks, B), fp64 partial sum / sum of squares per chunk;
//                     phase 2: one small workgroup per sample -> (mean, rstd) or (0, sqrt(ms+eps)).
//   ds_gnorm1_apply   one HBM pass (8 B/elt, 5 B/elt when pooling) fusing the normalisation with
//                     everything elementwise that follows it in ADMBaseBlock (adm.py:306-343):
//       kind 0:  SiLU((x - mean)*rstd*w[c] + b[c])                        norm1 -> act
//       kind 1:  SiLU((x/denom*w[c] + b[c])*te1[b,c] + te2[b,c])           norm2 -> FiLM -> act
//       kind 2:  x                                                         (residual branch input)
//     followed, when pool = 1, by the block's 2x2 average pooling (adm.py:316-319, 345-347).
#include "ds_common.h"

namespace {

constexpr int NT = 256;
constexpr int MAXCH = 256;   // phase-1 chunks per sample

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__global__ __launch_bounds__(NT) void k_g1_partial(double* part, const float* __restrict__ x, size_t n_per_sample,
                                                   int chunks) {
  __shared__ double red[2][NT / 64];
  const int b = blockIdx.y, ch = blockIdx.x;
  const size_t per = ((n_per_sample + chunks - 1) / chunks + 3) & ~(size_t)3;
  const size_t lo = (size_t)ch * per;
  size_t hi = lo + per;
  if (hi > n_per_sample) hi = n_per_sample;
  const float* src = x + (size_t)b * n_per_sample;
  double s = 0.0, q = 0.0;
  if (lo < hi) {
    const bool vec = ((reinterpret_cast<uintptr_t>(src + lo) & 15u) == 0);
    size_t i = lo;
    if (vec) {
      const size_t n4 = (hi - lo) / 4;
      const float4* p = reinterpret_cast<const float4*>(src + lo);
      for (size_t k = threadIdx.x; k < n4; k += NT) {
        const float4 v = p[k];
        const float fs = (v.x + v.y) + (v.z + v.w);
        const float fq = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        s += (double)fs;
        q += (double)fq;
      }
      i = lo + n4 * 4;
    }
    for (size_t k = i + threadIdx.x; k < hi; k += NT) {
      const float v = src[k];
      s += (double)v;
      q += (double)v * (double)v;
    }
  }
  s = wave_sum_d(s);
  q = wave_sum_d(q);
  const int wid = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[0][wid] = s; red[1][wid] = q; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double ts = 0.0, tq = 0.0;
    for (int i = 0; i < NT / 64; ++i) { ts += red[0][i]; tq += red[1][i]; }
    part[((size_t)b * chunks + ch) * 2 + 0] = ts;
    part[((size_t)b * chunks + ch) * 2 + 1] = tq;
  }
}

__global__ void k_g1_final(float* stats, const double* __restrict__ part, int chunks, double inv_n, float eps, int kind) {
  const int b = blockIdx.x;
  double s = 0.0, q = 0.0;
  for (int i = threadIdx.x; i < chunks; i += 64) {
    s += part[((size_t)b * chunks + i) * 2 + 0];
    q += part[((size_t)b * chunks + i) * 2 + 1];
  }
  s = wave_sum_d(s);
  q = wave_sum_d(q);
  if (threadIdx.x == 0) {
    if (kind == 0) {
      const double mean = s * inv_n;
      double var = q * inv_n - mean * mean;
      if (var < 0.0) var = 0.0;
      stats[2 * b + 0] = (float)mean;
      stats[2 * b + 1] = 1.0f / sqrtf((float)var + eps);
    } else {
      stats[2 * b + 0] = 0.f;
      stats[2 * b + 1] = sqrtf((float)(q * inv_n) + eps);      // denominator of the RMS norm
    }
  }
}

__device__ __forceinline__ float silu(float v) { return v / (1.0f + expf(-v)); }

// KIND 0: GroupNorm(1, C), stats = (mean, rstd); KIND 1: GroupRMSNorm(1, C), stats = (0, rms denominator);
// either may be followed by FiLM (the second norm of an ADM block, adm.py:306-307,331-333); KIND 2: copy
template <int KIND>
__device__ __forceinline__ float apply1(float x, float mean, float sd, float w, float b, bool film, float f1, float f2) {
  if (KIND == 2) return x;
  float v = KIND == 0 ? (x - mean) * sd * w + b : x / sd * w + b;
  if (film) v = v * f1 + f2;
  return silu(v);
}

// one thread = VEC (4 or 1) consecutive output pixels of one (b, c) row
template <int KIND, int POOL, int VEC>
__global__ __launch_bounds__(NT) void k_g1_apply(float* out, const float* __restrict__ x, const float* __restrict__ stats,
                                                 const float* __restrict__ w, const float* __restrict__ bias,
                                                 const float* __restrict__ film1, const float* __restrict__ film2,
                                                 int film_stride, int C, int Ho, int Wo, size_t total) {
  const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
  if (i >= total) return;
  const int wv = Wo / VEC;
  const int xq = (int)(i % wv);
  size_t t = i / wv;
  const int y = (int)(t % Ho); t /= Ho;
  const int c = (int)(t % C);
  const int b = (int)(t / C);
  const float mean = KIND == 2 ? 0.f : stats[2 * b], sd = KIND == 2 ? 1.f : stats[2 * b + 1];
  const float wc = (KIND == 2 || !w) ? 1.f : w[c], bc = (KIND == 2 || !bias) ? 0.f : bias[c];
  float f1 = 1.f, f2 = 0.f;
  const bool film = KIND != 2 && film1 != nullptr;
  if (film) {
    f1 = film1[(size_t)b * film_stride + c];
    f2 = film2[(size_t)b * film_stride + c];
  }
#define DS_A(v) apply1<KIND>(v, mean, sd, wc, bc, film, f1, f2)
  // torch avg_pool2d: running sum over (kh, kw) in row-major order, then one division
#define DS_P(p, q, r, s) ((((DS_A(p) + DS_A(q)) + DS_A(r)) + DS_A(s)) / 4.0f)
  float* dst = out + (((size_t)b * C + c) * Ho + y) * Wo + VEC * xq;
  if (VEC == 4) {
    float4 o;
    if (POOL == 0) {
      const float4 v = *reinterpret_cast<const float4*>(x + (((size_t)b * C + c) * Ho + y) * Wo + 4 * xq);
      o.x = DS_A(v.x); o.y = DS_A(v.y); o.z = DS_A(v.z); o.w = DS_A(v.w);
    } else {
      const int Wi = 2 * Wo;
      const float* r0 = x + (((size_t)b * C + c) * (2 * Ho) + 2 * y) * Wi + 8 * xq;
      const float4 a0 = *reinterpret_cast<const float4*>(r0), a1 = *reinterpret_cast<const float4*>(r0 + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(r0 + Wi), b1 = *reinterpret_cast<const float4*>(r0 + Wi + 4);
      o.x = DS_P(a0.x, a0.y, b0.x, b0.y);
      o.y = DS_P(a0.z, a0.w, b0.z, b0.w);
      o.z = DS_P(a1.x, a1.y, b1.x, b1.y);
      o.w = DS_P(a1.z, a1.w, b1.z, b1.w);
    }
    *reinterpret_cast<float4*>(dst) = o;
  } else {
    if (POOL == 0) {
      dst[0] = DS_A(x[(((size_t)b * C + c) * Ho + y) * Wo + xq]);
    } else {
      const int Wi = 2 * Wo;
      const float* r0 = x + (((size_t)b * C + c) * (2 * Ho) + 2 * y) * Wi + 2 * xq;
      dst[0] = DS_P(r0[0], r0[1], r0[Wi], r0[Wi + 1]);
    }
  }
#undef DS_P
#undef DS_A
}

// k_g1_apply with the result written as the consuming convolution's pre-split fp16 hi / lo images (ds_conv2d_h3_img; layout and
// rationale: ds_norm.hip, k_inorm_images).  One thread = one position of the padded image of one 8-channel group: border
// positions store zeros, the others apply the same arithmetic as k_g1_apply (bit-identical values) to their 8 channels and split.
typedef _Float16 g1_f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned g1_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void g1_split2(float a, float b, unsigned& hi, unsigned& lo) {        // ds_h3_common.h's split2
  g1_f16x2 h = {(_Float16)a, (_Float16)b};
  unsigned hp = __builtin_bit_cast(unsigned, h);
  asm volatile("" : "+v"(hp));
  const g1_f16x2 hq = __builtin_bit_cast(g1_f16x2, hp);
  g1_f16x2 l = {(_Float16)(a - (float)hq[0]), (_Float16)(b - (float)hq[1])};
  hi = hp;
  lo = __builtin_bit_cast(unsigned, l);
}

template <int KIND, int POOL>
__global__ __launch_bounds__(NT) void k_g1_apply_images(g1_u32x4* __restrict__ img, const float* __restrict__ x,
                                                        const float* __restrict__ stats, const float* __restrict__ w,
                                                        const float* __restrict__ bias, const float* __restrict__ film1,
                                                        const float* __restrict__ film2, int film_stride, int C, int nchunk,
                                                        int Ho, int Wo, size_t total) {
  const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
  if (i >= total) return;
  const int Hp = Ho + 2, Wp = Wo + 2;
  const int px = (int)(i % Wp);
  size_t t = i / Wp;
  const int py = (int)(t % Hp); t /= Hp;
  const int h = (int)(t & 1); t >>= 1;
  const int chunk = (int)(t % nchunk);
  const int b = (int)(t / nchunk);
  g1_u32x4* hi_img = img + ((((size_t)b * nchunk + chunk) * 2 + 0) * 2 + h) * Hp * Wp;
  g1_u32x4* lo_img = img + ((((size_t)b * nchunk + chunk) * 2 + 1) * 2 + h) * Hp * Wp;
  const size_t o = (size_t)py * Wp + px;
  g1_u32x4 qh = {0u, 0u, 0u, 0u}, ql = {0u, 0u, 0u, 0u};
  if (py >= 1 && py <= Ho && px >= 1 && px <= Wo) {
    const int y = py - 1, xq = px - 1;
    const float mean = stats[2 * b], sd = stats[2 * b + 1];
    const bool film = film1 != nullptr;
    float a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int c = 16 * chunk + 8 * h + k;
      if (c < C) {
        const float wc = w ? w[c] : 1.f, bc = bias ? bias[c] : 0.f;
        float f1 = 1.f, f2 = 0.f;
        if (film) { f1 = film1[(size_t)b * film_stride + c]; f2 = film2[(size_t)b * film_stride + c]; }
#define DS_A(v) apply1<KIND>(v, mean, sd, wc, bc, film, f1, f2)
        if (POOL == 0) {
          a[k] = DS_A(x[(((size_t)b * C + c) * Ho + y) * Wo + xq]);
        } else {
          const int Wi = 2 * Wo;
          const float* r0 = x + (((size_t)b * C + c) * (2 * Ho) + 2 * y) * Wi + 2 * xq;
          a[k] = (((DS_A(r0[0]) + DS_A(r0[1])) + DS_A(r0[Wi])) + DS_A(r0[Wi + 1])) / 4.0f;      // torch avg_pool2d's order
        }
#undef DS_A
      } else {
        a[k] = 0.f;
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      unsigned ph, pl;
      g1_split2(a[2 * j], a[2 * j + 1], ph, pl);
      qh[j] = ph; ql[j] = pl;
    }
  }
  hi_img[o] = qh;
  lo_img[o] = ql;
}

// The fused loader's activation SiLU((x - M) * A + C) from a norm table [B, ceil16(C), 4] (ds_inorm_table / ds_gnorm1_table),
// written as pre-split images: any plane size, any norm the tables describe.  Same arithmetic as the loader (hardware exp2 / rcp).
__global__ __launch_bounds__(NT) void k_table_apply_images(g1_u32x4* __restrict__ img, const float* __restrict__ x,
                                                           const float4* __restrict__ table, int C, int nchunk, int H, int W,
                                                           size_t total) {
  const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
  if (i >= total) return;
  const int Hp = H + 2, Wp = W + 2;
  const int px = (int)(i % Wp);
  size_t t = i / Wp;
  const int py = (int)(t % Hp); t /= Hp;
  const int h = (int)(t & 1); t >>= 1;
  const int chunk = (int)(t % nchunk);
  const int b = (int)(t / nchunk);
  g1_u32x4* hi_img = img + ((((size_t)b * nchunk + chunk) * 2 + 0) * 2 + h) * Hp * Wp;
  g1_u32x4* lo_img = img + ((((size_t)b * nchunk + chunk) * 2 + 1) * 2 + h) * Hp * Wp;
  const size_t o = (size_t)py * Wp + px;
  g1_u32x4 qh = {0u, 0u, 0u, 0u}, ql = {0u, 0u, 0u, 0u};
  if (py >= 1 && py <= H && px >= 1 && px <= W) {
    const float4* trow = table + ((size_t)b * nchunk + chunk) * 16 + 8 * h;
    float a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int c = 16 * chunk + 8 * h + k;
      if (c < C) {
        const float4 p = trow[k];
        const float v = (x[(((size_t)b * C + c) * H + (py - 1)) * W + (px - 1)] - p.x) * p.y + p.z;
        a[k] = v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.44269504088896341f));     // ds_h3::fast_silu
      } else {
        a[k] = 0.f;
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      unsigned ph, pl;
      g1_split2(a[2 * j], a[2 * j + 1], ph, pl);
      qh[j] = ph; ql[j] = pl;
    }
  }
  hi_img[o] = qh;
  lo_img[o] = ql;
}

__global__ __launch_bounds__(NT) void k_concat2(float* out, const float* __restrict__ a, const float* __restrict__ b,
                                                size_t na4, size_t nb4, size_t total4) {
  const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
  if (i >= total4) return;
  const size_t per = na4 + nb4;
  const size_t s = i / per, r = i - s * per;
  const float4 v = r < na4 ? reinterpret_cast<const float4*>(a)[s * na4 + r]
                           : reinterpret_cast<const float4*>(b)[s * nb4 + (r - na4)];
  reinterpret_cast<float4*>(out)[i] = v;
}

__global__ void k_add_act(float* out, const float* __restrict__ a, const float* __restrict__ add, int add_rows, int M,
                          int N, int act) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * N) return;
  const int m = i / N, n = i - m * N;
  float v = a[i];
  if (add) v = v + add[(size_t)(add_rows == 1 ? 0 : m) * N + n];
  if (act == 1) v = silu(v);
  else if (act == 2) v = fmaxf(v, 0.f);
  out[i] = v;
}

}  // namespace

extern "C" {

size_t ds_gnorm1_workspace_bytes(int B) { return (size_t)(B > 0 ? B : 0) * MAXCH * 2 * sizeof(double); }

int ds_gnorm1_stats(float* stats, void* workspace, const float* x, int B, int C, int HW, float eps, int kind,
                    void* stream) {
  DS_REQUIRE(stats && workspace && x, DS_ERR_NULL, "ds_gnorm1_stats: NULL pointer");
  DS_REQUIRE(B >= 0 && C > 0 && HW > 0, DS_ERR_SHAPE, "ds_gnorm1_stats: bad shape B=%d C=%d HW=%d", B, C, HW);
  DS_REQUIRE(kind == 0 || kind == 1, DS_ERR_UNSUPPORTED, "ds_gnorm1_stats: kind %d", kind);
  DS_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 7u) == 0, DS_ERR_SHAPE, "ds_gnorm1_stats: workspace must be 8-byte aligned");
  if (B == 0) return DS_OK;
  const size_t n = (size_t)C * HW;
  int chunks = (int)((n + (size_t)NT * 64 - 1) / ((size_t)NT * 64));
  if (chunks > MAXCH) chunks = MAXCH;
  if (chunks < 1) chunks = 1;
  hipStream_t s = ds::as_stream(stream);
  hipLaunchKernelGGL(k_g1_partial, dim3(chunks, B), dim3(NT), 0, s, reinterpret_cast<double*>(workspace), x, n, chunks);
  DS_CHECK_LAUNCH("ds_gnorm1_stats (partial)");
  hipLaunchKernelGGL(k_g1_final, dim3(B), dim3(64), 0, s, stats, reinterpret_cast<const double*>(workspace), chunks,
                     1.0 / (double)n, eps, kind);
  DS_CHECK_LAUNCH("ds_gnorm1_stats (final)");
  return DS_OK;
}

int ds_gnorm1_apply(float* out, const float* x, const float* stats, const float* w, const float* b,
                    const float* film_scale, const float* film_shift, int film_stride, int B, int C, int H, int W,
                    int kind, int pool, void* stream) {
  DS_REQUIRE(out && x, DS_ERR_NULL, "ds_gnorm1_apply: NULL pointer");
  DS_REQUIRE(kind >= 0 && kind <= 2 && (pool == 0 || pool == 1), DS_ERR_UNSUPPORTED, "ds_gnorm1_apply: kind %d pool %d", kind, pool);
  DS_REQUIRE(kind == 2 || stats, DS_ERR_NULL, "ds_gnorm1_apply: stats is NULL");
  DS_REQUIRE((film_scale == nullptr) == (film_shift == nullptr), DS_ERR_NULL, "ds_gnorm1_apply: FiLM scale and shift go together");
  DS_REQUIRE(B >= 0 && C > 0 && H > 0 && W > 0, DS_ERR_SHAPE, "ds_gnorm1_apply: bad shape");
  const int Ho = pool ? H / 2 : H, Wo = pool ? W / 2 : W;
  DS_REQUIRE(!pool || (H % 2 == 0 && W % 2 == 0), DS_ERR_SHAPE, "ds_gnorm1_apply: pooling needs even H, W");
  if (B == 0) return DS_OK;
  // 16-byte path when every row start is 16-byte aligned; scalar path otherwise (tiny / odd fields)
  const bool vec = Wo % 4 == 0 && ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(x)) & 15u) == 0;
  const size_t total = (size_t)B * C * Ho * (vec ? Wo / 4 : Wo);
  dim3 g((unsigned)((total + NT - 1) / NT)), t(NT);
  hipStream_t s = ds::as_stream(stream);
#define L(K, P, V) hipLaunchKernelGGL((k_g1_apply<K, P, V>), g, t, 0, s, out, x, stats, w, b, film_scale, film_shift, film_stride, C, Ho, Wo, total)
#define LV(K, P) do { if (vec) L(K, P, 4); else L(K, P, 1); } while (0)
  if (kind == 0) { if (pool) LV(0, 1); else LV(0, 0); }
  else if (kind == 1) { if (pool) LV(1, 1); else LV(1, 0); }
  else { if (pool) LV(2, 1); else LV(2, 0); }
#undef LV
#undef L
  DS_CHECK_LAUNCH("ds_gnorm1_apply");
  return DS_OK;
}

int ds_gnorm1_apply_images(void* images, const float* x, const float* stats, const float* w, const float* b,
                           const float* film_scale, const float* film_shift, int film_stride, int B, int C, int Ho, int Wo,
                           int kind, int pool, void* stream) {
  DS_REQUIRE(images && x && stats, DS_ERR_NULL, "ds_gnorm1_apply_images: NULL pointer");
  DS_REQUIRE(B >= 0 && C > 0 && Ho > 0 && Wo > 0, DS_ERR_SHAPE, "ds_gnorm1_apply_images: bad shape");
  DS_REQUIRE(kind == 0 || kind == 1, DS_ERR_UNSUPPORTED, "ds_gnorm1_apply_images: kind must be 0 (GroupNorm(1,C)) or 1 (GroupRMSNorm(1,C))");
  DS_REQUIRE((film_scale == nullptr) == (film_shift == nullptr), DS_ERR_NULL, "ds_gnorm1_apply_images: FiLM needs scale and shift");
  DS_REQUIRE((reinterpret_cast<uintptr_t>(images) & 15u) == 0, DS_ERR_SHAPE, "ds_gnorm1_apply_images: images must be 16-byte aligned");
  if (B == 0) return DS_OK;
  const int nchunk = (C + 15) / 16;
  const size_t total = (size_t)B * nchunk * 2 * (Ho + 2) * (Wo + 2);
  DS_REQUIRE((total + NT - 1) / NT < (1ull << 31), DS_ERR_SHAPE, "ds_gnorm1_apply_images: too many positions");
  g1_u32x4* img = reinterpret_cast<g1_u32x4*>(images);
  dim3 g((unsigned)((total + NT - 1) / NT)), t(NT);
  hipStream_t s = ds::as_stream(stream);
#define L(K, P) hipLaunchKernelGGL((k_g1_apply_images<K, P>), g, t, 0, s, img, x, stats, w, b, film_scale, film_shift, film_stride, C, nchunk, Ho, Wo, total)
  if (kind == 0) { if (pool) L(0, 1); else L(0, 0); }
  else { if (pool) L(1, 1); else L(1, 0); }
#undef L
  DS_CHECK_LAUNCH("ds_gnorm1_apply_images");
  return DS_OK;
}

int ds_table_apply_images(void* images, const float* x, const float* table, int B, int C, int H, int W, void* stream) {
  DS_REQUIRE(images && x && table, DS_ERR_NULL, "ds_table_apply_images: NULL pointer");
  DS_REQUIRE(B >= 0 && C > 0 && H > 0 && W > 0, DS_ERR_SHAPE, "ds_table_apply_images: bad shape");
  DS_REQUIRE(((reinterpret_cast<uintptr_t>(images) | reinterpret_cast<uintptr_t>(table)) & 15u) == 0, DS_ERR_SHAPE,
             "ds_table_apply_images: images and table must be 16-byte aligned");
  if (B == 0) return DS_OK;
  const int nchunk = (C + 15) / 16;
  const size_t total = (size_t)B * nchunk * 2 * (H + 2) * (W + 2);
  DS_REQUIRE((total + NT - 1) / NT < (1ull << 31), DS_ERR_SHAPE, "ds_table_apply_images: too many positions");
  hipLaunchKernelGGL(k_table_apply_images, dim3((unsigned)((total + NT - 1) / NT)), dim3(NT), 0, ds::as_stream(stream),
                     reinterpret_cast<g1_u32x4*>(images), x, reinterpret_cast<const float4*>(table), C, nchunk, H, W, total);
  DS_CHECK_LAUNCH("ds_table_apply_images");
  return DS_OK;
}

int ds_concat2(float* out, const float* a, const float* b, int B, size_t na, size_t nb, void* stream) {
  DS_REQUIRE(out && a && b, DS_ERR_NULL, "ds_concat2: NULL pointer");
  DS_REQUIRE(na % 4 == 0 && nb % 4 == 0, DS_ERR_UNSUPPORTED, "ds_concat2: per-sample sizes must be multiples of 4");
  DS_REQUIRE(((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15u) == 0,
             DS_ERR_SHAPE, "ds_concat2: pointers must be 16-byte aligned");
  if (B <= 0) return DS_OK;
  const size_t total4 = (size_t)B * (na + nb) / 4;
  hipLaunchKernelGGL(k_concat2, dim3((unsigned)((total4 + NT - 1) / NT)), dim3(NT), 0, ds::as_stream(stream), out, a, b,
                     na / 4, nb / 4, total4);
  DS_CHECK_LAUNCH("ds_concat2");
  return DS_OK;
}

int ds_add_act(float* out, const float* a, const float* add, int add_rows, int M, int N, int act, void* stream) {
  DS_REQUIRE(out && a, DS_ERR_NULL, "ds_add_act: NULL pointer");
  DS_REQUIRE(M >= 0 && N > 0 && act >= 0 && act <= 2, DS_ERR_SHAPE, "ds_add_act: bad arguments");
  DS_REQUIRE(add == nullptr || add_rows == 1 || add_rows == M, DS_ERR_SHAPE, "ds_add_act: add_rows=%d must be 1 or M=%d", add_rows, M);
  if (M == 0) return DS_OK;
  hipLaunchKernelGGL(k_add_act, dim3((M * N + 255) / 256), dim3(256), 0, ds::as_stream(stream), out, a, add, add_rows,
                     M, N, act);
  DS_CHECK_LAUNCH("ds_add_act");
  return DS_OK;
}

}  // extern "C"
