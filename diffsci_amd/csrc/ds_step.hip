// Karras stepper kernels: HBM-bound elementwise passes over the state tensor.
//
// Arithmetic is written in the reference's own operation order (one rounding per op, no FMA
// contraction: the file is compiled with -ffp-contract=off) so that, given the same network
// outputs, every result is bit-identical to the reference's torch fp32 path:
//   D     = c_out*F + c_skip*x                      karrasmodule.py:717-718
//   score = (D - x)/sigma^2                         karrasmodule.py:733
//   d     = (-(sigma*sigma'))*score [+ -(lambda*score)]   schedulers.py:267-274
//   Euler: x + dt*d ; Heun: x + (0.5*(d1+d2))*dt    integrators.py:35,46,53
// Layout: flat fp32; 16 B per lane per access (float4), grid-stride, <= 2048 workgroups.
#include "ds_common.h"

namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ float score_of(float x, float f, float fu, bool has_u, const ds_eval_coef& k) {
  float F = f;
  if (has_u) F = k.one_minus_guidance * fu + k.guidance * f;
  float so = k.c_out * F;
  float D = so + k.c_skip * x;
  return (D - x) / k.sigma_sq;
}

__device__ __forceinline__ float drift(float x, float f, float fu, bool has_u, const ds_eval_coef& k) {
  if (k.input_kind == DS_IN_DRIFT) return f;
  if (k.input_kind == DS_IN_FLOW) {              // f is a flow field v(x, t) (flowfield.py:441-458)
    float F = f;
    if (has_u) F = k.one_minus_guidance * fu + k.guidance * f;
    return k.neg_mult * (F / k.sigma_sq);
  }
  if (k.scaled) {                                // non-constant scaling (VP), schedulers.py:275-293
    const float xs = x / k.scale;                // score_fn(x / s, sigma)
    const float score = (k.input_kind == DS_IN_SCORE) ? f : score_of(xs, f, fu, has_u, k);
    float d = k.scale_mult * x + k.neg_mult * score;          // scale_multiplier*x - multiplier*score
    if (k.stochastic) d = d + k.neg_lang * score;             // -(langevin * 1/s * score)
    return d;
  }
  float score = (k.input_kind == DS_IN_SCORE) ? f : score_of(x, f, fu, has_u, k);
  float d = k.neg_mult * score;
  if (k.stochastic) d = d + k.neg_lang * score;
  return d;
}

// the next evaluation's network input: c_in * x, or c_in * (x / s) under a non-constant scaling
__device__ __forceinline__ float next_input(float r, float c_in_next, float next_scale) {
  return (next_scale == 1.0f || next_scale == 0.0f) ? c_in_next * r : c_in_next * (r / next_scale);   // 0: a zero-initialised struct
}

// the range guard's result check folded into the run's last step (nets/precision.py): inf / NaN in what this lane wrote raises
// the word; one atomic per wave that saw any, none on a finite run
__device__ __forceinline__ bool not_finite(float v) { return !(__builtin_fabsf(v) <= 3.402823466e+38f); }
__device__ __forceinline__ void raise_nonfinite(unsigned* word, bool bad) {
  if (word == nullptr) return;
  const unsigned long long m = __builtin_amdgcn_ballot_w64(bad);                 // called outside the loops: all lanes arrive
  if (m != 0ull && (int)(threadIdx.x & 63) == __builtin_ctzll(m)) atomicOr(word, 1u);
}

inline int grid_for(size_t n4) {
  size_t g = (n4 + kThreads - 1) / kThreads;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (int)g;
}

// ---- counter-based noise: Philox4x32-10 + Box-Muller (the in-kernel eps of SURVEY 8a6 / 8b) ----------------
// eps for element e of step j comes from counter  state[1] + offset_j + e/4  under key state[0]; its four 32-bit
// outputs give the four normals of elements 4*(e/4) .. 4*(e/4)+3.  Nothing depends on the launch geometry, so
// a replay with the same (seed, offset) reproduces the draw bit for bit and the launch is graph-capturable:
// (seed, base offset) live in device memory, the per-step offset is a by-value argument.
// oracle/philox_ref.py restates this stream in numpy.
__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
    c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
    k.x += 0x9E3779B9u;
    k.y += 0xBB67AE85u;
  }
  return c;
}

__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float& z0, float& z1) {
  const float u1 = (float)a * 2.3283064365386963e-10f + 1.1641532182693481e-10f;   // (0, 1]
  const float u2 = (float)b * 2.3283064365386963e-10f;                               // [0, 1]
  const float rad = sqrtf(-2.0f * logf(u1));
  float sn, cs;
  sincospif(2.0f * u2, &sn, &cs);
  z0 = rad * cs;
  z1 = rad * sn;
}

__device__ __forceinline__ float4 philox_normal4(const unsigned long long* __restrict__ state, unsigned long long offset,
                                                 size_t i4) {
  const unsigned long long seed = state[0], ctr = state[1] + offset + (unsigned long long)i4;
  const uint4 r = philox4x32_10(make_uint4((uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u),
                                make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
  float4 z;
  box_muller(r.x, r.y, z.x, z.y);
  box_muller(r.z, r.w, z.z, z.w);
  return z;
}

__device__ __forceinline__ float philox_normal1(const unsigned long long* __restrict__ state, unsigned long long offset,
                                                size_t e) {
  const float4 z = philox_normal4(state, offset, e >> 2);
  const int j = (int)(e & 3);
  return j == 0 ? z.x : j == 1 ? z.y : j == 2 ? z.z : z.w;
}

__global__ __launch_bounds__(kThreads) void k_philox_normal(float* __restrict__ out, const unsigned long long* state,
                                                            unsigned long long offset, size_t n4, size_t n) {
  size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
  size_t stride = (size_t)gridDim.x * kThreads;
  for (; i < n4; i += stride) reinterpret_cast<float4*>(out)[i] = philox_normal4(state, offset, i);
  for (size_t t = n4 * 4 + (size_t)blockIdx.x * kThreads + threadIdx.x; t < n; t += stride)
    out[t] = philox_normal1(state, offset, t);
}

__global__ __launch_bounds__(kThreads) void k_scale(float* __restrict__ out, const float* __restrict__ x, float s,
                                                    size_t n4, size_t n) {
  size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
  size_t stride = (size_t)gridDim.x * kThreads;
  for (; i < n4; i += stride) {
    float4 v = reinterpret_cast<const float4*>(x)[i];
    v.x = s * v.x; v.y = s * v.y; v.z = s * v.z; v.w = s * v.w;
    reinterpret_cast<float4*>(out)[i] = v;
  }
  // tail (n not a multiple of 4), or everything when a pointer is not 16-byte aligned (n4 = 0)
  for (size_t t = n4 * 4 + (size_t)blockIdx.x * kThreads + threadIdx.x; t < n; t += stride) out[t] = s * x[t];
}

template <bool HAS_U, bool HAS_EPS>
__device__ __forceinline__ void euler_one(float x, float f, float fu, float e, const ds_eval_coef& k, float dt,
                                          float noise_coef, float sq, float c_in_next, float& xo, float& xi) {
  float d = drift(x, f, fu, HAS_U, k);
  float r = x + dt * d;
  if (HAS_EPS) r = r + (noise_coef * e) * sq;
  xo = r;
  xi = next_input(r, c_in_next, k.next_scale);
}

template <bool HAS_U, int NOISE>        // NOISE: 0 none, 1 injected eps, 2 in-kernel Philox
__global__ __launch_bounds__(kThreads) void k_euler(float* x_out, float* xin_out, const float* x,
                                                    const float* __restrict__ f, const float* __restrict__ fu,
                                                    const float* __restrict__ eps, const unsigned long long* rng,
                                                    unsigned long long rng_offset, ds_eval_coef k, float dt,
                                                    float c_in_next, float noise_coef, float sq, size_t n4, size_t n) {
  constexpr bool HAS_EPS = NOISE != 0;
  size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
  size_t stride = (size_t)gridDim.x * kThreads;
  bool bad = false;
  for (; i < n4; i += stride) {
    float4 vx = reinterpret_cast<const float4*>(x)[i];
    float4 vf = reinterpret_cast<const float4*>(f)[i];
    float4 vu = make_float4(0, 0, 0, 0), ve = make_float4(0, 0, 0, 0);
    if (HAS_U) vu = reinterpret_cast<const float4*>(fu)[i];
    if (NOISE == 1) ve = reinterpret_cast<const float4*>(eps)[i];
    if (NOISE == 2) ve = philox_normal4(rng, rng_offset, i);
    float4 o, q;
    euler_one<HAS_U, HAS_EPS>(vx.x, vf.x, vu.x, ve.x, k, dt, noise_coef, sq, c_in_next, o.x, q.x);
    euler_one<HAS_U, HAS_EPS>(vx.y, vf.y, vu.y, ve.y, k, dt, noise_coef, sq, c_in_next, o.y, q.y);
    euler_one<HAS_U, HAS_EPS>(vx.z, vf.z, vu.z, ve.z, k, dt, noise_coef, sq, c_in_next, o.z, q.z);
    euler_one<HAS_U, HAS_EPS>(vx.w, vf.w, vu.w, ve.w, k, dt, noise_coef, sq, c_in_next, o.w, q.w);
    bad = bad || not_finite(o.x) || not_finite(o.y) || not_finite(o.z) || not_finite(o.w);
    if (x_out) reinterpret_cast<float4*>(x_out)[i] = o;
    if (xin_out) {
      reinterpret_cast<float4*>(xin_out)[i] = q;
      if (k.xin_copies == 2) reinterpret_cast<float4*>(xin_out + n)[i] = q;       // the batched-guidance evaluation reads [2B, ...]
    }
  }
  for (size_t t = n4 * 4 + (size_t)blockIdx.x * kThreads + threadIdx.x; t < n; t += stride) {
    float o, q;
    const float e = NOISE == 1 ? eps[t] : NOISE == 2 ? philox_normal1(rng, rng_offset, t) : 0.f;
    euler_one<HAS_U, HAS_EPS>(x[t], f[t], HAS_U ? fu[t] : 0.f, e, k, dt, noise_coef, sq, c_in_next, o, q);
    bad = bad || not_finite(o);
    if (x_out) x_out[t] = o;
    if (xin_out) {
      xin_out[t] = q;
      if (k.xin_copies == 2) xin_out[n + t] = q;
    }
  }
  raise_nonfinite(k.nonfinite, bad);
}

template <bool HAS_U>
__device__ __forceinline__ void heun_one(float x, float f1, float f1u, float f2, float f2u, const ds_eval_coef& k1,
                                         const ds_eval_coef& k2, float dt, float c_in_next, float& xo, float& xi) {
  float d1 = drift(x, f1, f1u, HAS_U, k1);
  float xe = x + dt * d1;
  float d2 = drift(xe, f2, f2u, HAS_U, k2);
  float r = x + (0.5f * (d1 + d2)) * dt;
  xo = r;
  xi = next_input(r, c_in_next, k2.next_scale);
}

template <bool HAS_U>
__global__ __launch_bounds__(kThreads) void k_heun(float* x_out, float* xin_out, const float* x,
                                                   const float* __restrict__ f1, const float* __restrict__ f1u,
                                                   const float* __restrict__ f2, const float* __restrict__ f2u,
                                                   ds_eval_coef k1, ds_eval_coef k2, float dt, float c_in_next,
                                                   size_t n4, size_t n) {
  size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
  size_t stride = (size_t)gridDim.x * kThreads;
  bool bad = false;
  for (; i < n4; i += stride) {
    float4 vx = reinterpret_cast<const float4*>(x)[i];
    float4 a = reinterpret_cast<const float4*>(f1)[i];
    float4 b = reinterpret_cast<const float4*>(f2)[i];
    float4 au = make_float4(0, 0, 0, 0), bu = make_float4(0, 0, 0, 0);
    if (HAS_U) {
      au = reinterpret_cast<const float4*>(f1u)[i];
      bu = reinterpret_cast<const float4*>(f2u)[i];
    }
    float4 o, q;
    heun_one<HAS_U>(vx.x, a.x, au.x, b.x, bu.x, k1, k2, dt, c_in_next, o.x, q.x);
    heun_one<HAS_U>(vx.y, a.y, au.y, b.y, bu.y, k1, k2, dt, c_in_next, o.y, q.y);
    heun_one<HAS_U>(vx.z, a.z, au.z, b.z, bu.z, k1, k2, dt, c_in_next, o.z, q.z);
    heun_one<HAS_U>(vx.w, a.w, au.w, b.w, bu.w, k1, k2, dt, c_in_next, o.w, q.w);
    bad = bad || not_finite(o.x) || not_finite(o.y) || not_finite(o.z) || not_finite(o.w);
    reinterpret_cast<float4*>(x_out)[i] = o;
    if (xin_out) {
      reinterpret_cast<float4*>(xin_out)[i] = q;
      if (k2.xin_copies == 2) reinterpret_cast<float4*>(xin_out + n)[i] = q;
    }
  }
  for (size_t t = n4 * 4 + (size_t)blockIdx.x * kThreads + threadIdx.x; t < n; t += stride) {
    float o, q;
    heun_one<HAS_U>(x[t], f1[t], HAS_U ? f1u[t] : 0.f, f2[t], HAS_U ? f2u[t] : 0.f, k1, k2, dt, c_in_next, o, q);
    bad = bad || not_finite(o);
    x_out[t] = o;
    if (xin_out) {
      xin_out[t] = q;
      if (k2.xin_copies == 2) xin_out[n + t] = q;
    }
  }
  raise_nonfinite(k2.nonfinite, bad);
}

template <bool HAS_U, bool SCORE_ONLY>
__global__ __launch_bounds__(kThreads) void k_drift(float* out, const float* x, const float* __restrict__ f,
                                                    const float* __restrict__ fu, ds_eval_coef k, size_t n) {
  size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
  size_t stride = (size_t)gridDim.x * kThreads;
  for (; i < n; i += stride) {
    float xv = x ? x[i] : 0.f;
    float fuv = HAS_U ? fu[i] : 0.f;
    out[i] = SCORE_ONLY ? score_of(xv, f[i], fuv, HAS_U, k) : drift(xv, f[i], fuv, HAS_U, k);
  }
}

template <bool PHILOX>
__global__ __launch_bounds__(kThreads) void k_churn(float* xhat, float* xin_out, const float* x,
                                                    const float* __restrict__ eps, const unsigned long long* rng,
                                                    unsigned long long rng_offset, float coef, float c_in, float ratio,
                                                    float scale, int xin_copies, size_t n4, size_t n) {
  size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
  size_t stride = (size_t)gridDim.x * kThreads;
  for (; i < n4; i += stride) {
    float4 vx = reinterpret_cast<const float4*>(x)[i];
    float4 ve = PHILOX ? philox_normal4(rng, rng_offset, i) : reinterpret_cast<const float4*>(eps)[i];
    float4 o, q;
    if (ratio != 1.0f) { vx.x = ratio * vx.x; vx.y = ratio * vx.y; vx.z = ratio * vx.z; vx.w = ratio * vx.w; }   // (s_hat/s)*x
    o.x = vx.x + coef * ve.x; o.y = vx.y + coef * ve.y; o.z = vx.z + coef * ve.z; o.w = vx.w + coef * ve.w;
    q.x = next_input(o.x, c_in, scale); q.y = next_input(o.y, c_in, scale); q.z = next_input(o.z, c_in, scale); q.w = next_input(o.w, c_in, scale);
    reinterpret_cast<float4*>(xhat)[i] = o;
    if (xin_out) {
      reinterpret_cast<float4*>(xin_out)[i] = q;
      if (xin_copies == 2) reinterpret_cast<float4*>(xin_out + n)[i] = q;
    }
  }
  for (size_t t = n4 * 4 + (size_t)blockIdx.x * kThreads + threadIdx.x; t < n; t += stride) {
    float o = (ratio != 1.0f ? ratio * x[t] : x[t]) + coef * (PHILOX ? philox_normal1(rng, rng_offset, t) : eps[t]);
    xhat[t] = o;
    if (xin_out) {
      xin_out[t] = next_input(o, c_in, scale);
      if (xin_copies == 2) xin_out[n + t] = xin_out[t];
    }
  }
}

__global__ __launch_bounds__(kThreads) void k_denoiser(float* out, const float* __restrict__ x,
                                                       const float* __restrict__ f, const float* __restrict__ fu,
                                                       float g, float omg, const float* __restrict__ c_out,
                                                       const float* __restrict__ c_skip, size_t nps, size_t total) {
  size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
  size_t stride = (size_t)gridDim.x * kThreads;
  for (; i < total; i += stride) {
    size_t b = i / nps;
    float F = f[i];
    if (fu) F = omg * fu[i] + g * F;
    float so = c_out[b] * F;
    out[i] = so + c_skip[b] * x[i];
  }
}

// out = x*(1 - mask) + y*mask, mask broadcast over the batch (schedulers.py:112,116,146: inpaint / repaint)
__global__ __launch_bounds__(kThreads) void k_mask_blend(float* out, const float* __restrict__ x,
                                                         const float* __restrict__ y, const float* __restrict__ mask,
                                                         size_t nps, size_t total) {
  size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
  size_t stride = (size_t)gridDim.x * kThreads;
  for (; i < total; i += stride) {
    const float m = mask[i % nps];
    const float a = x[i] * (1.0f - m);
    const float b = y[i] * m;
    out[i] = a + b;
  }
}

// out = a*x + b*y (y optional) and out = x / s: the non-constant-scaling branch of Scheduler.rhs
// (schedulers.py:275-293): score_fn(x/s, sigma);  (s'/s)*x - multiplier*score
__global__ __launch_bounds__(kThreads) void k_axpby(float* out, const float* __restrict__ x, float a,
                                                    const float* __restrict__ y, float b, size_t n) {
  size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
  size_t stride = (size_t)gridDim.x * kThreads;
  for (; i < n; i += stride) {
    float v = a * x[i];
    if (y) { const float w = b * y[i]; v = v + w; }
    out[i] = v;
  }
}
__global__ __launch_bounds__(kThreads) void k_div_scalar(float* out, const float* __restrict__ x, float s, size_t n) {
  size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
  size_t stride = (size_t)gridDim.x * kThreads;
  for (; i < n; i += stride) out[i] = x[i] / s;
}

// DimensionAgnosticBatchNorm in eval mode (aux_scripts/batchnorm.py:111-170), the reference's operation order:
//   FORWARD:  x = (x - mean) / sqrt(var + eps);  [x = x*w + b;]  x = x*sigma
//   inverse:  x = x / sigma;  [x = (x - b) / w;]  x = x*sqrt(var + eps) + mean
// statistics / affine hold nc = 1 (broadcast) or C entries; grid.y = B*C planes
template <bool FORWARD>
__global__ __launch_bounds__(kThreads) void k_batchnorm(float* out, const float* __restrict__ x, const float* __restrict__ mean,
                                                        const float* __restrict__ var, const float* __restrict__ w,
                                                        const float* __restrict__ b, float eps, float sigma, int C, int nc,
                                                        size_t HW) {
  const int plane = blockIdx.y;
  const int c = nc == 1 ? 0 : plane % C;
  const float m = mean[c], sd = sqrtf(var[c] + eps);
  const float wc = w ? w[c] : 1.0f, bc = b ? b[c] : 0.0f;
  const bool affine = w != nullptr;
  const float* xp = x + (size_t)plane * HW;
  float* op = out + (size_t)plane * HW;
  for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < HW; i += (size_t)gridDim.x * kThreads) {
    float v = xp[i];
    if (FORWARD) {
      v = (v - m) / sd;
      if (affine) v = v * wc + bc;
      v = v * sigma;
    } else {
      v = v / sigma;
      if (affine) v = (v - bc) / wc;
      v = v * sd + m;
    }
    op[i] = v;
  }
}

// out[i] = x1 + ((x2 - x1) * i) / (n - 1), i = 0..n-1  (torchutils.py:64-65, same operation order)
__global__ __launch_bounds__(kThreads) void k_lerp_stack(float* out, const float* __restrict__ x1,
                                                         const float* __restrict__ x2, int n, size_t numel) {
  size_t e = (size_t)blockIdx.x * kThreads + threadIdx.x;
  size_t stride = (size_t)gridDim.x * kThreads;
  const float den = (float)(n - 1);
  for (; e < numel; e += stride) {
    const float a = x1[e], d = x2[e] - a;
    for (int i = 0; i < n; ++i) out[(size_t)i * numel + e] = a + (d * (float)i) / den;
  }
}

__global__ __launch_bounds__(kThreads) void k_add(float* out, const float* __restrict__ a,
                                                  const float* __restrict__ b, size_t n4, size_t n) {
  size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
  size_t stride = (size_t)gridDim.x * kThreads;
  for (; i < n4; i += stride) {
    float4 va = reinterpret_cast<const float4*>(a)[i];
    float4 vb = reinterpret_cast<const float4*>(b)[i];
    va.x += vb.x; va.y += vb.y; va.z += vb.z; va.w += vb.w;
    reinterpret_cast<float4*>(out)[i] = va;
  }
  for (size_t t = n4 * 4 + (size_t)blockIdx.x * kThreads + threadIdx.x; t < n; t += stride) out[t] = a[t] + b[t];
}

inline bool blends(int input_kind) { return input_kind == DS_IN_NETWORK || input_kind == DS_IN_FLOW; }

// float4 iterations of a launch: n/4 when every (non-NULL) pointer is 16-byte aligned, else 0 -- the kernels'
// scalar grid-stride tail then covers everything (odd-sized states: history[i] / eps[i] slices of [B,2] toys with
// odd B, per-sample views x[b] of 3x3 fields, ...; the reference accepts any shape).
inline size_t vec4_count(size_t n, std::initializer_list<const void*> ptrs) {
  for (const void* p : ptrs)
    if (p && (reinterpret_cast<uintptr_t>(p) & 15u)) return 0;
  return n / 4;
}
inline int grid_elems(size_t n4, size_t n) { return grid_for(n4 ? n4 : (n + 3) / 4); }

}  // namespace

extern "C" {

int ds_karras_scale(float* out, const float* x, float s, size_t n, void* stream) {
  DS_REQUIRE(out && x, DS_ERR_NULL, "ds_karras_scale: NULL pointer");
  if (n == 0) return DS_OK;
  const size_t n4 = vec4_count(n, {out, x});
  hipLaunchKernelGGL(k_scale, dim3(grid_elems(n4, n)), dim3(kThreads), 0, ds::as_stream(stream), out, x, s, n4, n);
  DS_CHECK_LAUNCH("ds_karras_scale");
  return DS_OK;
}

int ds_philox_normal(float* out, const uint64_t* philox_state, uint64_t philox_offset, size_t n, void* stream) {
  DS_REQUIRE(out && philox_state, DS_ERR_NULL, "ds_philox_normal: NULL pointer");
  DS_REQUIRE((reinterpret_cast<uintptr_t>(philox_state) & 7u) == 0, DS_ERR_SHAPE, "ds_philox_normal: state must be 8-byte aligned");
  if (n == 0) return DS_OK;
  const size_t n4 = vec4_count(n, {out});
  hipLaunchKernelGGL(k_philox_normal, dim3(grid_elems(n4, n)), dim3(kThreads), 0, ds::as_stream(stream), out,
                     reinterpret_cast<const unsigned long long*>(philox_state), (unsigned long long)philox_offset, n4, n);
  DS_CHECK_LAUNCH("ds_philox_normal");
  return DS_OK;
}

int ds_karras_euler(float* x_out, float* xin_out, const float* x, const float* f, const float* fu,
                    const ds_eval_coef* k, float dt, float c_in_next, const float* eps, const uint64_t* philox_state,
                    uint64_t philox_offset, float noise_coef, float sqrt_abs_dt, size_t n, void* stream) {
  DS_REQUIRE(x && f && k, DS_ERR_NULL, "ds_karras_euler: NULL pointer");
  DS_REQUIRE(x_out || xin_out, DS_ERR_NULL, "ds_karras_euler: no output requested");
  DS_REQUIRE(!(eps && philox_state), DS_ERR_SHAPE, "ds_karras_euler: give injected eps or a Philox state, not both");
  DS_REQUIRE((reinterpret_cast<uintptr_t>(philox_state) & 7u) == 0, DS_ERR_SHAPE, "ds_karras_euler: state must be 8-byte aligned");
  DS_REQUIRE(!(!blends(k->input_kind) && fu), DS_ERR_SHAPE, "ds_karras_euler: guidance blend needs network outputs or flow fields");
  if (n == 0) return DS_OK;
  DS_REQUIRE(k->xin_copies >= 0 && k->xin_copies <= 2, DS_ERR_SHAPE, "ds_karras_euler: xin_copies %d", k->xin_copies);
  const size_t n4 = (k->xin_copies == 2 && (n & 3)) ? 0 : vec4_count(n, {x_out, xin_out, x, f, fu, eps});   // the second copy starts n floats in
  dim3 g(grid_elems(n4, n)), b(kThreads);
  hipStream_t s = ds::as_stream(stream);
  const unsigned long long* rng = reinterpret_cast<const unsigned long long*>(philox_state);
  const unsigned long long off = (unsigned long long)philox_offset;
#define L(U, E) \
  hipLaunchKernelGGL((k_euler<U, E>), g, b, 0, s, x_out, xin_out, x, f, fu, eps, rng, off, *k, dt, c_in_next, noise_coef, sqrt_abs_dt, n4, n)
  const int noise = eps ? 1 : (rng ? 2 : 0);
  if (fu) { if (noise == 1) L(true, 1); else if (noise == 2) L(true, 2); else L(true, 0); }
  else    { if (noise == 1) L(false, 1); else if (noise == 2) L(false, 2); else L(false, 0); }
#undef L
  DS_CHECK_LAUNCH("ds_karras_euler");
  return DS_OK;
}

int ds_karras_heun(float* x_out, float* xin_out, const float* x, const float* f1, const float* f1u,
                   const ds_eval_coef* k1, const float* f2, const float* f2u, const ds_eval_coef* k2, float dt,
                   float c_in_next, size_t n, void* stream) {
  DS_REQUIRE(x_out && x && f1 && f2 && k1 && k2, DS_ERR_NULL, "ds_karras_heun: NULL pointer");
  DS_REQUIRE((f1u == nullptr) == (f2u == nullptr), DS_ERR_SHAPE,
             "ds_karras_heun: f1u and f2u must both be given or both be NULL");
  DS_REQUIRE(!((!blends(k1->input_kind) || !blends(k2->input_kind)) && f1u), DS_ERR_SHAPE,
             "ds_karras_heun: guidance blend needs network outputs or flow fields");
  if (n == 0) return DS_OK;
  DS_REQUIRE(k2->xin_copies >= 0 && k2->xin_copies <= 2, DS_ERR_SHAPE, "ds_karras_heun: xin_copies %d", k2->xin_copies);
  const size_t n4 = (k2->xin_copies == 2 && (n & 3)) ? 0 : vec4_count(n, {x_out, xin_out, x, f1, f2, f1u, f2u});
  dim3 g(grid_elems(n4, n)), b(kThreads);
  hipStream_t s = ds::as_stream(stream);
  if (f1u)
    hipLaunchKernelGGL((k_heun<true>), g, b, 0, s, x_out, xin_out, x, f1, f1u, f2, f2u, *k1, *k2, dt, c_in_next, n4, n);
  else
    hipLaunchKernelGGL((k_heun<false>), g, b, 0, s, x_out, xin_out, x, f1, f1u, f2, f2u, *k1, *k2, dt, c_in_next, n4, n);
  DS_CHECK_LAUNCH("ds_karras_heun");
  return DS_OK;
}

static int launch_drift(float* out, const float* x, const float* f, const float* fu, const ds_eval_coef* k,
                        size_t n, void* stream, bool score_only, const char* who) {
  DS_REQUIRE(out && f && k, DS_ERR_NULL, "%s: NULL pointer", who);
  DS_REQUIRE(x || k->input_kind != DS_IN_NETWORK, DS_ERR_NULL, "%s: x is required for network outputs", who);
  DS_REQUIRE(!(!blends(k->input_kind) && fu), DS_ERR_SHAPE, "%s: guidance blend needs network outputs or flow fields", who);
  DS_REQUIRE(!(score_only && k->input_kind != DS_IN_NETWORK), DS_ERR_SHAPE, "%s: input is not a network output", who);
  if (n == 0) return DS_OK;
  dim3 g(grid_for((n + 3) / 4)), b(kThreads);
  hipStream_t s = ds::as_stream(stream);
  if (fu) {
    if (score_only) hipLaunchKernelGGL((k_drift<true, true>), g, b, 0, s, out, x, f, fu, *k, n);
    else hipLaunchKernelGGL((k_drift<true, false>), g, b, 0, s, out, x, f, fu, *k, n);
  } else {
    if (score_only) hipLaunchKernelGGL((k_drift<false, true>), g, b, 0, s, out, x, f, fu, *k, n);
    else hipLaunchKernelGGL((k_drift<false, false>), g, b, 0, s, out, x, f, fu, *k, n);
  }
  DS_CHECK_LAUNCH(who);
  return DS_OK;
}

int ds_karras_drift(float* d_out, const float* x, const float* f, const float* fu, const ds_eval_coef* k, size_t n,
                    void* stream) {
  return launch_drift(d_out, x, f, fu, k, n, stream, false, "ds_karras_drift");
}

int ds_karras_score(float* s_out, const float* x, const float* f, const float* fu, const ds_eval_coef* k, size_t n,
                    void* stream) {
  return launch_drift(s_out, x, f, fu, k, n, stream, true, "ds_karras_score");
}

int ds_karras_churn(float* xhat_out, float* xin_out, const float* x, const float* eps, const uint64_t* philox_state,
                    uint64_t philox_offset, float coef, float c_in, float ratio, float scale, int xin_copies, size_t n,
                    void* stream) {
  DS_REQUIRE(xhat_out && x, DS_ERR_NULL, "ds_karras_churn: NULL pointer");
  DS_REQUIRE((eps != nullptr) != (philox_state != nullptr), DS_ERR_NULL,
             "ds_karras_churn: exactly one of eps (injected noise) and philox_state (in-kernel noise) must be given");
  DS_REQUIRE((reinterpret_cast<uintptr_t>(philox_state) & 7u) == 0, DS_ERR_SHAPE, "ds_karras_churn: state must be 8-byte aligned");
  if (n == 0) return DS_OK;
  DS_REQUIRE(xin_copies >= 0 && xin_copies <= 2, DS_ERR_SHAPE, "ds_karras_churn: xin_copies %d", xin_copies);
  const size_t n4 = (xin_copies == 2 && (n & 3)) ? 0 : vec4_count(n, {xhat_out, xin_out, x, eps});
  dim3 g(grid_elems(n4, n)), b(kThreads);
  hipStream_t s = ds::as_stream(stream);
  if (eps)
    hipLaunchKernelGGL((k_churn<false>), g, b, 0, s, xhat_out, xin_out, x, eps, nullptr, 0ull, coef, c_in, ratio, scale, xin_copies, n4, n);
  else
    hipLaunchKernelGGL((k_churn<true>), g, b, 0, s, xhat_out, xin_out, x, eps,
                       reinterpret_cast<const unsigned long long*>(philox_state), (unsigned long long)philox_offset, coef,
                       c_in, ratio, scale, xin_copies, n4, n);
  DS_CHECK_LAUNCH("ds_karras_churn");
  return DS_OK;
}

int ds_karras_denoiser(float* out, const float* x, const float* f, const float* fu, float guidance,
                       float one_minus_guidance,
                       const float* c_out, const float* c_skip, int B, size_t n_per_sample, void* stream) {
  DS_REQUIRE(out && x && f && c_out && c_skip, DS_ERR_NULL, "ds_karras_denoiser: NULL pointer");
  DS_REQUIRE(B >= 0, DS_ERR_SHAPE, "ds_karras_denoiser: B < 0");
  size_t total = (size_t)B * n_per_sample;
  if (total == 0) return DS_OK;
  hipLaunchKernelGGL(k_denoiser, dim3(grid_for((total + 3) / 4)), dim3(kThreads), 0, ds::as_stream(stream), out, x, f,
                     fu, guidance, one_minus_guidance, c_out, c_skip, n_per_sample, total);
  DS_CHECK_LAUNCH("ds_karras_denoiser");
  return DS_OK;
}

int ds_mask_blend(float* out, const float* x, const float* y, const float* mask, size_t n_per_sample, int B,
                  void* stream) {
  DS_REQUIRE(out && x && y && mask, DS_ERR_NULL, "ds_mask_blend: NULL pointer");
  DS_REQUIRE(B >= 0 && n_per_sample > 0, DS_ERR_SHAPE, "ds_mask_blend: bad shape");
  if (B == 0) return DS_OK;
  const size_t total = n_per_sample * (size_t)B;
  hipLaunchKernelGGL(k_mask_blend, dim3(grid_for((total + 3) / 4)), dim3(kThreads), 0, ds::as_stream(stream), out, x, y,
                     mask, n_per_sample, total);
  DS_CHECK_LAUNCH("ds_mask_blend");
  return DS_OK;
}

int ds_axpby(float* out, const float* x, float a, const float* y, float b, size_t n, void* stream) {
  DS_REQUIRE(out && x, DS_ERR_NULL, "ds_axpby: NULL pointer");
  if (n == 0) return DS_OK;
  hipLaunchKernelGGL(k_axpby, dim3(grid_for((n + 3) / 4)), dim3(kThreads), 0, ds::as_stream(stream), out, x, a, y, b, n);
  DS_CHECK_LAUNCH("ds_axpby");
  return DS_OK;
}

int ds_div_scalar(float* out, const float* x, float s, size_t n, void* stream) {
  DS_REQUIRE(out && x, DS_ERR_NULL, "ds_div_scalar: NULL pointer");
  if (n == 0) return DS_OK;
  hipLaunchKernelGGL(k_div_scalar, dim3(grid_for((n + 3) / 4)), dim3(kThreads), 0, ds::as_stream(stream), out, x, s, n);
  DS_CHECK_LAUNCH("ds_div_scalar");
  return DS_OK;
}

int ds_batchnorm_eval(float* out, const float* x, const float* mean, const float* var, const float* weight,
                      const float* bias, float eps, float sigma, int inverse, int B, int C, int nc, size_t HW, void* stream) {
  DS_REQUIRE(out && x && mean && var, DS_ERR_NULL, "ds_batchnorm_eval: NULL pointer");
  DS_REQUIRE((weight == nullptr) == (bias == nullptr), DS_ERR_NULL, "ds_batchnorm_eval: weight and bias go together");
  DS_REQUIRE(B >= 0 && C > 0 && HW > 0 && (nc == 1 || nc == C), DS_ERR_SHAPE,
             "ds_batchnorm_eval: bad shape B=%d C=%d nc=%d (1 or C)", B, C, nc);
  DS_REQUIRE((long long)B * C < 65536, DS_ERR_SHAPE, "ds_batchnorm_eval: B*C must stay below 65536 planes");
  DS_REQUIRE(sigma != 0.0f, DS_ERR_SHAPE, "ds_batchnorm_eval: sigma = 0");
  if (B == 0) return DS_OK;
  const size_t per = (HW + 4 * kThreads - 1) / (4 * kThreads);
  dim3 g((unsigned)(per > 256 ? 256 : per), (unsigned)(B * C));
  if (inverse) hipLaunchKernelGGL((k_batchnorm<false>), g, dim3(kThreads), 0, ds::as_stream(stream), out, x, mean, var, weight, bias, eps, sigma, C, nc, HW);
  else hipLaunchKernelGGL((k_batchnorm<true>), g, dim3(kThreads), 0, ds::as_stream(stream), out, x, mean, var, weight, bias, eps, sigma, C, nc, HW);
  DS_CHECK_LAUNCH("ds_batchnorm_eval");
  return DS_OK;
}

int ds_lerp_stack(float* out, const float* x1, const float* x2, int n, size_t numel, void* stream) {
  DS_REQUIRE(out && x1 && x2, DS_ERR_NULL, "ds_lerp_stack: NULL pointer");
  DS_REQUIRE(n >= 2 && numel > 0, DS_ERR_SHAPE, "ds_lerp_stack: need n >= 2 interpolation points (got %d)", n);
  hipLaunchKernelGGL(k_lerp_stack, dim3(grid_for((numel + 3) / 4)), dim3(kThreads), 0, ds::as_stream(stream), out, x1,
                     x2, n, numel);
  DS_CHECK_LAUNCH("ds_lerp_stack");
  return DS_OK;
}

int ds_add(float* out, const float* a, const float* b, size_t n, void* stream) {
  DS_REQUIRE(out && a && b, DS_ERR_NULL, "ds_add: NULL pointer");
  if (n == 0) return DS_OK;
  const size_t n4 = vec4_count(n, {out, a, b});
  hipLaunchKernelGGL(k_add, dim3(grid_elems(n4, n)), dim3(kThreads), 0, ds::as_stream(stream), out, a, b, n4, n);
  DS_CHECK_LAUNCH("ds_add");
  return DS_OK;
}

}  // extern "C"
