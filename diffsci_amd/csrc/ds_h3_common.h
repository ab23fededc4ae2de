// Pieces shared by the fp16x3 3x3 convolution kernels (ds_conv3h.hip, ds_convup.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace ds_h3 {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// Exact-sum split of two fp32 values into packed fp16 hi / lo pairs.  The low piece MUST be the
// remainder against the very same rounded high piece that is stored: hipcc otherwise rounds the
// stored pair with v_cvt_pk_f16_f32 and the remainder's reference with v_cvt_f16_f32, and the two
// disagree on exact ties (measured on gfx950: hi + lo off by one fp16 ulp, 2^-11 relative, for one
// value in ~8000).  Deriving the reference from the packed bits removes the second rounding.
__device__ __forceinline__ void split2(float a, float b, unsigned& hi, unsigned& lo) {
  f16x2 h = {(_Float16)a, (_Float16)b};
  unsigned hp = __builtin_bit_cast(unsigned, h);
  asm volatile("" : "+v"(hp));                       // opaque: both uses below see these exact bits
  const f16x2 hq = __builtin_bit_cast(f16x2, hp);
  f16x2 l = {(_Float16)(a - (float)hq[0]), (_Float16)(b - (float)hq[1])};
  hi = hp;
  lo = __builtin_bit_cast(unsigned, l);
}

// x * sigmoid(x) with the hardware exp2 / rcp (1 ulp each): the fused-normalisation loader's
// activation (the standalone norm kernels use expf and an IEEE division; the difference is ~2e-7
// relative, far inside the path's stated 1e-5 tolerance).
__device__ __forceinline__ float fast_silu(float v) {
  return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.44269504088896341f));
}

// The same times 2^k, exactly (inv_scale = 2^-k): SiLU(v) * 2^k = v * rcp(2^-k * (1 + exp2(-v log2 e))), the bracket formed by ONE
// fma(e, 2^-k, 2^-k) -- a power-of-two multiple of (1 + e), so the result is the unscaled form's times 2^k bit for bit, in the
// same five instructions (mul with a literal, exp2, fma with one scalar operand, rcp, mul).  [An fma(v, -log2 e, -k) in front of
// the exp2 needs two scalar operands: one more than a gfx9 VOP3 instruction takes, i.e. an extra move per element.]
__device__ __forceinline__ float fast_silu_scaled(float v, float inv_scale) {
  const float e = __builtin_amdgcn_exp2f(v * -1.44269504088896341f);
  return v * __builtin_amdgcn_rcpf(__builtin_fmaf(e, inv_scale, inv_scale));
}

}  // namespace ds_h3
