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
  // three instructions per pair: the packed conversion, then lo = fp16(a - hi) by the mixed-precision FMA reading the stored
  // high piece itself (a - hi is exact in fp32; one rounding to fp16)
  unsigned hp, lp;
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hp) : "v"(a), "v"(b));
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lp) : "v"(hp), "v"(a));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lp) : "v"(hp), "v"(b));
  hi = hp;
  lo = lp;
}

// x * sigmoid(x) with the hardware exp2 / rcp (1 ulp each): the fused-normalisation loader's
// activation (the standalone norm kernels use expf and an IEEE division; the difference is ~2e-7
// relative, far inside the path's stated 1e-5 tolerance).
__device__ __forceinline__ float fast_silu(float v) {
  return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.44269504088896341f));
}

}  // namespace ds_h3
