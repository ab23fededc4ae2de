// Shared epilogue of the MFMA convolution kernels: the wave's 64-channel x (2 rows x 32 columns)
// accumulator tile is transposed through a wave-private LDS region so that the global stores (and
// the residual loads) are 16 bytes per lane -- 16 wave-instructions of 1 KiB instead of 64 of
// 256 B -- and the optional terms cost no per-element branch.
//   out = (((acc * unscale + bias[co]) + shift[b, co]) + res1) + res2        (terms in that order)
#pragma once
#include <hip/hip_runtime.h>

namespace ds_epi {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Args {
  float* out;
  const float* bias;
  const float* shift;     // + b*shift_stride + co
  const float* res1;
  const float* res2;
  float unscale;
  int shift_stride;
  int b, co_base;         // sample, first channel of the workgroup's 64-channel tile
  int y0, x0;             // first row of the WAVE's two rows, first column of the tile
  int Cout, H, W;
};

// W16 = false: the wave's 2 x 32 positions are 2 rows x 32 columns (row = y0 + r, column = x0 + x);
// W16 = true (narrow feature maps): they are 4 rows x 16 columns (row = y0 + 2r + x/16, column = x0 + x%16).
// tile:  wave-private LDS scratch of 64*2*32 floats (16 KiB), 16-byte aligned.
// bs:    LDS array [2][64]: bias and shift of the workgroup's 64 channels (zeros where absent),
//        written by the caller before the last barrier of the main loop.
template <bool W16 = false>
__device__ __forceinline__ void store_tile(const f32x16 (&acc)[2][2], float* tile, const float* bs, const Args& e) {
  const int lane = threadIdx.x & 63;
  const int li = lane & 31, lh = lane >> 5;
  // phase 1: accumulator layout -> [co][row][x]
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int co = 32 * m + (q & 3) + 8 * (q >> 2) + 4 * lh;
      const float bv = bs[co], sv = bs[64 + co];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        float v = acc[m][r][q] * e.unscale;
        v = v + bv;
        v = v + sv;
        tile[(co * 2 + r) * 32 + li] = v;
      }
    }
  // phase 2: 16 bytes per lane; lane -> (segment = 8*it + lane/8, quarter = lane%8)
  const int p4 = 4 * (lane & 7);
  const int gx = e.x0 + (W16 ? (p4 & 15) : p4);
  const int yq = W16 ? (p4 >> 4) : 0;
  const size_t plane = (size_t)e.H * e.W;
  if ((e.W & 3) == 0) {
#pragma unroll
    for (int half = 0; half < 4; ++half) {            // 4 batches of 4 wave-instructions
      f32x4 v[4], r1[4], r2[4];
      size_t idx[4];
      bool ok[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int seg = (half * 4 + k) * 8 + (lane >> 3);
        const int co = seg >> 1, r = seg & 1;
        const int gy = e.y0 + (W16 ? 2 * r + yq : r);
        ok[k] = (e.co_base + co < e.Cout) && gy < e.H && gx < e.W;
        idx[k] = ok[k] ? ((size_t)e.b * e.Cout + e.co_base + co) * plane + (size_t)gy * e.W + gx : (size_t)0;
        v[k] = *reinterpret_cast<const f32x4*>(&tile[seg * 32 + p4]);
      }
      if (e.res1) {
#pragma unroll
        for (int k = 0; k < 4; ++k) r1[k] = *reinterpret_cast<const f32x4*>(e.res1 + idx[k]);
      }
      if (e.res2) {
#pragma unroll
        for (int k = 0; k < 4; ++k) r2[k] = *reinterpret_cast<const f32x4*>(e.res2 + idx[k]);
      }
      if (e.res1) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = v[k] + r1[k];
      }
      if (e.res2) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = v[k] + r2[k];
      }
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (ok[k]) *reinterpret_cast<f32x4*>(e.out + idx[k]) = v[k];
    }
  } else {
    // ragged width: element-wise, compact loop (correctness path for odd shapes)
    for (int i = lane; i < 64 * 2 * 32; i += 64) {
      const int co = i >> 6, r = (i >> 5) & 1, x = i & 31;
      const int gy = e.y0 + (W16 ? 2 * r + (x >> 4) : r), gxx = e.x0 + (W16 ? (x & 15) : x);
      if (e.co_base + co < e.Cout && gy < e.H && gxx < e.W) {
        const size_t idx = ((size_t)e.b * e.Cout + e.co_base + co) * plane + (size_t)gy * e.W + gxx;
        float v = tile[i];
        if (e.res1) v = v + e.res1[idx];
        if (e.res2) v = v + e.res2[idx];
        e.out[idx] = v;
      }
    }
  }
}

// Fill bs[0..63] = bias (or 0), bs[64..127] = shift row (or 0) for the workgroup's channel tile.
__device__ __forceinline__ void load_bias_shift(float* bs, const float* bias, const float* shift, int shift_stride,
                                                int b, int co_base, int Cout) {
  const int t = threadIdx.x;
  if (t < 128) {
    const int co = co_base + (t & 63);
    float v = 0.f;
    if (co < Cout) {
      if (t < 64) v = bias ? bias[co] : 0.f;
      else v = shift ? shift[(size_t)b * shift_stride + co] : 0.f;
    }
    bs[t] = v;
  }
}

}  // namespace ds_epi
