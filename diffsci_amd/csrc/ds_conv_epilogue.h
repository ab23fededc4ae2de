// Shared epilogue of the MFMA convolution kernels: the wave's 64-channel x (2 rows x 32 columns)
// accumulator tile is transposed through a wave-private LDS region so that the global stores (and
// the residual loads) are 16 bytes per lane -- 16 wave-instructions of 1 KiB instead of 64 of
// 256 B -- and the optional terms cost no per-element branch.
//   out = (((acc * unscale + bias[co]) + shift[b, co]) + res1) + res2        (terms in that order)
#pragma once
#include <hip/hip_runtime.h>

#ifndef DS_AMAX_EPILOGUE
#define DS_AMAX_EPILOGUE 1      // 0: measurement builds only (tools/ab_bench.sh): what does the out_amax code cost the launches that do not use it?
#endif

namespace ds_epi {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Args {
  float* out;
  const float* bias;
  const float* shift;     // + b*shift_stride + co
  const float* res1;
  const float* res2;
  int res1_up;            // res1 is [B, Cout, H/2, W/2] and is added nearest-upsampled (x2): out += res1[.., y/2, x/2]
  float unscale;
  int shift_stride;
  int b, co_base;         // sample, first channel of the workgroup's 64-channel tile
  int y0, x0;             // first row of the WAVE's two rows, first column of the tile
  int Cout, H, W;
  // Optional per-(channel, tile) statistics of the stored values, for the normalisation that consumes
  // this tensor: tile_stats[(b*Cout + co)*ntiles + tile] = float4 (K, S, Q, n) with n valid pixels,
  // K one of the values, S = sum(x - K), Q = sum((x - K)^2).  Shifting by K keeps fp32 sums free of the
  // mean^2 cancellation; the table kernels (ds_normtab.hip) combine the tiles in fp64.
  float* tile_stats;
  int tile, ntiles;
  // ds_convup.hip only: the tile is in LOW-resolution coordinates (y0, x0; H, W are the output's) and covers output
  // rows 2y + pa
  int pa, pb;
  // Optional: max |stored value| of sample b, as float bits, merged with atomicMax into *out_amax (the slot of THIS
  // sample; zeroed by the host before the launch).  The fp16x3 kernel that consumes the tensor takes its per-sample
  // activation exponent from it (act_scale below) instead of a reduction pass over the tensor.
  unsigned* out_amax;
};

// ---- per-sample activation exponent of the fp16x3 kernels --------------------------------------------------------
// x = hi + lo in fp16 keeps 22 significand bits only while lo is a normal fp16 number, i.e. for |x| in [2^-3, 2^16):
// fp16 has 5 exponent bits where the reference's fp32 convolution (punetg.py:719-735 feeds raw user fields,
// preconditioners.py:139-161 c_in = 1) has 8.  A launch whose input is not normalised by construction therefore
// multiplies its input by 2^k in the loader, k chosen per SAMPLE so that the sample's max |x| lands in [2^13, 2^14),
// and undoes it exactly in the epilogue next to the weight scale: 2^-(wshift + k).  A block floating point with the
// sample's exponent: elements down to 2^-19 of the sample's maximum keep >= 20 bits, the absolute floor is 2^-39 of
// the maximum, the whole fp32 range of magnitudes is accepted (no overflow either).  Per sample, so a sample's result
// does not depend on what else is in the batch.
struct ActScale {
  float in_scale;     // 2^k
  float unscale;      // 2^-(wshift + k)
  float inv_scale;    // 2^-k: the fused norm + SiLU loader forms SiLU(v) * 2^k = v * rcp(2^-k (1 + exp2(-v log2 e))) (ds_h3_common.h)
};
// amax: per-sample max |x| as float bits (written by an earlier kernel), or NULL = no scaling (normalised input).
// Two stages, so that the load's round trip hides behind the kernel's first global loads: act_bits() ISSUES it right in front
// of them, act_scale_of() / act_exponent_of() consume it behind them.  A VECTOR load on purpose (the lane offset below is zero,
// but opaque to the compiler): scalar loads share one counter with the kernel-argument loads and return out of order, so the first
// wait for any kernel argument also waited for this L2 round trip in front of every workgroup's first loads (+2.4 % on the
// level-0 launches); vector loads return in order and are waited for one by one -- this one has landed when the first patch
// element has.
__device__ __forceinline__ unsigned act_bits(const unsigned* amax, int b) {
  if (!amax) return 0u;
  int z = 0;
  asm volatile("" : "+v"(z));
  return amax[b + z];
}
// The exponent k with max * 2^k in [2^13, 2^14), clamped to [lo, hi]; 0 for a zero / subnormal / non-finite maximum.
__device__ __forceinline__ int act_exponent_of(unsigned bits, int lo, int hi) {
  bits = __builtin_amdgcn_readfirstlane(bits);
  const int e = (int)((bits >> 23) & 0xffu);
  const int k = (e == 0 || e == 255) ? 0 : 140 - e;
  return k > hi ? hi : (k < lo ? lo : k);
}
__device__ __forceinline__ float pow2f(int k) { return __builtin_bit_cast(float, (unsigned)(127 + k) << 23); }   // -126 <= k <= 127
// x * 2^k by exponent arithmetic (x and the product normal floats): integer instructions, so a wave-uniform x stays in a
// scalar register (gfx950 has no scalar float multiply)
__device__ __forceinline__ float mul_pow2(float x, int k) { return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x) + ((unsigned)k << 23)); }
// 2^-(wshift + k) from in_scale = 2^k or inv_scale = 2^-k (exponent arithmetic): lets the kernels carry ONE scalar through their
// main loop and rebuild the epilogue's factor from it
__device__ __forceinline__ float unscale_from_in(float in_scale, int wshift) {
  return __builtin_bit_cast(float, ((unsigned)(254 - wshift) << 23) - __builtin_bit_cast(unsigned, in_scale));
}
__device__ __forceinline__ float unscale_from_inv(float inv_scale, int wshift) {
  return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, inv_scale) - ((unsigned)wshift << 23));
}
__device__ __forceinline__ ActScale act_scale_of(unsigned bits, int wshift) {
  // 2^k and 2^-(wshift + k) both stay normal floats
  const int hi = 126 - wshift < 126 ? 126 - wshift : 126, lo = -126 - wshift > -126 ? -126 - wshift : -126;
  const int k = act_exponent_of(bits, lo, hi);
  ActScale s;
  s.in_scale = __builtin_bit_cast(float, (unsigned)(127 + k) << 23);
  s.unscale = __builtin_bit_cast(float, (unsigned)(127 - wshift - k) << 23);
  s.inv_scale = __builtin_bit_cast(float, (unsigned)(127 - k) << 23);
  return s;
}

// wave-wide maximum of non-negative values, merged into the sample's slot
__device__ __forceinline__ void commit_amax(unsigned* slot, float m) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(slot, __builtin_bit_cast(unsigned, m));
}
__device__ __forceinline__ float abs_max4(f32x4 v) {
  return fmaxf(fmaxf(__builtin_fabsf(v.x), __builtin_fabsf(v.y)), fmaxf(__builtin_fabsf(v.z), __builtin_fabsf(v.w)));
}

// sum over the 8 lanes of an aligned lane octet, then over the pair of octets of a 16-lane row
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true));  // row_ror:8
  return v;
}
// lane 0 of each 16-lane row, broadcast to the row
__device__ __forceinline__ float row16_first(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x150, 0xF, 0xF, true));  // row_share:0
}

// W16 = false: the wave's 2 x 32 positions are 2 rows x 32 columns (row = y0 + r, column = x0 + x);
// W16 = true (narrow feature maps): they are 4 rows x 16 columns (row = y0 + 2r + x/16, column = x0 + x%16).
// MT:    32-channel tiles the wave owns (2: the whole 64-channel tile; 1: half of it -- the 8-wave kernels; then
//        e.co_base and bs point at the wave's own 32 channels).
// tile:  wave-private LDS scratch of 32*MT*2*32 floats (16 KiB for MT = 2), 16-byte aligned.
// bs:    LDS array [2][64]: bias and shift of the workgroup's 64 channels (zeros where absent),
//        written by the caller before the last barrier of the main loop.
// Phase 2 of the epilogue, from a wave-private LDS tile laid out [co][row r][32 positions] (see store_tile).
// The residual vectors of phase 2, loaded BEFORE phase 1 so that their latency hides behind the transposition (loaded inside
// the row loop they cost one to two serial memory round trips per batch of four store instructions: a level-0 launch with a
// residual took 284 us against 261 us without).  Whole-tile kernels only (MT = 2): 64 registers per residual.
template <int MT> struct Residuals {
  f32x4 r1[8 * MT], r2[8 * MT];          // res1_up: .xy = the two low-resolution columns
};
template <> struct Residuals<0> {};

// lane-constant parts of phase 2's 16 store instructions: segment 8*j + lane/8 -> channel 4*j + lane/16, row (lane/8) & 1
template <bool W16> struct RowPlan {
  int p4, gx, yq, gy_lane;
  bool pix_ok;
  size_t idx_lane, idx_step;
  __device__ __forceinline__ RowPlan(const Args& e) {
    const int lane = threadIdx.x & 63;
    p4 = 4 * (lane & 7);
    gx = e.x0 + (W16 ? (p4 & 15) : p4);
    yq = W16 ? (p4 >> 4) : 0;
    const int r_lane = (lane >> 3) & 1;
    gy_lane = e.y0 + (W16 ? 2 * r_lane + yq : r_lane);
    pix_ok = gy_lane < e.H && gx < e.W;
    const size_t plane = (size_t)e.H * e.W;
    idx_lane = ((size_t)e.b * e.Cout + e.co_base + (lane >> 4)) * plane + (size_t)gy_lane * e.W + gx;
    idx_step = 4 * plane;                 // four channels per instruction
  }
};

template <bool W16, int MT>
__device__ __forceinline__ void load_residuals(Residuals<MT>& R, const Args& e) {
  if ((e.W & 3) != 0 || (!e.res1 && !e.res2)) return;
  const int lane = threadIdx.x & 63;
  const RowPlan<W16> rp(e);
#pragma unroll
  for (int j = 0; j < 8 * MT; ++j) {
    const bool ok = rp.pix_ok && (e.co_base + 4 * j + (lane >> 4) < e.Cout);
    const size_t idx = ok ? rp.idx_lane + (size_t)j * rp.idx_step : (size_t)0;
    if (e.res1) {
      if (e.res1_up) {                              // 4 output columns = 2 low-resolution columns
        const size_t lidx = ok ? (((size_t)e.b * e.Cout + e.co_base + 4 * j + (lane >> 4)) * (e.H >> 1) + (rp.gy_lane >> 1)) * (e.W >> 1) + (rp.gx >> 1) : (size_t)0;
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        const f32x2_t lo = *reinterpret_cast<const f32x2_t*>(e.res1 + lidx);
        R.r1[j] = f32x4{lo[0], lo[1], 0.f, 0.f};
      } else {
        R.r1[j] = *reinterpret_cast<const f32x4*>(e.res1 + idx);
      }
    }
    if (e.res2) R.r2[j] = *reinterpret_cast<const f32x4*>(e.res2 + idx);
  }
}

template <bool W16, int MT, bool PREFETCHED>
__device__ __forceinline__ void store_tile_rows(float* tile, const Args& e, const Residuals<PREFETCHED ? MT : 0>& R);

template <bool W16 = false, int MT = 2>
__device__ __forceinline__ void store_tile(const f32x16 (&acc)[MT][2], float* tile, const float* bs, const Args& e) {
  const int lane = threadIdx.x & 63;
  const int li = lane & 31, lh = lane >> 5;
  constexpr bool PF = MT == 2;
  Residuals<PF ? MT : 0> R;
  if constexpr (PF) load_residuals<W16, MT>(R, e);
  // phase 1: accumulator layout -> [co][row][x]
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int co = 32 * m + (q & 3) + 8 * (q >> 2) + 4 * lh;
      const float bsv = bs[co] + bs[64 + co];          // bias + shift, added once per channel (the kernel is vector-issue / power bound)
#pragma unroll
      for (int r = 0; r < 2; ++r) tile[(co * 2 + r) * 32 + li] = __builtin_fmaf(acc[m][r][q], e.unscale, bsv);   // unscale is a power of two: the product is exact, one instruction
    }
  store_tile_rows<W16, MT, PF>(tile, e, R);
}

// The same for the 16x16 accumulator tiles of v_mfma_f32_16x16x32_f16: acc[m][n], m = 16-channel tile, n = 16-position
// tile (positions 16*(n&1) .. +15 of row r = n>>1); register q of lane l holds channel 16m + 4(l>>4) + q, position l&15.
// MT: 32-channel tiles of the wave (2: all 64 channels; 1: the eight-wave kernels, bs / e.co_base at the wave's own 32)
template <bool W16, int MT = 2>
__device__ __forceinline__ void store_tile16(const f32x4 (&acc)[2 * MT][4], float* tile, const float* bs, const Args& e) {
  const int lane = threadIdx.x & 63;
  const int i = lane & 15, g = lane >> 4;
  const float unscale = e.unscale;
  constexpr bool PF = MT == 2;
  Residuals<PF ? MT : 0> R;
  if constexpr (PF) load_residuals<W16, MT>(R, e);
#pragma unroll
  for (int m = 0; m < 2 * MT; ++m)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int co = 16 * m + 4 * g + q;
      const float bsv = bs[co] + bs[64 + co];
#pragma unroll
      for (int n = 0; n < 4; ++n) tile[(co * 2 + (n >> 1)) * 32 + 16 * (n & 1) + i] = __builtin_fmaf(acc[m][n][q], unscale, bsv);   // exact product (power of two)
    }
  store_tile_rows<W16, MT, PF>(tile, e, R);
}

template <bool W16, int MT, bool PREFETCHED>
__device__ __forceinline__ void store_tile_rows(float* tile, const Args& e, const Residuals<PREFETCHED ? MT : 0>& R) {
  const int lane = threadIdx.x & 63;
  // phase 2: 16 bytes per lane; lane -> (segment = 8*it + lane/8, quarter = lane%8)
  const RowPlan<W16> rp(e);
  const int p4 = rp.p4, gx = rp.gx, yq = rp.yq;
  const size_t plane = (size_t)e.H * e.W;
  const bool stats = e.tile_stats != nullptr;
  if ((e.W & 3) == 0) {
    const bool pix_ok = rp.pix_ok;
    const size_t idx_lane = rp.idx_lane, idx_step = rp.idx_step;
    const float cnt_row = stats ? row16_sum(pix_ok ? 4.f : 0.f) : 0.f;     // valid pixels of a channel in this wave: the same for every channel
    const bool want_amax = DS_AMAX_EPILOGUE && e.out_amax != nullptr;
    float amax = 0.f;
    float sK[8 * MT], ssum[8 * MT], ssq[8 * MT], scnt[8 * MT];   // per (half, k): channel 4*(4*half+k) + lane/16, valid in every lane of the row
#pragma unroll
    for (int half = 0; half < 2 * MT; ++half) {       // batches of 4 wave-instructions
      f32x4 v[4], r1[4], r2[4];
      size_t idx[4];
      bool ok[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int j = half * 4 + k;
        const int seg = j * 8 + (lane >> 3);
        const int co = seg >> 1;
        ok[k] = pix_ok && (e.co_base + co < e.Cout);
        idx[k] = ok[k] ? idx_lane + (size_t)j * idx_step : (size_t)0;
        v[k] = *reinterpret_cast<const f32x4*>(&tile[seg * 32 + p4]);
      }
      if constexpr (PREFETCHED) {
        if (e.res1) {
          if (e.res1_up) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { const f32x4 lo = R.r1[half * 4 + k]; r1[k] = f32x4{lo[0], lo[0], lo[1], lo[1]}; }
          } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) r1[k] = R.r1[half * 4 + k];
          }
        }
        if (e.res2) {
#pragma unroll
          for (int k = 0; k < 4; ++k) r2[k] = R.r2[half * 4 + k];
        }
      } else {
      if (e.res1) {
        if (e.res1_up) {                              // 4 output columns = 2 low-resolution columns
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int seg = (half * 4 + k) * 8 + (lane >> 3);
            const int co = seg >> 1, r = seg & 1;
            const int gy = e.y0 + (W16 ? 2 * r + yq : r);
            const size_t lidx = ok[k] ? (((size_t)e.b * e.Cout + e.co_base + co) * (e.H >> 1) + (gy >> 1)) * (e.W >> 1) + (gx >> 1) : (size_t)0;
            typedef float f32x2_t __attribute__((ext_vector_type(2)));
            const f32x2_t lo = *reinterpret_cast<const f32x2_t*>(e.res1 + lidx);
            r1[k] = f32x4{lo[0], lo[0], lo[1], lo[1]};
          }
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k) r1[k] = *reinterpret_cast<const f32x4*>(e.res1 + idx[k]);
        }
      }
      if (e.res2) {
#pragma unroll
        for (int k = 0; k < 4; ++k) r2[k] = *reinterpret_cast<const f32x4*>(e.res2 + idx[k]);
      }
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
      if (want_amax) {
#pragma unroll
        for (int k = 0; k < 4; ++k) amax = ok[k] ? fmaxf(amax, abs_max4(v[k])) : amax;
      }
      if (stats) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          // the row's 16 lanes hold both rows of channel seg>>1 (lanes 0-7: r = 0, 8-15: r = 1); lane 0
          // is the wave's first pixel of the channel: valid whenever any pixel of the wave is
          const float K = row16_first(ok[k] ? v[k].x : 0.f);
          const f32x4 d = v[k] - K;
          const float sv = ok[k] ? (d.x + d.y) + (d.z + d.w) : 0.f;
          const float qv = ok[k] ? (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w) : 0.f;
          sK[half * 4 + k] = K;
          ssum[half * 4 + k] = row16_sum(sv);
          ssq[half * 4 + k] = row16_sum(qv);
          scnt[half * 4 + k] = (e.co_base + ((half * 4 + k) * 4 + (lane >> 4)) < e.Cout) ? cnt_row : 0.f;
        }
      }
    }
    if (want_amax) commit_amax(e.out_amax, amax);
    if (stats) {                                      // all reads of the wave's tile are done: reuse its head as [32*MT co][4]
      if ((lane & 15) == 0) {
#pragma unroll
        for (int it = 0; it < 8 * MT; ++it) {
          f32x4 o = {sK[it], ssum[it], ssq[it], scnt[it]};
          *reinterpret_cast<f32x4*>(&tile[4 * (4 * it + (lane >> 4))]) = o;
        }
      }
    }
  } else {
    // ragged width: element-wise, compact loop (correctness path for odd shapes); one channel per iteration
    float amax = 0.f;
    for (int i = lane; i < 32 * MT * 2 * 32; i += 64) {
      const int co = i >> 6, r = (i >> 5) & 1, x = i & 31;
      const int gy = e.y0 + (W16 ? 2 * r + (x >> 4) : r), gxx = e.x0 + (W16 ? (x & 15) : x);
      float v = 0.f;
      const bool okv = e.co_base + co < e.Cout && gy < e.H && gxx < e.W;
      if (okv) {
        const size_t idx = ((size_t)e.b * e.Cout + e.co_base + co) * plane + (size_t)gy * e.W + gxx;
        v = tile[i];
        if (e.res1) v = v + (e.res1_up ? e.res1[(((size_t)e.b * e.Cout + e.co_base + co) * (e.H >> 1) + (gy >> 1)) * (e.W >> 1) + (gxx >> 1)]
                                       : e.res1[idx]);
        if (e.res2) v = v + e.res2[idx];
        e.out[idx] = v;
        amax = fmaxf(amax, __builtin_fabsf(v));
      }
      if (stats) {
        const float K = __shfl(v, 0, 64);                              // the wave's first pixel of the channel
        const float d = okv ? v - K : 0.f;
        float sv = d, qv = d * d, nv = okv ? 1.f : 0.f;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { sv += __shfl_xor(sv, o, 64); qv += __shfl_xor(qv, o, 64); nv += __shfl_xor(nv, o, 64); }
        // elements 4co .. 4co+3 of the tile belong to channel co' = co/16 <= co and were consumed
        // (co = 0: by this very iteration's reads, which precede the write in program order)
        if (lane == 0) { tile[4 * co] = K; tile[4 * co + 1] = sv; tile[4 * co + 2] = qv; tile[4 * co + 3] = nv; }
      }
    }
    if (DS_AMAX_EPILOGUE && e.out_amax) commit_amax(e.out_amax, amax);
  }
}

// Second half of the statistics: combine the four row-waves' partials of every channel (at the head of each
// wave's tile region, `wave_stride` floats apart) about one shift and store them.  Call after a
// __syncthreads().  MT = 2: four waves, each with [64][4]; MT = 1: eight waves, waves 4..7 hold channels 32..63
// ([32][4] each).  e.co_base: first channel of the WORKGROUP's 64-channel tile.
template <int MT = 2>
__device__ __forceinline__ void store_tile_stats(const float* tiles, int wave_stride, const Args& e, int t = -1) {
  if (t < 0) t = threadIdx.x;                             // t: the channel this thread combines (callers with several channel tiles pass it)
  if (t < 64 && e.co_base + t < e.Cout) {
    f32x4 p[4];
    const int w0 = MT == 1 ? 4 * (t >> 5) : 0, tl = MT == 1 ? (t & 31) : t;
#pragma unroll
    for (int w = 0; w < 4; ++w) p[w] = *reinterpret_cast<const f32x4*>(&tiles[(w0 + w) * wave_stride + 4 * tl]);
    const float K = p[0][0];                              // wave 0 owns the tile's first rows: valid if the tile is
    float S = 0.f, Q = 0.f, n = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float dk = p[w][3] > 0.f ? p[w][0] - K : 0.f;
      S += p[w][1] + p[w][3] * dk;
      Q += p[w][2] + 2.f * dk * p[w][1] + p[w][3] * dk * dk;
      n += p[w][3];
    }
    f32x4 o = {K, S, Q, n};
    *reinterpret_cast<f32x4*>(e.tile_stats + (((size_t)e.b * e.Cout + e.co_base + t) * e.ntiles + e.tile) * 4) = o;
  }
}

// bs[0..63] = bias (or 0), bs[64..127] = shift row (or 0) for the workgroup's channel tile, in two halves so that the load
// rides behind the first patch's loads instead of in front of them (a load + wait + LDS store at the top of the kernel put one
// more memory round trip into every workgroup's prologue: 2.8 us before the patch loads were even issued at level 0).
// cots channel tiles: threads 128c .. 128c+127 serve tile c (the caller passes THAT tile's co_base), bs is [cots][2][64]
__device__ __forceinline__ float fetch_bias_shift(const float* bias, const float* shift, int shift_stride, int b, int co_base,
                                                   int Cout, int cots = 1) {
  const int t = threadIdx.x;
  const int co = co_base + (t & 63);
  float v = 0.f;
  if (t < 128 * cots && co < Cout) {
    if ((t & 127) < 64) v = bias ? bias[co] : 0.f;
    else v = shift ? shift[(size_t)b * shift_stride + co] : 0.f;
  }
  return v;
}
__device__ __forceinline__ void commit_bias_shift(float* bs, float v, int cots = 1) {
  if (threadIdx.x < 128 * cots) bs[threadIdx.x] = v;
}

}  // namespace ds_epi
