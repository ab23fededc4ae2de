// Nearest-neighbour x2 upsampling followed by a 3x3 convolution (UpSampler, commonlayers.py; ADM "up" blocks,
// adm.py:292-349), evaluated at the LOW resolution with fp32 accuracy on the fp16 matrix cores (the fp16x3
// split of ds_conv3h.hip).
//
// Output pixel (2y + a, 2x + b) reads upsampled rows 2y + a + ky - 1, i.e. low-resolution rows
// y + floor((a + ky - 1) / 2): only TWO distinct rows (y + a - 1 and y + a), and likewise two columns.  For each
// of the four output parities (a, b) the 3x3 kernel therefore collapses to a 2x2 kernel on the low-resolution
// image whose taps are sums of the original ones,
//     a = 0: rows {ky 0} , {ky 1, 2}        a = 1: rows {ky 0, 1} , {ky 2}          (columns alike with b, kx)
// 16 multiply-adds per output instead of 36 when the generic kernel gathers the upsampled patch: 2.25x fewer
// MFMAs.  The tap sums are formed once at pack time (in fp64, rounded to fp32: the reference's
// w1*x + w2*x and our (w1 + w2)*x differ by one fp32 rounding of the weight).  Zero / periodic padding of the
// upsampled image is exactly zero / periodic padding of the low-resolution one.
//
// One workgroup = one ROW parity a x 64 output channels x (8 x 32 | 16 x 16) LOW-resolution pixels, BOTH column
// parities (two accumulator sets, 128 registers): the two share the staged patch (columns x-1, x, x+1) and their
// outputs interleave along a row, so the epilogue writes whole 16-byte vectors of contiguous output columns.  (A first
// version ran one workgroup per (a, b): its stride-2 dword stores and residual loads cost 1.7 % of config 2 end to
// end, measured by pointing them at contiguous addresses.)  Same pipeline as ds_conv3h.hip with a step =
// (chunk of 16 input channels, row tap s, column parity b), two column-tap MFMA blocks per step:
//     X  [2 buffers][piece 2][h 2][9*34 | 17*18 positions][8 ci] fp16                  2 x 19,584 B
//     W  [3-slot ring][piece 2][t 2][h 2][64 co][8 ci] fp16  (8 KiB slabs by LDS-DMA)   3 x  8,192 B
// The patch of chunk c+1 is fetched at step (c, 0), split and stored at the end of (c, 2), published by that step's
// barrier and first read by the operand prefetch at the end of (c, 3).
#include "ds_common.h"
#include "ds_conv_epilogue.h"
#include "ds_h3_common.h"

namespace {

using ds_epi::f32x16;
using ds_epi::f32x4;
using ds_h3::u32x4;
using ds_h3::f16x8;
using ds_h3::split2;
using ds_h3::fast_silu;

constexpr int COT = 64, NT = 256, KC = 16;
template <bool W16> struct Geo {
  static constexpr int TH = W16 ? 16 : 8, TW = W16 ? 16 : 32;
  static constexpr int PH = TH + 1, PW = TW + 2, NPOS = PH * PW;     // 306 either way: one halo row, two halo columns
};
constexpr int XBUF_VEC = 2 * 2 * 306;                       // 16-byte vectors per X buffer
constexpr int WSLAB_VEC = 2 * 2 * 2 * COT;                  // per (chunk, s, b) slab: 512
constexpr int WDMA = WSLAB_VEC / 64 / 4;                    // LDS-DMA wave-instructions per wave: 2
constexpr int STAGE_BYTES = (2 * XBUF_VEC + 3 * WSLAB_VEC) * 16;   // 63,744
constexpr int EPI_BYTES = 4 * 64 * 2 * 32 * 4;                     // the epilogue's transpose: 16 KiB per wave
constexpr int MAIN_BYTES = STAGE_BYTES > EPI_BYTES ? STAGE_BYTES : EPI_BYTES;
constexpr int STAT_BYTES = 4 * 64 * 4 * 4;                         // per wave [64 co][K, S, Q, n]
constexpr int LDS_BYTES = MAIN_BYTES + 128 * 4 + STAT_BYTES;       // 70,144
// 16x16x32 variant (S16): K = 32 of one instruction = the step's two column taps x 16 channels, so a step is ONE group of
// 48 MFMAs.  As in ds_conv3h.hip the two 8-channel halves of a piece must sit a multiple of 256 B apart (both land in one
// ds_read_b128 lane group): 14 pad vectors after each 306-position image.
constexpr int HS16 = 320;
constexpr int XBUF_VEC16 = 2 * 2 * HS16;
constexpr int STAGE_BYTES16 = (2 * XBUF_VEC16 + 3 * WSLAB_VEC) * 16;   // 65,536
constexpr int LDS_BYTES16 = STAGE_BYTES16 + 128 * 4 + STAT_BYTES;      // 71,936
static_assert(STAGE_BYTES16 >= EPI_BYTES && 2 * LDS_BYTES16 <= 160 * 1024, "two workgroups per CU");

struct UpArgs {
  float* out;
  const float* in;        // [B, Cin, Hl, Wl]
  const u32x4* wp;
  const float* bias;
  const float* shift;
  const float* res1;
  const float* res2;
  const float* prenorm;   // as ds_conv2d_h3, over the LOW-resolution input
  float* tile_stats;
  const unsigned* in_amax;   // per-sample max |input| (float bits) -> activation exponent (ds_conv_epilogue.h), or NULL
  unsigned* out_amax;        // per-sample max |output| slots, or NULL
  int wshift;
  int shift_stride;
  int res1_up;
  int B, Cin, Cout, Hl, Wl;
  int tiles_x, tiles_y, n_cot, n_chunks;
  unsigned tiles_x_magic;
};

struct Frags { f16x8 a[2][2]; f16x8 b[2][2]; };   // [piece][m] weights, [piece][r] input

// Epilogue of one 32-channel half (m) of the wave's tile: both column parities interleaved through the wave's 16 KiB
// LDS region as [co 32][r 2][64 output columns], then 16-byte stores / residual loads of contiguous output columns.
//   W16 = false: (r, p) -> low-resolution row y0 + r,            output column 2*x0 + p        (p = 2*li + b < 64)
//   W16 = true:            low-resolution row y0 + 2r + (p >> 5), output column 2*x0 + (p & 31)
// Per-channel shifted statistics of the stored values go to stat[(32m + co)*4 .. +3] = (K, S, Q, n).
//   SWZ (the 16x16x32 accumulators' phase 1): segments of channels with bit 2 set hold their two 32-float halves swapped, which
//   spreads that layout's four 16-lane write groups over both halves of the banks.
template <bool W16, bool SWZ>
__device__ __forceinline__ void store_half_rows(int m, const float* tile, float* stat, const ds_epi::Args& e);

template <bool W16>
__device__ __forceinline__ void store_half(const f32x16 (&acc0)[2], const f32x16 (&acc1)[2], int m, float* tile,
                                           const float* bs, float* stat, const ds_epi::Args& e) {
  const int lane = threadIdx.x & 63;
  const int li = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int co = (q & 3) + 8 * (q >> 2) + 4 * lh;                 // within the half
    const float bv = bs[32 * m + co], sv = bs[64 + 32 * m + co];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      float v0 = __builtin_fmaf(acc0[r][q], e.unscale, bv), v1 = __builtin_fmaf(acc1[r][q], e.unscale, bv);   // exact product (power of two)
      v0 = v0 + sv; v1 = v1 + sv;
      typedef float f32x2_t __attribute__((ext_vector_type(2)));
      *reinterpret_cast<f32x2_t*>(&tile[((co * 2 + r) * 32 + li) * 2]) = f32x2_t{v0, v1};
    }
  }
  store_half_rows<W16, false>(m, tile, stat, e);
}

// The same for the 16x16x32 accumulators acc[column parity][m16 4][n 4]: lane l holds channels 16*m16 + 4*(l >> 4) + q of pixel
// l & 15 of pixel group n, where n = (r, 16-pixel half of the 32-pixel segment) -- the 32x32 form's (r, li) with li = 16*(n & 1) + l%16.
template <bool W16>
__device__ __forceinline__ void store_half16(const f32x4 (&acc0)[4][4], const f32x4 (&acc1)[4][4], int m, float* tile,
                                             const float* bs, float* stat, const ds_epi::Args& e) {
  const int lane = threadIdx.x & 63;
  const int l16 = lane & 15, g = lane >> 4;
#pragma unroll
  for (int mm = 0; mm < 2; ++mm)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int co = 16 * mm + 4 * g + q;                             // within the half; (co >> 2) & 1 == g & 1
      const float bv = bs[32 * m + co], sv = bs[64 + 32 * m + co];
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        float v0 = __builtin_fmaf(acc0[2 * m + mm][n][q], e.unscale, bv), v1 = __builtin_fmaf(acc1[2 * m + mm][n][q], e.unscale, bv);
        v0 = v0 + sv; v1 = v1 + sv;
        const int li = (16 * (n & 1) + l16) ^ (16 * (g & 1));
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        *reinterpret_cast<f32x2_t*>(&tile[((co * 2 + (n >> 1)) * 32 + li) * 2]) = f32x2_t{v0, v1};
      }
    }
  store_half_rows<W16, true>(m, tile, stat, e);
}

template <bool W16, bool SWZ>
__device__ __forceinline__ void store_half_rows(int m, const float* tile, float* stat, const ds_epi::Args& e) {
  const int lane = threadIdx.x & 63;
  // 16 lanes per (co, r) segment of 64 floats; lane -> (segment = 4*it + lane/16, vector = lane%16)
  const int vq = lane & 15;
  const int p4 = 4 * vq;
  const int gx = 2 * e.x0 + (W16 ? (p4 & 31) : p4);
  const int yq = W16 ? (p4 >> 5) : 0;
  const size_t plane = (size_t)e.H * e.W;
  const bool stats = e.tile_stats != nullptr;
  float amax = 0.f;
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    const int seg = it * 4 + (lane >> 4);
    const int co = seg >> 1, r = seg & 1;
    const int gyl = e.y0 + (W16 ? 2 * r + yq : r);                  // low-resolution row
    const int gy = 2 * gyl + e.pa;
    const bool ok = e.co_base + 32 * m + co < e.Cout;
    const size_t idx = ok ? ((size_t)e.b * e.Cout + e.co_base + 32 * m + co) * plane + (size_t)gy * e.W + gx : (size_t)0;
    f32x4 v = *reinterpret_cast<const f32x4*>(&tile[seg * 64 + (SWZ ? p4 ^ (32 * ((co >> 2) & 1)) : p4)]);
    if (e.res1) {
      if (e.res1_up) {                                              // [B, Cout, H/2, W/2]: two low-resolution columns
        const size_t lidx = ok ? (((size_t)e.b * e.Cout + e.co_base + 32 * m + co) * (e.H >> 1) + gyl) * (e.W >> 1) + (gx >> 1) : (size_t)0;
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        const f32x2_t lo = *reinterpret_cast<const f32x2_t*>(e.res1 + lidx);
        v = v + f32x4{lo[0], lo[0], lo[1], lo[1]};
      } else {
        v = v + *reinterpret_cast<const f32x4*>(e.res1 + idx);
      }
    }
    if (e.res2) v = v + *reinterpret_cast<const f32x4*>(e.res2 + idx);
    if (ok) *reinterpret_cast<f32x4*>(e.out + idx) = v;
    if (e.out_amax) amax = ok ? fmaxf(amax, ds_epi::abs_max4(v)) : amax;
    if (stats) {
      // lanes 0-15 / 16-31 of a 32-lane half hold rows r = 0 / 1 of one channel; the shift K is the channel's first value
      const float K = __shfl(ok ? v.x : 0.f, lane & 32, 64);
      const f32x4 d = v - K;
      float s1 = ok ? (d.x + d.y) + (d.z + d.w) : 0.f;
      float s2 = ok ? (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w) : 0.f;
      s1 = ds_epi::row16_sum(s1);
      s2 = ds_epi::row16_sum(s2);
      s1 += __shfl_xor(s1, 16, 64);
      s2 += __shfl_xor(s2, 16, 64);
      if ((lane & 31) == 0) {
        f32x4 o = {K, s1, s2, ok ? 128.f : 0.f};                    // 2 rows x 64 columns per wave
        *reinterpret_cast<f32x4*>(&stat[(32 * m + co) * 4]) = o;
      }
    }
  }
  if (e.out_amax) ds_epi::commit_amax(e.out_amax, amax);
}

// IMGIN: the low-resolution input arrives as pre-split fp16 hi / lo images (ds_gnorm1_apply_images / ds_inorm_silu_images:
// [b][chunk][piece][h][Hl+2][Wl+2] vectors of 8 channels, zero border); an X buffer is filled by LDS-DMA, no registers, no split.
template <bool W16, bool PRE, bool CIRC, bool S16, bool IMGIN = false>
__global__ __launch_bounds__(NT, 2) void k_convup(const UpArgs a) {
  static_assert(!IMGIN || (S16 && !PRE && !CIRC), "image input: the plain zero-padded 16x16x32 kernel");
  constexpr int TH = Geo<W16>::TH, TW = Geo<W16>::TW, PW = Geo<W16>::PW, NPOS = Geo<W16>::NPOS;
  constexpr int XI = 3;
  static_assert(NPOS > NT && NPOS - NT <= NT / 2, "staging plan assumes 256 < NPOS <= 384");
  constexpr int HS = S16 ? HS16 : NPOS;                              // vectors between the two channel halves of a piece
  constexpr int XBV = S16 ? XBUF_VEC16 : XBUF_VEC;                   // ... between the two X buffers
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32x4* Xs = reinterpret_cast<u32x4*>(smem);                        // [buf][piece][h][pos]
  u32x4* Ws = Xs + 2 * XBV;                                          // [slot][piece][t][h][co]
  float* BS = reinterpret_cast<float*>(smem + (S16 ? STAGE_BYTES16 : MAIN_BYTES));   // [2][64] bias, shift
  float* ST = BS + 128;                                               // [4 waves][64 co][4]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int lane_pos = W16 ? (li >> 4) * PW + (li & 15) : li;
  const int wave_row = W16 ? 4 * wv : 2 * wv;
  constexpr int ROWS_PER_R = W16 ? 2 : 1;

  // grid = (2 row parities x channel tiles, pixel tiles, samples), renumbered so that each XCD owns a contiguous
  // run (see ds_conv3h.hip): the 2 * Cout/64 workgroups that read one input patch share an L2.
  unsigned cp_u, tile_u, b_u;
  {
    const unsigned nx = gridDim.x, ny = gridDim.y;
    const unsigned id = blockIdx.x + nx * (blockIdx.y + ny * blockIdx.z);
    const unsigned total = nx * ny * gridDim.z;
    const unsigned per = total >> 3, rem = total & 7u;
    const unsigned xcd = id & 7u, k = id >> 3;
    const unsigned logical = xcd * per + (xcd < rem ? xcd : rem) + k;
    cp_u = logical % nx;
    const unsigned rest = logical / nx;
    tile_u = rest % ny;
    b_u = rest / ny;
  }
  const int pa = (int)(cp_u & 1u), cot = (int)(cp_u >> 1);
  const int tile_id = (int)tile_u;
  const int b = (int)b_u;
  const int ty = a.tiles_x == 1 ? tile_id : (int)__umulhi((unsigned)tile_id, a.tiles_x_magic);
  const int tx = tile_id - ty * a.tiles_x;
  const int x0 = tx * TW, y0 = ty * TH;
  const int HW = a.Hl * a.Wl;
  const int n_steps = a.n_chunks * 4;

  // ---- staging plan (as ds_conv3h.hip): items 0 / 1 = position tid of channel half 0 / 1, item 2 = the
  //      patch's tail with h = wave / 2.  Patch origin: (y0 - 1 + a, x0 - 1). ----
  const int tail_h = wv >> 1;
  int xoff[XI], xlds[XI];
  unsigned xvalid = 0, xlive = 0;
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const int h = i == 0 ? 0 : (i == 1 ? 1 : tail_h);
    const int pos = i < 2 ? tid : NT + (tid & (NT / 2 - 1));
    const bool live = pos < NPOS;
    const int r = pos / PW;
    const int col = pos - r * PW;
    int gy = y0 + r - 1 + pa, gx = x0 + col - 1;
    if (CIRC) {                                       // tiles divide the image: at most one pixel outside
      gy = gy < 0 ? gy + a.Hl : (gy >= a.Hl ? gy - a.Hl : gy);
      gx = gx < 0 ? gx + a.Wl : (gx >= a.Wl ? gx - a.Wl : gx);
    }
    const bool ok = live && gy >= 0 && gy < a.Hl && gx >= 0 && gx < a.Wl;
    xoff[i] = ok ? gy * a.Wl + gx : 0;
    xlds[i] = h * HS + pos;
    if (ok) xvalid |= (1u << i);
    if (live) xlive |= (1u << i);
  }
  const float* in_b = a.in + (size_t)b * a.Cin * HW;
  const u32x4* wp = a.wp + (size_t)(cot * 2 + pa) * n_steps * WSLAB_VEC;
  ds_epi::ActScale ascale;

  float xr[XI][8];
  int xnch = KC;
  int xchunk = 0;
  auto x_fetch = [&](int chunk) __attribute__((always_inline)) {
    const int cbase = chunk * KC;
    const float* src = in_b + (size_t)cbase * HW;
    xchunk = chunk;
    xnch = a.Cin - cbase < KC ? a.Cin - cbase : KC;
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const int h = i == 0 ? 0 : (i == 1 ? 1 : tail_h);
      const float* p0 = src + xoff[i];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int c = 8 * h + k;
        xr[i][k] = p0[(c < xnch ? c : 0) * HW];
      }
    }
  };
  const float* pre_b = PRE ? a.prenorm + (size_t)b * a.n_chunks * KC * 4 : nullptr;
  auto x_activate = [&](int i) __attribute__((always_inline)) {
    typedef const __attribute__((address_space(4))) f32x4* cptr;
    cptr pp = (cptr)(reinterpret_cast<const f32x4*>(pre_b) + xchunk * KC);
    const int h = i == 0 ? 0 : (i == 1 ? 1 : tail_h);
    f32x4 p[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) p[k] = pp[8 * h + k];
    // the sample's activation exponent rides in the table's fourth column (2^-k, the same in every row; 0 = none): scalar select
    const float inv = p[0][3] == 0.f ? 1.0f : p[0][3];
    ascale.inv_scale = inv;                                          // the epilogue undoes it (unscale_from_inv)
#pragma unroll
    for (int k = 0; k < 8; ++k) xr[i][k] = ds_h3::fast_silu_scaled((xr[i][k] - p[k][0]) * p[k][1] + p[k][2], inv);
  };
  auto x_store = [&](int buf) __attribute__((always_inline)) {
    u32x4* xb = Xs + buf * XBV;
    if (PRE) { x_activate(0); x_activate(1); x_activate(2); }
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      if (i < 2 || ((xlive >> i) & 1u)) {
        const int h = i == 0 ? 0 : (i == 1 ? 1 : tail_h);
        const bool item_ok = (xvalid >> i) & 1u;
        u32x4 qh, ql;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float v0 = (item_ok && 8 * h + 2 * k < xnch) ? xr[i][2 * k] : 0.f;
          float v1 = (item_ok && 8 * h + 2 * k + 1 < xnch) ? xr[i][2 * k + 1] : 0.f;
          if constexpr (!PRE) { v0 *= ascale.in_scale; v1 *= ascale.in_scale; }   // exact (power of two)
          unsigned ph, pl;
          split2(v0, v1, ph, pl);
          qh[k] = ph; ql[k] = pl;
        }
        xb[xlds[i]] = qh;
        xb[2 * HS + xlds[i]] = ql;
      }
    }
  };
  auto w_fetch = [&](int step, int slot) __attribute__((always_inline)) {
    const u32x4* src = wp + (size_t)step * WSLAB_VEC;
    u32x4* dst = Ws + slot * WSLAB_VEC;
#pragma unroll
    for (int i = 0; i < WDMA; ++i) {
      const int k = wv + 4 * i;
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(src + 64 * k + lane),
          (__attribute__((address_space(3))) void*)(dst + 64 * k), 16, 0, 0);
    }
  };

  // ---- IMGIN: the X buffer [piece][h][HS] is one linear region of XBV = 20 x 64 vectors; DMA instruction k = wv + 4 i fills
  //      vectors 64 k ..; lane -> (image, position); the 14 pad vectors of an image re-read its last position (never used).
  //      Tiles divide the input, so every patch position lies inside the bordered image. ----
  constexpr int NDMA = IMGIN ? XBV / 64 / 4 : 1;
  static_assert(!IMGIN || XBV == 4 * 64 * NDMA, "whole DMA instructions per wave");
  int doff[NDMA];
  const int Wp = a.Wl + 2, Hp = a.Hl + 2;
  if constexpr (IMGIN) {
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
      const int L = 64 * (wv + 4 * i) + lane;
      const int q = L / HS;                                // image 2 * piece + h
      int pos = L - q * HS;
      pos = pos < NPOS ? pos : NPOS - 1;
      const int r = pos / PW, col = pos - r * PW;
      doff[i] = (q * Hp + (y0 + r + pa)) * Wp + x0 + col;  // bordered coordinates: patch origin (y0 - 1 + pa, x0 - 1) + 1
    }
  }
  const u32x4* img_b = IMGIN ? reinterpret_cast<const u32x4*>(a.in) + (size_t)b * a.n_chunks * 4 * Hp * Wp : nullptr;
  auto x_dma = [&](int chunk, int buf) __attribute__((always_inline)) {
    if constexpr (IMGIN) {
      const u32x4* src = img_b + (size_t)chunk * 4 * Hp * Wp;
      u32x4* dst = Xs + buf * XBV;
#pragma unroll
      for (int i = 0; i < NDMA; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + doff[i]),
                                         (__attribute__((address_space(3))) void*)(dst + 64 * (wv + 4 * i)), 16, 0, 0);
    }
  };

  // ---- prologue: patch 0, weight slabs 0 and 1 ----
  if constexpr (!IMGIN && !PRE) __builtin_amdgcn_sched_barrier(0);      // raw-input launches only: the others carry no exponent load
  const unsigned amax_bits = ds_epi::act_bits((IMGIN || PRE) ? nullptr : a.in_amax, b);   // see ds_conv3h.hip: issued here, consumed behind the first loads
  if constexpr (IMGIN) x_dma(0, 0); else
  x_fetch(0);
  const float bias_shift = ds_epi::fetch_bias_shift(a.bias, a.shift, a.shift_stride, b, cot * COT, a.Cout);
  w_fetch(0, 0);
  w_fetch(1, 1);                                                 // n_steps >= 4 always
  if constexpr (!IMGIN && !PRE) __builtin_amdgcn_sched_barrier(0);
  ascale = ds_epi::act_scale_of(PRE ? 0u : amax_bits, a.wshift);   // PRE: the table carries the exponent (x_activate)
  if constexpr (!IMGIN) x_store(0);
  ds_epi::commit_bias_shift(BS, bias_shift);
  __syncthreads();

  ds_epi::Args e;
  e.out = a.out; e.bias = a.bias; e.shift = a.shift; e.res1 = a.res1; e.res2 = a.res2; e.res1_up = a.res1_up;
  e.unscale = PRE ? ds_epi::unscale_from_inv(ascale.inv_scale, a.wshift) : ds_epi::unscale_from_in(ascale.in_scale, a.wshift);
    e.shift_stride = a.shift_stride;
  e.out_amax = nullptr;                                             // set behind the main loop (live range)
  e.b = b; e.co_base = cot * COT; e.y0 = y0 + wave_row; e.x0 = x0;
  e.Cout = a.Cout; e.H = 2 * a.Hl; e.W = 2 * a.Wl;
  e.pa = pa; e.pb = 0;
  // statistics: this workgroup covers the two column-parity "tiles" of its row parity; all 512 pixels are
  // accounted in the first entry, the second one is empty
  e.tile_stats = a.tile_stats; e.tile = (ty * a.tiles_x + tx) * 4 + pa * 2; e.ntiles = a.tiles_x * a.tiles_y * 4;
  float* tile = reinterpret_cast<float*>(smem) + wv * (64 * 2 * 32);
  float* stat = ST + wv * 256;

  if constexpr (S16) {
    f32x4 acc16[2][4][4];                                          // [column parity b][m16][n]
#pragma unroll
    for (int pb = 0; pb < 2; ++pb)
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc16[pb][m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    // lane group g = lane / 16: channel half g & 1 of column tap g >> 1
    const int l16 = lane & 15, gh = (lane >> 4) & 1, gt = lane >> 5;
    const int wlane = (gt * 2 + gh) * COT + l16;
    int xlane[4];
#pragma unroll
    for (int n = 0; n < 4; ++n)
      xlane[n] = gh * HS + (wave_row + (W16 ? 2 * (n >> 1) + (n & 1) : (n >> 1))) * PW + (W16 ? 0 : 16 * (n & 1)) + l16 + gt;
    auto step16 = [&](int chunk, int k, int slot) __attribute__((always_inline)) {
      const int g = chunk * 4 + k;
      const int s = k >> 1, pb = k & 1;
      const int xbuf = chunk & 1;
      if (g + 2 < n_steps) w_fetch(g + 2, slot >= 1 ? slot - 1 : 2);      // (slot + 2) % 3
      if (k == 0 && chunk + 1 < a.n_chunks) {
        // the other buffer's last readers finished before the previous step's barrier; this step's barrier publishes the DMA
        if constexpr (IMGIN) x_dma(chunk + 1, xbuf ^ 1); else x_fetch(chunk + 1);
      }
      __builtin_amdgcn_sched_barrier(0);
      const int wofs = slot * WSLAB_VEC + wlane;
      const int xofs = xbuf * XBV + s * PW + pb;
      f16x8 fa[2][4], fb[2][4];
#pragma unroll
      for (int p = 0; p < 2; ++p) {
#pragma unroll
        for (int m = 0; m < 4; ++m) fa[p][m] = *reinterpret_cast<const f16x8*>(&Ws[wofs + p * 4 * COT + 16 * m]);
#pragma unroll
        for (int n = 0; n < 4; ++n) fb[p][n] = *reinterpret_cast<const f16x8*>(&Xs[xofs + p * 2 * HS + xlane[n]]);
      }
      constexpr int PA[3] = {1, 0, 0};
      constexpr int PB[3] = {0, 1, 0};
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < 4; ++n)
            acc16[pb][m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[PA[t]][m], fb[PB[t]][n], acc16[pb][m][n], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!IMGIN)
        if (k == 2 && chunk + 1 < a.n_chunks) x_store(xbuf ^ 1);          // published by this step's barrier
      __syncthreads();
    };
    int chunk = 0;
    for (; chunk + 2 < a.n_chunks; chunk += 3) {
      step16(chunk, 0, 0); step16(chunk, 1, 1); step16(chunk, 2, 2); step16(chunk, 3, 0);
      step16(chunk + 1, 0, 1); step16(chunk + 1, 1, 2); step16(chunk + 1, 2, 0); step16(chunk + 1, 3, 1);
      step16(chunk + 2, 0, 2); step16(chunk + 2, 1, 0); step16(chunk + 2, 2, 1); step16(chunk + 2, 3, 2);
    }
    if (chunk < a.n_chunks) { step16(chunk, 0, 0); step16(chunk, 1, 1); step16(chunk, 2, 2); step16(chunk, 3, 0); ++chunk; }
    if (chunk < a.n_chunks) { step16(chunk, 0, 1); step16(chunk, 1, 2); step16(chunk, 2, 0); step16(chunk, 3, 1); }
    e.out_amax = a.out_amax ? a.out_amax + b : nullptr;
    e.unscale = PRE ? ds_epi::unscale_from_inv(ascale.inv_scale, a.wshift) : ds_epi::unscale_from_in(ascale.in_scale, a.wshift);
    store_half16<W16>(acc16[0], acc16[1], 0, tile, BS, stat, e);
    store_half16<W16>(acc16[0], acc16[1], 1, tile, BS, stat, e);
  } else {
  f32x16 acc[2][2][2];                                             // [column parity b][m][r]
#pragma unroll
  for (int pb = 0; pb < 2; ++pb)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[pb][m][r][q] = 0.f;

  // operands of (ring slot, X buffer, row tap s, column parity pb, column tap t): the patch column is x + pb + t
  auto frag_load = [&](Frags& f, int slot, int xbuf, int s, int pb, int t) __attribute__((always_inline)) {
    const u32x4* wb = Ws + slot * WSLAB_VEC;
    const u32x4* xb = Xs + xbuf * XBUF_VEC;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
      for (int m = 0; m < 2; ++m)
        f.a[p][m] = *reinterpret_cast<const f16x8*>(&wb[((p * 2 + t) * 2 + lh) * COT + 32 * m + li]);
#pragma unroll
      for (int r = 0; r < 2; ++r)
        f.b[p][r] = *reinterpret_cast<const f16x8*>(&xb[(p * 2 + lh) * NPOS + (wave_row + ROWS_PER_R * r + s) * PW + lane_pos + pb + t]);
    }
  };
  auto frag_mma = [&](const Frags& f, int pb) __attribute__((always_inline)) {       // lo*hi, hi*lo, hi*hi
    constexpr int PA[3] = {1, 0, 0};
    constexpr int PB[3] = {0, 1, 0};
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 2; ++r)
          acc[pb][m][r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[PA[t]][m], f.b[PB[t]][r], acc[pb][m][r], 0, 0, 0);
  };
  auto reads_between_mfmas = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
  };

  Frags fA, fB;
  frag_load(fA, 0, 0, 0, 0, 0);

  // One step = (chunk, k) with k = 2*s + pb.  fA holds the operands of t = 0 on entry and, on exit, those of the
  // next step's t = 0.  slot = g % 3 for step g = 4*chunk + k: compile-time at every call site (3-chunk unroll).
  auto step = [&](int chunk, int k, int slot) __attribute__((always_inline)) {
    const int g = chunk * 4 + k;
    const int s = k >> 1, pb = k & 1;
    const int xbuf = chunk & 1;
    if (g + 2 < n_steps) w_fetch(g + 2, slot >= 1 ? slot - 1 : 2);      // (slot + 2) % 3
    if (k == 0 && chunk + 1 < a.n_chunks) x_fetch(chunk + 1);
    __builtin_amdgcn_sched_barrier(0);
    frag_load(fB, slot, xbuf, s, pb, 1);
    frag_mma(fA, pb);
    reads_between_mfmas();
    __builtin_amdgcn_sched_barrier(0);
    if (g + 1 < n_steps) {
      const int nslot = slot == 2 ? 0 : slot + 1;
      const int nk = k == 3 ? 0 : k + 1;
      frag_load(fA, nslot, k == 3 ? xbuf ^ 1 : xbuf, nk >> 1, nk & 1, 0);
    }
    frag_mma(fB, pb);
    if (g + 1 < n_steps) reads_between_mfmas();
    __builtin_amdgcn_sched_barrier(0);
    if (k == 2 && chunk + 1 < a.n_chunks) x_store(xbuf ^ 1);            // published by this step's barrier
    __syncthreads();
  };

  // slot pattern of a chunk's four steps repeats every three chunks: (0,1,2,0) (1,2,0,1) (2,0,1,2)
  int chunk = 0;
  for (; chunk + 2 < a.n_chunks; chunk += 3) {
    step(chunk, 0, 0); step(chunk, 1, 1); step(chunk, 2, 2); step(chunk, 3, 0);
    step(chunk + 1, 0, 1); step(chunk + 1, 1, 2); step(chunk + 1, 2, 0); step(chunk + 1, 3, 1);
    step(chunk + 2, 0, 2); step(chunk + 2, 1, 0); step(chunk + 2, 2, 1); step(chunk + 2, 3, 2);
  }
  if (chunk < a.n_chunks) { step(chunk, 0, 0); step(chunk, 1, 1); step(chunk, 2, 2); step(chunk, 3, 0); ++chunk; }
  if (chunk < a.n_chunks) { step(chunk, 0, 1); step(chunk, 1, 2); step(chunk, 2, 0); step(chunk, 3, 1); }

    e.out_amax = a.out_amax ? a.out_amax + b : nullptr;
    e.unscale = PRE ? ds_epi::unscale_from_inv(ascale.inv_scale, a.wshift) : ds_epi::unscale_from_in(ascale.in_scale, a.wshift);
    store_half<W16>(acc[0][0], acc[1][0], 0, tile, BS, stat, e);
    store_half<W16>(acc[0][1], acc[1][1], 1, tile, BS, stat, e);
  }
  {
    if (a.tile_stats) {
      __syncthreads();
      ds_epi::store_tile_stats(ST, 256, e);
      if (tid < 64 && e.co_base + tid < e.Cout)
        *reinterpret_cast<f32x4*>(a.tile_stats + (((size_t)b * a.Cout + e.co_base + tid) * e.ntiles + e.tile + 1) * 4) =
            f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
}

// torch [Cout][Cin][3][3] fp32 -> [cot][row parity a][chunk][s 2][column parity b][piece 2][t 2][h 2][co 64][ci 8]
// fp16 of the collapsed 2x2 taps (times 2^wshift)
__global__ void k_pack_up(_Float16* packed, const float* __restrict__ w, int Cout, int Cin, int n_chunks, float scale,
                          size_t total) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  size_t t = i;
  const int c8 = t % 8; t /= 8;
  const int co64 = t % COT; t /= COT;
  const int h = t % 2; t /= 2;
  const int tt = t % 2; t /= 2;
  const int piece = t % 2; t /= 2;
  const int pb = t % 2; t /= 2;
  const int s = t % 2; t /= 2;
  const int chunk = t % n_chunks; t /= n_chunks;
  const int pa = t % 2; t /= 2;
  const int cot = (int)t;
  const int co = cot * COT + co64, ci = chunk * KC + 8 * h + c8;
  float v = 0.f;
  if (co < Cout && ci < Cin) {
    const float* wk = w + ((size_t)co * Cin + ci) * 9;
    // rows of parity pa: s = 0 -> ky in [0, pa], s = 1 -> ky in [pa + 1, 2]; columns alike
    const int ky0 = s == 0 ? 0 : pa + 1, ky1 = s == 0 ? pa : 2;
    const int kx0 = tt == 0 ? 0 : pb + 1, kx1 = tt == 0 ? pb : 2;
    double acc = 0.0;
    for (int ky = ky0; ky <= ky1; ++ky)
      for (int kx = kx0; kx <= kx1; ++kx) acc += (double)wk[ky * 3 + kx];
    v = (float)acc * scale;
  }
  const _Float16 hi = (_Float16)v;
  const _Float16 lo = (_Float16)(v - (float)hi);
  packed[i] = piece == 0 ? hi : lo;
}

// DS_CONV_SHAPE=32 keeps the 32x32x16 form (A/B measurements); default: 16x16x32
bool up_shape16() {
  static const bool v = [] { const char* e = getenv("DS_CONV_SHAPE"); return !(e && atoi(e) == 32); }();
  return v;
}

template <bool W16, bool PRE, bool CIRC, bool S16, bool IMGIN = false>
int launch_up_shape(const UpArgs& a, hipStream_t s) {
  constexpr int LDS = S16 ? LDS_BYTES16 : LDS_BYTES;
  {
    const int rc = ds::ensure_dynamic_lds<&k_convup<W16, PRE, CIRC, S16, IMGIN>>(LDS, "hipFuncSetAttribute(convup)");
    if (rc != DS_OK) return rc;
  }
  const long long tiles = (long long)a.tiles_y * a.tiles_x;
  DS_REQUIRE(tiles > 0 && tiles < 65536 && a.B < 65536, DS_ERR_SHAPE,
             "ds_conv2d_h3_up: %lld pixel tiles x %d samples exceed the grid limits (65535 each)", tiles, a.B);
  hipLaunchKernelGGL((k_convup<W16, PRE, CIRC, S16, IMGIN>), dim3((unsigned)a.n_cot * 2u, (unsigned)tiles, (unsigned)a.B), dim3(NT),
                     LDS, s, a);
  DS_CHECK_LAUNCH("ds_conv2d_h3_up");
  return DS_OK;
}

template <bool W16, bool PRE, bool CIRC>
int launch_up(const UpArgs& a, hipStream_t s) {
  return up_shape16() ? launch_up_shape<W16, PRE, CIRC, true>(a, s) : launch_up_shape<W16, PRE, CIRC, false>(a, s);
}

// 0: unsupported, 1: 8 x 32 tiles, 2: 16 x 16 tiles
int up_geometry(int Hl, int Wl) {
  if (Hl > 0 && Wl > 0 && Hl % 8 == 0 && Wl % 32 == 0) return 1;
  if (Hl > 0 && Wl > 0 && Hl % 16 == 0 && Wl % 16 == 0) return 2;
  return 0;
}

}  // namespace

extern "C" {

int ds_conv2d_h3_up_supported(int Hl, int Wl) { return up_geometry(Hl, Wl) != 0; }

size_t ds_conv2d_h3_up_packed_bytes(int Cout, int Cin) {
  if (Cout <= 0 || Cin <= 0) return 0;
  const size_t n_cot = (Cout + COT - 1) / COT, n_chunks = (Cin + KC - 1) / KC;
  return n_cot * 4 * n_chunks * 2 * (size_t)WSLAB_VEC * 16;
}

int ds_conv2d_h3_up_pack_weights(void* packed, const float* w, int Cout, int Cin, int wshift, void* stream) {
  DS_REQUIRE(packed && w, DS_ERR_NULL, "ds_conv2d_h3_up_pack_weights: NULL pointer");
  DS_REQUIRE(Cout > 0 && Cin > 0, DS_ERR_SHAPE, "ds_conv2d_h3_up_pack_weights: Cout=%d Cin=%d", Cout, Cin);
  DS_REQUIRE(wshift >= -40 && wshift <= 40, DS_ERR_SHAPE, "ds_conv2d_h3_up_pack_weights: wshift %d out of range", wshift);
  const int n_chunks = (Cin + KC - 1) / KC;
  const size_t total = ds_conv2d_h3_up_packed_bytes(Cout, Cin) / 2;
  hipLaunchKernelGGL(k_pack_up, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ds::as_stream(stream),
                     reinterpret_cast<_Float16*>(packed), w, Cout, Cin, n_chunks, ldexpf(1.0f, wshift), total);
  DS_CHECK_LAUNCH("ds_conv2d_h3_up_pack_weights");
  return DS_OK;
}

int ds_conv2d_h3_up(float* out, const float* in, const void* w_packed, int wshift, const float* bias, const float* shift,
                    int shift_stride, const float* res1, const float* res2, int B, int Cin, int Cout, int Hl, int Wl,
                    int flags, const float* prenorm, float* tile_stats, const unsigned* in_amax, unsigned* out_amax, void* stream) {
  DS_REQUIRE(out && in && w_packed, DS_ERR_NULL, "ds_conv2d_h3_up: NULL pointer");
  DS_REQUIRE(!(prenorm && in_amax), DS_ERR_UNSUPPORTED, "ds_conv2d_h3_up: in_amax is for raw inputs (with prenorm the table's fourth column carries the exponent)");
  DS_REQUIRE(B >= 0 && Cin > 0 && Cout > 0 && Hl > 0 && Wl > 0, DS_ERR_SHAPE,
             "ds_conv2d_h3_up: bad shape B=%d Cin=%d Cout=%d Hl=%d Wl=%d", B, Cin, Cout, Hl, Wl);
  DS_REQUIRE((flags & ~(DS_PAD_CIRCULAR | DS_RES1_UPSAMPLED)) == 0, DS_ERR_UNSUPPORTED, "ds_conv2d_h3_up: flags %d", flags);
  const int geo = up_geometry(Hl, Wl);
  DS_REQUIRE(geo != 0, DS_ERR_UNSUPPORTED,
             "ds_conv2d_h3_up: %d x %d input is not a whole number of 8x32 or 16x16 tiles (use ds_conv2d_h3 with DS_LOAD_UPSAMPLE2)",
             Hl, Wl);
  const int res1_up = (flags & DS_RES1_UPSAMPLED) ? 1 : 0;
  DS_REQUIRE(!res1_up || res1, DS_ERR_NULL, "ds_conv2d_h3_up: RES1_UPSAMPLED needs res1");
  DS_REQUIRE(shift == nullptr || shift_stride == 0 || shift_stride >= Cout, DS_ERR_SHAPE,
             "ds_conv2d_h3_up: shift_stride %d < Cout %d", shift_stride, Cout);
  DS_REQUIRE((reinterpret_cast<uintptr_t>(w_packed) & 15u) == 0, DS_ERR_SHAPE, "ds_conv2d_h3_up: w_packed must be 16-byte aligned");
  DS_REQUIRE((reinterpret_cast<uintptr_t>(prenorm) & 15u) == 0, DS_ERR_SHAPE, "ds_conv2d_h3_up: prenorm must be 16-byte aligned");
  DS_REQUIRE(!res1_up || (reinterpret_cast<uintptr_t>(res1) & 15u) == 0, DS_ERR_SHAPE, "ds_conv2d_h3_up: low-resolution res1 must be 16-byte aligned");
  DS_REQUIRE(wshift >= -40 && wshift <= 40, DS_ERR_SHAPE, "ds_conv2d_h3_up: wshift %d out of range", wshift);
  DS_REQUIRE((long long)Cin * Hl * Wl < (1ll << 31), DS_ERR_SHAPE, "ds_conv2d_h3_up: per-sample input exceeds 2^31 floats");
  if (B == 0) return DS_OK;
  UpArgs a;
  a.out = out; a.in = in; a.wp = reinterpret_cast<const u32x4*>(w_packed); a.bias = bias; a.shift = shift;
  a.res1 = res1; a.res2 = res2; a.prenorm = prenorm; a.tile_stats = tile_stats;
  a.wshift = wshift; a.in_amax = in_amax; a.out_amax = out_amax; a.shift_stride = shift_stride; a.res1_up = res1_up;
  a.B = B; a.Cin = Cin; a.Cout = Cout; a.Hl = Hl; a.Wl = Wl;
  const bool w16 = geo == 2;
  const int TW = w16 ? 16 : 32, TH = w16 ? 16 : 8;
  a.tiles_x = Wl / TW; a.tiles_y = Hl / TH;
  a.n_cot = (Cout + COT - 1) / COT;
  a.n_chunks = (Cin + KC - 1) / KC;
  a.tiles_x_magic = a.tiles_x == 1 ? 0u : (unsigned)((1ull << 32) / (unsigned)a.tiles_x) + 1u;
  hipStream_t s = ds::as_stream(stream);
  const bool circ = flags & DS_PAD_CIRCULAR;
#define DS_LU(W) (prenorm ? (circ ? launch_up<W, true, true>(a, s) : launch_up<W, true, false>(a, s)) \
                          : (circ ? launch_up<W, false, true>(a, s) : launch_up<W, false, false>(a, s)))
  return w16 ? DS_LU(true) : DS_LU(false);
#undef DS_LU
}

int ds_conv2d_h3_up_img(float* out, const void* images, const void* w_packed, int wshift, const float* bias, const float* shift,
                        int shift_stride, const float* res1, const float* res2, int B, int Cin, int Cout, int Hl, int Wl,
                        float* tile_stats, unsigned* out_amax, void* stream) {
  DS_REQUIRE(out && images && w_packed, DS_ERR_NULL, "ds_conv2d_h3_up_img: NULL pointer");
  DS_REQUIRE(B >= 0 && Cin > 0 && Cout > 0 && Hl > 0 && Wl > 0, DS_ERR_SHAPE,
             "ds_conv2d_h3_up_img: bad shape B=%d Cin=%d Cout=%d Hl=%d Wl=%d", B, Cin, Cout, Hl, Wl);
  const int geo = up_geometry(Hl, Wl);
  DS_REQUIRE(geo != 0, DS_ERR_UNSUPPORTED, "ds_conv2d_h3_up_img: %d x %d input is not a whole number of 8x32 or 16x16 tiles", Hl, Wl);
  DS_REQUIRE(shift == nullptr || shift_stride == 0 || shift_stride >= Cout, DS_ERR_SHAPE,
             "ds_conv2d_h3_up_img: shift_stride %d < Cout %d", shift_stride, Cout);
  DS_REQUIRE(((reinterpret_cast<uintptr_t>(w_packed) | reinterpret_cast<uintptr_t>(images)) & 15u) == 0, DS_ERR_SHAPE,
             "ds_conv2d_h3_up_img: images and w_packed must be 16-byte aligned");
  DS_REQUIRE(wshift >= -40 && wshift <= 40, DS_ERR_SHAPE, "ds_conv2d_h3_up_img: wshift %d out of range", wshift);
  DS_REQUIRE((long long)4 * (Hl + 2) * (Wl + 2) < (1ll << 27), DS_ERR_SHAPE, "ds_conv2d_h3_up_img: image planes too large");
  if (B == 0) return DS_OK;
  UpArgs a;
  a.out = out; a.in = reinterpret_cast<const float*>(images); a.wp = reinterpret_cast<const u32x4*>(w_packed); a.bias = bias;
  a.shift = shift; a.res1 = res1; a.res2 = res2; a.prenorm = nullptr; a.tile_stats = tile_stats;
  a.wshift = wshift; a.in_amax = nullptr; a.out_amax = out_amax; a.shift_stride = shift_stride; a.res1_up = 0;
  a.B = B; a.Cin = Cin; a.Cout = Cout; a.Hl = Hl; a.Wl = Wl;
  const bool w16 = geo == 2;
  const int TW = w16 ? 16 : 32, TH = w16 ? 16 : 8;
  a.tiles_x = Wl / TW; a.tiles_y = Hl / TH;
  a.n_cot = (Cout + COT - 1) / COT;
  a.n_chunks = (Cin + KC - 1) / KC;
  a.tiles_x_magic = a.tiles_x == 1 ? 0u : (unsigned)((1ull << 32) / (unsigned)a.tiles_x) + 1u;
  hipStream_t s = ds::as_stream(stream);
  return w16 ? launch_up_shape<true, false, false, true, true>(a, s) : launch_up_shape<false, false, false, true, true>(a, s);
}

}  // extern "C"
