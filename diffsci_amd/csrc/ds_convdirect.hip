// 3x3 convolution with very few output channels (Cout <= 4): the networks' output layers
// (PUNetG convout, punetg.py:415; ADM output_layer, adm.py:193-215).  On the MFMA kernels these pad
// Cout to a 64-channel tile -- 16-64x the necessary matrix work for a layer that only has to stream
// its input once.  Here: exact fp32 FMA chains (channels outer, taps inner), HBM-bound:
//   workgroup = 256 threads, tile = 16 rows x 64 columns, 4 consecutive pixels per thread;
//   8 input channels at a time are staged as zero- (or periodically) padded 18 x 66 patches in LDS;
//   weights are wave-uniform scalar loads (constant address space).
#include "ds_common.h"
#include <cstdlib>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int NT = 256, TH = 16, TW = 64, KCH = 8;
constexpr int PH = TH + 2, PSTR = 68;                   // row stride padded to a multiple of 4 floats
constexpr int PATCH = PH * PSTR;

template <int COUT>
__global__ __launch_bounds__(NT) void k_conv_direct(float* out, const float* __restrict__ in, const float* w,
                                                    const float* __restrict__ bias, int Cin, int H, int W,
                                                    int tiles_x, int tiles_y, int circular) {
  __shared__ __attribute__((aligned(16))) float patch[KCH * PATCH];
  const int tid = threadIdx.x;
  int bid = blockIdx.x;
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y; bid /= tiles_y;
  const int b = bid;
  const int x0 = tx * TW, y0 = ty * TH;
  const size_t HW = (size_t)H * W;
  const float* in_b = in + (size_t)b * Cin * HW;
  const int row = tid >> 4, c4 = (tid & 15) * 4;
  typedef const __attribute__((address_space(4))) float* cptr;
  cptr wc = (cptr)w;

  float acc[COUT][4];
#pragma unroll
  for (int co = 0; co < COUT; ++co)
#pragma unroll
    for (int p = 0; p < 4; ++p) acc[co][p] = 0.f;

  for (int c0 = 0; c0 < Cin; c0 += KCH) {
    const int nch = Cin - c0 < KCH ? Cin - c0 : KCH;
    __syncthreads();                                   // previous chunk fully consumed
    // staging: thread -> (column tid % 64 [and the two halo columns 64, 65 for tid % 64 < 2], rows tid/64 + 4k).
    // All loads of the chunk are issued before the first LDS write (40 + 10 in flight per thread): issued one
    // at a time the kernel was latency-bound at 4x the time of the MFMA path it replaces.
    {
      const int col = tid & 63, r0 = tid >> 6;
      int gx0 = x0 + col - 1, gx1 = x0 + 64 + col - 1;
      if (circular) {
        gx0 = (gx0 < 0 ? gx0 + W : gx0) % W;
        gx1 = gx1 % W;
      }
      const bool okx0 = gx0 >= 0 && gx0 < W, okx1 = col < 2 && gx1 >= 0 && gx1 < W;
      constexpr int NR = (PH + 3) / 4;                                  // 5 row slots per thread
      float v0[NR][KCH], v1[NR][KCH];
#pragma unroll
      for (int k = 0; k < NR; ++k) {
        const int r = r0 + 4 * k;
        int gy = y0 + r - 1;
        if (circular) gy = (gy < 0 ? gy + H : gy) % H;
        const bool oky = r < PH && gy >= 0 && gy < H;
        const float* src = in_b + (size_t)c0 * HW + (size_t)(oky ? gy : 0) * W;
#pragma unroll
        for (int ch = 0; ch < KCH; ++ch) {
          const bool okc = ch < nch;
          v0[k][ch] = (oky && okx0 && okc) ? src[(size_t)ch * HW + gx0] : 0.f;
          v1[k][ch] = (oky && okx1 && okc) ? src[(size_t)ch * HW + gx1] : 0.f;
        }
      }
#pragma unroll
      for (int k = 0; k < NR; ++k) {
        const int r = r0 + 4 * k;
        if (r < PH) {
#pragma unroll
          for (int ch = 0; ch < KCH; ++ch) {
            float* dst = &patch[ch * PATCH + r * PSTR];
            dst[col] = v0[k][ch];
            if (col < 2) dst[64 + col] = v1[k][ch];
          }
        }
      }
    }
    __syncthreads();
    for (int ch = 0; ch < nch; ++ch) {
      float v[3][6];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const float* p = &patch[ch * PATCH + (row + ky) * PSTR + c4];
        const f32x4 a = *reinterpret_cast<const f32x4*>(p);
        const f32x2 c = *reinterpret_cast<const f32x2*>(p + 4);
        v[ky][0] = a[0]; v[ky][1] = a[1]; v[ky][2] = a[2]; v[ky][3] = a[3]; v[ky][4] = c[0]; v[ky][5] = c[1];
      }
#pragma unroll
      for (int co = 0; co < COUT; ++co) {
        cptr wk = wc + ((size_t)co * Cin + (c0 + ch)) * 9;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const float wv = wk[ky * 3 + kx];
#pragma unroll
            for (int p = 0; p < 4; ++p) acc[co][p] = __builtin_fmaf(wv, v[ky][p + kx], acc[co][p]);
          }
      }
    }
  }
  const int gy = y0 + row, gx = x0 + c4;
  if (gy < H) {
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
      const float bv = bias ? bias[co] : 0.f;
      float* o = out + ((size_t)b * COUT + co) * HW + (size_t)gy * W + gx;
      if ((W & 3) == 0 && gx + 3 < W) {
        f32x4 r = {acc[co][0] + bv, acc[co][1] + bv, acc[co][2] + bv, acc[co][3] + bv};
        *reinterpret_cast<f32x4*>(o) = r;
      } else {
#pragma unroll
        for (int p = 0; p < 4; ++p)
          if (gx + p < W) o[p] = acc[co][p] + bv;
      }
    }
  }
}

// The same kernel for whole 64-column tiles (W a multiple of 64, Cin a multiple of 8, a 16-byte-aligned input): the patch's 64 interior
// columns are fetched by 16-BYTE loads (9 per thread and chunk instead of 40) and the NEXT chunk's loads are issued in front of this
// chunk's arithmetic (registers as the second buffer), so the stream from memory overlaps the FMA chains inside a workgroup as well.
// Same tile, same LDS patch, same chains in the same order: bit-identical to k_conv_direct (round 4, profiles/r04_direct_conv.log: 86 -> 80 us at config 2's
// output layer, 142 -> 105 us at config 5's (4 channels, 256 x 256), 412 -> 349 us at config 3's).
template <int COUT>
__global__ __launch_bounds__(NT, 4) void k_conv_direct_v(float* out, const float* __restrict__ in, const float* w,
                                                      const float* __restrict__ bias, int Cin, int H, int W,
                                                      int tiles_x, int tiles_y, int circular) {
  __shared__ __attribute__((aligned(16))) float patch[KCH * PATCH];
  const int tid = threadIdx.x;
  int bid = blockIdx.x;
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y; bid /= tiles_y;
  const int b = bid;
  const int x0 = tx * TW, y0 = ty * TH;
  const size_t HW = (size_t)H * W;
  const float* in_b = in + (size_t)b * Cin * HW;
  const int row = tid >> 4, c4 = (tid & 15) * 4;
  typedef const __attribute__((address_space(4))) float* cptr;
  cptr wc = (cptr)w;

  // staging plan: interior slot k = vector (t + 256 k) of [8 channels][18 rows][16 column groups]; halo slot 0 = element t, slot 1 =
  // element 256 + t (t < 32) of [8 channels][18 rows][left, right]
  constexpr int NV = KCH * PH * 16 / NT;                 // 9
  static_assert(KCH * PH * 16 % NT == 0 && KCH * PH * 2 <= 2 * NT, "staging plan");
  int goff[NV], loff[NV];
  unsigned vok = 0;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int idx = tid + NT * k;
    const int ch = idx / (PH * 16), rem = idx - ch * (PH * 16);
    const int r = rem >> 4, g = rem & 15;
    int gy = y0 + r - 1;
    if (circular) gy = (gy < 0 ? gy + H : gy) % H;
    const bool ok = gy >= 0 && gy < H;
    goff[k] = ch * (int)HW + (ok ? gy : 0) * W + x0 + 4 * g;
    loff[k] = ch * PATCH + r * PSTR + 1 + 4 * g;
    if (ok) vok |= 1u << k;
  }
  int hgoff[2], hloff[2];
  unsigned hok = 0;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int idx = tid + NT * k;
    const bool live = idx < KCH * PH * 2;
    const int ii = live ? idx : 0;
    const int ch = ii / (PH * 2), rem = ii - ch * (PH * 2);
    const int r = rem >> 1, side = rem & 1;
    int gy = y0 + r - 1, gx = side ? x0 + TW : x0 - 1;
    if (circular) { gy = (gy < 0 ? gy + H : gy) % H; gx = (gx < 0 ? gx + W : gx) % W; }
    const bool ok = live && gy >= 0 && gy < H && gx >= 0 && gx < W;
    hgoff[k] = ch * (int)HW + (ok ? gy * W + gx : 0);
    hloff[k] = live ? ch * PATCH + r * PSTR + (side ? TW + 1 : 0) : -1;
    if (ok) hok |= 1u << k;
  }

  float acc[COUT][4];
#pragma unroll
  for (int co = 0; co < COUT; ++co)
#pragma unroll
    for (int p = 0; p < 4; ++p) acc[co][p] = 0.f;

  f32x4 v[NV];
  float hv[2];
  auto fetch = [&](int c0) __attribute__((always_inline)) {
    const float* src = in_b + (size_t)c0 * HW;
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = *reinterpret_cast<const f32x4*>(src + goff[k]);
#pragma unroll
    for (int k = 0; k < 2; ++k) hv[k] = src[hgoff[k]];
  };
  fetch(0);
  for (int c0 = 0; c0 < Cin; c0 += KCH) {
    __syncthreads();                                   // previous chunk fully consumed
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const bool ok = (vok >> k) & 1u;
#pragma unroll
      for (int j = 0; j < 4; ++j) patch[loff[k] + j] = ok ? v[k][j] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 2; ++k)
      if (hloff[k] >= 0) patch[hloff[k]] = ((hok >> k) & 1u) ? hv[k] : 0.f;
    __syncthreads();
    if (c0 + KCH < Cin) fetch(c0 + KCH);               // in flight while this chunk multiplies
#pragma unroll 2
    for (int ch = 0; ch < KCH; ++ch) {
      float x[3][6];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const float* p = &patch[ch * PATCH + (row + ky) * PSTR + c4];
        const f32x4 a = *reinterpret_cast<const f32x4*>(p);
        const f32x2 c = *reinterpret_cast<const f32x2*>(p + 4);
        x[ky][0] = a[0]; x[ky][1] = a[1]; x[ky][2] = a[2]; x[ky][3] = a[3]; x[ky][4] = c[0]; x[ky][5] = c[1];
      }
#pragma unroll
      for (int co = 0; co < COUT; ++co) {
        cptr wk = wc + ((size_t)co * Cin + (c0 + ch)) * 9;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const float wv = wk[ky * 3 + kx];
#pragma unroll
            for (int p = 0; p < 4; ++p) acc[co][p] = __builtin_fmaf(wv, x[ky][p + kx], acc[co][p]);
          }
      }
    }
  }
  const int gy = y0 + row, gx = x0 + c4;
  if (gy < H) {
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
      const float bv = bias ? bias[co] : 0.f;
      f32x4 r = {acc[co][0] + bv, acc[co][1] + bv, acc[co][2] + bv, acc[co][3] + bv};
      *reinterpret_cast<f32x4*>(out + ((size_t)b * COUT + co) * HW + (size_t)gy * W + gx) = r;
    }
  }
}

}  // namespace

extern "C" int ds_conv2d_direct(float* out, const float* in, const float* w, const float* bias, int B, int Cin, int Cout,
                                int H, int W, int circular, void* stream) {
  DS_REQUIRE(out && in && w, DS_ERR_NULL, "ds_conv2d_direct: NULL pointer");
  DS_REQUIRE(B >= 0 && Cin > 0 && H > 0 && W > 0, DS_ERR_SHAPE, "ds_conv2d_direct: bad shape");
  DS_REQUIRE(Cout >= 1 && Cout <= 4, DS_ERR_UNSUPPORTED, "ds_conv2d_direct: Cout=%d (1..4 supported)", Cout);
  DS_REQUIRE((reinterpret_cast<uintptr_t>(out) & 15u) == 0, DS_ERR_SHAPE, "ds_conv2d_direct: out must be 16-byte aligned");
  if (B == 0) return DS_OK;
  const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
  const long long blocks = (long long)B * tiles_x * tiles_y;
  DS_REQUIRE(blocks < (1ll << 31), DS_ERR_SHAPE, "ds_conv2d_direct: grid too large");
  hipStream_t s = ds::as_stream(stream);
  dim3 g((unsigned)blocks), t(NT);
  // whole 64-column tiles: the 16-byte-load form (DS_DIRECT_VEC=0 keeps the general kernel: A/B runs; results are bit-identical)
  static const bool vec_on = [] { const char* e = getenv("DS_DIRECT_VEC"); return !(e && atoi(e) == 0); }();
  const bool vec = vec_on && W % TW == 0 && Cin % KCH == 0 && (reinterpret_cast<uintptr_t>(in) & 15u) == 0 && (long long)Cin * H * W < (1ll << 31);
#define L(C) do { if (vec) hipLaunchKernelGGL((k_conv_direct_v<C>), g, t, 0, s, out, in, w, bias, Cin, H, W, tiles_x, tiles_y, circular); \
                  else hipLaunchKernelGGL((k_conv_direct<C>), g, t, 0, s, out, in, w, bias, Cin, H, W, tiles_x, tiles_y, circular); } while (0)
  switch (Cout) { case 1: L(1); break; case 2: L(2); break; case 3: L(3); break; default: L(4); }
#undef L
  DS_CHECK_LAUNCH("ds_conv2d_direct");
  return DS_OK;
}
