// 3x3 convolution with fp32 accuracy on the bf16 matrix cores ("bf16x6" split MFMA).
//
// Every fp32 operand is split EXACTLY into three bf16 pieces, x = x1 + x2 + x3 (8 + 8 + 8
// significand bits, by truncation, each remainder computed exactly in fp32), and the product
// a*b is accumulated as the six piece products with i + j <= 4:
//     a1b1 + a1b2 + a2b1 + a1b3 + a2b2 + a3b1        (dropped terms are <= 2^-24 |ab|)
// Each bf16 x bf16 product is exact in fp32 and v_mfma_f32_32x32x16_bf16 accumulates in fp32,
// so the result carries fp32-level rounding error (measured: same error against an fp64
// reference as torch's fp32 convolution), at 6/16 of the instruction time of the exact-fp32
// MFMA (v_mfma_f32_32x32x2_f32): 2.67x its peak.  This is fp32 arithmetic emulated on the
// matrix cores, not a reduced-precision mode; the exact-fp32 kernel (ds_conv.hip) stays
// available for A/B comparison.
//
// Geometry (per workgroup = 4 waves): 64 output channels x (8 rows x 32 columns) of one sample;
// wave w owns rows 2w, 2w+1 -> 2 (channel tiles) x 2 (rows) accumulators of 32x32.
//   MFMA A (weights): lane (i = l&31, h = l>>5) holds piece[co = 32m+i][ci = 8h .. 8h+7]  (16 B)
//   MFMA B (input):   lane (j = l&31, h)        holds piece[ci = 8h .. 8h+7][row+ky][col j+kx]
// LDS images keep the 8 channels of one (position, h) contiguous (16 B) and positions
// contiguous per h, so every operand fetch is a conflict-free ds_read_b128 of a contiguous 1 KiB:
//   X  [piece 3][h 2][10 rows * 34 cols][8 ci] bf16    32,640 B (one buffer per 16-channel chunk)
//   W  [2 buffers][piece 3][kx 3][h 2][64 co][8 ci]    2 x 18,432 B (one (chunk, ky) slab each)
// K is walked in steps (chunk of 16 input channels, ky): the next step's weight slab and the
// next chunk's input patch are staged under the current step's MFMAs (weights by LDS-DMA, the
// input global -> registers -> LDS because it is split on the way);
// the fp32 -> 3 x bf16 split of the input happens once per staged element, in registers.
#include "ds_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int TH = 8, TW = 32, COT = 64, NT = 256;
constexpr int KC = 16;                      // input channels per chunk (= MFMA K)
constexpr int PH = TH + 2, PW = TW + 2, NPOS = PH * PW;     // 10 x 34 = 340 patch positions
constexpr int XITEMS = 2 * NPOS;            // (h, position) staging items per chunk: 680
constexpr int XI = (XITEMS + NT - 1) / NT;  // 3 per thread
constexpr int X_BYTES = 3 * 2 * NPOS * 16;  // 32,640
constexpr int WSLAB_VEC = 3 * 3 * 2 * COT;  // 16-byte vectors per (chunk, ky) slab: 1152
constexpr int WDMA = (WSLAB_VEC / 64 + 3) / 4;              // LDS-DMA wave-instructions per wave: 5 (18 in all)
constexpr int W_BYTES = WSLAB_VEC * 16;     // 18,432
constexpr int LDS_BYTES = X_BYTES + 2 * W_BYTES;            // 69,504 -> 2 workgroups per CU

struct Conv6Args {
  float* out;
  const float* in;
  const u32x4* wp;      // packed bf16 pieces
  const float* bias;
  const float* shift;
  const float* res1;
  const float* res2;
  int shift_stride;
  int B, Cin, Cout, H, W, Hin, Win;
  int tiles_x, tiles_y, n_cot, n_chunks;
};

// exact 3-way truncation split of two fp32 values, packed as bf16 pairs (lo = a, hi = b)
__device__ __forceinline__ void split_pack(float a, float b, unsigned& p1, unsigned& p2, unsigned& p3) {
  const unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
  const unsigned a1 = ua & 0xFFFF0000u, b1 = ub & 0xFFFF0000u;
  const float ra = a - __uint_as_float(a1), rb = b - __uint_as_float(b1);
  const unsigned a2 = __float_as_uint(ra) & 0xFFFF0000u, b2 = __float_as_uint(rb) & 0xFFFF0000u;
  const float sa = ra - __uint_as_float(a2), sb = rb - __uint_as_float(b2);
  p1 = (a1 >> 16) | b1;
  p2 = (a2 >> 16) | b2;
  p3 = (__float_as_uint(sa) >> 16) | (__float_as_uint(sb) & 0xFFFF0000u);
}

template <int MODE>
__global__ __launch_bounds__(NT, 2) void k_conv6(const Conv6Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32x4* Xs = reinterpret_cast<u32x4*>(smem);                       // [piece][h][pos]
  u32x4* Ws = reinterpret_cast<u32x4*>(smem + X_BYTES);             // [buf][piece][kx][h][co]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;

  int bid = blockIdx.x;
  const int cot = bid % a.n_cot; bid /= a.n_cot;
  const int tx = bid % a.tiles_x; bid /= a.tiles_x;
  const int ty = bid % a.tiles_y; bid /= a.tiles_y;
  const int b = bid;
  const int x0 = tx * TW, y0 = ty * TH;
  const int HWin = a.Hin * a.Win;

  // ---- staging plan: item e -> (h, position); offsets always in bounds, masking by select ----
  int xoff[XI];
  unsigned xvalid = 0;
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const int e = tid + NT * i;
    const int h = e / NPOS;
    const int pos = e - h * NPOS;
    const int r = pos / PW;
    const int col = pos - r * PW;
    const int gy = y0 + r - 1, gx = x0 + col - 1;
    const bool ok = (e < XITEMS) && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
    int off;
    if (MODE == DS_LOAD_PLAIN) off = gy * a.Win + gx;
    else if (MODE == DS_LOAD_MAXPOOL2) off = (2 * gy) * a.Win + 2 * gx;
    else off = (gy >> 1) * a.Win + (gx >> 1);
    xoff[i] = ok ? (8 * h * HWin + off) : 0;
    if (ok) xvalid |= (1u << i);
  }
  const float* in_b = a.in + (size_t)b * a.Cin * HWin;
  const u32x4* wp = a.wp + (size_t)cot * a.n_chunks * 3 * WSLAB_VEC;

  float xr[XI][8];
  unsigned xok = 0;      // bit (8*i + k): element k of item i is real data

  auto x_load = [&](int chunk) {
    const int cbase = chunk * KC;
    const float* src = in_b + (size_t)cbase * HWin;
    xok = 0;
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const int e = tid + NT * i;
      const int h = e / NPOS;
      const bool pos_ok = (xvalid >> i) & 1u;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const bool ok = pos_ok && (cbase + 8 * h + k < a.Cin);
        const float* p = src + (ok ? xoff[i] + k * HWin : 0);
        if (MODE == DS_LOAD_MAXPOOL2) {
          const float2 t0 = *reinterpret_cast<const float2*>(p);
          const float2 t1 = *reinterpret_cast<const float2*>(p + a.Win);
          xr[i][k] = fmaxf(fmaxf(t0.x, t0.y), fmaxf(t1.x, t1.y));
        } else {
          xr[i][k] = *p;
        }
        if (ok) xok |= 1u << (8 * i + k);
      }
    }
  };
  auto x_store = [&]() {
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const int e = tid + NT * i;
      if (NT * (i + 1) <= XITEMS || e < XITEMS) {
        u32x4 q1, q2, q3;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float v0 = ((xok >> (8 * i + 2 * k)) & 1u) ? xr[i][2 * k] : 0.f;
          const float v1 = ((xok >> (8 * i + 2 * k + 1)) & 1u) ? xr[i][2 * k + 1] : 0.f;
          unsigned p1, p2, p3;
          split_pack(v0, v1, p1, p2, p3);
          q1[k] = p1; q2[k] = p2; q3[k] = p3;
        }
        Xs[0 * 2 * NPOS + e] = q1;        // e = h*NPOS + pos
        Xs[1 * 2 * NPOS + e] = q2;
        Xs[2 * 2 * NPOS + e] = q3;
      }
    }
  };
  // weight slab: a contiguous 18 KiB copy -> LDS-DMA (global_load_lds_dwordx4, no staging
  // registers): wave-instruction k moves vectors [64k, 64k+64) to LDS at slab + 1 KiB*k.
  auto w_dma = [&](int step, int buf) {    // step = chunk*3 + ky
    const u32x4* src = wp + (size_t)step * WSLAB_VEC;
    u32x4* dst = Ws + buf * WSLAB_VEC;
#pragma unroll
    for (int i = 0; i < WDMA; ++i) {
      const int k = wv + 4 * i;              // wave-uniform
      if (k < WSLAB_VEC / 64) {
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(src + 64 * k + lane),
            (__attribute__((address_space(3))) void*)(dst + 64 * k), 16, 0, 0);
      }
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[m][r][q] = 0.f;

  struct Frag { bf16x8 a[3][2]; bf16x8 b[3][2]; };
  auto frag_load = [&](Frag& f, const u32x4* wb, int ky, int kx) {
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
      for (int m = 0; m < 2; ++m)
        f.a[p][m] = *reinterpret_cast<const bf16x8*>(&wb[((p * 3 + kx) * 2 + lh) * COT + 32 * m + li]);
#pragma unroll
      for (int r = 0; r < 2; ++r)
        f.b[p][r] = *reinterpret_cast<const bf16x8*>(&Xs[(p * 2 + lh) * NPOS + (2 * wv + r + ky) * PW + li + kx]);
    }
  };
  auto frag_mma = [&](const Frag& f) {
    // small cross terms first, the leading a1*b1 last
    constexpr int PA[6] = {0, 2, 1, 0, 1, 0};
    constexpr int PB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 2; ++r)
          acc[m][r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[PA[t]][m], f.b[PB[t]][r], acc[m][r], 0, 0, 0);
  };

  auto interleave_reads_with_mfma = [&]() {
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
  };

  // ---- prologue: chunk 0 input patch + step 0 weights ----
  x_load(0);
  w_dma(0, 0);
  x_store();
  __syncthreads();

  const int n_steps = a.n_chunks * 3;
  Frag f0, f1;
  for (int chunk = 0; chunk < a.n_chunks; ++chunk) {
    const bool more_chunks = chunk + 1 < a.n_chunks;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int step = chunk * 3 + ky;
      const u32x4* wb = Ws + (step & 1) * WSLAB_VEC;
      const bool more = step + 1 < n_steps;
      frag_load(f0, wb, ky, 0);
      if (more) w_dma(step + 1, (step + 1) & 1);  // next slab flies under this step's MFMAs
      if (ky == 0 && more_chunks) x_load(chunk + 1);   // next patch flies under this chunk's MFMAs
      __builtin_amdgcn_sched_barrier(0);
      // Operand prefetch: the 12 ds_read_b128 of the next kx are slotted one per MFMA into the
      // current kx's MFMA block (the wait in front of the block then covers only operands that
      // were requested a whole block earlier).
      frag_load(f1, wb, ky, 1);
      frag_mma(f0);
      interleave_reads_with_mfma();
      __builtin_amdgcn_sched_barrier(0);
      frag_load(f0, wb, ky, 2);
      frag_mma(f1);
      interleave_reads_with_mfma();
      __builtin_amdgcn_sched_barrier(0);
      frag_mma(f0);
      __builtin_amdgcn_sched_barrier(0);
      if (ky == 2 && more_chunks) {
        __syncthreads();                          // every wave is done reading the current patch
        x_store();
      }
      __syncthreads();
    }
  }

  // ---- epilogue (identical to the exact-fp32 kernel) ----
  const int gx = x0 + li;
  const size_t plane = (size_t)a.H * a.W;
  const bool has_bias = a.bias != nullptr, has_shift = a.shift != nullptr;
  const bool has_r1 = a.res1 != nullptr, has_r2 = a.res2 != nullptr;
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    float bv[16], sv[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int co = cot * COT + 32 * m + (q & 3) + 8 * (q >> 2) + 4 * lh;
      const int cs = co < a.Cout ? co : 0;
      bv[q] = has_bias ? a.bias[cs] : 0.f;
      sv[q] = has_shift ? a.shift[(size_t)b * a.shift_stride + cs] : 0.f;
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int gy = y0 + 2 * wv + r;
      const bool rowok = gy < a.H && gx < a.W;
      size_t idx[16];
      float r1[16], r2[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int co = cot * COT + 32 * m + (q & 3) + 8 * (q >> 2) + 4 * lh;
        const bool ok = rowok && co < a.Cout;
        idx[q] = ok ? ((size_t)b * a.Cout + co) * plane + (size_t)gy * a.W + gx : (size_t)0;
      }
      if (has_r1) {
#pragma unroll
        for (int q = 0; q < 16; ++q) r1[q] = a.res1[idx[q]];
      }
      if (has_r2) {
#pragma unroll
        for (int q = 0; q < 16; ++q) r2[q] = a.res2[idx[q]];
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int co = cot * COT + 32 * m + (q & 3) + 8 * (q >> 2) + 4 * lh;
        float v = acc[m][r][q];
        if (has_bias) v = v + bv[q];
        if (has_shift) v = v + sv[q];
        if (has_r1) v = v + r1[q];
        if (has_r2) v = v + r2[q];
        if (rowok && co < a.Cout) a.out[idx[q]] = v;
      }
    }
  }
}

// torch [Cout][Cin][3][3] fp32 -> [cot][chunk][ky][piece][kx][h][co 64][ci 8] bf16, zero padded
__global__ void k_pack6(unsigned short* packed, const float* __restrict__ w, int Cout, int Cin, int n_chunks,
                        size_t total) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  size_t t = i;
  const int c8 = t % 8; t /= 8;
  const int co64 = t % COT; t /= COT;
  const int h = t % 2; t /= 2;
  const int kx = t % 3; t /= 3;
  const int piece = t % 3; t /= 3;
  const int ky = t % 3; t /= 3;
  const int chunk = t % n_chunks; t /= n_chunks;
  const int cot = (int)t;
  const int co = cot * COT + co64, ci = chunk * KC + 8 * h + c8;
  float v = 0.f;
  if (co < Cout && ci < Cin) v = w[((size_t)co * Cin + ci) * 9 + ky * 3 + kx];
  const unsigned u = __float_as_uint(v);
  const unsigned p1 = u & 0xFFFF0000u;
  const float r = v - __uint_as_float(p1);
  const unsigned p2 = __float_as_uint(r) & 0xFFFF0000u;
  const float s = r - __uint_as_float(p2);
  const unsigned p3 = __float_as_uint(s);
  const unsigned sel = piece == 0 ? p1 : (piece == 1 ? p2 : p3);
  packed[i] = (unsigned short)(sel >> 16);
}

template <int MODE>
int launch_conv6(const Conv6Args& a, hipStream_t s) {
  {
    const int rc = ds::ensure_dynamic_lds<&k_conv6<MODE>>((int)(LDS_BYTES), "hipFuncSetAttribute(conv6)");
    if (rc != DS_OK) return rc;
  }
  const long long blocks = (long long)a.B * a.tiles_y * a.tiles_x * a.n_cot;
  DS_REQUIRE(blocks > 0 && blocks < (1ll << 31), DS_ERR_SHAPE, "ds_conv2d_x6: grid of %lld workgroups is out of range", blocks);
  hipLaunchKernelGGL((k_conv6<MODE>), dim3((unsigned)blocks), dim3(NT), LDS_BYTES, s, a);
  DS_CHECK_LAUNCH("ds_conv2d_x6");
  return DS_OK;
}

}  // namespace

extern "C" {

size_t ds_conv2d_x6_packed_bytes(int Cout, int Cin) {
  if (Cout <= 0 || Cin <= 0) return 0;
  const size_t n_cot = (Cout + COT - 1) / COT, n_chunks = (Cin + KC - 1) / KC;
  return n_cot * n_chunks * 3 * (size_t)W_BYTES;
}

int ds_conv2d_x6_pack_weights(void* packed, const float* w, int Cout, int Cin, void* stream) {
  DS_REQUIRE(packed && w, DS_ERR_NULL, "ds_conv2d_x6_pack_weights: NULL pointer");
  DS_REQUIRE(Cout > 0 && Cin > 0, DS_ERR_SHAPE, "ds_conv2d_x6_pack_weights: Cout=%d Cin=%d", Cout, Cin);
  const int n_chunks = (Cin + KC - 1) / KC;
  const size_t total = ds_conv2d_x6_packed_bytes(Cout, Cin) / 2;
  hipLaunchKernelGGL(k_pack6, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ds::as_stream(stream),
                     reinterpret_cast<unsigned short*>(packed), w, Cout, Cin, n_chunks, total);
  DS_CHECK_LAUNCH("ds_conv2d_x6_pack_weights");
  return DS_OK;
}

int ds_conv2d_x6(float* out, const float* in, const void* w_packed, const float* bias, const float* shift,
                 int shift_stride, const float* res1, const float* res2, int B, int Cin, int Cout, int H, int W,
                 int load_mode, void* stream) {
  DS_REQUIRE(out && in && w_packed, DS_ERR_NULL, "ds_conv2d_x6: NULL pointer");
  DS_REQUIRE(B >= 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, DS_ERR_SHAPE,
             "ds_conv2d_x6: bad shape B=%d Cin=%d Cout=%d H=%d W=%d", B, Cin, Cout, H, W);
  DS_REQUIRE(load_mode >= 0 && load_mode <= 2, DS_ERR_UNSUPPORTED, "ds_conv2d_x6: load_mode %d", load_mode);
  DS_REQUIRE(load_mode != DS_LOAD_UPSAMPLE2 || (H % 2 == 0 && W % 2 == 0), DS_ERR_SHAPE,
             "ds_conv2d_x6: UPSAMPLE2 needs even output H, W (got %d x %d)", H, W);
  DS_REQUIRE(shift == nullptr || shift_stride == 0 || shift_stride >= Cout, DS_ERR_SHAPE,
             "ds_conv2d_x6: shift_stride %d < Cout %d", shift_stride, Cout);
  DS_REQUIRE((reinterpret_cast<uintptr_t>(w_packed) & 15u) == 0, DS_ERR_SHAPE, "ds_conv2d_x6: w_packed must be 16-byte aligned");
  DS_REQUIRE(load_mode != DS_LOAD_MAXPOOL2 || (reinterpret_cast<uintptr_t>(in) & 7u) == 0, DS_ERR_SHAPE,
             "ds_conv2d_x6: MAXPOOL2 input must be 8-byte aligned");
  if (B == 0) return DS_OK;
  Conv6Args a;
  a.out = out; a.in = in; a.wp = reinterpret_cast<const u32x4*>(w_packed); a.bias = bias; a.shift = shift;
  a.res1 = res1; a.res2 = res2; a.shift_stride = shift_stride;
  a.B = B; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W;
  a.Hin = load_mode == DS_LOAD_MAXPOOL2 ? 2 * H : (load_mode == DS_LOAD_UPSAMPLE2 ? H / 2 : H);
  a.Win = load_mode == DS_LOAD_MAXPOOL2 ? 2 * W : (load_mode == DS_LOAD_UPSAMPLE2 ? W / 2 : W);
  DS_REQUIRE((long long)Cin * a.Hin * a.Win < (1ll << 31), DS_ERR_SHAPE, "ds_conv2d_x6: per-sample input exceeds 2^31 floats");
  a.tiles_x = (W + TW - 1) / TW; a.tiles_y = (H + TH - 1) / TH;
  a.n_cot = (Cout + COT - 1) / COT;
  a.n_chunks = (Cin + KC - 1) / KC;
  hipStream_t s = ds::as_stream(stream);
  if (load_mode == DS_LOAD_PLAIN) return launch_conv6<DS_LOAD_PLAIN>(a, s);
  if (load_mode == DS_LOAD_MAXPOOL2) return launch_conv6<DS_LOAD_MAXPOOL2>(a, s);
  return launch_conv6<DS_LOAD_UPSAMPLE2>(a, s);
}

}  // extern "C"
