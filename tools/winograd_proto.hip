// Prototype: Winograd F(2x2, 3x3) for the 3x3 convolutions with the fp16x3 split (fp32 accuracy on the fp16 matrix cores),
// built to measure whether 2.25x fewer multiplies buy anything on a chip that is power-limited in the matrix pipe
// (VERDICT r1 item 4b).  Standalone: random data, self-check against a direct fp64 convolution on a small
// case, then timings at the three layer shapes of config 2 next to the shipped direct kernel (ds_conv2d_h3).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/wino tools/winograd_proto.hip && /tmp/wino
//
// Formulation:  Y = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A  per 2x2 output tile, d = its 4x4 input patch.
//   workgroup = 8 waves, 16x16 output pixels (8x8 tiles, T = 64) x 64 output channels; wave w owns the transform-domain
//   POINTS 2w, 2w+1 (of 16): its accumulators are M_p[64 co][64 tiles] for two p = 128 registers.  Points are independent
//   GEMMs, so no operand is shared between waves:
//     A operand (transformed weights U_p, hi + lo fp16 pieces): straight from global / L2 into registers, 8 KB per wave
//       and 16-channel chunk, each half re-loaded right after its last use;
//     B operand (transformed input V_p): built on the fly from the raw fp32 18x18x16 patch in LDS -- four pixel reads
//       (two ds_read_b128 each), a signed 4-term sum (B^T rows have two +-1 entries), then the fp16 hi / lo split.
//   Epilogue: the 16 points of a (co, tile) meet through LDS in 8 passes of 8 channels, inverse transform, bias, store.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <algorithm>
#include "../include/diffsci_hip.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace wino {

constexpr int NT = 512, KC = 16, COT = 64;
constexpr int PR = 18;                         // patch rows / columns
constexpr int PC2 = 10;                        // columns per parity (9 used): two patch rows = 640 B = 128 mod 256
constexpr int CG_VEC = PR * 2 * PC2;           // 16-byte vectors per 4-channel group: 360
constexpr int PATCH_VEC = 4 * CG_VEC;          // 1440 (23,040 B)
constexpr int ESTRIDE = 72;                    // epilogue exchange: [p 16][co 8][72 (64 tiles used)] floats
constexpr int EPI_BYTES = 16 * 8 * ESTRIDE * 4;   // 36,864
constexpr int LDS_BYTES = 2 * PATCH_VEC * 16;     // 46,080
static_assert(EPI_BYTES <= LDS_BYTES, "epilogue aliases the patch buffers");

struct Args {
  float* out;
  const float* in;
  const u32x4* U;          // [cot][chunk][wave 8][pt 2][piece 2][mt 2][lane 64] 16-byte fragments
  const float* bias;
  float unscale;
  int B, Cin, Cout, H, W;
  int tiles_x, tiles_y, n_chunks, n_cot;
};

__device__ __forceinline__ void split2(float a, float b, unsigned& hi, unsigned& lo) {
  f16x2 h = {(_Float16)a, (_Float16)b};
  unsigned hp = __builtin_bit_cast(unsigned, h);
  asm volatile("" : "+v"(hp));
  const f16x2 hq = __builtin_bit_cast(f16x2, hp);
  f16x2 l = {(_Float16)(a - (float)hq[0]), (_Float16)(b - (float)hq[1])};
  hi = hp;
  lo = __builtin_bit_cast(unsigned, l);
}

// VARIANT bits (timing experiments; results are wrong unless 0): 1 = one epilogue pass instead of eight, 2 = B operand from
// registers (no LDS reads, no transform, no split), 4 = the first chunk's weights for every chunk (no L2 stream),
// 8 = the first chunk's patch for every chunk (no staging)
template <int VARIANT>
__global__ __launch_bounds__(NT, 1) void k_wino(const Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  f32x4* P = reinterpret_cast<f32x4*>(smem);                  // [buf 2][cg 4][row 18][parity 2][10] x 4 channels fp32
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 31, lh = lane >> 5;

  const int cot = blockIdx.x;
  const int tile_id = blockIdx.y;
  const int b = blockIdx.z;
  const int by = tile_id / a.tiles_x, bx = tile_id - by * a.tiles_x;
  const int y0 = by * 16, x0 = bx * 16;
  const int HW = a.H * a.W;

  // ---- staging plan: items = (4-channel group, patch pixel); thread handles items tid, tid + 512, tid + 1024 ----
  int xoff[3], xlds[3];
  unsigned xvalid = 0, xlive = 0;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int item = tid + NT * i;
    const bool live = item < 4 * PR * PR;
    const int cg = item / (PR * PR);
    const int pix = item - cg * (PR * PR);
    const int r = pix / PR, c = pix - r * PR;
    const int gy = y0 - 1 + r, gx = x0 - 1 + c;
    const bool ok = live && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
    xoff[i] = ok ? (cg * 4) * HW + gy * a.W + gx : 0;
    xlds[i] = ((cg * PR + r) * 2 + (c & 1)) * PC2 + (c >> 1);
    if (ok) xvalid |= 1u << i;
    if (live) xlive |= 1u << i;
  }
  const float* in_b = a.in + (size_t)b * a.Cin * HW;
  float xr[3][4];
  auto x_fetch = [&](int chunk) __attribute__((always_inline)) {
    const float* src = in_b + (size_t)chunk * KC * HW;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int k = 0; k < 4; ++k) xr[i][k] = src[xoff[i] + k * HW];
  };
  auto x_store = [&](int buf) __attribute__((always_inline)) {
    f32x4* pb = P + buf * PATCH_VEC;
#pragma unroll
    for (int i = 0; i < 3; ++i)
      if ((xlive >> i) & 1u) {
        const bool ok = (xvalid >> i) & 1u;
        pb[xlds[i]] = ok ? f32x4{xr[i][0], xr[i][1], xr[i][2], xr[i][3]} : f32x4{0.f, 0.f, 0.f, 0.f};
      }
  };

  // ---- the wave's two points: p = 4i + j; B^T row i = (k_a, s_a), (k_b, s_b) ----
  //   i = 0: (0,+)(2,-)   1: (1,+)(2,+)   2: (1,-)(2,+)   3: (1,+)(3,-)
  int poff[2][4];          // LDS vector offsets of the four pixels (a,c) (a,d) (b,c) (b,d) relative to the tile's patch origin
  float psgn[2][4];
#pragma unroll
  for (int pt = 0; pt < 2; ++pt) {
    const int p = 2 * wv + pt, i = p >> 2, j = p & 3;
    const int ka = i == 0 ? 0 : 1, kb = i == 3 ? 3 : 2;
    const float sa = i == 2 ? -1.f : 1.f, sb = (i == 0 || i == 3) ? -1.f : 1.f;
    const int lc = j == 0 ? 0 : 1, ld = j == 3 ? 3 : 2;
    const float sc = j == 2 ? -1.f : 1.f, sd = (j == 0 || j == 3) ? -1.f : 1.f;
    const int oc = (lc & 1) * PC2 + (lc >> 1), od = (ld & 1) * PC2 + (ld >> 1);
    poff[pt][0] = 2 * PC2 * ka + oc; psgn[pt][0] = sa * sc;
    poff[pt][1] = 2 * PC2 * ka + od; psgn[pt][1] = sa * sd;
    poff[pt][2] = 2 * PC2 * kb + oc; psgn[pt][2] = sb * sc;
    poff[pt][3] = 2 * PC2 * kb + od; psgn[pt][3] = sb * sd;
  }
  // lane part: tile t = 32 nb + n -> (ty, tx) = (t >> 3, t & 7); channel groups 2 lh, 2 lh + 1
  int lbase[2];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    const int t = 32 * nb + n;
    lbase[nb] = ((2 * lh * PR + 2 * (t >> 3)) * 2) * PC2 + (t & 7);
  }

  f32x16 acc[2][2][2];                                       // [pt][mt][nb]
#pragma unroll
  for (int pt = 0; pt < 2; ++pt)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[pt][mt][nb][q] = 0.f;

  const u32x4* Uw = a.U + ((size_t)cot * a.n_chunks * 8 + wv) * 512 + lane;     // + chunk * 8 * 512; 512 vectors per (chunk, wave)
  f16x8 A[2][2][2];                                          // [pt][piece][mt]
  auto a_fetch = [&](int chunk, int pt) __attribute__((always_inline)) {
    const u32x4* src = Uw + (size_t)chunk * 8 * 512 + pt * 256;
#pragma unroll
    for (int piece = 0; piece < 2; ++piece)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) A[pt][piece][mt] = __builtin_bit_cast(f16x8, src[(piece * 2 + mt) * 64]);
  };

  x_fetch(0);
  a_fetch(0, 0);
  a_fetch(0, 1);
  x_store(0);
  __syncthreads();

  for (int chunk = 0; chunk < a.n_chunks; ++chunk) {
    const bool more = chunk + 1 < a.n_chunks;
    if (more && !(VARIANT & 8)) x_fetch(chunk + 1);
    const f32x4* pb = P + ((VARIANT & 8) ? 0 : (chunk & 1)) * PATCH_VEC;
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        // V = s0 d_ac + s1 d_ad + s2 d_bc + s3 d_bd for the lane's 8 channels, then the fp16 split
        f32x4 v[2];
        if (VARIANT & 2) {
          v[0] = f32x4{1.f, 2.f, 3.f, 4.f} * psgn[pt][nb]; v[1] = v[0];
        } else
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const f32x4* q = pb + lbase[nb] + g * CG_VEC;
          f32x4 t = q[poff[pt][0]] * psgn[pt][0];
          t = t + q[poff[pt][1]] * psgn[pt][1];
          t = t + q[poff[pt][2]] * psgn[pt][2];
          t = t + q[poff[pt][3]] * psgn[pt][3];
          v[g] = t;
        }
        unsigned h0, h1, h2, h3, l0, l1, l2, l3;
        if (VARIANT & 2) {
          h0 = h1 = h2 = h3 = __builtin_bit_cast(unsigned, v[0][0]) | 0x3c003c00u; l0 = l1 = l2 = l3 = h0 ^ 0x00010001u;
        } else {
        split2(v[0][0], v[0][1], h0, l0);
        split2(v[0][2], v[0][3], h1, l1);
        split2(v[1][0], v[1][1], h2, l2);
        split2(v[1][2], v[1][3], h3, l3);
        }
        const u32x4 bh = {h0, h1, h2, h3}, bl = {l0, l1, l2, l3};
        const f16x8 Bh = __builtin_bit_cast(f16x8, bh), Bl = __builtin_bit_cast(f16x8, bl);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          acc[pt][mt][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[pt][1][mt], Bh, acc[pt][mt][nb], 0, 0, 0);
          acc[pt][mt][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[pt][0][mt], Bl, acc[pt][mt][nb], 0, 0, 0);
          acc[pt][mt][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[pt][0][mt], Bh, acc[pt][mt][nb], 0, 0, 0);
        }
      }
      if (more && !(VARIANT & 4)) a_fetch(chunk + 1, pt);    // this point's weights are dead: fetch the next chunk's
    }
    if (more && !(VARIANT & 8)) x_store((chunk + 1) & 1);
    __syncthreads();
  }

  // ---- epilogue: 8 passes of 8 channels (mt, g): channel 32 mt + 8 g + (q & 3) + 4 lh ----
  float* E = reinterpret_cast<float*>(smem);
  const int co8 = tid >> 6, tl = tid & 63;                   // reader: one (channel, tile) per thread
  const int ty = tl >> 3, tx = tl & 7;
  const int gy = y0 + 2 * ty, gx = x0 + 2 * tx;
#pragma unroll
  for (int mt = 0; mt < ((VARIANT & 1) ? 1 : 2); ++mt)
#pragma unroll
    for (int g = 0; g < ((VARIANT & 1) ? 1 : 4); ++g) {
      if (VARIANT & 1) {                                       // keep every accumulator alive
#pragma unroll
        for (int pt = 0; pt < 2; ++pt)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int q = 1; q < 16; ++q) acc[pt][0][nb][q & 3] += acc[pt][0][nb][q] + acc[pt][1][nb][q];
      }
#pragma unroll
      for (int pt = 0; pt < 2; ++pt)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int qq = 0; qq < 4; ++qq)
            E[((2 * wv + pt) * 8 + qq + 4 * lh) * ESTRIDE + 32 * nb + n] = acc[pt][mt][nb][4 * g + qq];
      __syncthreads();
      float m[16];
#pragma unroll
      for (int p = 0; p < 16; ++p) m[p] = E[(p * 8 + co8) * ESTRIDE + tl];
      // Y = A^T M A, A^T = [1 1 1 0; 0 1 -1 -1]
      float r0[4], r1[4];                                      // rows of A^T M
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        r0[j] = (m[j] + m[4 + j]) + m[8 + j];
        r1[j] = (m[4 + j] - m[8 + j]) - m[12 + j];
      }
      const int co = cot * COT + 32 * mt + 8 * g + co8;
      const float bv = a.bias ? a.bias[co] : 0.f;
      f32x2 o0 = {((r0[0] + r0[1]) + r0[2]) * a.unscale + bv, ((r0[1] - r0[2]) - r0[3]) * a.unscale + bv};
      f32x2 o1 = {((r1[0] + r1[1]) + r1[2]) * a.unscale + bv, ((r1[1] - r1[2]) - r1[3]) * a.unscale + bv};
      float* dst = a.out + ((size_t)b * a.Cout + co) * HW + (size_t)gy * a.W + gx;
      *reinterpret_cast<f32x2*>(dst) = o0;
      *reinterpret_cast<f32x2*>(dst + a.W) = o1;
      __syncthreads();
    }
}

}  // namespace wino

// ---- host: weight transform and packing ----
static void pack_U(std::vector<_Float16>& packed, const std::vector<float>& w, int Cout, int Cin, float scale) {
  const int n_cot = Cout / 64, n_chunks = Cin / 16;
  packed.assign((size_t)n_cot * n_chunks * 8 * 512 * 8, (_Float16)0.f);
  const double G[4][3] = {{1, 0, 0}, {.5, .5, .5}, {.5, -.5, .5}, {0, 0, 1}};
  for (int co = 0; co < Cout; ++co)
    for (int ci = 0; ci < Cin; ++ci) {
      const float* g = &w[((size_t)co * Cin + ci) * 9];
      double t[4][3], U[4][4];
      for (int i = 0; i < 4; ++i)
        for (int bb = 0; bb < 3; ++bb) t[i][bb] = G[i][0] * g[0 * 3 + bb] + G[i][1] * g[1 * 3 + bb] + G[i][2] * g[2 * 3 + bb];
      for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) U[i][j] = t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2];
      const int cot = co / 64, mt = (co % 64) / 32, m = co % 32;
      const int chunk = ci / 16, lh = (ci % 16) / 8, e = ci % 8;
      const int lane = lh * 32 + m;
      for (int p = 0; p < 16; ++p) {
        const float v = (float)(U[p >> 2][p & 3] * scale);
        const _Float16 hi = (_Float16)v, lo = (_Float16)(v - (float)hi);
        const int wvi = p >> 1, pt = p & 1;
        for (int piece = 0; piece < 2; ++piece) {
          const size_t vec = (((size_t)cot * n_chunks + chunk) * 8 + wvi) * 512 + pt * 256 + (piece * 2 + mt) * 64 + lane;
          packed[vec * 8 + e] = piece == 0 ? hi : lo;
        }
      }
    }
}

template <int VARIANT>
static int launch_wino_v(float* out, const float* in, const void* U, const float* bias, float unscale, int B, int Cin, int Cout, int S) {
  wino::Args a;
  a.out = out; a.in = in; a.U = reinterpret_cast<const u32x4*>(U); a.bias = bias; a.unscale = unscale;
  a.B = B; a.Cin = Cin; a.Cout = Cout; a.H = S; a.W = S;
  a.tiles_x = S / 16; a.tiles_y = S / 16; a.n_chunks = Cin / 16; a.n_cot = Cout / 64;
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wino::k_wino<VARIANT>), hipFuncAttributeMaxDynamicSharedMemorySize, wino::LDS_BYTES); attr = true; }
  hipLaunchKernelGGL(wino::k_wino<VARIANT>, dim3(a.n_cot, a.tiles_x * a.tiles_y, B), dim3(wino::NT), wino::LDS_BYTES, 0, a);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
static int g_variant = 0;
static int launch_wino(float* out, const float* in, const void* U, const float* bias, float unscale, int B, int Cin, int Cout, int S) {
  switch (g_variant) {
    case 1: return launch_wino_v<1>(out, in, U, bias, unscale, B, Cin, Cout, S);
    case 2: return launch_wino_v<2>(out, in, U, bias, unscale, B, Cin, Cout, S);
    case 3: return launch_wino_v<3>(out, in, U, bias, unscale, B, Cin, Cout, S);
    case 4: return launch_wino_v<4>(out, in, U, bias, unscale, B, Cin, Cout, S);
    case 7: return launch_wino_v<7>(out, in, U, bias, unscale, B, Cin, Cout, S);
    case 8: return launch_wino_v<8>(out, in, U, bias, unscale, B, Cin, Cout, S);
    case 15: return launch_wino_v<15>(out, in, U, bias, unscale, B, Cin, Cout, S);
    default: return launch_wino_v<0>(out, in, U, bias, unscale, B, Cin, Cout, S);
  }
}

int main(int argc, char** argv) {
  unsigned seed = 12345u;
  auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return ((seed >> 8) * (1.0f / 8388608.0f)) - 1.0f; };
  const float wscale = 256.f;
  {   // ---- self-check: B = 2, 32 -> 64 channels, 32 x 32 ----
    const int B = 2, Cin = 32, Cout = 64, S = 32;
    std::vector<float> hx((size_t)B * Cin * S * S), hw((size_t)Cout * Cin * 9), hb(Cout);
    for (auto& v : hx) v = rnd();
    for (auto& v : hw) v = rnd() * 0.06f;
    for (auto& v : hb) v = rnd();
    std::vector<_Float16> hU;
    pack_U(hU, hw, Cout, Cin, wscale);
    float *in, *out, *bias; void* U;
    hipMalloc(&in, hx.size() * 4); hipMalloc(&out, (size_t)B * Cout * S * S * 4); hipMalloc(&bias, Cout * 4); hipMalloc(&U, hU.size() * 2);
    hipMemcpy(in, hx.data(), hx.size() * 4, hipMemcpyHostToDevice); hipMemcpy(bias, hb.data(), Cout * 4, hipMemcpyHostToDevice);
    hipMemcpy(U, hU.data(), hU.size() * 2, hipMemcpyHostToDevice);
    launch_wino(out, in, U, bias, 1.f / wscale, B, Cin, Cout, S);
    if (hipDeviceSynchronize() != hipSuccess) { printf("self-check launch failed\n"); return 1; }
    std::vector<float> ho((size_t)B * Cout * S * S);
    hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost);
    double num = 0, den = 0, maxabs = 0;
    for (int b = 0; b < B; ++b)
      for (int co = 0; co < Cout; ++co)
        for (int y = 0; y < S; ++y)
          for (int x = 0; x < S; ++x) {
            double r = hb[co];
            for (int ci = 0; ci < Cin; ++ci)
              for (int ky = 0; ky < 3; ++ky)
                for (int kx = 0; kx < 3; ++kx) {
                  const int yy = y + ky - 1, xx = x + kx - 1;
                  if (yy < 0 || yy >= S || xx < 0 || xx >= S) continue;
                  r += (double)hw[((size_t)co * Cin + ci) * 9 + ky * 3 + kx] * hx[(((size_t)b * Cin + ci) * S + yy) * S + xx];
                }
            const double d = ho[(((size_t)b * Cout + co) * S + y) * S + x] - r;
            num += d * d; den += r * r; maxabs = std::max(maxabs, std::fabs(d));
          }
    printf("self-check vs fp64 direct convolution: rel-L2 %.3e, max-abs %.3e\n", std::sqrt(num / den), maxabs);
    hipFree(in); hipFree(out); hipFree(bias); hipFree(U);
    if (!(std::sqrt(num / den) < 1e-5)) { printf("FAILED\n"); return 1; }
  }
  // ---- timings: the three levels of config 2 (B = 64), next to the shipped direct kernel ----
  const int reps = argc > 1 ? atoi(argv[1]) : 200;
  g_variant = argc > 2 ? atoi(argv[2]) : 0;
  printf("variant %d\n", g_variant);
  const int shapes[3][2] = {{64, 128}, {128, 64}, {256, 32}};
  for (int round = 0; round < (g_variant ? 1 : 2); ++round)
    for (auto& sh : shapes) {
      const int B = 64, C = sh[0], S = sh[1];
      const size_t nin = (size_t)B * C * S * S;
      std::vector<float> hx(nin), hw((size_t)C * C * 9);
      for (auto& v : hx) v = rnd();
      for (auto& v : hw) v = rnd() * 0.04f;
      std::vector<_Float16> hU;
      pack_U(hU, hw, C, C, wscale);
      float *in, *out, *w; void *U, *wp;
      hipMalloc(&in, nin * 4); hipMalloc(&out, nin * 4); hipMalloc(&w, hw.size() * 4); hipMalloc(&U, hU.size() * 2);
      hipMemcpy(in, hx.data(), nin * 4, hipMemcpyHostToDevice); hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
      hipMemcpy(U, hU.data(), hU.size() * 2, hipMemcpyHostToDevice);
      hipMalloc(&wp, ds_conv2d_h3_packed_bytes(C, C));
      ds_conv2d_h3_pack_weights(wp, w, C, C, 8, nullptr);
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      float ms_w, ms_d;
      for (int i = 0; i < 20; ++i) launch_wino(out, in, U, nullptr, 1.f / wscale, B, C, C, S);
      hipEventRecord(e0);
      for (int i = 0; i < reps; ++i) launch_wino(out, in, U, nullptr, 1.f / wscale, B, C, C, S);
      hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms_w, e0, e1);
      for (int i = 0; i < 20; ++i) ds_conv2d_h3(out, in, wp, 8, nullptr, nullptr, 0, nullptr, nullptr, B, C, C, S, S, 0, nullptr, nullptr, nullptr);
      hipEventRecord(e0);
      for (int i = 0; i < reps; ++i) ds_conv2d_h3(out, in, wp, 8, nullptr, nullptr, 0, nullptr, nullptr, B, C, C, S, S, 0, nullptr, nullptr, nullptr);
      hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms_d, e0, e1);
      printf("round %d  B=%d C=%d %dx%d: winograd %.1f us   direct (ds_conv2d_h3) %.1f us   ratio %.2f\n", round, B, C, S, S,
             ms_w * 1e3 / reps, ms_d * 1e3 / reps, ms_d / ms_w);
      hipFree(in); hipFree(out); hipFree(w); hipFree(U); hipFree(wp);
    }
  return 0;
}
