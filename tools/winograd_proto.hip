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
constexpr int PATCH_USED = 4 * CG_VEC;         // 1440
constexpr int PATCH_VEC = 1536;                // padded to 96 LDS-DMA instructions of 16 vectors (12 per wave)
constexpr int NBUF = 3, NDMA = 12;
constexpr int ES = 72;                         // epilogue exchange: [p 16][co 32][72 (64 tiles used)] floats per pass
constexpr int EPI_BYTES = 16 * 32 * ES * 4;        // 147,456
constexpr int STAGE_BYTES = NBUF * PATCH_VEC * 16; // 73,728
constexpr int PLAN_BYTES = NDMA * NT * 4;           // 24,576
constexpr int LDS_BYTES = EPI_BYTES > STAGE_BYTES + PLAN_BYTES ? EPI_BYTES : STAGE_BYTES + PLAN_BYTES;

struct Args {
  float* out;
  const float* in;
  const u32x4* U;          // [cot][chunk][wave 8][pt 2][piece 2][mt 2][lane 64] 16-byte fragments, point signs folded in
  const float* bias;
  float unscale;
  int B, Cin, Cout, H, W;
  int tiles_x, tiles_y, n_chunks, n_cot;
  unsigned long long* stamps;      // [workgroup][8] s_memrealtime (100 MHz) of wave 0: start, prologue done, sum of the chunks' construct phases (to the barrier), sum of their matrix phases (to the barrier), main loop done, epilogue done
};

// fp32 pair -> fp16 hi pair + fp16 lo pair (lo = a - hi, exact in fp32, one rounding to fp16): three instructions
__device__ __forceinline__ void split2(float a, float b, unsigned& hi, unsigned& lo) {
  unsigned hp, lp;
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hp) : "v"(a), "v"(b));
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lp) : "v"(hp), "v"(a));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lp) : "v"(hp), "v"(b));
  hi = hp;
  lo = lp;
}

// VARIANT bits (timing experiments; results are wrong unless 0): 1 = one epilogue pass instead of two, 2 = B operand from
// registers (no LDS reads, no transform, no split), 4 = the first chunk's weights for every chunk (no L2 stream),
// 8 = no patch staging after the prologue
template <int VARIANT>
__global__ __launch_bounds__(NT, 1) void k_wino(const Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  f32x4* P = reinterpret_cast<f32x4*>(smem);                  // [buf 3][cg 4][row 18][parity 2][10] x 4 channels fp32
  float* Pf = reinterpret_cast<float*>(smem);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 31, lh = lane >> 5;

  const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
  unsigned long long t_compute = 0, t_wait = 0, t_c = 0, t_w = 0, t_m = 0;
  const int cot = blockIdx.x;
  const int tile_id = blockIdx.y;
  const int b = blockIdx.z;
  const int by = tile_id / a.tiles_x, bx = tile_id - by * a.tiles_x;
  const int y0 = by * 16, x0 = bx * 16;
  const int HW = a.H * a.W;

  // ---- LDS-DMA plan: instruction k = wv + 8 i moves 64 dwords = 16 (pixel slot) x 4 channels; lane = 4 * slot + channel.
  //      Every lane always reads a valid address (clamped); lanes whose pixel is outside the image zero their own dword
  //      after their own wait (zero padding), so the number of outstanding operations is the same for every wave. ----
  int* DT = reinterpret_cast<int*>(smem + STAGE_BYTES);      // [12][512] source offsets of the DMA plan: parked in LDS, the main loop has no registers to spare
  unsigned dzero = 0;
#pragma unroll
  for (int i = 0; i < NDMA; ++i) {
    const int s = 16 * (wv + 8 * i) + (lane >> 2), ci = lane & 3;
    const int cg = s / CG_VEC, rem = s - cg * CG_VEC;
    const int row = rem / (2 * PC2), r2 = rem - row * (2 * PC2);
    const int par = r2 / PC2, col2 = r2 - par * PC2;
    const int col = 2 * col2 + par;
    const bool live = s < PATCH_USED && col2 < 9;
    const int gy = y0 - 1 + row, gx = x0 - 1 + col;
    const bool inimg = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
    const int cy = gy < 0 ? 0 : (gy >= a.H ? a.H - 1 : gy), cx = gx < 0 ? 0 : (gx >= a.W ? a.W - 1 : gx);
    DT[i * NT + tid] = ((cg > 3 ? 3 : cg) * 4 + ci) * HW + cy * a.W + cx;
    if (live && !inimg) dzero |= 1u << i;
  }
  const float* in_b = a.in + (size_t)b * a.Cin * HW;
  auto dma = [&](int chunk, int buf) __attribute__((always_inline)) {
    const float* src = in_b + (size_t)chunk * KC * HW;
    float* dst = Pf + buf * (PATCH_VEC * 4) + 64 * wv;
#pragma unroll
    for (int i = 0; i < NDMA; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + DT[i * NT + tid]),
                                       (__attribute__((address_space(3))) void*)(dst + 512 * i), 4, 0, 0);
  };
  auto zero_fix = [&](int buf) __attribute__((always_inline)) {
    if (dzero) {
      float* dst = Pf + buf * (PATCH_VEC * 4) + 64 * wv + lane;
#pragma unroll
      for (int i = 0; i < NDMA; ++i)
        if ((dzero >> i) & 1u) dst[512 * i] = 0.f;
    }
  };

  // ---- the wave's two points: row i = wv >> 1 of B^T, columns j0, j0 + 1 with j0 = 2 (wv & 1); point signs are folded
  //      into U.  r_l = d[ka][l] + beta d[kb][l] for three columns l = c0 .. c0 + 2, then
  //      j0 = 0: V_a = r0 - r2, V_b = r1 + r2;   j0 = 2: V_a = r0 - r1, V_b = r0 - r2 ----
  const int irow = wv >> 1, j0 = 2 * (wv & 1);
  const int ka = irow == 0 ? 0 : 1, kb = irow == 3 ? 3 : 2;
  const float beta = irow == 1 ? 1.f : -1.f;
  // three patch columns per wave, ordered so that V_a = r[0] - r[1] for both kinds of wave and V_b = r[2] + gamma * r_d:
  //   j0 = 0: columns (0, 2, 1), r_d = r[1], gamma = +1 (V_0 = d0 - d2, V_1 = d1 + d2)
  //   j0 = 2: columns (1, 2, 3), r_d = r[0], gamma = -1 (-V_2 = d1 - d2, -V_3 = d3 - d1; the signs live in U)
  int oa[3], ob[3];
  const float gamma = j0 == 0 ? 1.f : -1.f;
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int l = j0 == 0 ? (t == 0 ? 0 : (t == 1 ? 2 : 1)) : 1 + t;
    const int co = (l & 1) * PC2 + (l >> 1);
    oa[t] = 2 * PC2 * ka + co;
    ob[t] = 2 * PC2 * kb + co;
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

  // ---- main loop, software-pipelined inside every wave: a STEP = the 12 matrix instructions of one half of the tiles (nb)
  //      interleaved with building the B fragments of the NEXT step -- 12 ds_read_b128 behind the first four MFMAs, the
  //      transform and the split behind the other eight -- so the vector / LDS work hides under the wave's own matrix time
  //      (as separate phases it took 1.1-1.25 us per chunk against 0.43 us of matrix issue).
  //      chunk c: step 0 = MFMA(c, nb 0) | build (c, nb 1);  barrier;  step 1 = MFMA(c, nb 1) | build (c + 1, nb 0), the next
  //      chunk's weights behind each point's last use, then the patch DMA three chunks ahead (issued AFTER the weight loads:
  //      vmcnt retires in order, and the weights are consumed first).
  struct BFr { f16x8 h[2], l[2]; };                          // [pt]
  auto build = [&](BFr& Bn, const f32x4* pb, int nb) __attribute__((always_inline)) {       // the prologue's, not scheduled
    f32x4 va[2], vb[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const f32x4* q = pb + lbase[nb] + g * CG_VEC;
      const f32x4 r0 = q[oa[0]] + q[ob[0]] * beta, r1 = q[oa[1]] + q[ob[1]] * beta, r2 = q[oa[2]] + q[ob[2]] * beta;
      va[g] = r0 - r1;
      vb[g] = r2 + (j0 == 0 ? r1 : r0) * gamma;
    }
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
      const f32x4 v0 = pt == 0 ? va[0] : vb[0], v1 = pt == 0 ? va[1] : vb[1];
      unsigned h0, h1, h2, h3, l0, l1, l2, l3;
      split2(v0[0], v0[1], h0, l0); split2(v0[2], v0[3], h1, l1); split2(v1[0], v1[1], h2, l2); split2(v1[2], v1[3], h3, l3);
      const u32x4 bh = {h0, h1, h2, h3}, bl = {l0, l1, l2, l3};
      Bn.h[pt] = __builtin_bit_cast(f16x8, bh); Bn.l[pt] = __builtin_bit_cast(f16x8, bl);
    }
  };
  // One step, scheduled by hand (sched_barrier(0) pins every group behind "its" MFMA): MFMA i of the current fragments, then
  // a slice of the next fragments' construction.  i = 6 pt + 2 t + mt; the weights of point 0 are dead after MFMA 5.
  auto step = [&](const BFr& Bc, int nbc, BFr& Bn, const f32x4* pbn, int nbn, bool do_build, int fetch_chunk)
      __attribute__((always_inline)) {
    f32x4 ra[2][3], rb[2][3], va[2], vb[2];
    unsigned hh[2][4], ll[2][4];
    auto MF = [&](int i) __attribute__((always_inline)) {
      const int pt = i / 6, t = (i % 6) >> 1, mt = i & 1;
      acc[pt][mt][nbc] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[pt][t == 0 ? 1 : 0][mt], t == 1 ? Bc.l[pt] : Bc.h[pt],
                                                                acc[pt][mt][nbc], 0, 0, 0);
    };
    auto reads = [&](int g) __attribute__((always_inline)) {
      const f32x4* q = pbn + lbase[nbn] + g * CG_VEC;
#pragma unroll
      for (int c = 0; c < 3; ++c) { ra[g][c] = q[oa[c]]; rb[g][c] = q[ob[c]]; }
    };
    auto transform = [&](int g) __attribute__((always_inline)) {
      const f32x4 r0 = ra[g][0] + rb[g][0] * beta, r1 = ra[g][1] + rb[g][1] * beta, r2 = ra[g][2] + rb[g][2] * beta;
      va[g] = r0 - r1;
      vb[g] = r2 + (j0 == 0 ? r1 : r0) * gamma;
    };
    auto splits = [&](int pt, int half) __attribute__((always_inline)) {          // half: channel group g = half
      const f32x4 v = pt == 0 ? va[half] : vb[half];
      split2(v[0], v[1], hh[pt][2 * half], ll[pt][2 * half]);
      split2(v[2], v[3], hh[pt][2 * half + 1], ll[pt][2 * half + 1]);
    };
    auto finish = [&](int pt) __attribute__((always_inline)) {
      const u32x4 bh = {hh[pt][0], hh[pt][1], hh[pt][2], hh[pt][3]}, bl = {ll[pt][0], ll[pt][1], ll[pt][2], ll[pt][3]};
      Bn.h[pt] = __builtin_bit_cast(f16x8, bh); Bn.l[pt] = __builtin_bit_cast(f16x8, bl);
    };
#define SB __builtin_amdgcn_sched_barrier(0)
    SB; MF(0); if (do_build) reads(0);
    SB; MF(1); if (do_build) transform(0);
    SB; MF(2); if (do_build) reads(1);
    SB; MF(3); if (do_build) splits(0, 0);
    SB; MF(4); if (do_build) transform(1);
    SB; MF(5); if (do_build) splits(1, 0);
    SB; if (fetch_chunk >= 0 && !(VARIANT & 4)) a_fetch(fetch_chunk, 0);
    SB; MF(6); if (do_build) { splits(0, 1); finish(0); }
    SB; MF(7); if (do_build) { splits(1, 1); finish(1); }
    SB; MF(8);
    SB; MF(9);
    SB; MF(10);
    SB; MF(11);
    SB; if (fetch_chunk >= 0 && !(VARIANT & 4)) a_fetch(fetch_chunk, 1);
    SB;
#undef SB
  };

  // ---- prologue: three patches, the first weights, the first fragments ----
  dma(0, 0);
  if (a.n_chunks > 1) dma(1, 1);
  if (a.n_chunks > 2) dma(2, 2);
  a_fetch(0, 0);
  a_fetch(0, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  zero_fix(0);
  if (a.n_chunks > 1) zero_fix(1);
  if (a.n_chunks > 2) zero_fix(2);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  BFr B0, B1;
  build(B0, P, 0);
  const unsigned long long t_pro = __builtin_amdgcn_s_memrealtime();

  for (int chunk = 0; chunk < a.n_chunks; ++chunk) {
    const bool more = chunk + 1 < a.n_chunks;
    step(B0, 0, B1, P + (chunk % NBUF) * PATCH_VEC, 1, true, -1);
    if (more) {
      if (chunk + 2 < a.n_chunks) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");   // all but the DMA issued one iteration ago
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      zero_fix((chunk + 1) % NBUF);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    step(B1, 1, B0, P + ((chunk + 1) % NBUF) * PATCH_VEC, 0, more, more ? chunk + 1 : -1);
    if (chunk + 3 < a.n_chunks && !(VARIANT & 8)) dma(chunk + 3, chunk % NBUF);
  }
  const unsigned long long t_main = __builtin_amdgcn_s_memrealtime();

  // ---- epilogue: two passes of 32 channels (mt); every wave parks its two points' 32 x 64 blocks in LDS, then each thread
  //      gathers the 16 points of two (channel, tile pair) items, applies A^T . A and stores 2 rows x 4 pixels ----
  float* E = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int mt = 0; mt < ((VARIANT & 1) ? 1 : 2); ++mt) {
    if (VARIANT & 1) {                                       // keep every accumulator alive
#pragma unroll
      for (int pt = 0; pt < 2; ++pt)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int q = 0; q < 16; ++q) acc[pt][0][nb][q] += acc[pt][1][nb][q];
    }
#pragma unroll
    for (int pt = 0; pt < 2; ++pt)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int q = 0; q < 16; ++q)
          E[((2 * wv + pt) * 32 + (q & 3) + 8 * (q >> 2) + 4 * lh) * ES + 32 * nb + n] = acc[pt][mt][nb][q];
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int item = tid + NT * it;
      const int co32 = item >> 5, ty = (item >> 2) & 7, txp = item & 3;
      f32x2 m[16];
#pragma unroll
      for (int p = 0; p < 16; ++p) m[p] = *reinterpret_cast<const f32x2*>(&E[(p * 32 + co32) * ES + 8 * ty + 2 * txp]);
      // Y = A^T M A, A^T = [1 1 1 0; 0 1 -1 -1], for the two tiles at once
      f32x2 r0[4], r1[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        r0[j] = (m[j] + m[4 + j]) + m[8 + j];
        r1[j] = (m[4 + j] - m[8 + j]) - m[12 + j];
      }
      const int co = cot * COT + 32 * mt + co32;
      const float bv = a.bias ? a.bias[co] : 0.f;
      const f32x2 y00 = ((r0[0] + r0[1]) + r0[2]) * a.unscale + bv, y01 = ((r0[1] - r0[2]) - r0[3]) * a.unscale + bv;
      const f32x2 y10 = ((r1[0] + r1[1]) + r1[2]) * a.unscale + bv, y11 = ((r1[1] - r1[2]) - r1[3]) * a.unscale + bv;
      float* dst = a.out + ((size_t)b * a.Cout + co) * HW + (size_t)(y0 + 2 * ty) * a.W + x0 + 4 * txp;
      *reinterpret_cast<f32x4*>(dst) = f32x4{y00[0], y01[0], y00[1], y01[1]};
      *reinterpret_cast<f32x4*>(dst + a.W) = f32x4{y10[0], y11[0], y10[1], y11[1]};
    }
    if (mt == 0 && !(VARIANT & 1)) __syncthreads();
  }
  if (a.stamps && tid == 0) {
    unsigned long long* st = a.stamps + 8 * ((size_t)blockIdx.x + gridDim.x * (blockIdx.y + (size_t)gridDim.y * blockIdx.z));
    st[0] = t_start; st[1] = t_pro; st[2] = t_compute; st[3] = t_wait; st[4] = t_main; st[5] = __builtin_amdgcn_s_memrealtime(); st[6] = t_c; st[7] = t_w | (t_m << 32);
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
        const double sgn = (((p >> 2) == 2) != ((p & 3) >= 2)) ? -1.0 : 1.0;     // point signs folded into U (see k_wino: rows i = 2, columns j = 2, 3)
        const float v = (float)(sgn * U[p >> 2][p & 3] * scale);
        const _Float16 hi = (_Float16)v, lo = (_Float16)(v - (float)hi);
        const int wvi = p >> 1, pt = p & 1;
        for (int piece = 0; piece < 2; ++piece) {
          const size_t vec = (((size_t)cot * n_chunks + chunk) * 8 + wvi) * 512 + pt * 256 + (piece * 2 + mt) * 64 + lane;
          packed[vec * 8 + e] = piece == 0 ? hi : lo;
        }
      }
    }
}

static unsigned long long* g_stamps = nullptr;
template <int VARIANT>
static int launch_wino_v(float* out, const float* in, const void* U, const float* bias, float unscale, int B, int Cin, int Cout, int S) {
  wino::Args a;
  a.out = out; a.in = in; a.U = reinterpret_cast<const u32x4*>(U); a.bias = bias; a.unscale = unscale;
  a.B = B; a.Cin = Cin; a.Cout = Cout; a.H = S; a.W = S;
  a.tiles_x = S / 16; a.tiles_y = S / 16; a.n_chunks = Cin / 16; a.n_cot = Cout / 64; a.stamps = g_stamps;
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
      {
        const int nwg = (C / 64) * (S / 16) * (S / 16) * B;
        hipMalloc(&g_stamps, (size_t)nwg * 64);
        launch_wino(out, in, U, nullptr, 1.f / wscale, B, C, C, S);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h((size_t)nwg * 8);
        hipMemcpy(h.data(), g_stamps, h.size() * 8, hipMemcpyDeviceToHost);
        double pro = 0, comp = 0, wait = 0, epi = 0, life = 0, sc = 0, sw = 0, sm = 0;
        unsigned long long tmin = ~0ull, tmax = 0;
        for (int w = 0; w < nwg; ++w) {
          pro += h[w * 8 + 1] - h[w * 8]; comp += h[w * 8 + 2]; wait += h[w * 8 + 3]; epi += h[w * 8 + 5] - h[w * 8 + 4]; life += h[w * 8 + 5] - h[w * 8];
          sc += h[w * 8 + 6]; sw += h[w * 8 + 7] & 0xffffffffull; sm += h[w * 8 + 7] >> 32;
          tmin = std::min(tmin, h[w * 8]); tmax = std::max(tmax, h[w * 8 + 5]);
        }
        double mainl = 0;
        for (int w = 0; w < nwg; ++w) mainl += h[w * 8 + 4] - h[w * 8 + 1];
        printf("    %d workgroups, span %.1f us; per workgroup (us): prologue %.2f, main loop %.2f (%.2f per chunk), epilogue %.2f, life %.2f\n",
               nwg, (tmax - tmin) / 100.0, pro / nwg / 100, mainl / nwg / 100, mainl / nwg / 100 / (C / 16), epi / nwg / 100, life / nwg / 100);
        hipFree(g_stamps); g_stamps = nullptr;
      }
      hipFree(in); hipFree(out); hipFree(w); hipFree(U); hipFree(wp);
    }
  return 0;
}
