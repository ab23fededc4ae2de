// fp32 convolution (3x3 / 1x1, zero "same" padding) as an MFMA implicit GEMM for gfx950.
//
//   M = output channels, N = output pixels of one sample, K = Cin * ks * ks
//   v_mfma_f32_32x32x2_f32 (exact fp32: a k-ordered fmaf chain, same rate as the fp32 VALU peak
//   but one operand VGPR per lane and the VALU left free for staging / epilogue).
//
// Workgroup = 256 threads = 4 waves; output tile = 64 channels x (8 rows x 32 columns) of one
// sample.  Wave w owns rows 2w, 2w+1 of the tile for all 64 channels: 2 (channel tiles) x 2 (row
// segments) accumulators of 32x32 = 64 accumulator registers.
//   A operand (weights):  lane (i = l&31, h = l>>5) holds W[co = 32m+i][ci = 2p+h][tap]
//   B operand (input):    lane (j = l&31, h)        holds X[ci = 2p+h][row + ky][col j + kx]
// so both operands are conflict-free ds_read_b32 (32 consecutive dwords per half wave).
//
// K is walked in chunks of KC input channels.  Per chunk the input patch (with its 1-pixel halo,
// zero filled outside the image) and the pre-packed weight block are staged global -> registers
// -> LDS one chunk ahead of the MFMA loop (two LDS buffers, one barrier per chunk).  The staging
// read applies the fused producer op: 2x2 max-pool (DownSampler) or nearest 2x up-sampling
// (UpSampler); the epilogue applies bias, the per-(sample,channel) time shift and up to two
// residual tensors, so none of these is a separate pass over HBM.
#include "ds_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TH = 8;     // tile rows
constexpr int TW = 32;    // tile columns (= MFMA N)
constexpr int COT = 64;   // output channels per workgroup
constexpr int NT = 256;   // threads

template <int KS> struct Geo {
  static constexpr int KC = (KS == 3) ? 8 : 32;       // input channels per chunk
  static constexpr int TAPS = KS * KS;
  static constexpr int PH = TH + KS - 1;              // patch rows
  static constexpr int PW = TW + KS - 1;              // patch columns
  static constexpr int NX = KC * PH * PW;             // staged input floats per chunk
  static constexpr int NXI = (NX + NT - 1) / NT;      // per thread
  static constexpr int NW = TAPS * KC * COT;          // staged weight floats per chunk
  static constexpr int NW4I = (NW / 4 + NT - 1) / NT; // float4 per thread
  static constexpr int STAGE_FLOATS = NX + NW;
};

struct ConvArgs {
  float* out;
  const float* in;
  const float* wp;
  const float* bias;
  const float* shift;
  const float* res1;
  const float* res2;
  int shift_stride;
  int B, Cin, Cout, H, W, Hin, Win;
  int tiles_x, tiles_y, n_cot, n_chunks;
};

// One prefetch group of MFMA operands: G channel pairs x (all taps).  Two such register sets are
// alternated so that the ds_reads of group g+1 are in flight while the MFMAs of group g issue.
template <int KS> struct Frag {
  static constexpr int G = (KS == 3) ? 1 : 4;            // channel pairs per group
  static constexpr int ROWS = KS + 1;                    // input rows feeding the wave's 2 output rows
  float a[G][KS * KS][2];
  float b[G][ROWS][KS];
};

template <int KS, int MODE>
__global__ __launch_bounds__(NT, 2) void k_conv(const ConvArgs a) {
  using G = Geo<KS>;
  using F = Frag<KS>;
  constexpr int KC = G::KC, PH = G::PH, PW = G::PW;
  constexpr int NG = (KC / 2) / F::G;                     // groups per chunk (4 for 3x3, 4 for 1x1)
  static_assert(NG % 2 == 0, "group loop is unrolled by two");
  extern __shared__ __attribute__((aligned(16))) float smem[];   // [2][NW | NX]
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

  // ---- per-thread staging plan (identical for every chunk): offsets are always in bounds,
  //      out-of-image / out-of-range elements are zeroed by a select, never by a branch ----
  int xoff[G::NXI];
  unsigned xvalid = 0;
#pragma unroll
  for (int i = 0; i < G::NXI; ++i) {
    const int e = tid + NT * i;
    const int c = e / (PH * PW);
    const int rem = e - c * (PH * PW);
    const int r = rem / PW;
    const int col = rem - r * PW;
    const int gy = y0 + r - (KS / 2), gx = x0 + col - (KS / 2);
    const bool ok = (e < G::NX) && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
    int off;
    if (MODE == DS_LOAD_PLAIN) off = gy * a.Win + gx;
    else if (MODE == DS_LOAD_MAXPOOL2) off = (2 * gy) * a.Win + 2 * gx;
    else off = (gy >> 1) * a.Win + (gx >> 1);
    xoff[i] = ok ? (c * HWin + off) : 0;
    if (ok) xvalid |= (1u << i);
  }
  const float* in_b = a.in + (size_t)b * a.Cin * HWin;
  const f32x4* wp4 = reinterpret_cast<const f32x4*>(a.wp) + (size_t)cot * a.n_chunks * (G::NW / 4);

  float xr[G::NXI];
  f32x4 wr[G::NW4I];

  // stage_load only ISSUES the global loads (plus the max of the pooled mode); zeroing of the
  // halo / out-of-range elements happens in stage_store, after the MFMA block, so that no
  // s_waitcnt vmcnt lands in front of the matrix instructions.
  unsigned xok = 0;
  auto stage_load = [&](int chunk) {
    const int cbase = chunk * KC;
    const float* src = in_b + (size_t)cbase * HWin;
    const bool full = cbase + KC <= a.Cin;                 // uniform: only the last chunk can be partial
    xok = xvalid;
#pragma unroll
    for (int i = 0; i < G::NXI; ++i) {
      const int e = tid + NT * i;
      const int c = e / (PH * PW);
      bool ok = (xvalid >> i) & 1u;
      if (!full && !(cbase + c < a.Cin)) {
        ok = false;
        xok &= ~(1u << i);
      }
      const float* p = src + (ok ? xoff[i] : 0);
      if (MODE == DS_LOAD_MAXPOOL2) {
        const float2 t0 = *reinterpret_cast<const float2*>(p);
        const float2 t1 = *reinterpret_cast<const float2*>(p + a.Win);
        xr[i] = fmaxf(fmaxf(t0.x, t0.y), fmaxf(t1.x, t1.y));
      } else {
        xr[i] = *p;
      }
    }
    const f32x4* wsrc = wp4 + (size_t)chunk * (G::NW / 4);
#pragma unroll
    for (int i = 0; i < G::NW4I; ++i) {
      const int e = tid + NT * i;
      wr[i] = wsrc[e < G::NW / 4 ? e : G::NW / 4 - 1];     // tail lanes re-read the last vector (never stored)
    }
  };
  auto stage_store = [&](int buf) {
    float* ws = smem + buf * G::STAGE_FLOATS;
    float* xs = ws + G::NW;
#pragma unroll
    for (int i = 0; i < G::NXI; ++i) {
      const int e = tid + NT * i;
      if (NT * (i + 1) <= G::NX || e < G::NX) xs[e] = ((xok >> i) & 1u) ? xr[i] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < G::NW4I; ++i) {
      const int e = tid + NT * i;
      if (NT * (i + 1) <= G::NW / 4 || e < G::NW / 4) reinterpret_cast<f32x4*>(ws)[e] = wr[i];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[m][r][q] = 0.f;

  auto frag_load = [&](F& f, const float* wl, const float* xl, int g) {
#pragma unroll
    for (int pp = 0; pp < F::G; ++pp) {
      const int p = g * F::G + pp;
#pragma unroll
      for (int tap = 0; tap < KS * KS; ++tap) {
        f.a[pp][tap][0] = wl[(tap * KC + 2 * p) * COT];
        f.a[pp][tap][1] = wl[(tap * KC + 2 * p) * COT + 32];
      }
#pragma unroll
      for (int row = 0; row < F::ROWS; ++row)
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) f.b[pp][row][kx] = xl[2 * p * (PH * PW) + row * PW + kx];
    }
  };
  auto frag_mma = [&](const F& f) {
#pragma unroll
    for (int pp = 0; pp < F::G; ++pp)
#pragma unroll
      for (int ky = 0; ky < KS; ++ky)
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) {
          const int tap = ky * KS + kx;
          acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[pp][tap][0], f.b[pp][0 + ky][kx], acc[0][0], 0, 0, 0);
          acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[pp][tap][0], f.b[pp][1 + ky][kx], acc[0][1], 0, 0, 0);
          acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[pp][tap][1], f.b[pp][0 + ky][kx], acc[1][0], 0, 0, 0);
          acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[pp][tap][1], f.b[pp][1 + ky][kx], acc[1][1], 0, 0, 0);
        }
  };

  stage_load(0);
  stage_store(0);
  __syncthreads();

  F f0, f1;
  for (int chunk = 0; chunk < a.n_chunks; ++chunk) {
    const int buf = chunk & 1;
    const bool more = chunk + 1 < a.n_chunks;
    const float* ws = smem + buf * G::STAGE_FLOATS;
    const float* xs = ws + G::NW;
    const float* wl = ws + lh * COT + li;                       // + (tap*KC + 2p)*COT + 32m
    const float* xl = xs + lh * (PH * PW) + (2 * wv) * PW + li; // + 2p*PH*PW + row*PW + kx
    frag_load(f0, wl, xl, 0);
    if (more) stage_load(chunk + 1);                            // global loads fly under the MFMAs
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < NG; g += 2) {
      // the scheduling fences keep each operand prefetch ahead of the previous group's MFMAs
      frag_load(f1, wl, xl, g + 1);
      __builtin_amdgcn_sched_barrier(0);
      frag_mma(f0);
      __builtin_amdgcn_sched_barrier(0);
      if (g + 2 < NG) frag_load(f0, wl, xl, g + 2);
      __builtin_amdgcn_sched_barrier(0);
      frag_mma(f1);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (more) stage_store(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: bias, time shift, residuals.  All loads of a 16-register group are issued
  //      before any is consumed; invalid lanes read a safe address and are masked at the store.
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

// torch [Cout][Cin][ks][ks] -> [cot][chunk][tap][kc][64], zero padded
__global__ void k_pack(float* packed, const float* __restrict__ w, int Cout, int Cin, int ks, int KC, int n_chunks,
                       size_t total) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int taps = ks * ks;
  size_t t = i;
  const int co64 = t % COT; t /= COT;
  const int kc = t % KC; t /= KC;
  const int tap = t % taps; t /= taps;
  const int chunk = t % n_chunks; t /= n_chunks;
  const int cot = (int)t;
  const int co = cot * COT + co64, ci = chunk * KC + kc;
  float v = 0.f;
  if (co < Cout && ci < Cin) v = w[((size_t)co * Cin + ci) * taps + tap];
  packed[i] = v;
}

inline int kc_for(int ks) { return ks == 3 ? Geo<3>::KC : Geo<1>::KC; }

template <int KS, int MODE>
int launch_conv(const ConvArgs& a, hipStream_t s) {
  using G = Geo<KS>;
  const size_t lds = 2 * G::STAGE_FLOATS * sizeof(float);
  {
    const int rc = ds::ensure_dynamic_lds<&k_conv<KS, MODE>>((int)lds, "hipFuncSetAttribute(conv)");
    if (rc != DS_OK) return rc;
  }
  const long long blocks = (long long)a.B * a.tiles_y * a.tiles_x * a.n_cot;
  DS_REQUIRE(blocks > 0 && blocks < (1ll << 31), DS_ERR_SHAPE, "ds_conv2d: grid of %lld workgroups is out of range", blocks);
  hipLaunchKernelGGL((k_conv<KS, MODE>), dim3((unsigned)blocks), dim3(NT), lds, s, a);
  DS_CHECK_LAUNCH("ds_conv2d");
  return DS_OK;
}

}  // namespace

extern "C" {

size_t ds_conv2d_packed_floats(int Cout, int Cin, int ks) {
  if (Cout <= 0 || Cin <= 0 || (ks != 1 && ks != 3)) return 0;
  const int KC = kc_for(ks);
  const size_t n_cot = (Cout + COT - 1) / COT, n_chunks = (Cin + KC - 1) / KC;
  return n_cot * n_chunks * (size_t)(ks * ks) * KC * COT;
}

int ds_conv2d_pack_weights(float* packed, const float* w, int Cout, int Cin, int ks, void* stream) {
  DS_REQUIRE(packed && w, DS_ERR_NULL, "ds_conv2d_pack_weights: NULL pointer");
  DS_REQUIRE(Cout > 0 && Cin > 0 && (ks == 1 || ks == 3), DS_ERR_SHAPE,
             "ds_conv2d_pack_weights: Cout=%d Cin=%d ks=%d unsupported", Cout, Cin, ks);
  const int KC = kc_for(ks);
  const int n_chunks = (Cin + KC - 1) / KC;
  const size_t total = ds_conv2d_packed_floats(Cout, Cin, ks);
  hipLaunchKernelGGL(k_pack, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ds::as_stream(stream), packed, w,
                     Cout, Cin, ks, KC, n_chunks, total);
  DS_CHECK_LAUNCH("ds_conv2d_pack_weights");
  return DS_OK;
}

int ds_conv2d(float* out, const float* in, const float* w_packed, const float* bias, const float* shift,
              int shift_stride, const float* res1, const float* res2, int B, int Cin, int Cout, int H, int W, int ks,
              int load_mode, void* stream) {
  DS_REQUIRE(out && in && w_packed, DS_ERR_NULL, "ds_conv2d: NULL pointer");
  DS_REQUIRE(B >= 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, DS_ERR_SHAPE,
             "ds_conv2d: bad shape B=%d Cin=%d Cout=%d H=%d W=%d", B, Cin, Cout, H, W);
  DS_REQUIRE(ks == 1 || ks == 3, DS_ERR_UNSUPPORTED, "ds_conv2d: kernel size %d (only 1 and 3)", ks);
  DS_REQUIRE(load_mode >= 0 && load_mode <= 2, DS_ERR_UNSUPPORTED, "ds_conv2d: load_mode %d", load_mode);
  DS_REQUIRE(load_mode != DS_LOAD_UPSAMPLE2 || (H % 2 == 0 && W % 2 == 0), DS_ERR_SHAPE,
             "ds_conv2d: UPSAMPLE2 needs even output H, W (got %d x %d)", H, W);
  DS_REQUIRE(shift == nullptr || shift_stride == 0 || shift_stride >= Cout, DS_ERR_SHAPE,
             "ds_conv2d: shift_stride %d < Cout %d", shift_stride, Cout);
  DS_REQUIRE((reinterpret_cast<uintptr_t>(w_packed) & 15u) == 0, DS_ERR_SHAPE, "ds_conv2d: w_packed must be 16-byte aligned");
  DS_REQUIRE(load_mode != DS_LOAD_MAXPOOL2 || (reinterpret_cast<uintptr_t>(in) & 7u) == 0, DS_ERR_SHAPE,
             "ds_conv2d: MAXPOOL2 input must be 8-byte aligned");
  if (B == 0) return DS_OK;
  ConvArgs a;
  a.out = out; a.in = in; a.wp = w_packed; a.bias = bias; a.shift = shift; a.res1 = res1; a.res2 = res2;
  a.shift_stride = shift_stride;
  a.B = B; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W;
  a.Hin = load_mode == DS_LOAD_MAXPOOL2 ? 2 * H : (load_mode == DS_LOAD_UPSAMPLE2 ? H / 2 : H);
  a.Win = load_mode == DS_LOAD_MAXPOOL2 ? 2 * W : (load_mode == DS_LOAD_UPSAMPLE2 ? W / 2 : W);
  DS_REQUIRE((long long)Cin * a.Hin * a.Win < (1ll << 31), DS_ERR_SHAPE, "ds_conv2d: per-sample input exceeds 2^31 floats");
  a.tiles_x = (W + TW - 1) / TW; a.tiles_y = (H + TH - 1) / TH;
  a.n_cot = (Cout + COT - 1) / COT;
  const int KC = kc_for(ks);
  a.n_chunks = (Cin + KC - 1) / KC;
  hipStream_t s = ds::as_stream(stream);
  if (ks == 3) {
    if (load_mode == DS_LOAD_PLAIN) return launch_conv<3, DS_LOAD_PLAIN>(a, s);
    if (load_mode == DS_LOAD_MAXPOOL2) return launch_conv<3, DS_LOAD_MAXPOOL2>(a, s);
    return launch_conv<3, DS_LOAD_UPSAMPLE2>(a, s);
  }
  if (load_mode == DS_LOAD_PLAIN) return launch_conv<1, DS_LOAD_PLAIN>(a, s);
  if (load_mode == DS_LOAD_MAXPOOL2) return launch_conv<1, DS_LOAD_MAXPOOL2>(a, s);
  return launch_conv<1, DS_LOAD_UPSAMPLE2>(a, s);
}

}  // extern "C"
