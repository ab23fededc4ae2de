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

template <int KS, int MODE>
__global__ __launch_bounds__(NT, 2) void k_conv(const ConvArgs a) {
  using G = Geo<KS>;
  constexpr int KC = G::KC, TAPS = G::TAPS, PH = G::PH, PW = G::PW;
  extern __shared__ __attribute__((aligned(16))) float smem[];   // [2][NW | NX]
  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;

  int bid = blockIdx.x;
  const int cot = bid % a.n_cot; bid /= a.n_cot;
  const int tx = bid % a.tiles_x; bid /= a.tiles_x;
  const int ty = bid % a.tiles_y; bid /= a.tiles_y;
  const int b = bid;
  const int x0 = tx * TW, y0 = ty * TH;
  const int HWin = a.Hin * a.Win;

  // ---- per-thread staging plan (identical for every chunk) ----
  int xoff[G::NXI];
  unsigned xvalid = 0;    // bit i: position inside the image
  unsigned xchan = 0;     // packed channel-in-chunk of element i is recomputed (cheap) below
  (void)xchan;
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
    xoff[i] = ok ? (c * HWin + off) : -1;
    if (ok) xvalid |= (1u << i);
  }
  const float* in_b = a.in + (size_t)b * a.Cin * HWin;
  const float4* wp4 = reinterpret_cast<const float4*>(a.wp) + (size_t)cot * a.n_chunks * (G::NW / 4);

  float xr[G::NXI];
  float4 wr[G::NW4I];

  auto stage_load = [&](int chunk) {
    const int cbase = chunk * KC;
    const float* src = in_b + (size_t)cbase * HWin;
#pragma unroll
    for (int i = 0; i < G::NXI; ++i) {
      const int e = tid + NT * i;
      const int c = e / (PH * PW);
      float v = 0.f;
      if (((xvalid >> i) & 1u) && (cbase + c < a.Cin)) {
        const float* p = src + xoff[i];
        if (MODE == DS_LOAD_MAXPOOL2) {
          const float2 t0 = *reinterpret_cast<const float2*>(p);
          const float2 t1 = *reinterpret_cast<const float2*>(p + a.Win);
          v = fmaxf(fmaxf(t0.x, t0.y), fmaxf(t1.x, t1.y));
        } else {
          v = *p;
        }
      }
      xr[i] = v;
    }
    const float4* wsrc = wp4 + (size_t)chunk * (G::NW / 4);
#pragma unroll
    for (int i = 0; i < G::NW4I; ++i) {
      const int e = tid + NT * i;
      if (e < G::NW / 4) wr[i] = wsrc[e];
    }
  };
  auto stage_store = [&](int buf) {
    float* ws = smem + buf * G::STAGE_FLOATS;
    float* xs = ws + G::NW;
#pragma unroll
    for (int i = 0; i < G::NXI; ++i) {
      const int e = tid + NT * i;
      if (e < G::NX) xs[e] = xr[i];
    }
#pragma unroll
    for (int i = 0; i < G::NW4I; ++i) {
      const int e = tid + NT * i;
      if (e < G::NW / 4) reinterpret_cast<float4*>(ws)[e] = wr[i];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[m][r][q] = 0.f;

  stage_load(0);
  stage_store(0);
  __syncthreads();

  for (int chunk = 0; chunk < a.n_chunks; ++chunk) {
    const int buf = chunk & 1;
    const bool more = chunk + 1 < a.n_chunks;
    if (more) stage_load(chunk + 1);

    const float* ws = smem + buf * G::STAGE_FLOATS;
    const float* xs = ws + G::NW;
    // lane-resolved bases
    const float* wl = ws + lh * COT + li;                       // + (tap*KC + 2p)*COT + 32m
    const float* xl = xs + lh * (PH * PW) + (2 * wv) * PW + li; // + 2p*PH*PW + (rt+ky)*PW + kx
#pragma unroll
    for (int p = 0; p < KC / 2; ++p) {
#pragma unroll
      for (int ky = 0; ky < KS; ++ky) {
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) {
          const int tap = ky * KS + kx;
          const float a0 = wl[(tap * KC + 2 * p) * COT];
          const float a1 = wl[(tap * KC + 2 * p) * COT + 32];
          const float b0 = xl[2 * p * (PH * PW) + (0 + ky) * PW + kx];
          const float b1 = xl[2 * p * (PH * PW) + (1 + ky) * PW + kx];
          acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
          acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
          acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
          acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
      }
    }
    if (more) stage_store(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: bias, time shift, residuals; 128-byte row segments per half wave ----
  const int gx = x0 + li;
  const size_t plane = (size_t)a.H * a.W;
#pragma unroll
  for (int m = 0; m < 2; ++m) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int co = cot * COT + 32 * m + (q & 3) + 8 * (q >> 2) + 4 * lh;
      if (co >= a.Cout) continue;
      float add = a.bias ? a.bias[co] : 0.f;
      const bool has_bias = a.bias != nullptr;
      float sh = 0.f;
      if (a.shift) sh = a.shift[(size_t)b * a.shift_stride + co];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int gy = y0 + 2 * wv + r;
        if (gy < a.H && gx < a.W) {
          const size_t idx = ((size_t)b * a.Cout + co) * plane + (size_t)gy * a.W + gx;
          float v = acc[m][r][q];
          if (has_bias) v = v + add;
          if (a.shift) v = v + sh;
          if (a.res1) v = v + a.res1[idx];
          if (a.res2) v = v + a.res2[idx];
          a.out[idx] = v;
        }
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
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv<KS, MODE>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return ds::hip_fail(e, "hipFuncSetAttribute(conv)");
    attr_set = true;
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
