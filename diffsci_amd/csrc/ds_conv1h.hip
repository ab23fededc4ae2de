// 1x1 convolution (pointwise channel mixing) with fp32 accuracy on the fp16 matrix cores -- the
// fp16x3 split of ds_conv3h.hip (x = hi + lo in fp16, lo*hi + hi*lo + hi*hi, fp32 accumulation).
// Used for ADM's residual projections (adm.py:345-349, with the block's nearest-upsampling or
// average pooling folded into the load) and the attention in/out projections.
//
// A 1x1 convolution has 1/9 of the 3x3's FLOPs per byte: with C <= 256 it is HBM-bound
// (C/4 flop/byte), so this kernel is organised around the loads, not the MFMAs:
//   workgroup = 4 waves, tile = 64 channels x 8 rows x 32 columns, 2 workgroups per CU;
//   one step = 16 input channels = 12 MFMAs per wave; thread t owns pixel t of the tile and
//   fetches its 16 channel values two steps ahead (two register sets), splits them to fp16
//   pieces one step ahead into a 3-deep ring of LDS images; the chunk's 4 KiB weight slab rides
//   the same path (one 16-byte vector per thread).  The step barrier is a bare s_barrier behind
//   lgkmcnt(0) only, so the global loads of the next two steps stay in flight across it.
//   blockIdx -> tile is XCD-aware: the Cout/64 workgroups sharing an input tile run on one XCD.
#include "ds_common.h"
#include "ds_conv_epilogue.h"

namespace {

using ds_epi::f32x16;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(1))) float* gfloat_ptr;
typedef const __attribute__((address_space(1))) char* gchar_ptr;
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(1))) f32x2* gfloat2_ptr;

constexpr int COT = 64, NT = 256, KC = 16;
constexpr int NPOS = 256;                          // pixels per tile = NT: one pixel per thread
constexpr int XBUF_VEC = 2 * 2 * NPOS;             // [piece][h][pos] 16-byte vectors: 1024 (16 KiB)
constexpr int WSLAB_VEC = 2 * 2 * COT;             // [piece][h][co] per chunk: 256 (4 KiB)
constexpr int NXB = 3, NWS = 3;
constexpr int RING_BYTES = (NXB * XBUF_VEC + NWS * WSLAB_VEC) * 16;     // 61,440
constexpr int EPI_BYTES = 4 * 64 * 2 * 32 * 4;                          // the epilogue's 4 x 16 KiB transposition tiles
constexpr int STAGE_BYTES = RING_BYTES > EPI_BYTES ? RING_BYTES : EPI_BYTES;
constexpr int LDS_BYTES = STAGE_BYTES + 128 * 4;
// TWO (round 3): the workgroup owns two 64-channel tiles of the same pixel tile -- eight waves, waves 0-3 the first tile, 4-7 the
// second; the chunk's 16 input channels are fetched and split ONCE for both (thread t: pixel t & 255, channel half t >> 8), so the
// split's vector work per output halves: with more than two output tiles per pixel tile the one-tile kernel is bound by it
// (200-235 TF/s-equivalent whatever the shape).  Ring: three X images + three pairs of slabs (73,728 B); the epilogue's eight
// 16 KiB wave tiles (131,072 B) reach past it; one workgroup per CU.
constexpr int TWO_RING_BYTES = (NXB * XBUF_VEC + NWS * 2 * WSLAB_VEC) * 16;
constexpr int TWO_EPI_BYTES = 8 * 64 * 2 * 32 * 4;
static_assert(TWO_EPI_BYTES >= TWO_RING_BYTES, "bias / shift rows behind the larger of the two");
constexpr int LDS_BYTES_TWO = TWO_EPI_BYTES + 2 * 128 * 4;
constexpr int NXCD = 8;

struct Conv1hArgs {
  float* out;
  const float* in;
  const u32x4* wp;
  const float* bias;
  const float* shift;
  const float* res1;
  const float* res2;
  float* tile_stats;      // see ds_conv_epilogue.h, or NULL
  int res1_up;
  const unsigned* in_amax;   // per-sample max |input| (float bits) -> activation exponent (ds_conv_epilogue.h), or NULL
  unsigned* out_amax;        // per-sample max |output| slots, or NULL
  int wshift;
  int amax_split;            // channels >= amax_split (> 0) report to out_amax[B + b]
  int shift_stride;
  int B, Cin, Cout, H, W, Hin, Win;
  int tiles_x, tiles_y, n_cot, n_chunks;
  int n_cot_wg;           // workgroups per pixel tile: n_cot, or n_cot / 2 with two channel tiles per workgroup
  unsigned n_blocks;
};

// see ds_conv3h.hip: the remainder must be taken against the stored (packed-converted) high piece
__device__ __forceinline__ void split2(float a, float b, unsigned& hi, unsigned& lo) {
  f16x2 h = {(_Float16)a, (_Float16)b};
  unsigned hp = __builtin_bit_cast(unsigned, h);
  asm volatile("" : "+v"(hp));                       // opaque: both uses below see these exact bits
  const f16x2 hq = __builtin_bit_cast(f16x2, hp);
  f16x2 l = {(_Float16)(a - (float)hq[0]), (_Float16)(b - (float)hq[1])};
  hi = hp;
  lo = __builtin_bit_cast(unsigned, l);
}

__device__ __forceinline__ void step_barrier() {
  // LDS writes of this wave are done (lgkmcnt); outstanding global loads are NOT waited for
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// W16: the 256 pixels of the tile are 16 rows x 16 columns (narrow feature maps) instead of 8 x 32.
template <int MODE, bool W16, bool TWO = false>
__global__ __launch_bounds__(TWO ? 2 * NT : NT, 2) void k_conv1h(const Conv1hArgs a) {
  constexpr int TW = W16 ? 16 : 32, TH = W16 ? 16 : 8;
  constexpr int COTS = TWO ? 2 : 1;
  constexpr int KR = TWO ? KC / 2 : KC;                              // channel values a thread fetches per chunk
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32x4* Xs = reinterpret_cast<u32x4*>(smem);                        // [buf 3][piece][h][pos]
  u32x4* Ws = Xs + NXB * XBUF_VEC;                                   // [slot 3][channel tile][piece][h][co]
  float* BS = reinterpret_cast<float*>(smem + (TWO ? TWO_EPI_BYTES : STAGE_BYTES));      // [channel tile][2][64] bias, shift
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cw = TWO ? wv >> 2 : 0;                                  // the wave's channel tile; the thread's channel half when staging
  const int wq = wv & 3;                                             // the wave's pixel rows
  const int pix = tid & (NT - 1);
  const int li = lane & 31, lh = lane >> 5;

  // XCD-aware order: hardware sends workgroup i to XCD i % 8; give each XCD a contiguous run of
  // logical tiles so the n_cot workgroups reading one input tile share an L2.
  unsigned bid;
  {
    const unsigned x = blockIdx.x % NXCD, k = blockIdx.x / NXCD;
    const unsigned per = a.n_blocks / NXCD, rem = a.n_blocks % NXCD;
    bid = x * per + (x < rem ? x : rem) + k;
  }
  const int cot = (int)(bid % a.n_cot_wg) * COTS; bid /= a.n_cot_wg;     // first channel tile of the workgroup
  const int tx = bid % a.tiles_x; bid /= a.tiles_x;
  const int ty = bid % a.tiles_y; bid /= a.tiles_y;
  const int b = bid;
  const int x0 = tx * TW, y0 = ty * TH;
  const int HWin = a.Hin * a.Win;
  const int n = a.n_chunks;

  // Pixel offset of this thread inside a channel plane.  Out-of-range pixels of a ragged tile and
  // channels past Cin of a ragged last chunk are fetched from clamped (valid) addresses and NOT
  // zeroed: a 1x1 convolution has no halo, so an out-of-range pixel only feeds outputs that are
  // never stored, and the padded weight rows of the channels past Cin are zero.
  const int gy = y0 + (W16 ? pix >> 4 : pix >> 5), gx = x0 + (W16 ? pix & 15 : pix & 31);
  const bool ok = gy < a.H && gx < a.W;
  unsigned off;
  if (MODE == DS_LOAD_PLAIN) off = gy * a.Win + gx;
  else if (MODE == DS_LOAD_UPSAMPLE2) off = (gy >> 1) * a.Win + (gx >> 1);
  else off = (2 * gy) * a.Win + 2 * gx;                               // DS_LOAD_AVGPOOL2
  if (!ok) off = 0;
  const unsigned offb = off * 4u, rowb = (unsigned)a.Win * 4u;        // BYTE offsets: a plane is < 2^29 floats (host check)
  const float* in_b = a.in + (size_t)b * a.Cin * HWin;                // uniform: loads are s[base] + v offset
  const u32x4* wp = a.wp + (size_t)cot * n * WSLAB_VEC;
  ds_epi::ActScale ascale;

  auto x_fetch = [&](float (&R)[KR], int chunk) {
    const int cbase = chunk * KC + (TWO ? KR * cw : 0);
    unsigned ob = offb;
    asm volatile("" : "+v"(ob));       // keep the zero-extension next to the loads (scalar-base + 32-bit offset form)
#pragma unroll
    for (int c = 0; c < KR; ++c) {
      const int ch = cbase + c < a.Cin ? cbase + c : a.Cin - 1;        // scalar clamp
      // uniform base, pinned to an SGPR pair (opaque to the optimiser, which would otherwise fold the
      // lane offset into a hoisted 64-bit vector base and pay a 64-bit VALU add per load)
      gchar_ptr p = (gchar_ptr)(in_b + (size_t)ch * HWin);
      asm volatile("" : "+s"(p));
      if (MODE == DS_LOAD_AVGPOOL2) {
        const f32x2 t0 = *(gfloat2_ptr)(p + ob);
        const f32x2 t1 = *(gfloat2_ptr)(p + (ob + rowb));
        R[c] = (((t0.x + t0.y) + t1.x) + t1.y) / 4.0f;                 // torch avg_pool2d's summation order
      } else {
        R[c] = *(gfloat_ptr)(p + ob);
      }
    }
  };
  auto x_store = [&](const float (&R)[KR], int buf) {
    u32x4* xb = Xs + buf * XBUF_VEC;
#pragma unroll
    for (int hr = 0; hr < KR / 8; ++hr) {
      const int h = TWO ? cw : hr;                                     // TWO: the thread holds one channel half
      u32x4 qh, ql;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        unsigned ph, pl;
        split2(R[8 * hr + 2 * k] * ascale.in_scale, R[8 * hr + 2 * k + 1] * ascale.in_scale, ph, pl);   // exact (power of two)
        qh[k] = ph; ql[k] = pl;
      }
      xb[h * NPOS + pix] = qh;
      xb[(2 + h) * NPOS + pix] = ql;
    }
  };
  // the 4 KiB weight slab of a chunk is one 16-byte vector per thread; it rides along with the
  // thread's pixel through the same register prefetch (no LDS-DMA: the compiler orders every later
  // ds_read behind an LDS-DMA with a full vmcnt wait, which would serialise the prefetch)
  // (TWO: threads 256.. carry the second channel tile's slab, n slabs further on)
  auto w_fetch = [&](u32x4& Wr, int chunk) { Wr = wp[((size_t)cw * n + chunk) * WSLAB_VEC + pix]; };
  auto w_store = [&](const u32x4& Wr, int slot) { Ws[(slot * COTS + cw) * WSLAB_VEC + pix] = Wr; };

  f32x16 acc[2][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[m][r][q] = 0.f;

  auto mma = [&](int slot, int buf) {
    const u32x4* wb = Ws + (slot * COTS + cw) * WSLAB_VEC;
    const u32x4* xb = Xs + buf * XBUF_VEC;
    f16x8 fa[2][2], fb[2][2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
      for (int m = 0; m < 2; ++m) fa[p][m] = *reinterpret_cast<const f16x8*>(&wb[(p * 2 + lh) * COT + 32 * m + li]);
#pragma unroll
      for (int r = 0; r < 2; ++r) fb[p][r] = *reinterpret_cast<const f16x8*>(&xb[(p * 2 + lh) * NPOS + (2 * wq + r) * 32 + li]);
    }
    constexpr int PA[3] = {1, 0, 0};
    constexpr int PB[3] = {0, 1, 0};
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 2; ++r)
          acc[m][r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[PA[t]][m], fb[PB[t]][r], acc[m][r], 0, 0, 0);
  };

  float R0[KR], R1[KR];
  u32x4 W0, W1;
  const float bias_shift = ds_epi::fetch_bias_shift(a.bias, a.shift, a.shift_stride, b, (cot + (TWO ? (tid >> 7) & 1 : 0)) * COT, a.Cout, COTS);
  __builtin_amdgcn_sched_barrier(0);
  const unsigned amax_bits = ds_epi::act_bits(a.in_amax, b);   // every input of this kernel is a raw tensor; consumed behind the first loads
  w_fetch(W0, 0);
  x_fetch(R0, 0);
  w_fetch(W1, n > 1 ? 1 : 0);
  x_fetch(R1, n > 1 ? 1 : 0);
  __builtin_amdgcn_sched_barrier(0);
  ascale = ds_epi::act_scale_of(amax_bits, a.wshift);
  w_store(W0, 0);
  x_store(R0, 0);
  ds_epi::commit_bias_shift(BS, bias_shift, COTS);
  __syncthreads();

  // chunk k: fetched into R[k & 1] at step k-2, split into image k % 3 at step k-1, used at step k
  int buf = 0;
  auto step = [&](float (&Ra)[KR], u32x4& Wa, float (&Rb)[KR], u32x4& Wb, int g) {
    // unconditional (clamped to the last chunk: a redundant, never-consumed fetch in the last two
    // steps) so the number of loads in flight is static and the waits below stay partial
    const int gf = g + 2 < n ? g + 2 : n - 1;
    w_fetch(Wa, gf);
    x_fetch(Ra, gf);
    mma(g % NWS, buf);
    const int nb = buf == NXB - 1 ? 0 : buf + 1;
    w_store(Wb, (g + 1) % NWS);          // never-consumed slots in the last step
    x_store(Rb, nb);
    buf = nb;
    step_barrier();
  };
  int g = 0;
  for (; g + 1 < n; g += 2) {
    step(R0, W0, R1, W1, g);
    step(R1, W1, R0, W0, g + 1);
  }
  if (g < n) step(R0, W0, R1, W1, g);
  __syncthreads();       // full fence before the staging buffers are reused by the epilogue

  {
    ds_epi::Args e;
    e.out = a.out; e.bias = a.bias; e.shift = a.shift; e.res1 = a.res1; e.res2 = a.res2; e.res1_up = a.res1_up;
    e.unscale = ds_epi::unscale_from_in(ascale.in_scale, a.wshift); e.shift_stride = a.shift_stride;
    e.out_amax = a.out_amax ? a.out_amax + b + ((a.amax_split > 0 && (cot + cw) * COT >= a.amax_split) ? a.B : 0) : nullptr;
    e.b = b; e.co_base = (cot + cw) * COT; e.y0 = y0 + (W16 ? 4 : 2) * wq; e.x0 = x0;
    e.Cout = a.Cout; e.H = a.H; e.W = a.W;
    e.tile_stats = a.tile_stats; e.tile = ty * a.tiles_x + tx; e.ntiles = a.tiles_x * a.tiles_y;
    float* tile = reinterpret_cast<float*>(smem) + wv * (64 * 2 * 32);
    ds_epi::store_tile<W16>(acc, tile, BS + 128 * cw, e);
    if (a.tile_stats) {
      __syncthreads();
      if constexpr (TWO) {                      // threads 0-63 combine the first channel tile's four waves, 64-127 the second's
        if (tid < 128) {
          e.co_base = (cot + (tid >> 6)) * COT;
          ds_epi::store_tile_stats(reinterpret_cast<const float*>(smem) + (tid >> 6) * 4 * (64 * 2 * 32), 64 * 2 * 32, e, tid & 63);
        }
      } else {
        ds_epi::store_tile_stats(reinterpret_cast<const float*>(smem), 64 * 2 * 32, e);
      }
    }
  }
}

// torch [Cout][Cin][1][1] fp32 (times 2^wshift) -> [cot][chunk][piece][h][co 64][ci 8] fp16
__global__ void k_pack1h(_Float16* packed, const float* __restrict__ w, int Cout, int Cin, int n_chunks, float scale,
                         size_t total) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  size_t t = i;
  const int c8 = t % 8; t /= 8;
  const int co64 = t % COT; t /= COT;
  const int h = t % 2; t /= 2;
  const int piece = t % 2; t /= 2;
  const int chunk = t % n_chunks; t /= n_chunks;
  const int cot = (int)t;
  const int co = cot * COT + co64, ci = chunk * KC + 8 * h + c8;
  float v = 0.f;
  if (co < Cout && ci < Cin) v = w[(size_t)co * Cin + ci] * scale;
  const _Float16 hi = (_Float16)v;
  const _Float16 lo = (_Float16)(v - (float)hi);
  packed[i] = piece == 0 ? hi : lo;
}

// Two channel tiles per workgroup where the output has an even number of them and the halved grid still fills the chip
// (DS_CONV1_TWO_MIN workgroups, 256 = one per CU).  Measured on MI355X (tools/conv1h_time.py, batch 32): 512 -> 256 at 128^2
// 626 -> 565 us, 768 -> 384 at 64^2 331 -> 292, 1024 -> 512 at 32^2 142 -> 118, 384 -> 128 at 256^2 1070 -> 962, the attention
// in-projection 256 -> 768 at 32^2 (batch 64) 127.5 -> 114.  DS_CONV1_TWO=0 switches it off, =2 drops the grid condition (A/B runs).
inline int conv1h_two() {
  static const int v = [] { const char* e = getenv("DS_CONV1_TWO"); return e ? atoi(e) : 1; }();
  return v;
}
inline long long conv1h_two_min() {
  static const long long v = [] { const char* e = getenv("DS_CONV1_TWO_MIN"); return e ? atoll(e) : 256ll; }();
  return v;
}

template <int MODE, bool W16>
int launch_conv1h(Conv1hArgs a, hipStream_t s) {
  const int two = conv1h_two();
  if (a.n_cot % 2 == 0 && ((two == 1 && (long long)a.n_blocks / 2 >= conv1h_two_min()) || two == 2)) {
    const int rc = ds::ensure_dynamic_lds<&k_conv1h<MODE, W16, true>>((int)(LDS_BYTES_TWO), "hipFuncSetAttribute(conv1h two)");
    if (rc != DS_OK) return rc;
    a.n_cot_wg = a.n_cot / 2;
    a.n_blocks /= 2;
    hipLaunchKernelGGL((k_conv1h<MODE, W16, true>), dim3(a.n_blocks), dim3(2 * NT), LDS_BYTES_TWO, s, a);
    DS_CHECK_LAUNCH("ds_conv1x1_h3");
    return DS_OK;
  }
  {
    const int rc = ds::ensure_dynamic_lds<&k_conv1h<MODE, W16>>((int)(LDS_BYTES), "hipFuncSetAttribute(conv1h)");
    if (rc != DS_OK) return rc;
  }
  a.n_cot_wg = a.n_cot;
  hipLaunchKernelGGL((k_conv1h<MODE, W16>), dim3(a.n_blocks), dim3(NT), LDS_BYTES, s, a);
  DS_CHECK_LAUNCH("ds_conv1x1_h3");
  return DS_OK;
}

}  // namespace

extern "C" {

size_t ds_conv1x1_h3_packed_bytes(int Cout, int Cin) {
  if (Cout <= 0 || Cin <= 0) return 0;
  const size_t n_cot = (Cout + COT - 1) / COT, n_chunks = (Cin + KC - 1) / KC;
  return n_cot * n_chunks * (size_t)WSLAB_VEC * 16;
}

int ds_conv1x1_h3_pack_weights(void* packed, const float* w, int Cout, int Cin, int wshift, void* stream) {
  DS_REQUIRE(packed && w, DS_ERR_NULL, "ds_conv1x1_h3_pack_weights: NULL pointer");
  DS_REQUIRE(Cout > 0 && Cin > 0, DS_ERR_SHAPE, "ds_conv1x1_h3_pack_weights: Cout=%d Cin=%d", Cout, Cin);
  DS_REQUIRE(wshift >= -40 && wshift <= 40, DS_ERR_SHAPE, "ds_conv1x1_h3_pack_weights: wshift %d out of range", wshift);
  const int n_chunks = (Cin + KC - 1) / KC;
  const size_t total = ds_conv1x1_h3_packed_bytes(Cout, Cin) / 2;
  hipLaunchKernelGGL(k_pack1h, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ds::as_stream(stream),
                     reinterpret_cast<_Float16*>(packed), w, Cout, Cin, n_chunks, ldexpf(1.0f, wshift), total);
  DS_CHECK_LAUNCH("ds_conv1x1_h3_pack_weights");
  return DS_OK;
}

int ds_conv1x1_h3(float* out, const float* in, const void* w_packed, int wshift, const float* bias, const float* shift,
                  int shift_stride, const float* res1, const float* res2, int B, int Cin, int Cout, int H, int W,
                  int load_mode, float* tile_stats, const unsigned* in_amax, unsigned* out_amax, int amax_split, void* stream) {
  DS_REQUIRE(out && in && w_packed, DS_ERR_NULL, "ds_conv1x1_h3: NULL pointer");
  DS_REQUIRE(B >= 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, DS_ERR_SHAPE,
             "ds_conv1x1_h3: bad shape B=%d Cin=%d Cout=%d H=%d W=%d", B, Cin, Cout, H, W);
  DS_REQUIRE(load_mode == DS_LOAD_PLAIN || load_mode == DS_LOAD_UPSAMPLE2 || load_mode == DS_LOAD_AVGPOOL2,
             DS_ERR_UNSUPPORTED, "ds_conv1x1_h3: load_mode %d", load_mode);
  DS_REQUIRE(load_mode != DS_LOAD_UPSAMPLE2 || (H % 2 == 0 && W % 2 == 0), DS_ERR_SHAPE,
             "ds_conv1x1_h3: UPSAMPLE2 needs even output H, W (got %d x %d)", H, W);
  DS_REQUIRE(shift == nullptr || shift_stride == 0 || shift_stride >= Cout, DS_ERR_SHAPE,
             "ds_conv1x1_h3: shift_stride %d < Cout %d", shift_stride, Cout);
  DS_REQUIRE((reinterpret_cast<uintptr_t>(w_packed) & 15u) == 0, DS_ERR_SHAPE, "ds_conv1x1_h3: w_packed must be 16-byte aligned");
  DS_REQUIRE(load_mode != DS_LOAD_AVGPOOL2 || (reinterpret_cast<uintptr_t>(in) & 7u) == 0, DS_ERR_SHAPE,
             "ds_conv1x1_h3: AVGPOOL2 input must be 8-byte aligned");
  DS_REQUIRE(wshift >= -40 && wshift <= 40, DS_ERR_SHAPE, "ds_conv1x1_h3: wshift %d out of range", wshift);
  DS_REQUIRE(amax_split >= 0 && amax_split % COT == 0, DS_ERR_SHAPE, "ds_conv1x1_h3: amax_split %d must be a multiple of 64", amax_split);
  if (B == 0) return DS_OK;
  Conv1hArgs a;
  a.out = out; a.in = in; a.wp = reinterpret_cast<const u32x4*>(w_packed); a.bias = bias; a.shift = shift;
  a.res1 = res1; a.res2 = res2; a.shift_stride = shift_stride; a.tile_stats = tile_stats; a.res1_up = 0;
  a.wshift = wshift; a.in_amax = in_amax; a.out_amax = out_amax; a.amax_split = amax_split;
  a.B = B; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W;
  a.Hin = load_mode == DS_LOAD_AVGPOOL2 ? 2 * H : (load_mode == DS_LOAD_UPSAMPLE2 ? H / 2 : H);
  a.Win = load_mode == DS_LOAD_AVGPOOL2 ? 2 * W : (load_mode == DS_LOAD_UPSAMPLE2 ? W / 2 : W);
  DS_REQUIRE((long long)a.Hin * a.Win < (1ll << 29), DS_ERR_SHAPE, "ds_conv1x1_h3: a channel plane exceeds 2^29 floats");
  const long long pad32 = (long long)((W + 31) / 32 * 32) * ((H + 7) / 8 * 8);
  const long long pad16 = (long long)((W + 15) / 16 * 16) * ((H + 15) / 16 * 16);
  const bool w16 = pad16 < pad32;                    // the tile shape that pads the image with fewer pixels
  const int TW = w16 ? 16 : 32, TH = w16 ? 16 : 8;
  a.tiles_x = (W + TW - 1) / TW; a.tiles_y = (H + TH - 1) / TH;
  a.n_cot = (Cout + COT - 1) / COT;
  a.n_chunks = (Cin + KC - 1) / KC;
  const long long blocks = (long long)B * a.tiles_y * a.tiles_x * a.n_cot;
  DS_REQUIRE(blocks > 0 && blocks < (1ll << 31), DS_ERR_SHAPE, "ds_conv1x1_h3: grid of %lld workgroups is out of range", blocks);
  a.n_blocks = (unsigned)blocks;
  hipStream_t s = ds::as_stream(stream);
  if (w16) {
    if (load_mode == DS_LOAD_PLAIN) return launch_conv1h<DS_LOAD_PLAIN, true>(a, s);
    if (load_mode == DS_LOAD_UPSAMPLE2) return launch_conv1h<DS_LOAD_UPSAMPLE2, true>(a, s);
    return launch_conv1h<DS_LOAD_AVGPOOL2, true>(a, s);
  }
  if (load_mode == DS_LOAD_PLAIN) return launch_conv1h<DS_LOAD_PLAIN, false>(a, s);
  if (load_mode == DS_LOAD_UPSAMPLE2) return launch_conv1h<DS_LOAD_UPSAMPLE2, false>(a, s);
  return launch_conv1h<DS_LOAD_AVGPOOL2, false>(a, s);
}

}  // extern "C"
