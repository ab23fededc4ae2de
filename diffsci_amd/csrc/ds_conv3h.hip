// 3x3 convolution with fp32 accuracy on the fp16 matrix cores ("fp16x3" split MFMA).
//
// Each fp32 operand is split into two fp16 pieces, x = hi + lo with hi = fp16(x) (round to
// nearest) and lo = fp16(x - hi) (the subtraction is exact), which keeps 22-23 significand bits;
// a*b is accumulated as  lo_a*hi_b + hi_a*lo_b + hi_a*hi_b  (the dropped lo*lo term is 2^-22
// relative).  fp16 x fp16 products are exact in fp32 and v_mfma_f32_32x32x16_f16 accumulates in
// fp32 and -- verified on gfx950 (tools/f16_denorm_test.hip) -- honours fp16 subnormal inputs,
// so small operands degrade to an ABSOLUTE error of 2^-25 instead of being flushed.  Weights are
// pre-multiplied by a per-layer power of two (undone exactly in the epilogue) so that their low
// pieces are normal numbers.  Measured representation error of a 64->64 3x3 convolution: 8.5e-8
// relative (torch's own fp32 convolution: 2.2e-7 from accumulation order alone).  Cost: 3 MFMAs
// per k-slab instead of 6 (bf16x6) or 8x8 (exact-fp32 MFMA).  Domain: the whole fp32 range of
// magnitudes -- every launch scales each SAMPLE by a power of two that puts its max |x| at 2^13
// (in_amax, or the fourth column of the prenorm table) and undoes it exactly in the epilogue
// (ds_conv_epilogue.h, ActScale); only an input whose channels lie more than 2^14 apart within one
// sample AND meet compensating weights is escalated to the exact-fp32 kernel (nets/precision.py).
//
// Pipeline (workgroup = 4 waves, tile = 64 channels x 8 rows x 32 columns, 2 workgroups per CU):
//   K walks in steps (chunk of 16 input channels, ky).  LDS holds
//     X  [2 buffers][piece 2][h 2][10*34 positions][8 ci] fp16      2 x 21,760 B
//     W  [3-slot ring][piece 2][kx 3][h 2][64 co][8 ci] fp16        3 x 12,288 B      (80,384 B)
//   * weight slab g+2 is fetched by LDS-DMA at the start of step g: it has landed and been
//     published by a barrier one whole step before it is used, so the first operands of every
//     step are prefetched during the previous step's last MFMA block;
//   * the next chunk's input patch is loaded to registers at ky = 0, split and written to the
//     OTHER X buffer at ky = 1 (published by that step's barrier), prefetchable at ky = 2;
//   * operand fragments are double-buffered per kx block, their ds_read_b128 slotted between the
//     MFMAs of the previous block.  One barrier per step, nothing waits on it.
#include "ds_common.h"
#include <cstring>
#include "ds_conv3h_args.h"

#ifdef DS_STAMP
unsigned long long* g_stamps = nullptr;
#define STAMP_SLOTS 16
#define STAMP_AT(slot) a.stamps[((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * STAMP_SLOTS + (slot)]
#define STAMP(slot) do { if (threadIdx.x == 0) STAMP_AT(slot) = __builtin_amdgcn_s_memrealtime(); } while (0)
#define STAMP_CLK(slot) do { if (threadIdx.x == 0) STAMP_AT(slot) = __builtin_amdgcn_s_memtime(); } while (0)
// where the workgroup runs: HW_REG_HW_ID (id 4) and HW_REG_XCC_ID (id 20), whole 32-bit registers
#define STAMP_PLACE() do { if (threadIdx.x == 0) { STAMP_AT(8) = __builtin_amdgcn_s_getreg((31 << 11) | 4); STAMP_AT(9) = __builtin_amdgcn_s_getreg((31 << 11) | 20); } } while (0)
#else
#define STAMP(slot) do {} while (0)
#define STAMP_CLK(slot) do {} while (0)
#define STAMP_PLACE() do {} while (0)
#endif

namespace {

using namespace ds_conv3;

template <int MT> struct Frags { f16x8 a[2][MT]; f16x8 b[2][2]; };   // [piece][m] weights, [piece][r] input

// NW = 4: four waves, each owning the whole 64-channel tile for two of the eight pixel rows (2 x 2 accumulators),
//         two workgroups per CU (2 waves per SIMD);
// NW = 8: eight waves = the same four row pairs x two 32-channel halves (1 x 2 accumulators per wave, 6 MFMAs per
//         operand block instead of 12), two workgroups per CU = 4 waves per SIMD.  A single wave per SIMD cannot keep
//         the matrix pipe busy through its own loader / staging / barrier gaps (stamped: 43 % while its partner
//         workgroup is in its prologue or epilogue, 91 % when both are in the main loop); with four waves per SIMD
//         one of them almost always has MFMAs ready.  Costs: 1.5x the LDS operand reads per MFMA.
// S16 (NW = 4, an even number of 16-channel chunks): the same tiling on v_mfma_f32_16x16x32_f16 -- the chip holds a
//         higher clock on that shape (tools/mfma_shape_bench.hip: +8-11 % FLOP/s in a bare loop).  K = 32 of one
//         instruction = two TAPS x 16 channels: lane group g = l>>4 reads channels 8(g&1).. of tap g>>1, so the LDS images
//         and the weight packing are unchanged.  The 18 taps of two chunks pair up as
//           (0,0)+(0,1) | (0,2)+(1,0) | (1,1)+(1,2) | (2,0)+(2,1) | (2,2)+(0,0)' | (0,1)'+(0,2)' | (1,0)'+(1,1)' | (1,2)'+(2,0)' | (2,1)'+(2,2)'
//         (' = the next chunk, whose patch and first weight slab are already resident when its (0,0) tap is consumed).
// IMGIN: the input arrives as pre-split fp16 hi / lo images in the LDS layout (ds_inorm_silu_images: [b][chunk][piece][h][H+2][W+2]
// vectors of 8 channels, zero border) and is staged by LDS-DMA -- no staging registers, no split in this kernel.  Staging the
// patches from fp32 costs 16-29 % of the kernel (profiles/r02_stamps_nostage.log), and at four channel tiles every element is
// split 5.3 times; the producer splits it once.  16x16x32 variant, four waves, plain load, zero padding, no fused norm.
// TWO (round 3; S16, eight waves): the workgroup owns TWO 64-channel tiles of the same pixel tile -- waves 0-3 the first, 4-7 the
//         second, each wave a whole 64 x 64 tile as with NW = 4 -- and stages (fetches, activates, splits) the input patch ONCE for
//         both: at 128 output channels every element is otherwise loaded, normalised and split twice.  One
//         workgroup per CU (X buffers shared, weight ring doubled, eight 16 KiB epilogue tiles: 132 KB of LDS), 2 waves per SIMD
//         as with two four-wave workgroups.
// VEC (round 4; S16, plain load, 32-wide tiles, W a multiple of 32, whole chunks, no column tap offset): the patch's 32 interior columns are
//         fetched by 16-BYTE loads -- a staging unit is 2 channels x 4 consecutive pixels (two global_load_dwordx4, eight elements, eight
//         ds_write_b32 of a packed channel pair) instead of 8 channels x 1 pixel (eight global_load_dword) -- and only the two halo
//         columns keep the one-pixel items.  The texture-address path takes a wave's load instruction at the same rate whatever its
//         width, so a chunk's patch costs 6 instead of 24 load instructions per lane.  Measured (profiles/r04_vec4_loads.log; outputs,
//         tile statistics and maxima bit-identical to the one-pixel plan): -4.5 % per launch with two channel tiles (128 channels, 64 x 64),
//         +-0 on the one-tile four-wave kernel at 64 channels, +0.9 % end to end on config 2 (same-box A/B, three alternating pairs).
//         The unit's channel pair differs from lane to lane (q = lane & 3: that is what keeps the 4-byte LDS stores on distinct banks), so
//         the norm's table rows come from LDS: wave 0 parks the chunk's 16 rows in the X buffer's pad vectors a step ahead.
template <int MODE, bool W16, bool PRE, bool CIRC, int NW, bool S16 = false, bool IMGIN = false, bool TWO = false, bool VEC = false>
__global__ __launch_bounds__(64 * NW, TWO ? 2 : NW / 2) void k_conv3h(const Conv3hArgs a) {
  static_assert(!TWO || (S16 && NW == 8 && !IMGIN), "two channel tiles per workgroup: the eight-wave 16x16x32 kernel");
  static_assert(!VEC || (S16 && !IMGIN && !W16 && MODE == DS_LOAD_PLAIN && (TWO || NW == 4)), "16-byte patch loads: plain 32-wide 16x16x32 kernels");
  constexpr int COTS = TWO ? 2 : 1;
  constexpr int M16 = TWO ? 4 : 16 / NW;                             // S16: 16-channel tiles per wave (4, or 2 with eight waves on one tile)
  static_assert(!IMGIN || (S16 && NW == 4 && MODE == DS_LOAD_PLAIN && !PRE && !CIRC), "image input: plain 16x16x32 four-wave kernel");
  constexpr int TH = Geo<W16>::TH, TW = Geo<W16>::TW, PW = Geo<W16>::PW, NPOS = Geo<W16>::NPOS;
  constexpr int NTH = 64 * NW;                                       // threads
  constexpr int MT = TWO ? 2 : 8 / NW;                               // 32-channel tiles per wave: 2 or 1
  constexpr int XI = (Geo<W16>::XITEMS + NTH - 1) / NTH;             // staging items per thread: 3 or 2
  constexpr int HS = S16 ? NPOS + HPAD16 : NPOS;                     // vectors between the h = 0 and h = 1 images of a piece
  constexpr int PS = S16 ? 2 * NPOS + HPAD16 : 2 * NPOS;             // ... between the two pieces
  constexpr int XBV = S16 ? XBUF_VEC16 : XBUF_VEC;                   // ... between the two X buffers
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32x4* Xs = reinterpret_cast<u32x4*>(smem);                        // [buf][piece][h][pos]
  u32x4* Ws = Xs + 2 * XBV;                                          // [slot][channel tile][piece][kx][h][co]
  constexpr int WSV = COTS * WSLAB_VEC;                              // vectors per ring slot
  float* BS = reinterpret_cast<float*>(smem + (TWO ? TWO_EPI_BYTES : (S16 ? STAGE_BYTES16 : STAGE_BYTES)));   // [channel tile][2][64] bias, shift
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rw = wv & 3;                               // row group of the wave
  const int mh = (NW == 8 && !TWO) ? (wv >> 2) : 0;    // its 32-channel half (NW = 8 on one channel tile)
  const int cw = TWO ? (wv >> 2) : 0;                  // its channel tile (TWO)
  const int li = lane & 31, lh = lane >> 5;
  // position of (wave, r, lane) inside the halo patch, before the (ky, kx) tap offset
  const int lane_pos = W16 ? (li >> 4) * PW + (li & 15) : li;
  const int wave_row = W16 ? 4 * rw : 2 * rw;
  constexpr int ROWS_PER_R = W16 ? 2 : 1;

  // grid = (channel tiles, pixel tiles, samples): no runtime integer division in the prologue except one
  // multiply-high by the precomputed reciprocal of tiles_x (exact: tile * tiles_x < 2^32)
  // XCD-aware tile order.  The dispatcher deals workgroups round-robin over the 8 XCDs (linear id % 8), each with
  // its own L2; renumbering so that every XCD owns one contiguous run of (channel tile, pixel tile, sample) work
  // lets the Cout/64 workgroups that read the same input patch, and the neighbouring tiles that share its halo,
  // hit in one L2 instead of fetching the patch once per XCD.
  unsigned cot_u, tile_u, b_u;
  {
    const unsigned nx = gridDim.x, ny = gridDim.y;
    const unsigned id = blockIdx.x + nx * (blockIdx.y + ny * blockIdx.z);
    const unsigned total = nx * ny * gridDim.z;
    const unsigned per = total >> 3, rem = total & 7u;
    const unsigned xcd = id & 7u, k = id >> 3;
    const unsigned logical = xcd * per + (xcd < rem ? xcd : rem) + k;
    cot_u = logical % nx;
    const unsigned rest = logical / nx;
    tile_u = rest % ny;
    b_u = rest / ny;
  }
  STAMP_PLACE();
#ifdef DS_STAMP
  if (a.stagger_ticks) {
    // Experiment (profiles/r02_stamps_stagger.log: no effect): a phase offset between the two workgroups that share a CU.
    const unsigned id = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    if (id >= a.stagger_lo && id < a.stagger_hi) {
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      while (__builtin_amdgcn_s_memrealtime() - t0 < a.stagger_ticks) __builtin_amdgcn_s_sleep(8);
    }
  }
#endif
  const int cot = (int)cot_u * COTS;                   // first channel tile of the workgroup
  const int tile_id = (int)tile_u;
  const int b = (int)b_u;
  const int ty = a.tiles_x == 1 ? tile_id : (int)__umulhi((unsigned)tile_id, a.tiles_x_magic);
  const int tx = tile_id - ty * a.tiles_x;
  const int x0 = tx * TW, y0 = ty * TH;
  const int HWin = a.Hin * a.Win;
  const int n_steps = a.n_chunks * 3;

  STAMP(0);
  // ---- input staging plan: item i of a thread -> (h = channel half, position of the halo patch);
  //      items 0 / 1: position tid of h = 0 / 1; item 2: the patch's tail (positions 256..NPOS-1),
  //      h = wave / 2, so h is wave-uniform for every item.  Addresses are always in bounds. ----
  //      NW = 8 (512 threads): item i = position tid of h = i; threads past NPOS idle.
  static_assert((NW == 4 && XI == 3 && NPOS > NT && NPOS - NT <= NT / 2) || (NW == 8 && XI == 2 && NPOS <= NTH),
                "staging plan assumes 256 < NPOS <= 384");
  const int tail_h = (wv >> 1) & 1;                    // wave-uniform
  auto item_h = [&](int i) __attribute__((always_inline)) { return NW == 8 ? i : (i == 0 ? 0 : (i == 1 ? 1 : tail_h)); };
  int xoff[XI], xlds[XI];
  unsigned xvalid = 0, xlive = 0;
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const int h = item_h(i);
    const int pos = (NW == 8 || i < 2) ? tid : NT + (tid & (NT / 2 - 1));
    const bool live = pos < NPOS;                      // the item exists
    const int r = pos / PW;
    const int col = pos - r * PW;
    int gy = y0 + r - 1 + a.oy, gx = x0 + col - 1 + a.ox;
    if (CIRC) {
      // wrap instead of zero padding: one halo pixel on either side, so a compare-and-add per edge
      // (no integer division: that cost 2-3 % of the whole kernel in the prologue); rows / columns
      // further outside only feed outputs of a ragged tile that are never stored -- clamp them
      gy = gy < 0 ? gy + a.H : (gy >= a.H ? gy - a.H : gy);
      gx = gx < 0 ? gx + a.W : (gx >= a.W ? gx - a.W : gx);
      gy = gy >= a.H ? a.H - 1 : gy;
      gx = gx >= a.W ? a.W - 1 : gx;
    }
    const bool ok = live && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
    int off;
    if (MODE == DS_LOAD_PLAIN) off = gy * a.Win + gx;
    else if (MODE == DS_LOAD_MAXPOOL2) off = (2 * gy) * a.Win + 2 * gx;
    else off = (gy >> 1) * a.Win + (gx >> 1);
    xoff[i] = ok ? off : 0;
    xlds[i] = h * HS + pos;
    if (ok) xvalid |= (1u << i);
    if (live) xlive |= (1u << i);
  }
  // ---- VEC staging plan.  Interior unit u -> (h, patch row r, column group g, channel pair q): 32 consecutive units = 4 channel pairs x
  //      2 column groups x 4 rows (rows 8, 9: x 4 column groups x 2 rows), which puts the 32 lanes of a ds_write_b32 group on 16 banks
  //      twice (free, MI355X_MICROARCH.md LDS); 64 consecutive units share h.  Slot i of a thread = unit i * NTH + tid; the LAST slot of
  //      waves 2 and 3 is a halo-column item instead (20 lanes each: row lane / 2, column 0 or 33, h = wave - 2; eight channels of one
  //      position, as in the scalar plan), so every wave stages three (TWO: two) slots of eight elements. ----
  static_assert(!VEC || (XI - 1) * NTH + 128 == 2 * 10 * 8 * 4, "640 interior units: all slots but the last, and the last slot of waves 0 and 1");
  const bool edge_wave = VEC && (wv == 2 || wv == 3);
  int voff[XI], vlds[XI], vrow[XI];                   // float offset inside the chunk, LDS byte address of pixel 0 (high piece, buffer 0), first table row
  unsigned vvalid = 0, vlive = 0;
  if constexpr (VEC) {
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const int u = i * NTH + tid;
      voff[i] = 0; vlds[i] = 0; vrow[i] = 0;
      if (i < XI - 1 || wv < 2) {
        const int q = u & 3, G = u >> 5;
        const int h = G >= 10 ? 1 : 0, Gp = G - 10 * h;
        int r, g;
        // (32 lanes = 4 channel pairs x the 8 column groups of ONE row -- whole 128-byte lines per channel, four-way store conflicts --
        //  measured the same: profiles/r04_vec4_loads.log)
        if (Gp < 8) { r = 4 * (Gp >> 2) + ((u >> 3) & 3); g = 2 * (Gp & 3) + ((u >> 2) & 1); }
        else { r = 8 + ((u >> 3) & 1); g = 4 * (Gp - 8) + 2 * ((u >> 4) & 1) + ((u >> 2) & 1); }
        int gy = y0 + r - 1 + a.oy;
        if (CIRC) { gy = gy < 0 ? gy + a.H : (gy >= a.H ? gy - a.H : gy); gy = gy >= a.H ? a.H - 1 : gy; }
        const bool ok = gy >= 0 && gy < a.H;
        voff[i] = (8 * h + 2 * q) * HWin + (ok ? gy * a.Win + x0 + 4 * g : 0);
        vlds[i] = 16 * (h * HS + r * PW + 1 + 4 * g) + 4 * q;
        vrow[i] = 8 * h + 2 * q;
        if (ok) vvalid |= 1u << i;
        vlive |= 1u << i;
      } else if (edge_wave) {
        const int e = lane < 20 ? lane : 19, h = wv - 2;
        const int r = e >> 1, col = (e & 1) ? PW - 1 : 0;
        int gy = y0 + r - 1 + a.oy, gx = x0 + col - 1;
        if (CIRC) {
          gy = gy < 0 ? gy + a.H : (gy >= a.H ? gy - a.H : gy);
          gx = gx < 0 ? gx + a.W : (gx >= a.W ? gx - a.W : gx);
          gy = gy >= a.H ? a.H - 1 : gy;
        }
        const bool ok = lane < 20 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        voff[i] = 8 * h * HWin + (ok ? gy * a.Win + gx : 0);
        vlds[i] = 16 * (h * HS + r * PW + col);
        vrow[i] = 8 * h;
        if (ok) vvalid |= 1u << i;
        if (lane < 20) vlive |= 1u << i;
      }
    }
  }
  // the pad vector that holds table row c of a chunk (12 pad vectors behind the h = 0 image of either piece)
  auto pad_vec = [&](int c) __attribute__((always_inline)) { return c < HPAD16 ? NPOS + c : PS + NPOS + (c - HPAD16); };
  const float* in_b = a.in + (size_t)b * a.Cin * HWin;
  const u32x4* wp = a.wp + (size_t)cot * n_steps * WSLAB_VEC;
  // the sample's power-of-two activation scale, undone in the epilogue: a raw input is multiplied by it, the fused
  // norm + SiLU loader produces its activation times it.  The load is issued here and consumed behind the first patch's loads.
  ds_epi::ActScale ascale;

  float xr[XI][8];
  int xnch = KC;
  int xchunk = 0;                                     // chunk held in xr (set by x_fetch)
  auto x_fetch = [&](int chunk) __attribute__((always_inline)) {                     // issue the global loads of one patch
    const int cbase = chunk * KC;
    const float* src = in_b + (size_t)cbase * HWin;
    xchunk = chunk;
    xnch = a.Cin - cbase < KC ? a.Cin - cbase : KC;   // uniform; < KC only for a ragged last chunk
    if constexpr (VEC) {
#pragma unroll
      for (int i = 0; i < XI; ++i) {
        const float* p0 = src + voff[i];
        if (i < XI - 1 || wv < 2) {
          const f32x4 t0 = *reinterpret_cast<const f32x4*>(p0);
          const f32x4 t1 = *reinterpret_cast<const f32x4*>(p0 + HWin);
#pragma unroll
          for (int k = 0; k < 4; ++k) { xr[i][k] = t0[k]; xr[i][4 + k] = t1[k]; }
        } else if (edge_wave) {
#pragma unroll
          for (int k = 0; k < 8; ++k) xr[i][k] = p0[k * HWin];
        }
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const int h = item_h(i);
      const float* p0 = src + xoff[i];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int c = 8 * h + k;
        const float* p = p0 + (c < xnch ? c : 0) * HWin;       // channels past Cin read channel 0 (zeroed later)
        if (MODE == DS_LOAD_MAXPOOL2) {
          const float2 t0 = *reinterpret_cast<const float2*>(p);
          const float2 t1 = *reinterpret_cast<const float2*>(p + a.Win);
          xr[i][k] = fmaxf(fmaxf(t0.x, t0.y), fmaxf(t1.x, t1.y));
        } else {
          xr[i][k] = *p;
        }
      }
      if (MODE == DS_LOAD_MAXPOOL2) __builtin_amdgcn_sched_barrier(0);   // one item at a time (registers)
    }
  };
  const float* pre_b = PRE ? a.prenorm + (size_t)b * a.n_chunks * KC * 4 : nullptr;   // [B][n_chunks*16][4]
  // PRE: normalise + SiLU one staging item in registers, right before its fp16 split.  (Slotting this
  // VALU / transcendental work between the MFMAs of the same wave was measured 2-3 % SLOWER than
  // leaving it in one block: the co-resident workgroup's waves already fill the matrix pipe then.)
  auto x_activate = [&](int i) __attribute__((always_inline)) {
    // constant address space: the table was written by an earlier kernel and is read-only here, which
    // is what lets the compiler use s_load (vector loads + vmcnt(0) per channel pair otherwise)
    typedef const __attribute__((address_space(4))) f32x4* cptr;
    cptr pp = (cptr)(reinterpret_cast<const f32x4*>(pre_b) + xchunk * KC);
    const int h = item_h(i);
    // the table is padded to whole 16-channel chunks (zeros), so the 8 rows of this item are one contiguous
    // 128-byte scalar load: one s_load + one wait per item instead of one dependent round trip per channel pair
    f32x4 p[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) p[k] = pp[8 * h + k];
#ifdef DS_PRE_UNSCALED       // measurement builds: what does the exponent cost the fused loader?
#pragma unroll
    for (int k = 0; k < 8; ++k) xr[i][k] = fast_silu((xr[i][k] - p[k][0]) * p[k][1] + p[k][2]);
#else
    // the sample's activation exponent rides in the table's fourth column (2^-k, the same in every row; 0 = none): scalar select
    const float inv = p[0][3] == 0.f ? 1.0f : p[0][3];
    ascale.inv_scale = inv;                                          // the epilogue undoes it (unscale_from_inv)
#pragma unroll
    for (int k = 0; k < 8; ++k) xr[i][k] = ds_h3::fast_silu_scaled((xr[i][k] - p[k][0]) * p[k][1] + p[k][2], inv);
#endif
  };
  auto x_store_vec = [&](int buf) __attribute__((always_inline)) {
    if constexpr (VEC) {
      unsigned char* xbytes = smem + (size_t)buf * XBV * 16;
      const u32x4* rows = Xs + buf * XBV;                    // the chunk's table rows in the buffer's pad vectors
#pragma unroll
      for (int i = 0; i < XI; ++i) {
        const bool ok = (vvalid >> i) & 1u;
        if (i < XI - 1 || wv < 2) {
          // interior unit: xr[i][0..3] = four pixels of channel 2q, xr[i][4..7] = of channel 2q + 1
          if constexpr (PRE) {
            const f32x4 p0 = __builtin_bit_cast(f32x4, rows[pad_vec(vrow[i])]);
            const f32x4 p1 = __builtin_bit_cast(f32x4, rows[pad_vec(vrow[i] + 1)]);
            const float inv = p0[3] == 0.f ? 1.0f : p0[3];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              xr[i][k] = ds_h3::fast_silu_scaled((xr[i][k] - p0[0]) * p0[1] + p0[2], inv);
              xr[i][4 + k] = ds_h3::fast_silu_scaled((xr[i][4 + k] - p1[0]) * p1[1] + p1[2], inv);
            }
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            float v0 = ok ? xr[i][k] : 0.f;
            float v1 = ok ? xr[i][4 + k] : 0.f;
            if constexpr (!PRE) { v0 *= ascale.in_scale; v1 *= ascale.in_scale; }
            unsigned ph, pl;
            split2(v0, v1, ph, pl);
            *reinterpret_cast<unsigned*>(xbytes + vlds[i] + 16 * k) = ph;
            *reinterpret_cast<unsigned*>(xbytes + vlds[i] + 16 * k + PS * 16) = pl;
          }
        } else if (edge_wave) {
          // halo-column item: eight channels of one position (wave-uniform h: the table rows through the scalar cache)
          if constexpr (PRE) {
            typedef const __attribute__((address_space(4))) f32x4* cptr;
            cptr pp = (cptr)(reinterpret_cast<const f32x4*>(pre_b) + xchunk * KC) + 8 * (wv - 2);
            f32x4 p[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) p[k] = pp[k];
            const float inv = p[0][3] == 0.f ? 1.0f : p[0][3];
#pragma unroll
            for (int k = 0; k < 8; ++k) xr[i][k] = ds_h3::fast_silu_scaled((xr[i][k] - p[k][0]) * p[k][1] + p[k][2], inv);
          }
          if ((vlive >> i) & 1u) {
            u32x4 qh, ql;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              float v0 = ok ? xr[i][2 * k] : 0.f;
              float v1 = ok ? xr[i][2 * k + 1] : 0.f;
              if constexpr (!PRE) { v0 *= ascale.in_scale; v1 *= ascale.in_scale; }
              unsigned ph, pl;
              split2(v0, v1, ph, pl);
              qh[k] = ph; ql[k] = pl;
            }
            *reinterpret_cast<u32x4*>(xbytes + vlds[i]) = qh;
            *reinterpret_cast<u32x4*>(xbytes + vlds[i] + PS * 16) = ql;
          }
        }
      }
    }
  };
  // VEC + PRE: wave 0 fetches the 16 table rows of a chunk (one 16-byte load in its first 16 lanes) and parks them in the pad vectors of
  // the X buffer the chunk is staged into, one barrier before the activation reads them
  f32x4 prow = {0.f, 0.f, 0.f, 0.f};
  auto rows_fetch = [&](int chunk) __attribute__((always_inline)) {
    if constexpr (VEC && PRE) {
      if (wv == 0) prow = reinterpret_cast<const f32x4*>(pre_b)[chunk * KC + (lane & 15)];
    }
  };
  auto rows_park = [&](int buf) __attribute__((always_inline)) {
    if constexpr (VEC && PRE) {
      if (wv == 0 && lane < 16) Xs[buf * XBV + pad_vec(lane)] = __builtin_bit_cast(u32x4, prow);
    }
  };
  auto x_store = [&](int buf) __attribute__((always_inline)) {                       // [normalise + SiLU,] split to fp16 pieces, write the LDS image
    if constexpr (VEC) { x_store_vec(buf); return; }
    u32x4* xb = Xs + buf * XBV;
    if (PRE) {
#pragma unroll
      for (int i = 0; i < XI; ++i) x_activate(i);
    }
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      if ((NW == 4 && i < 2) || ((xlive >> i) & 1u)) {
        const int h = item_h(i);
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
        xb[xlds[i]] = qh;                // piece 0: [h][pos]
        xb[PS + xlds[i]] = ql;           // piece 1
      }
    }
  };
  auto w_fetch = [&](int step, int slot) __attribute__((always_inline)) {            // LDS-DMA of slab `step` into ring slot (= step % 3)
    const u32x4* src = wp + (size_t)step * WSLAB_VEC;
    u32x4* dst = Ws + slot * WSV;
#pragma unroll
    for (int i = 0; i < (COTS * WPIECES + NW - 1) / NW; ++i) {
      const int k = wv + NW * i;                      // wave-uniform
      if (k < COTS * WPIECES) {
        // TWO: pieces WPIECES.. belong to the second channel tile, whose slabs start n_steps slabs further on
        const int kc = (TWO && k >= WPIECES) ? 1 : 0, kk = k - kc * WPIECES;
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(src + (size_t)kc * n_steps * WSLAB_VEC + 64 * kk + lane),
            (__attribute__((address_space(3))) void*)(dst + kc * WSLAB_VEC + 64 * kk), 16, 0, 0);
      }
    }
  };

  f32x16 acc[MT][2];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[m][r][q] = 0.f;

  using FragsT = Frags<MT>;
  // operand fetch for (weight slot, X buffer, ky, kx)
  auto frag_load = [&](FragsT& f, int slot, int xbuf, int ky, int kx) __attribute__((always_inline)) {
    const u32x4* wb = Ws + slot * WSLAB_VEC;
    const u32x4* xb = Xs + xbuf * XBV;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
      for (int m = 0; m < MT; ++m)
        f.a[p][m] = *reinterpret_cast<const f16x8*>(&wb[((p * 3 + kx) * 2 + lh) * COT + 32 * (m + mh) + li]);
#pragma unroll
      for (int r = 0; r < 2; ++r)
        f.b[p][r] = *reinterpret_cast<const f16x8*>(&xb[p * PS + lh * HS + (wave_row + ROWS_PER_R * r + ky) * PW + lane_pos + kx]);
    }
  };
  auto frag_mma = [&](const FragsT& f) __attribute__((always_inline)) {              // lo*hi, hi*lo, hi*hi
    constexpr int PA[3] = {1, 0, 0};
    constexpr int PB[3] = {0, 1, 0};
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 2; ++r)
          acc[m][r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[PA[t]][m], f.b[PB[t]][r], acc[m][r], 0, 0, 0);
  };
  auto reads_between_mfmas = [&]() __attribute__((always_inline)) {                  // the block's ds_read_b128, one behind each of its first MFMAs
    constexpr int NREADS = 2 * MT + 4, NMFMA = 6 * MT;
    constexpr int PAIRS = NREADS < NMFMA ? NREADS : NMFMA;
#pragma unroll
    for (int i = 0; i < PAIRS; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    if (NMFMA > PAIRS) __builtin_amdgcn_sched_group_barrier(0x008, NMFMA - PAIRS, 0);
  };

  // ---- IMGIN: an X buffer is one linear region of XBV vectors holding four images (piece, h) at 0, HS, PS, PS + HS; it is
  //      filled by NINST 64-lane DMA instructions, instruction k = wv + 4 i at vector 64 k -- the last one moved back so that it
  //      ends with the buffer (it rewrites a few vectors of its predecessor with the same data).  Lane -> linear index ->
  //      (image, position); the pad vectors behind the h = 0 images read the patch's last position (never used). ----
  constexpr int NINST = IMGIN ? (XBV + 63) / 64 : 0;
  static_assert(!IMGIN || NINST <= 24, "six instructions per wave");
  int doff[IMGIN ? 6 : 1];
  const int Wp = a.W + 2, Hp = a.H + 2;
  if constexpr (IMGIN) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int k = wv + 4 * i;
      const int L = (k < NINST - 1 ? 64 * k : XBV - 64) + lane;
      const int piece = L >= PS ? 1 : 0, Lp = L - piece * PS;
      const int hh = Lp >= HS ? 1 : 0;
      const int q = 2 * piece + hh;                        // image
      int pos = Lp - hh * HS;
      pos = pos < NPOS ? pos : NPOS - 1;
      const int r = pos / PW, col = pos - r * PW;
      int py = y0 + r, px = x0 + col;                      // padded coordinates: patch origin (y0 - 1, x0 - 1) + 1
      py = py < Hp ? py : Hp - 1;                          // ragged tiles: the zero border
      px = px < Wp ? px : Wp - 1;
      doff[i] = (q * Hp + py) * Wp + px;
    }
  }
  const u32x4* img_b = IMGIN ? reinterpret_cast<const u32x4*>(a.in) + (size_t)b * a.n_chunks * 4 * Hp * Wp : nullptr;
  auto x_dma = [&](int chunk, int buf) __attribute__((always_inline)) {
    if constexpr (IMGIN) {
      const u32x4* src = img_b + (size_t)chunk * 4 * Hp * Wp;
      u32x4* dst = Xs + buf * XBV;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int k = wv + 4 * i;                          // wave-uniform
        if (k < NINST)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + doff[i]),
                                           (__attribute__((address_space(3))) void*)(dst + (k < NINST - 1 ? 64 * k : XBV - 64)), 16, 0, 0);
      }
    }
  };

  // ---- prologue: patch 0, weight slabs 0 and 1 ----
  // (scalar loads share one counter: issued any earlier, the first wait for a kernel argument would wait for this load too)
  if constexpr (!IMGIN && !PRE) __builtin_amdgcn_sched_barrier(0);      // raw-input launches only: the others carry no exponent load
  const unsigned amax_bits = ds_epi::act_bits((IMGIN || PRE) ? nullptr : a.in_amax, b);
  if constexpr (IMGIN) x_dma(0, 0); else
  x_fetch(0);
  rows_fetch(0);
  // bias / shift of the workgroup's channel tile(s): threads 0-127 (TWO: 0-255, 128 per tile)
  const float bias_shift = ds_epi::fetch_bias_shift(a.bias, a.shift, a.shift_stride, b, (cot + (TWO ? (tid >> 7) & 1 : 0)) * COT, a.Cout, COTS);
  w_fetch(0, 0);
  if (n_steps > 1) w_fetch(1, 1);
  STAMP(1);
  if constexpr (!IMGIN && !PRE) __builtin_amdgcn_sched_barrier(0);
  ascale = ds_epi::act_scale_of(PRE ? 0u : amax_bits, a.wshift);   // PRE: the table carries the exponent (x_activate)
  if constexpr (VEC && PRE) {
    // the exponent for the epilogue (the same in every row of the sample's table), through the scalar cache; chunk 0's rows into
    // buffer 0's pads, published by a barrier of their own
    typedef const __attribute__((address_space(4))) f32x4* cptr;
    const float inv0 = ((cptr)(reinterpret_cast<const f32x4*>(pre_b)))[0][3];
    ascale.inv_scale = inv0 == 0.f ? 1.0f : inv0;
    rows_park(0);
    __syncthreads();
  }
  if constexpr (!IMGIN) x_store(0);
  ds_epi::commit_bias_shift(BS, bias_shift, COTS);
  __syncthreads();
  STAMP(2);
  STAMP_CLK(6);

  f32x4 acc16[M16][4];                                 // S16: [16-channel tile][16-position tile]
  if constexpr (S16) {
#pragma unroll
    for (int m = 0; m < M16; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc16[m][n][q] = 0.f;
    const int i16 = lane & 15, h16 = (lane >> 4) & 1;
    const bool tapB = lane >= 32;                        // lane groups 2, 3 read the pair's second tap
    // lane part of the operand addresses: weights [piece][kx][h][co], input [piece][h][position]
    const int wlane = h16 * COT + i16 + 32 * mh + cw * WSLAB_VEC;   // eight waves: the wave's 32-channel half, or (TWO) its channel tile's slab
    int xlane[4];
#pragma unroll
    for (int n = 0; n < 4; ++n)
      xlane[n] = h16 * HS + (wave_row + ROWS_PER_R * (n >> 1)) * PW + (W16 ? (n & 1) * PW + i16 : 16 * (n & 1) + i16);
    struct F16 { f16x8 a[2][M16]; f16x8 b[2][4]; };
    // one K = 32 group: taps (slot, ky, kx, X buffer) A and B of the pair; every argument is a compile-time constant
    auto pair = [&](int slotA, int kyA, int kxA, int xbA, int slotB, int kyB, int kxB, int xbB) __attribute__((always_inline)) {
      const int wofs = (tapB ? slotB * WSV + kxB * 2 * COT : slotA * WSV + kxA * 2 * COT) + wlane;
      const int xofs = tapB ? xbB * XBV + kyB * PW + kxB : xbA * XBV + kyA * PW + kxA;
      F16 f;
#pragma unroll
      for (int p = 0; p < 2; ++p) {
#pragma unroll
        for (int m = 0; m < M16; ++m) f.a[p][m] = *reinterpret_cast<const f16x8*>(&Ws[wofs + p * 6 * COT + 16 * m]);
#pragma unroll
        for (int n = 0; n < 4; ++n) f.b[p][n] = *reinterpret_cast<const f16x8*>(&Xs[xofs + p * PS + xlane[n]]);
      }
      constexpr int PA[3] = {1, 0, 0};
      constexpr int PB[3] = {0, 1, 0};
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int m = 0; m < M16; ++m)
#pragma unroll
          for (int n = 0; n < 4; ++n)
            acc16[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.a[PA[t]][m], f.b[PB[t]][n], acc16[m][n], 0, 0, 0);
    };
    const bool early_stage = a.two_early != 0;
    // staging of a step (weights two steps ahead, the next patch at ky = 0 / 1), as in the 32x32x16 schedule
    auto stage_in = [&](int chunk, int ky) __attribute__((always_inline)) {
      const int g = chunk * 3 + ky;
      if (g + 2 < n_steps) w_fetch(g + 2, (ky + 2) % 3);
#ifdef DS_STAMP
      const bool fetch = !a.no_stage;
#else
      constexpr bool fetch = true;
#endif
      if constexpr (IMGIN) {
        // the next patch straight into the other buffer: last read before the barrier that ended the previous step, landed
        // by the barrier that ends the step after this one (__syncthreads waits for the wave's outstanding DMA)
        if (ky == 0 && chunk + 1 < a.n_chunks) x_dma(chunk + 1, (chunk + 1) & 1);
      } else
      if (fetch && ky == 0 && chunk + 1 < a.n_chunks) { rows_fetch(chunk + 1); x_fetch(chunk + 1); }
      // TWO: the eight waves run one program behind one barrier per step, so unshifted all of them multiply at the same time and all
      // of them activate + split at the same time (profiles/r04_pmc_stalls.log: vector and matrix pipes co-executing 1.7 % of the
      // launch).  Waves 0-3 therefore stage their share of the next patch BEFORE this step's matrix instructions, waves 4-7 (their SIMD
      // partners) after them: each SIMD then holds one vector stream beside one matrix stream in both halves of the step
      // (MI355X_MICROARCH.md, Two waves per SIMD, item 9).  Legal: the patch goes to the OTHER X buffer, which nobody reads in this step.
      if constexpr (TWO && !IMGIN) {
        if (early_stage && wv < 4 && fetch && ky == 1 && chunk + 1 < a.n_chunks) x_store((chunk & 1) ^ 1);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    auto stage_out = [&](int chunk, int ky, int xbuf) __attribute__((always_inline)) {
      __builtin_amdgcn_sched_barrier(0);
#ifdef DS_STAMP
      const bool store = a.no_stage != 1;
#else
      constexpr bool store = true;
#endif
      if constexpr (!IMGIN) {
        if (store && ky == 0 && chunk + 1 < a.n_chunks) rows_park(xbuf ^ 1);
        if (store && ky == 1 && chunk + 1 < a.n_chunks && !(TWO && early_stage && wv < 4)) x_store(xbuf ^ 1);
      }
      __syncthreads();
    };
    for (int chunk = 0; chunk < a.n_chunks; chunk += 2) {          // n_chunks is even (checked by the launcher)
      stage_in(chunk, 0);     pair(0, 0, 0, 0, 0, 0, 1, 0); pair(0, 0, 2, 0, 1, 1, 0, 0);   stage_out(chunk, 0, 0);
      stage_in(chunk, 1);     pair(1, 1, 1, 0, 1, 1, 2, 0);                                 stage_out(chunk, 1, 0);
      stage_in(chunk, 2);     pair(2, 2, 0, 0, 2, 2, 1, 0); pair(2, 2, 2, 0, 0, 0, 0, 1);   stage_out(chunk, 2, 0);
      stage_in(chunk + 1, 0); pair(0, 0, 1, 1, 0, 0, 2, 1);                                 stage_out(chunk + 1, 0, 1);
      stage_in(chunk + 1, 1); pair(1, 1, 0, 1, 1, 1, 1, 1); pair(1, 1, 2, 1, 2, 2, 0, 1);   stage_out(chunk + 1, 1, 1);
      stage_in(chunk + 1, 2); pair(2, 2, 1, 1, 2, 2, 2, 1);                                 stage_out(chunk + 1, 2, 1);
    }
  } else {
  FragsT fA, fB;
  frag_load(fA, 0, 0, 0, 0);

  // One step = (chunk, ky).  `cur` holds the operands of kx = 0 on entry; on exit `oth` holds
  // the operands of the NEXT step's kx = 0 (three blocks per step flip the roles).
  // The ring slot of step g = 3*chunk + ky is g % 3 = ky and the X buffer is chunk & 1: both are
  // compile-time constants at every call site below, so all LDS addresses are base + immediate.
  auto step = [&](FragsT& cur, FragsT& oth, int chunk, int ky, int xbuf) __attribute__((always_inline)) {
    const int g = chunk * 3 + ky;
    const int slot = ky;
    const bool more_chunks = chunk + 1 < a.n_chunks;
    if (g + 2 < n_steps) w_fetch(g + 2, (ky + 2) % 3);         // lands a whole step ahead of its use
    if (ky == 0 && more_chunks) x_fetch(chunk + 1);
    __builtin_amdgcn_sched_barrier(0);
    frag_load(oth, slot, xbuf, ky, 1);
    frag_mma(cur);
    reads_between_mfmas();
    __builtin_amdgcn_sched_barrier(0);
    frag_load(cur, slot, xbuf, ky, 2);
    frag_mma(oth);
    reads_between_mfmas();
    __builtin_amdgcn_sched_barrier(0);
    if (g + 1 < n_steps) {                                     // next step's kx = 0 operands
      const int nky = ky == 2 ? 0 : ky + 1;
      frag_load(oth, nky, ky == 2 ? xbuf ^ 1 : xbuf, nky, 0);
    }
    frag_mma(cur);
    if (g + 1 < n_steps) reads_between_mfmas();
    __builtin_amdgcn_sched_barrier(0);
    if (ky == 1 && more_chunks) x_store(xbuf ^ 1);             // published by this step's barrier
    __syncthreads();
  };

  // Three blocks per step flip the operand-set roles, so the loop body is two chunks (six steps)
  // of straight-line code -- a parity branch instead costs ~100 VGPRs in the allocator.
  int chunk = 0;
  for (; chunk + 1 < a.n_chunks; chunk += 2) {
    step(fA, fB, chunk, 0, 0);
    step(fB, fA, chunk, 1, 0);
    step(fA, fB, chunk, 2, 0);
    step(fB, fA, chunk + 1, 0, 1);
    step(fA, fB, chunk + 1, 1, 1);
    step(fB, fA, chunk + 1, 2, 1);
  }
  if (chunk < a.n_chunks) {
    step(fA, fB, chunk, 0, 0);
    step(fB, fA, chunk, 1, 0);
    step(fA, fB, chunk, 2, 0);
  }

  }   // MFMA shape
  STAMP(3);
  STAMP_CLK(7);
  // ---- epilogue (ds_conv_epilogue.h): LDS transpose -> 16-byte stores.  The staging buffers are
  //      dead after the last step's barrier; each wave takes a private 16 KiB of them. ----
  {
    ds_epi::Args e;
    e.out = a.out; e.bias = a.bias; e.shift = a.shift; e.res1 = a.res1; e.res2 = a.res2; e.res1_up = a.res1_up;
    e.unscale = PRE ? ds_epi::unscale_from_inv(ascale.inv_scale, a.wshift) : ds_epi::unscale_from_in(ascale.in_scale, a.wshift);
    e.shift_stride = a.shift_stride;
    e.out_amax = a.out_amax ? a.out_amax + b : nullptr;
    e.b = b; e.co_base = (cot + cw) * COT + 32 * mh; e.y0 = y0 + wave_row; e.x0 = x0;
    e.Cout = a.Cout; e.H = a.H; e.W = a.W;
    e.tile_stats = a.tile_stats; e.tile = ty * a.tiles_x + tx; e.ntiles = a.tiles_x * a.tiles_y;
    constexpr int WTILE = 32 * MT * 2 * 32;                           // floats of the wave's private transpose region
    float* tile = reinterpret_cast<float*>(smem) + wv * WTILE;
    if constexpr (S16) {
      ds_epi::store_tile16<W16, MT>(acc16, tile, BS + 32 * mh + 128 * cw, e);
    } else {
      ds_epi::store_tile<W16, MT>(acc, tile, BS + 32 * mh, e);
    }
    if (a.tile_stats) {
      __syncthreads();
      if constexpr (TWO) {                      // threads 0-63 combine the first channel tile's four waves, 64-127 the second's
        if (tid < 128) {
          e.co_base = (cot + (tid >> 6)) * COT;
          ds_epi::store_tile_stats<2>(reinterpret_cast<const float*>(smem) + (tid >> 6) * 4 * WTILE, WTILE, e, tid & 63);
        }
      } else {
        e.co_base = cot * COT;
        ds_epi::store_tile_stats<MT>(reinterpret_cast<const float*>(smem), WTILE, e);
      }
    }
  }
#ifdef DS_STAMP
  STAMP(4);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  STAMP(5);
#endif
}

// torch [Cout][Cin][3][3] fp32 (times 2^wshift) -> [cot][chunk][ky][piece][kx][h][co 64][ci 8] fp16
__global__ void k_pack3h(_Float16* packed, const float* __restrict__ w, int Cout, int Cin, int n_chunks, float scale,
                         size_t total) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  size_t t = i;
  const int c8 = t % 8; t /= 8;
  const int co64 = t % COT; t /= COT;
  const int h = t % 2; t /= 2;
  const int kx = t % 3; t /= 3;
  const int piece = t % 2; t /= 2;
  const int ky = t % 3; t /= 3;
  const int chunk = t % n_chunks; t /= n_chunks;
  const int cot = (int)t;
  const int co = cot * COT + co64, ci = chunk * KC + 8 * h + c8;
  float v = 0.f;
  if (co < Cout && ci < Cin) v = w[((size_t)co * Cin + ci) * 9 + ky * 3 + kx] * scale;   // exact: power of two
  const _Float16 hi = (_Float16)v;
  const _Float16 lo = (_Float16)(v - (float)hi);
  packed[i] = piece == 0 ? hi : lo;
}

template <int MODE, bool W16, bool PRE, bool CIRC, int NW, bool S16 = false, bool IMGIN = false, bool TWO = false, bool VEC = false>
int launch_conv3h_w(const Conv3hArgs& a, hipStream_t s) {
  constexpr int LDSB = TWO ? LDS_BYTES_TWO : (S16 ? LDS_BYTES16 : LDS_BYTES);
  {
    const int rc = ds::ensure_dynamic_lds<&k_conv3h<MODE, W16, PRE, CIRC, NW, S16, IMGIN, TWO, VEC>>(LDSB, "hipFuncSetAttribute(conv3h)");
    if (rc != DS_OK) return rc;
  }
  const long long tiles = (long long)a.tiles_y * a.tiles_x;
  DS_REQUIRE(tiles > 0 && tiles < 65536 && a.B < 65536, DS_ERR_SHAPE,
             "ds_conv2d_h3: %lld pixel tiles x %d samples exceed the grid limits (65535 each)", tiles, a.B);
  hipLaunchKernelGGL((k_conv3h<MODE, W16, PRE, CIRC, NW, S16, IMGIN, TWO, VEC>), dim3((unsigned)(TWO ? a.n_cot / 2 : a.n_cot), (unsigned)tiles, (unsigned)a.B),
                     dim3(64 * NW), LDSB, s, a);
  DS_CHECK_LAUNCH("ds_conv2d_h3");
  return DS_OK;
}

// MFMA shape: v_mfma_f32_16x16x32_f16 wherever the layer has an even number of 16-channel chunks (measured on MI355X,
// profiles/r02_stamps_shape16.log: 190 -> 173 us at 256 channels, 236 -> 215 us at 128 channels with the fused loader;
// bench.py 72.8 -> 76.7 samples/s), v_mfma_f32_32x32x16_f16 otherwise.  DS_CONV_SHAPE=32 forces the latter (A/B runs).
inline bool conv3h_shape16() {
  static const bool on = [] { const char* e = getenv("DS_CONV_SHAPE"); return !(e && atoi(e) == 32); }();
  return on;
}

// Waves per workgroup.  Default: eight for the fused norm+SiLU loader (its staging VALU work spreads over twice the
// waves: 3-5 % faster at 64 and 128 channels), four otherwise (at 256 channels the extra LDS operand reads of the
// eight-wave tiling cost 2-3 %: the kernel is power-limited).  DS_CONV_WAVES=4|8 forces one (A/B runs).
inline int conv3h_waves(bool pre) {
  static const int forced = [] { const char* e = getenv("DS_CONV_WAVES"); const int v = e ? atoi(e) : 0; return (v == 4 || v == 8) ? v : 0; }();
  return forced ? forced : (pre ? 8 : 4);
}

// Waves of the 16x16x32 variant with the fused loader.  Eight (the staging's vector work spread over twice the waves, as in the
// 32x32x16 kernel) measured 75.5 -> 76.0 samples/s on one box and 79.6 -> 79.0 on another, ADM-128 33.3 -> 33.45 ms per
// evaluation: noise.  Default four; DS_CONV_WAVES16=8 selects eight up to 128 input channels (A/B runs).
inline int conv3h_waves16() {
  static const int v = [] { const char* e = getenv("DS_CONV_WAVES16"); return (e && atoi(e) == 8) ? 8 : 4; }();
  return v;
}

// Two channel tiles per workgroup (TWO): default for the fused norm + SiLU loader at exactly two channel tiles (the 128-channel
// level: the activation is otherwise computed once per tile) WHEN the halved grid still fills the chip -- at least
// DS_CONV_TWO_MIN (256: one per CU) workgroups.  Measured on MI355X: config 5's share (2048 such workgroups per launch)
// 9.24 -> 9.13 ms per evaluation with it; a [4, 4, 32, 32] latent on a 32-channel network (a handful of workgroups: their
// latency is the launch's) 21.85 -> 23.3 ms per 10-step forecast, so not there; the headline's level-1 launches (1024 such workgroups at
// batch 64): 79.9 -> 80.45 samples/s with it.
// DS_CONV_TWO=0 switches it off, =2 extends it to every even tile count, the plain loader and any grid (A/B runs).
inline int conv3h_two() {
  static const int v = [] { const char* e = getenv("DS_CONV_TWO"); return e ? atoi(e) : 1; }();
  return v;
}
inline long long conv3h_two_min() {
  static const long long v = [] { const char* e = getenv("DS_CONV_TWO_MIN"); return e ? atoll(e) : 256ll; }();
  return v;
}

// 16-byte patch loads (VEC): default on wherever the shape allows; DS_CONV_VEC=0 keeps the one-pixel staging items (A/B runs, and
// the bit-for-bit comparison of the two staging plans in tools/conv_vec_check.py).
inline bool conv3h_vec() {
  static const bool on = [] { const char* e = getenv("DS_CONV_VEC"); return !(e && atoi(e) == 0); }();
  return on;
}

template <int MODE, bool W16, bool PRE, bool CIRC>
int launch_conv3h_c(const Conv3hArgs& a, hipStream_t s) {
  bool vec = false;
  if constexpr (MODE == DS_LOAD_PLAIN && !W16)
    vec = conv3h_vec() && a.W % 32 == 0 && a.ox == 0 && a.Cin % KC == 0 && (reinterpret_cast<uintptr_t>(a.in) & 15u) == 0;
  if constexpr (MODE == DS_LOAD_PLAIN) {
    const int two = conv3h_two();
    if (conv3h_shape16() && a.n_chunks % 2 == 0 && a.n_cot % 2 == 0 && ((two == 1 && PRE && a.n_cot == 2 && (long long)a.tiles_y * a.tiles_x * a.B >= conv3h_two_min()) || two == 2)) {
      if constexpr (!W16) { if (vec) return launch_conv3h_w<MODE, W16, PRE, CIRC, 8, true, false, true, true>(a, s); }
      return launch_conv3h_w<MODE, W16, PRE, CIRC, 8, true, false, true>(a, s);
    }
  }
  if (conv3h_shape16() && a.n_chunks % 2 == 0) {
    if constexpr (PRE && MODE == DS_LOAD_PLAIN) {
      if (conv3h_waves16() == 8 && a.n_chunks <= 8) return launch_conv3h_w<MODE, W16, PRE, CIRC, 8, true>(a, s);   // up to 128 input channels
    }
    if constexpr (MODE == DS_LOAD_PLAIN && !W16) { if (vec) return launch_conv3h_w<MODE, W16, PRE, CIRC, 4, true, false, false, true>(a, s); }
    return launch_conv3h_w<MODE, W16, PRE, CIRC, 4, true>(a, s);
  }
  // the eight-wave max-pool loader would spill (180 B/lane of scratch at 128 VGPRs: four loads per element in flight): four waves
  if constexpr (MODE == DS_LOAD_MAXPOOL2) return launch_conv3h_w<MODE, W16, PRE, CIRC, 4>(a, s);
  else return conv3h_waves(PRE) == 8 ? launch_conv3h_w<MODE, W16, PRE, CIRC, 8>(a, s) : launch_conv3h_w<MODE, W16, PRE, CIRC, 4>(a, s);
}

template <int MODE, bool W16, bool PRE>
int launch_conv3h(const Conv3hArgs& a, hipStream_t s) {
  return a.circular ? launch_conv3h_c<MODE, W16, PRE, true>(a, s) : launch_conv3h_c<MODE, W16, PRE, false>(a, s);
}

}  // namespace

extern "C" {

size_t ds_conv2d_h3_packed_bytes(int Cout, int Cin) {
  if (Cout <= 0 || Cin <= 0) return 0;
  const size_t n_cot = (Cout + COT - 1) / COT, n_chunks = (Cin + KC - 1) / KC;
  return n_cot * n_chunks * 3 * (size_t)WSLAB_VEC * 16;
}

int ds_conv2d_h3_pack_weights(void* packed, const float* w, int Cout, int Cin, int wshift, void* stream) {
  DS_REQUIRE(packed && w, DS_ERR_NULL, "ds_conv2d_h3_pack_weights: NULL pointer");
  DS_REQUIRE(Cout > 0 && Cin > 0, DS_ERR_SHAPE, "ds_conv2d_h3_pack_weights: Cout=%d Cin=%d", Cout, Cin);
  DS_REQUIRE(wshift >= -40 && wshift <= 40, DS_ERR_SHAPE, "ds_conv2d_h3_pack_weights: wshift %d out of range", wshift);
  const int n_chunks = (Cin + KC - 1) / KC;
  const size_t total = ds_conv2d_h3_packed_bytes(Cout, Cin) / 2;
  hipLaunchKernelGGL(k_pack3h, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ds::as_stream(stream),
                     reinterpret_cast<_Float16*>(packed), w, Cout, Cin, n_chunks, ldexpf(1.0f, wshift), total);
  DS_CHECK_LAUNCH("ds_conv2d_h3_pack_weights");
  return DS_OK;
}

int ds_conv2d_h3(float* out, const float* in, const void* w_packed, int wshift, const float* bias, const float* shift,
                 int shift_stride, const float* res1, const float* res2, int B, int Cin, int Cout, int H, int W,
                 int load_mode, const float* prenorm, float* tile_stats, const unsigned* in_amax, unsigned* out_amax,
                 void* stream) {
  DS_REQUIRE(out && in && w_packed, DS_ERR_NULL, "ds_conv2d_h3: NULL pointer");
  DS_REQUIRE(!(prenorm && in_amax), DS_ERR_UNSUPPORTED, "ds_conv2d_h3: in_amax is for raw inputs (with prenorm the table's fourth column carries the exponent)");
  DS_REQUIRE(B >= 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, DS_ERR_SHAPE,
             "ds_conv2d_h3: bad shape B=%d Cin=%d Cout=%d H=%d W=%d", B, Cin, Cout, H, W);
  const int circular = (load_mode & DS_PAD_CIRCULAR) ? 1 : 0;
  const int res1_up = (load_mode & DS_RES1_UPSAMPLED) ? 1 : 0;
  // tap-origin offset, two signed 4-bit fields (DS_TAP_OFFSET): 0 for a plain 3x3 convolution
  const int oy = (int)((load_mode >> 8) & 15) - (((load_mode >> 8) & 8) ? 16 : 0);
  const int ox = (int)((load_mode >> 12) & 15) - (((load_mode >> 12) & 8) ? 16 : 0);
  load_mode &= ~(DS_PAD_CIRCULAR | DS_RES1_UPSAMPLED | 0xff00);
  // periodic padding with a tap offset (k x k kernels as shifted 3x3 blocks): the loader wraps once per edge, which covers a window
  // that reaches |offset| + 1 pixels outside on a plane at least that large (torch's own limit for F.pad(mode="circular"))
  DS_REQUIRE(!circular || (H >= (oy < 0 ? -oy : oy) + 1 && W >= (ox < 0 ? -ox : ox) + 1), DS_ERR_SHAPE,
             "ds_conv2d_h3: periodic padding with tap offset (%d, %d) needs a field at least as large as the window's reach (%d x %d)", oy, ox, H, W);
  DS_REQUIRE(!res1_up || (res1 && H % 2 == 0 && W % 2 == 0), DS_ERR_SHAPE, "ds_conv2d_h3: RES1_UPSAMPLED needs res1 and even H, W");
  DS_REQUIRE(load_mode >= 0 && load_mode <= 2, DS_ERR_UNSUPPORTED, "ds_conv2d_h3: load_mode %d", load_mode);
  DS_REQUIRE(load_mode != DS_LOAD_UPSAMPLE2 || (H % 2 == 0 && W % 2 == 0), DS_ERR_SHAPE,
             "ds_conv2d_h3: UPSAMPLE2 needs even output H, W (got %d x %d)", H, W);
  DS_REQUIRE(shift == nullptr || shift_stride == 0 || shift_stride >= Cout, DS_ERR_SHAPE,
             "ds_conv2d_h3: shift_stride %d < Cout %d", shift_stride, Cout);
  DS_REQUIRE((reinterpret_cast<uintptr_t>(w_packed) & 15u) == 0, DS_ERR_SHAPE, "ds_conv2d_h3: w_packed must be 16-byte aligned");
  DS_REQUIRE(load_mode != DS_LOAD_MAXPOOL2 || (reinterpret_cast<uintptr_t>(in) & 7u) == 0, DS_ERR_SHAPE,
             "ds_conv2d_h3: MAXPOOL2 input must be 8-byte aligned");
  DS_REQUIRE(wshift >= -40 && wshift <= 40, DS_ERR_SHAPE, "ds_conv2d_h3: wshift %d out of range", wshift);
  DS_REQUIRE(prenorm == nullptr || load_mode != DS_LOAD_MAXPOOL2, DS_ERR_UNSUPPORTED,
             "ds_conv2d_h3: prenorm cannot be combined with the max-pool load (pooling follows the activation)");
  DS_REQUIRE((reinterpret_cast<uintptr_t>(prenorm) & 15u) == 0, DS_ERR_SHAPE, "ds_conv2d_h3: prenorm must be 16-byte aligned");
  if (B == 0) return DS_OK;
  Conv3hArgs a;
  a.prenorm = prenorm; a.tile_stats = tile_stats; a.circular = circular; a.res1_up = res1_up; a.oy = oy; a.ox = ox;
  a.ntiles_magic40 = 0; a.ncot_magic40 = 0; a.pc_prio = 0; a.pc_skew_mask = 0;      // set by ds_conv3p.hip's launcher when it takes the launch
  {
    static const int early = [] { const char* e = getenv("DS_CONV_TWO_EARLY"); return e ? atoi(e) : 1; }();   // A/B switch of the two-tile kernel's staging shift
    a.two_early = early;
  }
  a.out = out; a.in = in; a.wp = reinterpret_cast<const u32x4*>(w_packed); a.bias = bias; a.shift = shift;
  a.res1 = res1; a.res2 = res2; a.shift_stride = shift_stride;
  a.wshift = wshift; a.in_amax = in_amax; a.out_amax = out_amax;
  a.B = B; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W;
  a.Hin = load_mode == DS_LOAD_MAXPOOL2 ? 2 * H : (load_mode == DS_LOAD_UPSAMPLE2 ? H / 2 : H);
  a.Win = load_mode == DS_LOAD_MAXPOOL2 ? 2 * W : (load_mode == DS_LOAD_UPSAMPLE2 ? W / 2 : W);
  DS_REQUIRE((long long)Cin * a.Hin * a.Win < (1ll << 31), DS_ERR_SHAPE, "ds_conv2d_h3: per-sample input exceeds 2^31 floats");
  // tile geometry: whichever pads the image with fewer pixels (16 x 16 wins for W <= 16, W = 48, ...)
  const long long pad32 = (long long)((W + 31) / 32 * 32) * ((H + 7) / 8 * 8);
  const long long pad16 = (long long)((W + 15) / 16 * 16) * ((H + 15) / 16 * 16);
  const bool w16 = pad16 < pad32;
  const int TW = w16 ? 16 : 32, TH = w16 ? 16 : 8;
  a.tiles_x = (W + TW - 1) / TW; a.tiles_y = (H + TH - 1) / TH;
  a.n_cot = (Cout + COT - 1) / COT;
  a.n_chunks = (Cin + KC - 1) / KC;
  a.tiles_x_magic = a.tiles_x == 1 ? 0u : (unsigned)((1ull << 32) / (unsigned)a.tiles_x) + 1u;   // tiles_x = 1 is special-cased in the kernel
#ifdef DS_STAMP
  {
    // experiment knob of the stamp build: DS_CONV_STAGGER="ticks[,lo,hi]" (10 ns ticks; default range = the second resident
    // workgroup of every CU under breadth-first dispatch: indices [256, 512))
    static const struct Stg { unsigned ticks = 0, lo = 256, hi = 512; Stg() { const char* e = getenv("DS_CONV_STAGGER"); if (e) sscanf(e, "%u,%u,%u", &ticks, &lo, &hi); } } stg;
    a.stagger_ticks = stg.ticks; a.stagger_lo = stg.lo; a.stagger_hi = stg.hi;
    // DS_CONV_NOSTAGE=1|2 (16x16x32 variant): what does staging the input patches cost?  (profiles/r02_stamps_nostage.log)
    static const unsigned nostage = [] { const char* e = getenv("DS_CONV_NOSTAGE"); return e ? (unsigned)atoi(e) : 0u; }();
    a.no_stage = nostage;
  }
#endif
#ifdef DS_STAMP
  a.stamps = g_stamps;
#endif
  hipStream_t s = ds::as_stream(stream);
  if (!w16 && load_mode == DS_LOAD_PLAIN && conv3h_shape16()) {
    // the persistent producer / consumer form (ds_conv3p.hip) takes the launches whose shape is its own
    bool launched = false;
    const int rc = conv3p_try_launch(a, s, &launched);
    if (rc != DS_OK || launched) return rc;
  }
#define DS_L3(M, W) (prenorm ? launch_conv3h<M, W, true>(a, s) : launch_conv3h<M, W, false>(a, s))
  if (w16) {
    if (load_mode == DS_LOAD_PLAIN) return DS_L3(DS_LOAD_PLAIN, true);
    if (load_mode == DS_LOAD_MAXPOOL2) return launch_conv3h<DS_LOAD_MAXPOOL2, true, false>(a, s);
    return DS_L3(DS_LOAD_UPSAMPLE2, true);
  }
  if (load_mode == DS_LOAD_PLAIN) return DS_L3(DS_LOAD_PLAIN, false);
  if (load_mode == DS_LOAD_MAXPOOL2) return launch_conv3h<DS_LOAD_MAXPOOL2, false, false>(a, s);
  return DS_L3(DS_LOAD_UPSAMPLE2, false);
#undef DS_L3
}

size_t ds_conv_images_bytes(int B, int C, int H, int W) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return 0;
  return (size_t)B * ((C + KC - 1) / KC) * 4 * (size_t)(H + 2) * (W + 2) * 16;
}

int ds_conv2d_h3_img(float* out, const void* images, const void* w_packed, int wshift, const float* bias, const float* shift,
                     int shift_stride, const float* res1, const float* res2, int B, int Cin, int Cout, int H, int W,
                     int flags, float* tile_stats, unsigned* out_amax, void* stream) {
  DS_REQUIRE(out && images && w_packed, DS_ERR_NULL, "ds_conv2d_h3_img: NULL pointer");
  DS_REQUIRE((flags & ~DS_RES1_UPSAMPLED) == 0, DS_ERR_UNSUPPORTED, "ds_conv2d_h3_img: flags %d (DS_RES1_UPSAMPLED only)", flags);
  DS_REQUIRE(!(flags & DS_RES1_UPSAMPLED) || (res1 && H % 2 == 0 && W % 2 == 0 && (reinterpret_cast<uintptr_t>(res1) & 7u) == 0),
             DS_ERR_SHAPE, "ds_conv2d_h3_img: RES1_UPSAMPLED needs res1 [B, Cout, H/2, W/2] (8-byte aligned) and even H, W");
  DS_REQUIRE(B >= 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, DS_ERR_SHAPE,
             "ds_conv2d_h3_img: bad shape B=%d Cin=%d Cout=%d H=%d W=%d", B, Cin, Cout, H, W);
  DS_REQUIRE(((Cin + KC - 1) / KC) % 2 == 0, DS_ERR_UNSUPPORTED,
             "ds_conv2d_h3_img: the image-input kernel is the 16x16x32 variant: an even number of 16-channel chunks (Cin = %d)", Cin);
  DS_REQUIRE(shift == nullptr || shift_stride == 0 || shift_stride >= Cout, DS_ERR_SHAPE,
             "ds_conv2d_h3_img: shift_stride %d < Cout %d", shift_stride, Cout);
  DS_REQUIRE((reinterpret_cast<uintptr_t>(w_packed) & 15u) == 0 && (reinterpret_cast<uintptr_t>(images) & 15u) == 0, DS_ERR_SHAPE,
             "ds_conv2d_h3_img: images and w_packed must be 16-byte aligned");
  DS_REQUIRE(wshift >= -40 && wshift <= 40, DS_ERR_SHAPE, "ds_conv2d_h3_img: wshift %d out of range", wshift);
  DS_REQUIRE((long long)4 * (H + 2) * (W + 2) < (1ll << 27), DS_ERR_SHAPE, "ds_conv2d_h3_img: image planes too large");
  if (B == 0) return DS_OK;
  Conv3hArgs a;
  memset(&a, 0, sizeof(a));
  a.out = out; a.in = reinterpret_cast<const float*>(images); a.wp = reinterpret_cast<const u32x4*>(w_packed);
  a.bias = bias; a.shift = shift; a.res1 = res1; a.res2 = res2; a.shift_stride = shift_stride; a.tile_stats = tile_stats;
  a.res1_up = (flags & DS_RES1_UPSAMPLED) ? 1 : 0;
  a.wshift = wshift; a.out_amax = out_amax;
  a.B = B; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W; a.Hin = H; a.Win = W;
  const long long pad32 = (long long)((W + 31) / 32 * 32) * ((H + 7) / 8 * 8);
  const long long pad16 = (long long)((W + 15) / 16 * 16) * ((H + 15) / 16 * 16);
  const bool w16 = pad16 < pad32;
  const int TW = w16 ? 16 : 32, TH = w16 ? 16 : 8;
  a.tiles_x = (W + TW - 1) / TW; a.tiles_y = (H + TH - 1) / TH;
  a.n_cot = (Cout + COT - 1) / COT;
  a.n_chunks = (Cin + KC - 1) / KC;
  a.tiles_x_magic = a.tiles_x == 1 ? 0u : (unsigned)((1ull << 32) / (unsigned)a.tiles_x) + 1u;
#ifdef DS_STAMP
  a.stamps = g_stamps;
#endif
  hipStream_t s = ds::as_stream(stream);
  if (!w16) {
    bool launched = false;
    const int rc = conv3p_try_launch_img(a, s, &launched);
    if (rc != DS_OK || launched) return rc;
  }
  return w16 ? launch_conv3h_w<DS_LOAD_PLAIN, true, false, false, 4, true, true>(a, s)
             : launch_conv3h_w<DS_LOAD_PLAIN, false, false, false, 4, true, true>(a, s);
}

}  // extern "C"
