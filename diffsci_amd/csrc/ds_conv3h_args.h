// Shared by the fp16x3 3x3 convolution kernels (ds_conv3h.hip: one tile per workgroup; ds_conv3p.hip: persistent
// producer / consumer workgroups): tile geometry, LDS image sizes, the launch argument block.
#pragma once
#include "ds_common.h"
#include "ds_conv_epilogue.h"
#include "ds_h3_common.h"

namespace ds_conv3 {

using ds_epi::f32x16;
using ds_epi::f32x4;
using ds_h3::u32x4;
using ds_h3::f16x8;
using ds_h3::split2;
using ds_h3::fast_silu;

constexpr int COT = 64, NT = 256;
constexpr int KC = 16;
// Pixel tile of a workgroup: 8 rows x 32 columns, or 16 x 16 for narrow feature maps (W16) where a
// 32-wide tile would be mostly padding.  The 32 positions of one MFMA B operand are one 32-pixel row
// segment, or two 16-pixel row segments.
template <bool W16> struct Geo {
  static constexpr int TH = W16 ? 16 : 8, TW = W16 ? 16 : 32;
  static constexpr int PH = TH + 2, PW = TW + 2, NPOS = PH * PW;     // 340 / 324
  static constexpr int XITEMS = 2 * NPOS;                            // (h, position) staging items
  static constexpr int XBUF_VEC = 2 * 2 * NPOS;                      // 16-byte vectors per X buffer
};
constexpr int XBUF_VEC = Geo<false>::XBUF_VEC;              // LDS is sized for the larger geometry: 1360
// 16x16x32 variant: the two 8-channel halves of a piece sit a multiple of 256 B apart (its operand reads put both
// halves into one ds_read_b128 lane group): 12 pad vectors between them, 24 per X buffer
constexpr int HPAD16 = 12;
constexpr int XBUF_VEC16 = XBUF_VEC + 2 * HPAD16;
constexpr int WSLAB_VEC = 2 * 3 * 2 * COT;                  // 16-byte vectors per (chunk, ky) slab: 768
constexpr int WPIECES = WSLAB_VEC / 64;                     // LDS-DMA wave-instructions per slab: 12
constexpr int STAGE_BYTES = (2 * XBUF_VEC + 3 * WSLAB_VEC) * 16;   // 80,384
constexpr int LDS_BYTES = STAGE_BYTES + 128 * 4;              // + bias / shift of the channel tile = 80,896
constexpr int STAGE_BYTES16 = (2 * XBUF_VEC16 + 3 * WSLAB_VEC) * 16;   // 81,152
constexpr int LDS_BYTES16 = STAGE_BYTES16 + 128 * 4;                   // 81,664: two workgroups still fit 160 KiB
static_assert(2 * LDS_BYTES16 <= 160 * 1024, "two workgroups per CU");
// TWO: staging = two X buffers + a 3-slot ring of two slabs (118,016 B); the epilogue's eight 16 KiB wave tiles (131,072 B) reach
// beyond it, so the bias / shift rows sit behind THEM
constexpr int TWO_STAGE_BYTES = (2 * XBUF_VEC16 + 3 * 2 * WSLAB_VEC) * 16;
constexpr int TWO_EPI_BYTES = 8 * 64 * 2 * 32 * 4;
static_assert(TWO_EPI_BYTES >= TWO_STAGE_BYTES, "bias / shift rows behind the larger of the two");
constexpr int LDS_BYTES_TWO = TWO_EPI_BYTES + 2 * 128 * 4;          // 132,096: one workgroup per CU

struct Conv3hArgs {
  float* out;
  const float* in;
  const u32x4* wp;
  const float* bias;
  const float* shift;
  const float* res1;
  const float* res2;
  const float* prenorm;   // [B][ceil16(Cin)][4] = (M, A, C, -), zero rows past Cin, or NULL: the loader applies SiLU((x - M)*A + C)
  float* tile_stats;      // see ds_conv_epilogue.h, or NULL
  const unsigned* in_amax;   // per-sample max |input| (float bits) -> the loader's activation exponent (ds_conv_epilogue.h), or NULL
  unsigned* out_amax;        // per-sample max |output| slots, merged with atomicMax, or NULL
  int wshift;                // the packed weights carry 2^wshift
  int shift_stride;
  int res1_up;            // res1 is at half resolution (see ds_conv_epilogue.h)
  int circular;           // periodic padding in both dimensions (CircularConv2d, commonlayers.py:918-971)
  int oy, ox;             // tap-origin offset: the 3x3 window is centred at (y + oy, x + ox) -- sub-kernels of larger kernels
  int B, Cin, Cout, H, W, Hin, Win;
  int tiles_x, tiles_y, n_cot, n_chunks;
  unsigned tiles_x_magic;   // floor(2^32 / tiles_x) + 1
  // ds_conv3p.hip only: floor(2^40 / d) + 1 for d = tiles_x * tiles_y and d = n_cot -- (n * magic) >> 40 is n / d exactly for n < 2^22,
  // d < 2^18 (the persistent kernel decodes an item index several times per tile: hardware has no integer divide)
  unsigned long long ntiles_magic40, ncot_magic40;
  int pc_prio;              // ds_conv3p.hip only: s_setprio level of the producer waves (DS_CONV_PC_PRIO, A/B runs)
  int pc_skew_mask;         // ds_conv3p.hip only: mask of the start-up stagger (DS_CONV_PC_SKEW)
  int two_early;            // ds_conv3h.hip, two channel tiles per workgroup: waves 0-3 stage the next patch before the step's matrix instructions (DS_CONV_TWO_EARLY)
#ifdef DS_STAMP
  unsigned long long* stamps;   // diagnostic build only (tools/conv3h_stamp.hip)
  unsigned stagger_lo, stagger_hi, stagger_ticks;   // experiment: workgroups with dispatch index in [lo, hi) start `ticks` x 10 ns late
  unsigned no_stage;                                // experiment (wrong results): 1 = the patches after the first are neither fetched nor split / stored, 2 = not fetched
#endif
};


// ds_conv3p.hip: the persistent producer / consumer form of the fused-loader launches (full tiles, an even number of 16-channel
// chunks, enough tiles to give every CU several).  Returns DS_OK and sets *launched when it took the launch; leaves *launched
// false (and launches nothing) when the shape is not its own.
int conv3p_try_launch(const Conv3hArgs& a, hipStream_t s, bool* launched);
// ... the image-input form (ds_conv2d_h3_img; full 8 x 32 tiles, Cin and Cout multiples of 64, no half-resolution residual)
int conv3p_try_launch_img(const Conv3hArgs& a, hipStream_t s, bool* launched);

}  // namespace ds_conv3
