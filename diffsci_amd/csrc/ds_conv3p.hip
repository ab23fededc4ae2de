// The fused-loader 3x3 convolution (fp16x3 on v_mfma_f32_16x16x32_f16) as PERSISTENT producer / consumer workgroups.
//
// Why (profiles/r03_mfma_busy_per_launch.csv): in ds_conv3h.hip every wave does everything -- fetches and
// activates the next patch, splits it, multiplies, transposes and stores its tile -- in program order, one wave per SIMD and
// workgroup.  At 64 and 128 channels a tile is only 4-8 chunks long, so the cold prologue, the vector work of the fused
// norm + SiLU loader and the store epilogue are as long as the matrix phase and two resident workgroups overlap them poorly:
// the matrix pipe is busy a third of the time with the SIMDs issuing a third of the time.  Here the roles are split:
//   waves 0-3  CONSUMERS  ds_read_b128 + MFMA only; each owns two of the tile's eight rows x 64 channels (64 accumulator registers),
//                         and at the end of a tile parks its accumulators (times 2^-shift, plus bias + time shift) in its own 16 KiB of LDS;
//   waves 4-7  PRODUCERS  everything else, for tiles ahead of and behind the consumers: patch loads (two chunks in flight in
//                         registers), norm + SiLU, fp16 hi / lo split and the LDS images; the weight slabs by LDS-DMA; and the
//                         PREVIOUS tile's store phase (residual loads a step ahead, 16-byte stores, tile statistics, output maxima)
//                         while the consumers already multiply the next tile.
// Consumer w and producer w + 4 share a SIMD (waves go to SIMDs cyclically), so every SIMD holds one matrix stream and one vector /
// memory stream: the two pipes issue side by side (MI355X_MICROARCH.md, Wave scheduling).  One workgroup per CU walks a list of
// (channel tile, pixel tile, sample) items -- the XCD-aware order of ds_conv3h.hip, dealt over the CUs of an XCD -- so nothing is
// cold after the first item.  Synchronisation is ONE workgroup barrier per step (chunk, ky) -- the same program point for both
// roles, so the counts cannot diverge -- with a counted vmcnt on the producer side: the weight slab's DMA is the oldest vector-memory
// operation of its step and only the operations issued behind it may stay in flight across the barrier (patch loads, residual
// loads, stores).  LDS: two X images, a 4-slot weight ring, 4 x 16 KiB accumulator tiles, statistics partials, two bias / shift
// rows = 160,032 B.
//
// Shapes it takes (the launcher falls back to ds_conv3h.hip otherwise): plain load, 8 x 32 pixel tiles that tile the plane
// exactly, Cout and Cin multiples of 64, and enough items to give every CU several.
#include "ds_conv3h_args.h"

#include <atomic>
#include <cstdlib>
#include <type_traits>

namespace ds_conv3 {
namespace {

constexpr int P_WOFF = 2 * XBUF_VEC16 * 16;                      // 44,288
constexpr int P_WSLOTS = 4;                                       // weight ring: a slab is fetched THREE steps ahead of its step
constexpr int P_OUTOFF = P_WOFF + P_WSLOTS * WSLAB_VEC * 16;     // 93,440
constexpr int P_BSOFF = P_OUTOFF + 4 * 16384;                    // 158,976
constexpr int P_BS_FLOATS = 132;                                 // [64 bias][64 shift][2^-(wshift + k), pad]
constexpr int P_LDS = P_BSOFF + 2 * P_BS_FLOATS * 4;             // 160,032
static_assert(P_LDS <= 160 * 1024, "one workgroup per CU");

#ifdef DS_STAMP
// diagnostic build: wave 0 (consumer) and wave 4 (producer) of every workgroup stamp their arrival at each barrier of the first items
#define PSTAMP_SLOTS 64
#define PSTAMP(slot) do { if (a.stamps && (wv == 0 || wv == 4) && lane == 0 && (slot) < PSTAMP_SLOTS) a.stamps[((size_t)blockIdx.x * 2 + (wv >> 2)) * PSTAMP_SLOTS + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PSTAMP(slot) do {} while (0)
#endif

// Wait until all but the `young` youngest vector-memory operations of this wave have completed (rounded down to a multiple of four:
// waiting for more than asked is always safe), and for every LDS operation; then the workgroup barrier.
__device__ __forceinline__ void bar_counted(int young) {
  const int y = __builtin_amdgcn_readfirstlane(young) >> 2;
  switch (y) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(28)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(36)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(40)" ::: "memory"); break;
    case 11: asm volatile("s_waitcnt vmcnt(44)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(48)" ::: "memory"); break;
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

struct Item { int cot, b, y0, x0, tile; };

struct Packed { u32x4 h[3], l[3]; };
struct ResRegs { f32x4 r1[4], r2[4]; };
struct Frag { f16x8 a[2][4], b[2][4]; };             // one K = 32 group's operands: [piece][16-channel tile], [piece][16-position tile]

// The weight slabs' LDS-DMA as inline assembly: global_load_lds is FLAT-encoded and touches both memories, which hipcc's wait-count
// pass books as a
// "pending flat" operation -- and while one is pending it forces vmcnt(0) on EVERY wait for a loaded register (found in the
// ISA: each norm + SiLU block waited for the DMA just issued and for all 24 loads of the next chunk).  gsrc: the wave's 1 KiB
// piece (wave-uniform), lane_bytes = 16 * lane, lds: byte address of the piece's destination.
__device__ __forceinline__ void lds_dma16(const void* gsrc, unsigned lane_bytes, unsigned lds) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(lds), "v"(lane_bytes), "s"(gsrc) : "memory", "m0");
}
__device__ __forceinline__ unsigned lds_address(const void* p) {
  return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)p;
}
// a read-only word written by an EARLIER kernel, through the scalar cache (s_load: no vector-memory counter involved)
__device__ __forceinline__ unsigned const_u32(const void* p, size_t i) {
  typedef const __attribute__((address_space(4))) unsigned* cptr;
  return ((cptr)p)[i];
}

// NRES: residual tensors added in the store phase (0, 1 = res1, 2 = res1 and res2) -- a template parameter so that the residual
// loads are unconditional instructions: hipcc counts only those when it sizes the wait for a loaded register, and a wait sized
// vmcnt(0) in the store phase also waits for the previous batch's STORES to complete (about 8,000 cycles, stamped).
template <bool PRE, bool CIRC, int NRES>
__global__ __launch_bounds__(512, 2) void k_conv3p(const Conv3hArgs a) {
  constexpr int PW = Geo<false>::PW, NPOS = Geo<false>::NPOS;
  constexpr int HS = NPOS + HPAD16, PS = 2 * NPOS + HPAD16, XBV = XBUF_VEC16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32x4* Xs = reinterpret_cast<u32x4*>(smem);
  u32x4* Ws = reinterpret_cast<u32x4*>(smem + P_WOFF);
  float* OUT = reinterpret_cast<float*>(smem + P_OUTOFF);
  float* BS = reinterpret_cast<float*>(smem + P_BSOFF);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool consumer = wv < 4;
  const int rw = wv & 3;                                   // consumer: its row pair; producer: the consumer whose tile it stores
  const int ptid = tid & 255;

  // ---- this workgroup's items: XCD x owns one contiguous run of the logical order (channel tile fastest, then pixel tile, then
  //      sample), dealt round-robin over the Q workgroups of the XCD, so the workgroups of an XCD work on neighbouring tiles ----
  const unsigned ntiles = (unsigned)(a.tiles_x * a.tiles_y);
  const unsigned total = (unsigned)a.n_cot * ntiles * (unsigned)a.B;
  const unsigned Q = gridDim.x >> 3, xcd = blockIdx.x & 7u, q = blockIdx.x >> 3;
  const unsigned per = total >> 3, rem = total & 7u;
  const unsigned first = xcd * per + (xcd < rem ? xcd : rem), cnt = per + (xcd < rem ? 1u : 0u);
  const int n_items = q < cnt ? (int)((cnt - q + Q - 1) / Q) : 0;
  if (n_items == 0) return;
  auto item_of = [&](int k) __attribute__((always_inline)) {
    const unsigned logical = first + q + Q * (unsigned)k;
    Item it;
    it.cot = (int)(logical % (unsigned)a.n_cot);
    const unsigned rest = logical / (unsigned)a.n_cot;
    it.tile = (int)(rest % ntiles);
    it.b = (int)(rest / ntiles);
    const int ty = a.tiles_x == 1 ? it.tile : (int)__umulhi((unsigned)it.tile, a.tiles_x_magic);
    const int tx = it.tile - ty * a.tiles_x;
    it.y0 = 8 * ty; it.x0 = 32 * tx;
    return it;
  };
  const int n_chunks = a.n_chunks, n_steps = 3 * n_chunks;
  const int HW = a.H * a.W;

  // =========================== consumer ===========================
  f32x4 acc[4][4];
  const int i16 = lane & 15, h16 = (lane >> 4) & 1, g16 = lane >> 4;
  const bool tapB = lane >= 32;
  const int wlane = h16 * COT + i16;
  int xlane[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) xlane[n] = h16 * HS + (2 * rw + (n >> 1)) * PW + 16 * (n & 1) + i16;
  // operands of one K = 32 group of the 16x16x32 schedule: taps (slot, ky, kx, X buffer) A and B (ds_conv3h.hip, S16)
  auto load_pair = [&](Frag& f, int slotA, int kyA, int kxA, int xbA, int slotB, int kyB, int kxB, int xbB) __attribute__((always_inline)) {
    const int wofs = (tapB ? slotB * WSLAB_VEC + kxB * 2 * COT : slotA * WSLAB_VEC + kxA * 2 * COT) + wlane;
    const int xofs = tapB ? xbB * XBV + kyB * PW + kxB : xbA * XBV + kyA * PW + kxA;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
      for (int m = 0; m < 4; ++m) f.a[p][m] = *reinterpret_cast<const f16x8*>(&Ws[wofs + p * 6 * COT + 16 * m]);
#pragma unroll
      for (int n = 0; n < 4; ++n) f.b[p][n] = *reinterpret_cast<const f16x8*>(&Xs[xofs + p * PS + xlane[n]]);
    }
  };
  auto mma_pair = [&](const Frag& f) __attribute__((always_inline)) {                // lo*hi, hi*lo, hi*hi
    constexpr int PA[3] = {1, 0, 0};
    constexpr int PB[3] = {0, 1, 0};
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.a[PA[t]][m], f.b[PB[t]][n], acc[m][n], 0, 0, 0);
  };
  // the NEXT group's 16 ds_read_b128 ride between this group's first matrix instructions (one wave per SIMD multiplies: nobody
  // else hides its LDS round trips)
  // ... between the matrix instructions of this group's SECOND product: its low weight pieces and (after that product) low input
  // pieces are dead by then, so two whole operand sets are never live together
  auto reads_between = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
  };
  // end of a tile: acc * 2^-(wshift + k) + (bias + shift) into the wave's LDS tile [co][row][32 px] (store_tile16's phase 1), acc = 0
  auto park_tile = [&](int par) __attribute__((always_inline)) {
    const float* bs = BS + par * P_BS_FLOATS;
    float* tile = OUT + rw * 4096;
    // every read of the bias / shift row BEFORE the first write of the tile: the compiler cannot tell the two LDS regions apart,
    // and one read per channel between the writes was sixteen dependent LDS round trips (2,700 cycles per tile, stamped)
    const float unscale = bs[128];
    float bsv[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int co = 16 * m + 4 * g16 + e;
        bsv[m][e] = bs[co] + bs[64 + co];
      }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int co = 16 * m + 4 * g16 + e;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          tile[(co * 2 + (n >> 1)) * 32 + 16 * (n & 1) + i16] = __builtin_fmaf(acc[m][n][e], unscale, bsv[m][e]);
          acc[m][n][e] = 0.f;
        }
      }
  };

  // =========================== producer ===========================
  // staging items of a thread (ds_conv3h.hip's four-wave plan over the 256 producer threads): items 0 / 1 = position ptid of
  // channel half 0 / 1, item 2 = the patch's tail (positions 256 ..), half (producer wave / 2)
  const int tail_h = (rw >> 1) & 1;
  auto item_h = [&](int i) __attribute__((always_inline)) { return i == 0 ? 0 : (i == 1 ? 1 : tail_h); };
  int xlds[3], xrow[3], xcol[3];
  const bool live2 = NT + (ptid & (NT / 2 - 1)) < NPOS;     // the thread's tail item exists
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int pos = i < 2 ? ptid : NT + (ptid & (NT / 2 - 1));
    xrow[i] = pos / PW;
    xcol[i] = pos - xrow[i] * PW;
    xlds[i] = item_h(i) * HS + pos;
  }
  // fetch cursor: the next chunk to load.  The loads are UNCONDITIONAL (past the end of the list the last item's first chunk is
  // loaded again and dropped): a conditional load between a load and its use makes hipcc wait for vmcnt(0) at the use -- the
  // full HBM latency of the loads just issued, measured at 3000 cycles per chunk in the first version of this kernel.
  int f_k = 0, f_chunk = 0, f_b = 0;
  int xoff[3] = {0, 0, 0};
  unsigned f_xvalid = 0;
  float f_in_scale = 1.f;
  auto plan_fetch = [&](const Item& it) __attribute__((always_inline)) {
    f_xvalid = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      int gy = it.y0 + xrow[i] - 1 + a.oy, gx = it.x0 + xcol[i] - 1 + a.ox;
      if (CIRC) {
        gy = gy < 0 ? gy + a.H : (gy >= a.H ? gy - a.H : gy);
        gx = gx < 0 ? gx + a.W : (gx >= a.W ? gx - a.W : gx);
        gy = gy >= a.H ? a.H - 1 : gy;
        gx = gx >= a.W ? a.W - 1 : gx;
      }
      const bool ok = (i < 2 || live2) && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
      xoff[i] = ok ? gy * a.W + gx : 0;
      if (ok) f_xvalid |= 1u << i;
    }
    f_b = it.b;
    if constexpr (!PRE) f_in_scale = ds_epi::act_scale_of(a.in_amax ? const_u32(a.in_amax, it.b) : 0u, a.wshift).in_scale;
  };
  // tag of a fetched chunk: bit 31 = it exists, bits 0-2 = which staging items lie inside the image; trow = its first table row
  const float* f_src = a.in;
  auto fetch_begin = [&](unsigned& tag, int& trow, float& tscale) __attribute__((always_inline)) {
    const bool valid = f_k < n_items;
    if (valid && f_chunk == 0) plan_fetch(item_of(f_k));
    tag = valid ? (0x80000000u | f_xvalid) : 0u;
    trow = (f_b * n_chunks + f_chunk) * KC;
    tscale = f_in_scale;
    f_src = a.in + ((size_t)f_b * a.Cin + (size_t)f_chunk * KC) * HW;
    if (valid && ++f_chunk == n_chunks) { f_chunk = 0; ++f_k; }
  };
  auto fetch_item = [&](float (&xr)[3][8], int i) __attribute__((always_inline)) {
    const float* p0 = f_src + xoff[i] + (size_t)(8 * item_h(i)) * HW;
#pragma unroll
    for (int k = 0; k < 8; ++k) xr[i][k] = p0[(size_t)k * HW];
  };
  auto fetch = [&](float (&xr)[3][8], unsigned& tag, int& trow, float& tscale) __attribute__((always_inline)) {
    fetch_begin(tag, trow, tscale);
#pragma unroll
    for (int i = 0; i < 3; ++i) fetch_item(xr, i);
    return 24;
  };
  // [norm + SiLU,] fp16 hi / lo split of one staging item of a fetched chunk, in registers
  auto activate_item = [&](float (&xr)[3][8], unsigned tag, int trow, float tscale, Packed& pk, int i) __attribute__((always_inline)) {
    if constexpr (PRE) {
      typedef const __attribute__((address_space(4))) f32x4* cptr;
      cptr pp = (cptr)(reinterpret_cast<const f32x4*>(a.prenorm)) + trow;
      const int h = item_h(i);
      f32x4 p[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) p[k] = pp[8 * h + k];
      const float inv = p[0][3] == 0.f ? 1.0f : p[0][3];
#pragma unroll
      for (int k = 0; k < 8; ++k) xr[i][k] = ds_h3::fast_silu_scaled((xr[i][k] - p[k][0]) * p[k][1] + p[k][2], inv);
    }
    const bool item_ok = (tag >> i) & 1u;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float v0 = item_ok ? xr[i][2 * k] : 0.f;
      float v1 = item_ok ? xr[i][2 * k + 1] : 0.f;
      if constexpr (!PRE) { v0 *= tscale; v1 *= tscale; }
      unsigned ph, pl;
      split2(v0, v1, ph, pl);
      pk.h[i][k] = ph; pk.l[i][k] = pl;
    }
  };
  auto activate = [&](float (&xr)[3][8], unsigned tag, int trow, float tscale, Packed& pk) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 3; ++i) activate_item(xr, tag, trow, tscale, pk, i);
  };
  // the loads of one chunk and the vector work of another, item by item: the loads' issue is paced by the memory pipeline (a CU's
  // share of HBM), and a wave stuck behind 24 of them in a row activates nothing meanwhile
  auto fetch_while_activating = [&](float (&xn)[3][8], unsigned& tagn, int& trown, float& tscn,
                                    float (&xo)[3][8], unsigned tago, int trowo, float tsco, Packed& pk) __attribute__((always_inline)) {
    fetch_begin(tagn, trown, tscn);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      fetch_item(xn, i);
      __builtin_amdgcn_sched_barrier(0);
      activate_item(xo, tago, trowo, tsco, pk, i);
      __builtin_amdgcn_sched_barrier(0);
    }
    return 24;
  };
  auto store_x = [&](const Packed& pk, unsigned tag, int buf) __attribute__((always_inline)) {
    if (!(tag >> 31)) return;
    u32x4* xb = Xs + buf * XBV;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      if (i < 2 || live2) {
        xb[xlds[i]] = pk.h[i];
        xb[PS + xlds[i]] = pk.l[i];
      }
    }
  };
  // weight cursor: the next slab to DMA.  Unconditional as well (past the end: the last item's first slab into a slot nobody reads).
  int w_k = 0, w_step = 0, w_cot = 0;
  auto wdma = [&](int slot) __attribute__((always_inline)) {                       // slot = (slab index) & 3
    const bool valid = w_k < n_items;
    if (valid && w_step == 0) w_cot = item_of(w_k).cot;
    const u32x4* src = a.wp + ((size_t)w_cot * n_steps + w_step) * WSLAB_VEC;
    u32x4* dst = Ws + slot * WSLAB_VEC;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int k = rw + 4 * i;                       // wave-uniform piece
      lds_dma16(src + 64 * k, 16u * (unsigned)lane, lds_address(dst + 64 * k));
    }
    if (valid && ++w_step == n_steps) { w_step = 0; ++w_k; }
    return 0;                                          // the DMA is what the barrier waits for: never among the `young`
  };
  // bias / shift row of an item: one register per thread (threads 0-63: bias, 64-127: shift), committed a step later together
  // with the item's 2^-(wshift + k) (thread 128; read through the scalar cache, so nothing waits for vector memory here)
  auto bs_fetch = [&](const Item& it) __attribute__((always_inline)) {
    float v = 0.f;
    const int co = it.cot * COT + (ptid & 63);
    if (ptid < 64) v = a.bias ? a.bias[co] : 0.f;
    else if (ptid < 128) v = a.shift ? a.shift[(size_t)it.b * a.shift_stride + co] : 0.f;
    return v;
  };
  auto unscale_of = [&](const Item& it) __attribute__((always_inline)) {
    if constexpr (PRE) {
      const float inv = __builtin_bit_cast(float, const_u32(a.prenorm, (size_t)it.b * n_chunks * KC * 4 + 3));
      return ds_epi::unscale_from_inv(inv == 0.f ? 1.0f : inv, a.wshift);
    } else {
      return ds_epi::act_scale_of(a.in_amax ? const_u32(a.in_amax, it.b) : 0u, a.wshift).unscale;
    }
  };
  auto bs_commit = [&](float v, float unscale, int par) __attribute__((always_inline)) {
    if (ptid < 128) BS[par * P_BS_FLOATS + ptid] = v;
    else if (ptid == 128) BS[par * P_BS_FLOATS + 128] = unscale;
  };

  // store phase of the previous item (producer rw stores consumer rw's tile): four batches of four 16-byte store instructions
  // (ds_conv_epilogue.h: store_tile_rows' aligned path, full tiles).  The wave's statistics partials [64 channels][4] go to the
  // head of its own tile, behind the batch that has just read it.
  int s_b = 0, s_cot = 0, s_tile = 0;
  size_t s_idx = 0;                                  // the lane's first output element
  const size_t s_step = 4 * (size_t)HW;
  float s_amax = 0.f;
  const int p4 = 4 * (lane & 7);
  const bool stats = a.tile_stats != nullptr, want_amax = a.out_amax != nullptr;
  constexpr int nres = 4 * NRES;
  auto plan_store = [&](const Item& it) __attribute__((always_inline)) {
    s_b = it.b; s_cot = it.cot; s_tile = it.tile;
    const int gy = it.y0 + 2 * rw + ((lane >> 3) & 1), gx = it.x0 + p4;
    const size_t ch = (size_t)it.b * a.Cout + it.cot * COT + (lane >> 4);
    s_idx = ch * HW + (size_t)gy * a.W + gx;
    s_amax = 0.f;
  };
  // residual vectors of one batch.  (No half-resolution res1 here: with both forms in one function the two loads share destination
  // registers, and hipcc then waits for vmcnt(0) in front of every residual load -- the launcher keeps such launches on ds_conv3h.hip.)
  auto res_prefetch = [&](ResRegs& R, int half) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int j = half * 4 + k;
      if constexpr (NRES >= 1) R.r1[k] = *reinterpret_cast<const f32x4*>(a.res1 + s_idx + (size_t)j * s_step);
      if constexpr (NRES >= 2) R.r2[k] = *reinterpret_cast<const f32x4*>(a.res2 + s_idx + (size_t)j * s_step);
    }
    return nres;
  };
  auto store_batch = [&](const ResRegs& R, int half) __attribute__((always_inline)) {
    float* tile = OUT + rw * 4096;
    f32x4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int seg = (half * 4 + k) * 8 + (lane >> 3);
      v[k] = *reinterpret_cast<const f32x4*>(&tile[seg * 32 + p4]);
    }
    if constexpr (NRES >= 1) {
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = v[k] + R.r1[k];
    }
    if constexpr (NRES >= 2) {
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = v[k] + R.r2[k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) *reinterpret_cast<f32x4*>(a.out + s_idx + (size_t)(half * 4 + k) * s_step) = v[k];
    if (want_amax) {
#pragma unroll
      for (int k = 0; k < 4; ++k) s_amax = fmaxf(s_amax, ds_epi::abs_max4(v[k]));
    }
    if (stats) {
      f32x4 o[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float K = ds_epi::row16_first(v[k].x);
        const f32x4 d = v[k] - K;
        const float sv = ds_epi::row16_sum((d.x + d.y) + (d.z + d.w));
        const float qv = ds_epi::row16_sum((d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w));
        o[k] = f32x4{K, sv, qv, 64.f};                                   // the wave's 2 x 32 pixels of the channel
      }
      // rows 16 half .. +15 of [64][4] at the tile's head: inside batch 0's region, which every lane of this wave has read by now
      if ((lane & 15) == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) *reinterpret_cast<f32x4*>(&tile[4 * (4 * (half * 4 + k) + (lane >> 4))]) = o[k];
      }
    }
    return 4;
  };
  auto commit_amax_asm = [&]() __attribute__((always_inline)) {        // ds_epi::commit_amax with the atomic as assembly
    float m = s_amax;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if (lane == 0) atomicMax(a.out_amax + s_b, __builtin_bit_cast(unsigned, m));
  };
  auto store_stats = [&]() __attribute__((always_inline)) {            // combine the four waves' partials of every channel (store_tile_stats<2>)
    const float* tiles = OUT;
    f32x4 pw[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) pw[w] = *reinterpret_cast<const f32x4*>(&tiles[w * 4096 + 4 * lane]);
    const float K = pw[0][0];
    float S = 0.f, Qs = 0.f, n = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float dk = pw[w][3] > 0.f ? pw[w][0] - K : 0.f;
      S += pw[w][1] + pw[w][3] * dk;
      Qs += pw[w][2] + 2.f * dk * pw[w][1] + pw[w][3] * dk * dk;
      n += pw[w][3];
    }
    *reinterpret_cast<f32x4*>(a.tile_stats + (((size_t)s_b * a.Cout + s_cot * COT + lane) * ntiles + s_tile) * 4) = f32x4{K, S, Qs, n};
  };

  float xrA[3][8], xrB[3][8];
  unsigned tagA = 0, tagB = 0, tagP = 0;
  int trowA = 0, trowB = 0;
  float tscA = 1.f, tscB = 1.f;
  Packed pk;
  ResRegs RA, RB;
  float bsv = 0.f;
  Frag F0, F1;

  // Both roles run the SAME loop nest (one generic lambda, instantiated per role): every barrier is one program point of that
  // nest, so the two roles execute the same number of barriers by construction, while each role's registers are live only in its
  // own instantiation (as one loop with role branches inside, the allocator had to keep both roles' state alive at once: 256
  // VGPRs and 700 spilled).
  auto run = [&](auto role_tag) __attribute__((always_inline)) {
    constexpr bool CONS = decltype(role_tag)::value;
    // Counted barrier wait of the producers.  A step's weight DMA (slab p + 3) is the FIRST vector-memory operation of step p and
    // must have landed by the END of step p + 1: everything issued behind it in step p (`prev`), step p + 1's own DMA (3) and
    // everything else of step p + 1 (`young`) may stay in flight.  Loads, LDS-DMA and stores complete in issue order on gfx9
    // (hipcc's own counted waits rely on it), so "all but the N youngest" names exactly the operations in front of them.
    int young = 0, prev = 0;
    auto barrier = [&]() __attribute__((always_inline)) {
      if constexpr (CONS) bar_counted(0);
      else { bar_counted(prev + 3 + young); prev = young; young = 0; }
    };
    int stamp = 1;
    (void)stamp;
    // One chunk pair (E, O): six steps, one barrier each.  Consumer: nine K = 32 groups P0 .. P8, the operands of group k + 1
    // fetched (into the other fragment set) while group k multiplies; nine is odd, so the two sets swap roles from one chunk pair
    // to the next and the loop body below is two chunk pairs = 12 steps = three turns of the 4-slot weight ring (BASE = 0 / 6:
    // the step position of (E,0); slab of position p sits in slot p & 3).  Producer duties per step (chunk numbers relative to E):
    //   (E,0)  DMA slab +3 | residuals of batch 0 | fetch chunk E+2 -> B  interleaved with  activate + split chunk O (A)
    //   (E,1)  DMA         | store chunk O -> X1  | batch 0, residuals of batches 1, 2
    //   (E,2)  DMA         | batches 1, 2, residuals of batch 3
    //   (O,0)  DMA         | fetch chunk O+2 -> A | batch 3, output maxima | next item's bias / shift loads
    //   (O,1)  DMA         | activate + split chunk E+2 (B), store -> X0  | statistics combine
    //   (O,2)  DMA         | commit the next item's bias / shift row
    // (batches = the PREVIOUS item's store phase, on the item's first chunk pair only.)  What a group reads is published one
    // barrier before the group that precedes it starts, because its operands are fetched during that predecessor.
    auto chunk_pair = [&](auto base_tag, Frag& Fa, Frag& Fb, bool head, bool last_of_item, int it) __attribute__((always_inline)) {
      constexpr int B = decltype(base_tag)::value;
      constexpr int E0 = (B + 0) & 3, E1 = (B + 1) & 3, E2 = (B + 2) & 3;          // ring slots of chunk E's slabs
      constexpr int O0 = (B + 3) & 3, O1 = (B + 4) & 3, O2 = (B + 5) & 3;          // ... of chunk O's
      constexpr int N0 = (B + 6) & 3;                                              // ... of the next chunk pair's first
      // ---------------- step (E, 0) ----------------
      if constexpr (CONS) {
        if (head) park_tile((it - 1) & 1);
        load_pair(Fb, E0, 0, 2, 0, E1, 1, 0, 0); mma_pair(Fa); reads_between();         // P0 | fetch P1
        load_pair(Fa, E1, 1, 1, 0, E1, 1, 2, 0); mma_pair(Fb); reads_between();         // P1 | fetch P2
      } else {
        wdma((B + 3) & 3);
        __builtin_amdgcn_sched_barrier(0);
        if (head) { plan_store(item_of(it - 1)); young += res_prefetch(RA, 0); }
        young += fetch_while_activating(xrB, tagB, trowB, tscB, xrA, tagA, trowA, tscA, pk);   // chunk E + 2 | chunk O
        tagP = tagA;
      }
      PSTAMP(stamp); ++stamp;
      barrier();
      // ---------------- step (E, 1) ----------------
      if constexpr (CONS) {
        load_pair(Fb, E2, 2, 0, 0, E2, 2, 1, 0); mma_pair(Fa); reads_between();         // P2 | fetch P3
      } else {
        wdma((B + 4) & 3);
        __builtin_amdgcn_sched_barrier(0);
        store_x(pk, tagP, 1);
        if (head) {
          young += store_batch(RA, 0);
          young += res_prefetch(RB, 1);
          young += res_prefetch(RA, 2);
        }
      }
      PSTAMP(stamp); ++stamp;
      barrier();
      // ---------------- step (E, 2) ----------------
      if constexpr (CONS) {
        load_pair(Fa, E2, 2, 2, 0, O0, 0, 0, 1); mma_pair(Fb); reads_between();         // P3 | fetch P4
        load_pair(Fb, O0, 0, 1, 1, O0, 0, 2, 1); mma_pair(Fa); reads_between();         // P4 | fetch P5
      } else {
        wdma((B + 5) & 3);
        __builtin_amdgcn_sched_barrier(0);
        if (head) {
          young += store_batch(RB, 1);
          young += res_prefetch(RB, 3);
          young += store_batch(RA, 2);
        }
      }
      PSTAMP(stamp); ++stamp;
      barrier();
      // ---------------- step (O, 0) ----------------
      if constexpr (CONS) {
        load_pair(Fa, O1, 1, 0, 1, O1, 1, 1, 1); mma_pair(Fb); reads_between();         // P5 | fetch P6
      } else {
        wdma((B + 6) & 3);
        __builtin_amdgcn_sched_barrier(0);
        young += fetch(xrA, tagA, trowA, tscA);       // chunk O + 2
        if (head) {
          young += store_batch(RB, 3);
          if (want_amax) commit_amax_asm();
        }
        if (last_of_item && it + 1 < n_items) bsv = bs_fetch(item_of(it + 1));
      }
      PSTAMP(stamp); ++stamp;
      barrier();
      // ---------------- step (O, 1) ----------------
      if constexpr (CONS) {
        load_pair(Fb, O1, 1, 2, 1, O2, 2, 0, 1); mma_pair(Fa); reads_between();         // P6 | fetch P7
        load_pair(Fa, O2, 2, 1, 1, O2, 2, 2, 1); mma_pair(Fb); reads_between();         // P7 | fetch P8
      } else {
        wdma((B + 7) & 3);
        __builtin_amdgcn_sched_barrier(0);
        activate(xrB, tagB, trowB, tscB, pk);         // chunk E + 2
        store_x(pk, tagB, 0);
        if (head && stats && wv == 4) store_stats();
      }
      PSTAMP(stamp); ++stamp;
      barrier();
      // ---------------- step (O, 2) ----------------
      if constexpr (CONS) {
        load_pair(Fb, N0, 0, 0, 0, N0, 0, 1, 0); mma_pair(Fa); reads_between();         // P8 | fetch the next chunk pair's P0
      } else {
        wdma((B + 8) & 3);
        __builtin_amdgcn_sched_barrier(0);
        if (last_of_item && it + 1 < n_items) bs_commit(bsv, unscale_of(item_of(it + 1)), (it + 1) & 1);
      }
      PSTAMP(stamp); ++stamp;
      barrier();
    };

    // ---- start-up stagger: every workgroup runs the same schedule on the same amount of work, so unskewed they would all fetch,
    //      multiply and store in phase and the memory system would see the chip's whole demand in bursts; spread them over
    //      about one chunk pair's time ----
    {
      const unsigned skew = (blockIdx.x * 11u) & 31u;
      for (unsigned i = 0; i < skew; ++i) __builtin_amdgcn_s_sleep(8);    // 8 x 64 clocks each
    }
    // ---- prologue: slabs 0, 1, 2, chunk 0 in X buffer 0, chunk 1 in flight, the first item's bias / shift row ----
    if constexpr (!CONS) {
      wdma(0);
      wdma(1);
      wdma(2);
      fetch(xrB, tagB, trowB, tscB);
      fetch(xrA, tagA, trowA, tscA);
      bsv = bs_fetch(item_of(0));
      activate(xrB, tagB, trowB, tscB, pk);
      store_x(pk, tagB, 0);
      bs_commit(bsv, unscale_of(item_of(0)), 0);
      asm volatile("s_waitcnt vmcnt(20)" ::: "memory");                  // the three slabs and chunk 0; chunk 1's loads may stay in flight
    } else {
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[m][n][e] = 0.f;
    }
    PSTAMP(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    young = 0; prev = 24;                             // behind the prologue's slabs: chunk 1's loads
    if constexpr (CONS) load_pair(F0, 0, 0, 0, 0, 0, 0, 1, 0);         // P0 of the first chunk pair

    const int ncp = n_chunks >> 1;                    // even (the launcher's rule): fragment sets and ring slots are back in place after an item
    for (int it = 0; it < n_items; ++it) {
      for (int cp = 0; cp < ncp; cp += 2) {
        chunk_pair(std::integral_constant<int, 0>{}, F0, F1, cp == 0 && it > 0, false, it);
        chunk_pair(std::integral_constant<int, 6>{}, F1, F0, false, cp + 2 == ncp, it);
      }
    }
    // ---- drain: the last item's store phase (its own barriers, nothing staged, nothing multiplied) ----
    if constexpr (CONS) {
      park_tile((n_items - 1) & 1);
    } else {
      plan_store(item_of(n_items - 1));
      res_prefetch(RA, 0);
    }
    bar_counted(0);
    if constexpr (!CONS) {
      store_batch(RA, 0);
      res_prefetch(RB, 1);
      store_batch(RB, 1);
      res_prefetch(RA, 2);
      store_batch(RA, 2);
      res_prefetch(RB, 3);
      store_batch(RB, 3);
      if (want_amax) commit_amax_asm();
    }
    bar_counted(0);
    if constexpr (!CONS) {
      if (stats && wv == 4) store_stats();
    }
  };
  if (consumer) run(std::true_type{});
  else run(std::false_type{});
}

template <bool PRE, bool CIRC, int NRES>
int launch_conv3p_r(const Conv3hArgs& a, int wgs, hipStream_t s) {
  const int rc = ds::ensure_dynamic_lds<&k_conv3p<PRE, CIRC, NRES>>(P_LDS, "hipFuncSetAttribute(conv3p)");
  if (rc != DS_OK) return rc;
  hipLaunchKernelGGL((k_conv3p<PRE, CIRC, NRES>), dim3((unsigned)wgs), dim3(512), P_LDS, s, a);
  DS_CHECK_LAUNCH("ds_conv2d_h3 (persistent)");
  return DS_OK;
}
template <bool PRE, bool CIRC>
int launch_conv3p(Conv3hArgs a, int wgs, hipStream_t s) {
  if (!a.res1 && a.res2) { a.res1 = a.res2; a.res2 = nullptr; }      // one residual: the order of the additions is the same
  if (!a.res1) return launch_conv3p_r<PRE, CIRC, 0>(a, wgs, s);
  return a.res2 ? launch_conv3p_r<PRE, CIRC, 2>(a, wgs, s) : launch_conv3p_r<PRE, CIRC, 1>(a, wgs, s);
}

// DS_CONV_PC: 0 (default while the one-tile kernel still measures faster) = never, 1 = the fused-loader launches with one channel
// tile, 2 = also in place of the two-channel-tile kernel, 3 = raw-input launches too.
// DS_CONV_PC_MIN: fewest items per workgroup (default 4).
int conv3p_mode() {
  static const int v = [] { const char* e = getenv("DS_CONV_PC"); return e ? atoi(e) : 0; }();
  return v;
}
int conv3p_min_items() {
  static const int v = [] { const char* e = getenv("DS_CONV_PC_MIN"); const int n = e ? atoi(e) : 4; return n < 1 ? 1 : n; }();
  return v;
}
int conv3p_cus() {
  // one entry per device: the CU count decides the grid
  static std::atomic<int> cus[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  int c = cus[dev & 63].load(std::memory_order_acquire);
  if (c == 0) {
    hipDeviceProp_t p;
    c = hipGetDeviceProperties(&p, dev) == hipSuccess ? p.multiProcessorCount : 256;
    cus[dev & 63].store(c, std::memory_order_release);
  }
  return c;
}

}  // namespace

int conv3p_try_launch(const Conv3hArgs& a, hipStream_t s, bool* launched) {
  *launched = false;
  const int mode = conv3p_mode();
  if (mode == 0) return DS_OK;
  if (a.H % 8 != 0 || a.W % 32 != 0 || a.Cout % COT != 0 || a.Cin % (4 * KC) != 0) return DS_OK;      // an even number of chunk PAIRS (the consumer's two fragment sets)
  if (a.Hin != a.H || a.Win != a.W || a.res1_up) return DS_OK;
  const long long total = (long long)a.n_cot * a.tiles_x * a.tiles_y * a.B;
  const int wgs = conv3p_cus() / 8 * 8;               // a multiple of the XCD count: the item order assumes round-robin dealing
  if (wgs <= 0 || total < (long long)wgs * conv3p_min_items() || total >= (1ll << 31)) return DS_OK;
  if (a.n_cot % 2 == 0 && a.prenorm && mode < 2) return DS_OK;     // the two-channel-tile kernel's launches
  if (!a.prenorm && mode < 3) return DS_OK;           // raw-input launches: measured separately (DS_CONV_PC=3)
  int rc;
  if (a.prenorm) rc = a.circular ? launch_conv3p<true, true>(a, wgs, s) : launch_conv3p<true, false>(a, wgs, s);
  else rc = a.circular ? launch_conv3p<false, true>(a, wgs, s) : launch_conv3p<false, false>(a, wgs, s);
  if (rc == DS_OK) *launched = true;
  return rc;
}

}  // namespace ds_conv3
