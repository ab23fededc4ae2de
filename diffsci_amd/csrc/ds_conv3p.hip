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
//
// As shipped at the end of round 4 (second session; every step bit-identical to ds_conv3h.hip):
//   * staging by 16-byte loads (VEC: a unit = 2 channels x 4 pixels, the fused norm's table rows through LDS pad vectors), the two
//     kinds of producer wave in their own instantiation of the loop;
//   * the CONSUMERS issue the weight slabs' DMA at the head of their step (DS_PC_CDMA): a DMA costs its wave 150-200 cycles to issue,
//     the consumers wait for the producers at every barrier anyway, and the producers then have no cross-wave vector-memory wait at all;
//   * the producers' vector work in slot-thirds, one per step, the previous item's store phase one batch per (E,1) / (O,1) step of the
//     item's first two chunk pairs (DS_PC_SPREAD);
//   * IMG: pre-split image input (the 256-channel level) -- the producers issue DMA only (image patches in front of the weight slab).
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
#ifdef DS_STAMP
// inside one step of the second item's last chunk pair (producer wave 4): slots 48..
#define PSTAMP_FINE(k) do { if (it == 1 && last_of_item) PSTAMP(48 + (k)); } while (0)
#else
#define PSTAMP_FINE(k) do {} while (0)
#endif

// Wait until all but the `young` youngest vector-memory operations of this wave have completed (the exact count: rounded down to a
// multiple of four, 24 loads behind a 3-instruction DMA made the barrier wait for the DMA just issued), and for every LDS operation;
// then the workgroup barrier.
#define DS_VMCNT_CASE(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;
__device__ __forceinline__ void bar_counted(int young) {
  const int y = __builtin_amdgcn_readfirstlane(young);
  switch (y < 63 ? y : 63) {
    DS_VMCNT_CASE(0) DS_VMCNT_CASE(1) DS_VMCNT_CASE(2) DS_VMCNT_CASE(3) DS_VMCNT_CASE(4) DS_VMCNT_CASE(5) DS_VMCNT_CASE(6) DS_VMCNT_CASE(7)
    DS_VMCNT_CASE(8) DS_VMCNT_CASE(9) DS_VMCNT_CASE(10) DS_VMCNT_CASE(11) DS_VMCNT_CASE(12) DS_VMCNT_CASE(13) DS_VMCNT_CASE(14) DS_VMCNT_CASE(15)
    DS_VMCNT_CASE(16) DS_VMCNT_CASE(17) DS_VMCNT_CASE(18) DS_VMCNT_CASE(19) DS_VMCNT_CASE(20) DS_VMCNT_CASE(21) DS_VMCNT_CASE(22) DS_VMCNT_CASE(23)
    DS_VMCNT_CASE(24) DS_VMCNT_CASE(25) DS_VMCNT_CASE(26) DS_VMCNT_CASE(27) DS_VMCNT_CASE(28) DS_VMCNT_CASE(29) DS_VMCNT_CASE(30) DS_VMCNT_CASE(31)
    DS_VMCNT_CASE(32) DS_VMCNT_CASE(33) DS_VMCNT_CASE(34) DS_VMCNT_CASE(35) DS_VMCNT_CASE(36) DS_VMCNT_CASE(37) DS_VMCNT_CASE(38) DS_VMCNT_CASE(39)
    DS_VMCNT_CASE(40) DS_VMCNT_CASE(41) DS_VMCNT_CASE(42) DS_VMCNT_CASE(43) DS_VMCNT_CASE(44) DS_VMCNT_CASE(45) DS_VMCNT_CASE(46) DS_VMCNT_CASE(47)
    DS_VMCNT_CASE(48) DS_VMCNT_CASE(49) DS_VMCNT_CASE(50) DS_VMCNT_CASE(51) DS_VMCNT_CASE(52) DS_VMCNT_CASE(53) DS_VMCNT_CASE(54) DS_VMCNT_CASE(55)
    DS_VMCNT_CASE(56) DS_VMCNT_CASE(57) DS_VMCNT_CASE(58) DS_VMCNT_CASE(59) DS_VMCNT_CASE(60) DS_VMCNT_CASE(61) DS_VMCNT_CASE(62)
    default: asm volatile("s_waitcnt vmcnt(63)" ::: "memory"); break;
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
#undef DS_VMCNT_CASE

struct Item { int cot, b, y0, x0, tile; };
#ifndef DS_PC_FETCH_LATE
#define DS_PC_FETCH_LATE 0
#endif
#ifndef DS_PC_SPREAD
#define DS_PC_SPREAD 1                               // 0: the whole store phase in the item's first chunk pair (measurement builds)
#endif
#ifndef DS_PC_IMG_CDMA
#define DS_PC_IMG_CDMA 0                             // image input: 1 = the consumers issue the weight DMA there too (measurement builds)
#endif
#ifndef DS_PC_CDMA
#define DS_PC_CDMA 1                                 // 0: the producers issue the weight DMA (measurement builds, tools/build_variant.sh)
#endif
#define DS_LPI 8                                    // vector-memory loads per staging item

template <int XI> struct Packed { u32x4 h[XI], l[XI]; };
template <int BK> struct ResRegs { f32x4 r1[BK], r2[BK]; };
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
// NPW: producer waves, 4 (two waves per SIMD: 256 registers each) or 8 (three per SIMD: 168 registers each; the consumer then keeps
// ONE operand set and fetches a group's operands right in front of its matrix instructions -- it has the time: the producers pace
// this kernel -- and every producer wave stages and stores half as much).
// VEC (four producers): the patch's 32 interior columns are fetched by 16-byte loads, a staging unit = 2 channels x 4 pixels, the two
// halo columns keep the one-pixel items in the last slot of producer waves 2 and 3 -- ds_conv3h.hip's VEC plan, lane for lane.  A
// chunk costs a producer wave 7 or 13 vector-memory loads instead of 24 (their issue paces the fetch steps, see above).  The two kinds of
// producer wave run their own instantiation of the loop (role 1: three interior slots, role 2: two and a halo slot), so that every load
// stays an unconditional instruction of its wave's program.  The table rows of the fused norm come from LDS (the unit's channel pair
// differs from lane to lane): every producer wave loads the chunk's 16 rows in front of the patch, producer wave 0 parks them in the
// pad vectors of the X buffer the chunk is staged into, at least one barrier before the activation reads them.
// IMG: the input arrives as pre-split fp16 hi / lo images (ds_inorm_silu_images; ds_conv3h.hip, IMGIN) -- the 256-channel level of
// config 2.  The producers then only issue DMA: an X buffer is filled by 22 wave-instructions (the one-shot kernel's plan, lane for
// lane), the weight slabs stay with them too (DS_PC_CDMA applies to the staging forms only), and the consumers are pure matrix streams.
template <bool PRE, bool CIRC, int NRES, int NPW, bool VEC = false, bool IMG = false>
__global__ __launch_bounds__(256 + 64 * NPW, NPW == 4 ? 2 : 3) void k_conv3p(const Conv3hArgs a) {
  static_assert(NPW == 4 || NPW == 8, "four or eight producer waves");
  static_assert(!VEC || NPW == 4, "16-byte patch loads: the four-producer form");
  static_assert(!IMG || (!PRE && !CIRC && !VEC && NPW == 4), "image input: plain zero-padded four-producer form");
  constexpr int XI = NPW == 4 ? 3 : 2;                     // staging items per producer thread
  constexpr int BK = NPW == 4 ? 4 : 2;                     // store instructions per batch (four batches per wave and item)
  using Packed = ds_conv3::Packed<XI>;
  using ResRegs = ds_conv3::ResRegs<BK>;
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
  const int ptid = tid - 256;                              // producer thread (negative in the consumer waves: unused there)
  const int pw = wv - 4;                                   // producer wave
  const int rws = NPW == 4 ? rw : ((pw >> 1) & 3);         // the consumer whose tile this producer stores
  const int joff = NPW == 4 ? 0 : 8 * (pw & 1);            // ... and its first store instruction (eight producers: 32 channels each)

  // ---- this workgroup's items: XCD x owns one contiguous run of the logical order (channel tile fastest, then pixel tile, then
  //      sample), dealt round-robin over the Q workgroups of the XCD, so the workgroups of an XCD work on neighbouring tiles ----
  const unsigned ntiles = (unsigned)(a.tiles_x * a.tiles_y);
  const unsigned total = (unsigned)a.n_cot * ntiles * (unsigned)a.B;
  const unsigned Q = gridDim.x >> 3, xcd = blockIdx.x & 7u, q = blockIdx.x >> 3;
  const unsigned per = total >> 3, rem = total & 7u;
  const unsigned first = xcd * per + (xcd < rem ? xcd : rem), cnt = per + (xcd < rem ? 1u : 0u);
  const int n_items = q < cnt ? (int)((cnt - q + Q - 1) / Q) : 0;
  if (n_items == 0) return;
  // item index -> (channel tile, pixel tile, sample): divisions by multiplication with the launcher's 2^40 reciprocals (exact for the
  // sizes the launcher admits).  Stamped before: the compiler's float-reciprocal division sequences made one decode cost more
  // than a thousand cycles, five times per tile, on the producers' critical path.
  auto div40 = [](unsigned n, unsigned long long magic) __attribute__((always_inline)) { return (unsigned)(((unsigned long long)n * magic) >> 40); };
  auto item_of = [&](int k) __attribute__((always_inline)) {
    const unsigned logical = first + q + Q * (unsigned)k;
    Item it;
    const unsigned rest = div40(logical, a.ncot_magic40);
    it.cot = (int)(logical - rest * (unsigned)a.n_cot);
    const unsigned bb = div40(rest, a.ntiles_magic40);
    it.tile = (int)(rest - bb * ntiles);
    it.b = (int)bb;
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
  // eight producers: waves 0-3 channel half 0, waves 4-7 half 1; item 0 = positions 64 (pw & 3) + lane, item 1 = the tail,
  // 21 positions per wave (256 + 21 (pw & 3) + lane, lanes 0-20)
  const int tail_h = (rw >> 1) & 1;
  auto item_h = [&](int i) __attribute__((always_inline)) { return NPW == 8 ? ((pw >> 2) & 1) : (i == 0 ? 0 : (i == 1 ? 1 : tail_h)); };
  int xlds[XI], xrow[XI], xcol[XI];
  const bool live_tail = NPW == 4 ? (NT + (ptid & (NT / 2 - 1)) < NPOS) : (lane < 21);     // the thread's tail item exists
  auto live = [&](int i) __attribute__((always_inline)) { return i < XI - 1 || live_tail; };
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    int pos;
    if (NPW == 4) pos = i < 2 ? (ptid & 255) : NT + (ptid & (NT / 2 - 1));
    else pos = i == 0 ? 64 * (pw & 3) + lane : 256 + 21 * (pw & 3) + (lane < 21 ? lane : 20);
    xrow[i] = pos / PW;
    xcol[i] = pos - xrow[i] * PW;
    xlds[i] = item_h(i) * HS + pos;
  }
  // VEC plan (ds_conv3h.hip): slot i = interior unit u = 256 i + ptid -> (h, patch row, column group, channel pair); the last slot of
  // producer waves 2 / 3 = halo item lane (20 lanes: row lane / 2, column 0 or 33) of h = 0 / 1
  int vr[XI], vx[XI], vch[XI], vlds[XI], vrow[XI];
  const bool halo_live = lane < 20;
  if constexpr (VEC) {
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      vr[i] = 0; vx[i] = 0; vch[i] = 0; vlds[i] = 0; vrow[i] = 0;
      if (i < XI - 1 || pw < 2) {
        const int u = i * 256 + (ptid & 255);
        const int q = u & 3, G = u >> 5;
        const int h = G >= 10 ? 1 : 0, Gp = G - 10 * h;
        int r, g;
        if (Gp < 8) { r = 4 * (Gp >> 2) + ((u >> 3) & 3); g = 2 * (Gp & 3) + ((u >> 2) & 1); }
        else { r = 8 + ((u >> 3) & 1); g = 4 * (Gp - 8) + 2 * ((u >> 4) & 1) + ((u >> 2) & 1); }
        vr[i] = r; vx[i] = 4 * g;
        vch[i] = (8 * h + 2 * q) * HW;
        vlds[i] = 16 * (h * HS + r * PW + 1 + 4 * g) + 4 * q;
        vrow[i] = 8 * h + 2 * q;
      } else {
        const int e = halo_live ? lane : 19, h = pw - 2;
        vr[i] = e >> 1; vx[i] = (e & 1) ? PW - 2 : -1;           // column offset from x0: the patch's column 33 / 0
        vch[i] = 8 * h * HW;
        vlds[i] = 16 * (h * HS + vr[i] * PW + vx[i] + 1);
        vrow[i] = 8 * h;
      }
    }
  }
  auto pad_vec = [&](int c) __attribute__((always_inline)) { return c < HPAD16 ? NPOS + c : PS + NPOS + (c - HPAD16); };
  // ---- IMG: DMA plan of an X buffer (ds_conv3h.hip: NINST = 22 instructions of 64 vectors, the last one moved back to end with the
  //      buffer; lane -> linear index -> (image, position)), relative to the tile's origin: full tiles only, so no clamping ----
  constexpr int NINST = (XBV + 63) / 64;
  const int Wp = a.W + 2, Hp = a.H + 2;
  unsigned dbase[6] = {};
  if constexpr (IMG) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int k = pw + 4 * i;
      const int L = (k < NINST - 1 ? 64 * k : XBV - 64) + lane;
      const int piece = L >= PS ? 1 : 0, Lp = L - piece * PS;
      const int hh = Lp >= HS ? 1 : 0;
      int pos = Lp - hh * HS;
      pos = pos < NPOS ? pos : NPOS - 1;
      const int r = pos / PW, col = pos - r * PW;
      dbase[i] = 16u * (unsigned)(((2 * piece + hh) * Hp + r) * Wp + col);        // bytes
    }
  }
  int i_k = 0, i_chunk = 0, i_b = 0;
  unsigned i_off = 0;
  // the next chunk's images into X buffer `buf`; returns the wave's instruction count (unconditional past the end, like the fetch)
  auto xdma = [&](int buf) __attribute__((always_inline)) {
    const bool valid = i_k < n_items;
    if (valid && i_chunk == 0) { const Item it = item_of(i_k); i_off = 16u * (unsigned)(it.y0 * Wp + it.x0); i_b = it.b; }
    const u32x4* src = reinterpret_cast<const u32x4*>(a.in) + ((size_t)i_b * n_chunks + i_chunk) * 4 * (size_t)(Hp * Wp);
    u32x4* dst = Xs + buf * XBV;
    int n = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int k = pw + 4 * i;                        // wave-uniform
      if (k < NINST) { lds_dma16(src, dbase[i] + i_off, lds_address(dst + (k < NINST - 1 ? 64 * k : XBV - 64))); ++n; }
    }
    if (valid && ++i_chunk == n_chunks) { i_chunk = 0; ++i_k; }
    return n;
  };
  // fetch cursor: the next chunk to load.  The loads are UNCONDITIONAL (past the end of the list the last item's first chunk is
  // loaded again and dropped): a conditional load between a load and its use makes hipcc wait for vmcnt(0) at the use -- the
  // full HBM latency of the loads just issued, measured at 3000 cycles per chunk in the first version of this kernel.
  int f_k = 0, f_chunk = 0, f_b = 0;
  int xoff[XI] = {};
  unsigned f_xvalid = 0;
  float f_in_scale = 1.f;
  auto plan_fetch = [&](const Item& it) __attribute__((always_inline)) {
    f_xvalid = 0;
    if constexpr (VEC) {
#pragma unroll
      for (int i = 0; i < XI; ++i) {
        const bool halo = i == XI - 1 && pw >= 2;
        int gy = it.y0 + vr[i] - 1 + a.oy, gx = it.x0 + vx[i];
        if (CIRC) {
          gy = gy < 0 ? gy + a.H : (gy >= a.H ? gy - a.H : gy);
          gx = gx < 0 ? gx + a.W : (gx >= a.W ? gx - a.W : gx);
          gy = gy >= a.H ? a.H - 1 : gy;
        }
        const bool ok = (!halo || halo_live) && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        xoff[i] = ok ? gy * a.W + gx : 0;
        if (ok) f_xvalid |= 1u << i;
      }
      f_b = it.b;
      if constexpr (!PRE) f_in_scale = ds_epi::act_scale_of(a.in_amax ? const_u32(a.in_amax, it.b) : 0u, a.wshift).in_scale;
      return;
    }
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      int gy = it.y0 + xrow[i] - 1 + a.oy, gx = it.x0 + xcol[i] - 1 + a.ox;
      if (CIRC) {
        gy = gy < 0 ? gy + a.H : (gy >= a.H ? gy - a.H : gy);
        gx = gx < 0 ? gx + a.W : (gx >= a.W ? gx - a.W : gx);
        gy = gy >= a.H ? a.H - 1 : gy;
        gx = gx >= a.W ? a.W - 1 : gx;
      }
      const bool ok = live(i) && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
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
  // HALO (compile time): this producer wave's last slot is a halo item (VEC, producer waves 2 and 3)
  auto fetch_item = [&](auto halo_tag, float (&xr)[XI][8], int i) __attribute__((always_inline)) {
    constexpr bool HALO = decltype(halo_tag)::value;
    if constexpr (VEC) {
      const float* p0 = f_src + vch[i] + xoff[i];
      if (HALO && i == XI - 1) {
#pragma unroll
        for (int k = 0; k < 8; ++k) xr[i][k] = p0[(size_t)k * HW];
      } else {
        const f32x4 t0 = *reinterpret_cast<const f32x4*>(p0);
        const f32x4 t1 = *reinterpret_cast<const f32x4*>(p0 + HW);
#pragma unroll
        for (int k = 0; k < 4; ++k) { xr[i][k] = t0[k]; xr[i][4 + k] = t1[k]; }
      }
    } else {
      const float* p0 = f_src + xoff[i] + (size_t)(8 * item_h(i)) * HW;
#pragma unroll
      for (int k = 0; k < 8; ++k) xr[i][k] = p0[(size_t)k * HW];
    }
  };
  // vector-memory loads of one fetched chunk, per wave
  auto fetch_count = [&](auto halo_tag) __attribute__((always_inline)) {
    constexpr bool HALO = decltype(halo_tag)::value;
    return VEC ? (PRE ? 1 : 0) + (HALO ? 2 * (XI - 1) + 8 : 2 * XI) : DS_LPI * XI;
  };
  // VEC + PRE: the chunk's 16 table rows, one 16-byte load in front of the patch's (every producer wave, so the counts stay uniform)
  auto rows_fetch = [&](f32x4& prow, int trow) __attribute__((always_inline)) {
    if constexpr (VEC && PRE) prow = reinterpret_cast<const f32x4*>(a.prenorm)[trow + (lane & 15)];
  };
  auto rows_park = [&](const f32x4& prow, int buf) __attribute__((always_inline)) {
    if constexpr (VEC && PRE) {
      if (pw == 0 && lane < 16) Xs[buf * XBV + pad_vec(lane)] = __builtin_bit_cast(u32x4, prow);
    }
  };
  auto fetch = [&](auto halo_tag, float (&xr)[XI][8], f32x4& prow, unsigned& tag, int& trow, float& tscale) __attribute__((always_inline)) {
    fetch_begin(tag, trow, tscale);
    rows_fetch(prow, trow);
#pragma unroll
    for (int i = 0; i < XI; ++i) fetch_item(halo_tag, xr, i);
    return fetch_count(halo_tag);
  };
  // [norm + SiLU,] fp16 hi / lo split of one staging item of a fetched chunk, in registers
  auto activate_item = [&](auto halo_tag, float (&xr)[XI][8], unsigned tag, int trow, float tscale, Packed& pk, int i, int buf) __attribute__((always_inline)) {
    constexpr bool HALO = decltype(halo_tag)::value;
    if constexpr (VEC) {
      if (!(HALO && i == XI - 1)) {
        // interior unit: xr[i][0..3] = four pixels of channel 2q, xr[i][4..7] = of channel 2q + 1; rows from the buffer's pad vectors
        if constexpr (PRE) {
          const u32x4* rows = Xs + buf * XBV;
          const f32x4 p0 = __builtin_bit_cast(f32x4, rows[pad_vec(vrow[i])]);
          const f32x4 p1 = __builtin_bit_cast(f32x4, rows[pad_vec(vrow[i] + 1)]);
          const float inv = p0[3] == 0.f ? 1.0f : p0[3];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            xr[i][k] = ds_h3::fast_silu_scaled((xr[i][k] - p0[0]) * p0[1] + p0[2], inv);
            xr[i][4 + k] = ds_h3::fast_silu_scaled((xr[i][4 + k] - p1[0]) * p1[1] + p1[2], inv);
          }
        }
        const bool item_ok = (tag >> i) & 1u;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float v0 = item_ok ? xr[i][k] : 0.f;
          float v1 = item_ok ? xr[i][4 + k] : 0.f;
          if constexpr (!PRE) { v0 *= tscale; v1 *= tscale; }
          unsigned ph, pl;
          split2(v0, v1, ph, pl);
          pk.h[i][k] = ph; pk.l[i][k] = pl;
        }
        return;
      }
    }
    if constexpr (PRE) {
      typedef const __attribute__((address_space(4))) f32x4* cptr;
      cptr pp = (cptr)(reinterpret_cast<const f32x4*>(a.prenorm)) + trow;
      const int h = VEC ? pw - 2 : item_h(i);
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
  auto activate = [&](auto halo_tag, float (&xr)[XI][8], unsigned tag, int trow, float tscale, Packed& pk, int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < XI; ++i) activate_item(halo_tag, xr, tag, trow, tscale, pk, i, buf);
  };
  // all staging slots of a chunk but the last (the last one rides beside the NEXT fetch: fetch_beside_last_slot)
  auto activate_first_slots = [&](auto halo_tag, float (&xr)[XI][8], unsigned tag, int trow, float tscale, Packed& pk, int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < XI - 1; ++i) activate_item(halo_tag, xr, tag, trow, tscale, pk, i, buf);
  };
  // the loads of chunk n beside the vector work of the LAST slot of chunk o (whose other slots were activated a step earlier)
  auto fetch_beside_last_slot = [&](auto halo_tag, float (&xn)[XI][8], f32x4& prown, unsigned& tagn, int& trown, float& tscn,
                                    float (&xo)[XI][8], unsigned tago, int trowo, float tsco, Packed& pk, int bufo) __attribute__((always_inline)) {
#if DS_PC_FETCH_LATE == 1   // measurement builds: the vector work first, the loads behind it (+3.5 %, profiles/r04_pc_smooth_schedule.log) ...
    activate_item(halo_tag, xo, tago, trowo, tsco, pk, XI - 1, bufo);
    __builtin_amdgcn_sched_barrier(0);
    fetch_begin(tagn, trown, tscn);
    rows_fetch(prown, trown);
#pragma unroll
    for (int i = 0; i < XI; ++i) fetch_item(halo_tag, xn, i);
#elif DS_PC_FETCH_LATE == 2   // ... or every load in front of it
    fetch_begin(tagn, trown, tscn);
    rows_fetch(prown, trown);
#pragma unroll
    for (int i = 0; i < XI; ++i) fetch_item(halo_tag, xn, i);
    __builtin_amdgcn_sched_barrier(0);
    activate_item(halo_tag, xo, tago, trowo, tsco, pk, XI - 1, bufo);
#else
    fetch_begin(tagn, trown, tscn);
    rows_fetch(prown, trown);
#pragma unroll
    for (int i = 0; i < XI - 1; ++i) fetch_item(halo_tag, xn, i);
    __builtin_amdgcn_sched_barrier(0);
    activate_item(halo_tag, xo, tago, trowo, tsco, pk, XI - 1, bufo);
    __builtin_amdgcn_sched_barrier(0);
    fetch_item(halo_tag, xn, XI - 1);
#endif
    return fetch_count(halo_tag);
  };
  auto store_x = [&](auto halo_tag, const Packed& pk, unsigned tag, int buf) __attribute__((always_inline)) {
    constexpr bool HALO = decltype(halo_tag)::value;
    if (!(tag >> 31)) return;
    if constexpr (VEC) {
      unsigned char* xbytes = smem + (size_t)buf * XBV * 16;
#pragma unroll
      for (int i = 0; i < XI; ++i) {
        if (HALO && i == XI - 1) {
          if (halo_live) {
            *reinterpret_cast<u32x4*>(xbytes + vlds[i]) = pk.h[i];
            *reinterpret_cast<u32x4*>(xbytes + vlds[i] + PS * 16) = pk.l[i];
          }
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            *reinterpret_cast<unsigned*>(xbytes + vlds[i] + 16 * k) = pk.h[i][k];
            *reinterpret_cast<unsigned*>(xbytes + vlds[i] + 16 * k + PS * 16) = pk.l[i][k];
          }
        }
      }
      return;
    }
    u32x4* xb = Xs + buf * XBV;
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      if (live(i)) {
        xb[xlds[i]] = pk.h[i];
        xb[PS + xlds[i]] = pk.l[i];
      }
    }
  };
  // weight cursor: the next slab to DMA.  Unconditional as well (past the end: the last item's first slab into a slot nobody reads).
  int w_k = 0, w_step = 0, w_cot = 0;
  // widx / nwv: the issuing wave's index among the nwv waves that share a slab's twelve pieces -- the producers, or (cdma) the four
  // CONSUMERS: a weight DMA costs its wave 150-200 cycles to issue (stamped), three per step; the producers pace this kernel and the
  // consumers wait for them at every barrier, so the consumers issue them, in front of the step's matrix instructions
  constexpr bool cdma = DS_PC_CDMA != 0 && (!IMG || DS_PC_IMG_CDMA != 0);            // compile time (as a kernel argument the two forms side by side cost 15 % -- measured)
  auto wdma = [&](int slot, int widx, int nwv) __attribute__((always_inline)) {     // slot = (slab index) & 3
    const bool valid = w_k < n_items;
    if (valid && w_step == 0) w_cot = item_of(w_k).cot;
    const u32x4* src = a.wp + ((size_t)w_cot * n_steps + w_step) * WSLAB_VEC;
    u32x4* dst = Ws + slot * WSLAB_VEC;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int k = widx + nwv * i;                   // wave-uniform piece of the slab's twelve
      if (k < 12) lds_dma16(src + 64 * k, 16u * (unsigned)lane, lds_address(dst + 64 * k));
    }
    if (valid && ++w_step == n_steps) { w_step = 0; ++w_k; }
    return 0;                                          // the DMA is what the barrier waits for: never among the `young`
  };
  // bias / shift row of an item: one register per thread (threads 0-63: bias, 64-127: shift), committed a step later together
  // with the item's 2^-(wshift + k) (thread 128; read through the scalar cache, so nothing waits for vector memory here)
  // (both values in EVERY lane, in registers of their own: as one register written under two lane masks the second load had to wait
  // for the first -- and with it for the 24 patch loads in front of it: 1,600 cycles in step (O,0), stamped)
  auto bs_fetch = [&](const Item& it, float& vb, float& vs) __attribute__((always_inline)) {
    const int co = it.cot * COT + (ptid & 63);
    vb = a.bias ? a.bias[co] : 0.f;
    vs = a.shift ? a.shift[(size_t)it.b * a.shift_stride + co] : 0.f;
  };
  auto unscale_of = [&](const Item& it) __attribute__((always_inline)) {
    if constexpr (PRE) {
      const float inv = __builtin_bit_cast(float, const_u32(a.prenorm, (size_t)it.b * n_chunks * KC * 4 + 3));
      return ds_epi::unscale_from_inv(inv == 0.f ? 1.0f : inv, a.wshift);
    } else {
      return ds_epi::act_scale_of(a.in_amax ? const_u32(a.in_amax, it.b) : 0u, a.wshift).unscale;
    }
  };
  auto bs_commit = [&](float vb, float vs, float unscale, int par) __attribute__((always_inline)) {
    if (ptid < 128) BS[par * P_BS_FLOATS + ptid] = ptid < 64 ? vb : vs;
    else if (ptid == 128) BS[par * P_BS_FLOATS + 128] = unscale;
  };

  // store phase of the previous item (producer rw stores consumer rw's tile): four batches of four 16-byte store instructions
  // (ds_conv_epilogue.h: store_tile_rows' aligned path, full tiles).  The wave's statistics partials [64 channels][4] go to the
  // head of its own tile, behind the batch that has just read it.
  int s_b = 0, s_cot = 0, s_tile = 0;
  f32x4 pend0[BK];                                   // eight producers: the statistics rows of batch 0, written with batch 1 (store_batch)
  size_t s_idx = 0;                                  // the lane's first output element
  const size_t s_step = 4 * (size_t)HW;
  float s_amax = 0.f;
  const int p4 = 4 * (lane & 7);
  const bool stats = a.tile_stats != nullptr, want_amax = a.out_amax != nullptr;
  constexpr int nres = BK * NRES;
  auto plan_store = [&](const Item& it) __attribute__((always_inline)) {
    s_b = it.b; s_cot = it.cot; s_tile = it.tile;
    const int gy = it.y0 + 2 * rws + ((lane >> 3) & 1), gx = it.x0 + p4;
    const size_t ch = (size_t)it.b * a.Cout + it.cot * COT + 4 * joff + (lane >> 4);
    s_idx = ch * HW + (size_t)gy * a.W + gx;
    s_amax = 0.f;
  };
  // residual vectors of one batch.  (No half-resolution res1 here: with both forms in one function the two loads share destination
  // registers, and hipcc then waits for vmcnt(0) in front of every residual load -- the launcher keeps such launches on ds_conv3h.hip.)
  auto res_prefetch = [&](ResRegs& R, int half) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < BK; ++k) {
      const int j = half * BK + k;
      if constexpr (NRES >= 1) R.r1[k] = *reinterpret_cast<const f32x4*>(a.res1 + s_idx + (size_t)j * s_step);
      if constexpr (NRES >= 2) R.r2[k] = *reinterpret_cast<const f32x4*>(a.res2 + s_idx + (size_t)j * s_step);
    }
    return nres;
  };
  auto store_batch = [&](const ResRegs& R, int half) __attribute__((always_inline)) {
    float* tile = OUT + rws * 4096;
    f32x4 v[BK];
#pragma unroll
    for (int k = 0; k < BK; ++k) {
      const int seg = (joff + half * BK + k) * 8 + (lane >> 3);
      v[k] = *reinterpret_cast<const f32x4*>(&tile[seg * 32 + p4]);
    }
    if constexpr (NRES >= 1) {
#pragma unroll
      for (int k = 0; k < BK; ++k) v[k] = v[k] + R.r1[k];
    }
    if constexpr (NRES >= 2) {
#pragma unroll
      for (int k = 0; k < BK; ++k) v[k] = v[k] + R.r2[k];
    }
#pragma unroll
    for (int k = 0; k < BK; ++k) *reinterpret_cast<f32x4*>(a.out + s_idx + (size_t)(half * BK + k) * s_step) = v[k];
    if (want_amax) {
#pragma unroll
      for (int k = 0; k < BK; ++k) s_amax = fmaxf(s_amax, ds_epi::abs_max4(v[k]));
    }
    if (stats) {
      f32x4 o[BK];
#pragma unroll
      for (int k = 0; k < BK; ++k) {
        const float K = ds_epi::row16_first(v[k].x);
        const f32x4 d = v[k] - K;
        const float sv = ds_epi::row16_sum((d.x + d.y) + (d.z + d.w));
        const float qv = ds_epi::row16_sum((d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w));
        o[k] = f32x4{K, sv, qv, 64.f};                                   // the wave's 2 x 32 pixels of the channel
      }
      // row (channel) 4 j + lane / 16 of [64][4] at the tile's head, i.e. inside the region store instructions 0-3 read.  Four producers:
      // that is this wave's own batch 0, read by now.  Eight producers: it is the FIRST wave's batches 0 / 1 -- the second wave of a
      // tile (and, for one rule, the first) holds its batch-0 rows back until its batch 1, which the schedule runs a barrier later
      if (NPW == 8 && half == 0) {
#pragma unroll
        for (int k = 0; k < BK; ++k) pend0[k] = o[k];
      } else if ((lane & 15) == 0) {
        if (NPW == 8 && half == 1) {
#pragma unroll
          for (int k = 0; k < BK; ++k) *reinterpret_cast<f32x4*>(&tile[4 * (4 * (joff + k) + (lane >> 4))]) = pend0[k];
        }
#pragma unroll
        for (int k = 0; k < BK; ++k) *reinterpret_cast<f32x4*>(&tile[4 * (4 * (joff + half * BK + k) + (lane >> 4))]) = o[k];
      }
    }
    return BK;
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

  float xrA[XI][8], xrB[XI][8];
  unsigned tagA = 0, tagB = 0, tagP = 0;
  int trowA = 0, trowB = 0;
  float tscA = 1.f, tscB = 1.f;
  Packed pk;
  f32x4 prowA = {0.f, 0.f, 0.f, 0.f}, prowB = {0.f, 0.f, 0.f, 0.f};      // VEC + PRE: the table rows that go with xrA / xrB
  ResRegs RA, RB;
  float bsb = 0.f, bss = 0.f;
  Frag F0, F1;

  // Both roles run the SAME loop nest (one generic lambda, instantiated per role): every barrier is one program point of that
  // nest, so the two roles execute the same number of barriers by construction, while each role's registers are live only in its
  // own instantiation (as one loop with role branches inside, the allocator had to keep both roles' state alive at once: 256
  // VGPRs and 700 spilled).
  auto run = [&](auto role_tag) __attribute__((always_inline)) {
    // role 0: consumer; 1: producer; 2 (VEC only): producer whose last staging slot is a halo item (producer waves 2 and 3)
    constexpr int ROLE = decltype(role_tag)::value;
    constexpr bool CONS = ROLE == 0;
    const std::integral_constant<bool, ROLE == 2> halo{};
    // Counted barrier wait of the producers.  A step's weight DMA (slab p + 3) is the FIRST vector-memory operation of step p and
    // must have landed by the END of step p + 1: everything issued behind it in step p (`prev`), step p + 1's own DMA (3) and
    // everything else of step p + 1 (`young`) may stay in flight.  Loads, LDS-DMA and stores complete in issue order on gfx9
    // (hipcc's own counted waits rely on it), so "all but the N youngest" names exactly the operations in front of them.
    int young = 0, prev = 0, nx = 0;                  // nx (IMG): this step's image DMA, issued in front of its weight DMA
    const int ndma = NPW == 4 ? 3 : (pw < 4 ? 2 : 1);             // DMA instructions of this wave per step
    auto barrier = [&]() __attribute__((always_inline)) {
      // (cdma: a consumer's only vector-memory operations are its three DMA per step -- those of the step that ends here may stay in
      //  flight, the previous step's have landed; the producers then have nothing another wave waits for)
      if constexpr (CONS) bar_counted(cdma ? 3 : 0);
      else {
        // (image input: the producers' image DMA is the oldest operation of its step whoever issues the weights)
        if constexpr (IMG) bar_counted(prev + nx + (cdma ? 0 : ndma) + young);
        else bar_counted(cdma ? 63 : prev + ndma + young);
        prev = young; young = 0; nx = 0;
      }
    };
    int stamp = 1;
    (void)stamp;
    // One chunk pair (E, O): six steps, one barrier each.  Consumer: nine K = 32 groups P0 .. P8, the operands of group k + 1
    // fetched (into the other fragment set) while group k multiplies; nine is odd, so the two sets swap roles from one chunk pair
    // to the next and the loop body below is two chunk pairs = 12 steps = three turns of the 4-slot weight ring (BASE = 0 / 6:
    // the step position of (E,0); slab of position p sits in slot p & 3).  Producer duties per step (chunk numbers relative to E):
    //   (E,0)  fetch chunk E+2 -> B  beside  activate + split the LAST slot of chunk O (A) | residuals of batch 0
    //   (E,1)  store chunk O -> X1, park chunk E+2's table rows | batch 0, residuals of batch 1
    //   (E,2)  activate + split the first slots of chunk E+2 (B) | batch 1, residuals of batch 2
    //   (O,0)  fetch chunk O+2 -> A  beside  the last slot of chunk E+2 | batch 2, residuals of batch 3 | next item's bias / shift loads
    //   (O,1)  store chunk E+2 -> X0, park chunk O+2's table rows | batch 3, output maxima
    //   (O,2)  activate + split the first slots of chunk O+2 (A) | statistics combine | commit the next item's bias / shift row
    // (+ the slab's DMA at the head of every step when the producers issue it: DS_PC_CDMA=0.)  One slot-third of the vector work per
    // step instead of a whole chunk in (E,0) and in (O,1): the consumer's steps are 1-2 K = 32 groups long, and a step lasts as long
    // as the slower role (round 4, second session: profiles/r04_pc_smooth_schedule.log).
    // (batches = the PREVIOUS item's store phase, on the item's first chunk pair only.)  What a group reads is published one
    // barrier before the group that precedes it starts, because its operands are fetched during that predecessor.
    // [A schedule that spreads the staging evenly over the six steps and the store phase over two chunk pairs -- fetches in (E,1)
    //  and (E,2), activations in (E,0) and (O,0) -- measured WORSE (profiles/r04_pc_stamps.log, v7: 81.1 -> 81.1 samples/s against
    //  +1.6 % for this one): with both chunks' 48 loads issued in consecutive steps next to the residual loads and stores the wave
    //  runs into its 64 outstanding vector-memory operations and the loads' issue stalls.]
    // head / second: the item's first / second chunk pair.  The previous item's store phase -- four batches of stores, residual
    // loads, statistics -- runs on the producers one batch per (E,1) / (O,1) step of these TWO pairs (DS_PC_SPREAD; every item has
    // at least two): a batch costs its wave 1,000-2,000 cycles, and with all four in the first pair's steps (E,1) .. (O,1) that
    // pair took 19,400 cycles against 12,200 for the others (profiles/r04_pc_smooth_schedule.log)
    constexpr bool SPREAD = DS_PC_SPREAD != 0;
    auto chunk_pair = [&](auto base_tag, Frag& Fa, Frag& Fb, bool head, bool second, bool last_of_item, int it) __attribute__((always_inline)) {
      constexpr int B = decltype(base_tag)::value;
      constexpr int E0 = (B + 0) & 3, E1 = (B + 1) & 3, E2 = (B + 2) & 3;          // ring slots of chunk E's slabs
      constexpr int O0 = (B + 3) & 3, O1 = (B + 4) & 3, O2 = (B + 5) & 3;          // ... of chunk O's
      constexpr int N0 = (B + 6) & 3;                                              // ... of the next chunk pair's first
      // the ten K = 32 groups in order: P0 .. P8 of this chunk pair and P0 of the next (slot, ky, kx, X buffer of taps A and B)
      constexpr int G[10][8] = {{E0, 0, 0, 0, E0, 0, 1, 0}, {E0, 0, 2, 0, E1, 1, 0, 0}, {E1, 1, 1, 0, E1, 1, 2, 0}, {E2, 2, 0, 0, E2, 2, 1, 0},
                                {E2, 2, 2, 0, O0, 0, 0, 1}, {O0, 0, 1, 1, O0, 0, 2, 1}, {O1, 1, 0, 1, O1, 1, 1, 1}, {O1, 1, 2, 1, O2, 2, 0, 1},
                                {O2, 2, 1, 1, O2, 2, 2, 1}, {N0, 0, 0, 0, N0, 0, 1, 0}};
      // group k: four producers -- multiply the set fetched during group k - 1 while fetching group k + 1 into the other set;
      // eight producers -- one set: fetch, then multiply
      auto group = [&](auto k_tag) __attribute__((always_inline)) {
        constexpr int k = decltype(k_tag)::value;
        if constexpr (NPW == 4) {
          Frag& cur = (k & 1) ? Fb : Fa;
          Frag& nxt = (k & 1) ? Fa : Fb;
          load_pair(nxt, G[k + 1][0], G[k + 1][1], G[k + 1][2], G[k + 1][3], G[k + 1][4], G[k + 1][5], G[k + 1][6], G[k + 1][7]);
          mma_pair(cur);
          reads_between();
        } else {
          load_pair(Fa, G[k][0], G[k][1], G[k][2], G[k][3], G[k][4], G[k][5], G[k][6], G[k][7]);
          mma_pair(Fa);
        }
      };
#define DS_GROUP(k) group(std::integral_constant<int, k>{})
      // ---------------- step (E, 0) ----------------
      if constexpr (CONS) {
        if (cdma) { wdma((B + 3) & 3, rw, 4); __builtin_amdgcn_sched_barrier(0); }
        if (head) park_tile((it - 1) & 1);
        DS_GROUP(0); DS_GROUP(1);
      } else {
        if constexpr (IMG) nx = xdma(1);              // chunk O -> X1: free since the barrier of the previous (O, 2), read from (E, 2) on
        if (!cdma) wdma((B + 3) & 3, pw, NPW);
        __builtin_amdgcn_sched_barrier(0);
        if (head) { plan_store(item_of(it - 1)); young += res_prefetch(RA, 0); }
        if constexpr (!IMG) {
          young += fetch_beside_last_slot(halo, xrB, prowB, tagB, trowB, tscB, xrA, tagA, trowA, tscA, pk, 1);   // chunk E + 2 | chunk O's last slot
          tagP = tagA;
        }
      }
      if (stamp < 48) PSTAMP(stamp);
      ++stamp;
      barrier();
      // ---------------- step (E, 1) ----------------
      if constexpr (CONS) {
        if (cdma) { wdma((B + 4) & 3, rw, 4); __builtin_amdgcn_sched_barrier(0); }
        DS_GROUP(2);
      } else {
        if (!cdma) wdma((B + 4) & 3, pw, NPW);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!IMG) {
          store_x(halo, pk, tagP, 1);
          rows_park(prowB, 0);                        // chunk E + 2's table rows: read by its activation from (E, 2) on
        }
        if (head) {
          young += store_batch(RA, 0);
          young += res_prefetch(RB, 1);
        }
        if constexpr (SPREAD) {
          if (second) { young += store_batch(RA, 2); young += res_prefetch(RB, 3); }
        }
      }
      if (stamp < 48) PSTAMP(stamp);
      ++stamp;
      barrier();
      // ---------------- step (E, 2) ----------------
      if constexpr (CONS) {
        if (cdma) { wdma((B + 5) & 3, rw, 4); __builtin_amdgcn_sched_barrier(0); }
        DS_GROUP(3); DS_GROUP(4);
      } else {
        if (!cdma) wdma((B + 5) & 3, pw, NPW);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!IMG) activate_first_slots(halo, xrB, tagB, trowB, tscB, pk, 0);      // chunk E + 2 (fetched in (E, 0), rows parked in (E, 1))
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!SPREAD) {
          if (head) {
            young += store_batch(RB, 1);
            young += res_prefetch(RA, 2);
          }
        }
      }
      if (stamp < 48) PSTAMP(stamp);
      ++stamp;
      barrier();
      // ---------------- step (O, 0) ----------------
      if constexpr (CONS) {
        if (cdma) { wdma((B + 6) & 3, rw, 4); __builtin_amdgcn_sched_barrier(0); }
        DS_GROUP(5);
      } else {
        PSTAMP_FINE(0);
        if constexpr (IMG) nx = xdma(0);              // chunk E + 2 -> X0: free since the barrier of (E, 2), read from (O, 2) on
        if (!cdma) wdma((B + 6) & 3, pw, NPW);
        __builtin_amdgcn_sched_barrier(0);
        PSTAMP_FINE(1);
        if constexpr (!IMG) young += fetch_beside_last_slot(halo, xrA, prowA, tagA, trowA, tscA, xrB, tagB, trowB, tscB, pk, 0);       // chunk O + 2 | chunk E + 2's last slot
        __builtin_amdgcn_sched_barrier(0);
        PSTAMP_FINE(2);
        if constexpr (!SPREAD) {
          if (head) {
            young += store_batch(RA, 2);
            young += res_prefetch(RB, 3);
          }
        }
        if (last_of_item && it + 1 < n_items) bs_fetch(item_of(it + 1), bsb, bss);
      }
      if (stamp < 48) PSTAMP(stamp);
      ++stamp;
      barrier();
      // ---------------- step (O, 1) ----------------
      if constexpr (CONS) {
        if (cdma) { wdma((B + 7) & 3, rw, 4); __builtin_amdgcn_sched_barrier(0); }
        DS_GROUP(6); DS_GROUP(7);
      } else {
        PSTAMP_FINE(4);
        if (!cdma) wdma((B + 7) & 3, pw, NPW);
        __builtin_amdgcn_sched_barrier(0);
        PSTAMP_FINE(5);
        if constexpr (!IMG) {
          store_x(halo, pk, tagB, 0);                 // chunk E + 2
          rows_park(prowA, 1);                        // chunk O + 2's table rows: read by its activation from (O, 2) on
        }
        PSTAMP_FINE(6);
        if constexpr (SPREAD) {
          if (head) { young += store_batch(RB, 1); young += res_prefetch(RA, 2); }
          if (second) { young += store_batch(RB, 3); if (want_amax) commit_amax_asm(); }
        } else {
          if (head) {
            young += store_batch(RB, 3);
            if (want_amax) commit_amax_asm();
          }
        }
      }
      if (stamp < 48) PSTAMP(stamp);
      ++stamp;
      barrier();
      // ---------------- step (O, 2) ----------------
      if constexpr (CONS) {
        if (cdma) { wdma((B + 8) & 3, rw, 4); __builtin_amdgcn_sched_barrier(0); }
        DS_GROUP(8);
      } else {
        if (!cdma) wdma((B + 8) & 3, pw, NPW);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!IMG) activate_first_slots(halo, xrA, tagA, trowA, tscA, pk, 1);      // chunk O + 2 (fetched in (O, 0), rows parked in (O, 1))
        __builtin_amdgcn_sched_barrier(0);
        if ((SPREAD ? second : head) && stats && wv == 4) store_stats();  // the four waves' batch-3 partials are behind the barrier of (O, 1)
        if (last_of_item && it + 1 < n_items) bs_commit(bsb, bss, unscale_of(item_of(it + 1)), (it + 1) & 1);
      }
      if (stamp < 48) PSTAMP(stamp);
      ++stamp;
      barrier();
#undef DS_GROUP
    };

    // ---- start-up stagger: every workgroup runs the same schedule on the same amount of work, so unskewed they would all fetch,
    //      multiply and store in phase and the memory system would see the chip's whole demand in bursts; spread them over
    //      about one chunk pair's time ----
    {
      const unsigned skew = (blockIdx.x * 11u) & (unsigned)a.pc_skew_mask;
      for (unsigned i = 0; i < skew; ++i) __builtin_amdgcn_s_sleep(8);    // 8 x 64 clocks each
    }
    // ---- prologue: slabs 0, 1, 2, chunk 0 in X buffer 0, chunk 1 in flight, the first item's bias / shift row ----
    if constexpr (!CONS) {
      if constexpr (IMG) xdma(0);                     // chunk 0 -> X0 (chunk 1 follows in the first (E, 0))
      if (!cdma) { wdma(0, pw, NPW); wdma(1, pw, NPW); wdma(2, pw, NPW); }
      if constexpr (!IMG) {
        fetch(halo, xrB, prowB, tagB, trowB, tscB);
        fetch(halo, xrA, prowA, tagA, trowA, tscA);
      }
      bs_fetch(item_of(0), bsb, bss);
      if constexpr (VEC && PRE) {
        // both chunks' table rows into the pads of their X buffers, published by a barrier of their own (once per workgroup: the
        // full wait costs nothing that matters)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        rows_park(prowB, 0);
        rows_park(prowA, 1);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      }
      if constexpr (!IMG) {
        activate(halo, xrB, tagB, trowB, tscB, pk, 0);
        store_x(halo, pk, tagB, 0);
        activate_first_slots(halo, xrA, tagA, trowA, tscA, pk, 1);      // chunk 1: as every (O, 2) leaves it
      }
      bs_commit(bsb, bss, unscale_of(item_of(0)), 0);
      if constexpr (VEC || IMG) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(20)" ::: "memory");     // the three slabs and chunk 0; chunk 1's loads may stay in flight (the bias / shift loads are younger still)
    } else {
      if (cdma) { wdma(0, rw, 4); wdma(1, rw, 4); wdma(2, rw, 4); }
      if constexpr (VEC && PRE) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // the producers' table-row barrier
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[m][n][e] = 0.f;
    }
    PSTAMP(0);
    if constexpr (CONS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // (cdma) the first three slabs
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    young = 0; prev = (VEC || IMG) ? 0 : DS_LPI * XI;          // behind the prologue's slabs: chunk 1's loads (VEC: the prologue waited for everything)
    if constexpr (CONS && NPW == 4) load_pair(F0, 0, 0, 0, 0, 0, 0, 1, 0);         // P0 of the first chunk pair

    const int ncp = n_chunks >> 1;                    // even (the launcher's rule): fragment sets and ring slots are back in place after an item
    for (int it = 0; it < n_items; ++it) {
      for (int cp = 0; cp < ncp; cp += 2) {
        chunk_pair(std::integral_constant<int, 0>{}, F0, F1, cp == 0 && it > 0, false, false, it);
        if constexpr (NPW == 4) chunk_pair(std::integral_constant<int, 6>{}, F1, F0, false, cp == 0 && it > 0, cp + 2 == ncp, it);
        else chunk_pair(std::integral_constant<int, 6>{}, F0, F1, false, cp == 0 && it > 0, cp + 2 == ncp, it);
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
    }
    bar_counted(0);                                   // (eight producers: batch 1 writes the statistics rows batch 0 held back)
    if constexpr (!CONS) {
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
  if (consumer) {
    run(std::integral_constant<int, 0>{});
  } else {
    // the producers pace the kernel: let their vector / memory instructions win the issue arbitration against the consumer's
    // matrix stream on the same SIMD (MI355X_MICROARCH.md, Two waves per SIMD: priority, then age)
    if (a.pc_prio == 1) __builtin_amdgcn_s_setprio(1);
    else if (a.pc_prio == 2) __builtin_amdgcn_s_setprio(2);
    else if (a.pc_prio == 3) __builtin_amdgcn_s_setprio(3);
    if (VEC && pw >= 2) run(std::integral_constant<int, VEC ? 2 : 1>{});
    else run(std::integral_constant<int, 1>{});
  }
}

// DS_CONV_PC_WAVES = 4 | 8: producer waves per workgroup (A/B runs; the default is what measured faster)
int conv3p_producer_waves() {
  static const int v = [] { const char* e = getenv("DS_CONV_PC_WAVES"); const int n = e ? atoi(e) : 4; return n == 8 ? 8 : 4; }();
  return v;
}

template <bool PRE, bool CIRC, int NRES, int NPW, bool VEC = false, bool IMG = false>
int launch_conv3p_w(const Conv3hArgs& a, int wgs, hipStream_t s) {
  const int rc = ds::ensure_dynamic_lds<&k_conv3p<PRE, CIRC, NRES, NPW, VEC, IMG>>(P_LDS, "hipFuncSetAttribute(conv3p)");
  if (rc != DS_OK) return rc;
  hipLaunchKernelGGL((k_conv3p<PRE, CIRC, NRES, NPW, VEC, IMG>), dim3((unsigned)wgs), dim3(256 + 64 * NPW), P_LDS, s, a);
  DS_CHECK_LAUNCH("ds_conv2d_h3 (persistent)");
  return DS_OK;
}
// DS_CONV_VEC=0: the one-pixel staging items (A/B runs; the same switch as ds_conv3h.hip's)
bool conv3p_vec() {
  static const bool on = [] { const char* e = getenv("DS_CONV_VEC"); return !(e && atoi(e) == 0); }();
  return on;
}
template <bool PRE, bool CIRC, int NRES>
int launch_conv3p_r(const Conv3hArgs& a, int wgs, hipStream_t s) {
  if (conv3p_producer_waves() == 8) return launch_conv3p_w<PRE, CIRC, NRES, 8>(a, wgs, s);
  // 16-byte patch loads: W is a multiple of 32 and Cin of 64 here (conv3p_try_launch); no column tap offset, an aligned input
  if (conv3p_vec() && a.ox == 0 && (reinterpret_cast<uintptr_t>(a.in) & 15u) == 0) return launch_conv3p_w<PRE, CIRC, NRES, 4, true>(a, wgs, s);
  return launch_conv3p_w<PRE, CIRC, NRES, 4>(a, wgs, s);
}
void conv3p_fill_args(Conv3hArgs& a) {
  a.ntiles_magic40 = (1ull << 40) / (unsigned long long)(a.tiles_x * a.tiles_y) + 1ull;
  a.ncot_magic40 = (1ull << 40) / (unsigned long long)a.n_cot + 1ull;
  static const int prio = [] { const char* e = getenv("DS_CONV_PC_PRIO"); const int v = e ? atoi(e) : 0; return v < 0 || v > 3 ? 0 : v; }();
  a.pc_prio = prio;
  // start-up stagger: workgroup w sleeps ((11 w) & mask) x 512 cycles before its prologue (DS_CONV_PC_SKEW = the mask: 0, 1, 3, 7, 15, 31).
  // Default 0: inside the network a launch waits for its predecessor to drain, and the stagger is lost time at both ends (+0.7 % end
  // to end without it, profiles/r04_pc_ab_bench.log)
  static const int skew = [] { const char* e = getenv("DS_CONV_PC_SKEW"); const int v = e ? atoi(e) : 0; return v < 0 || v > 31 ? 0 : v; }();
  a.pc_skew_mask = skew;
  if (!a.res1 && a.res2) { a.res1 = a.res2; a.res2 = nullptr; }      // one residual: the order of the additions is the same
}
template <bool PRE, bool CIRC>
int launch_conv3p(Conv3hArgs a, int wgs, hipStream_t s) {
  conv3p_fill_args(a);
  if (!a.res1) return launch_conv3p_r<PRE, CIRC, 0>(a, wgs, s);
  return a.res2 ? launch_conv3p_r<PRE, CIRC, 2>(a, wgs, s) : launch_conv3p_r<PRE, CIRC, 1>(a, wgs, s);
}

// DS_CONV_PC: 0 = never, 1 = the fused-loader launches with one channel tile, 2 (default since the consumers issue the weight DMA and the
// producers' schedule is smooth: +3.4 % end to end for mode 1 and +1.2 % more for mode 2 on config 2, same-box A/B,
// profiles/r04_pc_ab_bench.log) = also in place of the two-channel-tile kernel, 3 = raw-input launches too.
// DS_CONV_PC_MIN: fewest items per workgroup (default 1 since the second session of round 4: at one or two items per workgroup the
// persistent kernel is equal to 21 % faster than the one-shot kernels -- profiles/r04_pc_min_items.log, batch 8 / 16 / 32 of
// config 2's shapes; it was 4 for the first session's kernel).
int conv3p_mode() {
  static const int v = [] { const char* e = getenv("DS_CONV_PC"); return e ? atoi(e) : 2; }();
  return v;
}
int conv3p_min_items() {
  static const int v = [] { const char* e = getenv("DS_CONV_PC_MIN"); const int n = e ? atoi(e) : 1; return n < 1 ? 1 : n; }();
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

// Image input (ds_conv2d_h3_img): DS_CONV_PC_IMG=0 keeps the one-shot kernel (A/B runs).
int conv3p_try_launch_img(const Conv3hArgs& a0, hipStream_t s, bool* launched) {
  *launched = false;
  static const bool on = [] { const char* e = getenv("DS_CONV_PC_IMG"); return !(e && atoi(e) == 0); }();
  if (!on || conv3p_mode() == 0 || conv3p_producer_waves() != 4) return DS_OK;
  if (a0.H % 8 != 0 || a0.W % 32 != 0 || a0.Cout % COT != 0 || a0.Cin % (4 * KC) != 0 || a0.res1_up) return DS_OK;
  const long long total = (long long)a0.n_cot * a0.tiles_x * a0.tiles_y * a0.B;
  const int wgs = conv3p_cus() / 8 * 8;
  if (wgs <= 0 || total < (long long)wgs * conv3p_min_items() || total >= (1ll << 22)) return DS_OK;
  if ((long long)4 * (a0.H + 2) * (a0.W + 2) * 16 >= (1ll << 31)) return DS_OK;       // the DMA's 32-bit lane offsets stay inside one (sample, chunk)
  Conv3hArgs a = a0;
  conv3p_fill_args(a);
  int rc;
  if (!a.res1) rc = launch_conv3p_w<false, false, 0, 4, false, true>(a, wgs, s);
  else rc = a.res2 ? launch_conv3p_w<false, false, 2, 4, false, true>(a, wgs, s) : launch_conv3p_w<false, false, 1, 4, false, true>(a, wgs, s);
  if (rc == DS_OK) *launched = true;
  return rc;
}

int conv3p_try_launch(const Conv3hArgs& a, hipStream_t s, bool* launched) {
  *launched = false;
  const int mode = conv3p_mode();
  if (mode == 0) return DS_OK;
  if (a.H % 8 != 0 || a.W % 32 != 0 || a.Cout % COT != 0 || a.Cin % (4 * KC) != 0) return DS_OK;      // an even number of chunk PAIRS (the consumer's two fragment sets)
  if (a.Hin != a.H || a.Win != a.W || a.res1_up) return DS_OK;
  const long long total = (long long)a.n_cot * a.tiles_x * a.tiles_y * a.B;
  const int wgs = conv3p_cus() / 8 * 8;               // a multiple of the XCD count: the item order assumes round-robin dealing
  if (wgs <= 0 || total < (long long)wgs * conv3p_min_items() || total >= (1ll << 22)) return DS_OK;      // 2^22: the exactness bound of the item decode
  if (a.n_cot % 2 == 0 && a.prenorm && mode < 2) return DS_OK;     // the two-channel-tile kernel's launches
  if (!a.prenorm && mode < 3) return DS_OK;           // raw-input launches: measured separately (DS_CONV_PC=3)
  int rc;
  if (a.prenorm) rc = a.circular ? launch_conv3p<true, true>(a, wgs, s) : launch_conv3p<true, false>(a, wgs, s);
  else rc = a.circular ? launch_conv3p<false, true>(a, wgs, s) : launch_conv3p<false, false>(a, wgs, s);
  if (rc == DS_OK) *launched = true;
  return rc;
}

}  // namespace ds_conv3
