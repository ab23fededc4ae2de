// Single-head self-attention on the fp16 matrix cores with fp32 accuracy ("fp16x3", the scheme of
// ds_conv3h.hip): every fp32 operand of the two products
//     S^T = K (Q/sqrt(E))^T          O^T += V^T P^T
// is split into fp16 hi + lo pieces and lo*hi + hi*lo + hi*hi are accumulated in fp32 by
// v_mfma_f32_32x32x16_f16.  Softmax statistics, the running maximum / sum and the rescaling of O
// stay in fp32 on the accumulators.  Domain: |q|, |k|, |v| < 65504 and not far below 1 -- or, with in_amax [2][B] (the
// per-sample max |q, k| and max |v| the in-projection left, ds_conv_epilogue.h), any magnitude: q and k are staged times
// 2^kqk, v times 2^kv, S is read back times 2^-2kqk inside the softmax's FMA and O times 2^-kv with the final 1 / l.
//
// Layout: operands arrive channel-major ([B, 3E, L], what the 1x1 projections write) and the
// output is channel-major ([B, E, L]).  The regrouping the MFMA wants happens while staging:
//   K tile  -> LDS [piece][d-group of 8][key 32][8 d]      (A operand of S^T: lane = key)
//   V tile  -> LDS [piece][key-group of 8][d E][8 keys]    (A operand of O^T: lane = d)
//   Q       -> registers, [k-slab of 16 d][piece]           (B operand of S^T: lane = query)
// each item being 8 global loads (a 128-byte-coalesced column walk) + one exact split + one
// ds_write_b128 per piece.  P never touches LDS: the 32x32 accumulator holds, for the lane's
// query, keys (r&3)+8(r>>2)+4h in register r of lane half h; packing pairs to fp16 and two
// v_permlane32_swap per 16-key slab and piece turn that into the B operand of the next MFMA.
//
// Two staging forms.  IMG = false: every workgroup loads its K / V tiles as fp32, splits them and writes the LDS
// images itself (no workspace; every tile is split once per 128-query block of the sample).  IMG = true: a pre-pass
// (k_attn_images) splits K and V ONCE per sample into global images that already have the LDS layout, and the
// attention kernel stages a tile with eight 1-KiB LDS-DMA instructions per wave and operand -- no staging registers,
// no split VALU work in the loop.  At L = 4096 a tile is otherwise re-split by 32 workgroups.
//
// Workgroup = 4 waves x 32 queries; key tiles of 32.  K and V tiles are double-buffered in LDS
// (4 x 32 KiB at E = 256) and staged one tile ahead through ONE set of 32 registers: the next K
// tile is loaded under the S^T product and written out before the softmax, the next V tile is
// loaded under the softmax / O^T product -- one barrier per tile.  Q (E/2 regs), O (E/2 regs)
// live in the 512-entry unified register file (one wave per SIMD).
#include <cstdlib>

#include "ds_common.h"
#include "ds_conv_epilogue.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

constexpr int NT = 256;
constexpr int KB = 32;       // keys per tile
constexpr float RESCALE_T = 8.0f;

// Exact-sum split of two fp32 values into packed fp16 hi / lo pairs.  The low piece MUST be the
// remainder against the very same rounded high piece that is stored: hipcc otherwise rounds the
// stored pair with v_cvt_pk_f16_f32 and the remainder's reference with v_cvt_f16_f32, and the two
// disagree on exact ties (measured on gfx950: hi + lo off by one fp16 ulp, 2^-11 relative, for one
// value in ~8000).  Deriving the reference from the packed bits removes the second rounding.
// e^x for the softmax numerators (x <= RESCALE_T): one FMA and the hardware exp2 (1 ulp; the argument's rounding adds
// |x| * 2^-24 relative, i.e. < 1e-6 wherever the result is not negligible) instead of libm's range-reduced expf
constexpr float LOG2E = 1.44269504088896341f;

__device__ __forceinline__ void split2(float a, float b, unsigned& hi, unsigned& lo) {
  f16x2 h = {(_Float16)a, (_Float16)b};
  unsigned hp = __builtin_bit_cast(unsigned, h);
  asm volatile("" : "+v"(hp));                       // opaque: both uses below see these exact bits
  const f16x2 hq = __builtin_bit_cast(f16x2, hp);
  f16x2 l = {(_Float16)(a - (float)hq[0]), (_Float16)(b - (float)hq[1])};
  hi = hp;
  lo = __builtin_bit_cast(unsigned, l);
}

__device__ __forceinline__ f16x8 as_f16x8(u32x4 v) { return __builtin_bit_cast(f16x8, v); }

template <int ET, bool IMG>
__global__ __launch_bounds__(NT) void k_attn3h(float* out, const float* __restrict__ qkv, const u32x4* __restrict__ kimg,
                                               const u32x4* __restrict__ vimg, int L, float scale, float thr,
                                               const unsigned* in_amax, unsigned* out_amax) {
  constexpr int E = 32 * ET;
  constexpr int NDG = E / 8;                         // d-groups of 8
  constexpr int NS = E / 16;                         // k-slabs of the S^T product
  constexpr int KITEMS = NDG * KB;                   // (d-group, key) items per K tile
  constexpr int KI = (KITEMS + NT - 1) / NT;
  constexpr int VITEMS = E * 4;                      // (d, key-group) items per V tile
  constexpr int VI = (VITEMS + NT - 1) / NT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int KVEC = 2 * NDG * KB;                 // vectors per K buffer  [piece][dg][key]
  constexpr int VVEC = 2 * 4 * E;                    // vectors per V buffer  [piece][kg][d]
  u32x4* Kbuf = reinterpret_cast<u32x4*>(smem);      // [2][KVEC]
  u32x4* Vbuf = Kbuf + 2 * KVEC;                     // [2][VVEC]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  // XCD-aware order (see ds_conv3h.hip): the query blocks of one sample stream the same K / V tiles; keep them on
  // one XCD so that after the first block the tiles come from its L2
  unsigned qblk, b_u;
  {
    const unsigned nx = gridDim.x;
    const unsigned id = blockIdx.x + nx * blockIdx.y;
    const unsigned total = nx * gridDim.y;
    const unsigned per = total >> 3, rem = total & 7u;
    const unsigned xcd = id & 7u, k = id >> 3;
    const unsigned logical = xcd * per + (xcd < rem ? xcd : rem) + k;
    qblk = logical % nx;
    b_u = logical / nx;
  }
  const int b = (int)b_u;
  const int q0_raw = ((int)qblk * 4 + wv) * 32;
  const bool active = q0_raw < L;                    // a wave past the last query block recomputes
  const int q0 = active ? q0_raw : L - 32;           // the last block and does not store (no branches
                                                     // around the MFMA pipeline: they cost registers)
  const float* Qt = qkv + (size_t)b * 3 * E * L;
  const float* Kt = Qt + (size_t)E * L;
  const float* Vt = Kt + (size_t)E * L;
  // the sample's activation exponents (0 without in_amax): q, k times 2^ak, v times 2^av; S times 2^-2ak, O times 2^-av -- all exact
  const int ak = ds_epi::act_exponent_of(ds_epi::act_bits(in_amax, b), -60, 60);
  const int av = ds_epi::act_exponent_of(ds_epi::act_bits(in_amax, (int)gridDim.y + b), -120, 120);
  const float a_in = ds_epi::pow2f(ak), v_in = ds_epi::pow2f(av);
  const float s_un = ds_epi::pow2f(-2 * ak), o_un = ds_epi::pow2f(-av);
  scale = ds_epi::mul_pow2(scale, ak);

  // ---- Q fragments: lane (query li, half lh) holds d = 16s + 8lh + 0..7, scaled, split ----
  u32x4 qh[NS], ql[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k)
      v[k] = Qt[(size_t)(16 * s + 8 * lh + k) * L + q0 + li] * scale;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      unsigned h, l;
      split2(v[2 * k], v[2 * k + 1], h, l);
      qh[s][k] = h; ql[s][k] = l;
    }
  }

  f32x16 O[ET];
#pragma unroll
  for (int t = 0; t < ET; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) O[t][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  static_assert(KI * 8 == VI * 8, "K and V staging share one register set");
  float sr[KI][8];                                   // staging registers (K tile, then V tile)
  // Addresses are (wave-uniform row base) + (32-bit per-lane offset): the scalar-base load form
  // keeps them out of the vector registers (hoisted 64-bit per-load addresses cost 80 VGPRs).
  unsigned koff[KI], voff[VI];
#pragma unroll
  for (int i = 0; i < KI; ++i) {
    const int e = tid + NT * i;
    const int dg = (e / KB) % NDG, key = e % KB;            // % NDG keeps tail lanes in bounds
    koff[i] = (unsigned)(8 * dg) * (unsigned)L + (unsigned)key;
  }
#pragma unroll
  for (int i = 0; i < VI; ++i) {
    const int e = tid + NT * i;
    const int d = (e >> 2) % E, kg = e & 3;
    voff[i] = (unsigned)d * (unsigned)L + (unsigned)(8 * kg);
  }
  auto k_load = [&](int key0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float* row = Kt + (size_t)k * L + key0;         // uniform
#pragma unroll
      for (int i = 0; i < KI; ++i) sr[i][k] = row[koff[i]];
    }
  };
  auto k_store = [&](u32x4* Ks) {
#pragma unroll
    for (int i = 0; i < KI; ++i) {
      const int e = tid + NT * i;
      if (NT * (i + 1) <= KITEMS || e < KITEMS) {
        u32x4 h, l;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          unsigned a, c;
          split2(sr[i][2 * k] * a_in, sr[i][2 * k + 1] * a_in, a, c);
          h[k] = a; l[k] = c;
        }
        Ks[e] = h;                         // e = dg*32 + key
        Ks[NDG * KB + e] = l;
      }
    }
  };
  auto v_load = [&](int key0) {
    const float* base = Vt + key0;                           // uniform
#pragma unroll
    for (int i = 0; i < VI; ++i) {
      const float* p = base + voff[i];
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(p);
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
      for (int k = 0; k < 4; ++k) { sr[i][k] = v0[k]; sr[i][4 + k] = v1[k]; }
    }
  };
  auto v_store = [&](u32x4* Vs) {
#pragma unroll
    for (int i = 0; i < VI; ++i) {
      const int e = tid + NT * i;
      if (NT * (i + 1) <= VITEMS || e < VITEMS) {
        const int d = e >> 2, kg = e & 3;
        u32x4 h, l;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          unsigned a, c;
          split2(sr[i][2 * k] * v_in, sr[i][2 * k + 1] * v_in, a, c);
          h[k] = a; l[k] = c;
        }
        Vs[kg * E + d] = h;
        Vs[4 * E + kg * E + d] = l;
      }
    }
  };

  const int nkb = L / KB;
  // IMG: tile kb of sample b as LDS-ready images; a tile = KVEC (= VVEC = 8E) vectors = 2*ET 1-KiB pieces per wave
  static_assert(KVEC == VVEC && KVEC % (64 * 4) == 0, "image tiles are whole LDS-DMA pieces per wave");
  const u32x4* kimg_b = IMG ? kimg + (size_t)b * nkb * KVEC : nullptr;
  const u32x4* vimg_b = IMG ? vimg + (size_t)b * nkb * VVEC : nullptr;
  auto dma = [&](u32x4* dst, const u32x4* src) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < KVEC / 256; ++i) {
      const int k = wv + 4 * i;                      // wave-uniform piece
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 64 * k + lane),
                                       (__attribute__((address_space(3))) void*)(dst + 64 * k), 16, 0, 0);
    }
  };
  // ---- the two matrix products and the softmax between them, shared by both staging forms ----
  auto qk = [&](const u32x4* Ks) __attribute__((always_inline)) {                    // S^T[key][query]
    // two accumulation chains (even / odd 16-wide d slabs), summed at the end: consecutive matrix instructions are independent
    f32x16 S, S1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { S[r] = 0.f; S1[r] = 0.f; }
#pragma unroll
    for (int s = 0; s < NS; s += 2) {
      const f16x8 ah = as_f16x8(Ks[(2 * s + lh) * KB + li]);
      const f16x8 al = as_f16x8(Ks[NDG * KB + (2 * s + lh) * KB + li]);
      const f16x8 bh = as_f16x8(Ks[(2 * s + 2 + lh) * KB + li]);
      const f16x8 bl = as_f16x8(Ks[NDG * KB + (2 * s + 2 + lh) * KB + li]);
      S = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, as_f16x8(qh[s]), S, 0, 0, 0);
      S1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl, as_f16x8(qh[s + 1]), S1, 0, 0, 0);
      S = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, as_f16x8(ql[s]), S, 0, 0, 0);
      S1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, as_f16x8(ql[s + 1]), S1, 0, 0, 0);
      S = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, as_f16x8(qh[s]), S, 0, 0, 0);
      S1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, as_f16x8(qh[s + 1]), S1, 0, 0, 0);
    }
    return S + S1;
  };
  if constexpr (IMG) {
    // Software pipeline (one wave per SIMD: nothing else hides the softmax): iteration kb issues the matrix
    // instructions of S(kb+1) and the vector instructions of softmax(S(kb)) as independent streams, then the O^T
    // product of tile kb.  K therefore runs one tile ahead of V: K(kb+2) -> K buffer kb&1 (last read for S(kb) in
    // iteration kb-1), V(kb+1) -> V buffer (kb+1)&1 (last read in iteration kb-1).
    dma(Kbuf, kimg_b);
    dma(Vbuf, vimg_b);
    if (nkb > 1) dma(Kbuf + KVEC, kimg_b + KVEC);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    f32x16 S = qk(Kbuf);
    for (int kb = 0; kb < nkb; ++kb) {
      const u32x4* Vs = Vbuf + (kb & 1) * VVEC;
      if (kb + 2 < nkb) dma(Kbuf + (kb & 1) * KVEC, kimg_b + (size_t)(kb + 2) * KVEC);
      if (kb + 1 < nkb) dma(Vbuf + ((kb + 1) & 1) * VVEC, vimg_b + (size_t)(kb + 1) * VVEC);
      f32x16 Sn;
      if (kb + 1 < nkb) Sn = qk(Kbuf + ((kb + 1) & 1) * KVEC);       // matrix pipe: independent of everything below up to the O^T product
      // ---- online softmax of tile kb (vector pipe) ----
      float mx = S[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(mx, S[r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * s_un;
      if (__any(mx > m_run + thr)) {
        const float m_new = fmaxf(m_run, mx);
        const float alpha = expf(m_run - m_new);
        l_run = l_run * alpha;
        m_run = m_new;
#pragma unroll
        for (int t = 0; t < ET; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) O[t][r] *= alpha;
      }
      float rs = 0.f;
      const float mneg = -m_run;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        // e^(S - m): the difference FIRST (one fma with the logits' power of two: exact for nearby values), then log2(e) and the
        // hardware exp2.  [fma(S, log2 e, -m log2 e) saves an instruction but rounds m log2 e on its own: at logits of 1e12 -- an
        // untrained network's -- that is off by 1e5, the exponent overflows and the sample turns into NaN where the reference's
        // softmax is a finite one-hot.]
        S[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(S[r], s_un, mneg) * LOG2E);
        rs += S[r];
      }
      rs += __shfl_xor(rs, 32, 64);
      l_run = l_run + rs;
      u32x4 ph[2], pl[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        unsigned h[4], l[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) split2(S[8 * t + 2 * k], S[8 * t + 2 * k + 1], h[k], l[k]);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          u32x2 r = __builtin_amdgcn_permlane32_swap(h[k], h[2 + k], false, false);
          h[k] = r[0]; h[2 + k] = r[1];
          r = __builtin_amdgcn_permlane32_swap(l[k], l[2 + k], false, false);
          l[k] = r[0]; l[2 + k] = r[1];
        }
        ph[t] = u32x4{h[0], h[1], h[2], h[3]};
        pl[t] = u32x4{l[0], l[1], l[2], l[3]};
      }
#pragma unroll
      for (int t = 0; t < ET; ++t) {
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
          const f16x8 vh = as_f16x8(Vs[(2 * sl + lh) * E + 32 * t + li]);
          const f16x8 vl = as_f16x8(Vs[4 * E + (2 * sl + lh) * E + 32 * t + li]);
          O[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, as_f16x8(ph[sl]), O[t], 0, 0, 0);
          O[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, as_f16x8(pl[sl]), O[t], 0, 0, 0);
          O[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, as_f16x8(ph[sl]), O[t], 0, 0, 0);
        }
        if ((t & 1) == 1) __builtin_amdgcn_sched_barrier(0);
      }
      if (kb + 1 < nkb) S = Sn;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  } else {
  if (IMG) {
    dma(Kbuf, kimg_b);
    dma(Vbuf, vimg_b);
  } else {
    k_load(0);
    k_store(Kbuf);
    v_load(0);
    v_store(Vbuf);
  }
  if (IMG) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kb = 0; kb < nkb; ++kb) {
    const bool more = kb + 1 < nkb;
    const u32x4* Ks = Kbuf + (kb & 1) * KVEC;
    const u32x4* Vs = Vbuf + (kb & 1) * VVEC;
    u32x4* Kn = Kbuf + ((kb + 1) & 1) * KVEC;
    u32x4* Vn = Vbuf + ((kb + 1) & 1) * VVEC;
    if (IMG) {
      // both next-tile buffers were last read one iteration ago (before the barrier that ended it): the whole
      // iteration covers the transfer, and the barrier at its end publishes it
      if (more) { dma(Kn, kimg_b + (size_t)(kb + 1) * KVEC); dma(Vn, vimg_b + (size_t)(kb + 1) * VVEC); }
    } else if (more) {
      k_load((kb + 1) * KB);                         // flies under the S^T product
    }
    f32x16 S;
    {
      // ---- S^T[key][query]: the same two chains, in the same order, as qk() above (the two forms agree bit for bit) ----
      f32x16 S1;
#pragma unroll
      for (int r = 0; r < 16; ++r) { S[r] = 0.f; S1[r] = 0.f; }
#pragma unroll
      for (int s = 0; s < NS; s += 2) {
        const f16x8 ah = as_f16x8(Ks[(2 * s + lh) * KB + li]);
        const f16x8 al = as_f16x8(Ks[NDG * KB + (2 * s + lh) * KB + li]);
        const f16x8 bh = as_f16x8(Ks[(2 * s + 2 + lh) * KB + li]);
        const f16x8 bl = as_f16x8(Ks[NDG * KB + (2 * s + 2 + lh) * KB + li]);
        S = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, as_f16x8(qh[s]), S, 0, 0, 0);
        S1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl, as_f16x8(qh[s + 1]), S1, 0, 0, 0);
        S = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, as_f16x8(ql[s]), S, 0, 0, 0);
        S1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, as_f16x8(ql[s + 1]), S1, 0, 0, 0);
        S = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, as_f16x8(qh[s]), S, 0, 0, 0);
        S1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, as_f16x8(qh[s + 1]), S1, 0, 0, 0);
        // bound the operand look-ahead: without a fence the scheduler hoists all 2*NS fragment
        // reads (128 registers at E = 256) in front of the first MFMA
        if ((s & 2) == 2) __builtin_amdgcn_sched_barrier(0);
      }
      S = S + S1;
    }
    if (!IMG && more) {
      k_store(Kn);                                   // the other K buffer: last read two tiles ago
      v_load((kb + 1) * KB);                         // flies under the softmax and the O^T product
    }
    {
      // ---- online softmax (fp32, per query = per lane; halves combined with one shuffle) ----
      float mx = S[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(mx, S[r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * s_un;
      // Deferred rescaling: O and l are kept relative to a reference maximum m_run that only moves
      // when some query's block maximum exceeds it by more than RESCALE_T (so P <= e^RESCALE_T,
      // far inside fp16's range for the hi piece).  The O-wide multiply then sits in a rarely
      // taken, wave-uniform branch instead of on every tile's critical path.
      if (__any(mx > m_run + thr)) {
        const float m_new = fmaxf(m_run, mx);
        const float alpha = expf(m_run - m_new);      // exp(-inf) = 0 on the first tile (O = 0, l = 0)
        l_run = l_run * alpha;
        m_run = m_new;
#pragma unroll
        for (int t = 0; t < ET; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) O[t][r] *= alpha;
      }
      float rs = 0.f;
      const float mneg = -m_run;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        // e^(S - m): the difference FIRST (one fma with the logits' power of two: exact for nearby values), then log2(e) and the
        // hardware exp2.  [fma(S, log2 e, -m log2 e) saves an instruction but rounds m log2 e on its own: at logits of 1e12 -- an
        // untrained network's -- that is off by 1e5, the exponent overflows and the sample turns into NaN where the reference's
        // softmax is a finite one-hot.]
        S[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(S[r], s_un, mneg) * LOG2E);
        rs += S[r];
      }
      rs += __shfl_xor(rs, 32, 64);
      l_run = l_run + rs;
      // ---- P^T as B operand: per 16-key slab, registers 8t..8t+7 -> [A0 A1 B0 B1] + half swap ----
      u32x4 ph[2], pl[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        unsigned h[4], l[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) split2(S[8 * t + 2 * k], S[8 * t + 2 * k + 1], h[k], l[k]);
        // h[0],h[1] = u=0 (keys 16t + 4*half + 0..3); h[2],h[3] = u=1 (keys 16t + 8 + 4*half + 0..3)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          u32x2 r = __builtin_amdgcn_permlane32_swap(h[k], h[2 + k], false, false);
          h[k] = r[0]; h[2 + k] = r[1];
          r = __builtin_amdgcn_permlane32_swap(l[k], l[2 + k], false, false);
          l[k] = r[0]; l[2 + k] = r[1];
        }
        ph[t] = u32x4{h[0], h[1], h[2], h[3]};
        pl[t] = u32x4{l[0], l[1], l[2], l[3]};
      }
      // ---- O^T[d][query] += V^T P^T ----
#pragma unroll
      for (int t = 0; t < ET; ++t) {
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
          const f16x8 vh = as_f16x8(Vs[(2 * sl + lh) * E + 32 * t + li]);
          const f16x8 vl = as_f16x8(Vs[4 * E + (2 * sl + lh) * E + 32 * t + li]);
          O[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, as_f16x8(ph[sl]), O[t], 0, 0, 0);
          O[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, as_f16x8(pl[sl]), O[t], 0, 0, 0);
          O[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, as_f16x8(ph[sl]), O[t], 0, 0, 0);
        }
        if ((t & 1) == 1) __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (!IMG && more) v_store(Vn);
    if (IMG) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's LDS-DMA pieces have landed ...
    __syncthreads();                                            // ... and so have everyone's
  }
  }   // staging form
  float amax = 0.f;
  if (active) {
    const float inv = (1.0f / l_run) * o_un;
    float* ob = out + (size_t)b * E * L;
#pragma unroll
    for (int t = 0; t < ET; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int d = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float o = O[t][r] * inv;
        ob[(size_t)d * L + q0 + li] = o;
        amax = fmaxf(amax, __builtin_fabsf(o));
      }
  }
  if (out_amax) ds_epi::commit_amax(out_amax + b, amax);
}

// Pre-pass of the IMG form: one workgroup per (key tile, sample) writes the tile's K and V images
//   K [piece][d-group][key 32][8 d]   V [piece][key-group][d][8 keys]        (the LDS layouts above)
// with the split every attention workgroup would otherwise redo.  Reads K, V once (8 B/elt), writes as many bytes.
template <int ET>
__global__ __launch_bounds__(NT) void k_attn_images(u32x4* __restrict__ kimg, u32x4* __restrict__ vimg,
                                                   const float* __restrict__ qkv, int L, const unsigned* in_amax) {
  constexpr int E = 32 * ET, NDG = E / 8;
  constexpr int KITEMS = NDG * KB, VITEMS = E * 4;
  const int tid = threadIdx.x, kb = blockIdx.x, b = blockIdx.y, nkb = L / KB;
  const float a_in = ds_epi::pow2f(ds_epi::act_exponent_of(ds_epi::act_bits(in_amax, b), -60, 60));       // the attention kernel's exponents
  const float v_in = ds_epi::pow2f(ds_epi::act_exponent_of(ds_epi::act_bits(in_amax, (int)gridDim.y + b), -120, 120));
  const float* Kt = qkv + ((size_t)b * 3 + 1) * E * L + (size_t)kb * KB;
  const float* Vt = Kt + (size_t)E * L;
  u32x4* Kd = kimg + ((size_t)b * nkb + kb) * (2 * KITEMS);
  u32x4* Vd = vimg + ((size_t)b * nkb + kb) * (2 * VITEMS);
  for (int e = tid; e < KITEMS; e += NT) {
    const int dg = e / KB, key = e % KB;
    u32x4 h, l;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      unsigned a, c;
      split2(Kt[(size_t)(8 * dg + 2 * k) * L + key] * a_in, Kt[(size_t)(8 * dg + 2 * k + 1) * L + key] * a_in, a, c);
      h[k] = a; l[k] = c;
    }
    Kd[e] = h;
    Kd[KITEMS + e] = l;
  }
  for (int e = tid; e < VITEMS; e += NT) {
    const int d = e >> 2, kg = e & 3;
    const float* p = Vt + (size_t)d * L + 8 * kg;
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(p) * v_in, v1 = *reinterpret_cast<const f32x4*>(p + 4) * v_in;
    u32x4 h, l;
    unsigned a, c;
    split2(v0[0], v0[1], a, c); h[0] = a; l[0] = c;
    split2(v0[2], v0[3], a, c); h[1] = a; l[1] = c;
    split2(v1[0], v1[1], a, c); h[2] = a; l[2] = c;
    split2(v1[2], v1[3], a, c); h[3] = a; l[3] = c;
    Vd[kg * E + d] = h;
    Vd[VITEMS + kg * E + d] = l;
  }
}

template <int ET, bool IMG>
int launch_attn3h(float* out, const float* qkv, void* workspace, int B, int L, float scale, const unsigned* in_amax,
                  unsigned* out_amax, hipStream_t s) {
  constexpr int E = 32 * ET;
  const size_t lds = (size_t)2 * (2 * (E / 8) * KB + 2 * 4 * E) * 16;    // K and V, double-buffered
  if (lds > 48 * 1024) {
    const int rc = ds::ensure_dynamic_lds<&k_attn3h<ET, IMG>>((int)lds, "hipFuncSetAttribute(attn3h)");
    if (rc != DS_OK) return rc;
  }
  u32x4* kimg = nullptr;
  u32x4* vimg = nullptr;
  if (IMG) {
    kimg = reinterpret_cast<u32x4*>(workspace);
    vimg = kimg + (size_t)B * L * (E / 4);                                // K images: B * (L/32) tiles * 8E vectors
    hipLaunchKernelGGL((k_attn_images<ET>), dim3(L / KB, B), dim3(NT), 0, s, kimg, vimg, qkv, L, in_amax);
    DS_CHECK_LAUNCH("ds_attention_h3 (images)");
  }
  dim3 g((L + 127) / 128, B);
  static const float thr = [] { const char* e = getenv("DS_ATTN_T"); return e ? (float)atof(e) : RESCALE_T; }();   // diagnostic knob
  hipLaunchKernelGGL((k_attn3h<ET, IMG>), g, dim3(NT), lds, s, out, qkv, kimg, vimg, L, scale, thr, in_amax, out_amax);
  DS_CHECK_LAUNCH("ds_attention_h3");
  return DS_OK;
}

template <bool IMG>
int attention_h3(float* out, const float* qkv, void* workspace, int B, int E, int L, const unsigned* in_amax, unsigned* out_amax,
                 void* stream) {
  DS_REQUIRE(out && qkv, DS_ERR_NULL, "ds_attention_h3: NULL pointer");
  DS_REQUIRE(B >= 0 && E > 0 && L > 0, DS_ERR_SHAPE, "ds_attention_h3: bad shape B=%d E=%d L=%d", B, E, L);
  DS_REQUIRE(L % 32 == 0, DS_ERR_UNSUPPORTED, "ds_attention_h3: L=%d must be a multiple of 32", L);
  DS_REQUIRE(B < 65536 && L / 32 < 65536 * 32, DS_ERR_SHAPE, "ds_attention_h3: B=%d exceeds grid.y", B);
  DS_REQUIRE((reinterpret_cast<uintptr_t>(qkv) & 15u) == 0, DS_ERR_SHAPE, "ds_attention_h3: qkv must be 16-byte aligned");
  DS_REQUIRE(!IMG || (workspace && (reinterpret_cast<uintptr_t>(workspace) & 15u) == 0), DS_ERR_NULL,
             "ds_attention_h3_ws: a 16-byte aligned workspace of ds_attention_h3_workspace_bytes(B, E, L) bytes is required");
  if (B == 0) return DS_OK;
  const float scale = (float)sqrt(1.0 / (double)E);
  hipStream_t s = ds::as_stream(stream);
  switch (E) {
    case 32: return launch_attn3h<1, IMG>(out, qkv, workspace, B, L, scale, in_amax, out_amax, s);
    case 64: return launch_attn3h<2, IMG>(out, qkv, workspace, B, L, scale, in_amax, out_amax, s);
    case 128: return launch_attn3h<4, IMG>(out, qkv, workspace, B, L, scale, in_amax, out_amax, s);
    case 256: return launch_attn3h<8, IMG>(out, qkv, workspace, B, L, scale, in_amax, out_amax, s);
    default:
      ds::set_error("ds_attention_h3: E=%d unsupported (32, 64, 128, 256)", E);
      return DS_ERR_UNSUPPORTED;
  }
}

}  // namespace

extern "C" int ds_attention_h3(float* out, const float* qkv, int B, int E, int L, const unsigned* in_amax, unsigned* out_amax,
                               void* stream) {
  return attention_h3<false>(out, qkv, nullptr, B, E, L, in_amax, out_amax, stream);
}

extern "C" size_t ds_attention_h3_workspace_bytes(int B, int E, int L) {
  if (B <= 0 || E <= 0 || L <= 0) return 0;
  return (size_t)B * L * E * 8;                      // K and V images: 2 pieces x 2 bytes per element, each
}

extern "C" int ds_attention_h3_ws(float* out, const float* qkv, void* workspace, int B, int E, int L, const unsigned* in_amax,
                                  unsigned* out_amax, void* stream) {
  return attention_h3<true>(out, qkv, workspace, B, E, L, in_amax, out_amax, stream);
}
