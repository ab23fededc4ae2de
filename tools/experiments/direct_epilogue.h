// EXPERIMENT, measured negative, kept for the record (profiles/r03_b_direct_epilogue_ab.log: -1.4 % same-box,
// launch set 219.5 -> 223.5 us): the 16x16x32 epilogue WITHOUT the LDS transposition.  Not part of the product build.  To
// repeat the measurement: paste the function below into ds_conv_epilogue.h in front of store_tile16, make store_tile16 call it
// (store_tile16_direct<W16, MT>(acc, tile, bs, e); return;) and rebuild.
//
// Experiment: the 16x16x32 epilogue WITHOUT the LDS transposition.  Lane l (i = l & 15, g = l >> 4) holds, for its 16 (m, q)
// channels 16m + 4g + q, the four pixels n of column i: a store instruction covers four channels x 16 consecutive pixels (four
// 64-byte segments; the two halves of a 128-byte row segment follow each other), residuals are loaded the same way, the tile
// statistics reduce over the 16-lane rows (four channels per DPP sequence), nothing waits on an LDS round trip.  Any width
// (no 16-byte alignment to honour).  stat: wave-private LDS [32*MT channels][4] for store_tile_stats.
template <bool W16, int MT>
__device__ __forceinline__ void store_tile16_direct(const f32x4 (&acc)[2 * MT][4], float* stat, const float* bs, const Args& e) {
  const int lane = threadIdx.x & 63;
  const int i = lane & 15, g = lane >> 4;
  const unsigned plane = (unsigned)e.H * (unsigned)e.W;
  unsigned poff[4], roff[4];
  bool pok[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    const int row = e.y0 + (W16 ? n : (n >> 1)), col = e.x0 + (W16 ? i : 16 * (n & 1) + i);
    pok[n] = row < e.H && col < e.W;
    poff[n] = pok[n] ? (unsigned)row * (unsigned)e.W + (unsigned)col : 0u;
    roff[n] = pok[n] ? (unsigned)(row >> 1) * (unsigned)(e.W >> 1) + (unsigned)(col >> 1) : 0u;      // res1_up: [.., H/2, W/2]
  }
  const size_t lane_ch = 4u * (unsigned)g;                            // the lane's channel offset inside a 16-channel tile
  const bool stats = e.tile_stats != nullptr, want_amax = e.out_amax != nullptr;
  const size_t ch0 = (size_t)e.b * e.Cout + e.co_base;                // uniform
  float cnt = 0.f;
  if (stats) {
#pragma unroll
    for (int n = 0; n < 4; ++n) cnt += pok[n] ? 1.f : 0.f;
    cnt = row16_sum(cnt);
  }
  // residuals first: their latency hides behind the conversion of the accumulators
  float r1[2 * MT][4][4], r2[2 * MT][4][4];
  if (e.res1) {
    const unsigned rplane = e.res1_up ? (unsigned)(e.H >> 1) * (unsigned)(e.W >> 1) : plane;
#pragma unroll
    for (int m = 0; m < 2 * MT; ++m)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int co = 16 * m + q;                                    // + lane_ch
        const bool cok = e.co_base + co + (int)lane_ch < e.Cout;
        const size_t cidx = cok ? (ch0 + co + lane_ch) * rplane : (size_t)0;      // channels past Cout read element 0 (never used)
#pragma unroll
        for (int n = 0; n < 4; ++n) r1[m][q][n] = e.res1[cidx + (cok ? (e.res1_up ? roff[n] : poff[n]) : 0u)];
      }
  }
  if (e.res2) {
#pragma unroll
    for (int m = 0; m < 2 * MT; ++m)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int co = 16 * m + q;
        const bool cok = e.co_base + co + (int)lane_ch < e.Cout;
        const size_t cidx = cok ? (ch0 + co + lane_ch) * plane : (size_t)0;
#pragma unroll
        for (int n = 0; n < 4; ++n) r2[m][q][n] = e.res2[cidx + (cok ? poff[n] : 0u)];
      }
  }
  float amax = 0.f;
#pragma unroll
  for (int m = 0; m < 2 * MT; ++m)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int co = 16 * m + q;
      const bool cok = e.co_base + co + (int)lane_ch < e.Cout;
      const float bsv = bs[co + lane_ch] + bs[64 + co + lane_ch];
      float v[4];
#pragma unroll
      for (int n = 0; n < 4; ++n) v[n] = __builtin_fmaf(acc[m][n][q], e.unscale, bsv);
      if (e.res1) {
#pragma unroll
        for (int n = 0; n < 4; ++n) v[n] = v[n] + r1[m][q][n];
      }
      if (e.res2) {
#pragma unroll
        for (int n = 0; n < 4; ++n) v[n] = v[n] + r2[m][q][n];
      }
      const size_t cidx = (ch0 + co + lane_ch) * plane;
#pragma unroll
      for (int n = 0; n < 4; ++n)
        if (cok && pok[n]) e.out[cidx + poff[n]] = v[n];
      if (want_amax) {
#pragma unroll
        for (int n = 0; n < 4; ++n) amax = (cok && pok[n]) ? fmaxf(amax, __builtin_fabsf(v[n])) : amax;
      }
      if (stats) {
        // the 16 lanes of a row hold the channel's 64 pixels of this wave (4 each); lane i = 0 holds its first pixel (n = 0)
        const float K = row16_first((cok && pok[0]) ? v[0] : 0.f);
        float sv = 0.f, qv = 0.f;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          const float d = (cok && pok[n]) ? v[n] - K : 0.f;
          sv += d; qv += d * d;
        }
        sv = row16_sum(sv);
        qv = row16_sum(qv);
        if (i == 0) {
          f32x4 o = {K, sv, qv, cok ? cnt : 0.f};
          *reinterpret_cast<f32x4*>(&stat[4 * (co + (int)lane_ch)]) = o;
        }
      }
    }
  if (want_amax) commit_amax(e.out_amax, amax);
}

