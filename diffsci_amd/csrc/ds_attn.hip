// Single-head self-attention over the L = H*W positions of the UNet bottleneck, fp32 MFMA,
// flash-style (no L x L score matrix in memory).  nn.MultiheadAttention(E, num_heads=1) core:
//   q_scaled = q / sqrt(E); P = softmax(q_scaled k^T); out = P v        (attention.py:41-43,67)
//
// Everything is kept channel-major ([E][L], i.e. NCHW), which is what the 1x1-convolution
// projections produce and consume, and which makes every MFMA operand a contiguous row read:
//   S^T[key][query] = sum_d K^T[d][key] * Q^T[d][query]     A = K^T (LDS), B = Q^T (registers)
//   O^T[d][query]  += sum_key V^T[d][key] * P^T[key][query]  A = V^T (LDS, rows padded to 33),
//                                                            B = the S^T accumulator itself
// The 32x32 accumulator of v_mfma_f32_32x32x2_f32 holds column `query` on the lane and keys
// (r&3)+8(r>>2)+4h in register r of lane half h, which is exactly the (k = h) operand slot of the
// next MFMA -- so P never moves between lanes or through LDS; the V^T operand is fetched in the
// matching key order.  Softmax statistics are per lane (one query per lane, two lane halves
// combined with one cross-half shuffle).
//
// Workgroup = 4 waves x 32 queries; K^T / V^T tiles of 32 keys are staged global -> registers ->
// LDS one tile ahead.  Q^T (E/2 registers), O^T (E/2 registers) and S^T (16) live in the unified
// 512-entry register file (one wave per SIMD).
#include "ds_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));   // native vector: HIP's float4 struct defeats SROA here
constexpr int NT = 256;
constexpr int KB = 32;       // keys per tile
constexpr int VSTR = KB + 1; // padded V^T row

// ET = E/32; ETO = output channels per workgroup / 32.  ETO < ET (E > 256, where Q^T alone takes
// half the register file) splits the output channels over blockIdx.z; every such workgroup
// recomputes the full-E scores, which is cheap at the small L these wide bottlenecks have.
template <int ET, int ETO>
__global__ __launch_bounds__(NT) void k_attn(float* out, const float* __restrict__ qkv, int L, float scale) {
  constexpr int E = 32 * ET;
  constexpr int EO = 32 * ETO;
  constexpr int NLD = (E * KB / 4) / NT;   // float4 per thread per K^T tile (E*8/256 = ET)
  constexpr int NLV = (EO * KB / 4) / NT;  // ... per V^T tile
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ks = smem;              // [E][32]
  float* Vs = smem + E * KB;     // [EO][33]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int b = blockIdx.y;
  const int q0 = (blockIdx.x * 4 + wv) * 32;
  const bool active = q0 < L;
  const float* Qt = qkv + (size_t)b * 3 * E * L;
  const float* Kt = Qt + (size_t)E * L;
  const float* Vt = Kt + (size_t)E * L + (size_t)blockIdx.z * EO * L;

  float qreg[E / 2];
#pragma unroll
  for (int s = 0; s < E / 2; ++s) qreg[s] = active ? Qt[(size_t)(2 * s + lh) * L + q0 + li] * scale : 0.f;

  f32x16 O[ETO];
#pragma unroll
  for (int t = 0; t < ETO; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) O[t][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  f32x4 kr[NLD], vr[NLV];
  auto tile_load = [&](int key0) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int e = tid + NT * i;          // float4 index in [E][8]
      const int d = e >> 3, c4 = e & 7;
      kr[i] = *reinterpret_cast<const f32x4*>(Kt + (size_t)d * L + key0 + 4 * c4);
    }
#pragma unroll
    for (int i = 0; i < NLV; ++i) {
      const int e = tid + NT * i;
      const int d = e >> 3, c4 = e & 7;
      vr[i] = *reinterpret_cast<const f32x4*>(Vt + (size_t)d * L + key0 + 4 * c4);
    }
  };
  auto tile_store = [&]() {
#pragma unroll
    for (int i = 0; i < NLD; ++i) reinterpret_cast<f32x4*>(Ks)[tid + NT * i] = kr[i];
#pragma unroll
    for (int i = 0; i < NLV; ++i) {
      const int e = tid + NT * i;
      const int d = e >> 3, c4 = e & 7;
      float* vd = Vs + d * VSTR + 4 * c4;
      vd[0] = vr[i].x; vd[1] = vr[i].y; vd[2] = vr[i].z; vd[3] = vr[i].w;
    }
  };

  const int nkb = L / KB;
  tile_load(0);
  tile_store();
  __syncthreads();
  for (int kb = 0; kb < nkb; ++kb) {
    const bool more = kb + 1 < nkb;
    if (more) tile_load((kb + 1) * KB);
    if (active) {
      f32x16 S;
#pragma unroll
      for (int r = 0; r < 16; ++r) S[r] = 0.f;
      const float* kl = Ks + lh * KB + li;
#pragma unroll
      for (int s = 0; s < E / 2; ++s) S = __builtin_amdgcn_mfma_f32_32x32x2f32(kl[2 * s * KB], qreg[s], S, 0, 0, 0);
      float mx = S[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(mx, S[r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run, mx);
      const float alpha = expf(m_run - m_new);
      float rs = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        S[r] = expf(S[r] - m_new);
        rs += S[r];
      }
      rs += __shfl_xor(rs, 32, 64);
      l_run = l_run * alpha + rs;
      m_run = m_new;
#pragma unroll
      for (int t = 0; t < ETO; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[t][r] *= alpha;
#pragma unroll
      for (int t = 0; t < ETO; ++t) {
        const float* vl = Vs + (32 * t + li) * VSTR + 4 * lh;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          O[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(vl[(r & 3) + 8 * (r >> 2)], S[r], O[t], 0, 0, 0);
      }
    }
    __syncthreads();
    if (more) {
      tile_store();
      __syncthreads();
    }
  }
  if (active) {
    const float inv = 1.0f / l_run;
    float* ob = out + (size_t)b * E * L + (size_t)blockIdx.z * EO * L;
#pragma unroll
    for (int t = 0; t < ETO; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int d = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * lh;
        ob[(size_t)d * L + q0 + li] = O[t][r] * inv;
      }
  }
}

template <int ET, int ETO = ET>
int launch_attn(float* out, const float* qkv, int B, int L, float scale, hipStream_t s) {
  constexpr int E = 32 * ET;
  const size_t lds = (size_t)(E * KB + 32 * ETO * VSTR) * sizeof(float);
  if (lds > 48 * 1024) {
    const int rc = ds::ensure_dynamic_lds<&k_attn<ET, ETO>>((int)lds, "hipFuncSetAttribute(attn)");
    if (rc != DS_OK) return rc;
  }
  dim3 g((L + 127) / 128, B, ET / ETO);
  hipLaunchKernelGGL((k_attn<ET, ETO>), g, dim3(NT), lds, s, out, qkv, L, scale);
  DS_CHECK_LAUNCH("ds_attention");
  return DS_OK;
}

}  // namespace

extern "C" int ds_attention(float* out, const float* qkv, int B, int E, int L, void* stream) {
  DS_REQUIRE(out && qkv, DS_ERR_NULL, "ds_attention: NULL pointer");
  DS_REQUIRE(B >= 0 && E > 0 && L > 0, DS_ERR_SHAPE, "ds_attention: bad shape B=%d E=%d L=%d", B, E, L);
  DS_REQUIRE(L % 32 == 0, DS_ERR_UNSUPPORTED, "ds_attention: L=%d must be a multiple of 32", L);
  DS_REQUIRE(B < 65536, DS_ERR_SHAPE, "ds_attention: B=%d exceeds grid.y", B);
  DS_REQUIRE((reinterpret_cast<uintptr_t>(qkv) & 15u) == 0, DS_ERR_SHAPE, "ds_attention: qkv must be 16-byte aligned");
  if (B == 0) return DS_OK;
  const float scale = (float)sqrt(1.0 / (double)E);   // math.sqrt(1.0/E) cast to fp32, as torch does
  hipStream_t s = ds::as_stream(stream);
  switch (E) {
    case 32: return launch_attn<1>(out, qkv, B, L, scale, s);
    case 64: return launch_attn<2>(out, qkv, B, L, scale, s);
    case 128: return launch_attn<4>(out, qkv, B, L, scale, s);
    case 256: return launch_attn<8>(out, qkv, B, L, scale, s);
    case 384: return launch_attn<12, 4>(out, qkv, B, L, scale, s);
    case 512: return launch_attn<16, 2>(out, qkv, B, L, scale, s);
    default:
      ds::set_error("ds_attention: E=%d unsupported (32, 64, 128, 256, 384, 512)", E);
      return DS_ERR_UNSUPPORTED;
  }
}
