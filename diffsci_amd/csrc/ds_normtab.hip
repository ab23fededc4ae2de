// Normalisation folded into the consuming convolution (ds_conv2d_h3 `prenorm`): the producing
// convolution's epilogue leaves per-(sample, channel, pixel tile) shifted partial sums (`tile_stats`:
// float4 (K, S = sum(x-K), Q = sum((x-K)^2), n), ds_conv_epilogue.h); the tiny kernels here
// recombine them in fp64 (sum x = nK + S, sum x^2 = Q + 2KS + nK^2), in a fixed order, to the table
//   table[b][c] = (M, A, C, 0)   with   activation(x) = SiLU((x - M) * A + C)     (c < ceil16(C); zero rows past C)
// that the next convolution's loader applies while it stages its input -- the normalised tensor is
// never written to or read from HBM.
//   ds_inorm_table   PUNetG: GroupNorm(C,C) / GroupRMSNorm(C,C) per (sample, channel) plane
//                    (commonlayers.py:766-770, 372-384):   M = mean | 0,  A = rstd*w[c],  C = b[c];
//                    kind 2 (norm = Identity, commonlayers.py:891-899): (0, 1, 0)
//   ds_gnorm1_table  ADM: GroupNorm(1,C) / GroupRMSNorm(1,C)+FiLM per sample over (C,H,W), optionally
//                    over the channel concatenation of two tensors (adm.py:306-343, 385-406, 764-766):
//                    kind 0: M = mean_b, A = rstd_b*w[c], C = b[c];  kind 1: M = 0, A = w[c]/d_b, C = b[c];
//                    with FiLM rows (the block's second norm): A *= te1[b,c], C = C*te1[b,c] + te2[b,c]
// Fourth column: 2^-k, the sample's activation exponent for the consuming loader (0 = none).  U = max_c |A_c| sqrt(n m2_c) + |C_c|
// bounds every |(x - M)*A + C| of the sample (|x - mean| <= sqrt(n var), |x| <= sqrt(sum x^2)); k puts U at 2^13 and the loader
// produces SiLU(.) * 2^k at no cost (ds_h3_common.h: fast_silu_scaled), undone in the convolution's epilogue: SiLU(norm(x)) stays
// inside the fp16x3 window whatever the norm's affine parameters, FiLM rows or eps-dominated variances do
// (commonlayers.py:766-770: (x - mean)/sqrt(var + 1e-5) of a tensor of rms 1e-7 is 3e-5, not 1).  The value rides in the rows
// the loader reads anyway: no extra load, no atomics.
#include "ds_common.h"

namespace {

__device__ __forceinline__ double group_sum_d(double v, int width) {
  for (int o = width >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ void acc_tile(const float4 v, double& s, double& q) {
  const double K = v.x, S = v.y, Q = v.z, n = v.w;
  s += n * K + S;
  q += Q + 2.0 * K * S + n * K * K;
}

// 2^-k for a bound U on the activation's argument: U * 2^k in [2^13, 2^14); |k| <= 80 keeps 2^-(wshift + k) a normal float for
// every weight shift (|wshift| <= 40)
__device__ __forceinline__ float inv_scale_of(float U) {
  const unsigned bits = __builtin_bit_cast(unsigned, U);
  const int e = (int)((bits >> 23) & 0xffu);
  int k = (e == 0 || e == 255) ? 0 : 140 - e;
  k = k > 80 ? 80 : (k < -80 ? -80 : k);
  return __builtin_bit_cast(float, (unsigned)(127 - k) << 23);
}

// one workgroup per sample, 16 lanes per (b, c) plane, 64 planes at a time
constexpr int ITT = 1024;
__global__ __launch_bounds__(ITT) void k_inorm_table(float* table, const float* __restrict__ ts, const float* __restrict__ w,
                                                     const float* __restrict__ bias, int C, int Cpad, int ntiles,
                                                     double inv_n, float eps, int kind) {
  __shared__ unsigned smax;
  const int bb = blockIdx.x;
  const int l = threadIdx.x & 15, grp = threadIdx.x >> 4;
  if (threadIdx.x == 0) smax = 0u;
  __syncthreads();
  float4* rows = reinterpret_cast<float4*>(table) + (size_t)bb * Cpad;
  float U = 0.f;
  for (int c = grp; c < Cpad; c += ITT / 16) {
    const bool real = c < C;
    double s = 0.0, q = 0.0;
    if (real) {
      const float4* p = reinterpret_cast<const float4*>(ts) + ((size_t)bb * C + c) * ntiles;
      for (int t = l; t < ntiles; t += 16) acc_tile(p[t], s, q);
    }
    s = group_sum_d(s, 16);
    q = group_sum_d(q, 16);
    if (l != 0) continue;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);                  // rows with c >= C: the zero padding of the last 16-channel chunk
    if (real) {
      float M, rs;
      double dev2 = q;                          // sum of squared deviations from M: bounds every |x - M|^2
      if (kind == 0) {
        const double mean = s * inv_n;
        double var = q * inv_n - mean * mean;
        if (var < 0.0) var = 0.0;
        M = (float)mean;
        rs = 1.0f / sqrtf((float)var + eps);
        dev2 = var / inv_n;
      } else if (kind == 1) {
        M = 0.f;
        rs = 1.0f / sqrtf((float)(q * inv_n) + eps);
      } else {                                   // kind 2: no normalisation, the loader applies SiLU alone
        M = 0.f;
        rs = 1.0f;
      }
      o.x = M; o.y = rs * ((w && kind != 2) ? w[c] : 1.f); o.z = (bias && kind != 2) ? bias[c] : 0.f;
      U = fmaxf(U, fabsf(o.y) * (float)sqrt(dev2 > 0.0 ? dev2 : 0.0) * 1.0000005f + fabsf(o.z));
    }
    rows[c] = o;
  }
  if (l == 0 && U > 0.f) atomicMax(&smax, __builtin_bit_cast(unsigned, U));       // LDS
  __syncthreads();
  const float inv = inv_scale_of(__builtin_bit_cast(float, smax));
  for (int c = threadIdx.x; c < Cpad; c += ITT) reinterpret_cast<float*>(rows + c)[3] = inv;    // padded rows too: the loader reads any row's
}

// one workgroup of 1024 threads per sample: the tile statistics of one sample are up to 1 MiB (256 channels x 256
// tiles x 16 B at 256 x 256) and are read exactly once -- 256 threads took 30 us on them
constexpr int G1T = 1024;
__global__ __launch_bounds__(G1T) void k_gnorm1_table(float* table, const float* __restrict__ sa, int Ca, int nta,
                                                      const float* __restrict__ sb, int Cb, int ntb,
                                                      const float* __restrict__ w, const float* __restrict__ bias,
                                                      const float* __restrict__ f1, const float* __restrict__ f2,
                                                      int film_stride, double inv_n, float eps, int kind,
                                                      float* __restrict__ stats_out) {
  __shared__ double red[2][G1T / 64];
  __shared__ float st[3];
  __shared__ unsigned smax;
  if (threadIdx.x == 0) smax = 0u;
  const int b = blockIdx.x, C = Ca + Cb;
  double s = 0.0, q = 0.0;
  {
    const float4* p = reinterpret_cast<const float4*>(sa) + (size_t)b * Ca * nta;
    for (int i = threadIdx.x; i < Ca * nta; i += G1T) acc_tile(p[i], s, q);
  }
  if (Cb > 0) {
    const float4* p = reinterpret_cast<const float4*>(sb) + (size_t)b * Cb * ntb;
    for (int i = threadIdx.x; i < Cb * ntb; i += G1T) acc_tile(p[i], s, q);
  }
  s = group_sum_d(s, 64);
  q = group_sum_d(q, 64);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = q; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double ts = 0.0, tq = 0.0;                             // fixed order: deterministic
    for (int k = 0; k < G1T / 64; ++k) { ts += red[0][k]; tq += red[1][k]; }
    if (kind == 0) {
      const double mean = ts * inv_n;
      double var = tq * inv_n - mean * mean;
      if (var < 0.0) var = 0.0;
      st[0] = (float)mean;
      st[1] = 1.0f / sqrtf((float)var + eps);
      st[2] = (float)sqrt(var / inv_n) * 1.0000005f;
    } else {
      st[0] = 0.f;
      st[1] = 1.0f / sqrtf((float)(tq * inv_n) + eps);
      st[2] = (float)sqrt(tq > 0.0 ? tq : 0.0) * 1.0000005f;
    }
    if (stats_out) {                                       // ds_gnorm1_stats' convention: (mean, rstd) / (0, RMS denominator)
      stats_out[2 * b + 0] = st[0];
      stats_out[2 * b + 1] = kind == 0 ? st[1] : sqrtf((float)(tq * inv_n) + eps);
    }
  }
  if (!table) return;
  __syncthreads();
  const float M = st[0], rs = st[1], dev = st[2];
  float U = 0.f;
  const int Cpad = (C + 15) / 16 * 16;
  for (int c = C + threadIdx.x; c < Cpad; c += G1T) reinterpret_cast<float4*>(table)[(size_t)b * Cpad + c] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int c = threadIdx.x; c < C; c += G1T) {
    const float wc = w ? w[c] : 1.f, bc = bias ? bias[c] : 0.f;
    float4 o;
    o.x = M; o.y = rs * wc; o.z = bc;                       // M = 0 for the RMS norm
    if (f1) {                                               // FiLM: (n*w + b)*t1 + t2 = n*(w*t1) + (b*t1 + t2)
      const float t1 = f1[(size_t)b * film_stride + c], t2 = f2[(size_t)b * film_stride + c];
      o.y = rs * wc * t1; o.z = bc * t1 + t2;
    }
    o.w = 0.f;
    reinterpret_cast<float4*>(table)[(size_t)b * Cpad + c] = o;
    U = fmaxf(U, fabsf(o.y) * dev + fabsf(o.z));
  }
  for (int o = 32; o > 0; o >>= 1) U = fmaxf(U, __shfl_xor(U, o, 64));
  if ((threadIdx.x & 63) == 0 && U > 0.f) atomicMax(&smax, __builtin_bit_cast(unsigned, U));    // LDS
  __syncthreads();
  const float inv = inv_scale_of(__builtin_bit_cast(float, smax));
  for (int c = threadIdx.x; c < Cpad; c += G1T) reinterpret_cast<float*>(reinterpret_cast<float4*>(table) + (size_t)b * Cpad + c)[3] = inv;
}

}  // namespace

extern "C" {

int ds_inorm_table(float* table, const float* tile_stats, const float* w, const float* b, int B, int C, int ntiles,
                   int count, float eps, int kind, void* stream) {
  DS_REQUIRE(table && tile_stats, DS_ERR_NULL, "ds_inorm_table: NULL pointer");
  DS_REQUIRE(B >= 0 && C > 0 && ntiles > 0 && count > 0, DS_ERR_SHAPE, "ds_inorm_table: bad shape");
  DS_REQUIRE(kind >= 0 && kind <= 2, DS_ERR_UNSUPPORTED, "ds_inorm_table: kind %d (0 GroupLN, 1 GroupRMS, 2 none)", kind);
  DS_REQUIRE((long long)B * C < (1ll << 31), DS_ERR_SHAPE, "ds_inorm_table: too many planes");
  DS_REQUIRE((reinterpret_cast<uintptr_t>(table) & 15u) == 0 && (reinterpret_cast<uintptr_t>(tile_stats) & 15u) == 0,
             DS_ERR_SHAPE, "ds_inorm_table: misaligned pointer");
  if (B == 0) return DS_OK;
  const int Cpad = (C + 15) / 16 * 16;
  DS_REQUIRE(B < 65536 * 32, DS_ERR_SHAPE, "ds_inorm_table: B=%d", B);
  hipLaunchKernelGGL(k_inorm_table, dim3((unsigned)B), dim3(ITT), 0, ds::as_stream(stream), table, tile_stats, w,
                     b, C, Cpad, ntiles, 1.0 / (double)count, eps, kind);
  DS_CHECK_LAUNCH("ds_inorm_table");
  return DS_OK;
}

int ds_gnorm1_table(float* table, const float* stats_a, int Ca, int ntiles_a, const float* stats_b, int Cb,
                    int ntiles_b, const float* w, const float* b, const float* film_scale, const float* film_shift,
                    int film_stride, int B, long long count, float eps, int kind, void* stream) {
  DS_REQUIRE(table && stats_a, DS_ERR_NULL, "ds_gnorm1_table: NULL pointer");
  DS_REQUIRE(B >= 0 && Ca > 0 && ntiles_a > 0 && Cb >= 0 && count > 0, DS_ERR_SHAPE, "ds_gnorm1_table: bad shape");
  DS_REQUIRE(Cb == 0 || (stats_b && ntiles_b > 0), DS_ERR_NULL, "ds_gnorm1_table: second source missing");
  DS_REQUIRE(kind == 0 || kind == 1, DS_ERR_UNSUPPORTED, "ds_gnorm1_table: kind %d", kind);
  DS_REQUIRE((film_scale == nullptr) == (film_shift == nullptr), DS_ERR_NULL, "ds_gnorm1_table: FiLM scale and shift go together");
  DS_REQUIRE(((reinterpret_cast<uintptr_t>(table) | reinterpret_cast<uintptr_t>(stats_a) | reinterpret_cast<uintptr_t>(stats_b)) & 15u) == 0,
             DS_ERR_SHAPE, "ds_gnorm1_table: pointers must be 16-byte aligned");
  if (B == 0) return DS_OK;
  hipLaunchKernelGGL(k_gnorm1_table, dim3(B), dim3(G1T), 0, ds::as_stream(stream), table, stats_a, Ca, ntiles_a, stats_b,
                     Cb, ntiles_b, w, b, film_scale, film_shift, film_stride, 1.0 / (double)count, eps, kind, (float*)nullptr);
  DS_CHECK_LAUNCH("ds_gnorm1_table");
  return DS_OK;
}

/* ds_gnorm1_stats without reading the tensor: the (mean, rstd) / (0, RMS denominator) pairs of ds_gnorm1_apply[_images] from the
 * tile statistics its producer(s) left (one convolution output, or the channel concatenation of two). */
int ds_gnorm1_stats_tiles(float* stats, const float* stats_a, int Ca, int ntiles_a, const float* stats_b, int Cb, int ntiles_b,
                          int B, long long count, float eps, int kind, void* stream) {
  DS_REQUIRE(stats && stats_a, DS_ERR_NULL, "ds_gnorm1_stats_tiles: NULL pointer");
  DS_REQUIRE(B >= 0 && Ca > 0 && ntiles_a > 0 && Cb >= 0 && count > 0, DS_ERR_SHAPE, "ds_gnorm1_stats_tiles: bad shape");
  DS_REQUIRE(Cb == 0 || (stats_b && ntiles_b > 0), DS_ERR_NULL, "ds_gnorm1_stats_tiles: second source missing");
  DS_REQUIRE(kind == 0 || kind == 1, DS_ERR_UNSUPPORTED, "ds_gnorm1_stats_tiles: kind %d", kind);
  DS_REQUIRE(((reinterpret_cast<uintptr_t>(stats_a) | reinterpret_cast<uintptr_t>(stats_b)) & 15u) == 0, DS_ERR_SHAPE,
             "ds_gnorm1_stats_tiles: tile statistics must be 16-byte aligned");
  if (B == 0) return DS_OK;
  hipLaunchKernelGGL(k_gnorm1_table, dim3(B), dim3(G1T), 0, ds::as_stream(stream), (float*)nullptr, stats_a, Ca, ntiles_a,
                     stats_b, Cb, ntiles_b, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr,
                     (const float*)nullptr, 0, 1.0 / (double)count, eps, kind, stats);
  DS_CHECK_LAUNCH("ds_gnorm1_stats_tiles");
  return DS_OK;
}

/* Number of pixel tiles per channel plane that ds_conv2d_h3 / ds_conv1x1_h3 use for an H x W output
 * (the `ntiles` of their tile_stats layout). */
int ds_conv_tile_count(int H, int W) {
  if (H <= 0 || W <= 0) return 0;
  const long long pad32 = (long long)((W + 31) / 32 * 32) * ((H + 7) / 8 * 8);
  const long long pad16 = (long long)((W + 15) / 16 * 16) * ((H + 15) / 16 * 16);
  return pad16 < pad32 ? ((W + 15) / 16) * ((H + 15) / 16) : ((W + 31) / 32) * ((H + 7) / 8);
}

}  // extern "C"
