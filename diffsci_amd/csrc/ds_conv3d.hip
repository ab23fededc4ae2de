// 3x3x3 'same' convolution for volumes [B, C, D, H, W] (PUNetG with dimension = 3: torch.nn.Conv3d / CircularConv3d /
// MagnitudePreservingConv3d, commonlayers.py:25-160,973-1040, normedlayers.py:58-92), exact fp32 FMA chains on the
// vector ALUs.  First, correctness-first version of the 3-D path: the 2-D networks' convolutions run on the matrix
// cores (ds_conv3h.hip); volumes get there next.  Same fusions as the 2-D kernels so that the 3-D network never
// materialises a pooled / upsampled tensor: MaxPool3d(2) or nearest x2 upsampling in the loader, zero or periodic
// padding, bias / time shift / up to two residuals in the epilogue.
//
// Workgroup = 256 threads = a 4 x 8 x 8 (z, y, x) block of output voxels x 16 output channels.  Input channels are
// walked four at a time: their 6 x 10 x 10 halo patches and the 4 x 27 x 16 weights are staged in LDS; every
// thread then reads its 27 taps once per channel and reuses each across the 16 output channels (weights come
// as 16-byte broadcast reads).
#include "ds_common.h"

namespace {

constexpr int NT = 256;
constexpr int TZ = 4, TY = 8, TX = 8;
constexpr int PZ = TZ + 2, PY = TY + 2, PX = TX + 2;
constexpr int PVOL = PZ * PY * PX;                  // 600 floats per channel
constexpr int CO = 16, CC = 4;

struct C3Args {
  float* out;
  const float* in;
  const float* w;        // torch layout [Cout][Cin][3][3][3]
  const float* bias;
  const float* shift;
  const float* res1;
  const float* res2;
  int shift_stride;
  int B, Cin, Cout, D, H, W;      // output volume
  int Di, Hi, Wi;                 // input volume (2x for MAXPOOL2, /2 for UPSAMPLE2)
  int tiles_y, tiles_x;
};

__device__ __forceinline__ int wrap(int g, int n) {
  g = g < 0 ? g + n : g;
  return g >= n ? g - n : g;
}

template <int MODE, bool CIRC>
__global__ __launch_bounds__(NT) void k_conv3d(const C3Args a) {
  __shared__ __attribute__((aligned(16))) float patch[CC][PVOL];
  __shared__ __attribute__((aligned(16))) float wts[CC][27][CO];
  const int t = threadIdx.x;
  const int tx = t & 7, ty = (t >> 3) & 7, tz = t >> 6;
  int tile = blockIdx.x;
  const int bx = tile % a.tiles_x; tile /= a.tiles_x;
  const int by = tile % a.tiles_y;
  const int bz = tile / a.tiles_y;
  const int x0 = bx * TX, y0 = by * TY, z0 = bz * TZ;
  const int co0 = blockIdx.y * CO;
  const int b = blockIdx.z;
  const size_t in_plane = (size_t)a.Di * a.Hi * a.Wi;
  const float* in_b = a.in + (size_t)b * a.Cin * in_plane;

  float acc[CO];
#pragma unroll
  for (int k = 0; k < CO; ++k) acc[k] = 0.f;

  for (int c0 = 0; c0 < a.Cin; c0 += CC) {
    // ---- stage the halo patches of up to CC input channels ----
    for (int i = t; i < CC * PVOL; i += NT) {
      const int c = i / PVOL;
      int r = i - c * PVOL;
      const int pz = r / (PY * PX); r -= pz * (PY * PX);
      const int py = r / PX;
      const int px = r - py * PX;
      int gz = z0 + pz - 1, gy = y0 + py - 1, gx = x0 + px - 1;
      float v = 0.f;
      if (c0 + c < a.Cin) {
        if (CIRC) {
          // a ragged tile may reach more than one voxel past the volume: those positions only feed outputs that are
          // never stored; clamp after the single wrap
          gz = wrap(gz, a.D); gy = wrap(gy, a.H); gx = wrap(gx, a.W);
          gz = gz >= a.D ? a.D - 1 : gz; gy = gy >= a.H ? a.H - 1 : gy; gx = gx >= a.W ? a.W - 1 : gx;
        }
        if (gz >= 0 && gz < a.D && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
          const float* p = in_b + (size_t)(c0 + c) * in_plane;
          if (MODE == DS_LOAD_PLAIN) {
            v = p[((size_t)gz * a.Hi + gy) * a.Wi + gx];
          } else if (MODE == DS_LOAD_UPSAMPLE2) {
            v = p[((size_t)(gz >> 1) * a.Hi + (gy >> 1)) * a.Wi + (gx >> 1)];
          } else {                                            // MaxPool3d(2): 2 x 2 x 2 window
            const float* q = p + ((size_t)(2 * gz) * a.Hi + 2 * gy) * a.Wi + 2 * gx;
            const size_t sz = (size_t)a.Hi * a.Wi;
            float m0 = fmaxf(fmaxf(q[0], q[1]), fmaxf(q[a.Wi], q[a.Wi + 1]));
            float m1 = fmaxf(fmaxf(q[sz], q[sz + 1]), fmaxf(q[sz + a.Wi], q[sz + a.Wi + 1]));
            v = fmaxf(m0, m1);
          }
        }
      }
      patch[c][(pz * PY + py) * PX + px] = v;
    }
    // ---- stage the weights [c][tap][co] ----
    for (int i = t; i < CC * 27 * CO; i += NT) {
      const int co = i / (CC * 27);
      const int r = i - co * (CC * 27);
      const int c = r / 27, tap = r - c * 27;
      float v = 0.f;
      if (co0 + co < a.Cout && c0 + c < a.Cin) v = a.w[((size_t)(co0 + co) * a.Cin + c0 + c) * 27 + tap];
      wts[c][tap][co] = v;
    }
    __syncthreads();
#pragma unroll 1
    for (int c = 0; c < CC; ++c) {
#pragma unroll
      for (int dz = 0; dz < 3; ++dz)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) {
            const float v = patch[c][((tz + dz) * PY + ty + dy) * PX + tx + dx];
            const float4* wv = reinterpret_cast<const float4*>(&wts[c][(dz * 3 + dy) * 3 + dx][0]);
#pragma unroll
            for (int q = 0; q < CO / 4; ++q) {
              const float4 w4 = wv[q];
              acc[4 * q + 0] = __builtin_fmaf(v, w4.x, acc[4 * q + 0]);
              acc[4 * q + 1] = __builtin_fmaf(v, w4.y, acc[4 * q + 1]);
              acc[4 * q + 2] = __builtin_fmaf(v, w4.z, acc[4 * q + 2]);
              acc[4 * q + 3] = __builtin_fmaf(v, w4.w, acc[4 * q + 3]);
            }
          }
    }
    __syncthreads();
  }

  const int gz = z0 + tz, gy = y0 + ty, gx = x0 + tx;
  if (gz >= a.D || gy >= a.H || gx >= a.W) return;
  const size_t plane = (size_t)a.D * a.H * a.W;
  const size_t vox = ((size_t)gz * a.H + gy) * a.W + gx;
#pragma unroll
  for (int k = 0; k < CO; ++k) {
    const int co = co0 + k;
    if (co >= a.Cout) break;
    const size_t idx = ((size_t)b * a.Cout + co) * plane + vox;
    float v = acc[k];
    v = v + (a.bias ? a.bias[co] : 0.f);
    v = v + (a.shift ? a.shift[(size_t)b * a.shift_stride + co] : 0.f);
    if (a.res1) v = v + a.res1[idx];
    if (a.res2) v = v + a.res2[idx];
    a.out[idx] = v;
  }
}

template <int MODE>
int launch3d(const C3Args& a, bool circ, hipStream_t s) {
  const long long tiles = (long long)((a.D + TZ - 1) / TZ) * a.tiles_y * a.tiles_x;
  const int cots = (a.Cout + CO - 1) / CO;
  DS_REQUIRE(tiles < (1ll << 31) && cots < 65536 && a.B < 65536, DS_ERR_SHAPE, "ds_conv3d_direct: grid too large");
  dim3 g((unsigned)tiles, (unsigned)cots, (unsigned)a.B);
  if (circ) hipLaunchKernelGGL((k_conv3d<MODE, true>), g, dim3(NT), 0, s, a);
  else hipLaunchKernelGGL((k_conv3d<MODE, false>), g, dim3(NT), 0, s, a);
  DS_CHECK_LAUNCH("ds_conv3d_direct");
  return DS_OK;
}

}  // namespace

extern "C" int ds_conv3d_direct(float* out, const float* in, const float* w, const float* bias, const float* shift,
                                int shift_stride, const float* res1, const float* res2, int B, int Cin, int Cout, int D,
                                int H, int W, int load_mode, void* stream) {
  DS_REQUIRE(out && in && w, DS_ERR_NULL, "ds_conv3d_direct: NULL pointer");
  DS_REQUIRE(B >= 0 && Cin > 0 && Cout > 0 && D > 0 && H > 0 && W > 0, DS_ERR_SHAPE,
             "ds_conv3d_direct: bad shape B=%d Cin=%d Cout=%d D=%d H=%d W=%d", B, Cin, Cout, D, H, W);
  const bool circ = (load_mode & DS_PAD_CIRCULAR) != 0;
  load_mode &= ~DS_PAD_CIRCULAR;
  DS_REQUIRE(load_mode == DS_LOAD_PLAIN || load_mode == DS_LOAD_MAXPOOL2 || load_mode == DS_LOAD_UPSAMPLE2,
             DS_ERR_UNSUPPORTED, "ds_conv3d_direct: load_mode %d", load_mode);
  DS_REQUIRE(load_mode != DS_LOAD_UPSAMPLE2 || (D % 2 == 0 && H % 2 == 0 && W % 2 == 0), DS_ERR_SHAPE,
             "ds_conv3d_direct: UPSAMPLE2 needs an even output volume (got %d x %d x %d)", D, H, W);
  DS_REQUIRE(shift == nullptr || shift_stride == 0 || shift_stride >= Cout, DS_ERR_SHAPE,
             "ds_conv3d_direct: shift_stride %d < Cout %d", shift_stride, Cout);
  if (B == 0) return DS_OK;
  C3Args a;
  a.out = out; a.in = in; a.w = w; a.bias = bias; a.shift = shift; a.res1 = res1; a.res2 = res2;
  a.shift_stride = shift_stride;
  a.B = B; a.Cin = Cin; a.Cout = Cout; a.D = D; a.H = H; a.W = W;
  const int f2 = load_mode == DS_LOAD_MAXPOOL2 ? 2 : 1;
  a.Di = load_mode == DS_LOAD_UPSAMPLE2 ? D / 2 : D * f2;
  a.Hi = load_mode == DS_LOAD_UPSAMPLE2 ? H / 2 : H * f2;
  a.Wi = load_mode == DS_LOAD_UPSAMPLE2 ? W / 2 : W * f2;
  a.tiles_y = (H + TY - 1) / TY; a.tiles_x = (W + TX - 1) / TX;
  hipStream_t s = ds::as_stream(stream);
  if (load_mode == DS_LOAD_PLAIN) return launch3d<DS_LOAD_PLAIN>(a, circ, s);
  if (load_mode == DS_LOAD_MAXPOOL2) return launch3d<DS_LOAD_MAXPOOL2>(a, circ, s);
  return launch3d<DS_LOAD_UPSAMPLE2>(a, circ, s);
}

// ---------------------------------------------------------------------------------------------------------------
// Volumes on the matrix cores: a 3x3x3 convolution is three 3x3 convolutions over (H, W), one per depth tap,
//     out[b, :, z] = sum_kz conv2d(in[b, :, z + kz - 1], w[:, :, kz]),
// so the 2-D fp16x3 kernels (ds_conv3h.hip / ds_convup.hip) can do the work if a depth slice looks like a 2-D sample.
// The two kernels below move a volume between the network's layout [B, C, D, H, W] and a slice-major, depth-padded
// one  S[b][zp][c][y][x], zp = z + 1 in [0, D + 2):  the flat slice index b*(D+2) + zp then IS the 2-D batch index, a
// depth tap is a pointer offset of one slice, and the pad slices (zero, or the wrapped neighbours for periodic
// padding) make the taps that leave the volume read the right thing without any per-sample logic in the hot kernel.
// Outputs computed AT pad slices are garbage by construction and never read back.
// ---------------------------------------------------------------------------------------------------------------
namespace {

// depth_mode 0: copy; 1: max over the depth pairs (2z, 2z+1) -- the depth half of MaxPool3d(2), the (H, W) half is the
// 2-D kernel's MAXPOOL2 loader; 2: nearest x2 in depth (z >> 1) -- likewise for the upsampling
__global__ __launch_bounds__(256) void k_to_slices(float* S, const float* __restrict__ x, int C, int D, int Din, size_t HW,
                                                  int depth_mode, int circular, int pad) {
  const int DP = D + 2 * pad;                       // pad slices on each side: k/2 of a k-tap depth axis
  const int zp = blockIdx.y % DP;
  const int b = blockIdx.y / DP;
  const int c = blockIdx.z;
  float* dst = S + (((size_t)b * DP + zp) * C + c) * HW;
  int z = zp - pad;
  bool zero = false;
  if (z < 0) { zero = !circular; z = circular ? z + D : 0; }      // pad <= D is checked on the host
  if (z >= D) { zero = !circular; z = circular ? z - D : 0; }
  const float* src = x + ((size_t)b * C + c) * Din * HW;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < HW; i += (size_t)gridDim.x * 256) {
    float v = 0.f;
    if (!zero) {
      if (depth_mode == 0) v = src[(size_t)z * HW + i];
      else if (depth_mode == 1) v = fmaxf(src[(size_t)(2 * z) * HW + i], src[(size_t)(2 * z + 1) * HW + i]);
      else v = src[(size_t)(z >> 1) * HW + i];
    }
    dst[i] = v;
  }
}

__global__ __launch_bounds__(256) void k_from_slices(float* y, const float* __restrict__ S, const float* __restrict__ r1,
                                                    const float* __restrict__ r2, int C, int D, size_t HW, int pad) {
  const int z = blockIdx.y % D;
  const int b = blockIdx.y / D;
  const int c = blockIdx.z;
  const float* src = S + (((size_t)b * (D + 2 * pad) + z + pad) * C + c) * HW;
  const size_t o = (((size_t)b * C + c) * D + z) * HW;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < HW; i += (size_t)gridDim.x * 256) {
    float v = src[i];
    if (r1) v = v + r1[o + i];
    if (r2) v = v + r2[o + i];
    y[o + i] = v;
  }
}

// ---- the same two copies with the block's normalisation folded in (round 2) ----
// k_to_slices_act: the copy applies SiLU((x - M) A + C) from a per-(sample, channel) table (ds_inorm_table's rows) -- the
// standalone norm pass over the volume disappears; pad slices stay exactly zero, or hold the activated wrapped neighbours
// (circular: periodic padding along the depth).  Same expf / IEEE division as the standalone norm kernels (ds_norm.hip).
__global__ __launch_bounds__(256) void k_to_slices_act(float* S, const float* __restrict__ x, const float* __restrict__ table,
                                                      int C, int Cpad, int D, size_t HW, int circular) {
  const int zp = blockIdx.y % (D + 2);
  const int b = blockIdx.y / (D + 2);
  const int c = blockIdx.z;
  float* dst = S + (((size_t)b * (D + 2) + zp) * C + c) * HW;
  int z = zp - 1;
  bool zero = false;
  if (z < 0) { zero = !circular; z = D - 1; }
  if (z >= D) { zero = !circular; z = 0; }
  const float4 t = reinterpret_cast<const float4*>(table)[(size_t)b * Cpad + c];
  const float* src = x + (((size_t)b * C + c) * D + (zero ? 0 : z)) * HW;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < HW; i += (size_t)gridDim.x * 256) {
    float v = 0.f;
    if (!zero) {
      const float u = (src[i] - t.x) * t.y + t.z;
      v = u / (1.0f + expf(-u));
    }
    dst[i] = v;
  }
}

// k_from_slices_stats: the copy back also leaves the shifted partial sums (K, S, Q, n) of what it stores, one entry per
// workgroup, in ds_conv_epilogue.h's tile-statistics format: stats[(b*C + c)*(D*gx) + z*gx + blockIdx.x] -- the next
// block's first norm needs no pass of its own over the volume.
__global__ __launch_bounds__(256) void k_from_slices_stats(float* y, const float* __restrict__ S, const float* __restrict__ r1,
                                                          const float* __restrict__ r2, float* __restrict__ stats, int C,
                                                          int D, size_t HW, int pad) {
  __shared__ float sk;
  __shared__ float red[3][4];
  const int z = blockIdx.y % D;
  const int b = blockIdx.y / D;
  const int c = blockIdx.z;
  const float* src = S + (((size_t)b * (D + 2 * pad) + z + pad) * C + c) * HW;
  const size_t o = (((size_t)b * C + c) * D + z) * HW;
  const size_t first = (size_t)blockIdx.x * 256;
  if (threadIdx.x == 0) {
    float v = first < HW ? src[first] : 0.f;
    if (first < HW && r1) v = v + r1[o + first];
    if (first < HW && r2) v = v + r2[o + first];
    sk = v;
  }
  __syncthreads();
  const float K = sk;
  float s1 = 0.f, s2 = 0.f, n = 0.f;
  for (size_t i = first + threadIdx.x; i < HW; i += (size_t)gridDim.x * 256) {
    float v = src[i];
    if (r1) v = v + r1[o + i];
    if (r2) v = v + r2[o + i];
    y[o + i] = v;
    const float d = v - K;
    s1 += d; s2 += d * d; n += 1.f;
  }
  for (int k = 32; k > 0; k >>= 1) { s1 += __shfl_xor(s1, k, 64); s2 += __shfl_xor(s2, k, 64); n += __shfl_xor(n, k, 64); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s1; red[1][threadIdx.x >> 6] = s2; red[2][threadIdx.x >> 6] = n; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float S1 = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    const float S2 = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    const float Nn = (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]);
    reinterpret_cast<float4*>(stats)[((size_t)b * C + c) * ((size_t)D * gridDim.x) + (size_t)z * gridDim.x + blockIdx.x] =
        make_float4(K, S1, S2, Nn);
  }
}

// k_slice_tables: the norm table of a volume that STAYS slice-major between the two convolutions of a block.  Input: the
// tile statistics the last depth-tap launch left, [2-D sample j = slice - 1][c][tile]; a sample's D real slices are
// combined (fp64, fixed order) into one (M, A, C) row, written for each of them; pad slices get zero rows -- whatever the
// tap launches computed there becomes SiLU(0) = 0 in the consumer's loader, i.e. the zero padding of the depth axis -- or,
// circular, the sample's row as well: the pad slices then hold wrapped copies of real slices (ds_wrap_pad_slices).
// 16 lanes per (b, c).
__global__ __launch_bounds__(256) void k_slice_tables(float* table, const float* __restrict__ ts, const float* __restrict__ w,
                                                     const float* __restrict__ bias, int B, int C, int Cpad, int D, int ntiles,
                                                     double inv_n, float eps, int kind, int circular) {
  const int row = blockIdx.x * 16 + (threadIdx.x >> 4);
  const int l = threadIdx.x & 15;
  const int b = row / Cpad, c = row - b * Cpad;
  if (b >= B) return;
  const bool real = c < C;
  double s = 0.0, q = 0.0;
  if (real) {
    for (int z = 0; z < D; ++z) {
      const float4* p = reinterpret_cast<const float4*>(ts) + (((size_t)b * (D + 2) + z) * C + c) * ntiles;
      for (int t = l; t < ntiles; t += 16) {
        const float4 v = p[t];
        const double K = v.x, S = v.y, Q = v.z, n = v.w;
        s += n * K + S;
        q += Q + 2.0 * K * S + n * K * K;
      }
    }
  }
  for (int o = 8; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
  float4 o4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (real) {
    float M = 0.f, rs = 1.0f;
    if (kind == 0) {
      const double mean = s * inv_n;
      double var = q * inv_n - mean * mean;
      if (var < 0.0) var = 0.0;
      M = (float)mean;
      rs = 1.0f / sqrtf((float)var + eps);
    } else if (kind == 1) {
      rs = 1.0f / sqrtf((float)(q * inv_n) + eps);
    }
    o4 = make_float4(M, rs * ((w && kind != 2) ? w[c] : 1.f), (bias && kind != 2) ? bias[c] : 0.f, 0.f);
  }
  for (int zp = l; zp < D + 2; zp += 16)
    reinterpret_cast<float4*>(table)[((size_t)b * (D + 2) + zp) * Cpad + c] =
        ((zp >= 1 && zp <= D) || circular) ? o4 : make_float4(0.f, 0.f, 0.f, 0.f);
}

// Periodic depth padding of a slice-major volume that was PRODUCED slice-major (the intermediate of a folded block): pad slice 0
// of every sample takes its last real slice, pad slice D + 1 its first.  grid (blocks, 2 B, C).
__global__ __launch_bounds__(256) void k_wrap_pad_slices(float* S, int C, int D, size_t HW) {
  const int b = blockIdx.y >> 1, hi = blockIdx.y & 1, c = blockIdx.z;
  const float* src = S + (((size_t)b * (D + 2) + (hi ? 1 : D)) * C + c) * HW;
  float* dst = S + (((size_t)b * (D + 2) + (hi ? D + 1 : 0)) * C + c) * HW;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < HW; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

// AvgPool3d(2) / nearest x2 upsampling of volumes, the resampling of ADM blocks on volumes (adm.py:352-384: AvgPool3d,
// Upsample(mode='nearest')).  One thread per output voxel; the pooling adds its 8 inputs in (z, y, x) order and divides
// by 8, as torch's CPU kernel does.
__global__ __launch_bounds__(256) void k_avgpool3d(float* __restrict__ out, const float* __restrict__ x, int Do, int Ho, int Wo,
                                                  size_t total) {
  const int Di = 2 * Do, Hi = 2 * Ho, Wi = 2 * Wo;
  (void)Di;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int xo = (int)(i % Wo);
    size_t t = i / Wo;
    const int yo = (int)(t % Ho); t /= Ho;
    const int zo = (int)(t % Do);
    const size_t plane = t / Do;
    const float* p = x + ((plane * (2 * Do) + 2 * zo) * (size_t)Hi + 2 * yo) * Wi + 2 * xo;
    float s = 0.f;
#pragma unroll
    for (int dz = 0; dz < 2; ++dz)
#pragma unroll
      for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) s = s + p[((size_t)dz * Hi + dy) * Wi + dx];
    out[i] = s / 8.0f;
  }
}

__global__ __launch_bounds__(256) void k_upsample3d(float* __restrict__ out, const float* __restrict__ x, int Di, int Hi, int Wi,
                                                   size_t total) {
  const int Ho = 2 * Hi, Wo = 2 * Wi, Do = 2 * Di;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int xo = (int)(i % Wo);
    size_t t = i / Wo;
    const int yo = (int)(t % Ho); t /= Ho;
    const int zo = (int)(t % Do);
    const size_t plane = t / Do;
    out[i] = x[((plane * Di + (zo >> 1)) * (size_t)Hi + (yo >> 1)) * Wi + (xo >> 1)];
  }
}

}  // namespace

extern "C" int ds_avgpool3d(float* out, const float* x, int planes, int Do, int Ho, int Wo, void* stream) {
  DS_REQUIRE(out && x, DS_ERR_NULL, "ds_avgpool3d: NULL pointer");
  DS_REQUIRE(planes >= 0 && Do > 0 && Ho > 0 && Wo > 0, DS_ERR_SHAPE, "ds_avgpool3d: bad shape");
  const size_t total = (size_t)planes * Do * Ho * Wo;
  if (total == 0) return DS_OK;
  size_t g = (total + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(k_avgpool3d, dim3((unsigned)g), dim3(256), 0, ds::as_stream(stream), out, x, Do, Ho, Wo, total);
  DS_CHECK_LAUNCH("ds_avgpool3d");
  return DS_OK;
}

extern "C" int ds_upsample3d(float* out, const float* x, int planes, int Di, int Hi, int Wi, void* stream) {
  DS_REQUIRE(out && x, DS_ERR_NULL, "ds_upsample3d: NULL pointer");
  DS_REQUIRE(planes >= 0 && Di > 0 && Hi > 0 && Wi > 0, DS_ERR_SHAPE, "ds_upsample3d: bad shape");
  const size_t total = (size_t)planes * Di * Hi * Wi * 8;
  if (total == 0) return DS_OK;
  size_t g = (total + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(k_upsample3d, dim3((unsigned)g), dim3(256), 0, ds::as_stream(stream), out, x, Di, Hi, Wi, total);
  DS_CHECK_LAUNCH("ds_upsample3d");
  return DS_OK;
}

extern "C" int ds_volume_to_slices(float* slices, const float* x, int B, int C, int D, size_t HW, int depth_mode,
                                   int circular, int pad, void* stream) {
  DS_REQUIRE(slices && x, DS_ERR_NULL, "ds_volume_to_slices: NULL pointer");
  DS_REQUIRE(B >= 0 && C > 0 && D > 0 && HW > 0 && depth_mode >= 0 && depth_mode <= 2, DS_ERR_SHAPE,
             "ds_volume_to_slices: bad arguments B=%d C=%d D=%d mode=%d", B, C, D, depth_mode);
  DS_REQUIRE(depth_mode != 2 || D % 2 == 0, DS_ERR_SHAPE, "ds_volume_to_slices: upsampling needs an even output depth");
  DS_REQUIRE(pad >= 0 && pad <= 3 && (!circular || pad <= D), DS_ERR_SHAPE, "ds_volume_to_slices: pad=%d (0..3, at most D when circular)", pad);
  DS_REQUIRE((long long)B * (D + 2 * pad) < 65536 && C < 65536, DS_ERR_SHAPE, "ds_volume_to_slices: B*(D+2*pad) and C must stay below 65536");
  if (B == 0) return DS_OK;
  const int Din = depth_mode == 1 ? 2 * D : (depth_mode == 2 ? D / 2 : D);
  const size_t gx = (HW + 1023) / 1024;
  hipLaunchKernelGGL(k_to_slices, dim3((unsigned)(gx > 64 ? 64 : gx), (unsigned)(B * (D + 2 * pad)), (unsigned)C), dim3(256), 0,
                     ds::as_stream(stream), slices, x, C, D, Din, HW, depth_mode, circular, pad);
  DS_CHECK_LAUNCH("ds_volume_to_slices");
  return DS_OK;
}

extern "C" int ds_slices_to_volume(float* y, const float* slices, const float* res1, const float* res2, int B, int C, int D,
                                   size_t HW, int pad, void* stream) {
  DS_REQUIRE(y && slices, DS_ERR_NULL, "ds_slices_to_volume: NULL pointer");
  DS_REQUIRE(B >= 0 && C > 0 && D > 0 && HW > 0 && pad >= 0 && pad <= 3, DS_ERR_SHAPE, "ds_slices_to_volume: bad shape");
  DS_REQUIRE((long long)B * D < 65536 && C < 65536, DS_ERR_SHAPE, "ds_slices_to_volume: B*D and C must stay below 65536");
  if (B == 0) return DS_OK;
  const size_t gx = (HW + 1023) / 1024;
  hipLaunchKernelGGL(k_from_slices, dim3((unsigned)(gx > 64 ? 64 : gx), (unsigned)(B * D), (unsigned)C), dim3(256), 0,
                     ds::as_stream(stream), y, slices, res1, res2, C, D, HW, pad);
  DS_CHECK_LAUNCH("ds_slices_to_volume");
  return DS_OK;
}

static unsigned slice_copy_blocks(size_t HW) {
  const size_t gx = (HW + 1023) / 1024;
  return (unsigned)(gx > 64 ? 64 : gx);
}

extern "C" int ds_volume_stat_tiles(int D, size_t HW) { return D <= 0 || HW == 0 ? 0 : D * (int)slice_copy_blocks(HW); }

extern "C" int ds_volume_to_slices_act(float* slices, const float* x, const float* table, int B, int C, int D, size_t HW,
                                       int circular, void* stream) {
  DS_REQUIRE(slices && x && table, DS_ERR_NULL, "ds_volume_to_slices_act: NULL pointer");
  DS_REQUIRE(B >= 0 && C > 0 && D > 0 && HW > 0, DS_ERR_SHAPE, "ds_volume_to_slices_act: bad arguments B=%d C=%d D=%d", B, C, D);
  DS_REQUIRE((long long)B * (D + 2) < 65536 && C < 65536, DS_ERR_SHAPE, "ds_volume_to_slices_act: B*(D+2) and C must stay below 65536");
  DS_REQUIRE((reinterpret_cast<uintptr_t>(table) & 15u) == 0, DS_ERR_SHAPE, "ds_volume_to_slices_act: table must be 16-byte aligned");
  if (B == 0) return DS_OK;
  hipLaunchKernelGGL(k_to_slices_act, dim3(slice_copy_blocks(HW), (unsigned)(B * (D + 2)), (unsigned)C), dim3(256), 0,
                     ds::as_stream(stream), slices, x, table, C, (C + 15) / 16 * 16, D, HW, circular);
  DS_CHECK_LAUNCH("ds_volume_to_slices_act");
  return DS_OK;
}

extern "C" int ds_wrap_pad_slices(float* slices, int B, int C, int D, size_t HW, void* stream) {
  DS_REQUIRE(slices, DS_ERR_NULL, "ds_wrap_pad_slices: NULL pointer");
  DS_REQUIRE(B >= 0 && C > 0 && D > 0 && HW > 0, DS_ERR_SHAPE, "ds_wrap_pad_slices: bad arguments B=%d C=%d D=%d", B, C, D);
  DS_REQUIRE(2ll * B < 65536 && C < 65536, DS_ERR_SHAPE, "ds_wrap_pad_slices: 2*B and C must stay below 65536");
  if (B == 0) return DS_OK;
  hipLaunchKernelGGL(k_wrap_pad_slices, dim3(slice_copy_blocks(HW), (unsigned)(2 * B), (unsigned)C), dim3(256), 0, ds::as_stream(stream),
                     slices, C, D, HW);
  DS_CHECK_LAUNCH("ds_wrap_pad_slices");
  return DS_OK;
}

extern "C" int ds_slices_to_volume_stats(float* y, const float* slices, const float* res1, const float* res2, float* stats,
                                         int B, int C, int D, size_t HW, int pad, void* stream) {
  DS_REQUIRE(y && slices && stats, DS_ERR_NULL, "ds_slices_to_volume_stats: NULL pointer");
  DS_REQUIRE(B >= 0 && C > 0 && D > 0 && HW > 0 && pad >= 0 && pad <= 3, DS_ERR_SHAPE, "ds_slices_to_volume_stats: bad shape");
  DS_REQUIRE((long long)B * D < 65536 && C < 65536, DS_ERR_SHAPE, "ds_slices_to_volume_stats: B*D and C must stay below 65536");
  DS_REQUIRE((reinterpret_cast<uintptr_t>(stats) & 15u) == 0, DS_ERR_SHAPE, "ds_slices_to_volume_stats: stats must be 16-byte aligned");
  if (B == 0) return DS_OK;
  hipLaunchKernelGGL(k_from_slices_stats, dim3(slice_copy_blocks(HW), (unsigned)(B * D), (unsigned)C), dim3(256), 0,
                     ds::as_stream(stream), y, slices, res1, res2, stats, C, D, HW, pad);
  DS_CHECK_LAUNCH("ds_slices_to_volume_stats");
  return DS_OK;
}

extern "C" int ds_slice_tables(float* table, const float* tile_stats, const float* w, const float* b, int B, int C, int D,
                               int ntiles, long long count, float eps, int kind, int circular, void* stream) {
  DS_REQUIRE(table && tile_stats, DS_ERR_NULL, "ds_slice_tables: NULL pointer");
  DS_REQUIRE(B >= 0 && C > 0 && D > 0 && ntiles > 0 && count > 0, DS_ERR_SHAPE, "ds_slice_tables: bad shape");
  DS_REQUIRE(kind >= 0 && kind <= 2, DS_ERR_UNSUPPORTED, "ds_slice_tables: kind %d (0 GroupLN, 1 GroupRMS, 2 none)", kind);
  DS_REQUIRE((reinterpret_cast<uintptr_t>(table) & 15u) == 0 && (reinterpret_cast<uintptr_t>(tile_stats) & 15u) == 0,
             DS_ERR_SHAPE, "ds_slice_tables: misaligned pointer");
  if (B == 0) return DS_OK;
  const int Cpad = (C + 15) / 16 * 16;
  const long long rows = (long long)B * Cpad;
  DS_REQUIRE(rows < (1ll << 31), DS_ERR_SHAPE, "ds_slice_tables: too many rows");
  hipLaunchKernelGGL(k_slice_tables, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, ds::as_stream(stream), table, tile_stats,
                     w, b, B, C, Cpad, D, ntiles, 1.0 / (double)count, eps, kind, circular);
  DS_CHECK_LAUNCH("ds_slice_tables");
  return DS_OK;
}
