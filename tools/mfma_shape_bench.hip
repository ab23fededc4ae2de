// Which fp16 MFMA shape delivers more FLOP/s under this chip's power management?  (MI355X_MICROARCH.md, DVFS give-back
// item 7: bf16 16x16x32 ran 1.12-1.15x the FLOP/s of 32x32x16 at equal cycles per FLOP, because the chip held a higher
// clock.)  Same experiment for the f16 forms the fp16x3 convolutions use, shaped like their main loop: a 64 x 64 output
// tile per wave, every operand fragment re-read from LDS by ds_read_b128, random data, two workgroups of four waves per
// CU, sustained for about a second.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_shape tools/mfma_shape_bench.hip && /tmp/mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int KSLABS = 8;                  // K = 32 slabs held in LDS: [slab][A or B][4 x 64 lanes] 16-byte vectors
constexpr int LDS_VEC = KSLABS * 2 * 4 * 64;

template <int SHAPE>                       // 0: 32x32x16, 1: 16x16x32
__global__ __launch_bounds__(256, 2) void k_bench(float* out, const u32x4* __restrict__ src, int iters, unsigned long long* clk) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32x4* L = reinterpret_cast<u32x4*>(smem);
  for (int i = threadIdx.x; i < LDS_VEC; i += 256) L[i] = src[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  f32x16 acc32[2][2];
  f32x4 acc16[4][4];
  for (int m = 0; m < 2; ++m) for (int n = 0; n < 2; ++n) for (int q = 0; q < 16; ++q) acc32[m][n][q] = 0.f;
  for (int m = 0; m < 4; ++m) for (int n = 0; n < 4; ++n) for (int q = 0; q < 4; ++q) acc16[m][n][q] = 0.f;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < KSLABS; ++s) {
      asm volatile("" ::: "memory");       // the operand reads belong to this slab's iteration: no hoisting out of the loop
      const u32x4* A = L + (s * 2 + 0) * 256 + lane;
      const u32x4* B = L + (s * 2 + 1) * 256 + lane;
      f16x8 a[4], b[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) { a[j] = *reinterpret_cast<const f16x8*>(A + 64 * j); b[j] = *reinterpret_cast<const f16x8*>(B + 64 * j); }
      if (SHAPE == 0) {
        // two k-steps of 16: fragments (m, kstep) = a[2*ks + m], (n, kstep) = b[2*ks + n]
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
              acc32[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[2 * ks + m], b[2 * ks + n], acc32[m][n], 0, 0, 0);
      } else {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < 4; ++n)
            acc16[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[m], b[n], acc16[m][n], 0, 0, 0);
      }
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float v = 0.f;
  for (int m = 0; m < 2; ++m) for (int n = 0; n < 2; ++n) for (int q = 0; q < 16; ++q) v += acc32[m][n][q];
  for (int m = 0; m < 4; ++m) for (int n = 0; n < 4; ++n) for (int q = 0; q < 4; ++q) v += acc16[m][n][q];
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = v;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  const int blocks = 512;
  std::vector<_Float16> h((size_t)LDS_VEC * 8);
  unsigned s = 1234567u;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (_Float16)(((s >> 8) * (1.0f / 8388608.0f)) - 1.0f); }
  u32x4* src; float* out; unsigned long long* clk;
  hipMalloc(&src, h.size() * 2); hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  hipMalloc(&out, (size_t)blocks * 256 * 4); hipMalloc(&clk, blocks * 16);
  const size_t lds = (size_t)LDS_VEC * 16;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bench<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bench<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int round = 0; round < 3; ++round)
    for (int shape = 0; shape < 2; ++shape) {
      hipEventRecord(e0);
      if (shape == 0) hipLaunchKernelGGL(k_bench<0>, dim3(blocks), dim3(256), lds, 0, out, src, iters, clk);
      else hipLaunchKernelGGL(k_bench<1>, dim3(blocks), dim3(256), lds, 0, out, src, iters, clk);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      std::vector<unsigned long long> hc(blocks * 2);
      hipMemcpy(hc.data(), clk, blocks * 16, hipMemcpyDeviceToHost);
      std::vector<double> mhz;
      for (int b = 0; b < blocks; ++b) mhz.push_back((double)hc[2 * b] / (double)hc[2 * b + 1] * 100.0);
      std::sort(mhz.begin(), mhz.end());
      const double flop = (double)blocks * 4 * iters * KSLABS * 2.0 * 64 * 64 * 32;
      printf("round %d %s: %.1f ms, %.0f TFLOP/s, in-kernel clock median %.0f MHz\n", round, shape == 0 ? "32x32x16" : "16x16x32", ms,
             flop / (ms * 1e-3) / 1e12, mhz[blocks / 2]);
    }
  return 0;
}
