// Does v_mfma_f32_32x32x16_f16 honour fp16 subnormal inputs on gfx950?  (diagnostic, not product)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__global__ void k(float* out, float aval, float bval) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)0.f; b[i] = (_Float16)0.f; }
  // A[row][k]: lane (row=l&31, h=l>>5) holds k=8h+j.  Put aval at A[0][0], bval at B[0][0].
  if (threadIdx.x == 0) { a[0] = (_Float16)aval; b[0] = (_Float16)bval; }
  f32x16 c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  if (threadIdx.x == 0) out[0] = c[0];
}
int main() {
  float* d; hipMalloc(&d, 4);
  float tests[][2] = {{1.0f, 1.0f}, {3.0e-6f, 1024.0f}, {5.96e-8f, 16384.0f}, {3.0e-6f, 3.0e-6f}, {6.0e-5f, 2.0f}};
  for (auto& t : tests) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, t[0], t[1]);
    float h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    float ah = (float)(_Float16)t[0], bh = (float)(_Float16)t[1];
    printf("a=%g (fp16 %g) b=%g -> mfma %g  expected %g\n", t[0], ah, t[1], h, ah * bh);
  }
  return 0;
}
