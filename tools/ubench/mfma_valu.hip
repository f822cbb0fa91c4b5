// Does f32 VALU work co-issue with f32 MFMA (v_mfma_f32_32x32x2_f32) on a gfx950 SIMD?
// One 512-thread workgroup per CU: waves 0-3 (one per SIMD) run a pure MFMA chain, waves 4-7 a pure VALU chain.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE>  // 0: both, 1: MFMA waves only, 2: VALU waves only, 3: VALU = int ops, 4: VALU = f64
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  const int wave = threadIdx.x >> 6;
  if (wave < 4) {
    if (MODE == 2) return;
    f32x16 acc0 = {0}, acc1 = {0};
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    for (int i = 0; i < iters; ++i) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
    }
    out[blockIdx.x * 512 + threadIdx.x] = acc0[0] + acc1[3];
  } else {
    if (MODE == 1) return;
    if (MODE == 3) {
      int x0 = threadIdx.x, x1 = 3, x2 = 5, x3 = 7;
      for (int i = 0; i < iters * 8; ++i) { x0 = x0 * 3 + x1; x1 = x1 * 5 + x2; x2 = x2 * 7 + x3; x3 = x3 * 9 + x0; }
      out[blockIdx.x * 512 + threadIdx.x] = (float)(x0 ^ x1 ^ x2 ^ x3);
    } else if (MODE == 4) {
      double x0 = threadIdx.x, x1 = 1.5, x2 = 2.5, x3 = 3.5;
      for (int i = 0; i < iters * 4; ++i) { x0 = x0 * 1.0000001 + x1; x1 = x1 * 0.9999999 + x2; x2 = x2 * 1.0000002 + x3; x3 = x3 * 0.9999998 + x0; }
      out[blockIdx.x * 512 + threadIdx.x] = (float)(x0 + x1 + x2 + x3);
    } else {
      float x0 = threadIdx.x, x1 = 1.5f, x2 = 2.5f, x3 = 3.5f, x4 = 0.5f, x5 = 0.25f, x6 = 1.25f, x7 = 4.5f;
      for (int i = 0; i < iters * 4; ++i) {
        x0 = x0 * 1.0000001f + x1; x1 = x1 * 0.9999999f + x2; x2 = x2 * 1.0000002f + x3; x3 = x3 * 0.9999998f + x4;
        x4 = x4 * 1.0000003f + x5; x5 = x5 * 0.9999997f + x6; x6 = x6 * 1.0000004f + x7; x7 = x7 * 0.9999996f + x0;
      }
      out[blockIdx.x * 512 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    }
  }
}
template <int MODE> float run(float* d, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, d, iters);
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 5;
}
int main() {
  float* d; hipMalloc(&d, 256 * 512 * 4);
  const int iters = 20000;
  printf("MFMA only        %.3f ms\n", run<1>(d, iters));
  printf("VALU f32 only    %.3f ms\n", run<2>(d, iters));
  printf("MFMA + VALU f32  %.3f ms\n", run<0>(d, iters));
  printf("MFMA + VALU int  %.3f ms\n", run<3>(d, iters));
  printf("MFMA + VALU f64  %.3f ms\n", run<4>(d, iters));
  return 0;
}
