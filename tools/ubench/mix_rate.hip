// How do a wave that issues f16 MFMAs (v_mfma_f32_16x16x32_f16, four independent accumulators) and waves that issue f32 VALU
// instructions share one gfx950 SIMD?  One workgroup on one CU: MW waves per SIMD run the MFMA loop, VW waves per SIMD the VALU
// loop (8 independent chains), both for a fixed number of instructions; every wave stamps its own duration (s_memtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(1024) void k(float* out, long long* cyc, int mw, int vw, int n_mfma, int n_valu) {
  const int wave = threadIdx.x >> 6;           // waves go to SIMDs round robin: wave w sits on SIMD w & 3 (order 0,2,1,3: still one per SIMD per group of four)
  const int slot = wave >> 2;                  // 0 .. waves-per-SIMD - 1
  const bool is_mfma = slot < mw;
  __syncthreads();
  const long long t0 = clock64();
  if (is_mfma) {
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(threadIdx.x * 1e-3f + j); b[j] = (_Float16)(1.0f + j * 0.01f); }
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < n_mfma / 4; ++i) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, a, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, b, c3, 0, 0, 0);
    }
    out[threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
  } else if (slot < mw + vw) {
    float x[8];
    for (int c = 0; c < 8; ++c) x[c] = threadIdx.x * 1e-3f + c;
    float a = 1.0000001f, b = 1e-9f;
    asm volatile("" : "+v"(a), "+v"(b));
    for (int i = 0; i < n_valu / 8; ++i)
#pragma unroll
      for (int c = 0; c < 8; ++c) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b));
    float s = 0;
    for (int c = 0; c < 8; ++c) s += x[c];
    out[threadIdx.x] = s;
  }
  const long long t1 = clock64();
  if ((threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
}
int main() {
  float* d; long long* dc; (void)hipMalloc(&d, 1024 * 4); (void)hipMalloc(&dc, 16 * 8);
  const int n_mfma = 4096, n_valu = 8192;
  const int cfgs[][2] = {{1, 0}, {0, 1}, {0, 2}, {0, 3}, {1, 1}, {1, 2}, {1, 3}, {2, 0}, {2, 1}, {2, 2}, {3, 0}, {4, 0}, {3, 1}};
  for (auto& c : cfgs) {
    const int mw = c[0], vw = c[1], threads = 256 * (mw + vw);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(1), dim3(threads), 0, 0, d, dc, mw, vw, n_mfma, n_valu);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(1), dim3(threads), 0, 0, d, dc, mw, vw, n_mfma * 16, n_valu * 16);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(17);
    (void)hipMemcpy(h.data(), dc, 16 * 8, hipMemcpyDeviceToHost);
    // the stamps are the long run's (16x the instruction counts); the wall time is that launch's, by events
    double tm = 0, tv = 0;
    for (int w = 0; w < 4 * mw; ++w) tm += (double)h[w] / (4 * mw);
    for (int w = 4 * mw; w < 4 * (mw + vw); ++w) tv += (double)h[w] / (4 * vw);
    printf("%d MFMA + %d VALU waves per SIMD:", mw, vw);
    if (mw) printf("  MFMA wave %.1f ticks per MFMA (%.1f per SIMD)", tm / (16.0 * n_mfma), tm / (16.0 * n_mfma) / mw);
    if (vw) printf("  VALU wave %.1f ticks per v_fma_f32", tv / (16.0 * n_valu));
    printf("  [%.0f us wall", ms * 1e3);
    if (mw) printf(", %.1f M MFMA/s per SIMD = %.0f %% of the 2.5 PFLOP/s rate", mw * 16.0 * n_mfma / (ms * 1e3), mw * 16.0 * n_mfma / (ms * 1e3) / 149.3 * 100);
    printf("]\n");
  }
  return 0;
}
