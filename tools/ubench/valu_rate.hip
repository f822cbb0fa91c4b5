// f32 VALU issue rate on one gfx950 CU: cycles of SIMD time per wave64 instruction for v_fma_f32, v_pk_fma_f32 (plain form, no
// operand selects) and v_fma_mixlo_f16, 8 independent chains per lane, 1 / 2 / 3 / 4 waves per SIMD.  Question behind it: does the
// packed FMA issue in 4 cycles (two FMAs per lane per pass) or in 8 (no gain over two scalar FMAs)?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int OP>
__global__ void k(float* out, long long* cyc, int iters) {
  f32x2 x[8];
  for (int c = 0; c < 8; ++c) x[c] = f32x2{threadIdx.x * 1e-3f + c, 1.f + c};
  f32x2 a = {1.0000001f, 0.9999999f}, b = {1e-9f, 2e-9f};
  asm volatile("" : "+v"(a), "+v"(b));
  __syncthreads();
  const long long t0 = clock64();
  for (int i = 0; i < iters; ++i)
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[c][0]) : "v"(a[0]), "v"(b[0]));
      else if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b));
      else if (OP == 2) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "+v"(x[c][0]) : "v"(a[0]), "v"(b[0]));
      else asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[c][0]) : "v"(b[0]));
    }
  const long long t1 = clock64();
  float s = 0;
  for (int c = 0; c < 8; ++c) s += x[c][0] + x[c][1];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) *cyc = t1 - t0;
}
template <int OP> void run(float* d, long long* dc, int threads, const char* name) {
  const int iters = 4096;
  hipLaunchKernelGGL(k<OP>, dim3(1), dim3(threads), 0, 0, d, dc, iters);
  hipLaunchKernelGGL(k<OP>, dim3(1), dim3(threads), 0, 0, d, dc, iters);
  long long c; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
  const double per = (double)c / (iters * 8.0), wps = threads / 256.0;
  printf("threads %4d %-16s: %.2f cycles per instruction per wave -> %.2f cycles of SIMD time per wave-instruction\n", threads, name, per, per / wps);
}
int main() {
  float* d; long long* dc; hipMalloc(&d, 1024 * 4); hipMalloc(&dc, 8);
  for (int th : {256, 512, 768, 1024}) { run<0>(d, dc, th, "v_fma_f32"); run<1>(d, dc, th, "v_pk_fma_f32"); run<2>(d, dc, th, "v_fma_mixlo_f16"); run<3>(d, dc, th, "v_max_f32"); }
  return 0;
}
