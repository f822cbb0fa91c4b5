// fp64 VALU issue rate on one gfx950 CU: cycles per wave64 v_fma_f64 with 1 / 2 / 4 waves per SIMD and
// 1 / 8 independent dependency chains per lane (what bounds the single-workgroup PoseOptimization kernel).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CHAINS>
__global__ void k(double* out, long long* cyc, int iters) {
  double x[CHAINS];
  for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x * 1e-3 + c;
  const double a = 1.0000001, b = 1e-9;
  __syncthreads();
  const long long t0 = clock64();
  for (int i = 0; i < iters; ++i)
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) x[c] = x[c] * a + b;
  const long long t1 = clock64();
  double s = 0;
  for (int c = 0; c < CHAINS; ++c) s += x[c];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) *cyc = t1 - t0;
}
template <int OP>  // 0: v_rcp_f64, 1: v_rsq_f64, 2: full IEEE division, 3: sqrt()
__global__ void k_trans(double* out, long long* cyc, int iters) {
  double x[8];
  for (int c = 0; c < 8; ++c) x[c] = 1.5 + threadIdx.x * 1e-3 + c;
  __syncthreads();
  const long long t0 = clock64();
  for (int i = 0; i < iters; ++i)
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      if (OP == 0) x[c] = __builtin_amdgcn_rcp(x[c]) + 1.0;
      else if (OP == 1) x[c] = __builtin_amdgcn_rsq(x[c]) + 1.0;
      else if (OP == 2) x[c] = 1.0 / x[c] + 1.0;
      else x[c] = sqrt(x[c]) + 1.0;
    }
  const long long t1 = clock64();
  double s = 0;
  for (int c = 0; c < 8; ++c) s += x[c];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) *cyc = t1 - t0;
}
template <int OP> void run_trans(double* d, long long* dc, int threads, const char* name) {
  const int iters = 1024;
  hipLaunchKernelGGL(k_trans<OP>, dim3(1), dim3(threads), 0, 0, d, dc, iters);
  hipLaunchKernelGGL(k_trans<OP>, dim3(1), dim3(threads), 0, 0, d, dc, iters);
  long long c; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
  printf("threads %4d %-10s: %.1f ticks per op per wave (8 independent chains, +1 add each), %.1f ticks of CU time per wave-op\n", threads, name,
         (double)c / (iters * 8.0), (double)c / (iters * 8.0) / (threads / 64));
}
template <int CHAINS> void run(double* d, long long* dc, int threads) {
  const int iters = 4096;
  hipLaunchKernelGGL(k<CHAINS>, dim3(1), dim3(threads), 0, 0, d, dc, iters);
  hipLaunchKernelGGL(k<CHAINS>, dim3(1), dim3(threads), 0, 0, d, dc, iters);
  long long c; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
  const double per_wave_instr = (double)c / ((double)iters * CHAINS);          // cycles per FMA as seen by one wave
  const double waves_per_simd = threads / 256.0;
  printf("threads %4d chains %d: %.2f cycles per v_fma_f64 per wave  -> %.2f cycles of SIMD time per wave-instruction\n", threads, CHAINS,
         per_wave_instr, per_wave_instr / (waves_per_simd < 1 ? 1 : waves_per_simd));
}
int main() {
  double* d; long long* dc; hipMalloc(&d, 1024 * 8); hipMalloc(&dc, 8);
  for (int th : {256, 512, 1024}) { run<1>(d, dc, th); run<8>(d, dc, th); }
  for (int th : {64, 256, 1024}) { run_trans<0>(d, dc, th, "v_rcp_f64"); run_trans<1>(d, dc, th, "v_rsq_f64"); run_trans<2>(d, dc, th, "1.0/x"); run_trans<3>(d, dc, th, "sqrt(x)"); }
  return 0;
}
