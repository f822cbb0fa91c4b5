// Does a kernel that merely WAITS on the device (one workgroup polling a flag) slow a stream of large grids on another queue?
// Stream A: a chain of grid kernels shaped like the ASDNet layers (8000 workgroups of 256 threads, 48 KB of LDS each, 262 MB written).
// Stream B: nothing / a small spinner (64 threads) / a whole-CU spinner (512 threads, 112 KB LDS, 200+ VGPRs), released by the host.
// Prints the time per chain of stream A in the three settings.   hipcc --offload-arch=gfx950 -O3 -o spinner_beside spinner_beside.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_layer(const float* __restrict__ in, float* __restrict__ out, int n_per_wg) {
  extern __shared__ float lds[];
  const int t = threadIdx.x;
  for (int i = t; i < 12 * 1024; i += 256) lds[i] = in[(blockIdx.x * 131 + i) & 0xfffff];
  __syncthreads();
  float acc = 0.f;
  for (int k = 0; k < 256; ++k) acc = fmaf(lds[(t * 7 + k * 33) & (12 * 1024 - 1)], 1.0001f, acc);
  float* o = out + (size_t)blockIdx.x * n_per_wg;
  for (int i = t; i < n_per_wg; i += 256) o[i] = acc + (float)i;
}
template <int BIG>
__global__ __launch_bounds__(BIG ? 512 : 64) void k_spin(const unsigned* flag, unsigned value, int polls, unsigned* out) {
  extern __shared__ float lds2[];
  if (BIG) { asm volatile("v_mov_b32 v210, 0" ::: "v210"); lds2[threadIdx.x] = 0.f; }
  if (threadIdx.x == 0) {
    unsigned ok = 0;
    for (int i = 0; i < polls; ++i) {
      if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == value) { ok = 1; break; }
      __builtin_amdgcn_s_sleep(32);
    }
    *out = ok;
  }
}
int main() {
  const int n_wg = 8000, n_per_wg = 8192;   // 8000 x 32 KB = 262 MB per layer
  float *in, *out; unsigned* flag; unsigned* res;
  CK(hipMalloc(&in, 4 << 20)); CK(hipMalloc(&out, (size_t)n_wg * n_per_wg * 4)); CK(hipMalloc(&flag, 64)); CK(hipMalloc(&res, 64));
  CK(hipMemset(in, 0, 4 << 20)); CK(hipMemset(flag, 0, 64));
  int lo = 0, hi = 0; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  hipStream_t a, b; CK(hipStreamCreateWithPriority(&a, hipStreamDefault, lo)); CK(hipStreamCreateWithPriority(&b, hipStreamNonBlocking, lo));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_layer), hipFuncAttributeMaxDynamicSharedMemorySize, 48 * 1024));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_spin<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 112 * 1024));
  auto chain = [&](int reps) -> double {
    if (hipStreamSynchronize(a) != hipSuccess) return -1;
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r)
      for (int l = 0; l < 6; ++l) hipLaunchKernelGGL(k_layer, dim3(n_wg), dim3(256), 48 * 1024, a, in, out, n_per_wg);
    if (hipStreamSynchronize(a) != hipSuccess) return -1;
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / reps;
  };
  (void)chain(5);
  unsigned token = 0;
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      ++token;
      if (mode == 1) hipLaunchKernelGGL(k_spin<0>, dim3(1), dim3(64), 0, b, flag, token, 400000, res);
      if (mode == 2) hipLaunchKernelGGL(k_spin<1>, dim3(1), dim3(512), 112 * 1024, b, flag, token, 400000, res);
      const double ms = chain(40);
      CK(hipMemcpyAsync(flag, &token, 4, hipMemcpyHostToDevice, a));   // release the spinner
      CK(hipStreamSynchronize(a)); CK(hipStreamSynchronize(b));
      printf("%s: %.3f ms per chain of 6 layers\n", mode == 0 ? "no spinner      " : mode == 1 ? "small spinner   " : "whole-CU spinner", ms);
    }
  }
  return 0;
}
