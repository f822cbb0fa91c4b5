// split_mfma.hip -- accuracy and rate of an f32 GEMM tile computed on the bf16 matrix pipe with each f32 operand split
// exactly into three bf16 terms (x = h + m + l) and the six largest cross products accumulated in f32, against the
// f32 MFMA (v_mfma_f32_32x32x2_f32) and an f64 host reference.   hipcc -O3 --offload-arch=gfx950 split_mfma.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __host__ inline void split3(float x, uint16_t& h, uint16_t& m, uint16_t& l) {
  uint32_t u; memcpy(&u, &x, 4);
  const uint32_t hu = u & 0xffff0000u; float hf; memcpy(&hf, &hu, 4);
  const float r1 = x - hf; uint32_t r1u; memcpy(&r1u, &r1, 4);
  const uint32_t mu = r1u & 0xffff0000u; float mf; memcpy(&mf, &mu, 4);
  const float r2 = r1 - mf; uint32_t r2u; memcpy(&r2u, &r2, 4);
  h = hu >> 16; m = mu >> 16; l = r2u >> 16;
}

// one wave: C[32x32] = A[32xK] * B[Kx32]; A row-major [32][K], B stored [32 cols][K]
__global__ void k_f32(const float* A, const float* B, float* C, int K) {
  const int lane = threadIdx.x, li = lane & 31, h = lane >> 5;
  f32x16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[li * K + k + h], B[li * K + k + h], acc, 0, 0, 0);
  for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + li] = acc[r];
}
__global__ void k_split(const float* A, const float* B, float* C, int K, int nprod) {
  const int lane = threadIdx.x, li = lane & 31, h = lane >> 5;
  f32x16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int k = 0; k < K; k += 16) {
    union { uint16_t s[8]; bf16x8 v; } a[3], b[3];
    for (int j = 0; j < 8; ++j) {
      split3(A[li * K + k + 8 * h + j], a[0].s[j], a[1].s[j], a[2].s[j]);
      split3(B[li * K + k + 8 * h + j], b[0].s[j], b[1].s[j], b[2].s[j]);
    }
    const int pa[9] = {2, 0, 1, 1, 0, 0, 2, 1, 2}, pb[9] = {0, 2, 1, 0, 1, 0, 1, 2, 2};
    // order: (l,h) (h,l) (m,m) (m,h) (h,m) (h,h) then the three dropped ones for nprod = 9
    const int order6[6] = {0, 1, 2, 3, 4, 5};
    if (nprod == 9) for (int q = 6; q < 9; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[pa[q]].v, b[pb[q]].v, acc, 0, 0, 0);
    for (int q = (nprod == 3 ? 3 : 0); q < 6; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[pa[order6[q]]].v, b[pb[order6[q]]].v, acc, 0, 0, 0);
  }
  for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + li] = acc[r];
}

int main() {
  const int Ks[3] = {288, 1152, 8192};
  for (int K : Ks) {
    std::vector<float> A(32 * K), B(32 * K), C(1024);
    srand(1);
    for (auto& v : A) { float x = (float)rand() / RAND_MAX * 2.f - 0.6f; v = x > 0 ? x : 0.f; }      // post-ReLU-like
    for (auto& v : B) v = ((float)rand() / RAND_MAX * 2.f - 1.f) * 0.05f;
    std::vector<double> ref(1024);
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { double s = 0; for (int k = 0; k < K; ++k) s += (double)A[i * K + k] * B[j * K + k]; ref[i * 32 + j] = s; }
    float *dA, *dB, *dC; hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 4096);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    auto report = [&](const char* name) {
      hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
      double mx = 0, rms = 0, bias = 0, scale = 0;
      for (int i = 0; i < 1024; ++i) { const double e = C[i] - ref[i]; mx = fmax(mx, fabs(e)); rms += e * e; bias += e; scale += ref[i] * ref[i]; }
      printf("K=%5d %-14s max|err| %.3e  rms %.3e  mean err %+.3e  (rms of result %.3e)\n", K, name, mx, sqrt(rms / 1024), bias / 1024, sqrt(scale / 1024));
    };
    hipLaunchKernelGGL(k_f32, dim3(1), dim3(64), 0, 0, dA, dB, dC, K); hipDeviceSynchronize(); report("f32 mfma");
    hipLaunchKernelGGL(k_split, dim3(1), dim3(64), 0, 0, dA, dB, dC, K, 6); hipDeviceSynchronize(); report("bf16x3 6 prod");
    hipLaunchKernelGGL(k_split, dim3(1), dim3(64), 0, 0, dA, dB, dC, K, 9); hipDeviceSynchronize(); report("bf16x3 9 prod");
    hipLaunchKernelGGL(k_split, dim3(1), dim3(64), 0, 0, dA, dB, dC, K, 3); hipDeviceSynchronize(); report("bf16x3 3 prod");
    hipFree(dA); hipFree(dB); hipFree(dC);
  }
  return 0;
}
