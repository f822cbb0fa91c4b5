// north_star names "the ORBmatcher all-pairs float-descriptor distance ... MFMA with LDS-staged descriptor tiles".  The library
// does not use that form (Matchers): the reference's searches are windowed (~0.3 % of all pairs) and its
// DescriptorDistance is a sequential f32 sum that a GEMM cannot reproduce bit for bit.  This prototype measures what the
// all-pairs form would give: D[i][j] = |a_i|^2 + |b_j|^2 - 2 a_i . b_j with the dot products on v_mfma_f32_32x32x2_f32 (exact f32
// products, f32 accumulation), 64 x 64 output tiles, both descriptor tiles staged in LDS -- against the exact-order VALU kernel
// (the library's k_dist_matrix loop), at 2000 x 2000 x 128.  It reports time, HBM bytes per launch, and how the results differ:
// values not bit-equal, largest error, rows whose nearest neighbour changes, pairs that change side of TH_HIGH = 1.5.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -o allpairs_mfma allpairs_mfma.hip && ./allpairs_mfma
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(2); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int D = 128, TS = 64, LDP = D + 1;   // row stride 129 floats: ds_read_b32 of 32 consecutive rows hits 32 banks

// exact order (ORBmatcher.cc:1629-1650): block = 256 columns x 16 rows, as the library's k_dist_matrix
__global__ __launch_bounds__(256) void k_exact(const float* __restrict__ a, int na, const float* __restrict__ b, int nb, float* __restrict__ out) {
  __shared__ float4 sa[16][32];
  const int j = blockIdx.x * 256 + threadIdx.x, i0 = blockIdx.y * 16;
  for (int idx = threadIdx.x; idx < 16 * 32; idx += 256) sa[idx >> 5][idx & 31] = reinterpret_cast<const float4*>(a + (size_t)min(i0 + (idx >> 5), na - 1) * D)[idx & 31];
  __syncthreads();
  const float4* brow = reinterpret_cast<const float4*>(b + (size_t)min(j, nb - 1) * D);
  float acc[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int k = 0; k < 32; ++k) {
    const float4 y = brow[k];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float4 x = sa[r][k];
      float d;
      d = x.x - y.x; acc[r] = acc[r] + d * d;
      d = x.y - y.y; acc[r] = acc[r] + d * d;
      d = x.z - y.z; acc[r] = acc[r] + d * d;
      d = x.w - y.w; acc[r] = acc[r] + d * d;
    }
  }
  if (j < nb)
    for (int r = 0; r < 16; ++r)
      if (i0 + r < na) out[(size_t)(i0 + r) * nb + j] = acc[r];
}

__global__ __launch_bounds__(256) void k_norms(const float* __restrict__ x, int n, float* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int k = 0; k < D; ++k) s = s + x[(size_t)i * D + k] * x[(size_t)i * D + k];
  out[i] = s;
}

// 64 x 64 tile per workgroup, four waves = 2 x 2 sub-tiles of 32 x 32, K = 128 in 64 MFMA steps of two
__global__ __launch_bounds__(256) void k_mfma(const float* __restrict__ a, int na, const float* __restrict__ b, int nb, const float* __restrict__ an,
                                              const float* __restrict__ bn, float* __restrict__ out) {
  extern __shared__ float lds[];
  float* As = lds;
  float* Bs = lds + TS * LDP;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 31, h = lane >> 5;
  const int i0 = blockIdx.y * TS, j0 = blockIdx.x * TS;
  for (int idx = t; idx < TS * 32; idx += 256) {   // 64 rows x 32 float4 of each operand
    const int r = idx >> 5, c = idx & 31;
    const float4 va = reinterpret_cast<const float4*>(a + (size_t)min(i0 + r, na - 1) * D)[c];
    const float4 vb = reinterpret_cast<const float4*>(b + (size_t)min(j0 + r, nb - 1) * D)[c];
    float* pa = As + r * LDP + c * 4;
    float* pb = Bs + r * LDP + c * 4;
    pa[0] = va.x; pa[1] = va.y; pa[2] = va.z; pa[3] = va.w;
    pb[0] = vb.x; pb[1] = vb.y; pb[2] = vb.z; pb[3] = vb.w;
  }
  __syncthreads();
  const float* ar = As + ((wave >> 1) * 32 + li) * LDP + h;
  const float* br = Bs + ((wave & 1) * 32 + li) * LDP + h;
  f32x16 acc = {0};
#pragma unroll 16
  for (int k = 0; k < D; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[k], br[k], acc, 0, 0, 0);
  const int j = j0 + (wave & 1) * 32 + li;
  if (j < nb) {
    const float nbj = bn[j];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = i0 + (wave >> 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (i < na) out[(size_t)i * nb + j] = (an[i] + nbj) - 2.0f * acc[r];
    }
  }
}

int main() {
  const int na = 2000, nb = 2000;
  std::vector<float> ha((size_t)na * D), hb((size_t)nb * D);
  unsigned s = 4242;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xffff) / 65536.f - 0.5f; };
  for (int i = 0; i < na; ++i) {
    double n2 = 0;
    for (int k = 0; k < D; ++k) { ha[(size_t)i * D + k] = rnd(); n2 += (double)ha[(size_t)i * D + k] * ha[(size_t)i * D + k]; }
    for (int k = 0; k < D; ++k) ha[(size_t)i * D + k] = (float)(ha[(size_t)i * D + k] / std::sqrt(n2));
  }
  for (int j = 0; j < nb; ++j) {   // b_j = a_{pi(j)} + noise, renormalised: every row has a true match and a few near ones
    const int src = (int)(((long long)j * 7919) % na);
    double n2 = 0;
    for (int k = 0; k < D; ++k) { hb[(size_t)j * D + k] = ha[(size_t)src * D + k] + 0.05f * rnd(); n2 += (double)hb[(size_t)j * D + k] * hb[(size_t)j * D + k]; }
    for (int k = 0; k < D; ++k) hb[(size_t)j * D + k] = (float)(hb[(size_t)j * D + k] / std::sqrt(n2));
  }
  float *da, *db, *dan, *dbn, *d_exact, *d_mfma;
  CK(hipMalloc(&da, ha.size() * 4)); CK(hipMalloc(&db, hb.size() * 4)); CK(hipMalloc(&dan, na * 4)); CK(hipMalloc(&dbn, nb * 4));
  CK(hipMalloc(&d_exact, (size_t)na * nb * 4)); CK(hipMalloc(&d_mfma, (size_t)na * nb * 4));
  CK(hipMemcpy(da, ha.data(), ha.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
  const size_t lds = (size_t)2 * TS * LDP * 4;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_mfma), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time_it = [&](auto launch, int reps) {
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return 1e3f * ms / reps;
  };
  const float us_exact = time_it([&] { hipLaunchKernelGGL(k_exact, dim3((nb + 255) / 256, (na + 15) / 16), dim3(256), 0, 0, da, na, db, nb, d_exact); }, 50);
  const float us_norm = time_it([&] { hipLaunchKernelGGL(k_norms, dim3((na + 255) / 256), dim3(256), 0, 0, da, na, dan);
                                      hipLaunchKernelGGL(k_norms, dim3((nb + 255) / 256), dim3(256), 0, 0, db, nb, dbn); }, 50);
  const float us_mfma = time_it([&] { hipLaunchKernelGGL(k_mfma, dim3((nb + TS - 1) / TS, (na + TS - 1) / TS), dim3(256), lds, 0, da, na, db, nb, dan, dbn, d_mfma); }, 50);
  std::vector<float> he((size_t)na * nb), hm((size_t)na * nb);
  CK(hipMemcpy(he.data(), d_exact, he.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hm.data(), d_mfma, hm.size() * 4, hipMemcpyDeviceToHost));
  size_t neq = 0, flips_th = 0, nn_change = 0;
  double max_abs = 0, max_rel_small = 0;
  for (int i = 0; i < na; ++i) {
    int be = 0, bm = 0;
    for (int j = 0; j < nb; ++j) {
      const float e = he[(size_t)i * nb + j], m = hm[(size_t)i * nb + j];
      neq += memcmp(&e, &m, 4) != 0;
      const double d = std::fabs((double)e - m);
      if (d > max_abs) max_abs = d;
      if (e < 0.1f && e > 0 && d / e > max_rel_small) max_rel_small = d / e;
      flips_th += (e <= 1.5f) != (m <= 1.5f);
      if (e < he[(size_t)i * nb + be]) be = j;
      if (m < hm[(size_t)i * nb + bm]) bm = j;
    }
    nn_change += be != bm;
  }
  const double flop = 2.0 * na * nb * D, bytes = ((double)na + nb) * D * 4 + (double)na * nb * 4;
  printf("all-pairs squared L2, %d x %d x %d (f32)\n", na, nb, D);
  printf("  exact-order VALU kernel (reference summation order) : %7.1f us\n", us_exact);
  printf("  MFMA kernel (v_mfma_f32_32x32x2_f32, 64x64 LDS tiles) : %7.1f us  (+ %.1f us for the two norm kernels)  = %.1f TFLOP/s, %.2f TB/s of %.1f MB algorithmic bytes\n",
         us_mfma, us_norm, flop / us_mfma * 1e-6, bytes / us_mfma * 1e-6, bytes * 1e-6);
  printf("  values not bit-equal to the exact order: %zu of %zu (%.1f %%); largest |difference| %.3g; largest relative difference among distances < 0.1: %.3g\n",
         neq, he.size(), 100.0 * neq / he.size(), max_abs, max_rel_small);
  printf("  rows whose nearest neighbour changes: %zu of %d; pairs that change side of TH_HIGH = 1.5: %zu\n", nn_change, na, flips_th);
  return 0;
}
