// The matcher's exact-order all-pairs distance loop as its own code object (device only), compiled WITH the SLP vectoriser:
// the input of make_variants.py, which edits the resulting assembly to bisect the instruction pattern that returns wrong
// values beside a bf16 MFMA kernel (tools/ubench/mfma_pk_hazard.hip --co ...).
#include <hip/hip_runtime.h>
struct Stamp { unsigned long long t0, t1; unsigned hw, xcc; };
extern "C" __global__ __launch_bounds__(256) void victim_dist(const float* __restrict__ a, int na, const float* __restrict__ b, int nb,
                                                              float* __restrict__ out, Stamp* stamps) {
  unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  __shared__ float4 sa[16][32];
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int i0 = blockIdx.y * 16;
  for (int idx = threadIdx.x; idx < 16 * 32; idx += 256) {
    const int r = idx >> 5, c = idx & 31;
    const int ia = min(i0 + r, na - 1);
    sa[r][c] = reinterpret_cast<const float4*>(a + (size_t)ia * 128)[c];
  }
  __syncthreads();
  const float4* brow = reinterpret_cast<const float4*>(b + (size_t)min(j, nb - 1) * 128);
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
  if (threadIdx.x == 0) {
    Stamp st; st.t0 = t0; st.t1 = __builtin_amdgcn_s_memrealtime();
    unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v)); st.hw = v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); st.xcc = v & 15;
    stamps[blockIdx.y * gridDim.x + blockIdx.x] = st;
  }
}
