// cu_census.hip -- where workgroups land: (XCC, SE, SH, CU) of every workgroup of a big grid, and of 64 successive one-workgroup
// launches on an idle chip (which XCD does a single-workgroup kernel get?).  Input to the design of a software CU reservation.
// Build: hipcc --offload-arch=gfx950 -O2 -o cu_census cu_census.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ void k_where(unsigned* out) {
  if (threadIdx.x == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc;
  }
  __builtin_amdgcn_s_sleep(100);
}
int main() {
  const int G = 4096;
  unsigned* d; hipMalloc(&d, G * 8);
  std::vector<unsigned> h(2 * G);
  hipLaunchKernelGGL(k_where, dim3(G), dim3(256), 0, 0, d);
  hipMemcpy(h.data(), d, G * 8, hipMemcpyDeviceToHost);
  std::map<unsigned, int> cnt;
  for (int b = 0; b < G; ++b) {
    const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
    const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
    cnt[(xcc << 12) | (se << 8) | (sh << 4) | cu]++;
  }
  printf("distinct (xcc,se,sh,cu): %zu\n", cnt.size());
  for (auto& kv : cnt) printf("xcc %u se %u sh %u cu %2u : %d\n", kv.first >> 12, (kv.first >> 8) & 0xf, (kv.first >> 4) & 0xf, kv.first & 0xf, kv.second);
  printf("first 32 blocks -> xcc:");
  for (int b = 0; b < 32; ++b) printf(" %u", h[2 * b + 1] & 0xf);
  printf("\nsingle-workgroup launches -> (xcc,se,cu):");
  for (int i = 0; i < 48; ++i) {
    hipLaunchKernelGGL(k_where, dim3(1), dim3(256), 0, 0, d);
    hipMemcpy(h.data(), d, 8, hipMemcpyDeviceToHost);
    printf(" (%u,%u,%u)", h[1] & 0xf, (h[0] >> 13) & 7, (h[0] >> 8) & 0xf);
  }
  printf("\n");
  return 0;
}
