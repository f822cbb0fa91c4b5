// What does an exchange of 29 doubles per workgroup among W workgroups of ONE XCD cost per round?  (The question behind a
// multi-workgroup PoseOptimization: each pass would end in such an exchange.)  Workgroups b, b + 8, ... of a grid share an XCD
// (MI355X_MICROARCH.md); the grid has 8 W workgroups of which those with blockIdx.x % 8 == 0 take part.  Per round: every participant
// stores its 29 partials (sc1 stores), one lane adds 1 to an agent-scope counter, everybody polls the counter (sc1 loads, bounded),
// then reads all W x 29 partials (sc1 loads) and sums them in index order.  Reports shader cycles per round as seen by workgroup 0.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(64) void k(double* part, unsigned* counter, long long* out, int W, int rounds, unsigned* xcc_out) {
  if (blockIdx.x % 8 != 0) return;
  const int me = blockIdx.x / 8, lane = threadIdx.x;
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  if (lane == 0) xcc_out[me] = xcc & 0xf;
  double acc = 1.0 + me;
  long long t0 = 0;
  int failed = 0;
  for (int r = 0; r < rounds && !failed; ++r) {
    if (r == 8) t0 = clock64();
    double* mine = part + ((size_t)(r & 1) * W + me) * 32;
    if (lane < 29) __hip_atomic_store(mine + lane, acc + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned want = (unsigned)W * (r + 1);
    int spins = 0;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) { if (++spins > 2000000) { failed = 1; break; } }
    double s = 0.0;
    if (lane < 29)
      for (int w = 0; w < W; ++w) s += __hip_atomic_load(part + ((size_t)(r & 1) * W + w) * 32 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    acc = s * 0.25;
  }
  const long long t1 = clock64();
  if (lane == 0) { out[me * 2] = failed ? -1 : (t1 - t0) / (rounds - 8); out[me * 2 + 1] = (long long)(acc * 1000); }
}
int main() {
  double* part; unsigned* counter; long long* out; unsigned* xcc;
  (void)hipMalloc(&part, 2 * 8 * 32 * 8); (void)hipMalloc(&counter, 4); (void)hipMalloc(&out, 16 * 8); (void)hipMalloc(&xcc, 8 * 4);
  for (int W : {1, 2, 4, 8}) {
    (void)hipMemset(counter, 0, 4); (void)hipMemset(part, 0, 2 * 8 * 32 * 8);
    const int rounds = 2008;
    hipLaunchKernelGGL(k, dim3(8 * W), dim3(64), 0, 0, part, counter, out, W, rounds, xcc);
    (void)hipDeviceSynchronize();
    std::vector<long long> h(16); std::vector<unsigned> hx(8);
    (void)hipMemcpy(h.data(), out, 16 * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(hx.data(), xcc, 32, hipMemcpyDeviceToHost);
    printf("W = %d workgroups (XCC ids", W);
    for (int w = 0; w < W; ++w) printf(" %u", hx[w]);
    printf("): %lld shader cycles per exchange round (workgroup 0)%s\n", h[0], h[0] < 0 ? "  [a poll ran out: not co-resident?]" : "");
  }
  return 0;
}
