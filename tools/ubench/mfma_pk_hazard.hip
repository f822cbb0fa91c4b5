// Stand-alone reproducer for the round-1 co-residency failure (asdnet.hip, ASD_X3_S16): does a kernel that issues
// v_mfma_f32_16x16x32_bf16 corrupt packed-f32 vector arithmetic (v_pk_add_f32 / v_pk_mul_f32) of ANOTHER kernel running on
// the same CUs?  No library code: an aggressor kernel (bare MFMA loop, random bf16 operands in registers, optional LDS
// traffic) runs on stream A for tens of milliseconds; a victim kernel runs repeatedly on stream B meanwhile and every lane's
// packed-f32 chain is checked (a) in the kernel against the same chain issued as scalar v_sub/v_mul/v_add and (b) on the host
// against an IEEE evaluation of the same chain.  A second victim is the matcher's exact-order all-pairs distance loop
// compiled WITH the SLP vectoriser (how the library's k_dist_matrix got its packed instructions), checked on the host.
// Co-residency is measured, not assumed: both kernels stamp s_memrealtime and HW_ID / XCC_ID per workgroup and the host
// counts the victim workgroups that ran on a CU while an aggressor workgroup was running there.
//
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -o mfma_pk_hazard mfma_pk_hazard.hip && ./mfma_pk_hazard
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                                   \
  do {                                                                                          \
    hipError_t e_ = (x);                                                                        \
    if (e_ != hipSuccess) { printf("%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(2); } \
  } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

struct Stamp { unsigned long long t0, t1; unsigned hw, xcc; };

__device__ inline unsigned hw_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v)); return v; }
__device__ inline unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 15; }

__device__ inline unsigned lcg(unsigned& s) { s = s * 1664525u + 1013904223u; return s; }
// a bf16 pair in [0.5, 2) with random mantissas (random data: the chip's clock behaviour depends on operand toggling)
__device__ inline unsigned rnd_bf16_pair(unsigned& s) {
  const unsigned a = lcg(s), b = lcg(s);
  return (0x3F00u | ((a >> 9) & 0xFF)) | ((0x3F00u | ((b >> 9) & 0xFF)) << 16);
}

// MODE 0: v_mfma_f32_16x16x32_bf16, 1: v_mfma_f32_32x32x16_bf16, 2: no MFMA (f32 FMA chain), 3: v_mfma_f32_16x16x4_f32,
// 4: 16x16x32 with the A operand re-read from LDS by ds_read_b128 every step (closer to the ASDNet loop)
template <int MODE>
__global__ __launch_bounds__(256) void k_aggressor(float* out, Stamp* stamps, int iters) {
  __shared__ uint4 lds[256 * 4];
  unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  unsigned s = blockIdx.x * 256 + threadIdx.x + 12345u;
  union { unsigned u[4]; s16x8 v; uint4 q; } A, B;
  for (int i = 0; i < 4; ++i) { A.u[i] = rnd_bf16_pair(s); B.u[i] = rnd_bf16_pair(s); }
  for (int i = 0; i < 4; ++i) lds[threadIdx.x * 4 + i] = make_uint4(rnd_bf16_pair(s), rnd_bf16_pair(s), rnd_bf16_pair(s), rnd_bf16_pair(s));
  __syncthreads();
  float r = 0.f;
  if constexpr (MODE == 0 || MODE == 4) {
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < iters; ++i) {
      if constexpr (MODE == 4) { A.q = lds[((threadIdx.x + i) & 255) * 4 + (i & 3)]; }
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A.v), __builtin_bit_cast(bf16x8, B.v), c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, B.v), __builtin_bit_cast(bf16x8, A.v), c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A.v), __builtin_bit_cast(bf16x8, A.v), c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, B.v), __builtin_bit_cast(bf16x8, B.v), c3, 0, 0, 0);
      if ((i & 63) == 63) { c0 *= 1e-3f; c1 *= 1e-3f; c2 *= 1e-3f; c3 *= 1e-3f; }
    }
    r = c0[0] + c1[1] + c2[2] + c3[3];
  } else if constexpr (MODE == 1) {
    f32x16 c0 = {0}, c1 = {0};
    for (int i = 0; i < iters; ++i) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A.v), __builtin_bit_cast(bf16x8, B.v), c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, B.v), __builtin_bit_cast(bf16x8, A.v), c1, 0, 0, 0);
      if ((i & 63) == 63) { c0 *= 1e-3f; c1 *= 1e-3f; }
    }
    r = c0[0] + c1[5];
  } else if constexpr (MODE == 3) {
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0;
    const float a = __uint_as_float(0x3F800000u | (A.u[0] & 0x7FFFFF)) - 1.5f, b = __uint_as_float(0x3F800000u | (B.u[0] & 0x7FFFFF)) - 1.5f;
    for (int i = 0; i < iters * 2; ++i) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, c1, 0, 0, 0);
    }
    r = c0[0] + c1[1];
  } else {
    float x0 = __uint_as_float(0x3F800000u | (A.u[0] & 0x7FFFFF)), x1 = 1.5f, x2 = 2.5f, x3 = 3.5f;
    for (int i = 0; i < iters * 8; ++i) {
      x0 = __builtin_fmaf(x0, 0.9999999f, x1); x1 = __builtin_fmaf(x1, 0.9999998f, x2);
      x2 = __builtin_fmaf(x2, 0.9999997f, x3); x3 = __builtin_fmaf(x3, 0.9999996f, x0);
    }
    r = x0 + x1 + x2 + x3;
  }
  out[blockIdx.x * 256 + threadIdx.x] = r;
  if (threadIdx.x == 0) {
    Stamp st; st.t0 = t0; st.t1 = __builtin_amdgcn_s_memrealtime(); st.hw = hw_id(); st.xcc = xcc_id();
    stamps[blockIdx.x] = st;
  }
}

// ---- victim 1: packed chain in inline asm ------------------------------------------------------------------------
// per lane and step: d = x - y (v_pk_add_f32 with neg), p = d * d (v_pk_mul_f32), acc += p (v_pk_add_f32), x = x * g + h as
// two rounded steps (v_pk_mul_f32, v_pk_add_f32); the same chain is issued with scalar VALU instructions on a second set
// of registers.  Everything is a single IEEE operation per element, so packed and scalar forms must agree bit for bit.
constexpr int VSTEPS = 1024;
__device__ __host__ inline void victim_init(unsigned gid, float* x, float* y) {
  unsigned s = gid * 2654435761u + 99u;
  for (int e = 0; e < 2; ++e) {
    s = s * 1664525u + 1013904223u;
    unsigned ux = 0x3F800000u | (s >> 9);
    s = s * 1664525u + 1013904223u;
    unsigned uy = 0x3F800000u | (s >> 9);
    float fx, fy;
    memcpy(&fx, &ux, 4); memcpy(&fy, &uy, 4);
    x[e] = fx - 1.5f; y[e] = fy - 1.5f;
  }
}
constexpr float VG = 0.99993896484375f, VH = 1.52587890625e-05f;

__global__ __launch_bounds__(256) void k_victim_pk(float* out_pk, float* out_sc, unsigned* mismatch, Stamp* stamps) {
  unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  const unsigned gid = blockIdx.x * 256 + threadIdx.x;
  float xi[2], yi[2];
  victim_init(gid, xi, yi);
  f32x2 x = {xi[0], xi[1]}, y = {yi[0], yi[1]}, acc = {0.f, 0.f};
  f32x2 g = {VG, VG}, h = {VH, -VH};
  float sx0 = xi[0], sx1 = xi[1], sa0 = 0.f, sa1 = 0.f;
  const float sy0 = yi[0], sy1 = yi[1], sg = VG, sh0 = VH, sh1 = -VH;
  for (int i = 0; i < VSTEPS; ++i) {
    f32x2 d, p;
    asm volatile(
        "v_pk_add_f32 %0, %3, %4 neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_mul_f32 %1, %0, %0\n\t"
        "v_pk_add_f32 %2, %2, %1\n\t"
        : "=&v"(d), "=&v"(p), "+v"(acc)
        : "v"(x), "v"(y));
    asm volatile(
        "v_pk_mul_f32 %0, %0, %1\n\t"
        "v_pk_add_f32 %0, %0, %2\n\t"
        : "+v"(x)
        : "v"(g), "v"(h));
    float d0, d1, p0, p1;
    asm volatile(
        "v_sub_f32 %0, %6, %8\n\t"
        "v_sub_f32 %1, %7, %9\n\t"
        "v_mul_f32 %2, %0, %0\n\t"
        "v_mul_f32 %3, %1, %1\n\t"
        "v_add_f32 %4, %4, %2\n\t"
        "v_add_f32 %5, %5, %3\n\t"
        : "=&v"(d0), "=&v"(d1), "=&v"(p0), "=&v"(p1), "+v"(sa0), "+v"(sa1)
        : "v"(sx0), "v"(sx1), "v"(sy0), "v"(sy1));
    asm volatile(
        "v_mul_f32 %0, %0, %2\n\t"
        "v_mul_f32 %1, %1, %2\n\t"
        "v_add_f32 %0, %0, %3\n\t"
        "v_add_f32 %1, %1, %4\n\t"
        : "+v"(sx0), "+v"(sx1)
        : "v"(sg), "v"(sh0), "v"(sh1));
  }
  out_pk[gid * 2] = acc[0]; out_pk[gid * 2 + 1] = acc[1];
  out_sc[gid * 2] = sa0; out_sc[gid * 2 + 1] = sa1;
  if (__float_as_uint(acc[0]) != __float_as_uint(sa0) || __float_as_uint(acc[1]) != __float_as_uint(sa1)) atomicAdd(mismatch, 1u);
  if (threadIdx.x == 0) {
    Stamp st; st.t0 = t0; st.t1 = __builtin_amdgcn_s_memrealtime(); st.hw = hw_id(); st.xcc = xcc_id();
    stamps[blockIdx.x] = st;
  }
}

static void victim_host(unsigned gid, float* acc_out) {
  float x[2], y[2];
  victim_init(gid, x, y);
  volatile float acc[2] = {0.f, 0.f};
  const float h[2] = {VH, -VH};
  for (int i = 0; i < VSTEPS; ++i)
    for (int e = 0; e < 2; ++e) {
      volatile float d = x[e] - y[e];
      volatile float p = d * d;
      acc[e] = acc[e] + p;
      volatile float m = x[e] * VG;
      x[e] = m + h[e];
    }
  acc_out[0] = acc[0]; acc_out[1] = acc[1];
}

// ---- victim 2: the matcher's exact-order all-pairs distance loop, SLP-vectorised by the compiler --------------------
__global__ __launch_bounds__(256) void k_victim_dist(const float* __restrict__ a, int na, const float* __restrict__ b, int nb,
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
    Stamp st; st.t0 = t0; st.t1 = __builtin_amdgcn_s_memrealtime(); st.hw = hw_id(); st.xcc = xcc_id();
    stamps[blockIdx.y * gridDim.x + blockIdx.x] = st;
  }
}


// ---- victim 3: write-after-read probes ---------------------------------------------------------------------------------
// What the failing compiled loop does around every wrong value: a packed-f32 instruction reads a 64-bit register pair and
// the NEXT instruction of the same wave (a v_mov_b32) overwrites the LOW register of that pair.  Each probe repeats
//     p = (a, a);  reader: acc = acc (op) p;  [gap];  writer: p.lo or p.hi = b
// with p in v[20:21], and expects the value acc has when the reader always sees (a, a).  A reader that picks up the writer's
// value shows as a different acc.  The writer never feeds anything else, so only a broken read-before-write order can show.
constexpr int WSTEPS = 4096;
#define WAR_BODY(READER, GAP, WRITER) \
  "v_mov_b32 v20, %1\n\tv_mov_b32 v21, %1\n\ts_nop 4\n\t" READER "\n\t" GAP WRITER "\n\ts_nop 4\n\t"
#define WAR_KERNEL(NAME, READER, GAP, WRITER, A, B, ACC0)                                                              \
  __global__ __launch_bounds__(256) void NAME(float* out, Stamp* stamps) {                                              \
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();                                                           \
    f32x2 acc = {ACC0, ACC0};                                                                                           \
    float a = A, b = B, t = 0.f;                                                                                        \
    asm volatile("" : "+v"(a), "+v"(b));                                                                                \
    for (int i = 0; i < WSTEPS; i += 4)                                                                                 \
      asm volatile(WAR_BODY(READER, GAP, WRITER) WAR_BODY(READER, GAP, WRITER) WAR_BODY(READER, GAP, WRITER)            \
                       WAR_BODY(READER, GAP, WRITER)                                                                    \
                   : "+v"(acc), "+v"(a), "+v"(b), "+v"(t)::"v20", "v21");                                               \
    const unsigned gid = blockIdx.x * 256 + threadIdx.x;                                                                \
    out[gid * 2] = acc[0];                                                                                              \
    out[gid * 2 + 1] = acc[1];                                                                                          \
    if (threadIdx.x == 0) {                                                                                             \
      Stamp st; st.t0 = t0; st.t1 = __builtin_amdgcn_s_memrealtime(); st.hw = hw_id(); st.xcc = xcc_id();               \
      stamps[blockIdx.x] = st;                                                                                          \
    }                                                                                                                   \
  }
// operands: %0 acc (pair), %1 a, %2 b, %3 t (spare)
WAR_KERNEL(k_war_pkadd_lo, "v_pk_add_f32 %0, %0, v[20:21]", "", "v_mov_b32 v20, %2", 1.f, 4096.f, 0.f)
WAR_KERNEL(k_war_pkadd_hi, "v_pk_add_f32 %0, %0, v[20:21]", "", "v_mov_b32 v21, %2", 1.f, 4096.f, 0.f)
WAR_KERNEL(k_war_pkadd_lo_valu, "v_pk_add_f32 %0, %0, v[20:21]", "v_mov_b32 %3, %1\n\t", "v_mov_b32 v20, %2", 1.f, 4096.f, 0.f)
WAR_KERNEL(k_war_pkadd_lo_nop0, "v_pk_add_f32 %0, %0, v[20:21]", "s_nop 0\n\t", "v_mov_b32 v20, %2", 1.f, 4096.f, 0.f)
WAR_KERNEL(k_war_pkadd_lo_nop1, "v_pk_add_f32 %0, %0, v[20:21]", "s_nop 1\n\t", "v_mov_b32 v20, %2", 1.f, 4096.f, 0.f)
WAR_KERNEL(k_war_pkmul_lo, "v_pk_mul_f32 %0, %0, v[20:21]", "", "v_mov_b32 v20, %2", 1.f, 3.f, 1.f)
WAR_KERNEL(k_war_pkadd_lo_addw, "v_pk_add_f32 %0, %0, v[20:21]", "", "v_add_f32 v20, %2, %2", 1.f, 4096.f, 0.f)
WAR_KERNEL(k_war_pkfma_lo, "v_pk_fma_f32 %0, v[20:21], v[20:21], %0", "", "v_mov_b32 v20, %2", 1.f, 64.f, 0.f)
// 32-bit control: v_add_f32 reader, v_mov_b32 writer next
__global__ __launch_bounds__(256) void k_war_add32(float* out, Stamp* stamps) {
  unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  float acc = 0.f, a = 1.f, b = 4096.f;
  asm volatile("" : "+v"(a), "+v"(b));
#define A32_BODY "v_mov_b32 v20, %1\n\ts_nop 4\n\tv_add_f32 %0, %0, v20\n\tv_mov_b32 v20, %2\n\ts_nop 4\n\t"
  for (int i = 0; i < WSTEPS; i += 4) asm volatile(A32_BODY A32_BODY A32_BODY A32_BODY : "+v"(acc), "+v"(a), "+v"(b)::"v20");
  const unsigned gid = blockIdx.x * 256 + threadIdx.x;
  out[gid * 2] = acc;
  out[gid * 2 + 1] = 0.f;
  if (threadIdx.x == 0) {
    Stamp st; st.t0 = t0; st.t1 = __builtin_amdgcn_s_memrealtime(); st.hw = hw_id(); st.xcc = xcc_id();
    stamps[blockIdx.x] = st;
  }
}
// 64-bit operand of a double-precision instruction: p = 1.0 (lo 0, hi 0x3FF00000 -- a = 1.875f has those bits), writer puts
// 1.875f's bits into the LOW word, which would make p = 1.0000004...: acc (as a double in the pair) must stay an integer
__global__ __launch_bounds__(256) void k_war_addf64_lo(float* out, Stamp* stamps) {
  unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  double acc = 0.0;
  unsigned hi = 0x3FF00000u, lo = 0u, junk = 0x3FF00000u;
  asm volatile("" : "+v"(hi), "+v"(lo), "+v"(junk));
#define F64_BODY "v_mov_b32 v20, %2\n\tv_mov_b32 v21, %1\n\ts_nop 4\n\tv_add_f64 %0, %0, v[20:21]\n\tv_mov_b32 v20, %3\n\ts_nop 4\n\t"
  for (int i = 0; i < WSTEPS; i += 4)
    asm volatile(F64_BODY F64_BODY F64_BODY F64_BODY : "+v"(acc), "+v"(hi), "+v"(lo), "+v"(junk)::"v20", "v21");
  const unsigned gid = blockIdx.x * 256 + threadIdx.x;
  out[gid * 2] = (float)acc;                                  // expected WSTEPS
  out[gid * 2 + 1] = (acc == (double)WSTEPS) ? (float)WSTEPS : -1.f;
  if (threadIdx.x == 0) {
    Stamp st; st.t0 = t0; st.t1 = __builtin_amdgcn_s_memrealtime(); st.hw = hw_id(); st.xcc = xcc_id();
    stamps[blockIdx.x] = st;
  }
}
struct WarProbe { const char* name; void (*kern)(float*, Stamp*); float expect_lo, expect_hi; };
static const WarProbe kWarProbes[] = {
    {"v_pk_add_f32 ; v_mov lo", k_war_pkadd_lo, (float)WSTEPS, (float)WSTEPS},
    {"v_pk_add_f32 ; v_mov hi", k_war_pkadd_hi, (float)WSTEPS, (float)WSTEPS},
    {"v_pk_add_f32 ; VALU ; v_mov lo", k_war_pkadd_lo_valu, (float)WSTEPS, (float)WSTEPS},
    {"v_pk_add_f32 ; s_nop 0 ; v_mov lo", k_war_pkadd_lo_nop0, (float)WSTEPS, (float)WSTEPS},
    {"v_pk_add_f32 ; s_nop 1 ; v_mov lo", k_war_pkadd_lo_nop1, (float)WSTEPS, (float)WSTEPS},
    {"v_pk_mul_f32 ; v_mov lo", k_war_pkmul_lo, 1.f, 1.f},
    {"v_pk_add_f32 ; v_add_f32 lo", k_war_pkadd_lo_addw, (float)WSTEPS, (float)WSTEPS},
    {"v_pk_fma_f32 ; v_mov lo", k_war_pkfma_lo, (float)WSTEPS, (float)WSTEPS},
    {"v_add_f32 (32-bit) ; v_mov", k_war_add32, (float)WSTEPS, 0.f},
    {"v_add_f64 ; v_mov lo", k_war_addf64_lo, (float)WSTEPS, (float)WSTEPS},
};

// ---- victim 4: operand-half selection (op_sel / op_sel_hi) on packed-f32 instructions ------------------------------------
// What the bisection of the compiled loop (hazard_bisect/) points at: the wrong values come from the packed instructions whose
// src1 carries op_sel / op_sel_hi (one register of the pair broadcast to both halves).  p = v[20:21] = (a, b) = (1, 4096).
#define SEL_BODY(READER) "v_mov_b32 v20, %1\n\tv_mov_b32 v21, %2\n\ts_nop 4\n\t" READER "\n\ts_nop 4\n\t"
#define SEL_KERNEL(NAME, READER, ACC0)                                                                                  \
  __global__ __launch_bounds__(256) void NAME(float* out, Stamp* stamps) {                                              \
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();                                                           \
    f32x2 acc = {ACC0, ACC0};                                                                                           \
    float a = 1.f, b = 4096.f, t = 0.f;                                                                                 \
    asm volatile("" : "+v"(a), "+v"(b));                                                                                \
    for (int i = 0; i < WSTEPS; i += 4)                                                                                 \
      asm volatile(SEL_BODY(READER) SEL_BODY(READER) SEL_BODY(READER) SEL_BODY(READER)                                  \
                   : "+v"(acc), "+v"(a), "+v"(b), "+v"(t)::"v20", "v21");                                               \
    const unsigned gid = blockIdx.x * 256 + threadIdx.x;                                                                \
    out[gid * 2] = acc[0];                                                                                              \
    out[gid * 2 + 1] = acc[1];                                                                                          \
    if (threadIdx.x == 0) {                                                                                             \
      Stamp st; st.t0 = t0; st.t1 = __builtin_amdgcn_s_memrealtime(); st.hw = hw_id(); st.xcc = xcc_id();               \
      stamps[blockIdx.x] = st;                                                                                          \
    }                                                                                                                   \
  }
SEL_KERNEL(k_sel_add_plain, "v_pk_add_f32 %0, %0, v[20:21]", 0.f)
SEL_KERNEL(k_sel_add_hi_from_lo, "v_pk_add_f32 %0, %0, v[20:21] op_sel_hi:[1,0]", 0.f)
SEL_KERNEL(k_sel_add_lo_from_hi, "v_pk_add_f32 %0, %0, v[20:21] op_sel:[0,1]", 0.f)
SEL_KERNEL(k_sel_add_swap, "v_pk_add_f32 %0, %0, v[20:21] op_sel:[0,1] op_sel_hi:[1,0]", 0.f)
SEL_KERNEL(k_sel_add_hi_from_lo_neg, "v_pk_add_f32 %0, %0, v[20:21] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]", 0.f)
SEL_KERNEL(k_sel_add_src0_hi_from_lo, "v_pk_add_f32 %0, v[20:21], %0 op_sel_hi:[0,1]", 0.f)
SEL_KERNEL(k_sel_fma_hi_from_lo, "v_pk_fma_f32 %0, v[20:21], v[20:21], %0 op_sel_hi:[0,0,1]", 0.f)
#define SELW (float)WSTEPS
static const WarProbe kSelProbes[] = {
    {"v_pk_add_f32 (no op_sel)", k_sel_add_plain, SELW, SELW * 4096.f},
    {"v_pk_add_f32 src1 op_sel_hi:[1,0]", k_sel_add_hi_from_lo, SELW, SELW},
    {"v_pk_add_f32 src1 op_sel:[0,1]", k_sel_add_lo_from_hi, SELW * 4096.f, SELW * 4096.f},
    {"v_pk_add_f32 src1 swapped halves", k_sel_add_swap, SELW * 4096.f, SELW},
    {"v_pk_add_f32 op_sel_hi:[1,0] + neg", k_sel_add_hi_from_lo_neg, -SELW, -SELW},
    {"v_pk_add_f32 src0 op_sel_hi:[0,1]", k_sel_add_src0_hi_from_lo, SELW, SELW},
    {"v_pk_fma_f32 src0,1 op_sel_hi 0", k_sel_fma_hi_from_lo, SELW, SELW},
};

// ---- host ---------------------------------------------------------------------------------------------------------
static unsigned cu_key(const Stamp& s) {  // (xcc, se, sh, cu) of HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]
  return (s.xcc << 8) | ((s.hw >> 8) & 0xFF);
}
// fraction of victim workgroups that ran while an aggressor workgroup was running on the same CU
static double coresident(const std::vector<Stamp>& vic, const std::vector<Stamp>& agg) {
  size_t hit = 0;
  for (const Stamp& v : vic) {
    bool h = false;
    for (const Stamp& a : agg)
      if (cu_key(a) == cu_key(v) && a.t0 < v.t1 && v.t0 < a.t1) { h = true; break; }
    hit += h;
  }
  return vic.empty() ? 0.0 : (double)hit / (double)vic.size();
}

template <int MODE> static void launch_aggr(hipStream_t st, float* out, Stamp* stamps, int blocks, int iters) {
  hipLaunchKernelGGL(k_aggressor<MODE>, dim3(blocks), dim3(256), 0, st, out, stamps, iters);
}
static void launch_aggr_mode(int mode, hipStream_t st, float* out, Stamp* stamps, int blocks, int iters) {
  switch (mode) {
    case 0: launch_aggr<0>(st, out, stamps, blocks, iters); break;
    case 1: launch_aggr<1>(st, out, stamps, blocks, iters); break;
    case 2: launch_aggr<2>(st, out, stamps, blocks, iters); break;
    case 3: launch_aggr<3>(st, out, stamps, blocks, iters); break;
    default: launch_aggr<4>(st, out, stamps, blocks, iters); break;
  }
}

int main(int argc, char** argv) {
  int dev = 0;
  CK(hipSetDevice(dev));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, dev));
  const int ncu = prop.multiProcessorCount;
  printf("device: %s (%s), %d CUs\n", prop.name, prop.gcnArchName, ncu);
  const int agg_blocks = ncu * 2;      // two 4-wave workgroups per CU: 2 MFMA waves per SIMD, 24 wave slots left per CU
  const int vic_blocks = ncu * 8;      // victim grid: 8 workgroups per CU
  const bool co_mode = argc > 1 && !strcmp(argv[1], "--co");
  const bool sel_only = argc > 1 && !strcmp(argv[1], "--sel");
  const int rounds = (argc > 1 && !co_mode && !sel_only) ? atoi(argv[1]) : 12;
  int agg_iters = (argc > 2 && !co_mode) ? atoi(argv[2]) : 400000;  // ~25-50 ms of MFMAs

  hipStream_t sa, sb;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  float *d_aout, *d_pk, *d_sc;
  unsigned* d_mis;
  Stamp *d_as, *d_vs;
  CK(hipMalloc(&d_aout, (size_t)agg_blocks * 256 * 4));
  CK(hipMalloc(&d_pk, (size_t)vic_blocks * 256 * 8));
  CK(hipMalloc(&d_sc, (size_t)vic_blocks * 256 * 8));
  CK(hipMalloc(&d_mis, 4));
  CK(hipMalloc(&d_as, (size_t)agg_blocks * sizeof(Stamp)));
  const int na = 304, nb = 2048;  // dist victim: 8 x 19 = 152 workgroups per launch
  const int dist_blocks = (nb / 256) * ((na + 15) / 16);
  CK(hipMalloc(&d_vs, (size_t)std::max(vic_blocks, dist_blocks) * sizeof(Stamp)));

  // host references
  std::vector<float> ref_pk((size_t)vic_blocks * 256 * 2);
  for (unsigned g = 0; g < (unsigned)vic_blocks * 256; ++g) victim_host(g, &ref_pk[(size_t)g * 2]);
  std::vector<float> ha((size_t)na * 128), hb((size_t)nb * 128), ref_d((size_t)na * nb);
  {
    unsigned s = 777;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xFFFF) / 65536.f - 0.5f; };
    for (auto& v : ha) v = rnd() * 0.2f;
    for (auto& v : hb) v = rnd() * 0.2f;
    for (int i = 0; i < na; ++i)
      for (int j = 0; j < nb; ++j) {
        volatile float acc = 0.f;
        for (int k = 0; k < 128; ++k) { volatile float d = ha[(size_t)i * 128 + k] - hb[(size_t)j * 128 + k]; volatile float p = d * d; acc = acc + p; }
        ref_d[(size_t)i * nb + j] = acc;
      }
  }
  float *d_a, *d_b, *d_d;
  CK(hipMalloc(&d_a, ha.size() * 4)); CK(hipMalloc(&d_b, hb.size() * 4)); CK(hipMalloc(&d_d, ref_d.size() * 4));
  CK(hipMemcpy(d_a, ha.data(), ha.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_b, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));

  std::vector<float> got_pk(ref_pk.size()), got_sc(ref_pk.size()), got_d(ref_d.size());
  std::vector<Stamp> vs, as_(agg_blocks);

  auto run_victims = [&](const char* label, int mode) {  // mode < 0: victims alone
    size_t bad_pk_host = 0, bad_sc_host = 0, bad_dist = 0, lanes = 0, dist_vals = 0;
    unsigned bad_pk_kernel = 0;
    double co_pk = 0, co_dist = 0;
    int launches = 0;
    std::vector<std::vector<Stamp>> all_vs_pk, all_vs_d;
    std::vector<int> bad_lane_hist(64, 0);
    if (mode >= 0) launch_aggr_mode(mode, sa, d_aout, d_as, agg_blocks, agg_iters);
    for (int r = 0; r < rounds; ++r) {
      if (mode >= 0 && hipStreamQuery(sa) == hipSuccess) break;  // aggressor finished: later victims would run alone
      CK(hipMemsetAsync(d_mis, 0, 4, sb));
      hipLaunchKernelGGL(k_victim_pk, dim3(vic_blocks), dim3(256), 0, sb, d_pk, d_sc, d_mis, d_vs);
      CK(hipStreamSynchronize(sb));
      unsigned m;
      CK(hipMemcpy(&m, d_mis, 4, hipMemcpyDeviceToHost));
      bad_pk_kernel += m;
      CK(hipMemcpy(got_pk.data(), d_pk, got_pk.size() * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(got_sc.data(), d_sc, got_sc.size() * 4, hipMemcpyDeviceToHost));
      vs.resize(vic_blocks);
      CK(hipMemcpy(vs.data(), d_vs, vs.size() * sizeof(Stamp), hipMemcpyDeviceToHost));
      all_vs_pk.push_back(vs);
      for (size_t i = 0; i < got_pk.size(); ++i) {
        if (memcmp(&got_pk[i], &ref_pk[i], 4)) { ++bad_pk_host; ++bad_lane_hist[(i / 2) & 63]; }
        if (memcmp(&got_sc[i], &ref_pk[i], 4)) ++bad_sc_host;
      }
      lanes += got_pk.size();
      CK(hipMemsetAsync(d_d, 0xFF, ref_d.size() * 4, sb));
      hipLaunchKernelGGL(k_victim_dist, dim3(nb / 256, (na + 15) / 16), dim3(256), 0, sb, d_a, na, d_b, nb, d_d, d_vs);
      CK(hipStreamSynchronize(sb));
      CK(hipMemcpy(got_d.data(), d_d, got_d.size() * 4, hipMemcpyDeviceToHost));
      vs.resize(dist_blocks);
      CK(hipMemcpy(vs.data(), d_vs, vs.size() * sizeof(Stamp), hipMemcpyDeviceToHost));
      all_vs_d.push_back(vs);
      for (size_t i = 0; i < got_d.size(); ++i) bad_dist += memcmp(&got_d[i], &ref_d[i], 4) != 0;
      dist_vals += got_d.size();
      ++launches;
    }
    float agg_ms = 0;
    if (mode >= 0) {
      CK(hipStreamSynchronize(sa));
      CK(hipMemcpy(as_.data(), d_as, as_.size() * sizeof(Stamp), hipMemcpyDeviceToHost));
      unsigned long long lo = ~0ull, hi = 0;
      for (auto& s : as_) { lo = std::min(lo, s.t0); hi = std::max(hi, s.t1); }
      agg_ms = (float)((hi - lo) / 100e3);  // s_memrealtime ticks at 100 MHz
      for (auto& v : all_vs_pk) co_pk += coresident(v, as_);
      for (auto& v : all_vs_d) co_dist += coresident(v, as_);
      if (launches) { co_pk /= launches; co_dist /= launches; }
    }
    printf("%-34s victim launches %2d  aggressor %6.1f ms  co-resident WGs pk %5.1f%% dist %5.1f%% | pk-vs-scalar(in kernel) %u  pk-vs-host %zu / %zu  "
           "scalar-vs-host %zu  dist-vs-host %zu / %zu\n",
           label, launches, agg_ms, 100 * co_pk, 100 * co_dist, bad_pk_kernel, bad_pk_host, lanes, bad_sc_host, bad_dist, dist_vals);
    if (bad_pk_host) {
      printf("   wrong packed results by lane:");
      for (int l = 0; l < 64; ++l) printf(" %d", bad_lane_hist[l]);
      printf("\n");
    }
    return bad_pk_kernel + bad_pk_host + bad_sc_host + bad_dist;
  };

  // warm-up (code objects loaded, clocks up), also calibrates the aggressor length to ~40 ms
  {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch_aggr_mode(0, sa, d_aout, d_as, agg_blocks, 20000);
    CK(hipStreamSynchronize(sa));
    CK(hipEventRecord(e0, sa));
    launch_aggr_mode(0, sa, d_aout, d_as, agg_blocks, 20000);
    CK(hipEventRecord(e1, sa));
    CK(hipStreamSynchronize(sa));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("aggressor mode 0: 20000 iterations = %.2f ms (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n", ms, ms * 1e-3 * 2.4e9 / (20000.0 * 4 * 2));
    if (argc <= 2 || co_mode || sel_only) agg_iters = (int)(20000 * (60.0f / ms));
    hipLaunchKernelGGL(k_victim_pk, dim3(vic_blocks), dim3(256), 0, sb, d_pk, d_sc, d_mis, d_vs);
    CK(hipStreamSynchronize(sb));
  }

  // ---- bisect mode: the dist victim comes from code objects built by hazard_bisect/make_variants.py
  if (argc > 1 && !strcmp(argv[1], "--co")) {
    const int modes[] = {-1, 4, 1};
    const char* mode_name[] = {"alone", "16x16x32+lds", "32x32x16"};
    printf("%-34s", "variant");
    for (int mi = 0; mi < 3; ++mi) printf(" | %-14s wrong (co-res%%, launches)", mode_name[mi]);
    printf("\n");
    for (int ai = 2; ai < argc; ++ai) {
      hipModule_t mod;
      hipFunction_t fn;
      CK(hipModuleLoad(&mod, argv[ai]));
      CK(hipModuleGetFunction(&fn, mod, "victim_dist"));
      const char* base = strrchr(argv[ai], '/');
      printf("%-34s", base ? base + 1 : argv[ai]);
      for (int mi = 0; mi < 3; ++mi) {
        const int mode = modes[mi];
        size_t bad = 0;
        int launches = 0;
        double co = 0;
        std::vector<std::vector<Stamp>> all;
        if (mode >= 0) launch_aggr_mode(mode, sa, d_aout, d_as, agg_blocks, agg_iters / 2);
        for (int r = 0; r < 8; ++r) {
          if (mode >= 0 && hipStreamQuery(sa) == hipSuccess) break;
          CK(hipMemsetAsync(d_d, 0xFF, ref_d.size() * 4, sb));
          int na_ = na, nb_ = nb;
          void* args[] = {&d_a, &na_, &d_b, &nb_, &d_d, &d_vs};
          CK(hipModuleLaunchKernel(fn, nb / 256, (na + 15) / 16, 1, 256, 1, 1, 0, sb, args, nullptr));
          CK(hipStreamSynchronize(sb));
          CK(hipMemcpy(got_d.data(), d_d, got_d.size() * 4, hipMemcpyDeviceToHost));
          vs.resize(dist_blocks);
          CK(hipMemcpy(vs.data(), d_vs, vs.size() * sizeof(Stamp), hipMemcpyDeviceToHost));
          all.push_back(vs);
          for (size_t i = 0; i < got_d.size(); ++i) bad += memcmp(&got_d[i], &ref_d[i], 4) != 0;
          ++launches;
        }
        if (mode >= 0) {
          CK(hipStreamSynchronize(sa));
          CK(hipMemcpy(as_.data(), d_as, as_.size() * sizeof(Stamp), hipMemcpyDeviceToHost));
          for (auto& v : all) co += coresident(v, as_);
          if (launches) co /= launches;
        }
        printf(" | %14zu       (%5.1f%%, %d)      ", bad, 100 * co, launches);
      }
      printf("\n");
      CK(hipModuleUnload(mod));
    }
    return 0;
  }
  printf("aggressor: %d workgroups x 256 threads, %d iterations; pk victim: %d workgroups x 256 threads x %d steps; dist victim %d x %d\n",
         agg_blocks, agg_iters, vic_blocks, VSTEPS, na, nb);
  size_t total = 0;
  total += run_victims("victims alone", -1);
  total += run_victims("beside v_mfma_f32_16x16x32_bf16", 0);
  total += run_victims("beside 16x16x32_bf16 + ds_read_b128", 4);
  total += run_victims("beside v_mfma_f32_32x32x16_bf16", 1);
  total += run_victims("beside v_mfma_f32_16x16x4_f32", 3);
  total += run_victims("beside f32 FMA loop (no MFMA)", 2);
  total += run_victims("beside v_mfma_f32_16x16x32_bf16 #2", 0);

  // ---- write-after-read probes, alone and beside each aggressor
  {
    float* d_w;
    CK(hipMalloc(&d_w, (size_t)vic_blocks * 256 * 8));
    std::vector<float> w((size_t)vic_blocks * 256 * 2);
    const int modes[] = {-1, 4, 1, 0, 2};
    const char* mode_name[] = {"alone", "beside 16x16x32_bf16 + ds_read_b128", "beside 32x32x16_bf16", "beside 16x16x32_bf16", "beside f32 FMA loop"};
    printf("\ninstruction probes: wrong lanes / lanes checked (lanes 0-15 | 16-31 | 32-47 | 48-63 of the wave; lo | hi half)\n");
    for (int mi = 0; mi < 5; ++mi) {
      const int mode = modes[mi];
      std::vector<WarProbe> probes_(kSelProbes, kSelProbes + sizeof(kSelProbes) / sizeof(kSelProbes[0]));
      if (!sel_only) probes_.insert(probes_.end(), kWarProbes, kWarProbes + sizeof(kWarProbes) / sizeof(kWarProbes[0]));
      for (const WarProbe& pr : probes_) {
        size_t bad = 0, checked = 0, q[4] = {0, 0, 0, 0}, half[2] = {0, 0};
        float ex_lo = 0, ex_hi = 0;
        int launches = 0;
        double co = 0;
        std::vector<std::vector<Stamp>> all;
        if (mode >= 0) launch_aggr_mode(mode, sa, d_aout, d_as, agg_blocks, agg_iters / 3);
        for (int r = 0; r < 4; ++r) {
          if (mode >= 0 && hipStreamQuery(sa) == hipSuccess) break;
          hipLaunchKernelGGL(pr.kern, dim3(vic_blocks), dim3(256), 0, sb, d_w, d_vs);
          CK(hipStreamSynchronize(sb));
          CK(hipMemcpy(w.data(), d_w, w.size() * 4, hipMemcpyDeviceToHost));
          vs.resize(vic_blocks);
          CK(hipMemcpy(vs.data(), d_vs, vs.size() * sizeof(Stamp), hipMemcpyDeviceToHost));
          all.push_back(vs);
          for (size_t i = 0; i < w.size(); i += 2) {
            const bool blo = w[i] != pr.expect_lo, bhi = w[i + 1] != pr.expect_hi;
            if (blo || bhi) { if (!bad) { ex_lo = w[i]; ex_hi = w[i + 1]; } ++bad; ++q[((i / 2) & 63) >> 4]; half[0] += blo; half[1] += bhi; }
          }
          checked += w.size() / 2;
          ++launches;
        }
        if (mode >= 0) {
          CK(hipStreamSynchronize(sa));
          CK(hipMemcpy(as_.data(), d_as, as_.size() * sizeof(Stamp), hipMemcpyDeviceToHost));
          for (auto& v : all) co += coresident(v, as_);
          if (launches) co /= launches;
        }
        printf("  %-38s %-36s launches %d co-resident %5.1f%%  wrong %8zu / %zu  (%zu | %zu | %zu | %zu ; lo %zu hi %zu)\n", mode_name[mi], pr.name,
               launches, 100 * co, bad, checked, q[0], q[1], q[2], q[3], half[0], half[1]);
        if (bad) printf("      first wrong lane: got (%.9g, %.9g), expected (%.9g, %.9g)\n", ex_lo, ex_hi, pr.expect_lo, pr.expect_hi);
        total += bad;
      }
    }
  }
  printf("RESULT: %s (%zu wrong values in total)\n", total ? "CORRUPTION REPRODUCED" : "no corruption observed", total);
  return 0;
}
