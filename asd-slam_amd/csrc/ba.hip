// ba.hip -- Optimizer::PoseOptimization and the numeric core of Optimizer::LocalBundleAdjustment
// on gfx950, fp64.
//
// Reference: src/vslam/src/Optimizer.cc:239-413, :415-734; the arithmetic lives in the vendored g2o
// (src/g2o_catkin): Levenberg loop optimization_algorithm_levenberg.cpp:61-189, quadratic forms
// base_unary_edge.hpp:43-72 / base_binary_edge.hpp:55-120, Schur complement + lambda handling
// block_solver.hpp:354-604, Huber robust_kernel_impl.cpp:78-91, Jacobians
// types_six_dof_expmap.cpp:115-151,372-394, SE3 exp se3quat.h:223-257, dense solve
// linear_solver_dense.h:65-113.
//
// PoseOptimization (one 6-dof vertex, <= a few thousand unary edges) runs as ONE persistent
// workgroup: residuals, Jacobians, the 6x6 normal equations, the Levenberg trial loop, the four
// re-classification rounds -- all control flow stays on the device, no host round trip.
//
// LocalBundleAdjustment: g2o walks pointer graphs and hash maps edge by edge; here the problem is
// flat arrays in HBM and every step is a data-parallel kernel with FIXED reduction shapes (no
// float atomics), so results are bit-reproducible run to run:
//   k_ba_error      edge-parallel residual + robust cost            -> per-block partial sums
//   k_ba_linearize  edge-parallel Jacobians, Huber weight, per-edge blocks (Hpp/Hll/Hpl parts)
//   k_ba_reduce_*   per-pose / per-landmark segmented sums (CSR built once per round on the host)
//   (k_ba_edge_y, rounds 1-4: Y = B*Dinv, c = B*db per edge; since round 5 formed inside k_ba_schur)
//   k_ba_schur      one workgroup per upper 6x6 block: Hpp + lambda I - sum_l (B Dinv) B^T  (dense)
//   k_ba_chol       blocked Cholesky + triangular solves in one workgroup
//   k_ba_step       landmark-parallel xl = Dinv (bl - B^T xp) + pose-parallel oplus into the TRIAL buffer, gain-ratio partials
// The Levenberg accept/reject logic (optimization_algorithm_levenberg.cpp:61-189) runs ON THE DEVICE since round 3: the scalars
// live in LmState, a one-thread kernel (k_ba_lm_control) behind every trial takes the decision the host used to take after a
// synchronisation, and every other kernel reads lambda / "is this step needed" from that state -- a step whose answer is no
// exits at once.  The host enqueues whole iterations ("blocks" = linearise group + trial group + control) ahead and
// synchronises once per chunk of blocks instead of once per trial.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>
#include <thread>
#include <mutex>
#include <condition_variable>

#include "ctx.h"
#include "resolve2.h"
#include "se3.h"

namespace {

// "const float deltaMono = sqrt(5.991)" (Optimizer.cc:271,530): the kernel delta is a float
constexpr float kChi2Mono = 5.991f;
__host__ __device__ inline double huber_delta() { return (double)(float)2.4476519360399936; }  // (float)sqrt(5.991)

// ---------------------------------------------------------------- deterministic block reductions
// Wave-level sum of up to 32 values per lane as a reduce-scatter: each butterfly step exchanges only
// the half of the values the lane does not keep, so 32 values cost 16+8+4+2+1+1 = 32 cross-lane moves
// instead of 32*6.  Afterwards lane l holds the wave total of value (l >> 1).  Fixed order: bit-reproducible.
template <int M>
__device__ inline void rs_step(double (&v)[2 * M], double (&o)[M], int off, bool up) {
#pragma unroll
  for (int k = 0; k < M; ++k) {
    const double send = up ? v[k] : v[k + M];
    const double keep = up ? v[k + M] : v[k];
    o[k] = keep + __shfl_xor(send, off);
  }
}
// The two widest steps (partner 32 and 16 lanes away) are exactly what gfx950's v_permlane32_swap / v_permlane16_swap
// do: swap the upper half (odd 16-lane rows) of one register with the lower half (even rows) of another.  After the
// swap both registers hold "my value + partner's value" operands lane by lane, so a step is 2 swaps + 1 add per double
// instead of 4 selects + 2 ds_bpermute + 1 add, and the sums are bit-identical (a + b == b + a).
template <int M, int OFF>
__device__ inline void rs_step_swap(double (&v)[2 * M], double (&o)[M]) {
  static_assert(OFF == 32 || OFF == 16, "swap steps exist for 32 and 16 lanes");
#pragma unroll
  for (int k = 0; k < M; ++k) {
    const unsigned alo = (unsigned)__double2loint(v[k]), ahi = (unsigned)__double2hiint(v[k]);
    const unsigned blo = (unsigned)__double2loint(v[k + M]), bhi = (unsigned)__double2hiint(v[k + M]);
    double x, y;
    if constexpr (OFF == 32) {
      const auto lo = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
      const auto hi = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
      x = __hiloint2double((int)hi[0], (int)lo[0]);
      y = __hiloint2double((int)hi[1], (int)lo[1]);
    } else {
      const auto lo = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
      const auto hi = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
      x = __hiloint2double((int)hi[0], (int)lo[0]);
      y = __hiloint2double((int)hi[1], (int)lo[1]);
    }
    o[k] = x + y;
  }
}

__device__ inline double wave_reduce_scatter32(double (&v)[32]) {
  const int lane = threadIdx.x & 63;
  double a[16], b[8], c[4], d[2], e[1];
#ifdef ASD_NO_PERMLANE_SWAP   // diagnostic build: the two widest steps by __shfl_xor as well
  rs_step<16>(v, a, 32, (lane & 32) != 0);
  rs_step<8>(a, b, 16, (lane & 16) != 0);
#else
  rs_step_swap<16, 32>(v, a);
  rs_step_swap<8, 16>(a, b);
#endif
  rs_step<4>(b, c, 8, (lane & 8) != 0);
  rs_step<2>(c, d, 4, (lane & 4) != 0);
  rs_step<1>(d, e, 2, (lane & 2) != 0);
  return e[0] + __shfl_xor(e[0], 1);
}

__device__ inline double wave_reduce_scatter64(double (&v)[64]) {  // lane l ends with the total of value l
  const int lane = threadIdx.x & 63;
  double z[32], a[16], b[8], c[4], d[2], e[1];
  rs_step_swap<32, 32>(v, z);
  rs_step_swap<16, 16>(z, a);
  rs_step<8>(a, b, 8, (lane & 8) != 0);
  rs_step<4>(b, c, 4, (lane & 4) != 0);
  rs_step<2>(c, d, 2, (lane & 2) != 0);
  rs_step<1>(d, e, 1, (lane & 1) != 0);
  return e[0];
}

// red: [NW][32] for N <= 32, [NW][64] for 32 < N <= 64
template <int N, int NW = 4>
__device__ inline void block_reduce(double (&v)[N], double* red, double* out /*[N] in LDS*/) {
  static_assert(N <= 64, "block_reduce handles up to 64 values");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if constexpr (N > 32) {
    double w[64];
#pragma unroll
    for (int k = 0; k < 64; ++k) w[k] = k < N ? v[k] : 0.0;
    const double tot = wave_reduce_scatter64(w);
    red[wave * 64 + lane] = tot;
    asd_syncthreads();
    if (threadIdx.x < N) {
      double s = red[threadIdx.x];
#pragma unroll
      for (int w2 = 1; w2 < NW; ++w2) s += red[w2 * 64 + threadIdx.x];
      out[threadIdx.x] = s;
    }
    asd_syncthreads();
    return;
  }
  if constexpr (N <= 2) {
#pragma unroll
    for (int k = 0; k < N; ++k) {
      double x = v[k];
      for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off);
      if (lane == 0) red[wave * 32 + k] = x;
    }
  } else {
    double w[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) w[k] = k < N ? v[k] : 0.0;
    const double tot = wave_reduce_scatter32(w);
    if ((lane & 1) == 0) red[wave * 32 + (lane >> 1)] = tot;
  }
  asd_syncthreads();
  if (threadIdx.x < N) {
    const int k = threadIdx.x;
    double s = red[k];
#pragma unroll
    for (int w2 = 1; w2 < NW; ++w2) s += red[w2 * 32 + k];  // fixed order
    out[k] = s;
  }
  asd_syncthreads();
}

// ---------------------------------------------------------------- P1: PoseOptimization, one workgroup
// Edge data ([n][6] = Xw, obs, info), the stored errors and the level / outlier flags live in LDS for
// the whole optimisation (64 B + 2 B per edge: n <= 2300 fits the 160 KiB of a CU; larger n streams
// from HBM).  Every pass over the edges evaluates residuals, robust cost AND the normal equations
// at the trial pose, so an accepted Levenberg trial already holds the next iteration's system
// (g2o recomputes both at the same estimate: identical values, one pass instead of three).
struct PoseOptArgs {
  int n;
  const int* n_dev;     // fused chains: the edge count produced on the device by k_pose_edges (n then = capacity)
  // fused chains, LDS form: the kernel itself compacts the matched keypoints (keypoint order = the edge order of Optimizer.cc:281)
  // straight into its LDS edge store -- no k_pose_edges launch, no [n][6] records in HBM.  g_src != null selects this.
  const int* g_src;       // [g_ncur] row of g_tab, or -1
  const uint8_t* g_hold;  // [g_ncur] or null: the keypoint carries its own point (g_own)
  const float* g_tab;     // [.][3]
  const float* g_own;     // [g_ncur][3] or null
  const float4* g_kp;     // (x, y, octave bits, angle)
  int g_ncur;
  const double* edges;  // [n][6] as uploaded: Xw, obs, inv_sigma2
  double fx, fy, cx, cy;
  double* soa_g;        // [8][n] global scratch (used when the problem does not fit LDS)
  uint8_t* flags_g;     // [2n] global scratch: level, outlier
  double pose0[7];      // initial pose (by value: no read of host memory on the kernel's critical path)
  const double* pose0_dev;  // non-null (asd_track_frame: the stage behind another PoseOptimization): the initial pose is read from here
  double* io_dev;           // non-null: the result block is written here as well (device memory, for the kernels of the next stage)
  const AsdBetweenArgs* between;   // device memory or null.  asd_track_frame, motion-model stage: the kernel ends with the work between the
                                   // two stages (needs io_dev).  A pointer, not a member: a larger argument block costs the kernel a scratch copy
  double isg_tab[16];   // MODE 2: the distinct information values ...
  const uint8_t* isgi;  // ... and each edge's index into them (device)
  double* io;           // out: pose[7], n_bad (as double), then outlier bytes at io + 8
  int use_lds;
  int debug;  // ASD_POSE_DEBUG: thread 0 prints passes / iterations per round
};

#ifndef ASD_POSE_THREADS
#define ASD_POSE_THREADS 512
#endif
constexpr int kPoseThreads = ASD_POSE_THREADS, kPoseWaves = kPoseThreads / 64;
struct PoseShared {
  Pose7 T, T0, Tbak, Teval;
  double H[36], b[6], x[6];
  double sums[29];
  double red[kPoseWaves * 32];
  double isg_tab[16];
  double pad_;  // sizeof % 16 == 0
  double lambda, ni, currentChi, iniChi, rho;
  int qmax, cont, ok, sys_valid, stop, npass;
  long long cyc[4];  // debug: edge loop, reduction, solve+oplus, LM bookkeeping
  long long wend[16], dbg[4];
};
static_assert(sizeof(PoseShared) % 16 == 0, "keeps the dynamic LDS region 16-B aligned");

// one pass over the active edges at pose T: sums[0..20] = upper H, [21..26] = b, [27] = robust chi2, [28] = #active
// 1/x and 1/sqrt(x) from the hardware seeds plus two Newton steps (error ~1 ulp for the well-scaled operands here:
// depths, chi2 values).  The IEEE-exact expansions of `1.0 / z`, `sqrt(c)` and `hd / s` cost ~45 fp64 issue slots per
// edge, a quarter of the loop; results move by an ulp, far inside the 1e-8 pose tolerance.
__device__ inline double nr_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}
__device__ inline double nr_rsqrt(double x) {
  double r = __builtin_amdgcn_rsq(x);
  r = fma(fma(-0.5 * x * r, r, 0.5), r, r);
  r = fma(fma(-0.5 * x * r, r, 0.5), r, r);
  return r;
}

// Inverse of a symmetric positive definite 6x6 (PoseOptimization's normal equations, the pivot blocks of the LocalBA solve).  It runs
// on one wave between two barriers and is a chain of DEPENDENT fp64 instructions (~40 cycles a link), so what counts is the length of
// that chain, not the flop count.  With A = [[P, Q^T], [Q, R]] (3x3 blocks), C = adj(P), dP = det(P):
//   U' = Q C,  S' = dP R - U' Q^T  (= dP x the Schur complement),  C' = adj(S'),  dS' = det(S'),  G = C' U'
//   W22 = (dP / dS') C',   W21 = -G / dS',   W11 = C / dP + U'^T G / (dP dS')
// i.e. nothing waits for a quotient: the two reciprocals (1 / dP, 1 / dS': seed + two Newton steps each) run beside the 3x3 products,
// ~22 links where inverting P, forming the Schur complement and inverting it one after the other has ~37 (and a Cholesky with
// forward / backward substitution ~90).  The block is first scaled by the power of two that brings its largest diagonal entry to
// [1, 2): the products of cofactors reach the twelfth power of the entries' magnitude (exact scaling: the result is bit for bit
// what the unscaled formulas give where those do not overflow).  Positive definiteness <=> the leading minors of P and of S' are
// positive (g2o's dense LDLT fails on a non-positive pivot, linear_solver_dense.h:96).
// A: lower triangle valid, row-major [6][6]; W: full inverse [6][6]
__device__ inline bool inv6_sym(const double* A, double (&W)[36]) {
  const double dmax = fmax(fmax(fmax(A[0], A[7]), fmax(A[14], A[21])), fmax(A[28], A[35]));
  // 2^-floor(log2 dmax) from the exponent field (dmax > 0 for anything positive definite; otherwise the minors below say no)
  const int ex = (int)((__double_as_longlong(dmax) >> 52) & 0x7ff);
  const double sc = __longlong_as_double((long long)(2046 - ex) << 52);
  const bool sane = dmax > 0 && ex > 0 && ex < 2046;
  const double p00 = A[0] * sc, p10 = A[6] * sc, p20 = A[12] * sc, p11 = A[7] * sc, p21 = A[13] * sc, p22 = A[14] * sc;
  double Q[9], R[6];   // R: (0,0) (1,0) (2,0) (1,1) (2,1) (2,2)
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) Q[r * 3 + c] = A[(3 + r) * 6 + c] * sc;
  {
    int q = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int r = c; r < 3; ++r) R[q++] = A[(3 + r) * 6 + 3 + c] * sc;
  }
  // C = adj(P) (symmetric), dP
  const double c00 = p11 * p22 - p21 * p21, c01 = p20 * p21 - p10 * p22, c02 = p10 * p21 - p20 * p11;
  const double c11 = p00 * p22 - p20 * p20, c12 = p10 * p20 - p00 * p21, c22 = p00 * p11 - p10 * p10;
  const double dP = p00 * c00 + p10 * c01 + p20 * c02;
  const double r1 = nr_rcp(dP);
  const double CF[9] = {c00, c01, c02, c01, c11, c12, c02, c12, c22};
  double U[9];   // U' = Q C
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) U[r * 3 + c] = Q[r * 3] * CF[c] + Q[r * 3 + 1] * CF[3 + c] + Q[r * 3 + 2] * CF[6 + c];
  double S[6];   // S' = dP R - U' Q^T: (0,0) (1,0) (2,0) (1,1) (2,1) (2,2)
  {
    int q = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int r = c; r < 3; ++r) { S[q] = dP * R[q] - (U[r * 3] * Q[c * 3] + U[r * 3 + 1] * Q[c * 3 + 1] + U[r * 3 + 2] * Q[c * 3 + 2]); ++q; }
  }
  const double s00 = S[0], s10 = S[1], s20 = S[2], s11 = S[3], s21 = S[4], s22 = S[5];
  const double e00 = s11 * s22 - s21 * s21, e01 = s20 * s21 - s10 * s22, e02 = s10 * s21 - s20 * s11;
  const double e11 = s00 * s22 - s20 * s20, e12 = s10 * s20 - s00 * s21, e22 = s00 * s11 - s10 * s10;
  const double dS = s00 * e00 + s10 * e01 + s20 * e02;
  const double r2 = nr_rcp(dS);
  const double EF[9] = {e00, e01, e02, e01, e11, e12, e02, e12, e22};
  double G[9];   // G = C' U'
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) G[r * 3 + c] = EF[r * 3] * U[c] + EF[r * 3 + 1] * U[3 + c] + EF[r * 3 + 2] * U[6 + c];
  const double k22 = dP * r2 * sc, k21 = -r2 * sc, k11a = r1 * sc, k11b = r1 * r2 * sc;   // (the input scale goes back in here)
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const double utg = U[r] * G[c] + U[3 + r] * G[3 + c] + U[6 + r] * G[6 + c];   // (U'^T G)[r][c]
      W[r * 6 + c] = CF[r * 3 + c] * k11a + utg * k11b;
      const double w21 = G[r * 3 + c] * k21;
      W[(3 + r) * 6 + c] = w21;
      W[c * 6 + 3 + r] = w21;
      W[(3 + r) * 6 + 3 + c] = EF[r * 3 + c] * k22;
    }
  return sane && p00 > 0 && c22 > 0 && dP > 0 && s00 > 0 && e22 > 0 && dS > 0;
}

// Where the solver keeps its edges.  MODE 0: [6][n] doubles in global memory (problems that do not fit LDS); MODE 1: the same
// in LDS (50 B / edge with the flags); MODE 2: the compact LDS form, 35 B / edge -- X, Y, Z as doubles, the observation as
// two floats and the information value as an index into a <= 16-entry table.  The host picks MODE 2 when every observation
// is exactly representable in f32 (keypoint coordinates are f32 in the reference: cv::KeyPoint::pt) and the edges use at
// most 16 distinct information values (one per pyramid level, Optimizer.cc:300): nothing is rounded.  2000 edges then take
// 69 KB (130 KB with f64 observations and per-edge stored errors), so the workgroup fits on a CU beside a resident ASDNet
// workgroup (70-78 KB) instead of waiting for a whole CU to drain (rocprof under the pipeline: PoseOptimization of the
// local-map stage 146 us at best, 292 us median).
// g2o keeps every edge's error "as last computed"; the solver does not store them: the re-classification at the end of a
// round re-evaluates an inlier edge at the pose of the round's LAST pass (S.Teval) with the pass's own arithmetic
// (pass_error), which reproduces the value the pass computed bit for bit.
template <int MODE>
struct EdgeStore {
  double* soa;            // MODE 0 / 1: [6][n]
  double* xyz;            // MODE 2: LDS [3][n]
  float2* uv;             //         LDS [n]
  uint8_t* isgi;          //         LDS [n]
  const double* isg_tab;  //         LDS [16]
  int n;
  __device__ inline void load(int i, double& X, double& Y, double& Z, double& u, double& v, double& isg) const {
    if constexpr (MODE == 2) {
      X = xyz[i]; Y = xyz[n + i]; Z = xyz[2 * n + i];
      const float2 o = uv[i];
      u = (double)o.x; v = (double)o.y;
      isg = isg_tab[isgi[i]];
    } else {
      X = soa[i]; Y = soa[n + i]; Z = soa[2 * n + i];
      u = soa[3 * n + i]; v = soa[4 * n + i]; isg = soa[5 * n + i];
    }
  }
};

// residual of one edge the way a pass evaluates it: rotation matrix + translation, one Newton reciprocal (iz = 0 for an
// inactive lane).  Shared by pose_pass and the re-classification so both produce the same bits.
__device__ __forceinline__ void pass_error(const double (&R)[9], double ttx, double tty, double ttz, double X, double Y, double Z, double ou,
                                           double ov, double fx, double fy, double cx, double cy, bool act, double& iz, double& xz, double& yz,
                                           double& e0, double& e1) {
  const double x = R[0] * X + R[1] * Y + R[2] * Z + ttx;
  const double y = R[3] * X + R[4] * Y + R[5] * Z + tty;
  const double z = R[6] * X + R[7] * Y + R[8] * Z + ttz;
  iz = act ? nr_rcp(z) : 0.0;
  xz = x * iz; yz = y * iz;
  e0 = ou - (xz * fx + cx);
  e1 = ov - (yz * fy + cy);
}

// soa = [X | Y | Z | u | v | inv_sigma2], each [n]: consecutive lanes read consecutive doubles (the [n][6]
// records this replaces put 8 lanes on every LDS bank pair)
template <int MODE, class A>
__device__ inline void pose_pass(A& a, const EdgeStore<MODE>& E, const uint8_t* lvl, PoseShared& S, bool robust) {
  // fp64 issue on ONE CU bounds this loop (a wave64 fp64 op takes 4 cycles, a division ~35 ops), so the
  // arithmetic is kept lean: rotation matrix instead of the quaternion sandwich, one reciprocal per
  // edge, weighted Jacobian rows shared by all 27 accumulators.
  const Pose7 T = S.T;
  if (threadIdx.x == 0) S.Teval = T;  // the pose the stored-error semantics refer to (see EdgeStore)
  double R[9];
  quat_to_rot(T, R);
  // wave-uniform: keep the pose in scalar registers (it comes out of LDS, which the compiler cannot prove uniform)
  auto uni = [](double v) { return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v))); };
#pragma unroll
  for (int q = 0; q < 9; ++q) R[q] = uni(R[q]);
  const double ttx = uni(T.tx), tty = uni(T.ty), ttz = uni(T.tz);
  const double hd = huber_delta(), hd2 = hd * hd;
  double acc[29];
#pragma unroll
  for (int k = 0; k < 29; ++k) acc[k] = 0.0;
  const int n = E.n;   // (the edge count of the fused chains is made on the device: a.n is then only a capacity)
  const long long c0 = a.debug ? clock64() : 0;
  // Branch-free body (inactive or out-of-range lanes run with weight 0 and keep their stored error), unrolled by
  // two so the scheduler can interleave two independent edges: with one wave per SIMD nothing else hides the
  // fp64 dependency chains.  J0[4] = J1[3] = 0 by construction: the products they would enter are left out.
  const int iters = (n + kPoseThreads - 1) / kPoseThreads;
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    // a wave whose 64 edges of this iteration all lie beyond the list is done (wave-uniform): with 1250 edges on 512 threads the third
    // iteration holds edges for four of the eight waves only -- the other four ran it with weight 0, a sixth of the pass's fp64 issue slots
    if (it * kPoseThreads + (int)(threadIdx.x & ~63u) >= n) break;
    const int i0 = threadIdx.x + it * kPoseThreads;
    const int i = min(i0, n - 1);
    const bool act = i0 < n && !lvl[i];
    double X, Y, Z, ou, ov, isg;
    E.load(i, X, Y, Z, ou, ov, isg);
    double iz, xz, yz, e0, e1;
    pass_error(R, ttx, tty, ttz, X, Y, Z, ou, ov, a.fx, a.fy, a.cx, a.cy, act, iz, xz, yz, e0, e1);
    const double c = (e0 * e0 + e1 * e1) * isg;
    // Huber (robust_kernel_impl.cpp:78-91); a wave of 64 edges practically always holds an outlier, so no branch
    const bool hub = robust && c > hd2;
    const double rs = nr_rsqrt(hub ? c : hd2);  // 1/sqrt(chi2); the operand is >= hd2 > 0 on every lane
    const double r0 = hub ? 2 * (c * rs) * hd - hd2 : c;
    const double w = hub ? hd * rs : 1.0;
    const double fiz = a.fx * iz, giz = a.fy * iz;
    // Jacobian rows (types_six_dof_expmap.cpp:382-394), [omega | upsilon]: J0 = (a0 a1 a2 a3 0 a5), J1 = (b0 b1 b2 0 b4 b5)
    const double a0 = xz * yz * a.fx, a1 = -(1 + xz * xz) * a.fx, a2 = yz * a.fx, a3 = -fiz, a5 = xz * fiz;
    const double b0 = (1 + yz * yz) * a.fy, b1 = -xz * yz * a.fy, b2 = -xz * a.fy, b4 = -giz, b5 = yz * giz;
    const double om = act ? isg * w : 0.0;
    const double w0 = om * a0, w1 = om * a1, w2 = om * a2, w3 = om * a3, w5 = om * a5;
    const double v0 = om * b0, v1 = om * b1, v2 = om * b2, v4 = om * b4, v5 = om * b5;
    acc[0] += w0 * a0 + v0 * b0;  acc[1] += w0 * a1 + v0 * b1;  acc[2] += w0 * a2 + v0 * b2;
    acc[3] += w0 * a3;            acc[4] += v0 * b4;            acc[5] += w0 * a5 + v0 * b5;
    acc[6] += w1 * a1 + v1 * b1;  acc[7] += w1 * a2 + v1 * b2;  acc[8] += w1 * a3;
    acc[9] += v1 * b4;            acc[10] += w1 * a5 + v1 * b5;
    acc[11] += w2 * a2 + v2 * b2; acc[12] += w2 * a3;           acc[13] += v2 * b4;
    acc[14] += w2 * a5 + v2 * b5;
    acc[15] += w3 * a3;           /* acc[16]: H(3,4) = 0 */     acc[17] += w3 * a5;
    acc[18] += v4 * b4;           acc[19] += v4 * b5;
    acc[20] += w5 * a5 + v5 * b5;
    acc[21] -= w0 * e0 + v0 * e1; acc[22] -= w1 * e0 + v1 * e1; acc[23] -= w2 * e0 + v2 * e1;
    acc[24] -= w3 * e0;           acc[25] -= v4 * e1;           acc[26] -= w5 * e0 + v5 * e1;
    acc[27] += act ? r0 : 0.0;
    acc[28] += act ? 1.0 : 0.0;
  }
  const long long c1 = a.debug ? clock64() : 0;
  if (a.debug && (threadIdx.x & 63) == 0) S.wend[threadIdx.x >> 6] = c1 - c0;
  // per-wave totals (reduce-scatter: lane l ends with the wave's total of value l >> 1) into S.red[wave][32]; the cross-wave sum is
  // pose_sums_wave0's, behind ONE barrier
  {
    double w[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) w[k] = k < 29 ? acc[k] : 0.0;
    const double tot = wave_reduce_scatter32(w);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if ((lane & 1) == 0) S.red[wave * 32 + (lane >> 1)] = tot;
  }
  if (threadIdx.x == 0) {
    ++S.npass;
    if (a.debug) {
      S.cyc[0] += c1 - c0; S.cyc[1] += clock64() - c1;
      long long mx = 0;
      for (int w = 0; w < kPoseWaves; ++w) mx = S.wend[w] > mx ? S.wend[w] : mx;
      S.dbg[0] += mx;
    }
  }
}
// wave 0, after the barrier behind pose_pass: S.sums[k] = sum over the waves in wave order (what block_reduce did behind a second
// barrier).  The only readers in the Levenberg loop are this wave's own lanes, in program order behind these stores.
__device__ inline void pose_sums_wave0(PoseShared& S) {
  const int lane = threadIdx.x & 63;
  if (lane < 29) {
    double s = S.red[lane];
#pragma unroll
    for (int w2 = 1; w2 < kPoseWaves; ++w2) s += S.red[w2 * 32 + lane];  // fixed order
    S.sums[lane] = s;
  }
}

__device__ inline void pose_take_system(PoseShared& S) {  // thread 0: sums -> H, b
  int k = 0;
  for (int r = 0; r < 6; ++r)
    for (int c = r; c < 6; ++c) { S.H[r * 6 + c] = S.sums[k]; S.H[c * 6 + r] = S.sums[k]; ++k; }
  for (int r = 0; r < 6; ++r) S.b[r] = S.sums[21 + r];
}

// MODE (the EdgeStore form) is a template parameter, not a run-time switch: with a pointer that may be LDS or global the
// compiler falls back to FLAT loads, whose latency dominated the edge loop.
// (A = const PoseOptArgs: the kernel's argument segment)
template <int MODE, class A>
__device__ __forceinline__ void pose_opt_body(A& a) {
  // (a = the kernel's argument block itself.  No private copy of it: a copy that is indexed at run time (pose0[t], isg_tab[t]) lives in
  // scratch memory, and every a.fx / a.n of the passes then is a scratch load)
  const unsigned long long rt_start = __builtin_amdgcn_s_memrealtime();   // 100 MHz, device-wide: comparable with other kernels' stamps (ASD_TIMING)
  int ne = a.n;   // edge count (fused chains: made on the device below, a.n is then only the capacity)
  constexpr int kGatherChunks = 9;   // 150 KB / 35 B per edge: at most 4388 keypoints reach the LDS form
  __shared__ int g_cnt[kGatherChunks * kPoseWaves + 1];
  const bool gather = MODE == 2 && a.g_src != nullptr;
  if (gather) {   // count the matched keypoints per (chunk, wave), scan: edge count and every wave's first edge
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nchunks = (a.g_ncur + kPoseThreads - 1) / kPoseThreads;
    for (int c = 0; c < nchunks; ++c) {
      const int j = c * kPoseThreads + threadIdx.x;
      const bool has = j < a.g_ncur && ((a.g_hold && a.g_hold[j]) || a.g_src[j] >= 0);
      const unsigned long long m = __ballot(has);
      if (lane == 0) g_cnt[c * kPoseWaves + wave] = __popcll(m);
    }
    asd_syncthreads();
    if (threadIdx.x == 0) {
      int sum = 0;
      for (int i = 0; i < nchunks * kPoseWaves; ++i) { const int v = g_cnt[i]; g_cnt[i] = sum; sum += v; }
      g_cnt[kGatherChunks * kPoseWaves] = sum;
    }
    asd_syncthreads();
    ne = g_cnt[kGatherChunks * kPoseWaves];
  } else if (a.n_dev) {   // the matches were made on the device (asd_track_*): the edge count is only known there
    ne = *a.n_dev;
  }
  if ((gather || a.n_dev) && ne < 3) {   // Optimizer.cc:323-324: fewer than 3 correspondences -> pose untouched, nothing marked
    if (threadIdx.x == 0) {   // (static indices: a run-time index into the kernel arguments makes the compiler keep a private copy in scratch)
#pragma unroll
      for (int q = 0; q < 7; ++q) { const double v = a.pose0_dev ? a.pose0_dev[q] : a.pose0[q]; a.io[q] = v; if (a.io_dev) a.io_dev[q] = v; }
      a.io[7] = 0.0;
      if (a.io_dev) a.io_dev[7] = 0.0;
    }
    if (gather) {   // the per-keypoint form of the flags (below): all clear, and the edge count behind them
      unsigned long long* og8 = reinterpret_cast<unsigned long long*>(a.io + 8);
      unsigned long long* od8 = reinterpret_cast<unsigned long long*>(a.io_dev + 8);
      const int nw = (a.g_ncur + 7) / 8;
      for (int i = threadIdx.x; i < nw; i += kPoseThreads) { og8[i] = 0ull; if (a.io_dev) od8[i] = 0ull; }
      if (threadIdx.x == 0) { a.io[8 + nw] = (double)ne; if (a.io_dev) a.io_dev[8 + nw] = (double)ne; }
    }
    if (a.between) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asd_syncthreads();
      asd_between_body(*a.between, threadIdx.x, kPoseThreads);
    }
    return;
  }
  // this workgroup is the critical path of the tracking thread and usually shares its CU with ASDNet workgroups of
  // the read-ahead extractor: ask the SIMD arbiters to issue its waves first
  __builtin_amdgcn_s_setprio(3);
  __shared__ PoseShared S;
  extern __shared__ __attribute__((aligned(16))) double dyn[];
  const int t = threadIdx.x;
  EdgeStore<MODE> E{};
  E.n = ne;
  uint8_t *lvl, *outl;
  // a.edges = the [n][6] records as uploaded (coalesced read, transposed write); a.isgi = information-value index per edge
  if constexpr (MODE == 2) {
    E.xyz = dyn;
    E.uv = reinterpret_cast<float2*>(dyn + (size_t)3 * ne);
    E.isgi = reinterpret_cast<uint8_t*>(E.uv + ne);
    lvl = E.isgi + ne;
    outl = lvl + ne;
    E.isg_tab = S.isg_tab;
    float* uvf = reinterpret_cast<float*>(E.uv);
    if (gather) {
      const int lane = t & 63, wave = t >> 6;
      const int nchunks = (a.g_ncur + kPoseThreads - 1) / kPoseThreads;
      for (int c = 0; c < nchunks; ++c) {
        const int j = c * kPoseThreads + t;
        int row = -1;
        bool mine = false;
        if (j < a.g_ncur) {
          if (a.g_hold && a.g_hold[j]) mine = true;
          else row = a.g_src[j];
        }
        const bool has = mine || row >= 0;
        const unsigned long long m = __ballot(has);
        if (has) {
          const int e = g_cnt[c * kPoseWaves + wave] + __popcll(m & ((1ull << lane) - 1));
          const float* X = mine ? a.g_own + 3 * (size_t)j : a.g_tab + 3 * (size_t)row;
          const float4 k = a.g_kp[j];
          E.xyz[e] = (double)X[0]; E.xyz[(size_t)ne + e] = (double)X[1]; E.xyz[(size_t)2 * ne + e] = (double)X[2];
          uvf[2 * e] = k.x; uvf[2 * e + 1] = k.y;           // kpUn.pt is f32 in the reference: nothing is rounded
          E.isgi[e] = (uint8_t)__float_as_int(k.z);          // octave -> index into the information-value table
        }
      }
    } else {
    for (int i = t; i < 6 * ne; i += kPoseThreads) {
      const int k = i % 6, e = i / 6;
      const double v = a.edges[i];
      if (k < 3) E.xyz[(size_t)k * ne + e] = v;
      else if (k < 5) uvf[2 * e + (k - 3)] = (float)v;  // exact: checked on the host
    }
    for (int i = t; i < ne; i += kPoseThreads) E.isgi[i] = a.isgi[i];
    }
    if (t == 0) {
#pragma unroll
      for (int q = 0; q < 16; ++q) S.isg_tab[q] = a.isg_tab[q];
    }
  } else {
    if constexpr (MODE == 1) {
      E.soa = dyn;
      lvl = reinterpret_cast<uint8_t*>(dyn + (size_t)6 * ne);
      outl = lvl + ne;
    } else {
      E.soa = a.soa_g;
      lvl = a.flags_g;
      outl = a.flags_g + ne;
    }
    for (int i = t; i < 6 * ne; i += kPoseThreads) E.soa[(size_t)(i % 6) * ne + i / 6] = a.edges[i];
  }
  for (int i = t; i < ne; i += kPoseThreads) { lvl[i] = 0; outl[i] = 0; }
  if (t == 0) {
    Pose7 T0{a.pose0[0], a.pose0[1], a.pose0[2], a.pose0[3], a.pose0[4], a.pose0[5], a.pose0[6]};
    if (a.pose0_dev) T0 = Pose7{a.pose0_dev[0], a.pose0_dev[1], a.pose0_dev[2], a.pose0_dev[3], a.pose0_dev[4], a.pose0_dev[5], a.pose0_dev[6]};
    quat_normalize(T0.qx, T0.qy, T0.qz, T0.qw);
    S.T0 = T0;
    S.T = T0;
  }
  asd_syncthreads();
  bool robust = true;
  int nBad = 0;
  bool flags_changed = true;   // did the previous round's re-classification change any flag?
  for (int round = 0; round < 4; ++round) {
    // A round is a deterministic function of (input pose, active set, robust kernel on / off): it restarts from the input pose
    // (Optimizer.cc:337).  If the previous round's re-classification changed no flag, rounds 1 and 2 (robust kernel still on) would
    // repeat the previous round operation for operation -- same passes, same pose, same re-classification -- so they are not run:
    // the state they would leave is the state that is there.  (Typical steady-state frame: the outlier set is final after round 0 or 1.)
    const bool repeat = round >= 1 && round <= 2 && !flags_changed;
    if (!repeat) {
    if (t == 0) { S.T = S.T0; S.npass = 0; for (int q = 0; q < 4; ++q) { S.cyc[q] = 0; S.dbg[q] = 0; } }  // every round restarts from the input pose (Optimizer.cc:337)
    asd_syncthreads();
    pose_pass(a, E, lvl, S, robust);
    asd_syncthreads();
    if (t < 64) pose_sums_wave0(S);
    asd_syncthreads();
    const bool any_active = S.sums[28] > 0.5;
    if (any_active) {
      // ---- g2o optimize(10): Levenberg (optimization_algorithm_levenberg.cpp:61-189) as a state machine run by thread 0 between
      // the passes.  Two barriers per pass: one publishes the pose the pass evaluates (and whether there is a pass at all), one
      // collects the per-wave sums; the cross-wave sum, the accept / reject decision, the 6x6 solve and the pose update all happen
      // in wave 0 between them (the first form had four barriers per pass and took the sums through LDS twice).
      //   phase 1 = the pass evaluates a trial pose, 2 = it re-evaluates the current pose (after an iteration that ended on a
      //   rejected trial: the sums no longer describe the current estimate), 0 = the round's optimisation is over
      int nBadIt = 0, it = 0;
      // (both lambdas are forced inline: as an out-of-line function next_trial takes the kernel's argument block by reference, which
      //  makes the compiler keep a private copy of it in scratch -- 336 B per lane written at entry, every a.fx of the passes a scratch load)
      auto begin_iteration = [&]() __attribute__((always_inline)) {   // thread 0: the sums hold the system at the current estimate
        S.currentChi = S.sums[27];
        S.iniChi = S.sums[27];
        pose_take_system(S);
        if (it == 0) {
          double md = 0;
          for (int j = 0; j < 6; ++j) md = fmax(fabs(S.H[j * 7]), md);
          S.lambda = 1e-5 * md;
          S.ni = 2;
        }
        S.qmax = 0;
        S.rho = 0;
      };
      auto next_trial = [&]() __attribute__((always_inline)) {        // thread 0: solve (H + lambda I) x = b, T <- exp(x) T
        const long long c0 = a.debug ? clock64() : 0;
        S.Tbak = S.T;
        double Hl[36], W[36], x[6];
#pragma unroll
        for (int q = 0; q < 36; ++q) Hl[q] = S.H[q];
#pragma unroll
        for (int j = 0; j < 6; ++j) Hl[j * 7] += S.lambda;
        const bool ok = inv6_sym(Hl, W);   // positive definite <=> the leading minors checked inside are positive (the dense LDLT of g2o fails otherwise)
#pragma unroll
        for (int r = 0; r < 6; ++r) {
          double acc2 = 0.0;
#pragma unroll
          for (int c = 0; c < 6; ++c) acc2 += W[r * 6 + c] * S.b[c];
          x[r] = ok ? acc2 : 0.0;
        }
#pragma unroll
        for (int r = 0; r < 6; ++r) S.x[r] = x[r];
        S.ok = ok ? 1 : 0;
        const long long c1 = a.debug ? clock64() : 0;
        if (ok) S.T = pose_oplus(S.T, x);
        S.cont = 1;
        if (a.debug) { S.cyc[2] += clock64() - c0; S.dbg[1] += c1 - c0; }
      };
      if (t == 0) { begin_iteration(); next_trial(); }
      for (;;) {
        asd_syncthreads();               // S.T / S.cont of thread 0
        const int phase = S.cont;
        if (phase == 0) break;
        pose_pass(a, E, lvl, S, robust);
        asd_syncthreads();
        if (t < 64) {
          pose_sums_wave0(S);
          if (t == 0) {
            const long long c0 = a.debug ? clock64() : 0;
            bool fresh_system = phase == 2;   // the pass was a plain evaluation at the current pose: start the next iteration from it
            bool over = false;
            if (phase == 1) {
              double tempChi = S.sums[27];
              if (!S.ok) tempChi = 1.7976931348623157e308;
              double rho = S.currentChi - tempChi;
              double scale = 0;
              for (int j = 0; j < 6; ++j) scale += S.x[j] * (S.lambda * S.x[j] + S.b[j]);
              scale += 1e-3;
              rho *= nr_rcp(scale);    // (the IEEE division is ~15 dependent instructions on this one lane; rho's sign, its zero and the
                                       //  gain-ratio formula below are unaffected by the last ulp)
              bool accepted = false;
              if (rho > 0 && isfinite(tempChi)) {
                const double q = 2 * rho - 1;
                double alpha = 1. - q * q * q;
                alpha = fmin(alpha, 2. / 3.);
                S.lambda *= fmax(1. / 3., alpha);
                S.ni = 2;
                S.currentChi = tempChi;
                accepted = true;       // the sums hold the system at the accepted estimate
              } else {
                S.lambda *= S.ni;
                S.ni *= 2;
                S.T = S.Tbak;
              }
              S.rho = rho;
              S.qmax++;
              if (!(rho < 0 && S.qmax < 10)) {   // the iteration is over (levenberg.cpp:98-146)
                bool stop = S.qmax == 10 || rho == 0;
                if (!stop) {
                  if ((S.iniChi - S.currentChi) * 1e3 < S.iniChi) nBadIt++; else nBadIt = 0;
                  if (nBadIt >= 3) stop = true;
                }
                ++it;
                if (stop || it >= 10) over = true;
                else if (accepted) fresh_system = true;
                else { S.cont = 2; over = false; fresh_system = false; }   // next: a plain pass at the restored pose
                if (!over && !accepted) { if (a.debug) S.cyc[3] += clock64() - c0; goto published; }
              }
            }
            if (a.debug) S.cyc[3] += clock64() - c0;
            if (over) S.cont = 0;
            else {
              if (fresh_system) begin_iteration();
              next_trial();
            }
          published:;
          }
        }
      }
    } else {
      asd_syncthreads();
    }
    // ---- re-classification (Optimizer.cc:341-368)
    double nb[1] = {0.0};
    int chg = 0;
    {
      const Pose7 T = S.T, Te = S.Teval;
      double Re[9];
      quat_to_rot(Te, Re);
      const int n = ne;
      for (int i = t; i < n; i += kPoseThreads) {
        double X, Y, Z, ou, ov, isg, e0, e1;
        E.load(i, X, Y, Z, ou, ov, isg);
        if (outl[i]) {  // e->computeError() at the current estimate
          const double Xw[3] = {X, Y, Z};
          double Xc[3];
          pose_map(T, Xw, Xc);
          e0 = ou - (Xc[0] / Xc[2] * a.fx + a.cx);
          e1 = ov - (Xc[1] / Xc[2] * a.fy + a.cy);
        } else {        // inlier: active in every pass of this round, its error is the last pass's
          double iz, xz, yz;
          pass_error(Re, Te.tx, Te.ty, Te.tz, X, Y, Z, ou, ov, a.fx, a.fy, a.cx, a.cy, true, iz, xz, yz, e0, e1);
        }
        const float chi2 = (float)((e0 * e0 + e1 * e1) * isg);
        const uint8_t bad = chi2 > kChi2Mono ? 1 : 0;
        chg |= bad != outl[i];
        outl[i] = bad; lvl[i] = bad;
        if (bad) nb[0] += 1.0;
      }
    }
    flags_changed = asd_syncthreads_or(chg) != 0;
    block_reduce<1, kPoseWaves>(nb, S.red, S.sums);
    nBad = (int)(S.sums[0] + 0.5);
    asd_syncthreads();
    if (a.debug && t == 0)
      printf("[pose_opt] round %d: %d passes, nBad %d; cycles/pass: edges %lld (slowest wave %lld) reduce %lld solve+oplus %lld (solve %lld) accept %lld\n", round, S.npass,
             nBad, S.cyc[0] / S.npass, S.dbg[0] / S.npass, S.cyc[1] / S.npass, S.cyc[2] / S.npass, S.dbg[1] / S.npass, S.cyc[3] / S.npass);
    }   // !repeat
    if (round == 2) robust = false;  // e->setRobustKernel(0)
    if (ne < 10) break;             // optimizer.edges().size() < 10
  }
  if (t == 0) {
    a.io[0] = S.T.qx; a.io[1] = S.T.qy; a.io[2] = S.T.qz; a.io[3] = S.T.qw;
    a.io[4] = S.T.tx; a.io[5] = S.T.ty; a.io[6] = S.T.tz;
    a.io[7] = (double)nBad;
    if (a.io_dev) {
      a.io_dev[0] = S.T.qx; a.io_dev[1] = S.T.qy; a.io_dev[2] = S.T.qz; a.io_dev[3] = S.T.qw;
      a.io_dev[4] = S.T.tx; a.io_dev[5] = S.T.ty; a.io_dev[6] = S.T.tz;
      a.io_dev[7] = (double)nBad;
    }
  }
  if (gather) {
    // Fused chains: the flags go out per KEYPOINT (0 where the keypoint carries no edge) with the edge count behind them, so that the
    // host's completion is two copies instead of a walk over the keypoints that rebuilds the edge order.  The edge of keypoint j is
    // found as in the gather (prefix table g_cnt still in LDS); the flags are collected in LDS first -- the edge store is free now --
    // and leave in 8-byte stores.
    asd_syncthreads();
    uint8_t* kpf = reinterpret_cast<uint8_t*>(dyn);
    const int lane = t & 63, wave = t >> 6;
    const int nchunks = (a.g_ncur + kPoseThreads - 1) / kPoseThreads;
    uint8_t mine[kGatherChunks];
#pragma unroll
    for (int c = 0; c < kGatherChunks; ++c) {   // (read every edge's flag before the first store over the edge store: outl sits behind it, kpf in front)
      mine[c] = 0;
      if (c >= nchunks) continue;
      const int j = c * kPoseThreads + t;
      const bool has = j < a.g_ncur && ((a.g_hold && a.g_hold[j]) || a.g_src[j] >= 0);
      const unsigned long long m = __ballot(has);
      mine[c] = has ? outl[g_cnt[c * kPoseWaves + wave] + __popcll(m & ((1ull << lane) - 1))] : 0;
    }
    asd_syncthreads();
#pragma unroll
    for (int c = 0; c < kGatherChunks; ++c) {
      const int j = c * kPoseThreads + t;
      if (c < nchunks && j < (a.g_ncur + 7) / 8 * 8) kpf[j] = j < a.g_ncur ? mine[c] : 0;
    }
    asd_syncthreads();
    unsigned long long* og8 = reinterpret_cast<unsigned long long*>(a.io + 8);
    const unsigned long long* k8 = reinterpret_cast<const unsigned long long*>(kpf);
    const int nw = (a.g_ncur + 7) / 8;
    for (int i = t; i < nw; i += kPoseThreads) og8[i] = k8[i];
    if (t == 0) { a.io[8 + nw] = (double)ne; a.io[8 + nw + 1] = (double)(rt_start & 0xffffffffull); a.io[8 + nw + 2] = (double)(__builtin_amdgcn_s_memrealtime() & 0xffffffffull); }
    if (a.io_dev) {
      unsigned long long* od8 = reinterpret_cast<unsigned long long*>(a.io_dev + 8);
      for (int i = t; i < nw; i += kPoseThreads) od8[i] = k8[i];
      if (t == 0) a.io_dev[8 + nw] = (double)ne;
    }
    if (a.between) {   // the flags and the pose above are this workgroup's own stores: complete, then visible to all its waves
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asd_syncthreads();
      asd_between_body(*a.between, t, kPoseThreads);
    }
    return;
  }
  // outlier flags to the (pinned host) io block, eight per 8-byte store
  unsigned long long* og8 = reinterpret_cast<unsigned long long*>(a.io + 8);
  for (int i = t; i < (ne + 7) / 8; i += kPoseThreads) {
    unsigned long long w = 0;
    for (int k = 0; k < 8; ++k)
      if (8 * i + k < ne) w |= (unsigned long long)outl[8 * i + k] << (8 * k);
    og8[i] = w;
  }
}

template <int MODE>
__global__ __launch_bounds__(kPoseThreads) void k_pose_opt(PoseOptArgs a_in) { pose_opt_body<MODE>(a_in); }

// The claim replay and PoseOptimization of a tracking stage as ONE workgroup (asd_track_frame): resolve2_body on the solver's 512 threads,
// then the solver (gather form: its edges come from the match table the replay has just written).  Stand-alone, k_pose_opt needs most of a
// CU -- 8 waves x 221 registers, 70 KB of LDS -- and waited 35-50 us per launch for one beside the extractor's ASDNet workgroups, which
// take every slot a finished workgroup frees (device-clock stamps, profiles/r04_chain_device_clock.txt); here it inherits the replay's.
template <int KIND, int QPT>
__global__ __launch_bounds__(kPoseThreads) void k_resolve_pose(Resolve2Args r, PoseOptArgs a_in) {
  resolve2_body<KIND, QPT, kPoseThreads>(r);
  // the match table is this workgroup's own stores: complete, then visible to all its waves (and no stale line in the vector L1)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asd_syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  pose_opt_body<2>(a_in);
}

// ---------------------------------------------------------------- fused tracking chains: edges made on the device
// The unary edges of PoseOptimization (Optimizer.cc:272-310) straight from match results that live on the device: keypoint
// j gets an edge when it holds a map point -- src[j] >= 0 names a row of the world-position table tab (the match k_resolve
// wrote), or hold[j] != 0 says the keypoint already held one on entry with its position in own[j].  Edge order = keypoint
// order, as the reference's loop over mvpMapPoints (:281): the same records, in the same order, as asd_pose_optimize
// receives from a host that packs them -- so the fused chain returns the same bits as the two separate calls.
// Observation = the keypoint's f32 coordinates, information = invSigma2 of its octave (both exactly representable in the
// compact EdgeStore form: no check needed).
struct PoseEdgesArgs {
  int n_cur;
  const int* src;          // [n_cur] row of tab or -1
  const uint8_t* hold;     // [n_cur] or null
  const float* tab;        // [.][3]
  const float* own;        // [n_cur][3] or null
  const float4* kp;        // (x, y, octave bits, angle)
  float inv_sigma2[ASD_MAX_LEVELS];
  double* edges;           // out [n][6]
  uint8_t* isgi;           // out [n]
  int* n_out;              // out
};
__global__ __launch_bounds__(1024) void k_pose_edges(PoseEdgesArgs a) {
  __shared__ int wave_tot[16], base;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (t == 0) base = 0;
  asd_syncthreads();
  for (int j0 = 0; j0 < a.n_cur; j0 += 1024) {
    const int j = j0 + t;
    int row = -1;
    bool mine = false;
    if (j < a.n_cur) {
      if (a.hold && a.hold[j]) mine = true;
      else row = a.src[j];
    }
    const bool has = mine || row >= 0;
    const unsigned long long m = __ballot(has);
    if (lane == 0) wave_tot[wave] = __popcll(m);
    asd_syncthreads();
    int off = base;
    for (int w = 0; w < wave; ++w) off += wave_tot[w];
    if (has) {
      const int e = off + __popcll(m & ((1ull << lane) - 1));
      const float* X = mine ? a.own + 3 * (size_t)j : a.tab + 3 * (size_t)row;
      const float4 k = a.kp[j];
      const int oct = __float_as_int(k.z);
      double* E = a.edges + (size_t)e * 6;
      E[0] = (double)X[0]; E[1] = (double)X[1]; E[2] = (double)X[2];
      E[3] = (double)k.x; E[4] = (double)k.y; E[5] = (double)a.inv_sigma2[oct];
      a.isgi[e] = (uint8_t)oct;
    }
    asd_syncthreads();
    if (t == 0) { int s = 0; for (int w = 0; w < 16; ++w) s += wave_tot[w]; base += s; }
    asd_syncthreads();
  }
  if (t == 0) *a.n_out = base;
}

// ---------------------------------------------------------------- LocalBA kernels
// Levenberg state of the running round (device memory; k_ba_lm_control mirrors it into pinned host memory after every trial)
struct LmState {
  double lambda, ni, currentChi, iniChi, rho;
  unsigned long long maxdiag_bits;   // max |diag(H)| over poses and landmarks as the bit pattern of a non-negative double
  int it, qmax, nBad, done, need_lin, cur, trials, iters_done, chol_ok, iterations, first, pad_;   // cur: which of the two estimate buffers holds the accepted estimate
};
struct LmLog { double lambda, cur, temp, scale; };   // one record per trial (ASD_BA_DEBUG, tests)
constexpr int kLmLogCap = 128;

constexpr int kPoseSplit = 8;   // workgroups that share one pose's edge sum (k_ba_reduce_pose)

struct BaDev {
  // problem
  int P, L, E;
  // the estimate is double buffered: [lm->cur] = the accepted estimate, [1 - lm->cur] = the trial (solve + oplus of the accepted one).
  // Accepting a trial flips lm->cur, rejecting it does nothing: no backup / restore passes (g2o's push / pop, sparse_optimizer.cpp:422-435)
  Pose7* pose[2];                    // [P]
  double* pts[2];                    // [L][3]
  const int* e_pt; const int* e_ps;  // [E]
  const double* obs; const double* info;  // [E][2], [E]
  double* err;                       // [E][2]
  const uint8_t* lvl;                // [E] g2o level of the edge in the running round: 1 = moved out by the outlier gating (Optimizer.cc:612-631)
  double fx, fy, cx, cy;
  // active structure of the current round
  int Ea, nPf, nLa;
  const int* act;        // [Ea] edge ids, grouped by landmark (CSR order), within a landmark by pose h-index
  const int* pose_h;     // [P]  -> h index or -1
  const int* pt_h;       // [L]
  const int* pose_of_h;  // [nPf]
  const int* pt_of_h;    // [nLa]
  const int* pt_start;   // [nLa+1] into act-order k
  const int* ps_start;   // [nPf+1]
  const int* ps_edges;   // k indices of each free pose's edges
  const int* ps_h;       // ... and the landmark (h index) of each of them
  // per active edge (index k)
  double* Bk;   // [Ea][18]
  double* Hc;   // [Ea][27]  (21 upper of Jc^T W Jc, 6 of b)
  double* Hl;   // [Ea][9]   (6 upper of Jp^T W Jp, 3 of b)
  // per vertex
  double* Hpp;  // [nPf][27]
  double* HppPart;  // [nPf][kPoseSplit][27] partial sums of k_ba_reduce_pose
  double* Hll;  // [nLa][9]
  double* Dinv; // [nLa][6]
  double* db;   // [nLa][3]
  double* x;    // [6 nPf + 3 nLa]
  // dense system
  double* A;    // [n][n]
  double* Apack; // the same system as the solve kernel keeps it: lower triangle by 6x6 blocks, block (I, J <= I) at (I (I + 1) / 2 + J) * 36
  double* bs;   // [n]
  LmState* lm;        // device
  LmState* lm_host;   // pinned mirror, written by k_ba_lm_control / k_ba_lm_begin
  LmLog* lm_log;      // device, [kLmLogCap]
  double* partial;    // device: per-block partial sums: [0, scale_off) residual cost, [scale_off, ..) gain-ratio terms
  int* tickets;       // device, zero between launches: [0] workgroups of k_ba_error that have stored their partial sum, [1 + h] of k_ba_reduce's
                      // kPoseSplit workgroups of pose h.  The workgroup that draws the last ticket does what a kernel of its own did up to
                      // round 4 (the Levenberg control behind k_ba_error, the second stage of the pose sums) and puts the counter back to 0
  int scale_off, gE, gL, gP;
};

// lambda of the trial being computed: computeLambdaInit (block_solver.hpp:564-580, tau = 1e-5) until the first control step has
// stored it -- a pure function of the state, so the trial kernels need no "initialise lambda" launch in front of them
__device__ inline double lm_lambda(const LmState* lm) {
  return lm->first ? 1e-5 * __longlong_as_double((long long)lm->maxdiag_bits) : lm->lambda;
}

__device__ inline void ba_project_error(const BaDev& d, int e, int buf) {
  double Xc[3];
  pose_map(d.pose[buf][d.e_ps[e]], d.pts[buf] + 3 * d.e_pt[e], Xc);
  d.err[2 * e] = d.obs[2 * e] - (Xc[0] / Xc[2] * d.fx + d.cx);
  d.err[2 * e + 1] = d.obs[2 * e + 1] - (Xc[1] / Xc[2] * d.fy + d.cy);
}

// computeActiveErrors + activeRobustChi2 (sparse_optimizer.cpp:61-114): partial[blockIdx] = block sum
// Hand-over of a few values between the workgroups of ONE launch (they may sit on different XCDs, each with an L2 of its own): the values
// are stored write-through (agent-scope atomic stores: sc1) and read back the same way, every wave waits for its stores, the
// workgroup draws a ticket.  No release fence: an agent-scope release is a write-back of the XCD's whole L2 -- behind k_ba_linearize that is
// 12 MB of per-edge blocks, and __threadfence() in every workgroup made a trial 32 us SLOWER than the two launches this replaces.
__device__ inline void st_agent(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline double ld_agent(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// true in every thread of the workgroup that is the last of the launch's `total` to arrive here; the counter is back at zero afterwards
__device__ inline bool last_workgroup(int* ticket, int total) {
  __shared__ int last_flag;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asd_syncthreads();
  if (threadIdx.x == 0) {
    const int drawn = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last_flag = drawn == total - 1;
    if (last_flag) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asd_syncthreads();
  return last_flag != 0;
}
constexpr int kLmThreads = 256, kLmMaxPartials = 2048;
__device__ void lm_begin_body(const BaDev& d, int iterations, double* sh);
__device__ void lm_control_body(const BaDev& d, double* sh);

// computeActiveErrors + the robustified chi2 of the estimate under test; the workgroup that finishes last runs the Levenberg control on the
// launch's sums (k_ba_lm_begin / k_ba_lm_control were launches of their own up to round 4: 5 us per trial for one thread's work)
__global__ __launch_bounds__(256) void k_ba_error(BaDev d, int robust, int always, int iterations) {
  __shared__ double red[4 * 32], out[1];
  __shared__ double sh[kLmMaxPartials];
  static_assert(kLmThreads == 256, "the control code runs in a workgroup of k_ba_error");
  if (!always && d.lm->done) return;
  const int buf = always ? d.lm->cur : 1 - d.lm->cur;   // the round's first pass looks at the accepted estimate, every other at the trial
  const int k = blockIdx.x * 256 + threadIdx.x;
  double part[1] = {0.0};
  if (k < d.Ea) {
    const int e = d.act[k];
    if (!d.lvl[e]) {   // level-1 edges are outside the optimisation: their stored error stays as last computed, they add nothing
      ba_project_error(d, e, buf);
      const double c = (d.err[2 * e] * d.err[2 * e] + d.err[2 * e + 1] * d.err[2 * e + 1]) * d.info[e];
      double r0 = c, r1;
      if (robust) huber(c, huber_delta(), r0, r1);
      part[0] = r0;
    }
  }
  block_reduce<1>(part, red, out);
  if (threadIdx.x == 0) st_agent(d.partial + blockIdx.x, out[0]);
  if (!last_workgroup(d.tickets, (int)gridDim.x)) return;
  if (always) lm_begin_body(d, iterations, sh);
  else lm_control_body(d, sh);
}

// linearizeOplus + constructQuadraticForm per active edge (uses the stored error, like g2o)
__global__ __launch_bounds__(256) void k_ba_linearize(BaDev d, int robust) {
  if (d.lm->done || !d.lm->need_lin) return;   // a retry after a rejected trial keeps the system (only lambda changes)
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= d.Ea) return;
  const int e = d.act[k];
  if (d.lvl[e]) {   // a level-1 edge keeps its place in the round-1 structure and contributes exact zeros to every sum it is part of
    double* hl = d.Hl + (size_t)k * 9;
    for (int q = 0; q < 9; ++q) hl[q] = 0.0;
    if (d.pose_h[d.e_ps[e]] >= 0) {
      double* hc = d.Hc + (size_t)k * 27;
      for (int q = 0; q < 27; ++q) hc[q] = 0.0;
      double* b = d.Bk + (size_t)k * 18;
      for (int q = 0; q < 18; ++q) b[q] = 0.0;
    }
    return;
  }
  const int cur = d.lm->cur;
  const Pose7 T = d.pose[cur][d.e_ps[e]];
  double Xc[3], R[9], Jc[12], Jp[6];
  pose_map(T, d.pts[cur] + 3 * d.e_pt[e], Xc);
  quat_to_rot(T, R);
  const double x = Xc[0], y = Xc[1], z = Xc[2];
  const double tmp[6] = {d.fx, 0, -x / z * d.fx, 0, d.fy, -y / z * d.fy};
  for (int r = 0; r < 2; ++r)
    for (int c = 0; c < 3; ++c)
      Jp[r * 3 + c] = -1. / z * (tmp[r * 3] * R[c] + tmp[r * 3 + 1] * R[3 + c] + tmp[r * 3 + 2] * R[6 + c]);
  jac_pose(x, y, z, d.fx, d.fy, Jc);
  const double e0 = d.err[2 * e], e1 = d.err[2 * e + 1];
  double w = 1.0, r0;
  if (robust) huber((e0 * e0 + e1 * e1) * d.info[e], huber_delta(), r0, w);
  const double om = d.info[e] * w;
  double* hl = d.Hl + (size_t)k * 9;
  int q = 0;
  for (int r = 0; r < 3; ++r)
    for (int c = r; c < 3; ++c) hl[q++] = om * (Jp[r] * Jp[c] + Jp[3 + r] * Jp[3 + c]);
  for (int r = 0; r < 3; ++r) hl[6 + r] = -om * (Jp[r] * e0 + Jp[3 + r] * e1);
  if (d.pose_h[d.e_ps[e]] >= 0) {
    double* hc = d.Hc + (size_t)k * 27;
    q = 0;
    for (int r = 0; r < 6; ++r)
      for (int c = r; c < 6; ++c) hc[q++] = om * (Jc[r] * Jc[c] + Jc[6 + r] * Jc[6 + c]);
    for (int r = 0; r < 6; ++r) hc[21 + r] = -om * (Jc[r] * e0 + Jc[6 + r] * e1);
    double* b = d.Bk + (size_t)k * 18;
    for (int r = 0; r < 6; ++r)
      for (int c = 0; c < 3; ++c) b[r * 3 + c] = om * (Jc[r] * Jp[c] + Jc[6 + r] * Jp[3 + c]);
  }
}

__device__ inline void atomic_max_pos_double(unsigned long long* addr, double v) {
  // for non-negative doubles the IEEE bit pattern is monotone; max is order independent
  atomicMax(addr, (unsigned long long)__double_as_longlong(v));
}

// Hpp_i, bp_i = segmented sum over the pose's edges with a fixed shape, in two stages: kPoseSplit workgroups per pose take the
// edges i = s, s + kPoseSplit, ... of its list (thread-strided partial sums, fixed-order block reduction), a second kernel adds the
// kPoseSplit partials in index order.  Bit-reproducible.  (One workgroup per pose -- 24 workgroups reading 6 MB of per-edge blocks --
// took 16 us per iteration.)
// one launch, two kinds of workgroup: [0, nPf * kPoseSplit) the first stage of the pose sums, the rest the landmark sums
__global__ __launch_bounds__(256) void k_ba_reduce(BaDev d) {
  __shared__ double red[4 * 32], out[27];
  if (d.lm->done || !d.lm->need_lin) return;
  const int npose_blk = d.nPf * kPoseSplit;
  if ((int)blockIdx.x < npose_blk) {
    const int h = blockIdx.x / kPoseSplit, sp = blockIdx.x % kPoseSplit;
    const int b = d.ps_start[h], e = d.ps_start[h + 1];
    double acc[27];
#pragma unroll
    for (int q = 0; q < 27; ++q) acc[q] = 0.0;
    for (int i = b + sp * 256 + threadIdx.x; i < e; i += 256 * kPoseSplit) {
      const double* hc = d.Hc + (size_t)d.ps_edges[i] * 27;
#pragma unroll
      for (int q = 0; q < 27; ++q) acc[q] += hc[q];
    }
    block_reduce<27>(acc, red, out);
    if (threadIdx.x < 27) st_agent(d.HppPart + ((size_t)h * kPoseSplit + sp) * 27 + threadIdx.x, out[threadIdx.x]);
    // second stage, by the last of the pose's workgroups to get here: the kPoseSplit partials in index order (k_ba_reduce_pose2 up to round 4)
    if (!last_workgroup(d.tickets + 1 + h, kPoseSplit)) return;
    const int t = threadIdx.x;
    if (t >= 27) return;
    double s = 0.0;
    for (int q = 0; q < kPoseSplit; ++q) s += ld_agent(d.HppPart + ((size_t)h * kPoseSplit + q) * 27 + t);
    d.Hpp[(size_t)h * 27 + t] = s;
    // diagonal entries of the upper-packed 6x6: 0, 6, 11, 15, 18, 20
    if (t == 0 || t == 6 || t == 11 || t == 15 || t == 18 || t == 20) atomic_max_pos_double(&d.lm->maxdiag_bits, fabs(s));
    return;
  }
  const int h = ((int)blockIdx.x - npose_blk) * 256 + threadIdx.x;
  if (h >= d.nLa) return;
  double s[9];
  for (int q = 0; q < 9; ++q) s[q] = 0.0;
  for (int k = d.pt_start[h]; k < d.pt_start[h + 1]; ++k)
    for (int q = 0; q < 9; ++q) s[q] += d.Hl[(size_t)k * 9 + q];
  for (int q = 0; q < 9; ++q) d.Hll[(size_t)h * 9 + q] = s[q];
  atomic_max_pos_double(&d.lm->maxdiag_bits, fmax(fabs(s[0]), fmax(fabs(s[3]), fabs(s[5]))));
}

// per landmark: Dinv = (Hll + lambda I)^-1 (symmetric 3x3 by cofactors) and db = Dinv bl.  Evaluated where it is needed (per edge in
// k_ba_edge_y, per landmark in k_ba_backsub) instead of by a kernel of its own: forty flops against a launch, and the same
// instructions give the same bits in both places.
__device__ inline void point_dinv(const double* H, double lambda, double (&I)[9], double (&dbv)[3]) {
  const double a = H[0] + lambda, b = H[1], c = H[2], dd = H[3] + lambda, e = H[4], f = H[5] + lambda;
  const double c00 = dd * f - e * e, c01 = c * e - b * f, c02 = b * e - c * dd;
  const double id = 1.0 / (a * c00 + b * c01 + c * c02);
  I[0] = c00 * id; I[1] = c01 * id; I[2] = c02 * id; I[3] = c01 * id; I[4] = (a * f - c * c) * id; I[5] = (b * c - a * e) * id;
  I[6] = c02 * id; I[7] = (b * c - a * e) * id; I[8] = (a * dd - b * b) * id;
  for (int r = 0; r < 3; ++r) dbv[r] = I[r * 3] * H[6] + I[r * 3 + 1] * H[7] + I[r * 3 + 2] * H[8];
}

// Schur complement, one workgroup per upper block (bi <= bj):
//   A(bi,bj) = [bi==bj](Hpp + lambda I) - sum_pairs Y_a B_b^T ;  bs(bi) = bp - sum_edges c
// pairs of a block are (ka, kb) act-order indices, grouped on the host (fixed order).
// Y_a = B_a Dinv_l and c_k = B_k (Dinv_l bl) are formed HERE, per pair / per edge, from the landmark's Hll (round 5; up to round 4 a kernel
// of its own, k_ba_edge_y, wrote them per edge: 11 us and 5.6 MB per trial for ~100 flops per pair that this kernel's lanes have time for --
// they wait for their gathers).  The landmark of a pair / of a pose's edge comes with the lists (pair_h, ps_h), so its Hll is fetched beside
// the B blocks, not behind them.
struct SchurBlocks {
  const int* blk_i; const int* blk_j;  // [nblk]
  const int* pair_start;               // [nblk+1]
  const int2* pairs;
  const int* pair_h;                   // landmark (h index) of every pair: its two edges observe the same one
};
// Latency, not arithmetic, sets this kernel's time (6 M fp64 FMAs chip-wide): a diagonal block has ~1200 pairs whose operands sit
// behind two dependent gathers (pair -> edge indices -> 2 x 144 B).  Round 2's form -- 256 threads walking the list five deep, then
// the same workgroup walking the pose's ck list -- took 24 us.  Now: 512 threads, two pairs in flight per thread (indices of both
// first, then all four operand blocks), and the right-hand side sums bs = bp - sum ck in workgroups of their own ([nblk, nblk + nPf)).
constexpr int kSchurThreads = 512;
__global__ __launch_bounds__(kSchurThreads) void k_ba_schur(BaDev d, SchurBlocks sb, int nblk) {
  __shared__ double red[(kSchurThreads / 64) * 64], out[64];
  if (d.lm->done) return;
  const int t = threadIdx.x;
  if ((int)blockIdx.x >= nblk) {   // right-hand side of one pose
    const int h = blockIdx.x - nblk;
    double acc[6] = {0, 0, 0, 0, 0, 0};
    const double lam = lm_lambda(d.lm);
    for (int i = d.ps_start[h] + t; i < d.ps_start[h + 1]; i += kSchurThreads) {
      const double* B = d.Bk + (size_t)d.ps_edges[i] * 18;
      double I[9], dbv[3];
      point_dinv(d.Hll + (size_t)d.ps_h[i] * 9, lam, I, dbv);
#pragma unroll
      for (int r = 0; r < 6; ++r) acc[r] += B[r * 3] * dbv[0] + B[r * 3 + 1] * dbv[1] + B[r * 3 + 2] * dbv[2];
    }
    block_reduce<6, kSchurThreads / 64>(acc, red, out);
    if (t < 6) d.bs[6 * h + t] = d.Hpp[(size_t)h * 27 + 21 + t] - out[t];
    return;
  }
  const double lambda = lm_lambda(d.lm);
  const int blk = blockIdx.x, bi = sb.blk_i[blk], bj = sb.blk_j[blk];
  const int n = 6 * d.nPf;
  double acc[36];
#pragma unroll
  for (int q = 0; q < 36; ++q) acc[q] = 0.0;
  const int pend = sb.pair_start[blk + 1];
  for (int p0 = sb.pair_start[blk] + t; p0 < pend; p0 += 2 * kSchurThreads) {
    const int p1 = p0 + kSchurThreads;
    const bool two = p1 < pend;
    const int2 pr0 = sb.pairs[p0], pr1 = sb.pairs[two ? p1 : p0];
    const int h0 = sb.pair_h[p0], h1 = sb.pair_h[two ? p1 : p0];
    const double* Y0 = d.Bk + (size_t)pr0.x * 18;
    const double* B0 = d.Bk + (size_t)pr0.y * 18;
    const double* Y1 = d.Bk + (size_t)pr1.x * 18;
    const double* B1 = d.Bk + (size_t)pr1.y * 18;
    const double* H0 = d.Hll + (size_t)h0 * 9;
    const double* H1 = d.Hll + (size_t)h1 * 9;
    double y0[18], b0[18], y1[18], b1[18], hh0[9], hh1[9];
#pragma unroll
    for (int q = 0; q < 18; ++q) { y0[q] = Y0[q]; b0[q] = B0[q]; y1[q] = Y1[q]; b1[q] = B1[q]; }
#pragma unroll
    for (int q = 0; q < 9; ++q) { hh0[q] = H0[q]; hh1[q] = H1[q]; }
    {   // Y = B_a Dinv, row by row in place
      double I[9], dbv[3];
      point_dinv(hh0, lambda, I, dbv);
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        const double u0 = y0[r * 3], u1 = y0[r * 3 + 1], u2 = y0[r * 3 + 2];
#pragma unroll
        for (int q = 0; q < 3; ++q) y0[r * 3 + q] = u0 * I[q] + u1 * I[3 + q] + u2 * I[6 + q];
      }
      point_dinv(hh1, lambda, I, dbv);
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        const double u0 = y1[r * 3], u1 = y1[r * 3 + 1], u2 = y1[r * 3 + 2];
#pragma unroll
        for (int q = 0; q < 3; ++q) y1[r * 3 + q] = u0 * I[q] + u1 * I[3 + q] + u2 * I[6 + q];
      }
    }
    if (!two) {
#pragma unroll
      for (int q = 0; q < 18; ++q) y1[q] = 0.0;   // the second slot of a lone pair adds exact zeros
    }
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        acc[r * 6 + c] += y0[r * 3] * b0[c * 3] + y0[r * 3 + 1] * b0[c * 3 + 1] + y0[r * 3 + 2] * b0[c * 3 + 2];
        acc[r * 6 + c] += y1[r * 3] * b1[c * 3] + y1[r * 3 + 1] * b1[c * 3 + 1] + y1[r * 3 + 2] * b1[c * 3 + 2];
      }
  }
  block_reduce<36, kSchurThreads / 64>(acc, red, out);
  if (t < 36) {
    const int r = t / 6, c = t % 6;
    double v = -out[t];
    if (bi == bj) {
      const int rr = min(r, c), cc = max(r, c);
      const int q = rr * 6 - rr * (rr - 1) / 2 + (cc - rr);  // index in the upper-packed 6x6
      v += d.Hpp[(size_t)bi * 27 + q] + (r == c ? lambda : 0.0);
    }
    d.A[(size_t)(6 * bi + r) * n + 6 * bj + c] = v;
    if (bi != bj) d.A[(size_t)(6 * bj + c) * n + 6 * bi + r] = v;
    d.Apack[((size_t)bj * (bj + 1) / 2 + bi) * 36 + c * 6 + r] = v;   // lower block (bj, bi) = this upper block transposed
  }
}

// Dense SPD solve with the matrix resident in LDS: lower triangle packed by 6x6 blocks
// (block (I,J), J <= I at ((I(I+1)/2 + J) * 36), up to 32 pose blocks = 152 KB).  Right-looking
// block Cholesky with the right-hand side carried along as one more block row (so the forward
// substitution costs no extra barriers), then block backward substitution.  One workgroup.
// The chain of barriers and the serial 6x6 factorisations bound this kernel, not its 1 MFLOP: divisions are
// replaced by one reciprocal per pivot and the triangular block index comes from a table, not from sqrt().
constexpr int kCholThreads = 1024;
__global__ __launch_bounds__(kCholThreads) void k_ba_chol_lds(const double* __restrict__ A, const double* __restrict__ bs,
                                                              double* __restrict__ x, int n, LmState* lm) {
  // all LDS in the dynamic region (a static in front of it would shift its base off 16-B alignment)
  extern __shared__ __attribute__((aligned(16))) double L[];
  if (lm->done) return;
  const int t = threadIdx.x, nt = blockDim.x, nb = n / 6;
  const int nblk = nb * (nb + 1) / 2;
  double* xs = L + (size_t)nblk * 36;   // [192] right-hand side / solution
  double* invd = xs + 192;              // [192] reciprocals of the Cholesky pivots
  int& ok = *reinterpret_cast<int*>(invd + 192);
  short2* tri = reinterpret_cast<short2*>(invd + 194);  // [nblk] packed lower-triangle index -> (I, J)
#define LB(I, J) (L + ((size_t)((I) * ((I) + 1) / 2 + (J))) * 36)
  for (int I = t; I < nb; I += nt)
    for (int J = 0; J <= I; ++J) tri[I * (I + 1) / 2 + J] = make_short2((short)I, (short)J);
  if (t == 0) ok = 1;
  asd_syncthreads();
  for (int idx = t; idx < nblk * 36; idx += nt) {
    const int blk = idx / 36, e = idx % 36;
    const short2 ij = tri[blk];
    L[idx] = A[(size_t)(6 * ij.x + e / 6) * n + 6 * ij.y + e % 6];
  }
  for (int i = t; i < n; i += nt) xs[i] = bs[i];
  asd_syncthreads();
  for (int jb = 0; jb < nb; ++jb) {
    if (t == 0) {  // factor the diagonal block, forward-substitute its slice of the right-hand side: in registers
      double* ap = LB(jb, jb);
      double a[36], y[6], iv[6];
#pragma unroll
      for (int q = 0; q < 36; ++q) a[q] = ap[q];
#pragma unroll
      for (int q = 0; q < 6; ++q) y[q] = xs[6 * jb + q];
      bool good = true;
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        double dgl = a[j * 6 + j];
#pragma unroll
        for (int k = 0; k < j; ++k) dgl -= a[j * 6 + k] * a[j * 6 + k];
        if (!(dgl > 0)) { good = false; dgl = 1.0; }
        // 1/sqrt from the hardware seed + two Newton steps (~1 ulp) instead of the IEEE sqrt and division expansions: this
        // single-lane block factorisation is a chain of dependent fp64 instructions (44 cycles each), and the two expansions
        // were ~30 of the ~45 links per column
        const double inv = nr_rsqrt(dgl);
        dgl = dgl * inv;
        a[j * 6 + j] = dgl;
        iv[j] = inv;
#pragma unroll
        for (int i = j + 1; i < 6; ++i) {
          double s = a[i * 6 + j];
#pragma unroll
          for (int k = 0; k < j; ++k) s -= a[i * 6 + k] * a[j * 6 + k];
          a[i * 6 + j] = s * inv;
        }
#pragma unroll
        for (int c = j + 1; c < 6; ++c) a[j * 6 + c] = 0.0;
        double s = y[j];
#pragma unroll
        for (int k = 0; k < j; ++k) s -= a[j * 6 + k] * y[k];
        y[j] = s * inv;
      }
#pragma unroll
      for (int q = 0; q < 36; ++q) ap[q] = a[q];
#pragma unroll
      for (int q = 0; q < 6; ++q) { xs[6 * jb + q] = y[q]; invd[6 * jb + q] = iv[q]; }
      if (!good) ok = 0;
    }
    asd_syncthreads();
    const double* Ljj = LB(jb, jb);
    const double* iv = invd + 6 * jb;
    // panel: every row below solves against Ljj^T
    for (int row = t; row < (nb - jb - 1) * 6; row += nt) {
      const int I = jb + 1 + row / 6, r = row % 6;
      double* a = LB(I, jb) + r * 6;
      double v[6];
      for (int c = 0; c < 6; ++c) v[c] = a[c];
      for (int c = 0; c < 6; ++c) {
        double s = v[c];
        for (int k = 0; k < c; ++k) s -= v[k] * Ljj[c * 6 + k];
        v[c] = s * iv[c];
      }
      for (int c = 0; c < 6; ++c) a[c] = v[c];
    }
    asd_syncthreads();
    // trailing update: block (I,K), I >= K > jb, and the right-hand side rows below
    const int m = nb - jb - 1, mblk = m * (m + 1) / 2;
    for (int idx = t; idx < mblk * 36 + m * 6; idx += nt) {
      if (idx < mblk * 36) {
        const int bq = idx / 36, e = idx % 36;
        const short2 ik = tri[bq];
        const int I = jb + 1 + ik.x, K = jb + 1 + ik.y, r = e / 6, c = e % 6;
        const double* li = LB(I, jb) + r * 6;
        const double* lk = LB(K, jb) + c * 6;
        double s = 0.0;
        for (int k = 0; k < 6; ++k) s += li[k] * lk[k];
        LB(I, K)[e] -= s;
      } else {
        const int row = idx - mblk * 36;
        const int I = jb + 1 + row / 6, r = row % 6;
        const double* a = LB(I, jb) + r * 6;
        double s = 0.0;
        for (int c = 0; c < 6; ++c) s += a[c] * xs[6 * jb + c];
        xs[6 * I + r] -= s;
      }
    }
    asd_syncthreads();
  }
  // backward substitution (L^T)
  for (int jb = nb - 1; jb >= 0; --jb) {
    if (t == 0) {
      const double* ap = LB(jb, jb);
      double a[36], y[6], iv[6];
#pragma unroll
      for (int q = 0; q < 36; ++q) a[q] = ap[q];
#pragma unroll
      for (int q = 0; q < 6; ++q) { y[q] = xs[6 * jb + q]; iv[q] = invd[6 * jb + q]; }
#pragma unroll
      for (int r = 5; r >= 0; --r) {
        double s = y[r];
#pragma unroll
        for (int k = r + 1; k < 6; ++k) s -= a[k * 6 + r] * y[k];
        y[r] = s * iv[r];
      }
#pragma unroll
      for (int q = 0; q < 6; ++q) xs[6 * jb + q] = y[q];
    }
    asd_syncthreads();
    for (int row = t; row < jb * 6; row += nt) {
      const int I = row / 6, r = row % 6;  // x_I -= L(jb,I)^T x_jb
      const double* a = LB(jb, I);
      double s = 0.0;
      for (int c = 0; c < 6; ++c) s += a[c * 6 + r] * xs[6 * jb + c];
      xs[6 * I + r] -= s;
    }
    asd_syncthreads();
  }
#undef LB
  for (int i = t; i < n; i += nt) x[i] = xs[i];
  if (t == 0) lm->chol_ok = ok;
}

// The same solve by block elimination with EXPLICIT inverses of the 6x6 pivot blocks (default up to 31 pose blocks):
//   for j: W = A_jj^-1 ; T_I = A_Ij W (I > j) ; A_IK -= T_I A_Kj^T (I >= K > j) ; b_I -= T_I b_j        (block LDL^T, L_Ij = T_I)
//   then y_j = W_j b_j and x_j = y_j - sum_{I > j} T_Ij^T x_I.
// Why: the Cholesky above spends 24 x ~3.7k cycles in the single-lane factorisation of the diagonal block -- six pivots, each a
// chain of ~14 dependent fp64 instructions (pivot update, 1/sqrt with two Newton steps, column scaling) at ~44 cycles a link --
// plus a 21-link forward substitution per panel row.  The inverse of a symmetric 6x6 from two 3x3 cofactor inverses (P, then the
// Schur complement S = R - Q P^-1 Q^T) has two reciprocals on its critical path instead of six inverse square roots (~36 links),
// one wave computes it from LDS, and the panel / trailing
// update / both substitutions become 6-term dot products with no triangular dependency inside a block.
// Positive definiteness (linear_solver_dense.h:96 fails on a non-positive LDLT pivot) is checked on the leading minors of P and S.
constexpr int kSolveThreads = 1024, kSolveMaxBlocks = 30;
// Schedule of a step j (two barriers):
//   [panel]   T_I = A_Ij W_j for I > j, one thread per ROW of a block (six independent dot products: the fp64 latency of one chain hides
//             behind the other five); the same phase moves the PREVIOUS step's panel into A's place (nobody reads A_I,j-1 any more)
//   barrier
//   [update]  A_IK -= T_I A_Kj^T, again a row per thread.  Wave 0 takes the next pivot block A_j+1,j+1 first and inverts it at once
//             (look-ahead) while the other fifteen waves do the rest of the trailing matrix: the ~37 dependent fp64 instructions of the
//             inverse (~2k cycles) are the critical path of the whole factorisation and now run beside the update instead of after it
//   barrier
__device__ inline void solve_store_inverse(const double (&W)[36], double* dst, int lane) {
  if (lane < 36) {
    double v = 0.0;
#pragma unroll
    for (int q = 0; q < 36; ++q) v = lane == q ? W[q] : v;   // static register indices
    dst[lane] = v;
  }
}
__global__ __launch_bounds__(kSolveThreads) void k_ba_solve_lds(const double* __restrict__ A, const double* __restrict__ bs,
                                                                double* __restrict__ x, int n, LmState* lm) {
  extern __shared__ __attribute__((aligned(16))) double L[];
  if (lm->done) return;
  const int t = threadIdx.x, nt = blockDim.x, nb = n / 6, wave = t >> 6, lane = t & 63;
  const int nblk = nb * (nb + 1) / 2;
  double* Tp0 = L + (size_t)nblk * 36;                          // two panels [nb - 1][36]: T_I of the current and of the previous step
  double* xs = Tp0 + (size_t)2 * (kSolveMaxBlocks - 1) * 36;    // [192] right-hand side / solution
  double* Ww = xs + 192;                                        // [2][36] inverse of the current / next pivot block
  int& ok = *reinterpret_cast<int*>(Ww + 72);
  short2* tri = reinterpret_cast<short2*>(Ww + 74);             // [nblk] packed lower-triangle index -> (I, J)
#define LB(I, J) (L + ((size_t)((I) * ((I) + 1) / 2 + (J))) * 36)
  for (int I = t; I < nb; I += nt)
    for (int J = 0; J <= I; ++J) tri[I * (I + 1) / 2 + J] = make_short2((short)I, (short)J);
  if (t == 0) ok = 1;
  asd_syncthreads();
  for (int idx = t; idx < nblk * 36; idx += nt) L[idx] = A[idx];   // A = the packed lower triangle (k_ba_schur writes it in this layout)
  for (int i = t; i < n; i += nt) xs[i] = bs[i];
  asd_syncthreads();
  if (wave == 0) {   // the first pivot block
    double W[36];
    const bool good = inv6_sym(LB(0, 0), W);
    solve_store_inverse(W, Ww, lane);
    if (lane == 0 && !good) ok = 0;
  }
  asd_syncthreads();
  for (int jb = 0; jb < nb; ++jb) {
    const int m = nb - jb - 1;
    const double* Wc = Ww + (jb & 1) * 36;
    double* Tp = Tp0 + (size_t)(jb & 1) * (kSolveMaxBlocks - 1) * 36;
    // ---- panel rows: T_I[r][:] = A_Ij[r][:] W
    if (t < m * 6) {
      const int I = jb + 1 + t / 6, r = t % 6;
      const double* a = LB(I, jb) + r * 6;
      double av[6], o[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) av[k] = a[k];
#pragma unroll
      for (int c = 0; c < 6; ++c) o[c] = 0.0;
#pragma unroll
      for (int k = 0; k < 6; ++k)
#pragma unroll
        for (int c = 0; c < 6; ++c) o[c] += av[k] * Wc[k * 6 + c];
#pragma unroll
      for (int c = 0; c < 6; ++c) Tp[(t / 6) * 36 + r * 6 + c] = o[c];
    }
    // ... and the previous step's panel / pivot inverse take their places for the backward substitution (threads of the upper waves)
    if (jb > 0) {
      const double* Tq = Tp0 + (size_t)((jb - 1) & 1) * (kSolveMaxBlocks - 1) * 36;
      const int mp = m + 1;
      for (int idx = nt - 1 - t; idx < mp * 36 + 36; idx += nt) {
        if (idx < mp * 36) LB(jb + idx / 36, jb - 1)[idx % 36] = Tq[idx];
        else LB(jb - 1, jb - 1)[idx - mp * 36] = Ww[((jb - 1) & 1) * 36 + idx - mp * 36];
      }
    }
    asd_syncthreads();
    // ---- trailing update by rows; wave 0: next pivot block first, then its inverse (look-ahead)
    const int mblk = m * (m + 1) / 2;
    auto update_row = [&](int bq, int r) {   // row r of block (jb+1+ik.x, jb+1+ik.y)
      const short2 ik = tri[bq];
      const double* ti = Tp + ik.x * 36 + r * 6;
      const double* ak = LB(jb + 1 + ik.y, jb);
      double tv[6], o[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) tv[k] = ti[k];
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        double s2 = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) s2 += tv[k] * ak[c * 6 + k];
        o[c] = s2;
      }
      double* dst = LB(jb + 1 + ik.x, jb + 1 + ik.y) + r * 6;
#pragma unroll
      for (int c = 0; c < 6; ++c) dst[c] -= o[c];
    };
    if (wave == 0) {
      if (m > 0) {
        if (lane < 6) update_row(0, lane);            // block (jb+1, jb+1) = tri index 0
        double W[36];
        const bool good = inv6_sym(LB(jb + 1, jb + 1), W);
        solve_store_inverse(W, Ww + ((jb + 1) & 1) * 36, lane);
        if (lane == 0 && !good) ok = 0;
      }
    } else {
      const int items = (mblk - 1) * 6 + m * 6;       // the other blocks' rows, then the right-hand side rows
      for (int idx = t - 64; idx < items; idx += nt - 64) {
        if (idx < (mblk - 1) * 6) {
          update_row(1 + idx / 6, idx % 6);
        } else {
          const int row = idx - (mblk - 1) * 6;
          const double* ti = Tp + (row / 6) * 36 + (row % 6) * 6;
          double s2 = 0.0;
#pragma unroll
          for (int c = 0; c < 6; ++c) s2 += ti[c] * xs[6 * jb + c];
          xs[6 * (jb + 1) + row] -= s2;
        }
      }
    }
    asd_syncthreads();
  }
  // the last pivot's inverse (its panel is empty)
  if (t < 36) LB(nb - 1, nb - 1)[t] = Ww[((nb - 1) & 1) * 36 + t];
  asd_syncthreads();
  // y_j = W_j z_j, then x_j = y_j - sum_{I > j} T_Ij^T x_I right-looking from the last block
  double yv = 0.0;
  if (t < n) {
    const int j = t / 6, r = t % 6;
    const double* w = LB(j, j) + r * 6;
#pragma unroll
    for (int k = 0; k < 6; ++k) yv += w[k] * xs[6 * j + k];
  }
  asd_syncthreads();
  if (t < n) xs[t] = yv;
  asd_syncthreads();
  for (int jb = nb - 1; jb >= 1; --jb) {
    if (t < jb * 6) {
      const int K = t / 6, c = t % 6;
      const double* a = LB(jb, K);
      double s2 = 0.0;
#pragma unroll
      for (int r = 0; r < 6; ++r) s2 += a[r * 6 + c] * xs[6 * jb + r];
      xs[t] -= s2;
    }
    asd_syncthreads();
  }
#undef LB
  for (int i = t; i < n; i += nt) x[i] = xs[i];
  if (t == 0) lm->chol_ok = ok;
}

// dense SPD solve A x = bs (n = 6 nPf) in one workgroup: right-looking Cholesky on 6-wide panels,
// then forward / backward substitution by panels.  status = 0 if a pivot is not positive.
__global__ __launch_bounds__(1024) void k_ba_chol(double* __restrict__ A, const double* __restrict__ bs,
                                                  double* __restrict__ x, int n, LmState* lm) {
  __shared__ double Ljj[36];
  __shared__ int ok;
  if (lm->done) return;
  const int t = threadIdx.x, nt = blockDim.x;
  if (t == 0) ok = 1;
  asd_syncthreads();
  const int nb = n / 6;
  for (int jb = 0; jb < nb; ++jb) {
    const int j0 = jb * 6;
    if (t == 0) {  // factor the diagonal 6x6 block
      double a[36];
      for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) a[r * 6 + c] = A[(size_t)(j0 + r) * n + j0 + c];
      for (int j = 0; j < 6; ++j) {
        double dgl = a[j * 6 + j];
        for (int k = 0; k < j; ++k) dgl -= a[j * 6 + k] * a[j * 6 + k];
        if (!(dgl > 0)) { ok = 0; dgl = 1.0; }
        dgl = sqrt(dgl);
        a[j * 6 + j] = dgl;
        for (int i = j + 1; i < 6; ++i) {
          double s = a[i * 6 + j];
          for (int k = 0; k < j; ++k) s -= a[i * 6 + k] * a[j * 6 + k];
          a[i * 6 + j] = s / dgl;
        }
      }
      for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) {
        Ljj[r * 6 + c] = c <= r ? a[r * 6 + c] : 0.0;
        A[(size_t)(j0 + r) * n + j0 + c] = Ljj[r * 6 + c];
      }
    }
    asd_syncthreads();
    // panel: rows below, L[i][j0..j0+5] = A[i][j0..] * Ljj^-T
    for (int i = j0 + 6 + t; i < n; i += nt) {
      double row[6];
      for (int c = 0; c < 6; ++c) row[c] = A[(size_t)i * n + j0 + c];
      for (int c = 0; c < 6; ++c) {
        double s = row[c];
        for (int k = 0; k < c; ++k) s -= row[k] * Ljj[c * 6 + k];
        row[c] = s / Ljj[c * 6 + c];
      }
      for (int c = 0; c < 6; ++c) A[(size_t)i * n + j0 + c] = row[c];
    }
    asd_syncthreads();
    // trailing update (lower triangle incl. diagonal): A[i][k] -= sum_c L[i][j0+c] L[k][j0+c]
    const int m = n - j0 - 6;
    for (int idx = t; idx < m * m; idx += nt) {
      const int i = j0 + 6 + idx / m, k = j0 + 6 + idx % m;
      if (k > i) continue;
      double s = 0.0;
      for (int c = 0; c < 6; ++c) s += A[(size_t)i * n + j0 + c] * A[(size_t)k * n + j0 + c];
      A[(size_t)i * n + k] -= s;
    }
    asd_syncthreads();
  }
  // forward substitution L y = bs (y kept in x)
  for (int i = t; i < n; i += nt) x[i] = bs[i];
  asd_syncthreads();
  for (int jb = 0; jb < nb; ++jb) {
    const int j0 = jb * 6;
    if (t == 0)
      for (int r = 0; r < 6; ++r) {
        double s = x[j0 + r];
        for (int k = 0; k < r; ++k) s -= A[(size_t)(j0 + r) * n + j0 + k] * x[j0 + k];
        x[j0 + r] = s / A[(size_t)(j0 + r) * n + j0 + r];
      }
    asd_syncthreads();
    for (int i = j0 + 6 + t; i < n; i += nt) {
      double s = 0.0;
      for (int c = 0; c < 6; ++c) s += A[(size_t)i * n + j0 + c] * x[j0 + c];
      x[i] -= s;
    }
    asd_syncthreads();
  }
  // backward substitution L^T x = y
  for (int jb = nb - 1; jb >= 0; --jb) {
    const int j0 = jb * 6;
    if (t == 0)
      for (int r = 5; r >= 0; --r) {
        double s = x[j0 + r];
        for (int k = r + 1; k < 6; ++k) s -= A[(size_t)(j0 + k) * n + j0 + r] * x[j0 + k];
        x[j0 + r] = s / A[(size_t)(j0 + r) * n + j0 + r];
      }
    asd_syncthreads();
    for (int i = t; i < j0; i += nt) {
      double s = 0.0;
      for (int c = 0; c < 6; ++c) s += A[(size_t)(j0 + c) * n + i] * x[j0 + c];
      x[i] -= s;
    }
    asd_syncthreads();
  }
  if (t == 0) lm->chol_ok = ok;
}

// The step applied: one launch, two kinds of workgroup.  [0, gL): xl = Dinv (bl - B^T xp) and the trial point = accepted point + xl;
// [gL, gL + gP): the trial pose = exp(xp) * accepted pose (VertexSE3Expmap::oplusImpl).  Both write the TRIAL buffer and leave the
// accepted estimate alone; each workgroup also leaves its part of the gain-ratio denominator sum_j x_j (lambda x_j + b_j).
// solve for the landmarks (back-substitution), update the estimate (oplus), the gain ratio's denominator terms.  (Round 5 also tried the trial
// pass's errors and the Levenberg control in here -- a landmark's thread knows its new position, the free keyframes' new poses formed per
// workgroup in LDS -- to save k_ba_error's launch: the edge loop per landmark thread costs what the launch saves, 1172 against 1147 us
// per ten trials alone and no difference inside the pipeline; not kept.)
__global__ __launch_bounds__(256) void k_ba_step(BaDev d) {
  __shared__ double red[4 * 32], out[1];
  if (d.lm->done) return;
  const double lambda = lm_lambda(d.lm);
  const int cur = d.lm->cur, nxt = 1 - cur;
  double part[1] = {0.0};
  if ((int)blockIdx.x < d.gL) {
    const int h = blockIdx.x * 256 + threadIdx.x;
    if (h < d.nLa) {
      const double* H = d.Hll + (size_t)h * 9;
      double cl[3] = {H[6], H[7], H[8]};
      for (int k = d.pt_start[h]; k < d.pt_start[h + 1]; ++k) {
        const int ph = d.pose_h[d.e_ps[d.act[k]]];
        if (ph < 0) continue;
        const double* B = d.Bk + (size_t)k * 18;
        const double* xp = d.x + 6 * ph;
        for (int c = 0; c < 3; ++c)
          for (int r = 0; r < 6; ++r) cl[c] -= B[r * 3 + c] * xp[r];
      }
      double I[9], dbv[3];
      point_dinv(H, lambda, I, dbv);
      const double xl[3] = {I[0] * cl[0] + I[1] * cl[1] + I[2] * cl[2], I[1] * cl[0] + I[4] * cl[1] + I[5] * cl[2],
                            I[2] * cl[0] + I[5] * cl[1] + I[8] * cl[2]};
      const int l = d.pt_of_h[h];
      for (int r = 0; r < 3; ++r) {
        d.x[6 * d.nPf + 3 * h + r] = xl[r];
        d.pts[nxt][3 * l + r] = d.pts[cur][3 * l + r] + xl[r];
        part[0] += xl[r] * (lambda * xl[r] + H[6 + r]);
      }
    }
    block_reduce<1>(part, red, out);
    if (threadIdx.x == 0) d.partial[d.scale_off + blockIdx.x] = out[0];
    return;
  }
  const int pb = (int)blockIdx.x - d.gL;
  const int h = pb * 256 + threadIdx.x;
  if (h < d.nPf) {
    const int p = d.pose_of_h[h];
    double u[6];
    for (int r = 0; r < 6; ++r) {
      u[r] = d.x[6 * h + r];
      part[0] += u[r] * (lambda * u[r] + d.Hpp[(size_t)h * 27 + 21 + r]);
    }
    // (a pose whose edges were all moved to level 1 is no active vertex in g2o: its step is exactly zero here, and exp(0) * T would
    // still re-normalise the quaternion)
    const bool zero = u[0] == 0 && u[1] == 0 && u[2] == 0 && u[3] == 0 && u[4] == 0 && u[5] == 0;
    d.pose[nxt][p] = zero ? d.pose[cur][p] : pose_oplus(d.pose[cur][p], u);
  }
  block_reduce<1>(part, red, out);
  if (threadIdx.x == 0) d.partial[d.scale_off + d.gL + pb] = out[0];
}

// ---- Levenberg control on the device (optimization_algorithm_levenberg.cpp:61-189) ----------------------------------------
// One thread; the sums run over the per-workgroup partials in index order, the order the host loop used.
// x^3 as pow(x, 3) returns it: the product is formed with its rounding errors carried along (two fma residuals) and rounded once
__device__ inline double cube_rn(double x) {
  const double p = x * x, e = fma(x, x, -p);
  const double q = p * x, f = fma(p, x, -q);
  return q + (f + e * x);
}
// the per-workgroup partial sums come into LDS with one coalesced read; thread 0 then adds them in index order
__device__ inline int lm_stage_partials(const BaDev& d, double* sh) {
  const int np = min(d.scale_off + d.gL + d.gP, kLmMaxPartials);
  for (int i = threadIdx.x; i < np; i += kLmThreads) sh[i] = ld_agent(d.partial + i);   // this launch's own sums among them
  asd_syncthreads();
  return np;
}
// start of a round: computeActiveErrors + activeRobustChi2 have just run (k_ba_error, always)
__device__ void lm_begin_body(const BaDev& d, int iterations, double* sh) {
  lm_stage_partials(d, sh);
  if (threadIdx.x != 0) return;
  LmState S;
  double sum = 0;
  for (int i = 0; i < d.gE; ++i) sum += sh[i];
  S.lambda = -1; S.ni = 2; S.currentChi = sum; S.iniChi = sum; S.rho = 0;
  S.maxdiag_bits = 0;
  S.it = 0; S.qmax = 0; S.nBad = 0; S.done = iterations <= 0 ? 1 : 0; S.need_lin = 1; S.cur = d.lm->cur & 1; S.trials = 0; S.iters_done = 0;   // (cur carries over from the previous round)
  S.chol_ok = 1; S.iterations = iterations; S.first = 1; S.pad_ = 0;
  *d.lm = S;
  *d.lm_host = S;
}
// behind every trial (solve, update, computeActiveErrors): accept or reject, next lambda, end of iteration / of the round
__device__ void lm_control_body(const BaDev& d, double* sh) {
  lm_stage_partials(d, sh);
  if (threadIdx.x != 0) return;
  LmState S = *d.lm;
  if (S.first) {   // computeLambdaInit after the first buildSystem (:86-91)
    S.lambda = 1e-5 * __longlong_as_double((long long)S.maxdiag_bits);
    S.ni = 2; S.nBad = 0; S.first = 0;
  }
  S.need_lin = 0;
  double tempChi = 0, scale = 0;
  for (int i = 0; i < d.gE; ++i) tempChi += sh[i];
  if (d.nPf > 0) for (int i = 0; i < d.gP; ++i) scale += sh[d.scale_off + d.gL + i];   // poses first, then landmarks
  for (int i = 0; i < d.gL; ++i) scale += sh[d.scale_off + i];
  const bool ok2 = d.nPf == 0 || S.chol_ok == 1;
  if (S.trials < kLmLogCap) d.lm_log[S.trials] = LmLog{S.lambda, S.currentChi, tempChi, scale};
  if (!ok2) tempChi = 1.7976931348623157e308;
  double rho = S.currentChi - tempChi;
  scale += 1e-3;
  rho /= scale;
  if (rho > 0 && isfinite(tempChi)) {
    double alpha = 1. - cube_rn(2 * rho - 1);
    alpha = fmin(alpha, 2. / 3.);
    S.lambda *= fmax(1. / 3., alpha);
    S.ni = 2;
    S.currentChi = tempChi;
    S.cur ^= 1;      // the trial estimate becomes the accepted one (_optimizer->discardTop())
  } else {
    S.lambda *= S.ni;
    S.ni *= 2;       // (_optimizer->pop(): the accepted estimate was never touched)
  }
  S.rho = rho;
  S.qmax++;
  S.trials++;
  if (!(rho < 0 && S.qmax < 10)) {   // the do-while of :98-146 ends: the iteration is over
    S.iters_done++;
    int stop = (S.qmax == 10 || rho == 0) ? 1 : 0;
    if (!stop) {
      if ((S.iniChi - S.currentChi) * 1e3 < S.iniChi) S.nBad++; else S.nBad = 0;
      if (S.nBad >= 3) stop = 1;
    }
    S.it++;
    if (stop || S.it >= S.iterations) S.done = 1;
    else { S.need_lin = 1; S.iniChi = S.currentChi; S.qmax = 0; S.rho = 0; }
  }
  *d.lm = S;
  *d.lm_host = S;
}

// activeRobustChi2() over the stored errors (no recomputation)
__global__ __launch_bounds__(256) void k_ba_chi2_stored(BaDev d, int robust, double* __restrict__ sums /* pinned host */) {
  __shared__ double red[4 * 32], out[1];
  const int k = blockIdx.x * 256 + threadIdx.x;
  double part[1] = {0.0};
  if (k < d.Ea) {
    const int e = d.act[k];
    if (!d.lvl[e]) {
      const double c = (d.err[2 * e] * d.err[2 * e] + d.err[2 * e + 1] * d.err[2 * e + 1]) * d.info[e];
      double r0 = c, r1;
      if (robust) huber(c, huber_delta(), r0, r1);
      part[0] = r0;
    }
  }
  block_reduce<1>(part, red, out);
  if (threadIdx.x == 0) sums[blockIdx.x] = out[0];
}

// final per-edge outputs: chi2 from the stored error, isDepthPositive from the current estimate
// gate != null: the outlier gating between the rounds as well (Optimizer.cc:612-631): chi2 > 5.991 || !isDepthPositive -> level 1
__global__ __launch_bounds__(256) void k_ba_edge_report(BaDev d, double* chi2, uint8_t* depth_pos, uint8_t* gate_lvl, uint8_t* gate_out) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= d.E) return;
  const int cur = d.lm->cur;
  const double c = (d.err[2 * e] * d.err[2 * e] + d.err[2 * e + 1] * d.err[2 * e + 1]) * d.info[e];
  chi2[e] = c;
  double Xc[3];
  pose_map(d.pose[cur][d.e_ps[e]], d.pts[cur] + 3 * d.e_pt[e], Xc);
  const uint8_t dp = Xc[2] > 0.0 ? 1 : 0;
  depth_pos[e] = dp;
  if (gate_lvl) {
    const uint8_t bad = (c > 5.991 || !dp) ? 1 : 0;
    gate_lvl[e] = bad;
    gate_out[e] = bad;
  }
}

// ---------------------------------------------------------------- host state
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  int ensure(asd_ctx* ctx, size_t bytes) {
    if (bytes <= cap) return ASD_OK;
    if (p) (void)hipFree(p);
    p = nullptr; cap = 0;
    const size_t want = bytes + bytes / 4 + 256;
    ASD_HIP_CHECK(ctx, hipMalloc(&p, want));
    cap = want;
    return ASD_OK;
  }
  template <typename T> T* as() { return reinterpret_cast<T*>(p); }
};


// ---------------------------------------------------------------- the active structure, built on the device
// g2o's initializeOptimization + buildStructure (sparse_optimizer.cpp:206-267, 166-190; block_solver.hpp:117-300) as this solver
// keeps it: h-indices of the free poses / active landmarks, the edges grouped by landmark and ordered by pose inside a landmark,
// every free pose's edge list, and per upper block (i <= j) of the reduced system the list of edge pairs (a, b) that meet in it.
// The host loop that made these tables (kept below: ASD_BA_STRUCT=host, and for more than 32 free poses) took 0.35-0.47 ms of the
// 2.5 ms LocalBA with the device idle; the same tables come out of four small launches.  Every list is in the order the host
// loop produces (landmark, then position inside the landmark), so the sums the solver forms over them are the same sums.
struct BaStructDev {
  int P, L, E;
  const int* e_ps; const int* e_pt; const uint8_t* fixed;
  int* pose_h; int* pt_h; int* pose_of_h; int* pt_of_h; int* pt_start; int* act;
  int* ph_of_k;        // [E] pose h-index of the k-th edge in act order (-1: fixed pose)
  int* cursor;         // [L + 1] scratch
  unsigned* pt_mask;   // [L] bit i: the landmark has an edge to free pose i
  int* pt_free0;       // [L] act position of the landmark's first edge to a free pose
  int* ps_cnt;         // [32]
  int* ps_start; int* ps_edges;
  int* h_of_k;         // [E] landmark h-index of the k-th edge in act order
  int* ps_h;           // [E] landmark of every entry of ps_edges
  int* blk_i; int* blk_j; int* pair_cnt; int* pair_start; int2* pairs;
  int* pair_h;         // landmark of every pair
  int* counts;         // pinned host [4]: nPf, nLa, Ea, pairs
};
constexpr int kStructThreads = 1024, kStructMaxP = 1024, kStructMaxFree = 32;

__device__ inline int struct_excl_scan(int v, int* sh, int& total) {   // exclusive prefix of v over the 1024 threads; sh: [16]
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  int x = v;
  for (int off = 1; off < 64; off <<= 1) { const int y = __shfl_up(x, off); if (lane >= off) x += y; }
  if (lane == 63) sh[w] = x;
  asd_syncthreads();
  if (w == 0) {
    int q = lane < kStructThreads / 64 ? sh[lane] : 0;
    for (int off = 1; off < 16; off <<= 1) { const int y = __shfl_up(q, off); if (lane >= off) q += y; }
    if (lane < kStructThreads / 64) sh[lane] = q;
  }
  asd_syncthreads();
  const int base = w ? sh[w - 1] : 0;
  total = sh[kStructThreads / 64 - 1];
  asd_syncthreads();
  return base + x - v;
}

// The vertex tables and the landmark CSR in six small launches (edge-parallel where the work is per edge, one workgroup for the two
// prefix sums); atomics only where the order does not matter -- flags, counts, and a scatter whose segments are sorted afterwards.
// A single workgroup walking the 29 k edges seven times took 0.3 ms: three dependent L2 round trips per edge and pass.
__global__ __launch_bounds__(256) void k_ba_struct_flags(BaStructDev a, int* pflag) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e < a.E) { pflag[a.e_ps[e]] = 1; a.pt_h[a.e_pt[e]] = 1; }
}
// free poses in pose order, active landmarks in landmark order; the landmark counters cleared
__global__ __launch_bounds__(kStructThreads) void k_ba_struct_vertices(BaStructDev a, const int* pflag) {
  __shared__ int sh[16];
  constexpr int NT = kStructThreads;
  const int t = threadIdx.x;
  if (t == 0) {
    int h = 0;
    for (int p = 0; p < a.P; ++p) {
      int v = -1;
      if (pflag[p] && !a.fixed[p]) { v = h; a.pose_of_h[h++] = p; }
      a.pose_h[p] = v;
    }
    a.counts[0] = h;
  }
  if (t < kStructMaxFree) a.ps_cnt[t] = 0;
  const int chunk = (a.L + NT - 1) / NT, l0 = min(t * chunk, a.L), l1 = min(l0 + chunk, a.L);
  int c = 0;
  for (int l = l0; l < l1; ++l) c += a.pt_h[l];
  int nLa;
  int base = struct_excl_scan(c, sh, nLa);
  for (int l = l0; l < l1; ++l) {
    if (a.pt_h[l]) { a.pt_h[l] = base; a.pt_of_h[base] = l; ++base; } else a.pt_h[l] = -1;
  }
  for (int h = t; h <= nLa; h += NT) a.pt_start[h] = 0;
  if (t == 0) { a.counts[1] = nLa; a.cursor[a.L] = nLa; }   // cursor[L]: nLa for the kernels behind this one
}
__global__ __launch_bounds__(256) void k_ba_struct_count(BaStructDev a) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e < a.E) atomicAdd(&a.pt_start[a.pt_h[a.e_pt[e]] + 1], 1);
}
// counts -> offsets, in place: slot h + 1 is read and written by the thread that owns landmark h
__global__ __launch_bounds__(kStructThreads) void k_ba_struct_offsets(BaStructDev a) {
  __shared__ int sh[16];
  constexpr int NT = kStructThreads;
  const int t = threadIdx.x, nLa = a.cursor[a.L];
  asd_syncthreads();
  const int hchunk = (nLa + NT - 1) / NT, h0 = min(t * hchunk, nLa), h1 = min(h0 + hchunk, nLa);
  int cs = 0;
  for (int h = h0; h < h1; ++h) cs += a.pt_start[h + 1];
  int Ea;
  int run = struct_excl_scan(cs, sh, Ea);
  for (int h = h0; h < h1; ++h) { const int n = a.pt_start[h + 1]; a.cursor[h] = run; run += n; a.pt_start[h + 1] = run; }
  if (t == 0) a.counts[2] = Ea;
}
__global__ __launch_bounds__(256) void k_ba_struct_scatter(BaStructDev a) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e < a.E) { const int k = atomicAdd(&a.cursor[a.pt_h[a.e_pt[e]]], 1); a.act[k] = e; }
}
// inside a landmark: by pose h-index (fixed poses, -1, first), equal poses in edge order = the host's stable insertion sort;
// then the pose index of every position, the landmark's mask of free poses, where its free edges begin, the poses' edge counts
__global__ __launch_bounds__(256) void k_ba_struct_sort(BaStructDev a) {
  __shared__ int pcnt[kStructMaxFree];
  const int t = threadIdx.x, h = blockIdx.x * 256 + t, nLa = a.cursor[a.L];
  if (t < kStructMaxFree) pcnt[t] = 0;
  asd_syncthreads();
  if (h < nLa) {
    const int s0 = a.pt_start[h], s1 = a.pt_start[h + 1];
    for (int i = s0 + 1; i < s1; ++i) {
      const int e = a.act[i], pe = a.pose_h[a.e_ps[e]];
      int b = i - 1;
      for (; b >= s0; --b) {
        const int eb = a.act[b], pb = a.pose_h[a.e_ps[eb]];
        if (pb < pe || (pb == pe && eb < e)) break;
        a.act[b + 1] = eb;
      }
      a.act[b + 1] = e;
    }
    unsigned mask = 0;
    int free0 = s1;
    for (int i = s0; i < s1; ++i) {
      const int pv = a.pose_h[a.e_ps[a.act[i]]];
      a.ph_of_k[i] = pv;
      a.h_of_k[i] = h;
      if (pv >= 0) { free0 = min(free0, i); if (pv < kStructMaxFree) { mask |= 1u << pv; atomicAdd(&pcnt[pv], 1); } }
    }
    a.pt_mask[h] = mask;
    a.pt_free0[h] = free0;
  }
  asd_syncthreads();
  if (t < kStructMaxFree && pcnt[t]) atomicAdd(&a.ps_cnt[t], pcnt[t]);
}

// every free pose's edges (k indices) in k order: one workgroup per pose, an ordered compaction over the act list
__global__ __launch_bounds__(kStructThreads) void k_ba_struct_pose_edges(BaStructDev a, int Ea) {
  constexpr int NT = kStructThreads, NW = NT / 64;
  __shared__ int wsum[NW];
  const int h = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
  int start = 0;
  for (int q = 0; q < h; ++q) start += a.ps_cnt[q];
  if (t == 0) { a.ps_start[h] = start; if (h == (int)gridDim.x - 1) a.ps_start[h + 1] = start + a.ps_cnt[h]; }
  int run = start;
  for (int k0 = 0; k0 < Ea; k0 += NT) {
    const int k = k0 + t;
    const bool f = k < Ea && a.ph_of_k[k] == h;
    const unsigned long long bal = __ballot(f);
    if (lane == 0) wsum[w] = __popcll(bal);
    asd_syncthreads();
    int off = __popcll(bal & ((1ull << lane) - 1)), all = 0;
    for (int q = 0; q < NW; ++q) { if (q < w) off += wsum[q]; all += wsum[q]; }
    if (f) { a.ps_edges[run + off] = k; a.ps_h[run + off] = a.h_of_k[k]; }
    run += all;
    asd_syncthreads();
  }
}

// the pair list of upper block q = (i <= j): landmarks in order, inside a landmark the host's nested loop (a, then b >= a).
// FILL = false counts, FILL = true writes (its start = the counts of the blocks before it).  A landmark whose free edges go to
// distinct poses (every real one: an observation per keyframe) finds its pair by two popcounts of its mask; one with repeated poses
// walks its edge list like the host loop.
constexpr int kPairThreads = 1024;
template <bool FILL>
__global__ __launch_bounds__(kPairThreads) void k_ba_struct_pairs(BaStructDev a, int nPf, int nLa, int nblk) {
  constexpr int NW = kPairThreads / 64;
  __shared__ int wsum[NW], red[kPairThreads];
  const int q = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
  int i = 0, rem = q;
  while (rem >= nPf - i) { rem -= nPf - i; ++i; }
  const int j = i + rem;
  int start = 0;
  if (FILL) {
    int part = 0;
    for (int b = t; b < q; b += kPairThreads) part += a.pair_cnt[b];
    red[t] = part;
    asd_syncthreads();
    for (int off = kPairThreads / 2; off >= 1; off >>= 1) { if (t < off) red[t] += red[t + off]; asd_syncthreads(); }
    start = red[0];
    asd_syncthreads();
    if (t == 0) { a.blk_i[q] = i; a.blk_j[q] = j; a.pair_start[q] = start; }
  }
  const unsigned bi = 1u << i, bj = 1u << j;
  int run = start, total = 0;
  for (int h0 = 0; h0 < nLa; h0 += kPairThreads) {
    const int h = h0 + t;
    int cnt = 0, f0 = 0, s1 = 0;
    unsigned m = 0;
    bool simple = true;
    if (h < nLa) {
      m = a.pt_mask[h];
      if ((m & bi) && (m & bj)) {
        f0 = a.pt_free0[h]; s1 = a.pt_start[h + 1];
        simple = __popc(m) == s1 - f0;
        if (simple) cnt = 1;
        else
          for (int x = f0; x < s1; ++x)
            if (a.ph_of_k[x] == i)
              for (int y = x; y < s1; ++y) cnt += a.ph_of_k[y] == j;
      }
    }
    int xs = cnt;   // exclusive prefix of cnt over the workgroup
    for (int off = 1; off < 64; off <<= 1) { const int y = __shfl_up(xs, off); if (lane >= off) xs += y; }
    if (lane == 63) wsum[w] = xs;
    asd_syncthreads();
    int off = xs - cnt, all = 0;
    for (int b = 0; b < NW; ++b) { if (b < w) off += wsum[b]; all += wsum[b]; }
    if (FILL && cnt) {
      int o = run + off;
      if (simple) { a.pairs[o] = make_int2(f0 + __popc(m & (bi - 1)), f0 + __popc(m & (bj - 1))); a.pair_h[o] = h; }
      else
        for (int x = f0; x < s1; ++x)
          if (a.ph_of_k[x] == i)
            for (int y = x; y < s1; ++y)
              if (a.ph_of_k[y] == j) { a.pair_h[o] = h; a.pairs[o++] = make_int2(x, y); }
    }
    run += all; total += all;
    asd_syncthreads();
  }
  if (t == 0) {
    if (!FILL) a.pair_cnt[q] = total;
    else if (q == nblk - 1) { a.pair_start[nblk] = start + total; a.counts[3] = start + total; }
  }
}

struct BaState {
  DevBuf lvl, HppPart, fixed_d;
  int* h_counts = nullptr;      // pinned [4]: nPf, nLa, Ea, pairs of the structure built on the device
  hipEvent_t ev_counts = nullptr;   // behind k_ba_struct_offsets: nPf, nLa, Ea are on the host while the scatter and the sort still run
  DevBuf out1, Apack, sblk;   // sblk: the round's structure arrays in one block (uploaded from the pinned h_sblk)
  char* h_sblk = nullptr;
  size_t h_sblk_cap = 0;
  DevBuf pose, pose_bak, pts, pts_bak, e_pt, e_ps, obs, info, err, act, pose_h, pt_h, pose_of_h, pt_of_h, pt_start,
      ps_start, ps_edges, Bk, Hc, Hl, Yk, ck, Hpp, Hll, Dinv, db, x, A, bs, misc, partial, blk_i, blk_j, pair_start,
      pairs, chi2, dpos;
  DevBuf po_Xw, po_obs, po_info, po_err, po_level, po_outlier, po_pose;
  DevBuf pc_n;   // fused chains: edge count made on the device
  double* h_partial = nullptr;  // pinned
  size_t h_partial_cap = 0;
  LmState* h_lm = nullptr;      // pinned mirror of the device's Levenberg state
  DevBuf lm;                    // device: LmState + the per-trial log
  int lm_blocks[2] = {0, 0};    // trials the previous LocalBA's rounds took: length of the first chunk of blocks enqueued ahead
  char* h_po = nullptr;         // pinned staging of asd_pose_optimize
  size_t h_po_cap = 0;
  struct BaLane* lane = nullptr;   // asd_local_ba_submit / _wait: the optional LocalBA lane (own thread, stream and events)
};

// OPTIONAL lane, not the reference's order.  This fork of ORB-SLAM2 has no mapping thread: Tracking::CreateNewKeyFrame calls
// LocalMapping::DoMapping() in line (Tracking.cc:797 -> LocalMapping.cc:59-113, Optimizer::LocalBundleAdjustment at :89;
// LocalMapping::Run at :120 is dead residue, System.cc starts no thread), so LocalBA has finished before the next frame is tracked --
// that is asd_local_ba.  The lane (asd_local_ba_submit / _wait) runs the same solver as one job at a time on a stream of its
// own, beside the caller's next frames; those frames then read the map as it was BEFORE this LocalBA, which is a different data
// dependency from the reference (upstream ORB-SLAM2's threaded arrangement), offered for integrators who want it.
struct BaLane {
  std::thread th;
  std::mutex m;
  std::condition_variable cv;
  asd_ba_problem* pr = nullptr;
  asd_ba_result* res = nullptr;
  bool has_job = false, busy = false, done = false, quit = false;
  int rc = ASD_OK;
  float ms = 0.f;
  hipStream_t st = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
};

BaState* ba_state(asd_ctx* ctx) {
  if (!ctx->ba) ctx->ba = new BaState();
  return static_cast<BaState*>(ctx->ba);
}

}  // namespace

void ba_free(asd_ctx* ctx) {
  if (!ctx->ba) return;
  BaState* s = static_cast<BaState*>(ctx->ba);
  if (BaLane* ln = s->lane) {   // an outstanding job is allowed to finish (its buffers belong to the caller until then)
    {
      std::lock_guard<std::mutex> l(ln->m);
      ln->quit = true;
    }
    ln->cv.notify_all();
    if (ln->th.joinable()) ln->th.join();
    if (ln->st) { (void)hipStreamDestroy(ln->st); }
    if (ln->e0) (void)hipEventDestroy(ln->e0);
    if (ln->e1) (void)hipEventDestroy(ln->e1);
    delete ln;
    s->lane = nullptr;
  }
  DevBuf* all[] = {&s->pose, &s->pose_bak, &s->pts, &s->pts_bak, &s->e_pt, &s->e_ps, &s->obs, &s->info, &s->err, &s->act,
                   &s->pose_h, &s->pt_h, &s->pose_of_h, &s->pt_of_h, &s->pt_start, &s->ps_start, &s->ps_edges, &s->Bk,
                   &s->Hc, &s->Hl, &s->Yk, &s->ck, &s->Hpp, &s->Hll, &s->Dinv, &s->db, &s->x, &s->A, &s->bs, &s->misc, &s->lm, &s->lvl, &s->HppPart, &s->out1, &s->Apack, &s->sblk,
                   &s->partial, &s->blk_i, &s->blk_j, &s->pair_start, &s->pairs, &s->chi2, &s->dpos, &s->po_Xw,
                   &s->po_obs, &s->po_info, &s->po_err, &s->po_level, &s->po_outlier, &s->po_pose, &s->pc_n};
  for (DevBuf* b : all) if (b->p) (void)hipFree(b->p);
  if (s->h_partial) (void)hipHostFree(s->h_partial);
  if (s->h_lm) (void)hipHostFree(s->h_lm);
  if (s->h_counts) (void)hipHostFree(s->h_counts);
  if (s->ev_counts) (void)hipEventDestroy(s->ev_counts);
  if (s->h_sblk) (void)hipHostFree(s->h_sblk);
  if (s->h_po) (void)hipHostFree(s->h_po);
  delete s;
  ctx->ba = nullptr;
}

// Fused tracking chains (asd_track_motion_model / asd_track_local_map, matcher.hip): PoseOptimization enqueued on the context's
// stream directly behind the kernels that made the matches -- k_pose_edges builds the edge records from the device-resident
// match table, k_pose_opt reads their count from the device -- so the chain needs ONE synchronisation, at its end.  The results
// (pose, n_bad, outlier byte per edge in keypoint order) are copied to *h_io; the caller synchronises and unpacks them.
// Everything pose_chain_enqueue may have to ask the runtime for -- scratch buffers, the kernel's dynamic-LDS attribute -- done ahead of
// time (asd_track_frame calls it before its first launch).
int pose_chain_reserve(asd_ctx* ctx, int n_cur) {
  BaState* s = ba_state(ctx);
  int rc;
  const size_t idx_off = (size_t)n_cur * 48;
  if ((rc = s->po_Xw.ensure(ctx, idx_off + ((size_t)n_cur + 63) / 64 * 64 + 64)) || (rc = s->po_err.ensure(ctx, (size_t)n_cur * 64)) ||
      (rc = s->po_level.ensure(ctx, (size_t)2 * n_cur)) || (rc = s->pc_n.ensure(ctx, 16)))
    return rc;
  static AsdPerDeviceOnce attr_set;
  if (attr_set.need(ctx->cfg.device)) {
    ASD_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_pose_opt<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    const void* ks[] = {reinterpret_cast<const void*>(k_resolve_pose<0, 4>), reinterpret_cast<const void*>(k_resolve_pose<0, 8>),
                        reinterpret_cast<const void*>(k_resolve_pose<1, 4>), reinterpret_cast<const void*>(k_resolve_pose<1, 8>)};
    for (const void* k : ks) ASD_HIP_CHECK(ctx, hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    attr_set.done(ctx->cfg.device);
  }
  return ASD_OK;
}

bool pose_chain_fused_ok(const asd_ctx* ctx, int kind, int nq, int n_cur, size_t lds) {
  // kind 0: every last-frame point is a slot of the replay (8 per thread); kind 1: the map points with candidates are replayed in chunks
  // (resolve2.h), what bounds nq is the compaction's round count
  const int max_q = kind == 0 ? 8 * kPoseThreads : kResolve2MaxRounds * kPoseThreads;
  return (kind == 0 || kind == 1) && nq >= 1 && nq <= max_q && nq < 65536 && pose_chain_lds_form(ctx, n_cur) && lds <= 150 * 1024;
}

// the solver's argument block for device-resident matches (gather form, mode 2)
static PoseOptArgs pose_chain_args(asd_ctx* ctx, BaState* s, int n_cur, const int* d_src, const float4* d_kp, const float* d_tab, const uint8_t* d_hold,
                                   const float* d_own, const double* pose7, const double* K, double* d_io, const double* d_pose0, double* d_io_dev,
                                   const AsdBetweenArgs* between) {
  PoseOptArgs a{};
  a.n = n_cur;
  a.g_src = d_src; a.g_hold = d_hold; a.g_tab = d_tab; a.g_own = d_own; a.g_kp = d_kp; a.g_ncur = n_cur;
  for (int k = 0; k < 16; ++k) a.isg_tab[k] = k < ctx->cfg.n_levels ? (double)ctx->inv_sigma2[k] : 0.0;   // invSigma2 is a float in the reference (Optimizer.cc:300)
  if (pose7) memcpy(a.pose0, pose7, 56);
  a.pose0_dev = d_pose0; a.io_dev = d_io_dev;
  a.between = between;
  a.fx = K[0]; a.fy = K[1]; a.cx = K[2]; a.cy = K[3];
  a.soa_g = s->po_err.as<double>(); a.flags_g = s->po_level.as<uint8_t>(); a.io = d_io;
  a.use_lds = 2;
  a.debug = 0;
  return a;
}

int pose_chain_enqueue(asd_ctx* ctx, int n_cur, const int* d_src, const float4* d_kp, const float* d_tab, const uint8_t* d_hold,
                       const float* d_own, const double* pose7, const double* K, double* d_io, const double* d_pose0, double* d_io_dev,
                       const AsdBetweenArgs* between, const AsdFusedReplay* fused) {
  // every input is already on the device (the caller packed the tables into its one upload block), the results go to d_io
  // inside the caller's one result block: no copy is enqueued here
  BaState* s = ba_state(ctx);
  hipStream_t st = ctx->stream;
  int rc;
  if ((rc = pose_chain_reserve(ctx, n_cur)) != ASD_OK) return rc;
  const size_t idx_off = (size_t)n_cur * 48;
  const size_t lds_compact = (size_t)n_cur * 35 + 16;
  const int mode = pose_chain_lds_form(ctx, n_cur) ? 2 : 0;
  PoseOptArgs a = pose_chain_args(ctx, s, n_cur, d_src, d_kp, d_tab, d_hold, d_own, pose7, K, d_io, d_pose0, d_io_dev, between);
  a.use_lds = mode;
  if (mode != 2) {   // larger than LDS: edge records through HBM
    a.g_src = nullptr; a.g_hold = nullptr; a.g_tab = nullptr; a.g_own = nullptr; a.g_kp = nullptr; a.g_ncur = 0;
    PoseEdgesArgs e{};
    e.n_cur = n_cur; e.src = d_src; e.hold = d_hold; e.tab = d_tab; e.own = d_own; e.kp = d_kp;
    for (int l = 0; l < ASD_MAX_LEVELS; ++l) e.inv_sigma2[l] = l < ctx->cfg.n_levels ? ctx->inv_sigma2[l] : 0.f;
    e.edges = s->po_Xw.as<double>(); e.isgi = s->po_Xw.as<uint8_t>() + idx_off; e.n_out = s->pc_n.as<int>();
    hipLaunchKernelGGL(k_pose_edges, dim3(1), dim3(1024), 0, st, e);
    ASD_HIP_CHECK(ctx, hipGetLastError());
    a.n_dev = s->pc_n.as<int>();
    a.edges = s->po_Xw.as<double>(); a.isgi = s->po_Xw.as<uint8_t>() + idx_off;
  }
  if ((d_pose0 || d_io_dev) && mode != 2) { ctx->set_error("pose chain: the device-side hand-over needs the LDS form of the solver (frame too large)"); return ASD_ERR_CAPACITY; }
  if (fused) {
    if (mode != 2 || !pose_chain_fused_ok(ctx, fused->kind, fused->nq, n_cur, fused->lds)) { ctx->set_error("pose chain: no fused replay + solver form for this frame"); return ASD_ERR_CAPACITY; }
    const Resolve2Args& r = *static_cast<const Resolve2Args*>(fused->args);
    const size_t lds = std::max(lds_compact, fused->lds);
    const int qpt = fused->nq <= 4 * kPoseThreads ? 4 : 8;
    if (fused->kind == 0) {
      if (qpt == 4) hipLaunchKernelGGL((k_resolve_pose<0, 4>), dim3(1), dim3(kPoseThreads), lds, st, r, a);
      else hipLaunchKernelGGL((k_resolve_pose<0, 8>), dim3(1), dim3(kPoseThreads), lds, st, r, a);
    } else {
      if (qpt == 4) hipLaunchKernelGGL((k_resolve_pose<1, 4>), dim3(1), dim3(kPoseThreads), lds, st, r, a);
      else hipLaunchKernelGGL((k_resolve_pose<1, 8>), dim3(1), dim3(kPoseThreads), lds, st, r, a);
    }
  } else if (mode == 2) hipLaunchKernelGGL(k_pose_opt<2>, dim3(1), dim3(kPoseThreads), lds_compact, st, a);
  else hipLaunchKernelGGL(k_pose_opt<0>, dim3(1), dim3(kPoseThreads), 0, st, a);
  ASD_HIP_CHECK(ctx, hipGetLastError());
  ctx->pose_chain_kp_flags = mode == 2;   // the form of the flags in d_io: per keypoint + edge count (gather form), or per edge
  return ASD_OK;
}

extern "C" {

int asd_pose_optimize(asd_ctx* ctx, double* pose7, int32_t n, const double* Xw, const double* obs,
                      const double* inv_sigma2, const double* K, uint8_t* outlier, int32_t* n_inliers) {
  if (ctx && asd_track_busy(ctx, "asd_pose_optimize")) return ASD_ERR_INVALID;
  if (!ctx || !pose7 || n < 0 || !K || !n_inliers || (n > 0 && (!Xw || !obs || !inv_sigma2 || !outlier))) return ASD_ERR_INVALID;
  for (int i = 0; i < n; ++i) outlier[i] = 0;
  if (n < 3) { *n_inliers = 0; return ASD_OK; }  // Optimizer.cc:323-324
  (void)hipSetDevice(ctx->cfg.device);
  BaState* s = ba_state(ctx);
  int rc;
  const size_t idx_off = (size_t)n * 48, in_bytes = idx_off + ((size_t)n + 63) / 64 * 64 + 64, io_bytes = 64 + (size_t)n + 64;
  if ((rc = s->po_Xw.ensure(ctx, in_bytes)) || (rc = s->po_err.ensure(ctx, (size_t)n * 64)) ||
      (rc = s->po_level.ensure(ctx, (size_t)2 * n)) || (rc = s->po_pose.ensure(ctx, io_bytes)))
    return rc;
  if (s->h_po_cap < in_bytes + io_bytes) {
    if (s->h_po) (void)hipHostFree(s->h_po);
    s->h_po_cap = 0;
    ASD_HIP_CHECK(ctx, hipHostMalloc(&s->h_po, 2 * (in_bytes + io_bytes)));
    s->h_po_cap = 2 * (in_bytes + io_bytes);
  }
  static const bool timing = getenv("ASD_TIMING") != nullptr;
  static double tacc[2][4]; static long tcalls[2];
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto tp0 = now();
  // one pinned staging buffer: [n][6] edges, the information-value index per edge, then the io block
  double* hin = reinterpret_cast<double*>(s->h_po);
  uint8_t* hidx = reinterpret_cast<uint8_t*>(s->h_po + idx_off);
  PoseOptArgs a{};   // n_dev = nullptr: the edge count is the argument n
  int ntab = 0;
  bool compact = true;  // observations exactly f32, <= 16 distinct information values (see EdgeStore)
  for (int i = 0; i < n; ++i) {
    hin[6 * i] = Xw[3 * i]; hin[6 * i + 1] = Xw[3 * i + 1]; hin[6 * i + 2] = Xw[3 * i + 2];
    const double u = obs[2 * i], v = obs[2 * i + 1], w = inv_sigma2[i];
    hin[6 * i + 3] = u; hin[6 * i + 4] = v; hin[6 * i + 5] = w;
    if (compact) {
      if ((double)(float)u != u || (double)(float)v != v) { compact = false; continue; }
      int k = 0;
      while (k < ntab && memcmp(&a.isg_tab[k], &w, 8) != 0) ++k;   // bit pattern: NaNs and signed zeros stay themselves
      if (k == ntab) {
        if (ntab == 16) { compact = false; continue; }
        a.isg_tab[ntab++] = w;
      }
      hidx[i] = (uint8_t)k;
    }
  }
  for (int k = ntab; k < 16; ++k) a.isg_tab[k] = 0.0;
  double* hio = reinterpret_cast<double*>(s->h_po + in_bytes);
  memcpy(hio, pose7, 56);
  hipStream_t st = ctx->stream;
  auto tp1 = now();
  // edges and their information-value indices travel in one copy; the pose is a kernel argument
  ASD_HIP_CHECK(ctx, hipMemcpyAsync(s->po_Xw.p, hin, idx_off + (size_t)n, hipMemcpyHostToDevice, st));
  a.n = n;
  a.edges = s->po_Xw.as<double>();
  a.isgi = s->po_Xw.as<uint8_t>() + idx_off;
  memcpy(a.pose0, pose7, 56);
  a.fx = K[0]; a.fy = K[1]; a.cx = K[2]; a.cy = K[3];
  a.soa_g = s->po_err.as<double>();
  a.flags_g = s->po_level.as<uint8_t>();
  a.io = s->po_pose.as<double>();
  const size_t lds_full = (size_t)n * 48 + (size_t)2 * n + 16, lds_compact = (size_t)n * 35 + 16;
  int mode = 0;
  if (compact && lds_compact <= 150 * 1024) mode = 2;
  else if (lds_full <= 150 * 1024) mode = 1;
  a.use_lds = mode;
  a.debug = getenv("ASD_POSE_DEBUG") ? 1 : 0;
  static AsdPerDeviceOnce attr_set;
  if (attr_set.need(ctx->cfg.device)) {
    ASD_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_pose_opt<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    ASD_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_pose_opt<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    attr_set.done(ctx->cfg.device);
  }
  ASD_HIP_CHECK(ctx, hipEventRecord(ctx->ev0, st));
  if (mode == 2) hipLaunchKernelGGL(k_pose_opt<2>, dim3(1), dim3(kPoseThreads), lds_compact, st, a);
  else if (mode == 1) hipLaunchKernelGGL(k_pose_opt<1>, dim3(1), dim3(kPoseThreads), lds_full, st, a);
  else hipLaunchKernelGGL(k_pose_opt<0>, dim3(1), dim3(kPoseThreads), 0, st, a);
  ASD_HIP_CHECK(ctx, hipGetLastError());
  ASD_HIP_CHECK(ctx, hipEventRecord(ctx->ev1, st));
  ASD_HIP_CHECK(ctx, hipMemcpyAsync(hio, s->po_pose.p, 64 + (size_t)n + 8, hipMemcpyDeviceToHost, st));
  auto tp2 = now();
  ASD_HIP_CHECK(ctx, hipStreamSynchronize(st));
  auto tp3 = now();
  ASD_HIP_CHECK(ctx, hipEventElapsedTime(&ctx->ms_ba, ctx->ev0, ctx->ev1));
  if (timing) {
    const int k = n > 1500;
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    tacc[k][0] += ms(tp0, tp1); tacc[k][1] += ms(tp1, tp2); tacc[k][2] += ms(tp2, tp3); tacc[k][3] += ctx->ms_ba;
    if (++tcalls[k] % 200 == 0)
      fprintf(stderr, "[pose_optimize n%s1500] pack %.3f enqueue %.3f sync %.3f kernel %.3f ms\n", k ? ">" : "<=", tacc[k][0] / tcalls[k], tacc[k][1] / tcalls[k],
              tacc[k][2] / tcalls[k], tacc[k][3] / tcalls[k]);
  }
  memcpy(pose7, hio, 56);
  memcpy(outlier, hio + 8, n);
  *n_inliers = n - (int)(hio[7] + 0.5);
  return ASD_OK;
}

}  // extern "C"

namespace {

int local_ba_check(asd_ctx* ctx, const asd_ba_problem* pr, const asd_ba_result* res) {
  if (!ctx || !pr || !res || pr->n_poses < 1 || pr->n_points < 1 || pr->n_edges < 1 || !pr->poses || !pr->fixed ||
      !pr->points || !pr->e_point || !pr->e_pose || !pr->e_obs || !pr->e_info || !res->edge_chi2 ||
      !res->edge_depth_pos || !res->edge_outlier1)
    return ASD_ERR_INVALID;
  const int P = pr->n_poses, L = pr->n_points, E = pr->n_edges;
  for (int e = 0; e < E; ++e)
    if (pr->e_point[e] < 0 || pr->e_point[e] >= L || pr->e_pose[e] < 0 || pr->e_pose[e] >= P) {
      ctx->set_error("edge %d references vertex out of range", e);
      return ASD_ERR_INVALID;
    }
  return ASD_OK;
}

// the whole LocalBundleAdjustment on stream `st` (the context's stream for asd_local_ba, the lane's for asd_local_ba_submit);
// e0 / e1 bracket the device work, *ms receives its duration
int local_ba_impl(asd_ctx* ctx, BaState* s, asd_ba_problem* pr, asd_ba_result* res, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1, float* ms) {
  const int P = pr->n_poses, L = pr->n_points, E = pr->n_edges;
  (void)hipSetDevice(ctx->cfg.device);
  int rc;
  const auto t_enter = std::chrono::steady_clock::now();
#define ENS(buf, bytes) if ((rc = s->buf.ensure(ctx, (bytes))) != ASD_OK) return rc
  ENS(pose, (size_t)P * sizeof(Pose7)); ENS(pose_bak, (size_t)P * sizeof(Pose7));
  ENS(pts, (size_t)L * 24); ENS(pts_bak, (size_t)L * 24);
  ENS(e_pt, (size_t)E * 4); ENS(e_ps, (size_t)E * 4); ENS(obs, (size_t)E * 16); ENS(info, (size_t)E * 8);
  ENS(err, (size_t)E * 16); ENS(act, (size_t)E * 4); ENS(pose_h, (size_t)P * 4); ENS(pt_h, (size_t)L * 4);
  ENS(pose_of_h, (size_t)P * 4); ENS(pt_of_h, (size_t)L * 4); ENS(pt_start, (size_t)(L + 1) * 4);
  ENS(ps_start, (size_t)(P + 1) * 4); ENS(ps_edges, (size_t)E * 4);
  ENS(Bk, (size_t)E * 18 * 8); ENS(Hc, (size_t)E * 27 * 8); ENS(Hl, (size_t)E * 9 * 8); ENS(Hpp, (size_t)P * 27 * 8); ENS(Hll, (size_t)L * 9 * 8); ENS(Dinv, (size_t)L * 6 * 8);
  ENS(db, (size_t)L * 3 * 8); ENS(x, ((size_t)6 * P + 3 * L) * 8); ENS(A, (size_t)36 * P * P * 8); ENS(Apack, (size_t)18 * P * (P + 1) * 8); ENS(bs, (size_t)6 * P * 8);
  ENS(misc, 64); ENS(chi2, (size_t)E * 8); ENS(dpos, (size_t)E); ENS(lvl, (size_t)E); ENS(out1, (size_t)E); ENS(HppPart, (size_t)P * kPoseSplit * 27 * 8);
  const int nblk_e = (E + 255) / 256, nblk_l = (L + 255) / 256, nblk_p = (P + 255) / 256;
  const size_t npartial = (size_t)nblk_e + nblk_l + nblk_p + 8;
  ENS(partial, npartial * 8);
  if (s->h_partial_cap < npartial) {
    if (s->h_partial) (void)hipHostFree(s->h_partial);
    ASD_HIP_CHECK(ctx, hipHostMalloc(&s->h_partial, npartial * 2 * 8));
    s->h_partial_cap = npartial * 2;
  }
  if (!s->h_lm) ASD_HIP_CHECK(ctx, hipHostMalloc(reinterpret_cast<void**>(&s->h_lm), sizeof(LmState)));
  ENS(lm, sizeof(LmState) + 64 + sizeof(LmLog) * kLmLogCap + sizeof(int) * ((size_t)P + 2));
  if (npartial > (size_t)kLmMaxPartials) { ctx->set_error("asd_local_ba: %d edges exceed the control kernel's %d partial sums", E, kLmMaxPartials); return ASD_ERR_CAPACITY; }

  // upload the problem: poses normalised like SE3Quat's constructor does
  std::vector<Pose7> hp(P);
  for (int p = 0; p < P; ++p) {
    const double* q = pr->poses + 7 * p;
    hp[p] = Pose7{q[0], q[1], q[2], q[3], q[4], q[5], q[6]};
    quat_normalize(hp[p].qx, hp[p].qy, hp[p].qz, hp[p].qw);
  }
  // both estimate buffers start as the input (vertices no trial writes -- fixed poses, landmarks without an active edge -- must read the same in both)
  ASD_HIP_CHECK(ctx, hipMemcpyAsync(s->pose.p, hp.data(), (size_t)P * sizeof(Pose7), hipMemcpyHostToDevice, st));
  ASD_HIP_CHECK(ctx, hipMemcpyAsync(s->pts.p, pr->points, (size_t)L * 24, hipMemcpyHostToDevice, st));
  ASD_HIP_CHECK(ctx, hipMemcpyAsync(s->pose_bak.p, s->pose.p, (size_t)P * sizeof(Pose7), hipMemcpyDeviceToDevice, st));
  ASD_HIP_CHECK(ctx, hipMemcpyAsync(s->pts_bak.p, s->pts.p, (size_t)L * 24, hipMemcpyDeviceToDevice, st));
  ASD_HIP_CHECK(ctx, hipMemsetAsync(s->lm.p, 0, sizeof(LmState), st));   // lm->cur = 0
  ASD_HIP_CHECK(ctx, hipMemcpyAsync(s->e_pt.p, pr->e_point, (size_t)E * 4, hipMemcpyHostToDevice, st));
  ASD_HIP_CHECK(ctx, hipMemcpyAsync(s->e_ps.p, pr->e_pose, (size_t)E * 4, hipMemcpyHostToDevice, st));
  ASD_HIP_CHECK(ctx, hipMemcpyAsync(s->obs.p, pr->e_obs, (size_t)E * 16, hipMemcpyHostToDevice, st));
  ASD_HIP_CHECK(ctx, hipMemcpyAsync(s->info.p, pr->e_info, (size_t)E * 8, hipMemcpyHostToDevice, st));
  ASD_HIP_CHECK(ctx, hipMemsetAsync(s->err.p, 0, (size_t)E * 16, st));
  ASD_HIP_CHECK(ctx, hipMemsetAsync(s->lvl.p, 0, (size_t)E, st));
  ENS(fixed_d, (size_t)std::max(P, 1));
  ASD_HIP_CHECK(ctx, hipMemcpyAsync(s->fixed_d.p, pr->fixed, (size_t)P, hipMemcpyHostToDevice, st));
  if (!s->h_counts) ASD_HIP_CHECK(ctx, hipHostMalloc(reinterpret_cast<void**>(&s->h_counts), 64));
  if (!s->ev_counts) ASD_HIP_CHECK(ctx, hipEventCreateWithFlags(&s->ev_counts, hipEventDisableTiming));
  ASD_HIP_CHECK(ctx, hipEventRecord(ev0, st));

  BaDev d{};
  d.P = P; d.L = L; d.E = E;
  d.pose[0] = s->pose.as<Pose7>(); d.pose[1] = s->pose_bak.as<Pose7>();
  d.pts[0] = s->pts.as<double>(); d.pts[1] = s->pts_bak.as<double>();
  d.e_pt = s->e_pt.as<int>(); d.e_ps = s->e_ps.as<int>(); d.obs = s->obs.as<double>(); d.info = s->info.as<double>();
  d.err = s->err.as<double>(); d.lvl = s->lvl.as<uint8_t>(); d.HppPart = s->HppPart.as<double>();
  d.fx = pr->K[0]; d.fy = pr->K[1]; d.cx = pr->K[2]; d.cy = pr->K[3];
  d.act = s->act.as<int>(); d.pose_h = s->pose_h.as<int>(); d.pt_h = s->pt_h.as<int>();
  d.pose_of_h = s->pose_of_h.as<int>(); d.pt_of_h = s->pt_of_h.as<int>(); d.pt_start = s->pt_start.as<int>();
  d.ps_start = s->ps_start.as<int>(); d.ps_edges = s->ps_edges.as<int>();
  d.Bk = s->Bk.as<double>(); d.Hc = s->Hc.as<double>(); d.Hl = s->Hl.as<double>(); d.Hpp = s->Hpp.as<double>(); d.Hll = s->Hll.as<double>(); d.Dinv = s->Dinv.as<double>();
  d.db = s->db.as<double>(); d.x = s->x.as<double>(); d.A = s->A.as<double>(); d.Apack = s->Apack.as<double>(); d.bs = s->bs.as<double>();
  // status and the per-workgroup partial sums are read by the host after every trial: the kernels store them straight into
  // pinned host memory (a few hundred doubles), which removes two copy commands per trial from the lane's queue
  // the Levenberg scalars and the per-workgroup partial sums they are made of stay on the device (k_ba_lm_control); the host sees
  // a pinned mirror of the state, written after every trial, and reads it once per chunk of blocks
  d.lm = s->lm.as<LmState>();
  d.lm_host = s->h_lm;
  d.lm_log = reinterpret_cast<LmLog*>(s->lm.as<char>() + sizeof(LmState) + 64 - (sizeof(LmState) % 8));
  d.partial = s->partial.as<double>();
  d.tickets = reinterpret_cast<int*>(reinterpret_cast<char*>(d.lm_log) + sizeof(LmLog) * kLmLogCap);
  ASD_HIP_CHECK(ctx, hipMemsetAsync(d.tickets, 0, sizeof(int) * ((size_t)P + 2), st));

  const auto t_uploaded = std::chrono::steady_clock::now();
  std::vector<uint8_t> level(E, 0);
  std::vector<int> act, pose_h(P), pt_h(L), pose_of_h, pt_of_h, pt_start, ps_start, ps_edges, ps_hv, blk_i, blk_j, pair_start, ph_of_k, cursor,
      row_off;
  std::vector<int2> pairs;

  // one g2o initializeOptimization(level 0) + optimize(iterations) round
  const bool timing = getenv("ASD_TIMING") != nullptr;
  // The active structure (sparse_optimizer.cpp:206-267, 166-190) is built ONCE, for the first round (every edge at level 0).  The
  // second round (Optimizer.cc:647: initializeOptimization(0) after the outlier gating) runs over the same structure with the
  // level-1 edges masked: such an edge keeps its place and contributes exact zeros (k_ba_linearize, k_ba_error), a landmark that
  // lost all its edges gets a zero step, and x + 0.0 == x, so every sum equals the sum over the compacted lists g2o would build --
  // without the second host pass over the edges and its uploads (0.45 ms per LocalBA).
  int nPf = 0, nLa = 0, Ea = 0, nblk = 0;
  size_t n_pairs_total = 0;
  SchurBlocks sb{};
  std::chrono::steady_clock::time_point t_prep;
  auto run_round = [&](int round_idx, int iterations, bool robust, double* chi_out, int* iters_out) -> int {
    const auto t_round = std::chrono::steady_clock::now();
    int n_trials = 0;
    int r2;
    int n_free_max = 0;
    for (int p = 0; p < P; ++p) n_free_max += !pr->fixed[p];
    static const bool struct_host = getenv("ASD_BA_STRUCT") && !strcmp(getenv("ASD_BA_STRUCT"), "host");
    // a landmark observed twice by one pose (nothing forbids it, the host loop and the general branch of k_ba_struct_pairs handle it)
    // has more pairs than the device tables are sized for -- E (F + 1) / 2 assumes at most one edge per (landmark, free pose): such a
    // problem takes the host-built structure instead of writing past the pair list
    bool dup_edges = false;
    if (round_idx == 0 && !struct_host && E > 0 && P <= kStructMaxP && n_free_max <= kStructMaxFree) {
      const size_t words = ((size_t)P + 63) / 64;
      std::vector<uint64_t> seen((size_t)L * words, 0);
      for (int e = 0; e < E && !dup_edges; ++e) {
        uint64_t& w = seen[(size_t)pr->e_point[e] * words + (size_t)pr->e_pose[e] / 64];
        const uint64_t bit = 1ull << (pr->e_pose[e] % 64);
        dup_edges = (w & bit) != 0;
        w |= bit;
      }
    }
    const bool struct_dev = !struct_host && !dup_edges && E > 0 && P <= kStructMaxP && n_free_max <= kStructMaxFree;
    if (round_idx == 0 && struct_dev) {
      // ---- active structure, on the device (k_ba_struct_*): the tables are carved out of one block sized by upper bounds
      const int nblk_max = n_free_max * (n_free_max + 1) / 2;
      const size_t pairs_cap = (size_t)E * (size_t)(n_free_max + 1) / 2 + 1;
      size_t off = 0;
      auto place = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
      const size_t o_act = place((size_t)E * 4), o_pose_h = place((size_t)P * 4), o_pt_h = place((size_t)L * 4), o_pose_of_h = place((size_t)P * 4),
                   o_pt_of_h = place((size_t)L * 4), o_pt_start = place((size_t)(L + 1) * 4), o_ps_start = place((size_t)(P + 1) * 4),
                   o_ps_edges = place((size_t)E * 4), o_blk_i = place((size_t)std::max(nblk_max, 1) * 4), o_blk_j = place((size_t)std::max(nblk_max, 1) * 4),
                   o_pair_start = place((size_t)(nblk_max + 1) * 4), o_pairs = place(pairs_cap * 8), o_pair_h = place(pairs_cap * 4), o_ph = place((size_t)E * 4),
                   o_hk = place((size_t)E * 4), o_ps_h = place((size_t)E * 4),
                   o_cursor = place((size_t)(L + 1) * 4), o_mask = place((size_t)L * 4), o_pscnt = place(kStructMaxFree * 4),
                   o_paircnt = place((size_t)std::max(std::max(nblk_max, P), 1) * 4), o_free0 = place((size_t)L * 4);
      if ((r2 = s->sblk.ensure(ctx, off)) != ASD_OK) return r2;
      char* db = s->sblk.as<char>();
      BaStructDev sd{};
      sd.P = P; sd.L = L; sd.E = E;
      sd.e_ps = d.e_ps; sd.e_pt = d.e_pt; sd.fixed = s->fixed_d.as<uint8_t>();
      sd.pose_h = reinterpret_cast<int*>(db + o_pose_h); sd.pt_h = reinterpret_cast<int*>(db + o_pt_h);
      sd.pose_of_h = reinterpret_cast<int*>(db + o_pose_of_h); sd.pt_of_h = reinterpret_cast<int*>(db + o_pt_of_h);
      sd.pt_start = reinterpret_cast<int*>(db + o_pt_start); sd.act = reinterpret_cast<int*>(db + o_act);
      sd.ph_of_k = reinterpret_cast<int*>(db + o_ph); sd.cursor = reinterpret_cast<int*>(db + o_cursor);
      sd.pt_free0 = reinterpret_cast<int*>(db + o_free0);
      sd.pt_mask = reinterpret_cast<unsigned*>(db + o_mask); sd.ps_cnt = reinterpret_cast<int*>(db + o_pscnt);
      sd.ps_start = reinterpret_cast<int*>(db + o_ps_start); sd.ps_edges = reinterpret_cast<int*>(db + o_ps_edges);
      sd.blk_i = reinterpret_cast<int*>(db + o_blk_i); sd.blk_j = reinterpret_cast<int*>(db + o_blk_j);
      sd.pair_cnt = reinterpret_cast<int*>(db + o_paircnt); sd.pair_start = reinterpret_cast<int*>(db + o_pair_start);
      sd.pairs = reinterpret_cast<int2*>(db + o_pairs); sd.pair_h = reinterpret_cast<int*>(db + o_pair_h);
      sd.h_of_k = reinterpret_cast<int*>(db + o_hk); sd.ps_h = reinterpret_cast<int*>(db + o_ps_h);
      sd.counts = s->h_counts;
      s->h_counts[3] = 0;
      {
        int* pflag = sd.pair_cnt;   // [P] scratch until the pair kernels run (nblk_max >= P whenever a pose is free; else sized below)
        const int gEs = (E + 255) / 256;
        ASD_HIP_CHECK(ctx, hipMemsetAsync(sd.pt_h, 0, (size_t)L * 4, st));
        ASD_HIP_CHECK(ctx, hipMemsetAsync(pflag, 0, (size_t)P * 4, st));
        hipLaunchKernelGGL(k_ba_struct_flags, dim3(gEs), dim3(256), 0, st, sd, pflag);
        hipLaunchKernelGGL(k_ba_struct_vertices, dim3(1), dim3(kStructThreads), 0, st, sd, pflag);
        hipLaunchKernelGGL(k_ba_struct_count, dim3(gEs), dim3(256), 0, st, sd);
        hipLaunchKernelGGL(k_ba_struct_offsets, dim3(1), dim3(kStructThreads), 0, st, sd);
        ASD_HIP_CHECK(ctx, hipEventRecord(s->ev_counts, st));
        hipLaunchKernelGGL(k_ba_struct_scatter, dim3(gEs), dim3(256), 0, st, sd);
        hipLaunchKernelGGL(k_ba_struct_sort, dim3((L + 255) / 256), dim3(256), 0, st, sd);
      }
      ASD_HIP_CHECK(ctx, hipGetLastError());
      // the launch dimensions of everything that follows (nPf, nLa, Ea) are complete behind k_ba_struct_offsets: the host reads them and
      // enqueues the rest while the scatter and the sort run (a synchronisation behind the sort left the stream empty for 20-50 us)
      ASD_HIP_CHECK(ctx, hipEventSynchronize(s->ev_counts));
      nPf = s->h_counts[0]; nLa = s->h_counts[1]; Ea = s->h_counts[2];
      nblk = nPf * (nPf + 1) / 2;
      if (nPf > 0) {
        hipLaunchKernelGGL(k_ba_struct_pose_edges, dim3(nPf), dim3(kStructThreads), 0, st, sd, Ea);
        hipLaunchKernelGGL(k_ba_struct_pairs<false>, dim3(nblk), dim3(kPairThreads), 0, st, sd, nPf, nLa, nblk);
        hipLaunchKernelGGL(k_ba_struct_pairs<true>, dim3(nblk), dim3(kPairThreads), 0, st, sd, nPf, nLa, nblk);
        ASD_HIP_CHECK(ctx, hipGetLastError());
      }
      d.act = sd.act; d.pose_h = sd.pose_h; d.pt_h = sd.pt_h; d.pose_of_h = sd.pose_of_h; d.pt_of_h = sd.pt_of_h;
      d.pt_start = sd.pt_start; d.ps_start = sd.ps_start; d.ps_edges = sd.ps_edges; d.ps_h = sd.ps_h;
      sb = SchurBlocks{sd.blk_i, sd.blk_j, sd.pair_start, sd.pairs, sd.pair_h};
      d.Ea = Ea; d.nPf = nPf; d.nLa = nLa;
      n_pairs_total = 0;   // (on the device; read from the pinned counts when the round reports)
    } else
    if (round_idx == 0) {
    // ---- active structure, on the host
    std::vector<uint8_t> pa(P, 0), la(L, 0);
    for (int e = 0; e < E; ++e)
      if (!level[e]) { pa[pr->e_pose[e]] = 1; la[pr->e_point[e]] = 1; }
    pose_of_h.clear(); pt_of_h.clear();
    for (int p = 0; p < P; ++p) { pose_h[p] = -1; if (pa[p] && !pr->fixed[p]) { pose_h[p] = (int)pose_of_h.size(); pose_of_h.push_back(p); } }
    for (int l = 0; l < L; ++l) { pt_h[l] = -1; if (la[l]) { pt_h[l] = (int)pt_of_h.size(); pt_of_h.push_back(l); } }
    nPf = (int)pose_of_h.size(); nLa = (int)pt_of_h.size();
    // active edges grouped by landmark (CSR), inside a landmark ordered by pose h-index (fixed poses first)
    pt_start.assign(nLa + 1, 0);
    for (int e = 0; e < E; ++e) if (!level[e]) ++pt_start[pt_h[pr->e_point[e]] + 1];
    for (int h = 0; h < nLa; ++h) pt_start[h + 1] += pt_start[h];
    Ea = pt_start[nLa];
    act.assign(Ea, 0);
    ph_of_k.assign(std::max(Ea, 1), -1);  // pose h-index of the k-th active edge (-1 = fixed pose)
    {
      cursor.assign(pt_start.begin(), pt_start.end() - 1);
      for (int e = 0; e < E; ++e) if (!level[e]) act[cursor[pt_h[pr->e_point[e]]]++] = e;
      for (int h = 0; h < nLa; ++h) {  // stable insertion sort: a landmark has a handful of observations
        const int s0 = pt_start[h], s1 = pt_start[h + 1];
        for (int a = s0; a < s1; ++a) ph_of_k[a] = pose_h[pr->e_pose[act[a]]];
        for (int a = s0 + 1; a < s1; ++a) {
          const int ea = act[a], pa_ = ph_of_k[a];
          int b = a - 1;
          while (b >= s0 && ph_of_k[b] > pa_) { act[b + 1] = act[b]; ph_of_k[b + 1] = ph_of_k[b]; --b; }
          act[b + 1] = ea; ph_of_k[b + 1] = pa_;
        }
      }
    }
    // per free pose: its edges (k indices) in k order
    ps_start.assign(nPf + 1, 0);
    for (int k = 0; k < Ea; ++k) if (ph_of_k[k] >= 0) ++ps_start[ph_of_k[k] + 1];
    for (int h = 0; h < nPf; ++h) ps_start[h + 1] += ps_start[h];
    ps_edges.assign(std::max(ps_start[nPf], 1), 0);
    ps_hv.assign(ps_edges.size(), 0);
    {
      cursor.assign(ps_start.begin(), ps_start.end() - 1);
      for (int h = 0; h < nLa; ++h)
        for (int k = pt_start[h]; k < pt_start[h + 1]; ++k)   // (k ascending over the landmarks in order: the same order as a plain loop over k)
          if (ph_of_k[k] >= 0) { const int o = cursor[ph_of_k[k]]++; ps_edges[o] = k; ps_hv[o] = h; }
    }
    // Schur pair lists per upper block (bi <= bj), blocks numbered row-major over the upper triangle.  Every upper
    // block gets a workgroup, also those no landmark connects: the in-place Cholesky leaves fill-in in A, so blocks
    // without pairs must be rewritten (to zero) on every trial.
    const int nblk_all = nPf * (nPf + 1) / 2;
    row_off.assign(std::max(nPf, 1), 0);
    for (int i = 0; i < nPf; ++i) row_off[i] = i * nPf - i * (i - 1) / 2 - i;  // block id = row_off[i] + j
    pair_start.assign(nblk_all + 1, 0);
    for (int h = 0; h < nLa; ++h) {
      const int s1 = pt_start[h + 1];
      int a = pt_start[h];
      while (a < s1 && ph_of_k[a] < 0) ++a;  // fixed poses sort first
      for (; a < s1; ++a) {
        const int ro = row_off[ph_of_k[a]];
        for (int b = a; b < s1; ++b) ++pair_start[ro + ph_of_k[b] + 1];
      }
    }
    for (int q = 0; q < nblk_all; ++q) pair_start[q + 1] += pair_start[q];
    blk_i.resize(nblk_all); blk_j.resize(nblk_all);
    for (int i = 0, q = 0; i < nPf; ++i)
      for (int j = i; j < nPf; ++j, ++q) { blk_i[q] = i; blk_j[q] = j; }
    // ONE pinned block, ONE upload for the whole structure (eleven copies out of pageable vectors cost ~0.2 ms of the 0.45 ms this
    // pass took): the pair lists -- the largest array -- are written straight into it
    nblk = nblk_all;
    const size_t npairs = (size_t)std::max(pair_start[nblk_all], 1);
    size_t off = 0;
    auto place = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    const size_t o_act = place((size_t)std::max(Ea, 1) * 4), o_pose_h = place((size_t)P * 4), o_pt_h = place((size_t)L * 4),
                 o_pose_of_h = place((size_t)std::max(nPf, 1) * 4), o_pt_of_h = place((size_t)std::max(nLa, 1) * 4), o_pt_start = place((size_t)(nLa + 1) * 4),
                 o_ps_start = place((size_t)(nPf + 1) * 4), o_ps_edges = place(ps_edges.size() * 4), o_blk_i = place((size_t)std::max(nblk, 1) * 4),
                 o_blk_j = place((size_t)std::max(nblk, 1) * 4), o_pair_start = place((size_t)(nblk + 1) * 4), o_pairs = place(npairs * 8),
                 o_pair_h = place(npairs * 4), o_ps_h = place(ps_edges.size() * 4);
    if ((r2 = s->sblk.ensure(ctx, off)) != ASD_OK) return r2;
    if (s->h_sblk_cap < off) {
      if (s->h_sblk) (void)hipHostFree(s->h_sblk);
      s->h_sblk = nullptr; s->h_sblk_cap = 0;
      ASD_HIP_CHECK(ctx, hipHostMalloc(reinterpret_cast<void**>(&s->h_sblk), off + off / 4));
      s->h_sblk_cap = off + off / 4;
    }
    char* hb = s->h_sblk;
    auto put = [&](size_t o, const std::vector<int>& v) { if (!v.empty()) memcpy(hb + o, v.data(), v.size() * 4); };
    put(o_act, act); put(o_pose_h, pose_h); put(o_pt_h, pt_h); put(o_pose_of_h, pose_of_h); put(o_pt_of_h, pt_of_h); put(o_pt_start, pt_start);
    put(o_ps_start, ps_start); put(o_ps_edges, ps_edges); put(o_pair_start, pair_start); put(o_ps_h, ps_hv);
    {
      int* bi_ = reinterpret_cast<int*>(hb + o_blk_i);
      int* bj_ = reinterpret_cast<int*>(hb + o_blk_j);
      for (int i = 0, q = 0; i < nPf; ++i)
        for (int j = i; j < nPf; ++j, ++q) { bi_[q] = i; bj_[q] = j; }
      int2* pp = reinterpret_cast<int2*>(hb + o_pairs);
      int* ph_ = reinterpret_cast<int*>(hb + o_pair_h);
      cursor.assign(pair_start.begin(), pair_start.end() - 1);
      for (int h = 0; h < nLa; ++h) {
        const int s1 = pt_start[h + 1];
        int a = pt_start[h];
        while (a < s1 && ph_of_k[a] < 0) ++a;
        for (; a < s1; ++a) {
          const int ro = row_off[ph_of_k[a]];
          for (int b = a; b < s1; ++b) { const int o = cursor[ro + ph_of_k[b]]++; pp[o] = make_int2(a, b); ph_[o] = h; }
        }
      }
    }
    n_pairs_total = npairs;
    ASD_HIP_CHECK(ctx, hipMemcpyAsync(s->sblk.p, hb, off, hipMemcpyHostToDevice, st));
    char* db = s->sblk.as<char>();
    d.act = reinterpret_cast<const int*>(db + o_act); d.pose_h = reinterpret_cast<const int*>(db + o_pose_h); d.pt_h = reinterpret_cast<const int*>(db + o_pt_h);
    d.pose_of_h = reinterpret_cast<const int*>(db + o_pose_of_h); d.pt_of_h = reinterpret_cast<const int*>(db + o_pt_of_h);
    d.pt_start = reinterpret_cast<const int*>(db + o_pt_start); d.ps_start = reinterpret_cast<const int*>(db + o_ps_start);
    d.ps_edges = reinterpret_cast<const int*>(db + o_ps_edges); d.ps_h = reinterpret_cast<const int*>(db + o_ps_h);
    sb = SchurBlocks{reinterpret_cast<const int*>(db + o_blk_i), reinterpret_cast<const int*>(db + o_blk_j), reinterpret_cast<const int*>(db + o_pair_start),
                     reinterpret_cast<const int2*>(db + o_pairs), reinterpret_cast<const int*>(db + o_pair_h)};
    d.Ea = Ea; d.nPf = nPf; d.nLa = nLa;
    }   // round_idx == 0
    t_prep = std::chrono::steady_clock::now();
    *iters_out = 0;
    *chi_out = 0;
    if (Ea == 0) return ASD_OK;
    const int gE = (Ea + 255) / 256, gL = (nLa + 255) / 256, gP = std::max((nPf + 255) / 256, 1);
    const int n = 6 * nPf;

    d.gE = gE; d.gL = gL; d.gP = gP;
    // ---- the round on the device.  One "block" = one Levenberg trial as the device sees it: [linearise group -- runs only when
    // the state says a new iteration starts] [solve: Dinv, Y, Schur, dense solve, back-substitution, pose update] [errors at the
    // trial estimate] [control: accept / reject (= flip the estimate buffers or not), next lambda, end of iteration / round].
    // Blocks are enqueued ahead; the host reads the mirrored state once per chunk.  The first chunk is as long as the previous
    // LocalBA's round of the same index took (same map, similar problem), so the usual round needs one synchronisation.
    auto enqueue_block = [&]() -> int {
      hipLaunchKernelGGL(k_ba_linearize, dim3(gE), dim3(256), 0, st, d, robust ? 1 : 0);
      hipLaunchKernelGGL(k_ba_reduce, dim3(nPf * kPoseSplit + gL), dim3(256), 0, st, d);
      if (nPf > 0) {
        hipLaunchKernelGGL(k_ba_schur, dim3(nblk + nPf), dim3(kSchurThreads), 0, st, d, sb, nblk);
        const size_t nbk = (size_t)(nPf * (nPf + 1) / 2);
        static AsdPerDeviceOnce attr_set;   // the dynamic-LDS attribute is per device
        if (attr_set.need(ctx->cfg.device)) {
          ASD_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_ba_chol_lds), hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024));
          ASD_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_ba_solve_lds), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
          attr_set.done(ctx->cfg.device);
        }
        if (nPf <= kSolveMaxBlocks) {
          const size_t lds = (nbk * 36 + (size_t)2 * (kSolveMaxBlocks - 1) * 36 + 192 + 72 + 2) * sizeof(double) + nbk * sizeof(short2) + 16;
          hipLaunchKernelGGL(k_ba_solve_lds, dim3(1), dim3(kSolveThreads), lds, st, d.Apack, d.bs, d.x, n, d.lm);
        } else if (nPf <= 32) {
          const size_t lds = nbk * 36 * sizeof(double) + (2 * 192 + 2) * sizeof(double) + nbk * sizeof(short2) + 16;
          hipLaunchKernelGGL(k_ba_chol_lds, dim3(1), dim3(kCholThreads), lds, st, d.A, d.bs, d.x, n, d.lm);
        } else {
          hipLaunchKernelGGL(k_ba_chol, dim3(1), dim3(1024), 0, st, d.A, d.bs, d.x, n, d.lm);
        }
      }
      hipLaunchKernelGGL(k_ba_step, dim3(gL + gP), dim3(256), 0, st, d);
      hipLaunchKernelGGL(k_ba_error, dim3(gE), dim3(256), 0, st, d, robust ? 1 : 0, 0, 0);   // + the Levenberg control, in its last workgroup
      ASD_HIP_CHECK(ctx, hipGetLastError());
      return ASD_OK;
    };
    // computeActiveErrors + activeRobustChi2 at the round's first estimate (levenberg.cpp:70-76), state reset
    hipLaunchKernelGGL(k_ba_error, dim3(gE), dim3(256), 0, st, d, robust ? 1 : 0, 1, iterations);
    ASD_HIP_CHECK(ctx, hipGetLastError());
    int& predicted = s->lm_blocks[round_idx];
    int chunk = predicted > 0 ? predicted : iterations;
    int done = 0;
    // a round takes at most `iterations` iterations of at most ten trials each (levenberg.cpp:_maxTrialsAfterFailure): that, not a
    // fixed number of chunks, bounds the loop -- a slowly converging problem with many iterations is not a numeric failure
    const int max_trials = iterations * 10 + 2;
    while (iterations > 0 && n_trials < max_trials) {
      for (int b = 0; b < chunk; ++b) if ((r2 = enqueue_block()) != ASD_OK) return r2;
      n_trials += chunk;
      // activeRobustChi2() from the stored edge errors behind every chunk: if the round ended inside it, its report is already there
      hipLaunchKernelGGL(k_ba_chi2_stored, dim3(gE), dim3(256), 0, st, d, robust ? 1 : 0, s->h_partial);
      ASD_HIP_CHECK(ctx, hipGetLastError());
      ASD_HIP_CHECK(ctx, hipStreamSynchronize(st));
      if (s->h_lm->done) break;
      chunk = 2;
    }
    if (iterations > 0 && !s->h_lm->done) { ctx->set_error("asd_local_ba: the Levenberg state did not finish"); return ASD_ERR_NUMERIC; }
    done = s->h_lm->iters_done;
    predicted = s->h_lm->trials;
    if (getenv("ASD_BA_DEBUG")) {
      std::vector<LmLog> lg(kLmLogCap);
      ASD_HIP_CHECK(ctx, hipMemcpy(lg.data(), d.lm_log, sizeof(LmLog) * kLmLogCap, hipMemcpyDeviceToHost));
      for (int q = 0; q < std::min(s->h_lm->trials, kLmLogCap); ++q)
        fprintf(stderr, "[ba] trial %d lambda=%.6e cur=%.9e temp=%.9e scale=%.6e nPf=%d nLa=%d Ea=%d nblk=%d\n", q, lg[q].lambda, lg[q].cur, lg[q].temp, lg[q].scale, nPf, nLa, Ea, nblk);
      fprintf(stderr, "[ba] round %d: %d iterations, %d trials, %d blocks enqueued\n", round_idx, done, s->h_lm->trials, n_trials);
    }
    *iters_out = done;
    if (iterations <= 0) {   // no block ran: report the chi2 of the stored errors as they stand
      hipLaunchKernelGGL(k_ba_chi2_stored, dim3(gE), dim3(256), 0, st, d, robust ? 1 : 0, s->h_partial);
      ASD_HIP_CHECK(ctx, hipGetLastError());
      ASD_HIP_CHECK(ctx, hipStreamSynchronize(st));
    }
    double sum = 0;
    for (int i = 0; i < gE; ++i) sum += s->h_partial[i];
    *chi_out = sum;
    if (timing) {
      const auto t_end = std::chrono::steady_clock::now();
      fprintf(stderr, "[ba round] structure %.0f us (Ea=%d nPf=%d nLa=%d pairs=%zu), %d iterations / %d trials in %.0f us\n",
              std::chrono::duration<double, std::micro>(t_prep - t_round).count(), Ea, nPf, nLa, struct_dev ? (size_t)s->h_counts[3] : n_pairs_total, done, n_trials,
              std::chrono::duration<double, std::micro>(t_end - t_prep).count());
    }
    return ASD_OK;
  };

  d.scale_off = nblk_e;
  // optimizer.initializeOptimization(); optimizer.optimize(its_first)           (Optimizer.cc:601-602)
  rc = run_round(0, pr->its_first, true, &res->chi2_first, &res->iters_first);
  if (rc != ASD_OK) return rc;
  // outlier gating: chi2 > 5.991 || !isDepthPositive -> level 1; robust kernels off  (Optimizer.cc:612-631) -- on the device, no
  // host round trip between the rounds; the flags travel back with the final results
  hipLaunchKernelGGL(k_ba_edge_report, dim3(nblk_e), dim3(256), 0, st, d, s->chi2.as<double>(), s->dpos.as<uint8_t>(), s->lvl.as<uint8_t>(), s->out1.as<uint8_t>());
  ASD_HIP_CHECK(ctx, hipGetLastError());
  // optimizer.initializeOptimization(0); optimizer.optimize(its_second)         (Optimizer.cc:647-648)
  rc = run_round(1, pr->its_second, false, &res->chi2_second, &res->iters_second);
  if (rc != ASD_OK) return rc;
  hipLaunchKernelGGL(k_ba_edge_report, dim3(nblk_e), dim3(256), 0, st, d, s->chi2.as<double>(), s->dpos.as<uint8_t>(), (uint8_t*)nullptr, (uint8_t*)nullptr);
  ASD_HIP_CHECK(ctx, hipGetLastError());
  ASD_HIP_CHECK(ctx, hipEventRecord(ev1, st));
  const auto t_rounds = std::chrono::steady_clock::now();
  const int fin = s->h_lm->cur & 1;   // the buffer that holds the accepted estimate (mirrored at the round's last synchronisation)
  ASD_HIP_CHECK(ctx, hipMemcpyAsync(res->edge_chi2, s->chi2.p, (size_t)E * 8, hipMemcpyDeviceToHost, st));
  ASD_HIP_CHECK(ctx, hipMemcpyAsync(res->edge_depth_pos, s->dpos.p, (size_t)E, hipMemcpyDeviceToHost, st));
  ASD_HIP_CHECK(ctx, hipMemcpyAsync(res->edge_outlier1, s->out1.p, (size_t)E, hipMemcpyDeviceToHost, st));
  ASD_HIP_CHECK(ctx, hipMemcpyAsync(hp.data(), fin ? s->pose_bak.p : s->pose.p, (size_t)P * sizeof(Pose7), hipMemcpyDeviceToHost, st));
  ASD_HIP_CHECK(ctx, hipMemcpyAsync(pr->points, fin ? s->pts_bak.p : s->pts.p, (size_t)L * 24, hipMemcpyDeviceToHost, st));
  ASD_HIP_CHECK(ctx, hipStreamSynchronize(st));
  ASD_HIP_CHECK(ctx, hipEventElapsedTime(ms, ev0, ev1));
  for (int p = 0; p < P; ++p) {
    double* q = pr->poses + 7 * p;
    q[0] = hp[p].qx; q[1] = hp[p].qy; q[2] = hp[p].qz; q[3] = hp[p].qw; q[4] = hp[p].tx; q[5] = hp[p].ty; q[6] = hp[p].tz;
  }
#undef ENS
  if (timing) {
    auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    fprintf(stderr, "[ba call] buffers + problem upload %.0f us, rounds (structure, trials, gating) %.0f us, results %.0f us\n", us(t_enter, t_uploaded),
            us(t_uploaded, t_rounds), us(t_rounds, std::chrono::steady_clock::now()));
  }
  return ASD_OK;
}

bool lane_outstanding(BaState* s) {
  if (!s->lane) return false;
  std::lock_guard<std::mutex> l(s->lane->m);
  return s->lane->has_job || s->lane->busy || s->lane->done;
}

void lane_main(asd_ctx* ctx, BaState* s, BaLane* ln) {
  (void)hipSetDevice(ctx->cfg.device);
  for (;;) {
    asd_ba_problem* pr;
    asd_ba_result* res;
    {
      std::unique_lock<std::mutex> l(ln->m);
      ln->cv.wait(l, [&] { return ln->has_job || ln->quit; });
      if (!ln->has_job) return;
      pr = ln->pr; res = ln->res;
      ln->has_job = false;
      ln->busy = true;
    }
    float ms = 0.f;
    const int rc = local_ba_impl(ctx, s, pr, res, ln->st, ln->e0, ln->e1, &ms);
    {
      std::lock_guard<std::mutex> l(ln->m);
      ln->rc = rc; ln->ms = ms;
      ln->busy = false;
      ln->done = true;
    }
    ln->cv.notify_all();
  }
}

}  // namespace

extern "C" {

int asd_local_ba(asd_ctx* ctx, asd_ba_problem* pr, asd_ba_result* res) {
  int rc = local_ba_check(ctx, pr, res);
  if (rc != ASD_OK) return rc;
  if (asd_track_busy(ctx, "asd_local_ba")) return ASD_ERR_INVALID;   // runs on the context's stream behind the outstanding stage: use the lane
  BaState* s = ba_state(ctx);
  if (lane_outstanding(s)) {
    ctx->set_error("asd_local_ba: a run submitted with asd_local_ba_submit is outstanding (the solver's device buffers are in use); call asd_local_ba_wait first");
    return ASD_ERR_INVALID;
  }
  return local_ba_impl(ctx, s, pr, res, ctx->stream, ctx->ev0, ctx->ev1, &ctx->ms_ba);
}

int asd_local_ba_submit(asd_ctx* ctx, asd_ba_problem* pr, asd_ba_result* res) {
  int rc = local_ba_check(ctx, pr, res);
  if (rc != ASD_OK) return rc;
  (void)hipSetDevice(ctx->cfg.device);
  BaState* s = ba_state(ctx);
  if (!s->lane) {   // built completely before it is published
    BaLane* ln = new BaLane();
    auto fail = [&](const char* what) {
      if (ln->st) { (void)hipStreamDestroy(ln->st); }
      if (ln->e0) (void)hipEventDestroy(ln->e0);
      if (ln->e1) (void)hipEventDestroy(ln->e1);
      delete ln;
      ctx->set_error("asd_local_ba_submit: %s failed", what);
      return ASD_ERR_HIP;
    };
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    // Highest priority, like the tracking stream: a run is ~150 short kernels with a host decision every ten; while it is in
    // flight every small kernel of the tracking stream takes 10-30 us longer (rocprof: k_pose_edges 10 -> 34 us, the result copy
    // 5 -> 15 us), so the run should be over quickly.  Measured: 964-976 frames/s with the lane at the highest priority against
    // 922-931 at the lowest (and 850-875 with LocalBA in line).
    if (hipStreamCreateWithPriority(&ln->st, hipStreamNonBlocking, prio_greatest) != hipSuccess) return fail("hipStreamCreateWithPriority");
    if (hipEventCreate(&ln->e0) != hipSuccess || hipEventCreate(&ln->e1) != hipSuccess) return fail("hipEventCreate");
    ln->th = std::thread(lane_main, ctx, s, ln);
    s->lane = ln;
  }
  BaLane* ln = s->lane;
  {
    std::lock_guard<std::mutex> l(ln->m);
    if (ln->has_job || ln->busy || ln->done) {
      ctx->set_error("asd_local_ba_submit: the previous submission has not been collected with asd_local_ba_wait (one run at a time)");
      return ASD_ERR_INVALID;
    }
    ln->pr = pr; ln->res = res;
    ln->has_job = true;
  }
  ln->cv.notify_all();
  return ASD_OK;
}

int asd_local_ba_wait(asd_ctx* ctx) {
  if (!ctx) return ASD_ERR_INVALID;
  BaState* s = ba_state(ctx);
  BaLane* ln = s->lane;
  if (!ln) { ctx->set_error("asd_local_ba_wait: nothing was submitted"); return ASD_ERR_INVALID; }
  std::unique_lock<std::mutex> l(ln->m);
  if (!ln->has_job && !ln->busy && !ln->done) { ctx->set_error("asd_local_ba_wait: nothing was submitted"); return ASD_ERR_INVALID; }
  ln->cv.wait(l, [&] { return ln->done; });
  ln->done = false;
  ctx->ms_ba = ln->ms;
  return ln->rc;
}

int asd_local_ba_poll(asd_ctx* ctx) {
  if (!ctx) return ASD_ERR_INVALID;
  BaState* s = ba_state(ctx);
  BaLane* ln = s->lane;
  if (!ln) return 0;
  std::lock_guard<std::mutex> l(ln->m);
  return (ln->has_job || ln->busy) ? 1 : (ln->done ? 2 : 0);
}

}  // extern "C"
