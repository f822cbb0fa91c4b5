// asdnet.hip -- ASDNet patch -> descriptor forward pass on gfx950.
//
// Replaces module->forward in computeSIFTDescriptors (reference
// src/vslam/src/ORBextractor.cc:1125-1132), i.e. ASDNet.forward
// (ASDNet/ASDNet/ASDNet.py:334-370) with L2Norm (Utils.py:15-22).  The reference launches
// the TorchScript module once per pyramid level with an H2D and a D2H copy each
// (ORBextractor.cc:1217-1231); here all levels' patches go through one batch.
//
// Arithmetic: f32 results with f32 accumulation on the matrix cores, in one of two forms per layer (ctx->net_split):
//   K1  (k_conv_mfma, k_conv_mfma_p): f32 operands on v_mfma_f32_32x32x2_f32 (an exact f32 fma chain);
//   K1s (k_conv_x3, default): each f32 operand split exactly into three bf16 terms, six products per multiply-add on
//        the bf16 MFMA (v_mfma_f32_16x16x32_bf16) -- same error level, 3/8 of the matrix-pipe time (see K1s).
// BatchNorm (eval, affine=False) is folded into the conv weights and a per-channel bias at upload.  Activations are NHWC f32 so the contraction index (cin) is
// contiguous; every 3x3 conv is an implicit GEMM  [pixels x 9*cin] * [9*cin x cout]:
//   A operand  = zero-padded input band staged once per workgroup in LDS,
//   B operand  = per-(tap, cin-chunk) weight slices streamed through a 2-deep LDS ring,
//   k order    = permuted so each lane fetches 4 consecutive cin with one ds_read_b128
//                (lane half h, element jj <-> cin 8*c8 + 4*h + jj) -- the weight image is
//                laid out to match, see build_wimg().
#include "ctx.h"

#include <algorithm>
#include <cmath>
#include <cstring>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// (Weight-ring fill by LDS-DMA -- global_load_lds_dwordx4 instead of global -> VGPR -> ds_write -- was measured slower on every
// layer at N = 2000: conv2 342 -> 357, conv4 314 -> 334, conv5 208 -> 216, conv6 316 -> 322 us; the variant is gone.)

namespace {

struct LayerSpec { int cout, cin, k, stride, pad; };
const LayerSpec kLayers[7] = {{32, 1, 3, 1, 1},   {32, 32, 3, 1, 1},   {64, 32, 3, 2, 1},  {64, 64, 3, 1, 1},
                              {128, 64, 3, 2, 1}, {128, 128, 3, 1, 1}, {128, 128, 8, 1, 0}};

// ------------------------------------------------------------------------------------------
// K1: 3x3 conv (pad 1, stride S) + folded BN + ReLU as an implicit GEMM on f32 MFMA.
// One workgroup = ROWS output rows of one patch; 4 waves as WM (pixels) x WN (channels).
// ------------------------------------------------------------------------------------------
// RING = weight ring depth of k_conv_mfma.  3 = software-pipelined form: the stage written at the end of stage s is s+2, so the
// slot of stage s+1 has been complete since the previous barrier and the first operands of the next stage can be
// fetched from LDS BEFORE the barrier that ends the current one (with depth 2 every stage began with an exposed
// LDS round trip: 7-13 % of the MFMA loop).  2 = one slot less LDS; measured per layer: 3 wins for conv5/conv6
// (-22 us on conv6), 2 for conv2/conv4 where the extra slot costs a resident workgroup or nothing is gained.
template <int CIN, int COUT, int HIN, int S, int ROWS, int WM, int WN, int KC, int PP = 1, int RING = 2>
struct ConvCfg {
  static constexpr int HO = HIN / S;
  static constexpr int INROWS = (ROWS - 1) * S + 3;
  static constexpr int INCOLS = (HO - 1) * S + 3;
  static constexpr int CPAD = CIN + 4;  // 16-B aligned pixel stride, odd multiple of 4 floats -> b128 reads spread over banks
  static constexpr int NW = WM * WN;    // waves per workgroup (4 or 8)
  static constexpr int NTH = 64 * NW;
  static constexpr int M_PATCH = ROWS * HO;   // output pixels per patch band
  static constexpr int M_WG = PP * M_PATCH;   // PP patches share one weight ring
  static constexpr int MT = M_WG / 32 / WM;
  static constexpr int NT = COUT / 32 / WN;
  static constexpr int NCC = CIN / KC;
  static constexpr int NSTAGE = 9 * NCC;
  static constexpr int WCHUNK = KC * COUT;  // floats per weight stage
  static constexpr int WQUADS = WCHUNK / 4;  // float4 per weight stage
  static constexpr int WREGS = (WQUADS + NTH - 1) / NTH;
  static constexpr int ACT_FLOATS = INROWS * INCOLS * CPAD;  // per patch band
  static constexpr int LDS_BYTES = (PP * ACT_FLOATS + RING * WCHUNK) * 4;
  static_assert(RING == 2 || RING == 3, "weight ring depth");
  static_assert(NW == 4 || NW == 8, "4 or 8 waves per workgroup");
  static_assert(M_WG % (32 * WM) == 0 && COUT % (32 * WN) == 0, "tile split");
  static_assert(WQUADS % NTH == 0 || WQUADS < NTH, "weight stage: whole float4 rounds per thread, or a single partial round");
  static_assert(HO % ROWS == 0 && CIN % KC == 0 && KC % 8 == 0, "shape");
  static_assert(PP == 1 || ROWS == HO, "several patches per workgroup only for whole-patch bands");
  static_assert(M_PATCH % 32 == 0, "32-pixel MFMA tiles must not straddle patches");
};

// FUSE1: the kernel is conv2 and computes its own input (input_norm + conv1 + BN + ReLU, ASDNet.py:334-336,
// 360-365) from the raw u8 patch while filling the LDS band, so conv1's 128 KB/patch activation never
// exists in HBM.  `in` is then the u8 patch array and w1 / b1 the folded conv1 weights.
template <int CIN, int COUT, int HIN, int S, int ROWS, int WM, int WN, int KC, int PP = 1, int RING = 2, int ABL = 0, bool FUSE1 = false>
__global__ __launch_bounds__(64 * WM * WN) void k_conv_mfma(const void* __restrict__ in_, const float* __restrict__ wimg,
                                                           const float* __restrict__ bias, float* __restrict__ out,
                                                           const float* __restrict__ w1, const float* __restrict__ b1, int n) {
  using C = ConvCfg<CIN, COUT, HIN, S, ROWS, WM, WN, KC, PP, RING>;
  constexpr int NTH = C::NTH;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sact = smem;
  float* sw = smem + PP * C::ACT_FLOATS;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave / WN, wn = wave % WN;
  constexpr int BANDS = C::HO / ROWS;
  const int patch = (blockIdx.x / BANDS) * PP, band = blockIdx.x % BANDS;  // first patch of this workgroup
  const int r0 = band * ROWS;

  if constexpr (FUSE1) {
    static_assert(!FUSE1 || (CIN == 32 && HIN == 32 && S == 1 && PP == 1), "conv1 fusion is for conv2 only");
    // extra LDS behind the weight ring: normalised input rows r0-2 .. r0+ROWS+1 (34 wide, zero padded),
    // conv1 weights + bias, reduction scratch
    float* pin = sw + RING * C::WCHUNK;           // [(ROWS+4)][36]
    float* wsh = pin + (ROWS + 4) * 36;           // [32*9 + 32]
    float* red = wsh + 320;                       // [8]
    const uint8_t* patches = static_cast<const uint8_t*>(in_);
    for (int i = t; i < (ROWS + 4) * 36; i += NTH) pin[i] = 0.f;
    for (int i = t; i < 288; i += NTH) wsh[i] = w1[i];
    if (t < 32) wsh[288 + t] = b1[t];
    // the 1024 pixels of the patch sit in the first four waves (4 per lane); further waves only help with conv1 below
    const bool ld = C::NW == 4 || wave < 4;
    const uchar4 v = ld ? reinterpret_cast<const uchar4*>(patches + (size_t)patch * 1024)[t] : make_uchar4(0, 0, 0, 0);
    const float inv255 = (float)(1.0 / 255);  // ORBextractor.cc:1125
    float x[4] = {v.x * inv255, v.y * inv255, v.z * inv255, v.w * inv255};
    // the ROUNDED products are the operands of everything below: a constant patch must give mean == pixel and x - mean == 0
    // exactly (as the reference's double-accumulated mean does); contracted into fma(v, 1/255, -mean) the difference would be
    // the product's rounding error, which (x - mean) / (0 + 1e-7) turns into an O(0.1) input.  (HIP's __fmul_rn is a plain
    // multiply, so the barrier is an empty asm on the value.)
#pragma unroll
    for (int k = 0; k < 4; ++k) asm volatile("" : "+v"(x[k]));
    float s = (x[0] + x[1]) + (x[2] + x[3]);
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0 && ld) red[wave] = s;
    asd_syncthreads();
    const float mean = ((red[0] + red[1]) + (red[2] + red[3])) * (1.0f / 1024.0f);
    float d[4], ss = 0.f;
    for (int k = 0; k < 4; ++k) { d[k] = x[k] - mean; ss += d[k] * d[k]; }
    for (int off = 32; off >= 1; off >>= 1) ss += __shfl_xor(ss, off);
    if (lane == 0 && ld) red[4 + wave] = ss;
    asd_syncthreads();
    const float sd = sqrtf(((red[4] + red[5]) + (red[6] + red[7])) * (1.0f / 1023.0f)) + 1e-7f;  // unbiased std
    if (ld) {
      const int idx = t * 4, y = idx >> 5, x0 = idx & 31;  // this thread's 4 pixels sit in row y
      const int j = y - (r0 - 2);
      if (j >= 0 && j < ROWS + 4)
        for (int k = 0; k < 4; ++k) pin[j * 36 + x0 + k + 1] = d[k] / sd;
    }
    asd_syncthreads();
    // conv1 output for rows r0-1 .. r0+ROWS, cols -1 .. 32 (zero outside the 32x32 map: conv2's padding).
    // item = (pixel, quad of 4 couts); NTH is a multiple of 8, so a thread keeps the same quad for all its items and
    // holds that quad's 36 weights + 4 biases in registers (they were 40 LDS reads per item before)
    constexpr int NPIX = C::INROWS * C::INCOLS;
    static_assert(NTH % 8 == 0, "a thread must keep its cout quad across items");
    if (!(ABL & 1)) {
      const int q = t & 7;
      float wq[4][9], bq[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        bq[c] = wsh[288 + 4 * q + c];
#pragma unroll
        for (int k = 0; k < 9; ++k) wq[c][k] = wsh[(4 * q + c) * 9 + k];
      }
      for (int item = t; item < NPIX * 8; item += NTH) {
        const int pix = item >> 3;
        const int i = pix % C::INCOLS, j = pix / C::INCOLS;
        const int oy = r0 - 1 + j, ox = i - 1;
        f32x4 r = {0.f, 0.f, 0.f, 0.f};
        if (oy >= 0 && oy < 32 && ox >= 0 && ox < 32) {
          float a[9];
#pragma unroll
          for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) a[ky * 3 + kx] = pin[(j + ky) * 36 + ox + kx];  // pin row j <-> image row oy-1
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            float acc1 = bq[c];
#pragma unroll
            for (int k = 0; k < 9; ++k) acc1 += a[k] * wq[c][k];
            r[c] = acc1 > 0.f ? acc1 : 0.f;
          }
        }
        *reinterpret_cast<f32x4*>(sact + pix * C::CPAD + 4 * q) = r;
      }
    }
  } else {
    const float* inp = static_cast<const float*>(in_) + (size_t)patch * HIN * HIN * CIN;
    // ---- stage the zero-padded input band(s) (NHWC rows are contiguous: coalesced 16-B loads)
    constexpr int C4 = CIN / 4;
    constexpr int NPIXB = C::INROWS * C::INCOLS;
    if (!(ABL & 1))
    for (int idx = t; idx < PP * NPIXB * C4; idx += NTH) {
      const int c4 = idx % C4, pixg = idx / C4;
      const int pp = pixg / NPIXB, pix = pixg % NPIXB;
      const int i = pix % C::INCOLS, j = pix / C::INCOLS;
      const int iy = r0 * S - 1 + j, ix = i - 1;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (iy >= 0 && iy < HIN && ix >= 0 && ix < HIN && patch + pp < n)
        v = *reinterpret_cast<const f32x4*>(inp + (size_t)pp * HIN * HIN * CIN + ((size_t)iy * HIN + ix) * CIN + c4 * 4);
      *reinterpret_cast<f32x4*>(sact + pp * C::ACT_FLOATS + pix * C::CPAD + c4 * 4) = v;
    }
  }
  // ---- weight stage 0 (and 1 for the 3-deep ring)
  constexpr int NPRE = RING == 3 ? 2 : 1;
  for (int ps = 0; ps < NPRE && ps < C::NSTAGE; ++ps)
    for (int r = 0; r < C::WREGS; ++r)
      if (C::WQUADS >= NTH || t < C::WQUADS)
        *reinterpret_cast<f32x4*>(sw + ps * C::WCHUNK + (r * NTH + t) * 4) =
            *reinterpret_cast<const f32x4*>(wimg + (size_t)ps * C::WCHUNK + (r * NTH + t) * 4);
  asd_syncthreads();

  f32x16 acc[C::MT][C::NT];
  for (int mt = 0; mt < C::MT; ++mt)
    for (int nt = 0; nt < C::NT; ++nt)
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  const int h = lane >> 5, li = lane & 31;
  int abase[C::MT];
  for (int mt = 0; mt < C::MT; ++mt) {
    const int m = (wm * C::MT + mt) * 32 + li;
    const int pp = PP > 1 ? m / C::M_PATCH : 0, mm = m - pp * C::M_PATCH;
    const int rr = mm / C::HO, ox = mm % C::HO;
    abase[mt] = pp * C::ACT_FLOATS + ((rr * S) * C::INCOLS + ox * S) * C::CPAD + 4 * h;
  }
  const int bbase = (h * COUT + wn * C::NT * 32 + li) * 4;
  constexpr int STEPS = KC / 8;

  // operands of k-step (stage s, chunk c8): A from the band at the tap's offset, B from the stage's ring slot
  auto load_ops = [&](int s, int c8, f32x4 (&a)[C::MT], f32x4 (&b)[C::NT]) {
    const int tap = s / C::NCC, cc = s % C::NCC;
    const int tapoff = ((tap / 3) * C::INCOLS + (tap % 3)) * C::CPAD + cc * KC;
    const float* swb = sw + (s % RING) * C::WCHUNK;
#pragma unroll
    for (int mt = 0; mt < C::MT; ++mt) a[mt] = *reinterpret_cast<const f32x4*>(sact + abase[mt] + tapoff + c8 * 8);
#pragma unroll
    for (int nt = 0; nt < C::NT; ++nt) b[nt] = *reinterpret_cast<const f32x4*>(swb + bbase + c8 * 2 * COUT * 4 + nt * 128);
  };
  auto mfma_step = [&](const f32x4 (&a)[C::MT], const f32x4 (&b)[C::NT]) {
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
      for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < C::NT; ++nt)
          if (!(ABL & 2)) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt][jj], b[nt][jj], acc[mt][nt], 0, 0, 0);
          else { asm volatile("" :: "v"(a[mt][jj]), "v"(b[nt][jj])); }
  };

  if constexpr (RING == 3) {
    f32x4 a0[C::MT], b0[C::NT], a1[C::MT], b1[C::NT];
    load_ops(0, 0, a0, b0);
    for (int s = 0; s < C::NSTAGE; ++s) {
      f32x4 wreg[C::WREGS];
      const bool pre = s + 2 < C::NSTAGE && !(ABL & 16);
      if (pre) {
        const float* wsrc = wimg + (size_t)(s + 2) * C::WCHUNK;
        for (int r = 0; r < C::WREGS; ++r)
          if (C::WQUADS >= NTH || t < C::WQUADS) wreg[r] = *reinterpret_cast<const f32x4*>(wsrc + (r * NTH + t) * 4);
      }
      // k-steps of this stage, operands always one step ahead (the last step fetches the next stage's first operands:
      // its ring slot was completed before the previous barrier)
#pragma unroll
      for (int c8 = 0; c8 < STEPS; c8 += 2) {
        if (c8 + 1 < STEPS) load_ops(s, c8 + 1, a1, b1);
        else if (s + 1 < C::NSTAGE) load_ops(s + 1, 0, a1, b1);
        mfma_step(a0, b0);
        if (c8 + 1 < STEPS) {
          if (c8 + 2 < STEPS) load_ops(s, c8 + 2, a0, b0);
          else if (s + 1 < C::NSTAGE) load_ops(s + 1, 0, a0, b0);
          mfma_step(a1, b1);
        } else {
#pragma unroll
          for (int mt = 0; mt < C::MT; ++mt) a0[mt] = a1[mt];
#pragma unroll
          for (int nt = 0; nt < C::NT; ++nt) b0[nt] = b1[nt];
        }
      }
      if (pre) {
        float* swn = sw + ((s + 2) % 3) * C::WCHUNK;
        for (int r = 0; r < C::WREGS; ++r)
          if (C::WQUADS >= NTH || t < C::WQUADS) *reinterpret_cast<f32x4*>(swn + (r * NTH + t) * 4) = wreg[r];
      }
      if (!(ABL & 8)) asd_syncthreads();
    }
  } else {
    for (int s = 0; s < C::NSTAGE; ++s) {
      f32x4 wreg[C::WREGS];
      if (s + 1 < C::NSTAGE && !(ABL & 16)) {
        const float* wsrc = wimg + (size_t)(s + 1) * C::WCHUNK;
        for (int r = 0; r < C::WREGS; ++r)
          if (C::WQUADS >= NTH || t < C::WQUADS) wreg[r] = *reinterpret_cast<const f32x4*>(wsrc + (r * NTH + t) * 4);
      }
#pragma unroll
      for (int c8 = 0; c8 < STEPS; ++c8) {
        f32x4 a[C::MT], b[C::NT];
        load_ops(s, c8, a, b);
        mfma_step(a, b);
      }
      if (s + 1 < C::NSTAGE && !(ABL & 16)) {
        float* swn = sw + ((s + 1) & 1) * C::WCHUNK;
        for (int r = 0; r < C::WREGS; ++r)
          if (C::WQUADS >= NTH || t < C::WQUADS) *reinterpret_cast<f32x4*>(swn + (r * NTH + t) * 4) = wreg[r];
      }
      if (!(ABL & 8)) asd_syncthreads();
    }
  }

  // ---- epilogue: bias (folded BN) + ReLU, NHWC store.  Lane owns one cout column, 16 pixel rows.
  float* op = out + ((size_t)patch * C::HO + r0) * C::HO * COUT;
  for (int nt = 0; nt < C::NT; ++nt) {
    const int co = (wn * C::NT + nt) * 32 + li;
    const float bv = bias[co];
    for (int mt = 0; mt < C::MT; ++mt) {
      const int pp = PP > 1 ? ((wm * C::MT + mt) * 32) / C::M_PATCH : 0;  // a 32-pixel tile never straddles two patches
      if (PP > 1 && patch + pp >= n) continue;
      for (int r = 0; r < 16; ++r) {
        const int m = (wm * C::MT + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const size_t o = (size_t)pp * C::HO * C::HO * COUT + (size_t)(m - pp * C::M_PATCH) * COUT + co;
        float v = acc[mt][nt][r] + bv;
        if (!(ABL & 4)) op[o] = v > 0.f ? v : 0.f;
        else { asm volatile("" :: "v"(v)); }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// K1p (conv3 only; a 3-deep ring with operand prefetch as in K1 was measured here too: 221 vs 202 us):
// the same implicit GEMM as a PERSISTENT workgroup with a double-buffered input band: while the
// MFMAs of tile i run, the band of tile i+gridDim.x is fetched from HBM into registers (issued in
// stage 0 behind that stage's weight prefetch, so the counted vmcnt of the weight write does not
// drain it) and written to the other LDS buffer at the end of stage 1; the epilogue stores of tile i
// drain under tile i+1's MFMAs.  The weight ring simply keeps cycling across tiles.
// ------------------------------------------------------------------------------------------
template <int CIN, int COUT, int HIN, int S, int ROWS, int WM, int WN, int KC>
__global__ __launch_bounds__(256) void k_conv_mfma_p(const float* __restrict__ in, const float* __restrict__ wimg,
                                                     const float* __restrict__ bias, float* __restrict__ out, int ntiles) {
  using C = ConvCfg<CIN, COUT, HIN, S, ROWS, WM, WN, KC>;
  static_assert(C::NSTAGE >= 2, "need two stages to hide the band prefetch");
  static_assert(C::NW == 4, "the persistent form is written for 4 waves");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sact0 = smem;
  float* sw = smem + 2 * C::ACT_FLOATS;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave / WN, wn = wave % WN;
  constexpr int BANDS = C::HO / ROWS;
  constexpr int C4 = CIN / 4;
  constexpr int NELEM = C::INROWS * C::INCOLS * C4;
  constexpr int NPREF = (NELEM + 255) / 256;
  int tile = blockIdx.x;
  if (tile >= ntiles) return;

  f32x4 pre[NPREF];
#define LOAD_TILE(TILE)                                                                           \
  {                                                                                               \
    const int patch_ = (TILE) / BANDS, r0_ = ((TILE) % BANDS) * ROWS;                             \
    const float* inp_ = in + (size_t)patch_ * HIN * HIN * CIN;                                    \
    _Pragma("unroll") for (int p = 0; p < NPREF; ++p) {                                           \
      const int idx = t + p * 256;                                                                \
      f32x4 v = {0.f, 0.f, 0.f, 0.f};                                                             \
      if (idx < NELEM) {                                                                          \
        const int c4 = idx % C4, pix = idx / C4;                                                  \
        const int i = pix % C::INCOLS, j = pix / C::INCOLS;                                       \
        const int iy = r0_ * S - 1 + j, ix = i - 1;                                               \
        if (iy >= 0 && iy < HIN && ix >= 0 && ix < HIN)                                           \
          v = *reinterpret_cast<const f32x4*>(inp_ + ((size_t)iy * HIN + ix) * CIN + c4 * 4);     \
      }                                                                                           \
      pre[p] = v;                                                                                 \
    }                                                                                             \
  }
#define STORE_TILE(DST)                                                                           \
  {                                                                                               \
    _Pragma("unroll") for (int p = 0; p < NPREF; ++p) {                                           \
      const int idx = t + p * 256;                                                                \
      if (idx < NELEM) {                                                                          \
        const int c4 = idx % C4, pix = idx / C4;                                                  \
        *reinterpret_cast<f32x4*>((DST) + pix * C::CPAD + c4 * 4) = pre[p];                       \
      }                                                                                           \
    }                                                                                             \
  }
  LOAD_TILE(tile);
  STORE_TILE(sact0);
  for (int r = 0; r < C::WREGS; ++r)
    *reinterpret_cast<f32x4*>(sw + (r * 256 + t) * 4) = *reinterpret_cast<const f32x4*>(wimg + (r * 256 + t) * 4);
  asd_syncthreads();

  const int h = lane >> 5, li = lane & 31;
  int abase[C::MT];
  for (int mt = 0; mt < C::MT; ++mt) {
    const int m = (wm * C::MT + mt) * 32 + li;
    const int rr = m / C::HO, ox = m % C::HO;
    abase[mt] = ((rr * S) * C::INCOLS + ox * S) * C::CPAD + 4 * h;
  }
  const int bbase = (h * COUT + wn * C::NT * 32 + li) * 4;
  float bv[C::NT];
  for (int nt = 0; nt < C::NT; ++nt) bv[nt] = bias[(wn * C::NT + nt) * 32 + li];

  int cur = 0, g = 0;
  while (true) {
    const int next = tile + gridDim.x;
    const bool has_next = next < ntiles;
    const float* sact = sact0 + cur * C::ACT_FLOATS;
    f32x16 acc[C::MT][C::NT];
    for (int mt = 0; mt < C::MT; ++mt)
      for (int nt = 0; nt < C::NT; ++nt)
        for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
    for (int s = 0; s < C::NSTAGE; ++s) {
      f32x4 wreg[C::WREGS];
      {
        const int sn = (s + 1 == C::NSTAGE) ? 0 : s + 1;  // the ring keeps cycling across tiles
        const float* wsrc = wimg + (size_t)sn * C::WCHUNK;
        for (int r = 0; r < C::WREGS; ++r) wreg[r] = *reinterpret_cast<const f32x4*>(wsrc + (r * 256 + t) * 4);
      }
      if (s == 0 && has_next) LOAD_TILE(next);
      const int tap = s / C::NCC, cc = s % C::NCC;
      const int tapoff = ((tap / 3) * C::INCOLS + (tap % 3)) * C::CPAD + cc * KC;
      const float* swb = sw + (g & 1) * C::WCHUNK;
#pragma unroll
      for (int c8 = 0; c8 < KC / 8; ++c8) {
        f32x4 a[C::MT], b[C::NT];
#pragma unroll
        for (int mt = 0; mt < C::MT; ++mt) a[mt] = *reinterpret_cast<const f32x4*>(sact + abase[mt] + tapoff + c8 * 8);
#pragma unroll
        for (int nt = 0; nt < C::NT; ++nt)
          b[nt] = *reinterpret_cast<const f32x4*>(swb + bbase + c8 * 2 * COUT * 4 + nt * 128);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
          for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < C::NT; ++nt)
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt][jj], b[nt][jj], acc[mt][nt], 0, 0, 0);
      }
      {
        float* swn = sw + ((g + 1) & 1) * C::WCHUNK;
        for (int r = 0; r < C::WREGS; ++r) *reinterpret_cast<f32x4*>(swn + (r * 256 + t) * 4) = wreg[r];
      }
      if (s == 1 && has_next) STORE_TILE(sact0 + (cur ^ 1) * C::ACT_FLOATS);
      asd_syncthreads();
      ++g;
    }
    // epilogue: bias (folded BN) + ReLU, NHWC store; drains under the next tile's MFMAs
    {
      const int patch = tile / BANDS, r0 = (tile % BANDS) * ROWS;
      float* op = out + ((size_t)patch * C::HO + r0) * C::HO * COUT;
      for (int nt = 0; nt < C::NT; ++nt) {
        const int co = (wn * C::NT + nt) * 32 + li;
        for (int mt = 0; mt < C::MT; ++mt)
          for (int r = 0; r < 16; ++r) {
            const int m = (wm * C::MT + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const float v = acc[mt][nt][r] + bv[nt];
            op[(size_t)m * COUT + co] = v > 0.f ? v : 0.f;
          }
      }
    }
    if (!has_next) break;
    tile = next;
    cur ^= 1;
  }
#undef LOAD_TILE
#undef STORE_TILE
}

// ------------------------------------------------------------------------------------------
// K1s: the same 3x3 conv + folded BN + ReLU with f32 operands SPLIT into three bf16 terms.
//
// Every f32 value is written exactly as x = h + m + l, three bf16 numbers (8 significant bits each, by truncation:
// h = top 16 bits of x, m = top 16 bits of x - h, l = x - h - m, both subtractions exact).  The product of two such
// values is the sum of nine bf16 x bf16 products, each exact in f32; the six of relative size >= 2^-16
// (hh, hm, mh, mm, hl, lh) go through the bf16 MFMA (shape: ASD_X3_S16 below) with f32 accumulation, the three dropped ones are
// <= 2^-23 relative -- below the rounding of an f32 fma chain.  Measured (tools/ubench/split_mfma.hip, K = 288 /
// 1152 / 8192, rms error against f64): 8.7e-8 / 3.6e-7 / 2.7e-6 for this form, 8.9e-8 / 3.4e-7 / 2.4e-6 for
// v_mfma_f32_32x32x2_f32.  The bf16 pipe runs 16x the f32 MFMA rate, so six products cost 3/8 of the f32 MFMAs.
//
// Interface identical to K1 (f32 NHWC in, f32 NHWC out): the split happens while the zero-padded input band is
// staged into LDS ([pixel][cin/8][h | m | l][8 bf16], 16 B per operand fetch); the weights are split once at upload
// (build_wx3).  With the MFMA time cut to 3/8 the per-stage barrier of K1's weight ring would dominate, so there
// is no weight ring: every wave fetches the B operands of its own 32-cout column straight from global memory
// (L2-resident, two chunks ahead, registers) and runs barrier-free after the band is staged; the workgroup covers
// M_WG >= 128 pixels so that the weight stream stays at <= 16 B/clk/CU.
// ------------------------------------------------------------------------------------------
#ifndef ASD_X3_ABL
#define ASD_X3_ABL 0  // tuning: 1 = weight stream pinned to chunk 0/1 (L1 hits), 2 = no band staging, 4 = no output stores,
                      // 8 = no MFMAs, 32 = conv1 with one tap instead of nine, 64 = conv1 output written unsplit
#endif
// MFMA shape of the split-operand kernels: 1 = v_mfma_f32_16x16x32_bf16 (16-row tiles, 32-deep chunks; default),
// 0 = v_mfma_f32_32x32x16_bf16.  Both take the same cycles; under the 16x16x32 form the chip holds a higher clock (in-kernel
// 1.87-2.25 GHz against 1.67-1.99: conv4 168 -> 154, conv6 165 -> 150 us, ASDNet 0.89 -> 0.82 ms, and the latency-bound
// tracking kernels gain from the clock too: 700 -> 766 frames/s).
// HAZARD (gfx950, measured stand-alone: tools/ubench/mfma_pk_hazard.hip, profiles/r02_mfma_pk_hazard*.txt): a packed-f32
// instruction whose src1 takes its LOW-half operand from the HIGH register of the pair (v_pk_add_f32 ... op_sel:[0,1], what the
// SLP vectoriser emits for "both halves use y.y") drops the low-half result in lanes 48-63 now and then while a wave that
// issues bf16 MFMAs -- either shape, this kernel or anybody's -- is resident on the same CU.  It is a property of the device,
// not of this kernel: the 32x32x16 shape triggers it as well (round 1 thought it did not: it is 8x rarer next to this loop),
// a plain f32 loop does not.  The library therefore contains no packed-f32 arithmetic at all (-fno-slp-vectorize in the
// Makefile; `make check-isa` / tests/test_isa.py), and tests/test_frontend.py runs asd_dist_matrix beside the extractor.
#ifndef ASD_X3_S16
#define ASD_X3_S16 1
#endif
#define ASD_X3_PD 2  // A-operand prefetch distance in sub-tiles
#ifndef ASD_X3_RB
#define ASD_X3_RB 3  // chunks of B operands (weights) in flight per wave, where the layer's chunk count divides by it.  Round 5, N = 2000:
                     // 6 -> conv4 97 -> 105 us, conv5 66 -> 64, conv6 unchanged; 9 -> conv2 150 -> 275 (registers), conv3 96 -> 93, conv5 84, conv6 109
#endif
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int CIN, int COUT, int HIN, int S, int ROWS, int WM, int WN, int PP = 1, int NP = 3>
struct X3Cfg {
  static constexpr int HO = HIN / S;
  static constexpr int INROWS = (ROWS - 1) * S + 3;
  static constexpr int INCOLS = (HO - 1) * S + 3;
  static constexpr int PIXB = CIN * 2 * NP + 16;  // bytes per staged pixel (NP 16-bit pieces per element): an odd multiple of 16 B -> b128 reads of 16 consecutive pixels cover all banks
  static constexpr int GRPB = 16 * NP;           // bytes of one 8-cin group of a pixel: its NP pieces, 16 B each
  // Where the 16-B slot of (pixel, 8-cin group g, piece p) sits inside the pixel: g * GSTR + p * PSTR.  ds_read_b128 is served in the lane
  // groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH.md, LDS): eight rows of k-group kg and the OTHER eight rows of
  // kg + 1 at a time.  With [g][p] (k-groups 32 B apart) and an odd pixel stride the two halves fall on the same 16-B slots whatever the
  // stride is (every A read took two passes: SQ_LDS_BANK_CONFLICT = half of SQ_LDS_IDX_ACTIVE in profiles/r05_asdnet_sq_counters_default.txt)
  // unless the first half's pixels are the even and the second half's the odd ones: x3_tile_pixel below.  Stride-2 layers only ever
  // touch even slots that way; they keep the pieces in planes ([p][g]: k-groups 16 B apart) instead.
  static constexpr bool PLANAR = NP == 2 && S == 2 && ASD_X3_S16 != 0;
  static constexpr int GSTR = PLANAR ? 16 : GRPB;
  static constexpr int PSTR = PLANAR ? CIN * 2 : 16;
  static constexpr int NW = WM * WN;
  static constexpr int NTH = 64 * NW;
  static constexpr int M_PATCH = ROWS * HO;
  static constexpr int M_WG = PP * M_PATCH;
  static constexpr int MT = M_WG / 32 / WM;
  static constexpr int NT = COUT / 32 / WN;
  static constexpr int KCH = ASD_X3_S16 ? 32 : 16;  // cin per k-chunk = K of the MFMA shape
  static constexpr int NC16 = CIN / KCH;            // chunks per tap
  static constexpr int NCHUNK = 9 * NC16;
  static constexpr int CHUNKB = KCH * 2 * NP * COUT;  // bytes of weight image per chunk: [piece NP][k-group KCH/8][cout][8 x 16 bit]
  static constexpr int ACT_BYTES = INROWS * INCOLS * PIXB;
  static constexpr int LDS_BYTES = PP * ACT_BYTES;
  static_assert((PIXB / 16) % 2 == 1, "pixel stride must be an odd multiple of 16 B");
  static constexpr int NCW = COUT / WN;               // couts per wave: a multiple of the MFMA tile width (16 or 32)
  static_assert(M_WG % (32 * WM) == 0 && COUT % WN == 0 && NCW % (ASD_X3_S16 ? 16 : 32) == 0 && CIN % KCH == 0 && HO % ROWS == 0, "tile split");
  static_assert(PP == 1 || ROWS == HO, "several patches per workgroup only for whole-patch bands");
  static_assert(M_PATCH % 32 == 0, "32-pixel MFMA tiles must not straddle patches");
};

// exact 3-way bf16 split of 8 floats -> three packed operand quads (element j in bits 16j..16j+15 of the 128-bit value)
__device__ inline void split8(const f32x4& v0, const f32x4& v1, u32x4& ph, u32x4& pm, u32x4& pl) {
  float x[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
  uint32_t hu[8], mu[8], lu[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const uint32_t u = __float_as_uint(x[j]);
    const float r1 = x[j] - __uint_as_float(u & 0xffff0000u);
    const uint32_t r1u = __float_as_uint(r1);
    const float r2 = r1 - __uint_as_float(r1u & 0xffff0000u);
    hu[j] = u; mu[j] = r1u; lu[j] = __float_as_uint(r2);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {  // v_perm_b32: high halves of the odd (upper) and even (lower) element
    ph[q] = __builtin_amdgcn_perm(hu[2 * q + 1], hu[2 * q], 0x07060302u);
    pm[q] = __builtin_amdgcn_perm(mu[2 * q + 1], mu[2 * q], 0x07060302u);
    pl[q] = __builtin_amdgcn_perm(lu[2 * q + 1], lu[2 * q], 0x07060302u);
  }
}


// activations are multiplied by this before the fp16 split: |x| up to 4094 stays finite (a BN-normalised ReLU output is O(1));
// beyond that the high piece is inf, the low piece NaN and the descriptor comes out NaN rather than silently wrong
constexpr float kActScale = 16.f;

// The two-piece form (NP = 2, ASD_ASDNET_MATH=f16x2): x * 2^k = h + l with h = fp16(x 2^k) (round to nearest) and
// l = fp16(x 2^k - h): 22 significant bits instead of 24, three products (l h, h l, h h) instead of six.  The power-of-two
// pre-scale keeps h inside fp16's range and l out of its subnormals for everything but values that are negligible anyway.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ inline void split8_f16(const f32x4& v0, const f32x4& v1, float scale, u32x4& ph, u32x4& pl) {
  float x[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
  f16x8 h, l;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float xs = x[j] * scale;
    h[j] = (_Float16)xs;
    l[j] = (_Float16)(xs - (float)h[j]);
  }
  ph = __builtin_bit_cast(u32x4, h);
  pl = __builtin_bit_cast(u32x4, l);
}

// Two-piece fp16 split of four f32 values x times the power of two s (an accumulator sub-tile's four couts of one pixel), packed as
// the MFMA operands want them: h = (fp16(s x0), fp16(s x1)), (fp16(s x2), fp16(s x3)); l = the fp16 of (s x - h), the difference formed
// unrounded inside the fma and rounded once -- the values split8_f16 produces, in 8 instructions (the compiler's code for the scalar
// expression converts every h twice).  v_fma_mixhi_f16 keeps the low half of its destination; a VALU result written with a
// destination half-select needs one wait state before a VALU reads it on gfx940-class chips, which the interleaved order provides
// (the closing s_nop covers whatever the compiler schedules next).
__device__ __forceinline__ void split4_mix(float x0, float x1, float x2, float x3, float s, uint32_t (&h)[2], uint32_t (&l)[2]) {
  asm("v_fma_mixlo_f16 %0, %8, %4, 0\n\t"
      "v_fma_mixlo_f16 %1, %8, %6, 0\n\t"
      "v_fma_mixhi_f16 %0, %8, %5, 0\n\t"
      "v_fma_mixhi_f16 %1, %8, %7, 0\n\t"
      "v_fma_mixlo_f16 %2, %8, %4, -%0 op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixlo_f16 %3, %8, %6, -%1 op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %2, %8, %5, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %3, %8, %7, -%1 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
      "s_nop 0"
      : "=&v"(h[0]), "=&v"(h[1]), "=&v"(l[0]), "=&v"(l[1])
      : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "s"(s));
}

// (Round 3 ran the conv layers as PERSISTENT launches as well -- workgroups pulling tiles from a counter, a workgroup that found itself on a
// "reserved" CU leaving without one, to keep CUs free for the tracking stream in software: 998-1010 frames/s against 1069 with one workgroup per
// tile.  Persistent workgroups hold their CUs for a whole layer where the occupancy-based form frees a slot every few microseconds, and the
// tracking kernels' single workgroups did not land on the reserved CUs.  Removed in round 5 with its switches.)

// input_norm's statistics (ASDNet.py:360-365) of every patch, once per forward: mean and unbiased std + 1e-7 over the 1024 pixels.  conv2's
// workgroups (four bands per patch) used to compute them in their prologue, each for itself: the patch load, two wave + LDS reductions and two
// barriers in front of conv1 were 7.9 k of a workgroup's 23 k cycles under load (profiles/r04_asdnet_phases.txt).  Same thread <-> pixel
// assignment and the same expressions as that prologue (kept below for callers without a statistics buffer): the same bits.
__global__ __launch_bounds__(256) void k_patch_stats(const uint8_t* __restrict__ patches, float* __restrict__ stats, int n) {
  __shared__ float red[8];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, patch = blockIdx.x;
  const uchar4 v = reinterpret_cast<const uchar4*>(patches + (size_t)patch * 1024)[t];
  const float inv255 = (float)(1.0 / 255);  // ORBextractor.cc:1125
  float x[4] = {v.x * inv255, v.y * inv255, v.z * inv255, v.w * inv255};
#pragma unroll
  for (int k = 0; k < 4; ++k) asm volatile("" : "+v"(x[k]));   // the ROUNDED products are the operands (see conv2's prologue)
  float sum = (x[0] + x[1]) + (x[2] + x[3]);
  for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
  if (lane == 0) red[wave] = sum;
  asd_syncthreads();
  const float mean = ((red[0] + red[1]) + (red[2] + red[3])) * (1.0f / 1024.0f);
  float d[4], ss = 0.f;
  for (int k = 0; k < 4; ++k) { d[k] = x[k] - mean; ss += d[k] * d[k]; }
  for (int off = 32; off >= 1; off >>= 1) ss += __shfl_xor(ss, off);
  if (lane == 0) red[4 + wave] = ss;
  asd_syncthreads();
  const float sd = sqrtf(((red[4] + red[5]) + (red[6] + red[7])) * (1.0f / 1023.0f)) + 1e-7f;  // unbiased std
  if (t == 0) { stats[2 * (size_t)patch] = mean; stats[2 * (size_t)patch + 1] = sd; }
}

// PAIR (two-piece form only): activations travel between the layers as the fp16 pieces themselves -- per pixel and group of eight
// channels 16 B of h then 16 B of l, [pixel][c/8][h | l][8], the pieces of kActScale * x: the LDS band's own layout and the same
// 4 B per element as f32.  The producer's epilogue splits each value once (split4_mix); a consumer stages its band with plain 16-B
// copies instead of splitting every value of band and halo again on the vector ALU.  Identical pieces, identical products: the
// descriptors are bit-identical to the f32-activation form (tests/test_asdnet.py::test_pair_format_is_bit_identical).  The
// accumulators are kept transposed for it (weights as the MFMA's row operand): a lane then owns four consecutive couts of one pixel.
template <int CIN, int COUT, int HIN, int S, int ROWS, int WM, int WN, int PP, bool FUSE1 = false, int NP = 3, bool PAIR = false>
__device__ __forceinline__ void conv_x3_tile(const int bid, const void* __restrict__ in_, const uint8_t* __restrict__ wimg,
                                                         const float* __restrict__ bias, float* __restrict__ out,
                                                         const float* __restrict__ w1, const float* __restrict__ b1, int n,
                                                         unsigned long long* __restrict__ stamps, float in_scale, float out_scale,
                                                         const float* __restrict__ stats = nullptr) {
  // NP = 3: bf16 pieces, six products (in_scale = out_scale = 1); NP = 2: fp16 pieces of x * in_scale, three products, the
  // accumulators are multiplied by out_scale = 1 / (in_scale * the layer's weight scale) in the epilogue (powers of two: exact)
  using C = X3Cfg<CIN, COUT, HIN, S, ROWS, WM, WN, PP, NP>;
  constexpr int NTH = C::NTH, MT = C::MT, NCW = C::NCW;
  static_assert(!PAIR || (NP == 2 && ASD_X3_S16), "the pair format belongs to the two-piece form on the 16x16x32 shape");
  extern __shared__ __attribute__((aligned(16))) uint8_t smem_b[];
  // Inside the persistent launch this body is a loop body.  Everything below that depends only on the lane (operand offsets, the
  // weight stream's addresses) is loop invariant, and hoisted out of the tile loop it stays live across the whole body: 93 -> 240
  // VGPRs for conv3, spills in conv2.  The thread index is therefore made opaque per tile, so each tile recomputes what it needs.
  int t = threadIdx.x;
  asm volatile("" : "+v"(t));
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int h = lane >> 5, li = lane & 31;
  constexpr int BANDS = C::HO / ROWS;
  const int patch = (bid / BANDS) * PP, band = bid % BANDS;
  const int r0 = band * ROWS;
  unsigned long long st_t0 = 0, st_t1 = 0;   // diagnostic stamps (stamps == nullptr in every product launch)
  if (stamps) st_t0 = st_t1 = __builtin_amdgcn_s_memtime();
#define X3_STAMP(k) do { if (stamps && t == 0) stamps[16 * bid + (k)] = __builtin_amdgcn_s_memtime() - st_t0; } while (0)

  // MFMA shape: 32x32x16 -> lane = (k-half h, row/col li of 32), one sub-tile per 32x32 tile; 16x16x32 -> lane = (k-group of
  // four, row/col of 16), SUB = 2 sub-tiles per tile side
  constexpr bool S16 = ASD_X3_S16;
  constexpr int SUB = S16 ? 2 : 1, KG = C::KCH / 8, TW = 32 / SUB;
  const int kg = S16 ? lane >> 4 : h, lr = S16 ? lane & 15 : li;
  // ---- B operand stream: chunk c = tap * NC16 + c16, this lane's 8 k values of (piece, k-group kg, cout)
  const uint8_t* wl = wimg + ((size_t)kg * COUT + wn * NCW + lr) * 16;
  constexpr int NB = NCW / TW;  // B sub-tiles of this wave
  auto load_b = [&](int c, u32x4 (&b)[NB][NP]) {
    const uint8_t* wc = wl + (size_t)((ASD_X3_ABL & 1) ? (c & 1) : c) * C::CHUNKB;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int p = 0; p < NP; ++p) b[nb][p] = *reinterpret_cast<const u32x4*>(wc + (size_t)p * KG * COUT * 16 + nb * TW * 16);
  };
  // B operands run RB - 1 chunks ahead through a register ring of RB slots; the chunk loop is unrolled RB times so that the
  // slot of every chunk is a compile-time constant (no register copies)
  constexpr int RB = C::NCHUNK % ASD_X3_RB == 0 ? ASD_X3_RB : 3;
  static_assert(C::NCHUNK % RB == 0, "chunk count");
  u32x4 br[RB][NB][NP];
#pragma unroll
  for (int d = 0; d < RB - 1; ++d) load_b(d, br[d]);

  // ---- stage the zero-padded input band(s), split into bf16 terms.  Item = (pixel, 8-cin group); a thread keeps its
  // cin group; loads are issued SBATCH items at a time so that the HBM latency is paid once per batch, not per item
  if constexpr (FUSE1) {
    // conv2 computes its own input (input_norm + conv1 + BN + ReLU, ASDNet.py:334-336, 360-365) from the raw u8 patch, exactly
    // as K1's FUSE1 path does, and splits it on the way into the band.  Extra LDS behind the band: normalised input rows
    // r0-2 .. r0+ROWS+1 (34 wide, zero padded), conv1 weights + bias, reduction scratch
    static_assert(!FUSE1 || (CIN == 32 && HIN == 32 && S == 1 && PP == 1 && (C::NW == 4 || C::NW == 8)), "conv1 fusion is for conv2 only");
    float* pin = reinterpret_cast<float*>(smem_b + C::ACT_BYTES);  // [(ROWS+4)][36]
    float* wsh = pin + (ROWS + 4) * 36;                            // [32*9 + 32]
    float* red = wsh + 320;                                        // [8]
    const uint8_t* patches = static_cast<const uint8_t*>(in_);
    // zero padding of the normalised rows: every entry that is not a pixel of the patch (those are written by their owners below; no entry
    // is written twice, so the statistics path needs no barrier between the two)
    for (int i = t; i < (ROWS + 4) * 36; i += NTH) {
      const int j = i / 36, c = i % 36, y = r0 - 2 + j;
      if (!(c >= 1 && c <= 32 && y >= 0 && y < 32)) pin[i] = 0.f;
    }
    for (int i = t; i < 288; i += NTH) wsh[i] = w1[i];
    if (t < 32) wsh[288 + t] = b1[t];
    // the 1024 pixels of the patch sit in the first four waves (4 per lane); further waves only help with conv1 below
    const bool ld = C::NW == 4 || wave < 4;
    const uchar4 v = ld ? reinterpret_cast<const uchar4*>(patches + (size_t)patch * 1024)[t] : make_uchar4(0, 0, 0, 0);
    const float inv255 = (float)(1.0 / 255);  // ORBextractor.cc:1125
    float x[4] = {v.x * inv255, v.y * inv255, v.z * inv255, v.w * inv255};
    // the ROUNDED products are the operands of everything below: a constant patch must give mean == pixel and x - mean == 0
    // exactly (as the reference's double-accumulated mean does); contracted into fma(v, 1/255, -mean) the difference would be
    // the product's rounding error, which (x - mean) / (0 + 1e-7) turns into an O(0.1) input.  (HIP's __fmul_rn is a plain
    // multiply, so the barrier is an empty asm on the value.)
#pragma unroll
    for (int k = 0; k < 4; ++k) asm volatile("" : "+v"(x[k]));
    float d[4], sd;
    if (stats) {   // mean and std + 1e-7 of the patch from k_patch_stats (round 5): no reduction, no barrier in front of conv1
      const float mean = stats[2 * (size_t)patch];
      sd = stats[2 * (size_t)patch + 1];
      for (int k = 0; k < 4; ++k) d[k] = x[k] - mean;
      X3_STAMP(9);
    } else {
      float sum = (x[0] + x[1]) + (x[2] + x[3]);
      for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
      X3_STAMP(9);
      if (lane == 0 && ld) red[wave] = sum;
      asd_syncthreads();
      const float mean = ((red[0] + red[1]) + (red[2] + red[3])) * (1.0f / 1024.0f);
      float ss = 0.f;
      for (int k = 0; k < 4; ++k) { d[k] = x[k] - mean; ss += d[k] * d[k]; }
      for (int off = 32; off >= 1; off >>= 1) ss += __shfl_xor(ss, off);
      if (lane == 0 && ld) red[4 + wave] = ss;
      asd_syncthreads();
      sd = sqrtf(((red[4] + red[5]) + (red[6] + red[7])) * (1.0f / 1023.0f)) + 1e-7f;  // unbiased std
    }
    if (ld) {
      const int idx = t * 4, y = idx >> 5, x0 = idx & 31;  // this thread's 4 pixels sit in row y
      const int j = y - (r0 - 2);
      if (j >= 0 && j < ROWS + 4)
        for (int k = 0; k < 4; ++k) pin[j * 36 + x0 + k + 1] = d[k] / sd;
    }
    asd_syncthreads();
    if (stamps) st_t1 = __builtin_amdgcn_s_memtime();
    // conv1 output for rows r0-1 .. r0+ROWS, cols -1 .. 32 (zero outside the 32x32 map: conv2's padding).
    // item = (pixel, octet of 8 couts); a thread keeps its octet and holds its 72 weights + 8 biases in registers.
    // Same f32 operation order per output as K1 (bias, then the nine taps in order)
    constexpr int NPIX = C::INROWS * C::INCOLS;
    if (!(ASD_X3_ABL & 2)) {
      const int q = t & 3;
      float wq[8][9], bq1[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        bq1[c] = wsh[288 + 8 * q + c];
#pragma unroll
        for (int k = 0; k < 9; ++k) wq[c][k] = wsh[(8 * q + c) * 9 + k];
      }
      for (int item = t; item < NPIX * 4; item += NTH) {
        const int pix = item >> 2;
        const int i = pix % C::INCOLS, j = pix / C::INCOLS;
        const int oy = r0 - 1 + j, ox = i - 1;
        f32x4 ra = {0.f, 0.f, 0.f, 0.f}, rb = {0.f, 0.f, 0.f, 0.f};
        if (oy >= 0 && oy < 32 && ox >= 0 && ox < 32) {
          float a[9];
#pragma unroll
          for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) a[ky * 3 + kx] = pin[(j + ky) * 36 + ox + kx];  // pin row j <-> image row oy-1
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            float acc1 = bq1[c];
#pragma unroll
            for (int k = 0; k < ((ASD_X3_ABL & 32) ? 1 : 9); ++k) acc1 += a[k] * wq[c][k];
            const float r = acc1 > 0.f ? acc1 : 0.f;
            if (c < 4) ra[c] = r; else rb[c - 4] = r;
          }
        }
        u32x4 ph, pm, pl;
        uint8_t* dst = smem_b + pix * C::PIXB + q * C::GSTR;
        if constexpr (NP == 2) {
          split8_f16(ra, rb, in_scale, ph, pl);
          *reinterpret_cast<u32x4*>(dst) = ph;
          *reinterpret_cast<u32x4*>(dst + C::PSTR) = pl;
        } else {
          if constexpr ((ASD_X3_ABL & 64) != 0) { ph = __builtin_bit_cast(u32x4, ra); pm = __builtin_bit_cast(u32x4, rb); pl = ph; }
          else split8(ra, rb, ph, pm, pl);
          *reinterpret_cast<u32x4*>(dst) = ph;
          *reinterpret_cast<u32x4*>(dst + 16) = pm;
          *reinterpret_cast<u32x4*>(dst + 32) = pl;
        }
      }
    }
  } else if constexpr (PAIR) {
    // the band is a copy: item = (pixel, group of eight channels) = 32 B of pieces; pixels outside the map are zeros
    const uint8_t* inb = static_cast<const uint8_t*>(in_) + (size_t)patch * HIN * HIN * CIN * 4;
    constexpr int C8 = CIN / 8;
    constexpr int NPIXB = C::INROWS * C::INCOLS;
    static_assert(NTH % C8 == 0, "a thread keeps its channel group");
    constexpr int PSTEP = NTH / C8;
    constexpr int NROUND = (PP * NPIXB + PSTEP - 1) / PSTEP;
    constexpr int SBATCH = 8;
    const int c8 = t % C8, pix0 = t / C8;
    for (int rb = 0; rb < ((ASD_X3_ABL & 2) ? 0 : NROUND); rb += SBATCH) {
      u32x4 v0[SBATCH], v1[SBATCH];
#pragma unroll
      for (int b = 0; b < SBATCH; ++b) {
        const int pixg = pix0 + (rb + b) * PSTEP;
        const int pp = pixg / NPIXB, pix = pixg % NPIXB;
        const int i = pix % C::INCOLS, j = pix / C::INCOLS;
        const int iy = r0 * S - 1 + j, ix = i - 1;
        v0[b] = u32x4{0, 0, 0, 0}; v1[b] = u32x4{0, 0, 0, 0};
        if (rb + b < NROUND && pixg < PP * NPIXB && iy >= 0 && iy < HIN && ix >= 0 && ix < HIN && patch + pp < n) {
          const uint8_t* src = inb + ((size_t)pp * HIN * HIN * CIN + ((size_t)iy * HIN + ix) * CIN + c8 * 8) * 4;
          v0[b] = *reinterpret_cast<const u32x4*>(src);
          v1[b] = *reinterpret_cast<const u32x4*>(src + 16);
        }
      }
#pragma unroll
      for (int b = 0; b < SBATCH; ++b) {
        const int pixg = pix0 + (rb + b) * PSTEP;
        if (rb + b < NROUND && pixg < PP * NPIXB) {
          uint8_t* dst = smem_b + (pixg / NPIXB) * C::ACT_BYTES + (pixg % NPIXB) * C::PIXB + c8 * C::GSTR;
          *reinterpret_cast<u32x4*>(dst) = v0[b];
          *reinterpret_cast<u32x4*>(dst + C::PSTR) = v1[b];
        }
      }
    }
  } else
  {
    const float* inp = static_cast<const float*>(in_) + (size_t)patch * HIN * HIN * CIN;
    constexpr int C8 = CIN / 8;
    constexpr int NPIXB = C::INROWS * C::INCOLS;
    static_assert(NTH % C8 == 0, "a thread keeps its cin group");
    constexpr int PSTEP = NTH / C8;                          // pixels per round
    constexpr int NROUND = (PP * NPIXB + PSTEP - 1) / PSTEP;
    constexpr int SBATCH = 8;
    const int c8 = t % C8, pix0 = t / C8;
    for (int rb = 0; rb < ((ASD_X3_ABL & 2) ? 0 : NROUND); rb += SBATCH) {
      f32x4 v0[SBATCH], v1[SBATCH];
#pragma unroll
      for (int b = 0; b < SBATCH; ++b) {
        const int pixg = pix0 + (rb + b) * PSTEP;
        const int pp = pixg / NPIXB, pix = pixg % NPIXB;
        const int i = pix % C::INCOLS, j = pix / C::INCOLS;
        const int iy = r0 * S - 1 + j, ix = i - 1;
        v0[b] = f32x4{0.f, 0.f, 0.f, 0.f}; v1[b] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (rb + b < NROUND && pixg < PP * NPIXB && iy >= 0 && iy < HIN && ix >= 0 && ix < HIN && patch + pp < n) {
          const float* src = inp + (size_t)pp * HIN * HIN * CIN + ((size_t)iy * HIN + ix) * CIN + c8 * 8;
          v0[b] = *reinterpret_cast<const f32x4*>(src);
          v1[b] = *reinterpret_cast<const f32x4*>(src + 4);
        }
      }
#pragma unroll
      for (int b = 0; b < SBATCH; ++b) {
        const int pixg = pix0 + (rb + b) * PSTEP;
        if (rb + b < NROUND && pixg < PP * NPIXB) {
          u32x4 ph, pm, pl;
          uint8_t* dst = smem_b + (pixg / NPIXB) * C::ACT_BYTES + (pixg % NPIXB) * C::PIXB + c8 * C::GSTR;
          if constexpr (NP == 2) {
            split8_f16(v0[b], v1[b], in_scale, ph, pl);
            *reinterpret_cast<u32x4*>(dst) = ph;
            *reinterpret_cast<u32x4*>(dst + C::PSTR) = pl;
          } else {
            split8(v0[b], v1[b], ph, pm, pl);
            *reinterpret_cast<u32x4*>(dst) = ph;
            *reinterpret_cast<u32x4*>(dst + 16) = pm;
            *reinterpret_cast<u32x4*>(dst + 32) = pl;
          }
        }
      }
    }
  }
  X3_STAMP(12);
  asd_syncthreads();
  X3_STAMP(13);

  // accumulators: [A sub-tile][B sub-tile]; a sub-tile is 32x32 (16 floats per lane) or 16x16 (4 floats per lane)
  constexpr int NA = MT * SUB, AR = S16 ? 4 : 16;
  typedef float accv __attribute__((ext_vector_type(S16 ? 4 : 16)));
  accv acc[NA][NB];
  for (int ma = 0; ma < NA; ++ma)
    for (int nb = 0; nb < NB; ++nb)
      for (int r = 0; r < AR; ++r) acc[ma][nb][r] = 0.f;
  // pixel (index within the workgroup's M_WG output pixels) of row / column r16 of sub-tile sg.  Two-piece form on the 16x16x32 shape: the
  // order inside a sub-tile is chosen so that every ds_read_b128 lane group reads sixteen different 16-B slots (X3Cfg::PLANAR's comment;
  // checked for every tap, chunk and piece by tests/test_asdnet.py::test_lds_reads_are_conflict_free over the same formulas):
  //   stride 1, rows of >= 16 pixels: lanes 0-3, 12-15 take the even pixels of the sixteen, lanes 4-11 the odd ones;
  //   stride 2 (planar pieces): the natural order; with rows of eight pixels lanes 0-3, 12-15 take the first row, lanes 4-11 the second;
  //   stride 1, rows of eight pixels (conv6): a sub-tile is eight rows x two columns, the even column on lanes 0-3, 12-15.
  auto tile_pixel = [&](int sg, int r16) -> int {
    if constexpr (NP == 2 && S16) {
      if constexpr (S == 1 && C::HO >= 16) return sg * 16 + (r16 < 4 ? 2 * r16 : r16 < 12 ? 2 * r16 - 7 : 2 * r16 - 16);
      else if constexpr (S == 2 && C::HO >= 16) return sg * 16 + r16;
      else if constexpr (S == 2) return sg * 16 + (r16 < 4 ? r16 : r16 < 12 ? r16 + 4 : r16 - 8);
      else {
        static_assert(S == 2 || C::HO >= 16 || (PP == 1 && C::HO == 8 && ROWS == 8), "eight rows x two columns per sub-tile");
        const int j = r16 < 4 ? r16 : r16 < 12 ? r16 - 4 : r16 - 8;
        return j * C::HO + 2 * sg + (r16 >= 4 && r16 < 12 ? 1 : 0);
      }
    } else return sg * TW + r16;
  };
  int abase[NA];
  for (int ma = 0; ma < NA; ++ma) {
    const int m = tile_pixel((wm * MT * 32 + ma * TW) / TW, lr);
    const int pp = PP > 1 ? m / C::M_PATCH : 0, mm = m - pp * C::M_PATCH;
    const int rr = mm / C::HO, ox = mm % C::HO;
    abase[ma] = pp * C::ACT_BYTES + ((rr * S) * C::INCOLS + ox * S) * C::PIXB + kg * C::GSTR;
  }
  auto chunk_off = [&](int c) {
    const int tap = c / C::NC16, c16 = c % C::NC16;
    return ((tap / 3) * C::INCOLS + (tap % 3)) * C::PIXB + c16 * (KG * C::GSTR);
  };
  auto load_a = [&](int off, u32x4 (&a)[NP]) {
#pragma unroll
    for (int p = 0; p < NP; ++p) a[p] = *reinterpret_cast<const u32x4*>(smem_b + off + p * C::PSTR);
  };
  auto bf = [](const u32x4& v) { return __builtin_bit_cast(bf16x8, v); };
  auto hf = [](const u32x4& v) { return __builtin_bit_cast(f16x8, v); };
  auto mma = [&](const u32x4& x, const u32x4& y, accv& c) {
    if constexpr ((ASD_X3_ABL & 8) != 0) { asm volatile("" ::"v"(x), "v"(y)); }  // tuning: operands fetched, no MFMA
#ifdef ASD_PAIR_NOTRANS   // timing experiment only (results are wrong): the pair epilogue behind untransposed accumulators
    else if constexpr (PAIR) c = __builtin_amdgcn_mfma_f32_16x16x32_f16(hf(x), hf(y), c, 0, 0, 0);
#endif
    else if constexpr (PAIR) c = __builtin_amdgcn_mfma_f32_16x16x32_f16(hf(y), hf(x), c, 0, 0, 0);   // transposed: rows = couts, columns = pixels
    else if constexpr (NP == 2 && S16) c = __builtin_amdgcn_mfma_f32_16x16x32_f16(hf(x), hf(y), c, 0, 0, 0);
    else if constexpr (NP == 2) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(hf(x), hf(y), c, 0, 0, 0);
    else if constexpr (S16) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf(x), bf(y), c, 0, 0, 0);
    else c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(x), bf(y), c, 0, 0, 0);
  };

  // A operands run PD sub-tiles ahead of the MFMAs through a register ring of PD + 1 slots; a loop iteration covers RB chunks x
  // NA sub-tiles, a multiple of the ring size, so every sub-tile's slot is a compile-time constant
  constexpr int PD = ASD_X3_PD, RS = PD + 1;
  static_assert((RB * NA) % RS == 0, "ring slots must be static");
  u32x4 ar[RS][NP];
#pragma unroll
  for (int q = 0; q < PD; ++q) load_a(abase[q % NA] + chunk_off(q / NA), ar[q]);
  // diagnostic only (stamps == nullptr in every product launch): shader clock and 100 MHz wall clock around the MFMA loop
  unsigned long long st_c = 0, st_r = 0;
  if (stamps) { st_c = __builtin_amdgcn_s_memtime(); st_r = __builtin_amdgcn_s_memrealtime(); }
  for (int c0 = 0; c0 < C::NCHUNK; c0 += RB) {
    int offs[RB + PD / NA + 2];
#pragma unroll
    for (int u = 0; u < RB + PD / NA + 2; ++u) offs[u] = chunk_off(c0 + u < C::NCHUNK ? c0 + u : C::NCHUNK - 1);
#pragma unroll
    for (int u = 0; u < RB; ++u) {
      const int c = c0 + u;
      // past the end: a redundant re-read of the last chunk instead of a branch
      load_b(c + RB - 1 < C::NCHUNK ? c + RB - 1 : C::NCHUNK - 1, br[(u + RB - 1) % RB]);
      __builtin_amdgcn_sched_group_barrier(0x020, NP * NB, 0);
      const u32x4 (&bc)[NB][NP] = br[u];
#pragma unroll
      for (int ma = 0; ma < NA; ++ma) {
        const int q = u * NA + ma;            // sub-tile index within the iteration
        const int qn = q + PD;                // the sub-tile fetched now (past the end: re-reads the last chunk, unused)
        load_a(abase[qn % NA] + offs[qn / NA], ar[qn % RS]);
        const u32x4 (&ac)[NP] = ar[q % RS];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          if constexpr (NP == 2) {   // smallest products first: (l,h) (h,l) (h,h)
            mma(ac[1], bc[nb][0], acc[ma][nb]);
            mma(ac[0], bc[nb][1], acc[ma][nb]);
            mma(ac[0], bc[nb][0], acc[ma][nb]);
          } else {                   // (l,h) (h,l) (m,m) (m,h) (h,m) (h,h)
            mma(ac[2], bc[nb][0], acc[ma][nb]);
            mma(ac[0], bc[nb][2], acc[ma][nb]);
            mma(ac[1], bc[nb][1], acc[ma][nb]);
            mma(ac[1], bc[nb][0], acc[ma][nb]);
            mma(ac[0], bc[nb][1], acc[ma][nb]);
            mma(ac[0], bc[nb][0], acc[ma][nb]);
          }
        }
        // issue order within the sub-tile: one operand read behind every 2 * NB MFMAs
#pragma unroll
        for (int p = 0; p < NP; ++p) {   // NP (NP + 1) / 2 products per pair of sub-tiles, NP operand reads: spread evenly
          __builtin_amdgcn_sched_group_barrier(0x008, (NP * (NP + 1) / 2) * NB / NP, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
      }
    }
  }

  if (stamps) {
    asm volatile("s_nop 0" ::"v"(acc[NA - 1][NB - 1][0]));  // the last MFMA has retired before the closing stamp
    const unsigned long long e_c = __builtin_amdgcn_s_memtime(), e_r = __builtin_amdgcn_s_memrealtime();
    if (t == 0) { stamps[16 * bid] = e_c - st_c; stamps[16 * bid + 1] = e_r - st_r; stamps[16 * bid + 2] = st_t1 - st_t0; stamps[16 * bid + 3] = st_c - st_t1; stamps[16 * bid + 6] = st_t0; }
  }
  // ---- epilogue: bias (folded BN) + ReLU, NHWC f32 store.  Lane owns one cout column and AR pixel rows of each sub-tile
  // (32x32: rows (r & 3) + 8 (r >> 2) + 4 h; 16x16: rows 4 kg + r)
  if constexpr (PAIR) {
    // lane = (pixel lr of the sub-tile, couts 4 kg .. 4 kg + 3): bias, ReLU (a NaN passes), the split of kActScale * v, a swap with
    // the neighbouring quarter-group and one 16-B store into [pixel][cout / 8][h | l][8]
    uint8_t* opb = reinterpret_cast<uint8_t*>(out) + ((size_t)patch * C::HO + r0) * C::HO * COUT * 4;
    for (int nb = 0; nb < NB; ++nb) {
      const int co0 = wn * NCW + nb * TW + 4 * kg;
      const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + co0);
      for (int ma = 0; ma < NA; ++ma) {
        const int m0 = wm * MT * 32 + ma * TW;
        const int pp = PP > 1 ? m0 / C::M_PATCH : 0;
        if (PP > 1 && patch + pp >= n) continue;
        const int m = tile_pixel(m0 / TW, lr);
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float x = acc[ma][nb][r] * out_scale + bv[r]; v[r] = x < 0.f ? 0.f : x; }
        uint32_t hh[2], ll[2];
        split4_mix(v[0], v[1], v[2], v[3], kActScale, hh, ll);
        // v_permlane16_swap: the odd 16-lane rows of the first register trade places with the even rows of the second.  Lanes kg and
        // kg ^ 1 hold the two halves of one group of eight couts; after the swap an even-kg lane holds the group's whole h piece
        // (its own half, the partner's half), an odd-kg lane the whole l piece: one 16-B store per lane, 64 contiguous bytes per pixel
        const auto s0 = __builtin_amdgcn_permlane16_swap(hh[0], ll[0], false, false);
        const auto s1 = __builtin_amdgcn_permlane16_swap(hh[1], ll[1], false, false);
        uint8_t* o = opb + ((size_t)pp * C::HO * C::HO + (size_t)(m - pp * C::M_PATCH)) * COUT * 4 + (co0 >> 3) * 32 + (kg & 1) * 16;
        if (!(ASD_X3_ABL & 4) || hh[0] == 0x12345u) *reinterpret_cast<u32x4*>(o) = u32x4{s0[0], s1[0], s0[1], s1[1]};
      }
    }
  } else {
  float* op = out + ((size_t)patch * C::HO + r0) * C::HO * COUT;
  for (int nb = 0; nb < NB; ++nb) {
    const int co = wn * NCW + nb * TW + lr;
    const float bv = bias[co];
    for (int ma = 0; ma < NA; ++ma) {
      const int m0 = wm * MT * 32 + ma * TW;
      const int pp = PP > 1 ? m0 / C::M_PATCH : 0;  // a sub-tile never straddles two patches
      if (PP > 1 && patch + pp >= n) continue;
      for (int r = 0; r < AR; ++r) {
        const int m = S16 ? tile_pixel(m0 / TW, 4 * kg + r) : m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const size_t o = (size_t)pp * C::HO * C::HO * COUT + (size_t)(m - pp * C::M_PATCH) * COUT + co;
        const float v = (NP == 2 ? acc[ma][nb][r] * out_scale : acc[ma][nb][r]) + bv;
        // ReLU; the fp16 form lets a NaN through (an activation beyond fp16's range must reach the descriptor, not become 0)
        if (!(ASD_X3_ABL & 4) || v == 12345.f) op[o] = NP == 2 ? (v < 0.f ? 0.f : v) : (v > 0.f ? v : 0.f);
      }
    }
  }
  }
  if (stamps && t == 0) { const unsigned long long e = __builtin_amdgcn_s_memtime(); stamps[16 * bid + 5] = e - st_t0; stamps[16 * bid + 7] = e; }
}

#ifndef ASD_L2_MINWG
#define ASD_L2_MINWG 3   // conv2 (two-piece form): workgroups per CU the register allocation is held to
#endif
template <int CIN, int COUT, int HIN, int S, int ROWS, int WM, int WN, int PP, bool FUSE1 = false, int NP = 3, bool PAIR = false>
__global__ __launch_bounds__(64 * WM * WN, (FUSE1 && NP == 2) ? ASD_L2_MINWG : 1) void k_conv_x3(const void* __restrict__ in_, const uint8_t* __restrict__ wimg,
                                                         const float* __restrict__ bias, float* __restrict__ out,
                                                         const float* __restrict__ w1, const float* __restrict__ b1, int n,
                                                         unsigned long long* __restrict__ stamps, float in_scale, float out_scale,
                                                         const float* __restrict__ stats) {
  // conv3 reads the 262 MB conv2 has just written -- a little more than the memory-side cache (256 MB) holds.  It walks the patches from the LAST one written to
  // the first, so that what the cache still holds is read before it is evicted (in the order of writing, LRU evicts every line just before it is read): 101 -> 98 us.
  // The layers behind it read 131 MB or less: any order.
  const int bid = (S == 2 && CIN == 32) ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x;
  conv_x3_tile<CIN, COUT, HIN, S, ROWS, WM, WN, PP, FUSE1, NP, PAIR>(bid, in_, wimg, bias, out, w1, b1, n, stamps, in_scale, out_scale, stats);
}

// ------------------------------------------------------------------------------------------
// K2: last layer, 8x8 valid conv == GEMM [n x 8192] * [8192 x 128], split-K over gridDim.y.
// One workgroup = 32 patches x 128 couts x (8192 / SK) k; wave w owns couts [32w, 32w+32), so
// every weight element is used by exactly one wave: B goes global -> VGPR, only A through LDS.
// ------------------------------------------------------------------------------------------
constexpr int FC_SK = 16;    // split-K factor: (n / 32) x FC_SK workgroups (8 / 32 measured: 48 + 6 / 40 + 9 us against 37 + 7)
constexpr int FC_KCH = 128;  // k-chunk staged in LDS per step
constexpr int FC_MT = 1;  // 32-patch tiles per workgroup (2 was measured: no gain over 1, 51 us either way)
__global__ __launch_bounds__(256) void k_fc_mfma(const float* __restrict__ act, const float* __restrict__ wimg,
                                                 float* __restrict__ part, int n, int npad) {
  __shared__ __attribute__((aligned(16))) float sa[FC_MT * 32 * (FC_KCH + 4)];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int h = lane >> 5, li = lane & 31;
  const int p0 = blockIdx.x * 32 * FC_MT, sk = blockIdx.y;
  constexpr int KPER = 8192 / FC_SK;
  f32x16 acc[FC_MT];
  for (int m = 0; m < FC_MT; ++m)
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
  for (int st = 0; st < KPER / FC_KCH; ++st) {
    const int k0 = sk * KPER + st * FC_KCH;
    asd_syncthreads();
    for (int idx = t; idx < FC_MT * 32 * (FC_KCH / 4); idx += 256) {
      const int c4 = idx % (FC_KCH / 4), row = idx / (FC_KCH / 4);
      int p = p0 + row;
      if (p >= n) p = n - 1;
      *reinterpret_cast<f32x4*>(sa + row * (FC_KCH + 4) + c4 * 4) =
          *reinterpret_cast<const f32x4*>(act + (size_t)p * 8192 + k0 + c4 * 4);
    }
    asd_syncthreads();
    const float* wb = wimg + ((size_t)(k0 / 8) * 2 + h) * 128 * 4 + (wave * 32 + li) * 4;
#pragma unroll 4
    for (int c8 = 0; c8 < FC_KCH / 8; ++c8) {
      const f32x4 b = *reinterpret_cast<const f32x4*>(wb + (size_t)c8 * 2 * 128 * 4);
      f32x4 a[FC_MT];
#pragma unroll
      for (int m = 0; m < FC_MT; ++m) a[m] = *reinterpret_cast<const f32x4*>(sa + (m * 32 + li) * (FC_KCH + 4) + c8 * 8 + 4 * h);
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int m = 0; m < FC_MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m][jj], b[jj], acc[m], 0, 0, 0);
    }
  }
  for (int m = 0; m < FC_MT; ++m) {
    if (p0 + m * 32 >= npad) break;
    float* op = part + ((size_t)sk * npad + p0 + m * 32) * 128;
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
      op[(size_t)row * 128 + wave * 32 + li] = acc[m][r];
    }
  }
}

// K2s: the same GEMM with split operands on the bf16 matrix pipe (see K1s).  One workgroup = 64 patches x 128 couts x
// (8192 / FC_SK) k; the activations go through LDS in slabs of 128 k (one pixel position: 64 rows x 784 B, split while
// staged), wave w owns couts [32w, 32w+32) as two 16-wide sub-tiles and all four 16-patch sub-tiles; B operands (the split
// weight image [k/32][piece][k-group][cout][8]) stream from global memory through a 3-slot register ring, two chunks ahead,
// across slab boundaries.  MFMA shape 16x16x32 only.  Partials land in the same [FC_SK][npad][128] buffer K3 sums.
constexpr int FCS_MP = 64;              // patches per workgroup
constexpr int FCS_SLAB = 128;           // k per LDS slab
constexpr int FCS_ROWB = FCS_SLAB * 6 + 16;
__global__ __launch_bounds__(256) void k_fc_x3(const float* __restrict__ act, const uint8_t* __restrict__ wimg,
                                               float* __restrict__ part, int n, int npad) {
  __shared__ __attribute__((aligned(16))) uint8_t sa[FCS_MP * FCS_ROWB];
  typedef float f4 __attribute__((ext_vector_type(4)));
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int kg = lane >> 4, lr = lane & 15;
  const int p0 = blockIdx.x * FCS_MP, sk = blockIdx.y;
  constexpr int KPER = 8192 / FC_SK, NSLAB = KPER / FCS_SLAB, NCH = KPER / 32;  // chunks of 32 k per workgroup
  static_assert(KPER % FCS_SLAB == 0 && NCH >= 3, "split-K slice");
  constexpr int CHUNKB = 32 * 6 * 128;
  const uint8_t* wl = wimg + (size_t)(sk * (KPER / 32)) * CHUNKB + ((size_t)kg * 128 + wave * 32 + lr) * 16;
  auto load_b = [&](int c, u32x4 (&b)[2][3]) {
    const uint8_t* wc = wl + (size_t)c * CHUNKB;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int p = 0; p < 3; ++p) b[nb][p] = *reinterpret_cast<const u32x4*>(wc + (size_t)p * 4 * 128 * 16 + nb * 256);
  };
  auto bf = [](const u32x4& v) { return __builtin_bit_cast(bf16x8, v); };
  u32x4 br[3][2][3];
  load_b(0, br[0]);
  load_b(1, br[1]);
  f4 acc[4][2];
  for (int ma = 0; ma < 4; ++ma)
    for (int nb = 0; nb < 2; ++nb)
      for (int r = 0; r < 4; ++r) acc[ma][nb][r] = 0.f;
  int c = 0;  // chunk index within this workgroup's k range
  for (int slab = 0; slab < NSLAB; ++slab) {
    const int k0 = sk * KPER + slab * FCS_SLAB;
    asd_syncthreads();  // the previous slab has been consumed
    {
      // 64 rows x 16 groups of 8 k: four items per thread, loads first
      f32x4 v0[4], v1[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int item = t + i * 256, row = item >> 4, g = item & 15;
        int p = p0 + row;
        if (p >= n) p = n - 1;
        const float* src = act + (size_t)p * 8192 + k0 + g * 8;
        v0[i] = *reinterpret_cast<const f32x4*>(src);
        v1[i] = *reinterpret_cast<const f32x4*>(src + 4);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int item = t + i * 256, row = item >> 4, g = item & 15;
        u32x4 ph, pm, pl;
        split8(v0[i], v1[i], ph, pm, pl);
        uint8_t* dst = sa + row * FCS_ROWB + g * 48;
        *reinterpret_cast<u32x4*>(dst) = ph;
        *reinterpret_cast<u32x4*>(dst + 16) = pm;
        *reinterpret_cast<u32x4*>(dst + 32) = pl;
      }
    }
    asd_syncthreads();
#pragma unroll
    for (int cs = 0; cs < FCS_SLAB / 32; ++cs, ++c) {
      // ring slot of chunk c is c % 3; NSLAB * 4 chunks in all, the slot pattern repeats every 3 chunks, so index by (c % 3)
      // through a small switch that keeps every slot a static register
      auto step = [&](u32x4 (&bc)[2][3], u32x4 (&bn)[2][3]) {
        load_b(c + 2 < NCH ? c + 2 : NCH - 1, bn);  // past the end: a redundant re-read instead of a branch
#pragma unroll
        for (int ma = 0; ma < 4; ++ma) {
          u32x4 a[3];
          const uint8_t* ap = sa + (ma * 16 + lr) * FCS_ROWB + (cs * 4 + kg) * 48;
#pragma unroll
          for (int p = 0; p < 3; ++p) a[p] = *reinterpret_cast<const u32x4*>(ap + p * 16);
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) {
            acc[ma][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf(a[2]), bf(bc[nb][0]), acc[ma][nb], 0, 0, 0);
            acc[ma][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf(a[0]), bf(bc[nb][2]), acc[ma][nb], 0, 0, 0);
            acc[ma][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf(a[1]), bf(bc[nb][1]), acc[ma][nb], 0, 0, 0);
            acc[ma][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf(a[1]), bf(bc[nb][0]), acc[ma][nb], 0, 0, 0);
            acc[ma][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf(a[0]), bf(bc[nb][1]), acc[ma][nb], 0, 0, 0);
            acc[ma][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf(a[0]), bf(bc[nb][0]), acc[ma][nb], 0, 0, 0);
          }
        }
      };
      const int slot = c % 3;
      if (slot == 0) step(br[0], br[2]);
      else if (slot == 1) step(br[1], br[0]);
      else step(br[2], br[1]);
    }
  }
  if (p0 >= npad) return;
  float* op = part + ((size_t)sk * npad + p0) * 128;
  for (int ma = 0; ma < 4; ++ma)
    for (int nb = 0; nb < 2; ++nb)
      for (int r = 0; r < 4; ++r) {
        const int row = ma * 16 + 4 * kg + r;
        if (p0 + row < npad) op[(size_t)row * 128 + wave * 32 + nb * 16 + lr] = acc[ma][nb][r];
      }
}

// K2p: the same GEMM on the two-piece fp16 form (three products), reading conv6's output in the PAIR format: patch p's 8192
// values are [k / 8][h | l][8] with k = pixel * 128 + channel -- exactly what conv6's epilogue wrote -- so a slab of 128 k is a
// 512-B run per patch that goes to LDS as it is (row stride 528 B: an odd multiple of 16 B).  Weight image [k/32][piece 2][k-group
// 4][cout 128][8 fp16], the folded weights times the layer's power of two (weight_scale_f16); out_scale = 1 / (kActScale * that).
constexpr int FCP_ROWB = FCS_SLAB * 4 + 16;
__global__ __launch_bounds__(256) void k_fc_x2(const uint8_t* __restrict__ act, const uint8_t* __restrict__ wimg,
                                               float* __restrict__ part, int n, int npad, float out_scale) {
  __shared__ __attribute__((aligned(16))) uint8_t sa[FCS_MP * FCP_ROWB];
  typedef float f4 __attribute__((ext_vector_type(4)));
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int kg = lane >> 4, lr = lane & 15;
  const int p0 = blockIdx.x * FCS_MP, sk = blockIdx.y;
  constexpr int KPER = 8192 / FC_SK, NSLAB = KPER / FCS_SLAB, NCH = KPER / 32;
  constexpr int CHUNKB = 32 * 4 * 128;
  const uint8_t* wl = wimg + (size_t)(sk * (KPER / 32)) * CHUNKB + ((size_t)kg * 128 + wave * 32 + lr) * 16;
  auto load_b = [&](int c, u32x4 (&b)[2][2]) {
    const uint8_t* wc = wl + (size_t)c * CHUNKB;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int p = 0; p < 2; ++p) b[nb][p] = *reinterpret_cast<const u32x4*>(wc + (size_t)p * 4 * 128 * 16 + nb * 256);
  };
  auto hf = [](const u32x4& v) { return __builtin_bit_cast(f16x8, v); };
  u32x4 br[3][2][2];
  load_b(0, br[0]);
  load_b(1, br[1]);
  f4 acc[4][2];
  for (int ma = 0; ma < 4; ++ma)
    for (int nb = 0; nb < 2; ++nb)
      for (int r = 0; r < 4; ++r) acc[ma][nb][r] = 0.f;
  int c = 0;
  for (int slab = 0; slab < NSLAB; ++slab) {
    const int k0 = sk * KPER + slab * FCS_SLAB;
    asd_syncthreads();  // the previous slab has been consumed
    {
      u32x4 v0[4], v1[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int item = t + i * 256, row = item >> 4, g = item & 15;
        int p = p0 + row;
        if (p >= n) p = n - 1;
        const uint8_t* src = act + (size_t)p * 32768 + (size_t)(k0 / 8 + g) * 32;
        v0[i] = *reinterpret_cast<const u32x4*>(src);
        v1[i] = *reinterpret_cast<const u32x4*>(src + 16);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int item = t + i * 256, row = item >> 4, g = item & 15;
        uint8_t* dst = sa + row * FCP_ROWB + g * 32;
        *reinterpret_cast<u32x4*>(dst) = v0[i];
        *reinterpret_cast<u32x4*>(dst + 16) = v1[i];
      }
    }
    asd_syncthreads();
#pragma unroll
    for (int cs = 0; cs < FCS_SLAB / 32; ++cs, ++c) {
      auto step = [&](u32x4 (&bc)[2][2], u32x4 (&bn)[2][2]) {
        load_b(c + 2 < NCH ? c + 2 : NCH - 1, bn);  // past the end: a redundant re-read instead of a branch
#pragma unroll
        for (int ma = 0; ma < 4; ++ma) {
          const uint8_t* ap = sa + (ma * 16 + lr) * FCP_ROWB + (cs * 4 + kg) * 32;
          const u32x4 ah = *reinterpret_cast<const u32x4*>(ap), al = *reinterpret_cast<const u32x4*>(ap + 16);
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) {   // smallest products first: (l,h) (h,l) (h,h)
            acc[ma][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hf(al), hf(bc[nb][0]), acc[ma][nb], 0, 0, 0);
            acc[ma][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hf(ah), hf(bc[nb][1]), acc[ma][nb], 0, 0, 0);
            acc[ma][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hf(ah), hf(bc[nb][0]), acc[ma][nb], 0, 0, 0);
          }
        }
      };
      const int slot = c % 3;
      if (slot == 0) step(br[0], br[2]);
      else if (slot == 1) step(br[1], br[0]);
      else step(br[2], br[1]);
    }
  }
  if (p0 >= npad) return;
  float* op = part + ((size_t)sk * npad + p0) * 128;
  for (int ma = 0; ma < 4; ++ma)
    for (int nb = 0; nb < 2; ++nb)
      for (int r = 0; r < 4; ++r) {
        const int row = ma * 16 + 4 * kg + r;
        if (p0 + row < npad) op[(size_t)row * 128 + wave * 32 + nb * 16 + lr] = acc[ma][nb][r] * out_scale;
      }
}

// K3: sum split-K partials in fixed order (deterministic), folded BN bias, L2Norm (Utils.py:15-22).
// calibration: max |x| over a buffer as float bits (non-negative floats order like their bit patterns; a NaN / inf sorts above)
__global__ __launch_bounds__(256) void k_absmax(const float* __restrict__ p, size_t n, unsigned* __restrict__ out) {
  unsigned m = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) m = max(m, __float_as_uint(p[i]) & 0x7fffffffu);
  for (int off = 32; off >= 1; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off));
  if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

__global__ __launch_bounds__(256) void k_l2norm(const float* __restrict__ part, const float* __restrict__ bias,
                                                float* __restrict__ desc, int n, int npad, int* __restrict__ range_flag) {
  const int lane = threadIdx.x & 63, p = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= n) return;
  float v0 = 0.f, v1 = 0.f;
  for (int sk = 0; sk < FC_SK; ++sk) {
    v0 += part[((size_t)sk * npad + p) * 128 + lane];
    v1 += part[((size_t)sk * npad + p) * 128 + lane + 64];
  }
  v0 += bias[lane];
  v1 += bias[lane + 64];
  float ss = v0 * v0 + v1 * v1;
  for (int off = 32; off >= 1; off >>= 1) ss += __shfl_xor(ss, off);
  const float norm = sqrtf(ss + 1e-10f);
  // a row that is not finite: an activation left the range of the two-piece fp16 operand form somewhere upstream (the kernels let
  // the inf / NaN through on purpose).  The flag sits in pinned host memory; the synchronising entry points turn it into ASD_ERR_RANGE.
  if (range_flag && lane == 0 && !(fabsf(ss) <= 3.0e38f)) *range_flag = 1;
  desc[(size_t)p * 128 + lane] = v0 / norm;
  desc[(size_t)p * 128 + lane + 64] = v1 / norm;
}

constexpr bool kPairOK = ASD_X3_S16 != 0;   // the pair format's transposed epilogue is written for the 16x16x32 shape
// layer configurations: <CIN, COUT, HIN, S, ROWS, WM, WN, KC, PP, RING> (L3: persistent form, no PP / RING)
#define L2_CFG 32, 32, 32, 1, 4, 4, 1, 16, 1, 3
#define L3_CFG 32, 64, 32, 2, 4, 2, 2, 16
#define L4_CFG 64, 64, 16, 1, 8, 4, 1, 32, 1, 2
#define L5_CFG 64, 128, 16, 2, 4, 1, 4, 16, 1, 3
#define L6_CFG 128, 128, 8, 1, 8, 2, 2, 16, 1, 3
// split-operand kernels: <CIN, COUT, HIN, S, ROWS, WM, WN, PP>
// (two workgroups per CU each: one's band staging overlaps the other's MFMAs; whole-patch conv4 / two-patch conv6
// workgroups at one per CU measured 189 / 168 us against 163 / 160)
// Wave layout WM x WN (pixels x couts) of conv2 / conv3 / conv4.  A wave fetches the B operands (weights) of its own cout columns from
// L1 / L2 and the A operands (activations) of its own pixels from LDS, so fewer couts per wave mean less weight stream through the CU's
// vector-memory path and more LDS reads per MFMA.  Measured at N = 2000 (round 3, tools/build_variant.sh): conv3 as 1 x 4 (64 pixels x 16
// couts per wave, no weight fetched twice in a workgroup) 104 -> 96 us -- the default; conv2 as 2 x 2 164 against 146 and conv4 as 1 x 4 103
// against 96: there the doubled LDS traffic costs more than the halved weight stream saves.
#ifndef ASD_L2_WMN
#define ASD_L2_WMN 4, 1
#endif
#ifndef ASD_L3_WMN
#if ASD_X3_S16
#define ASD_L3_WMN 1, 4
#else
#define ASD_L3_WMN 2, 2     // (the 32x32x16 shape needs 32 couts per wave)
#endif
#endif
#ifndef ASD_L4_WMN
#define ASD_L4_WMN 2, 2
#endif
#define L2S_CFG 32, 32, 32, 1, 8, ASD_L2_WMN, 1
#ifndef ASD_L3_ROWS
#define ASD_L3_ROWS 4
#endif
#define L3S_CFG 32, 64, 32, 2, ASD_L3_ROWS, ASD_L3_WMN, 1
#define L4S_CFG 64, 64, 16, 1, 8, ASD_L4_WMN, 1
// conv5: rows per band, WM, WN.  With two fp16 pieces the whole 16 x 16 input of a patch is 79 KB of LDS, so two whole-patch workgroups
// fit a CU and every weight is fetched for 64 pixels instead of 32: 71 -> 66 us (round 2's three-piece form had room for one such
// workgroup only and was slower that way).  Measured and not kept (N = 2000): 8-wave workgroups for conv4 / conv5 / conv6 (64 pixels x 16
// couts per wave, two MFMA-issuing waves per SIMD and workgroup): 115 / 77 / 106 us against 97 / 71 / 95.
#ifndef ASD_L5_RWMN
#if ASD_X3_S16
#define ASD_L5_RWMN 8, 1, 4
#else
#define ASD_L5_RWMN 4, 1, 4     // (the 32x32x16 build keeps the 4-row bands: the whole-patch form costs it five more minutes of compile time)
#endif
#endif
#ifndef ASD_L6_WMN
#define ASD_L6_WMN 1, 4
#endif
#define L5S_CFG 64, 128, 16, 2, ASD_L5_RWMN, 1
#define L6S_CFG 128, 128, 8, 1, 8, ASD_L6_WMN, 1

template <int CIN, int COUT, int HIN, int S, int ROWS, int WM, int WN, int KC, int PP = 1, int RING = 2, bool FUSE1 = false>
hipError_t launch_conv(hipStream_t st, const void* in, const float* wimg, const float* bias, float* out, int n,
                       const float* w1 = nullptr, const float* b1 = nullptr) {
  using C = ConvCfg<CIN, COUT, HIN, S, ROWS, WM, WN, KC, PP, RING>;
  auto kern = k_conv_mfma<CIN, COUT, HIN, S, ROWS, WM, WN, KC, PP, RING, 0, FUSE1>;
  constexpr int lds = C::LDS_BYTES + (FUSE1 ? ((ROWS + 4) * 36 + 320 + 8) * 4 : 0);
  static_assert(lds <= 160 * 1024, "band + weight ring do not fit LDS");
  static AsdPerDeviceOnce attr_set;   // per instantiation; the attribute belongs to the current device (the caller selected the context's)
  int dev_ = 0;
  (void)hipGetDevice(&dev_);
  if (attr_set.need(dev_)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    attr_set.done(dev_);
  }
  hipLaunchKernelGGL(kern, dim3(((n + PP - 1) / PP) * (C::HO / ROWS)), dim3(C::NTH), lds, st, in, wimg, bias, out, w1, b1, n);
  return hipGetLastError();
}

template <int CIN, int COUT, int HIN, int S, int ROWS, int WM, int WN, int KC>
hipError_t launch_conv_p(hipStream_t st, const float* in, const float* wimg, const float* bias, float* out, int n, int num_cu) {
  using C = ConvCfg<CIN, COUT, HIN, S, ROWS, WM, WN, KC>;
  auto kern = k_conv_mfma_p<CIN, COUT, HIN, S, ROWS, WM, WN, KC>;
  constexpr int lds = (2 * C::ACT_FLOATS + 2 * C::WCHUNK) * 4;
  static_assert(lds <= 160 * 1024, "double-buffered band does not fit LDS");
  static AsdPerDeviceOnce attr_set;   // per instantiation; the attribute belongs to the current device (the caller selected the context's)
  int dev_ = 0;
  (void)hipGetDevice(&dev_);
  if (attr_set.need(dev_)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    attr_set.done(dev_);
  }
  const int ntiles = n * (C::HO / ROWS);
  const int per_cu = std::max(1, std::min(4, (160 * 1024) / lds));
  const int grid = std::min(ntiles, num_cu * per_cu);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, in, wimg, bias, out, ntiles);
  return hipGetLastError();
}

template <int CIN, int COUT, int HIN, int S, int ROWS, int WM, int WN, int PP, bool FUSE1 = false, int NP = 3, bool PAIR = false>
hipError_t launch_conv_x3(hipStream_t st, const void* in, const void* wimg, const float* bias, float* out, int n,
                          const float* w1 = nullptr, const float* b1 = nullptr, unsigned long long* stamps = nullptr, int* grid_out = nullptr,
                          float in_scale = 1.f, float out_scale = 1.f, const float* stats = nullptr) {
  using C = X3Cfg<CIN, COUT, HIN, S, ROWS, WM, WN, PP, NP>;
  auto kern = k_conv_x3<CIN, COUT, HIN, S, ROWS, WM, WN, PP, FUSE1, NP, PAIR>;
  // (padding the request to force one workgroup per CU was measured in round 2: ASDNet 0.79 -> 0.99 ms, 703 frames/s; not kept)
  constexpr int lds0 = C::LDS_BYTES + (FUSE1 ? ((ROWS + 4) * 36 + 320 + 8) * 4 : 0);
  static_assert(lds0 <= 160 * 1024, "band does not fit LDS");
  // (fewer of these workgroups per CU, to leave the tracking stream's kernels room, was measured in round 4: within noise or slower)
  const int lds = lds0;
  static AsdPerDeviceOnce attr_set;   // per instantiation; the attribute belongs to the current device (the caller selected the context's)
  int dev_ = 0;
  (void)hipGetDevice(&dev_);
  if (attr_set.need(dev_)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    attr_set.done(dev_);
  }
  const int ntiles = ((n + PP - 1) / PP) * (C::HO / ROWS);
  if (grid_out) *grid_out = ntiles;
  hipLaunchKernelGGL(kern, dim3(ntiles), dim3(C::NTH), lds, st, in, static_cast<const uint8_t*>(wimg), bias, out, w1, b1, n, stamps,
                     in_scale, out_scale, stats);
  return hipGetLastError();
}

// exact three-term bf16 split of one f32 (host twin of split8)
inline void split3_host(float x, uint16_t& h, uint16_t& m, uint16_t& l) {
  uint32_t u, r1u, r2u;
  memcpy(&u, &x, 4);
  const uint32_t hu = u & 0xffff0000u;
  float hf; memcpy(&hf, &hu, 4);
  const float r1 = x - hf;
  memcpy(&r1u, &r1, 4);
  const uint32_t mu = r1u & 0xffff0000u;
  float mf; memcpy(&mf, &mu, 4);
  const float r2 = r1 - mf;
  memcpy(&r2u, &r2, 4);
  h = (uint16_t)(hu >> 16); m = (uint16_t)(mu >> 16); l = (uint16_t)(r2u >> 16);
}

// two-term fp16 split of one (pre-scaled) f32, round to nearest (host twin of split8_f16)
inline void split2_host(float xs, uint16_t& h, uint16_t& l) {
  const _Float16 hh = (_Float16)xs;
  const _Float16 ll = (_Float16)(xs - (float)hh);
  memcpy(&h, &hh, 2);
  memcpy(&l, &ll, 2);
}

// power of two that brings the largest folded weight of a layer into [2^13, 2^14): fp16's range with a factor 4 to spare
inline float weight_scale_f16(const float* w, size_t count, const std::vector<float>& inv, size_t per_cout) {
  float mx = 0.f;
  for (size_t i = 0; i < count; ++i) mx = std::max(mx, std::fabs(w[i] * inv[i / per_cout]));
  if (!(mx > 0.f) || !std::isfinite(mx)) return 1.f;
  int e;
  (void)std::frexp(mx, &e);   // mx = f 2^e, f in [0.5, 1)
  return std::ldexp(1.f, 14 - e);
}

// split B-operand image of a 3x3 layer: [tap][cin/KCH][piece][k-group][cout][8 x 16 bit] with cin = KCH*c + 8*group + j
// (KCH = 16 for the 32x32x16 MFMA shape, 32 for 16x16x32), BN scale folded in f32 first (the same folded value the f32 image holds).
// np = 3: exact bf16 pieces; np = 2: fp16 pieces of the folded weight times wscale.
void build_wx3(const LayerSpec& L, const float* w, const std::vector<float>& inv, std::vector<uint16_t>& img, int np = 3, float wscale = 1.f) {
  img.assign((size_t)9 * L.cin * L.cout * np, 0);
  constexpr int kch = ASD_X3_S16 ? 32 : 16, kg = kch / 8;
  const int nc16 = L.cin / kch;
  for (int tap = 0; tap < 9; ++tap)
    for (int ci = 0; ci < L.cin; ++ci)
      for (int co = 0; co < L.cout; ++co) {
        const float v = w[((size_t)co * L.cin + ci) * 9 + tap] * inv[co];
        uint16_t p[3];
        if (np == 3) split3_host(v, p[0], p[1], p[2]);
        else split2_host(v * wscale, p[0], p[1]);
        const int c16 = ci / kch, hh = (ci % kch) / 8, j = ci % 8;
        for (int q = 0; q < np; ++q)
          img[(((((size_t)tap * nc16 + c16) * np + q) * kg + hh) * L.cout + co) * 8 + j] = p[q];
      }
}

// B-operand image of a 3x3 layer: [tap][cin/8][h][cout][jj] with cin = 8*c8 + 4*h + jj, BN scale folded.
void build_wimg(const LayerSpec& L, const float* w, const std::vector<float>& inv, std::vector<float>& img) {
  img.assign((size_t)9 * L.cin * L.cout, 0.f);
  for (int tap = 0; tap < 9; ++tap)
    for (int ci = 0; ci < L.cin; ++ci)
      for (int co = 0; co < L.cout; ++co) {
        const int c8 = ci / 8, hh = (ci % 8) / 4, jj = ci % 4;
        const size_t dst = ((((size_t)tap * (L.cin / 8) + c8) * 2 + hh) * L.cout + co) * 4 + jj;
        img[dst] = w[((size_t)co * L.cin + ci) * 9 + tap] * inv[co];
      }
}

}  // namespace

int asdnet_alloc(asd_ctx* ctx) {
  const size_t np = (size_t)ctx->cfg.max_patches;
  const size_t npad = (np + 31) / 32 * 32;
  ASD_HIP_CHECK(ctx, hipMalloc(&ctx->d_act[0], np * 32768 * sizeof(float)));
  ASD_HIP_CHECK(ctx, hipMalloc(&ctx->d_act[1], np * 32768 * sizeof(float)));
  ASD_HIP_CHECK(ctx, hipMalloc(&ctx->d_part, (size_t)FC_SK * npad * 128 * sizeof(float)));
  ASD_HIP_CHECK(ctx, hipMalloc(&ctx->d_patches, np * 1024));
  ASD_HIP_CHECK(ctx, hipMalloc(&ctx->d_desc, np * 128 * sizeof(float)));
  ASD_HIP_CHECK(ctx, hipHostMalloc(reinterpret_cast<void**>(&ctx->h_range), 64));
  *ctx->h_range = 0;
  if (const char* e = getenv("ASD_ASDNET_PAIR")) ctx->net_pair = atoi(e) != 0;
  if (const char* e = getenv("ASD_ASDNET_RING")) ctx->ring_mask = atoi(e);
  return ASD_OK;
}

void asdnet_free(asd_ctx* ctx) {
  for (int i = 0; i < 2; ++i) if (ctx->d_act[i]) (void)hipFree(ctx->d_act[i]);
  if (ctx->d_part) (void)hipFree(ctx->d_part);
  if (ctx->d_patches) (void)hipFree(ctx->d_patches);
  if (ctx->d_desc) (void)hipFree(ctx->d_desc);
  if (ctx->h_range) (void)hipHostFree(ctx->h_range);
  if (ctx->d_w1) (void)hipFree(ctx->d_w1);
  for (int i = 0; i < 7; ++i) {
    if (ctx->d_bias[i]) (void)hipFree(ctx->d_bias[i]);
    if (ctx->d_wimg[i]) (void)hipFree(ctx->d_wimg[i]);
    if (ctx->d_wx3[i]) (void)hipFree(ctx->d_wx3[i]);
    if (ctx->d_wx2[i]) (void)hipFree(ctx->d_wx2[i]);
  }
}

// Range calibration of the two-piece fp16 operand form (kActScale = 16: an activation beyond 4094 is not representable).  64
// synthetic patches -- uniform noise, step edges in four orientations, checkerboards of period 2 / 4 / 8, ramps, single dots: the
// patterns that drive a conv stack hardest after the per-patch normalisation -- go through the network as loaded; the largest
// |activation| of every layer is measured (conv1, which never reaches HBM, on the host from the same patches; conv2 .. conv6 by
// k_absmax behind each layer).  Anything above kCalibLimit = 2048 (a factor two of headroom) switches the context to the
// three-piece bf16 form, which has no range limit; asd_last_error / asd_asdnet_pieces report it.
static int asdnet_calibrate(asd_ctx* ctx, const float* const conv_w[7], const float* const bn_mean[7], const float* const bn_var[7], float eps);

int asdnet_load_weights(asd_ctx* ctx, const float* const conv_w[7], const float* const bn_mean[7],
                        const float* const bn_var[7], float eps) {
  for (int l = 0; l < 7; ++l) {
    const LayerSpec& L = kLayers[l];
    std::vector<float> inv(L.cout), bias(L.cout);
    for (int c = 0; c < L.cout; ++c) {
      inv[c] = 1.0f / std::sqrt(bn_var[l][c] + eps);
      bias[c] = -bn_mean[l][c] * inv[c];
    }
    if (!ctx->d_bias[l]) ASD_HIP_CHECK(ctx, hipMalloc(&ctx->d_bias[l], L.cout * sizeof(float)));
    ASD_HIP_CHECK(ctx, hipMemcpy(ctx->d_bias[l], bias.data(), L.cout * sizeof(float), hipMemcpyHostToDevice));
    std::vector<float> img;
    if (l == 0) {
      img.resize(32 * 9);
      for (int co = 0; co < 32; ++co)
        for (int k = 0; k < 9; ++k) img[co * 9 + k] = conv_w[0][co * 9 + k] * inv[co];
      if (!ctx->d_w1) ASD_HIP_CHECK(ctx, hipMalloc(&ctx->d_w1, img.size() * sizeof(float)));
      ASD_HIP_CHECK(ctx, hipMemcpy(ctx->d_w1, img.data(), img.size() * sizeof(float), hipMemcpyHostToDevice));
      continue;
    }
    if (l < 6) {
      build_wimg(L, conv_w[l], inv, img);
    } else {
      // k = px*128 + c (NHWC flatten of the 8x8x128 input), image [k/8][h][cout][jj]
      img.assign((size_t)8192 * 128, 0.f);
      for (int px = 0; px < 64; ++px)
        for (int c = 0; c < 128; ++c)
          for (int co = 0; co < 128; ++co) {
            const int k = px * 128 + c, k8 = k / 8, hh = (k % 8) / 4, jj = k % 4;
            img[(((size_t)k8 * 2 + hh) * 128 + co) * 4 + jj] = conv_w[6][((size_t)co * 128 + c) * 64 + px] * inv[co];
          }
    }
    if (!ctx->d_wimg[l]) ASD_HIP_CHECK(ctx, hipMalloc(&ctx->d_wimg[l], img.size() * sizeof(float)));
    ASD_HIP_CHECK(ctx, hipMemcpy(ctx->d_wimg[l], img.data(), img.size() * sizeof(float), hipMemcpyHostToDevice));
    if (l == 6) {
      // split image of the 8x8 conv for K2s: k = px*128 + c, [k/32][piece][k-group 4][cout 128][8 bf16]
      std::vector<uint16_t> x3((size_t)8192 * 128 * 3, 0);
      for (int px = 0; px < 64; ++px)
        for (int c = 0; c < 128; ++c)
          for (int co = 0; co < 128; ++co) {
            const int k = px * 128 + c, c32 = k / 32, g = (k % 32) / 8, j = k % 8;
            uint16_t pc[3];
            split3_host(conv_w[6][((size_t)co * 128 + c) * 64 + px] * inv[co], pc[0], pc[1], pc[2]);
            for (int q = 0; q < 3; ++q) x3[((((size_t)c32 * 3 + q) * 4 + g) * 128 + co) * 8 + j] = pc[q];
          }
      if (!ctx->d_wx3[l]) ASD_HIP_CHECK(ctx, hipMalloc(&ctx->d_wx3[l], x3.size() * sizeof(uint16_t)));
      ASD_HIP_CHECK(ctx, hipMemcpy(ctx->d_wx3[l], x3.data(), x3.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
      // the two-piece fp16 image for K2p: [k/32][piece 2][k-group 4][cout 128][8]
      const float ws = weight_scale_f16(conv_w[6], (size_t)128 * 128 * 64, inv, (size_t)128 * 64);
      ctx->wx2_scale[l] = ws;
      std::vector<uint16_t> x2((size_t)8192 * 128 * 2, 0);
      for (int px = 0; px < 64; ++px)
        for (int c = 0; c < 128; ++c)
          for (int co = 0; co < 128; ++co) {
            const int k = px * 128 + c, c32 = k / 32, g = (k % 32) / 8, j = k % 8;
            uint16_t pc[2];
            split2_host(conv_w[6][((size_t)co * 128 + c) * 64 + px] * inv[co] * ws, pc[0], pc[1]);
            for (int q = 0; q < 2; ++q) x2[((((size_t)c32 * 2 + q) * 4 + g) * 128 + co) * 8 + j] = pc[q];
          }
      if (!ctx->d_wx2[l]) ASD_HIP_CHECK(ctx, hipMalloc(&ctx->d_wx2[l], x2.size() * sizeof(uint16_t)));
      ASD_HIP_CHECK(ctx, hipMemcpy(ctx->d_wx2[l], x2.data(), x2.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    }
    if (l < 6) {
      std::vector<uint16_t> x3;
      build_wx3(L, conv_w[l], inv, x3);
      if (!ctx->d_wx3[l]) ASD_HIP_CHECK(ctx, hipMalloc(&ctx->d_wx3[l], x3.size() * sizeof(uint16_t)));
      ASD_HIP_CHECK(ctx, hipMemcpy(ctx->d_wx3[l], x3.data(), x3.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
      // the two-piece fp16 image (ASD_ASDNET_MATH=f16x2) and the power of two its weights were scaled by
      const float ws = weight_scale_f16(conv_w[l], (size_t)L.cout * L.cin * 9, inv, (size_t)L.cin * 9);
      ctx->wx2_scale[l] = ws;
      build_wx3(L, conv_w[l], inv, x3, 2, ws);
      if (!ctx->d_wx2[l]) ASD_HIP_CHECK(ctx, hipMalloc(&ctx->d_wx2[l], x3.size() * sizeof(uint16_t)));
      ASD_HIP_CHECK(ctx, hipMemcpy(ctx->d_wx2[l], x3.data(), x3.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    }
  }
  ctx->weights_loaded = true;
  return asdnet_calibrate(ctx, conv_w, bn_mean, bn_var, eps);
}

int asdnet_profile_collect_set(asd_ctx* ctx, int set);

static int asdnet_forward_one(asd_ctx* ctx, const uint8_t* d_patches, int n, float* d_desc, hipStream_t st, int* range_flag);
int asdnet_forward_device(asd_ctx* ctx, const uint8_t* d_patches, int n, float* d_desc, hipStream_t st, int* range_flag) {
  if (!ctx->weights_loaded) { ctx->set_error("asd_load_weights has not been called"); return ASD_ERR_NO_WEIGHTS; }
  if (n < 0 || n > ctx->cfg.max_patches) { ctx->set_error("n=%d exceeds max_patches=%d", n, ctx->cfg.max_patches); return ASD_ERR_CAPACITY; }
  if (n == 0) return ASD_OK;
  if (!st) st = ctx->stream;
  std::lock_guard<std::mutex> prof_lock(ctx->prof_mutex);  // prof_* state is shared with asd_profile_enable / _get
  return asdnet_forward_one(ctx, d_patches, n, d_desc, st, range_flag);
}
static int asdnet_forward_one(asd_ctx* ctx, const uint8_t* d_patches, int n, float* d_desc, hipStream_t st, int* range_flag) {
  float *a0 = ctx->d_act[0], *a1 = ctx->d_act[1];
  const int npad = (n + 31) / 32 * 32;
  const bool prof = ctx->prof_on;
  const int pset = ctx->prof_cur;
  if (prof && ctx->prof_pending[pset]) { int rc = asdnet_profile_collect_set(ctx, pset); if (rc != ASD_OK) return rc; }  // two forwards ago
#define PROF_MARK(i) do { if (prof) ASD_HIP_CHECK(ctx, hipEventRecord(ctx->prof_ev[pset][i], st)); } while (0)
  PROF_MARK(0);
  PROF_MARK(1);  // layer 0 (input_norm + conv1) is fused into conv2's band fill: no launch of its own
  // ASD_ASDNET_MATH=f16x2: two fp16 pieces per operand, three products (kActScale and the per-layer weight scale are undone in the epilogue)
  const bool p2 = ctx->net_pieces == 2;
  // pair format between the layers: every layer on the two-piece kernels, and not the calibration pass (k_absmax reads f32)
  const bool pair = kPairOK && p2 && (ctx->net_split & 63) == 63 && ctx->net_pair && !ctx->d_calib;
  const float* x3_stats = nullptr;   // conv2 only: the patches' normalisation statistics (k_patch_stats)
#define X3_LAUNCH(CFG, FUSE, l, src, dst, w1p, b1p)                                                                                       \
  (pair ? launch_conv_x3<CFG, FUSE, 2, kPairOK>(st, src, ctx->d_wx2[l], ctx->d_bias[l], dst, n, w1p, b1p, nullptr, nullptr, kActScale,       \
                                     1.f / (kActScale * ctx->wx2_scale[l]), x3_stats)                                                               \
   : p2 ? launch_conv_x3<CFG, FUSE, 2>(st, src, ctx->d_wx2[l], ctx->d_bias[l], dst, n, w1p, b1p, nullptr, nullptr, kActScale,               \
                                     1.f / (kActScale * ctx->wx2_scale[l]), x3_stats)                                                               \
      : launch_conv_x3<CFG, FUSE, 3>(st, src, ctx->d_wx3[l], ctx->d_bias[l], dst, n, w1p, b1p, nullptr, nullptr, 1.f, 1.f,                \
                                     x3_stats))
  if (ctx->net_split & 1) {
    // input_norm's mean / std of every patch once (k_patch_stats), in the head of the last layer's partial-sum buffer (free until that layer)
    hipLaunchKernelGGL(k_patch_stats, dim3(n), dim3(256), 0, st, d_patches, ctx->d_part, n);
    ASD_HIP_CHECK(ctx, hipGetLastError());
    x3_stats = ctx->d_part;
    ASD_HIP_CHECK(ctx, (X3_LAUNCH(L2S_CFG, true, 1, d_patches, a1, ctx->d_w1, ctx->d_bias[0])));
    x3_stats = nullptr;
  }
  else ASD_HIP_CHECK(ctx, (launch_conv<L2_CFG, true>(st, d_patches, ctx->d_wimg[1], ctx->d_bias[1], a1, n, ctx->d_w1, ctx->d_bias[0])));
  // calibration (asd_load_weights): the largest |activation| each layer hands to the next one
#define CALIB(l, buf, elems) do { if (ctx->d_calib) { hipLaunchKernelGGL(k_absmax, dim3(256), dim3(256), 0, st, buf, (size_t)n * (elems), ctx->d_calib + (l)); ASD_HIP_CHECK(ctx, hipGetLastError()); } } while (0)
  CALIB(1, a1, 32 * 32 * 32);
  PROF_MARK(2);
  if (ctx->net_split & 2) ASD_HIP_CHECK(ctx, (X3_LAUNCH(L3S_CFG, false, 2, a1, a0, nullptr, nullptr)));
  else
  ASD_HIP_CHECK(ctx, (launch_conv_p<L3_CFG>(st, a1, ctx->d_wimg[2], ctx->d_bias[2], a0, n, ctx->num_cu)));
  CALIB(2, a0, 16 * 16 * 64);
  PROF_MARK(3);
  const int ring = pair ? ctx->ring_mask : 0;   // asdnet_ring.hip (whole-patch LDS images, weights through an LDS-DMA ring): pair form only
  float *src6 = a0, *dst6 = a1;   // conv6's input and output (the fc layer reads dst6)
  if (ring & 4) {   // conv4 + conv5 in one launch (conv4's output never leaves LDS); the profile charges the launch to conv5's slot
    PROF_MARK(4);
    { const int rc = asdnet_ring_conv45(ctx, a0, a1, n, st); if (rc != ASD_OK) return rc; }
    src6 = a1; dst6 = a0;
  } else {
    if (ring & 1) { const int rc = asdnet_ring_conv(ctx, 3, a0, a1, n, st); if (rc != ASD_OK) return rc; }
    else if (ctx->net_split & 4) ASD_HIP_CHECK(ctx, (X3_LAUNCH(L4S_CFG, false, 3, a0, a1, nullptr, nullptr)));
    else ASD_HIP_CHECK(ctx, (launch_conv<L4_CFG>(st, a0, ctx->d_wimg[3], ctx->d_bias[3], a1, n)));
    CALIB(3, a1, 16 * 16 * 64);
    PROF_MARK(4);
    if (ctx->net_split & 8) ASD_HIP_CHECK(ctx, (X3_LAUNCH(L5S_CFG, false, 4, a1, a0, nullptr, nullptr)));
    else
    ASD_HIP_CHECK(ctx, (launch_conv<L5_CFG>(st, a1, ctx->d_wimg[4], ctx->d_bias[4], a0, n)));
  }
  CALIB(4, src6, 8 * 8 * 128);
  PROF_MARK(5);
  if (ring & 2) { const int rc = asdnet_ring_conv(ctx, 5, src6, dst6, n, st); if (rc != ASD_OK) return rc; }
  else if (ctx->net_split & 16) ASD_HIP_CHECK(ctx, (X3_LAUNCH(L6S_CFG, false, 5, src6, dst6, nullptr, nullptr)));
  else ASD_HIP_CHECK(ctx, (launch_conv<L6_CFG>(st, src6, ctx->d_wimg[5], ctx->d_bias[5], dst6, n)));
#undef X3_LAUNCH
  CALIB(5, dst6, 8 * 8 * 128);
#undef CALIB
  PROF_MARK(6);
  a1 = dst6;   // (the last layer's input)
  ctx->d_act6 = dst6;
  if (pair)
    hipLaunchKernelGGL(k_fc_x2, dim3((npad + FCS_MP - 1) / FCS_MP, FC_SK), dim3(256), 0, st, reinterpret_cast<const uint8_t*>(a1),
                       static_cast<const uint8_t*>(ctx->d_wx2[6]), ctx->d_part, n, npad, 1.f / (kActScale * ctx->wx2_scale[6]));
  else if (ctx->net_split & 32)
    hipLaunchKernelGGL(k_fc_x3, dim3((npad + FCS_MP - 1) / FCS_MP, FC_SK), dim3(256), 0, st, a1, static_cast<const uint8_t*>(ctx->d_wx3[6]), ctx->d_part, n, npad);
  else
    hipLaunchKernelGGL(k_fc_mfma, dim3((npad / 32 + FC_MT - 1) / FC_MT, FC_SK), dim3(256), 0, st, a1, ctx->d_wimg[6], ctx->d_part, n, npad);
  ASD_HIP_CHECK(ctx, hipGetLastError());
  PROF_MARK(7);
  hipLaunchKernelGGL(k_l2norm, dim3((n + 3) / 4), dim3(256), 0, st, ctx->d_part, ctx->d_bias[6], d_desc, n, npad, range_flag);
  ASD_HIP_CHECK(ctx, hipGetLastError());
  PROF_MARK(8);
#undef PROF_MARK
  if (prof) { ctx->prof_pending[pset] = true; ctx->prof_pending_n[pset] = n; ctx->prof_cur ^= 1; }
  return ASD_OK;
}

static int asdnet_calibrate(asd_ctx* ctx, const float* const conv_w[7], const float* const bn_mean[7], const float* const bn_var[7], float eps) {
  ctx->calib_note.clear();
  ctx->net_pieces = ctx->net_pieces_req;   // every load starts from the form the context was created with: a fall-back is not sticky
  if (ctx->net_pieces != 2 || !(ctx->net_split & 31)) return ASD_OK;   // only the fp16 form has a range
  if (const char* e = getenv("ASD_ASDNET_CALIBRATE")) if (atoi(e) == 0) return ASD_OK;   // tests of the run-time flag switch the guard at load off
  const int N = std::min(64, ctx->cfg.max_patches);   // (a context sized for fewer patches than the calibration set calibrates on what fits)
  constexpr float kCalibLimit = 2048.f;
  std::vector<uint8_t> pat((size_t)N * 1024);
  uint32_t rng = 0x9e3779b9u;
  auto rnd = [&]() { rng = rng * 1664525u + 1013904223u; return rng >> 8; };
  for (int p = 0; p < N; ++p)
    for (int y = 0; y < 32; ++y)
      for (int x = 0; x < 32; ++x) {
        int v;
        const int kind = p % 8, var = p / 8;
        switch (kind) {
          case 0: v = (int)(rnd() & 255); break;                                              // uniform noise
          case 1: v = (x < 8 + 2 * var) ? 20 : 235; break;                                     // vertical edge
          case 2: v = (y < 8 + 2 * var) ? 235 : 20; break;                                     // horizontal edge
          case 3: v = (x + y < 24 + 2 * var) ? 10 : 245; break;                                // diagonal edge
          case 4: v = (((x >> (var % 3)) + (y >> (var % 3))) & 1) ? 255 : 0; break;            // checkerboard, period 2 / 4 / 8
          case 5: v = (x * 255) / 31; break;                                                   // ramp
          case 6: v = (x == 4 * var + 1 && y == 3 * var + 2) ? 255 : 0; break;                 // single dot (std -> tiny: large normalised peak)
          default: v = 128 + (int)(rnd() & 3); break;                                          // almost flat
        }
        pat[(size_t)p * 1024 + y * 32 + x] = (uint8_t)v;
      }
  // conv1 on the host: input_norm (ASDNet.py:360-365) + 3x3 conv + folded BN + ReLU, its maximum only
  float max1 = 0.f;
  {
    std::vector<float> inv(32), bias(32);
    for (int c = 0; c < 32; ++c) { inv[c] = 1.0f / std::sqrt(bn_var[0][c] + eps); bias[c] = -bn_mean[0][c] * inv[c]; }
    std::vector<float> in(34 * 34);
    for (int p = 0; p < N; ++p) {
      double s1 = 0, s2 = 0;
      for (int i = 0; i < 1024; ++i) s1 += pat[(size_t)p * 1024 + i] / 255.0;
      const double mean = s1 / 1024;
      for (int i = 0; i < 1024; ++i) { const double d = pat[(size_t)p * 1024 + i] / 255.0 - mean; s2 += d * d; }
      const double sd = std::sqrt(s2 / 1023) + 1e-7;
      std::fill(in.begin(), in.end(), 0.f);
      for (int y = 0; y < 32; ++y)
        for (int x = 0; x < 32; ++x) in[(y + 1) * 34 + x + 1] = (float)((pat[(size_t)p * 1024 + y * 32 + x] / 255.0 - mean) / sd);
      for (int c = 0; c < 32; ++c)
        for (int y = 0; y < 32; ++y)
          for (int x = 0; x < 32; ++x) {
            float a = bias[c];
            for (int k = 0; k < 9; ++k) a += in[(y + k / 3) * 34 + x + k % 3] * conv_w[0][c * 9 + k] * inv[c];
            max1 = std::max(max1, a);
          }
    }
  }
  // conv2 .. conv6 on the device, as loaded
  uint8_t* d_pat = nullptr;
  unsigned* d_max = nullptr;
  float* d_out = nullptr;
  int rc = ASD_OK;
  unsigned hmax[8] = {};
  do {
    if (hipMalloc(&d_pat, pat.size()) != hipSuccess || hipMalloc(&d_max, 8 * sizeof(unsigned)) != hipSuccess ||
        hipMalloc(&d_out, (size_t)N * 128 * sizeof(float)) != hipSuccess) { rc = ASD_ERR_HIP; break; }   // (whatever was allocated is freed below)
    if (hipMemcpy(d_pat, pat.data(), pat.size(), hipMemcpyHostToDevice) != hipSuccess || hipMemset(d_max, 0, 8 * sizeof(unsigned)) != hipSuccess) { rc = ASD_ERR_HIP; break; }
    ctx->d_calib = d_max;
    rc = asdnet_forward_device(ctx, d_pat, N, d_out, ctx->stream, nullptr);
    ctx->d_calib = nullptr;
    if (rc != ASD_OK) break;
    if (hipStreamSynchronize(ctx->stream) != hipSuccess || hipMemcpy(hmax, d_max, sizeof hmax, hipMemcpyDeviceToHost) != hipSuccess) rc = ASD_ERR_HIP;
  } while (0);
  if (d_pat) (void)hipFree(d_pat);
  if (d_max) (void)hipFree(d_max);
  if (d_out) (void)hipFree(d_out);
  if (rc != ASD_OK) { ctx->set_error("asd_load_weights: the range calibration pass failed"); return rc; }
  float worst = max1;
  int worst_layer = 1;
  for (int l = 1; l <= 5; ++l) {
    float v;
    memcpy(&v, &hmax[l], 4);
    if (!(v <= 3.0e38f)) v = INFINITY;
    if (v > worst) { worst = v; worst_layer = l + 1; }
  }
  if (worst > kCalibLimit) {
    ctx->net_pieces = 3;
    char buf[256];
    snprintf(buf, sizeof buf, "asd_load_weights: calibration found |activation| = %g behind conv%d (limit %g for the two-piece fp16 form): this context uses the three-piece bf16 form (asd_asdnet_pieces = 3)",
             (double)worst, worst_layer, (double)kCalibLimit);
    ctx->calib_note = buf;   // a note, not an error: asd_calibration_note() hands it out, asd_last_error stays as it was
  }
  return ASD_OK;
}

int asdnet_profile_collect_set(asd_ctx* ctx, int set) {
  if (!ctx->prof_pending[set]) return ASD_OK;
  ASD_HIP_CHECK(ctx, hipEventSynchronize(ctx->prof_ev[set][8]));
  for (int l = 0; l < 8; ++l) {
    float ms = 0;
    ASD_HIP_CHECK(ctx, hipEventElapsedTime(&ms, ctx->prof_ev[set][l], ctx->prof_ev[set][l + 1]));
    ctx->prof_ms[l] += ms;
    ctx->prof_calls[l] += 1;
    ctx->prof_patches[l] += ctx->prof_pending_n[set];
  }
  ctx->prof_pending[set] = false;
  return ASD_OK;
}

int asdnet_profile_collect(asd_ctx* ctx) {
  std::lock_guard<std::mutex> prof_lock(ctx->prof_mutex);
  for (int set = 0; set < 2; ++set) {
    const int rc = asdnet_profile_collect_set(ctx, set);
    if (rc != ASD_OK) return rc;
  }
  return ASD_OK;
}

// debug / tuning aid (not part of the C ABI header): time conv2 with parts of the kernel removed
// mode 0 = full, 1 = no activation staging, 2 = no MFMA, 3 = no epilogue stores
template <int ABL>
static int ablate_one(asd_ctx* ctx, int layer, int n, int reps, float* ms) {
  hipStream_t st = ctx->stream;
  float *a0 = ctx->d_act[0], *a1 = ctx->d_act[1];
  hipError_t e = hipSuccess;
  auto run = [&](int count) {
    for (int r = 0; r < count; ++r) {
      if (layer == 2) {
        using C = ConvCfg<L2_CFG>;
        constexpr int rows = C::M_PATCH / C::HO;
        constexpr int lds = C::LDS_BYTES + ((rows + 4) * 36 + 320 + 8) * 4;
        auto k = k_conv_mfma<L2_CFG, ABL, true>;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL(k, dim3(n * (C::HO / rows)), dim3(C::NTH), lds, st, (const void*)ctx->d_patches, ctx->d_wimg[1], ctx->d_bias[1], a1, ctx->d_w1, ctx->d_bias[0], n);
      } else if (layer == 4) {
        using C = ConvCfg<L4_CFG>;
        auto k = k_conv_mfma<L4_CFG, ABL, false>;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        hipLaunchKernelGGL(k, dim3(n * (C::HO * C::HO / C::M_PATCH)), dim3(C::NTH), C::LDS_BYTES, st, (const void*)a0, ctx->d_wimg[3], ctx->d_bias[3], a1, (const float*)nullptr, (const float*)nullptr, n);
      } else {
        using C = ConvCfg<L6_CFG>;
        auto k = k_conv_mfma<L6_CFG, ABL, false>;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        hipLaunchKernelGGL(k, dim3(((n + C::M_WG / C::M_PATCH - 1) / (C::M_WG / C::M_PATCH)) * (C::HO * C::HO / C::M_PATCH)), dim3(C::NTH), C::LDS_BYTES, st, (const void*)a0, ctx->d_wimg[5], ctx->d_bias[5], a1, (const float*)nullptr, (const float*)nullptr, n);
      }
    }
  };
  run(3);
  ASD_HIP_CHECK(ctx, hipEventRecord(ctx->ev0, st));
  run(reps);
  ASD_HIP_CHECK(ctx, hipEventRecord(ctx->ev1, st));
  ASD_HIP_CHECK(ctx, hipEventSynchronize(ctx->ev1));
  ASD_HIP_CHECK(ctx, hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
  *ms /= reps;
  (void)e;
  return ASD_OK;
}

// debug / test aid (not part of the C ABI header): the activation conv6 handed to the last layer in the most recent forward, first n
// patches, as f32 [n][64 pixels][128 channels]; a context in the pair format returns (h + l) / kActScale (exact in f32).
extern "C" int asd_debug_act6(asd_ctx* ctx, int n, float* out) {
  if (n < 0 || n > ctx->cfg.max_patches) return ASD_ERR_CAPACITY;
  ASD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  const bool pair = kPairOK && ctx->net_pieces == 2 && (ctx->net_split & 63) == 63 && ctx->net_pair;
  const float* src = ctx->d_act6 ? ctx->d_act6 : ctx->d_act[1];
  if (!pair) { ASD_HIP_CHECK(ctx, hipMemcpy(out, src, (size_t)n * 8192 * 4, hipMemcpyDeviceToHost)); return ASD_OK; }
  std::vector<uint16_t> raw((size_t)n * 8192 * 2);
  ASD_HIP_CHECK(ctx, hipMemcpy(raw.data(), src, raw.size() * 2, hipMemcpyDeviceToHost));
  for (size_t e = 0; e < (size_t)n * 8192; ++e) {
    const size_t g = e / 8, j = e % 8;
    _Float16 h, l;
    memcpy(&h, &raw[g * 16 + j], 2);
    memcpy(&l, &raw[g * 16 + 8 + j], 2);
    out[e] = ((float)h + (float)l) / kActScale;
  }
  return ASD_OK;
}

// debug / tuning aid (not part of the C ABI header): in-kernel clock of a split-operand layer (MI355X_MICROARCH.md, DVFS item 6).
// Runs the layer `reps` times back to back with stamps around the MFMA loop; returns the median over workgroups of the loop's
// shader cycles and of shader cycles per 10 ns wall tick (x 0.1 = GHz).  Uses d_part as the stamp buffer.
extern "C" int asd_debug_x3_clock(asd_ctx* ctx, int layer, int n, int reps, double* loop_cycles, double* ghz) {
  hipStream_t st = ctx->stream;
  float *a0 = ctx->d_act[0], *a1 = ctx->d_act[1];
  unsigned long long* stamps = reinterpret_cast<unsigned long long*>(ctx->d_part);
  int grid = 0;
  for (int r = 0; r < reps; ++r) {
    unsigned long long* sp = r + 1 == reps ? stamps : nullptr;
    hipError_t e = hipErrorInvalidValue;
    const bool p2 = ctx->net_pieces == 2;
    const bool pair = kPairOK && p2 && (ctx->net_split & 63) == 63 && ctx->net_pair;
#define X3_CLK(CFG, FUSE, l, src, dst, w1p, b1p)                                                                                          \
  (pair ? launch_conv_x3<CFG, FUSE, 2, kPairOK>(st, src, ctx->d_wx2[l], ctx->d_bias[l], dst, n, w1p, b1p, sp, &grid, kActScale,               \
                                     1.f / (kActScale * ctx->wx2_scale[l]))                                                                \
   : p2 ? launch_conv_x3<CFG, FUSE, 2>(st, src, ctx->d_wx2[l], ctx->d_bias[l], dst, n, w1p, b1p, sp, &grid, kActScale,                       \
                                     1.f / (kActScale * ctx->wx2_scale[l]))                                                                \
      : launch_conv_x3<CFG, FUSE, 3>(st, src, ctx->d_wx3[l], ctx->d_bias[l], dst, n, w1p, b1p, sp, &grid))
    if (layer == 2) e = X3_CLK(L2S_CFG, true, 1, ctx->d_patches, a1, ctx->d_w1, ctx->d_bias[0]);
    else if (layer == 3) e = X3_CLK(L3S_CFG, false, 2, a1, a0, nullptr, nullptr);
    else if (layer == 4) e = X3_CLK(L4S_CFG, false, 3, a0, a1, nullptr, nullptr);
    else if (layer == 5) e = X3_CLK(L5S_CFG, false, 4, a1, a0, nullptr, nullptr);
    else if (layer == 6) e = X3_CLK(L6S_CFG, false, 5, a0, a1, nullptr, nullptr);
#undef X3_CLK
    ASD_HIP_CHECK(ctx, e);
  }
  ASD_HIP_CHECK(ctx, hipStreamSynchronize(st));
  if ((size_t)grid * 128 > (size_t)FC_SK * ((ctx->cfg.max_patches + 31) / 32 * 32) * 128 * sizeof(float)) return ASD_ERR_CAPACITY;
  std::vector<unsigned long long> hs((size_t)grid * 16);
  ASD_HIP_CHECK(ctx, hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost));
  std::vector<double> cyc(grid), rate(grid);
  for (int i = 0; i < grid; ++i) { cyc[i] = (double)hs[16 * i]; rate[i] = hs[16 * i + 1] ? (double)hs[16 * i] / (double)hs[16 * i + 1] : 0.0; }
  std::nth_element(cyc.begin(), cyc.begin() + grid / 2, cyc.end());
  std::nth_element(rate.begin(), rate.begin() + grid / 2, rate.end());
  *loop_cycles = cyc[grid / 2];
  *ghz = rate[grid / 2] * 0.1;
  // phases of a workgroup's life (medians, shader cycles): entry -> input ready for conv1 / staging, -> band staged, MFMA loop, whole
  // life; ASD_X3_PHASES=1 prints them
  if (getenv("ASD_X3_PHASES")) {
    auto med = [&](int k) { std::vector<double> v(grid); for (int i = 0; i < grid; ++i) v[i] = (double)hs[16 * i + k]; std::nth_element(v.begin(), v.begin() + grid / 2, v.end()); return v[grid / 2]; };
    fprintf(stderr, "  layer %d: %d workgroups; prologue %.0f, staging %.0f, MFMA loop %.0f, whole life %.0f cycles (medians)\n", layer, grid, med(2), med(3),
            med(0), med(5));
    if (layer == 2) fprintf(stderr, "     since entry: patch arrived %.0f, staging loop done %.0f, barrier passed %.0f\n", med(9), med(12), med(13));
  }
  return ASD_OK;
}

// debug / tuning aid (not part of the C ABI header): time conv2 (fused) / conv4 / conv6 with parts removed.
// mode bits: 1 = no input staging, 2 = no MFMA, 4 = no epilogue stores, 8 = no per-stage barrier, 16 = no weight streaming
extern "C" int asd_debug_conv_ablate(asd_ctx* ctx, int layer, int n, int mode, int reps, float* ms) {
  switch (mode) {
    case 0: return ablate_one<0>(ctx, layer, n, reps, ms);
    case 1: return ablate_one<1>(ctx, layer, n, reps, ms);
    case 2: return ablate_one<2>(ctx, layer, n, reps, ms);
    case 3: return ablate_one<3>(ctx, layer, n, reps, ms);
    case 4: return ablate_one<4>(ctx, layer, n, reps, ms);
    case 5: return ablate_one<5>(ctx, layer, n, reps, ms);
    case 6: return ablate_one<6>(ctx, layer, n, reps, ms);
    case 7: return ablate_one<7>(ctx, layer, n, reps, ms);
    case 13: return ablate_one<13>(ctx, layer, n, reps, ms);   // MFMA loop, no per-stage barrier
    case 21: return ablate_one<21>(ctx, layer, n, reps, ms);   // MFMA loop, no weight streaming
    case 29: return ablate_one<29>(ctx, layer, n, reps, ms);   // MFMA loop, neither
    default: return ASD_ERR_INVALID;
  }
}
