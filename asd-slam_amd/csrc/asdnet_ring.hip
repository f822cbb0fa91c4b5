// asdnet_ring.hip -- ASDNet conv layers (ASDNet.py:334-356) with BOTH MFMA operands read from LDS: whole-patch images, weights through an
// LDS-DMA ring, conv4 -> conv5 fused through LDS.  EXPERIMENTAL (round 5): bit-identical to the layer-by-layer kernels and opt-in
// (ASD_ASDNET_RING = bit 0 conv4 | bit 1 conv6 | bit 2 conv4 + conv5 in one launch); measured SLOWER than k_conv_x3 so far, numbers below.
//
// What round 4's stamps said about k_conv_x3 (asdnet.hip): a workgroup spends 20-60 % of its life in the MFMA loop, every wave streams the
// B operands (weights) of its own cout columns through the CU's vector-memory path, and every layer's activations make a round trip
// through HBM (1.44 GB per forward).  This file is the structure VERDICT r04 item 2 asked for:
//  * one persistent workgroup (eight waves, one per CU) works on WHOLE patches; the input image of a layer lives in LDS with a one-pixel zero
//    halo (row pitch H + 1: a row's right halo pixel is the next row's left one), unpadded pixels ([pixel][c/8][h | l][8 x fp16], the pair
//    format of asdnet.hip) and an XOR swizzle of the 16-B slots -- a layout LDS-DMA can fill (1 KB per instruction, the swizzle applied on
//    the per-lane SOURCE address);
//  * the weights of a k-chunk (one tap x 32 input channels x all couts, both fp16 pieces: 128 x COUT bytes) arrive by LDS-DMA
//    (global_load_lds_dwordx4) in a ring of four slots, three chunks ahead, ONE raw s_barrier per chunk behind a counted vmcnt; all waves
//    read their B sub-tiles from the slot, so a weight crosses the vector-memory path once per workgroup instead of once per wave;
//  * operands are double buffered in registers by chunk (reads of chunk c + 1 under the MFMAs of chunk c, issue order pinned);
//  * layers are FUSED where the images fit: conv4's epilogue writes conv5's LDS image in place (all of a layer's MFMAs are issued before
//    its epilogue runs, so its input is dead by then) -- conv4's output (131 MB written + read per forward) never touches HBM.
// Arithmetic is that of k_conv_x3's two-piece form, operation for operation: x = h + l in fp16 pieces of 16 x, products l h, h l, h h in
// that order per 32-deep chunk, chunks in (tap, cin / 32) order, f32 accumulate, epilogue acc * out_scale + bias, ReLU, split4_mix -- the
// outputs are the same BITS as the layer-by-layer kernels' (tests/test_asdnet.py::test_ring_kernels_are_bit_identical).
//
// MEASURED (N = 2000, one MI355X, tools/time_asdnet.py; ablations by tools/ring_variants.sh + tools/ring_abl.sh, counters by tools/ring_pmc.sh):
//   conv4 + conv5 fused 178-188 us against 96 + 66 = 162 layer by layer; conv6 (two patches per tile) 108-112 against 92; conv4 alone 153.
//   Where the fused kernel's 50 k cycles per patch go (its MFMAs need 20.7 k cycles of a SIMD's matrix pipe: busy 40 %): without MFMAs the
//   launch still takes 142 us; of that the image DMA is 20, the weight DMA 23, the output stores 12, the 36 barriers per patch 25 (~200
//   cycles each with eight waves) and the bare operand reads + epilogue arithmetic 80 -- LDS array busy 33 % of the launch, 27 % of its cycles
//   bank conflicts (ds_read_b128 is served in lane groups {0-3, 12-15, 20-27} ...: eight lanes each of two k-groups, whose slots the
//   XOR swizzle does not keep apart), 950 non-MFMA vector instructions per wave and patch (swizzled addresses: two per A read) against 648
//   MFMAs, i.e. two waves per SIMD fill its issue port.  What would have to change for it to win: addresses as immediates (k-groups in
//   separate 256-B-aligned planes, pixels padded by 16 B: conflict-free for the true lane groups, no address arithmetic in the loop), one
//   barrier per TAP instead of per chunk, the next patch's image staged through registers under the last layer -- and the stride-2
//   layers need a column-de-interleaved image (a stride-2 walk over a row-major image reaches only every second bank), which no longer
//   fits beside a 64 KB ring.  Not done in round 5; the default forward stays on k_conv_x3.
#include <stdint.h>

#include <hip/hip_runtime.h>

#include "ctx.h"

#ifndef ASD_RING_RSPREAD
#define ASD_RING_RSPREAD 1
#endif
#ifndef ASD_RING_ABL
#define ASD_RING_ABL 0   // tuning builds only (tools/ring_variants.sh): 1 no image DMA, 2 no output stores, 4 no MFMAs, 8 no in-loop barriers, 16 no weight DMA
#endif
namespace {
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr float kActScaleR = 16.f;   // = kActScale of asdnet.hip

// two-piece fp16 split of four f32 values times s (asdnet.hip::split4_mix, same instruction sequence: same bits)
__device__ __forceinline__ void split4_mix_r(float x0, float x1, float x2, float x3, float s, uint32_t (&h)[2], uint32_t (&l)[2]) {
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

// one LDS-DMA: 64 lanes x 16 B from per-lane global addresses to lds_wave_base + 16 * lane (lds_wave_base is wave uniform: it goes to M0)
__device__ __forceinline__ void dma16(const void* g, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// ---- geometry of one 3x3 layer (pad 1) whose input is an LDS image -------------------------------------------------------------------
// image: (H + 2) x (H + 2) pixels (halo included) of CIN * 4 bytes at row pitch PW = H + 1; slot s = (c / 8) * 2 + piece of pixel (y, x) (padded coordinates) sits at
// ((y * PW + x) * PIXB) + ((s ^ key(y, x)) << 4).  key() is chosen per CONSUMER so that the 16 lanes of an A sub-tile (16 consecutive output
// pixels, row major) read 16 different 16-B bank slots for every tap:
//   H = 16, S = 1 (one output row per sub-tile):        key = x & 15
//   H =  8, S = 1 (two output rows of 8 per sub-tile):   key = (x + 8 (y & 1)) & 15
//   H = 16, S = 2 (two output rows of 8, inputs 2 apart): key = ((x >> 1) + 8 ((y >> 1) & 1)) & 15
template <int CIN_, int COUT_, int H_, int S_>
struct Geo {
  static constexpr int CIN = CIN_, COUT = COUT_, H = H_, S = S_;
  // row pitch H + 1: the right halo pixel of row y IS the left halo pixel of row y + 1 (both zero), so an image is (H + 2) (H + 1) + 1 pixels
  static constexpr int HO = H / S, PW = H + 1, PIXB = CIN * 4, NPIX = (H + 2) * PW + 1, IMGB = NPIX * PIXB;
  static constexpr int NC16 = CIN / 32, NCHUNK = 9 * NC16, CHUNKB = COUT * 128;
  static constexpr int SLOTS = PIXB / 16;   // 16-B slots per pixel
  static_assert(PIXB % 256 == 0, "a pixel covers whole 256-B bank rows (CIN >= 64)");
  static_assert((H == 16 && (S == 1 || S == 2)) || (H == 8 && S == 1), "swizzle keys are defined for these geometries");
  __device__ static __forceinline__ int key(int y, int x) {
    if constexpr (H == 16 && S == 1) return x & 15;
    else if constexpr (H == 8) return (x + 8 * (y & 1)) & 15;
    else return ((x >> 1) + 8 * ((y >> 1) & 1)) & 15;
  }
};

// ---- the weight ring -------------------------------------------------------------------------------------------------------------------
constexpr int kRingSlots = 4;
// chunk c of a layer's weight image (build_wx3 layout, np = 2: [tap][cin / 32][piece][k-group][cout][8 x fp16] = CHUNKB contiguous bytes)
// into ring slot `slot`: every wave issues its share of the CHUNKB / 1024 DMA instructions
template <int CHUNKB, int NW>
__device__ __forceinline__ void ring_issue(const uint8_t* __restrict__ wimg, int c, uint8_t* ring, int slot, int wave, int lane) {
  constexpr int NI = CHUNKB / 1024;
  static_assert(NI % NW == 0, "every wave issues the same number of DMA instructions per chunk");
  if (ASD_RING_ABL & 16) return;
#pragma unroll
  for (int i = 0; i < NI / NW; ++i) {
    const int piece = i * NW + wave;   // wave uniform
    dma16(wimg + (size_t)c * CHUNKB + (size_t)piece * 1024 + lane * 16, ring + (size_t)slot * CHUNKB + piece * 1024);
  }
}

// ---- one layer's MFMA loop over an LDS image ---------------------------------------------------------------------------------------------
// Workgroup of NW = WM x WN waves; wave (wm, wn) owns pixels [wm * MW, (wm + 1) * MW) of the tile's M = PP * HO * HO output pixels and couts
// [wn * NCW, (wn + 1) * NCW): NA x NB accumulator sub-tiles of 16 x 16 (transposed: rows = couts 4 kg + r, columns = pixels lr).
template <class G, int PP, int WM, int WN>
struct Tile {
  static constexpr int NW = WM * WN, NTH = 64 * NW;
  static constexpr int M = PP * G::HO * G::HO, MW = M / WM, NA = MW / 16, NCW = G::COUT / WN, NB = NCW / 16;
  static constexpr int NWW = G::CHUNKB / 1024 / NW;   // DMA instructions per wave and chunk
  static_assert(M % (16 * WM) == 0 && G::COUT % (16 * WN) == 0, "tile split");
  static_assert((G::HO * G::HO) % 16 == 0, "a sub-tile never straddles two patches");
};

typedef float accv __attribute__((ext_vector_type(4)));

// issue order inside a sub-tile step: the step's RPS operand reads spread evenly over its TOT MFMAs (one read behind every ~TOT / RPS MFMAs)
template <int R, int RPS, int TOT>
struct SchedStep {
  static __device__ __forceinline__ void run() {
    constexpr int n = (R + 1) * TOT / RPS - R * TOT / RPS;
    if constexpr (n > 0) __builtin_amdgcn_sched_group_barrier(0x008, n, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    if constexpr (R + 1 < RPS) SchedStep<R + 1, RPS, TOT>::run();
  }
};

// The chunk loop.  On entry: the image(s) are complete and visible, ring slots 0..2 hold chunks 0..2 (issued by ring_prologue, waited for
// and published by a barrier) and this wave has nothing outstanding on the vector-memory counter.
// The loop over the nine taps is a real loop (fully unrolled, the compiler computed every operand address of every chunk up front and kept
// them all live: 255 + 256 registers and spills); inside a tap the cin chunks are unrolled, so the register double buffer of the B operands
// and the A operand ring (four slots, two sub-tiles ahead) have compile-time indices.
template <class G, int PP, int WM, int WN>
__device__ __forceinline__ void conv_ring_loop(const uint8_t* __restrict__ img, uint8_t* __restrict__ ring, const uint8_t* __restrict__ wimg,
                                               accv (&acc)[Tile<G, PP, WM, WN>::NA][Tile<G, PP, WM, WN>::NB], const int wave, const int lane) {
  using T = Tile<G, PP, WM, WN>;
  constexpr int NA = T::NA, NB = T::NB, NC16 = G::NC16;
  static_assert(NC16 % 2 == 0, "static register sets across taps");
  const int wm = wave / WN, wn = wave % WN;
  const int kg = lane >> 4, lr = lane & 15;
  // A addressing: abase[ma] = the sub-tile's output pixel's top-left input pixel (padded coordinates: tap (0, 0))
  int abase[NA];
  int oyp = 0, oxl = 0;
#pragma unroll
  for (int ma = 0; ma < NA; ++ma) {
    const int m = wm * T::MW + ma * 16 + lr;
    const int pp = m / (G::HO * G::HO), mm = m % (G::HO * G::HO);
    const int oy = mm / G::HO, ox = mm % G::HO;
    abase[ma] = pp * G::IMGB + ((oy * G::S) * G::PW + ox * G::S) * G::PIXB;
    if (ma == 0) { oyp = oy; oxl = ox; }   // (oy & 1) and ox are the same for every sub-tile of the lane: see Geo::key
  }
  const int bbase = (kg * G::COUT + wn * T::NCW + lr) * 16;
  // per tap: byte offset of the tap's pixel and the lane's swizzle term (key ^ (kg << 1)) << 4
  auto tap_off = [&](int tap) { const int dy = tap / 3, dx = tap - 3 * dy; return (dy * G::PW + dx) * G::PIXB; };
  auto tap_kk = [&](int tap) { const int dy = tap / 3, dx = tap - 3 * dy; return (G::key(oyp * G::S + dy, oxl * G::S + dx) ^ (kg << 1)) << 4; };
  auto load_a = [&](int off, int kk, int c16, int ma, u32x4 (&a)[2]) {
#pragma unroll
    for (int p = 0; p < 2; ++p) a[p] = *reinterpret_cast<const u32x4*>(img + abase[ma] + off + (kk ^ ((c16 * 8 + p) << 4)));
  };
  auto load_b = [&](int c, u32x4 (&b)[NB][2]) {
    const uint8_t* s = ring + (c & (kRingSlots - 1)) * G::CHUNKB + bbase;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int p = 0; p < 2; ++p) b[nb][p] = *reinterpret_cast<const u32x4*>(s + p * 4 * G::COUT * 16 + nb * 256);
  };
  auto hf = [](const u32x4& v) { return __builtin_bit_cast(f16x8, v); };
  // Operands are double buffered in registers by CHUNK: while the MFMAs of chunk c run on set c & 1, the reads of chunk c + 1 (its ring slot has
  // been visible since the barrier before chunk c) fill the other set, one read behind every few MFMAs.  At the barrier that ends the chunk
  // every operand of the next one has been requested long ago, so the first MFMA behind the barrier does not wait for an LDS round trip
  // that eight waves have just queued up for (SQ_WAIT_ANY was 37 % of the wave-cycles with the operands fetched two sub-tiles ahead).
  u32x4 br[2][NB][2];
  u32x4 ar[2][NA][2];
  load_b(0, br[0]);
  {
    const int off0 = tap_off(0), kk0 = tap_kk(0);
#pragma unroll
    for (int ma = 0; ma < NA; ++ma) load_a(off0, kk0, 0, ma, ar[0][ma]);
  }
  constexpr int NRD = 2 * (NA + NB), NMF = 3 * NA * NB;   // LDS reads and MFMAs per chunk
#pragma unroll 1
  for (int tap = 0; tap < 9; ++tap) {
    const int tn = tap < 8 ? tap + 1 : 8;    // (past the end: a redundant re-read, unused)
    const int off_c = tap_off(tap), kk_c = tap_kk(tap), off_n = tap_off(tn), kk_n = tap_kk(tn);
#pragma unroll
    for (int c16 = 0; c16 < NC16; ++c16) {
      const int c = tap * NC16 + c16;
      if (c + 3 < G::NCHUNK) ring_issue<G::CHUNKB, T::NW>(wimg, c + 3, ring, (c + 3) & (kRingSlots - 1), wave, lane);
      // the next chunk's operands (past the last chunk: a redundant read of a landed slot / of the last tap)
      const int cur = c16 & 1, nxt = cur ^ 1;
      load_b(c + 1, br[nxt]);
#pragma unroll
      for (int ma = 0; ma < NA; ++ma) {
        if (c16 + 1 < NC16) load_a(off_c, kk_c, c16 + 1, ma, ar[nxt][ma]);
        else load_a(off_n, kk_n, 0, ma, ar[nxt][ma]);
      }
#pragma unroll
      for (int ma = 0; ma < NA; ++ma) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {   // smallest products first: (l, h) (h, l) (h, h); transposed (weights as the row operand)
          if constexpr ((ASD_RING_ABL & 4) != 0) { asm volatile("" ::"v"(br[cur][nb][0]), "v"(br[cur][nb][1]), "v"(ar[cur][ma][0]), "v"(ar[cur][ma][1])); continue; }
          acc[ma][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hf(br[cur][nb][0]), hf(ar[cur][ma][1]), acc[ma][nb], 0, 0, 0);
          acc[ma][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hf(br[cur][nb][1]), hf(ar[cur][ma][0]), acc[ma][nb], 0, 0, 0);
          acc[ma][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hf(br[cur][nb][0]), hf(ar[cur][ma][0]), acc[ma][nb], 0, 0, 0);
        }
      }
      // issue order: one of the chunk's NRD reads behind each of its first MFMAs (RSPREAD MFMAs per read), so the last read has the rest of
      // the chunk to come back in -- the first MFMA behind the barrier waits for ALL of them (the compiler's lgkmcnt(0))
      constexpr int RSPREAD = ASD_RING_RSPREAD;
      SchedStep<0, NRD, (NRD * RSPREAD < NMF ? NRD * RSPREAD : NMF)>::run();
      // chunk c + 2 has landed (this wave's share): everything but the youngest NWW instructions (chunk c + 3's) is complete; the barrier
      // publishes it.  (The slot refilled behind the barrier, c & 3, was read a chunk ago: those reads fed MFMAs that have issued.)
      if (c + 1 < G::NCHUNK) {
        if (c + 3 < G::NCHUNK) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(T::NWW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // (the library's rule, tools/check_barriers.py: a wave's LDS operations are complete before it enters a barrier -- the next chunk's
        // operand reads were issued in the first half of this chunk, so the wait is short)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (!(ASD_RING_ABL & 8)) __builtin_amdgcn_s_barrier();
      }
    }
  }
}

// chunks 0..2 of a layer into slots 0..2
template <class G, int NW>
__device__ __forceinline__ void ring_prologue(const uint8_t* __restrict__ wimg, uint8_t* ring, int wave, int lane) {
#pragma unroll
  for (int c = 0; c < 3 && c < G::NCHUNK; ++c) ring_issue<G::CHUNKB, NW>(wimg, c, ring, c, wave, lane);
}

// ---- image fill by LDS-DMA: one patch's activations (global, linear pair format [pixel][slot]) into the interior of a swizzled image ---------
template <class G, int NW>
__device__ __forceinline__ void image_issue(const uint8_t* __restrict__ src, uint8_t* img, int wave, int lane) {
  constexpr int NPD = 1024 / G::PIXB;            // pixels per DMA instruction
  constexpr int PER_ROW = G::H / NPD, NI = G::H * PER_ROW;
  static_assert(NPD >= 1 && G::H % NPD == 0 && NI % NW == 0, "DMA pieces");
  const int pj = lane / G::SLOTS, q = lane % G::SLOTS;   // pixel within the piece, LDS slot
  if (ASD_RING_ABL & 1) return;
#pragma unroll
  for (int i = 0; i < NI / NW; ++i) {
    const int piece = i * NW + wave;
    const int iy = piece / PER_ROW, ix0 = (piece % PER_ROW) * NPD;
    const int ix = ix0 + pj;
    const int qs = q ^ G::key(iy + 1, ix + 1);   // LDS slot q of pixel (iy + 1, ix + 1) holds the pixel's slot q ^ key
    dma16(src + ((size_t)(iy * G::H + ix) * G::SLOTS + qs) * 16, img + ((iy + 1) * G::PW + ix0 + 1) * G::PIXB);
  }
}
template <class G>
__device__ __forceinline__ void image_zero_halo(uint8_t* img, int t, int nth) {
  for (int i = t; i < G::NPIX * G::SLOTS; i += nth) {
    const int p = i / G::SLOTS, y = p / G::PW, x = p % G::PW;
    if (y >= 1 && y <= G::H && x >= 1) continue;   // interior (x runs over 0 .. H: x = 0 is a halo pixel)
    *reinterpret_cast<u32x4*>(img + (size_t)i * 16) = u32x4{0, 0, 0, 0};
  }
}

// ---- epilogues -------------------------------------------------------------------------------------------------------------------------------
// bias + ReLU + the fp16 split of kActScale * v (k_conv_x3's PAIR epilogue): lane (kg, lr) holds couts co0 .. co0 + 3 of pixel lr of the sub-tile;
// after the swap an even-kg lane holds the h piece of the group of eight couts, an odd-kg lane the l piece: one 16-B value for slot
// (co0 >> 3) * 2 + (kg & 1)
__device__ __forceinline__ u32x4 finish4(const accv& a, const f32x4& bv, float out_scale) {
  float v[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { const float x = a[r] * out_scale + bv[r]; v[r] = x < 0.f ? 0.f : x; }
  uint32_t hh[2], ll[2];
  split4_mix_r(v[0], v[1], v[2], v[3], kActScaleR, hh, ll);
  const auto s0 = __builtin_amdgcn_permlane16_swap(hh[0], ll[0], false, false);
  const auto s1 = __builtin_amdgcn_permlane16_swap(hh[1], ll[1], false, false);
  return u32x4{s0[0], s1[0], s0[1], s1[1]};
}
// -> global memory, linear pair format [patch][pixel][slot]; patch0 + pp >= n is skipped
template <class G, int PP, int WM, int WN>
__device__ __forceinline__ void epilogue_global(const accv (&acc)[Tile<G, PP, WM, WN>::NA][Tile<G, PP, WM, WN>::NB], const float* __restrict__ bias,
                                                float out_scale, uint8_t* __restrict__ out, int patch0, int n, int wave, int lane) {
  using T = Tile<G, PP, WM, WN>;
  const int wm = wave / WN, wn = wave % WN, kg = lane >> 4, lr = lane & 15;
#pragma unroll
  for (int nb = 0; nb < T::NB; ++nb) {
    const int co0 = wn * T::NCW + nb * 16 + 4 * kg;
    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + co0);
#pragma unroll
    for (int ma = 0; ma < T::NA; ++ma) {
      const int m = wm * T::MW + ma * 16 + lr;
      const int pp = m / (G::HO * G::HO), mm = m % (G::HO * G::HO);
      const u32x4 v = finish4(acc[ma][nb], bv, out_scale);
      if (patch0 + pp < n && (!(ASD_RING_ABL & 2) || v[0] == 0x12345u))
        *reinterpret_cast<u32x4*>(out + ((size_t)(patch0 + pp) * G::HO * G::HO + mm) * G::COUT * 4 + (co0 >> 3) * 32 + (kg & 1) * 16) = v;
    }
  }
}
// -> the interior of the NEXT layer's LDS image (geometry GN: H = G::HO, CIN = G::COUT), one patch per image
template <class G, class GN, int PP, int WM, int WN>
__device__ __forceinline__ void epilogue_image(const accv (&acc)[Tile<G, PP, WM, WN>::NA][Tile<G, PP, WM, WN>::NB], const float* __restrict__ bias,
                                               float out_scale, uint8_t* __restrict__ img, int wave, int lane) {
  using T = Tile<G, PP, WM, WN>;
  static_assert(GN::H == G::HO && GN::CIN == G::COUT, "the next layer's image takes this layer's output");
  const int wm = wave / WN, wn = wave % WN, kg = lane >> 4, lr = lane & 15;
#pragma unroll
  for (int nb = 0; nb < T::NB; ++nb) {
    const int co0 = wn * T::NCW + nb * 16 + 4 * kg;
    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + co0);
#pragma unroll
    for (int ma = 0; ma < T::NA; ++ma) {
      const int m = wm * T::MW + ma * 16 + lr;
      const int pp = m / (G::HO * G::HO), mm = m % (G::HO * G::HO);
      const int y = mm / G::HO + 1, x = mm % G::HO + 1;
      const u32x4 v = finish4(acc[ma][nb], bv, out_scale);
      const int s = (co0 >> 3) * 2 + (kg & 1);
      *reinterpret_cast<u32x4*>(img + pp * GN::IMGB + (y * GN::PW + x) * GN::PIXB + ((s ^ GN::key(y, x)) << 4)) = v;
    }
  }
}

// ---- layer kernels -----------------------------------------------------------------------------------------------------------------------------
// One layer, whole patches: image(s) by LDS-DMA, ring loop, output to global memory.  Persistent over tiles (grid = what fits the chip).
template <int CIN, int COUT, int H, int S, int PP, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN, WM * WN / 4) void k_conv_ring(const uint8_t* __restrict__ in, const uint8_t* __restrict__ wimg, const float* __restrict__ bias,
                                                              uint8_t* __restrict__ out, int n, float out_scale) {
  using G = Geo<CIN, COUT, H, S>;
  using T = Tile<G, PP, WM, WN>;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem_r[];
  uint8_t* img = smem_r;                       // PP images
  uint8_t* ring = smem_r + PP * G::IMGB;       // kRingSlots x CHUNKB
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  for (int pp = 0; pp < PP; ++pp) image_zero_halo<G>(img + pp * G::IMGB, t, T::NTH);
  const int ntiles = (n + PP - 1) / PP;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int patch0 = tile * PP;
    // every wave has finished with the previous tile's image and ring (its LDS reads are complete: the loop's last MFMAs consumed them) -- the
    // barrier makes that true for ALL waves before anything is overwritten
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int pp = 0; pp < PP; ++pp)
      if (patch0 + pp < n) image_issue<G, T::NW>(in + (size_t)(patch0 + pp) * H * H * CIN * 4, img + pp * G::IMGB, wave, lane);
    ring_prologue<G, T::NW>(wimg, ring, wave, lane);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    accv acc[T::NA][T::NB];
#pragma unroll
    for (int ma = 0; ma < T::NA; ++ma)
#pragma unroll
      for (int nb = 0; nb < T::NB; ++nb) acc[ma][nb] = accv{0.f, 0.f, 0.f, 0.f};
    conv_ring_loop<G, PP, WM, WN>(img, ring, wimg, acc, wave, lane);
    epilogue_global<G, PP, WM, WN>(acc, bias, out_scale, out, patch0, n, wave, lane);
  }
}

// conv4 -> conv5 fused (ASDNet.py:343-349): one patch per tile; conv4's output is written into the image in place (as conv5's input, with conv5's
// swizzle), conv5's output goes to global memory.  Two waves split pixels, two split couts in both layers.
template <int WM4, int WN4, int WM5, int WN5>
__global__ __launch_bounds__(64 * WM4 * WN4, WM4 * WN4 / 4) void k_conv45_ring(const uint8_t* __restrict__ in, const uint8_t* __restrict__ w4, const float* __restrict__ b4, float os4,
                                                        const uint8_t* __restrict__ w5, const float* __restrict__ b5, float os5, uint8_t* __restrict__ out, int n) {
  using G4 = Geo<64, 64, 16, 1>;
  using G5 = Geo<64, 128, 16, 2>;
  using T4 = Tile<G4, 1, WM4, WN4>;
  using T5 = Tile<G5, 1, WM5, WN5>;
  static_assert(T4::NW == T5::NW && G4::IMGB == G5::IMGB, "one image region, one set of waves");
  constexpr int NW = T4::NW;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem_r[];
  uint8_t* img = smem_r;
  uint8_t* ring = smem_r + G4::IMGB;           // kRingSlots x max(CHUNKB) = 4 x 16 KB
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  image_zero_halo<G4>(img, t, 64 * NW);
  for (int patch = blockIdx.x; patch < n; patch += gridDim.x) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    image_issue<G4, NW>(in + (size_t)patch * 16 * 16 * 64 * 4, img, wave, lane);
    ring_prologue<G4, NW>(w4, ring, wave, lane);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    {
      accv acc[T4::NA][T4::NB];
#pragma unroll
      for (int ma = 0; ma < T4::NA; ++ma)
#pragma unroll
        for (int nb = 0; nb < T4::NB; ++nb) acc[ma][nb] = accv{0.f, 0.f, 0.f, 0.f};
      conv_ring_loop<G4, 1, WM4, WN4>(img, ring, w4, acc, wave, lane);
      // all of conv4's operand reads are complete in every wave before the image is rewritten; conv5's first weight chunks go out meanwhile
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      ring_prologue<G5, NW>(w5, ring, wave, lane);
      epilogue_image<G4, G5, 1, WM4, WN4>(acc, b4, os4, img, wave, lane);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    {
      accv acc[T5::NA][T5::NB];
#pragma unroll
      for (int ma = 0; ma < T5::NA; ++ma)
#pragma unroll
        for (int nb = 0; nb < T5::NB; ++nb) acc[ma][nb] = accv{0.f, 0.f, 0.f, 0.f};
      conv_ring_loop<G5, 1, WM5, WN5>(img, ring, w5, acc, wave, lane);
      epilogue_global<G5, 1, WM5, WN5>(acc, b5, os5, out, patch, n, wave, lane);
    }
  }
}

template <class K>
hipError_t set_lds(K kern, int lds, AsdPerDeviceOnce& once) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!once.need(dev)) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (e == hipSuccess) once.done(dev);
  return e;
}
}  // namespace

// ---- launchers (asdnet.hip calls these when the context runs the two-piece pair form) -------------------------------------------------------
// layer: 3 = conv4, 5 = conv6 (indices of ctx->d_wx2 / d_bias / wx2_scale)
int asdnet_ring_conv(asd_ctx* ctx, int layer, const void* in, void* out, int n, hipStream_t st) {
  const float os = 1.f / (kActScaleR * ctx->wx2_scale[layer]);
  const uint8_t* w = static_cast<const uint8_t*>(ctx->d_wx2[layer]);
  if (layer == 5) {
    constexpr int PP = 2;
    using G = Geo<128, 128, 8, 1>;
    auto kern = k_conv_ring<128, 128, 8, 1, PP, 2, 4>;
    constexpr int lds = PP * G::IMGB + kRingSlots * G::CHUNKB;
    static_assert(lds <= 160 * 1024, "LDS");
    static AsdPerDeviceOnce once;
    ASD_HIP_CHECK(ctx, set_lds(kern, lds, once));
    const int ntiles = (n + PP - 1) / PP;
    hipLaunchKernelGGL(kern, dim3(std::min(ntiles, ctx->num_cu)), dim3(512), lds, st, static_cast<const uint8_t*>(in), w, ctx->d_bias[layer],
                       static_cast<uint8_t*>(out), n, os);
  } else if (layer == 3) {
    using G = Geo<64, 64, 16, 1>;
    auto kern = k_conv_ring<64, 64, 16, 1, 1, 4, 2>;
    constexpr int lds = G::IMGB + kRingSlots * G::CHUNKB;
    static AsdPerDeviceOnce once;
    ASD_HIP_CHECK(ctx, set_lds(kern, lds, once));
    hipLaunchKernelGGL(kern, dim3(std::min(n, ctx->num_cu)), dim3(512), lds, st, static_cast<const uint8_t*>(in), w, ctx->d_bias[layer],
                       static_cast<uint8_t*>(out), n, os);
  } else {
    ctx->set_error("asdnet_ring_conv: no ring kernel for layer %d", layer);
    return ASD_ERR_INVALID;
  }
  ASD_HIP_CHECK(ctx, hipGetLastError());
  return ASD_OK;
}

// conv4 + conv5 in one launch: in = conv3's output, out = conv5's output (both global, linear pair format)
int asdnet_ring_conv45(asd_ctx* ctx, const void* in, void* out, int n, hipStream_t st) {
  using G4 = Geo<64, 64, 16, 1>;
  using G5 = Geo<64, 128, 16, 2>;
  auto kern = k_conv45_ring<4, 2, 2, 4>;
  constexpr int lds = G4::IMGB + kRingSlots * G5::CHUNKB;
  static_assert(lds <= 160 * 1024, "LDS");
  static AsdPerDeviceOnce once;
  ASD_HIP_CHECK(ctx, set_lds(kern, lds, once));
  hipLaunchKernelGGL(kern, dim3(std::min(n, ctx->num_cu)), dim3(512), lds, st, static_cast<const uint8_t*>(in), static_cast<const uint8_t*>(ctx->d_wx2[3]),
                     ctx->d_bias[3], 1.f / (kActScaleR * ctx->wx2_scale[3]), static_cast<const uint8_t*>(ctx->d_wx2[4]), ctx->d_bias[4],
                     1.f / (kActScaleR * ctx->wx2_scale[4]), static_cast<uint8_t*>(out), n);
  ASD_HIP_CHECK(ctx, hipGetLastError());
  return ASD_OK;
}
