// capi.cpp -- C ABI entry points of libasdhip: context, ASDNet, utilities.
// (extractor: frontend.hip, matchers: matcher.hip, optimizer: ba.hip)
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>

#include "ctx.h"

int frontend_alloc(asd_ctx* ctx);
void frontend_free(asd_ctx* ctx);
void matcher_free(asd_ctx* ctx);
void ba_free(asd_ctx* ctx);
void mapping_free(asd_ctx* ctx);
void bow_free(asd_ctx* ctx);

namespace {
inline int cv_round(double v) { return (int)std::lrint(v); }  // cvRound: round-half-to-even
inline int cv_floor(double v) { return (int)std::floor(v); }
inline int cv_ceil(double v) { return (int)std::ceil(v); }

// ORBextractor::ORBextractor (ORBextractor.cc:452-512): scale tables, per-level quotas, umax.
void build_tables(asd_ctx* c) {
  const int nl = c->cfg.n_levels;
  const float scaleFactor = c->cfg.scale_factor;
  c->scale[0] = 1.0f;
  c->sigma2[0] = 1.0f;
  for (int i = 1; i < nl; i++) {
    c->scale[i] = (float)(c->scale[i - 1] * (double)scaleFactor);  // member scaleFactor is a double (ORBextractor.h:102)
    c->sigma2[i] = c->scale[i] * c->scale[i];
  }
  for (int i = 0; i < nl; i++) {
    c->inv_scale[i] = 1.0f / c->scale[i];
    c->inv_sigma2[i] = 1.0f / c->sigma2[i];
  }
  {  // level_thr (see ctx.h): bisection is valid because ceil(logf(r) / ls) does not decrease with r (asd_debug_level_sweep checks it)
    const float ls = std::log(scaleFactor);  // Frame.cc:74: float log of the float scale factor
    auto level_of = [&](float r) { return (int)std::ceil(std::log(r) / ls); };
    c->level_thr[0] = 0.f;
    for (int k = 1; k < ASD_MAX_LEVELS; ++k) {
      uint32_t lo, hi;  // bit patterns of positive floats order like the values
      const float flo = 0.25f, fhi = 1.0e30f;
      memcpy(&lo, &flo, 4); memcpy(&hi, &fhi, 4);
      while (lo < hi) {  // smallest pattern with level_of >= k
        const uint32_t mid = lo + (hi - lo) / 2;
        float fm; memcpy(&fm, &mid, 4);
        if (level_of(fm) >= k) hi = mid; else lo = mid + 1;
      }
      memcpy(&c->level_thr[k], &lo, 4);
    }
  }
  const int nfeatures = c->cfg.n_features;
  float factor = (float)(1.0f / (double)scaleFactor);
  float nDesired = nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nl));
  int sum = 0;
  for (int level = 0; level < nl - 1; level++) {
    c->features_per_level[level] = cv_round(nDesired);
    sum += c->features_per_level[level];
    nDesired *= factor;
  }
  c->features_per_level[nl - 1] = std::max(nfeatures - sum, 0);
  const int HP = 15;
  int v, v0, vmax = cv_floor(HP * std::sqrt(2.f) / 2 + 1);
  int vmin = cv_ceil(HP * std::sqrt(2.f) / 2);
  const double hp2 = HP * HP;
  for (v = 0; v <= vmax; ++v) c->umax[v] = cv_round(std::sqrt(hp2 - v * v));
  for (v = HP, v0 = 0; v >= vmin; --v) {
    while (c->umax[v0] == c->umax[v0 + 1]) ++v0;
    c->umax[v] = v0;
    ++v0;
  }
}
}  // namespace

void asd_compute_quotas(int nfeatures, float scaleFactor, int nl, int* out) {
  float factor = (float)(1.0f / (double)scaleFactor);
  float nDesired = nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nl));
  int sum = 0;
  for (int level = 0; level < nl - 1; level++) {
    out[level] = cv_round(nDesired);
    sum += out[level];
    nDesired *= factor;
  }
  out[nl - 1] = std::max(nfeatures - sum, 0);
}

extern "C" {

const char* asd_version(void) { return "asdhip 0.1 (gfx950)"; }

static thread_local std::string g_create_error;

int asd_ctx_create(const asd_config* cfg, asd_ctx** out) {
  if (!cfg || !out) return ASD_ERR_INVALID;
  *out = nullptr;
  if (cfg->n_levels < 1 || cfg->n_levels > ASD_MAX_LEVELS || cfg->n_features < 1 || cfg->scale_factor <= 1.0f ||
      cfg->max_width < 64 || cfg->max_height < 64 || cfg->max_patches < cfg->n_features)
    return ASD_ERR_INVALID;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device < 0 || cfg->device >= ndev) {
    fprintf(stderr, "libasdhip: no usable HIP device (count=%d, requested %d); there is no CPU fallback\n", ndev,
            cfg->device);
    return ASD_ERR_NO_DEVICE;
  }
  if (hipSetDevice(cfg->device) != hipSuccess) return ASD_ERR_NO_DEVICE;
  asd_ctx* c = new (std::nothrow) asd_ctx();
  if (!c) return ASD_ERR_INVALID;
  c->cfg = *cfg;
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg->device) == hipSuccess && prop.multiProcessorCount > 0) c->num_cu = prop.multiProcessorCount;
  }
  build_tables(c);
  // ASDNet arithmetic: split-operand kernels by default (two fp16 terms; ASD_ASDNET_MATH=bf16x3 for three bf16 terms);
  // ASD_ASDNET_MATH=f32 keeps every layer on the f32 MFMA kernels,
  // ASD_ASDNET_SPLIT_LAYERS=<mask> picks layers (bit 0 = conv2 ... bit 4 = conv6, bit 5 = fc)
  c->net_split = 0x3f;
  if (const char* e = getenv("ASD_MATCH_REPLAY")) c->match_replay_host = !strcmp(e, "host");   // matcher.hip, k_resolve
  if (const char* e = getenv("ASD_ASDNET_MATH")) {
    if (!strcmp(e, "f32")) c->net_split = 0;
    else if (!strcmp(e, "f16x2")) c->net_pieces = 2;
    else if (!strcmp(e, "bf16x3")) c->net_pieces = 3;
    else if (strcmp(e, "split")) fprintf(stderr, "libasdhip: ASD_ASDNET_MATH=%s not understood (f32 | split | f16x2 | bf16x3); using f16x2\n", e);
  }
  c->net_pieces_req = c->net_pieces;
  if (const char* e = getenv("ASD_ASDNET_SPLIT_LAYERS")) c->net_split = (int)strtol(e, nullptr, 0) & 0x3f;
  // tracking kernels (small, latency critical) outrank the pipelined extractor's stream
  int prio_least = 0, prio_greatest = 0;
  (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
  if (hipStreamCreateWithPriority(&c->stream, hipStreamDefault, prio_greatest) != hipSuccess || hipEventCreate(&c->ev0) != hipSuccess ||
      hipEventCreate(&c->ev1) != hipSuccess || hipEventCreate(&c->ev2) != hipSuccess) {
    delete c;
    return ASD_ERR_NO_DEVICE;
  }
  int rc = asdnet_alloc(c);
  if (rc == ASD_OK) rc = frontend_alloc(c);
  if (rc != ASD_OK) {
    fprintf(stderr, "libasdhip: %s\n", c->err.c_str());
    asd_ctx_destroy(c);
    return rc;
  }
  *out = c;
  return ASD_OK;
}

int asd_ctx_destroy(asd_ctx* ctx) {
  if (!ctx) return ASD_ERR_INVALID;
  (void)hipSetDevice(ctx->cfg.device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  if (ctx->stream_prep) (void)hipStreamSynchronize(ctx->stream_prep);
  frontend_async_shutdown(ctx);
  asdnet_free(ctx);
  frontend_free(ctx);
  matcher_free(ctx);
  ba_free(ctx);
  mapping_free(ctx);
  bow_free(ctx);
  ctx->scratch.release();
  ctx->stereo_scratch.release();
  ctx->up.release();
  ctx->down.release();
  if (ctx->ev_chain) (void)hipEventDestroy(ctx->ev_chain);
  if (ctx->ev_adopt) (void)hipEventDestroy(ctx->ev_adopt);
  if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
  if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
  if (ctx->ev2) (void)hipEventDestroy(ctx->ev2);
  for (int set = 0; set < 2; ++set)
    for (int i = 0; i < 9; ++i) if (ctx->prof_ev[set][i]) (void)hipEventDestroy(ctx->prof_ev[set][i]);
  if (ctx->ev_prep) (void)hipEventDestroy(ctx->ev_prep);
  if (ctx->stream_prep) (void)hipStreamDestroy(ctx->stream_prep);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return ASD_OK;
}

const char* asd_last_error(const asd_ctx* ctx) { return ctx ? const_cast<asd_ctx*>(ctx)->last_error() : "null context"; }

int asd_get_scale_tables(const asd_ctx* ctx, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2,
                         int32_t* fpl) {
  if (!ctx) return ASD_ERR_INVALID;
  const int nl = ctx->cfg.n_levels;
  for (int i = 0; i < nl; ++i) {
    if (scale) scale[i] = ctx->scale[i];
    if (inv_scale) inv_scale[i] = ctx->inv_scale[i];
    if (sigma2) sigma2[i] = ctx->sigma2[i];
    if (inv_sigma2) inv_sigma2[i] = ctx->inv_sigma2[i];
    if (fpl) fpl[i] = ctx->features_per_level[i];
  }
  return ASD_OK;
}

int asd_load_weights(asd_ctx* ctx, const float* const conv_w[7], const float* const bn_mean[7],
                     const float* const bn_var[7], float bn_eps) {
  if (!ctx || !conv_w || !bn_mean || !bn_var) return ASD_ERR_INVALID;
  for (int i = 0; i < 7; ++i)
    if (!conv_w[i] || !bn_mean[i] || !bn_var[i]) { ctx->set_error("null weight tensor %d", i); return ASD_ERR_INVALID; }
  (void)hipSetDevice(ctx->cfg.device);
  return asdnet_load_weights(ctx, conv_w, bn_mean, bn_var, bn_eps);
}

int asd_describe_device(asd_ctx* ctx, const uint8_t* d_patches, int32_t n, float* d_desc) {
  if (!ctx || (n > 0 && (!d_patches || !d_desc))) return ASD_ERR_INVALID;
  if (asd_extractor_busy(ctx, "asd_describe_device")) return ASD_ERR_INVALID;
  return asdnet_forward_device(ctx, d_patches, n, d_desc, ctx->stream, ctx->h_range);   // (no synchronisation here: asd_sync reports the flag)
}

int asd_describe(asd_ctx* ctx, const uint8_t* patches, int32_t n, float* desc) {
  if (!ctx || n < 0 || (n > 0 && (!patches || !desc))) return ASD_ERR_INVALID;
  if (n > ctx->cfg.max_patches) { ctx->set_error("n=%d exceeds max_patches=%d", n, ctx->cfg.max_patches); return ASD_ERR_CAPACITY; }
  if (!ctx->weights_loaded) { ctx->set_error("asd_load_weights has not been called"); return ASD_ERR_NO_WEIGHTS; }
  if (n == 0) return ASD_OK;
  if (asd_extractor_busy(ctx, "asd_describe")) return ASD_ERR_INVALID;
  (void)hipSetDevice(ctx->cfg.device);
  ASD_HIP_CHECK(ctx, hipMemcpyAsync(ctx->d_patches, patches, (size_t)n * 1024, hipMemcpyHostToDevice, ctx->stream));
  ASD_HIP_CHECK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  *ctx->h_range = 0;
  int rc = asdnet_forward_device(ctx, ctx->d_patches, n, ctx->d_desc, ctx->stream, ctx->h_range);
  if (rc != ASD_OK) return rc;
  ASD_HIP_CHECK(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  ASD_HIP_CHECK(ctx, hipMemcpyAsync(desc, ctx->d_desc, (size_t)n * 128 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  ASD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  ASD_HIP_CHECK(ctx, hipEventElapsedTime(&ctx->ms_asdnet, ctx->ev0, ctx->ev1));
  return asd_range_status(ctx, ctx->h_range, "asd_describe");
}

int asd_describe_timed(asd_ctx* ctx, const uint8_t* d_patches, int32_t n, float* d_desc, int32_t reps, float* avg_ms) {
  if (!ctx || !avg_ms || reps < 1 || n < 1) return ASD_ERR_INVALID;
  if (asd_extractor_busy(ctx, "asd_describe_timed")) return ASD_ERR_INVALID;
  (void)hipSetDevice(ctx->cfg.device);
  ASD_HIP_CHECK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  for (int r = 0; r < reps; ++r) {
    int rc = asdnet_forward_device(ctx, d_patches, n, d_desc, ctx->stream);
    if (rc != ASD_OK) return rc;
  }
  ASD_HIP_CHECK(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  ASD_HIP_CHECK(ctx, hipEventSynchronize(ctx->ev1));
  float ms = 0;
  ASD_HIP_CHECK(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
  *avg_ms = ms / reps;
  ctx->ms_asdnet = *avg_ms;
  return ASD_OK;
}

int asd_last_stage_ms(const asd_ctx* ctx, const char* stage, float* ms) {
  if (!ctx || !stage || !ms) return ASD_ERR_INVALID;
  if (!strcmp(stage, "asdnet")) *ms = ctx->ms_asdnet;
  else if (!strcmp(stage, "extract")) *ms = ctx->ms_extract;
  else if (!strcmp(stage, "match")) *ms = ctx->ms_match;
  else if (!strcmp(stage, "ba")) *ms = ctx->ms_ba;
  else return ASD_ERR_INVALID;
  return ASD_OK;
}

int32_t asd_asdnet_split_mask(const asd_ctx* ctx) { return ctx ? (ctx->net_split & 0x3f) : 0; }
int32_t asd_asdnet_pieces(const asd_ctx* ctx) { return ctx ? ctx->net_pieces : 0; }
const char* asd_calibration_note(const asd_ctx* ctx) { return ctx ? ctx->calib_note.c_str() : ""; }

int asd_profile_enable(asd_ctx* ctx, int32_t on) {
  if (!ctx) return ASD_ERR_INVALID;
  (void)hipSetDevice(ctx->cfg.device);
  if (!on) { int rc = asdnet_profile_collect(ctx); if (rc != ASD_OK) return rc; }
  std::lock_guard<std::mutex> prof_lock(ctx->prof_mutex);
  if (on && !ctx->prof_ev[0][0])
    for (int set = 0; set < 2; ++set)
      for (int i = 0; i < 9; ++i) ASD_HIP_CHECK(ctx, hipEventCreate(&ctx->prof_ev[set][i]));
  ctx->prof_on = on != 0;
  if (on) for (int l = 0; l < 8; ++l) { ctx->prof_ms[l] = 0; ctx->prof_calls[l] = 0; ctx->prof_patches[l] = 0; }
  return ASD_OK;
}

int asd_profile_get(asd_ctx* ctx, int32_t layer, double* total_ms, int32_t* calls, int64_t* patches) {
  if (!ctx || layer < 0 || layer >= 8 || !total_ms || !calls || !patches) return ASD_ERR_INVALID;
  { int rc = asdnet_profile_collect(ctx); if (rc != ASD_OK) return rc; }
  std::lock_guard<std::mutex> prof_lock(ctx->prof_mutex);
  *total_ms = ctx->prof_ms[layer];
  *calls = ctx->prof_calls[layer];
  *patches = ctx->prof_patches[layer];
  return ASD_OK;
}

void* asd_ctx_stream(asd_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

// MapPoint::PredictScale's level by the comparison rule the device uses (ctx.h, level_thr)
static inline int level_by_thresholds(const asd_ctx* ctx, float ratio) {
  int s = 0;
  for (int k = 1; k < ctx->cfg.n_levels; ++k) s += ratio >= ctx->level_thr[k];
  return s;
}
int32_t asd_debug_level_sweep(const asd_ctx* ctx, float lo, float hi, int64_t* n_checked) {
  // every float in [lo, hi]: the threshold rule against ceil(logf(r) / logf(scaleFactor)) clamped to the level range
  if (!ctx || !(lo > 0.f) || !(hi >= lo) || !n_checked) return -1;
  const float ls = std::log(ctx->cfg.scale_factor);
  uint32_t a, b;
  memcpy(&a, &lo, 4); memcpy(&b, &hi, 4);
  int32_t bad = 0;
  for (uint32_t u = a;; ++u) {
    float r; memcpy(&r, &u, 4);
    int s = (int)std::ceil(std::log(r) / ls);
    if (s < 0) s = 0; else if (s >= ctx->cfg.n_levels) s = ctx->cfg.n_levels - 1;
    bad += s != level_by_thresholds(ctx, r);
    if (u == b) break;
  }
  *n_checked = (int64_t)b - (int64_t)a + 1;
  return bad;
}

int asd_device_alloc(asd_ctx* ctx, uint64_t bytes, void** dptr) {
  if (!ctx || !dptr) return ASD_ERR_INVALID;
  (void)hipSetDevice(ctx->cfg.device);
  ASD_HIP_CHECK(ctx, hipMalloc(dptr, bytes));
  return ASD_OK;
}
int asd_device_free(asd_ctx* ctx, void* dptr) {
  if (!ctx) return ASD_ERR_INVALID;
  ASD_HIP_CHECK(ctx, hipFree(dptr));
  return ASD_OK;
}
int asd_extract_keep_pyramid(asd_ctx* ctx, int32_t on) {
  if (!ctx) return ASD_ERR_INVALID;
  if (asd_extractor_busy(ctx, "asd_extract_keep_pyramid")) return ASD_ERR_INVALID;
  ctx->keep_pyramid = on != 0;
  if (!on) ctx->d_pyr_view = nullptr;
  return ASD_OK;
}
int asd_host_alloc(asd_ctx* ctx, uint64_t bytes, void** hptr) {
  if (!ctx || !hptr) return ASD_ERR_INVALID;
  (void)hipSetDevice(ctx->cfg.device);
  ASD_HIP_CHECK(ctx, hipHostMalloc(hptr, bytes));
  return ASD_OK;
}
int asd_host_free(asd_ctx* ctx, void* hptr) {
  if (!ctx) return ASD_ERR_INVALID;
  ASD_HIP_CHECK(ctx, hipHostFree(hptr));
  return ASD_OK;
}
int asd_memcpy_h2d(asd_ctx* ctx, void* dst, const void* src, uint64_t bytes) {
  if (!ctx) return ASD_ERR_INVALID;
  ASD_HIP_CHECK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  ASD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return ASD_OK;
}
int asd_memcpy_d2h(asd_ctx* ctx, void* dst, const void* src, uint64_t bytes) {
  if (!ctx) return ASD_ERR_INVALID;
  ASD_HIP_CHECK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  ASD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return ASD_OK;
}
int asd_sync(asd_ctx* ctx) {
  if (!ctx) return ASD_ERR_INVALID;
  ASD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return ctx->h_range ? asd_range_status(ctx, ctx->h_range, "asd_describe_device") : ASD_OK;
}

// Converter::toSE3Quat (Converter.cc:37-47) + SE3Quat(R,t) ctor (se3quat.h:58-60): Eigen
// Quaterniond(Matrix3d) then normalizeRotation (w >= 0, unit norm).
int asd_tcw_to_pose7(const float* T, double* p) {
  if (!T || !p) return ASD_ERR_INVALID;
  double R[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) R[i][j] = (double)T[i * 4 + j];
  double q[4];  // x y z w
  const double tr = R[0][0] + R[1][1] + R[2][2];
  if (tr > 0) {  // Eigen quaternion_base_assign_impl<Other,3,3>
    double t = std::sqrt(tr + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (R[2][1] - R[1][2]) * t;
    q[1] = (R[0][2] - R[2][0]) * t;
    q[2] = (R[1][0] - R[0][1]) * t;
  } else {
    int i = 0;
    if (R[1][1] > R[0][0]) i = 1;
    if (R[2][2] > R[i][i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    double t = std::sqrt(R[i][i] - R[j][j] - R[k][k] + 1.0);
    q[i] = 0.5 * t;
    t = 0.5 / t;
    q[3] = (R[k][j] - R[j][k]) * t;
    q[j] = (R[j][i] + R[i][j]) * t;
    q[k] = (R[k][i] + R[i][k]) * t;
  }
  if (q[3] < 0) for (int i = 0; i < 4; ++i) q[i] = -q[i];
  const double nrm = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int i = 0; i < 4; ++i) p[i] = q[i] / nrm;
  p[4] = T[3];
  p[5] = T[7];
  p[6] = T[11];
  return ASD_OK;
}

// Converter::toCvMat(SE3Quat) (Converter.cc:57-71): to_homogeneous_matrix, cast to float.
int asd_pose7_to_tcw(const double* p, float* T) {
  if (!T || !p) return ASD_ERR_INVALID;
  const double x = p[0], y = p[1], z = p[2], w = p[3];
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y,
               tyz = tz * y, tzz = tz * z;
  const double R[9] = {1 - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1 - (txx + tzz), tyz - twx,
                       txz - twy,       tyz + twx, 1 - (txx + tyy)};
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) T[i * 4 + j] = (float)R[i * 3 + j];
    T[i * 4 + 3] = (float)p[4 + i];
  }
  T[12] = T[13] = T[14] = 0.f;
  T[15] = 1.f;
  return ASD_OK;
}

}  // extern "C"

bool asd_track_busy(asd_ctx* ctx, const char* who) {
  if (!ctx->track_has_pending) return false;
  ctx->set_error("%s: an asd_track_* call started with asd_track_async is outstanding -- asd_track_finish first (meanwhile only asd_extract_submit/"
                 "wait*, asd_frame_set on another slot, asd_bank_put* and asd_local_ba_submit/wait/poll may be called)", who);
  return true;
}
