// track_loop.h -- C entry points of libasdtrack.so (host/track_loop.cpp): the per-frame tracking step of bench.py as C++ host code
// over the C ABI of include/asd_slam.h.  Used by bench.py (ctypes), by host/asd_replay --chain and by tests.
#pragma once
#include <stdint.h>

#include "../../include/asd_slam.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct asd_track_stats {
  int32_t n_kp, m1, m2, inliers;
  double ba_chi2;
  int32_t has_m1, has_m2, has_inliers, has_ba;
} asd_track_stats;

typedef struct asd_track_handle asd_track_handle;

/* frames: n_frames device pointers to W x H u8 images resident in HBM (cycled: frame t = d_frames[t % n_frames]); K32 = fx fy cx cy;
 * T = the predicted Tcw the motion-model search projects with, pose0 = the pose PoseOptimization starts from; inv_sigma2 / scale32 =
 * the extractor's level tables; ba = the nominal LocalBA problem (arrays owned by the caller for the handle's lifetime);
 * lookahead = frames of read-ahead extraction (0 .. ASD_EXTRACT_QUEUE). */
asd_track_handle* asd_track_create(asd_ctx* ctx, int32_t n_frames, const void* const* d_frames, int32_t W, int32_t H, const float* K32,
                                   const float* T, const double* pose0, const double* inv_sigma2, const float* scale32,
                                   const asd_ba_problem* ba, int32_t kf_interval, int32_t lookahead);
void asd_track_destroy(asd_track_handle* h);
void asd_track_set_fused(asd_track_handle* h, int32_t on);
void asd_track_set_async_ba(asd_track_handle* h, int32_t on);   /* 1 = LocalBA on the optional lane (NOT the reference's order) */
void asd_track_set_split(asd_track_handle* h, int32_t on);
/* 1 (default, with fused + split): both stages as one submission (asd_track_frame), the next frame constructed on the context's second
 * stream beside them (asd_prep_async); 0 = two submissions with the host in between */
void asd_track_set_chain(asd_track_handle* h, int32_t on);
/* 1 = the frame pointers given to asd_track_create are page-locked HOST memory: every frame's image goes host -> device inside the step
 * (asd_extract_submit(device_resident = 0)), as kitti.cc:116-155 hands images over; 0 (default) = frames resident in HBM */
void asd_track_set_frames_on_host(asd_track_handle* h, int32_t on);
/* waits for every read-ahead extraction this handle has outstanding and forgets its last frame: another handle on the same context may
 * then run (the next asd_track_run of this one starts like a first frame) */
int asd_track_drain(asd_track_handle* h);
/* accumulated since asd_track_create: wall time inside LocalBA, time blocked on the extractor (ms), steps run */
void asd_track_get_times(const asd_track_handle* h, double* ba_ms, double* extract_wait_ms, int64_t* steps);
/* where the stand-in map points of frame t land in frame t+1: u' = (u - cx) z + cx - dx z, v' = (v - cy) z + cy - dy z.  Default = the
 * synthetic stream of synth.scene_frame (cx 620.5, cy 188, z 1.003, dx 3, dy 0.2); a real sequence uses z = 1, dx = dy = 0. */
void asd_track_set_drift(asd_track_handle* h, float cx, float cy, float z, float dx, float dy);
/* n frames t0 .. t0+n-1; frames after the last one are read ahead only when the replay continues (prefetch_beyond) */
int asd_track_run(asd_track_handle* h, int32_t t0, int32_t n, int32_t prefetch_beyond, asd_track_stats* stats);

#ifdef __cplusplus
}
#endif
