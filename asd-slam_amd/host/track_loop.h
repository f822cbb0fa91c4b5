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
  int32_t stereo_matched, has_stereo;   /* stereo mode: Frame::ComputeStereoMatches' surviving matches of the frame */
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
/* the stand-in local map holds `copies` candidates per point of the last frame (default 2; bench.py's local_map_sweep: 2 / 4 / 8 = 4 k / 8 k / 16 k candidates) */
void asd_track_set_map_copies(asd_track_handle* h, int32_t copies);
/* 1 = the frame pointers given to asd_track_create are page-locked HOST memory: every frame's image goes host -> device inside the step
 * (asd_extract_submit(device_resident = 0)), as kitti.cc:116-155 hands images over; 0 (default) = frames resident in HBM */
void asd_track_set_frames_on_host(asd_track_handle* h, int32_t on);
/* Stereo mode (BASELINE configs[3]): ctx_right = the right image's extractor context (same device, weights loaded), d_frames_right =
 * the right images resident in HBM (as many as the left ones); every frame is extracted on both contexts (both read-ahead queues kept
 * full), the right frame goes into a slot of the left context (asd_frame_set_from_ctx) and Frame::ComputeStereoMatches (Frame.cc:360-535)
 * runs as asd_stereo_match during frame construction, before the left frame is tracked.  mb = baseline (m), mbf = baseline * fx. */
int asd_track_set_stereo(asd_track_handle* h, asd_ctx* ctx_right, const void* const* d_frames_right, float mb, float mbf);
/* waits for every read-ahead extraction this handle has outstanding and forgets its last frame: another handle on the same context may
 * then run (the next asd_track_run of this one starts like a first frame) */
int asd_track_drain(asd_track_handle* h);
/* accumulated since asd_track_create: wall time inside LocalBA, time blocked on the extractor (ms), steps run */
void asd_track_get_times(const asd_track_handle* h, double* ba_ms, double* extract_wait_ms, int64_t* steps);
/* where the stand-in map points of frame t land in frame t+1: u' = (u - cx) z + cx - dx z, v' = (v - cy) z + cy - dy z.  Default = the
 * synthetic stream of synth.scene_frame (cx 620.5, cy 188, z 1.003, dx 3, dy 0.2); a real sequence uses z = 1, dx = dy = 0. */
void asd_track_set_drift(asd_track_handle* h, float cx, float cy, float z, float dx, float dy);
/* LocalMapping::DoMapping's per-keyframe work in front of LocalBundleAdjustment (LocalMapping.cc:59-113: CreateNewMapPoints :299-545,
 * SearchInNeighbors :557-636, ComputeDistinctiveDescriptors of the touched points, MapPoint.cc:271-338) as the three batched submissions of
 * the library -- run at every keyframe, before asd_local_ba, when set (NULL = off, the default: the metric is tracking + LocalBA).  All
 * arrays are the caller's and must outlive the handle: a stand-in neighbourhood (current keyframe + its covisible keyframes resident in
 * frame slots, their FeatureVectors set) like the nominal LocalBA problem is a stand-in map. */
typedef struct asd_do_mapping_inputs {
  int32_t slot_cur; const uint8_t* has_mp_cur; const float* Tcw_cur; const float* K_cur;
  int32_t n_nb; const asd_kf_neighbor* nb;
  int32_t n_fuse_calls; const asd_fuse_call* fuse_calls;
  int32_t n_fuse_total; const uint8_t* valid; const float* Xw; const float* normal; const float* min_dist; const float* max_dist;
  const int32_t* desc_rows; float th;
  int32_t n_sets; const int32_t* set_start; const float* set_desc;
} asd_do_mapping_inputs;
void asd_track_set_do_mapping(asd_track_handle* h, const asd_do_mapping_inputs* in, int32_t n_cur /* keypoints of slot_cur */);
/* accumulated wall time inside that stage (ms) and the number of times it ran */
void asd_track_get_do_mapping_times(const asd_track_handle* h, double* ms, int64_t* calls);
/* n frames t0 .. t0+n-1; frames after the last one are read ahead only when the replay continues (prefetch_beyond) */
int asd_track_run(asd_track_handle* h, int32_t t0, int32_t n, int32_t prefetch_beyond, asd_track_stats* stats);

#ifdef __cplusplus
}
#endif
