/*
 * asd_slam.h -- C ABI of libasdhip: the MI355X (gfx950) implementation of ASD-SLAM's
 * per-frame front-end and local-BA hot path.
 *
 * The reference has no plugin / FFI layer: the hot path is reached through C++ methods
 * of libvslam with OpenCV types in every signature.  This header is the
 * flat-array boundary drawn INSIDE those methods; each entry point cites the reference
 * code it replaces (paths relative to the reference root, src/vslam/...).
 * INTEGRATION.md shows the adapter a maintainer adds on the reference side.
 *
 * Conventions
 *   - plain C types, caller-owned buffers, row-major, no torch / OpenCV types;
 *   - every call returns ASD_OK (0) or a negative asd_status; asd_last_error(ctx)
 *     returns a human readable message for the last failure on that context;
 *   - one asd_ctx = one HIP device + one HIP stream; a ctx is not thread-safe,
 *     different ctxs may be used from different threads / processes;
 *   - "host" pointers are ordinary CPU memory.  Results that the next stage consumes
 *     (pyramid, descriptors, grid) also stay resident in HBM inside the ctx;
 *   - there is NO CPU fallback: without a usable HIP device asd_ctx_create fails with
 *     ASD_ERR_NO_DEVICE.
 */
#ifndef ASD_SLAM_H
#define ASD_SLAM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ASD_DESC_DIM 128      /* ORBmatcher.cc:1638 assumes exactly 128 columns          */
#define ASD_PATCH 32          /* ORBextractor.cc:1115 Rect(x-16,y-16,32,32)               */
#define ASD_MAX_LEVELS 16
#define ASD_GRID_COLS 64      /* Frame.h:37  FRAME_GRID_COLS                              */
#define ASD_GRID_ROWS 48      /* Frame.h:38  FRAME_GRID_ROWS                              */
#define ASD_HISTO_LENGTH 30   /* ORBmatcher.cc:39                                         */

typedef enum asd_status {
  ASD_OK = 0,
  ASD_ERR_INVALID = -1,     /* bad argument (null pointer, size out of range)             */
  ASD_ERR_NO_DEVICE = -2,   /* no HIP device / HIP runtime failure at create              */
  ASD_ERR_HIP = -3,         /* a HIP call or kernel failed                                */
  ASD_ERR_NO_WEIGHTS = -4,  /* asd_load_weights was not called                            */
  ASD_ERR_CAPACITY = -5,    /* input exceeds the capacity given at asd_ctx_create         */
  ASD_ERR_NUMERIC = -6,     /* solver breakdown (non positive definite system)            */
  ASD_ERR_RANGE = -7        /* ASDNet (two-piece fp16 operand form): an activation left fp16's range, the
                               descriptors of this call are not valid (see asd_asdnet_pieces)            */
} asd_status;

typedef struct asd_ctx asd_ctx;

/* Replaces the ORBextractor constructor arguments (ORBextractor.cc:452-456; values given
 * in Tracking.cc:77-85) plus device selection. */
typedef struct asd_config {
  int32_t n_features;     /* --feature_count, 2000                                        */
  float   scale_factor;   /* --feature_scale_factor, 1.2                                  */
  int32_t n_levels;       /* --feature_level, 8                                           */
  int32_t ini_th_fast;    /* 20 (Tracking.cc:80)                                          */
  int32_t min_th_fast;    /* 7  (Tracking.cc:81)                                          */
  int32_t max_width;      /* largest image the ctx will see                               */
  int32_t max_height;
  int32_t max_patches;    /* capacity of asd_describe / extract (>= 2*n_features: init)   */
  int32_t device;         /* HIP device ordinal                                           */
} asd_config;

/* cv::KeyPoint as produced by ExtractDesc (ORBextractor.cc:887-898,1236-1242). */
typedef struct asd_keypoint {
  float   x, y;       /* level-0 pixel coordinates (pt * mvScaleFactor[octave])           */
  float   size;       /* (int)(31 * scale[octave]) stored as float                        */
  float   angle;      /* IC_Angle, degrees in [0,360)                                     */
  float   response;   /* FAST score                                                       */
  int32_t octave;
} asd_keypoint;

/* ---- context ------------------------------------------------------------------------ */
int asd_ctx_create(const asd_config* cfg, asd_ctx** out);
int asd_ctx_destroy(asd_ctx* ctx);
const char* asd_last_error(const asd_ctx* ctx);
const char* asd_version(void);

/* Scale tables of the extractor: ORBextractor.cc:459-492 (mvScaleFactor, mvLevelSigma2,
 * their inverses, mnFeaturesPerLevel).  Each output holds n_levels entries; any may be
 * NULL. */
int asd_get_scale_tables(const asd_ctx* ctx, float* scale, float* inv_scale, float* sigma2,
                         float* inv_sigma2, int32_t* features_per_level);

/* ---- ASDNet descriptor (E6) --------------------------------------------------------- */
/* Replaces torch::jit::load of bestmodel_c.pt (ORBextractor.cc:457-458).  Takes the 7
 * conv weights in torch layout [cout][cin][kh][kw] and the 7 BatchNorm running
 * mean/var vectors (affine=False, eps as given) of ASDNet.py:334-356; BN is folded into
 * the conv on upload. */
int asd_load_weights(asd_ctx* ctx, const float* const conv_w[7], const float* const bn_mean[7],
                     const float* const bn_var[7], float bn_eps);

/* Replaces computeSIFTDescriptors' forward pass (ORBextractor.cc:1125-1132 ->
 * ASDNet.forward, ASDNet.py:360-370): n u8 patches [n][32][32] -> n unit-norm
 * descriptors [n][128] f32.  Both pointers are host memory. */
int asd_describe(asd_ctx* ctx, const uint8_t* patches, int32_t n, float* desc);

/* Same, device resident: d_patches / d_desc are HIP device pointers; no copies and no
 * synchronisation -- the work is enqueued on the ctx stream (used by bench + extract). */
int asd_describe_device(asd_ctx* ctx, const uint8_t* d_patches, int32_t n, float* d_desc);

/* ---- extractor (E1-E7) -------------------------------------------------------------- */
/* Replaces ORBextractor::ExtractDesc(image, mask, keypoints, descriptors, use_orb=false)
 * (ORBextractor.cc:1137-1249): pyramid, FAST + quadtree, orientation, blur, patch gather,
 * ASDNet.  kps / desc have room for max_patches entries; *n_out receives the count.
 * n_features_override > 0 replaces cfg.n_features for this call (the reference keeps a
 * second extractor with 2*nFeatures for initialisation, Tracking.cc:85). */
int asd_extract(asd_ctx* ctx, const uint8_t* image, int32_t width, int32_t height, int32_t stride,
                int32_t n_features_override, asd_keypoint* kps, float* desc, int32_t* n_out);

/* Same with the image already resident in HBM (d_image = HIP device pointer, row stride in
 * bytes): what a camera driver with a device-side ring buffer, or bench.py, calls. */
int asd_extract_device(asd_ctx* ctx, const uint8_t* d_image, int32_t width, int32_t height, int32_t stride,
                       int32_t n_features_override, asd_keypoint* kps, float* desc, int32_t* n_out);

/* Pipelined extraction.  ExtractDesc of frame t+1 does not depend on the tracking of frame t, and a
 * replay reads its images from disk (Examples/Monocular/kitti.cc:116-155), so they can overlap:
 * asd_extract_submit queues an extraction for a worker thread with its own HIP streams and returns at
 * once; asd_extract_wait blocks until the OLDEST outstanding submission has finished and hands back the
 * same results asd_extract would.  Up to ASD_EXTRACT_QUEUE submissions may be outstanding; the worker runs
 * the front half (pyramid .. patch gather) of the next queued frame underneath the ASDNet pass of the
 * previous one.  `image` must stay valid until its wait.  While submissions are outstanding the worker owns the
 * shared front-end and ASDNet buffers: asd_extract*, asd_describe*, asd_get_level_image / asd_get_raw_corners and
 * asd_stereo_match return ASD_ERR_INVALID ("... submission(s) outstanding") instead of racing with it.  If the
 * extractor's streams / buffers cannot be created the call fails and the next submit starts over.  The device-resident
 * descriptors of a waited frame (asd_frame_set with desc == NULL) stay valid for two further submissions. */
#define ASD_EXTRACT_QUEUE 3
int asd_extract_submit(asd_ctx* ctx, const uint8_t* image, int32_t device_resident, int32_t width,
                       int32_t height, int32_t stride, int32_t n_features_override);
int asd_extract_wait(asd_ctx* ctx, asd_keypoint* kps, float* desc, int32_t* n_out);
/* Same without the copy: *kps / *desc point into the library's own result buffers of that submission (the
 * descriptors sit in pinned host memory) and stay valid until two further submissions have been made -- e.g. to
 * wrap them as the Frame's mvKeys / mDescriptors (cv::Mat header over foreign data) for the frame's lifetime in
 * the tracker, which is two frames. */
int asd_extract_wait_view(asd_ctx* ctx, const asd_keypoint** kps, const float** desc, int32_t* n_out);
/* The lifetime of a view, checkable: asd_extract_last_view returns the id of the view handed out by the most recent
 * asd_extract_wait / asd_extract_wait_view (ids count submissions from 0), and asd_extract_view_valid(ctx, id) is 1 as long
 * as no later submission has been given that view's buffers -- exactly: while fewer than ASD_EXTRACT_QUEUE + 2 submissions
 * have been made after the one the view belongs to -- and 0 afterwards (or for an id that was never handed out).  A caller
 * that keeps a view across frames (mCurrentFrame / mLastFrame) can assert this where it dereferences it. */
uint64_t asd_extract_last_view(const asd_ctx* ctx);
int32_t asd_extract_view_valid(asd_ctx* ctx, uint64_t view_id);

/* Intermediate products of the last asd_extract, for tests and for callers that read
 * ORBextractor::mvImagePyramid (ORBextractor.h:87).  Level images are returned WITHOUT
 * the 19 px border.  blurred != 0 selects the GaussianBlur'ed copy
 * (ORBextractor.cc:1226-1227). */
int asd_get_level_size(const asd_ctx* ctx, int32_t level, int32_t* width, int32_t* height);
int asd_get_level_image(asd_ctx* ctx, int32_t level, int32_t blurred, uint8_t* out);
/* Raw FAST corners of a level before the quadtree (ORBextractor.cc:858-876), in the
 * reference's vToDistributeKeys order; coordinates are relative to minBorder (16). */
int asd_get_raw_corners(asd_ctx* ctx, int32_t level, int32_t capacity, float* x, float* y,
                        float* response, int32_t* n_out);

/* ---- frames, grid (G1) and matchers (M0-M2, M4 init, M5) ----------------------------- */
/* A frame slot keeps what the reference keeps in Frame: undistorted keypoints
 * (mvKeysUn), descriptors (mDescriptors) and the 64x48 grid (Frame.cc:123-138), all
 * resident in HBM.  Slots are small integers in [0, ASD_MAX_FRAMES). */
#define ASD_MAX_FRAMES 64   /* (round 4: a keyframe and its 20 + 5 x 20 covisible neighbours resident for the batched per-keyframe stage) */
/* Replaces Frame::Frame(...) bookkeeping after ExtractORB: UndistortKeyPoints is the
 * identity for zero distortion (Frame.cc:298-304), image bounds (Frame.cc:351-357),
 * AssignFeaturesToGrid (Frame.cc:123-138).  desc may be NULL to adopt the descriptors of
 * the last asd_extract without a round trip. */
int asd_frame_set(asd_ctx* ctx, int32_t slot, const asd_keypoint* kps, const float* desc,
                  int32_t n, float min_x, float max_x, float min_y, float max_y);

/* Replaces Frame::GetFeaturesInArea (Frame.cc:219-274): indices in reference order. */
int asd_frame_features_in_area(asd_ctx* ctx, int32_t slot, float x, float y, float r,
                               int32_t min_level, int32_t max_level, int32_t capacity,
                               int32_t* idx_out, int32_t* n_out);

/* M0: ORBmatcher::DescriptorDistance (ORBmatcher.cc:1629-1650) for every pair, in the
 * reference's summation order (sequential f32, no FMA): out[na][nb].  Also the kernel of
 * MapPoint::ComputeDistinctiveDescriptors (MapPoint.cc:271-338). */
int asd_dist_matrix(asd_ctx* ctx, const float* a, int32_t na, const float* b, int32_t nb, float* out);

/* M5: MapPoint::ComputeDistinctiveDescriptors (MapPoint.cc:271-338) for one map point:
 * n observations' descriptors [n][128] -> index of the descriptor with the least median
 * distance to the others. */
int asd_distinctive_descriptor(asd_ctx* ctx, const float* desc, int32_t n, int32_t* best_idx);
/* The same for many map points in one call (LocalMapping::ProcessNewKeyFrame / SearchInNeighbors update hundreds
 * per keyframe, LocalMapping.cc:245, 535, 628): set s holds the observation descriptors
 * desc[set_start[s] .. set_start[s+1]) (row-major [.][128], set_start[0] = 0); best_idx[s] = index inside set s. */
int asd_distinctive_descriptor_batch(asd_ctx* ctx, int32_t n_sets, const int32_t* set_start, const float* desc,
                                     int32_t* best_idx);

/* M1: ORBmatcher::SearchByProjection(Frame& cur, const Frame& last, th, bMono=true)
 * (ORBmatcher.cc:1318-1452).  Inputs mirror what the method reads:
 *   last frame: slot_last (keypoint octave/angle), per keypoint i: has_mp[i] (mvpMapPoints[i]
 *   != NULL && !mvbOutlier[i]), world position Xw[i][3] (f32, MapPoint::GetWorldPos) and
 *   the map point's descriptor mp_desc[i][128] (MapPoint::GetDescriptor);
 *   cur frame: slot_cur, pose Tcw[16] f32 row-major 4x4, intrinsics K = fx,fy,cx,cy.
 * Output: match_cur[n_cur] = index i of the last-frame keypoint whose map point was
 * written into CurrentFrame.mvpMapPoints[j], or -1; *n_matches = the method's return
 * value.  cur.mvpMapPoints is taken as all-NULL on entry (Tracking.cc:670).
 * check_orientation mirrors mbCheckOrientation.
 * mp_obs_positive[i] (NULL = all non-zero): MapPoint::Observations() > 0 of last-frame map point i.  A current keypoint
 * that received a map point WITHOUT observations earlier in the same call is not skipped by later map points
 * (ORBmatcher.cc:1392-1395) and may be overwritten; each write counts in the return value and enters the rotation
 * histogram, exactly as in the reference (a keypoint written twice can be removed -- and subtracted -- twice). */
int asd_match_project_frame(asd_ctx* ctx, int32_t slot_cur, int32_t slot_last,
                            const uint8_t* has_mp, const float* Xw, const float* mp_desc,
                            const float* Tcw, const float* K, float th, int32_t check_orientation,
                            int32_t* match_cur, int32_t* n_matches, const uint8_t* mp_obs_positive);

/* M2: ORBmatcher::SearchByProjection(Frame& F, const vector<MapPoint*>&, th)
 * (ORBmatcher.cc:44-122) after Frame::isInFrustum (Frame.cc:160-217) has filled the
 * tracking fields.  Per map point m: in_view[m] (mbTrackInView && !isBad), proj[m] = (u,v),
 * level[m] = mnTrackScaleLevel, view_cos[m], desc[m][128].  occupied[j] != 0 marks current
 * keypoints that already hold a map point with Observations()>0 on entry.
 * Output: match_cur[n_cur] = map point index m written to F.mvpMapPoints[j] or -1 (entries
 * occupied on entry stay -1); *n_matches = reference return value
 * (counts each match twice, ORBmatcher.cc:116-117).
 * mp_obs_positive[m] (NULL = all non-zero): Observations() > 0 of map point m; a keypoint given a map point without
 * observations earlier in this call stays available to later map points (ORBmatcher.cc:86-88). */
int asd_match_project_points(asd_ctx* ctx, int32_t slot_cur, int32_t n_mp, const uint8_t* in_view,
                             const float* proj, const int32_t* level, const float* view_cos,
                             const float* desc, const uint8_t* occupied, float th, float nn_ratio,
                             int32_t* match_cur, int32_t* n_matches, const uint8_t* mp_obs_positive);

/* Device-resident descriptor bank.  The reference reads MapPoint::GetDescriptor() for every query of
 * every frame (ORBmatcher.cc:72, :1371) but only rewrites it once per keyframe
 * (MapPoint::ComputeDistinctiveDescriptors, MapPoint.cc:330-337).  An integration that gives each
 * MapPoint a bank row uploads a descriptor when it changes and passes row ids per frame instead of
 * 512-byte descriptors: asd_match_project_frame_bank / asd_match_project_points_bank are
 * asd_match_project_frame / _points with `rows` in place of the descriptor table. */
int asd_bank_put(asd_ctx* ctx, int32_t first_row, int32_t n, const float* desc);
/* bank[first_row + i] = descriptor of keypoint i of frame `slot`, i in [0, n): device to device */
int asd_bank_put_from_frame(asd_ctx* ctx, int32_t slot, int32_t first_row, int32_t n);
int asd_match_project_frame_bank(asd_ctx* ctx, int32_t slot_cur, int32_t slot_last,
                                 const uint8_t* has_mp, const float* Xw, const int32_t* mp_rows,
                                 const float* Tcw, const float* K, float th, int32_t check_orientation,
                                 int32_t* match_cur, int32_t* n_matches, const uint8_t* mp_obs_positive);
int asd_match_project_points_bank(asd_ctx* ctx, int32_t slot_cur, int32_t n_mp, const uint8_t* in_view,
                                  const float* proj, const int32_t* level, const float* view_cos,
                                  const int32_t* rows, const uint8_t* occupied, float th, float nn_ratio,
                                  int32_t* match_cur, int32_t* n_matches, const uint8_t* mp_obs_positive);

/* Fused tracking chains: the numeric bodies of Tracking::TrackWithMotionModel (Tracking.cc:664-723) and Tracking::TrackLocalMap
 * (:725-736 with the matcher call of SearchLocalPoints, :803-851) as ONE submission each -- search, claim replay, edge assembly
 * and Optimizer::PoseOptimization enqueued back to back on the device, one synchronisation at the end (the separate calls
 * cost two host round trips).  Results are bit-identical to calling the matcher and asd_pose_optimize one after the other.
 *
 * asd_track_motion_model = asd_match_project_frame (same arguments) followed by PoseOptimization over the matches: keypoint j
 * with match_cur[j] = i >= 0 contributes the edge (Xw[i], keypoint j, invSigma2 of its octave: Optimizer.cc:272-310), edges in
 * keypoint order.  pose7 in = SE3Quat of the predicted Tcw (Tracking.cc:672), out = the optimised pose; outlier[n_cur] =
 * mvbOutlier (0 where the keypoint has no match); *n_inliers = nInitialCorrespondences - nBad.  The caller keeps the
 * control flow: the nmatches < 20 retry with 2*th (:681-685) is a second call, the outlier discard (:695-714) a loop over
 * `outlier`.  Fewer than 3 matches: pose7 untouched, *n_inliers = 0 (Optimizer.cc:323-324). */
int asd_track_motion_model(asd_ctx* ctx, int32_t slot_cur, int32_t slot_last, const uint8_t* has_mp, const float* Xw,
                           const float* mp_desc, const float* Tcw, const float* K, float th, int32_t check_orientation,
                           const uint8_t* mp_obs_positive, double* pose7, int32_t* match_cur, int32_t* n_matches,
                           uint8_t* outlier, int32_t* n_inliers);
int asd_track_motion_model_bank(asd_ctx* ctx, int32_t slot_cur, int32_t slot_last, const uint8_t* has_mp, const float* Xw,
                                const int32_t* mp_rows, const float* Tcw, const float* K, float th, int32_t check_orientation,
                                const uint8_t* mp_obs_positive, double* pose7, int32_t* match_cur, int32_t* n_matches,
                                uint8_t* outlier, int32_t* n_inliers);
/* asd_track_local_map = asd_match_project_points (same arguments) followed by PoseOptimization over ALL map points of the
 * frame: keypoint j contributes an edge if occupied[j] != 0 (it held a map point on entry: world position cur_Xw[j][3]) or
 * match_cur[j] = m >= 0 (world position mp_Xw[m][3]).  mp_Xw[n_mp][3] / cur_Xw[n_cur][3] are f32 world positions
 * (MapPoint::GetWorldPos); the other outputs as above. */
int asd_track_local_map(asd_ctx* ctx, int32_t slot_cur, int32_t n_mp, const uint8_t* in_view, const float* proj,
                        const int32_t* level, const float* view_cos, const float* desc, const float* mp_Xw,
                        const uint8_t* occupied, const float* cur_Xw, float th, float nn_ratio,
                        const uint8_t* mp_obs_positive, const float* K, double* pose7, int32_t* match_cur,
                        int32_t* n_matches, uint8_t* outlier, int32_t* n_inliers);
int asd_track_local_map_bank(asd_ctx* ctx, int32_t slot_cur, int32_t n_mp, const uint8_t* in_view, const float* proj,
                             const int32_t* level, const float* view_cos, const int32_t* rows, const float* mp_Xw,
                             const uint8_t* occupied, const float* cur_Xw, float th, float nn_ratio,
                             const uint8_t* mp_obs_positive, const float* K, double* pose7, int32_t* match_cur,
                             int32_t* n_matches, uint8_t* outlier, int32_t* n_inliers);

/* The same with Tracking::SearchLocalPoints' frustum loop (Tracking.cc:817-835) on the device as well: the map points are given
 * as for asd_frustum (position, normal, raw mfMinDistance / mfMaxDistance) together with the frame pose Tcw they are projected
 * with; isInFrustum, PredictScale (the level by comparison with thresholds derived from the host's logf: same value as libm's)
 * and the search windows are computed on the device, then asd_track_local_map's chain runs.  Xw doubles as the position
 * table of the new matches' edges. */
int asd_track_local_points(asd_ctx* ctx, int32_t slot_cur, int32_t n_mp, const float* Xw, const float* normal,
                           const float* min_dist, const float* max_dist, const float* desc, const float* Tcw, const float* K,
                           float viewing_cos_limit, const uint8_t* occupied, const float* cur_Xw, float th, float nn_ratio,
                           const uint8_t* mp_obs_positive, double* pose7, int32_t* match_cur, int32_t* n_matches,
                           uint8_t* outlier, int32_t* n_inliers);
int asd_track_local_points_bank(asd_ctx* ctx, int32_t slot_cur, int32_t n_mp, const float* Xw, const float* normal,
                                const float* min_dist, const float* max_dist, const int32_t* rows, const float* Tcw,
                                const float* K, float viewing_cos_limit, const uint8_t* occupied, const float* cur_Xw, float th,
                                float nn_ratio, const uint8_t* mp_obs_positive, double* pose7, int32_t* match_cur,
                                int32_t* n_matches, uint8_t* outlier, int32_t* n_inliers);

/* asd_track_local_points_bank with the map points named by ROW of both banks (descriptor bank + the attribute bank below: position, normal,
 * distance range as asd_mpbank_put stored them): Tracking::SearchLocalPoints' loop (Tracking.cc:817-835) walks mvpLocalMapPoints, whose
 * attributes change once per keyframe (UpdateNormalAndDepth) -- per frame the host then names the points it selected (4 bytes each) instead
 * of gathering and uploading 36 bytes per point.  Results as asd_track_local_points_bank on the same attributes, bit for bit.  n_mp >= 1;
 * sizes beyond the device replay return ASD_ERR_CAPACITY (use the _bank form).  Honours asd_track_async / asd_track_finish. */
int asd_track_local_points_rows(asd_ctx* ctx, int32_t slot_cur, int32_t n_mp, const int32_t* rows, const float* Tcw, const float* K,
                                float viewing_cos_limit, const uint8_t* occupied, const float* cur_Xw, float th, float nn_ratio,
                                const uint8_t* mp_obs_positive, double* pose7, int32_t* match_cur, int32_t* n_matches, uint8_t* outlier,
                                int32_t* n_inliers);

/* Map-point attribute bank, the companion of the descriptor bank (same row ids): MapPoint::mWorldPos, mNormalVector,
 * mfMinDistance, mfMaxDistance (MapPoint.cc:60-77, 340-391; rewritten by UpdateNormalAndDepth once per keyframe, read by
 * Frame::isInFrustum for every local map point of every frame, Frame.cc:160-217).  Xw[n][3], normal[n][3], min_dist[n],
 * max_dist[n] (raw values: isInFrustum's 0.8 / 1.2 factors are applied by the kernels).  The arrays are consumed on return. */
int asd_mpbank_put(asd_ctx* ctx, int32_t first_row, int32_t n, const float* Xw, const float* normal, const float* min_dist,
                   const float* max_dist);

/* Both tracking stages of a frame as ONE submission -- an OPTIONAL form with a precondition the reference's Tracking does not
 * meet in general; the form that binds to Tracking.cc as it stands is the two-call one (asd_track_motion_model_bank -> host ->
 * asd_track_local_points_bank), and that is what bench.py's `value` is measured on.
 * What it does: asd_track_motion_model_bank, then -- on the device, where the host of the two-call form sits -- what Tracking
 * does between the stages: matches PoseOptimization marked as outliers are dropped (Tracking.cc:695-714), the optimised pose
 * becomes the frame's pose (Frame::SetPose, Frame.cc:150-158), the map points the motion-model stage matched -- kept, or
 * dropped as outliers (both carry mnLastFrameSeen = this frame: :705-707, :811-823) -- are left out of the local-map search;
 * then asd_track_local_points_bank over the remaining candidates, from the motion-model stage's pose and kept matches.
 * The precondition: the reference selects mvpLocalMapPoints BETWEEN the stages (Tracking::UpdateLocalMap, Tracking.cc:730 ->
 * :871-879: UpdateLocalKeyFrames :907-1000 votes over the matches the motion-model stage has just made, UpdateLocalPoints
 * :881-905 collects those keyframes' points).  This call has no place for that decision: the local map is passed as a list
 * known BEFORE the frame is tracked -- n_cand rows of the attribute + descriptor bank (cand_rows), and per keypoint of the last
 * frame the candidate index of the map point it holds (last_cand[i], -1 = not among the candidates; NULL = none is).  Results
 * equal the reference's only when that list IS the set UpdateLocalMap would select: localisation against a fixed map whose
 * local set the caller fixes per frame, or a caller that deliberately tracks against the previous frame's local map (one
 * frame of lag in the keyframe vote).  Any other superset lets the local-map stage match points the reference would not have
 * searched.  Given the same candidate list the results are those of the two calls bit for bit (match2[j] is a CANDIDATE index).
 * pose7: in = SE3Quat of the predicted pose, out = the local-map stage's optimum; pose1 = the motion-model stage's.  The
 * control flow stays with the caller as before: if the motion-model stage made too few matches (nmatches < 20,
 * Tracking.cc:681-685) it discards match2 / pose7 and runs the retry with the separate calls.  Sizes beyond the one-submission
 * form (more than 4096 last-frame keypoints, more ACTIVE local-map lists than one replay workgroup takes, frames beyond the
 * solver's LDS form) return ASD_ERR_CAPACITY: use the two calls.  Honours asd_track_async / asd_track_finish. */
typedef struct asd_track_frame_args {
  int32_t slot_cur, slot_last;
  const uint8_t* has_mp;            /* [n_last] motion-model stage: as asd_track_motion_model_bank */
  const float* Xw_last;             /* [n_last][3] */
  const int32_t* last_rows;         /* [n_last] descriptor-bank rows */
  const int32_t* last_cand;         /* [n_last] or NULL */
  const uint8_t* last_obs_positive; /* [n_last] or NULL */
  const float* Tcw;                 /* [16] predicted pose the motion-model search projects with */
  float th;                         /* its window factor (15 mono, Tracking.cc:676-680) */
  int32_t check_orientation;
  int32_t n_cand;                   /* local-map stage: as asd_track_local_points_bank, attributes from the bank */
  const int32_t* cand_rows;         /* [n_cand] rows of the attribute and descriptor banks */
  const uint8_t* cand_obs_positive; /* [n_cand] or NULL */
  float viewing_cos_limit, th_local, nn_ratio;
  const float* K;                   /* fx fy cx cy */
  double* pose7;                    /* [7] in / out */
  double* pose1;                    /* [7] out */
  int32_t *match1, *n_matches1; uint8_t* outlier1; int32_t* n_inliers1;   /* [n_cur] tables, as the stage calls return them */
  int32_t *match2, *n_matches2; uint8_t* outlier2; int32_t* n_inliers2;
} asd_track_frame_args;
int asd_track_frame(asd_ctx* ctx, const asd_track_frame_args* args);

/* Frame construction beside the stages in flight.  Between asd_prep_async(ctx, 1) and asd_prep_async(ctx, 0) the device
 * work of asd_frame_set, asd_bank_put_from_frame and asd_mpbank_put is enqueued on a second stream of the context instead of
 * behind the outstanding asd_track_* stage (the reference's Frame constructor runs before the frame is tracked, not after the
 * previous one: Tracking.cc:96-118); closing the bracket orders everything enqueued on the context AFTERWARDS behind it.  The
 * caller guarantees what ordering on one stream gave for free: nothing written inside the bracket (the frame slot, the bank
 * rows) is read by a stage still in flight. */
int asd_prep_async(asd_ctx* ctx, int32_t on);

/* Split-phase execution of the asd_track_* calls.  asd_track_async(ctx) arms the NEXT asd_track_motion_model[_bank] /
 * asd_track_local_map[_bank] / asd_track_local_points[_bank] call on this context: it returns as soon as its work is enqueued
 * -- every INPUT array has been consumed by then and may be reused, the OUTPUT arrays (pose7, match_cur, n_matches,
 * outlier, n_inliers) must stay valid -- and asd_track_finish(ctx) waits for the device, fills the outputs and returns the
 * status the synchronous call would have returned (same results bit for bit: same kernels, same order).  The host work of
 * the next frame that does not depend on this result -- Frame construction in the reference: asd_extract_wait[_view],
 * asd_frame_set on ANOTHER slot, asd_extract_submit, asd_bank_put*, and asd_local_ba_submit / _wait / _poll -- may run in
 * between (it is enqueued behind the chain on the context's stream); any other call that uses the matcher or the pose solver
 * returns ASD_ERR_INVALID until asd_track_finish.  An armed call that fails returns its error and leaves nothing outstanding;
 * asd_track_finish with nothing outstanding is ASD_ERR_INVALID. */
int asd_track_async(asd_ctx* ctx);
int asd_track_finish(asd_ctx* ctx);

/* Frame::isInFrustum (Frame.cc:160-217) + MapPoint::PredictScale (MapPoint.cc:438-453) for
 * n map points: Xw[n][3], normal[n][3] (GetNormal), min_dist[n] / max_dist[n] = the map
 * point's raw mfMinDistance / mfMaxDistance (the 0.8 / 1.2 invariance factors of
 * MapPoint.cc:409-419 are applied inside).  Outputs as consumed by M2.  Runs on the host:
 * a few thousand points x ~50 flops, and PredictScale's logf must match libm bit for bit. */
int asd_frustum(asd_ctx* ctx, int32_t slot_cur, int32_t n, const float* Xw, const float* normal,
                const float* min_dist, const float* max_dist, const float* Tcw, const float* K,
                float viewing_cos_limit, uint8_t* in_view, float* proj, int32_t* level, float* view_cos);

/* M4 (bootstrap only): ORBmatcher::SearchForInitialization (ORBmatcher.cc:416-531).
 * prev_matched[n1][2] in/out (vbPrevMatched), matches12[n1] out (index in frame 2 or -1). */
int asd_match_init(asd_ctx* ctx, int32_t slot1, int32_t slot2, float* prev_matched, int32_t window,
                   float nn_ratio, int32_t check_orientation, int32_t* matches12, int32_t* n_matches);

/* DBoW2::FeatureVector of a frame (node id at `levelsup` = 4 -> keypoint indices; Frame::ComputeBoW,
 * Frame.cc:289-296), as produced by asd_compute_bow below or by the caller's own DBoW2.
 * node_id ascending (std::map order), start has n_nodes+1 entries into idx. */
typedef struct asd_feature_vector {
  int32_t n_nodes;
  const int32_t* node_id;
  const int32_t* start;
  const int32_t* idx;
} asd_feature_vector;

/* ---- relocalisation / loop-closing variants of the projection searches ----
 * Flat-array forms of the remaining ORBmatcher overloads.  min_dist / max_dist are the raw mfMinDistance /
 * mfMaxDistance (the 0.8 / 1.2 factors of Get{Min,Max}DistanceInvariance are applied inside), K = fx fy cx cy,
 * poses row-major 4x4 f32.  Pointer-graph side effects stay with the caller. */

/* ORBmatcher::SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const set<MapPoint*>& sAlreadyFound, float th,
 * float ORBdist) (ORBmatcher.cc:1455-1582, Tracking::Relocalization).  Arrays over pKF's keypoints (n_kf):
 * valid[i] = pMP && !isBad() && !sAlreadyFound.count(pMP); kf_angle[i] = pKF->mvKeysUn[i].angle;
 * occupied[j] = CurrentFrame.mvpMapPoints[j] != NULL on entry.  match_cur[j] = i (pKF's map point assigned to the
 * current frame's keypoint j) or -1; *n_matches = return value. */
int asd_match_project_keyframe(asd_ctx* ctx, int32_t slot_cur, int32_t n_kf, const uint8_t* valid, const float* Xw,
                               const float* min_dist, const float* max_dist, const float* desc, const float* kf_angle,
                               const uint8_t* occupied, const float* Tcw, const float* K, float th, float orb_dist,
                               int32_t check_orientation, int32_t* match_cur, int32_t* n_matches);

/* ORBmatcher::SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const vector<MapPoint*>& vpPoints,
 * vector<MapPoint*>& vpMatched, int th) (:300-413, LoopClosing).  valid[i] = !isBad() && !spAlreadyFound.count(pMP).
 * matched_kp[j] in/out over pKF's keypoints: -1 = vpMatched[j] is NULL, anything else = occupied on entry; the call
 * writes the index i of the point it assigns.  *n_matches = return value. */
int asd_match_project_sim3(asd_ctx* ctx, int32_t slot_kf, const float* Scw, int32_t n_mp, const uint8_t* valid,
                           const float* Xw, const float* normal, const float* min_dist, const float* max_dist,
                           const float* desc, const float* K, int32_t th, int32_t* matched_kp, int32_t* n_matches);

/* ORBmatcher::Fuse(KeyFrame* pKF, cv::Mat Scw, const vector<MapPoint*>& vpPoints, float th,
 * vector<MapPoint*>& vpReplacePoint) (:963-1086), search half like asd_fuse_search (no chi2 gate in this overload). */
int asd_fuse_search_sim3(asd_ctx* ctx, int32_t slot_kf, const float* Scw, int32_t n_mp, const uint8_t* valid,
                         const float* Xw, const float* normal, const float* min_dist, const float* max_dist,
                         const float* desc, const float* K, float th, int32_t* best_idx, float* best_dist);

/* ORBmatcher::SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th) (:1090-1314).  Arrays over each keyframe's
 * keypoints: has[i] = map point present, not bad and not already matched (vbAlreadyMatched1/2); T1w / T2w the
 * keyframe poses; R12 row-major 3x3, t12[3].  match12[i1] = i2 for mutually consistent pairs, else -1. */
int asd_match_sim3(asd_ctx* ctx, int32_t slot1, int32_t slot2, const uint8_t* has1, const uint8_t* has2, const float* Xw1,
                   const float* Xw2, const float* min_dist1, const float* max_dist1, const float* min_dist2,
                   const float* max_dist2, const float* desc1, const float* desc2, const float* T1w, const float* T2w,
                   float s12, const float* R12, const float* t12, const float* K, float th, int32_t* match12,
                   int32_t* n_matches);

/* ---- vocabulary ----
 * ORBVocabulary = TemplatedVocabulary<FSift::TDescriptor, FSift> (ORBVocabulary.h:34).  The tree is handed
 * over as flat arrays indexed by DBoW2 node id (0 = root, as in m_nodes): children of node i are
 * child_ids[child_start[i] .. child_start[i+1]) in m_nodes[i].children order (= file order,
 * TemplatedVocabulary.h:1455-1497), weight[i] = m_nodes[i].weight, word_id[i] = m_nodes[i].word_id for
 * leaves and -1 for inner nodes, desc[i] = m_nodes[i].descriptor (128 f32; row 0 unused).
 * weighting: 0 TF_IDF, 1 TF, 2 IDF, 3 BINARY; scoring: 0 L1, 1 L2, 2 CHI_SQUARE, 3 KL, 4 BHATTACHARYYA,
 * 5 DOT_PRODUCT (BowVector.h enums).  ASD_ERR_INVALID unless the arrays describe a tree rooted at 0 whose
 * leaves all carry a word id; ASD_ERR_CAPACITY for a branching factor above 64. */
int asd_voc_load(asd_ctx* ctx, int32_t n_nodes, int32_t k, int32_t L, int32_t weighting, int32_t scoring,
                 const int32_t* child_start, const int32_t* child_ids, const double* weight,
                 const int32_t* word_id, const float* desc);
/* TemplatedVocabulary::transform(feature, word_id, weight, nid, levelsup) (TemplatedVocabulary.h:1219-1260)
 * for n descriptors: desc != NULL -> host descriptors [n][128]; desc == NULL -> the descriptors resident in
 * frame slot `slot` (n must equal the slot's keypoint count).  Any of word / node / weight may be NULL.
 * A leaf met above level L - levelsup reports itself as node (the reference leaves *nid unset there). */
int asd_bow_descend(asd_ctx* ctx, int32_t slot, const float* desc, int32_t n, int32_t levelsup, int32_t* word,
                    int32_t* node, double* weight);
/* Frame::ComputeBoW (Frame.cc:289-296) = transform(features, BowVector, FeatureVector, levelsup)
 * (TemplatedVocabulary.h:1125-1197).  Outputs (caller-owned, capacity n, fv_start n+1): BowVector as
 * (bow_id ascending, bow_val) with *n_words entries, FeatureVector as CSR over *n_fv_nodes nodes in the
 * asd_feature_vector layout. */
int asd_compute_bow(asd_ctx* ctx, int32_t slot, const float* desc, int32_t n, int32_t levelsup, int32_t* bow_id,
                    double* bow_val, int32_t* n_words, int32_t* fv_node, int32_t* fv_start, int32_t* fv_idx,
                    int32_t* n_fv_nodes);

/* M3: ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame& F, vector<MapPoint*>& vpMapPointMatches)
 * (ORBmatcher.cc:156-297): slot_kf / slot_f hold the two frames, has_mp_kf[i] = pKF has a good map
 * point at keypoint i.  Output match_f[n_f] = keypoint index in pKF whose map point was assigned
 * to F's keypoint, or -1; *n_matches = return value. */
int asd_match_bow(asd_ctx* ctx, int32_t slot_kf, int32_t slot_f, const asd_feature_vector* fv_kf,
                  const asd_feature_vector* fv_f, const uint8_t* has_mp_kf, float nn_ratio,
                  int32_t check_orientation, int32_t* match_f, int32_t* n_matches);

/* M4: ORBmatcher::SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo=false)
 * (ORBmatcher.cc:669-822): has_mp1 / has_mp2 mark keypoints that already hold a map point,
 * F12[9] row-major f32 fundamental matrix, epipole (ex, ey) of camera 1 in image 2 as computed at
 * :675-683, sigma2 level table of pKF2 comes from the ctx.  Output matches12[n1] = index in
 * KF2 or -1; *n_matches = return value. */
int asd_match_triangulate(asd_ctx* ctx, int32_t slot1, int32_t slot2, const asd_feature_vector* fv1,
                          const asd_feature_vector* fv2, const uint8_t* has_mp1, const uint8_t* has_mp2,
                          const float* F12, float ex, float ey, int32_t check_orientation,
                          int32_t* matches12, int32_t* n_matches);

/* ORBmatcher::Fuse(KeyFrame* pKF, const vector<MapPoint*>& vpMapPoints, float th) (ORBmatcher.cc:825-960),
 * search half: for every candidate map point (valid[i] = pMP && !isBad() && !IsInKeyFrame(pKF)) project into
 * the keyframe, apply the depth / viewing-angle / scale gates, search the window and return the keypoint
 * with the smallest descriptor distance among those passing the level and 5.99-chi2 gates.  best_idx[i] = -1
 * when nothing qualifies (bestDist > TH_LOW).  The caller performs the pointer-graph side effects
 * (Replace / AddObservation / AddMapPoint, :938-956) in map point order; they do not influence the search.
 * min_dist / max_dist are the raw mfMinDistance / mfMaxDistance like asd_frustum. */
int asd_fuse_search(asd_ctx* ctx, int32_t slot_kf, int32_t n_mp, const uint8_t* valid, const float* Xw,
                    const float* normal, const float* min_dist, const float* max_dist, const float* desc,
                    const float* Tcw, const float* K, float th, int32_t* best_idx, float* best_dist);

/* LocalMapping::CreateNewMapPoints, per-match body (LocalMapping.cc:386-519, monocular branch): for each
 * match (idx1[i] in frame slot1 = current keyframe, idx2[i] in slot2 = neighbour) run the parallax gate
 * (0 < cos < 0.9998), the linear triangulation (4x4 SVD, :437-453), cheirality in both cameras, the 5.991
 * reprojection gates and the scale-consistency gate (ratioFactor = 1.5*scaleFactor).  ok[i] = 1 and x3D[i]
 * = the new point when the reference would create a MapPoint, else ok[i] = 0 and x3D[i] = 0.  The caller
 * keeps the pointer-graph part (new MapPoint, AddObservation, ComputeDistinctiveDescriptors ->
 * asd_distinctive_descriptor, UpdateNormalAndDepth, :497-515).  Tcw row-major 4x4 f32, K = fx fy cx cy. */
int asd_triangulate_pairs(asd_ctx* ctx, int32_t slot1, int32_t slot2, int32_t n_pairs, const int32_t* idx1,
                          const int32_t* idx2, const float* Tcw1, const float* Tcw2, const float* K1,
                          const float* K2, float* x3D, uint8_t* ok, int32_t* n_ok);
/* ---- the per-keyframe work in front of LocalBA, batched (LocalMapping::DoMapping, LocalMapping.cc:59-113) ----
 * CreateNewMapPoints (:299-545) runs SearchForTriangulation + the per-match triangulation against nn = 20 covisible keyframes,
 * SearchInNeighbors (:557-636) runs Fuse against 20 + second neighbours both ways: 40-100 matcher calls per keyframe, each a
 * few hundred microseconds as a call of its own.  The batch entry points take all of them in ONE submission: one upload, one
 * launch chain, one synchronisation.  What makes that legitimate: inside one call the searches do not feed each other --
 * vbMatched2 is never set in SearchForTriangulation (ORBmatcher.cc:688, :729), Fuse's Replace / AddObservation side effects
 * never enter the search (:938-956) -- and the only coupling BETWEEN the reference's consecutive calls is a filter the caller
 * applies when it walks the results in the reference's order: a keypoint of the current keyframe that an earlier neighbour
 * already gave a map point is skipped for later neighbours (pKF1->GetMapPoint(idx1), ORBmatcher.cc:703-707 after
 * LocalMapping.cc:497-503), a map point an earlier Fuse replaced is bad for later ones (:840-846).  Every result equals the
 * per-pair call's (asd_match_triangulate + asd_triangulate_pairs, asd_fuse_search) bit for bit. */

/* The keyframe's DBoW2::FeatureVector resident beside its descriptors (slot must hold the frame): node ids ascending. */
int asd_frame_set_bow(asd_ctx* ctx, int32_t slot, const asd_feature_vector* fv);

typedef struct asd_kf_neighbor {
  int32_t slot;              /* frame slot of the neighbour keyframe (asd_frame_set + asd_frame_set_bow done) */
  const uint8_t* has_mp;     /* [n_keypoints of that slot] it already holds a map point there */
  float F12[9];              /* fundamental matrix, current keyframe -> neighbour (LocalMapping.cc:547-555), row-major f32 */
  float ex, ey;              /* epipole of the current camera in the neighbour's image (ORBmatcher.cc:675-683) */
  float Tcw[16];             /* the neighbour's pose */
  float K[4];                /* its intrinsics fx fy cx cy */
} asd_kf_neighbor;
/* SearchForTriangulation (check_orientation = false, as LocalMapping.cc:370 calls it) + the per-match triangulation body for
 * every neighbour: matches12[b][i] = keypoint of neighbour b matched to keypoint i of the current keyframe or -1 (has_mp_cur
 * as at entry: see above), x3D[b][i][3] / ok[b][i] as asd_triangulate_pairs returns them for that pair (0 where no match).
 * Arrays are [n_nb][n_cur] row-major. */
int asd_create_map_points_batch(asd_ctx* ctx, int32_t slot_cur, const uint8_t* has_mp_cur, const float* Tcw_cur, const float* K_cur,
                                int32_t n_nb, const asd_kf_neighbor* nb, int32_t* matches12, int32_t* n_matches, float* x3D, uint8_t* ok);

typedef struct asd_fuse_call {
  int32_t slot_kf;           /* target keyframe */
  int32_t first, n;          /* its candidate map points: rows [first, first + n) of the tables below */
  float Tcw[16];
  float K[4];
} asd_fuse_call;
/* asd_fuse_search for n_calls (keyframe, map point list) pairs over shared tables (valid, Xw, normal, min_dist, max_dist,
 * and the map points' descriptors: desc [n_total][128], or desc == NULL and desc_rows [n_total] = rows of the descriptor bank --
 * 4 bytes per point uploaded instead of 512); best_idx / best_dist [n_total] as asd_fuse_search returns them per call. */
int asd_fuse_search_batch(asd_ctx* ctx, int32_t n_calls, const asd_fuse_call* calls, int32_t n_total, const uint8_t* valid, const float* Xw,
                          const float* normal, const float* min_dist, const float* max_dist, const float* desc, const int32_t* desc_rows,
                          float th, int32_t* best_idx, float* best_dist);

/* vt.row(3) of cv::SVD::compute(A, w, u, vt, MODIFY_A|FULL_UV) for n row-major 4x4 f32 matrices
 * (the call at LocalMapping.cc:444); v is [n][4].  Exposed for the parity tests of the SVD itself. */
int asd_svd4_null(asd_ctx* ctx, int32_t n, const float* A, float* v);

/* ---- stereo association (dead code in the reference: no stereo Frame constructor survives) ----
 * Frame::ComputeStereoMatches (Frame.cc:360-535).  The reference keeps two extractors (mpORBextractorLeft / Right,
 * Frame.h), hence two contexts here: extract the left image on ctx_left and the right image on ctx_right (same size,
 * same device), put mvKeys + mDescriptors and mvKeysRight + mDescriptorsRight into two frame slots of ctx_left, then
 * call this before the next extraction overwrites either pyramid.  u_right[N] / depth[N] = mvuRight / mvDepth (-1 =
 * no stereo match); *n_matched = matches that survive the 1.5 * 1.4 * median SAD filter (:517-531).
 * mb = baseline in metres, mbf = baseline * fx. */
/* asd_frame_set(desc == NULL) with the descriptors of src's last extraction instead of ctx's own (both contexts on one device): the
 * right frame of a stereo pair, extracted by its own context, put into a slot of the left one for asd_stereo_match. */
int asd_frame_set_from_ctx(asd_ctx* ctx, int32_t slot, const asd_keypoint* kps, int32_t n, float min_x, float max_x, float min_y,
                           float max_y, asd_ctx* src);
/* With the pipelined extractor (asd_extract_submit / asd_extract_wait*) the shared pyramid of a context holds a LATER frame by the
 * time both extractions of a stereo pair have been waited for: asd_extract_keep_pyramid(ctx, 1) makes every submission keep its own
 * copy of its pyramid (one device-to-device copy in the front half), and asd_stereo_match then reads the copies of the submissions
 * waited for last on the two contexts -- valid as long as their descriptors are (two further submissions).  Inside an
 * asd_prep_async bracket of ctx_left the call runs on that context's second stream, beside tracking stages in flight. */
int asd_extract_keep_pyramid(asd_ctx* ctx, int32_t on);
/* Read-ahead extraction on hold: while on != 0 the extractor's workers enqueue no further ASDNet forward (front halves continue, forwards
 * already enqueued finish).  For a caller that runs LocalMapping::DoMapping in line (Tracking.cc:797 -> LocalMapping.cc:59-113): the extractor
 * is told to stand back for the length of asd_local_ba and makes the time up beside the tracking stages.  asd_extract_wait[_view] on a
 * submission that is held ends the hold (it could not return otherwise). */
int asd_extract_hold(asd_ctx* ctx, int32_t on);
int asd_stereo_match(asd_ctx* ctx_left, asd_ctx* ctx_right, int32_t slot_left, int32_t slot_right, float mb, float mbf,
                     float* u_right, float* depth, int32_t* n_matched);

/* ---- optimizer (P1, B1-B5, C1) ------------------------------------------------------- */
/* Optimizer::PoseOptimization (Optimizer.cc:239-413) on g2o's EdgeSE3ProjectXYZOnlyPose
 * (types_six_dof_expmap.h:194-222, .cpp:372-394) with Levenberg
 * (optimization_algorithm_levenberg.cpp:61-189).
 *   pose7: in/out (qx,qy,qz,qw,tx,ty,tz) f64, world->camera, as Converter::toSE3Quat
 *   produces from the f32 Tcw (Converter.cc:37-47);
 *   Xw[n][3], obs[n][2], inv_sigma2[n] as f64 (values are f32-representable);
 *   outlier[n] out (mvbOutlier); *n_inliers = nInitialCorrespondences - nBad. */
int asd_pose_optimize(asd_ctx* ctx, double* pose7, int32_t n, const double* Xw, const double* obs,
                      const double* inv_sigma2, const double* K, uint8_t* outlier, int32_t* n_inliers);

typedef struct asd_ba_problem {
  int32_t n_poses;          /* local + fixed keyframes, ordered by ascending KF id        */
  int32_t n_points;         /* local map points, ordered by ascending MP id               */
  int32_t n_edges;          /* in the reference's insertion order (Optimizer.cc:534-593)  */
  double*        poses;     /* [n_poses][7] in/out (qx,qy,qz,qw,tx,ty,tz)                 */
  const uint8_t* fixed;     /* [n_poses] vSE3->setFixed (Optimizer.cc:490,505)            */
  double*        points;    /* [n_points][3] in/out                                       */
  const int32_t* e_point;   /* [n_edges] point index                                      */
  const int32_t* e_pose;    /* [n_edges] pose index                                       */
  const double*  e_obs;     /* [n_edges][2] kpUn.pt                                       */
  const double*  e_info;    /* [n_edges] information scale (invSigma2, x10 global map)    */
  double K[4];              /* fx, fy, cx, cy                                             */
  int32_t its_first;        /* 5  (Optimizer.cc:602)                                      */
  int32_t its_second;       /* 10 (Optimizer.cc:648)                                      */
} asd_ba_problem;

typedef struct asd_ba_result {
  double*  edge_chi2;        /* [n_edges] final e->chi2()                                 */
  uint8_t* edge_depth_pos;   /* [n_edges] final e->isDepthPositive()                      */
  uint8_t* edge_outlier1;    /* [n_edges] moved to level 1 after the first round          */
  double   chi2_first;       /* active robust chi2 after optimize(its_first)              */
  double   chi2_second;      /* active chi2 after optimize(its_second)                    */
  int32_t  iters_first;      /* LM iterations actually run                                */
  int32_t  iters_second;
} asd_ba_result;

/* Numeric core of Optimizer::LocalBundleAdjustment (Optimizer.cc:484-650): BlockSolver_6_3
 * (block_solver.hpp:354-604) + LinearSolverDense + Levenberg + Huber(sqrt 5.991), two
 * rounds with the chi2 > 5.991 || depth <= 0 gating in between.  The caller applies the
 * erase policy (Optimizer.cc:652-700) from edge_chi2 / edge_depth_pos. */
int asd_local_ba(asd_ctx* ctx, asd_ba_problem* problem, asd_ba_result* result);
/* The same computation on an OPTIONAL lane of the library -- not the reference's order.  In this reference LocalBundleAdjustment
 * runs in line: Tracking::CreateNewKeyFrame -> LocalMapping::DoMapping (Tracking.cc:797, LocalMapping.cc:59-113, the call at
 * :89); there is no mapping thread (LocalMapping::Run, :120, is never started), so frame t+1 is tracked against the map that
 * the keyframe's LocalBA has already rewritten: that is asd_local_ba above.  These entry points offer upstream ORB-SLAM2's
 * concurrent arrangement to an integrator who wants it: _submit hands the problem to a worker thread that runs it on a HIP
 * stream of its own and returns at once; the tracking entry points (asd_extract*, asd_frame_set, asd_match_*, asd_track_*,
 * asd_pose_optimize) may be called meanwhile -- the frames tracked meanwhile read the PRE-BA map, so poses depart from the
 * reference's.  *problem, *result and every array they point to belong to the library until _wait
 * returns; _wait blocks until the run has finished and returns ITS status (results bit-identical to asd_local_ba:
 * same kernels, same order).  One run at a time: a second _submit before _wait, asd_local_ba while a run is outstanding
 * and _wait with nothing submitted return ASD_ERR_INVALID.  asd_local_ba_poll: 0 = idle, 1 = running, 2 = finished and
 * waiting to be collected.  Aborting a run (mbAbortBA) is not offered: a run is ~4 ms. */
int asd_local_ba_submit(asd_ctx* ctx, asd_ba_problem* problem, asd_ba_result* result);
int asd_local_ba_wait(asd_ctx* ctx);
int asd_local_ba_poll(asd_ctx* ctx);

/* Converter::toSE3Quat / toCvMat (Converter.cc:37-71): f32 4x4 <-> f64 quaternion + t. */
int asd_tcw_to_pose7(const float* Tcw16, double* pose7);
int asd_pose7_to_tcw(const double* pose7, float* Tcw16);

/* ---- instrumentation ----------------------------------------------------------------- */
/* Device-side duration of the kernels enqueued by the most recent call of the named stage,
 * measured with hipEvents on the ctx stream.  stage: "asdnet", "extract", "match", "ba". */
int asd_last_stage_ms(const asd_ctx* ctx, const char* stage, float* ms);
/* Per-kernel device time of the ASDNet forward, accumulated with hipEvents recorded on the ctx
 * stream between the layer launches of every forward while enabled.  layer 0 = input_norm+conv1,
 * 1..5 = conv2..conv6 (MFMA implicit GEMM), 6 = 8x8 conv (split-K GEMM), 7 = reduce+L2Norm. */
int asd_profile_enable(asd_ctx* ctx, int32_t on);
int asd_profile_get(asd_ctx* ctx, int32_t layer, double* total_ms, int32_t* calls, int64_t* patches);
/* Which ASDNet layers run on the split-operand kernels: bit 0 = conv2 ... bit 4 = conv6, bit 5 = the 8x8 conv.
 * Both kernel families compute in f32: the split kernels write every f32 operand exactly as the sum of three bf16 terms
 * and accumulate the six significant cross products in f32 on the bf16 matrix pipe (error at the level of the f32 MFMA chain,
 * see asdnet.hip); the others use v_mfma_f32_32x32x2_f32.  Chosen at asd_ctx_create: environment ASD_ASDNET_MATH=f32 clears
 * every bit, ASD_ASDNET_SPLIT_LAYERS=<mask> sets them individually; default all conv layers split. */
int32_t asd_asdnet_split_mask(const asd_ctx* ctx);
/* How the split-operand 3x3 conv kernels carry an f32 operand (chosen at asd_ctx_create by ASD_ASDNET_MATH):
 *   2  ("f16x2", the default; "split" is an alias)  x 2^k = h + l with two fp16 terms (22 significant bits), the three products
 *      l h, h l, h h accumulated in f32 on the f16 matrix pipe; against a float64 forward the descriptors are as close as those
 *      of the f32 MFMA chain (tests/test_asdnet.py).  Activations must stay below 4094 in magnitude (BatchNorm keeps them O(1)).
 *      Two guards: asd_load_weights runs a calibration batch (64 synthetic patches: noise, edges, checkerboards, ramps, dots)
 *      and, if any layer's largest activation leaves less than a factor two of headroom (> 2048), switches this context to
 *      the three-piece form below (asd_asdnet_pieces then reports 3, asd_last_error says why); and a descriptor that comes out
 *      non-finite at run time raises a device flag: asd_describe / asd_extract* / asd_extract_wait* return ASD_ERR_RANGE for
 *      that call (asd_describe_device, which does not synchronise, reports it from the next asd_sync) -- the reference's f32
 *      libtorch path (ORBextractor.cc:1127-1132) has no such failure mode, so it is an error here, never a silent NaN.
 *   3  ("bf16x3")  exact sum of three bf16 terms, six products; no range restriction; 1.35x the ASDNet time of the default. */
int32_t asd_asdnet_pieces(const asd_ctx* ctx);
/* what the calibration pass of the last asd_load_weights found ("" = nothing to report; non-empty when it switched the context to
 * the three-piece form).  A note on a successful call: asd_last_error is not touched by it. */
const char* asd_calibration_note(const asd_ctx* ctx);
/* Raw handles for harnesses that keep inputs resident (bench.py): the ctx stream
 * (hipStream_t) and device scratch. */
void* asd_ctx_stream(asd_ctx* ctx);
int asd_device_alloc(asd_ctx* ctx, uint64_t bytes, void** dptr);
int asd_device_free(asd_ctx* ctx, void* dptr);
/* page-locked host memory (an image buffer handed to asd_extract_submit(device_resident = 0) from here is copied by the front
 * half's stream without a staging pass) */
int asd_host_alloc(asd_ctx* ctx, uint64_t bytes, void** hptr);
int asd_host_free(asd_ctx* ctx, void* hptr);
int asd_memcpy_h2d(asd_ctx* ctx, void* dst, const void* src, uint64_t bytes);
int asd_memcpy_d2h(asd_ctx* ctx, void* dst, const void* src, uint64_t bytes);
int asd_sync(asd_ctx* ctx);
/* Test aid: MapPoint::PredictScale (MapPoint.cc:438-453) is evaluated on the device by comparing the distance ratio with
 * per-level thresholds derived from the host's logf at asd_ctx_create; this sweeps EVERY float in [lo, hi] and returns the
 * number of ratios for which the comparison rule and ceil(logf(r) / logf(scaleFactor)) (clamped) disagree (expected 0). */
int32_t asd_debug_level_sweep(const asd_ctx* ctx, float lo, float hi, int64_t* n_checked);
/* Runs `reps` back-to-back repetitions of the ASDNet forward on resident buffers and
 * returns the average per-repetition device time (hipEvents on the ctx stream). */
int asd_describe_timed(asd_ctx* ctx, const uint8_t* d_patches, int32_t n, float* d_desc,
                       int32_t reps, float* avg_ms);

#ifdef __cplusplus
}
#endif
#endif /* ASD_SLAM_H */
