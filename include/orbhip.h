/*
 * orbhip.h -- C ABI of the MI355X-native ORB front-end and descriptor matching.
 *
 * Drop-in boundary for ORB-SLAM2's hot path (citations are file:line in the
 * reference tree).  Every entry point is plain C: pointers + sizes, int status
 * (0 = ok, < 0 = error), never throws.  Host-pointer entry points are
 * synchronous; *_device entry points enqueue on the handle's HIP stream and
 * take/return device pointers.
 *
 *   orbhip_extractor_*        replaces class ORBextractor
 *                             (include/ORBextractor.h:45-110, src/ORBextractor.cc:410-1132)
 *   orbhip_extract            replaces ORBextractor::operator()  (include/ORBextractor.h:59-61)
 *   orbhip_pyramid_level*     replaces the public member mvImagePyramid (include/ORBextractor.h:85)
 *   orbhip_matcher_*          replaces class ORBmatcher (include/ORBmatcher.h:37-103)
 *   orbhip_descriptor_distance  ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:1647-1663)
 *   orbhip_search_for_initialization  ORBmatcher::SearchForInitialization (src/ORBmatcher.cc:405-520)
 *   orbhip_search_by_projection_frame ORBmatcher::SearchByProjection(Frame&,const Frame&,th,bMono)
 *                                     (src/ORBmatcher.cc:1328-1470)
 *   orbhip_search_by_projection_points ORBmatcher::SearchByProjection(Frame&,vector<MapPoint*>&,th)
 *                                     (src/ORBmatcher.cc:45-129)
 *   orbhip_compute_stereo_matches     Frame::ComputeStereoMatches (src/Frame.cc:466-640)
 *   orbhip_search_by_projection_keyframe ORBmatcher::SearchByProjection(Frame&,KeyFrame*,sAlreadyFound,th,ORBdist)
 *                                     (src/ORBmatcher.cc:1472-1599, relocalisation)
 *   orbhip_search_by_projection_sim3  ORBmatcher::SearchByProjection(KeyFrame*,Scw,vpPoints,vpMatched,th)
 *                                     (src/ORBmatcher.cc:290-403, loop closing)
 *   orbhip_search_best_in_window      inner search of ORBmatcher::Fuse x2 (src/ORBmatcher.cc:825-1100) and of both
 *                                     directions of SearchBySim3 (:1102-1326)
 *   orbhip_undistort_keypoints        Frame::UndistortKeyPoints (src/Frame.cc:404-434)
 *   orbhip_assign_features_to_grid    Frame::AssignFeaturesToGrid (src/Frame.cc:230-245)
 *   orbhip_compute_stereo_from_rgbd   Frame::ComputeStereoFromRGBD (src/Frame.cc:643-664)
 *   orbhip_distinctive_descriptors    MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:242-307), batched
 *   orbhip_vocabulary_*               ORBVocabulary (DBoW2::TemplatedVocabulary<FORB>) loadFromTextFile + transform,
 *                                     i.e. Frame::ComputeBoW (src/Frame.cc:395-402)
 *   orbhip_search_by_bow              ORBmatcher::SearchByBoW(KeyFrame*,Frame&,..) (src/ORBmatcher.cc:159-288) and
 *                                     SearchByBoW(KeyFrame*,KeyFrame*,..) (:522-655); vocabulary node ids are inputs
 *   orbhip_search_for_triangulation   ORBmatcher::SearchForTriangulation (:657-823) incl. CheckDistEpipolarLine (:140-157)
 */
#ifndef ORBHIP_H
#define ORBHIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORBHIP_OK 0
#define ORBHIP_E_ARG (-1)       /* bad argument */
#define ORBHIP_E_HIP (-2)       /* HIP runtime error (see orbhip_last_error) */
#define ORBHIP_E_CAPACITY (-3)  /* caller buffer / internal capacity too small */
#define ORBHIP_E_SIZE (-4)      /* unsupported image geometry */
#define ORBHIP_E_NODEVICE (-5)  /* no usable gfx950 device */

#define ORBHIP_MAX_LEVELS 16

/* Same 28-byte layout as cv::KeyPoint (pt.x, pt.y, size, angle, response, octave, class_id). */
typedef struct orbhip_keypoint {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} orbhip_keypoint;

typedef struct orbhip_extractor orbhip_extractor;
typedef struct orbhip_matcher orbhip_matcher;

const char *orbhip_last_error(void);      /* thread-local text of the last failure */
int orbhip_device_count(int *count);

/* ---- ORBextractor ------------------------------------------------------- */

/* ORBextractor::ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST)
 * (src/ORBextractor.cc:410-470) on HIP device `device`. */
int orbhip_extractor_create(int nfeatures, float scale_factor, int nlevels, int ini_th_fast,
                            int min_th_fast, int device, orbhip_extractor **out);
void orbhip_extractor_destroy(orbhip_extractor *e);

/* GetLevels/GetScaleFactor(s)/GetInverseScaleFactors/GetScaleSigmaSquares/
 * GetInverseScaleSigmaSquares (include/ORBextractor.h:63-83); arrays of nlevels floats,
 * any pointer may be NULL.  feat_per_level = mnFeaturesPerLevel. */
int orbhip_extractor_levels(const orbhip_extractor *e);
int orbhip_extractor_tables(const orbhip_extractor *e, float *scale, float *inv_scale,
                            float *sigma2, float *inv_sigma2, int32_t *feat_per_level);

/* Upper bound of keypoints one frame can return for images of rows x cols
 * (per level: max(quota + 3, 4 * initial nodes), src/ORBextractor.cc:669-737). */
int orbhip_extractor_capacity(orbhip_extractor *e, int rows, int cols, int *cap);

/* 7-tap blur weights (default {18,34,49,55,49,34,18}: OpenCV<=3.3 integer kernel for
 * GaussianBlur(7x7, sigma 2) on 8U; out = sat((sum_v sum_u w_v w_u p + 2^15) >> 16)). */
int orbhip_extractor_set_blur_kernel(orbhip_extractor *e, const int32_t w[7]);

/* mvImagePyramid[0] on demand (include/ORBextractor.h:85; only Frame::ComputeStereoMatches, src/Frame.cc:563-580, ever
 * reads it -- a monocular Tracking thread never does).  on != 0: extractions stop writing the padded level-0 plane
 * (copyMakeBorder of the image, src/ORBextractor.cc:1127); FAST and the descriptor kernel read level 0 from the caller's
 * image (BORDER_REFLECT_101 by index where a border keypoint's window overshoots it), results are bit-identical.  The
 * first accessor that needs the plane afterwards (orbhip_pyramid_level / _download / orbhip_blurred_level_download for
 * level 0, orbhip_compute_stereo_matches*) writes it then, from the image buffer of the last extraction: callers of
 * orbhip_extract_batch_device must leave that buffer untouched until then (host entry points keep their own copy).
 * Default 0: every extraction materialises it, as the reference does. */
int orbhip_extractor_set_lazy_level0(orbhip_extractor *e, int on);

/* Several extractor handles on several streams (the reference runs two on two threads for a stereo sensor,
 * src/Frame.cc:78-81; a batch front-end runs a few pipelines side by side): which of their kernels meet on the GPU decides
 * how well they share it, and free-running streams settle into one of several phase patterns.  A stage gate pins the
 * pattern: before launching stage `stage` (0 pyramid, 1 FAST, 2 octree, 3 descriptors) the handle's stream waits for
 * `wait_event` (hipEvent_t, null = no wait), after it `record_event` is recorded (null = none).  Chaining the FAST stages
 * of N handles in a ring (handle h waits for the event handle h-1 records) keeps the N VALU-bound FAST kernels from
 * running beside each other.  Events are owned by the caller and must outlive their use; results never depend on gates.
 * ORBHIP_GATE_KEEP for either event leaves that side of the stage's gate as it is. */
#define ORBHIP_GATE_KEEP ((void *)(intptr_t)-1)
int orbhip_extractor_set_stage_gate(orbhip_extractor *e, int stage, void *wait_event, void *record_event);

/* ORBextractor::operator()(image, mask(ignored), keypoints, descriptors)
 * (src/ORBextractor.cc:1043-1105).  image: rows x cols uint8, row stride `stride` bytes (host).
 * kps[cap], desc[cap*32] host buffers; *n = number of keypoints (0 is success). */
int orbhip_extract(orbhip_extractor *e, const uint8_t *image, int rows, int cols, int stride,
                   orbhip_keypoint *kps, uint8_t *desc, int cap, int *n);

/* Batch of `batch` same-size frames, host buffers.  Frame b starts at
 * images + b*frame_stride; outputs of frame b at kps + b*cap, desc + b*cap*32, n[b].
 * ORBHIP_E_CAPACITY (cap smaller than orbhip_extractor_capacity()): every frame's n[b] <= cap and its first n[b]
 * keypoints / descriptors are still delivered (truncated in output order) before the error is returned. */
int orbhip_extract_batch(orbhip_extractor *e, const uint8_t *images, int batch, int rows, int cols,
                         int stride, size_t frame_stride, orbhip_keypoint *kps, uint8_t *desc,
                         int cap, int32_t *n);

/* Same with device pointers; asynchronous on the handle's stream. d_n: int32[batch].
 * d_status: int32[batch] (0 ok, ORBHIP_E_CAPACITY if cap was too small), may be NULL. */
int orbhip_extract_batch_device(orbhip_extractor *e, const void *d_images, int batch, int rows,
                                int cols, int stride, size_t frame_stride, void *d_kps,
                                void *d_desc, int cap, void *d_n, void *d_status);
int orbhip_extractor_sync(orbhip_extractor *e);
void *orbhip_extractor_stream(orbhip_extractor *e); /* hipStream_t */
/* Launch on a caller-owned hipStream_t instead (NULL: back to the handle's own stream). */
int orbhip_extractor_set_stream(orbhip_extractor *e, void *stream);

/* mvImagePyramid[level] of frame `frame` of the last extract call: size and device pointer of
 * the level ROI (valid until the next extract on this handle); row stride in bytes.  The ROI is
 * surrounded by the 19-px BORDER_REFLECT_101 frame the reference allocates
 * (src/ORBextractor.cc:1113-1128), so d_roi[-19*stride-19] is addressable. */
int orbhip_pyramid_level(orbhip_extractor *e, int frame, int level, int *rows, int *cols,
                         int *stride, const void **d_roi);
/* Copy a level to host.  with_border != 0 copies the (rows+38)x(cols+38) padded plane. */
int orbhip_pyramid_level_download(orbhip_extractor *e, int frame, int level, int with_border,
                                  uint8_t *dst, int dst_stride);
/* Debug/parity taps of the last call (host copies).  The blurred planes are not part of the pipeline (only the
 * keypoints' patches are blurred, inside the descriptor kernel): the first request after an extraction runs the
 * 7x7 Gaussian blur over the whole pyramid of the last batch. */
int orbhip_blurred_level_download(orbhip_extractor *e, int frame, int level, uint8_t *dst,
                                  int dst_stride);
/* FAST candidates of a level in reference order: x,y (relative to minBorder), score. */
int orbhip_level_candidates(orbhip_extractor *e, int frame, int level, int32_t *x, int32_t *y,
                            int32_t *score, int cap, int *n);

/* Per-stage device time, microseconds, averaged over the extract calls made since
 * orbhip_extractor_set_profiling(e, 1) (at most the last 256), measured with HIP events on the
 * handle's stream: [0] pyramid (7 launches), [1] FAST+NMS cells, [2] octree, [3] always 0 (the separate blur stage of
 * round 1: the descriptor kernel blurs the patches it samples, nothing is launched or timed there), [4] blur of the
 * patches + orientation + descriptors, [5] whole call. */
int orbhip_extractor_set_profiling(orbhip_extractor *e, int on);
int orbhip_extractor_stage_times(orbhip_extractor *e, float us[6]);

/* ---- ORBmatcher --------------------------------------------------------- */

/* Flat view of the Frame fields the matchers read (include/Frame.h). Host pointers. */
typedef struct orbhip_frame_view {
    int32_t n;                       /* N */
    const orbhip_keypoint *keys;     /* mvKeysUn */
    const uint8_t *desc;             /* mDescriptors, n x 32 */
    const float *u_right;            /* mvuRight or NULL (monocular: all -1) */
    float min_x, min_y, max_x, max_y; /* mnMinX, mnMinY, mnMaxX, mnMaxY */
    float grid_inv_w, grid_inv_h;    /* mfGridElementWidthInv, mfGridElementHeightInv */
    int32_t n_levels;
    const float *scale_factors;      /* mvScaleFactors */
} orbhip_frame_view;

/* One already-projected query of SearchByProjection (the shim does the projection with the
 * reference's own cv::Mat arithmetic, src/ORBmatcher.cc:1360-1390 / src/Frame.cc:269-325). */
typedef struct orbhip_query {
    int32_t valid;        /* 0: skipped (no map point, outlier, bad, not in view, behind camera) */
    float u, v;           /* projection in the searched frame */
    float radius;         /* window half-size in px (th * scale[level], src/ORBmatcher.cc:1381,:69) */
    int32_t min_level, max_level; /* GetFeaturesInArea level window (max_level -1 = open) */
    float ur;             /* projected right-image coordinate (stereo consistency check) */
    int32_t level_aux;    /* informational */
    float angle;          /* angle of the query keypoint (rotation histogram) */
    int32_t observed;     /* pMP->Observations() > 0: once assigned it blocks that slot */
} orbhip_query;

int orbhip_matcher_create(int device, orbhip_matcher **out);
void orbhip_matcher_destroy(orbhip_matcher *m);

/* Batched DescriptorDistance: dist[i*nb + j] = popcount(a_i xor b_j) (host buffers). */
int orbhip_descriptor_distance(orbhip_matcher *m, const uint8_t *a, int na, const uint8_t *b,
                               int nb, int32_t *dist);

/* ORBmatcher::SearchForInitialization(F1,F2,vbPrevMatched,vnMatches12,windowSize) with
 * mfNNratio = nnratio, mbCheckOrientation = check_ori.  prev_matched_xy: n1 x 2 floats, in/out.
 * matches12: n1 int32 out. */
int orbhip_search_for_initialization(orbhip_matcher *m, const orbhip_frame_view *f1,
                                     const orbhip_frame_view *f2, float *prev_matched_xy,
                                     int32_t *matches12, int window_size, float nnratio,
                                     int check_ori, int *nmatches);

/* ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono) after projection.
 * q[nq], qdesc[nq*32] = pMP->GetDescriptor(); taken[n] (may be NULL): slot already holds an
 * observed map point.  assign[n] out: query index now held by each current keypoint or -1.
 * Sizes (this and the points / keyframe / sim3 forms below): any n and nq.  Queries with valid == 0 are dropped
 * on the host before anything is staged (a local map or a loop-closing point set is mostly such entries); up to
 * 4096 train keypoints and 4096 VALID queries the resolve state is LDS resident, beyond that it moves to an HBM
 * workspace (slower, same results).  Only orbhip_search_for_initialization keeps a hard limit (4096 keypoints per
 * frame, ORBHIP_E_CAPACITY beyond): its match stealing is replayed by one wavefront over LDS state. */
int orbhip_search_by_projection_frame(orbhip_matcher *m, const orbhip_frame_view *cur,
                                      const orbhip_query *q, const uint8_t *qdesc, int nq,
                                      const uint8_t *taken, int32_t *assign, int check_ori,
                                      int *nmatches);

/* ORBmatcher::SearchByProjection(F, vpMapPoints, th) after Frame::isInFrustum. */
int orbhip_search_by_projection_points(orbhip_matcher *m, const orbhip_frame_view *f,
                                       const orbhip_query *q, const uint8_t *qdesc, int nq,
                                       const uint8_t *taken, int32_t *assign, float nnratio,
                                       int *nmatches);

/* ORBmatcher::SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist) after projection
 * (src/ORBmatcher.cc:1490-1527 stay in the caller: valid = map point exists, !isBad(), not in sAlreadyFound, projects
 * inside the image and inside its scale-invariance range; radius = th*scale[predicted]; levels [pred-1, pred+1]).
 * taken[n]: CurrentFrame.mvpMapPoints[i2] != NULL on entry.  Every accepted match blocks its slot (:1538-1539);
 * accept iff best distance <= orb_dist (:1552); rotation histogram as in the frame-to-frame search. */
int orbhip_search_by_projection_keyframe(orbhip_matcher *m, const orbhip_frame_view *cur, const orbhip_query *q,
                                         const uint8_t *qdesc, int nq, const uint8_t *taken, int32_t *assign,
                                         int orb_dist, int check_ori, int *nmatches);

/* ORBmatcher::SearchByProjection(pKF, Scw, vpPoints, vpMatched, th) after the Sim3 projection (:315-359 stay in the
 * caller; radius = th*scale[pred], levels [pred-1, pred]).  kf: the KeyFrame's keypoints/descriptors/grid;
 * matched[n]: vpMatched[idx] != NULL on entry; accept iff best distance <= TH_LOW (:393); no orientation check. */
int orbhip_search_by_projection_sim3(orbhip_matcher *m, const orbhip_frame_view *kf, const orbhip_query *q,
                                     const uint8_t *qdesc, int nq, const uint8_t *matched, int32_t *assign,
                                     int *nmatches);

/* Independent best match per query (no slot blocking): the search loop of ORBmatcher::Fuse(pKF, vpMapPoints, th)
 * (src/ORBmatcher.cc:893-950), Fuse(pKF, Scw, ...) (:1045-1075) and of both directions of SearchBySim3 (:1199-1219,
 * :1279-1299).  Candidates = KeyFrame::GetFeaturesInArea(u, v, radius) with octave in [min_level, max_level]
 * (= [pred-1, pred]); chi2_gate != 0 adds Fuse's reprojection gate (e2*mvInvLevelSigma2[level] > 5.99 mono / 7.8
 * when mvuRight[idx] >= 0, using q.ur).  best_idx[nq] (-1 if no candidate), best_dist[nq] (256 if none): the caller
 * applies its threshold (TH_LOW / TH_HIGH) and the map-graph side effects in order. */
int orbhip_search_best_in_window(orbhip_matcher *m, const orbhip_frame_view *kf, const orbhip_query *q,
                                 const uint8_t *qdesc, int nq, int chi2_gate, const float *inv_level_sigma2,
                                 int32_t *best_idx, int32_t *best_dist);

/* Vocabulary-guided matching.  The DBoW2 vocabulary lookup (Frame::ComputeBoW, src/Frame.cc:395-403) stays with
 * the caller; what arrives here is the FeatureVector flattened to one node id per keypoint (node1[n1], node2[n2]:
 * the NodeId at levelsup = 4 that DBoW2 stored the feature under, ORBHIP_NO_NODE if absent).  Node lists are visited
 * in ascending node id and ascending feature index, which is how DBoW2 builds and std::map iterates them.
 *
 * orbhip_search_by_bow: for every f1 keypoint with valid1 != 0 (map point exists and is not bad; null = all), best
 * and second-best Hamming distance among the f2 keypoints of the same node that are neither blocked2 (null = none;
 * key-frame overload: "no good map point in pKF2") nor matched by an earlier f1 keypoint; accepted when
 * best <= max_dist (TH_LOW = 50 for the Frame overload :262, 49 for the key-frame overload's strict "<" :598) and
 * (float)best < nnratio*(float)second; rotation-histogram cull when check_ori.  matches12[n1] = f2 index or -1
 * (the Frame overload's vpMapPointMatches[idx2] = map point of idx1 is the inverse of this 1:1 map).
 * No size limit beyond memory (one wavefront per common node). */
#define ORBHIP_NO_NODE 0xffffffffu
int orbhip_search_by_bow(orbhip_matcher *m, const orbhip_frame_view *f1, const uint32_t *node1, const uint8_t *valid1,
                         const orbhip_frame_view *f2, const uint32_t *node2, const uint8_t *blocked2, int max_dist,
                         float nnratio, int check_ori, int32_t *matches12, int *nmatches);

/* Device-resident, batched orbhip_search_by_bow: pair p matches frame f1_first + p*f1_step of the first set of
 * arrays (the key-frame side) against frame f2_first + p*f2_step of the second set; both sets are in the layout the
 * extractor and orbhip_vocabulary_transform_device write (d_kps [frames][cap] orbhip_keypoint, d_desc [frames][cap][32],
 * d_n [frames] int32, d_node [frames][cap] uint32) and may be the same arrays; d_valid1 / d_blocked2 [frames][cap]
 * uint8 are optional (null).  Outputs d_matches12 [pairs][cap] int32, d_nmatches [pairs] int32.  One launch on the
 * matcher's stream, asynchronous; the (node, index) ordering, grouping, matching and rotation cull all happen on
 * the device.  cap <= 4096. */
int orbhip_search_by_bow_device(orbhip_matcher *m, int pairs, int cap, const void *d_kps1, const void *d_desc1,
                                const void *d_n1, const void *d_node1, const void *d_valid1, int f1_first, int f1_step,
                                const void *d_kps2, const void *d_desc2, const void *d_n2, const void *d_node2,
                                const void *d_blocked2, int f2_first, int f2_step, int max_dist, float nnratio,
                                int check_ori, void *d_matches12, void *d_nmatches);

/* ORBmatcher::SearchForTriangulation (src/ORBmatcher.cc:657-823).  valid1 / valid2: keypoint has no map point yet
 * (null = all); stereo flags come from the views' u_right (>= 0, null = monocular); f12: the fundamental matrix of
 * LocalMapping::ComputeF12 (src/LocalMapping.cc:536-553) row-major; (ex, ey): epipole of camera 1 in image 2
 * (:664-670); level_sigma2[f2->n_levels] = pKF2->mvLevelSigma2; f2->scale_factors must be set.  Candidates of the
 * same node with Hamming distance <= TH_LOW that pass the epipole gate (:741-747) and CheckDistEpipolarLine
 * (:140-157); smallest distance wins, the later index on ties (":735 dist>bestDist"); the reference never sets
 * vbMatched2, so queries do not block each other.  matches12[n1] = f2 index or -1 (vMatchedPairs = the non-negative
 * entries in index order).  n1, n2 <= 4096 (ORBHIP_E_CAPACITY otherwise). */
int orbhip_search_for_triangulation(orbhip_matcher *m, const orbhip_frame_view *f1, const uint32_t *node1,
                                    const uint8_t *valid1, const orbhip_frame_view *f2, const uint32_t *node2,
                                    const uint8_t *valid2, const float *f12, float ex, float ey,
                                    const float *level_sigma2, int only_stereo, int check_ori, int32_t *matches12,
                                    int *nmatches);

/* MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:242-307), batched: map point p owns the observed
 * descriptors desc[offsets[p] .. offsets[p+1]) (32 bytes each, in the order the reference collects them, i.e.
 * std::map<KeyFrame*,size_t> order without bad key frames).  best_idx[p] = index inside that range of the descriptor
 * with the least median Hamming distance to the others (vDists[0.5*(N-1)], first minimum), -1 for an empty range.
 * At most 2048 observations per map point (ORBHIP_E_CAPACITY otherwise). */
int orbhip_distinctive_descriptors(orbhip_matcher *m, const uint8_t *desc, const int32_t *offsets, int npoints,
                                   int32_t *best_idx);

/* ---- Frame constructor glue either side of the path (rectified / undistorted cameras) -------------------------
 * Frame::AssignFeaturesToGrid (src/Frame.cc:230-245, PosInGrid :382-392): the 64 x 48 grid mGrid as CSR in the order
 * GetFeaturesInArea walks it.  cell c = posX*48 + posY; cell_of[n] = c or -1 (PosInGrid rejects the keypoint);
 * cell_start[64*48 + 1]; cell_items[n] (the first cell_start[3072] entries are used): keypoint indices of every cell
 * in push_back (ascending) order.  f->keys are mvKeysUn.  n <= 4096. */
int orbhip_assign_features_to_grid(orbhip_matcher *m, const orbhip_frame_view *f, int32_t *cell_of, int32_t *cell_start,
                                   int32_t *cell_items);
/* device-resident, batched: d_kps [frames][cap], d_n [frames]; outputs d_cell_of / d_cell_items [frames][cap] int32,
 * d_cell_start [frames][3073] int32.  cap <= 4096.  Asynchronous on the matcher's stream. */
int orbhip_assign_features_to_grid_device(orbhip_matcher *m, int frames, const void *d_kps, const void *d_n, int cap,
                                          float min_x, float min_y, float grid_inv_w, float grid_inv_h, void *d_cell_of,
                                          void *d_cell_start, void *d_cell_items);
/* Frame::UndistortKeyPoints (src/Frame.cc:404-434): keys_un = keys with pt replaced by
 * cv::undistortPoints(pt, mK, mDistCoef, R = I, P = mK); dist5 = {k1, k2, p1, p2, k3} (mDistCoef, src/Tracking.cc:66-81,
 * k3 = 0 when absent); dist5[0] == 0 copies the input (:406-410).  cv::undistortPoints is restated from the published
 * algorithm of OpenCV 2.4 - 3.3 (double arithmetic, 5 fixed-point iterations): parity with a given OpenCV build is
 * unpinned.  The four image corners of Frame::ComputeImageBounds (:436-463) go through the same call. */
int orbhip_undistort_keypoints(orbhip_matcher *m, const orbhip_keypoint *keys, int n, float fx, float fy, float cx, float cy,
                               const float *dist5, orbhip_keypoint *keys_un);
/* device-resident, batched: d_kps / d_kps_un [frames][cap] (may be the same array), d_n [frames] */
int orbhip_undistort_keypoints_device(orbhip_matcher *m, int frames, const void *d_kps, const void *d_n, int cap, float fx,
                                      float fy, float cx, float cy, const float *dist5, void *d_kps_un);
/* Frame::ComputeStereoFromRGBD (src/Frame.cc:643-664): depth = CV_32F image (rows x cols, stride in floats), sampled at
 * the truncated coordinates of keys (mvKeys); u_right[i] = keys_un[i].x - mbf/d, depth_out[i] = d where d > 0, else
 * -1 / -1.  keys_un null = keys.  A keypoint outside the image reads d = 0 (the reference would read out of bounds). */
int orbhip_compute_stereo_from_rgbd(orbhip_matcher *m, const orbhip_keypoint *keys, const orbhip_keypoint *keys_un, int n,
                                    const float *depth, int rows, int cols, int stride_floats, float mbf, float *u_right,
                                    float *depth_out);
/* device-resident, batched: frame f reads d_depth + f * frame_stride_floats; d_kps_un null = d_kps */
int orbhip_compute_stereo_from_rgbd_device(orbhip_matcher *m, int frames, const void *d_kps, const void *d_kps_un,
                                           const void *d_n, int cap, const void *d_depth, int rows, int cols,
                                           int stride_floats, size_t frame_stride_floats, float mbf, void *d_u_right,
                                           void *d_depth_out);

/* ---- DBoW2 vocabulary: ORBVocabulary::loadFromTextFile + transform ---------------------------------------------
 * Replaces, for Frame::ComputeBoW / KeyFrame::ComputeBoW (src/Frame.cc:395-402, src/KeyFrame.cc ComputeBoW), the
 * calls mpORBvocabulary->transform(vCurrentDesc, mBowVec, mFeatVec, 4) into
 * Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1127-1199 (+ :1218-1262 per feature, BowVector.cpp:36-88,
 * FeatureVector.cpp:31-46, FORB.cpp:81-101).  scoring / weighting use DBoW2's enum values (BowVector.h:36-53:
 * scoring 0 L1_NORM .. 5 DOT_PRODUCT; weighting 0 TF_IDF, 1 TF, 2 IDF, 3 BINARY; ORBvoc.txt is "10 6 0 0"). */
typedef struct orbhip_vocabulary orbhip_vocabulary;

/* Text format of TemplatedVocabulary::loadFromTextFile (:1338-1424): first line "k L scoring weighting", then one
 * line per node "parent isLeaf d0 .. d31 weight" (node ids in file order from 1, root = 0, word ids in order of the
 * leaf lines).  Blank lines are skipped (the reference turns the empty line after the final newline into an extra
 * root child with an uninitialised descriptor). */
int orbhip_vocabulary_load_text(const char *path, int device, orbhip_vocabulary **out);
/* The same tree from arrays: entry i describes node i+1; parent[i] in [0, i]. */
int orbhip_vocabulary_create(int k, int L, int scoring, int weighting, int n_nodes, const int32_t *parent,
                             const uint8_t *is_leaf, const uint8_t *desc, const double *weight, int device,
                             orbhip_vocabulary **out);
void orbhip_vocabulary_destroy(orbhip_vocabulary *v);
/* any output pointer may be null; n_nodes counts the root */
int orbhip_vocabulary_info(const orbhip_vocabulary *v, int *k, int *L, int *scoring, int *weighting, int *n_nodes,
                           int *n_words);
int orbhip_vocabulary_set_stream(orbhip_vocabulary *v, void *hip_stream);   /* null = the handle's own stream */
int orbhip_vocabulary_sync(orbhip_vocabulary *v);

/* transform(features, BowVector, FeatureVector, levelsup) for n descriptors (n x 32 bytes, n <= 8192).
 * Per feature (each array n entries, nullable): word_id, word_weight, node_id = the FeatureVector node (ancestor at
 * level L - levelsup, 0 = root when levelsup >= L) or ORBHIP_NO_NODE when the word is stopped (weight <= 0); this is
 * the node1 / node2 input of orbhip_search_by_bow.  BowVector: bow_ids ascending with bow_vals (capacity n each,
 * nullable), *n_bow entries; values are accumulated and normalised in the reference's order, so they are
 * bit-identical doubles. */
int orbhip_vocabulary_transform(orbhip_vocabulary *v, const uint8_t *desc, int n, int levelsup, uint32_t *word_id,
                                double *word_weight, uint32_t *node_id, uint32_t *bow_ids, double *bow_vals, int *n_bow);
/* Device-resident, batched: descriptors in the extractor's output layout d_desc [frames][cap][32] with d_n [frames]
 * int32 counts; outputs d_word_id / d_node_id / d_bow_ids [frames][cap] uint32, d_word_weight / d_bow_vals
 * [frames][cap] double, d_n_bow [frames] int32.  Asynchronous on the handle's stream.  cap <= 8192. */
int orbhip_vocabulary_transform_device(orbhip_vocabulary *v, int frames, const void *d_desc, const void *d_n, int cap,
                                       int levelsup, void *d_word_id, void *d_word_weight, void *d_node_id,
                                       void *d_bow_ids, void *d_bow_vals, void *d_n_bow);

/* Device-resident, batched forms of the two SearchByProjection searches: `pairs` independent frame pairs,
 * asynchronous on the matcher's stream.  Train side in the extractor's output layout: d_kps [pairs][cap]
 * orbhip_keypoint, d_desc [pairs][cap][32], d_n [pairs] int32; optional d_u_right [pairs][cap] float and
 * d_taken [pairs][cap] uint8.  Query side: d_q [pairs][qcap] orbhip_query, d_qdesc [pairs][qcap][32],
 * d_nq [pairs] int32.  The image bounds / grid scale are shared (one camera).  Outputs: d_assign [pairs][cap]
 * int32 (query index held by each keypoint or -1), d_nmatches [pairs] int32.  No size limit: up to 4096 keypoints
 * and queries the resolve state is LDS resident, beyond that it lives in an HBM workspace (slower, same results). */
int orbhip_search_by_projection_frame_device(orbhip_matcher *m, int pairs, const void *d_kps, const void *d_desc,
                                             const void *d_n, int cap, const void *d_u_right, const void *d_taken,
                                             float min_x, float min_y, float grid_inv_w, float grid_inv_h,
                                             const void *d_q, const void *d_qdesc, const void *d_nq, int qcap,
                                             int check_ori, void *d_assign, void *d_nmatches);
int orbhip_search_by_projection_points_device(orbhip_matcher *m, int pairs, const void *d_kps, const void *d_desc,
                                              const void *d_n, int cap, const void *d_u_right, const void *d_taken,
                                              float min_x, float min_y, float grid_inv_w, float grid_inv_h,
                                              const void *d_q, const void *d_qdesc, const void *d_nq, int qcap,
                                              float nnratio, void *d_assign, void *d_nmatches);
/* ORBmatcher::SearchForInitialization, device resident and batched (src/ORBmatcher.cc:405-520): pair p matches
 * F1 = frame f1_first + p*f1_step against F2 = frame f2_first + p*f2_step of the extractor output arrays.
 * d_prev_matched [pairs][cap][2] float = vbPrevMatched (in/out; reset_prev != 0 first sets it to F1's keypoint
 * positions, src/Tracking.cc:578-580); outputs d_matches12 [pairs][cap] int32 (vnMatches12) and d_nmatches [pairs].
 * One wavefront per pair replays the match-stealing loop.  cap <= 4096. */
int orbhip_search_for_initialization_device(orbhip_matcher *m, int pairs, const void *d_kps, const void *d_desc,
                                            const void *d_n, int cap, int f1_first, int f1_step, int f2_first,
                                            int f2_step, float min_x, float min_y, float grid_inv_w, float grid_inv_h,
                                            int reset_prev, void *d_prev_matched, int window_size, float nnratio,
                                            int check_ori, void *d_matches12, void *d_nmatches);

/* ---- projection prologues on the device -------------------------------------------
 * The arithmetic in front of the window search of the two SearchByProjection overloads.  The reference does it with
 * cv::Mat expressions (CV_32F); the operation order used here is stated in DESIGN.md section 3 (per row
 * ((r0*x + r1*y) + r2*z) + t in float without contraction; cv::norm / Mat::dot accumulate in double; logf of
 * MapPoint::PredictScale is a deterministic, correctly rounded log). */
typedef struct orbhip_camera {
    float fx, fy, cx, cy, mbf, mb;           /* Frame::fx, fy, cx, cy, mbf, mb (src/Frame.cc:104-112) */
    float min_x, max_x, min_y, max_y;        /* mnMinX, mnMaxX, mnMinY, mnMaxY */
    int32_t n_levels;                        /* mnScaleLevels */
    float log_scale_factor;                  /* mfLogScaleFactor */
    float scale_factors[ORBHIP_MAX_LEVELS];  /* mvScaleFactors */
} orbhip_camera;
#define ORBHIP_POINT_PRESENT 1   /* the map point exists and takes part (frame search: pMP && !mvbOutlier[i];
                                    frustum: !isBad() and not already matched in this frame) */
#define ORBHIP_POINT_OBSERVED 2  /* pMP->Observations() > 0 */

/* Prologue of ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono), src/ORBmatcher.cc:1339-1390:
 * Tcw / Tlw = top three rows of CurrentFrame.mTcw / LastFrame.mTcw, row-major (12 floats); world [n][3] =
 * pMP->GetWorldPos() of LastFrame.mvpMapPoints[i]; flags [n] = ORBHIP_POINT_* bits; last_keys = LastFrame.mvKeysUn
 * (octave: :1379, angle: :1433).  q [n] out, ready for orbhip_search_by_projection_frame with
 * qdesc = the map points' descriptors.  Host buffers, synchronous. */
int orbhip_project_last_frame(orbhip_matcher *m, const orbhip_camera *cam, const float *Tcw, const float *Tlw, int n,
                              const float *world, const uint8_t *flags, const orbhip_keypoint *last_keys, float th,
                              int mono, orbhip_query *q);

/* Frame::isInFrustum(pMP, viewing_cos_limit) (src/Frame.cc:269-325) with MapPoint::PredictScale
 * (src/MapPoint.cc:400-418) for n map points of the local map, followed by the window of
 * SearchByProjection(F, vpMapPoints, th) (src/ORBmatcher.cc:52-69, RadiusByViewingCos :131-137).  world / normal [n][3]
 * = GetWorldPos() / GetNormal(); max_dist / min_dist [n] = mfMaxDistance / mfMinDistance (the 1.2 / 0.8 factors of
 * Get{Max,Min}DistanceInvariance are applied here).  q[i].valid = mbTrackInView, u / v / ur / level_aux =
 * mTrackProjX / mTrackProjY / mTrackProjXR / mnTrackScaleLevel; view_cos [n] (nullable) = mTrackViewCos. */
int orbhip_frustum_queries(orbhip_matcher *m, const orbhip_camera *cam, const float *Tcw, int n, const float *world,
                           const float *normal, const float *max_dist, const float *min_dist, const uint8_t *flags,
                           float viewing_cos_limit, float th, orbhip_query *q, float *view_cos);

/* Prologue of ORBmatcher::Fuse(pKF, vpMapPoints, th) (src/ORBmatcher.cc:853-888), Fuse(pKF, Scw, ...) (:1005-1048) and
 * of one direction of SearchBySim3 (:1155-1180 / :1235-1260): projection of n map points into a key frame, written
 * as orbhip_query records for orbhip_search_best_in_window (level window [pred-1, pred], radius th*scale[pred]).
 *   mode 0 (Fuse): T1 = [Rcw | tcw] (for the Sim3 overload the caller passes Rcw = sRcw/scw, tcw = t/scw, :1000-1003);
 *     gates: depth, KeyFrame::IsInImage, scale-invariance range on |X - Ow|, viewing angle PO.dot(Pn) >= 0.5*dist3D;
 *     q.ur = u - bf*invz for the stereo chi-square gate of the search.
 *   mode 1 (SearchBySim3): T1 = [R1w | t1w], T2 = [sR21 | t21] (the caller computes them as :1119-1121); gates: depth,
 *     IsInImage, range on |Pc2|; no normal.
 * double_invz: invz = 1.0/z evaluated in double (:1156, :1017) or 1/z in float (:862).  flags: ORBHIP_POINT_PRESENT =
 * map point exists, !isBad(), not IsInKeyFrame / not already matched.  max_dist / min_dist = mfMaxDistance /
 * mfMinDistance.  Host buffers, synchronous. */
int orbhip_keyframe_queries(orbhip_matcher *m, const orbhip_camera *cam, int mode, int double_invz, const float *T1,
                            const float *T2, int n, const float *world, const float *normal, const float *max_dist,
                            const float *min_dist, const uint8_t *flags, float th, orbhip_query *q);

/* ORBmatcher::Fuse up to the decision (src/ORBmatcher.cc:825-947 / :975-1075): prologue + best key point in the window
 * with the chi-square gate; best_idx[n] / best_dist[n] out (-1 / 256 when nothing passes).  The replace-or-add decision
 * (:949-971, :1077-1095: bestDist <= TH_LOW, pKF->GetMapPoint(bestIdx), Replace / AddObservation) touches the map
 * graph, depends on the order of the points and stays in the caller.  sim3_form != 0: the Scw overload (invz in double).
 * point_desc [n][32] = pMP->GetDescriptor(). */
int orbhip_fuse(orbhip_matcher *m, const orbhip_frame_view *kf, const orbhip_camera *cam, const float *Tcw, int sim3_form,
                int n, const float *world, const float *normal, const float *max_dist, const float *min_dist,
                const uint8_t *flags, const uint8_t *point_desc, float th, const float *inv_level_sigma2, int32_t *best_idx,
                int32_t *best_dist);

/* ORBmatcher::SearchBySim3 (src/ORBmatcher.cc:1102-1326), complete: both projection directions, both searches
 * (accept bestDist <= TH_HIGH) and the mutual-agreement pass.  kf1 / kf2: the key frames' keypoints, descriptors and
 * grid; per key-frame slot i: world*[i], max/min_dist*[i], desc*[i] = the slot's map point (GetWorldPos,
 * mfMax/MinDistance, GetDescriptor), flags*[i] = ORBHIP_POINT_PRESENT iff the slot has a good map point that is not in
 * vpMatches12 already (:1132-1143).  matches12[n1] = matched kf2 slot or -1 (the caller sets vpMatches12[i1] =
 * vpMapPoints2[matches12[i1]]); *nfound = the return value. */
int orbhip_search_by_sim3(orbhip_matcher *m, const orbhip_frame_view *kf1, const orbhip_frame_view *kf2,
                          const orbhip_camera *cam, const float *T1w, const float *T2w, const float *S21, const float *S12,
                          const float *world1, const float *max_dist1, const float *min_dist1, const uint8_t *flags1,
                          const uint8_t *desc1, const float *world2, const float *max_dist2, const float *min_dist2,
                          const uint8_t *flags2, const uint8_t *desc2, float th, int32_t *matches12, int *nfound);

/* Device-resident, batched.  Frames live in the extractor's output layout (d_kps [frames][cap] orbhip_keypoint,
 * d_desc [frames][cap][32], d_n [frames] int32); pair p has CurrentFrame = frame cur_first + p*cur_step and
 * LastFrame = frame last_first + p*last_step.  d_Tcw / d_Tlw [pairs][12] float; d_world [frames][cap][3] float and
 * d_flags [frames][cap] uint8 are indexed by the LAST frame (one map point per last-frame keypoint).
 *
 * orbhip_project_last_frame_device writes d_q [pairs][cap] orbhip_query and d_nq [pairs] int32 (= the last frame's n).
 *
 * orbhip_track_last_frame_device = that prologue + SearchByProjection's search, resolve and rotation cull in one call
 * (Tracking::TrackWithMotionModel's matching step, src/Tracking.cc:880-885): the queries' descriptors are the last
 * frame's (the tracked map points were created from / last seen in it); optional d_u_right [frames][cap] float
 * (indexed by the current frame) and d_taken [pairs][cap] uint8; outputs d_assign [pairs][cap] int32 (last-frame
 * keypoint index now held by each current keypoint or -1), d_nmatches [pairs] int32. */
int orbhip_project_last_frame_device(orbhip_matcher *m, int pairs, const orbhip_camera *cam, const void *d_Tcw,
                                     const void *d_Tlw, const void *d_kps, const void *d_n, int cap, int last_first,
                                     int last_step, const void *d_world, const void *d_flags, float th, int mono,
                                     void *d_q, void *d_nq);
int orbhip_track_last_frame_device(orbhip_matcher *m, int pairs, const orbhip_camera *cam, const void *d_Tcw,
                                   const void *d_Tlw, const void *d_kps, const void *d_desc, const void *d_n, int cap,
                                   int cur_first, int cur_step, int last_first, int last_step, const void *d_world,
                                   const void *d_flags, const void *d_u_right, const void *d_taken, float th, int mono,
                                   int check_ori, void *d_assign, void *d_nmatches);
/* Frame::isInFrustum for `frames` frames at once: d_Tcw [frames][12]; the points of frame f are d_world / d_normal
 * [frames][pcap][3], d_max_dist / d_min_dist [frames][pcap], d_flags [frames][pcap] uint8, d_np [frames] int32.
 * Outputs d_q [frames][pcap] orbhip_query, d_view_cos [frames][pcap] float (nullable). */
int orbhip_frustum_queries_device(orbhip_matcher *m, int frames, const orbhip_camera *cam, const void *d_Tcw, int pcap,
                                  const void *d_np, const void *d_world, const void *d_normal, const void *d_max_dist,
                                  const void *d_min_dist, const void *d_flags, float viewing_cos_limit, float th,
                                  void *d_q, void *d_view_cos);

/* Launch on a caller-owned hipStream_t (NULL: the handle's own stream); wait for the handle's stream. */
int orbhip_matcher_set_stream(orbhip_matcher *m, void *stream);
int orbhip_matcher_sync(orbhip_matcher *m);

/* Frame::ComputeStereoMatches.  The image pyramids are the ones left in the two extractor
 * handles by their last extract call (frame indices frame_l / frame_r of those batches), i.e.
 * mpORBextractorLeft/Right->mvImagePyramid.  keys/desc: host buffers (mvKeys, mDescriptors,
 * mvKeysRight, mDescriptorsRight).  mb is passed explicitly (the reference reads it before
 * assignment, src/Frame.cc:496 vs :114).  u_right[nl], depth[nl] out (mvuRight, mvDepth).
 * At most 2^20 right keypoints (ORBHIP_E_CAPACITY beyond: the match key holds distance << 20 | index). */
int orbhip_compute_stereo_matches(orbhip_matcher *m, orbhip_extractor *left, int frame_l,
                                  orbhip_extractor *right, int frame_r,
                                  const orbhip_keypoint *keys_l, const uint8_t *desc_l, int nl,
                                  const orbhip_keypoint *keys_r, const uint8_t *desc_r, int nr,
                                  float mbf, float mb, float *u_right, float *depth, int *nmatches);

/* Device-resident, batched Frame::ComputeStereoMatches over `pairs` stereo pairs: pair p uses frame l0 + p*ls of
 * the left extractor's last batch (pyramid and rows of d_kps_l / d_desc_l / d_n_l) and frame r0 + p*rs of the right
 * one (left and right may be the same handle holding an interleaved batch: l0=0, ls=2, r0=1, rs=2).  Arrays are in
 * the extractor's output layout with stride `cap`.  Outputs: d_u_right / d_depth [pairs][cap] float (entries beyond
 * the left frame's count are untouched), d_nmatches [pairs] int32.  Asynchronous on the matcher's stream; the caller
 * orders it after the extractions (same stream, or orbhip_extractor_sync).  cap <= 2^20 (ORBHIP_E_CAPACITY beyond). */
int orbhip_compute_stereo_matches_device(orbhip_matcher *m, orbhip_extractor *left, int l0, int ls,
                                         orbhip_extractor *right, int r0, int rs, int pairs, const void *d_kps_l,
                                         const void *d_desc_l, const void *d_n_l, const void *d_kps_r,
                                         const void *d_desc_r, const void *d_n_r, int cap, float mbf, float mb,
                                         void *d_u_right, void *d_depth, void *d_nmatches);

#ifdef __cplusplus
}
#endif
#endif /* ORBHIP_H */
