// C++ host-side mirror of ORB_SLAM2::ORBextractor (include/ORBextractor.h:45-110) and
// ORB_SLAM2::ORBmatcher (include/ORBmatcher.h:37-103) above the C ABI of orbhip.h.
// Header-only; no OpenCV dependency: images are (pointer, rows, cols, step) views and keypoints
// are orbhip_keypoint PODs, which have the memory layout of cv::KeyPoint.  INTEGRATION.md shows
// the 20-line adapter that gives these classes the reference's cv::InputArray/OutputArray
// signatures inside the ORB-SLAM2 tree.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../orbhip.h"

namespace orbhip {

typedef orbhip_keypoint KeyPoint;

struct ImageView {   // the part of cv::Mat the extractor reads (CV_8UC1)
    const uint8_t *data;
    int rows, cols;
    size_t step;
    bool empty() const { return !data || rows <= 0 || cols <= 0; }
};

struct Error : std::runtime_error {
    int code;
    Error(int c, const char *where) : std::runtime_error(std::string(where) + ": " + orbhip_last_error()), code(c) {}
};
inline void check(int rc, const char *where) { if (rc != ORBHIP_OK) throw Error(rc, where); }

class ORBextractor {
public:
    enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

    // ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST)
    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST, int device = 0)
        : h_(nullptr), nlevels_(nlevels)
    {
        check(orbhip_extractor_create(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, device, &h_),
              "orbhip_extractor_create");
    }
    ~ORBextractor() { orbhip_extractor_destroy(h_); }
    ORBextractor(const ORBextractor &) = delete;
    ORBextractor &operator=(const ORBextractor &) = delete;

    // void operator()(InputArray image, InputArray mask, vector<KeyPoint>& keypoints, OutputArray descriptors)
    // `mask` is ignored, as in the reference.  descriptors: keypoints.size() x 32 bytes, row-major.
    void operator()(const ImageView &image, const ImageView * /*mask*/, std::vector<KeyPoint> &keypoints,
                    std::vector<uint8_t> &descriptors)
    {
        keypoints.clear();
        descriptors.clear();
        if (image.empty()) return;   // src/ORBextractor.cc:1046-1047
        int cap = 0;
        check(orbhip_extractor_capacity(h_, image.rows, image.cols, &cap), "orbhip_extractor_capacity");
        keypoints.resize(cap);
        descriptors.resize((size_t)cap * 32);
        int n = 0;
        check(orbhip_extract(h_, image.data, image.rows, image.cols, (int)image.step, keypoints.data(),
                             descriptors.data(), cap, &n), "orbhip_extract");
        keypoints.resize(n);
        descriptors.resize((size_t)n * 32);   // n == 0: descriptors.release() (:1064-1065)
    }

    // A batch of same-size frames in one call (one ORBextractor::operator() per frame in the reference): frame b starts at
    // images + b * frame_stride.  keypoints[b] / descriptors[b] as operator() fills them.
    void ExtractBatch(const uint8_t *images, int batch, int rows, int cols, size_t step, size_t frame_stride,
                      std::vector<std::vector<KeyPoint> > &keypoints, std::vector<std::vector<uint8_t> > &descriptors)
    {
        keypoints.assign(batch, std::vector<KeyPoint>());
        descriptors.assign(batch, std::vector<uint8_t>());
        if (!images || batch <= 0 || rows <= 0 || cols <= 0) return;
        int cap = 0;
        check(orbhip_extractor_capacity(h_, rows, cols, &cap), "orbhip_extractor_capacity");
        std::vector<KeyPoint> k((size_t)batch * cap);
        std::vector<uint8_t> d((size_t)batch * cap * 32);
        std::vector<int32_t> n(batch, 0);
        check(orbhip_extract_batch(h_, images, batch, rows, cols, (int)step, frame_stride, k.data(), d.data(), cap, n.data()),
              "orbhip_extract_batch");
        for (int b = 0; b < batch; ++b) {
            keypoints[b].assign(k.begin() + (size_t)b * cap, k.begin() + (size_t)b * cap + n[b]);
            descriptors[b].assign(d.begin() + (size_t)b * cap * 32, d.begin() + ((size_t)b * cap + n[b]) * 32);
        }
    }

    // mvImagePyramid[0] on demand: a monocular Tracking thread never reads it (only Frame::ComputeStereoMatches does)
    void SetLazyLevel0(bool on) { check(orbhip_extractor_set_lazy_level0(h_, on ? 1 : 0), "orbhip_extractor_set_lazy_level0"); }

    // several extractors on several streams: chain a stage (0 pyramid, 1 FAST, 2 octree, 3 descriptors) behind another
    // handle's through hipEvent_t handles (scheduling only, see orbhip_extractor_set_stage_gate)
    void SetStageGate(int stage, void *waitEvent, void *recordEvent)
    {
        check(orbhip_extractor_set_stage_gate(h_, stage, waitEvent, recordEvent), "orbhip_extractor_set_stage_gate");
    }

    int GetLevels() { return nlevels_; }
    float GetScaleFactor() { return tab(0).size() > 1 ? tab(0)[1] : 1.f; }
    std::vector<float> GetScaleFactors() { return tab(0); }
    std::vector<float> GetInverseScaleFactors() { return tab(1); }
    std::vector<float> GetScaleSigmaSquares() { return tab(2); }
    std::vector<float> GetInverseScaleSigmaSquares() { return tab(3); }

    // mvImagePyramid[level] of the last call: host copy of the level (rows x cols, tightly packed)
    std::vector<uint8_t> ImagePyramidLevel(int level, int &rows, int &cols)
    {
        int stride = 0;
        const void *d = nullptr;
        check(orbhip_pyramid_level(h_, 0, level, &rows, &cols, &stride, &d), "orbhip_pyramid_level");
        std::vector<uint8_t> out((size_t)rows * cols);
        check(orbhip_pyramid_level_download(h_, 0, level, 0, out.data(), cols), "orbhip_pyramid_level_download");
        return out;
    }
    orbhip_extractor *handle() { return h_; }

private:
    std::vector<float> tab(int which)
    {
        std::vector<float> t[4];
        for (auto &v : t) v.resize(nlevels_);
        check(orbhip_extractor_tables(h_, t[0].data(), t[1].data(), t[2].data(), t[3].data(), nullptr),
              "orbhip_extractor_tables");
        return t[which];
    }
    orbhip_extractor *h_;
    int nlevels_;
};

class ORBmatcher {
public:
    static const int TH_LOW = 50, TH_HIGH = 100, HISTO_LENGTH = 30;

    ORBmatcher(float nnratio = 0.6f, bool checkOri = true, int device = 0)
        : m_(nullptr), mfNNratio(nnratio), mbCheckOrientation(checkOri)
    {
        check(orbhip_matcher_create(device, &m_), "orbhip_matcher_create");
    }
    ~ORBmatcher() { orbhip_matcher_destroy(m_); }
    ORBmatcher(const ORBmatcher &) = delete;
    ORBmatcher &operator=(const ORBmatcher &) = delete;

    // static int DescriptorDistance(const cv::Mat &a, const cv::Mat &b): host popcount, same result
    static int DescriptorDistance(const uint8_t *a, const uint8_t *b)
    {
        int d = 0;
        for (int i = 0; i < 32; ++i) d += __builtin_popcount((unsigned)(a[i] ^ b[i]));
        return d;
    }

    // int SearchForInitialization(Frame &F1, Frame &F2, vector<Point2f> &vbPrevMatched, vector<int> &vnMatches12, int windowSize)
    int SearchForInitialization(const orbhip_frame_view &F1, const orbhip_frame_view &F2, std::vector<float> &vbPrevMatchedXY,
                                std::vector<int> &vnMatches12, int windowSize = 10)
    {
        vnMatches12.assign(F1.n > 0 ? F1.n : 1, -1);
        int n = 0;
        check(orbhip_search_for_initialization(m_, &F1, &F2, vbPrevMatchedXY.data(), vnMatches12.data(), windowSize,
                                               mfNNratio, mbCheckOrientation, &n), "orbhip_search_for_initialization");
        vnMatches12.resize(F1.n);
        return n;
    }

    // int SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, float th, bool bMono) after projection
    int SearchByProjection(const orbhip_frame_view &CurrentFrame, const std::vector<orbhip_query> &q,
                           const uint8_t *qdesc, const uint8_t *taken, std::vector<int> &assign)
    {
        assign.assign(CurrentFrame.n > 0 ? CurrentFrame.n : 1, -1);
        int n = 0;
        check(orbhip_search_by_projection_frame(m_, &CurrentFrame, q.data(), qdesc, (int)q.size(), taken, assign.data(),
                                                mbCheckOrientation, &n), "orbhip_search_by_projection_frame");
        assign.resize(CurrentFrame.n);
        return n;
    }

    // int SearchByProjection(Frame &F, const vector<MapPoint*> &vpMapPoints, float th) after isInFrustum
    int SearchByProjectionPoints(const orbhip_frame_view &F, const std::vector<orbhip_query> &q, const uint8_t *qdesc,
                                 const uint8_t *taken, std::vector<int> &assign)
    {
        assign.assign(F.n > 0 ? F.n : 1, -1);
        int n = 0;
        check(orbhip_search_by_projection_points(m_, &F, q.data(), qdesc, (int)q.size(), taken, assign.data(), mfNNratio,
                                                 &n), "orbhip_search_by_projection_points");
        assign.resize(F.n);
        return n;
    }

    // int SearchByBoW(KeyFrame *pKF, Frame &F, vector<MapPoint*> &vpMapPointMatches)   (max_dist = TH_LOW)
    // int SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, vector<MapPoint*> &vpMatches12) (max_dist = TH_LOW - 1)
    // node1/node2: FeatureVector node id per keypoint; matches12[i1] = i2 or -1
    int SearchByBoW(const orbhip_frame_view &F1, const uint32_t *node1, const uint8_t *valid1, const orbhip_frame_view &F2,
                    const uint32_t *node2, const uint8_t *blocked2, std::vector<int> &matches12, int max_dist = TH_LOW)
    {
        matches12.assign(F1.n > 0 ? F1.n : 1, -1);
        int n = 0;
        check(orbhip_search_by_bow(m_, &F1, node1, valid1, &F2, node2, blocked2, max_dist, mfNNratio, mbCheckOrientation,
                                   matches12.data(), &n), "orbhip_search_by_bow");
        matches12.resize(F1.n);
        return n;
    }

    // int SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, cv::Mat F12, vector<pair<size_t,size_t>> &vMatchedPairs,
    //                            bool bOnlyStereo)
    int SearchForTriangulation(const orbhip_frame_view &F1, const uint32_t *node1, const uint8_t *valid1,
                               const orbhip_frame_view &F2, const uint32_t *node2, const uint8_t *valid2, const float F12[9],
                               float ex, float ey, const float *level_sigma2,
                               std::vector<std::pair<size_t, size_t> > &vMatchedPairs, bool bOnlyStereo)
    {
        std::vector<int> m12(F1.n > 0 ? F1.n : 1, -1);
        int n = 0;
        check(orbhip_search_for_triangulation(m_, &F1, node1, valid1, &F2, node2, valid2, F12, ex, ey, level_sigma2,
                                              bOnlyStereo, mbCheckOrientation, m12.data(), &n),
              "orbhip_search_for_triangulation");
        vMatchedPairs.clear();
        vMatchedPairs.reserve(n);
        for (int i = 0; i < F1.n; ++i)
            if (m12[i] >= 0) vMatchedPairs.push_back(std::make_pair((size_t)i, (size_t)m12[i]));
        return n;
    }

    // ---- projection prologues (the cv::Mat arithmetic in front of the searches, on the device) ---------------------
    // SearchByProjection(CurrentFrame, LastFrame, th, bMono), src/ORBmatcher.cc:1339-1390: Tcw / Tlw = top three rows of
    // the two mTcw (12 floats, row-major); world[n][3], flags[n] (ORBHIP_POINT_*) per LastFrame.mvpMapPoints[i]
    std::vector<orbhip_query> ProjectLastFrame(const orbhip_camera &cam, const float *Tcw, const float *Tlw, int n,
                                               const float *world, const uint8_t *flags, const KeyPoint *lastKeysUn, float th,
                                               bool bMono)
    {
        std::vector<orbhip_query> q(n > 0 ? n : 1);
        check(orbhip_project_last_frame(m_, &cam, Tcw, Tlw, n, world, flags, lastKeysUn, th, bMono ? 1 : 0, q.data()),
              "orbhip_project_last_frame");
        q.resize(n);
        return q;
    }

    // Frame::isInFrustum + MapPoint::PredictScale for the local map (src/Frame.cc:269-325, src/MapPoint.cc:400-418) and
    // the window of SearchByProjection(F, vpMapPoints, th) (src/ORBmatcher.cc:52-69)
    std::vector<orbhip_query> FrustumQueries(const orbhip_camera &cam, const float *Tcw, int n, const float *world,
                                             const float *normal, const float *maxDist, const float *minDist,
                                             const uint8_t *flags, float viewingCosLimit, float th,
                                             std::vector<float> *viewCos = nullptr)
    {
        std::vector<orbhip_query> q(n > 0 ? n : 1);
        if (viewCos) viewCos->assign(n > 0 ? n : 1, 0.f);
        check(orbhip_frustum_queries(m_, &cam, Tcw, n, world, normal, maxDist, minDist, flags, viewingCosLimit, th, q.data(),
                                     viewCos ? viewCos->data() : nullptr), "orbhip_frustum_queries");
        q.resize(n);
        if (viewCos) viewCos->resize(n);
        return q;
    }

    // prologue of Fuse x2 (mode 0) and of one direction of SearchBySim3 (mode 1), src/ORBmatcher.cc:853-888, 1005-1048,
    // 1155-1180, 1235-1260
    std::vector<orbhip_query> KeyFrameQueries(const orbhip_camera &cam, int mode, bool doubleInvz, const float *T1,
                                              const float *T2, int n, const float *world, const float *normal,
                                              const float *maxDist, const float *minDist, const uint8_t *flags, float th)
    {
        std::vector<orbhip_query> q(n > 0 ? n : 1);
        check(orbhip_keyframe_queries(m_, &cam, mode, doubleInvz ? 1 : 0, T1, T2, n, world, normal, maxDist, minDist, flags, th,
                                      q.data()), "orbhip_keyframe_queries");
        q.resize(n);
        return q;
    }

    // int SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const set<MapPoint*> &sAlreadyFound, float th, int ORBdist)
    // (relocalisation, src/ORBmatcher.cc:1472-1599) after projection; taken[n] = CurrentFrame.mvpMapPoints[i] != NULL
    int SearchByProjectionKeyFrame(const orbhip_frame_view &CurrentFrame, const std::vector<orbhip_query> &q,
                                   const uint8_t *qdesc, const uint8_t *taken, std::vector<int> &assign, int ORBdist)
    {
        assign.assign(CurrentFrame.n > 0 ? CurrentFrame.n : 1, -1);
        int n = 0;
        check(orbhip_search_by_projection_keyframe(m_, &CurrentFrame, q.data(), qdesc, (int)q.size(), taken, assign.data(),
                                                   ORBdist, mbCheckOrientation, &n), "orbhip_search_by_projection_keyframe");
        assign.resize(CurrentFrame.n);
        return n;
    }

    // int SearchByProjection(KeyFrame *pKF, cv::Mat Scw, const vector<MapPoint*> &vpPoints, vector<MapPoint*> &vpMatched,
    // int th) (loop closing, src/ORBmatcher.cc:290-403) after the Sim3 projection; matched[n] = vpMatched[idx] != NULL
    int SearchByProjectionSim3(const orbhip_frame_view &KF, const std::vector<orbhip_query> &q, const uint8_t *qdesc,
                               const uint8_t *matched, std::vector<int> &assign)
    {
        assign.assign(KF.n > 0 ? KF.n : 1, -1);
        int n = 0;
        check(orbhip_search_by_projection_sim3(m_, &KF, q.data(), qdesc, (int)q.size(), matched, assign.data(), &n),
              "orbhip_search_by_projection_sim3");
        assign.resize(KF.n);
        return n;
    }

    // int Fuse(KeyFrame *pKF, const vector<MapPoint*> &vpMapPoints, float th)            (sim3Form = false, :825-975)
    // int Fuse(KeyFrame *pKF, cv::Mat Scw, const vector<MapPoint*> &vpPoints, float th, ...) (sim3Form = true, :977-1100)
    // up to the replace-or-add decision, which touches the map graph and stays with the caller: bestIdx[i] / bestDist[i]
    // = the key point the i-th map point would fuse with (-1 / 256 when nothing passes the gates)
    void Fuse(const orbhip_frame_view &KF, const orbhip_camera &cam, const float *Tcw, bool sim3Form, int n, const float *world,
              const float *normal, const float *maxDist, const float *minDist, const uint8_t *flags, const uint8_t *pointDesc,
              float th, const float *invLevelSigma2, std::vector<int> &bestIdx, std::vector<int> &bestDist)
    {
        bestIdx.assign(n > 0 ? n : 1, -1);
        bestDist.assign(n > 0 ? n : 1, 256);
        check(orbhip_fuse(m_, &KF, &cam, Tcw, sim3Form ? 1 : 0, n, world, normal, maxDist, minDist, flags, pointDesc, th,
                          invLevelSigma2, bestIdx.data(), bestDist.data()), "orbhip_fuse");
        bestIdx.resize(n);
        bestDist.resize(n);
    }

    // int SearchBySim3(KeyFrame *pKF1, KeyFrame *pKF2, vector<MapPoint*> &vpMatches12, const float &s12, const cv::Mat &R12,
    // const cv::Mat &t12, const float th) (src/ORBmatcher.cc:1102-1326), complete.  T1w / T2w: [R | t] of the two key
    // frames; S21 = [sR21 | t21], S12 = [sR12 | t12] as the reference computes them (:1119-1121).  Per key-frame slot: the
    // slot's map point (world, mfMax/MinDistance, descriptor) and ORBHIP_POINT_PRESENT iff it takes part (:1132-1143).
    int SearchBySim3(const orbhip_frame_view &KF1, const orbhip_frame_view &KF2, const orbhip_camera &cam, const float *T1w,
                     const float *T2w, const float *S21, const float *S12, const float *world1, const float *maxDist1,
                     const float *minDist1, const uint8_t *flags1, const uint8_t *desc1, const float *world2,
                     const float *maxDist2, const float *minDist2, const uint8_t *flags2, const uint8_t *desc2, float th,
                     std::vector<int> &matches12)
    {
        matches12.assign(KF1.n > 0 ? KF1.n : 1, -1);
        int n = 0;
        check(orbhip_search_by_sim3(m_, &KF1, &KF2, &cam, T1w, T2w, S21, S12, world1, maxDist1, minDist1, flags1, desc1, world2,
                                    maxDist2, minDist2, flags2, desc2, th, matches12.data(), &n), "orbhip_search_by_sim3");
        matches12.resize(KF1.n);
        return n;
    }

    // the search loop of Fuse / SearchBySim3 on caller-made queries (src/ORBmatcher.cc:893-950, 1199-1219)
    void SearchBestInWindow(const orbhip_frame_view &KF, const std::vector<orbhip_query> &q, const uint8_t *qdesc, bool chi2Gate,
                            const float *invLevelSigma2, std::vector<int> &bestIdx, std::vector<int> &bestDist)
    {
        bestIdx.assign(q.size() ? q.size() : 1, -1);
        bestDist.assign(q.size() ? q.size() : 1, 256);
        check(orbhip_search_best_in_window(m_, &KF, q.data(), qdesc, (int)q.size(), chi2Gate ? 1 : 0, invLevelSigma2, bestIdx.data(),
                                           bestDist.data()), "orbhip_search_best_in_window");
        bestIdx.resize(q.size());
        bestDist.resize(q.size());
    }

    // MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:242-307), batched over map points
    std::vector<int> DistinctiveDescriptors(const uint8_t *desc, const std::vector<int32_t> &offsets)
    {
        const int np = (int)offsets.size() - 1;
        std::vector<int> best(np > 0 ? np : 1, -1);
        if (np > 0) check(orbhip_distinctive_descriptors(m_, desc, offsets.data(), np, best.data()), "orbhip_distinctive_descriptors");
        best.resize(np > 0 ? np : 0);
        return best;
    }

    // Frame glue either side of the searches: UndistortKeyPoints, AssignFeaturesToGrid, ComputeStereoFromRGBD
    // (src/Frame.cc:404-434, 230-245, 643-664)
    std::vector<KeyPoint> UndistortKeyPoints(const std::vector<KeyPoint> &keys, float fx, float fy, float cx, float cy,
                                             const float dist5[5])
    {
        std::vector<KeyPoint> un(keys.size() ? keys.size() : 1);
        check(orbhip_undistort_keypoints(m_, keys.data(), (int)keys.size(), fx, fy, cx, cy, dist5, un.data()),
              "orbhip_undistort_keypoints");
        un.resize(keys.size());
        return un;
    }
    void ComputeStereoFromRGBD(const std::vector<KeyPoint> &keys, const std::vector<KeyPoint> &keysUn, const float *depth, int rows,
                               int cols, int strideFloats, float mbf, std::vector<float> &mvuRight, std::vector<float> &mvDepth)
    {
        mvuRight.assign(keys.size() ? keys.size() : 1, -1.f);
        mvDepth.assign(keys.size() ? keys.size() : 1, -1.f);
        check(orbhip_compute_stereo_from_rgbd(m_, keys.data(), keysUn.data(), (int)keys.size(), depth, rows, cols, strideFloats, mbf,
                                              mvuRight.data(), mvDepth.data()), "orbhip_compute_stereo_from_rgbd");
        mvuRight.resize(keys.size());
        mvDepth.resize(keys.size());
    }

    // the Frame statics the prologues read (src/Frame.cc:97-112), as one record
    static orbhip_camera MakeCamera(float fx, float fy, float cx, float cy, float mbf, float mb, float minX, float maxX, float minY,
                                    float maxY, const std::vector<float> &scaleFactors, float logScaleFactor)
    {
        orbhip_camera c = {};
        c.fx = fx; c.fy = fy; c.cx = cx; c.cy = cy; c.mbf = mbf; c.mb = mb;
        c.min_x = minX; c.max_x = maxX; c.min_y = minY; c.max_y = maxY;
        c.n_levels = (int32_t)scaleFactors.size();
        c.log_scale_factor = logScaleFactor;
        for (size_t i = 0; i < scaleFactors.size() && i < (size_t)ORBHIP_MAX_LEVELS; ++i) c.scale_factors[i] = scaleFactors[i];
        return c;
    }
    // a Frame / KeyFrame as the searches see it (mvKeysUn, mDescriptors, mvuRight, image bounds, 64 x 48 grid scale)
    static orbhip_frame_view MakeFrameView(const std::vector<KeyPoint> &keysUn, const std::vector<uint8_t> &desc, const float *uRight,
                                           float minX, float minY, float maxX, float maxY, const std::vector<float> &scaleFactors)
    {
        orbhip_frame_view v = {};
        v.n = (int32_t)keysUn.size(); v.keys = keysUn.data(); v.desc = desc.data(); v.u_right = uRight;
        v.min_x = minX; v.min_y = minY; v.max_x = maxX; v.max_y = maxY;
        v.grid_inv_w = 64.f / (maxX - minX); v.grid_inv_h = 48.f / (maxY - minY);   // src/Frame.cc:101-102
        v.n_levels = (int32_t)scaleFactors.size(); v.scale_factors = scaleFactors.data();
        return v;
    }

    // void Frame::ComputeStereoMatches(): pyramids come from the two extractor objects
    int ComputeStereoMatches(ORBextractor &left, ORBextractor &right, const std::vector<KeyPoint> &keysL,
                             const std::vector<uint8_t> &descL, const std::vector<KeyPoint> &keysR,
                             const std::vector<uint8_t> &descR, float mbf, float mb, std::vector<float> &mvuRight,
                             std::vector<float> &mvDepth)
    {
        mvuRight.assign(keysL.size() ? keysL.size() : 1, -1.f);
        mvDepth.assign(keysL.size() ? keysL.size() : 1, -1.f);
        int n = 0;
        check(orbhip_compute_stereo_matches(m_, left.handle(), 0, right.handle(), 0, keysL.data(), descL.data(),
                                            (int)keysL.size(), keysR.data(), descR.data(), (int)keysR.size(), mbf, mb,
                                            mvuRight.data(), mvDepth.data(), &n), "orbhip_compute_stereo_matches");
        mvuRight.resize(keysL.size());
        mvDepth.resize(keysL.size());
        return n;
    }

protected:
    orbhip_matcher *m_;
    float mfNNratio;
    bool mbCheckOrientation;
};

}  // namespace orbhip
