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
