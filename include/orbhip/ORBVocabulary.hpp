// C++ host-side mirror of ORB_SLAM2::ORBVocabulary (include/ORBVocabulary.h:32-33, a typedef of
// DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>) above the C ABI of orbhip.h, for the calls the front-end
// makes: loadFromTextFile (src/System.cc:68-76) and transform(features, BowVector, FeatureVector, levelsup)
// (src/Frame.cc:395-402, src/KeyFrame.cc ComputeBoW).  Header-only, no OpenCV / DBoW2 dependency: BowVector and
// FeatureVector are the same std::map types DBoW2 derives from (BowVector.h:56-57, FeatureVector.h:20-21).
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

#include "ORBextractor.hpp"

namespace orbhip {

typedef std::map<unsigned int, double> BowVector;                        // WordId -> WordValue
typedef std::map<unsigned int, std::vector<unsigned int> > FeatureVector;   // NodeId -> feature indices

class ORBVocabulary {
public:
    explicit ORBVocabulary(int device = 0) : v_(nullptr), device_(device) {}
    ~ORBVocabulary() { orbhip_vocabulary_destroy(v_); }
    ORBVocabulary(const ORBVocabulary &) = delete;
    ORBVocabulary &operator=(const ORBVocabulary &) = delete;

    // bool loadFromTextFile(const std::string &filename): false on a missing or malformed file
    bool loadFromTextFile(const std::string &filename)
    {
        orbhip_vocabulary *nv = nullptr;
        if (orbhip_vocabulary_load_text(filename.c_str(), device_, &nv) != ORBHIP_OK) return false;
        orbhip_vocabulary_destroy(v_);
        v_ = nv;
        return true;
    }
    bool empty() const { return size() == 0; }
    unsigned int size() const
    {
        int words = 0;
        if (v_) orbhip_vocabulary_info(v_, nullptr, nullptr, nullptr, nullptr, nullptr, &words);
        return (unsigned int)words;
    }

    // void transform(const std::vector<TDescriptor> &features, BowVector &v, FeatureVector &fv, int levelsup) const
    // descriptors: n x 32 bytes (cv::Mat mDescriptors is already in this layout).  node_ids (optional) receives the
    // flat per-feature node ids that orbhip_search_by_bow takes.
    void transform(const uint8_t *descriptors, int n, BowVector &v, FeatureVector &fv, int levelsup,
                   std::vector<uint32_t> *node_ids = nullptr) const
    {
        v.clear();
        fv.clear();
        if (node_ids) node_ids->assign(n > 0 ? n : 0, ORBHIP_NO_NODE);
        if (!v_ || n <= 0) return;
        std::vector<uint32_t> node(n), ids(n);
        std::vector<double> vals(n);
        int nb = 0;
        check(orbhip_vocabulary_transform(v_, descriptors, n, levelsup, nullptr, nullptr, node.data(), ids.data(),
                                          vals.data(), &nb), "orbhip_vocabulary_transform");
        for (int i = 0; i < nb; ++i) v.insert(v.end(), std::make_pair(ids[i], vals[i]));
        for (int i = 0; i < n; ++i)
            if (node[i] != ORBHIP_NO_NODE) fv[node[i]].push_back((unsigned int)i);
        if (node_ids) node_ids->swap(node);
    }
    orbhip_vocabulary *handle() { return v_; }

private:
    orbhip_vocabulary *v_;
    int device_;
};

}  // namespace orbhip
