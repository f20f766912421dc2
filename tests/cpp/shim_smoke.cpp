// Exercises the C++ mirror (include/orbhip/ORBextractor.hpp) exactly like Frame::ExtractORB
// (src/Frame.cc:247-253) calls the reference class.  Reads a raw uint8 image, writes
// "n\n" + keypoints + descriptors as binary to stdout's file argument.  With two more arguments (vocabulary text
// file, output file) it also runs Frame::ComputeBoW (src/Frame.cc:395-402) through orbhip::ORBVocabulary and
// SearchByBoW of the frame against itself.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "orbhip/ORBVocabulary.hpp"

int main(int argc, char **argv)
{
    if (argc < 6) { std::fprintf(stderr, "usage: shim_smoke in.raw rows cols nfeatures out.bin\n"); return 2; }
    const int rows = std::atoi(argv[2]), cols = std::atoi(argv[3]), nf = std::atoi(argv[4]);
    std::vector<uint8_t> img((size_t)rows * cols);
    FILE *f = std::fopen(argv[1], "rb");
    if (!f || std::fread(img.data(), 1, img.size(), f) != img.size()) return 3;
    std::fclose(f);
    try {
        orbhip::ORBextractor ext(nf, 1.2f, 8, 20, 7);
        std::vector<orbhip::KeyPoint> kps;
        std::vector<uint8_t> desc;
        orbhip::ImageView view{img.data(), rows, cols, (size_t)cols};
        ext(view, nullptr, kps, desc);
        FILE *o = std::fopen(argv[5], "wb");
        int n = (int)kps.size();
        std::fwrite(&n, 4, 1, o);
        std::fwrite(kps.data(), sizeof(orbhip::KeyPoint), kps.size(), o);
        std::fwrite(desc.data(), 1, desc.size(), o);
        std::fclose(o);
        std::printf("levels %d scale %.9g keypoints %d\n", ext.GetLevels(), ext.GetScaleFactor(), n);
        if (argc >= 8) {
            orbhip::ORBVocabulary voc;
            if (voc.loadFromTextFile("/nonexistent/voc.txt") || !voc.empty()) return 6;
            if (!voc.loadFromTextFile(argv[6]) || voc.empty()) return 7;
            orbhip::BowVector bow;
            orbhip::FeatureVector fv;
            std::vector<uint32_t> node;
            voc.transform(desc.data(), n, bow, fv, 4, &node);
            orbhip_frame_view view2 = {};
            view2.n = n; view2.keys = kps.data(); view2.desc = desc.data();
            orbhip::ORBmatcher matcher(0.7f, true);
            std::vector<int> m12;
            const int nm = matcher.SearchByBoW(view2, node.data(), nullptr, view2, node.data(), nullptr, m12);
            FILE *o2 = std::fopen(argv[7], "wb");
            int nb = (int)bow.size(), nfv = 0;
            for (orbhip::FeatureVector::const_iterator it = fv.begin(); it != fv.end(); ++it) nfv += (int)it->second.size();
            std::fwrite(&nb, 4, 1, o2);
            for (orbhip::BowVector::const_iterator it = bow.begin(); it != bow.end(); ++it) {
                std::fwrite(&it->first, 4, 1, o2);
                std::fwrite(&it->second, 8, 1, o2);
            }
            std::fwrite(node.data(), 4, node.size(), o2);
            std::fwrite(&nm, 4, 1, o2);
            std::fwrite(m12.data(), 4, m12.size(), o2);
            std::fclose(o2);
            std::printf("words %u bow %d feature-vector entries %d self matches %d\n", voc.size(), nb, nfv, nm);
        }
        orbhip::ImageView empty{nullptr, 0, 0, 0};
        ext(empty, nullptr, kps, desc);
        if (!kps.empty() || !desc.empty()) return 4;
    } catch (const orbhip::Error &e) {
        std::fprintf(stderr, "orbhip error %d: %s\n", e.code, e.what());
        return 5;
    }
    return 0;
}
