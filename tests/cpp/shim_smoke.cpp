// Exercises the C++ mirror (include/orbhip/ORBextractor.hpp) exactly like Frame::ExtractORB
// (src/Frame.cc:247-253) calls the reference class.  Reads a raw uint8 image, writes
// "n\n" + keypoints + descriptors as binary to stdout's file argument.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "orbhip/ORBextractor.hpp"

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
        orbhip::ImageView empty{nullptr, 0, 0, 0};
        ext(empty, nullptr, kps, desc);
        if (!kps.empty() || !desc.empty()) return 4;
    } catch (const orbhip::Error &e) {
        std::fprintf(stderr, "orbhip error %d: %s\n", e.code, e.what());
        return 5;
    }
    return 0;
}
