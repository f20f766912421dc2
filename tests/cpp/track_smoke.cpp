// The Tracking-side calls of the hot path through the C++ mirror (include/orbhip/ORBextractor.hpp), built with g++ against
// liborbhip.so: what Tracking::TrackWithMotionModel (src/Tracking.cc:867-928), the stereo Frame constructor
// (src/Frame.cc:61-117) and LoopClosing::ComputeSim3 (src/LoopClosing.cc) ask of ORBextractor / ORBmatcher.
//   track_smoke in.blob out.blob
// in.blob: records {int32 bytes, payload} written by tests/test_cpp_shim_gpu.py (images, camera, poses, map points);
// out.blob: keypoints / descriptors of the three frames, the projected queries, SearchByProjection's assignment,
// mvuRight / mvDepth, SearchBySim3's matches -- compared with the oracle by the pytest wrapper.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "orbhip/ORBextractor.hpp"

static std::vector<std::vector<uint8_t> > read_blob(const char *path)
{
    std::vector<std::vector<uint8_t> > rec;
    FILE *f = std::fopen(path, "rb");
    if (!f) return rec;
    int32_t nb;
    while (std::fread(&nb, 4, 1, f) == 1) {
        std::vector<uint8_t> r((size_t)nb);
        if (nb && std::fread(r.data(), 1, (size_t)nb, f) != (size_t)nb) { rec.clear(); break; }
        rec.push_back(r);
    }
    std::fclose(f);
    return rec;
}
static void put(FILE *f, const void *p, size_t bytes)
{
    const int32_t nb = (int32_t)bytes;
    std::fwrite(&nb, 4, 1, f);
    if (bytes) std::fwrite(p, 1, bytes, f);
}
template <class T> static const T *as(const std::vector<uint8_t> &r) { return reinterpret_cast<const T *>(r.data()); }

int main(int argc, char **argv)
{
    if (argc < 3) { std::fprintf(stderr, "usage: track_smoke in.blob out.blob\n"); return 2; }
    const std::vector<std::vector<uint8_t> > in = read_blob(argv[1]);
    if (in.size() != 24) { std::fprintf(stderr, "bad input blob (%zu records)\n", in.size()); return 3; }
    const int rows = as<int32_t>(in[0])[0], cols = as<int32_t>(in[0])[1], nf = as<int32_t>(in[0])[2];
    try {
        orbhip::ORBextractor extL(nf, 1.2f, 8, 20, 7), extR(nf, 1.2f, 8, 20, 7), extC(nf, 1.2f, 8, 20, 7);
        extC.SetLazyLevel0(true);    // the current frame is only tracked: nobody reads its mvImagePyramid[0]
        std::vector<orbhip::KeyPoint> kL, kR, kC;
        std::vector<uint8_t> dL, dR, dC;
        const orbhip::ImageView vL{in[1].data(), rows, cols, (size_t)cols}, vR{in[2].data(), rows, cols, (size_t)cols},
                                vC{in[3].data(), rows, cols, (size_t)cols};
        extL(vL, nullptr, kL, dL);
        extR(vR, nullptr, kR, dR);
        extC(vC, nullptr, kC, dC);
        const std::vector<float> sf = extL.GetScaleFactors();
        orbhip_camera cam;
        std::memcpy(&cam, in[4].data(), sizeof(cam));
        FILE *o = std::fopen(argv[2], "wb");
        put(o, kL.data(), kL.size() * sizeof(orbhip::KeyPoint)); put(o, dL.data(), dL.size());
        put(o, kR.data(), kR.size() * sizeof(orbhip::KeyPoint)); put(o, dR.data(), dR.size());
        put(o, kC.data(), kC.size() * sizeof(orbhip::KeyPoint)); put(o, dC.data(), dC.size());

        // ---- TrackWithMotionModel: SearchByProjection(mCurrentFrame, mLastFrame, th, bMono) ----
        orbhip::ORBmatcher matcher(0.9f, true);
        const int nL = (int)kL.size();
        if (in[7].size() != (size_t)nL * 12 || in[8].size() != (size_t)nL) { std::fprintf(stderr, "map arrays do not fit %d keypoints\n", nL); return 4; }
        const float th = as<float>(in[9])[0];
        const bool mono = as<int32_t>(in[9])[1] != 0;
        std::vector<orbhip_query> q = matcher.ProjectLastFrame(cam, as<float>(in[5]), as<float>(in[6]), nL, as<float>(in[7]), in[8].data(),
                                                               kL.data(), th, mono);
        const orbhip_frame_view fC = orbhip::ORBmatcher::MakeFrameView(kC, dC, nullptr, cam.min_x, cam.min_y, cam.max_x, cam.max_y, sf);
        std::vector<int> assign;
        const int nproj = matcher.SearchByProjection(fC, q, dL.data(), nullptr, assign);
        put(o, q.data(), q.size() * sizeof(orbhip_query));
        put(o, assign.data(), assign.size() * 4);
        put(o, &nproj, 4);

        // ---- stereo Frame constructor: ComputeStereoMatches on the two extractors' pyramids ----
        std::vector<float> uR, depth;
        const int nst = matcher.ComputeStereoMatches(extL, extR, kL, dL, kR, dR, as<float>(in[23])[0], as<float>(in[23])[1], uR, depth);
        put(o, uR.data(), uR.size() * 4);
        put(o, depth.data(), depth.size() * 4);
        put(o, &nst, 4);

        // ---- LoopClosing::ComputeSim3: SearchBySim3 between the left frame and the current frame as key frames ----
        const orbhip_frame_view f1 = orbhip::ORBmatcher::MakeFrameView(kL, dL, nullptr, cam.min_x, cam.min_y, cam.max_x, cam.max_y, sf);
        orbhip::ORBmatcher m75(0.75f, true);
        std::vector<int> m12;
        const int nsim = m75.SearchBySim3(f1, fC, cam, as<float>(in[10]), as<float>(in[11]), as<float>(in[12]), as<float>(in[13]),
                                          as<float>(in[14]), as<float>(in[15]), as<float>(in[16]), in[17].data(), dL.data(),
                                          as<float>(in[18]), as<float>(in[19]), as<float>(in[20]), in[21].data(), dC.data(),
                                          as<float>(in[22])[0], m12);
        put(o, m12.data(), m12.size() * 4);
        put(o, &nsim, 4);
        // mvImagePyramid[0] of the lazily extracted current frame, produced on demand
        int pr = 0, pc = 0;
        const std::vector<uint8_t> lvl0 = extC.ImagePyramidLevel(0, pr, pc);
        put(o, lvl0.data(), lvl0.size());
        std::fclose(o);
        std::printf("keypoints %zu / %zu / %zu, projection matches %d, stereo matches %d, sim3 matches %d\n", kL.size(), kR.size(),
                    kC.size(), nproj, nst, nsim);
    } catch (const orbhip::Error &e) {
        std::fprintf(stderr, "orbhip error %d: %s\n", e.code, e.what());
        return 5;
    }
    return 0;
}
