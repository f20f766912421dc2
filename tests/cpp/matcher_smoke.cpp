// The remaining ORBmatcher.h methods through the C++ mirror, called from a g++-built program: SearchForInitialization
// (Tracking::MonocularInitialization), SearchByProjection(F, vpMapPoints) behind FrustumQueries (Tracking::SearchLocalPoints),
// SearchByProjection(CurrentFrame, pKF, ...) (relocalisation), SearchByProjection(pKF, Scw, ...) (loop closing), Fuse in both
// forms and SearchForTriangulation (LocalMapping).  Inputs come from a record blob written by tests/test_cpp_shim_gpu.py, the
// outputs go back the same way and are compared with the oracle there.
//   matcher_smoke in.blob out.blob
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "orbhip/ORBextractor.hpp"

typedef std::vector<uint8_t> Rec;
static std::vector<Rec> read_blob(const char *path)
{
    std::vector<Rec> rec;
    FILE *f = std::fopen(path, "rb");
    if (!f) return rec;
    int32_t nb;
    while (std::fread(&nb, 4, 1, f) == 1) {
        Rec r((size_t)nb);
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
template <class T> static const T *as(const Rec &r) { return reinterpret_cast<const T *>(r.data()); }
template <class T> static std::vector<T> vec(const Rec &r) { return std::vector<T>(as<T>(r), as<T>(r) + r.size() / sizeof(T)); }
static const uint8_t *opt(const Rec &r) { return r.empty() ? nullptr : r.data(); }

int main(int argc, char **argv)
{
    if (argc < 3) { std::fprintf(stderr, "usage: matcher_smoke in.blob out.blob\n"); return 2; }
    const std::vector<Rec> in = read_blob(argv[1]);
    if (in.size() != 37) { std::fprintf(stderr, "bad input blob (%zu records)\n", in.size()); return 3; }
    try {
        const std::vector<orbhip::KeyPoint> k1 = vec<orbhip::KeyPoint>(in[0]), k2 = vec<orbhip::KeyPoint>(in[2]);
        const std::vector<uint8_t> d1 = in[1], d2 = in[3];
        const std::vector<float> sf = vec<float>(in[4]);
        const float *b = as<float>(in[5]);
        orbhip_camera cam;
        std::memcpy(&cam, in[6].data(), sizeof(cam));
        const std::vector<float> ur1 = vec<float>(in[7]), ur2 = vec<float>(in[8]);
        const orbhip_frame_view F1 = orbhip::ORBmatcher::MakeFrameView(k1, d1, nullptr, b[0], b[1], b[2], b[3], sf);
        const orbhip_frame_view F2 = orbhip::ORBmatcher::MakeFrameView(k2, d2, nullptr, b[0], b[1], b[2], b[3], sf);
        const orbhip_frame_view F1s = orbhip::ORBmatcher::MakeFrameView(k1, d1, ur1.data(), b[0], b[1], b[2], b[3], sf);
        const orbhip_frame_view F2s = orbhip::ORBmatcher::MakeFrameView(k2, d2, ur2.data(), b[0], b[1], b[2], b[3], sf);
        FILE *o = std::fopen(argv[2], "wb");

        {   // SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize), nnratio 0.9 (src/Tracking.cc:599-600)
            orbhip::ORBmatcher m(0.9f, true);
            std::vector<float> prev = vec<float>(in[9]);
            std::vector<int> m12;
            const int n = m.SearchForInitialization(F1, F2, prev, m12, as<int32_t>(in[10])[0]);
            put(o, m12.data(), m12.size() * 4); put(o, prev.data(), prev.size() * 4); put(o, &n, 4);
        }
        {   // SearchLocalPoints: isInFrustum for the local map, then SearchByProjection(F, vpMapPoints, th), nnratio 0.8
            orbhip::ORBmatcher m(0.8f, true);
            const int np = (int)in[16].size();
            std::vector<float> vc;
            std::vector<orbhip_query> q = m.FrustumQueries(cam, as<float>(in[11]), np, as<float>(in[12]), as<float>(in[13]),
                                                           as<float>(in[14]), as<float>(in[15]), in[16].data(), 0.5f,
                                                           as<float>(in[17])[0], &vc);
            std::vector<int> assign;
            const int n = m.SearchByProjectionPoints(F2, q, in[18].data(), opt(in[19]), assign);
            put(o, q.data(), q.size() * sizeof(orbhip_query)); put(o, vc.data(), vc.size() * 4);
            put(o, assign.data(), assign.size() * 4); put(o, &n, 4);
        }
        {   // relocalisation: SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist); loop closing: (pKF, Scw, ...)
            orbhip::ORBmatcher m(0.9f, true);
            const std::vector<orbhip_query> q = vec<orbhip_query>(in[20]), q3 = vec<orbhip_query>(in[23]);
            std::vector<int> a, a3;
            const int n = m.SearchByProjectionKeyFrame(F2, q, in[21].data(), opt(in[22]), a, as<int32_t>(in[24])[0]);
            const int n3 = m.SearchByProjectionSim3(F2, q3, in[21].data(), opt(in[22]), a3);
            put(o, a.data(), a.size() * 4); put(o, &n, 4); put(o, a3.data(), a3.size() * 4); put(o, &n3, 4);
        }
        for (int form = 0; form < 2; ++form) {   // Fuse(pKF, vpMapPoints, th) and Fuse(pKF, Scw, vpPoints, th, ...) up to the decision
            orbhip::ORBmatcher m(0.6f, true);
            const int np = (int)in[30].size();
            std::vector<int> bi, bd;
            m.Fuse(F2s, cam, as<float>(in[25]), form != 0, np, as<float>(in[26]), as<float>(in[27]), as<float>(in[28]),
                   as<float>(in[29]), in[30].data(), in[31].data(), as<float>(in[32])[form], as<float>(in[33]), bi, bd);
            put(o, bi.data(), bi.size() * 4); put(o, bd.data(), bd.size() * 4);
        }
        {   // SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo), nnratio 0.6
            orbhip::ORBmatcher m(0.6f, true);
            std::vector<std::pair<size_t, size_t> > pairs;
            const float *misc = as<float>(in[36]);       // F12[9], ex, ey, then level sigma2
            const int n = m.SearchForTriangulation(F1s, as<uint32_t>(in[34]), nullptr, F2s, as<uint32_t>(in[35]), nullptr, misc, misc[9],
                                                   misc[10], misc + 11, pairs, false);
            std::vector<int> flat;
            for (size_t i = 0; i < pairs.size(); ++i) { flat.push_back((int)pairs[i].first); flat.push_back((int)pairs[i].second); }
            put(o, flat.data(), flat.size() * 4); put(o, &n, 4);
        }
        std::fclose(o);
    } catch (const orbhip::Error &e) {
        std::fprintf(stderr, "orbhip error %d: %s\n", e.code, e.what());
        return 5;
    }
    return 0;
}
