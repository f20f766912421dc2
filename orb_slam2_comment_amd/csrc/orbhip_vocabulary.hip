// MI355X (gfx950) bag-of-words conversion behind the C ABI of include/orbhip.h.
// Replaces, for ORB descriptors, DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>
//   loadFromTextFile                      Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1338-1424
//   transform(features, BowVector&, FeatureVector&, levelsup)                          :1127-1199
//   transform(feature, word_id, weight, nid, levelsup)                                 :1218-1262
// as called by Frame::ComputeBoW / KeyFrame::ComputeBoW (src/Frame.cc:395-402, levelsup 4).
//
// The tree lives in HBM with every node's children stored contiguously (descriptor rows in child-list order), so one
// level of the descent is one coalesced 32*k-byte read per feature.  16 lanes walk one feature (4 features per
// wavefront); the BowVector (std::map<WordId, double> filled in feature order, then L1/L2-normalised in map order) is
// rebuilt by one workgroup per frame: LDS bitonic sort of (word, feature) keys, segment heads add their weights in
// feature order, and the norm is accumulated by a single lane in ascending word order so that every double rounding
// happens in the reference's order.  Integer/bitwise + a handful of fp64 ops: no MFMA.
#include "orbhip_internal.h"

#include <cerrno>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

namespace orbhip {

constexpr int kBowMax = 8192;          // features per frame the assemble kernel sorts in LDS
constexpr uint32_t kNoNode = 0xffffffffu;

struct VocDev {
    const uint32_t *child_off;   // [n_nodes + 1] offsets into the child-ordered arrays
    const uint32_t *child_id;    // [n_nodes - 1] node id of every child slot
    const uint8_t *child_desc;   // [n_nodes - 1][32] descriptor of every child slot
    const uint32_t *word_id;     // [n_nodes]
    const double *weight;        // [n_nodes]
    int L, scoring, weighting, n_words;
};

struct VocBatch {
    const int *n_dev;   // per-frame feature counts (device) or null
    int cap;            // feature stride per frame
};

__device__ __forceinline__ int group16_min(int v)
{
    // 16-lane groups are DPP rows: the butterfly never leaves the group and needs no LDS crossbar round trip
    return orbhip::row16_min(v);
}

// One 16-lane group per feature.  key = distance * 32 + child position, so the group minimum is the reference's
// first strict minimum (:1240-1249).
__global__ __launch_bounds__(256) void k_voc_descend(VocDev V, const uint8_t *__restrict__ desc, int n, int levelsup,
                                                     uint32_t *__restrict__ out_word, double *__restrict__ out_weight,
                                                     uint32_t *__restrict__ out_node, VocBatch B)
{
    const int frame = blockIdx.y;
    desc += (size_t)frame * B.cap * 32;
    out_word += (size_t)frame * B.cap;
    out_weight += (size_t)frame * B.cap;
    out_node += (size_t)frame * B.cap;
    if (B.n_dev) n = min(B.n_dev[frame], B.cap);
    const int sub = threadIdx.x & 15;
    const int i = blockIdx.x * 16 + (threadIdx.x >> 4);
    const bool live = i < n;                      // whole group shares `live`
    uint32_t f[8];
    const uint32_t *fp = reinterpret_cast<const uint32_t *>(desc + (size_t)(live ? i : 0) * 32);
#pragma unroll
    for (int w = 0; w < 8; ++w) f[w] = live ? fp[w] : 0u;
    const int nid_level = V.L - levelsup;
    uint32_t nid = 0;
    bool nid_set = nid_level <= 0;
    uint32_t node = 0;
    int level = 0;
    // every lane of the wave iterates until all four groups have reached a leaf (shuffles need full participation)
    bool done = !live;
    while (true) {
        const uint32_t off = done ? 0u : V.child_off[node];
        const int cc = done ? 0 : (int)(V.child_off[node + 1] - off);
        if (!done && cc == 0) done = true;        // leaf
        if (__ballot(!done) == 0ull) break;
        int best = INT_MAX;
        for (int c = sub; c < cc; c += 16) {
            const uint32_t *cp = reinterpret_cast<const uint32_t *>(V.child_desc + (size_t)(off + c) * 32);
            int d = 0;
#pragma unroll
            for (int w = 0; w < 8; ++w) d += __popc(f[w] ^ cp[w]);
            best = min(best, d * 32 + c);
        }
        best = group16_min(best);
        if (!done) {
            node = V.child_id[off + (uint32_t)(best & 31)];
            ++level;
            if (level == nid_level) { nid = node; nid_set = true; }
        }
    }
    if (live && sub == 0) {
        const double w = V.weight[node];
        out_word[i] = V.word_id[node];
        out_weight[i] = w;
        if (!nid_set) nid = node;                 // leaf above level L - levelsup (reference leaves *nid unset)
        out_node[i] = w > 0 ? nid : kNoNode;      // FeatureVector entry only for non-stopped words (:1161-1165)
    }
}

// BowVector of one frame.  Dynamic LDS: kBowMax 64-bit keys, reused for the values during normalisation.
__global__ __launch_bounds__(1024) void k_bow_assemble(int scoring, int weighting, const uint32_t *__restrict__ word,
                                                       const double *__restrict__ weight, int n,
                                                       uint32_t *__restrict__ bow_ids, double *__restrict__ bow_vals,
                                                       int *__restrict__ n_bow, VocBatch B)
{
    extern __shared__ unsigned long long bow_lds[];
    __shared__ int s_cnt[1024];
    __shared__ int s_total;
    __shared__ double s_norm;
    const int frame = blockIdx.x, tid = threadIdx.x, T = 1024;
    word += (size_t)frame * B.cap;
    weight += (size_t)frame * B.cap;
    bow_ids += (size_t)frame * B.cap;
    bow_vals += (size_t)frame * B.cap;
    n_bow += frame;
    if (B.n_dev) n = min(B.n_dev[frame], B.cap);
    n = min(n, kBowMax);
    int P = 1024;
    while (P < n) P <<= 1;
    unsigned long long *key = bow_lds;
    for (int i = tid; i < P; i += T)
        key[i] = (i < n && weight[i] > 0) ? (((unsigned long long)word[i] << 32) | (unsigned)i) : ~0ull;
    __syncthreads();
    for (int k = 2; k <= P; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < P; i += T) {
                const int l = i ^ j;
                if (l > i) {
                    const unsigned long long a = key[i], b = key[l];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { key[i] = b; key[l] = a; }
                }
            }
            __syncthreads();
        }
    // segment heads -> output slots (each thread owns P/T consecutive sorted positions)
    const int per = P / T, p0 = tid * per;
    int heads = 0;
    for (int p = p0; p < p0 + per; ++p) {
        const unsigned long long kv = key[p];
        if (kv != ~0ull && (p == 0 || (uint32_t)(key[p - 1] >> 32) != (uint32_t)(kv >> 32))) ++heads;
    }
    s_cnt[tid] = heads;
    __syncthreads();
    for (int off = 1; off < T; off <<= 1) {   // inclusive Hillis-Steele scan
        const int v = tid >= off ? s_cnt[tid - off] : 0;
        __syncthreads();
        s_cnt[tid] += v;
        __syncthreads();
    }
    int pos = s_cnt[tid] - heads;
    if (tid == T - 1) s_total = s_cnt[tid];
    const bool accumulate = weighting == 0 || weighting == 1;   // TF_IDF / TF: addWeight; IDF / BINARY: addIfNotExist
    for (int p = p0; p < p0 + per; ++p) {
        const unsigned long long kv = key[p];
        if (kv == ~0ull) break;
        const uint32_t id = (uint32_t)(kv >> 32);
        if (p != 0 && (uint32_t)(key[p - 1] >> 32) == id) continue;
        double val = weight[(uint32_t)kv];
        if (accumulate)
            for (int qn = p + 1; qn < P && (uint32_t)(key[qn] >> 32) == id && key[qn] != ~0ull; ++qn)
                val += weight[(uint32_t)key[qn]];           // BowVector::addWeight, feature order
        bow_ids[pos] = id;
        bow_vals[pos] = val;
        ++pos;
    }
    __syncthreads();
    const int nb = s_total;
    const bool must = scoring != 5;            // DotProductScoring does not normalise
    double *vals = reinterpret_cast<double *>(bow_lds);
    for (int i = tid; i < nb; i += T) {
        double v = bow_vals[i];
        if (accumulate && !must) v /= (double)nb;            // :1170-1176
        vals[i] = v;
    }
    __syncthreads();
    if (must) {                                // BowVector::normalize, map order
        if (tid == 0) {
            double norm = 0.0;
            if (scoring != 1) for (int i = 0; i < nb; ++i) norm += fabs(vals[i]);
            else { for (int i = 0; i < nb; ++i) norm += vals[i] * vals[i]; norm = sqrt(norm); }
            s_norm = norm;
        }
        __syncthreads();
        const double norm = s_norm;
        if (norm > 0.0) for (int i = tid; i < nb; i += T) vals[i] /= norm;
    }
    for (int i = tid; i < nb; i += T) bow_vals[i] = vals[i];
    if (tid == 0) *n_bow = nb;
}

}  // namespace orbhip

using namespace orbhip;

struct orbhip_vocabulary {
    int device = 0;
    int k = 0, L = 0, scoring = 0, weighting = 0, n_nodes = 0, n_words = 0;
    hipStream_t stream = nullptr, own_stream = nullptr;
    void *d_tree = nullptr;             // one allocation holding all tree arrays
    VocDev dev{};
    void *d_work = nullptr; size_t work_bytes = 0;
    uint8_t *h_in = nullptr; size_t h_in_bytes = 0;
    uint8_t *h_out = nullptr; size_t h_out_bytes = 0;
    bool lds_attr_set = false;
};

namespace {

inline size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }

struct HostTree {
    int k = 0, L = 0, scoring = 0, weighting = 0;
    std::vector<int32_t> parent;      // per node (root: 0)
    std::vector<uint8_t> leaf_flag;
    std::vector<uint8_t> desc;        // n_nodes * 32
    std::vector<double> weight;
};

int upload_tree(const HostTree &T, int device, orbhip_vocabulary **out)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        set_error("no HIP device %d (found %d)", device, ndev);
        return ORBHIP_E_NODEVICE;
    }
    const size_t N = T.parent.size();
    // children lists in order of appearance (m_nodes[pid].children.push_back(nid), :1386)
    std::vector<uint32_t> off(N + 1, 0), cid(N > 1 ? N - 1 : 1), wid(N, 0);
    for (size_t i = 1; i < N; ++i) off[(size_t)T.parent[i] + 1]++;
    for (size_t i = 0; i < N; ++i) off[i + 1] += off[i];
    std::vector<uint32_t> fill(off.begin(), off.end() - 1);
    for (size_t i = 1; i < N; ++i) cid[fill[(size_t)T.parent[i]]++] = (uint32_t)i;
    for (size_t i = 0; i < N; ++i)
        if (off[i + 1] - off[i] > 32) {   // the descent kernel packs the child position into 5 bits (DBoW2: k <= 20)
            set_error("vocabulary: node %zu has %u children (limit 32)", i, off[i + 1] - off[i]);
            return ORBHIP_E_ARG;
        }
    std::vector<uint8_t> cdesc((N > 1 ? N - 1 : 1) * 32);
    for (size_t s = 0; s + 1 < N; ++s) memcpy(&cdesc[s * 32], &T.desc[(size_t)cid[s] * 32], 32);
    int n_words = 0;
    for (size_t i = 1; i < N; ++i)
        if (T.leaf_flag[i]) wid[i] = (uint32_t)n_words++;   // :1407-1413
    orbhip_vocabulary *v = new (std::nothrow) orbhip_vocabulary();
    if (!v) return ORBHIP_E_ARG;
    v->device = device;
    v->k = T.k; v->L = T.L; v->scoring = T.scoring; v->weighting = T.weighting;
    v->n_nodes = (int)N; v->n_words = n_words;
    const size_t b_off = al256((N + 1) * 4), b_cid = al256(cid.size() * 4), b_desc = al256(cdesc.size()),
                 b_wid = al256(N * 4), b_w = al256(N * 8);
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&v->own_stream, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc(&v->d_tree, b_off + b_cid + b_desc + b_wid + b_w) != hipSuccess) {
        set_error("vocabulary: device allocation of %zu bytes failed", b_off + b_cid + b_desc + b_wid + b_w);
        if (v->own_stream) (void)hipStreamDestroy(v->own_stream);
        delete v;
        return ORBHIP_E_HIP;
    }
    v->stream = v->own_stream;
    uint8_t *d = (uint8_t *)v->d_tree;
    // one-off upload of the tree: stream-ordered copies on the handle's own stream (the host vectors outlive the wait below)
    hipStream_t us = v->own_stream;
    bool ok = hipMemcpyAsync(d, off.data(), (N + 1) * 4, hipMemcpyHostToDevice, us) == hipSuccess;
    v->dev.child_off = (const uint32_t *)d; d += b_off;
    ok = ok && hipMemcpyAsync(d, cid.data(), cid.size() * 4, hipMemcpyHostToDevice, us) == hipSuccess;
    v->dev.child_id = (const uint32_t *)d; d += b_cid;
    ok = ok && hipMemcpyAsync(d, cdesc.data(), cdesc.size(), hipMemcpyHostToDevice, us) == hipSuccess;
    v->dev.child_desc = d; d += b_desc;
    ok = ok && hipMemcpyAsync(d, wid.data(), N * 4, hipMemcpyHostToDevice, us) == hipSuccess;
    v->dev.word_id = (const uint32_t *)d; d += b_wid;
    ok = ok && hipMemcpyAsync(d, T.weight.data(), N * 8, hipMemcpyHostToDevice, us) == hipSuccess;
    v->dev.weight = (const double *)d;
    ok = ok && hipStreamSynchronize(us) == hipSuccess;
    v->dev.L = T.L; v->dev.scoring = T.scoring; v->dev.weighting = T.weighting; v->dev.n_words = n_words;
    if (!ok) {
        set_error("vocabulary: upload failed");
        orbhip_vocabulary_destroy(v);
        return ORBHIP_E_HIP;
    }
    *out = v;
    return ORBHIP_OK;
}

int check_header(int k, int L, int scoring, int weighting)
{
    // TemplatedVocabulary.h:1362-1366
    if (k < 0 || k > 20 || L < 1 || L > 10 || scoring < 0 || scoring > 5 || weighting < 0 || weighting > 3) {
        set_error("vocabulary: header k=%d L=%d scoring=%d weighting=%d is not a DBoW2 text vocabulary", k, L, scoring, weighting);
        return ORBHIP_E_ARG;
    }
    return ORBHIP_OK;
}

int ensure(orbhip_vocabulary *v, size_t work, size_t in, size_t out)
{
    if (work > v->work_bytes) {
        ORBHIP_HIP_CHECK(hipStreamSynchronize(v->stream));
        if (v->d_work) (void)hipFree(v->d_work);
        v->d_work = nullptr; v->work_bytes = 0;
        ORBHIP_HIP_CHECK(hipMalloc(&v->d_work, work));
        v->work_bytes = work;
    }
    if (in > v->h_in_bytes) {
        if (v->h_in) (void)hipHostFree(v->h_in);
        v->h_in = nullptr; v->h_in_bytes = 0;
        ORBHIP_HIP_CHECK(hipHostMalloc((void **)&v->h_in, in, hipHostMallocDefault));
        v->h_in_bytes = in;
    }
    if (out > v->h_out_bytes) {
        if (v->h_out) (void)hipHostFree(v->h_out);
        v->h_out = nullptr; v->h_out_bytes = 0;
        ORBHIP_HIP_CHECK(hipHostMalloc((void **)&v->h_out, out, hipHostMallocDefault));
        v->h_out_bytes = out;
    }
    return ORBHIP_OK;
}

int launch_transform(orbhip_vocabulary *v, int frames, const uint8_t *d_desc, const int *d_n, int n, int cap, int levelsup,
                     uint32_t *d_word, double *d_weight, uint32_t *d_node, uint32_t *d_bow_ids, double *d_bow_vals,
                     int *d_nbow)
{
    if (!v->lds_attr_set) {
        ORBHIP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_bow_assemble),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, kBowMax * 8));
        v->lds_attr_set = true;
    }
    const VocBatch B = {d_n, cap};
    const int nmax = d_n ? cap : n;
    hipLaunchKernelGGL(k_voc_descend, dim3((nmax + 15) / 16, frames), dim3(256), 0, v->stream, v->dev, d_desc, n, levelsup,
                       d_word, d_weight, d_node, B);
    hipLaunchKernelGGL(k_bow_assemble, dim3(frames), dim3(1024), kBowMax * 8, v->stream, v->scoring, v->weighting,
                       (const uint32_t *)d_word, (const double *)d_weight, n, d_bow_ids, d_bow_vals, d_nbow, B);
    ORBHIP_HIP_CHECK(hipGetLastError());
    return ORBHIP_OK;
}

}  // namespace

extern "C" {

int orbhip_vocabulary_create(int k, int L, int scoring, int weighting, int n_nodes, const int32_t *parent,
                             const uint8_t *is_leaf, const uint8_t *desc, const double *weight, int device,
                             orbhip_vocabulary **out)
{
    if (!out || n_nodes < 0 || (n_nodes > 0 && (!parent || !is_leaf || !desc || !weight))) return ORBHIP_E_ARG;
    int rc = check_header(k, L, scoring, weighting);
    if (rc) return rc;
    HostTree T;
    T.k = k; T.L = L; T.scoring = scoring; T.weighting = weighting;
    const size_t N = (size_t)n_nodes + 1;
    T.parent.assign(N, 0); T.leaf_flag.assign(N, 0); T.desc.assign(N * 32, 0); T.weight.assign(N, 0.0);
    for (int i = 0; i < n_nodes; ++i) {
        if (parent[i] < 0 || parent[i] > i) {   // node i+1 must hang under an earlier node (0 = root)
            set_error("vocabulary: node %d has parent %d (must be an earlier node)", i + 1, parent[i]);
            return ORBHIP_E_ARG;
        }
        T.parent[i + 1] = parent[i];
        T.leaf_flag[i + 1] = is_leaf[i] != 0;
        memcpy(&T.desc[(size_t)(i + 1) * 32], desc + (size_t)i * 32, 32);
        T.weight[i + 1] = weight[i];
    }
    return upload_tree(T, device, out);
}

int orbhip_vocabulary_load_text(const char *path, int device, orbhip_vocabulary **out)
{
    if (!path || !out) return ORBHIP_E_ARG;
    FILE *f = fopen(path, "r");
    if (!f) { set_error("vocabulary: cannot open %s: %s", path, strerror(errno)); return ORBHIP_E_ARG; }
    HostTree T;
    std::vector<char> line(1 << 12);
    int rc = ORBHIP_E_ARG;
    if (fgets(line.data(), (int)line.size(), f)) {
        int k = -1, L = -1, n1 = -1, n2 = -1;
        if (sscanf(line.data(), "%d %d %d %d", &k, &L, &n1, &n2) != 4) set_error("vocabulary: %s has no 'k L scoring weighting' header", path);
        else if ((rc = check_header(k, L, n1, n2)) == ORBHIP_OK) { T.k = k; T.L = L; T.scoring = n1; T.weighting = n2; }
    } else set_error("vocabulary: %s is empty", path);
    if (rc == ORBHIP_OK) {
        T.parent.push_back(0); T.leaf_flag.push_back(0); T.weight.push_back(0.0); T.desc.resize(32, 0);   // root
        long lineno = 1;
        while (fgets(line.data(), (int)line.size(), f)) {
            ++lineno;
            char *p = line.data(), *e;
            while (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\n') ++p;
            if (!*p) continue;   // blank line: see the note on the trailing line in DESIGN.md
            const long nid = (long)T.parent.size();
            const long pid = strtol(p, &e, 10);
            bool ok = e != p && pid >= 0 && pid < nid;
            p = e;
            const long leaf = ok ? strtol(p, &e, 10) : 0;
            ok = ok && e != p;
            p = e;
            uint8_t d[32];
            for (int i = 0; ok && i < 32; ++i) {
                const long b = strtol(p, &e, 10);
                ok = e != p;
                d[i] = (uint8_t)b;
                p = e;
            }
            const double w = ok ? strtod(p, &e) : 0.0;
            ok = ok && e != p;
            if (!ok) { set_error("vocabulary: %s line %ld is not 'parent isLeaf d0..d31 weight'", path, lineno); rc = ORBHIP_E_ARG; break; }
            T.parent.push_back((int32_t)pid);
            T.leaf_flag.push_back(leaf > 0);
            T.desc.insert(T.desc.end(), d, d + 32);
            T.weight.push_back(w);
        }
    }
    fclose(f);
    if (rc) return rc;
    return upload_tree(T, device, out);
}

void orbhip_vocabulary_destroy(orbhip_vocabulary *v)
{
    if (!v) return;
    (void)hipSetDevice(v->device);
    if (v->stream) (void)hipStreamSynchronize(v->stream);
    if (v->d_tree) (void)hipFree(v->d_tree);
    if (v->d_work) (void)hipFree(v->d_work);
    if (v->h_in) (void)hipHostFree(v->h_in);
    if (v->h_out) (void)hipHostFree(v->h_out);
    if (v->own_stream) (void)hipStreamDestroy(v->own_stream);
    delete v;
}

int orbhip_vocabulary_info(const orbhip_vocabulary *v, int *k, int *L, int *scoring, int *weighting, int *n_nodes,
                           int *n_words)
{
    if (!v) return ORBHIP_E_ARG;
    if (k) *k = v->k;
    if (L) *L = v->L;
    if (scoring) *scoring = v->scoring;
    if (weighting) *weighting = v->weighting;
    if (n_nodes) *n_nodes = v->n_nodes;
    if (n_words) *n_words = v->n_words;
    return ORBHIP_OK;
}

int orbhip_vocabulary_set_stream(orbhip_vocabulary *v, void *stream)
{
    if (!v) return ORBHIP_E_ARG;
    ORBHIP_HIP_CHECK(hipSetDevice(v->device));
    ORBHIP_HIP_CHECK(hipStreamSynchronize(v->stream));
    v->stream = stream ? (hipStream_t)stream : v->own_stream;
    return ORBHIP_OK;
}

int orbhip_vocabulary_sync(orbhip_vocabulary *v)
{
    if (!v) return ORBHIP_E_ARG;
    ORBHIP_HIP_CHECK(hipSetDevice(v->device));
    ORBHIP_HIP_CHECK(hipStreamSynchronize(v->stream));
    return ORBHIP_OK;
}

int orbhip_vocabulary_transform(orbhip_vocabulary *v, const uint8_t *desc, int n, int levelsup, uint32_t *word_id,
                                double *word_weight, uint32_t *node_id, uint32_t *bow_ids, double *bow_vals, int *n_bow)
{
    if (!v || n < 0 || (n > 0 && !desc) || !n_bow) return ORBHIP_E_ARG;
    *n_bow = 0;
    if (n > kBowMax) {
        set_error("vocabulary: %d features exceed the per-frame limit %d", n, kBowMax);
        return ORBHIP_E_CAPACITY;
    }
    if (n == 0) return ORBHIP_OK;
    if (v->n_words == 0) {   // empty(): v and fv stay empty (:1134-1137)
        for (int i = 0; i < n; ++i) {
            if (word_id) word_id[i] = 0;
            if (word_weight) word_weight[i] = 0.0;
            if (node_id) node_id[i] = kNoNode;
        }
        return ORBHIP_OK;
    }
    ORBHIP_HIP_CHECK(hipSetDevice(v->device));
    const size_t N = (size_t)n;
    // device/work layout: desc | word u32 | node u32 | bow_ids u32 | n_bow | weight f64 | bow_vals f64
    const size_t o_word = al256(N * 32), o_node = o_word + al256(N * 4), o_bid = o_node + al256(N * 4),
                 o_nb = o_bid + al256(N * 4), o_w = o_nb + 256, o_bv = o_w + al256(N * 8), total = o_bv + al256(N * 8);
    ORBHIP_HIP_CHECK(hipStreamSynchronize(v->stream));
    int rc = ensure(v, total, N * 32, total);
    if (rc) return rc;
    memcpy(v->h_in, desc, N * 32);
    uint8_t *d = (uint8_t *)v->d_work;
    ORBHIP_HIP_CHECK(hipMemcpyAsync(d, v->h_in, N * 32, hipMemcpyHostToDevice, v->stream));
    rc = launch_transform(v, 1, d, nullptr, n, n, levelsup, (uint32_t *)(d + o_word), (double *)(d + o_w),
                          (uint32_t *)(d + o_node), (uint32_t *)(d + o_bid), (double *)(d + o_bv), (int *)(d + o_nb));
    if (rc) return rc;
    ORBHIP_HIP_CHECK(hipMemcpyAsync(v->h_out + o_word, d + o_word, total - o_word, hipMemcpyDeviceToHost, v->stream));
    ORBHIP_HIP_CHECK(hipStreamSynchronize(v->stream));
    const int nb = *reinterpret_cast<const int *>(v->h_out + o_nb);
    if (word_id) memcpy(word_id, v->h_out + o_word, N * 4);
    if (word_weight) memcpy(word_weight, v->h_out + o_w, N * 8);
    if (node_id) memcpy(node_id, v->h_out + o_node, N * 4);
    if (bow_ids) memcpy(bow_ids, v->h_out + o_bid, (size_t)nb * 4);
    if (bow_vals) memcpy(bow_vals, v->h_out + o_bv, (size_t)nb * 8);
    *n_bow = nb;
    return ORBHIP_OK;
}

int orbhip_vocabulary_transform_device(orbhip_vocabulary *v, int frames, const void *d_desc, const void *d_n, int cap,
                                       int levelsup, void *d_word_id, void *d_word_weight, void *d_node_id,
                                       void *d_bow_ids, void *d_bow_vals, void *d_n_bow)
{
    if (!v || frames < 0 || cap < 1 || !d_desc || !d_n || !d_word_id || !d_word_weight || !d_node_id || !d_bow_ids ||
        !d_bow_vals || !d_n_bow)
        return ORBHIP_E_ARG;
    if (cap > kBowMax) {
        set_error("vocabulary: capacity %d exceeds the per-frame limit %d", cap, kBowMax);
        return ORBHIP_E_CAPACITY;
    }
    if (frames == 0) return ORBHIP_OK;
    if (v->n_words == 0) { set_error("vocabulary: empty vocabulary"); return ORBHIP_E_ARG; }
    ORBHIP_HIP_CHECK(hipSetDevice(v->device));
    return launch_transform(v, frames, (const uint8_t *)d_desc, (const int *)d_n, 0, cap, levelsup, (uint32_t *)d_word_id,
                            (double *)d_word_weight, (uint32_t *)d_node_id, (uint32_t *)d_bow_ids, (double *)d_bow_vals,
                            (int *)d_n_bow);
}

}  // extern "C"
