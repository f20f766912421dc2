// MI355X (gfx950) descriptor matching behind the C ABI of include/orbhip.h.
// Replaces ORBmatcher::SearchByProjection (x2), SearchForInitialization,
// DescriptorDistance (src/ORBmatcher.cc) and Frame::ComputeStereoMatches
// (src/Frame.cc:466-640).
//
// Structure: a fully parallel "window search" kernel (one wavefront per query:
// Frame::GetFeaturesInArea membership test + 256-bit XOR/popcount Hamming
// distance, candidates compacted with wave ballots and sorted by
// (distance, reference visiting order)), followed by a single-wavefront
// "resolve" kernel that replays the reference's order-dependent bookkeeping
// (slots taken by earlier queries, match stealing, rotation histogram) over
// state held in LDS.  Integer/bitwise path: no MFMA.
#include "orbhip_internal.h"

#include <algorithm>
#include <climits>
#include <new>
#include <vector>

namespace orbhip {

constexpr int TH_HIGH = 100, TH_LOW = 50, HISTO_LENGTH = 30;
constexpr int GRID_ROWS = 48, GRID_COLS = 64;   // include/Frame.h:37-38
constexpr int kResolveMax = 4096;               // LDS-resident state of the resolve kernel
constexpr uint32_t kNoCell = 0xffffffffu;

struct DevFrame {
    int n;
    const orbhip_keypoint *keys;
    const uint8_t *desc;
    const float *u_right;  // nullable
    float min_x, min_y, inv_w, inv_h;
};

// Batched (device-resident) calls: pair p = blockIdx.y uses element strides cap / qcap; the single-pair
// host path passes zero strides and null count pointers.
struct Batch {
    const int *n_dev;    // per-frame train keypoint counts, or null (use F.n)
    const int *nq_dev;   // per-pair query counts, or null (use the nq argument)
    int cap, qcap;
    int t0 = 0, ts = 1;    // train side of pair p = frame t0 + p*ts of the extractor-layout arrays
    int qd0 = 0, qds = 1;  // query descriptors of pair p = frame qd0 + p*qds of their array
};
__device__ __forceinline__ void batch_frame(DevFrame &F, const Batch &B, int pair)
{
    const size_t f = (size_t)(B.t0 + pair * B.ts);
    F.keys += f * B.cap;
    F.desc += f * B.cap * 32;
    if (F.u_right) F.u_right += f * B.cap;
    if (B.n_dev) F.n = min(B.n_dev[f], B.cap);
}

__device__ __forceinline__ int hamming256(const uint32_t *a, const uint32_t *b)
{
    int d = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) d += __popc(a[i] ^ b[i]);
    return d;
}

__device__ __forceinline__ int wave_reduce_add_i(int v)
{
    return wave_sum(v);
}

// 64-bit minimum over the wavefront (wave-uniform result): the same DPP butterfly + row broadcasts as wave_sum, both
// halves of the key travelling together
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v)
{
#define ORBHIP_MIN64_STEP(ctrl, rmask)                                                                        \
    {                                                                                                         \
        const uint32_t lo_ = (uint32_t)v, hi_ = (uint32_t)(v >> 32);                                          \
        const uint32_t ol_ = (uint32_t)__builtin_amdgcn_update_dpp((int)lo_, (int)lo_, ctrl, rmask, 0xf, false); \
        const uint32_t oh_ = (uint32_t)__builtin_amdgcn_update_dpp((int)hi_, (int)hi_, ctrl, rmask, 0xf, false); \
        const unsigned long long o_ = ((unsigned long long)oh_ << 32) | ol_;                                  \
        v = o_ < v ? o_ : v;                                                                                  \
    }
    ORBHIP_MIN64_STEP(0xB1, 0xf)
    ORBHIP_MIN64_STEP(0x4E, 0xf)
    ORBHIP_MIN64_STEP(0x141, 0xf)
    ORBHIP_MIN64_STEP(0x140, 0xf)
    ORBHIP_MIN64_STEP(0x142, 0xa)
    ORBHIP_MIN64_STEP(0x143, 0xc)
#undef ORBHIP_MIN64_STEP
    return ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(v >> 32), 63) << 32) |
           (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, 63);
}

// The smallest keys of a list whose entries are spread over the lanes, in ascending order, EXACTLY: every lane hands in
// its two smallest keys (a1 < a2, ~0 = none) and whether it holds more than those two.  Minimum after minimum is
// extracted; once a lane has given both of its keys and still holds others, a later minimum could be one of those, so
// the extraction stops there.  Keys are unique.  Returns the number of keys written to head[] (wave-uniform, <= kHeadMax);
// *complete = the list has no further key (the extraction ran dry, not into the stop condition or the limit).
constexpr int kHeadMax = 8;
__device__ __forceinline__ int wave_sorted_head(unsigned long long a1, unsigned long long a2, bool more,
                                                unsigned long long (&head)[kHeadMax], bool *complete)
{
    int given = 0, hl = 0;
    bool stop = false, dry = false;
#pragma unroll
    for (int k = 0; k < kHeadMax; ++k) {
        head[k] = ~0ull;
        if (stop || dry) continue;          // wave-uniform
        const unsigned long long c = given == 0 ? a1 : (given == 1 ? a2 : ~0ull);
        const unsigned long long g = wave_min_u64(c);
        if (g == ~0ull) { dry = true; continue; }
        head[k] = g;
        hl = k + 1;
        const bool mine = c == g;
        if (mine) ++given;
        stop = __any(mine && given == 2 && more) != 0;
    }
    // a lane that still holds an unextracted tracked key, or more keys than it tracks, means the list goes on
    const bool rest = (given == 0 && a1 != ~0ull) || (given <= 1 && a2 != ~0ull) || more;
    *complete = !__any(rest);
    return hl;
}

// Visiting-order key of every train keypoint, (posX*48+posY) << 20 | index, or kNoCell when PosInGrid rejects it
// (k_best_in_window scans the frame with it).
__global__ void k_grid_order(DevFrame F, uint32_t *__restrict__ ord, Batch B)
{
    batch_frame(F, B, blockIdx.y);
    ord += (size_t)blockIdx.y * B.cap;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= F.n) return;
    const orbhip_keypoint kp = F.keys[j];
    int px = (int)roundf(__fmul_rn(__fsub_rn(kp.x, F.min_x), F.inv_w));
    int py = (int)roundf(__fmul_rn(__fsub_rn(kp.y, F.min_y), F.inv_h));
    bool in = !(px < 0 || px >= GRID_COLS || py < 0 || py >= GRID_ROWS);
    ord[j] = in ? (((uint32_t)(px * GRID_ROWS + py) << 20) | (uint32_t)j) : kNoCell;
}

// mGrid of the train frame (Frame::AssignFeaturesToGrid, Frame.cc:230-245) as CSR, one workgroup per frame: cell
// c = posX * 48 + posY owns the records rec[start[c] .. start[c+1]).  A record carries everything the window search
// needs of a keypoint -- (x, y, octave, index) and, in a parallel array, its descriptor -- so a query reads its
// candidates with addresses that depend on the cell table only (one memory round trip, not index -> keypoint ->
// descriptor).  Counting sort with LDS atomics; the order INSIDE a cell is arbitrary: every consumer orders candidates
// by the key (cell << 20 | index) = the visiting order of Frame::GetFeaturesInArea (cell-major ix outer / iy inner,
// insertion order inside a cell; Frame.cc:350-358).  Keypoints that PosInGrid rejects (Frame.cc:382-392) are in no
// cell and never candidates.
constexpr int kGridCells = GRID_COLS * GRID_ROWS;
struct GridRec { float x, y; int octave; uint32_t key; };   // key = cell << 20 | index
__device__ __forceinline__ void grid_build_body(DevFrame F, int *__restrict__ cell_start, GridRec *__restrict__ rec,
                                                uint4 *__restrict__ rdesc, float *__restrict__ rur, const Batch &B, const int pair)
{
    __shared__ int s_cnt[kGridCells];
    __shared__ int s_scan[8];
    const int tid = threadIdx.x;
    batch_frame(F, B, pair);
    const size_t rbase = (size_t)pair * (B.cap > 0 ? B.cap : F.n);
    cell_start += (size_t)pair * (kGridCells + 1);
    rec += rbase; rdesc += rbase * 2; rur += rbase;
    for (int c = tid; c < kGridCells; c += 256) s_cnt[c] = 0;
    __syncthreads();
    for (int j = tid; j < F.n; j += 256) {
        const orbhip_keypoint kp = F.keys[j];
        const int px = (int)roundf(__fmul_rn(__fsub_rn(kp.x, F.min_x), F.inv_w));
        const int py = (int)roundf(__fmul_rn(__fsub_rn(kp.y, F.min_y), F.inv_h));
        if (!(px < 0 || px >= GRID_COLS || py < 0 || py >= GRID_ROWS)) atomicAdd(&s_cnt[px * GRID_ROWS + py], 1);
    }
    __syncthreads();
    // exclusive scan over the 3072 cells: 12 consecutive cells per thread
    constexpr int PER = kGridCells / 256;
    int local[PER], sum = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) { local[k] = s_cnt[tid * PER + k]; sum += local[k]; }
    int incl = sum;
    const int lane = tid & 63, wave = tid >> 6;
    incl = wave_incl_scan_add(incl);
    if (lane == 63) s_scan[wave] = incl;
    __syncthreads();
    int base = incl - sum;
    for (int w = 0; w < wave; ++w) base += s_scan[w];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER; ++k) { s_cnt[tid * PER + k] = base; cell_start[tid * PER + k] = base; base += local[k]; }
    if (tid == 255) cell_start[kGridCells] = base;
    __syncthreads();
    for (int j = tid; j < F.n; j += 256) {
        const orbhip_keypoint kp = F.keys[j];
        const int px = (int)roundf(__fmul_rn(__fsub_rn(kp.x, F.min_x), F.inv_w));
        const int py = (int)roundf(__fmul_rn(__fsub_rn(kp.y, F.min_y), F.inv_h));
        if (!(px < 0 || px >= GRID_COLS || py < 0 || py >= GRID_ROWS)) {
            const int cell = px * GRID_ROWS + py;
            const int pos = atomicAdd(&s_cnt[cell], 1);
            GridRec r; r.x = kp.x; r.y = kp.y; r.octave = kp.octave; r.key = ((uint32_t)cell << 20) | (uint32_t)j;
            rec[pos] = r;
            const uint4 *d = reinterpret_cast<const uint4 *>(F.desc + (size_t)j * 32);
            rdesc[2 * pos] = d[0]; rdesc[2 * pos + 1] = d[1];
            rur[pos] = F.u_right ? F.u_right[j] : -1.0f;
        }
    }
}
__global__ __launch_bounds__(256) void k_grid_build(DevFrame F, int *__restrict__ cell_start, GridRec *__restrict__ rec,
                                                    uint4 *__restrict__ rdesc, float *__restrict__ rur, Batch B)
{
    grid_build_body(F, cell_start, rec, rdesc, rur, B, (int)blockIdx.x);
}

// One wavefront per query.  The cells of the query's window (Frame.cc:332-346) are dealt to the lanes; the records of
// all those cells are then flattened over the lanes (wave prefix sum of the cell populations + an LDS scatter of the
// record positions), so one pass tests 64 candidates: level window, |dx|,|dy| < r, stereo gate, Hamming distance.
// Output: dist << 32 | (cell << 20 | index) keys; with <= 64 candidates the list is bitonic-sorted (= (distance,
// reference visiting order)) into ccand[q*64 ..] and cnt[q] > 0, otherwise it stays unsorted in cand[q*stride ..] with
// cnt[q] = -count.
constexpr int kCompact = 64;
__global__ __launch_bounds__(256) void k_window_search(DevFrame F, const int *__restrict__ cell_start,
                                                       const GridRec *__restrict__ rec, const uint4 *__restrict__ rdesc,
                                                       const float *__restrict__ rur,
                                                       const orbhip_query *__restrict__ q,
                                                       const uint8_t *__restrict__ qdesc, int nq,
                                                       unsigned long long *__restrict__ cand,
                                                       unsigned long long *__restrict__ ccand,
                                                       int *__restrict__ cnt, int stride, int use_ur, Batch B,
                                                       const uint8_t *__restrict__ taken, int max_dist)
{
    __shared__ unsigned long long stage[4][64];
    __shared__ int spos[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int pair = blockIdx.y;
    const bool has_ur = F.u_right != nullptr;
    batch_frame(F, B, pair);
    const size_t rbase = (size_t)pair * (B.cap > 0 ? B.cap : F.n);
    cell_start += (size_t)pair * (kGridCells + 1);
    rec += rbase; rdesc += rbase * 2; rur += rbase;
    q += (size_t)pair * B.qcap;
    qdesc += (size_t)(B.qd0 + pair * B.qds) * B.qcap * 32;
    cand += (size_t)pair * B.qcap * stride;
    ccand += (size_t)pair * B.qcap * kCompact;
    cnt += (size_t)pair * B.qcap;
    if (taken) taken += (size_t)pair * B.cap;
    if (B.nq_dev) nq = min(B.nq_dev[pair], B.qcap);
    const int qi = blockIdx.x * 4 + wv;
    if (qi >= nq) return;
    const orbhip_query Q = q[qi];
    if (!Q.valid) { if (lane == 0) cnt[qi] = 0; return; }
    const float x = Q.u, y = Q.v, r = Q.radius;
    // Frame.cc:332-346
    const int nMinCellX = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(x, F.min_x), r), F.inv_w)));
    const int nMaxCellX = min(GRID_COLS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(x, F.min_x), r), F.inv_w)));
    const int nMinCellY = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(y, F.min_y), r), F.inv_h)));
    const int nMaxCellY = min(GRID_ROWS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(y, F.min_y), r), F.inv_h)));
    if (nMinCellX >= GRID_COLS || nMaxCellX < 0 || nMinCellY >= GRID_ROWS || nMaxCellY < 0 || nMaxCellX < nMinCellX ||
        nMaxCellY < nMinCellY) {
        if (lane == 0) cnt[qi] = 0;
        return;
    }
    const bool bCheckLevels = (Q.min_level > 0) || (Q.max_level >= 0);
    uint32_t qd[8];
    const uint32_t *qp = reinterpret_cast<const uint32_t *>(qdesc + (size_t)qi * 32);
#pragma unroll
    for (int i = 0; i < 8; ++i) qd[i] = qp[i];
    unsigned long long *out = cand + (size_t)qi * stride;
    const int ncy = nMaxCellY - nMinCellY + 1, ncell = (nMaxCellX - nMinCellX + 1) * ncy;
    int total = 0;
    unsigned long long b1 = ~0ull, b2 = ~0ull;   // this lane's two smallest keys among candidates not taken on entry
    int nb = 0;                                  // ... out of how many
    // a column of the window is a contiguous run of cells (cell = ix * 48 + iy), so its records are one contiguous CSR
    // range: lane = window column, range = [start[ix*48 + y0], start[ix*48 + y1 + 1])
    const int ncol = nMaxCellX - nMinCellX + 1;
    (void)ncell;
    for (int c0 = 0; c0 < ncol; c0 += 64) {
        const int c = c0 + lane;
        int start = 0, ncand = 0;
        if (c < ncol) {
            const int cell0 = (nMinCellX + c) * GRID_ROWS + nMinCellY;
            start = cell_start[cell0];
            ncand = cell_start[cell0 + ncy] - start;
        }
        // exclusive prefix of the column populations over the wave
        int incl = ncand;
        incl = wave_incl_scan_add(incl);
        const int nrec = __builtin_amdgcn_readlane(incl, 63);
        const int excl = incl - ncand;
        for (int r0 = 0; r0 < nrec; r0 += 64) {
            // scatter: flat record t of this chunk comes from CSR position spos[t - r0]
            __builtin_amdgcn_wave_barrier();
            for (int k = max(0, r0 - excl); k < ncand && excl + k < r0 + 64; ++k) spos[wv][excl + k - r0] = start + k;
            __builtin_amdgcn_wave_barrier();
            bool ok = false;
            unsigned long long key = 0;
            if (r0 + lane < nrec) {
                const int p = spos[wv][lane];
                const GridRec R = rec[p];
                const uint4 d0 = rdesc[2 * p], d1 = rdesc[2 * p + 1];
                ok = true;
                if (bCheckLevels) {
                    if (R.octave < Q.min_level) ok = false;
                    if (Q.max_level >= 0 && R.octave > Q.max_level) ok = false;
                }
                ok = ok && fabsf(__fsub_rn(R.x, x)) < r && fabsf(__fsub_rn(R.y, y)) < r;
                if (ok && use_ur && has_ur) {
                    const float ur = rur[p];
                    if (ur > 0 && fabsf(__fsub_rn(Q.ur, ur)) > r) ok = false;
                }
                const int dist = __popc(qd[0] ^ d0.x) + __popc(qd[1] ^ d0.y) + __popc(qd[2] ^ d0.z) + __popc(qd[3] ^ d0.w) +
                                 __popc(qd[4] ^ d1.x) + __popc(qd[5] ^ d1.y) + __popc(qd[6] ^ d1.z) + __popc(qd[7] ^ d1.w);
                key = ((unsigned long long)dist << 32) | R.key;
                // searches that take the best candidate alone (no second best, no ratio test) never look past the first
                // free entry, and an entry beyond their acceptance threshold can only mean "no match": it need not be listed
                ok = ok && dist <= max_dist;
            }
            const unsigned long long bal = __ballot(ok);
            if (ok) {
                const int pos = total + __popcll(bal & ((1ull << lane) - 1ull));
                out[pos] = key;
                if (pos < 64) stage[wv][pos] = key;
                if (!(taken && taken[key & 0xfffffu])) {
                    if (key < b1) { b2 = b1; b1 = key; } else if (key < b2) b2 = key;
                    ++nb;
                }
            }
            total += __popcll(bal);
        }
    }
    if (total > 0 && total <= 64) {
        // rank sort of up to 64 distinct keys: lane i counts the keys below its own -- key j comes to all lanes through
        // v_readlane (j is wave-uniform), so a key costs two readlanes, one 64-bit compare and one add -- and stores its
        // key at its rank; cheaper than a bitonic network (12 instructions per stage, 10 .. 21 stages) at every length
        __builtin_amdgcn_wave_barrier();
        const unsigned long long v = lane < total ? stage[wv][lane] : ~0ull;
        const uint32_t vlo = (uint32_t)v, vhi = (uint32_t)(v >> 32);
        const int tu = __builtin_amdgcn_readfirstlane(total);
        int rank = 0;
        for (int j = 0; j < tu; ++j) {
            const unsigned long long kj = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)vhi, j) << 32) |
                                          (uint32_t)__builtin_amdgcn_readlane((int)vlo, j);
            rank += kj < v ? 1 : 0;
        }
        if (lane < total) ccand[(size_t)qi * kCompact + rank] = v;
        if (lane == 0) cnt[qi] = total;
    } else {
        // more than 64 candidates: the list stays unsorted in `cand`; the first (up to kHeadMax) keys of its sorted order
        // among the candidates not taken on entry go to the head of the compact row -- {keys[8], count | complete << 8} --
        // so the resolve walks a sorted head like it does for short lists and only scans the list when the head runs out
        if (total > 64) {
            unsigned long long head[kHeadMax];
            bool complete;
            const int hl = wave_sorted_head(b1, b2, nb > 2, head, &complete);
            if (lane < kHeadMax) {
                unsigned long long v = head[0];
#pragma unroll
                for (int k = 1; k < kHeadMax; ++k) if (lane == k) v = head[k];
                ccand[(size_t)qi * kCompact + lane] = v;
            }
            if (lane == 0) ccand[(size_t)qi * kCompact + kHeadMax] = (unsigned long long)hl | (complete ? 0x100ull : 0ull);
        }
        if (lane == 0) cnt[qi] = -total;
    }
}

// ---- vocabulary-node guided searches (SearchByBoW :159-288 / :522-655, SearchForTriangulation :657-823) -------
// Both visit the key frame's features in FeatureVector order (node id, then feature index) and only compare
// features filed under the same vocabulary node, in ascending index.
//
// SearchByBoW: matches of one node never interact with another node (a feature sits in exactly one node), so one
// wavefront owns one common node and replays its queries in order; lanes hold the node's candidates (the first 64
// in registers), the "already matched" flags live in slot space and each slot is only ever touched by the lane
// that owns it.  Best / second-best = two wave minima of (distance << 32 | slot).
struct NodeGroup { int q_begin, q_end, t_begin, t_end; };
__device__ __forceinline__ int rot_bin(float a1, float a2);
__device__ __forceinline__ void three_maxima(const int *h, int &ind1, int &ind2, int &ind3);

__global__ __launch_bounds__(256) void k_bow_groups(const NodeGroup *__restrict__ groups, int ngroups,
                                                    const uint8_t *__restrict__ qdesc, const float *__restrict__ qangle,
                                                    const uint32_t *__restrict__ t_order, const uint8_t *__restrict__ tdesc,
                                                    const orbhip_keypoint *__restrict__ tkeys,
                                                    uint8_t *__restrict__ matched, int max_dist, float nnratio,
                                                    int check_ori, int *__restrict__ match_out,
                                                    uint8_t *__restrict__ bin_out, int *__restrict__ hist)
{
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= ngroups) return;
    const NodeGroup G = groups[g];
    const int tc = G.t_end - G.t_begin;
    // first chunk of candidates in registers
    uint32_t td0[8];
    const bool have0 = lane < tc;
    {
        const uint32_t *tp = reinterpret_cast<const uint32_t *>(tdesc + (size_t)(have0 ? t_order[G.t_begin + lane] : 0) * 32);
#pragma unroll
        for (int i = 0; i < 8; ++i) td0[i] = have0 ? tp[i] : 0u;
    }
    bool used0 = have0 ? matched[G.t_begin + lane] != 0 : true;
    uint32_t qnext[8];
    {
        const uint32_t *qp = reinterpret_cast<const uint32_t *>(qdesc + (size_t)G.q_begin * 32);
#pragma unroll
        for (int i = 0; i < 8; ++i) qnext[i] = qp[i];
    }
    for (int qi = G.q_begin; qi < G.q_end; ++qi) {
        uint32_t qd[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) qd[i] = qnext[i];
        if (qi + 1 < G.q_end) {   // the next query's descriptor travels while this one is resolved
            const uint32_t *qp = reinterpret_cast<const uint32_t *>(qdesc + (size_t)(qi + 1) * 32);
#pragma unroll
            for (int i = 0; i < 8; ++i) qnext[i] = qp[i];
        }
        unsigned long long k1 = ~0ull, k2 = ~0ull;
        if (!used0) k1 = ((unsigned long long)hamming256(qd, td0) << 32) | (unsigned)(G.t_begin + lane);
        for (int c = G.t_begin + 64 + lane; c < G.t_end; c += 64) {
            if (matched[c]) continue;
            const uint32_t *tp = reinterpret_cast<const uint32_t *>(tdesc + (size_t)t_order[c] * 32);
            uint32_t td[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) td[i] = tp[i];
            const unsigned long long k = ((unsigned long long)hamming256(qd, td) << 32) | (unsigned)c;
            if (k < k1) { k2 = k1; k1 = k; } else if (k < k2) k2 = k;
        }
        const unsigned long long m1 = wave_min_u64(k1);
        int res = -1;
        if (m1 != ~0ull) {
            const unsigned long long m2 = wave_min_u64(k1 == m1 ? k2 : k1);
            const int bestDist1 = (int)(m1 >> 32), bestDist2 = m2 == ~0ull ? 256 : (int)(m2 >> 32);
            if (bestDist1 <= max_dist && (float)bestDist1 < __fmul_rn(nnratio, (float)bestDist2)) {   // :262-264 / :598-600
                const int slot = (int)(uint32_t)m1;
                if (((slot - G.t_begin) & 63) == lane) {
                    matched[slot] = 1;
                    if (slot - G.t_begin < 64) used0 = true;
                }
                res = (int)t_order[slot];
            }
        }
        if (lane == 0) {
            match_out[qi] = res;
            int bin = 0xff;
            if (res >= 0 && check_ori) {
                bin = rot_bin(qangle[qi], tkeys[res].angle);
                if (bin >= 0) atomicAdd(&hist[bin], 1); else bin = 0xff;
            }
            bin_out[qi] = (uint8_t)bin;
        }
    }
}

// Device-resident, batched SearchByBoW: one 1024-thread workgroup per (key frame, frame) pair does everything the
// host path splits between std::sort, k_bow_groups and k_bow_cull: both sides' (node << 32 | index) keys are sorted
// in LDS (= FeatureVector order), group heads are found on the sorted query keys, the node's candidate range by
// binary search in the sorted train keys, then the 16 wavefronts take the common nodes round-robin and replay each
// node's queries in order exactly like k_bow_groups; rotation histogram, cull and count finish in the same launch.
constexpr int kBowPairMax = 4096;
constexpr int kBowStage = 64;       // queries of a node whose descriptors a wavefront fetches in one go
struct BowPairShared {
    unsigned long long qkey[kBowPairMax], tkey[kBowPairMax];
    int mout[kBowPairMax];
    unsigned short ghead[kBowPairMax];
    unsigned char matched[kBowPairMax], bin[kBowPairMax];
    int hist[HISTO_LENGTH];
    int nq, nt, ngroups, nmatch, next_group;
    uint32_t qstage[16][kBowStage * 8];   // per wavefront: the descriptors of up to kBowStage queries of the node it replays
    unsigned short gt0[kBowPairMax], gt1[kBowPairMax], gq1[kBowPairMax];   // per group: train range, end of the query range
};
struct BowSide {   // one side of the pairs, in the extractor / vocabulary output layout
    const orbhip_keypoint *kps;
    const uint8_t *desc;
    const int *n;
    const uint32_t *node;
    const uint8_t *flag;   // query side: valid1 (null = all); train side: blocked2 (null = none)
    int f0, fs;            // frame index of pair p = f0 + p * fs
};

__device__ __forceinline__ int lower_bound_u64(const unsigned long long *a, int n, unsigned long long v)
{
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(1024) void k_bow_pairs(BowSide Q, BowSide T, int cap, int max_dist, float nnratio,
                                                    int check_ori, int *__restrict__ matches12, int *__restrict__ nmatches)
{
    extern __shared__ unsigned char bow_pair_lds[];
    BowPairShared &S = *reinterpret_cast<BowPairShared *>(bow_pair_lds);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, NT = 1024;
    const int pair = blockIdx.x, fq = Q.f0 + pair * Q.fs, ft = T.f0 + pair * T.fs;
    const int n1 = min(Q.n[fq], cap), n2 = min(T.n[ft], cap);
    const orbhip_keypoint *qk = Q.kps + (size_t)fq * cap, *tk = T.kps + (size_t)ft * cap;
    const uint8_t *qd = Q.desc + (size_t)fq * cap * 32, *td = T.desc + (size_t)ft * cap * 32;
    const uint32_t *qn = Q.node + (size_t)fq * cap, *tn = T.node + (size_t)ft * cap;
    const uint8_t *valid1 = Q.flag ? Q.flag + (size_t)fq * cap : nullptr;
    const uint8_t *blocked2 = T.flag ? T.flag + (size_t)ft * cap : nullptr;
    matches12 += (size_t)pair * cap;
    int P = 1024;
    while (P < max(n1, n2)) P <<= 1;
    for (int i = tid; i < P; i += NT) {
        const bool okq = i < n1 && qn[i] != 0xffffffffu && (!valid1 || valid1[i]);
        S.qkey[i] = okq ? (((unsigned long long)qn[i] << 32) | (unsigned)i) : ~0ull;
        const bool okt = i < n2 && tn[i] != 0xffffffffu;
        S.tkey[i] = okt ? (((unsigned long long)tn[i] << 32) | (unsigned)i) : ~0ull;
        S.mout[i] = -1;
        S.bin[i] = 0xff;
    }
    for (int i = tid; i < cap; i += NT) matches12[i] = -1;
    if (tid < HISTO_LENGTH) S.hist[tid] = 0;
    if (tid == 0) { S.nq = 0; S.nt = 0; S.ngroups = 0; S.nmatch = 0; S.next_group = 0; }
    __syncthreads();
    for (int k = 2; k <= P; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < P; i += NT) {
                const int l = i ^ j;
                if (l > i) {
                    const bool up = (i & k) == 0;
                    const unsigned long long a = S.qkey[i], b = S.qkey[l];
                    if ((a > b) == up) { S.qkey[i] = b; S.qkey[l] = a; }
                    const unsigned long long c = S.tkey[i], d = S.tkey[l];
                    if ((c > d) == up) { S.tkey[i] = d; S.tkey[l] = c; }
                }
            }
            // A thread's elements are tid, tid + 1024, ...: a wavefront owns aligned blocks of 64 elements.  While the
            // partner distance of this step and of the next one is below 64, every element a wavefront touches belongs to
            // it: its DS operations execute in order and no workgroup barrier is needed (41 of the 55 steps of P = 1024).
            const int jn = j > 1 ? (j >> 1) : k;            // partner distance of the next step
            if (j >= 64 || jn >= 64) __syncthreads();
            else __builtin_amdgcn_wave_barrier();
        }
    for (int i = tid; i < P; i += NT) {   // sizes of the valid prefixes; group heads of the query side
        if (S.qkey[i] != ~0ull) {
            if (i + 1 == P || S.qkey[i + 1] == ~0ull) S.nq = i + 1;
            if (i == 0 || (uint32_t)(S.qkey[i - 1] >> 32) != (uint32_t)(S.qkey[i] >> 32))
                S.ghead[atomicAdd(&S.ngroups, 1)] = (unsigned short)i;
        }
        if (S.tkey[i] != ~0ull) {
            if (i + 1 == P || S.tkey[i + 1] == ~0ull) S.nt = i + 1;
            S.matched[i] = (unsigned char)(blocked2 ? blocked2[(uint32_t)S.tkey[i]] != 0 : 0);
        }
    }
    __syncthreads();
    const int nq = S.nq, nt = S.nt, ng = S.ngroups;
    // ranges of every group by one thread each (three binary searches that the wavefront replaying the group would
    // otherwise run one after the other, 30 dependent LDS reads per group on the serial path)
    for (int g = tid; g < ng; g += NT) {
        const unsigned long long nodekey = S.qkey[S.ghead[g]] & 0xffffffff00000000ull;
        S.gt0[g] = (unsigned short)lower_bound_u64(S.tkey, nt, nodekey);
        S.gt1[g] = (unsigned short)lower_bound_u64(S.tkey, nt, nodekey + (1ull << 32));
        S.gq1[g] = (unsigned short)lower_bound_u64(S.qkey, nq, nodekey + (1ull << 32));
    }
    __syncthreads();
    // the wavefronts take the groups from a counter, not round-robin: groups differ in length
    for (;;) {
        int g = 0;
        if (lane == 0) g = atomicAdd(&S.next_group, 1);
        g = __builtin_amdgcn_readfirstlane(g);
        if (g >= ng) break;
        const int q_begin = S.ghead[g];
        const int t_begin = S.gt0[g], t_end = S.gt1[g];
        if (t_end <= t_begin) continue;   // node absent from the frame
        const int q_end = S.gq1[g];
        const int tc = t_end - t_begin;
        uint32_t td0[8];
        const bool have0 = lane < tc;
        {
            const uint32_t *tp = reinterpret_cast<const uint32_t *>(td + (size_t)(have0 ? (uint32_t)S.tkey[t_begin + lane] : 0) * 32);
#pragma unroll
            for (int i = 0; i < 8; ++i) td0[i] = have0 ? tp[i] : 0u;
        }
        bool used0 = have0 ? S.matched[t_begin + lane] != 0 : true;
        uint32_t *qs = S.qstage[wave];
        for (int qi = q_begin; qi < q_end; ++qi) {
            // The queries of a node are replayed one after the other (each may take a slot from the next), but their
            // descriptors do not depend on that: the wavefront fetches them kBowStage at a time -- lanes = dwords, a few
            // load instructions, ONE memory round trip -- into its LDS stage instead of waiting for a 32-byte load per query.
            const int qo = (qi - q_begin) & (kBowStage - 1);
            if (qo == 0) {
                const int nst = min(kBowStage, q_end - qi);
                __builtin_amdgcn_wave_barrier();          // the previous chunk's reads are done (DS operations execute in order)
                for (int e = lane; e < nst * 8; e += 64)
                    qs[e] = reinterpret_cast<const uint32_t *>(qd + (size_t)(uint32_t)S.qkey[qi + (e >> 3)] * 32)[e & 7];
                __builtin_amdgcn_wave_barrier();
            }
            uint32_t qdw[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) qdw[i] = qs[qo * 8 + i];
            // key = distance << 12 | slot (distances <= 256, slots < kBowPairMax = 4096): a wave minimum is six v_min on the
            // DPP path instead of six 64-bit compare-select steps, twice per query
            constexpr int kNone = 0x7fffffff;
            int k1 = kNone, k2 = kNone;
            if (!used0) k1 = (hamming256(qdw, td0) << 12) | (t_begin + lane);
            for (int c = t_begin + 64 + lane; c < t_end; c += 64) {
                if (S.matched[c]) continue;
                const uint32_t *tp = reinterpret_cast<const uint32_t *>(td + (size_t)(uint32_t)S.tkey[c] * 32);
                uint32_t t8[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) t8[i] = tp[i];
                const int k = (hamming256(qdw, t8) << 12) | c;
                if (k < k1) { k2 = k1; k1 = k; } else if (k < k2) k2 = k;
            }
            const int m1 = wave_min(k1);
            if (m1 == kNone) continue;
            const int m2 = wave_min(k1 == m1 ? k2 : k1);
            const int bestDist1 = m1 >> 12, bestDist2 = m2 == kNone ? 256 : m2 >> 12;
            if (bestDist1 <= max_dist && (float)bestDist1 < __fmul_rn(nnratio, (float)bestDist2)) {
                const int slot = m1 & 4095;
                if (((slot - t_begin) & 63) == lane) {
                    S.matched[slot] = 1;
                    if (slot - t_begin < 64) used0 = true;
                }
                if (lane == 0) S.mout[qi] = (int)(uint32_t)S.tkey[slot];
            }
        }
    }
    __syncthreads();
    // rotation bins of the accepted matches, all at once: the two angle loads per match used to sit in the replay loop, one
    // dependent memory round trip per accepted query on a single lane
    if (check_ori) {
        for (int p = tid; p < nq; p += NT) {
            const int idx2 = S.mout[p];
            if (idx2 < 0) continue;
            const int b = rot_bin(qk[(uint32_t)S.qkey[p]].angle, tk[idx2].angle);
            if (b >= 0) { atomicAdd(&S.hist[b], 1); S.bin[p] = (unsigned char)b; }
        }
        __syncthreads();
    }
    int ind1 = -1, ind2 = -1, ind3 = -1;
    if (check_ori) three_maxima(S.hist, ind1, ind2, ind3);
    int cnt = 0;
    for (int p = tid; p < nq; p += NT) {
        const int mval = S.mout[p];
        if (mval < 0) continue;
        const int b = S.bin[p];
        if (check_ori && b != ind1 && b != ind2 && b != ind3) continue;
        matches12[(uint32_t)S.qkey[p]] = mval;
        ++cnt;
    }
    if (cnt) atomicAdd(&S.nmatch, cnt);
    __syncthreads();
    if (tid == 0) nmatches[pair] = S.nmatch;
}

// rotation-consistency cull + count (ORBmatcher.cc:271-285 / :633-651); one workgroup
__global__ __launch_bounds__(1024) void k_bow_cull(int nq, int check_ori, const int *__restrict__ hist,
                                                   const uint8_t *__restrict__ bin, int *__restrict__ match, int *out_n)
{
    __shared__ int s_hist[HISTO_LENGTH];
    __shared__ int s_n;
    if (threadIdx.x < HISTO_LENGTH) s_hist[threadIdx.x] = hist[threadIdx.x];
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    int ind1 = -1, ind2 = -1, ind3 = -1;
    if (check_ori) three_maxima(s_hist, ind1, ind2, ind3);
    int cntl = 0;
    for (int i = threadIdx.x; i < nq; i += blockDim.x) {
        if (match[i] < 0) continue;
        const int b = bin[i];
        if (check_ori && b != ind1 && b != ind2 && b != ind3) match[i] = -1;
        else ++cntl;
    }
    if (cntl) atomicAdd(&s_n, cntl);
    __syncthreads();
    if (threadIdx.x == 0) *out_n = s_n;
}

// SearchForTriangulation: one wavefront per query; the epipole and epipolar-line gates are applied here and only the
// winner (smallest distance, LAST index on ties, ":735 dist>bestDist") is kept.  The reference never sets
// vbMatched2, so queries are independent; k_resolve_par mode 4 then only does the rotation cull.
struct TriParams {
    float f12[9];
    float ex, ey;
    float sigma2[ORBHIP_MAX_LEVELS], sf[ORBHIP_MAX_LEVELS];
};

__global__ __launch_bounds__(256) void k_tri_search(DevFrame F, const uint32_t *__restrict__ tnode,
                                                    const uint8_t *__restrict__ tvalid,
                                                    const orbhip_query *__restrict__ q,
                                                    const uint8_t *__restrict__ qdesc, int nq,
                                                    unsigned long long *__restrict__ cand, int *__restrict__ cnt,
                                                    int stride, TriParams P)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int qi = blockIdx.x * 4 + wv;
    if (qi >= nq) return;
    const orbhip_query Q = q[qi];
    const uint32_t qnode = (uint32_t)Q.level_aux;
    uint32_t qd[8];
    const uint32_t *qp = reinterpret_cast<const uint32_t *>(qdesc + (size_t)qi * 32);
#pragma unroll
    for (int i = 0; i < 8; ++i) qd[i] = qp[i];
    const bool stereo1 = Q.ur >= 0;
    // epipolar line of the query in image 2, ORBmatcher.cc:143-145
    const float la = __fadd_rn(__fadd_rn(__fmul_rn(Q.u, P.f12[0]), __fmul_rn(Q.v, P.f12[3])), P.f12[6]);
    const float lb = __fadd_rn(__fadd_rn(__fmul_rn(Q.u, P.f12[1]), __fmul_rn(Q.v, P.f12[4])), P.f12[7]);
    const float lc = __fadd_rn(__fadd_rn(__fmul_rn(Q.u, P.f12[2]), __fmul_rn(Q.v, P.f12[5])), P.f12[8]);
    const float den = __fadd_rn(__fmul_rn(la, la), __fmul_rn(lb, lb));
    unsigned long long best = ~0ull;
    for (int j = lane; j < F.n; j += 64) {
        if (tnode[j] != qnode || (tvalid && !tvalid[j])) continue;
        const uint32_t *tp = reinterpret_cast<const uint32_t *>(F.desc + (size_t)j * 32);
        uint32_t td[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) td[i] = tp[i];
        const int dist = hamming256(qd, td);
        if (dist > TH_LOW) continue;
        const orbhip_keypoint kp = F.keys[j];
        const bool stereo2 = F.u_right && F.u_right[j] >= 0;
        if (!stereo1 && !stereo2) {   // :741-747
            const float dx = __fsub_rn(P.ex, kp.x), dy = __fsub_rn(P.ey, kp.y);
            if (__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)) < __fmul_rn(100.f, P.sf[kp.octave])) continue;
        }
        // CheckDistEpipolarLine :147-156
        const float num = __fadd_rn(__fadd_rn(__fmul_rn(la, kp.x), __fmul_rn(lb, kp.y)), lc);
        if (den == 0) continue;
        const float dsqr = __fdiv_rn(__fmul_rn(num, num), den);
        if (!((double)dsqr < 3.84 * (double)P.sigma2[kp.octave])) continue;
        const unsigned long long k = ((unsigned long long)dist << 32) | (uint32_t)(0xfffff - j);
        best = k < best ? k : best;
    }
    best = wave_min_u64(best);
    if (lane == 0) {
        if (best == ~0ull) cnt[qi] = 0;
        else {
            cand[(size_t)qi * stride] = (best & 0xffffffff00000000ull) | (uint32_t)(0xfffff - (int)(best & 0xfffffu));
            cnt[qi] = 1;
        }
    }
}

// ---- independent best match per query (no slot blocking) ------------------------------------
// Inner loop of ORBmatcher::Fuse (ORBmatcher.cc:893-950, :1045-1075) and of both directions of SearchBySim3
// (:1199-1219, :1279-1299): GetFeaturesInArea window, level window, optional chi-square gate on the
// reprojection error (Fuse: 5.99 mono / 7.8 stereo, scaled by mvInvLevelSigma2[level]), Hamming argmin with the
// reference's first-minimum tie-break.  One wavefront per query.
struct SigmaTab { float inv_sigma2[ORBHIP_MAX_LEVELS]; };

__global__ __launch_bounds__(256) void k_best_in_window(DevFrame F, const uint32_t *__restrict__ ord,
                                                        const orbhip_query *__restrict__ q,
                                                        const uint8_t *__restrict__ qdesc, int nq, int chi2_gate,
                                                        SigmaTab sig, int *__restrict__ best_idx,
                                                        int *__restrict__ best_dist)
{
    const int lane = threadIdx.x & 63;
    const int qi = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (qi >= nq) return;
    const orbhip_query Q = q[qi];
    unsigned long long best = ~0ull;
    if (Q.valid) {
        const float x = Q.u, y = Q.v, r = Q.radius;
        const int nMinCellX = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(x, F.min_x), r), F.inv_w)));
        const int nMaxCellX = min(GRID_COLS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(x, F.min_x), r), F.inv_w)));
        const int nMinCellY = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(y, F.min_y), r), F.inv_h)));
        const int nMaxCellY = min(GRID_ROWS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(y, F.min_y), r), F.inv_h)));
        if (!(nMinCellX >= GRID_COLS || nMaxCellX < 0 || nMinCellY >= GRID_ROWS || nMaxCellY < 0)) {
            uint32_t qd[8];
            const uint32_t *qp = reinterpret_cast<const uint32_t *>(qdesc + (size_t)qi * 32);
#pragma unroll
            for (int i = 0; i < 8; ++i) qd[i] = qp[i];
            for (int j0 = 0; j0 < F.n; j0 += 64) {
                const int j = j0 + lane;
                if (j >= F.n) continue;
                const uint32_t o = ord[j];
                if (o == kNoCell) continue;
                const int cell = (int)(o >> 20);
                const int px = cell / GRID_ROWS, py = cell - px * GRID_ROWS;
                if (!(px >= nMinCellX && px <= nMaxCellX && py >= nMinCellY && py <= nMaxCellY)) continue;
                const orbhip_keypoint kp = F.keys[j];
                if (!(fabsf(__fsub_rn(kp.x, x)) < r && fabsf(__fsub_rn(kp.y, y)) < r)) continue;
                if (kp.octave < Q.min_level || kp.octave > Q.max_level) continue;   // kpLevel<pred-1 || kpLevel>pred
                if (chi2_gate) {
                    const float ex = __fsub_rn(x, kp.x), ey = __fsub_rn(y, kp.y);
                    float e2 = __fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey));
                    const float kpr = F.u_right ? F.u_right[j] : -1.0f;
                    const float is2 = sig.inv_sigma2[kp.octave];
                    if (kpr >= 0) {
                        const float er = __fsub_rn(Q.ur, kpr);
                        e2 = __fadd_rn(e2, __fmul_rn(er, er));
                        if ((double)__fmul_rn(e2, is2) > 7.8) continue;
                    } else if ((double)__fmul_rn(e2, is2) > 5.99) continue;
                }
                const uint32_t *tp = reinterpret_cast<const uint32_t *>(F.desc + (size_t)j * 32);
                uint32_t td[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) td[i] = tp[i];
                const unsigned long long key = ((unsigned long long)hamming256(qd, td) << 32) | o;
                best = key < best ? key : best;
            }
        }
    }
    best = wave_min_u64(best);
    if (lane == 0) {
        best_idx[qi] = best == ~0ull ? -1 : (int)(best & 0xfffffu);
        best_dist[qi] = best == ~0ull ? 256 : (int)(best >> 32);
    }
}

// ---- resolve ---------------------------------------------------------------------------
// mode 0: SearchByProjection(Frame,Frame)  ORBmatcher.cc:1397-1467
// mode 1: SearchByProjection(Frame,points) ORBmatcher.cc:76-125
// mode 2: SearchForInitialization          ORBmatcher.cc:432-511
struct ResolveShared {
    unsigned short block[kResolveMax];   // mode 0/1: slot taken (0/1); mode 2: vMatchedDistance (0xffff = INT_MAX)
    int assign[kResolveMax];             // mode 0/1: assign[]; mode 2: vnMatches21
    int m12[kResolveMax];                // mode 2: vnMatches12
    unsigned char evbin[kResolveMax];    // rotation-histogram bin of the event of query i (0xff none)
    unsigned short evidx[kResolveMax];   // mode 0: bestIdx2 pushed into rotHist
    // read-only operands staged once so that the serial loop never waits on HBM
    float t_angle[kResolveMax];          // train keypoint angle
    float q_angle[kResolveMax];          // query keypoint angle
    int q_cnt[kResolveMax];              // candidate count per query (sign = unsorted)
    unsigned char t_oct[kResolveMax];    // train keypoint octave
    unsigned char q_obs[kResolveMax];    // query map point observed
    int hist[HISTO_LENGTH];
};

__device__ __forceinline__ int rot_bin(float a1, float a2)
{
    const float factor = 1.0f / HISTO_LENGTH;
    float rot = __fsub_rn(a1, a2);
    if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
    int bin = (int)roundf(__fmul_rn(rot, factor));
    if (bin == HISTO_LENGTH) bin = 0;
    // keypoint angles are caller data: outside [0, 360) (or NaN) the reference trips its assert(bin>=0 && bin<HISTO_LENGTH);
    // here such a match simply takes no part in the rotation histogram (-1) instead of writing outside it
    return (bin >= 0 && bin < HISTO_LENGTH) ? bin : -1;
}

// First / second usable candidate of a query.  `usable(idx, dist)` is evaluated by every lane.
template <class Usable>
__device__ __forceinline__ void pick2(const unsigned long long *list, unsigned long long v, int c, int lane, Usable usable,
                                      unsigned long long &k1, unsigned long long &k2)
{
    k1 = ~0ull; k2 = ~0ull;
    if (c > 0) {  // sorted, <= 64; `v` = list[lane], prefetched by the caller
        bool u = lane < c && usable((int)(v & 0xfffffu), (int)(v >> 32));
        unsigned long long bal = __ballot(u);
        if (bal) {   // the winning lanes are wave-uniform: v_readlane, not a ds_bpermute round trip
            const int l1 = __ffsll((long long)bal) - 1;
            k1 = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(v >> 32), l1) << 32) |
                 (unsigned)__builtin_amdgcn_readlane((int)v, l1);
            bal &= bal - 1;
            if (bal) {
                const int l2 = __ffsll((long long)bal) - 1;
                k2 = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(v >> 32), l2) << 32) |
                     (unsigned)__builtin_amdgcn_readlane((int)v, l2);
            }
        }
    } else if (c < 0) {  // unsorted: two masked min-reductions
        c = -c;
        unsigned long long m1 = ~0ull;
        for (int i = lane; i < c; i += 64) {
            unsigned long long v = list[i];
            if (usable((int)(v & 0xfffffu), (int)(v >> 32))) m1 = v < m1 ? v : m1;
        }
        k1 = wave_min_u64(m1);
        unsigned long long m2 = ~0ull;
        for (int i = lane; i < c; i += 64) {
            unsigned long long v = list[i];
            if (v != k1 && usable((int)(v & 0xfffffu), (int)(v >> 32))) m2 = v < m2 ? v : m2;
        }
        k2 = wave_min_u64(m2);
    }
}

__device__ __forceinline__ void three_maxima(const int *h, int &ind1, int &ind2, int &ind3)
{
    // ComputeThreeMaxima (src/ORBmatcher.cc:1601-1645) as selects on values: with the three indices passed by reference
    // through a chain of branches the compiler kept them in scratch memory (150 scratch accesses per call)
    int max1 = 0, max2 = 0, max3 = 0, i1 = -1, i2 = -1, i3 = -1;
    int hv[HISTO_LENGTH];   // all bins first (independent loads, one wait), then the scan on registers
#pragma unroll
    for (int i = 0; i < HISTO_LENGTH; ++i) hv[i] = h[i];
#pragma unroll
    for (int i = 0; i < HISTO_LENGTH; ++i) {
        const int s = hv[i];
        const bool g1 = s > max1, g2 = s > max2, g3 = s > max3;
        max3 = g2 ? max2 : (g3 ? s : max3);  i3 = g2 ? i2 : (g3 ? i : i3);
        max2 = g1 ? max1 : (g2 ? s : max2);  i2 = g1 ? i1 : (g2 ? i : i2);
        max1 = g1 ? s : max1;                i1 = g1 ? i : i1;
    }
    if ((float)max2 < __fmul_rn(0.1f, (float)max1)) { i2 = -1; i3 = -1; }
    else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) { i3 = -1; }
    ind1 = i1; ind2 = i2; ind3 = i3;
}

__global__ __launch_bounds__(64) void k_resolve(int mode, DevFrame F, const orbhip_keypoint *__restrict__ qkeys,
                                                const orbhip_query *__restrict__ q, int nq,
                                                const unsigned long long *__restrict__ cand,
                                                const unsigned long long *__restrict__ ccand,
                                                const int *__restrict__ cnt, int stride,
                                                const uint8_t *__restrict__ taken_in, float nnratio,
                                                int check_ori, int *__restrict__ out, int *__restrict__ out_n, Batch B)
{
    extern __shared__ unsigned char resolve_lds[];
    ResolveShared &S = *reinterpret_cast<ResolveShared *>(resolve_lds);
    const int lane = threadIdx.x;
    if (B.qcap > 0) {   // batched (device-resident) call: one wavefront per pair
        const int pair = blockIdx.x;
        batch_frame(F, B, pair);
        if (qkeys) qkeys += (size_t)(B.qd0 + pair * B.qds) * B.qcap;
        q += (size_t)pair * B.qcap;
        cand += (size_t)pair * B.qcap * stride;
        ccand += (size_t)pair * B.qcap * kCompact;
        cnt += (size_t)pair * B.qcap;
        if (taken_in) taken_in += (size_t)pair * B.cap;
        out += (size_t)pair * (mode == 2 ? B.qcap : B.cap);
        out_n += pair;
        if (B.nq_dev) nq = min(B.nq_dev[pair], B.qcap);
    }
    const int n = F.n;
    for (int i = lane; i < n; i += 64) {
        S.block[i] = (mode == 2) ? 0xffff : (unsigned short)(taken_in ? taken_in[i] != 0 : 0);
        S.assign[i] = -1;
        S.t_angle[i] = F.keys[i].angle;
        S.t_oct[i] = (unsigned char)F.keys[i].octave;
    }
    for (int i = lane; i < nq; i += 64) {
        S.evbin[i] = 0xff;
        if (mode == 2) S.m12[i] = -1;
        S.q_cnt[i] = cnt[i];
        S.q_angle[i] = (mode == 2) ? qkeys[i].angle : q[i].angle;
        S.q_obs[i] = (unsigned char)(mode == 2 ? 0 : (q[i].observed != 0));
    }
    if (lane < HISTO_LENGTH) S.hist[lane] = 0;
    __syncthreads();
    int nmatches = 0;
    // software prefetch: the list head of query i+1 is in flight while query i is resolved
    int c_next = nq > 0 ? S.q_cnt[0] : 0;
    unsigned long long v_next = (c_next > 0 && lane < c_next) ? ccand[lane] : ~0ull;   // sorted lists live in the compact array
    for (int i = 0; i < nq; ++i) {
        const int c = c_next;
        const unsigned long long v = v_next;
        if (i + 1 < nq) {
            c_next = S.q_cnt[i + 1];
            v_next = (c_next > 0 && lane < c_next) ? ccand[(size_t)(i + 1) * kCompact + lane] : ~0ull;
        }
        if (c == 0) continue;
        const unsigned long long *list = cand + (size_t)i * stride;
        unsigned long long k1, k2;
        if (mode == 2) pick2(list, v, c, lane, [&](int idx, int dist) { return !((int)S.block[idx] <= dist); }, k1, k2);
        else pick2(list, v, c, lane, [&](int idx, int) { return S.block[idx] == 0; }, k1, k2);
        if (k1 == ~0ull) continue;
        const int bestDist = (int)(k1 >> 32), bestIdx = (int)(k1 & 0xfffffu);
        if (mode == 0) {
            if (bestDist <= TH_HIGH) {
                if (lane == 0) {
                    S.assign[bestIdx] = i;
                    S.block[bestIdx] = (unsigned short)S.q_obs[i];
                    if (check_ori) {
                        int bin = rot_bin(S.q_angle[i], S.t_angle[bestIdx]);
                        if (bin >= 0) {
                            S.hist[bin]++;
                            S.evbin[i] = (unsigned char)bin;
                            S.evidx[i] = (unsigned short)bestIdx;
                        }
                    }
                }
                nmatches++;
            }
        } else if (mode == 1) {
            if (bestDist <= TH_HIGH) {
                const int bestDist2 = k2 == ~0ull ? 256 : (int)(k2 >> 32);
                const int bestLevel = S.t_oct[bestIdx];
                const int bestLevel2 = k2 == ~0ull ? -1 : (int)S.t_oct[(int)(k2 & 0xfffffu)];
                if (!(bestLevel == bestLevel2 && (float)bestDist > __fmul_rn(nnratio, (float)bestDist2))) {
                    if (lane == 0) {
                        S.assign[bestIdx] = i;
                        S.block[bestIdx] = (unsigned short)S.q_obs[i];
                    }
                    nmatches++;
                }
            }
        } else {
            if (bestDist <= TH_LOW) {
                // bestDist < (float)bestDist2 * mfNNratio with bestDist2 = INT_MAX when absent
                const float d2 = k2 == ~0ull ? (float)INT_MAX : (float)(int)(k2 >> 32);
                if ((float)bestDist < __fmul_rn(d2, nnratio)) {
                    const int prev = S.assign[bestIdx];  // vnMatches21
                    if (prev >= 0) nmatches--;
                    if (lane == 0) {
                        if (prev >= 0) S.m12[prev] = -1;
                        S.m12[i] = bestIdx;
                        S.assign[bestIdx] = i;
                        S.block[bestIdx] = (unsigned short)bestDist;
                        if (check_ori) {
                            int bin = rot_bin(S.q_angle[i], S.t_angle[bestIdx]);
                            if (bin >= 0) { S.hist[bin]++; S.evbin[i] = (unsigned char)bin; }
                        }
                    }
                    nmatches++;
                }
            }
        }
        // single wavefront: DS operations execute in issue order, so the next query's LDS reads see lane 0's
        // writes; no s_barrier / vmcnt(0) here -- it would drain the prefetch of the next list head
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    if (check_ori && mode != 1) {
        int ind1, ind2, ind3;
        three_maxima(S.hist, ind1, ind2, ind3);
        // events are independent of each other: every one in a losing bin clears its slot
        int dec = 0;
        if (mode == 0) {
            for (int i = lane; i < nq; i += 64) {
                int b = S.evbin[i];
                if (b != 0xff && b != ind1 && b != ind2 && b != ind3) { S.assign[S.evidx[i]] = -1; dec++; }
            }
        } else {
            for (int i = lane; i < nq; i += 64) {
                int b = S.evbin[i];
                if (b != 0xff && b != ind1 && b != ind2 && b != ind3 && S.m12[i] >= 0) { S.m12[i] = -1; dec++; }
            }
        }
        nmatches -= wave_reduce_add_i(dec);
    }
    __syncthreads();
    if (mode == 2) for (int i = lane; i < nq; i += 64) out[i] = S.m12[i];
    else for (int i = lane; i < n; i += 64) out[i] = S.assign[i];
    if (lane == 0) *out_n = nmatches;
}

// ---- parallel resolve: modes 0 / 1 (SearchByProjection overloads) and 4 (SearchForTriangulation: one pre-gated
// candidate per query, no blocking, rotation cull, output per query) ---------------------------------------------
// The sequential reference loop hands every query, in index order, the first candidate of its list (sorted by
// (distance, visiting order)) that is neither taken on entry nor already held by an accepted, observed query with a
// smaller index.  That is serial dictatorship.
//  * Modes 0 / 4 (frame-to-frame search, triangulation): acceptance depends on the first usable entry alone, so nobody
//    ever gives a held slot up, and the outcome is the unique stable matching of "queries prefer list order, slots
//    prefer the smaller query index", which deferred acceptance reaches from any proposal order: unsettled queries
//    propose (atomicMin on the slot's holder) to the first entry no smaller query holds; a holder only ever gets
//    smaller, so each query's list cursor is monotone and the total walk is the list length, not list length x rounds.
//  * Mode 1 (map points): the ratio test applies only when best and second best share a level, so losing the second
//    candidate to a smaller query can turn an acceptance into a rejection -- holders are not monotone.  There the
//    rounds re-pick every query against the previous round's claims and rebuild the claims (fixed-point iteration).
// One workgroup per pair, state in LDS.
// State of the parallel resolve.  Up to kResolveMax train keypoints and queries it lives in LDS (GS = false); beyond
// that the same arrays are carved out of an HBM workspace (GS = true: same code, global loads / atomics).
struct ResolveParState {
    int *owner[2];                       // [0]: holder of every slot (smallest accepted, observed proposer); [1]: cursors
    int *choice;                         // slot picked by query i, -1 if none accepted
    float *t_angle, *q_angle;
    unsigned char *t_oct, *q_obs, *taken, *evbin;
    int *hist;                           // HISTO_LENGTH bins
    int *vars;                           // [0], [3], [4]: rotating "a choice changed" flags; [1] accepted; [2] culled
    unsigned short *cur1, *cur2;         // per query: first list entry not known to be unavailable (best / second best)
    int *q_cnt;                          // candidates per query (sign = unsorted)
    unsigned short *wl[2];               // work lists of the event-driven rounds (queries to step next)
    uint32_t *lc;                        // first lcn entries of every query's sorted list as (distance << 20 | slot), row
    int lcn;                             // stride lcn + 1: the steps re-read list heads and must not wait for HBM
};
constexpr int kResolveLdsBudget = 156 * 1024;   // dynamic LDS the LDS variant may ask for (160 KB per CU)
constexpr int kResolveHead = 8;
__host__ __device__ inline size_t resolve_par_bytes(size_t n, size_t nq, size_t lcn = 0)
{
    n = (n + 3) & ~(size_t)3; nq = (nq + 3) & ~(size_t)3;
    return (2 * n + nq + n + nq) * 4 + 2 * n + 2 * nq + (HISTO_LENGTH + 2 + 16) * 4 + 12 * nq + (lcn ? nq * (lcn + 1) * 4 : 0);
}
__device__ __forceinline__ void resolve_par_carve(ResolveParState &S, unsigned char *base, size_t n, size_t nq, int lcn = 0)
{
    n = (n + 3) & ~(size_t)3; nq = (nq + 3) & ~(size_t)3;
    S.lcn = lcn;
    S.lc = reinterpret_cast<uint32_t *>(base + resolve_par_bytes(n, nq, 0));
    int *p = reinterpret_cast<int *>(base);
    S.owner[0] = p; p += n; S.owner[1] = p; p += n; S.choice = p; p += nq;
    S.t_angle = reinterpret_cast<float *>(p); p += n; S.q_angle = reinterpret_cast<float *>(p); p += nq;
    S.hist = p; p += HISTO_LENGTH + 2; S.vars = p; p += 16;
    S.cur1 = reinterpret_cast<unsigned short *>(p); S.cur2 = S.cur1 + nq; p += nq;
    S.wl[0] = reinterpret_cast<unsigned short *>(p); S.wl[1] = S.wl[0] + nq; p += nq;
    S.q_cnt = p; p += nq;
    unsigned char *c = reinterpret_cast<unsigned char *>(p);
    S.t_oct = c; c += n; S.taken = c; c += n; S.q_obs = c; c += nq; S.evbin = c;
}

#ifdef ORBHIP_DEVTOOLS
__device__ unsigned int g_resolve_stats[4];   // development builds: {launched workgroups, sum of rounds, max rounds, -}
#endif
template <bool GS>
__global__ __launch_bounds__(1024) void k_resolve_par(int mode, DevFrame F, const orbhip_query *__restrict__ q, int nq,
                                                      const unsigned long long *__restrict__ cand,
                                                      const unsigned long long *__restrict__ ccand,
                                                      const int *__restrict__ cnt, int stride,
                                                      const uint8_t *__restrict__ taken_in, float nnratio,
                                                      int check_ori, int *__restrict__ out, int *__restrict__ out_n, Batch B,
                                                      int th_accept, int all_block, unsigned char *__restrict__ gstate,
                                                      size_t gstate_stride, int n_alloc, int nq_alloc, int lcn)
{
    extern __shared__ unsigned char resolve_lds[];
    const int tid = threadIdx.x, T = blockDim.x;
    {
        const int pair = blockIdx.x;
        batch_frame(F, B, pair);
        q += (size_t)pair * B.qcap;
        cand += (size_t)pair * B.qcap * stride;
        if (ccand) ccand += (size_t)pair * B.qcap * kCompact;
        cnt += (size_t)pair * B.qcap;
        if (taken_in) taken_in += (size_t)pair * B.cap;
        out += (size_t)pair * B.cap;
        out_n += pair;
        if (B.nq_dev) nq = min(B.nq_dev[pair], B.qcap);
    }
    const int n = F.n;
    ResolveParState S;
    if (GS) resolve_par_carve(S, gstate + (size_t)blockIdx.x * gstate_stride, (size_t)n, (size_t)nq);
    else resolve_par_carve(S, resolve_lds, (size_t)n_alloc, (size_t)nq_alloc, lcn);
    int *holder = S.owner[0];
    unsigned short *cur1 = S.cur1, *cur2 = S.cur2;
    // The list heads of this thread's first query are requested before anything else, without waiting for the list
    // length (a compact list always has its 64 entries allocated), so that ONE memory round trip covers them and the
    // per-frame state below; heads and lengths then live in LDS: the rounds must never wait for HBM.
    const bool lc_on = !GS && S.lcn > 0;
    unsigned long long hv[kResolveHead];
    int head_c = 0;
    if (tid < nq) {
        head_c = cnt[tid];
        if (lc_on && ccand) {
            const unsigned long long *l0 = ccand + (size_t)tid * kCompact;
#pragma unroll
            for (int e = 0; e < kResolveHead; ++e) hv[e] = l0[e];
        }
    }
    for (int i = tid; i < n; i += T) {
        holder[i] = INT_MAX;
        S.taken[i] = (unsigned char)(taken_in ? taken_in[i] != 0 : 0);
        S.t_angle[i] = F.keys[i].angle;
        S.t_oct[i] = (unsigned char)F.keys[i].octave;
    }
    for (int i = tid; i < nq; i += T) {
        S.choice[i] = -2;   // "not evaluated yet"
        S.q_angle[i] = q[i].angle;
        S.q_obs[i] = (unsigned char)(all_block || q[i].observed != 0);
        S.evbin[i] = 0xff;
        cur1[i] = 0; cur2[i] = 0;
        S.q_cnt[i] = i == tid ? head_c : cnt[i];
    }
    if (tid < HISTO_LENGTH) S.hist[tid] = 0;
    if (tid < 16) S.vars[tid] = 0;
    if (lc_on) {
        for (int i = tid; i < nq; i += T) {
            const int c = S.q_cnt[i];
            if (c == 0) continue;
            if (c < 0) {   // unsorted list: the window search left the head of its sorted order in the compact row
                if (ccand && S.lcn >= kHeadMax) {
                    const unsigned long long *hrow = ccand + (size_t)i * kCompact;
#pragma unroll
                    for (int e = 0; e < kHeadMax; ++e) {
                        const unsigned long long h = hrow[e];
                        S.lc[i * (S.lcn + 1) + e] = ((uint32_t)(h >> 32) << 20) | (uint32_t)(h & 0xfffffu);
                    }
                    S.lc[i * (S.lcn + 1) + S.lcn] = (uint32_t)hrow[kHeadMax];
                }
                continue;
            }
            if (i != tid || !ccand) {
                const unsigned long long *l0 = ccand ? ccand + (size_t)i * kCompact : cand + (size_t)i * stride;
#pragma unroll
                for (int e = 0; e < kResolveHead; ++e) hv[e] = (e < c && e < S.lcn) ? l0[e] : ~0ull;
            }
#pragma unroll
            for (int e = 0; e < kResolveHead; ++e)
                if (e < c && e < S.lcn) S.lc[i * (S.lcn + 1) + e] = ((uint32_t)(hv[e] >> 32) << 20) | (uint32_t)(hv[e] & 0xfffffu);
        }
    }
    __syncthreads();
    // One deferred-acceptance step of query i against the holders as they are right now (any interleaving of proposals
    // is a valid execution, the atomics are the only synchronisation the matching needs).  Returns "the choice changed".
    auto held_by_smaller = [&](int idx, int i) -> bool {
        return S.taken[idx] || __hip_atomic_load(&holder[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < i;
    };
    unsigned short *wl_next = nullptr;   // set by the event-driven rounds
    int *wl_count = nullptr;
    int *claim = holder;                 // where an accepted, observed query files its claim
    bool claims_rebuilt = false;         // mode 1: the claims are rebuilt from scratch every round
    // pick: best (and, mode 1, second best) available entry of query i's list.  A sorted list (<= 64 candidates) is walked
    // by the thread that owns the query; an unsorted one (more candidates: wide windows) by the whole wavefront, lanes over
    // the entries (step_group below).
    auto pick_sorted = [&](int i, int c, unsigned long long &k1, unsigned long long &k2) {
        const unsigned long long *list = ccand ? ccand + (size_t)i * kCompact : cand + (size_t)i * stride;
        const uint32_t *lrow = S.lc + i * (S.lcn + 1);
        auto entry = [&](int e) -> unsigned long long {
            if (e >= c) return ~0ull;
            if (!GS && e < S.lcn) { const uint32_t w = lrow[e]; return ((unsigned long long)(w >> 20) << 32) | (w & 0xfffffu); }
            return list[e];
        };
        // four entries per trip: their eight LDS reads (taken, holder) are in flight together
        int e1 = mode == 1 ? 0 : cur1[i];
        while (e1 < c) {
            const unsigned long long v0 = entry(e1), v1 = entry(e1 + 1), v2 = entry(e1 + 2), v3 = entry(e1 + 3);
            const bool a0 = !held_by_smaller((int)(v0 & 0xfffffu), i);
            const bool a1 = v1 != ~0ull && !held_by_smaller((int)(v1 & 0xfffffu), i);
            const bool a2 = v2 != ~0ull && !held_by_smaller((int)(v2 & 0xfffffu), i);
            const bool a3 = v3 != ~0ull && !held_by_smaller((int)(v3 & 0xfffffu), i);
            if (a0) { k1 = v0; break; }
            if (a1) { k1 = v1; e1 += 1; break; }
            if (a2) { k1 = v2; e1 += 2; break; }
            if (a3) { k1 = v3; e1 += 3; break; }
            e1 += 4;
        }
        e1 = min(e1, c);
        if (mode != 1) cur1[i] = (unsigned short)e1;
        if (mode == 1 && e1 < c) {
            for (int e2 = e1 + 1; e2 < c; ++e2) {
                const unsigned long long v = entry(e2);
                if (!held_by_smaller((int)(v & 0xfffffu), i)) { k2 = v; break; }
            }
        }
    };
    auto pick_unsorted_wave = [&](int i, int len, unsigned long long &k1, unsigned long long &k2) {   // every lane, same (i, len)
        const unsigned long long *list = cand + (size_t)i * stride;
        unsigned long long a1 = ~0ull, a2 = ~0ull;
        for (int e = (int)(threadIdx.x & 63); e < len; e += 64) {
            const unsigned long long v = list[e];
            if (held_by_smaller((int)(v & 0xfffffu), i)) continue;
            if (v < a1) { a2 = a1; a1 = v; } else if (v < a2) a2 = v;
        }
        k1 = wave_min_u64(a1);
        k2 = mode == 1 ? wave_min_u64(a1 == k1 ? a2 : a1) : ~0ull;   // keys are unique (the slot is part of the key)
    };
    auto commit = [&](int i, unsigned long long k1, unsigned long long k2) -> bool {
        int newc = -1;
        if (k1 != ~0ull) {
            const int bestDist = (int)(k1 >> 32), bestIdx = (int)(k1 & 0xfffffu);
            bool acc = bestDist <= th_accept;
            if (acc && mode == 1) {
                const int bestDist2 = k2 == ~0ull ? 256 : (int)(k2 >> 32);
                const int bestLevel = S.t_oct[bestIdx];
                const int bestLevel2 = k2 == ~0ull ? -1 : (int)S.t_oct[(int)(k2 & 0xfffffu)];
                if (bestLevel == bestLevel2 && (float)bestDist > __fmul_rn(nnratio, (float)bestDist2)) acc = false;
            }
            if (acc) newc = bestIdx;
        }
        const bool changed = newc != S.choice[i];
        if (!changed && !claims_rebuilt) return false;
        S.choice[i] = newc;
        if (newc >= 0 && S.q_obs[i]) {
            const int old = atomicMin(&claim[newc], i);
            // event-driven rounds: whoever loses the slot steps again -- me if a smaller query got there first, the
            // previous holder if I displaced it
            if (wl_next) {
                const int again = old < i ? i : (old != INT_MAX ? old : -1);
                if (again >= 0) wl_next[atomicAdd(wl_count, 1)] = (unsigned short)again;
            }
        }
        return changed;
    };
    // One deferred-acceptance / fixed-point step for the queries of a whole wavefront (lane: query i, `valid` false for
    // idle lanes; every lane of the wavefront must call).  Queries with unsorted lists are taken one after the other by
    // the whole wavefront -- a single thread re-scanning a 300-entry list in HBM on every step is what made wide windows
    // slow (th = 60: 293 us, th = 100: 556 us per 32 pairs).
    // Unsorted lists with a sorted head in LDS (the LDS variant with list heads; `head_on`): the owner thread walks the
    // head -- kHeadMax keys in list order, candidates taken on entry left out -- like a sorted list.  Frame search: the
    // cursor is monotone, and when the head runs out before the list does the wavefront refills it with the next keys
    // behind the last one (one scan of the list per kHeadMax steps instead of one per step).  Map-point search: first and
    // second free entry of the initial head; if the head cannot tell (not both found and the list goes on) the wavefront
    // scans the whole list.  Returns "needs the wavefront".
    const bool head_on = lc_on && S.lcn >= kHeadMax && ccand != nullptr;
    auto expand = [](uint32_t w) -> unsigned long long { return ((unsigned long long)(w >> 20) << 32) | (w & 0xfffffu); };
    auto head_walk = [&](int i, unsigned long long &k1, unsigned long long &k2) -> bool {
        const uint32_t *lrow = S.lc + i * (S.lcn + 1);
        const uint32_t meta = lrow[S.lcn];
        const int hl = (int)(meta & 0xffu);
        const bool complete = (meta & 0x100u) != 0;
        if (mode != 1) {
            int e = cur1[i];
            while (e < hl) {
                const uint32_t w = lrow[e];
                if (!held_by_smaller((int)(w & 0xfffffu), i)) { k1 = expand(w); break; }
                ++e;
            }
            cur1[i] = (unsigned short)e;
            return e >= hl && !complete;
        }
        k1 = k2 = ~0ull;
        for (int e = 0; e < hl; ++e) {
            const uint32_t w = lrow[e];
            if (held_by_smaller((int)(w & 0xfffffu), i)) continue;
            if (k1 == ~0ull) k1 = expand(w); else { k2 = expand(w); break; }
        }
        return k2 == ~0ull && !complete;
    };
    auto refill_head_wave = [&](int i, int len) {   // every lane, same (i, len): the next keys behind the head's last one
        const int lane = (int)(threadIdx.x & 63);
        uint32_t *lrow = S.lc + i * (S.lcn + 1);
        const int hl = (int)(lrow[S.lcn] & 0xffu);
        // the head's last key in full: the LDS copy holds (distance, slot); the cell bits of the key are the slot's grid
        // cell (PosInGrid, src/Frame.cc:382-392, the expressions of the grid build)
        unsigned long long cursor = 0ull;
        if (hl > 0) {
            const uint32_t w = lrow[hl - 1];
            const int slot = (int)(w & 0xfffffu);
            const orbhip_keypoint kp = F.keys[slot];
            const int px = (int)roundf(__fmul_rn(__fsub_rn(kp.x, F.min_x), F.inv_w));
            const int py = (int)roundf(__fmul_rn(__fsub_rn(kp.y, F.min_y), F.inv_h));
            cursor = ((unsigned long long)(w >> 20) << 32) | ((unsigned long long)(uint32_t)(px * GRID_ROWS + py) << 20) | (uint32_t)slot;
        }
        const unsigned long long *list = cand + (size_t)i * stride;
        unsigned long long a1 = ~0ull, a2 = ~0ull;
        int na = 0;
        for (int e = lane; e < len; e += 64) {
            const unsigned long long v = list[e];
            if ((hl > 0 && v <= cursor) || S.taken[(int)(v & 0xfffffu)]) continue;
            if (v < a1) { a2 = a1; a1 = v; } else if (v < a2) a2 = v;
            ++na;
        }
        unsigned long long head[kHeadMax];
        bool complete;
        const int nh = wave_sorted_head(a1, a2, na > 2, head, &complete);
        __builtin_amdgcn_wave_barrier();
        if (lane < kHeadMax) {
            unsigned long long v = head[0];
#pragma unroll
            for (int k = 1; k < kHeadMax; ++k) if (lane == k) v = head[k];
            lrow[lane] = ((uint32_t)(v >> 32) << 20) | (uint32_t)(v & 0xfffffu);
        }
        if (lane == 0) lrow[S.lcn] = (uint32_t)nh | (complete ? 0x100u : 0u);
        __builtin_amdgcn_wave_barrier();
    };
    // One deferred-acceptance / fixed-point step for the queries of a whole wavefront (lane: query i, `valid` false for
    // idle lanes; every lane of the wavefront must call).  What an unsorted list needs from the whole wavefront -- a
    // refill of its head, or a scan -- is done for one query after the other, lanes over the list entries: a single
    // thread re-scanning a 300-entry list in HBM on every step is what made wide windows slow (th = 60: 293 us,
    // th = 100: 556 us per 32 pairs).
    auto step_group = [&](int i, bool valid) -> bool {
        const int lane = (int)(threadIdx.x & 63);
        const int c = valid ? S.q_cnt[i] : 0;
        unsigned long long k1 = ~0ull, k2 = ~0ull;
        if (c > 0) pick_sorted(i, c, k1, k2);
        bool need = c < 0;
        if (need && head_on) need = head_walk(i, k1, k2);
        if (head_on && mode != 1) {
            for (;;) {
                unsigned long long longs = __ballot(need);
                if (!longs) break;
                while (longs) {
                    const int src = __ffsll((long long)longs) - 1;
                    refill_head_wave(__builtin_amdgcn_readlane(i, src), -__builtin_amdgcn_readlane(c, src));
                    longs &= longs - 1;
                }
                if (need) { cur1[i] = 0; need = head_walk(i, k1, k2); }
            }
        } else {
            unsigned long long longs = __ballot(need);
            while (longs) {
                const int src = __ffsll((long long)longs) - 1;
                unsigned long long t1, t2;
                pick_unsorted_wave(__builtin_amdgcn_readlane(i, src), -__builtin_amdgcn_readlane(c, src), t1, t2);
                if (lane == src) { k1 = t1; k2 = t2; }
                longs &= longs - 1;
            }
        }
        return valid && c != 0 ? commit(i, k1, k2) : (valid ? commit(i, ~0ull, ~0ull) : false);
    };
    if (mode != 1) {
        // Event-driven rounds (no second candidate, so a query's choice can only change when it loses its slot): the
        // first round steps every query, every later one only the queries the previous round pushed out -- the whole
        // pair lives on one CU, so the rounds are bound by instruction issue, and the work lists keep most wavefronts
        // out of them.  Counters vars[5..7] rotate: read / filled / reset.  Queries that do not hold slots
        // (unobserved) are not told when their pick is taken, they take one more step at the end.
        int c_in = 5, c_out = 6, c_clr = 7, rounds = 0;
        unsigned short *w_in = S.wl[0], *w_out = S.wl[1];
        wl_next = w_out; wl_count = &S.vars[c_out];
        for (int i0 = 0; i0 < nq; i0 += T) step_group(i0 + tid, i0 + tid < nq);
        __syncthreads();
        for (;; ++rounds) {
            { unsigned short *t_ = w_in; w_in = w_out; w_out = t_; }
            { const int t_ = c_in; c_in = c_out; c_out = c_clr; c_clr = t_; }
            const int nw = S.vars[c_in];
            if (nw == 0 || rounds > 65 * nq) break;
            if (tid == 0) S.vars[c_clr] = 0;
            wl_next = w_out; wl_count = &S.vars[c_out];
            for (int k0 = 0; k0 < nw; k0 += T) step_group(k0 + tid < nw ? (int)w_in[k0 + tid] : 0, k0 + tid < nw);
            __syncthreads();
        }
        wl_next = nullptr;
        bool unobs = false;
        for (int i = tid; i < nq; i += T) unobs |= !S.q_obs[i];
        if (__syncthreads_or(unobs))
            for (int i0 = 0; i0 < nq; i0 += T) step_group(i0 + tid, i0 + tid < nq && !S.q_obs[i0 + tid]);
#ifdef ORBHIP_DEVTOOLS
        if (tid == 0) { atomicAdd(&g_resolve_stats[0], 1u); atomicAdd(&g_resolve_stats[1], (unsigned)rounds + 1); atomicMax(&g_resolve_stats[2], (unsigned)rounds + 1); }
#endif
    } else {
        // mode 1 (map points: second candidate, ratio test among candidates of the same level).  Here a query CAN have
        // to give a held slot up: when its second candidate is taken by a smaller query, the next one may sit on the best
        // candidate's level and fail the ratio test that the previous pair never had to take -- holders are not
        // monotone, so no deferred acceptance.  The sequential loop is still the unique solution of "every query picks
        // against the claims of the smaller queries", found by fixed-point iteration: each round all queries re-pick
        // against the claims of the previous round and the claims are rebuilt from scratch (query i is final once all
        // j < i are: at most nq + 1 rounds, a handful in practice).
        int *own_cur = S.owner[0], *own_nxt = S.owner[1];
        claims_rebuilt = true;
        for (int round = 0; round <= nq + 1; ++round) {
            if (tid == 0) S.vars[0] = 0;
            for (int c = tid; c < n; c += T) own_nxt[c] = INT_MAX;
            __syncthreads();
            holder = own_cur; claim = own_nxt;
            bool ch = false;
            for (int i0 = 0; i0 < nq; i0 += T) ch |= step_group(i0 + tid, i0 + tid < nq);
            if (ch) S.vars[0] = 1;
            __syncthreads();
            const int changed = S.vars[0];
            { int *t_ = own_cur; own_cur = own_nxt; own_nxt = t_; }
            __syncthreads();
            if (!changed) {
#ifdef ORBHIP_DEVTOOLS
                if (tid == 0) { atomicAdd(&g_resolve_stats[0], 1u); atomicAdd(&g_resolve_stats[1], (unsigned)round + 1); atomicMax(&g_resolve_stats[2], (unsigned)round + 1); }
#endif
                break;
            }
        }
    }
    __syncthreads();
    // ---- outputs: assign[slot] = last accepted query that picked it; rotation-histogram cull (mode 0) ----
    int *assign = S.owner[1];
    for (int c = tid; c < n; c += T) assign[c] = -1;
    __syncthreads();
    int acc_local = 0;
    const bool ori = (mode == 0 || mode == 4) && check_ori;
    for (int i0 = 0; i0 < nq; i0 += T) {
        const int i = i0 + tid;
        const int c = i < nq ? S.choice[i] : -1;
        int bin = -1;
        if (c >= 0) {
            ++acc_local;
            atomicMax(&assign[c], i);
            if (ori) {
                bin = rot_bin(S.q_angle[i], S.t_angle[c]);
                if (bin >= 0) S.evbin[i] = (unsigned char)bin;
            }
        }
        // one LDS atomic per distinct bin of the wavefront (most matches of a frame pair share a rotation bin)
        unsigned long long todo = __ballot(bin >= 0);
        while (todo) {
            const int lead = __builtin_amdgcn_readlane(bin, __ffsll((long long)todo) - 1);
            const unsigned long long same = __ballot(bin == lead);
            if ((tid & 63) == __ffsll((long long)todo) - 1) atomicAdd(&S.hist[lead], __popcll(same));
            todo &= ~same;
        }
    }
    acc_local = wave_reduce_add_i(acc_local);
    if ((tid & 63) == 0 && acc_local) atomicAdd(&S.vars[1], acc_local);
    __syncthreads();
    if (ori) {
        if (tid < 64) {   // one wavefront ranks the bins (the pair's 16 wavefronts share one CU's issue slots)
            int i1, i2, i3;
            three_maxima(S.hist, i1, i2, i3);
            if (tid == 0) { S.vars[8] = i1; S.vars[9] = i2; S.vars[10] = i3; }
        }
        __syncthreads();
        const int ind1 = S.vars[8], ind2 = S.vars[9], ind3 = S.vars[10];
        int cull = 0;
        for (int i = tid; i < nq; i += T) {
            const int b = S.evbin[i];
            if (b != 0xff && b != ind1 && b != ind2 && b != ind3) {
                if (mode == 4) S.choice[i] = -1; else assign[S.choice[i]] = -1;
                ++cull;
            }
        }
        cull = wave_reduce_add_i(cull);
        if ((tid & 63) == 0 && cull) atomicAdd(&S.vars[2], cull);
        __syncthreads();
    }
    if (mode == 4) for (int i = tid; i < nq; i += T) out[i] = S.choice[i];   // per query, no slot exclusivity
    else for (int c = tid; c < n; c += T) out[c] = assign[c];
    if (tid == 0) *out_n = S.vars[1] - S.vars[2];
}

// ---- parallel resolve of SearchForInitialization (ORBmatcher.cc:405-520) -----------------------------------------
// The reference visits the queries in index order; a candidate slot b is usable for query i iff no earlier accepted
// query left a distance <= dist(i, b) on it (vMatchedDistance), the query takes its best usable candidate if it passes
// TH_LOW and the ratio test against the second best usable one, and a later, closer query steals the slot (the loser is
// NOT re-matched).  Every decision therefore depends only on the decisions of the queries with a smaller index: the
// sequential result is the unique fixed point of "every query decides against the claims of the smaller queries", and
// Jacobi iteration reaches it (query i is final once all j < i are).  A round rebuilds, per slot, a linked list of this
// round's claimants (atomicExch on the slot's head); the next round's queries walk the list of a candidate slot and take
// the minimum claimed distance among the claimants with a smaller index.  One workgroup per pair, state in LDS
// (n, nq <= 4096: the limit the entry points already state); the serial replay (k_resolve mode 2) this replaces took
// about 200 ns per query on one wavefront.
struct InitState {
    int *q_cnt; float *q_angle, *t_angle;
    uint32_t *claim[2];          // per query: distance << 20 | slot, or kNoClaim
    int *head[2];                // per slot: a claimant of this round, -1 if none
    unsigned short *next[2];     // per query: next claimant of the same slot, 0xffff = end
    unsigned char *evbin;
    int *hist, *vars;
    unsigned short *active;      // queries with a sorted list (<= 64 candidates), in index order (SearchForInitialization
                                 // only matches level-0 keypoints: about a fifth of the queries have candidates at all)
    unsigned short *active_u;    // queries with more than 64 candidates (unsorted list): a whole wavefront steps each
    uint32_t *lc; int lcn;
};
constexpr uint32_t kNoClaim = 0xffffffffu;
__host__ __device__ inline size_t init_state_bytes(size_t n, size_t nq, size_t lcn)
{
    n = (n + 3) & ~(size_t)3; nq = (nq + 3) & ~(size_t)3;
    return nq * 4 + nq * 4 + n * 4 + 2 * nq * 4 + 2 * n * 4 + 2 * nq * 2 + nq + 4 * nq + (HISTO_LENGTH + 2 + 64) * 4 + (lcn ? nq * (lcn + 1) * 4 : 0);
}
__global__ __launch_bounds__(1024) void k_resolve_init(DevFrame F, const orbhip_keypoint *__restrict__ qkeys, int nq,
                                                       const unsigned long long *__restrict__ cand,
                                                       const unsigned long long *__restrict__ ccand,
                                                       const int *__restrict__ cnt, int stride, float nnratio, int check_ori,
                                                       int *__restrict__ out, int *__restrict__ out_n, Batch B, int n_alloc,
                                                       int nq_alloc, int lcn)
{
    extern __shared__ unsigned char resolve_lds[];
    const int tid = threadIdx.x, T = blockDim.x;
    if (B.qcap > 0) {   // batched (device-resident) call: one workgroup per pair
        const int pair = blockIdx.x;
        batch_frame(F, B, pair);
        if (qkeys) qkeys += (size_t)(B.qd0 + pair * B.qds) * B.qcap;
        cand += (size_t)pair * B.qcap * stride;
        ccand += (size_t)pair * B.qcap * kCompact;
        cnt += (size_t)pair * B.qcap;
        out += (size_t)pair * B.qcap;
        out_n += pair;
        if (B.nq_dev) nq = min(B.nq_dev[pair], B.qcap);
    }
    const int n = F.n;
    InitState S;
    {
        const size_t na = ((size_t)n_alloc + 3) & ~(size_t)3, nqa = ((size_t)nq_alloc + 3) & ~(size_t)3;
        int *p = reinterpret_cast<int *>(resolve_lds);
        S.q_cnt = p; p += nqa;
        S.q_angle = reinterpret_cast<float *>(p); p += nqa;
        S.t_angle = reinterpret_cast<float *>(p); p += na;
        S.claim[0] = reinterpret_cast<uint32_t *>(p); p += nqa; S.claim[1] = reinterpret_cast<uint32_t *>(p); p += nqa;
        S.head[0] = p; p += na; S.head[1] = p; p += na;
        S.hist = p; p += HISTO_LENGTH + 2; S.vars = p; p += 64;
        S.next[0] = reinterpret_cast<unsigned short *>(p); S.next[1] = S.next[0] + nqa; p += nqa;
        S.evbin = reinterpret_cast<unsigned char *>(p); p += nqa / 4;
        S.active = reinterpret_cast<unsigned short *>(p); p += nqa / 2;
        S.active_u = reinterpret_cast<unsigned short *>(p); p += nqa / 2;
        S.lc = reinterpret_cast<uint32_t *>(p); S.lcn = lcn;
    }
    // list heads of this thread's first query, requested before anything else (one memory round trip with the rest)
    unsigned long long hv[kResolveHead];
    int head_c = 0;
    if (tid < nq) {
        head_c = cnt[tid];
        if (S.lcn > 0) {
            const unsigned long long *l0 = ccand + (size_t)tid * kCompact;
#pragma unroll
            for (int e = 0; e < kResolveHead; ++e) hv[e] = l0[e];
        }
    }
    for (int i = tid; i < n; i += T) { S.t_angle[i] = F.keys[i].angle; S.head[0][i] = -1; }
    for (int i = tid; i < nq; i += T) {
        S.q_cnt[i] = i == tid ? head_c : cnt[i];
        S.q_angle[i] = qkeys[i].angle;
        S.claim[0][i] = kNoClaim; S.claim[1][i] = kNoClaim;
        S.next[0][i] = 0xffff;
        S.evbin[i] = 0xff;
    }
    if (tid < HISTO_LENGTH) S.hist[tid] = 0;
    if (tid < 64) S.vars[tid] = 0;
    if (S.lcn > 0) {
        for (int i = tid; i < nq; i += T) {
            const int c = i == tid ? head_c : cnt[i];
            if (c <= 0) continue;
            if (i != tid) {
                const unsigned long long *l0 = ccand + (size_t)i * kCompact;
#pragma unroll
                for (int e = 0; e < kResolveHead; ++e) hv[e] = (e < c && e < S.lcn) ? l0[e] : ~0ull;
            }
#pragma unroll
            for (int e = 0; e < kResolveHead; ++e)
                if (e < c && e < S.lcn) S.lc[i * (S.lcn + 1) + e] = ((uint32_t)(hv[e] >> 32) << 20) | (uint32_t)(hv[e] & 0xfffffu);
        }
    }
    __syncthreads();
    // compact the queries that have candidates (index order; sorted and unsorted lists separately): thread t owns
    // queries [t * per, (t + 1) * per)
    int nact, nact_u;
    {
        const int per = (nq + T - 1) / T, i_lo = tid * per, i_hi = min(i_lo + per, nq);
        int c = 0, cu = 0;
        for (int i = i_lo; i < i_hi; ++i) { c += S.q_cnt[i] > 0; cu += S.q_cnt[i] < 0; }
        const int incl = wave_incl_scan_add(c), inclu = wave_incl_scan_add(cu);
        if ((tid & 63) == 63) { S.vars[16 + (tid >> 6)] = incl; S.vars[32 + (tid >> 6)] = inclu; }
        __syncthreads();
        int base = incl - c, tot = 0, baseu = inclu - cu, totu = 0;
        for (int w = 0; w < (T >> 6); ++w) {
            const int v = S.vars[16 + w], vu = S.vars[32 + w];
            if (w < (tid >> 6)) { base += v; baseu += vu; }
            tot += v; totu += vu;
        }
        for (int i = i_lo; i < i_hi; ++i) {
            if (S.q_cnt[i] > 0) S.active[base++] = (unsigned short)i;
            else if (S.q_cnt[i] < 0) S.active_u[baseu++] = (unsigned short)i;
        }
        nact = tot; nact_u = totu;
        __syncthreads();
    }
    int cur = 0;
    // vars[0..2]: "a claim changed" flags in rotation (raised in a round, reset one round ahead)
    int f_cur = 0, f_nxt = 1;
    for (int round = 0; round <= nq + 1; ++round) {
        const int nxt = cur ^ 1;
        const uint32_t *claim_c = S.claim[cur];
        const int *head_c2 = S.head[cur];
        const unsigned short *next_c = S.next[cur];
        uint32_t *claim_n = S.claim[nxt];
        int *head_n = S.head[nxt];
        unsigned short *next_n = S.next[nxt];
        for (int b = tid; b < n; b += T) head_n[b] = -1;
        if (tid == 0) S.vars[f_nxt] = 0;
        __syncthreads();
        bool ch = false;
        for (int a = tid; a < nact; a += T) {
            const int i = S.active[a];
            const int c = S.q_cnt[i];
            uint32_t k1 = kNoClaim;
            int d2 = INT_MAX;
            // the smallest distance an accepted query with a smaller index has left on slot b (vMatchedDistance)
            auto left_on = [&](int b) -> int {
                int D = INT_MAX;
                for (int j = head_c2[b]; j >= 0; j = next_c[j] == 0xffff ? -1 : (int)next_c[j])
                    if (j < i) D = min(D, (int)(claim_c[j] >> 20));
                return D;
            };
            if (c > 0) {          // sorted by (distance, visiting order): first and second usable entry
                const unsigned long long *list = ccand + (size_t)i * kCompact;
                const uint32_t *lrow = S.lc + i * (S.lcn + 1);
                for (int e = 0; e < c; ++e) {
                    uint32_t w;
                    if (e < S.lcn) w = lrow[e];
                    else { const unsigned long long v = list[e]; w = ((uint32_t)(v >> 32) << 20) | (uint32_t)(v & 0xfffffu); }
                    const int d = (int)(w >> 20), b = (int)(w & 0xfffffu);
                    if (left_on(b) <= d) continue;
                    if (k1 == kNoClaim) k1 = w; else { d2 = d; break; }
                }
            }
            uint32_t mine = kNoClaim;
            if (k1 != kNoClaim) {
                const int d1 = (int)(k1 >> 20);
                // bestDist <= TH_LOW && bestDist < (float)bestDist2 * mfNNratio, bestDist2 = INT_MAX when absent
                if (d1 <= TH_LOW && (float)d1 < __fmul_rn((float)d2, nnratio)) mine = k1;
            }
            claim_n[i] = mine;
            if (mine != kNoClaim) {
                const int old = atomicExch(&head_n[(int)(mine & 0xfffffu)], i);
                next_n[i] = old < 0 ? (unsigned short)0xffff : (unsigned short)old;
            }
            ch |= mine != claim_c[i];
        }
        // queries with more than 64 candidates (unsorted lists in HBM): one wavefront per query, lanes over the
        // candidates, two wave minima = smallest and second smallest usable key
        for (int u = tid >> 6; u < nact_u; u += T >> 6) {
            const int i = S.active_u[u];
            const int c = -S.q_cnt[i];
            const unsigned long long *list = cand + (size_t)i * stride;
            auto left_on = [&](int b) -> int {
                int D = INT_MAX;
                for (int j = head_c2[b]; j >= 0; j = next_c[j] == 0xffff ? -1 : (int)next_c[j])
                    if (j < i) D = min(D, (int)(claim_c[j] >> 20));
                return D;
            };
            unsigned long long m1 = ~0ull;
            for (int e = tid & 63; e < c; e += 64) {
                const unsigned long long v = list[e];
                if (left_on((int)(v & 0xfffffu)) > (int)(v >> 32)) m1 = v < m1 ? v : m1;
            }
            const unsigned long long k1v = wave_min_u64(m1);
            unsigned long long m2 = ~0ull;
            for (int e = tid & 63; e < c; e += 64) {
                const unsigned long long v = list[e];
                if (v != k1v && left_on((int)(v & 0xfffffu)) > (int)(v >> 32)) m2 = v < m2 ? v : m2;
            }
            const unsigned long long k2v = wave_min_u64(m2);
            if ((tid & 63) == 0) {
                uint32_t mine = kNoClaim;
                if (k1v != ~0ull) {
                    const int d1 = (int)(k1v >> 32), d2 = k2v == ~0ull ? INT_MAX : (int)(k2v >> 32);
                    if (d1 <= TH_LOW && (float)d1 < __fmul_rn((float)d2, nnratio)) mine = ((uint32_t)d1 << 20) | (uint32_t)(k1v & 0xfffffu);
                }
                claim_n[i] = mine;
                if (mine != kNoClaim) {
                    const int old = atomicExch(&head_n[(int)(mine & 0xfffffu)], i);
                    next_n[i] = old < 0 ? (unsigned short)0xffff : (unsigned short)old;
                }
                ch |= mine != claim_c[i];
            }
        }
        if (ch) S.vars[f_cur] = 1;
        __syncthreads();
        const int changed = S.vars[f_cur];
        cur = nxt;
        { const int t_ = f_cur; f_cur = f_nxt; f_nxt = 3 - t_ - f_nxt; }
        if (!changed) break;
    }
    // ---- outcome: a claim holds its slot unless a later query claimed the same slot (it was closer: it stole it);
    // every accepted query enters the rotation histogram, stolen or not, as the reference's rotHist does ----
    const uint32_t *claim = S.claim[cur];
    const int *head = S.head[cur];
    const unsigned short *next = S.next[cur];
    int kept = 0;
    for (int i0 = 0; i0 < nq; i0 += T) {
        const int i = i0 + tid;
        int m = -1, bin = -1;
        if (i < nq && claim[i] != kNoClaim) {
            const int b = (int)(claim[i] & 0xfffffu);
            bool stolen = false;
            for (int j = head[b]; j >= 0; j = next[j] == 0xffff ? -1 : (int)next[j]) stolen |= j > i;
            if (!stolen) m = b;
            if (check_ori) {
                bin = rot_bin(S.q_angle[i], S.t_angle[b]);
                if (bin >= 0) S.evbin[i] = (unsigned char)bin;
            }
        }
        if (i < nq) { S.q_cnt[i] = m; kept += m >= 0; }   // q_cnt is free now: vnMatches12
        unsigned long long todo = __ballot(bin >= 0);
        while (todo) {
            const int first = __ffsll((long long)todo) - 1;
            const int lead = __builtin_amdgcn_readlane(bin, first);
            const unsigned long long same = __ballot(bin == lead);
            if ((tid & 63) == first) atomicAdd(&S.hist[lead], __popcll(same));
            todo &= ~same;
        }
    }
    kept = wave_reduce_add_i(kept);
    if ((tid & 63) == 0 && kept) atomicAdd(&S.vars[4], kept);
    __syncthreads();
    if (check_ori) {
        if (tid < 64) {
            int i1, i2, i3;
            three_maxima(S.hist, i1, i2, i3);
            if (tid == 0) { S.vars[8] = i1; S.vars[9] = i2; S.vars[10] = i3; }
        }
        __syncthreads();
        const int ind1 = S.vars[8], ind2 = S.vars[9], ind3 = S.vars[10];
        int cull = 0;
        for (int i = tid; i < nq; i += T) {
            const int b = S.evbin[i];
            if (b != 0xff && b != ind1 && b != ind2 && b != ind3 && S.q_cnt[i] >= 0) { S.q_cnt[i] = -1; ++cull; }
        }
        cull = wave_reduce_add_i(cull);
        if ((tid & 63) == 0 && cull) atomicAdd(&S.vars[5], cull);
        __syncthreads();
    }
    for (int i = tid; i < nq; i += T) out[i] = S.q_cnt[i];
    if (tid == 0) *out_n = S.vars[4] - S.vars[5];
}

// ---- DescriptorDistance, batched (ORBmatcher.cc:1647-1663) --------------------------------
__global__ void k_distance_matrix(const uint8_t *__restrict__ a, int na, const uint8_t *__restrict__ b, int nb,
                                  int *__restrict__ dist)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j >= nb || i >= na) return;
    const uint32_t *pa = reinterpret_cast<const uint32_t *>(a + (size_t)i * 32);
    const uint32_t *pb = reinterpret_cast<const uint32_t *>(b + (size_t)j * 32);
    int d = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) d += __popc(pa[k] ^ pb[k]);
    dist[(size_t)i * nb + j] = d;
}

// ---- MapPoint::ComputeDistinctiveDescriptors (MapPoint.cc:242-307), batched over map points ----------------
// One wavefront per map point with N observations: for every row i the lanes hold the distances d(i, j); the
// median vDists[0.5*(N-1)] of the sorted row (self distance 0 included) is the k-th smallest value, found by a
// 9-step bisection on the value range [0, 256] with ballot/popcount counting (rows longer than 64 keep their
// distances in LDS).  Smallest median wins, first index on ties (:296-300).
constexpr int kDistinctMax = 2048;

__global__ __launch_bounds__(256) void k_distinctive(const uint8_t *__restrict__ desc, const int *__restrict__ offsets,
                                                     int npoints, int *__restrict__ best)
{
    __shared__ unsigned short srow[4][kDistinctMax];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int p = blockIdx.x * 4 + wv;
    if (p >= npoints) return;
    const int o0 = offsets[p], N = offsets[p + 1] - o0;
    if (N <= 0) { if (lane == 0) best[p] = -1; return; }
    const uint8_t *D = desc + (size_t)o0 * 32;
    const int k = (N - 1) >> 1;   // (int)(0.5*(N-1))
    uint32_t t0[8];
    {
        const uint32_t *tp = reinterpret_cast<const uint32_t *>(D + (size_t)(lane < N ? lane : 0) * 32);
#pragma unroll
        for (int w = 0; w < 8; ++w) t0[w] = tp[w];
    }
    int bestMedian = INT_MAX, bestIdx = 0;
    for (int i = 0; i < N; ++i) {
        uint32_t di[8];
        const uint32_t *ip = reinterpret_cast<const uint32_t *>(D + (size_t)i * 32);
#pragma unroll
        for (int w = 0; w < 8; ++w) di[w] = ip[w];
        const int d0 = lane < N ? hamming256(di, t0) : 0x7fff;
        if (N > 64) {
            for (int j = lane + 64; j < N; j += 64) {
                const uint32_t *tp = reinterpret_cast<const uint32_t *>(D + (size_t)j * 32);
                uint32_t t[8];
#pragma unroll
                for (int w = 0; w < 8; ++w) t[w] = tp[w];
                srow[wv][j] = (unsigned short)hamming256(di, t);
            }
            __builtin_amdgcn_wave_barrier();
        }
        int lo = 0, hi = 256;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            int cnt = __popcll(__ballot(d0 <= mid));
            if (N > 64) {
                int c = 0;
                for (int j = lane + 64; j < N; j += 64) c += srow[wv][j] <= mid;
                cnt += wave_reduce_add_i(c);
            }
            if (cnt >= k + 1) hi = mid; else lo = mid + 1;
        }
        if (lo < bestMedian) { bestMedian = lo; bestIdx = i; }
        if (N > 64) __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) best[p] = bestIdx;
}

// ---- Frame constructor glue on the device ------------------------------------------------
// Frame::AssignFeaturesToGrid (Frame.cc:230-245): mGrid[64][48] as CSR.  One workgroup per frame: keys
// (cell << 12 | index) are bitonic-sorted in LDS, which is push_back order inside every cell; cell sizes come from LDS
// atomics and an exclusive scan.
constexpr int kGridMax = 4096;
__global__ __launch_bounds__(1024) void k_grid_csr(DevFrame F, Batch B, int *__restrict__ cell_of,
                                                   int *__restrict__ cell_start, int *__restrict__ cell_items)
{
    __shared__ uint32_t key[kGridMax];
    __shared__ int cnt[kGridCells + 1];
    __shared__ int wsum[16];
    const int tid = threadIdx.x, NT = 1024, frame = blockIdx.x;
    batch_frame(F, B, frame);
    cell_of += (size_t)frame * B.cap;
    cell_items += (size_t)frame * B.cap;
    cell_start += (size_t)frame * (kGridCells + 1);
    const int n = min(F.n, kGridMax);
    int P = 1024;
    while (P < n) P <<= 1;
    for (int c = tid; c <= kGridCells; c += NT) cnt[c] = 0;
    __syncthreads();
    for (int i = tid; i < P; i += NT) {
        uint32_t k = 0xffffffffu;
        if (i < n) {
            const orbhip_keypoint kp = F.keys[i];
            const int px = (int)roundf(__fmul_rn(__fsub_rn(kp.x, F.min_x), F.inv_w));
            const int py = (int)roundf(__fmul_rn(__fsub_rn(kp.y, F.min_y), F.inv_h));
            const bool in = !(px < 0 || px >= GRID_COLS || py < 0 || py >= GRID_ROWS);
            const int c = px * GRID_ROWS + py;
            cell_of[i] = in ? c : -1;
            if (in) { k = ((uint32_t)c << 12) | (uint32_t)i; atomicAdd(&cnt[c], 1); }
        }
        key[i] = k;
    }
    __syncthreads();
    for (int k = 2; k <= P; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < P; i += NT) {
                const int l = i ^ j;
                if (l > i) {
                    const uint32_t a = key[i], b = key[l];
                    if ((a > b) == ((i & k) == 0)) { key[i] = b; key[l] = a; }
                }
            }
            __syncthreads();
        }
    for (int i = tid; i < n; i += NT)
        if (key[i] != 0xffffffffu) cell_items[i] = (int)(key[i] & 0xfffu);
    // exclusive scan of the 3072 cell sizes: 3 per thread, wave scan, then the 16 wave totals
    const int c0 = tid * 3;
    const int v0 = cnt[c0], v1 = cnt[c0 + 1], v2 = cnt[c0 + 2];
    int incl = v0 + v1 + v2;
    const int lane = tid & 63, wv = tid >> 6;
    incl = wave_incl_scan_add(incl);
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wv; ++w) base += wsum[w];
    const int excl = base + incl - (v0 + v1 + v2);
    cell_start[c0] = excl; cell_start[c0 + 1] = excl + v0; cell_start[c0 + 2] = excl + v0 + v1;
    if (tid == NT - 1) cell_start[kGridCells] = excl + v0 + v1 + v2;
}

// Frame::UndistortKeyPoints (Frame.cc:404-434) = cv::undistortPoints(mat, mat, mK, mDistCoef, Mat(), mK): OpenCV 2.4 - 3.3
// cvUndistortPoints restated from its published algorithm (double arithmetic, 5 fixed-point iterations of the inverse
// Brown model, re-projection with P = mK).  The translation unit is built with -ffp-contract=off, so every product
// and sum below rounds exactly like the C oracle's.
struct UndistortParams { double fx, fy, cx, cy, k[5]; };
__global__ void k_undistort(const orbhip_keypoint *__restrict__ keys, const int *__restrict__ n_dev, int n, int cap,
                            UndistortParams P, orbhip_keypoint *__restrict__ keys_un)
{
    const int frame = blockIdx.y;
    keys += (size_t)frame * cap; keys_un += (size_t)frame * cap;
    if (n_dev) n = min(n_dev[frame], cap);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    orbhip_keypoint kp = keys[i];
    const double ifx = 1. / P.fx, ify = 1. / P.fy;
    double x = kp.x, y = kp.y;
    const double x0 = x = (x - P.cx) * ifx;
    const double y0 = y = (y - P.cy) * ify;
    const double k0 = P.k[0], k1 = P.k[1], k2 = P.k[2], k3 = P.k[3], k4 = P.k[4];
#pragma unroll 1
    for (int j = 0; j < 5; ++j) {
        const double r2 = x * x + y * y;
        const double icdist = (1 + ((0.0 * r2 + 0.0) * r2 + 0.0) * r2) / (1 + ((k4 * r2 + k1) * r2 + k0) * r2);
        const double deltaX = 2 * k2 * x * y + k3 * (r2 + 2 * x * x);
        const double deltaY = k2 * (r2 + 2 * y * y) + 2 * k3 * x * y;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    const double xx = P.fx * x + 0.0 * y + P.cx;
    const double yy = 0.0 * x + P.fy * y + P.cy;
    const double ww = 1. / (0.0 * x + 0.0 * y + 1.0);
    kp.x = (float)(xx * ww);
    kp.y = (float)(yy * ww);
    keys_un[i] = kp;
}

// Frame::ComputeStereoFromRGBD (Frame.cc:643-664)
__global__ void k_stereo_from_rgbd(const orbhip_keypoint *__restrict__ keys, const orbhip_keypoint *__restrict__ keys_un,
                                   const int *__restrict__ n_dev, int n, int cap, const float *__restrict__ depth, int rows,
                                   int cols, int stride, size_t frame_stride, float mbf, float *__restrict__ u_right,
                                   float *__restrict__ depth_out)
{
    const int frame = blockIdx.y;
    keys += (size_t)frame * cap; keys_un += (size_t)frame * cap;
    u_right += (size_t)frame * cap; depth_out += (size_t)frame * cap;
    depth += (size_t)frame * frame_stride;
    if (n_dev) n = min(n_dev[frame], cap);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int v = (int)keys[i].y, u = (int)keys[i].x;      // Mat::at<float>(int,int) with float arguments: truncation
    float d = 0.f;
    if (v >= 0 && v < rows && u >= 0 && u < cols) d = depth[(size_t)v * stride + u];
    float ur = -1.0f, dp = -1.0f;
    if (d > 0) { dp = d; ur = __fsub_rn(keys_un[i].x, __fdiv_rn(mbf, d)); }
    u_right[i] = ur;
    depth_out[i] = dp;
}

// ---- Frame::ComputeStereoMatches (Frame.cc:466-640) -----------------------------------------
// ---------------------------------------------------------------------------
// Projection prologues (include/orbhip.h "projection prologues on the device"): one thread per map point.  Every
// float operation that feeds a comparison is written out un-contracted, in the order DESIGN.md section 3 states.
// ---------------------------------------------------------------------------
__device__ __forceinline__ float dot3_row(const float *r, float x, float y, float z, float t)
{
    return __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(r[0], x), __fmul_rn(r[1], y)), __fmul_rn(r[2], z)), t);
}

// fdlibm log in IEEE double, separate operations, one rounding to float (the deterministic logf of DESIGN.md section 3)
__device__ __forceinline__ float det_logf(float xf)
{
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                 Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                 Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    if (!(xf > 0.0f)) return xf == 0.0f ? -INFINITY : NAN;
    if (isinf(xf)) return xf;
    double x = (double)xf;
    unsigned long long bits = (unsigned long long)__double_as_longlong(x);
    int hx = (int)(bits >> 32);
    int k = (hx >> 20) - 1023;
    hx &= 0x000fffff;
    const int i = (hx + 0x95f64) & 0x100000;
    bits = ((unsigned long long)(unsigned)(hx | (i ^ 0x3ff00000)) << 32) | (bits & 0xffffffffull);
    x = __longlong_as_double((long long)bits);
    k += i >> 20;
    const double f = __dsub_rn(x, 1.0);
    const double s = __ddiv_rn(f, __dadd_rn(2.0, f));
    const double dk = (double)k;
    const double z = __dmul_rn(s, s);
    const double w = __dmul_rn(z, z);
    const double t1 = __dmul_rn(w, __dadd_rn(Lg2, __dmul_rn(w, __dadd_rn(Lg4, __dmul_rn(w, Lg6)))));
    const double t2 = __dmul_rn(z, __dadd_rn(Lg1, __dmul_rn(w, __dadd_rn(Lg3, __dmul_rn(w, __dadd_rn(Lg5, __dmul_rn(w, Lg7)))))));
    const double R = __dadd_rn(t2, t1);
    const double hfsq = __dmul_rn(__dmul_rn(0.5, f), f);
    const double r = __dsub_rn(__dmul_rn(dk, ln2_hi),
                               __dsub_rn(__dsub_rn(hfsq, __dadd_rn(__dmul_rn(s, __dadd_rn(hfsq, R)), __dmul_rn(dk, ln2_lo))), f));
    return (float)r;
}

struct ProjBatch {
    const float *Tcw, *Tlw;            // [pairs][12]
    const orbhip_keypoint *keys;       // [frames][cap]
    const int *n_dev;                  // [frames] or null (use n)
    const float *world;                // [frames][cap][3]
    const uint8_t *flags;              // [frames][cap]
    orbhip_query *q;                   // [pairs][cap]
    int *nq;                           // [pairs] or null
    int n, cap, l0, ls;
};

// src/ORBmatcher.cc:1339-1390
__device__ __forceinline__ void project_last_frame_body(const ProjBatch &B, const orbhip_camera &cam, const float th, const int mono,
                                                        const int pair, const int i)
{
    const size_t fl = (size_t)(B.l0 + pair * B.ls);
    const int n = B.n_dev ? min(B.n_dev[fl], B.cap) : B.n;
    if (i == 0 && B.nq) B.nq[pair] = n;
    if (i >= n) return;
    const float *Tcw = B.Tcw + (size_t)pair * 12, *Tlw = B.Tlw + (size_t)pair * 12;
    orbhip_query Q;
    Q.valid = 0; Q.u = 0; Q.v = 0; Q.radius = 0; Q.min_level = 0; Q.max_level = 0; Q.ur = 0; Q.level_aux = 0; Q.angle = 0; Q.observed = 0;
    orbhip_query *dst = B.q + (size_t)pair * B.cap + i;
    const unsigned fg = B.flags[fl * B.cap + i];
    if (fg & ORBHIP_POINT_PRESENT) {
        // twc = -Rcw.t()*tcw; tlc = Rlw*twc+tlw (:1342-1347); a dozen flops, recomputed per thread
        float twc[3];
#pragma unroll
        for (int c = 0; c < 3; ++c)
            twc[c] = -__fadd_rn(__fadd_rn(__fmul_rn(Tcw[c], Tcw[3]), __fmul_rn(Tcw[4 + c], Tcw[7])), __fmul_rn(Tcw[8 + c], Tcw[11]));
        const float tlc_z = dot3_row(Tlw + 8, twc[0], twc[1], twc[2], Tlw[11]);
        const bool forward = tlc_z > cam.mb && !mono, backward = -tlc_z > cam.mb && !mono;
        const float *X = B.world + (fl * B.cap + i) * 3;
        const float x = X[0], y = X[1], z = X[2];
        const float xc = dot3_row(Tcw, x, y, z, Tcw[3]);
        const float yc = dot3_row(Tcw + 4, x, y, z, Tcw[7]);
        const float zc = dot3_row(Tcw + 8, x, y, z, Tcw[11]);
        const float invzc = (float)__ddiv_rn(1.0, (double)zc);
        if (!(invzc < 0)) {
            const float u = __fadd_rn(__fmul_rn(__fmul_rn(cam.fx, xc), invzc), cam.cx);
            const float v = __fadd_rn(__fmul_rn(__fmul_rn(cam.fy, yc), invzc), cam.cy);
            if (!(u < cam.min_x || u > cam.max_x) && !(v < cam.min_y || v > cam.max_y)) {
                const orbhip_keypoint kp = B.keys[fl * B.cap + i];
                const int o = kp.octave;
                Q.valid = 1; Q.u = u; Q.v = v;
                Q.radius = __fmul_rn(th, cam.scale_factors[min(max(o, 0), ORBHIP_MAX_LEVELS - 1)]);
                if (forward) { Q.min_level = o; Q.max_level = -1; }
                else if (backward) { Q.min_level = 0; Q.max_level = o; }
                else { Q.min_level = o - 1; Q.max_level = o + 1; }
                Q.ur = __fsub_rn(u, __fmul_rn(cam.mbf, invzc));
                Q.level_aux = o;
                Q.angle = kp.angle;
                Q.observed = (fg & ORBHIP_POINT_OBSERVED) ? 1 : 0;
            }
        }
    }
    *dst = Q;
}
__global__ __launch_bounds__(256) void k_project_last_frame(ProjBatch B, orbhip_camera cam, float th, int mono)
{
    project_last_frame_body(B, cam, th, mono, (int)blockIdx.y, (int)(blockIdx.x * 256 + threadIdx.x));
}
// TrackWithMotionModel's matching step: the projection prologue and the CSR grid of the current frames do not depend on
// each other, so they share one launch: per pair one workgroup builds the grid, `npb` workgroups project
__global__ __launch_bounds__(256) void k_grid_build_project(DevFrame F, int *__restrict__ cell_start, GridRec *__restrict__ rec,
                                                            uint4 *__restrict__ rdesc, float *__restrict__ rur, Batch B,
                                                            ProjBatch P, orbhip_camera cam, float th, int mono, int npb)
{
    const int pair = (int)blockIdx.x / (npb + 1), role = (int)blockIdx.x - pair * (npb + 1);
    if (role == 0) grid_build_body(F, cell_start, rec, rdesc, rur, B, pair);
    else project_last_frame_body(P, cam, th, mono, pair, (role - 1) * 256 + (int)threadIdx.x);
}
struct ProjLaunch { ProjBatch P; orbhip_camera cam; float th; int mono; };

struct FrustumBatch {
    const float *Tcw;                       // [frames][12]
    const int *np_dev;                      // [frames] or null (use n)
    const float *world, *normal;            // [frames][pcap][3]
    const float *max_dist, *min_dist;       // [frames][pcap]
    const uint8_t *flags;                   // [frames][pcap]
    orbhip_query *q;                        // [frames][pcap]
    float *view_cos;                        // [frames][pcap] or null
    int n, pcap;
};

// src/Frame.cc:269-325, src/MapPoint.cc:400-418, src/ORBmatcher.cc:52-69, :131-137
__global__ __launch_bounds__(256) void k_frustum_queries(FrustumBatch B, orbhip_camera cam, float cos_limit, float th)
{
    const int fr = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    const int n = B.np_dev ? min(B.np_dev[fr], B.pcap) : B.n;
    if (i >= n) return;
    const size_t e = (size_t)fr * B.pcap + i;
    const float *Tcw = B.Tcw + (size_t)fr * 12;
    orbhip_query Q;
    Q.valid = 0; Q.u = 0; Q.v = 0; Q.radius = 0; Q.min_level = 0; Q.max_level = 0; Q.ur = 0; Q.level_aux = 0; Q.angle = 0; Q.observed = 0;
    float vc = 0.0f;
    const unsigned fg = B.flags[e];
    do {
        if (!(fg & ORBHIP_POINT_PRESENT)) break;
        float Ow[3];   // mOw = -mRcw.t()*mtcw
#pragma unroll
        for (int c = 0; c < 3; ++c)
            Ow[c] = -__fadd_rn(__fadd_rn(__fmul_rn(Tcw[c], Tcw[3]), __fmul_rn(Tcw[4 + c], Tcw[7])), __fmul_rn(Tcw[8 + c], Tcw[11]));
        const float *P = B.world + e * 3;
        const float px = P[0], py = P[1], pz = P[2];
        const float PcX = dot3_row(Tcw, px, py, pz, Tcw[3]);
        const float PcY = dot3_row(Tcw + 4, px, py, pz, Tcw[7]);
        const float PcZ = dot3_row(Tcw + 8, px, py, pz, Tcw[11]);
        if (PcZ < 0.0f) break;
        const float invz = __fdiv_rn(1.0f, PcZ);
        const float u = __fadd_rn(__fmul_rn(__fmul_rn(cam.fx, PcX), invz), cam.cx);
        const float v = __fadd_rn(__fmul_rn(__fmul_rn(cam.fy, PcY), invz), cam.cy);
        if (u < cam.min_x || u > cam.max_x) break;
        if (v < cam.min_y || v > cam.max_y) break;
        const float md = B.max_dist[e];
        const float maxDistance = __fmul_rn(1.2f, md), minDistance = __fmul_rn(0.8f, B.min_dist[e]);
        const float ox = __fsub_rn(px, Ow[0]), oy = __fsub_rn(py, Ow[1]), oz = __fsub_rn(pz, Ow[2]);
        const double ss = __dadd_rn(__dadd_rn(__dmul_rn((double)ox, (double)ox), __dmul_rn((double)oy, (double)oy)), __dmul_rn((double)oz, (double)oz));
        const float dist = (float)__dsqrt_rn(ss);
        if (dist < minDistance || dist > maxDistance) break;
        const float *Pn = B.normal + e * 3;
        const double dot = __dadd_rn(__dadd_rn(__dmul_rn((double)ox, (double)Pn[0]), __dmul_rn((double)oy, (double)Pn[1])), __dmul_rn((double)oz, (double)Pn[2]));
        const float viewCos = (float)__ddiv_rn(dot, (double)dist);
        if (viewCos < cos_limit) break;
        const float ratio = __fdiv_rn(md, dist);
        const float fl = ceilf(__fdiv_rn(det_logf(ratio), cam.log_scale_factor));
        int nScale = fl >= (float)cam.n_levels ? cam.n_levels - 1 : (fl < 0 ? 0 : (int)fl);
        if (!(fl == fl)) nScale = 0;
        float r = (double)viewCos > 0.998 ? 2.5f : 4.0f;
        if (th != 1.0f) r = __fmul_rn(r, th);
        Q.valid = 1; Q.u = u; Q.v = v;
        Q.radius = __fmul_rn(r, cam.scale_factors[min(max(nScale, 0), ORBHIP_MAX_LEVELS - 1)]);
        Q.min_level = nScale - 1; Q.max_level = nScale;
        Q.ur = __fsub_rn(u, __fmul_rn(cam.mbf, invz));
        Q.level_aux = nScale;
        Q.observed = (fg & ORBHIP_POINT_OBSERVED) ? 1 : 0;
        vc = viewCos;
    } while (0);
    B.q[e] = Q;
    if (B.view_cos) B.view_cos[e] = vc;
}

// SearchForInitialization, device-resident form: queries = level-0 keypoints of F1 searched around vbPrevMatched
// (src/ORBmatcher.cc:418-425); `reset` first sets vbPrevMatched[i] = F1.mvKeysUn[i].pt (src/Tracking.cc:578-580).
__global__ __launch_bounds__(256) void k_init_queries(const orbhip_keypoint *__restrict__ keys, const int *__restrict__ n_dev,
                                                      int cap, int f0, int fs, float *__restrict__ prev, int reset,
                                                      float window, orbhip_query *__restrict__ q, int *__restrict__ nq)
{
    const int pair = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    const size_t f1 = (size_t)(f0 + pair * fs);
    const int n = min(n_dev[f1], cap);
    if (i == 0) nq[pair] = n;
    if (i >= n) return;
    const orbhip_keypoint kp = keys[f1 * cap + i];
    float *pm = prev + ((size_t)pair * cap + i) * 2;
    if (reset) { pm[0] = kp.x; pm[1] = kp.y; }
    orbhip_query Q;
    Q.valid = kp.octave > 0 ? 0 : 1;
    Q.u = pm[0]; Q.v = pm[1];
    Q.radius = window;
    Q.min_level = kp.octave; Q.max_level = kp.octave;
    Q.ur = 0; Q.level_aux = 0; Q.angle = kp.angle; Q.observed = 0;
    q[(size_t)pair * cap + i] = Q;
}

// vbPrevMatched[i1] = F2.mvKeysUn[vnMatches12[i1]].pt for every match (src/ORBmatcher.cc:515-517)
__global__ __launch_bounds__(256) void k_init_update_prev(const orbhip_keypoint *__restrict__ keys, int cap, int f0, int fs,
                                                          const int *__restrict__ nq, const int *__restrict__ m12,
                                                          float *__restrict__ prev)
{
    const int pair = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nq[pair]) return;
    const int j = m12[(size_t)pair * cap + i];
    if (j < 0) return;
    const orbhip_keypoint kp = keys[(size_t)(f0 + pair * fs) * cap + j];
    float *pm = prev + ((size_t)pair * cap + i) * 2;
    pm[0] = kp.x; pm[1] = kp.y;
}

// Prologue of ORBmatcher::Fuse (both overloads) and of one direction of SearchBySim3: see include/orbhip.h
// (orbhip_keyframe_queries).  One thread per map point.
struct KfQueryArgs {
    const float *T1, *T2;                   // 12 floats each (T2 unused in mode 0)
    const float *world, *normal, *max_dist, *min_dist;
    const uint8_t *flags;
    orbhip_query *q;
    int n, mode, double_invz;
};
__global__ __launch_bounds__(256) void k_keyframe_queries(KfQueryArgs A, orbhip_camera cam, float th)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= A.n) return;
    orbhip_query Q;
    Q.valid = 0; Q.u = 0; Q.v = 0; Q.radius = 0; Q.min_level = 0; Q.max_level = 0; Q.ur = 0; Q.level_aux = 0; Q.angle = 0; Q.observed = 0;
    do {
        if (!(A.flags[i] & ORBHIP_POINT_PRESENT)) break;
        const float *T1 = A.T1;
        const float *P = A.world + (size_t)i * 3;
        const float px = P[0], py = P[1], pz = P[2];
        float X = dot3_row(T1, px, py, pz, T1[3]);
        float Y = dot3_row(T1 + 4, px, py, pz, T1[7]);
        float Z = dot3_row(T1 + 8, px, py, pz, T1[11]);
        if (A.mode == 1) {
            const float *T2 = A.T2;
            const float x1 = X, y1 = Y, z1 = Z;
            X = dot3_row(T2, x1, y1, z1, T2[3]);
            Y = dot3_row(T2 + 4, x1, y1, z1, T2[7]);
            Z = dot3_row(T2 + 8, x1, y1, z1, T2[11]);
        }
        if (Z < 0.0f) break;
        const float invz = A.double_invz ? (float)__ddiv_rn(1.0, (double)Z) : __fdiv_rn(1.0f, Z);
        const float x = __fmul_rn(X, invz), y = __fmul_rn(Y, invz);
        const float u = __fadd_rn(__fmul_rn(cam.fx, x), cam.cx), v = __fadd_rn(__fmul_rn(cam.fy, y), cam.cy);
        if (!(u >= cam.min_x && u < cam.max_x && v >= cam.min_y && v < cam.max_y)) break;   // KeyFrame::IsInImage
        const float md = A.max_dist[i];
        const float maxDistance = __fmul_rn(1.2f, md), minDistance = __fmul_rn(0.8f, A.min_dist[i]);
        float dist3D;
        if (A.mode == 0) {
            float Ow[3];
#pragma unroll
            for (int c = 0; c < 3; ++c)
                Ow[c] = -__fadd_rn(__fadd_rn(__fmul_rn(T1[c], T1[3]), __fmul_rn(T1[4 + c], T1[7])), __fmul_rn(T1[8 + c], T1[11]));
            const float ox = __fsub_rn(px, Ow[0]), oy = __fsub_rn(py, Ow[1]), oz = __fsub_rn(pz, Ow[2]);
            const double ss = __dadd_rn(__dadd_rn(__dmul_rn((double)ox, (double)ox), __dmul_rn((double)oy, (double)oy)), __dmul_rn((double)oz, (double)oz));
            dist3D = (float)__dsqrt_rn(ss);
            if (dist3D < minDistance || dist3D > maxDistance) break;
            const float *Pn = A.normal + (size_t)i * 3;
            const double dot = __dadd_rn(__dadd_rn(__dmul_rn((double)ox, (double)Pn[0]), __dmul_rn((double)oy, (double)Pn[1])), __dmul_rn((double)oz, (double)Pn[2]));
            if (dot < __dmul_rn(0.5, (double)dist3D)) break;
        } else {
            const double ss = __dadd_rn(__dadd_rn(__dmul_rn((double)X, (double)X), __dmul_rn((double)Y, (double)Y)), __dmul_rn((double)Z, (double)Z));
            dist3D = (float)__dsqrt_rn(ss);
            if (dist3D < minDistance || dist3D > maxDistance) break;
        }
        const float fl = ceilf(__fdiv_rn(det_logf(__fdiv_rn(md, dist3D)), cam.log_scale_factor));
        int lvl = fl >= (float)cam.n_levels ? cam.n_levels - 1 : (fl < 0 ? 0 : (int)fl);
        if (!(fl == fl)) lvl = 0;
        Q.valid = 1; Q.u = u; Q.v = v;
        Q.radius = __fmul_rn(th, cam.scale_factors[min(max(lvl, 0), ORBHIP_MAX_LEVELS - 1)]);
        Q.min_level = lvl - 1; Q.max_level = lvl;
        Q.ur = A.mode == 0 ? __fsub_rn(u, __fmul_rn(cam.mbf, invz)) : 0.0f;
        Q.level_aux = lvl;
    } while (0);
    A.q[i] = Q;
}

struct StereoGeom {
    int nlevels, nrows;
    const uint8_t *left[ORBHIP_MAX_LEVELS], *right[ORBHIP_MAX_LEVELS];
    int pitch_l[ORBHIP_MAX_LEVELS], pitch_r[ORBHIP_MAX_LEVELS], cols_r[ORBHIP_MAX_LEVELS];
    float sf[ORBHIP_MAX_LEVELS], isf[ORBHIP_MAX_LEVELS];
    float mbf, mb;
};

// rows a right keypoint's band can reach from floor(y): ceil(2 * largest scale factor) + 1
static int stereo_row_reach(const StereoGeom &G)
{
    float mx = 0.f;
    for (int l = 0; l < G.nlevels; ++l) mx = std::max(mx, G.sf[l]);
    return (int)ceilf(2.0f * mx) + 1;
}
struct StereoScales;
static StereoScales stereo_scales(const StereoGeom &G);
// Batched stereo: pair p uses frame l0 + p*ls of the left arrays/pyramids and r0 + p*rs of the right ones.
struct StereoBatch {
    const int *n_l, *n_r;   // per-frame keypoint counts on the device, or null (use the nl / nr arguments)
    int l0, ls, r0, rs;     // frame index mapping
    int cap;                // keypoint stride per frame
    unsigned fb_l, fb_r;    // pyramid bytes per frame of the two extractor handles
};

// Right keypoints bucketed by image row (counting sort on floor(y)): a left keypoint's candidates -- the right keypoints
// whose row band [floor(y - r), ceil(y + r)], r = 2 * scale[octave] (Frame.cc:483-493) contains its row -- all lie within
// R = ceil(2 * largest scale) + 1 rows of it, i.e. in ONE contiguous range of the sorted order, instead of anywhere among
// the nr right keypoints.  One workgroup per pair; rowstart[nrows + 1], order[nr] (order inside a row is irrelevant: the
// match is the minimum of (distance, index) keys).
constexpr int kStereoRowsMax = 4096;
struct StereoScales { float sf[ORBHIP_MAX_LEVELS]; };
__global__ __launch_bounds__(256) void k_stereo_sort(const orbhip_keypoint *__restrict__ kr, int nr, int nrows,
                                                     int *__restrict__ rowstart, int4 *__restrict__ order, StereoBatch B,
                                                     StereoScales SF)
{
    __shared__ int s_cnt[kStereoRowsMax];
    __shared__ int s_wave[4];
    const int tid = threadIdx.x, pair = blockIdx.x, fr = B.r0 + pair * B.rs;
    kr += (size_t)fr * B.cap;
    rowstart += (size_t)pair * (nrows + 1);
    order += (size_t)pair * B.cap;
    if (B.n_r) nr = min(B.n_r[fr], B.cap);
    for (int r = tid; r < nrows; r += 256) s_cnt[r] = 0;
    __syncthreads();
    for (int i = tid; i < nr; i += 256) atomicAdd(&s_cnt[min(max((int)floorf(kr[i].y), 0), nrows - 1)], 1);
    __syncthreads();
    const int per = (nrows + 255) / 256, r0 = tid * per, r1 = min(r0 + per, nrows);
    int sum = 0;
    for (int r = r0; r < r1; ++r) sum += s_cnt[r];
    const int incl = wave_incl_scan_add(sum);
    if ((tid & 63) == 63) s_wave[tid >> 6] = incl;
    __syncthreads();
    int base = incl - sum;
    for (int w = 0; w < (tid >> 6); ++w) base += s_wave[w];
    for (int r = r0; r < r1; ++r) { const int c = s_cnt[r]; s_cnt[r] = base; rowstart[r] = base; base += c; }
    if (tid == 255) rowstart[nrows] = base;
    __syncthreads();
    // the sorted order holds RECORDS, not indices: everything k_stereo_match needs to gate a candidate -- x, the row band
    // [floor(y - r), ceil(y + r)] with r = 2 * scale[octave] (same float operations as Frame.cc:483-493; clamped to
    // [0, 4095], which no comparison against an image row can tell), the octave, the index -- in one 16-byte load instead
    // of an index load followed by a dependent 28-byte keypoint load and a 15-way select of the scale per candidate
    for (int i = tid; i < nr; i += 256) {
        const orbhip_keypoint k = kr[i];
        float sfo = SF.sf[0];
#pragma unroll
        for (int l = 1; l < ORBHIP_MAX_LEVELS; ++l) sfo = k.octave == l ? SF.sf[l] : sfo;
        const float r = __fmul_rn(2.0f, sfo);
        const int minr = min(max((int)floorf(__fsub_rn(k.y, r)), 0), 4095), maxr = min(max((int)ceilf(__fadd_rn(k.y, r)), 0), 4095);
        const int pos = atomicAdd(&s_cnt[min(max((int)floorf(k.y), 0), nrows - 1)], 1);
        order[pos] = make_int4(__float_as_int(k.x), minr | (maxr << 12) | ((k.octave & 15) << 24), i, 0);
    }
}

__global__ __launch_bounds__(256) void k_stereo_match(const orbhip_keypoint *__restrict__ kl,
                                                      const uint8_t *__restrict__ dl, int nl,
                                                      const orbhip_keypoint *__restrict__ kr,
                                                      const uint8_t *__restrict__ dr, int nr,
                                                      const int *__restrict__ rowstart, const int4 *__restrict__ order, int R,
                                                      StereoGeom G,
                                                      float *__restrict__ uRight, float *__restrict__ depth,
                                                      int *__restrict__ sad, StereoBatch B)
{
    const int lane = threadIdx.x & 63;
    size_t pyr_off_l, pyr_off_r;   // this pair's frames inside the two pyramid batches
    {
        const int pair = blockIdx.y, fl = B.l0 + pair * B.ls, fr = B.r0 + pair * B.rs;
        pyr_off_l = (size_t)fl * B.fb_l; pyr_off_r = (size_t)fr * B.fb_r;
        kl += (size_t)fl * B.cap; dl += (size_t)fl * B.cap * 32;
        kr += (size_t)fr * B.cap; dr += (size_t)fr * B.cap * 32;
        if (rowstart) { rowstart += (size_t)pair * (G.nrows + 1); order += (size_t)pair * B.cap; }
        uRight += (size_t)pair * B.cap; depth += (size_t)pair * B.cap; sad += (size_t)pair * B.cap;
        if (B.n_l) nl = min(B.n_l[fl], B.cap);
        if (B.n_r) nr = min(B.n_r[fr], B.cap);
    }
    const int iL = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (iL >= nl) return;
    if (lane == 0) { uRight[iL] = -1.0f; depth[iL] = -1.0f; sad[iL] = -1; }
    const orbhip_keypoint kpL = kl[iL];
    // one keypoint per wavefront: the level is wave-uniform, and saying so keeps the per-level tables of G in scalar
    // registers (a lane-indexed by-value array would be copied to scratch by every wave)
    const int levelL = __builtin_amdgcn_readfirstlane(kpL.octave);
    const float vL = kpL.y, uL = kpL.x;
    const int row = (int)vL;
    if (row < 0 || row >= G.nrows) return;
    const float minZ = G.mb, minD = 0.f;
    const float maxD = __fdiv_rn(G.mbf, minZ);
    const float minU = __fsub_rn(uL, maxD), maxU = __fsub_rn(uL, minD);
    if (maxU < 0) return;
    uint32_t qd[8];
    const uint32_t *qp = reinterpret_cast<const uint32_t *>(dl + (size_t)iL * 32);
#pragma unroll
    for (int i = 0; i < 8; ++i) qd[i] = qp[i];
    // The kernel is a chain of dependent memory round trips (keypoint -> row range -> candidate records -> descriptors ->
    // right patch), so what does not depend on the match leaves early: the left 11x11 patch of the sub-pixel refinement
    // (:555-592; it needs the left keypoint alone and lies inside the padded plane for every keypoint) is requested here.
    const float scaleFactor = G.isf[levelL];
    const float scaleduL = roundf(__fmul_rn(kpL.x, scaleFactor));
    const float scaledvL = roundf(__fmul_rn(kpL.y, scaleFactor));
    const int w = 5, L = 5;
    const uint8_t *imL = G.left[levelL] + pyr_off_l, *imR = G.right[levelL] + pyr_off_r;
    const int stL = G.pitch_l[levelL], stR = G.pitch_r[levelL];
    const int cu = (int)scaleduL, cv = (int)scaledvL;
    // Both patches travel as byte-unaligned DWORD loads into a per-wavefront LDS tile -- one load instruction for the left
    // 11 x 12-byte window, two for the right 11 x 24-byte one (11 x 21 needed: 11 columns x 11 shifts) -- and are read
    // from there byte by byte: the 35 byte-gather loads per lane this replaces (2 + 11 x 3) kept the texture addresser,
    // not the ALUs, busy (67 us per 64 pairs with any amount of arithmetic removed).
    __shared__ uint32_t s_patch[4][11 * 3 + 11 * 6 + 1];
    uint32_t *sl = s_patch[threadIdx.x >> 6], *sr = sl + 11 * 3;
    uint32_t lw = 0;
    if (lane < 33) {
        const int prow = (lane * 43) >> 7, pc = lane - prow * 3;     // lane / 3 for lane < 33
        __builtin_memcpy(&lw, imL + (ptrdiff_t)(cv - w + prow) * stL + (cu - w) + 4 * pc, 4);
    }
    // each lane owns up to 2 of the 121 patch pixels
    int dyv[2], dxv[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        int p = lane + 64 * t;
        dyv[t] = p / 11 - w; dxv[t] = p % 11 - w;
    }
    // best right keypoint on this row: smallest (dist, iR) with dist < TH_HIGH (:522-549).  Candidates: the rows within R
    // of this one in the row-sorted order (all right keypoints when there is no order)
    // key = distance << 20 | index (distances <= 256, indices < 2^20: checked on the host): the wave minimum is six
    // v_min_u32 on the DPP path instead of six 64-bit compare-select steps
    uint32_t best = (uint32_t)TH_HIGH << 20;
    float bestx = 0.f;               // x of this lane's best candidate (travels with the key: no keypoint reload after the minimum)
    int jb = 0, je = nr;
    if (rowstart) { jb = rowstart[max(row - R, 0)]; je = rowstart[min(row + R + 1, G.nrows)]; }
    for (int j0 = jb; j0 < je; j0 += 64) {
        const int j = j0 + lane;
        if (j < je) {
            int iR, minr, maxr, octR;
            float xR;
            if (rowstart) {
                const int4 rc = order[j];
                xR = __int_as_float(rc.x); minr = rc.y & 4095; maxr = (rc.y >> 12) & 4095; octR = rc.y >> 24; iR = rc.z;
            } else {
                iR = j;
                const orbhip_keypoint kpR = kr[iR];
                float sfo = G.sf[0];   // scale of the right keypoint's level, by selects (a lane-indexed read would put G into scratch)
#pragma unroll
                for (int l = 1; l < ORBHIP_MAX_LEVELS; ++l) sfo = kpR.octave == l ? G.sf[l] : sfo;
                const float r = __fmul_rn(2.0f, sfo);
                minr = (int)floorf(__fsub_rn(kpR.y, r)); maxr = (int)ceilf(__fadd_rn(kpR.y, r));
                octR = kpR.octave; xR = kpR.x;
            }
            if (row >= minr && row <= maxr && !(octR < levelL - 1 || octR > levelL + 1) && xR >= minU && xR <= maxU) {
                const uint32_t *tp = reinterpret_cast<const uint32_t *>(dr + (size_t)iR * 32);
                uint32_t td[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) td[i] = tp[i];
                const uint32_t key = ((uint32_t)hamming256(qd, td) << 20) | (uint32_t)iR;
                if (key < best) { best = key; bestx = xR; }
            }
        }
    }
    const uint32_t mine_key = best;
    best = (uint32_t)wave_min((int)best);          // keys are < 2^31: the signed minimum is the unsigned one
    const int bestDist = (int)(best >> 20);
    const int thOrbDist = (TH_HIGH + TH_LOW) / 2;
    if (!(bestDist < thOrbDist)) return;
    // sub-pixel refinement by 11x11 SAD over 11 shifts at the keypoint's level (:555-592); x of the winner from the lane that holds it
    const unsigned long long owner = __ballot(mine_key == best);
    const float uR0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bestx), __ffsll((long long)owner) - 1));
    const float scaleduR0 = roundf(__fmul_rn(uR0, scaleFactor));
    const float iniu = __fsub_rn(__fadd_rn(scaleduR0, (float)L), (float)w);
    const float endu = __fadd_rn(__fadd_rn(__fadd_rn(scaleduR0, (float)L), (float)w), 1.0f);
    if (iniu < 0 || endu >= (float)G.cols_r[levelL]) return;
    const int cr = (int)scaleduR0;
    // stage the two windows: right window = rows cv-5 .. cv+5, bytes cr-10 .. cr+13 (columns beyond cr+10 are never read;
    // they lie inside the padded plane: cr + 11 < cols was just checked)
    if (lane < 33) sl[lane] = lw;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int idx = lane + 64 * t;
        if (idx < 66) {
            const int prow = (idx * 43) >> 8, pc = idx - prow * 6;     // idx / 6 for idx < 66
            uint32_t rw;
            __builtin_memcpy(&rw, imR + (ptrdiff_t)(cv - w + prow) * stR + (cr - 2 * w) + 4 * pc, 4);
            sr[idx] = rw;
        }
    }
    __builtin_amdgcn_wave_barrier();          // DS operations of a wavefront execute in order
    const uint8_t *bl = reinterpret_cast<const uint8_t *>(sl), *br = reinterpret_cast<const uint8_t *>(sr);
    const int cL = bl[5 * 12 + 5];
    int pl[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) pl[t] = lane + 64 * t < 121 ? (int)bl[(dyv[t] + w) * 12 + dxv[t] + w] - cL : 0;
    // SAD of the 11 shifts; a shift's total is at most 121 * 510 < 2^16, so two shifts share one wave sum
    int accs[11];
#pragma unroll
    for (int s = 0; s < 11; ++s) {
        const int incR = s - L;
        const int cR = br[5 * 24 + 2 * w + incR];
        int acc = 0;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            int p = lane + 64 * t;
            if (p < 121) {
                int b = (int)br[(dyv[t] + w) * 24 + 2 * w + incR + dxv[t]] - cR;
                acc += abs(pl[t] - b);
            }
        }
        accs[s] = acc;
    }
    int dists[11];
#pragma unroll
    for (int s = 0; s < 10; s += 2) {
        const uint32_t two = (uint32_t)wave_reduce_add_i(accs[s] | (accs[s + 1] << 16));
        dists[s] = (int)(two & 0xffffu); dists[s + 1] = (int)(two >> 16);
    }
    dists[10] = wave_reduce_add_i(accs[10]);
    int bestD = INT_MAX, bestincR = 0;
#pragma unroll
    for (int s = 0; s < 11; ++s) if (dists[s] < bestD) { bestD = dists[s]; bestincR = s - L; }
    if (bestincR == -L || bestincR == L) return;
    float dist1 = 0, dist2 = 0, dist3 = 0;
#pragma unroll
    for (int s = 1; s < 10; ++s) if (s == L + bestincR) { dist1 = (float)dists[s - 1]; dist2 = (float)dists[s]; dist3 = (float)dists[s + 1]; }
    const float deltaR = __fdiv_rn(__fsub_rn(dist1, dist3),
                                   __fmul_rn(2.0f, __fsub_rn(__fadd_rn(dist1, dist3), __fmul_rn(2.0f, dist2))));
    if (deltaR < -1 || deltaR > 1) return;
    float bestuR = __fmul_rn(G.sf[levelL], __fadd_rn(__fadd_rn(scaleduR0, (float)bestincR), deltaR));
    float disparity = __fsub_rn(uL, bestuR);
    if (disparity >= minD && disparity < maxD) {
        if (disparity <= 0) {
            disparity = 0.01f;                              // float(0.01)
            bestuR = (float)((double)uL - 0.01);            // float - double literal
        }
        if (lane == 0) {
            depth[iL] = __fdiv_rn(G.mbf, disparity);
            uRight[iL] = bestuR;
            sad[iL] = bestD;
        }
    }
}

static StereoScales stereo_scales(const StereoGeom &G)
{
    StereoScales S;
    for (int l = 0; l < ORBHIP_MAX_LEVELS; ++l) S.sf[l] = G.sf[l < G.nlevels ? l : 0];
    return S;
}

// median-based outlier cull (:626-639): thDist = 1.5f*1.4f*median of the SAD list sorted by
// (dist, iL); entries with dist >= thDist are removed.
__global__ __launch_bounds__(256) void k_stereo_cull(int nl, const int *__restrict__ sad, float *__restrict__ uRight,
                                                     float *__restrict__ depth, int *__restrict__ out_n, StereoBatch B)
{
    __shared__ int s_nd, s_med, s_cnt;
    const int tid = threadIdx.x;
    {
        const int pair = blockIdx.x;
        sad += (size_t)pair * B.cap; uRight += (size_t)pair * B.cap; depth += (size_t)pair * B.cap;
        out_n += pair;
        if (B.n_l) nl = min(B.n_l[B.l0 + pair * B.ls], B.cap);
    }
    if (tid == 0) { s_nd = 0; s_med = 0; s_cnt = 0; }
    __syncthreads();
    int local = 0;
    for (int i = tid; i < nl; i += 256) local += sad[i] >= 0;
    atomicAdd(&s_nd, local);
    __syncthreads();
    const int nd = s_nd;
    if (nd == 0) { if (tid == 0) *out_n = 0; return; }
    const int target = nd / 2;
    // median = the (nd/2)-th smallest valid SAD (src/Frame.cc:628-630 sorts the (SAD, index) pairs and takes the middle
    // one; only its value is used).  One wavefront finds it by bisection on the value: count(d <= mid) over an LDS copy
    // of the list with a DPP wave sum per probe -- 16 probes, no barrier; the O(n^2) rank count this replaces took
    // 130 us per 64 pairs.  Lists longer than the LDS copy are read from HBM.
    __shared__ int s_sad[kResolveMax];
    const bool in_lds = nl <= kResolveMax;
    if (in_lds) for (int i = tid; i < nl; i += 256) s_sad[i] = sad[i];
    __syncthreads();
    // The values are L1 norms of two 11 x 11 byte windows, each minus its centre pixel: <= 11 * 11 * 510 < 2^16.  Two-level
    // radix select over all 256 threads: histogram of the high bytes, the bin that holds rank `target` (smallest v with
    // count(d <= v) > target), then the histogram of the low bytes inside that bin -- four barriers instead of 16
    // dependent bisection probes by one wavefront (26 -> 9 us per 64 pairs).
    __shared__ int s_hist[256];
    __shared__ int s_bin, s_rank;
    int rank = target, value = 0;
#pragma unroll 1
    for (int level = 0; level < 2; ++level) {
        s_hist[tid] = 0;
        __syncthreads();
        for (int i = tid; i < nl; i += 256) {
            const int d = in_lds ? s_sad[i] : sad[i];
            if (d < 0) continue;
            if (level == 0) atomicAdd(&s_hist[d >> 8], 1);
            else if ((d >> 8) == value) atomicAdd(&s_hist[d & 255], 1);
        }
        __syncthreads();
        if (tid < 64) {
            const int h0 = s_hist[4 * tid], h1 = s_hist[4 * tid + 1], h2 = s_hist[4 * tid + 2], h3 = s_hist[4 * tid + 3];
            const int sum = h0 + h1 + h2 + h3, incl = wave_incl_scan_add(sum);
            const unsigned long long over = __ballot(incl > rank);       // non-empty: the total count exceeds the rank
            if (tid == __ffsll((long long)over) - 1) {
                int c = incl - sum, bin = 4 * tid;                        // first bin of this lane whose running count passes the rank
                if (c + h0 > rank) { }
                else if (c + h0 + h1 > rank) { c += h0; bin += 1; }
                else if (c + h0 + h1 + h2 > rank) { c += h0 + h1; bin += 2; }
                else { c += h0 + h1 + h2; bin += 3; }
                s_bin = bin; s_rank = rank - c;
            }
        }
        __syncthreads();
        value = level == 0 ? s_bin : (value << 8) | s_bin;
        rank = s_rank;
        __syncthreads();
    }
    if (tid == 0) s_med = value;
    __syncthreads();
    const float median = (float)s_med;
    const float thDist = __fmul_rn(1.5f * 1.4f, median);
    int kept = 0;
    for (int i = tid; i < nl; i += 256) {
        const int d = sad[i];
        if (d < 0) continue;
        if (!((float)d < thDist)) { uRight[i] = -1; depth[i] = -1; }
        else kept++;
    }
    atomicAdd(&s_cnt, kept);
    __syncthreads();
    if (tid == 0) *out_n = s_cnt;
}

}  // namespace orbhip

// =============================================================================================
using namespace orbhip;

struct orbhip_matcher {
    int device = 0;
    hipStream_t stream = nullptr;       // stream every launch goes to
    hipStream_t own_stream = nullptr;   // created with the handle
    // grow-only device scratch
    void *buf[16] = {};
    size_t cap[16] = {};
    bool lds_attr_set = false, bow_attr_set = false;
    // pinned host staging: all inputs of a call travel in one DMA, all outputs in one
    uint8_t *h_stage = nullptr; size_t h_stage_bytes = 0;
    uint8_t *h_out = nullptr; size_t h_out_bytes = 0;
};

static inline size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }

struct Stage {
    uint8_t *h, *d;
    size_t off;
    void *put(const void *src, size_t bytes)
    {
        void *dev = d + off;
        if (bytes) memcpy(h + off, src, bytes);
        off += al256(bytes);
        return dev;
    }
};

static int scratch(orbhip_matcher *m, int slot, size_t bytes, void **out)
{
    if (bytes < 256) bytes = 256;
    if (bytes > m->cap[slot]) {
        ORBHIP_HIP_CHECK(hipStreamSynchronize(m->stream));
        (void)hipFree(m->buf[slot]);
        m->buf[slot] = nullptr; m->cap[slot] = 0;
        ORBHIP_HIP_CHECK(hipMalloc(&m->buf[slot], bytes));
        m->cap[slot] = bytes;
    }
    *out = m->buf[slot];
    return ORBHIP_OK;
}

enum { S_KEYS = 0, S_DESC, S_UR, S_ORD, S_Q, S_QDESC, S_CAND, S_CNT, S_TAKEN, S_OUT, S_QKEYS, S_MISC, S_STATE, S_CSR, S_CCAND, S_NSLOTS };

static int stage_begin(orbhip_matcher *m, size_t total, Stage *st)
{
    total = al256(total) + 256;
    ORBHIP_HIP_CHECK(hipStreamSynchronize(m->stream));   // previous call's staging is free
    if (total > m->h_stage_bytes) {
        if (m->h_stage) (void)hipHostFree(m->h_stage);
        m->h_stage = nullptr; m->h_stage_bytes = 0;
        ORBHIP_HIP_CHECK(hipHostMalloc((void **)&m->h_stage, total, hipHostMallocDefault));
        m->h_stage_bytes = total;
    }
    void *d;
    int rc = scratch(m, S_MISC, total, &d);
    if (rc) return rc;
    st->h = m->h_stage; st->d = (uint8_t *)d; st->off = 0;
    return ORBHIP_OK;
}
static int stage_commit(orbhip_matcher *m, Stage *st)
{
    ORBHIP_HIP_CHECK(hipMemcpyAsync(st->d, st->h, st->off, hipMemcpyHostToDevice, m->stream));
    return ORBHIP_OK;
}
static int out_buffer(orbhip_matcher *m, size_t bytes, uint8_t **h)
{
    if (bytes > m->h_out_bytes) {
        if (m->h_out) (void)hipHostFree(m->h_out);
        m->h_out = nullptr; m->h_out_bytes = 0;
        ORBHIP_HIP_CHECK(hipHostMalloc((void **)&m->h_out, bytes, hipHostMallocDefault));
        m->h_out_bytes = bytes;
    }
    *h = m->h_out;
    return ORBHIP_OK;
}

static int ensure_resolve_attr(orbhip_matcher *m)
{
    if (!m->lds_attr_set) {   // > 64 KB of dynamic LDS needs the opt-in attribute (per device)
        ORBHIP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_resolve), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)sizeof(ResolveShared)));
        ORBHIP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_resolve_par<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             kResolveLdsBudget));
        ORBHIP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_resolve_init), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             kResolveLdsBudget));
        m->lds_attr_set = true;
    }
    return ORBHIP_OK;
}

// SearchForInitialization's resolve: the parallel kernel while its state fits the LDS budget (always, within the
// 4096-keypoint limit of the entry points), the serial replay otherwise
static void launch_resolve_init(orbhip_matcher *m, int pairs, const DevFrame &D, const orbhip_keypoint *d_qkeys, const orbhip_query *d_q,
                                int nq, int n_train, const unsigned long long *d_cand, const unsigned long long *d_ccand,
                                const int *d_cnt, int stride, float nnratio, int check_ori, int *d_out, int *d_out_n, const Batch &B)
{
    const size_t bare = init_state_bytes((size_t)n_train, (size_t)nq, 0);
    if (n_train <= kResolveMax && nq <= kResolveMax && bare <= (size_t)kResolveLdsBudget) {
        const int lcn = init_state_bytes((size_t)n_train, (size_t)nq, kResolveHead) <= (size_t)kResolveLdsBudget ? kResolveHead : 0;
        hipLaunchKernelGGL(k_resolve_init, dim3(pairs), dim3(1024), init_state_bytes((size_t)n_train, (size_t)nq, (size_t)lcn), m->stream,
                           D, d_qkeys, nq, d_cand, d_ccand, d_cnt, stride, nnratio, check_ori, d_out, d_out_n, B, n_train, nq, lcn);
    } else {
        hipLaunchKernelGGL(k_resolve, dim3(pairs), dim3(64), sizeof(ResolveShared), m->stream, 2, D, d_qkeys, d_q, nq, d_cand, d_ccand,
                           d_cnt, stride, (const uint8_t *)nullptr, nnratio, check_ori, d_out, d_out_n, B);
    }
}

// launch of the parallel resolve: LDS state up to kResolveMax train keypoints / queries, HBM state beyond
static int launch_resolve_par(orbhip_matcher *m, int pairs, int mode, const DevFrame &D, const orbhip_query *d_q, int nq,
                              int n_train, const unsigned long long *d_cand, const unsigned long long *d_ccand,
                              const int *d_cnt, int stride,
                              const uint8_t *d_taken, float nnratio, int check_ori, int *d_out, int *d_out_n, const Batch &B,
                              int th_accept, int all_block)
{
    const int threads = 1024;
    if (n_train <= kResolveMax && nq <= kResolveMax) {
        // LDS state sized by the batch's capacities; the list heads go to LDS when the budget allows
        const size_t state = resolve_par_bytes((size_t)n_train, (size_t)nq, 0);
        const int lcn = resolve_par_bytes((size_t)n_train, (size_t)nq, kResolveHead) <= (size_t)kResolveLdsBudget ? kResolveHead : 0;
        (void)state;
        hipLaunchKernelGGL(k_resolve_par<false>, dim3(pairs), dim3(threads), resolve_par_bytes((size_t)n_train, (size_t)nq, (size_t)lcn),
                           m->stream, mode, D, d_q, nq, d_cand, d_ccand, d_cnt, stride, d_taken, nnratio, check_ori, d_out, d_out_n, B,
                           th_accept, all_block, (unsigned char *)nullptr, (size_t)0, n_train, nq, lcn);
    } else {
        const size_t per = al256(resolve_par_bytes((size_t)n_train, (size_t)nq));
        void *p;
        int rc = scratch(m, S_STATE, per * (size_t)pairs, &p);
        if (rc) return rc;
        hipLaunchKernelGGL(k_resolve_par<true>, dim3(pairs), dim3(1024), 0, m->stream, mode, D, d_q, nq, d_cand, d_ccand, d_cnt,
                           stride, d_taken, nnratio, check_ori, d_out, d_out_n, B, th_accept, all_block, (unsigned char *)p, per, 0, 0, 0);
    }
    return ORBHIP_OK;
}

// CSR grid of the train frames + the cell-window search: candidates of every query
static int launch_window_search(orbhip_matcher *m, int pairs, const DevFrame &D, int n_train_cap, const orbhip_query *d_q,
                                const uint8_t *d_qdesc, int nq, unsigned long long *d_cand, int *d_cnt, int stride, int use_ur,
                                const Batch &B, unsigned long long **d_ccand_out, const uint8_t *d_taken, int max_dist,
                                const ProjLaunch *proj = nullptr)
{
    void *p;
    int rc;
    // CSR workspace per pair: cell table, records (16 B), descriptors (32 B), uRight (4 B) in CSR order
    const size_t start_bytes = al256((size_t)pairs * (kGridCells + 1) * sizeof(int));
    const size_t nrec = (size_t)pairs * n_train_cap;
    if ((rc = scratch(m, S_CSR, start_bytes + al256(nrec * sizeof(GridRec)) + al256(nrec * 32) + al256(nrec * 4), &p))) return rc;
    static_assert(sizeof(GridRec) == 16, "GridRec is one dwordx4");
    int *d_start = (int *)p;
    GridRec *d_rec = (GridRec *)((uint8_t *)p + start_bytes);
    uint4 *d_rdesc = (uint4 *)((uint8_t *)d_rec + al256(nrec * sizeof(GridRec)));
    float *d_rur = (float *)((uint8_t *)d_rdesc + al256(nrec * 32));
    if ((rc = scratch(m, S_CCAND, (size_t)pairs * nq * kCompact * sizeof(unsigned long long), &p))) return rc;
    unsigned long long *d_ccand = (unsigned long long *)p;
    if (proj) {   // the projection prologue rides in the grid launch
        const int npb = (proj->P.cap + 255) / 256;
        hipLaunchKernelGGL(k_grid_build_project, dim3(pairs * (npb + 1)), dim3(256), 0, m->stream, D, d_start, d_rec, d_rdesc, d_rur, B,
                           proj->P, proj->cam, proj->th, proj->mono, npb);
    } else {
        hipLaunchKernelGGL(k_grid_build, dim3(pairs), dim3(256), 0, m->stream, D, d_start, d_rec, d_rdesc, d_rur, B);
    }
    hipLaunchKernelGGL(k_window_search, dim3((nq + 3) / 4, pairs), dim3(256), 0, m->stream, D, d_start, d_rec, d_rdesc, d_rur, d_q,
                       d_qdesc, nq, d_cand, d_ccand, d_cnt, stride, use_ur, B, d_taken, max_dist);
    *d_ccand_out = d_ccand;
    return ORBHIP_OK;
}

// shared driver of the three windowed searches.  Modes 0 / 1 (the SearchByProjection family): queries with valid == 0
// (no map point, not in view, ...) never reach the device -- local maps and loop-closing point sets are mostly that --
// and there is no size limit (beyond kResolveMax the resolve state moves from LDS to HBM).  Mode 2
// (SearchForInitialization) replays the match stealing on one wavefront with its state in LDS: <= kResolveMax.
static int run_search(orbhip_matcher *m, int mode, const orbhip_frame_view *train, const orbhip_query *q,
                      const uint8_t *qdesc, const orbhip_keypoint *qkeys, int nq, const uint8_t *taken,
                      float nnratio, int check_ori, int32_t *out, int nout, int *nmatches, int th_accept = TH_HIGH,
                      int all_block = 0, int use_ur = 1)
{
    ORBHIP_HIP_CHECK(hipSetDevice(m->device));
    if (mode == 2 && (train->n > kResolveMax || nq > kResolveMax)) {
        set_error("SearchForInitialization: %d / %d keypoints exceed the LDS-resident limit %d", train->n, nq, kResolveMax);
        return ORBHIP_E_CAPACITY;
    }
    if (train->n >= (1 << 20)) { set_error("matcher: too many train keypoints"); return ORBHIP_E_CAPACITY; }
    for (int i = 0; i < nout; ++i) out[i] = -1;
    *nmatches = 0;
    if (nq == 0 || train->n == 0) return ORBHIP_OK;
    // stable compaction of the valid queries (order = the reference's loop order)
    static thread_local std::vector<int> vidx;
    vidx.clear();
    if (mode != 2) {
        for (int i = 0; i < nq; ++i) if (q[i].valid) vidx.push_back(i);
        if (vidx.empty()) return ORBHIP_OK;
    }
    const bool compact = mode != 2 && (int)vidx.size() < nq;
    const int nqv = compact ? (int)vidx.size() : nq;
    const size_t n = (size_t)train->n;
    Stage st;
    int rc;
    if ((rc = stage_begin(m, al256(n * sizeof(orbhip_keypoint)) + al256(n * 32) + al256(n * 4) + al256(n) +
                                 al256((size_t)nqv * sizeof(orbhip_query)) + al256((size_t)nqv * 32) +
                                 al256((size_t)nqv * sizeof(orbhip_keypoint)), &st))) return rc;
    DevFrame D;
    D.n = train->n; D.min_x = train->min_x; D.min_y = train->min_y; D.inv_w = train->grid_inv_w; D.inv_h = train->grid_inv_h;
    D.keys = (const orbhip_keypoint *)st.put(train->keys, n * sizeof(orbhip_keypoint));
    D.desc = (const uint8_t *)st.put(train->desc, n * 32);
    D.u_right = train->u_right ? (const float *)st.put(train->u_right, n * sizeof(float)) : nullptr;
    const uint8_t *d_taken = taken ? (const uint8_t *)st.put(taken, n) : nullptr;
    const orbhip_query *d_q;
    const uint8_t *d_qdesc;
    if (compact) {
        orbhip_query *hq = reinterpret_cast<orbhip_query *>(st.h + st.off);
        d_q = (const orbhip_query *)st.put(nullptr, 0);
        for (int k = 0; k < nqv; ++k) hq[k] = q[vidx[k]];
        st.off += al256((size_t)nqv * sizeof(orbhip_query));
        uint8_t *hd = st.h + st.off;
        d_qdesc = (const uint8_t *)st.put(nullptr, 0);
        for (int k = 0; k < nqv; ++k) memcpy(hd + (size_t)k * 32, qdesc + (size_t)vidx[k] * 32, 32);
        st.off += al256((size_t)nqv * 32);
    } else {
        d_q = (const orbhip_query *)st.put(q, (size_t)nq * sizeof(orbhip_query));
        d_qdesc = (const uint8_t *)st.put(qdesc, (size_t)nq * 32);
    }
    const orbhip_keypoint *d_qkeys = qkeys ? (const orbhip_keypoint *)st.put(qkeys, (size_t)nq * sizeof(orbhip_keypoint)) : nullptr;
    if ((rc = stage_commit(m, &st))) return rc;
    void *p;
    const int stride = (train->n + 1) & ~1;
    if ((rc = scratch(m, S_CAND, (size_t)nqv * stride * sizeof(unsigned long long), &p))) return rc;
    unsigned long long *d_cand = (unsigned long long *)p;
    if ((rc = scratch(m, S_CNT, (size_t)nqv * sizeof(int), &p))) return rc;
    int *d_cnt = (int *)p;
    if ((rc = scratch(m, S_OUT, (size_t)(nout + 1) * sizeof(int), &p))) return rc;
    int *d_out = (int *)p;
    uint8_t *h_out;
    if ((rc = out_buffer(m, (size_t)(nout + 1) * sizeof(int), &h_out))) return rc;
    const Batch one = {nullptr, nullptr, 0, 0};
    unsigned long long *d_ccand;
    if ((rc = launch_window_search(m, 1, D, train->n, d_q, d_qdesc, nqv, d_cand, d_cnt, stride, mode != 2 && use_ur, one, &d_ccand,
                                   mode != 2 ? d_taken : nullptr, mode == 0 ? th_accept : 256)))
        return rc;
    if ((rc = ensure_resolve_attr(m))) return rc;
    if (mode == 2)   // SearchForInitialization: match stealing depends on the running minimum distance per slot
        launch_resolve_init(m, 1, D, d_qkeys, d_q, nq, train->n, d_cand, d_ccand, d_cnt, stride, nnratio, check_ori, d_out, d_out + nout, one);
    else if ((rc = launch_resolve_par(m, 1, mode, D, d_q, nqv, train->n, d_cand, d_ccand, d_cnt, stride, d_taken, nnratio, check_ori,
                                      d_out, d_out + nout, one, th_accept, all_block)))
        return rc;
    ORBHIP_HIP_CHECK(hipGetLastError());
    ORBHIP_HIP_CHECK(hipMemcpyAsync(h_out, d_out, (size_t)(nout + 1) * sizeof(int), hipMemcpyDeviceToHost, m->stream));
    ORBHIP_HIP_CHECK(hipStreamSynchronize(m->stream));
    memcpy(out, h_out, (size_t)nout * sizeof(int));
    if (compact)   // slots hold indices into the compacted query list: map them back
        for (int i = 0; i < nout; ++i) if (out[i] >= 0) out[i] = vidx[out[i]];
    *nmatches = reinterpret_cast<const int *>(h_out)[nout];
    return ORBHIP_OK;
}

// processing order of the reference: FeatureVector nodes ascending, feature indices ascending inside a node
static void node_order(const uint32_t *node, const uint8_t *valid, int n, std::vector<int> &order)
{
    static thread_local std::vector<unsigned long long> keys;
    keys.clear();
    keys.reserve(n);
    for (int i = 0; i < n; ++i)
        if (node[i] != ORBHIP_NO_NODE && (!valid || valid[i])) keys.push_back(((unsigned long long)node[i] << 32) | (unsigned)i);
    std::sort(keys.begin(), keys.end());
    order.resize(keys.size());
    for (size_t k = 0; k < keys.size(); ++k) order[k] = (int)(uint32_t)keys[k];
}

// SearchByBoW, both overloads
static int run_bow(orbhip_matcher *m, const orbhip_frame_view *f1, const uint32_t *node1, const uint8_t *valid1,
                   const orbhip_frame_view *f2, const uint32_t *node2, const uint8_t *blocked2, int max_dist,
                   float nnratio, int check_ori, int32_t *matches12, int *nmatches)
{
    ORBHIP_HIP_CHECK(hipSetDevice(m->device));
    const int n1 = f1->n, n2 = f2->n;
    for (int i = 0; i < n1; ++i) matches12[i] = -1;
    *nmatches = 0;
    if (n1 == 0 || n2 == 0) return ORBHIP_OK;
    std::vector<int> order, torder;
    node_order(node1, valid1, n1, order);
    node_order(node2, nullptr, n2, torder);   // blocked features keep their slot: they only start out "matched"
    const int nq = (int)order.size(), nt = (int)torder.size();
    if (nq == 0 || nt == 0) return ORBHIP_OK;
    std::vector<NodeGroup> groups;            // common nodes, the merge of ORBmatcher.cc:185-286
    for (int a = 0, b = 0; a < nq && b < nt;) {
        const uint32_t na = node1[order[a]], nb = node2[torder[b]];
        if (na == nb) {
            int ae = a, be = b;
            while (ae < nq && node1[order[ae]] == na) ++ae;
            while (be < nt && node2[torder[be]] == na) ++be;
            groups.push_back(NodeGroup{a, ae, b, be});
            a = ae; b = be;
        } else if (na < nb) ++a;
        else ++b;
    }
    const int ng = (int)groups.size();
    if (ng == 0) return ORBHIP_OK;
    const size_t n = (size_t)n2;
    Stage st;
    int rc;
    if ((rc = stage_begin(m, al256(n * sizeof(orbhip_keypoint)) + al256(n * 32) + al256((size_t)nt * 4) + al256((size_t)nt) +
                                 al256((size_t)nq * 32) + al256((size_t)nq * 4) + al256((size_t)ng * sizeof(NodeGroup)) +
                                 al256(HISTO_LENGTH * 4), &st))) return rc;
    const orbhip_keypoint *d_tkeys = (const orbhip_keypoint *)st.put(f2->keys, n * sizeof(orbhip_keypoint));
    const uint8_t *d_tdesc = (const uint8_t *)st.put(f2->desc, n * 32);
    const uint32_t *d_torder = (const uint32_t *)st.put(torder.data(), (size_t)nt * 4);
    uint8_t *hm = st.h + st.off;
    uint8_t *d_matched = (uint8_t *)st.put(nullptr, 0);
    for (int c = 0; c < nt; ++c) hm[c] = (uint8_t)(blocked2 && blocked2[torder[c]]);
    st.off += al256((size_t)nt);
    uint8_t *hqd = st.h + st.off;
    const uint8_t *d_qdesc = (const uint8_t *)st.put(nullptr, 0);
    st.off += al256((size_t)nq * 32);
    float *hqa = reinterpret_cast<float *>(st.h + st.off);
    const float *d_qangle = (const float *)st.put(nullptr, 0);
    st.off += al256((size_t)nq * 4);
    for (int p = 0; p < nq; ++p) {
        memcpy(hqd + (size_t)p * 32, f1->desc + (size_t)order[p] * 32, 32);
        hqa[p] = f1->keys[order[p]].angle;
    }
    const NodeGroup *d_groups = (const NodeGroup *)st.put(groups.data(), (size_t)ng * sizeof(NodeGroup));
    memset(st.h + st.off, 0, HISTO_LENGTH * 4);
    int *d_hist = (int *)st.put(nullptr, 0);
    st.off += al256(HISTO_LENGTH * 4);
    if ((rc = stage_commit(m, &st))) return rc;
    void *p;
    if ((rc = scratch(m, S_OUT, (size_t)(nq + 1) * sizeof(int), &p))) return rc;
    int *d_out = (int *)p;
    if ((rc = scratch(m, S_CNT, (size_t)nq, &p))) return rc;
    uint8_t *d_bin = (uint8_t *)p;
    uint8_t *h_out;
    if ((rc = out_buffer(m, (size_t)(nq + 1) * sizeof(int), &h_out))) return rc;
    // queries whose node does not occur in f2 belong to no group: they stay at -1
    ORBHIP_HIP_CHECK(hipMemsetAsync(d_out, 0xff, (size_t)nq * sizeof(int), m->stream));
    hipLaunchKernelGGL(k_bow_groups, dim3((ng + 3) / 4), dim3(256), 0, m->stream, d_groups, ng, d_qdesc, d_qangle, d_torder,
                       d_tdesc, d_tkeys, d_matched, max_dist, nnratio, check_ori, d_out, d_bin, d_hist);
    hipLaunchKernelGGL(k_bow_cull, dim3(1), dim3(1024), 0, m->stream, nq, check_ori, (const int *)d_hist,
                       (const uint8_t *)d_bin, d_out, d_out + nq);
    ORBHIP_HIP_CHECK(hipGetLastError());
    ORBHIP_HIP_CHECK(hipMemcpyAsync(h_out, d_out, (size_t)(nq + 1) * sizeof(int), hipMemcpyDeviceToHost, m->stream));
    ORBHIP_HIP_CHECK(hipStreamSynchronize(m->stream));
    const int *res = reinterpret_cast<const int *>(h_out);
    for (const NodeGroup &G : groups)
        for (int q = G.q_begin; q < G.q_end; ++q) matches12[order[q]] = res[q];
    *nmatches = res[nq];
    return ORBHIP_OK;
}

// SearchForTriangulation
static int run_tri(orbhip_matcher *m, const TriParams *tri, const orbhip_frame_view *f1, const uint32_t *node1,
                   const uint8_t *valid1, const orbhip_frame_view *f2, const uint32_t *node2, const uint8_t *valid2,
                   int only_stereo, int check_ori, int32_t *matches12, int *nmatches)
{
    ORBHIP_HIP_CHECK(hipSetDevice(m->device));
    const int n1 = f1->n, n2 = f2->n;
    if (n1 > kResolveMax || n2 > kResolveMax) {
        set_error("matcher: %d / %d keypoints exceed the LDS-resident limit %d", n1, n2, kResolveMax);
        return ORBHIP_E_CAPACITY;
    }
    for (int i = 0; i < n1; ++i) matches12[i] = -1;
    *nmatches = 0;
    if (n1 == 0 || n2 == 0) return ORBHIP_OK;
    std::vector<int> order;
    node_order(node1, valid1, n1, order);
    if (only_stereo) {   // :706-708
        size_t w = 0;
        for (size_t r = 0; r < order.size(); ++r)
            if (f1->u_right && f1->u_right[order[r]] >= 0) order[w++] = order[r];
        order.resize(w);
    }
    const int nq = (int)order.size();
    if (nq == 0) return ORBHIP_OK;
    const size_t n = (size_t)n2;
    Stage st;
    int rc;
    if ((rc = stage_begin(m, al256(n * sizeof(orbhip_keypoint)) + al256(n * 32) + 2 * al256(n * 4) + al256(n) +
                                 al256((size_t)nq * sizeof(orbhip_query)) + al256((size_t)nq * 32), &st))) return rc;
    DevFrame D;
    D.n = n2; D.min_x = D.min_y = 0.f; D.inv_w = D.inv_h = 0.f;
    D.keys = (const orbhip_keypoint *)st.put(f2->keys, n * sizeof(orbhip_keypoint));
    D.desc = (const uint8_t *)st.put(f2->desc, n * 32);
    D.u_right = f2->u_right ? (const float *)st.put(f2->u_right, n * sizeof(float)) : nullptr;
    const uint32_t *d_tnode = (const uint32_t *)st.put(node2, n * sizeof(uint32_t));
    const uint8_t *d_mask = nullptr;   // candidate filter: no map point yet (+ bOnlyStereo :725-729)
    if (valid2 || only_stereo) {
        uint8_t *hm = st.h + st.off;
        d_mask = (const uint8_t *)st.put(nullptr, 0);
        for (int j = 0; j < n2; ++j)
            hm[j] = (uint8_t)((!valid2 || valid2[j]) && (!only_stereo || (f2->u_right && f2->u_right[j] >= 0)));
        st.off += al256(n);
    }
    orbhip_query *hq = reinterpret_cast<orbhip_query *>(st.h + st.off);
    const orbhip_query *d_q = (const orbhip_query *)st.put(nullptr, 0);
    st.off += al256((size_t)nq * sizeof(orbhip_query));
    uint8_t *hqd = st.h + st.off;
    const uint8_t *d_qdesc = (const uint8_t *)st.put(nullptr, 0);
    st.off += al256((size_t)nq * 32);
    for (int p = 0; p < nq; ++p) {
        const int i = order[p];
        orbhip_query &Q = hq[p];
        memset(&Q, 0, sizeof(Q));
        Q.valid = 1;
        Q.u = f1->keys[i].x; Q.v = f1->keys[i].y;
        Q.ur = f1->u_right ? f1->u_right[i] : -1.0f;
        Q.level_aux = (int32_t)node1[i];
        Q.angle = f1->keys[i].angle;
        memcpy(hqd + (size_t)p * 32, f1->desc + (size_t)i * 32, 32);
    }
    if ((rc = stage_commit(m, &st))) return rc;
    void *p;
    const int stride = 2;
    if ((rc = scratch(m, S_CAND, (size_t)nq * stride * sizeof(unsigned long long), &p))) return rc;
    unsigned long long *d_cand = (unsigned long long *)p;
    if ((rc = scratch(m, S_CNT, (size_t)nq * sizeof(int), &p))) return rc;
    int *d_cnt = (int *)p;
    if ((rc = scratch(m, S_OUT, (size_t)(nq + 1) * sizeof(int), &p))) return rc;
    int *d_out = (int *)p;
    uint8_t *h_out;
    if ((rc = out_buffer(m, (size_t)(nq + 1) * sizeof(int), &h_out))) return rc;
    const Batch one = {nullptr, nullptr, 0, 0};
    if ((rc = ensure_resolve_attr(m))) return rc;
    hipLaunchKernelGGL(k_tri_search, dim3((nq + 3) / 4), dim3(256), 0, m->stream, D, d_tnode, d_mask, d_q, d_qdesc, nq,
                       d_cand, d_cnt, stride, *tri);
    if ((rc = launch_resolve_par(m, 1, 4, D, d_q, nq, f2->n, d_cand, nullptr, d_cnt, stride, (const uint8_t *)nullptr, 0.f, check_ori,
                                 d_out, d_out + nq, one, TH_LOW, 0)))
        return rc;
    ORBHIP_HIP_CHECK(hipGetLastError());
    ORBHIP_HIP_CHECK(hipMemcpyAsync(h_out, d_out, (size_t)(nq + 1) * sizeof(int), hipMemcpyDeviceToHost, m->stream));
    ORBHIP_HIP_CHECK(hipStreamSynchronize(m->stream));
    const int *res = reinterpret_cast<const int *>(h_out);
    for (int q = 0; q < nq; ++q) matches12[order[q]] = res[q];
    *nmatches = res[nq];
    return ORBHIP_OK;
}

extern "C" {

int orbhip_matcher_create(int device, orbhip_matcher **out)
{
    if (!out) return ORBHIP_E_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        set_error("no HIP device %d (found %d)", device, ndev);
        return ORBHIP_E_NODEVICE;
    }
    orbhip_matcher *m = new (std::nothrow) orbhip_matcher();
    if (!m) return ORBHIP_E_ARG;
    m->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&m->own_stream, hipStreamNonBlocking) != hipSuccess) {
        set_error("hipSetDevice/hipStreamCreate failed");
        delete m;
        return ORBHIP_E_HIP;
    }
    m->stream = m->own_stream;
    *out = m;
    return ORBHIP_OK;
}

void orbhip_matcher_destroy(orbhip_matcher *m)
{
    if (!m) return;
    (void)hipSetDevice(m->device);
    if (m->stream) (void)hipStreamSynchronize(m->stream);
    for (int i = 0; i < S_NSLOTS; ++i) (void)hipFree(m->buf[i]);
    if (m->h_stage) (void)hipHostFree(m->h_stage);
    if (m->h_out) (void)hipHostFree(m->h_out);
    if (m->own_stream) (void)hipStreamDestroy(m->own_stream);
    delete m;
}

int orbhip_descriptor_distance(orbhip_matcher *m, const uint8_t *a, int na, const uint8_t *b, int nb, int32_t *dist)
{
    if (!m || !a || !b || !dist || na < 0 || nb < 0) return ORBHIP_E_ARG;
    if (na == 0 || nb == 0) return ORBHIP_OK;
    ORBHIP_HIP_CHECK(hipSetDevice(m->device));
    void *pa, *pb, *pd;
    int rc;
    if ((rc = scratch(m, S_DESC, (size_t)na * 32, &pa))) return rc;
    if ((rc = scratch(m, S_QDESC, (size_t)nb * 32, &pb))) return rc;
    if ((rc = scratch(m, S_CAND, (size_t)na * nb * sizeof(int), &pd))) return rc;
    ORBHIP_HIP_CHECK(hipMemcpyAsync(pa, a, (size_t)na * 32, hipMemcpyHostToDevice, m->stream));
    ORBHIP_HIP_CHECK(hipMemcpyAsync(pb, b, (size_t)nb * 32, hipMemcpyHostToDevice, m->stream));
    hipLaunchKernelGGL(k_distance_matrix, dim3((nb + 255) / 256, na), dim3(256), 0, m->stream, (const uint8_t *)pa, na,
                       (const uint8_t *)pb, nb, (int *)pd);
    ORBHIP_HIP_CHECK(hipGetLastError());
    ORBHIP_HIP_CHECK(hipMemcpyAsync(dist, pd, (size_t)na * nb * sizeof(int), hipMemcpyDeviceToHost, m->stream));
    ORBHIP_HIP_CHECK(hipStreamSynchronize(m->stream));
    return ORBHIP_OK;
}

int orbhip_search_for_initialization(orbhip_matcher *m, const orbhip_frame_view *f1, const orbhip_frame_view *f2,
                                     float *prev_matched_xy, int32_t *matches12, int window_size, float nnratio,
                                     int check_ori, int *nmatches)
{
    if (!m || !f1 || !f2 || !prev_matched_xy || !matches12 || !nmatches) return ORBHIP_E_ARG;
    const int n1 = f1->n;
    // queries: level-0 keypoints of F1 searched around vbPrevMatched (ORBmatcher.cc:418-425)
    std::vector<orbhip_query> q((size_t)std::max(n1, 1));
    for (int i = 0; i < n1; ++i) {
        orbhip_query &Q = q[i];
        memset(&Q, 0, sizeof(Q));
        const int level1 = f1->keys[i].octave;
        Q.valid = level1 > 0 ? 0 : 1;
        Q.u = prev_matched_xy[2 * i]; Q.v = prev_matched_xy[2 * i + 1];
        Q.radius = (float)window_size;
        Q.min_level = level1; Q.max_level = level1;
        Q.angle = f1->keys[i].angle;
    }
    int rc = run_search(m, 2, f2, q.data(), f1->desc, f1->keys, n1, nullptr, nnratio, check_ori, matches12, n1, nmatches);
    if (rc) return rc;
    for (int i = 0; i < n1; ++i)  // :515-517
        if (matches12[i] >= 0) {
            prev_matched_xy[2 * i] = f2->keys[matches12[i]].x;
            prev_matched_xy[2 * i + 1] = f2->keys[matches12[i]].y;
        }
    return ORBHIP_OK;
}

int orbhip_search_by_projection_frame(orbhip_matcher *m, const orbhip_frame_view *cur, const orbhip_query *q,
                                      const uint8_t *qdesc, int nq, const uint8_t *taken, int32_t *assign,
                                      int check_ori, int *nmatches)
{
    if (!m || !cur || (nq > 0 && (!q || !qdesc)) || !assign || !nmatches || nq < 0) return ORBHIP_E_ARG;
    return run_search(m, 0, cur, q, qdesc, nullptr, nq, taken, 0.f, check_ori, assign, cur->n, nmatches);
}

int orbhip_search_by_projection_keyframe(orbhip_matcher *m, const orbhip_frame_view *cur, const orbhip_query *q,
                                         const uint8_t *qdesc, int nq, const uint8_t *taken, int32_t *assign,
                                         int orb_dist, int check_ori, int *nmatches)
{
    if (!m || !cur || (nq > 0 && (!q || !qdesc)) || !assign || !nmatches || nq < 0) return ORBHIP_E_ARG;
    return run_search(m, 0, cur, q, qdesc, nullptr, nq, taken, 0.f, check_ori, assign, cur->n, nmatches, orb_dist, 1, 0);
}

int orbhip_search_by_projection_sim3(orbhip_matcher *m, const orbhip_frame_view *kf, const orbhip_query *q,
                                     const uint8_t *qdesc, int nq, const uint8_t *matched, int32_t *assign, int *nmatches)
{
    if (!m || !kf || (nq > 0 && (!q || !qdesc)) || !assign || !nmatches || nq < 0) return ORBHIP_E_ARG;
    return run_search(m, 0, kf, q, qdesc, nullptr, nq, matched, 0.f, 0, assign, kf->n, nmatches, TH_LOW, 1, 0);
}

int orbhip_assign_features_to_grid_device(orbhip_matcher *m, int frames, const void *d_kps, const void *d_n, int cap,
                                          float min_x, float min_y, float grid_inv_w, float grid_inv_h, void *d_cell_of,
                                          void *d_cell_start, void *d_cell_items)
{
    if (!m || frames < 0 || cap < 1 || !d_kps || !d_n || !d_cell_of || !d_cell_start || !d_cell_items) return ORBHIP_E_ARG;
    if (cap > kGridMax) { set_error("assign_features_to_grid: capacity %d exceeds %d", cap, kGridMax); return ORBHIP_E_CAPACITY; }
    if (frames == 0) return ORBHIP_OK;
    ORBHIP_HIP_CHECK(hipSetDevice(m->device));
    DevFrame D;
    D.n = 0; D.keys = (const orbhip_keypoint *)d_kps; D.desc = nullptr; D.u_right = nullptr;
    D.min_x = min_x; D.min_y = min_y; D.inv_w = grid_inv_w; D.inv_h = grid_inv_h;
    const Batch B = {(const int *)d_n, nullptr, cap, 0};
    hipLaunchKernelGGL(k_grid_csr, dim3(frames), dim3(1024), 0, m->stream, D, B, (int *)d_cell_of, (int *)d_cell_start,
                       (int *)d_cell_items);
    ORBHIP_HIP_CHECK(hipGetLastError());
    return ORBHIP_OK;
}

int orbhip_assign_features_to_grid(orbhip_matcher *m, const orbhip_frame_view *f, int32_t *cell_of, int32_t *cell_start,
                                   int32_t *cell_items)
{
    if (!m || !f || f->n < 0 || !cell_start || (f->n > 0 && (!f->keys || !cell_of || !cell_items))) return ORBHIP_E_ARG;
    const int n = f->n;
    if (n > kGridMax) { set_error("assign_features_to_grid: %d keypoints exceed %d", n, kGridMax); return ORBHIP_E_CAPACITY; }
    if (n == 0) { for (int c = 0; c <= kGridCells; ++c) cell_start[c] = 0; return ORBHIP_OK; }
    ORBHIP_HIP_CHECK(hipSetDevice(m->device));
    Stage st;
    int rc;
    if ((rc = stage_begin(m, al256((size_t)n * sizeof(orbhip_keypoint)) + 256, &st))) return rc;
    const void *d_keys = st.put(f->keys, (size_t)n * sizeof(orbhip_keypoint));
    int *hn = reinterpret_cast<int *>(st.h + st.off);
    *hn = n;
    const void *d_n = st.put(nullptr, 0);
    st.off += 256;
    if ((rc = stage_commit(m, &st))) return rc;
    const size_t ob = ((size_t)2 * n + kGridCells + 1) * sizeof(int);
    void *p;
    if ((rc = scratch(m, S_OUT, ob, &p))) return rc;
    int *d_out = (int *)p;
    uint8_t *h_out;
    if ((rc = out_buffer(m, ob, &h_out))) return rc;
    if ((rc = orbhip_assign_features_to_grid_device(m, 1, d_keys, d_n, n, f->min_x, f->min_y, f->grid_inv_w, f->grid_inv_h,
                                                    d_out, d_out + 2 * n, d_out + n))) return rc;
    ORBHIP_HIP_CHECK(hipMemcpyAsync(h_out, d_out, ob, hipMemcpyDeviceToHost, m->stream));
    ORBHIP_HIP_CHECK(hipStreamSynchronize(m->stream));
    const int *r = reinterpret_cast<const int *>(h_out);
    memcpy(cell_of, r, (size_t)n * sizeof(int));
    memcpy(cell_items, r + n, (size_t)n * sizeof(int));
    memcpy(cell_start, r + 2 * n, (size_t)(kGridCells + 1) * sizeof(int));
    return ORBHIP_OK;
}

int orbhip_undistort_keypoints_device(orbhip_matcher *m, int frames, const void *d_kps, const void *d_n, int cap, float fx,
                                      float fy, float cx, float cy, const float *dist5, void *d_kps_un)
{
    if (!m || frames < 0 || cap < 1 || !d_kps || !d_n || !dist5 || !d_kps_un || fx == 0.f || fy == 0.f) return ORBHIP_E_ARG;
    if (frames == 0) return ORBHIP_OK;
    ORBHIP_HIP_CHECK(hipSetDevice(m->device));
    if (dist5[0] == 0.0f) {   // mvKeysUn = mvKeys (:406-410)
        if (d_kps_un != d_kps)
            ORBHIP_HIP_CHECK(hipMemcpyAsync(d_kps_un, d_kps, (size_t)frames * cap * sizeof(orbhip_keypoint), hipMemcpyDeviceToDevice, m->stream));
        return ORBHIP_OK;
    }
    UndistortParams P;
    P.fx = fx; P.fy = fy; P.cx = cx; P.cy = cy;
    for (int i = 0; i < 5; ++i) P.k[i] = dist5[i];
    hipLaunchKernelGGL(k_undistort, dim3((cap + 255) / 256, frames), dim3(256), 0, m->stream, (const orbhip_keypoint *)d_kps,
                       (const int *)d_n, 0, cap, P, (orbhip_keypoint *)d_kps_un);
    ORBHIP_HIP_CHECK(hipGetLastError());
    return ORBHIP_OK;
}

int orbhip_undistort_keypoints(orbhip_matcher *m, const orbhip_keypoint *keys, int n, float fx, float fy, float cx, float cy,
                               const float *dist5, orbhip_keypoint *keys_un)
{
    if (!m || n < 0 || (n > 0 && (!keys || !keys_un)) || !dist5 || fx == 0.f || fy == 0.f) return ORBHIP_E_ARG;
    if (n == 0) return ORBHIP_OK;
    ORBHIP_HIP_CHECK(hipSetDevice(m->device));
    Stage st;
    int rc;
    if ((rc = stage_begin(m, al256((size_t)n * sizeof(orbhip_keypoint)) + 256, &st))) return rc;
    const void *d_keys = st.put(keys, (size_t)n * sizeof(orbhip_keypoint));
    int *hn = reinterpret_cast<int *>(st.h + st.off);
    *hn = n;
    const void *d_n = st.put(nullptr, 0);
    st.off += 256;
    if ((rc = stage_commit(m, &st))) return rc;
    void *p;
    if ((rc = scratch(m, S_OUT, (size_t)n * sizeof(orbhip_keypoint), &p))) return rc;
    uint8_t *h_out;
    if ((rc = out_buffer(m, (size_t)n * sizeof(orbhip_keypoint), &h_out))) return rc;
    if ((rc = orbhip_undistort_keypoints_device(m, 1, d_keys, d_n, n, fx, fy, cx, cy, dist5, p))) return rc;
    ORBHIP_HIP_CHECK(hipMemcpyAsync(h_out, p, (size_t)n * sizeof(orbhip_keypoint), hipMemcpyDeviceToHost, m->stream));
    ORBHIP_HIP_CHECK(hipStreamSynchronize(m->stream));
    memcpy(keys_un, h_out, (size_t)n * sizeof(orbhip_keypoint));
    return ORBHIP_OK;
}

int orbhip_compute_stereo_from_rgbd_device(orbhip_matcher *m, int frames, const void *d_kps, const void *d_kps_un,
                                           const void *d_n, int cap, const void *d_depth, int rows, int cols,
                                           int stride_floats, size_t frame_stride_floats, float mbf, void *d_u_right,
                                           void *d_depth_out)
{
    if (!m || frames < 0 || cap < 1 || !d_kps || !d_n || !d_depth || rows < 1 || cols < 1 || stride_floats < cols ||
        !d_u_right || !d_depth_out)
        return ORBHIP_E_ARG;
    if (frames == 0) return ORBHIP_OK;
    ORBHIP_HIP_CHECK(hipSetDevice(m->device));
    hipLaunchKernelGGL(k_stereo_from_rgbd, dim3((cap + 255) / 256, frames), dim3(256), 0, m->stream,
                       (const orbhip_keypoint *)d_kps, (const orbhip_keypoint *)(d_kps_un ? d_kps_un : d_kps), (const int *)d_n,
                       0, cap, (const float *)d_depth, rows, cols, stride_floats, frame_stride_floats, mbf, (float *)d_u_right,
                       (float *)d_depth_out);
    ORBHIP_HIP_CHECK(hipGetLastError());
    return ORBHIP_OK;
}

int orbhip_compute_stereo_from_rgbd(orbhip_matcher *m, const orbhip_keypoint *keys, const orbhip_keypoint *keys_un, int n,
                                    const float *depth, int rows, int cols, int stride_floats, float mbf, float *u_right,
                                    float *depth_out)
{
    if (!m || n < 0 || (n > 0 && (!keys || !u_right || !depth_out)) || !depth || rows < 1 || cols < 1 || stride_floats < cols)
        return ORBHIP_E_ARG;
    if (n == 0) return ORBHIP_OK;
    ORBHIP_HIP_CHECK(hipSetDevice(m->device));
    if (!keys_un) keys_un = keys;
    // the depth image travels like the grey image does for extraction: packed rows through the pinned staging buffer
    Stage st;
    int rc;
    const size_t kb = al256((size_t)n * sizeof(orbhip_keypoint)), ib = al256((size_t)rows * cols * sizeof(float));
    if ((rc = stage_begin(m, 2 * kb + ib + 256, &st))) return rc;
    const void *d_keys = st.put(keys, (size_t)n * sizeof(orbhip_keypoint));
    const void *d_un = st.put(keys_un, (size_t)n * sizeof(orbhip_keypoint));
    float *hd = reinterpret_cast<float *>(st.h + st.off);
    const void *d_depth = st.put(nullptr, 0);
    for (int r = 0; r < rows; ++r) memcpy(hd + (size_t)r * cols, depth + (size_t)r * stride_floats, (size_t)cols * sizeof(float));
    st.off += ib;
    int *hn = reinterpret_cast<int *>(st.h + st.off);
    *hn = n;
    const void *d_n = st.put(nullptr, 0);
    st.off += 256;
    if ((rc = stage_commit(m, &st))) return rc;
    void *p;
    if ((rc = scratch(m, S_OUT, (size_t)2 * n * sizeof(float), &p))) return rc;
    float *d_out = (float *)p;
    uint8_t *h_out;
    if ((rc = out_buffer(m, (size_t)2 * n * sizeof(float), &h_out))) return rc;
    if ((rc = orbhip_compute_stereo_from_rgbd_device(m, 1, d_keys, d_un, d_n, n, d_depth, rows, cols, cols, 0, mbf, d_out,
                                                     d_out + n))) return rc;
    ORBHIP_HIP_CHECK(hipMemcpyAsync(h_out, d_out, (size_t)2 * n * sizeof(float), hipMemcpyDeviceToHost, m->stream));
    ORBHIP_HIP_CHECK(hipStreamSynchronize(m->stream));
    memcpy(u_right, h_out, (size_t)n * sizeof(float));
    memcpy(depth_out, h_out + (size_t)n * sizeof(float), (size_t)n * sizeof(float));
    return ORBHIP_OK;
}

int orbhip_distinctive_descriptors(orbhip_matcher *m, const uint8_t *desc, const int32_t *offsets, int npoints,
                                   int32_t *best_idx)
{
    if (!m || npoints < 0 || (npoints > 0 && (!offsets || !best_idx))) return ORBHIP_E_ARG;
    if (npoints == 0) return ORBHIP_OK;
    if (offsets[0] < 0) return ORBHIP_E_ARG;
    for (int p = 0; p < npoints; ++p) {
        const long long N = (long long)offsets[p + 1] - offsets[p];
        if (N < 0) { set_error("distinctive_descriptors: offsets must be non-decreasing (point %d)", p); return ORBHIP_E_ARG; }
        if (N > kDistinctMax) {
            set_error("distinctive_descriptors: map point %d has %lld observations (limit %d)", p, N, kDistinctMax);
            return ORBHIP_E_CAPACITY;
        }
    }
    const size_t total = (size_t)offsets[npoints];
    if (total > 0 && !desc) return ORBHIP_E_ARG;
    ORBHIP_HIP_CHECK(hipSetDevice(m->device));
    Stage st;
    int rc;
    if ((rc = stage_begin(m, al256(total * 32) + al256((size_t)(npoints + 1) * 4), &st))) return rc;
    const uint8_t *d_desc = (const uint8_t *)st.put(desc, total * 32);
    const int *d_off = (const int *)st.put(offsets, (size_t)(npoints + 1) * 4);
    if ((rc = stage_commit(m, &st))) return rc;
    void *p;
    if ((rc = scratch(m, S_OUT, (size_t)npoints * sizeof(int), &p))) return rc;
    uint8_t *h_out;
    if ((rc = out_buffer(m, (size_t)npoints * sizeof(int), &h_out))) return rc;
    hipLaunchKernelGGL(k_distinctive, dim3((npoints + 3) / 4), dim3(256), 0, m->stream, d_desc, d_off, npoints, (int *)p);
    ORBHIP_HIP_CHECK(hipGetLastError());
    ORBHIP_HIP_CHECK(hipMemcpyAsync(h_out, p, (size_t)npoints * sizeof(int), hipMemcpyDeviceToHost, m->stream));
    ORBHIP_HIP_CHECK(hipStreamSynchronize(m->stream));
    memcpy(best_idx, h_out, (size_t)npoints * sizeof(int));
    return ORBHIP_OK;
}

int orbhip_search_by_bow(orbhip_matcher *m, const orbhip_frame_view *f1, const uint32_t *node1, const uint8_t *valid1,
                         const orbhip_frame_view *f2, const uint32_t *node2, const uint8_t *blocked2, int max_dist,
                         float nnratio, int check_ori, int32_t *matches12, int *nmatches)
{
    if (!m || !f1 || !f2 || !matches12 || !nmatches || f1->n < 0 || f2->n < 0) return ORBHIP_E_ARG;
    if ((f1->n > 0 && (!node1 || !f1->keys || !f1->desc)) || (f2->n > 0 && (!node2 || !f2->keys || !f2->desc)))
        return ORBHIP_E_ARG;
    return run_bow(m, f1, node1, valid1, f2, node2, blocked2, max_dist, nnratio, check_ori, matches12, nmatches);
}

int orbhip_search_by_bow_device(orbhip_matcher *m, int pairs, int cap, const void *d_kps1, const void *d_desc1,
                                const void *d_n1, const void *d_node1, const void *d_valid1, int f1_first, int f1_step,
                                const void *d_kps2, const void *d_desc2, const void *d_n2, const void *d_node2,
                                const void *d_blocked2, int f2_first, int f2_step, int max_dist, float nnratio,
                                int check_ori, void *d_matches12, void *d_nmatches)
{
    if (!m || pairs < 0 || cap < 1 || !d_kps1 || !d_desc1 || !d_n1 || !d_node1 || !d_kps2 || !d_desc2 || !d_n2 || !d_node2 ||
        !d_matches12 || !d_nmatches || f1_first < 0 || f2_first < 0 || f1_step < 0 || f2_step < 0)
        return ORBHIP_E_ARG;
    if (cap > kBowPairMax) {
        set_error("search_by_bow_device: capacity %d exceeds the LDS-resident limit %d", cap, kBowPairMax);
        return ORBHIP_E_CAPACITY;
    }
    if (pairs == 0) return ORBHIP_OK;
    ORBHIP_HIP_CHECK(hipSetDevice(m->device));
    if (!m->bow_attr_set) {
        ORBHIP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_bow_pairs),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(BowPairShared)));
        m->bow_attr_set = true;
    }
    const BowSide Q = {(const orbhip_keypoint *)d_kps1, (const uint8_t *)d_desc1, (const int *)d_n1, (const uint32_t *)d_node1,
                       (const uint8_t *)d_valid1, f1_first, f1_step};
    const BowSide T = {(const orbhip_keypoint *)d_kps2, (const uint8_t *)d_desc2, (const int *)d_n2, (const uint32_t *)d_node2,
                       (const uint8_t *)d_blocked2, f2_first, f2_step};
    hipLaunchKernelGGL(k_bow_pairs, dim3(pairs), dim3(1024), sizeof(BowPairShared), m->stream, Q, T, cap, max_dist, nnratio,
                       check_ori, (int *)d_matches12, (int *)d_nmatches);
    ORBHIP_HIP_CHECK(hipGetLastError());
    return ORBHIP_OK;
}

int orbhip_search_for_triangulation(orbhip_matcher *m, const orbhip_frame_view *f1, const uint32_t *node1,
                                    const uint8_t *valid1, const orbhip_frame_view *f2, const uint32_t *node2,
                                    const uint8_t *valid2, const float *f12, float ex, float ey,
                                    const float *level_sigma2, int only_stereo, int check_ori, int32_t *matches12,
                                    int *nmatches)
{
    if (!m || !f1 || !f2 || !f12 || !level_sigma2 || !matches12 || !nmatches || f1->n < 0 || f2->n < 0) return ORBHIP_E_ARG;
    if ((f1->n > 0 && (!node1 || !f1->keys || !f1->desc)) || (f2->n > 0 && (!node2 || !f2->keys || !f2->desc)))
        return ORBHIP_E_ARG;
    if (f2->n_levels < 1 || f2->n_levels > ORBHIP_MAX_LEVELS || !f2->scale_factors) {
        set_error("search_for_triangulation: f2 needs n_levels in [1,%d] and scale_factors", ORBHIP_MAX_LEVELS);
        return ORBHIP_E_ARG;
    }
    TriParams P;
    memset(&P, 0, sizeof(P));
    for (int i = 0; i < 9; ++i) P.f12[i] = f12[i];
    P.ex = ex; P.ey = ey;
    for (int l = 0; l < f2->n_levels; ++l) { P.sigma2[l] = level_sigma2[l]; P.sf[l] = f2->scale_factors[l]; }
    for (int j = 0; j < f2->n; ++j)
        if (f2->keys[j].octave < 0 || f2->keys[j].octave >= f2->n_levels) {
            set_error("search_for_triangulation: keypoint %d has octave %d outside [0,%d)", j, f2->keys[j].octave, f2->n_levels);
            return ORBHIP_E_ARG;
        }
    return run_tri(m, &P, f1, node1, valid1, f2, node2, valid2, only_stereo, check_ori, matches12, nmatches);
}

int orbhip_search_best_in_window(orbhip_matcher *m, const orbhip_frame_view *kf, const orbhip_query *q,
                                 const uint8_t *qdesc, int nq, int chi2_gate, const float *inv_level_sigma2,
                                 int32_t *best_idx, int32_t *best_dist)
{
    if (!m || !kf || nq < 0 || (nq > 0 && (!q || !qdesc || !best_idx || !best_dist)) || (chi2_gate && !inv_level_sigma2))
        return ORBHIP_E_ARG;
    for (int i = 0; i < nq; ++i) { best_idx[i] = -1; best_dist[i] = 256; }
    if (nq == 0 || kf->n == 0) return ORBHIP_OK;
    if (kf->n >= (1 << 20)) { set_error("too many keypoints"); return ORBHIP_E_CAPACITY; }
    ORBHIP_HIP_CHECK(hipSetDevice(m->device));
    const size_t n = (size_t)kf->n;
    Stage st;
    int rc;
    if ((rc = stage_begin(m, al256(n * sizeof(orbhip_keypoint)) + al256(n * 32) + al256(n * 4) +
                                 al256((size_t)nq * sizeof(orbhip_query)) + al256((size_t)nq * 32), &st))) return rc;
    DevFrame D;
    D.n = kf->n; D.min_x = kf->min_x; D.min_y = kf->min_y; D.inv_w = kf->grid_inv_w; D.inv_h = kf->grid_inv_h;
    D.keys = (const orbhip_keypoint *)st.put(kf->keys, n * sizeof(orbhip_keypoint));
    D.desc = (const uint8_t *)st.put(kf->desc, n * 32);
    D.u_right = kf->u_right ? (const float *)st.put(kf->u_right, n * sizeof(float)) : nullptr;
    const orbhip_query *d_q = (const orbhip_query *)st.put(q, (size_t)nq * sizeof(orbhip_query));
    const uint8_t *d_qdesc = (const uint8_t *)st.put(qdesc, (size_t)nq * 32);
    if ((rc = stage_commit(m, &st))) return rc;
    void *p;
    if ((rc = scratch(m, S_ORD, n * sizeof(uint32_t), &p))) return rc;
    uint32_t *d_ord = (uint32_t *)p;
    if ((rc = scratch(m, S_OUT, (size_t)nq * 2 * sizeof(int), &p))) return rc;
    int *d_out = (int *)p;
    uint8_t *h_out;
    if ((rc = out_buffer(m, (size_t)nq * 2 * sizeof(int), &h_out))) return rc;
    SigmaTab sig;
    memset(&sig, 0, sizeof(sig));
    if (inv_level_sigma2) for (int l = 0; l < std::min(kf->n_levels, ORBHIP_MAX_LEVELS); ++l) sig.inv_sigma2[l] = inv_level_sigma2[l];
    const Batch one = {nullptr, nullptr, 0, 0};
    hipLaunchKernelGGL(k_grid_order, dim3((kf->n + 255) / 256), dim3(256), 0, m->stream, D, d_ord, one);
    hipLaunchKernelGGL(k_best_in_window, dim3((nq + 3) / 4), dim3(256), 0, m->stream, D, d_ord, d_q, d_qdesc, nq, chi2_gate,
                       sig, d_out, d_out + nq);
    ORBHIP_HIP_CHECK(hipGetLastError());
    ORBHIP_HIP_CHECK(hipMemcpyAsync(h_out, d_out, (size_t)nq * 2 * sizeof(int), hipMemcpyDeviceToHost, m->stream));
    ORBHIP_HIP_CHECK(hipStreamSynchronize(m->stream));
    memcpy(best_idx, h_out, (size_t)nq * sizeof(int));
    memcpy(best_dist, h_out + (size_t)nq * sizeof(int), (size_t)nq * sizeof(int));
    return ORBHIP_OK;
}

int orbhip_search_by_projection_points(orbhip_matcher *m, const orbhip_frame_view *f, const orbhip_query *q,
                                       const uint8_t *qdesc, int nq, const uint8_t *taken, int32_t *assign,
                                       float nnratio, int *nmatches)
{
    if (!m || !f || (nq > 0 && (!q || !qdesc)) || !assign || !nmatches || nq < 0) return ORBHIP_E_ARG;
    return run_search(m, 1, f, q, qdesc, nullptr, nq, taken, nnratio, 0, assign, f->n, nmatches);
}

static int search_device(orbhip_matcher *m, int mode, int pairs, const void *d_kps, const void *d_desc, const void *d_n,
                         int cap, const void *d_u_right, const void *d_taken, float min_x, float min_y, float grid_inv_w,
                         float grid_inv_h, const void *d_q, const void *d_qdesc, const void *d_nq, int qcap, float nnratio,
                         int check_ori, void *d_assign, void *d_nmatches, int t0 = 0, int ts = 1, int qd0 = 0, int qds = 1,
                         const ProjLaunch *proj = nullptr)
{
    if (!m || pairs <= 0 || !d_kps || !d_desc || !d_n || !d_q || !d_qdesc || !d_nq || !d_assign || !d_nmatches || cap <= 0 || qcap <= 0)
        return ORBHIP_E_ARG;
    ORBHIP_HIP_CHECK(hipSetDevice(m->device));
    int rc;
    void *p;
    const int stride = (cap + 1) & ~1;
    if ((rc = scratch(m, S_CAND, (size_t)pairs * qcap * stride * sizeof(unsigned long long), &p))) return rc;
    unsigned long long *d_cand = (unsigned long long *)p;
    if ((rc = scratch(m, S_CNT, (size_t)pairs * qcap * sizeof(int), &p))) return rc;
    int *d_cnt = (int *)p;
    if ((rc = ensure_resolve_attr(m))) return rc;
    DevFrame D;
    D.n = cap; D.keys = (const orbhip_keypoint *)d_kps; D.desc = (const uint8_t *)d_desc; D.u_right = (const float *)d_u_right;
    D.min_x = min_x; D.min_y = min_y; D.inv_w = grid_inv_w; D.inv_h = grid_inv_h;
    const Batch B = {(const int *)d_n, (const int *)d_nq, cap, qcap, t0, ts, qd0, qds};
    unsigned long long *d_ccand;
    if ((rc = launch_window_search(m, pairs, D, cap, (const orbhip_query *)d_q, (const uint8_t *)d_qdesc, qcap, d_cand, d_cnt, stride, 1,
                                   B, &d_ccand, (const uint8_t *)d_taken, mode == 0 ? TH_HIGH : 256, proj)))
        return rc;
    if ((rc = launch_resolve_par(m, pairs, mode, D, (const orbhip_query *)d_q, qcap, cap, d_cand, d_ccand, d_cnt, stride,
                                 (const uint8_t *)d_taken, nnratio, check_ori, (int *)d_assign, (int *)d_nmatches, B, TH_HIGH, 0)))
        return rc;
    ORBHIP_HIP_CHECK(hipGetLastError());
    return ORBHIP_OK;
}

int orbhip_search_by_projection_frame_device(orbhip_matcher *m, int pairs, const void *d_kps, const void *d_desc,
                                             const void *d_n, int cap, const void *d_u_right, const void *d_taken,
                                             float min_x, float min_y, float grid_inv_w, float grid_inv_h,
                                             const void *d_q, const void *d_qdesc, const void *d_nq, int qcap,
                                             int check_ori, void *d_assign, void *d_nmatches)
{
    return search_device(m, 0, pairs, d_kps, d_desc, d_n, cap, d_u_right, d_taken, min_x, min_y, grid_inv_w, grid_inv_h, d_q,
                         d_qdesc, d_nq, qcap, 0.f, check_ori, d_assign, d_nmatches);
}

int orbhip_search_by_projection_points_device(orbhip_matcher *m, int pairs, const void *d_kps, const void *d_desc,
                                              const void *d_n, int cap, const void *d_u_right, const void *d_taken,
                                              float min_x, float min_y, float grid_inv_w, float grid_inv_h,
                                              const void *d_q, const void *d_qdesc, const void *d_nq, int qcap,
                                              float nnratio, void *d_assign, void *d_nmatches)
{
    return search_device(m, 1, pairs, d_kps, d_desc, d_n, cap, d_u_right, d_taken, min_x, min_y, grid_inv_w, grid_inv_h, d_q,
                         d_qdesc, d_nq, qcap, nnratio, 0, d_assign, d_nmatches);
}

int orbhip_compute_stereo_matches_device(orbhip_matcher *m, orbhip_extractor *left, int l0, int ls,
                                         orbhip_extractor *right, int r0, int rs, int pairs, const void *d_kps_l,
                                         const void *d_desc_l, const void *d_n_l, const void *d_kps_r,
                                         const void *d_desc_r, const void *d_n_r, int cap, float mbf, float mb,
                                         void *d_u_right, void *d_depth, void *d_nmatches)
{
    if (!m || !left || !right || pairs <= 0 || cap <= 0 || !d_kps_l || !d_desc_l || !d_n_l || !d_kps_r || !d_desc_r ||
        !d_n_r || !d_u_right || !d_depth || !d_nmatches) return ORBHIP_E_ARG;
    if (!left->bound || !right->bound || left->device != m->device || right->device != m->device ||
        left->nlevels != right->nlevels || l0 < 0 || r0 < 0 || ls < 0 || rs < 0 ||
        l0 + (pairs - 1) * ls >= left->last_batch || r0 + (pairs - 1) * rs >= right->last_batch) {
        set_error("stereo: extractor handles do not hold the requested frames on device %d", m->device);
        return ORBHIP_E_ARG;
    }
    if (cap > (1 << 20)) { set_error("stereo: capacity %d exceeds the 2^20 keypoints the match key holds", cap); return ORBHIP_E_CAPACITY; }
    ORBHIP_HIP_CHECK(hipSetDevice(m->device));
    // mvImagePyramid[0] of handles that produce it on demand; this launch is ordered behind the copy
    if (int rc = ensure_level0(left, m->stream)) return rc;
    if (right != left) { if (int rc = ensure_level0(right, m->stream)) return rc; }
    StereoGeom G;
    memset(&G, 0, sizeof(G));
    G.nlevels = left->nlevels; G.nrows = left->G.lv[0].h; G.mbf = mbf; G.mb = mb;
    for (int l = 0; l < G.nlevels; ++l) {
        const LevelGeom &A = left->G.lv[l], &Bv = right->G.lv[l];
        G.left[l] = left->d_pyr + A.plane_off + (size_t)kEdge * A.pitch + kPadL;
        G.right[l] = right->d_pyr + Bv.plane_off + (size_t)kEdge * Bv.pitch + kPadL;
        G.pitch_l[l] = A.pitch; G.pitch_r[l] = Bv.pitch; G.cols_r[l] = Bv.w;
        G.sf[l] = left->sf[l]; G.isf[l] = left->isf[l];
    }
    for (int l = G.nlevels; l < ORBHIP_MAX_LEVELS; ++l) { G.left[l] = G.left[0]; G.right[l] = G.right[0]; }
    void *p;
    int rc;
    if ((rc = scratch(m, S_ORD, al256((size_t)pairs * ((size_t)G.nrows + 1) * sizeof(int)) + (size_t)pairs * cap * sizeof(int4), &p))) return rc;
    int *d_rowstart = (int *)p;
    int4 *d_order = reinterpret_cast<int4 *>((uint8_t *)p + al256((size_t)pairs * ((size_t)G.nrows + 1) * sizeof(int)));
    if ((rc = scratch(m, S_CNT, (size_t)pairs * cap * sizeof(int), &p))) return rc;
    int *d_sad = (int *)p;
    const StereoBatch B = {(const int *)d_n_l, (const int *)d_n_r, l0, ls, r0, rs, cap, left->G.frame_bytes, right->G.frame_bytes};
    const int R = stereo_row_reach(G);
    if (G.nrows <= kStereoRowsMax)
        hipLaunchKernelGGL(k_stereo_sort, dim3(pairs), dim3(256), 0, m->stream, (const orbhip_keypoint *)d_kps_r, cap, G.nrows, d_rowstart,
                           d_order, B, stereo_scales(G));
    else d_rowstart = nullptr;   // taller images: every right keypoint is a candidate
    hipLaunchKernelGGL(k_stereo_match, dim3((cap + 3) / 4, pairs), dim3(256), 0, m->stream, (const orbhip_keypoint *)d_kps_l,
                       (const uint8_t *)d_desc_l, cap, (const orbhip_keypoint *)d_kps_r, (const uint8_t *)d_desc_r, cap,
                       (const int *)d_rowstart, (const int4 *)d_order, R, G, (float *)d_u_right, (float *)d_depth, d_sad, B);
    hipLaunchKernelGGL(k_stereo_cull, dim3(pairs), dim3(256), 0, m->stream, cap, d_sad, (float *)d_u_right, (float *)d_depth,
                       (int *)d_nmatches, B);
    ORBHIP_HIP_CHECK(hipGetLastError());
    return ORBHIP_OK;
}

int orbhip_search_for_initialization_device(orbhip_matcher *m, int pairs, const void *d_kps, const void *d_desc,
                                            const void *d_n, int cap, int f1_first, int f1_step, int f2_first,
                                            int f2_step, float min_x, float min_y, float grid_inv_w, float grid_inv_h,
                                            int reset_prev, void *d_prev_matched, int window_size, float nnratio,
                                            int check_ori, void *d_matches12, void *d_nmatches)
{
    if (!m || pairs <= 0 || cap <= 0 || !d_kps || !d_desc || !d_n || !d_prev_matched || !d_matches12 || !d_nmatches)
        return ORBHIP_E_ARG;
    if (cap > kResolveMax) {
        set_error("matcher: cap %d exceeds the LDS-resident limit %d", cap, kResolveMax);
        return ORBHIP_E_CAPACITY;
    }
    ORBHIP_HIP_CHECK(hipSetDevice(m->device));
    int rc;
    void *p;
    const int stride = (cap + 1) & ~1;
    if ((rc = scratch(m, S_CAND, (size_t)pairs * cap * stride * sizeof(unsigned long long), &p))) return rc;
    unsigned long long *d_cand = (unsigned long long *)p;
    if ((rc = scratch(m, S_CNT, (size_t)pairs * cap * sizeof(int), &p))) return rc;
    int *d_cnt = (int *)p;
    if ((rc = scratch(m, S_Q, (size_t)pairs * cap * sizeof(orbhip_query), &p))) return rc;
    orbhip_query *d_q = (orbhip_query *)p;
    if ((rc = scratch(m, S_TAKEN, (size_t)pairs * sizeof(int), &p))) return rc;
    int *d_nq = (int *)p;
    if ((rc = ensure_resolve_attr(m))) return rc;
    const orbhip_keypoint *keys = (const orbhip_keypoint *)d_kps;
    hipLaunchKernelGGL(k_init_queries, dim3((cap + 255) / 256, pairs), dim3(256), 0, m->stream, keys, (const int *)d_n, cap,
                       f1_first, f1_step, (float *)d_prev_matched, reset_prev, (float)window_size, d_q, d_nq);
    DevFrame D;
    D.n = cap; D.keys = keys; D.desc = (const uint8_t *)d_desc; D.u_right = nullptr;
    D.min_x = min_x; D.min_y = min_y; D.inv_w = grid_inv_w; D.inv_h = grid_inv_h;
    const Batch B = {(const int *)d_n, d_nq, cap, cap, f2_first, f2_step, f1_first, f1_step};
    unsigned long long *d_ccand;
    if ((rc = launch_window_search(m, pairs, D, cap, d_q, (const uint8_t *)d_desc, cap, d_cand, d_cnt, stride, 0, B, &d_ccand, nullptr, 256))) return rc;
    launch_resolve_init(m, pairs, D, keys, d_q, cap, cap, d_cand, d_ccand, d_cnt, stride, nnratio, check_ori, (int *)d_matches12,
                        (int *)d_nmatches, B);
    hipLaunchKernelGGL(k_init_update_prev, dim3((cap + 255) / 256, pairs), dim3(256), 0, m->stream, keys, cap, f2_first, f2_step,
                       d_nq, (const int *)d_matches12, (float *)d_prev_matched);
    ORBHIP_HIP_CHECK(hipGetLastError());
    return ORBHIP_OK;
}

int orbhip_project_last_frame_device(orbhip_matcher *m, int pairs, const orbhip_camera *cam, const void *d_Tcw,
                                     const void *d_Tlw, const void *d_kps, const void *d_n, int cap, int last_first,
                                     int last_step, const void *d_world, const void *d_flags, float th, int mono,
                                     void *d_q, void *d_nq)
{
    if (!m || !cam || pairs <= 0 || cap <= 0 || !d_Tcw || !d_Tlw || !d_kps || !d_n || !d_world || !d_flags || !d_q ||
        cam->n_levels < 1 || cam->n_levels > ORBHIP_MAX_LEVELS)
        return ORBHIP_E_ARG;
    ORBHIP_HIP_CHECK(hipSetDevice(m->device));
    ProjBatch B = {(const float *)d_Tcw, (const float *)d_Tlw, (const orbhip_keypoint *)d_kps, (const int *)d_n,
                   (const float *)d_world, (const uint8_t *)d_flags, (orbhip_query *)d_q, (int *)d_nq, 0, cap, last_first, last_step};
    hipLaunchKernelGGL(k_project_last_frame, dim3((cap + 255) / 256, pairs), dim3(256), 0, m->stream, B, *cam, th, mono);
    ORBHIP_HIP_CHECK(hipGetLastError());
    return ORBHIP_OK;
}

int orbhip_track_last_frame_device(orbhip_matcher *m, int pairs, const orbhip_camera *cam, const void *d_Tcw,
                                   const void *d_Tlw, const void *d_kps, const void *d_desc, const void *d_n, int cap,
                                   int cur_first, int cur_step, int last_first, int last_step, const void *d_world,
                                   const void *d_flags, const void *d_u_right, const void *d_taken, float th, int mono,
                                   int check_ori, void *d_assign, void *d_nmatches)
{
    if (!m || !cam || pairs <= 0 || cap <= 0 || !d_desc) return ORBHIP_E_ARG;
    if (!d_Tcw || !d_Tlw || !d_kps || !d_n || !d_world || !d_flags || cam->n_levels < 1 || cam->n_levels > ORBHIP_MAX_LEVELS)
        return ORBHIP_E_ARG;
    ORBHIP_HIP_CHECK(hipSetDevice(m->device));   // before scratch(): its buffers must land on the matcher's device
    int rc;
    void *p;
    if ((rc = scratch(m, S_Q, (size_t)pairs * cap * sizeof(orbhip_query), &p))) return rc;
    orbhip_query *d_q = (orbhip_query *)p;
    if ((rc = scratch(m, S_TAKEN, (size_t)pairs * sizeof(int), &p))) return rc;   // per-pair query counts
    int *d_nq = (int *)p;
    ProjLaunch proj;
    proj.P = {(const float *)d_Tcw, (const float *)d_Tlw, (const orbhip_keypoint *)d_kps, (const int *)d_n,
              (const float *)d_world, (const uint8_t *)d_flags, d_q, d_nq, 0, cap, last_first, last_step};
    proj.cam = *cam; proj.th = th; proj.mono = mono;
    // Frame::ComputeImageBounds / mfGridElement{Width,Height}Inv (src/Frame.cc:99-100) from the camera's bounds
    const float inv_w = (float)GRID_COLS / (cam->max_x - cam->min_x), inv_h = (float)GRID_ROWS / (cam->max_y - cam->min_y);
    return search_device(m, 0, pairs, d_kps, d_desc, d_n, cap, d_u_right, d_taken, cam->min_x, cam->min_y, inv_w, inv_h, d_q,
                         d_desc, d_nq, cap, 0.f, check_ori, d_assign, d_nmatches, cur_first, cur_step, last_first, last_step, &proj);
}

int orbhip_frustum_queries_device(orbhip_matcher *m, int frames, const orbhip_camera *cam, const void *d_Tcw, int pcap,
                                  const void *d_np, const void *d_world, const void *d_normal, const void *d_max_dist,
                                  const void *d_min_dist, const void *d_flags, float viewing_cos_limit, float th,
                                  void *d_q, void *d_view_cos)
{
    if (!m || !cam || frames <= 0 || pcap <= 0 || !d_Tcw || !d_np || !d_world || !d_normal || !d_max_dist || !d_min_dist ||
        !d_flags || !d_q || cam->n_levels < 1 || cam->n_levels > ORBHIP_MAX_LEVELS)
        return ORBHIP_E_ARG;
    ORBHIP_HIP_CHECK(hipSetDevice(m->device));
    FrustumBatch B = {(const float *)d_Tcw, (const int *)d_np, (const float *)d_world, (const float *)d_normal,
                      (const float *)d_max_dist, (const float *)d_min_dist, (const uint8_t *)d_flags, (orbhip_query *)d_q,
                      (float *)d_view_cos, 0, pcap};
    hipLaunchKernelGGL(k_frustum_queries, dim3((pcap + 255) / 256, frames), dim3(256), 0, m->stream, B, *cam, viewing_cos_limit, th);
    ORBHIP_HIP_CHECK(hipGetLastError());
    return ORBHIP_OK;
}

int orbhip_project_last_frame(orbhip_matcher *m, const orbhip_camera *cam, const float *Tcw, const float *Tlw, int n,
                              const float *world, const uint8_t *flags, const orbhip_keypoint *last_keys, float th,
                              int mono, orbhip_query *q)
{
    if (!m || !cam || n < 0 || !Tcw || !Tlw || !q || cam->n_levels < 1 || cam->n_levels > ORBHIP_MAX_LEVELS) return ORBHIP_E_ARG;
    if (n == 0) return ORBHIP_OK;
    if (!world || !flags || !last_keys) return ORBHIP_E_ARG;
    ORBHIP_HIP_CHECK(hipSetDevice(m->device));
    Stage st;
    int rc = stage_begin(m, 2 * al256(48) + al256((size_t)n * 12) + al256((size_t)n) + al256((size_t)n * sizeof(orbhip_keypoint)), &st);
    if (rc) return rc;
    const float *dT = (const float *)st.put(Tcw, 48), *dL = (const float *)st.put(Tlw, 48);
    const float *dW = (const float *)st.put(world, (size_t)n * 12);
    const uint8_t *dF = (const uint8_t *)st.put(flags, (size_t)n);
    const orbhip_keypoint *dK = (const orbhip_keypoint *)st.put(last_keys, (size_t)n * sizeof(orbhip_keypoint));
    if ((rc = stage_commit(m, &st))) return rc;
    void *p;
    if ((rc = scratch(m, S_Q, (size_t)n * sizeof(orbhip_query), &p))) return rc;
    ProjBatch B = {dT, dL, dK, nullptr, dW, dF, (orbhip_query *)p, nullptr, n, n, 0, 0};
    hipLaunchKernelGGL(k_project_last_frame, dim3((n + 255) / 256, 1), dim3(256), 0, m->stream, B, *cam, th, mono);
    ORBHIP_HIP_CHECK(hipGetLastError());
    ORBHIP_HIP_CHECK(hipMemcpyAsync(q, p, (size_t)n * sizeof(orbhip_query), hipMemcpyDeviceToHost, m->stream));
    ORBHIP_HIP_CHECK(hipStreamSynchronize(m->stream));
    return ORBHIP_OK;
}

int orbhip_frustum_queries(orbhip_matcher *m, const orbhip_camera *cam, const float *Tcw, int n, const float *world,
                           const float *normal, const float *max_dist, const float *min_dist, const uint8_t *flags,
                           float viewing_cos_limit, float th, orbhip_query *q, float *view_cos)
{
    if (!m || !cam || n < 0 || !Tcw || !q || cam->n_levels < 1 || cam->n_levels > ORBHIP_MAX_LEVELS) return ORBHIP_E_ARG;
    if (n == 0) return ORBHIP_OK;
    if (!world || !normal || !max_dist || !min_dist || !flags) return ORBHIP_E_ARG;
    ORBHIP_HIP_CHECK(hipSetDevice(m->device));
    Stage st;
    int rc = stage_begin(m, al256(48) + 2 * al256((size_t)n * 12) + 2 * al256((size_t)n * 4) + al256((size_t)n), &st);
    if (rc) return rc;
    const float *dT = (const float *)st.put(Tcw, 48);
    const float *dW = (const float *)st.put(world, (size_t)n * 12), *dN = (const float *)st.put(normal, (size_t)n * 12);
    const float *dMx = (const float *)st.put(max_dist, (size_t)n * 4), *dMn = (const float *)st.put(min_dist, (size_t)n * 4);
    const uint8_t *dF = (const uint8_t *)st.put(flags, (size_t)n);
    if ((rc = stage_commit(m, &st))) return rc;
    void *p, *pv;
    if ((rc = scratch(m, S_Q, (size_t)n * sizeof(orbhip_query), &p))) return rc;
    if ((rc = scratch(m, S_OUT, (size_t)n * sizeof(float), &pv))) return rc;
    FrustumBatch B = {dT, nullptr, dW, dN, dMx, dMn, dF, (orbhip_query *)p, (float *)pv, n, n};
    hipLaunchKernelGGL(k_frustum_queries, dim3((n + 255) / 256, 1), dim3(256), 0, m->stream, B, *cam, viewing_cos_limit, th);
    ORBHIP_HIP_CHECK(hipGetLastError());
    ORBHIP_HIP_CHECK(hipMemcpyAsync(q, p, (size_t)n * sizeof(orbhip_query), hipMemcpyDeviceToHost, m->stream));
    if (view_cos) ORBHIP_HIP_CHECK(hipMemcpyAsync(view_cos, pv, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, m->stream));
    ORBHIP_HIP_CHECK(hipStreamSynchronize(m->stream));
    return ORBHIP_OK;
}

#ifdef ORBHIP_DEVTOOLS
// development builds: {workgroups, sum of rounds, max rounds} of the parallel resolve since the last call
int orbhip_dev_resolve_stats(unsigned int out[4])
{
    unsigned int z[4] = {0, 0, 0, 0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(orbhip::g_resolve_stats), sizeof(z)) != hipSuccess) return ORBHIP_E_HIP;
    if (hipMemcpyToSymbol(HIP_SYMBOL(orbhip::g_resolve_stats), z, sizeof(z)) != hipSuccess) return ORBHIP_E_HIP;
    return ORBHIP_OK;
}
#endif

int orbhip_keyframe_queries(orbhip_matcher *m, const orbhip_camera *cam, int mode, int double_invz, const float *T1,
                            const float *T2, int n, const float *world, const float *normal, const float *max_dist,
                            const float *min_dist, const uint8_t *flags, float th, orbhip_query *q)
{
    if (!m || !cam || n < 0 || !T1 || !q || (mode != 0 && mode != 1) || (mode == 1 && !T2) || cam->n_levels < 1 ||
        cam->n_levels > ORBHIP_MAX_LEVELS)
        return ORBHIP_E_ARG;
    if (n == 0) return ORBHIP_OK;
    if (!world || !max_dist || !min_dist || !flags || (mode == 0 && !normal)) return ORBHIP_E_ARG;
    ORBHIP_HIP_CHECK(hipSetDevice(m->device));
    Stage st;
    int rc = stage_begin(m, 2 * al256(48) + 2 * al256((size_t)n * 12) + 2 * al256((size_t)n * 4) + al256((size_t)n), &st);
    if (rc) return rc;
    KfQueryArgs A;
    A.T1 = (const float *)st.put(T1, 48);
    A.T2 = T2 ? (const float *)st.put(T2, 48) : nullptr;
    A.world = (const float *)st.put(world, (size_t)n * 12);
    A.normal = normal ? (const float *)st.put(normal, (size_t)n * 12) : nullptr;
    A.max_dist = (const float *)st.put(max_dist, (size_t)n * 4);
    A.min_dist = (const float *)st.put(min_dist, (size_t)n * 4);
    A.flags = (const uint8_t *)st.put(flags, (size_t)n);
    A.n = n; A.mode = mode; A.double_invz = double_invz;
    if ((rc = stage_commit(m, &st))) return rc;
    void *p;
    if ((rc = scratch(m, S_Q, (size_t)n * sizeof(orbhip_query), &p))) return rc;
    A.q = (orbhip_query *)p;
    hipLaunchKernelGGL(k_keyframe_queries, dim3((n + 255) / 256), dim3(256), 0, m->stream, A, *cam, th);
    ORBHIP_HIP_CHECK(hipGetLastError());
    ORBHIP_HIP_CHECK(hipMemcpyAsync(q, p, (size_t)n * sizeof(orbhip_query), hipMemcpyDeviceToHost, m->stream));
    ORBHIP_HIP_CHECK(hipStreamSynchronize(m->stream));
    return ORBHIP_OK;
}

int orbhip_fuse(orbhip_matcher *m, const orbhip_frame_view *kf, const orbhip_camera *cam, const float *Tcw, int sim3_form,
                int n, const float *world, const float *normal, const float *max_dist, const float *min_dist,
                const uint8_t *flags, const uint8_t *point_desc, float th, const float *inv_level_sigma2, int32_t *best_idx,
                int32_t *best_dist)
{
    if (!m || !kf || n < 0 || (n > 0 && (!best_idx || !best_dist || !point_desc))) return ORBHIP_E_ARG;
    if (n == 0) return ORBHIP_OK;
    static thread_local std::vector<orbhip_query> q;
    q.resize((size_t)n);
    int rc = orbhip_keyframe_queries(m, cam, 0, sim3_form ? 1 : 0, Tcw, nullptr, n, world, normal, max_dist, min_dist, flags, th, q.data());
    if (rc) return rc;
    return orbhip_search_best_in_window(m, kf, q.data(), point_desc, n, 1, inv_level_sigma2, best_idx, best_dist);
}

int orbhip_search_by_sim3(orbhip_matcher *m, const orbhip_frame_view *kf1, const orbhip_frame_view *kf2,
                          const orbhip_camera *cam, const float *T1w, const float *T2w, const float *S21, const float *S12,
                          const float *world1, const float *max_dist1, const float *min_dist1, const uint8_t *flags1,
                          const uint8_t *desc1, const float *world2, const float *max_dist2, const float *min_dist2,
                          const uint8_t *flags2, const uint8_t *desc2, float th, int32_t *matches12, int *nfound)
{
    if (!m || !kf1 || !kf2 || !cam || !T1w || !T2w || !S21 || !S12 || !matches12 || !nfound) return ORBHIP_E_ARG;
    const int n1 = kf1->n, n2 = kf2->n;
    for (int i = 0; i < n1; ++i) matches12[i] = -1;
    *nfound = 0;
    if (n1 == 0 || n2 == 0) return ORBHIP_OK;
    static thread_local std::vector<orbhip_query> q1, q2;
    static thread_local std::vector<int32_t> m1, d1, m2, d2;
    q1.resize(n1); q2.resize(n2); m1.resize(n1); d1.resize(n1); m2.resize(n2); d2.resize(n2);
    int rc;
    // KF1's map points into KF2 (:1146-1226) and KF2's into KF1 (:1228-1303)
    if ((rc = orbhip_keyframe_queries(m, cam, 1, 1, T1w, S21, n1, world1, nullptr, max_dist1, min_dist1, flags1, th, q1.data()))) return rc;
    if ((rc = orbhip_keyframe_queries(m, cam, 1, 1, T2w, S12, n2, world2, nullptr, max_dist2, min_dist2, flags2, th, q2.data()))) return rc;
    if ((rc = orbhip_search_best_in_window(m, kf2, q1.data(), desc1, n1, 0, nullptr, m1.data(), d1.data()))) return rc;
    if ((rc = orbhip_search_best_in_window(m, kf1, q2.data(), desc2, n2, 0, nullptr, m2.data(), d2.data()))) return rc;
    int found = 0;
    for (int i1 = 0; i1 < n1; ++i1) {   // agreement, :1306-1323
        const int idx2 = (m1[i1] >= 0 && d1[i1] <= TH_HIGH) ? m1[i1] : -1;
        if (idx2 < 0) continue;
        const int idx1 = (m2[idx2] >= 0 && d2[idx2] <= TH_HIGH) ? m2[idx2] : -1;
        if (idx1 == i1) { matches12[i1] = idx2; ++found; }
    }
    *nfound = found;
    return ORBHIP_OK;
}

int orbhip_matcher_set_stream(orbhip_matcher *m, void *stream)
{
    if (!m) return ORBHIP_E_ARG;
    ORBHIP_HIP_CHECK(hipSetDevice(m->device));
    ORBHIP_HIP_CHECK(hipStreamSynchronize(m->stream));
    m->stream = stream ? (hipStream_t)stream : m->own_stream;
    return ORBHIP_OK;
}

int orbhip_matcher_sync(orbhip_matcher *m)
{
    if (!m) return ORBHIP_E_ARG;
    ORBHIP_HIP_CHECK(hipSetDevice(m->device));
    ORBHIP_HIP_CHECK(hipStreamSynchronize(m->stream));
    return ORBHIP_OK;
}

int orbhip_compute_stereo_matches(orbhip_matcher *m, orbhip_extractor *left, int frame_l, orbhip_extractor *right,
                                  int frame_r, const orbhip_keypoint *keys_l, const uint8_t *desc_l, int nl,
                                  const orbhip_keypoint *keys_r, const uint8_t *desc_r, int nr, float mbf, float mb,
                                  float *u_right, float *depth, int *nmatches)
{
    if (!m || !left || !right || !u_right || !depth || !nmatches || nl < 0 || nr < 0) return ORBHIP_E_ARG;
    if (!left->bound || !right->bound || frame_l < 0 || frame_l >= left->last_batch || frame_r < 0 ||
        frame_r >= right->last_batch || left->device != m->device || right->device != m->device ||
        left->nlevels != right->nlevels) {
        set_error("stereo: extractor handles do not hold matching pyramids on device %d", m->device);
        return ORBHIP_E_ARG;
    }
    for (int i = 0; i < nl; ++i) { u_right[i] = -1.0f; depth[i] = -1.0f; }
    *nmatches = 0;
    if (nl == 0 || nr == 0) return ORBHIP_OK;
    if (nr > (1 << 20)) { set_error("stereo: %d right keypoints exceed the 2^20 the match key holds", nr); return ORBHIP_E_CAPACITY; }
    ORBHIP_HIP_CHECK(hipSetDevice(m->device));
    if (int rc = ensure_level0(left, nullptr)) return rc;     // mvImagePyramid[0] of handles that produce it on demand
    if (int rc = ensure_level0(right, nullptr)) return rc;
    // the pyramids were produced on the extractors' streams
    ORBHIP_HIP_CHECK(hipStreamSynchronize(left->stream));
    ORBHIP_HIP_CHECK(hipStreamSynchronize(right->stream));
    StereoGeom G;
    memset(&G, 0, sizeof(G));
    G.nlevels = left->nlevels; G.nrows = left->G.lv[0].h; G.mbf = mbf; G.mb = mb;
    for (int l = 0; l < G.nlevels; ++l) {
        const LevelGeom &A = left->G.lv[l], &B = right->G.lv[l];
        G.left[l] = left->d_pyr + (size_t)frame_l * left->G.frame_bytes + A.plane_off + (size_t)kEdge * A.pitch + kPadL;
        G.right[l] = right->d_pyr + (size_t)frame_r * right->G.frame_bytes + B.plane_off + (size_t)kEdge * B.pitch + kPadL;
        G.pitch_l[l] = A.pitch; G.pitch_r[l] = B.pitch; G.cols_r[l] = B.w;
        G.sf[l] = left->sf[l]; G.isf[l] = left->isf[l];
    }
    void *p;
    int rc;
    Stage st;
    if ((rc = stage_begin(m, al256((size_t)nl * sizeof(orbhip_keypoint)) + al256((size_t)nl * 32) +
                                 al256((size_t)nr * sizeof(orbhip_keypoint)) + al256((size_t)nr * 32), &st))) return rc;
    const orbhip_keypoint *d_kl = (const orbhip_keypoint *)st.put(keys_l, (size_t)nl * sizeof(orbhip_keypoint));
    const uint8_t *d_dl = (const uint8_t *)st.put(desc_l, (size_t)nl * 32);
    const orbhip_keypoint *d_kr = (const orbhip_keypoint *)st.put(keys_r, (size_t)nr * sizeof(orbhip_keypoint));
    const uint8_t *d_dr = (const uint8_t *)st.put(desc_r, (size_t)nr * 32);
    if ((rc = stage_commit(m, &st))) return rc;
    if ((rc = scratch(m, S_ORD, al256(((size_t)G.nrows + 1) * sizeof(int)) + (size_t)nr * sizeof(int4), &p))) return rc;
    int *d_rowstart = (int *)p;
    int4 *d_order = reinterpret_cast<int4 *>((uint8_t *)p + al256(((size_t)G.nrows + 1) * sizeof(int)));
    // outputs contiguous: u_right[nl] | depth[nl] | sad[nl] | n
    if ((rc = scratch(m, S_OUT, (size_t)(3 * nl + 1) * sizeof(float), &p))) return rc;
    float *d_ur = (float *)p, *d_depth = d_ur + nl;
    int *d_sad = (int *)(d_depth + nl), *d_n = d_sad + nl;
    uint8_t *h_out;
    if ((rc = out_buffer(m, (size_t)(3 * nl + 1) * sizeof(float), &h_out))) return rc;
    const StereoBatch one = {nullptr, nullptr, 0, 0, 0, 0, 0, 0, 0};
    if (G.nrows <= kStereoRowsMax)
        hipLaunchKernelGGL(k_stereo_sort, dim3(1), dim3(256), 0, m->stream, d_kr, nr, G.nrows, d_rowstart, d_order, one, stereo_scales(G));
    else d_rowstart = nullptr;
    hipLaunchKernelGGL(k_stereo_match, dim3((nl + 3) / 4), dim3(256), 0, m->stream, d_kl, d_dl, nl, d_kr, d_dr, nr,
                       (const int *)d_rowstart, (const int4 *)d_order, stereo_row_reach(G), G, d_ur, d_depth, d_sad, one);
    hipLaunchKernelGGL(k_stereo_cull, dim3(1), dim3(256), 0, m->stream, nl, d_sad, d_ur, d_depth, d_n, one);
    ORBHIP_HIP_CHECK(hipGetLastError());
    ORBHIP_HIP_CHECK(hipMemcpyAsync(h_out, d_ur, (size_t)(3 * nl + 1) * sizeof(float), hipMemcpyDeviceToHost, m->stream));
    ORBHIP_HIP_CHECK(hipStreamSynchronize(m->stream));
    memcpy(u_right, h_out, (size_t)nl * sizeof(float));
    memcpy(depth, h_out + (size_t)nl * sizeof(float), (size_t)nl * sizeof(float));
    *nmatches = reinterpret_cast<const int *>(h_out)[3 * nl];
    return ORBHIP_OK;
}

}  // extern "C"
