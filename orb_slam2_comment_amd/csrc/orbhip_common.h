// Shared host/device helpers for the orbhip C-ABI library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "orbhip.h"

namespace orbhip {

constexpr int kEdge = 19;        // EDGE_THRESHOLD, src/ORBextractor.cc:74
constexpr int kPadL = 32;        // left pad of a pyramid row (>= kEdge); keeps the ROI 32-B aligned
constexpr int kHalfPatch = 15;   // HALF_PATCH_SIZE
constexpr int kPatchSize = 31;   // PATCH_SIZE
constexpr int kWave = 64;        // CDNA wavefront

void set_error(const char *fmt, ...);

#define ORBHIP_HIP_CHECK(expr)                                                          \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess) {                                                         \
            ::orbhip::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),  \
                                __FILE__, __LINE__);                                    \
            return ORBHIP_E_HIP;                                                        \
        }                                                                               \
    } while (0)

// Pyramid geometry of one level for one image size; passed to kernels by value.
struct LevelGeom {
    int w, h;              // level size (ROI)
    int pitch, prows;      // padded plane: row pitch in bytes, rows = h + 2*kEdge
    unsigned plane_off;    // byte offset of the padded plane inside a frame's pyramid block
    int maxBX, maxBY;      // maxBorderX/Y = w-16 / h-16 (minBorder = 16), ORBextractor.cc:773-776
    int cell_base, ncells; // cells of this level inside the cell table
    int nIni;              // initial octree nodes, :543
    float hX;              // :545
    int quota;             // mnFeaturesPerLevel[level]
    int kp_base, kp_cap;   // slice of the per-frame selected-keypoint array
    int cand_base, cand_cap; // slice of the per-frame octree key workspace
    int patch;             // scaledPatchSize, :837
    float scale;           // mvScaleFactor[level]
};

struct PyrGeom {
    int nlevels;
    int rows, cols;
    int ncells_total, slot_cap;
    int kp_cap_total, cand_cap_total;
    unsigned frame_bytes;  // bytes of one frame's pyramid block
    int ini_th, min_th;
    LevelGeom lv[ORBHIP_MAX_LEVELS];
};

// One FAST cell (one cv::FAST call of the reference, ORBextractor.cc:789-829).
struct CellDesc {
    short level;
    short x0, y0, x1, y1;  // sub-image [x0,x1) x [y0,y1) in level ROI coordinates
    short offx, offy;      // x0 - minBorderX, y0 - minBorderY (= j*wCell, i*hCell)
    short pad;
};

// Wave-level scan / reductions on the DPP data path (no LDS crossbar round trips).  Rows are 16 lanes: Hillis-Steele
// inside a row with row_shr (lanes shifted in from outside the row contribute 0), then the row totals travel with
// row_bcast:15 (into rows 1 and 3) and row_bcast:31 (into rows 2 and 3).
__device__ __forceinline__ int wave_incl_scan_add(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31
    return v;
}
// sum over the wavefront, wave-uniform result (butterfly inside the rows, then the row totals; lane 63 holds the sum)
__device__ __forceinline__ int wave_sum(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);    // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);    // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false);   // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, false);   // row_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
    return __builtin_amdgcn_readlane(v, 63);
}

// minimum over each row of 16 lanes, in every lane of the row (butterfly on the DPP data path)
__device__ __forceinline__ int row16_min(int v)
{
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xf, 0xf, false));
    return v;
}
// minimum over the wavefront, wave-uniform result
__device__ __forceinline__ int wave_min(int v)
{
    v = row16_min(v);
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x142, 0xa, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x143, 0xc, 0xf, false));
    return __builtin_amdgcn_readlane(v, 63);
}

__device__ __forceinline__ int reflect101(int p, int len)
{
    // BORDER_REFLECT_101; |overshoot| < len is guaranteed by the callers (19-px border)
    if (p < 0) p = -p;
    if (p >= len) p = 2 * (len - 1) - p;
    return p;
}

}  // namespace orbhip
