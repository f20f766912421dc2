// Internal (non-ABI) definitions shared by the extractor and matcher translation units.
#pragma once
#include "orbhip_common.h"

#include <vector>

struct orbhip_extractor;
namespace orbhip {
struct BlurW { int w[7]; };
struct TileDesc { short level, tx, ty, pad; };
constexpr int kBlurTW = 64, kBlurTH = 58;
// LDS geometry of k_fast_cells, derived on the host from the largest cell of the image size
struct FastLds {
    int strideW;      // row stride in dwords (odd) = the kernel's template argument: staged dwords + one real dword on the left
    int scoreW;       // score-map row stride in dwords (= strideW - 2)
    int img_words, score_words, list_words;
    int list_cap;     // survivor-list entries (uint16); a round with more survivors is repeated in row bands
};
// One FAST cell as k_fast_cells2 wants it: every derived quantity precomputed, one 32-byte scalar load per wavefront
struct alignas(32) FastCell {
    uint32_t src_off;        // byte offset inside a frame's pyramid block of LDS (row 0, col 0): row y0, column gxb - 4
    uint32_t magic;          // ceil(2^18 / ngrp), ngrp = four-pixel groups per detection row
    unsigned short pitch;    // row pitch of the level's padded plane
    short kpx, kpy;          // keypoint = (LDS col + kpx, detection row + kpy) relative to (minBorderX, minBorderY)
    unsigned char sw, sh, a; // sub-image size; a = x0 & 3
    unsigned char img;       // 1: src_off / pitch address the caller's image (level 0 without a materialised plane)
    unsigned char pad[14];
};
static_assert(sizeof(FastCell) == 32, "FastCell is one s_load_dwordx8");
struct FastParams {
    uint32_t frame_bytes, rcp_cells;
    int ncells_total, slot_cap, ini_th, min_th, img_words, score_words, list_cap;
    int dev;   // development builds only: timing floors / stamped kernel
};
// Tables of the pyramid kernels (k_pyr_base / k_pyr_rows): a wavefront owns kPyrRows padded destination rows (its row
// records come through scalar loads) and one destination dword (resize) or 16-byte group (level-0 copy) per lane.
struct PyrRow { int s0, s1; uint32_t B0, B1; };        // source rows of a destination row; Q11 row coefficients << 12
struct PyrCol { uint32_t sel[4]; uint32_t alv[4]; };   // per destination dword: v_perm selectors of the 4 tap pairs inside the
                                                       // 8-byte source window; Q11 column coefficients a0 | a1 << 16
struct PyrCopyCol { uint32_t selA[4], selB[4]; };      // level 0: 16 destination bytes gathered from a 16-byte source window
struct PyrLevelTab {
    const PyrRow *row;       // [prows]                      (resize only)
    const void *col;         // [words] PyrCol / PyrCopyCol
    const uint32_t *lo;      // [words] first source byte of the lane's window
    int words;               // lanes per padded row: dwords (resize) or 16-byte groups (copy)
    int nchunks, chunk_w;    // a row is split into nchunks runs of chunk_w <= 64 lanes
    uint32_t rcp_chunks;     // ceil(2^32 / nchunks)
    int units;               // wavefront work items = row groups x nchunks
    int prows, pitch;        // destination plane
    unsigned plane_off;
    int src_h, src_pitch;    // source: rows (level-0 copy: REFLECT_101 of the row index), pitch for levels >= 2
    unsigned src_off;        // source ROI inside the frame's pyramid block (levels >= 2)
};
int ensure_level0(orbhip_extractor *e, hipStream_t consumer);   // orbhip_extractor.hip
}  // namespace orbhip

struct orbhip_extractor {
    int nfeatures = 0;
    double scaleFactor = 1.2;
    int nlevels = 8, iniTh = 20, minTh = 7, device = 0;
    float sf[ORBHIP_MAX_LEVELS], isf[ORBHIP_MAX_LEVELS], sig2[ORBHIP_MAX_LEVELS], isig2[ORBHIP_MAX_LEVELS];
    int nfeat[ORBHIP_MAX_LEVELS];
    int umax[orbhip::kHalfPatch + 1];
    orbhip::BlurW blurw;
    hipStream_t stream = nullptr;       // stream every launch of this handle goes to
    hipStream_t own_stream = nullptr;   // created with the handle; `stream` may be re-pointed
    // profiling: ring of event sets (kProfEv events per extract call), averaged on read-out
    static constexpr int kProfEv = 5;   // before the pyramid, after pyramid / FAST / octree / descriptors
    static constexpr int kProfRing = 256;
    std::vector<hipEvent_t> ev;         // kProfRing * kProfEv, created lazily
    bool profiling = false;
    long prof_calls = 0;

    // geometry (bound to an image size)
    bool bound = false;
    orbhip::PyrGeom G;
    std::vector<orbhip::CellDesc> cells;
    std::vector<orbhip::TileDesc> tiles;
    int octree_maxn = 512;
    int octree_threads = 256;   // workgroup size of k_octree (256 / 512 / 1024)
    orbhip::FastLds fast_lds;   // k_fast_cells dynamic LDS carve-up
    int fast_lds_bytes = 0;
#ifdef ORBHIP_DEVTOOLS
    int stage_mask = 31;        // development build (tools/coexec.py): stages launch_pipeline runs
    int fast_variant = 0;       // development build (tools/fast_ab.py): 2 stamped build, 3 / 4 timing floors
#endif
    orbhip::CellDesc *d_cells = nullptr;
    std::vector<orbhip::FastCell> cells2;
    orbhip::FastCell *d_cells2 = nullptr;
    orbhip::FastParams fast_params;
    int num_cus = 256;              // compute units of the device (launch shaping)
    orbhip::TileDesc *d_tiles = nullptr;
    int *d_desc_tab = nullptr;      // descriptor kernel tables: disc chunks {dword index, u weights, v weights}[4][64], row-pass items[64]
    bool blur_valid = false;        // d_blur holds the blurred planes of the last batch
    // mvImagePyramid[0] on demand (orbhip_extractor_set_lazy_level0)
    bool lazy_l0 = false;           // the caller's choice
    bool l0_in_image = false;       // the last extraction read level 0 from the image and did not write the padded plane
    bool l0_valid = true;           // d_pyr holds the padded level-0 planes of the last batch
    int cells_stride = 0;           // image row stride the level-0 entries of d_cells2 are built for (0: the padded plane)
    const uint8_t *src_images = nullptr; int src_stride = 0; size_t src_frame_stride = 0;   // image buffer of the last extraction
    hipEvent_t ev_l0 = nullptr;
    // stage gates (orbhip_extractor_set_stage_gate): events other pipelines' handles record / wait for
    hipEvent_t gate_wait[4] = {nullptr, nullptr, nullptr, nullptr}, gate_rec[4] = {nullptr, nullptr, nullptr, nullptr};
    float4 *d_patternf = nullptr;   // rBRIEF pattern as floats: (x0, y0, x1, y1) per test
    uint8_t *d_pyrtab = nullptr;    // row / column tables of the pyramid kernels
    orbhip::PyrLevelTab ptab[ORBHIP_MAX_LEVELS];
    bool ptab_rows[ORBHIP_MAX_LEVELS];   // level goes through the table-driven resize (all its tap windows fit 8 bytes)

    // per-batch-capacity buffers
    int batch_cap = 0;
    int last_batch = 0;
    uint8_t *d_pyr = nullptr, *d_blur = nullptr;
    int *d_cell_cnt = nullptr;
    uint32_t *d_cell_kp = nullptr;
    uint32_t *d_keys = nullptr;
    unsigned short *d_knode = nullptr;
    uint32_t *d_sel = nullptr;
    int *d_sel_cnt = nullptr;
    int *d_status = nullptr;
    // staging for the host-pointer API
    uint8_t *d_img = nullptr; size_t d_img_bytes = 0;
    orbhip_keypoint *d_okp = nullptr; uint8_t *d_odesc = nullptr; int *d_on = nullptr;
    size_t out_slots = 0;   // keypoint slots allocated in d_okp / d_odesc
    int out_batch = 0;      // entries allocated in d_on
    // host-pointer API: H2D copy + 13 launches + 4 D2H copies captured once per (batch, cap) and replayed with one
    // hipGraphLaunch -- a single frame is launch-bound, not GPU-bound
    hipGraphExec_t graph_exec = nullptr;
    int graph_batch = 0, graph_cap = 0;
    // chunked host path: copy-in / copy-out streams and per-chunk events (created on first use)
    hipStream_t s_in = nullptr, s_out = nullptr;
    std::vector<hipEvent_t> ev_chunk;
    // pinned host staging (pageable 2-D copies are an order of magnitude slower than one pinned DMA)
    uint8_t *h_in = nullptr; size_t h_in_bytes = 0;
    uint8_t *h_out = nullptr; size_t h_out_bytes = 0;
};

