// MI355X (gfx950) ORB extractor: pyramid, per-cell FAST-9/16 + NMS, octree
// distribution, IC-angle, 7x7 blur, steered rBRIEF -- behind the C ABI of
// include/orbhip.h.  Replaces ORBextractor (src/ORBextractor.cc:410-1132).
//
// Batch-first design: every kernel takes a frame index in blockIdx.y, so one
// launch covers a whole batch of frames; a frame's pyramid (8 padded uchar
// planes) stays resident in HBM between the stages.  Integer/bitwise path:
// no MFMA.  All float arithmetic that feeds a rounding decision is written
// with explicit, un-contracted operations.
#include "orbhip_internal.h"
#include "rbrief_pattern.h"

#include <cfloat>
#include <cmath>
#include <ctime>
#include <mutex>
#include <new>
#include <vector>

namespace orbhip {

static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------
// XCD-aware work mapping (speed only, never correctness): workgroups are dealt round-robin over the
// 8 XCDs in linear order, each XCD has a private 4 MB L2.  Re-index so that XCD k owns a contiguous
// run of (frame, unit) pairs: with a batch that is a multiple of 8 every kernel of the pipeline
// then touches frame f from the same XCD, and neighbouring cells/tiles share their cache lines.
__device__ __forceinline__ void xcd_remap(int &unit, int &frame)
{
    const unsigned T = gridDim.x * gridDim.y, lin = blockIdx.y * gridDim.x + blockIdx.x;
    const unsigned q = T >> 3, r = T & 7, x = lin & 7;
    const unsigned lin2 = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (lin >> 3);
    frame = (int)(lin2 / gridDim.x);
    unit = (int)(lin2 - (unsigned)frame * gridDim.x);
}

__device__ __forceinline__ int wave_reduce_add(int v) { return wave_sum(v); }

// Exclusive scan of one int per thread over a T-thread block (T = 256, 512 or 1024).  `sh` = T / 64 ints of LDS.
// Returns the exclusive prefix; *total receives the block sum.  Ends with a barrier.
template <int T>
__device__ __forceinline__ int block_excl_scan(int v, int *sh, int *total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int incl = wave_incl_scan_add(v);
    __syncthreads();  // protect sh reuse across consecutive calls
    if (lane == 63) sh[wave] = incl;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < T / 64; ++w) {
        int s = sh[w];
        if (w < wave) base += s;
        tot += s;
    }
    *total = tot;
    return base + incl - v;
}
__device__ __forceinline__ int block_excl_scan256(int v, int *sh, int *total) { return block_excl_scan<256>(v, sh, total); }

typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
typedef short i16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t udot2(uint32_t a, uint32_t b, uint32_t c)
{
    return __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2_t, a), __builtin_bit_cast(u16x2_t, b), c, false);
}

// cv::fastAtan2 (degrees in [0,360)); same float operation sequence as the oracle.
__device__ __forceinline__ float fast_atan2_deg(float y, float x)
{
    const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
    const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
    const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
    const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
    float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = __fdiv_rn(ay, __fadd_rn(ax, (float)DBL_EPSILON));
        c2 = __fmul_rn(c, c);
        a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
    } else {
        c = __fdiv_rn(ax, __fadd_rn(ay, (float)DBL_EPSILON));
        c2 = __fmul_rn(c, c);
        a = __fsub_rn(90.f, __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c));
    }
    if (x < 0) a = __fsub_rn(180.f, a);
    if (y < 0) a = __fsub_rn(360.f, a);
    return a;
}

// Deterministic stand-in for (float)cos / (float)sin of ORBextractor.cc:113: IEEE double,
// separate multiplies and adds (no FMA), one final rounding to float.  Bit-identical to
// oracle_det_sincos by construction.
__device__ __forceinline__ void det_sincos(float angle_rad, float *c, float *s)
{
    const double two_over_pi = 6.36619772367581382433e-01;
    const double pio2_hi = 1.57079632673412561417e+00;
    const double pio2_lo = 6.07710050650619224932e-11;
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double x = (double)angle_rad;
    double kd = floor(__dadd_rn(__dmul_rn(x, two_over_pi), 0.5));
    int k = (int)kd;
    double r = __dsub_rn(__dsub_rn(x, __dmul_rn(kd, pio2_hi)), __dmul_rn(kd, pio2_lo));
    double z = __dmul_rn(r, r);
    double ps = __dadd_rn(S1, __dmul_rn(z, __dadd_rn(S2, __dmul_rn(z, __dadd_rn(S3, __dmul_rn(z, __dadd_rn(S4, __dmul_rn(z, __dadd_rn(S5, __dmul_rn(z, S6))))))))));
    double sn = __dadd_rn(r, __dmul_rn(__dmul_rn(r, z), ps));
    double pc = __dadd_rn(C1, __dmul_rn(z, __dadd_rn(C2, __dmul_rn(z, __dadd_rn(C3, __dmul_rn(z, __dadd_rn(C4, __dmul_rn(z, __dadd_rn(C5, __dmul_rn(z, C6))))))))));
    double cs = __dadd_rn(__dsub_rn(1.0, __dmul_rn(0.5, z)), __dmul_rn(__dmul_rn(z, z), pc));
    double so, co;
    switch (k & 3) {
    case 0: so = sn; co = cs; break;
    case 1: so = cs; co = -sn; break;
    case 2: so = -sn; co = -cs; break;
    default: so = -cs; co = sn; break;
    }
    *s = (float)so;
    *c = (float)co;
}

// ---------------------------------------------------------------------------
// K1: pyramid.  Level 0 = copyMakeBorder(image, REFLECT_101); level l =
// resize(level l-1, INTER_LINEAR) + copyMakeBorder (ORBextractor.cc:1107-1132).
// The table-driven kernels further down are the default path; k_pyr_resize is the
// general kernel for levels whose tap groups do not fit an 8-byte source window
// (scale factors > 2.3): one thread produces 4 consecutive bytes of 4 padded
// destination rows, border pixels recompute the reflected interior pixel.
// ---------------------------------------------------------------------------
// resize(INTER_LINEAR, 8U) in OpenCV's fixed-point arithmetic.  A thread derives the source column / row and the
// Q11 coefficient pair of each of its 4 destination columns and 4 rows in registers (resize_coef).  Groups whose four
// taps fit one 8-byte source window (always, for scale factors <= 2.3) take the fast path: one byte-aligned 8-byte
// load per source row, taps picked with v_perm_b32, horizontal filter v_dot2_u32_u16; groups that touch the
// REFLECT_101 border revisit the reflected interior columns through the same path.
__device__ __forceinline__ uint32_t resize_px(const uint8_t *S0, const uint8_t *S1, int sx, int a0, int a1, int b0, int b1)
{
    // sx+1 may be the first border byte when sx == sw-1; a1 is 0 there.
    const int r0 = __mul24(S0[sx], a0) + __mul24(S0[sx + 1], a1);
    const int r1 = __mul24(S1[sx], a0) + __mul24(S1[sx + 1], a1);
    return (uint32_t)(((__mul24(b0, r0 >> 4) >> 16) + (__mul24(b1, r1 >> 4) >> 16) + 2) >> 2) & 0xffu;
}

constexpr int kPyrRows = 4;   // destination rows per thread (independent loads in flight)

// OpenCV's fixed-point bilinear coefficients of one destination coordinate d (resize INTER_LINEAR, 8U):
// f = (float)((d + 0.5) * scale - 0.5) in double, s = floor(f), f -= s, clamped at the ends; coefficients
// cvRound((1 - f) * 2048), cvRound(f * 2048).  Same operations as the host-built tables (bind_geometry), so a
// thread can derive its coefficients in registers instead of waiting for a table load before it can address
// its source pixels.
__device__ __forceinline__ void resize_coef(int d, double scale, int slen, bool clamp_both, int &s0, int &s1, int &c0, int &c1)
{
    float f = (float)__dsub_rn(__dmul_rn(__dadd_rn((double)d, 0.5), scale), 0.5);
    int s = (int)floorf(f);
    f = __fsub_rn(f, (float)s);
    if (clamp_both) {      // rows: both taps clamped independently, the fraction is kept
        s0 = s < 0 ? 0 : (s < slen ? s : slen - 1);
        s1 = s + 1 < 0 ? 0 : (s + 1 < slen ? s + 1 : slen - 1);
    } else {               // columns: the fraction is dropped at the ends
        if (s < 0) { f = 0.f; s = 0; }
        if (s >= slen - 1) { f = 0.f; s = slen - 1; }
        s0 = s; s1 = s + 1;
    }
    c0 = __float2int_rn(__fmul_rn(__fsub_rn(1.f, f), 2048.f));
    c1 = __float2int_rn(__fmul_rn(f, 2048.f));
}
struct ResizeScale { double sx, sy; };

__global__ __launch_bounds__(256) void k_pyr_resize(uint8_t *__restrict__ pyr, PyrGeom G, int level, ResizeScale RS)
{
    const LevelGeom L = G.lv[level];
    const LevelGeom P = G.lv[level - 1];
    const int words = L.pitch >> 2;
    const int nquads = (L.prows + kPyrRows - 1) / kPyrRows;
    int bx, fr;
    xcd_remap(bx, fr);
    const int idx = bx * 256 + threadIdx.x;
    if (idx >= words * nquads) return;
    const int rq = idx / words, pw = idx - rq * words;
    uint8_t *frame = pyr + (size_t)fr * G.frame_bytes;
    const uint8_t *sroi = frame + P.plane_off + (size_t)kEdge * P.pitch + kPadL;
    uint8_t *dst = frame + L.plane_off;
    const int x0 = pw * 4 - kPadL;
    // destination columns of this group after REFLECT_101 (border groups revisit interior columns,
    // possibly in descending order); pixels beyond the 19-px frame are written as 0
    int sxk[4];
    uint32_t alv[4];
    uint32_t vmask = 0;
    const bool interior = x0 >= 0 && x0 + 3 < L.w;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int x = x0 + k;
        if (x >= -kEdge && x < L.w + kEdge) vmask |= 0xffu << (8 * k);
        const int dx = interior ? x : reflect101(min(max(x, -kEdge), L.w + kEdge - 1), L.w);
        int s1, a0, a1;
        resize_coef(dx, RS.sx, P.w, false, sxk[k], s1, a0, a1);
        alv[k] = (uint32_t)a0 | ((uint32_t)a1 << 16);
    }
    const int lo = min(min(sxk[0], sxk[1]), min(sxk[2], sxk[3]));
    const int hi = max(max(sxk[0], sxk[1]), max(sxk[2], sxk[3]));
    short4 yt[kPyrRows];
#pragma unroll
    for (int r = 0; r < kPyrRows; ++r) {
        const int py = min(rq * kPyrRows + r, L.prows - 1);
        int s0, s1, b0, b1;
        resize_coef(reflect101(py - kEdge, L.h), RS.sy, P.h, true, s0, s1, b0, b1);
        yt[r] = make_short4((short)s0, (short)s1, (short)b0, (short)b1);
    }
    uint32_t out[kPyrRows];
    if (hi - lo <= 6) {   // the 8-byte source window covers all four taps (scale factors <= 2.3)
        unsigned long long w0[kPyrRows], w1[kPyrRows];   // source bytes lo .. lo+7 of the two source rows (unaligned loads)
#pragma unroll
        for (int r = 0; r < kPyrRows; ++r) {
            // rows < 4038 and pitch < 4200 fit the full-rate 24-bit multiplier (a 64-bit mad is quarter rate)
            __builtin_memcpy(&w0[r], sroi + (uint32_t)__mul24((int)yt[r].x, P.pitch) + lo, 8);
            __builtin_memcpy(&w1[r], sroi + (uint32_t)__mul24((int)yt[r].y, P.pitch) + lo, 8);
        }
        // v_perm_b32 selector of tap k: bytes (rel, rel + 1) of the 8-byte window, zero-extended to two uint16 lanes
        uint32_t sel[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t rel = (uint32_t)(sxk[k] - lo);
            sel[k] = 0x0c000c00u | rel | ((rel + 1u) << 16);
        }
#pragma unroll
        for (int r = 0; r < kPyrRows; ++r) {
            const int b0 = yt[r].z, b1 = yt[r].w;
            const uint32_t w0l = (uint32_t)w0[r], w0h = (uint32_t)(w0[r] >> 32), w1l = (uint32_t)w1[r], w1h = (uint32_t)(w1[r] >> 32);
            uint32_t acc = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t h0 = udot2(__builtin_amdgcn_perm(w0h, w0l, sel[k]), alv[k], 0);
                const uint32_t h1 = udot2(__builtin_amdgcn_perm(w1h, w1l, sel[k]), alv[k], 0);
                // b <= 2048 and h >> 4 < 2^15: the 24-bit multiplier is exact and full rate (v_mul_lo_u32 is quarter rate)
                const uint32_t v = (uint32_t)(((__mul24(b0, (int)(h0 >> 4)) >> 16) + (__mul24(b1, (int)(h1 >> 4)) >> 16) + 2) >> 2) & 0xffu;
                acc |= v << (8 * k);
            }
            out[r] = acc & vmask;
        }
    } else {
#pragma unroll
        for (int r = 0; r < kPyrRows; ++r) {
            const uint8_t *S0 = sroi + (uint32_t)__mul24((int)yt[r].x, P.pitch), *S1 = sroi + (uint32_t)__mul24((int)yt[r].y, P.pitch);
            uint32_t acc = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                acc |= resize_px(S0, S1, sxk[k], (int)(alv[k] & 0xffff), (int)(alv[k] >> 16), yt[r].z, yt[r].w) << (8 * k);
            out[r] = acc & vmask;
        }
    }
#pragma unroll
    for (int r = 0; r < kPyrRows; ++r) {
        const int py = rq * kPyrRows + r;
        if (py < L.prows) *reinterpret_cast<uint32_t *>(dst + (uint32_t)__mul24(py, L.pitch) + pw * 4) = out[r];
    }
}

// ---------------------------------------------------------------------------
// Table-driven pyramid kernels (the default path; k_pyr_level0 / k_pyr_resize above remain for levels whose scale
// factor puts a tap group outside an 8-byte window).  A WAVEFRONT owns kPyrRows consecutive padded destination
// rows and a run of <= 64 lanes of one row: everything that depends on the rows (source rows, row coefficients,
// row addresses) is wave-uniform and lives on the scalar unit; everything that depends on the column comes from a
// per-lane table record that is the same for every row and every frame.  The REFLECT_101 frame is part of the
// tables (a border lane gathers the reflected interior columns through the same instructions), so no wavefront
// ever runs a second code path.
//  * level 0 (copyMakeBorder of the input image): per lane one unaligned 16-byte load + six v_perm_b32 + one
//    16-byte store per row;
//  * level l >= 1 (resize INTER_LINEAR 8U + copyMakeBorder): per lane two unaligned 8-byte loads per row, and per
//    pixel and source row v_perm_b32 (tap pair) + v_dot2_u32_u16 (Q11 horizontal filter) + v_and (the >> 4 of
//    OpenCV's vertical pass, kept in place) + v_mul_hi_u32_u24 with the row coefficient << 12 (= (b * (h >> 4)) >> 16);
//    a source row shared by two consecutive destination rows is filtered once (the choice is wave-uniform).
// k_pyr_base runs level 0 and level 1 in ONE launch: level 1 reads the input image itself, not the padded level-0
// plane, so the two do not depend on each other.
// ---------------------------------------------------------------------------
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t mulhi24_s(uint32_t a, uint32_t sb)
{
    uint32_t r;
    asm("v_mul_hi_u32_u24 %0, %2, %1" : "=v"(r) : "v"(a), "s"(sb));
    return r;
}
__device__ __forceinline__ PyrRow load_row_rec(const PyrRow *rows, int py)
{
    const u32x4_t raw = *reinterpret_cast<const __attribute__((address_space(4))) u32x4_t *>(reinterpret_cast<uintptr_t>(rows + py));
    return __builtin_bit_cast(PyrRow, raw);
}

// unit -> (row group, lane's column); false when the lane has nothing to do
__device__ __forceinline__ bool pyr_unit(const PyrLevelTab &T, int unit, int &rg, int &col)
{
    rg = T.nchunks == 1 ? unit : (int)__umulhi((uint32_t)unit, T.rcp_chunks);
    const int c = unit - rg * T.nchunks;
    const int lane = threadIdx.x & 63;
    col = c * T.chunk_w + lane;
    return lane < T.chunk_w && col < T.words;
}

__device__ __forceinline__ void pyr_copy_rows(const PyrLevelTab &T, int unit, const uint8_t *__restrict__ src, int stride,
                                              uint8_t *__restrict__ dplane)
{
    int rg, g;
    if (!pyr_unit(T, unit, rg, g)) return;
    const PyrCopyCol C = reinterpret_cast<const PyrCopyCol *>(T.col)[g];
    const uint32_t base = T.lo[g];
    uint4 win[kPyrRows];
#pragma unroll
    for (int r = 0; r < kPyrRows; ++r) {
        const int py = min(rg * kPyrRows + r, T.prows - 1);
        const int sy = reflect101(py - kEdge, T.src_h);
        __builtin_memcpy(&win[r], src + (size_t)((uint32_t)sy * (uint32_t)stride) + base, 16);
    }
#pragma unroll
    for (int r = 0; r < kPyrRows; ++r) {
        const int py = rg * kPyrRows + r;
        if (py >= T.prows) break;
        uint4 o;
        o.x = __builtin_amdgcn_perm(win[r].y, win[r].x, C.selA[0]) | __builtin_amdgcn_perm(win[r].w, win[r].z, C.selB[0]);
        o.y = __builtin_amdgcn_perm(win[r].y, win[r].x, C.selA[1]) | __builtin_amdgcn_perm(win[r].w, win[r].z, C.selB[1]);
        o.z = __builtin_amdgcn_perm(win[r].y, win[r].x, C.selA[2]) | __builtin_amdgcn_perm(win[r].w, win[r].z, C.selB[2]);
        o.w = __builtin_amdgcn_perm(win[r].y, win[r].x, C.selA[3]) | __builtin_amdgcn_perm(win[r].w, win[r].z, C.selB[3]);
        *reinterpret_cast<uint4 *>(dplane + (size_t)((uint32_t)py * (uint32_t)T.pitch) + (uint32_t)g * 16u) = o;
    }
}

__device__ __forceinline__ void pyr_hfilter(unsigned long long w, const PyrCol &C, uint32_t (&h)[4])
{
    const uint32_t wl = (uint32_t)w, wh = (uint32_t)(w >> 32);
#pragma unroll
    for (int k = 0; k < 4; ++k) h[k] = udot2(__builtin_amdgcn_perm(wh, wl, C.sel[k]), C.alv[k], 0) & ~15u;
}

__device__ __forceinline__ void pyr_resize_rows(const PyrLevelTab &T, int unit, const uint8_t *__restrict__ sroi, int spitch,
                                                uint8_t *__restrict__ dplane)
{
    int rg, pw;
    if (!pyr_unit(T, unit, rg, pw)) return;
    const PyrCol C = reinterpret_cast<const PyrCol *>(T.col)[pw];
    const uint32_t lo = T.lo[pw];
    PyrRow R[kPyrRows];
    unsigned long long w0[kPyrRows], w1[kPyrRows];
#pragma unroll
    for (int r = 0; r < kPyrRows; ++r) {
        R[r] = load_row_rec(T.row, min(rg * kPyrRows + r, T.prows - 1));
        // the upper source row of a destination row that is the previous row's lower one is not loaded again (wave-uniform):
        // a quarter of the kernel's load instructions (the chain of seven launches: 78 -> 69.5 us per 64 frames).  Measured
        // and dropped after that: staging a 16-row tile's source rows through LDS with 16-byte loads and reading the windows
        // from LDS (4.5x fewer load instructions, bit-exact, 75 us), 6 / 8 / 12 destination rows per wavefront (73 / 84 / 73 us).
        // Knock-out timing of the chain (not bit-exact, timing only): 70 us as is; without the stores 57, without the source
        // loads 56, without the table loads 61, with none of them 46 = the arithmetic (27 us of issue) plus seven dependent
        // launches' fixed costs -- the memory operations add 24 us on top of a floor that is already 46
        w0[r] = 0;
        if (!(r > 0 && R[r].s0 == R[r - 1].s1))
            __builtin_memcpy(&w0[r], sroi + (size_t)((uint32_t)R[r].s0 * (uint32_t)spitch) + lo, 8);
        __builtin_memcpy(&w1[r], sroi + (size_t)((uint32_t)R[r].s1 * (uint32_t)spitch) + lo, 8);
    }
    uint32_t ha[4], hb[4];
#pragma unroll
    for (int r = 0; r < kPyrRows; ++r) {
        if (r > 0 && R[r].s0 == R[r - 1].s1) {   // wave-uniform: the upper source row is the previous row's lower one
#pragma unroll
            for (int k = 0; k < 4; ++k) ha[k] = hb[k];
        } else {
            pyr_hfilter(w0[r], C, ha);
        }
        pyr_hfilter(w1[r], C, hb);
        uint32_t acc = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t v = (mulhi24_s(ha[k], R[r].B0) + mulhi24_s(hb[k], R[r].B1) + 2u) >> 2;
            acc |= v << (8 * k);
        }
        const int py = rg * kPyrRows + r;
        if (py < T.prows) *reinterpret_cast<uint32_t *>(dplane + (size_t)((uint32_t)py * (uint32_t)T.pitch) + (uint32_t)pw * 4u) = acc;
    }
}

// level 0 + level 1 of a frame in one launch: workgroups [0, nb0) copy, the rest resize from the input image
__global__ __launch_bounds__(256) void k_pyr_base(const uint8_t *__restrict__ images, int stride, size_t frame_stride,
                                                  uint8_t *__restrict__ pyr, uint32_t frame_bytes, PyrLevelTab T0,
                                                  PyrLevelTab T1, int nb0, int *__restrict__ status)
{
    int bx, fr;
    xcd_remap(bx, fr);
    if (status && bx == 0 && threadIdx.x == 0) status[fr] = 0;   // the frame's status word starts every extraction at ORBHIP_OK
                                                                 // (null: the lazy level-0 copy after an extraction)
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint8_t *src = images + (size_t)fr * frame_stride;
    uint8_t *frame = pyr + (size_t)fr * frame_bytes;
    if (bx < nb0) {
        const int unit = bx * 4 + wave;
        if (unit < T0.units) pyr_copy_rows(T0, unit, src, stride, frame + T0.plane_off);
    } else {
        const int unit = (bx - nb0) * 4 + wave;
        if (unit < T1.units) pyr_resize_rows(T1, unit, src, stride, frame + T1.plane_off);
    }
}

__global__ __launch_bounds__(256) void k_pyr_rows(uint8_t *__restrict__ pyr, uint32_t frame_bytes, PyrLevelTab T)
{
    int bx, fr;
    xcd_remap(bx, fr);
    const int unit = bx * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (unit >= T.units) return;
    uint8_t *frame = pyr + (size_t)fr * frame_bytes;
    pyr_resize_rows(T, unit, frame + T.src_off, T.src_pitch, frame + T.plane_off);
}

constexpr int kSubMax = 72;              // max (wCell+6), (hCell+6)
constexpr int kFastStageU = 12;         // k_fast_cells: row groups per staging batch (48 rows of a stride <= 16 dwords)
constexpr int kFastLdsPerCu = 160 * 1024;  // gfx950
constexpr int kFastLdsGranule = 1280;      // LDS is handed out in 1280-byte granules on gfx950 (tools/micro/lds_oob.hip: a 4 KB request owns 5 KB)
constexpr int kFastLead = 1;             // k_fast_cells: LDS column of sub-image x is x + kFastLead + 4

__device__ __forceinline__ int min3i(int a, int b, int c) { return min(min(a, b), c); }
__device__ __forceinline__ int max3i(int a, int b, int c) { return max(max(a, b), c); }
__device__ __forceinline__ int lane_prefix(unsigned long long m)
{
    return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
}

// ---------------------------------------------------------------------------
// K2+K3: FAST-9/16 + NMS, one WAVEFRONT per reference cell (= the reference's cv::FAST call(s) for
// that cell, ORBextractor.cc:789-829); 64-thread workgroups, every hand-off is wave-local.
//  1. staging: the (wCell+6)x(hCell+6) sub-image goes to LDS as aligned dwords; lane = (row mod RPI,
//     dword column), the row advance is scalar.  The LDS row stride SW is a template parameter, so
//     every LDS access of the hot loops is "one base register + immediate offset".
//  2. dense compass pre-test at the pass threshold (ring pixels 0,4,8,12: an arc of 9 holds one pixel
//     of every opposite pair), 4 px per lane, on packed uint16 WITHOUT unpacking: a dword of four
//     pixels is read as two uint16 lanes whose high bytes are pixels 1 and 3 -- the low byte only
//     perturbs the value by < 1 gray level, which can let a non-corner through (the score re-checks
//     exactly) but never drops one; pixels 0 and 2 use the same dwords masked with 0x00ff00ff
//     (exact).  With A = min(max(p0,p8), max(p4,p12)) and B = max(min(p0,p8), min(p4,p12)) a pixel is
//     a candidate iff max(A - v, v - B) > t (saturating differences): ONE comparison per pixel writes
//     its lane mask (v_cmp -> SGPR pair), the masks are combined on the scalar unit, the ordered
//     survivor compaction is one v_mbcnt chain and a running LDS pointer.  The item's row, list entry
//     and LDS address come from one multiply-shift and two multiply-adds of a strength-reduced counter.
//  3. the threshold-independent score S = max(dark,bright)-1 (cornerScore<16>) only for survivors:
//     a survivor's polarity follows from its compass pixels, so the min3 network runs once, on x = p
//     (bright) or x = ~p (dark): the centre value only shifts an arc's minimum, it is added at the end
//     (S = max_k min_arc(x) + (bright ? ~v : v)); a pixel that passes the compass test for both
//     polarities (0.3 % of the survivors) takes the max3 network too, under a wave-uniform branch.
//     Exactness: a brighter and a darker arc of 9 cannot coexist on a ring of 16, so at most one
//     polarity exceeds t and the other one is <= t < S.
//  4. NMS (strict maximum over the 8 neighbours; outside the detection rectangle = 0, like the
//     reference's zero-initialised score rows) over the corner list, fused with the emission.
//  5. pass 0 runs at iniThFAST; a cell that keeps nothing repeats 2-4 at minThFAST (:809-816).
// Every compaction is stable and lanes walk the cell row-major, so the emitted list is already in
// the reference's order.
// ---------------------------------------------------------------------------
typedef unsigned short u16x2_v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u16x2_v as_u16x2(uint32_t v) { return __builtin_bit_cast(u16x2_v, v); }
__device__ __forceinline__ uint32_t as_u32(u16x2_v v) { return __builtin_bit_cast(uint32_t, v); }
// full-rate 24-bit multiplies, forced: LLVM folds the mul24 intrinsics back into v_mul_lo_u32 when it cannot
// bound an operand
__device__ __forceinline__ uint32_t mulu24(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ int madi24(int a, int b, int c)
{
    int r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// same with a wave-uniform second factor held in an SGPR (one constant-bus operand)
__device__ __forceinline__ uint32_t mulu24_s(uint32_t a, uint32_t sb)
{
    uint32_t r;
    asm("v_mul_u32_u24 %0, %2, %1" : "=v"(r) : "v"(a), "s"(sb));
    return r;
}
__device__ __forceinline__ int madi24_s(int a, int sb, int c)
{
    int r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(sb), "v"(c));
    return r;
}
// x + (this lane's bit of a lane mask): the mask is the carry-in, one instruction
__device__ __forceinline__ int add_lane_bit(int x, unsigned long long mask)
{
    int r;
    asm("v_addc_co_u32 %0, vcc, 0, %1, %2" : "=v"(r) : "v"(x), "s"(mask) : "vcc");
    return r;
}

// Diagnostic build only (STAMPS = true is instantiated under -DORBHIP_DEVTOOLS alone): per-phase s_memtime sums of all
// waves, read by tools/fast_ab.py through orbhip_dev_fast_stamps.  Shares, not lengths, are meaningful (the stamps fence).
#ifdef ORBHIP_DEVTOOLS
__device__ unsigned long long g_fast_stamps[8];
#endif
__device__ __forceinline__ unsigned long long stamp_now()
{
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}

#ifndef ORBHIP_FAST_WAVES
#define ORBHIP_FAST_WAVES 8   // resident waves per SIMD = what the LDS share of bind_geometry admits (32 workgroups per CU)
#endif
template <int SW, bool STAMPS = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(ORBHIP_FAST_WAVES, ORBHIP_FAST_WAVES))) void k_fast_cells(const uint8_t *__restrict__ pyr, const FastCell *__restrict__ cells,
                                                    int *__restrict__ cell_cnt, uint32_t *__restrict__ cell_kp,
                                                    FastParams P, const uint8_t *__restrict__ images, size_t frame_stride)
{
    extern __shared__ uint32_t lds[];
    constexpr int SB = SW * 4;                 // bytes per staged row
    constexpr int SS = SB - 8;                 // bytes per score-map row (detection width + 2 halo bytes fit: dw + 2 <= sw - 4)
    constexpr int LPR = SW <= 16 ? 16 : 32;    // lanes per staged row
    constexpr int RPI = 64 / LPR;              // rows per staging instruction
    constexpr int U = kFastStageU;             // staging loads in flight per lane
    uint32_t *simg = lds;                                   // staged sub-image, LDS col 0 = global column gxb - 4
    uint32_t *sscore = simg + P.img_words;                   // score map with a 1-px zero halo
    typedef __attribute__((address_space(3))) unsigned short lds_u16;
    lds_u16 *slist = (lds_u16 *)(sscore + P.score_words);   // (detection row << 8) | detection column

    unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0}, tprev = 0;
#define FAST_STAMP(i) do { if constexpr (STAMPS) { const unsigned long long t_ = stamp_now(); tacc[i] += t_ - tprev; tprev = t_; } } while (0)
    if constexpr (STAMPS) tprev = stamp_now();
    // XCD-aware (cell, frame) of this wavefront (see xcd_remap); the division by the cell count is a multiply by the
    // host-computed reciprocal (exact for lin2 < 2^32 / ncells, checked on the host), all on the scalar unit
    int cell, fr;
    {
        const unsigned T = gridDim.x * gridDim.y, lin = blockIdx.y * gridDim.x + blockIdx.x;
        const unsigned q = T >> 3, r = T & 7, x = lin & 7;
        const unsigned lin2 = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (lin >> 3);
        fr = (int)__umulhi(lin2, P.rcp_cells);
        cell = (int)(lin2 - (unsigned)fr * gridDim.x);
    }
    const int lane = threadIdx.x;
    // everything this cell needs is one 32-byte record, fetched with one scalar load (constant address space)
    typedef uint32_t u32x8_t __attribute__((ext_vector_type(8)));
    const u32x8_t raw = *reinterpret_cast<const __attribute__((address_space(4))) u32x8_t *>(reinterpret_cast<uintptr_t>(cells + cell));
    const FastCell cd = __builtin_bit_cast(FastCell, raw);
    const int sh = cd.sh, dw = cd.sw - 6, dh = cd.sh - 6;   // sub-image rows; detection rectangle
    const int pitch = cd.pitch;
    const size_t out_cell = (size_t)fr * P.ncells_total + cell;

    if (dw <= 0 || dh <= 0) {  // cv::FAST on an image narrower than 7 px finds nothing
        if (lane == 0) cell_cnt[out_cell] = 0;
        return;
    }
    FAST_STAMP(0);   // prologue
#ifdef ORBHIP_DEVTOOLS
    if (P.dev == 3) { if (lane == 0) cell_cnt[out_cell] = 0; return; }   // timing floor: launch + prologue only
#endif
    // LDS column of sub-image x: col = x + a + 4 (a = misalignment of x0; the dword on the left holds real pixels).
    // Lane = (row mod RPI, dword column); a batch is U row groups, all loads issued before the first LDS store.  Row
    // groups past the sub-image re-read its last rows (scalar clamp) into LDS rows that nothing looks at.
    {
        const int c = lane & (LPR - 1), r = lane / LPR;
        // level-0 cells of a handle that does not materialise mvImagePyramid[0] read the caller's image itself (a FAST
        // sub-image never reaches the REFLECT_101 frame: the cell grid starts 13 px inside the level)
        const uint8_t *sb = cd.img ? images + (size_t)fr * frame_stride + cd.src_off : pyr + (size_t)fr * P.frame_bytes + cd.src_off;
        const uint32_t voff = (uint32_t)(__mul24(r, pitch) + 4 * c);
        uint32_t *sdst = simg + (r * SW + c);
        if (c < SW) {
            for (int r0 = 0; r0 < sh; r0 += RPI * U) {
                uint32_t v[U];
#pragma unroll
                for (int u = 0; u < U; ++u)
                    v[u] = *reinterpret_cast<const uint32_t *>(sb + (uint32_t)(min(r0 + u * RPI, sh - 1) * pitch) + voff);
#pragma unroll
                for (int u = 0; u < U; ++u) sdst[(r0 + u * RPI) * SW] = v[u];
            }
        }
    }
    // score-map clear, 16 bytes per lane (the map starts on a 16-byte boundary: img_words is a multiple of 4; the last
    // store may run a few bytes into the survivor list, which nothing has written yet)
    for (int i = lane; i < ((dh + 2) * SS + 15) / 16; i += 64) reinterpret_cast<uint4 *>(sscore)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    FAST_STAMP(1);   // staging (global -> LDS) + score-map clear
#ifdef ORBHIP_DEVTOOLS
    if (P.dev == 4) { if (lane == 0) cell_cnt[out_cell] = (int)simg[lane] & 0; return; }   // timing floor: + staging
#endif

    // column groups: group g' covers LDS cols 8 + 4g' .. 8 + 4g' + 3; the detection columns are LDS cols [8, 8 + dw)
    // (every cell is staged with the same lead, bind_geometry: the first centre column sits on a dword boundary)
    static_assert(kFastLead + 4 + 3 == 8, "the first detection column is LDS column 8");
    const int ngrp = (dw + 3) >> 2;
    const int nwork = ngrp * dh;                        // (row, group) work items, row-major
    const uint32_t magic = cd.magic;                    // ceil(2^18 / ngrp): floor(i / ngrp) = (4 i * magic) >> 20 for i < 4096
    const int lj = (dw - 1) & 3;                        // last valid pixel of a row's last group
    const uint8_t *img8 = reinterpret_cast<const uint8_t *>(simg);
    uint8_t *score8 = reinterpret_cast<uint8_t *>(sscore);
    uint32_t *out = cell_kp + out_cell * P.slot_cap;
    const int kpx = cd.kpx + 8, kpy = cd.kpy;           // keypoint = (entry column + kpx, y + kpy) relative to (minBorderX, minBorderY)
    // list entry of a pixel: (y << 8) | (4 g' + j) = (detection row << 8) | detection column.  With it4 = 4 * item:
    //   entry of pixel 0 = y * K1 + it4,  LDS byte address of the item's dword (row y, LDS col 8 + 4 g') = entry - y * K2 + 8
    const int K1 = 256 - 4 * ngrp;
    constexpr int K2 = 256 - SB;
    const uint32_t last4 = 4u * (uint32_t)(ngrp - 1);
    // pixels of a row's last group beyond the detection rectangle
    unsigned long long en[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        // built by an opaque instruction: the compiler would re-derive the masks from lj inside the loop
        unsigned half;
        asm("s_ashr_i32 %0, %1, 31" : "=s"(half) : "s"(lj - j) : "scc");      // j > lj ? ~0 : 0
        en[j] = ((unsigned long long)half << 32) | half;
    }
    // The survivor list holds P.list_cap entries -- far fewer than the cell has pixels, so that 8 wavefronts per SIMD fit
    // the LDS.  It is the LAST array of the workgroup's LDS: entries past the end fall outside the allocation and are
    // dropped by the hardware's LDS range check (DS addresses are checked against the wavefront's LDS_SIZE: out-of-range
    // writes are ignored -- tools/micro/lds_oob.hip: 32,768 workgroups writing 8 KB behind their 4 KB, 50 rounds, no word of
    // any workgroup changed, only the workgroup's own granule padding took the data; the dense-image tests and sweeps
    // would show a neighbouring workgroup's corrupted tile otherwise.
    // A software check -- stop a round when fewer than 256 entries of room are left -- was measured at +4 %: it sends every
    // cell within 256 entries of the capacity to the band path).  A round whose survivor count exceeds the capacity is repeated in bands
    // of rows that cannot overflow (4 * ngrp pixels per row); banded rounds leave their corners in the score map only and
    // the NMS walks the map instead of the corner list.
    const int band_items = (P.list_cap / (4 * ngrp)) * ngrp;
    int total = 0;
    for (int pass = 0; pass < 2; ++pass) {
        // the reference calls FAST(iniThFAST) first and FAST(minThFAST) only for cells that kept nothing (:809-816)
        const int tmin = pass ? P.min_th : P.ini_th;
        // thresholds of the packed test: pixels 1, 3 sit in the HIGH byte of their 16-bit lane (t * 256), pixels 0, 2 are
        // exact; the high lane compares as a 32-bit number whose low half only breaks ties (passes, never drops)
        const uint32_t Tn_lo = (uint32_t)tmin << 8, Tn_hi = ((uint32_t)tmin << 24) | 0xffffu;
        const uint32_t Te_lo = (uint32_t)tmin, Te_hi = ((uint32_t)tmin << 16) | 0xffffu;
        int tv;                                      // tmin in a VGPR: an SGPR operand makes a vector instruction slow
        asm("v_mov_b32 %0, %1" : "=v"(tv) : "s"(tmin));
        int band_lo = 0, band_hi = nwork, ncorn = 0;
        bool banded = false;
        for (;;) {
            // ---- dense compass pre-test over items [band_lo, band_hi), survivors -> slist in row-major order ----
            int nsurv = 0;
            uint32_t it4 = 4u * (uint32_t)(band_lo + lane);
            const uint32_t lim4 = 4u * (uint32_t)band_hi;
            for (int it0 = band_lo; it0 < band_hi; it0 += 64, it4 += 256) {
                // lanes past the end compute on rows below the detection rectangle (inside the LDS) and are masked out
                const int y = (int)(mulu24_s(it4, magic) >> 20);          // detection row; sub-image row y + 3
                const int ent = madi24_s(y, K1, (int)it4);
                const uint8_t *pb = img8 + madi24(y, -K2, ent);
                const uint32_t up = *reinterpret_cast<const uint32_t *>(pb + 8),
                               c0 = *reinterpret_cast<const uint32_t *>(pb + 3 * SB + 4),
                               c1 = *reinterpret_cast<const uint32_t *>(pb + 3 * SB + 8),
                               c2 = *reinterpret_cast<const uint32_t *>(pb + 3 * SB + 12),
                               dn = *reinterpret_cast<const uint32_t *>(pb + 6 * SB + 8);
                const uint32_t e4 = __builtin_amdgcn_alignbyte(c2, c1, 3);    // ring pixel 4  (x+3)
                const uint32_t e12 = __builtin_amdgcn_alignbyte(c1, c0, 1);   // ring pixel 12 (x-3)
                // a bright arc of 9 holds p0 or p8 and p4 or p12, all brighter than v + t; a dark arc the mirror image:
                // candidate <=> max(A - v, v - B) > t with A = min(max(p0, p8), max(p4, p12)), B = max(min, min).  Each
                // comparison writes its lane mask directly (v_cmp -> SGPR pair); masks combine on the scalar unit.
                unsigned long long m[4];
#pragma unroll
                for (int h = 0; h < 2; ++h) {   // h = 0: pixels 1, 3 (high bytes, low byte is noise); h = 1: pixels 0, 2 (exact)
                    const uint32_t mk = h ? 0x00ff00ffu : 0xffffffffu;
                    const u16x2_v v = as_u16x2(c1 & mk), q0 = as_u16x2(dn & mk), q8 = as_u16x2(up & mk),
                                  q4 = as_u16x2(e4 & mk), q12 = as_u16x2(e12 & mk);
                    const u16x2_v A = __builtin_elementwise_min(__builtin_elementwise_max(q0, q8), __builtin_elementwise_max(q4, q12));
                    const u16x2_v B = __builtin_elementwise_max(__builtin_elementwise_min(q0, q8), __builtin_elementwise_min(q4, q12));
                    const uint32_t M = as_u32(__builtin_elementwise_max(__builtin_elementwise_sub_sat(A, v), __builtin_elementwise_sub_sat(v, B)));
                    m[1 - h] = __builtin_amdgcn_ballot_w64((unsigned short)M > (unsigned short)(h ? Te_lo : Tn_lo));
                    m[3 - h] = __builtin_amdgcn_ballot_w64(M > (h ? Te_hi : Tn_hi));
                }
                const unsigned long long mlive = __builtin_amdgcn_ballot_w64(it4 < lim4),
                                         mlast = __builtin_amdgcn_ballot_w64((uint8_t)ent == (uint8_t)last4);
#pragma unroll
                for (int j = 0; j < 4; ++j) m[j] &= j ? mlive & ~(en[j] & mlast) : mlive;      // pixel 0 of a group is always inside
                // ordered append: lane-major, then pixel = row-major (y, x), the order the reference emits keypoints in; every
                // later compaction is stable, so the final list needs no sorting
                int pos = __builtin_amdgcn_mbcnt_hi((unsigned)(m[0] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m[0], (unsigned)nsurv));
#pragma unroll
                for (int j = 1; j < 4; ++j)
                    pos = __builtin_amdgcn_mbcnt_hi((unsigned)(m[j] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m[j], (unsigned)pos));
                nsurv += __popcll(m[0]) + __popcll(m[1]) + __popcll(m[2]) + __popcll(m[3]);
                lds_u16 *dst = slist + pos;
                uint32_t ev = (uint32_t)ent;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (__builtin_amdgcn_inverse_ballot_w64(m[j])) {
                        *dst = (unsigned short)ev;
                        asm("v_add_u32 %0, 2, %0" : "+v"(dst));     // ++dst, in place (the compiler adds into a copy)
                    }
                    ++ev;
                }
            }
            __syncthreads();
            FAST_STAMP(2);   // dense pre-test + compaction
            if (nsurv > P.list_cap) {   // first round only: a band cannot overflow
                banded = true; band_hi = band_items;
                continue;
            }

            // ---- score of the survivors; corners at this threshold -> score map (+ slist, in place, unless banded) ----
            for (int i0 = 0; i0 < nsurv; i0 += 64) {
                const int i = i0 + lane;
                const bool act = i < nsurv;
                const unsigned e = slist[min(i, nsurv - 1)];
                const int y = e >> 8, colr = e & 255;
                const uint8_t *w = img8 + (madi24(y, SB, colr) + 5);   // top-left pixel of the 7x7 window: LDS col (8 + colr) - 3
                const int v = w[3 * SB + 3];
                int pr[16];
                pr[0] = w[6 * SB + 3];  pr[1] = w[6 * SB + 4];  pr[2] = w[5 * SB + 5];  pr[3] = w[4 * SB + 6];
                pr[4] = w[3 * SB + 6];  pr[5] = w[2 * SB + 6];  pr[6] = w[1 * SB + 5];  pr[7] = w[4];
                pr[8] = w[3];           pr[9] = w[2];           pr[10] = w[1 * SB + 1]; pr[11] = w[2 * SB];
                pr[12] = w[3 * SB];     pr[13] = w[4 * SB];     pr[14] = w[5 * SB + 1]; pr[15] = w[6 * SB + 2];
                const int Ab = min(max(pr[0], pr[8]), max(pr[4], pr[12])), Bd = max(min(pr[0], pr[8]), min(pr[4], pr[12]));
                const bool brc = Ab > v + tv, dkc = Bd < v - tv;
                // not a bright candidate: a bright arc is impossible (<= t), so the dark network alone gives S (or rejects).
                // The min network runs on x = p (bright) or ~p = -p - 1 (dark): min over an arc commutes with the constant
                // shift by v, so S = max_k min_arc(x) + (bright ? ~v : v) -- one xor per ring pixel instead of a multiply-add
                const int s = brc ? 0 : -1;
                int d[16];
#pragma unroll
                for (int kx = 0; kx < 16; ++kx) d[kx] = pr[kx] ^ s;
                int m3v[16];
#pragma unroll
                for (int kx = 0; kx < 16; ++kx) m3v[kx] = min3i(d[kx], d[(kx + 1) & 15], d[(kx + 2) & 15]);
                int best = -512;
#pragma unroll
                for (int kx = 0; kx < 16; kx += 2) {
                    const int q0 = min3i(m3v[kx], m3v[(kx + 3) & 15], m3v[(kx + 6) & 15]);
                    const int q1 = min3i(m3v[kx + 1], m3v[(kx + 4) & 15], m3v[(kx + 7) & 15]);
                    best = max3i(best, q0, q1);
                }
                int sc = best + ~(v ^ s);                 // cornerScore; corner at t <=> S >= t
                if (__any(brc & dkc)) {   // both polarities pass the compass test: evaluate the dark one as well (d = p there)
                    int M3v[16];
#pragma unroll
                    for (int kx = 0; kx < 16; ++kx) M3v[kx] = max3i(d[kx], d[(kx + 1) & 15], d[(kx + 2) & 15]);
                    int worst = 512;
#pragma unroll
                    for (int kx = 0; kx < 16; kx += 2) {
                        const int q0 = max3i(M3v[kx], M3v[(kx + 3) & 15], M3v[(kx + 6) & 15]);
                        const int q1 = max3i(M3v[kx + 1], M3v[(kx + 4) & 15], M3v[(kx + 7) & 15]);
                        worst = min3i(worst, q0, q1);
                    }
                    if (brc & dkc) sc = max(sc, v - worst - 1);
                }
                const unsigned long long m = __builtin_amdgcn_ballot_w64(sc >= tmin) & __builtin_amdgcn_ballot_w64(act);
                if (__builtin_amdgcn_inverse_ballot_w64(m)) {
                    score8[madi24(y, SS, colr) + (SS + 1)] = (uint8_t)sc;         // score column = detection x + 1, row y + 1
                    if (!banded) slist[ncorn + lane_prefix(m)] = (unsigned short)e;   // write index <= read index: in place is safe
                }
                ncorn += __popcll(m);
            }
            __syncthreads();
            FAST_STAMP(3);   // score network
            if (!banded) break;
            band_lo = band_hi;
            if (band_lo >= nwork) break;
            band_hi = min(nwork, band_lo + band_items);
        }

        // ---- NMS over the corner list (banded rounds: over every pixel of the score map) + emission, row-major already ----
        int nfin = 0;
        const int ncand = banded ? dw * dh : ncorn;
        for (int i0 = 0; i0 < ncand; i0 += 64) {
            const int i = i0 + lane, ic = min(i, ncand - 1);
            int y, colr;
            if (banded) { y = ic / dw; colr = ic - y * dw; }
            else { const unsigned e = slist[ic]; y = e >> 8; colr = e & 255; }
            const uint8_t *s = score8 + madi24(y, SS, colr);           // top-left neighbour
            const int v = s[SS + 1];
            const int nb = max3i(max3i(s[0], s[1], s[2]), max3i(s[SS], s[SS + 2], s[2 * SS]), max((int)s[2 * SS + 1], (int)s[2 * SS + 2]));
            const unsigned long long m = __builtin_amdgcn_ballot_w64(i < ncand) & __builtin_amdgcn_ballot_w64(v > nb);
            const int pos = nfin + lane_prefix(m);
            if (__builtin_amdgcn_inverse_ballot_w64(m & __builtin_amdgcn_ballot_w64(pos < P.slot_cap)))
                out[pos] = (uint32_t)(colr + kpx) | ((uint32_t)(y + kpy) << 12) | ((uint32_t)v << 24);
            nfin += __popcll(m);
        }
        total = nfin;
        FAST_STAMP(4);   // NMS + emission
        if (nfin != 0) break;       // vKeysCell.empty() -> FAST(minThFAST); the score map keeps its valid entries
        __syncthreads();
    }
    if (lane == 0) cell_cnt[out_cell] = min(total, P.slot_cap);
#ifdef ORBHIP_DEVTOOLS
    if constexpr (STAMPS) {
        if (lane == 0 && (cell & 63) == 0) {   // one wave in 64 reports (78,080 waves on six addresses would serialise)
#pragma unroll
            for (int i = 0; i < 5; ++i) atomicAdd(&g_fast_stamps[i], tacc[i]);
            atomicAdd(&g_fast_stamps[7], 1ull);
        }
    }
#endif
#undef FAST_STAMP
}

// ---------------------------------------------------------------------------
// K4: DistributeOctTree (ORBextractor.cc:539-763), one workgroup per (level, frame).
// Parallel restatement: keys never move, each carries the index of
// the node that owns it; a pass counts the four children of every expandable node
// with LDS atomics, decides which nodes split (phase 1: all; phase 2: by descending
// key count, newer node first, until the list reaches N), rebuilds the node table in
// list order (new children reversed in front, survivors behind) and relabels keys.
// ---------------------------------------------------------------------------
constexpr int kOctU = 4;            // independent keys per thread in the key loops
// The candidate keys and their owning nodes live in the HBM workspace (L2-resident while a workgroup works on them), not
// in LDS: with them the workgroup needed 70 KB and two fitted a CU; at 34 KB four do, which is what hides the latencies
// of this kernel -- 170 frames alone 98 -> 84 us, three pipelines 278.8 -> 283.4 k frames/s; a single frame and a
// 64-frame batch measure the same either way.
template <int MAXN>
struct OctShared {
    short x0[2][MAXN], x1[2][MAXN], y0[2][MAXN], y1[2][MAXN];
    int cnt[2][MAXN];
    int ccnt[2][MAXN * 4];     // child counters of a pass (then: new index of every created child); double-buffered so that
                               // the next pass's buffer is cleared while the keys are still being relabelled from this one
    int nmap[MAXN];            // survivors: new index; split nodes: 0x40000000
    unsigned short ord[MAXN];  // processing order -> node
    unsigned short rnk[MAXN];  // node -> processing rank (0xffff: not expandable)
    int cincl[MAXN];           // inclusive scan of child counts in processing order
    int scan[16];
    int vars[8];
};

template <int MAXN, int T>
__device__ __forceinline__ void octree_body(OctShared<MAXN> &S, const PyrGeom &G, const LevelGeom &L,
                                            const int level, const int b, const int K,
                                            const int *__restrict__ ccnt_in, const uint32_t *__restrict__ ckp_in,
                                            uint32_t *__restrict__ gkeys, unsigned short *__restrict__ gnode,
                                            uint32_t *__restrict__ sel_kp, int *__restrict__ sel_cnt,
                                            int *__restrict__ frame_status)
{
#define keys(k) gkeys[k]        // candidate keys (x | y<<12 | score<<24)
#define knode(k) gnode[k]      // owning node of each key
    const int tid = threadIdx.x;
    const int N = L.quota;
    // ---- initial nodes (:543-585); their key counters live in counter buffer 1 (buffer 0 holds the cell offsets) ----
    const int nIni = L.nIni;
    const int height = L.maxBY - 16;
    int *RC = S.ccnt[1];
    for (int i = tid; i < nIni; i += T) {
        S.x0[0][i] = (short)(int)__fmul_rn(L.hX, (float)i);
        S.x1[0][i] = (short)(int)__fmul_rn(L.hX, (float)(i + 1));
        S.y0[0][i] = 0;
        S.y1[0][i] = (short)height;
        RC[i] = 0;
    }
    __syncthreads();   // also publishes the cell offsets the caller wrote
    // gather the per-cell survivor lists into one array in reference order (flat loop over the K keys, owning cell by
    // binary search over the exclusive offsets, loads stay independent) and, in the same pass, file every key under
    // its initial node; with a handful of initial nodes all keys would hammer the same few LDS counters, so a
    // wavefront adds its keys up with ballots first
    for (int k0 = 0; k0 < K; k0 += T * 4) {
        uint32_t v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = k0 + u * T + tid;
            v[u] = 0;
            if (k < K) {
                int lo = 0, hi = L.ncells - 1;      // last cell c with offset[c] <= k
                while (lo < hi) {
                    const int mid = (lo + hi + 1) >> 1;
                    if (S.ccnt[0][mid] <= k) lo = mid; else hi = mid - 1;
                }
                v[u] = ckp_in[(size_t)lo * G.slot_cap + (k - S.ccnt[0][lo])];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = k0 + u * T + tid;
            const bool ok = k < K;
            const int bin = min((int)__fdiv_rn((float)(v[u] & 0xfffu), L.hX), nIni - 1);
            if (ok) { keys(k) = v[u]; knode(k) = (unsigned short)bin; }
            if (nIni <= 4) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned long long mk = __ballot(ok && bin == j);
                    if ((tid & 63) == 0 && mk) atomicAdd(&RC[j], __popcll(mk));
                }
            } else if (ok) {
                atomicAdd(&RC[bin], 1);
            }
        }
    }
    __syncthreads();
    // one wavefront drops the empty initial nodes (order kept) and clears the first pass's child counters
    if (tid < 64) {
        const int lane = tid;
        const int per = (nIni + 63) >> 6;
        const int i_lo = lane * per, i_hi = min(i_lo + per, nIni);
        int ne = 0;
        for (int i = i_lo; i < i_hi; ++i) ne += RC[i] > 0;
        int incl = ne;
        incl = wave_incl_scan_add(incl);
        const int n0 = __builtin_amdgcn_readlane(incl, 63);
        int pos = incl - ne;
        for (int i = i_lo; i < i_hi; ++i) {
            const int c = RC[i];
            S.nmap[i] = pos;
            if (c > 0) {
                S.x0[1][pos] = S.x0[0][i]; S.x1[1][pos] = S.x1[0][i];
                S.y0[1][pos] = S.y0[0][i]; S.y1[1][pos] = S.y1[0][i];
                S.cnt[1][pos] = c;
                ++pos;
            }
        }
        if (lane == 0) S.vars[0] = n0;
        for (int i = lane; i < n0 * 4; i += 64) S.ccnt[0][i] = 0;
    }
    __syncthreads();
    int n = S.vars[0];  // list size
    if (n != nIni) {    // some initial node was empty: the keys' node labels shift
        for (int k0 = 0; k0 < K; k0 += T * kOctU) {
            int nd[kOctU];
#pragma unroll
            for (int u = 0; u < kOctU; ++u) { const int k = k0 + u * T + tid; nd[u] = k < K ? (int)knode(k) : 0; }
#pragma unroll
            for (int u = 0; u < kOctU; ++u) nd[u] = S.nmap[nd[u]];
#pragma unroll
            for (int u = 0; u < kOctU; ++u) { const int k = k0 + u * T + tid; if (k < K) knode(k) = (unsigned short)nd[u]; }
        }
    }
    int cur = 1;
    int phase = 1;
    bool finish = (K == 0);
    int cb = 0;     // child-counter buffer of this pass

    // A pass = [all wavefronts] count the children of every expandable node over the keys -> barrier ->
    // [ONE wavefront] the node-table work: which nodes split, the new table in list order, where every created child and
    // every survivor lands -> barrier -> [all wavefronts] relabel the keys.  The node table has a few hundred entries
    // at most; doing its scans with the whole workgroup cost 15 barriers and 8 wavefronts' worth of instructions per
    // pass on a CU whose issue slots the workgroup shares with another level's, so one wavefront does it with
    // wave-level scans (a lane owns `per` consecutive nodes) while the others wait at the barrier.
    while (!finish) {
        const int prevSize = n;
        int *CC = S.ccnt[cb];
        // B: count children of expandable nodes (DivideNode :481-526)
        for (int k0 = 0; k0 < K; k0 += T * kOctU) {
            // kOctU independent keys per thread: the LDS round trips of the chains overlap
            int nd[kOctU], cn[kOctU];
            uint32_t kv[kOctU];
#pragma unroll
            for (int u = 0; u < kOctU; ++u) {
                const int k = k0 + u * T + tid;
                nd[u] = k < K ? (int)knode(k) : 0;
                kv[u] = k < K ? keys(k) : 0u;
            }
#pragma unroll
            for (int u = 0; u < kOctU; ++u) cn[u] = S.cnt[cur][nd[u]];
            int bx0[kOctU], bx1[kOctU], by0[kOctU], by1[kOctU];
#pragma unroll
            for (int u = 0; u < kOctU; ++u) {
                bx0[u] = S.x0[cur][nd[u]]; bx1[u] = S.x1[cur][nd[u]];
                by0[u] = S.y0[cur][nd[u]]; by1[u] = S.y1[cur][nd[u]];
            }
#pragma unroll
            for (int u = 0; u < kOctU; ++u) {
                const int k = k0 + u * T + tid;
                if (k < K && cn[u] > 1) {
                    const int x = kv[u] & 0xfff, y = (kv[u] >> 12) & 0xfff;
                    const int mx = bx0[u] + ((bx1[u] - bx0[u] + 1) >> 1), my = by0[u] + ((by1[u] - by0[u] + 1) >> 1);
                    atomicAdd(&CC[4 * nd[u] + (x < mx ? 0 : 1) + (y < my ? 0 : 2)], 1);
                }
            }
        }
        __syncthreads();
        if (phase == 2) {
            // processing order of the careful phase: (size desc, list position asc) -- the reference sorts (size, node
            // address) ascending and walks from the back (:684-685); "newer node first" stands in for the address.
            // O(n^2) comparisons, spread over the whole workgroup (once per level).
            for (int i = tid; i < n; i += T) {
                const int ci = S.cnt[cur][i];
                if (ci > 1) {
                    int r = 0;
#pragma unroll 8
                    for (int j = 0; j < n; ++j) {
                        const int cj = S.cnt[cur][j];
                        r += (cj > 1) && (cj > ci || (cj == ci && j < i));
                    }
                    S.rnk[i] = (unsigned short)r;
                    S.ord[r] = (unsigned short)i;
                } else {
                    S.rnk[i] = 0xffff;
                }
            }
            __syncthreads();
        }
        const int nxt = cur ^ 1;
        if (tid < 64) {
            const int lane = tid;
            const int per = (n + 63) >> 6;                 // consecutive nodes (or ranks) per lane
            const int i_lo = lane * per, i_hi = min(i_lo + per, n);
            // C: processing rank of every expandable node (phase 1: list order)
            int m;
            {
                int e_cnt = 0;
                for (int i = i_lo; i < i_hi; ++i) e_cnt += S.cnt[cur][i] > 1;
                int incl = e_cnt;
                incl = wave_incl_scan_add(incl);
                m = __builtin_amdgcn_readlane(incl, 63);
                if (phase == 1) {
                    int r = incl - e_cnt;
                    for (int i = i_lo; i < i_hi; ++i) {
                        if (S.cnt[cur][i] > 1) { S.rnk[i] = (unsigned short)r; S.ord[r] = (unsigned short)i; ++r; }
                        else S.rnk[i] = 0xffff;
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            // D: inclusive scan of the child counts in processing order; the break index J of the careful phase
            const int perm = (m + 63) >> 6;
            const int j_lo = lane * perm, j_hi = min(j_lo + perm, m);
            int J, Gc;
            {
                int sum = 0;
                for (int j = j_lo; j < j_hi; ++j) {
                    const int nd = S.ord[j];
                    sum += (CC[4 * nd] > 0) + (CC[4 * nd + 1] > 0) + (CC[4 * nd + 2] > 0) + (CC[4 * nd + 3] > 0);
                }
                int incl = sum;
                incl = wave_incl_scan_add(incl);
                int run = incl - sum, jf = 0x7fffffff;
                for (int j = j_lo; j < j_hi; ++j) {
                    const int nd = S.ord[j];
                    run += (CC[4 * nd] > 0) + (CC[4 * nd + 1] > 0) + (CC[4 * nd + 2] > 0) + (CC[4 * nd + 3] > 0);
                    S.cincl[j] = run;
                    if (phase == 2 && jf == 0x7fffffff && n + run - (j + 1) >= N) jf = j;
                }
                jf = wave_min(jf);
                J = jf == 0x7fffffff ? m - 1 : jf;
            }
            __builtin_amdgcn_wave_barrier();
            Gc = m > 0 ? S.cincl[J] : 0;   // children created this pass
            // E: the new table: children reversed in front, survivors behind in order
            int nToExpand = 0;
            {
                int surv_cnt = 0;
                for (int i = i_lo; i < i_hi; ++i) { const int r = S.rnk[i]; surv_cnt += !(r != 0xffff && r <= J); }
                int incl = surv_cnt;
                incl = wave_incl_scan_add(incl);
                int spos = incl - surv_cnt;
                for (int i = i_lo; i < i_hi; ++i) {
                    const int r = S.rnk[i];
                    const bool split = r != 0xffff && r <= J;
                    if (!split) {
                        const int ni = Gc + spos++;
                        S.nmap[i] = ni;
                        S.x0[nxt][ni] = S.x0[cur][i]; S.x1[nxt][ni] = S.x1[cur][i];
                        S.y0[nxt][ni] = S.y0[cur][i]; S.y1[nxt][ni] = S.y1[cur][i];
                        S.cnt[nxt][ni] = S.cnt[cur][i];
                    } else {
                        S.nmap[i] = 0x40000000;
                        const int px0 = S.x0[cur][i], px1 = S.x1[cur][i], py0 = S.y0[cur][i], py1 = S.y1[cur][i];
                        const int mx = px0 + ((px1 - px0 + 1) >> 1), my = py0 + ((py1 - py0 + 1) >> 1);
                        int g = r > 0 ? S.cincl[r - 1] : 0;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int c = CC[4 * i + q];
                            if (c > 0) {
                                const int ni = Gc - 1 - g;
                                S.x0[nxt][ni] = (short)((q & 1) ? mx : px0);
                                S.x1[nxt][ni] = (short)((q & 1) ? px1 : mx);
                                S.y0[nxt][ni] = (short)((q & 2) ? my : py0);
                                S.y1[nxt][ni] = (short)((q & 2) ? py1 : my);
                                S.cnt[nxt][ni] = c;
                                CC[4 * i + q] = ni;          // the keys of this quadrant move to node ni
                                nToExpand += (c > 1);
                                ++g;
                            }
                        }
                    }
                }
            }
            // G: bookkeeping (:669-673, :734)
            nToExpand = wave_sum(nToExpand);
            const int n_new = Gc + (n - (m > 0 ? J + 1 : 0));
            int fin = 0, ph = phase;
            if (n_new >= N || n_new == prevSize) fin = 1;
            else if (phase == 1 && n_new + nToExpand * 3 > N) ph = 2;
            if (lane == 0) { S.vars[0] = n_new; S.vars[1] = fin; S.vars[2] = ph; }
            // the next pass's counters
            int *CN = S.ccnt[cb ^ 1];
            for (int i = lane; i < n_new * 4; i += 64) CN[i] = 0;
        }
        __syncthreads();
        // F: relabel keys
        for (int k0 = 0; k0 < K; k0 += T * kOctU) {
            int nd[kOctU], mp[kOctU];
            uint32_t kv[kOctU];
#pragma unroll
            for (int u = 0; u < kOctU; ++u) {
                const int k = k0 + u * T + tid;
                nd[u] = k < K ? (int)knode(k) : 0;
                kv[u] = k < K ? keys(k) : 0u;
            }
#pragma unroll
            for (int u = 0; u < kOctU; ++u) mp[u] = S.nmap[nd[u]];
            int bx0[kOctU], bx1[kOctU], by0[kOctU], by1[kOctU];
#pragma unroll
            for (int u = 0; u < kOctU; ++u) {
                bx0[u] = S.x0[cur][nd[u]]; bx1[u] = S.x1[cur][nd[u]];
                by0[u] = S.y0[cur][nd[u]]; by1[u] = S.y1[cur][nd[u]];
            }
#pragma unroll
            for (int u = 0; u < kOctU; ++u) {
                const int k = k0 + u * T + tid;
                if (k >= K) continue;
                if (mp[u] & 0x40000000) {
                    const int x = kv[u] & 0xfff, y = (kv[u] >> 12) & 0xfff;
                    const int mx = bx0[u] + ((bx1[u] - bx0[u] + 1) >> 1), my = by0[u] + ((by1[u] - by0[u] + 1) >> 1);
                    knode(k) = (unsigned short)CC[4 * nd[u] + (x < mx ? 0 : 1) + (y < my ? 0 : 2)];
                } else {
                    knode(k) = (unsigned short)mp[u];
                }
            }
        }
        n = S.vars[0];
        finish = S.vars[1] != 0;
        phase = S.vars[2];
        cur = nxt;
        cb ^= 1;
        // no barrier here: the next pass's counting reads only what the node-table wavefront published before the barrier
        // above and the node labels this thread wrote itself; S.vars is rewritten after the next barrier
    }
    __syncthreads();

    // ---- retain the best key of every node, first index wins ties (:742-760) ----
    unsigned int *best = reinterpret_cast<unsigned int *>(S.ccnt[0]);
    for (int i = tid; i < n; i += T) best[i] = 0;
    __syncthreads();
    for (int k0 = 0; k0 < K; k0 += T * kOctU) {
        int nd[kOctU];
        uint32_t kv[kOctU];
#pragma unroll
        for (int u = 0; u < kOctU; ++u) {
            const int k = k0 + u * T + tid;
            nd[u] = k < K ? (int)knode(k) : 0;
            kv[u] = k < K ? keys(k) : 0u;
        }
#pragma unroll
        for (int u = 0; u < kOctU; ++u) {
            const int k = k0 + u * T + tid;
            if (k < K) atomicMax(&best[nd[u]], ((kv[u] >> 24) << 24) | (0xffffffu - (uint32_t)k));
        }
    }
    __syncthreads();
    uint32_t *out = sel_kp + (size_t)b * G.kp_cap_total + L.kp_base;
    const int nout = min(n, L.kp_cap);
    for (int i = tid; i < nout; i += T) out[i] = keys(0xffffffu - (best[i] & 0xffffffu));
    if (tid == 0) {
        sel_cnt[b * ORBHIP_MAX_LEVELS + level] = nout;
        if (n > L.kp_cap) atomicExch(&frame_status[b], ORBHIP_E_CAPACITY);
    }
#undef keys
#undef knode
}

template <int MAXN, int T>
__global__ __launch_bounds__(T) void k_octree(PyrGeom G, const int *__restrict__ cell_cnt,
                                                const uint32_t *__restrict__ cell_kp,
                                                uint32_t *__restrict__ keys_ws,
                                                unsigned short *__restrict__ node_ws,
                                                uint32_t *__restrict__ sel_kp,
                                                int *__restrict__ sel_cnt, int *__restrict__ frame_status)
{
    __shared__ OctShared<MAXN> S;
    int level, b;
    xcd_remap(level, b);
    const int tid = threadIdx.x;
    const LevelGeom L = G.lv[level];
    uint32_t *gkeys = keys_ws + (size_t)b * G.cand_cap_total + L.cand_base;
    unsigned short *gnode = node_ws + (size_t)b * G.cand_cap_total + L.cand_base;
    const int *ccnt_in = cell_cnt + (size_t)b * G.ncells_total + L.cell_base;
    const uint32_t *ckp_in = cell_kp + ((size_t)b * G.ncells_total + L.cell_base) * G.slot_cap;
    int tot;
    // ---- offsets of the per-cell survivor lists inside one array in reference order ----
    int K = 0;
    for (int c0 = 0; c0 < L.ncells; c0 += T) {
        int c = c0 + tid;
        int n = c < L.ncells ? ccnt_in[c] : 0;
        int base = block_excl_scan<T>(n, S.scan, &tot);
        if (c < L.ncells) S.ccnt[0][c] = K + base;  // ncells <= MAXN*4 checked on the host
        K += tot;
    }
    octree_body<MAXN, T>(S, G, L, level, b, K, ccnt_in, ckp_in, gkeys, gnode, sel_kp, sel_cnt, frame_status);
}

// ---------------------------------------------------------------------------
// K6: GaussianBlur(7x7, sigma 2, REFLECT_101) per level (:1085-1086).  Separable integer
// kernel through LDS.  A 64x58 output tile stages 64 rows x 72 bytes of the padded plane with
// aligned dword loads (the 19-px REFLECT_101 frame supplies the border).  Row pass: one thread
// = 4 pixels x 2 rows, two v_dot4_u32_u8 per pixel on byte-aligned windows; the uint16 row sums
// (<= 255*257) of rows 2r,2r+1 are stored interleaved in one dword so that the column pass is
// three v_dot2_u32_u16 + one multiply per pixel; (sum + 2^15) >> 16, saturated; dword stores.
// ---------------------------------------------------------------------------
constexpr int kBIn = kBlurTH + 6;   // staged input rows (64)
constexpr int kBInW = 18;           // staged dwords per row: tile bytes x0-4 .. x0+67

__device__ __forceinline__ uint32_t mad24u(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// two int16 lanes -> two saturated bytes in the low half
__device__ __forceinline__ uint32_t sat_pk_u8(uint32_t v)
{
    uint32_t r;
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(r) : "v"(v));
    return r;
}

__global__ __launch_bounds__(256) void k_blur(const uint8_t *__restrict__ pyr, uint8_t *__restrict__ blur,
                                              PyrGeom G, const TileDesc *__restrict__ tiles, BlurW W)
{
    __shared__ uint32_t sin[kBIn * kBInW];
    __shared__ uint4 srow[(kBIn / 2) * (kBlurTW / 4)];   // [pair-row][group]: 4 px x (row 2r | row 2r+1 << 16)
    int tile, fr;
    xcd_remap(tile, fr);
    const TileDesc t = tiles[tile];
    const LevelGeom L = G.lv[t.level];
    const size_t fo = (size_t)fr * G.frame_bytes + L.plane_off + (size_t)kEdge * L.pitch + kPadL;
    const uint8_t *roi = pyr + fo;
    uint8_t *out = blur + fo;
    const int x0 = t.tx * kBlurTW, y0 = t.ty * kBlurTH;
    const int rows = min(kBlurTH, L.h - y0);
    const int nin = rows + 6, npair = (nin + 1) >> 1;
    const int tid = threadIdx.x;
    const int gxmax = (L.w + 12) & ~3;
    {   // issue all staging loads of this thread before the first LDS store
        constexpr int U = (kBIn * kBInW + 255) / 256;   // 5
        uint32_t v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = min(u * 256 + tid, nin * kBInW - 1);
            const int r = (i * 3641) >> 16, c = i - r * kBInW;   // i / 18 for i < 1152
            const int gy = min(y0 - 3 + r, L.h + kEdge - 1), gx = min(x0 - 4 + 4 * c, gxmax);
            v[u] = *reinterpret_cast<const uint32_t *>(roi + (ptrdiff_t)(__mul24(gy, L.pitch) + gx));   // 24-bit operands: full rate
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = u * 256 + tid;
            if (i < nin * kBInW) sin[i] = v[u];
        }
    }
    // row pass: item = (pair-row, 4-px group).  Output j of a group needs tile bytes 4g+1+j .. 4g+7+j, i.e. a window that
    // starts 1 + j bytes into three consecutive dwords: instead of aligning the data, the taps are laid against the
    // dwords (ten wave-uniform weight words), 10 v_dot4_u32_u8 per 4 pixels
    const uint32_t B0 = W.w[0], B1 = W.w[1], B2 = W.w[2], B3 = W.w[3], B4 = W.w[4], B5 = W.w[5], B6 = W.w[6];
    const uint32_t k0a = (B0 << 8) | (B1 << 16) | (B2 << 24), k0b = B3 | (B4 << 8) | (B5 << 16) | (B6 << 24);
    const uint32_t k1a = (B0 << 16) | (B1 << 24), k1b = B2 | (B3 << 8) | (B4 << 16) | (B5 << 24), k1c = B6;
    const uint32_t k2a = B0 << 24, k2b = B1 | (B2 << 8) | (B3 << 16) | (B4 << 24), k2c = B5 | (B6 << 8);
    const uint32_t k3b = B0 | (B1 << 8) | (B2 << 16) | (B3 << 24), k3c = B4 | (B5 << 8) | (B6 << 16);
    __syncthreads();
    for (int it = tid; it < npair * 16; it += 256) {
        const int pr = it >> 4, g = it & 15;
        uint32_t s[2][4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t *p = &sin[madi24(min(2 * pr + h, nin - 1), kBInW, g)];
            const uint32_t d0 = p[0], d1 = p[1], d2 = p[2];
            s[h][0] = __builtin_amdgcn_udot4(d0, k0a, __builtin_amdgcn_udot4(d1, k0b, 0, false), false);
            s[h][1] = __builtin_amdgcn_udot4(d0, k1a, __builtin_amdgcn_udot4(d1, k1b, __builtin_amdgcn_udot4(d2, k1c, 0, false), false), false);
            s[h][2] = __builtin_amdgcn_udot4(d0, k2a, __builtin_amdgcn_udot4(d1, k2b, __builtin_amdgcn_udot4(d2, k2c, 0, false), false), false);
            s[h][3] = __builtin_amdgcn_udot4(d1, k3b, __builtin_amdgcn_udot4(d2, k3c, 0, false), false);
        }
        // both sums fit uint16 (<= 255 * 257): one v_perm_b32 interleaves the two rows
        srow[it] = make_uint4(__builtin_amdgcn_perm(s[1][0], s[0][0], 0x05040100u), __builtin_amdgcn_perm(s[1][1], s[0][1], 0x05040100u),
                              __builtin_amdgcn_perm(s[1][2], s[0][2], 0x05040100u), __builtin_amdgcn_perm(s[1][3], s[0][3], 0x05040100u));
    }
    __syncthreads();
    // column pass: item = (output row pair, group).  Rows 2p and 2p+1 read the same four staged pair-rows: the even
    // row pairs the taps (w0,w1)(w2,w3)(w4,w5)+w6 with them, the odd row w0+(w1,w2)(w3,w4)(w5,w6).
    const uint32_t w01 = (uint32_t)W.w[0] | ((uint32_t)W.w[1] << 16), w23 = (uint32_t)W.w[2] | ((uint32_t)W.w[3] << 16),
                   w45 = (uint32_t)W.w[4] | ((uint32_t)W.w[5] << 16), w12 = (uint32_t)W.w[1] | ((uint32_t)W.w[2] << 16),
                   w34 = (uint32_t)W.w[3] | ((uint32_t)W.w[4] << 16), w56 = (uint32_t)W.w[5] | ((uint32_t)W.w[6] << 16),
                   w6l = (uint32_t)W.w[6], w0h = (uint32_t)W.w[0] << 16;   // single taps: the other uint16 lane gets weight 0
    const int nprow = (rows + 1) >> 1;
    for (int it = tid; it < nprow * 16; it += 256) {
        const int p = it >> 4, g = it & 15;
        if (x0 + 4 * g >= L.w) continue;
        const uint4 P0 = srow[p * 16 + g], P1 = srow[(p + 1) * 16 + g], P2 = srow[(p + 2) * 16 + g], P3 = srow[(p + 3) * 16 + g];
        const uint32_t a0[4] = {P0.x, P0.y, P0.z, P0.w}, a1[4] = {P1.x, P1.y, P1.z, P1.w},
                       a2[4] = {P2.x, P2.y, P2.z, P2.w}, a3[4] = {P3.x, P3.y, P3.z, P3.w};
        // acc + 2^15 (the rounding term rides in the multiply-add); the result is its upper half, <= 257 (the taps sum
        // to 257): v_perm_b32 packs two upper halves into one dword and v_sat_pk_u8_i16 saturates both to bytes
        uint32_t acce[4], acco[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acce[j] = udot2(a0[j], w01, udot2(a1[j], w23, udot2(a2[j], w45, udot2(a3[j], w6l, 32768u))));
            acco[j] = udot2(a1[j], w12, udot2(a2[j], w34, udot2(a3[j], w56, udot2(a0[j], w0h, 32768u))));
        }
        const uint32_t rese = sat_pk_u8(__builtin_amdgcn_perm(acce[1], acce[0], 0x07060302u)) |
                              (sat_pk_u8(__builtin_amdgcn_perm(acce[3], acce[2], 0x07060302u)) << 16);
        const uint32_t reso = sat_pk_u8(__builtin_amdgcn_perm(acco[1], acco[0], 0x07060302u)) |
                              (sat_pk_u8(__builtin_amdgcn_perm(acco[3], acco[2], 0x07060302u)) << 16);
        uint8_t *o = out + (uint32_t)(__mul24(y0 + 2 * p, L.pitch) + x0 + 4 * g);
        *reinterpret_cast<uint32_t *>(o) = rese;
        if (2 * p + 1 < rows) *reinterpret_cast<uint32_t *>(o + L.pitch) = reso;
    }
}

// ---------------------------------------------------------------------------
// K5+K6+K7 fused: IC_Angle + GaussianBlur(7x7) of the patch + steered rBRIEF, one wavefront per keypoint.
// The blurred level is only ever read inside the 37 x 37 patch of a keypoint (:108-147), so the wavefront blurs that
// patch itself instead of reading a blurred plane that a separate kernel wrote for the whole level:
//  * ONE staged tile: the 43 x 44-byte window (rows ky-21 .. ky+21, bytes kx-21 .. kx+22) of the unblurred padded
//    plane, byte-unaligned dword loads so that window column 0 is byte 0 of every row; it holds the 31 x 31
//    orientation disc and every tap of the patch's blur (the 19-px REFLECT_101 frame covers the 2 px a keypoint at
//    the minimum border distance reaches outside the level);
//  * moments from the tile as before (12 disc pixels per lane, v_dot2_i32_i16);
//  * row pass: item = (pair of window rows, 4 columns); the taps are laid against the three dwords an output group
//    touches (10 v_dot4_u32_u8 per 4 outputs), the uint16 sums of the two rows go interleaved into one dword;
//  * column pass: item = (pair of patch rows, 4 columns): 4 v_dot2_u32_u16 per output (rounding term in the
//    accumulator), v_perm_b32 + v_sat_pk_u8_i16 pack; the blurred patch overwrites the window tile;
//  * descriptor tests sample the blurred patch in LDS.
// Same arithmetic as k_blur (which remains for orbhip_blurred_level_download): integer, exact.
// ---------------------------------------------------------------------------
constexpr int kWinRows = 43, kWinDw = 11;     // staged window
constexpr int kWinRpi = 5;                    // rows per staging instruction (55 of 64 lanes)
constexpr int kWinLoads = 9;                  // 9 x 5 = 45 >= 43 rows
constexpr int kWinWords = kWinRows * kWinDw + 7;   // + the dwords the last column group reads past the last row
constexpr int kHPairs = 22, kHGroups = 10;    // row-pass results: 22 pair-rows x 10 groups of 4 columns (uint4 each)

#ifndef ORBHIP_DESC_WAVES
#define ORBHIP_DESC_WAVES 7   // resident waves per SIMD the register allocation is held to (the kernel needs 66 VGPRs; LDS allows 7
                              // workgroups per CU).  tools/ab_build.sh, 512 frames as three pipelines whose pyramid stages are
                              // chained by a stage gate (bench.py --gates 0): 5: 254 k, 6: 259 k, 7: 260 k frames/s.  Free-running
                              // pipelines (no gate) preferred 5 (244-248 k against 235-246 k): fewer descriptor waves left the
                              // other pipelines' kernels room
#endif
// Consecutive keypoint slots per wavefront of k_describe_fused (a launch argument).  More slots amortise the wavefront's
// start-up (tables, level search) but leave fewer wavefronts to fill the GPU's 7 x 1024 slots evenly: per 64 frames alone
// 1: 86 us, 2: 64 us, 3: 79 us, 4: 79 us (4.6 rounds of wavefronts at 2, 3.1 / 2.3 at 3 / 4); per 170 frames alone 2, 3
// and 4 measure the same (162-164 us) and three pipelines prefer 4 (282.5 / 286.0 / 286.6 k frames/s).  The launch takes
// the largest count that still leaves kDescMinRounds rounds.
constexpr int kDescWaveSlotsPerCu = 4 * ORBHIP_DESC_WAVES, kDescMinRounds = 6;
static int desc_per_wave(int kp_cap_total, int batch, int num_cus)
{
    const double slots = (double)num_cus * kDescWaveSlotsPerCu;
    for (int p = 4; p > 2; --p)
        if ((double)kp_cap_total * batch / p >= kDescMinRounds * slots) return p;
    return 2;
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(ORBHIP_DESC_WAVES, ORBHIP_DESC_WAVES))) void k_describe_fused(const uint8_t *__restrict__ pyr, PyrGeom G,
                                                        const uint32_t *__restrict__ sel_kp,
                                                        const int *__restrict__ sel_cnt,
                                                        const int *__restrict__ desc_tab,
                                                        const float4 *__restrict__ patternf,
                                                        orbhip_keypoint *__restrict__ out_kp,
                                                        uint8_t *__restrict__ out_desc, int cap,
                                                        int *__restrict__ out_n, int *__restrict__ status, int per_wave, BlurW W,
                                                        const uint32_t *__restrict__ hitem_tab,
                                                        const uint8_t *__restrict__ images, int img_stride, size_t img_frame_stride)
{
    __shared__ uint32_t swin[4][kWinWords];
    __shared__ uint4 shsum[4][kHPairs * kHGroups];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int bx, b;
    xcd_remap(bx, b);
    // orientation disc as aligned dwords of the window tile: the 749 disc pixels sit in 213 (row, 4-column) chunks, four
    // per lane; per chunk the dword index inside the tile and the signed byte weights u (column offset) and v (row offset)
    // of its four pixels, 0 outside the disc
    int cw[4], uw[4], vw[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { cw[k] = desc_tab[k * 64 + lane]; uw[k] = desc_tab[256 + k * 64 + lane]; vw[k] = desc_tab[512 + k * 64 + lane]; }
    float4 pat[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) pat[j] = patternf[j * 64 + lane];
    const int *cnts = sel_cnt + b * ORBHIP_MAX_LEVELS;
    int total = 0;
    for (int l = 0; l < G.nlevels; ++l) total += cnts[l];
    if (bx == 0 && wv == 0 && lane == 0) {
        out_n[b] = min(total, cap);
        if (total > cap && status) atomicExch(&status[b], ORBHIP_E_CAPACITY);
    }
    // tap words of the row pass (output j of a group starts j bytes into dword 0) and of the column pass
    const uint32_t B0 = W.w[0], B1 = W.w[1], B2 = W.w[2], B3 = W.w[3], B4 = W.w[4], B5 = W.w[5], B6 = W.w[6];
    const uint32_t h0a = B0 | (B1 << 8) | (B2 << 16) | (B3 << 24), h0b = B4 | (B5 << 8) | (B6 << 16);
    const uint32_t h1a = (B0 << 8) | (B1 << 16) | (B2 << 24), h1b = B3 | (B4 << 8) | (B5 << 16) | (B6 << 24);
    const uint32_t h2a = (B0 << 16) | (B1 << 24), h2b = B2 | (B3 << 8) | (B4 << 16) | (B5 << 24), h2c = B6;
    const uint32_t h3a = B0 << 24, h3b = B1 | (B2 << 8) | (B3 << 16) | (B4 << 24), h3c = B5 | (B6 << 8);
    const uint32_t w01 = B0 | (B1 << 16), w23 = B2 | (B3 << 16), w45 = B4 | (B5 << 16), w12 = B1 | (B2 << 16),
                   w34 = B3 | (B4 << 16), w56 = B5 | (B6 << 16), w6l = B6, w0h = B0 << 16;
    const int srow = (lane * 373) >> 12, scol = lane - srow * kWinDw;   // lane / 11 for lane < 64; lanes 55..63 idle
    // row-pass items of this lane: only the 189 of the 220 (pair-row, group) items that a descriptor test can reach
    // (rotated pattern offsets stay inside a disc of radius 18.4 + rounding) are computed -- 3 rounds instead of 4
    const uint32_t hitems = hitem_tab[lane];
    uint32_t *win = swin[wv];
    uint4 *hs = shsum[wv];
  int level = 0, before = 0, mine = 0, kp_base = 0, kp_end = -1, pitch = 0;
  for (int kk = 0; kk < per_wave; ++kk) {
    const int slot = (bx * 4 + wv) * per_wave + kk;  // wave-uniform: everything up to the pixel loads is scalar work
    if (slot >= G.kp_cap_total) break;
    // level of the slot, keypoints of the levels before it and of its own.  The slots of a wavefront are consecutive, so
    // the search over the level table (230 scalar instructions and 16 dependent scalar loads through the kernel argument;
    // the scalar pipe of this kernel is as busy as the vector pipe) runs again only when a slot leaves the level of the
    // one before it: 72.4 -> 68.3 us per 64 frames.  Folding the whole table with static indices before the loop (one
    // wait for all loads) measured the same and costs 14 more spilled SGPRs
    if (slot >= kp_end) {
        level = 0;
        for (int l = 1; l < G.nlevels; ++l) if (slot >= G.lv[l].kp_base) level = l;
        before = 0; mine = 0;
        for (int l = 0; l < G.nlevels; ++l) {
            const int c = cnts[l];
            if (l < level) before += c;
            if (l == level) mine = c;
        }
        kp_base = G.lv[level].kp_base; kp_end = kp_base + G.lv[level].kp_cap; pitch = G.lv[level].pitch;
    }
    const int i = slot - kp_base;
    if (i >= mine) continue;
    const int oidx = before + i;
    if (oidx >= cap) continue;

    const uint32_t kv = sel_kp[(size_t)b * G.kp_cap_total + slot];
    const int kx = (int)(kv & 0xfff) + 16, ky = (int)((kv >> 12) & 0xfff) + 16;  // + minBorder (:843-844)
    const int resp = (int)(kv >> 24);
    uint32_t wvv[kWinLoads];
    // level 0 of a handle that does not materialise mvImagePyramid[0]: the window comes from the caller's image; the up
    // to 5 px it overshoots the image near the border are REFLECT_101 by index (what copyMakeBorder would have written)
    const bool from_img = images != nullptr && level == 0;
    const bool inside = kx >= 21 && kx + 22 < G.lv[0].w && ky >= 21 && ky + 21 < G.lv[0].h;
    if (from_img && !inside) {   // wave-uniform: a border keypoint gathers its window byte by byte
        const uint8_t *ib = images + (size_t)b * img_frame_stride;
        uint32_t cx[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) cx[k] = (uint32_t)reflect101(kx - 21 + 4 * scol + k, G.lv[0].w);
#pragma unroll
        for (int u = 0; u < kWinLoads; ++u) {
            wvv[u] = 0u;
            if (lane < kWinRpi * kWinDw && (u < kWinLoads - 1 || srow < kWinRows - (kWinLoads - 1) * kWinRpi)) {
                const uint8_t *rp = ib + (size_t)((uint32_t)reflect101(ky - 21 + srow + u * kWinRpi, G.lv[0].h) * (uint32_t)img_stride);
                wvv[u] = (uint32_t)rp[cx[0]] | ((uint32_t)rp[cx[1]] << 8) | ((uint32_t)rp[cx[2]] << 16) | ((uint32_t)rp[cx[3]] << 24);
            }
        }
    } else {
        const int wpitch = from_img ? img_stride : pitch;
        const uint8_t *wb = from_img
            ? images + (size_t)b * img_frame_stride + (size_t)(ky - 21) * img_stride + (kx - 21)
            : pyr + (size_t)b * G.frame_bytes + G.lv[level].plane_off + (size_t)kEdge * pitch + kPadL + (size_t)(ky - 21) * pitch + (kx - 21);   // window byte (0, 0), uniform
        const uint32_t voff = (uint32_t)(__mul24(srow, wpitch) + 4 * scol);
#pragma unroll
        for (int u = 0; u < kWinLoads; ++u) {
            wvv[u] = 0u;
            if (lane < kWinRpi * kWinDw && (u < kWinLoads - 1 || srow < kWinRows - (kWinLoads - 1) * kWinRpi))
                __builtin_memcpy(&wvv[u], wb + (uint32_t)(u * kWinRpi * wpitch) + voff, 4);
        }
    }
    __builtin_amdgcn_wave_barrier();                     // the previous keypoint's LDS reads are done
    if (lane < kWinRpi * kWinDw) {
        uint32_t *dd = win + (srow * kWinDw + scol);
#pragma unroll
        for (int u = 0; u < kWinLoads; ++u)
            if (u < kWinLoads - 1 || srow < kWinRows - (kWinLoads - 1) * kWinRpi) dd[u * kWinRpi * kWinDw] = wvv[u];
    }
    __builtin_amdgcn_wave_barrier();
    // ---- moments ----
    // sum u * I = sum u * (I - 128) because the weights of the disc sum to zero: the pixels become signed bytes (xor
    // 0x80) and a chunk costs two v_dot4c_i32_i8
    int m10 = 0, m01 = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int d = (int)(win[cw[k]] ^ 0x80808080u);
        m10 = __builtin_amdgcn_sdot4(d, uw[k], m10, false);
        m01 = __builtin_amdgcn_sdot4(d, vw[k], m01, false);
    }
    // ---- row pass: 7-tap sums of window rows, two rows per item ----
#pragma unroll 1
    for (int q = 0; q < 3; ++q) {
        const int it = (int)((hitems >> (8 * q)) & 0xffu);         // this lane's q-th (pair-row, group) item
        const int pr = (it * 205) >> 11, g = it - pr * kHGroups;   // it / 10
        uint32_t s[2][4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t *p = &win[madi24(min(2 * pr + h, kWinRows - 1), kWinDw, g)];
            const uint32_t d0 = p[0], d1 = p[1], d2 = p[2];
            s[h][0] = __builtin_amdgcn_udot4(d0, h0a, __builtin_amdgcn_udot4(d1, h0b, 0, false), false);
            s[h][1] = __builtin_amdgcn_udot4(d0, h1a, __builtin_amdgcn_udot4(d1, h1b, 0, false), false);
            s[h][2] = __builtin_amdgcn_udot4(d0, h2a, __builtin_amdgcn_udot4(d1, h2b, __builtin_amdgcn_udot4(d2, h2c, 0, false), false), false);
            s[h][3] = __builtin_amdgcn_udot4(d0, h3a, __builtin_amdgcn_udot4(d1, h3b, __builtin_amdgcn_udot4(d2, h3c, 0, false), false), false);
        }
        hs[it] = make_uint4(__builtin_amdgcn_perm(s[1][0], s[0][0], 0x05040100u), __builtin_amdgcn_perm(s[1][1], s[0][1], 0x05040100u),
                            __builtin_amdgcn_perm(s[1][2], s[0][2], 0x05040100u), __builtin_amdgcn_perm(s[1][3], s[0][3], 0x05040100u));
    }
    m10 = wave_reduce_add(m10);
    m01 = wave_reduce_add(m01);
    const float angle = fast_atan2_deg((float)m01, (float)m10);
    const float factorPI = (float)(3.14159265358979323846 / 180.f);
    float a, bsn;
    det_sincos(__fmul_rn(angle, factorPI), &a, &bsn);
    __builtin_amdgcn_wave_barrier();                     // DS operations of a wavefront execute in order
    // ---- column pass on demand: a descriptor test needs the blurred patch at two points only, so the 7-tap column sum
    // is evaluated right at the 8 sample points of a lane (4 dwords of row sums -- two window rows each -- against the
    // tap pairs of the point's row parity) instead of over all 37 x 37 patch pixels first ----
    const uint32_t *hd = reinterpret_cast<const uint32_t *>(hs);   // row sums as dwords: [pair-row][column], 40 columns
    // cvRound of a rotated pattern coordinate plus 18, the patch index: adding 1.5 * 2^23 rounds the sum to an integer --
    // to nearest, ties to even, exactly what v_rndne / lrint do, for |t| < 2^22 -- and leaves it in the low mantissa bits:
    // two full-rate instructions (v_add_f32, v_sub_u32 with the + 18 folded in) instead of v_rndne + v_cvt_i32 + v_add
    auto round18 = [](float t) -> int {
        return (int)(__float_as_uint(__fadd_rn(t, 12582912.0f)) - (0x4B400000u - 18u));
    };
    uint32_t vw0, vw1, vw2, vw3, vx0, vx1, vx2, vx3;     // even-row tap pairs and their difference to the odd-row ones, in VGPRs
    asm("v_mov_b32 %0, %1" : "=v"(vw0) : "s"(w01)); asm("v_mov_b32 %0, %1" : "=v"(vx0) : "s"(w01 ^ w0h));
    asm("v_mov_b32 %0, %1" : "=v"(vw1) : "s"(w23)); asm("v_mov_b32 %0, %1" : "=v"(vx1) : "s"(w23 ^ w12));
    asm("v_mov_b32 %0, %1" : "=v"(vw2) : "s"(w45)); asm("v_mov_b32 %0, %1" : "=v"(vx2) : "s"(w45 ^ w34));
    asm("v_mov_b32 %0, %1" : "=v"(vw3) : "s"(w6l)); asm("v_mov_b32 %0, %1" : "=v"(vx3) : "s"(w6l ^ w56));
    auto blurred = [&](int R, int C) -> int {             // patch row / column; window rows R .. R + 6
        const uint32_t *p = hd + madi24(R >> 1, kHGroups * 4, C);
        const uint32_t d0 = p[0], d1 = p[kHGroups * 4], d2 = p[2 * kHGroups * 4], d3 = p[3 * kHGroups * 4];
        // tap pairs of the row's parity, selected by mask arithmetic on VGPR copies of the words: v_and + v_xor issue at the
        // full rate, the v_cmp + four v_cndmask of `odd ? a : b` do not (68.3 -> 67.2 us per 64 frames)
        const uint32_t om = 0u - (uint32_t)(R & 1);
        const uint32_t acc = udot2(d0, vw0 ^ (om & vx0), udot2(d1, vw1 ^ (om & vx1), udot2(d2, vw2 ^ (om & vx2),
                                   udot2(d3, vw3 ^ (om & vx3), 32768u))));
        return min((int)(acc >> 16), 255);                 // the taps sum to 257: saturate like the byte store did
    };
    unsigned long long bits[4];
    int t0v[4], t1v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float px0 = pat[j].x, py0 = pat[j].y, px1 = pat[j].z, py1 = pat[j].w;
        const int r0 = round18(__fadd_rn(__fmul_rn(px0, bsn), __fmul_rn(py0, a)));
        const int c0 = round18(__fsub_rn(__fmul_rn(px0, a), __fmul_rn(py0, bsn)));
        const int r1 = round18(__fadd_rn(__fmul_rn(px1, bsn), __fmul_rn(py1, a)));
        const int c1 = round18(__fsub_rn(__fmul_rn(px1, a), __fmul_rn(py1, bsn)));
        t0v[j] = blurred(r0, c0);
        t1v[j] = blurred(r1, c1);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) bits[j] = __ballot(t0v[j] < t1v[j]);
    if (lane < 4) {
        unsigned long long v = lane == 0 ? bits[0] : lane == 1 ? bits[1] : lane == 2 ? bits[2] : bits[3];
        reinterpret_cast<unsigned long long *>(out_desc + ((size_t)b * cap + oidx) * 32)[lane] = v;
    }
    if (lane == 0) {
        orbhip_keypoint kp;
        float fx = (float)kx, fy = (float)ky;
        const float scale = G.lv[level].scale;
        if (level != 0) { fx = __fmul_rn(fx, scale); fy = __fmul_rn(fy, scale); }
        kp.x = fx; kp.y = fy; kp.size = (float)G.lv[level].patch; kp.angle = angle; kp.response = (float)resp;
        kp.octave = level; kp.class_id = -1;
        out_kp[(size_t)b * cap + oidx] = kp;
    }
  }
}

__global__ void k_zero_status(int *status, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) status[i] = 0;
}

}  // namespace orbhip

// ===========================================================================
// host side
// ===========================================================================
using namespace orbhip;

// Development builds (-DORBHIP_DEVTOOLS, tools/ab_build.sh) can run a subset of the stages on buffers a full run left
// behind (tools/coexec.py) and launch the stamped FAST kernel (tools/fast_ab.py); the product library has neither.
#ifdef ORBHIP_DEVTOOLS
#define ORBHIP_STAGE_MASK(e) ((e)->stage_mask)
#else
#define ORBHIP_STAGE_MASK(e) 31
#endif

static int cv_round(double v) { return (int)lrint(v); }

static void drop_graph(orbhip_extractor *e)
{
    if (e->graph_exec) (void)hipGraphExecDestroy(e->graph_exec);
    e->graph_exec = nullptr; e->graph_batch = 0; e->graph_cap = 0;
}
static void free_geometry(orbhip_extractor *e)
{
    drop_graph(e);
    (void)hipFree(e->d_cells); (void)hipFree(e->d_cells2); (void)hipFree(e->d_tiles); (void)hipFree(e->d_pyrtab);
    e->d_cells = nullptr; e->d_cells2 = nullptr; e->d_tiles = nullptr; e->d_pyrtab = nullptr;
    e->bound = false;
}
static void free_batch(orbhip_extractor *e)
{
    drop_graph(e);
    (void)hipFree(e->d_pyr); (void)hipFree(e->d_blur); (void)hipFree(e->d_cell_cnt); (void)hipFree(e->d_cell_kp);
    (void)hipFree(e->d_keys); (void)hipFree(e->d_knode); (void)hipFree(e->d_sel); (void)hipFree(e->d_sel_cnt); (void)hipFree(e->d_status);
    e->d_pyr = e->d_blur = nullptr; e->d_cell_cnt = nullptr; e->d_cell_kp = nullptr; e->d_keys = nullptr;
    e->d_knode = nullptr; e->d_sel = nullptr; e->d_sel_cnt = nullptr; e->d_status = nullptr;
    e->batch_cap = 0;
    e->blur_valid = false;
}

// OpenCV's fixed-point bilinear coefficients of one destination coordinate (the host twin of resize_coef):
// columns drop the fraction at the ends (clamp_both = false), rows clamp both taps and keep it.
static void host_resize_coef(int d, double scale, int slen, bool clamp_both, int &s0, int &s1, int &c0, int &c1)
{
#pragma clang fp contract(off)
    const double fd = ((double)d + 0.5) * scale;
    float f = (float)(fd - 0.5);
    int s = (int)floorf(f);
    f = f - (float)s;
    if (clamp_both) {
        s0 = s < 0 ? 0 : (s < slen ? s : slen - 1);
        s1 = s + 1 < 0 ? 0 : (s + 1 < slen ? s + 1 : slen - 1);
    } else {
        if (s < 0) { f = 0.f; s = 0; }
        if (s >= slen - 1) { f = 0.f; s = slen - 1; }
        s0 = s; s1 = s + 1;
    }
    const float g = 1.f - f;
    c0 = (int)lrintf(g * 2048.f);
    c1 = (int)lrintf(f * 2048.f);
}
static int host_reflect101(int p, int len)
{
    if (p < 0) p = -p;
    if (p >= len) p = 2 * (len - 1) - p;
    return p;
}
static void pyr_chunks(PyrLevelTab &T, int rows_per_unit)
{
    T.nchunks = (T.words + 63) / 64;
    T.chunk_w = (T.words + T.nchunks - 1) / T.nchunks;
    T.rcp_chunks = T.nchunks > 1 ? (uint32_t)(((1ull << 32) + (unsigned)T.nchunks - 1) / (unsigned)T.nchunks) : 0u;
    T.units = ((T.prows + rows_per_unit - 1) / rows_per_unit) * T.nchunks;
}

// Row / column tables of k_pyr_base / k_pyr_rows for the bound image size (one device buffer).
static int build_pyr_tables(orbhip_extractor *e)
{
    const PyrGeom &G = e->G;
    std::vector<uint8_t> blob;
    auto put = [&blob](const void *p, size_t n) {
        const size_t at = (blob.size() + 63) & ~(size_t)63;
        blob.resize(at + n);
        memcpy(blob.data() + at, p, n);
        return at;
    };
    size_t off_row[ORBHIP_MAX_LEVELS] = {0}, off_col[ORBHIP_MAX_LEVELS] = {0}, off_lo[ORBHIP_MAX_LEVELS] = {0};
    for (int l = 0; l < G.nlevels; ++l) {
        const LevelGeom &L = G.lv[l];
        PyrLevelTab &T = e->ptab[l];
        memset(&T, 0, sizeof(T));
        T.prows = L.prows; T.pitch = L.pitch; T.plane_off = L.plane_off;
        e->ptab_rows[l] = false;
        if (l == 0) {
            T.words = L.pitch >> 4;
            T.src_h = L.h;
            std::vector<PyrCopyCol> col(T.words);
            std::vector<uint32_t> lo(T.words);
            for (int g = 0; g < T.words; ++g) {
                int rk[16], base = 1 << 30;
                for (int k = 0; k < 16; ++k) {
                    const int x = g * 16 - kPadL + k;
                    rk[k] = (x >= -kEdge && x < L.w + kEdge) ? host_reflect101(x, L.w) : -1;
                    if (rk[k] >= 0) base = std::min(base, rk[k]);
                }
                if (base == (1 << 30)) base = 0;
                base = std::min(base, L.w - 16);
                lo[g] = (uint32_t)base;
                for (int j = 0; j < 4; ++j) {
                    uint32_t a = 0, b = 0;
                    for (int k = 0; k < 4; ++k) {
                        const int idx = rk[4 * j + k] < 0 ? -1 : rk[4 * j + k] - base;
                        a |= (uint32_t)((idx >= 0 && idx < 8) ? idx : 0x0c) << (8 * k);
                        b |= (uint32_t)((idx >= 8 && idx < 16) ? idx - 8 : 0x0c) << (8 * k);
                    }
                    col[g].selA[j] = a; col[g].selB[j] = b;
                }
            }
            pyr_chunks(T, kPyrRows);
            off_col[l] = put(col.data(), col.size() * sizeof(PyrCopyCol));
            off_lo[l] = put(lo.data(), lo.size() * sizeof(uint32_t));
            e->ptab_rows[l] = true;
            continue;
        }
        const LevelGeom &P = G.lv[l - 1];
        const double sx = 1. / ((double)L.w / P.w), sy = 1. / ((double)L.h / P.h);   // OpenCV: scale = 1. / ((double)dst / src)
        T.words = (kPadL + L.w + kEdge + 3) >> 2;
        T.src_h = P.h; T.src_pitch = P.pitch;
        T.src_off = P.plane_off + (unsigned)(kEdge * P.pitch + kPadL);
        std::vector<PyrCol> col(T.words);
        std::vector<uint32_t> lo(T.words);
        bool ok = P.w >= 8;
        for (int pw = 0; pw < T.words && ok; ++pw) {
            int sxk[4], a0[4], a1[4], lo_ = 1 << 30, hi_ = -1;
            bool valid[4];
            for (int k = 0; k < 4; ++k) {
                const int x = pw * 4 - kPadL + k;
                valid[k] = x >= -kEdge && x < L.w + kEdge;
                sxk[k] = 0; a0[k] = a1[k] = 0;
                if (!valid[k]) continue;
                int s1;
                host_resize_coef(host_reflect101(x, L.w), sx, P.w, false, sxk[k], s1, a0[k], a1[k]);
                lo_ = std::min(lo_, sxk[k]); hi_ = std::max(hi_, sxk[k]);
            }
            if (hi_ < 0) { lo_ = 0; hi_ = 0; }
            if (hi_ - lo_ > 6) { ok = false; break; }
            lo_ = std::min(lo_, P.w - 8);
            lo[pw] = (uint32_t)lo_;
            for (int k = 0; k < 4; ++k) {
                if (!valid[k]) { col[pw].sel[k] = 0x0c0c0c0cu; col[pw].alv[k] = 0; continue; }
                const int rel = sxk[k] - lo_;
                int rel1 = rel + 1;
                if (sxk[k] + 1 > P.w - 1) { rel1 = rel; if (a1[k] != 0) ok = false; }   // the end column: its second tap has weight 0
                if (rel < 0 || rel1 > 7) ok = false;
                col[pw].sel[k] = 0x0c000c00u | (uint32_t)rel | ((uint32_t)rel1 << 16);
                col[pw].alv[k] = (uint32_t)a0[k] | ((uint32_t)a1[k] << 16);
            }
        }
        if (!ok) continue;   // this level keeps the general kernel
        std::vector<PyrRow> row(L.prows);
        for (int py = 0; py < L.prows; ++py) {
            int s0, s1, b0, b1;
            host_resize_coef(host_reflect101(py - kEdge, L.h), sy, P.h, true, s0, s1, b0, b1);
            row[py].s0 = s0; row[py].s1 = s1; row[py].B0 = (uint32_t)b0 << 12; row[py].B1 = (uint32_t)b1 << 12;
        }
        pyr_chunks(T, kPyrRows);
        off_row[l] = put(row.data(), row.size() * sizeof(PyrRow));
        off_col[l] = put(col.data(), col.size() * sizeof(PyrCol));
        off_lo[l] = put(lo.data(), lo.size() * sizeof(uint32_t));
        e->ptab_rows[l] = true;
    }
    ORBHIP_HIP_CHECK(hipMalloc(&e->d_pyrtab, blob.size() + 64));
    ORBHIP_HIP_CHECK(hipMemcpyAsync(e->d_pyrtab, blob.data(), blob.size(), hipMemcpyHostToDevice, e->stream));
    ORBHIP_HIP_CHECK(hipStreamSynchronize(e->stream));   // `blob` dies at return
    for (int l = 0; l < G.nlevels; ++l) {
        if (!e->ptab_rows[l]) continue;
        PyrLevelTab &T = e->ptab[l];
        T.row = reinterpret_cast<const PyrRow *>(e->d_pyrtab + off_row[l]);
        T.col = e->d_pyrtab + off_col[l];
        T.lo = reinterpret_cast<const uint32_t *>(e->d_pyrtab + off_lo[l]);
    }
    return ORBHIP_OK;
}

// One cell of the FAST kernel's table.  img_stride > 0: a level-0 cell that reads the caller's image (row stride
// img_stride) instead of the padded level-0 plane.
static FastCell make_fast_cell(const PyrGeom &G, const CellDesc &c, int img_stride)
{
    const LevelGeom &Lc = G.lv[c.level];
    FastCell f; memset(&f, 0, sizeof(f));
    // The sub-image is staged so that its first detection column (x0 + 3) lands on a dword boundary of the LDS row
    // whatever x0 is: LDS column 0 = x0 - 5 (global rows are read with byte-unaligned dword loads), i.e. the
    // kernel's "a" is the constant 1 and the first centre column is LDS column 8.  A row of 31 or 32 detection
    // columns is then 8 four-pixel groups, never 9 -- for 32 rows that is 256 dense items = exactly 4 rounds of
    // the wavefront instead of 4.5 (5).
    const int a = kFastLead, gxb = c.x0 - a, sw = c.x1 - c.x0, shh = c.y1 - c.y0;
    if (img_stride > 0 && c.level == 0) {
        f.src_off = (unsigned)(c.y0 * img_stride + gxb - 4);   // x0 >= 13: never left of the image
        f.pitch = (unsigned short)img_stride; f.img = 1;
    } else {
        f.src_off = Lc.plane_off + (unsigned)((kEdge + c.y0) * Lc.pitch + kPadL + gxb - 4);
        f.pitch = (unsigned short)Lc.pitch;
    }
    f.sw = (unsigned char)sw; f.sh = (unsigned char)shh; f.a = (unsigned char)a;
    f.kpx = (short)(c.offx - 4 - a); f.kpy = (short)(c.offy + 3);
    const int dwc = sw - 6, c_lo = a + 7, c_hi = c_lo + dwc;
    const int ngrp = dwc > 0 ? ((c_hi - 1) >> 2) - (c_lo >> 2) + 1 : 1;
    f.magic = ((1u << 18) + ngrp - 1) / ngrp;
    return f;
}

// Bind the handle to an image size: level geometry (:1111-1113), cell table (:769-829),
// OpenCV resize tables, blur tiles.
static int bind_geometry(orbhip_extractor *e, int rows, int cols)
{
    if (e->bound && e->G.rows == rows && e->G.cols == cols) return ORBHIP_OK;
    ORBHIP_HIP_CHECK(hipSetDevice(e->device));
    ORBHIP_HIP_CHECK(hipStreamSynchronize(e->stream));
    free_geometry(e);
    free_batch(e);
    PyrGeom &G = e->G;
    memset(&G, 0, sizeof(G));
    G.nlevels = e->nlevels; G.rows = rows; G.cols = cols; G.ini_th = e->iniTh; G.min_th = e->minTh;
    e->cells.clear(); e->tiles.clear();
    unsigned off = 0;
    int kp_base = 0, cand_base = 0, slot_cap = 1;
    for (int l = 0; l < e->nlevels; ++l) {
        LevelGeom &L = G.lv[l];
        const float scale = e->isf[l];
        L.w = cv_round((float)cols * scale);
        L.h = cv_round((float)rows * scale);
        if (L.w < 2 * kEdge + 2 || L.h < 2 * kEdge + 2 || L.w > 4000 || L.h > 4000) {
            set_error("level %d is %dx%d: unsupported (need 40..4000 px per side)", l, L.w, L.h);
            return ORBHIP_E_SIZE;
        }
        L.pitch = (kPadL + L.w + kEdge + 63) & ~63;
        L.prows = L.h + 2 * kEdge;
        L.plane_off = off;
        off += (unsigned)L.pitch * L.prows;
        off = (off + 255u) & ~255u;
        L.maxBX = L.w - kEdge + 3; L.maxBY = L.h - kEdge + 3;
        L.quota = e->nfeat[l];
        L.patch = (int)(kPatchSize * e->sf[l]);
        L.scale = e->sf[l];
        // cells (:781-829)
        const int minB = kEdge - 3;
        const float width = (float)(L.maxBX - minB), height = (float)(L.maxBY - minB);
        const float W = 30;
        const int nCols = (int)(width / W), nRows = (int)(height / W);
        L.cell_base = (int)e->cells.size();
        if (nCols >= 1 && nRows >= 1) {
            const int wCell = (int)ceilf(width / nCols), hCell = (int)ceilf(height / nRows);
            if (wCell + 6 > kSubMax || hCell + 6 > kSubMax) { set_error("cell too large"); return ORBHIP_E_SIZE; }
            for (int i = 0; i < nRows; ++i) {
                const float iniY = (float)(minB + i * hCell);
                float maxY = iniY + hCell + 6;
                if (iniY >= L.maxBY - 3) continue;
                if (maxY > L.maxBY) maxY = (float)L.maxBY;
                for (int j = 0; j < nCols; ++j) {
                    const float iniX = (float)(minB + j * wCell);
                    float maxX = iniX + wCell + 6;
                    if (iniX >= L.maxBX - 6) continue;
                    if (maxX > L.maxBX) maxX = (float)L.maxBX;
                    CellDesc c;
                    c.level = (short)l; c.x0 = (short)iniX; c.y0 = (short)iniY; c.x1 = (short)maxX; c.y1 = (short)maxY;
                    c.offx = (short)(j * wCell); c.offy = (short)(i * hCell); c.pad = 0;
                    e->cells.push_back(c);
                    int dw = c.x1 - c.x0 - 6, dh = c.y1 - c.y0 - 6;
                    if (dw > 0 && dh > 0) slot_cap = std::max(slot_cap, ((dw + 1) / 2) * ((dh + 1) / 2));
                }
            }
        }
        L.ncells = (int)e->cells.size() - L.cell_base;
        // octree roots (:543-545)
        const int bw = L.maxBX - minB, bh = L.maxBY - minB;
        int nIni = bh > 0 ? (int)roundf((float)bw / bh) : 0;
        if (nIni < 1) nIni = 1;  // the reference divides by zero for tall images; one root instead
        L.nIni = nIni;
        L.hX = (float)bw / nIni;
        L.kp_cap = std::max(L.quota + 3, 4 * nIni);
        L.kp_base = kp_base; kp_base += L.kp_cap;
        // blur tiles
        for (int ty = 0; ty * kBlurTH < L.h; ++ty)
            for (int tx = 0; tx * kBlurTW < L.w; ++tx) {
                TileDesc t; t.level = (short)l; t.tx = (short)tx; t.ty = (short)ty; t.pad = 0;
                e->tiles.push_back(t);
            }
    }
    {   // dynamic LDS carve-up of k_fast_cells from the largest cell of this geometry
        int msh = 7, mdh = 1, mdw = 1, mndw = 2;
        for (const CellDesc &c : e->cells) {
            msh = std::max(msh, c.y1 - c.y0);
            mdh = std::max(mdh, c.y1 - c.y0 - 6);
            mdw = std::max(mdw, c.x1 - c.x0 - 6);
            mndw = std::max(mndw, (kFastLead + (c.x1 - c.x0) + 3) >> 2);
        }
        FastLds &F2 = e->fast_lds;
        // compile-time row stride (odd: rows rotate over the LDS banks): staged dwords + one real dword on the left
        const int need = mndw + 1;
        F2.strideW = need <= 11 ? 11 : need <= 13 ? 13 : need <= 15 ? 15 : need <= 17 ? 17 : 21;
        if (need > 21) { set_error("cell geometry exceeds the FAST kernel limits"); return ORBHIP_E_SIZE; }
        F2.scoreW = F2.strideW - 2;
        F2.score_words = (mdh + 2) * F2.scoreW;
        // staging batches write whole row groups: round the image rows up to what they touch
        {
            const int lpr = F2.strideW <= 16 ? 16 : 32, rpi = 64 / lpr, rows_batch = rpi * kFastStageU;
            const int rows = ((msh + rows_batch - 1) / rows_batch) * rows_batch;
            F2.img_words = rows * F2.strideW;
        }
        // uint16 survivor list.  Every pixel of a cell may pass the pre-test, but few do: the list gets what is left of the
        // LDS share that lets 32 workgroups (8 wavefronts per SIMD) live on a CU -- or of the next larger share that holds
        // 128 entries and a whole row of the widest cell -- and the kernel repeats a round that overflows in row bands
        {
            const int full = mdw * mdh, row_px = 4 * ((mdw + 3) >> 2), fixed = (F2.img_words + F2.score_words) * 4;
            int cap = full;
            for (int wgs = 32; wgs >= 4; wgs -= 4) {
                const int share = kFastLdsPerCu / wgs / kFastLdsGranule * kFastLdsGranule;
                const int room = (share - fixed) / 2 - 2;
                if (room >= std::max(128, row_px)) { cap = std::min(full, room); break; }
            }
            F2.list_cap = cap;
            F2.list_words = (cap + 1) / 2 + 1;
        }
        e->fast_lds_bytes = (F2.img_words + F2.score_words + F2.list_words) * 4;
        e->cells2.clear();
        for (const CellDesc &c : e->cells) e->cells2.push_back(make_fast_cell(G, c, 0));
        e->cells_stride = 0;
        FastParams &FP = e->fast_params;
        memset(&FP, 0, sizeof(FP));
        FP.img_words = F2.img_words; FP.score_words = F2.score_words; FP.list_cap = F2.list_cap;
        FP.ini_th = G.ini_th; FP.min_th = G.min_th;
    }
    G.frame_bytes = off;
    G.ncells_total = (int)e->cells.size();
    G.slot_cap = slot_cap;
    G.kp_cap_total = kp_base;
    int maxn = 0;
    for (int l = 0; l < e->nlevels; ++l) {
        LevelGeom &L = G.lv[l];
        L.cand_base = cand_base;
        L.cand_cap = L.ncells * slot_cap;
        cand_base += L.cand_cap;
        maxn = std::max(maxn, std::max(L.kp_cap + 4, (L.ncells + 3) / 4));
    }
    G.cand_cap_total = std::max(cand_base, 1);
    {   // workgroup size of the octree kernel: the level-0 workgroup is the critical path (all (level, frame) workgroups
        // are co-resident); 512 threads halve its key loops (61 -> 54 us per 64 KITTI frames), 1024 would leave one
        // workgroup per CU (75 us).  Re-measured with the keys in HBM under three pipelines: 256 and 512 level, 1024 -4.5 %
        int max_cells = 0;
        for (int l = 0; l < e->nlevels; ++l) max_cells = std::max(max_cells, G.lv[l].ncells);
        e->octree_threads = max_cells >= 128 ? 512 : 256;
    }
    if (maxn <= 512) e->octree_maxn = 512;
    else if (maxn <= 2048) e->octree_maxn = 2048;
    else { set_error("nfeatures too large for the octree kernel (per-level cap %d > 2048)", maxn); return ORBHIP_E_ARG; }
    if (G.ncells_total == 0) { /* tiny image: no FAST cells anywhere; still a valid (empty) result */ }
    e->fast_params.frame_bytes = G.frame_bytes; e->fast_params.ncells_total = G.ncells_total; e->fast_params.slot_cap = G.slot_cap;
    if (!e->cells.empty()) {
        ORBHIP_HIP_CHECK(hipMalloc(&e->d_cells, e->cells.size() * sizeof(CellDesc)));
        ORBHIP_HIP_CHECK(hipMemcpyAsync(e->d_cells, e->cells.data(), e->cells.size() * sizeof(CellDesc), hipMemcpyHostToDevice, e->stream));
        ORBHIP_HIP_CHECK(hipMalloc(&e->d_cells2, e->cells2.size() * sizeof(FastCell)));
        ORBHIP_HIP_CHECK(hipMemcpyAsync(e->d_cells2, e->cells2.data(), e->cells2.size() * sizeof(FastCell), hipMemcpyHostToDevice, e->stream));
    }
    ORBHIP_HIP_CHECK(hipMalloc(&e->d_tiles, e->tiles.size() * sizeof(TileDesc)));
    ORBHIP_HIP_CHECK(hipMemcpyAsync(e->d_tiles, e->tiles.data(), e->tiles.size() * sizeof(TileDesc), hipMemcpyHostToDevice, e->stream));
    // uploads go through the handle's own stream (a legacy-stream hipMemcpy would tangle with another host thread's
    // stream capture: the stereo constructor runs two extractors on two threads); the host vectors die at return
    ORBHIP_HIP_CHECK(hipStreamSynchronize(e->stream));
    if (int rc = build_pyr_tables(e)) return rc;
    e->bound = true;
    return ORBHIP_OK;
}

static int ensure_batch(orbhip_extractor *e, int batch)
{
    if (batch <= e->batch_cap) return ORBHIP_OK;
    ORBHIP_HIP_CHECK(hipStreamSynchronize(e->stream));
    free_batch(e);
    const PyrGeom &G = e->G;
    const size_t B = (size_t)batch;
    ORBHIP_HIP_CHECK(hipMalloc(&e->d_pyr, B * G.frame_bytes));   // d_blur: allocated when a blurred plane is asked for
    ORBHIP_HIP_CHECK(hipMalloc(&e->d_cell_cnt, B * std::max(G.ncells_total, 1) * sizeof(int)));
    ORBHIP_HIP_CHECK(hipMalloc(&e->d_cell_kp, B * std::max(G.ncells_total, 1) * G.slot_cap * sizeof(uint32_t)));
    ORBHIP_HIP_CHECK(hipMalloc(&e->d_keys, B * G.cand_cap_total * sizeof(uint32_t)));
    ORBHIP_HIP_CHECK(hipMalloc(&e->d_knode, B * G.cand_cap_total * sizeof(unsigned short)));
    ORBHIP_HIP_CHECK(hipMalloc(&e->d_sel, B * G.kp_cap_total * sizeof(uint32_t)));
    ORBHIP_HIP_CHECK(hipMalloc(&e->d_sel_cnt, B * ORBHIP_MAX_LEVELS * sizeof(int)));
    ORBHIP_HIP_CHECK(hipMalloc(&e->d_status, B * sizeof(int)));
    e->batch_cap = batch;
    return ORBHIP_OK;
}

// Every extraction entry calls this before it enqueues anything: whatever the accessors cached about the previous
// batch (the lazily produced blurred planes) is stale from here on -- also when the work itself is a graph replay that
// never passes through launch_pipeline.
static int begin_extraction(orbhip_extractor *e, const uint8_t *d_images, int stride, size_t frame_stride)
{
    e->blur_valid = false;
    // mvImagePyramid[0] on demand (orbhip_extractor_set_lazy_level0): FAST and the descriptor kernel read level 0 from
    // the caller's image, the padded plane is only written when an accessor asks for it (ensure_level0).  Needs level 1
    // to be resized from the image too (its table-driven kernel does; the general kernel reads the padded plane).
    const bool lazy = e->lazy_l0 && e->G.nlevels > 1 && e->ptab_rows[1] && stride <= 65535 &&
                      (unsigned long long)e->G.rows * (unsigned long long)stride < (1ull << 31);
    const int want = lazy ? stride : 0;
    if (want != e->cells_stride && !e->cells.empty()) {
        // level-0 entries of the FAST cell table follow the image's row stride: rebuilt when it changes (rare), never
        // inside a stream capture
        ORBHIP_HIP_CHECK(hipStreamSynchronize(e->stream));
        drop_graph(e);
        for (size_t i = 0; i < e->cells.size(); ++i)
            if (e->cells[i].level == 0) e->cells2[i] = make_fast_cell(e->G, e->cells[i], want);
        ORBHIP_HIP_CHECK(hipMemcpyAsync(e->d_cells2, e->cells2.data(), e->cells2.size() * sizeof(FastCell), hipMemcpyHostToDevice, e->stream));
        ORBHIP_HIP_CHECK(hipStreamSynchronize(e->stream));
    }
    e->cells_stride = want;
    e->l0_in_image = lazy;
    e->l0_valid = !lazy;
    e->src_images = d_images; e->src_stride = stride; e->src_frame_stride = frame_stride;
    return ORBHIP_OK;
}

// The padded level-0 planes of the last batch, for the accessors that read them (orbhip_pyramid_level*, the blurred
// planes, ComputeStereoMatches): written now, from the image buffer of the last extraction, if that extraction skipped
// them.  `consumer`: a stream that will read the planes (ordered behind the copy), or null.
int orbhip::ensure_level0(orbhip_extractor *e, hipStream_t consumer)
{
    if (!e || !e->bound || e->last_batch <= 0) return ORBHIP_OK;
    ORBHIP_HIP_CHECK(hipSetDevice(e->device));
    if (!e->l0_valid) {
        PyrLevelTab none; memset(&none, 0, sizeof(none));
        const int nb0 = (e->ptab[0].units + 3) / 4;
        hipLaunchKernelGGL(k_pyr_base, dim3(nb0, e->last_batch), dim3(256), 0, e->stream, e->src_images, e->src_stride,
                           e->src_frame_stride, e->d_pyr, e->G.frame_bytes, e->ptab[0], none, nb0, (int *)nullptr);
        ORBHIP_HIP_CHECK(hipGetLastError());
        e->l0_valid = true;
    }
    if (consumer && consumer != e->stream) {
        if (!e->ev_l0) ORBHIP_HIP_CHECK(hipEventCreateWithFlags(&e->ev_l0, hipEventDisableTiming));
        ORBHIP_HIP_CHECK(hipEventRecord(e->ev_l0, e->stream));
        ORBHIP_HIP_CHECK(hipStreamWaitEvent(consumer, e->ev_l0, 0));
    }
    return ORBHIP_OK;
}

// `frame0`: first internal frame slot of this launch (the host path runs a batch as several chunks, each in its own
// slots, so that every frame's pyramid stays resident for orbhip_pyramid_level / ComputeStereoMatches afterwards)
static int launch_pipeline(orbhip_extractor *e, const uint8_t *d_images, int batch, int stride,
                           size_t frame_stride, orbhip_keypoint *d_kps, uint8_t *d_desc, int cap,
                           int *d_n, int *d_status, int frame0 = 0)
{
    const PyrGeom &G = e->G;
    hipStream_t s = e->stream;
    // per-frame internal buffers of this launch
    uint8_t *const b_pyr = e->d_pyr + (size_t)frame0 * G.frame_bytes;
    int *const b_cell_cnt = e->d_cell_cnt + (size_t)frame0 * std::max(G.ncells_total, 1);
    uint32_t *const b_cell_kp = e->d_cell_kp + (size_t)frame0 * std::max(G.ncells_total, 1) * G.slot_cap;
    uint32_t *const b_keys = e->d_keys + (size_t)frame0 * G.cand_cap_total;
    unsigned short *const b_knode = e->d_knode + (size_t)frame0 * G.cand_cap_total;
    uint32_t *const b_sel = e->d_sel + (size_t)frame0 * G.kp_cap_total;
    int *const b_sel_cnt = e->d_sel_cnt + (size_t)frame0 * ORBHIP_MAX_LEVELS;
    const bool prof = e->profiling;
    hipEvent_t *ev = prof ? &e->ev[(size_t)(e->prof_calls % orbhip_extractor::kProfRing) * orbhip_extractor::kProfEv] : nullptr;
    int *status = d_status ? d_status : e->d_status + frame0;
    const int sm = ORBHIP_STAGE_MASK(e);
    const bool lazy = e->l0_in_image;   // begin_extraction: level 0 is read from the image, its padded plane is not written
    if (!(sm & 1))   // k_pyr_base clears the status words itself
        hipLaunchKernelGGL(k_zero_status, dim3((batch + 255) / 256), dim3(256), 0, s, status, batch);
    if (prof) (void)hipEventRecord(ev[0], s);
#define ORBHIP_GATE_IN(st) do { if (e->gate_wait[st]) (void)hipStreamWaitEvent(s, e->gate_wait[st], 0); } while (0)
#define ORBHIP_GATE_OUT(st) do { if (e->gate_rec[st]) (void)hipEventRecord(e->gate_rec[st], s); } while (0)
    ORBHIP_GATE_IN(0);
    if (sm & 1) {
        // level 0 and, when its taps allow, level 1 in one launch
        const bool l1_rows = G.nlevels > 1 && e->ptab_rows[1];
        PyrLevelTab T1 = e->ptab[1];
        if (!l1_rows) memset(&T1, 0, sizeof(T1));
        const int nb0 = lazy ? 0 : (e->ptab[0].units + 3) / 4, nb1 = (T1.units + 3) / 4;   // lazy implies l1_rows: nb1 > 0
        hipLaunchKernelGGL(k_pyr_base, dim3(nb0 + nb1, batch), dim3(256), 0, s, d_images, stride, frame_stride, b_pyr,
                           G.frame_bytes, e->ptab[0], T1, nb0, status);
        for (int l = l1_rows ? 2 : 1; l < G.nlevels; ++l) {
            if (e->ptab_rows[l]) {
                hipLaunchKernelGGL(k_pyr_rows, dim3((e->ptab[l].units + 3) / 4, batch), dim3(256), 0, s, b_pyr, G.frame_bytes, e->ptab[l]);
                continue;
            }
            const LevelGeom &Ll = G.lv[l];
            int nl = (Ll.pitch >> 2) * ((Ll.prows + kPyrRows - 1) / kPyrRows);
            const LevelGeom &Pl = G.lv[l - 1];
            const ResizeScale RS = {1. / ((double)Ll.w / Pl.w), 1. / ((double)Ll.h / Pl.h)};   // OpenCV: scale = 1. / ((double)dst / src)
            hipLaunchKernelGGL(k_pyr_resize, dim3((nl + 255) / 256, batch), dim3(256), 0, s, b_pyr, G, l, RS);
        }
    }
    ORBHIP_GATE_OUT(0);
    if (prof) (void)hipEventRecord(ev[1], s);
    ORBHIP_GATE_IN(1);
    if (G.ncells_total > 0 && (sm & 2)) {
        const dim3 grid(G.ncells_total, batch);
        const size_t lb = (size_t)e->fast_lds_bytes;
        FastParams P = e->fast_params;
        // remap: frame = lin2 / ncells by multiplication; exact while lin2 * ncells < 2^32
        P.rcp_cells = (uint32_t)(((1ull << 32) + (unsigned)G.ncells_total - 1) / (unsigned)G.ncells_total);
        if ((unsigned long long)G.ncells_total * G.ncells_total * (unsigned long long)batch >= (1ull << 32)) {
            set_error("batch too large for the FAST kernel's work mapping"); return ORBHIP_E_SIZE;
        }
#define ORBHIP_FAST2(SWv) hipLaunchKernelGGL(k_fast_cells<SWv>, grid, dim3(64), lb, s, b_pyr, e->d_cells2, b_cell_cnt, b_cell_kp, P, d_images, frame_stride)
#ifdef ORBHIP_DEVTOOLS
        P.dev = e->fast_variant;
        if (e->fast_variant == 2 && e->fast_lds.strideW == 11)   // stamped diagnostic build (tools/fast_ab.py)
            hipLaunchKernelGGL((k_fast_cells<11, true>), grid, dim3(64), lb, s, b_pyr, e->d_cells2, b_cell_cnt, b_cell_kp, P, d_images, frame_stride);
        else
#endif
        switch (e->fast_lds.strideW) {
        case 11: ORBHIP_FAST2(11); break;
        case 13: ORBHIP_FAST2(13); break;
        case 15: ORBHIP_FAST2(15); break;
        case 17: ORBHIP_FAST2(17); break;
        default: ORBHIP_FAST2(21); break;
        }
#undef ORBHIP_FAST2
    }
    ORBHIP_GATE_OUT(1);
    if (prof) (void)hipEventRecord(ev[2], s);
    ORBHIP_GATE_IN(2);
    if (sm & 4) {
        const int ot = e->octree_threads;
#define ORBHIP_OCT(MAXNv, Tv) hipLaunchKernelGGL((k_octree<MAXNv, Tv>), dim3(G.nlevels, batch), dim3(Tv), 0, s, G, b_cell_cnt, b_cell_kp, \
                                                 b_keys, b_knode, b_sel, b_sel_cnt, status)
        if (e->octree_maxn == 512) {
            if (ot == 1024) ORBHIP_OCT(512, 1024); else if (ot == 512) ORBHIP_OCT(512, 512); else ORBHIP_OCT(512, 256);
        } else {
            if (ot == 1024) ORBHIP_OCT(2048, 1024); else if (ot == 512) ORBHIP_OCT(2048, 512); else ORBHIP_OCT(2048, 256);
        }
#undef ORBHIP_OCT
    }
    ORBHIP_GATE_OUT(2);
    if (prof) (void)hipEventRecord(ev[3], s);
    ORBHIP_GATE_IN(3);
    if (sm & 16) {
        // IC_Angle + the 7x7 blur of the keypoint's patch + rBRIEF in one kernel: no blurred plane exists unless
        // orbhip_blurred_level_download asks for one
        const int kDescPerWave = desc_per_wave(G.kp_cap_total, batch, e->num_cus);
        const dim3 grid((G.kp_cap_total + 4 * kDescPerWave - 1) / (4 * kDescPerWave), batch);
        hipLaunchKernelGGL(k_describe_fused, grid, dim3(256), 0, s, b_pyr, G, b_sel, b_sel_cnt, e->d_desc_tab,
                           e->d_patternf, d_kps, d_desc, cap, d_n, status, kDescPerWave, e->blurw,
                           reinterpret_cast<const uint32_t *>(e->d_desc_tab + 768), lazy ? d_images : (const uint8_t *)nullptr, stride,
                           frame_stride);
    }
    ORBHIP_GATE_OUT(3);
#undef ORBHIP_GATE_IN
#undef ORBHIP_GATE_OUT
    if (prof) { (void)hipEventRecord(ev[4], s); e->prof_calls++; }
    e->last_batch = frame0 + batch;
    ORBHIP_HIP_CHECK(hipGetLastError());
    return ORBHIP_OK;
}

extern "C" {

const char *orbhip_last_error(void) { return g_err; }

int orbhip_device_count(int *count)
{
    if (!count) return ORBHIP_E_ARG;
    int n = 0;
    hipError_t err = hipGetDeviceCount(&n);
    if (err != hipSuccess) { *count = 0; set_error("hipGetDeviceCount: %s", hipGetErrorString(err)); return ORBHIP_E_NODEVICE; }
    *count = n;
    return ORBHIP_OK;
}

int orbhip_extractor_create(int nfeatures, float scale_factor, int nlevels, int ini_th, int min_th,
                            int device, orbhip_extractor **out)
{
    if (!out || nlevels < 1 || nlevels > ORBHIP_MAX_LEVELS || nfeatures < 0 || !(scale_factor > 1.0f)) {
        set_error("orbhip_extractor_create: bad argument");
        return ORBHIP_E_ARG;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        set_error("no HIP device %d (found %d)", device, ndev);
        return ORBHIP_E_NODEVICE;
    }
    orbhip_extractor *e = new (std::nothrow) orbhip_extractor();
    if (!e) return ORBHIP_E_ARG;
    e->nfeatures = nfeatures; e->scaleFactor = scale_factor; e->nlevels = nlevels;
    e->iniTh = std::min(std::max(ini_th, 0), 255); e->minTh = std::min(std::max(min_th, 0), 255);
    e->device = device;
    // scale tables and quotas, src/ORBextractor.cc:415-446
    e->sf[0] = 1.0f; e->sig2[0] = 1.0f;
    for (int i = 1; i < nlevels; ++i) { e->sf[i] = (float)(e->sf[i - 1] * e->scaleFactor); e->sig2[i] = e->sf[i] * e->sf[i]; }
    for (int i = 0; i < nlevels; ++i) { e->isf[i] = 1.0f / e->sf[i]; e->isig2[i] = 1.0f / e->sig2[i]; }
    float factor = (float)(1.0f / e->scaleFactor);
    float nDesired = nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int l = 0; l < nlevels - 1; ++l) { e->nfeat[l] = cv_round(nDesired); sum += e->nfeat[l]; nDesired *= factor; }
    e->nfeat[nlevels - 1] = std::max(nfeatures - sum, 0);
    // umax, :454-469
    int v, v0, vmax = (int)floor(kHalfPatch * sqrtf(2.f) / 2 + 1), vmin = (int)ceil(kHalfPatch * sqrtf(2.f) / 2);
    const double hp2 = kHalfPatch * kHalfPatch;
    for (v = 0; v <= vmax; ++v) e->umax[v] = cv_round(sqrt(hp2 - v * v));
    for (v = kHalfPatch, v0 = 0; v >= vmin; --v) { while (e->umax[v0] == e->umax[v0 + 1]) ++v0; e->umax[v] = v0; ++v0; }
    const int bw[7] = {18, 34, 49, 55, 49, 34, 18};
    for (int i = 0; i < 7; ++i) e->blurw.w[i] = bw[i];

    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&e->own_stream, hipStreamNonBlocking) != hipSuccess) {
        set_error("hipSetDevice/hipStreamCreate failed");
        delete e;
        return ORBHIP_E_HIP;
    }
    e->stream = e->own_stream;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) e->num_cus = prop.multiProcessorCount;
    }
    // orientation disc (IC_Angle, :77-104: rows v = -15 .. 15, columns |u| <= umax[|v|]) cut into the aligned dwords of the
    // descriptor kernel's window tile (row stride 11 dwords, disc centre at row 21, byte column 21): chunk t = (row, dword);
    // lane l owns chunks l, 64 + l, 128 + l, 192 + l (consecutive lanes read consecutive dwords); integer sums do not
    // depend on the order
    int dtab[768 + 64];
    memset(dtab, 0, sizeof(dtab));
    {
        int nchunk = 0, npix = 0;
        for (int v = -kHalfPatch; v <= kHalfPatch; ++v) {
            const int d = e->umax[v < 0 ? -v : v];
            for (int c = (21 - d) / 4; c <= (21 + d) / 4; ++c) {
                uint32_t uwt = 0, vwt = 0;
                for (int k = 0; k < 4; ++k) {
                    const int u = 4 * c + k - 21;
                    if (u < -d || u > d) continue;
                    uwt |= (uint32_t)(uint8_t)(signed char)u << (8 * k);
                    vwt |= (uint32_t)(uint8_t)(signed char)v << (8 * k);
                    ++npix;
                }
                if (nchunk < 256) {
                    dtab[nchunk] = (v + 21) * 11 + c;
                    dtab[256 + nchunk] = (int)uwt;
                    dtab[512 + nchunk] = (int)vwt;
                }
                ++nchunk;
            }
        }
        if (npix != 749 || nchunk > 256) { set_error("disc table: %d pixels in %d chunks", npix, nchunk); orbhip_extractor_destroy(e); return ORBHIP_E_ARG; }
    }
    float patf[1024];
    for (int t = 0; t < 1024; ++t) patf[t] = (float)orbhip_rbrief_pattern[t];
    if (hipMalloc(&e->d_patternf, sizeof(patf)) != hipSuccess || hipMalloc(&e->d_desc_tab, sizeof(dtab)) != hipSuccess) {
        set_error("hipMalloc failed"); orbhip_extractor_destroy(e); return ORBHIP_E_HIP;
    }
    {
        // A descriptor test samples the blurred patch at (round(x*sin + y*cos), round(x*cos - y*sin)) of a pattern point
        // (x, y): |offset component| <= round(max radius) and offset length <= max radius + sqrt(0.5).  Patch row r'
        // (offset r' - 18) needs the row sums of window rows r' .. r' + 6; a row-pass item covers window rows 2pr, 2pr+1
        // and columns 4g .. 4g+3.
        double rmax = 0;
        for (int t = 0; t < 512; ++t) rmax = std::max(rmax, std::hypot((double)orbhip_rbrief_pattern[2 * t], (double)orbhip_rbrief_pattern[2 * t + 1]));
        const int cmax = (int)floor(rmax + 0.5);
        const double lim2 = (rmax + 0.7072) * (rmax + 0.7072);
        bool need[22 * 10] = {false};
        if (cmax > 18) { set_error("rBRIEF pattern radius exceeds the staged patch"); orbhip_extractor_destroy(e); return ORBHIP_E_ARG; }
        for (int dy = -cmax; dy <= cmax; ++dy)
            for (int dx = -cmax; dx <= cmax; ++dx) {
                if ((double)(dy * dy + dx * dx) > lim2) continue;
                for (int k = 0; k < 7; ++k) need[((dy + 18 + k) / 2) * 10 + (dx + 18) / 4] = true;
            }
        std::vector<int> items;
        for (int it = 0; it < 220; ++it) if (need[it]) items.push_back(it);
        if (items.size() > 192) {   // cannot happen for the rBRIEF pattern (189); keep a correct fallback: 4 rounds are not available
            set_error("row-pass item table overflow (%d)", (int)items.size()); orbhip_extractor_destroy(e); return ORBHIP_E_ARG;
        }
        while (items.size() < 192) items.push_back(items[0]);   // idle slots repeat an item (same value written twice)
        for (int l = 0; l < 64; ++l) dtab[768 + l] = items[l] | (items[64 + l] << 8) | (items[128 + l] << 16);
    }
    if (hipMemcpyAsync(e->d_desc_tab, dtab, sizeof(dtab), hipMemcpyHostToDevice, e->stream) != hipSuccess ||
        hipMemcpyAsync(e->d_patternf, patf, sizeof(patf), hipMemcpyHostToDevice, e->stream) != hipSuccess ||
        hipStreamSynchronize(e->stream) != hipSuccess) {
        set_error("upload of the orientation / rBRIEF tables failed"); orbhip_extractor_destroy(e); return ORBHIP_E_HIP;
    }
    *out = e;
    return ORBHIP_OK;
}

void orbhip_extractor_destroy(orbhip_extractor *e)
{
    if (!e) return;
    (void)hipSetDevice(e->device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    free_geometry(e);
    free_batch(e);
    (void)hipFree(e->d_patternf); (void)hipFree(e->d_desc_tab); (void)hipFree(e->d_img); (void)hipFree(e->d_okp); (void)hipFree(e->d_odesc); (void)hipFree(e->d_on);
    if (e->h_in) (void)hipHostFree(e->h_in);
    if (e->h_out) (void)hipHostFree(e->h_out);
    for (hipEvent_t v : e->ev) (void)hipEventDestroy(v);
    for (hipEvent_t v : e->ev_chunk) (void)hipEventDestroy(v);
    if (e->s_in) (void)hipStreamDestroy(e->s_in);
    if (e->s_out) (void)hipStreamDestroy(e->s_out);
    if (e->ev_l0) (void)hipEventDestroy(e->ev_l0);
    if (e->own_stream) (void)hipStreamDestroy(e->own_stream);
    delete e;
}

int orbhip_extractor_levels(const orbhip_extractor *e) { return e ? e->nlevels : ORBHIP_E_ARG; }

int orbhip_extractor_tables(const orbhip_extractor *e, float *scale, float *inv_scale, float *sigma2,
                            float *inv_sigma2, int32_t *feat)
{
    if (!e) return ORBHIP_E_ARG;
    for (int i = 0; i < e->nlevels; ++i) {
        if (scale) scale[i] = e->sf[i];
        if (inv_scale) inv_scale[i] = e->isf[i];
        if (sigma2) sigma2[i] = e->sig2[i];
        if (inv_sigma2) inv_sigma2[i] = e->isig2[i];
        if (feat) feat[i] = e->nfeat[i];
    }
    return ORBHIP_OK;
}

int orbhip_extractor_capacity(orbhip_extractor *e, int rows, int cols, int *cap)
{
    if (!e || !cap || rows <= 0 || cols <= 0) return ORBHIP_E_ARG;
    int rc = bind_geometry(e, rows, cols);
    if (rc) return rc;
    *cap = e->G.kp_cap_total;
    return ORBHIP_OK;
}

int orbhip_extractor_set_blur_kernel(orbhip_extractor *e, const int32_t w[7])
{
    if (!e || !w) return ORBHIP_E_ARG;
    int s = 0;
    for (int i = 0; i < 7; ++i) { if (w[i] < 0 || w[i] > 255) return ORBHIP_E_ARG; s += w[i]; }
    if (s > 257) { set_error("blur weights sum %d > 257 overflows the uint16 row pass", s); return ORBHIP_E_ARG; }
    for (int i = 0; i < 7; ++i) e->blurw.w[i] = w[i];
    drop_graph(e);   // the weights are a by-value kernel argument of the captured launches
    return ORBHIP_OK;
}

int orbhip_extractor_set_lazy_level0(orbhip_extractor *e, int on)
{
    if (!e) return ORBHIP_E_ARG;
    ORBHIP_HIP_CHECK(hipSetDevice(e->device));
    ORBHIP_HIP_CHECK(hipStreamSynchronize(e->stream));
    drop_graph(e);   // which kernels run, and their arguments, depend on it
    e->lazy_l0 = on != 0;
    return ORBHIP_OK;
}

int orbhip_extractor_set_stage_gate(orbhip_extractor *e, int stage, void *wait_event, void *record_event)
{
    if (!e || stage < 0 || stage > 3) return ORBHIP_E_ARG;
    ORBHIP_HIP_CHECK(hipSetDevice(e->device));
    ORBHIP_HIP_CHECK(hipStreamSynchronize(e->stream));
    drop_graph(e);
    if (wait_event != ORBHIP_GATE_KEEP) e->gate_wait[stage] = (hipEvent_t)wait_event;
    if (record_event != ORBHIP_GATE_KEEP) e->gate_rec[stage] = (hipEvent_t)record_event;
    return ORBHIP_OK;
}

int orbhip_extract_batch_device(orbhip_extractor *e, const void *d_images, int batch, int rows, int cols,
                                int stride, size_t frame_stride, void *d_kps, void *d_desc, int cap,
                                void *d_n, void *d_status)
{
    if (!e || !d_images || !d_kps || !d_desc || !d_n || batch <= 0 || rows <= 0 || cols <= 0 || stride < cols || cap <= 0) {
        set_error("orbhip_extract_batch_device: bad argument");
        return ORBHIP_E_ARG;
    }
    ORBHIP_HIP_CHECK(hipSetDevice(e->device));
    int rc = bind_geometry(e, rows, cols);
    if (rc) return rc;
    rc = ensure_batch(e, batch);
    if (rc) return rc;
    if ((rc = begin_extraction(e, (const uint8_t *)d_images, stride, frame_stride))) return rc;
    return launch_pipeline(e, (const uint8_t *)d_images, batch, stride, frame_stride, (orbhip_keypoint *)d_kps,
                           (uint8_t *)d_desc, cap, (int *)d_n, (int *)d_status);
}

int orbhip_extract_batch(orbhip_extractor *e, const uint8_t *images, int batch, int rows, int cols, int stride,
                         size_t frame_stride, orbhip_keypoint *kps, uint8_t *desc, int cap, int32_t *n)
{
    if (!e || !kps || !desc || !n || batch <= 0 || cap <= 0) { set_error("orbhip_extract_batch: bad argument"); return ORBHIP_E_ARG; }
    if (!images || rows <= 0 || cols <= 0) {  // empty image: silent return (:1046-1047)
        for (int b = 0; b < batch; ++b) n[b] = 0;
        return ORBHIP_OK;
    }
    if (stride < cols) return ORBHIP_E_ARG;
    ORBHIP_HIP_CHECK(hipSetDevice(e->device));
    int rc = bind_geometry(e, rows, cols);
    if (rc) return rc;
    // Large batches run as a software pipeline of 16-frame chunks on three streams (below).  There, a caller that keeps its
    // images in page-locked memory (hipHostMalloc / hipHostRegister, e.g. a pinned cv::Mat allocator) skips the staging
    // copy: the DMA engine reads every frame where it is with ONE 1-D copy of its rows in the caller's own row stride,
    // and the kernels read that layout (the stride is a kernel argument) -- no transfer is ever a rectangle copy.
    constexpr int kChunk = 16;
    const bool chunked = batch >= 2 * kChunk && !e->profiling;
    bool pinned_src = false;
    if (chunked) {
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, images) == hipSuccess) pinned_src = attr.type == hipMemoryTypeHost;
        else (void)hipGetLastError();   // pageable memory is "invalid value" for this query: not an error here
    }
    const int dstride = pinned_src ? stride : cols;                   // row stride of the frames in the device staging
    const size_t fbytes = (size_t)rows * dstride;                     // one frame slot there
    const size_t fcopy = (size_t)(rows - 1) * dstride + cols;         // bytes of a frame that are ever read
    const size_t img_bytes = (size_t)batch * fbytes;
    if (img_bytes > e->d_img_bytes) {
        drop_graph(e);
        (void)hipFree(e->d_img); e->d_img = nullptr; e->d_img_bytes = 0;
        ORBHIP_HIP_CHECK(hipMalloc(&e->d_img, img_bytes));
        e->d_img_bytes = img_bytes;
    }
    if ((size_t)cap * batch > e->out_slots || batch > e->out_batch) {
        drop_graph(e);
        (void)hipFree(e->d_okp); (void)hipFree(e->d_odesc); (void)hipFree(e->d_on);
        e->d_okp = nullptr; e->d_odesc = nullptr; e->d_on = nullptr;
        e->out_slots = 0; e->out_batch = 0;
        const size_t slots = (size_t)cap * batch;
        ORBHIP_HIP_CHECK(hipMalloc(&e->d_okp, slots * sizeof(orbhip_keypoint)));
        ORBHIP_HIP_CHECK(hipMalloc(&e->d_odesc, slots * 32));
        ORBHIP_HIP_CHECK(hipMalloc(&e->d_on, (size_t)batch * sizeof(int)));
        e->out_slots = slots; e->out_batch = batch;
    }
    // host -> pinned staging (row copies on the CPU) -> one DMA
    if (!pinned_src && img_bytes > e->h_in_bytes) {
        drop_graph(e);
        if (e->h_in) (void)hipHostFree(e->h_in);
        e->h_in = nullptr; e->h_in_bytes = 0;
        ORBHIP_HIP_CHECK(hipHostMalloc((void **)&e->h_in, img_bytes, hipHostMallocDefault));
        e->h_in_bytes = img_bytes;
    }
    const size_t out_bytes = (size_t)batch * (2 * sizeof(int) + (size_t)cap * (sizeof(orbhip_keypoint) + 32));
    if (out_bytes > e->h_out_bytes) {
        drop_graph(e);
        if (e->h_out) (void)hipHostFree(e->h_out);
        e->h_out = nullptr; e->h_out_bytes = 0;
        ORBHIP_HIP_CHECK(hipHostMalloc((void **)&e->h_out, out_bytes, hipHostMallocDefault));
        e->h_out_bytes = out_bytes;
    }
    ORBHIP_HIP_CHECK(hipStreamSynchronize(e->stream));   // the staging buffer of the previous call is free
    rc = ensure_batch(e, batch);
    if (rc) return rc;
    if ((rc = begin_extraction(e, e->d_img, dstride, fbytes))) return rc;   // frame f of this call = e->d_img + f * fbytes
    int *h_n = reinterpret_cast<int *>(e->h_out), *h_st = h_n + batch;
    orbhip_keypoint *h_kp = reinterpret_cast<orbhip_keypoint *>(h_st + batch);
    uint8_t *h_desc = reinterpret_cast<uint8_t *>(h_kp + (size_t)batch * cap);
    auto stage_in = [&](int b0, int nb) {   // caller's frames -> pinned staging (row copies on the CPU)
        for (int b = b0; b < b0 + nb; ++b) {
            const uint8_t *src = images + b * frame_stride;
            uint8_t *dst = e->h_in + (size_t)b * rows * cols;
            if (stride == cols) memcpy(dst, src, (size_t)rows * cols);
            else for (int r = 0; r < rows; ++r) memcpy(dst + (size_t)r * cols, src + (size_t)r * stride, cols);
        }
    };
    auto deliver = [&](int b0, int nb, int &bad) {   // pinned staging -> caller's buffers; truncated frames are still delivered
        for (int b = b0; b < b0 + nb; ++b) {
            n[b] = std::min(h_n[b], cap);
            if (h_st[b] != 0 && bad < 0) bad = b;
            if (n[b] > 0) {
                memcpy(kps + (size_t)b * cap, h_kp + (size_t)b * cap, (size_t)n[b] * sizeof(orbhip_keypoint));
                memcpy(desc + (size_t)b * cap * 32, h_desc + (size_t)b * cap * 32, (size_t)n[b] * 32);
            }
        }
    };
    // The chunk pipeline: while the kernels of chunk k run, chunk k+1 is copied into pinned memory by the CPU and over
    // PCIe by the DMA engine, and the results of chunk k-1 travel back and are handed to the caller.  What a Tracking
    // thread sees is still one synchronous call.
    if (chunked) {
        if (!e->s_in) {
            ORBHIP_HIP_CHECK(hipStreamCreateWithFlags(&e->s_in, hipStreamNonBlocking));
            ORBHIP_HIP_CHECK(hipStreamCreateWithFlags(&e->s_out, hipStreamNonBlocking));
        }
        const int nchunks = (batch + kChunk - 1) / kChunk;
        while ((int)e->ev_chunk.size() < 3 * nchunks) {
            hipEvent_t ev;
            ORBHIP_HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            e->ev_chunk.push_back(ev);
        }
        for (int c = 0; c < nchunks; ++c) {
            const int b0 = c * kChunk, nb = std::min(kChunk, batch - b0);
            hipEvent_t ev_in = e->ev_chunk[3 * c], ev_k = e->ev_chunk[3 * c + 1], ev_out = e->ev_chunk[3 * c + 2];
            if (pinned_src) {
                if (frame_stride == fbytes)        // frames back to back in the caller's layout: one copy per chunk
                    ORBHIP_HIP_CHECK(hipMemcpyAsync(e->d_img + b0 * fbytes, images + b0 * frame_stride, (nb - 1) * fbytes + fcopy,
                                                    hipMemcpyHostToDevice, e->s_in));
                else
                    for (int b = b0; b < b0 + nb; ++b)
                        ORBHIP_HIP_CHECK(hipMemcpyAsync(e->d_img + b * fbytes, images + b * frame_stride, fcopy, hipMemcpyHostToDevice, e->s_in));
            } else {
                stage_in(b0, nb);
                ORBHIP_HIP_CHECK(hipMemcpyAsync(e->d_img + b0 * fbytes, e->h_in + b0 * fbytes, nb * fbytes, hipMemcpyHostToDevice, e->s_in));
            }
            ORBHIP_HIP_CHECK(hipEventRecord(ev_in, e->s_in));
            ORBHIP_HIP_CHECK(hipStreamWaitEvent(e->stream, ev_in, 0));
            rc = launch_pipeline(e, e->d_img + b0 * fbytes, nb, dstride, fbytes, e->d_okp + (size_t)b0 * cap, e->d_odesc + (size_t)b0 * cap * 32,
                                 cap, e->d_on + b0, nullptr, b0);
            if (rc) { (void)hipDeviceSynchronize(); return rc; }
            ORBHIP_HIP_CHECK(hipEventRecord(ev_k, e->stream));
            ORBHIP_HIP_CHECK(hipStreamWaitEvent(e->s_out, ev_k, 0));
            ORBHIP_HIP_CHECK(hipMemcpyAsync(h_n + b0, e->d_on + b0, nb * sizeof(int), hipMemcpyDeviceToHost, e->s_out));
            ORBHIP_HIP_CHECK(hipMemcpyAsync(h_st + b0, e->d_status + b0, nb * sizeof(int), hipMemcpyDeviceToHost, e->s_out));
            ORBHIP_HIP_CHECK(hipMemcpyAsync(h_kp + (size_t)b0 * cap, e->d_okp + (size_t)b0 * cap, (size_t)nb * cap * sizeof(orbhip_keypoint),
                                            hipMemcpyDeviceToHost, e->s_out));
            ORBHIP_HIP_CHECK(hipMemcpyAsync(h_desc + (size_t)b0 * cap * 32, e->d_odesc + (size_t)b0 * cap * 32, (size_t)nb * cap * 32,
                                            hipMemcpyDeviceToHost, e->s_out));
            ORBHIP_HIP_CHECK(hipEventRecord(ev_out, e->s_out));
        }
        int bad = -1;
        for (int c = 0; c < nchunks; ++c) {
            ORBHIP_HIP_CHECK(hipEventSynchronize(e->ev_chunk[3 * c + 2]));
            deliver(c * kChunk, std::min(kChunk, batch - c * kChunk), bad);
        }
        ORBHIP_HIP_CHECK(hipStreamSynchronize(e->stream));
        if (bad >= 0) { set_error("frame %d: capacity exceeded (cap %d); outputs are truncated", bad, cap); return ORBHIP_E_CAPACITY; }
        return ORBHIP_OK;
    }
    stage_in(0, batch);
    auto enqueue = [&]() -> int {
        ORBHIP_HIP_CHECK(hipMemcpyAsync(e->d_img, e->h_in, img_bytes, hipMemcpyHostToDevice, e->stream));
        int r = launch_pipeline(e, e->d_img, batch, cols, (size_t)rows * cols, e->d_okp, e->d_odesc, cap, e->d_on, nullptr);
        if (r) return r;
        ORBHIP_HIP_CHECK(hipMemcpyAsync(h_n, e->d_on, batch * sizeof(int), hipMemcpyDeviceToHost, e->stream));
        ORBHIP_HIP_CHECK(hipMemcpyAsync(h_st, e->d_status, batch * sizeof(int), hipMemcpyDeviceToHost, e->stream));
        ORBHIP_HIP_CHECK(hipMemcpyAsync(h_kp, e->d_okp, (size_t)batch * cap * sizeof(orbhip_keypoint), hipMemcpyDeviceToHost, e->stream));
        ORBHIP_HIP_CHECK(hipMemcpyAsync(h_desc, e->d_odesc, (size_t)batch * cap * 32, hipMemcpyDeviceToHost, e->stream));
        return ORBHIP_OK;
    };
    // Every pointer and by-value argument of the sequence is fixed for a (geometry, batch, cap, buffers) combination:
    // capture it once, replay it with one launch.  Anything that would change an argument drops the graph.
    bool gated = false;   // stage gates wait for / record other streams' events: plain launches, no capture
    for (int st = 0; st < 4; ++st) gated = gated || e->gate_wait[st] || e->gate_rec[st];
    if (!e->profiling && !gated) {
        if (!e->graph_exec || e->graph_batch != batch || e->graph_cap != cap) {
            drop_graph(e);
            hipGraph_t graph = nullptr;
            ORBHIP_HIP_CHECK(hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal));
            rc = enqueue();
            const hipError_t ce = hipStreamEndCapture(e->stream, &graph);
            if (rc || ce != hipSuccess || !graph) {
                if (graph) (void)hipGraphDestroy(graph);
                set_error("stream capture of the extraction sequence failed: %s", hipGetErrorString(ce));
                return rc ? rc : ORBHIP_E_HIP;
            }
            const hipError_t ie = hipGraphInstantiate(&e->graph_exec, graph, nullptr, nullptr, 0);
            (void)hipGraphDestroy(graph);
            if (ie != hipSuccess) { e->graph_exec = nullptr; set_error("hipGraphInstantiate failed: %s", hipGetErrorString(ie)); return ORBHIP_E_HIP; }
            e->graph_batch = batch; e->graph_cap = cap;
        }
        ORBHIP_HIP_CHECK(hipGraphLaunch(e->graph_exec, e->stream));
        e->last_batch = batch;   // launch_pipeline's bookkeeping (a replay does not run it)
    } else {
        rc = enqueue();
        if (rc) return rc;
    }
    ORBHIP_HIP_CHECK(hipStreamSynchronize(e->stream));
    // every frame's (possibly truncated) result is delivered before a capacity error is reported: n[] and the first
    // n[b] <= cap keypoints / descriptors of every frame are valid either way
    int bad = -1;
    deliver(0, batch, bad);
    if (bad >= 0) { set_error("frame %d: capacity exceeded (cap %d); outputs are truncated", bad, cap); return ORBHIP_E_CAPACITY; }
    return ORBHIP_OK;
}

int orbhip_extract(orbhip_extractor *e, const uint8_t *image, int rows, int cols, int stride,
                   orbhip_keypoint *kps, uint8_t *desc, int cap, int *n)
{
    if (!n) return ORBHIP_E_ARG;
    int32_t nn = 0;
    int rc = orbhip_extract_batch(e, image, 1, rows, cols, stride, 0, kps, desc, cap, &nn);
    *n = nn;
    return rc;
}

int orbhip_extractor_sync(orbhip_extractor *e)
{
    if (!e) return ORBHIP_E_ARG;
    ORBHIP_HIP_CHECK(hipSetDevice(e->device));
    ORBHIP_HIP_CHECK(hipStreamSynchronize(e->stream));
    return ORBHIP_OK;
}

void *orbhip_extractor_stream(orbhip_extractor *e) { return e ? (void *)e->stream : nullptr; }

int orbhip_extractor_set_stream(orbhip_extractor *e, void *stream)
{
    if (!e) return ORBHIP_E_ARG;
    ORBHIP_HIP_CHECK(hipSetDevice(e->device));
    ORBHIP_HIP_CHECK(hipStreamSynchronize(e->stream));
    drop_graph(e);
    e->stream = stream ? (hipStream_t)stream : e->own_stream;
    return ORBHIP_OK;
}

int orbhip_pyramid_level(orbhip_extractor *e, int frame, int level, int *rows, int *cols, int *stride, const void **d_roi)
{
    if (!e || !e->bound || frame < 0 || frame >= e->last_batch || level < 0 || level >= e->nlevels) return ORBHIP_E_ARG;
    const LevelGeom &L = e->G.lv[level];
    if (level == 0 && d_roi) { if (int rc = ensure_level0(e, nullptr)) return rc; }
    if (rows) *rows = L.h;
    if (cols) *cols = L.w;
    if (stride) *stride = L.pitch;
    if (d_roi) *d_roi = e->d_pyr + (size_t)frame * e->G.frame_bytes + L.plane_off + (size_t)kEdge * L.pitch + kPadL;
    return ORBHIP_OK;
}

// Transfer rule of this library: every copy between host and device is a 1-D hipMemcpyAsync on a stream of the handle,
// to or from page-locked staging memory the handle owns, followed by a wait on that stream and a CPU copy to / from the
// caller's memory.  No legacy-stream copies, no rectangle (2-D) copies, no DMA into caller-owned pageable pages.
static int ensure_host_out(orbhip_extractor *e, size_t bytes)
{
    if (bytes <= e->h_out_bytes) return ORBHIP_OK;
    drop_graph(e);   // a captured sequence holds the old staging pointer
    ORBHIP_HIP_CHECK(hipStreamSynchronize(e->stream));
    if (e->h_out) (void)hipHostFree(e->h_out);
    e->h_out = nullptr; e->h_out_bytes = 0;
    ORBHIP_HIP_CHECK(hipHostMalloc((void **)&e->h_out, bytes, hipHostMallocDefault));
    e->h_out_bytes = bytes;
    return ORBHIP_OK;
}

static int download_plane(orbhip_extractor *e, const uint8_t *base, int frame, int level, int with_border, uint8_t *dst, int dst_stride)
{
    if (!e || !e->bound || !dst || frame < 0 || frame >= e->last_batch || level < 0 || level >= e->nlevels) return ORBHIP_E_ARG;
    const LevelGeom &L = e->G.lv[level];
    const int bo = with_border ? kEdge : 0;
    const int w = L.w + 2 * bo, h = L.h + 2 * bo;
    if (dst_stride < w) return ORBHIP_E_ARG;
    // whole padded rows in one contiguous stream-ordered copy, cropped on the host
    const uint8_t *src = base + (size_t)frame * e->G.frame_bytes + L.plane_off + (size_t)(kEdge - bo) * L.pitch;
    ORBHIP_HIP_CHECK(hipSetDevice(e->device));
    const size_t bytes = (size_t)h * L.pitch;
    if (int rc = ensure_host_out(e, bytes)) return rc;
    ORBHIP_HIP_CHECK(hipMemcpyAsync(e->h_out, src, bytes, hipMemcpyDeviceToHost, e->stream));
    ORBHIP_HIP_CHECK(hipStreamSynchronize(e->stream));
    for (int r = 0; r < h; ++r) memcpy(dst + (size_t)r * dst_stride, e->h_out + (size_t)r * L.pitch + (kPadL - bo), (size_t)w);
    return ORBHIP_OK;
}

int orbhip_pyramid_level_download(orbhip_extractor *e, int frame, int level, int with_border, uint8_t *dst, int dst_stride)
{
    if (level == 0) { if (int rc = ensure_level0(e, nullptr)) return rc; }
    return download_plane(e, e ? e->d_pyr : nullptr, frame, level, with_border, dst, dst_stride);
}

int orbhip_blurred_level_download(orbhip_extractor *e, int frame, int level, uint8_t *dst, int dst_stride)
{
    if (e && e->bound && !e->blur_valid && e->last_batch > 0) {
        // the pipeline blurs patches inside the descriptor kernel; the full blurred planes (and their buffer) are
        // produced on request
        if (int rc = ensure_level0(e, nullptr)) return rc;
        if (!e->d_blur) ORBHIP_HIP_CHECK(hipMalloc(&e->d_blur, (size_t)e->batch_cap * e->G.frame_bytes));
        hipLaunchKernelGGL(k_blur, dim3((unsigned)e->tiles.size(), e->last_batch), dim3(256), 0, e->stream, e->d_pyr, e->d_blur, e->G,
                           e->d_tiles, e->blurw);
        ORBHIP_HIP_CHECK(hipGetLastError());
        e->blur_valid = true;
    }
    return download_plane(e, e ? e->d_blur : nullptr, frame, level, 0, dst, dst_stride);
}

int orbhip_level_candidates(orbhip_extractor *e, int frame, int level, int32_t *x, int32_t *y, int32_t *score, int cap, int *n)
{
    if (!e || !e->bound || !n || frame < 0 || frame >= e->last_batch || level < 0 || level >= e->nlevels) return ORBHIP_E_ARG;
    const PyrGeom &G = e->G;
    const LevelGeom &L = G.lv[level];
    ORBHIP_HIP_CHECK(hipSetDevice(e->device));
    const size_t cnt_bytes = (size_t)std::max(L.ncells, 1) * sizeof(int);
    const size_t kp_off = (cnt_bytes + 255) & ~(size_t)255, kp_bytes = (size_t)std::max(L.ncells, 1) * G.slot_cap * sizeof(uint32_t);
    if (int rc = ensure_host_out(e, kp_off + kp_bytes)) return rc;
    const int *cnt = reinterpret_cast<const int *>(e->h_out);
    const uint32_t *kp = reinterpret_cast<const uint32_t *>(e->h_out + kp_off);
    if (L.ncells > 0) {
        ORBHIP_HIP_CHECK(hipMemcpyAsync(e->h_out, e->d_cell_cnt + (size_t)frame * G.ncells_total + L.cell_base, L.ncells * sizeof(int),
                                        hipMemcpyDeviceToHost, e->stream));
        ORBHIP_HIP_CHECK(hipMemcpyAsync(e->h_out + kp_off, e->d_cell_kp + ((size_t)frame * G.ncells_total + L.cell_base) * G.slot_cap,
                                        (size_t)L.ncells * G.slot_cap * sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream));
    }
    ORBHIP_HIP_CHECK(hipStreamSynchronize(e->stream));
    int k = 0;
    for (int c = 0; c < L.ncells; ++c)
        for (int i = 0; i < cnt[c]; ++i) {
            uint32_t v = kp[(size_t)c * G.slot_cap + i];
            if (k < cap) { if (x) x[k] = v & 0xfff; if (y) y[k] = (v >> 12) & 0xfff; if (score) score[k] = v >> 24; }
            ++k;
        }
    *n = k;
    return k > cap ? ORBHIP_E_CAPACITY : ORBHIP_OK;
}

#ifdef ORBHIP_DEVTOOLS
// development exports (not part of include/orbhip.h, absent from the product library): tools/coexec.py, tools/fast_ab.py
int orbhip_dev_set_stage_mask(orbhip_extractor *e, int mask) { if (!e) return ORBHIP_E_ARG; e->stage_mask = mask; return ORBHIP_OK; }
int orbhip_dev_set_octree_threads(orbhip_extractor *e, int t) { if (!e) return ORBHIP_E_ARG; e->octree_threads = t; return ORBHIP_OK; }
int orbhip_dev_set_fast_variant(orbhip_extractor *e, int v)
{
    if (!e) return ORBHIP_E_ARG;
    e->fast_variant = v;                    // 0 product; 2 stamped diagnostic build; 3 / 4 exit after the prologue / after staging
    return ORBHIP_OK;
}

// diagnostic: read (and clear) the per-phase cycle sums of the stamped FAST build (variant 2)
int orbhip_dev_fast_stamps(unsigned long long out[8])
{
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(orbhip::g_fast_stamps), sizeof(z)) != hipSuccess) return ORBHIP_E_HIP;
    if (hipMemcpyToSymbol(HIP_SYMBOL(orbhip::g_fast_stamps), z, sizeof(z)) != hipSuccess) return ORBHIP_E_HIP;
    return ORBHIP_OK;
}
#endif

int orbhip_extractor_set_profiling(orbhip_extractor *e, int on)
{
    if (!e) return ORBHIP_E_ARG;
    ORBHIP_HIP_CHECK(hipSetDevice(e->device));
    if (on && e->ev.empty()) {
        e->ev.resize((size_t)orbhip_extractor::kProfRing * orbhip_extractor::kProfEv);
        for (auto &v : e->ev) ORBHIP_HIP_CHECK(hipEventCreate(&v));
    }
    e->profiling = on != 0;
    e->prof_calls = 0;
    return ORBHIP_OK;
}

int orbhip_extractor_stage_times(orbhip_extractor *e, float us[6])
{
    if (!e || !us || e->prof_calls == 0) return ORBHIP_E_ARG;
    ORBHIP_HIP_CHECK(hipSetDevice(e->device));
    ORBHIP_HIP_CHECK(hipStreamSynchronize(e->stream));
    const long n = std::min<long>(e->prof_calls, orbhip_extractor::kProfRing);
    double acc[6] = {0, 0, 0, 0, 0, 0};
    // stage -> (begin, end) event of the set, all on the launch stream
    // us[3] was the separate blur stage; the blur is fused into the descriptor kernel, nothing is launched (or timed) there
    static const int span[6][2] = {{0, 1}, {1, 2}, {2, 3}, {-1, -1}, {3, 4}, {0, 4}};
    for (long c = 0; c < n; ++c) {
        hipEvent_t *ev = &e->ev[(size_t)c * orbhip_extractor::kProfEv];
        for (int i = 0; i < 6; ++i) {
            float ms = 0;
            if (span[i][0] < 0) continue;
            ORBHIP_HIP_CHECK(hipEventElapsedTime(&ms, ev[span[i][0]], ev[span[i][1]]));
            acc[i] += ms * 1000.0;
        }
    }
    for (int i = 0; i < 6; ++i) us[i] = (float)(acc[i] / n);
    return ORBHIP_OK;
}

}  // extern "C"
