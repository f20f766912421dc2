/*
 * orb_oracle.c -- CPU oracle (plain C) for the ORB extract + match hot path.
 * TEST INFRASTRUCTURE ONLY; see orb_oracle.h for scope, citations and the
 * "parity unpinned" statement.  Build: make -C oracle  (gcc -O2 -ffp-contract=off).
 */
#include "orb_oracle.h"
#include "rbrief_pattern.h"

#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define EDGE_THRESHOLD 19   /* ORBextractor.cc:74 */
#define HALF_PATCH_SIZE 15  /* :73 */
#define PATCH_SIZE 31       /* :72 */

/* ------------------------------------------------------------------------ */
/* OpenCV scalar helpers (restated)                                          */
/* ------------------------------------------------------------------------ */

/* cvRound: x86 cvtsd2si under the default rounding mode = round-half-to-even */
int oracle_cvround(double v) { return (int)lrint(v); }
static int cv_floor(double v) { return (int)floor(v); }
static int cv_ceil(double v) { return (int)ceil(v); }
static short sat_short(int v) { return (short)(v < -32768 ? -32768 : v > 32767 ? 32767 : v); }
static uint8_t sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }

/* cv::fastAtan2 (degrees, [0,360)), 7th-order odd polynomial; strict float ops */
float oracle_fast_atan2(float y, float x)
{
    const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
    const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
    const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
    const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
    float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

/* Deterministic cosf/sinf stand-in for ORBextractor.cc:113 ((float)cos(angle)):
 * Cody-Waite reduction by pi/2 and the classic degree-13/12 minimax kernels,
 * evaluated in IEEE double with separate mul/add, then rounded once to float.
 * The same operation sequence is used by the HIP kernel, so both sides agree
 * bit-for-bit; tests report how often this equals libm's cosf/sinf. */
void oracle_det_sincos(float angle_rad, float *c, float *s)
{
    const double two_over_pi = 6.36619772367581382433e-01;
    const double pio2_hi = 1.57079632673412561417e+00; /* 33 bits of pi/2 */
    const double pio2_lo = 6.07710050650619224932e-11;
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double x = (double)angle_rad;
    double kd = floor(x * two_over_pi + 0.5);
    int k = (int)kd;
    double r = (x - kd * pio2_hi) - kd * pio2_lo;
    double z = r * r;
    double ps = S1 + z * (S2 + z * (S3 + z * (S4 + z * (S5 + z * S6))));
    double sn = r + (r * z) * ps;
    double pc = C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6))));
    double cs = (1.0 - 0.5 * z) + (z * z) * pc;
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

/* BORDER_REFLECT_101 index */
static int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * (len - 1) - p;
    }
    return p;
}

/* cv::copyMakeBorder(src, dst, b,b,b,b, BORDER_REFLECT_101 [+ISOLATED]) */
static void copy_make_border101(const uint8_t *src, int sstep, int w, int h, uint8_t *dst,
                                int dstep, int b)
{
    for (int y = -b; y < h + b; ++y) {
        const uint8_t *srow = src + (size_t)reflect101(y, h) * sstep;
        uint8_t *drow = dst + (size_t)(y + b) * dstep;
        for (int x = -b; x < w + b; ++x) drow[x + b] = srow[reflect101(x, w)];
    }
}

/* cv::resize(src,dst,dsize,0,0,INTER_LINEAR) for 8UC1: fixed-point Q11
 * coefficients, HResizeLinear int32 rows, VResizeLinear<uchar> rounding. */
void oracle_resize_linear(const uint8_t *src, int sstep, int sw, int sh, uint8_t *dst,
                          int dstep, int dw, int dh)
{
    const int ONE = 2048; /* INTER_RESIZE_COEF_SCALE */
    double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    int *xofs = (int *)malloc(sizeof(int) * dw);
    short *alpha = (short *)malloc(sizeof(short) * 2 * dw);
    int *row0 = (int *)malloc(sizeof(int) * dw), *row1 = (int *)malloc(sizeof(int) * dw);
    for (int dx = 0; dx < dw; ++dx) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[dx] = sx;
        alpha[2 * dx] = sat_short(oracle_cvround((1.f - fx) * ONE));
        alpha[2 * dx + 1] = sat_short(oracle_cvround(fx * ONE));
    }
    for (int dy = 0; dy < dh; ++dy) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor(fy);
        fy -= sy;
        short b0 = sat_short(oracle_cvround((1.f - fy) * ONE));
        short b1 = sat_short(oracle_cvround(fy * ONE));
        int sy0 = sy < 0 ? 0 : (sy < sh ? sy : sh - 1);
        int sy1 = sy + 1 < 0 ? 0 : (sy + 1 < sh ? sy + 1 : sh - 1);
        const uint8_t *S0 = src + (size_t)sy0 * sstep, *S1 = src + (size_t)sy1 * sstep;
        for (int dx = 0; dx < dw; ++dx) {
            int sx = xofs[dx];
            if (sx + 1 < sw) {
                row0[dx] = S0[sx] * alpha[2 * dx] + S0[sx + 1] * alpha[2 * dx + 1];
                row1[dx] = S1[sx] * alpha[2 * dx] + S1[sx + 1] * alpha[2 * dx + 1];
            } else {
                row0[dx] = S0[sx] * ONE;
                row1[dx] = S1[sx] * ONE;
            }
        }
        uint8_t *D = dst + (size_t)dy * dstep;
        for (int dx = 0; dx < dw; ++dx)
            D[dx] = (uint8_t)((((b0 * (row0[dx] >> 4)) >> 16) + ((b1 * (row1[dx] >> 4)) >> 16) + 2) >> 2);
    }
    free(xofs); free(alpha); free(row0); free(row1);
}

/* cv::GaussianBlur(src,dst,Size(7,7),2,2,BORDER_REFLECT_101) for 8UC1.
 * Integer kernel cvRound(g*256) = {18,34,49,55,49,34,18}; row pass keeps int32,
 * column pass (sum + 2^15) >> 16, saturate.  (OpenCV<=3.3 scalar path; see header) */
static const int g_blur_w[7] = {18, 34, 49, 55, 49, 34, 18};
void oracle_gauss7(const uint8_t *src, int sstep, int w, int h, uint8_t *dst, int dstep)
{
    int *tmp = (int *)malloc(sizeof(int) * (size_t)w * h);
    for (int y = 0; y < h; ++y) {
        const uint8_t *S = src + (size_t)y * sstep;
        for (int x = 0; x < w; ++x) {
            int acc = 0;
            for (int k = -3; k <= 3; ++k) acc += g_blur_w[k + 3] * S[reflect101(x + k, w)];
            tmp[(size_t)y * w + x] = acc;
        }
    }
    for (int y = 0; y < h; ++y) {
        uint8_t *D = dst + (size_t)y * dstep;
        for (int x = 0; x < w; ++x) {
            int acc = 0;
            for (int k = -3; k <= 3; ++k)
                acc += g_blur_w[k + 3] * tmp[(size_t)reflect101(y + k, h) * w + x];
            D[x] = sat_u8((acc + 32768) >> 16);
        }
    }
    free(tmp);
}

/* ------------------------------------------------------------------------ */
/* cv::FAST, TYPE_9_16 (FAST_t<16> + cornerScore<16>), restated               */
/* ------------------------------------------------------------------------ */
static const int fast_off16[16][2] = {
    {0, 3}, {1, 3}, {2, 2}, {3, 1}, {3, 0}, {3, -1}, {2, -2}, {1, -3},
    {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

static int fast_corner_score16(const uint8_t *ptr, const int *pixel, int threshold)
{
    const int K = 8, N = K * 3 + 1;
    int k, v = ptr[0];
    short d[25];
    for (k = 0; k < N; k++) d[k] = (short)(v - ptr[pixel[k]]);
    int a0 = threshold;
    for (k = 0; k < 16; k += 2) {
        int a = d[k + 1] < d[k + 2] ? d[k + 1] : d[k + 2];
        a = a < d[k + 3] ? a : d[k + 3];
        if (a <= a0) continue;
        for (int j = 4; j <= 8; ++j) a = a < d[k + j] ? a : d[k + j];
        int t = a < d[k] ? a : d[k];
        a0 = a0 > t ? a0 : t;
        t = a < d[k + 9] ? a : d[k + 9];
        a0 = a0 > t ? a0 : t;
    }
    int b0 = -a0;
    for (k = 0; k < 16; k += 2) {
        int b = d[k + 1] > d[k + 2] ? d[k + 1] : d[k + 2];
        for (int j = 3; j <= 5; ++j) b = b > d[k + j] ? b : d[k + j];
        if (b >= b0) continue;
        for (int j = 6; j <= 8; ++j) b = b > d[k + j] ? b : d[k + j];
        int t = b > d[k] ? b : d[k];
        b0 = b0 < t ? b0 : t;
        t = b > d[k + 9] ? b : d[k + 9];
        b0 = b0 < t ? b0 : t;
    }
    return -b0 - 1;
}

int oracle_fast(const uint8_t *img, int step, int w, int h, int threshold, int nonmax, int *ox,
                int *oy, int *oscore)
{
    const int K = 8, N = 25;
    int pixel[25];
    for (int k = 0; k < 16; ++k) pixel[k] = fast_off16[k][0] + fast_off16[k][1] * step;
    for (int k = 16; k < 25; ++k) pixel[k] = pixel[k - 16];
    if (threshold < 0) threshold = 0;
    if (threshold > 255) threshold = 255;
    if (w < 7 || h < 7) return 0;
    uint8_t *score = (uint8_t *)calloc((size_t)w * h, 1);
    uint8_t *is_corner = (uint8_t *)calloc((size_t)w * h, 1);
    for (int i = 3; i < h - 3; ++i) {
        const uint8_t *ptr = img + (size_t)i * step + 3;
        for (int j = 3; j < w - 3; ++j, ++ptr) {
            int v = ptr[0];
            int found = 0;
            { /* darker arc: 9 contiguous p < v - t */
                int vt = v - threshold, count = 0;
                for (int k = 0; k < N; ++k) {
                    if (ptr[pixel[k]] < vt) { if (++count > K) { found = 1; break; } }
                    else count = 0;
                }
            }
            if (!found) { /* brighter arc: 9 contiguous p > v + t */
                int vt = v + threshold, count = 0;
                for (int k = 0; k < N; ++k) {
                    if (ptr[pixel[k]] > vt) { if (++count > K) { found = 1; break; } }
                    else count = 0;
                }
            }
            if (found) {
                is_corner[(size_t)i * w + j] = 1;
                score[(size_t)i * w + j] = (uint8_t)fast_corner_score16(ptr, pixel, threshold);
            }
        }
    }
    int n = 0;
    for (int i = 3; i < h - 3; ++i)
        for (int j = 3; j < w - 3; ++j) {
            if (!is_corner[(size_t)i * w + j]) continue;
            int s = score[(size_t)i * w + j];
            const uint8_t *p = score + (size_t)i * w + j;
            if (!nonmax || (s > p[1] && s > p[-1] && s > p[-w - 1] && s > p[-w] && s > p[-w + 1] &&
                            s > p[w - 1] && s > p[w] && s > p[w + 1])) {
                ox[n] = j; oy[n] = i; oscore[n] = s; ++n;
            }
        }
    free(score); free(is_corner);
    return n;
}

/* ------------------------------------------------------------------------ */
/* DistributeOctTree (ORBextractor.cc:481-763), list-based restatement        */
/* ------------------------------------------------------------------------ */
typedef struct {
    int x0, x1, y0, y1;  /* UL.x, UR.x, UL.y, BR.y (all four corners stay a rectangle) */
    int *keys; int nkeys;
    int no_more;
    int prev, next;      /* list links (-1 = none) */
    int seq;             /* creation sequence: tie-break stand-in for the heap address */
    int alive;
} onode;

typedef struct {
    onode *n; int cap, cnt;
    int head, tail, size;
    int seq;
} olist;

static int ol_new(olist *L)
{
    if (L->cnt == L->cap) { L->cap = L->cap ? L->cap * 2 : 64; L->n = (onode *)realloc(L->n, sizeof(onode) * L->cap); }
    onode *nd = &L->n[L->cnt];
    memset(nd, 0, sizeof(*nd));
    nd->prev = nd->next = -1; nd->alive = 1; nd->seq = L->seq++;
    return L->cnt++;
}
static void ol_push_front(olist *L, int id)
{
    L->n[id].prev = -1; L->n[id].next = L->head;
    if (L->head >= 0) L->n[L->head].prev = id; else L->tail = id;
    L->head = id; L->size++;
}
static void ol_push_back(olist *L, int id)
{
    L->n[id].next = -1; L->n[id].prev = L->tail;
    if (L->tail >= 0) L->n[L->tail].next = id; else L->head = id;
    L->tail = id; L->size++;
}
static int ol_erase(olist *L, int id) /* returns next */
{
    int p = L->n[id].prev, nx = L->n[id].next;
    if (p >= 0) L->n[p].next = nx; else L->head = nx;
    if (nx >= 0) L->n[nx].prev = p; else L->tail = p;
    L->n[id].alive = 0; free(L->n[id].keys); L->n[id].keys = NULL;
    L->size--;
    return nx;
}

/* DivideNode :481-537; returns child ids in c[4] (allocated, not yet linked) */
static void divide_node(olist *L, int id, const float *kx, const float *ky, int c[4])
{
    for (int q = 0; q < 4; ++q) c[q] = ol_new(L); /* may realloc: re-read parent below */
    onode *P = &L->n[id];
    const int halfX = (int)ceilf((float)(P->x1 - P->x0) / 2);
    const int halfY = (int)ceilf((float)(P->y1 - P->y0) / 2);
    onode *n1 = &L->n[c[0]], *n2 = &L->n[c[1]], *n3 = &L->n[c[2]], *n4 = &L->n[c[3]];
    n1->x0 = P->x0; n1->x1 = P->x0 + halfX; n1->y0 = P->y0; n1->y1 = P->y0 + halfY;
    n2->x0 = P->x0 + halfX; n2->x1 = P->x1; n2->y0 = P->y0; n2->y1 = P->y0 + halfY;
    n3->x0 = P->x0; n3->x1 = P->x0 + halfX; n3->y0 = P->y0 + halfY; n3->y1 = P->y1;
    n4->x0 = P->x0 + halfX; n4->x1 = P->x1; n4->y0 = P->y0 + halfY; n4->y1 = P->y1;
    for (int q = 0; q < 4; ++q) L->n[c[q]].keys = (int *)malloc(sizeof(int) * (P->nkeys ? P->nkeys : 1));
    for (int i = 0; i < P->nkeys; ++i) {
        int k = P->keys[i];
        onode *dst;
        if (kx[k] < n1->x1) dst = (ky[k] < n1->y1) ? n1 : n3;
        else dst = (ky[k] < n1->y1) ? n2 : n4;
        dst->keys[dst->nkeys++] = k;
    }
    for (int q = 0; q < 4; ++q) if (L->n[c[q]].nkeys == 1) L->n[c[q]].no_more = 1;
}

typedef struct { int size, seq, id; } szptr;
static int szptr_cmp(const void *a, const void *b)
{
    const szptr *A = (const szptr *)a, *B = (const szptr *)b;
    if (A->size != B->size) return A->size < B->size ? -1 : 1;
    return A->seq < B->seq ? -1 : (A->seq > B->seq ? 1 : 0);
}

int oracle_distribute_octree(const float *x, const float *y, const float *resp, int n, int minX,
                             int maxX, int minY, int maxY, int N, int *out_idx, int out_cap)
{
    /* :543 round(), :545 */
    int nIni = (int)roundf((float)(maxX - minX) / (maxY - minY));
    if (n == 0) return 0;   /* reference: empty list -> empty result */
    if (nIni < 1) nIni = 1; /* portrait levels (w < h/2): the reference divides by zero and indexes an empty
                               vector (:545,:569); the documented choice is a single root node */
    const float hX = (float)(maxX - minX) / nIni;
    olist L; memset(&L, 0, sizeof(L)); L.head = L.tail = -1;
    int *ini = (int *)malloc(sizeof(int) * nIni);
    for (int i = 0; i < nIni; ++i) {
        int id = ol_new(&L);
        onode *nd = &L.n[id];
        nd->x0 = (int)(hX * (float)i); nd->x1 = (int)(hX * (float)(i + 1));
        nd->y0 = 0; nd->y1 = maxY - minY;
        nd->keys = (int *)malloc(sizeof(int) * n);
        ol_push_back(&L, id);
        ini[i] = id;
    }
    for (int i = 0; i < n; ++i) { /* :566-570 */
        int b = (int)(x[i] / hX);
        if (b >= nIni) b = nIni - 1; /* cannot happen for x < maxX-minX; guards the UB */
        onode *nd = &L.n[ini[b]];
        nd->keys[nd->nkeys++] = i;
    }
    free(ini);
    for (int it = L.head; it >= 0;) { /* :572-585 */
        if (L.n[it].nkeys == 1) { L.n[it].no_more = 1; it = L.n[it].next; }
        else if (L.n[it].nkeys == 0) it = ol_erase(&L, it);
        else it = L.n[it].next;
    }
    int finish = 0;
    szptr *vsp = NULL; int nvsp = 0, capvsp = 0;
#define VSP_PUSH(sz, idv) do { if (nvsp == capvsp) { capvsp = capvsp ? capvsp * 2 : 256; vsp = (szptr *)realloc(vsp, sizeof(szptr) * capvsp); } \
        vsp[nvsp].size = (sz); vsp[nvsp].seq = L.n[idv].seq; vsp[nvsp].id = (idv); ++nvsp; } while (0)
    while (!finish) {
        int prevSize = L.size;
        int nToExpand = 0;
        nvsp = 0;
        for (int it = L.head; it >= 0;) {
            if (L.n[it].no_more) { it = L.n[it].next; continue; }
            int c[4];
            divide_node(&L, it, x, y, c);
            for (int q = 0; q < 4; ++q) {
                if (L.n[c[q]].nkeys > 0) {
                    ol_push_front(&L, c[q]);
                    if (L.n[c[q]].nkeys > 1) { nToExpand++; VSP_PUSH(L.n[c[q]].nkeys, c[q]); }
                } else { L.n[c[q]].alive = 0; free(L.n[c[q]].keys); L.n[c[q]].keys = NULL; }
            }
            it = ol_erase(&L, it);
        }
        if (L.size >= N || L.size == prevSize) finish = 1;
        else if (L.size + nToExpand * 3 > N) {
            while (!finish) {
                prevSize = L.size;
                int nprev = nvsp;
                szptr *prev = (szptr *)malloc(sizeof(szptr) * (nprev ? nprev : 1));
                memcpy(prev, vsp, sizeof(szptr) * nprev);
                nvsp = 0;
                qsort(prev, nprev, sizeof(szptr), szptr_cmp); /* :684, (size, address) -> (size, seq) */
                for (int j = nprev - 1; j >= 0; --j) {
                    int c[4];
                    divide_node(&L, prev[j].id, x, y, c);
                    for (int q = 0; q < 4; ++q) {
                        if (L.n[c[q]].nkeys > 0) {
                            ol_push_front(&L, c[q]);
                            if (L.n[c[q]].nkeys > 1) VSP_PUSH(L.n[c[q]].nkeys, c[q]);
                        } else { L.n[c[q]].alive = 0; free(L.n[c[q]].keys); L.n[c[q]].keys = NULL; }
                    }
                    ol_erase(&L, prev[j].id);
                    if (L.size >= N) break;
                }
                free(prev);
                if (L.size >= N || L.size == prevSize) finish = 1;
            }
        }
    }
#undef VSP_PUSH
    int nout = 0;
    for (int it = L.head; it >= 0; it = L.n[it].next) { /* :742-760 */
        onode *nd = &L.n[it];
        int best = nd->keys[0];
        float maxr = resp[best];
        for (int k = 1; k < nd->nkeys; ++k)
            if (resp[nd->keys[k]] > maxr) { best = nd->keys[k]; maxr = resp[best]; }
        if (nout < out_cap) out_idx[nout] = best;
        ++nout;
    }
    for (int i = 0; i < L.cnt; ++i) free(L.n[i].keys);
    free(L.n); free(vsp);
    return nout;
}

/* ------------------------------------------------------------------------ */
/* extractor object                                                           */
/* ------------------------------------------------------------------------ */
struct oracle_extractor {
    int nfeatures; double scaleFactor; int nlevels, iniTh, minTh;
    float sf[ORACLE_MAX_LEVELS], isf[ORACLE_MAX_LEVELS], sig2[ORACLE_MAX_LEVELS], isig2[ORACLE_MAX_LEVELS];
    int nfeat[ORACLE_MAX_LEVELS];
    int umax[HALF_PATCH_SIZE + 1];
    /* last-call intermediates */
    int lw[ORACLE_MAX_LEVELS], lh[ORACLE_MAX_LEVELS];
    uint8_t *padded[ORACLE_MAX_LEVELS];
    uint8_t *blurred[ORACLE_MAX_LEVELS];
    float *cx[ORACLE_MAX_LEVELS], *cy[ORACLE_MAX_LEVELS], *cr[ORACLE_MAX_LEVELS];
    int ncand[ORACLE_MAX_LEVELS];
    oracle_kp *lk[ORACLE_MAX_LEVELS]; int nlk[ORACLE_MAX_LEVELS];
};

oracle_extractor *oracle_create(int nfeatures, float scale_factor, int nlevels, int ini_th, int min_th)
{
    if (nlevels < 1 || nlevels > ORACLE_MAX_LEVELS) return NULL;
    oracle_extractor *e = (oracle_extractor *)calloc(1, sizeof(*e));
    e->nfeatures = nfeatures; e->scaleFactor = scale_factor; e->nlevels = nlevels;
    e->iniTh = ini_th; e->minTh = min_th;
    e->sf[0] = 1.0f; e->sig2[0] = 1.0f;
    for (int i = 1; i < nlevels; ++i) { /* :419-423: float * double member */
        e->sf[i] = (float)(e->sf[i - 1] * e->scaleFactor);
        e->sig2[i] = e->sf[i] * e->sf[i];
    }
    for (int i = 0; i < nlevels; ++i) { e->isf[i] = 1.0f / e->sf[i]; e->isig2[i] = 1.0f / e->sig2[i]; }
    float factor = (float)(1.0f / e->scaleFactor); /* :436 */
    float nDesired = nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int l = 0; l < nlevels - 1; ++l) {
        e->nfeat[l] = oracle_cvround(nDesired);
        sum += e->nfeat[l];
        nDesired *= factor;
    }
    e->nfeat[nlevels - 1] = nfeatures - sum > 0 ? nfeatures - sum : 0;
    /* umax :454-469 */
    int v, v0, vmax = cv_floor(HALF_PATCH_SIZE * sqrtf(2.f) / 2 + 1);
    int vmin = cv_ceil(HALF_PATCH_SIZE * sqrtf(2.f) / 2);
    const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
    for (v = 0; v <= vmax; ++v) e->umax[v] = oracle_cvround(sqrt(hp2 - v * v));
    for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
        while (e->umax[v0] == e->umax[v0 + 1]) ++v0;
        e->umax[v] = v0;
        ++v0;
    }
    return e;
}

static void free_intermediates(oracle_extractor *e)
{
    for (int l = 0; l < ORACLE_MAX_LEVELS; ++l) {
        free(e->padded[l]); e->padded[l] = NULL;
        free(e->blurred[l]); e->blurred[l] = NULL;
        free(e->cx[l]); free(e->cy[l]); free(e->cr[l]); e->cx[l] = e->cy[l] = e->cr[l] = NULL;
        free(e->lk[l]); e->lk[l] = NULL;
        e->ncand[l] = e->nlk[l] = 0;
    }
}

void oracle_destroy(oracle_extractor *e)
{
    if (!e) return;
    free_intermediates(e);
    free(e);
}

void oracle_get_tables(const oracle_extractor *e, float *scale, float *inv_scale, float *sigma2,
                       float *inv_sigma2, int *feat_per_level, int *umax16)
{
    for (int i = 0; i < e->nlevels; ++i) {
        if (scale) scale[i] = e->sf[i];
        if (inv_scale) inv_scale[i] = e->isf[i];
        if (sigma2) sigma2[i] = e->sig2[i];
        if (inv_sigma2) inv_sigma2[i] = e->isig2[i];
        if (feat_per_level) feat_per_level[i] = e->nfeat[i];
    }
    if (umax16) for (int i = 0; i <= HALF_PATCH_SIZE; ++i) umax16[i] = e->umax[i];
}

/* IC_Angle :77-104 */
static float ic_angle(const uint8_t *center, int step, const int *u_max)
{
    int m_01 = 0, m_10 = 0;
    for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
    for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
        int v_sum = 0, d = u_max[v];
        for (int u = -d; u <= d; ++u) {
            int val_plus = center[u + v * step], val_minus = center[u - v * step];
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    return oracle_fast_atan2((float)m_01, (float)m_10);
}

/* Test hook: evaluate the descriptor steering with this platform's libm cosf/sinf instead of
 * det_sincos, to measure how far the deterministic choice is from a glibc build. */
static int g_use_libm_sincos = 0;
void oracle_use_libm_sincos(int on) { g_use_libm_sincos = on; }

/* computeOrbDescriptor :108-147 */
static void orb_descriptor(float kp_angle, const uint8_t *center, int step, uint8_t *desc)
{
    const float factorPI = (float)(3.14159265358979323846 / 180.f);
    float angle = kp_angle * factorPI;
    float a, b;
    if (g_use_libm_sincos) { a = cosf(angle); b = sinf(angle); }
    else oracle_det_sincos(angle, &a, &b);
    const signed char *pat = oracle_rbrief_pattern;
    for (int i = 0; i < 32; ++i, pat += 32) {
        int val = 0;
        for (int j = 0; j < 8; ++j) {
            float x0 = (float)pat[4 * j], y0 = (float)pat[4 * j + 1];
            float x1 = (float)pat[4 * j + 2], y1 = (float)pat[4 * j + 3];
            int t0 = center[oracle_cvround(x0 * b + y0 * a) * step + oracle_cvround(x0 * a - y0 * b)];
            int t1 = center[oracle_cvround(x1 * b + y1 * a) * step + oracle_cvround(x1 * a - y1 * b)];
            val |= (t0 < t1) << j;
        }
        desc[i] = (uint8_t)val;
    }
}

/* ComputePyramid :1107-1132 */
static void compute_pyramid(oracle_extractor *e, const uint8_t *img, int rows, int cols, int stride)
{
    for (int level = 0; level < e->nlevels; ++level) {
        float scale = e->isf[level];
        int w = oracle_cvround((float)cols * scale), h = oracle_cvround((float)rows * scale);
        e->lw[level] = w; e->lh[level] = h;
        int pw = w + 2 * EDGE_THRESHOLD, ph = h + 2 * EDGE_THRESHOLD;
        e->padded[level] = (uint8_t *)malloc((size_t)pw * ph);
        if (level != 0) {
            int pwp = e->lw[level - 1] + 2 * EDGE_THRESHOLD;
            const uint8_t *src = e->padded[level - 1] + (size_t)EDGE_THRESHOLD * pwp + EDGE_THRESHOLD;
            uint8_t *tmp = (uint8_t *)malloc((size_t)w * h);
            oracle_resize_linear(src, pwp, e->lw[level - 1], e->lh[level - 1], tmp, w, w, h);
            copy_make_border101(tmp, w, w, h, e->padded[level], pw, EDGE_THRESHOLD);
            free(tmp);
        } else {
            copy_make_border101(img, stride, w, h, e->padded[level], pw, EDGE_THRESHOLD);
        }
    }
}

/* ComputeKeyPointsOctTree :765-853 */
static void compute_keypoints_octree(oracle_extractor *e)
{
    const float W = 30;
    for (int level = 0; level < e->nlevels; ++level) {
        const int w = e->lw[level], h = e->lh[level], pw = w + 2 * EDGE_THRESHOLD;
        const uint8_t *roi = e->padded[level] + (size_t)EDGE_THRESHOLD * pw + EDGE_THRESHOLD;
        const int minBorderX = EDGE_THRESHOLD - 3, minBorderY = minBorderX;
        const int maxBorderX = w - EDGE_THRESHOLD + 3, maxBorderY = h - EDGE_THRESHOLD + 3;
        const float width = (float)(maxBorderX - minBorderX), height = (float)(maxBorderY - minBorderY);
        const int nCols = (int)(width / W), nRows = (int)(height / W);
        int capc = 0, nc = 0;
        float *cx = NULL, *cy = NULL, *cr = NULL;
        if (nCols >= 1 && nRows >= 1) {
            const int wCell = (int)ceilf(width / nCols), hCell = (int)ceilf(height / nRows);
            int maxpix = (wCell + 6) * (hCell + 6);
            int *ox = (int *)malloc(sizeof(int) * maxpix), *oy = (int *)malloc(sizeof(int) * maxpix),
                *os = (int *)malloc(sizeof(int) * maxpix);
            for (int i = 0; i < nRows; ++i) {
                const float iniY = (float)(minBorderY + i * hCell);
                float maxY = iniY + hCell + 6;
                if (iniY >= maxBorderY - 3) continue;
                if (maxY > maxBorderY) maxY = (float)maxBorderY;
                for (int j = 0; j < nCols; ++j) {
                    const float iniX = (float)(minBorderX + j * wCell);
                    float maxX = iniX + wCell + 6;
                    if (iniX >= maxBorderX - 6) continue;
                    if (maxX > maxBorderX) maxX = (float)maxBorderX;
                    int y0 = (int)iniY, y1 = (int)maxY, x0 = (int)iniX, x1 = (int)maxX;
                    const uint8_t *sub = roi + (size_t)y0 * pw + x0;
                    int cnt = oracle_fast(sub, pw, x1 - x0, y1 - y0, e->iniTh, 1, ox, oy, os);
                    if (cnt == 0) cnt = oracle_fast(sub, pw, x1 - x0, y1 - y0, e->minTh, 1, ox, oy, os);
                    for (int k = 0; k < cnt; ++k) {
                        if (nc == capc) {
                            capc = capc ? capc * 2 : 4096;
                            cx = (float *)realloc(cx, sizeof(float) * capc);
                            cy = (float *)realloc(cy, sizeof(float) * capc);
                            cr = (float *)realloc(cr, sizeof(float) * capc);
                        }
                        cx[nc] = (float)ox[k] + j * wCell;
                        cy[nc] = (float)oy[k] + i * hCell;
                        cr[nc] = (float)os[k];
                        ++nc;
                    }
                }
            }
            free(ox); free(oy); free(os);
        }
        e->cx[level] = cx; e->cy[level] = cy; e->cr[level] = cr; e->ncand[level] = nc;
        int capk = e->nfeat[level] + 8 > 8 ? e->nfeat[level] + 8 : 8;
        int *sel = (int *)malloc(sizeof(int) * (capk + nc + 1));
        int nsel = 0;
        if (nc > 0 && maxBorderY - minBorderY > 0)
            nsel = oracle_distribute_octree(cx, cy, cr, nc, minBorderX, maxBorderX, minBorderY, maxBorderY,
                                            e->nfeat[level], sel, capk + nc);
        const int scaledPatchSize = (int)(PATCH_SIZE * e->sf[level]);
        e->lk[level] = (oracle_kp *)malloc(sizeof(oracle_kp) * (nsel ? nsel : 1));
        e->nlk[level] = nsel;
        for (int i = 0; i < nsel; ++i) {
            oracle_kp *k = &e->lk[level][i];
            k->x = cx[sel[i]] + minBorderX; k->y = cy[sel[i]] + minBorderY;
            k->size = (float)scaledPatchSize; k->angle = -1; k->response = cr[sel[i]];
            k->octave = level; k->class_id = -1;
        }
        free(sel);
    }
    for (int level = 0; level < e->nlevels; ++level) { /* :851-852 */
        const int pw = e->lw[level] + 2 * EDGE_THRESHOLD;
        const uint8_t *roi = e->padded[level] + (size_t)EDGE_THRESHOLD * pw + EDGE_THRESHOLD;
        for (int i = 0; i < e->nlk[level]; ++i) {
            oracle_kp *k = &e->lk[level][i];
            const uint8_t *center = roi + (size_t)oracle_cvround(k->y) * pw + oracle_cvround(k->x);
            k->angle = ic_angle(center, pw, e->umax);
        }
    }
}

int oracle_extract(oracle_extractor *e, const uint8_t *img, int rows, int cols, int stride,
                   oracle_kp *kps, uint8_t *desc, int cap)
{
    free_intermediates(e);
    if (!img || rows <= 0 || cols <= 0) return 0; /* :1046 */
    compute_pyramid(e, img, rows, cols, stride);
    compute_keypoints_octree(e);
    int total = 0;
    for (int l = 0; l < e->nlevels; ++l) total += e->nlk[l];
    if (total > cap) return -1;
    int offset = 0;
    for (int level = 0; level < e->nlevels; ++level) {
        int nk = e->nlk[level];
        if (nk == 0) continue;
        const int w = e->lw[level], h = e->lh[level], pw = w + 2 * EDGE_THRESHOLD;
        const uint8_t *roi = e->padded[level] + (size_t)EDGE_THRESHOLD * pw + EDGE_THRESHOLD;
        e->blurred[level] = (uint8_t *)malloc((size_t)w * h);
        oracle_gauss7(roi, pw, w, h, e->blurred[level], w); /* :1085-1086 clone + blur */
        for (int i = 0; i < nk; ++i) {
            const oracle_kp *k = &e->lk[level][i];
            const uint8_t *center = e->blurred[level] + (size_t)oracle_cvround(k->y) * w + oracle_cvround(k->x);
            orb_descriptor(k->angle, center, w, desc + (size_t)(offset + i) * 32);
            kps[offset + i] = *k;
            if (level != 0) { /* :1095-1101 */
                float scale = e->sf[level];
                kps[offset + i].x *= scale; kps[offset + i].y *= scale;
            }
        }
        offset += nk;
    }
    return total;
}

int oracle_level_size(const oracle_extractor *e, int level, int *w, int *h)
{
    if (level < 0 || level >= e->nlevels || !e->padded[level]) return -1;
    *w = e->lw[level]; *h = e->lh[level];
    return 0;
}
const uint8_t *oracle_level_padded(const oracle_extractor *e, int level) { return e->padded[level]; }
const uint8_t *oracle_level_blurred(const oracle_extractor *e, int level) { return e->blurred[level]; }
int oracle_level_candidates(const oracle_extractor *e, int level, const float **x, const float **y, const float **resp)
{
    *x = e->cx[level]; *y = e->cy[level]; *resp = e->cr[level];
    return e->ncand[level];
}
int oracle_level_keypoints(const oracle_extractor *e, int level, const oracle_kp **kps)
{
    *kps = e->lk[level];
    return e->nlk[level];
}

/* ------------------------------------------------------------------------ */
/* matching                                                                   */
/* ------------------------------------------------------------------------ */
#define TH_HIGH 100
#define TH_LOW 50
#define HISTO_LENGTH 30
#define FRAME_GRID_ROWS 48
#define FRAME_GRID_COLS 64

/* DescriptorDistance, ORBmatcher.cc:1647-1663 (SWAR popcount over 8 x u32) */
int oracle_descriptor_distance(const uint8_t *a, const uint8_t *b)
{
    int dist = 0;
    for (int i = 0; i < 8; ++i) {
        uint32_t pa, pb;
        memcpy(&pa, a + 4 * i, 4); memcpy(&pb, b + 4 * i, 4);
        uint32_t v = pa ^ pb;
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}

/* ComputeThreeMaxima :1601-1642 */
void oracle_three_maxima(const int *hs, int L, int *ind1, int *ind2, int *ind3)
{
    int max1 = 0, max2 = 0, max3 = 0;
    *ind1 = *ind2 = *ind3 = -1;
    for (int i = 0; i < L; ++i) {
        const int s = hs[i];
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; *ind3 = *ind2; *ind2 = *ind1; *ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; *ind3 = *ind2; *ind2 = i; }
        else if (s > max3) { max3 = s; *ind3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { *ind2 = -1; *ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) { *ind3 = -1; }
}

/* PosInGrid Frame.cc:382-392 */
static int pos_in_grid(const oracle_frame *f, const oracle_kp *kp, int *px, int *py)
{
    *px = (int)roundf((kp->x - f->min_x) * f->grid_inv_w);
    *py = (int)roundf((kp->y - f->min_y) * f->grid_inv_h);
    return !(*px < 0 || *px >= FRAME_GRID_COLS || *py < 0 || *py >= FRAME_GRID_ROWS);
}

/* AssignFeaturesToGrid (Frame.cc:230-245): mGrid as CSR -- cell c = ix * FRAME_GRID_ROWS + iy holds the indices
 * items[start[c] .. start[c+1]) in insertion (= index) order.  The reference builds the grid once per Frame; the
 * search functions below do the same at entry. */
typedef struct { int32_t *start, *items; } oracle_grid;
static void grid_build(const oracle_frame *f, oracle_grid *g)
{
    const int ncell = FRAME_GRID_COLS * FRAME_GRID_ROWS;
    g->start = (int32_t *)calloc((size_t)ncell + 1, sizeof(int32_t));
    g->items = (int32_t *)malloc(sizeof(int32_t) * (size_t)(f->n > 0 ? f->n : 1));
    int32_t *cell = (int32_t *)malloc(sizeof(int32_t) * (size_t)(f->n > 0 ? f->n : 1));
    for (int j = 0; j < f->n; ++j) {
        int px, py;
        cell[j] = pos_in_grid(f, &f->keys[j], &px, &py) ? px * FRAME_GRID_ROWS + py : -1;
        if (cell[j] >= 0) g->start[cell[j] + 1]++;
    }
    for (int c = 0; c < ncell; ++c) g->start[c + 1] += g->start[c];
    int32_t *fill = (int32_t *)malloc(sizeof(int32_t) * (size_t)ncell);
    memcpy(fill, g->start, sizeof(int32_t) * (size_t)ncell);
    for (int j = 0; j < f->n; ++j) if (cell[j] >= 0) g->items[fill[cell[j]]++] = j;
    free(fill); free(cell);
}
static void grid_free(oracle_grid *g) { free(g->start); free(g->items); g->start = g->items = NULL; }

/* GetFeaturesInArea, Frame.cc:327-380 (cells ix outer / iy inner, insertion order inside a cell) */
static int features_in_area_grid(const oracle_frame *f, const oracle_grid *g, float x, float y, float r, int minLevel,
                                 int maxLevel, int32_t *out, int cap)
{
    int n = 0;
    const int nMinCellX = cv_floor((x - f->min_x - r) * f->grid_inv_w) > 0 ? cv_floor((x - f->min_x - r) * f->grid_inv_w) : 0;
    if (nMinCellX >= FRAME_GRID_COLS) return 0;
    int t = cv_ceil((x - f->min_x + r) * f->grid_inv_w);
    const int nMaxCellX = t < FRAME_GRID_COLS - 1 ? t : FRAME_GRID_COLS - 1;
    if (nMaxCellX < 0) return 0;
    const int nMinCellY = cv_floor((y - f->min_y - r) * f->grid_inv_h) > 0 ? cv_floor((y - f->min_y - r) * f->grid_inv_h) : 0;
    if (nMinCellY >= FRAME_GRID_ROWS) return 0;
    t = cv_ceil((y - f->min_y + r) * f->grid_inv_h);
    const int nMaxCellY = t < FRAME_GRID_ROWS - 1 ? t : FRAME_GRID_ROWS - 1;
    if (nMaxCellY < 0) return 0;
    const int bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    for (int ix = nMinCellX; ix <= nMaxCellX; ++ix)
        for (int iy = nMinCellY; iy <= nMaxCellY; ++iy) {
            const int c = ix * FRAME_GRID_ROWS + iy;
            for (int k = g->start[c]; k < g->start[c + 1]; ++k) {
                const int j = g->items[k];
                const oracle_kp *kp = &f->keys[j];
                if (bCheckLevels) {
                    if (kp->octave < minLevel) continue;
                    if (maxLevel >= 0 && kp->octave > maxLevel) continue;
                }
                const float distx = kp->x - x, disty = kp->y - y;
                if (fabsf(distx) < r && fabsf(disty) < r) { if (n < cap) out[n] = j; ++n; }
            }
        }
    return n;
}

int oracle_features_in_area(const oracle_frame *f, float x, float y, float r, int minLevel,
                            int maxLevel, int32_t *out, int cap)
{
    oracle_grid g;
    grid_build(f, &g);
    const int n = features_in_area_grid(f, &g, x, y, r, minLevel, maxLevel, out, cap);
    grid_free(&g);
    return n;
}

static int rot_bin(float a1, float a2)
{
    const float factor = 1.0f / HISTO_LENGTH;
    float rot = a1 - a2;
    if (rot < 0.0) rot += 360.0f;
    int bin = (int)roundf(rot * factor);
    if (bin == HISTO_LENGTH) bin = 0;
    return bin;
}

/* SearchForInitialization, ORBmatcher.cc:405-520 */
int oracle_search_for_initialization(const oracle_frame *f1, const oracle_frame *f2,
                                     float *prev_matched, int32_t *matches12, int windowSize,
                                     float nnratio, int check_ori)
{
    oracle_grid grid;
    grid_build(f2, &grid);
    int nmatches = 0;
    const int n1 = f1->n, n2 = f2->n;
    for (int i = 0; i < n1; ++i) matches12[i] = -1;
    int *hist = (int *)malloc(sizeof(int) * HISTO_LENGTH * (n1 + 1));
    int hs[HISTO_LENGTH] = {0};
    int *matchedDist = (int *)malloc(sizeof(int) * (n2 + 1));
    int *matches21 = (int *)malloc(sizeof(int) * (n2 + 1));
    int32_t *ind = (int32_t *)malloc(sizeof(int32_t) * (n2 + 1));
    for (int i = 0; i < n2; ++i) { matchedDist[i] = INT_MAX; matches21[i] = -1; }
    for (int i1 = 0; i1 < n1; ++i1) {
        const oracle_kp *kp1 = &f1->keys[i1];
        int level1 = kp1->octave;
        if (level1 > 0) continue;
        int nind = features_in_area_grid(f2, &grid, prev_matched[2 * i1], prev_matched[2 * i1 + 1],
                                           (float)windowSize, level1, level1, ind, n2);
        if (nind == 0) continue;
        const uint8_t *d1 = f1->desc + (size_t)i1 * 32;
        int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
        for (int c = 0; c < nind; ++c) {
            int i2 = ind[c];
            int dist = oracle_descriptor_distance(d1, f2->desc + (size_t)i2 * 32);
            if (matchedDist[i2] <= dist) continue;
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = i2; }
            else if (dist < bestDist2) bestDist2 = dist;
        }
        if (bestDist <= TH_LOW) {
            if (bestDist < (float)bestDist2 * nnratio) {
                if (matches21[bestIdx2] >= 0) { matches12[matches21[bestIdx2]] = -1; nmatches--; }
                matches12[i1] = bestIdx2;
                matches21[bestIdx2] = i1;
                matchedDist[bestIdx2] = bestDist;
                nmatches++;
                if (check_ori) {
                    int bin = rot_bin(f1->keys[i1].angle, f2->keys[bestIdx2].angle);
                    hist[bin * n1 + hs[bin]++] = i1;
                }
            }
        }
    }
    if (check_ori) {
        int ind1, ind2, ind3;
        oracle_three_maxima(hs, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < HISTO_LENGTH; ++i) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int j = 0; j < hs[i]; ++j) {
                int idx1 = hist[i * n1 + j];
                if (matches12[idx1] >= 0) { matches12[idx1] = -1; nmatches--; }
            }
        }
    }
    for (int i1 = 0; i1 < n1; ++i1)
        if (matches12[i1] >= 0) {
            prev_matched[2 * i1] = f2->keys[matches12[i1]].x;
            prev_matched[2 * i1 + 1] = f2->keys[matches12[i1]].y;
        }
    free(hist); free(matchedDist); free(matches21); free(ind);
    grid_free(&grid); return nmatches;
}

/* SearchByProjection(Frame&,const Frame&,th,bMono), ORBmatcher.cc:1351-1469
 * (the projection :1360-1390 is done by the caller and arrives in q[]) */
int oracle_search_by_projection_frame(const oracle_frame *cur, const oracle_query *q,
                                      const uint8_t *qdesc, int nq, const uint8_t *taken_in,
                                      int32_t *assign, int check_ori)
{
    oracle_grid grid;
    grid_build(cur, &grid);
    int nmatches = 0;
    const int n = cur->n;
    uint8_t *taken = (uint8_t *)malloc(n + 1);
    int32_t *ind = (int32_t *)malloc(sizeof(int32_t) * (n + 1));
    int *hist = (int *)malloc(sizeof(int) * HISTO_LENGTH * (nq + 1));
    int hs[HISTO_LENGTH] = {0};
    for (int i = 0; i < n; ++i) { taken[i] = taken_in ? taken_in[i] : 0; assign[i] = -1; }
    for (int i = 0; i < nq; ++i) {
        if (!q[i].valid) continue;
        const float u = q[i].u, v = q[i].v, radius = q[i].radius;
        int nind = features_in_area_grid(cur, &grid, u, v, radius, q[i].min_level, q[i].max_level, ind, n);
        if (nind == 0) continue;
        const uint8_t *dMP = qdesc + (size_t)i * 32;
        int bestDist = 256, bestIdx2 = -1;
        for (int c = 0; c < nind; ++c) {
            const int i2 = ind[c];
            if (taken[i2]) continue;
            if (cur->u_right && cur->u_right[i2] > 0) {
                const float er = fabsf(q[i].ur - cur->u_right[i2]);
                if (er > radius) continue;
            }
            const int dist = oracle_descriptor_distance(dMP, cur->desc + (size_t)i2 * 32);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
        if (bestDist <= TH_HIGH) {
            assign[bestIdx2] = i;
            taken[bestIdx2] = (uint8_t)(q[i].observed != 0);
            nmatches++;
            if (check_ori) {
                int bin = rot_bin(q[i].angle, cur->keys[bestIdx2].angle);
                hist[bin * nq + hs[bin]++] = bestIdx2;
            }
        }
    }
    if (check_ori) {
        int ind1, ind2, ind3;
        oracle_three_maxima(hs, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < HISTO_LENGTH; ++i)
            if (i != ind1 && i != ind2 && i != ind3)
                for (int j = 0; j < hs[i]; ++j) { assign[hist[i * nq + j]] = -1; nmatches--; }
    }
    free(taken); free(ind); free(hist);
    grid_free(&grid); return nmatches;
}

/* SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist), ORBmatcher.cc:1472-1599 after the projection
 * (:1490-1527 arrive in q[]); with max_dist = TH_LOW and check_ori = 0 it is also the matching loop of
 * SearchByProjection(pKF, Scw, vpPoints, vpMatched, th), :361-398 (same grid query, every match blocks its slot). */
int oracle_search_by_projection_block(const oracle_frame *cur, const oracle_query *q, const uint8_t *qdesc, int nq,
                                      const uint8_t *taken_in, int32_t *assign, int max_dist, int check_ori)
{
    oracle_grid grid;
    grid_build(cur, &grid);
    int nmatches = 0;
    const int n = cur->n;
    uint8_t *taken = (uint8_t *)malloc(n + 1);
    int32_t *ind = (int32_t *)malloc(sizeof(int32_t) * (n + 1));
    int *hist = (int *)malloc(sizeof(int) * HISTO_LENGTH * (nq + 1));
    int hs[HISTO_LENGTH] = {0};
    for (int i = 0; i < n; ++i) { taken[i] = taken_in ? taken_in[i] : 0; assign[i] = -1; }
    for (int i = 0; i < nq; ++i) {
        if (!q[i].valid) continue;
        int nind = features_in_area_grid(cur, &grid, q[i].u, q[i].v, q[i].radius, q[i].min_level, q[i].max_level, ind, n);
        if (nind == 0) continue;
        const uint8_t *dMP = qdesc + (size_t)i * 32;
        int bestDist = 256, bestIdx2 = -1;
        for (int c = 0; c < nind; ++c) {
            const int i2 = ind[c];
            if (taken[i2]) continue;                       /* :1538-1539 / :371-372 */
            const int dist = oracle_descriptor_distance(dMP, cur->desc + (size_t)i2 * 32);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
        if (bestDist <= max_dist) {                        /* :1552 / :393 */
            assign[bestIdx2] = i;
            taken[bestIdx2] = 1;
            nmatches++;
            if (check_ori) {
                int bin = rot_bin(q[i].angle, cur->keys[bestIdx2].angle);
                hist[bin * nq + hs[bin]++] = bestIdx2;
            }
        }
    }
    if (check_ori) {
        int ind1, ind2, ind3;
        oracle_three_maxima(hs, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < HISTO_LENGTH; ++i)
            if (i != ind1 && i != ind2 && i != ind3)
                for (int j = 0; j < hs[i]; ++j) { assign[hist[i * nq + j]] = -1; nmatches--; }
    }
    free(taken); free(ind); free(hist);
    grid_free(&grid); return nmatches;
}

/* Search loop of Fuse (ORBmatcher.cc:893-950; the Sim3 overload :1045-1075 has no chi2 gate) and of SearchBySim3
 * (:1199-1219, :1279-1299): best candidate per query, no blocking.  inv_sigma2 == NULL disables the chi2 gate. */
void oracle_search_best_in_window(const oracle_frame *kf, const oracle_query *q, const uint8_t *qdesc, int nq,
                                  const float *inv_sigma2, int32_t *best_idx, int32_t *best_dist)
{
    oracle_grid grid;
    grid_build(kf, &grid);
    const int n = kf->n;
    int32_t *ind = (int32_t *)malloc(sizeof(int32_t) * (n + 1));
    for (int i = 0; i < nq; ++i) {
        best_idx[i] = -1; best_dist[i] = 256;
        if (!q[i].valid) continue;
        const float u = q[i].u, v = q[i].v, ur = q[i].ur;
        int nind = features_in_area_grid(kf, &grid, u, v, q[i].radius, -1, -1, ind, n);  /* KeyFrame::GetFeaturesInArea */
        int bestDist = 256, bestIdx = -1;
        for (int c = 0; c < nind; ++c) {
            const int idx = ind[c];
            const oracle_kp *kp = &kf->keys[idx];
            const int kpLevel = kp->octave;
            if (kpLevel < q[i].min_level || kpLevel > q[i].max_level) continue;
            if (inv_sigma2) {
                if (kf->u_right && kf->u_right[idx] >= 0) {
                    const float ex = u - kp->x, ey = v - kp->y, er = ur - kf->u_right[idx];
                    const float e2 = ex * ex + ey * ey + er * er;
                    if (e2 * inv_sigma2[kpLevel] > 7.8) continue;
                } else {
                    const float ex = u - kp->x, ey = v - kp->y;
                    const float e2 = ex * ex + ey * ey;
                    if (e2 * inv_sigma2[kpLevel] > 5.99) continue;
                }
            }
            const int dist = oracle_descriptor_distance(qdesc + (size_t)i * 32, kf->desc + (size_t)idx * 32);
            if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
        }
        best_idx[i] = bestIdx; best_dist[i] = bestDist;
    }
    free(ind);
    grid_free(&grid);
}

/* SearchByProjection(Frame&,const vector<MapPoint*>&,th), ORBmatcher.cc:45-129 */
int oracle_search_by_projection_points(const oracle_frame *f, const oracle_query *q,
                                       const uint8_t *qdesc, int nq, const uint8_t *taken_in,
                                       int32_t *assign, float nnratio)
{
    oracle_grid grid;
    grid_build(f, &grid);
    int nmatches = 0;
    const int n = f->n;
    uint8_t *taken = (uint8_t *)malloc(n + 1);
    int32_t *ind = (int32_t *)malloc(sizeof(int32_t) * (n + 1));
    for (int i = 0; i < n; ++i) { taken[i] = taken_in ? taken_in[i] : 0; assign[i] = -1; }
    for (int iMP = 0; iMP < nq; ++iMP) {
        if (!q[iMP].valid) continue;
        const float r = q[iMP].radius;
        int nind = features_in_area_grid(f, &grid, q[iMP].u, q[iMP].v, r, q[iMP].min_level, q[iMP].max_level, ind, n);
        if (nind == 0) continue;
        const uint8_t *dMP = qdesc + (size_t)iMP * 32;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int c = 0; c < nind; ++c) {
            const int idx = ind[c];
            if (taken[idx]) continue;
            if (f->u_right && f->u_right[idx] > 0) {
                const float er = fabsf(q[iMP].ur - f->u_right[idx]);
                if (er > r) continue;
            }
            const int dist = oracle_descriptor_distance(dMP, f->desc + (size_t)idx * 32);
            if (dist < bestDist) {
                bestDist2 = bestDist; bestDist = dist;
                bestLevel2 = bestLevel; bestLevel = f->keys[idx].octave; bestIdx = idx;
            } else if (dist < bestDist2) {
                bestLevel2 = f->keys[idx].octave; bestDist2 = dist;
            }
        }
        if (bestDist <= TH_HIGH) {
            if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;
            assign[bestIdx] = iMP;
            taken[bestIdx] = (uint8_t)(q[iMP].observed != 0);
            nmatches++;
        }
    }
    free(taken); free(ind);
    grid_free(&grid); return nmatches;
}

/* ---- vocabulary-node guided searches ------------------------------------------------------
 * DBoW2::FeatureVector (Thirdparty/DBoW2/DBoW2/FeatureVector.h) is a std::map<NodeId, vector<feature index>>;
 * DBoW2 fills it while walking the features in index order, so every node's list is ascending.  The vocabulary
 * itself is outside the path: callers pass the node id of every keypoint (ORACLE_NO_NODE = not in the vector). */
typedef struct { uint32_t node; int32_t idx; } nodeidx;
static int nodeidx_cmp(const void *a, const void *b)
{
    const nodeidx *A = (const nodeidx *)a, *B = (const nodeidx *)b;
    if (A->node != B->node) return A->node < B->node ? -1 : 1;
    return A->idx < B->idx ? -1 : (A->idx > B->idx);
}
static nodeidx *feature_vector(const uint32_t *node, int n, int *count)
{
    nodeidx *v = (nodeidx *)malloc(sizeof(nodeidx) * (n + 1));
    int m = 0;
    for (int i = 0; i < n; ++i)
        if (node[i] != ORACLE_NO_NODE) { v[m].node = node[i]; v[m].idx = i; ++m; }
    qsort(v, m, sizeof(nodeidx), nodeidx_cmp);
    *count = m;
    return v;
}
static int node_end(const nodeidx *v, int m, int b)
{
    int e = b;
    while (e < m && v[e].node == v[b].node) ++e;
    return e;
}

/* SearchByBoW(KeyFrame*,Frame&,vpMapPointMatches) ORBmatcher.cc:159-288 (max_dist = TH_LOW) and
 * SearchByBoW(KeyFrame*,KeyFrame*,vpMatches12) :522-655 (strict "<TH_LOW": max_dist = TH_LOW-1) */
int oracle_search_by_bow(const oracle_frame *f1, const uint32_t *node1, const uint8_t *valid1,
                         const oracle_frame *f2, const uint32_t *node2, const uint8_t *blocked2,
                         int max_dist, float nnratio, int check_ori, int32_t *matches12)
{
    const int n1 = f1->n, n2 = f2->n;
    int m1, m2, nmatches = 0;
    nodeidx *v1 = feature_vector(node1, n1, &m1), *v2 = feature_vector(node2, n2, &m2);
    uint8_t *matched2 = (uint8_t *)calloc(n2 + 1, 1);
    int *hist = (int *)malloc(sizeof(int) * HISTO_LENGTH * (n1 + 1));
    int hs[HISTO_LENGTH] = {0};
    for (int i = 0; i < n1; ++i) matches12[i] = -1;
    int a = 0, b = 0;
    while (a < m1 && b < m2) {
        if (v1[a].node == v2[b].node) {
            const int ae = node_end(v1, m1, a), be = node_end(v2, m2, b);
            for (int i1 = a; i1 < ae; ++i1) {
                const int idx1 = v1[i1].idx;
                if (valid1 && !valid1[idx1]) continue;          /* !pMP || pMP->isBad() */
                const uint8_t *d1 = f1->desc + (size_t)idx1 * 32;
                int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
                for (int i2 = b; i2 < be; ++i2) {
                    const int idx2 = v2[i2].idx;
                    if (matched2[idx2] || (blocked2 && blocked2[idx2])) continue;
                    const int dist = oracle_descriptor_distance(d1, f2->desc + (size_t)idx2 * 32);
                    if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx2 = idx2; }
                    else if (dist < bestDist2) bestDist2 = dist;
                }
                if (bestDist1 <= max_dist) {
                    if ((float)bestDist1 < nnratio * (float)bestDist2) {
                        matches12[idx1] = bestIdx2;
                        matched2[bestIdx2] = 1;
                        if (check_ori) {
                            int bin = rot_bin(f1->keys[idx1].angle, f2->keys[bestIdx2].angle);
                            hist[bin * n1 + hs[bin]++] = idx1;
                        }
                        nmatches++;
                    }
                }
            }
            a = ae; b = be;
        } else if (v1[a].node < v2[b].node) {
            while (a < m1 && v1[a].node < v2[b].node) ++a;      /* lower_bound */
        } else {
            while (b < m2 && v2[b].node < v1[a].node) ++b;
        }
    }
    if (check_ori) {
        int ind1, ind2, ind3;
        oracle_three_maxima(hs, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < HISTO_LENGTH; ++i)
            if (i != ind1 && i != ind2 && i != ind3)
                for (int j = 0; j < hs[i]; ++j) { matches12[hist[i * n1 + j]] = -1; nmatches--; }
    }
    free(v1); free(v2); free(matched2); free(hist);
    return nmatches;
}

/* CheckDistEpipolarLine, ORBmatcher.cc:140-157 (F12 row-major 3x3, float) */
static int check_dist_epipolar_line(const oracle_kp *kp1, const oracle_kp *kp2, const float *F12,
                                    const float *level_sigma2)
{
    const float a = kp1->x * F12[0] + kp1->y * F12[3] + F12[6];
    const float b = kp1->x * F12[1] + kp1->y * F12[4] + F12[7];
    const float c = kp1->x * F12[2] + kp1->y * F12[5] + F12[8];
    const float num = a * kp2->x + b * kp2->y + c;
    const float den = a * a + b * b;
    if (den == 0) return 0;
    const float dsqr = num * num / den;
    return dsqr < 3.84 * level_sigma2[kp2->octave];
}

/* SearchForTriangulation, ORBmatcher.cc:657-823.  (ex,ey): epipole in the second image (:664-670, computed by the
 * caller); valid1[i] = key frame 1 has no map point at i, valid2 likewise.  vbMatched2 is never set by the reference,
 * so there is no blocking between queries. */
int oracle_search_for_triangulation(const oracle_frame *f1, const uint32_t *node1, const uint8_t *valid1,
                                    const oracle_frame *f2, const uint32_t *node2, const uint8_t *valid2,
                                    const float *F12, float ex, float ey, const float *level_sigma2,
                                    int only_stereo, int check_ori, int32_t *matches12)
{
    const int n1 = f1->n, n2 = f2->n;
    int m1, m2, nmatches = 0;
    nodeidx *v1 = feature_vector(node1, n1, &m1), *v2 = feature_vector(node2, n2, &m2);
    int *hist = (int *)malloc(sizeof(int) * HISTO_LENGTH * (n1 + 1));
    int hs[HISTO_LENGTH] = {0};
    for (int i = 0; i < n1; ++i) matches12[i] = -1;
    int a = 0, b = 0;
    while (a < m1 && b < m2) {
        if (v1[a].node == v2[b].node) {
            const int ae = node_end(v1, m1, a), be = node_end(v2, m2, b);
            for (int i1 = a; i1 < ae; ++i1) {
                const int idx1 = v1[i1].idx;
                if (valid1 && !valid1[idx1]) continue;
                const int bStereo1 = f1->u_right ? f1->u_right[idx1] >= 0 : 0;
                if (only_stereo && !bStereo1) continue;
                const oracle_kp *kp1 = &f1->keys[idx1];
                const uint8_t *d1 = f1->desc + (size_t)idx1 * 32;
                int bestDist = TH_LOW, bestIdx2 = -1;
                for (int i2 = b; i2 < be; ++i2) {
                    const int idx2 = v2[i2].idx;
                    if (valid2 && !valid2[idx2]) continue;
                    const int bStereo2 = f2->u_right ? f2->u_right[idx2] >= 0 : 0;
                    if (only_stereo && !bStereo2) continue;
                    const int dist = oracle_descriptor_distance(d1, f2->desc + (size_t)idx2 * 32);
                    if (dist > TH_LOW || dist > bestDist) continue;
                    const oracle_kp *kp2 = &f2->keys[idx2];
                    if (!bStereo1 && !bStereo2) {
                        const float distex = ex - kp2->x, distey = ey - kp2->y;
                        if (distex * distex + distey * distey < 100 * f2->scale_factors[kp2->octave]) continue;
                    }
                    if (check_dist_epipolar_line(kp1, kp2, F12, level_sigma2)) { bestIdx2 = idx2; bestDist = dist; }
                }
                if (bestIdx2 >= 0) {
                    matches12[idx1] = bestIdx2;
                    nmatches++;
                    if (check_ori) {
                        int bin = rot_bin(kp1->angle, f2->keys[bestIdx2].angle);
                        hist[bin * n1 + hs[bin]++] = idx1;
                    }
                }
            }
            a = ae; b = be;
        } else if (v1[a].node < v2[b].node) {
            while (a < m1 && v1[a].node < v2[b].node) ++a;
        } else {
            while (b < m2 && v2[b].node < v1[a].node) ++b;
        }
    }
    if (check_ori) {
        int ind1, ind2, ind3;
        oracle_three_maxima(hs, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < HISTO_LENGTH; ++i)
            if (i != ind1 && i != ind2 && i != ind3)
                for (int j = 0; j < hs[i]; ++j) { matches12[hist[i * n1 + j]] = -1; nmatches--; }
    }
    free(v1); free(v2); free(hist);
    return nmatches;
}

/* Frame::ComputeStereoMatches, Frame.cc:466-640 */
typedef struct { int dist, idx; } distidx;
static int distidx_cmp(const void *a, const void *b)
{
    const distidx *A = (const distidx *)a, *B = (const distidx *)b;
    if (A->dist != B->dist) return A->dist < B->dist ? -1 : 1;
    return A->idx < B->idx ? -1 : (A->idx > B->idx);
}

int oracle_compute_stereo_matches(const oracle_kp *kl, const uint8_t *dl, int N, const oracle_kp *kr,
                                  const uint8_t *dr, int Nr, const oracle_pyramids *pyr, int nRows,
                                  float mbf, float mb, float *uRight, float *depth)
{
    for (int i = 0; i < N; ++i) { uRight[i] = -1.0f; depth[i] = -1.0f; }
    const int thOrbDist = (TH_HIGH + TH_LOW) / 2;
    /* row table :476-493 (candidate order within a row = iR ascending) */
    int *rowCnt = (int *)calloc(nRows + 1, sizeof(int));
    int **rows = (int **)calloc(nRows + 1, sizeof(int *));
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1) for (int i = 0; i < nRows; ++i) { rows[i] = (int *)malloc(sizeof(int) * (rowCnt[i] + 1)); rowCnt[i] = 0; }
        for (int iR = 0; iR < Nr; ++iR) {
            const float kpY = kr[iR].y;
            const float r = 2.0f * pyr->scale_factors[kr[iR].octave];
            const int maxr = (int)ceilf(kpY + r), minr = (int)floorf(kpY - r);
            for (int yi = minr; yi <= maxr; ++yi) {
                if (yi < 0 || yi >= nRows) continue; /* reference would write out of bounds */
                if (pass == 1) rows[yi][rowCnt[yi]] = iR;
                rowCnt[yi]++;
            }
        }
    }
    const float minZ = mb, minD = 0, maxD = mbf / minZ;
    distidx *vDistIdx = (distidx *)malloc(sizeof(distidx) * (N + 1));
    int nd = 0;
    for (int iL = 0; iL < N; ++iL) {
        const oracle_kp *kpL = &kl[iL];
        const int levelL = kpL->octave;
        const float vL = kpL->y, uL = kpL->x;
        const int row = (int)vL;
        if (row < 0 || row >= nRows || rowCnt[row] == 0) continue;
        const float minU = uL - maxD, maxU = uL - minD;
        if (maxU < 0) continue;
        int bestDist = TH_HIGH, bestIdxR = 0;
        const uint8_t *dL = dl + (size_t)iL * 32;
        for (int iC = 0; iC < rowCnt[row]; ++iC) {
            const int iR = rows[row][iC];
            const oracle_kp *kpR = &kr[iR];
            if (kpR->octave < levelL - 1 || kpR->octave > levelL + 1) continue;
            const float uR = kpR->x;
            if (uR >= minU && uR <= maxU) {
                const int dist = oracle_descriptor_distance(dL, dr + (size_t)iR * 32);
                if (dist < bestDist) { bestDist = dist; bestIdxR = iR; }
            }
        }
        if (bestDist < thOrbDist) {
            const float uR0 = kr[bestIdxR].x;
            const float scaleFactor = pyr->inv_scale_factors[kpL->octave];
            const float scaleduL = roundf(kpL->x * scaleFactor);
            const float scaledvL = roundf(kpL->y * scaleFactor);
            const float scaleduR0 = roundf(uR0 * scaleFactor);
            const int w = 5, L = 5;
            const uint8_t *imL = pyr->left[kpL->octave], *imR = pyr->right[kpL->octave];
            const int stL = pyr->step_left[kpL->octave], stR = pyr->step_right[kpL->octave];
            const int cu = (int)scaleduL, cv = (int)scaledvL, cr = (int)scaleduR0;
            int bestD = INT_MAX, bestincR = 0;
            float vDists[11];
            const float iniu = scaleduR0 + L - w, endu = scaleduR0 + L + w + 1;
            if (iniu < 0 || endu >= pyr->cols_right[kpL->octave]) continue;
            const int cL = imL[(size_t)cv * stL + cu];
            for (int incR = -L; incR <= L; ++incR) {
                const int cR = imR[(size_t)cv * stR + cr + incR];
                /* cv::norm(IL,IR,NORM_L1) of centre-subtracted float patches: all
                 * terms are small integers, so the sum is exact */
                int sad = 0;
                for (int dy = -w; dy <= w; ++dy)
                    for (int dx = -w; dx <= w; ++dx) {
                        int a = imL[(size_t)(cv + dy) * stL + cu + dx] - cL;
                        int b = imR[(size_t)(cv + dy) * stR + cr + incR + dx] - cR;
                        sad += abs(a - b);
                    }
                float dist = (float)sad;
                if (dist < bestD) { bestD = (int)dist; bestincR = incR; }
                vDists[L + incR] = dist;
            }
            if (bestincR == -L || bestincR == L) continue;
            const float dist1 = vDists[L + bestincR - 1], dist2 = vDists[L + bestincR], dist3 = vDists[L + bestincR + 1];
            const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
            if (deltaR < -1 || deltaR > 1) continue;
            float bestuR = pyr->scale_factors[kpL->octave] * ((float)scaleduR0 + (float)bestincR + deltaR);
            float disparity = (uL - bestuR);
            if (disparity >= minD && disparity < maxD) {
                if (disparity <= 0) { disparity = 0.01; bestuR = uL - 0.01; }
                depth[iL] = mbf / disparity;
                uRight[iL] = bestuR;
                vDistIdx[nd].dist = bestD; vDistIdx[nd].idx = iL; ++nd;
            }
        }
    }
    if (nd > 0) { /* :626-639 (the reference reads vDistIdx[0] of an empty vector: UB) */
        qsort(vDistIdx, nd, sizeof(distidx), distidx_cmp);
        const float median = (float)vDistIdx[nd / 2].dist;
        const float thDist = 1.5f * 1.4f * median;
        for (int i = nd - 1; i >= 0; --i) {
            if (vDistIdx[i].dist < thDist) break;
            uRight[vDistIdx[i].idx] = -1; depth[vDistIdx[i].idx] = -1;
        }
    }
    int nm = 0;
    for (int i = 0; i < N; ++i) nm += depth[i] > 0;
    for (int i = 0; i < nRows; ++i) free(rows[i]);
    free(rows); free(rowCnt); free(vDistIdx);
    return nm;
}

/* MapPoint::ComputeDistinctiveDescriptors, MapPoint.cc:272-307: all-pairs Hamming distances between the N
 * observations of a map point; the descriptor with the least median distance (vDists[0.5*(N-1)] of the sorted row,
 * self distance included) wins, first minimum on ties.  Returns the index or -1 for N == 0. */
static int int_cmp(const void *a, const void *b) { return *(const int *)a - *(const int *)b; }
int oracle_distinctive_descriptor(const uint8_t *desc, int N)
{
    if (N <= 0) return -1;
    int *row = (int *)malloc(sizeof(int) * (size_t)N);
    int BestMedian = INT_MAX, BestIdx = 0;
    for (int i = 0; i < N; ++i) {
        for (int j = 0; j < N; ++j)
            row[j] = i == j ? 0 : oracle_descriptor_distance(desc + (size_t)i * 32, desc + (size_t)j * 32);
        qsort(row, N, sizeof(int), int_cmp);
        const int median = row[(int)(0.5 * (N - 1))];
        if (median < BestMedian) { BestMedian = median; BestIdx = i; }
    }
    free(row);
    return BestIdx;
}

/* Frame::AssignFeaturesToGrid, Frame.cc:230-245 with PosInGrid :382-392.  mGrid[64][48] as CSR in the order
 * GetFeaturesInArea walks it: cell c = posX*48 + posY, cell_start[c] .. cell_start[c+1] index cell_items, items in
 * push_back (= ascending keypoint index) order; cell_of[i] = c or -1 when PosInGrid rejects the keypoint. */
void oracle_assign_features_to_grid(const oracle_frame *f, int32_t *cell_of, int32_t *cell_start, int32_t *cell_items)
{
    const int ncell = FRAME_GRID_COLS * FRAME_GRID_ROWS;
    int *cnt = (int *)calloc(ncell + 1, sizeof(int));
    for (int i = 0; i < f->n; ++i) {
        const oracle_kp *kp = &f->keys[i];
        const int posX = (int)roundf((kp->x - f->min_x) * f->grid_inv_w);
        const int posY = (int)roundf((kp->y - f->min_y) * f->grid_inv_h);
        if (posX < 0 || posX >= FRAME_GRID_COLS || posY < 0 || posY >= FRAME_GRID_ROWS) { cell_of[i] = -1; continue; }
        cell_of[i] = posX * FRAME_GRID_ROWS + posY;
        cnt[cell_of[i]]++;
    }
    cell_start[0] = 0;
    for (int c = 0; c < ncell; ++c) cell_start[c + 1] = cell_start[c] + cnt[c];
    memset(cnt, 0, sizeof(int) * ncell);
    for (int i = 0; i < f->n; ++i)
        if (cell_of[i] >= 0) cell_items[cell_start[cell_of[i]] + cnt[cell_of[i]]++] = i;
    free(cnt);
}

/* Frame::ComputeStereoFromRGBD, Frame.cc:643-664: depth image CV_32F sampled at the DISTORTED keypoint (float
 * coordinates truncated by Mat::at<float>(int,int)); mvuRight from the undistorted x. */
void oracle_compute_stereo_from_rgbd(const oracle_kp *keys, const oracle_kp *keys_un, int n, const float *depth,
                                     int stride_floats, float mbf, float *u_right, float *depth_out)
{
    for (int i = 0; i < n; ++i) {
        u_right[i] = -1.0f; depth_out[i] = -1.0f;
        const float v = keys[i].y, u = keys[i].x;
        const float d = depth[(size_t)(int)v * stride_floats + (int)u];
        if (d > 0) {
            depth_out[i] = d;
            u_right[i] = keys_un[i].x - mbf / d;
        }
    }
}

/* Frame::UndistortKeyPoints, Frame.cc:404-434: cv::undistortPoints(mat, mat, mK, mDistCoef, Mat(), mK) on the keypoint
 * coordinates; everything else of the KeyPoint is copied.  cv::undistortPoints is restated from OpenCV's published
 * algorithm (cvUndistortPoints of OpenCV 2.4 - 3.3, imgproc/undistort.cpp; OpenCV is not on disk): float in, double
 * arithmetic, normalise with the camera matrix, 5 fixed-point iterations of the inverse Brown model, re-project with
 * P = mK, float out.  dist = {k1, k2, p1, p2, k3} (mDistCoef, Tracking.cc:66-81); dist[0] == 0 copies the input
 * (:406-410).  Parity unpinned like every other OpenCV restatement. */
void oracle_undistort_keypoints(const oracle_kp *keys, int n, float fxf, float fyf, float cxf, float cyf,
                                const float *dist, oracle_kp *keys_un)
{
    for (int i = 0; i < n; ++i) keys_un[i] = keys[i];
    if (dist[0] == 0.0f) return;
    const double fx = fxf, fy = fyf, cx = cxf, cy = cyf, ifx = 1. / fx, ify = 1. / fy;
    const double k0 = dist[0], k1 = dist[1], k2 = dist[2], k3 = dist[3], k4 = dist[4];
    const double k5 = 0, k6 = 0, k7 = 0;   /* rational terms: absent from mDistCoef */
    for (int i = 0; i < n; ++i) {
        double x = keys[i].x, y = keys[i].y;
        const double x0 = x = (x - cx) * ifx;
        const double y0 = y = (y - cy) * ify;
        for (int j = 0; j < 5; ++j) {
            const double r2 = x * x + y * y;
            const double icdist = (1 + ((k7 * r2 + k6) * r2 + k5) * r2) / (1 + ((k4 * r2 + k1) * r2 + k0) * r2);
            const double deltaX = 2 * k2 * x * y + k3 * (r2 + 2 * x * x);
            const double deltaY = k2 * (r2 + 2 * y * y) + 2 * k3 * x * y;
            x = (x0 - deltaX) * icdist;
            y = (y0 - deltaY) * icdist;
        }
        /* RR = P * I with P = mK = [fx 0 cx; 0 fy cy; 0 0 1] */
        const double xx = fx * x + 0.0 * y + cx;
        const double yy = 0.0 * x + fy * y + cy;
        const double ww = 1. / (0.0 * x + 0.0 * y + 1.0);
        keys_un[i].x = (float)(xx * ww);
        keys_un[i].y = (float)(yy * ww);
    }
}

/* ---- DBoW2 vocabulary: loadFromTextFile + transform --------------------------------------------------------
 * Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1338-1424 (text format), :1127-1199 (transform of a feature set),
 * :1218-1262 (descent of one feature), BowVector.cpp:36-88 (addWeight / addIfNotExist / normalize),
 * FeatureVector.cpp:31-46, FORB.cpp:81-101 (Hamming distance).  Deviations, both where the reference's behaviour is
 * undefined: (1) blank lines are skipped -- the reference's `while(!f.eof())` loop turns the empty line after the
 * last '\n' into one more child of the root whose descriptor is uninitialised memory; (2) when a leaf is reached
 * above level L-levelsup the reference leaves *nid unset; here the leaf's own id is reported. */
typedef struct {
    int parent, is_leaf_flag, nchild, cap;
    int *children;
    uint8_t desc[32];
    double weight;
    uint32_t word_id;
} voc_node;
struct oracle_vocabulary {
    int k, L, scoring, weighting, n_nodes, n_words;
    voc_node *nodes;
};

void oracle_vocabulary_destroy(oracle_vocabulary *v)
{
    if (!v) return;
    for (int i = 0; i < v->n_nodes; ++i) free(v->nodes[i].children);
    free(v->nodes);
    free(v);
}

static int voc_push_node(oracle_vocabulary *v, int *cap)
{
    if (v->n_nodes == *cap) {
        *cap = *cap * 2 + 16;
        v->nodes = (voc_node *)realloc(v->nodes, sizeof(voc_node) * (size_t)*cap);
    }
    memset(&v->nodes[v->n_nodes], 0, sizeof(voc_node));
    return v->n_nodes++;
}

oracle_vocabulary *oracle_vocabulary_load_text(const char *path)
{
    FILE *f = fopen(path, "r");
    if (!f) return NULL;
    oracle_vocabulary *v = (oracle_vocabulary *)calloc(1, sizeof(*v));
    size_t lcap = 1 << 12;
    char *line = (char *)malloc(lcap);
    int cap = 0, ok = 0;
    if (fgets(line, (int)lcap, f)) {
        int n1 = -1, n2 = -1;
        v->k = -1; v->L = -1;
        if (sscanf(line, "%d %d %d %d", &v->k, &v->L, &n1, &n2) == 4 &&
            !(v->k < 0 || v->k > 20 || v->L < 1 || v->L > 10 || n1 < 0 || n1 > 5 || n2 < 0 || n2 > 3)) {
            v->scoring = n1; v->weighting = n2;
            ok = 1;
        }
    }
    if (ok) {
        voc_push_node(v, &cap); /* root, id 0 */
        while (fgets(line, (int)lcap, f)) {
            char *p = line, *e;
            while (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\n') ++p;
            if (!*p) continue;
            const int nid = voc_push_node(v, &cap);
            voc_node *nd = &v->nodes[nid];
            const long pid = strtol(p, &e, 10);
            if (e == p || pid < 0 || pid >= nid) { ok = 0; break; }
            p = e;
            nd->parent = (int)pid;
            voc_node *par = &v->nodes[pid];
            if (par->nchild == par->cap) {
                par->cap = par->cap * 2 + 4;
                par->children = (int *)realloc(par->children, sizeof(int) * (size_t)par->cap);
            }
            par->children[par->nchild++] = nid;
            nd->is_leaf_flag = (int)strtol(p, &e, 10);
            if (e == p) { ok = 0; break; }
            p = e;
            for (int i = 0; i < 32; ++i) {
                const long b = strtol(p, &e, 10);
                if (e == p) { ok = 0; break; }
                nd->desc[i] = (uint8_t)b;
                p = e;
            }
            if (!ok) break;
            nd->weight = strtod(p, &e);
            if (e == p) { ok = 0; break; }
            if (nd->is_leaf_flag > 0) nd->word_id = (uint32_t)v->n_words++;
        }
    }
    free(line);
    fclose(f);
    if (!ok) { oracle_vocabulary_destroy(v); return NULL; }
    return v;
}

void oracle_vocabulary_info(const oracle_vocabulary *v, int *k, int *L, int *scoring, int *weighting, int *n_nodes,
                            int *n_words)
{
    *k = v->k; *L = v->L; *scoring = v->scoring; *weighting = v->weighting;
    *n_nodes = v->n_nodes; *n_words = v->n_words;
}

/* transform(feature, word_id, weight, nid, levelsup), TemplatedVocabulary.h:1218-1262 */
static void voc_transform_one(const oracle_vocabulary *v, const uint8_t *feature, int levelsup, uint32_t *word_id,
                              double *weight, uint32_t *nid)
{
    const int nid_level = v->L - levelsup;
    int nid_set = 0;
    if (nid_level <= 0) { *nid = 0; nid_set = 1; }
    int final_id = 0, current_level = 0;
    do {
        ++current_level;
        const voc_node *nd = &v->nodes[final_id];
        final_id = nd->children[0];
        double best_d = oracle_descriptor_distance(feature, v->nodes[final_id].desc);
        for (int c = 1; c < nd->nchild; ++c) {
            const int id = nd->children[c];
            const double d = oracle_descriptor_distance(feature, v->nodes[id].desc);
            if (d < best_d) { best_d = d; final_id = id; }
        }
        if (current_level == nid_level) { *nid = (uint32_t)final_id; nid_set = 1; }
    } while (v->nodes[final_id].nchild != 0);
    if (!nid_set) *nid = (uint32_t)final_id;
    *word_id = v->nodes[final_id].word_id;
    *weight = v->nodes[final_id].weight;
}

typedef struct { uint32_t id; int idx; } wordidx;
static int wordidx_cmp(const void *a, const void *b)
{
    const wordidx *A = (const wordidx *)a, *B = (const wordidx *)b;
    if (A->id != B->id) return A->id < B->id ? -1 : 1;
    return A->idx < B->idx ? -1 : (A->idx > B->idx);
}

/* transform(features, BowVector, FeatureVector, levelsup), :1127-1199.  word_id/weight/node_id: per feature
 * (node_id = ORACLE_NO_NODE for stopped words, w <= 0); bow_ids/bow_vals: the BowVector in map order. */
int oracle_vocabulary_transform(const oracle_vocabulary *v, const uint8_t *desc, int n, int levelsup,
                                uint32_t *word_id, double *weight, uint32_t *node_id, uint32_t *bow_ids,
                                double *bow_vals)
{
    for (int i = 0; i < n; ++i) { word_id[i] = 0; weight[i] = 0; node_id[i] = ORACLE_NO_NODE; }
    if (v->n_words == 0 || n == 0) return 0;   /* empty() */
    const int must = v->scoring != 5;           /* DotProductScoring: no normalisation */
    const int l2 = v->scoring == 1;
    wordidx *w = (wordidx *)malloc(sizeof(wordidx) * (size_t)(n + 1));
    int m = 0;
    for (int i = 0; i < n; ++i) {
        uint32_t nid;
        voc_transform_one(v, desc + (size_t)i * 32, levelsup, &word_id[i], &weight[i], &nid);
        if (weight[i] > 0) { node_id[i] = nid; w[m].id = word_id[i]; w[m].idx = i; ++m; }
    }
    /* std::map<WordId,WordValue> filled in feature order == stable grouping by word id */
    qsort(w, m, sizeof(wordidx), wordidx_cmp);
    int nb = 0;
    const int accumulate = v->weighting == 0 || v->weighting == 1;   /* TF_IDF, TF: addWeight; IDF, BINARY: addIfNotExist */
    for (int a = 0; a < m;) {
        int b = a;
        double val = weight[w[a].idx];
        for (b = a + 1; b < m && w[b].id == w[a].id; ++b)
            if (accumulate) val += weight[w[b].idx];
        bow_ids[nb] = w[a].id; bow_vals[nb] = val; ++nb;
        a = b;
    }
    free(w);
    if (accumulate && nb > 0 && !must) {
        const double nd = nb;
        for (int i = 0; i < nb; ++i) bow_vals[i] /= nd;
    }
    if (must) {   /* BowVector::normalize */
        double norm = 0.0;
        if (!l2) for (int i = 0; i < nb; ++i) norm += fabs(bow_vals[i]);
        else { for (int i = 0; i < nb; ++i) norm += bow_vals[i] * bow_vals[i]; norm = sqrt(norm); }
        if (norm > 0.0) for (int i = 0; i < nb; ++i) bow_vals[i] /= norm;
    }
    return nb;
}

/* ------------------------------------------------------------------------------------------------------------------
 * Projection prologues.  The reference does this arithmetic with cv::Mat expressions on CV_32F matrices; OpenCV is
 * not on disk, so the operation order is stated here (parity unpinned):
 *   - 3x3 * 3x1 + 3x1 (Rcw*x3Dw+tcw): per row ((r0*x + r1*y) + r2*z) + t in float, left to right, no contraction
 *     (cv::gemm's small-matrix path for CV_32F);
 *   - cv::norm (NORM_L2, CV_32F): squares accumulated in double in element order, sqrt in double, result to float;
 *   - Mat::dot (CV_32F): products accumulated in double in element order;
 *   - log(float) of MapPoint::PredictScale is logf; oracle_det_logf stands in for it (correctly rounded except on
 *     ~1e-8 of the inputs).
 * ------------------------------------------------------------------------------------------------------------------ */
float oracle_det_logf(float xf)
{
    /* fdlibm e_log.c, general path only */
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                 Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                 Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    if (!(xf > 0.0f)) return xf == 0.0f ? -INFINITY : NAN;
    if (isinf(xf)) return xf;
    double x = (double)xf;             /* every positive float is a normal double */
    uint64_t bits;
    memcpy(&bits, &x, 8);
    int hx = (int)(bits >> 32);
    int k = (hx >> 20) - 1023;
    hx &= 0x000fffff;
    int i = (hx + 0x95f64) & 0x100000;
    bits = ((uint64_t)(uint32_t)(hx | (i ^ 0x3ff00000)) << 32) | (bits & 0xffffffffu);   /* normalise x or x/2 */
    memcpy(&x, &bits, 8);
    k += i >> 20;
    const double f = x - 1.0;
    const double s = f / (2.0 + f);
    const double dk = (double)k;
    const double z = s * s;
    const double w = z * z;
    const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    const double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    const double R = t2 + t1;
    const double hfsq = (0.5 * f) * f;
    return (float)(dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f));
}

static float dot3_row(const float *r, float x, float y, float z, float t)
{
    return ((r[0] * x + r[1] * y) + r[2] * z) + t;
}

/* src/ORBmatcher.cc:1339-1390 */
void oracle_project_last_frame(const oracle_camera *cam, const float *Tcw, const float *Tlw, int n, const float *world,
                               const uint8_t *flags, const oracle_kp *last_keys, float th, int mono, oracle_query *q)
{
    /* twc = -Rcw.t()*tcw; tlc = Rlw*twc+tlw (:1342-1347): only tlc.z is used */
    float twc[3];
    for (int i = 0; i < 3; ++i) twc[i] = -((Tcw[0 + i] * Tcw[3] + Tcw[4 + i] * Tcw[7]) + Tcw[8 + i] * Tcw[11]);
    const float tlc_z = dot3_row(Tlw + 8, twc[0], twc[1], twc[2], Tlw[11]);
    const int forward = tlc_z > cam->mb && !mono, backward = -tlc_z > cam->mb && !mono;   /* :1349-1350 */
    for (int i = 0; i < n; ++i) {
        memset(&q[i], 0, sizeof(q[i]));
        if (!(flags[i] & ORACLE_POINT_PRESENT)) continue;          /* pMP && !mvbOutlier[i] */
        const float *X = world + 3 * i;
        const float xc = dot3_row(Tcw, X[0], X[1], X[2], Tcw[3]);
        const float yc = dot3_row(Tcw + 4, X[0], X[1], X[2], Tcw[7]);
        const float zc = dot3_row(Tcw + 8, X[0], X[1], X[2], Tcw[11]);
        const float invzc = (float)(1.0 / (double)zc);                /* const float invzc = 1.0/x3Dc.at<float>(2) */
        if (invzc < 0) continue;
        const float u = cam->fx * xc * invzc + cam->cx;
        const float v = cam->fy * yc * invzc + cam->cy;
        if (u < cam->min_x || u > cam->max_x) continue;
        if (v < cam->min_y || v > cam->max_y) continue;
        const int o = last_keys[i].octave;
        q[i].valid = 1;
        q[i].u = u; q[i].v = v;
        q[i].radius = th * cam->scale_factors[o];
        if (forward) { q[i].min_level = o; q[i].max_level = -1; }
        else if (backward) { q[i].min_level = 0; q[i].max_level = o; }
        else { q[i].min_level = o - 1; q[i].max_level = o + 1; }
        q[i].ur = u - cam->mbf * invzc;
        q[i].level_aux = o;
        q[i].angle = last_keys[i].angle;
        q[i].observed = (flags[i] & ORACLE_POINT_OBSERVED) ? 1 : 0;
    }
}

/* src/Frame.cc:269-325, src/MapPoint.cc:400-418, src/ORBmatcher.cc:52-69 + :131-137 */
void oracle_frustum_queries(const oracle_camera *cam, const float *Tcw, int n, const float *world, const float *normal,
                            const float *max_dist, const float *min_dist, const uint8_t *flags, float viewing_cos_limit,
                            float th, oracle_query *q, float *view_cos)
{
    float Ow[3];   /* mOw = -mRcw.t()*mtcw, Frame.cc:266 */
    for (int i = 0; i < 3; ++i) Ow[i] = -((Tcw[0 + i] * Tcw[3] + Tcw[4 + i] * Tcw[7]) + Tcw[8 + i] * Tcw[11]);
    const int bFactor = th != 1.0;
    for (int i = 0; i < n; ++i) {
        memset(&q[i], 0, sizeof(q[i]));
        if (view_cos) view_cos[i] = 0.0f;
        if (!(flags[i] & ORACLE_POINT_PRESENT)) continue;
        const float *P = world + 3 * i;
        const float PcX = dot3_row(Tcw, P[0], P[1], P[2], Tcw[3]);
        const float PcY = dot3_row(Tcw + 4, P[0], P[1], P[2], Tcw[7]);
        const float PcZ = dot3_row(Tcw + 8, P[0], P[1], P[2], Tcw[11]);
        if (PcZ < 0.0f) continue;
        const float invz = 1.0f / PcZ;
        const float u = cam->fx * PcX * invz + cam->cx;
        const float v = cam->fy * PcY * invz + cam->cy;
        if (u < cam->min_x || u > cam->max_x) continue;
        if (v < cam->min_y || v > cam->max_y) continue;
        const float maxDistance = 1.2f * max_dist[i], minDistance = 0.8f * min_dist[i];   /* MapPoint.cc:370-383 */
        const float PO[3] = {P[0] - Ow[0], P[1] - Ow[1], P[2] - Ow[2]};
        const float dist = (float)sqrt(((double)PO[0] * PO[0] + (double)PO[1] * PO[1]) + (double)PO[2] * PO[2]);
        if (dist < minDistance || dist > maxDistance) continue;
        const float *Pn = normal + 3 * i;
        const double dot = ((double)PO[0] * Pn[0] + (double)PO[1] * Pn[1]) + (double)PO[2] * Pn[2];
        const float viewCos = (float)(dot / (double)dist);
        if (viewCos < viewing_cos_limit) continue;
        /* PredictScale */
        const float ratio = max_dist[i] / dist;
        const float fl = ceilf(oracle_det_logf(ratio) / cam->log_scale_factor);
        int nScale = fl >= (float)cam->n_levels ? cam->n_levels - 1 : (fl < 0 ? 0 : (int)fl);   /* also tames inf / nan */
        if (!(fl == fl)) nScale = 0;
        float r = (double)viewCos > 0.998 ? 2.5f : 4.0f;   /* RadiusByViewingCos */
        if (bFactor) r *= th;
        q[i].valid = 1;
        q[i].u = u; q[i].v = v;
        q[i].radius = r * cam->scale_factors[nScale];
        q[i].min_level = nScale - 1; q[i].max_level = nScale;
        q[i].ur = u - cam->mbf * invz;
        q[i].level_aux = nScale;
        q[i].angle = 0.0f;
        q[i].observed = (flags[i] & ORACLE_POINT_OBSERVED) ? 1 : 0;
        if (view_cos) view_cos[i] = viewCos;
    }
}

static int predict_scale(float max_dist, float dist, const oracle_camera *cam)   /* MapPoint::PredictScale, MapPoint.cc:385-418 */
{
    const float ratio = max_dist / dist;
    const float fl = ceilf(oracle_det_logf(ratio) / cam->log_scale_factor);
    int nScale = fl >= (float)cam->n_levels ? cam->n_levels - 1 : (fl < 0 ? 0 : (int)fl);
    if (!(fl == fl)) nScale = 0;
    return nScale;
}

void oracle_keyframe_queries(const oracle_camera *cam, int mode, int double_invz, const float *T1, const float *T2, int n,
                             const float *world, const float *normal, const float *max_dist, const float *min_dist,
                             const uint8_t *flags, float th, oracle_query *q)
{
    float Ow[3] = {0, 0, 0};
    if (mode == 0) for (int i = 0; i < 3; ++i) Ow[i] = -((T1[0 + i] * T1[3] + T1[4 + i] * T1[7]) + T1[8 + i] * T1[11]);
    for (int i = 0; i < n; ++i) {
        memset(&q[i], 0, sizeof(q[i]));
        if (!(flags[i] & ORACLE_POINT_PRESENT)) continue;
        const float *P = world + 3 * i;
        float X = dot3_row(T1, P[0], P[1], P[2], T1[3]);
        float Y = dot3_row(T1 + 4, P[0], P[1], P[2], T1[7]);
        float Z = dot3_row(T1 + 8, P[0], P[1], P[2], T1[11]);
        if (mode == 1) {
            const float x1 = X, y1 = Y, z1 = Z;
            X = dot3_row(T2, x1, y1, z1, T2[3]);
            Y = dot3_row(T2 + 4, x1, y1, z1, T2[7]);
            Z = dot3_row(T2 + 8, x1, y1, z1, T2[11]);
        }
        if (Z < 0.0f) continue;
        const float invz = double_invz ? (float)(1.0 / (double)Z) : 1.0f / Z;
        const float x = X * invz, y = Y * invz;
        const float u = cam->fx * x + cam->cx, v = cam->fy * y + cam->cy;
        if (!(u >= cam->min_x && u < cam->max_x && v >= cam->min_y && v < cam->max_y)) continue;   /* KeyFrame::IsInImage */
        const float maxDistance = 1.2f * max_dist[i], minDistance = 0.8f * min_dist[i];
        float dist3D;
        if (mode == 0) {
            const float PO[3] = {P[0] - Ow[0], P[1] - Ow[1], P[2] - Ow[2]};
            dist3D = (float)sqrt(((double)PO[0] * PO[0] + (double)PO[1] * PO[1]) + (double)PO[2] * PO[2]);
            if (dist3D < minDistance || dist3D > maxDistance) continue;
            const float *Pn = normal + 3 * i;
            const double dot = ((double)PO[0] * Pn[0] + (double)PO[1] * Pn[1]) + (double)PO[2] * Pn[2];
            if (dot < 0.5 * (double)dist3D) continue;
        } else {
            dist3D = (float)sqrt(((double)X * X + (double)Y * Y) + (double)Z * Z);
            if (dist3D < minDistance || dist3D > maxDistance) continue;
        }
        const int lvl = predict_scale(max_dist[i], dist3D, cam);
        q[i].valid = 1;
        q[i].u = u; q[i].v = v;
        q[i].radius = th * cam->scale_factors[lvl];
        q[i].min_level = lvl - 1; q[i].max_level = lvl;
        q[i].ur = mode == 0 ? u - cam->mbf * invz : 0.0f;
        q[i].level_aux = lvl;
    }
}

int oracle_search_by_sim3(const oracle_frame *kf1, const oracle_frame *kf2, const oracle_camera *cam, const float *T1w,
                          const float *T2w, const float *S21, const float *S12, const float *world1, const float *max1,
                          const float *min1, const uint8_t *flags1, const uint8_t *desc1, const float *world2,
                          const float *max2, const float *min2, const uint8_t *flags2, const uint8_t *desc2, float th,
                          int32_t *matches12)
{
    const int n1 = kf1->n, n2 = kf2->n;
    oracle_query *q1 = (oracle_query *)malloc(sizeof(oracle_query) * (size_t)(n1 + 1));
    oracle_query *q2 = (oracle_query *)malloc(sizeof(oracle_query) * (size_t)(n2 + 1));
    int32_t *m1 = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n1 + 1)), *d1 = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n1 + 1));
    int32_t *m2 = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n2 + 1)), *d2 = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n2 + 1));
    oracle_keyframe_queries(cam, 1, 1, T1w, S21, n1, world1, NULL, max1, min1, flags1, th, q1);   /* KF1 points into KF2 */
    oracle_keyframe_queries(cam, 1, 1, T2w, S12, n2, world2, NULL, max2, min2, flags2, th, q2);   /* KF2 points into KF1 */
    oracle_search_best_in_window(kf2, q1, desc1, n1, NULL, m1, d1);
    oracle_search_best_in_window(kf1, q2, desc2, n2, NULL, m2, d2);
    int nFound = 0;
    for (int i1 = 0; i1 < n1; ++i1) {
        matches12[i1] = -1;
        const int idx2 = (m1[i1] >= 0 && d1[i1] <= TH_HIGH) ? m1[i1] : -1;      /* vnMatch1, :1222-1225 */
        if (idx2 >= 0) {
            const int idx1 = (m2[idx2] >= 0 && d2[idx2] <= TH_HIGH) ? m2[idx2] : -1;
            if (idx1 == i1) { matches12[i1] = idx2; nFound++; }
        }
    }
    free(q1); free(q2); free(m1); free(d1); free(m2); free(d2);
    return nFound;
}
