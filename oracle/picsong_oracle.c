/*
 * picsong_oracle.c -- CPU restatement (plain C99) of the PICSONG hot path:
 *   pad -> level shift -> DWT 5/3 | 9/7(+quant) -> BPC-PaCo (2 passes, k=0) -> BitStreamBuilder
 * and the exact inverse.
 *
 * TEST INFRASTRUCTURE ONLY (see picsong_oracle.h).  PARITY UNPINNED against the CUDA binary.
 *
 * All file:line citations are relative to /root/reference/CUDA_ImCod/.
 *
 * Floating point (9/7): the reference is built with nvcc -use_fast_math (CMakeLists.txt:9), so
 * its results are not bit-reproducible.  This restatement fixes IEEE fp32 with the contraction
 * nvcc applies by default to `b += (a+c)*w` (one fused multiply-add), true divisions, and no
 * re-association.  Build with -ffp-contract=off so the compiler adds no contraction of its own.
 */
#include "picsong_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* Threads used by the parallel loops (codeblocks, DWT rows / columns); 1 = scalar port.
 * Only the CPU-baseline leg of bench.py raises it; results do not depend on it. */
static int po_threads = 1;
void po_set_threads(int n) { po_threads = n < 1 ? 1 : n; }
int po_get_threads(void) { return po_threads; }
int po_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_num_procs();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------------------------ */
/* geometry / ingest                                                                            */
/* ------------------------------------------------------------------------------------------ */

/* SupportFunctions/AuxiliarFunctions.cpp:22-26 -- ceil(v/64)*64 */
int po_pad_dim(int v) { return ((v + PO_CB - 1) / PO_CB) * PO_CB; }

/* IO/IOManager.ipp:72-112 -- pad right then bottom by edge-inclusive mirror:
 * col W+j = col W-1-j (per row, :99-106); row H+r = row H-1-r of the already widened image
 * (:107-110). */
/* Mirror padding of IOManager::loadFrameCAdaptedSizes (IO/IOManager.ipp:72-112).  The reference's
 * insert loop reads memblock[W - 1 - j] of the row for added column j and row H - 1 - r for added row
 * r: with more added columns than columns (2W < AW) or more added rows than rows (2H < AH) it indexes
 * before the vector's begin (undefined behaviour on row 0) -- such frames are refused, -1. */
int po_pad_frame(const uint8_t *in, int W, int H, uint8_t *out, int AW, int AH)
{
    if (W <= 0 || H <= 0 || AW < W || AH < H || AW - W > W || AH - H > H) return -1;
    for (int y = 0; y < H; y++) {
        memcpy(out + (size_t)y * AW, in + (size_t)y * W, (size_t)W);
        for (int j = 0; j < AW - W; j++)
            out[(size_t)y * AW + W + j] = in[(size_t)y * W + (W - 1 - j)];
    }
    for (int r = 0; r < AH - H; r++)
        memcpy(out + (size_t)(H + r) * AW, out + (size_t)(H - 1 - r) * AW, (size_t)AW);
    return 0;
}

void po_crop_frame_u8(const uint8_t *in, int AW, int AH, uint8_t *out, int W, int H)
{
    (void)AH;
    for (int y = 0; y < H; y++)
        memcpy(out + (size_t)y * W, in + (size_t)y * AW, (size_t)W);
}

/* integer-only synthetic frame. */
static int po_tri(int v, int p) { int m = v % (2 * p); int d = m - p; return d < 0 ? -d : d; }

void po_gen_frame(uint8_t *out, int W, int H, uint32_t frame, uint32_t seed)
{
    uint32_t z = seed ^ (frame * 0x9E3779B9u);
    for (int y = 0; y < H; y++) {
        int by = po_tri(y, 384) * 255 / 384;
        for (int x = 0; x < W; x++) {
            z = 1664525u * z + 1013904223u;
            int base = (po_tri(x, 512) * 255 / 512 + by) / 2;
            int noise = (int)((z >> 24) & 15u) - 8;
            int checker = 16 * (((x >> 5) ^ (y >> 5)) & 1);
            int p = base + noise + checker;
            out[(size_t)y * W + x] = (uint8_t)(p < 0 ? 0 : (p > 255 ? 255 : p));
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* level shift                                                                                  */
/* ------------------------------------------------------------------------------------------ */

/* Engines/CodingEngine.cu:581-588 -- out = (T)in - (1 << (bitDepth-1)) */
void po_level_shift_fwd_i32(const uint8_t *in, int32_t *out, size_t n, int bit_depth)
{
    int off = 1 << (bit_depth - 1);
    for (size_t i = 0; i < n; i++) out[i] = (int32_t)in[i] - off;
}

void po_level_shift_fwd_f32(const uint8_t *in, float *out, size_t n, int bit_depth)
{
    float off = (float)(1 << (bit_depth - 1));
    for (size_t i = 0; i < n; i++) out[i] = (float)in[i] - off;
}

/* Engines/DecodingEngine.cu:720-729 -- max(min(x + offset, 255), 0) */
void po_level_shift_inv_i32(int32_t *data, size_t n, int bit_depth)
{
    int off = 1 << (bit_depth - 1);
    for (size_t i = 0; i < n; i++) {
        int v = data[i] + off;
        data[i] = v > 255 ? 255 : (v < 0 ? 0 : v);
    }
}

/* Engines/DecodingEngine.cu:706-715 -- fmaxf(fminf(__float2int_rn(x + offset + 0.01f), 255), 0);
 * __float2int_rn = round to nearest even -> rintf under the default rounding mode. */
void po_level_shift_inv_f32(float *data, size_t n, int bit_depth)
{
    float off = (float)(1 << (bit_depth - 1));
    for (size_t i = 0; i < n; i++) {
        float t = data[i] + off;
        t = t + 0.01f;
        float r = rintf(t);
        r = r > 255.0f ? 255.0f : r;
        r = r < 0.0f ? 0.0f : r;
        data[i] = r;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* colour transforms (RGB path)                                                                 */
/* ------------------------------------------------------------------------------------------ */

/* Engines/CodingEngine.cuh:25, Engines/DecodingEngine.cuh:41 */
static const float PO_ICT_F[3][3] = { { 0.299f, 0.587f, 0.114f }, { -0.168736f, -0.331264f, 0.5f },
                                      { 0.5f, -0.418688f, -0.081312f } };
static const float PO_ICT_B[3][3] = { { 1.0f, 0.0f, 1.402f }, { 1.0f, -0.344136f, -0.714136f },
                                      { 1.0f, 1.772f, 0.0f } };

/* RGBTransformLossless Engines/CodingEngine.cu:357-379: level shift, then
 * c0 = floor((R + 2G + B) / 4), c1 = B - G, c2 = R - G */
void po_rct_forward(const uint8_t *r, const uint8_t *g, const uint8_t *b, int32_t *c0, int32_t *c1,
                    int32_t *c2, size_t n, int bit_depth)
{
    int off = 1 << (bit_depth - 1);
    for (size_t i = 0; i < n; i++) {
        int R = (int)r[i] - off, G = (int)g[i] - off, B = (int)b[i] - off;
        c0[i] = (R + 2 * G + B) >> 2;          /* floor: arithmetic shift */
        c1[i] = B - G;
        c2[i] = R - G;
    }
}

/* RGBTransformLossless Engines/DecodingEngine.cu:599-623: G = c0 - floor((c1 + c2) / 4),
 * R = c2 + G, B = c1 + G, + offset, clamp to 0..255 */
void po_rct_inverse(const int32_t *c0, const int32_t *c1, const int32_t *c2, uint8_t *r, uint8_t *g,
                    uint8_t *b, size_t n, int bit_depth)
{
    int off = 1 << (bit_depth - 1);
    for (size_t i = 0; i < n; i++) {
        int G = c0[i] - ((c1[i] + c2[i]) >> 2);
        int R = c2[i] + G, B = c1[i] + G;
        R += off; G += off; B += off;
        r[i] = (uint8_t)(R > 255 ? 255 : (R < 0 ? 0 : R));
        g[i] = (uint8_t)(G > 255 ? 255 : (G < 0 ? 0 : G));
        b[i] = (uint8_t)(B > 255 ? 255 : (B < 0 ? 0 : B));
    }
}

/* RGBTransformLossy Engines/CodingEngine.cu:384-403: m0*R + m1*G + m2*B left to right, with the
 * contraction nvcc applies by default: fmaf(m2, B, fmaf(m1, G, m0*R)) */
void po_ict_forward(const uint8_t *r, const uint8_t *g, const uint8_t *b, float *c0, float *c1, float *c2,
                    size_t n, int bit_depth)
{
    float off = (float)(1 << (bit_depth - 1));
    float *out[3] = { c0, c1, c2 };
    for (size_t i = 0; i < n; i++) {
        float R = (float)r[i] - off, G = (float)g[i] - off, B = (float)b[i] - off;
        for (int k = 0; k < 3; k++)
            out[k][i] = fmaf(PO_ICT_F[k][2], B, fmaf(PO_ICT_F[k][1], G, PO_ICT_F[k][0] * R));
    }
}

/* RGBTransformLossy Engines/DecodingEngine.cu:628-650: __float2int_rn(m . c + 0.01f) + offset, clamp */
void po_ict_inverse(const float *c0, const float *c1, const float *c2, uint8_t *r, uint8_t *g, uint8_t *b,
                    size_t n, int bit_depth)
{
    int off = 1 << (bit_depth - 1);
    uint8_t *out[3] = { r, g, b };
    for (size_t i = 0; i < n; i++)
        for (int k = 0; k < 3; k++) {
            float t = fmaf(PO_ICT_B[k][2], c2[i], fmaf(PO_ICT_B[k][1], c1[i], PO_ICT_B[k][0] * c0[i]));
            t = t + 0.01f;
            int v = (int)rintf(t) + off;
            out[k][i] = (uint8_t)(v > 255 ? 255 : (v < 0 ? 0 : v));
        }
}

/* ------------------------------------------------------------------------------------------ */
/* DWT                                                                                          */
/* ------------------------------------------------------------------------------------------ */

/* Engines/CodingEngine.cu:170-177 -- extra = sum_{l=1}^{wl-1} (AW>>l)(AH>>l) */
size_t po_dwt_extra(int AW, int AH, int wl)
{
    size_t e = 0;
    for (int l = 1; l < wl; l++) e += (size_t)(AW >> l) * (size_t)(AH >> l);
    return e;
}

/* DWT/DWTGenerator.cuh:168-179 -- quantisation steps, columns LL,HL,LH,HH, row = level */
static const float PO_QSTEPS[10][4] = {
    { 1.965908f, 1.0112865f, 1.0112865f, 0.52021784f },
    { 4.1224113f, 1.9968134f, 1.9968134f, 0.96721643f },
    { 8.416739f, 4.1833673f, 4.1833673f, 2.0792568f },
    { 16.935543f, 8.534108f, 8.534108f, 4.3004827f },
    { 33.924816f, 17.166693f, 17.166693f, 8.686718f },
    { 67.87687f, 34.385098f, 34.385098f, 17.41882f },
    { 135.76744f, 68.7964f, 68.7964f, 34.860676f },
    { 271.5416f, 137.60588f, 137.60588f, 69.73287f },
    { 543.0866f, 275.21814f, 275.21814f, 139.47136f },
    { 1086.1624f, 550.43286f, 550.43286f, 278.94202f }
};

/* DWT/DWTGenerator.cuh:16-22 */
#define PO_A1 (-1.586134342059924f)
#define PO_A2 (-0.052980118572961f)
#define PO_A3 (0.882911075530934f)
#define PO_A4 (0.443506852043971f)
#define PO_N1 (1.230174104914001f)
#define PO_N2 (0.812893066f)

/* One reversible 5/3 analysis of x[0], x[st], ..., x[(n-1)st] in place (interleaved s,d).
 * Lifting: DWTGenerator.cu:72-76; order and boundary rule: vertical :137-157,
 * horizontal :279-292 (lane 31 / last row re-use their own even sample = whole-sample
 * symmetric extension; lane 0 / row 0 re-use d[0]). */
static void po_53_fwd_1d(int32_t *x, size_t st, int n)
{
    for (int i = 1; i < n - 1; i += 2)
        x[i * st] -= (x[(i - 1) * st] + x[(i + 1) * st]) >> 1;
    x[(n - 1) * st] -= (x[(n - 2) * st] + x[(n - 2) * st]) >> 1;
    x[0] += (x[st] + x[st] + 2) >> 2;
    for (int i = 2; i < n; i += 2)
        x[i * st] += (x[(i - 1) * st] + x[(i + 1) * st] + 2) >> 2;
}

/* DWTGenerator.cu:81-85 lifting; vertical :160-181, horizontal :295-308 */
static void po_53_inv_1d(int32_t *x, size_t st, int n)
{
    x[0] -= (x[st] + x[st] + 2) >> 2;
    for (int i = 2; i < n; i += 2)
        x[i * st] -= (x[(i - 1) * st] + x[(i + 1) * st] + 2) >> 2;
    for (int i = 1; i < n - 1; i += 2)
        x[i * st] += (x[(i - 1) * st] + x[(i + 1) * st]) >> 1;
    x[(n - 1) * st] += (x[(n - 2) * st] + x[(n - 2) * st]) >> 1;
}

/* 9/7 analysis, DWTGenerator.cu:91-104 lifting, vertical :184-227, horizontal :311-323.
 * `*b += (a+c)*w` -> fmaf(a+c, w, b) (nvcc default -fmad=true); step four:
 * b = (b + (a+c)*d) * N2 -> fmaf(a+c, d, b) * N2; finally odd *= N1. */
static void po_97_fwd_1d(float *x, size_t st, int n)
{
    for (int i = 1; i < n - 1; i += 2)
        x[i * st] = fmaf(x[(i - 1) * st] + x[(i + 1) * st], PO_A1, x[i * st]);
    x[(n - 1) * st] = fmaf(x[(n - 2) * st] + x[(n - 2) * st], PO_A1, x[(n - 1) * st]);
    x[0] = fmaf(x[st] + x[st], PO_A2, x[0]);
    for (int i = 2; i < n; i += 2)
        x[i * st] = fmaf(x[(i - 1) * st] + x[(i + 1) * st], PO_A2, x[i * st]);
    for (int i = 1; i < n - 1; i += 2)
        x[i * st] = fmaf(x[(i - 1) * st] + x[(i + 1) * st], PO_A3, x[i * st]);
    x[(n - 1) * st] = fmaf(x[(n - 2) * st] + x[(n - 2) * st], PO_A3, x[(n - 1) * st]);
    x[0] = fmaf(x[st] + x[st], PO_A4, x[0]) * PO_N2;
    for (int i = 2; i < n; i += 2)
        x[i * st] = fmaf(x[(i - 1) * st] + x[(i + 1) * st], PO_A4, x[i * st]) * PO_N2;
    for (int i = 1; i < n; i += 2)
        x[i * st] = x[i * st] * PO_N1;
}

/* 9/7 synthesis, DWTGenerator.cu:110-122 lifting, vertical :230-272, horizontal :326-339.
 * odd /= N1; even = even/N2 - (a+c)*A4 ; odd -= (a+c)*A3 ; even -= (a+c)*A2 ; odd -= (a+c)*A1 */
static void po_97_inv_1d(float *x, size_t st, int n)
{
    for (int i = 1; i < n; i += 2)
        x[i * st] = x[i * st] / PO_N1;
    x[0] = fmaf(-(x[st] + x[st]), PO_A4, x[0] / PO_N2);
    for (int i = 2; i < n; i += 2)
        x[i * st] = fmaf(-(x[(i - 1) * st] + x[(i + 1) * st]), PO_A4, x[i * st] / PO_N2);
    for (int i = 1; i < n - 1; i += 2)
        x[i * st] = fmaf(-(x[(i - 1) * st] + x[(i + 1) * st]), PO_A3, x[i * st]);
    x[(n - 1) * st] = fmaf(-(x[(n - 2) * st] + x[(n - 2) * st]), PO_A3, x[(n - 1) * st]);
    x[0] = fmaf(-(x[st] + x[st]), PO_A2, x[0]);
    for (int i = 2; i < n; i += 2)
        x[i * st] = fmaf(-(x[(i - 1) * st] + x[(i + 1) * st]), PO_A2, x[i * st]);
    for (int i = 1; i < n - 1; i += 2)
        x[i * st] = fmaf(-(x[(i - 1) * st] + x[(i + 1) * st]), PO_A1, x[i * st]);
    x[(n - 1) * st] = fmaf(-(x[(n - 2) * st] + x[(n - 2) * st]), PO_A1, x[(n - 1) * st]);
}

/* Forward driver, DWTGenerator.cu:1268-1342 + placement :699-723 / :403-433.
 * Level l reads `in` (l = 0, stride AW) or out + sum_{j<l} Wj*Hj (packed); vertical then
 * horizontal lifting; LL -> out + sum_{j<=l} Wj*Hj packed (stride Wl/2), or out[0..] stride AW on
 * the last level; HL/LH/HH -> Mallat positions in out, stride AW. */
void po_dwt53_forward(const int32_t *in, int32_t *out, int AW, int AH, int wl)
{
    int W = AW, H = AH;
    size_t off = 0;
    int32_t *tmp = (int32_t *)malloc((size_t)AW * AH * sizeof(int32_t));
    const int32_t *src = in;
    for (int l = 0; l < wl; l++) {
        int last = (l == wl - 1);
        memcpy(tmp, src, (size_t)W * H * sizeof(int32_t));
        _Pragma("omp parallel for num_threads(po_threads) schedule(static)")
        for (int x = 0; x < W; x++) po_53_fwd_1d(tmp + x, (size_t)W, H);
        _Pragma("omp parallel for num_threads(po_threads) schedule(static)")
        for (int y = 0; y < H; y++) po_53_fwd_1d(tmp + (size_t)y * W, 1, W);
        off += (size_t)W * H;
        int32_t *ll = last ? out : out + off;
        size_t llst = last ? (size_t)AW : (size_t)(W >> 1);
        _Pragma("omp parallel for num_threads(po_threads) schedule(static)")
        for (int y = 0; y < H; y += 2)
            for (int x = 0; x < W; x += 2) {
                size_t r = (size_t)(y >> 1), c = (size_t)(x >> 1);
                ll[r * llst + c] = tmp[(size_t)y * W + x];
                out[r * AW + c + (W >> 1)] = tmp[(size_t)y * W + x + 1];
                out[(r + (H >> 1)) * AW + c] = tmp[(size_t)(y + 1) * W + x];
                out[(r + (H >> 1)) * AW + c + (W >> 1)] = tmp[(size_t)(y + 1) * W + x + 1];
            }
        src = out + off;
        W >>= 1; H >>= 1;
    }
    free(tmp);
}

/* Reverse driver, DWTGenerator.cu:1349-1424, reads :477-509, writes :556-691.
 * Coarsest level first; level wl-1 takes LL from the Mallat array, finer levels take LL from the
 * previous packed output; horizontal inverse then vertical inverse (:1116-1117); each level's
 * output is packed at out + writeOffset; the full image ends at out + po_dwt_extra(). */
void po_dwt53_inverse(const int32_t *in, int32_t *out, int AW, int AH, int wl)
{
    int W = AW >> (wl - 1), H = AH >> (wl - 1);
    size_t read_off = 0, write_off = 0;
    int32_t *tmp = (int32_t *)malloc((size_t)AW * AH * sizeof(int32_t));
    for (int l = wl - 1; l >= 0; l--) {
        int first = (l == wl - 1);
        const int32_t *ll = first ? in : out + read_off;
        size_t llst = first ? (size_t)AW : (size_t)(W >> 1);
        _Pragma("omp parallel for num_threads(po_threads) schedule(static)")
        for (int y = 0; y < H; y += 2)
            for (int x = 0; x < W; x += 2) {
                size_t r = (size_t)(y >> 1), c = (size_t)(x >> 1);
                tmp[(size_t)y * W + x] = ll[r * llst + c];
                tmp[(size_t)y * W + x + 1] = in[r * AW + c + (W >> 1)];
                tmp[(size_t)(y + 1) * W + x] = in[(r + (H >> 1)) * AW + c];
                tmp[(size_t)(y + 1) * W + x + 1] = in[(r + (H >> 1)) * AW + c + (W >> 1)];
            }
        _Pragma("omp parallel for num_threads(po_threads) schedule(static)")
        for (int y = 0; y < H; y++) po_53_inv_1d(tmp + (size_t)y * W, 1, W);
        _Pragma("omp parallel for num_threads(po_threads) schedule(static)")
        for (int x = 0; x < W; x++) po_53_inv_1d(tmp + x, (size_t)W, H);
        memcpy(out + write_off, tmp, (size_t)W * H * sizeof(int32_t));
        read_off = write_off;
        write_off += (size_t)W * H;
        W <<= 1; H <<= 1;
    }
    free(tmp);
}

/* DWTGenerator.cu:405-419 -- quantisation on write: v * Q[l][sb] * qs (left to right);
 * LL only on the last level with Q[wl-1][0]. */
void po_dwt97_forward(const float *in, float *out, int AW, int AH, int wl, float qs)
{
    int W = AW, H = AH;
    size_t off = 0;
    float *tmp = (float *)malloc((size_t)AW * AH * sizeof(float));
    const float *src = in;
    for (int l = 0; l < wl; l++) {
        int last = (l == wl - 1);
        memcpy(tmp, src, (size_t)W * H * sizeof(float));
        _Pragma("omp parallel for num_threads(po_threads) schedule(static)")
        for (int x = 0; x < W; x++) po_97_fwd_1d(tmp + x, (size_t)W, H);
        _Pragma("omp parallel for num_threads(po_threads) schedule(static)")
        for (int y = 0; y < H; y++) po_97_fwd_1d(tmp + (size_t)y * W, 1, W);
        off += (size_t)W * H;
        float *ll = last ? out : out + off;
        size_t llst = last ? (size_t)AW : (size_t)(W >> 1);
        const float *q = PO_QSTEPS[l];
        _Pragma("omp parallel for num_threads(po_threads) schedule(static)")
        for (int y = 0; y < H; y += 2)
            for (int x = 0; x < W; x += 2) {
                size_t r = (size_t)(y >> 1), c = (size_t)(x >> 1);
                float vll = tmp[(size_t)y * W + x];
                ll[r * llst + c] = last ? (vll * q[0]) * qs : vll;
                out[r * AW + c + (W >> 1)] = (tmp[(size_t)y * W + x + 1] * q[1]) * qs;
                out[(r + (H >> 1)) * AW + c] = (tmp[(size_t)(y + 1) * W + x] * q[2]) * qs;
                out[(r + (H >> 1)) * AW + c + (W >> 1)] =
                    (tmp[(size_t)(y + 1) * W + x + 1] * q[3]) * qs;
            }
        src = out + off;
        W >>= 1; H >>= 1;
    }
    free(tmp);
}

/* DWTGenerator.cu:513-553 -- de-quantisation on read:
 * v == 0 -> 0 else ((|v| + 0.5) * sgn(v)) / Q[l][sb] / qs ; LL de-quantised only on the
 * coarsest level (:531-542), finer levels read the float LL of the previous output (:546-553). */
static float po_dequant(int32_t v, float q, float qs)
{
    if (v == 0) return 0.0f;
    float m = fabsf((float)v) + 0.5f;
    float s = (v < 0) ? -1.0f : 1.0f;
    return ((m * s) / q) / qs;
}

void po_dwt97_inverse(const int32_t *in, float *out, int AW, int AH, int wl, float qs)
{
    int W = AW >> (wl - 1), H = AH >> (wl - 1);
    size_t read_off = 0, write_off = 0;
    float *tmp = (float *)malloc((size_t)AW * AH * sizeof(float));
    for (int l = wl - 1; l >= 0; l--) {
        int first = (l == wl - 1);
        const float *q = PO_QSTEPS[l];
        _Pragma("omp parallel for num_threads(po_threads) schedule(static)")
        for (int y = 0; y < H; y += 2)
            for (int x = 0; x < W; x += 2) {
                size_t r = (size_t)(y >> 1), c = (size_t)(x >> 1);
                tmp[(size_t)y * W + x] = first ? po_dequant(in[r * AW + c], q[0], qs)
                                               : out[read_off + r * (size_t)(W >> 1) + c];
                tmp[(size_t)y * W + x + 1] = po_dequant(in[r * AW + c + (W >> 1)], q[1], qs);
                tmp[(size_t)(y + 1) * W + x] = po_dequant(in[(r + (H >> 1)) * AW + c], q[2], qs);
                tmp[(size_t)(y + 1) * W + x + 1] =
                    po_dequant(in[(r + (H >> 1)) * AW + c + (W >> 1)], q[3], qs);
            }
        _Pragma("omp parallel for num_threads(po_threads) schedule(static)")
        for (int y = 0; y < H; y++) po_97_inv_1d(tmp + (size_t)y * W, 1, W);
        _Pragma("omp parallel for num_threads(po_threads) schedule(static)")
        for (int x = 0; x < W; x++) po_97_inv_1d(tmp + x, (size_t)W, H);
        memcpy(out + write_off, tmp, (size_t)W * H * sizeof(float));
        read_off = write_off;
        write_off += (size_t)W * H;
        W <<= 1; H <<= 1;
    }
    free(tmp);
}

/* ------------------------------------------------------------------------------------------ */
/* LUT loader                                                                                   */
/* ------------------------------------------------------------------------------------------ */

/* IO/IOManager.ipp:363-386 + Engines/Engine.cu:190-210: 8 lines "KEY;int". */
static int po_lut_header(const char *folder, po_lut *l)
{
    char path[1024];
    snprintf(path, sizeof path, "%sheader.txt", folder);
    FILE *f = fopen(path, "rb");
    if (!f) {
        snprintf(path, sizeof path, "%s/header.txt", folder);
        f = fopen(path, "rb");
    }
    if (!f) return -1;
    int v[8] = { 0 };
    char line[256];
    int n = 0;
    while (n < 8 && fgets(line, sizeof line, f)) {
        char *semi = strchr(line, ';');
        if (!semi) continue;
        v[n++] = atoi(semi + 1);
    }
    fclose(f);
    if (n < 8) return -2;
    l->n_bitplanes = v[0]; l->n_subbands = v[1]; l->ctx_ref = v[2]; l->ctx_sign = v[3];
    l->ctx_sig = v[4]; l->precision = v[5]; l->n_files = v[6];
    l->n_bp_files = v[7] > 32 ? 32 : v[7];
    return 0;
}

/* One section of IO/IOManager.ipp:404-612 (ref :455-476, sig :478-510, sign :512-546).
 * Lines "lvl sb bp : v0 .. v(C-1)".  On a group change (bp <= previous bp) the previous group's
 * remaining planes are filled with 64 and the write index advances by nBp*C; the break test
 * (lvl+1 > wl && sb > 0) comes after that.  Entries never written keep `fill`. */
static int po_lut_section(const char *folder, const char *stem, int component, int file_index, int C,
                          int nBp, int wl, int32_t *T, int base)
{
    static const char *suffix[4] = { ".txt_", "R.txt_", "G.txt_", "B.txt_" };
    char path[1024];
    size_t fl = strlen(folder);
    const char *sep = (fl && folder[fl - 1] != '/') ? "/" : "";
    snprintf(path, sizeof path, "%s%s%s%s%d", folder, sep, stem, suffix[component & 3], file_index);
    FILE *f = fopen(path, "rb");
    if (!f) return -1;
    int i = base, prev = -1, lvl, sb, bp, v[16];
    for (;;) {
        if (fscanf(f, "%d %d %d :", &lvl, &sb, &bp) != 3) break;
        int ok = 1;
        for (int c = 0; c < C; c++)
            if (fscanf(f, "%d", &v[c]) != 1) { ok = 0; break; }
        if (!ok) break;
        if (bp <= prev) {
            for (int z = 0; z < (nBp - prev - 1) * C; z++) T[i + prev * C + z + C] = 64;
            i += nBp * C;
        }
        if ((lvl + 1) > wl && sb > 0) break;
        prev = bp;
        for (int c = 0; c < C; c++) T[i + bp * C + c] = v[c];
    }
    fclose(f);
    return 0;
}

/* k > 0 (Engines/Engine.cu:12-56): the tables of files _0 .. _(n_tables-1) are laid out back to
 * back, table j at offset j * (n_ref + n_sig + n_sign) == j * _LUTPointerSizePerS
 * (BPC/BPCEngine.cu:333,1959-1961).  n_tables <= 0 means "all AMOUNT_OF_BITPLANE_FILES". */
int po_lut_load_k(const char *folder, int component, int wl, int fill, int n_tables, po_lut *l)
{
    memset(l, 0, sizeof *l);
    if (po_lut_header(folder, l)) return -1;
    l->wl = wl;
    if (n_tables <= 0) n_tables = l->n_bp_files;
    if (n_tables < 1) n_tables = 1;
    l->n_tables = n_tables;
    int nBp = l->n_bitplanes, nS = l->n_subbands;
    /* section sizes, IOManager.ipp:431-433 */
    l->n_ref = nS * nBp * l->ctx_ref * wl + nBp * l->ctx_ref;
    l->n_sig = nS * nBp * l->ctx_sig * wl + nBp * l->ctx_sig;
    l->n_sign = nS * nBp * l->ctx_sign * wl + nBp * l->ctx_sign;
    int total = l->n_ref + l->n_sig + l->n_sign;
    /* generous tail: a group-change fill on the very last group may write one group past it */
    size_t n = (size_t)total * (size_t)n_tables + (size_t)nBp * 16;
    l->table = (int32_t *)malloc(n * sizeof(int32_t));
    if (!l->table) return -3;
    for (size_t i = 0; i < n; i++) l->table[i] = fill;
    /* sections are loaded in order ref, sig, sign; a later section's group-change fill can spill
     * into nothing (it stays inside its own section for the shipped files). */
    for (int j = 0; j < n_tables; j++) {
        int32_t *T = l->table + (size_t)j * (size_t)total;
        if (po_lut_section(folder, "ref", component, j, l->ctx_ref, nBp, wl, T, 0)) return -4;
        if (po_lut_section(folder, "sig", component, j, l->ctx_sig, nBp, wl, T, l->n_ref)) return -5;
        if (po_lut_section(folder, "sign", component, j, l->ctx_sign, nBp, wl, T, l->n_ref + l->n_sig))
            return -6;
    }
    return 0;
}

/* k = 0 (Engines/Engine.cu:101-141): file _0 only */
int po_lut_load(const char *folder, int component, int wl, int fill, po_lut *l)
{
    return po_lut_load_k(folder, component, wl, fill, 1, l);
}

/* -cp 3, k = 0: the table is [ref | sig | sign | cp_sig | cp_sign]; the last two sections have the
 * geometry of sig and sign and are parsed by the same loop (IO/IOManager.ipp:539-606: bases
 * refinementNumber + significanceNumber + signNumber and that + significanceNumber). */
int po_lut_load_cp(const char *folder, int component, int wl, int fill, int cp, po_lut *l)
{
    if (cp != 3) { int rc = po_lut_load_k(folder, component, wl, fill, 1, l); if (!rc) l->cp = 2; return rc; }
    memset(l, 0, sizeof *l);
    if (po_lut_header(folder, l)) return -1;
    l->wl = wl; l->n_tables = 1; l->cp = 3;
    int nBp = l->n_bitplanes, nS = l->n_subbands;
    l->n_ref = nS * nBp * l->ctx_ref * wl + nBp * l->ctx_ref;
    l->n_sig = nS * nBp * l->ctx_sig * wl + nBp * l->ctx_sig;
    l->n_sign = nS * nBp * l->ctx_sign * wl + nBp * l->ctx_sign;
    size_t n = (size_t)l->n_ref + 2 * ((size_t)l->n_sig + l->n_sign) + (size_t)nBp * 16;
    l->table = (int32_t *)malloc(n * sizeof(int32_t));
    if (!l->table) return -3;
    for (size_t i = 0; i < n; i++) l->table[i] = fill;
    int32_t *T = l->table;
    const int b3 = l->n_ref + l->n_sig + l->n_sign;
    if (po_lut_section(folder, "ref", component, 0, l->ctx_ref, nBp, wl, T, 0)) return -4;
    if (po_lut_section(folder, "sig", component, 0, l->ctx_sig, nBp, wl, T, l->n_ref)) return -5;
    if (po_lut_section(folder, "sign", component, 0, l->ctx_sign, nBp, wl, T, l->n_ref + l->n_sig)) return -6;
    if (po_lut_section(folder, "cp_sig", component, 0, l->ctx_sig, nBp, wl, T, b3)) return -7;
    if (po_lut_section(folder, "cp_sign", component, 0, l->ctx_sign, nBp, wl, T, b3 + l->n_sig)) return -8;
    return 0;
}

/* ints of one table of the array */
static int po_lut_table_ints(const po_lut *l)
{
    return l->n_ref + (l->cp == 3 ? 2 : 1) * (l->n_sig + l->n_sign);
}

void po_lut_free(po_lut *l) { free(l->table); l->table = NULL; }

/* ------------------------------------------------------------------------------------------ */
/* BPC-PaCo, 2 coding passes, k = 0                                                             */
/* ------------------------------------------------------------------------------------------ */

/* BPC/BPCEngine.cu:143-170 */
void po_find_subband(int x, int y, int AW, int AH, int wl, int *level, int *sb)
{
    for (int a = 1; a <= wl; a++) {
        int cx = x >= (AW >> a), cy = y >= (AH >> a);
        if (cx || cy) {
            *level = a - 1;
            *sb = cx ? (cy ? 2 : 0) : 1;
            return;
        }
    }
    *level = wl;
    *sb = 0;
}

typedef struct {
    uint32_t TD[32][128];        /* coefficient words, BPCEngine.cu:1969; layout SURVEY a6 */
    uint32_t L[32], S[32];       /* interval lower / size, :1673-1675 */
    int slot[32];                /* reserved codeword, relative to staging[1] */
    uint32_t cw[32];             /* decoder: current codeword */
    int count;                   /* codeStreamShared[warp], :1996 */
    int ref_p[32], sig_p[32], sign_p[32];   /* LUT pointers per lane, :329-350 */
    const po_lut *lut;
    int lut_total;               /* ints in the whole table array (n_tables tables) */
    int lut_off;                 /* s * _LUTPointerSizePerS of this codeblock, BPCEngine.cu:333 */
    int32_t *stage;              /* this codeblock's 4096 ints */
} po_cb;

static int po_lut_at(const po_cb *cb, int idx)
{
    /* raw index into [ref|sig|sign] exactly as the reference; clamped only to stay in memory */
    if (idx < 0) idx = 0;
    if (idx >= cb->lut_total) idx = cb->lut_total - 1;
    return cb->lut->table[idx];
}

/* BPCEngine.cu:329-350; LUTOffset = s * _LUTPointerSizePerS is cb->lut_off (0 when k = 0) */
static void po_lut_init(po_cb *cb, int t, int level, int sb, int msb)
{
    const po_lut *l = cb->lut;
    int nS = l->n_subbands, nB = l->n_bitplanes;
    cb->ref_p[t] = level * nS * nB * l->ctx_ref + sb * nB * l->ctx_ref + msb * l->ctx_ref + cb->lut_off;
    cb->sig_p[t] = level * nS * nB * l->ctx_sig + sb * nB * l->ctx_sig + msb * l->ctx_sig + l->n_ref +
                   cb->lut_off;
    cb->sign_p[t] = level * nS * nB * l->ctx_sign + sb * nB * l->ctx_sign + msb * l->ctx_sign +
                    l->n_ref + l->n_sig + cb->lut_off;
}

/* L2Norm, BPC/BPCEngine.cuh:158-169 (columns LL, HL, LH, HH; row = level) */
static const float PO_L2NORM[10][4] = {
    { 1.965908f, 1.0112865f, 1.0112865f, 0.52021784f },
    { 4.1224113f, 1.9968134f, 1.9968134f, 0.96721643f },
    { 8.416739f, 4.1833673f, 4.1833673f, 2.0792568f },
    { 16.935543f, 8.534108f, 8.534108f, 4.3004827f },
    { 33.924816f, 17.166693f, 17.166693f, 8.686718f },
    { 67.87687f, 34.385098f, 34.385098f, 17.41882f },
    { 135.76744f, 68.7964f, 68.7964f, 34.860676f },
    { 271.5416f, 137.60588f, 137.60588f, 69.73287f },
    { 543.0866f, 275.21814f, 275.21814f, 139.47136f },
    { 1086.1624f, 550.43286f, 550.43286f, 278.94202f }
};

/* consecutiveBitplanes, Encode BPCEngine.cu:1684-1692 / Decode :1794-1802: IEEE float division and
 * product, floor, max 0.  (The reference is built -use_fast_math, so its quotient may differ in the
 * last place: parity of k > 0 with the CUDA binary is unpinned like everything else.)  The level /
 * subband are those of the codeblock's lane 0: the reference evaluates them per lane, which for a
 * codeblock straddling subbands would make the warp's plane loops diverge around full-mask shuffles
 * (undefined there); frames the CLI accepts at the default wl never straddle. */
int po_consecutive_bitplanes(int msb, float k, int level, int sb, int wl)
{
    if (!(k > 0.0f)) return 0;
    int lv = level > 9 ? 9 : level;
    float nrm = (wl == level) ? PO_L2NORM[(lv - 1) > 0 ? (lv - 1) : 0][0] : PO_L2NORM[lv][3 - sb];
    float q = k / nrm;
    int c = (int)floorf((float)msb * q);
    return c > 0 ? c : 0;
}

/* BPCEngine.cu:252-293 */
static int po_sign_ctx_hv(int h, int v)
{
    if (h == 0) return v == 0 ? 0 : (v > 0 ? 2 : 3);
    if (h > 0) return v == 0 ? 4 : (v > 0 ? 6 : 0);
    return v == 0 ? 5 : (v > 0 ? 1 : 7);
}

/* BPCEngine.cu:296-308 */
static int po_sign_ctx(uint32_t up, uint32_t left, uint32_t right, uint32_t bottom)
{
#define PO_CONTRIB(w) (((w) >> 31) == 0 ? 0 : (((w) & 1u) ? -1 : 1))
    int h = PO_CONTRIB(left) + PO_CONTRIB(right);
    int v = PO_CONTRIB(up) + PO_CONTRIB(bottom);
#undef PO_CONTRIB
    return po_sign_ctx_hv(h, v);
}

/* Grid view of the 32 lanes x 128 words: word at (row, col), col = 2*lane + side; outside the
 * codeblock = 0 (BPCEngine.cu:465-484 for lanes 0/31, :785,831 for rows 0/63). */
static uint32_t po_w(const po_cb *cb, int row, int col)
{
    if (row < 0 || row > 63 || col < 0 || col > 63) return 0;
    return cb->TD[col >> 1][2 * row + (col & 1)];
}

/* One call site of arithmeticEncoder (BPCEngine.cu:371-399) executed in lock step by the lanes
 * in `act`: lanes whose interval is exhausted reserve slots in ascending lane order
 * (__activemask/__popc, :378-383), then every active lane codes its symbol. */
/* Statistics of the lock-step scan (tools/lockstep_stats.py: how full the call sites are): sites with at least one
 * coding lane, coding lanes, sites at which some lane starts a codeword.  Relaxed atomics: the codeblocks run under OpenMP. */
static unsigned long long po_stat[3];
/* (counted only between po_stats_reset and po_stats_get: three shared atomics at every call site of sixteen OpenMP
 * threads cost the timed CPU baseline five sixths of its rate) */
static int po_stat_on;
void po_stats_reset(void) { po_stat[0] = po_stat[1] = po_stat[2] = 0; po_stat_on = 1; }
void po_stats_get(unsigned long long *out) { out[0] = po_stat[0]; out[1] = po_stat[1]; out[2] = po_stat[2]; po_stat_on = 0; }

static void po_enc_site(po_cb *cb, const uint8_t *act, const uint8_t *sym, const int *prob)
{
    int r = 0;
    if (po_stat_on) {
        int n = 0;
        for (int t = 0; t < 32; t++) n += act[t] ? 1 : 0;
        if (n) {
            __atomic_fetch_add(&po_stat[0], 1ull, __ATOMIC_RELAXED);
            __atomic_fetch_add(&po_stat[1], (unsigned long long)n, __ATOMIC_RELAXED);
        }
    }
    for (int t = 0; t < 32; t++)
        if (act[t] && cb->S[t] == 0) {
            cb->L[t] = 0;
            cb->S[t] = 0xFFFFu;
            int s = r + cb->count;
            cb->slot[t] = s > 4094 ? 4094 : s;
            r++;
        }
    if (r) {
        int c = cb->count + r;
        cb->count = c > 4095 ? 4095 : c;
        if (po_stat_on) __atomic_fetch_add(&po_stat[2], 1ull, __ATOMIC_RELAXED);
    }
    int prec = cb->lut->precision;
    for (int t = 0; t < 32; t++)
        if (act[t]) {
            uint32_t a = ((cb->S[t] * (uint32_t)prob[t]) >> prec) + sym[t];
            if (sym[t] == 0) cb->S[t] = a;
            else { cb->S[t] -= a; cb->L[t] += a; }
            if (cb->S[t] == 0) cb->stage[1 + cb->slot[t]] = (int32_t)cb->L[t];
        }
}

/* arithmeticDecoder call site (BPCEngine.cu:405-442) */
static void po_dec_site(po_cb *cb, const uint8_t *act, uint8_t *sym, const int *prob)
{
    int r = 0;
    for (int t = 0; t < 32; t++)
        if (act[t] && cb->S[t] == 0) {
            cb->L[t] = 0;
            cb->S[t] = 0xFFFFu;
            int s = r + cb->count;
            s = s > 4094 ? 4094 : s;
            cb->cw[t] = (uint32_t)cb->stage[1 + s];
            r++;
        }
    if (r) { int c = cb->count + r; cb->count = c > 4095 ? 4095 : c; }
    int prec = cb->lut->precision;
    for (int t = 0; t < 32; t++)
        if (act[t]) {
            uint32_t a = ((cb->S[t] * (uint32_t)prob[t]) >> prec) + 1u;
            uint32_t a2 = cb->L[t] + a;
            if (cb->cw[t] >= a2) { cb->S[t] -= a; cb->L[t] = a2; sym[t] = 1; }
            else { cb->S[t] = a - 1u; sym[t] = 0; }
        }
}

/* Significance propagation pass over one bit-plane: SPPEncoderLauncher BPCEngine.cu:770-843 /
 * SPPEncoder :490-516 (encode), SPPDecoderLauncher + SPPDecoder :559-594 (decode).
 * Row by row; all 32 lanes do their left column, then all 32 their right column; inside each of
 * the two, the significance call site precedes the sign call site. */
static void po_spp(po_cb *cb, int bp, int decode, uint32_t mask)
{
    uint8_t act[32], sym[32], act2[32], sym2[32];
    int prob[32], prob2[32], sctx[32];
    for (int i = 0; i < 64; i++)
        for (int side = 0; side < 2; side++) {
            for (int t = 0; t < 32; t++) {
                int col = 2 * t + side;
                uint32_t w = cb->TD[t][2 * i + side];
                act[t] = (uint8_t)!(w >> 31);
                act2[t] = 0;
                if (!act[t]) continue;
                uint32_t n1 = po_w(cb, i - 1, col - 1), n2 = po_w(cb, i - 1, col),
                         n3 = po_w(cb, i - 1, col + 1), n4 = po_w(cb, i, col - 1),
                         n5 = po_w(cb, i, col + 1), n6 = po_w(cb, i + 1, col - 1),
                         n7 = po_w(cb, i + 1, col), n8 = po_w(cb, i + 1, col + 1);
                /* computeContext :222-230 */
                int ctx = (int)((n1 >> 31) + (n2 >> 31) + (n3 >> 31) + (n4 >> 31) + (n5 >> 31) +
                                (n6 >> 31) + (n7 >> 31) + (n8 >> 31));
                prob[t] = po_lut_at(cb, cb->sig_p[t] + ctx);
                sym[t] = (uint8_t)((w >> (bp + 1)) & 1u);
                sctx[t] = po_sign_ctx(n2, n4, n5, n7);
            }
            if (decode) po_dec_site(cb, act, sym, prob);
            else po_enc_site(cb, act, sym, prob);
            for (int t = 0; t < 32; t++) {
                if (!act[t] || sym[t] != 1) continue;
                uint32_t *w = &cb->TD[t][2 * i + side];
                if (decode) *w |= mask;                       /* :577 */
                *w |= (1u << 31);                             /* :507 / :579 */
                *w |= ((uint32_t)bp << 24);                   /* :509 / :581 */
                act2[t] = 1;
                prob2[t] = po_lut_at(cb, cb->sign_p[t] + (sctx[t] >> 1));
                sym2[t] = (uint8_t)((((*w) & 1u) == (uint32_t)(sctx[t] & 1)) ? 0 : 1);  /* :513 */
            }
            if (decode) {
                po_dec_site(cb, act2, sym2, prob2);
                for (int t = 0; t < 32; t++)
                    if (act2[t]) {
                        uint32_t s = ((sym2[t] & 1u) == (uint32_t)(sctx[t] & 1)) ? 0u : 1u;  /* :587 */
                        cb->TD[t][2 * i + side] |= s;
                    }
            } else {
                po_enc_site(cb, act2, sym2, prob2);
            }
        }
}

/* Magnitude refinement pass: MRPEncoderLauncher BPCEngine.cu:1249-1261 / MRPEncoder :726-736,
 * MRPDecoder :743-762. */
static void po_mrp(po_cb *cb, int bp, int decode, uint32_t mask)
{
    uint8_t act[32], sym[32];
    int prob[32];
    for (int i = 0; i < 64; i++)
        for (int side = 0; side < 2; side++) {
            for (int t = 0; t < 32; t++) {
                uint32_t *w = &cb->TD[t][2 * i + side];
                act[t] = (uint8_t)(((*w) >> 29) & 1u);
                if (act[t]) {
                    sym[t] = (uint8_t)(((*w) >> (bp + 1)) & 1u);
                    prob[t] = po_lut_at(cb, cb->ref_p[t]);
                } else if ((*w) >> 31) {
                    *w |= (1u << 29);
                }
            }
            if (decode) {
                po_dec_site(cb, act, sym, prob);
                for (int t = 0; t < 32; t++)
                    if (act[t]) {
                        uint32_t *w = &cb->TD[t][2 * i + side];
                        *w &= ~mask;                                              /* :755 */
                        *w |= (mask & ((((uint32_t)sym[t] << 1) + 1u) << bp));    /* :757 */
                    }
            } else {
                po_enc_site(cb, act, sym, prob);
            }
        }
}

/* ------------------------------------------------------------------------------------------ */
/* -cp 3: three coding passes (deprecated in the reference, IO/CommandLineParser.cpp:34)        */
/* ------------------------------------------------------------------------------------------ */
/* Bit 30 of a coefficient word is the cleanup flag: set on every word before coding
 * (readCoefficients3CP BPCEngine.cu:65-90 / initializeCoefficients3CP :124-137), set again by the
 * significance pass on a coefficient it leaves to the cleanup pass, cleared by the cleanup pass. */

/* Significance propagation pass of the 3-pass mode: SPPEncoderLauncher3CP BPCEngine.cu:850-925 +
 * SPPEncoder3CP :521-553 / SPPDecoderLauncher3CP :1010-1085 + SPPDecoder3CP :599-640.  Scan order and
 * neighbour exchange as in po_spp; an insignificant coefficient is coded only if one of its eight
 * neighbours is significant when the scan reaches it, otherwise it is flagged for the cleanup pass. */
static void po_spp3(po_cb *cb, int bp, int decode, uint32_t mask)
{
    uint8_t act[32], sym[32], act2[32], sym2[32];
    int prob[32], prob2[32], sctx[32];
    for (int i = 0; i < 64; i++)
        for (int side = 0; side < 2; side++) {
            for (int t = 0; t < 32; t++) {
                int col = 2 * t + side;
                uint32_t *w = &cb->TD[t][2 * i + side];
                act[t] = 0; act2[t] = 0; sym[t] = 0;
                if ((*w) >> 31) continue;
                uint32_t n1 = po_w(cb, i - 1, col - 1), n2 = po_w(cb, i - 1, col),
                         n3 = po_w(cb, i - 1, col + 1), n4 = po_w(cb, i, col - 1),
                         n5 = po_w(cb, i, col + 1), n6 = po_w(cb, i + 1, col - 1),
                         n7 = po_w(cb, i + 1, col), n8 = po_w(cb, i + 1, col + 1);
                int ctx = (int)((n1 >> 31) + (n2 >> 31) + (n3 >> 31) + (n4 >> 31) + (n5 >> 31) +
                                (n6 >> 31) + (n7 >> 31) + (n8 >> 31));
                if (ctx == 0) { *w |= (1u << 30); continue; }     /* :551 / :638 */
                act[t] = 1;
                prob[t] = po_lut_at(cb, cb->sig_p[t] + ctx);
                sym[t] = (uint8_t)(((*w) >> (bp + 1)) & 1u);
                sctx[t] = po_sign_ctx(n2, n4, n5, n7);
            }
            if (decode) po_dec_site(cb, act, sym, prob);
            else po_enc_site(cb, act, sym, prob);
            for (int t = 0; t < 32; t++) {
                if (!act[t] || sym[t] != 1) continue;
                uint32_t *w = &cb->TD[t][2 * i + side];
                if (decode) *w |= mask;                       /* :620 */
                *w |= (1u << 31);                             /* :541 / :622 */
                *w |= ((uint32_t)bp << 24);                   /* :543 / :624 */
                act2[t] = 1;
                prob2[t] = po_lut_at(cb, cb->sign_p[t] + (sctx[t] >> 1));
                sym2[t] = (uint8_t)((((*w) & 1u) == (uint32_t)(sctx[t] & 1)) ? 0 : 1);  /* :547 */
            }
            if (decode) {
                po_dec_site(cb, act2, sym2, prob2);
                for (int t = 0; t < 32; t++)
                    if (act2[t]) {
                        uint32_t s = ((sym2[t] & 1u) == (uint32_t)(sctx[t] & 1)) ? 0u : 1u;  /* :630 */
                        cb->TD[t][2 * i + side] |= s;
                    }
            } else {
                po_enc_site(cb, act2, sym2, prob2);
            }
        }
}

/* Cleanup pass: CPEncoderLauncher BPCEngine.cu:1090-1165 + CPEncoder :645-680 / CPDecoderLauncher
 * :1170-1243 + CPDecoder :686-719.  Codes the flagged coefficients with the cp_sig / cp_sign tables (the
 * ordinary pointers + section sizes of sig and sign, Encode3CP :1744-1748); a coefficient that becomes
 * significant here is at once eligible for refinement (bit 29). */
static void po_cp(po_cb *cb, int bp, int decode, uint32_t mask)
{
    uint8_t act[32], sym[32], act2[32], sym2[32];
    int prob[32], prob2[32], sctx[32];
    const int aux = cb->lut->n_sig + cb->lut->n_sign;
    for (int i = 0; i < 64; i++)
        for (int side = 0; side < 2; side++) {
            for (int t = 0; t < 32; t++) {
                int col = 2 * t + side;
                uint32_t w = cb->TD[t][2 * i + side];
                act[t] = (uint8_t)((w >> 30) & 1u); act2[t] = 0; sym[t] = 0;
                if (!act[t]) continue;
                uint32_t n1 = po_w(cb, i - 1, col - 1), n2 = po_w(cb, i - 1, col),
                         n3 = po_w(cb, i - 1, col + 1), n4 = po_w(cb, i, col - 1),
                         n5 = po_w(cb, i, col + 1), n6 = po_w(cb, i + 1, col - 1),
                         n7 = po_w(cb, i + 1, col), n8 = po_w(cb, i + 1, col + 1);
                int ctx = (int)((n1 >> 31) + (n2 >> 31) + (n3 >> 31) + (n4 >> 31) + (n5 >> 31) +
                                (n6 >> 31) + (n7 >> 31) + (n8 >> 31));
                prob[t] = po_lut_at(cb, cb->sig_p[t] + aux + ctx);
                sym[t] = (uint8_t)((w >> (bp + 1)) & 1u);
                sctx[t] = po_sign_ctx(n2, n4, n5, n7);
            }
            if (decode) po_dec_site(cb, act, sym, prob);
            else po_enc_site(cb, act, sym, prob);
            for (int t = 0; t < 32; t++) {
                if (!act[t]) continue;
                uint32_t *w = &cb->TD[t][2 * i + side];
                *w &= 0xBFFFFFFFu;                            /* :664 / :703 */
                if (sym[t] != 1) continue;
                if (decode) *w |= mask;                       /* :708 */
                *w |= (1u << 31) | (1u << 29);                /* :669-670 / :710-711 */
                *w |= ((uint32_t)bp << 24);
                act2[t] = 1;
                prob2[t] = po_lut_at(cb, cb->sign_p[t] + aux + (sctx[t] >> 1));
                sym2[t] = (uint8_t)((((*w) & 1u) == (uint32_t)(sctx[t] & 1)) ? 0 : 1);
            }
            if (decode) {
                po_dec_site(cb, act2, sym2, prob2);
                for (int t = 0; t < 32; t++)
                    if (act2[t]) {
                        uint32_t s = ((sym2[t] & 1u) == (uint32_t)(sctx[t] & 1)) ? 0u : 1u;
                        cb->TD[t][2 * i + side] |= s;
                    }
            } else {
                po_enc_site(cb, act2, sym2, prob2);
            }
        }
}

/* computeContextBulk BPCEngine.cu:236-243 */
static int po_ctx_bulk(const uint32_t n[8], int B)
{
    int c = 0;
    for (int j = 0; j < 8; j++) c += (int)(((n[j] >> 24) & 31u) >= (uint32_t)B);
    return c;
}

/* computeSignContextBulk BPCEngine.cu:311-323 */
static int po_sign_ctx_bulk(uint32_t up, uint32_t left, uint32_t right, uint32_t bottom, int q)
{
#define PO_CONTRIB_B(w) ((((w) >> 31) == 0 || (((w) >> 24) & 31u) < (uint32_t)q) ? 0 : (((w) & 1u) ? -1 : 1))
    int h = PO_CONTRIB_B(left) + PO_CONTRIB_B(right);
    int v = PO_CONTRIB_B(up) + PO_CONTRIB_B(bottom);
#undef PO_CONTRIB_B
    return po_sign_ctx_hv(h, v);
}

/* Complexity-scalable bulk mode: encodeBulkMode / decodeBulkMode BPCEngine.cu:1640-1662,
 * encode/decodeLeft/RightCoefficients :1320-1448 / :1506-1634, encode/decodeBulkProcessing
 * :1285-1314 / :1454-1500.  Row by row; all lanes their left coefficient, then all lanes their
 * right one; per coefficient ALL planes B..0 in one go: refinement call site (already significant),
 * significance call site, sign call site -- three lock-step call sites per plane.  The 8-neighbour
 * context is taken ONCE per coefficient, from the words as they stand when the row/side is reached
 * (computeContextBulk for B != 0: neighbours whose significance plane field is >= B; plain
 * computeContext for B == 0), the four sign neighbours are captured at the same moment (by value,
 * :1345,:1374...) and filtered per plane by computeSignContextBulk.  LUT entries of plane q are the
 * plane-B pointers minus ctx * (B - q) (:1293,:1298,:1308). */
static void po_bulk(po_cb *cb, int B, int decode, uint32_t mask0)
{
    uint8_t act[32], sym[32], newsig[32];
    int prob[32], ctx[32], sctx[32];
    uint32_t nb[32][4];
    const po_lut *l = cb->lut;
    for (int i = 0; i < 64; i++)
        for (int side = 0; side < 2; side++) {
            for (int t = 0; t < 32; t++) {
                int col = 2 * t + side;
                uint32_t n[8] = { po_w(cb, i - 1, col - 1), po_w(cb, i - 1, col), po_w(cb, i - 1, col + 1),
                                  po_w(cb, i, col - 1), po_w(cb, i, col + 1), po_w(cb, i + 1, col - 1),
                                  po_w(cb, i + 1, col), po_w(cb, i + 1, col + 1) };
                if (B != 0) ctx[t] = po_ctx_bulk(n, B);
                else {
                    ctx[t] = 0;
                    for (int j = 0; j < 8; j++) ctx[t] += (int)(n[j] >> 31);
                }
                nb[t][0] = n[1]; nb[t][1] = n[3]; nb[t][2] = n[4]; nb[t][3] = n[6];
            }
            uint32_t mask = mask0;
            for (int q = B; q >= 0; q--) {
                int d = B - q;
                /* refinement call site: coefficients that are significant now (:1291 / :1463) */
                for (int t = 0; t < 32; t++) {
                    uint32_t w = cb->TD[t][2 * i + side];
                    act[t] = (uint8_t)(w >> 31);
                    sym[t] = (uint8_t)((w >> (q + 1)) & 1u);
                    prob[t] = po_lut_at(cb, cb->ref_p[t] - l->ctx_ref * d);
                }
                if (decode) {
                    po_dec_site(cb, act, sym, prob);
                    for (int t = 0; t < 32; t++)
                        if (act[t]) {
                            uint32_t *w = &cb->TD[t][2 * i + side];
                            *w &= ~mask;                                              /* :1467 */
                            *w |= (mask & ((((uint32_t)sym[t] << 1) + 1u) << q));     /* :1469 */
                        }
                } else {
                    po_enc_site(cb, act, sym, prob);
                }
                /* significance call site: the others (:1296 / :1472) */
                for (int t = 0; t < 32; t++) {
                    uint32_t w = cb->TD[t][2 * i + side];
                    act[t] = (uint8_t)!(w >> 31);
                    sym[t] = (uint8_t)((w >> (q + 1)) & 1u);
                    prob[t] = po_lut_at(cb, cb->sig_p[t] + ctx[t] - l->ctx_sig * d);
                }
                if (decode) po_dec_site(cb, act, sym, prob);
                else po_enc_site(cb, act, sym, prob);
                for (int t = 0; t < 32; t++) {
                    newsig[t] = (uint8_t)(act[t] && sym[t] == 1);
                    if (!newsig[t]) continue;
                    uint32_t *w = &cb->TD[t][2 * i + side];
                    if (decode) *w |= mask;                       /* :1478 */
                    *w |= (1u << 31);                             /* :1302 / :1480 */
                    *w |= ((uint32_t)q << 24);                    /* :1304 / :1482 */
                    sctx[t] = po_sign_ctx_bulk(nb[t][0], nb[t][1], nb[t][2], nb[t][3], q);
                    prob[t] = po_lut_at(cb, cb->sign_p[t] + (sctx[t] >> 1) - l->ctx_sign * d);
                    sym[t] = (uint8_t)((((*w) & 1u) == (uint32_t)(sctx[t] & 1)) ? 0 : 1);   /* :1308 */
                }
                /* sign call site */
                if (decode) {
                    po_dec_site(cb, newsig, sym, prob);
                    for (int t = 0; t < 32; t++)
                        if (newsig[t]) {
                            uint32_t sg = ((sym[t] & 1u) == (uint32_t)(sctx[t] & 1)) ? 0u : 1u;   /* :1488 */
                            cb->TD[t][2 * i + side] |= sg;
                        }
                } else {
                    po_enc_site(cb, newsig, sym, prob);
                }
                mask >>= 1;                                       /* :1494-1496 */
                if (q == 1) mask = 0x2u;
            }
        }
}

static void po_lut_step(po_cb *cb)
{
    for (int t = 0; t < 32; t++) {        /* updateLUTPointers :353-358 */
        cb->sig_p[t] -= cb->lut->ctx_sig;
        cb->sign_p[t] -= cb->lut->ctx_sign;
        cb->ref_p[t] -= cb->lut->ctx_ref;
    }
}

/* Encode BPCEngine.cu:1668-1721; cbp = consecutiveBitplanes (0 when k = 0) */
static void po_cb_encode(po_cb *cb, int msb, int cbp)
{
    for (int t = 0; t < 32; t++) { cb->L[t] = 0; cb->S[t] = 0; cb->slot[t] = -1; }
    int bp = msb;
    for (; bp >= cbp; bp--) {
        po_spp(cb, bp, 0, 0);
        po_mrp(cb, bp, 0, 0);
        po_lut_step(cb);
    }
    if (bp >= 0) po_bulk(cb, bp, 0, 0);   /* :1712-1716 */
    /* flush :1719 -- every lane stores its L into its current slot */
    for (int t = 0; t < 32; t++)
        if (cb->slot[t] >= 0) cb->stage[1 + cb->slot[t]] = (int32_t)cb->L[t];
}

/* Decode BPCEngine.cu:1777-1837 */
static void po_cb_decode(po_cb *cb, int msb, int cbp)
{
    for (int t = 0; t < 32; t++) { cb->L[t] = 0; cb->S[t] = 0; cb->cw[t] = 0; }
    uint32_t mask = 0x3u << msb;
    if (msb == 0) mask &= 0x2u;
    int bp = msb;
    for (; bp >= cbp; bp--) {
        po_spp(cb, bp, 1, mask);
        po_mrp(cb, bp, 1, mask);
        mask >>= 1;
        if (bp == 1) mask = 0x2u;
        po_lut_step(cb);
    }
    if (bp >= 0) po_bulk(cb, bp, 1, mask);   /* :1831-1835 */
}

/* Encode3CP BPCEngine.cu:1727-1776: the top plane takes the cleanup pass only */
static void po_cb_encode3(po_cb *cb, int msb)
{
    for (int t = 0; t < 32; t++) {
        cb->L[t] = 0; cb->S[t] = 0; cb->slot[t] = -1;
        for (int i = 0; i < 128; i++) cb->TD[t][i] |= (1u << 30);     /* readCoefficients3CP :84-86 */
    }
    int bp = msb;
    po_cp(cb, bp, 0, 0);
    bp--;
    po_lut_step(cb);
    for (; bp >= 0; bp--) {
        po_spp3(cb, bp, 0, 0);
        po_mrp(cb, bp, 0, 0);
        po_cp(cb, bp, 0, 0);
        po_lut_step(cb);
    }
    for (int t = 0; t < 32; t++)
        if (cb->slot[t] >= 0) cb->stage[1 + cb->slot[t]] = (int32_t)cb->L[t];
}

/* Decode3CP BPCEngine.cu:1844-1900 */
static void po_cb_decode3(po_cb *cb, int msb)
{
    for (int t = 0; t < 32; t++) {
        cb->L[t] = 0; cb->S[t] = 0; cb->cw[t] = 0;
        for (int i = 0; i < 128; i++) cb->TD[t][i] = (1u << 30);      /* initializeCoefficients3CP :124-137 */
    }
    uint32_t mask = 0x3u << msb;
    int bp = msb;
    if (bp == 0) mask &= 0x2u;
    po_cp(cb, bp, 1, mask);
    bp--;
    mask >>= 1;
    if (bp == 0) mask = 0x2u;
    po_lut_step(cb);
    for (; bp >= 0; bp--) {
        po_spp3(cb, bp, 1, mask);
        po_mrp(cb, bp, 1, mask);
        po_cp(cb, bp, 1, mask);
        mask >>= 1;
        if (bp == 1) mask = 0x2u;
        po_lut_step(cb);
    }
}

/* table index s = min(consecutiveBitplanes, MSB) (:1694-1697 / :1807-1810), clamped to the tables
 * that were loaded */
static void po_cb_select_table(po_cb *cb, int msb, int cbp, float k)
{
    int s = 0;
    if (k > 0.0f) s = cbp < msb ? cbp : msb;
    int nt = cb->lut->n_tables > 0 ? cb->lut->n_tables : 1;
    if (s > nt - 1) s = nt - 1;
    cb->lut_off = s * po_lut_table_ints(cb->lut);
}

static int po_msb_of(const po_cb *cb)
{
    /* findMSB :176-192 -- 32 - ffs(brev(max)); 32 when every magnitude is 0 */
    uint32_t m = 0;
    for (int t = 0; t < 32; t++)
        for (int i = 0; i < 128; i++) m |= cb->TD[t][i] >> 1;
    if (m == 0) return 32;
    int k = 31;
    while (!((m >> k) & 1u)) k--;
    return k;
}

/* after Encode: sizeArray + expansion fallback, kernelBPCCoder BPCEngine.cu:2002-2024 */
static int po_cb_finish_encode(po_cb *cb)
{
    int size = cb->count + 1;
    if (size == PO_CB_WORDS)      /* expansionFix :1905-1912 */
        for (int t = 0; t < 32; t++)
            for (int i = 0; i < 128; i++)
                cb->stage[t * 128 + i] = (int32_t)(cb->TD[t][i] & 0xFFFFu);
    return size;
}

void po_bpc_encode(const void *coeffs, int is_float, int AW, int AH, int wl, const po_lut *lut,
                   int32_t *staging, int32_t *sizes)
{
    po_bpc_encode_k(coeffs, is_float, AW, AH, wl, lut, 0.0f, staging, sizes);
}

void po_bpc_encode_k(const void *coeffs, int is_float, int AW, int AH, int wl, const po_lut *lut, float k,
                     int32_t *staging, int32_t *sizes)
{
    int ncx = AW / PO_CB, ncy = AH / PO_CB;
    /* BPCEngine::deviceMemoryAllocator :2429-2441 -- staging memset to 0xFF */
    memset(staging, 0xFF, (size_t)AW * AH * sizeof(int32_t));
    _Pragma("omp parallel num_threads(po_threads)")
    {
    po_cb *cb = (po_cb *)malloc(sizeof(po_cb));
    cb->lut = lut;
    cb->lut_total = po_lut_table_ints(lut) * (lut->n_tables > 0 ? lut->n_tables : 1);
    cb->lut_off = 0;
    _Pragma("omp for schedule(dynamic, 4)")
    for (int id = 0; id < ncx * ncy; id++) {
            int cy = id / ncx, cx = id % ncx;
            cb->stage = staging + (size_t)id * PO_CB_WORDS;
            int level[32], sb[32];
            for (int t = 0; t < 32; t++) {
                /* per-lane subband: x = lane's first column, y = top row (:1975-1990) */
                po_find_subband(cx * 64 + 2 * t, cy * 64, AW, AH, wl, &level[t], &sb[t]);
                for (int i = 0; i < 64; i++)
                    for (int s = 0; s < 2; s++) {
                        size_t idx = (size_t)(cy * 64 + i) * AW + (size_t)(cx * 64 + 2 * t + s);
                        /* readCoefficients :41-63 -- (int) truncation, sign-magnitude word */
                        int32_t v = is_float ? (int32_t)((const float *)coeffs)[idx]
                                             : ((const int32_t *)coeffs)[idx];
                        uint32_t neg = v < 0;
                        uint32_t mag = (uint32_t)(v < 0 ? -(int64_t)v : (int64_t)v);
                        cb->TD[t][2 * i + s] = (mag << 1) + neg;
                    }
            }
            int msb = po_msb_of(cb);
            cb->count = 0;
            cb->stage[0] = msb;
            if (msb != 32) {
                int cbp = po_consecutive_bitplanes(msb, k, level[0], sb[0], wl);
                po_cb_select_table(cb, msb, cbp, k);
                for (int t = 0; t < 32; t++) po_lut_init(cb, t, level[t], sb[t], msb);
                if (lut->cp == 3) po_cb_encode3(cb, msb);        /* kernelBPCCoder3CP :2029-2121 (k is ignored there) */
                else po_cb_encode(cb, msb, cbp);
            }
            sizes[id] = po_cb_finish_encode(cb);
    }
    free(cb);
    }
}

int po_bpc_encode_block_uniform(const int32_t *block, int level, int sb, int wl, const po_lut *lut,
                                int32_t *staging4096)
{
    (void)wl;
    po_cb *cb = (po_cb *)malloc(sizeof(po_cb));
    cb->lut = lut;
    cb->lut_total = po_lut_table_ints(lut) * (lut->n_tables > 0 ? lut->n_tables : 1);
    cb->lut_off = 0;
    cb->stage = staging4096;
    memset(staging4096, 0xFF, PO_CB_WORDS * sizeof(int32_t));
    for (int t = 0; t < 32; t++)
        for (int i = 0; i < 64; i++)
            for (int s = 0; s < 2; s++) {
                int32_t v = block[i * 64 + 2 * t + s];
                uint32_t neg = v < 0;
                uint32_t mag = (uint32_t)(v < 0 ? -v : v);
                cb->TD[t][2 * i + s] = (mag << 1) + neg;
            }
    int msb = po_msb_of(cb);
    cb->count = 0;
    cb->stage[0] = msb;
    if (msb != 32) {
        for (int t = 0; t < 32; t++) po_lut_init(cb, t, level, sb, msb);
        po_cb_encode(cb, msb, 0);
    }
    int size = po_cb_finish_encode(cb);
    free(cb);
    return size;
}

/* kernelBPCDecoder BPCEngine.cu:2126-2215, writeCoefficients :94-111, copyEntireCodeblock
 * :1915-1922 */
void po_bpc_decode(const int32_t *staging, const int32_t *sizes, int AW, int AH, int wl,
                   const po_lut *lut, int32_t *coeffs)
{
    po_bpc_decode_k(staging, sizes, AW, AH, wl, lut, 0.0f, coeffs);
}

void po_bpc_decode_k(const int32_t *staging, const int32_t *sizes, int AW, int AH, int wl,
                     const po_lut *lut, float k, int32_t *coeffs)
{
    int ncx = AW / PO_CB, ncy = AH / PO_CB;
    po_cb *cb = (po_cb *)malloc(sizeof(po_cb));
    cb->lut = lut;
    cb->lut_total = po_lut_table_ints(lut) * (lut->n_tables > 0 ? lut->n_tables : 1);
    cb->lut_off = 0;
    for (int cy = 0; cy < ncy; cy++)
        for (int cx = 0; cx < ncx; cx++) {
            int id = cy * ncx + cx;
            cb->stage = (int32_t *)(staging + (size_t)id * PO_CB_WORDS);
            memset(cb->TD, 0, sizeof cb->TD);
            cb->count = 0;
            int msb = cb->stage[0];
            if (sizes[id] == PO_CB_WORDS) {
                for (int t = 0; t < 32; t++)
                    for (int i = 0; i < 128; i++) cb->TD[t][i] = (uint32_t)cb->stage[t * 128 + i];
            } else if (msb != 32) {
                int level0, sb0;
                po_find_subband(cx * 64, cy * 64, AW, AH, wl, &level0, &sb0);
                int cbp = po_consecutive_bitplanes(msb, k, level0, sb0, wl);
                po_cb_select_table(cb, msb, cbp, k);
                for (int t = 0; t < 32; t++) {
                    int level, sb;
                    po_find_subband(cx * 64 + 2 * t, cy * 64, AW, AH, wl, &level, &sb);
                    po_lut_init(cb, t, level, sb, msb);
                }
                if (lut->cp == 3) po_cb_decode3(cb, msb);        /* kernelBPCDecoder3CP :2221-2299 */
                else po_cb_decode(cb, msb, cbp);
            }
            for (int t = 0; t < 32; t++)
                for (int i = 0; i < 64; i++)
                    for (int s = 0; s < 2; s++) {
                        uint32_t w = cb->TD[t][2 * i + s];
                        int32_t v = (int32_t)((w & 0xFFFFFFu) >> 1);
                        if (w & 1u) v = -v;
                        coeffs[(size_t)(cy * 64 + i) * AW + (size_t)(cx * 64 + 2 * t + s)] = v;
                    }
        }
    free(cb);
}

/* ------------------------------------------------------------------------------------------ */
/* BitStreamBuilder                                                                             */
/* ------------------------------------------------------------------------------------------ */

/* BitStreamBuilder/BitStreamBuilder.cpp:35-94 */
void po_header_pack(const po_header *h, uint16_t o[PO_HDR_SHORTS])
{
    o[0] = (uint16_t)(h->n_samples & 0xFFFFu);
    o[1] = (uint16_t)(h->n_samples >> 16);
    o[2] = (uint16_t)((h->cp == 2 ? 0 : 1) | (h->cb_height << 1) | (h->cb_width << 8) |
                      ((h->wl & 1) << 15));
    o[3] = (uint16_t)(((h->wl & 7) >> 1) | (h->bit_depth << 3) | ((h->lossy ? 1 : 0) << 10) |
                      ((h->qs_1e4 & 31) << 11));
    o[4] = (uint16_t)((h->qs_1e4 >> 5) | ((h->components & 127) << 9));
    o[5] = (uint16_t)((h->components >> 7) | ((h->is_rgb ? 1 : 0) << 7) | (h->height << 8));
    o[6] = (uint16_t)((h->height >> 8) | (h->endianess << 8) | (h->bps << 9) |
                      ((h->is_signed ? 1 : 0) << 14) | ((h->frames & 1) << 15));
    o[7] = (uint16_t)((h->frames >> 1) & 0xFFFF);
    o[8] = (uint16_t)h->k_1e3;
}

/* Engines/DecodingEngine.cu:567-585 */
void po_header_unpack(const uint16_t e[PO_HDR_SHORTS], po_header *h)
{
    h->n_samples = (uint32_t)e[0] | ((uint32_t)e[1] << 16);
    h->cp = (e[2] & 1) ? 3 : 2;
    h->cb_height = (e[2] >> 1) & 127;
    h->cb_width = (e[2] >> 8) & 127;
    h->wl = ((e[2] >> 15) & 1) | ((e[3] & 7) << 1);
    h->bit_depth = (e[3] >> 3) & 127;
    h->lossy = (e[3] >> 10) & 1;
    h->qs_1e4 = ((e[3] >> 11) & 31) | ((e[4] & 511) << 5);
    h->components = ((e[4] >> 9) & 127) | ((e[5] & 127) << 9);
    h->is_rgb = (e[5] >> 7) & 1;
    h->height = ((e[5] >> 8) & 255) | ((e[6] & 255) << 8);
    h->endianess = (e[6] >> 8) & 1;
    h->bps = (e[6] >> 9) & 31;
    h->is_signed = (e[6] >> 14) & 1;
    h->frames = ((e[6] >> 15) & 1) | ((int)e[7] << 1);
    h->k_1e3 = e[8];
}

/* total shorts = sum(len) + 9 + 2 nCB - nCB + 1 (BitStreamBuilder.cu:300-305) */
size_t po_bitstream_total(const int32_t *sizes, int n_cb)
{
    size_t s = 0;
    for (int i = 0; i < n_cb; i++) s += (size_t)sizes[i];
    return s + PO_HDR_SHORTS + (size_t)n_cb + 1;
}

/* createBitStream BitStreamBuilder.cpp:100-114; layout from buildBitStreamLUTBS
 * BitStreamBuilder.cu:106-137 + binarySearchLUTBS :33-101: out[9+2cb] = staging[cb*4096] (MSB),
 * out[10+2cb] = len, payload element j (1 <= j < len) of cb at 9 + 2 nCB + sum_{i<cb}(len_i-1) +
 * (j-1); everything else (incl. one trailing short) stays 0xFFFF (memset :274). */
size_t po_bitstream_pack(const int32_t *staging, const int32_t *sizes, int n_cb,
                         const uint16_t *header, uint16_t *out)
{
    size_t total = po_bitstream_total(sizes, n_cb);
    memset(out, 0xFF, total * sizeof(uint16_t));
    if (header) memcpy(out, header, PO_HDR_SHORTS * sizeof(uint16_t));
    size_t pos = PO_HDR_SHORTS + 2 * (size_t)n_cb;
    for (int cb = 0; cb < n_cb; cb++) {
        const int32_t *st = staging + (size_t)cb * PO_CB_WORDS;
        out[PO_HDR_SHORTS + 2 * cb] = (uint16_t)st[0];
        out[PO_HDR_SHORTS + 2 * cb + 1] = (uint16_t)sizes[cb];
        for (int j = 1; j < sizes[cb]; j++) out[pos++] = (uint16_t)st[j];
    }
    return total;
}

/* createCodeStream BitStreamBuilder.cpp:119-153 + buildCodeStreamLUTBS BitStreamBuilder.cu:142-171 */
void po_bitstream_unpack(const uint16_t *in, int n_cb, int32_t *staging, int32_t *sizes)
{
    memset(staging, 0xFF, (size_t)n_cb * PO_CB_WORDS * sizeof(int32_t));
    size_t pos = PO_HDR_SHORTS + 2 * (size_t)n_cb;
    for (int cb = 0; cb < n_cb; cb++) {
        int32_t *st = staging + (size_t)cb * PO_CB_WORDS;
        sizes[cb] = in[PO_HDR_SHORTS + 1 + 2 * cb];      /* retrieveSizeArray :119-129 */
        st[0] = in[PO_HDR_SHORTS + 2 * cb];
        for (int j = 1; j < sizes[cb]; j++) st[j] = in[pos++];
    }
}

/* ------------------------------------------------------------------------------------------ */
/* whole frame                                                                                  */
/* ------------------------------------------------------------------------------------------ */

size_t po_encode_frame(const uint8_t *frame, int W, int H, int wl, int lossy, float qs,
                       const po_lut *lut, int iter, int frames, uint16_t *out)
{
    return po_encode_frame_k(frame, W, H, wl, lossy, qs, 0.0f, lut, iter, frames, out);
}

/* k > 0: `lut` must hold the bit-plane tables (po_lut_load_k); the header carries (int)(k * 1000)
 * (BitStreamBuilder.cpp:82-93) and the decoding side works with that value / 1000.0
 * (Engines/DecodingEngine.cu:56,163). */
size_t po_encode_frame_k(const uint8_t *frame, int W, int H, int wl, int lossy, float qs, float k,
                         const po_lut *lut, int iter, int frames, uint16_t *out)
{
    int AW = po_pad_dim(W), AH = po_pad_dim(H);
    size_t P = (size_t)AW * AH, extra = po_dwt_extra(AW, AH, wl);
    int n_cb = (AW / PO_CB) * (AH / PO_CB);
    uint8_t *pad = (uint8_t *)malloc(P);
    int32_t *staging = (int32_t *)malloc(P * sizeof(int32_t));
    int32_t *sizes = (int32_t *)malloc((size_t)n_cb * sizeof(int32_t));
    if (po_pad_frame(frame, W, H, pad, AW, AH) != 0) { free(pad); free(staging); free(sizes); return 0; }
    if (!lossy) {
        int32_t *a = (int32_t *)malloc(P * sizeof(int32_t));
        int32_t *b = (int32_t *)malloc((P + extra) * sizeof(int32_t));
        po_level_shift_fwd_i32(pad, a, P, 8);
        po_dwt53_forward(a, b, AW, AH, wl);
        po_bpc_encode_k(b, 0, AW, AH, wl, lut, k, staging, sizes);
        free(a); free(b);
    } else {
        float *a = (float *)malloc(P * sizeof(float));
        float *b = (float *)malloc((P + extra) * sizeof(float));
        po_level_shift_fwd_f32(pad, a, P, 8);
        po_dwt97_forward(a, b, AW, AH, wl, qs);
        po_bpc_encode_k(b, 1, AW, AH, wl, lut, k, staging, sizes);
        free(a); free(b);
    }
    uint16_t hdr[PO_HDR_SHORTS];
    po_header h;
    memset(&h, 0, sizeof h);
    h.n_samples = (uint32_t)W * (uint32_t)H;
    h.cp = lut->cp == 3 ? 3 : 2; h.cb_height = 18; h.cb_width = 64; h.wl = wl; h.bit_depth = 8; h.lossy = lossy;
    h.qs_1e4 = (int)(qs * 10000); h.components = 1; h.is_rgb = 0; h.height = H; h.endianess = 0;
    h.bps = 8; h.is_signed = 0; h.frames = frames; h.k_1e3 = (int)(k * 1000);
    po_header_pack(&h, hdr);
    size_t total = po_bitstream_pack(staging, sizes, n_cb, iter == 0 ? hdr : NULL, out);
    free(pad); free(staging); free(sizes);
    return total;
}

int po_decode_frame(const uint16_t *stream, int W, int H, int wl, int lossy, float qs,
                    const po_lut *lut, uint8_t *frame_out)
{
    return po_decode_frame_k(stream, W, H, wl, lossy, qs, 0.0f, lut, frame_out);
}

int po_decode_frame_k(const uint16_t *stream, int W, int H, int wl, int lossy, float qs, float k,
                      const po_lut *lut, uint8_t *frame_out)
{
    int AW = po_pad_dim(W), AH = po_pad_dim(H);
    size_t P = (size_t)AW * AH, extra = po_dwt_extra(AW, AH, wl);
    int n_cb = (AW / PO_CB) * (AH / PO_CB);
    int32_t *staging = (int32_t *)malloc(P * sizeof(int32_t));
    int32_t *sizes = (int32_t *)malloc((size_t)n_cb * sizeof(int32_t));
    int32_t *coef = (int32_t *)malloc(P * sizeof(int32_t));
    po_bitstream_unpack(stream, n_cb, staging, sizes);
    po_bpc_decode_k(staging, sizes, AW, AH, wl, lut, k, coef);
    if (!lossy) {
        int32_t *img = (int32_t *)malloc((P + extra) * sizeof(int32_t));
        po_dwt53_inverse(coef, img, AW, AH, wl);
        po_level_shift_inv_i32(img + extra, P, 8);
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++)
                frame_out[(size_t)y * W + x] = (uint8_t)img[extra + (size_t)y * AW + x];
        free(img);
    } else {
        float *img = (float *)malloc((P + extra) * sizeof(float));
        po_dwt97_inverse(coef, img, AW, AH, wl, qs);
        po_level_shift_inv_f32(img + extra, P, 8);
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++)
                frame_out[(size_t)y * W + x] = (uint8_t)img[extra + (size_t)y * AW + x];
        free(img);
    }
    free(staging); free(sizes); free(coef);
    return 0;
}
