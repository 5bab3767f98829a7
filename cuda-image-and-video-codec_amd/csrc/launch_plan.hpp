// launch_plan.hpp -- host-side launch geometry for the per-level DWT kernels (pure host code,
// shared by the C-ABI implementation and by the CPU wave-emulator tests so both walk the levels
// the same way).  Level order and scratch offsets follow DWTEngine::DWTForward / DWTReverse
// (reference DWT/DWTGenerator.cu:1268-1424).
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <cmath>
#include <vector>

#include "dwt_kernels.hpp"

namespace picsong {

// DWT/DWTGenerator.cuh:168-179 -- quantisation steps, columns LL,HL,LH,HH, row = level
static const float kQSteps[10][4] = {
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

struct FwdLaunch { DwtFwdArgs a; unsigned gx, gy; bool u8; int band; bool vec; };
struct InvLaunch { DwtInvArgs a; unsigned gx, gy; int band; bool vec; bool fast; };

// The vector-only kernel instantiations need whole 4-column groups and 16-byte aligned rows
// (PICSONG_DWT_NOVEC=1 forces the per-column kernels, used by the tests to cross-check both).
inline bool dwt_vec_ok(int W, int aw, const void *p0, const void *p1)
{
    if (const char *e = getenv("PICSONG_DWT_NOVEC")) if (atoi(e) != 0) return false;
    return W >= 4 && (W & 3) == 0 && (aw & 3) == 0 && (((uintptr_t)p0 | (uintptr_t)p1) & 15u) == 0;
}

// Band height per level: big levels want taller bands (less vertical halo re-read), small levels
// want many short waves (a level with a handful of tall waves is bound by one wave's serial
// instruction time, not by memory); PICSONG_DWT_BANDS="16,8,4,..." overrides per level.
inline int band_rows_override(int level)
{
    if (const char *e = getenv("PICSONG_DWT_BANDS")) {
        int l = 0;
        for (const char *p = e; *p; l++) {
            int v = atoi(p);
            if (l == level && (v == 4 || v == 8 || v == 16 || v == 32)) return v;
            while (*p && *p != ',') p++;
            if (*p == ',') p++;
        }
    }
    return 0;
}
inline int fwd_band_rows(int level, int strips, int H)
{
    if (const int v = band_rows_override(level)) return v;
    // measured on MI355X (8K: 16,8,4,4,4 best; 4K: 8,8,4,4,4): by level size in samples
    const long n = (long)H * (long)strips * kStripUseful;
    return n >= (16L << 20) ? 16 : (n >= (4L << 20) ? 8 : 4);
}
// The 9/7 synthesis of a context whose reciprocal divisions verified (dwt_inv97_kernel) is bound by its vector
// instructions, a third of which a 16-row band spends on run-in rows (12 iterations for 8 row pairs; 32 rows: 20
// for 16): taller bands than the other kernels want.  Measured on an 8K frame, wl = 6 (round 2):
// 32,16,8,4,4,4 rows for levels 0..5 81.1 us; 32,8,4,4,4,4 83.9; 16,8,4,4,4,4 95.2; 32-row bands on level 1
// double level 0's time (38.7 -> 76.5 us; same words out).
inline int inv97_band_rows(int level, int strips, int H)
{
    if (const int v = band_rows_override(level)) return v;
    const long n = (long)H * (long)strips * kStripUseful;
    return n >= (16L << 20) ? 32 : (n >= (4L << 20) ? 16 : (n >= (1L << 20) ? 8 : 4));
}

// -k > 0: bytes the COMPACT table copy of the frame's widest codeblock needs (bulk_setup<true>, bpc_kernels.hpp): the
// groups g0 .. g1 its 32 lanes' subbands span (findSubband BPC/BPCEngine.cu:143-170 at x = 64 cbx + 2 t, y = 64 cby), of
// the three sections, each with its slack planes.  A function of the geometry alone; the host takes the COMPACT
// instantiations of the -k > 0 kernels when it is at most kBulkCompactBytes.
inline int bulk_max_span_bytes(int aw, int ah, int wl, int nBp, int nSub, int cRef, int cSig, int cSign)
{
    auto group = [&](int x, int y) {
        for (int a = 1; a <= wl; a++) {
            const bool cx = x >= (aw >> a), cy = y >= (ah >> a);
            if (cx || cy) return (a - 1) * nSub + (cx ? (cy ? 2 : 0) : 1);
        }
        return wl * nSub;
    };
    int worst = 1;
    for (int cby = 0; cby < ah / 64; cby++)
        for (int cbx = 0; cbx < aw / 64; cbx++) {
            int g0 = 1 << 30, g1 = -1;
            for (int t = 0; t < 32; t++) { const int g = group(cbx * 64 + 2 * t, cby * 64); g0 = g < g0 ? g : g0; g1 = g > g1 ? g : g1; }
            worst = g1 - g0 + 1 > worst ? g1 - g0 + 1 : worst;
        }
    const int sl = nBp < 15 ? 16 - nBp : 1;
    return (worst * nBp + sl) * (cRef + cSig + cSign);
}

// May the frame paths carry their coded coefficients as 16-bit integers (DwtFwdArgs::c16)?  The largest magnitude a
// subband sample can take is  max|sample| x G_x x G_y  x (9/7: the quantisation weight QSTEP[l][sb] x qs), G = the L1
// gain of the multi-level 1-D analysis from the input to a level's low / high output: at most 1.3803 / 2.6253 for the
// 9/7 transform as implemented here, 1.7141 / 2.8601 for 5/3 whatever the level (tools/dwt_gain_bounds.py applies the
// transforms to the identity; the 5/3 floors move a sample by less than one per lifting step).  `in_max`: largest
// sample magnitude after the level shift / colour transform (128 for a grey 8-bit frame, 255 for RCT chroma).
// 2^15 with a margin for rounding; PICSONG_C16=0 keeps the 32-bit arrays (the tests cross-check both).
inline bool coef16_ok(bool lossy, int wl, float qs, int in_max)
{
    if (const char *e = getenv("PICSONG_C16")) if (atoi(e) == 0) return false;
    if (wl < 1 || wl > 10) return false;
    if (!lossy) return (double)in_max * 2.8601 * 2.8601 + 64.0 < 30000.0;
    if (!(qs > 0.0f)) return false;
    const double gl = 1.3803, gh = 2.6253, g[4] = { gl * gl, gl * gh, gl * gh, gh * gh };
    double worst = 0.0;
    for (int l = 0; l < wl; l++)
        for (int k = (l == wl - 1 ? 0 : 1); k < 4; k++) {
            const double b = (double)in_max * g[k] * (double)kQSteps[l][k] * (double)qs;
            worst = b > worst ? b : worst;
        }
    return worst < 30000.0;
}

inline std::vector<FwdLaunch> plan_dwt_forward(const void *d_in, bool u8in, void *d_out, int aw, int ah,
                                               int wl, float qs, bool c16 = false)
{
    std::vector<FwdLaunch> v;
    int W = aw, H = ah;
    size_t off = 0;
    const char *src = (const char *)d_in;
    int src_stride = aw;
    for (int l = 0; l < wl; l++) {
        const bool last = (l == wl - 1);
        off += (size_t)W * (size_t)H;
        FwdLaunch f;
        DwtFwdArgs &a = f.a;
        a.src = src; a.src_stride = src_stride; a.W = W; a.H = H;
        a.ll = last ? d_out : (void *)((char *)d_out + off * 4);
        a.ll_stride = last ? aw : (W >> 1);
        a.mallat = d_out; a.AW = aw; a.level = l; a.last = last ? 1 : 0; a.qs = qs;
        a.src_z = 0; a.dst_z = 0; a.pair_base = 0; a.pair_end = 0; a.c16 = c16 ? 1 : 0;
        a.src_g = a.src_b = nullptr;
        for (int k = 0; k < 4; k++) a.q[k] = kQSteps[l][k];
        const int strips = (W + kStripUseful - 1) / kStripUseful;
        f.band = fwd_band_rows(l, strips, H);
        f.gx = (unsigned)((strips + 3) / 4);
        f.gy = (unsigned)(((H >> 1) + f.band / 2 - 1) / (f.band / 2));
        f.u8 = u8in && l == 0;
        // rows of every buffer this level touches start 16-byte aligned when W % 4 == 0: strides are
        // aw, W or W/2 (even), scratch offsets are sums of W*H (multiples of 16 elements)
        f.vec = dwt_vec_ok(W, aw, d_in, d_out) && (W >> 1) % 2 == 0;
        v.push_back(f);
        src = (const char *)d_out + off * 4;
        src_stride = W >> 1;
        W >>= 1; H >>= 1;
    }
    // (the 16-bit form exists in the vector-only kernel instantiations: every level or none)
    bool all_vec = true;
    for (const FwdLaunch &f : v) all_vec = all_vec && f.vec;
    if (!all_vec) for (FwdLaunch &f : v) f.a.c16 = 0;
    return v;
}
inline bool plan_is_c16(const std::vector<FwdLaunch> &plan) { return !plan.empty() && plan[0].a.c16 != 0; }

// Level 0 restricted to the input rows [row0, row0 + rows) (both even): the launch then produces the row
// pairs [row0 / 2, (row0 + rows) / 2) of HL / LH / HH (Mallat) and of LL (scratch, or Mallat when wl = 1).
inline void plan_restrict_band(FwdLaunch &f, int row0, int rows)
{
    f.a.pair_base = row0 >> 1;
    f.a.pair_end = (row0 + rows) >> 1;
    f.gy = (unsigned)(((rows >> 1) + f.band / 2 - 1) / (f.band / 2));
}

// May the 9/7 synthesis of a context with this qs use the reciprocal form of its divisions
// (div_rc, dwt_kernels.hpp)?  Checks, with the very arithmetic the kernels run, every value the
// de-quantisation can see: m = |v| + 0.5 for |v| < 65536 (16 bit-planes) over each quantisation step
// q of the wl levels, m / q == div_rc(m, q) and (m / q) / qs == div_rc(m / q, qs).  ~3 M divisions,
// a few milliseconds, once per context.  PICSONG_DWT_EXACTDIV=1 forces the dividing kernels.
inline bool dequant_fast_ok(float qs, int wl)
{
    if (const char *e = getenv("PICSONG_DWT_EXACTDIV")) if (atoi(e) != 0) return false;
    if (!(qs >= 0x1p-20f && qs <= 0x1p20f)) return false;
    const volatile float one = 1.0f;                      // the reciprocals stay IEEE divisions at -O3 too
    const float rqs = one / qs;
    float seen[40];
    int nseen = 0;
    for (int l = 0; l < wl && l < 10; l++)
        for (int k = 0; k < 4; k++) {
            const float q = kQSteps[l][k];
            bool dup = false;
            for (int i = 0; i < nseen; i++) dup = dup || seen[i] == q;
            if (dup) continue;
            seen[nseen++] = q;
            const float rq = one / q;
            for (int n = 0; n < 65536; n++) {
                const float m = (float)n + 0.5f;
                const float t = m / q;
                if (div_rc(m, q, rq) != t || div_rc(t, qs, rqs) != t / qs) return false;
            }
        }
    return true;
}

// Levels 0 and 1 of a forward plan as ONE launch of dwt_fwd2_kernel (dwt_kernels.hpp): the frame path
// (u8 input: a band's 41-53 rows fit the registers as one dword each), both levels on the vector path,
// enough rows for the mirrored run-in.  `wanted`: the context's choice (not when it is told that other
// frames share the GPU, picsong_ctx_set_pipelined); PICSONG_DWT_NOFUSE01=1 / PICSONG_DWT_FUSE01=1 force
// two launches / the fused one (the tests cross-check both).
struct Fwd2Launch { DwtFwd2Args a; unsigned gx, gy; };
// nb_override > 0: level-1 row pairs per band (the RGB head: kF2PairsRgb)
inline bool plan_dwt_fwd2(const std::vector<FwdLaunch> &plan, Fwd2Launch &f, bool wanted = true, bool lossy = false,
                          int nb_override = 0)
{
    if (const char *e = getenv("PICSONG_DWT_NOFUSE01")) if (atoi(e) != 0) return false;
    if (const char *e = getenv("PICSONG_DWT_FUSE01")) wanted = wanted || atoi(e) != 0;
    if (!wanted) return false;
    if (plan.size() < 2 || !plan[0].vec || !plan[1].vec || !plan[0].u8) return false;
    const DwtFwdArgs &l0 = plan[0].a;
    if (l0.H < 64 || (l0.H & 3) || l0.W < 8 || (l0.W & 7)) return false;
    const int pairs1 = l0.H >> 2;                                   // level-1 row pairs
    f.a.l0 = l0; f.a.l1 = plan[1].a;
    const int strips = (l0.W + kF2Useful - 1) / kF2Useful;
    f.gx = (unsigned)((strips + 3) / 4);
    const int nb = nb_override > 0 ? nb_override : (lossy ? kF2PairsLossy : kF2Pairs);
    if (pairs1 % nb) return false;                                  // whole bands only (dwt_fwd2_band)
    f.gy = (unsigned)((pairs1 + nb - 1) / nb);
    return true;
}

// the 16-bit form exists in the vector-only kernel instantiations: every level of the frame on the vector path
inline bool dwt_c16_geometry_ok(int aw, int ah, int wl)
{   // every level on the vector path: level widths multiples of 4 (and rows 16-byte aligned: aw % 4 == 0)
    if (const char *e = getenv("PICSONG_DWT_NOVEC")) if (atoi(e) != 0) return false;
    for (int l = 0; l < wl; l++) {
        const int W = aw >> l;
        if (W < 4 || (W & 3) || ((W >> 1) & 1)) return false;
    }
    return (aw & 3) == 0 && (ah >> (wl - 1)) >= 2;
}
// c16: the coded coefficients at d_in are an int16 Mallat array (the decode frame paths, DwtInvArgs::c16): the vector
// kernels' C16 instantiations on every level, or not at all (the caller looks at plan_inv_is_c16)
inline std::vector<InvLaunch> plan_dwt_inverse(const int32_t *d_in, void *d_out, int aw, int ah, int wl,
                                               float qs, bool fast = false, bool c16 = false)
{
    std::vector<InvLaunch> v;
    int W = aw >> (wl - 1), H = ah >> (wl - 1);
    size_t read_off = 0, write_off = 0;
    for (int l = wl - 1; l >= 0; l--) {
        const bool first = (l == wl - 1);
        InvLaunch f;
        DwtInvArgs &a = f.a;
        a.mallat = d_in; a.AW = aw;
        a.ll = first ? (const void *)d_in : (const void *)((const char *)d_out + read_off * 4);
        a.ll_stride = first ? aw : (W >> 1);
        a.first = first ? 1 : 0;
        a.W = W; a.H = H;
        a.dst = (char *)d_out + write_off * 4;
        a.dst_u8 = nullptr; a.off = 0;
        a.mallat_z = a.ll_z = a.dst_z = a.u8_z = 0;
        {   // qs = 2^k: dividing by it is exact scaling, dwt_inv97_kernel folds it into the step
            int e = 0;
            a.one_div = std::frexp(qs, &e) == 0.5f ? 1 : 0;
            const char *x = getenv("PICSONG_DWT_EXACT_REPLAY");
            a.exact_replay = x && atoi(x) != 0 ? 1 : 0;
        }
        a.qs = qs;
        a.rqs = 1.0f / qs;
        for (int k = 0; k < 4; k++) { a.q[k] = kQSteps[l][k]; a.rq[k] = 1.0f / a.q[k]; }
        f.fast = fast;
        const int strips = (W + kStripUseful - 1) / kStripUseful;
        f.band = fast ? inv97_band_rows(l, strips, H) : fwd_band_rows(l, strips, H);
        f.gx = (unsigned)((strips + 3) / 4);
        f.gy = (unsigned)(((H >> 1) + f.band / 2 - 1) / (f.band / 2));
        f.vec = dwt_vec_ok(W, aw, d_in, d_out);
        a.c16 = c16 ? 1 : 0;
        v.push_back(f);
        read_off = write_off;
        write_off += (size_t)W * (size_t)H;
        W <<= 1; H <<= 1;
    }
    bool all_vec = true;
    for (const InvLaunch &f : v) all_vec = all_vec && f.vec;
    if (!all_vec) for (InvLaunch &f : v) f.a.c16 = 0;
    return v;
}
inline bool plan_inv_is_c16(const std::vector<InvLaunch> &plan) { return !plan.empty() && plan[0].a.c16 != 0; }

// May a decode context take the 16-bit coefficient form between its decoder and its synthesis?  The bound of coef16_ok
// (an honest stream of such a context has no magnitude of 2^15 or more), the vector kernels on every level, at least
// two levels (the coarsest level that also writes pixels has no C16 instantiation), and for 9/7 the verified
// reciprocal divisions (the lean kernels are the ones with a C16 form).  PICSONG_DEC_C16=0 keeps the 32-bit arrays.
inline bool dec_c16_ok(bool lossy, int wl, float qs, int in_max, int aw, int ah, bool fast_div)
{
    if (const char *e = getenv("PICSONG_DEC_C16")) if (atoi(e) == 0) return false;
    if (const char *e = getenv("PICSONG_DWT_INV97")) if (atoi(e) == 0) return false;
    return wl >= 2 && (!lossy || fast_div) && coef16_ok(lossy, wl, qs, in_max) && dwt_c16_geometry_ok(aw, ah, wl);
}

// Synthesis levels 1 and 0 of an inverse plan as ONE launch of dwt_inv2_kernel (dwt_kernels.hpp): the decode frame
// paths (16-bit coefficients in, pixels out), level 1 not the coarsest, whole 32-row bands.  PICSONG_DWT_NOFUSE_INV=1
// keeps the two launches (the tests cross-check both).
struct Inv2Launch { DwtInv2Args a; unsigned gx, gy; };
inline bool plan_dwt_inv2(const std::vector<InvLaunch> &plan, Inv2Launch &f, bool lossy)
{
    if (const char *e = getenv("PICSONG_DWT_NOFUSE_INV")) if (atoi(e) != 0) return false;
    if (plan.size() < 3) return false;
    const InvLaunch &p1 = plan[plan.size() - 2], &p0 = plan.back();
    if (!p0.vec || !p1.vec || !p0.a.c16 || !p0.a.dst_u8 || p1.a.first) return false;
    // 9/7: built, bit-identical and NOT the default -- the 9/7 synthesis is bound by its arithmetic (three-instruction
    // divisions the reference's rounding demands), not by traffic, and the fused form pays for the LL0 bytes it saves
    // with level 1's longer run-in (14 steps for 8 row pairs) and 128 registers (4 waves a SIMD): an 8K frame's
    // levels 1 + 0 take 48.0 us fused against 15.3 + 35.1 us, decode with three calls in flight 150 against 157
    // Gpixel/s (round 4, profiles/NOTES.md).  PICSONG_DWT_FUSE_INV97=1 selects it (the tests run both).
    if (lossy) {
        const char *e = getenv("PICSONG_DWT_FUSE_INV97");
        if (!(e && atoi(e) != 0) || !p0.fast) return false;
    }
    if (p0.a.W < 16 || (p0.a.W & 7) || p0.a.H < 64 || (p0.a.H % (4 * kI2Pairs))) return false;
    f.a.l1 = p1.a; f.a.l0 = p0.a;
    const int useful = lossy ? i2_useful<true>() : i2_useful<false>();
    const int strips = (p0.a.W + useful - 1) / useful;
    f.gx = (unsigned)((strips + 3) / 4);
    f.gy = (unsigned)(p0.a.H / (4 * kI2Pairs));
    return true;
}

}  // namespace picsong
