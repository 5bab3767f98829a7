// TEST INFRASTRUCTURE ONLY -- runs the product's kernel sources on the CPU wave emulator and
// exposes host-memory entry points (ctypes) so tests can compare them with the oracle without a
// GPU.  Mirrors the launch sequence of cuda-image-and-video-codec_amd/csrc/picsong_hip.hip.
#include <hip/hip_runtime.h>

#include "../../cuda-image-and-video-codec_amd/csrc/bpc_kernels.hpp"
#include "../../cuda-image-and-video-codec_amd/csrc/dwt_kernels.hpp"
#include "../../cuda-image-and-video-codec_amd/csrc/launch_plan.hpp"
#include "../../cuda-image-and-video-codec_amd/csrc/pack_kernels.hpp"

using namespace picsong;

// what picsong_ctx_create decides once per context (cached here per (qs, wl))
static bool emu_fast_div(int lossy, float qs, int wl)
{
    static float c_qs = -1.0f;
    static int c_wl = -1;
    static bool c_ok = false;
    if (!lossy) return false;
    if (getenv("PICSONG_DWT_EXACTDIV") || qs != c_qs || wl != c_wl) { c_ok = dequant_fast_ok(qs, wl); c_qs = qs; c_wl = wl; }
    return c_ok;
}
extern "C" int emu_dequant_fast_ok(float qs, int wl) { return dequant_fast_ok(qs, wl) ? 1 : 0; }
// x / c next to the reciprocal form, for the test that sweeps them against each other
extern "C" long emu_div_mismatches(float c, unsigned first_bits, unsigned last_bits, unsigned step)
{
    const volatile float one = 1.0f;
    const float rc = one / c;
    long bad = 0;
    for (uint64_t u = first_bits; u <= last_bits; u += step) {
        float x;
        const uint32_t b = (uint32_t)u;
        memcpy(&x, &b, 4);
        if (div_rc(x, c, rc) != x / c) bad++;
    }
    return bad;
}

template <int BAND, bool VEC> static void emu_inv_v(const InvLaunch &f, int lossy)
{
    DwtInvArgs a = f.a;
    if (lossy && f.fast) emu::launch(dim3(f.gx, f.gy), dim3(256), [&] { dwt_inv_kernel<float, true, BAND, VEC, false, true>(a); });
    else if (lossy) emu::launch(dim3(f.gx, f.gy), dim3(256), [&] { dwt_inv_kernel<float, true, BAND, VEC>(a); });
    else emu::launch(dim3(f.gx, f.gy), dim3(256), [&] { dwt_inv_kernel<int, false, BAND, VEC>(a); });
}
// mirrors launch_inv (picsong_hip.hip): the lean 9/7 kernel for the vector launches of a verified context
static bool emu_lean97() { const char *e = getenv("PICSONG_DWT_INV97"); return !(e && atoi(e) == 0); }
template <int BAND> static void emu_inv(const InvLaunch &f, int lossy)
{
    DwtInvArgs a = f.a;
    if (a.c16) {                        // the decode frame paths' 16-bit coefficients (mirrors launch_inv)
        const dim3 grid(f.gx, f.gy);
        if (lossy) {
            if (a.dst_u8) {
                if (a.one_div) emu::launch(grid, dim3(256), [&] { dwt_inv97_kernel<BAND, true, false, true, true>(a); });
                else emu::launch(grid, dim3(256), [&] { dwt_inv97_kernel<BAND, true, false, false, true>(a); });
            } else if (a.first) {
                if (a.one_div) emu::launch(grid, dim3(256), [&] { dwt_inv97_kernel<BAND, false, true, true, true>(a); });
                else emu::launch(grid, dim3(256), [&] { dwt_inv97_kernel<BAND, false, true, false, true>(a); });
            } else {
                if (a.one_div) emu::launch(grid, dim3(256), [&] { dwt_inv97_kernel<BAND, false, false, true, true>(a); });
                else emu::launch(grid, dim3(256), [&] { dwt_inv97_kernel<BAND, false, false, false, true>(a); });
            }
        } else if (a.dst_u8) emu::launch(grid, dim3(256), [&] { dwt_inv_kernel<int, false, BAND, true, true, false, true>(a); });
        else emu::launch(grid, dim3(256), [&] { dwt_inv_kernel<int, false, BAND, true, false, false, true>(a); });
        return;
    }
    if (f.vec && lossy && f.fast && emu_lean97() && !(a.first && a.dst_u8)) {
        const dim3 grid(f.gx, f.gy);
        if (a.dst_u8) {
            if (a.one_div) emu::launch(grid, dim3(256), [&] { dwt_inv97_kernel<BAND, true, false, true>(a); });
            else emu::launch(grid, dim3(256), [&] { dwt_inv97_kernel<BAND, true, false, false>(a); });
        } else if (a.first) {
            if (a.one_div) emu::launch(grid, dim3(256), [&] { dwt_inv97_kernel<BAND, false, true, true>(a); });
            else emu::launch(grid, dim3(256), [&] { dwt_inv97_kernel<BAND, false, true, false>(a); });
        } else {
            if (a.one_div) emu::launch(grid, dim3(256), [&] { dwt_inv97_kernel<BAND, false, false, true>(a); });
            else emu::launch(grid, dim3(256), [&] { dwt_inv97_kernel<BAND, false, false, false>(a); });
        }
    } else if (f.vec && a.dst_u8) {     // finest level of the frame path: pixels out, clamp fused
        if (lossy && f.fast) emu::launch(dim3(f.gx, f.gy), dim3(256), [&] { dwt_inv_kernel<float, true, BAND, true, true, true>(a); });
        else if (lossy) emu::launch(dim3(f.gx, f.gy), dim3(256), [&] { dwt_inv_kernel<float, true, BAND, true, true>(a); });
        else emu::launch(dim3(f.gx, f.gy), dim3(256), [&] { dwt_inv_kernel<int, false, BAND, true, true>(a); });
    } else if (f.vec) emu_inv_v<BAND, true>(f, lossy);
    else emu_inv_v<BAND, false>(f, lossy);
}

template <int BAND, bool VEC> static void emu_fwd_v(const FwdLaunch &f, int lossy)
{
    DwtFwdArgs a = f.a;
    if (lossy) {
        if (f.u8) emu::launch(dim3(f.gx, f.gy), dim3(256), [&] { dwt_fwd_kernel<float, true, true, BAND, VEC>(a); });
        else emu::launch(dim3(f.gx, f.gy), dim3(256), [&] { dwt_fwd_kernel<float, true, false, BAND, VEC>(a); });
    } else {
        if (f.u8) emu::launch(dim3(f.gx, f.gy), dim3(256), [&] { dwt_fwd_kernel<int, false, true, BAND, VEC>(a); });
        else emu::launch(dim3(f.gx, f.gy), dim3(256), [&] { dwt_fwd_kernel<int, false, false, BAND, VEC>(a); });
    }
}
template <int BAND> static void emu_fwd(const FwdLaunch &f, int lossy)
{
    if (f.vec) emu_fwd_v<BAND, true>(f, lossy); else emu_fwd_v<BAND, false>(f, lossy);
}

// frame paths: coded coefficients as int16 (DwtFwdArgs::c16 / BpcArgs::c16), switched on by the tests
static int g_c16 = 0;
extern "C" void emu_set_c16(int on) { g_c16 = on; }
extern "C" int emu_coef16_ok(int lossy, int wl, float qs, int in_max, int aw, int ah)
{
    return coef16_ok(lossy != 0, wl, qs, in_max) && dwt_c16_geometry_ok(aw, ah, wl) ? 1 : 0;
}

static void emu_fwd2(const Fwd2Launch &f, int lossy)
{
    DwtFwd2Args a = f.a;
    const dim3 grid(f.gx, f.gy);
    if (a.l0.c16) {
        if (lossy) emu::launch(grid, dim3(256), [&] { dwt_fwd2_kernel<float, true, true, kF2PairsLossy, true>(a); });
        else emu::launch(grid, dim3(256), [&] { dwt_fwd2_kernel<int, false, true, kF2Pairs, true>(a); });
        return;
    }
    if (lossy) emu::launch(grid, dim3(256), [&] { dwt_fwd2_kernel<float, true, true, kF2PairsLossy>(a); });
    else emu::launch(grid, dim3(256), [&] { dwt_fwd2_kernel<int, false, true, kF2Pairs>(a); });
}

extern "C" {

// how many levels of the plans take the vector-only kernel instantiations (tests assert on it)
int emu_dwt_vec_levels(const void *in, void *out, int aw, int ah, int wl)
{
    int n = 0;
    for (const FwdLaunch &f : plan_dwt_forward(in, true, out, aw, ah, wl, 1.0f)) n += f.vec ? 1 : 0;
    for (const InvLaunch &f : plan_dwt_inverse((const int32_t *)in, out, aw, ah, wl, 1.0f)) n += f.vec ? 1 : 0;
    return n;
}

static void emu_fwd_any(const FwdLaunch &f, int lossy)
{
    switch (f.band) {
    case 32: emu_fwd<32>(f, lossy); break;
    case 16: emu_fwd<16>(f, lossy); break;
    case 8: emu_fwd<8>(f, lossy); break;
    default: emu_fwd<4>(f, lossy); break;
    }
}

// mirrors launch_fwd_levels (picsong_hip.hip)
static void emu_fwd_levels(const std::vector<FwdLaunch> &plan, size_t from, int lossy)
{
    for (size_t l = from; l < plan.size(); l++) emu_fwd_any(plan[l], lossy);
}

// mirrors dwt_forward_impl (picsong_hip.hip); returns 1 when levels 0 and 1 went through the fused kernel
int emu_dwt_forward(const void *in, int u8in, void *out, int aw, int ah, int wl, int lossy, float qs)
{
    const std::vector<FwdLaunch> plan = plan_dwt_forward(in, u8in != 0, out, aw, ah, wl, qs, g_c16 != 0);
    Fwd2Launch f2;
    const bool fused01 = plan_dwt_fwd2(plan, f2, true, lossy != 0);
    if (fused01) {
        emu_fwd2(f2, lossy);
    }
    emu_fwd_levels(plan, fused01 ? 2 : 0, lossy);
    return fused01 ? 1 : 0;
}

// mirrors picsong_encode_rgb_frame's transform: the colour transform (RCT / ICT) in the fused head's load stage, one launch for the
// three components (out: three coefficient buffers of `stride` bytes, int16 Mallat arrays at their starts); returns 1
// when the fused form applies
int emu_dwt_forward_rgb(const uint8_t *r, const uint8_t *g, const uint8_t *b, void *out, size_t stride, int aw, int ah, int wl,
                        int lossy, float qs)
{
    std::vector<FwdLaunch> plan = plan_dwt_forward(r, true, out, aw, ah, wl, qs, true);
    Fwd2Launch f2;
    if (!plan_is_c16(plan) || !plan_dwt_fwd2(plan, f2, true, lossy != 0, kF2PairsRgb)) return 0;
    for (size_t l = 0; l < plan.size(); l++) {
        plan[l].a.src_z = l == 0 ? 0ull : (unsigned long long)stride;
        plan[l].a.dst_z = (unsigned long long)stride;
    }
    plan[0].a.src_g = g; plan[0].a.src_b = b;
    f2.a.l0 = plan[0].a; f2.a.l1 = plan[1].a;
    DwtFwd2Args a2 = f2.a;
    if (lossy) emu::launch(dim3(f2.gx, f2.gy, 3), dim3(256), [&] { dwt_fwd2_kernel<float, true, true, kF2PairsRgb, true, true>(a2); });
    else emu::launch(dim3(f2.gx, f2.gy, 3), dim3(256), [&] { dwt_fwd2_kernel<int, false, true, kF2PairsRgb, true, true>(a2); });
    for (size_t l = 2; l < plan.size(); l++) {
        const FwdLaunch &f = plan[l];
        DwtFwdArgs a = f.a;
        const dim3 grid(f.gx, f.gy, 3);
        if (lossy) {
            switch (f.band) {
            case 32: emu::launch(grid, dim3(256), [&] { dwt_fwd_kernel<float, true, false, 32, true>(a); }); break;
            case 16: emu::launch(grid, dim3(256), [&] { dwt_fwd_kernel<float, true, false, 16, true>(a); }); break;
            case 8: emu::launch(grid, dim3(256), [&] { dwt_fwd_kernel<float, true, false, 8, true>(a); }); break;
            default: emu::launch(grid, dim3(256), [&] { dwt_fwd_kernel<float, true, false, 4, true>(a); }); break;
            }
        } else {
            switch (f.band) {
            case 32: emu::launch(grid, dim3(256), [&] { dwt_fwd_kernel<int, false, false, 32, true>(a); }); break;
            case 16: emu::launch(grid, dim3(256), [&] { dwt_fwd_kernel<int, false, false, 16, true>(a); }); break;
            case 8: emu::launch(grid, dim3(256), [&] { dwt_fwd_kernel<int, false, false, 8, true>(a); }); break;
            default: emu::launch(grid, dim3(256), [&] { dwt_fwd_kernel<int, false, false, 4, true>(a); }); break;
            }
        }
    }
    return 1;
}

// mirror picsong_dwt_forward_band / picsong_dwt_forward_tail (picsong_hip.hip)
void emu_dwt_forward_band(const void *in, void *out, int aw, int ah, int wl, int lossy, float qs, int row0, int rows)
{
    std::vector<FwdLaunch> plan = plan_dwt_forward(in, true, out, aw, ah, wl, qs);
    plan_restrict_band(plan[0], row0, rows);
    emu_fwd_any(plan[0], lossy);
}

void emu_dwt_forward_tail(void *out, int aw, int ah, int wl, int lossy, float qs)
{
    const std::vector<FwdLaunch> plan = plan_dwt_forward(out, false, out, aw, ah, wl, qs);
    emu_fwd_levels(plan, 1, lossy);
}

void emu_dwt_inverse(const int32_t *in, void *out, int aw, int ah, int wl, int lossy, float qs)
{
    for (const InvLaunch &f : plan_dwt_inverse(in, out, aw, ah, wl, qs, emu_fast_div(lossy, qs, wl))) {
        switch (f.band) {
        case 32: emu_inv<32>(f, lossy); break;
        case 16: emu_inv<16>(f, lossy); break;
        case 8: emu_inv<8>(f, lossy); break;
        default: emu_inv<4>(f, lossy); break;
        }
    }
}

// the frame path's inverse: the finest level writes clamped pixels (mirrors dwt_inverse_impl);
// returns 1 when that fused kernel applied
int emu_dwt_inverse_u8(const int32_t *in, void *scratch, uint8_t *pixels, int aw, int ah, int wl, int lossy, float qs)
{
    std::vector<InvLaunch> plan = plan_dwt_inverse(in, scratch, aw, ah, wl, qs, emu_fast_div(lossy, qs, wl));
    const bool fused = !plan.empty() && plan.back().vec && (((uintptr_t)pixels) & 3u) == 0;
    if (fused) { plan.back().a.dst_u8 = pixels; plan.back().a.off = 128; }
    for (const InvLaunch &f : plan) {
        switch (f.band) {
        case 32: emu_inv<32>(f, lossy); break;
        case 16: emu_inv<16>(f, lossy); break;
        case 8: emu_inv<8>(f, lossy); break;
        default: emu_inv<4>(f, lossy); break;
        }
    }
    return fused ? 1 : 0;
}

// the decode frame paths with 16-bit coefficients (mirrors inverse_plan + run_inverse, picsong_hip.hip): `in16` is an
// int16 Mallat array; returns bit 0: the finest level wrote the pixels, bit 1: levels 1 and 0 ran as ONE launch
// (dwt_inv2_kernel), bit 2: the plan took the 16-bit form (0: the caller's geometry / context does not allow it)
int emu_dwt_inverse_u8_c16(const int16_t *in16, void *scratch, uint8_t *pixels, int aw, int ah, int wl, int lossy, float qs)
{
    const bool fast = emu_fast_div(lossy, qs, wl);
    if (!dec_c16_ok(lossy != 0, wl, qs, 128, aw, ah, fast)) return 0;
    std::vector<InvLaunch> plan = plan_dwt_inverse((const int32_t *)in16, scratch, aw, ah, wl, qs, fast, true);
    if (!plan_inv_is_c16(plan)) return 0;
    int res = 4;
    if (plan.back().vec && (((uintptr_t)pixels) & 3u) == 0) { plan.back().a.dst_u8 = pixels; plan.back().a.off = 128; res |= 1; }
    Inv2Launch f2;
    const bool fused10 = plan_dwt_inv2(plan, f2, lossy != 0);
    const size_t n = fused10 ? plan.size() - 2 : plan.size();
    for (size_t l = 0; l < n; l++) {
        const InvLaunch &f = plan[l];
        switch (f.band) {
        case 32: emu_inv<32>(f, lossy); break;
        case 16: emu_inv<16>(f, lossy); break;
        case 8: emu_inv<8>(f, lossy); break;
        default: emu_inv<4>(f, lossy); break;
        }
    }
    if (fused10) {
        DwtInv2Args a2 = f2.a;
        const dim3 grid(f2.gx, f2.gy);
        if (!lossy) emu::launch(grid, dim3(256), [&] { dwt_inv2_kernel<false, false>(a2); });
        else if (a2.l0.one_div) emu::launch(grid, dim3(256), [&] { dwt_inv2_kernel<true, true>(a2); });
        else emu::launch(grid, dim3(256), [&] { dwt_inv2_kernel<true, false>(a2); });
        res |= 2;
    }
    return res;
}

// picsong_decode_rgb_frame's synthesis with 16-bit coefficients (5/3: dwt_inv_rgb_kernel; 9/7: dwt_inv97_rgb_kernel): the levels above the finest per component,
// then the finest level of all three components + the inverse colour transform as ONE launch (dwt_inv_rgb_kernel).
// in16: three int16 Mallat arrays in_z BYTES apart; scratch: three work buffers wrk_z bytes apart.  Returns 1 when the
// form applies.
int emu_dwt_inverse_rgb(const int16_t *in16, size_t in_z, void *scratch, size_t wrk_z, uint8_t *r, uint8_t *g, uint8_t *b,
                        int aw, int ah, int wl, int lossy, float qs)
{
    const bool fast = emu_fast_div(lossy, qs, wl);
    if (!dec_c16_ok(lossy != 0, wl, qs, 255, aw, ah, fast)) return 0;
    std::vector<InvLaunch> plan0;
    for (int c = 0; c < 3; c++) {
        std::vector<InvLaunch> plan = plan_dwt_inverse((const int32_t *)((const char *)in16 + c * in_z), (char *)scratch + c * wrk_z,
                                                       aw, ah, wl, qs, fast, true);
        if (!plan_inv_is_c16(plan) || plan.size() < 2 || !plan.back().vec) return 0;
        for (size_t l = 0; l + 1 < plan.size(); l++) {
            const InvLaunch &f = plan[l];
            switch (f.band) {
            case 32: emu_inv<32>(f, lossy); break;
            case 16: emu_inv<16>(f, lossy); break;
            case 8: emu_inv<8>(f, lossy); break;
            default: emu_inv<4>(f, lossy); break;
            }
        }
        if (c == 0) plan0 = plan;
    }
    const InvLaunch &f = plan0.back();
    DwtInvArgs fa = f.a;
    fa.mallat_z = in_z; fa.ll_z = wrk_z; fa.off = 128;
    if (lossy) {
        // 9/7: the three components as the three waves of a workgroup (dwt_inv97_rgb_kernel)
        const dim3 grid((unsigned)((fa.W + kStripUseful - 1) / kStripUseful), f.gy);
#define EMU_INV97_RGB(B)                                                                                          \
        do { if (fa.one_div) emu::launch(grid, dim3(192), [&] { dwt_inv97_rgb_kernel<B, true>(fa, r, g, b); });    \
             else emu::launch(grid, dim3(192), [&] { dwt_inv97_rgb_kernel<B, false>(fa, r, g, b); }); } while (0)
        switch (f.band) {
        case 32: EMU_INV97_RGB(32); break;
        case 16: EMU_INV97_RGB(16); break;
        case 8: EMU_INV97_RGB(8); break;
        default: EMU_INV97_RGB(4); break;
        }
#undef EMU_INV97_RGB
        return 1;
    }
    const dim3 grid(f.gx, f.gy);
    switch (f.band) {
    case 32: emu::launch(grid, dim3(256), [&] { dwt_inv_rgb_kernel<32>(fa, r, g, b); }); break;
    case 16: emu::launch(grid, dim3(256), [&] { dwt_inv_rgb_kernel<16>(fa, r, g, b); }); break;
    case 8: emu::launch(grid, dim3(256), [&] { dwt_inv_rgb_kernel<8>(fa, r, g, b); }); break;
    default: emu::launch(grid, dim3(256), [&] { dwt_inv_rgb_kernel<4>(fa, r, g, b); }); break;
    }
    return 1;
}

void emu_level_shift_inv(void *data, size_t n, int lossy)
{
    if (lossy) emu::launch(dim3(4), dim3(256), [&] { level_shift_inv_f32_kernel((float *)data, n, 128.0f); });
    else emu::launch(dim3(4), dim3(256), [&] { level_shift_inv_i32_kernel((int32_t *)data, n, 128); });
}

void emu_clamp_to_u8(const void *data, uint8_t *out, size_t n, int lossy)
{
    if (lossy) emu::launch(dim3(4), dim3(256), [&] { clamp_to_u8_f32_kernel((const float *)data, out, n / 4, 128.0f); });
    else emu::launch(dim3(4), dim3(256), [&] { clamp_to_u8_i32_kernel((const int32_t *)data, out, n / 4, 128); });
}

void emu_rgb_forward(const uint8_t *r, const uint8_t *g, const uint8_t *b, void *c0, void *c1, void *c2, size_t n, int lossy)
{
    if (lossy) emu::launch(dim3(3), dim3(256), [&] { rgb_forward_kernel<float>(r, g, b, (float *)c0, (float *)c1, (float *)c2, n / 4, 128); });
    else emu::launch(dim3(3), dim3(256), [&] { rgb_forward_kernel<int32_t>(r, g, b, (int32_t *)c0, (int32_t *)c1, (int32_t *)c2, n / 4, 128); });
}

void emu_rgb_inverse(const void *c0, const void *c1, const void *c2, uint8_t *r, uint8_t *g, uint8_t *b, size_t n, int lossy)
{
    if (lossy) emu::launch(dim3(3), dim3(256), [&] { rgb_inverse_kernel<float>((const float *)c0, (const float *)c1, (const float *)c2, r, g, b, n / 4, 128); });
    else emu::launch(dim3(3), dim3(256), [&] { rgb_inverse_kernel<int32_t>((const int32_t *)c0, (const int32_t *)c1, (const int32_t *)c2, r, g, b, n / 4, 128); });
}

void emu_level_shift_fwd(const uint8_t *in, void *out, size_t n, int lossy)
{
    if (lossy) emu::launch(dim3(4), dim3(256), [&] { level_shift_fwd_kernel<float>(in, (float *)out, n / 4, 128); });
    else emu::launch(dim3(4), dim3(256), [&] { level_shift_fwd_kernel<int32_t>(in, (int32_t *)out, n / 4, 128); });
}

// mirrors bulk_compact (picsong_hip.hip)
static bool emu_bulk_compact(int aw, int ah, int wl, const int *geo)
{
    if (const char *e = getenv("PICSONG_BULK_FULLTAB")) if (atoi(e) != 0) return false;
    return bulk_max_span_bytes(aw, ah, wl, geo[0], geo[1], geo[2], geo[4], geo[3]) <= kBulkCompactBytes;
}

static BpcArgs mk(int aw, int ah, int wl, const int32_t *lut, const int *geo, int32_t *staging, int32_t *sizes,
                  int *flag)
{
    BpcArgs a;
    memset(&a, 0, sizeof a);
    a.AW = aw; a.AH = ah; a.wl = wl; a.ncx = aw / 64; a.nCB = (aw / 64) * (ah / 64);
    a.lut = lut;
    a.g.nBp = geo[0]; a.g.nSub = geo[1]; a.g.cRef = geo[2]; a.g.cSign = geo[3]; a.g.cSig = geo[4];
    a.g.prec = geo[5]; a.g.nRef = geo[6]; a.g.nSig = geo[7]; a.g.nSign = geo[8];
    a.staging = staging; a.sizes = sizes; a.range_flag = flag;
    a.c16 = g_c16;
    return a;
}

// k > 0 (n_tables bit-plane tables in lut) runs the BULK instantiations, like picsong_hip.hip
// codeblocks [cb_begin, cb_begin + cb_count) of the frame (cb_count < 0: all), like bpc_encode_impl
void emu_bpc_encode_range(const void *coeffs, int is_float, int aw, int ah, int wl, const int32_t *lut, const int *geo,
                          int32_t *staging, int32_t *sizes, int *flag, int cb_begin, int cb_count)
{
    BpcArgs a = mk(aw, ah, wl, lut, geo, nullptr, sizes, flag);
    a.coeffs_in = coeffs; a.is_float = is_float;
    a.k = 0.0f; a.n_tables = 1;
    if (cb_count < 0) cb_count = a.nCB - cb_begin;
    std::vector<uint16_t> st16((size_t)a.nCB * 4096, 0xDEADu);        // the encoders' 16-bit staging, poisoned
    a.staging16 = st16.data();
    a.cb_base = cb_begin; a.nCB = cb_begin + cb_count;
    const unsigned wgs = (unsigned)(((cb_count + 1) / 2 + kBpcEncWgWaves - 1) / kBpcEncWgWaves);
    std::vector<uint32_t> plane_scratch((size_t)wgs * kBpcEncWgWaves * kEncScratchDwordsPerWave, 0xDEADBEEFu);
    a.plane_scratch = plane_scratch.data();
    emu::launch(dim3(wgs), dim3(64 * kBpcEncWgWaves), [&] { bpc_encode_kernel<false>(a); });
    // (as picsong_bpc_encode: widened into the caller's int32 array, words 0 .. len - 1 of the range's codeblocks)
    emu::launch(dim3((unsigned)cb_count), dim3(256), [&] { widen_staging_kernel(st16.data(), sizes, cb_begin, staging); });
}

void emu_bpc_encode(const void *coeffs, int is_float, int aw, int ah, int wl, const int32_t *lut, const int *geo,
                    int32_t *staging, int32_t *sizes, int *flag, float k, int n_tables)
{
    BpcArgs a = mk(aw, ah, wl, lut, geo, nullptr, sizes, flag);
    a.coeffs_in = coeffs; a.is_float = is_float;
    a.k = k; a.n_tables = n_tables;
    std::vector<uint16_t> st16((size_t)a.nCB * 4096, 0xDEADu);
    a.staging16 = st16.data();
    std::vector<uint32_t> plane_scratch((size_t)(((a.nCB + 1) / 2 + kBpcEncWgWaves - 1) / kBpcEncWgWaves * kBpcEncWgWaves) * kEncScratchDwordsPerWave, 0xDEADBEEFu);
    a.plane_scratch = plane_scratch.data();
    memset(staging, 0xFF, (size_t)aw * ah * 4);
    if (k > 0.0f && emu_bulk_compact(aw, ah, wl, geo)) emu::launch(dim3((unsigned)((a.nCB + 1) / 2)), dim3(64), [&] { bpc_encode_kernel<true, true>(a); });
    else if (k > 0.0f) emu::launch(dim3((unsigned)((a.nCB + 1) / 2)), dim3(64), [&] { bpc_encode_kernel<true>(a); });
    else emu::launch(dim3((unsigned)(((a.nCB + 1) / 2 + kBpcEncWgWaves - 1) / kBpcEncWgWaves)), dim3(64 * kBpcEncWgWaves), [&] { bpc_encode_kernel<false>(a); });
    emu::launch(dim3((unsigned)a.nCB), dim3(256), [&] { widen_staging_kernel(st16.data(), sizes, 0, staging); });
}

void emu_bpc_decode(const int32_t *staging, const int32_t *sizes, int aw, int ah, int wl, const int32_t *lut,
                    const int *geo, int32_t *coeffs, int *flag, float k, int n_tables)
{
    BpcArgs a = mk(aw, ah, wl, lut, geo, const_cast<int32_t *>(staging), const_cast<int32_t *>(sizes), flag);
    a.coeffs_out = coeffs;
    a.k = k; a.n_tables = n_tables;
    const dim3 grid((unsigned)((a.nCB + 1) / 2));
    if (k > 0.0f) {
        std::vector<uint32_t> plane_scratch((size_t)grid.x * kEncScratchDwordsPerWave, 0xDEADBEEFu);
        a.plane_scratch = plane_scratch.data();
        if (emu_bulk_compact(aw, ah, wl, geo)) emu::launch(grid, dim3(64), [&] { bpc_decode_kernel<true, kDecSmallPlanes, false, false, true>(a); });
        else emu::launch(grid, dim3(64), [&] { bpc_decode_kernel<true, kDecSmallPlanes>(a); });
    } else {
        const dim3 wgs((grid.x + kBpcDecWgWaves - 1) / kBpcDecWgWaves);
        // (the decoder parks its finished planes in the scratch; poisoned: planes above a codeblock's MSB are never written)
        std::vector<uint32_t> plane_scratch((size_t)wgs.x * kBpcDecWgWaves * kEncScratchDwordsPerWave, 0xDEADBEEFu);
        a.plane_scratch = plane_scratch.data();
        emu::launch(wgs, dim3(64 * kBpcDecWgWaves), [&] { bpc_decode_kernel<false, kDecSmallPlanes>(a); });
    }
}

// the frame paths' decoder: lengths + offsets out of the stream (scan_stream_kernel), codewords read from the stream
// itself (bpc_decode_kernel<false, NP, true>); `stream` holds stream_shorts shorts (a load may start at the last pair).
// Returns the damaged-lengths flag.
int emu_bpc_decode_stream(const uint16_t *stream, unsigned stream_shorts, int aw, int ah, int wl, const int32_t *lut,
                          const int *geo, int32_t *coeffs, int *flag)
{
    const int ncb = (aw / 64) * (ah / 64);
    std::vector<int32_t> sizes(ncb), offsets(ncb);
    int32_t total = 0;
    int bad = 0;
    emu::launch(dim3(1), dim3(scan_threads(ncb)), [&] { scan_stream_kernel(stream, ncb, sizes.data(), offsets.data(), &total, &bad, 0); });
    BpcArgs a = mk(aw, ah, wl, lut, geo, nullptr, sizes.data(), flag);
    a.coeffs_out = coeffs;
    a.k = 0.0f; a.n_tables = 1;
    a.cw16 = stream; a.cw16_offsets = offsets.data(); a.cw16_total = &total; a.cw16_max = stream_shorts;
    const dim3 wgs(((unsigned)((a.nCB + 1) / 2) + kBpcDecWgWaves - 1) / kBpcDecWgWaves);
    std::vector<uint32_t> plane_scratch((size_t)wgs.x * kBpcDecWgWaves * kEncScratchDwordsPerWave, 0xDEADBEEFu);
    a.plane_scratch = plane_scratch.data();
    emu::launch(wgs, dim3(64 * kBpcDecWgWaves), [&] { bpc_decode_kernel<false, kDecSmallPlanes, true>(a); });
    return bad;
}

// -k > 0 from the packed stream (bpc_decode_kernel<true, NP, true>: both plane-count classes over the grid)
int emu_bpc_decode_stream_k(const uint16_t *stream, unsigned stream_shorts, int aw, int ah, int wl, const int32_t *lut,
                            const int *geo, int32_t *coeffs, int *flag, float k, int n_tables, int c16)
{
    const int ncb = (aw / 64) * (ah / 64);
    std::vector<int32_t> sizes(ncb), offsets(ncb);
    int32_t total = 0;
    int bad = 0;
    emu::launch(dim3(1), dim3(scan_threads(ncb)), [&] { scan_stream_kernel(stream, ncb, sizes.data(), offsets.data(), &total, &bad, 0); });
    BpcArgs a = mk(aw, ah, wl, lut, geo, nullptr, sizes.data(), flag);
    a.coeffs_out = coeffs;
    a.k = k; a.n_tables = n_tables;
    a.cw16 = stream; a.cw16_offsets = offsets.data(); a.cw16_total = &total; a.cw16_max = stream_shorts;
    const dim3 grid((unsigned)((a.nCB + 1) / 2));
    std::vector<uint32_t> plane_scratch((size_t)grid.x * kEncScratchDwordsPerWave, 0xDEADBEEFu);
    a.plane_scratch = plane_scratch.data();
    // c16: `coeffs` is an int16 Mallat array (the C16 instantiations)
    if (c16) {
        if (emu_bulk_compact(aw, ah, wl, geo)) emu::launch(grid, dim3(64), [&] { bpc_decode_kernel<true, kDecSmallPlanes, true, true, true>(a); });
        else emu::launch(grid, dim3(64), [&] { bpc_decode_kernel<true, kDecSmallPlanes, true, true>(a); });
    } else {
        if (emu_bulk_compact(aw, ah, wl, geo)) emu::launch(grid, dim3(64), [&] { bpc_decode_kernel<true, kDecSmallPlanes, true, false, true>(a); });
        else emu::launch(grid, dim3(64), [&] { bpc_decode_kernel<true, kDecSmallPlanes, true>(a); });
    }
    return bad;
}

// the same, the coefficients leaving as an int16 Mallat array (bpc_decode_kernel's C16 form)
int emu_bpc_decode_stream16(const uint16_t *stream, unsigned stream_shorts, int aw, int ah, int wl, const int32_t *lut,
                            const int *geo, int16_t *coeffs16, int *flag)
{
    const int ncb = (aw / 64) * (ah / 64);
    std::vector<int32_t> sizes(ncb), offsets(ncb);
    int32_t total = 0;
    int bad = 0;
    emu::launch(dim3(1), dim3(scan_threads(ncb)), [&] { scan_stream_kernel(stream, ncb, sizes.data(), offsets.data(), &total, &bad, 0); });
    BpcArgs a = mk(aw, ah, wl, lut, geo, nullptr, sizes.data(), flag);
    a.coeffs_out = reinterpret_cast<int32_t *>(coeffs16);
    a.k = 0.0f; a.n_tables = 1;
    a.cw16 = stream; a.cw16_offsets = offsets.data(); a.cw16_total = &total; a.cw16_max = stream_shorts;
    const dim3 wgs(((unsigned)((a.nCB + 1) / 2) + kBpcDecWgWaves - 1) / kBpcDecWgWaves);
    std::vector<uint32_t> plane_scratch((size_t)wgs.x * kBpcDecWgWaves * kEncScratchDwordsPerWave, 0xDEADBEEFu);
    a.plane_scratch = plane_scratch.data();
    emu::launch(wgs, dim3(64 * kBpcDecWgWaves), [&] { bpc_decode_kernel<false, kDecSmallPlanes, true, true>(a); });
    return bad;
}

// -cp 3: geo[6..8] = nRef, nSig, nSign; lut = [ref | sig | sign | cp_sig | cp_sign]
void emu_bpc3_encode(const void *coeffs, int is_float, int aw, int ah, int wl, const int32_t *lut, const int *geo,
                     int32_t *staging, int32_t *sizes, int *flag)
{
    BpcArgs a = mk(aw, ah, wl, lut, geo, nullptr, sizes, flag);
    a.coeffs_in = coeffs; a.is_float = is_float; a.n_tables = 1;
    std::vector<uint16_t> st16((size_t)a.nCB * 4096, 0xDEADu);
    a.staging16 = st16.data();
    memset(staging, 0xFF, (size_t)aw * ah * 4);
    const unsigned wgs3 = (unsigned)(((a.nCB + 1) / 2 + kBpc3WgWaves - 1) / kBpc3WgWaves);
    std::vector<uint32_t> plane_scratch((size_t)wgs3 * kBpc3WgWaves * kEncScratchDwordsPerWave, 0xDEADBEEFu);
    a.plane_scratch = plane_scratch.data();
    emu::launch(dim3(wgs3), dim3(64 * kBpc3WgWaves), [&] { bpc3_kernel<false>(a); });
    emu::launch(dim3((unsigned)a.nCB), dim3(256), [&] { widen_staging_kernel(st16.data(), sizes, 0, staging); });
}

void emu_bpc3_decode(const int32_t *staging, const int32_t *sizes, int aw, int ah, int wl, const int32_t *lut,
                     const int *geo, int32_t *coeffs, int *flag)
{
    BpcArgs a = mk(aw, ah, wl, lut, geo, const_cast<int32_t *>(staging), const_cast<int32_t *>(sizes), flag);
    a.coeffs_out = coeffs; a.n_tables = 1;
    const unsigned wgs3 = (unsigned)(((a.nCB + 1) / 2 + kBpc3WgWaves - 1) / kBpc3WgWaves);
    std::vector<uint32_t> plane_scratch((size_t)wgs3 * kBpc3WgWaves * kEncScratchDwordsPerWave, 0xDEADBEEFu);
    a.plane_scratch = plane_scratch.data();
    emu::launch(dim3(wgs3), dim3(64 * kBpc3WgWaves), [&] { bpc3_kernel<true>(a); });
}

int emu_pack(const int32_t *staging, const int32_t *sizes, int ncb, const uint16_t *header, uint16_t *out)
{
    std::vector<int32_t> offsets(ncb);
    int32_t total = 0;
    HeaderArg h;
    memset(&h, 0, sizeof h);
    if (header) { memcpy(h.h, header, sizeof h.h); h.has = 1; }
    emu::launch(dim3(1), dim3(scan_threads(ncb)), [&] { scan_sizes_kernel(sizes, ncb, offsets.data(), &total); });
    emu::launch(dim3((unsigned)ncb), dim3(256), [&] { pack_kernel<int32_t>(staging, sizes, offsets.data(), &total, ncb, h, out); });
    return total;
}

// the same from the encoders' 16-bit staging (the frame paths' pack)
int emu_pack16(const uint16_t *staging16, const int32_t *sizes, int ncb, const uint16_t *header, uint16_t *out)
{
    std::vector<int32_t> offsets(ncb);
    int32_t total = 0;
    HeaderArg h;
    memset(&h, 0, sizeof h);
    if (header) { memcpy(h.h, header, sizeof h.h); h.has = 1; }
    emu::launch(dim3(1), dim3(scan_threads(ncb)), [&] { scan_sizes_kernel(sizes, ncb, offsets.data(), &total); });
    emu::launch(dim3((unsigned)ncb), dim3(256), [&] { pack_kernel<uint16_t>(staging16, sizes, offsets.data(), &total, ncb, h, out); });
    return total;
}

int emu_unpack(const uint16_t *stream, int ncb, int32_t *staging, int32_t *sizes)
{
    std::vector<int32_t> offsets(ncb);
    int32_t total = 0;
    int flag = 0;
    memset(staging, 0xFF, (size_t)ncb * 4096 * 4);
    emu::launch(dim3((unsigned)((ncb + 255) / 256)), dim3(256), [&] { read_sizes_kernel(stream, ncb, sizes, &flag); });
    emu::launch(dim3(1), dim3(scan_threads(ncb)), [&] { scan_sizes_kernel(sizes, ncb, offsets.data(), &total); });
    emu::launch(dim3((unsigned)ncb), dim3(256), [&] { unpack_kernel(stream, sizes, offsets.data(), ncb, staging); });
    return flag;
}

}  // extern "C"
