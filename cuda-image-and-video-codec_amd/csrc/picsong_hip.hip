// picsong_hip.hip -- C-ABI implementation (include/picsong_hip.h): host launch logic for the
// gfx950 kernels in dwt_kernels.hpp / bpc_kernels.hpp / pack_kernels.hpp.
// No CPU fallback exists: without a GPU every device entry point returns PICSONG_ERR_NODEVICE.
#include "../../include/picsong_hip.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "bpc_kernels.hpp"
#include "dwt_kernels.hpp"
#include "launch_plan.hpp"
#include "pack_kernels.hpp"

using namespace picsong;

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return fail(PICSONG_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                               \
    } while (0)


}  // namespace

struct picsong_ctx {
    picsong_params p;
    int device;
    int aw, ah, ncb;
    size_t P, extra;
    bool fast_div;        // 9/7 synthesis: reciprocal form of the divisions verified for this qs
    bool c16;             // frame paths: coded coefficients travel as int16 between transform and coder (coef16_ok)
    bool c16_dec;         // decode frame paths: ... and between decoder and synthesis (dec_c16_ok)
    int bulk_compact[3];  // -k > 0: the component's table geometry lets every codeblock use the compact LDS copy (-1: not looked at yet)
    bool pipelined;       // picsong_ctx_set_pipelined: other frames share the GPU (throughput over latency)
    // LUT
    picsong_lut_info li[3];
    int32_t *d_lut[3];
    bool has_lut[3];
    bool lut_borrowed[3]; // d_lut[k] is the caller's device table (picsong_ctx_set_lut_device): not freed here
    // small scratch
    int32_t *d_offsets;   // nCB
    uint32_t *d_plane_scratch;   // encoder: planes below the 8 held in registers, 8 KB per wave (lazy)
    int32_t *d_total;     // 1
    int *d_flag;          // 1
    int32_t *h_pinned;    // [0] total, [1] flag
    // frame pipeline workspace (lazy)
    void *d_coef;         // T[P + extra]
    int32_t *d_staging;   // int32[P]
    int32_t *d_sizes;     // int32[nCB]
    int32_t *d_coef_i;    // int32[P] (decode)
    // batched frame path (picsong_encode_frames): workspaces for batch_cap frames, laid frame after frame
    int batch_cap;
    void *b_coef; int32_t *b_staging, *b_sizes, *b_offsets, *b_total;
    int32_t *b_coef_i;    // decoded coefficients of a batch (picsong_decode_frames; lazy)
    int b_coef_i_cap;
    uint32_t *b_plane_scratch;
    int32_t *h_totals;    // pinned, batch_cap
    int last_batch;       // frames of the most recent picsong_encode_frames
    // stage profiling (HIP events on the launch stream)
    std::vector<hipEvent_t> *prof_ev;   // 4 per frame
    int prof_cap, prof_n;
};

template <int BAND>
static void launch_inv(const picsong_ctx *c, const InvLaunch &f, hipStream_t s, unsigned frames = 1)
{
    dim3 grid(f.gx, f.gy, frames);
    // f.fast: the 9/7 divisions in their reciprocal form (verified for this context's qs at creation); the vector
    // launches of such a context are the lean kernel's (PICSONG_DWT_INV97=0: dwt_inv_kernel's FAST instantiations)
    static const bool lean97 = !(getenv("PICSONG_DWT_INV97") && atoi(getenv("PICSONG_DWT_INV97")) == 0);
    // (a coarsest level that also writes pixels, wl = 1, stays with dwt_inv_kernel)
    const bool l97 = c->p.lossy && f.fast && lean97 && f.vec && !(f.a.first && f.a.dst_u8);
    if (f.a.c16) {
        // the decode frame paths' 16-bit coefficients (dec_c16_ok: vector kernels, 9/7 through the lean kernel, the
        // coarsest level never the one that writes pixels)
        if (c->p.lossy) {
            if (f.a.dst_u8) {
                if (f.a.one_div) dwt_inv97_kernel<BAND, true, false, true, true><<<grid, 256, 0, s>>>(f.a);
                else dwt_inv97_kernel<BAND, true, false, false, true><<<grid, 256, 0, s>>>(f.a);
            } else if (f.a.first) {
                if (f.a.one_div) dwt_inv97_kernel<BAND, false, true, true, true><<<grid, 256, 0, s>>>(f.a);
                else dwt_inv97_kernel<BAND, false, true, false, true><<<grid, 256, 0, s>>>(f.a);
            } else {
                if (f.a.one_div) dwt_inv97_kernel<BAND, false, false, true, true><<<grid, 256, 0, s>>>(f.a);
                else dwt_inv97_kernel<BAND, false, false, false, true><<<grid, 256, 0, s>>>(f.a);
            }
        } else if (f.a.dst_u8) dwt_inv_kernel<int, false, BAND, true, true, false, true><<<grid, 256, 0, s>>>(f.a);
        else dwt_inv_kernel<int, false, BAND, true, false, false, true><<<grid, 256, 0, s>>>(f.a);
        return;
    }
    if (l97) {
        if (f.a.dst_u8) {
            if (f.a.one_div) dwt_inv97_kernel<BAND, true, false, true><<<grid, 256, 0, s>>>(f.a);
            else dwt_inv97_kernel<BAND, true, false, false><<<grid, 256, 0, s>>>(f.a);
        } else if (f.a.first) {
            if (f.a.one_div) dwt_inv97_kernel<BAND, false, true, true><<<grid, 256, 0, s>>>(f.a);
            else dwt_inv97_kernel<BAND, false, true, false><<<grid, 256, 0, s>>>(f.a);
        } else {
            if (f.a.one_div) dwt_inv97_kernel<BAND, false, false, true><<<grid, 256, 0, s>>>(f.a);
            else dwt_inv97_kernel<BAND, false, false, false><<<grid, 256, 0, s>>>(f.a);
        }
    } else if (f.vec && f.a.dst_u8) {   // finest level of the frame path: pixels out, clamp fused
        if (c->p.lossy && f.fast) dwt_inv_kernel<float, true, BAND, true, true, true><<<grid, 256, 0, s>>>(f.a);
        else if (c->p.lossy) dwt_inv_kernel<float, true, BAND, true, true><<<grid, 256, 0, s>>>(f.a);
        else dwt_inv_kernel<int, false, BAND, true, true><<<grid, 256, 0, s>>>(f.a);
    } else if (f.vec) {
        if (c->p.lossy && f.fast) dwt_inv_kernel<float, true, BAND, true, false, true><<<grid, 256, 0, s>>>(f.a);
        else if (c->p.lossy) dwt_inv_kernel<float, true, BAND, true><<<grid, 256, 0, s>>>(f.a);
        else dwt_inv_kernel<int, false, BAND, true><<<grid, 256, 0, s>>>(f.a);
    } else {
        if (c->p.lossy && f.fast) dwt_inv_kernel<float, true, BAND, false, false, true><<<grid, 256, 0, s>>>(f.a);
        else if (c->p.lossy) dwt_inv_kernel<float, true, BAND, false><<<grid, 256, 0, s>>>(f.a);
        else dwt_inv_kernel<int, false, BAND, false><<<grid, 256, 0, s>>>(f.a);
    }
}

template <int BAND, bool VEC>
static void launch_fwd_v(bool lossy, const FwdLaunch &f, dim3 grid, hipStream_t s)
{
    if (lossy) {
        if (f.u8) dwt_fwd_kernel<float, true, true, BAND, VEC><<<grid, 256, 0, s>>>(f.a);
        else dwt_fwd_kernel<float, true, false, BAND, VEC><<<grid, 256, 0, s>>>(f.a);
    } else {
        if (f.u8) dwt_fwd_kernel<int, false, true, BAND, VEC><<<grid, 256, 0, s>>>(f.a);
        else dwt_fwd_kernel<int, false, false, BAND, VEC><<<grid, 256, 0, s>>>(f.a);
    }
}

template <int BAND>
static void launch_fwd(const picsong_ctx *c, const FwdLaunch &f, hipStream_t s, unsigned frames = 1)
{
    dim3 grid(f.gx, f.gy, frames);
    if (f.vec) launch_fwd_v<BAND, true>(c->p.lossy != 0, f, grid, s);
    else launch_fwd_v<BAND, false>(c->p.lossy != 0, f, grid, s);
}

static void launch_fwd2(bool lossy, const Fwd2Launch &f, hipStream_t s, unsigned frames = 1)
{
    dim3 grid(f.gx, f.gy, frames);
    if (f.a.l0.c16) {                     // frame paths: coded subbands as int16 (DwtFwdArgs::c16)
        if (lossy) dwt_fwd2_kernel<float, true, true, kF2PairsLossy, true><<<grid, 256, 0, s>>>(f.a);
        else dwt_fwd2_kernel<int, false, true, kF2Pairs, true><<<grid, 256, 0, s>>>(f.a);
        return;
    }
    if (lossy) dwt_fwd2_kernel<float, true, true, kF2PairsLossy><<<grid, 256, 0, s>>>(f.a);
    else dwt_fwd2_kernel<int, false, true, kF2Pairs><<<grid, 256, 0, s>>>(f.a);
}

// Levels [from, end) of a forward plan, one launch each.  `frames` = grid.z of a batched call.
static int launch_fwd_levels(const picsong_ctx *c, const std::vector<FwdLaunch> &plan, size_t from, hipStream_t s,
                             unsigned frames = 1)
{
    for (size_t l = from; l < plan.size(); l++) {
        const FwdLaunch &f = plan[l];
        switch (f.band) {
        case 32: launch_fwd<32>(c, f, s, frames); break;
        case 16: launch_fwd<16>(c, f, s, frames); break;
        case 8: launch_fwd<8>(c, f, s, frames); break;
        default: launch_fwd<4>(c, f, s, frames); break;
        }
        HIP_TRY(hipGetLastError());
    }
    return PICSONG_OK;
}


extern "C" {

const char *picsong_last_error(void) { return g_err; }
const char *picsong_version(void) { return "picsong-mi355x 0.1 (gfx950)"; }

int picsong_pad_dim(int v) { return ((v + PICSONG_CB - 1) / PICSONG_CB) * PICSONG_CB; }

size_t picsong_dwt_extra(int aw, int ah, int wl)
{
    size_t e = 0;
    for (int l = 1; l < wl; l++) e += (size_t)(aw >> l) * (size_t)(ah >> l);
    return e;
}

size_t picsong_max_stream_shorts(int aw, int ah)
{
    size_t ncb = (size_t)(aw / PICSONG_CB) * (size_t)(ah / PICSONG_CB);
    return PICSONG_HDR_SHORTS + 2 * ncb + (size_t)aw * (size_t)ah + 1;
}

// ---------------------------------------------------------------------------------------------
// header
// ---------------------------------------------------------------------------------------------
int picsong_header_pack(const picsong_params *p, uint16_t o[PICSONG_HDR_SHORTS])
{
    if (!p || !o) return fail(PICSONG_ERR_ARG, "header_pack: null argument");
    if (p->width <= 0 || p->height <= 0 || p->height > 65535 || p->components <= 0 || p->frames < 0 ||
        p->frames >= (1 << 17) ||
        (uint64_t)p->width * (uint64_t)p->height * (uint64_t)p->components >= ((uint64_t)1 << 32))
        return fail(PICSONG_ERR_ARG, "header_pack: a field exceeds its width (height 16 bits, frames 17, samples 32)");
    const uint32_t n = (uint32_t)p->width * (uint32_t)p->height * (uint32_t)p->components;
    const int qs4 = (int)(p->qs * 10000), k3 = (int)(p->k * 1000);
    o[0] = (uint16_t)(n & 0xFFFFu);
    o[1] = (uint16_t)(n >> 16);
    o[2] = (uint16_t)((p->cp == 2 ? 0 : 1) | (p->cb_height << 1) | (p->cb_width << 8) | ((p->wl & 1) << 15));
    o[3] = (uint16_t)(((p->wl & 7) >> 1) | (p->bit_depth << 3) | ((p->lossy ? 1 : 0) << 10) | ((qs4 & 31) << 11));
    o[4] = (uint16_t)((qs4 >> 5) | ((p->components & 127) << 9));
    o[5] = (uint16_t)((p->components >> 7) | ((p->is_rgb ? 1 : 0) << 7) | (p->height << 8));
    o[6] = (uint16_t)((p->height >> 8) | (0 << 8) | (p->bit_depth << 9) | (0 << 14) | ((p->frames & 1) << 15));
    o[7] = (uint16_t)((p->frames >> 1) & 0xFFFF);
    o[8] = (uint16_t)k3;
    return PICSONG_OK;
}

int picsong_header_unpack(const uint16_t e[PICSONG_HDR_SHORTS], picsong_params *p)
{
    if (!p || !e) return fail(PICSONG_ERR_ARG, "header_unpack: null argument");
    memset(p, 0, sizeof *p);
    const uint32_t n = (uint32_t)e[0] | ((uint32_t)e[1] << 16);
    p->cp = (e[2] & 1) ? 3 : 2;
    p->cb_height = (e[2] >> 1) & 127;
    p->cb_width = (e[2] >> 8) & 127;
    p->wl = ((e[2] >> 15) & 1) | ((e[3] & 7) << 1);
    p->bit_depth = (e[3] >> 3) & 127;
    p->lossy = (e[3] >> 10) & 1;
    p->qs = (float)((((e[3] >> 11) & 31) | ((e[4] & 511) << 5)) / 10000.0);   // DecodingEngine.cu:161
    p->components = ((e[4] >> 9) & 127) | ((e[5] & 127) << 9);
    p->is_rgb = (e[5] >> 7) & 1;
    p->height = ((e[5] >> 8) & 255) | ((e[6] & 255) << 8);
    p->frames = ((e[6] >> 15) & 1) | ((int)e[7] << 1);
    p->k = (float)(e[8] / 1000.0);
    if (p->height <= 0 || p->components <= 0) return fail(PICSONG_ERR_ARG, "header_unpack: bad header");
    p->width = (int)(n / (uint32_t)p->height / (uint32_t)p->components);       // DecodingEngine.cu:146
    return PICSONG_OK;
}

// ---------------------------------------------------------------------------------------------
// LUT text parser
// ---------------------------------------------------------------------------------------------
static int lut_section(const std::string &folder, const char *stem, int component, int file_index, int C, int nBp,
                       int wl, int32_t *T, int base, size_t cap)
{
    static const char *suffix[4] = { ".txt_", "R.txt_", "G.txt_", "B.txt_" };
    std::string path = folder + stem + suffix[component & 3] + std::to_string(file_index);
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return fail(PICSONG_ERR_IO, "cannot open LUT file %s", path.c_str());
    int i = base, prev = -1, lvl, sb, bp, v[16];
    for (;;) {
        if (fscanf(f, "%d %d %d :", &lvl, &sb, &bp) != 3) break;
        bool ok = true;
        for (int c = 0; c < C; c++)
            if (fscanf(f, "%d", &v[c]) != 1) { ok = false; break; }
        if (!ok) break;
        if (bp <= prev) {
            // remaining planes of the previous group get the 7-bit mid value (IOManager.ipp:459)
            for (int z = 0; z < (nBp - prev - 1) * C; z++) {
                size_t at = (size_t)(i + prev * C + z + C);
                if (at < cap) T[at] = 64;
            }
            i += nBp * C;
        }
        if ((lvl + 1) > wl && sb > 0) break;
        prev = bp;
        for (int c = 0; c < C; c++) {
            size_t at = (size_t)(i + bp * C + c);
            if (at < cap) T[at] = v[c];
        }
    }
    fclose(f);
    return PICSONG_OK;
}

int picsong_lut_load(const char *folder_c, int component, int wl, int fill, picsong_lut_info *info,
                     int32_t *table, size_t cap)
{
    return picsong_lut_load_k(folder_c, component, wl, fill, 1, info, table, cap);
}

int picsong_lut_load_k(const char *folder_c, int component, int wl, int fill, int n_tables, picsong_lut_info *info,
                       int32_t *table, size_t cap)
{
    if (!folder_c || !info) return fail(PICSONG_ERR_ARG, "lut_load: null argument");
    if (wl < 1 || wl > 10) return fail(PICSONG_ERR_ARG, "lut_load: wl %d out of range", wl);
    std::string folder(folder_c);
    if (!folder.empty() && folder.back() != '/') folder += '/';
    FILE *f = fopen((folder + "header.txt").c_str(), "rb");
    if (!f) return fail(PICSONG_ERR_IO, "cannot open %sheader.txt", folder.c_str());
    int v[8], n = 0;
    char line[256];
    while (n < 8 && fgets(line, sizeof line, f)) {
        const char *semi = strchr(line, ';');
        if (semi) v[n++] = atoi(semi + 1);
    }
    fclose(f);
    if (n < 8) return fail(PICSONG_ERR_IO, "%sheader.txt: expected 8 KEY;value lines", folder.c_str());
    info->n_bitplanes = v[0]; info->n_subbands = v[1]; info->ctx_ref = v[2]; info->ctx_sign = v[3];
    info->ctx_sig = v[4]; info->precision = v[5]; info->n_files = v[6];
    info->n_bp_files = v[7] > 32 ? 32 : v[7];
    const int nBp = v[0], nS = v[1];
    info->n_ref = nS * nBp * info->ctx_ref * wl + nBp * info->ctx_ref;
    info->n_sig = nS * nBp * info->ctx_sig * wl + nBp * info->ctx_sig;
    info->n_sign = nS * nBp * info->ctx_sign * wl + nBp * info->ctx_sign;
    if (n_tables <= 0) n_tables = info->n_bp_files;
    if (n_tables < 1) n_tables = 1;
    info->n_tables = n_tables;
    info->cp = 2;
    if (!table) return PICSONG_OK;
    const size_t total = (size_t)info->n_ref + info->n_sig + info->n_sign;
    if (cap < total * (size_t)n_tables)
        return fail(PICSONG_ERR_ARG, "lut_load: table capacity %zu < %zu", cap, total * (size_t)n_tables);
    if (info->ctx_ref > 16 || info->ctx_sig > 16 || info->ctx_sign > 16)
        return fail(PICSONG_ERR_ARG, "lut_load: context counts above 16 are not supported");
    for (size_t i = 0; i < total * (size_t)n_tables; i++) table[i] = fill;
    for (int j = 0; j < n_tables; j++) {
        int32_t *T = table + (size_t)j * total;
        // a group-change fill may reach past its table: inside the array it lands in the next table
        // (which is parsed afterwards), at the very end it is dropped -- as in the oracle
        const size_t room = total * (size_t)(n_tables - j);
        int rc;
        if ((rc = lut_section(folder, "ref", component, j, info->ctx_ref, nBp, wl, T, 0, room))) return rc;
        if ((rc = lut_section(folder, "sig", component, j, info->ctx_sig, nBp, wl, T, info->n_ref, room))) return rc;
        if ((rc = lut_section(folder, "sign", component, j, info->ctx_sign, nBp, wl, T, info->n_ref + info->n_sig,
                              room))) return rc;
    }
    return PICSONG_OK;
}

int picsong_lut_load_cp(const char *folder_c, int component, int wl, int fill, int cp, picsong_lut_info *info,
                        int32_t *table, size_t cap)
{
    if (cp != 3) {
        const int rc = picsong_lut_load_k(folder_c, component, wl, fill, 1, info, table, cap);
        if (rc == PICSONG_OK) info->cp = 2;
        return rc;
    }
    if (!folder_c || !info) return fail(PICSONG_ERR_ARG, "lut_load: null argument");
    int rc = picsong_lut_load_k(folder_c, component, wl, fill, 1, info, nullptr, 0);      // header + section sizes
    if (rc) return rc;
    info->cp = 3;
    if (!table) return PICSONG_OK;
    const size_t b3 = (size_t)info->n_ref + info->n_sig + info->n_sign, total = b3 + info->n_sig + info->n_sign;
    if (cap < total) return fail(PICSONG_ERR_ARG, "lut_load: table capacity %zu < %zu", cap, total);
    std::string folder(folder_c);
    if (!folder.empty() && folder.back() != '/') folder += '/';
    for (size_t i = 0; i < total; i++) table[i] = fill;
    const int nBp = info->n_bitplanes;
    if ((rc = lut_section(folder, "ref", component, 0, info->ctx_ref, nBp, wl, table, 0, total))) return rc;
    if ((rc = lut_section(folder, "sig", component, 0, info->ctx_sig, nBp, wl, table, info->n_ref, total))) return rc;
    if ((rc = lut_section(folder, "sign", component, 0, info->ctx_sign, nBp, wl, table, info->n_ref + info->n_sig, total))) return rc;
    if ((rc = lut_section(folder, "cp_sig", component, 0, info->ctx_sig, nBp, wl, table, (int)b3, total))) return rc;
    if ((rc = lut_section(folder, "cp_sign", component, 0, info->ctx_sign, nBp, wl, table, (int)b3 + info->n_sig, total))) return rc;
    return PICSONG_OK;
}

// ---------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------
int picsong_ctx_create(const picsong_params *p, int device, picsong_ctx **out)
{
    if (!p || !out) return fail(PICSONG_ERR_ARG, "ctx_create: null argument");
    *out = nullptr;
    // Launcher.cu:132 validation + header limits
    if (p->width <= 0 || p->height <= 0) return fail(PICSONG_ERR_ARG, "xSize/ySize must be positive");
    if (p->wl < 1 || p->wl > 7) return fail(PICSONG_ERR_ARG, "wl %d outside 1..7", p->wl);
    if (p->cp != 2 && p->cp != 3) return fail(PICSONG_ERR_ARG, "cp %d: 2 or 3 coding passes", p->cp);
    if (p->cp == 3 && p->k > 0.0f) return fail(PICSONG_ERR_ARG, "-cp 3 has no complexity-scalable mode (k must be 0)");
    if (!(p->k >= 0.0f && p->k <= 65.535f)) return fail(PICSONG_ERR_ARG, "k %g outside [0, 65.535]", p->k);
    if (p->lossy && !(p->qs > 0.0f && p->qs <= 1.0f)) return fail(PICSONG_ERR_ARG, "qs %g outside (0,1]", p->qs);
    if (p->bit_depth != 8) return fail(PICSONG_ERR_ARG, "only 8-bit samples are implemented");
    if (!((p->components == 1 && !p->is_rgb) || (p->components == 3 && p->is_rgb)))
        return fail(PICSONG_ERR_ARG, "components must be 1 (grey) or 3 with is_rgb (got %d, is_rgb %d)", p->components,
                    p->is_rgb);
    // header field widths (BitStreamBuilder.cpp:54-93): height 16 bits, frames 17 bits, samples 32 bits
    if (p->height > 65535) return fail(PICSONG_ERR_ARG, "ySize %d exceeds the header's 16 bits", p->height);
    if (p->frames < 0 || p->frames >= (1 << 17)) return fail(PICSONG_ERR_ARG, "frames %d outside the header's 17 bits", p->frames);
    if ((uint64_t)p->width * (uint64_t)p->height * (uint64_t)p->components >= ((uint64_t)1 << 32))
        return fail(PICSONG_ERR_ARG, "xSize * ySize * components exceeds the header's 32 bits");
    const int aw = picsong_pad_dim(p->width), ah = picsong_pad_dim(p->height);
    // the mirror padding of IOManager::loadFrameCAdaptedSizes is only defined while the added columns / rows
    // do not outnumber the frame's own (picsong_pad_frame_host)
    if (aw - p->width > p->width || ah - p->height > p->height)
        return fail(PICSONG_ERR_ARG, "frame %dx%d is too small to be mirror-padded to %dx%d", p->width, p->height, aw, ah);
    if ((aw >> (p->wl - 1)) < 4 || (ah >> (p->wl - 1)) < 4 || ((aw >> (p->wl - 1)) & 1) || ((ah >> (p->wl - 1)) & 1))
        return fail(PICSONG_ERR_ARG, "image %dx%d too small for %d wavelet levels", aw, ah, p->wl);
    if ((size_t)aw * (size_t)ah >= ((size_t)1 << 30)) return fail(PICSONG_ERR_ARG, "frame too large");

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(PICSONG_ERR_NODEVICE, "no HIP device: this library has no CPU path");
    if (device < 0 || device >= ndev) return fail(PICSONG_ERR_ARG, "device %d of %d", device, ndev);
    HIP_TRY(hipSetDevice(device));

    picsong_ctx *c = new picsong_ctx();
    memset(c, 0, sizeof *c);
    c->p = *p;
    c->device = device;
    c->aw = aw; c->ah = ah;
    c->ncb = (aw / PICSONG_CB) * (ah / PICSONG_CB);
    c->P = (size_t)aw * (size_t)ah;
    c->extra = picsong_dwt_extra(aw, ah, p->wl);
    c->fast_div = p->lossy != 0 && dequant_fast_ok(p->qs, p->wl);
    // 8-bit samples: 128 after the level shift; 255 covers the chroma differences of the RGB path's RCT
    c->c16 = p->bit_depth == 8 && dwt_c16_geometry_ok(c->aw, c->ah, p->wl) &&
             coef16_ok(p->lossy != 0, p->wl, p->qs, p->is_rgb ? 255 : 128);
    c->bulk_compact[0] = c->bulk_compact[1] = c->bulk_compact[2] = -1;
    // (an RGB context: picsong_decode_rgb_frame's three components; the plane-by-plane calls keep the 32-bit arrays)
    c->c16_dec = p->cp != 3 && p->bit_depth == 8 &&
                 dec_c16_ok(p->lossy != 0, p->wl, p->qs, p->is_rgb ? 255 : 128, c->aw, c->ah, c->fast_div);
    hipError_t e = hipMalloc(&c->d_offsets, sizeof(int32_t) * (size_t)c->ncb);
    if (e == hipSuccess) e = hipMalloc(&c->d_total, sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc(&c->d_flag, sizeof(int));
    if (e == hipSuccess) e = hipHostMalloc(&c->h_pinned, 2 * sizeof(int32_t));
    if (e == hipSuccess) e = hipMemset(c->d_flag, 0, sizeof(int));
    if (e == hipSuccess) e = hipMemset(c->d_total, 0, sizeof(int32_t));
    if (e != hipSuccess) {
        picsong_ctx_destroy(c);
        return fail(PICSONG_ERR_HIP, "ctx_create: %s", hipGetErrorString(e));
    }
    *out = c;
    return PICSONG_OK;
}

void picsong_ctx_destroy(picsong_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    for (int k = 0; k < 3; k++)
        if (c->d_lut[k] && !c->lut_borrowed[k]) (void)hipFree(c->d_lut[k]);
    if (c->d_offsets) (void)hipFree(c->d_offsets);
    if (c->d_plane_scratch) (void)hipFree(c->d_plane_scratch);
    if (c->d_total) (void)hipFree(c->d_total);
    if (c->d_flag) (void)hipFree(c->d_flag);
    if (c->h_pinned) (void)hipHostFree(c->h_pinned);
    if (c->d_coef) (void)hipFree(c->d_coef);
    if (c->d_staging) (void)hipFree(c->d_staging);
    if (c->d_sizes) (void)hipFree(c->d_sizes);
    if (c->d_coef_i) (void)hipFree(c->d_coef_i);
    if (c->b_coef) (void)hipFree(c->b_coef);
    if (c->b_staging) (void)hipFree(c->b_staging);
    if (c->b_sizes) (void)hipFree(c->b_sizes);
    if (c->b_offsets) (void)hipFree(c->b_offsets);
    if (c->b_total) (void)hipFree(c->b_total);
    if (c->b_plane_scratch) (void)hipFree(c->b_plane_scratch);
    if (c->b_coef_i) (void)hipFree(c->b_coef_i);
    c->b_coef_i = nullptr; c->b_coef_i_cap = 0;
    if (c->h_totals) (void)hipHostFree(c->h_totals);
    if (c->prof_ev) {
        for (hipEvent_t e : *c->prof_ev) (void)hipEventDestroy(e);
        delete c->prof_ev;
    }
    delete c;
}

int picsong_ctx_set_lut_component(picsong_ctx *c, int comp, const picsong_lut_info *info, const int32_t *host_table)
{
    if (!c || !info || !host_table) return fail(PICSONG_ERR_ARG, "set_lut: null argument");
    if (comp < 0 || comp > 2) return fail(PICSONG_ERR_ARG, "set_lut: component %d outside 0..2", comp);
    // the context formation of BPCEngine.cu:222-308 is fixed to 9 / 4 / 1 contexts
    if (info->ctx_sig != 9 || info->ctx_sign != 4 || info->ctx_ref != 1)
        return fail(PICSONG_ERR_ARG, "LUT contexts must be 9/4/1 (sig/sign/ref), got %d/%d/%d", info->ctx_sig,
                    info->ctx_sign, info->ctx_ref);
    if (info->precision < 1 || info->precision > 8) return fail(PICSONG_ERR_ARG, "LUT precision %d", info->precision);
    const bool cp3 = c->p.cp == 3;
    if (cp3 != (info->cp == 3))
        return fail(PICSONG_ERR_ARG, "the context codes %d passes, the table is laid out for %d (picsong_lut_load_cp)",
                    c->p.cp, info->cp == 3 ? 3 : 2);
    const size_t one = (size_t)info->n_ref + (cp3 ? 2 : 1) * ((size_t)info->n_sig + info->n_sign);
    const int n_tables = info->n_tables > 0 ? info->n_tables : 1;
    if (one + (size_t)kLutSlack > (size_t)(cp3 ? kLutLdsMax3 : kLutLdsMax))
        return fail(PICSONG_ERR_ARG, "LUT table of %zu entries exceeds the %d the coder kernels hold in LDS", one,
                    (cp3 ? kLutLdsMax3 : kLutLdsMax) - kLutSlack);
    const size_t total = one * (size_t)n_tables;
    for (size_t i = 0; i < total; i++)
        if (host_table[i] < 0 || host_table[i] > 255)
            return fail(PICSONG_ERR_ARG, "LUT entry %zu = %d outside 0..255", i, host_table[i]);
    HIP_TRY(hipSetDevice(c->device));
    if (c->d_lut[comp] && !c->lut_borrowed[comp]) (void)hipFree(c->d_lut[comp]);
    c->d_lut[comp] = nullptr;
    c->lut_borrowed[comp] = false;
    HIP_TRY(hipMalloc(&c->d_lut[comp], total * sizeof(int32_t)));
    HIP_TRY(hipMemcpy(c->d_lut[comp], host_table, total * sizeof(int32_t), hipMemcpyHostToDevice));
    c->li[comp] = *info;
    c->li[comp].n_tables = n_tables;
    c->has_lut[comp] = true;
    c->bulk_compact[comp] = -1;
    return PICSONG_OK;
}

int picsong_ctx_set_lut_device(picsong_ctx *c, int comp, const picsong_lut_info *info, const int32_t *d_table)
{
    if (!c || !info || !d_table) return fail(PICSONG_ERR_ARG, "set_lut_device: null argument");
    if (comp < 0 || comp > 2) return fail(PICSONG_ERR_ARG, "set_lut_device: component %d outside 0..2", comp);
    if (info->ctx_sig != 9 || info->ctx_sign != 4 || info->ctx_ref != 1)
        return fail(PICSONG_ERR_ARG, "LUT contexts must be 9/4/1 (sig/sign/ref), got %d/%d/%d", info->ctx_sig,
                    info->ctx_sign, info->ctx_ref);
    if (info->precision < 1 || info->precision > 8) return fail(PICSONG_ERR_ARG, "LUT precision %d", info->precision);
    picsong_lut_info li = *info;
    // section sizes follow from the geometry (IO/IOManager.ipp:431-433) when the caller left them 0
    const int wl = c->p.wl;
    if (li.n_ref <= 0) li.n_ref = li.n_subbands * li.n_bitplanes * li.ctx_ref * wl + li.n_bitplanes * li.ctx_ref;
    if (li.n_sig <= 0) li.n_sig = li.n_subbands * li.n_bitplanes * li.ctx_sig * wl + li.n_bitplanes * li.ctx_sig;
    if (li.n_sign <= 0) li.n_sign = li.n_subbands * li.n_bitplanes * li.ctx_sign * wl + li.n_bitplanes * li.ctx_sign;
    // the borrowed table is laid out for the context's coding passes: [ref | sig | sign] for -cp 2, the five sections
    // [ref | sig | sign | cp_sig | cp_sign] for -cp 3 (what bpc3_kernel copies to LDS); `info->cp` as the façade's
    // geometry() leaves it (0) means "the context's"; a table declared for the other mode is refused
    const bool cp3 = c->p.cp == 3;
    if (info->cp != 0 && cp3 != (info->cp == 3))
        return fail(PICSONG_ERR_ARG, "the context codes %d passes, the device table is laid out for %d", c->p.cp,
                    info->cp == 3 ? 3 : 2);
    li.cp = c->p.cp;
    const size_t one = (size_t)li.n_ref + (cp3 ? 2 : 1) * ((size_t)li.n_sig + li.n_sign);
    if (one + (size_t)kLutSlack > (size_t)(cp3 ? kLutLdsMax3 : kLutLdsMax))
        return fail(PICSONG_ERR_ARG, "LUT table of %zu entries exceeds the %d the coder kernels hold in LDS", one,
                    (cp3 ? kLutLdsMax3 : kLutLdsMax) - kLutSlack);
    if (li.n_tables <= 0) li.n_tables = 1;
    if (c->d_lut[comp] && !c->lut_borrowed[comp]) (void)hipFree(c->d_lut[comp]);
    c->d_lut[comp] = const_cast<int32_t *>(d_table);
    c->lut_borrowed[comp] = true;
    c->li[comp] = li;
    c->has_lut[comp] = true;
    c->bulk_compact[comp] = -1;
    return PICSONG_OK;
}

int picsong_ctx_set_lut(picsong_ctx *c, const picsong_lut_info *info, const int32_t *host_table)
{
    return picsong_ctx_set_lut_component(c, 0, info, host_table);
}

int picsong_ctx_set_pipelined(picsong_ctx *c, int on)
{
    if (!c) return fail(PICSONG_ERR_ARG, "set_pipelined: null context");
    c->pipelined = on != 0;
    return PICSONG_OK;
}

int picsong_ctx_padded_dims(const picsong_ctx *c, int *aw, int *ah, int *ncb)
{
    if (!c) return fail(PICSONG_ERR_ARG, "null ctx");
    if (aw) *aw = c->aw;
    if (ah) *ah = c->ah;
    if (ncb) *ncb = c->ncb;
    return PICSONG_OK;
}

// ---------------------------------------------------------------------------------------------
// level shift
// ---------------------------------------------------------------------------------------------
int picsong_level_shift_fwd(picsong_ctx *c, const uint8_t *d_in, void *d_out, void *stream)
{
    if (!c || !d_in || !d_out) return fail(PICSONG_ERR_ARG, "level_shift_fwd: null argument");
    hipStream_t s = (hipStream_t)stream;
    const size_t n4 = c->P / 4;
    const int off = 1 << (c->p.bit_depth - 1);
    const int grid = (int)((n4 + 255) / 256 > 4096 ? 4096 : (n4 + 255) / 256);
    if (c->p.lossy) level_shift_fwd_kernel<float><<<grid, 256, 0, s>>>(d_in, (float *)d_out, n4, off);
    else level_shift_fwd_kernel<int32_t><<<grid, 256, 0, s>>>(d_in, (int32_t *)d_out, n4, off);
    HIP_TRY(hipGetLastError());
    return PICSONG_OK;
}

int picsong_level_shift_inv(picsong_ctx *c, void *d_data, void *stream)
{
    if (!c || !d_data) return fail(PICSONG_ERR_ARG, "level_shift_inv: null argument");
    hipStream_t s = (hipStream_t)stream;
    const int off = 1 << (c->p.bit_depth - 1);
    const int grid = (int)((c->P + 255) / 256 > 8192 ? 8192 : (c->P + 255) / 256);
    if (c->p.lossy) level_shift_inv_f32_kernel<<<grid, 256, 0, s>>>((float *)d_data, c->P, (float)off);
    else level_shift_inv_i32_kernel<<<grid, 256, 0, s>>>((int32_t *)d_data, c->P, off);
    HIP_TRY(hipGetLastError());
    return PICSONG_OK;
}

// ---------------------------------------------------------------------------------------------
// DWT
// ---------------------------------------------------------------------------------------------
// want_c16: the context's choice of the 16-bit coefficient form (picsong_ctx::c16); *got_c16: what THIS call's plan
// delivers -- the form exists in the vector kernels only, and whether those apply also depends on the caller's pointers
// (16-byte alignment) and on PICSONG_DWT_NOVEC at call time, so a call may fall back to the 32-bit arrays and the
// coder must be told (plan_dwt_forward clears c16 on every level then)
static int dwt_forward_impl(picsong_ctx *c, const void *d_in, bool u8in, void *d_out, hipStream_t s, bool want_c16 = false,
                            bool *got_c16 = nullptr)
{
    const std::vector<FwdLaunch> plan = plan_dwt_forward(d_in, u8in, d_out, c->aw, c->ah, c->p.wl, c->p.qs, want_c16);
    if (got_c16) *got_c16 = plan_is_c16(plan);
    else if (want_c16 && !plan_is_c16(plan)) return fail(PICSONG_ERR_ARG, "the 16-bit coefficient form needs the vector kernels on every level");
    Fwd2Launch f2;
    const bool fused01 = plan_dwt_fwd2(plan, f2, true, c->p.lossy != 0);
    if (fused01) {                       // levels 0 and 1 in one launch, LL1 stays in registers
        launch_fwd2(c->p.lossy != 0, f2, s);
        HIP_TRY(hipGetLastError());
    }
    return launch_fwd_levels(c, plan, fused01 ? 2 : 0, s);
}

int picsong_dwt_forward(picsong_ctx *c, const void *d_in, void *d_out, void *stream)
{
    if (!c || !d_in || !d_out) return fail(PICSONG_ERR_ARG, "dwt_forward: null argument");
    return dwt_forward_impl(c, d_in, false, d_out, (hipStream_t)stream);
}

int picsong_dwt_forward_u8(picsong_ctx *c, const uint8_t *d_in, void *d_out, void *stream)
{
    if (!c || !d_in || !d_out) return fail(PICSONG_ERR_ARG, "dwt_forward_u8: null argument");
    return dwt_forward_impl(c, d_in, true, d_out, (hipStream_t)stream);
}

static void launch_fwd_any(picsong_ctx *c, const FwdLaunch &f, hipStream_t s)
{
    switch (f.band) {
    case 32: launch_fwd<32>(c, f, s); break;
    case 16: launch_fwd<16>(c, f, s); break;
    case 8: launch_fwd<8>(c, f, s); break;
    default: launch_fwd<4>(c, f, s); break;
    }
}

int picsong_dwt_forward_band(picsong_ctx *c, const uint8_t *d_frame, int row0, int rows, void *d_out, void *stream)
{
    if (!c || !d_frame || !d_out) return fail(PICSONG_ERR_ARG, "dwt_forward_band: null argument");
    if (row0 < 0 || rows <= 0 || (row0 & 1) || (rows & 1) || row0 + rows > c->ah)
        return fail(PICSONG_ERR_ARG, "dwt_forward_band: rows [%d, %d) must be even and inside [0, %d)", row0, row0 + rows, c->ah);
    std::vector<FwdLaunch> plan = plan_dwt_forward(d_frame, true, d_out, c->aw, c->ah, c->p.wl, c->p.qs);
    plan_restrict_band(plan[0], row0, rows);
    launch_fwd_any(c, plan[0], (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return PICSONG_OK;
}

int picsong_dwt_forward_tail(picsong_ctx *c, void *d_out, void *stream)
{
    if (!c || !d_out) return fail(PICSONG_ERR_ARG, "dwt_forward_tail: null argument");
    // (the level-0 source is irrelevant here: only the launches of levels >= 1 are used)
    const std::vector<FwdLaunch> plan = plan_dwt_forward(d_out, false, d_out, c->aw, c->ah, c->p.wl, c->p.qs);
    return launch_fwd_levels(c, plan, 1, (hipStream_t)stream);
}

// The frame paths' synthesis.  d_pixels != nullptr: the finest level writes clamped u8 pixels there (level shift +
// clamp fused, when its vector kernel applies) instead of T samples into d_out; *fused says so.
// frames > 1 (picsong_decode_frames): grid.z = frame; frame z's coded coefficients at d_in + z * P coefficients, its
// work buffer at d_out + z * (P + extra) elements, its pixels at d_pixels + z * pix_stride bytes.
// want_c16: the decoder may write 16-bit coefficients (picsong_ctx::c16_dec) -- the plan says whether this call's
// pointers allow it (plan_inv_is_c16), BEFORE the decoder is launched: inverse_plan, then the decoder, then run_inverse.
// (the grey frame paths take the 16-bit form only with the fused pixel store; planes_out: an RGB frame's components,
// whose finest level writes T samples for the inverse colour transform, take it too)
static std::vector<InvLaunch> inverse_plan(picsong_ctx *c, const int32_t *d_in, void *d_out, uint8_t *d_pixels, bool *fused,
                                           unsigned frames, size_t pix_stride, bool want_c16, bool planes_out = false)
{
    if (fused) *fused = false;
    const bool px = d_pixels && (((uintptr_t)d_pixels) & 3u) == 0 && (pix_stride & 3u) == 0;
    std::vector<InvLaunch> plan = plan_dwt_inverse(d_in, d_out, c->aw, c->ah, c->p.wl, c->p.qs, c->fast_div,
                                                   want_c16 && (px || planes_out));
    if (px && !plan.empty() && plan.back().vec) {
        plan.back().a.dst_u8 = d_pixels;
        plan.back().a.off = 1 << (c->p.bit_depth - 1);
        if (fused) *fused = true;
    }
    if (frames > 1) {
        const unsigned long long in_z = (unsigned long long)c->P * (plan_inv_is_c16(plan) ? 2ull : 4ull);
        const unsigned long long wrk_z = (unsigned long long)(c->P + c->extra) * 4ull;
        for (InvLaunch &f : plan) {
            f.a.mallat_z = in_z;
            f.a.ll_z = f.a.first ? in_z : wrk_z;            // the coarsest level's LL comes from the coded array
            f.a.dst_z = wrk_z;
            f.a.u8_z = (unsigned long long)pix_stride;
        }
    }
    return plan;
}

static int run_inverse(picsong_ctx *c, const std::vector<InvLaunch> &plan, hipStream_t s, unsigned frames = 1)
{
    // 16-bit coefficients in, pixels out: synthesis levels 1 and 0 as one launch (dwt_inv2_kernel), LL0 in registers
    Inv2Launch f2;
    const bool fused10 = plan_dwt_inv2(plan, f2, c->p.lossy != 0);
    const size_t n = fused10 ? plan.size() - 2 : plan.size();
    for (size_t l = 0; l < n; l++) {
        const InvLaunch &f = plan[l];
        switch (f.band) {
        case 32: launch_inv<32>(c, f, s, frames); break;
        case 16: launch_inv<16>(c, f, s, frames); break;
        case 8: launch_inv<8>(c, f, s, frames); break;
        default: launch_inv<4>(c, f, s, frames); break;
        }
        HIP_TRY(hipGetLastError());
    }
    if (fused10) {
        const dim3 grid(f2.gx, f2.gy, frames);
        if (!c->p.lossy) dwt_inv2_kernel<false, false><<<grid, 256, 0, s>>>(f2.a);
        else if (f2.a.l0.one_div) dwt_inv2_kernel<true, true><<<grid, 256, 0, s>>>(f2.a);
        else dwt_inv2_kernel<true, false><<<grid, 256, 0, s>>>(f2.a);
        HIP_TRY(hipGetLastError());
    }
    return PICSONG_OK;
}

static int dwt_inverse_impl(picsong_ctx *c, const int32_t *d_in, void *d_out, uint8_t *d_pixels, bool *fused,
                            hipStream_t s, unsigned frames = 1, size_t pix_stride = 0)
{
    return run_inverse(c, inverse_plan(c, d_in, d_out, d_pixels, fused, frames, pix_stride, false), s, frames);
}

int picsong_dwt_inverse(picsong_ctx *c, const int32_t *d_in, void *d_out, void *stream)
{
    if (!c || !d_in || !d_out) return fail(PICSONG_ERR_ARG, "dwt_inverse: null argument");
    hipStream_t s = (hipStream_t)stream;
    for (const InvLaunch &f : plan_dwt_inverse(d_in, d_out, c->aw, c->ah, c->p.wl, c->p.qs, c->fast_div)) {
        switch (f.band) {
        case 32: launch_inv<32>(c, f, s); break;
        case 16: launch_inv<16>(c, f, s); break;
        case 8: launch_inv<8>(c, f, s); break;
        default: launch_inv<4>(c, f, s); break;
        }
        HIP_TRY(hipGetLastError());
    }
    return PICSONG_OK;
}

// ---------------------------------------------------------------------------------------------
// BPC
// ---------------------------------------------------------------------------------------------
static int bpc_args(picsong_ctx *c, BpcArgs &a, int comp = 0)
{
    if (comp < 0 || comp > 2) return fail(PICSONG_ERR_ARG, "component %d outside 0..2", comp);
    if (!c->has_lut[comp]) return fail(PICSONG_ERR_ARG, "no LUT loaded for component %d: call picsong_ctx_set_lut first", comp);
    memset(&a, 0, sizeof a);
    const picsong_lut_info &li = c->li[comp];
    a.AW = c->aw; a.AH = c->ah; a.wl = c->p.wl; a.nCB = c->ncb; a.ncx = c->aw / PICSONG_CB;
    a.lut = c->d_lut[comp];
    a.g.nBp = li.n_bitplanes; a.g.nSub = li.n_subbands; a.g.cRef = li.ctx_ref;
    a.g.cSign = li.ctx_sign; a.g.cSig = li.ctx_sig; a.g.prec = li.precision;
    a.g.nRef = li.n_ref; a.g.nSig = li.n_sig; a.g.nSign = li.n_sign;
    a.range_flag = c->d_flag;
    a.k = c->p.k; a.n_tables = li.n_tables > 0 ? li.n_tables : 1;
    return PICSONG_OK;
}

// -k > 0: may the launch take the kernels' COMPACT table copies (bulk_max_span_bytes: the geometry's widest codeblock)?
// PICSONG_BULK_FULLTAB=1 keeps the whole-table instantiations (the tests cross-check both)
static bool bulk_compact(picsong_ctx *c, int comp)
{
    if (const char *e = getenv("PICSONG_BULK_FULLTAB")) if (atoi(e) != 0) return false;
    if (c->bulk_compact[comp] < 0) {
        const picsong_lut_info &li = c->li[comp];
        c->bulk_compact[comp] = bulk_max_span_bytes(c->aw, c->ah, c->p.wl, li.n_bitplanes, li.n_subbands, li.ctx_ref, li.ctx_sig,
                                                    li.ctx_sign) <= kBulkCompactBytes ? 1 : 0;
    }
    return c->bulk_compact[comp] == 1;
}

// the coders' bit-plane scratch: kEncScratchDwordsPerWave per wave of a frame's launch (whole workgroups), allocated
// at the first use
static int ensure_plane_scratch(picsong_ctx *c)
{
    static_assert(kBpcEncWgWaves == kBpcDecWgWaves, "one scratch serves the launches of both directions");
    static_assert(kBpcEncWgWaves % kBpc3WgWaves == 0, "-cp 3 launches fit the same allocation");
    if (c->d_plane_scratch) return PICSONG_OK;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMalloc(&c->d_plane_scratch, (size_t)(((c->ncb + 1) / 2 + kBpcEncWgWaves - 1) / kBpcEncWgWaves * kBpcEncWgWaves) * kEncScratchDwordsPerWave * sizeof(uint32_t)));
    return PICSONG_OK;
}

// d_stage16: the encoders' 16-bit staging, uint16[nCB * 4096] (BpcArgs::staging16) -- the context's own staging
// buffers hold it on the frame paths; picsong_bpc_encode widens it into the caller's int32 array
static int bpc_encode_impl(picsong_ctx *c, const void *d_coeffs, uint16_t *d_stage16, int32_t *d_sizes,
                           hipStream_t s, int cb_begin = 0, int cb_count = -1, int comp = 0, bool c16 = false)
{
    BpcArgs a;
    int rc = bpc_args(c, a, comp);
    if (rc) return rc;
    a.c16 = c16 ? 1 : 0;
    if (cb_count < 0) cb_count = c->ncb - cb_begin;
    a.cb_base = cb_begin;
    a.nCB = cb_begin + cb_count;
    a.coeffs_in = d_coeffs; a.is_float = c->p.lossy ? 1 : 0;
    a.staging16 = d_stage16; a.sizes = d_sizes;
    if (int rc2 = ensure_plane_scratch(c)) return rc2;
    a.plane_scratch = c->d_plane_scratch;
    if (c->p.cp == 3) {        // three coding passes: one kernel for both directions (bpc3_kernel)
        bpc3_kernel<false><<<(unsigned)(((cb_count + 1) / 2 + kBpc3WgWaves - 1) / kBpc3WgWaves), 64 * kBpc3WgWaves, 0, s>>>(a);
        HIP_TRY(hipGetLastError());
        return PICSONG_OK;
    }
    // -k > 0: the BULK instantiation (bulk scan below the consecutive bit-planes, table s in LDS)
    if (a.k > 0.0f) {
        // Two instantiations of the -k > 0 encoder: with compact table copies it is asked for six waves a SIMD (80
        // registers, some of its prologue spilled) -- what frames in flight want: 133 -> 142 Gpixel/s at k = 0.5; with
        // whole tables its LDS bounds it to four waves anyway, it takes 102 registers and spills nothing -- what a lone
        // frame wants, whose 4080 waves are four to a SIMD whatever the kernel allows: 0.376 against 0.411 ms.  The
        // context's hint (picsong_ctx_set_pipelined) chooses.
        if (bulk_compact(c, comp) && c->pipelined) bpc_encode_kernel<true, true><<<(unsigned)((cb_count + 1) / 2), 64, 0, s>>>(a);
        else bpc_encode_kernel<true><<<(unsigned)((cb_count + 1) / 2), 64, 0, s>>>(a);
    }
    else bpc_encode_kernel<false><<<(unsigned)(((cb_count + 1) / 2 + kBpcEncWgWaves - 1) / kBpcEncWgWaves), 64 * kBpcEncWgWaves, 0, s>>>(a);
    HIP_TRY(hipGetLastError());
    return PICSONG_OK;
}

// The stage-level call keeps the reference's contract -- an int32 array of 4096 words a codeblock, 0xFFFFFFFF wherever
// nothing was written (BPCEngine::deviceMemoryAllocator BPCEngine.cu:2429-2441) -- on top of the encoders' 16-bit
// staging: the coder writes the context's own buffer, widen_staging_kernel copies words 0 .. len - 1 of every codeblock.
static int bpc_encode_widened(picsong_ctx *c, const void *d_coeffs, int32_t *d_staging, int32_t *d_sizes, hipStream_t s, int comp)
{
    HIP_TRY(hipSetDevice(c->device));
    if (!c->d_staging) HIP_TRY(hipMalloc(&c->d_staging, c->P * sizeof(int32_t)));
    if (d_staging == c->d_staging) return fail(PICSONG_ERR_ARG, "bpc_encode: the context's own staging passed as the output");
    uint16_t *const st16 = reinterpret_cast<uint16_t *>(c->d_staging);
    HIP_TRY(hipMemsetAsync(d_staging, 0xFF, c->P * sizeof(int32_t), s));
    int rc = bpc_encode_impl(c, d_coeffs, st16, d_sizes, s, 0, -1, comp);
    if (rc) return rc;
    widen_staging_kernel<<<(unsigned)c->ncb, 256, 0, s>>>(st16, d_sizes, 0, d_staging);
    HIP_TRY(hipGetLastError());
    return PICSONG_OK;
}

int picsong_bpc_encode(picsong_ctx *c, const void *d_coeffs, int32_t *d_staging, int32_t *d_sizes, void *stream)
{
    if (!c || !d_coeffs || !d_staging || !d_sizes) return fail(PICSONG_ERR_ARG, "bpc_encode: null argument");
    return bpc_encode_widened(c, d_coeffs, d_staging, d_sizes, (hipStream_t)stream, 0);
}

// PICSONG_DEC_STAGING=1: the frame paths unpack into the 32-bit staging first, as picsong_bitstream_unpack +
// picsong_bpc_decode do (A/B of the decoder that reads the stream itself)
static bool dec_from_stream(const picsong_ctx *c)
{
    static const bool staged = [] { const char *e = getenv("PICSONG_DEC_STAGING"); return e && atoi(e) != 0; }();
    return !staged && c->p.cp != 3;
}

// d_stream16 != nullptr (k = 0, -cp 2): the codewords come from the packed stream, d_offsets the scan of its lengths
// (scan_stream_kernel); d_staging is then not read
// c16 (with d_stream16): the coefficients leave as an int16 Mallat array at d_coeffs (bpc_decode_kernel's C16 form)
static int bpc_decode_impl(picsong_ctx *c, const int32_t *d_staging, const int32_t *d_sizes, int32_t *d_coeffs,
                           hipStream_t s, int comp = 0, const uint16_t *d_stream16 = nullptr,
                           const int32_t *d_offsets = nullptr, bool c16 = false)
{
    BpcArgs a;
    int rc = bpc_args(c, a, comp);
    if (rc) return rc;
    a.coeffs_out = d_coeffs;
    a.staging = const_cast<int32_t *>(d_staging);
    a.sizes = const_cast<int32_t *>(d_sizes);
    if (int rc2 = ensure_plane_scratch(c)) return rc2;      // the decoder parks its finished planes there too
    a.plane_scratch = c->d_plane_scratch;
    const unsigned waves = (unsigned)((c->ncb + 1) / 2);
    if (c->p.cp == 3) {
        bpc3_kernel<true><<<(waves + kBpc3WgWaves - 1) / kBpc3WgWaves, 64 * kBpc3WgWaves, 0, s>>>(a);
        HIP_TRY(hipGetLastError());
        return PICSONG_OK;
    }
    if (a.k > 0.0f) {
        // (one launch: the two-pass planes are parked in the scratch whatever their number)
        const bool cmp = bulk_compact(c, comp);
        if (d_stream16) {
            a.cw16 = d_stream16; a.cw16_offsets = d_offsets; a.cw16_total = c->d_total;
            a.cw16_max = (uint32_t)picsong_max_stream_shorts(c->aw, c->ah);
            if (c16) {
                if (cmp) bpc_decode_kernel<true, kDecSmallPlanes, true, true, true><<<waves, 64, 0, s>>>(a);
                else bpc_decode_kernel<true, kDecSmallPlanes, true, true><<<waves, 64, 0, s>>>(a);
            } else {
                if (cmp) bpc_decode_kernel<true, kDecSmallPlanes, true, false, true><<<waves, 64, 0, s>>>(a);
                else bpc_decode_kernel<true, kDecSmallPlanes, true><<<waves, 64, 0, s>>>(a);
            }
        } else {
            if (c16) return fail(PICSONG_ERR_ARG, "the 16-bit coefficient form decodes from the stream itself");
            if (cmp) bpc_decode_kernel<true, kDecSmallPlanes, false, false, true><<<waves, 64, 0, s>>>(a);
            else bpc_decode_kernel<true, kDecSmallPlanes><<<waves, 64, 0, s>>>(a);
        }
    } else {
        const unsigned wgs = (waves + kBpcDecWgWaves - 1) / kBpcDecWgWaves;
        if (d_stream16) {
            a.cw16 = d_stream16; a.cw16_offsets = d_offsets; a.cw16_total = c->d_total;
            a.cw16_max = (uint32_t)picsong_max_stream_shorts(c->aw, c->ah);
            if (c16) bpc_decode_kernel<false, kDecSmallPlanes, true, true><<<wgs, 64 * kBpcDecWgWaves, 0, s>>>(a);
            else bpc_decode_kernel<false, kDecSmallPlanes, true><<<wgs, 64 * kBpcDecWgWaves, 0, s>>>(a);
        } else {
            if (c16) return fail(PICSONG_ERR_ARG, "the 16-bit coefficient form decodes from the stream itself");
            bpc_decode_kernel<false, kDecSmallPlanes><<<wgs, 64 * kBpcDecWgWaves, 0, s>>>(a);
        }
    }
    HIP_TRY(hipGetLastError());
    return PICSONG_OK;
}

// a frame path's decoder: lengths and offsets out of the stream (one launch), then the coder reading the stream itself;
// or the unpack into the staging (-k > 0, -cp 3, PICSONG_DEC_STAGING)
static int unpack_impl(picsong_ctx *c, const uint16_t *d_stream, int32_t *d_staging, int32_t *d_sizes,
                       bool memset_staging, hipStream_t s);
static int decode_stream_impl(picsong_ctx *c, const uint16_t *d_stream, int32_t *d_coeffs, hipStream_t s, int comp,
                              bool c16 = false)
{
    int rc;
    if (!dec_from_stream(c)) {
        if (c16) return fail(PICSONG_ERR_ARG, "the 16-bit coefficient form decodes from the stream itself");
        if ((rc = unpack_impl(c, d_stream, c->d_staging, c->d_sizes, false, s))) return rc;
        return bpc_decode_impl(c, c->d_staging, c->d_sizes, d_coeffs, s, comp);
    }
    scan_stream_kernel<<<1, scan_threads(c->ncb), 0, s>>>(d_stream, c->ncb, c->d_sizes, c->d_offsets, c->d_total, c->d_flag, 0);
    HIP_TRY(hipGetLastError());
    return bpc_decode_impl(c, nullptr, c->d_sizes, d_coeffs, s, comp, d_stream, c->d_offsets, c16);
}

int picsong_bpc_decode(picsong_ctx *c, const int32_t *d_staging, const int32_t *d_sizes, int32_t *d_coeffs,
                       void *stream)
{
    if (!c || !d_coeffs || !d_staging || !d_sizes) return fail(PICSONG_ERR_ARG, "bpc_decode: null argument");
    return bpc_decode_impl(c, d_staging, d_sizes, d_coeffs, (hipStream_t)stream);
}

int picsong_bpc_encode_component(picsong_ctx *c, int comp, const void *d_coeffs, int32_t *d_staging, int32_t *d_sizes,
                                 void *stream)
{
    if (!c || !d_coeffs || !d_staging || !d_sizes) return fail(PICSONG_ERR_ARG, "bpc_encode: null argument");
    return bpc_encode_widened(c, d_coeffs, d_staging, d_sizes, (hipStream_t)stream, comp);
}

int picsong_bpc_decode_component(picsong_ctx *c, int comp, const int32_t *d_staging, const int32_t *d_sizes,
                                 int32_t *d_coeffs, void *stream)
{
    if (!c || !d_coeffs || !d_staging || !d_sizes) return fail(PICSONG_ERR_ARG, "bpc_decode: null argument");
    return bpc_decode_impl(c, d_staging, d_sizes, d_coeffs, (hipStream_t)stream, comp);
}

int picsong_selftest_lds_order(int device, int *mismatches)
{
    if (!mismatches) return fail(PICSONG_ERR_ARG, "selftest: null argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(PICSONG_ERR_NODEVICE, "no HIP device: this library has no CPU path");
    if (device < 0 || device >= ndev) return fail(PICSONG_ERR_ARG, "device %d of %d", device, ndev);
    HIP_TRY(hipSetDevice(device));
    uint32_t *d = nullptr, h = 0;
    HIP_TRY(hipMalloc(&d, sizeof(uint32_t)));
    HIP_TRY(hipMemset(d, 0, sizeof(uint32_t)));
    lds_order_selftest_kernel<<<2048, 256>>>(2000, 0x5EED1234u, d);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(&h, d, sizeof h, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(PICSONG_ERR_HIP, "selftest: %s", hipGetErrorString(e));
    *mismatches = (int)h;
    return PICSONG_OK;
}

int picsong_range_flag(picsong_ctx *c, void *stream, int *h_flag)
{
    if (!c || !h_flag) return fail(PICSONG_ERR_ARG, "range_flag: null argument");
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipMemcpyAsync(&c->h_pinned[1], c->d_flag, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemsetAsync(c->d_flag, 0, sizeof(int), s));      // read and clear: the next query covers later calls only
    HIP_TRY(hipStreamSynchronize(s));
    *h_flag = c->h_pinned[1];
    return PICSONG_OK;
}

// ---------------------------------------------------------------------------------------------
// BitStreamBuilder
// ---------------------------------------------------------------------------------------------
int picsong_last_total(picsong_ctx *c, void *stream, int *h_total)
{
    if (!c || !h_total) return fail(PICSONG_ERR_ARG, "last_total: null argument");
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipMemcpyAsync(&c->h_pinned[0], c->d_total, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    *h_total = c->h_pinned[0];
    return PICSONG_OK;
}

// W: uint16_t = the encoders' own staging (the frame paths), int32_t = a caller's array (picsong_bitstream_pack)
extern "C++" {
template <typename W>
static int pack_range(picsong_ctx *c, const W *d_staging, const int32_t *d_sizes, int n,
                      const uint16_t *h_header, uint16_t *d_stream, hipStream_t s)
{
    HeaderArg h;
    memset(&h, 0, sizeof h);
    if (h_header) { memcpy(h.h, h_header, sizeof h.h); h.has = 1; }
    c->last_batch = 0;                                      // the most recent total is d_total (picsong_copy_last_totals)
    scan_sizes_kernel<<<1, scan_threads(n), 0, s>>>(d_sizes, n, c->d_offsets, c->d_total);
    HIP_TRY(hipGetLastError());
    pack_kernel<W><<<(unsigned)n, 256, 0, s>>>(d_staging, d_sizes, c->d_offsets, c->d_total, n, h, d_stream);
    HIP_TRY(hipGetLastError());
    return PICSONG_OK;
}
}  // extern "C++"

int picsong_bitstream_pack(picsong_ctx *c, const int32_t *d_staging, const int32_t *d_sizes,
                           const uint16_t *h_header, uint16_t *d_stream, int *h_total, void *stream)
{
    if (!c || !d_staging || !d_sizes || !d_stream) return fail(PICSONG_ERR_ARG, "bitstream_pack: null argument");
    int rc = pack_range(c, d_staging, d_sizes, c->ncb, h_header, d_stream, (hipStream_t)stream);
    if (rc) return rc;
    if (h_total) return picsong_last_total(c, stream, h_total);
    return PICSONG_OK;
}

static int unpack_impl(picsong_ctx *c, const uint16_t *d_stream, int32_t *d_staging, int32_t *d_sizes,
                       bool memset_staging, hipStream_t s)
{
    // BSEngine::deviceMemoryAllocator BitStreamBuilder.cu:281-284.  Slots beyond a codeblock's length
    // are never read by the decoder, so the frame path skips this 4*AW*AH-byte fill.
    if (memset_staging) HIP_TRY(hipMemsetAsync(d_staging, 0xFF, c->P * sizeof(int32_t), s));
    read_sizes_kernel<<<(unsigned)((c->ncb + 255) / 256), 256, 0, s>>>(d_stream, c->ncb, d_sizes, c->d_flag);
    HIP_TRY(hipGetLastError());
    scan_sizes_kernel<<<1, scan_threads(c->ncb), 0, s>>>(d_sizes, c->ncb, c->d_offsets, c->d_total);
    HIP_TRY(hipGetLastError());
    unpack_kernel<<<(unsigned)c->ncb, 256, 0, s>>>(d_stream, d_sizes, c->d_offsets, c->ncb, d_staging);
    HIP_TRY(hipGetLastError());
    return PICSONG_OK;
}

int picsong_bitstream_unpack(picsong_ctx *c, const uint16_t *d_stream, int32_t *d_staging, int32_t *d_sizes,
                             void *stream)
{
    if (!c || !d_staging || !d_sizes || !d_stream) return fail(PICSONG_ERR_ARG, "bitstream_unpack: null argument");
    return unpack_impl(c, d_stream, d_staging, d_sizes, true, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------
// whole frame
// ---------------------------------------------------------------------------------------------
static int ensure_workspace(picsong_ctx *c, bool decode)
{
    HIP_TRY(hipSetDevice(c->device));
    if (!c->d_coef) HIP_TRY(hipMalloc(&c->d_coef, (c->P + c->extra) * 4));
    if (!c->d_staging) HIP_TRY(hipMalloc(&c->d_staging, c->P * sizeof(int32_t)));
    if (!c->d_sizes) HIP_TRY(hipMalloc(&c->d_sizes, (size_t)c->ncb * sizeof(int32_t)));
    if (decode && !c->d_coef_i) HIP_TRY(hipMalloc(&c->d_coef_i, c->P * sizeof(int32_t)));
    return PICSONG_OK;
}

int picsong_profile_begin(picsong_ctx *c, int capacity)
{
    if (!c || capacity < 0) return fail(PICSONG_ERR_ARG, "profile_begin: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    if (!c->prof_ev) c->prof_ev = new std::vector<hipEvent_t>();
    while ((int)c->prof_ev->size() < 4 * capacity) {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        c->prof_ev->push_back(e);
    }
    c->prof_cap = capacity;
    c->prof_n = 0;
    return PICSONG_OK;
}

int picsong_profile_read(picsong_ctx *c, int *n_frames, float *ms, int cap)
{
    if (!c || !n_frames || !ms) return fail(PICSONG_ERR_ARG, "profile_read: null argument");
    const int n = c->prof_n < cap ? c->prof_n : cap;
    for (int f = 0; f < n; f++) {
        hipEvent_t *e = c->prof_ev->data() + 4 * f;
        HIP_TRY(hipEventSynchronize(e[3]));
        for (int k = 0; k < 3; k++) HIP_TRY(hipEventElapsedTime(&ms[3 * f + k], e[k], e[k + 1]));
    }
    *n_frames = n;
    return PICSONG_OK;
}

int picsong_encode_frame(picsong_ctx *c, const uint8_t *d_frame, int iter, uint16_t *d_stream, void *stream)
{
    if (!c || !d_frame || !d_stream) return fail(PICSONG_ERR_ARG, "encode_frame: null argument");
    int rc = ensure_workspace(c, false);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    hipEvent_t *ev = nullptr;
    if (c->prof_cap > 0 && c->prof_n < c->prof_cap) ev = c->prof_ev->data() + 4 * (c->prof_n++);
    if (ev) HIP_TRY(hipEventRecord(ev[0], s));
    // (coefficients between the transform and the coder as int16 where their magnitudes are bounded: c->c16)
    // (Round 3 tried, for a lone frame, the coder's launch for the codeblock rows below AH/4 -- final once the fused head has
    // run -- at once on the caller's stream, and the transform's small levels + the coder's launch for the top rows on a
    // second stream beside it: byte-identical, and 70 us SLOWER at 8K (0.308 -> 0.379 ms): the top-left codeblocks are the
    // ones with the most planes, the launch's critical waves, and the split starts exactly those 20-40 us late.)
    // (an unaligned frame pointer, or PICSONG_DWT_NOVEC set after the context was created, takes the per-column kernels
    // and with them the 32-bit arrays: `c16` is what this call's plan delivers, as in picsong_encode_frames)
    bool c16 = false;
    if ((rc = dwt_forward_impl(c, d_frame, true, c->d_coef, s, c->c16, &c16))) return rc;
    if (ev) HIP_TRY(hipEventRecord(ev[1], s));
    uint16_t *const st16 = reinterpret_cast<uint16_t *>(c->d_staging);      // (the context's staging holds the 16-bit form)
    if ((rc = bpc_encode_impl(c, c->d_coef, st16, c->d_sizes, s, 0, -1, 0, c16))) return rc;
    if (ev) HIP_TRY(hipEventRecord(ev[2], s));
    uint16_t hdr[PICSONG_HDR_SHORTS];
    if (iter == 0) picsong_header_pack(&c->p, hdr);
    rc = pack_range(c, st16, c->d_sizes, c->ncb, iter == 0 ? hdr : nullptr, d_stream, s);
    if (ev) HIP_TRY(hipEventRecord(ev[3], s));
    c->last_batch = 0;                                      // the most recent call's total is d_total
    return rc;
}

int picsong_decode_frame(picsong_ctx *c, const uint16_t *d_stream, uint8_t *d_frame_out, void *stream)
{
    if (!c || !d_frame_out || !d_stream) return fail(PICSONG_ERR_ARG, "decode_frame: null argument");
    int rc = ensure_workspace(c, true);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    // (16-bit coefficients between the decoder and the synthesis where the context's magnitudes are bounded and this
    // call's pointers take the vector kernels: c->c16_dec, plan_inv_is_c16)
    bool fused = false;
    const std::vector<InvLaunch> plan = inverse_plan(c, c->d_coef_i, c->d_coef, d_frame_out, &fused, 1, 0,
                                                     c->c16_dec && dec_from_stream(c));
    if ((rc = decode_stream_impl(c, d_stream, c->d_coef_i, s, 0, plan_inv_is_c16(plan)))) return rc;
    if ((rc = run_inverse(c, plan, s))) return rc;
    if (fused) return PICSONG_OK;            // the finest level wrote the pixels itself
    const void *img = (const char *)c->d_coef + c->extra * 4;
    const size_t n4 = c->P / 4;
    const int off = 1 << (c->p.bit_depth - 1);
    const int grid = (int)((n4 + 255) / 256 > 8192 ? 8192 : (n4 + 255) / 256);
    if (c->p.lossy) clamp_to_u8_f32_kernel<<<grid, 256, 0, s>>>((const float *)img, d_frame_out, n4, (float)off);
    else clamp_to_u8_i32_kernel<<<grid, 256, 0, s>>>((const int32_t *)img, d_frame_out, n4, off);
    HIP_TRY(hipGetLastError());
    return PICSONG_OK;
}

int picsong_encode_stripe_coded(picsong_ctx *c, const void *d_coeffs, int cb_begin, int cb_count, uint16_t *d_stream,
                                void *stream)
{
    if (!c || !d_coeffs || !d_stream) return fail(PICSONG_ERR_ARG, "encode_stripe_coded: null argument");
    if (cb_begin < 0 || cb_count <= 0 || cb_begin + cb_count > c->ncb)
        return fail(PICSONG_ERR_ARG, "encode_stripe_coded: codeblocks [%d, %d) outside [0, %d)", cb_begin,
                    cb_begin + cb_count, c->ncb);
    int rc = ensure_workspace(c, false);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    uint16_t *const st16 = reinterpret_cast<uint16_t *>(c->d_staging);
    if ((rc = bpc_encode_impl(c, d_coeffs, st16, c->d_sizes, s, cb_begin, cb_count))) return rc;
    return pack_range(c, st16 + (size_t)cb_begin * PICSONG_CB_WORDS, c->d_sizes + cb_begin, cb_count, nullptr,
                      d_stream, s);
}

int picsong_encode_frame_stripe(picsong_ctx *c, const uint8_t *d_frame, int cb_begin, int cb_count,
                                uint16_t *d_stream, void *stream)
{
    if (!c || !d_frame || !d_stream) return fail(PICSONG_ERR_ARG, "encode_frame_stripe: null argument");
    if (cb_begin < 0 || cb_count <= 0 || cb_begin + cb_count > c->ncb)
        return fail(PICSONG_ERR_ARG, "encode_frame_stripe: codeblocks [%d, %d) outside [0, %d)", cb_begin,
                    cb_begin + cb_count, c->ncb);
    int rc = ensure_workspace(c, false);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    if ((rc = dwt_forward_impl(c, d_frame, true, c->d_coef, s))) return rc;
    uint16_t *const st16 = reinterpret_cast<uint16_t *>(c->d_staging);
    if ((rc = bpc_encode_impl(c, c->d_coef, st16, c->d_sizes, s, cb_begin, cb_count))) return rc;
    return pack_range(c, st16 + (size_t)cb_begin * PICSONG_CB_WORDS, c->d_sizes + cb_begin, cb_count, nullptr,
                      d_stream, s);
}

// ---------------------------------------------------------------------------------------------
// batched frames: n frames of a video in ONE launch per stage (grid.z = frame for the DWT levels, n x the
// codeblock waves for the coder, grid.y = frame for the pack).  A 4K frame alone is 1020 coder waves -- one
// per SIMD, a quarter of what the coder needs to keep the vector pipes busy ("batch several
// frames per launch"); the reference's answer is -numberOfStreams worker threads with a stream each
// (Engines/CodingEngine.cu:990-1061), this is the same frames-in-flight idea without depending on the
// runtime's queues.
// ---------------------------------------------------------------------------------------------
static void free_batch(picsong_ctx *c)
{
    if (c->b_coef) (void)hipFree(c->b_coef);
    if (c->b_staging) (void)hipFree(c->b_staging);
    if (c->b_sizes) (void)hipFree(c->b_sizes);
    if (c->b_offsets) (void)hipFree(c->b_offsets);
    if (c->b_total) (void)hipFree(c->b_total);
    if (c->b_plane_scratch) (void)hipFree(c->b_plane_scratch);
    if (c->b_coef_i) (void)hipFree(c->b_coef_i);
    c->b_coef_i = nullptr; c->b_coef_i_cap = 0;
    if (c->h_totals) (void)hipHostFree(c->h_totals);
    c->b_coef = nullptr; c->b_staging = c->b_sizes = c->b_offsets = c->b_total = nullptr;
    c->b_plane_scratch = nullptr; c->h_totals = nullptr; c->batch_cap = 0;
}

static int ensure_batch(picsong_ctx *c, int n)
{
    if (n <= c->batch_cap) return PICSONG_OK;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipDeviceSynchronize());                 // a smaller batch may still be running on the old buffers
    free_batch(c);
    // (scratch for whole workgroups per frame: the RGB form pads every component's waves to workgroups)
    const size_t waves = (size_t)n * (size_t)(((c->ncb + 1) / 2 + kBpcEncWgWaves - 1) / kBpcEncWgWaves * kBpcEncWgWaves);
    HIP_TRY(hipMalloc(&c->b_coef, (size_t)n * (c->P + c->extra) * 4));
    HIP_TRY(hipMalloc(&c->b_staging, (size_t)n * c->P * sizeof(int32_t)));
    HIP_TRY(hipMalloc(&c->b_sizes, (size_t)n * (size_t)c->ncb * sizeof(int32_t)));
    HIP_TRY(hipMalloc(&c->b_offsets, (size_t)n * (size_t)c->ncb * sizeof(int32_t)));
    HIP_TRY(hipMalloc(&c->b_total, (size_t)n * sizeof(int32_t)));
    HIP_TRY(hipMalloc(&c->b_plane_scratch, waves * kEncScratchDwordsPerWave * sizeof(uint32_t)));
    HIP_TRY(hipHostMalloc(&c->h_totals, (size_t)n * sizeof(int32_t)));
    c->batch_cap = n;
    return PICSONG_OK;
}

// n planes of P 32-bit words: the decoded coefficients of a batch / the colour-transformed components of an RGB frame
static int ensure_coef_i(picsong_ctx *c, int n)
{
    if (c->b_coef_i_cap >= n) return PICSONG_OK;
    HIP_TRY(hipDeviceSynchronize());
    if (c->b_coef_i) (void)hipFree(c->b_coef_i);
    c->b_coef_i = nullptr; c->b_coef_i_cap = 0;
    HIP_TRY(hipMalloc(&c->b_coef_i, (size_t)n * c->P * sizeof(int32_t)));
    c->b_coef_i_cap = n;
    return PICSONG_OK;
}

// -k > 0 over the frames (or the three components) of a batched call: the BULK coder instantiations, one-wave
// workgroups -- a frame is exactly its codeblock pairs, no padding waves --, chosen as bpc_encode_impl / bpc_decode_impl
// choose them for one frame (`cmp`: every table the launch uses takes the compact LDS copies)
static void launch_bulk_encode_frames(picsong_ctx *c, BpcArgs &a, unsigned frames, bool cmp, hipStream_t s)
{
    a.waves_per_frame = (c->ncb + 1) / 2;
    const unsigned wgs = frames * (unsigned)a.waves_per_frame;
    if (cmp && c->pipelined) bpc_encode_kernel<true, true><<<wgs, 64, 0, s>>>(a);
    else bpc_encode_kernel<true><<<wgs, 64, 0, s>>>(a);
}
static void launch_bulk_decode_frames(picsong_ctx *c, BpcArgs &a, unsigned frames, bool cmp, bool direct, bool c16, hipStream_t s)
{
    a.waves_per_frame = (c->ncb + 1) / 2;
    const unsigned wgs = frames * (unsigned)a.waves_per_frame;
    if (direct && c16) {
        if (cmp) bpc_decode_kernel<true, kDecSmallPlanes, true, true, true><<<wgs, 64, 0, s>>>(a);
        else bpc_decode_kernel<true, kDecSmallPlanes, true, true><<<wgs, 64, 0, s>>>(a);
    } else if (direct) {
        if (cmp) bpc_decode_kernel<true, kDecSmallPlanes, true, false, true><<<wgs, 64, 0, s>>>(a);
        else bpc_decode_kernel<true, kDecSmallPlanes, true><<<wgs, 64, 0, s>>>(a);
    } else {
        if (cmp) bpc_decode_kernel<true, kDecSmallPlanes, false, false, true><<<wgs, 64, 0, s>>>(a);
        else bpc_decode_kernel<true, kDecSmallPlanes><<<wgs, 64, 0, s>>>(a);
    }
}

int picsong_encode_frames(picsong_ctx *c, int n, const uint8_t *d_frames, size_t frame_stride, int first_iter,
                          uint16_t *d_streams, size_t stream_stride, void *stream)
{
    if (!c || !d_frames || !d_streams) return fail(PICSONG_ERR_ARG, "encode_frames: null argument");
    if (n < 1 || n > 64) return fail(PICSONG_ERR_ARG, "encode_frames: %d frames outside 1..64", n);
    if (n > 1 && (frame_stride < c->P || stream_stride < picsong_max_stream_shorts(c->aw, c->ah)))
        return fail(PICSONG_ERR_ARG, "encode_frames: strides smaller than a padded frame / a worst-case codestream");
    if (c->p.cp == 3 || c->p.is_rgb)
        return fail(PICSONG_ERR_ARG, "encode_frames: grey -cp 2 contexts only (-cp 3 is coded frame by frame: picsong_encode_frame; "
                                     "an RGB frame's components: picsong_encode_rgb_frame)");
    if (((uintptr_t)d_frames | frame_stride) & 15u) return fail(PICSONG_ERR_ARG, "encode_frames: frames must be 16-byte aligned");
    HIP_TRY(hipSetDevice(c->device));                       // (a caller with several devices may be on another one)
    BpcArgs a;
    int rc = bpc_args(c, a, 0);
    if (rc) return rc;
    if ((rc = ensure_batch(c, n))) return rc;
    hipStream_t s = (hipStream_t)stream;
    const size_t coef_z = (c->P + c->extra) * 4;
    hipEvent_t *ev = nullptr;                             // stage timers: one set per batch (picsong_profile_begin)
    if (c->prof_cap > 0 && c->prof_n < c->prof_cap) ev = c->prof_ev->data() + 4 * (c->prof_n++);
    if (ev) HIP_TRY(hipEventRecord(ev[0], s));

    // ---- DWT: the single-frame plan of frame 0 with grid.z = n
    std::vector<FwdLaunch> plan = plan_dwt_forward(d_frames, true, c->b_coef, c->aw, c->ah, c->p.wl, c->p.qs, c->c16);
    a.c16 = plan_is_c16(plan) ? 1 : 0;
    for (size_t l = 0; l < plan.size(); l++) {
        plan[l].a.src_z = l == 0 ? (unsigned long long)frame_stride : (unsigned long long)coef_z;
        plan[l].a.dst_z = (unsigned long long)coef_z;
    }
    Fwd2Launch f2;
    const bool fused01 = plan_dwt_fwd2(plan, f2, true, c->p.lossy != 0);
    if (fused01) {
        launch_fwd2(c->p.lossy != 0, f2, s, (unsigned)n);
        HIP_TRY(hipGetLastError());
    }
    if (int rc = launch_fwd_levels(c, plan, fused01 ? 2 : 0, s, (unsigned)n)) return rc;

    if (ev) HIP_TRY(hipEventRecord(ev[1], s));
    // ---- coder: one grid over the n frames' codeblock pairs
    const int wpf = (c->ncb + 1) / 2;
    a.cb_base = 0; a.nCB = c->ncb;
    a.coeffs_in = c->b_coef; a.is_float = c->p.lossy ? 1 : 0;
    a.staging16 = reinterpret_cast<uint16_t *>(c->b_staging);   // (16-bit staging: frame f's at + f * P shorts)
    a.sizes = c->b_sizes; a.plane_scratch = c->b_plane_scratch;
    a.frames = n; a.waves_per_frame = wpf; a.coef_z = coef_z;
    const size_t waves = (size_t)n * (size_t)wpf;
    if (a.k > 0.0f) launch_bulk_encode_frames(c, a, (unsigned)n, bulk_compact(c, 0), s);
    else bpc_encode_kernel<false><<<(unsigned)((waves + kBpcEncWgWaves - 1) / kBpcEncWgWaves), 64 * kBpcEncWgWaves, 0, s>>>(a);
    HIP_TRY(hipGetLastError());

    if (ev) HIP_TRY(hipEventRecord(ev[2], s));
    // ---- pack
    HeaderArg h;
    memset(&h, 0, sizeof h);
    if (first_iter <= 0 && first_iter + n > 0) {          // the batch holds the video's frame 0
        uint16_t hdr[PICSONG_HDR_SHORTS];
        picsong_header_pack(&c->p, hdr);
        memcpy(h.h, hdr, sizeof h.h);
        h.has = -first_iter + 1;
    }
    scan_sizes_kernel<<<(unsigned)n, scan_threads(c->ncb), 0, s>>>(c->b_sizes, c->ncb, c->b_offsets, c->b_total);
    HIP_TRY(hipGetLastError());
    pack_kernel<uint16_t><<<dim3((unsigned)c->ncb, (unsigned)n), 256, 0, s>>>(a.staging16, c->b_sizes, c->b_offsets, c->b_total,
                                                                              c->ncb, h, d_streams, c->P, stream_stride);
    HIP_TRY(hipGetLastError());
    if (ev) HIP_TRY(hipEventRecord(ev[3], s));
    c->last_batch = n;
    return PICSONG_OK;
}

int picsong_last_totals(picsong_ctx *c, void *stream, int n, int *h_totals)
{
    if (!c || !h_totals) return fail(PICSONG_ERR_ARG, "last_totals: null argument");
    if (n < 1 || n > c->last_batch) return fail(PICSONG_ERR_ARG, "last_totals: %d frames, the last batch had %d", n, c->last_batch);
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipMemcpyAsync(c->h_totals, c->b_total, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    for (int i = 0; i < n; i++) h_totals[i] = c->h_totals[i];
    return PICSONG_OK;
}

// n frames decoded through ONE launch per stage: the mirror of picsong_encode_frames (unpack with blockIdx.y =
// frame, one decoder grid over n x nCB codeblocks, grid.z = frame for the inverse transform's levels)
int picsong_decode_frames(picsong_ctx *c, int n, const uint16_t *d_streams, size_t stream_stride, uint8_t *d_frames_out,
                          size_t frame_stride, void *stream)
{
    if (!c || !d_streams || !d_frames_out) return fail(PICSONG_ERR_ARG, "decode_frames: null argument");
    if (n < 1 || n > 64) return fail(PICSONG_ERR_ARG, "decode_frames: n = %d outside 1..64", n);
    if (c->p.cp == 3 || c->p.is_rgb)
        return fail(PICSONG_ERR_ARG, "decode_frames: grey -cp 2 contexts only (-cp 3: frame by frame; RGB: picsong_decode_rgb_frame)");
    if (n > 1 && (stream_stride < picsong_max_stream_shorts(c->aw, c->ah) || frame_stride < c->P))
        return fail(PICSONG_ERR_ARG, "decode_frames: strides %zu shorts / %zu bytes too small", stream_stride, frame_stride);
    if (n == 1) return picsong_decode_frame(c, d_streams, d_frames_out, stream);
    HIP_TRY(hipSetDevice(c->device));
    int rc = ensure_batch(c, n);
    if (rc) return rc;
    c->last_batch = -1;                                     // the batch buffers hold a decode now: no encode totals to hand out
    if ((rc = ensure_coef_i(c, n))) return rc;
    hipStream_t s = (hipStream_t)stream;
    // ---- unpack: lengths, offsets, codewords of the n streams
    const bool direct = dec_from_stream(c);                 // the coder reads the streams themselves: no unpack, no staging
    if (direct) {
        scan_stream_kernel<<<(unsigned)n, scan_threads(c->ncb), 0, s>>>(d_streams, c->ncb, c->b_sizes, c->b_offsets, c->b_total, c->d_flag, stream_stride);
        HIP_TRY(hipGetLastError());
    } else {
        read_sizes_kernel<<<dim3((unsigned)((c->ncb + 255) / 256), (unsigned)n), 256, 0, s>>>(d_streams, c->ncb, c->b_sizes, c->d_flag,
                                                                                          stream_stride);
        HIP_TRY(hipGetLastError());
        scan_sizes_kernel<<<(unsigned)n, scan_threads(c->ncb), 0, s>>>(c->b_sizes, c->ncb, c->b_offsets, c->b_total);
        HIP_TRY(hipGetLastError());
        unpack_kernel<<<dim3((unsigned)c->ncb, (unsigned)n), 256, 0, s>>>(d_streams, c->b_sizes, c->b_offsets, c->ncb, c->b_staging,
                                                                         stream_stride, c->P);
        HIP_TRY(hipGetLastError());
    }
    // ---- decoder: one grid over the n frames' codeblock pairs, both plane-count classes
    BpcArgs a;
    if ((rc = bpc_args(c, a, 0))) return rc;
    const int wpf = (c->ncb + 1) / 2;
    a.cb_base = 0; a.nCB = c->ncb;
    a.coeffs_out = c->b_coef_i; a.staging = c->b_staging; a.sizes = c->b_sizes; a.plane_scratch = c->b_plane_scratch;
    // (the synthesis is planned first: it says whether this call's coefficients can travel as int16)
    bool fused = false;
    const std::vector<InvLaunch> plan = inverse_plan(c, c->b_coef_i, c->b_coef, d_frames_out, &fused, (unsigned)n, frame_stride,
                                                     c->c16_dec && direct);
    const bool c16 = plan_inv_is_c16(plan);
    a.frames = n; a.waves_per_frame = wpf; a.coef_z = (unsigned long long)c->P * (c16 ? 2ull : 4ull);
    const unsigned wgs = (unsigned)(((size_t)n * (size_t)wpf + kBpcDecWgWaves - 1) / kBpcDecWgWaves);
    if (direct) {
        a.cw16 = d_streams; a.cw16_offsets = c->b_offsets; a.cw16_total = c->b_total; a.cw16_stride = stream_stride;
        a.cw16_max = (uint32_t)picsong_max_stream_shorts(c->aw, c->ah);
    }
    if (a.k > 0.0f) launch_bulk_decode_frames(c, a, (unsigned)n, bulk_compact(c, 0), direct, c16, s);
    else if (direct) {
        if (c16) bpc_decode_kernel<false, kDecSmallPlanes, true, true><<<wgs, 64 * kBpcDecWgWaves, 0, s>>>(a);
        else bpc_decode_kernel<false, kDecSmallPlanes, true><<<wgs, 64 * kBpcDecWgWaves, 0, s>>>(a);
    } else {
        bpc_decode_kernel<false, kDecSmallPlanes><<<wgs, 64 * kBpcDecWgWaves, 0, s>>>(a);
    }
    HIP_TRY(hipGetLastError());
    // ---- inverse transform, pixels out of the finest level where its vector kernel applies
    if ((rc = run_inverse(c, plan, s, (unsigned)n))) return rc;
    if (fused) return PICSONG_OK;
    const size_t n4 = c->P / 4;
    const int off = 1 << (c->p.bit_depth - 1);
    const int grid = (int)((n4 + 255) / 256 > 8192 ? 8192 : (n4 + 255) / 256);
    for (int f = 0; f < n; f++) {
        const void *img = (const char *)c->b_coef + ((size_t)f * (c->P + c->extra) + c->extra) * 4;
        if (c->p.lossy) clamp_to_u8_f32_kernel<<<grid, 256, 0, s>>>((const float *)img, d_frames_out + (size_t)f * frame_stride, n4, (float)off);
        else clamp_to_u8_i32_kernel<<<grid, 256, 0, s>>>((const int32_t *)img, d_frames_out + (size_t)f * frame_stride, n4, off);
    }
    HIP_TRY(hipGetLastError());
    return PICSONG_OK;
}

int picsong_copy_last_totals(picsong_ctx *c, void *stream, int n, int32_t *d_totals)
{
    if (!c || !d_totals) return fail(PICSONG_ERR_ARG, "copy_last_totals: null argument");
    // last_batch = 0: the most recent call that packed a stream was a single-frame one (picsong_encode_frame, a stripe,
    // a plane, picsong_bitstream_pack: every writer of d_total resets it); -1: a batched decode has used the buffers
    const bool single = n == 1 && c->last_batch == 0;
    if (!single && (n < 1 || n > c->last_batch))
        return fail(PICSONG_ERR_ARG, "copy_last_totals: %d frames, the last batch had %d", n, c->last_batch);
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpyAsync(d_totals, single ? c->d_total : c->b_total, (size_t)n * sizeof(int32_t),
                           hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return PICSONG_OK;
}

// ---------------------------------------------------------------------------------------------
// RGB path
// ---------------------------------------------------------------------------------------------
int picsong_rgb_forward(picsong_ctx *c, const uint8_t *d_r, const uint8_t *d_g, const uint8_t *d_b, void *d_c0,
                        void *d_c1, void *d_c2, void *stream)
{
    if (!c || !d_r || !d_g || !d_b || !d_c0 || !d_c1 || !d_c2) return fail(PICSONG_ERR_ARG, "rgb_forward: null argument");
    hipStream_t s = (hipStream_t)stream;
    const size_t n4 = c->P / 4;
    const int off = 1 << (c->p.bit_depth - 1);
    const int grid = (int)((n4 + 255) / 256 > 8192 ? 8192 : (n4 + 255) / 256);
    if (c->p.lossy) rgb_forward_kernel<float><<<grid, 256, 0, s>>>(d_r, d_g, d_b, (float *)d_c0, (float *)d_c1, (float *)d_c2, n4, off);
    else rgb_forward_kernel<int32_t><<<grid, 256, 0, s>>>(d_r, d_g, d_b, (int32_t *)d_c0, (int32_t *)d_c1, (int32_t *)d_c2, n4, off);
    HIP_TRY(hipGetLastError());
    return PICSONG_OK;
}

int picsong_rgb_inverse(picsong_ctx *c, const void *d_c0, const void *d_c1, const void *d_c2, uint8_t *d_r,
                        uint8_t *d_g, uint8_t *d_b, void *stream)
{
    if (!c || !d_r || !d_g || !d_b || !d_c0 || !d_c1 || !d_c2) return fail(PICSONG_ERR_ARG, "rgb_inverse: null argument");
    hipStream_t s = (hipStream_t)stream;
    const size_t n4 = c->P / 4;
    const int off = 1 << (c->p.bit_depth - 1);
    const int grid = (int)((n4 + 255) / 256 > 8192 ? 8192 : (n4 + 255) / 256);
    if (c->p.lossy) rgb_inverse_kernel<float><<<grid, 256, 0, s>>>((const float *)d_c0, (const float *)d_c1, (const float *)d_c2, d_r, d_g, d_b, n4, off);
    else rgb_inverse_kernel<int32_t><<<grid, 256, 0, s>>>((const int32_t *)d_c0, (const int32_t *)d_c1, (const int32_t *)d_c2, d_r, d_g, d_b, n4, off);
    HIP_TRY(hipGetLastError());
    return PICSONG_OK;
}

int picsong_encode_plane(picsong_ctx *c, const void *d_plane, int comp, int with_header, uint16_t *d_stream, void *stream)
{
    if (!c || !d_plane || !d_stream) return fail(PICSONG_ERR_ARG, "encode_plane: null argument");
    int rc = ensure_workspace(c, false);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    if ((rc = dwt_forward_impl(c, d_plane, false, c->d_coef, s))) return rc;
    uint16_t *const st16 = reinterpret_cast<uint16_t *>(c->d_staging);
    if ((rc = bpc_encode_impl(c, c->d_coef, st16, c->d_sizes, s, 0, -1, comp))) return rc;
    uint16_t hdr[PICSONG_HDR_SHORTS];
    if (with_header) picsong_header_pack(&c->p, hdr);
    return pack_range(c, st16, c->d_sizes, c->ncb, with_header ? hdr : nullptr, d_stream, s);
}

int picsong_decode_plane(picsong_ctx *c, const uint16_t *d_stream, int comp, void *d_plane_out, void *stream)
{
    if (!c || !d_plane_out || !d_stream) return fail(PICSONG_ERR_ARG, "decode_plane: null argument");
    int rc = ensure_workspace(c, true);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    if ((rc = decode_stream_impl(c, d_stream, c->d_coef_i, s, comp))) return rc;
    return picsong_dwt_inverse(c, c->d_coef_i, d_plane_out, stream);
}


// The three component tables of an RGB context for ONE coder grid: same geometry, every frame f of the batched
// launch coding with table f; waves_per_frame = whole workgroups.
static int bpc_args_rgb(picsong_ctx *c, BpcArgs &a)
{
    int rc = bpc_args(c, a, 0);
    if (rc) return rc;
    for (int k = 1; k < 3; k++) {
        if (!c->has_lut[k]) return fail(PICSONG_ERR_ARG, "no LUT loaded for component %d", k);
        const picsong_lut_info &x = c->li[0], &y = c->li[k];
        if (x.n_bitplanes != y.n_bitplanes || x.n_subbands != y.n_subbands || x.precision != y.precision ||
            x.n_ref != y.n_ref || x.n_sig != y.n_sig || x.n_sign != y.n_sign || x.n_tables != y.n_tables)
            return fail(PICSONG_ERR_ARG, "the components' tables differ in geometry: code the planes one by one (picsong_encode_plane)");
    }
    for (int k = 0; k < 3; k++) a.lut_c[k] = c->d_lut[k];
    const int wpf = (c->ncb + 1) / 2;
    a.cb_base = 0; a.nCB = c->ncb;
    a.frames = 3; a.waves_per_frame = (wpf + kBpcEncWgWaves - 1) / kBpcEncWgWaves * kBpcEncWgWaves;
    return PICSONG_OK;
}

int picsong_encode_rgb_frame(picsong_ctx *c, const uint8_t *d_r, const uint8_t *d_g, const uint8_t *d_b, int header_mask,
                             uint16_t *d_streams, size_t stream_stride, void *stream)
{
    if (!c || !d_r || !d_g || !d_b || !d_streams) return fail(PICSONG_ERR_ARG, "encode_rgb_frame: null argument");
    if (!c->p.is_rgb) return fail(PICSONG_ERR_ARG, "encode_rgb_frame: the context is not an RGB one");
    if (c->p.cp == 3) return fail(PICSONG_ERR_ARG, "encode_rgb_frame: -cp 3 codes its planes one by one (picsong_encode_plane)");
    if (stream_stride < picsong_max_stream_shorts(c->aw, c->ah)) return fail(PICSONG_ERR_ARG, "encode_rgb_frame: stream stride smaller than a worst-case codestream");
    HIP_TRY(hipSetDevice(c->device));
    BpcArgs a;
    int rc = bpc_args_rgb(c, a);
    if (rc) return rc;
    if ((rc = ensure_batch(c, 3))) return rc;
    if ((rc = ensure_coef_i(c, 3))) return rc;
    hipStream_t s = (hipStream_t)stream;
    const size_t coef_z = (c->P + c->extra) * 4;
    char *planes = (char *)c->b_coef_i;
    // ---- lossless: the colour transform in the fused head's load stage (dwt_fwd2_kernel<..., RGB>): the head reads the
    // three u8 planes and delivers component blockIdx.z -- no component plane is ever written (the separate transform
    // kernel reads 100 MB and writes 400 MB of them per 8K frame, and level 0 reads them back)
    bool fused_rgb = false;
    // (the fused head's buffer loads want all three planes 16-byte aligned like a grey frame's; others take the
    // separate colour transform below)
    const bool planes_aligned = ((((uintptr_t)d_r) | ((uintptr_t)d_g) | ((uintptr_t)d_b)) & 15u) == 0;
    if (c->c16 && planes_aligned && !getenv("PICSONG_RGB_NOFUSE")) {
        std::vector<FwdLaunch> plan = plan_dwt_forward(d_r, true, c->b_coef, c->aw, c->ah, c->p.wl, c->p.qs, true);
        Fwd2Launch f2;
        if (plan_is_c16(plan) && plan_dwt_fwd2(plan, f2, true, c->p.lossy != 0, kF2PairsRgb)) {
            for (size_t l = 0; l < plan.size(); l++) {
                plan[l].a.src_z = l == 0 ? 0ull : (unsigned long long)coef_z;      // level 0: every component reads the three planes
                plan[l].a.dst_z = (unsigned long long)coef_z;
            }
            plan[0].a.src_g = d_g; plan[0].a.src_b = d_b;
            f2.a.l0 = plan[0].a; f2.a.l1 = plan[1].a;
            // (RCT on the integer head, ICT on the 9/7 one: the component planes are never written)
            if (c->p.lossy) dwt_fwd2_kernel<float, true, true, kF2PairsRgb, true, true><<<dim3(f2.gx, f2.gy, 3u), 256, 0, s>>>(f2.a);
            else dwt_fwd2_kernel<int, false, true, kF2PairsRgb, true, true><<<dim3(f2.gx, f2.gy, 3u), 256, 0, s>>>(f2.a);
            HIP_TRY(hipGetLastError());
            if ((rc = launch_fwd_levels(c, plan, 2, s, 3u))) return rc;
            a.c16 = 1;
            fused_rgb = true;
        }
    }
    if (!fused_rgb) {
    // ---- colour transform (level shift fused) into three planes, then the transform of all three per launch
    if ((rc = picsong_rgb_forward(c, d_r, d_g, d_b, planes, planes + c->P * 4, planes + 2 * c->P * 4, stream))) return rc;
    std::vector<FwdLaunch> plan = plan_dwt_forward(planes, false, c->b_coef, c->aw, c->ah, c->p.wl, c->p.qs, c->c16);
    a.c16 = plan_is_c16(plan) ? 1 : 0;
    for (size_t l = 0; l < plan.size(); l++) {
        plan[l].a.src_z = l == 0 ? (unsigned long long)c->P * 4ull : (unsigned long long)coef_z;
        plan[l].a.dst_z = (unsigned long long)coef_z;
    }
    if ((rc = launch_fwd_levels(c, plan, 0, s, 3u))) return rc;
    }
    // ---- coder: one grid over the three components' codeblock pairs, component f with table f
    a.coeffs_in = c->b_coef; a.is_float = c->p.lossy ? 1 : 0;
    a.staging16 = reinterpret_cast<uint16_t *>(c->b_staging);
    a.sizes = c->b_sizes; a.plane_scratch = c->b_plane_scratch; a.coef_z = coef_z;
    if (a.k > 0.0f) {
        launch_bulk_encode_frames(c, a, 3u, bulk_compact(c, 0) && bulk_compact(c, 1) && bulk_compact(c, 2), s);
    } else {
        bpc_encode_kernel<false><<<(unsigned)(3 * a.waves_per_frame / kBpcEncWgWaves), 64 * kBpcEncWgWaves, 0, s>>>(a);
    }
    HIP_TRY(hipGetLastError());
    // ---- pack: the populated header on the components of header_mask
    HeaderArg h;
    memset(&h, 0, sizeof h);
    if (header_mask & 7) {
        uint16_t hdr[PICSONG_HDR_SHORTS];
        picsong_header_pack(&c->p, hdr);
        memcpy(h.h, hdr, sizeof h.h);
        h.has = -(header_mask & 7);
    }
    scan_sizes_kernel<<<3, scan_threads(c->ncb), 0, s>>>(c->b_sizes, c->ncb, c->b_offsets, c->b_total);
    HIP_TRY(hipGetLastError());
    pack_kernel<uint16_t><<<dim3((unsigned)c->ncb, 3u), 256, 0, s>>>(a.staging16, c->b_sizes, c->b_offsets, c->b_total, c->ncb, h,
                                                          d_streams, c->P, stream_stride);
    HIP_TRY(hipGetLastError());
    c->last_batch = 3;
    return PICSONG_OK;
}

int picsong_decode_rgb_frame(picsong_ctx *c, const uint16_t *d_streams, size_t stream_stride, uint8_t *d_r, uint8_t *d_g,
                             uint8_t *d_b, void *stream)
{
    if (!c || !d_streams || !d_r || !d_g || !d_b) return fail(PICSONG_ERR_ARG, "decode_rgb_frame: null argument");
    if (!c->p.is_rgb) return fail(PICSONG_ERR_ARG, "decode_rgb_frame: the context is not an RGB one");
    if (c->p.cp == 3) return fail(PICSONG_ERR_ARG, "decode_rgb_frame: -cp 3 decodes its planes one by one (picsong_decode_plane)");
    if (stream_stride < picsong_max_stream_shorts(c->aw, c->ah)) return fail(PICSONG_ERR_ARG, "decode_rgb_frame: stream stride smaller than a worst-case codestream");
    HIP_TRY(hipSetDevice(c->device));
    BpcArgs a;
    int rc = bpc_args_rgb(c, a);
    if (rc) return rc;
    if ((rc = ensure_batch(c, 3))) return rc;
    if ((rc = ensure_coef_i(c, 3))) return rc;
    c->last_batch = -1;
    hipStream_t s = (hipStream_t)stream;
    const bool direct = dec_from_stream(c);
    if (direct) {
        scan_stream_kernel<<<3, scan_threads(c->ncb), 0, s>>>(d_streams, c->ncb, c->b_sizes, c->b_offsets, c->b_total, c->d_flag, stream_stride);
        HIP_TRY(hipGetLastError());
    } else {
        read_sizes_kernel<<<dim3((unsigned)((c->ncb + 255) / 256), 3u), 256, 0, s>>>(d_streams, c->ncb, c->b_sizes, c->d_flag, stream_stride);
        HIP_TRY(hipGetLastError());
        scan_sizes_kernel<<<3, scan_threads(c->ncb), 0, s>>>(c->b_sizes, c->ncb, c->b_offsets, c->b_total);
        HIP_TRY(hipGetLastError());
        unpack_kernel<<<dim3((unsigned)c->ncb, 3u), 256, 0, s>>>(d_streams, c->b_sizes, c->b_offsets, c->ncb, c->b_staging, stream_stride, c->P);
        HIP_TRY(hipGetLastError());
    }
    // (16-bit coefficients between the decoder and the synthesis where the context's magnitudes are bounded: the
    // plan says whether this call's arrays take the vector kernels)
    const std::vector<InvLaunch> plan = inverse_plan(c, c->b_coef_i, c->b_coef, nullptr, nullptr, 3u, 0, c->c16_dec && direct, true);
    const bool c16 = plan_inv_is_c16(plan);
    a.coeffs_out = c->b_coef_i; a.staging = c->b_staging; a.sizes = c->b_sizes; a.plane_scratch = c->b_plane_scratch;
    a.coef_z = (unsigned long long)c->P * (c16 ? 2ull : 4ull);
    const unsigned wgs3 = (unsigned)(3 * a.waves_per_frame / kBpcDecWgWaves);
    if (direct) {
        a.cw16 = d_streams; a.cw16_offsets = c->b_offsets; a.cw16_total = c->b_total; a.cw16_stride = stream_stride;
        a.cw16_max = (uint32_t)picsong_max_stream_shorts(c->aw, c->ah);
    }
    if (a.k > 0.0f) {
        launch_bulk_decode_frames(c, a, 3u, bulk_compact(c, 0) && bulk_compact(c, 1) && bulk_compact(c, 2), direct, c16, s);
    } else if (direct) {
        if (c16) bpc_decode_kernel<false, kDecSmallPlanes, true, true><<<wgs3, 64 * kBpcDecWgWaves, 0, s>>>(a);
        else bpc_decode_kernel<false, kDecSmallPlanes, true><<<wgs3, 64 * kBpcDecWgWaves, 0, s>>>(a);
    } else {
        bpc_decode_kernel<false, kDecSmallPlanes><<<wgs3, 64 * kBpcDecWgWaves, 0, s>>>(a);
    }
    HIP_TRY(hipGetLastError());
    // 5/3 with 16-bit coefficients: the finest level of the three components and the inverse colour transform as ONE
    // launch (dwt_inv_rgb_kernel: the 32-bit planes are never written); PICSONG_RGB_NOFUSE=1 keeps the two
    const bool px_aligned = ((((uintptr_t)d_r) | ((uintptr_t)d_g) | ((uintptr_t)d_b)) & 3u) == 0;
    if (!c->p.lossy && c16 && plan.size() >= 2 && plan.back().vec && px_aligned && !getenv("PICSONG_RGB_NOFUSE")) {
        std::vector<InvLaunch> head(plan.begin(), plan.end() - 1);
        if ((rc = run_inverse(c, head, s, 3u))) return rc;
        const InvLaunch &f = plan.back();
        DwtInvArgs fa = f.a;
        fa.off = 1 << (c->p.bit_depth - 1);
        const dim3 grid(f.gx, f.gy, 1);
        switch (f.band) {
        case 32: dwt_inv_rgb_kernel<32><<<grid, 256, 0, s>>>(fa, d_r, d_g, d_b); break;
        case 16: dwt_inv_rgb_kernel<16><<<grid, 256, 0, s>>>(fa, d_r, d_g, d_b); break;
        case 8: dwt_inv_rgb_kernel<8><<<grid, 256, 0, s>>>(fa, d_r, d_g, d_b); break;
        default: dwt_inv_rgb_kernel<4><<<grid, 256, 0, s>>>(fa, d_r, d_g, d_b); break;
        }
        HIP_TRY(hipGetLastError());
        return PICSONG_OK;
    }
    // 9/7 with 16-bit coefficients (the lean kernel's domain): the finest level of the three components as the three
    // waves of a workgroup, a row pair exchanged through LDS, the inverse ICT at the stores (dwt_inv97_rgb_kernel)
    if (c->p.lossy && c16 && plan.size() >= 2 && plan.back().vec && plan.back().fast && px_aligned && !getenv("PICSONG_RGB_NOFUSE") &&
        !(getenv("PICSONG_DWT_INV97") && atoi(getenv("PICSONG_DWT_INV97")) == 0)) {
        std::vector<InvLaunch> head(plan.begin(), plan.end() - 1);
        if ((rc = run_inverse(c, head, s, 3u))) return rc;
        const InvLaunch &f = plan.back();
        DwtInvArgs fa = f.a;
        fa.off = 1 << (c->p.bit_depth - 1);
        const dim3 grid((unsigned)((fa.W + kStripUseful - 1) / kStripUseful), f.gy, 1);
#define PS_INV97_RGB(B)                                                                              \
        do { if (fa.one_div) dwt_inv97_rgb_kernel<B, true><<<grid, 192, 0, s>>>(fa, d_r, d_g, d_b);   \
             else dwt_inv97_rgb_kernel<B, false><<<grid, 192, 0, s>>>(fa, d_r, d_g, d_b); } while (0)
        switch (f.band) {
        case 32: PS_INV97_RGB(32); break;
        case 16: PS_INV97_RGB(16); break;
        case 8: PS_INV97_RGB(8); break;
        default: PS_INV97_RGB(4); break;
        }
#undef PS_INV97_RGB
        HIP_TRY(hipGetLastError());
        return PICSONG_OK;
    }
    if ((rc = run_inverse(c, plan, s, 3u))) return rc;
    const char *img = (const char *)c->b_coef + c->extra * 4;
    const size_t z = (c->P + c->extra) * 4;
    return picsong_rgb_inverse(c, img, img + z, img + 2 * z, d_r, d_g, d_b, stream);
}

int picsong_pad_frame_host(const uint8_t *in, int w, int h, uint8_t *out, int aw, int ah)
{
    if (!in || !out || w <= 0 || h <= 0 || aw < w || ah < h) return fail(PICSONG_ERR_ARG, "pad_frame: bad argument");
    // column w + j mirrors column w - 1 - j, row h + r mirrors row h - 1 - r: with more added columns
    // (rows) than columns (rows) the reference's loop indexes before its vector's begin
    // (IO/IOManager.ipp:101-108, undefined behaviour on row 0) -- refused here
    if (aw - w > w || ah - h > h)
        return fail(PICSONG_ERR_ARG, "pad_frame: %dx%d -> %dx%d adds more columns/rows than the frame has", w, h, aw, ah);
    for (int y = 0; y < h; y++) {
        memcpy(out + (size_t)y * aw, in + (size_t)y * w, (size_t)w);
        for (int j = 0; j < aw - w; j++) out[(size_t)y * aw + w + j] = in[(size_t)y * w + (w - 1 - j)];
    }
    for (int r = 0; r < ah - h; r++) memcpy(out + (size_t)(h + r) * aw, out + (size_t)(h - 1 - r) * aw, (size_t)aw);
    return PICSONG_OK;
}

}  // extern "C"

#ifdef PICSONG_DWT_TRACE
// variant builds only (tools/dwt_trace.py): device buffer the fused DWT head's waves stamp their phases into
extern "C" int picsong_debug_set_trace(void *d_buf)
{
    unsigned long long *p = (unsigned long long *)d_buf;
    return hipMemcpyToSymbol(HIP_SYMBOL(picsong::g_dwt_trace), &p, sizeof p) == hipSuccess ? 0 : -1;
}
extern "C" int picsong_debug_set_bpc_trace(void *d_buf)
{
    unsigned long long *p = (unsigned long long *)d_buf;
    return hipMemcpyToSymbol(HIP_SYMBOL(picsong::g_bpc_trace), &p, sizeof p) == hipSuccess ? 0 : -1;
}
#endif
