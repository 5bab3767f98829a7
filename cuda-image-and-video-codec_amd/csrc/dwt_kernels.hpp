// dwt_kernels.hpp -- one level of the 2-D lifting DWT for gfx950 (CDNA4), forward and inverse,
// reversible 5/3 (int32) and irreversible 9/7 (fp32, quantisation gain fused into the write,
// de-quantisation fused into the read).
//
// Replaces kernelDWTForward / kernelDWTForwardLossy / kernelDWTReverse / kernelDWTReverseLossy
// (reference DWT/DWTGenerator.cu:753-812, 878-936, 1002-1124) and offsetImage
// (Engines/CodingEngine.cu:581-588, fused into level 0 when the input is u8).  Arithmetic and
// operation order per sample are those of the reference's lifting steps (:72-122, vertical
// :137-272, horizontal :279-339, quantisation :405-419, de-quantisation :513-553), so the
// coefficients are bit-identical to whole-image lifting with whole-sample symmetric extension.
//
// Design (not the reference's 64x18 register tile per 32-lane warp):
//   * a wave64 owns a strip of 256 columns, 4 adjacent columns (16 B) per lane, and STREAMS down
//     a band of rows: vertical lifting is a sliding window of 4 (9/7) / 2 (5/3) state registers
//     per column, so every input row is loaded once per band, fully coalesced (1 KiB per wave);
//   * horizontal lifting runs in registers; the two samples a lane needs from its neighbours come
//     by DPP wave_shr:1 / wave_shl:1 (no LDS, no bank conflicts);
//   * image borders use no special-case code: out-of-image rows / columns are loaded from their
//     mirror position (whole-sample symmetric extension of the signal), after which interior
//     lifting is exact; the first/last lane of a wave and the first rows of a band are overlap
//     that is recomputed, never written (4 columns each side: 1.6 % re-read per strip);
//   * subband writes are 8-byte vectors, contiguous across the wave per subband row.
//   * VEC instantiations (level width a multiple of 4, 16-byte aligned rows -- every level of every
//     frame the CLI accepts down to W = 4) contain ONLY vector memory instructions: 4-byte (u8) or
//     16-byte loads, 8-byte subband stores, 16-byte image stores.  They need no mirrored column
//     loads at all: the extended signal is symmetric about column 0 and column W-1, every lifting
//     step preserves that symmetry, so the lane that owns column 0 (W-1) takes the value it would
//     have fetched from its left (right) neighbour from its own mirrored sample instead.  The
//     !VEC instantiations keep per-column mirrored scalar accesses for odd geometries.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cmath>
#include <type_traits>

namespace picsong {

// DWT/DWTGenerator.cuh:16-22
#define PS_A1 (-1.586134342059924f)
#define PS_A2 (-0.052980118572961f)
#define PS_A3 (0.882911075530934f)
#define PS_A4 (0.443506852043971f)
#define PS_N1 (1.230174104914001f)
#define PS_N2 (0.812893066f)

#ifndef PICSONG_DWT_EDGE_LANES
#define PICSONG_DWT_EDGE_LANES 1
#endif
constexpr int kStripCols = 256;                 // columns held by one wave
constexpr int kEdgeLanes = PICSONG_DWT_EDGE_LANES;          // recomputed, never-written lanes per side (>= 1)
constexpr int kStripUseful = kStripCols - 8 * kEdgeLanes;   // lanes kEdgeLanes .. 63-kEdgeLanes write
#ifndef PICSONG_DWT_INV_GROUP
#define PICSONG_DWT_INV_GROUP 3       // 9/7 inverse kernels: unrolled iterations per trip of the counted loop (at most)
#endif
constexpr int inv_group(int iters, int most) { return iters % most == 0 ? most : inv_group(iters, most - 1); }
#ifndef PICSONG_DWT_INV97_GROUP
#define PICSONG_DWT_INV97_GROUP 2     // dwt_inv97_kernel: likewise (3 spills at its 96 registers)
#endif
#ifndef PICSONG_DWT_INV_AHEAD
#define PICSONG_DWT_INV_AHEAD 6       // inverse kernels: row pairs whose loads are in flight ahead of the math
#endif
// Output rows per band are a template parameter (BAND = 4, 8, 16 or 32): big levels want tall bands
// (less vertical halo), small levels want many short waves (a level with 18 tall waves is bound by
// one wave's serial instruction time, not by memory).  Row pairs fetched ahead = min(4, BAND/2).
// (the inverse kernel takes its band height the same way)

struct DwtFwdArgs {
    const void *src;        // level input: T[ H x W ] (stride src_stride) or u8 at level 0
    int src_stride;
    int W, H;               // this level's dimensions
    void *ll;               // LL destination (packed scratch or Mallat origin on the last level)
    int ll_stride;
    void *mallat;           // Mallat array origin (row stride AW)
    int AW;
    int level;              // 0 = finest (quantisation row)
    int last;               // last level: LL gets quantised (lossy)
    float qs;
    float q[4];             // quantisation steps of this level: LL, HL, LH, HH
    // batched launches (grid.z = frames of one picsong_encode_frames call): frame z reads src + z * src_z
    // and writes ll / mallat + z * dst_z (bytes); 0 for a single frame
    unsigned long long src_z, dst_z;
    // row band (picsong_dwt_forward_band, intra-frame sharding): the launch produces the row pairs
    // [pair_base, pair_end) of this level only; 0, 0 = the whole level.  Input rows are still addressed in
    // frame coordinates: the band's rows and its halo (2 rows either side for 5/3, 4 for 9/7) must be there.
    int pair_base, pair_end;
    // frame paths: the CODED subbands (HL / LH / HH of every level, LL of the last) leave as 16-bit integers -- the
    // value the coder's load makes of a coefficient anyway (truncation toward zero of a float one) -- in a Mallat array
    // of int16 with row stride AW at `mallat`; the LL a next level reads stays T in the scratch.  Half the bytes the
    // transform writes and the coder reads.  Only where the magnitudes are bounded below 2^15 (coef16_ok,
    // launch_plan.hpp: 8-bit samples); the stage-by-stage API keeps the reference's T arrays.
    int c16;
    // fused head, RGB frames (dwt_fwd2_kernel<..., RGB>): `src` is the R plane, these the G and B planes (padded u8,
    // same stride); the launch's blockIdx.z is the COMPONENT the RCT delivers (src_z = 0: every component reads all three)
    const void *src_g, *src_b;
};

// two integer coefficients as one dword of int16 (v_cvt_pk_i16_i32: saturating, which coef16_ok's bound never needs)
__device__ __forceinline__ uint32_t pack_i16(int x, int y)
{
#if defined(__AMDGCN__)
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    const s16x2 v = __builtin_amdgcn_cvt_pk_i16(x, y);
    return __builtin_bit_cast(uint32_t, v);
#else
    auto sat = [](int v) -> uint32_t { return (uint32_t)(v > 32767 ? 32767 : (v < -32768 ? -32768 : v)) & 0xFFFFu; };
    return sat(x) | (sat(y) << 16);
#endif
}
__device__ __forceinline__ uint32_t pack_c16(int x, int y) { return pack_i16(x, y); }
__device__ __forceinline__ uint32_t pack_c16(float x, float y) { return pack_i16((int)x, (int)y); }     // BPCEngine.cu:49: toward zero
__device__ __forceinline__ int c16_lo(uint32_t w) { return (int)(int16_t)(w & 0xFFFFu); }
__device__ __forceinline__ int c16_hi(uint32_t w) { return (int)w >> 16; }

// frame blockIdx.z of a batched launch
__device__ __forceinline__ void dwt_fwd_select_frame(DwtFwdArgs &a, unsigned bz)
{
    const unsigned long long z = bz;
    a.src = (const char *)a.src + z * a.src_z;
    a.ll = (char *)a.ll + z * a.dst_z;
    a.mallat = (char *)a.mallat + z * a.dst_z;
}
__device__ __forceinline__ void dwt_fwd_select_frame(DwtFwdArgs &a) { dwt_fwd_select_frame(a, blockIdx.z); }

struct DwtInvArgs {
    const int32_t *mallat;  // coded coefficients, int32, row stride AW
    int AW;
    const void *ll;         // LL source: Mallat (first == 1) or previous level's packed output (T)
    int ll_stride;
    int first;              // coarsest level: LL comes from `mallat` (and is de-quantised if lossy)
    int W, H;               // this level's OUTPUT dimensions
    void *dst;              // packed output T[H x W]
    float qs;
    float q[4];
    float rqs, rq[4];       // correctly rounded reciprocals 1.0f / qs, 1.0f / q[k] (FAST kernels)
    uint8_t *dst_u8;        // U8OUT kernels (finest level of the frame path): pixels, row stride W
    int off;                // level shift to add back (128 for 8-bit)
    // batched launches (grid.z = frames of one picsong_decode_frames call): frame z reads mallat + z * mallat_z and
    // ll + z * ll_z, writes dst + z * dst_z and dst_u8 + z * u8_z (bytes); 0 for a single frame
    unsigned long long mallat_z, ll_z, dst_z, u8_z;
    // dwt_inv97_kernel: qs is a power of two (the two de-quantising divisions are one: x / (q * qs) is exact scaling)
    int one_div;
    int exact_replay;       // debug: every wave of dwt_inv97_kernel runs its band a second time with true divisions
    // decode frame paths: the coded coefficients at `mallat` are 16-bit integers (an int16 Mallat array, row stride AW) --
    // what the decoder's C16 instantiation writes where every magnitude stays below 2^15 (coef16_ok); the C16
    // instantiations of the synthesis kernels read them.  The LL a level hands the next stays T in the scratch.
    int c16;
};

// frame blockIdx.z of a batched launch
__device__ __forceinline__ void dwt_inv_select_frame(DwtInvArgs &a)
{
    const unsigned long long z = blockIdx.z;
    a.mallat = (const int32_t *)((const char *)a.mallat + z * a.mallat_z);
    a.ll = (const char *)a.ll + z * a.ll_z;
    a.dst = (char *)a.dst + z * a.dst_z;
    if (a.dst_u8) a.dst_u8 += z * a.u8_z;
}

__device__ __forceinline__ int reflect(int i, int n)
{
    // whole-sample symmetric extension: x[-k] = x[k], x[n-1+k] = x[n-1-k]
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return i < 0 ? 0 : (i >= n ? n - 1 : i);
}
// subband-domain mirrors of the same extension (K = samples in the subband)
__device__ __forceinline__ int reflect_s(int m, int K)
{
    if (m < 0) m = -m;
    if (m >= K) m = 2 * K - 1 - m;
    return m < 0 ? 0 : (m >= K ? K - 1 : m);
}
__device__ __forceinline__ int reflect_d(int m, int K)
{
    if (m < 0) m = -m - 1;
    if (m >= K) m = 2 * K - 2 - m;
    return m < 0 ? 0 : (m >= K ? K - 1 : m);
}

// bit casts between the 32-bit sample types and raw dwords (no pointer punning)
__device__ __forceinline__ uint32_t as_u32(int v) { return (uint32_t)v; }
__device__ __forceinline__ uint32_t as_u32(float v) { return __float_as_uint(v); }
template <typename T> __device__ __forceinline__ T from_u32(uint32_t u);
template <> __device__ __forceinline__ int from_u32<int>(uint32_t u) { return (int)u; }
template <> __device__ __forceinline__ float from_u32<float>(uint32_t u) { return __uint_as_float(u); }

template <typename T> __device__ __forceinline__ T dpp_prev(T v);
template <typename T> __device__ __forceinline__ T dpp_next(T v);
template <> __device__ __forceinline__ int dpp_prev<int>(int v)
{ return (int)__builtin_amdgcn_update_dpp(0u, (uint32_t)v, 0x138, 0xf, 0xf, false); }
template <> __device__ __forceinline__ int dpp_next<int>(int v)
{ return (int)__builtin_amdgcn_update_dpp(0u, (uint32_t)v, 0x130, 0xf, 0xf, false); }
template <> __device__ __forceinline__ float dpp_prev<float>(float v)
{ return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x138, 0xf, 0xf, false)); }
template <> __device__ __forceinline__ float dpp_next<float>(float v)
{ return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x130, 0xf, 0xf, false)); }

// ---- wave-uniform row offset + per-lane byte offset addressing ---------------------------------
// A plane addressed as base + row offset (an SGPR: 32-bit scalar arithmetic, P * 4 < 2^32) + lane offset (one
// VGPR per column position, computed once per wave): buffer instructions, so a row's address costs one or
// two scalar instructions and no vector instruction at all (per-row 64-bit pointers cost the fused kernel
// a v_mad_u64 or two per access and 40 % of its instructions were scalar address arithmetic).  A lane that
// must not store gets the offset kRbDrop: past the resource's num_records, the hardware drops the access
// (raw buffer range check: lane offset against num_records), so stores need no exec mask.
constexpr uint32_t kRbDrop = 0x80000000u;
struct RowBuf {
#if defined(__AMDGCN__)
    __amdgpu_buffer_rsrc_t rs;
#else
    char *base;
#endif
};
__device__ __forceinline__ RowBuf rowbuf(const void *p)
{
    RowBuf b;
#if defined(__AMDGCN__)
    b.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, 0x7FFFFFFF, 0x00020000);
#else
    b.base = const_cast<char *>((const char *)p);
#endif
    return b;
}
__device__ __forceinline__ uint32_t rb_load32(const RowBuf &b, uint32_t lane_off, uint32_t row_off)
{
#if defined(__AMDGCN__)
    return __builtin_amdgcn_raw_buffer_load_b32(b.rs, lane_off, row_off, 0);
#else
    uint32_t v = 0;
    if (lane_off < kRbDrop) memcpy(&v, b.base + row_off + lane_off, 4);
    return v;
#endif
}
__device__ __forceinline__ uint4 rb_load128(const RowBuf &b, uint32_t lane_off, uint32_t row_off)
{
#if defined(__AMDGCN__)
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(b.rs, lane_off, row_off, 0);
    uint4 r; r.x = w.x; r.y = w.y; r.z = w.z; r.w = w.w;
    return r;
#else
    uint4 r = { 0u, 0u, 0u, 0u };
    if (lane_off < kRbDrop) memcpy(&r, b.base + row_off + lane_off, 16);
    return r;
#endif
}
// a 16-bit word, sign-extended
__device__ __forceinline__ int rb_load16s(const RowBuf &b, uint32_t lane_off, uint32_t row_off)
{
#if defined(__AMDGCN__)
    return (int)(short)__builtin_amdgcn_raw_buffer_load_b16(b.rs, lane_off, row_off, 0);
#else
    int16_t v = 0;
    if (lane_off < kRbDrop) memcpy(&v, b.base + row_off + lane_off, 2);
    return (int)v;
#endif
}
__device__ __forceinline__ uint2 rb_load64(const RowBuf &b, uint32_t lane_off, uint32_t row_off)
{
#if defined(__AMDGCN__)
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(b.rs, lane_off, row_off, 0);
    uint2 r; r.x = w.x; r.y = w.y;
    return r;
#else
    uint2 r = { 0u, 0u };
    if (lane_off < kRbDrop) memcpy(&r, b.base + row_off + lane_off, 8);
    return r;
#endif
}
__device__ __forceinline__ void rb_store128(const RowBuf &b, uint32_t lane_off, uint32_t row_off, uint32_t x, uint32_t y,
                                            uint32_t z, uint32_t w)
{
#if defined(__AMDGCN__)
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 v; v.x = x; v.y = y; v.z = z; v.w = w;
    __builtin_amdgcn_raw_buffer_store_b128(v, b.rs, lane_off, row_off, 0);
    // A buffer store of more than 64 bits reads its data registers for a few cycles after issue, and a vector
    // instruction that overwrites them in the next two wait states wins in the last lanes read (lanes 12-15 of each
    // row of 16 got the NEXT value: seen as frexp exponents in place of samples: round 2, profiles/NOTES.md).  The
    // compiler's hazard recognizer inserts the wait states for flat / global stores and for buffer stores WITHOUT a
    // scalar offset only (GCNHazardRecognizer::createsVALUHazard: "this hazard only exists if the instruction is not
    // using a register in the soffset field") -- on gfx950 it exists with one too.  So the two wait states are here,
    // fenced so that nothing is scheduled between the store and them.
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 1");
    __builtin_amdgcn_sched_barrier(0);
#else
    if (lane_off < kRbDrop) { const uint32_t v[4] = { x, y, z, w }; memcpy(b.base + row_off + lane_off, v, 16); }
#endif
}
__device__ __forceinline__ void rb_store32(const RowBuf &b, uint32_t lane_off, uint32_t row_off, uint32_t x)
{
#if defined(__AMDGCN__)
    __builtin_amdgcn_raw_buffer_store_b32(x, b.rs, lane_off, row_off, 0);
#else
    if (lane_off < kRbDrop) memcpy(b.base + row_off + lane_off, &x, 4);
#endif
}
__device__ __forceinline__ void rb_store16(const RowBuf &b, uint32_t lane_off, uint32_t row_off, uint32_t x)
{
#if defined(__AMDGCN__)
    __builtin_amdgcn_raw_buffer_store_b16((short)x, b.rs, lane_off, row_off, 0);
#else
    if (lane_off < kRbDrop) { const uint16_t h = (uint16_t)x; memcpy(b.base + row_off + lane_off, &h, 2); }
#endif
}
__device__ __forceinline__ void rb_store64(const RowBuf &b, uint32_t lane_off, uint32_t row_off, uint32_t x, uint32_t y)
{
#if defined(__AMDGCN__)
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    u32x2 w; w.x = x; w.y = y;
    __builtin_amdgcn_raw_buffer_store_b64(w, b.rs, lane_off, row_off, 0);
#else
    if (lane_off < kRbDrop) { memcpy(b.base + row_off + lane_off, &x, 4); memcpy(b.base + row_off + lane_off + 4, &y, 4); }
#endif
}

// ---- time-resolved trace (variant builds only: -DPICSONG_DWT_TRACE) ------------------------------
// Each wave of the fused head stamps s_memrealtime (100 MHz, chip-wide) at six points into a buffer the
// tool sets with picsong_debug_set_trace (tools/dwt_trace.py).  Not compiled into the product library.
#ifdef PICSONG_DWT_TRACE
__device__ unsigned long long *g_dwt_trace = nullptr;
#define PS_TRACE(slot) do { if (g_dwt_trace && (threadIdx.x & 63) == 0) g_dwt_trace[(((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x * 4 + blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define PS_TRACE_WAIT() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define PS_TRACE(slot) do { } while (0)
#define PS_TRACE_WAIT() do { } while (0)
#endif

// ---- division by a constant ------------------------------------------------------------------
// The 9/7 synthesis divides (the reference writes x / K, x / 0.812893066, (m / Q) / qs; the forward
// transform multiplies), and a correctly rounded fp32 division is ~11 instructions.  With the
// correctly rounded reciprocal rc = 1.0f / c at hand, q = x*rc; r = fma(-q, c, x); q + r*rc (one
// Newton step on the residual, Markstein) is the SAME correctly rounded quotient in 3.  Where that is
// used, it is checked, not assumed: for the two lifting constants exhaustively over every float
// with exponent >= -96 (tools/div_check.c; below that the residual underflows, and such a value
// takes the division: div_n1n2), for the quantisation steps over their whole input domain (|v| + 0.5,
// |v| < 65536) and for the context's qs over every value that can reach it (dequant_fast_ok,
// launch_plan.hpp, at context creation; a qs that fails selects the kernels that divide).
__host__ __device__ __forceinline__ float div_rc(float x, float c, float rc)
{
    const float q = x * rc;
    const float r = fmaf(-q, c, x);
    return fmaf(r, rc, q);
}
#define PS_RN1 (1.0f / PS_N1)
#define PS_RN2 (1.0f / PS_N2)
// (bits << 1) - 1: 0xFFFFFFFF for +-0, small for a nonzero value of small magnitude
__device__ __forceinline__ uint32_t tiny_key(float x) { return (__float_as_uint(x) << 1) - 1u; }
constexpr uint32_t kTinyKey = ((uint32_t)(127 - 96) << 24) - 1u;       // keys below: 0 < |x| < 2^-96
__device__ __forceinline__ void keep_in_branch(float &x)
{   // the dividing path must stay a branch: if-converted, every lane would pay for both forms
#if defined(__AMDGCN__)
    asm volatile("" : "+v"(x));
#else
    (void)x;
#endif
}
// h0, h1 /= N1 and l0, l1 /= N2: the lifting divisions of one lane's two sample pairs.  The
// reciprocal form unless some lane of the wave holds a value so small that its residual would
// underflow (never seen on image data; then the whole wave divides).
template <bool FAST>
__device__ __forceinline__ void div_n1n2(float &h0, float &h1, float &l0, float &l1)
{
    if constexpr (FAST) {
        const uint32_t ka = tiny_key(h0) < tiny_key(h1) ? tiny_key(h0) : tiny_key(h1);
        const uint32_t kb = tiny_key(l0) < tiny_key(l1) ? tiny_key(l0) : tiny_key(l1);
        if (__builtin_amdgcn_ballot_w64((ka < kb ? ka : kb) < kTinyKey) == 0ull) {
            h0 = div_rc(h0, PS_N1, PS_RN1); h1 = div_rc(h1, PS_N1, PS_RN1);
            l0 = div_rc(l0, PS_N2, PS_RN2); l1 = div_rc(l1, PS_N2, PS_RN2);
            return;
        }
        keep_in_branch(h0); keep_in_branch(h1); keep_in_branch(l0); keep_in_branch(l1);
    }
    h0 = h0 / PS_N1; h1 = h1 / PS_N1;
    l0 = l0 / PS_N2; l1 = l1 / PS_N2;
}

// ---- horizontal analysis of one row held as (e0,o0,e1,o1) per lane ----------------------------
// le / re: the lane owns the first / last four columns of the row (VEC kernels): the neighbour
// sample it needs is its own mirror image (x[-1] = x[1], x[W] = x[W-2]; same for every
// intermediate of the lifting).  Both are false in the !VEC kernels, whose edge lanes were loaded
// with mirrored columns.
template <typename T> __device__ __forceinline__ T nxt(T own, T mine, bool re) { T n = dpp_next<T>(own); return re ? mine : n; }
template <typename T> __device__ __forceinline__ T prv(T own, T mine, bool le) { T n = dpp_prev<T>(own); return le ? mine : n; }

__device__ __forceinline__ void hfwd(int v[4], bool le, bool re)
{   // DWTGenerator.cu:279-292, lifting :72-76
    int en = nxt<int>(v[0], v[2], re);
    v[1] -= (v[0] + v[2]) >> 1;
    v[3] -= (v[2] + en) >> 1;
    int dp = prv<int>(v[3], v[1], le);
    v[0] += (dp + v[1] + 2) >> 2;
    v[2] += (v[1] + v[3] + 2) >> 2;
}
__device__ __forceinline__ void hfwd(float v[4], bool le, bool re)
{   // DWTGenerator.cu:311-323, lifting :91-104
    float en = nxt<float>(v[0], v[2], re);
    v[1] = fmaf(v[0] + v[2], PS_A1, v[1]);
    v[3] = fmaf(v[2] + en, PS_A1, v[3]);
    float dp = prv<float>(v[3], v[1], le);
    v[0] = fmaf(v[1] + dp, PS_A2, v[0]);
    v[2] = fmaf(v[3] + v[1], PS_A2, v[2]);
    float sn = nxt<float>(v[0], v[2], re);
    v[1] = fmaf(v[0] + v[2], PS_A3, v[1]);
    v[3] = fmaf(v[2] + sn, PS_A3, v[3]);
    dp = prv<float>(v[3], v[1], le);
    v[0] = fmaf(v[1] + dp, PS_A4, v[0]) * PS_N2;
    v[2] = fmaf(v[3] + v[1], PS_A4, v[2]) * PS_N2;
    v[1] *= PS_N1;
    v[3] *= PS_N1;
}
// ---- horizontal synthesis of one row held as (s0,d0,s1,d1) per lane ---------------------------
template <bool FAST>
__device__ __forceinline__ void hinv(int v[4], bool le, bool re)
{   // DWTGenerator.cu:295-308, lifting :81-85
    int dp = prv<int>(v[3], v[1], le);
    v[0] -= (v[1] + dp + 2) >> 2;
    v[2] -= (v[3] + v[1] + 2) >> 2;
    int sn = nxt<int>(v[0], v[2], re);
    v[1] += (v[0] + v[2]) >> 1;
    v[3] += (v[2] + sn) >> 1;
}
template <bool FAST>
__device__ __forceinline__ void hinv(float v[4], bool le, bool re)
{   // DWTGenerator.cu:326-339, lifting :110-122
    div_n1n2<FAST>(v[1], v[3], v[0], v[2]);
    float dp = prv<float>(v[3], v[1], le);
    v[0] = fmaf(-(v[1] + dp), PS_A4, v[0]);
    v[2] = fmaf(-(v[3] + v[1]), PS_A4, v[2]);
    float sn = nxt<float>(v[0], v[2], re);
    v[1] = fmaf(-(v[0] + v[2]), PS_A3, v[1]);
    v[3] = fmaf(-(v[2] + sn), PS_A3, v[3]);
    dp = prv<float>(v[3], v[1], le);
    v[0] = fmaf(-(v[1] + dp), PS_A2, v[0]);
    v[2] = fmaf(-(v[3] + v[1]), PS_A2, v[2]);
    sn = nxt<float>(v[0], v[2], re);
    v[1] = fmaf(-(v[0] + v[2]), PS_A1, v[1]);
    v[3] = fmaf(-(v[2] + sn), PS_A1, v[3]);
}

// ---- forward --------------------------------------------------------------------------------
// A row segment as it comes from memory (conversion deferred so that many loads can be in flight):
// u8 input: one dword = 4 samples; T input: four dwords.
template <bool U8IN> struct RawRow;
template <> struct RawRow<true> { uint32_t w; };
template <> struct RawRow<false> { uint32_t w[4]; };

template <typename T, bool U8IN, bool VEC>
__device__ __forceinline__ RawRow<U8IN> load_raw(const DwtFwdArgs &a, int y, int c0)
{
    RawRow<U8IN> r;
    const int ry = reflect(y, a.H);
    if constexpr (U8IN) {
        const uint8_t *p = (const uint8_t *)a.src + (size_t)ry * (size_t)a.src_stride;
        if constexpr (VEC) {
            // uniform row base + unsigned 32-bit lane offset: the saddr addressing form, no 64-bit VALU
            r.w = *reinterpret_cast<const uint32_t *>(p + (uint32_t)c0);
        } else {
            r.w = (uint32_t)p[reflect(c0, a.W)] | ((uint32_t)p[reflect(c0 + 1, a.W)] << 8) |
                  ((uint32_t)p[reflect(c0 + 2, a.W)] << 16) | ((uint32_t)p[reflect(c0 + 3, a.W)] << 24);
        }
    } else {
        const uint32_t *p = (const uint32_t *)a.src + (size_t)ry * (size_t)a.src_stride;
        if constexpr (VEC) {
            const uint4 q = *reinterpret_cast<const uint4 *>(p + (uint32_t)c0);
            r.w[0] = q.x; r.w[1] = q.y; r.w[2] = q.z; r.w[3] = q.w;
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) r.w[k] = p[reflect(c0 + k, a.W)];
        }
    }
    return r;
}

// a wave-uniform float held in a vector register the optimiser cannot turn back into its scalar source
__device__ __forceinline__ float in_vgpr(float x)
{
#if defined(__AMDGCN__)
    float v;
    asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"(x));
    return v;
#else
    return x;
#endif
}
// byte N of a dword as a float: one conversion (v_cvt_f32_ubyteN) and, for the level shift, one full-rate
// v_add_f32 -- left to itself the compiler subtracts in the integer domain first (v_add_u32_sdwa +
// v_cvt_f32_i32: two half-rate instructions per sample)
template <int N>
__device__ __forceinline__ float byte_to_float(uint32_t w)
{
#if defined(__AMDGCN__)
    float f;
    if constexpr (N == 0) asm("v_cvt_f32_ubyte0 %0, %1" : "=v"(f) : "v"(w));
    else if constexpr (N == 1) asm("v_cvt_f32_ubyte1 %0, %1" : "=v"(f) : "v"(w));
    else if constexpr (N == 2) asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(f) : "v"(w));
    else asm("v_cvt_f32_ubyte3 %0, %1" : "=v"(f) : "v"(w));
    return f;
#else
    return (float)(int)((w >> (8 * N)) & 0xFFu);
#endif
}
template <typename T, bool U8IN>
__device__ __forceinline__ void unpack_row(const RawRow<U8IN> &r, T v[4])
{
    if constexpr (U8IN && std::is_same<T, float>::value) {
        // offsetImage Engines/CodingEngine.cu:581-588 fused: (T)u8 - 128
        v[0] = byte_to_float<0>(r.w) - 128.0f; v[1] = byte_to_float<1>(r.w) - 128.0f;
        v[2] = byte_to_float<2>(r.w) - 128.0f; v[3] = byte_to_float<3>(r.w) - 128.0f;
    } else if constexpr (U8IN) {
        v[0] = (T)(int)(r.w & 0xFFu) - (T)128; v[1] = (T)(int)((r.w >> 8) & 0xFFu) - (T)128;
        v[2] = (T)(int)((r.w >> 16) & 0xFFu) - (T)128; v[3] = (T)(int)(r.w >> 24) - (T)128;
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = from_u32<T>(r.w[k]);
    }
}

template <typename T, bool VEC>
__device__ __forceinline__ void store2(T *p, T x, T y, bool two)
{
    if constexpr (VEC) {
        uint2 w;
        w.x = as_u32(x);
        w.y = as_u32(y);
        *reinterpret_cast<uint2 *>(p) = w;
    } else { p[0] = x; if (two) p[1] = y; }
}

// writeSubbands DWTGenerator.cu:403-433 + placement :719-723.  Lrow/Hrow: vertically low / high
// rows after horizontal analysis (s0,d0,s1,d1): s -> LL / LH, d -> HL / HH.
template <typename T, bool LOSSY, bool VEC>
__device__ __forceinline__ void emit_pair(const DwtFwdArgs &a, int m, int pc, bool wr, bool le, bool re,
                                          T Lr[4], T Hr[4])
{
    hfwd(Lr, le, re);
    hfwd(Hr, le, re);
    if (!wr || m < 0 || m >= (a.H >> 1)) return;
    T ll0 = Lr[0], ll1 = Lr[2], hl0 = Lr[1], hl1 = Lr[3];
    T lh0 = Hr[0], lh1 = Hr[2], hh0 = Hr[1], hh1 = Hr[3];
    if (LOSSY) {
        if (a.last) { ll0 = (T)(((float)ll0 * a.q[0]) * a.qs); ll1 = (T)(((float)ll1 * a.q[0]) * a.qs); }
        hl0 = (T)(((float)hl0 * a.q[1]) * a.qs); hl1 = (T)(((float)hl1 * a.q[1]) * a.qs);
        lh0 = (T)(((float)lh0 * a.q[2]) * a.qs); lh1 = (T)(((float)lh1 * a.q[2]) * a.qs);
        hh0 = (T)(((float)hh0 * a.q[3]) * a.qs); hh1 = (T)(((float)hh1 * a.q[3]) * a.qs);
    }
    const int hW = a.W >> 1, hH = a.H >> 1;
    const bool two = pc + 1 < hW;
    T *mal = (T *)a.mallat;
    if constexpr (VEC) {
        // uniform row bases (SALU) + one unsigned 32-bit lane offset shared by the four stores
        const uint32_t vo = (uint32_t)pc;
        if (a.c16) {                                         // (wave-uniform)
            int16_t *q0 = (int16_t *)a.mallat + (size_t)m * (size_t)a.AW, *q1 = (int16_t *)a.mallat + (size_t)(m + hH) * (size_t)a.AW;
            if (a.last) *reinterpret_cast<uint32_t *>(q0 + vo) = pack_c16(ll0, ll1);
            else store2<T, true>((T *)a.ll + (size_t)m * (size_t)a.ll_stride + vo, ll0, ll1, true);
            *reinterpret_cast<uint32_t *>(q0 + hW + vo) = pack_c16(hl0, hl1);
            *reinterpret_cast<uint32_t *>(q1 + vo) = pack_c16(lh0, lh1);
            *reinterpret_cast<uint32_t *>(q1 + hW + vo) = pack_c16(hh0, hh1);
            return;
        }
        T *rl = (T *)a.ll + (size_t)m * (size_t)a.ll_stride;
        T *r0 = mal + (size_t)m * (size_t)a.AW, *r1 = mal + (size_t)(m + hH) * (size_t)a.AW;
        store2<T, true>(rl + vo, ll0, ll1, true);
        store2<T, true>(r0 + hW + vo, hl0, hl1, true);
        store2<T, true>(r1 + vo, lh0, lh1, true);
        store2<T, true>(r1 + hW + vo, hh0, hh1, true);
        return;
    }
    store2<T, VEC>((T *)a.ll + (size_t)m * (size_t)a.ll_stride + pc, ll0, ll1, two);
    store2<T, VEC>(mal + (size_t)m * (size_t)a.AW + hW + pc, hl0, hl1, two);
    store2<T, VEC>(mal + (size_t)(m + hH) * (size_t)a.AW + pc, lh0, lh1, two);
    store2<T, VEC>(mal + (size_t)(m + hH) * (size_t)a.AW + hW + pc, hh0, hh1, two);
}

// grid.x = ceil(strips / 4), grid.y = bands; block = 256 threads = 4 waves = 4 adjacent strips
#ifdef PICSONG_DWT_WAVES_PER_EU
#define PS_DWT_OCC __attribute__((amdgpu_waves_per_eu(PICSONG_DWT_WAVES_PER_EU, PICSONG_DWT_WAVES_PER_EU)))
#else
#define PS_DWT_OCC
#endif
template <typename T, bool LOSSY, bool U8IN, int BAND, bool VEC>
__global__ __launch_bounds__(256) PS_DWT_OCC void dwt_fwd_kernel(DwtFwdArgs a)
{
    constexpr int kFwdBandRows = BAND;
#ifdef PICSONG_DWT_CHUNKED
    constexpr int kFwdChunk = BAND / 2 < 4 ? BAND / 2 : 4;
#else
    // the whole band's input rows are fetched before the first store is issued (see below)
    constexpr int kFwdChunk = LOSSY ? BAND / 2 + 3 : BAND / 2;
#endif
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int strip = blockIdx.x * 4 + wave;
    if (strip * kStripUseful >= a.W) return;               // whole wave idle (no cross-lane use)
    dwt_fwd_select_frame(a);
    const int c0 = strip * kStripUseful - 4 * kEdgeLanes + 4 * lane;    // first of the lane's 4 columns
    const int m0 = a.pair_base + blockIdx.y * (kFwdBandRows / 2);
    int m1 = m0 + kFwdBandRows / 2;
    const int mend = a.pair_end > 0 ? a.pair_end : (a.H >> 1);
    if (m1 > mend) m1 = mend;
    const bool wr = lane >= kEdgeLanes && lane <= 63 - kEdgeLanes && c0 >= 0 && c0 < a.W;
    // VEC: every lane loads a whole in-image vector (out-of-image lanes a clamped one they never
    // use); the lanes owning columns 0 / W-4 mirror in registers (hfwd).  !VEC: per-column mirrors.
    const bool le = VEC && c0 == 0, re = VEC && c0 + 4 == a.W;
    const int cl = VEC ? (c0 < 0 ? 0 : (c0 > a.W - 4 ? a.W - 4 : c0)) : c0;
    const int pc = cl >> 1;                                  // (lanes with a clamped column never write)

    // All input rows of the band are fetched into raw registers before the first subband store is
    // issued (19 dwords for a 16-row u8 band, 44 for an 8-row int32 band).  vmcnt counts loads and
    // stores in one in-order queue, so a load issued after a store could only be waited for by
    // draining that store too; issued first, every wait is a counted vmcnt that leaves the stores in
    // flight.  (-DPICSONG_DWT_CHUNKED fetches 4 row pairs at a time, double-buffered, instead.)
    // The loads are issued at raised wave priority (s_setprio 3): a CU's memory instructions go
    // through one in-order pipe, and a time-resolved trace (s_memrealtime per wave) showed the median
    // wave needing 8.6 us just to ISSUE its 19 loads behind the stores of the waves that had started
    // earlier -- the store stream then thinned out into a 7 us tail.  With the loads first, level 0
    // of an 8K frame went from 35.7 to 27.6 us (6.0 TB/s).
    if constexpr (!LOSSY) {
        // vertical 5/3, DWTGenerator.cu:137-157: d[m] = x[2m+1] - ((x[2m]+x[2m+2])>>1);
        // s[m] = x[2m] + ((d[m-1]+d[m]+2)>>2)
        constexpr int NCH = (kFwdBandRows / 2 + kFwdChunk - 1) / kFwdChunk;
        T xe[4], xo[4], xn[4], dp[4];
        RawRow<U8IN> raw[2][2 * kFwdChunk];
        __builtin_amdgcn_s_setprio(3);          // see above: loads go first
        const RawRow<U8IN> r0 = load_raw<T, U8IN, VEC>(a, 2 * m0 - 2, cl);
        const RawRow<U8IN> r1 = load_raw<T, U8IN, VEC>(a, 2 * m0 - 1, cl);
        const RawRow<U8IN> r2 = load_raw<T, U8IN, VEC>(a, 2 * m0, cl);
#pragma unroll
        for (int r = 0; r < 2 * kFwdChunk; r++) raw[0][r] = load_raw<T, U8IN, VEC>(a, 2 * m0 + 1 + r, cl);
        __builtin_amdgcn_s_setprio(0);
        unpack_row<T, U8IN>(r0, xe);
        unpack_row<T, U8IN>(r1, xo);
        unpack_row<T, U8IN>(r2, xn);
#pragma unroll
        for (int k = 0; k < 4; k++) { dp[k] = xo[k] - ((xe[k] + xn[k]) >> 1); xe[k] = xn[k]; }
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            const int mc = m0 + c * kFwdChunk;
            if (c + 1 < NCH) {
#pragma unroll
                for (int r = 0; r < 2 * kFwdChunk; r++)
                    raw[(c + 1) & 1][r] = load_raw<T, U8IN, VEC>(a, 2 * (mc + kFwdChunk) + 1 + r, cl);
            }
#pragma unroll
            for (int q = 0; q < kFwdChunk; q++) {
                const int m = mc + q;
                unpack_row<T, U8IN>(raw[c & 1][2 * q], xo);
                unpack_row<T, U8IN>(raw[c & 1][2 * q + 1], xn);
                T Lr[4], Hr[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    T d = xo[k] - ((xe[k] + xn[k]) >> 1);
                    Lr[k] = xe[k] + ((dp[k] + d + 2) >> 2);
                    Hr[k] = d;
                    dp[k] = d; xe[k] = xn[k];
                }
                emit_pair<T, LOSSY, VEC>(a, m, pc, wr && m < m1, le, re, Lr, Hr);
            }
        }
    } else {
        // vertical 9/7, DWTGenerator.cu:184-227, streamed: at step j the pair j-1 completes
        constexpr int NCH = (kFwdBandRows / 2 + 3 + kFwdChunk - 1) / kFwdChunk;
        T xe[4], xo[4], xn[4], d1p[4], s1p[4], d2p[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { d1p[k] = s1p[k] = d2p[k] = (T)0; }
        RawRow<U8IN> raw[2][2 * kFwdChunk];
        __builtin_amdgcn_s_setprio(3);
        const RawRow<U8IN> r0 = load_raw<T, U8IN, VEC>(a, 2 * m0 - 4, cl);
#pragma unroll
        for (int r = 0; r < 2 * kFwdChunk; r++) raw[0][r] = load_raw<T, U8IN, VEC>(a, 2 * (m0 - 2) + 1 + r, cl);
        __builtin_amdgcn_s_setprio(0);
        unpack_row<T, U8IN>(r0, xe);
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            const int jc = m0 - 2 + c * kFwdChunk;
            if (c + 1 < NCH) {
#pragma unroll
                for (int r = 0; r < 2 * kFwdChunk; r++)
                    raw[(c + 1) & 1][r] = load_raw<T, U8IN, VEC>(a, 2 * (jc + kFwdChunk) + 1 + r, cl);
            }
#pragma unroll
            for (int q = 0; q < kFwdChunk; q++) {
                const int j = jc + q;
                unpack_row<T, U8IN>(raw[c & 1][2 * q], xo);
                unpack_row<T, U8IN>(raw[c & 1][2 * q + 1], xn);
                T Lr[4], Hr[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    float d1 = fmaf((float)xe[k] + (float)xn[k], PS_A1, (float)xo[k]);
                    float s1 = fmaf((float)d1p[k] + d1, PS_A2, (float)xe[k]);
                    float d2 = fmaf((float)s1p[k] + s1, PS_A3, (float)d1p[k]);
                    float s2 = fmaf((float)d2p[k] + d2, PS_A4, (float)s1p[k]);
                    Lr[k] = (T)(s2 * PS_N2);
                    Hr[k] = (T)(d2 * PS_N1);
                    d1p[k] = (T)d1; s1p[k] = (T)s1; d2p[k] = (T)d2; xe[k] = xn[k];
                }
                emit_pair<T, LOSSY, VEC>(a, j - 1, pc, wr && j - 1 >= m0 && j <= m1, le, re, Lr, Hr);
            }
        }
    }
}

// ---- levels 0 and 1 in one launch -------------------------------------------------------------
// The LL rows of level 0 never leave the registers: as the band streams down, every pair of level-0
// row pairs feeds one step of a second sliding window (the lane's two LL samples of a row are one
// (even, odd) pair of the level-1 row, so its horizontal lifting is the 2-wide form of hfwd over the
// same DPP neighbours).  Saves the 4 B/sample write and read of LL1 (67 MB of an 8K frame's 256 MB)
// and a launch.  Costs a taller run-in (level 1's run-in rows are level-0 rows that have to be
// recomputed: 9 extra input rows per 32-row band for 5/3, 21 for 9/7) and 3 recomputed lanes per side
// instead of 1 (the level-1 lifting reaches 2-3 lanes further).  8K: 27.4 + 13.4 -> 31.5 us (5/3),
// 29.0 + 16.6 -> 43.1 us (9/7).
// Iteration i does level-0 steps sA = S0 + 2i and sA + 1 (input rows 2sA+1 .. 2sA+4) and one level-1
// step on the two LL rows they deliver; 5/3 steps emit their own pair, 9/7 steps the pair before.
constexpr int kF2Edge = 3;
constexpr int kF2Useful = kStripCols - 8 * kF2Edge;          // 232 columns written per wave
#ifndef PICSONG_DWT_F2_WAVES
#define PICSONG_DWT_F2_WAVES 5      // 96 VGPRs: the 9/7 instantiation needs 98 without the cap (43.3 -> 41.3 us)
#endif
// level-1 row pairs per band (8 = 32 input rows), by transform: the 9/7 band's run-in is 21 input rows, the 5/3 band's 9
#ifndef PICSONG_DWT_F2_PAIRS
#define PICSONG_DWT_F2_PAIRS 8
#endif
#ifndef PICSONG_DWT_F2_PAIRS_LOSSY
#define PICSONG_DWT_F2_PAIRS_LOSSY 8
#endif
#ifndef PICSONG_DWT_F2_WAVES_LOSSY
#define PICSONG_DWT_F2_WAVES_LOSSY PICSONG_DWT_F2_WAVES
#endif
constexpr int kF2Pairs = PICSONG_DWT_F2_PAIRS, kF2PairsLossy = PICSONG_DWT_F2_PAIRS_LOSSY;
struct DwtFwd2Args { DwtFwdArgs l0, l1; };

__device__ __forceinline__ void hfwd2(int v[2], bool le, bool re)
{
    int en = nxt<int>(v[0], v[0], re);
    v[1] -= (v[0] + en) >> 1;
    int dp = prv<int>(v[1], v[1], le);
    v[0] += (dp + v[1] + 2) >> 2;
}
__device__ __forceinline__ void hfwd2(float v[2], bool le, bool re)
{
    float en = nxt<float>(v[0], v[0], re);
    v[1] = fmaf(v[0] + en, PS_A1, v[1]);
    float dp = prv<float>(v[1], v[1], le);
    v[0] = fmaf(v[1] + dp, PS_A2, v[0]);
    float sn = nxt<float>(v[0], v[0], re);
    v[1] = fmaf(v[0] + sn, PS_A3, v[1]);
    dp = prv<float>(v[1], v[1], le);
    v[0] = fmaf(v[1] + dp, PS_A4, v[0]) * PS_N2;
    v[1] *= PS_N1;
}

// One vertical analysis step over N columns: consumes rows xo = x[2j+1], xn = x[2j+2] (xe = x[2j] is
// state).  5/3 (st[0] = d[j-1]) delivers pair j, 9/7 (st = d1[j-1], s1[j-1], d2[j-2]) pair j-1.
template <typename T, bool LOSSY, int N>
__device__ __forceinline__ void vstep(T (&xe)[N], T (&st)[3][N], const T (&xo)[N], const T (&xn)[N], T (&L)[N], T (&H)[N])
{
#pragma unroll
    for (int k = 0; k < N; k++) {
        if constexpr (LOSSY) {
            float d1 = fmaf((float)xe[k] + (float)xn[k], PS_A1, (float)xo[k]);
            float s1 = fmaf((float)st[0][k] + d1, PS_A2, (float)xe[k]);
            float d2 = fmaf((float)st[1][k] + s1, PS_A3, (float)st[0][k]);
            float s2 = fmaf((float)st[2][k] + d2, PS_A4, (float)st[1][k]);
            L[k] = (T)(s2 * PS_N2);
            H[k] = (T)(d2 * PS_N1);
            st[0][k] = (T)d1; st[1][k] = (T)s1; st[2][k] = (T)d2;
        } else {
            T d = xo[k] - ((xe[k] + xn[k]) >> 1);
            L[k] = xe[k] + ((st[0][k] + d + 2) >> 2);
            H[k] = d;
            st[0][k] = d;
        }
        xe[k] = xn[k];
    }
}

// `row` = an LL row of the lane (2 samples) with index N + over in a band of N rows: for over >= 0 it
// is replaced by row N - 2 - over, which was delivered 2 * (over + 1) rows ago (hist[0] = the row
// before this one); then the history moves on.  `over` is wave-uniform.
template <typename T, int NH>
__device__ __forceinline__ void ll_row_or_mirror(T (&row)[2], T (&hist)[NH][2], int over)
{
    if (over >= 0) {
#pragma unroll
        for (int d = 0; d < NH / 2; d++)
            if (over == d) { row[0] = hist[2 * d + 1][0]; row[1] = hist[2 * d + 1][1]; }
    }
#pragma unroll
    for (int k = NH - 1; k > 0; k--) { hist[k][0] = hist[k - 1][0]; hist[k][1] = hist[k - 1][1]; }
    hist[0][0] = row[0]; hist[0][1] = row[1];
}

template <bool LOSSY, int NB> constexpr int f2_iters() { return NB + (LOSSY ? 5 : 2); }
#ifndef PICSONG_DWT_F2_RGB_AHEAD
#define PICSONG_DWT_F2_RGB_AHEAD 3
#endif
constexpr int kF2RgbAhead = PICSONG_DWT_F2_RGB_AHEAD;        // RGB head: iterations whose raw rows are in flight

// The same band, lean (round 2): the whole band unrolled, every row address a scalar offset into a buffer
// resource (RowBuf), store conditions resolved at compile time (bands are whole: plan_dwt_fwd2 takes the fused
// launch only when the level-1 row pairs are a multiple of NB, so pair `rel` of a band is stored iff
// 0 <= rel < 2 NB), lanes that own no output column parked on the dropped offset instead of an exec mask per
// store, level 1's bottom mirror looked at only in the iterations that can reach it.  u8 input.
// Measured and not kept: the band's rows staged through LDS by 16-byte-per-lane loads shared by the workgroup's
// four waves (a quarter of the load instructions, no prefetched rows in registers; one phase of 51 KB, or two of
// 28 KB with the second half's rows waiting in registers): 36-41 us against 35.6 for 9/7, 30.2-31.1 against 30.4
// for 5/3 -- the launch's first 6-7 us are the frame's 45-55 MB of input arriving at HBM speed whoever issues the
// loads, and the rest is its 135 MB of output leaving at the 6.3 TB/s the memory takes writes at.
// One row of component `comp` of the reversible colour transform from the rows' R, G, B dwords (four samples each):
// RGBTransformLossless Engines/CodingEngine.cu:357-403 with the level shift fused -- c0 = floor((R + 2G + B) / 4),
// c1 = B - G, c2 = R - G on the shifted samples -- as ONE linear form with wave-uniform coefficients,
// (cr R + cg G + cb B) >> sh, so that the component is no branch in the unrolled band.
struct RctCoef { int cr, cg, cb, sh; };
__device__ __forceinline__ RctCoef rct_coef(int comp)
{
    RctCoef k;
    k.cr = comp == 1 ? 0 : 1; k.cg = comp == 0 ? 2 : -1; k.cb = comp == 2 ? 0 : 1; k.sh = comp == 0 ? 2 : 0;
    return k;
}
__device__ __forceinline__ void unpack_rct(uint32_t wr, uint32_t wg, uint32_t wb, const RctCoef &k, int v[4])
{
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int r = (int)((wr >> (8 * i)) & 0xFFu) - 128, g = (int)((wg >> (8 * i)) & 0xFFu) - 128, b = (int)((wb >> (8 * i)) & 0xFFu) - 128;
        v[i] = (r * k.cr + g * k.cg + b * k.cb) >> k.sh;
    }
}
// ... and of the irreversible one (RGBTransformLossy Engines/CodingEngine.cu:384-449, level shift fused): row `comp` of
// the matrix rgb_forward_kernel<float> applies, in its operation order -- m2 b + (m1 g + (m0 r)), each step one fmaf --
// so the component planes the separate launch would have written never exist.  (The coefficients sit in vector
// registers: a multiply with a scalar operand issues at half rate.)
struct IctCoef { float m0, m1, m2; };
__device__ __forceinline__ IctCoef ict_coef(int comp)
{
    IctCoef k;
    k.m0 = comp == 0 ? 0.299f : (comp == 1 ? -0.168736f : 0.5f);
    k.m1 = comp == 0 ? 0.587f : (comp == 1 ? -0.331264f : -0.418688f);
    k.m2 = comp == 0 ? 0.114f : (comp == 1 ? 0.5f : -0.081312f);
    return k;
}
__device__ __forceinline__ void unpack_ict(uint32_t wr, uint32_t wg, uint32_t wb, const IctCoef &k, float v[4])
{
#pragma unroll
    for (int i = 0; i < 4; i++) {
        // (small integers: (float)(x - 128) and (float)x - 128.0f are the same value)
        const float r = (float)((wr >> (8 * i)) & 0xFFu) - 128.0f, g = (float)((wg >> (8 * i)) & 0xFFu) - 128.0f,
                    b = (float)((wb >> (8 * i)) & 0xFFu) - 128.0f;
        v[i] = fmaf(k.m2, b, fmaf(k.m1, g, k.m0 * r));
    }
}

template <typename T, bool LOSSY, int NB, bool EDGE, bool C16, bool RGB = false>
__device__ __forceinline__ void dwt_fwd2_band(const DwtFwdArgs &a, const DwtFwdArgs &a1, int strip, int lane, unsigned by, unsigned bz)
{
    static_assert(!RGB || (LOSSY ? std::is_same<T, float>::value : std::is_same<T, int>::value), "the colour transform in the head's load stage: RCT on integers, ICT on floats");
    constexpr uint32_t kCB = C16 ? 2u : 4u;                  // bytes of a coded coefficient
    constexpr int kIters = f2_iters<LOSSY, NB>();
    constexpr int kRel0 = LOSSY ? 7 : 3;                     // iteration i delivers the level-0 pairs 2 n0 + 2 i - kRel0, + 1
    constexpr int kLag1 = LOSSY ? 5 : 2;                     // ... and the level-1 pair n0 + i - kLag1
    constexpr int kHist = LOSSY ? 6 : 2;                     // LL rows kept for level 1's bottom mirror
    const int c0 = strip * kF2Useful - 4 * kF2Edge + 4 * lane;
    const int n0 = (int)by * NB;
    const bool wr = lane >= kF2Edge && lane <= 63 - kF2Edge && c0 >= 0 && c0 < a.W;
    const bool le = EDGE && c0 == 0, re = EDGE && c0 + 4 == a.W;
    const int cl = c0 < 0 ? 0 : (c0 > a.W - 4 ? a.W - 4 : c0);
    const int hW = a.W >> 1, hH = a.H >> 1, hW1 = a1.W >> 1, hH1 = a1.H >> 1;
    const bool lastb = 2 * (n0 + NB) >= a1.H;                // the band that ends at the bottom of the image
    const int y0 = 2 * (2 * n0 - (LOSSY ? 6 : 3));           // first input row (2 S0)

    const RowBuf in = rowbuf(a.src), mal = rowbuf(a.mallat), ll2 = rowbuf(a1.ll);
    const uint32_t vin = (uint32_t)cl;
    const uint32_t vlh = wr ? (uint32_t)(cl >> 1) * kCB : kRbDrop, vhl = wr ? (uint32_t)(hW + (cl >> 1)) * kCB : kRbDrop;
    const uint32_t vlh1 = wr ? (uint32_t)(cl >> 2) * kCB : kRbDrop, vhl1 = wr ? (uint32_t)(hW1 + (cl >> 2)) * kCB : kRbDrop;
    // (level 1's LL: 32-bit in the scratch, unless level 1 is the transform's last and its LL a coded subband)
    const bool ll16 = C16 && a1.last;
    const uint32_t vll1 = wr ? (uint32_t)(cl >> 2) * (ll16 ? 2u : 4u) : kRbDrop;
    const uint32_t aw4 = (uint32_t)a.AW * kCB, ll4 = ll16 ? aw4 : (uint32_t)a1.ll_stride * 4u;
    // level-1 LL: quantised on the transform's last level only; x * 1.0f * 1.0f is x
    const float qll = LOSSY && a1.last ? a1.q[0] : 1.0f, qsll = LOSSY && a1.last ? a1.qs : 1.0f;
    // level 0's steps and qs as vector registers: a multiply with a scalar-register operand issues at half rate
    // (tools/valu_probe), and there are 24 of them per iteration
    const float vq1 = in_vgpr(a.q[1]), vq2 = in_vgpr(a.q[2]), vq3 = in_vgpr(a.q[3]), vqs = in_vgpr(a.qs);

    T xe[4], st0[3][4], xe1[2], st1[3][2], hist[kHist][2];
#pragma unroll
    for (int k = 0; k < kHist; k++) { hist[k][0] = hist[k][1] = (T)0; }
#pragma unroll
    for (int k = 0; k < 4; k++) { st0[0][k] = st0[1][k] = st0[2][k] = (T)0; }
#pragma unroll
    for (int k = 0; k < 2; k++) { xe1[k] = st1[0][k] = st1[1][k] = st1[2][k] = (T)0; }

    // all of the band's input rows go out before the first store (one in-order memory pipe), at raised priority.
    // RGB: the rows of all three planes -- three times the registers -- so only kF2RgbAhead iterations' rows are in
    // flight, the rows of iteration i + kF2RgbAhead going out in iteration i BEFORE its stores (the synthesis kernels'
    // scheme): 36 raw registers instead of 3 x 4 x kIters, which is what lets that form run whole 32-row bands too.
    constexpr int kRing = RGB ? (kF2RgbAhead < kIters ? kF2RgbAhead : kIters) : kIters;
    RawRow<true> r0, raw[kRing][4];
    uint32_t g0 = 0u, b0 = 0u, rawG[RGB ? kRing : 1][4], rawB[RGB ? kRing : 1][4];
    const RowBuf ing = rowbuf(RGB ? a.src_g : a.src), inb = rowbuf(RGB ? a.src_b : a.src);
    const RctCoef rct = rct_coef(RGB ? (int)bz : 0);
    IctCoef ict = ict_coef(RGB ? (int)bz : 0);
    ict.m0 = in_vgpr(ict.m0); ict.m1 = in_vgpr(ict.m1); ict.m2 = in_vgpr(ict.m2);
    auto load_iter = [&](int p) {                            // (p: a compile-time value after unrolling)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t ro = (uint32_t)reflect(y0 + 1 + 4 * p + q, a.H) * (uint32_t)a.src_stride;
            raw[p % kRing][q].w = rb_load32(in, vin, ro);
            if constexpr (RGB) { rawG[p % kRing][q] = rb_load32(ing, vin, ro); rawB[p % kRing][q] = rb_load32(inb, vin, ro); }
        }
    };
    auto unpack_rgb = [&](uint32_t wr_, uint32_t wg_, uint32_t wb_, T v[4]) {
        if constexpr (LOSSY) unpack_ict(wr_, wg_, wb_, ict, v);
        else unpack_rct(wr_, wg_, wb_, rct, v);
    };
    PS_TRACE(0);
    __builtin_amdgcn_s_setprio(3);
    r0.w = rb_load32(in, vin, (uint32_t)reflect(y0, a.H) * (uint32_t)a.src_stride);
    if constexpr (RGB) {
        g0 = rb_load32(ing, vin, (uint32_t)reflect(y0, a.H) * (uint32_t)a.src_stride);
        b0 = rb_load32(inb, vin, (uint32_t)reflect(y0, a.H) * (uint32_t)a.src_stride);
    }
#pragma unroll
    for (int p = 0; p < kRing; p++) load_iter(p);
    __builtin_amdgcn_s_setprio(0);
    PS_TRACE(1);
    if constexpr (RGB) unpack_rgb(r0.w, g0, b0, xe);
    else unpack_row<T, true>(r0, xe);
    PS_TRACE(2);

#pragma unroll
    for (int i = 0; i < kIters; i++) {
        if (i == kIters / 2) PS_TRACE(3);
        T x[4][4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if constexpr (RGB) unpack_rgb(raw[i % kRing][q].w, rawG[i % kRing][q], rawB[i % kRing][q], x[q]);
            else unpack_row<T, true>(raw[i % kRing][q], x[q]);
        }
        if (i + kRing < kIters) load_iter(i + kRing);        // (RGB only: the ring's next rows, ahead of this iteration's stores)
        const int rel = 2 * i - kRel0;                       // (compile-time after unrolling)
        T LA[4], HA[4], LB[4], HB[4];
        vstep<T, LOSSY, 4>(xe, st0, x[0], x[1], LA, HA);
        vstep<T, LOSSY, 4>(xe, st0, x[2], x[3], LB, HB);
#pragma unroll
        for (int h = 0; h < 2; h++) {                        // the two level-0 pairs of this iteration
            T *Lr = h ? LB : LA, *Hr = h ? HB : HA;
            const int r = rel + h;
            hfwd(Lr, le, re);
            if (r >= 0 && r < 2 * NB) {
                hfwd(Hr, le, re);
                T hl0 = Lr[1], hl1 = Lr[3], lh0 = Hr[0], lh1 = Hr[2], hh0 = Hr[1], hh1 = Hr[3];
                if (LOSSY) {
                    hl0 = (T)(((float)hl0 * vq1) * vqs); hl1 = (T)(((float)hl1 * vq1) * vqs);
                    lh0 = (T)(((float)lh0 * vq2) * vqs); lh1 = (T)(((float)lh1 * vq2) * vqs);
                    hh0 = (T)(((float)hh0 * vq3) * vqs); hh1 = (T)(((float)hh1 * vq3) * vqs);
                }
                const uint32_t row0 = (uint32_t)(2 * n0 + r) * aw4, row1 = (uint32_t)(2 * n0 + r + hH) * aw4;
                if constexpr (C16) {
                    rb_store32(mal, vhl, row0, pack_c16(hl0, hl1));
                    rb_store32(mal, vlh, row1, pack_c16(lh0, lh1));
                    rb_store32(mal, vhl, row1, pack_c16(hh0, hh1));
                } else {
                    rb_store64(mal, vhl, row0, as_u32(hl0), as_u32(hl1));
                    rb_store64(mal, vlh, row1, as_u32(lh0), as_u32(lh1));
                    rb_store64(mal, vhl, row1, as_u32(hh0), as_u32(hh1));
                }
            }
        }
        // level 1: LL rows 2 n0 + rel (odd row of its pair) and + 1 (the even row after it).  Past the bottom
        // of the image they are level 1's OWN mirror, LL[N + k] = LL[N - 2 - k]: mirrored input rows do not give
        // that (the input's mirror centre H - 1 is an odd row, so the even-row subsequence comes out half-sample
        // symmetric) -- taken from the rows kept in `hist`; only the last band's last iterations get there.
        T la[2] = { LA[0], LA[2] }, lb[2] = { LB[0], LB[2] };
        ll_row_or_mirror<T, kHist>(la, hist, rel >= 2 * NB && lastb ? rel - 2 * NB : -1);
        ll_row_or_mirror<T, kHist>(lb, hist, rel + 1 >= 2 * NB && lastb ? rel + 1 - 2 * NB : -1);
        T L1[2], H1[2];
        vstep<T, LOSSY, 2>(xe1, st1, la, lb, L1, H1);
        if (i >= kLag1) {
            hfwd2(L1, le, re);
            hfwd2(H1, le, re);
            T ll = L1[0], hl = L1[1], lh = H1[0], hh = H1[1];
            if (LOSSY) {
                ll = (T)(((float)ll * qll) * qsll);
                hl = (T)(((float)hl * a1.q[1]) * vqs);      // (a1.qs == a.qs: one transform, one qs)
                lh = (T)(((float)lh * a1.q[2]) * vqs);
                hh = (T)(((float)hh * a1.q[3]) * vqs);
            }
            const uint32_t n = (uint32_t)(n0 + i - kLag1);
            if constexpr (C16) {
                if (ll16) rb_store16(ll2, vll1, n * ll4, pack_c16(ll, ll));
                else rb_store32(ll2, vll1, n * ll4, as_u32(ll));
                rb_store16(mal, vhl1, n * aw4, pack_c16(hl, hl));
                rb_store16(mal, vlh1, (n + (uint32_t)hH1) * aw4, pack_c16(lh, lh));
                rb_store16(mal, vhl1, (n + (uint32_t)hH1) * aw4, pack_c16(hh, hh));
            } else {
                rb_store32(ll2, vll1, n * ll4, as_u32(ll));
                rb_store32(mal, vhl1, n * aw4, as_u32(hl));
                rb_store32(mal, vlh1, (n + (uint32_t)hH1) * aw4, as_u32(lh));
                rb_store32(mal, vhl1, (n + (uint32_t)hH1) * aw4, as_u32(hh));
            }
        }
    }
    PS_TRACE(4);
#ifdef PICSONG_DWT_TRACE
    PS_TRACE_WAIT();
    PS_TRACE(5);
#endif
}

// RGB: the colour transform in the load stage (blockIdx.z = component)
#ifndef PICSONG_DWT_F2_RGB_PAIRS
#define PICSONG_DWT_F2_RGB_PAIRS 8
#endif
#ifndef PICSONG_DWT_F2_RGB_WAVES
#define PICSONG_DWT_F2_RGB_WAVES 4
#endif
constexpr int kF2PairsRgb = PICSONG_DWT_F2_RGB_PAIRS;
template <typename T, bool LOSSY, bool U8IN, int NB, bool C16 = false, bool RGB = false>
__global__ __launch_bounds__(256, RGB ? PICSONG_DWT_F2_RGB_WAVES : (LOSSY ? PICSONG_DWT_F2_WAVES_LOSSY : PICSONG_DWT_F2_WAVES)) void dwt_fwd2_kernel(DwtFwd2Args a2)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if constexpr (RGB) {
        // The three components of a tile read the same rows of the same three planes.  Workgroups go to the eight XCDs
        // round robin by their linear index, each XCD with an L2 of its own: in launch order (component = blockIdx.z,
        // the slowest) a tile's three workgroups run a third of the launch apart, wherever their indices fall, and
        // every plane comes from the memory side three times.  Re-indexed so that they are 8 apart -- same XCD, in
        // flight together -- the second and third read hit that XCD's L2 (8K: head 118.6 -> see profiles/NOTES.md).
        const unsigned tiles = gridDim.x * gridDim.y;
        if ((tiles & 7u) == 0u && gridDim.z == 3u) {
            const unsigned L = bx + gridDim.x * (by + gridDim.y * bz);
            const unsigned tile = (L / 24u) * 8u + (L & 7u);
            bz = (L >> 3) % 3u;
            bx = tile % gridDim.x; by = tile / gridDim.x;
        }
    }
    const int strip = (int)bx * 4 + wave;
    dwt_fwd_select_frame(a2.l0, bz);
    dwt_fwd_select_frame(a2.l1, bz);
    // the wave's 256 columns start at strip * kF2Useful - 4 * kF2Edge: does it hold column 0 or W - 4?
    const int first = strip * kF2Useful - 4 * kF2Edge;
    // (only the 9/7 kernel, which is bound by vector instructions, gets the second instantiation)
    static_assert(U8IN, "the fused head ingests u8 frames");
    if (strip * kF2Useful >= a2.l0.W) return;               // whole wave idle (no cross-lane use)
    if (!LOSSY || first <= 0 || first + kStripCols >= a2.l0.W)
        dwt_fwd2_band<T, LOSSY, NB, true, C16, RGB>(a2.l0, a2.l1, strip, lane, by, bz);
    else dwt_fwd2_band<T, LOSSY, NB, LOSSY ? false : true, C16, RGB>(a2.l0, a2.l1, strip, lane, by, bz);
}

// ---- the small levels of the forward transform ---------------------------------------------------------
// Past level 2 of an 8K frame a level is 2 MB and less and a launch costs 4-5 us, most of it the dependency drain
// and one wave's serial chain.  Measured in round 2 and not kept: ONE launch for levels 3..5 (3..4 at wl = 5) --
// 64 x 32-sample tiles with the 32 (16)-sample halo three (two) levels of lifting reach, all levels in LDS in
// place at stride 2^j, a thread lifting a segment of 8 >> j pairs in registers between two barriers, level 1's
// and 2's mirrors re-made from the tile's own LL samples; bit-identical on the emulator and on the GPU.  9.9 us
// against 3 x 4.7 (9/7), 6.6 against 2 x 5 (5/3): the input's round trip to memory 2.7 us, level 0's two passes
// 2.3, emit + mirror 1.1, level 1 1.9, level 2 1.1 (time-resolved trace).  A lone frame gained 2-4 us of 55, and
// frames in flight LOST 9-14 % (137.6 against 150.9 Gpixel/s at 8K, 126.6 against 147.5 at 4K): a 1024-thread
// workgroup with 50 KB of LDS waits for a whole CU's worth of resources among the coder's waves.  The levels
// stay one launch each.

// ---- inverse --------------------------------------------------------------------------------
// readSubbands* DWTGenerator.cu:477-553: de-quantisation (|v| + 0.5) * sgn(v) / Q / qs, 0 -> 0
template <bool FAST>
__device__ __forceinline__ float dequant(int32_t v, float q, float rq, float qs, float rqs)
{
    if (v == 0) return 0.0f;
    float m = fabsf((float)v) + 0.5f;
    float s = v < 0 ? -1.0f : 1.0f;
    if constexpr (FAST) {
        // the domain dequant_fast_ok has verified for this qs: 16 bit-planes, what the coder delivers
        if ((uint32_t)(v + 65535) <= 131070u) return div_rc(div_rc(m * s, q, rq), qs, rqs);
        keep_in_branch(m);
    }
    return ((m * s) / q) / qs;
}

// One subband row pair-segment as it comes from memory: d-type values (HL / HH) for pair columns pc,
// pc+1 and s-type values (LL / LH) likewise; conversion (de-quantisation) is deferred so that the
// loads of several row pairs can be in flight.
struct SubRaw { uint32_t d0, d1, s0, s1; };

// C16 (vector kernels only): the coded coefficients are an int16 Mallat array; a pair of them stays PACKED in d0 (and
// in s0 when the s-type values are coded ones too: LH, or the coarsest level's LL) until convert_sub unpacks it -- an
// unpack at the load would put a wait for the load right behind it
template <typename T, bool VEC, bool C16 = false>
__device__ __forceinline__ SubRaw load_sub_raw(const DwtInvArgs &a, int row_s_or_d, bool high_row, int pc, bool inside)
{
    // row index already reflected by the caller.  low row: s = LL, d = HL; high row: s = LH, d = HH
    const int hW = a.W >> 1, hH = a.H >> 1;
    if constexpr (C16) {
        static_assert(VEC, "the 16-bit coefficient form exists in the vector kernels");
        const int16_t *mrow = reinterpret_cast<const int16_t *>(a.mallat) + (size_t)(row_s_or_d + (high_row ? hH : 0)) * (size_t)a.AW;
        SubRaw r;
        r.d0 = *reinterpret_cast<const uint32_t *>(mrow + hW + pc); r.d1 = 0u;
        if (high_row || a.first) { r.s0 = *reinterpret_cast<const uint32_t *>(mrow + pc); r.s1 = 0u; }
        else {
            const uint2 sv = *reinterpret_cast<const uint2 *>((const uint32_t *)a.ll + (size_t)row_s_or_d * (size_t)a.ll_stride + pc);
            r.s0 = sv.x; r.s1 = sv.y;
        }
        return r;
    }
    const int32_t *mrow = a.mallat + (size_t)(row_s_or_d + (high_row ? hH : 0)) * (size_t)a.AW;
    const bool from_mallat = high_row || a.first;
    const uint32_t *srow = from_mallat ? (const uint32_t *)mrow
                                       : (const uint32_t *)a.ll + (size_t)row_s_or_d * (size_t)a.ll_stride;
    SubRaw r;
    if constexpr (VEC) {
        // pc is even and already clamped into [0, hW-2]: two aligned 8-byte loads per subband row
        const uint2 d = *reinterpret_cast<const uint2 *>(mrow + hW + pc);
        const uint2 sv = *reinterpret_cast<const uint2 *>(srow + pc);
        r.d0 = d.x; r.d1 = d.y; r.s0 = sv.x; r.s1 = sv.y;
        return r;
    }
    int cs0 = pc, cs1 = pc + 1, cd0 = pc, cd1 = pc + 1;
    if (!inside) {
        cs0 = reflect_s(pc, hW); cs1 = reflect_s(pc + 1, hW);
        cd0 = reflect_d(pc, hW); cd1 = reflect_d(pc + 1, hW);
    }
    r.d0 = (uint32_t)mrow[hW + cd0]; r.d1 = (uint32_t)mrow[hW + cd1];
    r.s0 = srow[cs0]; r.s1 = srow[cs1];
    return r;
}

// (s0, d0, s1, d1) of the lane's two pairs as samples: coded coefficients are int32 (de-quantised when
// lossy), the previous level's LL is already T
template <typename T, bool LOSSY, bool FAST, bool C16 = false>
__device__ __forceinline__ void convert_sub(const DwtInvArgs &a, const SubRaw &r0, bool high_row, T v[4])
{
    const float qd = high_row ? a.q[3] : a.q[1], rqd = high_row ? a.rq[3] : a.rq[1];
    const float qsb = high_row ? a.q[2] : a.q[0], rqsb = high_row ? a.rq[2] : a.rq[0];
    SubRaw r = r0;
    if constexpr (C16) {                                     // the packed pairs of load_sub_raw<.., C16>
        r.d1 = (uint32_t)c16_hi(r0.d0); r.d0 = (uint32_t)c16_lo(r0.d0);
        if (high_row || a.first) { r.s1 = (uint32_t)c16_hi(r0.s0); r.s0 = (uint32_t)c16_lo(r0.s0); }
    }
    if (LOSSY) { v[1] = (T)dequant<FAST>((int32_t)r.d0, qd, rqd, a.qs, a.rqs); v[3] = (T)dequant<FAST>((int32_t)r.d1, qd, rqd, a.qs, a.rqs); }
    else { v[1] = (T)(int32_t)r.d0; v[3] = (T)(int32_t)r.d1; }
    if (high_row || a.first) {
        if (LOSSY) { v[0] = (T)dequant<FAST>((int32_t)r.s0, qsb, rqsb, a.qs, a.rqs); v[2] = (T)dequant<FAST>((int32_t)r.s1, qsb, rqsb, a.qs, a.rqs); }
        else { v[0] = (T)(int32_t)r.s0; v[2] = (T)(int32_t)r.s1; }
    } else { v[0] = from_u32<T>(r.s0); v[2] = from_u32<T>(r.s1); }
}

template <typename T, bool VEC>
__device__ __forceinline__ void store_row4(const DwtInvArgs &a, int y, int c0, const T v[4])
{
    T *p = (T *)a.dst + (size_t)y * (size_t)a.W + c0;
    if constexpr (VEC) {
        uint4 w;
        w.x = as_u32(v[0]); w.y = as_u32(v[1]); w.z = as_u32(v[2]); w.w = as_u32(v[3]);
        *reinterpret_cast<uint4 *>(p) = w;
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++) if (c0 + k < a.W) p[k] = v[k];
    }
}

// removeOffsetAndApplyMaxMin / ...Lossy (Engines/DecodingEngine.cu:706-729) + the u8 conversion of
// IOManager::writeImage (IO/IOManager.ipp:332-335) fused into the finest level's store: 4 pixels per
// lane, one dword.  Same arithmetic as clamp_to_u8_i32_kernel / clamp_to_u8_f32_kernel.
__device__ __forceinline__ uint32_t to_pixel(int v, int off)
{
    const int t = v + off;
    return (uint32_t)(t > 255 ? 255 : (t < 0 ? 0 : t));
}
__device__ __forceinline__ uint32_t to_pixel(float v, int off)
{
    float t = v + (float)off;
    t = t + 0.01f;
    float r = rintf(t);                      // __float2int_rn
    r = r > 255.0f ? 255.0f : r;
    r = r < 0.0f ? 0.0f : r;
    return (uint32_t)(int)r;
}
// four float samples to one dword of pixels, to_pixel's arithmetic: the two roundings of v + off and + 0.01, round
// to nearest even, then v_cvt_pk_u8_f32 -- the conversion saturates to 0..255 (NaN: 0) and drops the byte into
// place, where the compare / select clamp, the integer conversion and the shift-or cost five instructions more
// per pixel.  (tests/test_gpu_parity.py::test_lossy_pixels_clamp_like_the_oracle holds the hardware to that.)
__device__ __forceinline__ uint32_t pack_pixels(const float v[4], float foff)
{
#if defined(__AMDGCN__)
    uint32_t w = 0u;
#pragma unroll
    for (int i = 0; i < 4; i++) w = __builtin_amdgcn_cvt_pk_u8_f32(rintf((v[i] + foff) + 0.01f), (uint32_t)i, w);
    return w;
#else
    uint32_t w = 0u;
    for (int i = 0; i < 4; i++) {
        float r = rintf((v[i] + foff) + 0.01f);
        r = r > 255.0f ? 255.0f : r;
        r = r < 0.0f ? 0.0f : r;                 // (NaN stays NaN through both and converts to 0 on the GPU)
        w |= (r == r ? (uint32_t)(int)r : 0u) << (8 * i);
    }
    return w;
#endif
}
template <typename T>
__device__ __forceinline__ void store_row4_u8(const DwtInvArgs &a, int y, int c0, const T v[4])
{
    const uint32_t w = to_pixel(v[0], a.off) | (to_pixel(v[1], a.off) << 8) | (to_pixel(v[2], a.off) << 16) |
                       (to_pixel(v[3], a.off) << 24);
    *reinterpret_cast<uint32_t *>(a.dst_u8 + (size_t)y * (size_t)a.W + (uint32_t)c0) = w;
}

template <typename T, bool LOSSY, int BAND, bool VEC, bool U8OUT = false, bool FAST = false, bool C16 = false>
__global__ __launch_bounds__(256) void dwt_inv_kernel(DwtInvArgs a)
{
    constexpr int kInvBandRows = BAND;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int strip = blockIdx.x * 4 + wave;
    if (strip * kStripUseful >= a.W) return;
    dwt_inv_select_frame(a);
    const int c0 = strip * kStripUseful - 4 * kEdgeLanes + 4 * lane;
    const int pc = c0 >> 1;                                  // arithmetic shift: -4 -> -2
    const int hW = a.W >> 1, hH = a.H >> 1;
    const int m0 = blockIdx.y * (kInvBandRows / 2);
    int m1 = m0 + kInvBandRows / 2;
    if (m1 > hH) m1 = hH;
    const bool inside = pc >= 0 && pc + 1 < hW;
    const bool wr = lane >= kEdgeLanes && lane <= 63 - kEdgeLanes && c0 >= 0 && c0 < a.W;
    const bool le = VEC && c0 == 0, re = VEC && c0 + 4 == a.W;
    const int pl = VEC ? (pc < 0 ? 0 : (pc > hW - 2 ? hW - 2 : pc)) : pc;

    // The band's row pairs plus the filter's run-in: a compile-time trip count (rows past the band's
    // end at the bottom of the image are mirrored reads whose results are not stored), so the loop
    // unrolls and the raw loads of the next kInvAhead iterations are issued BEFORE this iteration's
    // stores (one in-order memory pipe, and the compiler cannot prove dst != src).  5/3: fully
    // unrolled.  9/7: a counted loop over groups of kInvAhead unrolled iterations (kInvAhead divides the
    // trip count, so there is no remainder loop): unrolled 12 times the 16-row kernel was 77 KB of
    // code for a 64 KB instruction cache.
    constexpr int kRunIn = LOSSY ? 2 : 1;
    constexpr int kIters = kInvBandRows / 2 + 2 * kRunIn;
    constexpr int kInvAhead = !LOSSY ? (kIters < PICSONG_DWT_INV_AHEAD ? kIters : PICSONG_DWT_INV_AHEAD)
                              : inv_group(kIters, PICSONG_DWT_INV_GROUP);
    static_assert(!LOSSY || kIters % kInvAhead == 0, "9/7: groups of kInvAhead iterations");
    const int j0 = m0 - kRunIn;
    SubRaw rawL[kInvAhead], rawH[kInvAhead];
    __builtin_amdgcn_s_setprio(3);
#pragma unroll
    for (int p = 0; p < kInvAhead; p++) {
        rawL[p] = load_sub_raw<T, VEC, C16>(a, reflect_s(j0 + p, hH), false, pl, inside);
        rawH[p] = load_sub_raw<T, VEC, C16>(a, reflect_d(j0 + p, hH), true, pl, inside);
    }
    __builtin_amdgcn_s_setprio(0);

    if constexpr (!LOSSY) {
        // vertical 5/3 synthesis, DWTGenerator.cu:160-181, streamed: at step j pair j-1 completes
        T Hp[4], sp[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { Hp[k] = sp[k] = 0; }
#pragma unroll
        for (int it = 0; it < kIters; it++) {
            const int j = j0 + it;
            T Lr[4], Hr[4];
            convert_sub<T, LOSSY, FAST, C16>(a, rawL[it % kInvAhead], false, Lr);
            convert_sub<T, LOSSY, FAST, C16>(a, rawH[it % kInvAhead], true, Hr);
            if (it + kInvAhead < kIters) {
                rawL[it % kInvAhead] = load_sub_raw<T, VEC, C16>(a, reflect_s(j + kInvAhead, hH), false, pl, inside);
                rawH[it % kInvAhead] = load_sub_raw<T, VEC, C16>(a, reflect_d(j + kInvAhead, hH), true, pl, inside);
            }
            hinv<FAST>(Lr, le, re);
            hinv<FAST>(Hr, le, re);
            T ev[4], od[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                T s = Lr[k] - ((Hp[k] + Hr[k] + 2) >> 2);
                ev[k] = sp[k];
                od[k] = Hp[k] + ((sp[k] + s) >> 1);
                sp[k] = s; Hp[k] = Hr[k];
            }
            if (it >= 2 && j - 1 < m1 && wr) {
                if constexpr (U8OUT) {
                    store_row4_u8<T>(a, 2 * (j - 1), c0, ev);
                    store_row4_u8<T>(a, 2 * (j - 1) + 1, c0, od);
                } else {
                    store_row4<T, VEC>(a, 2 * (j - 1), c0, ev);
                    store_row4<T, VEC>(a, 2 * (j - 1) + 1, c0, od);
                }
            }
        }
    } else {
        // vertical 9/7 synthesis, DWTGenerator.cu:230-272, streamed: at step j pair j-2 completes
        T ddp[4], s1p[4], d1p[4], s0p[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { ddp[k] = s1p[k] = d1p[k] = s0p[k] = (T)0; }
#pragma unroll 1
        for (int g = 0; g < kIters / kInvAhead; g++) {
#pragma unroll
        for (int r = 0; r < kInvAhead; r++) {
            const int it = g * kInvAhead + r;
            const int j = j0 + it;
            T Lr[4], Hr[4];
            convert_sub<T, LOSSY, FAST, C16>(a, rawL[r], false, Lr);
            convert_sub<T, LOSSY, FAST, C16>(a, rawH[r], true, Hr);
            if (g + 1 < kIters / kInvAhead) {
                rawL[r] = load_sub_raw<T, VEC, C16>(a, reflect_s(j + kInvAhead, hH), false, pl, inside);
                rawH[r] = load_sub_raw<T, VEC, C16>(a, reflect_d(j + kInvAhead, hH), true, pl, inside);
            }
            hinv<FAST>(Lr, le, re);
            hinv<FAST>(Hr, le, re);
            T ev[4], od[4];
            float hn[4], ln[4];
#pragma unroll
            for (int k = 0; k < 4; k++) { hn[k] = (float)Hr[k]; ln[k] = (float)Lr[k]; }
            div_n1n2<FAST>(hn[0], hn[1], ln[0], ln[1]);
            div_n1n2<FAST>(hn[2], hn[3], ln[2], ln[3]);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                float dd = hn[k];
                float s1 = fmaf(-((float)ddp[k] + dd), PS_A4, ln[k]);
                float d1 = fmaf(-((float)s1p[k] + s1), PS_A3, (float)ddp[k]);      // d1[j-1]
                float s0 = fmaf(-((float)d1p[k] + d1), PS_A2, (float)s1p[k]);      // s0[j-1]
                float xo = fmaf(-((float)s0p[k] + s0), PS_A1, (float)d1p[k]);      // x[2(j-2)+1]
                ev[k] = s0p[k];
                od[k] = (T)xo;
                ddp[k] = (T)dd; s1p[k] = (T)s1; d1p[k] = (T)d1; s0p[k] = (T)s0;
            }
            if (it >= 4 && j - 2 < m1 && wr) {
                if constexpr (U8OUT) {
                    store_row4_u8<T>(a, 2 * (j - 2), c0, ev);
                    store_row4_u8<T>(a, 2 * (j - 2) + 1, c0, od);
                } else {
                    store_row4<T, VEC>(a, 2 * (j - 2), c0, ev);
                    store_row4<T, VEC>(a, 2 * (j - 2) + 1, c0, od);
                }
            }
        }
        }
    }
}

// ---- RGB frames, 5/3: the finest synthesis level of all three components and the inverse colour transform -----
// picsong_decode_rgb_frame's last two launches were the finest level of the three components (grid.z = component:
// 3 x 67 MB of int16 coefficients in, 3 x 134 MB of 32-bit planes out) and rgb_inverse_kernel (those planes in, 100 MB
// of pixels out).  Here ONE wave runs the level for the same columns of all three components -- three copies of the
// streamed vertical state, the rows of kInvRgbAhead iterations in flight per component -- and a finished row pair
// goes through the inverse RCT (RGBTransformLossless's inverse, Engines/DecodingEngine.cu:599-626: G = Y - ((Cb + Cr)
// >> 2), R = Cr + G, B = Cb + G), the level shift and the clamp on its way out: the planes are never written.
// a: the finest level's arguments as the plan made them for a three-frame launch (component c's coefficients at
// mallat + c * mallat_z, its LL at ll + c * ll_z); dr / dg / db: the pixel planes, row stride W.
#ifndef PICSONG_DWT_INV_RGB_AHEAD
#define PICSONG_DWT_INV_RGB_AHEAD 3
#endif
__device__ __forceinline__ void store_rgb_row4(uint8_t *dr, uint8_t *dg, uint8_t *db, size_t at, int off, const int (&v)[3][4])
{
    uint32_t wr_ = 0u, wg_ = 0u, wb_ = 0u;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int G = v[0][k] - ((v[1][k] + v[2][k]) >> 2), R = v[2][k] + G, B = v[1][k] + G;
        wr_ |= to_pixel(R, off) << (8 * k); wg_ |= to_pixel(G, off) << (8 * k); wb_ |= to_pixel(B, off) << (8 * k);
    }
    *reinterpret_cast<uint32_t *>(dr + at) = wr_;
    *reinterpret_cast<uint32_t *>(dg + at) = wg_;
    *reinterpret_cast<uint32_t *>(db + at) = wb_;
}
#ifndef PICSONG_DWT_INV_RGB_WAVES
#define PICSONG_DWT_INV_RGB_WAVES 3
#endif
template <int BAND>
__global__ __launch_bounds__(256, PICSONG_DWT_INV_RGB_WAVES) void dwt_inv_rgb_kernel(DwtInvArgs a, uint8_t *dr, uint8_t *dg, uint8_t *db)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int strip = blockIdx.x * 4 + wave;
    if (strip * kStripUseful >= a.W) return;
    const int c0 = strip * kStripUseful - 4 * kEdgeLanes + 4 * lane;
    const int pc = c0 >> 1;
    const int hW = a.W >> 1, hH = a.H >> 1;
    const int m0 = blockIdx.y * (BAND / 2);
    int m1 = m0 + BAND / 2;
    if (m1 > hH) m1 = hH;
    const bool inside = pc >= 0 && pc + 1 < hW;
    const bool wr = lane >= kEdgeLanes && lane <= 63 - kEdgeLanes && c0 >= 0 && c0 < a.W;
    const bool le = c0 == 0, re = c0 + 4 == a.W;
    const int pl = pc < 0 ? 0 : (pc > hW - 2 ? hW - 2 : pc);
    DwtInvArgs ac[3] = { a, a, a };
#pragma unroll
    for (int c = 1; c < 3; c++) {
        ac[c].mallat = (const int32_t *)((const char *)a.mallat + (unsigned long long)c * a.mallat_z);
        ac[c].ll = (const char *)a.ll + (unsigned long long)c * a.ll_z;
    }
    constexpr int kIters = BAND / 2 + 2;
    constexpr int kAhead = kIters < PICSONG_DWT_INV_RGB_AHEAD ? kIters : PICSONG_DWT_INV_RGB_AHEAD;
    const int j0 = m0 - 1;
    SubRaw rawL[3][kAhead], rawH[3][kAhead];
    __builtin_amdgcn_s_setprio(3);
#pragma unroll
    for (int p = 0; p < kAhead; p++)
#pragma unroll
        for (int c = 0; c < 3; c++) {
            rawL[c][p] = load_sub_raw<int, true, true>(ac[c], reflect_s(j0 + p, hH), false, pl, inside);
            rawH[c][p] = load_sub_raw<int, true, true>(ac[c], reflect_d(j0 + p, hH), true, pl, inside);
        }
    __builtin_amdgcn_s_setprio(0);
    int Hp[3][4], sp[3][4];
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int k = 0; k < 4; k++) { Hp[c][k] = sp[c][k] = 0; }
#pragma unroll
    for (int it = 0; it < kIters; it++) {
        const int j = j0 + it;
        int ev[3][4], od[3][4];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            int Lr[4], Hr[4];
            convert_sub<int, false, false, true>(ac[c], rawL[c][it % kAhead], false, Lr);
            convert_sub<int, false, false, true>(ac[c], rawH[c][it % kAhead], true, Hr);
            if (it + kAhead < kIters) {
                rawL[c][it % kAhead] = load_sub_raw<int, true, true>(ac[c], reflect_s(j + kAhead, hH), false, pl, inside);
                rawH[c][it % kAhead] = load_sub_raw<int, true, true>(ac[c], reflect_d(j + kAhead, hH), true, pl, inside);
            }
            hinv<false>(Lr, le, re);
            hinv<false>(Hr, le, re);
#pragma unroll
            for (int k = 0; k < 4; k++) {                    // vertical 5/3 synthesis, as dwt_inv_kernel
                const int s_ = Lr[k] - ((Hp[c][k] + Hr[k] + 2) >> 2);
                ev[c][k] = sp[c][k];
                od[c][k] = Hp[c][k] + ((sp[c][k] + s_) >> 1);
                sp[c][k] = s_; Hp[c][k] = Hr[k];
            }
        }
        if (it >= 2 && j - 1 < m1 && wr) {
            const size_t at = (size_t)(2 * (j - 1)) * (size_t)a.W + (size_t)(uint32_t)c0;
            store_rgb_row4(dr, dg, db, at, a.off, ev);
            store_rgb_row4(dr, dg, db, at + (size_t)a.W, a.off, od);
        }
    }
}

// ---- 9/7 synthesis of one level, lean (round 2) ---------------------------------------------------
// The vector launches of a context whose reciprocal divisions verified (InvLaunch::fast).  Same strips, bands and
// streamed vertical synthesis as dwt_inv_kernel, same arithmetic value for value; what is gone is everything that
// is not arithmetic (the kernel is bound by vector-instruction issue and by its waves' dependent chains: 54 us for
// level 0 of an 8K frame in dwt_inv_kernel's FAST form against 33 us for the 5/3 kernel over the same bytes):
//  * de-quantisation without control flow.  (|v| + 0.5) * sgn(v) is v + clamp(v, -0.5, 0.5) exactly (v_cvt, v_med3_f32,
//    v_add), 0 stays 0.  When qs is a power of two the second division is exact scaling and folds into the step
//    (ONE_DIV).  dequant()'s zero and domain tests were two exec-mask regions per coefficient;
//  * no branch in the band's loop but the (wave-uniform) one around an iteration's stores.  The reciprocal form of the lifting divisions is wrong for a nonzero value below
//    2^-96 (never seen on image data): instead of testing before each division, the wave keeps the smallest
//    exponent it divided (one v_frexp_exp per value that can be small -- not the freshly de-quantised ones, 0 or
//    >= 0.75 / (q qs) -- and a running minimum) and looks at it ONCE, after the band: a wave that met such a value
//    runs its band again with true divisions (EXACT) and overwrites what it stored.  The same second pass takes a
//    wave that de-quantised a coefficient outside the 16 bit-planes the reciprocal form of the step divisions was
//    verified for (a running v_max3_f32 of magnitudes: coded planes never get there, the 23-bit words of a
//    raw-fallback codeblock or a caller's own array can).  So an iteration is one basic block up to its stores and
//    the scheduler overlaps its chains;
//  * the step constants live in vector registers (a VOP3 fma with a scalar operand issues at half rate), the lifting
//    constants are literals of v_fmamk / v_mul; rows addressed by scalar offsets into buffer resources; waves that
//    hold no image-edge column without the mirror selects (EDGE); the coarsest level's LL de-quantisation is an
//    instantiation (FIRST), not a test per row.
struct DivK { float rc, nc; };                  // a divisor's correctly rounded reciprocal and its negative, in VGPRs
__device__ __forceinline__ float div_rcv(float x, const DivK &k)
{   // div_rc(x, c, rc): fma(q, -c, x) is fma(-q, c, x)
    const float q = x * k.rc;
    const float r = fmaf(q, k.nc, x);
    return fmaf(r, k.rc, q);
}
__device__ __forceinline__ float fmed3(float x, float lo, float hi)
{
#if defined(__AMDGCN__)
    return __builtin_amdgcn_fmed3f(x, lo, hi);
#else
    return x < lo ? lo : (x > hi ? hi : x);
#endif
}
__device__ __forceinline__ float fmax3abs(float m, float a, float b)
{   // max(m, |a|, |b|): one v_max3_f32 with source modifiers
    return __builtin_fmaxf(__builtin_fmaxf(m, __builtin_fabsf(a)), __builtin_fabsf(b));
}
// two coefficients of one subband; vmax: the largest magnitude the wave's lanes have de-quantised (TRACK; 16-bit
// coefficients cannot leave the verified domain)
template <bool TRACK = true>
__device__ __forceinline__ void dequant1x2(uint32_t raw0, uint32_t raw1, const DivK &k, float &x0, float &x1, float &vmax)
{   // (|v| + 0.5) sgn v = v + clamp(v, -0.5, 0.5), exact for |v| < 2^23; 0 stays 0
    const float y0 = (float)(int)raw0, y1 = (float)(int)raw1;
    if constexpr (TRACK) vmax = fmax3abs(vmax, y0, y1);
    x0 = div_rcv(y0 + fmed3(y0, -0.5f, 0.5f), k);
    x1 = div_rcv(y1 + fmed3(y1, -0.5f, 0.5f), k);
}
// binary exponent e of x = m 2^e, 0.5 <= |m| < 1; 0 for zero (and infinities / NaN)
__device__ __forceinline__ int frexp_exp(float x)
{
#if defined(__AMDGCN__)
    return __builtin_amdgcn_frexp_expf(x);
#else
    if (x == 0.0f || !std::isfinite(x)) return 0;
    int e;
    (void)std::frexp(x, &e);
    return e;
#endif
}
constexpr int kTinyExp = -96;                   // |x| < 2^-96 (tiny_key's bound): e <= -96
__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
template <bool EXACT> __device__ __forceinline__ float div_n1(float x)
{
    if constexpr (EXACT) return x / PS_N1;
    const float q = x * PS_RN1;
    return fmaf(fmaf(q, -PS_N1, x), PS_RN1, q);
}
template <bool EXACT> __device__ __forceinline__ float div_n2(float x)
{
    if constexpr (EXACT) return x / PS_N2;
    const float q = x * PS_RN2;
    return fmaf(fmaf(q, -PS_N2, x), PS_RN2, q);
}

// hinv(float) without its test
template <bool EDGE, bool EXACT>
__device__ __forceinline__ void hinv97(float v[4], bool le, bool re)
{   // DWTGenerator.cu:326-339, lifting :110-122
    v[1] = div_n1<EXACT>(v[1]); v[3] = div_n1<EXACT>(v[3]);
    v[0] = div_n2<EXACT>(v[0]); v[2] = div_n2<EXACT>(v[2]);
    const bool l = EDGE && le, r = EDGE && re;
    float dp = prv<float>(v[3], v[1], l);
    v[0] = fmaf(-(v[1] + dp), PS_A4, v[0]);
    v[2] = fmaf(-(v[3] + v[1]), PS_A4, v[2]);
    float sn = nxt<float>(v[0], v[2], r);
    v[1] = fmaf(-(v[0] + v[2]), PS_A3, v[1]);
    v[3] = fmaf(-(v[2] + sn), PS_A3, v[3]);
    dp = prv<float>(v[3], v[1], l);
    v[0] = fmaf(-(v[1] + dp), PS_A2, v[0]);
    v[2] = fmaf(-(v[3] + v[1]), PS_A2, v[2]);
    sn = nxt<float>(v[0], v[2], r);
    v[1] = fmaf(-(v[0] + v[2]), PS_A1, v[1]);
    v[3] = fmaf(-(v[2] + sn), PS_A1, v[3]);
}

struct Inv97Steps { DivK ll, hl, lh, hh, qs; };

// one subband row pair-segment to samples (s0, d0, s1, d1); HIGH: LH / HH row, else LL / HL (LL de-quantised on
// the coarsest level only: DEQ_S)
template <bool HIGH, bool DEQ_S, bool ONE_DIV, bool EXACT, bool C16 = false>
__device__ __forceinline__ void convert97(const DwtInvArgs &a, const Inv97Steps &k, const SubRaw &r0, float v[4], float &vmax)
{
    SubRaw r = r0;
    if constexpr (C16) {                                     // packed pairs of 16-bit coefficients (load_pair)
        r.d1 = (uint32_t)c16_hi(r0.d0); r.d0 = (uint32_t)c16_lo(r0.d0);
        if (DEQ_S) { r.s1 = (uint32_t)c16_hi(r0.s0); r.s0 = (uint32_t)c16_lo(r0.s0); }
    }
    if constexpr (EXACT) {
        const float qd = HIGH ? a.q[3] : a.q[1], qsb = HIGH ? a.q[2] : a.q[0];
        v[1] = dequant<false>((int32_t)r.d0, qd, 0.0f, a.qs, 0.0f); v[3] = dequant<false>((int32_t)r.d1, qd, 0.0f, a.qs, 0.0f);
        if (DEQ_S) { v[0] = dequant<false>((int32_t)r.s0, qsb, 0.0f, a.qs, 0.0f); v[2] = dequant<false>((int32_t)r.s1, qsb, 0.0f, a.qs, 0.0f); }
        else { v[0] = __uint_as_float(r.s0); v[2] = __uint_as_float(r.s1); }
        return;
    }
    const DivK &kd = HIGH ? k.hh : k.hl, &ks = HIGH ? k.lh : k.ll;
    dequant1x2<!C16>(r.d0, r.d1, kd, v[1], v[3], vmax);
    if (DEQ_S) dequant1x2<!C16>(r.s0, r.s1, ks, v[0], v[2], vmax);
    else { v[0] = __uint_as_float(r.s0); v[2] = __uint_as_float(r.s1); }
    if (!ONE_DIV) {
        v[1] = div_rcv(v[1], k.qs); v[3] = div_rcv(v[3], k.qs);
        if (DEQ_S) { v[0] = div_rcv(v[0], k.qs); v[2] = div_rcv(v[2], k.qs); }
    }
}

// RGB (dwt_inv97_rgb_kernel): the wave is one of the three that run the finest level of an RGB frame's three components
// over the same columns; a finished row pair is exchanged through LDS and the wave delivers ONE pixel plane -- row `comp`
// of the inverse ICT (rgb_inverse_kernel<float>'s arithmetic), level shift and clamp included.
struct Rgb97Out {
    float4 *xb;             // [2 (iteration parity)][3 components][2 rows][64 lanes]
    int comp;               // this wave's component in, pixel plane out (0 / 1 / 2: Y -> R, Cb -> G, Cr -> B)
    uint8_t *plane;         // the pixel plane this wave writes, row stride W
    float m0, m1, m2;       // row `comp` of the inverse matrix
    int off;
};
__device__ __forceinline__ uint32_t ict_pixels(const float4 &y, const float4 &cb, const float4 &cr, const Rgb97Out &o)
{
    const float yv[4] = { y.x, y.y, y.z, y.w }, bv[4] = { cb.x, cb.y, cb.z, cb.w }, rv[4] = { cr.x, cr.y, cr.z, cr.w };
    uint32_t w = 0u;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int v = (int)rintf(fmaf(o.m2, rv[i], fmaf(o.m1, bv[i], o.m0 * yv[i])) + 0.01f) + o.off;
        w |= (uint32_t)(v > 255 ? 255 : (v < 0 ? 0 : v)) << (8 * i);
    }
    return w;
}

// returns (wave-uniform) whether some lane divided a value too small for the reciprocal form: the band must be
// run again with EXACT
template <int BAND, bool U8OUT, bool FIRST, bool ONE_DIV, bool EDGE, bool EXACT, bool C16 = false, bool RGB = false>
__device__ __forceinline__ bool dwt_inv97_band(const DwtInvArgs &a, int strip, int lane, const Rgb97Out *rgb = nullptr)
{
    const int c0 = strip * kStripUseful - 4 * kEdgeLanes + 4 * lane;
    const int pc = c0 >> 1;                                  // arithmetic shift: -4 -> -2
    const int hW = a.W >> 1, hH = a.H >> 1;
    const int m0 = blockIdx.y * (BAND / 2);
    const int m1 = m0 + BAND / 2 > hH ? hH : m0 + BAND / 2;
    const bool wr = lane >= kEdgeLanes && lane <= 63 - kEdgeLanes && c0 >= 0 && c0 < a.W;
    const bool le = EDGE && c0 == 0, re = EDGE && c0 + 4 == a.W;
    const int pl = pc < 0 ? 0 : (pc > hW - 2 ? hW - 2 : pc);

    const RowBuf mal = rowbuf(a.mallat), lls = rowbuf(FIRST ? (const void *)a.mallat : a.ll);
    const RowBuf out = rowbuf(U8OUT ? (const void *)a.dst_u8 : (const void *)a.dst);
    constexpr uint32_t kCB = C16 ? 2u : 4u;                  // bytes of a coded coefficient
    const uint32_t vd = (uint32_t)(hW + pl) * kCB, vs = (uint32_t)pl * kCB, vsl = FIRST ? vs : (uint32_t)pl * 4u;
    const uint32_t vo = wr ? (U8OUT ? (uint32_t)c0 : (uint32_t)c0 * 4u) : kRbDrop;
    const uint32_t aw4 = (uint32_t)a.AW * kCB, ll4 = FIRST ? aw4 : (uint32_t)a.ll_stride * 4u;
    const uint32_t ow = U8OUT ? (uint32_t)a.W : (uint32_t)a.W * 4u;

    // the steps (times qs when that is exact scaling) as reciprocal / negative pairs in vector registers
    Inv97Steps k;
    if constexpr (!EXACT) {
        const float vsc = in_vgpr(ONE_DIV ? a.qs : 1.0f), vrsc = in_vgpr(ONE_DIV ? a.rqs : 1.0f);
        k.ll.rc = in_vgpr(a.rq[0]) * vrsc; k.ll.nc = -(in_vgpr(a.q[0]) * vsc);
        k.hl.rc = in_vgpr(a.rq[1]) * vrsc; k.hl.nc = -(in_vgpr(a.q[1]) * vsc);
        k.lh.rc = in_vgpr(a.rq[2]) * vrsc; k.lh.nc = -(in_vgpr(a.q[2]) * vsc);
        k.hh.rc = in_vgpr(a.rq[3]) * vrsc; k.hh.nc = -(in_vgpr(a.q[3]) * vsc);
        k.qs.rc = in_vgpr(a.rqs); k.qs.nc = in_vgpr(-a.qs);
    }

    const float foff = U8OUT ? in_vgpr((float)a.off) : 0.0f;

    constexpr int kIters = BAND / 2 + 4;                     // two pairs of run-in either side
    constexpr int kGroup = inv_group(kIters, PICSONG_DWT_INV97_GROUP);
    const int j0 = m0 - 2;
    SubRaw rawL[kGroup], rawH[kGroup];
    auto load_pair = [&](int j, SubRaw &L, SubRaw &H) {
        const uint32_t rl = (uint32_t)reflect_s(j, hH), rh = (uint32_t)(reflect_d(j, hH) + hH);
        if constexpr (C16) {                                 // a pair of 16-bit coefficients = one dword, unpacked by convert97
            L.d0 = rb_load32(mal, vd, rl * aw4); L.d1 = 0u;
            if constexpr (FIRST) { L.s0 = rb_load32(lls, vsl, rl * ll4); L.s1 = 0u; }
            else { const uint2 ls = rb_load64(lls, vsl, rl * ll4); L.s0 = ls.x; L.s1 = ls.y; }
            H.d0 = rb_load32(mal, vd, rh * aw4); H.d1 = 0u;
            H.s0 = rb_load32(mal, vs, rh * aw4); H.s1 = 0u;
        } else {
            const uint2 ld = rb_load64(mal, vd, rl * aw4), ls = rb_load64(lls, vsl, rl * ll4);
            const uint2 hd = rb_load64(mal, vd, rh * aw4), hs = rb_load64(mal, vs, rh * aw4);
            L.d0 = ld.x; L.d1 = ld.y; L.s0 = ls.x; L.s1 = ls.y;
            H.d0 = hd.x; H.d1 = hd.y; H.s0 = hs.x; H.s1 = hs.y;
        }
    };
    __builtin_amdgcn_s_setprio(3);
#pragma unroll
    for (int p = 0; p < kGroup; p++) load_pair(j0 + p, rawL[p], rawH[p]);
    __builtin_amdgcn_s_setprio(0);

    // vertical 9/7 synthesis, DWTGenerator.cu:230-272, streamed: at step j pair j-2 completes
    float ddp[4], s1p[4], d1p[4], s0p[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { ddp[i] = s1p[i] = d1p[i] = s0p[i] = 0.0f; }
    int emin = 0;                                            // smallest exponent a lifting division has seen
    float vmax = 0.0f;                                       // largest coefficient magnitude de-quantised
#pragma unroll 1
    for (int g = 0; g < kIters / kGroup; g++) {
#pragma unroll
        for (int r = 0; r < kGroup; r++) {
            const int it = g * kGroup + r;
            const int j = j0 + it;
            float ln[4], hn[4];
            convert97<false, FIRST, ONE_DIV, EXACT, C16>(a, k, rawL[r], ln, vmax);
            convert97<true, true, ONE_DIV, EXACT, C16>(a, k, rawH[r], hn, vmax);
            // (also in the last trip, whose rows nobody uses: an `if` here makes the compiler merge loaded and kept
            // registers with copies behind a full s_waitcnt, which serialises the prefetch; reflect_* keep every
            // row inside the subband).  Fenced: left to itself the scheduler sinks the loads to the end of the trip.
#if defined(__AMDGCN__)
            __builtin_amdgcn_sched_barrier(0);
#endif
            load_pair(j + kGroup, rawL[r], rawH[r]);
#if defined(__AMDGCN__)
            __builtin_amdgcn_sched_barrier(0);
#endif
            // the previous level's LL samples are the only ones that can be tiny at this point
            if (!EXACT && !FIRST) emin = imin(emin, imin(frexp_exp(ln[0]), frexp_exp(ln[2])));
            hinv97<EDGE, EXACT>(ln, le, re);
            hinv97<EDGE, EXACT>(hn, le, re);
            if (!EXACT) {
                emin = imin(emin, imin(imin(frexp_exp(hn[0]), frexp_exp(hn[1])), imin(frexp_exp(hn[2]), frexp_exp(hn[3]))));
                emin = imin(emin, imin(imin(frexp_exp(ln[0]), frexp_exp(ln[1])), imin(frexp_exp(ln[2]), frexp_exp(ln[3]))));
            }
            float ev[4], od[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float dd = div_n1<EXACT>(hn[i]);
                const float s1 = fmaf(-(ddp[i] + dd), PS_A4, div_n2<EXACT>(ln[i]));
                const float d1 = fmaf(-(s1p[i] + s1), PS_A3, ddp[i]);       // d1[j-1]
                const float s0 = fmaf(-(d1p[i] + d1), PS_A2, s1p[i]);       // s0[j-1]
                const float xo = fmaf(-(s0p[i] + s0), PS_A1, d1p[i]);       // x[2(j-2)+1]
                ev[i] = s0p[i];
                od[i] = xo;
                ddp[i] = dd; s1p[i] = s1; d1p[i] = d1; s0p[i] = s0;
            }
            // pair j - 2: stored when it is one of the band's (not a run-in pair, not past the image's last pair)
            if (it >= 4 && j - 2 < m1) {
                const uint32_t y = (uint32_t)(2 * (j - 2));
                if constexpr (RGB) {
                    // (the three waves of the workgroup run the same strip and band: the same trips take this branch)
                    float4 *const xb = rgb->xb + (size_t)(it & 1) * (3 * 2 * 64);
                    xb[(rgb->comp * 2 + 0) * 64 + lane] = make_float4(ev[0], ev[1], ev[2], ev[3]);
                    xb[(rgb->comp * 2 + 1) * 64 + lane] = make_float4(od[0], od[1], od[2], od[3]);
                    __syncthreads();
                    const uint32_t we = ict_pixels(xb[0 * 64 + lane], xb[2 * 64 + lane], xb[4 * 64 + lane], *rgb);
                    const uint32_t wo = ict_pixels(xb[1 * 64 + lane], xb[3 * 64 + lane], xb[5 * 64 + lane], *rgb);
                    if (wr) {
                        *reinterpret_cast<uint32_t *>(rgb->plane + (size_t)y * (size_t)a.W + (uint32_t)c0) = we;
                        *reinterpret_cast<uint32_t *>(rgb->plane + (size_t)(y + 1u) * (size_t)a.W + (uint32_t)c0) = wo;
                    }
                } else if constexpr (U8OUT) {
                    rb_store32(out, vo, y * ow, pack_pixels(ev, foff));
                    rb_store32(out, vo, (y + 1u) * ow, pack_pixels(od, foff));
                } else {
                    rb_store128(out, vo, y * ow, as_u32(ev[0]), as_u32(ev[1]), as_u32(ev[2]), as_u32(ev[3]));
                    rb_store128(out, vo, (y + 1u) * ow, as_u32(od[0]), as_u32(od[1]), as_u32(od[2]), as_u32(od[3]));
                }
            }
        }
    }
    // (65536: the reciprocal form of the de-quantisation is verified for 16 bit-planes, dequant_fast_ok)
    return !EXACT && __builtin_amdgcn_ballot_w64(emin <= kTinyExp || !(vmax < 65536.0f)) != 0ull;
}

#ifndef PICSONG_DWT_INV97_WAVES
#define PICSONG_DWT_INV97_WAVES 5     // dwt_inv97_kernel: resident waves per SIMD the register budget is set for
#endif
template <int BAND, bool U8OUT, bool FIRST, bool ONE_DIV, bool C16 = false>
__global__ __launch_bounds__(256, PICSONG_DWT_INV97_WAVES) void dwt_inv97_kernel(DwtInvArgs a)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int strip = blockIdx.x * 4 + wave;
    if (strip * kStripUseful >= a.W) return;                 // whole wave idle (no cross-lane use)
    dwt_inv_select_frame(a);
    const int first = strip * kStripUseful - 4 * kEdgeLanes;
    bool again;
    if (first <= 0 || first + kStripCols >= a.W) again = dwt_inv97_band<BAND, U8OUT, FIRST, ONE_DIV, true, false, C16>(a, strip, lane);
    else again = dwt_inv97_band<BAND, U8OUT, FIRST, ONE_DIV, false, false, C16>(a, strip, lane);
    // (a.exact_replay: PICSONG_DWT_EXACT_REPLAY=1, the tests' way into the second pass)
    if (__builtin_expect(again || a.exact_replay, 0)) dwt_inv97_band<BAND, U8OUT, FIRST, ONE_DIV, true, true, C16>(a, strip, lane);
}

// RGB frames, 9/7: the finest synthesis level of the three components as the three waves of a workgroup (same strip,
// same band), the inverse ICT, level shift and clamp at the stores -- the three 134 MB float planes the separate launches
// wrote and rgb_inverse_kernel read back are never made.  (Three copies of the lean kernel's state do not fit ONE wave, as
// dwt_inv_rgb_kernel does it for 5/3; here a finished row pair crosses LDS, 12 KB a workgroup, one barrier a row pair.)
// A wave that must run its band again with true divisions takes the workgroup with it.
// a: the finest level's arguments of a three-frame plan (component c at mallat + c * mallat_z, LL at ll + c * ll_z).
template <int BAND, bool ONE_DIV>
__global__ __launch_bounds__(192, PICSONG_DWT_INV97_WAVES) void dwt_inv97_rgb_kernel(DwtInvArgs a, uint8_t *dr, uint8_t *dg, uint8_t *db)
{
    __shared__ float4 xb[2 * 3 * 2 * 64];
    __shared__ int again_any;
    const int lane = threadIdx.x & 63, comp = threadIdx.x >> 6;
    const int strip = blockIdx.x;
    if (strip * kStripUseful >= a.W) return;                 // (the whole workgroup: one strip)
    if (threadIdx.x == 0) again_any = 0;
    a.mallat = (const int32_t *)((const char *)a.mallat + (unsigned long long)comp * a.mallat_z);
    a.ll = (const char *)a.ll + (unsigned long long)comp * a.ll_z;
    Rgb97Out o;
    o.xb = xb; o.comp = comp; o.plane = comp == 0 ? dr : (comp == 1 ? dg : db); o.off = a.off;
    // rgb_inverse_kernel<float>'s matrix, row = the plane
    o.m0 = in_vgpr(1.0f);
    o.m1 = in_vgpr(comp == 0 ? 0.0f : (comp == 1 ? -0.344136f : 1.772f));
    o.m2 = in_vgpr(comp == 0 ? 1.402f : (comp == 1 ? -0.714136f : 0.0f));
    __syncthreads();
    const int first = strip * kStripUseful - 4 * kEdgeLanes;
    bool again;
    if (first <= 0 || first + kStripCols >= a.W) again = dwt_inv97_band<BAND, false, false, ONE_DIV, true, false, true, true>(a, strip, lane, &o);
    else again = dwt_inv97_band<BAND, false, false, ONE_DIV, false, false, true, true>(a, strip, lane, &o);
    if (again && lane == 0) atomicOr(&again_any, 1);
    __syncthreads();
    if (__builtin_expect(again_any != 0 || a.exact_replay, 0)) dwt_inv97_band<BAND, false, false, ONE_DIV, true, true, true, true>(a, strip, lane, &o);
}

// ---- synthesis levels 1 and 0 in one launch (the decode frame paths' tail; round 4) ----------------------------
// The mirror of dwt_fwd2_kernel.  The rows level 1 delivers -- LL0, the low-low input of level 0 -- never leave the
// registers: as a band streams down, every level-1 step hands two LL0 rows to two level-0 steps, whose other inputs
// (HL0 / LH0 / HH0) come from the coded array.  Saves the write and the read of the LL0 plane (2 x 33.4 MB of an 8K
// frame's 32-bit words) and a launch; with the coded coefficients as int16 (the decoder's C16 instantiation) the finest
// level reads 50 MB of subbands where the 32-bit form reads 134 MB, and writes 33 MB of pixels.  Costs recomputed
// run-in: a band of NB level-1 row pairs (4 NB output rows) runs NB + 4 (5/3) / NB + 6 (9/7) level-1 steps and
// 2 NB + 4 / 2 NB + 6 level-0 steps.  A lane owns 4 output columns = 2 level-0 column pairs = 1 level-1 pair; 1 (5/3)
// / 3 (9/7) recomputed lanes per side.  Arithmetic, value for value: dwt_inv_kernel's (5/3) and dwt_inv97_band's (9/7:
// reciprocal divisions, optimistic execution, the band run again with true divisions for a wave that met a value below
// 2^-96).  Reference: DWTEngine::DWTReverse DWT/DWTGenerator.cu:1349-1424, kernelDWTReverse(Lossy) :1002-1124.
//
// Iteration it of a band (n0 = its first level-1 pair, M0 = 2 n0):  level-1 step j1 = n0 - R1 + it completes level-1
// pair j1 - G1 = LL0 rows m, m + 1 with m = M0 - 2 (R1 + G1) + 2 it;  level-0 steps m and m + 1 complete the output
// pairs m - G0 and m + 1 - G0.  5/3: R1 = 2, G1 = G0 = 1;  9/7: R1 = 3, G1 = G0 = 2.  The first real LL0 rows are
// M0 - G0 .. at it = F0 = R1 + G1 - ... (2 / 4): the iterations before are level 1's run-in and touch no level-0 data.
// The image's top needs nothing special (mirrored subband rows make level 1 deliver the mirrored LL0 rows level 0 wants);
// at the bottom level 0 wants LL0[K] = LL0[K - 1], LL0[K + 1] = LL0[K - 2] where mirrored level-1 rows would deliver
// LL0[K - 2], LL0[K - 3]: the last band's last iteration takes the rows of the iteration before, swapped.
template <bool LOSSY> constexpr int i2_edge() { return LOSSY ? 3 : 1; }
template <bool LOSSY> constexpr int i2_useful() { return kStripCols - 8 * i2_edge<LOSSY>(); }
constexpr int kI2Pairs = 8;                               // level-1 row pairs per band: 32 output rows
struct DwtInv2Args { DwtInvArgs l1, l0; };

__device__ __forceinline__ void hinv2(int v[2], bool le, bool re)
{   // hinv<>(int) on ONE (s, d) pair per lane
    const int dp = prv<int>(v[1], v[1], le);
    v[0] -= (v[1] + dp + 2) >> 2;
    const int sn = nxt<int>(v[0], v[0], re);
    v[1] += (v[0] + sn) >> 1;
}
template <bool EDGE, bool EXACT>
__device__ __forceinline__ void hinv97_2(float v[2], bool le, bool re)
{   // hinv97 on ONE (s, d) pair per lane
    v[1] = div_n1<EXACT>(v[1]);
    v[0] = div_n2<EXACT>(v[0]);
    const bool l = EDGE && le, r = EDGE && re;
    float dp = prv<float>(v[1], v[1], l);
    v[0] = fmaf(-(v[1] + dp), PS_A4, v[0]);
    float sn = nxt<float>(v[0], v[0], r);
    v[1] = fmaf(-(v[0] + sn), PS_A3, v[1]);
    dp = prv<float>(v[1], v[1], l);
    v[0] = fmaf(-(v[1] + dp), PS_A2, v[0]);
    sn = nxt<float>(v[0], v[0], r);
    v[1] = fmaf(-(v[0] + sn), PS_A1, v[1]);
}
// one vertical synthesis step over N columns: rows L (vertically low) and H (high) of step j; delivers the row pair
// j - 1 (5/3: st = H[j-1], s[j-1]) / j - 2 (9/7: st = dd, s1, d1, s0 of the steps before) as ev / od
template <bool LOSSY, bool EXACT, int N, typename T>
__device__ __forceinline__ void vinv_step(T (&st)[4][N], const T (&L)[N], const T (&H)[N], T (&ev)[N], T (&od)[N])
{
#pragma unroll
    for (int k = 0; k < N; k++) {
        if constexpr (LOSSY) {                               // dwt_inv97_band's step, DWTGenerator.cu:230-272
            const float dd = div_n1<EXACT>(H[k]);
            const float s1 = fmaf(-(st[0][k] + dd), PS_A4, div_n2<EXACT>(L[k]));
            const float d1 = fmaf(-(st[1][k] + s1), PS_A3, st[0][k]);
            const float s0 = fmaf(-(st[2][k] + d1), PS_A2, st[1][k]);
            const float xo = fmaf(-(st[3][k] + s0), PS_A1, st[2][k]);
            ev[k] = st[3][k];
            od[k] = xo;
            st[0][k] = dd; st[1][k] = s1; st[2][k] = d1; st[3][k] = s0;
        } else {                                             // dwt_inv_kernel's step, DWTGenerator.cu:160-181
            const T sv = L[k] - ((st[0][k] + H[k] + 2) >> 2);
            ev[k] = st[1][k];
            od[k] = st[0][k] + ((st[1][k] + sv) >> 1);
            st[1][k] = sv; st[0][k] = H[k];
        }
    }
}
struct Inv2Steps { DivK hl1, lh1, hh1, hl0, lh0, hh0, qs; };
// the raw words of one iteration: level 1's four subband samples of the lane's pair (LL1 a T word, the coded ones
// sign-extended 16-bit loads), level 0's HL / LH / HH pairs (packed int16) of its two steps
struct Inv2Raw { uint32_t ll1; int hl1, lh1, hh1; uint32_t hl0[2], lh0[2], hh0[2]; };

template <bool LOSSY, bool ONE_DIV, bool EXACT>
__device__ __forceinline__ float i2_deq(int c, const DivK &k, const DivK &kqs, float q, const DwtInvArgs &a)
{   // dequant1x2's arithmetic for one 16-bit coefficient (EXACT: readSubbands' own divisions)
    if constexpr (EXACT) return dequant<false>(c, q, 0.0f, a.qs, 0.0f);
    const float y = (float)c;
    float x = div_rcv(y + fmed3(y, -0.5f, 0.5f), k);
    if (!ONE_DIV) x = div_rcv(x, kqs);
    return x;
}

template <bool LOSSY, bool ONE_DIV, bool EDGE, bool EXACT>
__device__ __forceinline__ bool dwt_inv2_band(const DwtInvArgs &a1, const DwtInvArgs &a0, int strip, int lane)
{
    using T = typename std::conditional<LOSSY, float, int>::type;
    constexpr int NB = kI2Pairs, kE = i2_edge<LOSSY>();
    constexpr int R1 = LOSSY ? 3 : 2, G1 = LOSSY ? 2 : 1, G0 = G1, F0 = LOSSY ? 4 : 2;
    constexpr int kIters = NB + R1 + G1 + 1;                 // NB + 4 / NB + 6
    constexpr int kMain = kIters - F0;                       // NB + 2 iterations with level-0 work
    constexpr int kG = 2;                                    // iterations whose loads are in flight ahead of the arithmetic
    static_assert(kMain % kG == 0 && F0 % kG == 0, "the main loop runs in groups of kG iterations");
    const int c0 = strip * i2_useful<LOSSY>() - 4 * kE + 4 * lane;
    const int W0 = a0.W, hW0 = W0 >> 1, hH0 = a0.H >> 1, hW1 = a1.W >> 1, hH1 = a1.H >> 1;
    const int n0 = (int)blockIdx.y * NB, M0 = 2 * n0;
    const bool lastb = 2 * (M0 + 2 * NB) >= a0.H;            // the band that ends at the bottom of the image
    const bool wr = lane >= kE && lane <= 63 - kE && c0 >= 0 && c0 < W0;
    const bool le = EDGE && c0 == 0, re = EDGE && c0 + 4 == W0;
    const int pc = c0 >> 1;
    const int pl = pc < 0 ? 0 : (pc > hW0 - 2 ? hW0 - 2 : pc), ql = pl >> 1;

    const RowBuf mal = rowbuf(a0.mallat), ll1b = rowbuf(a1.ll), out = rowbuf(a0.dst_u8);
    const uint32_t aw2 = (uint32_t)a0.AW * 2u, l1s = (uint32_t)a1.ll_stride * 4u;
    const uint32_t vll1 = (uint32_t)ql * 4u, vlh1 = (uint32_t)ql * 2u, vhl1 = (uint32_t)(hW1 + ql) * 2u;
    const uint32_t vlh0 = (uint32_t)pl * 2u, vhl0 = (uint32_t)(hW0 + pl) * 2u;
    const uint32_t vo = wr ? (uint32_t)c0 : kRbDrop, ow = (uint32_t)W0;

    Inv2Steps k;
    if constexpr (LOSSY && !EXACT) {
        const float vsc = in_vgpr(ONE_DIV ? a0.qs : 1.0f), vrsc = in_vgpr(ONE_DIV ? a0.rqs : 1.0f);
        k.hl1.rc = in_vgpr(a1.rq[1]) * vrsc; k.hl1.nc = -(in_vgpr(a1.q[1]) * vsc);
        k.lh1.rc = in_vgpr(a1.rq[2]) * vrsc; k.lh1.nc = -(in_vgpr(a1.q[2]) * vsc);
        k.hh1.rc = in_vgpr(a1.rq[3]) * vrsc; k.hh1.nc = -(in_vgpr(a1.q[3]) * vsc);
        k.hl0.rc = in_vgpr(a0.rq[1]) * vrsc; k.hl0.nc = -(in_vgpr(a0.q[1]) * vsc);
        k.lh0.rc = in_vgpr(a0.rq[2]) * vrsc; k.lh0.nc = -(in_vgpr(a0.q[2]) * vsc);
        k.hh0.rc = in_vgpr(a0.rq[3]) * vrsc; k.hh0.nc = -(in_vgpr(a0.q[3]) * vsc);
        k.qs.rc = in_vgpr(a0.rqs); k.qs.nc = in_vgpr(-a0.qs);
    }
    const float foff = LOSSY ? in_vgpr((float)a0.off) : 0.0f;

    // loads of iteration `it`: level 1's rows of step j1, level 0's rows of steps m, m + 1 (any row index: reflected)
    auto load1 = [&](int it, Inv2Raw &r) {
        const int j1 = n0 - R1 + it;
        const uint32_t rl = (uint32_t)reflect_s(j1, hH1), rh = (uint32_t)(reflect_d(j1, hH1) + hH1);
        r.ll1 = rb_load32(ll1b, vll1, rl * l1s);
        r.hl1 = rb_load16s(mal, vhl1, rl * aw2);
        r.lh1 = rb_load16s(mal, vlh1, rh * aw2);
        r.hh1 = rb_load16s(mal, vhl1, rh * aw2);
    };
    auto load0 = [&](int it, Inv2Raw &r) {
        const int m = M0 - 2 * (R1 + G1) + 2 * it;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const uint32_t rl = (uint32_t)reflect_s(m + h, hH0), rh = (uint32_t)(reflect_d(m + h, hH0) + hH0);
            r.hl0[h] = rb_load32(mal, vhl0, rl * aw2);
            r.lh0[h] = rb_load32(mal, vlh0, rh * aw2);
            r.hh0[h] = rb_load32(mal, vhl0, rh * aw2);
        }
    };
    Inv2Raw raw[kG];
    __builtin_amdgcn_s_setprio(3);
#pragma unroll
    for (int p = 0; p < kG; p++) { load1(p, raw[p]); raw[p].hl0[0] = raw[p].hl0[1] = raw[p].lh0[0] = raw[p].lh0[1] = raw[p].hh0[0] = raw[p].hh0[1] = 0u; }
    __builtin_amdgcn_s_setprio(0);

    T st1[4][2], st0[4][4], pev[2], pod[2];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        st1[q][0] = st1[q][1] = (T)0;
        st0[q][0] = st0[q][1] = st0[q][2] = st0[q][3] = (T)0;
    }
    pev[0] = pev[1] = pod[0] = pod[1] = (T)0;
    int emin = 0;                                            // smallest exponent a lifting division has seen (9/7)

    // one level-1 step on the raw words of an iteration: the LL0 rows m (ev) and m + 1 (od)
    auto level1 = [&](const Inv2Raw &r, T (&ev)[2], T (&od)[2]) {
        T L[2], H[2];
        if constexpr (LOSSY) {
            L[0] = __uint_as_float(r.ll1);
            L[1] = i2_deq<LOSSY, ONE_DIV, EXACT>(r.hl1, k.hl1, k.qs, a1.q[1], a1);
            H[0] = i2_deq<LOSSY, ONE_DIV, EXACT>(r.lh1, k.lh1, k.qs, a1.q[2], a1);
            H[1] = i2_deq<LOSSY, ONE_DIV, EXACT>(r.hh1, k.hh1, k.qs, a1.q[3], a1);
            if (!EXACT) emin = imin(emin, frexp_exp(L[0]));  // the previous level's LL sample: the only one that can be tiny here
            hinv97_2<EDGE, EXACT>(L, le, re);
            hinv97_2<EDGE, EXACT>(H, le, re);
            if (!EXACT) emin = imin(emin, imin(imin(frexp_exp(L[0]), frexp_exp(L[1])), imin(frexp_exp(H[0]), frexp_exp(H[1]))));
        } else {
            L[0] = (T)(int)r.ll1; L[1] = (T)r.hl1; H[0] = (T)r.lh1; H[1] = (T)r.hh1;
            hinv2(L, le, re);
            hinv2(H, le, re);
        }
        vinv_step<LOSSY, EXACT, 2, T>(st1, L, H, ev, od);
    };
    // one level-0 step: LL0 row `ll` with the raw HL / LH / HH words of row m + h; stores the output pair `rel` of the band
    auto level0 = [&](const T (&ll)[2], const Inv2Raw &r, int h, int rel) {
        T L[4], H[4];
        if constexpr (LOSSY) {
            L[0] = ll[0]; L[2] = ll[1];
            L[1] = i2_deq<LOSSY, ONE_DIV, EXACT>(c16_lo(r.hl0[h]), k.hl0, k.qs, a0.q[1], a0);
            L[3] = i2_deq<LOSSY, ONE_DIV, EXACT>(c16_hi(r.hl0[h]), k.hl0, k.qs, a0.q[1], a0);
            H[0] = i2_deq<LOSSY, ONE_DIV, EXACT>(c16_lo(r.lh0[h]), k.lh0, k.qs, a0.q[2], a0);
            H[2] = i2_deq<LOSSY, ONE_DIV, EXACT>(c16_hi(r.lh0[h]), k.lh0, k.qs, a0.q[2], a0);
            H[1] = i2_deq<LOSSY, ONE_DIV, EXACT>(c16_lo(r.hh0[h]), k.hh0, k.qs, a0.q[3], a0);
            H[3] = i2_deq<LOSSY, ONE_DIV, EXACT>(c16_hi(r.hh0[h]), k.hh0, k.qs, a0.q[3], a0);
            if (!EXACT) emin = imin(emin, imin(frexp_exp(L[0]), frexp_exp(L[2])));
            hinv97<EDGE, EXACT>(L, le, re);
            hinv97<EDGE, EXACT>(H, le, re);
            if (!EXACT) {
                emin = imin(emin, imin(imin(frexp_exp(H[0]), frexp_exp(H[1])), imin(frexp_exp(H[2]), frexp_exp(H[3]))));
                emin = imin(emin, imin(imin(frexp_exp(L[0]), frexp_exp(L[1])), imin(frexp_exp(L[2]), frexp_exp(L[3]))));
            }
        } else {
            L[0] = ll[0]; L[2] = ll[1]; L[1] = (T)c16_lo(r.hl0[h]); L[3] = (T)c16_hi(r.hl0[h]);
            H[0] = (T)c16_lo(r.lh0[h]); H[2] = (T)c16_hi(r.lh0[h]); H[1] = (T)c16_lo(r.hh0[h]); H[3] = (T)c16_hi(r.hh0[h]);
            hinv<false>(L, le, re);
            hinv<false>(H, le, re);
        }
        T ev[4], od[4];
        vinv_step<LOSSY, EXACT, 4, T>(st0, L, H, ev, od);
        if (rel >= 0 && rel < 2 * NB) {                      // (wave-uniform)
            const uint32_t y = (uint32_t)(2 * (M0 + rel));
            if constexpr (LOSSY) {
                rb_store32(out, vo, y * ow, pack_pixels(ev, foff));
                rb_store32(out, vo, (y + 1u) * ow, pack_pixels(od, foff));
            } else {
                rb_store32(out, vo, y * ow, to_pixel(ev[0], a0.off) | (to_pixel(ev[1], a0.off) << 8) | (to_pixel(ev[2], a0.off) << 16) | (to_pixel(ev[3], a0.off) << 24));
                rb_store32(out, vo, (y + 1u) * ow, to_pixel(od[0], a0.off) | (to_pixel(od[1], a0.off) << 8) | (to_pixel(od[2], a0.off) << 16) | (to_pixel(od[3], a0.off) << 24));
            }
        }
    };
    auto fence = [] {
#if defined(__AMDGCN__)
        __builtin_amdgcn_sched_barrier(0);
#endif
    };

    // ---- level 1's run-in: F0 iterations without level-0 work.  Their loads' successors are the first main
    // iterations', level-0 rows included
#pragma unroll
    for (int it = 0; it < F0; it++) {
        const Inv2Raw r = raw[it % kG];
        fence();
        load1(it + kG, raw[it % kG]);
        if (it + kG >= F0) load0(it + kG, raw[it % kG]);
        fence();
        T ev[2], od[2];
        level1(r, ev, od);
    }
    // ---- the band: kMain iterations of one level-1 step and two level-0 steps
#pragma unroll 1
    for (int g = 0; g < kMain / kG; g++) {
#pragma unroll
        for (int q = 0; q < kG; q++) {
            const int it = F0 + g * kG + q;
            const Inv2Raw r = raw[q];                         // ((F0 + q) % kG == q: F0 is a multiple of kG)
            // (the loads of the last trips reach past the band: rows nobody uses, kept inside the subbands by reflect_*;
            // a condition here would make the compiler merge loaded and kept registers behind a full wait)
            fence();
            load1(it + kG, raw[q]);
            load0(it + kG, raw[q]);
            fence();
            T ev[2], od[2];
            level1(r, ev, od);
            if (lastb && it == kIters - 1) {                 // LL0 rows K, K + 1 = rows K - 1, K - 2 (see above)
                ev[0] = pod[0]; ev[1] = pod[1]; od[0] = pev[0]; od[1] = pev[1];
            }
            pev[0] = ev[0]; pev[1] = ev[1]; pod[0] = od[0]; pod[1] = od[1];
            const int rel = 2 * it - 2 * (R1 + G1) - G0;     // output pair of the first level-0 step, relative to M0
            level0(ev, r, 0, rel);
            level0(od, r, 1, rel + 1);
        }
    }
    return LOSSY && !EXACT && __builtin_amdgcn_ballot_w64(emin <= kTinyExp) != 0ull;
}

// resident waves per SIMD the register budget is set for: the 5/3 form fits 64 registers; the 9/7 form carries both
// levels' window state (24), two iterations' raw words (20) and seven step constants (14): 128 registers, 4 waves
#ifndef PICSONG_DWT_INV2_WAVES
#define PICSONG_DWT_INV2_WAVES 8
#endif
#ifndef PICSONG_DWT_INV2_WAVES_LOSSY
#define PICSONG_DWT_INV2_WAVES_LOSSY 4
#endif
template <bool LOSSY, bool ONE_DIV>
__global__ __launch_bounds__(256, LOSSY ? PICSONG_DWT_INV2_WAVES_LOSSY : PICSONG_DWT_INV2_WAVES) void dwt_inv2_kernel(DwtInv2Args a2)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int strip = blockIdx.x * 4 + wave;
    if (strip * i2_useful<LOSSY>() >= a2.l0.W) return;       // whole wave idle (no cross-lane use)
    dwt_inv_select_frame(a2.l1);
    dwt_inv_select_frame(a2.l0);
    if constexpr (!LOSSY) {
        (void)dwt_inv2_band<false, false, true, false>(a2.l1, a2.l0, strip, lane);
    } else {
        const int first = strip * i2_useful<LOSSY>() - 4 * i2_edge<LOSSY>();
        bool again;
        if (first <= 0 || first + kStripCols >= a2.l0.W) again = dwt_inv2_band<true, ONE_DIV, true, false>(a2.l1, a2.l0, strip, lane);
        else again = dwt_inv2_band<true, ONE_DIV, false, false>(a2.l1, a2.l0, strip, lane);
        if (__builtin_expect(again || a2.l0.exact_replay, 0)) (void)dwt_inv2_band<true, ONE_DIV, true, true>(a2.l1, a2.l0, strip, lane);
    }
}

// ---- level shift kernels (used when the stages are called one by one) -------------------------
// offsetImage Engines/CodingEngine.cu:581-588
template <typename T>
__global__ __launch_bounds__(256) void level_shift_fwd_kernel(const uint8_t *in, T *out, size_t n4, int off)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t w = reinterpret_cast<const uint32_t *>(in)[i];
        T v[4] = { (T)(int)(w & 0xFFu) - (T)off, (T)(int)((w >> 8) & 0xFFu) - (T)off,
                   (T)(int)((w >> 16) & 0xFFu) - (T)off, (T)(int)(w >> 24) - (T)off };
        uint4 o;
        o.x = as_u32(v[0]); o.y = as_u32(v[1]); o.z = as_u32(v[2]); o.w = as_u32(v[3]);
        reinterpret_cast<uint4 *>(out)[i] = o;
    }
}
// removeOffsetAndApplyMaxMin / ...Lossy, Engines/DecodingEngine.cu:706-729
__global__ __launch_bounds__(256) void level_shift_inv_i32_kernel(int32_t *d, size_t n, int off)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        int v = d[i] + off;
        d[i] = v > 255 ? 255 : (v < 0 ? 0 : v);
    }
}
__global__ __launch_bounds__(256) void level_shift_inv_f32_kernel(float *d, size_t n, float off)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float t = d[i] + off;
        t = t + 0.01f;
        float r = rintf(t);                      // __float2int_rn
        r = r > 255.0f ? 255.0f : r;
        d[i] = r < 0.0f ? 0.0f : r;
    }
}
// removeOffsetAndApplyMaxMin(/Lossy) fused with the u8 conversion the reference does on the host
// (IOManager::writeImage IO/IOManager.ipp:332-335): one pass T[P] -> u8[P], 4 samples per lane
__global__ __launch_bounds__(256) void clamp_to_u8_i32_kernel(const int32_t *in, uint8_t *out, size_t n4, int off)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 w = reinterpret_cast<const uint4 *>(in)[i];
        int v[4] = { (int)w.x + off, (int)w.y + off, (int)w.z + off, (int)w.w + off };
        uint32_t o = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) o |= (uint32_t)(v[k] > 255 ? 255 : (v[k] < 0 ? 0 : v[k])) << (8 * k);
        reinterpret_cast<uint32_t *>(out)[i] = o;
    }
}
__global__ __launch_bounds__(256) void clamp_to_u8_f32_kernel(const float *in, uint8_t *out, size_t n4, float off)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 w = reinterpret_cast<const uint4 *>(in)[i];
        float v[4] = { __uint_as_float(w.x), __uint_as_float(w.y), __uint_as_float(w.z), __uint_as_float(w.w) };
        uint32_t o = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            float t = v[k] + off;
            t = t + 0.01f;
            float r = rintf(t);                  // __float2int_rn
            r = r > 255.0f ? 255.0f : r;
            r = r < 0.0f ? 0.0f : r;
            o |= (uint32_t)(int)r << (8 * k);
        }
        reinterpret_cast<uint32_t *>(out)[i] = o;
    }
}
// ---- colour transforms of the RGB path (RGBTransformLossless / RGBTransformLossy, reference
// Engines/CodingEngine.cu:357-403 and Engines/DecodingEngine.cu:599-650), level shift fused, four
// samples per lane.  RCT: c0 = floor((R+2G+B)/4), c1 = B-G, c2 = R-G.  ICT: 3x3 matrix, evaluated
// m0*R then two fmaf (the contraction nvcc applies), so results are bit-identical to the oracle.
__device__ __forceinline__ void unpack4(uint32_t w, int v[4], int off)
{
    v[0] = (int)(w & 0xFFu) - off; v[1] = (int)((w >> 8) & 0xFFu) - off;
    v[2] = (int)((w >> 16) & 0xFFu) - off; v[3] = (int)(w >> 24) - off;
}

template <typename T>
__global__ __launch_bounds__(256) void rgb_forward_kernel(const uint8_t *r, const uint8_t *g, const uint8_t *b,
                                                          T *c0, T *c1, T *c2, size_t n4, int off)
{
    const float M[3][3] = { { 0.299f, 0.587f, 0.114f }, { -0.168736f, -0.331264f, 0.5f }, { 0.5f, -0.418688f, -0.081312f } };
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        int R[4], G[4], B[4];
        unpack4(reinterpret_cast<const uint32_t *>(r)[i], R, off);
        unpack4(reinterpret_cast<const uint32_t *>(g)[i], G, off);
        unpack4(reinterpret_cast<const uint32_t *>(b)[i], B, off);
        uint4 o0, o1, o2;
        uint32_t *q0 = &o0.x, *q1 = &o1.x, *q2 = &o2.x;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if constexpr (std::is_integral<T>::value) {                      // integer: RCT
                q0[k] = (uint32_t)((R[k] + 2 * G[k] + B[k]) >> 2);
                q1[k] = (uint32_t)(B[k] - G[k]);
                q2[k] = (uint32_t)(R[k] - G[k]);
            } else {                                                      // float: ICT
                const float fr = (float)R[k], fg = (float)G[k], fb = (float)B[k];
                q0[k] = __float_as_uint(fmaf(M[0][2], fb, fmaf(M[0][1], fg, M[0][0] * fr)));
                q1[k] = __float_as_uint(fmaf(M[1][2], fb, fmaf(M[1][1], fg, M[1][0] * fr)));
                q2[k] = __float_as_uint(fmaf(M[2][2], fb, fmaf(M[2][1], fg, M[2][0] * fr)));
            }
        }
        reinterpret_cast<uint4 *>(c0)[i] = o0;
        reinterpret_cast<uint4 *>(c1)[i] = o1;
        reinterpret_cast<uint4 *>(c2)[i] = o2;
    }
}

__device__ __forceinline__ uint32_t clamp_u8(int v) { return (uint32_t)(v > 255 ? 255 : (v < 0 ? 0 : v)); }

template <typename T>
__global__ __launch_bounds__(256) void rgb_inverse_kernel(const T *c0, const T *c1, const T *c2, uint8_t *r,
                                                          uint8_t *g, uint8_t *b, size_t n4, int off)
{
    const float M[3][3] = { { 1.0f, 0.0f, 1.402f }, { 1.0f, -0.344136f, -0.714136f }, { 1.0f, 1.772f, 0.0f } };
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 w0 = reinterpret_cast<const uint4 *>(c0)[i], w1 = reinterpret_cast<const uint4 *>(c1)[i],
                    w2 = reinterpret_cast<const uint4 *>(c2)[i];
        const uint32_t *q0 = &w0.x, *q1 = &w1.x, *q2 = &w2.x;
        uint32_t oR = 0, oG = 0, oB = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int R, G, B;
            if constexpr (std::is_integral<T>::value) {
                const int y = (int)q0[k], cb = (int)q1[k], cr = (int)q2[k];
                G = y - ((cb + cr) >> 2);
                R = cr + G;
                B = cb + G;
            } else {
                const float y = __uint_as_float(q0[k]), cb = __uint_as_float(q1[k]), cr = __uint_as_float(q2[k]);
                R = (int)rintf(fmaf(M[0][2], cr, fmaf(M[0][1], cb, M[0][0] * y)) + 0.01f);
                G = (int)rintf(fmaf(M[1][2], cr, fmaf(M[1][1], cb, M[1][0] * y)) + 0.01f);
                B = (int)rintf(fmaf(M[2][2], cr, fmaf(M[2][1], cb, M[2][0] * y)) + 0.01f);
            }
            oR |= clamp_u8(R + off) << (8 * k);
            oG |= clamp_u8(G + off) << (8 * k);
            oB |= clamp_u8(B + off) << (8 * k);
        }
        reinterpret_cast<uint32_t *>(r)[i] = oR;
        reinterpret_cast<uint32_t *>(g)[i] = oG;
        reinterpret_cast<uint32_t *>(b)[i] = oB;
    }
}

}  // namespace picsong
