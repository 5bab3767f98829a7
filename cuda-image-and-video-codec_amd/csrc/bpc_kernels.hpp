// bpc_kernels.hpp -- BPC-PaCo bit-plane coder / decoder for gfx950 (CDNA4), 2 coding passes,
// k = 0 and the complexity-scalable mode k > 0 (BULK instantiations).
//
// Replaces kernelBPCCoder<T> / kernelBPCDecoder<int> (reference BPC/BPCEngine.cu:1929-2026,
// 2126-2215) and everything they call (SPP :490-516/:559-594/:770-843, MRP :726-762/:1249-1279,
// arithmetic coder :371-442, LUT pointers :329-358, findSubband :143-170, findMSB :176-192,
// expansion fallback :1905-1922).  The emitted codewords, their slot order and sizeArray are
// bit-identical to that algorithm under the lock-step reading of __activemask;
// the implementation is not a translation:
//
//   * one wave64 = TWO codeblocks (lanes 0-31 / 32-63); every lane owns two columns x 64 rows,
//     like a reference lane, but coefficients are held TRANSPOSED: per column one 64-bit row-mask
//     per bit-plane (2 VGPRs), plus 64-bit significance / sign / refinement row-masks.  A plane
//     visit is bit-field extraction (v_alignbit / v_bfe / v_bcnt) on those masks instead of 128
//     live 32-bit coefficient words per lane, and the "current plane" is always register 0: the
//     plane registers rotate once per plane, so no register is indexed dynamically.
//   * neighbour columns arrive by DPP wave_shr:1 / wave_shl:1 -- once per plane in the encoder,
//     in the decoder only after a lane actually changed state -- not by three LDS-crossbar
//     shuffles per coefficient.
//   * codeword slot reservation is a 64-bit ballot split per 32-lane half + v_mbcnt; the per-half
//     counter lives in a VGPR that is uniform across the half.
//   * rows in which no lane of the wave has anything to code are skipped (wave-uniform row mask).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace picsong {

// time-resolved trace of the encoder's waves (variant builds only: -DPICSONG_DWT_TRACE, tools/bpc_trace.py): start,
// planes parked, plane loop done (s_memrealtime, 100 MHz) and the wave's plane count, into the buffer of
// picsong_debug_set_bpc_trace
#ifdef PICSONG_DWT_TRACE
__device__ unsigned long long *g_bpc_trace = nullptr;
#define PS_BPC_TRACE(slot, val) do { if (g_bpc_trace && (threadIdx.x & 63) == 0) g_bpc_trace[(size_t)gwave * 4 + (slot)] = (val); } while (0)
#else
#define PS_BPC_TRACE(slot, val) do { } while (0)
#endif
constexpr int kMaxPlanes = 16;     // bit-planes a codeblock may have: supports MSB <= 15
// The encoder holds ONE bit-plane of its two columns in registers (4 VGPRs): the one being coded.  All
// planes are built in a single pass over the coefficients (8 at a time, in registers that are free
// before the plane loop), parked in a scratch array in HBM -- [plane][4][lane] dwords, 16 KB per wave,
// every access a 256-byte row -- and read back one plane per plane step, a plane ahead.  With 8 planes
// resident (round 1) the kernel needed 96 VGPRs and still spilled 17; now 8 waves fit a SIMD.
constexpr int kEncPassPlanes = 8;                       // planes built per pass over the coefficients
constexpr int kEncPlaneDwords = 4 * 64;                 // one plane of a wave: L.lo, L.hi, R.lo, R.hi x 64 lanes
constexpr int kEncScratchDwordsPerWave = kMaxPlanes * kEncPlaneDwords;

struct LutGeo {
    int nBp, nSub, cRef, cSign, cSig, prec;
    int nRef, nSig, nSign;         // section sizes; table = [ref | sig | sign]
};

struct BpcArgs {
    const void *coeffs_in;         // encoder: Mallat T[AW*AH]
    int32_t *coeffs_out;           // decoder: Mallat int32[AW*AH]
    int is_float;                  // encoder input is float (truncated toward zero on load)
    int c16;                       // encode frame paths: the Mallat array coeffs_in is int16, row stride AW
    int AW, AH, wl, nCB, ncx;
    int cb_base;                   // first codeblock of this launch (intra-frame striping), nCB = end
    const int32_t *lut;
    LutGeo g;
    int32_t *staging;              // DECODERS (staging form): int32[nCB*4096], the reference's array (BPCEngine.cu:2429-2441)
    // ENCODERS: uint16[nCB*4096] -- a staging word never holds more than 16 bits (a codeword, the MSB, a raw block's
    // (magnitude << 1 | sign) & 0xFFFF), and as 16-bit words the codeword stores of a frame are 20 MB instead of 40 and a
    // sector of them is evicted half as often before it is full: coder WRITE 101.8 -> 83.8 MB per 8K frame, 183.6 -> 186.4
    // Gpixel/s.  Word 0 of a codeblock = its MSB (or a raw block's word 0), word 1 + k = codeword k, as in the reference's
    // array; picsong_bpc_encode widens into the caller's int32 array (widen_staging_kernel).
    uint16_t *staging16;
    int32_t *sizes;                // int32[nCB]
    int *range_flag;               // set to 1 if a codeblock has MSB > 15
    uint32_t *plane_scratch;       // encoder: kEncScratchDwordsPerWave dwords per wave of the launch
    float k;                       // complexity-scalability factor (-k); > 0 only in BULK kernels
    int n_tables;                  // bit-plane tables laid back to back in `lut` (1 when k = 0)
    // batched launches (picsong_encode_frames): the grid covers `frames` frames of waves_per_frame waves
    // each; frame f reads coeffs_in + f * coef_z bytes, codes into staging + f * AW*AH and sizes + f * nCB.
    // 0 / 1 frames = a single frame (waves_per_frame unused)
    int frames, waves_per_frame;
    unsigned long long coef_z;
    // the frames of a batched launch are the three COMPONENTS of one RGB frame (picsong_encode_rgb_frame): frame f
    // codes with table lut_c[f] (same geometry); waves_per_frame is then a whole number of workgroups, a workgroup's
    // LDS copy of the table being its frame's.  lut_c[0] = nullptr: every frame uses `lut`
    const int32_t *lut_c[3];
    // decoder, frame paths (k = 0, -cp 2): the codewords are read from the packed stream itself -- no unpack launch, no
    // staging.  cw16 = the frame's stream (9 header shorts, nCB x (MSB, length), the codewords: BitStreamBuilder.cu:106-137),
    // cw16_offsets[cb] = the scan of the lengths and *cw16_total the stream's length as they give it (scan_stream_kernel):
    // a load stays inside those shorts -- the ring reads ahead of a codeblock's length, and the caller's buffer may end
    // with the stream -- and inside cw16_max, a worst-case stream, whatever damaged lengths claim.  Frame f of a batched
    // launch reads cw16 + f * cw16_stride shorts, cw16_offsets + f * nCB, cw16_total + f.  nullptr: the 32-bit staging
    // (picsong_bpc_decode).
    const uint16_t *cw16;
    const int32_t *cw16_offsets;
    const int32_t *cw16_total;
    unsigned long long cw16_stride;
    uint32_t cw16_max;
};

// ---- cross-lane helpers ---------------------------------------------------------------------
// value held by lane-1 / lane+1 of the same 32-lane half; 0 at the half's edge
// (== correctCBBorders, BPCEngine.cu:465-484).
__device__ __forceinline__ uint32_t from_prev32(uint32_t v, uint32_t t)
{
    uint32_t r = __builtin_amdgcn_update_dpp(0u, v, 0x138 /*wave_shr:1*/, 0xf, 0xf, true);
    return t == 0u ? 0u : r;
}
__device__ __forceinline__ uint32_t from_next32(uint32_t v, uint32_t t)
{
    uint32_t r = __builtin_amdgcn_update_dpp(0u, v, 0x130 /*wave_shl:1*/, 0xf, 0xf, true);
    return t == 31u ? 0u : r;
}
__device__ __forceinline__ uint32_t half_or(uint32_t v)
{
    v |= __shfl_xor(v, 16);
    v |= __shfl_xor(v, 8);
    v |= __shfl_xor(v, 4);
    v |= __shfl_xor(v, 2);
    v |= __shfl_xor(v, 1);
    return v;
}

// 64-bit row mask helpers.  X-form: lo = rows 0..31, hi = rows 32..63.
// W-form for half-pass hw (rows 32hw .. 32hw+31): bit (ii+1) = row 32hw+ii, bit 0 = row above,
// bit 33 = row below, zero outside the codeblock (BPCEngine.cu:785,831).
struct M64 { uint32_t lo, hi; };

__device__ __forceinline__ M64 to_w(M64 x, int hw)
{
    M64 w;
    if (hw == 0) { w.lo = x.lo << 1; w.hi = (x.lo >> 31) | (x.hi << 1); }
    else         { w.lo = (x.lo >> 31) | (x.hi << 1); w.hi = x.hi >> 31; }
    return w;
}
// rows (i-1, i, i+1) of a W-form mask as bits 0..2
__device__ __forceinline__ uint32_t triple(M64 w, uint32_t ii)
{
    return __builtin_amdgcn_alignbit(w.hi, w.lo, ii) & 7u;
}
__device__ __forceinline__ void w_set(M64 &w, uint32_t ii, uint32_t bit)
{
    uint64_t s = (uint64_t)bit << (ii + 1u);
    w.lo |= (uint32_t)s;
    w.hi |= (uint32_t)(s >> 32);
}
// the 32 rows of this half-pass back in X-form
__device__ __forceinline__ uint32_t w_rows(M64 w) { return __builtin_amdgcn_alignbit(w.hi, w.lo, 1u); }

// sign context, computeSignContext BPCEngine.cu:252-308.  c for (sign(h)+1)*3 + (sign(v)+1):
// h<0: v<0 7, v=0 5, v>0 1 | h=0: 3, 0, 2 | h>0: 0, 4, 6  -> 3 bits each, packed LSB first.
__device__ __forceinline__ uint32_t sign_ctx(int h, int v)
{
    const uint32_t K = 7u | (5u << 3) | (1u << 6) | (3u << 9) | (0u << 12) | (2u << 15) |
                       (0u << 18) | (4u << 21) | (6u << 24);
    const int hs = h < -1 ? -1 : (h > 1 ? 1 : h), vs = v < -1 ? -1 : (v > 1 ? 1 : v);   // v_med3_i32
    return (K >> (3 * (hs * 3 + vs + 4))) & 7u;
}

// findSubband BPCEngine.cu:143-170
__device__ __forceinline__ void find_subband(int x, int y, int AW, int AH, int wl, int &level, int &sb)
{
    level = wl; sb = 0;
    for (int a = 1; a <= wl; a++) {
        bool cx = x >= (AW >> a), cy = y >= (AH >> a);
        if (cx || cy) { level = a - 1; sb = cx ? (cy ? 2 : 0) : 1; break; }
    }
}

__device__ __forceinline__ uint32_t lut_at(const int32_t *lut, int idx, int total)
{
    idx = idx < 0 ? 0 : (idx >= total ? total - 1 : idx);
    return (uint32_t)lut[idx] & 0xFFu;
}

// S * p of the interval update: S < 2^16, p < 2^8.  Spelled as the instruction because the optimiser,
// once it knows p is a byte (refinement probability straight from the LDS table), rewrites the 24-bit
// multiply intrinsic into a generic 32-bit multiply and then selects the quarter-rate v_mul_lo_u32.
__device__ __forceinline__ uint32_t mul_u24(uint32_t a, uint32_t b)
{
#if defined(__AMDGCN__)
    uint32_t r;
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    return __umul24(a, b);          // CPU wave-emulator build of this header (tests/hipemu)
#endif
}

// Per-lane arithmetic coder state (arithmeticEncoder/Decoder BPCEngine.cu:371-442).
struct Coder {
    uint32_t L, S;      // interval lower bound / size; the DECODER keeps D = codeword - lower bound in L (dec_site_m)
    uint32_t slot;      // reserved codeword slot (staging index = 1 + slot)
    uint32_t cw;        // (unused)
    uint32_t cnt_lo, cnt_hi;   // codeword counters (codeStreamShared) of the codeblocks in lanes 0-31 /
                               // 32-63: wave-uniform, they live in SGPRs and are updated by SALU
    uint64_t emptym;           // encoder: ballot(S == 0) as of the end of the previous call site
    // decoder: the codeblocks' next codewords wait in an LDS ring (see dec_ring_*): window edges of the two
    // codeblocks (wave-uniform), the lane's ring (its codeblock's) and lane index inside its half
    uint32_t next_lo, next_hi;
    uint16_t *ring;
    uint32_t t;
    uint32_t *ldscnt;          // LDS form of the reservation: the lane's codeblock's codeword counter
    uint32_t ringaddr;         // LDS form: LDS byte address of the lane's ring (1 KB aligned)
    uint32_t pend;             // LDS form: the codeblock's counter as read at the start of the previous row (dec_ring_row)
    // where the ring is filled from (dec_ring_fill): element srcoff + k of srcbase is the lane's codeblock's codeword k
    // -- 32-bit staging words (picsong_bpc_decode: stage[1 + k]) or, src16, the 16-bit words of the packed stream
    // itself (the frame paths: no unpack launch, no staging); srclim = the last element a load may start at.
    // srcbase, srclim and src16 are wave-uniform.
    const void *srcbase;
    uint32_t srcoff, srclim, src16;
};

// LDS operations of the wave's other lanes have completed
__device__ __forceinline__ void wave_lds_done()
{
#if defined(__AMDGCN__)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#else
    (void)__builtin_amdgcn_ballot_w64(true);
#endif
}

// LDS form of the codeword slot reservation (see enc_reserve below): the default on the GPU; the CPU wave emulator
// of the tests builds the v_mbcnt form.
#ifndef PICSONG_ENC_LDS_RESERVE
#define PICSONG_ENC_LDS_RESERVE 1
#endif
// (the CPU wave emulator of the tests runs lanes as coroutines, not in lane order between two cross-lane
// operations: it builds the v_mbcnt form, which states the order explicitly)
#if PICSONG_ENC_LDS_RESERVE && defined(__AMDGCN__)
#define PS_ENC_LDS 1
#else
#define PS_ENC_LDS 0
#endif

// the LDS byte address of a pointer into a __shared__ array
__device__ __forceinline__ uint32_t lds_addr_of(const void *p)
{
#if defined(__AMDGCN__)
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)p;
#else
    (void)p;
    return 0u;
#endif
}

// ---- decoder: codeword ring ------------------------------------------------------------------------
// A reservation of the decoder is followed at once by the use of the codeword it reserved
// (arithmeticDecoder BPCEngine.cu:420-428), so a load from the staging array puts an L2 / HBM round trip
// (500-900 cycles) into the dependent chain of every call site that starts a codeword -- about 1100 times
// per wave.  Codewords are consumed in slot order, so the wave keeps a window of its two codeblocks'
// streams in LDS: 256 entries per codeblock, filled 64 at a time (two coalesced loads by the lanes of that
// half) up to an edge F (next_lo / next_hi) that dec_ring_keep holds at least 192 ahead of the codeblock's
// counter.  A reservation then reads its codeword with LDS latency.  The plane loops look after the window
// once per ROW (a row's call sites reserve at most 4 x 32 slots of a codeblock: two columns, bit and sign),
// not per call site: seven scalar instructions and a branch fewer at every site that starts a codeword, which
// nearly every site does for some lane.  (-k's row scan looks once per plane of a coefficient -- three sites --, -cp 3
// keeps the check in the call site.)
constexpr int kDecRing = 512;                 // 16-bit entries per codeblock: 1 KB
constexpr uint32_t kDecRingAhead = 192u;      // per-site check (dec_site_m<true>): exact counters
constexpr uint32_t kDecRingAheadRow = 256u;   // per-row check (dec_ring_row): counters one row old
// LDS form of the reservation (GPU build): the codeblock's counter counts BYTES of its 16-bit ring (2 per codeword),
// so that a lane's pre-add value masked to the ring's size IS its slot's offset in the ring; the window edges
// next_lo / next_hi are kept in the same unit.
#if PS_ENC_LDS
constexpr uint32_t kDecCntUnit = 2u;
#else
constexpr uint32_t kDecCntUnit = 1u;
#endif
__device__ __forceinline__ void dec_ring_fill(const Coder &c, uint32_t first)
{   // entries first .. first + 63 of the lane's codeblock.  What lies beyond the codeblock's length is never used, so a
    // load is only kept inside the buffer: the staging's last word, the stream's last pair of shorts.
    if (c.src16) {
        // the stream's own 16-bit words: one dword a lane, entries first + 2 t and first + 2 t + 1, copied as they lie
        // (a dword load from an address that is a multiple of 2, an aligned dword into the ring)
        const uint32_t e = first + 2u * c.t;
        uint32_t idx = c.srcoff + e;
        idx = idx < c.srclim ? idx : c.srclim;
        uint32_t v;
        __builtin_memcpy(&v, static_cast<const uint16_t *>(c.srcbase) + idx, 4);
        *reinterpret_cast<uint32_t *>(c.ring + (e & (kDecRing - 1))) = v;
    } else {
        const uint32_t e0 = first + c.t, e1 = first + 32u + c.t;
        uint32_t i0 = c.srcoff + e0, i1 = c.srcoff + e1;
        i0 = i0 < c.srclim ? i0 : c.srclim; i1 = i1 < c.srclim ? i1 : c.srclim;
        const int32_t *const w = static_cast<const int32_t *>(c.srcbase);
        const int32_t v0 = w[i0], v1 = w[i1];
        c.ring[e0 & (kDecRing - 1)] = (uint16_t)v0;
        c.ring[e1 & (kDecRing - 1)] = (uint16_t)v1;
    }
}
// codewords 0 .. 511 of both codeblocks
__device__ __forceinline__ void dec_ring_init(Coder &c)
{
    // (unrolled: the loads go out together, not as eight round trips one after the other)
#pragma unroll
    for (uint32_t f = 0; f < (uint32_t)kDecRing; f += 64u) dec_ring_fill(c, f);
    c.next_lo = c.next_hi = (uint32_t)kDecRing * kDecCntUnit;
    c.pend = 0u;
}
// the window of both codeblocks at least `ahead` codewords ahead of the counters cnt_lo / cnt_hi (codewords; exact or
// older values).  A fill overwrites entries next - 512 .. next - 449, all consumed: it happens only while
// next - cnt < ahead <= 256.
__device__ __forceinline__ void dec_ring_refill(Coder &c, uint32_t upper_mask, uint32_t cnt_lo, uint32_t cnt_hi,
                                                uint32_t ahead)
{
    bool lo = c.next_lo - cnt_lo * kDecCntUnit < ahead * kDecCntUnit, hi = c.next_hi - cnt_hi * kDecCntUnit < ahead * kDecCntUnit;     // wave-uniform
    while (lo || hi) {
        wave_lds_done();                                   // every lane has read the codewords of its earlier sites
        const uint32_t edge = (upper_mask ? c.next_hi : c.next_lo) / kDecCntUnit;
        if (upper_mask ? hi : lo) dec_ring_fill(c, edge);
        wave_lds_done();                                   // (a later reservation of another lane reads them)
        if (lo) c.next_lo = __builtin_amdgcn_readfirstlane(c.next_lo + 64u * kDecCntUnit);
        if (hi) c.next_hi = __builtin_amdgcn_readfirstlane(c.next_hi + 64u * kDecCntUnit);
        lo = c.next_lo - cnt_lo * kDecCntUnit < ahead * kDecCntUnit; hi = c.next_hi - cnt_hi * kDecCntUnit < ahead * kDecCntUnit;
    }
}
// per call site (dec_site_m<true>: -k's row scan, -cp 3): the exact counters c.cnt_lo / c.cnt_hi
__device__ __forceinline__ void dec_ring_keep(Coder &c, uint32_t upper_mask)
{
    const uint32_t a = c.next_lo - c.cnt_lo * kDecCntUnit, b = c.next_hi - c.cnt_hi * kDecCntUnit;
    if ((a < b ? a : b) < kDecRingAhead * kDecCntUnit) dec_ring_refill(c, upper_mask, c.cnt_lo, c.cnt_hi, kDecRingAhead);
}
// Once per ROW of call sites of the plane loops (a row reserves at most 4 x 32 slots of a codeblock: two columns, bit
// and sign).  LDS form: the call sites keep no scalar counters at all (two popcounts and two additions of the scalar
// unit at every site that starts a codeword, which nearly every site does for some lane); the codeblocks' LDS
// counters are read here instead, and used ONE ROW LATE -- the read of row r is consumed at row r + 1, so no wait
// for the LDS sits in a row's path: with v the value read at the start of row r - 1, the counter at the end of row
// r is at most v + 256, and the window is kept 256 codewords ahead of v.
__device__ __forceinline__ void dec_ring_row(Coder &c, uint32_t upper_mask)
{
#if PS_ENC_LDS
    const uint32_t vlo = __builtin_amdgcn_readlane(c.pend, 0), vhi = __builtin_amdgcn_readlane(c.pend, 32);   // bytes
    const uint32_t a = c.next_lo - vlo, b = c.next_hi - vhi;
    if ((a < b ? a : b) < kDecRingAheadRow * kDecCntUnit) dec_ring_refill(c, upper_mask, vlo / kDecCntUnit, vhi / kDecCntUnit, kDecRingAheadRow);
    c.pend = __hip_atomic_load(c.ldscnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
#else
    const uint32_t a = c.next_lo - c.cnt_lo, b = c.next_hi - c.cnt_hi;
    if ((a < b ? a : b) < kDecRingAheadRow) dec_ring_refill(c, upper_mask, c.cnt_lo, c.cnt_hi, kDecRingAheadRow);
#endif
}
// significance probabilities for contexts 0..8 packed as bytes: w0 = ctx 0-3, w1 = ctx 4-7, p8.
struct PlaneLut { uint32_t sig0, sig1, sig8, sign, ref, sig8x4; };

// The probability table of a codeblock sits in LDS as bytes (3360 at wl = 5): a plane's 14 entries
// are read with LDS latency.  Read from global memory instead, every plane would wait behind the
// codeword stores the wave has in flight (loads and stores drain through ONE in-order vmcnt).
constexpr int kLutLdsMax = 4672;       // bytes of one table: wl <= 7 -> 15 * (3*7 + 1) * 14 = 4620

struct LutView {
    const uint8_t *lds;            // this codeblock's table (table s of the bit-plane files when k > 0)
    const int32_t *glob;           // the whole table array, for indices outside the table (as lut_at)
    int total, glob_total, loff;   // entries of one table / of the array / offset of table s
};
// SINGLE: the array is ONE table and the LDS copy holds all of it (k = 0, -cp 3), so an index outside the table
// clamps into the LDS copy exactly as lut_at clamps it into the array -- no branch and no global fallback (which cost
// an exec-masked region with a 64-bit address per entry: 14 entries a plane, 140 vector and 110 scalar instructions)
// !SINGLE (-k > 0: table s of several): the LDS copy holds the table AND the kLutSlack entries behind it as lut_at would
// deliver them (the next table's first entries; the array's last entry past its end: bulk_setup).  No index the coders
// form is negative, and none reaches further than one bit-plane group past its section's end -- bit-plane 15 of a
// 15-plane table, at most 9 entries -- so the read is a plain LDS byte, no branch and no global fallback
// (which was an exec-masked region with two compares, a 64-bit address and a wait at EVERY call site of the bulk scan).
constexpr int kLutSlack = 16;
template <bool SINGLE = false>
__device__ __forceinline__ uint32_t lut_get(const LutView &v, int idx)
{
    if constexpr (SINGLE) return v.lds[idx < 0 ? 0 : (idx >= v.total ? v.total - 1 : idx)];
    return v.lds[idx];
}
template <bool SINGLE = false>
__device__ __forceinline__ PlaneLut plane_lut(const LutView &v, const LutGeo &g, int grp, int bp, int aux = 0)
{
    PlaneLut pl;
    const int ri = (grp * g.nBp + bp) * g.cRef;
    const int si = (grp * g.nBp + bp) * g.cSig + g.nRef + aux;
    const int gi = (grp * g.nBp + bp) * g.cSign + g.nRef + g.nSig + aux;
    pl.ref = lut_get<SINGLE>(v, ri);
    pl.sig0 = lut_get<SINGLE>(v, si + 0) | (lut_get<SINGLE>(v, si + 1) << 8) | (lut_get<SINGLE>(v, si + 2) << 16) | (lut_get<SINGLE>(v, si + 3) << 24);
    pl.sig1 = lut_get<SINGLE>(v, si + 4) | (lut_get<SINGLE>(v, si + 5) << 8) | (lut_get<SINGLE>(v, si + 6) << 16) | (lut_get<SINGLE>(v, si + 7) << 24);
    pl.sig8 = lut_get<SINGLE>(v, si + 8);
    pl.sign = lut_get<SINGLE>(v, gi + 0) | (lut_get<SINGLE>(v, gi + 1) << 8) | (lut_get<SINGLE>(v, gi + 2) << 16) | (lut_get<SINGLE>(v, gi + 3) << 24);
    pl.sig8x4 = pl.sig8 * 0x01010101u;
    return pl;
}
// k = 0: every codeblock of the workgroup uses table 0; every thread copies its share of it
__device__ __forceinline__ void lut_to_lds(const int32_t *lut, int total, uint8_t *lds)
{
    // Eight loads in flight a thread, no branch around a load (an index that stays inside).  The plain loop -- a load, its
    // wait, a byte store per trip -- was 13 round trips to the L2 one after the other at the start of every workgroup of
    // both coders (105 at the start of every -k > 0 wave: bulk_setup).
    const int nt = (int)blockDim.x;
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
    for (int base = (int)threadIdx.x; base < total; base += 8 * nt) {
        int32_t v[8];
#pragma unroll
        for (int q = 0; q < 8; q++) { const int j = base + q * nt; v[q] = lut[j < total ? j : total - 1]; }
#pragma unroll
        for (int q = 0; q < 8; q++) { const int j = base + q * nt; if (j < total) lds[j] = (uint8_t)((uint32_t)v[q] & 0xFFu); }
    }
    __syncthreads();
}
// Workgroup shape of the k = 0 coder kernels: 4 waves = 8 codeblocks, so that the dispatcher hands a
// CU's four SIMDs one wave each (with one-wave workgroups and room for 5 waves per SIMD it fills
// SIMD after SIMD, and a lone launch runs on unevenly loaded SIMDs); the waves share nothing but the
// LDS copy of the probability table.  The -k > 0 kernels keep one wave per workgroup (a table copy
// per codeblock).
#ifndef PICSONG_BPC_ENC_WG
#define PICSONG_BPC_ENC_WG 4
#endif
#ifndef PICSONG_BPC_DEC_WG
#define PICSONG_BPC_DEC_WG 4
#endif
constexpr int kBpcEncWgWaves = PICSONG_BPC_ENC_WG, kBpcDecWgWaves = PICSONG_BPC_DEC_WG;
// stores of the wave's other lanes to addresses this lane is about to overwrite have completed
__device__ __forceinline__ void wave_stores_issued()
{
#if defined(__AMDGCN__)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#else
    (void)__builtin_amdgcn_ballot_w64(true);        // emulator: lanes are coroutines, this joins them
#endif
}

// =============================================================================================
// Encoder, second formulation: contexts without neighbour exchange.
//
// The encoder knows every coefficient, so the significance state a coefficient SEES when the
// lock-step scan reaches it is a pure function of the data: a neighbour is significant if it
// became significant in an earlier plane (mask A), or in this plane (mask N = cur & ~A) AND it is
// visited before the coefficient.  With the reference's scan order (row by row; all left columns,
// then all right columns -- SPPEncoderLauncher BPCEngine.cu:770-843):
//     row above      : already visited            -> A | N
//     same row       : left coeff sees odd columns not yet visited  -> A
//                      right coeff sees even columns already visited -> A | N
//     row below      : not yet visited            -> A
// So per plane the 8-neighbour count (computeContext :222-230) of all 64 rows of a column is a
// carry-save addition of eight 64-bit row masks (4 result bit-planes n0..n3), and the sign context
// (computeSignContext :252-308) is a handful of mask operations giving its three bits c0..c2.
// The per-coefficient work shrinks to bit extraction + the arithmetic coder call; the two
// neighbour columns are fetched ONCE per plane by DPP instead of twice per row; rows in which no
// lane of the wave has anything to code are skipped by a wave-uniform row mask.
// Codeword slot order is untouched: call sites are still visited in the lock-step order.
// =============================================================================================

struct U64 { uint32_t lo, hi; };
__device__ __forceinline__ U64 u_and(U64 a, U64 b) { return U64{ a.lo & b.lo, a.hi & b.hi }; }
__device__ __forceinline__ U64 u_or(U64 a, U64 b) { return U64{ a.lo | b.lo, a.hi | b.hi }; }
__device__ __forceinline__ U64 u_xor(U64 a, U64 b) { return U64{ a.lo ^ b.lo, a.hi ^ b.hi }; }
__device__ __forceinline__ U64 u_andn(U64 a, U64 b) { return U64{ a.lo & ~b.lo, a.hi & ~b.hi }; }   // a & ~b
__device__ __forceinline__ U64 u_prev(U64 a, uint32_t t) { return U64{ from_prev32(a.lo, t), from_prev32(a.hi, t) }; }
__device__ __forceinline__ U64 u_next(U64 a, uint32_t t) { return U64{ from_next32(a.lo, t), from_next32(a.hi, t) }; }

// per-column side information for the 32 rows of one half-pass, bit-sliced
struct ColHalf {
    uint32_t n0, n1, n2, n3;    // significance context (0..8)
    uint32_t c1, c2;            // sign context bits 1, 2 (LUT index = c >> 1)
    uint32_t s2;                // sign symbol = sign bit ^ context bit 0
};
__device__ __forceinline__ void fa32(uint32_t a, uint32_t b, uint32_t c, uint32_t &s, uint32_t &cy)
{
    uint32_t x = a ^ b;
    s = x ^ c;
    cy = (a & b) | (c & x);
}
__device__ __forceinline__ uint32_t w_of(U64 a, int hw) { return hw ? a.hi : a.lo; }
__device__ __forceinline__ uint32_t up_of(U64 a, int hw) { return hw ? ((a.hi << 1) | (a.lo >> 31)) : (a.lo << 1); }
__device__ __forceinline__ uint32_t dn_of(U64 a, int hw) { return hw ? (a.hi >> 1) : ((a.lo >> 1) | (a.hi << 31)); }

// x1..x3: row above (post-pass state), x4,x5: same row, x6..x8: row below (pre-pass state);
// sign neighbours: (us,ug) up, (ds,dg) down, (ls,lg) left, (rs,rg) right -- visible-significance
// and sign masks, already shifted onto the coefficient's row.
__device__ __forceinline__ ColHalf make_col(uint32_t x1, uint32_t x2, uint32_t x3, uint32_t x4, uint32_t x5,
                                            uint32_t x6, uint32_t x7, uint32_t x8, uint32_t us, uint32_t ug,
                                            uint32_t ds, uint32_t dg, uint32_t ls, uint32_t lg, uint32_t rs,
                                            uint32_t rg, uint32_t self_sgn)
{
    ColHalf r;
    uint32_t s1, c1, s2, c2, s3, c3, c4, t1, d1, d2;
    fa32(x1, x2, x3, s1, c1);
    fa32(x4, x5, x6, s2, c2);
    s3 = x7 ^ x8; c3 = x7 & x8;
    fa32(s1, s2, s3, r.n0, c4);
    fa32(c1, c2, c3, t1, d1);
    r.n1 = t1 ^ c4; d2 = t1 & c4;
    r.n2 = d1 ^ d2; r.n3 = d1 & d2;
    // contributions: +1 significant & positive, -1 significant & negative (BPCEngine.cu:302-305)
    uint32_t pu = us & ~ug, nu = us & ug, pd = ds & ~dg, nd = ds & dg;
    uint32_t pl = ls & ~lg, nl = ls & lg, pr = rs & ~rg, nr = rs & rg;
    uint32_t hp = (pl | pr) & ~(nl | nr), hn = (nl | nr) & ~(pl | pr);
    uint32_t vp = (pu | pd) & ~(nu | nd), vn = (nu | nd) & ~(pu | pd);
    // context table :258-290: h0: v0 0, v+ 2, v- 3 | h+: v0 4, v+ 6, v- 0 | h-: v0 5, v+ 1, v- 7
    uint32_t same = (hp & vp) | (hn & vn);
    uint32_t c0 = hn | (vn & ~hp);
    r.c1 = (~(hp | hn) & (vp | vn)) | same;
    r.c2 = ((hp | hn) & ~(vp | vn)) | same;
    r.s2 = self_sgn ^ c0;
    return r;
}

// OR over the 64 lanes, wave-uniform result.  DPP row shifts + row broadcasts (6 VALU, no LDS crossbar:
// the ds_bpermute butterfly this replaces cost six LDS round trips at every half-pass of every plane).
__device__ __forceinline__ uint32_t wave_or32(uint32_t v)
{
#if defined(__AMDGCN__)
    v |= __builtin_amdgcn_update_dpp(0u, v, 0x111 /*row_shr:1*/, 0xf, 0xf, true);
    v |= __builtin_amdgcn_update_dpp(0u, v, 0x112 /*row_shr:2*/, 0xf, 0xf, true);
    v |= __builtin_amdgcn_update_dpp(0u, v, 0x114 /*row_shr:4*/, 0xf, 0xf, true);
    v |= __builtin_amdgcn_update_dpp(0u, v, 0x118 /*row_shr:8*/, 0xf, 0xf, true);       // lane 15 of a row: the row's OR
    v |= __builtin_amdgcn_update_dpp(0u, v, 0x142 /*row_bcast:15*/, 0xa, 0xf, false);   // rows 1, 3 take rows 0, 2
    v |= __builtin_amdgcn_update_dpp(0u, v, 0x143 /*row_bcast:31*/, 0xc, 0xf, false);   // rows 2, 3 take lane 31
    return __builtin_amdgcn_readlane(v, 63);
#else
    v |= __shfl_xor(v, 32);
    v |= __shfl_xor(v, 16);
    v |= __shfl_xor(v, 8);
    v |= __shfl_xor(v, 4);
    v |= __shfl_xor(v, 2);
    v |= __shfl_xor(v, 1);
    return __builtin_amdgcn_readfirstlane(v);
#endif
}
// OR over the 32 lanes of each half, every lane receiving its half's value
__device__ __forceinline__ uint32_t half_or_dpp(uint32_t v, uint32_t upper_mask)
{
#if defined(__AMDGCN__)
    v |= __builtin_amdgcn_update_dpp(0u, v, 0x111, 0xf, 0xf, true);
    v |= __builtin_amdgcn_update_dpp(0u, v, 0x112, 0xf, 0xf, true);
    v |= __builtin_amdgcn_update_dpp(0u, v, 0x114, 0xf, 0xf, true);
    v |= __builtin_amdgcn_update_dpp(0u, v, 0x118, 0xf, 0xf, true);
    v |= __builtin_amdgcn_update_dpp(0u, v, 0x142, 0xa, 0xf, false);                    // lanes 31 / 63: the halves' ORs
    const uint32_t lo = __builtin_amdgcn_readlane(v, 31), hi = __builtin_amdgcn_readlane(v, 63);
    return lo ^ (upper_mask & (lo ^ hi));
#else
    (void)upper_mask;
    return half_or(v);
#endif
}

// A lane mask (all ones / zero) the optimiser must keep as a register: it otherwise turns every
// `mask & x` back into a select on the condition the mask came from (a move and a select where one
// v_and does), at the three or four slot reservations of every scanned coefficient.
__device__ __forceinline__ uint32_t opaque_mask(uint32_t m)
{
#if defined(__AMDGCN__)
    asm volatile("" : "+v"(m));
#endif
    return m;
}

// ballot(v == 0)
__device__ __forceinline__ uint64_t zero_mask(uint32_t v)
{
#if defined(__AMDGCN__)
    uint64_t m;
    asm("v_cmp_eq_u32_e64 %0, 0, %1" : "=s"(m) : "v"(v));
    return m;
#else
    return __builtin_amdgcn_ballot_w64(v == 0u);
#endif
}

// Slot reservation: m = ballot of the lanes that need a codeword.  The rank of a lane among the
// requesting lanes of ITS codeblock is v_mbcnt over the two ballot words; v_mbcnt_lo counts all of the
// lower word for lanes 32-63, so those start at (their counter - popcount(lower word)).  The counters
// are scalars: the ballot, its popcounts, the counter updates and their clamps are all SALU work.
__device__ __forceinline__ void reserve_enc(Coder &c, bool need, uint64_t m, uint32_t upper_mask)
{
    const uint32_t mlo = (uint32_t)m, mhi = (uint32_t)(m >> 32);
    const uint32_t nlo = (uint32_t)__builtin_popcount(mlo), nhi = (uint32_t)__builtin_popcount(mhi);
    const uint32_t base = c.cnt_lo + (upper_mask & (c.cnt_hi - nlo - c.cnt_lo));      // v_and + v_add, scalars folded by SALU
    uint32_t s = __builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, base));
    s = s > 4094u ? 4094u : s;
    if (need) { c.L = 0u; c.S = 0xFFFFu; c.slot = s; }
    const uint32_t a = c.cnt_lo + nlo, b = c.cnt_hi + nhi;
    // (readfirstlane: states the uniformity; without it the decoder's counters ended up as per-lane
    // VGPR values updated by a dozen vector instructions at every call site)
    c.cnt_lo = __builtin_amdgcn_readfirstlane(a > 4095u ? 4095u : a);
    c.cnt_hi = __builtin_amdgcn_readfirstlane(b > 4095u ? 4095u : b);
}

// ---- encoder call site (arithmeticEncoder BPCEngine.cu:371-399) ------------------------------------
// Per-lane state of the encoder: interval (L, S) and the BYTE offset `off`, from the wave's staging
// base, of the codeword slot the lane has reserved.  Wave state: the two codeblocks' codeword counters
// (SGPRs), the ballot of exhausted intervals, and constants of the wave.
// PICSONG_ENC_LDS_RESERVE (default 1): a codeword slot is taken with ONE LDS atomic -- ds_add_rtn_u32 on the
// codeblock's codeword counter, every requesting lane adding 1 -- instead of ballot popcounts, v_mbcnt ranks
// and scalar counters.  The reference's order (requesting lanes of a call site in ascending lane order,
// BPCEngine.cu:380-393) is the order in which the LDS hands the pre-add values back to the lanes of one
// instruction that hit one address; that is how gfx950 resolves the conflict, it is checked against the
// v_mbcnt form by tests/test_gpu_parity.py::test_lds_atomic_reservation_order and by every codestream
// comparison with the oracle, and PICSONG_ENC_LDS_RESERVE=0 builds the v_mbcnt form.  The returned value is
// not needed until the lane's NEXT reservation (it only addresses the deferred store), so no wait for the
// LDS sits in the call site's dependent chain.
// (PICSONG_ENC_LDS_RESERVE / PS_ENC_LDS: defined at the top of this header)

// the encoders' staging words are 16-bit (BpcArgs::staging16): bytes per word, bytes per codeblock
constexpr uint32_t kStageBytes = 2u, kStageCb = 4096u * kStageBytes;
struct EncCoder {
    uint32_t L, S, off;
    uint32_t slot;              // LDS form: raw pre-add value of the lane's reservation (clamped when used)
    uint32_t *ldscnt;           // LDS form: the lane's codeblock's codeword counter
    uint32_t cnt_lo, cnt_hi;
    uint64_t emptym;            // ballot(S == 0) as of the end of the previous call site
    uint32_t slot0;             // byte offset of slot 0 of the lane's codeblock in the wave's staging: half * 8192 + 2
    uint32_t lim;               // LDS form: slot0 + 2 * 4094, the byte offset of the codeblock's last slot
    uint32_t pone;              // 1 << prec: the "probability" that leaves an idle lane's interval alone
    char *stw;                  // staging of the wave's first codeblock (wave-uniform)
};

// Slot reservation for the lanes in m (!= 0): as reserve_enc, plus the store of the codeword the lane
// has just finished.  The reference stores L the moment the interval is exhausted (:395-397) and again
// at the flush (:1719); a lane writes a slot only with the value it holds when it leaves the slot, so the
// store can wait until the lane moves on -- here, where an exec-masked region exists anyway -- and the
// call site loses a compare-and-branch region of its own.  A lane's first reservation stores the L = 0
// it starts with to word 0 of its codeblock's staging (off starts there), which the kernel's epilogue
// overwrites with the MSB after the wave's stores have drained.
__device__ __forceinline__ void enc_reserve(EncCoder &c, uint64_t m, uint32_t upper_mask)
{
#if PS_ENC_LDS
    (void)upper_mask;
    if (__builtin_amdgcn_inverse_ballot_w64(m)) {
        // the codeword this lane has just finished goes to the slot it reserved last time (slot starts at
        // -1: word 0 of the staging, see above)
        // (the codeblock's LDS counter counts BYTES of the wave's staging from slot0 on, 2 a codeword: what a lane
        // gets back is its slot's byte offset, and the guard against a 4096th codeword is one v_min with the lane's
        // `lim` -- no shift-and-add, no literal)
        // (Round 3 also wrote this region as its instructions -- exec set instead of saved and restored, no skip of an
        // empty region: 0.4 % faster and WRONG.  The counter's return lands in c.slot some hundred cycles after the
        // ds_add, a wait the compiler places before the next use only if it issued the ds_add itself; blind to it, it
        // moved the epilogue's min(c.slot, lim) above the hand-written wait, and one codeblock in some ten thousand
        // flushed its last codeword to a stale slot -- caught by the 16K frame's codestream against the oracle, by none
        // of the smaller cases.  The atomic stays the compiler's.)
        const uint32_t sl = c.slot < c.lim ? c.slot : c.lim;
        *reinterpret_cast<uint16_t *>(c.stw + sl) = (uint16_t)c.L;
        c.slot = __hip_atomic_fetch_add(c.ldscnt, kStageBytes, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        c.L = 0u; c.S = 0xFFFFu;
    }
#else
    const uint32_t mlo = (uint32_t)m, mhi = (uint32_t)(m >> 32);
    const uint32_t nlo = (uint32_t)__builtin_popcount(mlo), nhi = (uint32_t)__builtin_popcount(mhi);
    const uint32_t base = c.cnt_lo + (upper_mask & (c.cnt_hi - nlo - c.cnt_lo));
    if (__builtin_amdgcn_inverse_ballot_w64(m)) {
        *reinterpret_cast<uint16_t *>(c.stw + c.off) = (uint16_t)c.L;
        uint32_t s = __builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, base));
        s = s > 4094u ? 4094u : s;
        c.off = s * kStageBytes + c.slot0;
        c.L = 0u; c.S = 0xFFFFu;
    }
    const uint32_t a = c.cnt_lo + nlo, b = c.cnt_hi + nhi;
    c.cnt_lo = __builtin_amdgcn_readfirstlane(a > 4095u ? 4095u : a);
    c.cnt_hi = __builtin_amdgcn_readfirstlane(b > 4095u ? 4095u : b);
#endif
}

// One call site.  onm = ballot of the lanes that code a symbol here, onem = those of them whose symbol
// is 1 (both wave-uniform SGPR pairs the callers have anyway):  a0 = (S * p) >> prec;  a lane coding a 1
// takes S' = S - a0 - 1, L' = L + a0 + 1 (a = a0 + sym), a lane coding a 0 takes S' = a0.
//
// What this kernel is bound by is vector-instruction ISSUE, and on gfx950 issue cost depends on the
// instruction (tools/valu_probe, profiles/r02_valu_probe.txt): and / or / xor / add / sub / mov on VGPR,
// inline or literal operands go at one per ~2.4 cycles per SIMD, but shifts, v_min, every VOP3 form
// (v_cndmask_e64, v_mad, v_alignbit, v_perm, v_bfe ...), every compare, v_mul_u32_u24, DPP moves and any
// instruction with an SGPR operand at one per ~4.2.  So the update is written as the instructions
// themselves, over exec masks instead of selects: 2 half-rate + 4 full-rate + the exhausted compare,
// where the compiler's select form needs 6 half-rate + 2 full-rate.
// J >= 0: `p` is a word of four probabilities and this site takes its byte J -- selected by the multiply itself
// (SDWA src1_sel), so no extraction instruction exists; J < 0: `p` is the probability.
template <int J = -1>
__device__ __forceinline__ void enc_update(EncCoder &c, uint64_t onm, uint64_t onem, uint32_t p, uint32_t prec)
{
#if defined(__AMDGCN__)
    // Call sites sit in wave-uniform control flow of kernels launched with whole waves: exec is all ones on
    // entry, so it is not saved, the product and the shift run in every lane (an idle lane's `a` is never
    // used), and exec goes back to all ones.  Scalar instructions are not free either -- the CU's one
    // scalar unit issues about one per cycle for its four SIMDs (tools/valu_probe) -- so: three exec writes.
    uint32_t a;
    uint64_t em;
#define PS_ENC_UPDATE_TAIL                                                                           \
        "v_lshrrev_b32 %[a], %[pr], %[a]\n\t"          /* a0 */                                          \
        "s_mov_b64 exec, %[one]\n\t"                                                                    \
        "v_add_u32 %[a], 1, %[a]\n\t"                   /* lanes coding a 1: a = a0 + 1 */               \
        "v_add_u32 %[L], %[L], %[a]\n\t"                /*   L += a */                                   \
        "v_sub_u32 %[a], %[S], %[a]\n\t"                /*   a = S - a  (their new S) */                 \
        "s_mov_b64 exec, %[on]\n\t"                                                                     \
        "v_mov_b32 %[S], %[a]\n\t"                      /* lanes coding a 0 still hold a0 in a */        \
        "s_mov_b64 exec, -1\n\t"                                                                        \
        "v_cmp_eq_u32_e64 %[em], 0, %[S]"
#define PS_ENC_UPDATE(MUL)                                                                           \
    asm volatile(MUL PS_ENC_UPDATE_TAIL                                                              \
                 : [S] "+v"(c.S), [L] "+v"(c.L), [a] "=&v"(a), [em] "=s"(em)                         \
                 : [on] "s"(onm), [one] "s"(onem), [p] "v"(p), [pr] "s"(prec))
    if constexpr (J == 0) PS_ENC_UPDATE("v_mul_u32_u24_sdwa %[a], %[S], %[p] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t");
    else if constexpr (J == 1) PS_ENC_UPDATE("v_mul_u32_u24_sdwa %[a], %[S], %[p] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t");
    else if constexpr (J == 2) PS_ENC_UPDATE("v_mul_u32_u24_sdwa %[a], %[S], %[p] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n\t");
    else if constexpr (J == 3) PS_ENC_UPDATE("v_mul_u32_u24_sdwa %[a], %[S], %[p] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n\t");
    else PS_ENC_UPDATE("v_mul_u32_u24 %[a], %[S], %[p]\n\t");
#undef PS_ENC_UPDATE
#undef PS_ENC_UPDATE_TAIL
    c.emptym = em;
#else
    if (J >= 0) p = (p >> (8 * (J < 0 ? 0 : J))) & 0xFFu;
    const uint32_t pe = __builtin_amdgcn_inverse_ballot_w64(onm) ? p : c.pone;     // (S * pone) >> prec == S
    const uint32_t a0 = mul_u24(c.S, pe) >> prec;
    const bool one = __builtin_amdgcn_inverse_ballot_w64(onem);
    c.S = one ? c.S + ~a0 : a0;
    c.L = one ? c.L + a0 + 1u : c.L;
    c.emptym = zero_mask(c.S);
#endif
}
template <int J = -1>
__device__ __forceinline__ void enc_site2(EncCoder &c, uint64_t onm, uint64_t onem, uint32_t p, uint32_t prec,
                                          uint32_t upper_mask)
{
    const uint64_t m = c.emptym & onm;
    if (m != 0ull) enc_reserve(c, m, upper_mask);
    enc_update<J>(c, onm, onem, p, prec);
}
// generic form (bulk scan): `inact` = 1 for lanes that sit the site out
__device__ __forceinline__ void enc_site(EncCoder &c, uint32_t inact, uint32_t sym, uint32_t p, uint32_t prec,
                                         uint32_t upper_mask, int32_t *)
{
    const uint64_t onm = __builtin_amdgcn_ballot_w64(inact == 0u);
    enc_site2(c, onm, onm & __builtin_amdgcn_ballot_w64(sym != 0u), p, prec, upper_mask);
}

__device__ __forceinline__ uint32_t rotr32(uint32_t v, uint32_t sh) { return __builtin_amdgcn_alignbit(v, v, sh); }
__device__ __forceinline__ uint32_t bfi32(uint32_t mask, uint32_t a, uint32_t b) { return (a & mask) | (b & ~mask); }

// ---- probabilities of FOUR rows at a time -----------------------------------------------------------
// The context of a row is spread over the bit-sliced masks n0..n3 (c1, c2 for the sign); gathered row by
// row it costs three or four rotates, as many selects and two byte permutes per coefficient (11 half-rate
// instructions).  Gathered for rows 4g .. 4g+3 at once: a mask's nibble times 0x204081 puts bit i at bit 8i
// (the four shifted copies land on 16 distinct bits, so no carry forms), three such words make the four
// rows' byte selectors, ONE v_perm looks up four probabilities, and a row takes its byte with one v_bfe.
// nibble g4/4 of `mask`, bit i of it at bit 8i + SH of the result (SH = 0, 1, 2); the other product bits are
// strays the callers mask off
template <int SH>
__device__ __forceinline__ uint32_t spread4_raw(uint32_t mask, uint32_t g4)
{
    return mul_u24((mask >> g4) & 0xFu, 0x204081u << SH);              // 0x810204 still fits 24 bits
}
__device__ __forceinline__ uint32_t spread4(uint32_t mask, uint32_t g4) { return spread4_raw<0>(mask, g4) & 0x01010101u; }
// significance probabilities (computeContext BPCEngine.cu:222-230 + the LUT read) of rows g4 .. g4+3
__device__ __forceinline__ uint32_t sig_probs4(const ColHalf &cp, const PlaneLut &pl, uint32_t g4)
{
    // the three context bits land on bits 0, 1, 2 of the rows' bytes by the multiplier's own shift; two
    // bit-field inserts keep exactly those (strays of one product never reach the bits taken from another)
    uint32_t sel = bfi32(0x01010101u, spread4_raw<0>(cp.n0, g4), spread4_raw<1>(cp.n1, g4));
    sel = bfi32(0x03030303u, sel, spread4_raw<2>(cp.n2, g4)) & 0x07070707u;      // bytes 0..7: the context
    const uint32_t p07 = __builtin_amdgcn_perm(pl.sig1, pl.sig0, sel);
    // context 8 (n3 set => n0 = n1 = n2 = 0): those bytes take p8 -- a second permute whose selector is the
    // identity (bytes 0..3 = p07's) plus 4 where n3 is set (bytes 4..7 = p8 four times)
    const uint32_t sel8 = (spread4(cp.n3, g4) << 2) | 0x03020100u;
    return __builtin_amdgcn_perm(pl.sig8x4, p07, sel8);
}
// sign probabilities (computeSignContext :252-308: LUT index c >> 1 = c2 c1) of rows g4 .. g4+3
__device__ __forceinline__ uint32_t sign_probs4(const ColHalf &cp, const PlaneLut &pl, uint32_t g4)
{
    const uint32_t sel = bfi32(0x01010101u, spread4_raw<0>(cp.c1, g4), spread4_raw<1>(cp.c2, g4)) & 0x03030303u;
    return __builtin_amdgcn_perm(0u, pl.sign, sel);
}

// The row's bit as a VGPR value: `x & rowbit` with the bit in an SGPR issues at half rate (any SGPR operand
// does), with both operands in VGPRs at full rate; one move per row serves its five to seven mask tests.
__device__ __forceinline__ uint32_t vgpr_of(uint32_t s)
{
#if defined(__AMDGCN__)
    uint32_t v;
    asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"(s));
    return v;
#else
    return s;
#endif
}

// x <<= 1 in every lane, and the ballot of the bits shifted out: v_add_co_u32 x, m, x, x -- ONE half-rate
// instruction where testing a row's bit takes an and and a compare.  With a column's 32-row mask bit-reversed
// (row 0 at bit 31) a half-pass that visits its rows in order reads each row's lane mask off the carry.
// (The refinement pass's dense half-passes use it: coder alone 0.260 -> 0.254 ms.  The same for the significance pass
// -- three masks a column -- measured slower, 0.267 ms: its rows thin out as the planes go down, and a row nobody
// codes still pays its six shifts.)
__device__ __forceinline__ uint64_t shl_carry(uint32_t &x)
{
#if defined(__AMDGCN__)
    uint64_t m;
    asm volatile("v_add_co_u32 %0, %1, %0, %0" : "+v"(x), "=s"(m));
    return m;
#else
    const uint64_t m = __builtin_amdgcn_ballot_w64((x >> 31) != 0u);
    x <<= 1;
    return m;
#endif
}
// x = (x << 1) | (the lane's bit of m): v_addc_co_u32 x, -, x, x, m -- the decoder's dense half-passes collect a
// column's decoded bits this way (row 0 first, so the word comes out bit-reversed)
__device__ __forceinline__ void shl_in(uint32_t &x, uint64_t m)
{
#if defined(__AMDGCN__)
    uint64_t co;
    asm volatile("v_addc_co_u32 %0, %1, %0, %0, %2" : "+v"(x), "=s"(co) : "s"(m));
#else
    x = (x << 1) | (__builtin_amdgcn_inverse_ballot_w64(m) ? 1u : 0u);
#endif
}
__device__ __forceinline__ uint32_t bitrev32(uint32_t x)
{
#if defined(__AMDGCN__)
    return __builtin_bitreverse32(x);
#else
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
    x = ((x >> 8) & 0x00FF00FFu) | ((x & 0x00FF00FFu) << 8);
    return (x >> 16) | (x << 16);
#endif
}

// One coefficient of the significance propagation pass.  rowbit = 1 << ii (VGPR); J = ii & 3: the row's
// byte in the group's probability words P4 (significance) and Q4 (sign).  A: significant-before mask
// of the column's 32 rows (all ones for an idle half: never on); N: becomes significant in this plane (0
// for an idle half).
template <int J>
__device__ __forceinline__ void enc_spp_coeff(EncCoder &c, uint32_t rowbit, uint32_t A, uint32_t N,
                                              uint32_t s2, uint32_t P4, uint32_t Q4, uint32_t prec, uint32_t upper_mask)
{
    const uint64_t onm = __builtin_amdgcn_ballot_w64((A & rowbit) == 0u);
    if (onm == 0ull) return;                                // no lane has this column's coefficient to code
    const uint64_t onem = __builtin_amdgcn_ballot_w64((N & rowbit) != 0u);       // N is a subset of ~A
    enc_site2<J>(c, onm, onem, P4, prec, upper_mask);
    if (onem != 0ull)
        enc_site2<J>(c, onem, onem & __builtin_amdgcn_ballot_w64((s2 & rowbit) != 0u), Q4, prec, upper_mask);
}

__device__ __forceinline__ uint32_t dec_site(Coder &c, uint32_t inact, uint32_t p, uint32_t prec,
                                             uint32_t upper_mask, const int32_t *stage);
template <bool KEEP>
__device__ __forceinline__ uint64_t dec_site_m(Coder &c, bool on, uint64_t onm, uint32_t p, uint32_t prec,
                                               uint32_t upper_mask, const int32_t *stage, bool &one);
__device__ __forceinline__ void sign_table2_fill(uint8_t *tab, uint32_t lane);

// =============================================================================================
// Complexity-scalable mode, -k > 0 (Encode BPCEngine.cu:1684-1716, Decode :1794-1835,
// encode/decodeBulkMode :1640-1662 and callees :1285-1634).  A codeblock codes its planes
// MSB .. cbp with the ordinary two passes and the planes below cbp = floor(MSB * k / L2Norm) in ONE
// row-major scan: per coefficient all remaining planes, three lock-step call sites per plane
// (refinement if already significant, else significance, then sign).  The LUT is table
// s = min(cbp, MSB) of the bit-plane files.
//
// The bulk scan needs, per neighbour, "is it significant, since which plane, which sign".  The
// reference keeps that in flag bits of 128 live coefficient words per lane; here every coefficient
// the scan has to look at is one packed word pw built when the scan reaches its row:
//     bit 0 = sign, bit 1 = significant before the bulk scan, bits 2.. = the B+1 low magnitude bits
//     (only once the scan has PROCESSED the coefficient; 0 before)
// so that  computeContextBulk's "plane field >= B"  ==  bit1 | bit(2+B)   (:236-243; also == the
// plain count computeContext takes when B = 0), and computeSignContextBulk's "significant and plane
// field >= q"  ==  bit1 | (pw >> (2+q)) != 0   (:311-323).  A lane keeps the processed words of the
// row above for its two columns; the neighbour lanes' words come by DPP, the row below is
// unprocessed by construction.  The plane-q probabilities are read from an LDS copy of table s.
// =============================================================================================

constexpr int kBulkLutMax = kLutLdsMax;

// L2Norm, BPC/BPCEngine.cuh:158-169
__device__ __forceinline__ float l2norm(int level, int col)
{
    const float T[10][4] = {
        { 1.965908f, 1.0112865f, 1.0112865f, 0.52021784f }, { 4.1224113f, 1.9968134f, 1.9968134f, 0.96721643f },
        { 8.416739f, 4.1833673f, 4.1833673f, 2.0792568f }, { 16.935543f, 8.534108f, 8.534108f, 4.3004827f },
        { 33.924816f, 17.166693f, 17.166693f, 8.686718f }, { 67.87687f, 34.385098f, 34.385098f, 17.41882f },
        { 135.76744f, 68.7964f, 68.7964f, 34.860676f }, { 271.5416f, 137.60588f, 137.60588f, 69.73287f },
        { 543.0866f, 275.21814f, 275.21814f, 139.47136f }, { 1086.1624f, 550.43286f, 550.43286f, 278.94202f } };
    return T[level > 9 ? 9 : level][col];
}

// consecutiveBitplanes :1684-1692 for the codeblock of this half (level / subband of its lane 0)
__device__ __forceinline__ int consecutive_bitplanes(int msb, float k, int level, int sb, int wl)
{
    const float nrm = (wl == level) ? l2norm(level - 1 > 0 ? level - 1 : 0, 0) : l2norm(level, 3 - sb);
    const float q = k / nrm;
    const int c = (int)floorf((float)msb * q);
    return c > 0 ? c : 0;
}

struct BulkLane {
    int Bh;                        // first (highest) bulk plane of this lane's codeblock, -1 = none
    uint32_t ref0, sig0, sign0;    // LDS byte index of the plane-0 entries of the lane's LUT group
    uint32_t cRef, cSig, cSign;    // contexts per plane
    LutView v;                     // this codeblock's table
    const uint8_t *sgt;            // the sign table (sign_table2_fill)
    int bh_lo, bh_hi;              // Bh of the codeblocks in lanes 0-31 / 32-63 (wave-uniform)
};

__device__ __forceinline__ uint32_t bulk_lut(const BulkLane &b, uint32_t idx) { return b.v.lds[idx]; }   // (the LDS copy's slack: lut_get)
// computeContextBulk / computeContext term of one neighbour word
__device__ __forceinline__ uint32_t bulk_cc(uint32_t pw, uint32_t sh) { return ((pw >> 1) | (pw >> sh)) & 1u; }
// computeSignContextBulk term of one neighbour word at plane q: -1 / 0 / +1
__device__ __forceinline__ int bulk_sc(uint32_t pw, uint32_t q)
{
    const bool on = ((pw >> 1) & 1u) != 0u || (pw >> (2u + q)) != 0u;
    return on ? ((pw & 1u) ? -1 : 1) : 0;
}

// All remaining planes of ONE coefficient (encodeBulkProcessing :1285-1314 / decodeBulkProcessing
// :1454-1500).  u: the coefficient's unprocessed word; low: its low magnitude bits (encoder).
// Returns the processed word.
// Round 4: the lanes' roles at a plane are LANE MASKS (scalar registers) -- refinement = on & significant, significance =
// on & ~significant, sign = became significant -- combined by scalar instructions; the encoder reads a plane's bits off
// the carries of its shifted low word (shl_carry) and forms its sign context ONCE per coefficient, at the plane where the
// coefficient becomes significant (its own top low bit: the only plane at which the lane codes a sign), the decoder
// shifts its decoded bits in (shl_in); the call sites are the two-pass kernels' (enc_site2 / dec_site_m).  Before, every
// site rebuilt its masks from 0 / 1 integers and computed the sign context per site: ~30 vector instructions a site.
template <bool DEC, class CT>
__device__ __forceinline__ uint32_t bulk_coeff(CT &c, uint32_t u, uint32_t low, uint32_t ctx, uint32_t up,
                                               uint32_t lf, uint32_t rt, uint32_t dn, const BulkLane &b, int Bmax,
                                               uint32_t prec, uint32_t upper_mask,
                                               int32_t *st)
{
    uint64_t sigm = __builtin_amdgcn_ballot_w64((u & 2u) != 0u);
    uint32_t neg = u & 1u;
    uint32_t lowx = 0u, acc = 0u, saddr = 0u;
    uint64_t ssymm = 0ull;
    if constexpr (!DEC) {
        lowx = low << (31 - Bmax);                          // plane Bmax at bit 31: a plane's bit is the next carry
        // computeSignContextBulk :311-323 at the plane where THIS lane's coefficient becomes significant (its top low bit),
        // by the decoders' table (sign_table2_fill: index = up | left << 2 | down << 4 | right << 6, a neighbour's field
        // (significant, sign) once it counts at that plane; entry = 8 * (c >> 1) | (c & 1) << 6)
        const uint32_t sq = 2u + (low ? 31u - (uint32_t)__builtin_clz(low) : 0u);   // (a lane that never becomes significant: unused)
        auto field = [&](uint32_t w) -> uint32_t { return ((w >> sq) | (w & 2u)) != 0u ? (((w & 1u) << 1) | 1u) : 0u; };
        const uint32_t tv = b.sgt[field(up) | (field(lf) << 2) | (field(dn) << 4) | (field(rt) << 6)];
        saddr = b.sign0 + ((tv >> 3) & 3u);
        ssymm = __builtin_amdgcn_ballot_w64(((neg ^ (tv >> 6)) & 1u) != 0u);    // :1308
    }
    const uint32_t sgaddr = b.sig0 + ctx;
    // decoder: a neighbour's field of the sign table's index once it counts -- significant | sign << 1 (sign_table2_fill)
    uint32_t fup = 0u, flf = 0u, fdn = 0u, frt = 0u;
    if constexpr (DEC) { fup = ((up & 1u) << 1) | 1u; flf = ((lf & 1u) << 1) | 1u; fdn = ((dn & 1u) << 1) | 1u; frt = ((rt & 1u) << 1) | 1u; }
    for (int q = Bmax; q >= 0; q--) {
        // (decoder: the codeword window once per plane of a coefficient -- its three sites reserve at most 64 slots of a
        // codeblock -- from the LDS counters, one iteration late, as the plane loops do once per row: dec_ring_row)
        if constexpr (DEC) dec_ring_row(c, upper_mask);
        // (the lanes whose codeblock scans plane q: a function of the halves' Bh alone -- scalar work, not a compare)
        const uint64_t onq = (q <= b.bh_lo ? 0xFFFFFFFFull : 0ull) | (q <= b.bh_hi ? 0xFFFFFFFF00000000ull : 0ull);
        uint64_t bitm = 0ull, dm = 0ull, nsm = 0ull;
        if constexpr (!DEC) bitm = shl_carry(lowx);
        // refinement call site: coefficients that are significant by now
        const uint64_t mA = onq & sigm;
        if (mA != 0ull) {
            const uint32_t p = bulk_lut(b, b.ref0 + (uint32_t)q * b.cRef);
            if constexpr (DEC) { bool one; dm = dec_site_m<false>(c, __builtin_amdgcn_inverse_ballot_w64(mA), mA, p, prec, upper_mask, st, one); }
            else enc_site2(c, mA, mA & bitm, p, prec, upper_mask);
        }
        // significance call site: the others
        const uint64_t mB = onq & ~sigm;
        if (mB != 0ull) {
            const uint32_t p = bulk_lut(b, sgaddr + (uint32_t)q * b.cSig);
            if constexpr (DEC) { bool one; nsm = dec_site_m<false>(c, __builtin_amdgcn_inverse_ballot_w64(mB), mB, p, prec, upper_mask, st, one); dm |= nsm; }
            else { enc_site2(c, mB, mB & bitm, p, prec, upper_mask); nsm = mB & bitm; }
        }
        // sign call site: coefficients that just became significant
        if (nsm != 0ull) {
            if constexpr (DEC) {
                // computeSignContextBulk :311-323 by the decoder's table (sign_table2_fill): a neighbour counts at plane q
                // when it was significant before the scan or holds a bit above q -- index = up | left << 2 | down << 4 |
                // right << 6, entry = 8 * (c >> 1) | (c & 1) << 6
                const uint32_t sq = 2u + (uint32_t)q;
                uint32_t idx = (((up >> sq) | (up & 2u)) != 0u ? fup : 0u) | ((((lf >> sq) | (lf & 2u)) != 0u ? flf : 0u) << 2);
                idx |= ((((dn >> sq) | (dn & 2u)) != 0u ? fdn : 0u) << 4) | ((((rt >> sq) | (rt & 2u)) != 0u ? frt : 0u) << 6);
                const uint32_t tv = b.sgt[idx];
                const uint32_t p = bulk_lut(b, b.sign0 + (uint32_t)q * b.cSign + ((tv >> 3) & 3u));
                bool one;
                const uint64_t s2m = dec_site_m<false>(c, __builtin_amdgcn_inverse_ballot_w64(nsm), nsm, p, prec, upper_mask, st, one);
                if (__builtin_amdgcn_inverse_ballot_w64(nsm)) neg = (__builtin_amdgcn_inverse_ballot_w64(s2m) ? 1u : 0u) ^ ((tv >> 6) & 1u);   // :1488-1490
            } else {
                const uint32_t p = bulk_lut(b, saddr + (uint32_t)q * b.cSign);
                enc_site2(c, nsm, nsm & ssymm, p, prec, upper_mask);
            }
            sigm |= nsm;
        }
        if constexpr (DEC) shl_in(acc, dm);                 // (plane Bmax first: after the loop plane q sits at bit q)
    }
    if constexpr (DEC) low = acc;
    return neg | (u & 2u) | (low << 2);
}

// One row of the bulk scan for the lane's two columns.  uL/uR: unprocessed words of the row,
// dL/dR: of the row below, pUL/pUR: processed words of the row above (updated to this row's).
template <bool DEC, class CT>
__device__ __forceinline__ void bulk_row(CT &c, uint32_t t, uint32_t uL, uint32_t uR, uint32_t lowL, uint32_t lowR,
                                         uint32_t dL, uint32_t dR, uint32_t &pUL, uint32_t &pUR, const BulkLane &b,
                                         int Bmax, uint32_t prec, uint32_t upper_mask,
                                         int32_t *st)
{
    const uint32_t sh = 2u + (uint32_t)(b.Bh < 0 ? 0 : b.Bh);
    // computeContextBulk's term of every word the two contexts count, formed ONCE where the word lives: a neighbour lane's
    // terms travel as bits (lanes of one codeblock share Bh) -- lane-1's right column of the rows above / this / below
    // in one DPP move, lane+1's left column above / below in another -- where round 3 moved five words and formed
    // sixteen terms a row
    const uint32_t ca = bulk_cc(pUL, sh), cb = bulk_cc(pUR, sh), cr = bulk_cc(uR, sh), ce = bulk_cc(dL, sh), cf = bulk_cc(dR, sh);
    const uint32_t Pb = from_prev32(cb | (cr << 1) | (cf << 2), t), Nb = from_next32(ca | (ce << 1), t);
    const uint32_t P_r = from_prev32(uR, t);                 // (words only where a sign context needs them)
    const uint32_t own = ca + cb + ce + cf;
    // left coefficients of all lanes (encodeLeftCoefficients :1320-1381)
    const uint32_t ctxL = (uint32_t)__builtin_popcount(Pb) + own + cr;
    const uint32_t nL = bulk_coeff<DEC, CT>(c, uL, lowL, ctxL, pUL, P_r, uR, dL, b, Bmax, prec, upper_mask, st);
    // right coefficients (encodeRightCoefficients :1387-1448): the left ones of this row are done
    const uint32_t N_l = from_next32(nL, t);
    const uint32_t ctxR = (uint32_t)__builtin_popcount(Nb) + own + bulk_cc(nL, sh) + bulk_cc(N_l, sh);
    const uint32_t nR = bulk_coeff<DEC, CT>(c, uR, lowR, ctxR, pUR, nL, N_l, dR, b, Bmax, prec, upper_mask, st);
    pUL = nL; pUR = nR;
}

// Per-half set-up shared by both kernels: consecutive planes, table choice, LDS copy of the table.
// Returns cbp; fills b (Bh, LDS indices) and loff (int offset of table s inside a.lut).
// consecutiveBitplanes of the lane's codeblock (level / subband of the codeblock's lane 0: see the oracle's note -- per-lane
// values would diverge); 0 for a codeblock that codes nothing
__device__ __forceinline__ int bulk_cbp(const BpcArgs &a, bool coded, int msb, int cbx, int cby)
{
    int lv0, sb0;
    find_subband(cbx * 64, cby * 64, a.AW, a.AH, a.wl, lv0, sb0);
    return coded ? consecutive_bitplanes(msb, a.k, lv0, sb0, a.wl) : 0;
}
// COMPACT (round 4): the LDS copy holds only the table's GROUPS the codeblock's lanes use -- groups g0 .. g1 of each of
// the three sections, section after section, each with one plane's worth of slack (lut_get) -- instead of the whole
// table: 210 bytes a group at the shipped geometry against 3360 for a whole wl = 5 table, so that a wave's two copies
// stop being what bounds the -k > 0 kernels' occupancy (two whole tables + the decoder's ring: 11.7 KB a wave, 3 waves a
// SIMD).  The host takes the COMPACT instantiations when every codeblock of the frame spans at most what
// kBulkCompactBytes holds (bulk_max_span_bytes: a function of the geometry alone), the whole-table ones otherwise.
// `gl`: the LUT geometry with the compact section sizes in nRef / nSig, `grpc`: the lane's group inside the copy -- what
// plane_lut takes for the two-pass planes.
constexpr int kBulkCompactBytes = 2048;
__device__ __forceinline__ int half_min32(int v)
{
    { int o = __shfl_xor(v, 16); v = v < o ? v : o; }
    { int o = __shfl_xor(v, 8); v = v < o ? v : o; }
    { int o = __shfl_xor(v, 4); v = v < o ? v : o; }
    { int o = __shfl_xor(v, 2); v = v < o ? v : o; }
    { int o = __shfl_xor(v, 1); v = v < o ? v : o; }
    return v;
}
template <bool COMPACT>
__device__ __forceinline__ int bulk_setup(const BpcArgs &a, bool coded, int msb, int cbx, int cby, int grp,
                                          uint32_t t, uint8_t *lds_half, BulkLane &b, int &loff, LutGeo &gl, int &grpc)
{
    const int total = a.g.nRef + a.g.nSig + a.g.nSign;
    const int cbp = bulk_cbp(a, coded, msb, cbx, cby);
    int s = 0;
    if (coded) {
        s = cbp < msb ? cbp : msb;
        if (s > a.n_tables - 1) s = a.n_tables - 1;
    }
    loff = s * total;
    const int glast = total * a.n_tables - 1;
    gl = a.g;
    grpc = grp;
    if constexpr (COMPACT) {
        const int g0 = half_min32(grp), g1 = -half_min32(-grp), span = g1 - g0 + 1;
        const int sl = a.g.nBp < 15 ? 16 - a.g.nBp : 1;      // planes of slack behind a section's groups (bit-plane 15)
        const int RB = (span * a.g.nBp + sl) * a.g.cRef, SB = (span * a.g.nBp + sl) * a.g.cSig, GB = (span * a.g.nBp + sl) * a.g.cSign;
        const int br = loff + g0 * a.g.nBp * a.g.cRef, bs = loff + a.g.nRef + g0 * a.g.nBp * a.g.cSig - RB,
                  bg = loff + a.g.nRef + a.g.nSig + g0 * a.g.nBp * a.g.cSign - RB - SB;
        const int ncopy = RB + SB + GB;                      // (<= kBulkCompactBytes: the host chose this instantiation)
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
        for (int base = (int)t; base < ncopy; base += 8 * 32) {
            int32_t v[8];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int jj = base + q * 32, j = jj < ncopy ? jj : ncopy - 1;
                const int src = j + (j < RB ? br : (j < RB + SB ? bs : bg));
                v[q] = a.lut[src < glast ? src : glast];
            }
#pragma unroll
            for (int q = 0; q < 8; q++) { const int j = base + q * 32; if (j < ncopy) lds_half[j] = (uint8_t)((uint32_t)v[q] & 0xFFu); }
        }
        gl.nRef = RB; gl.nSig = SB; gl.nSign = GB;
        grpc = grp - g0;
    } else {
        // (eight loads in flight a lane, as lut_to_lds; the kLutSlack entries behind the table as lut_at delivers them: the
        // next table's, or the array's last entry past its end)
        const int ncopy = total + kLutSlack;
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
        for (int base = (int)t; base < ncopy; base += 8 * 32) {
            int32_t v[8];
#pragma unroll
            for (int q = 0; q < 8; q++) { const int j = loff + base + q * 32; v[q] = a.lut[j < glast ? j : glast]; }
#pragma unroll
            for (int q = 0; q < 8; q++) { const int j = base + q * 32; if (j < ncopy) lds_half[j] = (uint8_t)((uint32_t)v[q] & 0xFFu); }
        }
    }
    __syncthreads();
    b.Bh = coded ? (msb < cbp - 1 ? msb : cbp - 1) : -1;
    b.cRef = (uint32_t)a.g.cRef; b.cSig = (uint32_t)a.g.cSig; b.cSign = (uint32_t)a.g.cSign;
    b.ref0 = (uint32_t)(grpc * a.g.nBp * a.g.cRef);
    b.sig0 = (uint32_t)(grpc * a.g.nBp * a.g.cSig + gl.nRef);
    b.sign0 = (uint32_t)(grpc * a.g.nBp * a.g.cSign + gl.nRef + gl.nSig);
    b.v.lds = lds_half; b.v.glob = a.lut; b.v.total = total; b.v.glob_total = total * a.n_tables; b.v.loff = loff;
    b.sgt = nullptr;
    b.bh_lo = (int)__builtin_amdgcn_readfirstlane((uint32_t)b.Bh);
    b.bh_hi = (int)__builtin_amdgcn_readfirstlane((uint32_t)__shfl_xor(b.Bh, 32));
    return cbp;
}

// Waves per SIMD the register allocator must leave room for (512 VGPRs per SIMD lane: 8 waves = 64,
// 7 = 72, 6 = 80, 5 = 96).  Round 1 (8 planes resident): 5.  With one plane resident the kernel fits 64.
// The k = 0 encoder: 7 (72 registers) since the end of round 3.  Its plane loops need fewer; what stood in the way was the
// prologue's spilling.  With three calls in flight, 8K: 4 / 5 / 6 / 7 / 8 waves gave 167.7 / 188.0 / 191.8 / 187.0 / 183.9
// Gpixel/s while the prologue's halves were two unrolled copies (6 it was), and 190.7 (5) / 195.0 (6) / 203.0 (7) / 194.7 (8)
// once they were a loop (enc_transpose_pass).  The -k > 0 instantiation asks for 6 (it gets 4: its planes live in registers;
// the request alone takes a lone frame from 0.536 to 0.465 ms at k = 0.5).
#ifndef PICSONG_BPC_ENC_WAVES
#define PICSONG_BPC_ENC_WAVES 7
#endif
// the -k > 0 instantiations (one wave a workgroup): resident waves per SIMD asked for -- what they get is also bounded by
// their LDS (two table copies a wave: whole tables 4, compact copies 7 and more).  8K, three hinted calls in flight,
// encode / decode Gpixel/s at k = 0.5 | 1.5: 5 waves 144 / 109, 6 waves 146 / 113 | 147 / 110, 7 waves (72 registers) 154 / 117
// | 154 / 112, 8 waves 151 / 102 | 154 / 98; a lone frame the same at 6, 7, 8 (0.37 / 0.475 ms), 0.41 ms at 5.
#ifndef PICSONG_BPC_BULK_WAVES
#define PICSONG_BPC_BULK_WAVES 7
#endif

// one row (two coefficients of the lane) of the coefficient array as magnitudes and sign bits
// (byte offset `off` from the array's base: AW * AH * 4 < 2^32, so a row is one uniform base + a 32-bit
// lane offset, and stepping a row down is one add -- no 64-bit address per row to keep or spill)
__device__ __forceinline__ void load_row(const BpcArgs &a, uint32_t off, uint32_t &m0, uint32_t &m1, uint32_t &n0, uint32_t &n1)
{
    int32_t v0, v1;
    const char *const p = reinterpret_cast<const char *>(a.coeffs_in) + (size_t)off;
    if (a.c16) {                                            // (int16 array: `off` counts its bytes)
        const uint32_t w = *reinterpret_cast<const uint32_t *>(p);
        v0 = (int32_t)(int16_t)(w & 0xFFFFu); v1 = (int32_t)w >> 16;
    } else if (a.is_float) {
        float2 f = *reinterpret_cast<const float2 *>(p);
        v0 = (int32_t)f.x; v1 = (int32_t)f.y;               // BPCEngine.cu:49: truncation toward zero
    } else {
        int2 q = *reinterpret_cast<const int2 *>(p);
        v0 = q.x; v1 = q.y;
    }
    m0 = (uint32_t)(v0 < 0 ? -v0 : v0); m1 = (uint32_t)(v1 < 0 ? -v1 : v1);
    n0 = (uint32_t)v0 >> 31; n1 = (uint32_t)v1 >> 31;
}

// a value the optimiser cannot see through (no hoisting of what is computed from it)
__device__ __forceinline__ void opaque32(uint32_t &x)
{
#if defined(__AMDGCN__)
    asm volatile("" : "+v"(x));
#else
    (void)x;
#endif
}
// nothing is scheduled across this point
__device__ __forceinline__ void sched_fence()
{
#if defined(__AMDGCN__)
    __builtin_amdgcn_sched_barrier(0);
#endif
}

// the same row as signed integers (BPCEngine.cu:49: a float coefficient is truncated toward zero)
// MODE: 0 = int32, 1 = float, 2 = the frame paths' int16 array
template <int MODE>
__device__ __forceinline__ void load_row_raw(const BpcArgs &a, uint32_t off, int32_t &v0, int32_t &v1)
{
    const char *const p = reinterpret_cast<const char *>(a.coeffs_in) + (size_t)off;
    if constexpr (MODE == 2) {
        const uint32_t w = *reinterpret_cast<const uint32_t *>(p);
        v0 = (int32_t)(int16_t)(w & 0xFFFFu); v1 = (int32_t)w >> 16;
    } else if constexpr (MODE == 1) {
        float2 f = *reinterpret_cast<const float2 *>(p);
        v0 = (int32_t)f.x; v1 = (int32_t)f.y;
    } else {
        int2 q = *reinterpret_cast<const int2 *>(p);
        v0 = q.x; v1 = q.y;
    }
}

// Four 8 x 8 bit matrices at once, one per byte lane of the eight words (row j of a lane's matrix = that byte of
// x[j]), transposed in place by the three block-swap stages (1-, 2- and 4-bit blocks): afterwards bit j of byte b
// of x[k] is what bit k of byte b of x[j] was.
__device__ __forceinline__ void bit_swap_pair(uint32_t &lo, uint32_t &hi, uint32_t mask, uint32_t sh)
{
    const uint32_t a = lo, b = hi;
    lo = bfi32(mask, a, b << sh);                           // (b & mask) << sh lands on the bits mask leaves out
    hi = bfi32(mask, a >> sh, b);
}
__device__ __forceinline__ void bit_transpose_8x8x4(uint32_t (&x)[8])
{
#pragma unroll
    for (int j = 0; j < 8; j += 2) bit_swap_pair(x[j], x[j + 1], 0x55555555u, 1u);
#pragma unroll
    for (int j = 0; j < 8; j++) if ((j & 2) == 0) bit_swap_pair(x[j], x[j + 2], 0x33333333u, 2u);
#pragma unroll
    for (int j = 0; j < 4; j++) bit_swap_pair(x[j], x[j + 4], 0x0F0F0F0Fu, 4u);
}

// One pass over a lane's two columns: planes 8 pass .. 8 pass + 7 of all 64 rows as row masks, parked in the wave's
// scratch as they come out ([plane][L rows 0-31, L rows 32-63, R rows 0-31, R rows 32-63][lane]: all eight planes of
// the pass, whatever the codeblock's MSB turns out to be -- holding them back until it is known costs 32 registers
// the kernel does not have); pass 0 also gathers the OR of the magnitudes and the sign masks.
// (The OR of the magnitudes and the sign masks are gathered in EVERY pass -- the second pass, which only waves with a
// codeblock of MSB >= 8 run, finds the same values again -- because `if (pass == 0)` inside the unrolled rows was a branch
// after every load: the compiler does not unswitch the passes' loop, every load was waited for before the next one went
// out, and the prologue of a lone frame was 64 round trips to the memory side one after the other.  As a template
// parameter instead the six inlined variants cost the whole kernel its register allocation: 620 bytes of scratch.)
template <int MODE>
__device__ __forceinline__ void enc_transpose_pass(const BpcArgs &a, int pass, uint32_t cbyte, uint32_t rstride,
                                                   uint32_t *pscr, uint32_t &ormag, U64 &sgL, U64 &sgR)
{
    // byte `pass` of a magnitude: the lower / upper pair of a word's four rows
    const uint32_t sel_lo = pass == 0 ? 0x0c0c0400u : 0x0c0c0501u, sel_hi = pass == 0 ? 0x04000c0cu : 0x05010c0cu;
    // (the two 32-row halves are a LOOP, not two copies: unrolled, the scheduler let the second half's loads run into the
    // first half's tail and the kernel carried 76 spilled dwords through its prologue -- 19 KB of private scratch a wave,
    // all of it L2 traffic; as a loop 26 dwords, the last group's loads go out together like the others, the coder's
    // L2-miss traffic drops from 210 to 196 MB a frame and three calls in flight gain 1-2 %)
#pragma unroll 1
    for (int hw = 0; hw < 2; hw++) {
        uint32_t B0[8], B1[8], sa0[4] = { 0u, 0u, 0u, 0u }, sa1[4] = { 0u, 0u, 0u, 0u };
        uint32_t roff = cbyte + (uint32_t)(32 * hw + 7) * rstride;
        opaque32(roff);       // (the rows' addresses are not to be computed ahead of the passes' loop: 64 pointers spill)
#pragma unroll
        for (int j = 7; j >= 0; j--) {                       // (descending: a row's sign is pushed in below the later rows')
            uint32_t m0[4], m1[4];
#pragma unroll
            for (int b = 0; b < 4; b++) {
                int32_t v0, v1;
                load_row_raw<MODE>(a, roff + (uint32_t)(8 * b) * rstride, v0, v1);
                m0[b] = (uint32_t)(v0 < 0 ? -v0 : v0); m1[b] = (uint32_t)(v1 < 0 ? -v1 : v1);
                ormag |= m0[b] | m1[b];
                // acc = (acc << 1) | sign: one funnel shift; rows 8 b + 7 .. 8 b in turn leave row 8 b + j at bit j
                sa0[b] = __builtin_amdgcn_alignbit(sa0[b], (uint32_t)v0, 31u);
                sa1[b] = __builtin_amdgcn_alignbit(sa1[b], (uint32_t)v1, 31u);
            }
            roff -= rstride;
            B0[j] = __builtin_amdgcn_perm(m0[1], m0[0], sel_lo) | __builtin_amdgcn_perm(m0[3], m0[2], sel_hi);
            B1[j] = __builtin_amdgcn_perm(m1[1], m1[0], sel_lo) | __builtin_amdgcn_perm(m1[3], m1[2], sel_hi);
            // eight rows (16 registers) in flight, not all 32: the loop is unrolled for its constant word indices,
            // and the scheduler would otherwise hoist every load to the top and spill
            if ((j & 1) == 0) sched_fence();
        }
        {
            const uint32_t s0 = sa0[0] | (sa0[1] << 8) | (sa0[2] << 16) | (sa0[3] << 24);
            const uint32_t s1 = sa1[0] | (sa1[1] << 8) | (sa1[2] << 16) | (sa1[3] << 24);
            if (hw == 0) { sgL.lo = s0; sgR.lo = s1; } else { sgL.hi = s0; sgR.hi = s1; }
        }
        bit_transpose_8x8x4(B0);
        bit_transpose_8x8x4(B1);
        uint32_t *q = pscr + (size_t)(pass * kEncPassPlanes) * kEncPlaneDwords + hw * 64;
#pragma unroll
        for (int k = 0; k < kEncPassPlanes; k++) { q[k * kEncPlaneDwords] = B0[k]; q[k * kEncPlaneDwords + 128] = B1[k]; }
    }
}

// (Round 3 tried the scan of a frame's codeblock lengths by the LAST wave of the launch to finish -- sizes stored with
// device-scope stores, an arrival counter, the last wave reading all nCB lengths back with device-scope loads, 2 x 128
// per lane -- in place of the one-workgroup scan launch that follows the coder: byte-identical, neutral with frames in
// flight (172.2 against 172.4 Gpixel/s), and 36 us SLOWER for a lone 8K frame: the coder's launch grew by 44 us, the
// tail of ONE wave's loads from the memory side, where scan + pack shrank by 9.  Commit 'Experiment: the sizes' scan by
// the coder's last wave' holds the code)

// Wave priority by plane count (round 4).  A lone frame's coder launch lasts as long as its DEEPEST waves: all of a frame's
// waves are resident at once (four to a SIMD at 8K), the typical one codes five planes, the few that hold the coarse
// levels' codeblocks ten or eleven -- and those spend the first two thirds of their life taking turns with three
// others for the SIMD's issue slots.  A wave that knows it has more planes than most asks for priority: the SIMD's
// arbiter serves it first, the shallow waves fill the gaps (same work, same throughput), and the launch's critical
// wave runs nearly as if alone.  PICSONG_BPC_PRIO=0 builds without.
#ifndef PICSONG_BPC_PRIO
#define PICSONG_BPC_PRIO 1
#endif
__device__ __forceinline__ void prio_by_planes(int np)
{
#if PICSONG_BPC_PRIO && defined(__AMDGCN__)
    if (np >= 9) __builtin_amdgcn_s_setprio(3);
    else if (np >= 7) __builtin_amdgcn_s_setprio(2);
    else if (np >= 6) __builtin_amdgcn_s_setprio(1);
#else
    (void)np;
#endif
}

// BULK = the -k > 0 instantiation (bulk scan after the ordinary planes, table s of the bit-plane
// LUT files, LDS copy of that table); the k = 0 instantiation compiles to the plain coder.
template <bool BULK, bool COMPACT = false>
__global__ __launch_bounds__(BULK ? 64 : 64 * kBpcEncWgWaves, BULK ? PICSONG_BPC_BULK_WAVES : PICSONG_BPC_ENC_WAVES) void bpc_encode_kernel(BpcArgs a)
{
    static_assert(BULK || !COMPACT, "compact table copies belong to the -k > 0 instantiations");
    constexpr int kTab = COMPACT ? kBulkCompactBytes : kLutLdsMax;       // bytes of one LDS table copy
    __shared__ uint8_t lds_lut[(BULK ? 2 : 1) * kTab];
    __shared__ uint32_t lds_cnt[2 * (BULK ? 1 : kBpcEncWgWaves)];      // codeword counters of the workgroup's codeblocks
    __shared__ uint8_t sign_tab[BULK ? 256 : 4];             // -k > 0: the bulk scan's sign contexts (sign_table2_fill)
    const uint32_t lane = threadIdx.x & 63u, half = lane >> 5, t = lane & 31u;
    if constexpr (BULK) sign_table2_fill(sign_tab, lane);   // (the table copy ends with the barrier)
    if (t == 0u) lds_cnt[(threadIdx.x >> 6) * 2u + half] = half * kStageCb + kStageBytes;   // bytes of the staging (enc_reserve); (the table copy below ends with a barrier)
    const int gwave = BULK ? (int)blockIdx.x
                           : (int)blockIdx.x * kBpcEncWgWaves + (int)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int wave = gwave;                                       // wave within its frame
    if (a.frames > 1) {
        const int f = gwave / a.waves_per_frame;            // wave-uniform
        wave = gwave - f * a.waves_per_frame;
        if (f >= a.frames) { wave = a.waves_per_frame; }    // padding wave of the last workgroup: codes nothing
        else {
            a.coeffs_in = (const char *)a.coeffs_in + (unsigned long long)f * a.coef_z;
            a.staging16 += (size_t)f * (size_t)a.AW * (size_t)a.AH;
            a.sizes += (size_t)f * (size_t)(a.nCB - a.cb_base);
            // (scalar selects: indexing the argument struct with f would move all of it to scratch memory)
            if (a.lut_c[0]) a.lut = f == 0 ? a.lut_c[0] : (f == 1 ? a.lut_c[1] : a.lut_c[2]);
        }
    }
    const int cb = a.cb_base + 2 * wave + (int)half;
    const bool valid = cb < a.nCB;
    const int cbx = valid ? cb % a.ncx : 0, cby = valid ? cb / a.ncx : 0;
    const size_t cbase = (size_t)(cby * 64) * (size_t)a.AW + (size_t)(cbx * 64) + 2u * t;
    const uint32_t esz = a.c16 ? 2u : 4u;                    // bytes of a coefficient
    const uint32_t cbyte = (uint32_t)cbase * esz, rstride = (uint32_t)a.AW * esz;  // byte offset of row 0 / of a row step
    uint16_t *const stw = a.staging16 + (size_t)(a.cb_base + 2 * wave) * 4096u;  // wave-uniform: the pair's first codeblock
    uint16_t *const st = stw + (size_t)half * 4096u;                            // (an invalid upper half never touches it)
    const uint32_t prec = (uint32_t)a.g.prec;
    const uint32_t upper_mask = opaque_mask(half ? 0xFFFFFFFFu : 0u);
    uint32_t *const pscr = a.plane_scratch + (size_t)gwave * (size_t)kEncScratchDwordsPerWave + lane;

    // ---- ONE pass over the coefficients: findMSB (BPCEngine.cu:176-192) and the transposition, plane k
    // of row i -> bit i of the plane's row mask; planes 0..7 of every block are built (a block's planes
    // above its MSB are zero) and parked in the scratch.  A second pass builds planes 8..15 only for a wave
    // that has a codeblock with MSB >= 8.
    U64 sgL = { 0u, 0u }, sgR = { 0u, 0u };
    uint32_t ormag = 0u;
    int msb = 32, msbmax = -1;
    bool coded = false;
    PS_BPC_TRACE(0, __builtin_amdgcn_s_memrealtime());
#pragma unroll 1
    for (int pass = 0; pass < kMaxPlanes / kEncPassPlanes; pass++) {
        // The transposition, per column and half (32 rows x the pass's 8 planes), as a bit-matrix transpose
        // instead of a bit at a time (2 instructions per bit, 16 per coefficient and pass): the rows' plane bytes
        // are packed four to a word, row 8 b + j into byte b of word j -- two byte permutes and an OR per word --
        // and the eight words go through the three block-swap stages of an 8 x 8 bit transpose, all four byte
        // lanes at once (two shifts and two bit-field inserts per pair of words); word k then holds bit k of row
        // 8 b + j at bit 8 b + j: the plane's row mask.  2.3 instructions per coefficient and pass.
        if (valid) {
            // (the coefficient type is wave-uniform: one branch around the whole pass, not one per row)
            if (a.c16) enc_transpose_pass<2>(a, pass, cbyte, rstride, pscr, ormag, sgL, sgR);
            else if (a.is_float) enc_transpose_pass<1>(a, pass, cbyte, rstride, pscr, ormag, sgL, sgR);
            else enc_transpose_pass<0>(a, pass, cbyte, rstride, pscr, ormag, sgL, sgR);
        }
        if (pass == 0) {
            ormag = half_or_dpp(ormag, upper_mask);
            msb = ormag ? 31 - __builtin_clz(ormag) : 32;
            if (valid && msb != 32 && msb > kMaxPlanes - 1) { atomicOr(a.range_flag, 1); msb = kMaxPlanes - 1; }
            coded = valid && msb != 32;
            int mm = coded ? msb : -1;
            { int o = __shfl_xor(mm, 32); mm = mm > o ? mm : o; }
            msbmax = (int)__builtin_amdgcn_readfirstlane((uint32_t)mm);
        }
        if (msbmax < (pass + 1) * kEncPassPlanes) break;
    }

    int level, sb;
    find_subband(cbx * 64 + 2 * (int)t, cby * 64, a.AW, a.AH, a.wl, level, sb);
    const int grp = level * a.g.nSub + sb;
    int cbp = 0, loff = 0;                                   // planes >= cbp take the two passes
    BulkLane bl;
    LutGeo gl = a.g;                                         // (COMPACT: the copy's own section sizes and the lane's group in it)
    int grpc = grp;
    if constexpr (BULK) { cbp = bulk_setup<COMPACT>(a, coded, msb, cbx, cby, grp, t, lds_lut + half * kTab, bl, loff, gl, grpc); bl.sgt = sign_tab; }
    else lut_to_lds(a.lut, a.g.nRef + a.g.nSig + a.g.nSign, lds_lut);
    const LutView lv = { lds_lut + (BULK ? half * kTab : 0u), a.lut, a.g.nRef + a.g.nSig + a.g.nSign,
                         (a.g.nRef + a.g.nSig + a.g.nSign) * (BULK ? a.n_tables : 1), loff };

    int np = coded ? (msb + 1 - cbp > 0 ? msb + 1 - cbp : 0) : 0;
    { int o = __shfl_xor(np, 32); np = np > o ? np : o; }
    np = (int)__builtin_amdgcn_readfirstlane((uint32_t)np);      // wave-uniform: keep it scalar
    prio_by_planes(np);

    // the plane a lane codes at step p is plane msb - p of ITS codeblock
    auto load_plane = [&](int p, U64 &bL, U64 &bR) {
        const int bp = msb - p;
        bL = U64{ 0u, 0u }; bR = U64{ 0u, 0u };
        if (coded && bp >= 0) {
            const uint32_t *q = pscr + (size_t)bp * kEncPlaneDwords;
            bL.lo = q[0]; bL.hi = q[64]; bR.lo = q[128]; bR.hi = q[192];
        }
    };

    EncCoder c;
    c.L = 0u; c.S = 0u; c.off = half * kStageCb;              // a first reservation "stores" L = 0 to word 0 (see enc_reserve)
    c.cnt_lo = 0u; c.cnt_hi = 0u; c.emptym = ~0ull;
    c.slot = half * kStageCb;                              // word 0 of the lane's codeblock (LDS form: byte offsets)
    c.ldscnt = &lds_cnt[(threadIdx.x >> 6) * 2u + half];
    c.slot0 = half * kStageCb + kStageBytes; c.lim = c.slot0 + kStageBytes * 4094u; c.pone = 1u << prec;
    c.stw = reinterpret_cast<char *>(stw);
    U64 AL = { 0u, 0u }, AR = { 0u, 0u };                 // significant before the current plane
    U64 BLn, BRn;
    load_plane(0, BLn, BRn);

    bool live = coded;                                     // false once the codeblock is bound for the raw fallback
    PS_BPC_TRACE(1, __builtin_amdgcn_s_memrealtime());
    PS_BPC_TRACE(3, (unsigned long long)np);
    for (int p = 0; p < np; p++) {
        const int bp = msb - p;
        if (!BULK && p > 0) {
            // A codeblock that has used up its 4095 codeword slots ends as raw words whatever else is coded
            // (size = 4096, expansionFix): its half sits the remaining planes out, and a wave with nothing left
            // stops.  (The wl = 6 tables' zero-probability groups send the deepest level's blocks this way after a
            // plane or two of a dozen; their waves were the tail of the whole launch.)
#if PS_ENC_LDS
            wave_lds_done();
            const uint32_t used = (*c.ldscnt - c.slot0) / kStageBytes;
#else
            const uint32_t used = half ? c.cnt_hi : c.cnt_lo;
#endif
            live = live && used < 4095u;
            if (__builtin_amdgcn_ballot_w64(live) == 0ull) break;
        }
        const bool act = live && bp >= cbp;

        PlaneLut pl = { 0u, 0u, 0u, 0u, 0u, 0u };
        if (act) pl = plane_lut<!BULK>(lv, gl, grpc, bp);

        const U64 BL = BLn, BR = BRn;
        const U64 AL2 = u_or(AL, BL), AR2 = u_or(AR, BR);            // state after this plane's SPP
        {
            const U64 sgPL = u_prev(sgR, t), sgNL = u_next(sgL, t);  // neighbour sign columns
            const U64 APL = u_prev(AR, t), APL2 = u_prev(AR2, t);    // lane-1's right column
            const U64 ANL = u_next(AL, t), ANL2 = u_next(AL2, t);    // lane+1's left column

            // ---- significance propagation pass, 32 rows at a time; only rows where some lane of the
            // wave still has an insignificant coefficient
#pragma unroll
            for (int hw = 0; hw < 2; hw++) {
                const ColHalf cpL = make_col(up_of(APL2, hw), up_of(AL2, hw), up_of(AR2, hw), w_of(APL, hw), w_of(AR, hw),
                                             dn_of(APL, hw), dn_of(AL, hw), dn_of(AR, hw),
                                             up_of(AL2, hw), up_of(sgL, hw), dn_of(AL, hw), dn_of(sgL, hw),
                                             w_of(APL, hw), w_of(sgPL, hw), w_of(AR, hw), w_of(sgR, hw), w_of(sgL, hw));
                const ColHalf cpR = make_col(up_of(AL2, hw), up_of(AR2, hw), up_of(ANL2, hw), w_of(AL2, hw), w_of(ANL2, hw),
                                             dn_of(AL, hw), dn_of(AR, hw), dn_of(ANL, hw),
                                             up_of(AR2, hw), up_of(sgR, hw), dn_of(AR, hw), dn_of(sgR, hw),
                                             w_of(AL2, hw), w_of(sgL, hw), w_of(ANL2, hw), w_of(sgNL, hw), w_of(sgR, hw));
                const uint32_t al = act ? w_of(AL, hw) : 0xFFFFFFFFu, ar = act ? w_of(AR, hw) : 0xFFFFFFFFu;
                const uint32_t nl = w_of(BL, hw) & ~al, nr = w_of(BR, hw) & ~ar;      // become significant in this plane
                uint32_t rows = wave_or32(~(al & ar));
                while (rows) {
                    // a group of four rows: their probabilities are gathered together (sig_probs4)
                    const uint32_t g4 = (uint32_t)__builtin_ctz(rows) & ~3u;
                    uint32_t sub = (rows >> g4) & 0xFu;
                    rows &= ~(0xFu << g4);
                    const uint32_t P4L = sig_probs4(cpL, pl, g4), P4R = sig_probs4(cpR, pl, g4);
                    const uint32_t Q4L = sign_probs4(cpL, pl, g4), Q4R = sign_probs4(cpR, pl, g4);
                    // the group's rows, unrolled: the byte of P4 / Q4 a row takes is then a constant of the
                    // instruction (SDWA), and a row costs two scalar instructions of loop control, not six
                    const uint32_t gbit = 1u << g4;
#define PS_SPP_ROW(J)                                                                                    \
                    if (sub & (1u << J)) {                                                                \
                        const uint32_t rowbit = vgpr_of(gbit << J);                                       \
                        enc_spp_coeff<J>(c, rowbit, al, nl, cpL.s2, P4L, Q4L, prec, upper_mask);          \
                        enc_spp_coeff<J>(c, rowbit, ar, nr, cpR.s2, P4R, Q4R, prec, upper_mask);          \
                    }
                    PS_SPP_ROW(0) PS_SPP_ROW(1) PS_SPP_ROW(2) PS_SPP_ROW(3)
#undef PS_SPP_ROW
                }
            }
        }
        // the next plane is on its way from the scratch while the refinement pass runs
        if (p + 1 < np) load_plane(p + 1, BLn, BRn);
        // ---- magnitude refinement pass: coefficients significant before this plane
#pragma unroll
        for (int hw = 0; hw < 2; hw++) {
            const uint32_t ml = act ? w_of(AL, hw) : 0u, mr = act ? w_of(AR, hw) : 0u;
            const uint32_t bl = w_of(BL, hw) & ml, br = w_of(BR, hw) & mr;
            uint32_t rows = wave_or32(ml | mr);
            const int last = rows ? 32 - __builtin_clz(rows) : 0;           // rows [0, last) hold every row with work
            if (5 * __builtin_popcount(rows) >= 3 * last && last > 0) {
                // dense half-pass (most rows have work: the lower planes): every row in order, its lane masks
                // the carries of the bit-reversed column masks -- 4 half-rate instructions a row where the
                // row-bit tests take 4 ands, 4 compares and the row bit's move
                uint32_t xml = bitrev32(ml), xbl = bitrev32(bl), xmr = bitrev32(mr), xbr = bitrev32(br);
#pragma unroll 1
                for (int ii = 0; ii < last; ii++) {
                    const uint64_t mL = shl_carry(xml), oL = shl_carry(xbl);
                    if (mL != 0ull) enc_site2(c, mL, oL, pl.ref, prec, upper_mask);
                    const uint64_t mR = shl_carry(xmr), oR = shl_carry(xbr);
                    if (mR != 0ull) enc_site2(c, mR, oR, pl.ref, prec, upper_mask);
                }
                rows = 0u;
            }
            while (rows) {
                const uint32_t ii = (uint32_t)__builtin_ctz(rows);
                const uint32_t rowbit = vgpr_of(1u << ii);
                rows &= rows - 1u;
                const uint64_t mL = __builtin_amdgcn_ballot_w64((ml & rowbit) != 0u);
                if (mL != 0ull) enc_site2(c, mL, __builtin_amdgcn_ballot_w64((bl & rowbit) != 0u), pl.ref, prec, upper_mask);
                const uint64_t mR = __builtin_amdgcn_ballot_w64((mr & rowbit) != 0u);
                if (mR != 0ull) enc_site2(c, mR, __builtin_amdgcn_ballot_w64((br & rowbit) != 0u), pl.ref, prec, upper_mask);
            }
        }
        AL = AL2; AR = AR2;
    }

    PS_BPC_TRACE(2, __builtin_amdgcn_s_memrealtime());
    // ---- bulk scan of the planes below cbp (encodeBulkMode :1640-1648): the rows are read again
    // from the coefficient array (L2-resident), three at a time in flight
    if constexpr (BULK) {
        int Bmax = bl.Bh;
        { int o = __shfl_xor(Bmax, 32); Bmax = Bmax > o ? Bmax : o; }
        Bmax = (int)__builtin_amdgcn_readfirstlane((uint32_t)Bmax);
        if (Bmax >= 0) {
            const uint32_t lowmask = bl.Bh >= 0 ? ((2u << bl.Bh) - 1u) : 0u;
            const uint32_t magmask = coded ? ((2u << msb) - 1u) : 0u;
            const uint32_t sbsh = (uint32_t)(bl.Bh + 2);            // word >> (Bh+2) = magnitude >> (Bh+1)
            auto row_words = [&](int i, uint32_t &w0, uint32_t &w1) {
                uint32_t m0, m1, n0, n1;
                load_row(a, cbyte + (uint32_t)i * rstride, m0, m1, n0, n1);
                w0 = ((m0 & magmask) << 1) | n0;
                w1 = ((m1 & magmask) << 1) | n1;
            };
            // unprocessed word: sign | significant-before-the-scan << 1
            auto unp = [&](uint32_t w) -> uint32_t { return (w & 1u) | ((w >> sbsh) != 0u ? 2u : 0u); };
            // (the rows come two ahead: a row's scan needs the row below it at once, and a load issued at the top of the
            // row it is needed in was an L2 round trip in every row's path)
            uint32_t pUL = 0u, pUR = 0u, c0, c1, n0 = 0u, n1 = 0u, m0 = 0u, m1 = 0u;
            row_words(0, c0, c1);
            row_words(1, n0, n1);
            uint32_t uc0 = unp(c0), uc1 = unp(c1);           // (a row's unprocessed words are formed once: as the row below, then reused)
            for (int i = 0; i < 64; i++) {
                if (i < 62) row_words(i + 2, m0, m1); else { m0 = 0u; m1 = 0u; }
                const uint32_t un0 = i < 63 ? unp(n0) : 0u, un1 = i < 63 ? unp(n1) : 0u;
                bulk_row<false>(c, t, uc0, uc1, (c0 >> 1) & lowmask, (c1 >> 1) & lowmask, un0, un1, pUL, pUR, bl, Bmax, prec,
                                upper_mask, (int32_t *)nullptr);
                c0 = n0; c1 = n1; n0 = m0; n1 = m1; uc0 = un0; uc1 = un1;
            }
        }
    }

    // flush (Encode BPCEngine.cu:1719) + sizeArray (:2010) + MSB slot (:1998)
#if PS_ENC_LDS
    if (coded) {
        const uint32_t sl = c.slot < c.lim ? c.slot : c.lim;
        *reinterpret_cast<uint16_t *>(c.stw + sl) = (uint16_t)c.L;
    }
    wave_lds_done();                                       // every lane's last atomic has landed
    const uint32_t cw_count = (*c.ldscnt - c.slot0) / kStageBytes;
    const uint32_t size = (cw_count > 4095u ? 4095u : cw_count) + 1u;
#else
    if (coded) *reinterpret_cast<uint16_t *>(c.stw + c.off) = (uint16_t)c.L;
    const uint32_t size = (half ? c.cnt_hi : c.cnt_lo) + 1u;
#endif
    if (valid && t == 0u) a.sizes[cb] = (int32_t)size;
    // word 0 (the MSB) and expansionFix :1905-1912 (which overwrites the whole block) must land after every
    // codeword store of the block, the lanes' first-reservation stores to word 0 included
    wave_stores_issued();
    if (valid && size == 4096u) {
        for (int i = 0; i < 64; i++) {
            uint32_t m0, m1, n0, n1;
            load_row(a, cbyte + (uint32_t)i * rstride, m0, m1, n0, n1);
            const uint32_t w0 = ((m0 << 1) + n0) & 0xFFFFu, w1 = ((m1 << 1) + n1) & 0xFFFFu;
            *reinterpret_cast<uint32_t *>(st + t * 128u + 2u * (uint32_t)i) = w0 | (w1 << 16);
        }
    } else if (valid && t == 0u) {
        st[0] = (uint16_t)msb;
    }
}


// Self-test of the property the LDS form of the slot reservation rests on: when several lanes of ONE
// ds_add_rtn_u32 hit one address, the pre-add values come back in ascending lane order.  Every wave draws
// `iters` pseudo-random 64-bit lane masks, lets the lanes in the mask add 1 to their half's counter and
// compares what they get with the counter's value before the instruction plus their v_mbcnt rank.
__global__ __launch_bounds__(256) void lds_order_selftest_kernel(int iters, uint32_t seed, uint32_t *mismatches)
{
    __shared__ uint32_t cnt[8];
    const uint32_t lane = threadIdx.x & 63u, half = lane >> 5, w = threadIdx.x >> 6;
    if ((lane & 31u) == 0u) cnt[w * 2u + half] = 0u;
    __syncthreads();
    uint32_t z = seed ^ (blockIdx.x * 0x9E3779B9u) ^ (w * 0x85EBCA6Bu), bad = 0u, expect_lo = 0u, expect_hi = 0u;
    for (int i = 0; i < iters; i++) {
        z = 1664525u * z + 1013904223u;
        uint32_t mlo = z;
        z = 1664525u * z + 1013904223u;
        uint32_t mhi = z;
        // every density occurs: all lanes, sparse masks, one half only
        if ((i & 7) == 1) { mlo &= mlo >> 3; mhi &= mhi << 5; }
        if ((i & 7) == 2) { mlo = ~0u; mhi = ~0u; }
        if ((i & 7) == 3) mlo = 0u;
        mlo = __builtin_amdgcn_readfirstlane(mlo); mhi = __builtin_amdgcn_readfirstlane(mhi);
        const uint64_t m = ((uint64_t)mhi << 32) | mlo;
        uint32_t got = 0u;
        const bool in = __builtin_amdgcn_inverse_ballot_w64(m);
        if (in) got = __hip_atomic_fetch_add(&cnt[w * 2u + half], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        const uint32_t rank = half ? __builtin_amdgcn_mbcnt_hi(mhi, 0u) : __builtin_amdgcn_mbcnt_lo(mlo, 0u);
        if (in && got != (half ? expect_hi : expect_lo) + rank) bad++;
        expect_lo += (uint32_t)__builtin_popcount(mlo);
        expect_hi += (uint32_t)__builtin_popcount(mhi);
    }
    if (bad) atomicAdd(mismatches, bad);
}

// =============================================================================================
// Decoder (kernelBPCDecoder BPCEngine.cu:2126-2215, Decode :1777-1837, SPPDecoder :559-594,
// MRPDecoder :743-762, arithmeticDecoder :405-442, writeCoefficients :94-111,
// copyEntireCodeblock :1915-1922).  Significance is only known as symbols decode, so the
// decoder keeps LIVE row masks in W-form (row i of the current 32-row half-pass at bit ii+1, the
// rows above / below at bits ii / ii+2) and refreshes its copies of the two neighbour columns by
// DPP only after some lane became significant in the phase before.  Decoded planes rotate upward
// so that plane register k ends up holding plane k; all planes are decoded (k = 0), so the
// mid-point approximation bits of the reference (:577,:755-757) never reach the output.
// =============================================================================================

// arithmeticDecoder BPCEngine.cu:405-442, one call site
// `on` / onm = ballot(on): as enc_site_on.  Returns the ballot of the lanes that decoded a 1 and
// sets `one` in them.  The interval split and the compare run in every lane (the result of a lane
// that is off is never used) so that the ballot is the compare's own mask: a bool that leaves an
// exec-masked region costs a select and a second compare to get back into a mask.
// KEEP: look after the codeword window here (callers without a per-row dec_ring_keep)
template <bool KEEP>
__device__ __forceinline__ uint64_t dec_site_m(Coder &c, bool on, uint64_t onm, uint32_t p, uint32_t prec,
                                               uint32_t upper_mask, const int32_t *stage, bool &one)
{
    // The decoder keeps D = cw - L in place of (cw, L) (Coder::L holds D): arithmeticDecoder's test cw >= L + a with
    // a = a0 + 1 is D > a0, and L += a is D -= a.  cw >= L holds at every step of ANY stream -- L starts at 0 and
    // only moves to an L + a that the test has just found <= cw -- so D never wraps and the form is the reference's
    // for damaged streams too.  One register, one instruction per call site and one per reservation fewer.
    const bool empty = c.S == 0u;
    const uint64_t m = __builtin_amdgcn_ballot_w64(empty) & onm;
    if (m != 0ull) {
#if PS_ENC_LDS
        // the slot by one LDS atomic add per requesting lane (see the encoder's enc_reserve)
        if (__builtin_amdgcn_inverse_ballot_w64(m)) {
            // (the counter counts bytes of the 16-bit ring: pre-add value & 0x3FE = the slot's offset in the 1 KB
            // aligned ring, one v_and_or away from its LDS address)
            const uint32_t off = __hip_atomic_fetch_add(c.ldscnt, kDecCntUnit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            c.L = *(const __attribute__((address_space(3))) uint16_t *)(uintptr_t)(c.ringaddr | (off & (uint32_t)(kDecRing * 2 - 2)));
            c.S = 0xFFFFu;
        }
        if constexpr (KEEP) {                              // (the plane loops read the LDS counters once a row instead)
            c.cnt_lo = __builtin_amdgcn_readfirstlane(c.cnt_lo + (uint32_t)__builtin_popcount((uint32_t)m));
            c.cnt_hi = __builtin_amdgcn_readfirstlane(c.cnt_hi + (uint32_t)__builtin_popcount((uint32_t)(m >> 32)));
        }
#else
        reserve_enc(c, on && empty, m, upper_mask);
        if (on && empty) c.L = c.ring[c.slot & (kDecRing - 1)];          // D = cw - 0
#endif
        if constexpr (KEEP) dec_ring_keep(c, upper_mask);
        (void)stage;                                          // (the codewords come from c.srcbase: dec_ring_fill)
    }
#if defined(__AMDGCN__)
    // a0 = (S * p) >> prec;  D > a0 decodes a 1: S' = S - a0 - 1, D' = D - a0 - 1;  else S' = a0
    // (arithmeticDecoder BPCEngine.cu:405-442).  As the encoder's update: the instructions themselves over
    // exec masks (exec is all ones on entry), full-rate moves and adds where the compiler's form selects.
    // (v_cmpx on this target writes the lane mask to its scalar destination AND to exec: with exec = the coding lanes
    // the compare delivers "lanes decoding a 1" in both -- no scalar and)
    uint32_t a;
    uint64_t onem;
    asm volatile(
        "v_mul_u32_u24 %[a], %[S], %[p]\n\t"
        "v_lshrrev_b32 %[a], %[pr], %[a]\n\t"          // a0
        "s_mov_b64 exec, %[on]\n\t"
        "v_cmpx_gt_u32_e64 %[onem], %[D], %[a]\n\t"    // exec = onem = lanes decoding a 1
        "v_not_b32 %[a], %[a]\n\t"                      //   -(a0 + 1)
        "v_add_u32 %[D], %[D], %[a]\n\t"
        "v_add_u32 %[a], %[S], %[a]\n\t"                //   a = S - a0 - 1 (their new S)
        "s_mov_b64 exec, %[on]\n\t"
        "v_mov_b32 %[S], %[a]\n\t"                      // lanes decoding a 0 still hold a0
        "s_mov_b64 exec, -1"
        : [S] "+v"(c.S), [D] "+v"(c.L), [a] "=&v"(a), [onem] "=&s"(onem)
        : [on] "s"(onm), [p] "v"(p), [pr] "s"(prec));
    (void)on;
    one = __builtin_amdgcn_inverse_ballot_w64(onem);
    return onem;
#else
    const uint32_t a0 = mul_u24(c.S, p) >> prec;
    const bool ge = c.L > a0;
    const uint64_t gem = __builtin_amdgcn_ballot_w64(ge);
    if (on) {
        c.S = ge ? c.S + ~a0 : a0;
        c.L = ge ? c.L + ~a0 : c.L;
    }
    one = ge && on;
    return gem & onm;
#endif
}
template <bool KEEP = true>
__device__ __forceinline__ bool dec_site_on(Coder &c, bool on, uint64_t onm, uint32_t p, uint32_t prec,
                                            uint32_t upper_mask, const int32_t *stage)
{
    bool one;
    (void)dec_site_m<KEEP>(c, on, onm, p, prec, upper_mask, stage, one);
    return one;
}
__device__ __forceinline__ uint32_t dec_site(Coder &c, uint32_t inact, uint32_t p, uint32_t prec,
                                             uint32_t upper_mask, const int32_t *stage)
{
    const bool on = inact == 0u;
    return dec_site_on(c, on, __builtin_amdgcn_ballot_w64(on), p, prec, upper_mask, stage) ? 1u : 0u;
}

// Sign context of the decoder by table: the four neighbours' (significant, sign) pairs make an 8-bit
// index (up, down, left, right; bit 2n = significant, bit 2n+1 = sign), the LDS table holds for each
// the context c of computeSignContext (BPCEngine.cu:252-308) in bits 0-2 and the bit offset
// 8 * (c >> 1) of its probability inside PlaneLut::sign in bits 3-7.  Replaces ~30 compares / selects.
__device__ __forceinline__ void sign_table_fill(uint8_t *tab, uint32_t lane)
{
    for (uint32_t idx = lane; idx < 256u; idx += 64u) {
        const int up = (int)(idx & 1u) - (int)(idx & 2u), dn = (int)((idx >> 2) & 1u) - (int)((idx >> 2) & 2u);
        const int lf = (int)((idx >> 4) & 1u) - (int)((idx >> 4) & 2u), rt = (int)((idx >> 6) & 1u) - (int)((idx >> 6) & 2u);
        const uint32_t sc = sign_ctx(lf + rt, up + dn);
        tab[idx] = (uint8_t)(sc | ((8u * (sc >> 1)) << 3));
    }
}

// ---- decoder, significance propagation pass: interleaved column masks ("C-form") --------------------------------
// What a visit needs of a neighbour column is (significant, sign) of three rows; kept as separate row masks (round 2)
// that is six funnel shifts and a dozen shift-and-mask instructions to assemble the 8-bit index of the sign table.
// Here a column is ONE mask of 2 bits per row -- row r: significant at bit 2 (r + 1), sign at bit 2 (r + 1) + 1, rows
// -1 and 64 (always zero) included: 132 bits, five words -- so that ONE funnel shift of the pair (word k, word k + 1)
// by 2 j puts rows 16 k + j - 1, + j, + j + 1 at bits 0..5: `x & 0x15` are the three significance bits of the context
// count, `x & 0x33` / `x & 0xC` the fields of the sign index.  The pass runs in four blocks of 16 rows, each on the
// static register pair of its rows; a newly significant coefficient ORs (1 | sign << 1) << (2 j + 2) into its pair.
// The codeblock edge (lane 0 has no left neighbour, lane 31 no right one: correctCBBorders BPCEngine.cu:465-484) is
// folded into the per-lane masks the neighbour fields are extracted with, so the DPP copies need no fix-up; a half
// that codes nothing in this plane carries all-significant masks for the duration (never "on", no test per site).
//
// Sign table (LDS, 256 bytes), index = up | left << 2 | down << 4 | right << 6, each field (significant, sign):
// bits 0-4 = bit offset 8 * (c >> 1) of the context's probability inside PlaneLut::sign (v_bfe_u32 takes exactly
// these five bits of its offset operand), bit 6 = c & 1: the polarity (computeSignContext BPCEngine.cu:252-308), so
// that entry >> 5 is the 0 / 2 that flips the sign bit of the value ORed into the column mask.
__device__ __forceinline__ void sign_table2_fill(uint8_t *tab, uint32_t lane)
{
    for (uint32_t idx = lane; idx < 256u; idx += 64u) {
        const int up = (int)(idx & 1u) - (int)(idx & 2u), lf = (int)((idx >> 2) & 1u) - (int)((idx >> 2) & 2u);
        const int dn = (int)((idx >> 4) & 1u) - (int)((idx >> 4) & 2u), rt = (int)((idx >> 6) & 1u) - (int)((idx >> 6) & 2u);
        const uint32_t sc = sign_ctx(lf + rt, up + dn);
        tab[idx] = (uint8_t)((8u * (sc >> 1)) | ((sc & 1u) << 6));
    }
}

// popcount(x) + acc: v_bcnt_u32_b32's own accumulator
__device__ __forceinline__ uint32_t bcnt_acc(uint32_t x, uint32_t acc)
{
#if defined(__AMDGCN__)
    uint32_t r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
#else
    return (uint32_t)__builtin_popcount(x) + acc;
#endif
}
__device__ __forceinline__ uint32_t dpp_prev(uint32_t v) { return __builtin_amdgcn_update_dpp(0u, v, 0x138 /*wave_shr:1*/, 0xf, 0xf, true); }
__device__ __forceinline__ uint32_t dpp_next(uint32_t v) { return __builtin_amdgcn_update_dpp(0u, v, 0x130 /*wave_shl:1*/, 0xf, 0xf, true); }

// per-lane constants of the pass: the masks neighbour fields are extracted with (zero at the codeblock's edge)
struct DecEdge { uint32_t p15, pC, n15, nC; };

// arithmeticDecoder (BPCEngine.cu:405-442) for the lanes in onm with probability p; returns the ballot of the
// lanes that decoded a 1.  (The caller looks after the codeword window: dec_ring_keep once a row.)
__device__ __forceinline__ uint64_t dec_site_lean(Coder &c, uint64_t onm, uint32_t p, uint32_t prec, uint32_t upper_mask,
                                                  const int32_t *stage)
{
    bool one;
    return dec_site_m<false>(c, __builtin_amdgcn_inverse_ballot_w64(onm), onm, p, prec, upper_mask, stage, one);
}

// One block of 16 rows (rows 16 K .. 16 K + 15 of the codeblocks) of the significance propagation pass
// (SPPDecoderLauncher BPCEngine.cu:770-843 order: row by row, all left columns, then all right columns).
// (l0, l1) / (r0, r1): words K, K + 1 of the lane's left / right column masks; sxL / sxR: the X-form significance
// word of these rows' 32-row half, `xb` = bit of the block's row 0 in it; rows: the block's rows in which some
// lane of the wave has an insignificant coefficient.
__device__ __forceinline__ void dec_spp_block(Coder &c, uint32_t rows, uint32_t &l0, uint32_t &l1, uint32_t &r0, uint32_t &r1,
                                              uint32_t &sxL, uint32_t &sxR, uint32_t xb, const DecEdge &e,
                                              const PlaneLut &pl, uint32_t prec, uint32_t upper_mask,
                                              const int32_t *cw, const uint8_t *sgt)
{
    const uint32_t kSel = 0x0C0C0C00u;                     // byte 0 of the permute selector = the context; bytes 1..3 zero
    // lane-1's right column, lane+1's left column (as they are now; refreshed after a phase that changed them)
    uint32_t p0 = dpp_prev(r0), p1 = dpp_prev(r1), n0 = dpp_next(l0), n1 = dpp_next(l1);
    while (rows) {
        const uint32_t j = (uint32_t)__builtin_ctz(rows);
        rows &= rows - 1u;
        const uint32_t sh = 2u * j;
        dec_ring_row(c, upper_mask);
        // ---- all lanes: left column; neighbours = lane-1's right column | own right column
        const uint32_t xo = __builtin_amdgcn_alignbit(l1, l0, sh), xp = __builtin_amdgcn_alignbit(p1, p0, sh);
        const uint32_t xr = __builtin_amdgcn_alignbit(r1, r0, sh);
        const uint32_t cR = bcnt_acc(xr & 0x15u, kSel);     // own right column: shared by both visits of the row
        uint32_t xl = xo;                                   // the left column as the right column's visit sees it
        {
            const uint64_t onm = __builtin_amdgcn_ballot_w64((xo & 4u) == 0u);
            const uint32_t sel = bcnt_acc(xo & 0x15u, bcnt_acc(xp & e.p15, cR));      // computeContext :222-230
            const uint32_t p07 = __builtin_amdgcn_perm(pl.sig1, pl.sig0, sel);
            const uint64_t onem = dec_site_lean(c, onm, sel > (kSel | 7u) ? pl.sig8 : p07, prec, upper_mask, cw);
            if (onem != 0ull) {
                // sign: index = up | left << 2 | down << 4 | right << 6 (each: significant, sign)
                uint32_t idx = (xp & e.pC) | (xo & 0x33u);
                idx |= (xr & 0xCu) << 4;
                uint32_t tv = sgt[idx];
                opaque32(tv);             // (the zero extension belongs to the load: used from another block it costs an and)
                const uint32_t p2 = __builtin_amdgcn_ubfe(pl.sign, tv, 8u);
                const uint64_t s2m = dec_site_lean(c, onem, p2, prec, upper_mask, cw);
                if (__builtin_amdgcn_inverse_ballot_w64(onem)) {
                    const uint32_t val = (__builtin_amdgcn_inverse_ballot_w64(s2m) ? 3u : 1u) ^ (tv >> 5);   // :587-589
                    const uint64_t sv = (uint64_t)val << (sh + 2u);
                    l0 |= (uint32_t)sv; l1 |= (uint32_t)(sv >> 32);
                    sxL |= 1u << (xb + j);
                }
                // lane+1's left column as it is after this row's left phase (:791, shfl_down)
                n0 = dpp_next(l0); n1 = dpp_next(l1);
                xl = __builtin_amdgcn_alignbit(l1, l0, sh);
            }
        }
        // ---- all lanes: right column; neighbours = own left column | lane+1's left column
        {
            const uint32_t xn = __builtin_amdgcn_alignbit(n1, n0, sh);
            const uint64_t onm = __builtin_amdgcn_ballot_w64((xr & 4u) == 0u);
            const uint32_t sel = bcnt_acc(xl & 0x15u, bcnt_acc(xn & e.n15, cR));
            const uint32_t p07 = __builtin_amdgcn_perm(pl.sig1, pl.sig0, sel);
            const uint64_t onem = dec_site_lean(c, onm, sel > (kSel | 7u) ? pl.sig8 : p07, prec, upper_mask, cw);
            if (onem != 0ull) {
                uint32_t idx = (xl & 0xCu) | (xr & 0x33u);
                idx |= (xn & e.nC) << 4;
                uint32_t tv = sgt[idx];
                opaque32(tv);
                const uint32_t p2 = __builtin_amdgcn_ubfe(pl.sign, tv, 8u);
                const uint64_t s2m = dec_site_lean(c, onem, p2, prec, upper_mask, cw);
                if (__builtin_amdgcn_inverse_ballot_w64(onem)) {
                    const uint32_t val = (__builtin_amdgcn_inverse_ballot_w64(s2m) ? 3u : 1u) ^ (tv >> 5);
                    const uint64_t sv = (uint64_t)val << (sh + 2u);
                    r0 |= (uint32_t)sv; r1 |= (uint32_t)(sv >> 32);
                    sxR |= 1u << (xb + j);
                }
                // lane-1's right column as it is after this row's right phase (:804, shfl_up)
                p0 = dpp_prev(r0); p1 = dpp_prev(r1);
            }
        }
    }
}

// the sign bits (odd positions) of 32 rows of a C-form column -- words a, b, c hold rows 32 h - 1 .. 32 h + 31 -- as an
// X-form row mask
__device__ __forceinline__ uint32_t cform_signs(uint32_t a, uint32_t b, uint32_t c2)
{
    auto odd16 = [](uint32_t y) -> uint32_t {              // bits 1, 3, .. 31 -> bits 0 .. 15
        uint32_t x = (y >> 1) & 0x55555555u;
        x = (x | (x >> 1)) & 0x33333333u;
        x = (x | (x >> 2)) & 0x0F0F0F0Fu;
        x = (x | (x >> 4)) & 0x00FF00FFu;
        return (x | (x >> 8)) & 0xFFFFu;
    };
    const uint32_t lo = __builtin_amdgcn_alignbit(b, a, 2u), hi = __builtin_amdgcn_alignbit(c2, b, 2u);
    return odd16(lo) | (odd16(hi) << 16);
}

// The decoded planes of one column half (row masks PL / PR, plane k in word k) back to coefficients: the same 8 x 8
// bit-matrix transpose as the encoder's prologue (bit_transpose_8x8x4 is its own inverse) turns eight plane words into
// eight words of row bytes -- the magnitude byte of row 8 b + j is byte b of word j -- instead of a bit at a time.
// C16: `out` is the lane's place in an int16 Mallat array (row stride AW 16-bit words): a row's two coefficients leave
// as ONE dword -- the decode frame paths where every magnitude stays below 2^15 (BpcArgs::c16)
template <bool C16>
__device__ __forceinline__ void store_coef_pair(int32_t *out, size_t row, int AW, int32_t v0, int32_t v1)
{
    if constexpr (C16) *reinterpret_cast<uint32_t *>(reinterpret_cast<int16_t *>(out) + row * (size_t)AW) = ((uint32_t)v0 & 0xFFFFu) | ((uint32_t)v1 << 16);
    else *reinterpret_cast<int2 *>(out + row * (size_t)AW) = make_int2(v0, v1);
}
template <int NP, int NA, bool C16 = false>
__device__ __forceinline__ void write_rows(const uint32_t (&PL)[NA], const uint32_t (&PR)[NA],
                                           uint32_t sgL, uint32_t sgR, int row0, bool valid, int32_t sz,
                                           const int32_t *stage, uint32_t t, int32_t *out, int AW,
                                           const uint16_t *raw16 = nullptr, uint32_t rawoff = 0u, uint32_t rawlast = 0u,
                                           int32_t word0 = 0)
{
    if (!valid) return;
    if (sz == 4096) {                                       // raw codeblock (expansionFix): words, not planes
#pragma unroll 1
        for (int ii = 0; ii < 32; ii++) {
            const int i = row0 + ii;
            int2 w;
            if (raw16) {
                // the packed stream: words 1 .. 4095 are the codeblock's 4095 shorts, word 0 sits in the MSB's place
                // (pack_kernel: out[9 + 2 cb] = st[0])
                // (raw16 + rawoff = the codeblock's first short; rawlast = the buffer's last short: lengths are clamped
                // to 4096 and the buffer holds a worst-case stream, so the guard only ever acts on a buffer that is not)
                const uint32_t k = t * 128u + 2u * (uint32_t)i;
                const uint32_t i0 = rawoff + k - 1u, i1 = rawoff + k;
                w.x = k == 0u ? word0 : (int32_t)raw16[i0 < rawlast ? i0 : rawlast];
                w.y = (int32_t)raw16[i1 < rawlast ? i1 : rawlast];
            } else {
                w = *reinterpret_cast<const int2 *>(stage + t * 128u + 2u * (uint32_t)i);
            }
            int32_t v0 = (int32_t)(((uint32_t)w.x & 0xFFFFFFu) >> 1); if (w.x & 1) v0 = -v0;
            int32_t v1 = (int32_t)(((uint32_t)w.y & 0xFFFFFFu) >> 1); if (w.y & 1) v1 = -v1;
            store_coef_pair<C16>(out, (size_t)i, AW, v0, v1);
        }
        return;
    }
    static_assert(NP == 8 || NP == 16, "planes in groups of eight");
    uint32_t X0[8], X1[8], Y0[8], Y1[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        X0[k] = PL[k]; X1[k] = PR[k];
        Y0[k] = NP > 8 ? PL[(NP > 8 ? 8 : 0) + k] : 0u; Y1[k] = NP > 8 ? PR[(NP > 8 ? 8 : 0) + k] : 0u;
    }
    bit_transpose_8x8x4(X0);
    bit_transpose_8x8x4(X1);
    if (NP > 8) { bit_transpose_8x8x4(Y0); bit_transpose_8x8x4(Y1); }
#pragma unroll
    for (int b = 0; b < 4; b++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int ii = 8 * b + j;
            uint32_t m0 = (X0[j] >> (8 * b)) & 0xFFu, m1 = (X1[j] >> (8 * b)) & 0xFFu;
            if (NP > 8) { m0 |= ((Y0[j] >> (8 * b)) & 0xFFu) << 8; m1 |= ((Y1[j] >> (8 * b)) & 0xFFu) << 8; }
            const int32_t v0 = ((sgL >> ii) & 1u) ? -(int32_t)m0 : (int32_t)m0;
            const int32_t v1 = ((sgR >> ii) & 1u) ? -(int32_t)m1 : (int32_t)m1;
            store_coef_pair<C16>(out, (size_t)(row0 + ii), AW, v0, v1);
        }
    }
}

// One wave64 = codeblocks cb_base + 2*wave (lanes 0-31) and +1 (lanes 32-63).
#ifndef PICSONG_BPC_DEC_WAVES
#define PICSONG_BPC_DEC_WAVES 4
#endif
// The k = 0 decoder (one plane in registers, the finished ones parked in the scratch) fits 64 VGPRs in its plane
// loop -- only the prologue and the epilogue spill a few dwords, once per wave -- so eight waves share a SIMD: the
// decoder's call sites wait on LDS (codeword ring, sign table, slot reservation) where the encoder's do not, and
// resident waves hide that.  Measured, 8K lossless, three streams: 5 / 6 / 7 / 8 waves 113 / 118 / 121 / 127 Gpixel/s.
#ifndef PICSONG_BPC_DEC_WAVES8
#define PICSONG_BPC_DEC_WAVES8 8
#endif
// -k > 0: two instantiations share a launch's waves: NP = 8 keeps 8 plane registers per column half
// and takes the waves whose two codeblocks have at most 8 coded planes -- nearly all --,
// NP = 16 (4 waves / SIMD) takes the rest; the host launches both, a wave of the other class returns
// at once.
constexpr int kDecSmallPlanes = 8;
// S16: the frame paths' instantiation, codewords read from the packed stream (BpcArgs::cw16)
// C16 (with S16): the coefficients leave as an int16 Mallat array (row stride AW) at coeffs_out, for the synthesis
// kernels' C16 instantiations -- half the bytes the decoder writes and the transform reads.  Only in contexts whose
// magnitudes are bounded below 2^15 (coef16_ok: an honest stream's codeblocks have MSB <= 14 there; a damaged table
// that claims more raises the range flag and decodes to wrapped values -- "something", as every damaged stream does)
template <bool BULK, int NP, bool S16 = false, bool C16 = false, bool COMPACT = false>
__global__ __launch_bounds__(BULK ? 64 : 64 * kBpcDecWgWaves, !BULK ? PICSONG_BPC_DEC_WAVES8 : PICSONG_BPC_BULK_WAVES)
void bpc_decode_kernel(BpcArgs a)
{
    static_assert(NP == kDecSmallPlanes, "one instantiation for every plane count (the planes are parked in the scratch)");
    static_assert(!C16 || S16, "the 16-bit coefficient form belongs to the frame paths");
    static_assert(BULK || !COMPACT, "compact table copies belong to the -k > 0 instantiations");
    constexpr int kTab = COMPACT ? kBulkCompactBytes : kLutLdsMax;       // bytes of one LDS table copy
    __shared__ uint8_t lds_lut[(BULK ? 2 : 1) * kTab];
    __shared__ uint8_t sign_tab[256];
    __shared__ __attribute__((aligned(1024))) uint16_t cw_ring[(BULK ? 1 : kBpcDecWgWaves) * 2 * kDecRing];
    __shared__ uint32_t lds_cnt[(BULK ? 1 : kBpcDecWgWaves) * 2];
    const uint32_t lane = threadIdx.x & 63u, half = lane >> 5, t = lane & 31u;
    if (t == 0u) lds_cnt[(threadIdx.x >> 6) * 2u + half] = 0u;        // (barrier: below, with the table copy)
    sign_table2_fill(sign_tab, lane);                       // (the LUT copy below ends with the barrier)
    const int gwave = BULK ? (int)blockIdx.x : (int)blockIdx.x * kBpcDecWgWaves + (int)(threadIdx.x >> 6);
    PS_BPC_TRACE(0, __builtin_amdgcn_s_memrealtime());     // (trace builds only, tools/bpc_trace.py dec)
    int wave = gwave;                                       // wave within its frame
    if (a.frames > 1) {                                     // batched launch (picsong_decode_frames), as in the encoder
        const int f = gwave / a.waves_per_frame;            // wave-uniform
        wave = gwave - f * a.waves_per_frame;
        if (f >= a.frames) { wave = a.waves_per_frame; }    // padding wave of the last workgroup: decodes nothing
        else {
            a.coeffs_out = (int32_t *)((char *)a.coeffs_out + (unsigned long long)f * a.coef_z);
            a.staging += (size_t)f * (size_t)a.AW * (size_t)a.AH;
            a.sizes += (size_t)f * (size_t)(a.nCB - a.cb_base);
            if (S16) { a.cw16 += (size_t)f * (size_t)a.cw16_stride; a.cw16_offsets += (size_t)f * (size_t)a.nCB; a.cw16_total += f; }
            // (scalar selects: indexing the argument struct with f would move all of it to scratch memory)
            if (a.lut_c[0]) a.lut = f == 0 ? a.lut_c[0] : (f == 1 ? a.lut_c[1] : a.lut_c[2]);
        }
    }
    const int cb = a.cb_base + 2 * wave + (int)half;
    const bool valid = cb < a.nCB;
    const int cbx = valid ? cb % a.ncx : 0, cby = valid ? cb / a.ncx : 0;
    // (C16: the lane's first coefficient in 16-bit words, expressed in the 32-bit words of coeffs_out's type)
    const size_t cbase = (size_t)(cby * 64) * (size_t)a.AW + (size_t)(cbx * 64) + 2u * t;
    int32_t *const obase = C16 ? reinterpret_cast<int32_t *>(reinterpret_cast<int16_t *>(a.coeffs_out) + cbase) : a.coeffs_out + cbase;
    constexpr bool s16 = S16;                               // the stream itself, not the staging
    const int cbs = valid ? cb : a.cb_base;
    const int32_t *stage = s16 ? nullptr : a.staging + (size_t)cbs * 4096u;   // (32-bit form) word 0 = the MSB, slot k = stage[1 + k]
    const int32_t *cw = s16 ? nullptr : stage + 1;
    const uint32_t prec = (uint32_t)a.g.prec;
    const uint32_t upper_mask = opaque_mask(half ? 0xFFFFFFFFu : 0u);

    // ONE plane in registers, the one being decoded; a finished plane is parked in the wave's scratch (the
    // encoder's layout: [plane][L rows 0-31, L rows 32-63, R rows 0-31, R rows 32-63][lane], every access a 256-byte
    // row) and the epilogue reads the planes back eight at a time for the transposition -- 28 registers fewer than
    // eight resident planes, none of the 28 moves a plane that rotated them upward, and one kernel for every plane
    // count (round 2: two instantiations over the same grid, NP = 8 and NP = 16, a launch of nothing for most frames).
    // -k > 0 (round 4) parks its two-pass planes the same way: the epilogue writes the coefficients they make, and the
    // bulk scan that follows ORs its low bits into them row by row (until round 4 it kept eight planes in registers
    // and assembled every row from them bit by bit: 32 registers and ~50 vector instructions a row).
    constexpr int NPR = 1;
    uint32_t PLlo[NPR], PLhi[NPR], PRlo[NPR], PRhi[NPR];
#pragma unroll
    for (int k = 0; k < NPR; k++) { PLlo[k] = PLhi[k] = PRlo[k] = PRhi[k] = 0u; }
    uint32_t *const pscr = a.plane_scratch + (size_t)gwave * (size_t)kEncScratchDwordsPerWave + lane;
    // significance + sign of the lane's two columns, interleaved (C-form, see dec_spp_block)
    // (ten scalars, not two arrays: an array lives in a register tuple, and a masked update of one element copies it)
    uint32_t CL0 = 0u, CL1 = 0u, CL2 = 0u, CL3 = 0u, CL4 = 0u, CR0 = 0u, CR1 = 0u, CR2 = 0u, CR3 = 0u, CR4 = 0u;
    const DecEdge edge = { t == 0u ? 0u : 0x15u, t == 0u ? 0u : 0xCu, t == 31u ? 0u : 0x15u, t == 31u ? 0u : 0xCu };
    int msb = 32;
    int32_t sz = 0;
    if (valid) { msb = s16 ? (int)a.cw16[9 + 2 * cb] : stage[0]; sz = a.sizes[cb]; }
    const int32_t word0 = msb;                              // (a raw codeblock's word 0 travels in the MSB's place)
    if (valid && sz != 4096 && msb != 32 && (msb < 0 || msb > kMaxPlanes - 1)) {
        atomicOr(a.range_flag, 1);
        msb = kMaxPlanes - 1;
    }
    if (C16 && valid && sz != 4096 && msb == kMaxPlanes - 1) atomicOr(a.range_flag, 1);     // magnitudes of 16 bits: not in an int16
    const bool coded = valid && msb != 32 && sz != 4096;

    int level, sb;
    find_subband(cbx * 64 + 2 * (int)t, cby * 64, a.AW, a.AH, a.wl, level, sb);
    const int grp = level * a.g.nSub + sb;

    Coder c = { 0u, 0u, 0u, 0u, 0u, 0u, ~0ull, 64u, 64u, nullptr, t, nullptr, 0u, 0u };
    c.ldscnt = &lds_cnt[(threadIdx.x >> 6) * 2u + half];
    c.ring = cw_ring + ((threadIdx.x >> 6) * 2u + half) * kDecRing;
    c.ringaddr = lds_addr_of(c.ring);
    c.src16 = s16 ? 1u : 0u;
    if (s16) {
        const uint32_t total = (uint32_t)*a.cw16_total;     // >= 9 + 2 nCB + 1
        c.srcbase = a.cw16; c.srclim = (total < a.cw16_max ? total : a.cw16_max) - 2u;
        c.srcoff = 9u + 2u * (uint32_t)a.nCB + (uint32_t)a.cw16_offsets[cbs];
    } else {
        c.srcbase = a.staging; c.srclim = (uint32_t)a.AW * (uint32_t)a.AH - 1u;
        c.srcoff = (uint32_t)cbs * 4096u + 1u;
    }
    dec_ring_init(c);
    wave_lds_done();
    M64 sigL = { 0u, 0u }, sigR = { 0u, 0u }, refL = { 0u, 0u }, refR = { 0u, 0u };

    int cbp = 0, loff = 0;
    BulkLane bl;
    LutGeo gl = a.g;                                         // (COMPACT: the copy's own section sizes and the lane's group in it)
    int grpc = grp;
    if constexpr (BULK) { cbp = bulk_setup<COMPACT>(a, coded, msb, cbx, cby, grp, t, lds_lut + half * kTab, bl, loff, gl, grpc); bl.sgt = sign_tab; }
    else lut_to_lds(a.lut, a.g.nRef + a.g.nSig + a.g.nSign, lds_lut);
    const LutView lv = { lds_lut + (BULK ? half * kTab : 0u), a.lut, a.g.nRef + a.g.nSig + a.g.nSign,
                         (a.g.nRef + a.g.nSig + a.g.nSign) * (BULK ? a.n_tables : 1), loff };

    int np = coded ? (msb + 1 - cbp > 0 ? msb + 1 - cbp : 0) : 0;
    { int o = __shfl_xor(np, 32); np = np > o ? np : o; }
    np = (int)__builtin_amdgcn_readfirstlane((uint32_t)np);
    // (does a codeblock of the wave hold a plane above 7?  then the epilogue transposes sixteen planes)
    int msbw = coded ? msb : -1;
    { int o = __shfl_xor(msbw, 32); msbw = msbw > o ? msbw : o; }
    msbw = (int)__builtin_amdgcn_readfirstlane((uint32_t)msbw);
    prio_by_planes(np);
    PS_BPC_TRACE(1, __builtin_amdgcn_s_memrealtime() * 256ull + (unsigned long long)np);

    for (int p = 0; p < np; p++) {
        const int bp = msb - p;
        const bool act = coded && bp >= cbp;

        PLlo[0] = PLhi[0] = PRlo[0] = PRhi[0] = 0u;

        PlaneLut pl = { 0u, 0u, 0u, 0u, 0u, 0u };
        if (act) pl = plane_lut<!BULK>(lv, gl, grpc, bp);

        // ---- significance propagation pass (SPPDecoderLauncher), rows with an insignificant coeff.  A half that
        // codes nothing in this plane (its codeblock is done, all zero, raw or beyond the last one) shows every
        // coefficient as significant to the pass: never "on", and its lanes are nobody's neighbours (DecEdge).
        {
            const uint32_t allsig = act ? 0u : 0x55555555u;
            CL0 |= allsig; CL1 |= allsig; CL2 |= allsig; CL3 |= allsig; CL4 |= allsig;
            CR0 |= allsig; CR1 |= allsig; CR2 |= allsig; CR3 |= allsig; CR4 |= allsig;
        }
        {
            const uint32_t rows32 = wave_or32(act ? ~(sigL.lo & sigR.lo) : 0u);
            if (rows32 & 0xFFFFu) dec_spp_block(c, rows32 & 0xFFFFu, CL0, CL1, CR0, CR1, sigL.lo, sigR.lo, 0u, edge, pl, prec, upper_mask, cw, sign_tab);
            if (rows32 >> 16) dec_spp_block(c, rows32 >> 16, CL1, CL2, CR1, CR2, sigL.lo, sigR.lo, 16u, edge, pl, prec, upper_mask, cw, sign_tab);
            // coefficients that became significant have a 1 in this plane (:577), the others a 0 so far
            // (an idle half's register 0 holds its last plane: sig ^ ref is zero there)
            PLlo[0] |= sigL.lo ^ refL.lo; PRlo[0] |= sigR.lo ^ refR.lo;
        }
        {
            const uint32_t rows32 = wave_or32(act ? ~(sigL.hi & sigR.hi) : 0u);
            if (rows32 & 0xFFFFu) dec_spp_block(c, rows32 & 0xFFFFu, CL2, CL3, CR2, CR3, sigL.hi, sigR.hi, 0u, edge, pl, prec, upper_mask, cw, sign_tab);
            if (rows32 >> 16) dec_spp_block(c, rows32 >> 16, CL3, CL4, CR3, CR4, sigL.hi, sigR.hi, 16u, edge, pl, prec, upper_mask, cw, sign_tab);
            PLhi[0] |= sigL.hi ^ refL.hi; PRhi[0] |= sigR.hi ^ refR.hi;
        }

        // ---- magnitude refinement pass (MRPDecoderLauncher): coefficients significant before this
        // plane; afterwards every significant one is eligible
#pragma unroll
        for (int hw = 0; hw < 2; hw++) {
            uint32_t curL = hw ? PLhi[0] : PLlo[0], curR = hw ? PRhi[0] : PRlo[0];
            const uint32_t rL = act ? (hw ? refL.hi : refL.lo) : 0u, rR = act ? (hw ? refR.hi : refR.lo) : 0u;
            uint32_t rows = wave_or32(rL | rR);
            const int last = rows ? 32 - __builtin_clz(rows) : 0;           // rows [0, last) hold every row with work
            if (5 * __builtin_popcount(rows) >= 3 * last && last > 0) {
                // dense half-pass (the lower planes): every row in order, the lanes that refine a row off the carry
                // of the bit-reversed mask (shl_carry, as in the encoder), the decoded bits shifted into a word per
                // column (shl_in) instead of a select and an OR each
                uint32_t xL = bitrev32(rL), xR = bitrev32(rR), accL = 0u, accR = 0u;
#pragma unroll 1
                for (int ii = 0; ii < last; ii++) {
                    bool one;
                    dec_ring_row(c, upper_mask);
                    const uint64_t mL = shl_carry(xL);
                    uint64_t dL = 0ull, dR = 0ull;
                    if (mL != 0ull) dL = dec_site_m<false>(c, __builtin_amdgcn_inverse_ballot_w64(mL), mL, pl.ref, prec, upper_mask, cw, one);
                    shl_in(accL, dL);
                    const uint64_t mR = shl_carry(xR);
                    if (mR != 0ull) dR = dec_site_m<false>(c, __builtin_amdgcn_inverse_ballot_w64(mR), mR, pl.ref, prec, upper_mask, cw, one);
                    shl_in(accR, dR);
                }
                // row 0 sits at bit last - 1 of the accumulators
                curL |= bitrev32(accL << (32 - last));
                curR |= bitrev32(accR << (32 - last));
                rows = 0u;
            }
            while (rows) {
                const uint32_t ii = (uint32_t)__builtin_ctz(rows);
                rows &= rows - 1u;
                const bool oL = ((rL >> ii) & 1u) != 0u, oR = ((rR >> ii) & 1u) != 0u;
                const uint64_t mL = __builtin_amdgcn_ballot_w64(oL), mR = __builtin_amdgcn_ballot_w64(oR);
                dec_ring_row(c, upper_mask);
                if (mL != 0ull) curL |= dec_site_on<false>(c, oL, mL, pl.ref, prec, upper_mask, cw) ? (1u << ii) : 0u;
                if (mR != 0ull) curR |= dec_site_on<false>(c, oR, mR, pl.ref, prec, upper_mask, cw) ? (1u << ii) : 0u;
            }
            if (hw == 0) { PLlo[0] = curL; PRlo[0] = curR; } else { PLhi[0] = curL; PRhi[0] = curR; }
        }
        refL = sigL; refR = sigR;
        if (act) {                                             // plane bp of this lane's codeblock is complete
            uint32_t *q = pscr + (size_t)bp * kEncPlaneDwords;
            q[0] = PLlo[0]; q[64] = PLhi[0]; q[128] = PRlo[0]; q[192] = PRhi[0];
        }
    }
    PS_BPC_TRACE(2, __builtin_amdgcn_s_memrealtime());
    // the signs back as X-form row masks (the all-significant marks of an idle half sit on the even bits)
    const M64 sgnL = { cform_signs(CL0, CL1, CL2), cform_signs(CL2, CL3, CL4) };
    const M64 sgnR = { cform_signs(CR0, CR1, CR2), cform_signs(CR2, CR3, CR4) };

    // writeCoefficients BPCEngine.cu:94-111 / copyEntireCodeblock :1915-1922.
    // The planes come back from the scratch, eight (sixteen for a wave with a codeblock of MSB >= 8) of one
    // 32-row half at a time; planes a codeblock never decoded in its two passes -- above its MSB, and (-k > 0) below
    // its cbp: the bulk scan's -- were never written and read as zero.
    {
        const bool have = coded;
        auto planes_of = [&](auto &A, auto &B, int hw, int n) {
#pragma unroll
            for (int k = 0; k < (int)(sizeof(A) / sizeof(A[0])); k++) {
                A[k] = 0u; B[k] = 0u;
                if (k < n && have && k <= msb && k >= cbp) {
                    const uint32_t *q = pscr + (size_t)k * kEncPlaneDwords + hw * 64;
                    A[k] = q[0]; B[k] = q[128];
                }
            }
        };
        const uint16_t *const raw16 = s16 ? a.cw16 : nullptr;
        if (msbw >= kDecSmallPlanes) {
#pragma unroll 1
            for (int hw = 0; hw < 2; hw++) {
                uint32_t A[kMaxPlanes], B[kMaxPlanes];
                planes_of(A, B, hw, kMaxPlanes);
                write_rows<kMaxPlanes, kMaxPlanes, C16>(A, B, hw ? sgnL.hi : sgnL.lo, hw ? sgnR.hi : sgnR.lo, 32 * hw, valid, sz, stage, t,
                                                        obase, a.AW, raw16, c.srcoff, c.srclim + 1u, word0);
            }
        } else {
#pragma unroll 1
            for (int hw = 0; hw < 2; hw++) {
                uint32_t A[kDecSmallPlanes], B[kDecSmallPlanes];
                planes_of(A, B, hw, kDecSmallPlanes);
                write_rows<kDecSmallPlanes, kDecSmallPlanes, C16>(A, B, hw ? sgnL.hi : sgnL.lo, hw ? sgnR.hi : sgnR.lo, 32 * hw, valid, sz, stage, t,
                                                                  obase, a.AW, raw16, c.srcoff, c.srclim + 1u, word0);
            }
        }
    }
    PS_BPC_TRACE(3, __builtin_amdgcn_s_memrealtime());
    if constexpr (BULK) {
        // ---- bulk scan (decodeBulkMode :1653-1662): the planes below cbp, row by row; a row's two coefficients as the
        // epilogue left them (magnitude bits of the two-pass planes, sign) are read back two rows ahead, the scan's low bits
        // and the signs it decoded go in, the row is stored (L2-resident: the wave has just written it)
        int Bmax = bl.Bh;
        { int o = __shfl_xor(Bmax, 32); Bmax = Bmax > o ? Bmax : o; }
        Bmax = (int)__builtin_amdgcn_readfirstlane((uint32_t)Bmax);
        if (Bmax >= 0) {
            wave_stores_issued();                            // (the epilogue's stores of this lane's rows have landed)
            auto unp = [&](const M64 &sg, const M64 &sn, int r) -> uint32_t {
                if (r > 63) return 0u;
                const uint32_t g = ((r < 32 ? sg.lo : sg.hi) >> (r & 31)) & 1u, n = ((r < 32 ? sn.lo : sn.hi) >> (r & 31)) & 1u;
                return (g << 1) | (g & n);
            };
            const bool mine = valid && sz != 4096 && bl.Bh >= 0;        // this lane's codeblock has bulk planes
            // (C16: a row's two coefficients are one dword of the int16 array)
            auto row_at = [&](int i) -> int2 {
                if (!(mine && i < 64)) return make_int2(0, 0);
                if constexpr (C16) {
                    const uint32_t w = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const int16_t *>(obase) + (size_t)i * (size_t)a.AW);
                    return make_int2((int32_t)(int16_t)(w & 0xFFFFu), (int32_t)w >> 16);
                } else {
                    return *reinterpret_cast<const int2 *>(obase + (size_t)i * (size_t)a.AW);
                }
            };
            uint32_t pUL = 0u, pUR = 0u;
            int2 w0 = row_at(0), w1 = row_at(1);
            uint32_t uL = unp(sigL, sgnL, 0), uR = unp(sigR, sgnR, 0);
            for (int i = 0; i < 64; i++) {
                const int2 w2 = row_at(i + 2);
                const uint32_t dL = unp(sigL, sgnL, i + 1), dR = unp(sigR, sgnR, i + 1);
                bulk_row<true>(c, t, uL, uR, 0u, 0u, dL, dR, pUL, pUR, bl,
                               Bmax, prec, upper_mask, const_cast<int32_t *>(cw));
                uL = dL; uR = dR;
                if (mine) {
                    const uint32_t m0 = (uint32_t)(w0.x < 0 ? -w0.x : w0.x) | (pUL >> 2), m1 = (uint32_t)(w0.y < 0 ? -w0.y : w0.y) | (pUR >> 2);
                    const int32_t v0 = (pUL & 1u) ? -(int32_t)m0 : (int32_t)m0, v1 = (pUR & 1u) ? -(int32_t)m1 : (int32_t)m1;
                    store_coef_pair<C16>(obase, (size_t)i, a.AW, v0, v1);
                }
                w0 = w1; w1 = w2;
            }
        }
    }
}


// =============================================================================================
// -cp 3: three coding passes (kernelBPCCoder3CP / kernelBPCDecoder3CP BPC/BPCEngine.cu:2029-2121,
// 2221-2299; Encode3CP :1727-1776, Decode3CP :1844-1900; SPPEncoder3CP :521-553, SPPDecoder3CP :599-640,
// CPEncoder :645-680, CPDecoder :686-719 and their launchers :850-925, :1010-1243).  Deprecated in the
// reference (IO/CommandLineParser.cpp:34) and tuned only as far as its registers go: one kernel for both directions over LIVE
// row masks, as the 2-pass decoder keeps them -- in this mode even the encoder cannot form its contexts
// ahead of the scan, because whether the significance pass codes a coefficient at all (one of its eight
// neighbours is significant when the scan reaches it) depends on what the pass did to the coefficients
// before it.
//   plane MSB      : cleanup pass over every coefficient (all are flagged at the start)
//   planes below   : significance pass (insignificant coefficients with a significant neighbour; the others
//                    get the cleanup flag), refinement pass (significant before this plane), cleanup pass (the
//                    flagged ones, cp_sig / cp_sign tables = the ordinary LUT pointers + nSig + nSign)
// The table is [ref | sig | sign | cp_sig | cp_sign]; the call sites are the 2-pass kernels' (enc_site2 /
// dec_site_m), so slot order, LDS reservation and the decoder's codeword ring are shared.
// =============================================================================================
constexpr int kLutLdsMax3 = 2 * kLutLdsMax;        // bytes of one 5-section table

// One coefficient of the significance pass (CLEANUP = false) or of the cleanup pass (true), encoder
// (DEC = false: `cur` holds the plane's bits, `so` every coefficient's sign) or decoder (`cur` and `so`
// receive them).  wo/wl/wr: W-form significance of the own / left / right column, so/sl/sr: signs.  flag:
// the 32 rows' cleanup flags of this column half; elig: rows that become eligible for refinement at once
// (cleanup pass, :669-670).  Returns the ballot of the lanes whose coefficient became significant.
template <bool DEC, bool CLEANUP, class CT>
__device__ __forceinline__ uint64_t cp3_coeff(CT &c, bool idle, uint32_t ii, M64 &wo, const M64 &wl, const M64 &wr,
                                              M64 &so, const M64 &sl, const M64 &sr, uint32_t &cur, uint32_t &flag,
                                              uint32_t &elig, const PlaneLut &pl, uint32_t prec, uint32_t upper_mask,
                                              const int32_t *cwarr, const uint8_t *sgt)
{
    const uint32_t to = triple(wo, ii), tl = triple(wl, ii), tr = triple(wr, ii);
    const uint32_t ctx = (uint32_t)__builtin_popcount(to & 5u) + (uint32_t)__builtin_popcount(tl) +
                         (uint32_t)__builtin_popcount(tr);
    bool on;
    if constexpr (CLEANUP) {
        on = !idle && ((flag >> ii) & 1u) != 0u;
    } else {
        const bool cand = !idle && (to & 2u) == 0u;
        on = cand && ctx != 0u;
        if (cand && ctx == 0u) flag |= 1u << ii;                       // left to the cleanup pass (:551 / :638)
    }
    const uint64_t onm = __builtin_amdgcn_ballot_w64(on);
    if (onm == 0ull) return 0ull;
    const uint32_t p07 = __builtin_amdgcn_perm(pl.sig1, pl.sig0, ctx | 0x0C0C0C00u);
    const uint32_t p = ctx >= 8u ? pl.sig8 : p07;
    bool one;
    uint64_t onem;
    if constexpr (DEC) {
        onem = dec_site_m<true>(c, on, onm, p, prec, upper_mask, cwarr, one);
    } else {
        one = on && ((cur >> ii) & 1u) != 0u;
        onem = __builtin_amdgcn_ballot_w64(one);
        enc_site2(c, onm, onem, p, prec, upper_mask);
    }
    if constexpr (CLEANUP) { if (on) flag &= ~(1u << ii); }            // :664 / :703
    if (onem != 0ull) {
        // sign context from the four neighbours' (significant, sign) pairs; the encoder holds the signs
        // of coefficients that are not significant yet, which must not count
        const uint32_t xo = triple(so, ii) & to, xl = triple(sl, ii) & tl, xr = triple(sr, ii) & tr;
        uint32_t idx = (to & 5u) | ((xo & 5u) << 1);
        idx |= ((tl & 2u) << 3) | ((xl & 2u) << 4) | ((tr & 2u) << 5) | ((xr & 2u) << 6);
        const uint32_t tv = sgt[idx];
        const uint32_t p2 = (pl.sign >> (tv >> 3)) & 0xFFu;
        if constexpr (DEC) {
            const bool s2 = dec_site_on(c, one, onem, p2, prec, upper_mask, cwarr);
            if (one) w_set(so, ii, (s2 ? 1u : 0u) ^ (tv & 1u));
        } else {
            const uint32_t sgn = (triple(so, ii) >> 1) & 1u;            // the coefficient's own sign
            enc_site2(c, onem, onem & __builtin_amdgcn_ballot_w64(((sgn ^ tv) & 1u) != 0u), p2, prec, upper_mask);
        }
        if (one) {
            w_set(wo, ii, 1u);
            if constexpr (DEC) cur |= 1u << ii;
            if constexpr (CLEANUP) elig |= 1u << ii;
        }
    }
    return onem;
}

#ifndef PICSONG_BPC3_WG
#define PICSONG_BPC3_WG 4
#endif
constexpr int kBpc3WgWaves = PICSONG_BPC3_WG;
// (round 4) The planes live in the wave's scratch as in the two-pass kernels -- the encoder's prologue is theirs
// (enc_transpose_pass: until then a bit at a time, ~6 K instructions a wave, into 64 plane registers), the decoder parks
// a finished plane and its epilogue reads them back eight at a time -- and ONE plane is in registers: 125 / 160
// registers (4 / 3 waves a SIMD) become 80, six waves.  8K lossless, lone frame / three calls in flight: encode 0.651 ms /
// 71 Gpixel/s -> 0.60 ms / 90, decode 0.925 ms / 55 -> 0.69 ms / 76 (5 / 6 / 7 / 8 waves asked for: the same within
// 2 %; four waves to a workgroup -- one table copy in LDS for four -- over two: lone decode 0.77 -> 0.69 ms).
#ifndef PICSONG_BPC3_WAVES
#define PICSONG_BPC3_WAVES 6
#endif

template <bool DEC>
__global__ __launch_bounds__(64 * kBpc3WgWaves, PICSONG_BPC3_WAVES) void bpc3_kernel(BpcArgs a)
{
    using CT = typename std::conditional<DEC, Coder, EncCoder>::type;
    __shared__ uint8_t lds_lut[kLutLdsMax3];
    __shared__ uint8_t sign_tab[256];
    __shared__ __attribute__((aligned(1024))) uint16_t cw_ring[kBpc3WgWaves * 2 * kDecRing];
    __shared__ uint32_t lds_cnt[kBpc3WgWaves * 2];
    const uint32_t lane = threadIdx.x & 63u, half = lane >> 5, t = lane & 31u;
    sign_table_fill(sign_tab, lane);
    if (t == 0u) lds_cnt[(threadIdx.x >> 6) * 2u + half] = 0u;
    const int wave = (int)blockIdx.x * kBpc3WgWaves + (int)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cb = a.cb_base + 2 * wave + (int)half;
    const bool valid = cb < a.nCB;
    const int cbx = valid ? cb % a.ncx : 0, cby = valid ? cb / a.ncx : 0;
    const size_t cbase = (size_t)(cby * 64) * (size_t)a.AW + (size_t)(cbx * 64) + 2u * t;
    const uint32_t esz = a.c16 ? 2u : 4u;                    // bytes of a coefficient (encoder input)
    const uint32_t cbyte = (uint32_t)cbase * esz, rstride = (uint32_t)a.AW * esz;
    // the decoder's staging (32-bit, the reference's array) / the encoder's (16-bit words, BpcArgs::staging16)
    int32_t *const st = DEC ? a.staging + (size_t)(a.cb_base + 2 * wave + (int)half) * 4096u : nullptr;
    uint16_t *const stw16 = DEC ? nullptr : a.staging16 + (size_t)(a.cb_base + 2 * wave) * 4096u;
    uint16_t *const st16 = DEC ? nullptr : stw16 + (size_t)half * 4096u;
    // (a half beyond the last codeblock reads the launch's first codeblock: its ring is never used)
    const int32_t *const cw = DEC ? (valid ? st : a.staging + (size_t)a.cb_base * 4096u) + 1 : nullptr;
    const uint32_t prec = (uint32_t)a.g.prec;
    const uint32_t upper_mask = opaque_mask(half ? 0xFFFFFFFFu : 0u);
    const int total = a.g.nRef + 2 * (a.g.nSig + a.g.nSign), aux = a.g.nSig + a.g.nSign;

    // the plane being coded: rows 0-31 / 32-63 of the left and of the right column
    uint32_t PLlo = 0u, PLhi = 0u, PRlo = 0u, PRhi = 0u;
    const int gwave = (int)blockIdx.x * kBpc3WgWaves + (int)(threadIdx.x >> 6);
    uint32_t *const pscr = a.plane_scratch + (size_t)gwave * (size_t)kEncScratchDwordsPerWave + lane;
    M64 sgnL = { 0u, 0u }, sgnR = { 0u, 0u };              // X-form here: lo = rows 0-31, hi = rows 32-63
    int msb = 32;
    int32_t sz = 0;
    if constexpr (DEC) {
        if (valid) { msb = st[0]; sz = a.sizes[cb]; }
        if (valid && sz != 4096 && msb != 32 && (msb < 0 || msb > kMaxPlanes - 1)) { atomicOr(a.range_flag, 1); msb = kMaxPlanes - 1; }
    } else {
        // findMSB3CP :198-216 (the cleanup flag does not count) and the planes as row masks into the scratch: the
        // two-pass encoder's prologue (planes 8-15 only for a wave with a codeblock of MSB >= 8)
        U64 sgL = { 0u, 0u }, sgR = { 0u, 0u };
        uint32_t ormag = 0u;
        int msbmax = -1;
#pragma unroll 1
        for (int pass = 0; pass < kMaxPlanes / kEncPassPlanes; pass++) {
            if (valid) {
                if (a.c16) enc_transpose_pass<2>(a, pass, cbyte, rstride, pscr, ormag, sgL, sgR);
                else if (a.is_float) enc_transpose_pass<1>(a, pass, cbyte, rstride, pscr, ormag, sgL, sgR);
                else enc_transpose_pass<0>(a, pass, cbyte, rstride, pscr, ormag, sgL, sgR);
            }
            if (pass == 0) {
                ormag = half_or_dpp(ormag, upper_mask);
                msb = ormag ? 31 - __builtin_clz(ormag) : 32;
                if (valid && msb != 32 && msb > kMaxPlanes - 1) { atomicOr(a.range_flag, 1); msb = kMaxPlanes - 1; }
                int mm = valid && msb != 32 ? msb : -1;
                { int o = __shfl_xor(mm, 32); mm = mm > o ? mm : o; }
                msbmax = (int)__builtin_amdgcn_readfirstlane((uint32_t)mm);
            }
            if (msbmax < (pass + 1) * kEncPassPlanes) break;
        }
        sgnL = M64{ sgL.lo, sgL.hi }; sgnR = M64{ sgR.lo, sgR.hi };
    }
    const bool coded = valid && msb != 32 && sz != 4096;

    int level, sb;
    find_subband(cbx * 64 + 2 * (int)t, cby * 64, a.AW, a.AH, a.wl, level, sb);
    const int grp = level * a.g.nSub + sb;
    lut_to_lds(a.lut, total, lds_lut);                     // (ends with the workgroup barrier)
    const LutView lv = { lds_lut, a.lut, total, total, 0 };

    CT c;
    if constexpr (DEC) {
        c = Coder{ 0u, 0u, 0u, 0u, 0u, 0u, ~0ull, 64u, 64u, nullptr, t, nullptr, 0u, 0u };
        c.ldscnt = &lds_cnt[(threadIdx.x >> 6) * 2u + half];
        c.ring = cw_ring + ((threadIdx.x >> 6) * 2u + half) * kDecRing;
        c.ringaddr = lds_addr_of(c.ring);
        c.srcbase = a.staging; c.src16 = 0u; c.srclim = (uint32_t)a.AW * (uint32_t)a.AH - 1u;
        c.srcoff = (uint32_t)(valid ? cb : a.cb_base) * 4096u + 1u;
        dec_ring_init(c);
        wave_lds_done();
    } else {
        c.L = 0u; c.S = 0u; c.off = half * kStageCb;
        c.cnt_lo = 0u; c.cnt_hi = 0u; c.emptym = ~0ull;
        c.slot = half * kStageCb;
        c.ldscnt = &lds_cnt[(threadIdx.x >> 6) * 2u + half];
        c.slot0 = half * kStageCb + kStageBytes; c.lim = c.slot0 + kStageBytes * 4094u; c.pone = 1u << prec;
        if (t == 0u) *c.ldscnt = c.slot0;               // (an encoder's counter: bytes of the staging, enc_reserve)
        wave_lds_done();
        c.stw = reinterpret_cast<char *>(stw16);
    }

    int np = coded ? msb + 1 : 0;
    { int o = __shfl_xor(np, 32); np = np > o ? np : o; }
    np = (int)__builtin_amdgcn_readfirstlane((uint32_t)np);
    prio_by_planes(np);

    M64 sigL = { 0u, 0u }, sigR = { 0u, 0u }, refL = { 0u, 0u }, refR = { 0u, 0u };
    M64 flgL = { ~0u, ~0u }, flgR = { ~0u, ~0u };          // readCoefficients3CP :84-86: every coefficient flagged

    for (int p = 0; p < np; p++) {
        const int bp = msb - p;
        const bool act = coded && bp >= 0;
        const bool idle = !act;
        PLlo = PLhi = PRlo = PRhi = 0u;
        if constexpr (!DEC) {
            if (act) {                                     // the plane's row masks, as the prologue parked them
                const uint32_t *q = pscr + (size_t)bp * kEncPlaneDwords;
                PLlo = q[0]; PLhi = q[64]; PRlo = q[128]; PRhi = q[192];
            }
        }
        PlaneLut pl = { 0u, 0u, 0u, 0u, 0u, 0u }, plc = pl;
        if (act) { pl = plane_lut<true>(lv, a.g, grp, bp, 0); plc = plane_lut<true>(lv, a.g, grp, bp, aux); }

        // ---- significance pass, then refinement pass (not on a codeblock's top plane, Encode3CP :1744-1751)
        const bool top = act && p == 0 && bp == msb;       // this lane's codeblock is at its MSB plane
        const bool spp_idle = idle || top;
#pragma unroll
        for (int pass = 0; pass < 2; pass++) {
            // pass 0: significance, pass 1 (after the refinement loop below): cleanup -- same scan
            if (pass == 1) {
                // refinement pass: coefficients significant before this plane (bit 29), :726-736 / :743-762
#pragma unroll
                for (int hw = 0; hw < 2; hw++) {
                    uint32_t curL = hw ? PLhi : PLlo, curR = hw ? PRhi : PRlo;
                    const uint32_t rL = spp_idle ? 0u : (hw ? refL.hi : refL.lo), rR = spp_idle ? 0u : (hw ? refR.hi : refR.lo);
                    uint32_t rows = wave_or32(rL | rR);
                    while (rows) {
                        const uint32_t ii = (uint32_t)__builtin_ctz(rows);
                        rows &= rows - 1u;
                        const bool oL = ((rL >> ii) & 1u) != 0u, oR = ((rR >> ii) & 1u) != 0u;
                        const uint64_t mL = __builtin_amdgcn_ballot_w64(oL), mR = __builtin_amdgcn_ballot_w64(oR);
                        if constexpr (DEC) {
                            if (mL != 0ull) curL |= dec_site_on(c, oL, mL, pl.ref, prec, upper_mask, cw) ? (1u << ii) : 0u;
                            if (mR != 0ull) curR |= dec_site_on(c, oR, mR, pl.ref, prec, upper_mask, cw) ? (1u << ii) : 0u;
                        } else {
                            if (mL != 0ull) enc_site2(c, mL, mL & __builtin_amdgcn_ballot_w64(((curL >> ii) & 1u) != 0u), pl.ref, prec, upper_mask);
                            if (mR != 0ull) enc_site2(c, mR, mR & __builtin_amdgcn_ballot_w64(((curR >> ii) & 1u) != 0u), pl.ref, prec, upper_mask);
                        }
                    }
                    if constexpr (DEC) { if (hw == 0) { PLlo = curL; PRlo = curR; } else { PLhi = curL; PRhi = curR; } }
                }
                // coefficients the significance pass made significant become eligible now (MRP's else branch)
                if (!spp_idle) { refL.lo |= sigL.lo; refL.hi |= sigL.hi; refR.lo |= sigR.lo; refR.hi |= sigR.hi; }
            }
            const bool pidle = pass == 0 ? spp_idle : idle;
            const PlaneLut &plp = pass == 0 ? pl : plc;
#pragma unroll
            for (int hw = 0; hw < 2; hw++) {
                M64 wL = to_w(sigL, hw), wR = to_w(sigR, hw);
                M64 sL = to_w(sgnL, hw), sR = to_w(sgnR, hw);
                M64 wPR = { from_prev32(wR.lo, t), from_prev32(wR.hi, t) };
                M64 wNL = { from_next32(wL.lo, t), from_next32(wL.hi, t) };
                M64 sPR = { from_prev32(sR.lo, t), from_prev32(sR.hi, t) };
                M64 sNL = { from_next32(sL.lo, t), from_next32(sL.hi, t) };
                uint32_t curL = hw ? PLhi : PLlo, curR = hw ? PRhi : PRlo;
                uint32_t fL = hw ? flgL.hi : flgL.lo, fR = hw ? flgR.hi : flgR.lo;
                uint32_t eL = 0u, eR = 0u;
                const uint32_t xl = hw ? sigL.hi : sigL.lo, xr = hw ? sigR.hi : sigR.lo;
                uint32_t rows = wave_or32(pidle ? 0u : (pass == 0 ? ~(xl & xr) : (fL | fR)));
                while (rows) {
                    const uint32_t ii = (uint32_t)__builtin_ctz(rows);
                    rows &= rows - 1u;
                    uint64_t bL, bR;
                    if (pass == 0) bL = cp3_coeff<DEC, false>(c, pidle, ii, wL, wPR, wR, sL, sPR, sR, curL, fL, eL, plp, prec, upper_mask, cw, sign_tab);
                    else           bL = cp3_coeff<DEC, true>(c, pidle, ii, wL, wPR, wR, sL, sPR, sR, curL, fL, eL, plp, prec, upper_mask, cw, sign_tab);
                    if (bL != 0ull) {                      // lane+1's left column as it is after the left phase
                        wNL.lo = from_next32(wL.lo, t); wNL.hi = from_next32(wL.hi, t);
                        if constexpr (DEC) { sNL.lo = from_next32(sL.lo, t); sNL.hi = from_next32(sL.hi, t); }
                    }
                    if (pass == 0) bR = cp3_coeff<DEC, false>(c, pidle, ii, wR, wL, wNL, sR, sL, sNL, curR, fR, eR, plp, prec, upper_mask, cw, sign_tab);
                    else           bR = cp3_coeff<DEC, true>(c, pidle, ii, wR, wL, wNL, sR, sL, sNL, curR, fR, eR, plp, prec, upper_mask, cw, sign_tab);
                    if (bR != 0ull) {                      // lane-1's right column after the right phase
                        wPR.lo = from_prev32(wR.lo, t); wPR.hi = from_prev32(wR.hi, t);
                        if constexpr (DEC) { sPR.lo = from_prev32(sR.lo, t); sPR.hi = from_prev32(sR.hi, t); }
                    }
                }
                if (hw == 0) { sigL.lo = w_rows(wL); sigR.lo = w_rows(wR); flgL.lo = fL; flgR.lo = fR; refL.lo |= eL; refR.lo |= eR; }
                else         { sigL.hi = w_rows(wL); sigR.hi = w_rows(wR); flgL.hi = fL; flgR.hi = fR; refL.hi |= eL; refR.hi |= eR; }
                if constexpr (DEC) {
                    if (hw == 0) { sgnL.lo = w_rows(sL); sgnR.lo = w_rows(sR); PLlo = curL; PRlo = curR; }
                    else         { sgnL.hi = w_rows(sL); sgnR.hi = w_rows(sR); PLhi = curL; PRhi = curR; }
                }
            }
        }
        if constexpr (DEC) {
            if (act) {                                     // plane bp of this lane's codeblock is complete
                uint32_t *q = pscr + (size_t)bp * kEncPlaneDwords;
                q[0] = PLlo; q[64] = PLhi; q[128] = PRlo; q[192] = PRhi;
            }
        }
    }

    if constexpr (DEC) {
        // the planes back from the scratch, eight (sixteen for a wave with a codeblock of MSB >= 8) of one 32-row half
        // at a time, as in bpc_decode_kernel; planes above a codeblock's MSB were never written and read as zero
        int msbw = coded ? msb : -1;
        { int o = __shfl_xor(msbw, 32); msbw = msbw > o ? msbw : o; }
        msbw = (int)__builtin_amdgcn_readfirstlane((uint32_t)msbw);
        auto planes_of = [&](auto &A, auto &B, int hw) {
#pragma unroll
            for (int k = 0; k < (int)(sizeof(A) / sizeof(A[0])); k++) {
                A[k] = 0u; B[k] = 0u;
                if (coded && k <= msb) {
                    const uint32_t *q = pscr + (size_t)k * kEncPlaneDwords + hw * 64;
                    A[k] = q[0]; B[k] = q[128];
                }
            }
        };
        if (msbw >= kDecSmallPlanes) {
#pragma unroll 1
            for (int hw = 0; hw < 2; hw++) {
                uint32_t A[kMaxPlanes], B[kMaxPlanes];
                planes_of(A, B, hw);
                write_rows<kMaxPlanes, kMaxPlanes>(A, B, hw ? sgnL.hi : sgnL.lo, hw ? sgnR.hi : sgnR.lo, 32 * hw, valid, sz, st, t, a.coeffs_out + cbase, a.AW);
            }
        } else {
#pragma unroll 1
            for (int hw = 0; hw < 2; hw++) {
                uint32_t A[kDecSmallPlanes], B[kDecSmallPlanes];
                planes_of(A, B, hw);
                write_rows<kDecSmallPlanes, kDecSmallPlanes>(A, B, hw ? sgnL.hi : sgnL.lo, hw ? sgnR.hi : sgnR.lo, 32 * hw, valid, sz, st, t, a.coeffs_out + cbase, a.AW);
            }
        }
    } else {
        // flush + sizeArray + MSB word + expansion fallback, as bpc_encode_kernel
#if PS_ENC_LDS
        if (coded) {
            const uint32_t sl = c.slot < c.lim ? c.slot : c.lim;
            *reinterpret_cast<uint16_t *>(c.stw + sl) = (uint16_t)c.L;
        }
        wave_lds_done();
        const uint32_t cw_count = (*c.ldscnt - c.slot0) / kStageBytes;
        const uint32_t size = (cw_count > 4095u ? 4095u : cw_count) + 1u;
#else
        if (coded) *reinterpret_cast<uint16_t *>(c.stw + c.off) = (uint16_t)c.L;
        const uint32_t size = (half ? c.cnt_hi : c.cnt_lo) + 1u;
#endif
        if (valid && t == 0u) a.sizes[cb] = (int32_t)size;
        wave_stores_issued();
        if (valid && size == 4096u) {
            for (int i = 0; i < 64; i++) {
                uint32_t m0, m1, n0, n1;
                load_row(a, cbyte + (uint32_t)i * rstride, m0, m1, n0, n1);
                const uint32_t w0 = ((m0 << 1) + n0) & 0xFFFFu, w1 = ((m1 << 1) + n1) & 0xFFFFu;
                *reinterpret_cast<uint32_t *>(st16 + t * 128u + 2u * (uint32_t)i) = w0 | (w1 << 16);
            }
        } else if (valid && t == 0u) {
            st16[0] = (uint16_t)msb;
        }
    }
}

}  // namespace picsong
