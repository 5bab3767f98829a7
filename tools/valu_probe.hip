// valu_probe.hip -- measures what one MI355X SIMD actually issues, for the instruction mix of the BPC
// coder (bpc_kernels.hpp): independent and dependent chains of v_and_b32 / v_alignbit_b32 /
// v_mul_u32_u24 / v_perm_b32 / v_mbcnt / DPP moves, and the coder's call-site round trips
// (VALU compare -> SGPR mask -> SALU -> exec-masked VALU, with and without a branch), at 1..8 resident
// waves per SIMD.  Prints, per pattern and occupancy, cycles per wave-instruction as one wave sees them
// (s_memtime) and wave-instructions per cycle per SIMD for the whole chip (HIP events + measured clock).
// bench.py takes its VALU issue peak from this table (tools/valu_probe.json) and from the guide.
//
// build: hipcc -O3 --offload-arch=gfx950 -o valu_probe valu_probe.hip ; run: ./valu_probe [json path]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

enum Kind {
    AND_INDEP, AND_DEP, ALIGNBIT_INDEP, ALIGNBIT_DEP, MUL24_INDEP, MUL24_DEP, PERM_INDEP, PERM_DEP,
    MBCNT_DEP, DPP_INDEP, DPP_DEP, CNDMASK_DEP, BFE_INDEP,
    CMP_SALU_CND,        // v_cmp -> s_and -> v_cndmask, dependent (the ballot / mask round trip)
    CMP_SAVEEXEC,        // v_cmp -> s_and_saveexec -> v_add -> s_or exec (an exec-masked region)
    CMP_BRANCH_NT,       // v_cmp -> s_and -> s_cmp -> s_cbranch (not taken) -> v_add
    CMP_BRANCH_T,        // the same, taken (forward over 2 instructions)
    SALU_DEP,            // s_add chain
    SITE_LIKE,           // one call site of the coder, as compiled today (instruction classes and order)
    CNDMASK_E64_DEP,     // v_cndmask_b32_e64 with an SGPR-pair mask, dependent
    CNDMASK_VCC_INDEP,   // v_cndmask_b32 (vcc) x8 independent
    ADD_INDEP, LSHL_OR_INDEP, AND_OR_INDEP, XAD_DEP, ADD3_DEP, CMP_INDEP,
    SITE_V2,             // the round-2 call site: one reservation round trip, update in every lane
    OR_INDEP, XOR_INDEP, LSHL_INDEP, LSHR_SGPR_INDEP, SUB_INDEP, MOV_INDEP, MIN_INDEP, NOT_INDEP, FFBL_INDEP, BCNT_INDEP,
    CMP_E32_INDEP,       // v_cmp_eq_u32_e32 -> vcc
    CMP_CND_E32_PAIR,    // v_cmp_e32 vcc ; v_cndmask_e32 vcc
    CND_E32_AFTER_SMOV,  // s_mov_b64 vcc once per block, then v_cndmask_e32 x8 independent
    ADDCO_DEP,           // v_add_co_u32 x, vcc, x, x (shift left by one, bit out to vcc)
    AND_SGPR_INDEP, AND_LIT_INDEP, EXEC_MOV,      // s_mov exec, m ; v_mov ; s_mov exec, -1
    SDWA_AND_INDEP, PK_ADD_INDEP, MUL_LO_INDEP, MAD24_INDEP, BFI_INDEP, LSHL_ADD_INDEP,
    // fp32 classes of the 9/7 DWT kernels (dwt_kernels.hpp)
    F_ADD_INDEP, F_MUL_INDEP, F_FMAC_INDEP, F_FMA_INDEP, F_FMAMK_INDEP, F_ADD_DPP_INDEP, F_CVT_UBYTE_INDEP, F_CVT_I32_INDEP,
    PK_FMA_INDEP, PK_ADDF_INDEP, PK_MULF_INDEP, PK_FMA_SGPR_INDEP, MOV_B64_INDEP, F_ADD_DEP, PK_FMA_DEP,
    NKINDS
};
static const char *kKindName[NKINDS] = {
    "v_and_b32 x8 independent", "v_and_b32 dependent", "v_alignbit_b32 x8 independent", "v_alignbit_b32 dependent",
    "v_mul_u32_u24 x8 independent", "v_mul_u32_u24 dependent", "v_perm_b32 x8 independent", "v_perm_b32 dependent",
    "v_mbcnt_lo+hi dependent", "v_mov_b32 dpp wave_shr:1 x8 independent", "v_mov_b32 dpp wave_shr:1 dependent",
    "v_cndmask_b32 dependent", "v_bfe_u32 x8 independent",
    "v_cmp -> s_and_b64 -> v_cndmask (3 inst round trip)", "v_cmp -> s_and_saveexec -> v_add -> s_or exec (4 inst)",
    "v_cmp -> s_and -> s_cmp -> s_cbranch not taken -> v_add (5 inst)", "v_cmp -> s_and -> s_cmp -> s_cbranch taken -> v_add (5 inst)",
    "s_add_u32 dependent", "coder call site mix (24 inst: 15 VALU, 9 SALU)",
    "v_cndmask_b32_e64 sgpr mask dependent", "v_cndmask_b32 vcc x8 independent", "v_add_u32 x8 independent",
    "v_lshl_or_b32 x8 independent", "v_and_or_b32 x8 independent", "v_xad_u32 dependent", "v_add3_u32 dependent",
    "v_cmp_ne_u32 -> sgpr pair x4 independent", "round-2 call site (21 inst: 17 VALU, 4 SALU)",
    "v_or_b32 x8 independent", "v_xor_b32 x8 independent", "v_lshlrev_b32 (inline shift) x8 independent",
    "v_lshrrev_b32 (sgpr shift) x8 independent", "v_sub_u32 x8 independent", "v_mov_b32 x8 independent",
    "v_min_u32 x8 independent", "v_not_b32 x8 independent", "v_ffbl_b32 x8 independent", "v_bcnt_u32_b32 x8 independent",
    "v_cmp_eq_u32_e32 -> vcc x8 independent", "v_cmp_e32 vcc ; v_cndmask_e32 vcc (pairs, x8 independent)",
    "s_mov vcc ; 8 x v_cndmask_e32 vcc independent (9 inst)", "v_add_co_u32 x, vcc, x, x dependent",
    "v_and_b32 v, s, v x8 independent", "v_and_b32 v, literal, v x8 independent",
    "s_mov exec, m ; v_mov ; s_mov exec, -1 (3 inst)", "v_and_b32_sdwa x8 independent", "v_pk_add_u16 x8 independent",
    "v_mul_lo_u32 x8 independent", "v_mad_u32_u24 x8 independent", "v_bfi_b32 x8 independent", "v_lshl_add_u32 x8 independent",
    "v_add_f32 x8 independent", "v_mul_f32 x8 independent", "v_fmac_f32 x8 independent", "v_fma_f32 x8 independent",
    "v_fmaak_f32 (literal) x8 independent", "v_add_f32_dpp wave_shl:1 x8 independent", "v_cvt_f32_ubyte1 x8 independent",
    "v_cvt_f32_i32 x8 independent", "v_pk_fma_f32 x8 independent", "v_pk_add_f32 x8 independent", "v_pk_mul_f32 x8 independent",
    "v_pk_fma_f32 v, s, v x8 independent", "v_mov_b64 x8 independent", "v_add_f32 dependent", "v_pk_fma_f32 dependent" };
// instructions per unrolled block (what "per instruction" divides by)
static const int kBlockInsts[NKINDS] = { 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 64, 32, 32, 24, 32, 40, 40, 32, 24, 32, 32, 32, 32, 32, 32, 32, 32, 21, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 64, 36, 32, 32, 32, 24, 32, 32, 32, 32, 32, 32,
    32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32 };
static const int kBlockValu[NKINDS] = { 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 16, 16, 16, 16, 0, 15, 32, 32, 32, 32, 32, 32, 32, 32, 17, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 64, 32, 32, 32, 32, 8, 32, 32, 32, 32, 32, 32,
    32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32 };

#define R4(x) x x x x
#define R8(x) R4(x) R4(x)
#define R16(x) R8(x) R8(x)
#define R32(x) R16(x) R16(x)

#define IND8_2(op) asm volatile(R4(op " %0, %0, %8\n" op " %1, %1, %8\n" op " %2, %2, %8\n" op " %3, %3, %8\n" \
                                    op " %4, %4, %8\n" op " %5, %5, %8\n" op " %6, %6, %8\n" op " %7, %7, %8\n") \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k))
#define IND8_2R(op) asm volatile(R4(op " %0, %8, %0\n" op " %1, %8, %1\n" op " %2, %8, %2\n" op " %3, %8, %3\n" \
                                     op " %4, %8, %4\n" op " %5, %8, %5\n" op " %6, %8, %6\n" op " %7, %8, %7\n") \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(sc))
#define IND8_1(op) asm volatile(R4(op " %0, %0\n" op " %1, %1\n" op " %2, %2\n" op " %3, %3\n" \
                                    op " %4, %4\n" op " %5, %5\n" op " %6, %6\n" op " %7, %7\n") \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7))
#define IND8_3(op, tail) asm volatile(R4(op " %0, %0, %8" tail "\n" op " %1, %1, %8" tail "\n" op " %2, %2, %8" tail "\n" op " %3, %3, %8" tail "\n" \
                                    op " %4, %4, %8" tail "\n" op " %5, %5, %8" tail "\n" op " %6, %6, %8" tail "\n" op " %7, %7, %8" tail "\n") \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k))

#define PK8(o0, o1, o2, o3, o4, o5, o6, o7, cls) asm volatile(R4(o0 o1 o2 o3 o4 o5 o6 o7) \
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : cls(pk))

template <int KIND>
__global__ __launch_bounds__(256) void probe_kernel(uint32_t *out, uint64_t *cyc, int iters, uint32_t seed)
{
    uint32_t a0 = threadIdx.x * 2654435761u + seed, a1 = a0 ^ 0x9E3779B9u, a2 = a0 + 77u, a3 = a1 + 5u;
    uint32_t a4 = a0 * 3u, a5 = a1 * 5u, a6 = a2 * 7u, a7 = a3 * 11u, k = seed | 0x01010101u;
    uint64_t sm = 0, sm2 = 0x5555aaaa3333ccccull, sm3 = ~0ull, sm4 = 0;            // scalar mask scratch
    uint32_t sc = seed;
    // 64-bit operands of the packed fp32 classes (finite, non-trivial floats in both halves)
    uint64_t p0 = 0x3f8000013f800003ull + threadIdx.x, p1 = p0 + 17u, p2 = p0 + 33u, p3 = p0 + 49u, p4 = p0 + 65u, p5 = p0 + 81u,
             p6 = p0 + 97u, p7 = p0 + 113u, pk = 0x3f7fff003f7ffe00ull;
    __builtin_amdgcn_s_barrier();
    const uint64_t r0 = __builtin_amdgcn_s_memrealtime();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        if constexpr (KIND == AND_INDEP) {
            asm volatile(R4("v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n"
                            "v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));
        } else if constexpr (KIND == AND_DEP) {
            asm volatile(R32("v_and_b32 %0, %0, %1\n") : "+v"(a0) : "v"(k));
        } else if constexpr (KIND == ALIGNBIT_INDEP) {
            asm volatile(R4("v_alignbit_b32 %0, %0, %0, 3\n v_alignbit_b32 %1, %1, %1, 3\n v_alignbit_b32 %2, %2, %2, 3\n v_alignbit_b32 %3, %3, %3, 3\n"
                            "v_alignbit_b32 %4, %4, %4, 3\n v_alignbit_b32 %5, %5, %5, 3\n v_alignbit_b32 %6, %6, %6, 3\n v_alignbit_b32 %7, %7, %7, 3\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if constexpr (KIND == ALIGNBIT_DEP) {
            asm volatile(R32("v_alignbit_b32 %0, %0, %0, 3\n") : "+v"(a0));
        } else if constexpr (KIND == MUL24_INDEP) {
            asm volatile(R4("v_mul_u32_u24 %0, %0, %8\n v_mul_u32_u24 %1, %1, %8\n v_mul_u32_u24 %2, %2, %8\n v_mul_u32_u24 %3, %3, %8\n"
                            "v_mul_u32_u24 %4, %4, %8\n v_mul_u32_u24 %5, %5, %8\n v_mul_u32_u24 %6, %6, %8\n v_mul_u32_u24 %7, %7, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));
        } else if constexpr (KIND == MUL24_DEP) {
            asm volatile(R32("v_mul_u32_u24 %0, %0, %1\n") : "+v"(a0) : "v"(k));
        } else if constexpr (KIND == PERM_INDEP) {
            asm volatile(R4("v_perm_b32 %0, %0, %8, %8\n v_perm_b32 %1, %1, %8, %8\n v_perm_b32 %2, %2, %8, %8\n v_perm_b32 %3, %3, %8, %8\n"
                            "v_perm_b32 %4, %4, %8, %8\n v_perm_b32 %5, %5, %8, %8\n v_perm_b32 %6, %6, %8, %8\n v_perm_b32 %7, %7, %8, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));
        } else if constexpr (KIND == PERM_DEP) {
            asm volatile(R32("v_perm_b32 %0, %0, %1, %1\n") : "+v"(a0) : "v"(k));
        } else if constexpr (KIND == MBCNT_DEP) {
            asm volatile(R16("v_mbcnt_lo_u32_b32 %0, %1, %0\n v_mbcnt_hi_u32_b32 %0, %1, %0\n") : "+v"(a0) : "s"(sc));
        } else if constexpr (KIND == DPP_INDEP) {
            asm volatile(R4("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
                            "v_mov_b32_dpp %2, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
                            "v_mov_b32_dpp %4, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %5 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
                            "v_mov_b32_dpp %6, %6 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %7 wave_shr:1 row_mask:0xf bank_mask:0xf\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if constexpr (KIND == DPP_DEP) {
            asm volatile(R32("s_nop 1\n v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n") : "+v"(a0));
        } else if constexpr (KIND == CNDMASK_DEP) {
            asm volatile(R32("v_cndmask_b32 %0, %0, %1, vcc\n") : "+v"(a0) : "v"(k) : "vcc");
        } else if constexpr (KIND == BFE_INDEP) {
            asm volatile(R4("v_bfe_u32 %0, %0, %8, 9\n v_bfe_u32 %1, %1, %8, 9\n v_bfe_u32 %2, %2, %8, 9\n v_bfe_u32 %3, %3, %8, 9\n"
                            "v_bfe_u32 %4, %4, %8, 9\n v_bfe_u32 %5, %5, %8, 9\n v_bfe_u32 %6, %6, %8, 9\n v_bfe_u32 %7, %7, %8, 9\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));
        } else if constexpr (KIND == CMP_SALU_CND) {
            asm volatile(R8("v_cmp_ne_u32_e64 %1, %0, %2\n s_and_b64 %1, %1, exec\n v_cndmask_b32_e64 %0, %0, %2, %1\n")
                         : "+v"(a0), "+s"(sm) : "v"(k) : "scc");
        } else if constexpr (KIND == CMP_SAVEEXEC) {
            asm volatile(R8("v_cmp_ne_u32_e64 vcc, %0, %2\n s_and_saveexec_b64 %1, vcc\n v_add_u32 %0, %0, %2\n s_or_b64 exec, exec, %1\n")
                         : "+v"(a0), "+s"(sm) : "v"(k) : "vcc", "scc");
        } else if constexpr (KIND == CMP_BRANCH_NT) {
            // a0 != k always (k has bit 0 set per byte, a0 changes): mask nonzero -> scc1 branch on "== 0" not taken
            asm volatile(R8("v_cmp_ne_u32_e64 %1, %0, %2\n s_and_b64 %1, %1, exec\n s_cmp_eq_u64 %1, 0\n s_cbranch_scc1 1\n v_add_u32 %0, %0, %2\n")
                         : "+v"(a0), "+s"(sm) : "v"(k) : "scc");
        } else if constexpr (KIND == CMP_BRANCH_T) {
            asm volatile(R8("v_cmp_ne_u32_e64 %1, %0, %2\n s_and_b64 %1, %1, exec\n s_cmp_lg_u64 %1, 0\n s_cbranch_scc1 1\n v_add_u32 %0, %0, %2\n")
                         : "+v"(a0), "+s"(sm) : "v"(k) : "scc");
        } else if constexpr (KIND == SALU_DEP) {
            asm volatile(R32("s_add_u32 %0, %0, 3\n") : "+s"(sc) : : "scc");
        } else if constexpr (KIND == SITE_LIKE) {
            // the significance call site of bpc_encode_kernel as compiled in round 1 (see the disassembly):
            // on-compare, need-mask, reservation under exec, update under exec, exhausted-compare, store region
            asm volatile(
                "v_and_b32 %2, %4, %0\n v_cmp_ne_u32_e64 vcc, 0, %2\n s_and_b64 %3, vcc, exec\n s_cmp_eq_u64 %3, 0\n s_cbranch_scc1 1\n s_nop 0\n"
                "s_and_saveexec_b64 %3, vcc\n v_alignbit_b32 %2, %0, %0, 5\n v_and_b32 %2, 1, %2\n v_alignbit_b32 %1, %1, %1, 7\n"
                "v_and_or_b32 %2, %1, 2, %2\n v_perm_b32 %2, %0, %1, %2\n v_mul_u32_u24 %2, %0, %2\n v_lshrrev_b32 %2, 7, %2\n"
                "v_add_u32 %2, %2, %4\n v_sub_u32 %1, %0, %2\n v_cndmask_b32 %0, %1, %2, vcc\n v_mad_u32_u24 %1, %2, %4, %1\n"
                "s_or_b64 exec, exec, %3\n v_cmp_ne_u32_e64 %3, 0, %0\n s_and_b64 %3, %3, vcc\n s_and_saveexec_b64 %3, %3\n v_add_u32 %1, %1, %4\n s_or_b64 exec, exec, %3\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+s"(sm) : "v"(k) : "vcc", "scc");
        } else if constexpr (KIND == CNDMASK_E64_DEP) {
            asm volatile(R32("v_cndmask_b32_e64 %0, %0, %1, %2\n") : "+v"(a0) : "v"(k), "s"(sm2));
        } else if constexpr (KIND == CNDMASK_VCC_INDEP) {
            asm volatile(R4("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                            "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k) : "vcc");
        } else if constexpr (KIND == ADD_INDEP) {
            asm volatile(R4("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                            "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));
        } else if constexpr (KIND == LSHL_OR_INDEP) {
            asm volatile(R4("v_lshl_or_b32 %0, %0, 3, %8\n v_lshl_or_b32 %1, %1, 3, %8\n v_lshl_or_b32 %2, %2, 3, %8\n v_lshl_or_b32 %3, %3, 3, %8\n"
                            "v_lshl_or_b32 %4, %4, 3, %8\n v_lshl_or_b32 %5, %5, 3, %8\n v_lshl_or_b32 %6, %6, 3, %8\n v_lshl_or_b32 %7, %7, 3, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));
        } else if constexpr (KIND == AND_OR_INDEP) {
            asm volatile(R4("v_and_or_b32 %0, %0, %8, %8\n v_and_or_b32 %1, %1, %8, %8\n v_and_or_b32 %2, %2, %8, %8\n v_and_or_b32 %3, %3, %8, %8\n"
                            "v_and_or_b32 %4, %4, %8, %8\n v_and_or_b32 %5, %5, %8, %8\n v_and_or_b32 %6, %6, %8, %8\n v_and_or_b32 %7, %7, %8, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));
        } else if constexpr (KIND == XAD_DEP) {
            asm volatile(R32("v_xad_u32 %0, %0, %1, %1\n") : "+v"(a0) : "v"(k));
        } else if constexpr (KIND == ADD3_DEP) {
            asm volatile(R32("v_add3_u32 %0, %0, %1, 1\n") : "+v"(a0) : "v"(k));
        } else if constexpr (KIND == CMP_INDEP) {
            asm volatile(R8("v_cmp_ne_u32_e64 %4, %0, %8\n v_cmp_ne_u32_e64 %5, %1, %8\n v_cmp_ne_u32_e64 %6, %2, %8\n v_cmp_ne_u32_e64 %7, %3, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(sm), "+s"(sm2), "+s"(sm3), "+s"(sm4) : "v"(k));
        } else if constexpr (KIND == SITE_V2) {
            // two row-bit ballots, need mask + branch, (reservation skipped), probability select for idle lanes,
            // interval update in every lane, exhausted compare: the round-2 call site
            asm volatile(
                "v_and_b32 %2, %5, %0\n v_cmp_eq_u32_e64 %3, 0, %2\n v_and_b32 %2, %5, %1\n v_cmp_ne_u32_e64 %4, 0, %2\n"
                "s_and_b64 vcc, %6, %3\n s_cmp_eq_u64 vcc, 0\n s_cbranch_scc1 1\n s_nop 0\n"
                "v_alignbit_b32 %2, %0, %0, 5\n v_and_b32 %2, 1, %2\n v_alignbit_b32 %1, %1, %1, 7\n v_and_or_b32 %2, %1, 2, %2\n v_perm_b32 %2, %0, %1, %2\n"
                "v_cndmask_b32_e64 %2, %5, %2, %3\n v_mul_u32_u24 %2, %0, %2\n v_lshrrev_b32 %2, 7, %2\n v_xad_u32 %1, %2, -1, %0\n"
                "v_cndmask_b32_e64 %0, %2, %1, %4\n v_add3_u32 %2, %1, %2, 1\n v_cndmask_b32_e64 %1, %1, %2, %4\n v_cmp_eq_u32_e64 %6, 0, %0\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+s"(sm), "+s"(sm2), "+v"(a3), "+s"(sm3) : : "vcc", "scc");
        } else if constexpr (KIND == OR_INDEP) { IND8_2("v_or_b32");
        } else if constexpr (KIND == XOR_INDEP) { IND8_2("v_xor_b32");
        } else if constexpr (KIND == LSHL_INDEP) {
            asm volatile(R4("v_lshlrev_b32 %0, 3, %0\n v_lshlrev_b32 %1, 3, %1\n v_lshlrev_b32 %2, 3, %2\n v_lshlrev_b32 %3, 3, %3\n"
                            "v_lshlrev_b32 %4, 3, %4\n v_lshlrev_b32 %5, 3, %5\n v_lshlrev_b32 %6, 3, %6\n v_lshlrev_b32 %7, 3, %7\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if constexpr (KIND == LSHR_SGPR_INDEP) { IND8_2R("v_lshrrev_b32");
        } else if constexpr (KIND == SUB_INDEP) { IND8_2("v_sub_u32");
        } else if constexpr (KIND == MOV_INDEP) {
            asm volatile(R4("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));
        } else if constexpr (KIND == MIN_INDEP) { IND8_2("v_min_u32");
        } else if constexpr (KIND == NOT_INDEP) { IND8_1("v_not_b32");
        } else if constexpr (KIND == FFBL_INDEP) { IND8_1("v_ffbl_b32");
        } else if constexpr (KIND == BCNT_INDEP) { IND8_2("v_bcnt_u32_b32");
        } else if constexpr (KIND == CMP_E32_INDEP) {
            asm volatile(R4("v_cmp_eq_u32_e32 vcc, %0, %8\n v_cmp_eq_u32_e32 vcc, %1, %8\n v_cmp_eq_u32_e32 vcc, %2, %8\n v_cmp_eq_u32_e32 vcc, %3, %8\n"
                            "v_cmp_eq_u32_e32 vcc, %4, %8\n v_cmp_eq_u32_e32 vcc, %5, %8\n v_cmp_eq_u32_e32 vcc, %6, %8\n v_cmp_eq_u32_e32 vcc, %7, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k) : "vcc");
        } else if constexpr (KIND == CMP_CND_E32_PAIR) {
            asm volatile(R4("v_cmp_eq_u32_e32 vcc, %0, %8\n v_cndmask_b32 %0, %0, %8, vcc\n v_cmp_eq_u32_e32 vcc, %1, %8\n v_cndmask_b32 %1, %1, %8, vcc\n"
                            "v_cmp_eq_u32_e32 vcc, %2, %8\n v_cndmask_b32 %2, %2, %8, vcc\n v_cmp_eq_u32_e32 vcc, %3, %8\n v_cndmask_b32 %3, %3, %8, vcc\n"
                            "v_cmp_eq_u32_e32 vcc, %4, %8\n v_cndmask_b32 %4, %4, %8, vcc\n v_cmp_eq_u32_e32 vcc, %5, %8\n v_cndmask_b32 %5, %5, %8, vcc\n"
                            "v_cmp_eq_u32_e32 vcc, %6, %8\n v_cndmask_b32 %6, %6, %8, vcc\n v_cmp_eq_u32_e32 vcc, %7, %8\n v_cndmask_b32 %7, %7, %8, vcc\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k) : "vcc");
        } else if constexpr (KIND == CND_E32_AFTER_SMOV) {
            asm volatile(R4("s_mov_b64 vcc, %9\n v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                            "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k), "s"(sm2) : "vcc");
        } else if constexpr (KIND == ADDCO_DEP) {
            asm volatile(R32("v_add_co_u32 %0, vcc, %0, %0\n") : "+v"(a0) : : "vcc");
        } else if constexpr (KIND == AND_SGPR_INDEP) { IND8_2R("v_and_b32");
        } else if constexpr (KIND == AND_LIT_INDEP) {
            asm volatile(R4("v_and_b32 %0, 0x12345678, %0\n v_and_b32 %1, 0x12345678, %1\n v_and_b32 %2, 0x12345678, %2\n v_and_b32 %3, 0x12345678, %3\n"
                            "v_and_b32 %4, 0x12345678, %4\n v_and_b32 %5, 0x12345678, %5\n v_and_b32 %6, 0x12345678, %6\n v_and_b32 %7, 0x12345678, %7\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if constexpr (KIND == EXEC_MOV) {
            asm volatile(R8("s_mov_b64 exec, %1\n v_mov_b32 %0, %2\n s_mov_b64 exec, -1\n") : "+v"(a0) : "s"(sm2), "v"(k));
        } else if constexpr (KIND == SDWA_AND_INDEP) { IND8_3("v_and_b32_sdwa", " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD");
        } else if constexpr (KIND == PK_ADD_INDEP) { IND8_2("v_pk_add_u16");
        } else if constexpr (KIND == MUL_LO_INDEP) { IND8_2("v_mul_lo_u32");
        } else if constexpr (KIND == MAD24_INDEP) { IND8_3("v_mad_u32_u24", ", %8");
        } else if constexpr (KIND == BFI_INDEP) { IND8_3("v_bfi_b32", ", %8");
        } else if constexpr (KIND == LSHL_ADD_INDEP) { IND8_3("v_lshl_add_u32", ", %8");
        } else if constexpr (KIND == F_ADD_INDEP) { IND8_2("v_add_f32");
        } else if constexpr (KIND == F_MUL_INDEP) { IND8_2("v_mul_f32");
        } else if constexpr (KIND == F_FMAC_INDEP) { IND8_2("v_fmac_f32");
        } else if constexpr (KIND == F_FMA_INDEP) { IND8_3("v_fma_f32", ", %8");
        } else if constexpr (KIND == F_FMAMK_INDEP) { IND8_3("v_fmaak_f32", ", 0x3f99999a");
        } else if constexpr (KIND == F_ADD_DPP_INDEP) { IND8_3("v_add_f32_dpp", " wave_shl:1 row_mask:0xf bank_mask:0xf");
        } else if constexpr (KIND == F_CVT_UBYTE_INDEP) { IND8_1("v_cvt_f32_ubyte1");
        } else if constexpr (KIND == F_CVT_I32_INDEP) { IND8_1("v_cvt_f32_i32");
        } else if constexpr (KIND == PK_FMA_INDEP) { PK8("v_pk_fma_f32 %0, %0, %8, %0\n", "v_pk_fma_f32 %1, %1, %8, %1\n", "v_pk_fma_f32 %2, %2, %8, %2\n", "v_pk_fma_f32 %3, %3, %8, %3\n", "v_pk_fma_f32 %4, %4, %8, %4\n", "v_pk_fma_f32 %5, %5, %8, %5\n", "v_pk_fma_f32 %6, %6, %8, %6\n", "v_pk_fma_f32 %7, %7, %8, %7\n", "v");
        } else if constexpr (KIND == PK_ADDF_INDEP) { PK8("v_pk_add_f32 %0, %0, %8\n", "v_pk_add_f32 %1, %1, %8\n", "v_pk_add_f32 %2, %2, %8\n", "v_pk_add_f32 %3, %3, %8\n", "v_pk_add_f32 %4, %4, %8\n", "v_pk_add_f32 %5, %5, %8\n", "v_pk_add_f32 %6, %6, %8\n", "v_pk_add_f32 %7, %7, %8\n", "v");
        } else if constexpr (KIND == PK_MULF_INDEP) { PK8("v_pk_mul_f32 %0, %0, %8\n", "v_pk_mul_f32 %1, %1, %8\n", "v_pk_mul_f32 %2, %2, %8\n", "v_pk_mul_f32 %3, %3, %8\n", "v_pk_mul_f32 %4, %4, %8\n", "v_pk_mul_f32 %5, %5, %8\n", "v_pk_mul_f32 %6, %6, %8\n", "v_pk_mul_f32 %7, %7, %8\n", "v");
        } else if constexpr (KIND == PK_FMA_SGPR_INDEP) { PK8("v_pk_fma_f32 %0, %0, %8, %0\n", "v_pk_fma_f32 %1, %1, %8, %1\n", "v_pk_fma_f32 %2, %2, %8, %2\n", "v_pk_fma_f32 %3, %3, %8, %3\n", "v_pk_fma_f32 %4, %4, %8, %4\n", "v_pk_fma_f32 %5, %5, %8, %5\n", "v_pk_fma_f32 %6, %6, %8, %6\n", "v_pk_fma_f32 %7, %7, %8, %7\n", "s");
        } else if constexpr (KIND == MOV_B64_INDEP) { PK8("v_mov_b64 %0, %8\n", "v_mov_b64 %1, %8\n", "v_mov_b64 %2, %8\n", "v_mov_b64 %3, %8\n", "v_mov_b64 %4, %8\n", "v_mov_b64 %5, %8\n", "v_mov_b64 %6, %8\n", "v_mov_b64 %7, %8\n", "v");
        } else if constexpr (KIND == F_ADD_DEP) { asm volatile(R32("v_add_f32 %0, %0, %1\n") : "+v"(a0) : "v"(k));
        } else if constexpr (KIND == PK_FMA_DEP) { asm volatile(R32("v_pk_fma_f32 %0, %0, %1, %0\n") : "+v"(p0) : "v"(pk));
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    const uint64_t r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (uint32_t)sm ^ sc;
    out[blockIdx.x * blockDim.x + threadIdx.x] ^= (uint32_t)sm2 ^ (uint32_t)sm3 ^ (uint32_t)sm4;
    out[blockIdx.x * blockDim.x + threadIdx.x] ^= (uint32_t)((p0 ^ p1 ^ p2 ^ p3 ^ p4 ^ p5 ^ p6 ^ p7) >> 7);
    if ((threadIdx.x & 63u) == 0u) {
        uint64_t *c = cyc + 3 * (size_t)(blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64);
        c[0] = t1 - t0; c[1] = r0; c[2] = r1;
    }
}

__global__ void clock_kernel(uint64_t *out)
{
    const uint64_t c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    uint64_t r1 = r0;
    while (r1 - r0 < 200000ull) r1 = __builtin_amdgcn_s_memrealtime();     // 2 ms of the 100 MHz counter
    const uint64_t c1 = __builtin_amdgcn_s_memtime();
    out[0] = c1 - c0; out[1] = r1 - r0;
}

static int g_first = 0;

template <int KIND>
static void run_kind(int ncu, double mhz, uint32_t *d_out, uint64_t *d_cyc, std::string &json)
{
    const int iters = 40000;
    char buf[512];
    printf("%-66s", kKindName[KIND]);
    fflush(stdout);
    json += std::string("  {\"pattern\": \"") + kKindName[KIND] + "\", \"rows\": [";
    for (int w = 1; w <= 8; w++) {
        if (KIND >= OR_INDEP && w != 1 && w != 4 && w != 8) continue;
        const int blocks = ncu * w;                    // 256-thread workgroups: one wave per SIMD each
        std::vector<uint64_t> h((size_t)blocks * 4 * 3);
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        probe_kernel<KIND><<<blocks, 256>>>(d_out, d_cyc, 100, 12345u);        // warm-up
        CK(hipEventRecord(e0));
        probe_kernel<KIND><<<blocks, 256>>>(d_out, d_cyc, iters, 12345u);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost));
        double mean = 0, rsum = 0;
        uint64_t rmin = ~0ull, rmax = 0;
        const size_t nw = (size_t)blocks * 4;
        for (size_t i = 0; i < nw; i++) {
            mean += (double)h[3 * i];
            rsum += (double)(h[3 * i + 2] - h[3 * i + 1]);
            if (h[3 * i + 1] < rmin) rmin = h[3 * i + 1];
            if (h[3 * i + 2] > rmax) rmax = h[3 * i + 2];
        }
        const double load_mhz = mean / rsum * 100.0;                 // shader cycles per 100 MHz tick inside the loops
        const double resident = rsum / (double)(rmax - rmin) / (ncu * 4.0);      // mean waves per SIMD over the launch's span
        mean /= (double)nw;
        const double per_wave = mean / ((double)iters * kBlockInsts[KIND]);
        // instructions per shader cycle per SIMD over the span in which waves ran (device timestamps, clock as measured under load)
        const double chip = (double)nw * iters * kBlockInsts[KIND] / ((double)(rmax - rmin) / 100.0 * load_mhz) / (ncu * 4.0);
        (void)ms;
        printf(" %5.2f/%4.3f/%3.1f/%4.0f", per_wave, chip, resident, load_mhz);
        fflush(stdout);
        snprintf(buf, sizeof buf, "%s{\"waves_per_simd\": %d, \"cycles_per_inst_one_wave\": %.3f, \"insts_per_cycle_per_simd\": %.4f, \"valu_per_cycle_per_simd\": %.4f, \"resident_waves_per_simd\": %.2f, \"shader_mhz_under_load\": %.0f}",
                 w == 1 ? "" : ", ", w, per_wave, chip, chip * kBlockValu[KIND] / kBlockInsts[KIND], resident, load_mhz);
        json += buf;
        CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    }
    printf("\n");
    json += "]}";
}

template <int K>
static void run_all(int ncu, double mhz, uint32_t *d_out, uint64_t *d_cyc, std::string &json)
{
    if constexpr (K < NKINDS) {
        if (K >= g_first) {
            if (K > g_first) json += ",\n";
            run_kind<K>(ncu, mhz, d_out, d_cyc, json);
        }
        run_all<K + 1>(ncu, mhz, d_out, d_cyc, json);
    }
}

int main(int argc, char **argv)
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    uint32_t *d_out; uint64_t *d_cyc, *d_clk;
    CK(hipMalloc(&d_out, (size_t)ncu * 8 * 256 * 4));
    CK(hipMalloc(&d_cyc, (size_t)ncu * 8 * 4 * 8 * 3));
    CK(hipMalloc(&d_clk, 16));
    uint64_t clk[2];
    clock_kernel<<<1, 64>>>(d_clk);
    CK(hipMemcpy(clk, d_clk, 16, hipMemcpyDeviceToHost));
    clock_kernel<<<1, 64>>>(d_clk);
    CK(hipMemcpy(clk, d_clk, 16, hipMemcpyDeviceToHost));
    const double mhz = (double)clk[0] / (double)clk[1] * 100.0;
    printf("%s: %d CUs, s_memtime runs at %.0f MHz (vs the 100 MHz s_memrealtime), clockRate %d kHz\n", prop.name, ncu, mhz,
           prop.clockRate);
    printf("columns: launched waves per SIMD 1..8; cell = cycles per instruction as one wave sees them / instructions per cycle per SIMD "
           "(chip-wide, device timestamps) / mean resident waves per SIMD / shader MHz inside the loops\n");
    std::string json = "{\"device\": \"" + std::string(prop.name) + "\", \"cus\": " + std::to_string(ncu) +
                       ", \"shader_mhz\": " + std::to_string(mhz) + ", \"patterns\": [\n";
    if (argc > 2) g_first = atoi(argv[2]);
    run_all<0>(ncu, mhz, d_out, d_cyc, json);
    json += "\n]}\n";
    if (argc > 1) {
        FILE *f = fopen(argv[1], "w");
        if (f) { fputs(json.c_str(), f); fclose(f); }
    }
    return 0;
}
