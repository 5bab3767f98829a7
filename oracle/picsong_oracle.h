/*
 * picsong_oracle.h -- CPU restatement of the PICSONG DWT -> BPC -> BitStreamBuilder hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or call it, and there only
 * as the checker / timed CPU baseline.  The product (cuda-image-and-video-codec_amd/) never
 * includes, links or calls this code.
 *
 * PARITY UNPINNED against the CUDA binary: the reference is 100 % CUDA device code (no CPU path),
 * cannot be compiled here (no nvcc) and ships no tests / golden vectors.  This file restates the
 * algorithm from the reference source text; every function cites the file:line it follows
 * (paths relative to /root/reference/CUDA_ImCod/).  It is pinned only by (i) the shipped LUT
 * tables, (ii) constants in the source, (iii) encode->decode identity, (iv) the survey-time
 * known-answer vectors of SURVEY.md Appendix A.10 (independent model of the same reading).
 */
#ifndef PICSONG_ORACLE_H
#define PICSONG_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PO_CB 64            /* codeblock edge (BPC/BPCEngine.cuh:29-36)          */
#define PO_CB_WORDS 4096    /* staging ints per codeblock                        */
#define PO_HDR_SHORTS 9     /* global header shorts (BitStreamBuilder.cpp:54-93) */

/* LUT geometry + table (Engines/Engine.cu:101-141, IO/IOManager.ipp:363-612). */
typedef struct po_lut {
    int n_bitplanes;   /* LUT_N_BITPLANES           */
    int n_subbands;    /* LUT_N_SUBBANDS            */
    int ctx_ref;       /* N_CONTEXT_REFINEMENT      */
    int ctx_sign;      /* N_CONTEXT_SIGN            */
    int ctx_sig;       /* N_CONTEXT_SIGNIFICANCE    */
    int precision;     /* MULT_PRECISION            */
    int n_files;       /* LUT_N_FILES               */
    int n_bp_files;    /* AMOUNT_OF_BITPLANE_FILES (capped 32) */
    int wl;            /* wavelet levels the table was laid out for */
    int n_ref, n_sig, n_sign;  /* section sizes in ints */
    int32_t *table;    /* n_tables x [ref | sig | sign], malloc'd */
    int n_tables;      /* 1 for k = 0; the bit-plane files _0.._(n-1) for k > 0 (Engine.cu:12-56) */
    int cp;            /* coding passes the table was loaded for: 2 = [ref | sig | sign]; 3 = the same followed by
                        * [cp_sig | cp_sign], the cleanup pass's tables (IO/IOManager.ipp:539-606); 0 == 2 */
} po_lut;

/* Header fields (BitStreamBuilder.cpp:35-94 <-> Engines/DecodingEngine.cu:567-585). */
typedef struct po_header {
    uint32_t n_samples;   /* W*H*components */
    int cp;               /* 2 or 3 */
    int cb_height;        /* 7 bits */
    int cb_width;         /* 7 bits */
    int wl;               /* 0..7 (header can hold 4 bits: b15 of [2] + b0-1 of [3]) */
    int bit_depth;        /* 7 bits */
    int lossy;            /* 0/1 */
    int qs_1e4;           /* floor(qs*10000), 14 bits */
    int components;       /* 14 bits */
    int is_rgb;
    int height;           /* 16 bits */
    int endianess;
    int bps;              /* 5 bits */
    int is_signed;
    int frames;           /* 17 bits */
    int k_1e3;            /* floor(k*1000) */
} po_header;

/* ---- threading of the CPU-baseline leg (OpenMP over codeblocks / DWT rows+columns); default 1 */
void po_set_threads(int n);
/* statistics of the encoder's lock-step scan since the last reset: call sites with a coding lane, coding lanes, sites
 * that start a codeword (tools/lockstep_stats.py) */
void po_stats_reset(void);
void po_stats_get(unsigned long long *out3);
int  po_get_threads(void);
int  po_max_threads(void);

/* ---- geometry / ingest (IO/IOManager.ipp:72-112, SupportFunctions/AuxiliarFunctions.cpp:22-26) */
int  po_pad_dim(int v);
int po_pad_frame(const uint8_t *in, int W, int H, uint8_t *out, int AW, int AH);   /* -1: 2W < AW or 2H < AH */
void po_crop_frame_u8(const uint8_t *in, int AW, int AH, uint8_t *out, int W, int H);
/* integer-only synthetic frame generator, SURVEY.md 8(d) */
void po_gen_frame(uint8_t *out, int W, int H, uint32_t frame, uint32_t seed);

/* ---- level shift (Engines/CodingEngine.cu:581-588, Engines/DecodingEngine.cu:706-729) */
void po_level_shift_fwd_i32(const uint8_t *in, int32_t *out, size_t n, int bit_depth);
void po_level_shift_fwd_f32(const uint8_t *in, float *out, size_t n, int bit_depth);
void po_level_shift_inv_i32(int32_t *data, size_t n, int bit_depth);
void po_level_shift_inv_f32(float *data, size_t n, int bit_depth);

/* ---- colour transforms of the RGB path (Engines/CodingEngine.cu:357-403,
 *      Engines/DecodingEngine.cu:599-650), level shift fused as in the reference */
void po_rct_forward(const uint8_t *r, const uint8_t *g, const uint8_t *b, int32_t *c0, int32_t *c1,
                    int32_t *c2, size_t n, int bit_depth);
void po_rct_inverse(const int32_t *c0, const int32_t *c1, const int32_t *c2, uint8_t *r, uint8_t *g,
                    uint8_t *b, size_t n, int bit_depth);
void po_ict_forward(const uint8_t *r, const uint8_t *g, const uint8_t *b, float *c0, float *c1, float *c2,
                    size_t n, int bit_depth);
void po_ict_inverse(const float *c0, const float *c1, const float *c2, uint8_t *r, uint8_t *g, uint8_t *b,
                    size_t n, int bit_depth);

/* ---- DWT (DWT/DWTGenerator.cu) ; out buffers hold P + po_dwt_extra() elements */
size_t po_dwt_extra(int AW, int AH, int wl);
void po_dwt53_forward(const int32_t *in, int32_t *out, int AW, int AH, int wl);
void po_dwt53_inverse(const int32_t *in, int32_t *out, int AW, int AH, int wl);
void po_dwt97_forward(const float *in, float *out, int AW, int AH, int wl, float qs);
void po_dwt97_inverse(const int32_t *in, float *out, int AW, int AH, int wl, float qs);

/* ---- LUT */
/* component: 0 = "ref.txt_0" naming, 1/2/3 = R/G/B naming (IOManager.ipp:438-449).
 * fill: value for table entries the reference loader never writes; the
 * reference leaves them uninitialised, de-facto 0.  Returns 0 on success. */
int  po_lut_load(const char *folder, int component, int wl, int fill, po_lut *out);
/* k > 0: tables of files _0 .. _(n_tables-1) back to back (n_tables <= 0: all of them) */
int  po_lut_load_k(const char *folder, int component, int wl, int fill, int n_tables, po_lut *out);
/* -cp 3: file _0 of ref / sig / sign / cp_sig / cp_sign (Engine::initLUT's codingPasses == 3 sizes,
 * Engines/Engine.cu:66-68; loader IO/IOManager.ipp:539-606).  The BPC functions below take the three
 * coding passes (Encode3CP / Decode3CP, BPC/BPCEngine.cu:1727-1776,1844-1900) whenever lut->cp == 3. */
int  po_lut_load_cp(const char *folder, int component, int wl, int fill, int cp, po_lut *out);
/* consecutiveBitplanes of a codeblock, BPC/BPCEngine.cu:1684-1692 */
int  po_consecutive_bitplanes(int msb, float k, int level, int sb, int wl);
void po_lut_free(po_lut *lut);

/* ---- BPC, 2 coding passes, k = 0 (BPC/BPCEngine.cu) */
void po_find_subband(int x, int y, int AW, int AH, int wl, int *level, int *sb);
/* is_float: coefficients are float (truncated toward zero on load, BPCEngine.cu:49) */
void po_bpc_encode(const void *coeffs, int is_float, int AW, int AH, int wl, const po_lut *lut,
                   int32_t *staging /* AW*AH, caller memsets to -1 or not: function does it */,
                   int32_t *sizes /* nCB */);
void po_bpc_decode(const int32_t *staging, const int32_t *sizes, int AW, int AH, int wl,
                   const po_lut *lut, int32_t *coeffs /* AW*AH Mallat */);
/* complexity-scalable mode -k > 0 (encodeBulkMode / decodeBulkMode BPC/BPCEngine.cu:1285-1662,
 * Encode :1684-1716, Decode :1794-1835): planes below consecutiveBitplanes are coded in one
 * row-major bulk scan; lut must come from po_lut_load_k.  k = 0 == the functions above. */
void po_bpc_encode_k(const void *coeffs, int is_float, int AW, int AH, int wl, const po_lut *lut, float k,
                     int32_t *staging, int32_t *sizes);
void po_bpc_decode_k(const int32_t *staging, const int32_t *sizes, int AW, int AH, int wl,
                     const po_lut *lut, float k, int32_t *coeffs);
/* single-codeblock entry used by the known-answer tests: all 32 lanes use (level, sb) */
int  po_bpc_encode_block_uniform(const int32_t *block64x64, int level, int sb, int wl,
                                 const po_lut *lut, int32_t *staging4096);

/* ---- BitStreamBuilder (BitStreamBuilder/BitStreamBuilder.{cpp,cu}) */
void po_header_pack(const po_header *h, uint16_t out[PO_HDR_SHORTS]);
void po_header_unpack(const uint16_t in[PO_HDR_SHORTS], po_header *h);
/* returns total shorts written; header may be NULL (iter != 0: 9 x 0xFFFF) */
size_t po_bitstream_pack(const int32_t *staging, const int32_t *sizes, int n_cb,
                         const uint16_t *header, uint16_t *out);
size_t po_bitstream_total(const int32_t *sizes, int n_cb);
void po_bitstream_unpack(const uint16_t *in, int n_cb, int32_t *staging /* n_cb*4096 */,
                         int32_t *sizes);

/* ---- whole-frame convenience (call sequence of Engines/CodingEngine.cu:634-674 and
 *      Engines/DecodingEngine.cu:770-794).  out must hold 9 + 2 nCB + AW*AH + 1 shorts. */
size_t po_encode_frame(const uint8_t *frame, int W, int H, int wl, int lossy, float qs,
                       const po_lut *lut, int iter, int frames, uint16_t *out);
int po_decode_frame(const uint16_t *stream, int W, int H, int wl, int lossy, float qs,
                    const po_lut *lut, uint8_t *frame_out);
size_t po_encode_frame_k(const uint8_t *frame, int W, int H, int wl, int lossy, float qs, float k,
                         const po_lut *lut, int iter, int frames, uint16_t *out);
int po_decode_frame_k(const uint16_t *stream, int W, int H, int wl, int lossy, float qs, float k,
                      const po_lut *lut, uint8_t *frame_out);

#ifdef __cplusplus
}
#endif
#endif
