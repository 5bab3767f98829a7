/*
 * picsong_hip.h -- C ABI of the MI355X-native PICSONG hot path
 *                  (level shift -> DWT 5/3 | 9/7 -> BPC-PaCo -> BitStreamBuilder, and inverse).
 *
 * The reference (13Karl/CUDA-Image-and-Video-codec) exposes no FFI; the seam this library replaces
 * is the C++ facade layer its engines call with device pointers and a stream.
 * Each entry point cites the reference interface it stands in for (paths relative to
 * CUDA_ImCod/).  Conventions:
 *   - plain C linkage, POD arguments, caller-owned device memory, no torch / C++ types;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *   - every function returns PICSONG_OK (0) or a negative error code; nothing calls exit()
 *     (the reference prints and exits, SupportFunctions/AuxiliarFunctions.cpp:39-56);
 *     picsong_last_error() returns a thread-local message for the last failure;
 *   - stage functions are asynchronous on `stream` unless their name ends in _sync or they
 *     return a host value (documented per function);
 *   - a context is thread-safe for concurrent use only with distinct workspaces/streams, as the
 *     reference's facades are (one set of scratch buffers per worker, CodingEngine.cu:157-197).
 */
#ifndef PICSONG_HIP_H
#define PICSONG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PICSONG_OK 0
#define PICSONG_ERR_ARG (-1)        /* invalid argument / validity limit of SURVEY A.9 */
#define PICSONG_ERR_HIP (-2)        /* a HIP runtime call failed */
#define PICSONG_ERR_IO (-3)         /* LUT folder / file problem */
#define PICSONG_ERR_NOMEM (-4)
#define PICSONG_ERR_RANGE (-5)      /* a codeblock exceeded the supported magnitude range */
#define PICSONG_ERR_NODEVICE (-6)   /* no usable GPU: the product has no CPU fallback */

#define PICSONG_CB 64               /* codeblock edge, BPC/BPCEngine.cuh:29-36 */
#define PICSONG_CB_WORDS 4096       /* staging ints per codeblock */
#define PICSONG_HDR_SHORTS 9        /* global header, BitStreamBuilder.cpp:54-93 */

/* Coding parameters == the flag globals of Launcher.cu:8-29 that reach the hot path. */
typedef struct picsong_params {
    int width, height;      /* -xSize / -ySize (unpadded) */
    int wl;                 /* -wl, 1..7 (header limit) */
    int cp;                 /* -cp: 2 coding passes, or 3 (deprecated in the reference; needs the cp_sig / cp_sign tables) */
    int lossy;              /* -type: 0 = 5/3 reversible, 1 = 9/7 + quantisation */
    float qs;               /* -qs */
    float k;                /* -k: 0 = two coding passes on every plane; > 0 = complexity-scalable bulk mode */
    int cb_width, cb_height;/* -cbWidth / -cbHeight: header-only */
    int bit_depth;          /* -bps (8) */
    int frames;             /* -frames (header only) */
    int components;         /* -components: 1 (grey) or 3 (with is_rgb) */
    int is_rgb;             /* -isRGB: planar R,G,B planes, RCT (lossless) / ICT (lossy) colour transform */
} picsong_params;

/* LUT geometry == header.txt (Engines/Engine.cu:190-210) + section sizes
 * (IO/IOManager.ipp:431-433). */
typedef struct picsong_lut_info {
    int n_bitplanes, n_subbands, ctx_ref, ctx_sign, ctx_sig, precision, n_files, n_bp_files;
    int n_ref, n_sig, n_sign;       /* section sizes in ints; table = [ref | sig | sign] */
    int n_tables;                   /* tables laid back to back: 1 (k = 0, file _0) or the bit-plane
                                     * files _0.._(n-1) of -k > 0 (Engines/Engine.cu:12-56); 0 == 1 */
    int cp;                         /* coding passes the table is laid out for: 2 (or 0) = [ref | sig | sign];
                                     * 3 = that followed by [cp_sig | cp_sign] (IO/IOManager.ipp:539-606) */
} picsong_lut_info;

typedef struct picsong_ctx picsong_ctx;

const char *picsong_last_error(void);
const char *picsong_version(void);

/* ---- geometry (SupportFunctions/AuxiliarFunctions.cpp:22-26, CodingEngine.cu:170-177) ---- */
int    picsong_pad_dim(int v);
size_t picsong_dwt_extra(int aw, int ah, int wl);
/* upper bound of one frame's codestream in shorts: 9 + 2 nCB + AW*AH + 1 */
size_t picsong_max_stream_shorts(int aw, int ah);

/* ---- header (BitStreamBuilder::setExtraInformation BitStreamBuilder.cpp:35-94 <->
 *      DecodingEngine::getExtraInformation Engines/DecodingEngine.cu:567-585), host only ---- */
int picsong_header_pack(const picsong_params *p, uint16_t out[PICSONG_HDR_SHORTS]);
int picsong_header_unpack(const uint16_t in[PICSONG_HDR_SHORTS], picsong_params *p);

/* ---- LUT text parser (IOManager::loadLUTHeaders IO/IOManager.ipp:363-386 +
 *      IOManager::loadLUTUpgraded :404-612), host only.  component 1/2/3 = R/G/B files,
 *      0 = un-suffixed.  `fill` = value of entries the reference never writes (de-facto 0,
 *      SURVEY fact 5).  table may be NULL to query info->n_* first. ---- */
int picsong_lut_load(const char *folder, int component, int wl, int fill,
                     picsong_lut_info *info, int32_t *table, size_t table_capacity);
/* -k > 0 (Engine::initLUT's multi-file branch, Engines/Engine.cu:12-56): the tables of files
 * _0 .. _(n_tables-1), table j at offset j * (n_ref + n_sig + n_sign); n_tables <= 0 = all
 * AMOUNT_OF_BITPLANE_FILES.  table needs n_tables * (n_ref + n_sig + n_sign) ints
 * (query with table == NULL: info->n_tables is filled in). */
int picsong_lut_load_k(const char *folder, int component, int wl, int fill, int n_tables,
                       picsong_lut_info *info, int32_t *table, size_t table_capacity);

/* -cp 3 (Engine::initLUT with codingPasses == 3, Engines/Engine.cu:56-100; loader IO/IOManager.ipp:539-606):
 * file _0 of ref, sig, sign, cp_sig and cp_sign; table needs n_ref + 2 (n_sig + n_sign) ints.  cp = 2 is
 * picsong_lut_load. */
int picsong_lut_load_cp(const char *folder, int component, int wl, int fill, int cp,
                        picsong_lut_info *info, int32_t *table, size_t table_capacity);

/* ---- context: replaces `new DWT<T,Y>(...)` / `new BPCCuda<T>(...)` + Engine::initLUT's
 *      cudaMalloc/cudaMemcpy of the table (Engines/Engine.cu:111-136).  Owns only the LUT copy
 *      and a small internal workspace (scan scratch, flags, the frame pipeline's buffers when
 *      picsong_encode_frame/picsong_decode_frame are used). ---- */
int  picsong_ctx_create(const picsong_params *p, int device, picsong_ctx **out);
void picsong_ctx_destroy(picsong_ctx *ctx);
int  picsong_ctx_set_lut(picsong_ctx *ctx, const picsong_lut_info *info, const int32_t *host_table);
int  picsong_ctx_padded_dims(const picsong_ctx *ctx, int *aw, int *ah, int *n_codeblocks);
/* Hint, no effect on results: on != 0 says that frames of OTHER contexts / streams are in flight on
 * this GPU while this context's frames run (the reference's -numberOfStreams > 1 video engine,
 * Engines/CodingEngine.cu:758-1069): throughput over latency.  What it selects today: the -k > 0 encoder's
 * instantiation (six waves a SIMD with compact table copies when hinted, the register-rich four-wave one for a
 * lone frame).  The k = 0 kernels take the same launches either way -- their waves ask for issue priority by
 * plane count instead, which serves a lone frame and costs frames in flight nothing.  Default: off. */
int  picsong_ctx_set_pipelined(picsong_ctx *ctx, int on);
/* RGB: component c (0,1,2) uses its own table, files {ref,sig,sign}{R,G,B}.txt_0
 * (Engine::initLUT Engines/Engine.cu:124-136: _LUTInformation[i]); picsong_ctx_set_lut == component 0 */
int  picsong_ctx_set_lut_component(picsong_ctx *ctx, int component, const picsong_lut_info *info,
                                   const int32_t *host_table);

/* The reference hands BPCCuda<T>::Code / Decode the DEVICE copy of the table with its geometry on every call
 * (`int* LUTInformation` + six integers, BPC/BPCEngine.hpp:16-17; Engine::initLUT made that copy,
 * Engines/Engine.cu:111-136).  This adopts such a caller-owned device table for component slot c without
 * copying it: it must stay valid while the context uses it.  info->n_ref / n_sig / n_sign == 0 are derived
 * from the geometry and the context's wl (IO/IOManager.ipp:431-433). */
int  picsong_ctx_set_lut_device(picsong_ctx *ctx, int component, const picsong_lut_info *info,
                                const int32_t *d_table);

/* ---- level shift: offsetImage<T> Engines/CodingEngine.cu:581-588 and
 *      removeOffsetAndApplyMaxMin(/Lossy) Engines/DecodingEngine.cu:706-729.
 *      T = int32 (lossless ctx) or float (lossy ctx); n = AW*AH. ---- */
int picsong_level_shift_fwd(picsong_ctx *ctx, const uint8_t *d_in, void *d_out, void *stream);
int picsong_level_shift_inv(picsong_ctx *ctx, void *d_data, void *stream);

/* ---- DWT: DWT<T,Y>::DWTEncode(T* in, T* out, stream) / DWTDecode(int* in, T* out, stream)
 *      (DWT/DWTGenerator.hpp:27-29, DWT/DWTGenerator.cu:1268-1424).  Same buffer contract:
 *      d_out has AW*AH + picsong_dwt_extra() elements; forward leaves the Mallat-layout
 *      coefficients (row stride AW) in d_out[0 .. AW*AH); inverse leaves the image at
 *      d_out + picsong_dwt_extra().  Quantisation / de-quantisation fused for lossy. ---- */
int picsong_dwt_forward(picsong_ctx *ctx, const void *d_in, void *d_out, void *stream);
int picsong_dwt_inverse(picsong_ctx *ctx, const int32_t *d_in, void *d_out, void *stream);
/* level-0 u8 ingest with the level shift fused (the reference's deprecated DWTEncodeChar,
 * DWT/DWTGenerator.cu:1141-1261, is the model); results identical to shift + forward. */
int picsong_dwt_forward_u8(picsong_ctx *ctx, const uint8_t *d_in, void *d_out, void *stream);

/* ---- BPC: BPCEngine<T>::kernelLauncher(CODE|DECODE) BPC/BPCEngine.cu:2307-2424 preceded by
 *      deviceMemoryAllocator's 0xFF memset (:2429-2441).  d_coeffs: Mallat T[AW*AH];
 *      d_staging: int32[AW*AH] (4096 per codeblock: [0] = MSB, [1..len) codewords);
 *      d_sizes: int32[nCB].  (The encoders themselves keep a 16-bit staging -- no staging word holds more
 *      than 16 bits -- in a buffer of the context; picsong_bpc_encode fills d_staging with 0xFF bytes and
 *      widens words [0, len) of every codeblock into it: the array a caller sees is the reference's.) ---- */
int picsong_bpc_encode(picsong_ctx *ctx, const void *d_coeffs, int32_t *d_staging,
                       int32_t *d_sizes, void *stream);
int picsong_bpc_decode(picsong_ctx *ctx, const int32_t *d_staging, const int32_t *d_sizes,
                       int32_t *d_coeffs, void *stream);

/* the same with the table of component slot 0..2 (RGB: the reference passes _LUTInformation[i],
 * Engines/CodingEngine.cu:617, Engines/DecodingEngine.cu:799-823) */
int picsong_bpc_encode_component(picsong_ctx *ctx, int component, const void *d_coeffs, int32_t *d_staging,
                                 int32_t *d_sizes, void *stream);
int picsong_bpc_decode_component(picsong_ctx *ctx, int component, const int32_t *d_staging,
                                 const int32_t *d_sizes, int32_t *d_coeffs, void *stream);

/* ---- BitStreamBuilder: createBitStream BitStreamBuilder.cpp:100-114 (CUB InclusiveSum +
 *      index LUT + buildBitStreamLUTBS BitStreamBuilder.cu:106-137,290-323) and createCodeStream
 *      BitStreamBuilder.cpp:134-153.  h_header NULL == iter != 0 (header shorts stay 0xFFFF).
 *      pack: *h_total (host) receives the stream length in shorts after an internal stream
 *      synchronisation, exactly like HTotalBSSize[0]; pass NULL to stay asynchronous and read
 *      the length later with picsong_last_total(). ---- */
int picsong_bitstream_pack(picsong_ctx *ctx, const int32_t *d_staging, const int32_t *d_sizes,
                           const uint16_t *h_header, uint16_t *d_stream, int *h_total,
                           void *stream);
int picsong_bitstream_unpack(picsong_ctx *ctx, const uint16_t *d_stream, int32_t *d_staging,
                             int32_t *d_sizes, void *stream);
/* synchronises `stream` and returns the total (shorts) of the most recent pack on this ctx */
int picsong_last_total(picsong_ctx *ctx, void *stream, int *h_total);

/* ---- whole frame: the grey call sequences of CodingEngine::runImage/runVideo
 *      (Engines/CodingEngine.cu:634-674,819-872) and DecodingEngine::runImage
 *      (Engines/DecodingEngine.cu:770-794).  d_frame: padded u8[AW*AH] (caller pads as
 *      IOManager::loadFrameCAdaptedSizes does, or uses picsong_pad_frame_host).  iter == 0
 *      writes the populated header.  Asynchronous; length via picsong_last_total().
 *      picsong_decode_frame reads the stream's own shorts and nothing beyond them (-cp 2: the coder
 *      takes its codewords from d_stream itself; -cp 3 through the staging, as
 *      picsong_bitstream_unpack does); lengths outside 1..4096 are clamped and raise the range flag.  A
 *      DAMAGED length table can still claim more codewords than the stream holds: reads then reach up to
 *      picsong_max_stream_shorts() shorts, so an untrusted stream belongs in a buffer of that size (as with
 *      picsong_bitstream_unpack).
 *      Alignment: none required of d_frame.  A 16-byte aligned frame (what an allocator returns) takes the vector
 *      kernels and the 16-bit coefficient form; any other pointer -- a view at an odd offset -- takes the per-column
 *      kernels with the 32-bit arrays: same codestream, slower. ---- */
int picsong_encode_frame(picsong_ctx *ctx, const uint8_t *d_frame, int iter, uint16_t *d_stream,
                         void *stream);
int picsong_decode_frame(picsong_ctx *ctx, const uint16_t *d_stream, uint8_t *d_frame_out,
                         void *stream);
/* ---- batched frames: n consecutive frames of a video through ONE launch per stage (grid.z = frame for the
 *      DWT levels, one coder grid over n x nCB codeblocks, one pack grid) -- CodingEngine::runVideo's call
 *      sequence (Engines/CodingEngine.cu:819-872) for frames first_iter .. first_iter + n - 1, which the
 *      reference spreads over -numberOfStreams worker threads (:990-1061).  Frame f is the padded
 *      u8[AW*AH] at d_frames + f * frame_stride (bytes, 16-byte aligned), its codestream lands at
 *      d_streams + f * stream_stride (shorts, >= picsong_max_stream_shorts); only the video's frame 0
 *      (first_iter + f == 0) carries the populated header.  n = 1..64; the context grows its workspace to
 *      the largest n seen (about 10 bytes per pixel per frame).  Asynchronous; picsong_last_totals
 *      synchronises `stream` and returns the n lengths in shorts.  Byte-identical to n calls of
 *      picsong_encode_frame.  Grey -cp 2 contexts, any -k (k > 0: the BULK coder instantiations over the
 *      n frames of the launch); -cp 3 is coded frame by frame. ---- */
int picsong_encode_frames(picsong_ctx *ctx, int n, const uint8_t *d_frames, size_t frame_stride, int first_iter,
                          uint16_t *d_streams, size_t stream_stride, void *stream);
int picsong_last_totals(picsong_ctx *ctx, void *stream, int n, int *h_totals);
/* The mirror for decoding: n codestreams (stream f at d_streams + f * stream_stride shorts) to n padded u8 frames (frame f
 * at d_frames_out + f * frame_stride bytes, 4-byte aligned strides) through one launch per stage -- DecodingEngine's
 * video loop (Engines/DecodingEngine.cu:734-1141) for n consecutive frames.  Byte-identical to n calls of
 * picsong_decode_frame; grey -cp 2 contexts, any -k. */
int picsong_decode_frames(picsong_ctx *ctx, int n, const uint16_t *d_streams, size_t stream_stride, uint8_t *d_frames_out,
                          size_t frame_stride, void *stream);
/* The same lengths without a wait: copies the totals of the most recent picsong_encode_frame (n = 1) or
 * picsong_encode_frames (n = its frame count) into d_totals (device, int32[n]) on `stream`.  A caller that keeps
 * many calls in flight (the frame-sharded multi-GPU exchange, bench.py --gpus N) collects them per bucket of
 * calls and reads them back once, instead of synchronising after every call. */
int picsong_copy_last_totals(picsong_ctx *ctx, void *stream, int n, int32_t *d_totals);

/* ---- RGB path: RGBTransformLossless / RGBTransformLossy with the level shift
 *      fused (Engines/CodingEngine.cu:357-403,408-449; Engines/DecodingEngine.cu:599-701), then each
 *      component is coded as a frame of its own with its own LUT (CodingEngine.cu:598-633,676-712).
 *      Planes are padded AW*AH arrays; T = int32 (lossless ctx) or float (lossy ctx).
 *      encode_plane = DWTEncode + Code for one component; with_header == the reference's iter == 0.
 *      decode_plane = Decode + DWTDecode: d_plane_out has AW*AH + picsong_dwt_extra() elements, the
 *      component lands at d_plane_out + extra (no clamp: the inverse colour transform clamps). ---- */
int picsong_rgb_forward(picsong_ctx *ctx, const uint8_t *d_r, const uint8_t *d_g, const uint8_t *d_b,
                        void *d_c0, void *d_c1, void *d_c2, void *stream);
int picsong_rgb_inverse(picsong_ctx *ctx, const void *d_c0, const void *d_c1, const void *d_c2,
                        uint8_t *d_r, uint8_t *d_g, uint8_t *d_b, void *stream);
int picsong_encode_plane(picsong_ctx *ctx, const void *d_plane, int component, int with_header,
                         uint16_t *d_stream, void *stream);
int picsong_decode_plane(picsong_ctx *ctx, const uint16_t *d_stream, int component, void *d_plane_out,
                         void *stream);
/* One RGB frame through ONE launch per stage -- the colour transform, then its three components as the three frames
 * of the batched grid: grid.z = 3 for the transform's levels, one coder grid in which component c codes with table c,
 * one scan + pack -- in place of picsong_rgb_forward + 3 x picsong_encode_plane (the reference codes the components
 * one after the other, Engines/CodingEngine.cu:598-633).  d_r / d_g / d_b: padded u8[AW*AH] planes; component c's
 * codestream lands at d_streams + c * stream_stride (shorts, >= picsong_max_stream_shorts); header_mask bit c = that
 * component carries the populated header (image: 1 -- component 0 only, iter = component; video frame 0: 7; else 0).
 * Lengths: picsong_last_totals(ctx, stream, 3, ..) / picsong_copy_last_totals.  Byte-identical to the plane-by-plane
 * calls; -cp 2 RGB contexts (any -k) whose three tables share one geometry.  The planes need 4-byte alignment; the
 * fused head (the colour transform, RCT or ICT, in the transform's load stage) runs when all three are 16-byte
 * aligned, the separate colour-transform kernel otherwise (same streams).  The decoder's mirror takes the three
 * codestreams (same stride) to the three padded u8 planes (Engines/DecodingEngine.cu:599-701, 736-769); its 5/3 form
 * runs the inverse colour transform inside the finest synthesis level. */
int picsong_encode_rgb_frame(picsong_ctx *ctx, const uint8_t *d_r, const uint8_t *d_g, const uint8_t *d_b, int header_mask,
                             uint16_t *d_streams, size_t stream_stride, void *stream);
int picsong_decode_rgb_frame(picsong_ctx *ctx, const uint16_t *d_streams, size_t stream_stride, uint8_t *d_r, uint8_t *d_g,
                             uint8_t *d_b, void *stream);

/* ---- intra-frame sharding (BASELINE config 5): codeblocks are independent
 *      (correctCBBorders zeroes outside neighbours, BPC/BPCEngine.cu:465-484), so a rank can code
 *      the stripe [cb_begin, cb_begin + cb_count) of the frame's raster-ordered codeblocks.  The
 *      stripe leaves as a self-describing mini-stream with the normal layout for cb_count blocks:
 *      9 x 0xFFFF | cb_count x (MSB, len) | payload | 0xFFFF.  The writer rank splices: header(9) +
 *      all pair tables in stripe order + all payloads in stripe order + 0xFFFF == the 1-GPU
 *      stream.  The DWT is computed for the whole frame on every rank (it needs all rows).
 *      Asynchronous; length via picsong_last_total(). ---- */
int picsong_encode_frame_stripe(picsong_ctx *ctx, const uint8_t *d_frame, int cb_begin, int cb_count,
                                uint16_t *d_stream, void *stream);
/* Row-band sharding of the transform for the same split ("partition level 0 by row bands
 * with a halo of 2 (5/3) or 4 (9/7) rows per side ... then all-gather LL1 and compute levels >= 1
 * redundantly"): no rank transforms or even holds the whole frame.
 *   picsong_dwt_forward_band: level 0 (u8 ingest, level shift fused) of the input rows [row0, row0 + rows)
 *     only -- row0 and rows even; d_frame is addressed in frame coordinates (row y at d_frame + y * AW) but
 *     only the band's rows and its halo need to be present.  Writes rows [row0/2, (row0+rows)/2) of HL, LH and
 *     HH into the Mallat array at d_out and of LL1 into the scratch behind it (d_out + AW*AH elements, row
 *     stride AW/2) -- the same places picsong_dwt_forward_u8 writes them (DWTEngine::DWTForward's buffer
 *     contract, DWT/DWTGenerator.cu:1268-1342).
 *   picsong_dwt_forward_tail: levels 1 .. wl-1 from the complete LL1 in that scratch (after the ranks have
 *     all-gathered their LL1 row bands in place).
 *   picsong_encode_stripe_coded: coder + pack of the codeblocks [cb_begin, cb_begin + cb_count) from a
 *     coefficient array (picsong_encode_frame_stripe without its transform); mini-stream as above.
 * With N ranks and AH a multiple of 128 N, rank k transforms input rows [k AH/N, (k+1) AH/N) and codes the
 * codeblock rows [k R, (k+1) R) and [AH/128 + k R, ...), R = AH / (128 N): exactly the coefficients its own
 * band and the shared tail produce.  Asynchronous. */
int picsong_dwt_forward_band(picsong_ctx *ctx, const uint8_t *d_frame, int row0, int rows, void *d_out,
                             void *stream);
int picsong_dwt_forward_tail(picsong_ctx *ctx, void *d_out, void *stream);
int picsong_encode_stripe_coded(picsong_ctx *ctx, const void *d_coeffs, int cb_begin, int cb_count,
                                uint16_t *d_stream, void *stream);
/* host helper: IOManager::loadFrameCAdaptedSizes' mirror padding (IO/IOManager.ipp:72-112).
 * PICSONG_ERR_ARG when aw - w > w or ah - h > h: the reference's loop is undefined there. */
int picsong_pad_frame_host(const uint8_t *in, int w, int h, uint8_t *out, int aw, int ah);

/* ---- measurement: per-stage durations of picsong_encode_frame, taken with HIP events recorded
 *      on the launch stream (the reference accumulates host chrono time around its BPC kernel,
 *      BPC/BPCEngine.cu:2318-2422, "BPC acum time").  profile_begin(capacity) arms a ring of event
 *      sets; every later encode_frame records into the next set without synchronising;
 *      profile_read synchronises the last set and returns, per recorded frame, 3 floats:
 *      {dwt_ms (all levels), bpc_ms (bpc_kernel), pack_ms (scan + pack)}.  capacity 0 disarms. ---- */
int picsong_profile_begin(picsong_ctx *ctx, int capacity);
int picsong_profile_read(picsong_ctx *ctx, int *n_frames, float *ms, int ms_capacity_frames);

/* ---- self-test of the hardware property the coder's slot reservation uses (one LDS atomic add per
 *      codeword; lanes of one instruction that hit one counter are served in ascending lane order, the
 *      order of arithmeticEncoder's __activemask reservation, BPC/BPCEngine.cu:380-393): 16 M random lane
 *      masks against the v_mbcnt ranks.  *mismatches must come back 0; a build with
 *      -DPICSONG_ENC_LDS_RESERVE=0 does not depend on it. ---- */
int picsong_selftest_lds_order(int device, int *mismatches);

/* ---- diagnostics: nonzero if, since the previous query (reading clears it), any codeblock of a
 *      bpc call on ctx had MSB > 15 (outside the LUT's 15 bit-planes), or if
 *      picsong_bitstream_unpack / picsong_decode_frame
 *      met a codeblock length outside 1..4096 (damaged stream: the length is clamped, so no access
 *      leaves the staging or 9 + 2n + 4095n + 1 shorts of the stream buffer); synchronises `stream`. ---- */
int picsong_range_flag(picsong_ctx *ctx, void *stream, int *h_flag);

#ifdef __cplusplus
}
#endif
#endif
