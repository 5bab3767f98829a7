// pack_kernels.hpp -- BitStreamBuilder ("CR" compaction) for gfx950: per-codeblock lengths ->
// exclusive offsets -> dense uint16 codestream, and the inverse.
//
// Replaces cub::DeviceScan::InclusiveSum + binarySearchLUTBSB + buildBitStreamLUTBS /
// buildCodeStreamLUTBS (reference BitStreamBuilder/BitStreamBuilder.cu:33-171,198-229,290-323).
// The stream layout is the reference's; the method is not: the reference launches
// one thread per staged value and binary-searches the prefix array through a 256-entry index;
// here one workgroup owns one codeblock, so both sides of the copy are contiguous bursts and no
// search exists.  The scan is a single-workgroup wave scan (nCB <= 65,536).
// (Round 2 measured the scan folded away -- a workgroup per group of 1..16 codeblocks summing the lengths before
// its group itself, sixteen loads in flight, the group's payload copied as one run: bit-identical, one launch
// instead of two (three on the decoder's side), and SLOWER: 31 us against 8 + 14.5 for an 8K frame, 146 against
// 150 Gpixel/s with frames in flight.  Every workgroup then starts with the same chain of round trips -- lengths,
// wave reduction, LDS, barrier, group scan, barrier -- before it copies a byte, and a launch of 2040 such
// workgroups is one round of them; 8160 small workgroups behind a scan that costs one round trip overlap better.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace picsong {

struct HeaderArg { uint16_t h[9]; int has; };

// offsets[cb] = sum_{i<cb} (sizes[i] - 1); *total = 9 + 2n + sum(sizes - 1) + 1
// (BitStreamBuilder.cu:300-305).  One block of 1024 threads per frame (blockIdx.x = frame of a batched
// launch: sizes / offsets advance by n, total by 1).
// The workgroup is 256 threads for frames of up to 16384 codeblocks (scan_threads), one wave a SIMD, and the kernels keep
// to 32 registers (32-bit offsets from the scalar bases, eight loads in flight and no more): the launch sits between a frame's coder and its pack while other frames' coder waves -- seven to a
// SIMD at 72 registers -- hold all but 8 registers of every SIMD (a scan's wave starts when one of them ends), and a workgroup starts only when ONE CU has room for
// all of its waves.  (Measured with three calls in flight: the 1024-thread scan at 64 registers, four waves a SIMD that
// fit nowhere until coder waves have drained, cost 13 % of the step, 183.0 -> 159.3 Gpixel/s.)
__host__ __device__ inline unsigned scan_threads(int n) { return n <= 16384 ? 256u : 1024u; }
// (A thread's lengths are read eight at a time, all eight loads before the first use: one load, its wait, its add per
// trip is a round trip to the L2 per length -- 8 a thread at 8K, 64 at 16K -- in a launch that is nothing but latency.)
__device__ __forceinline__ int32_t scan_block(int32_t sum, int32_t *s_wave, int32_t &base_out)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int32_t inc = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int32_t o = __shfl_up(inc, d);
        if (lane >= d) inc += o;
    }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    int32_t base = 0;
    for (int w = 0; w < wave; w++) base += s_wave[w];
    base_out = base;
    return inc;                                             // inclusive within the wave; base = the waves before
}
// A thread's lengths b .. e - 1 go in groups of eight -- eight loads in flight from one 32-bit offset -- and what is left
// over (the array's last thread or two) one by one.
// offsets[i] = run, run += sizes[i] - 1
__device__ __forceinline__ void scan_write_offsets(const int32_t *sizes, int32_t *offsets, uint32_t b, uint32_t e, int32_t run)
{
    uint32_t g = b;
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
    for (; g + 8u <= e; g += 8u) {
        int32_t v[8], o[8];
        __builtin_memcpy(v, sizes + g, 32);
#pragma unroll
        for (uint32_t q = 0; q < 8u; q++) { o[q] = run; run += v[q] - 1; }
        __builtin_memcpy(offsets + g, o, 32);
    }
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
    for (; g < e; g++) { offsets[g] = run; run += sizes[g] - 1; }
}
__global__ __launch_bounds__(1024) void scan_sizes_kernel(const int32_t *sizes, int n, int32_t *offsets,
                                                          int32_t *total)
{
    __shared__ int32_t s_wave[16];
    sizes += (size_t)blockIdx.x * (size_t)n; offsets += (size_t)blockIdx.x * (size_t)n; total += blockIdx.x;
    const int tid = threadIdx.x;
    const uint32_t chunk = ((uint32_t)n + blockDim.x - 1u) / blockDim.x;
    const uint32_t b = (uint32_t)tid * chunk < (uint32_t)n ? (uint32_t)tid * chunk : (uint32_t)n, e = b + chunk < (uint32_t)n ? b + chunk : (uint32_t)n;
    int32_t sum = 0;
    uint32_t g = b;
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
    for (; g + 8u <= e; g += 8u) {
        int32_t v[8];
        __builtin_memcpy(v, sizes + g, 32);
#pragma unroll
        for (uint32_t q = 0; q < 8u; q++) sum += v[q] - 1;
    }
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
    for (; g < e; g++) sum += sizes[g] - 1;
    int32_t base;
    const int32_t inc = scan_block(sum, s_wave, base);
    scan_write_offsets(sizes, offsets, b, e, base + inc - sum);
    if (tid == (int)blockDim.x - 1) *total = 9 + 2 * n + (base + inc) + 1;
}

// one workgroup per codeblock (buildBitStreamLUTBS BitStreamBuilder.cu:106-137 layout); blockIdx.y = frame
// of a batched launch (staging advances by frame_words, sizes / offsets by n, total by 1, out by
// out_stride shorts; only the frame hdr.has - 1 == blockIdx.y carries the populated header -- has = 0: none;
// has < 0: -has is a bit mask of the frames that carry it (the components of an RGB frame))
// W = the staging's word: uint16_t, the encoders' own (BpcArgs::staging16: the frame paths), or int32_t, the reference's
// array as a caller of picsong_bitstream_pack holds it
template <typename W>
__global__ __launch_bounds__(256) void pack_kernel(const W *staging, const int32_t *sizes,
                                                   const int32_t *offsets, const int32_t *total, int n,
                                                   HeaderArg hdr, uint16_t *out, size_t frame_words = 0,
                                                   size_t out_stride = 0)
{
    const int cb = blockIdx.x, tid = threadIdx.x;
    {
        const size_t f = blockIdx.y;
        staging += f * frame_words; sizes += f * (size_t)n; offsets += f * (size_t)n; total += f; out += f * out_stride;
        hdr.has = hdr.has < 0 ? (int)(((unsigned)(-hdr.has) >> f) & 1u) : ((hdr.has != 0 && (size_t)(hdr.has - 1) == f) ? 1 : 0);
    }
    const W *st = staging + (size_t)cb * 4096u;
    const int len = sizes[cb];
    uint16_t *dst = out + 9 + 2 * (size_t)n + (size_t)offsets[cb];
    if constexpr (sizeof(W) == 2) {
        // 16-bit staging: the copy moves PAIRS of words -- a dword load from staging word 1 + 2 p, a dword store to
        // stream short 2 p (either may sit at an odd short: the target takes unaligned dword accesses) -- so a codeblock's
        // 4095 words are at most eight pairs a thread, all eight loads ahead of the stores; an odd count's last word is
        // stored as a short.
        const int nw = len - 1, np = (nw + 1) >> 1;
        uint32_t v[8];
        if (np > 0) {
#pragma unroll
            for (int q = 0; q < 8; q++) {
                // (no branch around a load: a thread without a pair re-reads the last one; the one pair that would
                // reach past the codeblock's 4096 words -- word 4095 of a raw block -- reads words 4094, 4095 instead)
                const int pr = tid + 256 * q, pc = pr < np ? pr : np - 1;
                const int w0 = 1 + 2 * pc, wl = w0 < 4094 ? w0 : 4094;
                __builtin_memcpy(&v[q], st + wl, 4);
                if (wl != w0) v[q] >>= 16;
            }
        }
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int pr = tid + 256 * q;
            if (pr < np) {
                if (2 * pr + 1 < nw) __builtin_memcpy(dst + 2 * pr, &v[q], 4);
                else dst[2 * pr] = (uint16_t)v[q];
            }
        }
    } else {
        // a codeblock is at most 4096 words: 16 a thread, in two groups of eight whose loads all go out before the first
        // store (one load, one wait, one store per trip left every wave parked on a round trip per element)
#pragma unroll
        for (int k = 0; k < 16; k += 8) {
            if (1 + 256 * k >= len) break;                  // (uniform: the whole workgroup is done)
            W v[8];
#pragma unroll
            for (int q = 0; q < 8; q++) { const int j = 1 + tid + 256 * (k + q); v[q] = j < len ? st[j] : (W)0; }
#pragma unroll
            for (int q = 0; q < 8; q++) { const int j = 1 + tid + 256 * (k + q); if (j < len) dst[j - 1] = (uint16_t)v[q]; }
        }
    }
    if (tid == 0) {
        out[9 + 2 * cb] = (uint16_t)st[0];
        out[9 + 2 * cb + 1] = (uint16_t)len;
    }
    if (cb == 0) {
        // deviceMemoryAllocator BitStreamBuilder.cu:270-279: 0xFFFF everywhere not written,
        // header copied only when iter == 0; one trailing short stays 0xFFFF
        if (tid < 9) out[tid] = hdr.has ? hdr.h[tid] : (uint16_t)0xFFFFu;
        if (tid == 9) out[*total - 1] = (uint16_t)0xFFFFu;
    }
}

// retrieveSizeArray BitStreamBuilder.cpp:119-129 on the device.  A length outside 1..4096 cannot
// come from the encoder; it is clamped (and flagged) so that a damaged stream can neither write
// outside a codeblock's staging nor read past 9 + 2n + 4095n + 1 shorts of the stream buffer.
// (blockIdx.y = frame of a batched launch: the stream advances by stream_stride shorts, sizes by n)
__global__ __launch_bounds__(256) void read_sizes_kernel(const uint16_t *stream, int n, int32_t *sizes, int *flag,
                                                         size_t stream_stride = 0)
{
    stream += (size_t)blockIdx.y * stream_stride; sizes += (size_t)blockIdx.y * (size_t)n;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        int v = stream[10 + 2 * i];
        if (v < 1 || v > 4096) { atomicOr(flag, 1); v = v < 1 ? 1 : 4096; }
        sizes[i] = v;
    }
}

// picsong_bpc_encode's last step: the encoders' 16-bit staging widened into the caller's int32 array (the reference's
// contract, BPCEngine.cu:2429-2441: 0xFFFFFFFF wherever nothing was written -- the caller's memset -- and words
// 0 .. len - 1 of every codeblock).  One workgroup per codeblock, as pack_kernel.
__global__ __launch_bounds__(256) void widen_staging_kernel(const uint16_t *staging16, const int32_t *sizes, int cb_base,
                                                            int32_t *staging)
{
    const int cb = cb_base + (int)blockIdx.x, tid = threadIdx.x;
    const uint16_t *src = staging16 + (size_t)cb * 4096u;
    int32_t *dst = staging + (size_t)cb * 4096u;
    const int len = sizes[cb];
#pragma unroll
    for (int k = 0; k < 16; k += 8) {
        if (256 * k >= len) break;
        uint16_t v[8];
#pragma unroll
        for (int q = 0; q < 8; q++) { const int j = tid + 256 * (k + q); v[q] = j < len ? src[j] : (uint16_t)0; }
#pragma unroll
        for (int q = 0; q < 8; q++) { const int j = tid + 256 * (k + q); if (j < len) dst[j] = (int32_t)v[q]; }
    }
}

// read_sizes_kernel + scan_sizes_kernel in one launch, for the decoder that reads its codewords from the stream itself
// (bpc_decode_kernel<false, NP, true>): the lengths as retrieveSizeArray reads them, clamped and flagged as above, and
// their scan.  One block of 1024 threads per frame (blockIdx.x: the stream advances by stream_stride shorts).
__global__ __launch_bounds__(1024) void scan_stream_kernel(const uint16_t *stream, int n, int32_t *sizes, int32_t *offsets,
                                                           int32_t *total, int *flag, size_t stream_stride)
{
    __shared__ int32_t s_wave[16];
    stream += (size_t)blockIdx.x * stream_stride;
    sizes += (size_t)blockIdx.x * (size_t)n; offsets += (size_t)blockIdx.x * (size_t)n; total += blockIdx.x;
    const int tid = threadIdx.x;
    const uint32_t chunk = ((uint32_t)n + blockDim.x - 1u) / blockDim.x;
    const uint32_t b = (uint32_t)tid * chunk < (uint32_t)n ? (uint32_t)tid * chunk : (uint32_t)n, e = b + chunk < (uint32_t)n ? b + chunk : (uint32_t)n;
    int32_t sum = 0;
    bool bad = false;
    uint32_t g = b;
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
    for (; g + 8u <= e; g += 8u) {
        // eight (MSB, length) pairs as they lie: 32 bytes from short 9 + 2 g, the lengths their upper halves
        uint32_t w[8];
        __builtin_memcpy(w, stream + 9u + 2u * g, 32);
#pragma unroll
        for (uint32_t q = 0; q < 8u; q++) {
            const uint32_t x = w[q] >> 16;
            w[q] = x < 1u ? 1u : (x > 4096u ? 4096u : x);
            bad = bad || w[q] != x;
            sum += (int32_t)w[q] - 1;
        }
        __builtin_memcpy(sizes + g, w, 32);
    }
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
    for (; g < e; g++) {
        int x = stream[10u + 2u * g];
        if (x < 1 || x > 4096) { bad = true; x = x < 1 ? 1 : 4096; }
        sizes[g] = x; sum += x - 1;
    }
    if (bad) atomicOr(flag, 1);
    int32_t base;
    const int32_t inc = scan_block(sum, s_wave, base);
    scan_write_offsets(sizes, offsets, b, e, base + inc - sum);
    if (tid == (int)blockDim.x - 1) *total = 9 + 2 * n + (base + inc) + 1;
}

// buildCodeStreamLUTBS BitStreamBuilder.cu:142-171 layout
// (blockIdx.y = frame of a batched launch: stream += stream_stride shorts, sizes / offsets += n, staging += frame_words)
__global__ __launch_bounds__(256) void unpack_kernel(const uint16_t *stream, const int32_t *sizes,
                                                     const int32_t *offsets, int n, int32_t *staging,
                                                     size_t stream_stride = 0, size_t frame_words = 0)
{
    {
        const size_t f = blockIdx.y;
        stream += f * stream_stride; sizes += f * (size_t)n; offsets += f * (size_t)n; staging += f * frame_words;
    }
    const int cb = blockIdx.x, tid = threadIdx.x;
    int32_t *st = staging + (size_t)cb * 4096u;
    const int len = sizes[cb];
    const uint16_t *src = stream + 9 + 2 * (size_t)n + (size_t)offsets[cb];
#pragma unroll
    for (int k = 0; k < 16; k += 8) {                       // (as pack_kernel: eight loads in flight a thread)
        if (1 + 256 * k >= len) break;
        uint16_t v[8];
#pragma unroll
        for (int q = 0; q < 8; q++) { const int j = 1 + tid + 256 * (k + q); v[q] = j < len ? src[j - 1] : (uint16_t)0; }
#pragma unroll
        for (int q = 0; q < 8; q++) { const int j = 1 + tid + 256 * (k + q); if (j < len) st[j] = (int32_t)v[q]; }
    }
    if (tid == 0) st[0] = (int32_t)stream[9 + 2 * cb];
}

}  // namespace picsong
