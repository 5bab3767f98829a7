// TEST PROGRAM: drives one frame through the reference-shaped facade classes of
// include/picsong_facade.hpp (DWT<T,Y>, BPCCuda<T>) with call sites in the shape of
// Engines/CodingEngine.cu:661-662 and Engines/DecodingEngine.cu:773-779 (the reference's full argument
// lists, its member names), and checks the codestream against picsong_encode_frame and the
// reconstruction against the input.
//   usage: facade_demo W H wl lossy qs LUTFolder
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "picsong_facade.hpp"

struct int2_ { int x, y; };
struct float2_ { float x, y; };

// stands in for the reference's Image (Image/Image.hpp): the four getters the facades use
struct DemoImage {
    int w, h;
    int getWidth() const { return w; }
    int getHeight() const { return h; }
    int getBitDepth() const { return 8; }
    int getComponents() const { return 1; }
};

#define HIPCK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)

template <class T, class Y>
static int run(int W, int H, int wl, bool lossy, float qs, const char *lutdir)
{
    DemoImage img{ W, H };
    const int aw = picsong_pad_dim(W), ah = picsong_pad_dim(H), ncb = (aw / 64) * (ah / 64);
    const size_t P = (size_t)aw * ah, extra = picsong_dwt_extra(aw, ah, wl), max_shorts = picsong_max_stream_shorts(aw, ah);

    std::vector<uint8_t> frame((size_t)W * H), padded(P);
    uint32_t z = 12345u;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            z = 1664525u * z + 1013904223u;
            frame[(size_t)y * W + x] = (uint8_t)(((x * 255) / W + (y * 255) / H) / 2 + ((z >> 24) & 15));
        }
    picsong_facade::check(picsong_pad_frame_host(frame.data(), W, H, padded.data(), aw, ah));

    picsong_lut_info info;
    picsong_facade::check(picsong_lut_load(lutdir, 1, wl, 0, &info, nullptr, 0));
    std::vector<int32_t> table((size_t)info.n_ref + info.n_sig + info.n_sign);
    picsong_facade::check(picsong_lut_load(lutdir, 1, wl, 0, &info, table.data(), table.size()));

    hipStream_t s;
    HIPCK(hipStreamCreate(&s));
    uint8_t *d_u8; T *d_coef, *d_img; int *d_staging, *d_sizes, *d_coef_i; unsigned short *d_bits, *d_bits_ref;
    HIPCK(hipMalloc(&d_u8, P));
    HIPCK(hipMalloc(&d_coef, (P + extra) * sizeof(T)));
    HIPCK(hipMalloc(&d_img, (P + extra) * sizeof(T)));
    HIPCK(hipMalloc(&d_staging, P * sizeof(int)));
    HIPCK(hipMalloc(&d_sizes, (size_t)ncb * sizeof(int)));
    HIPCK(hipMalloc(&d_coef_i, P * sizeof(int)));
    HIPCK(hipMalloc(&d_bits, max_shorts * 2));
    HIPCK(hipMalloc(&d_bits_ref, max_shorts * 2));
    HIPCK(hipMemcpy(d_u8, padded.data(), P, hipMemcpyHostToDevice));

    // the members CodingEngine / DecodingEngine hold (Engines/Engine.cuh:60-160), under their names
    DemoImage *_frameStructure = &img;
    const int _waveletLevels = wl, _DWTCBWidth = 64, _DWTCBHeight = 18, _codingPasses = 2, _LUTAmountOfBitplaneFiles = 1;
    const bool _waveletType = lossy;
    const float _quantizationSize = qs, _k = 0.0f;
    const int _LUTNumberOfBitplanes = info.n_bitplanes, _LUTNumberOfSubbands = info.n_subbands,
              _LUTContextRefinement = info.ctx_ref, _LUTContextSign = info.ctx_sign,
              _LUTContextSignificance = info.ctx_sig, _LUTMultPrecision = info.precision, _HLUTBSTableSteps = 0,
              _numberOfFrames = 0;
    int *_LUTInformation[3] = { nullptr, nullptr, nullptr };      // Engine::initLUT: device copies of the tables
    HIPCK(hipMalloc(&_LUTInformation[0], table.size() * sizeof(int)));
    HIPCK(hipMemcpy(_LUTInformation[0], table.data(), table.size() * sizeof(int), hipMemcpyHostToDevice));
    T *_DWaveletCoefficients = d_coef;
    int *_DCodeStreamValues = d_staging, *_DSizeArray = d_sizes, *_DPrefixedArray = nullptr, *_DTempStoragePArray = nullptr,
        *_DLUTBSTable = nullptr;
    unsigned short _HExtraInformation[9], *_DBitStreamValues = d_bits;
    int _HTotalBSSize[1] = { 0 };
    double _measurementsBPC[1] = { 0.0 };
    const hipStream_t cudaStreamDefault = s;

    // ---- encode: the two statements of CodingEngine.cu:661-662, argument for argument
    int total = 0;
    {
        DWT<T, Y> *DWTGen = new DWT<T, Y>(_frameStructure, _waveletType, _waveletLevels, _DWTCBWidth, _DWTCBHeight, _quantizationSize);
        DWTGen->DWTEncodeChar(d_u8, _DWaveletCoefficients, cudaStreamDefault);
        BPCCuda<T>* BPC = new BPCCuda<T>(_frameStructure, _DWaveletCoefficients, _waveletLevels, _DWTCBWidth, _DWTCBHeight, _codingPasses, _waveletType, _quantizationSize, _k, _LUTAmountOfBitplaneFiles);
        BPC->Code(_LUTNumberOfBitplanes, _LUTNumberOfSubbands, _LUTContextRefinement, _LUTContextSign, _LUTContextSignificance, _LUTMultPrecision, _LUTInformation[0], _DCodeStreamValues, _DPrefixedArray, _DTempStoragePArray, _DSizeArray, _HExtraInformation, _DBitStreamValues, _HTotalBSSize, _DLUTBSTable, _HLUTBSTableSteps, 0, cudaStreamDefault, _numberOfFrames, &_measurementsBPC[0]);
        total = _HTotalBSSize[0];
        delete BPC;
        delete DWTGen;
        // the short form gives the same stream
        int total2 = 0;
        BPCCuda<T> bpc(&img, d_coef, wl, 64, 18, 2, lossy, qs, 0.0f, 1);
        bpc.setLUT(info, table.data());
        bpc.Code(d_staging, d_sizes, d_bits_ref, &total2, 0, s, 0);
        std::vector<unsigned short> a(total), b(total2);
        HIPCK(hipMemcpy(a.data(), d_bits, (size_t)total * 2, hipMemcpyDeviceToHost));
        HIPCK(hipMemcpy(b.data(), d_bits_ref, (size_t)total2 * 2, hipMemcpyDeviceToHost));
        if (total != total2 || std::memcmp(a.data(), b.data(), (size_t)total * 2) != 0) {
            std::printf("FACADE MISMATCH: the reference-signature Code and the short Code differ (%d / %d shorts)\n", total, total2);
            return 1;
        }
        uint16_t hdr[9];
        picsong_params hp = picsong_facade::params_of(&img, lossy, wl, 64, 18, qs);
        picsong_facade::check(picsong_header_pack(&hp, hdr));
        if (std::memcmp(hdr, _HExtraInformation, sizeof hdr) != 0) { std::printf("FACADE MISMATCH: HExtraInformation\n"); return 1; }
    }
    // ---- the same frame through the fused entry point
    int total_ref = 0;
    {
        picsong_params p = picsong_facade::params_of(&img, lossy, wl, 64, 18, qs);
        picsong_ctx *ctx;
        picsong_facade::check(picsong_ctx_create(&p, 0, &ctx));
        picsong_facade::check(picsong_ctx_set_lut(ctx, &info, table.data()));
        picsong_facade::check(picsong_encode_frame(ctx, d_u8, 0, d_bits_ref, s));
        picsong_facade::check(picsong_last_total(ctx, s, &total_ref));
        picsong_ctx_destroy(ctx);
    }
    std::vector<unsigned short> a(total), b(total_ref);
    HIPCK(hipMemcpy(a.data(), d_bits, (size_t)total * 2, hipMemcpyDeviceToHost));
    HIPCK(hipMemcpy(b.data(), d_bits_ref, (size_t)total_ref * 2, hipMemcpyDeviceToHost));
    if (total != total_ref || std::memcmp(a.data(), b.data(), (size_t)total * 2) != 0) {
        std::printf("FACADE MISMATCH: facade stream (%d shorts) != picsong_encode_frame (%d shorts)\n", total, total_ref);
        return 1;
    }

    // ---- decode: the statements of DecodingEngine.cu:773-779, argument for argument
    {
        std::vector<unsigned short> hostStream = a;                      // readCompressedImage's host copy
        unsigned short *_HBitStreamValues = hostStream.data();
        std::vector<int> hSizes((size_t)ncb);
        int *_HSizeArray = hSizes.data(), *_DWaveletCoefficientsI = d_coef_i;
        int _HBasicInformation[8] = { 0, _codingPasses, 0, 0, _waveletLevels, 0, 0, 0 };   // getExtraInformation
        T *_DImagePixels = d_img;
        BPCCuda<unsigned short>* BPC = new BPCCuda<unsigned short>(_frameStructure, _HBitStreamValues, _waveletLevels, _DWTCBWidth, _DWTCBHeight, _codingPasses, _waveletType, _quantizationSize, _k, _LUTAmountOfBitplaneFiles);
        BPC->Decode(aw * ah, _LUTNumberOfBitplanes, _LUTNumberOfSubbands, _LUTContextRefinement, _LUTContextSign, _LUTContextSignificance, _LUTMultPrecision, _LUTInformation[0], _DPrefixedArray, _DSizeArray, _HBasicInformation, _DTempStoragePArray, _DBitStreamValues, _DCodeStreamValues, _HSizeArray, _HTotalBSSize, _DWaveletCoefficientsI, cudaStreamDefault, _HLUTBSTableSteps, _DLUTBSTable, &_measurementsBPC[0]);
        DWT<T, Y>* DWTGen = new DWT<T, Y>(_frameStructure, _waveletType, _waveletLevels, _DWTCBWidth, _DWTCBHeight, _quantizationSize);
        DWTGen->DWTDecode(_DWaveletCoefficientsI, _DImagePixels, cudaStreamDefault);
        delete BPC;
        delete DWTGen;
        if (_HTotalBSSize[0] != total) { std::printf("FACADE MISMATCH: Decode's HTotalBSSize %d != %d\n", _HTotalBSSize[0], total); return 1; }
    }
    std::vector<T> rec(P);
    HIPCK(hipMemcpy(rec.data(), d_img + extra, P * sizeof(T), hipMemcpyDeviceToHost));
    double se = 0.0;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            // removeOffsetAndApplyMaxMin(/Lossy), Engines/DecodingEngine.cu:706-729, on the host
            double v = (double)rec[(size_t)y * aw + x] + 128.0;
            if (lossy) v = std::nearbyint(v + 0.01);
            v = v > 255 ? 255 : (v < 0 ? 0 : v);
            const double d = v - frame[(size_t)y * W + x];
            se += d * d;
        }
    const double mse = se / ((double)W * H);
    if (!lossy && mse != 0.0) { std::printf("FACADE MISMATCH: lossless reconstruction differs (mse %g)\n", mse); return 1; }
    if (lossy && 10.0 * std::log10(255.0 * 255.0 / (mse > 1e-12 ? mse : 1e-12)) < 35.0) {
        std::printf("FACADE MISMATCH: lossy PSNR too low (mse %g)\n", mse);
        return 1;
    }
    std::printf("FACADE OK %d shorts, mse %g\n", total, mse);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 7) { std::printf("usage: facade_demo W H wl lossy qs LUTFolder\n"); return 2; }
    const int W = std::atoi(argv[1]), H = std::atoi(argv[2]), wl = std::atoi(argv[3]);
    const bool lossy = std::atoi(argv[4]) != 0;
    const float qs = (float)std::atof(argv[5]);
    return lossy ? run<float, float2_>(W, H, wl, true, qs, argv[6]) : run<int, int2_>(W, H, wl, false, qs, argv[6]);
}
