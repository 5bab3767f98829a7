// picsong_facade.hpp -- the reference's three C++ facade classes re-expressed over the C ABI of
// picsong_hip.h, so that a code base written against
//     DWT<T,Y>::DWTEncode / DWTEncodeChar / DWTDecode        (reference DWT/DWTGenerator.hpp:22-29)
//     BPCCuda<T>::Code / Decode                               (reference BPC/BPCEngine.hpp:15-17)
// (the way Engines/CodingEngine.cu:634-674 and Engines/DecodingEngine.cu:770-794 are) can link
// libpicsong_hip.so instead of the CUDA translation units.  Same class names, constructor
// arguments and member signatures: BPCCuda<T>::Code / Decode take the reference's full argument lists
// (cudaStream_t spelled hipStream_t), so a call site such as Engines/CodingEngine.cu:661-662 or
// Engines/DecodingEngine.cu:774-775 compiles unchanged; the scratch arguments the MI355X kernels do not
// need (prefix array, CUB temp storage, binary-search LUT) are accepted and ignored.  Short overloads
// (LUT handed over once with setLUT) are kept for new code.  Error behaviour is the reference's: print
// the message and exit (SupportFunctions/AuxiliarFunctions.cpp:39-56).
//
// `ImageT` is anything with getWidth() / getHeight() / getBitDepth() / getComponents() -- the
// reference's Image class (Image/Image.hpp) qualifies unchanged.
//
// tests/facade_demo.cpp drives a whole encode + decode through these classes and checks it against
// picsong_encode_frame / the input (tests/test_cli.py::test_facade_classes_drive_the_library).
#pragma once
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "picsong_hip.h"

namespace picsong_facade {

inline void check(int rc)
{   // == GPU_HANDLE_ERROR: print + exit
    if (rc != PICSONG_OK) { std::puts(picsong_last_error()); std::exit(EXIT_FAILURE); }
}

template <class ImageT>
inline picsong_params params_of(ImageT *img, bool lossy, int wl, int cbW, int cbH, float qs, int cp = 2,
                                float k = 0.0f, int frames = 0)
{
    picsong_params p{};
    p.width = img->getWidth(); p.height = img->getHeight(); p.wl = wl; p.cp = cp; p.lossy = lossy ? 1 : 0;
    p.qs = qs; p.k = k; p.cb_width = cbW; p.cb_height = cbH; p.bit_depth = img->getBitDepth();
    p.frames = frames; p.components = img->getComponents(); p.is_rgb = p.components == 3 ? 1 : 0;
    return p;
}

}  // namespace picsong_facade

// DWT/DWTGenerator.hpp:22-29.  T = int (5/3) or float (9/7); Y is the reference's paired vector
// type (int2 / float2) and is unused here.
template <class T, class Y> class DWT {
    picsong_ctx *ctx_ = nullptr;

public:
    template <class ImageT> DWT(ImageT *img, bool lossy, int wl, int cbW, int cbH, float qs, int device = 0)
    {
        picsong_params p = picsong_facade::params_of(img, lossy, wl, cbW, cbH, qs);
        picsong_facade::check(picsong_ctx_create(&p, device, &ctx_));
    }
    ~DWT() { picsong_ctx_destroy(ctx_); }
    DWT(const DWT &) = delete;
    DWT &operator=(const DWT &) = delete;
    // the reference's calls return after the stream has drained
    void DWTEncode(T *dIn, T *dOut, hipStream_t s)
    {
        picsong_facade::check(picsong_dwt_forward(ctx_, dIn, dOut, s));
        (void)hipStreamSynchronize(s);
    }
    void DWTEncodeChar(unsigned char *dIn, T *dOut, hipStream_t s)
    {
        picsong_facade::check(picsong_dwt_forward_u8(ctx_, dIn, dOut, s));
        (void)hipStreamSynchronize(s);
    }
    void DWTDecode(int *dIn, T *dOut, hipStream_t s)
    {
        picsong_facade::check(picsong_dwt_inverse(ctx_, dIn, dOut, s));
        (void)hipStreamSynchronize(s);
    }
};

// BPC/BPCEngine.hpp:15-17 + BitStreamBuilder (createBitStream / createCodeStream run inside
// Code / Decode in the reference, BPC/BPCEngine.ipp:25-58).  T = int / float for coding (the type of
// the coefficient array), unsigned short for decoding (the type of the bit-stream), as in the
// reference's instantiations.
template <class T> class BPCCuda {
    picsong_ctx *ctx_ = nullptr;
    T *data_;
    picsong_params p_;
    int lut_files_;

    picsong_lut_info geometry(int nBp, int nSub, int cRef, int cSign, int cSig, int prec) const
    {
        picsong_lut_info li{};
        li.n_bitplanes = nBp; li.n_subbands = nSub; li.ctx_ref = cRef; li.ctx_sign = cSign; li.ctx_sig = cSig;
        li.precision = prec; li.n_files = 3; li.n_bp_files = lut_files_;
        li.n_tables = p_.k > 0.0f ? lut_files_ : 1;         // Engine::initLUT, Engines/Engine.cu:12-56
        return li;                                            // section sizes: derived by the library
    }

public:
    template <class ImageT>
    BPCCuda(ImageT *img, T *data, int wl, int cbW, int cbH, int cp, bool lossy, float qs, float k,
            int amountOfLUTFiles, int device = 0)
        : data_(data), lut_files_(amountOfLUTFiles)
    {
        p_ = picsong_facade::params_of(img, lossy, wl, cbW, cbH, qs, cp, k);
        picsong_facade::check(picsong_ctx_create(&p_, device, &ctx_));
    }
    ~BPCCuda() { picsong_ctx_destroy(ctx_); }
    BPCCuda(const BPCCuda &) = delete;
    BPCCuda &operator=(const BPCCuda &) = delete;

    // ---- the reference's member functions, argument for argument (BPC/BPCEngine.hpp:16-17) ----------
    // LUTInformation: the DEVICE table of the component being coded (the engines pass _LUTInformation[i]);
    // DCodeStreamValues: int staging; DSizeArray: per-codeblock lengths; HExtraInformation: receives the 9
    // header shorts when iter == 0; HTotalBSSize[0]: stream length in shorts; measurementsBPC[0]: seconds
    // spent in the coder kernel, accumulated (BPCEngine.cu:2318-2422).  DPrefixedArray, DTempStoragePArray,
    // DLUTBSTable and HLUTBSTableSteps belong to the reference's scan / search and are not used.
    void Code(int LUTNumberOfBitplanes, int LUTNumberOfSubbands, int LUTContextRefinement, int LUTContextSign,
              int LUTContextSignificance, int LUTMultPrecision, int *LUTInformation, int *DCodeStreamValues,
              int * /*DPrefixedArray*/, int * /*DTempStoragePArray*/, int *DSizeArray,
              unsigned short *HExtraInformation, unsigned short *DBitStreamValues, int *HTotalBSSize,
              int * /*DLUTBSTable*/, int /*HLUTBSTableSteps*/, int iter, hipStream_t mainStream, int numberOfFrames,
              double *measurementsBPC)
    {
        const picsong_lut_info li = geometry(LUTNumberOfBitplanes, LUTNumberOfSubbands, LUTContextRefinement,
                                             LUTContextSign, LUTContextSignificance, LUTMultPrecision);
        picsong_facade::check(picsong_ctx_set_lut_device(ctx_, 0, &li, LUTInformation));
        const auto t0 = std::chrono::steady_clock::now();
        picsong_facade::check(picsong_bpc_encode(ctx_, data_, DCodeStreamValues, DSizeArray, mainStream));
        (void)hipStreamSynchronize(mainStream);
        if (measurementsBPC)
            measurementsBPC[0] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        uint16_t hdr[PICSONG_HDR_SHORTS];
        if (iter == 0) {
            picsong_params p = p_;
            p.frames = numberOfFrames;
            picsong_facade::check(picsong_header_pack(&p, hdr));
            if (HExtraInformation)
                for (int i = 0; i < PICSONG_HDR_SHORTS; i++) HExtraInformation[i] = hdr[i];
        }
        picsong_facade::check(picsong_bitstream_pack(ctx_, DCodeStreamValues, DSizeArray, iter == 0 ? hdr : nullptr,
                                                     DBitStreamValues, HTotalBSSize, mainStream));
    }
    // `data` of the constructor is the HOST copy of the bit-stream (retrieveSizeArray walks it,
    // BitStreamBuilder.cpp:119-129): HSizeArray, when given, is filled from it the same way; the device
    // copy DBitStreamValues is what the kernels read.  HBasicInformation[1] / [4] (coding passes, wavelet
    // levels) must agree with the constructor's, as they do in DecodingEngine.cu:774.
    void Decode(int /*size*/, int LUTNumberOfBitplanes_, int LUTNumberOfSubbands_, int LUTContextRefinement_,
                int LUTContextSign_, int LUTContextSignificance_, int LUTMultPrecision_, int *LUTInformation_,
                int * /*DPrefixedArray*/, int *DSizeArray, int *HBasicInformation, int * /*DTempStoragePArray*/,
                unsigned short *DBitStreamValues, int *DCodeStreamValues, int *HSizeArray, int *HTotalBSSize,
                int *DWaveletCoefficients, hipStream_t mainStream, int /*HLUTBSTableSteps*/, int * /*DLUTBSTable*/,
                double *measurementsBPC)
    {
        if (HBasicInformation && (HBasicInformation[1] != p_.cp || HBasicInformation[4] != p_.wl)) {
            std::puts("BPCCuda::Decode: HBasicInformation disagrees with the constructor's coding passes / wavelet levels");
            std::exit(EXIT_FAILURE);
        }
        const picsong_lut_info li = geometry(LUTNumberOfBitplanes_, LUTNumberOfSubbands_, LUTContextRefinement_,
                                             LUTContextSign_, LUTContextSignificance_, LUTMultPrecision_);
        picsong_facade::check(picsong_ctx_set_lut_device(ctx_, 0, &li, LUTInformation_));
        int aw = 0, ah = 0, ncb = 0;
        picsong_facade::check(picsong_ctx_padded_dims(ctx_, &aw, &ah, &ncb));
        const unsigned short *host = reinterpret_cast<const unsigned short *>(data_);
        if (HSizeArray && host) {
            long sum = 0;
            for (int i = 0; i < ncb; i++) { HSizeArray[i] = host[PICSONG_HDR_SHORTS + 1 + 2 * i]; sum += HSizeArray[i] - 1; }
            if (HTotalBSSize) HTotalBSSize[0] = (int)(PICSONG_HDR_SHORTS + 2 * (long)ncb + sum + 1);
        }
        picsong_facade::check(picsong_bitstream_unpack(ctx_, DBitStreamValues, DCodeStreamValues, DSizeArray, mainStream));
        const auto t0 = std::chrono::steady_clock::now();
        picsong_facade::check(picsong_bpc_decode(ctx_, DCodeStreamValues, DSizeArray, DWaveletCoefficients, mainStream));
        (void)hipStreamSynchronize(mainStream);
        if (measurementsBPC)
            measurementsBPC[0] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }

    // ---- short forms for new code: the table is handed over once per object ---------------------------
    // Engine::initLUT's table (Engines/Engine.cu:101-141) from the host; component slot 0..2
    void setLUT(const picsong_lut_info &info, const int32_t *hostTable, int component = 0)
    {
        picsong_facade::check(picsong_ctx_set_lut_component(ctx_, component, &info, hostTable));
    }
    // staging fill + coder kernel + createBitStream with component's table.  hTotal[0] = shorts written.
    void Code(int *dStaging, int *dSizes, unsigned short *dBitstream, int *hTotal, int iter, hipStream_t s,
              int numberOfFrames, int component = 0)
    {
        picsong_facade::check(picsong_bpc_encode_component(ctx_, component, data_, dStaging, dSizes, s));
        uint16_t hdr[PICSONG_HDR_SHORTS];
        picsong_params p = p_;
        p.frames = numberOfFrames;
        if (iter == 0) picsong_facade::check(picsong_header_pack(&p, hdr));
        picsong_facade::check(picsong_bitstream_pack(ctx_, dStaging, dSizes, iter == 0 ? hdr : nullptr, dBitstream,
                                                     hTotal, s));
    }
    // createCodeStream + decoder kernel; `data` of the constructor is the DEVICE bit-stream here
    void Decode(int *dStaging, int *dSizes, int *dCoeffs, hipStream_t s, int component = 0)
    {
        picsong_facade::check(picsong_bitstream_unpack(ctx_, (const uint16_t *)data_, dStaging, dSizes, s));
        picsong_facade::check(picsong_bpc_decode_component(ctx_, component, dStaging, dSizes, dCoeffs, s));
        (void)hipStreamSynchronize(s);
    }
};
