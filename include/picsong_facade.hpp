// picsong_facade.hpp -- the reference's three C++ facade classes re-expressed over the C ABI of
// picsong_hip.h, so that a code base written against
//     DWT<T,Y>::DWTEncode / DWTEncodeChar / DWTDecode        (reference DWT/DWTGenerator.hpp:22-29)
//     BPCCuda<T>::Code / Decode                               (reference BPC/BPCEngine.hpp:15-17)
// (the way Engines/CodingEngine.cu:634-674 and Engines/DecodingEngine.cu:770-794 are) can link
// libpicsong_hip.so instead of the CUDA translation units.  Same class names, constructor
// arguments and call sequence; the scratch arguments of Code / Decode that the MI355X kernels do
// not need (prefix arrays, CUB temp storage, binary-search LUT) are gone, and the LUT is handed over
// once (setLUT) instead of as seven geometry integers per call.  Error behaviour is the
// reference's: print the message and exit (SupportFunctions/AuxiliarFunctions.cpp:39-56).
//
// `ImageT` is anything with getWidth() / getHeight() / getBitDepth() / getComponents() -- the
// reference's Image class (Image/Image.hpp) qualifies unchanged.
//
// tests/facade_demo.cpp drives a whole encode + decode through these classes and checks it against
// picsong_encode_frame / the input (tests/test_cli.py::test_facade_classes_drive_the_library).
#pragma once
#include <hip/hip_runtime.h>

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

public:
    template <class ImageT>
    BPCCuda(ImageT *img, T *data, int wl, int cbW, int cbH, int cp, bool lossy, float qs, float k,
            int /*amountOfLUTFiles*/, int device = 0)
        : data_(data)
    {
        p_ = picsong_facade::params_of(img, lossy, wl, cbW, cbH, qs, cp, k);
        picsong_facade::check(picsong_ctx_create(&p_, device, &ctx_));
    }
    ~BPCCuda() { picsong_ctx_destroy(ctx_); }
    BPCCuda(const BPCCuda &) = delete;
    BPCCuda &operator=(const BPCCuda &) = delete;

    // Engine::initLUT's table (Engines/Engine.cu:101-141), once per object; component 0..2
    void setLUT(const picsong_lut_info &info, const int32_t *hostTable, int component = 0)
    {
        picsong_facade::check(picsong_ctx_set_lut_component(ctx_, component, &info, hostTable));
    }
    // == BPCCuda<T>::Code: staging fill + coder kernel + createBitStream.  hTotal[0] = shorts written.
    void Code(int *dStaging, int *dSizes, unsigned short *dBitstream, int *hTotal, int iter, hipStream_t s,
              int numberOfFrames, int component = 0)
    {
        (void)component;
        picsong_facade::check(picsong_bpc_encode(ctx_, data_, dStaging, dSizes, s));
        uint16_t hdr[PICSONG_HDR_SHORTS];
        picsong_params p = p_;
        p.frames = numberOfFrames;
        if (iter == 0) picsong_facade::check(picsong_header_pack(&p, hdr));
        picsong_facade::check(picsong_bitstream_pack(ctx_, dStaging, dSizes, iter == 0 ? hdr : nullptr, dBitstream,
                                                     hTotal, s));
    }
    // == BPCCuda<unsigned short>::Decode: createCodeStream + decoder kernel; `data` of the
    // constructor is the device bit-stream
    void Decode(int *dStaging, int *dSizes, int *dCoeffs, hipStream_t s)
    {
        picsong_facade::check(picsong_bitstream_unpack(ctx_, (const uint16_t *)data_, dStaging, dSizes, s));
        picsong_facade::check(picsong_bpc_decode(ctx_, dStaging, dSizes, dCoeffs, s));
        (void)hipStreamSynchronize(s);
    }
};
