// TEST PROGRAM: drives one frame through the reference-shaped facade classes of
// include/picsong_facade.hpp (DWT<T,Y>, BPCCuda<T>) exactly as Engines/CodingEngine.cu:634-674 and
// Engines/DecodingEngine.cu:770-794 call them, and checks the codestream against
// picsong_encode_frame and the reconstruction against the input.
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

    // ---- encode, the reference's call sequence (CodingEngine.cu:651-667)
    int total = 0;
    {
        DWT<T, Y> dwt(&img, lossy, wl, 64, 18, qs);
        dwt.DWTEncodeChar(d_u8, d_coef, s);
        BPCCuda<T> bpc(&img, d_coef, wl, 64, 18, 2, lossy, qs, 0.0f, 1);
        bpc.setLUT(info, table.data());
        bpc.Code(d_staging, d_sizes, d_bits, &total, 0, s, 0);
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

    // ---- decode, the reference's call sequence (DecodingEngine.cu:774-784)
    {
        BPCCuda<unsigned short> bpc(&img, d_bits, wl, 64, 18, 2, lossy, qs, 0.0f, 1);
        bpc.setLUT(info, table.data());
        bpc.Decode(d_staging, d_sizes, d_coef_i, s);
        DWT<T, Y> dwt(&img, lossy, wl, 64, 18, qs);
        dwt.DWTDecode(d_coef_i, d_img, s);
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
