// PICSONG command-line tool for MI355X -- host side (C++) of the hot path, written over the C ABI
// of include/picsong_hip.h.  It keeps the reference's flags, defaults, validation, file formats and
// console vocabulary (Launcher.cu:8-29,36-163; IO/IOManager.ipp:72-112,176-231,267-344,615-620) so
// it is a drop-in for the greyscale and RGB (planar R,G,B; RCT / ICT) image / video encode + decode
// paths, including the complexity-scalable mode -k > 0.  The reference's -cp 3 (deprecated, LUT files
// not shipped) is not built
// and are refused with a message instead of being silently ignored.
//
// Pipeline (video): `-numberOfStreams N` HIP streams, each with its own picsong_ctx, pinned host
// frame buffer and device buffers; frame f runs on stream f mod N; while the GPU works on up to N
// frames the host thread reads and pads the next one.  No spin-wait flag arrays (the reference's
// _doubleBufferInput/_doubleBufferOutput, CodingEngine.cu:233-239): ordering comes from the
// streams, completion from picsong_last_total()'s stream synchronisation.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/picsong_hip.h"

namespace {

struct Options {
    std::string input, output, lut_folder;
    int cd = 2, x = 0, y = 0, cb_width = 64, cb_height = 18, wl = 5, cp = 2, endianess = 0, bps = 8;
    int signed_or_unsigned = 0, video = 0, frames = 0, avoid_size_check = 0, components = 1, streams = 2;
    int is_rgb = 0, type = 0, device = 0, lut_fill = 0;
    float qs = 1.0f, k = 0.0f;
    std::string metrics;
};

[[noreturn]] void die(const std::string &msg)
{
    std::cout << msg << std::endl;
    std::exit(-1);
}

#define CK(expr)                                                                        \
    do {                                                                                \
        int rc_ = (expr);                                                               \
        if (rc_ != PICSONG_OK) die(std::string(#expr) + " failed: " + picsong_last_error()); \
    } while (0)
#define HIPCK(expr)                                                                     \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess) die(std::string(#expr) + " failed: " + hipGetErrorString(e_)); \
    } while (0)

// flag lookup: the value is the token after the flag (IO/CommandLineParser.cpp:10-25)
struct Args {
    std::vector<std::string> t;
    Args(int argc, char **argv) { for (int i = 1; i < argc; i++) t.emplace_back(argv[i]); }
    bool has(const std::string &f) const { for (auto &s : t) if (s == f) return true; return false; }
    std::string get(const std::string &f) const
    {
        for (size_t i = 0; i + 1 < t.size(); i++) if (t[i] == f) return t[i + 1];
        return "";
    }
};

void help()
{
    std::cout <<
        "PICSONG (MI355X build) -- wavelet image / video codec, DWT + BPC-PaCo hot path on gfx950\n"
        " -cd 0|1            0 = code, 1 = decode\n"
        " -i <file> -o <file> input / output\n"
        " -xSize W -ySize H   frame dimensions (coding; optional when the input is a P5 PGM)\n"
        " -wl N               wavelet levels (1..7, default 5)\n"
        " -type 0|1           0 = lossless 5/3, 1 = lossy 9/7 (+ -qs in (0,1])\n"
        " -cp 2               coding passes (3 is deprecated in the reference and not built here)\n"
        " -cbWidth 64 -cbHeight 18  kept for header compatibility\n"
        " -video 0|1 -frames F  video mode (raw planar frames; output + <o>_SIZE sidecar)\n"
        " -LUTFolder <dir>    probability tables (header.txt, {ref,sig,sign}R.txt_0)\n"
        " -numberOfStreams N  frames in flight (default 2)\n"
        " -isRGB 1 -components 3  planar R,G,B planes per frame (RCT lossless / ICT lossy)\n"
        " -k 0..65.535        complexity-scalable factor (needs the bit-plane LUT files _0.._14)\n"
        " -device D           GPU index (default 0);  --metrics <file>  JSON stage timings\n"
        " --lut-fill V        value of LUT entries the loader never writes (default 0)\n";
}

template <typename T> void echo(const char *flag, const T &v)
{
    std::cout << "User entered " << flag << " command " << v << std::endl;
}

Options parse(const Args &a)
{
    Options o;
    auto geti = [&](const char *f, int &dst) { if (a.has(f)) { dst = std::stoi(a.get(f)); echo(f, dst); } };
    auto getf = [&](const char *f, float &dst) { if (a.has(f)) { dst = std::stof(a.get(f)); echo(f, dst); } };
    auto gets = [&](const char *f, std::string &dst) { if (a.has(f)) { dst = a.get(f); echo(f, dst); } };
    geti("-cd", o.cd);
    gets("-i", o.input); gets("-o", o.output); gets("-LUTFolder", o.lut_folder);
    geti("-numberOfStreams", o.streams); geti("-video", o.video);
    geti("-device", o.device); geti("--lut-fill", o.lut_fill); gets("--metrics", o.metrics);
    if (o.cd == 0) {
        geti("-xSize", o.x); geti("-ySize", o.y); geti("-cbWidth", o.cb_width); geti("-cbHeight", o.cb_height);
        geti("-wl", o.wl); geti("-cp", o.cp); geti("-endianess", o.endianess); geti("-bps", o.bps);
        geti("-signedOrUnsigned", o.signed_or_unsigned); geti("-frames", o.frames);
        geti("-avoidSizeCheck", o.avoid_size_check); geti("-components", o.components);
        geti("-isRGB", o.is_rgb); geti("-type", o.type); getf("-qs", o.qs); getf("-k", o.k);
    }
    return o;
}

// ---- file helpers -----------------------------------------------------------------------------
struct Pgm { bool is_pgm = false; int w = 0, h = 0; size_t offset = 0; };

Pgm sniff_pgm(const std::string &path)
{
    Pgm p;
    std::ifstream f(path, std::ios::binary);
    char magic[2] = { 0, 0 };
    f.read(magic, 2);
    if (!f || magic[0] != 'P' || magic[1] != '5') return p;
    int vals[3], n = 0;
    while (n < 3 && f) {
        int c = f.peek();
        if (c == '#') { std::string line; std::getline(f, line); continue; }
        if (isspace(c)) { f.get(); continue; }
        f >> vals[n++];
    }
    if (n < 3) return p;
    f.get();                                   // single whitespace after maxval
    p.is_pgm = true; p.w = vals[0]; p.h = vals[1]; p.offset = (size_t)f.tellg();
    return p;
}

bool read_frame(std::ifstream &f, size_t base, size_t frame, int w, int h, uint8_t *dst)
{
    f.clear();
    f.seekg((std::streamoff)(base + frame * (size_t)w * (size_t)h));
    f.read(reinterpret_cast<char *>(dst), (std::streamsize)((size_t)w * h));
    return (size_t)f.gcount() == (size_t)w * (size_t)h;
}

struct Worker {
    picsong_ctx *ctx = nullptr;
    hipStream_t stream = nullptr;
    uint8_t *h_in = nullptr, *d_in = nullptr;      // padded frame, pinned / device
    uint16_t *h_out = nullptr, *d_out = nullptr;   // codestream, pinned / device
    uint8_t *h_raw = nullptr;                      // unpadded frame (host)
    long frame = -1;
};

// component c uses the {ref,sig,sign}{R,G,B}.txt_0 files (Engine::initLUT Engines/Engine.cu:124-136);
// with k > 0 every bit-plane file _0 .. _(AMOUNT_OF_BITPLANE_FILES-1) (Engines/Engine.cu:12-56)
void load_lut(picsong_ctx *ctx, const Options &o, int wl, int components = 1, float k = 0.0f)
{
    if (o.lut_folder.empty()) die("Incorrect parameters. Please choose valid values. (-LUTFolder is required)");
    const int n_tables = k > 0.0f ? 0 : 1;
    for (int c = 0; c < components; c++) {
        picsong_lut_info info;
        CK(picsong_lut_load_k(o.lut_folder.c_str(), c + 1, wl, o.lut_fill, n_tables, &info, nullptr, 0));
        std::vector<int32_t> table(((size_t)info.n_ref + info.n_sig + info.n_sign) * (size_t)info.n_tables);
        CK(picsong_lut_load_k(o.lut_folder.c_str(), c + 1, wl, o.lut_fill, n_tables, &info, table.data(), table.size()));
        CK(picsong_ctx_set_lut_component(ctx, c, &info, table.data()));
    }
}

picsong_params make_params(const Options &o)
{
    picsong_params p;
    memset(&p, 0, sizeof p);
    p.width = o.x; p.height = o.y; p.wl = o.wl; p.cp = o.cp; p.lossy = o.type ? 1 : 0; p.qs = o.qs; p.k = o.k;
    p.cb_width = o.cb_width; p.cb_height = o.cb_height; p.bit_depth = o.bps; p.frames = o.frames;
    p.components = o.components;
    p.is_rgb = o.is_rgb ? 1 : 0;
    return p;
}

void write_metrics(const Options &o, const char *mode, long frames, double seconds, double dwt, double bpc,
                   double pack, long shorts)
{
    if (o.metrics.empty()) return;
    std::ofstream m(o.metrics, std::ios::trunc);
    m << "{\"mode\": \"" << mode << "\", \"frames\": " << frames << ", \"seconds\": " << seconds
      << ", \"mpixels_per_s\": " << (seconds > 0 ? (double)frames * o.x * o.y / seconds / 1e6 : 0.0)
      << ", \"dwt_ms\": " << dwt << ", \"bpc_ms\": " << bpc << ", \"pack_ms\": " << pack
      << ", \"stream_shorts\": " << shorts << "}\n";
}

// ---- RGB coding (CodingEngine::runImage / runVideo, RGB branches: CodingEngine.cu:598-633,676-712,
// 760-818,873-930): the input holds planar R, G, B planes per frame; RCT / ICT with the level shift
// fused, then every component is coded as a frame of its own with its own LUT and appended to <o>,
// its length to <o>_SIZE.  Header: image -> component 0 only (iter = component), video -> all three
// components of frame 0 (iter = frame).
int run_encode_rgb(const Options &o, size_t file_base, long nframes)
{
    HIPCK(hipSetDevice(o.device));
    const int aw = picsong_pad_dim(o.x), ah = picsong_pad_dim(o.y);
    const size_t P = (size_t)aw * ah, max_shorts = picsong_max_stream_shorts(aw, ah);
    picsong_params params = make_params(o);
    picsong_ctx *ctx = nullptr;
    CK(picsong_ctx_create(&params, o.device, &ctx));
    load_lut(ctx, o, o.wl, 3, o.k);
    hipStream_t s;
    HIPCK(hipStreamCreate(&s));
    uint8_t *h_in, *d_in[3];
    void *d_c[3];
    uint16_t *h_out, *d_out;
    HIPCK(hipHostMalloc(&h_in, P));
    for (int c = 0; c < 3; c++) { HIPCK(hipMalloc(&d_in[c], P)); HIPCK(hipMalloc(&d_c[c], P * 4)); }
    HIPCK(hipHostMalloc(&h_out, max_shorts * 2));
    HIPCK(hipMalloc(&d_out, max_shorts * 2));
    std::vector<uint8_t> raw((size_t)o.x * o.y);
    std::ifstream in(o.input, std::ios::binary);
    if (!in) die("Cannot open input file " + o.input);
    std::ofstream out(o.output, std::ios::binary | std::ios::app);
    std::ofstream sizes(o.output + "_SIZE", std::ios::binary | std::ios::app);
    long total_shorts = 0;
    auto t0 = std::chrono::steady_clock::now();
    for (long f = 0; f < nframes; f++) {
        for (int c = 0; c < 3; c++) {
            if (!read_frame(in, file_base, (size_t)(f * 3 + c), o.x, o.y, raw.data())) die("Input file is shorter than the requested frames.");
            CK(picsong_pad_frame_host(raw.data(), o.x, o.y, h_in, aw, ah));
            HIPCK(hipMemcpyAsync(d_in[c], h_in, P, hipMemcpyHostToDevice, s));
            HIPCK(hipStreamSynchronize(s));          // h_in is reused for the next plane
        }
        CK(picsong_rgb_forward(ctx, d_in[0], d_in[1], d_in[2], d_c[0], d_c[1], d_c[2], s));
        for (int c = 0; c < 3; c++) {
            const int with_header = o.video ? (f == 0) : (c == 0);
            CK(picsong_encode_plane(ctx, d_c[c], c, with_header, d_out, s));
            int total = 0;
            CK(picsong_last_total(ctx, s, &total));
            HIPCK(hipMemcpyAsync(h_out, d_out, (size_t)total * 2, hipMemcpyDeviceToHost, s));
            HIPCK(hipStreamSynchronize(s));
            out.write(reinterpret_cast<const char *>(h_out), (std::streamsize)total * 2);
            if (f == 0 && c == 0) sizes << total; else sizes << "," << total;
            total_shorts += total;
        }
    }
    double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::cout << "The time spent with the app without considering allocation periods is: " << sec << std::endl;
    write_metrics(o, "encode_rgb", nframes, sec, 0, 0, 0, total_shorts);
    picsong_ctx_destroy(ctx);
    (void)hipStreamDestroy(s);
    (void)hipHostFree(h_in); (void)hipHostFree(h_out); (void)hipFree(d_out);
    for (int c = 0; c < 3; c++) { (void)hipFree(d_in[c]); (void)hipFree(d_c[c]); }
    return 0;
}

// ---- coding engine: CodingEngine::runImage / runVideo call sequence ---------------------------
int run_encode(Options o)
{
    Pgm pgm = sniff_pgm(o.input);
    if (pgm.is_pgm) {
        if (o.x <= 0) o.x = pgm.w;
        if (o.y <= 0) o.y = pgm.h;
        if (o.x != pgm.w || o.y != pgm.h) die("Incorrect parameters. -xSize/-ySize differ from the PGM header.");
    }
    // Launcher.cu:132
    if (o.qs < 0 || o.qs > 1 || o.wl < 1 || o.x <= 0 || o.y <= 0 || o.wl > 10 || o.input.empty() || o.output.empty() ||
        o.cb_width % 64 != 0 || o.cb_height > 20 || o.cb_height < 18 || o.cp < 2 || o.cp > 3 || o.k < 0 || o.k > 65.535f)
        die("Incorrect parameters. Please choose valid values.");
    if (!((o.is_rgb && o.components == 3) || (!o.is_rgb && o.components == 1)))
        die("Incorrect parameters. Use -components 1, or -isRGB 1 -components 3 (planar R,G,B planes).");
    if (o.cp != 2) die("-cp 3 (deprecated in the reference) is not built in this MI355X hot-path build.");
    if (o.signed_or_unsigned != 0 || o.bps != 8) die("Only unsigned 8-bit samples are built in this MI355X hot-path build.");
    const long nframes = o.video ? o.frames : 1;
    if (nframes <= 0) die("Incorrect parameters. Please choose valid values. (-frames)");
    if (o.is_rgb) return run_encode_rgb(o, pgm.is_pgm ? pgm.offset : 0, nframes);
    const int nstreams = o.video ? (o.streams < 1 ? 1 : o.streams) : 1;

    HIPCK(hipSetDevice(o.device));
    const int aw = picsong_pad_dim(o.x), ah = picsong_pad_dim(o.y);
    const size_t P = (size_t)aw * ah, max_shorts = picsong_max_stream_shorts(aw, ah);
    picsong_params params = make_params(o);
    std::vector<Worker> w((size_t)nstreams);
    for (auto &k : w) {
        CK(picsong_ctx_create(&params, o.device, &k.ctx));
        load_lut(k.ctx, o, o.wl, 1, o.k);
        HIPCK(hipStreamCreate(&k.stream));
        HIPCK(hipHostMalloc(&k.h_in, P));
        HIPCK(hipMalloc(&k.d_in, P));
        HIPCK(hipHostMalloc(&k.h_out, max_shorts * 2));
        HIPCK(hipMalloc(&k.d_out, max_shorts * 2));
        k.h_raw = (uint8_t *)malloc((size_t)o.x * o.y);
        CK(picsong_profile_begin(k.ctx, (int)((nframes + nstreams - 1) / nstreams)));
    }
    std::ifstream in(o.input, std::ios::binary);
    if (!in) die("Cannot open input file " + o.input);
    // image: one truncating write (IOManager.ipp:615-620); video: append + _SIZE (:176-190)
    std::ofstream out(o.output, std::ios::binary | (o.video ? std::ios::app : std::ios::trunc));
    std::ofstream sizes;
    if (o.video) sizes.open(o.output + "_SIZE", std::ios::binary | std::ios::app);
    long total_shorts = 0;
    auto t0 = std::chrono::steady_clock::now();

    auto finish = [&](Worker &k) {
        if (k.frame < 0) return;
        int total = 0;
        CK(picsong_last_total(k.ctx, k.stream, &total));
        HIPCK(hipMemcpyAsync(k.h_out, k.d_out, (size_t)total * 2, hipMemcpyDeviceToHost, k.stream));
        HIPCK(hipStreamSynchronize(k.stream));
        out.write(reinterpret_cast<const char *>(k.h_out), (std::streamsize)total * 2);
        if (o.video) { if (k.frame == 0) sizes << total; else sizes << "," << total; }
        total_shorts += total;
        k.frame = -1;
    };
    for (long f = 0; f < nframes; f++) {
        Worker &k = w[(size_t)(f % nstreams)];
        finish(k);
        if (!read_frame(in, pgm.is_pgm ? pgm.offset : 0, (size_t)f, o.x, o.y, k.h_raw)) die("Input file is shorter than the requested frames.");
        CK(picsong_pad_frame_host(k.h_raw, o.x, o.y, k.h_in, aw, ah));
        HIPCK(hipMemcpyAsync(k.d_in, k.h_in, P, hipMemcpyHostToDevice, k.stream));
        CK(picsong_encode_frame(k.ctx, k.d_in, f == 0 ? 0 : 1, k.d_out, k.stream));
        k.frame = f;
    }
    // drain in frame order
    for (long f = nframes - nstreams < 0 ? 0 : nframes - nstreams; f < nframes; f++) finish(w[(size_t)(f % nstreams)]);
    double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

    double dwt = 0, bpc = 0, pack = 0;
    long counted = 0;
    for (auto &k : w) {
        int n = 0;
        std::vector<float> ms(3 * (size_t)((nframes + nstreams - 1) / nstreams) + 3);
        CK(picsong_profile_read(k.ctx, &n, ms.data(), (int)(ms.size() / 3)));
        for (int i = 0; i < n; i++) { dwt += ms[3 * i]; bpc += ms[3 * i + 1]; pack += ms[3 * i + 2]; counted++; }
    }
    std::cout << "The time spent with the app without considering allocation periods is: " << sec << std::endl;
    std::cout << "BPC acum time is: " << bpc / 1e3 << std::endl;
    write_metrics(o, "encode", nframes, sec, counted ? dwt / counted : 0, counted ? bpc / counted : 0,
                  counted ? pack / counted : 0, total_shorts);
    for (auto &k : w) {
        picsong_ctx_destroy(k.ctx);
        (void)hipStreamDestroy(k.stream);
        (void)hipHostFree(k.h_in); (void)hipFree(k.d_in); (void)hipHostFree(k.h_out); (void)hipFree(k.d_out);
        free(k.h_raw);
    }
    return 0;
}

// ---- decoding engine: DecodingEngine::runImage / runVideo call sequence -----------------------
void write_pgm(const std::string &path, const uint8_t *pix, int w, int h, int bit_depth)
{
    // IOManager::writeImage IO/IOManager.ipp:267-344: "P5\n<w> <h>\n<maxval>\n" + w*h bytes
    std::ofstream f(path, std::ios::binary | std::ios::trunc);
    f << "P5\n" << w << " " << h << "\n" << ((2 << (bit_depth - 1)) - 1) << "\n";
    f.write(reinterpret_cast<const char *>(pix), (std::streamsize)((size_t)w * h));
}

// ---- RGB decoding (DecodingEngine::runImage / runVideo RGB branches, Engines/DecodingEngine.cu:
// 736-769,868-940): three component streams per frame -> Decode + DWTDecode each -> inverse colour
// transform (offset + clamp fused) -> planar R, G, B planes of W*H bytes appended to <o>
// (IOManager::writeDecodedFrameUChar / writeDecodedFrameComponentUChar, IO/IOManager.ipp:236-262).
int run_decode_rgb(const Options &o, const picsong_params &p, picsong_ctx *ctx, std::ifstream &in,
                   const std::vector<long> &shorts, long nframes, int aw, int ah)
{
    const size_t P = (size_t)aw * ah, max_shorts = picsong_max_stream_shorts(aw, ah);
    const size_t extra = picsong_dwt_extra(aw, ah, p.wl);
    hipStream_t s;
    HIPCK(hipStreamCreate(&s));
    uint16_t *h_in, *d_in;
    uint8_t *h_pix, *d_pix[3];
    char *d_plane[3];
    HIPCK(hipHostMalloc(&h_in, max_shorts * 2));
    HIPCK(hipMalloc(&d_in, max_shorts * 2));
    HIPCK(hipHostMalloc(&h_pix, P));
    for (int c = 0; c < 3; c++) { HIPCK(hipMalloc(&d_pix[c], P)); HIPCK(hipMalloc(&d_plane[c], (P + extra) * 4)); }
    std::vector<uint8_t> crop((size_t)p.width * p.height);
    { std::ofstream trunc(o.output, std::ios::binary | std::ios::trunc); }
    auto t0 = std::chrono::steady_clock::now();
    size_t pos = 0;
    for (long f = 0; f < nframes; f++) {
        for (int c = 0; c < 3; c++) {
            const size_t n = (size_t)shorts[(size_t)(f * 3 + c)];
            if (n > max_shorts) die("Component codestream longer than the maximum for this geometry.");
            in.clear();
            in.seekg((std::streamoff)(pos * 2));
            in.read(reinterpret_cast<char *>(h_in), (std::streamsize)(n * 2));
            if ((size_t)in.gcount() != n * 2) die("Input file is shorter than its _SIZE sidecar says.");
            pos += n;
            HIPCK(hipMemcpyAsync(d_in, h_in, n * 2, hipMemcpyHostToDevice, s));
            CK(picsong_decode_plane(ctx, d_in, c, d_plane[c], s));
            HIPCK(hipStreamSynchronize(s));          // h_in / d_in are reused for the next component
        }
        CK(picsong_rgb_inverse(ctx, d_plane[0] + extra * 4, d_plane[1] + extra * 4, d_plane[2] + extra * 4, d_pix[0],
                               d_pix[1], d_pix[2], s));
        std::ofstream out(o.output, std::ios::binary | std::ios::app);
        for (int c = 0; c < 3; c++) {
            HIPCK(hipMemcpyAsync(h_pix, d_pix[c], P, hipMemcpyDeviceToHost, s));
            HIPCK(hipStreamSynchronize(s));
            for (int y = 0; y < p.height; y++) memcpy(&crop[(size_t)y * p.width], h_pix + (size_t)y * aw, (size_t)p.width);
            out.write(reinterpret_cast<const char *>(crop.data()), (std::streamsize)crop.size());
        }
    }
    double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::cout << "The time spent with the app without considering allocation periods and I/O is: " << sec << std::endl;
    picsong_ctx_destroy(ctx);
    (void)hipStreamDestroy(s);
    (void)hipHostFree(h_in); (void)hipFree(d_in); (void)hipHostFree(h_pix);
    for (int c = 0; c < 3; c++) { (void)hipFree(d_pix[c]); (void)hipFree(d_plane[c]); }
    return 0;
}

int run_decode(const Options &o)
{
    if (o.input.empty() || o.output.empty()) die("Incorrect parameters. Please choose valid values.");
    std::ifstream in(o.input, std::ios::binary);
    if (!in) die("Cannot open input file " + o.input);
    uint16_t hdr[PICSONG_HDR_SHORTS];
    in.read(reinterpret_cast<char *>(hdr), sizeof hdr);
    if ((size_t)in.gcount() != sizeof hdr) die("Input file too short for a PICSONG header.");
    picsong_params p;
    CK(picsong_header_unpack(hdr, &p));
    if (p.cp != 2 || !((p.components == 1 && !p.is_rgb) || (p.components == 3 && p.is_rgb)))
        die("This stream uses -cp 3 / a component layout not built here.");
    const long nframes = o.video ? p.frames : 1;
    std::vector<long> frame_shorts;
    if (o.video || p.is_rgb) {
        // IOManager::readBulkSizes IO/IOManager.ipp:196-208
        std::ifstream sz(o.input + "_SIZE");
        if (!sz) die("Cannot open " + o.input + "_SIZE");
        std::string tok;
        while (std::getline(sz, tok, ',')) if (!tok.empty()) frame_shorts.push_back(std::stol(tok));
        if ((long)frame_shorts.size() < nframes * p.components) die("_SIZE sidecar lists fewer streams than the header.");
    } else {
        in.seekg(0, std::ios::end);
        frame_shorts.push_back((long)((size_t)in.tellg() / 2));
    }
    HIPCK(hipSetDevice(o.device));
    picsong_ctx *ctx = nullptr;
    CK(picsong_ctx_create(&p, o.device, &ctx));
    Options lo = o;
    load_lut(ctx, lo, p.wl, p.components, p.k);
    int aw, ah, ncb;
    CK(picsong_ctx_padded_dims(ctx, &aw, &ah, &ncb));
    if (p.is_rgb) return run_decode_rgb(o, p, ctx, in, frame_shorts, nframes, aw, ah);
    const size_t P = (size_t)aw * ah, max_shorts = picsong_max_stream_shorts(aw, ah);
    hipStream_t s;
    HIPCK(hipStreamCreate(&s));
    uint16_t *h_in, *d_in;
    uint8_t *h_pix, *d_pix;
    HIPCK(hipHostMalloc(&h_in, max_shorts * 2));
    HIPCK(hipMalloc(&d_in, max_shorts * 2));
    HIPCK(hipHostMalloc(&h_pix, P));
    HIPCK(hipMalloc(&d_pix, P));
    std::vector<uint8_t> crop((size_t)p.width * p.height);
    if (o.video) { std::ofstream trunc(o.output, std::ios::binary | std::ios::app); }
    auto t0 = std::chrono::steady_clock::now();
    size_t pos = 0;
    for (long f = 0; f < nframes; f++) {
        const size_t n = (size_t)frame_shorts[(size_t)f];
        if (n > max_shorts) die("Frame codestream longer than the maximum for this geometry.");
        in.clear();
        in.seekg((std::streamoff)(pos * 2));
        in.read(reinterpret_cast<char *>(h_in), (std::streamsize)(n * 2));
        if ((size_t)in.gcount() != n * 2) die("Input file is shorter than its _SIZE sidecar says.");
        pos += n;
        HIPCK(hipMemcpyAsync(d_in, h_in, n * 2, hipMemcpyHostToDevice, s));
        CK(picsong_decode_frame(ctx, d_in, d_pix, s));
        HIPCK(hipMemcpyAsync(h_pix, d_pix, P, hipMemcpyDeviceToHost, s));
        HIPCK(hipStreamSynchronize(s));
        for (int y = 0; y < p.height; y++) memcpy(&crop[(size_t)y * p.width], h_pix + (size_t)y * aw, (size_t)p.width);
        if (o.video) {
            // IOManager::writeDecodedFrame IO/IOManager.ipp:214-231: raw W*H bytes appended
            std::ofstream out(o.output, std::ios::binary | std::ios::app);
            out.write(reinterpret_cast<const char *>(crop.data()), (std::streamsize)crop.size());
        } else {
            write_pgm(o.output, crop.data(), p.width, p.height, p.bit_depth);
        }
    }
    double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::cout << "The time spent with the app without considering allocation periods and I/O is: " << sec << std::endl;
    Options mo = o;
    mo.x = p.width; mo.y = p.height;
    write_metrics(mo, "decode", nframes, sec, 0, 0, 0, (long)pos);
    picsong_ctx_destroy(ctx);
    (void)hipStreamDestroy(s);
    (void)hipHostFree(h_in); (void)hipFree(d_in); (void)hipHostFree(h_pix); (void)hipFree(d_pix);
    return 0;
}

}  // namespace

int main(int argc, char **argv)
{
    auto start = std::chrono::steady_clock::now();
    Args a(argc, argv);
    if (a.has("-h") || argc == 1) { help(); return 0; }
    Options o = parse(a);
    int rc;
    if (o.cd == 0) rc = run_encode(o);
    else if (o.cd == 1) rc = run_decode(o);
    else die("Incorrect parameters. Please choose valid values.");
    double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();
    std::cout << "The time spent with the app is: " << sec << std::endl;
    return rc;
}
