// PICSONG command-line tool for MI355X -- host side (C++) of the hot path, written over the C ABI
// of include/picsong_hip.h.  It keeps the reference's flags, defaults, validation, file formats and
// console vocabulary (Launcher.cu:8-29,36-163; IO/IOManager.ipp:72-112,176-231,267-344,615-620) so
// it is a drop-in for the greyscale and RGB (planar R,G,B; RCT / ICT) image / video encode + decode
// paths, including the complexity-scalable mode -k > 0 and the three-coding-pass mode -cp 3 (deprecated in
// the reference, whose tree ships no cp_sig / cp_sign tables for it: -LUTFolder must hold them).
//
// Pipeline (video encode): the reference's reader / worker / writer structure
// (CodingEngine.cu:212-262,463-498,758-1069) with condition variables instead of its spin-wait flag
// arrays (_doubleBufferInput/_doubleBufferOutput, :233-239).  A ring of slots, each with its own
// picsong_ctx, HIP stream, pinned host buffers and device buffers; frame f lives in slot f mod S
// (S = max(-numberOfStreams, 3) + 3).  Two reader threads fill the pinned input buffers straight from
// the file (pread), the main thread launches H2D copy + encode on the slot's stream, a collector
// thread waits for the streams in frame order, copies the codestream back and fixes its file offset,
// two writer threads pwrite the payloads; the _SIZE entries are appended in frame order at the end.
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "../../include/picsong_hip.h"

namespace {

struct Options {
    std::string input, output, lut_folder;
    int cd = 2, x = 0, y = 0, cb_width = 64, cb_height = 18, wl = 5, cp = 2, endianess = 0, bps = 8;
    int signed_or_unsigned = 0, video = 0, frames = 0, avoid_size_check = 0, components = 1, streams = 2;
    int is_rgb = 0, type = 0, device = 0, lut_fill = 0;
    int gpus = 1, frames_per_launch = 0;        // -gpus N: frames sharded over N devices; -framesPerLaunch B (0 = by frame size)
    std::string devices;                        // --devices a,b,..: explicit device list (a device may repeat)
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
        " -cp 2|3             coding passes (3: deprecated in the reference; needs cp_sig / cp_sign tables in -LUTFolder)\n"
        " -cbWidth 64 -cbHeight 18  kept for header compatibility\n"
        " -video 0|1 -frames F  video mode (raw planar frames; output + <o>_SIZE sidecar)\n"
        " -LUTFolder <dir>    probability tables (header.txt, {ref,sig,sign}R.txt_0)\n"
        " -numberOfStreams N  frames in flight (default 2)\n"
        " -isRGB 1 -components 3  planar R,G,B planes per frame (RCT lossless / ICT lossy)\n"
        " -k 0..65.535        complexity-scalable factor (needs the bit-plane LUT files _0.._14)\n"
        " -device D           GPU index (default 0);  --metrics <file>  JSON stage timings\n"
        " -gpus N             video coding: groups of frames sharded round-robin over devices D .. D+N-1 of this node\n"
        "                     (--devices a,b,c names them explicitly); every device copies its codestreams to the\n"
        "                     writer's pinned ring over its own host link, the output file is the 1-GPU file\n"
        " -framesPerLaunch B  frames coded / decoded per launch (default: 4 up to 4K frames, 1 above)\n"
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
    geti("-gpus", o.gpus); geti("-framesPerLaunch", o.frames_per_launch); gets("--devices", o.devices);
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
    int device = 0;                                // the GPU this slot's context, stream and buffers live on
    uint8_t *h_in = nullptr, *d_in = nullptr;      // padded frame, pinned / device
    uint16_t *h_out = nullptr, *d_out = nullptr;   // codestream, pinned / device
    uint8_t *h_raw = nullptr;                      // unpadded frame (host)
    long frame = -1;
};

// component c uses the {ref,sig,sign}{R,G,B}.txt_0 files (Engine::initLUT Engines/Engine.cu:124-136);
// with k > 0 every bit-plane file _0 .. _(AMOUNT_OF_BITPLANE_FILES-1) (Engines/Engine.cu:12-56)
// -cp 3: the five sections ref, sig, sign, cp_sig, cp_sign of file _0 (IO/IOManager.ipp:539-606)
void load_lut(picsong_ctx *ctx, const Options &o, int wl, int components = 1, float k = 0.0f, int cp = 2)
{
    if (o.lut_folder.empty()) die("Incorrect parameters. Please choose valid values. (-LUTFolder is required)");
    const int n_tables = k > 0.0f ? 0 : 1;
    for (int c = 0; c < components; c++) {
        picsong_lut_info info;
        if (cp == 3) {
            CK(picsong_lut_load_cp(o.lut_folder.c_str(), c + 1, wl, o.lut_fill, 3, &info, nullptr, 0));
            std::vector<int32_t> table((size_t)info.n_ref + 2 * ((size_t)info.n_sig + info.n_sign));
            CK(picsong_lut_load_cp(o.lut_folder.c_str(), c + 1, wl, o.lut_fill, 3, &info, table.data(), table.size()));
            CK(picsong_ctx_set_lut_component(ctx, c, &info, table.data()));
            continue;
        }
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
    load_lut(ctx, o, o.wl, 3, o.k, o.cp);
    hipStream_t s;
    HIPCK(hipStreamCreate(&s));
    uint8_t *h_in, *d_in[3];
    void *d_c[3];
    uint16_t *h_out, *d_out;
    HIPCK(hipHostMalloc(&h_in, P));
    for (int c = 0; c < 3; c++) { HIPCK(hipMalloc(&d_in[c], P)); HIPCK(hipMalloc(&d_c[c], P * 4)); }
    // the three components of a frame through ONE launch per stage (picsong_encode_rgb_frame) wherever the library
    // offers it: -cp 2, k = 0; otherwise plane by plane
    const bool batched = o.cp != 3;
    HIPCK(hipHostMalloc(&h_out, max_shorts * 2));
    HIPCK(hipMalloc(&d_out, max_shorts * 2 * (batched ? 3 : 1)));
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
        int totals[3] = { 0, 0, 0 };
        if (batched) {
            CK(picsong_encode_rgb_frame(ctx, d_in[0], d_in[1], d_in[2], o.video ? (f == 0 ? 7 : 0) : 1, d_out, max_shorts, s));
            CK(picsong_last_totals(ctx, s, 3, totals));
        } else {
            CK(picsong_rgb_forward(ctx, d_in[0], d_in[1], d_in[2], d_c[0], d_c[1], d_c[2], s));
        }
        for (int c = 0; c < 3; c++) {
            int total = totals[c];
            if (!batched) {
                const int with_header = o.video ? (f == 0) : (c == 0);
                CK(picsong_encode_plane(ctx, d_c[c], c, with_header, d_out, s));
                CK(picsong_last_total(ctx, s, &total));
            }
            HIPCK(hipMemcpyAsync(h_out, d_out + (batched ? (size_t)c * max_shorts : 0), (size_t)total * 2, hipMemcpyDeviceToHost, s));
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

constexpr long kProfFrames = 256;     // frames per pipeline slot whose stages are timed with HIP events

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
    if (o.cp == 3 && o.k > 0) die("Incorrect parameters. -cp 3 has no complexity-scalable mode (-k must be 0).");
    if (o.signed_or_unsigned != 0 || o.bps != 8) die("Only unsigned 8-bit samples are built in this MI355X hot-path build.");
    const long nframes = o.video ? o.frames : 1;
    if (nframes <= 0) die("Incorrect parameters. Please choose valid values. (-frames)");
    if (o.is_rgb) return run_encode_rgb(o, pgm.is_pgm ? pgm.offset : 0, nframes);
    // ---- devices: -gpus N takes devices D .. D+N-1, --devices names them (a device may repeat: several
    // worker sets on one GPU, which is also how the sharded path is exercised on a one-GPU box)
    std::vector<int> devs;
    {
        int ndev = 0;
        HIPCK(hipGetDeviceCount(&ndev));
        if (!o.devices.empty()) {
            std::stringstream ss(o.devices);
            std::string tok;
            while (std::getline(ss, tok, ',')) if (!tok.empty()) devs.push_back(std::stoi(tok));
        } else {
            for (int i = 0; i < (o.gpus < 1 ? 1 : o.gpus); i++) devs.push_back(o.device + i);
        }
        if (devs.empty()) die("Incorrect parameters. Please choose valid values. (--devices)");
        for (int d : devs)
            if (d < 0 || d >= ndev) die("Incorrect parameters. -gpus / --devices name GPU " + std::to_string(d) + ", this node has " + std::to_string(ndev));
        if (!o.video && devs.size() > 1) devs.resize(1);           // one image = one frame = one GPU
    }
    const int ndevs = (int)devs.size();
    const int aw = picsong_pad_dim(o.x), ah = picsong_pad_dim(o.y);
    const size_t P = (size_t)aw * ah, max_shorts = picsong_max_stream_shorts(aw, ah);
    // frames per launch: a 4K frame is 1020 coder waves, one per SIMD; four of them fill the GPU like an 8K frame
    int B = o.frames_per_launch > 0 ? o.frames_per_launch : (P <= (size_t)3840 * 2176 ? 4 : 1);
    if (!o.video || o.cp == 3) B = 1;
    if (B > 16) B = 16;
    if ((long)B > nframes) B = (int)nframes;
    const long ngroups = (nframes + B - 1) / B;
    // slots of the pipeline: at least 6 for a video (2 readers | launch | run | collect | 2 writers overlap),
    // a multiple of the device count so that slot i always belongs to device i mod ndevs
    int nstreams = o.video ? ((o.streams < 3 ? 3 : o.streams) + 3) : 1;
    if (o.video) nstreams = (nstreams + ndevs - 1) / ndevs * ndevs;
    if (nstreams < ndevs) nstreams = ndevs;

    const size_t frame_bytes = (size_t)o.x * o.y, file_base = pgm.is_pgm ? pgm.offset : 0;
    const bool padded_already = (o.x == aw && o.y == ah);
    picsong_params params = make_params(o);
    std::vector<Worker> w((size_t)nstreams);
    for (int i = 0; i < nstreams; i++) {
        Worker &k = w[(size_t)i];
        k.device = devs[(size_t)(i % ndevs)];
        HIPCK(hipSetDevice(k.device));
        CK(picsong_ctx_create(&params, k.device, &k.ctx));
        if (nstreams > 1) CK(picsong_ctx_set_pipelined(k.ctx, 1));     // the video engine keeps frames in flight
        load_lut(k.ctx, o, o.wl, 1, o.k, o.cp);
        HIPCK(hipStreamCreate(&k.stream));
        HIPCK(hipHostMalloc(&k.h_in, P * B));
        HIPCK(hipMalloc(&k.d_in, P * B));
        HIPCK(hipHostMalloc(&k.h_out, max_shorts * 2 * B));
        HIPCK(hipMalloc(&k.d_out, max_shorts * 2 * B));
        k.h_raw = (uint8_t *)malloc(frame_bytes);
        // stage timers (HIP events): at most kProfFrames launches per slot are timed, so the event count does
        // not grow with the video; "BPC acum time" scales their mean to all frames
        CK(picsong_profile_begin(k.ctx, (int)std::min<long>((ngroups + nstreams - 1) / nstreams, kProfFrames)));
    }
    const int fd = open(o.input.c_str(), O_RDONLY);
    if (fd < 0) die("Cannot open input file " + o.input);
    // image: one truncating write (IOManager.ipp:615-620); video: append + _SIZE (:176-190)
    const int ofd = open(o.output.c_str(), O_WRONLY | O_CREAT | (o.video ? 0 : O_TRUNC), 0644);
    if (ofd < 0) die("Cannot open output file " + o.output);
    const off_t out_base = o.video ? lseek(ofd, 0, SEEK_END) : 0;
    long total_shorts = 0;
    std::vector<int> frame_totals((size_t)nframes, 0);
    auto t0 = std::chrono::steady_clock::now();

    // slot state machine: FREE -(reader)-> FILLED -(main)-> LAUNCHED -(collector)-> COLLECTED
    // -(writer)-> FREE; a slot carries one GROUP of up to B consecutive frames; `expect` is the group the
    // slot takes next, so groups g and g + S never race
    enum { FREE = 0, FILLED = 1, LAUNCHED = 2, COLLECTED = 3 };
    struct SlotState { int state = FREE; long expect = 0; int n = 0; off_t offset[16]; int total[16]; };
    std::vector<SlotState> st((size_t)nstreams);
    for (int i = 0; i < nstreams; i++) st[(size_t)i].expect = i;
    auto group_frames = [&](long g) { return (int)std::min<long>(B, nframes - g * B); };
    std::mutex mu;
    std::condition_variable cv;
    std::string failure;                      // first error of a helper thread (reported by main)
    // where the host threads spend their time (reported with --metrics)
    std::vector<double> t_read(16, 0.0);
    double t_wait_gpu = 0.0, t_write = 0.0, t_main_wait = 0.0, t_launch = 0.0;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double>(b - a).count(); };

    auto reader = [&](int r, int nreaders) {
        for (long g = r; g < ngroups; g += nreaders) {
            const size_t si = (size_t)(g % nstreams);
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return !failure.empty() || (st[si].state == FREE && st[si].expect == g); });
                if (!failure.empty()) return;
            }
            Worker &k = w[si];
            const int n = group_frames(g);
            const auto tr0 = now();
            std::string err;
            for (int j = 0; j < n && err.empty(); j++) {
                const long f = g * B + j;
                uint8_t *dst = padded_already ? k.h_in + (size_t)j * P : k.h_raw;
                size_t got = 0;
                while (got < frame_bytes) {
                    ssize_t m = pread(fd, dst + got, frame_bytes - got, (off_t)(file_base + (size_t)f * frame_bytes + got));
                    if (m <= 0) break;
                    got += (size_t)m;
                }
                if (got != frame_bytes) err = "Input file is shorter than the requested frames.";
                else if (!padded_already && picsong_pad_frame_host(k.h_raw, o.x, o.y, k.h_in + (size_t)j * P, aw, ah) != PICSONG_OK)
                    err = picsong_last_error();
            }
            t_read[(size_t)r] += secs(tr0, now());
            std::lock_guard<std::mutex> lk(mu);
            if (!err.empty() && failure.empty()) failure = err;
            st[si].n = n;
            st[si].state = FILLED;
            cv.notify_all();
        }
    };
    // collector (one thread, group order): waits for the group's lengths and fixes every frame's place in
    // the file; writers (any order): copy the codestreams back over the slot's device's own host link,
    // pwrite them there and free the slot
    auto collector = [&] {
        off_t offset = out_base;
        for (long g = 0; g < ngroups; g++) {
            const size_t si = (size_t)(g % nstreams);
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return !failure.empty() || (st[si].state == LAUNCHED && st[si].expect == g); });
                if (!failure.empty()) return;
            }
            Worker &k = w[si];
            (void)hipSetDevice(k.device);
            const int n = st[si].n;
            int totals[16] = { 0 };
            std::string err;
            const auto tw0 = now();
            if (B == 1) { if (picsong_last_total(k.ctx, k.stream, &totals[0]) != PICSONG_OK) err = picsong_last_error(); }
            else if (picsong_last_totals(k.ctx, k.stream, n, totals) != PICSONG_OK) err = picsong_last_error();
            t_wait_gpu += secs(tw0, now());
            std::lock_guard<std::mutex> lk(mu);
            if (!err.empty() && failure.empty()) failure = err;
            for (int j = 0; j < n; j++) {
                frame_totals[(size_t)(g * B + j)] = totals[j];
                total_shorts += totals[j];
                st[si].offset[j] = offset;
                st[si].total[j] = totals[j];
                offset += (off_t)totals[j] * 2;
            }
            st[si].state = COLLECTED;
            cv.notify_all();
        }
    };
    std::vector<double> t_write_v(8, 0.0);
    auto writer = [&](int r, int nwriters) {
        for (long g = r; g < ngroups; g += nwriters) {
            const size_t si = (size_t)(g % nstreams);
            SlotState ss;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return !failure.empty() || (st[si].state == COLLECTED && st[si].expect == g); });
                if (!failure.empty()) return;
                ss = st[si];
            }
            const auto tw1 = now();
            std::string err;
            (void)hipSetDevice(w[si].device);
            for (int j = 0; j < ss.n && err.empty(); j++)
                if (hipMemcpyAsync(w[si].h_out + (size_t)j * max_shorts, w[si].d_out + (size_t)j * max_shorts,
                                   (size_t)ss.total[j] * 2, hipMemcpyDeviceToHost, w[si].stream) != hipSuccess)
                    err = "HIP error while copying a codestream back";
            if (err.empty() && hipStreamSynchronize(w[si].stream) != hipSuccess) err = "HIP error while copying a codestream back";
            for (int j = 0; j < ss.n && err.empty(); j++) {
                const char *src = reinterpret_cast<const char *>(w[si].h_out + (size_t)j * max_shorts);
                size_t left = (size_t)ss.total[j] * 2, done = 0;
                while (left) {
                    ssize_t m = pwrite(ofd, src + done, left, ss.offset[j] + (off_t)done);
                    if (m <= 0) { err = "Cannot write the output file " + o.output; break; }
                    done += (size_t)m; left -= (size_t)m;
                }
            }
            t_write_v[(size_t)r] += secs(tw1, now());
            std::lock_guard<std::mutex> lk(mu);
            if (!err.empty() && failure.empty()) failure = err;
            st[si].state = FREE;
            st[si].expect = g + nstreams;
            cv.notify_all();
        }
    };
    int nreaders = ngroups > 1 ? 2 : 1;
    if (const char *e = getenv("PICSONG_READERS")) { int v = atoi(e); if (v >= 1 && v <= 16) nreaders = v; }
    if (nreaders > nstreams - 1 && nstreams > 1) nreaders = nstreams - 1;
    std::vector<std::thread> threads;
    for (int r = 0; r < nreaders; r++) threads.emplace_back(reader, r, nreaders);
    int nwriters = ngroups > 1 ? 2 : 1;
    if (const char *e = getenv("PICSONG_WRITERS")) { int v = atoi(e); if (v >= 1 && v <= 8) nwriters = v; }
    threads.emplace_back(collector);
    for (int r = 0; r < nwriters; r++) threads.emplace_back(writer, r, nwriters);
    for (long g = 0; g < ngroups; g++) {
        const size_t si = (size_t)(g % nstreams);
        const auto tm0 = now();
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return !failure.empty() || (st[si].state == FILLED && st[si].expect == g); });
            if (!failure.empty()) break;
        }
        const auto tm1 = now();
        t_main_wait += secs(tm0, tm1);
        Worker &k = w[si];
        const int n = st[si].n;
        std::string err;
        if (hipSetDevice(k.device) != hipSuccess ||
            hipMemcpyAsync(k.d_in, k.h_in, P * (size_t)n, hipMemcpyHostToDevice, k.stream) != hipSuccess) err = "HIP error in the frame upload";
        else if (B == 1) { if (picsong_encode_frame(k.ctx, k.d_in, g == 0 ? 0 : 1, k.d_out, k.stream) != PICSONG_OK) err = picsong_last_error(); }
        else if (picsong_encode_frames(k.ctx, n, k.d_in, P, (int)(g * B), k.d_out, max_shorts, k.stream) != PICSONG_OK) err = picsong_last_error();
        t_launch += secs(tm1, now());
        std::lock_guard<std::mutex> lk(mu);
        if (!err.empty() && failure.empty()) failure = err;
        st[si].state = LAUNCHED;
        cv.notify_all();
    }
    for (auto &t : threads) t.join();
    close(fd);
    close(ofd);
    if (!failure.empty()) die(failure);
    if (o.video) {
        // IOManager.ipp:176-190: lengths in shorts, comma separated, appended in frame order
        std::ofstream sizes(o.output + "_SIZE", std::ios::binary | std::ios::app);
        for (long f = 0; f < nframes; f++) { if (f) sizes << ","; sizes << frame_totals[(size_t)f]; }
    }
    for (double v : t_write_v) t_write += v;
    double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (!o.metrics.empty()) {
        double rd = 0; for (double v : t_read) rd += v;
        std::cout << "host pipeline seconds: read(sum of " << nreaders << " readers) " << rd << ", main waiting for frames "
                  << t_main_wait << ", main launching " << t_launch << ", writer waiting for the GPU " << t_wait_gpu
                  << ", writer writing " << t_write << "; " << ndevs << " device(s), " << B << " frame(s) per launch" << std::endl;
    }

    double dwt = 0, bpc = 0, pack = 0;
    long counted = 0;
    for (auto &k : w) {
        int n = 0;
        std::vector<float> ms(3 * (size_t)std::min<long>((ngroups + nstreams - 1) / nstreams, kProfFrames) + 3);
        (void)hipSetDevice(k.device);
        CK(picsong_profile_read(k.ctx, &n, ms.data(), (int)(ms.size() / 3)));
        for (int i = 0; i < n; i++) { dwt += ms[3 * i]; bpc += ms[3 * i + 1]; pack += ms[3 * i + 2]; counted++; }
    }
    std::cout << "The time spent with the app without considering allocation periods is: " << sec << std::endl;
    std::cout << "BPC acum time is: " << (counted ? bpc / counted * (double)ngroups : 0.0) / 1e3 << std::endl;
    write_metrics(o, "encode", nframes, sec, counted ? dwt / counted / B : 0, counted ? bpc / counted / B : 0,
                  counted ? pack / counted / B : 0, total_shorts);
    for (auto &k : w) {
        (void)hipSetDevice(k.device);
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
    const bool batched = p.cp != 3;      // (picsong_decode_rgb_frame: one launch per stage for the three components)
    HIPCK(hipHostMalloc(&h_in, max_shorts * 2));
    HIPCK(hipMalloc(&d_in, max_shorts * 2 * (batched ? 3 : 1)));
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
            HIPCK(hipMemcpyAsync(d_in + (batched ? (size_t)c * max_shorts : 0), h_in, n * 2, hipMemcpyHostToDevice, s));
            if (!batched) CK(picsong_decode_plane(ctx, d_in, c, d_plane[c], s));
            HIPCK(hipStreamSynchronize(s));          // h_in (and, plane by plane, d_in) are reused for the next component
        }
        if (batched)
            CK(picsong_decode_rgb_frame(ctx, d_in, max_shorts, d_pix[0], d_pix[1], d_pix[2], s));
        else
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

// ---- video decode pipeline (DecodingEngine::runVideo, Engines/DecodingEngine.cu:866-1141: reader,
// worker and writer threads).  Every offset is known up front (the _SIZE sidecar gives the stream
// chunks, frames are W*H bytes), so: two reader threads pread chunks into pinned buffers, the main
// thread launches H2D + decode + D2H on the slot's stream, two writer threads wait for their slot's
// stream, crop and pwrite the frame at f * W * H (IOManager::writeDecodedFrame IO/IOManager.ipp:214-231
// appends raw W*H bytes per frame).
int run_decode_video(const Options &o, const picsong_params &p, const std::vector<long> &frame_shorts, long nframes)
{
    const int aw = picsong_pad_dim(p.width), ah = picsong_pad_dim(p.height);
    const size_t P = (size_t)aw * ah, max_shorts = picsong_max_stream_shorts(aw, ah);
    const size_t frame_bytes = (size_t)p.width * p.height;
    const int nslots = (o.streams < 3 ? 3 : o.streams) + 3;
    // groups of B consecutive frames per launch (picsong_decode_frames), as the encoder's video engine codes them:
    // a 4K frame alone is one decoder wave per SIMD
    int B = o.frames_per_launch > 0 ? o.frames_per_launch : (P <= (size_t)3840 * 2176 ? 4 : 1);
    if (p.cp == 3) B = 1;
    if (B > 16) B = 16;
    if ((long)B > nframes) B = (int)nframes;
    const long ngroups = (nframes + B - 1) / B;
    struct Slot {
        picsong_ctx *ctx = nullptr; hipStream_t stream = nullptr;
        uint16_t *h_in = nullptr, *d_in = nullptr; uint8_t *h_pix = nullptr, *d_pix = nullptr;
        std::vector<uint8_t> crop;
        int state = 0; long expect = 0;
    };
    enum { FREE = 0, FILLED = 1, LAUNCHED = 2 };
    std::vector<Slot> sl((size_t)nslots);
    for (int i = 0; i < nslots; i++) {
        Slot &k = sl[(size_t)i];
        CK(picsong_ctx_create(&p, o.device, &k.ctx));
        load_lut(k.ctx, o, p.wl, 1, p.k, p.cp);
        HIPCK(hipStreamCreate(&k.stream));
        HIPCK(hipHostMalloc(&k.h_in, max_shorts * 2 * B));
        HIPCK(hipMalloc(&k.d_in, max_shorts * 2 * B));
        HIPCK(hipHostMalloc(&k.h_pix, P * B));
        HIPCK(hipMalloc(&k.d_pix, P * B));
        if (aw != p.width) k.crop.resize(frame_bytes);
        k.expect = i;
    }
    std::vector<size_t> in_off((size_t)nframes + 1, 0);
    for (long f = 0; f < nframes; f++) {
        if ((size_t)frame_shorts[(size_t)f] > max_shorts) die("Frame codestream longer than the maximum for this geometry.");
        in_off[(size_t)f + 1] = in_off[(size_t)f] + (size_t)frame_shorts[(size_t)f];
    }
    const int ifd = open(o.input.c_str(), O_RDONLY);
    if (ifd < 0) die("Cannot open input file " + o.input);
    const int ofd = open(o.output.c_str(), O_WRONLY | O_CREAT, 0644);
    if (ofd < 0) die("Cannot open output file " + o.output);
    const off_t out_base = lseek(ofd, 0, SEEK_END);          // appended, like the reference
    std::mutex mu;
    std::condition_variable cv;
    std::string failure;
    auto t0 = std::chrono::steady_clock::now();

    auto reader = [&](int r, int nreaders) {
        for (long g = r; g < ngroups; g += nreaders) {
            Slot &k = sl[(size_t)(g % nslots)];
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return !failure.empty() || (k.state == FREE && k.expect == g); });
                if (!failure.empty()) return;
            }
            bool short_file = false;
            for (long f = g * B; f < nframes && f < (g + 1) * B; f++) {
                const size_t bytes = (size_t)frame_shorts[(size_t)f] * 2;
                char *dst = reinterpret_cast<char *>(k.h_in + (size_t)(f - g * B) * max_shorts);
                size_t got = 0;
                while (got < bytes) {
                    ssize_t n = pread(ifd, dst + got, bytes - got, (off_t)(in_off[(size_t)f] * 2 + got));
                    if (n <= 0) break;
                    got += (size_t)n;
                }
                short_file = short_file || got != bytes;
            }
            std::lock_guard<std::mutex> lk(mu);
            if (short_file && failure.empty()) failure = "Input file is shorter than its _SIZE sidecar says.";
            k.state = FILLED;
            cv.notify_all();
        }
    };
    auto writer = [&](int r, int nwriters) {
        (void)hipSetDevice(o.device);
        for (long g = r; g < ngroups; g += nwriters) {
            Slot &k = sl[(size_t)(g % nslots)];
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return !failure.empty() || (k.state == LAUNCHED && k.expect == g); });
                if (!failure.empty()) return;
            }
            std::string err;
            if (hipStreamSynchronize(k.stream) != hipSuccess) err = "HIP error while decoding a frame";
            for (long f = g * B; err.empty() && f < nframes && f < (g + 1) * B; f++) {
                const uint8_t *pix = k.h_pix + (size_t)(f - g * B) * P, *src = pix;
                if (aw != p.width) {
                    for (int y = 0; y < p.height; y++) memcpy(&k.crop[(size_t)y * p.width], pix + (size_t)y * aw, (size_t)p.width);
                    src = k.crop.data();
                }
                size_t left = frame_bytes, done = 0;
                while (left) {
                    ssize_t n = pwrite(ofd, src + done, left, out_base + (off_t)((size_t)f * frame_bytes + done));
                    if (n <= 0) { err = "Cannot write the output file " + o.output; break; }
                    done += (size_t)n; left -= (size_t)n;
                }
            }
            std::lock_guard<std::mutex> lk(mu);
            if (!err.empty() && failure.empty()) failure = err;
            k.state = FREE;
            k.expect = g + nslots;
            cv.notify_all();
        }
    };
    int nreaders = 2, nwriters = 2;
    if (const char *e = getenv("PICSONG_READERS")) { int v = atoi(e); if (v >= 1 && v <= 16) nreaders = v; }
    if (const char *e = getenv("PICSONG_WRITERS")) { int v = atoi(e); if (v >= 1 && v <= 16) nwriters = v; }
    if (nreaders > nslots - 1) nreaders = nslots - 1;
    if (nwriters > nslots - 1) nwriters = nslots - 1;
    std::vector<std::thread> threads;
    for (int r = 0; r < nreaders; r++) threads.emplace_back(reader, r, nreaders);
    for (int r = 0; r < nwriters; r++) threads.emplace_back(writer, r, nwriters);
    for (long g = 0; g < ngroups; g++) {
        Slot &k = sl[(size_t)(g % nslots)];
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return !failure.empty() || (k.state == FILLED && k.expect == g); });
            if (!failure.empty()) break;
        }
        const int n = (int)std::min<long>(B, nframes - g * B);
        std::string err;
        for (int b = 0; err.empty() && b < n; b++)
            if (hipMemcpyAsync(k.d_in + (size_t)b * max_shorts, k.h_in + (size_t)b * max_shorts,
                               (size_t)frame_shorts[(size_t)(g * B + b)] * 2, hipMemcpyHostToDevice, k.stream) != hipSuccess)
                err = "HIP error in the codestream upload";
        if (!err.empty()) { }
        else if (picsong_decode_frames(k.ctx, n, k.d_in, max_shorts, k.d_pix, P, k.stream) != PICSONG_OK) err = picsong_last_error();
        else if (hipMemcpyAsync(k.h_pix, k.d_pix, P * (size_t)n, hipMemcpyDeviceToHost, k.stream) != hipSuccess)
            err = "HIP error in the frame download";
        std::lock_guard<std::mutex> lk(mu);
        if (!err.empty() && failure.empty()) failure = err;
        k.state = LAUNCHED;
        cv.notify_all();
    }
    for (auto &t : threads) t.join();
    close(ifd);
    close(ofd);
    if (!failure.empty()) die(failure);
    double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::cout << "The time spent with the app without considering allocation periods and I/O is: " << sec << std::endl;
    Options mo = o;
    mo.x = p.width; mo.y = p.height;
    write_metrics(mo, "decode", nframes, sec, 0, 0, 0, (long)in_off[(size_t)nframes]);
    for (auto &k : sl) {
        picsong_ctx_destroy(k.ctx);
        (void)hipStreamDestroy(k.stream);
        (void)hipHostFree(k.h_in); (void)hipFree(k.d_in); (void)hipHostFree(k.h_pix); (void)hipFree(k.d_pix);
    }
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
    if (!((p.components == 1 && !p.is_rgb) || (p.components == 3 && p.is_rgb)))
        die("This stream uses a component layout not built here.");
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
    load_lut(ctx, lo, p.wl, p.components, p.k, p.cp);
    int aw, ah, ncb;
    CK(picsong_ctx_padded_dims(ctx, &aw, &ah, &ncb));
    if (p.is_rgb) return run_decode_rgb(o, p, ctx, in, frame_shorts, nframes, aw, ah);
    if (o.video) {
        picsong_ctx_destroy(ctx);
        in.close();
        return run_decode_video(lo, p, frame_shorts, nframes);
    }
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
