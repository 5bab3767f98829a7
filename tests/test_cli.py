"""PICSONG command-line tool (cuda-image-and-video-codec_amd/host): flag validation on CPU, and on a
GPU the file-level contract -- image encode == oracle codestream, decode -> P5 PGM round trip,
video encode -> <o> + <o>_SIZE in frame order, video decode -> raw frames."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "cuda-image-and-video-codec_amd", "host", "PICSONG")


def _run(*args):
    return subprocess.run([BIN, *map(str, args)], capture_output=True, text=True)


@pytest.fixture(scope="module", autouse=True)
def _built():
    if not os.path.exists(BIN):
        subprocess.check_call(["make", "-C", os.path.dirname(BIN)])


def test_help_and_validation():
    assert "-LUTFolder" in _run("-h").stdout
    r = _run("-cd", 0, "-i", "/etc/hostname", "-o", "/tmp/x", "-xSize", 0, "-ySize", 10)
    assert r.returncode == 255 and "Incorrect parameters. Please choose valid values." in r.stdout
    assert "User entered -xSize command 0" in r.stdout
    for extra, msg in ((("-wl", 11), "Incorrect parameters"), (("-cbWidth", 65), "Incorrect parameters"),
                       (("-cbHeight", 21), "Incorrect parameters"), (("-qs", 1.5), "Incorrect parameters"),
                       (("-isRGB", 1), "Incorrect parameters"), (("-components", 3), "Incorrect parameters"),
                       (("-cp", 4), "Incorrect parameters"), (("-cp", 3, "-k", 1), "no complexity-scalable mode"),
                       (("-k", 70), "Incorrect parameters"),
                       (("-k", -1), "Incorrect parameters")):
        base = () if extra[0] == "-wl" else ("-wl", 1)
        r = _run("-cd", 0, "-i", "/etc/hostname", "-o", "/tmp/x", "-xSize", 64, "-ySize", 64, *base, *extra)
        assert r.returncode == 255 and msg in r.stdout, (extra, r.stdout)
    assert _run("-cd", 2, "-i", "a", "-o", "b").returncode == 255


@pytest.mark.gpu
def test_image_files_roundtrip_and_oracle_parity(oracle, tmp_path):
    W, H, wl = 700, 500, 4
    img = oracle.gen_frame(W, H, 3)
    lutdir = os.path.join(oracle.LUT_DIR, "n1_lossless")
    raw, enc, dec = tmp_path / "in.raw", tmp_path / "out.enc", tmp_path / "out.pgm"
    img.tofile(raw)
    r = _run("-cd", 0, "-i", raw, "-o", enc, "-xSize", W, "-ySize", H, "-wl", wl, "-type", 0, "-LUTFolder", lutdir,
             "--metrics", tmp_path / "m.json")
    assert r.returncode == 0, r.stdout + r.stderr
    assert "BPC acum time is:" in r.stdout and "The time spent with the app is:" in r.stdout
    ref = oracle.encode_frame(img, wl, False, 1.0, oracle.lut_for(False, wl), 0, 0)
    assert np.array_equal(np.fromfile(enc, np.uint16), ref)
    r = _run("-cd", 1, "-i", enc, "-o", dec, "-LUTFolder", lutdir)
    assert r.returncode == 0, r.stdout + r.stderr
    data = open(dec, "rb").read()
    head = f"P5\n{W} {H}\n255\n".encode()
    assert data.startswith(head) and np.array_equal(np.frombuffer(data[len(head):], np.uint8).reshape(H, W), img)
    # P5 input: sizes taken from the header
    pgm = tmp_path / "in.pgm"
    pgm.write_bytes(head + img.tobytes())
    enc2 = tmp_path / "out2.enc"
    assert _run("-cd", 0, "-i", pgm, "-o", enc2, "-wl", wl, "-LUTFolder", lutdir).returncode == 0
    assert np.array_equal(np.fromfile(enc2, np.uint16), ref)


@pytest.mark.gpu
def test_video_files_roundtrip(oracle, tmp_path):
    W, H, wl, F = 256, 192, 2, 5
    frames = [oracle.gen_frame(W, H, f) for f in range(F)]
    lutdir = os.path.join(oracle.LUT_DIR, "n1_lossy")
    raw, enc, dec = tmp_path / "v.raw", tmp_path / "v.enc", tmp_path / "v.dec"
    np.concatenate([f.ravel() for f in frames]).tofile(raw)
    r = _run("-cd", 0, "-i", raw, "-o", enc, "-xSize", W, "-ySize", H, "-wl", wl, "-type", 1, "-qs", 0.5,
             "-video", 1, "-frames", F, "-numberOfStreams", 3, "-LUTFolder", lutdir)
    assert r.returncode == 0, r.stdout + r.stderr
    lut = oracle.lut_for(True, wl)
    ref = [oracle.encode_frame(frames[f], wl, True, 0.5, lut, 0 if f == 0 else 1, F) for f in range(F)]
    assert np.array_equal(np.fromfile(enc, np.uint16), np.concatenate(ref))
    assert open(str(enc) + "_SIZE").read() == ",".join(str(x.size) for x in ref)
    r = _run("-cd", 1, "-i", enc, "-o", dec, "-video", 1, "-LUTFolder", lutdir)
    assert r.returncode == 0, r.stdout + r.stderr
    got = np.fromfile(dec, np.uint8).reshape(F, H, W)
    for f in range(F):
        assert np.array_equal(got[f], oracle.decode_frame(ref[f], W, H, wl, True, 0.5, lut))


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [
    ["-framesPerLaunch", 3],                                   # groups of 3 frames per launch, 11 = 3+3+3+2
    ["-framesPerLaunch", 2, "--devices", "0,0"],               # two worker sets (the -gpus sharding, on one GPU)
    ["-framesPerLaunch", 1, "--devices", "0,0,0", "-numberOfStreams", 4],
    ["-framesPerLaunch", 3, "-k", 0.5],                        # -k > 0 video: groups of frames per launch too (round 4)
])
def test_video_sharded_and_batched_equals_oracle(oracle, tmp_path, extra):
    """The video engine with groups of frames per launch and with the groups sharded round-robin over several
    worker sets (-gpus N / --devices): the output file and _SIZE are the 1-GPU, frame-by-frame ones
    (CodingEngine::engineManager Engines/CodingEngine.cu:990-1061 round-robins frames over its workers the
    same way, iteration % N)."""
    W, H, wl, F = 320, 256, 3, 11
    frames = [oracle.gen_frame(W, H, 40 + f) for f in range(F)]
    lutdir = os.path.join(oracle.LUT_DIR, "n1_lossless")
    raw, enc, dec = tmp_path / "v.raw", tmp_path / "v.enc", tmp_path / "v.dec"
    np.concatenate([f.ravel() for f in frames]).tofile(raw)
    r = _run("-cd", 0, "-i", raw, "-o", enc, "-xSize", W, "-ySize", H, "-wl", wl, "-type", 0,
             "-video", 1, "-frames", F, "-LUTFolder", lutdir, *extra)
    assert r.returncode == 0, r.stdout + r.stderr
    k = float(extra[extra.index("-k") + 1]) if "-k" in extra else 0.0
    lut = oracle.lut_for_k(False, wl) if k > 0 else oracle.lut_for(False, wl)
    ref = [oracle.encode_frame(frames[f], wl, False, 1.0, lut, 0 if f == 0 else 1, F, k=k) for f in range(F)]
    assert np.array_equal(np.fromfile(enc, np.uint16), np.concatenate(ref))
    assert open(str(enc) + "_SIZE").read() == ",".join(str(x.size) for x in ref)
    r = _run("-cd", 1, "-i", enc, "-o", dec, "-video", 1, "-LUTFolder", lutdir)
    assert r.returncode == 0, r.stdout + r.stderr
    assert np.array_equal(np.fromfile(dec, np.uint8), np.fromfile(raw, np.uint8))
    # the decoder's video engine takes groups of frames per launch too (default 4: 4 + 4 + 3 above)
    for b in (1, 3):
        d2 = tmp_path / f"v{b}.dec"
        r = _run("-cd", 1, "-i", enc, "-o", d2, "-video", 1, "-LUTFolder", lutdir, "-framesPerLaunch", b)
        assert r.returncode == 0, r.stdout + r.stderr
        assert np.array_equal(np.fromfile(d2, np.uint8), np.fromfile(raw, np.uint8))


@pytest.mark.gpu
def test_three_coding_passes_files(oracle, tmp_path):
    """-cp 3 through the CLI: the coded file equals the oracle's stream, the decoder takes the mode from the
    header (Engines/DecodingEngine.cu:567-585) and returns the image."""
    W, H, wl = 520, 390, 3
    img = oracle.gen_frame(W, H, 12)
    lutdir = os.path.join(oracle.LUT_CP3_DIR, "n1_lossless")
    raw, enc, dec = tmp_path / "i.raw", tmp_path / "i.enc", tmp_path / "i.pgm"
    img.tofile(raw)
    r = _run("-cd", 0, "-i", raw, "-o", enc, "-xSize", W, "-ySize", H, "-wl", wl, "-type", 0, "-cp", 3, "-LUTFolder", lutdir)
    assert r.returncode == 0, r.stdout + r.stderr
    ref = oracle.encode_frame(img, wl, False, 1.0, oracle.lut_for_cp3(False, wl))
    assert np.array_equal(np.fromfile(enc, np.uint16), ref)
    r = _run("-cd", 1, "-i", enc, "-o", dec, "-LUTFolder", lutdir)
    assert r.returncode == 0, r.stdout + r.stderr
    data = open(dec, "rb").read()
    assert np.array_equal(np.frombuffer(data[-W * H:], np.uint8).reshape(H, W), img)


@pytest.mark.gpu
def test_more_gpus_than_the_node_has_is_refused(oracle, tmp_path):
    raw = tmp_path / "v.raw"
    oracle.gen_frame(256, 256, 0).tofile(raw)
    r = _run("-cd", 0, "-i", raw, "-o", tmp_path / "v.enc", "-xSize", 256, "-ySize", 256, "-wl", 2, "-video", 1,
             "-frames", 1, "-gpus", 64, "-LUTFolder", os.path.join(oracle.LUT_DIR, "n1_lossless"))
    assert r.returncode != 0 and "this node has" in r.stdout


@pytest.mark.gpu
def test_4k_video_file_equals_oracle_frame_by_frame(oracle, tmp_path):
    """BASELINE configs[3] on one GPU through the CLI: a 4K (3840x2160) greyscale video, -type 0, wl 5,
    -numberOfStreams 3; every frame's codestream equals the oracle's (header on frame 0 only,
    BitStreamBuilder.cu:277-278), _SIZE lists the lengths in frame order, decode returns the frames
    (CodingEngine::runVideo Engines/CodingEngine.cu:819-872, writeCodedFrame IO/IOManager.ipp:176-190)."""
    W, H, wl, F = 3840, 2160, 5, 18
    oracle.set_threads(oracle.usable_threads())
    try:
        lutdir = os.path.join(oracle.LUT_DIR, "n1_lossless")
        raw, enc, dec = tmp_path / "v4k.raw", tmp_path / "v4k.enc", tmp_path / "v4k.dec"
        with open(raw, "wb") as f:
            for i in range(F):
                f.write(oracle.gen_frame(W, H, i).tobytes())
        r = _run("-cd", 0, "-i", raw, "-o", enc, "-xSize", W, "-ySize", H, "-wl", wl, "-type", 0,
                 "-video", 1, "-frames", F, "-numberOfStreams", 3, "-LUTFolder", lutdir)
        assert r.returncode == 0, r.stdout + r.stderr
        lut = oracle.lut_for(False, wl)
        got = np.fromfile(enc, np.uint16)
        sizes = [int(x) for x in open(str(enc) + "_SIZE").read().split(",")]
        assert len(sizes) == F and sum(sizes) == got.size
        pos = 0
        for i in range(F):
            ref = oracle.encode_frame(oracle.gen_frame(W, H, i), wl, False, 1.0, lut, 0 if i == 0 else 1, F)
            assert sizes[i] == ref.size, f"frame {i}: _SIZE {sizes[i]} != oracle {ref.size}"
            assert np.array_equal(got[pos:pos + sizes[i]], ref), f"frame {i} differs from the oracle"
            pos += sizes[i]
        r = _run("-cd", 1, "-i", enc, "-o", dec, "-video", 1, "-LUTFolder", lutdir)
        assert r.returncode == 0, r.stdout + r.stderr
        assert np.array_equal(np.fromfile(dec, np.uint8), np.fromfile(raw, np.uint8))
    finally:
        oracle.set_threads(1)


@pytest.mark.gpu
@pytest.mark.parametrize("lossy,video,k", [(False, False, 0.0), (True, True, 0.0), (False, True, 0.5), (True, False, 1.5)])
def test_rgb_files_roundtrip_and_oracle_parity(oracle, tmp_path, lossy, video, k):
    """-isRGB 1 -components 3: planar R,G,B planes; every component stream equals the oracle's
    (RCT/ICT + per-component LUT), header on component 0 (image) / on frame 0's three components
    (video), decode returns planar planes.  With -k > 0 too (every component with its own bit-plane tables)."""
    W, H, wl, qs, F = 256, 192, 2, (0.5 if lossy else 1.0), (2 if video else 1)
    lutdir = os.path.join(oracle.LUT_DIR, "n1_lossy" if lossy else "n1_lossless")
    planes = [[oracle.gen_frame(W, H, 10 * f + c) for c in range(3)] for f in range(F)]
    raw, enc, dec = tmp_path / "rgb.raw", tmp_path / "rgb.enc", tmp_path / "rgb.dec"
    np.concatenate([p.ravel() for fr in planes for p in fr]).tofile(raw)
    args = ["-cd", 0, "-i", raw, "-o", enc, "-xSize", W, "-ySize", H, "-wl", wl, "-type", int(lossy), "-qs", qs,
            "-isRGB", 1, "-components", 3, "-LUTFolder", lutdir]
    if k > 0:
        args += ["-k", k]
    if video:
        args += ["-video", 1, "-frames", F]
    r = _run(*args)
    assert r.returncode == 0, r.stdout + r.stderr
    hdr = oracle.header_pack(n_samples=W * H * 3, cp=2, cb_height=18, cb_width=64, wl=wl, bit_depth=8, lossy=int(lossy),
                             qs_1e4=int(qs * 10000), components=3, is_rgb=1, height=H, endianess=0, bps=8,
                             is_signed=0, frames=F if video else 0, k_1e3=int(round(k * 1000)))
    ref = []
    for f in range(F):
        comps = oracle.rgb_forward(*[oracle.pad_frame(p) for p in planes[f]], lossy)
        for c in range(3):
            with_hdr = (f == 0) if video else (c == 0)
            ref.append(oracle.encode_plane(comps[c], wl, lossy, qs, oracle.lut_for_component(lossy, wl, c, k=k),
                                           hdr if with_hdr else None, k=k))
    assert np.array_equal(np.fromfile(enc, np.uint16), np.concatenate(ref))
    assert open(str(enc) + "_SIZE").read() == ",".join(str(x.size) for x in ref)
    args = ["-cd", 1, "-i", enc, "-o", dec, "-LUTFolder", lutdir] + (["-video", 1] if video else [])
    r = _run(*args)
    assert r.returncode == 0, r.stdout + r.stderr
    got = np.fromfile(dec, np.uint8).reshape(F, 3, H, W)
    for f in range(F):
        for c in range(3):
            if not lossy:
                assert np.array_equal(got[f, c], planes[f][c])
            else:
                mse = np.mean((got[f, c].astype(np.float64) - planes[f][c]) ** 2)
                assert 10 * np.log10(255 ** 2 / mse) > 35.0


@pytest.mark.gpu
@pytest.mark.parametrize("lossy,k", [(False, 0.3), (False, 2.0), (True, 0.5)])
def test_complexity_scalable_files_roundtrip_and_oracle_parity(oracle, tmp_path, lossy, k):
    """-k > 0: bit-plane LUT files, header word 8, decoder takes k from the stream."""
    W, H, wl, qs = 520, 390, 3, 0.5
    img = oracle.gen_frame(W, H, 9)
    lutdir = os.path.join(oracle.LUT_DIR, "n1_lossy" if lossy else "n1_lossless")
    raw, enc, dec = tmp_path / "in.raw", tmp_path / "out.enc", tmp_path / "out.pgm"
    img.tofile(raw)
    r = _run("-cd", 0, "-i", raw, "-o", enc, "-xSize", W, "-ySize", H, "-wl", wl, "-type", int(lossy), "-qs", qs,
             "-k", k, "-LUTFolder", lutdir)
    assert r.returncode == 0, r.stdout + r.stderr
    assert f"User entered -k command" in r.stdout
    lut = oracle.lut_for_k(lossy, wl)
    ref = oracle.encode_frame(img, wl, lossy, qs, lut, 0, 0, k=k)      # the header carries -qs as given
    got = np.fromfile(enc, np.uint16)
    assert np.array_equal(got, ref)
    assert oracle.header_unpack(got[:9])["k_1e3"] == int(np.float32(k) * np.float32(1000))
    r = _run("-cd", 1, "-i", enc, "-o", dec, "-LUTFolder", lutdir)
    assert r.returncode == 0, r.stdout + r.stderr
    data = open(dec, "rb").read()
    head = f"P5\n{W} {H}\n255\n".encode()
    out = np.frombuffer(data[len(head):], np.uint8).reshape(H, W)
    kdec = np.float32(oracle.header_unpack(got[:9])["k_1e3"] / 1000.0)
    assert np.array_equal(out, oracle.decode_frame(ref, W, H, wl, lossy, qs, lut, k=float(kdec)))
    if not lossy:
        assert np.array_equal(out, img)


FACADE = os.path.join(ROOT, "tests", "facade", "facade_demo")


@pytest.mark.gpu
@pytest.mark.parametrize("lossy", [0, 1])
def test_facade_classes_drive_the_library(oracle, lossy):
    """include/picsong_facade.hpp: the reference's DWT<T,Y> / BPCCuda<T> classes over the C ABI, called
    in the reference engines' order, give picsong_encode_frame's codestream and reconstruct the frame."""
    if not os.path.exists(FACADE):
        subprocess.check_call(["make", "-C", os.path.dirname(FACADE)])
    lutdir = os.path.join(oracle.LUT_DIR, "n1_lossy" if lossy else "n1_lossless")
    r = subprocess.run([FACADE, "600", "410", "4", str(lossy), "0.5", lutdir], capture_output=True, text=True)
    assert r.returncode == 0 and "FACADE OK" in r.stdout, r.stdout + r.stderr
