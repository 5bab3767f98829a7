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
                       (("-isRGB", 1), "not built"), (("-cp", 3), "not built"), (("-k", 0.5), "not built")):
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
