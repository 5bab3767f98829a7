"""The committed golden vectors (tests/golden/vectors.json, made by tests/golden/make_golden.py from
the CPU oracle): the oracle still reproduces them, the kernels on the CPU wave emulator
reproduce them, and (-m gpu) so does the HIP path through the C ABI -- the latter two WITHOUT calling
the oracle: inputs come from a numpy restatement of the integer generator, expected values from the
file."""
import hashlib
import json
import os
import sys
import zlib

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
VECTORS = json.load(open(os.path.join(HERE, "golden", "vectors.json")))["cases"]
IDS = [c["name"] for c in VECTORS]


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def gen_frame_np(W, H, frame=0, seed=0x5EED1234):
    """SURVEY.md 8d generator, restated in Python: triangle-wave base + LCG noise + checkerboard."""
    def tri(v, p):
        return np.abs(v % (2 * p) - p)
    y, x = np.mgrid[0:H, 0:W].astype(np.int64)
    base = (tri(x, 512) * 255 // 512 + tri(y, 384) * 255 // 384) // 2
    z = (seed ^ (frame * 0x9E3779B9)) & 0xFFFFFFFF
    noise = np.empty(W * H, np.int64)
    for i in range(W * H):
        z = (1664525 * z + 1013904223) & 0xFFFFFFFF
        noise[i] = ((z >> 24) & 15) - 8
    checker = 16 * (((x >> 5) ^ (y >> 5)) & 1)
    return np.clip(base + noise.reshape(H, W) + checker, 0, 255).astype(np.uint8)


def check_stages(v, dwt_words, staging, sizes, stream):
    """dwt_words: the transform output as raw 32-bit words (P of them); staging: int32[nCB*4096]."""
    n = v["W"]
    words = np.ascontiguousarray(dwt_words)[:n * n]
    assert crc(words.view(np.uint32)) == v["dwt_words_crc32"]
    st = np.asarray(staging).reshape(-1, 4096)
    assert [int(s) for s in sizes] == v["sizes"]
    assert [int(m) for m in st[:, 0]] == v["msb"]
    for cb, want in v["codewords"].items():
        got = [int(w) & 0xFFFF for w in st[int(cb), 1:1 + len(want)]]
        assert got == want, f"codeblock {cb}"
    stream = np.ascontiguousarray(stream).view(np.uint16)
    assert stream.size == v["stream_shorts"]
    assert hashlib.sha256(stream.tobytes()).hexdigest() == v["stream_sha256"]


def test_generator_restatement_matches_the_vectors():
    for v in VECTORS:
        if v["W"] <= 128:
            assert crc(gen_frame_np(v["W"], v["H"], v["frame"])) == v["input_crc32"]


@pytest.mark.parametrize("v", VECTORS, ids=IDS)
def test_oracle_reproduces_golden_vectors(v):
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import make_golden
    got = make_golden.case_vector(v["W"], v["lossy"], v["wl"], v["qs"], v["k"])
    assert got == v


@pytest.mark.parametrize("v", [c for c in VECTORS if c["W"] <= 128], ids=[n for n in IDS if not n.startswith("512")])
def test_emulated_kernels_reproduce_golden_vectors(v):
    import emu_lib as E
    import oracle_lib as orc              # LUT loading and the 9-short header only
    n, lossy, wl, qs, k = v["W"], v["lossy"], v["wl"], v["qs"], v["k"]
    img = gen_frame_np(n, n, v["frame"])
    assert crc(img) == v["input_crc32"]
    lut = orc.lut_for_k(lossy, wl) if k > 0 else orc.lut_for(lossy, wl)
    f = E.dwt_forward(img, wl, lossy, qs, extra=n * n)           # fused u8 ingest
    coef = f[:n * n].reshape(n, n)
    staging, sizes, flag = E.bpc_encode(coef, wl, lut, k=k)
    assert flag == 0
    hdr = orc.header_pack(n_samples=n * n, cp=2, cb_height=18, cb_width=64, wl=wl, bit_depth=8, lossy=int(lossy),
                          qs_1e4=int(qs * 10000), components=1, is_rgb=0, height=n, endianess=0, bps=8,
                          is_signed=0, frames=0, k_1e3=int(k * 1000))
    stream = E.pack(staging, sizes, hdr)
    check_stages(v, coef.view(np.uint32), staging, sizes, stream)
    assert crc(np.trunc(coef).astype(np.int32) if lossy else coef.astype(np.int32)) == v["coded_coefficients_crc32"]


@pytest.fixture(scope="module")
def torch():
    import torch as t
    if not t.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU: the HIP path has no CPU fallback")
    return t


@pytest.fixture(scope="module")
def pa():
    import picsong_amd
    picsong_amd.load()          # raises if the extension is missing
    return picsong_amd


@pytest.mark.gpu
@pytest.mark.parametrize("v", VECTORS, ids=IDS)
def test_hip_path_reproduces_golden_vectors(v, pa, torch):
    n, lossy, wl, qs, k = v["W"], v["lossy"], v["wl"], v["qs"], v["k"]
    img = gen_frame_np(n, n, v["frame"])
    assert crc(img) == v["input_crc32"]
    lutdir = os.path.join(HERE, "golden", "lut", "n1_lossy" if lossy else "n1_lossless")
    c = pa.Codec(n, n, wl=wl, lossy=lossy, qs=qs, lut_folder=lutdir, k=k)
    d_img = torch.from_numpy(img).cuda()
    f = c.dwt_forward(d_img)
    st, sz = c.bpc_encode(f[:n * n].clone())
    assert c.range_flag() == 0
    stream = c.encode_frame(d_img, 0).cpu().numpy()
    check_stages(v, f.cpu().numpy()[:n * n].view(np.uint32), st.cpu().numpy(), sz.cpu().numpy(), stream)
    dec = c.decode_frame(torch.from_numpy(stream).cuda()).cpu().numpy()
    if not lossy:
        assert np.array_equal(dec, img)
    c.close()
