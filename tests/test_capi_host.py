"""C-ABI checks that need no GPU: the library loads, exports every symbol include/picsong_hip.h
declares, its host-side functions (header, LUT parser, padding, geometry) agree with the oracle,
and device entry points fail loudly (no CPU fallback) when no GPU is present."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import picsong_amd as pa

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    L = pa.load()
    hdr = open(os.path.join(ROOT, "include", "picsong_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(picsong_[a-z0-9_]+)\s*\(", hdr)))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/picsong_hip.h but not exported"
    assert sorted(pa.EXPORTS) == declared


def test_geometry_helpers(oracle):
    for v in (1, 63, 64, 65, 2160, 4320, 7680):
        assert pa.pad_dim(v) == oracle.pad_dim(v)
    for aw, ah, wl in ((512, 512, 3), (3840, 2176, 5), (7680, 4352, 6), (16384, 16384, 5)):
        assert pa.dwt_extra(aw, ah, wl) == oracle.dwt_extra(aw, ah, wl)
    assert pa.load().picsong_max_stream_shorts(128, 64) == 9 + 2 * 2 + 128 * 64 + 1


def test_header_matches_oracle(oracle):
    p = pa.make_params(3840, 2160, wl=5, lossy=True, qs=0.5, frames=256)
    s = pa.header_pack(p)
    ref = oracle.header_pack(n_samples=3840 * 2160, cp=2, cb_height=18, cb_width=64, wl=5, bit_depth=8,
                             lossy=1, qs_1e4=5000, components=1, is_rgb=0, height=2160, endianess=0,
                             bps=8, is_signed=0, frames=256, k_1e3=0)
    assert np.array_equal(s, ref)
    q = pa.header_unpack(s)
    assert (q.width, q.height, q.wl, q.lossy, q.cp, q.frames) == (3840, 2160, 5, 1, 2, 256)
    assert abs(q.qs - 0.5) < 1e-6 and q.cb_width == 64 and q.cb_height == 18


@pytest.mark.parametrize("folder,wl", [("n1_lossless", 3), ("n1_lossless", 5), ("n1_lossy", 6)])
def test_lut_parser_matches_oracle(oracle, folder, wl):
    path = os.path.join(oracle.LUT_DIR, folder)
    info, table = pa.lut_load(path, wl, component=1, fill=0)
    ref = oracle.Lut(path, wl, 1, 0)
    assert (info.n_ref, info.n_sig, info.n_sign) == (ref.c.n_ref, ref.c.n_sig, ref.c.n_sign)
    assert (info.ctx_sig, info.ctx_sign, info.ctx_ref, info.precision) == (9, 4, 1, 7)
    assert np.array_equal(table, ref.table)


def test_lut_parser_bit_plane_tables_match_oracle(oracle):
    path = os.path.join(oracle.LUT_DIR, "n1_lossless")
    info, table = pa.lut_load(path, 4, component=1, fill=0, n_tables=0)      # the -k > 0 layout
    ref = oracle.lut_for_k(False, 4)
    assert info.n_tables == ref.n_tables == 15
    assert np.array_equal(table, ref.table)
    info1, table1 = pa.lut_load(path, 4, component=1, fill=0)
    assert info1.n_tables == 1 and np.array_equal(table1, table[:table1.size])


def test_lut_parser_cp3_matches_oracle(oracle):
    path = os.path.join(oracle.LUT_CP3_DIR, "n1_lossless")
    info, table = pa.lut_load(path, 4, component=1, fill=0, cp=3)
    ref = oracle.lut_for_cp3(False, 4)
    assert info.cp == 3 and table.size == ref.table.size
    assert np.array_equal(table, ref.table)
    info2, _ = pa.lut_load(path, 4, component=1, fill=0)
    assert info2.cp == 2


def test_lut_missing_folder_reports_error():
    info = pa.LutInfo()
    rc = pa.load().picsong_lut_load(b"/nonexistent/", 1, 5, 0, C.byref(info), None, 0)
    assert rc == -3 and b"header.txt" in pa.load().picsong_last_error()


def test_pad_frame_host_matches_oracle(oracle):
    img = oracle.gen_frame(100, 70)
    out = np.empty((128, 128), np.uint8)
    rc = pa.load().picsong_pad_frame_host(img.ctypes.data_as(C.c_void_p), 100, 70,
                                          out.ctypes.data_as(C.c_void_p), 128, 128)
    assert rc == 0 and np.array_equal(out, oracle.pad_frame(img))


@pytest.mark.parametrize("w,h", [(10, 10), (100, 20), (31, 64), (64, 31)])
def test_pad_frame_refuses_frames_smaller_than_their_padding(oracle, w, h):
    """Column w + j mirrors column w - 1 - j: with 2w < AW (or 2h < AH) the reference's loop
    (IO/IOManager.ipp:101-108) indexes before its vector -- refused by the library, the oracle and
    picsong_ctx_create alike instead of reading out of bounds."""
    L = pa.load()
    aw, ah = pa.pad_dim(w), pa.pad_dim(h)
    # guard pages' worth of poison around the input: an under-read would not go unnoticed under ASan,
    # and here the call must not touch anything at all
    img = np.full(w * h, 7, np.uint8)
    out = np.full((ah, aw), 0xAB, np.uint8)
    rc = L.picsong_pad_frame_host(img.ctypes.data_as(C.c_void_p), w, h, out.ctypes.data_as(C.c_void_p), aw, ah)
    assert rc == -1 and b"pad_frame" in L.picsong_last_error()
    assert (out == 0xAB).all()
    with pytest.raises(ValueError):
        oracle.pad_frame(img.reshape(h, w))
    hnd = C.c_void_p()
    p = pa.make_params(w, h, wl=1)
    assert L.picsong_ctx_create(C.byref(p), 0, C.byref(hnd)) == -1
    assert b"mirror-padded" in L.picsong_last_error()


@pytest.mark.parametrize("w,h", [(32, 32), (33, 40), (64, 32), (127, 65)])
def test_pad_frame_smallest_accepted_frames_match_oracle(oracle, w, h):
    L = pa.load()
    aw, ah = pa.pad_dim(w), pa.pad_dim(h)
    img = oracle.gen_frame(w, h)
    out = np.empty((ah, aw), np.uint8)
    assert L.picsong_pad_frame_host(img.ctypes.data_as(C.c_void_p), w, h, out.ctypes.data_as(C.c_void_p), aw, ah) == 0
    assert np.array_equal(out, oracle.pad_frame(img))
    assert np.array_equal(out[:h, w:], img[:, ::-1][:, :aw - w])          # edge-inclusive mirror
    assert np.array_equal(out[h:], out[:h][::-1][:ah - h])


def test_header_field_widths_are_enforced():
    """height is 16 bits, frames 17 bits, samples 32 bits in the 9-short header
    (BitStreamBuilder.cpp:54-93): a value that would spill into its neighbour is refused."""
    L = pa.load()
    hnd = C.c_void_p()
    out = (C.c_uint16 * 9)()
    for kw in (dict(width=64, height=70000), dict(width=64, height=64, frames=1 << 17), dict(width=70000, height=65535)):
        p = pa.make_params(kw["width"], kw["height"], wl=1, frames=kw.get("frames", 0))
        assert L.picsong_ctx_create(C.byref(p), 0, C.byref(hnd)) == -1, kw
        assert L.picsong_header_pack(C.byref(p), out) == -1, kw
    p = pa.make_params(64, 65535, wl=1, frames=(1 << 17) - 1)
    assert L.picsong_header_pack(C.byref(p), out) == 0
    q = pa.header_unpack(np.frombuffer(out, np.uint16))
    assert (q.width, q.height, q.frames) == (64, 65535, (1 << 17) - 1)


def test_invalid_parameters_are_rejected_without_exit():
    L = pa.load()
    h = C.c_void_p()
    for kw in (dict(wl=0), dict(wl=8), dict(width=0)):
        p = pa.make_params(kw.get("width", 512), 512, wl=kw.get("wl", 3))
        assert L.picsong_ctx_create(C.byref(p), 0, C.byref(h)) == -1
    p = pa.make_params(512, 512, wl=3)
    p.cp = 4
    assert L.picsong_ctx_create(C.byref(p), 0, C.byref(h)) == -1
    p = pa.make_params(512, 512, wl=3, k=0.5)
    p.cp = 3                                               # -cp 3 has no complexity-scalable mode
    assert L.picsong_ctx_create(C.byref(p), 0, C.byref(h)) == -1
    p = pa.make_params(64, 64, wl=6)
    assert L.picsong_ctx_create(C.byref(p), 0, C.byref(h)) == -1        # too small for 6 levels (2x2 at the last)


def test_copy_last_totals_rejects_null_arguments():
    import ctypes as C
    L = C.CDLL(os.path.join(ROOT, "cuda-image-and-video-codec_amd", "csrc", "libpicsong_hip.so"))
    L.picsong_copy_last_totals.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    assert L.picsong_copy_last_totals(None, None, 1, None) != 0


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    L = pa.load()
    h = C.c_void_p()
    p = pa.make_params(512, 512, wl=3)
    assert L.picsong_ctx_create(C.byref(p), 0, C.byref(h)) == -6       # PICSONG_ERR_NODEVICE
    assert b"no CPU path" in L.picsong_last_error()
    with pytest.raises(RuntimeError):
        pa.Codec(512, 512, wl=3)
