"""ctypes binding of the CPU wave emulator build of the product kernels (tests/hipemu).
TEST INFRASTRUCTURE ONLY -- lets `-m "not gpu"` tests run the real kernel sources on the CPU."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_DIR = os.path.join(ROOT, "tests", "hipemu")
SO_PATH = os.path.join(EMU_DIR, "_build", "libpicsong_emu.so")
# PICSONG_EMU_SO: run the emulated tests on another build of the same sources (tools/sanitize_emu.sh:
# -fsanitize=undefined / address builds, with the sanitizer runtime preloaded)
SO_OVERRIDE = os.environ.get("PICSONG_EMU_SO")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if SO_OVERRIDE:
            _lib = C.CDLL(SO_OVERRIDE)
            _lib.emu_pack.restype = C.c_int
            return _lib
        subprocess.check_call(["make", "-C", EMU_DIR, "-s"])
        _lib = C.CDLL(SO_PATH)
        _lib.emu_pack.restype = C.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


C16 = False


def set_c16(on):
    """The encode frame paths' form: coded coefficients as int16 between the transform and the coder (DwtFwdArgs::c16).
    dwt_forward then returns a buffer whose first AW*AH int16 are the Mallat array (`mallat16`) and bpc_encode takes an
    int16 array."""
    global C16
    C16 = bool(on)
    lib().emu_set_c16(int(C16))


def mallat16(buf, AW, AH):
    return buf.reshape(-1).view(np.int16)[:AW * AH].reshape(AH, AW)


def coef16_ok(lossy, wl, qs, in_max, AW, AH):
    return bool(lib().emu_coef16_ok(int(lossy), wl, C.c_float(qs), in_max, AW, AH))


def _geo(lut):
    g = lut.geometry()
    return np.array([g["n_bitplanes"], g["n_subbands"], g["ctx_ref"], g["ctx_sign"], g["ctx_sig"],
                     g["precision"], g["n_ref"], g["n_sig"], g["n_sign"]], np.int32)


def aligned_zeros(n, dtype, align=64):
    """numpy buffer whose data pointer is `align`-byte aligned (the vector kernels need 16)."""
    raw = np.zeros(n * np.dtype(dtype).itemsize + align, np.uint8)
    off = (-raw.ctypes.data) % align
    return raw[off:off + n * np.dtype(dtype).itemsize].view(dtype)


def aligned_copy(x):
    out = aligned_zeros(x.size, x.dtype).reshape(x.shape)
    out[...] = x
    return out


def dwt_vec_levels(AW, AH, wl):
    """Levels (forward + inverse plans) that run the vector-only kernel instantiations."""
    a = aligned_zeros(AW * AH * 2, np.int32)
    b = aligned_zeros(AW * AH * 2, np.int32)
    return lib().emu_dwt_vec_levels(_p(a), _p(b), AW, AH, wl)


def dwt_forward(x, wl, lossy, qs=1.0, extra=0):
    """x: (AH, AW) uint8 (fused level shift) / int32 / float32."""
    AH, AW = x.shape
    x = aligned_copy(np.ascontiguousarray(x))
    out = aligned_zeros(AW * AH + extra, np.float32 if lossy else np.int32)
    dwt_forward.fused01 = bool(lib().emu_dwt_forward(_p(x), int(x.dtype == np.uint8), _p(out), AW, AH, wl,
                                                     int(lossy), C.c_float(qs)))
    return out


def dwt_forward_rgb(r, g, b, wl, extra, lossy=False, qs=1.0):
    """The three components' int16 Mallat arrays of an RGB frame, the colour transform (RCT; lossy: ICT) in the fused
    head's load stage; None when that form does not apply to the geometry."""
    AH, AW = r.shape
    r, g, b = (aligned_copy(np.ascontiguousarray(x, np.uint8)) for x in (r, g, b))
    stride = (AW * AH + extra) * 4
    out = aligned_zeros(3 * (AW * AH + extra), np.int32)
    if not lib().emu_dwt_forward_rgb(_p(r), _p(g), _p(b), _p(out), C.c_size_t(stride), AW, AH, wl, int(lossy), C.c_float(qs)):
        return None
    return [mallat16(out[k * (AW * AH + extra):(k + 1) * (AW * AH + extra)], AW, AH).copy() for k in range(3)]


def dwt_forward_band(x_u8, out, wl, lossy, qs, row0, rows):
    """Level 0 of the input rows [row0, row0 + rows) of the (AH, AW) uint8 frame into `out` (Mallat + scratch)."""
    AH, AW = x_u8.shape
    assert x_u8.dtype == np.uint8 and x_u8.flags["C_CONTIGUOUS"]
    lib().emu_dwt_forward_band(_p(x_u8), _p(out), AW, AH, wl, int(lossy), C.c_float(qs), row0, rows)


def dwt_forward_tail(out, AW, AH, wl, lossy, qs=1.0):
    lib().emu_dwt_forward_tail(_p(out), AW, AH, wl, int(lossy), C.c_float(qs))


def bpc_encode_range(coef, wl, lut, cb_begin, cb_count):
    """Staging / sizes of the whole frame with only the codeblocks [cb_begin, +cb_count) coded."""
    AH, AW = coef.shape
    coef = np.ascontiguousarray(coef)
    staging = np.full(AW * AH, -1, np.int32)
    sizes = np.zeros((AW // 64) * (AH // 64), np.int32)
    flag = np.zeros(1, np.int32)
    tab = np.ascontiguousarray(lut.table, np.int32)
    geo = _geo(lut)
    lib().emu_bpc_encode_range(_p(coef), int(coef.dtype == np.float32), AW, AH, wl, _p(tab), _p(geo),
                               _p(staging), _p(sizes), _p(flag), cb_begin, cb_count)
    return staging, sizes, int(flag[0])


def dwt_inverse(coef, wl, lossy, qs=1.0, extra=0):
    AH, AW = coef.shape
    coef = aligned_copy(np.ascontiguousarray(coef, np.int32))
    out = aligned_zeros(AW * AH + extra, np.float32 if lossy else np.int32)
    lib().emu_dwt_inverse(_p(coef), _p(out), AW, AH, wl, int(lossy), C.c_float(qs))
    return out


def dwt_inverse_u8(coef, wl, lossy, qs=1.0, extra=0):
    """Frame-path inverse: returns (pixels u8 (AH, AW), fused flag)."""
    AH, AW = coef.shape
    coef = aligned_copy(np.ascontiguousarray(coef, np.int32))
    scratch = aligned_zeros(AW * AH + extra, np.float32 if lossy else np.int32)
    pix = aligned_zeros(AW * AH, np.uint8)
    fused = lib().emu_dwt_inverse_u8(_p(coef), _p(scratch), _p(pix), AW, AH, wl, int(lossy), C.c_float(qs))
    return pix.reshape(AH, AW), bool(fused)


def dwt_inverse_u8_c16(coef16, wl, lossy, qs=1.0, extra=0):
    """The decode frame paths with 16-bit coefficients: `coef16` an int16 (AH, AW) Mallat array.  Returns (pixels u8
    (AH, AW), flags): bit 0 the finest level wrote the pixels, bit 1 synthesis levels 1 and 0 ran as one launch
    (dwt_inv2_kernel), bit 2 the 16-bit form applied at all (0: nothing was run)."""
    AH, AW = coef16.shape
    coef16 = aligned_copy(np.ascontiguousarray(coef16, np.int16))
    scratch = aligned_zeros(AW * AH + extra, np.float32 if lossy else np.int32)
    pix = aligned_zeros(AW * AH, np.uint8)
    flags = lib().emu_dwt_inverse_u8_c16(_p(coef16), _p(scratch), _p(pix), AW, AH, wl, int(lossy), C.c_float(qs))
    return pix.reshape(AH, AW), int(flags)


def dwt_inverse_rgb(coefs16, wl, extra, lossy=False, qs=1.0):
    """picsong_decode_rgb_frame's lossless synthesis: `coefs16` three int16 (AH, AW) Mallat arrays (Y, Cb, Cr); the
    finest level of the three components and the inverse colour transform run as one launch.  Returns the three pixel
    planes, or None when the form does not apply."""
    AH, AW = coefs16[0].shape
    P = AW * AH
    inp = aligned_zeros(3 * P, np.int16)
    for c in range(3):
        inp[c * P:(c + 1) * P] = np.ascontiguousarray(coefs16[c], np.int16).ravel()
    scratch = aligned_zeros(3 * (P + extra), np.float32 if lossy else np.int32)
    out = [aligned_zeros(P, np.uint8) for _ in range(3)]
    ok = lib().emu_dwt_inverse_rgb(_p(inp), C.c_size_t(P * 2), _p(scratch), C.c_size_t((P + extra) * 4), _p(out[0]), _p(out[1]),
                                   _p(out[2]), AW, AH, wl, int(lossy), C.c_float(qs))
    return [o.reshape(AH, AW) for o in out] if ok else None


def bpc_decode_stream16(stream, AW, AH, wl, lut):
    """bpc_decode_stream with the coefficients leaving as int16 (the decoder's C16 instantiation)."""
    stream = np.ascontiguousarray(stream, np.uint16)
    coef = np.full((AH, AW), 0x5A5A, np.int16)
    flag = np.zeros(1, np.int32)
    tab = np.ascontiguousarray(lut.table, np.int32)
    geo = _geo(lut)
    bad = lib().emu_bpc_decode_stream16(_p(stream), int(stream.size), AW, AH, wl, _p(tab), _p(geo), _p(coef), _p(flag))
    bpc_decode_stream16.last_flag = int(flag[0])
    bpc_decode_stream16.last_bad = int(bad)
    return coef


def level_shift_inv(x):
    x = np.ascontiguousarray(x).copy()
    lib().emu_level_shift_inv(_p(x), C.c_size_t(x.size), int(x.dtype == np.float32))
    return x


def clamp_to_u8(x):
    x = np.ascontiguousarray(x)
    out = np.empty(x.shape, np.uint8)
    lib().emu_clamp_to_u8(_p(x), _p(out), C.c_size_t(x.size), int(x.dtype == np.float32))
    return out


def rgb_forward(r, g, b, lossy):
    r, g, b = (np.ascontiguousarray(x) for x in (r, g, b))
    outs = [np.empty(r.shape, np.float32 if lossy else np.int32) for _ in range(3)]
    lib().emu_rgb_forward(_p(r), _p(g), _p(b), _p(outs[0]), _p(outs[1]), _p(outs[2]), C.c_size_t(r.size), int(lossy))
    return outs


def rgb_inverse(c0, c1, c2):
    c0, c1, c2 = (np.ascontiguousarray(x) for x in (c0, c1, c2))
    outs = [np.empty(c0.shape, np.uint8) for _ in range(3)]
    lib().emu_rgb_inverse(_p(c0), _p(c1), _p(c2), _p(outs[0]), _p(outs[1]), _p(outs[2]), C.c_size_t(c0.size),
                          int(c0.dtype == np.float32))
    return outs


def level_shift_fwd(u8, lossy):
    u8 = np.ascontiguousarray(u8)
    out = np.empty(u8.shape, np.float32 if lossy else np.int32)
    lib().emu_level_shift_fwd(_p(u8), _p(out), C.c_size_t(u8.size), int(lossy))
    return out


def bpc_encode(coef, wl, lut, k=0.0):
    AH, AW = coef.shape
    coef = np.ascontiguousarray(coef)
    staging = np.empty(AW * AH, np.int32)
    sizes = np.empty((AW // 64) * (AH // 64), np.int32)
    flag = np.zeros(1, np.int32)
    tab = np.ascontiguousarray(lut.table, np.int32)
    geo = _geo(lut)
    lib().emu_bpc_encode(_p(coef), int(coef.dtype == np.float32), AW, AH, wl, _p(tab), _p(geo),
                         _p(staging), _p(sizes), _p(flag), C.c_float(k), int(getattr(lut, "n_tables", 1)))
    return staging, sizes, int(flag[0])


def bpc3_encode(coef, wl, lut):
    """-cp 3 (lut: oracle_lib.lut_for_cp3)."""
    AH, AW = coef.shape
    coef = np.ascontiguousarray(coef)
    staging = np.empty(AW * AH, np.int32)
    sizes = np.empty((AW // 64) * (AH // 64), np.int32)
    flag = np.zeros(1, np.int32)
    tab = np.ascontiguousarray(lut.table, np.int32)
    geo = _geo(lut)
    lib().emu_bpc3_encode(_p(coef), int(coef.dtype == np.float32), AW, AH, wl, _p(tab), _p(geo), _p(staging),
                          _p(sizes), _p(flag))
    return staging, sizes, int(flag[0])


def bpc3_decode(staging, sizes, AW, AH, wl, lut):
    staging = np.ascontiguousarray(staging, np.int32)
    sizes = np.ascontiguousarray(sizes, np.int32)
    coef = np.empty((AH, AW), np.int32)
    flag = np.zeros(1, np.int32)
    tab = np.ascontiguousarray(lut.table, np.int32)
    geo = _geo(lut)
    lib().emu_bpc3_decode(_p(staging), _p(sizes), AW, AH, wl, _p(tab), _p(geo), _p(coef), _p(flag))
    return coef


def bpc_decode(staging, sizes, AW, AH, wl, lut, k=0.0):
    staging = np.ascontiguousarray(staging, np.int32)
    sizes = np.ascontiguousarray(sizes, np.int32)
    coef = np.empty((AH, AW), np.int32)
    flag = np.zeros(1, np.int32)
    tab = np.ascontiguousarray(lut.table, np.int32)
    geo = _geo(lut)
    lib().emu_bpc_decode(_p(staging), _p(sizes), AW, AH, wl, _p(tab), _p(geo), _p(coef), _p(flag),
                         C.c_float(k), int(getattr(lut, "n_tables", 1)))
    bpc_decode.last_flag = int(flag[0])
    return coef


def bpc_decode_stream(stream, AW, AH, wl, lut):
    """The frame paths' decoder: lengths, offsets and codewords straight from the packed stream (no unpack, no staging)."""
    stream = np.ascontiguousarray(stream, np.uint16)
    coef = np.empty((AH, AW), np.int32)
    flag = np.zeros(1, np.int32)
    tab = np.ascontiguousarray(lut.table, np.int32)
    geo = _geo(lut)
    bad = lib().emu_bpc_decode_stream(_p(stream), int(stream.size), AW, AH, wl, _p(tab), _p(geo), _p(coef), _p(flag))
    bpc_decode_stream.last_flag = int(flag[0])
    bpc_decode_stream.last_bad = int(bad)
    return coef


def bpc_decode_stream_k(stream, AW, AH, wl, lut, k, c16=False):
    """-k > 0 straight from the packed stream (the frame paths' decoder of a complexity-scalable context); c16: into an
    int16 Mallat array (the C16 instantiations)."""
    stream = np.ascontiguousarray(stream, np.uint16)
    coef = np.empty((AH, AW), np.int16 if c16 else np.int32)
    flag = np.zeros(1, np.int32)
    tab = np.ascontiguousarray(lut.table, np.int32)
    geo = _geo(lut)
    bad = lib().emu_bpc_decode_stream_k(_p(stream), int(stream.size), AW, AH, wl, _p(tab), _p(geo), _p(coef), _p(flag),
                                        C.c_float(k), int(getattr(lut, "n_tables", 1)), int(c16))
    assert int(bad) == 0
    return coef


def pack(staging, sizes, header=None):
    staging = np.ascontiguousarray(staging, np.int32)
    sizes = np.ascontiguousarray(sizes, np.int32)
    n = sizes.size
    out = np.zeros(9 + 2 * n + int(sizes.sum()) + 1, np.uint16)
    hp = _p(np.ascontiguousarray(header, np.uint16)) if header is not None else None
    total = lib().emu_pack(_p(staging), _p(sizes), n, hp, _p(out))
    return out[:total].copy()


def pack16(staging16, sizes, header=None):
    """pack_kernel over the encoders' 16-bit staging (the frame paths' form)."""
    staging16 = np.ascontiguousarray(staging16, np.uint16)
    sizes = np.ascontiguousarray(sizes, np.int32)
    n = sizes.size
    out = np.zeros(9 + 2 * n + int(sizes.sum()) + 1, np.uint16)
    hp = _p(np.ascontiguousarray(header, np.uint16)) if header is not None else None
    total = lib().emu_pack16(_p(staging16), _p(sizes), n, hp, _p(out))
    return out[:total].copy()


def unpack(stream, n_cb):
    stream = np.ascontiguousarray(stream, np.uint16)
    staging = np.empty(n_cb * 4096, np.int32)
    sizes = np.empty(n_cb, np.int32)
    unpack.last_flag = int(lib().emu_unpack(_p(stream), n_cb, _p(staging), _p(sizes)))
    return staging, sizes
