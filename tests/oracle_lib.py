"""ctypes binding of the CPU oracle (oracle/picsong_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke() as the checker.  The product never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
SO_PATH = os.path.join(ORACLE_DIR, "_build", "libpicsong_oracle.so")
LUT_DIR = os.path.join(ROOT, "tests", "golden", "lut")


def build(force=False):
    src = [os.path.join(ORACLE_DIR, f) for f in ("picsong_oracle.c", "picsong_oracle.h", "Makefile")]
    stale = (not os.path.exists(SO_PATH)) or any(
        os.path.getmtime(s) > os.path.getmtime(SO_PATH) for s in src)
    if force or stale:
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"] + (["-B"] if force else []))
    return SO_PATH


class PoLut(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "n_bitplanes", "n_subbands", "ctx_ref", "ctx_sign", "ctx_sig", "precision", "n_files",
        "n_bp_files", "wl", "n_ref", "n_sig", "n_sign")] + [("table", C.POINTER(C.c_int32)),
                                                              ("n_tables", C.c_int), ("cp", C.c_int)]


class PoHeader(C.Structure):
    _fields_ = [("n_samples", C.c_uint32)] + [(n, C.c_int) for n in (
        "cp", "cb_height", "cb_width", "wl", "bit_depth", "lossy", "qs_1e4", "components",
        "is_rgb", "height", "endianess", "bps", "is_signed", "frames", "k_1e3")]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    try:
        L = C.CDLL(SO_PATH)
    except OSError:
        build(force=True)
        L = C.CDLL(SO_PATH)
    vp, i32, sz, f32 = C.c_void_p, C.c_int, C.c_size_t, C.c_float
    L.po_pad_dim.restype = i32
    L.po_pad_dim.argtypes = [i32]
    L.po_pad_frame.argtypes = [vp, i32, i32, vp, i32, i32]
    L.po_pad_frame.restype = i32
    L.po_gen_frame.argtypes = [vp, i32, i32, C.c_uint32, C.c_uint32]
    L.po_level_shift_fwd_i32.argtypes = [vp, vp, sz, i32]
    L.po_level_shift_fwd_f32.argtypes = [vp, vp, sz, i32]
    L.po_level_shift_inv_i32.argtypes = [vp, sz, i32]
    L.po_level_shift_inv_f32.argtypes = [vp, sz, i32]
    L.po_dwt_extra.restype = sz
    L.po_dwt_extra.argtypes = [i32, i32, i32]
    L.po_dwt53_forward.argtypes = [vp, vp, i32, i32, i32]
    L.po_dwt53_inverse.argtypes = [vp, vp, i32, i32, i32]
    L.po_dwt97_forward.argtypes = [vp, vp, i32, i32, i32, f32]
    L.po_dwt97_inverse.argtypes = [vp, vp, i32, i32, i32, f32]
    L.po_lut_load.restype = i32
    L.po_lut_load.argtypes = [C.c_char_p, i32, i32, i32, C.POINTER(PoLut)]
    L.po_lut_load_k.restype = i32
    L.po_lut_load_k.argtypes = [C.c_char_p, i32, i32, i32, i32, C.POINTER(PoLut)]
    L.po_lut_load_cp.argtypes = [C.c_char_p, i32, i32, i32, i32, C.POINTER(PoLut)]
    L.po_consecutive_bitplanes.restype = i32
    L.po_consecutive_bitplanes.argtypes = [i32, f32, i32, i32, i32]
    L.po_lut_free.argtypes = [C.POINTER(PoLut)]
    L.po_find_subband.argtypes = [i32, i32, i32, i32, i32, C.POINTER(i32), C.POINTER(i32)]
    L.po_bpc_encode.argtypes = [vp, i32, i32, i32, i32, C.POINTER(PoLut), vp, vp]
    L.po_bpc_decode.argtypes = [vp, vp, i32, i32, i32, C.POINTER(PoLut), vp]
    L.po_bpc_encode_k.argtypes = [vp, i32, i32, i32, i32, C.POINTER(PoLut), f32, vp, vp]
    L.po_bpc_decode_k.argtypes = [vp, vp, i32, i32, i32, C.POINTER(PoLut), f32, vp]
    L.po_bpc_encode_block_uniform.restype = i32
    L.po_bpc_encode_block_uniform.argtypes = [vp, i32, i32, i32, C.POINTER(PoLut), vp]
    L.po_header_pack.argtypes = [C.POINTER(PoHeader), vp]
    L.po_header_unpack.argtypes = [vp, C.POINTER(PoHeader)]
    L.po_bitstream_total.restype = sz
    L.po_bitstream_total.argtypes = [vp, i32]
    L.po_bitstream_pack.restype = sz
    L.po_bitstream_pack.argtypes = [vp, vp, i32, vp, vp]
    L.po_bitstream_unpack.argtypes = [vp, i32, vp, vp]
    L.po_encode_frame.restype = sz
    L.po_encode_frame.argtypes = [vp, i32, i32, i32, i32, f32, C.POINTER(PoLut), i32, i32, vp]
    L.po_decode_frame.restype = i32
    L.po_decode_frame.argtypes = [vp, i32, i32, i32, i32, f32, C.POINTER(PoLut), vp]
    L.po_encode_frame_k.restype = sz
    L.po_encode_frame_k.argtypes = [vp, i32, i32, i32, i32, f32, f32, C.POINTER(PoLut), i32, i32, vp]
    L.po_decode_frame_k.restype = i32
    L.po_decode_frame_k.argtypes = [vp, i32, i32, i32, i32, f32, f32, C.POINTER(PoLut), vp]
    for fn in (L.po_rct_forward, L.po_rct_inverse, L.po_ict_forward, L.po_ict_inverse):
        fn.argtypes = [vp, vp, vp, vp, vp, vp, sz, i32]
    L.po_set_threads.argtypes = [i32]
    L.po_get_threads.restype = i32
    L.po_max_threads.restype = i32
    _lib = L
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def set_threads(n):
    lib().po_set_threads(int(n))


def max_threads():
    return lib().po_max_threads()


def host_cores():
    """CPU cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def usable_threads():
    """Threads worth giving the oracle's OpenMP loops on this box."""
    return max(1, min(max_threads(), host_cores()))


class Lut:
    """Loaded LUT (keeps the C struct alive; .table is a numpy copy)."""

    def __init__(self, folder, wl, component=1, fill=0, n_tables=1, cp=2):
        """n_tables = 1: file _0 (k = 0); n_tables = 0: every bit-plane file (k > 0); cp = 3: the five
        sections of the 3-coding-pass mode (file _0 of ref, sig, sign, cp_sig, cp_sign)."""
        self.c = PoLut()
        f = folder if folder.endswith("/") else folder + "/"
        if cp == 3:
            rc = lib().po_lut_load_cp(f.encode(), component, wl, fill, 3, C.byref(self.c))
        else:
            rc = lib().po_lut_load_k(f.encode(), component, wl, fill, n_tables, C.byref(self.c))
        if rc != 0:
            raise RuntimeError(f"po_lut_load({folder}, cp={cp}) failed: {rc}")
        self.cp = cp
        self.total = self.c.n_ref + (2 if cp == 3 else 1) * (self.c.n_sig + self.c.n_sign)      # one table
        self.n_tables = self.c.n_tables
        self.table = np.ctypeslib.as_array(self.c.table, shape=(self.total * self.n_tables,)).copy()
        self.wl = wl

    def geometry(self):
        c = self.c
        return dict(n_bitplanes=c.n_bitplanes, n_subbands=c.n_subbands, ctx_ref=c.ctx_ref,
                    ctx_sign=c.ctx_sign, ctx_sig=c.ctx_sig, precision=c.precision,
                    n_ref=c.n_ref, n_sig=c.n_sig, n_sign=c.n_sign)

    def __del__(self):
        try:
            lib().po_lut_free(C.byref(self.c))
        except Exception:
            pass


def lut_for(lossy, wl, fill=0):
    return Lut(os.path.join(LUT_DIR, "n1_lossy" if lossy else "n1_lossless"), wl, 1, fill)


LUT_CP3_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lut_cp3")


def lut_for_cp3(lossy, wl, fill=0):
    """Tables of the 3-coding-pass mode (tests/golden/make_cp3_tables.py)."""
    return Lut(os.path.join(LUT_CP3_DIR, "n1_lossy" if lossy else "n1_lossless"), wl, 1, fill, cp=3)


def lut_for_k(lossy, wl, fill=0):
    """All bit-plane tables (files _0 .. _14) of the R component: the -k > 0 layout."""
    return Lut(os.path.join(LUT_DIR, "n1_lossy" if lossy else "n1_lossless"), wl, 1, fill, n_tables=0)


def consecutive_bitplanes(msb, k, level, sb, wl):
    return lib().po_consecutive_bitplanes(msb, k, level, sb, wl)


def pad_dim(v):
    return lib().po_pad_dim(v)


def gen_frame(W, H, frame=0, seed=0x5EED1234):
    out = np.empty((H, W), np.uint8)
    lib().po_gen_frame(_p(out), W, H, frame, seed)
    return out


def pad_frame(img):
    H, W = img.shape
    AW, AH = pad_dim(W), pad_dim(H)
    out = np.empty((AH, AW), np.uint8)
    img = np.ascontiguousarray(img)
    if lib().po_pad_frame(_p(img), W, H, _p(out), AW, AH) != 0:
        raise ValueError(f"pad_frame: {W}x{H} needs more added columns/rows than it has (2W < AW or 2H < AH)")
    return out


def rgb_forward(r, g, b, lossy):
    """RCT (lossless) / ICT (lossy) with the level shift fused; planes in, 3 component planes out."""
    r, g, b = (np.ascontiguousarray(x, np.uint8) for x in (r, g, b))
    outs = [np.empty(r.shape, np.float32 if lossy else np.int32) for _ in range(3)]
    (lib().po_ict_forward if lossy else lib().po_rct_forward)(_p(r), _p(g), _p(b), _p(outs[0]), _p(outs[1]),
                                                              _p(outs[2]), r.size, 8)
    return outs


def rgb_inverse(c0, c1, c2):
    c0, c1, c2 = (np.ascontiguousarray(x) for x in (c0, c1, c2))
    lossy = c0.dtype == np.float32
    outs = [np.empty(c0.shape, np.uint8) for _ in range(3)]
    (lib().po_ict_inverse if lossy else lib().po_rct_inverse)(_p(c0), _p(c1), _p(c2), _p(outs[0]), _p(outs[1]),
                                                              _p(outs[2]), c0.size, 8)
    return outs


def lut_for_component(lossy, wl, component, fill=0, k=0.0):
    """component 0/1/2 -> the R/G/B table files of the fixture folder (k > 0: every bit-plane file of the component)."""
    return Lut(os.path.join(LUT_DIR, "n1_lossy" if lossy else "n1_lossless"), wl, component + 1, fill,
               n_tables=0 if k > 0 else 1)


def encode_plane(plane, wl, lossy, qs, lut, header, k=0.0):
    """One component: DWT + BPC + pack of an already transformed / shifted padded plane."""
    AH, AW = plane.shape
    f = dwt_forward(np.ascontiguousarray(plane), wl, qs)
    st, sz = bpc_encode(f[:AW * AH].reshape(AH, AW), wl, lut, k=k)
    return bitstream_pack(st, sz, header)


def decode_plane(stream, AW, AH, wl, lossy, qs, lut, k=0.0):
    """Inverse of encode_plane up to (not including) clamping: returns the (AH, AW) component."""
    st, sz = bitstream_unpack(stream, (AW // 64) * (AH // 64))
    coef = bpc_decode(st, sz, AW, AH, wl, lut, k=k)
    out, extra = dwt_inverse(coef, wl, lossy, qs)
    return out[extra:].reshape(AH, AW)


def level_shift_fwd(u8, lossy):
    u8 = np.ascontiguousarray(u8)
    out = np.empty(u8.shape, np.float32 if lossy else np.int32)
    (lib().po_level_shift_fwd_f32 if lossy else lib().po_level_shift_fwd_i32)(
        _p(u8), _p(out), u8.size, 8)
    return out


def level_shift_inv(x):
    x = np.ascontiguousarray(x).copy()
    if x.dtype == np.float32:
        lib().po_level_shift_inv_f32(_p(x), x.size, 8)
    else:
        lib().po_level_shift_inv_i32(_p(x), x.size, 8)
    return x


def dwt_extra(AW, AH, wl):
    return lib().po_dwt_extra(AW, AH, wl)


def dwt_forward(x, wl, qs=1.0):
    """x: (AH, AW) int32 (5/3) or float32 (9/7).  Returns flat array of P+extra elements."""
    AH, AW = x.shape
    x = np.ascontiguousarray(x)
    out = np.zeros(AW * AH + dwt_extra(AW, AH, wl), x.dtype)
    if x.dtype == np.int32:
        lib().po_dwt53_forward(_p(x), _p(out), AW, AH, wl)
    else:
        lib().po_dwt97_forward(_p(x), _p(out), AW, AH, wl, qs)
    return out


def dwt_inverse(coef, wl, lossy, qs=1.0):
    """coef: (AH, AW) int32 Mallat.  Returns (flat P+extra buffer, extra)."""
    AH, AW = coef.shape
    coef = np.ascontiguousarray(coef, np.int32)
    extra = dwt_extra(AW, AH, wl)
    out = np.zeros(AW * AH + extra, np.float32 if lossy else np.int32)
    if lossy:
        lib().po_dwt97_inverse(_p(coef), _p(out), AW, AH, wl, qs)
    else:
        lib().po_dwt53_inverse(_p(coef), _p(out), AW, AH, wl)
    return out, extra


def bpc_encode(coef, wl, lut, k=0.0):
    AH, AW = coef.shape
    coef = np.ascontiguousarray(coef)
    is_float = int(coef.dtype == np.float32)
    if not is_float:
        coef = coef.astype(np.int32, copy=False)
    staging = np.empty(AW * AH, np.int32)
    sizes = np.empty((AW // 64) * (AH // 64), np.int32)
    lib().po_bpc_encode_k(_p(coef), is_float, AW, AH, wl, C.byref(lut.c), k, _p(staging), _p(sizes))
    return staging, sizes


def bpc_decode(staging, sizes, AW, AH, wl, lut, k=0.0):
    staging = np.ascontiguousarray(staging, np.int32)
    sizes = np.ascontiguousarray(sizes, np.int32)
    coef = np.empty((AH, AW), np.int32)
    lib().po_bpc_decode_k(_p(staging), _p(sizes), AW, AH, wl, C.byref(lut.c), k, _p(coef))
    return coef


def bpc_encode_block_uniform(block, level, sb, wl, lut):
    block = np.ascontiguousarray(block, np.int32)
    st = np.empty(4096, np.int32)
    n = lib().po_bpc_encode_block_uniform(_p(block), level, sb, wl, C.byref(lut.c), _p(st))
    return st, n


def header_pack(**kw):
    h = PoHeader()
    for k, v in kw.items():
        setattr(h, k, v)
    out = np.zeros(9, np.uint16)
    lib().po_header_pack(C.byref(h), _p(out))
    return out


def header_unpack(shorts):
    shorts = np.ascontiguousarray(shorts, np.uint16)
    h = PoHeader()
    lib().po_header_unpack(_p(shorts), C.byref(h))
    return {n: getattr(h, n) for n, _ in PoHeader._fields_}


def bitstream_pack(staging, sizes, header=None):
    staging = np.ascontiguousarray(staging, np.int32)
    sizes = np.ascontiguousarray(sizes, np.int32)
    n_cb = sizes.size
    total = lib().po_bitstream_total(_p(sizes), n_cb)
    out = np.empty(total, np.uint16)
    hp = _p(np.ascontiguousarray(header, np.uint16)) if header is not None else None
    lib().po_bitstream_pack(_p(staging), _p(sizes), n_cb, hp, _p(out))
    return out


def bitstream_unpack(stream, n_cb):
    stream = np.ascontiguousarray(stream, np.uint16)
    staging = np.empty(n_cb * 4096, np.int32)
    sizes = np.empty(n_cb, np.int32)
    lib().po_bitstream_unpack(_p(stream), n_cb, _p(staging), _p(sizes))
    return staging, sizes


def encode_frame(img, wl, lossy, qs, lut, iter_=0, frames=0, k=0.0):
    H, W = img.shape
    img = np.ascontiguousarray(img, np.uint8)
    AW, AH = pad_dim(W), pad_dim(H)
    out = np.empty(9 + 2 * (AW // 64) * (AH // 64) + AW * AH + 1, np.uint16)
    n = lib().po_encode_frame_k(_p(img), W, H, wl, int(lossy), qs, k, C.byref(lut.c), iter_, frames,
                                _p(out))
    return out[:n].copy()


def decode_frame(stream, W, H, wl, lossy, qs, lut, k=0.0):
    stream = np.ascontiguousarray(stream, np.uint16)
    out = np.empty((H, W), np.uint8)
    lib().po_decode_frame_k(_p(stream), W, H, wl, int(lossy), qs, k, C.byref(lut.c), _p(out))
    return out
