"""Thin ctypes binding of libpicsong_hip.so (include/picsong_hip.h) for tests and bench.py.

PyTorch is used only as plumbing: device memory (torch tensors -> raw pointers) and the HIP stream.
There is NO fallback: if the shared library is missing or no GPU is present this module raises.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
SO_PATH = os.environ.get("PICSONG_SO", os.path.join(PKG, "csrc", "libpicsong_hip.so"))

PICSONG_OK = 0


class Params(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("wl", C.c_int), ("cp", C.c_int),
                ("lossy", C.c_int), ("qs", C.c_float), ("k", C.c_float), ("cb_width", C.c_int),
                ("cb_height", C.c_int), ("bit_depth", C.c_int), ("frames", C.c_int),
                ("components", C.c_int), ("is_rgb", C.c_int)]


class LutInfo(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("n_bitplanes", "n_subbands", "ctx_ref", "ctx_sign", "ctx_sig",
                                       "precision", "n_files", "n_bp_files", "n_ref", "n_sig", "n_sign",
                                       "n_tables", "cp")]


EXPORTS = [
    "picsong_last_error", "picsong_version", "picsong_pad_dim", "picsong_dwt_extra",
    "picsong_max_stream_shorts", "picsong_header_pack", "picsong_header_unpack", "picsong_lut_load",
    "picsong_lut_load_k",
    "picsong_ctx_create", "picsong_ctx_destroy", "picsong_ctx_set_lut", "picsong_ctx_padded_dims",
    "picsong_ctx_set_pipelined",
    "picsong_level_shift_fwd", "picsong_level_shift_inv", "picsong_dwt_forward", "picsong_dwt_inverse",
    "picsong_dwt_forward_u8", "picsong_bpc_encode", "picsong_bpc_decode", "picsong_bitstream_pack",
    "picsong_bitstream_unpack", "picsong_last_total", "picsong_encode_frame", "picsong_decode_frame",
    "picsong_pad_frame_host", "picsong_range_flag", "picsong_profile_begin", "picsong_profile_read",
    "picsong_encode_frame_stripe", "picsong_ctx_set_lut_component", "picsong_rgb_forward", "picsong_rgb_inverse",
    "picsong_encode_plane", "picsong_decode_plane",
    "picsong_ctx_set_lut_device", "picsong_bpc_encode_component", "picsong_bpc_decode_component",
    "picsong_encode_frames", "picsong_last_totals", "picsong_selftest_lds_order",
    "picsong_dwt_forward_band", "picsong_dwt_forward_tail", "picsong_encode_stripe_coded", "picsong_lut_load_cp",
    "picsong_copy_last_totals", "picsong_decode_frames", "picsong_encode_rgb_frame", "picsong_decode_rgb_frame",
]

_lib = None


def load():
    """Load the HIP library; raises (never falls back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: its wheel bundles a HIP runtime that it exposes process-wide; loading it before
    # libpicsong_hip.so makes the library bind to that same runtime, so the device pointers and
    # hipStream_t handles torch hands out are valid inside the library (one runtime per process).
    import torch  # noqa: F401
    if not os.path.exists(SO_PATH):
        raise RuntimeError(f"{SO_PATH} not built: run __graft_entry__.build() (hipcc, gfx950). "
                           "There is no CPU fallback.")
    L = C.CDLL(SO_PATH)
    vp, i = C.c_void_p, C.c_int
    L.picsong_last_error.restype = C.c_char_p
    L.picsong_version.restype = C.c_char_p
    L.picsong_pad_dim.argtypes = [i]
    L.picsong_dwt_extra.restype = C.c_size_t
    L.picsong_dwt_extra.argtypes = [i, i, i]
    L.picsong_max_stream_shorts.restype = C.c_size_t
    L.picsong_max_stream_shorts.argtypes = [i, i]
    L.picsong_header_pack.argtypes = [C.POINTER(Params), vp]
    L.picsong_header_unpack.argtypes = [vp, C.POINTER(Params)]
    L.picsong_lut_load.argtypes = [C.c_char_p, i, i, i, C.POINTER(LutInfo), vp, C.c_size_t]
    L.picsong_lut_load_k.argtypes = [C.c_char_p, i, i, i, i, C.POINTER(LutInfo), vp, C.c_size_t]
    L.picsong_ctx_create.argtypes = [C.POINTER(Params), i, C.POINTER(vp)]
    L.picsong_ctx_destroy.argtypes = [vp]
    L.picsong_ctx_destroy.restype = None
    L.picsong_ctx_set_lut.argtypes = [vp, C.POINTER(LutInfo), vp]
    L.picsong_ctx_padded_dims.argtypes = [vp, C.POINTER(i), C.POINTER(i), C.POINTER(i)]
    L.picsong_ctx_set_pipelined.argtypes = [vp, i]
    L.picsong_level_shift_fwd.argtypes = [vp, vp, vp, vp]
    L.picsong_level_shift_inv.argtypes = [vp, vp, vp]
    L.picsong_dwt_forward.argtypes = [vp, vp, vp, vp]
    L.picsong_dwt_forward_u8.argtypes = [vp, vp, vp, vp]
    L.picsong_dwt_inverse.argtypes = [vp, vp, vp, vp]
    L.picsong_bpc_encode.argtypes = [vp, vp, vp, vp, vp]
    L.picsong_bpc_decode.argtypes = [vp, vp, vp, vp, vp]
    L.picsong_bitstream_pack.argtypes = [vp, vp, vp, vp, vp, C.POINTER(i), vp]
    L.picsong_bitstream_unpack.argtypes = [vp, vp, vp, vp, vp]
    L.picsong_last_total.argtypes = [vp, vp, C.POINTER(i)]
    L.picsong_encode_frame.argtypes = [vp, vp, i, vp, vp]
    L.picsong_decode_frame.argtypes = [vp, vp, vp, vp]
    L.picsong_encode_frame_stripe.argtypes = [vp, vp, i, i, vp, vp]
    L.picsong_ctx_set_lut_component.argtypes = [vp, i, C.POINTER(LutInfo), vp]
    L.picsong_rgb_forward.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
    L.picsong_rgb_inverse.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
    L.picsong_encode_plane.argtypes = [vp, vp, i, i, vp, vp]
    L.picsong_decode_plane.argtypes = [vp, vp, i, vp, vp]
    L.picsong_pad_frame_host.argtypes = [vp, i, i, vp, i, i]
    L.picsong_range_flag.argtypes = [vp, vp, C.POINTER(i)]
    L.picsong_profile_begin.argtypes = [vp, i]
    L.picsong_profile_read.argtypes = [vp, C.POINTER(i), vp, i]
    if hasattr(L, "picsong_encode_frames"):          # (A/B runs load older variant libraries through PICSONG_SO)
        L.picsong_encode_frames.argtypes = [vp, i, vp, C.c_size_t, i, vp, C.c_size_t, vp]
        L.picsong_last_totals.argtypes = [vp, vp, i, C.POINTER(i)]
        L.picsong_ctx_set_lut_device.argtypes = [vp, i, C.POINTER(LutInfo), vp]
        L.picsong_bpc_encode_component.argtypes = [vp, i, vp, vp, vp, vp]
        L.picsong_bpc_decode_component.argtypes = [vp, i, vp, vp, vp, vp]
    if hasattr(L, "picsong_copy_last_totals"):
        L.picsong_copy_last_totals.argtypes = [vp, vp, i, vp]
    if hasattr(L, "picsong_decode_frames"):
        L.picsong_decode_frames.argtypes = [vp, i, vp, C.c_size_t, vp, C.c_size_t, vp]
    if hasattr(L, "picsong_encode_rgb_frame"):
        L.picsong_encode_rgb_frame.argtypes = [vp, vp, vp, vp, i, vp, C.c_size_t, vp]
        L.picsong_decode_rgb_frame.argtypes = [vp, vp, C.c_size_t, vp, vp, vp, vp]
    if hasattr(L, "picsong_lut_load_cp"):
        L.picsong_lut_load_cp.argtypes = [C.c_char_p, i, i, i, i, C.POINTER(LutInfo), vp, C.c_size_t]
    if hasattr(L, "picsong_dwt_forward_band"):
        L.picsong_dwt_forward_band.argtypes = [vp, vp, i, i, vp, vp]
        L.picsong_dwt_forward_tail.argtypes = [vp, vp, vp]
        L.picsong_encode_stripe_coded.argtypes = [vp, vp, i, i, vp, vp]
    _lib = L
    return L


class PicsongError(RuntimeError):
    pass


def _check(rc):
    if rc != PICSONG_OK:
        raise PicsongError(f"picsong error {rc}: {load().picsong_last_error().decode()}")


def pad_dim(v):
    return load().picsong_pad_dim(v)


def dwt_extra(aw, ah, wl):
    return load().picsong_dwt_extra(aw, ah, wl)


def lut_load(folder, wl, component=1, fill=0, n_tables=1, cp=2):
    """Returns (LutInfo, np.int32 table) parsed by the library's own host parser.
    n_tables = 1: file _0 (k = 0); n_tables = 0: every bit-plane file (the -k > 0 layout);
    cp = 3: the five sections of the 3-coding-pass mode."""
    L = load()
    info = LutInfo()
    if cp == 3:
        _check(L.picsong_lut_load_cp(folder.encode(), component, wl, fill, 3, C.byref(info), None, 0))
        table = np.empty(info.n_ref + 2 * (info.n_sig + info.n_sign), np.int32)
        _check(L.picsong_lut_load_cp(folder.encode(), component, wl, fill, 3, C.byref(info),
                                     table.ctypes.data_as(C.c_void_p), table.size))
        return info, table
    _check(L.picsong_lut_load_k(folder.encode(), component, wl, fill, n_tables, C.byref(info), None, 0))
    table = np.empty((info.n_ref + info.n_sig + info.n_sign) * info.n_tables, np.int32)
    _check(L.picsong_lut_load_k(folder.encode(), component, wl, fill, n_tables, C.byref(info),
                                table.ctypes.data_as(C.c_void_p), table.size))
    return info, table


def make_params(width, height, wl=5, lossy=False, qs=1.0, frames=0, rgb=False, k=0.0, cp=2):
    return Params(width=width, height=height, wl=wl, cp=cp, lossy=int(lossy), qs=qs, k=k,
                  cb_width=64, cb_height=18, bit_depth=8, frames=frames, components=3 if rgb else 1,
                  is_rgb=int(rgb))


def header_pack(params):
    out = np.zeros(9, np.uint16)
    _check(load().picsong_header_pack(C.byref(params), out.ctypes.data_as(C.c_void_p)))
    return out


def header_unpack(shorts):
    shorts = np.ascontiguousarray(shorts, np.uint16)
    p = Params()
    _check(load().picsong_header_unpack(shorts.ctypes.data_as(C.c_void_p), C.byref(p)))
    return p


class Codec:
    """One coding context (== the reference's per-frame DWT<T,Y> / BPCCuda<T> / BitStreamBuilder
    objects + the LUT upload of Engine::initLUT).  All tensor arguments are torch CUDA tensors."""

    def __init__(self, width, height, wl=5, lossy=False, qs=1.0, lut_folder=None, lut_fill=0,
                 device=0, frames=0, rgb=False, k=0.0, pipelined=False, cp=2):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("no GPU visible: the picsong HIP path has no CPU fallback")
        self.torch = torch
        self.L = load()
        self.params = make_params(width, height, wl, lossy, qs, frames, rgb, k, cp)
        self.rgb = bool(rgb)
        self.device = device
        h = C.c_void_p()
        _check(self.L.picsong_ctx_create(C.byref(self.params), device, C.byref(h)))
        self.h = h
        if pipelined:                 # frames of other contexts share the GPU: throughput over latency
            _check(self.L.picsong_ctx_set_pipelined(self.h, 1))
        aw, ah, ncb = C.c_int(), C.c_int(), C.c_int()
        _check(self.L.picsong_ctx_padded_dims(self.h, C.byref(aw), C.byref(ah), C.byref(ncb)))
        self.aw, self.ah, self.ncb = aw.value, ah.value, ncb.value
        self.P = self.aw * self.ah
        self.extra = dwt_extra(self.aw, self.ah, wl)
        self.lossy = bool(lossy)
        self.dtype = torch.float32 if lossy else torch.int32
        self.dev = torch.device("cuda", device)
        if lut_folder is not None:
            for comp in range(3 if rgb else 1):
                # files ...R/G/B.txt_0 (k = 0) or every bit-plane file ...txt_0.._14 (k > 0)
                info, table = lut_load(lut_folder, wl, comp + 1, lut_fill, 0 if k > 0 else 1, cp)
                self.set_lut(info, table, comp)

    def close(self):
        if getattr(self, "h", None):
            self.L.picsong_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.dev).cuda_stream)

    @staticmethod
    def _p(t):
        return C.c_void_p(t.data_ptr())

    def set_lut(self, info, table, component=0):
        table = np.ascontiguousarray(table, np.int32)
        _check(self.L.picsong_ctx_set_lut_component(self.h, component, C.byref(info),
                                                    table.ctypes.data_as(C.c_void_p)))

    # ---- RGB path ----
    def rgb_forward(self, r, g, b):
        outs = [self.torch.empty(self.P, dtype=self.dtype, device=self.dev) for _ in range(3)]
        _check(self.L.picsong_rgb_forward(self.h, self._p(r), self._p(g), self._p(b), self._p(outs[0]),
                                          self._p(outs[1]), self._p(outs[2]), self._stream()))
        return outs

    def rgb_inverse(self, c0, c1, c2):
        outs = [self.torch.empty(self.P, dtype=self.torch.uint8, device=self.dev) for _ in range(3)]
        _check(self.L.picsong_rgb_inverse(self.h, self._p(c0), self._p(c1), self._p(c2), self._p(outs[0]),
                                          self._p(outs[1]), self._p(outs[2]), self._stream()))
        return [o.view(self.ah, self.aw) for o in outs]

    def encode_plane(self, plane, component, with_header):
        out = self.torch.empty(self.max_stream_shorts(), dtype=self.torch.int16, device=self.dev)
        _check(self.L.picsong_encode_plane(self.h, self._p(plane), component, int(with_header), self._p(out),
                                           self._stream()))
        return out[:self.last_total()]

    def encode_rgb_frame(self, r, g, b, header_mask=1):
        """One RGB frame (padded u8 planes) through one launch per stage; returns the three codestreams."""
        out = self.torch.empty((3, self.max_stream_shorts()), dtype=self.torch.int16, device=self.dev)
        _check(self.L.picsong_encode_rgb_frame(self.h, self._p(r), self._p(g), self._p(b), header_mask, self._p(out),
                                               out.stride(0), self._stream()))
        totals = self.last_totals(3)
        return [out[k, :totals[k]] for k in range(3)]

    def decode_rgb_frame(self, streams):
        """streams: int16 [3, >= max_stream_shorts()]; returns the three padded u8 planes."""
        outs = [self.torch.empty((self.ah, self.aw), dtype=self.torch.uint8, device=self.dev) for _ in range(3)]
        _check(self.L.picsong_decode_rgb_frame(self.h, self._p(streams), streams.stride(0), self._p(outs[0]),
                                               self._p(outs[1]), self._p(outs[2]), self._stream()))
        return outs

    def decode_plane(self, stream, component):
        out = self.torch.empty(self.P + self.extra, dtype=self.dtype, device=self.dev)
        _check(self.L.picsong_decode_plane(self.h, self._p(stream), component, self._p(out), self._stream()))
        return out[self.extra:]

    # ---- stage functions (device tensors in/out) ----
    def level_shift_fwd(self, u8):
        out = self.torch.empty(self.P, dtype=self.dtype, device=self.dev)
        _check(self.L.picsong_level_shift_fwd(self.h, self._p(u8), self._p(out), self._stream()))
        return out

    def level_shift_inv(self, data):
        _check(self.L.picsong_level_shift_inv(self.h, self._p(data), self._stream()))
        return data

    def dwt_forward(self, x):
        out = self.torch.zeros(self.P + self.extra, dtype=self.dtype, device=self.dev)
        fn = self.L.picsong_dwt_forward_u8 if x.dtype == self.torch.uint8 else self.L.picsong_dwt_forward
        _check(fn(self.h, self._p(x), self._p(out), self._stream()))
        return out

    def dwt_inverse(self, coef_i32):
        out = self.torch.zeros(self.P + self.extra, dtype=self.dtype, device=self.dev)
        _check(self.L.picsong_dwt_inverse(self.h, self._p(coef_i32), self._p(out), self._stream()))
        return out

    def bpc_encode(self, coef):
        staging = self.torch.empty(self.P, dtype=self.torch.int32, device=self.dev)
        sizes = self.torch.empty(self.ncb, dtype=self.torch.int32, device=self.dev)
        _check(self.L.picsong_bpc_encode(self.h, self._p(coef), self._p(staging), self._p(sizes),
                                         self._stream()))
        return staging, sizes

    def bpc_decode(self, staging, sizes):
        coef = self.torch.empty(self.P, dtype=self.torch.int32, device=self.dev)
        _check(self.L.picsong_bpc_decode(self.h, self._p(staging), self._p(sizes), self._p(coef),
                                         self._stream()))
        return coef

    def bitstream_pack(self, staging, sizes, header=None):
        out = self.torch.empty(self.max_stream_shorts(), dtype=self.torch.int16, device=self.dev)
        total = C.c_int()
        hp = None
        if header is not None:
            header = np.ascontiguousarray(header, np.uint16)
            hp = header.ctypes.data_as(C.c_void_p)
        _check(self.L.picsong_bitstream_pack(self.h, self._p(staging), self._p(sizes), hp, self._p(out),
                                             C.byref(total), self._stream()))
        return out[:total.value]

    def bitstream_unpack(self, stream):
        staging = self.torch.empty(self.P, dtype=self.torch.int32, device=self.dev)
        sizes = self.torch.empty(self.ncb, dtype=self.torch.int32, device=self.dev)
        _check(self.L.picsong_bitstream_unpack(self.h, self._p(stream), self._p(staging), self._p(sizes),
                                               self._stream()))
        return staging, sizes

    def max_stream_shorts(self):
        return self.L.picsong_max_stream_shorts(self.aw, self.ah)

    def range_flag(self):
        f = C.c_int()
        _check(self.L.picsong_range_flag(self.h, self._stream(), C.byref(f)))
        return f.value

    def profile_begin(self, capacity):
        _check(self.L.picsong_profile_begin(self.h, capacity))

    def profile_read(self, capacity):
        """(n, 3) float32 array of {dwt_ms, bpc_ms, pack_ms} per encode_frame since profile_begin."""
        ms = np.zeros((capacity, 3), np.float32)
        n = C.c_int()
        _check(self.L.picsong_profile_read(self.h, C.byref(n), ms.ctypes.data_as(C.c_void_p), capacity))
        return ms[:n.value]

    # ---- whole frame (asynchronous; last_total() synchronises) ----
    def encode_frame_async(self, frame_u8_padded, out_stream, iter_=0):
        _check(self.L.picsong_encode_frame(self.h, self._p(frame_u8_padded), iter_, self._p(out_stream),
                                           self._stream()))

    def last_total(self):
        t = C.c_int()
        _check(self.L.picsong_last_total(self.h, self._stream(), C.byref(t)))
        return t.value

    def encode_frame(self, frame_u8_padded, iter_=0):
        out = self.torch.empty(self.max_stream_shorts(), dtype=self.torch.int16, device=self.dev)
        self.encode_frame_async(frame_u8_padded, out, iter_)
        return out[:self.last_total()]

    # ---- batched frames: n frames, one launch per stage (picsong_encode_frames) ----
    def encode_frames_async(self, frames_u8_padded, out_streams, first_iter=0):
        """frames_u8_padded: uint8 [n, AH*AW] (rows contiguous, any row stride that is a multiple of 16);
        out_streams: int16 [n, >= max_stream_shorts()]."""
        n = frames_u8_padded.shape[0]
        assert out_streams.shape[0] >= n and frames_u8_padded.stride(-1) == 1 and out_streams.stride(-1) == 1
        _check(self.L.picsong_encode_frames(self.h, n, self._p(frames_u8_padded), frames_u8_padded.stride(0),
                                            first_iter, self._p(out_streams), out_streams.stride(0), self._stream()))
        return n

    def last_totals(self, n):
        t = (C.c_int * n)()
        _check(self.L.picsong_last_totals(self.h, self._stream(), n, t))
        return list(t)

    def decode_frames(self, streams, out=None):
        """streams: int16 [n, >= max_stream_shorts()] (codestream f in row f); returns uint8 [n, AH, AW]."""
        n = streams.shape[0]
        assert streams.stride(-1) == 1
        if out is None:
            out = self.torch.empty((n, self.ah, self.aw), dtype=self.torch.uint8, device=self.dev)
        _check(self.L.picsong_decode_frames(self.h, n, self._p(streams), streams.stride(0), self._p(out), out.stride(0),
                                            self._stream()))
        return out

    def copy_last_totals(self, n, d_totals):
        """The lengths of the most recent encode_frame (n = 1) / encode_frames call into an int32 device tensor, on
        this context's stream, without a wait."""
        assert d_totals.dtype == self.torch.int32 and d_totals.numel() >= n
        _check(self.L.picsong_copy_last_totals(self.h, self._stream(), n, self._p(d_totals)))

    def encode_frames(self, frames_u8_padded, first_iter=0):
        n = frames_u8_padded.shape[0]
        out = self.torch.empty((n, self.max_stream_shorts()), dtype=self.torch.int16, device=self.dev)
        self.encode_frames_async(frames_u8_padded, out, first_iter)
        return [out[i, :t] for i, t in enumerate(self.last_totals(n))]

    def encode_frame_stripe(self, frame_u8_padded, cb_begin, cb_count):
        """Mini-stream (9 x 0xFFFF | pairs | payload | 0xFFFF) of codeblocks [cb_begin, +cb_count)."""
        out = self.torch.empty(9 + 2 * cb_count + cb_count * 4096 + 1, dtype=self.torch.int16, device=self.dev)
        _check(self.L.picsong_encode_frame_stripe(self.h, self._p(frame_u8_padded), cb_begin, cb_count,
                                                  self._p(out), self._stream()))
        return out[:self.last_total()]

    # ---- row-band sharding of the transform (intra-frame split) ----
    def new_coef_buffer(self):
        return self.torch.zeros(self.P + self.extra, dtype=self.torch.float32 if self.lossy else self.torch.int32,
                                device=self.dev)

    def dwt_forward_band(self, frame_u8_frame_coords, row0, rows, coef):
        _check(self.L.picsong_dwt_forward_band(self.h, self._p(frame_u8_frame_coords), row0, rows, self._p(coef),
                                               self._stream()))

    def dwt_forward_tail(self, coef):
        _check(self.L.picsong_dwt_forward_tail(self.h, self._p(coef), self._stream()))

    def encode_stripe_coded(self, coef, cb_begin, cb_count):
        out = self.torch.empty(9 + 2 * cb_count + cb_count * 4096 + 1, dtype=self.torch.int16, device=self.dev)
        _check(self.L.picsong_encode_stripe_coded(self.h, self._p(coef), cb_begin, cb_count, self._p(out),
                                                  self._stream()))
        return out[:self.last_total()]

    def decode_frame(self, stream):
        out = self.torch.empty(self.P, dtype=self.torch.uint8, device=self.dev)
        _check(self.L.picsong_decode_frame(self.h, self._p(stream), self._p(out), self._stream()))
        return out.view(self.ah, self.aw)
