"""The product's kernel SOURCES (cuda-image-and-video-codec_amd/csrc/*_kernels.hpp) executed on the
CPU wave emulator (tests/hipemu) and compared bit-for-bit with the oracle.  This is kernel-logic
coverage for the no-GPU suite; the `-m gpu` tests repeat the same comparisons through the C ABI on
a real MI355X."""
import os

import numpy as np
import pytest

import emu_lib as emu


@pytest.fixture(scope="module")
def E():
    emu.lib()
    return emu


def _coef(oracle, img, wl, lossy, qs=1.0):
    x = oracle.level_shift_fwd(oracle.pad_frame(img), lossy)
    f = oracle.dwt_forward(x, wl, qs)
    return f[:x.size].reshape(x.shape)


@pytest.fixture(params=["vec", "novec"])
def dwt_path(request, monkeypatch):
    """Run the DWT tests through both kernel families: vector-only instantiations (the default for
    every real frame) and the per-column ones (PICSONG_DWT_NOVEC=1, odd geometries)."""
    if request.param == "novec":
        monkeypatch.setenv("PICSONG_DWT_NOVEC", "1")
    else:
        monkeypatch.delenv("PICSONG_DWT_NOVEC", raising=False)
    return request.param


def test_dwt_vector_kernels_are_selected(E, monkeypatch):
    monkeypatch.delenv("PICSONG_DWT_NOVEC", raising=False)
    assert E.dwt_vec_levels(320, 192, 3) == 6          # W = 320, 160, 80: all multiples of 4
    assert E.dwt_vec_levels(64, 64, 5) == 10           # down to W = 4
    assert E.dwt_vec_levels(64, 64, 6) == 10           # W = 2 at the sixth level: per-column kernel
    monkeypatch.setenv("PICSONG_DWT_NOVEC", "1")
    assert E.dwt_vec_levels(320, 192, 3) == 0


@pytest.mark.parametrize("W,H,wl", [(320, 192, 3), (512, 64, 1), (128, 128, 2), (768, 128, 2), (64, 64, 5)])
def test_dwt53_forward_inverse_bit_exact(oracle, E, dwt_path, W, H, wl):
    img = oracle.gen_frame(W, H, 3)
    x = oracle.level_shift_fwd(img, False)
    extra = oracle.dwt_extra(W, H, wl)
    ref = oracle.dwt_forward(x, wl)
    assert np.array_equal(E.dwt_forward(img, wl, False, extra=extra)[:W * H], ref[:W * H])   # fused u8
    assert np.array_equal(E.dwt_forward(x, wl, False, extra=extra)[:W * H], ref[:W * H])
    coef = ref[:W * H].reshape(H, W)
    ri, ex = oracle.dwt_inverse(coef, wl, False)
    gi = E.dwt_inverse(coef, wl, False, extra=extra)
    assert np.array_equal(gi[extra:], ri[ex:])
    assert np.array_equal(gi[extra:].reshape(H, W), x)


@pytest.mark.parametrize("W,H,wl,qs", [(320, 192, 3, 0.5), (256, 128, 2, 1.0), (576, 64, 1, 0.25), (64, 128, 5, 0.5)])
def test_dwt97_forward_inverse_bit_exact(oracle, E, dwt_path, W, H, wl, qs):
    img = oracle.gen_frame(W, H, 5)
    xf = oracle.level_shift_fwd(img, True)
    extra = oracle.dwt_extra(W, H, wl)
    ref = oracle.dwt_forward(xf, wl, qs)
    got = E.dwt_forward(img, wl, True, qs, extra=extra)
    assert np.array_equal(got[:W * H].view(np.uint32), ref[:W * H].view(np.uint32))
    got2 = E.dwt_forward(xf, wl, True, qs, extra=extra)
    assert np.array_equal(got2[:W * H].view(np.uint32), ref[:W * H].view(np.uint32))
    q = np.trunc(ref[:W * H]).astype(np.int32).reshape(H, W)
    ri, ex = oracle.dwt_inverse(q, wl, True, qs)
    gi = E.dwt_inverse(q, wl, True, qs, extra=extra)
    assert np.array_equal(gi[extra:].view(np.uint32), ri[ex:].view(np.uint32))
    assert np.array_equal(E.level_shift_inv(gi[extra:]), oracle.level_shift_inv(ri[ex:]))


@pytest.mark.parametrize("W,H,wl,lossy,qs", [(320, 192, 3, False, 1.0), (256, 128, 2, True, 0.5), (64, 64, 1, False, 1.0)])
def test_dwt_inverse_fused_pixel_output(oracle, E, W, H, wl, lossy, qs):
    """Finest inverse level with level shift + clamp + u8 conversion fused == inverse, shift, clamp."""
    rng = np.random.default_rng(17)
    img = rng.integers(0, 256, (H, W), dtype=np.uint8)
    x = oracle.level_shift_fwd(img, lossy)
    extra = oracle.dwt_extra(W, H, wl)
    f = oracle.dwt_forward(x, wl, qs) if lossy else oracle.dwt_forward(x, wl)
    coef = f[:W * H].astype(np.int32).reshape(H, W)
    coef[3, 5] += 4000                                   # force clamping on both ends
    coef[9, 70 % W] -= 4000
    ref, ex = oracle.dwt_inverse(coef, wl, lossy, qs)
    want = oracle.level_shift_inv(ref[ex:]).reshape(H, W).astype(np.uint8)
    got, fused = E.dwt_inverse_u8(coef, wl, lossy, qs, extra=extra)
    assert fused and np.array_equal(got, want)


def test_dwt_odd_half_width_scalar_paths(oracle, E):
    """(AW >> l) / 2 odd at the last level: 8-byte vector stores are illegal, scalar path used."""
    W, H, wl = 192, 64, 2            # level 1: 96x32, half width 48 (even); level... use wl where odd
    W, H, wl = 320, 64, 3            # level 2: 80x16, half 40; fine -- exercise W%4 != 0 instead:
    W, H, wl = 448, 64, 3            # level 2: 112x16 -> half 56 even; level input 224, 112
    for (W, H, wl) in ((448, 64, 3), (192, 192, 4), (832, 64, 4)):   # 832>>3 = 104, >>4 = 52 -> half 26
        img = oracle.gen_frame(W, H, 1)
        x = oracle.level_shift_fwd(img, False)
        extra = oracle.dwt_extra(W, H, wl)
        ref = oracle.dwt_forward(x, wl)
        got = E.dwt_forward(img, wl, False, extra=extra)
        assert np.array_equal(got[:W * H], ref[:W * H])
        gi = E.dwt_inverse(ref[:W * H].reshape(H, W), wl, False, extra=extra)
        assert np.array_equal(gi[extra:].reshape(H, W), x)


def test_level_shift_kernels(oracle, E):
    img = oracle.gen_frame(256, 64, 2)
    for lossy in (False, True):
        assert np.array_equal(E.level_shift_fwd(img, lossy), oracle.level_shift_fwd(img, lossy))
    rng = np.random.default_rng(0)
    xi = rng.integers(-300, 300, 4096).astype(np.int32)
    assert np.array_equal(E.level_shift_inv(xi), oracle.level_shift_inv(xi))
    xf = (rng.standard_normal(4096) * 150).astype(np.float32)
    xf[:8] = [0.49, 0.5, 1.5, -128.51, 126.49, 126.5, 127.4, -0.5]
    assert np.array_equal(E.level_shift_inv(xf), oracle.level_shift_inv(xf))
    # fused clamp + u8 conversion of the frame path
    assert np.array_equal(E.clamp_to_u8(xi), oracle.level_shift_inv(xi).astype(np.uint8))
    assert np.array_equal(E.clamp_to_u8(xf), oracle.level_shift_inv(xf).astype(np.uint8))


@pytest.mark.parametrize("W,H,wl", [(192, 128, 2), (128, 64, 1)])
def test_bpc_encode_decode_bit_exact(oracle, E, W, H, wl):
    lut = oracle.lut_for(False, wl)
    coef = _coef(oracle, oracle.gen_frame(W, H), wl, False)
    st_ref, sz_ref = oracle.bpc_encode(coef, wl, lut)
    st, sz, flag = E.bpc_encode(coef, wl, lut)
    assert flag == 0
    assert np.array_equal(sz, sz_ref) and np.array_equal(st, st_ref)
    assert np.array_equal(E.bpc_decode(st_ref, sz_ref, W, H, wl, lut), coef)


def test_bpc_subband_straddle_and_odd_codeblock_count(oracle, E):
    """AW = 192, wl = 2: boundary 192>>2 = 48 falls inside codeblock column 0, so lanes of one
    codeblock use different LUT rows (per-lane subband); 3x1 codeblocks = odd count,
    the upper half of the last wave idles."""
    W, H, wl = 192, 64, 2
    lut = oracle.lut_for(False, wl)
    coef = _coef(oracle, oracle.gen_frame(W, H, 7), wl, False)
    st_ref, sz_ref = oracle.bpc_encode(coef, wl, lut)
    st, sz, _ = E.bpc_encode(coef, wl, lut)
    assert np.array_equal(sz, sz_ref) and np.array_equal(st, st_ref)
    assert np.array_equal(E.bpc_decode(st_ref, sz_ref, W, H, wl, lut), coef)


def test_bpc_float_input_truncates_toward_zero(oracle, E):
    W, H, wl, qs = 128, 64, 1, 0.5
    lut = oracle.lut_for(True, wl)
    coef = _coef(oracle, oracle.gen_frame(W, H, 2), wl, True, qs)
    assert coef.dtype == np.float32
    st_ref, sz_ref = oracle.bpc_encode(coef, wl, lut)
    st, sz, _ = E.bpc_encode(coef, wl, lut)
    assert np.array_equal(sz, sz_ref) and np.array_equal(st, st_ref)
    assert np.array_equal(E.bpc_decode(st, sz, W, H, wl, lut), np.trunc(coef).astype(np.int32))


def test_bpc_raw_fallback_zero_block_and_mixed_msb(oracle, E):
    """One wave holds a noise block (raw fallback, size 4096) next to an all-zero block (MSB 32);
    the second wave pairs MSB 0 with MSB 12."""
    rng = np.random.default_rng(4)
    coef = np.zeros((64, 256), np.int32)
    coef[:, 0:64] = rng.integers(-30000, 30000, (64, 64))
    coef[:, 128:192] = rng.integers(-1, 2, (64, 64))
    coef[:, 192:256] = (rng.standard_normal((64, 64)) * 600).astype(np.int32)
    lut = oracle.lut_for(False, 1)
    st_ref, sz_ref = oracle.bpc_encode(coef, 1, lut)
    assert sz_ref[0] == 4096 and sz_ref[1] == 1
    st, sz, flag = E.bpc_encode(coef, 1, lut)
    assert flag == 0 and np.array_equal(sz, sz_ref) and np.array_equal(st, st_ref)
    assert np.array_equal(E.bpc_decode(st_ref, sz_ref, 256, 64, 1, lut), coef)


def test_bpc_zero_probability_lut_holes(oracle, E):
    """wl = 6 leaves LUT groups unwritten (de-facto p = 0): every coded 0 costs a
    whole codeword.  Exercised directly with an all-zero LUT."""
    lut = oracle.lut_for(False, 1)
    lut.table[:] = 0
    import ctypes as C
    C.memmove(lut.c.table, lut.table.ctypes.data, lut.table.nbytes)
    rng = np.random.default_rng(5)
    coef = rng.integers(-3, 4, (64, 64)).astype(np.int32)
    st_ref, sz_ref = oracle.bpc_encode(coef, 1, lut)
    st, sz, _ = E.bpc_encode(coef, 1, lut)
    assert np.array_equal(sz, sz_ref) and np.array_equal(st, st_ref)
    assert np.array_equal(E.bpc_decode(st_ref, sz_ref, 64, 64, 1, lut), coef)


def test_bpc_range_flag(oracle, E):
    coef = np.zeros((64, 64), np.int32)
    coef[3, 3] = 1 << 17
    _, _, flag = E.bpc_encode(coef, 1, oracle.lut_for(False, 1))
    assert flag == 1


def test_pack_unpack_kernels(oracle, E):
    rng = np.random.default_rng(6)
    n_cb = 37
    sizes = rng.integers(1, 900, n_cb).astype(np.int32)
    sizes[5] = 1
    sizes[9] = 4096
    staging = np.full(n_cb * 4096, -1, np.int32)
    for cb in range(n_cb):
        staging[cb * 4096: cb * 4096 + sizes[cb]] = rng.integers(0, 65536, sizes[cb])
    hdr = oracle.header_pack(n_samples=64 * 64 * n_cb, cp=2, cb_height=18, cb_width=64, wl=1,
                             bit_depth=8, lossy=0, qs_1e4=10000, components=1, is_rgb=0, height=64,
                             endianess=0, bps=8, is_signed=0, frames=0, k_1e3=0)
    for h in (hdr, None):
        ref = oracle.bitstream_pack(staging, sizes, h)
        got = E.pack(staging, sizes, h)
        assert np.array_equal(got, ref)
        # the frame paths' form: the encoders' 16-bit staging (what lies beyond a codeblock's length is never read)
        assert np.array_equal(E.pack16(staging.astype(np.uint16), sizes, h), ref)
    st2, sz2 = E.unpack(ref, n_cb)
    assert np.array_equal(sz2, sizes) and np.array_equal(st2, staging)


@pytest.mark.parametrize("W,H,wl", [(192, 128, 2), (192, 64, 2), (128, 64, 1)])
def test_decoder_reads_the_packed_stream_itself(oracle, E, W, H, wl):
    """The frame paths' decoder (scan_stream_kernel + bpc_decode_kernel<false, NP, true>): lengths, offsets and
    codewords straight from the packed stream, which is exactly as long as its total -- the ring's loads ahead of a
    codeblock's length are kept inside it.  (192 x 64: an odd codeblock count, the last wave's upper half idles.)"""
    lut = oracle.lut_for(False, wl)
    coef = _coef(oracle, oracle.gen_frame(W, H, 3), wl, False)
    st, sz = oracle.bpc_encode(coef, wl, lut)
    stream = oracle.bitstream_pack(st, sz)
    assert stream.size == 9 + 2 * sz.size + int((sz - 1).sum()) + 1
    got = E.bpc_decode_stream(stream, W, H, wl, lut)
    assert E.bpc_decode_stream.last_bad == 0 and E.bpc_decode_stream.last_flag == 0
    assert np.array_equal(got, coef)
    assert np.array_equal(got, E.bpc_decode(st, sz, W, H, wl, lut))


def test_stream_decoder_raw_blocks_and_damaged_lengths(oracle, E):
    """A raw codeblock (size 4096: its word 0 travels in the MSB's place), an all-zero block and two coded ones read
    from the stream; then the same stream with a damaged length: flagged, clamped, and decoded without reading
    outside the buffer (the wrapper hands the kernel the buffer's exact length)."""
    rng = np.random.default_rng(4)
    coef = np.zeros((64, 256), np.int32)
    coef[:, 0:64] = rng.integers(-30000, 30000, (64, 64))
    coef[:, 128:192] = rng.integers(-1, 2, (64, 64))
    coef[:, 192:256] = (rng.standard_normal((64, 64)) * 600).astype(np.int32)
    lut = oracle.lut_for(False, 1)
    st, sz = oracle.bpc_encode(coef, 1, lut)
    assert sz[0] == 4096 and sz[1] == 1
    stream = oracle.bitstream_pack(st, sz)
    assert np.array_equal(E.bpc_decode_stream(stream, 256, 64, 1, lut), coef)
    assert E.bpc_decode_stream.last_bad == 0
    bad = stream.copy()
    bad[10 + 2 * 2] = 60000                               # block 2 claims 60000 words
    E.bpc_decode_stream(bad, 256, 64, 1, lut)
    assert E.bpc_decode_stream.last_bad == 1


def test_unpack_clamps_damaged_lengths(oracle, E):
    """A codeblock length outside 1..4096 is clamped and flagged: nothing is written outside the
    codeblock's own 4096 staging words."""
    n_cb = 3
    st = np.full(n_cb * 4096, -1, np.int32)
    sz = np.array([5, 9, 3], np.int32)
    for cb in range(n_cb):
        st[cb * 4096] = 7
        st[cb * 4096 + 1:cb * 4096 + sz[cb]] = np.arange(1, sz[cb]) + 100 * cb
    stream = np.concatenate([oracle.bitstream_pack(st, sz), np.zeros(3 * 4096, np.uint16)])
    good, gsz = E.unpack(stream, n_cb)
    assert E.unpack.last_flag == 0 and np.array_equal(gsz, sz)
    bad = stream.copy()
    bad[10 + 2 * 1] = 60000                               # block 1 claims 60000 words
    bad[10 + 2 * 2] = 0                                   # block 2 claims none
    got, bsz = E.unpack(bad, n_cb)
    assert E.unpack.last_flag == 1 and bsz.tolist() == [5, 4096, 1]
    assert np.array_equal(got[:4096][:5], good[:5])       # block 0 untouched by its neighbours


def test_whole_frame_codestream_identical_to_oracle(oracle, E):
    """u8 frame -> fused DWT -> BPC -> pack on the emulated kernels == oracle codestream, and the
    emulated decode path returns the input."""
    W, H, wl = 192, 128, 2
    img = oracle.gen_frame(W, H, 11)
    lut = oracle.lut_for(False, wl)
    ref = oracle.encode_frame(img, wl, False, 1.0, lut)
    extra = oracle.dwt_extra(W, H, wl)
    coef = E.dwt_forward(img, wl, False, extra=extra)[:W * H].reshape(H, W)
    st, sz, _ = E.bpc_encode(coef, wl, lut)
    got = E.pack(st, sz, ref[:9])
    assert np.array_equal(got, ref)
    st2, sz2 = E.unpack(got, sz.size)
    c2 = E.bpc_decode(st2, sz2, W, H, wl, lut)
    out = E.level_shift_inv(E.dwt_inverse(c2, wl, False, extra=extra)[extra:]).reshape(H, W)
    assert np.array_equal(out.astype(np.uint8), img)


@pytest.mark.parametrize("W,H,wl,lossy,qs", [(192, 128, 2, False, 1.0), (320, 192, 3, True, 0.5), (256, 256, 5, False, 1.0),
                                              (512, 320, 6, True, 0.5), (128, 64, 1, True, 2.0), (256, 128, 2, True, 0.5)])
def test_frame_path_with_16_bit_coefficients(oracle, E, monkeypatch, W, H, wl, lossy, qs):
    """The encode frame paths' compact form (DwtFwdArgs::c16): the transform writes its coded subbands as int16 -- the
    value the coder's load makes of a coefficient anyway -- and the coder reads them: the codestream is the oracle's
    (fused head and per-level kernels), and it decodes to the oracle's pixels."""
    assert E.coef16_ok(lossy, wl, qs, 128, W, H)
    img = oracle.gen_frame(W, H, 13)
    lut = oracle.lut_for(lossy, wl)
    ref = oracle.encode_frame(img, wl, lossy, qs, lut)
    extra = oracle.dwt_extra(W, H, wl)
    ref_pix = oracle.decode_frame(ref, W, H, wl, lossy, qs, lut)
    E.set_c16(True)
    try:
        for nofuse in ("0", "1"):
            monkeypatch.setenv("PICSONG_DWT_NOFUSE01", nofuse)
            buf = E.dwt_forward(img, wl, lossy, qs, extra=extra)
            coef = E.mallat16(buf, W, H)
            ref_coef = oracle.dwt_forward(oracle.level_shift_fwd(img, lossy), wl, qs)[:W * H].reshape(H, W)
            assert np.array_equal(coef.astype(np.int32), ref_coef.astype(np.int32))       # (astype truncates toward zero)
            st, sz, flag = E.bpc_encode(coef, wl, lut)
            assert flag == 0
            got = E.pack(st, sz, ref[:9])
            assert np.array_equal(got, ref), f"nofuse={nofuse}"
    finally:
        E.set_c16(False)
    st2, sz2 = E.unpack(got, sz.size)
    c2 = E.bpc_decode(st2, sz2, W, H, wl, lut)
    pix, fused = E.dwt_inverse_u8(c2, wl, lossy, qs, extra=extra)
    assert fused and np.array_equal(pix, ref_pix)


@pytest.mark.parametrize("W,H,wl,lossy,qs", [(320, 192, 3, False, 1.0), (256, 128, 2, False, 1.0), (512, 320, 5, False, 1.0),
                                             (320, 192, 3, True, 0.5), (512, 320, 5, True, 1.0), (256, 64, 2, True, 0.3)])
def test_rgb_colour_transform_in_the_fused_heads_load_stage(oracle, E, W, H, wl, lossy, qs):
    """dwt_fwd2_kernel<..., RGB>: the head reads the R, G, B planes and delivers component blockIdx.z of the colour
    transform (RCT on the 5/3 head, ICT on the 9/7 one; the rows of three iterations in flight, 32-row bands, the
    three components of a tile re-indexed onto one XCD) -- the three components' coefficients equal the oracle's colour
    transform + transform (+ quantisation, truncated as the coder reads them) of each component."""
    planes = [oracle.pad_frame(oracle.gen_frame(W, H, 80 + c)) for c in range(3)]
    planes[1][:9, :13] = 255; planes[2][-7:, -5:] = 0              # extremes of the chroma range at two corners
    got = E.dwt_forward_rgb(*planes, wl, oracle.dwt_extra(W, H, wl), lossy, qs)
    assert got is not None
    comps = oracle.rgb_forward(*planes, lossy)
    for k in range(3):
        ref = (oracle.dwt_forward(comps[k], wl, qs) if lossy else oracle.dwt_forward(comps[k], wl))[:W * H].reshape(H, W)
        assert np.array_equal(got[k].astype(np.int32), np.trunc(ref).astype(np.int32)), f"component {k}"


@pytest.mark.parametrize("W,H,wl", [(320, 192, 3), (256, 128, 2), (512, 320, 5), (200, 100, 2)])
def test_rgb_inverse_colour_transform_in_the_finest_synthesis_level(oracle, E, W, H, wl):
    """dwt_inv_rgb_kernel: the finest 5/3 synthesis level of the three components in one wave, the inverse RCT, the level
    shift and the clamp at its stores -- the planes of the frame, from the coefficients the forward side makes of them
    (extremes of the chroma range included), and the oracle's pixels for coefficients that overshoot (a coarse
    perturbation: the clamp at both ends)."""
    planes = [oracle.pad_frame(oracle.gen_frame(W, H, 90 + c)) for c in range(3)]
    planes[0][:11, :7] = 255; planes[1][:11, :7] = 0; planes[2][-9:, -6:] = 255; planes[0][-9:, -6:] = 0
    AH, AW = planes[0].shape
    comps = oracle.rgb_forward(*planes, False)
    coefs = [oracle.dwt_forward(comps[k], wl)[:AW * AH].reshape(AH, AW).astype(np.int16) for k in range(3)]
    got = E.dwt_inverse_rgb(coefs, wl, oracle.dwt_extra(W, H, wl))
    assert got is not None
    for k in range(3):
        assert np.array_equal(got[k], planes[k]), f"plane {k}"
    rng = np.random.default_rng(3)
    noisy = [c + (rng.integers(-40, 41, c.shape) * (rng.integers(0, 16, c.shape) == 0)).astype(np.int16) for c in coefs]
    got = E.dwt_inverse_rgb(noisy, wl, oracle.dwt_extra(W, H, wl))
    back = []
    for k in range(3):
        inv, extra = oracle.dwt_inverse(noisy[k].astype(np.int32), wl, False, 1.0)
        back.append(inv[extra:].reshape(AH, AW))
    ref = oracle.rgb_inverse(*back)
    for k in range(3):
        assert np.array_equal(got[k], ref[k]), f"plane {k} (perturbed)"
    assert (got[0] == 0).any() and (got[0] == 255).any()


@pytest.mark.parametrize("W,H,wl,qs,replay", [(320, 192, 3, 0.5, "0"), (512, 320, 5, 1.0, "0"), (256, 128, 2, 0.3, "0"), (320, 192, 3, 0.5, "1")])
def test_rgb_inverse_ict_across_the_three_waves_of_a_workgroup(oracle, E, monkeypatch, W, H, wl, qs, replay):
    """dwt_inv97_rgb_kernel: the finest 9/7 synthesis level of the three components as the three waves of a workgroup, a
    finished row pair exchanged through LDS, each wave delivering one pixel plane of the inverse ICT -- the oracle's
    pixels (its synthesis of every component, then its inverse colour transform), the clamp at both ends included, and
    through the pass with true divisions the whole workgroup repeats (PICSONG_DWT_EXACT_REPLAY=1)."""
    monkeypatch.setenv("PICSONG_DWT_EXACT_REPLAY", replay)
    planes = [oracle.pad_frame(oracle.gen_frame(W, H, 95 + c)) for c in range(3)]
    planes[0][:11, :7] = 255; planes[1][:11, :7] = 0; planes[2][-9:, -6:] = 255; planes[0][-9:, -6:] = 0
    AH, AW = planes[0].shape
    comps = oracle.rgb_forward(*planes, True)
    coefs = [np.trunc(oracle.dwt_forward(comps[k], wl, qs)[:AW * AH].reshape(AH, AW)).astype(np.int16) for k in range(3)]
    got = E.dwt_inverse_rgb(coefs, wl, oracle.dwt_extra(W, H, wl), True, qs)
    assert got is not None
    back = []
    for k in range(3):
        inv, extra = oracle.dwt_inverse(coefs[k].astype(np.int32), wl, True, qs)
        back.append(inv[extra:].reshape(AH, AW))
    ref = oracle.rgb_inverse(*back)
    for k in range(3):
        assert np.array_equal(got[k], ref[k]), f"plane {k}"
    assert (got[0] == 0).any() and (got[0] == 255).any()


def test_16_bit_coefficient_bound(E):
    """coef16_ok: the 16-bit form only where magnitudes are bounded below 2^15 (8-bit samples; 9/7: times the
    quantisation weights): BASELINE's configurations qualify, a fine qs or a geometry off the vector kernels does not."""
    assert E.coef16_ok(False, 5, 1.0, 128, 3840, 2176) and E.coef16_ok(True, 6, 0.5, 128, 7680, 4352)
    assert E.coef16_ok(False, 5, 1.0, 255, 16384, 16384)
    assert not E.coef16_ok(True, 6, 8.0, 128, 7680, 4352)         # 882 x 17.4 x 8 > 2^15
    assert not E.coef16_ok(True, 3, 0.5, 128, 200, 128)           # a level width that is no multiple of 4


@pytest.mark.parametrize("lossy", [False, True])
def test_rgb_colour_transform_kernels(oracle, E, lossy):
    r, g, b = (oracle.gen_frame(128, 64, k) for k in (1, 2, 3))
    r[0, :8] = [0, 255, 0, 255, 128, 127, 1, 254]
    g[0, :8] = [255, 0, 0, 255, 128, 129, 3, 2]
    b[0, :8] = [0, 0, 255, 255, 128, 126, 200, 100]
    ref = oracle.rgb_forward(r, g, b, lossy)
    got = E.rgb_forward(r, g, b, lossy)
    view = np.uint32 if lossy else np.int32
    for x, y in zip(got, ref):
        assert np.array_equal(x.view(view), y.view(view))
    back_ref = oracle.rgb_inverse(*ref)
    back = E.rgb_inverse(*ref)
    for x, y in zip(back, back_ref):
        assert np.array_equal(x, y)
    if not lossy:
        for x, y in zip(back, (r, g, b)):
            assert np.array_equal(x, y)                      # RCT is reversible
    # out-of-range components clamp like the reference
    big = [np.full((64, 128), v, np.float32 if lossy else np.int32) for v in (300, -300, 50)]
    for x, y in zip(E.rgb_inverse(*big), oracle.rgb_inverse(*big)):
        assert np.array_equal(x, y)


# ---- complexity-scalable mode -k > 0: BULK kernel instantiations vs the oracle -------------------

def _bulk_coeffs(oracle, W, H, wl, lossy, seed):
    rng = np.random.default_rng(seed)
    img = np.clip(rng.normal(120, 45, (H, W)), 0, 255).astype(np.uint8)
    img[H // 4:H // 2, W // 8:W // 2] = 230
    img[:, W // 2:] = (img[:, W // 2:] // 8) * 8
    x = oracle.level_shift_fwd(img, lossy)
    f = oracle.dwt_forward(x, wl, 0.5) if lossy else oracle.dwt_forward(x, wl)
    return f[:W * H].reshape(H, W)


@pytest.mark.parametrize("W,H,wl,k,fulltab", [(256, 128, 2, 0.3, False), (256, 128, 2, 1.0, True), (128, 192, 1, 4.0, False),
                                              (192, 128, 2, 65.0, False), (128, 128, 3, 0.7, False),
                                              (128, 128, 5, 0.7, False)])      # (wl 5 at 128 x 128: a codeblock spans 13 table groups)
def test_bulk_mode_kernels_bit_exact(oracle, E, monkeypatch, W, H, wl, k, fulltab):
    """-k > 0 on the emulator against the oracle.  The kernels keep only the table GROUPS a codeblock's lanes use in LDS
    (COMPACT) where the geometry's widest codeblock fits, whole tables otherwise (the last case) or when told to
    (PICSONG_BULK_FULLTAB=1: the second)."""
    if fulltab:
        monkeypatch.setenv("PICSONG_BULK_FULLTAB", "1")
    coef = _bulk_coeffs(oracle, W, H, wl, False, 11)
    lut = oracle.lut_for_k(False, wl)
    st_o, sz_o = oracle.bpc_encode(coef, wl, lut, k=k)
    st_e, sz_e, flag = E.bpc_encode(coef, wl, lut, k=k)
    assert flag == 0
    assert np.array_equal(sz_e, sz_o)
    for cb in range(sz_o.size):
        n = sz_o[cb]
        assert np.array_equal(st_e[cb * 4096:cb * 4096 + n], st_o[cb * 4096:cb * 4096 + n]), cb
    back = E.bpc_decode(st_o, sz_o, W, H, wl, lut, k=k)
    assert np.array_equal(back, coef)
    # ... and straight from the packed stream, as the frame paths of a -k > 0 context decode (round 4; a raw codeblock,
    # whose word 0 travels in the MSB's place, included when the data makes one)
    stream = oracle.bitstream_pack(st_o, sz_o, None)
    assert np.array_equal(E.bpc_decode_stream_k(stream, W, H, wl, lut, k), coef)
    # ... and into an int16 Mallat array (the decode frame paths' 16-bit form: the bulk scan reads a row's two
    # coefficients back as one dword)
    assert np.array_equal(E.bpc_decode_stream_k(stream, W, H, wl, lut, k, c16=True).astype(np.int32), coef)


def test_bulk_mode_stream_decode_with_a_raw_codeblock(oracle, E):
    rng = np.random.default_rng(31)
    W, H, wl, k = 256, 64, 1, 0.8
    coef = _bulk_coeffs(oracle, W, H, wl, False, 5).copy()
    coef[:, 64:128] = rng.integers(-30000, 30000, (64, 64))       # no model fits noise: the expansion fallback
    lut = oracle.lut_for_k(False, wl)
    st_o, sz_o = oracle.bpc_encode(coef, wl, lut, k=k)
    assert (sz_o == 4096).any() and (sz_o < 4096).any()
    stream = oracle.bitstream_pack(st_o, sz_o, None)
    ref = oracle.bpc_decode(st_o, sz_o, W, H, wl, lut, k=k)
    assert np.array_equal(E.bpc_decode_stream_k(stream, W, H, wl, lut, k), ref)
    assert np.array_equal(E.bpc_decode_stream_k(stream, W, H, wl, lut, k, c16=True), ref.astype(np.int16))


def test_bulk_mode_kernels_float_input_and_odd_block_count(oracle, E):
    W, H, wl, k = 192, 64, 1, 0.9            # 3 codeblocks: the last wave has an idle half
    coef = _bulk_coeffs(oracle, W, H, wl, True, 3)
    lut = oracle.lut_for_k(True, wl)
    st_o, sz_o = oracle.bpc_encode(coef, wl, lut, k=k)
    st_e, sz_e, _ = E.bpc_encode(coef, wl, lut, k=k)
    assert np.array_equal(sz_e, sz_o)
    for cb in range(sz_o.size):
        n = sz_o[cb]
        assert np.array_equal(st_e[cb * 4096:cb * 4096 + n], st_o[cb * 4096:cb * 4096 + n]), cb
    assert np.array_equal(E.bpc_decode(st_o, sz_o, W, H, wl, lut, k=k), oracle.bpc_decode(st_o, sz_o, W, H, wl, lut, k=k))


def test_bulk_mode_k0_tables_unchanged(oracle, E):
    # the BULK build path is only taken for k > 0; with the multi-table LUT and k = 0 the plain
    # kernels must still see table 0
    W, H, wl = 128, 128, 2
    coef = _bulk_coeffs(oracle, W, H, wl, False, 7)
    lutk = oracle.lut_for_k(False, wl)
    st_o, sz_o = oracle.bpc_encode(coef, wl, oracle.lut_for(False, wl))
    st_e, sz_e, _ = E.bpc_encode(coef, wl, lutk, k=0.0)
    assert np.array_equal(sz_e, sz_o)


@pytest.mark.parametrize("k", [0.0, 0.4])
def test_bpc_deep_planes_beyond_the_register_file(oracle, E, k):
    """Codeblocks with MSB 8..15 (more planes than the encoder holds in registers: the lower ones take
    the HBM scratch path), paired in one wave with shallow and with empty codeblocks."""
    rng = np.random.default_rng(23)
    W, H, wl = 256, 128, 1
    peak = np.array([[6000, 5, 300, 0], [40000, 1000, 2, 200]])         # per 64x64 block; 40000 -> MSB 15
    coef = np.zeros((H, W), np.int32)
    for by in range(2):
        for bx in range(4):
            s = int(peak[by, bx])
            if not s:
                continue
            blk = rng.integers(-2, 3, (64, 64))                          # compressible: small noise ...
            ys, xs = rng.integers(0, 64, 60), rng.integers(0, 64, 60)
            blk[ys, xs] = rng.integers(-s, s + 1, 60)                    # ... plus a few deep spikes
            coef[by * 64:by * 64 + 64, bx * 64:bx * 64 + 64] = blk
    coef[70, 3] = 65535                                                  # MSB 15 exactly
    lut = oracle.lut_for_k(False, wl) if k > 0 else oracle.lut_for(False, wl)
    st_o, sz_o = oracle.bpc_encode(coef, wl, lut, k=k)
    st_e, sz_e, flag = E.bpc_encode(coef, wl, lut, k=k)
    assert flag == 0 and np.array_equal(sz_e, sz_o)
    assert (sz_o < 4096).all()                                          # none took the raw fallback
    msbs = st_o[::4096]
    coded = msbs[msbs != 32]                                             # 32 = all-zero codeblock
    assert coded.max() == 15 and (coded >= 8).sum() >= 4 and (coded < 8).sum() >= 3
    for cb in range(sz_o.size):
        n = sz_o[cb]
        assert np.array_equal(st_e[cb * 4096:cb * 4096 + n], st_o[cb * 4096:cb * 4096 + n]), cb
    assert np.array_equal(E.bpc_decode(st_o, sz_o, W, H, wl, lut, k=k), oracle.bpc_decode(st_o, sz_o, W, H, wl, lut, k=k))


# ---- 9/7 synthesis: divisions in reciprocal form -------------------------------------------------
_LIFT_DIVISORS = (1.230174104914001, 0.812893066)
_QSTEPS = (1.965908, 1.0112865, 0.52021784, 4.1224113, 1.9968134, 0.96721643, 8.416739, 4.1833673, 2.0792568,
           16.935543, 8.534108, 4.3004827, 33.924816, 17.166693, 8.686718, 67.87687, 34.385098, 17.41882)


def test_reciprocal_division_equals_division(E):
    """div_rc == `/` bit for bit: a strided sweep over every exponent the kernels send to it
    (tools/div_check.c is the exhaustive version), and the de-quantisation domain check itself."""
    import ctypes as C
    L = E.lib()
    L.emu_div_mismatches.restype = C.c_long
    L.emu_div_mismatches.argtypes = [C.c_float, C.c_uint, C.c_uint, C.c_uint]
    L.emu_dequant_fast_ok.argtypes = [C.c_float, C.c_int]
    lo, hi = (127 - 96) << 23, ((127 + 100) << 23) - 1
    for c in _LIFT_DIVISORS:
        assert L.emu_div_mismatches(c, lo, hi, 1009) == 0
        assert L.emu_div_mismatches(c, lo | 0x80000000, hi | 0x80000000, 4099) == 0       # negative x
    for c in _QSTEPS:
        assert L.emu_div_mismatches(c, (127 - 2) << 23, (127 + 17) << 23, 257) == 0
    for qs in (1.0, 0.5, 0.25, 0.3, 0.05):
        assert L.emu_dequant_fast_ok(qs, 6) == 1
    assert L.emu_dequant_fast_ok(1e-9, 6) == 0                     # outside the checked range: divide
    # below exponent -107 the residual underflows and the forms differ: what div_lift's test is for
    assert L.emu_div_mismatches(_LIFT_DIVISORS[0], 1 << 23, (127 - 110) << 23, 1009) > 0


@pytest.mark.parametrize("W,H,wl,qs", [(320, 192, 3, 0.5), (64, 128, 5, 0.3)])
def test_dwt97_inverse_fast_and_dividing_kernels_agree(oracle, E, monkeypatch, W, H, wl, qs):
    """Same coefficients through the reciprocal-form kernels and (PICSONG_DWT_EXACTDIV=1) the dividing
    ones, values beyond 16 bit-planes included (they take the division inside the fast kernels too)."""
    rng = np.random.default_rng(23)
    coef = rng.integers(-300, 301, (H, W)).astype(np.int32)
    coef[rng.random((H, W)) < 0.5] = 0
    coef[5, 7] = 70000
    coef[11, 3] = -131071
    coef[H // 2 + 1, W // 2 + 2] = 65535
    extra = oracle.dwt_extra(W, H, wl)
    ref, ex = oracle.dwt_inverse(coef, wl, True, qs)
    monkeypatch.delenv("PICSONG_DWT_EXACTDIV", raising=False)
    fast = E.dwt_inverse(coef, wl, True, qs, extra=extra)
    monkeypatch.setenv("PICSONG_DWT_EXACTDIV", "1")
    exact = E.dwt_inverse(coef, wl, True, qs, extra=extra)
    assert np.array_equal(exact[extra:].view(np.uint32), ref[ex:].view(np.uint32))
    assert np.array_equal(fast[extra:].view(np.uint32), ref[ex:].view(np.uint32))


@pytest.mark.parametrize("W,H,wl,qs", [(320, 192, 3, 0.5), (64, 128, 5, 0.3), (512, 320, 6, 0.5),
                                       (328, 192, 3, 0.7), (256, 64, 1, 0.5), (1864, 128, 2, 0.25)])
@pytest.mark.parametrize("replay", [False, True])
def test_dwt97_inverse_lean_kernel(oracle, E, monkeypatch, W, H, wl, qs, replay):
    """The lean 9/7 synthesis (dwt_inv97_kernel: branch-free de-quantisation, one division when qs is a power of
    two, the coarsest level's instantiation, bands past the bottom edge, edge and interior waves, coefficients
    beyond the verified 16 bit-planes) against the oracle, samples and fused pixels; `replay`: every wave runs its
    band a second time with true divisions (the pass a too-small lifting operand or a too-big coefficient asks
    for) and must leave the same words."""
    monkeypatch.delenv("PICSONG_DWT_EXACTDIV", raising=False)
    monkeypatch.delenv("PICSONG_DWT_INV97", raising=False)
    if replay:
        monkeypatch.setenv("PICSONG_DWT_EXACT_REPLAY", "1")
    else:
        monkeypatch.delenv("PICSONG_DWT_EXACT_REPLAY", raising=False)
    rng = np.random.default_rng(W + wl)
    coef = rng.integers(-300, 301, (H, W)).astype(np.int32)
    coef[rng.random((H, W)) < 0.6] = 0
    coef[3, 5] = 65535
    coef[H // 2 + 1, W // 2 + 2] = -65535
    coef[H - 1, W - 1] = 4000
    if W >= 512:                                         # beyond 16 bit-planes: those waves take the second pass
        coef[7, 300] = 65536
        coef[H // 2 + 3, 40] = -(1 << 23) + 1
    coef[0, :8] = rng.integers(-20000, 20000, 8)
    extra = oracle.dwt_extra(W, H, wl)
    ref, ex = oracle.dwt_inverse(coef, wl, True, qs)
    got = E.dwt_inverse(coef, wl, True, qs, extra=extra)
    assert np.array_equal(got[extra:].view(np.uint32), ref[ex:].view(np.uint32))
    want = oracle.level_shift_inv(ref[ex:]).reshape(H, W).astype(np.uint8)
    pix, fused = E.dwt_inverse_u8(coef, wl, True, qs, extra=extra)
    assert fused and np.array_equal(pix, want)
    # and the same launches through dwt_inv_kernel's FAST instantiations
    monkeypatch.setenv("PICSONG_DWT_INV97", "0")
    old = E.dwt_inverse(coef, wl, True, qs, extra=extra)
    assert np.array_equal(old[extra:].view(np.uint32), ref[ex:].view(np.uint32))


# ---- levels 0 + 1 of the forward transform in one launch (dwt_fwd2_kernel) ---------------------------
@pytest.mark.parametrize("W,H,wl,lossy,qs", [(320, 192, 3, False, 1.0), (64, 64, 5, False, 1.0), (256, 64, 2, False, 1.0),
                                             (1024, 320, 4, False, 1.0), (768, 128, 2, True, 0.5),
                                             (64, 128, 5, True, 0.3), (512, 256, 6, True, 1.0),
                                             # widths that are 8 mod 16, two workgroups
                                             (328, 192, 3, False, 1.0), (1864, 64, 2, True, 0.7), (936, 128, 3, False, 1.0)])
def test_dwt_fused_levels_0_and_1(oracle, E, monkeypatch, W, H, wl, lossy, qs):
    """The frame path's first two levels as one launch: LL1 stays in registers, level 1's own mirror at
    the bottom of the image comes from the kept rows.  Same words as the oracle and as two launches."""
    monkeypatch.delenv("PICSONG_DWT_NOVEC", raising=False)
    monkeypatch.delenv("PICSONG_DWT_NOFUSE01", raising=False)
    img = oracle.gen_frame(W, H, 7)
    x = oracle.level_shift_fwd(img, lossy)
    extra = oracle.dwt_extra(W, H, wl)
    ref = oracle.dwt_forward(x, wl, qs) if lossy else oracle.dwt_forward(x, wl)
    got = E.dwt_forward(img, wl, lossy, qs, extra=extra)
    assert E.dwt_forward.fused01
    assert np.array_equal(got[:W * H].view(np.uint32), ref[:W * H].view(np.uint32))
    E.dwt_forward(x, wl, lossy, qs, extra=extra)
    assert not E.dwt_forward.fused01                      # 32-bit input: the two-launch path
    monkeypatch.setenv("PICSONG_DWT_NOFUSE01", "1")
    two = E.dwt_forward(img, wl, lossy, qs, extra=extra)
    assert not E.dwt_forward.fused01
    assert np.array_equal(two[:W * H].view(np.uint32), ref[:W * H].view(np.uint32))


@pytest.mark.parametrize("lossy,wl,world,W,H", [(False, 3, 2, 320, 256), (True, 3, 2, 256, 256), (False, 2, 4, 192, 512),
                                                (False, 1, 2, 128, 256)])
def test_banded_transform_equals_full_on_every_ranks_stripes(oracle, E, lossy, wl, world, W, H):
    """picsong_dwt_forward_band / _tail: each rank transforms its row band from an input in which
    only the band and a 4-row halo exist, the LL1 bands are exchanged, levels >= 1 are computed by everyone,
    and the coefficients under the codeblocks the rank codes equal the whole-frame transform's."""
    import picsong_dist as pd
    qs = 0.5
    img = oracle.gen_frame(W, H, 3)
    AH, AW = img.shape
    extra = oracle.dwt_extra(AW, AH, wl)
    os.environ["PICSONG_DWT_NOFUSE01"] = "1"          # the per-level form writes LL1 to the scratch (the fused one keeps it in registers)
    try:
        full = E.dwt_forward(E.aligned_copy(img), wl, lossy, qs, extra)
    finally:
        del os.environ["PICSONG_DWT_NOFUSE01"]
    plan = pd.band_plan(AW, AH, world)
    outs = []
    for k, p in enumerate(plan):
        x = np.full((AH, AW), 0xA5, np.uint8)                      # rows outside band +- halo: poison
        lo, hi = max(0, p["row0"] - 4), min(AH, p["row0"] + p["rows"] + 4)
        x[lo:hi] = img[lo:hi]
        x = E.aligned_copy(x)
        out = E.aligned_zeros(AW * AH + extra, np.float32 if lossy else np.int32)
        out[:] = 12345
        E.dwt_forward_band(x, out, wl, lossy, qs, p["row0"], p["rows"])
        outs.append(out)
    if wl > 1:
        ll1 = np.concatenate([outs[k][AW * AH + p["ll1_begin"]:AW * AH + p["ll1_begin"] + p["ll1_count"]]
                              for k, p in enumerate(plan)])
        assert np.array_equal(ll1, full[AW * AH:AW * AH + (AW // 2) * (AH // 2)])
    for k, p in enumerate(plan):
        if wl > 1:
            outs[k][AW * AH:AW * AH + ll1.size] = ll1                  # the all-gather
            E.dwt_forward_tail(outs[k], AW, AH, wl, lossy, qs)
        got, ref = outs[k][:AW * AH].reshape(AH, AW), full[:AW * AH].reshape(AH, AW)
        for b, n in p["stripes"]:
            r0, r1 = (b // (AW // 64)) * 64, ((b + n) // (AW // 64)) * 64
            if wl == 1 and r0 < AH // 2:
                # one level: the top-left quadrant is LL1 itself, of which a rank holds only its own rows
                assert np.array_equal(got[r0:r1], ref[r0:r1])
            else:
                assert np.array_equal(got[r0:r1], ref[r0:r1]), f"rank {k}, codeblock rows {r0 // 64}..{r1 // 64}"


@pytest.mark.parametrize("W,H,wl,lossy", [(192, 128, 2, False), (128, 128, 3, True), (64, 64, 1, False)])
def test_three_coding_passes_bit_exact(oracle, E, W, H, wl, lossy):
    """-cp 3: bpc3_kernel (encode and decode through the wave emulator) against the oracle's restatement of
    Encode3CP / Decode3CP -- per-codeblock staging, sizes and the decoded coefficients."""
    qs = 0.5
    lut = oracle.lut_for_cp3(lossy, wl)
    img = oracle.pad_frame(oracle.gen_frame(W, H, 2))
    AH, AW = img.shape
    coef = oracle.dwt_forward(oracle.level_shift_fwd(img, lossy), wl, qs)[:AW * AH].reshape(AH, AW)
    st_ref, sz_ref = oracle.bpc_encode(coef, wl, lut)
    st, sz, flag = E.bpc3_encode(coef, wl, lut)
    assert flag == 0
    assert np.array_equal(sz, sz_ref)
    for cb in range(sz.size):                                    # words beyond a codeblock's length are unspecified
        n = int(sz[cb])
        assert np.array_equal(st[cb * 4096:cb * 4096 + n], st_ref[cb * 4096:cb * 4096 + n]), f"codeblock {cb}"
    dec = E.bpc3_decode(st_ref, sz_ref, AW, AH, wl, lut)
    ref_dec = oracle.bpc_decode(st_ref, sz_ref, AW, AH, wl, lut)
    assert np.array_equal(dec, ref_dec)
    if not lossy:
        assert np.array_equal(dec, coef)


def test_three_coding_passes_stress_blocks(oracle, E):
    """-cp 3 corner cases: an all-zero codeblock, an impulse, noise that takes the raw fallback, and a codeblock of
    twelve planes (sparse large coefficients: the planes above the eighth come from the prologue's second pass and go
    through the decoder's sixteen-plane epilogue)."""
    lut = oracle.lut_for_cp3(False, 1)
    rng = np.random.default_rng(7)
    coef = np.zeros((64, 384), np.int32)
    coef[10, 64 + 20] = -37                                      # impulse in codeblock 1
    coef[:, 128:192] = rng.integers(-255, 256, (64, 64))         # noise: expands -> raw fallback (size 4096)
    coef[:, 192:256] = rng.integers(-3, 4, (64, 64))
    ys, xs = rng.integers(0, 64, 40), rng.integers(0, 64, 40)
    coef[ys, 256 + xs] = rng.integers(-3000, 3001, 40)           # codeblock 4: MSB 11, mostly zeros
    coef[:, 320:384] = rng.integers(-1, 2, (64, 64)) * (rng.integers(0, 8, (64, 64)) == 0)
    coef[5, 330] = 20000                                         # codeblock 5: MSB 14
    st_ref, sz_ref = oracle.bpc_encode(coef, 1, lut)
    st, sz, flag = E.bpc3_encode(coef, 1, lut)
    assert flag == 0 and np.array_equal(sz, sz_ref) and sz_ref[0] == 1
    assert sz_ref[4] < 4096 and sz_ref[5] < 4096 and st_ref[4 * 4096] == 11 and st_ref[5 * 4096] == 14
    for cb in range(sz.size):
        n = int(sz[cb])
        assert np.array_equal(st[cb * 4096:cb * 4096 + n], st_ref[cb * 4096:cb * 4096 + n]), f"codeblock {cb}"
    dec = E.bpc3_decode(st_ref, sz_ref, 384, 64, 1, lut)
    assert np.array_equal(dec, oracle.bpc_decode(st_ref, sz_ref, 384, 64, 1, lut))
    assert np.array_equal(dec[:, 256:], coef[:, 256:])


@pytest.mark.parametrize("W,H,wl,lossy,qs", [(320, 192, 3, False, 1.0), (512, 128, 4, False, 1.0), (256, 64, 3, True, 0.5),
                                             (576, 192, 5, True, 0.5), (320, 128, 3, True, 0.3), (256, 128, 2, False, 1.0),
                                             (64, 64, 3, False, 1.0)])
def test_decode_frame_path_with_16_bit_coefficients_and_fused_levels(oracle, E, monkeypatch, W, H, wl, lossy, qs):
    """The decode frame paths' 16-bit form (round 4): the decoder's C16 instantiation writes an int16 Mallat array from
    the packed stream, the synthesis kernels' C16 instantiations read it, and synthesis levels 1 and 0 run as ONE launch
    (dwt_inv2_kernel: LL0 never leaves the registers; not when level 1 is the coarsest, wl = 2) -- the oracle's pixels
    either way, with the fused launch and with the two launches (PICSONG_DWT_NOFUSE_INV=1), and through the second,
    truly dividing pass of the 9/7 kernels (PICSONG_DWT_EXACT_REPLAY=1)."""
    img = oracle.gen_frame(W, H, 21)
    if W >= 512:
        img[:7, :9] = 255; img[-5:, -11:] = 0                  # clamping at both ends, at the image's corners
    lut = oracle.lut_for(lossy, wl)
    ref = oracle.encode_frame(img, wl, lossy, qs, lut)
    ref_pix = oracle.decode_frame(ref, W, H, wl, lossy, qs, lut)
    extra = oracle.dwt_extra(W, H, wl)
    c32 = E.bpc_decode_stream(ref, W, H, wl, lut)
    c16 = E.bpc_decode_stream16(ref, W, H, wl, lut)
    assert E.bpc_decode_stream16.last_bad == 0 and E.bpc_decode_stream16.last_flag == 0
    assert np.array_equal(c16.astype(np.int32), c32)
    monkeypatch.setenv("PICSONG_DWT_FUSE_INV97", "1")        # (9/7: the fused form is built and not the default)
    for nofuse, replay in (("0", "0"), ("1", "0"), ("0", "1")):
        if replay == "1" and not lossy:
            continue
        monkeypatch.setenv("PICSONG_DWT_NOFUSE_INV", nofuse)
        monkeypatch.setenv("PICSONG_DWT_EXACT_REPLAY", replay)
        pix, flags = E.dwt_inverse_u8_c16(c16, wl, lossy, qs, extra=extra)
        assert flags & 4 and flags & 1, (nofuse, replay, flags)
        assert bool(flags & 2) == (nofuse == "0" and wl >= 3), (nofuse, flags)
        assert np.array_equal(pix, ref_pix), (nofuse, replay)
    if lossy:                                                # the default 9/7 form: 16-bit coefficients, one launch per level
        monkeypatch.delenv("PICSONG_DWT_FUSE_INV97")
        monkeypatch.setenv("PICSONG_DWT_NOFUSE_INV", "0")
        monkeypatch.setenv("PICSONG_DWT_EXACT_REPLAY", "0")
        pix, flags = E.dwt_inverse_u8_c16(c16, wl, lossy, qs, extra=extra)
        assert flags == 5 and np.array_equal(pix, ref_pix)
