"""CPU oracle pinned against everything available: the shipped LUT tables, the
Appendix A.10 known-answer vectors, header bit layout, structural DWT properties, and
encode -> decode identity.  Parity with the CUDA binary itself is unpinned (cannot be built)."""
import numpy as np
import pytest


def _kat_block(M, O, z0):
    z = z0
    blk = np.zeros((64, 64), np.int32)
    for y in range(64):
        for x in range(64):
            z = (1103515245 * z + 12345) % (1 << 31)
            blk[y, x] = ((z >> 16) % M) - O
    return blk


KATS = [
    (15, 7, 12345, 0, 2, 2, 1081, [65429, 58283, 28962, 65200, 52297, 41111, 58821, 64786],
     [53436, 37376, 37376, 21316], 42880196, 15583),
    (401, 200, 777, 2, 0, 7, 2417, [41348, 62746, 65527, 64578, 65265, 57988, 64206, 64827],
     [36352, 36352, 20164, 36352], 94766947, 7602),
    (3, 1, 42, 5, 0, 0, 547, [17310, 47007, 32770, 49664, 193, 42410, 49884, 23653],
     [0, 23552, 0, 33792], 15110958, 62032),
]


@pytest.mark.parametrize("kat", KATS)
def test_bpc_known_answer_vectors(oracle, kat):
    M, O, z0, level, sb, msb, length, first8, last4, sum32, xorfold = kat
    lut = oracle.lut_for(False, 5)
    st, n = oracle.bpc_encode_block_uniform(_kat_block(M, O, z0), level, sb, 5, lut)
    cw = st[1:n].astype(np.int64)
    assert st[0] == msb and n == length
    assert cw[:8].tolist() == first8 and cw[-4:].tolist() == last4
    assert int(cw.sum() % (1 << 32)) == sum32
    xf = 0
    for i, c in enumerate(cw):
        xf ^= (int(c) * (i + 1)) & 0xFFFF
    assert xf == xorfold


def test_lut_geometry_and_sections(oracle):
    lut = oracle.lut_for(False, 5)
    g = lut.geometry()
    assert (g["n_bitplanes"], g["n_subbands"], g["ctx_ref"], g["ctx_sign"], g["ctx_sig"],
            g["precision"]) == (15, 3, 1, 4, 9, 7)
    assert (g["n_ref"], g["n_sig"], g["n_sign"]) == (240, 2160, 960)
    t = lut.table
    # first significance row of the file: "0 0 0 : 67 56 53 51 49 47 46 44 43"
    assert t[240:249].tolist() == [67, 56, 53, 51, 49, 47, 46, 44, 43]
    # group (0,0) has planes 0..9 in the file; 10..14 are filled with 64 on the group change
    assert t[240 + 10 * 9: 240 + 15 * 9].tolist() == [64] * 45
    # LL group of wl=5 is the file's "5 0" group at level*3*15*9
    assert t[240 + 5 * 3 * 15 * 9: 240 + 5 * 3 * 15 * 9 + 3].tolist() != [0, 0, 0]
    assert (t >= 0).all() and (t <= 128).all()


def test_lut_holes_only_for_wl6(oracle):
    """wl=6 leaves groups 16,17,18 of every section unwritten; wl=3/5 none."""
    for wl, holes in ((3, []), (5, []), (6, [16, 17, 18])):
        lut = oracle.Lut(oracle.LUT_DIR + "/n1_lossless", wl, 1, fill=-7)
        g = lut.geometry()
        base = 0
        for ctx, n in ((g["ctx_ref"], g["n_ref"]), (g["ctx_sig"], g["n_sig"]),
                       (g["ctx_sign"], g["n_sign"])):
            sec = lut.table[base:base + n].reshape(-1, 15 * ctx)
            unwritten = [i for i in range(sec.shape[0]) if (sec[i] == -7).all()]
            partially = [i for i in range(sec.shape[0])
                         if (sec[i] == -7).any() and not (sec[i] == -7).all()]
            assert unwritten == holes and partially == []
            base += n


def test_lut_ll_group_for_small_wl(oracle):
    """For wl < 5 the LL group comes from the file's "level=wl, subband 0" rows."""
    l3 = oracle.lut_for(False, 3)
    l5 = oracle.lut_for(False, 5)
    g = 15 * 9
    # file group (3,0) sits at group index 9 for both layouts
    a = l3.table[l3.c.n_ref + 9 * g: l3.c.n_ref + 10 * g]
    b = l5.table[l5.c.n_ref + 9 * g: l5.c.n_ref + 10 * g]
    assert a.tolist() == b.tolist()


def test_find_subband(oracle):
    import ctypes as C
    L = oracle.lib()
    lv, sb = C.c_int(), C.c_int()

    def fs(x, y, AW, AH, wl):
        L.po_find_subband(x, y, AW, AH, wl, C.byref(lv), C.byref(sb))
        return lv.value, sb.value
    assert fs(0, 0, 512, 512, 3) == (3, 0)           # LL
    assert fs(256, 0, 512, 512, 3) == (0, 0)         # HL level 0
    assert fs(0, 256, 512, 512, 3) == (0, 1)         # LH
    assert fs(256, 256, 512, 512, 3) == (0, 2)       # HH
    assert fs(64, 0, 512, 512, 3) == (2, 0)
    assert fs(0, 128, 512, 512, 3) == (1, 1)
    # 4K straddle: boundary 3840>>3 = 480 is inside codeblock column 7 (448..511)
    assert fs(478, 0, 3840, 2176, 5) == (3, 0) and fs(480, 0, 3840, 2176, 5) == (2, 0)


def test_header_roundtrip_and_layout(oracle):
    h = dict(n_samples=3840 * 2160, cp=2, cb_height=18, cb_width=64, wl=5, bit_depth=8, lossy=1,
             qs_1e4=5000, components=1, is_rgb=0, height=2160, endianess=0, bps=8, is_signed=0,
             frames=256, k_1e3=0)
    s = oracle.header_pack(**h)
    assert s[0] == (3840 * 2160) & 0xFFFF and s[1] == (3840 * 2160) >> 16
    assert s[2] == (0 | (18 << 1) | (64 << 8) | (1 << 15))
    assert s[3] == ((5 & 7) >> 1 | (8 << 3) | (1 << 10) | ((5000 & 31) << 11)) & 0xFFFF
    assert s[7] == 128 and s[8] == 0
    assert oracle.header_unpack(s) == h


def test_pad_mirror(oracle):
    img = np.arange(70 * 100, dtype=np.uint32).reshape(70, 100).astype(np.uint8)
    p = oracle.pad_frame(img)
    assert p.shape == (128, 128)
    assert (p[:70, :100] == img).all()
    assert (p[:70, 100:128] == img[:, 99:71:-1]).all()          # col W+j = col W-1-j
    assert (p[70:128] == p[69:11:-1]).all()                     # row H+r = row H-1-r


def test_generator_is_integer_only_and_stable(oracle):
    a = oracle.gen_frame(96, 80, 0)
    assert int(a.astype(np.int64).sum()) == int(oracle.gen_frame(96, 80, 0).astype(np.int64).sum())
    assert not np.array_equal(a, oracle.gen_frame(96, 80, 1))
    # hand evaluation of pixel (0,0): tri(0,512)=512 -> 255, tri(0,384)=384 -> 255, base 255
    z = (1664525 * 0x5EED1234 + 1013904223) & 0xFFFFFFFF
    exp = min(255, max(0, 255 + ((z >> 24) & 15) - 8 + 0))
    assert a[0, 0] == exp


@pytest.mark.parametrize("wl", [1, 2, 3])
def test_dwt53_perfect_reconstruction(oracle, wl):
    rng = np.random.default_rng(wl)
    x = rng.integers(-128, 128, (128, 192), dtype=np.int32)
    f = oracle.dwt_forward(x, wl)
    coef = f[:x.size].reshape(x.shape)
    r, extra = oracle.dwt_inverse(coef, wl, False)
    assert np.array_equal(r[extra:].reshape(x.shape), x)


def test_dwt53_constant_and_impulse(oracle):
    c = np.full((64, 64), 37, np.int32)
    f = oracle.dwt_forward(c, 1)[:4096].reshape(64, 64)
    assert (f[:32, :32] == 37).all() and (f[32:] == 0).all() and (f[:, 32:] == 0).all()
    # impulse at an odd/odd position only touches HH and its lifting neighbourhood
    x = np.zeros((64, 64), np.int32)
    x[9, 9] = 64
    f = oracle.dwt_forward(x, 1)[:4096].reshape(64, 64)
    assert f[32 + 4, 32 + 4] == 64                       # HH gets the sample itself
    assert f[4, 32 + 4] == 16 and f[5, 32 + 4] == 16     # HL: (0 + 64 + 2) >> 2 vertically
    assert f[4, 4] == 4                                   # LL: (16 + 0 + 2) >> 2


def test_dwt97_reconstruction_close(oracle):
    rng = np.random.default_rng(0)
    x = rng.integers(-128, 128, (128, 128)).astype(np.float32)
    f = oracle.dwt_forward(x, 3, qs=1.0)
    q = np.trunc(f[:x.size]).astype(np.int32).reshape(x.shape)
    r, extra = oracle.dwt_inverse(q, 3, True, qs=1.0)
    err = np.abs(r[extra:].reshape(x.shape) - x)
    assert err.max() < 2.5 and err.mean() < 0.6


@pytest.mark.parametrize("name", ["zeros", "c255", "random", "impulse", "synthetic"])
def test_lossless_roundtrip_edge_inputs(oracle, name):
    W, H, wl = 192, 128, 2
    rng = np.random.default_rng(1)
    img = {"zeros": np.zeros((H, W), np.uint8), "c255": np.full((H, W), 255, np.uint8),
           "random": rng.integers(0, 256, (H, W), dtype=np.uint8),
           "impulse": np.zeros((H, W), np.uint8), "synthetic": oracle.gen_frame(W, H)}[name]
    if name == "impulse":
        img[77, 33] = 255
    lut = oracle.lut_for(False, wl)
    s = oracle.encode_frame(img, wl, False, 1.0, lut)
    assert s[-1] == 0xFFFF
    assert np.array_equal(oracle.decode_frame(s, W, H, wl, False, 1.0, lut), img)


def test_random_noise_takes_raw_fallback(oracle):
    rng = np.random.default_rng(2)
    coef = rng.integers(-30000, 30000, (64, 128), dtype=np.int32)
    lut = oracle.lut_for(False, 1)
    st, sizes = oracle.bpc_encode(coef, 1, lut)
    assert (sizes == 4096).all()
    w = st.reshape(2, 32, 64, 2)                       # [cb][lane][row][side]
    assert w[0, 3, 5, 1] == ((abs(int(coef[5, 7])) << 1) | int(coef[5, 7] < 0))
    assert np.array_equal(oracle.bpc_decode(st, sizes, 128, 64, 1, lut), coef)


def test_all_zero_block_codes_nothing(oracle):
    coef = np.zeros((64, 64), np.int32)
    lut = oracle.lut_for(False, 1)
    st, sizes = oracle.bpc_encode(coef, 1, lut)
    assert sizes.tolist() == [1] and st[0] == 32 and (st[1:] == -1).all()


def test_ragged_size_roundtrip(oracle):
    img = oracle.gen_frame(100, 70)
    lut = oracle.lut_for(False, 1)
    s = oracle.encode_frame(img, 1, False, 1.0, lut)
    assert np.array_equal(oracle.decode_frame(s, 100, 70, 1, False, 1.0, lut), img)


def test_config1_512_lossless_wl3(oracle):
    """BASELINE config 1: 512x512, -type 0 -wl 3 -cp 2, CPU round trip."""
    img = oracle.gen_frame(512, 512)
    lut = oracle.lut_for(False, 3)
    s = oracle.encode_frame(img, 3, False, 1.0, lut)
    h = oracle.header_unpack(s[:9])
    assert h["n_samples"] == 512 * 512 and h["wl"] == 3 and h["lossy"] == 0 and h["height"] == 512
    n_cb = 64
    total = 9 + 2 * n_cb + int((s[10:10 + 2 * n_cb:2].astype(np.int64) - 1).sum()) + 1
    assert total == s.size
    assert np.array_equal(oracle.decode_frame(s, 512, 512, 3, False, 1.0, lut), img)


def test_lossy_psnr(oracle):
    img = oracle.gen_frame(256, 256)
    lut = oracle.lut_for(True, 3)
    s = oracle.encode_frame(img, 3, True, 0.5, lut)
    d = oracle.decode_frame(s, 256, 256, 3, True, 0.5, lut)
    mse = np.mean((d.astype(np.float64) - img) ** 2)
    assert 10 * np.log10(255 ** 2 / mse) > 40.0
    assert s.size * 16 < 256 * 256 * 8        # it compresses


def test_pack_unpack_inverse(oracle):
    rng = np.random.default_rng(3)
    n_cb = 7
    sizes = rng.integers(1, 300, n_cb).astype(np.int32)
    sizes[2] = 1
    staging = np.full(n_cb * 4096, -1, np.int32)
    for cb in range(n_cb):
        staging[cb * 4096: cb * 4096 + sizes[cb]] = rng.integers(0, 65536, sizes[cb])
        staging[cb * 4096] = rng.integers(0, 15)
    s = oracle.bitstream_pack(staging, sizes, None)
    assert (s[:9] == 0xFFFF).all() and s[-1] == 0xFFFF
    assert s.size == 9 + 2 * n_cb + int((sizes - 1).sum()) + 1
    st2, sz2 = oracle.bitstream_unpack(s, n_cb)
    assert np.array_equal(sz2, sizes) and np.array_equal(st2, staging)


# ---- complexity-scalable mode, -k > 0 -------------------------------------------

def _k_coeffs(O, W=256, H=192, wl=2, seed=5):
    rng = np.random.default_rng(seed)
    img = (np.clip(rng.normal(128, 40, (H, W)), 0, 255)).astype(np.uint8)
    img[40:90, 60:200] = 200
    x = O.level_shift_fwd(img, False)
    return O.dwt_forward(x, wl)[:W * H].reshape(H, W)


def test_consecutive_bitplanes_formula(oracle):
    O = oracle
    # Encode BPCEngine.cu:1684-1692: LL uses L2Norm[max(level-1,0)][0], others L2Norm[level][3-sb]
    assert O.consecutive_bitplanes(10, 0.0, 0, 2, 5) == 0
    assert O.consecutive_bitplanes(10, 1.0, 0, 2, 5) == int(np.floor(np.float32(10) * (np.float32(1.0) / np.float32(1.0112865))))
    assert O.consecutive_bitplanes(10, 1.0, 0, 0, 5) == int(np.floor(np.float32(10) * (np.float32(1.0) / np.float32(0.52021784))))
    assert O.consecutive_bitplanes(12, 2.0, 5, 0, 5) == int(np.floor(np.float32(12) * (np.float32(2.0) / np.float32(33.924816))))
    assert O.consecutive_bitplanes(7, 65.0, 1, 1, 5) > 7          # everything in bulk


def test_lut_multi_table_layout(oracle):
    O = oracle
    l1 = O.lut_for(False, 3)
    lk = O.lut_for_k(False, 3)
    assert lk.n_tables == 15 and lk.table.size == 15 * l1.total
    assert np.array_equal(lk.table[:l1.total], l1.table)          # table 0 == the k = 0 table
    assert np.array_equal(lk.table[l1.total:2 * l1.total], l1.table)          # shipped _1 == _0
    assert not np.array_equal(lk.table[2 * l1.total:3 * l1.total], l1.table)  # _2 onward differ


@pytest.mark.parametrize("k", [0.05, 0.3, 1.0, 4.0, 65.0])
def test_bulk_mode_round_trip_lossless(oracle, k):
    O = oracle
    wl = 2
    coef = _k_coeffs(O, wl=wl)
    H, W = coef.shape
    lut = O.lut_for_k(False, wl)
    st, sz = O.bpc_encode(coef, wl, lut, k=k)
    back = O.bpc_decode(st, sz, W, H, wl, lut, k=k)
    assert np.array_equal(back, coef)
    assert (sz > 1).all()


def test_bulk_mode_k0_equals_plain_and_small_k_differs(oracle):
    O = oracle
    wl = 2
    coef = _k_coeffs(O, wl=wl)
    lut0 = O.lut_for(False, wl)
    lutk = O.lut_for_k(False, wl)
    st0, sz0 = O.bpc_encode(coef, wl, lut0)
    stk, szk = O.bpc_encode(coef, wl, lutk, k=0.0)
    assert np.array_equal(sz0, szk) and np.array_equal(st0, stk)
    st1, sz1 = O.bpc_encode(coef, wl, lutk, k=1.0)
    assert not np.array_equal(sz1, sz0)


def test_bulk_mode_frame_round_trip_and_header(oracle):
    O = oracle
    W, H, wl = 200, 150, 3
    img = O.gen_frame(W, H, 1)
    lut = O.lut_for_k(False, wl)
    s = O.encode_frame(img, wl, False, 1.0, lut, k=0.3)
    assert O.header_unpack(s[:9])["k_1e3"] == 300
    assert np.array_equal(O.decode_frame(s, W, H, wl, False, 1.0, lut, k=0.3), img)
    lutl = O.lut_for_k(True, wl)
    sl = O.encode_frame(img, wl, True, 0.5, lutl, k=0.5)
    d = O.decode_frame(sl, W, H, wl, True, 0.5, lutl, k=0.5).astype(np.float64)
    mse = np.mean((d - img) ** 2)
    assert 10 * np.log10(255 ** 2 / mse) > 35


# ---- -cp 3: three coding passes (Encode3CP / Decode3CP, BPC/BPCEngine.cu:1727-1776,1844-1900) ----------
@pytest.mark.parametrize("W,H,wl,lossy", [(320, 192, 3, False), (256, 128, 2, True)])
def test_cp3_round_trip_and_header(oracle, W, H, wl, lossy):
    """The oracle's 3-pass restatement decodes what it encodes (identity for 5/3; for 9/7 the same pixels the
    2-pass stream decodes to: both carry the same quantised coefficients), the streams of the two modes differ,
    and header bit 0 of word 2 says which one it is (BitStreamBuilder.cpp:59)."""
    qs = 0.5
    img = oracle.gen_frame(W, H, 5)
    l3, l2 = oracle.lut_for_cp3(lossy, wl), oracle.lut_for(lossy, wl)
    s3, s2 = oracle.encode_frame(img, wl, lossy, qs, l3), oracle.encode_frame(img, wl, lossy, qs, l2)
    assert s3[2] & 1 == 1 and s2[2] & 1 == 0
    assert s3.size != s2.size or not np.array_equal(s3, s2)
    d3 = oracle.decode_frame(s3, W, H, wl, lossy, qs, l3)
    assert np.array_equal(d3, oracle.decode_frame(s2, W, H, wl, lossy, qs, l2))
    if not lossy:
        assert np.array_equal(d3, img)


def test_cp3_table_layout(oracle):
    """[ref | sig | sign | cp_sig | cp_sign]: the cleanup tables sit nSig + nSign entries after the ordinary
    ones (Encode3CP's LUTPointerAux, :1744-1745), parsed by the same loop (IO/IOManager.ipp:539-606)."""
    l3, l2 = oracle.lut_for_cp3(False, 3), oracle.lut_for(False, 3)
    g = l3.geometry()
    b3 = g["n_ref"] + g["n_sig"] + g["n_sign"]
    assert l3.table.size == b3 + g["n_sig"] + g["n_sign"]
    assert np.array_equal(l3.table[:b3], l2.table)
    sig, cps = l3.table[g["n_ref"]:g["n_ref"] + g["n_sig"]], l3.table[b3:b3 + g["n_sig"]]
    written = sig > 0
    assert np.array_equal(cps[written & (sig != 64)], np.minimum(127, (3 * sig[written & (sig != 64)] + 127) // 4))
