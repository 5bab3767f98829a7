"""Checks on the compiled gfx950 code itself (hipcc cross-compiles here; no GPU needed).

A buffer store of more than 64 bits with a scalar offset keeps reading its data registers after it issues, and the
compiler's hazard recognizer does not cover that form: every such store in the library must be
followed by two wait states before anything else.  Found on hardware as a few wrong samples in one decode of
fifteen; this test keeps a new wide store from coming in without its guard."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "cuda-image-and-video-codec_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


FLAGS = ["-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "--offload-arch=gfx950"]
# the flag sets the tools build besides the default (tools/*_variants.sh, dwt_trace / bpc_trace): the guard must hold
# in each of them -- a regression there shows only as a few wrong samples in one decode of a dozen
VARIANTS = {
    "default": [],
    "trace": ["-DPICSONG_DWT_TRACE"],
    "inv97_group3_waves4": ["-DPICSONG_DWT_INV97_GROUP=3", "-DPICSONG_DWT_INV97_WAVES=4", "-DPICSONG_DWT_INV_GROUP=2"],
}


def _compile(tmp_path_factory, name, src, extra):
    if not os.path.exists(HIPCC) and shutil.which("hipcc") is None:
        pytest.skip("hipcc not installed")
    out = tmp_path_factory.mktemp("isa") / (name + ".s")
    r = subprocess.run([HIPCC if os.path.exists(HIPCC) else "hipcc", *FLAGS, *extra, "--cuda-device-only", "-S", "-o", str(out),
                        src], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    return open(out).read().splitlines()


@pytest.fixture(scope="module")
def device_asm(tmp_path_factory):
    mk = open(os.path.join(CSRC, "Makefile")).read()
    for f in FLAGS[2:4]:
        assert f in mk, f"the library is no longer built with {f}: update this test's flags"
    return _compile(tmp_path_factory, "picsong", os.path.join(CSRC, "picsong_hip.hip"), [])


def _check_wide_stores(asm):
    wide = re.compile(r"^\s*buffer_store_dwordx[34]\b")
    n = 0
    for i, line in enumerate(asm):
        if not wide.match(line):
            continue
        n += 1
        nxt = next(x.strip() for x in asm[i + 1:] if x.strip() and not x.strip().startswith(";"))
        m = re.match(r"s_nop (\d+)", nxt)
        assert m and int(m.group(1)) >= 1, f"line {i + 1}: '{line.strip()}' is followed by '{nxt}'"
    return n


def test_wide_buffer_stores_are_followed_by_two_wait_states(device_asm):
    assert _check_wide_stores(device_asm) > 0          # (rb_store128 is in use; if it goes, so can this test)


@pytest.mark.parametrize("variant", [v for v in VARIANTS if v != "default"])
def test_wide_buffer_stores_in_the_variant_builds(tmp_path_factory, variant):
    asm = _compile(tmp_path_factory, variant, os.path.join(CSRC, "picsong_hip.hip"), VARIANTS[variant])
    assert _check_wide_stores(asm) > 0


def test_the_issue_rate_probe_has_no_unguarded_wide_store(tmp_path_factory):
    asm = _compile(tmp_path_factory, "valu_probe", os.path.join(ROOT, "tools", "valu_probe.hip"), [])
    _check_wide_stores(asm)


def test_lean_synthesis_kernels_do_not_spill(device_asm):
    name = None
    seen = 0
    for line in device_asm:
        m = re.match(r"^\s*\.amdhsa_kernel (\S+)", line)
        if m:
            name = m.group(1)
        m = re.match(r"^\s*\.amdhsa_private_segment_fixed_size (\d+)", line)
        if m and name and "dwt_inv97_kernel" in name:
            seen += 1
            assert int(m.group(1)) == 0, f"{name} uses {m.group(1)} bytes of scratch"
    assert seen >= 12


def _kernel_body(asm, key):
    start = next(i for i, ln in enumerate(asm) if re.match(r"^_ZN7picsong\S*" + key + r"\S*:", ln))
    end = next(i for i in range(start, len(asm)) if "s_endpgm" in asm[i])
    return asm[start:end]


def test_coder_kernels_keep_their_plane_loops_free_of_scratch(device_asm):
    """The coders' scratch is a few spilled dwords of their prologues / epilogues, once per wave.  Indexing the by-value
    argument struct with a run-time value (round 3: `a.lut_c[f]`) moves the whole struct to scratch memory and every use
    of an argument in the plane loops becomes a scratch load -- 15 % of the encoder's rate, with every test still green.
    So: no more than a handful of scratch instructions from the first interval update (v_mul_u32_u24) on, and a bounded
    frame."""
    # (the decoder three times: from the 32-bit staging, the frame paths' instantiation that reads the packed stream, and
    # that one writing 16-bit coefficients; round 4: the -k > 0 instantiations with compact table copies, which are asked
    # for seven waves a SIMD (72 registers) and spill part of their prologues / epilogues -- not their plane loops: the
    # decoder's ~90 scratch instructions sit in its epilogue's transposes and stores, measured faster all the same)
    for key, first, most, frame in (("17bpc_encode_kernelILb0E", "v_mul_u32_u24", 8, 320),
                                    ("17bpc_decode_kernelILb0ELi8ELb0E", "v_bcnt_u32_b32", 80, 192),
                                    ("17bpc_decode_kernelILb0ELi8ELb1ELb0E", "v_bcnt_u32_b32", 80, 192),
                                    ("17bpc_decode_kernelILb0ELi8ELb1ELb1E", "v_bcnt_u32_b32", 80, 192),
                                    ("17bpc_encode_kernelILb1ELb1E", "v_mul_u32_u24", 16, 320),
                                    ("17bpc_decode_kernelILb1ELi8ELb1ELb0ELb1E", "v_bcnt_u32_b32", 104, 320),
                                    ("17bpc_decode_kernelILb1ELi8ELb1ELb1ELb1E", "v_bcnt_u32_b32", 104, 320)):
        body = _kernel_body(device_asm, key)
        i0 = next(i for i, ln in enumerate(body) if first in ln)
        tail = [ln for ln in body[i0:] if re.match(r"^\s*scratch_", ln)]
        assert len(tail) <= most, f"{key}: {len(tail)} scratch instructions after the first '{first}'"
    name, sizes = None, {}
    for line in device_asm:
        m = re.match(r"^\s*\.amdhsa_kernel (\S+)", line)
        if m:
            name = m.group(1)
        m = re.match(r"^\s*\.amdhsa_private_segment_fixed_size (\d+)", line)
        if m and name:
            sizes[name] = int(m.group(1))
    enc = [v for k, v in sizes.items() if "bpc_encode_kernelILb0E" in k]
    dec = [v for k, v in sizes.items() if "bpc_decode_kernelILb0ELi8" in k]
    assert enc and len(dec) == 3 and enc[0] <= 320 and max(dec) <= 192, (enc, dec)
