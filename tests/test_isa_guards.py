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


@pytest.fixture(scope="module")
def device_asm(tmp_path_factory):
    if not os.path.exists(HIPCC) and shutil.which("hipcc") is None:
        pytest.skip("hipcc not installed")
    out = tmp_path_factory.mktemp("isa") / "picsong.s"
    flags = ["-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "--offload-arch=gfx950"]
    mk = open(os.path.join(CSRC, "Makefile")).read()
    for f in flags[2:4]:
        assert f in mk, f"the library is no longer built with {f}: update this test's flags"
    r = subprocess.run([HIPCC if os.path.exists(HIPCC) else "hipcc", *flags, "--cuda-device-only", "-S", "-o", str(out),
                        os.path.join(CSRC, "picsong_hip.hip")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    return open(out).read().splitlines()


def test_wide_buffer_stores_are_followed_by_two_wait_states(device_asm):
    wide = re.compile(r"^\s*buffer_store_dwordx[34]\b")
    n = 0
    for i, line in enumerate(device_asm):
        if not wide.match(line):
            continue
        n += 1
        nxt = next(x.strip() for x in device_asm[i + 1:] if x.strip() and not x.strip().startswith(";"))
        m = re.match(r"s_nop (\d+)", nxt)
        assert m and int(m.group(1)) >= 1, f"line {i + 1}: '{line.strip()}' is followed by '{nxt}'"
    assert n > 0          # (rb_store128 is in use; if it goes, so can this test)


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
