"""The product's HIP kernel sources, compiled for the CPU wave emulator, run under UBSan and ASan
(tools/sanitize_emu.sh).  GPU sanitizers are not available on the pool; in the emulator every
global / LDS access of a kernel is a real host access, so out-of-bounds indexing and undefined
behaviour in the kernels show up here."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name):
    out = subprocess.run(["gcc", f"-print-file-name={name}"], capture_output=True, text=True).stdout.strip()
    return out if os.path.isabs(out) and os.path.exists(out) else None


@pytest.mark.parametrize("mode,rt", [("ubsan", "libubsan.so"), ("asan", "libasan.so")])
def test_emulated_kernels_are_sanitizer_clean(mode, rt):
    if shutil.which("g++") is None or _runtime(rt) is None:
        pytest.skip(f"{rt} not installed")
    env = dict(os.environ)
    env.pop("PICSONG_EMU_SO", None)
    r = subprocess.run([os.path.join(ROOT, "tools", "sanitize_emu.sh"), mode], capture_output=True, text=True, env=env,
                       timeout=1500)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0 and " passed" in r.stdout and "runtime error" not in tail and "AddressSanitizer" not in tail, tail
