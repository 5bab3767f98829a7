#!/usr/bin/env python3
"""Generates tests/golden/vectors.json: per-stage dumps of the CPU oracle (oracle/picsong_oracle.c)
on small synthetic inputs.  The reference ships no golden vectors and cannot be built
here, so these pin the ORACLE (and, through the parity tests, the HIP path) against regressions --
they are not outputs of the CUDA binary ("parity unpinned", oracle header / DESIGN.md 2).

    python tests/golden/make_golden.py            # rewrites vectors.json

Cases: 64x64 / 128x128 / 512x512 frames of the integer-only generator (frame index 0) through
`-type 0 -wl 3`, `-type 0 -wl 5`, `-type 1 -qs 0.5 -wl 5` (cp 2, k 0, LUT n1_lossless / n1_lossy,
component R), plus one `-k 0.5` case.  Per case: CRC-32 of the input, of the DWT output (the raw
32-bit words: int32 for 5/3, float32 bits for 9/7) and of the coded coefficients (truncated int32),
every codeblock's MSB and length (`sizeArray`), the first 64 codewords of the first / middle / last
codeblock, and length + SHA-256 of the whole codestream."""
import hashlib
import json
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as orc  # noqa: E402

CASES = [(n, lossy, wl, qs, 0.0) for n in (64, 128, 512)
         for (lossy, wl, qs) in ((False, 3, 1.0), (False, 5, 1.0), (True, 5, 0.5))] + [(128, False, 4, 1.0, 0.5)]


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def case_vector(n, lossy, wl, qs, k):
    img = orc.gen_frame(n, n, 0)
    lut = orc.lut_for_k(lossy, wl) if k > 0 else orc.lut_for(lossy, wl)
    x = orc.level_shift_fwd(img, lossy)
    full = orc.dwt_forward(x, wl, qs) if lossy else orc.dwt_forward(x, wl)
    coef_t = full[:n * n].reshape(n, n)
    coded = np.trunc(coef_t).astype(np.int32) if lossy else coef_t.astype(np.int32)
    staging, sizes = orc.bpc_encode(coef_t, wl, lut, k=k)
    ncb = sizes.size
    st = staging.reshape(ncb, 4096)
    picks = sorted({0, ncb // 2, ncb - 1})
    stream = orc.encode_frame(img, wl, lossy, qs, lut, 0, 0, k=k)
    return {
        "name": f"{n}x{n}_type{int(lossy)}_wl{wl}" + (f"_qs{qs}" if lossy else "") + (f"_k{k}" if k > 0 else ""),
        "W": n, "H": n, "frame": 0, "lossy": bool(lossy), "wl": wl, "qs": qs, "k": k,
        "input_crc32": crc(img),
        "dwt_words_crc32": crc(coef_t.view(np.uint32)),
        "coded_coefficients_crc32": crc(coded),
        "msb": [int(v) for v in st[:, 0]],
        "sizes": [int(v) for v in sizes],
        "codewords": {str(cb): [int(v) & 0xFFFF for v in st[cb, 1:1 + min(64, max(int(sizes[cb]) - 1, 0))]]
                      for cb in picks if sizes[cb] != 4096},
        "stream_shorts": int(stream.size),
        "stream_sha256": hashlib.sha256(stream.tobytes()).hexdigest(),
    }


def main():
    out = {"generator": "tests/golden/make_golden.py", "oracle": "oracle/picsong_oracle.c",
           "cases": [case_vector(*c) for c in CASES]}
    with open(os.path.join(HERE, "vectors.json"), "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    for c in out["cases"]:
        print(c["name"], c["stream_shorts"], c["stream_sha256"][:16])


if __name__ == "__main__":
    main()
