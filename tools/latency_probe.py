#!/usr/bin/env python3
"""Single-frame encode / decode latency (one stream, resident buffers) on the bench frame and on a
smooth frame whose codeblocks differ widely in bit-plane count; --k=0.5 times the -k > 0 kernels.
PICSONG_SO selects a library variant."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-image-and-video-codec_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import oracle_lib as orc
import picsong_amd as pa

W, H, wl = 7680, 4320, 5
lut = os.path.join(orc.LUT_DIR, "n1_lossless")
k = 0.0
for a in sys.argv[1:]:
    if a.startswith("--k="):
        k = float(a.split("=")[1])          # complexity-scalable mode (-k): the BULK kernels
c = pa.Codec(W, H, wl=wl, lossy=False, qs=1.0, lut_folder=lut, k=k)
print(f"k = {k}")


def smooth_frame():
    y, x = np.mgrid[0:H, 0:W]
    v = 128 + 90 * np.sin(x / 700.0) * np.cos(y / 500.0) + 20 * np.sin((x + 2 * y) / 37.0) * (x > W // 2)
    rng = np.random.default_rng(1)
    v = v + rng.integers(-1, 2, v.shape) * (y > H // 2)
    return np.clip(v, 0, 255).astype(np.uint8)


for name, img in (("bench", orc.gen_frame(W, H, 0)), ("smooth", smooth_frame())):
    frame = torch.from_numpy(orc.pad_frame(img)).cuda()
    s = c.encode_frame(frame).clone()
    for what, fn in (("encode", lambda: c.encode_frame(frame)), ("decode", lambda: c.decode_frame(s))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        n = 30
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"{name:7s} {what}: {dt * 1e3:.3f} ms/frame  ({s.numel() * 2 / 1e6:.1f} MB stream)")
    assert torch.equal(c.decode_frame(s), frame.view(c.ah, c.aw))
