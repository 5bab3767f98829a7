#!/usr/bin/env python3
"""RGB frame path timing (one stream): colour transform + 3 x (DWT + BPC + pack), 8K, lossless / lossy."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-image-and-video-codec_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

import oracle_lib as orc
import picsong_amd as pa

W, H = 7680, 4320
for lossy, wl, qs in ((False, 5, 1.0), (True, 5, 0.5)):
    lut = os.path.join(orc.LUT_DIR, "n1_lossy" if lossy else "n1_lossless")
    c = pa.Codec(W, H, wl=wl, lossy=lossy, qs=qs, lut_folder=lut, rgb=True)
    planes = [torch.from_numpy(orc.pad_frame(orc.gen_frame(W, H, f))).cuda() for f in range(3)]

    def frame():
        comps = c.rgb_forward(*planes)
        return [c.encode_plane(comps[k], k, k == 0) for k in range(3)]

    for _ in range(2):
        streams = frame()
    torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        streams = frame()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    back = c.rgb_inverse(*[c.decode_plane(streams[k], k) for k in range(3)])
    ok = all(torch.equal(back[k], planes[k].view(c.ah, c.aw)) for k in range(3)) if not lossy else None
    print(f"RGB 8K lossy={lossy}: {dt * 1e3:.3f} ms/frame = {3 * W * H / dt / 1e6:.0f} Msample/s "
          f"({sum(s.numel() for s in streams) * 2 / 1e6:.1f} MB), roundtrip_ok={ok}")
    c.close()
