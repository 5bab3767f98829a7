#!/usr/bin/env python3
"""Single-frame encode / decode calls on one stream, timed from Python (wall clock over 30 calls, launch overhead
included) at 8K and 4K: what an image codec's caller sees (profiles/r03_lone_frame.txt).  PICSONG_SO selects a library."""
import os, sys, time
sys.path.insert(0, "cuda-image-and-video-codec_amd/python"); sys.path.insert(0, "tests")
import torch, oracle_lib as orc, picsong_amd as pa
for (W, H) in ((7680, 4320), (3840, 2160)):
    c = pa.Codec(W, H, wl=5, lossy=False, qs=1.0, lut_folder=os.path.join(orc.LUT_DIR, "n1_lossless"))
    frame = torch.from_numpy(orc.pad_frame(orc.gen_frame(W, H, 0))).cuda()
    s = c.encode_frame(frame).clone()
    for name, fn in (("encode", lambda: c.encode_frame(frame)), ("decode", lambda: c.decode_frame(s))):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
        print(f"{W}x{H} lone {name}: {dt*1e3:.4f} ms = {W*H/dt/1e6:.0f} Mpx/s")
    c.close()
