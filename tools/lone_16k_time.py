#!/usr/bin/env python3
"""BASELINE config 5's geometry on ONE GPU: a lone 16384 x 16384 frame (65,536 codeblocks) through
picsong_encode_frame / picsong_decode_frame on one stream, round trip checked.  GPU box."""
import os, sys, time
sys.path.insert(0, "cuda-image-and-video-codec_amd/python"); sys.path.insert(0, "tests")
import torch, oracle_lib as orc, picsong_amd as pa
W = H = 16384
c = pa.Codec(W, H, wl=5, lossy=False, qs=1.0, lut_folder=os.path.join(orc.LUT_DIR, "n1_lossless"))
tile = torch.from_numpy(orc.gen_frame(2048, 2048, 1)).cuda()
frame = tile.repeat(8, 8).contiguous()
out = torch.empty(c.max_stream_shorts(), dtype=torch.int16, device="cuda")
s = c.encode_frame(frame).clone()
for name, fn in (("encode", lambda: c.encode_frame_async(frame, out, 0)), ("decode", lambda: c.decode_frame(s))):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"16384x16384 lone {name}: {dt*1e3:.3f} ms = {W*H/dt/1e9:.1f} Gpixel/s")
print("round trip", bool(torch.equal(c.decode_frame(s), frame)))
