#!/usr/bin/env python3
"""A -k > 0 video of 4K frames (BASELINE configs[3]'s frames in the complexity-scalable mode): frames one per call and
B per call (picsong_encode_frames / picsong_decode_frames, the BULK coder instantiations over the B frames of a launch),
three hinted contexts on three streams.  usage: tools/video_k_time.py [k] [B]   (GPU box)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-image-and-video-codec_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

import oracle_lib as orc
import picsong_amd as pa

k = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
B = int(sys.argv[2]) if len(sys.argv) > 2 else 6
W, H, wl = 3840, 2160, 5
lut = os.path.join(orc.LUT_DIR, "n1_lossless")
frames = torch.stack([torch.from_numpy(orc.pad_frame(orc.gen_frame(W, H, f))).cuda().view(-1) for f in range(B)])
sts = [torch.cuda.Stream() for _ in range(3)]
for kk in (0.0, k):
    cs = [pa.Codec(W, H, wl=wl, lut_folder=lut, k=kk, pipelined=True) for _ in range(3)]
    outs = [torch.empty((B, cs[0].max_stream_shorts()), dtype=torch.int16, device="cuda") for _ in range(3)]
    res = {}
    for b in (1, B):
        def enc(i):
            cs[i].encode_frames_async(frames[:b], outs[i][:b], 1)
        def dec(i):
            cs[i].decode_frames(outs[i][:b])
        for name, fn in (("encode", enc), ("decode", dec)):
            for i in range(3):
                with torch.cuda.stream(sts[i]):
                    fn(i); fn(i)
            torch.cuda.synchronize()
            n = 60
            t0 = time.perf_counter()
            for j in range(n):
                with torch.cuda.stream(sts[j % 3]):
                    fn(j % 3)
            torch.cuda.synchronize()
            res[(name, b)] = W * H * b * n / (time.perf_counter() - t0) / 1e9
    ok = bool(torch.equal(cs[0].decode_frames(outs[0])[B - 1].view(-1)[:frames.shape[1]], frames[B - 1]))
    print(f"4K k = {kk}: encode {res[('encode', 1)]:.0f} Gpixel/s one frame a call, {res[('encode', B)]:.0f} with {B} a call; "
          f"decode {res[('decode', 1)]:.0f} / {res[('decode', B)]:.0f}; round trip {ok}")
    for c in cs:
        c.close()
