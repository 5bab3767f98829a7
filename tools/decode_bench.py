#!/usr/bin/env python3
"""Times picsong_decode_frame (unpack + BPC decode + inverse DWT + clamp) on a resident 8K
codestream; prints Mpixel/s and checks the round trip.  Secondary figure (the headline is encode)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-image-and-video-codec_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

import oracle_lib as orc
import picsong_amd as pa

lossy = "lossy" in sys.argv[1:]
nstreams = 1
for a in sys.argv[1:]:
    if a.startswith("--streams="):
        nstreams = int(a.split("=")[1])
W, H, wl, qs = (7680, 4320, 6, 0.5) if lossy else (7680, 4320, 5, 1.0)
if "4k" in sys.argv[1:]:
    W, H = 3840, 2160
lut = os.path.join(orc.LUT_DIR, "n1_lossy" if lossy else "n1_lossless")
c = pa.Codec(W, H, wl=wl, lossy=lossy, qs=qs, lut_folder=lut)
frame = torch.from_numpy(orc.pad_frame(orc.gen_frame(W, H, 0))).cuda()
s = c.encode_frame(frame).clone()
for _ in range(3):
    d = c.decode_frame(s)
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n):
    d = c.decode_frame(s)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
ok = bool(torch.equal(d, frame.view(c.ah, c.aw))) if not lossy else None
print(f"decode {W}x{H} lossy={lossy}: {dt * 1e3:.3f} ms/frame = {W * H / dt / 1e6:.0f} Mpixel/s, roundtrip_ok={ok}")
if nstreams > 1:
    # frames pipelined over several streams, each with its own context (like bench.py does for encode)
    cs = [c] + [pa.Codec(W, H, wl=wl, lossy=lossy, qs=qs, lut_folder=lut) for _ in range(nstreams - 1)]
    sts = [torch.cuda.Stream() for _ in range(nstreams)]
    outs = [None] * nstreams
    torch.cuda.synchronize()
    n = 60
    for rep in range(2):
        t0 = time.perf_counter()
        for i in range(n):
            k = i % nstreams
            with torch.cuda.stream(sts[k]):
                outs[k] = cs[k].decode_frame(s)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
    okp = all(bool(torch.equal(o, frame.view(c.ah, c.aw))) for o in outs) if not lossy else None
    print(f"decode pipelined over {nstreams} streams: {dt * 1e3:.3f} ms/frame = {W * H / dt / 1e6:.0f} Mpixel/s, roundtrip_ok={okp}")

for a in sys.argv[1:]:
    if a.startswith("--batch="):
        # n frames per picsong_decode_frames call (one launch per stage), the calls alternating over the streams
        nb = int(a.split("=")[1])
        sb = torch.stack([torch.nn.functional.pad(s, (0, c.max_stream_shorts() - s.numel())) for _ in range(nb)])
        cs = [pa.Codec(W, H, wl=wl, lossy=lossy, qs=qs, lut_folder=lut) for _ in range(max(nstreams, 1))]
        sts = [torch.cuda.Stream() for _ in cs]
        outs = [torch.empty((nb, c.ah, c.aw), dtype=torch.uint8, device="cuda") for _ in cs]
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ncalls = max(6, 60 // nb)
            for i in range(ncalls):
                k = i % len(cs)
                with torch.cuda.stream(sts[k]):
                    cs[k].decode_frames(sb, outs[k])
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / (ncalls * nb)
        okb = all(bool(torch.equal(o[j], frame.view(c.ah, c.aw))) for o in outs for j in range(nb)) if not lossy else None
        print(f"decode {nb} frames per call over {len(cs)} streams: {dt * 1e3:.3f} ms/frame = {W * H / dt / 1e6:.0f} Mpixel/s, roundtrip_ok={okb}")
