#!/usr/bin/env python3
"""Encode throughput of resident frames over N streams without per-stage events, and the host time
of the launch loop alone: tools/stream_sweep.py <4k|8k> <streams> [steps]."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-image-and-video-codec_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

import oracle_lib as orc
import picsong_amd as pa

W, H = (3840, 2160) if sys.argv[1] == "4k" else (7680, 4320)
ns = int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
lut = os.path.join(orc.LUT_DIR, "n1_lossless")
cs = [pa.Codec(W, H, wl=5, lut_folder=lut) for _ in range(ns)]
sts = [torch.cuda.Stream() for _ in range(ns)]
frame = torch.from_numpy(orc.pad_frame(orc.gen_frame(W, H, 0))).cuda()
outs = [torch.empty(cs[0].max_stream_shorts(), dtype=torch.int16, device="cuda") for _ in range(ns)]
ref = cs[0].encode_frame(frame, 1).clone()


def run(n):
    for i in range(n):
        k = i % ns
        with torch.cuda.stream(sts[k]):
            cs[k].encode_frame_async(frame, outs[k], 1)
    torch.cuda.synchronize()


run(3 * ns)
t0 = time.perf_counter()
for i in range(steps):                      # host time of the launches alone
    k = i % ns
    with torch.cuda.stream(sts[k]):
        cs[k].encode_frame_async(frame, outs[k], 1)
host = (time.perf_counter() - t0) / steps
torch.cuda.synchronize()
t0 = time.perf_counter()
run(steps)
dt = (time.perf_counter() - t0) / steps
ok = all(torch.equal(o[:ref.numel()], ref) for o in outs)
print(f"{sys.argv[1]} streams {ns}: {dt * 1e3:.4f} ms/frame = "
      f"{W * H / dt / 1e6:.0f} Mpixel/s, host launch loop {host * 1e3:.4f} ms/frame, identical={ok}")
