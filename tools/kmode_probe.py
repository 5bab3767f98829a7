#!/usr/bin/env python3
"""Lone 8K frames through one mode of the coder (-k K, default 0.5): a dozen encodes and decodes on one stream -- the
workload the rocprofv3 passes over the -k > 0 kernels run (tools/kmode_prof.sh)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-image-and-video-codec_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

import oracle_lib as orc
import picsong_amd as pa

k = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
W, H, wl = 7680, 4320, 5
frame = torch.from_numpy(orc.pad_frame(orc.gen_frame(W, H, 0))).cuda()
c = pa.Codec(W, H, wl=wl, lut_folder=os.path.join(orc.LUT_DIR, "n1_lossless"), k=k)
s = c.encode_frame(frame).clone()
out = torch.empty(c.max_stream_shorts(), dtype=torch.int16, device="cuda")
for _ in range(12):
    c.encode_frame_async(frame, out, 0)
torch.cuda.synchronize()
for _ in range(12):
    d = c.decode_frame(s)
torch.cuda.synchronize()
print("k", k, "ok", bool(torch.equal(d, frame.view(c.ah, c.aw))))
