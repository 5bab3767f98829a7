#!/usr/bin/env python3
"""What the other modes cost: a lone 8K frame (one stream, resident buffers) and three calls in flight, for the plain
two-pass coder, -k 0.5 / -k 1.5 (complexity-scalable bulk scan) and -cp 3 (three coding passes).  Run on the GPU box."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-image-and-video-codec_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

import oracle_lib as orc
import picsong_amd as pa

W, H, wl = 7680, 4320, 5
frame = torch.from_numpy(orc.pad_frame(orc.gen_frame(W, H, 0))).cuda()
sts = [torch.cuda.Stream() for _ in range(3)]
only = sys.argv[1] if len(sys.argv) > 1 else ""             # e.g. "cp 3": that mode's line alone
for name, kw in (("-cp 2, k = 0", {}), ("-k 0.5", {"k": 0.5}), ("-k 1.5", {"k": 1.5}), ("-cp 3", {"cp": 3})):
    if only not in name:
        continue
    lut = orc.LUT_CP3_DIR + "/n1_lossless" if kw.get("cp") == 3 else os.path.join(orc.LUT_DIR, "n1_lossless")
    if not os.path.isdir(lut):
        lut = orc.LUT_CP3_DIR
    # a context for the lone frame, three hinted ones (picsong_ctx_set_pipelined) for the calls in flight
    c = pa.Codec(W, H, wl=wl, lut_folder=lut, **kw)
    cs = [pa.Codec(W, H, wl=wl, lut_folder=lut, pipelined=True, **kw) for _ in range(3)]
    s = c.encode_frame(frame).clone()
    ok = bool(torch.equal(c.decode_frame(s), frame.view(c.ah, c.aw)))
    res = []
    outs = [torch.empty(c.max_stream_shorts(), dtype=torch.int16, device="cuda") for _ in range(3)]
    # (encode_frame_async: no wait for the length, as the bench's loop)
    for fn in (lambda k=-1: (c if k < 0 else cs[k]).encode_frame_async(frame, outs[max(k, 0)], 0),
               lambda k=-1: (c if k < 0 else cs[k]).decode_frame(s)):
        for k in (-1, 0, 1, 2, -1):                          # (every context's workspace exists before anything is timed)
            fn(k)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(15):
            fn()
        torch.cuda.synchronize(); lone = (time.perf_counter() - t0) / 15
        for rep in range(2):                                # (the first round creates the streams' hardware queues)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for i in range(150):
                with torch.cuda.stream(sts[i % 3]):
                    fn(i % 3)
            torch.cuda.synchronize(); pipe = (time.perf_counter() - t0) / 150
        res.append((lone, pipe))
    px = W * H / 1e9
    print(f"{name:14s} {s.numel() * 2 / 1e6:6.1f} MB  encode lone {res[0][0] * 1e3:.3f} ms = {px / res[0][0]:.0f} Gpixel/s, three in flight "
          f"{px / res[0][1]:.0f};  decode lone {res[1][0] * 1e3:.3f} ms = {px / res[1][0]:.0f}, three in flight {px / res[1][1]:.0f};  round trip {ok}")
    for x in cs + [c]:
        x.close()
