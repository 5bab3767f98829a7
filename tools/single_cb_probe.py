#!/usr/bin/env python3
"""How much shorter a coder wave's chain is with ONE codeblock instead of two: the stage-level coder call on a frame's
coefficients and on the same with every second codeblock emptied (its wave partner then codes nothing).  GPU box."""
import os, sys, time
sys.path.insert(0, "cuda-image-and-video-codec_amd/python"); sys.path.insert(0, "tests")
import torch, oracle_lib as orc, picsong_amd as pa
for (W, H) in ((3840, 2160), (7680, 4320)):
    c = pa.Codec(W, H, wl=5, lossy=False, qs=1.0, lut_folder=os.path.join(orc.LUT_DIR, "n1_lossless"))
    frame = torch.from_numpy(orc.pad_frame(orc.gen_frame(W, H, 0))).cuda()
    coef = c.dwt_forward(frame)
    full = coef[:c.P].clone()
    half = full.view(c.ah, c.aw).clone()
    # every second codeblock of a row (odd cbx) emptied: each wave then codes ONE codeblock beside an empty one
    hv = half.view(c.ah, c.aw // 128, 2, 64)
    hv[:, :, 1, :] = 0
    half = half.view(-1)
    staging = torch.empty(c.P, dtype=torch.int32, device="cuda"); sizes = torch.empty(c.ncb, dtype=torch.int32, device="cuda")
    def t(x):
        for _ in range(3): pa._check(c.L.picsong_bpc_encode(c.h, c._p(x), c._p(staging), c._p(sizes), c._stream()))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): pa._check(c.L.picsong_bpc_encode(c.h, c._p(x), c._p(staging), c._p(sizes), c._stream()))
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / 20
    a, b = t(full), t(half)
    print(f"{W}x{H}: picsong_bpc_encode (memset + coder, 32-bit coefficients + widen) full {a*1e3:.3f} ms, every second codeblock empty {b*1e3:.3f} ms")
    c.close()
