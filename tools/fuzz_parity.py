#!/usr/bin/env python3
"""Random geometries through the frame paths against the oracle (GPU box): encode == the oracle's codestream, decode ==
the oracle's pixels, for random W x H (ragged, padded by the caller as the CLI does), wl, 5/3 and 9/7, the coder's modes
(two passes; -k > 0 with and without the pipelined hint; -cp 3), single frames and batched calls.  A last safety net after changes to the coders' boundaries; not part of the test suite.
usage: tools/fuzz_parity.py [n_cases] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-image-and-video-codec_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import oracle_lib as orc
import picsong_amd as pa

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
orc.set_threads(orc.usable_threads())
bad = 0
for case in range(n_cases):
    W = int(rng.integers(65, 2200)); H = int(rng.integers(65, 1400))
    lossy = bool(rng.integers(0, 2))
    wl = int(rng.integers(1, 7))
    while (orc.pad_dim(W) >> wl) < 2 or (orc.pad_dim(H) >> wl) < 2:
        wl -= 1
    qs = float(rng.choice([1.0, 0.5, 0.25])) if lossy else 1.0
    kind = int(rng.integers(0, 4))
    if kind == 0:
        img = orc.gen_frame(W, H, int(rng.integers(0, 100)))
    elif kind == 1:
        img = rng.integers(0, 256, (H, W), dtype=np.uint8)           # noise: raw fallback blocks
    elif kind == 2:
        img = np.full((H, W), int(rng.integers(0, 256)), np.uint8)   # flat: empty codeblocks
    else:
        img = orc.gen_frame(W, H, 3); img[: H // 2] = rng.integers(0, 256, (H // 2, W), dtype=np.uint8)
    mode = int(rng.integers(0, 10))                # 0-4 two passes, 5-7 -k > 0, 8-9 -cp 3
    k, cp, hint = 0.0, 2, False
    sub = "n1_lossy" if lossy else "n1_lossless"
    lut, lutdir = orc.lut_for(lossy, wl), os.path.join(orc.LUT_DIR, sub)
    if 5 <= mode <= 7:
        k, hint = float(rng.choice([0.3, 0.5, 0.7, 0.9])), bool(rng.integers(0, 2))
        lut = orc.lut_for_k(lossy, wl)
    elif mode >= 8:
        cp, lut, lutdir = 3, orc.lut_for_cp3(lossy, wl), os.path.join(orc.LUT_CP3_DIR, sub)
    try:
        ref = orc.encode_frame(img, wl, lossy, qs, lut, 0, 0, k=k)
        c = pa.Codec(W, H, wl=wl, lossy=lossy, qs=qs, lut_folder=lutdir, k=k, cp=cp, pipelined=hint)
        frame = torch.from_numpy(orc.pad_frame(img)).cuda()
        s = c.encode_frame(frame)
        ok_enc = s.numel() == ref.size and np.array_equal(s.cpu().numpy().view(np.uint16), ref)
        d = c.decode_frame(s.clone())
        refpix = orc.decode_frame(ref, W, H, wl, lossy, qs, lut, k=k)
        got = d.cpu().numpy()[:H, :W]
        ok_dec = np.array_equal(got, refpix[:H, :W]) and (lossy or np.array_equal(got, img))
        # batched: three copies of the frame through encode_frames / decode_frames
        ok_b = True
        if c.ncb <= 4096 and cp == 2:                  # (the batched calls: -cp 2, any k)
            frames = torch.stack([frame.view(-1)] * 3)
            out = torch.empty((3, c.max_stream_shorts()), dtype=torch.int16, device="cuda")
            c.encode_frames_async(frames, out, 1)
            torch.cuda.synchronize()
            ok_b = bool(torch.equal(out[1, 9:s.numel()], s[9:]))      # (frames 1.. of a video carry no header)
        c.close()
    except Exception as e:                                      # noqa: BLE001
        print(f"case {case}: {W}x{H} wl {wl} lossy {lossy} qs {qs} kind {kind} k {k} cp {cp} hint {hint}: EXCEPTION {e!r}")
        bad += 1
        continue
    flag = "" if (ok_enc and ok_dec and ok_b) else "   <-- MISMATCH"
    bad += 0 if not flag else 1
    print(f"case {case}: {W}x{H} wl {wl} lossy {lossy} qs {qs} kind {kind} k {k} cp {cp} hint {hint}: enc {ok_enc} dec {ok_dec} batched {ok_b}{flag}")
print("mismatches:", bad)
sys.exit(1 if bad else 0)
