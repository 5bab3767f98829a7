#!/usr/bin/env python3
"""9/7 frames of odd geometries through the frame path (encode, decode) against the oracle's codestream and pixels."""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "cuda-image-and-video-codec_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle_lib as orc, picsong_amd as pa
bad = 0
for (W, H, wl, qs) in [(64, 64, 1, 0.5), (128, 64, 2, 0.3), (1000, 300, 4, 0.5), (704, 576, 5, 0.7), (2000, 1100, 6, 0.5),
                       (320, 4000, 3, 0.25), (4100, 200, 2, 1.0), (1920, 1080, 5, 0.5), (832, 192, 4, 1.0)]:
    img = orc.gen_frame(W, H, 3)
    lut = orc.lut_for(True, wl)
    ref = orc.encode_frame(img, wl, True, qs, lut)
    want = orc.decode_frame(ref, W, H, wl, True, qs, lut)
    c = pa.Codec(W, H, wl=wl, lossy=True, qs=qs, lut_folder=os.path.join(orc.LUT_DIR, "n1_lossy"))
    s = c.encode_frame(torch.from_numpy(orc.pad_frame(img)).cuda()).cpu().numpy().view(np.uint16)
    ok_s = np.array_equal(s, ref)
    oks = [bool(np.array_equal(c.decode_frame(torch.from_numpy(ref.view(np.int16)).cuda()).cpu().numpy()[:H, :W], want)) for _ in range(5)]
    print(W, H, wl, qs, "stream", ok_s, "pixels", oks)
    bad += (not ok_s) + sum(1 for o in oks if not o)
    c.close()
print("FAILURES", bad)
sys.exit(1 if bad else 0)
