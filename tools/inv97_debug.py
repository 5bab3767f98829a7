#!/usr/bin/env python3
"""Repeated decodes of the wl = 6 9/7 test frame against the oracle's pixels (race hunting)."""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "cuda-image-and-video-codec_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle_lib as orc, picsong_amd as pa
W, H, wl, qs = 2048, 2048, 6, 0.5
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
img = orc.gen_frame(W, H, 0)
lut = orc.lut_for(True, wl)
ref = orc.encode_frame(img, wl, True, qs, lut)
want = orc.decode_frame(ref, W, H, wl, True, qs, lut)
c = pa.Codec(W, H, wl=wl, lossy=True, qs=qs, lut_folder=os.path.join(orc.LUT_DIR, "n1_lossy"))
s = torch.from_numpy(ref.view(np.int16)).cuda()
bad = []
for i in range(N):
    got = c.decode_frame(s).cpu().numpy()
    d = np.argwhere(got != want)
    if len(d):
        bad.append((i, len(d), d[:, 0].min(), d[:, 0].max(), d[:, 1].min(), d[:, 1].max()))
print(os.path.basename(os.environ.get("PICSONG_SO", "default")), "INV97=" + os.environ.get("PICSONG_DWT_INV97", "-"),
      "REPLAY=" + os.environ.get("PICSONG_DWT_EXACT_REPLAY", "-"), "failures %d/%d" % (len(bad), N), bad[:6])
