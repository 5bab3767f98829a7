#!/usr/bin/env python3
"""Level-by-level comparison of the 9/7 synthesis work buffer: runs of picsong_dwt_inverse against a reference
buffer (`save` under PICSONG_DWT_INV97=0, i.e. the other kernel family, then `cmp N`)."""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "cuda-image-and-video-codec_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle_lib as orc, picsong_amd as pa
W, H, wl, qs = 2048, 2048, 6, 0.5
img = orc.gen_frame(W, H, 0)
lut = orc.lut_for(True, wl)
ref = orc.encode_frame(img, wl, True, qs, lut)
c = pa.Codec(W, H, wl=wl, lossy=True, qs=qs, lut_folder=os.path.join(orc.LUT_DIR, "n1_lossy"))
st, sz = c.bitstream_unpack(torch.from_numpy(ref.view(np.int16)).cuda())
co = c.bpc_decode(st, sz)
path = os.path.join(ROOT, "gpurun_out", "inv97_ref.npy")
if sys.argv[1] == "save":
    np.save(path, c.dwt_inverse(co).cpu().numpy().view(np.uint32))
    print("saved")
    sys.exit(0)
want = np.load(path)
N = int(sys.argv[2])
# work buffer: levels wl-1 .. 0 outputs one after the other (level l output: (W >> l) x (H >> l))
offs = []
o = 0
for l in range(wl - 1, -1, -1):
    w, h = W >> l, H >> l
    offs.append((l, o, w, h)); o += w * h
nbad = 0
for i in range(N):
    got = c.dwt_inverse(co).cpu().numpy().view(np.uint32)
    if np.array_equal(got, want):
        continue
    nbad += 1
    if nbad > 4:
        continue
    msg = []
    for l, o, w, h in offs:
        d = np.argwhere((got[o:o + w * h] != want[o:o + w * h]).reshape(h, w))
        if len(d) and l == 1:
            g = got[o:o + w * h].reshape(h, w).view(np.float32); wv = want[o:o + w * h].reshape(h, w).view(np.float32)
            for y, x in d[:10]:
                print("    L1 bad (%d,%d): got %r want %r" % (y, x, float(g[y, x]), float(wv[y, x])))
            ys = sorted(set(d[:, 0].tolist())); print("    L1 bad rows:", ys[:40])
        if len(d):
            msg.append("L%d: %d bad rows %d-%d cols %d-%d" % (l, len(d), d[:, 0].min(), d[:, 0].max(), d[:, 1].min(), d[:, 1].max()))
    print(" run", i, "; ".join(msg))
print(os.path.basename(os.environ.get("PICSONG_SO", "default")), "failures %d/%d" % (nbad, N))
