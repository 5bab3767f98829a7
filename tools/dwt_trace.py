#!/usr/bin/env python3
"""Time-resolved trace of the fused forward-DWT head (levels 0 + 1), from a library variant built with
-DPICSONG_DWT_TRACE (PICSONG_SO=...): every wave stamps s_memrealtime (100 MHz) at entry, after issuing its
loads, after the first row is unpacked, at mid-band, after its last store is issued and after its stores
are acknowledged.  Prints the distribution of each phase over the waves of one launch, and forward-DWT
timings (HIP events, frames rotating over a pool larger than the Infinity Cache).
usage: PICSONG_SO=.../variants/trace.so python tools/dwt_trace.py [lossy] [--frames=N per call]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-image-and-video-codec_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import oracle_lib as orc
import picsong_amd as pa

lossy = "lossy" in sys.argv[1:]
W, H = 7680, 4320
wl, qs = (6, 0.5) if lossy else (5, 1.0)
lut = os.path.join(orc.LUT_DIR, "n1_lossy" if lossy else "n1_lossless")
c = pa.Codec(W, H, wl=wl, lossy=lossy, qs=qs, lut_folder=lut)
pool = [torch.from_numpy(orc.pad_frame(orc.gen_frame(W, H, i))).cuda() for i in range(12)]
out = torch.zeros(c.P + c.extra, dtype=c.dtype, device="cuda")


def fwd(i):
    pa._check(c.L.picsong_dwt_forward_u8(c.h, c._p(pool[i % len(pool)]), c._p(out), c._stream()))


for i in range(6):
    fwd(i)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
ev[0].record()
for i in range(40):
    fwd(i)
    ev[i + 1].record()
torch.cuda.synchronize()
ts = sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(40))
print(f"forward DWT ({'9/7 wl 6' if lossy else '5/3 wl 5'}), all levels, per frame: median {ts[20]:.1f} us, min {ts[0]:.1f} us")

if hasattr(c.L, "picsong_debug_set_trace"):
    nslots = 8 * 4 * 64 * 4096
    buf = torch.zeros(nslots, dtype=torch.int64, device="cuda")
    c.L.picsong_debug_set_trace.argtypes = [C.c_void_p]
    assert c.L.picsong_debug_set_trace(C.c_void_p(buf.data_ptr())) == 0
    fwd(3)
    torch.cuda.synchronize()
    assert c.L.picsong_debug_set_trace(C.c_void_p(0)) == 0
    t = buf.cpu().numpy().reshape(-1, 8)
    t = t[t[:, 0] != 0][:, :6].astype(np.float64)
    t0 = t[:, 0].min()
    t = (t - t0) / 100.0                           # us since the first wave started
    names = ["start", "loads issued", "first row unpacked", "mid band", "last store issued", "stores acknowledged"]
    print(f"{len(t)} waves; us since the first wave's start: min / p10 / median / p90 / max")
    for k, n in enumerate(names):
        q = np.percentile(t[:, k], [0, 10, 50, 90, 100])
        print(f"  {n:22s} " + " / ".join(f"{v:6.2f}" for v in q))
    d = np.diff(t, axis=1)
    print("per-wave phase durations (us): min / median / max")
    for k in range(5):
        q = np.percentile(d[:, k], [0, 50, 100])
        print(f"  {names[k]:>20s} -> {names[k + 1]:22s} " + " / ".join(f"{v:6.2f}" for v in q))
    order = np.argsort(t[:, 0])
    print("start-time histogram (us): ", np.histogram(t[:, 0], bins=8)[0].tolist())
    print("end-time histogram (us):   ", np.histogram(t[:, 5], bins=8)[0].tolist(), "range", t[:, 5].min(), t[:, 5].max())
