#!/usr/bin/env python3
"""Time-resolved trace of the BPC encoder's waves (library variant built with -DPICSONG_DWT_TRACE, PICSONG_SO=...):
every wave stamps s_memrealtime at its start, when its planes are parked and when its plane loop is done, and leaves its
plane count.  Prints the distribution of the waves' durations over one lone 8K launch -- what a lone frame's coder
time is made of (the launch lasts as long as its slowest wave).
usage: PICSONG_SO=.../variants/trace.so python tools/bpc_trace.py [lossy] [4k]"""
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
DEC = "dec" in sys.argv[1:]                     # the decoder's waves: start, plane loop, epilogue
W, H = (3840, 2160) if "4k" in sys.argv[1:] else (7680, 4320)
wl, qs = (6, 0.5) if lossy else (5, 1.0)
lut = os.path.join(orc.LUT_DIR, "n1_lossy" if lossy else "n1_lossless")
c = pa.Codec(W, H, wl=wl, lossy=lossy, qs=qs, lut_folder=lut)
frame = torch.from_numpy(orc.pad_frame(orc.gen_frame(W, H, 0))).cuda()
for _ in range(3):
    c.encode_frame(frame)
import time
torch.cuda.synchronize()
t_0 = time.perf_counter()
for _ in range(50):
    s_ = c.encode_frame(frame)
torch.cuda.synchronize()
print(f"lone-frame encode (incl. the host's wait for the length): {(time.perf_counter() - t_0) / 50 * 1e3:.3f} ms;  round trip {bool(torch.equal(c.decode_frame(s_), frame.view(c.ah, c.aw))) if not lossy else None}")
nw = c.ncb                                      # (a variant with one codeblock per wave has that many)
buf = torch.zeros(4 * (nw + 8), dtype=torch.int64, device="cuda")
c.L.picsong_debug_set_bpc_trace.argtypes = [C.c_void_p]
assert c.L.picsong_debug_set_bpc_trace(C.c_void_p(buf.data_ptr())) == 0
if DEC:
    torch.cuda.synchronize()
    assert c.L.picsong_debug_set_bpc_trace(C.c_void_p(0)) == 0
    for _ in range(3):
        c.decode_frame(s_)
    torch.cuda.synchronize()
    t_0 = time.perf_counter()
    for _ in range(50):
        c.decode_frame(s_)
    torch.cuda.synchronize()
    print(f"lone-frame decode: {(time.perf_counter() - t_0) / 50 * 1e3:.3f} ms")
    buf.zero_()
    assert c.L.picsong_debug_set_bpc_trace(C.c_void_p(buf.data_ptr())) == 0
    c.decode_frame(s_)
    torch.cuda.synchronize()
    assert c.L.picsong_debug_set_bpc_trace(C.c_void_p(0)) == 0
    t = buf.cpu().numpy().reshape(-1, 4)[:nw]
    t = t[t[:, 0] != 0]
    npl = (t[:, 1] & 255).astype(np.float64)
    t = t.astype(np.float64)
    t[:, 1] = np.floor(t[:, 1] / 256.0)
    t0 = t[:, 0].min()
    start, loop0, loop1, end = [(t[:, k] - t0) / 100 for k in range(4)]
    print(f"decoder, {W}x{H} {'9/7 wl 6' if lossy else '5/3 wl 5'}: {len(t)} waves; the launch's last wave is done at {end.max():.1f} us")
    q = lambda v: " / ".join(f"{x:7.1f}" for x in np.percentile(v, [0, 10, 50, 90, 99, 100]))
    print("us, min / p10 / median / p90 / p99 / max")
    print("  start          ", q(start))
    print("  prologue       ", q(loop0 - start))
    print("  plane loop     ", q(loop1 - loop0))
    print("  epilogue       ", q(end - loop1))
    print("  whole wave     ", q(end - start))
    for k in sorted(set(npl.astype(int))):
        m = npl == k
        print(f"  waves with {k:2d} planes: {int(m.sum()):5d}, plane loop median {np.median((loop1 - loop0)[m]):6.1f} max {(loop1 - loop0)[m].max():6.1f}; "
              f"prologue median {np.median((loop0 - start)[m]):5.1f}; epilogue median {np.median((end - loop1)[m]):5.1f}; wave done median {np.median(end[m]):6.1f} max {end[m].max():6.1f} us")
    slow = np.argsort(-end)[:8]
    print("the last eight waves (wave, planes, start, prologue, loop, epilogue, done):")
    for w in slow:
        print(f"  {w:5d}  {int(npl[w]):2d}  {start[w]:6.1f}  {(loop0 - start)[w]:6.1f}  {(loop1 - loop0)[w]:6.1f}  {(end - loop1)[w]:6.1f}  {end[w]:6.1f}")
    sys.exit(0)
c.encode_frame(frame)
torch.cuda.synchronize()
assert c.L.picsong_debug_set_bpc_trace(C.c_void_p(0)) == 0
t = buf.cpu().numpy().reshape(-1, 4)[:nw]
t = t[t[:, 0] != 0].astype(np.float64)
nw = len(t)
t0 = t[:, 0].min()
start, parked, done, npl = (t[:, 0] - t0) / 100, (t[:, 1] - t0) / 100, (t[:, 2] - t0) / 100, t[:, 3]
dur = done - start
print(f"{W}x{H} {'9/7 wl 6' if lossy else '5/3 wl 5'}: {nw} waves; the launch's last wave is done at {done.max():.1f} us")
q = lambda v: " / ".join(f"{x:7.1f}" for x in np.percentile(v, [0, 10, 50, 90, 99, 100]))
print("us, min / p10 / median / p90 / p99 / max")
print("  start          ", q(start))
print("  prologue       ", q(parked - start))
print("  plane loop     ", q(done - parked))
print("  whole wave     ", q(dur))
print("  planes per wave", q(npl))
for k in sorted(set(npl.astype(int))):
    m = npl == k
    print(f"  waves with {k:2d} planes: {int(m.sum()):5d}, duration median {np.median(dur[m]):6.1f} max {dur[m].max():6.1f} us")
slow = np.argsort(-done)[:8]
print("the last eight waves (wave, first codeblock, planes, start, duration):")
for w in slow:
    print(f"  {w:5d}  cb {2 * w:5d}  {int(npl[w]):2d}  {start[w]:6.1f}  {dur[w]:6.1f}")
print(f"sum of wave durations / (launch x 5120 wave slots) = {dur.sum() / (done.max() * 5120):.2f}")
