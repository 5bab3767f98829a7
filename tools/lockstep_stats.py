#!/usr/bin/env python3
"""How full the encoder's lock-step call sites are (CPU oracle, test infrastructure): per codeblock, the call sites at
which at least one of the 32 lanes codes a symbol, the lanes that code there, and the sites that start a codeword --
the figures behind DESIGN.md's estimate of a per-lane free-running coder.
usage: tools/lockstep_stats.py [W H wl]   (default: the bench's 8K lossless frame)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as orc

W, H, wl = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (7680, 4320, 5)
L = orc.lib()
orc.set_threads(orc.usable_threads())
img = orc.gen_frame(W, H, 0)
L.po_stats_reset()
s = orc.encode_frame(img, wl, False, 1.0, orc.lut_for(False, wl))
st = (C.c_ulonglong * 3)()
L.po_stats_get(st)
sites, lanes, starts = st[0], st[1], st[2]
ncb = (orc.pad_dim(W) // 64) * (orc.pad_dim(H) // 64)
print(f"{W}x{H} wl {wl}: {ncb} codeblocks, {s.size} shorts")
print(f"call sites with a coding lane: {sites} ({sites / ncb:.0f} per codeblock); coding lanes: {lanes} "
      f"({lanes / sites:.2f} of 32 per site = {lanes / sites / 32:.3f}); symbols per codeblock {lanes / ncb:.0f} "
      f"= {lanes / ncb / 32:.0f} per lane; sites that start a codeword: {starts} ({starts / sites:.3f})")
