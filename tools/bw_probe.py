#!/usr/bin/env python3
"""Measured device-copy ceilings for the DWT roofline ("use the measured device copy
bandwidth as the roof and state both").  Plain torch elementwise kernels with the same traffic shape
as the DWT levels: copy (4 B in, 4 B out), widen (1 B in, 4 B out = level 0 with u8 ingest), fill."""
import json
import sys
import torch


def timed(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def probe(P=7680 * 4352, device="cuda:0"):
    torch.cuda.set_device(device)
    out = {}
    a = torch.randint(-1000, 1000, (P,), dtype=torch.int32, device=device)
    b = torch.empty_like(a)
    u = torch.randint(0, 255, (P,), dtype=torch.uint8, device=device)
    t = timed(lambda: b.copy_(a))
    out["copy_i32"] = {"us": t * 1e6, "GBps": 8 * P / t / 1e9}
    t = timed(lambda: b.copy_(u))
    out["widen_u8_to_i32"] = {"us": t * 1e6, "GBps": 5 * P / t / 1e9}
    t = timed(lambda: b.fill_(7))
    out["fill_i32"] = {"us": t * 1e6, "GBps": 4 * P / t / 1e9}
    s = torch.empty((), dtype=torch.int64, device=device)
    t = timed(lambda: torch.sum(a, dim=(0,), dtype=torch.int64, out=s))
    out["read_i32"] = {"us": t * 1e6, "GBps": 4 * P / t / 1e9}
    # quarter-size (level 1 domain) copy: launch-bound or bandwidth-bound?
    q = P // 4
    t = timed(lambda: b[:q].copy_(a[:q]))
    out["copy_i32_quarter"] = {"us": t * 1e6, "GBps": 8 * q / t / 1e9}
    return out


if __name__ == "__main__":
    json.dump(probe(), sys.stdout, indent=1)
    print()
