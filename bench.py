#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native PICSONG hot path.

Metric (BASELINE.json): Mpixels/s, encode (level shift + DWT + BPC + pack), 8K greyscale frame,
-type 0 (5/3 lossless), device-resident u8 frame in -> device-resident uint16 codestream out;
decode must round-trip bit-exactly (checked outside the timed region, reported as roundtrip_ok).

A "step" = one 7680x4320 frame per rank through picsong_encode_frame.  With N > 1 (launched by
torch.distributed.run, one rank per GPU) frames are sharded one per rank per step ("weak"), and
the step ends with the path's only exchange: codestream lengths all-gathered and payloads gathered
to rank 0 over RCCL; value = pixels all ranks encoded / max-over-ranks time.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "cuda-image-and-video-codec_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

WORKLOADS = {
    # name: (W, H, wl, lossy, qs)
    "8k_lossless": (7680, 4320, 5, False, 1.0),     # metric workload ("headline run")
    "4k_lossless": (3840, 2160, 5, False, 1.0),     # configs[1]
    "8k_lossy": (7680, 4320, 6, True, 0.5),         # configs[2]
}


def dwt_bytes(P, wl, s0):
    """P*(s0+4) + 8*P*sum_{l=1}^{wl-1} 4^-l."""
    return P * (s0 + 4) + 8 * P * sum(4.0 ** -l for l in range(1, wl))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="8k_lossless", choices=sorted(WORKLOADS))
    ap.add_argument("--streams", type=int, default=2,
                    help="HIP streams (each with its own context) the frames of consecutive steps alternate on")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-rows", type=int, default=0,
                    help="rows of the frame the CPU baseline encodes (0 = whole frame)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with "
                  f"python -m torch.distributed.run --nproc-per-node {args.gpus} ...", file=sys.stderr)
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the picsong HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import picsong_amd as pa
    import picsong_dist as pdist
    import oracle_lib as orc          # checker + cpu_baseline only

    W, H, wl, lossy, qs = WORKLOADS[args.workload]
    lut_dir = os.path.join(orc.LUT_DIR, "n1_lossy" if lossy else "n1_lossless")
    nstreams = max(1, args.streams)
    codecs = [pa.Codec(W, H, wl=wl, lossy=lossy, qs=qs, lut_folder=lut_dir, device=local_rank)
              for _ in range(nstreams)]
    streams = [torch.cuda.Stream(device=local_rank) for _ in range(nstreams)]
    codec = codecs[0]
    AW, AH, nCB, P = codec.aw, codec.ah, codec.ncb, codec.P

    # synthetic frames (SURVEY 8d generator), one distinct frame per rank, resident in HBM
    frame_np = orc.pad_frame(orc.gen_frame(W, H, rank))
    frame = torch.from_numpy(frame_np).cuda()
    outs = [torch.empty(codec.max_stream_shorts(), dtype=torch.int16, device="cuda") for _ in range(nstreams)]
    out = outs[0]
    gather_bufs = None
    if world > 1 and rank == 0:
        gather_bufs = [torch.empty(codec.max_stream_shorts(), dtype=torch.int16, device="cuda")
                       for _ in range(world - 1)]
    dev = torch.device("cuda", local_rank)

    def step(it):
        # consecutive frames alternate over the streams: frame i's BPC tail overlaps frame i+1's
        # DWT/BPC head (each stream has its own context = its own workspace)
        k = it % nstreams
        with torch.cuda.stream(streams[k]):
            codecs[k].encode_frame_async(frame, outs[k], 0 if it == 0 else 1)
            if world > 1:
                # the only exchange of the frame-sharded path (picsong_dist.gather_round, covered by
                # the gloo tests): lengths all-gathered, then payload gatherv to rank 0 over RCCL
                total = codecs[k].last_total()
                pdist.gather_round(outs[k][:total], rank, world, dev, recv_bufs=gather_bufs)

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    sync_all()
    for c in codecs:
        c.profile_begin(args.steps)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(1 + i)
    sync_all()
    dt = time.perf_counter() - t0
    stage_ms = np.concatenate([c.profile_read(args.steps) for c in codecs], axis=0)
    for c in codecs:
        c.profile_begin(0)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    total_shorts = codec.last_total()
    flag = codec.range_flag()

    # ---- correctness outside the timed region: decode(encode(x)) == x
    stream0 = codec.encode_frame(frame, 0)
    dec = codec.decode_frame(stream0)
    if lossy:
        a = dec[:H, :W].float()
        b = frame.view(AH, AW)[:H, :W].float()
        mse = torch.mean((a - b) ** 2).item()
        psnr = 10 * np.log10(255.0 ** 2 / max(mse, 1e-12))
        roundtrip_ok = bool(psnr >= 40.0)
    else:
        psnr = None
        roundtrip_ok = bool(torch.equal(dec, frame.view(AH, AW)))

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    ms_per_step = dt / args.steps * 1e3
    mpix = (W * H * world * args.steps) / dt / 1e6

    # ---- roofline of the dominant kernel (bpc_kernel<false>): algorithmic bytes per launch
    # (nCB*16384 coefficient bytes + 4*nCB sizes + 2*sum(ncw) codeword bytes) over the
    # kernel's mean duration measured with HIP events on the launch stream inside the timed region.
    sizes = stream0.cpu().numpy().view(np.uint16)[10:10 + 2 * nCB:2].astype(np.int64)
    ncw = int((sizes - 1).sum())
    bpc_bytes = nCB * 16384 + 4 * nCB + 2 * ncw
    dwt_ms, bpc_ms, pack_ms = [float(x) for x in stage_ms.mean(axis=0)]
    bpc_gbs = bpc_bytes / (bpc_ms * 1e-3) / 1e9
    dwt_b = dwt_bytes(P, wl, 1)
    roofline = {"kernel": "bpc_kernel<false> (BPC-PaCo encode)", "bound": "hbm",
                "achieved": round(bpc_gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(bpc_gbs / HBM_PEAK_GBS, 5), "traffic": None,
                "algorithmic_bytes_per_launch": bpc_bytes, "avg_launch_ms": round(bpc_ms, 4),
                "codeblocks_per_s": round(nCB / (bpc_ms * 1e-3), 1),
                "note": "BPC is integer/latency-bound, not HBM-bound (SURVEY 8d): codeblocks/s is "
                        "the figure of merit; the HBM fraction is reported for completeness"}
    roofline_dwt = {"kernel": "dwt_fwd_kernel (all levels, u8 ingest fused)", "bound": "hbm",
                    "achieved": round(dwt_b / (dwt_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(dwt_b / (dwt_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                    "traffic": None, "algorithmic_bytes_per_launch": int(dwt_b),
                    "avg_launch_ms": round(dwt_ms, 4)}

    # ---- CPU baseline: the oracle (scalar C port), bounded sample, rank 0, N = 1 only
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        rows = H if args.cpu_sample_rows <= 0 else min(H, max(64 << (wl - 1), (args.cpu_sample_rows // 64) * 64))
        sample = np.ascontiguousarray(orc.gen_frame(W, H, 0)[:rows])
        lut = orc.lut_for(lossy, wl)
        t1 = time.perf_counter()
        ref_stream = orc.encode_frame(sample, wl, lossy, qs, lut, 0, 0)
        cdt = time.perf_counter() - t1
        cpu = {"value": round(W * rows / cdt / 1e6, 3), "unit": "Mpixels/s", "cores": 1, "kind": "port",
               "sample": f"1 frame {W}x{rows} of the workload, full encode (pad+shift+DWT+BPC+pack), "
                         f"oracle/picsong_oracle.c -O3, {cdt:.1f} s",
               "host_cpu_count": os.cpu_count()}
        if rows == H:
            cpu["codestream_matches_gpu"] = bool(np.array_equal(
                ref_stream, stream0.cpu().numpy().view(np.uint16)))

    line = {
        "metric": "Mpixels/s encode (DWT+BPC) 8K P5 lossless; round-trip bit-exact"
                  if args.workload == "8k_lossless" else f"Mpixels/s encode (DWT+BPC) {args.workload}",
        "value": round(mpix, 2), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32" if lossy else "int32", "data": "synthetic",
        "config": {"workload": f"{W}x{H} greyscale u8 frame (padded {AW}x{AH}), "
                               f"-type {int(lossy)} {'9/7 qs=%g' % qs if lossy else '5/3 lossless'}, "
                               f"wl={wl}, cp=2, k=0, LUT {'n1_lossy' if lossy else 'n1_lossless'}, "
                               f"1 frame/step/GPU, frames sharded over {world} GPU(s), "
                               f"{nstreams} HIP stream(s) per GPU",
                   "codeblocks": nCB, "stream_shorts": int(total_shorts),
                   "bits_per_pixel": round(total_shorts * 16 / (W * H), 4)},
        "roundtrip_ok": roundtrip_ok, "range_flag": flag,
        "stage_ms": {"dwt": round(dwt_ms, 4), "bpc": round(bpc_ms, 4), "pack": round(pack_ms, 4)},
        "roofline": roofline, "roofline_dwt": roofline_dwt, "cpu_baseline": cpu,
    }
    if psnr is not None:
        line["psnr_db"] = round(psnr, 3)
    print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
