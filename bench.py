#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native PICSONG hot path.

Metric (BASELINE.json): Mpixels/s, encode (level shift + DWT + BPC + pack), 8K greyscale frame,
-type 0 (5/3 lossless), device-resident u8 frame in -> device-resident uint16 codestream out;
decode must round-trip bit-exactly (checked outside the timed region, reported as roundtrip_ok).

A "step" = one batch of --frames-per-step 7680x4320 frames per rank (default 360, about 58 ms of coding: the
driver's `--steps 20` times 1.15 s), taken from a pool of --pool distinct device-resident frames (default 16 = 535 MB at 8K, past the
256 MiB Infinity Cache, so that every frame's input comes from HBM) and coded --batch frames per call of
picsong_encode_frames, the calls alternating over --streams HIP streams.  With N > 1 (launched by
torch.distributed.run, one rank per GPU) every rank codes its own frames ("weak"), and each step ends
with the path's only exchange: codestream lengths all-gathered and payloads gathered over RCCL --
frame f of every rank's step to rank f mod N by default (--gather rotate: the writer role rotates, every xGMI
link carries 1/N of a step), or all of them to rank 0 (--gather root0: 99 GB/s per peer at 8K lossless, more
than one link direction carries); value = pixels all ranks encoded / max-over-ranks time.

`python bench.py --gpus N` with N > 1 and no torch.distributed environment starts the N ranks itself (a child
`python -m torch.distributed.run`, before this process touches the GPU) and relays rank 0's line; fewer than N GPUs
on the node is an error.  Prints ONE JSON line on rank 0.
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


def host_cores():
    """CPU cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def _profile_tag():
    """Newest round whose counter summaries are committed: profiles/<tag>_pmc_hbm.csv, <tag>_pmc_sq.csv,
    <tag>_valu_probe.json, <tag>_library.sha256 (tools/collect_profiles.sh + tools/publish_profiles.sh)."""
    for tag in ("r03", "r02"):
        if os.path.exists(os.path.join(ROOT, "profiles", tag + "_pmc_sq.csv")):
            return tag
    return "r03"


PROFILE_TAG = _profile_tag()


def library_hashes():
    """sha256 (first 16 hex digits) of the library this run loads and of the one the committed counter passes were
    taken from (profiles/<tag>_library.sha256): the OFFLINE figures of the line (`traffic`, `valu_issue`) describe the
    latter."""
    import hashlib
    so = os.environ.get("PICSONG_SO") or os.path.join(ROOT, "cuda-image-and-video-codec_amd", "csrc", "libpicsong_hip.so")
    try:
        mine = hashlib.sha256(open(so, "rb").read()).hexdigest()[:16]
    except OSError:
        mine = None
    try:
        prof = open(os.path.join(ROOT, "profiles", PROFILE_TAG + "_library.sha256")).read().split()[0][:16]
    except (OSError, IndexError):
        prof = None
    return {"library_sha256_16": mine, "profiles_tag": PROFILE_TAG, "profiles_library_sha256_16": prof,
            "profiles_match_library": (mine == prof) if (mine and prof) else None}



def pmc_traffic(workload, batch=1):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/*_pmc_hbm.csv: FETCH_SIZE and
    WRITE_SIZE collected in separate runs of this command with --streams 1 --batch 1, unit 1024 B;
    FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 note -- calibrated on the DWT kernels' known read
    bytes), scaled to the `batch` frames of a launch.  OFFLINE figures: they describe the build the
    profiles were taken from.  None when no summary for this workload is committed."""
    import csv
    path = os.path.join(ROOT, "profiles", PROFILE_TAG + "_pmc_hbm.csv")
    if workload != "8k_lossless" or not os.path.exists(path):
        return None, None
    bpc = dwt = 0.0
    frames = None
    rows = list(csv.DictReader(ln for ln in open(path) if not ln.startswith("#")))
    for r in rows:
        if "bpc_encode_kernel" in r["Kernel_Name"]:
            frames = int(r["Dispatches"])
    for r in rows:
        mult = 2.0 if r["Counter_Name"] == "FETCH_SIZE" else 1.0
        b = float(r["MeanValue"]) * 1024.0 * mult
        if "bpc_encode_kernel" in r["Kernel_Name"]:
            bpc += b
        elif "dwt_fwd" in r["Kernel_Name"] and frames:
            dwt += b * int(r["Dispatches"]) / frames          # all levels of one frame
    return (int(bpc * batch) if bpc else None), (int(dwt * batch) if dwt else None)


def pmc_valu(workload):
    """VALU wave-instructions per FRAME of the BPC encoder from the committed SQ counter pass
    (profiles/*_pmc_sq.csv, SQ_INSTS_VALU)."""
    import csv
    path = os.path.join(ROOT, "profiles", PROFILE_TAG + "_pmc_sq.csv")
    if workload != "8k_lossless" or not os.path.exists(path):
        return None
    for r in csv.DictReader(ln for ln in open(path) if not ln.startswith("#")):
        if "bpc_encode_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "SQ_INSTS_VALU":
            return float(r["MeanValue"])
    return None


def pmc_valu_busy(workload, insts_per_frame_all, step_s):
    """rocprof's VALUBusy for the coder -- 100 x sum(SQ_ACTIVE_INST_VALU) / CU_NUM / max(GRBM_GUI_ACTIVE): the share of
    the time in which a SIMD has a vector instruction in its pipe, one instruction = one quad-cycle of one of a CU's
    four SIMDs -- from the committed counter pass (profiles/*_pmc_sq_pipelined.csv, collected over the DEFAULT
    three-stream command).  rocprofv3 SERIALISES the dispatches of a --pmc run (the pass's own kernel trace shows no two
    kernels overlapping, against a hundred overlapping pairs in the plain trace of the same command), so the counters
    describe every kernel running ALONE whatever --streams says: `lone_kernel`.  `pipelined` is the same definition
    applied to the counter-measured instruction counts of all of a frame's kernels and this run's measured time per
    frame; values near or above 1 mean the vector ALUs are the limit (full-rate instructions take less than a
    quad-cycle: tools/valu_probe).  OFFLINE counters, like `traffic`."""
    import csv
    path = os.path.join(ROOT, "profiles", PROFILE_TAG + "_pmc_sq_pipelined.csv")
    if workload != "8k_lossless" or not os.path.exists(path):
        return None
    c = {}
    for r in csv.DictReader(ln for ln in open(path) if not ln.startswith("#")):
        if "bpc_encode_kernel" in r["Kernel_Name"]:
            c[r["Counter_Name"]] = float(r["MeanValue"])
    if not c.get("SQ_ACTIVE_INST_VALU") or not c.get("GRBM_GUI_ACTIVE"):
        return None
    xcc, cus, simds = 8, 256, 1024
    gui = c["GRBM_GUI_ACTIVE"] / xcc                      # (the summary adds the eight XCCs' values of a dispatch)
    out = {"source": "profiles/%s_pmc_sq_pipelined.csv" % PROFILE_TAG,
           "lone_kernel": {"VALUBusy": round(c["SQ_ACTIVE_INST_VALU"] / cus / gui, 4),
                           "gpu_cycles_per_dispatch": int(gui)}}
    wc = c.get("SQ_WAVE_CYCLES")
    if wc:
        for name, key in (("wave_cycles_issuing", "SQ_ACTIVE_INST_ANY"), ("wave_cycles_waiting_waitcnt", "SQ_WAIT_ANY"),
                          ("wave_cycles_waiting_issue", "SQ_WAIT_INST_ANY")):
            if key in c:
                out["lone_kernel"][name] = round(c[key] / wc, 4)
    if insts_per_frame_all and step_s:
        pr = probe_rates() or {}
        hz = (pr.get("shader_mhz") or 2400.0) * 1e6
        out["pipelined"] = {"VALUBusy_same_definition": round(insts_per_frame_all * 4.0 / (simds * step_s * hz), 4),
                            "valu_wave_insts_per_frame_all_kernels": int(insts_per_frame_all), "shader_hz": hz,
                            "note": "instruction counts from the counter passes (coder + transform + pack), time per "
                                    "frame from this run's timed loop"}
    return out


def pmc_valu_all(workload):
    """VALU wave-instructions per frame of ALL the encode path's kernels (coder, transform levels, scan, pack) from the
    committed SQ pass (profiles/*_pmc_sq.csv: mean per dispatch x dispatches per frame)."""
    import csv
    path = os.path.join(ROOT, "profiles", PROFILE_TAG + "_pmc_sq.csv")
    if workload != "8k_lossless" or not os.path.exists(path):
        return None
    rows = [r for r in csv.DictReader(ln for ln in open(path) if not ln.startswith("#")) if r["Counter_Name"] == "SQ_INSTS_VALU"]
    frames = next((int(r["Dispatches"]) for r in rows if "bpc_encode_kernel" in r["Kernel_Name"]), 0)
    if not frames:
        return None
    tot = 0.0
    for r in rows:
        if any(k in r["Kernel_Name"] for k in ("bpc_encode_kernel", "dwt_fwd", "scan_sizes", "pack_kernel")):
            tot += float(r["MeanValue"]) * int(r["Dispatches"]) / frames
    return tot


def probe_rates():
    """Issue rates measured by tools/valu_probe on an MI355X (profiles/*_valu_probe.json): wave64
    instructions per cycle per SIMD with 8 waves resident, for a full-rate and a half-rate class."""
    path = os.path.join(ROOT, "profiles", PROFILE_TAG + "_valu_probe.json")
    try:
        d = json.load(open(path))
    except (OSError, ValueError):
        return None
    out = {"shader_mhz": d.get("shader_mhz")}
    for pat in d["patterns"]:
        best = max(r["insts_per_cycle_per_simd"] for r in pat["rows"])
        if pat["pattern"].startswith("v_and_b32 x8"):
            out["full_rate"] = best
        elif pat["pattern"].startswith("v_alignbit_b32 dependent"):
            out["half_rate"] = best
        elif pat["pattern"].startswith("coder call site mix"):
            out["call_site_mix_valu"] = max(r["valu_per_cycle_per_simd"] for r in pat["rows"])
    return out


def valu_issue(insts, step_s, iso_s):
    """The encoder is bound by vector-instruction issue.  Peaks, wave64 instructions per second on 256 CUs x
    4 SIMDs at 2.4 GHz: (a) the guide's 2 cycles per instruction on a SIMD-32 (MI355X_MICROARCH.md l.53),
    (b) what tools/valu_probe measures on the box: and / or / add / mov on VGPR operands reach ~0.41 per
    cycle per SIMD, every other class this kernel is made of -- shifts, v_min, compares, every VOP3 form
    (v_alignbit, v_perm, v_bfe, v_mbcnt, v_cndmask_e64, v_mad), v_mul_u32_u24, DPP moves, any SGPR operand
    -- ~0.24, i.e. one per 4.2 cycles however many waves are resident."""
    if not insts:
        return None
    simd_hz = 256 * 4 * 2.4e9
    pr = probe_rates() or {}
    res = {"valu_wave_insts_per_frame": int(insts),
           "achieved_per_cycle_per_simd": {"single_stream": round(insts / iso_s / simd_hz, 4),
                                           "pipelined": round(insts / step_s / simd_hz, 4)},
           "peak_guide_per_cycle_per_simd": 0.5,
           "frac_of_guide_peak": {"single_stream": round(insts / iso_s / simd_hz / 0.5, 4),
                                  "pipelined": round(insts / step_s / simd_hz / 0.5, 4)},
           "source": "profiles/%s_pmc_sq.csv (SQ_INSTS_VALU, rocprofv3 --pmc pass, offline) / tools/valu_probe" % PROFILE_TAG}
    if "half_rate" in pr:
        res["probe"] = pr
        res["frac_of_probe_half_rate_peak"] = {"single_stream": round(insts / iso_s / simd_hz / pr["half_rate"], 4),
                                               "pipelined": round(insts / step_s / simd_hz / pr["half_rate"], 4)}
    return res


def dwt_bytes(P, wl, s0):
    """P*(s0+4) + 8*P*sum_{l=1}^{wl-1} 4^-l."""
    return P * (s0 + 4) + 8 * P * sum(4.0 ** -l for l in range(1, wl))


def free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def launch_ranks(args):
    """`python bench.py --gpus N` with no torch.distributed environment: start the N ranks ourselves, as a CHILD
    process (python -m torch.distributed.run, one rank per GPU) -- before this process has made any HIP call: a
    process that has touched the GPU must never be replaced or forked -- relay the ranks' output and exit with the
    child's code.  A node with fewer than N GPUs is an error, not a quiet world = 1 run."""
    import subprocess
    if not args.dry:
        import torch                                    # (device_count() does not initialise the GPU)
        have = torch.cuda.device_count()
        if have < args.gpus:
            print(f"bench.py: --gpus {args.gpus} but this node has {have} GPU(s)", file=sys.stderr)
            return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True)
    line = None
    for out in child.stdout:
        if out.lstrip().startswith("{"):
            line = out.strip()                          # rank 0's JSON line (printed once, below)
        else:
            sys.stderr.write(out)
    rc = child.wait()
    if rc == 0 and line is None:
        print("bench.py: the ranks exited without a result line", file=sys.stderr)
        rc = 1
    if line is not None:
        print(line, flush=True)
    return rc


def dry_main(args, rank, world):
    """--dry: the N-rank protocol of this benchmark without a GPU and without the codec (CPU tests of the launcher):
    rendezvous, per-step exchange of stand-in codestreams over picsong_dist.gather_step, barrier-bracketed timing,
    MAX over ranks, one JSON line on rank 0."""
    import torch
    import torch.distributed as dist
    import picsong_dist as pdist
    if world > 1:
        dist.init_process_group(args.backend)
    dev = torch.device("cpu")
    fps = max(1, min(args.frames_per_step, 8))
    def streams_of(r):                                  # every rank can rebuild what a peer sends
        g = torch.Generator().manual_seed(1234 + r)
        return [torch.randint(-32768, 32767, (64 + 8 * ((r + f) % 5),), dtype=torch.int16, generator=g) for f in range(fps)]
    streams = streams_of(rank)
    expect = [streams_of(r) for r in range(world)]
    ok = True

    def step():
        nonlocal ok
        if world == 1:
            return
        res = pdist.gather_step(streams, rank, world, dev, rotate=args.gather == "rotate")
        if res is not None:
            for r in range(world):
                for f, v in enumerate(res[r]):
                    if v is not None:
                        ok = ok and bool(torch.equal(v, expect[r][f]))
    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        okt = torch.tensor([1 if ok else 0], dtype=torch.int32)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        ok = bool(okt.item())
    if rank == 0:
        print(json.dumps({"metric": "dry run of the N-rank protocol (no GPU, no codec)", "value": None, "unit": "Mpixels/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(dt / max(args.steps, 1) * 1e3, 4), "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dry": True, "backend": args.backend,
                          "exchange": {"payloads_ok": ok}}), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # a step = --frames-per-step frames; the default timed region is > 1 s (22 x 360 8K frames at ~0.16 ms)
    ap.add_argument("--steps", type=int, default=None,
                    help="timed steps (default 22 at 8K, 80 at 4K: about 1.2 s of coding)")
    ap.add_argument("--warmup", type=int, default=2)
    # 360 frames (whole calls at 1, 2, 3, 4 and 6 frames per call): the driver's `--steps 20` then times more than a second
    # (7200 8K frames at ~0.16 ms; 288, then 324 frames did while a frame took more than 0.174 / 0.155 ms)
    ap.add_argument("--frames-per-step", type=int, default=360)
    ap.add_argument("--batch", type=int, default=0,
                    help="frames per picsong_encode_frames call (0 = the workload's default)")
    ap.add_argument("--pool", type=int, default=16, help="distinct device-resident input frames the steps rotate over")
    ap.add_argument("--workload", default="8k_lossless", choices=sorted(WORKLOADS))
    ap.add_argument("--streams", type=int, default=0,
                    help="HIP streams (each with its own context) consecutive calls alternate on (0 = the workload's default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gather", choices=["rotate", "root0"], default="rotate",
                    help="N > 1: where a step's codestreams go -- 'rotate': frame f of every rank's step to rank f mod N "
                         "(the writer role rotates: every xGMI link carries 1/N of a step), 'root0': everything to rank 0 "
                         "(99 GB/s per peer at 8K against the ~77 GB/s of one link direction)")
    ap.add_argument("--force-exchange", action="store_true",
                    help="run the N > 1 exchange path (per-step bucketed gather) at N = 1 too: exercises the code on a one-GPU box")
    ap.add_argument("--no-b3", action="store_true",
                    help="skip the three-frames-per-call measurement of the transform (traced runs: keeps every kernel's "
                         "average a single-frame launch's)")
    ap.add_argument("--cpu-sample-rows", type=int, default=0,
                    help="rows of the frame the CPU baseline encodes (0 = whole frame)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="torch.distributed backend of the ranks (gloo: only with --dry)")
    ap.add_argument("--dry", action="store_true",
                    help="the N-rank protocol only (launcher, rendezvous, per-step exchange, timing) on CPU tensors: no GPU, "
                         "no codec, no throughput -- what the CPU tests of the launcher run")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 80 if args.workload.startswith("4k") else 22
    if args.backend == "gloo" and not args.dry:
        raise SystemExit("bench.py: --backend gloo only with --dry (the codec has no CPU path)")

    in_dist = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not in_dist:
        sys.exit(launch_ranks(args))                    # (nothing above has touched the GPU)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if args.dry:
        sys.exit(dry_main(args, rank, world))

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the picsong HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    exch = world > 1 or args.force_exchange          # codestreams gathered at rank 0 (the frame-sharded path's exchange)
    if exch:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        # RCCL's send / receive kernels share the GPU with the coder, which keeps every SIMD's wave slots
        # full: put them on a high-priority stream so that they are dispatched as slots free up
        os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")
        # RCCL prints its version banner on STDOUT when the communicator comes up: this program's stdout is ONE JSON
        # line, so file descriptor 1 points at stderr until the first collective has run
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

    import picsong_amd as pa
    import picsong_dist as pdist
    import oracle_lib as orc          # checker + cpu_baseline only

    W, H, wl, lossy, qs = WORKLOADS[args.workload]
    lut_dir = os.path.join(orc.LUT_DIR, "n1_lossy" if lossy else "n1_lossless")
    # defaults by workload (measured at the end of round 3, three streams): an 8K frame is 4080 coder waves, a 4K frame
    # 1020 -- one per SIMD -- so 4K frames go six to a call (4 / 6 / 8 / 12: 187.8 / 194.5 / 194.1 / 192.9 Gpixel/s); 8K
    # frames three (1 / 2 / 3 / 6: 192.7 / 195.0 / 195.3 / 194.8: the transform's level launches serve three frames),
    # 8K 9/7 frames six (1 / 2 / 3 / 4 / 6: 197.1 / 208.8 / 215.9 / 219.1 / 221.5: its transform is what gains)
    batch = args.batch if args.batch > 0 else (6 if (W * H <= 3840 * 2160 or lossy) else 3)
    nstreams = args.streams if args.streams > 0 else 3
    fps = max(batch, (args.frames_per_step // batch) * batch)          # frames per step: whole calls
    pool_n = max(batch, (max(args.pool, 1) + batch - 1) // batch * batch)
    # several calls in flight: the contexts are told so (picsong_ctx_set_pipelined, a hint)
    codecs = [pa.Codec(W, H, wl=wl, lossy=lossy, qs=qs, lut_folder=lut_dir, device=local_rank,
                       pipelined=nstreams > 1) for _ in range(nstreams)]
    streams = [torch.cuda.Stream(device=local_rank) for _ in range(nstreams)]
    codec = codecs[0]
    AW, AH, nCB, P = codec.aw, codec.ah, codec.ncb, codec.P

    # synthetic frames (SURVEY 8d generator), pool_n distinct frames per rank, resident in HBM
    orc.set_threads(orc.usable_threads())
    pool_np = np.stack([orc.pad_frame(orc.gen_frame(W, H, rank * pool_n + i)).reshape(-1) for i in range(pool_n)])
    orc.set_threads(1)
    pool = torch.from_numpy(pool_np).cuda()                               # [pool_n, P] u8
    frame_np = pool_np[0].reshape(AH, AW)
    frame = pool[0]
    outs = [torch.empty((batch, codec.max_stream_shorts()), dtype=torch.int16, device="cuda") for _ in range(nstreams)]
    out = outs[0][0]
    dev = torch.device("cuda", local_rank)
    calls_per_step = fps // batch
    # ---- N > 1: the only exchange of the frame-sharded path, bucketed per STEP (picsong_dist.gather_step, covered by
    # the gloo tests): a step's codestreams land in per-step slots, their lengths stay on the device
    # (picsong_copy_last_totals), and ONE all-gather of the lengths + ONE grouped batch of sends / receives (to the
    # frames' writers: rank f mod N, or rank 0 with --gather root0) move the whole step over RCCL -- run one step late
    # (DeferredExchange), on a stream of its own, while the next step is being coded.  (Per call -- three host waits and two collectives every 0.2 ms -- the exchange was bound by
    # launch latency, not by xGMI.)  Every step's exchange is inside the timed region (flush).
    dx = pdist.DeferredExchange() if exch else None
    if exch:
        mss = (codec.max_stream_shorts() + 7) & ~7      # (slot rows and received streams on 16-byte boundaries)
        slots = [torch.empty((fps, mss), dtype=torch.int16, device="cuda") for _ in range(2)]
        totals_dev = [torch.zeros(fps, dtype=torch.int32, device="cuda") for _ in range(2)]
        rotate = args.gather == "rotate"
        if rotate:      # every rank receives ceil(fps / world) frames from every peer
            gather_bufs = [torch.empty((fps + world - 1) // world * mss, dtype=torch.int16, device="cuda") for _ in range(world - 1)]
        else:
            gather_bufs = [torch.empty(fps * mss, dtype=torch.int16, device="cuda") for _ in range(world - 1)] if rank == 0 else None
        last_res = [None]
        xstream = torch.cuda.Stream(device=local_rank)
        step_done = [[torch.cuda.Event() for _ in range(nstreams)] for _ in range(2)]
        slots_free = [None, None]                       # recorded on xstream when a parity's slots have been sent
        nstep = [0]
        last_lens = [None]
    ncall = [0]
    last_call = {}

    def step(first):
        # consecutive calls alternate over the streams: call i's coder tail overlaps call i+1's DWT / coder
        # head (each stream has its own context = its own workspace); `first`: this step holds frame 0 of the
        # video (the populated header)
        par = 0
        if exch:
            par = nstep[0] & 1
            nstep[0] += 1
            if slots_free[par] is not None:             # the step before last has left these slots
                for st in streams:
                    st.wait_event(slots_free[par])
        for j in range(calls_per_step):
            i = ncall[0]
            ncall[0] += 1
            k = i % nstreams
            f0 = (i * batch) % pool_n
            dst = slots[par][j * batch:(j + 1) * batch] if exch else outs[k]
            with torch.cuda.stream(streams[k]):
                if batch == 1:
                    codecs[k].encode_frame_async(pool[f0], dst[0], 0 if (first and j == 0) else 1)
                else:
                    codecs[k].encode_frames_async(pool[f0:f0 + batch], dst, 0 if (first and j == 0) else 1)
                if exch:
                    codecs[k].copy_last_totals(batch, totals_dev[par][j * batch:(j + 1) * batch])
            last_call[k] = (f0, dst)
        if exch:
            for k in range(nstreams):
                step_done[par][k].record(streams[k])

            def exchange(par=par):
                for e in step_done[par]:
                    e.synchronize()                     # this step is done; the next one keeps the GPU busy
                lens = totals_dev[par].tolist()
                last_lens[0] = lens
                with torch.cuda.stream(xstream):
                    # every frame's stream goes out from its slot (no packing copy); the slots are free again
                    # when the sends have been handed to RCCL's stream and completed (q.wait() orders xstream)
                    res = pdist.gather_step([slots[par][f, :lens[f]] for f in range(fps)], rank, world, dev,
                                            recv_bufs=gather_bufs, rotate=rotate)
                    last_res[0] = (par, lens, res)
                    ev = torch.cuda.Event()
                    ev.record(xstream)
                    slots_free[par] = ev
                    return res
            dx.submit(exchange)

    def sync_all():
        if exch:
            dx.flush()
            xstream.synchronize()
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i == 0)
    sync_all()
    prof_cap = min(256, (args.steps * calls_per_step + nstreams - 1) // nstreams)      # calls timed per stream
    for c in codecs:
        c.profile_begin(prof_cap)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(False)
    sync_all()
    dt = time.perf_counter() - t0
    stage_ms = np.concatenate([c.profile_read(prof_cap) for c in codecs], axis=0) / batch   # per frame
    for c in codecs:
        c.profile_begin(0)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- the timed loop's own last outputs (every stream's last call) against a fresh single-frame encode
    loop_ok = True
    for k, (f0, dst) in last_call.items():
        with torch.cuda.stream(streams[k]):
            totals = [codecs[k].last_total()] if batch == 1 else codecs[k].last_totals(batch)
            for b, tl in enumerate(totals):
                ref1 = codec.encode_frame(pool[f0 + b], 1)
                loop_ok = loop_ok and tl == ref1.numel() and bool(torch.equal(dst[b, :tl], ref1))
    if exch and last_lens[0] is not None:               # the lengths the last exchange moved are the streams' own
        loop_ok = loop_ok and all(9 + 2 * nCB + 1 < ln <= codec.max_stream_shorts() for ln in last_lens[0])
    torch.cuda.synchronize()
    # ---- the last exchange's payloads, end to end: every sender's per-frame checksums (all-gathered, outside the
    # timed region) against the checksums of what the receivers hold
    exchange_ok = None
    if exch and last_res[0] is not None:
        par, lens, res = last_res[0]

        def cksum(v):
            v = v.to(torch.int64)
            return (v * (torch.arange(v.numel(), device=v.device, dtype=torch.int64) % 251 + 1)).sum()
        mine = torch.stack([cksum(slots[par][f, :lens[f]]) for f in range(fps)])
        allck = torch.zeros(world * fps, dtype=torch.int64, device="cuda")
        dist.all_gather_into_tensor(allck, mine)
        allck = allck.view(world, fps)
        ok = True
        if res is not None:
            for r in range(world):
                for f in range(fps):
                    v = res[r][f]
                    if v is not None:
                        ok = ok and int(cksum(v).item()) == int(allck[r, f].item())
        okt = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cuda")
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        exchange_ok = bool(okt.item())
        loop_ok = loop_ok and exchange_ok

    total_shorts = codec.last_total()
    flag = codec.range_flag()

    # ---- the same frames on ONE stream, nothing else on the GPU: per-stage kernel time in isolation
    # (a context of its own, not told that anything shares the GPU: what a single-stream caller gets)
    iso_n = 12
    iso = codec if nstreams == 1 else pa.Codec(W, H, wl=wl, lossy=lossy, qs=qs, lut_folder=lut_dir,
                                                device=local_rank)

    def iso_call(i):
        f0 = (i * batch) % pool_n
        if batch == 1:
            iso.encode_frame_async(pool[f0], outs[0][0], 1)
        else:
            iso.encode_frames_async(pool[f0:f0 + batch], outs[0], 1)
    for i in range(2):
        iso_call(i)
    torch.cuda.synchronize()
    iso.profile_begin(iso_n)
    for i in range(iso_n):
        iso_call(2 + i)
    torch.cuda.synchronize()
    iso_ms = iso.profile_read(iso_n).mean(axis=0) / batch              # per frame
    iso.profile_begin(0)
    # ---- a LONE frame (an image encoder cannot batch): single-frame calls on one stream, nothing else on the GPU
    lone_ms = iso_ms
    if batch != 1:
        for i in range(2 + iso_n):
            if i == 2:
                torch.cuda.synchronize()
                iso.profile_begin(iso_n)
            iso.encode_frame_async(pool[i % pool_n], outs[0][0], 1)
        torch.cuda.synchronize()
        lone_ms = iso.profile_read(iso_n).mean(axis=0)
        iso.profile_begin(0)
    # ---- and as a video engine hands them over: three frames per call (picsong_encode_frames: every launch of
    # the transform's levels serves three frames), still one stream with nothing else on the GPU
    b3_ms = None
    if batch == 1 and pool_n >= 6 and not args.no_b3:
        out3 = torch.empty((3, codec.max_stream_shorts()), dtype=torch.int16, device="cuda")
        for i in range(2 + iso_n):
            if i == 2:
                torch.cuda.synchronize()
                iso.profile_begin(iso_n)
            f0 = (3 * i) % (pool_n - 2)
            iso.encode_frames_async(pool[f0:f0 + 3], out3, 1)
        torch.cuda.synchronize()
        b3_ms = iso.profile_read(iso_n).mean(axis=0) / 3.0              # per frame
        iso.profile_begin(0)
        del out3

    # ---- measured device-copy roof ("use the measured device copy bandwidth as the roof
    # and state both"): plain torch copy / fill over one coefficient plane, outside the timed region
    def _rate(fn, nbytes, iters=10):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return nbytes / (e0.elapsed_time(e1) / iters * 1e-3) / 1e9
    _a = torch.zeros(P, dtype=torch.int32, device="cuda")
    _b = torch.empty_like(_a)
    copy_gbs = _rate(lambda: _b.copy_(_a), 8 * P)
    fill_gbs = _rate(lambda: _b.fill_(1), 4 * P)
    del _a, _b

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
        if exch:
            dist.destroy_process_group()
        return

    ms_per_step = dt / args.steps * 1e3
    ms_per_frame = ms_per_step / fps
    mpix = (W * H * world * args.steps * fps) / dt / 1e6

    # ---- roofline of the dominant kernel (bpc_encode_kernel): algorithmic bytes per launch
    # (nCB*16384 coefficient bytes + 4*nCB sizes + 2*sum(ncw) codeword bytes per frame, x the
    # `batch` frames of a launch) over the kernel's mean duration measured with HIP events on the launch
    # stream inside the timed region.
    sizes = stream0.cpu().numpy().view(np.uint16)[10:10 + 2 * nCB:2].astype(np.int64)
    ncw = int((sizes - 1).sum())
    bpc_bytes = (nCB * 16384 + 4 * nCB + 2 * ncw) * batch
    dwt_ms, bpc_ms, pack_ms = [float(x) for x in stage_ms.mean(axis=0)]        # per frame
    bpc_launch_ms, dwt_launch_ms = bpc_ms * batch, dwt_ms * batch
    bpc_gbs = bpc_bytes / (bpc_launch_ms * 1e-3) / 1e9
    dwt_b = dwt_bytes(P, wl, 1) * batch
    bpc_traffic, dwt_traffic = pmc_traffic(args.workload, batch)
    roofline = {"kernel": "bpc_encode_kernel (BPC-PaCo encode)", "bound": "hbm",
                "achieved": round(bpc_gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(bpc_gbs / HBM_PEAK_GBS, 5), "traffic": bpc_traffic,
                "algorithmic_bytes_per_launch": bpc_bytes, "frames_per_launch": batch,
                "avg_launch_ms": round(bpc_launch_ms, 4),
                "codeblocks_per_s": round(nCB * batch / (bpc_launch_ms * 1e-3), 1),
                "single_stream": {"avg_launch_ms": round(float(iso_ms[1]) * batch, 4),
                                  "codeblocks_per_s": round(nCB / (float(iso_ms[1]) * 1e-3), 1)},
                "valu_issue": valu_issue(pmc_valu(args.workload), ms_per_frame * 1e-3, float(iso_ms[1]) * 1e-3),
                "valu_busy": pmc_valu_busy(args.workload, pmc_valu_all(args.workload), ms_per_frame * 1e-3),
                "source": library_hashes(),
                "note": "BPC is bound by vector-instruction issue, not HBM (SURVEY 8d): codeblocks/s and "
                        "valu_issue are the figures of merit, the HBM fraction is reported for completeness. "
                        "`traffic` is the PMC figure of the committed counter passes (profiles/): the coefficients "
                        "are read once, the transposed bit-planes go through a 16 KB-per-wave scratch.  "
                        "frames_per_step x the isolated launch time exceeds ms_per_step: legal because the coder "
                        "kernels of the calls in flight on the %d streams CO-RESIDE (a frame's launch fills 4 of a "
                        "SIMD's 7 wave slots; the next call's waves take slots as they free up), so `avg_launch_ms` "
                        "(HIP events around a launch that shares the GPU) is longer than a frame's share of the step" % nstreams}
    roofline_dwt = {"kernel": "dwt_fwd_kernel / dwt_fwd2_kernel (all levels, u8 ingest fused)", "bound": "hbm",
                    "achieved": round(dwt_b / (dwt_launch_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(dwt_b / (dwt_launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                    "traffic": dwt_traffic, "algorithmic_bytes_per_launch": int(dwt_b), "frames_per_launch": batch,
                    "avg_launch_ms": round(dwt_launch_ms, 4),
                    "single_stream": {"avg_launch_ms": round(float(iso_ms[0]) * batch, 4),
                                      "achieved": round(dwt_b / (float(iso_ms[0]) * batch * 1e-3) / 1e9, 2),
                                      "frac": round(dwt_b / (float(iso_ms[0]) * batch * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)},
                    # a lone frame per call (lone_frame's transform stage): what the launches cost with nothing to share
                    "lone_frame": {"ms": round(float(lone_ms[0]), 4),
                                   "achieved": round(dwt_b / batch / (float(lone_ms[0]) * 1e-3) / 1e9, 2),
                                   "frac": round(dwt_b / batch / (float(lone_ms[0]) * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)},
                    # (batch = 3: `single_stream` IS the three-frames-per-call shape; batch = 1: measured on the side)
                    "three_frames_per_call": ({"ms_per_frame": round(float(iso_ms[0]), 4),
                                               "achieved": round(dwt_b / batch / (float(iso_ms[0]) * 1e-3) / 1e9, 2),
                                               "frac": round(dwt_b / batch / (float(iso_ms[0]) * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}
                                              if batch == 3 else None) if b3_ms is None else {
                        "ms_per_frame": round(float(b3_ms[0]), 4),
                        "achieved": round(dwt_b / batch / (float(b3_ms[0]) * 1e-3) / 1e9, 2),
                        "frac": round(dwt_b / batch / (float(b3_ms[0]) * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)},
                    "measured_roof": {"copy_i32_GBps": round(copy_gbs, 1), "fill_i32_GBps": round(fill_gbs, 1),
                                      "frac_of_copy_single_stream":
                                          round(dwt_b / (float(iso_ms[0]) * batch * 1e-3) / 1e9 / copy_gbs, 5)},
                    "note": "all of a frame's level launches counted as one; `achieved` uses HIP-event times "
                            "inside the timed region, where the calls of the other stream(s) share the GPU; "
                            "`single_stream` is the same call shape on one stream with nothing else running, "
                            "`lone_frame` one frame per call, "
                            "`three_frames_per_call` picsong_encode_frames over three frames on one stream (the level "
                            "launches serve three frames each); the "
                            "input frames rotate over a pool larger than the Infinity Cache, so every frame's "
                            "pixels come from HBM; `measured_roof` is a plain device copy / fill of a 134 MB plane"}

    # ---- CPU baseline: the oracle (C port, OpenMP over codeblocks / DWT rows+columns) on the
    # box's host cores, rank 0, N = 1 only.  Same stage boundaries as the GPU step (level shift +
    # DWT, BPC, pack) on preallocated, warmed buffers; file I/O and padding excluded.
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        import ctypes as C
        L = orc.lib()
        lut = orc.lut_for(lossy, wl)
        rows = AH if args.cpu_sample_rows <= 0 else min(AH, max(64 << (wl - 1), (args.cpu_sample_rows // 64) * 64))
        pad = np.ascontiguousarray(frame_np[:rows]) if rank == 0 else None
        Pc = AW * rows
        ncb_c = (AW // 64) * (rows // 64)
        extra_c = orc.dwt_extra(AW, rows, wl)
        ftype = np.float32 if lossy else np.int32
        shifted = np.zeros(Pc, ftype)
        coef = np.zeros(Pc + extra_c, ftype)
        staging = np.zeros(Pc, np.int32)
        sizes_c = np.zeros(ncb_c, np.int32)
        out_c = np.zeros(9 + 2 * ncb_c + Pc + 1, np.uint16)
        vp = lambda a: a.ctypes.data_as(C.c_void_p)

        def cpu_encode():
            if lossy:
                L.po_level_shift_fwd_f32(vp(pad), vp(shifted), Pc, 8)
                L.po_dwt97_forward(vp(shifted), vp(coef), AW, rows, wl, C.c_float(qs))
            else:
                L.po_level_shift_fwd_i32(vp(pad), vp(shifted), Pc, 8)
                L.po_dwt53_forward(vp(shifted), vp(coef), AW, rows, wl)
            L.po_bpc_encode(vp(coef), int(lossy), AW, rows, wl, C.byref(lut.c), vp(staging), vp(sizes_c))
            return L.po_bitstream_pack(vp(staging), vp(sizes_c), ncb_c, None, vp(out_c))

        nthr = min(orc.max_threads(), host_cores())
        orc.set_threads(nthr)
        cpu_encode()                                   # warm-up (page faults, thread pool)
        reps, t1 = 0, time.perf_counter()
        while True:
            total_c = cpu_encode()
            reps += 1
            if time.perf_counter() - t1 > 8.0 or reps >= 64:
                break
        mt = (time.perf_counter() - t1) / reps
        orc.set_threads(1)
        t1 = time.perf_counter()
        cpu_encode()
        st_ = time.perf_counter() - t1
        cpu = {"value": round(W * min(H, rows) / mt / 1e6, 2), "unit": "Mpixels/s", "cores": nthr, "kind": "port",
               "sample": f"{reps} x 1 frame {W}x{min(H, rows)} of the workload, level shift + DWT + BPC + pack on "
                         f"warmed buffers, oracle/picsong_oracle.c -O3 -fopenmp, {nthr} threads, {mt * reps:.1f} s",
               "single_thread_value": round(W * min(H, rows) / st_ / 1e6, 3), "host_cpu_count": os.cpu_count(),
               "host_cpu_quota": host_cores()}
        if rows == AH:
            ref_stream = out_c[:total_c]
            cpu["codestream_matches_gpu"] = bool(np.array_equal(
                ref_stream[9:], stream0.cpu().numpy().view(np.uint16)[9:]))

    line = {
        "metric": "Mpixels/s encode (DWT+BPC) 8K P5 lossless; round-trip bit-exact"
                  if args.workload == "8k_lossless" else f"Mpixels/s encode (DWT+BPC) {args.workload}",
        "value": round(mpix, 2), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32" if lossy else "int32", "data": "synthetic",
        "config": {"workload": f"{W}x{H} greyscale u8 frames (padded {AW}x{AH}), "
                               f"-type {int(lossy)} {'9/7 qs=%g' % qs if lossy else '5/3 lossless'}, "
                               f"wl={wl}, cp=2, k=0, LUT {'n1_lossy' if lossy else 'n1_lossless'}; "
                               f"a step = {fps} frames per GPU from a pool of {pool_n} distinct HBM-resident frames "
                               f"({pool_n * P / 1e6:.0f} MB), {batch} frame(s) per call, calls alternating over "
                               f"{nstreams} HIP stream(s); frames sharded over {world} GPU(s)",
                   "frames_per_step": fps, "frames_per_call": batch, "streams": nstreams, "pool_frames": pool_n,
                   "codeblocks": nCB, "stream_shorts": int(total_shorts),
                   "bits_per_pixel": round(total_shorts * 16 / (W * H), 4)},
        "ms_per_frame": round(ms_per_frame, 5), "timed_seconds": round(dt, 3),
        "roundtrip_ok": roundtrip_ok, "timed_loop_outputs_ok": loop_ok, "range_flag": flag,
        "exchange": None if not exch else {
            "form": "frame f of a rank's step to rank f mod N, one grouped RCCL send / receive batch per step, one step late"
                    if rotate else "every codestream to rank 0, one grouped RCCL send / receive batch per step, one step late",
            "payloads_ok": exchange_ok},
        "stage_ms": {"dwt": round(dwt_ms, 4), "bpc": round(bpc_ms, 4), "pack": round(pack_ms, 4),
                     "note": "per frame, HIP events on the launch streams inside the timed region"},
        "stage_ms_single_stream": {"dwt": round(float(iso_ms[0]), 4), "bpc": round(float(iso_ms[1]), 4),
                                   "pack": round(float(iso_ms[2]), 4)},
        "lone_frame": {"ms": round(float(sum(lone_ms)), 4), "mpixels_per_s": round(W * H / (float(sum(lone_ms)) * 1e-3) / 1e6, 1),
                       "stage_ms": {"dwt": round(float(lone_ms[0]), 4), "bpc": round(float(lone_ms[1]), 4), "pack": round(float(lone_ms[2]), 4)},
                       "note": "one frame per call on one stream with nothing else on the GPU (HIP events around the stages): "
                               "what a single image costs"},
        "roofline": roofline, "roofline_dwt": roofline_dwt, "cpu_baseline": cpu,
    }
    if psnr is not None:
        line["psnr_db"] = round(psnr, 3)
    print(json.dumps(line))
    if exch:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
