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
    # configs[4]: ONE 16384 x 16384 frame, -type 0 wl 5; N = 1: picsong_encode_frame, N > 1: rows of the transform and
    # codeblock stripes sharded over the ranks (picsong_dist.encode_frame_banded), "strong" scaling (intra_main)
    "16k_intra": (16384, 16384, 5, False, 1.0),
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
    for tag in ("r04", "r03", "r02"):
        if os.path.exists(os.path.join(ROOT, "profiles", tag + "_pmc_sq.csv")):
            return tag
    return "r04"


PROFILE_TAG = _profile_tag()
CSRC = os.path.join(ROOT, "cuda-image-and-video-codec_amd", "csrc")


def source_hash():
    """sha256 (16 hex digits) over the kernel sources the library is built from, in a fixed order.  The counter
    summaries under profiles/ carry this hash next to the binary's: a rebuild of the same sources changes the binary's
    (embedded build ids) but not this one."""
    import hashlib
    h = hashlib.sha256()
    try:
        for name in ("picsong_hip.hip", "bpc_kernels.hpp", "dwt_kernels.hpp", "pack_kernels.hpp", "launch_plan.hpp", "Makefile"):
            h.update(open(os.path.join(CSRC, name), "rb").read())
    except OSError:
        return None
    return h.hexdigest()[:16]


def library_hashes():
    """The library this run loads and the one the committed counter passes were taken from
    (profiles/<tag>_library.sha256: line 1 the binary's sha256, line 2 the sources' -- source_hash()).  The OFFLINE
    figures of the line (`traffic`, `valu_issue`, `valu_busy`) describe the latter: they are reported only when the
    SOURCES match (`profiles_match_library`), None otherwise."""
    import hashlib
    so = os.environ.get("PICSONG_SO") or os.path.join(CSRC, "libpicsong_hip.so")
    try:
        mine = hashlib.sha256(open(so, "rb").read()).hexdigest()[:16]
    except OSError:
        mine = None
    prof = prof_src = None
    try:
        lines = open(os.path.join(ROOT, "profiles", PROFILE_TAG + "_library.sha256")).read().split()
        prof = lines[0][:16]
        prof_src = lines[1][:16] if len(lines) > 1 else None
    except (OSError, IndexError):
        pass
    src = None if os.environ.get("PICSONG_SO") else source_hash()      # (a variant library is not the tree's sources)
    if prof_src and src:
        match = prof_src == src
    else:
        match = (mine == prof) if (mine and prof) else None
    return {"library_sha256_16": mine, "source_sha256_16": src, "profiles_tag": PROFILE_TAG,
            "profiles_library_sha256_16": prof, "profiles_source_sha256_16": prof_src, "profiles_match_library": match}


def _pmc_file(kind, workload):
    """profiles/<tag>_pmc_<kind>[_<workload>].csv: the 8K lossless passes carry no suffix (the headline workload)."""
    suffix = "" if workload == "8k_lossless" else "_" + workload
    return os.path.join(ROOT, "profiles", "%s_pmc_%s%s.csv" % (PROFILE_TAG, kind, suffix))


def _offline_ok():
    """Offline counters describe the committed profiles' build: used only when that is this run's build."""
    return library_hashes()["profiles_match_library"] is True or os.environ.get("PICSONG_BENCH_STALE_PMC") == "1"



def pmc_traffic(workload, batch=1):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/*_pmc_hbm.csv: FETCH_SIZE and
    WRITE_SIZE collected in separate runs of this command with --streams 1 --batch 1, unit 1024 B;
    FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 note -- calibrated on the DWT kernels' known read
    bytes), scaled to the `batch` frames of a launch.  OFFLINE figures: they describe the build the
    profiles were taken from.  None when no summary for this workload is committed."""
    import csv
    path = _pmc_file("hbm", workload)
    if not os.path.exists(path) or not _offline_ok():
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
    path = _pmc_file("sq", workload)
    if not os.path.exists(path) or not _offline_ok():
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
    path = _pmc_file("sq_pipelined", workload)
    if not os.path.exists(path) or not _offline_ok():
        return None
    c = {}
    for r in csv.DictReader(ln for ln in open(path) if not ln.startswith("#")):
        if "bpc_encode_kernel" in r["Kernel_Name"]:
            c[r["Counter_Name"]] = float(r["MeanValue"])
    if not c.get("SQ_ACTIVE_INST_VALU") or not c.get("GRBM_GUI_ACTIVE"):
        return None
    xcc = 8                                               # XCDs of an MI355X (the summary adds their GRBM values)
    cus = _device_cus()
    simds = 4 * cus
    gui = c["GRBM_GUI_ACTIVE"] / xcc                      # (the summary adds the eight XCCs' values of a dispatch)
    out = {"source": os.path.relpath(path, ROOT),
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


def _device_cus():
    try:
        import torch
        return int(torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count)
    except Exception:
        return 256


def pmc_valu_all(workload):
    """VALU wave-instructions per frame of ALL the encode path's kernels (coder, transform levels, scan, pack) from the
    committed SQ pass (profiles/*_pmc_sq.csv: mean per dispatch x dispatches per frame)."""
    import csv
    path = _pmc_file("sq", workload)
    if not os.path.exists(path) or not _offline_ok():
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
    simd_hz = _device_cus() * 4 * 2.4e9
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


def dwt_required_bytes(P, wl, c16=True, fused01=True):
    """Bytes the forward transform AS BUILT has to move per frame: u8 pixels in, the coded subbands
    (HL / LH / HH of every level, LL of the last) out as int16 when `c16` (32-bit otherwise), the LL a next level reads
    as 32-bit words out and in again -- except LL1 when levels 0 and 1 are one launch (`fused01`: it never leaves the
    registers).  No halo or run-in re-reads: those are the kernel's overhead, `traffic` shows them."""
    cb = 2 if c16 else 4
    total = float(P)                                       # level 0 ingests u8
    for l in range(wl):
        n = P / 4.0 ** l                                   # samples of this level's domain
        last = l == wl - 1
        if l > 0 and not (l == 1 and fused01):
            total += 4.0 * n                               # read LL_l
        total += cb * 0.75 * n                             # HL, LH, HH out
        if last:
            total += cb * 0.25 * n                         # the last LL is a coded subband
        elif not (l == 0 and fused01):
            total += 4.0 * 0.25 * n                        # LL_{l+1} out for the next launch
    return total


def count_gpus_without_hip():
    """GPUs of this node from the KFD topology in sysfs (a node with simd_count > 0 is a GPU, the others are CPUs):
    no HIP call, so a process that goes on to start its ranks as children has not touched the GPU.  0 where there is
    no KFD at all, None when the topology is there and cannot be read (the ranks then fail on their own device)."""
    base = "/sys/class/kfd/kfd/topology/nodes"
    if not os.path.isdir("/sys/class/kfd"):
        return 0                                           # no amdgpu compute driver on this machine: no GPU
    try:
        n = 0
        for node in os.listdir(base):
            for ln in open(os.path.join(base, node, "properties")):
                k, _, v = ln.partition(" ")
                if k == "simd_count" and int(v) > 0:
                    n += 1
        return n
    except (OSError, ValueError):
        return None


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
        have = count_gpus_without_hip()                 # (sysfs: nothing here may open the HIP runtime before the fork)
        if have is not None and have < args.gpus:
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


def rccl_witness(torch, dist, local_rank, world):
    """What RCCL itself saw, so that "did N ranks on N distinct GPUs take part" can be read off the line: `ranks_seen` =
    an all-reduce (sum) of 1 over the communicator, `devices` = every rank's (hostname, local rank, PCI bus id, device
    name) all-gathered through it.  Outside the timed region."""
    import socket
    one = torch.ones(1, dtype=torch.int32, device="cuda")
    dist.all_reduce(one, op=dist.ReduceOp.SUM)
    pr = torch.cuda.get_device_properties(local_rank)
    bus = "%04x:%02x:%02x" % (int(getattr(pr, "pci_domain_id", 0)), int(getattr(pr, "pci_bus_id", -1)) & 0xFF,
                              int(getattr(pr, "pci_device_id", 0)) & 0xFF) if hasattr(pr, "pci_bus_id") else "?"
    me = ("%s|%d|%s|%s" % (socket.gethostname(), local_rank, bus, pr.name)).encode()[:127]
    buf = torch.zeros(128, dtype=torch.uint8, device="cuda")
    buf[:len(me)] = torch.tensor(list(me), dtype=torch.uint8, device="cuda")
    allb = torch.zeros(128 * world, dtype=torch.uint8, device="cuda")
    dist.all_gather_into_tensor(allb, buf)
    devs = []
    for r in range(world):
        raw = bytes(allb[128 * r:128 * (r + 1)].tolist()).rstrip(b"\0").decode(errors="replace").split("|")
        devs.append({"rank": r, "host": raw[0], "local_rank": int(raw[1]) if len(raw) > 1 and raw[1].isdigit() else None,
                     "pci_bus_id": raw[2] if len(raw) > 2 else None, "name": raw[3] if len(raw) > 3 else None})
    distinct = len({(d["host"], d["pci_bus_id"]) for d in devs})
    return {"ranks_seen": int(one.item()), "devices": devs, "distinct_devices": distinct,
            "backend": dist.get_backend()}


def cpu_baseline_leg(orc, frame_np, AW, AH, W, H, wl, lossy, qs, sample_rows, gpu_stream_u16, budget_s=8.0):
    """The oracle (C port, OpenMP over codeblocks / DWT rows + columns) on the box's host cores.  Same stage boundaries
    as the GPU step (level shift + DWT, BPC, pack) on preallocated, warmed buffers; file I/O and padding excluded.
    `sample_rows` > 0: only that many rows of the frame (a bounded sample of the workload)."""
    import ctypes as C
    L = orc.lib()
    lut = orc.lut_for(lossy, wl)
    rows = AH if sample_rows <= 0 else min(AH, max(64 << (wl - 1), (sample_rows // 64) * 64))
    pad = np.ascontiguousarray(frame_np[:rows])
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
        if time.perf_counter() - t1 > budget_s or reps >= 64:
            break
    mt = (time.perf_counter() - t1) / reps
    st_ = None
    if W * H <= 7680 * 4320:                       # (a 16K frame on one thread is half a minute)
        orc.set_threads(1)
        t1 = time.perf_counter()
        cpu_encode()
        st_ = time.perf_counter() - t1
    orc.set_threads(1)
    cpu = {"value": round(W * min(H, rows) / mt / 1e6, 2), "unit": "Mpixels/s", "cores": nthr, "kind": "port",
           "sample": f"{reps} x 1 frame {W}x{min(H, rows)} of the workload, level shift + DWT + BPC + pack on "
                     f"warmed buffers, oracle/picsong_oracle.c -O3 -fopenmp, {nthr} threads, {mt * reps:.1f} s",
           "single_thread_value": round(W * min(H, rows) / st_ / 1e6, 3) if st_ else None,
           "host_cpu_count": os.cpu_count(), "host_cpu_quota": host_cores()}
    if rows == AH and gpu_stream_u16 is not None:
        cpu["codestream_matches_gpu"] = bool(np.array_equal(out_c[:total_c][9:], gpu_stream_u16[9:]))
    return cpu


def intra_dry_ops(torch, aw, ah, rank, world, plan):
    """Stand-in codec of the --dry form of the 16k_intra workload: the transform's band writes a rank-specific pattern
    into its LL1 rows, a stripe's mini-stream is a deterministic function of its codeblock range AND of a checksum of the
    whole LL1 plane -- so the splice rank 0 ends up with is right only if the all-gather and both gathers moved what
    they should.  expected(): the same splice computed without any exchange."""
    n_ll1 = (aw // 2) * (ah // 2)

    def pattern(k):
        p = plan[k]
        return (torch.arange(p["ll1_count"], dtype=torch.int32) * (2 * k + 3) + 17 * k) % 251

    def mini(b, n, ck):
        pairs, payload = [], []
        for cb in range(b, b + n):
            ln = 1 + (cb * 7 + ck) % 5
            pairs += [cb % 11, ln]
            payload += [(cb * 31 + j + ck) % 32749 for j in range(ln - 1)]
        v = [-1] * 9 + pairs + payload + [-1]
        return torch.tensor(v, dtype=torch.int16)

    class Ops:
        def __init__(self):
            self.plane = torch.zeros(n_ll1, dtype=torch.int32)

        def dwt_band(self, row0, rows):
            p = plan[rank]
            self.plane[p["ll1_begin"]:p["ll1_begin"] + p["ll1_count"]] = pattern(rank)

        def ll1(self):
            return self.plane

        def dwt_tail(self):
            pass

        def encode_stripe(self, b, n):
            return mini(b, n, int(self.plane.sum().item()) % 1009)

    def expected(header9):
        import picsong_dist as pdist
        full = torch.cat([pattern(k) for k in range(world)])
        ck = int(full.sum().item()) % 1009
        minis = [mini(*p["stripes"][0], ck) for p in plan] + [mini(*p["stripes"][1], ck) for p in plan]
        counts = [p["stripes"][0][1] for p in plan] + [p["stripes"][1][1] for p in plan]
        return pdist.splice_stripes(header9, minis, counts)
    return Ops(), expected


def dry_intra(args, rank, world):
    """--dry --workload 16k_intra: the N-rank protocol of the intra-frame split (picsong_dist.encode_frame_banded: LL1
    all-gather + two gathers + splice, barrier-bracketed timing, MAX over ranks, one JSON line) on CPU tensors over gloo,
    with a stand-in codec on a small geometry."""
    import torch
    import torch.distributed as dist
    import picsong_dist as pdist
    if world > 1:
        dist.init_process_group(args.backend)
    aw, ah = 256, 128 * world * 2
    plan = pdist.band_plan(aw, ah, world)
    ops, expected = intra_dry_ops(torch, aw, ah, rank, world, plan)
    hdr = torch.arange(9, dtype=torch.int16)
    dev = torch.device("cpu")
    ok = True

    def step():
        nonlocal ok
        full = pdist.encode_frame_banded(aw, ah, ops, hdr, rank, world, dev)
        if rank == 0:
            ok = ok and bool(torch.equal(full, expected(hdr)))
        else:
            ok = ok and full is None
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
    ranks_seen = 1
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        okt = torch.tensor([1 if ok else 0, 1], dtype=torch.int32)
        dist.all_reduce(okt[:1], op=dist.ReduceOp.MIN)
        one = torch.ones(1, dtype=torch.int32)
        dist.all_reduce(one, op=dist.ReduceOp.SUM)
        ok, ranks_seen = bool(okt[0].item()), int(one.item())
    if rank == 0:
        print(json.dumps({"metric": "dry run of the intra-frame protocol (no GPU, no codec)", "value": None, "unit": "Mpixels/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(dt / max(args.steps, 1) * 1e3, 4), "higher_is_better": True,
                          "scaling": "strong", "vs_baseline": None, "dry": True, "backend": args.backend,
                          "config": {"workload": "16k_intra (stand-in codec, %dx%d)" % (aw, ah)},
                          "exchange": {"splice_ok": ok, "ranks_seen": ranks_seen}}), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0


def intra_main(args, rank, local_rank, world, witness):
    """BASELINE configs[4]: ONE 16384 x 16384 greyscale frame, -type 0 wl 5.  N = 1: picsong_encode_frame, one frame per
    step.  N > 1: the frame's rows sharded over the ranks (second form; picsong_dist.encode_frame_banded):
    level 0 of the transform on the rank's row band, ONE all-gather of the LL1 row bands over RCCL, levels >= 1
    redundantly, the rank's two codeblock stripes coded, two gathers to rank 0, which splices the 1-GPU codestream.
    value = the frame's pixels / max-over-ranks time per step: "strong" scaling.  Outside the timed region rank 0 encodes
    the whole frame alone and compares (crc32 and bytes) with the splice."""
    import zlib
    import torch
    import torch.distributed as dist
    import picsong_amd as pa
    import picsong_dist as pdist
    import oracle_lib as orc          # frame generator, checker + cpu_baseline only
    W, H, wl, lossy, qs = WORKLOADS["16k_intra"]
    steps = args.steps
    codec = pa.Codec(W, H, wl=wl, lossy=lossy, qs=qs, lut_folder=os.path.join(orc.LUT_DIR, "n1_lossless"), device=local_rank)
    AW, AH, nCB, P = codec.aw, codec.ah, codec.ncb, codec.P
    orc.set_threads(orc.usable_threads())
    frame_np = orc.pad_frame(orc.gen_frame(W, H, 0))
    orc.set_threads(1)
    frame = torch.from_numpy(frame_np).cuda().view(-1)
    dev = torch.device("cuda", local_rank)
    out = torch.empty(codec.max_stream_shorts(), dtype=torch.int16, device="cuda")
    hdr = torch.from_numpy(pa.header_pack(codec.params).view(np.int16).copy()).cuda()
    banded = world > 1 or args.force_exchange
    plan = pdist.band_plan(AW, AH, world)
    last = [None]
    if banded:
        assert plan is not None, "16k_intra: AH must be a multiple of 128 * N"
        coef = codec.new_coef_buffer()
        n_ll1 = (AW // 2) * (AH // 2)

        class Ops:
            def dwt_band(self, row0, rows):
                codec.dwt_forward_band(frame, row0, rows, coef)

            def ll1(self):
                return coef[P:P + n_ll1]

            def dwt_tail(self):
                codec.dwt_forward_tail(coef)

            def encode_stripe(self, b, n):
                return codec.encode_stripe_coded(coef, b, n)
        ops = Ops()

        def step():
            last[0] = pdist.encode_frame_banded(AW, AH, ops, hdr, rank, world, dev)
    else:
        def step():
            codec.encode_frame_async(frame, out, 0)

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    sync_all()
    if not banded:
        codec.profile_begin(min(steps, 256))
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync_all()
    dt = time.perf_counter() - t0
    stage = None
    if not banded:
        stage = codec.profile_read(min(steps, 256)).mean(axis=0)
        codec.profile_begin(0)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if args.phase != "all":                           # traced run: the timed shape alone
        if rank == 0:
            print(json.dumps({"phase": args.phase, "workload": "16k_intra", "n_gpus": world, "steps": steps,
                              "ms_per_step": round(dt / steps * 1e3, 4),
                              "stage_ms": None if stage is None else {"dwt": round(float(stage[0]), 4), "bpc": round(float(stage[1]), 4),
                                                                      "pack": round(float(stage[2]), 4)},
                              "source": library_hashes()}))
        if world > 1:
            dist.destroy_process_group()
        return
    # ---- outside the timed region: the whole frame on ONE GPU (rank 0), its round trip, and the splice against it
    ok_splice = crc_one = crc_splice = None
    roundtrip_ok = None
    single = None
    if rank == 0:
        single = codec.encode_frame(frame, 0).clone()
        crc_one = zlib.crc32(single.cpu().numpy().tobytes()) & 0xFFFFFFFF
        dec = codec.decode_frame(single)
        roundtrip_ok = bool(torch.equal(dec.view(-1), frame))
        del dec
        if banded:
            crc_splice = zlib.crc32(last[0].cpu().numpy().tobytes()) & 0xFFFFFFFF
            ok_splice = bool(crc_splice == crc_one and torch.equal(last[0], single))
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    flag = codec.range_flag()
    ms_per_step = dt / steps * 1e3
    mpix = W * H * steps / dt / 1e6
    su16 = single.cpu().numpy().view(np.uint16)
    sizes = su16[10:10 + 2 * nCB:2].astype(np.int64)
    ncw = int((sizes - 1).sum())
    bpc_bytes = nCB * 16384 + 4 * nCB + 2 * ncw
    roofline = roofline_dwt = None
    if stage is not None:
        dwt_ms, bpc_ms, pack_ms = [float(x) for x in stage]
        g = bpc_bytes / (bpc_ms * 1e-3) / 1e9
        roofline = {"kernel": "bpc_encode_kernel (BPC-PaCo encode)", "bound": "hbm", "achieved": round(g, 2),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(g / HBM_PEAK_GBS, 5), "traffic": None,
                    "algorithmic_bytes_per_launch": bpc_bytes, "frames_per_launch": 1, "avg_launch_ms": round(bpc_ms, 4),
                    "codeblocks_per_s": round(nCB / (bpc_ms * 1e-3), 1), "source": library_hashes(),
                    "note": "one launch over the frame's 65,536 codeblocks = 32,768 waves, 4.6 rounds of the GPU's 7 wave "
                            "slots per SIMD; bound by vector-instruction issue, not HBM (SURVEY 8d); HIP events on the launch stream"}
        req, alg = dwt_required_bytes(P, wl, True, True), dwt_bytes(P, wl, 1)
        roofline_dwt = {"kernel": "dwt_fwd2_kernel + dwt_fwd_kernel (levels >= 2)", "bound": "hbm",
                        "achieved": round(req / (dwt_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(req / (dwt_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                        "frac_of_required": round(req / (dwt_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                        "frac_survey_8d": round(alg / (dwt_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                        "required_bytes_per_launch": int(req), "algorithmic_bytes_per_launch": int(alg), "traffic": None,
                        "avg_launch_ms": round(dwt_ms, 4)}
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline_leg(orc, frame_np, AW, AH, W, H, wl, lossy, qs, args.cpu_sample_rows, su16, budget_s=6.0)
    line = {"metric": "Mpixels/s encode (DWT+BPC) 16K x 16K single frame, lossless (BASELINE configs[4])",
            "value": round(mpix, 2), "unit": "Mpixels/s", "n_gpus": world, "steps": steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "int32", "data": "synthetic",
            "config": {"workload": f"ONE {W}x{H} greyscale u8 frame per step, -type 0 5/3 lossless, wl={wl}, cp=2, k=0, LUT n1_lossless; "
                                   + (f"rows of the transform and codeblock stripes sharded over {world} GPU(s): level 0 on the rank's row band, "
                                      "LL1 all-gathered over RCCL, levels >= 1 redundantly, two codeblock stripes per rank coded, "
                                      "two gathers to rank 0, splice" if banded else "picsong_encode_frame on one GPU, one stream"),
                       "codeblocks": nCB, "stream_shorts": int(single.numel()),
                       "bits_per_pixel": round(single.numel() * 16 / (W * H), 4), "form": "banded" if banded else "single"},
            "timed_seconds": round(dt, 3), "roundtrip_ok": roundtrip_ok, "range_flag": flag,
            "stream_crc32": "%08x" % crc_one,
            "exchange": None if not banded else {
                "form": "LL1 all-gather (P/4 samples) + two gathers of codeblock-stripe mini-streams to rank 0 per frame",
                "splice_crc32": "%08x" % crc_splice, "splice_equals_single_gpu_stream": ok_splice,
                "ranks_seen": witness["ranks_seen"] if witness else None,
                "distinct_devices": witness["distinct_devices"] if witness else None,
                "devices": witness["devices"] if witness else None},
            "stage_ms": None if stage is None else {"dwt": round(float(stage[0]), 4), "bpc": round(float(stage[1]), 4),
                                                    "pack": round(float(stage[2]), 4),
                                                    "note": "HIP events on the launch stream inside the timed region"},
            "roofline": roofline, "roofline_dwt": roofline_dwt, "cpu_baseline": cpu}
    print(json.dumps(line))
    if world > 1 or witness:
        dist.destroy_process_group()


def dry_main(args, rank, world):
    """--dry: the N-rank protocol of this benchmark without a GPU and without the codec (CPU tests of the launcher):
    rendezvous, per-step exchange of stand-in codestreams over picsong_dist.gather_step, barrier-bracketed timing,
    MAX over ranks, one JSON line on rank 0."""
    if args.workload == "16k_intra":
        return dry_intra(args, rank, world)
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
    ap.add_argument("--phase", choices=["all", "pipelined", "iso", "lone"], default="all",
                    help="traced runs (tools/collect_profiles.sh): run ONE launch shape only, so that every kernel's average in a "
                         "rocprofv3 summary is that shape's -- 'pipelined': the timed loop alone (no output checks, no round trip); "
                         "'iso': --batch frames per call on one stream, nothing else on the GPU; 'lone': one frame per call on one "
                         "stream.  Prints a reduced line.  'all' (default): the contract line")
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
        args.steps = 80 if args.workload.startswith("4k") else (400 if args.workload == "16k_intra" else 22)
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

    witness = rccl_witness(torch, dist, local_rank, world) if exch else None

    import picsong_amd as pa
    import picsong_dist as pdist
    import oracle_lib as orc          # checker + cpu_baseline only

    if args.workload == "16k_intra":
        return intra_main(args, rank, local_rank, world, witness)
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

    phase = args.phase
    if phase in ("iso", "lone"):
        # ---- a traced run of ONE launch shape: `--batch` frames per call (iso) or one frame per call (lone) on one
        # stream with nothing else on the GPU; no other launch of the library happens in this process
        nb = batch if phase == "iso" else 1
        calls = max(12, args.steps)
        iso = pa.Codec(W, H, wl=wl, lossy=lossy, qs=qs, lut_folder=lut_dir, device=local_rank)
        dst = outs[0]
        for i in range(2 + calls):
            if i == 2:
                torch.cuda.synchronize()
                iso.profile_begin(calls)
            f0 = (i * nb) % pool_n
            if nb == 1:
                iso.encode_frame_async(pool[f0], dst[0], 1)
            else:
                iso.encode_frames_async(pool[f0:f0 + nb], dst, 1)
        torch.cuda.synchronize()
        ms = iso.profile_read(calls).mean(axis=0) / nb
        if rank == 0:
            print(json.dumps({"phase": phase, "workload": args.workload, "frames_per_call": nb, "calls": calls, "streams": 1,
                              "stage_ms_per_frame": {"dwt": round(float(ms[0]), 4), "bpc": round(float(ms[1]), 4),
                                                     "pack": round(float(ms[2]), 4)},
                              "mpixels_per_s": round(W * H / (float(ms.sum()) * 1e-3) / 1e6, 1),
                              "source": library_hashes()}))
        return

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
    if phase == "pipelined":
        # ---- a traced run of the pipelined shape alone: the timed loop and nothing else (no output checks, no isolated
        # phases, no round trip -- each of those would add launches of another shape to the trace)
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        if rank == 0:
            sm = stage_ms.mean(axis=0)
            print(json.dumps({"phase": phase, "workload": args.workload, "frames_per_call": batch, "streams": nstreams,
                              "n_gpus": world, "steps": args.steps, "frames_per_step": fps,
                              "ms_per_step": round(dt / args.steps * 1e3, 4),
                              "value": round((W * H * world * args.steps * fps) / dt / 1e6, 2), "unit": "Mpixels/s",
                              "stage_ms_per_frame": {"dwt": round(float(sm[0]), 4), "bpc": round(float(sm[1]), 4),
                                                     "pack": round(float(sm[2]), 4)},
                              "source": library_hashes()}))
        if exch:
            dist.destroy_process_group()
        return
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

    # ---- the pipelined shape with ONE frame per call (the reference's CodingEngine hands its workers single frames,
    # Engines/CodingEngine.cu:990-1061; rounds 1-2 of this benchmark ran that shape): same streams, same pool, a
    # shorter loop -- the figure that is like for like with the earlier rounds' headline
    one_per_call = None
    if batch != 1 and world == 1 and not exch:
        nf = max(nstreams * 4, min(fps, 120))
        def run1(n):
            for i in range(n):
                k = i % nstreams
                with torch.cuda.stream(streams[k]):
                    codecs[k].encode_frame_async(pool[i % pool_n], outs[k][0], 1)
        run1(nf)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(4):
            run1(nf)
        torch.cuda.synchronize()
        d1 = (time.perf_counter() - t1) / (4 * nf)
        one_per_call = {"value": round(W * H / d1 / 1e6, 2), "unit": "Mpixels/s", "ms_per_frame": round(d1 * 1e3, 5),
                        "frames": 4 * nf, "streams": nstreams,
                        "note": "picsong_encode_frame calls alternating over the streams (rounds 1-2 ran this shape)"}

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
    # (1 GiB buffers: a plane of an 8K frame, 134 MB, lives in the 256 MiB Infinity Cache and "copied" at 8.5 TB/s --
    # above the HBM's own 8 TB/s; the guide's measured HBM copy is 6.3 TB/s)
    roof_n = 1 << 28                                       # int32 elements = 1 GiB
    _a = torch.zeros(roof_n, dtype=torch.int32, device="cuda")
    _b = torch.empty_like(_a)
    copy_gbs = _rate(lambda: _b.copy_(_a), 8 * roof_n, iters=6)
    fill_gbs = _rate(lambda: _b.fill_(1), 4 * roof_n, iters=6)
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
    dwt_req = dwt_required_bytes(P, wl, True, True) * batch     # what the fused int16 design has to move
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
    def _dwt_fr(ms_per_launch, frames):
        """One shape of the transform: S8(d)'s algorithmic bytes, the bytes the design must move, and (offline
        counters) the bytes it did move, each over the launches' time and the 8 TB/s peak."""
        t = ms_per_launch * 1e-3
        f = frames / batch
        d = {"ms": round(ms_per_launch, 4), "frames": frames,
             "achieved_survey_8d": round(dwt_b * f / t / 1e9, 2), "frac_survey_8d": round(dwt_b * f / t / 1e9 / HBM_PEAK_GBS, 5),
             "achieved_required": round(dwt_req * f / t / 1e9, 2), "frac_of_required": round(dwt_req * f / t / 1e9 / HBM_PEAK_GBS, 5)}
        if dwt_traffic:
            d["frac_of_traffic"] = round(dwt_traffic * f / t / 1e9 / HBM_PEAK_GBS, 5)
        return d
    dwt_fr = _dwt_fr(float(iso_ms[0]) * batch, batch)
    roofline_dwt = {"kernel": "dwt_fwd2_kernel (levels 0 + 1, u8 ingest fused) + dwt_fwd_kernel (levels >= 2)", "bound": "hbm",
                    # The top-level figures are the call shape of the timed loop (`frames_per_launch` frames per call) on ONE
                    # stream with nothing else on the GPU -- a kernel's own rate; `in_timed_region` holds the HIP-event times
                    # of the same launches while the other streams' coder kernels share the GPU.
                    # `frac` is against the bytes the design MUST move (int16 coded subbands, LL1 never written): it cannot
                    # exceed 1.  SURVEY 8(d)'s count (4-byte outputs, LL1 written and read) is kept beside it as
                    # `frac_survey_8d` -- the contract's figure, which the int16 / fused design can push past 1
                    "achieved": dwt_fr["achieved_required"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": dwt_fr["frac_of_required"], "frac_of_required": dwt_fr["frac_of_required"],
                    "frac_survey_8d": dwt_fr["frac_survey_8d"], "frac_of_traffic": dwt_fr.get("frac_of_traffic"),
                    "traffic": dwt_traffic, "required_bytes_per_launch": int(dwt_req),
                    "algorithmic_bytes_per_launch": int(dwt_b), "frames_per_launch": batch,
                    "avg_launch_ms": round(float(iso_ms[0]) * batch, 4),
                    "in_timed_region": _dwt_fr(dwt_launch_ms, batch),
                    "single_stream": _dwt_fr(float(iso_ms[0]) * batch, batch),
                    # a lone frame per call (lone_frame's transform stage): what the launches cost with nothing to share
                    "lone_frame": _dwt_fr(float(lone_ms[0]), 1),
                    "three_frames_per_call": (_dwt_fr(float(b3_ms[0]) * 3, 3) if b3_ms is not None else
                                              (_dwt_fr(float(iso_ms[0]) * 3, 3) if batch == 3 else None)),
                    "measured_roof": {"copy_i32_GBps": round(copy_gbs, 1), "fill_i32_GBps": round(fill_gbs, 1),
                                      "buffer_bytes": 4 << 28, "guide_hbm_copy_GBps": 6290.0,
                                      "required_frac_of_copy_single_stream":
                                          round(dwt_req / (float(iso_ms[0]) * batch * 1e-3) / 1e9 / copy_gbs, 5)},
                    "note": "all of a frame's level launches counted as one; the top-level figures (= `single_stream`) are the "
                            "timed loop's call shape on one stream with nothing else running, `in_timed_region` the HIP-event "
                            "times of the same launches inside the timed region, where the other streams' coder kernels share the GPU; "
                            "`lone_frame` one frame per call, `three_frames_per_call` picsong_encode_frames over three "
                            "frames on one stream (the level launches serve three frames each); the input frames rotate "
                            "over a pool larger than the Infinity Cache, so every frame's pixels come from HBM; "
                            "`measured_roof` is a plain device copy / fill of 1 GiB buffers (past the 256 MiB Infinity Cache)"}

    # ---- CPU baseline: the oracle on the box's host cores, rank 0, N = 1 only (cpu_baseline_leg)
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline_leg(orc, frame_np, AW, AH, W, H, wl, lossy, qs, args.cpu_sample_rows,
                               stream0.cpu().numpy().view(np.uint16))

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
            "payloads_ok": exchange_ok,
            "ranks_seen": witness["ranks_seen"], "distinct_devices": witness["distinct_devices"],
            "devices": witness["devices"], "backend": witness["backend"]},
        "stage_ms": {"dwt": round(dwt_ms, 4), "bpc": round(bpc_ms, 4), "pack": round(pack_ms, 4),
                     "note": "per frame, HIP events on the launch streams inside the timed region"},
        "stage_ms_single_stream": {"dwt": round(float(iso_ms[0]), 4), "bpc": round(float(iso_ms[1]), 4),
                                   "pack": round(float(iso_ms[2]), 4)},
        "lone_frame": {"ms": round(float(sum(lone_ms)), 4), "mpixels_per_s": round(W * H / (float(sum(lone_ms)) * 1e-3) / 1e6, 1),
                       "stage_ms": {"dwt": round(float(lone_ms[0]), 4), "bpc": round(float(lone_ms[1]), 4), "pack": round(float(lone_ms[2]), 4)},
                       "note": "one frame per call on one stream with nothing else on the GPU (HIP events around the stages): "
                               "what a single image costs"},
        "one_frame_per_call": one_per_call,
        "roofline": roofline, "roofline_dwt": roofline_dwt, "cpu_baseline": cpu,
    }
    if psnr is not None:
        line["psnr_db"] = round(psnr, 3)
    print(json.dumps(line))
    if exch:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
