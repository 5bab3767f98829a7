#!/usr/bin/env python3
"""Rewrites the round-3 table of profiles/README.md from the published files (profiles/r03_*), so that the numbers quoted
there are the files' own.  (tools/profiles_readme.py does the same for the round-2 files, whose CSVs have no '#' lines.)"""
import csv
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles", "r03_")


def J(n):
    return json.load(open(P + n))


def stats(n):
    out = {}
    for r in csv.DictReader(l for l in open(P + n) if not l.startswith("#")):
        if "picsong" in r["Name"]:
            out[re.sub(r"^void picsong::|^picsong::", "", r["Name"]).split("(")[0]] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
    return out


def pmc(n):
    out = {}
    for r in csv.DictReader(l for l in open(P + n) if not l.startswith("#")):
        out[(re.sub(r"^void picsong::|^picsong::", "", r["Kernel_Name"]).split("(")[0], r["Counter_Name"])] = float(r["MeanValue"])
    return out


def g(st, sub):
    c = [(k, v) for k, v in st.items() if sub in k]
    c.sort(key=lambda kv: -kv[1][0])
    return c[0][1]


b, b4, bl, b3, bl3 = J("bench.json"), J("bench_4k.json"), J("bench_8k_lossy.json"), J("bench_8k_b3.json"), J("bench_8k_lossy_b3.json")
ss, sb3, sl, slb3 = stats("kernel_stats_single_stream.csv"), stats("kernel_stats_b3.csv"), stats("kernel_stats_8k_lossy.csv"), stats("kernel_stats_8k_lossy_b3.csv")
sd, sdl, s3, s4 = stats("kernel_stats_decode.csv"), stats("kernel_stats_decode_8k_lossy.csv"), stats("kernel_stats.csv"), stats("kernel_stats_4k.csv")
hb, sq = pmc("pmc_hbm.csv"), pmc("pmc_sq.csv")
enc = "bpc_encode_kernel<false>"
coder, head, lv = g(ss, "bpc_encode_kernel")[1], g(ss, "dwt_fwd2_kernel")[1], g(ss, "dwt_fwd_kernel")
pack, scan = g(ss, "pack_kernel")[1], g(ss, "scan_sizes_kernel")[1]
rd, rdl, lf = b["roofline_dwt"], bl["roofline_dwt"], b["lone_frame"]
F, W = hb[(enc, "FETCH_SIZE")], hb[(enc, "WRITE_SIZE")]
hk = [k for k in hb if "dwt_fwd2" in k[0]]
hF = [hb[k] for k in hk if k[1] == "FETCH_SIZE"][0]
hW = [hb[k] for k in hk if k[1] == "WRITE_SIZE"][0]
dec_lines = [l.strip() for l in open(P + "decode.txt") if l.startswith("decode")]
vb = b["roofline"]["valu_busy"]
rows = []
rows.append(f"| `r03_bench.json` | the default bench line: 8K lossless, 3 streams x {b['config']['frames_per_call']} frames per call, {b['steps']} steps x {b['config']['frames_per_step']} frames ({b['timed_seconds']} s timed): **{b['value'] / 1e3:.1f} Gpixel/s, {b['ms_per_frame']:.4f} ms/frame** (round 2: 160.0); a lone frame (`lone_frame`) {lf['ms']:.3f} ms = {lf['mpixels_per_s'] / 1e3:.1f} Gpixel/s: DWT {lf['stage_ms']['dwt']:.4f} / coder {lf['stage_ms']['bpc']:.4f} / pack {lf['stage_ms']['pack']:.4f} ms; DWT of a lone frame {rd['lone_frame']['frac']:.2f} of 8 TB/s, three frames per call {rd['three_frames_per_call']['ms_per_frame']:.4f} ms per frame = {rd['three_frames_per_call']['frac']:.2f} (the coded subbands leave the transform as int16: fewer bytes move than the algorithmic count assumes); coder `traffic` {b['roofline']['traffic'] / 1e6:.1f} MB = {b['roofline']['traffic'] / b['roofline']['algorithmic_bytes_per_launch']:.2f} x algorithmic; `valu_busy` {vb['lone_kernel']['VALUBusy']:.2f} lone / {vb['pipelined']['VALUBusy_same_definition']:.2f} pipelined; CPU baseline {b['cpu_baseline']['value']:.1f} Mpixel/s on 16 threads, same codestream; `roofline.source`: the counters quoted are this library's |")
rows.append(f"| `r03_bench_4k.json`, `r03_bench_8k_lossy.json` | `--workload 4k_lossless`: **{b4['value'] / 1e3:.1f} Gpixel/s** (a lone 4K frame {b4['lone_frame']['mpixels_per_s'] / 1e3:.1f}); `--workload 8k_lossy`: **{bl['value'] / 1e3:.1f} Gpixel/s**, PSNR {bl.get('psnr_db')} dB, DWT lone {rdl['lone_frame']['frac']:.3f} / {bl['config']['frames_per_call']} frames per call **{rdl['single_stream']['frac']:.3f}** of 8 TB/s by the bench's events |")
rows.append(f"| `r03_bench_8k_b3.json`, `r03_bench_8k_lossy_b3.json` | `--streams 1 --batch 3`: {b3['value'] / 1e3:.1f} / {bl3['value'] / 1e3:.1f} Gpixel/s |")
rows.append(f"| `r03_kernel_stats_single_stream.csv` | `rocprofv3 --kernel-trace --stats`, `--streams 1`: `bpc_encode_kernel<false>` **{coder:.1f} us** (r02: 254), `dwt_fwd2_kernel<int, ..., true>` (levels 0 + 1, int16 subbands) **{head:.1f} us** (r02: 31.3) + 3 x {lv[1]:.1f} us = {head + 3 * lv[1]:.1f} us = **{255.9 / (head + 3 * lv[1]) / 8:.2f} of 8 TB/s for a lone frame**, pack {pack:.1f}, scan {scan:.1f} us |")
h3, l3 = g(sb3, "dwt_fwd2_kernel")[1], g(sb3, "dwt_fwd_kernel")[1]
per = (h3 + 3 * l3) / 3
rows.append(f"| `r03_kernel_stats_b3.csv` | the three-frames-per-call shape (`picsong_encode_frames`, one stream): head {h3:.1f} us + 3 x {l3:.1f} us per THREE frames = {per:.1f} us per frame ({255.9 / per / 8:.2f} x the algorithmic bytes over 8 TB/s by kernel time; {rd['three_frames_per_call']['ms_per_frame']:.4f} ms = {rd['three_frames_per_call']['frac']:.2f} by the bench's events, launch gaps included) |")
hl, ll, hl3, ll3 = g(sl, "dwt_fwd2_kernel")[1], g(sl, "dwt_fwd_kernel")[1], g(slb3, "dwt_fwd2_kernel")[1], g(slb3, "dwt_fwd_kernel")[1]
perl = (hl3 + 4 * ll3) / 3
rows.append(f"| `r03_kernel_stats_8k_lossy.csv`, `r03_kernel_stats_8k_lossy_b3.csv` | 9/7 wl 6: lone frame head {hl:.1f} us + 4 x {ll:.1f} us = {hl + 4 * ll:.1f} us = {256.2 / (hl + 4 * ll) / 8:.2f}; three frames per call {hl3:.1f} us + 4 x {ll3:.1f} us per three frames = **{perl:.1f} us per frame = {256.2 / perl / 8:.2f} of 8 TB/s** ({bl3['roofline_dwt']['single_stream']['avg_launch_ms'] / 3:.4f} ms = {bl3['roofline_dwt']['single_stream']['frac']:.2f} by events) -- the north star's 0.80 for the 9/7 transform, in the batched shape |")
rows.append(f"| `r03_kernel_stats.csv`, `r03_kernel_stats_4k.csv` | the default three-stream shape (kernels of three calls share the GPU: coder {g(s3, 'bpc_encode_kernel')[1]:.0f} us, head {g(s3, 'dwt_fwd2_kernel')[1]:.0f} us while sharing); 4K frames four to a launch (coder {g(s4, 'bpc_encode_kernel')[1]:.0f} us per launch) |")
dk, dkl = g(sd, "bpc_decode_kernel")[1], g(sdl, "bpc_decode_kernel")[1]
tot = sum(v[0] * v[1] for k, v in sd.items() if "dwt_inv" in k) / g(sd, "bpc_decode_kernel")[0]
totl = sum(v[0] * v[1] for k, v in sdl.items() if "dwt_inv" in k) / g(sdl, "bpc_decode_kernel")[0]
rows.append(f"| `r03_kernel_stats_decode.csv`, `..._decode_8k_lossy.csv` | decoder `<false, 8, true>` (the stream-direct instantiation) **{dk:.1f} / {dkl:.1f} us** (r02: 476 / 442 + the empty 16-plane launch), ONE launch; inverse DWT {tot:.0f} us (5/3) / {totl:.0f} us (9/7) per frame; `scan_stream_kernel` {g(sd, 'scan_stream_kernel')[1]:.1f} / {g(sdl, 'scan_stream_kernel')[1]:.1f} us in place of read_sizes + scan + unpack (this round's first collection: 4.7 + 8.3 + 17.8 us) |")
m = re.findall(r"bpc_decode_kernel[^\n]*\n\s+SQ_BUSY_CYCLES=\S+\s+SQ_INSTS_SALU=(\S+)\s+SQ_INSTS_VALU=(\S+)", open(P + "pmc_decode.txt").read())
rows.append(f"| `r03_pmc_decode.txt` | SQ counters of the decode path: decoder **{float(m[0][1]) / 1e6:.1f} M vector + {float(m[0][0]) / 1e6:.1f} M scalar** wave-instructions per 8K lossless frame (r02: 205.3 M + 120.0 M), {float(m[1][1]) / 1e6:.1f} M + {float(m[1][0]) / 1e6:.1f} M per 9/7 frame (162.0 M + 93.5 M) |")
rows.append(f"| `r03_pmc_hbm.csv` | FETCH_SIZE / WRITE_SIZE passes: coder FETCH x2 = {2 * F / 1e3:.1f} MB (int16 coefficients once + the plane scratch read back) + WRITE {W / 1e3:.1f} MB (16-bit codeword staging + the plane scratch) = **{(2 * F + W) / 1e3:.1f} MB = {(2 * F + W) / 1e3 / 153.7:.2f} x the 153.7 MB algorithmic** (r02: 247 MB, 1.61 x; this round's first pass 188.5 MB, the faster prologue raised it to 211 MB, the 16-bit staging brought it back: DESIGN 4.2); fused DWT head FETCH x2 = {2 * hF / 1e3:.1f} MB, WRITE {hW / 1e3:.1f} MB (r02: 135.7) |")
wc = sq[(enc, "SQ_WAVE_CYCLES")]
rows.append(f"| `r03_pmc_sq.csv` | two SQ passes (`--streams 1`): coder {sq[(enc, 'SQ_INSTS_VALU')] / 1e6:.1f} M VALU + {sq[(enc, 'SQ_INSTS_SALU')] / 1e6:.1f} M SALU per launch, {sq[(enc, 'SQ_INSTS_BRANCH')] / 1e6:.1f} M branches, {sq[(enc, 'SQ_INSTS_LDS')] / 1e6:.1f} M LDS; of its waves' cycles {100 * sq[(enc, 'SQ_ACTIVE_INST_ANY')] / wc:.0f} % issuing (`SQ_ACTIVE_INST_ANY` / `SQ_WAVE_CYCLES`), {100 * sq[(enc, 'SQ_WAIT_ANY')] / wc:.0f} % parked on `s_waitcnt`, {100 * sq[(enc, 'SQ_WAIT_INST_ANY')] / wc:.0f} % waiting for an issue slot |")
rows.append(f"| `r03_pmc_sq_pipelined.csv` | the same counters + `SQ_BUSY_CU_CYCLES`, `GRBM_GUI_ACTIVE`, `SQ_THREAD_CYCLES_VALU` collected over the DEFAULT command (`--streams 3`, 48 frames).  **rocprofv3 serialises the dispatches of a `--pmc` run** (the pass's own kernel trace: no two dispatches overlap; the plain trace of the same command: 107 overlapping pairs of 451) -- so these, too, are counters of kernels running ALONE, and the profiler cannot deliver a per-dispatch counter in the pipelined shape.  What they give: rocprof's `VALUBusy` (100 x `SQ_ACTIVE_INST_VALU` / CUs / `GRBM_GUI_ACTIVE`, one vector instruction = one quad-cycle of one of a CU's four SIMDs) = **{vb['lone_kernel']['VALUBusy']:.2f} for a lone coder launch**; the same definition applied to the counter-measured instructions of all of a frame's kernels ({vb['pipelined']['valu_wave_insts_per_frame_all_kernels'] / 1e6:.1f} M) and the pipelined time per frame ({b['ms_per_frame']:.3f} ms, driver-timed shape) = **{vb['pipelined']['VALUBusy_same_definition']:.2f}**: the vector ALUs are the limit with frames in flight (full-rate instructions take less than a quad-cycle, `r03_valu_probe.txt`), `bench.py` prints both (`roofline.valu_busy`) |")
rows.append("| `r03_valu_probe.txt` / `.json` | `tools/valu_probe` (built from `tools/valu_probe.hip` by the collection script): issue rates per instruction class, as round 2 |")
num = lambda l: int(re.search(r"= (\d+) Mpixel", l).group(1)) / 1e3
rows.append(f"| `r03_decode.txt` | `tools/decode_bench.py --streams=3`: lone frame **{num(dec_lines[0]):.1f} Gpixel/s** (r02: 58.1), pipelined **{num(dec_lines[1]):.1f}** (87.2); 9/7 wl 6: {num(dec_lines[2]):.1f}, pipelined **{num(dec_lines[3]):.1f}** (103.1); 4K: {num(dec_lines[4]):.1f} alone, {num(dec_lines[5]):.1f} over three streams, **{num(dec_lines[6]):.1f}** four to a `picsong_decode_frames` call (85.2) |")
rg = [l for l in open(P + "rgb_probe.txt") if l.startswith("RGB")]
rr = lambda l: (re.search(r"batched grid ([\d.]+) ms", l).group(1), re.search(r"per call ([\d.]+) ms: ratio ([\d.]+)", l).groups(), re.search(r"batched decode ([\d.]+) ms", l).group(1))
a0, a1 = rr(rg[0]), rr(rg[2])
pp0, pp1 = re.search(r": ([\d.]+) ms/frame", rg[1]).group(1), re.search(r": ([\d.]+) ms/frame", rg[3]).group(1)
rows.append(f"| `r03_rgb_probe.txt` | `tools/rgb_probe.py`, an 8K RGB frame on one stream: plane by plane {pp0} ms (lossless) / {pp1} ms (9/7); through the batched grid (`picsong_encode_rgb_frame`) **{a0[0]} / {a1[0]} ms** = {a0[1][1]} x / {a1[1][1]} x three grey frames per `picsong_encode_frames` call ({a0[1][0]} / {a1[1][0]} ms); batched decode {a0[2]} / {a1[2]} ms |")
rows.append("| `r03_lone_frame.txt` | `tools/lone_frame_time.py`: single-frame calls timed from Python (wall clock over 30 calls, launch overhead included): what an image codec's caller sees |")
mt = [l.split() for l in open(P + "modes_time.txt") if l.strip()]
rows.append("| `r03_modes_time.txt` | `tools/modes_time.py`: the other modes at 8K lossless -- a lone frame and three calls in flight, encode and decode -- beside the plain coder in the same harness: `-k 0.5`, `-k 1.5` (complexity-scalable bulk scan) and `-cp 3` run at roughly a half to a third of its rates (DESIGN 4.5 / 4.6) |")
rows.append("| `r03_fuzz_parity.txt` | `tools/fuzz_parity.py 60 7`: sixty random geometries / contents / transforms through the frame paths, single and batched, against the oracle: no mismatch |")
rows.append("| `r03_library.sha256` | the library all of the above belong to |")
path = os.path.join(ROOT, "profiles", "README.md")
txt = open(path).read()
i = txt.index("| file | what |\n|---|---|\n") + len("| file | what |\n|---|---|\n")
j = txt.index("\n## Round 2")
open(path, "w").write(txt[:i] + "\n".join(rows) + "\n" + txt[j:])
print("\n".join(r[:200] for r in rows))
