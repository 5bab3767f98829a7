#!/usr/bin/env python3
"""Writes the round-4 section of profiles/README.md from the published files themselves (profiles/r04_*), so that every
number quoted there IS a file's: a kernel time is one row of a kernel_stats CSV (AverageNs), a fraction is one such row
and one division.  Usage: python tools/profiles_readme.py > (replaces the text between the r04 markers)."""
import csv
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles", "r04_")


def J(n):
    return json.load(open(P + n))


def short(name):
    return re.sub(r"^void picsong::|^picsong::", "", name).split("(")[0]


def stats(n):
    out = {}
    for r in csv.DictReader(open(P + "kernel_stats_" + n + ".csv")):
        if "picsong" in r["Name"]:
            out[short(r["Name"])] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3)
    return out


def pmc(n):
    out = {}
    for r in csv.DictReader(l for l in open(P + "pmc_" + n + ".csv") if not l.startswith("#")):
        out[(short(r["Kernel_Name"]), r["Counter_Name"])] = float(r["MeanValue"])
    return out


def find(st, sub):
    c = [(k, v) for k, v in st.items() if sub in k]
    c.sort(key=lambda kv: -kv[1][0] * kv[1][1])
    return c[0][1]


def levels(st, sub, per_call):
    """all launches of the per-level kernels of one call: sum of calls x avg over the instantiations / calls of the head"""
    return sum(v[0] * v[1] for k, v in st.items() if sub in k) / per_call


def main():
    b, b4, bl, b16, b16b = J("bench.json"), J("bench_4k.json"), J("bench_8k_lossy.json"), J("bench_16k_intra.json"), J("bench_16k_intra_banded_w1.json")
    rows = []
    A = rows.append
    A("| file | what it is, and the figures DESIGN.md / README.md take from it |")
    A("|---|---|")
    lf = b["lone_frame"]
    A(f"| `r04_bench.json` | the contract line (`python bench.py`): 8K `-type 0` wl 5, 3 streams x {b['config']['frames_per_call']} frames per call, {b['steps']} steps x {b['config']['frames_per_step']} frames: **{b['value'] / 1e3:.1f} Gpixel/s**, {b['ms_per_frame']:.4f} ms/frame; `one_frame_per_call` {b['one_frame_per_call']['value'] / 1e3:.1f}; `lone_frame` {lf['ms']:.3f} ms = {lf['mpixels_per_s'] / 1e3:.1f} Gpixel/s (DWT {lf['stage_ms']['dwt']:.4f} / coder {lf['stage_ms']['bpc']:.4f} / pack {lf['stage_ms']['pack']:.4f}); `roofline.traffic` {b['roofline']['traffic'] / 1e6:.1f} MB per three-frame launch; `cpu_baseline` {b['cpu_baseline']['value']:.1f} Mpixel/s on {b['cpu_baseline']['cores']} threads, same codestream |")
    A(f"| `r04_bench_4k.json`, `r04_bench_8k_lossy.json` | `--workload 4k_lossless`: **{b4['value'] / 1e3:.1f} Gpixel/s** (lone frame {b4['lone_frame']['mpixels_per_s'] / 1e3:.1f}); `--workload 8k_lossy`: **{bl['value'] / 1e3:.1f} Gpixel/s**, PSNR {bl.get('psnr_db')} dB, lone frame {bl['lone_frame']['mpixels_per_s'] / 1e3:.1f}; `traffic` non-null in both |")
    A(f"| `r04_bench_16k_intra.json`, `..._banded_w1.json` | `--workload 16k_intra` (BASELINE configs[4] on one GPU): **{b16['value'] / 1e3:.1f} Gpixel/s**, {b16['ms_per_step']:.3f} ms per frame (DWT {b16['stage_ms']['dwt']:.3f} / coder {b16['stage_ms']['bpc']:.3f} / pack {b16['stage_ms']['pack']:.3f}), round trip {b16['roundtrip_ok']}; `--force-exchange`: the banded form at world = 1 through RCCL, {b16b['ms_per_step']:.2f} ms, `splice_equals_single_gpu_stream` {b16b['exchange']['splice_equals_single_gpu_stream']}, `ranks_seen` {b16b['exchange']['ranks_seen']} |")
    sp, s3, s1 = stats("pipelined"), stats("b3"), stats("lone")
    c3, c1 = find(s3, "bpc_encode_kernel"), find(s1, "bpc_encode_kernel")
    h3, h1 = find(s3, "dwt_fwd2_kernel"), find(s1, "dwt_fwd2_kernel")
    l3, l1 = levels(s3, "dwt_fwd_kernel", h3[0]), levels(s1, "dwt_fwd_kernel", h1[0])
    cp = find(sp, "bpc_encode_kernel")
    A(f"| `r04_kernel_stats_pipelined.csv` | `rocprofv3 --kernel-trace --stats -- python3 bench.py --phase pipelined --steps 2 --warmup 1 --frames-per-step 36`: the headline's shape and nothing else (three-frame launches of three streams sharing the GPU): `bpc_encode_kernel` avg **{cp[1]:.0f} us** (min {cp[2]:.0f}, max {cp[3]:.0f}: they co-reside) = the line's `roofline.avg_launch_ms` |")
    A(f"| `r04_kernel_stats_b3.csv` | `--phase iso --batch 3`: three frames per call, one stream: coder {c3[1]:.1f} us (min {c3[2]:.1f} / max {c3[3]:.1f}); head {h3[1]:.1f} + levels {l3:.1f} us per three frames = **{(h3[1] + l3) / 3:.1f} us per frame**: 255.9 MB / that / 8 TB/s = **{255.8976 / ((h3[1] + l3) / 3) / 8:.2f}** (SURVEY 8d bytes), 122.2 MB / that / 8 = {122.204 / ((h3[1] + l3) / 3) / 8:.2f} (required bytes) |")
    A(f"| `r04_kernel_stats_lone.csv` | `--phase lone`: one frame per call: coder **{c1[1]:.1f} us** (min {c1[2]:.1f} / max {c1[3]:.1f}), head {h1[1]:.1f} + levels {l1:.1f} = {h1[1] + l1:.1f} us: {255.8976 / (h1[1] + l1) / 8:.2f} (8d) / {122.204 / (h1[1] + l1) / 8:.2f} (required); pack {find(s1, 'pack_kernel')[1]:.1f}, scan {find(s1, 'scan_sizes')[1]:.1f} us |")
    y6, y1 = stats("lossy_b6"), stats("lossy_lone")
    hy6, hy1 = find(y6, "dwt_fwd2_kernel"), find(y1, "dwt_fwd2_kernel")
    ly6, ly1 = levels(y6, "dwt_fwd_kernel", hy6[0]), levels(y1, "dwt_fwd_kernel", hy1[0])
    A(f"| `r04_kernel_stats_lossy_b6.csv`, `..._lossy_lone.csv`, `..._lossy_pipelined.csv` | 8K 9/7 wl 6: six frames per call head {hy6[1]:.1f} + levels {ly6:.1f} us per six = **{(hy6[1] + ly6) / 6:.1f} us per frame = {256.162 / ((hy6[1] + ly6) / 6) / 8:.2f} of 8 TB/s** by 8d bytes ({122.465 / ((hy6[1] + ly6) / 6) / 8:.2f} by required); lone {hy1[1]:.1f} + {ly1:.1f} = {hy1[1] + ly1:.1f} us = {256.162 / (hy1[1] + ly1) / 8:.2f} ({122.465 / (hy1[1] + ly1) / 8:.2f}); coder lone {find(y1, 'bpc_encode_kernel')[1]:.1f} us |")
    k6, k1 = stats("4k_b6"), stats("4k_lone")
    A(f"| `r04_kernel_stats_4k_b6.csv`, `..._4k_lone.csv`, `..._4k_pipelined.csv` | 4K: six frames per call coder {find(k6, 'bpc_encode_kernel')[1]:.1f} us per launch; a lone 4K frame coder **{find(k1, 'bpc_encode_kernel')[1]:.1f} us** (1020 waves: one per SIMD), head {find(k1, 'dwt_fwd2_kernel')[1]:.1f} us |")
    s16 = stats("16k")
    A(f"| `r04_kernel_stats_16k.csv` | `--workload 16k_intra --phase pipelined`: coder {find(s16, 'bpc_encode_kernel')[1]:.0f} us, head {find(s16, 'dwt_fwd2_kernel')[1]:.0f} us, pack {find(s16, 'pack_kernel')[1]:.0f} us, scan {find(s16, 'scan_sizes')[1]:.0f} us per 16K x 16K frame |")
    A("| `r04_kernel_stats_*.line.json` | the reduced line each traced run printed (`phase`, per-frame stage times by HIP events, the library's hashes) |")
    d5, d9 = stats("decode"), stats("decode_lossy")
    dk5, dk9 = find(d5, "bpc_decode_kernel"), find(d9, "bpc_decode_kernel")
    i5 = sum(v[0] * v[1] for k, v in d5.items() if "dwt_inv" in k) / dk5[0]
    i9 = sum(v[0] * v[1] for k, v in d9.items() if "dwt_inv" in k) / dk9[0]
    A(f"| `r04_kernel_stats_decode.csv`, `..._decode_lossy.csv` | `tools/decode_bench.py [lossy]` (lone frames): decoder **{dk5[1]:.1f} / {dk9[1]:.1f} us**; inverse transform {i5:.1f} us (5/3: `dwt_inv2_kernel` {find(d5, 'dwt_inv2_kernel')[1]:.1f} + the small levels) / {i9:.1f} us (9/7) per frame; `scan_stream_kernel` {find(d5, 'scan_stream')[1]:.1f} us |")
    hb, sq = pmc("hbm"), pmc("sq")
    enc = [k for k in {k[0] for k in hb} if "bpc_encode_kernel" in k][0]
    head = [k for k in {k[0] for k in hb} if "dwt_fwd2_kernel" in k][0]
    F, W = hb[(enc, "FETCH_SIZE")] * 1024 / 1e6, hb[(enc, "WRITE_SIZE")] * 1024 / 1e6
    hF, hW = hb[(head, "FETCH_SIZE")] * 1024 / 1e6, hb[(head, "WRITE_SIZE")] * 1024 / 1e6
    A(f"| `r04_pmc_hbm.csv`, `r04_pmc_hbm_8k_lossy.csv`, `r04_pmc_hbm_4k_lossless.csv` | FETCH_SIZE / WRITE_SIZE passes (`--phase lone`, unit 1024 B, mean per dispatch; FETCH doubled per the guide's gfx950 note): 8K lossless coder 2 x {F:.1f} + {W:.1f} = **{2 * F + W:.1f} MB = {(2 * F + W) / 153.7:.2f} x the 153.7 MB algorithmic**; fused head 2 x {hF:.1f} + {hW:.1f} = {2 * hF + hW:.1f} MB |")
    wc = sq[(enc, "SQ_WAVE_CYCLES")]
    A(f"| `r04_pmc_sq.csv` (+ `_8k_lossy`, `_4k_lossless`) | two SQ passes: coder **{sq[(enc, 'SQ_INSTS_VALU')] / 1e6:.1f} M VALU + {sq[(enc, 'SQ_INSTS_SALU')] / 1e6:.1f} M SALU** per 8K frame, {sq[(enc, 'SQ_INSTS_BRANCH')] / 1e6:.1f} M branches, {sq[(enc, 'SQ_INSTS_LDS')] / 1e6:.1f} M LDS; of its waves' cycles {100 * sq[(enc, 'SQ_ACTIVE_INST_ANY')] / wc:.0f} % issuing, {100 * sq[(enc, 'SQ_WAIT_ANY')] / wc:.0f} % on `s_waitcnt`, {100 * sq[(enc, 'SQ_WAIT_INST_ANY')] / wc:.0f} % waiting for an issue slot |")
    vb = b["roofline"]["valu_busy"]
    A(f"| `r04_pmc_sq_pipelined.csv` | the VALUBusy terms over `--phase pipelined` (rocprofv3 serialises the dispatches of a `--pmc` run: counters of kernels running alone): `VALUBusy` {vb['lone_kernel']['VALUBusy']:.2f} for a lone three-frame launch; the same definition over all of a frame's kernels ({vb['pipelined']['valu_wave_insts_per_frame_all_kernels'] / 1e6:.1f} M instructions) and the line's {b['ms_per_frame']:.4f} ms per frame: {vb['pipelined']['VALUBusy_same_definition']:.2f} |")
    m = re.findall(r"bpc_decode_kernel[^\n]*\n\s+SQ_BUSY_CYCLES=\S+\s+SQ_INSTS_SALU=(\S+)\s+SQ_INSTS_VALU=(\S+)", open(P + "pmc_decode.txt").read())
    fw = re.findall(r"(bpc_decode_kernel|dwt_inv2_kernel)[^\n]*\n\s+(FETCH_SIZE|WRITE_SIZE)=(\S+)", open(P + "pmc_decode.txt").read())
    fwd = {(a, c): float(v) * 1024 / 1e6 for a, c, v in fw}
    A(f"| `r04_pmc_decode.txt` | `tools/pmc_decode.sh`: decoder **{float(m[0][1]) / 1e6:.1f} M vector + {float(m[0][0]) / 1e6:.1f} M scalar** per 8K 5/3 frame, {float(m[1][1]) / 1e6:.1f} M + {float(m[1][0]) / 1e6:.1f} M per 9/7 frame; HBM: decoder FETCH x 2 = {2 * fwd[('bpc_decode_kernel', 'FETCH_SIZE')]:.1f} MB, WRITE **{fwd[('bpc_decode_kernel', 'WRITE_SIZE')]:.1f} MB** (int16 coefficients + the plane scratch); `dwt_inv2_kernel` FETCH x 2 = {2 * fwd[('dwt_inv2_kernel', 'FETCH_SIZE')]:.1f} MB, WRITE {fwd[('dwt_inv2_kernel', 'WRITE_SIZE')]:.1f} MB |")
    dec = [l.strip() for l in open(P + "decode.txt") if l.startswith("decode")]
    num = lambda l: int(re.search(r"= (\d+) Mpixel", l).group(1)) / 1e3
    A(f"| `r04_decode.txt` | `tools/decode_bench.py --streams=3`: 8K 5/3 lone **{num(dec[0]):.1f}** / three calls in flight **{num(dec[1]):.1f} Gpixel/s**; 9/7 {num(dec[2]):.1f} / **{num(dec[3]):.1f}**; 4K {num(dec[4]):.1f} alone, {num(dec[5]):.1f} over three streams, **{num(dec[6]):.1f}** four to a `picsong_decode_frames` call |")
    A("| `r04_modes_time.txt` | `tools/modes_time.py`: a lone 8K frame and three hinted calls in flight, encode and decode, for k = 0, `-k 0.5`, `-k 1.5`, `-cp 3` (DESIGN 4.5 / 4.6) |")
    A("| `r04_lone_frame.txt`, `r04_rgb_probe.txt`, `r04_fuzz_parity.txt` | `tools/lone_frame_time.py` (single-frame calls timed from Python, launch overhead included); `tools/rgb_probe.py` (an 8K RGB frame through one launch per stage against three grey frames: DESIGN 4.7's table); `tools/fuzz_parity.py 120 4` (120 random geometries / contents / transforms / coder modes — 39 with -k > 0, 19 with -cp 3 — through the frame paths against the oracle: 0 mismatches) |")
    A("| `r04_kernel_stats_rgb.csv` | `rocprofv3 --kernel-trace --stats -- python3 tools/rgb_probe.py`: the RGB frame paths' kernels (DESIGN 4.7): the fused heads `dwt_fwd2_kernel<..., RGB>`, `dwt_inv_rgb_kernel`, `dwt_inv97_rgb_kernel`, next to the grey heads over the same three planes |")
    A("| `r04_bench_exchange_w1.json` | `python bench.py --force-exchange --steps 6 --no-cpu-baseline`: the frame-sharded path's per-step exchange (`picsong_dist.gather_step`: one all-gather of lengths + one grouped point-to-point batch) through RCCL at world = 1 -- the only form a one-GPU box can run; `exchange.ranks_seen` 1 |")
    A("| `r04_pmc_kmode.txt` | `tools/kmode_prof.sh`: the -k 0.5 kernels over lone 8K frames: encoder (the lone frame's register-rich instantiation) 314.5 us, 136.0 M vector + 70.9 M scalar instructions; decoder (compact table copies, int16 coefficients out) 446.1 us, 169.2 M + 95.5 M |")
    A("| `r04_video_k_time.txt` | `tools/video_k_time.py 0.5 6`: 4K frames on three streams, one and six to a `picsong_encode_frames` / `picsong_decode_frames` call, k = 0 and k = 0.5 (the -k > 0 contexts' batched launches) |")
    A("| `r04_valu_probe.txt` / `.json` | `tools/valu_probe`: issue rates per instruction class (DESIGN 4) |")
    A("| `r04_library.sha256` | line 1: sha256 of the `libpicsong_hip.so` all of the above belong to; line 2: `bench.py: source_hash()` of the kernel sources it was built from -- what `bench.py` compares before it quotes an offline counter (`roofline.source`) |")
    text = "\n".join(rows)
    path = os.path.join(ROOT, "profiles", "README.md")
    s = open(path).read()
    a, z = "<!-- r04 table -->", "<!-- /r04 table -->"
    assert a in s and z in s
    s = s[:s.index(a) + len(a)] + "\n" + text + "\n" + s[s.index(z):]
    open(path, "w").write(s)
    print(text)


if __name__ == "__main__":
    main()
