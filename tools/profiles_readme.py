#!/usr/bin/env python3
"""Regenerates the round table of profiles/README.md from the published summaries (profiles/<tag>_*), so that the
numbers quoted there are the files' own.  usage: tools/profiles_readme.py r02  (rewrites the block between the
'<!-- <tag> table -->' markers)."""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
P = lambda n: os.path.join(ROOT, "profiles", f"{tag}_{n}")


def J(n):
    return json.load(open(P(n)))


def stats(n):
    out = {}
    for r in csv.DictReader(open(P(n))):
        if "picsong" in r["Name"]:
            k = re.sub(r"^void picsong::|^picsong::", "", r["Name"])
            out[k.split("(")[0]] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
    return out


def find(st, sub, most=True):
    c = [(k, v) for k, v in st.items() if sub in k]
    c.sort(key=lambda kv: -kv[1][0])
    return c[0][1] if c else (0, float("nan"))


def pmc(n):
    out = {}
    for r in csv.DictReader(open(P(n))):
        out[(r["Kernel_Name"].split("(")[0].replace("void picsong::", ""), r["Counter_Name"])] = float(r["MeanValue"])
    return out


b, b4, bl, b3, bl3 = J("bench.json"), J("bench_4k.json"), J("bench_8k_lossy.json"), J("bench_8k_b3.json"), J("bench_8k_lossy_b3.json")
ss, sl, sd, s4 = stats("kernel_stats_single_stream.csv"), stats("kernel_stats_8k_lossy.csv"), stats("kernel_stats.csv"), stats("kernel_stats_4k.csv")
sdec, sdecl = stats("kernel_stats_decode.csv"), stats("kernel_stats_decode_8k_lossy.csv")
hb, sq = pmc("pmc_hbm.csv"), pmc("pmc_sq.csv")
enc = "bpc_encode_kernel<false>"
head_i, head_f = "dwt_fwd2_kernel<int, false, true, 8>", "dwt_fwd2_kernel<float, true, true, 8>"
fetch, write = hb[(enc, "FETCH_SIZE")] * 1024 * 2 / 1e6, hb[(enc, "WRITE_SIZE")] * 1024 / 1e6
alg = b["roofline"]["algorithmic_bytes_per_launch"] / 1e6
dec = open(P("decode.txt")).read()
decn = re.findall(r"= (\d+) Mpixel/s", dec)
rd, r3 = b["roofline_dwt"], b["roofline_dwt"]["three_frames_per_call"]
rl, rl3 = bl["roofline_dwt"], bl["roofline_dwt"]["three_frames_per_call"]
# the lean 9/7 synthesis kernel's launches of one decoded frame, finest level first (band height 32, 16, 8, 4 ...)
_inv97 = sorted(((k, v) for k, v in sdecl.items() if "dwt_inv97_kernel<" in k),
                key=lambda kv: (-int(kv[0].split("dwt_inv97_kernel<")[1].split(",")[0]), "false, true, " in kv[0]))
_per_frame = min(v[0] for _, v in _inv97) if _inv97 else 1
inv97_txt = " + ".join(("%d x %.1f" % (v[0] // _per_frame, v[1])) if v[0] // _per_frame > 1 else "%.1f" % v[1] for _, v in _inv97)
inv97_sum = sum(v[0] // _per_frame * v[1] for _, v in _inv97)
rows = [
    ("`%s_bench.json`" % tag,
     "the default bench line: 8K lossless, 3 streams x 1 frame per call, %d steps x %d frames (%.2f s timed): **%.1f Gpixel/s, %.4f ms/frame**; "
     "single-stream stages DWT %.4f (%.3f of 8 TB/s; three frames per call: %.4f ms per frame = **%.2f**) / coder %.4f / pack %.4f ms; "
     "`timed_loop_outputs_ok` %s, round trip %s; CPU baseline (oracle, %d threads of the box's quota) %.1f Mpixel/s with the same codestream"
     % (b["steps"], b["config"]["frames_per_step"], b["timed_seconds"], b["value"] / 1e3, b["ms_per_frame"],
        b["stage_ms_single_stream"]["dwt"], rd["single_stream"]["frac"], r3["ms_per_frame"], r3["frac"],
        b["stage_ms_single_stream"]["bpc"], b["stage_ms_single_stream"]["pack"], b["timed_loop_outputs_ok"], b["roundtrip_ok"],
        b["cpu_baseline"]["cores"], b["cpu_baseline"]["value"])),
    ("`%s_bench_4k.json`" % tag, "`--workload 4k_lossless` (3 streams x 4 frames per `picsong_encode_frames` call): **%.1f Gpixel/s**, %.4f ms per 4K frame (round 1: 59)"
     % (b4["value"] / 1e3, b4["ms_per_frame"])),
    ("`%s_bench_8k_lossy.json`" % tag, "`--workload 8k_lossy` (9/7, qs 0.5, wl 6): %.1f Gpixel/s, PSNR %.2f dB; DWT of a lone frame %.4f ms = %.2f of 8 TB/s, three frames per call %.4f ms per frame = **%.2f**"
     % (bl["value"] / 1e3, bl["psnr_db"], bl["stage_ms_single_stream"]["dwt"], rl["single_stream"]["frac"], rl3["ms_per_frame"], rl3["frac"])),
    ("`%s_bench_8k_b3.json`, `%s_bench_8k_lossy_b3.json`" % (tag, tag),
     "`--streams 1 --batch 3`: three 8K frames per call on ONE stream: %.1f / %.1f Gpixel/s; DWT %.4f / %.4f ms per frame; coder %.3f ms per frame inside a three-frame launch"
     % (b3["value"] / 1e3, bl3["value"] / 1e3, b3["stage_ms"]["dwt"], bl3["stage_ms"]["dwt"], b3["stage_ms"]["bpc"])),
    ("`%s_kernel_stats_single_stream.csv`" % tag,
     "`rocprofv3 --kernel-trace --stats`, `--streams 1 --no-b3`: `bpc_encode_kernel<false>` **%.0f us** (round 1: 430), `dwt_fwd2_kernel` (levels 0 + 1) %.1f us + 3 x %.1f us, pack %.1f us, scan %.1f us"
     % (ss[enc][1], ss[head_i][1], find(ss, "dwt_fwd_kernel<int")[1], find(ss, "pack_kernel")[1], find(ss, "scan_sizes")[1])),
    ("`%s_kernel_stats.csv`" % tag, "the default command shape (3 streams; kernels of three calls share the GPU: coder %.0f us, fused DWT head %.1f us while sharing)"
     % (sd[enc][1], sd[head_i][1])),
    ("`%s_kernel_stats_8k_lossy.csv`" % tag, "`--workload 8k_lossy --streams 1`: coder %.0f us (the wl = 6 LUT holes send the level-5 blocks through the raw fallback; their halves stop once their 4095 slots are used), `dwt_fwd2_kernel<float>` **%.1f us** + 4 x %.1f us"
     % (sl[enc][1], sl[head_f][1], find(sl, "dwt_fwd_kernel<float")[1])),
    ("`%s_kernel_stats_4k.csv`" % tag, "`--workload 4k_lossless` (4 frames per launch): coder %.0f us per 4-frame launch, fused DWT head %.1f us per 4 frames"
     % (s4[enc][1], s4[head_i][1])),
    ("`%s_kernel_stats_decode.csv`, `..._decode_8k_lossy.csv`" % tag,
     "`rocprofv3 --kernel-trace --stats -- python3 tools/decode_bench.py [lossy]`: decoder `<false, 8>` %.0f / %.0f us, inverse DWT %.1f + %.1f + 3 x %.1f us (5/3), %s = **%.0f us** (9/7, `dwt_inv97_kernel`, levels 0 .. 5; before it: 54.5 + 22.8 + 4 x 10.2 = 118), unpack %.1f / %.1f us"
     % (find(sdec, "bpc_decode_kernel<false, 8>")[1], find(sdecl, "bpc_decode_kernel<false, 8>")[1],
        find(sdec, "dwt_inv_kernel<int, false, 16")[1], find(sdec, "dwt_inv_kernel<int, false, 8")[1], find(sdec, "dwt_inv_kernel<int, false, 4")[1],
        inv97_txt, inv97_sum,
        find(sdec, "unpack_kernel")[1], find(sdecl, "unpack_kernel")[1])),
    ("`%s_pmc_hbm.csv`" % tag,
     "FETCH_SIZE / WRITE_SIZE passes: coder FETCH x2 = %.1f MB (coefficients once + the plane scratch read back) + WRITE %.1f MB = **%.0f MB = %.2f x the %.1f MB algorithmic** (round 1: 344 MB, 2.24 x); fused DWT head FETCH x2 = %.1f MB, WRITE %.1f MB"
     % (fetch, write, fetch + write, (fetch + write) / alg, alg, hb[(head_i, "FETCH_SIZE")] * 2048 / 1e6, hb[(head_i, "WRITE_SIZE")] * 1024 / 1e6)),
    ("`%s_pmc_sq.csv`" % tag,
     "two SQ passes: coder **%.1f M VALU + %.1f M SALU** wave-instructions per 8K launch (round 1: 165.5 M + 106.4 M), %.1f M branches, %.1f M LDS; fused 5/3 DWT head %.2f M VALU + %.2f M SALU (round 1: 8.81 M + 6.03 M); `SQ_ACTIVE_INST_VALU` = `SQ_INSTS_VALU` (it counts instructions on this part)"
     % (sq[(enc, "SQ_INSTS_VALU")] / 1e6, sq[(enc, "SQ_INSTS_SALU")] / 1e6, sq[(enc, "SQ_INSTS_BRANCH")] / 1e6, sq[(enc, "SQ_INSTS_LDS")] / 1e6,
        sq[(head_i, "SQ_INSTS_VALU")] / 1e6, sq[(head_i, "SQ_INSTS_SALU")] / 1e6)),
    ("`%s_valu_probe.txt` / `.json`" % tag,
     "`tools/valu_probe`: issue rates per instruction class, 1..8 waves per SIMD (DESIGN.md 4.0): and/or/xor/add/sub/mov on VGPR or literal operands, `v_add/mul/fmac_f32` 0.38-0.43 per cycle per SIMD; shifts, min, compares, every VOP3 form (`v_fma_f32` too), packed fp32, 24-bit multiplies, DPP, SDWA, conversions, any SGPR operand 0.22-0.27; `v_cndmask_e32` on a scalar-written VCC 0.044; scalar ALU 0.23 per SIMD"),
    ("`%s_decode.txt`" % tag, "`tools/decode_bench.py --streams=3` (lossless) and `lossy`: lone frame %.1f Gpixel/s, pipelined **%.1f Gpixel/s**; 9/7 wl 6: %.1f, pipelined **%.1f**; 4K frames: %.1f alone, %.1f over three streams, **%.1f** four to a `picsong_decode_frames` call; round trip checked"
     % tuple(int(x) / 1e3 for x in (decn + ["0"] * 7)[:7])),
]
insts = sq[(enc, "SQ_INSTS_VALU")] / 1e6
table = "| file | what |\n|---|---|\n" + "\n".join("| %s | %s |" % r for r in rows) + "\n"
tail = ("\nThe coder's issue arithmetic from these files: %.1f M vector instructions per frame; about a third of them (adds, subs, moves,\n"
        "VGPR-operand ands) are of the 2.6-cycle class, the rest of the 4.2-cycle class: about %.2f ms of issue for coder + transform on\n"
        "1024 SIMDs at the 2.39 GHz the probe measures under load, against %.4f ms per frame in the bench: **%.2f of issue saturation**\n"
        "(`roofline.valu_issue.frac_of_probe_half_rate_peak` takes every instruction at the half rate and so prints > 1).  Against the guide's\n"
        "2 cycles per instruction the same figure is %.2f.\n"
        % (insts, (insts + 8.0) * (0.35 * 2.6 + 0.65 * 4.2) / 1024 / 2.39, b["ms_per_frame"],
           (insts + 8.0) * (0.35 * 2.6 + 0.65 * 4.2) / 1024 / 2.39 / b["ms_per_frame"], (insts + 8.0) * 2.0 / 1024 / 2.39 / b["ms_per_frame"]))
path = os.path.join(ROOT, "profiles", "README.md")
s = open(path).read()
m0, m1 = "<!-- %s table -->" % tag, "<!-- /%s table -->" % tag
block = m0 + "\n" + table + tail + m1
if m0 in s:
    s = s[:s.index(m0)] + block + s[s.index(m1) + len(m1):]
else:
    print("markers not found: printing"); print(block); sys.exit(1)
open(path, "w").write(s)
print(block)
