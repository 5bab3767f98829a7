#!/usr/bin/env python3
"""Makes the probability tables of the 3-coding-pass mode (-cp 3) for the tests.

The reference reads five table files per component when codingPasses == 3 -- ref, sig, sign and the
cleanup pass's cp_sig, cp_sign (IO/IOManager.ipp:404-606) -- but ships none for that mode (its LUT folders
hold ref / sig / sign only), so its own tree cannot run -cp 3.  This script builds a complete folder from
the shipped n1_* tables: ref / sig / sign R files are the fixtures of tests/golden/lut/ unchanged, cp_sig and
cp_sign are derived from sig and sign by a fixed rule (a cleanup-pass coefficient has no significant
neighbour when the significance pass meets it, so zeros are likelier: p0' = min(127, (3 p0 + 127) / 4);
the sign table is mirrored about 64).  What the rule is does not matter to the tests -- they check that
oracle, emulated kernels and HIP kernels agree on WHICH table entry every symbol uses -- only that the cp_*
tables differ from sig / sign everywhere, so that a wrong section or offset cannot go unnoticed.

usage: python tests/golden/make_cp3_tables.py     (writes tests/golden/lut_cp3/<folder>/)"""
import os
import shutil

HERE = os.path.dirname(os.path.abspath(__file__))


def derive(src, dst, rule):
    out = []
    for line in open(src):
        if ":" not in line:
            continue
        key, vals = line.split(":")
        out.append(key.strip() + " : " + " ".join(str(rule(int(v))) for v in vals.split()) + " \n")
    with open(dst, "w") as f:
        f.writelines(out)


def main():
    for folder in ("n1_lossless", "n1_lossy"):
        src = os.path.join(HERE, "lut", folder)
        dst = os.path.join(HERE, "lut_cp3", folder)
        os.makedirs(dst, exist_ok=True)
        shutil.copy(os.path.join(src, "header.txt"), os.path.join(dst, "header.txt"))
        for stem in ("ref", "sig", "sign"):
            shutil.copy(os.path.join(src, stem + "R.txt_0"), os.path.join(dst, stem + "R.txt_0"))
        derive(os.path.join(src, "sigR.txt_0"), os.path.join(dst, "cp_sigR.txt_0"),
               lambda v: min(127, (3 * v + 127) // 4))
        derive(os.path.join(src, "signR.txt_0"), os.path.join(dst, "cp_signR.txt_0"),
               lambda v: max(1, min(127, 128 - v)))


if __name__ == "__main__":
    main()
