#!/usr/bin/env python3
"""Static VALU instruction mix of the extractor kernels by measured issue class (tools/micro/valu_rate*.hip on MI355X,
profiles/r03_issue_rates.txt): "fast" instructions issue every ~2.5 cycles per SIMD (1.06 ns: v_add/sub_u32, and/or/xor,
mov, lshr/ashr_b32, 16-bit VOP2 arithmetic, f32 add/mul/fma on VGPR operands), everything else every ~4.2 cycles (1.78 ns:
v_min/max_*32, v_min3/max3, every v_pk_*, v_cmp_*, v_perm, v_dot*, v_mad/mul_*24, v_lshl*, cvt, DPP / SDWA forms, any VALU
instruction with an SGPR operand).  Prints per kernel the static counts and the mean issue cost of its mix; bench.py prices
`roofline.valu_issue` with it (PMC SQ_INSTS_VALU x that cost / 1024 SIMDs).

  python tools/valu_mix.py            -> JSON on stdout (compiles csrc/orbhip_extractor.hip to ISA with hipcc -S)
"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAST_NS, SLOW_NS, F64_NS = 1.06, 1.78, 2.26
FAST = re.compile(r"^v_(add|sub|subrev)_(u32|co_u32|f32|u16|i16)|^v_(and|or|xor)_b32|^v_mov_b32|^v_(lshrrev|ashrrev)_(b32|i32)|"
                  r"^v_(max|min)_(u16|i16|f16)|^v_(mul|fma|mac|fmac)_f32|^v_lshlrev_b16|^v_mul_lo_u16|^v_(add|sub)_f16")


def classify(line):
    m = line.split()
    op = m[0]
    if not op.startswith("v_"):
        return None
    ops = " ".join(m[1:])
    if "_sdwa" in op or "_dpp" in op or "sdwa" in ops or "row_" in ops or "quad_perm" in ops:
        return "slow"
    body = op.replace("_e32", "").replace("_e64", "")
    if body.endswith("_f64") and not body.startswith("v_cvt") and not body.startswith("v_cmp"):
        return "f64"
    if FAST.match(body):
        # an SGPR source operand moves the instruction to the slow class
        srcs = ops.split(",")[1:]
        if any(re.match(r"\s*(s\d+|s\[\d+:\d+\]|vcc|exec)", s) for s in srcs):
            return "slow"
        return "fast"
    return "slow"


def main():
    src = os.path.join(ROOT, "orb_slam2_comment_amd", "csrc", "orbhip_extractor.hip")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "ext.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
                        "-I", os.path.join(ROOT, "include"), "-I", os.path.dirname(src), "-S", "--cuda-device-only", "-w", src, "-o", out],
                       check=True)
        text = open(out).read()
    res = {}
    cur = None
    for line in text.split("\n"):
        t = line.strip()
        m = re.match(r"^(_ZN6orbhip\w+):", t)
        if m:
            cur = m.group(1)
            res[cur] = {"fast": 0, "slow": 0, "f64": 0}
            continue
        if t.startswith(".Lfunc_end"):
            cur = None
        if cur is None or not t or t[0] in ";.":
            continue
        c = classify(t)
        if c:
            res[cur][c] += 1
    out = {}
    for k, v in res.items():
        n = v["fast"] + v["slow"] + v["f64"]
        if n == 0:
            continue
        short = re.sub(r"^_ZN6orbhip\d+", "", k)
        short = re.match(r"[a-z_]+[a-z]", short).group(0) + ("<11>" if "ILi11ELb0" in k else "") if "fast_cells" in k else re.match(r"[a-z_0-9]+?(?=E|I)", short).group(0)
        if "fast_cells" in k and "ILi11ELb0" not in k:
            continue
        out[short] = {"static_valu": n, "fast": v["fast"], "slow": v["slow"], "f64": v["f64"],
                      "mean_issue_ns": round((v["fast"] * FAST_NS + v["slow"] * SLOW_NS + v["f64"] * F64_NS) / n, 3)}
    print(json.dumps({"_model": "static instruction mix x measured issue cost per class (fast %.2f ns, slow %.2f ns, f64 arithmetic %.2f ns per "
                                "wave-instruction per SIMD, tools/micro on MI355X)" % (FAST_NS, SLOW_NS, F64_NS), "kernels": out}, indent=1))


if __name__ == "__main__":
    sys.exit(main())
