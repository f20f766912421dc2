#!/usr/bin/env python3
"""profiles/pmc_traffic.json from a tools/pmc_summarize.py summary: HBM bytes per launch of every
stage (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, KiB -> bytes), keyed like bench.py's stages."""
import json
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc_summary.json"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 64     # frames per launch in the profiled run
d = json.load(open(src))


def find(sub):
    ks = [k for k in d if sub in k]
    if not ks:
        raise KeyError(sub)
    return ks[0]


def hbm(k):
    v = d[k]
    return v.get("hbm_read_bytes_corrected", 0) + v.get("hbm_write_bytes", 0)


kf, kd, ko = find("k_fast_cells"), find("k_describe_fused"), find("k_octree")
npyr_rows = int(sys.argv[3]) if len(sys.argv) > 3 else 6   # k_pyr_rows launches per frame batch (levels 2 .. 7)
out = {
    "_source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --steps 4 --min-time 0 --handles 1 "
               "(%d frames 1241x376 per launch); bytes are per frame; FETCH_SIZE x2 (gfx950, wide coalesced reads: an upper "
               "bound for gather-shaped kernels)" % frames,
    "fast": {"kernel": kf, "hbm_bytes_per_frame": round(hbm(kf) / frames)},
    "blur": {"kernel": "(fused into k_describe_fused: no blurred plane is written)", "hbm_bytes_per_frame": 0},
    "describe": {"kernel": kd, "hbm_bytes_per_frame": round(hbm(kd) / frames)},
    "pyramid": {"kernel": "k_pyr_base + %d x k_pyr_rows" % npyr_rows,
                "hbm_bytes_per_frame": round((hbm(find("k_pyr_base")) + npyr_rows * hbm(find("k_pyr_rows"))) / frames)},
    "octree": {"kernel": ko, "hbm_bytes_per_frame": round(hbm(ko) / frames)},
}
# VALU issue load of every stage ("report VALU utilisation alongside"): PMC SQ_INSTS_VALU x the mean issue
# cost of the kernel's instruction mix (tools/valu_mix.py: static mix x the per-class costs measured by tools/micro on
# MI355X -- 1.06 ns for the fast class, 1.78 ns for the rest, 2.26 ns for f64 arithmetic, per wave-instruction per SIMD)
# / 1024 SIMDs.  The SALU stream (1.74 ns per instruction per SIMD) issues beside it and is reported for k_fast_cells.
import os
import subprocess
mix = json.loads(subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "valu_mix.py")],
                                check=True, capture_output=True, text=True).stdout)
cost = {k: v["mean_issue_ns"] for k, v in mix["kernels"].items()}
model = mix["_model"]
for stage, key, mk in (("fast", kf, "k_fast_cells<11>"), ("describe", kd, "k_describe_fused"), ("octree", ko, "k_octree")):
    v = d[key]
    if "SQ_INSTS_VALU" in v:
        out[stage]["valu_insts_per_frame"] = round(v["SQ_INSTS_VALU"] / frames)
        out[stage]["valu_issue_ns_per_inst"] = cost[mk]
        out[stage]["valu_issue_us_per_frame"] = round(v["SQ_INSTS_VALU"] / frames * cost[mk] * 1e-3 / 1024, 4)
        out[stage]["valu_issue_model"] = model
        if "SQ_INSTS_SALU" in v:
            out[stage]["salu_insts_per_frame"] = round(v["SQ_INSTS_SALU"] / frames)
            out[stage]["salu_issue_us_per_frame"] = round(v["SQ_INSTS_SALU"] / frames * 1.74e-3 / 1024, 4)
v = [d[find("k_pyr_base")], d[find("k_pyr_rows")]]
if all("SQ_INSTS_VALU" in x for x in v):
    n = v[0]["SQ_INSTS_VALU"] + npyr_rows * v[1]["SQ_INSTS_VALU"]
    out["pyramid"]["valu_insts_per_frame"] = round(n / frames)
    out["pyramid"]["valu_issue_ns_per_inst"] = cost["k_pyr_rows"]
    out["pyramid"]["valu_issue_us_per_frame"] = round(n / frames * cost["k_pyr_rows"] * 1e-3 / 1024, 4)
    out["pyramid"]["valu_issue_model"] = model
json.dump(out, open("profiles/pmc_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
