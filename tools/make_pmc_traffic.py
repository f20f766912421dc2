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


kf, kb, kd, ko = find("k_fast_cells"), find("k_blur"), find("k_orient_describe"), find("k_octree")
out = {
    "_source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --steps 4 --min-time 0 --handles 1 "
               "(%d frames 1241x376 per launch); bytes are per frame; FETCH_SIZE x2 (gfx950, wide coalesced reads: an upper "
               "bound for gather-shaped kernels)" % frames,
    "fast": {"kernel": kf, "hbm_bytes_per_frame": round(hbm(kf) / frames)},
    "blur": {"kernel": kb, "hbm_bytes_per_frame": round(hbm(kb) / frames)},
    "describe": {"kernel": kd, "hbm_bytes_per_frame": round(hbm(kd) / frames)},
    "pyramid": {"kernel": "k_pyr_level0 + 7 x k_pyr_resize",
                "hbm_bytes_per_frame": round((hbm(find("k_pyr_level0")) + 7 * hbm(find("k_pyr_resize"))) / frames)},
    "octree": {"kernel": ko, "hbm_bytes_per_frame": round(hbm(ko) / frames)},
}
# VALU issue load of every stage ("report VALU utilisation alongside"): one VALU instruction occupies its
# SIMD for 4 cycles (wave64 on a 16-lane SIMD); 256 CUs x 4 SIMDs, 2.4 GHz
for stage, key in (("fast", kf), ("blur", kb), ("describe", kd)):
    v = d[key]
    if "SQ_INSTS_VALU" in v:
        out[stage]["valu_insts_per_frame"] = round(v["SQ_INSTS_VALU"] / frames)
        out[stage]["valu_issue_us_per_frame"] = round(v["SQ_INSTS_VALU"] / frames * 4 / 1024 / 2.4e3, 4)
json.dump(out, open("profiles/pmc_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
