#!/usr/bin/env python3
"""profiles/pmc_traffic.json from a tools/pmc_summarize.py summary: HBM bytes per launch of every
stage (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, KiB -> bytes), keyed like bench.py's stages."""
import json
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc_summary.json"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 64     # frames per launch in the profiled run
d = json.load(open(src))


def hbm(k):
    v = d[k]
    return v.get("hbm_read_bytes_corrected", 0) + v.get("hbm_write_bytes", 0)


out = {
    "_source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --steps 4 --handles 1 (%d frames 1241x376 per launch); bytes are per frame" % frames,
    "fast": {"kernel": "k_fast_cells", "hbm_bytes_per_frame": round(hbm("orbhip::k_fast_cells") / frames)},
    "blur": {"kernel": "k_blur", "hbm_bytes_per_frame": round(hbm("orbhip::k_blur") / frames)},
    "describe": {"kernel": "k_orient_describe", "hbm_bytes_per_frame": round(hbm("orbhip::k_orient_describe") / frames)},
    "pyramid": {"kernel": "k_pyr_level0 + 7 x k_pyr_resize",
                "hbm_bytes_per_frame": round((hbm("orbhip::k_pyr_level0") + 7 * hbm("orbhip::k_pyr_resize")) / frames)},
    "octree": {"kernel": "k_octree", "hbm_bytes_per_frame": round(hbm([k for k in d if "k_octree" in k][0]) / frames)},
}
json.dump(out, open("profiles/pmc_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
