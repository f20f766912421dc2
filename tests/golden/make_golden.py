#!/usr/bin/env python3
"""Generates tests/golden/extract_golden.npz and match_golden.npz from the CPU oracle
(oracle/orb_oracle.c) on synth_frame inputs.  These vectors pin the oracle against
regressions and let the GPU path be checked without the oracle; they are NOT outputs of the
reference (which cannot be built here: OpenCV is absent -- "parity unpinned").

usage: python tests/golden/make_golden.py
"""
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle_py as O  # noqa: E402
from orb_slam2_comment_amd.synth import synth_frame  # noqa: E402


def z(a):
    return np.frombuffer(zlib.compress(np.ascontiguousarray(a).tobytes(), 9), np.uint8)


out = {}
for seed, W, H, nf in [(1, 320, 240, 500), (2, 320, 240, 500), (3, 320, 240, 500), (1, 1241, 376, 1000),
                       (1, 752, 480, 2000)]:
    e = O.OracleExtractor(nf, 1.2, 8, 20, 7)
    k, d = e.extract(synth_frame(seed, W, H))
    key = "f_%d_%d_%d_%d" % (seed, W, H, nf)
    out[key + "_kps"] = z(k)
    out[key + "_desc"] = z(d)
    print(key, len(k))
np.savez(os.path.join(ROOT, "tests", "golden", "extract_golden.npz"), **out)
print("bytes", os.path.getsize(os.path.join(ROOT, "tests", "golden", "extract_golden.npz")))

# ---- matcher fixtures (oracle outputs on oracle-extracted features; inputs are reproducible from synth) ----
from oracle.oracle_py import make_frame  # noqa: E402
from orb_slam2_comment_amd.synth import synth_stereo  # noqa: E402

mout = {}
W, H, nf = 640, 480, 800
e = O.OracleExtractor(nf, 1.2, 8, 20, 7)
k1, d1 = e.extract(synth_frame(21, W, H))
k2, d2 = e.extract(synth_frame(21, W, H, shift_xy=(4, 0)))
sf = e.tables()["scale"]
keep = []
b = (0.0, 0.0, float(W), float(H))
f1, f2 = make_frame(k1, d1, None, b, sf, keep), make_frame(k2, d2, None, b, sf, keep)
prev = np.stack([k1["x"], k1["y"]], 1).astype(np.float32)
n, m12, pm = O.search_for_initialization(f1, f2, prev, 100, 0.9, True)
mout["init_n"] = np.int32(n); mout["init_m12"] = m12
q = np.zeros(len(k1), O.QUERY_DTYPE)
q["valid"] = 1; q["u"] = k1["x"] + 4; q["v"] = k1["y"]; q["radius"] = 15 * sf[k1["octave"]]
q["min_level"] = k1["octave"] - 1; q["max_level"] = k1["octave"] + 1; q["angle"] = k1["angle"]; q["observed"] = np.arange(len(k1)) % 3 != 0
n, assign = O.search_by_projection_frame(f2, q, d1, None, True)
mout["proj_n"] = np.int32(n); mout["proj_assign"] = assign
q2 = q.copy(); q2["radius"] = 4.0 * sf[k1["octave"]]; q2["min_level"] = k1["octave"] - 1; q2["max_level"] = k1["octave"]
n, assign = O.search_by_projection_points(f2, q2, d1, None, 0.8)
mout["points_n"] = np.int32(n); mout["points_assign"] = assign
left, right = synth_stereo(22, W, H)
oL, oR = O.OracleExtractor(nf, 1.2, 8, 20, 7), O.OracleExtractor(nf, 1.2, 8, 20, 7)
kl, dl = oL.extract(left); kr, dr = oR.extract(right)
lv_l = [np.ascontiguousarray(oL.level_padded(l))[19:-19, 19:-19] for l in range(8)]
lv_r = [np.ascontiguousarray(oR.level_padded(l))[19:-19, 19:-19] for l in range(8)]
t = oL.tables()
mbf = float(np.float32(386.1448)); mb = float(np.float32(386.1448) / np.float32(718.856))
n, ur, dp = O.compute_stereo_matches(kl, dl, kr, dr, lv_l, lv_r, t["scale"], t["inv_scale"], mbf, mb)
mout["stereo_n"] = np.int32(n); mout["stereo_ur"] = ur; mout["stereo_depth"] = dp
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "match_golden.npz"), **mout)
print("match golden:", {k: (v.shape if hasattr(v, "shape") and v.shape else int(v)) for k, v in mout.items()},
      os.path.getsize(os.path.join(ROOT, "tests", "golden", "match_golden.npz")))

# ---- vocabulary / BoW fixtures (bow_golden.npz): same frames as the matcher fixtures ----
sys.path.insert(0, os.path.join(ROOT, "tests"))
import tempfile  # noqa: E402
from helpers import make_vocabulary, write_vocabulary  # noqa: E402

bout = {}
voc = make_vocabulary(8, 3, seed=31)
with tempfile.TemporaryDirectory() as td:
    ov = O.OracleVocabulary(write_vocabulary(os.path.join(td, "voc.txt"), voc))
    r1, r2 = ov.transform(d1, 2), ov.transform(d2, 2)     # levelsup 2 of L = 3: 8 FeatureVector nodes
for key in ("word_id", "word_weight", "node_id", "bow_ids", "bow_vals"):
    bout["t1_" + key] = r1[key]
    bout["t2_" + key] = r2[key]
n, m12 = O.search_by_bow(f1, r1["node_id"], None, f2, r2["node_id"], None, 50, 0.7, True)
bout["bow_n"] = np.int32(n); bout["bow_m12"] = m12
sigma2 = (sf * sf).astype(np.float32)
F12 = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32)
n, m12 = O.search_for_triangulation(f1, r1["node_id"], None, f2, r2["node_id"], None, F12, 200.0, 150.0, sigma2, False, True)
bout["tri_n"] = np.int32(n); bout["tri_m12"] = m12
groups = [d1[i:i + 2 + (i % 9)] for i in range(0, 300, 11)]
bout["distinct"] = np.array([O.distinctive_descriptor(g) for g in groups], np.int32)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "bow_golden.npz"), **bout)
print("bow golden:", {k: (v.shape if hasattr(v, "shape") and v.shape else int(v)) for k, v in bout.items()},
      os.path.getsize(os.path.join(ROOT, "tests", "golden", "bow_golden.npz")))
