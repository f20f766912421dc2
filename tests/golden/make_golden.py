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
