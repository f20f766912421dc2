#!/usr/bin/env python3
"""Secondary measurement (not the headline): wall time per call of the matcher entry points through
the host-pointer C ABI (PCIe-inclusive, synchronous), next to the C oracle on one host core.
configs[2] (KITTI-shape stereo pair + SearchByProjection) and configs[4] (EuRoC-shape
SearchForInitialization @2000)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orb_slam2_comment_amd as pkg  # noqa: E402
from orb_slam2_comment_amd import matcher as M  # noqa: E402
from orb_slam2_comment_amd.synth import synth_frame, synth_stereo  # noqa: E402
from oracle import oracle_py as O  # noqa: E402


def timeit(fn, n=20):
    fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t0) / n * 1e3


out = {}
# configs[2]: stereo pair
left, right = synth_stereo(1)
eL, eR = pkg.ORBextractor(1000, 1.2, 8, 20, 7), pkg.ORBextractor(1000, 1.2, 8, 20, 7)
kl, dl = eL(left)
kr, dr = eR(right)
m = pkg.ORBmatcher(0.9, True)
mbf, fx = 386.1448, 718.856
out["stereo_gpu_ms"] = timeit(lambda: m.ComputeStereoMatches(eL, eR, kl, dl, kr, dr, mbf, mbf / fx))
oL, oR = O.OracleExtractor(1000, 1.2, 8, 20, 7), O.OracleExtractor(1000, 1.2, 8, 20, 7)
oL.extract(left); oR.extract(right)
lv_l = [np.ascontiguousarray(oL.level_padded(l))[19:-19, 19:-19] for l in range(8)]
lv_r = [np.ascontiguousarray(oR.level_padded(l))[19:-19, 19:-19] for l in range(8)]
t = oL.tables()
out["stereo_cpu1_ms"] = timeit(lambda: O.compute_stereo_matches(kl, dl, kr, dr, lv_l, lv_r, t["scale"], t["inv_scale"], mbf, mbf / fx), 5)
# SearchByProjection frame t -> t+1
ext = pkg.ORBextractor(1000, 1.2, 8, 20, 7)
img0, img1 = synth_frame(4), synth_frame(4, shift_xy=(3, 0))
k0, d0 = ext(img0)
k1, d1 = ext(img1)
sf = ext.GetScaleFactors()
b = (0.0, 0.0, 1241.0, 376.0)
gv = pkg.FrameView(k1, d1, sf, b)
keep = []
ov = O.make_frame(k1, d1, None, b, sf, keep)
q = np.zeros(len(k0), pkg.QUERY_DTYPE)
q["valid"] = 1; q["u"] = k0["x"] + 3; q["v"] = k0["y"]; q["radius"] = 15 * sf[k0["octave"]]
q["min_level"] = k0["octave"] - 1; q["max_level"] = k0["octave"] + 1; q["angle"] = k0["angle"]; q["observed"] = 1
out["proj_frame_gpu_ms"] = timeit(lambda: m.SearchByProjectionFrame(gv, q, d0))
out["proj_frame_cpu1_ms"] = timeit(lambda: O.search_by_projection_frame(ov, q, d0, None, True), 3)
out["proj_frame_matches"] = int(m.SearchByProjectionFrame(gv, q, d0)[0])
# configs[4]: EuRoC init
ext2 = pkg.ORBextractor(2000, 1.2, 8, 20, 7)
a0, a1 = synth_frame(1, 752, 480), synth_frame(1, 752, 480, shift_xy=(5, 0))
ka, da = ext2(a0)
kb, db = ext2(a1)
b2 = (0.0, 0.0, 752.0, 480.0)
g1, g2 = pkg.FrameView(ka, da, sf, b2), pkg.FrameView(kb, db, sf, b2)
o1, o2 = O.make_frame(ka, da, None, b2, sf, keep), O.make_frame(kb, db, None, b2, sf, keep)
prev = np.stack([ka["x"], ka["y"]], 1).astype(np.float32)
out["init_gpu_ms"] = timeit(lambda: m.SearchForInitialization(g1, g2, prev, 100))
out["init_cpu1_ms"] = timeit(lambda: O.search_for_initialization(o1, o2, prev, 100, 0.9, True), 3)
out["init_matches"] = int(m.SearchForInitialization(g1, g2, prev, 100)[0])
# single-frame extract latency through the host API (PCIe-inclusive)
out["extract_single_gpu_ms"] = timeit(lambda: ext(img0))
frames = np.stack([synth_frame(1 + i % 8) for i in range(64)])
out["extract_batch64_host_ms"] = timeit(lambda: ext.extract_batch(frames), 5)
out["extract_batch64_host_fps"] = 64 / out["extract_batch64_host_ms"] * 1e3
# Frame::ComputeBoW + SearchByBoW (reference-key-frame tracking / relocalisation chain), ORBvoc-shaped tree k=10
from orb_slam2_comment_amd.synth import synth_vocabulary as regular_tree  # noqa: E402

from helpers import write_vocabulary  # noqa: E402
import tempfile  # noqa: E402
for L in (5, 6):
    par, leaf, vdesc, wgt = regular_tree(10, L, 1)
    voc = pkg.ORBVocabulary.from_arrays(10, L, 0, 0, par, leaf, vdesc, wgt)
    near = vdesc[leaf == 1][np.random.default_rng(2).integers(0, int(leaf.sum()), len(d0))].copy()   # descend to full depth
    near[:, 5] ^= 0x11
    out["bow_transform_L%d_gpu_ms" % L] = timeit(lambda: voc.transform(near, 4))
    if L == 6:   # levelsup 4 -> 100 FeatureVector nodes, as with ORBvoc.txt
        r0, r1 = voc.transform(d0, 4), voc.transform(d1, 4)
        gk0 = pkg.FrameView(k0, d0, sf, b)
        mb = pkg.ORBmatcher(0.7, True)
        out["search_by_bow_gpu_ms"] = timeit(lambda: mb.SearchByBoW(gk0, r0["node_id"], None, gv, r1["node_id"], None, 50))
        ok0 = O.make_frame(k0, d0, None, b, sf, keep)
        out["search_by_bow_cpu1_ms"] = timeit(lambda: O.search_by_bow(ok0, r0["node_id"], None, ov, r1["node_id"], None, 50, 0.7, True), 5)
        out["search_by_bow_matches"] = int(mb.SearchByBoW(gk0, r0["node_id"], None, gv, r1["node_id"], None, 50)[0])
        out["search_by_bow_nodes"] = int(len(np.unique(r0["node_id"])))
    if L == 5:
        with tempfile.TemporaryDirectory() as td:
            ovoc = O.OracleVocabulary(write_vocabulary(os.path.join(td, "voc.txt"), dict(
                k=10, L=L, scoring=0, weighting=0, parent=par, is_leaf=leaf, desc=vdesc, weight=wgt)))
        out["bow_transform_L5_cpu1_ms"] = timeit(lambda: ovoc.transform(near, 4), 5)
        # device-resident batch: 64 frames x cap features, timed with HIP events
        import torch
        B, cap = 64, 1280
        dev = torch.device("cuda:0")
        reps = (cap + len(near) - 1) // len(near)
        dd = torch.from_numpy(np.tile(near, (reps, 1))[:cap]).to(dev).unsqueeze(0).repeat(B, 1, 1).contiguous()
        dn = torch.full((B,), len(near) if len(near) < cap else cap, dtype=torch.int32, device=dev)
        bufs = [torch.zeros((B, cap), dtype=dt, device=dev) for dt in (torch.int32, torch.float64, torch.int32, torch.int32, torch.float64)]
        dnb = torch.zeros(B, dtype=torch.int32, device=dev)
        stream = torch.cuda.Stream()
        voc.set_stream(stream.cuda_stream)
        with torch.cuda.stream(stream):
            def run():
                voc.transform_device(B, dd.data_ptr(), dn.data_ptr(), cap, 4, bufs[0].data_ptr(), bufs[1].data_ptr(),
                                     bufs[2].data_ptr(), bufs[3].data_ptr(), bufs[4].data_ptr(), dnb.data_ptr())
            run(); stream.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(20):
                run()
            e1.record(stream); stream.synchronize()
        out["bow_transform_device_batch64_ms"] = e0.elapsed_time(e1) / 20
        out["bow_transform_device_fps"] = B / out["bow_transform_device_batch64_ms"] * 1e3
        voc.set_stream(0)
print(json.dumps({k: round(v, 3) for k, v in out.items()}))
