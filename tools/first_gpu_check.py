"""Scratch: first on-GPU comparison of the extractor against the oracle (stage by stage)."""
import ctypes as C, sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle_py as O
from orb_slam2_comment_amd.synth import synth_frame
from orb_slam2_comment_amd import capi

L = C.CDLL(capi.LIB_PATH)
for name, res, args in capi.SYMBOLS:
    if hasattr(L, name):
        fn = getattr(L, name); fn.restype = res; fn.argtypes = args
capi._LIB = L
from orb_slam2_comment_amd.extractor import ORBextractor

for (W, H, nf) in [(1241, 376, 1000), (752, 480, 2000), (320, 240, 500)]:
    img = synth_frame(1, W, H)
    e = ORBextractor(nf, 1.2, 8, 20, 7)
    t0 = time.time(); kps, desc = e(img); t1 = time.time()
    o = O.OracleExtractor(nf, 1.2, 8, 20, 7)
    okps, odesc = o.extract(img)
    print("size", W, H, "gpu n", len(kps), "oracle n", len(okps), "time", t1 - t0)
    for l in range(8):
        gp = e.image_pyramid(l, with_border=True); op = o.level_padded(l)
        pyr_ok = gp.shape == op.shape and np.array_equal(gp, op)
        gx, gy, gs = e.level_candidates(l); ox, oy, orr = o.level_candidates(l)
        cand_ok = len(gx) == len(ox) and np.array_equal(gx, ox.astype(np.int32)) and np.array_equal(gy, oy.astype(np.int32)) and np.array_equal(gs, orr.astype(np.int32))
        ob = o.level_blurred(l); gb = e.blurred_level(l)
        blur_ok = ob is None or np.array_equal(gb, ob)
        if not pyr_ok:
            d = np.argwhere(gp != op); print("   pyr diff count", len(d), d[:5])
        if not cand_ok:
            print("   cand n", len(gx), len(ox))
        if not blur_ok:
            d = np.argwhere(gb != ob); print("   blur diff count", len(d), d[:5], gb[tuple(d[0])], ob[tuple(d[0])])
        print("  level", l, "pyr", pyr_ok, "cand", cand_ok, "blur", blur_ok)
    n = min(len(kps), len(okps))
    same_kp = len(kps) == len(okps) and all(np.array_equal(kps[f], okps[f]) for f in kps.dtype.names)
    same_desc = len(kps) == len(okps) and np.array_equal(desc, odesc)
    print("  keypoints identical:", same_kp, " descriptors identical:", same_desc)
    if not same_kp:
        for f in kps.dtype.names:
            bad = np.nonzero(kps[f][:n] != okps[f][:n])[0]
            print("    field", f, "mismatches", len(bad), bad[:5], kps[f][bad[:3]], okps[f][bad[:3]])
    if same_kp and not same_desc:
        bad = np.nonzero((desc != odesc).any(1))[0]; print("    desc mismatches", len(bad), bad[:5])
    e.set_profiling(True)
    batch = np.stack([synth_frame(s, W, H) for s in range(1, 9)])
    res = e.extract_batch(batch)
    print("  batch stage times us:", e.stage_times_us())
    k0, d0 = res[0]
    print("  batch[0]==single:", np.array_equal(d0, desc), " counts", [len(r[0]) for r in res])
