#!/usr/bin/env python3
"""Randomised parity sweep on the GPU: ORBextractor (stage by stage) and the two SearchByProjection searches against the
oracle on random image sizes, feature counts, scale factors, level counts, thresholds, shifts, `taken` / `observed`
patterns, SearchForInitialization with random windows, and Frame::ComputeStereoMatches on every third case.  Complements the fixed-seed suite in tests/ (which covers every entry point); this one hunts for rare
geometry- or data-dependent mismatches.  Exit code 1 on the first mismatch (the failing case is printed).

  python tools/stress_parity.py [--cases 60] [--seed 1]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    import orb_slam2_comment_amd as pkg
    from oracle import oracle_py as O
    from helpers import assert_kps_equal, assert_stagewise_equal, frame_bounds, synth_frame, synth_stereo
    rng = np.random.default_rng(args.seed)
    t0 = time.time()
    for case in range(args.cases):
        W = int(rng.integers(160, 1500)); H = int(rng.integers(120, 760))
        nf = int(rng.choice([150, 400, 1000, 1000, 2000, 3000]))
        scale = float(rng.choice([1.1, 1.2, 1.2, 1.2, 1.3, 1.5, 2.0]))
        nlev = int(rng.integers(2, 9))
        ini = int(rng.integers(8, 40)); mn = int(rng.integers(3, ini + 1))
        seed = int(rng.integers(1, 10000))
        sx, sy = int(rng.integers(-6, 7)), int(rng.integers(-3, 4))
        desc = "case %d: %dx%d nf %d scale %.1f levels %d th %d/%d seed %d shift (%d,%d)" % (case, W, H, nf, scale, nlev, ini, mn, seed, sx, sy)
        try:
            ext = pkg.ORBextractor(nf, scale, nlev, ini, mn)
            ora = O.OracleExtractor(nf, scale, nlev, ini, mn)
        except Exception as e:      # geometry the library refuses (level too small): both sides must refuse alike
            print(desc, "-> constructor/geometry refused:", str(e)[:80])
            continue
        lazy = bool(rng.random() < 0.5)              # mvImagePyramid[0] on demand: the accessors below must still see level 0
        ext.set_lazy_level0(lazy)
        desc += " lazy0" if lazy else ""
        img0, img1 = synth_frame(seed, W, H), synth_frame(seed, W, H, shift_xy=(sx, sy))
        dense = rng.random()
        if dense < 0.25:
            # regions where (nearly) every pixel passes the FAST pre-test -- white noise, a 2- or 3-px checkerboard -- overflow
            # the kernel's survivor list and take its row-band path; both frames get the same patch (shifted)
            yy, xx = np.mgrid[0:H, 0:W]
            per = int(rng.choice([2, 3]))
            pat = rng.integers(0, 256, (H, W + 16), dtype=np.uint8) if dense < 0.12 else \
                np.pad(((((xx // per) + (yy // per)) & 1) * int(rng.integers(90, 200)) + 30).astype(np.uint8), ((0, 0), (0, 16)), mode="wrap")
            x0, x1 = sorted(int(v) for v in rng.integers(0, W, 2)); y0, y1 = sorted(int(v) for v in rng.integers(0, H, 2))
            x1 = max(x1, min(W, x0 + 60)); y1 = max(y1, min(H, y0 + 60))
            img0 = img0.copy(); img1 = img1.copy()
            img0[y0:y1, x0:x1] = pat[y0:y1, x0:x1]
            img1[y0:y1, x0:x1] = pat[y0:y1, x0 + 3:x1 + 3]
            desc += " dense[%d:%d,%d:%d]" % (y0, y1, x0, x1)
        try:
            k0, d0 = ext(img0)
        except Exception as e:
            print(desc, "-> extraction refused:", str(e)[:100])
            continue
        ok0, od0 = ora.extract(img0)
        assert_stagewise_equal(ext, ora, nlev, desc)
        assert_kps_equal(k0, ok0, desc)
        assert np.array_equal(d0, od0), desc
        k1, d1 = ext(img1)
        ok1, od1 = ora.extract(img1)
        assert_kps_equal(k1, ok1, desc + " (second frame)")
        assert np.array_equal(d1, od1), desc
        if case % 4 == 1:
            assert_stagewise_equal(ext, ora, nlev, desc + " (second frame, replayed graph)")
        if len(k0) < 8 or len(k1) < 8:
            print(desc, "-> %d / %d keypoints, matching skipped" % (len(k0), len(k1)))
            continue
        sf = ext.GetScaleFactors()
        b = frame_bounds(img1)
        ur = np.where(rng.random(len(k1)) < 0.5, k1["x"] - rng.uniform(2, 60, len(k1)), -1).astype(np.float32) \
            if rng.random() < 0.5 else None
        keep = []
        gv = pkg.FrameView(k1, d1, sf, b, ur)
        ov = O.make_frame(k1, d1, ur, b, sf, keep)
        nq = len(k0)
        q = np.zeros(nq, pkg.QUERY_DTYPE)
        q["valid"] = rng.random(nq) < 0.9
        q["u"] = k0["x"] + sx + rng.normal(0, 1.5, nq).astype(np.float32)
        q["v"] = k0["y"] + sy + rng.normal(0, 1.5, nq).astype(np.float32)
        th = float(rng.choice([3, 7, 15, 30, 60, 120]))      # 60 / 120: candidate lists beyond 64 entries
        q["radius"] = th * sf[k0["octave"]]
        mode_levels = rng.random()
        q["min_level"] = np.where(mode_levels < 0.6, k0["octave"] - 1, np.where(mode_levels < 0.8, k0["octave"], 0))
        q["max_level"] = np.where(mode_levels < 0.6, k0["octave"] + 1, np.where(mode_levels < 0.8, -1, k0["octave"]))
        q["ur"] = q["u"] - rng.uniform(1, 50, nq).astype(np.float32)
        q["angle"] = k0["angle"]
        q["observed"] = rng.random(nq) < rng.choice([0.0, 0.7, 1.0])
        taken = (rng.random(len(k1)) < 0.05).astype(np.uint8) if rng.random() < 0.5 else None
        ori = bool(rng.random() < 0.7)
        m = pkg.ORBmatcher(0.9, ori)
        n, assign = m.SearchByProjectionFrame(gv, q, d0, taken)
        on, oassign = O.search_by_projection_frame(ov, q, d0, taken, ori)
        assert n == on and np.array_equal(assign, oassign), desc + " SearchByProjection(frame)"
        nnr = float(rng.choice([0.6, 0.8, 0.9]))
        m2 = pkg.ORBmatcher(nnr, ori)
        q2 = q.copy()
        q2["min_level"] = k0["octave"] - 1; q2["max_level"] = k0["octave"]
        n2, assign2 = m2.SearchByProjectionPoints(gv, q2, d0, taken)
        on2, oassign2 = O.search_by_projection_points(ov, q2, d0, taken, nnr)
        assert n2 == on2 and np.array_equal(assign2, oassign2), desc + " SearchByProjection(points)"
        if case % 5 == 0:
            # the batch entry on the same size: 5 frames in one call = the single-frame results
            fr = np.stack([img0, img1, synth_frame(seed + 1, W, H), img0, synth_frame(seed + 2, W, H)])
            res = ext.extract_batch(fr)
            assert_kps_equal(res[0][0], ok0, desc + " batch frame 0"); assert np.array_equal(res[0][1], od0), desc
            assert_kps_equal(res[1][0], ok1, desc + " batch frame 1"); assert np.array_equal(res[1][1], od1), desc
            assert_kps_equal(res[3][0], ok0, desc + " batch frame 3"); assert np.array_equal(res[3][1], od0), desc
            ok2, od2 = ora.extract(fr[2])
            assert_kps_equal(res[2][0], ok2, desc + " batch frame 2"); assert np.array_equal(res[2][1], od2), desc
        ni = -1
        if len(k0) <= 4096 and len(k1) <= 4096:
            # SearchForInitialization(F1, F2, vbPrevMatched, windowSize) with random window / ratio / perturbed start points
            keep1 = []
            g1 = pkg.FrameView(k0, d0, sf, frame_bounds(img0))
            o1 = O.make_frame(k0, d0, None, frame_bounds(img0), sf, keep1)
            prev = np.stack([k0["x"], k0["y"]], 1).astype(np.float32) + rng.normal(0, 2.0, (len(k0), 2)).astype(np.float32)
            win = int(rng.choice([20, 50, 100, 160])); nni = float(rng.choice([0.6, 0.9])); ori_i = bool(rng.random() < 0.7)
            mi = pkg.ORBmatcher(nni, ori_i)
            gv0 = pkg.FrameView(k1, d1, sf, b)
            keep2 = []
            ov0 = O.make_frame(k1, d1, None, b, sf, keep2)
            ni, m12, pm = mi.SearchForInitialization(g1, gv0, prev, win)
            oni, om12, opm = O.search_for_initialization(o1, ov0, prev, win, nni, ori_i)
            assert ni == oni and np.array_equal(m12, om12) and np.array_equal(pm, opm), desc + " SearchForInitialization(window %d)" % win
        ns = -1
        if case % 3 == 0 and W >= 400 and H >= 200:
            # Frame::ComputeStereoMatches on a synthetic rectified pair of the same size and extractor parameters
            left, right = synth_stereo(seed, W, H)
            eR = pkg.ORBextractor(nf, scale, nlev, ini, mn)
            kl, dl = ext(left)
            kr, dr = eR(right)
            if len(kl) >= 8 and len(kr) >= 8:
                mbf = np.float32(386.1448); mb = np.float32(mbf / np.float32(718.856))
                ns, urr, dpp = pkg.ORBmatcher().ComputeStereoMatches(ext, eR, kl, dl, kr, dr, float(mbf), float(mb))
                oL, oR = O.OracleExtractor(nf, scale, nlev, ini, mn), O.OracleExtractor(nf, scale, nlev, ini, mn)
                okl, odl = oL.extract(left)
                okr, odr = oR.extract(right)
                lv_l = [np.ascontiguousarray(oL.level_padded(l))[19:-19, 19:-19] for l in range(nlev)]
                lv_r = [np.ascontiguousarray(oR.level_padded(l))[19:-19, 19:-19] for l in range(nlev)]
                t = oL.tables()
                on3, our, odp = O.compute_stereo_matches(okl, odl, okr, odr, lv_l, lv_r, t["scale"], t["inv_scale"], float(mbf), float(mb))
                assert ns == on3 and np.array_equal(urr, our) and np.array_equal(dpp, odp), desc + " ComputeStereoMatches"
        print("%s -> ok (%d / %d keypoints, %d / %d matches, init %d, stereo %d) [%.0f s]" % (desc, len(k0), len(k1), n, n2, ni, ns, time.time() - t0), flush=True)
    print("stress parity: all cases passed")
    return 0


if __name__ == "__main__":
    sys.exit(main())
