"""Pinned adversarial cases for the parallel resolves (fixed seeds, hand-built frames): the situations the sequential
reference loops handle through their visiting order and that a parallel restatement gets wrong first --
  * many queries with EQUAL distances to the same slot (serial dictatorship decides by query index),
  * candidate lists longer than 64 entries (unsorted lists, stepped by a whole wavefront),
  * chains of steals in SearchForInitialization (src/ORBmatcher.cc:444-467: a later, closer query takes the slot, the
    loser is not re-matched; an equal distance does NOT steal),
  * the level-conditional ratio test of the map-point search that makes an accepted query release its slot
    (src/ORBmatcher.cc:96-103),
  * a stereo row band with far more than 64 right keypoints (src/Frame.cc:483-549).
Everything is compared with the oracle: identical assignment arrays and counts."""
import numpy as np
import pytest

from helpers import synth_frame

pytestmark = pytest.mark.gpu

W, H = 1241, 376


@pytest.fixture(scope="module")
def env(oracle):
    import orb_slam2_comment_amd as pkg
    return pkg, oracle


def _scale_factors(n=8):
    return np.cumprod(np.concatenate([[np.float32(1)], np.full(n - 1, np.float32(1.2))])).astype(np.float32)


def _keys(pkg, xy, octave, angle=None):
    k = np.zeros(len(xy), pkg.capi.KP_DTYPE)
    k["x"], k["y"] = xy[:, 0], xy[:, 1]
    k["octave"] = octave
    k["size"] = 31.0
    k["angle"] = 0.0 if angle is None else angle
    k["response"] = 20.0
    k["class_id"] = -1
    return k


def _family(rng, n, base=None, flips=(0, 1, 2, 3)):
    """n descriptors at small, heavily TIED Hamming distances from one base: the base with 0..3 of four fixed bits set."""
    if base is None:
        base = rng.integers(0, 256, 32, dtype=np.uint8)
    d = np.repeat(base[None, :], n, 0).copy()
    bits = [(3, 1), (11, 4), (19, 16), (27, 64)]
    for i in range(n):
        for j in range(int(rng.choice(flips))):
            d[i, bits[j][0]] ^= np.uint8(bits[j][1])
    return d, base


def _views(pkg, O, k, d, sf, ur=None):
    keep = []
    b = (0.0, 0.0, float(W), float(H))
    return pkg.FrameView(k, d, sf, b, ur), O.make_frame(k, d, ur, b, sf, keep), keep


def _queries(pkg, uv, radius, lmin, lmax, angle=0.0, observed=1, ur=-1.0):
    q = np.zeros(len(uv), pkg.QUERY_DTYPE)
    q["valid"] = 1
    q["u"], q["v"] = uv[:, 0], uv[:, 1]
    q["radius"] = radius
    q["min_level"], q["max_level"] = lmin, lmax
    q["ur"] = ur
    q["angle"] = angle
    q["observed"] = observed
    return q


@pytest.mark.parametrize("seed,ntrain,nq,radius", [(1, 40, 60, 30.0), (2, 150, 200, 40.0), (3, 300, 90, 60.0),
                                                    (4, 500, 700, 25.0)])
def test_frame_search_ties_long_lists_and_contention(env, seed, ntrain, nq, radius):
    """SearchByProjection(CurrentFrame, LastFrame): every query sees (nearly) the same cluster of train keypoints at tied
    distances, with lists of up to 300 candidates; observed / unobserved queries and taken slots mixed."""
    pkg, O = env
    rng = np.random.default_rng(seed)
    sf = _scale_factors()
    centre = np.array([600.0, 190.0], np.float32)
    txy = (centre + rng.uniform(-18, 18, (ntrain, 2))).astype(np.float32)
    tk = _keys(pkg, txy, rng.integers(0, 3, ntrain), angle=rng.choice([10.0, 10.0, 10.0, 200.0], ntrain).astype(np.float32))
    td, base = _family(rng, ntrain)
    # a second, far cluster so that some queries have short lists
    far = (np.array([200.0, 100.0], np.float32) + rng.uniform(-10, 10, (20, 2))).astype(np.float32)
    tk = np.concatenate([tk, _keys(pkg, far, 0)])
    td = np.concatenate([td, _family(rng, 20, base)[0]])
    gv, ov, keep = _views(pkg, O, tk, td, sf)
    quv = np.concatenate([(centre + rng.uniform(-4, 4, (nq - 10, 2))), np.array([200.0, 100.0]) + rng.uniform(-3, 3, (10, 2))]).astype(np.float32)
    q = _queries(pkg, quv, radius, 0, 2, angle=10.0, observed=(rng.random(nq) < 0.8).astype(np.int32))
    q["valid"] = rng.random(nq) < 0.95
    qd = _family(rng, nq, base, flips=(0, 0, 1, 2))[0]
    for taken in (None, (rng.random(len(tk)) < 0.1).astype(np.uint8)):
        for ori in (False, True):
            m = pkg.ORBmatcher(0.9, ori)
            n, a = m.SearchByProjectionFrame(gv, q, qd, taken)
            on, oa = O.search_by_projection_frame(ov, q, qd, taken, ori)
            assert n == on and np.array_equal(a, oa), (seed, taken is not None, ori)
    assert on > 0


@pytest.mark.parametrize("seed,ntrain,nq,radius,nnratio", [(11, 60, 80, 30.0, 0.8), (12, 200, 260, 45.0, 0.8),
                                                            (13, 320, 120, 60.0, 0.6), (14, 90, 400, 20.0, 0.9)])
def test_point_search_ratio_release_with_ties_and_long_lists(env, seed, ntrain, nq, radius, nnratio):
    """SearchByProjection(F, vpMapPoints): best and second best on the same level trigger the ratio test, so a query whose
    second candidate goes to a smaller query can turn from accepted to rejected and must release its slot."""
    pkg, O = env
    rng = np.random.default_rng(seed)
    sf = _scale_factors()
    centre = np.array([400.0, 200.0], np.float32)
    txy = (centre + rng.uniform(-20, 20, (ntrain, 2))).astype(np.float32)
    tk = _keys(pkg, txy, rng.integers(1, 3, ntrain))      # levels 1 and 2: bestLevel == bestLevel2 half of the time
    td, base = _family(rng, ntrain, flips=(0, 1, 1, 2, 3))
    gv, ov, keep = _views(pkg, O, tk, td, sf)
    quv = (centre + rng.uniform(-5, 5, (nq, 2))).astype(np.float32)
    q = _queries(pkg, quv, radius, 1, 2, observed=1)
    q["valid"] = rng.random(nq) < 0.97
    qd = _family(rng, nq, base, flips=(0, 1, 2))[0]
    for taken in (None, (rng.random(len(tk)) < 0.15).astype(np.uint8)):
        m = pkg.ORBmatcher(nnratio, True)
        n, a = m.SearchByProjectionPoints(gv, q, qd, taken)
        on, oa = O.search_by_projection_points(ov, q, qd, taken, nnratio)
        assert n == on and np.array_equal(a, oa), (seed, taken is not None)


@pytest.mark.parametrize("seed,n1,n2,window,nnratio", [(21, 120, 80, 100, 0.9), (22, 400, 300, 100, 0.9), (23, 250, 500, 160, 0.6),
                                                        (24, 64, 64, 20, 0.9)])
def test_search_for_initialization_steal_chains(env, seed, n1, n2, window, nnratio):
    """Level-0 keypoints of both frames packed into one window: every query competes for the same few F2 keypoints, later
    queries come closer and steal (strictly smaller distance only), victims stay unmatched; lists exceed 64 candidates."""
    pkg, O = env
    rng = np.random.default_rng(seed)
    sf = _scale_factors()
    centre = np.array([500.0, 180.0], np.float32)
    k1 = _keys(pkg, (centre + rng.uniform(-30, 30, (n1, 2))).astype(np.float32), np.where(rng.random(n1) < 0.9, 0, 1),
               angle=rng.choice([30.0, 30.0, 31.0, 250.0], n1).astype(np.float32))
    k2 = _keys(pkg, (centre + rng.uniform(-30, 30, (n2, 2))).astype(np.float32), np.where(rng.random(n2) < 0.9, 0, 2),
               angle=rng.choice([30.0, 31.0, 100.0], n2).astype(np.float32))
    d2, base = _family(rng, n2)
    # descending distances along the query index: query i is (mostly) closer to the cluster than query i - 1
    d1 = np.repeat(base[None, :], n1, 0).copy()
    for i in range(n1):
        for b in range(max(0, 6 - (7 * i) // n1) + int(rng.integers(0, 2))):
            d1[i, 5 + b] ^= np.uint8(1 << (b % 8))
    g1, o1, keep1 = _views(pkg, O, k1, d1, sf)
    g2, o2, keep2 = _views(pkg, O, k2, d2, sf)
    prev = np.stack([k1["x"], k1["y"]], 1).astype(np.float32)
    for ori in (True, False):
        m = pkg.ORBmatcher(nnratio, ori)
        n, m12, pm = m.SearchForInitialization(g1, g2, prev, window)
        on, om12, opm = O.search_for_initialization(o1, o2, prev, window, nnratio, ori)
        assert n == on and np.array_equal(m12, om12) and np.array_equal(pm, opm), (seed, ori)


def test_stereo_row_band_with_hundreds_of_right_keypoints(env):
    """Frame::ComputeStereoMatches on a pair whose texture sits in one 36-px band: a left keypoint's row holds several
    hundred right keypoints (vRowIndices, src/Frame.cc:483-493), far beyond one wavefront's 64 lanes."""
    pkg, O = env
    rng = np.random.default_rng(5)
    left = np.full((H, W), 118, np.uint8)
    band = synth_frame(77, W, H)[170:206]
    left[170:206] = band
    right = np.full((H, W), 118, np.uint8)
    right[170:206, :-23] = band[:, 23:]                      # disparity 23 px
    noise = rng.integers(-2, 3, (H, W))
    left = np.clip(left.astype(np.int32) + noise, 0, 255).astype(np.uint8)
    right = np.clip(right.astype(np.int32) + np.roll(noise, 5, 1), 0, 255).astype(np.uint8)
    nf = 3000
    eL, eR = pkg.ORBextractor(nf, 1.2, 8, 20, 7), pkg.ORBextractor(nf, 1.2, 8, 20, 7)
    kl, dl = eL(left)
    kr, dr = eR(right)
    rows = np.round(kr["y"]).astype(int)
    assert np.bincount(rows).max() > 20 and len(kr) > 600          # hundreds of right keypoints within one +-r row band
    mbf = np.float32(386.1448); mb = np.float32(mbf / np.float32(718.856))
    n, ur, dp = pkg.ORBmatcher().ComputeStereoMatches(eL, eR, kl, dl, kr, dr, float(mbf), float(mb))
    oL, oR = O.OracleExtractor(nf, 1.2, 8, 20, 7), O.OracleExtractor(nf, 1.2, 8, 20, 7)
    okl, odl = oL.extract(left)
    okr, odr = oR.extract(right)
    lv_l = [np.ascontiguousarray(oL.level_padded(l))[19:-19, 19:-19] for l in range(8)]
    lv_r = [np.ascontiguousarray(oR.level_padded(l))[19:-19, 19:-19] for l in range(8)]
    t = oL.tables()
    on, our, odp = O.compute_stereo_matches(okl, odl, okr, odr, lv_l, lv_r, t["scale"], t["inv_scale"], float(mbf), float(mb))
    assert n == on and n > 30
    assert np.array_equal(ur, our) and np.array_equal(dp, odp)
