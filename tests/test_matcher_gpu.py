"""Parity of the HIP matchers against the oracle, through the C ABI.
Stated tolerance (BASELINE north_star: "match sets within a stated tolerance"): match sets,
assignment arrays and match counts must be IDENTICAL (integer Hamming distances, same visiting
order); mvuRight / mvDepth must agree to 1e-6 relative (they are produced by the same float
operation sequence, so in practice they are bit-identical)."""
import os

import numpy as np
import pytest

from helpers import frame_bounds, synth_frame, synth_stereo

pytestmark = pytest.mark.gpu

KITTI_FX, KITTI_BF = 718.856, 386.1448     # Examples/Stereo/KITTI00-02.yaml:8,25


@pytest.fixture(scope="module")
def env(oracle):
    import orb_slam2_comment_amd as pkg
    from orb_slam2_comment_amd import matcher as M
    return pkg, M, oracle


def _views(pkg, O, img, kps, desc, sf, u_right=None):
    keep = []
    gv = pkg.FrameView(kps, desc, sf, frame_bounds(img), u_right)
    ov = O.make_frame(kps, desc, u_right, frame_bounds(img), sf, keep)
    return gv, ov, keep


def test_descriptor_distance_matrix(env):
    pkg, M, O = env
    rng = np.random.default_rng(0)
    a = rng.integers(0, 256, (300, 32), dtype=np.uint8)
    b = rng.integers(0, 256, (257, 32), dtype=np.uint8)
    b[0] = a[0]; b[1] = ~a[1]
    m = pkg.ORBmatcher()
    d = m.DescriptorDistance(a, b)
    ref = np.unpackbits(a[:, None, :] ^ b[None, :, :], axis=2).sum(2)
    assert np.array_equal(d, ref)
    assert d[0, 0] == 0 and d[1, 1] == 256
    for i in (0, 5, 17):
        assert d[i, 3] == O.descriptor_distance(a[i], b[3])


@pytest.mark.parametrize("W,H,nf,shift,window", [(752, 480, 2000, 5, 100), (640, 480, 1000, 12, 30),
                                                  (1241, 376, 2000, 3, 100)])
def test_search_for_initialization(env, W, H, nf, shift, window):
    """configs[4]: EuRoC 752x480 @2000 + SearchForInitialization(windowSize=100, nnratio 0.9)
    (src/Tracking.cc:599-600)."""
    pkg, M, O = env
    ext = pkg.ORBextractor(nf, 1.2, 8, 20, 7)
    img1, img2 = synth_frame(1, W, H), synth_frame(1, W, H, shift_xy=(shift, 0))
    k1, d1 = ext(img1)
    k2, d2 = ext(img2)
    sf = ext.GetScaleFactors()
    g1, o1, keep1 = _views(pkg, O, img1, k1, d1, sf)
    g2, o2, keep2 = _views(pkg, O, img2, k2, d2, sf)
    prev = np.stack([k1["x"], k1["y"]], 1).astype(np.float32)
    for nnratio, ori in ((0.9, True), (0.6, False)):
        m = pkg.ORBmatcher(nnratio, ori)
        n, m12, pm = m.SearchForInitialization(g1, g2, prev, window)
        on, om12, opm = O.search_for_initialization(o1, o2, prev, window, nnratio, ori)
        assert n == on and np.array_equal(m12, om12)
        assert np.array_equal(pm, opm)
        assert n > 50
        # second round from the updated vbPrevMatched (Tracking calls it once per frame)
        n2, m12b, _ = m.SearchForInitialization(g1, g2, pm, window)
        on2, om12b, _ = O.search_for_initialization(o1, o2, opm, window, nnratio, ori)
        assert n2 == on2 and np.array_equal(m12b, om12b)


def _queries_from_last_frame(M, img, k_last, sf, th, rng, bf=0.0, fwd=False, bwd=False, shift=3.0):
    fx = fy = KITTI_FX
    cx, cy = img.shape[1] / 2.0, img.shape[0] / 2.0
    z = rng.uniform(4, 40, len(k_last)).astype(np.float32)
    X = np.stack([(k_last["x"] - cx) * z / fx, (k_last["y"] - cy) * z / fy, z], 1).astype(np.float32)
    Tcw = np.eye(4, dtype=np.float32)
    Tcw[0, 3] = shift * 15.0 / fx          # ~shift px at 15 m
    valid = rng.random(len(k_last)) < 0.85
    observed = rng.random(len(k_last)) < 0.7
    q = M.project_last_frame(Tcw, (fx, fy, cx, cy), (0, 0, img.shape[1], img.shape[0]), X, k_last["octave"],
                             k_last["angle"], valid, observed, sf, th, mbf=bf, bForward=fwd, bBackward=bwd)
    return q


@pytest.mark.parametrize("th,stereo,fwd,bwd", [(15, False, False, False), (7, True, False, False),
                                               (30, False, False, False), (7, True, True, False),
                                               (7, True, False, True)])
def test_search_by_projection_frame(env, th, stereo, fwd, bwd):
    """configs[2]: SearchByProjection(CurrentFrame, LastFrame, th) on a KITTI-shape pair
    (src/Tracking.cc:880-892: th 15 mono / 7 stereo, retry with 2*th)."""
    pkg, M, O = env
    rng = np.random.default_rng(th + 2 * stereo + 4 * fwd + 8 * bwd)
    ext = pkg.ORBextractor(1000, 1.2, 8, 20, 7)
    img_last, img_cur = synth_frame(4), synth_frame(4, shift_xy=(3, 0))
    k_last, d_last = ext(img_last)
    k_cur, d_cur = ext(img_cur)
    sf = ext.GetScaleFactors()
    ur = None
    if stereo:
        ur = np.where(rng.random(len(k_cur)) < 0.6, k_cur["x"] - rng.uniform(2, 60, len(k_cur)), -1).astype(np.float32)
    gv, ov, keep = _views(pkg, O, img_cur, k_cur, d_cur, sf, ur)
    q = _queries_from_last_frame(M, img_cur, k_last, sf, th, rng, bf=KITTI_BF if stereo else 0.0, fwd=fwd, bwd=bwd)
    taken = (rng.random(len(k_cur)) < 0.05).astype(np.uint8)
    for ori in (True, False):
        m = pkg.ORBmatcher(0.9, ori)
        n, assign = m.SearchByProjectionFrame(gv, q, d_last, taken)
        on, oassign = O.search_by_projection_frame(ov, q, d_last, taken, ori)
        assert n == on and np.array_equal(assign, oassign)
        assert n > 100
    # no `taken` array, every query unobserved: later queries may overwrite earlier ones (:1428)
    q2 = q.copy(); q2["observed"] = 0
    n, assign = pkg.ORBmatcher(0.9, True).SearchByProjectionFrame(gv, q2, d_last)
    on, oassign = O.search_by_projection_frame(ov, q2, d_last, None, True)
    assert n == on and np.array_equal(assign, oassign)


@pytest.mark.parametrize("th,nnratio", [(1, 0.8), (3, 0.8), (5, 0.8), (3, 0.6)])
def test_search_by_projection_points(env, th, nnratio):
    """SearchByProjection(F, vpMapPoints, th) as called from SearchLocalPoints
    (src/Tracking.cc:1184-1191: nnratio 0.8, th 1/3/5)."""
    pkg, M, O = env
    rng = np.random.default_rng(int(th * 10 + nnratio * 100))
    ext = pkg.ORBextractor(1000, 1.2, 8, 20, 7)
    img_map, img_cur = synth_frame(6), synth_frame(6, shift_xy=(2, 1))
    k_map, d_map = ext(img_map)
    k_cur, d_cur = ext(img_cur)
    sf = ext.GetScaleFactors()
    ur = np.where(rng.random(len(k_cur)) < 0.5, k_cur["x"] - rng.uniform(2, 60, len(k_cur)), -1).astype(np.float32)
    gv, ov, keep = _views(pkg, O, img_cur, k_cur, d_cur, sf, ur)
    nq = len(k_map)
    q = np.zeros(nq, pkg.QUERY_DTYPE)
    q["valid"] = rng.random(nq) < 0.9                       # mbTrackInView && !isBad()
    q["u"] = k_map["x"] + 2 + rng.normal(0, 1.0, nq).astype(np.float32)
    q["v"] = k_map["y"] + 1 + rng.normal(0, 1.0, nq).astype(np.float32)
    pred = np.clip(k_map["octave"] + rng.integers(-1, 2, nq), 0, 7)
    view_cos = rng.uniform(0.99, 1.0, nq)
    r = np.array([M.RadiusByViewingCos(c) for c in view_cos], np.float32)
    if th != 1.0:
        r = r * np.float32(th)
    q["radius"] = r * sf[pred]
    q["min_level"], q["max_level"] = pred - 1, pred
    q["ur"] = q["u"] - rng.uniform(2, 60, nq).astype(np.float32)
    q["level_aux"] = pred
    q["observed"] = rng.random(nq) < 0.8
    taken = (rng.random(len(k_cur)) < 0.1).astype(np.uint8)
    m = pkg.ORBmatcher(nnratio, True)
    n, assign = m.SearchByProjectionPoints(gv, q, d_map, taken)
    on, oassign = O.search_by_projection_points(ov, q, d_map, taken, nnratio)
    assert n == on and np.array_equal(assign, oassign)
    assert n > 50


def test_matcher_degenerate_inputs(env):
    pkg, M, O = env
    ext = pkg.ORBextractor(500, 1.2, 8, 20, 7)
    img = synth_frame(2, 640, 480)
    k, d = ext(img)
    sf = ext.GetScaleFactors()
    gv, ov, keep = _views(pkg, O, img, k, d, sf)
    m = pkg.ORBmatcher(0.9, True)
    # no queries
    n, a = m.SearchByProjectionFrame(gv, np.zeros(0, pkg.QUERY_DTYPE), np.zeros((0, 32), np.uint8))
    assert n == 0 and np.all(a == -1)
    # all queries invalid / far outside the image
    q = np.zeros(10, pkg.QUERY_DTYPE)
    q["valid"] = 1; q["u"] = 5000; q["v"] = 5000; q["radius"] = 10; q["min_level"] = -1; q["max_level"] = -1
    n, a = m.SearchByProjectionFrame(gv, q, d[:10])
    on, oa = O.search_by_projection_frame(ov, q, d[:10], None, True)
    assert n == on == 0 and np.array_equal(a, oa)
    # a query sitting on every keypoint with its own descriptor: everything matches at distance 0
    q = np.zeros(len(k), pkg.QUERY_DTYPE)
    q["valid"] = 1; q["u"] = k["x"]; q["v"] = k["y"]; q["radius"] = 3; q["min_level"] = -1; q["max_level"] = -1
    q["angle"] = k["angle"]; q["observed"] = 1
    n, a = m.SearchByProjectionFrame(gv, q, d)
    on, oa = O.search_by_projection_frame(ov, q, d, None, True)
    assert n == on and np.array_equal(a, oa) and n > 0.9 * len(k)
    # huge window: more than 64 candidates per query exercises the unsorted path
    q["radius"] = 400
    q["observed"] = np.arange(len(k)) % 2
    n, a = m.SearchByProjectionFrame(gv, q[:200], d[:200])
    on, oa = O.search_by_projection_frame(ov, q[:200], d[:200], None, True)
    assert n == on and np.array_equal(a, oa)
    n, a = m.SearchByProjectionPoints(gv, q[:200], d[:200])
    on, oa = O.search_by_projection_points(ov, q[:200], d[:200], None, 0.9)
    assert n == on and np.array_equal(a, oa)


@pytest.mark.parametrize("seed,W,H,nf", [(1, 1241, 376, 1000), (2, 1241, 376, 2000), (3, 752, 480, 1000)])
def test_compute_stereo_matches(env, seed, W, H, nf):
    """configs[2]: stereo_kitti-shape pair, Frame::ComputeStereoMatches (src/Frame.cc:466-640);
    mb = mbf/fx as intended by the reference (src/Frame.cc:114)."""
    pkg, M, O = env
    left, right = synth_stereo(seed, W, H)
    eL = pkg.ORBextractor(nf, 1.2, 8, 20, 7)
    eR = pkg.ORBextractor(nf, 1.2, 8, 20, 7)
    kl, dl = eL(left)
    kr, dr = eR(right)
    mbf = np.float32(KITTI_BF)
    mb = np.float32(mbf / np.float32(KITTI_FX))
    m = pkg.ORBmatcher()
    n, ur, dp = m.ComputeStereoMatches(eL, eR, kl, dl, kr, dr, float(mbf), float(mb))
    oL, oR = O.OracleExtractor(nf, 1.2, 8, 20, 7), O.OracleExtractor(nf, 1.2, 8, 20, 7)
    okl, odl = oL.extract(left)
    okr, odr = oR.extract(right)
    lv_l = [np.ascontiguousarray(oL.level_padded(l))[19:-19, 19:-19] for l in range(8)]
    lv_r = [np.ascontiguousarray(oR.level_padded(l))[19:-19, 19:-19] for l in range(8)]
    t = oL.tables()
    on, our, odp = O.compute_stereo_matches(okl, odl, okr, odr, lv_l, lv_r, t["scale"], t["inv_scale"], float(mbf), float(mb))
    assert n == on and n > 100
    assert np.array_equal(ur > 0, our > 0) and np.array_equal(dp > 0, odp > 0)
    assert np.allclose(ur, our, rtol=1e-6, atol=0) and np.allclose(dp, odp, rtol=1e-6, atol=0)
    assert np.array_equal(ur, our) and np.array_equal(dp, odp)     # in practice bit-identical
    # disparity sanity: uL - uR within the scene's [2, 80] px layers (+- sub-pixel refinement)
    disp = kl["x"][ur > 0] - ur[ur > 0]
    assert np.all(disp >= 0) and np.median(disp) > 2 and np.percentile(disp, 90) < 90   # a few false matches reach maxD


def test_device_batched_search_by_projection_equals_host_api(env):
    """orbhip_search_by_projection_{frame,points}_device over 4 frame pairs straight from the extractor's
    device outputs == the host-pointer entry points pair by pair (which are checked against the oracle)."""
    import torch
    pkg, M, O = env
    dev = torch.device("cuda:0")
    B, H, W = 8, 376, 1241
    frames = np.stack([synth_frame(30 + b // 2, W, H, shift_xy=(3 * (b % 2), b % 2)) for b in range(B)])
    ext = pkg.ORBextractor(1000, 1.2, 8, 20, 7)
    cap = ext.capacity(H, W)
    d_img = torch.from_numpy(frames).to(dev)
    d_kps = torch.zeros((B, cap, 7), dtype=torch.int32, device=dev)
    d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    d_n = torch.zeros(B, dtype=torch.int32, device=dev)
    ext.extract_batch_device(d_img.data_ptr(), B, H, W, d_kps.data_ptr(), d_desc.data_ptr(), cap, d_n.data_ptr())
    ext.sync()
    kps = d_kps.cpu().numpy().view(np.uint8).reshape(B, cap, 28).copy().view(pkg.KP_DTYPE).reshape(B, cap)
    desc = d_desc.cpu().numpy()
    n = d_n.cpu().numpy()
    sf = ext.GetScaleFactors()
    pairs = B // 2
    rng = np.random.default_rng(5)
    q = np.zeros((pairs, cap), pkg.QUERY_DTYPE)
    qd = np.zeros((pairs, cap, 32), np.uint8)
    nq = np.zeros(pairs, np.int32)
    taken = (rng.random((pairs, cap)) < 0.05).astype(np.uint8)
    ur = np.where(rng.random((pairs, cap)) < 0.5, rng.uniform(5, 1200, (pairs, cap)), -1).astype(np.float32)
    for p in range(pairs):
        k0 = kps[2 * p, :n[2 * p]]
        m0 = len(k0)
        nq[p] = m0
        q[p, :m0]["valid"] = rng.random(m0) < 0.9
        q[p, :m0]["u"] = k0["x"] + 3; q[p, :m0]["v"] = k0["y"] + 1
        q[p, :m0]["radius"] = 15 * sf[k0["octave"]]
        q[p, :m0]["min_level"] = k0["octave"] - 1; q[p, :m0]["max_level"] = k0["octave"] + 1
        q[p, :m0]["ur"] = k0["x"] - 20; q[p, :m0]["angle"] = k0["angle"]
        q[p, :m0]["observed"] = rng.random(m0) < 0.7
        qd[p, :m0] = desc[2 * p, :m0]
    # train side = odd frames: gather them into contiguous [pairs, cap] device tensors
    t_kps = d_kps[1::2].contiguous(); t_desc = d_desc[1::2].contiguous(); t_n = d_n[1::2].contiguous()
    t_q = torch.from_numpy(q.view(np.uint8).reshape(pairs, cap, 40).copy()).to(dev)
    t_qd = torch.from_numpy(qd).to(dev); t_nq = torch.from_numpy(nq).to(dev)
    t_taken = torch.from_numpy(taken).to(dev); t_ur = torch.from_numpy(ur).to(dev)
    t_assign = torch.zeros((pairs, cap), dtype=torch.int32, device=dev)
    t_nm = torch.zeros(pairs, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    bounds = (0.0, 0.0, float(W), float(H))
    for mode in ("frame", "points"):
        m = pkg.ORBmatcher(0.8, True)
        fn = m.SearchByProjectionFrameDevice if mode == "frame" else m.SearchByProjectionPointsDevice
        fn(pairs, t_kps.data_ptr(), t_desc.data_ptr(), t_n.data_ptr(), cap, bounds, t_q.data_ptr(), t_qd.data_ptr(),
           t_nq.data_ptr(), cap, t_assign.data_ptr(), t_nm.data_ptr(), d_u_right=t_ur.data_ptr(), d_taken=t_taken.data_ptr())
        m.sync()
        assign = t_assign.cpu().numpy(); nm = t_nm.cpu().numpy()
        for p in range(pairs):
            n1 = int(n[2 * p + 1])
            view = pkg.FrameView(kps[2 * p + 1, :n1], desc[2 * p + 1, :n1], sf, bounds, ur[p, :n1])
            host = m.SearchByProjectionFrame if mode == "frame" else m.SearchByProjectionPoints
            hn, hassign = host(view, q[p, :nq[p]], qd[p, :nq[p]], taken[p, :n1])
            assert hn == nm[p] and np.array_equal(hassign, assign[p, :n1]), (mode, p)
            assert hn > 100


def _big_frame(pkg, O, n, seed, W=1400, H=1000):
    """n keypoints spread over a large image with descriptors in loose clusters (so that windows hold real candidates)."""
    rng = np.random.default_rng(seed)
    k = np.zeros(n, pkg.KP_DTYPE)
    k["x"] = rng.uniform(20, W - 20, n).astype(np.float32)
    k["y"] = rng.uniform(20, H - 20, n).astype(np.float32)
    k["octave"] = rng.integers(0, 8, n)
    k["angle"] = rng.uniform(0, 360, n).astype(np.float32)
    proto = rng.integers(0, 256, (40, 32), dtype=np.uint8)
    d = proto[rng.integers(0, 40, n)] ^ (rng.random((n, 32)) < 0.02).astype(np.uint8) * rng.integers(1, 256, (n, 32), dtype=np.uint8)
    sf = np.cumprod(np.concatenate([[np.float32(1)], np.full(7, np.float32(1.2))])).astype(np.float32)
    keep = []
    gv = pkg.FrameView(k, d, sf, (0, 0, W, H))
    ov = O.make_frame(k, d, None, (0, 0, W, H), sf, keep)
    return k, d, sf, gv, ov, keep, rng


@pytest.mark.parametrize("n_train,nq,valid_frac", [(1500, 8000, 0.2), (1500, 6000, 0.95), (5000, 3000, 0.9), (4500, 5200, 1.0)])
def test_projection_searches_have_no_size_limit(env, n_train, nq, valid_frac):
    """Local maps (Tracking::SearchLocalPoints) and loop-closing point sets exceed 4096 entries, most of them not in
    view: invalid queries are dropped on the host, and beyond 4096 train keypoints / valid queries the resolve state
    moves from LDS to HBM.  Same results as the oracle in every regime (reference: no limit)."""
    pkg, M, O = env
    k, d, sf, gv, ov, keep, rng = _big_frame(pkg, O, n_train, n_train + nq)
    q = np.zeros(nq, pkg.QUERY_DTYPE)
    src = rng.integers(0, n_train, nq)
    q["valid"] = rng.random(nq) < valid_frac
    q["u"] = k["x"][src] + rng.normal(0, 3, nq).astype(np.float32)
    q["v"] = k["y"][src] + rng.normal(0, 3, nq).astype(np.float32)
    q["radius"] = rng.uniform(8, 25, nq).astype(np.float32)
    q["min_level"] = k["octave"][src] - 1
    q["max_level"] = k["octave"][src] + 1
    q["angle"] = (k["angle"][src] + rng.normal(0, 4, nq)).astype(np.float32) % np.float32(360)
    q["observed"] = rng.random(nq) < 0.7
    qd = d[src] ^ (rng.random((nq, 32)) < 0.03).astype(np.uint8) * rng.integers(1, 256, (nq, 32), dtype=np.uint8)
    taken = (rng.random(n_train) < 0.05).astype(np.uint8)
    m = pkg.ORBmatcher(0.8, True)
    n, assign = m.SearchByProjectionFrame(gv, q, qd, taken)
    on, oassign = O.search_by_projection_frame(ov, q, qd, taken, True)
    assert n == on and np.array_equal(assign, oassign)
    assert n > 300
    q2 = q.copy(); q2["max_level"] = q2["min_level"] + 1                        # points overload: levels [pred-1, pred]
    n, assign = m.SearchByProjectionPoints(gv, q2, qd, taken)
    on, oassign = O.search_by_projection_points(ov, q2, qd, taken, 0.8)
    assert n == on and np.array_equal(assign, oassign)
    n, assign = m.SearchByProjectionKeyFrame(gv, q, qd, taken, ORBdist=64)
    on, oassign = O.search_by_projection_block(ov, q, qd, taken, 64, True)
    assert n == on and np.array_equal(assign, oassign)


def test_search_for_initialization_keeps_its_limit(env):
    """SearchForInitialization replays the match stealing over LDS state: more than 4096 keypoints is an explicit
    E_CAPACITY (the initialiser extracts 2 x nFeatures = 2000, src/Tracking.cc:125)."""
    pkg, M, O = env
    n = 4100
    k = np.zeros(n, pkg.KP_DTYPE); k["x"] = np.arange(n) % 600 + 20; k["y"] = np.arange(n) // 600 * 5 + 20
    d = np.random.default_rng(0).integers(0, 256, (n, 32), dtype=np.uint8)
    fv = pkg.FrameView(k, d, np.ones(8, np.float32), (0, 0, 640, 480))
    with pytest.raises(pkg.OrbHipError) as ei:
        pkg.ORBmatcher(0.9, True).SearchForInitialization(fv, fv, np.stack([k["x"], k["y"]], 1), 100)
    assert ei.value.code == -3


@pytest.mark.parametrize("th,orb_dist,ori", [(10, 100, True), (3, 64, True), (10, 100, False)])
def test_search_by_projection_keyframe_and_sim3(env, th, orb_dist, ori):
    """SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist) (src/Tracking.cc:1452,1466: th 10/3,
    ORBdist 100/64) and SearchByProjection(pKF, Scw, vpPoints, vpMatched, th) (src/LoopClosing.cc:239,589)."""
    pkg, M, O = env
    rng = np.random.default_rng(th * 1000 + orb_dist + ori)
    ext = pkg.ORBextractor(1000, 1.2, 8, 20, 7)
    img_kf, img_cur = synth_frame(8), synth_frame(8, shift_xy=(2, 1))
    k_kf, d_kf = ext(img_kf)
    k_cur, d_cur = ext(img_cur)
    sf = ext.GetScaleFactors()
    gv, ov, keep = _views(pkg, O, img_cur, k_cur, d_cur, sf)
    nq = len(k_kf)
    q = np.zeros(nq, pkg.QUERY_DTYPE)
    q["valid"] = rng.random(nq) < 0.9
    q["u"] = k_kf["x"] + 2 + rng.normal(0, 1.5, nq).astype(np.float32)
    q["v"] = k_kf["y"] + 1 + rng.normal(0, 1.5, nq).astype(np.float32)
    pred = np.clip(k_kf["octave"] + rng.integers(-1, 2, nq), 0, 7)
    q["radius"] = np.float32(th) * sf[pred]
    q["min_level"], q["max_level"] = pred - 1, pred + 1
    q["angle"] = k_kf["angle"]
    taken = (rng.random(len(k_cur)) < 0.15).astype(np.uint8)
    m = pkg.ORBmatcher(0.9, ori)
    n, a = m.SearchByProjectionKeyFrame(gv, q, d_kf, taken, ORBdist=orb_dist)
    on, oa = O.search_by_projection_block(ov, q, d_kf, taken, orb_dist, ori)
    assert n == on and np.array_equal(a, oa) and n > 50
    # loop-closing variant: levels [pred-1, pred], TH_LOW, no orientation check
    q["min_level"], q["max_level"] = pred - 1, pred
    n, a = m.SearchByProjectionSim3(gv, q, d_kf, taken)
    on, oa = O.search_by_projection_block(ov, q, d_kf, taken, 50, False)
    assert n == on and np.array_equal(a, oa) and n > 50


@pytest.mark.parametrize("th,stereo,gate", [(3.0, True, True), (3.0, False, True), (7.5, False, False), (4.0, True, False)])
def test_search_best_in_window_fuse_and_sim3(env, th, stereo, gate):
    """Search loop of ORBmatcher::Fuse (th 3.0 in LocalMapping, chi2 gate 5.99/7.8) and of SearchBySim3 (th 7.5,
    no gate): independent best match per query."""
    pkg, M, O = env
    rng = np.random.default_rng(int(th * 10) + 2 * stereo + gate)
    ext = pkg.ORBextractor(1000, 1.2, 8, 20, 7)
    img_a, img_b = synth_frame(12), synth_frame(12, shift_xy=(1, 1))
    ka, da = ext(img_a)
    kb, db = ext(img_b)
    sf = ext.GetScaleFactors()
    inv_sig2 = ext.GetInverseScaleSigmaSquares()
    ur = np.where(rng.random(len(kb)) < 0.6, kb["x"] - rng.uniform(2, 60, len(kb)), -1).astype(np.float32) if stereo else None
    gv, ov, keep = _views(pkg, O, img_b, kb, db, sf, ur)
    nq = len(ka)
    q = np.zeros(nq, pkg.QUERY_DTYPE)
    q["valid"] = rng.random(nq) < 0.9
    q["u"] = ka["x"] + 1 + rng.normal(0, 1.0, nq).astype(np.float32)
    q["v"] = ka["y"] + 1 + rng.normal(0, 1.0, nq).astype(np.float32)
    pred = np.clip(ka["octave"] + rng.integers(0, 2, nq), 0, 7)
    q["radius"] = np.float32(th) * sf[pred]
    q["min_level"], q["max_level"] = pred - 1, pred
    q["ur"] = q["u"] - rng.uniform(2, 60, nq).astype(np.float32)
    m = pkg.ORBmatcher()
    bi, bd = m.SearchBestInWindow(gv, q, da, inv_sig2 if gate else None)
    obi, obd = O.search_best_in_window(ov, q, da, inv_sig2 if gate else None)
    assert np.array_equal(bi, obi) and np.array_equal(bd, obd)
    assert (bd <= 50).sum() > 50


def test_device_batched_stereo_equals_host_api(env):
    """orbhip_compute_stereo_matches_device on an interleaved (L0,R0,L1,R1,...) batch held by ONE extractor
    handle == the host entry point pair by pair."""
    import torch
    pkg, M, O = env
    dev = torch.device("cuda:0")
    P, H, W = 3, 376, 1241
    frames = []
    for p in range(P):
        l, r = synth_stereo(60 + p, W, H)
        frames += [l, r]
    frames = np.stack(frames)
    B = 2 * P
    ext = pkg.ORBextractor(1000, 1.2, 8, 20, 7)
    cap = ext.capacity(H, W)
    d_img = torch.from_numpy(frames).to(dev)
    d_kps = torch.zeros((B, cap, 7), dtype=torch.int32, device=dev)
    d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    d_n = torch.zeros(B, dtype=torch.int32, device=dev)
    ext.extract_batch_device(d_img.data_ptr(), B, H, W, d_kps.data_ptr(), d_desc.data_ptr(), cap, d_n.data_ptr())
    ext.sync()
    d_ur = torch.full((P, cap), -7.0, dtype=torch.float32, device=dev)
    d_dp = torch.full((P, cap), -7.0, dtype=torch.float32, device=dev)
    d_nm = torch.zeros(P, dtype=torch.int32, device=dev)
    mbf = float(np.float32(KITTI_BF)); mb = float(np.float32(KITTI_BF) / np.float32(KITTI_FX))
    m = pkg.ORBmatcher()
    m.ComputeStereoMatchesDevice(ext, 0, 2, ext, 1, 2, P, d_kps.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(),
                                 d_kps.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), cap, mbf, mb, d_ur.data_ptr(),
                                 d_dp.data_ptr(), d_nm.data_ptr())
    m.sync()
    kps = d_kps.cpu().numpy().view(np.uint8).reshape(B, cap, 28).copy().view(pkg.KP_DTYPE).reshape(B, cap)
    desc = d_desc.cpu().numpy(); n = d_n.cpu().numpy()
    ur, dp, nm = d_ur.cpu().numpy(), d_dp.cpu().numpy(), d_nm.cpu().numpy()
    for p in range(P):
        nl, nr = int(n[2 * p]), int(n[2 * p + 1])
        hn, hur, hdp = m.ComputeStereoMatches(ext, ext, kps[2 * p, :nl], desc[2 * p, :nl], kps[2 * p + 1, :nr],
                                              desc[2 * p + 1, :nr], mbf, mb, frame_l=2 * p, frame_r=2 * p + 1)
        assert hn == nm[p] and hn > 100
        assert np.array_equal(ur[p, :nl], hur) and np.array_equal(dp[p, :nl], hdp)
        assert np.all(ur[p, nl:] == -7.0)


def test_matcher_golden_fixtures(env):
    """Committed oracle outputs (tests/golden/make_golden.py): the GPU path must reproduce them without the oracle."""
    import os
    pkg, M, O = env
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "match_golden.npz"))
    W, H, nf = 640, 480, 800
    ext = pkg.ORBextractor(nf, 1.2, 8, 20, 7)
    k1, d1 = ext(synth_frame(21, W, H))
    k2, d2 = ext(synth_frame(21, W, H, shift_xy=(4, 0)))
    sf = ext.GetScaleFactors()
    b = (0.0, 0.0, float(W), float(H))
    F1, F2 = pkg.FrameView(k1, d1, sf, b), pkg.FrameView(k2, d2, sf, b)
    prev = np.stack([k1["x"], k1["y"]], 1).astype(np.float32)
    n, m12, _ = pkg.ORBmatcher(0.9, True).SearchForInitialization(F1, F2, prev, 100)
    assert n == int(g["init_n"]) and np.array_equal(m12, g["init_m12"])
    q = np.zeros(len(k1), pkg.QUERY_DTYPE)
    q["valid"] = 1; q["u"] = k1["x"] + 4; q["v"] = k1["y"]; q["radius"] = 15 * sf[k1["octave"]]
    q["min_level"] = k1["octave"] - 1; q["max_level"] = k1["octave"] + 1; q["angle"] = k1["angle"]
    q["observed"] = np.arange(len(k1)) % 3 != 0
    n, a = pkg.ORBmatcher(0.9, True).SearchByProjectionFrame(F2, q, d1)
    assert n == int(g["proj_n"]) and np.array_equal(a, g["proj_assign"])
    q["radius"] = 4.0 * sf[k1["octave"]]; q["max_level"] = k1["octave"]
    n, a = pkg.ORBmatcher(0.8, True).SearchByProjectionPoints(F2, q, d1)
    assert n == int(g["points_n"]) and np.array_equal(a, g["points_assign"])
    left, right = synth_stereo(22, W, H)
    eL, eR = pkg.ORBextractor(nf, 1.2, 8, 20, 7), pkg.ORBextractor(nf, 1.2, 8, 20, 7)
    kl, dl = eL(left); kr, dr = eR(right)
    mbf = float(np.float32(386.1448)); mb = float(np.float32(386.1448) / np.float32(718.856))
    n, ur, dp = pkg.ORBmatcher().ComputeStereoMatches(eL, eR, kl, dl, kr, dr, mbf, mb)
    assert n == int(g["stereo_n"]) and np.array_equal(ur, g["stereo_ur"]) and np.array_equal(dp, g["stereo_depth"])


# ---- vocabulary-guided searches (SearchByBoW x2, SearchForTriangulation) ---------------------------------
def _pseudo_nodes(desc, n_nodes, rng, p_absent=0.03):
    """Stand-in for the DBoW2 FeatureVector node of each descriptor (the vocabulary file is not part of the path):
    a coarse hash of the leading descriptor bits, so that similar descriptors mostly share a node."""
    node = (desc[:, 0].astype(np.uint32) * 7 + (desc[:, 1].astype(np.uint32) >> 6)) % np.uint32(n_nodes)
    node = node * np.uint32(37) + np.uint32(11)          # sparse, non-contiguous ids
    node[rng.random(len(node)) < p_absent] = 0xFFFFFFFF
    return node


@pytest.mark.parametrize("W,H,nf,n_nodes", [(1241, 376, 2000, 100), (752, 480, 1000, 100), (640, 480, 3000, 7)])
def test_search_by_bow(env, W, H, nf, n_nodes):
    """SearchByBoW(KeyFrame*,Frame&) (Tracking::TrackReferenceKeyFrame nnratio 0.7, Relocalization 0.75) and
    SearchByBoW(KeyFrame*,KeyFrame*) (LoopClosing 0.75): assignments identical to the oracle."""
    pkg, M, O = env
    rng = np.random.default_rng(W + nf)
    ext = pkg.ORBextractor(nf, 1.2, 8, 20, 7)
    img1, img2 = synth_frame(5, W, H), synth_frame(5, W, H, shift_xy=(4, 1))
    k1, d1 = ext(img1)
    k2, d2 = ext(img2)
    sf = ext.GetScaleFactors()
    g1, o1, keep1 = _views(pkg, O, img1, k1, d1, sf)
    g2, o2, keep2 = _views(pkg, O, img2, k2, d2, sf)
    node1, node2 = _pseudo_nodes(d1, n_nodes, rng), _pseudo_nodes(d2, n_nodes, rng)
    valid1 = (rng.random(len(k1)) < 0.8).astype(np.uint8)
    blocked2 = (rng.random(len(k2)) < 0.2).astype(np.uint8)
    total = 0
    for nnratio, ori, max_dist, v1, b2 in ((0.7, True, 50, valid1, None), (0.75, True, 49, valid1, blocked2),
                                           (0.9, False, 50, None, None), (0.75, False, 100, None, blocked2)):
        m = pkg.ORBmatcher(nnratio, ori)
        n, m12 = m.SearchByBoW(g1, node1, v1, g2, node2, b2, max_dist)
        on, om12 = O.search_by_bow(o1, node1, v1, o2, node2, b2, max_dist, nnratio, ori)
        assert n == on and np.array_equal(m12, om12)
        assert n == int((m12 >= 0).sum())
        hit = m12[m12 >= 0]
        assert len(np.unique(hit)) == len(hit)                       # vbMatched2 / vpMapPointMatches exclusivity
        if b2 is not None:
            assert not b2[hit].any()
        if v1 is not None:
            assert v1[m12 >= 0].all()
        assert np.array_equal(node1[m12 >= 0], node2[hit])
        total += n
    assert total > 100


def test_search_by_bow_random_collisions(env):
    """Random descriptors pressed into few nodes with many exact duplicates: long unsorted candidate lists (> 64),
    distance ties and chains of blocked slots."""
    pkg, M, O = env
    rng = np.random.default_rng(77)
    sf = np.float32(1.2) ** np.arange(8, dtype=np.float32)
    for n1, n2, n_nodes in ((900, 1100, 3), (4096, 4096, 40), (1, 1, 1), (300, 2, 2), (6000, 9000, 11)):
        base = rng.integers(0, 256, (40, 32), dtype=np.uint8)
        def noisy(n):
            d = base[rng.integers(0, len(base), n)].copy()
            flips = rng.integers(0, 256, (n, 32), dtype=np.uint8) & rng.integers(0, 256, (n, 32), dtype=np.uint8) \
                & rng.integers(0, 256, (n, 32), dtype=np.uint8) & rng.integers(0, 256, (n, 32), dtype=np.uint8)
            d ^= flips * (rng.random((n, 1)) < 0.7)
            return d
        d1, d2 = noisy(n1), noisy(n2)
        def keys(n):
            k = np.zeros(n, pkg.KP_DTYPE)
            k["x"] = rng.uniform(0, 640, n); k["y"] = rng.uniform(0, 480, n)
            k["angle"] = rng.uniform(0, 360, n).astype(np.float32); k["octave"] = rng.integers(0, 8, n)
            return k
        k1, k2 = keys(n1), keys(n2)
        img = np.zeros((480, 640), np.uint8)
        g1, o1, keep1 = _views(pkg, O, img, k1, d1, sf)
        g2, o2, keep2 = _views(pkg, O, img, k2, d2, sf)
        node1 = rng.integers(0, n_nodes, n1).astype(np.uint32)
        node2 = rng.integers(0, n_nodes, n2).astype(np.uint32)
        for nnratio, ori, max_dist in ((0.75, True, 50), (1.5, False, 100)):
            m = pkg.ORBmatcher(nnratio, ori)
            n, m12 = m.SearchByBoW(g1, node1, None, g2, node2, None, max_dist)
            on, om12 = O.search_by_bow(o1, node1, None, o2, node2, None, max_dist, nnratio, ori)
            assert n == on and np.array_equal(m12, om12)
    # SearchForTriangulation keeps per-query state in LDS: capacity error instead of silent truncation
    big = np.zeros(4097, pkg.KP_DTYPE)
    gb = pkg.FrameView(big, np.zeros((4097, 32), np.uint8), sf, (0, 0, 640, 480))
    with pytest.raises(pkg.OrbHipError):
        pkg.ORBmatcher().SearchForTriangulation(gb, np.zeros(4097, np.uint32), None, gb, np.zeros(4097, np.uint32), None,
                                                np.eye(3, dtype=np.float32), (0, 0), sf * sf)


@pytest.mark.parametrize("W,H,nf,only_stereo", [(1241, 376, 2000, False), (752, 480, 1500, False), (1241, 376, 2000, True)])
def test_search_for_triangulation(env, W, H, nf, only_stereo):
    """LocalMapping::CreateNewMapPoints -> SearchForTriangulation(pKF1, pKF2, F12, pairs, false), nnratio 0.6."""
    pkg, M, O = env
    rng = np.random.default_rng(nf + W)
    ext = pkg.ORBextractor(nf, 1.2, 8, 20, 7)
    img1, img2 = synth_frame(9, W, H), synth_frame(9, W, H, shift_xy=(6, 0))
    k1, d1 = ext(img1)
    k2, d2 = ext(img2)
    sf = ext.GetScaleFactors()
    sigma2 = ext.GetScaleSigmaSquares()
    ur1 = np.where(rng.random(len(k1)) < 0.5, k1["x"] - 10, -1).astype(np.float32)
    ur2 = np.where(rng.random(len(k2)) < 0.5, k2["x"] - 10, -1).astype(np.float32)
    g1, o1, keep1 = _views(pkg, O, img1, k1, d1, sf, ur1)
    g2, o2, keep2 = _views(pkg, O, img2, k2, d2, sf, ur2)
    node1, node2 = _pseudo_nodes(d1, 100, rng), _pseudo_nodes(d2, 100, rng)
    valid1 = (rng.random(len(k1)) < 0.7).astype(np.uint8)
    valid2 = (rng.random(len(k2)) < 0.7).astype(np.uint8)
    # sideways motion: epipolar line of (x1, y1) is the row y2 = y1; epipole placed inside the image so that the
    # "too close to the epipole" gate (:741-747) rejects part of the monocular candidates
    F_rows = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32)
    th = np.float32(0.01)
    F_tilt = (F_rows + np.array([[0, th, 0], [-th, 0, 0], [0, 0, 0]], np.float32) * np.float32(0.01)).astype(np.float32)
    total = 0
    for F12, epi, ori in ((F_rows, (W / 2.0, H / 2.0), True), (F_tilt, (W / 3.0, H / 1.5), False),
                          (np.zeros((3, 3), np.float32), (0.0, 0.0), True)):
        m = pkg.ORBmatcher(0.6, ori)
        n, m12 = m.SearchForTriangulation(g1, node1, valid1, g2, node2, valid2, F12, epi, sigma2, only_stereo)
        on, om12 = O.search_for_triangulation(o1, node1, valid1, o2, node2, valid2, F12, epi[0], epi[1], sigma2,
                                              only_stereo, ori)
        assert n == on and np.array_equal(m12, om12)
        assert n == int((m12 >= 0).sum())
        assert valid1[m12 >= 0].all() and valid2[m12[m12 >= 0]].all()
        if only_stereo:
            assert (ur1[m12 >= 0] >= 0).all() and (ur2[m12[m12 >= 0]] >= 0).all()
        total += n
    assert total > (20 if only_stereo else 100)
    n0, m0 = pkg.ORBmatcher(0.6, True).SearchForTriangulation(g1, node1, None, g2, node2, None, np.zeros(9, np.float32),
                                                              (0, 0), sigma2)
    assert n0 == 0 and (m0 == -1).all()            # den == 0 -> CheckDistEpipolarLine false (:152-153)


def test_distinctive_descriptors(env):
    """MapPoint::ComputeDistinctiveDescriptors batched over map points: same chosen observation as the oracle,
    for short (N <= 64, register path) and long (LDS path) observation lists, with duplicates forcing median ties."""
    pkg, M, O = env
    rng = np.random.default_rng(12)
    base = rng.integers(0, 256, (30, 32), dtype=np.uint8)
    lists = []
    for N in [0, 1, 2, 3, 4, 5, 8, 17, 63, 64, 65, 100, 129, 300, 2048] + list(rng.integers(1, 40, 200)):
        d = base[rng.integers(0, 30, N)].copy()
        if N:
            noise = rng.integers(0, 256, (N, 32), dtype=np.uint8) & rng.integers(0, 256, (N, 32), dtype=np.uint8) \
                & rng.integers(0, 256, (N, 32), dtype=np.uint8)
            d ^= noise * (rng.random((N, 1)) < 0.6)
        lists.append(d)
    m = pkg.ORBmatcher()
    got = m.ComputeDistinctiveDescriptors(lists)
    want = np.array([O.distinctive_descriptor(d) for d in lists], np.int32)
    assert np.array_equal(got, want)
    assert got[0] == -1 and got[1] == 0
    # independent numpy check of the definition on one list
    d = lists[12].astype(np.uint8)
    dist = np.unpackbits(d[:, None, :] ^ d[None, :, :], axis=2).sum(2)
    med = np.sort(dist, axis=1)[:, int(0.5 * (len(d) - 1))]
    assert got[12] == int(np.argmin(med))
    with pytest.raises(pkg.OrbHipError):
        m.ComputeDistinctiveDescriptors([np.zeros((2049, 32), np.uint8)])
    assert len(m.ComputeDistinctiveDescriptors([])) == 0


def test_search_by_bow_device_matches_host_api(env):
    """orbhip_search_by_bow_device (sorting, grouping, matching and cull in one launch) against the host-pointer API
    and the oracle, with valid / blocked flags, few crowded nodes (candidate lists > 64) and an empty pair."""
    import torch
    pkg, M, O = env
    rng = np.random.default_rng(5)
    dev = torch.device("cuda:0")
    cap, frames = 1400, 4
    sf = np.float32(1.2) ** np.arange(8, dtype=np.float32)
    base = rng.integers(0, 256, (60, 32), dtype=np.uint8)
    ns = [1300, 1400, 0, 900]
    kps = np.zeros((frames, cap), pkg.KP_DTYPE)
    desc = np.zeros((frames, cap, 32), np.uint8)
    node = np.full((frames, cap), 0xFFFFFFFF, np.uint32)
    valid = np.zeros((frames, cap), np.uint8)
    blocked = np.zeros((frames, cap), np.uint8)
    for f, n in enumerate(ns):
        d = base[rng.integers(0, len(base), n)].copy()
        d ^= (rng.integers(0, 256, (n, 32), dtype=np.uint8) & rng.integers(0, 256, (n, 32), dtype=np.uint8)
              & rng.integers(0, 256, (n, 32), dtype=np.uint8)) * (rng.random((n, 1)) < 0.7)
        desc[f, :n] = d
        kps[f, :n]["x"] = rng.uniform(0, 640, n); kps[f, :n]["y"] = rng.uniform(0, 480, n)
        kps[f, :n]["angle"] = rng.uniform(0, 360, n).astype(np.float32); kps[f, :n]["octave"] = rng.integers(0, 8, n)
        node[f, :n] = rng.integers(0, 9, n) * 1000 + 5
        node[f, :n][rng.random(n) < 0.05] = 0xFFFFFFFF
        valid[f, :n] = rng.random(n) < 0.8
        blocked[f, :n] = rng.random(n) < 0.2
    t = {k: torch.from_numpy(v.view(np.uint8) if v.dtype == pkg.KP_DTYPE else v).to(dev)
         for k, v in dict(kps=kps, desc=desc, valid=valid, blocked=blocked).items()}
    t_node = torch.from_numpy(node.view(np.int32)).to(dev)
    t_n = torch.tensor(ns, dtype=torch.int32, device=dev)
    img = np.zeros((480, 640), np.uint8)
    for nnratio, ori, max_dist, use_flags in ((0.75, True, 50, True), (0.9, False, 100, False)):
        m = pkg.ORBmatcher(nnratio, ori)
        pairs = 3                                   # (0,1), (1,2) with an empty frame, (2,3) empty key-frame side
        d_m12 = torch.full((pairs, cap), -5, dtype=torch.int32, device=dev)
        d_nm = torch.full((pairs,), -5, dtype=torch.int32, device=dev)
        side = (t["kps"].data_ptr(), t["desc"].data_ptr(), t_n.data_ptr(), t_node.data_ptr())
        m.SearchByBoWDevice(pairs, cap, side, 0, 1, side, 1, 1, d_m12.data_ptr(), d_nm.data_ptr(), max_dist,
                            t["valid"].data_ptr() if use_flags else 0, t["blocked"].data_ptr() if use_flags else 0)
        m.sync()
        got, gn = d_m12.cpu().numpy(), d_nm.cpu().numpy()
        for p in range(pairs):
            f1, f2 = p, p + 1
            g1, o1, keep1 = _views(pkg, O, img, kps[f1, :ns[f1]], desc[f1, :ns[f1]], sf)
            g2, o2, keep2 = _views(pkg, O, img, kps[f2, :ns[f2]], desc[f2, :ns[f2]], sf)
            v1 = valid[f1, :ns[f1]] if use_flags else None
            b2 = blocked[f2, :ns[f2]] if use_flags else None
            on, om12 = O.search_by_bow(o1, node[f1, :ns[f1]], v1, o2, node[f2, :ns[f2]], b2, max_dist, nnratio, ori)
            assert gn[p] == on and np.array_equal(got[p, :ns[f1]], om12), (p, gn[p], on)
            assert (got[p, ns[f1]:] == -1).all()
            hn, hm12 = m.SearchByBoW(g1, node[f1, :ns[f1]], v1, g2, node[f2, :ns[f2]], b2, max_dist)
            assert hn == on and np.array_equal(hm12, om12)
        assert gn[0] > 50
    with pytest.raises(pkg.OrbHipError):
        m.SearchByBoWDevice(1, 4097, side, 0, 1, side, 1, 1, d_m12.data_ptr(), d_nm.data_ptr())


def test_frame_glue_grid_and_rgbd(env):
    """Frame::AssignFeaturesToGrid and Frame::ComputeStereoFromRGBD (host and device-resident forms) vs the oracle;
    the CSR grid must also reproduce GetFeaturesInArea's visiting order when walked cell-major."""
    import torch
    pkg, M, O = env
    rng = np.random.default_rng(3)
    W, H = 1241, 376
    img = synth_frame(6, W, H)
    ext = pkg.ORBextractor(2000, 1.2, 8, 20, 7)
    kps, desc = ext(img)
    sf = ext.GetScaleFactors()
    kun = kps.copy()
    kun["x"] += rng.uniform(-30, 30, len(kps)).astype(np.float32)     # "undistorted" keys partly outside the bounds
    kun["y"] += rng.uniform(-30, 30, len(kps)).astype(np.float32)
    bounds = (-12.5, -7.25, W + 9.0, H + 3.5)                          # bounds of an undistorted image (Frame.cc:436-463)
    keep = []
    gv = pkg.FrameView(kun, desc, sf, bounds)
    ov = O.make_frame(kun, desc, None, bounds, sf, keep)
    m = pkg.ORBmatcher()
    cell_of, start, items = m.AssignFeaturesToGrid(gv)
    ocell, ostart, oitems = O.assign_features_to_grid(ov)
    assert np.array_equal(cell_of, ocell) and np.array_equal(start, ostart) and np.array_equal(items, oitems)
    assert (cell_of == -1).sum() > 0 and start[-1] == (cell_of >= 0).sum()
    # walking the CSR cell-major over a window = Frame::GetFeaturesInArea without the distance test
    whole = np.concatenate([items[start[c]:start[c + 1]] for c in range(64 * 48)])
    area = O.features_in_area(ov, W / 2.0, H / 2.0, 1e6)
    assert np.array_equal(whole, area)
    # RGB-D: depth image with holes (0), negatives and NaN-free positives
    depth = rng.uniform(0.3, 40.0, (H, W)).astype(np.float32)
    depth[rng.random((H, W)) < 0.2] = 0.0
    depth[rng.random((H, W)) < 0.05] = -1.0
    ur, dp = m.ComputeStereoFromRGBD(kps, kun, depth, 386.1448)
    our, odp = O.compute_stereo_from_rgbd(kps, kun, depth, np.float32(386.1448))
    assert np.array_equal(ur, our) and np.array_equal(dp, odp)
    assert (dp > 0).sum() > 0.5 * len(kps) and (dp == -1).sum() > 0.1 * len(kps)
    # device-resident batch of 2 frames (second one empty)
    dev = torch.device("cuda:0")
    cap = len(kps) + 7
    dk = torch.zeros((2, cap, 28), dtype=torch.uint8, device=dev)
    dku = torch.zeros((2, cap, 28), dtype=torch.uint8, device=dev)
    dk[0, :len(kps)] = torch.from_numpy(kps.view(np.uint8).reshape(-1, 28)).to(dev)
    dku[0, :len(kps)] = torch.from_numpy(kun.view(np.uint8).reshape(-1, 28)).to(dev)
    dn = torch.tensor([len(kps), 0], dtype=torch.int32, device=dev)
    d_cell = torch.full((2, cap), -9, dtype=torch.int32, device=dev)
    d_items = torch.full((2, cap), -9, dtype=torch.int32, device=dev)
    d_start = torch.full((2, 64 * 48 + 1), -9, dtype=torch.int32, device=dev)
    L = pkg.capi.lib()
    b = [np.float32(v) for v in bounds]
    pkg.capi.check(L.orbhip_assign_features_to_grid_device(m._h, 2, dku.data_ptr(), dn.data_ptr(), cap, b[0], b[1],
                                                           np.float32(64) / (b[2] - b[0]), np.float32(48) / (b[3] - b[1]),
                                                           d_cell.data_ptr(), d_start.data_ptr(), d_items.data_ptr()),
                   "orbhip_assign_features_to_grid_device")
    dd = torch.from_numpy(np.stack([depth, depth])).to(dev)
    d_ur = torch.zeros((2, cap), dtype=torch.float32, device=dev)
    d_dp = torch.zeros((2, cap), dtype=torch.float32, device=dev)
    pkg.capi.check(L.orbhip_compute_stereo_from_rgbd_device(m._h, 2, dk.data_ptr(), dku.data_ptr(), dn.data_ptr(), cap,
                                                            dd.data_ptr(), H, W, W, H * W, np.float32(386.1448),
                                                            d_ur.data_ptr(), d_dp.data_ptr()),
                   "orbhip_compute_stereo_from_rgbd_device")
    m.sync()
    assert np.array_equal(d_cell[0, :len(kps)].cpu().numpy(), ocell)
    assert np.array_equal(d_start[0].cpu().numpy(), ostart) and np.array_equal(d_items[0, :ostart[-1]].cpu().numpy(), oitems)
    assert (d_start[1].cpu().numpy() == 0).all()
    assert np.array_equal(d_ur[0, :len(kps)].cpu().numpy(), our) and np.array_equal(d_dp[0, :len(kps)].cpu().numpy(), odp)
    with pytest.raises(pkg.OrbHipError):
        big = np.zeros(4097, pkg.KP_DTYPE)
        m.AssignFeaturesToGrid(pkg.FrameView(big, np.zeros((4097, 32), np.uint8), sf, bounds))


def test_undistort_keypoints(env):
    """Frame::UndistortKeyPoints: host and device-resident forms bit-identical to the oracle (fp64, no contraction)."""
    import torch
    pkg, M, O = env
    rng = np.random.default_rng(2)
    n = 3000
    k = np.zeros(n, pkg.KP_DTYPE)
    k["x"] = rng.uniform(-5, 760, n).astype(np.float32); k["y"] = rng.uniform(-5, 490, n).astype(np.float32)
    k["angle"] = rng.uniform(0, 360, n).astype(np.float32); k["octave"] = rng.integers(0, 8, n); k["class_id"] = -1
    m = pkg.ORBmatcher()
    cases = ((458.654, 457.296, 367.215, 248.375, (-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05)),     # EuRoC.yaml
             (517.306408, 516.469215, 318.643040, 255.313989, (0.262383, -0.953104, -0.005358, 0.002628, 1.163314)),  # TUM1.yaml
             (718.856, 718.856, 607.1928, 185.2157, (0.0, 0.0, 0.0, 0.0)))                                    # KITTI: copy
    for fx, fy, cx, cy, dist in cases:
        got = m.UndistortKeyPoints(k, fx, fy, cx, cy, dist)
        want = O.undistort_keypoints(k, fx, fy, cx, cy, dist)
        assert all(np.array_equal(got[f], want[f]) for f in k.dtype.names), dist
    dev = torch.device("cuda:0")
    cap = n + 5
    dk = torch.zeros((2, cap, 28), dtype=torch.uint8, device=dev)
    dk[0, :n] = torch.from_numpy(k.view(np.uint8).reshape(-1, 28)).to(dev)
    dk[1, :100] = dk[0, 50:150]
    dn = torch.tensor([n, 100], dtype=torch.int32, device=dev)
    fx, fy, cx, cy, dist = cases[1]
    d5 = np.array(dist, np.float32)
    pkg.capi.check(pkg.capi.lib().orbhip_undistort_keypoints_device(m._h, 2, dk.data_ptr(), dn.data_ptr(), cap, fx, fy, cx, cy,
                                                                    pkg.capi.ptr(d5), dk.data_ptr()), "undistort_device")   # in place
    m.sync()
    want = O.undistort_keypoints(k, fx, fy, cx, cy, dist)
    got0 = dk[0, :n].cpu().numpy().view(pkg.KP_DTYPE).reshape(-1)
    got1 = dk[1, :100].cpu().numpy().view(pkg.KP_DTYPE).reshape(-1)
    assert all(np.array_equal(got0[f], want[f]) for f in k.dtype.names)
    assert all(np.array_equal(got1[f], want[50:150][f]) for f in k.dtype.names)


def test_randomised_parity_sweep():
    """tools/stress_parity.py: random sizes, feature counts, scale factors, level counts, thresholds, observed / taken
    patterns for the extractor and both SearchByProjection searches.  Seed 11 contains the case (605x581, 2 levels) in
    which a map-point query has to give a held slot up because its second candidate changes level -- the reason the
    map-point search resolves by fixed-point iteration and not by deferred acceptance."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_parity.py"), "--cases", "24", "--seed", "11"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "all cases passed" in r.stdout
