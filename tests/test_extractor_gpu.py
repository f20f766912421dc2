"""Parity of the HIP extractor against the oracle, through the C ABI.  Bit-exact bar:
keypoint fields (x, y, size, angle, response, octave, class_id) and 32-byte descriptors."""
import os
import zlib

import numpy as np
import pytest

from helpers import assert_kps_equal, assert_stagewise_equal, synth_frame

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods(oracle):
    import orb_slam2_comment_amd as pkg
    return pkg, oracle


@pytest.mark.parametrize("W,H,nf,seed", [
    (1241, 376, 1000, 1), (1241, 376, 1000, 2), (1241, 376, 1000, 3),   # BASELINE configs[1] (KITTI shape)
    (752, 480, 2000, 1), (752, 480, 1000, 4),                           # configs[4] (EuRoC shape)
    (320, 240, 500, 1), (320, 240, 500, 2), (641, 479, 777, 5),         # odd sizes / small
])
def test_extract_matches_oracle_stage_by_stage(mods, W, H, nf, seed):
    pkg, O = mods
    img = synth_frame(seed, W, H)
    ext = pkg.ORBextractor(nf, 1.2, 8, 20, 7)
    ora = O.OracleExtractor(nf, 1.2, 8, 20, 7)
    kps, desc = ext(img)
    okps, odesc = ora.extract(img)
    assert_stagewise_equal(ext, ora, 8, "seed %d" % seed)
    assert_kps_equal(kps, okps)
    assert np.array_equal(desc, odesc)
    assert len(kps) >= 0.95 * nf   # textured input fills (nearly) every level quota


@pytest.mark.parametrize("scale,nlevels,ini,mn", [(1.2, 4, 20, 7), (1.5, 5, 30, 10), (1.1, 8, 12, 5), (2.0, 3, 20, 7),
                                                  (2.6, 3, 20, 7)])   # 2.6: tap groups wider than 8 bytes -> the general resize kernel
def test_other_constructor_arguments(mods, scale, nlevels, ini, mn):
    pkg, O = mods
    img = synth_frame(11, 640, 480)
    ext = pkg.ORBextractor(800, scale, nlevels, ini, mn)
    ora = O.OracleExtractor(800, scale, nlevels, ini, mn)
    kps, desc = ext(img)
    okps, odesc = ora.extract(img)
    assert_stagewise_equal(ext, ora, nlevels)
    assert_kps_equal(kps, okps)
    assert np.array_equal(desc, odesc)
    t = ora.tables()
    assert np.array_equal(ext.GetScaleFactors(), t["scale"])
    assert np.array_equal(ext.GetInverseScaleFactors(), t["inv_scale"])
    assert np.array_equal(ext.GetScaleSigmaSquares(), t["sigma2"])
    assert np.array_equal(ext.GetInverseScaleSigmaSquares(), t["inv_sigma2"])
    assert np.array_equal(ext.features_per_level(), t["feat"])
    assert ext.GetLevels() == nlevels


def test_batch_equals_single_and_oracle(mods):
    pkg, O = mods
    ext = pkg.ORBextractor(1000, 1.2, 8, 20, 7)
    ora = O.OracleExtractor(1000, 1.2, 8, 20, 7)
    frames = np.stack([synth_frame(s) for s in range(20, 26)])
    res = ext.extract_batch(frames)
    for b in range(len(frames)):
        okps, odesc = ora.extract(frames[b])
        assert_kps_equal(res[b][0], okps, "frame %d" % b)
        assert np.array_equal(res[b][1], odesc)
    # handle reuse with a different size and back
    k2, d2 = ext(synth_frame(1, 320, 240))
    assert len(k2) > 0
    k3, d3 = ext(frames[0])
    assert np.array_equal(d3, res[0][1])


def test_edge_cases(mods):
    pkg, O = mods
    ext = pkg.ORBextractor(500, 1.2, 8, 20, 7)
    ora = O.OracleExtractor(500, 1.2, 8, 20, 7)
    # empty image: silent return, no keypoints (src/ORBextractor.cc:1046-1047)
    k, d = ext(np.zeros((0, 0), np.uint8))
    assert len(k) == 0 and d.shape == (0, 32)
    # non-8UC1 input: the reference asserts (:1050)
    with pytest.raises(TypeError):
        ext(np.zeros((240, 320), np.float32))
    # flat and saturated images: zero keypoints, descriptors released (:1064-1065)
    for v in (0, 128, 255):
        k, d = ext(np.full((240, 320), v, np.uint8))
        assert len(k) == 0
    # minThFAST fallback: low-contrast texture is only found at threshold 7 (:812-816)
    rng = np.random.default_rng(3)
    low = (128 + rng.integers(-12, 13, (240, 320))).astype(np.uint8)
    k, d = ext(low)
    ok, od = ora.extract(low)
    assert_kps_equal(k, ok)
    assert np.array_equal(d, od) and len(k) > 50
    # row stride != cols (a cv::Mat ROI)
    big = synth_frame(9, 400, 300)
    view = big[20:260, 30:350]
    k, d = ext(view)
    ok, od = ora.extract(np.ascontiguousarray(view))
    assert_kps_equal(k, ok)
    assert np.array_equal(d, od)
    # fewer candidates than the quota: isolated squares
    sparse = np.full((240, 320), 90, np.uint8)
    for i, (x, y) in enumerate([(60, 60), (200, 80), (120, 170), (260, 190)]):
        sparse[y:y + 14, x:x + 14] = 200 - 10 * i
    k, d = ext(sparse)
    ok, od = ora.extract(sparse)
    assert_kps_equal(k, ok)
    assert np.array_equal(d, od) and 0 < len(k) < 100
    # image too small for an 8-level pyramid: explicit error, not garbage
    with pytest.raises(pkg.OrbHipError) as ei:
        ext(synth_frame(1, 100, 80))
    assert ei.value.code == -4


def test_maximum_quota_and_candidate_pressure(mods):
    """Dense checkerboard-like texture: thousands of candidates per level, quota 5000."""
    pkg, O = mods
    rng = np.random.default_rng(7)
    blocks = rng.integers(0, 2, (60, 160)).astype(np.uint8) * 120 + 60
    img = np.kron(blocks, np.ones((8, 8), np.uint8))[:376, :1241].copy()
    img = np.ascontiguousarray(np.pad(img, ((0, 376 - img.shape[0]), (0, 1241 - img.shape[1])), mode="edge"))
    ext = pkg.ORBextractor(5000, 1.2, 8, 20, 7)
    ora = O.OracleExtractor(5000, 1.2, 8, 20, 7)
    k, d = ext(img)
    ok, od = ora.extract(img)
    assert_stagewise_equal(ext, ora, 8)
    assert_kps_equal(k, ok)
    assert np.array_equal(d, od)
    assert len(k) > 3000


@pytest.mark.parametrize("kind,W,H,scale,nlev,ini,mn", [
    ("checker3", 1241, 376, 1.2, 8, 20, 7), ("checker3", 640, 480, 1.5, 4, 20, 7), ("noise", 752, 480, 1.2, 8, 20, 7),
    ("noise", 1241, 376, 1.2, 8, 9, 2), ("mixed", 901, 403, 1.3, 6, 20, 7), ("checker2", 333, 251, 1.2, 5, 30, 10)])
def test_fast_survivor_list_overflow_runs_in_row_bands(mods, kind, W, H, scale, nlev, ini, mn):
    """k_fast_cells keeps a survivor list far smaller than a cell (8 wavefronts per SIMD fit the LDS that way) and repeats a
    round with more survivors in row bands.  Images on which (nearly) EVERY pixel passes the compass pre-test -- a 3-px
    checkerboard: the four compass pixels of every centre are at the opposite grey level; white noise -- force that path on
    whole levels; the mixed image has both kinds of cell side by side."""
    pkg, O = mods
    rng = np.random.default_rng(19)
    yy, xx = np.mgrid[0:H, 0:W]
    p = 3 if kind != "checker2" else 2
    checker = ((((xx // p) + (yy // p)) & 1) * 180 + 40).astype(np.uint8)
    noise = rng.integers(0, 256, (H, W), dtype=np.uint8)
    if kind.startswith("checker"):
        img = checker
    elif kind == "noise":
        img = noise
    else:
        img = synth_frame(3, W, H)
        img[:, W // 3:2 * W // 3] = checker[:, W // 3:2 * W // 3]
        img[H // 2:, 2 * W // 3:] = noise[H // 2:, 2 * W // 3:]
    img = np.ascontiguousarray(img)
    ext = pkg.ORBextractor(3000, scale, nlev, ini, mn)
    ora = O.OracleExtractor(3000, scale, nlev, ini, mn)
    k, d = ext(img)
    ok, od = ora.extract(img)
    assert_stagewise_equal(ext, ora, nlev, kind)
    assert_kps_equal(k, ok, kind)
    assert np.array_equal(d, od)
    # the same handle on an ordinary frame afterwards, and a lazy level 0 handle on the dense one
    img2 = synth_frame(5, W, H)
    k2, d2 = ext(img2)
    ok2, od2 = ora.extract(img2)
    assert_kps_equal(k2, ok2, kind + " (ordinary frame after)")
    assert np.array_equal(d2, od2)
    ext.set_lazy_level0(True)
    k3, d3 = ext(img)
    assert_kps_equal(k3, ok, kind + " (lazy level 0)")
    assert np.array_equal(d3, od)


def test_blur_kernel_is_a_data_table(mods):
    """The 7 blur weights are configuration, not code (OpenCV-version dependent)."""
    pkg, O = mods
    ext = pkg.ORBextractor(300, 1.2, 4, 20, 7)
    img = synth_frame(2, 320, 240)
    ext.set_blur_kernel([0, 0, 0, 255, 0, 0, 0])     # near-identity (255/256)^2: <= 2 grey levels off
    ext(img)
    d = np.abs(ext.blurred_level(0).astype(int) - ext.image_pyramid(0).astype(int))
    assert d.max() <= 2
    ext.set_blur_kernel([18, 34, 49, 55, 49, 34, 18])
    ext(img)
    assert not np.array_equal(ext.blurred_level(0), ext.image_pyramid(0))


def test_golden_fixtures(mods):
    """Committed oracle outputs (tests/golden/make_golden.py): GPU must reproduce them bit for bit."""
    pkg, O = mods
    path = os.path.join(os.path.dirname(__file__), "golden", "extract_golden.npz")
    g = np.load(path)
    for key in sorted(k[:-4] for k in g.files if k.endswith("_kps")):
        seed, W, H, nf = (int(v) for v in key.split("_")[1:])
        ext = pkg.ORBextractor(nf, 1.2, 8, 20, 7)
        k, d = ext(synth_frame(seed, W, H))
        gk = np.frombuffer(zlib.decompress(g[key + "_kps"].tobytes()), pkg.KP_DTYPE)
        gd = np.frombuffer(zlib.decompress(g[key + "_desc"].tobytes()), np.uint8).reshape(-1, 32)
        assert_kps_equal(k, gk, key)
        assert np.array_equal(d, gd), key


def test_full_size_batch_properties(mods):
    """BASELINE size: 64 frames 1241x376 through the device API; size-independent properties:
    determinism, count bounds, level-major ordering, keypoints inside the level borders."""
    import torch
    pkg, O = mods
    B, H, W = 64, 376, 1241
    frames = np.stack([synth_frame(100 + (s % 8)) for s in range(B)])
    ext = pkg.ORBextractor(1000, 1.2, 8, 20, 7)
    cap = ext.capacity(H, W)
    dev = torch.device("cuda:0")
    d_img = torch.from_numpy(frames).to(dev)
    d_kps = torch.zeros((B, cap, 7), dtype=torch.int32, device=dev)
    d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    d_n = torch.zeros(B, dtype=torch.int32, device=dev)
    d_st = torch.zeros(B, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    outs = []
    for _ in range(2):
        ext.extract_batch_device(d_img.data_ptr(), B, H, W, d_kps.data_ptr(), d_desc.data_ptr(), cap,
                                 d_n.data_ptr(), d_st.data_ptr())
        ext.sync()
        outs.append((d_kps.cpu().numpy().copy(), d_desc.cpu().numpy().copy(), d_n.cpu().numpy().copy()))
    assert np.all(d_st.cpu().numpy() == 0)
    n = outs[0][2]
    assert np.array_equal(n, outs[1][2])
    for b in range(B):
        assert np.array_equal(outs[0][0][b, :n[b]], outs[1][0][b, :n[b]])
        assert np.array_equal(outs[0][1][b, :n[b]], outs[1][1][b, :n[b]])
    assert np.all((n >= 1000) & (n <= cap))
    # frames repeat with period 8 -> identical results
    for b in range(8, B):
        assert n[b] == n[b % 8]
        assert np.array_equal(outs[0][1][b, :n[b]], outs[0][1][b % 8, :n[b]])
    k0 = outs[0][0][0, :n[0]].copy().view(pkg.KP_DTYPE).reshape(-1)
    assert np.all(np.diff(k0["octave"]) >= 0)
    sf = ext.GetScaleFactors()
    assert np.all(k0["x"] >= 19) and np.all(k0["x"] < W) and np.all(k0["y"] < H)
    # spot-check 3 frames against the oracle
    ora = O.OracleExtractor(1000, 1.2, 8, 20, 7)
    for b in (0, 3, 7):
        ok, od = ora.extract(frames[b])
        assert_kps_equal(outs[0][0][b, :n[b]].copy().view(pkg.KP_DTYPE).reshape(-1), ok)
        assert np.array_equal(outs[0][1][b, :n[b]], od)


def _content(kind, W, H, rng):
    if kind == "noise":
        return rng.integers(0, 256, (H, W), dtype=np.uint8)
    if kind == "lownoise":
        return (rng.integers(100, 140, (H, W))).astype(np.uint8)
    if kind == "checker":
        s = int(rng.integers(3, 17))
        yy, xx = np.mgrid[0:H, 0:W]
        return (((yy // s + xx // s) % 2) * int(rng.integers(30, 200)) + 20).astype(np.uint8)
    if kind == "gradient":
        yy, xx = np.mgrid[0:H, 0:W]
        return ((xx * 255 // max(W - 1, 1) + yy * 3) % 256).astype(np.uint8)
    if kind == "blobs":
        img = np.full((H, W), 60, np.int32)
        for _ in range(300):
            x, y, r = int(rng.integers(0, W)), int(rng.integers(0, H)), int(rng.integers(2, 9))
            img[max(0, y - r):y + r, max(0, x - r):x + r] = int(rng.integers(0, 256))
        return img.astype(np.uint8)
    return synth_frame(int(rng.integers(1, 1000)), W, H)


@pytest.mark.parametrize("case", range(14))
def test_randomised_sizes_and_contents(mods, case):
    """Random geometry (every cell alignment / partial cell / odd pitch) x content classes that stress
    individual stages: dense corners (checker/noise: full candidate pressure, plateaus), threshold fallback
    (low noise), no corners (gradient), sparse blobs (fewer candidates than the quota)."""
    pkg, O = mods
    rng = np.random.default_rng(1000 + case)
    W, H = int(rng.integers(170, 900)), int(rng.integers(170, 700))
    kind = ["noise", "lownoise", "checker", "gradient", "blobs", "synth", "checker"][case % 7]
    nlevels = int(rng.integers(2, 6))           # small images: keep every level >= 40 px
    nf = int(rng.integers(50, 1500))
    ini, mn = int(rng.integers(10, 40)), int(rng.integers(3, 10))
    img = _content(kind, W, H, rng)
    ext = pkg.ORBextractor(nf, 1.2, nlevels, ini, mn)
    ora = O.OracleExtractor(nf, 1.2, nlevels, ini, mn)
    k, d = ext(img)
    ok, od = ora.extract(img)
    assert_stagewise_equal(ext, ora, nlevels, "%s %dx%d" % (kind, W, H))
    assert_kps_equal(k, ok, "%s %dx%d" % (kind, W, H))
    assert np.array_equal(d, od)


def test_two_extractors_on_two_host_threads(mods):
    """Frame's stereo constructor runs the left and right extractor on two std::threads
    (src/Frame.cc:78-81): handles must be independent."""
    import threading
    pkg, O = mods
    from helpers import synth_stereo
    left, right = synth_stereo(5, 752, 480)
    eL, eR = pkg.ORBextractor(1200, 1.2, 8, 20, 7), pkg.ORBextractor(1200, 1.2, 8, 20, 7)
    res = {}

    def run(name, ext, img):
        for _ in range(5):
            res[name] = ext(img)
    tl = threading.Thread(target=run, args=("L", eL, left))
    tr = threading.Thread(target=run, args=("R", eR, right))
    tl.start(); tr.start(); tl.join(); tr.join()
    ora = O.OracleExtractor(1200, 1.2, 8, 20, 7)
    for name, img in (("L", left), ("R", right)):
        ok, od = ora.extract(img)
        assert_kps_equal(res[name][0], ok, name)
        assert np.array_equal(res[name][1], od)


def test_batch_with_frame_and_row_strides(mods):
    """orbhip_extract_batch on frames that are ROIs of a larger buffer (row stride > cols, frame stride > frame)."""
    import ctypes as C
    pkg, O = mods
    from orb_slam2_comment_amd import capi
    B, H, W, SH, SW = 3, 240, 320, 260, 352
    big = np.zeros((B, SH, SW), np.uint8)
    for b in range(B):
        big[b, 10:10 + H, 16:16 + W] = synth_frame(40 + b, W, H)
    ext = pkg.ORBextractor(400, 1.2, 6, 20, 7)
    cap = ext.capacity(H, W)
    kps = np.zeros((B, cap), pkg.KP_DTYPE); desc = np.zeros((B, cap, 32), np.uint8); n = np.zeros(B, np.int32)
    first = big[0, 10:, 16:]
    capi.check(capi.lib().orbhip_extract_batch(ext._h, first.ctypes.data_as(C.c_void_p), B, H, W, SW, SH * SW,
                                               capi.ptr(kps), capi.ptr(desc), cap, capi.ptr(n)), "orbhip_extract_batch")
    ora = O.OracleExtractor(400, 1.2, 6, 20, 7)
    for b in range(B):
        ok, od = ora.extract(np.ascontiguousarray(big[b, 10:10 + H, 16:16 + W]))
        assert_kps_equal(kps[b, :n[b]], ok, "frame %d" % b)
        assert np.array_equal(desc[b, :n[b]], od)


def test_device_entry_with_row_and_frame_strides(mods):
    """orbhip_extract_batch_device on frames that are ROIs of a larger DEVICE buffer: row stride > cols and an odd byte
    offset -- the pyramid kernels read the input image itself (level 0 and level 1), not a compacted copy."""
    import torch
    pkg, O = mods
    B, H, W, SH, SW = 3, 240, 321, 262, 357
    big = np.zeros((B, SH, SW), np.uint8)
    frames = [synth_frame(50 + b, W, H) for b in range(B)]
    for b in range(B):
        big[b, 11:11 + H, 17:17 + W] = frames[b]
    dev = torch.device("cuda", 0)
    d_big = torch.from_numpy(big).to(dev)
    ext = pkg.ORBextractor(400, 1.2, 6, 20, 7)
    cap = ext.capacity(H, W)
    d_k = torch.zeros((B, cap, 7), dtype=torch.int32, device=dev)
    d_d = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    d_n = torch.zeros(B, dtype=torch.int32, device=dev)
    d_s = torch.zeros(B, dtype=torch.int32, device=dev)
    ext.extract_batch_device(d_big.data_ptr() + 11 * SW + 17, B, H, W, d_k.data_ptr(), d_d.data_ptr(), cap, d_n.data_ptr(),
                             d_s.data_ptr(), stride=SW, frame_stride=SH * SW)
    ext.sync()
    n = d_n.cpu().numpy()
    assert int(d_s.abs().sum().item()) == 0
    kps = d_k.cpu().numpy().view(pkg.KP_DTYPE).reshape(B, cap)
    desc = d_d.cpu().numpy()
    ora = O.OracleExtractor(400, 1.2, 6, 20, 7)
    for b in range(B):
        ok, od = ora.extract(frames[b])
        assert_kps_equal(kps[b, :n[b]], ok, "frame %d" % b)
        assert np.array_equal(desc[b, :n[b]], od)
        assert np.array_equal(ext.image_pyramid(0, with_border=True, frame=b), ora.level_padded(0))
        assert np.array_equal(ext.image_pyramid(1, with_border=True, frame=b), ora.level_padded(1))


@pytest.mark.parametrize("B", [136, 200])
def test_large_device_batches_use_more_keypoint_slots_per_wavefront(mods, B):
    """k_describe_fused gives a wavefront 2, 3 or 4 consecutive keypoint slots depending on how many wavefronts the launch
    has (orbhip_extractor.hip, desc_per_wave): 136 frames at nFeatures 1000 take 3, 200 frames 4 -- a level boundary then
    falls INSIDE a wavefront's slots.  Every frame of the batch must equal the single-frame extraction (2 slots) of the same
    image and the oracle."""
    import torch
    pkg, O = mods
    H, W, nf = 240, 320, 1000
    base = [synth_frame(70 + i, W, H) for i in range(5)]
    frames = np.stack([base[i % 5] for i in range(B)])
    dev = torch.device("cuda", 0)
    d_img = torch.from_numpy(frames).to(dev)
    ext = pkg.ORBextractor(nf, 1.2, 8, 20, 7)
    cap = ext.capacity(H, W)
    d_k = torch.zeros((B, cap, 7), dtype=torch.int32, device=dev)
    d_d = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    d_n = torch.zeros(B, dtype=torch.int32, device=dev)
    d_s = torch.zeros(B, dtype=torch.int32, device=dev)
    ext.extract_batch_device(d_img.data_ptr(), B, H, W, d_k.data_ptr(), d_d.data_ptr(), cap, d_n.data_ptr(), d_s.data_ptr())
    ext.sync()
    assert int(d_s.abs().sum().item()) == 0
    n = d_n.cpu().numpy()
    kps = d_k.cpu().numpy().view(pkg.KP_DTYPE).reshape(B, cap)
    desc = d_d.cpu().numpy()
    ora = O.OracleExtractor(nf, 1.2, 8, 20, 7)
    single = pkg.ORBextractor(nf, 1.2, 8, 20, 7)
    ref = []
    for i in range(5):
        ok, od = ora.extract(base[i])
        sk, sd = single(base[i])
        assert_kps_equal(sk, ok, "single frame %d" % i)
        assert np.array_equal(sd, od)
        ref.append((ok, od))
    for b in range(B):
        ok, od = ref[b % 5]
        assert_kps_equal(kps[b, :n[b]], ok, "frame %d of %d" % (b, B))
        assert np.array_equal(desc[b, :n[b]], od)


def test_host_api_graph_replay_is_invalidated_correctly(mods):
    """The host-pointer entry replays a captured hipGraph; every event that changes a captured argument must drop it:
    batch size, capacity (new geometry), image size, blur weights, stream.  Results must not depend on the history."""
    pkg, O = mods
    ext = pkg.ORBextractor(500, 1.2, 8, 20, 7)
    a, b = synth_frame(11, 640, 360), synth_frame(12, 640, 360)
    big = synth_frame(13, 752, 480)
    ref = {}
    for name, img in (("a", a), ("b", b), ("big", big)):
        ref[name] = O.OracleExtractor(500, 1.2, 8, 20, 7).extract(img)

    def same(got, want):
        assert_kps_equal(got[0], want[0])
        assert np.array_equal(got[1], want[1])

    same(ext(a), ref["a"])                      # capture (batch 1)
    same(ext(b), ref["b"])                      # replay with new pixels
    res = ext.extract_batch(np.stack([a, b, a]))            # batch 3: new graph
    same(res[0], ref["a"]); same(res[1], ref["b"]); same(res[2], ref["a"])
    same(ext(b), ref["b"])                      # back to batch 1
    same(ext(big), ref["big"])                  # other image size: geometry rebinding
    same(ext(a), ref["a"])
    ext.set_blur_kernel([0, 0, 0, 255, 0, 0, 0])           # identity-like blur: descriptors change ...
    k2, d2 = ext(a)
    assert_kps_equal(k2, ref["a"][0])
    assert not np.array_equal(d2, ref["a"][1])
    ext.set_blur_kernel([18, 34, 49, 55, 49, 34, 18])      # ... and come back with the default weights
    same(ext(a), ref["a"])
    ext.set_profiling(True)                     # profiling path = plain launches
    same(ext(b), ref["b"])
    ext.set_profiling(False)
    same(ext(b), ref["b"])
    import torch
    st = torch.cuda.Stream()
    ext.set_stream(st.cuda_stream)
    same(ext(a), ref["a"])                      # capture on the caller's stream
    ext.set_stream(0)
    same(ext(b), ref["b"])


def test_capacity_error_still_delivers_every_frame(mods):
    """orbhip_extract_batch with cap below the needed capacity: ORBHIP_E_CAPACITY, but n[] and the first n[b] = cap
    keypoints / descriptors of EVERY frame are delivered (truncated in output order), not just the frames in front of
    the first overflowing one."""
    import ctypes as C
    pkg, O = mods
    from orb_slam2_comment_amd.capi import KP_DTYPE, lib, ptr
    ext = pkg.ORBextractor(1000, 1.2, 8, 20, 7)
    frames = np.stack([synth_frame(40 + s) for s in range(3)])
    full = ext.extract_batch(frames)
    B, rows, cols = frames.shape
    cap = 500
    kps = np.zeros((B, cap), KP_DTYPE)
    desc = np.zeros((B, cap, 32), np.uint8)
    n = np.full(B, -7, np.int32)
    rc = lib().orbhip_extract_batch(ext._h, ptr(frames), B, rows, cols, cols, rows * cols, ptr(kps), ptr(desc), cap, ptr(n))
    assert rc == pkg.capi.E_CAPACITY
    assert n.tolist() == [cap] * B
    for b in range(B):
        assert_kps_equal(kps[b], full[b][0][:cap], "frame %d" % b)
        assert np.array_equal(desc[b], full[b][1][:cap])
    # the handle stays usable
    again = ext.extract_batch(frames[:1])
    assert np.array_equal(again[0][1], full[0][1])


def test_large_host_batch_runs_as_a_chunk_pipeline(mods):
    """Host batches >= 32 frames run as 16-frame chunks on three streams (copy-in / kernels / copy-out).  Results must
    not depend on that, and every frame's pyramid stays resident afterwards (mvImagePyramid of a chunked batch)."""
    pkg, O = mods
    ext = pkg.ORBextractor(1000, 1.2, 8, 20, 7)
    uniq = [synth_frame(60 + s) for s in range(5)]
    frames = np.stack([uniq[i % 5] for i in range(40)])          # 2.5 chunks
    res = ext.extract_batch(frames)
    ora = O.OracleExtractor(1000, 1.2, 8, 20, 7)
    for s in range(5):
        okps, odesc = ora.extract(uniq[s])
        for b in range(s, 40, 5):
            assert_kps_equal(res[b][0], okps, "frame %d" % b)
            assert np.array_equal(res[b][1], odesc)
        for level in (0, 3, 7):
            ref = ora.level_padded(level)
            for b in (s, s + 15, s + 35):                          # one frame in each chunk
                assert np.array_equal(ext.image_pyramid(level, frame=b, with_border=True), ref), (b, level)
    # the same handle afterwards on a small batch (graph replay path) and on a larger one again
    small = ext.extract_batch(frames[:3])
    assert np.array_equal(small[2][1], res[2][1])
    again = ext.extract_batch(frames[:33])
    assert np.array_equal(again[32][1], res[32][1])


def test_stage_accessors_follow_every_extraction_of_a_reused_handle(mods):
    """mvImagePyramid / the blurred planes / the FAST candidates belong to the LAST extraction of a handle, also when
    that extraction was a hipGraph replay (same shape, batch and capacity as the previous call) and the blurred planes
    of the previous image had been produced on request in between (they are cached behind a validity flag)."""
    pkg, O = mods
    ext = pkg.ORBextractor(500, 1.2, 6, 20, 7)
    ora = O.OracleExtractor(500, 1.2, 6, 20, 7)
    imgs = [synth_frame(70 + s, 480, 320) for s in range(3)]
    for it, img in enumerate(imgs + imgs[:1]):          # capture, replay, replay, replay
        kps, desc = ext(img)
        okps, odesc = ora.extract(img)
        assert_kps_equal(kps, okps, "call %d" % it)
        assert np.array_equal(desc, odesc)
        assert_stagewise_equal(ext, ora, 6, "call %d" % it)   # downloads pyramid, candidates and blurred planes
    # the same through the device-pointer entry (plain launches) and the batch entry
    res = ext.extract_batch(np.stack(imgs))
    ora.extract(imgs[0])
    for l in range(6):
        if ora.level_blurred(l) is not None:
            assert np.array_equal(ext.blurred_level(l, frame=0), ora.level_blurred(l))
    assert np.array_equal(res[0][1], O.OracleExtractor(500, 1.2, 6, 20, 7).extract(imgs[0])[1])


def test_page_locked_strided_source_is_read_in_place(mods):
    """Host batches >= 32 frames whose images sit in page-locked memory skip the staging copy: every frame travels as one
    1-D copy of its rows in the caller's row stride (no rectangle copy anywhere) and the kernels read that layout."""
    import torch
    pkg, O = mods
    from orb_slam2_comment_amd.capi import KP_DTYPE, check, lib
    import ctypes as C
    Wd, Hd, stride, fstride, B = 400, 300, 448, 448 * 300 + 1024, 33
    pin = torch.zeros(B * fstride, dtype=torch.uint8).pin_memory()
    host = pin.numpy()
    host[:] = 0xA5                                                  # the gaps must not leak into any level
    uniq = [synth_frame(80 + s, Wd, Hd) for s in range(4)]
    for b in range(B):
        v = host[b * fstride: b * fstride + Hd * stride].reshape(Hd, stride)
        v[:, :Wd] = uniq[b % 4]
    ext = pkg.ORBextractor(600, 1.2, 7, 20, 7)
    cap = ext.capacity(Hd, Wd)
    kps = np.zeros((B, cap), KP_DTYPE); desc = np.zeros((B, cap, 32), np.uint8); n = np.zeros(B, np.int32)
    check(lib().orbhip_extract_batch(ext._h, C.c_void_p(pin.data_ptr()), B, Hd, Wd, stride, fstride,
                                     kps.ctypes.data_as(C.c_void_p), desc.ctypes.data_as(C.c_void_p), cap,
                                     n.ctypes.data_as(C.c_void_p)), "orbhip_extract_batch")
    ora = O.OracleExtractor(600, 1.2, 7, 20, 7)
    for s in range(4):
        okps, odesc = ora.extract(uniq[s])
        for b in range(s, B, 4):
            assert_kps_equal(kps[b, :n[b]], okps, "frame %d" % b)
            assert np.array_equal(desc[b, :n[b]], odesc)
        assert np.array_equal(ext.image_pyramid(0, frame=s + 28, with_border=True), ora.level_padded(0))


@pytest.mark.parametrize("W,H,nf,scale,nlev", [(1241, 376, 1000, 1.2, 8), (752, 480, 2000, 1.2, 8), (333, 251, 400, 1.3, 5),
                                               (640, 360, 500, 2.6, 3), (400, 300, 300, 1.2, 1)])
def test_lazy_level0_is_bit_identical_and_accessors_still_work(mods, W, H, nf, scale, nlev):
    """orbhip_extractor_set_lazy_level0: level 0 is read from the image (REFLECT_101 by index for the windows of border
    keypoints), mvImagePyramid[0] is written only when an accessor asks.  Same keypoints, same descriptors, same
    accessor results; scale 2.6 (level 1 through the general resize kernel, which reads the padded plane) and a one-level
    pyramid silently keep materialising."""
    pkg, O = mods
    ext = pkg.ORBextractor(nf, scale, nlev, 20, 7)
    ext.set_lazy_level0(True)
    ora = O.OracleExtractor(nf, scale, nlev, 20, 7)
    for seed in (31, 32):
        img = synth_frame(seed, W, H)
        kps, desc = ext(img)                       # capture, then replay
        okps, odesc = ora.extract(img)
        assert_kps_equal(kps, okps, "seed %d" % seed)
        assert np.array_equal(desc, odesc)
        l0 = kps[kps["octave"] == 0]
        if W >= 600:
            assert ((l0["x"] < 21) | (l0["y"] < 21) | (l0["x"] > W - 23) | (l0["y"] > H - 22)).any(), "no border keypoint in the test image"
    assert_stagewise_equal(ext, ora, nlev, "lazy")  # level 0 (with border) and its blurred plane are produced now
    # batch entry, then back to materialising
    res = ext.extract_batch(np.stack([synth_frame(31, W, H), img]))
    assert np.array_equal(res[1][1], odesc)
    assert np.array_equal(ext.image_pyramid(0, frame=1, with_border=True), ora.level_padded(0))
    ext.set_lazy_level0(False)
    kps2, desc2 = ext(img)
    assert_kps_equal(kps2, okps)
    assert np.array_equal(desc2, odesc)
    assert_stagewise_equal(ext, ora, nlev, "eager again")


def test_lazy_level0_device_entry_with_strides_and_stereo(mods):
    """Device-pointer entry with a row stride wider than the image and a frame stride with a gap, lazy level 0; then
    Frame::ComputeStereoMatches on two lazy handles (it needs mvImagePyramid[0] of both: built on demand)."""
    import torch
    pkg, O = mods
    from helpers import synth_stereo
    Wd, Hd, stride, B = 500, 300, 544, 4
    fstride = stride * Hd + 4096
    left, right = synth_stereo(9, Wd, Hd)
    imgs = [left, right, synth_frame(33, Wd, Hd), left]
    buf = np.full(B * fstride, 0x5A, np.uint8)
    for b in range(B):
        buf[b * fstride: b * fstride + Hd * stride].reshape(Hd, stride)[:, :Wd] = imgs[b]
    d_img = torch.from_numpy(buf).cuda()
    ext = pkg.ORBextractor(700, 1.2, 6, 20, 7)
    ext.set_lazy_level0(True)
    cap = ext.capacity(Hd, Wd)
    d_k = torch.zeros((B, cap, 7), dtype=torch.int32, device="cuda"); d_d = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
    d_n = torch.zeros(B, dtype=torch.int32, device="cuda"); d_s = torch.zeros(B, dtype=torch.int32, device="cuda")
    ext.set_stream(torch.cuda.current_stream().cuda_stream)
    ext.extract_batch_device(d_img.data_ptr(), B, Hd, Wd, d_k.data_ptr(), d_d.data_ptr(), cap, d_n.data_ptr(), d_s.data_ptr(),
                             stride=stride, frame_stride=fstride)
    torch.cuda.synchronize()
    n = d_n.cpu().numpy()
    kps = d_k.cpu().numpy().view(np.uint8).reshape(B, cap, 28).view(pkg.capi.KP_DTYPE).reshape(B, cap)
    desc = d_d.cpu().numpy()
    assert int(d_s.abs().sum().item()) == 0
    ora = O.OracleExtractor(700, 1.2, 6, 20, 7)
    for b in range(B):
        okps, odesc = ora.extract(imgs[b])
        assert_kps_equal(kps[b, :n[b]], okps, "frame %d" % b)
        assert np.array_equal(desc[b, :n[b]], odesc)
    assert np.array_equal(ext.image_pyramid(0, frame=3, with_border=True), ora.level_padded(0))
    ext.set_stream(0)
    # stereo: two lazy handles against two eager ones
    res = []
    for lazy in (True, False):
        eL, eR = pkg.ORBextractor(700, 1.2, 6, 20, 7), pkg.ORBextractor(700, 1.2, 6, 20, 7)
        eL.set_lazy_level0(lazy); eR.set_lazy_level0(lazy)
        kl, dl = eL(left); kr, dr = eR(right)
        res.append(pkg.ORBmatcher().ComputeStereoMatches(eL, eR, kl, dl, kr, dr, 386.1448, 386.1448 / 718.856))
    assert res[0][0] == res[1][0] and res[0][0] > 50
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])


def test_stage_gates_do_not_change_results(mods):
    """orbhip_extractor_set_stage_gate: three handles on three streams with their pyramid stages chained in a ring and
    their descriptor stages gated on another handle's FAST stage -- scheduling only, results identical to an ungated run."""
    import torch
    pkg, O = mods
    imgs = [synth_frame(90 + s, 640, 360) for s in range(6)]
    d_img = torch.from_numpy(np.stack(imgs)).cuda()
    exts = [pkg.ORBextractor(500, 1.2, 8, 20, 7) for _ in range(3)]
    streams = [torch.cuda.Stream() for _ in range(3)]
    evs = [[torch.cuda.Event() for _ in range(3)] for _ in range(2)]
    for h in range(3):
        exts[h].set_stream(streams[h].cuda_stream)
        for ring in evs:
            ring[h].record(streams[h])
    torch.cuda.synchronize()
    for h in range(3):
        exts[h].set_stage_gate(0, evs[0][(h - 1) % 3].cuda_event, evs[0][h].cuda_event)      # pyramid ring
        exts[h].set_stage_gate_record(1, evs[1][h].cuda_event)                                 # FAST of h ...
        exts[h].set_stage_gate_wait(3, evs[1][(h + 1) % 3].cuda_event)                         # ... releases descriptors of h - 1
    cap = exts[0].capacity(360, 640)
    k = torch.zeros((6, cap, 7), dtype=torch.int32, device="cuda"); d = torch.zeros((6, cap, 32), dtype=torch.uint8, device="cuda")
    n = torch.zeros(6, dtype=torch.int32, device="cuda"); st = torch.zeros(6, dtype=torch.int32, device="cuda")
    for rep in range(3):
        for h in range(3):
            sl = slice(2 * h, 2 * h + 2)
            exts[h].extract_batch_device(d_img[sl].data_ptr(), 2, 360, 640, k[sl].data_ptr(), d[sl].data_ptr(), cap, n[sl].data_ptr(),
                                         st[sl].data_ptr())
    torch.cuda.synchronize()
    assert int(st.abs().sum().item()) == 0
    nn = n.cpu().numpy()
    kk = k.cpu().numpy().view(np.uint8).reshape(6, cap, 28).view(pkg.capi.KP_DTYPE).reshape(6, cap)
    dd = d.cpu().numpy()
    ora = O.OracleExtractor(500, 1.2, 8, 20, 7)
    for b in range(6):
        okps, odesc = ora.extract(imgs[b])
        assert_kps_equal(kk[b, :nn[b]], okps, "frame %d" % b)
        assert np.array_equal(dd[b, :nn[b]], odesc)
    # the host entry with gates set (single frame: normally a hipGraph replay) takes plain launches instead
    torch.cuda.synchronize()
    kh, dh = exts[0](imgs[0])
    okps, odesc = ora.extract(imgs[0])
    assert_kps_equal(kh, okps)
    assert np.array_equal(dh, odesc)
    for e in exts:
        e.set_stage_gate(0, 0, 0); e.set_stage_gate(1, 0, 0); e.set_stage_gate(3, 0, 0); e.set_stream(0)
