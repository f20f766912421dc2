"""Parity of the HIP vocabulary transform (Frame::ComputeBoW) against the oracle, through the C ABI.
Words, weights, FeatureVector nodes and BowVector ids must be identical; BowVector values are doubles produced by
the same operation order and must be BIT-identical (asserted with array_equal, tolerance 0).
Parity unpinned: the reference ships no vocabulary file and no BoW fixtures, so the oracle itself is only checked
against hand-worked cases (tests/test_oracle_known_answers.py)."""
import numpy as np
import pytest

from helpers import frame_bounds, make_vocabulary, synth_frame, write_vocabulary

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env(oracle):
    import orb_slam2_comment_amd as pkg
    return pkg, oracle


def _same(g, o):
    for key in ("word_id", "word_weight", "node_id", "bow_ids", "bow_vals"):
        assert g[key].dtype == o[key].dtype and np.array_equal(g[key], o[key]), key


@pytest.mark.parametrize("k,L,scoring,weighting,irregular,order", [
    (10, 4, 0, 0, False, "bfs"),            # ORBvoc.txt shape (10^6 words there, 10^4 here), L1 + tf-idf
    (10, 3, 1, 0, False, "bfs"),            # L2 norm
    (4, 6, 5, 1, True, "interleaved"),      # dot product (no normalisation, /nd), ragged tree, scattered children
    (20, 2, 2, 2, True, "bfs"),             # widest branching, IDF: addIfNotExist
    (3, 5, 4, 3, True, "interleaved"),      # BINARY
    (1, 3, 0, 0, False, "bfs"),             # degenerate chain
])
def test_transform_matches_oracle(env, tmp_path, k, L, scoring, weighting, irregular, order):
    pkg, O = env
    voc = make_vocabulary(k, L, seed=100 * k + L, scoring=scoring, weighting=weighting, irregular=irregular, order=order)
    path = write_vocabulary(tmp_path / "voc.txt", voc)
    gv = pkg.ORBVocabulary()
    assert gv.loadFromTextFile(path)
    ov = O.OracleVocabulary(path)
    info = ov.info()
    assert (gv.getBranchingFactor(), gv.getDepthLevels(), gv.getScoringType(), gv.getWeightingType(), gv.size()) == \
        (info["k"], info["L"], info["scoring"], info["weighting"], info["n_words"])
    assert not gv.empty()
    rng = np.random.default_rng(7)
    ext = pkg.ORBextractor(2000, 1.2, 8, 20, 7)
    _, d_img = ext(synth_frame(2, 752, 480))
    # descriptors near tree nodes (deep, meaningful descents) + image descriptors + pure noise + exact node copies
    near = voc["desc"][rng.integers(0, len(voc["desc"]), 1500)].copy()
    near ^= (rng.integers(0, 256, near.shape, dtype=np.uint8) & rng.integers(0, 256, near.shape, dtype=np.uint8)
             & rng.integers(0, 256, near.shape, dtype=np.uint8))
    feats = np.concatenate([near, d_img, rng.integers(0, 256, (300, 32), dtype=np.uint8), voc["desc"][:200]])
    for levelsup in (4, 0, 1, L, L + 3):
        _same(gv.transform(feats, levelsup), ov.transform(feats, levelsup))
    for n in (0, 1, 15, 16, 17, 1024, 1025):
        _same(gv.transform(feats[:n], 4), ov.transform(feats[:n], 4))
    r = gv.transform(feats, 4)
    assert len(r["bow_ids"]) > 0 and (np.diff(r["bow_ids"].astype(np.int64)) > 0).all()
    if scoring != 5:
        norm = np.abs(r["bow_vals"]).sum() if scoring != 1 else np.sqrt((r["bow_vals"] ** 2).sum())
        assert abs(norm - 1.0) < 1e-9
    # same tree from arrays
    ga = pkg.ORBVocabulary.from_arrays(k, L, scoring, weighting, voc["parent"], voc["is_leaf"], voc["desc"], voc["weight"])
    _same(ga.transform(feats, 4), r)
    # a file without the trailing newline loads to the same tree
    g2 = pkg.ORBVocabulary()
    assert g2.loadFromTextFile(write_vocabulary(tmp_path / "voc2.txt", voc, trailing_newline=False))
    _same(g2.transform(feats[:500], 4), ov.transform(feats[:500], 4))


def test_load_errors_and_limits(env, tmp_path):
    pkg, O = env
    v = pkg.ORBVocabulary()
    assert not v.loadFromTextFile(tmp_path / "missing.txt") and v.empty()
    bad = tmp_path / "bad.txt"
    bad.write_text("10 11 0 0\n")                                    # L > 10 (:1362)
    assert not v.loadFromTextFile(bad)
    bad.write_text("10 6 0 0\n0 1 1 2 3\n")                           # truncated node line
    assert not v.loadFromTextFile(bad)
    bad.write_text("10 6 0 0\n5 1 " + " ".join(["0"] * 32) + " 1.0\n")   # parent after the node itself
    assert not v.loadFromTextFile(bad)
    with pytest.raises(RuntimeError):
        v.transform(np.zeros((4, 32), np.uint8))
    # header only: loads, but empty() -> transform leaves everything empty (:1134-1137)
    bad.write_text("10 6 0 0\n")
    assert v.loadFromTextFile(bad) and v.empty()
    r = v.transform(np.zeros((4, 32), np.uint8))
    assert r["bow_ids"].size == 0 and (r["node_id"] == pkg.capi.NO_NODE).all()
    voc = make_vocabulary(3, 2, seed=1)
    g = pkg.ORBVocabulary.from_arrays(3, 2, 0, 0, voc["parent"], voc["is_leaf"], voc["desc"], voc["weight"])
    with pytest.raises(pkg.OrbHipError):
        g.transform(np.zeros((8193, 32), np.uint8))                  # per-frame capacity
    _ = g.transform(np.zeros((8192, 32), np.uint8))
    with pytest.raises(pkg.OrbHipError):
        pkg.ORBVocabulary.from_arrays(3, 2, 0, 0, [0, 3], [1, 1], np.zeros((2, 32), np.uint8), [1.0, 1.0])


def test_device_batch_and_bow_matching_chain(env, tmp_path):
    """extract (device batch) -> transform_device -> node ids feed SearchByBoW: the relocalisation / reference-key-
    frame tracking chain (Tracking::TrackReferenceKeyFrame, src/Tracking.cc:755-771) with nothing leaving the GPU
    between extraction and the BoW conversion."""
    import torch
    pkg, O = env
    from test_matcher_gpu import _views
    voc = make_vocabulary(10, 4, seed=5)
    path = write_vocabulary(tmp_path / "voc.txt", voc)
    gv = pkg.ORBVocabulary()
    assert gv.loadFromTextFile(path)
    ov = O.OracleVocabulary(path)
    W, H, B = 752, 480, 3
    imgs = [synth_frame(4, W, H), synth_frame(4, W, H, shift_xy=(5, 2)), synth_frame(8, W, H)]
    ext = pkg.ORBextractor(1500, 1.2, 8, 20, 7)
    cap = ext.capacity(H, W)
    dev = torch.device("cuda:0")
    d_img = torch.from_numpy(np.stack(imgs)).to(dev)
    d_kps = torch.zeros((B, cap, 28), dtype=torch.uint8, device=dev)
    d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    d_n = torch.zeros(B, dtype=torch.int32, device=dev)
    gv.set_stream(ext.stream())          # one in-order queue: extraction, then the BoW conversion, no host sync between
    ext.extract_batch_device(d_img.data_ptr(), B, H, W, d_kps.data_ptr(), d_desc.data_ptr(), cap, d_n.data_ptr())
    d_word = torch.zeros((B, cap), dtype=torch.int32, device=dev)
    d_node = torch.zeros((B, cap), dtype=torch.int32, device=dev)
    d_bid = torch.zeros((B, cap), dtype=torch.int32, device=dev)
    d_w = torch.zeros((B, cap), dtype=torch.float64, device=dev)
    d_bv = torch.zeros((B, cap), dtype=torch.float64, device=dev)
    d_nb = torch.zeros(B, dtype=torch.int32, device=dev)
    gv.transform_device(B, d_desc.data_ptr(), d_n.data_ptr(), cap, 4, d_word.data_ptr(), d_w.data_ptr(), d_node.data_ptr(),
                        d_bid.data_ptr(), d_bv.data_ptr(), d_nb.data_ptr())
    # ... and the BoW-guided matching of (frame 0 -> 1) and (frame 1 -> 2), still in the same queue
    mdev = pkg.ORBmatcher(0.7, True)
    mdev.set_stream(ext.stream())
    d_m12 = torch.full((2, cap), -7, dtype=torch.int32, device=dev)
    d_nm = torch.zeros(2, dtype=torch.int32, device=dev)
    side = (d_kps.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), d_node.data_ptr())
    mdev.SearchByBoWDevice(2, cap, side, 0, 1, side, 1, 1, d_m12.data_ptr(), d_nm.data_ptr(), 50)
    gv.sync()
    gv.set_stream(0)
    mdev.set_stream(0)
    n = d_n.cpu().numpy()
    nb = d_nb.cpu().numpy()
    frames = []
    for b in range(B):
        desc = d_desc[b, :n[b]].cpu().numpy()
        kps = d_kps[b, :n[b]].cpu().numpy().view(pkg.KP_DTYPE).reshape(-1)
        o = ov.transform(desc, 4)
        assert np.array_equal(d_word[b, :n[b]].cpu().numpy().view(np.uint32), o["word_id"])
        assert np.array_equal(d_node[b, :n[b]].cpu().numpy().view(np.uint32), o["node_id"])
        assert np.array_equal(d_w[b, :n[b]].cpu().numpy(), o["word_weight"])
        assert nb[b] == len(o["bow_ids"])
        assert np.array_equal(d_bid[b, :nb[b]].cpu().numpy().view(np.uint32), o["bow_ids"])
        assert np.array_equal(d_bv[b, :nb[b]].cpu().numpy(), o["bow_vals"])
        frames.append((kps, desc, o["node_id"]))
    sf = ext.GetScaleFactors()
    (k1, d1, n1), (k2, d2, n2) = frames[0], frames[1]
    g1, o1, keep1 = _views(pkg, O, imgs[0], k1, d1, sf)
    g2, o2, keep2 = _views(pkg, O, imgs[1], k2, d2, sf)
    m = pkg.ORBmatcher(0.7, True)
    nm, m12 = m.SearchByBoW(g1, n1, None, g2, n2, None, 50)
    on, om12 = O.search_by_bow(o1, n1, None, o2, n2, None, 50, 0.7, True)
    assert nm == on and np.array_equal(m12, om12)
    assert nm > 100                       # shifted copy of the same scene: most features land in the same node
    # device-resident batched form: pair 0 = frames (0, 1), pair 1 = frames (1, 2)
    assert int(d_nm[0]) == on and np.array_equal(d_m12[0, :len(k1)].cpu().numpy(), om12)
    assert (d_m12[0, len(k1):].cpu().numpy() == -1).all()
    (k3, d3, n3) = frames[2]
    g3, o3, keep3 = _views(pkg, O, imgs[2], k3, d3, sf)
    on2, om12b = O.search_by_bow(o2, n2, None, o3, n3, None, 50, 0.7, True)
    assert int(d_nm[1]) == on2 and np.array_equal(d_m12[1, :len(k2)].cpu().numpy(), om12b)
    fv = pkg.ORBVocabulary.feature_vector(n1)
    assert sum(len(v) for v in fv.values()) == int((n1 != pkg.capi.NO_NODE).sum())


def test_bow_golden_fixtures(env, tmp_path):
    """Committed oracle outputs (tests/golden/bow_golden.npz, made by tests/golden/make_golden.py): the GPU path must
    reproduce the vocabulary transform, both BoW-guided searches and the distinctive-descriptor choice WITHOUT the
    oracle (descriptors come from the GPU extractor, which equals the oracle's on these frames)."""
    import os
    pkg, _ = env
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "bow_golden.npz"))
    W, H, nf = 640, 480, 800
    ext = pkg.ORBextractor(nf, 1.2, 8, 20, 7)
    img1, img2 = synth_frame(21, W, H), synth_frame(21, W, H, shift_xy=(4, 0))
    k1, d1 = ext(img1)
    k2, d2 = ext(img2)
    sf = ext.GetScaleFactors()
    voc = pkg.ORBVocabulary()
    assert voc.loadFromTextFile(write_vocabulary(tmp_path / "voc.txt", make_vocabulary(8, 3, seed=31)))
    r1, r2 = voc.transform(d1, 2), voc.transform(d2, 2)
    for key in ("word_id", "word_weight", "node_id", "bow_ids", "bow_vals"):
        assert np.array_equal(r1[key], g["t1_" + key]) and np.array_equal(r2[key], g["t2_" + key]), key
    F1 = pkg.FrameView(k1, d1, sf, frame_bounds(img1))
    F2 = pkg.FrameView(k2, d2, sf, frame_bounds(img2))
    n, m12 = pkg.ORBmatcher(0.7, True).SearchByBoW(F1, r1["node_id"], None, F2, r2["node_id"], None, 50)
    assert n == int(g["bow_n"]) and np.array_equal(m12, g["bow_m12"])
    F12 = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32)
    n, m12 = pkg.ORBmatcher(0.6, True).SearchForTriangulation(F1, r1["node_id"], None, F2, r2["node_id"], None, F12,
                                                              (200.0, 150.0), (sf * sf).astype(np.float32))
    assert n == int(g["tri_n"]) and np.array_equal(m12, g["tri_m12"])
    groups = [d1[i:i + 2 + (i % 9)] for i in range(0, 300, 11)]
    assert np.array_equal(pkg.ORBmatcher().ComputeDistinctiveDescriptors(groups), g["distinct"])


def test_full_size_vocabulary_properties(env):
    """ORBvoc-sized tree (k = 10, L = 6: 1,111,110 nodes, 10^6 words) -- properties that need no oracle:
    a leaf's own descriptor almost always descends to that leaf, FeatureVector nodes are the level-2 ancestors of the words, the L1-normalised BowVector
    sums to 1 and equals the numpy recomputation from the per-feature words."""
    pkg, O = env
    from orb_slam2_comment_amd.synth import synth_vocabulary
    parent, leaf, desc, weight = synth_vocabulary(10, 6, 3)
    voc = pkg.ORBVocabulary.from_arrays(10, 6, 0, 0, parent, leaf, desc, weight)
    assert voc.size() == 10 ** 6 and voc.getDepthLevels() == 6
    rng = np.random.default_rng(0)
    first_leaf = len(leaf) - 10 ** 6
    pick = rng.integers(0, 10 ** 6, 4000)
    r = voc.transform(desc[first_leaf + pick], 4)
    # node ids in file order are 1-based: word w is node first_leaf + w + 1
    got_nodes = first_leaf + r["word_id"].astype(np.int64) + 1
    same = np.all(desc[got_nodes - 1] == desc[first_leaf + pick], axis=1)
    assert same.mean() > 0.97                             # the greedy descent almost always finds the leaf's own path
    assert (r["word_id"] == pick).mean() > 0.95           # (ties between duplicate siblings go to the first one)
    # whatever leaf was reached, it is at least as close as the feature's own leaf is to its siblings' best
    dist = np.unpackbits(desc[got_nodes - 1] ^ desc[first_leaf + pick], axis=1).sum(1)
    assert dist.max() <= 64
    anc = got_nodes.copy()
    for _ in range(4):
        anc = parent[anc - 1]
    assert np.array_equal(r["node_id"].astype(np.int64), anc)
    assert np.array_equal(r["word_weight"], weight[got_nodes - 1])
    assert abs(r["bow_vals"].sum() - 1.0) < 1e-9 and (np.diff(r["bow_ids"].astype(np.int64)) > 0).all()
    ids, inv = np.unique(r["word_id"], return_inverse=True)
    assert np.array_equal(ids, r["bow_ids"])
    raw = np.zeros(len(ids)); np.add.at(raw, inv, r["word_weight"])
    assert np.allclose(r["bow_vals"], raw / raw.sum(), rtol=1e-12)
