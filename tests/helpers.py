"""Shared helpers for the parity tests (GPU path vs oracle on the same inputs)."""
import numpy as np

from orb_slam2_comment_amd.synth import synth_frame, synth_stereo  # noqa: F401


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def assert_kps_equal(a, b, what=""):
    assert len(a) == len(b), "%s count %d vs %d" % (what, len(a), len(b))
    for f in a.dtype.names:
        assert np.array_equal(a[f], b[f]), "%s field %s differs at %s" % (what, f, np.nonzero(a[f] != b[f])[0][:5])


def assert_stagewise_equal(ext, ora, nlevels, what=""):
    """pyramid (with 19-px border), FAST candidates, blurred levels -- bit exact."""
    for l in range(nlevels):
        gp, op = ext.image_pyramid(l, with_border=True), ora.level_padded(l)
        assert gp.shape == op.shape and np.array_equal(gp, op), "%s pyramid level %d" % (what, l)
        gx, gy, gs = ext.level_candidates(l)
        ox, oy, orr = ora.level_candidates(l)
        assert len(gx) == len(ox), "%s candidates level %d: %d vs %d" % (what, l, len(gx), len(ox))
        assert np.array_equal(gx, ox.astype(np.int32)) and np.array_equal(gy, oy.astype(np.int32)) \
            and np.array_equal(gs, orr.astype(np.int32)), "%s candidates level %d" % (what, l)
        ob = ora.level_blurred(l)
        if ob is not None:
            assert np.array_equal(ext.blurred_level(l), ob), "%s blurred level %d" % (what, l)


def frame_bounds(img):
    """Frame::ComputeImageBounds without distortion (src/Frame.cc:458-463)."""
    return (0.0, 0.0, float(img.shape[1]), float(img.shape[0]))


def make_vocabulary(k, L, seed, scoring=0, weighting=0, irregular=False, order="bfs", stop_frac=0.05, dup_frac=0.1,
                    flip_bits=40):
    """Synthetic DBoW2 vocabulary tree (the real ORBvoc.txt is not part of the reference checkout).  Children are
    bit-flipped copies of their parent (a crude hierarchical clustering), siblings are sometimes exact duplicates
    (distance ties) and some words have weight 0 (stopped words).  Returns arrays in FILE order: entry i = node i+1;
    parent[i], is_leaf[i], desc[i, 32], weight[i].  order="bfs" is what saveToTextFile writes (children of a node
    contiguous); "interleaved" scatters siblings so that child lists are not contiguous in id space."""
    rng = np.random.default_rng(seed)
    nodes = [dict(parent=-1, depth=0, desc=rng.integers(0, 256, 32, dtype=np.uint8), children=[])]
    frontier = [0]
    while frontier:
        nxt = []
        for p in frontier:
            depth = nodes[p]["depth"]
            if depth >= L:
                continue
            if irregular and depth > 0 and rng.random() < 0.15:
                continue                                       # early leaf
            nc = k if not irregular else int(rng.integers(1, k + 1))
            for c in range(nc):
                d = nodes[p]["desc"].copy()
                if c > 0 and rng.random() < dup_frac:
                    d = nodes[nodes[p]["children"][-1]]["desc"].copy()
                else:
                    bits = rng.integers(0, 256, max(1, flip_bits >> depth))
                    for b in bits:
                        d[b >> 3] ^= np.uint8(1 << (b & 7))
                nodes.append(dict(parent=p, depth=depth + 1, desc=d, children=[]))
                nodes[p]["children"].append(len(nodes) - 1)
                nxt.append(len(nodes) - 1)
        frontier = nxt
    # file order
    if order == "bfs":
        seq, queue = [], [0]
        while queue:
            p = queue.pop(0)
            for c in nodes[p]["children"]:
                seq.append(c)
                queue.append(c)
    else:
        seq, level = [], [0]
        while level:
            lists = [list(nodes[p]["children"]) for p in level]
            nxt = []
            while any(lists):
                for lst in lists:
                    if lst:
                        c = lst.pop(0)
                        seq.append(c)
                        nxt.append(c)
            level = nxt
    new_id = {0: 0}
    for i, old in enumerate(seq):
        new_id[old] = i + 1
    parent = np.array([new_id[nodes[o]["parent"]] for o in seq], np.int32)
    is_leaf = np.array([0 if nodes[o]["children"] else 1 for o in seq], np.uint8)
    desc = np.stack([nodes[o]["desc"] for o in seq]).astype(np.uint8)
    weight = np.where(is_leaf == 1, rng.uniform(0.05, 12.0, len(seq)), 0.0)
    weight[(is_leaf == 1) & (rng.random(len(seq)) < stop_frac)] = 0.0
    return dict(k=k, L=L, scoring=scoring, weighting=weighting, parent=parent, is_leaf=is_leaf, desc=desc, weight=weight)


def write_vocabulary(path, voc, trailing_newline=True):
    """TemplatedVocabulary::saveToTextFile layout (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1429-1466)."""
    with open(path, "w") as f:
        f.write("%d %d %d %d\n" % (voc["k"], voc["L"], voc["scoring"], voc["weighting"]))
        lines = []
        for i in range(len(voc["parent"])):
            lines.append("%d %d %s %s" % (voc["parent"][i], voc["is_leaf"][i], " ".join(str(int(b)) for b in voc["desc"][i]),
                                          repr(float(voc["weight"][i]))))
        f.write("\n".join(lines))
        if trailing_newline:
            f.write("\n")
    return path
