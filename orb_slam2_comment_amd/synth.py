"""Deterministic synthetic camera frames.

No KITTI/EuRoC data exists in this environment, so every benchmark and parity
test uses `synth_frame(seed, W, H)`: a smooth random background, filled
rectangles and discs (corner-rich, KITTI-like FAST candidate density) and +-3
uniform noise.  `synth_stereo` renders the same layered scene from a second
eye with a per-shape horizontal disparity, so uL - uR = d for every shape.
"""
import numpy as np


def _scene(seed, W, H, n_rect, n_disc, n_small=None):
    rng = np.random.default_rng(seed)
    if n_small is None:  # fine texture (foliage/gravel stand-in), scaled with the area
        n_small = int(3000 * (W * H) / (1241.0 * 376.0))
    gy, gx = 9, 17
    grid = rng.uniform(40, 215, size=(gy, gx))
    shapes = []
    for _ in range(n_rect):
        w, h = rng.integers(8, 121, size=2)
        x0 = rng.integers(-w // 2, W - w // 2)
        y0 = rng.integers(-h // 2, H - h // 2)
        shapes.append(("r", int(x0), int(y0), int(w), int(h), int(rng.integers(0, 256)),
                       int(rng.integers(2, 81))))
    for _ in range(n_disc):
        r = rng.integers(3, 31)
        cx = rng.integers(0, W)
        cy = rng.integers(0, H)
        shapes.append(("d", int(cx), int(cy), int(r), 0, int(rng.integers(0, 256)),
                       int(rng.integers(2, 81))))
    for _ in range(n_small):
        w, h = rng.integers(2, 11, size=2)
        x0 = rng.integers(0, W)
        y0 = rng.integers(0, H)
        shapes.append(("r", int(x0), int(y0), int(w), int(h), int(rng.integers(0, 256)),
                       int(rng.integers(2, 81))))
    # paint far-to-near so that larger disparity (nearer) occludes
    shapes.sort(key=lambda s: s[6])
    return grid, shapes


def _background(grid, W, H):
    gy, gx = grid.shape
    ys = np.linspace(0, gy - 1, H)
    xs = np.linspace(0, gx - 1, W)
    y0 = np.clip(np.floor(ys).astype(int), 0, gy - 2)
    x0 = np.clip(np.floor(xs).astype(int), 0, gx - 2)
    fy = (ys - y0)[:, None]
    fx = (xs - x0)[None, :]
    g = grid
    return ((1 - fy) * (1 - fx) * g[y0][:, x0] + (1 - fy) * fx * g[y0][:, x0 + 1]
            + fy * (1 - fx) * g[y0 + 1][:, x0] + fy * fx * g[y0 + 1][:, x0 + 1])


def _render(grid, shapes, W, H, eye_shift, noise_seed, shift_xy=(0, 0)):
    img = _background(grid, W, H)
    sx, sy = shift_xy
    for kind, a, b, c, d, gray, disp in shapes:
        off = -disp if eye_shift else 0
        if kind == "r":
            x0, y0, w, h = a + off + sx, b + sy, c, d
            xa, xb = max(x0, 0), min(x0 + w, W)
            ya, yb = max(y0, 0), min(y0 + h, H)
            if xa < xb and ya < yb:
                img[ya:yb, xa:xb] = gray
        else:
            cx, cy, r = a + off + sx, b + sy, c
            xa, xb = max(cx - r, 0), min(cx + r + 1, W)
            ya, yb = max(cy - r, 0), min(cy + r + 1, H)
            if xa < xb and ya < yb:
                yy, xx = np.ogrid[ya:yb, xa:xb]
                m = (yy - cy) ** 2 + (xx - cx) ** 2 <= r * r
                img[ya:yb, xa:xb][m] = gray
    rng = np.random.default_rng(noise_seed)
    img = img + rng.integers(-3, 4, size=(H, W))
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def synth_frame(seed, W=1241, H=376, n_rect=400, n_disc=200, shift_xy=(0, 0)):
    """uint8 [H, W] frame; `shift_xy` translates the shapes (frame t -> t+1)."""
    grid, shapes = _scene(seed, W, H, n_rect, n_disc)
    return _render(grid, shapes, W, H, False, (seed << 8) + 1 + 7 * shift_xy[0] + 13 * shift_xy[1],
                   shift_xy)


def synth_stereo(seed, W=1241, H=376, n_rect=400, n_disc=200, shift_xy=(0, 0)):
    """(left, right) uint8 frames of one layered scene; disparities in [2, 80] px.  `shift_xy` translates the shapes in
    both eyes (stereo frame t -> t+1)."""
    grid, shapes = _scene(seed, W, H, n_rect, n_disc)
    ns = 7 * shift_xy[0] + 13 * shift_xy[1]
    left = _render(grid, shapes, W, H, False, (seed << 8) + 1 + ns, shift_xy)
    right = _render(grid, shapes, W, H, True, (seed << 8) + 2 + ns, shift_xy)
    return left, right


def synth_batch(first_seed, count, W=1241, H=376):
    """uint8 [count, H, W]"""
    return np.stack([synth_frame(first_seed + i, W, H) for i in range(count)])


def synth_vocabulary(k=10, L=6, seed=1):
    """Regular k-ary vocabulary tree with the shape of ORBvoc.txt (k=10, L=6: 1.1 M nodes, 10^6 words) in
    saveToTextFile (BFS) order, for benchmarks: the real vocabulary is not part of the reference checkout.
    Children are their parent with a few random bit flips.  Returns (parent, is_leaf, desc, weight), entry i = node i+1,
    ready for ORBVocabulary.from_arrays."""
    rng = np.random.default_rng(seed)
    parent, leaf, desc = [], [], []
    prev_desc, prev_first = rng.integers(0, 256, (1, 32), dtype=np.uint8), 0
    nid = 1
    for depth in range(1, L + 1):
        n = prev_desc.shape[0] * k
        d = np.repeat(prev_desc, k, axis=0)
        for _ in range(max(2, 48 >> depth)):
            bit = rng.integers(0, 256, n)
            d[np.arange(n), bit >> 3] ^= (1 << (bit & 7)).astype(np.uint8)
        parent.append(np.repeat(np.arange(prev_first, prev_first + prev_desc.shape[0], dtype=np.int32), k))
        leaf.append(np.full(n, depth == L, np.uint8))
        desc.append(d)
        prev_desc, prev_first = d, nid
        nid += n
    parent, leaf, desc = np.concatenate(parent), np.concatenate(leaf), np.concatenate(desc)
    weight = np.where(leaf == 1, rng.uniform(0.5, 12.0, len(leaf)), 0.0)
    return parent, leaf, desc, weight
