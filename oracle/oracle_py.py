"""ctypes binding of the CPU oracle (oracle/liborb_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28

QUERY_DTYPE = np.dtype([("valid", "<i4"), ("u", "<f4"), ("v", "<f4"), ("radius", "<f4"),
                        ("min_level", "<i4"), ("max_level", "<i4"), ("ur", "<f4"),
                        ("level_aux", "<i4"), ("angle", "<f4"), ("observed", "<i4")])
assert QUERY_DTYPE.itemsize == 40


class Frame(C.Structure):
    _fields_ = [("n", C.c_int32), ("keys", C.c_void_p), ("desc", C.c_void_p),
                ("u_right", C.c_void_p), ("min_x", C.c_float), ("min_y", C.c_float),
                ("max_x", C.c_float), ("max_y", C.c_float), ("grid_inv_w", C.c_float),
                ("grid_inv_h", C.c_float), ("n_levels", C.c_int32),
                ("scale_factors", C.c_void_p)]


class Camera(C.Structure):
    _fields_ = [("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float), ("mbf", C.c_float),
                ("mb", C.c_float), ("min_x", C.c_float), ("max_x", C.c_float), ("min_y", C.c_float),
                ("max_y", C.c_float), ("n_levels", C.c_int32), ("log_scale_factor", C.c_float),
                ("scale_factors", C.c_float * 8)]


class Pyramids(C.Structure):
    _fields_ = [("n_levels", C.c_int32), ("left", C.c_void_p), ("right", C.c_void_p),
                ("step_left", C.c_void_p), ("step_right", C.c_void_p),
                ("cols_right", C.c_void_p), ("scale_factors", C.c_void_p),
                ("inv_scale_factors", C.c_void_p)]


def build():
    """Compile the oracle with gcc (idempotent)."""
    subprocess.run(["make", "-s", "-C", _HERE], check=True)


_SO_OVERRIDE = None


def use_library(path):
    """bench.py's cpu_baseline leg: load a build of the same source made for the host it runs on."""
    global _LIB, _SO_OVERRIDE
    _LIB, _SO_OVERRIDE = None, path


def lib():
    global _LIB
    if _LIB is None:
        so = _SO_OVERRIDE or os.path.join(_HERE, "liborb_oracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        L.oracle_create.restype = C.c_void_p
        L.oracle_create.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int]
        L.oracle_destroy.argtypes = [C.c_void_p]
        L.oracle_get_tables.argtypes = [C.c_void_p] + [C.c_void_p] * 6
        L.oracle_extract.restype = C.c_int
        L.oracle_extract.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                     C.c_void_p, C.c_void_p, C.c_int]
        L.oracle_level_size.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.oracle_level_padded.restype = C.c_void_p
        L.oracle_level_padded.argtypes = [C.c_void_p, C.c_int]
        L.oracle_level_blurred.restype = C.c_void_p
        L.oracle_level_blurred.argtypes = [C.c_void_p, C.c_int]
        L.oracle_level_candidates.argtypes = [C.c_void_p, C.c_int] + [C.POINTER(C.c_void_p)] * 3
        L.oracle_level_keypoints.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
        L.oracle_cvround.argtypes = [C.c_double]
        L.oracle_fast_atan2.restype = C.c_float
        L.oracle_fast_atan2.argtypes = [C.c_float, C.c_float]
        L.oracle_det_sincos.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.oracle_resize_linear.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                           C.c_int, C.c_int, C.c_int]
        L.oracle_gauss7.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.oracle_fast.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_distribute_octree.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                               C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                               C.c_void_p, C.c_int]
        L.oracle_descriptor_distance.argtypes = [C.c_void_p, C.c_void_p]
        L.oracle_three_maxima.argtypes = [C.c_void_p, C.c_int] + [C.POINTER(C.c_int)] * 3
        L.oracle_features_in_area.argtypes = [C.POINTER(Frame), C.c_float, C.c_float, C.c_float,
                                              C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.oracle_search_for_initialization.argtypes = [C.POINTER(Frame), C.POINTER(Frame),
                                                       C.c_void_p, C.c_void_p, C.c_int,
                                                       C.c_float, C.c_int]
        L.oracle_det_logf.restype = C.c_float
        L.oracle_det_logf.argtypes = [C.c_float]
        L.oracle_project_last_frame.restype = None
        L.oracle_project_last_frame.argtypes = [C.POINTER(Camera), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                                C.c_void_p, C.c_float, C.c_int, C.c_void_p]
        L.oracle_frustum_queries.restype = None
        L.oracle_frustum_queries.argtypes = [C.POINTER(Camera), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        L.oracle_keyframe_queries.restype = None
        L.oracle_keyframe_queries.argtypes = [C.POINTER(Camera), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
        L.oracle_search_by_sim3.restype = C.c_int
        L.oracle_search_by_sim3.argtypes = [C.POINTER(Frame), C.POINTER(Frame), C.POINTER(Camera)] + [C.c_void_p] * 14 + \
            [C.c_float, C.c_void_p]
        L.oracle_search_by_projection_frame.argtypes = [C.POINTER(Frame), C.c_void_p, C.c_void_p,
                                                        C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.oracle_search_by_projection_block.argtypes = [C.POINTER(Frame), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                                        C.c_void_p, C.c_int, C.c_int]
        L.oracle_search_best_in_window.argtypes = [C.POINTER(Frame), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                                   C.c_void_p, C.c_void_p]
        L.oracle_search_by_projection_points.argtypes = [C.POINTER(Frame), C.c_void_p, C.c_void_p,
                                                         C.c_int, C.c_void_p, C.c_void_p, C.c_float]
        L.oracle_search_by_bow.argtypes = [C.POINTER(Frame), C.c_void_p, C.c_void_p, C.POINTER(Frame), C.c_void_p,
                                           C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p]
        L.oracle_search_for_triangulation.argtypes = [C.POINTER(Frame), C.c_void_p, C.c_void_p, C.POINTER(Frame),
                                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float,
                                                      C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.oracle_vocabulary_load_text.restype = C.c_void_p
        L.oracle_vocabulary_load_text.argtypes = [C.c_char_p]
        L.oracle_vocabulary_destroy.argtypes = [C.c_void_p]
        L.oracle_vocabulary_info.argtypes = [C.c_void_p] + [C.POINTER(C.c_int)] * 6
        L.oracle_vocabulary_transform.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                                  C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_distinctive_descriptor.argtypes = [C.c_void_p, C.c_int]
        L.oracle_assign_features_to_grid.argtypes = [C.POINTER(Frame), C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_compute_stereo_from_rgbd.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_float,
                                                      C.c_void_p, C.c_void_p]
        L.oracle_undistort_keypoints.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float,
                                                 C.c_void_p, C.c_void_p]
        L.oracle_compute_stereo_matches.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                                    C.c_void_p, C.c_int, C.POINTER(Pyramids),
                                                    C.c_int, C.c_float, C.c_float, C.c_void_p,
                                                    C.c_void_p]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class OracleExtractor:
    """Mirror of ORBextractor (include/ORBextractor.h:45-110) over the C oracle."""

    def __init__(self, nfeatures=1000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7):
        self.L = lib()
        self.h = self.L.oracle_create(nfeatures, scale_factor, nlevels, ini_th, min_th)
        if not self.h:
            raise ValueError("oracle_create failed")
        self.nfeatures, self.nlevels = nfeatures, nlevels

    def close(self):
        if self.h:
            self.L.oracle_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def tables(self):
        n = self.nlevels
        sf, isf, s2, is2 = (np.zeros(n, np.float32) for _ in range(4))
        nf = np.zeros(n, np.int32)
        um = np.zeros(16, np.int32)
        self.L.oracle_get_tables(self.h, _p(sf), _p(isf), _p(s2), _p(is2), _p(nf), _p(um))
        return dict(scale=sf, inv_scale=isf, sigma2=s2, inv_sigma2=is2, feat=nf, umax=um)

    def extract(self, img):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        cap = 2 * self.nfeatures + 64 * self.nlevels + 256    # >= sum over levels of max(quota + 3, 4 * nIni)
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = self.L.oracle_extract(self.h, _p(img), img.shape[0], img.shape[1], img.strides[0],
                                  _p(kps), _p(desc), cap)
        if n < 0:
            raise RuntimeError("oracle capacity")
        return kps[:n].copy(), desc[:n].copy()

    def level_size(self, level):
        w, h = C.c_int(), C.c_int()
        if self.L.oracle_level_size(self.h, level, C.byref(w), C.byref(h)) != 0:
            raise IndexError(level)
        return w.value, h.value

    def level_padded(self, level):
        w, h = self.level_size(level)
        p = self.L.oracle_level_padded(self.h, level)
        buf = (C.c_uint8 * ((w + 38) * (h + 38))).from_address(p)
        return np.frombuffer(buf, np.uint8).reshape(h + 38, w + 38).copy()

    def level_blurred(self, level):
        w, h = self.level_size(level)
        p = self.L.oracle_level_blurred(self.h, level)
        if not p:
            return None
        buf = (C.c_uint8 * (w * h)).from_address(p)
        return np.frombuffer(buf, np.uint8).reshape(h, w).copy()

    def level_candidates(self, level):
        px, py, pr = C.c_void_p(), C.c_void_p(), C.c_void_p()
        n = self.L.oracle_level_candidates(self.h, level, C.byref(px), C.byref(py), C.byref(pr))
        if n == 0:
            z = np.zeros(0, np.float32)
            return z, z, z

        def arr(p):
            return np.frombuffer((C.c_float * n).from_address(p.value), np.float32).copy()
        return arr(px), arr(py), arr(pr)

    def level_keypoints(self, level):
        p = C.c_void_p()
        n = self.L.oracle_level_keypoints(self.h, level, C.byref(p))
        if n == 0:
            return np.zeros(0, KP_DTYPE)
        return np.frombuffer((C.c_uint8 * (n * 28)).from_address(p.value), KP_DTYPE).copy()


def make_frame(keys, desc, u_right, bounds, scale_factors, keep):
    """Build a Frame view; `keep` collects the arrays so they outlive the struct.
    bounds = (min_x, min_y, max_x, max_y); the grid is 64x48 (include/Frame.h:37-38)."""
    keys = np.ascontiguousarray(keys, KP_DTYPE)
    desc = np.ascontiguousarray(desc, np.uint8)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    ur = None if u_right is None else np.ascontiguousarray(u_right, np.float32)
    keep += [keys, desc, sf, ur]
    f = Frame()
    f.n = len(keys)
    f.keys, f.desc, f.u_right = _p(keys), _p(desc), _p(ur)
    f.min_x, f.min_y, f.max_x, f.max_y = [np.float32(b) for b in bounds]
    # Frame.cc:101-102: static_cast<float>(FRAME_GRID_COLS)/static_cast<float>(mnMaxX-mnMinX)
    f.grid_inv_w = np.float32(64.0) / (np.float32(bounds[2]) - np.float32(bounds[0]))
    f.grid_inv_h = np.float32(48.0) / (np.float32(bounds[3]) - np.float32(bounds[1]))
    f.n_levels = len(sf)
    f.scale_factors = _p(sf)
    return f


def features_in_area(frame, x, y, r, min_level=-1, max_level=-1):
    out = np.zeros(max(frame.n, 1), np.int32)
    n = lib().oracle_features_in_area(C.byref(frame), x, y, r, min_level, max_level, _p(out), len(out))
    return out[:n].copy()


def descriptor_distance(a, b):
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    return lib().oracle_descriptor_distance(_p(a), _p(b))


def search_for_initialization(f1, f2, prev_matched, window_size=100, nnratio=0.9, check_ori=True):
    pm = np.ascontiguousarray(prev_matched, np.float32).copy()
    m12 = np.zeros(max(f1.n, 1), np.int32)
    n = lib().oracle_search_for_initialization(C.byref(f1), C.byref(f2), _p(pm), _p(m12),
                                               window_size, nnratio, int(check_ori))
    return n, m12[:f1.n].copy(), pm


def search_by_projection_frame(cur, queries, qdesc, taken=None, check_ori=True):
    q = np.ascontiguousarray(queries, QUERY_DTYPE)
    qd = np.ascontiguousarray(qdesc, np.uint8)
    tk = None if taken is None else np.ascontiguousarray(taken, np.uint8)
    out = np.zeros(max(cur.n, 1), np.int32)
    n = lib().oracle_search_by_projection_frame(C.byref(cur), _p(q), _p(qd), len(q), _p(tk),
                                                _p(out), int(check_ori))
    return n, out[:cur.n].copy()


def camera_from(cam):
    """oracle_camera with the values of an orbhip Camera (same leading layout, 8 scale factors)."""
    o = Camera()
    for f, _ in Camera._fields_[:-1]:
        setattr(o, f, getattr(cam, f))
    for i in range(8):
        o.scale_factors[i] = cam.scale_factors[i]
    return o


def det_logf(x):
    return float(lib().oracle_det_logf(float(x)))


def project_last_frame(cam, Tcw, Tlw, world, flags, last_keys, th, mono):
    Tc = np.ascontiguousarray(np.asarray(Tcw, np.float32)[:3, :4])
    Tl = np.ascontiguousarray(np.asarray(Tlw, np.float32)[:3, :4])
    world = np.ascontiguousarray(world, np.float32).reshape(-1, 3)
    flags = np.ascontiguousarray(flags, np.uint8)
    keys = np.ascontiguousarray(last_keys, KP_DTYPE)
    q = np.zeros(len(keys), QUERY_DTYPE)
    oc = camera_from(cam)
    lib().oracle_project_last_frame(C.byref(oc), _p(Tc), _p(Tl), len(keys), _p(world), _p(flags), _p(keys), float(th),
                                    int(mono), _p(q))
    return q


def frustum_queries(cam, Tcw, world, normal, max_dist, min_dist, flags, viewing_cos_limit, th):
    Tc = np.ascontiguousarray(np.asarray(Tcw, np.float32)[:3, :4])
    world = np.ascontiguousarray(world, np.float32).reshape(-1, 3)
    normal = np.ascontiguousarray(normal, np.float32).reshape(-1, 3)
    mx, mn = np.ascontiguousarray(max_dist, np.float32), np.ascontiguousarray(min_dist, np.float32)
    flags = np.ascontiguousarray(flags, np.uint8)
    q, vc = np.zeros(len(world), QUERY_DTYPE), np.zeros(len(world), np.float32)
    oc = camera_from(cam)
    lib().oracle_frustum_queries(C.byref(oc), _p(Tc), len(world), _p(world), _p(normal), _p(mx), _p(mn), _p(flags),
                                 float(viewing_cos_limit), float(th), _p(q), _p(vc))
    return q, vc


def keyframe_queries(cam, mode, double_invz, T1, T2, world, normal, max_dist, min_dist, flags, th):
    T1 = np.ascontiguousarray(np.asarray(T1, np.float32)[:3, :4])
    T2 = None if T2 is None else np.ascontiguousarray(np.asarray(T2, np.float32)[:3, :4])
    world = np.ascontiguousarray(world, np.float32).reshape(-1, 3)
    normal = None if normal is None else np.ascontiguousarray(normal, np.float32).reshape(-1, 3)
    mx, mn = np.ascontiguousarray(max_dist, np.float32), np.ascontiguousarray(min_dist, np.float32)
    flags = np.ascontiguousarray(flags, np.uint8)
    q = np.zeros(len(world), QUERY_DTYPE)
    oc = camera_from(cam)
    lib().oracle_keyframe_queries(C.byref(oc), int(mode), int(double_invz), _p(T1), _p(T2), len(world), _p(world), _p(normal),
                                  _p(mx), _p(mn), _p(flags), float(th), _p(q))
    return q


def search_by_sim3(kf1, kf2, cam, T1w, T2w, S21, S12, pts1, pts2, th):
    mats = [np.ascontiguousarray(np.asarray(T, np.float32)[:3, :4]) for T in (T1w, T2w, S21, S12)]
    a = []
    for p in (pts1, pts2):
        a.append((np.ascontiguousarray(p[0], np.float32).reshape(-1, 3), np.ascontiguousarray(p[1], np.float32),
                  np.ascontiguousarray(p[2], np.float32), np.ascontiguousarray(p[3], np.uint8),
                  np.ascontiguousarray(p[4], np.uint8).reshape(-1, 32)))
    m12 = np.full(max(kf1.n, 1), -1, np.int32)
    oc = camera_from(cam)
    n = lib().oracle_search_by_sim3(C.byref(kf1), C.byref(kf2), C.byref(oc), _p(mats[0]), _p(mats[1]), _p(mats[2]), _p(mats[3]),
                                    _p(a[0][0]), _p(a[0][1]), _p(a[0][2]), _p(a[0][3]), _p(a[0][4]),
                                    _p(a[1][0]), _p(a[1][1]), _p(a[1][2]), _p(a[1][3]), _p(a[1][4]), float(th), _p(m12))
    return n, m12[:kf1.n].copy()


def search_by_projection_block(cur, queries, qdesc, taken=None, max_dist=100, check_ori=True):
    q = np.ascontiguousarray(queries, QUERY_DTYPE)
    qd = np.ascontiguousarray(qdesc, np.uint8)
    tk = None if taken is None else np.ascontiguousarray(taken, np.uint8)
    out = np.zeros(max(cur.n, 1), np.int32)
    n = lib().oracle_search_by_projection_block(C.byref(cur), _p(q), _p(qd), len(q), _p(tk), _p(out), max_dist,
                                                int(check_ori))
    return n, out[:cur.n].copy()


def search_best_in_window(kf, queries, qdesc, inv_sigma2=None):
    q = np.ascontiguousarray(queries, QUERY_DTYPE)
    qd = np.ascontiguousarray(qdesc, np.uint8)
    sig = None if inv_sigma2 is None else np.ascontiguousarray(inv_sigma2, np.float32)
    bi = np.zeros(max(len(q), 1), np.int32)
    bd = np.zeros(max(len(q), 1), np.int32)
    lib().oracle_search_best_in_window(C.byref(kf), _p(q), _p(qd), len(q), _p(sig), _p(bi), _p(bd))
    return bi[:len(q)].copy(), bd[:len(q)].copy()


def search_by_projection_points(f, queries, qdesc, taken=None, nnratio=0.8):
    q = np.ascontiguousarray(queries, QUERY_DTYPE)
    qd = np.ascontiguousarray(qdesc, np.uint8)
    tk = None if taken is None else np.ascontiguousarray(taken, np.uint8)
    out = np.zeros(max(f.n, 1), np.int32)
    n = lib().oracle_search_by_projection_points(C.byref(f), _p(q), _p(qd), len(q), _p(tk),
                                                 _p(out), nnratio)
    return n, out[:f.n].copy()


NO_NODE = 0xFFFFFFFF


class OracleVocabulary:
    """DBoW2 text vocabulary + transform (TemplatedVocabulary.h:1338-1424, :1127-1262), CPU restatement."""

    def __init__(self, path):
        self._h = lib().oracle_vocabulary_load_text(str(path).encode())
        if not self._h:
            raise ValueError("not a DBoW2 text vocabulary: %s" % path)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().oracle_vocabulary_destroy(self._h)
            self._h = None

    def info(self):
        v = [C.c_int(0) for _ in range(6)]
        lib().oracle_vocabulary_info(self._h, *[C.byref(x) for x in v])
        return dict(zip(("k", "L", "scoring", "weighting", "n_nodes", "n_words"), [x.value for x in v]))

    def transform(self, descriptors, levelsup=4):
        d = np.ascontiguousarray(descriptors, np.uint8).reshape(-1, 32)
        n = len(d)
        word = np.zeros(max(n, 1), np.uint32)
        wgt = np.zeros(max(n, 1), np.float64)
        node = np.zeros(max(n, 1), np.uint32)
        bid = np.zeros(max(n, 1), np.uint32)
        bval = np.zeros(max(n, 1), np.float64)
        nb = lib().oracle_vocabulary_transform(self._h, _p(d), n, int(levelsup), _p(word), _p(wgt), _p(node), _p(bid),
                                               _p(bval))
        return {"word_id": word[:n].copy(), "word_weight": wgt[:n].copy(), "node_id": node[:n].copy(),
                "bow_ids": bid[:nb].copy(), "bow_vals": bval[:nb].copy()}



def search_by_bow(f1, node1, valid1, f2, node2, blocked2=None, max_dist=50, nnratio=0.7, check_ori=True):
    """ORBmatcher::SearchByBoW (both overloads); returns (nmatches, matches12[n1])."""
    n1a = np.ascontiguousarray(node1, np.uint32)
    n2a = np.ascontiguousarray(node2, np.uint32)
    v1 = None if valid1 is None else np.ascontiguousarray(valid1, np.uint8)
    b2 = None if blocked2 is None else np.ascontiguousarray(blocked2, np.uint8)
    m12 = np.zeros(max(f1.n, 1), np.int32)
    n = lib().oracle_search_by_bow(C.byref(f1), _p(n1a), _p(v1), C.byref(f2), _p(n2a), _p(b2), int(max_dist),
                                   nnratio, int(check_ori), _p(m12))
    return n, m12[:f1.n].copy()


def search_for_triangulation(f1, node1, valid1, f2, node2, valid2, F12, ex, ey, level_sigma2, only_stereo=False,
                             check_ori=True):
    """ORBmatcher::SearchForTriangulation; returns (nmatches, matches12[n1])."""
    n1a = np.ascontiguousarray(node1, np.uint32)
    n2a = np.ascontiguousarray(node2, np.uint32)
    v1 = None if valid1 is None else np.ascontiguousarray(valid1, np.uint8)
    v2 = None if valid2 is None else np.ascontiguousarray(valid2, np.uint8)
    F = np.ascontiguousarray(F12, np.float32).reshape(9)
    sg = np.ascontiguousarray(level_sigma2, np.float32)
    m12 = np.zeros(max(f1.n, 1), np.int32)
    n = lib().oracle_search_for_triangulation(C.byref(f1), _p(n1a), _p(v1), C.byref(f2), _p(n2a), _p(v2), _p(F),
                                              float(ex), float(ey), _p(sg), int(only_stereo), int(check_ori), _p(m12))
    return n, m12[:f1.n].copy()


def distinctive_descriptor(desc):
    """MapPoint::ComputeDistinctiveDescriptors for one map point; index of the chosen observation or -1."""
    d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
    return lib().oracle_distinctive_descriptor(_p(d), len(d))


def assign_features_to_grid(frame):
    """Frame::AssignFeaturesToGrid; returns (cell_of[n], cell_start[3073], cell_items[m])."""
    cell_of = np.zeros(max(frame.n, 1), np.int32)
    start = np.zeros(64 * 48 + 1, np.int32)
    items = np.zeros(max(frame.n, 1), np.int32)
    lib().oracle_assign_features_to_grid(C.byref(frame), _p(cell_of), _p(start), _p(items))
    return cell_of[:frame.n].copy(), start, items[:start[-1]].copy()


def compute_stereo_from_rgbd(keys, keys_un, depth, mbf):
    """Frame::ComputeStereoFromRGBD; depth: 2-D float32 image.  Returns (mvuRight, mvDepth)."""
    k = np.ascontiguousarray(keys, KP_DTYPE)
    ku = np.ascontiguousarray(keys_un, KP_DTYPE)
    d = np.ascontiguousarray(depth, np.float32)
    ur = np.zeros(max(len(k), 1), np.float32)
    dp = np.zeros(max(len(k), 1), np.float32)
    lib().oracle_compute_stereo_from_rgbd(_p(k), _p(ku), len(k), _p(d), d.shape[1], mbf, _p(ur), _p(dp))
    return ur[:len(k)].copy(), dp[:len(k)].copy()


def undistort_keypoints(keys, fx, fy, cx, cy, dist):
    """Frame::UndistortKeyPoints; dist = (k1, k2, p1, p2, k3).  Returns mvKeysUn."""
    k = np.ascontiguousarray(keys, KP_DTYPE)
    d = np.zeros(5, np.float32)
    d[:len(dist)] = dist
    out = np.zeros(max(len(k), 1), KP_DTYPE)
    lib().oracle_undistort_keypoints(_p(k), len(k), fx, fy, cx, cy, _p(d), _p(out))
    return out[:len(k)].copy()


def compute_stereo_matches(keys_l, desc_l, keys_r, desc_r, levels_l, levels_r, scale, inv_scale,
                           mbf, mb):
    """levels_l/levels_r: lists of 2-D uint8 arrays (level ROIs, any row stride)."""
    nl = len(levels_l)
    keys_l = np.ascontiguousarray(keys_l, KP_DTYPE)
    keys_r = np.ascontiguousarray(keys_r, KP_DTYPE)
    desc_l = np.ascontiguousarray(desc_l, np.uint8)
    desc_r = np.ascontiguousarray(desc_r, np.uint8)
    pl = (C.c_void_p * nl)(*[a.ctypes.data for a in levels_l])
    pr = (C.c_void_p * nl)(*[a.ctypes.data for a in levels_r])
    sl = np.array([a.strides[0] for a in levels_l], np.int32)
    sr = np.array([a.strides[0] for a in levels_r], np.int32)
    cr = np.array([a.shape[1] for a in levels_r], np.int32)
    sf = np.ascontiguousarray(scale, np.float32)
    isf = np.ascontiguousarray(inv_scale, np.float32)
    P = Pyramids(nl, C.cast(pl, C.c_void_p), C.cast(pr, C.c_void_p), _p(sl), _p(sr), _p(cr),
                 _p(sf), _p(isf))
    ur = np.zeros(max(len(keys_l), 1), np.float32)
    dp = np.zeros(max(len(keys_l), 1), np.float32)
    n = lib().oracle_compute_stereo_matches(_p(keys_l), _p(desc_l), len(keys_l), _p(keys_r),
                                            _p(desc_r), len(keys_r), C.byref(P),
                                            levels_l[0].shape[0], mbf, mb, _p(ur), _p(dp))
    return n, ur[:len(keys_l)].copy(), dp[:len(keys_l)].copy()
