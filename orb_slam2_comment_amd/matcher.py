"""Host-side mirror of class ORBmatcher (include/ORBmatcher.h:37-103) and of
Frame::ComputeStereoMatches (src/Frame.cc:466-640) over the C ABI.

`FrameView` carries the handful of Frame fields the matchers read (mvKeysUn,
mDescriptors, mvuRight, image bounds, 64x48 grid scale, mvScaleFactors).  The
projection that precedes the two SearchByProjection searches
(src/ORBmatcher.cc:1360-1390, src/Frame.cc:269-325) is done by the caller and
arrives as `QUERY_DTYPE` records; `project_last_frame` / `project_map_points`
below restate it for callers that hold plain arrays.
"""
import ctypes as C

import numpy as np

from . import capi
from .capi import KP_DTYPE, QUERY_DTYPE, check, lib, ptr

TH_HIGH, TH_LOW, HISTO_LENGTH = 100, 50, 30
FRAME_GRID_ROWS, FRAME_GRID_COLS = 48, 64


class FrameView:
    """Flat view of a Frame (include/Frame.h) for the matchers."""

    def __init__(self, keys_un, descriptors, scale_factors, bounds, u_right=None):
        self.keys = np.ascontiguousarray(keys_un, KP_DTYPE)
        self.desc = np.ascontiguousarray(descriptors, np.uint8).reshape(-1, 32)
        self.scale_factors = np.ascontiguousarray(scale_factors, np.float32)
        self.u_right = None if u_right is None else np.ascontiguousarray(u_right, np.float32)
        self.bounds = tuple(np.float32(b) for b in bounds)  # mnMinX, mnMinY, mnMaxX, mnMaxY
        # src/Frame.cc:101-102
        self.grid_inv_w = np.float32(FRAME_GRID_COLS) / (self.bounds[2] - self.bounds[0])
        self.grid_inv_h = np.float32(FRAME_GRID_ROWS) / (self.bounds[3] - self.bounds[1])
        self.N = len(self.keys)

    def c_view(self):
        v = capi.FrameView()
        v.n = self.N
        v.keys, v.desc, v.u_right = ptr(self.keys), ptr(self.desc), ptr(self.u_right)
        v.min_x, v.min_y, v.max_x, v.max_y = self.bounds
        v.grid_inv_w, v.grid_inv_h = self.grid_inv_w, self.grid_inv_h
        v.n_levels = len(self.scale_factors)
        v.scale_factors = ptr(self.scale_factors)
        return v


class ORBmatcher:
    TH_HIGH, TH_LOW, HISTO_LENGTH = TH_HIGH, TH_LOW, HISTO_LENGTH

    def __init__(self, nnratio=0.6, checkOri=True, device=0):
        self._lib = lib()
        h = C.c_void_p()
        check(self._lib.orbhip_matcher_create(device, C.byref(h)), "orbhip_matcher_create")
        self._h = h
        self.mfNNratio = float(nnratio)
        self.mbCheckOrientation = bool(checkOri)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.orbhip_matcher_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- DescriptorDistance (src/ORBmatcher.cc:1647-1663), batched ------------
    def DescriptorDistance(self, a, b):
        a = np.ascontiguousarray(a, np.uint8).reshape(-1, 32)
        b = np.ascontiguousarray(b, np.uint8).reshape(-1, 32)
        out = np.zeros((len(a), len(b)), np.int32)
        check(self._lib.orbhip_descriptor_distance(self._h, ptr(a), len(a), ptr(b), len(b), ptr(out)),
              "orbhip_descriptor_distance")
        return out

    # -- SearchForInitialization (src/ORBmatcher.cc:405-520) -------------------
    def SearchForInitialization(self, F1, F2, vbPrevMatched, windowSize=10):
        """Returns (nmatches, vnMatches12, updated vbPrevMatched)."""
        pm = np.ascontiguousarray(vbPrevMatched, np.float32).reshape(-1, 2).copy()
        m12 = np.full(max(F1.N, 1), -1, np.int32)
        n = C.c_int()
        v1, v2 = F1.c_view(), F2.c_view()
        check(self._lib.orbhip_search_for_initialization(self._h, C.byref(v1), C.byref(v2), ptr(pm), ptr(m12),
                                                         int(windowSize), self.mfNNratio,
                                                         int(self.mbCheckOrientation), C.byref(n)),
              "orbhip_search_for_initialization")
        return n.value, m12[:F1.N].copy(), pm

    # -- SearchByProjection(CurrentFrame, LastFrame, th, bMono) (:1328-1470) ---
    def SearchByProjectionFrame(self, CurrentFrame, queries, query_desc, taken=None):
        """queries: QUERY_DTYPE[nq] (already projected).  Returns (nmatches, assign[N])."""
        q = np.ascontiguousarray(queries, QUERY_DTYPE)
        qd = np.ascontiguousarray(query_desc, np.uint8).reshape(-1, 32)
        tk = None if taken is None else np.ascontiguousarray(taken, np.uint8)
        out = np.full(max(CurrentFrame.N, 1), -1, np.int32)
        n = C.c_int()
        v = CurrentFrame.c_view()
        check(self._lib.orbhip_search_by_projection_frame(self._h, C.byref(v), ptr(q), ptr(qd), len(q), ptr(tk),
                                                          ptr(out), int(self.mbCheckOrientation), C.byref(n)),
              "orbhip_search_by_projection_frame")
        return n.value, out[:CurrentFrame.N].copy()

    # -- SearchByProjection(F, vpMapPoints, th) (:45-129) ----------------------
    def SearchByProjectionPoints(self, F, queries, query_desc, taken=None):
        q = np.ascontiguousarray(queries, QUERY_DTYPE)
        qd = np.ascontiguousarray(query_desc, np.uint8).reshape(-1, 32)
        tk = None if taken is None else np.ascontiguousarray(taken, np.uint8)
        out = np.full(max(F.N, 1), -1, np.int32)
        n = C.c_int()
        v = F.c_view()
        check(self._lib.orbhip_search_by_projection_points(self._h, C.byref(v), ptr(q), ptr(qd), len(q), ptr(tk),
                                                           ptr(out), self.mfNNratio, C.byref(n)),
              "orbhip_search_by_projection_points")
        return n.value, out[:F.N].copy()

    # -- SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist) (:1472-1599) ---
    def SearchByProjectionKeyFrame(self, CurrentFrame, queries, query_desc, taken=None, ORBdist=100):
        q = np.ascontiguousarray(queries, QUERY_DTYPE)
        qd = np.ascontiguousarray(query_desc, np.uint8).reshape(-1, 32)
        tk = None if taken is None else np.ascontiguousarray(taken, np.uint8)
        out = np.full(max(CurrentFrame.N, 1), -1, np.int32)
        n = C.c_int()
        v = CurrentFrame.c_view()
        check(self._lib.orbhip_search_by_projection_keyframe(self._h, C.byref(v), ptr(q), ptr(qd), len(q), ptr(tk),
                                                             ptr(out), int(ORBdist), int(self.mbCheckOrientation),
                                                             C.byref(n)), "orbhip_search_by_projection_keyframe")
        return n.value, out[:CurrentFrame.N].copy()

    # -- SearchByProjection(pKF, Scw, vpPoints, vpMatched, th) (:290-403) -------
    def SearchByProjectionSim3(self, KF, queries, query_desc, matched=None):
        q = np.ascontiguousarray(queries, QUERY_DTYPE)
        qd = np.ascontiguousarray(query_desc, np.uint8).reshape(-1, 32)
        tk = None if matched is None else np.ascontiguousarray(matched, np.uint8)
        out = np.full(max(KF.N, 1), -1, np.int32)
        n = C.c_int()
        v = KF.c_view()
        check(self._lib.orbhip_search_by_projection_sim3(self._h, C.byref(v), ptr(q), ptr(qd), len(q), ptr(tk), ptr(out),
                                                         C.byref(n)), "orbhip_search_by_projection_sim3")
        return n.value, out[:KF.N].copy()

    # -- inner search of Fuse x2 (:825-1100) and SearchBySim3 (:1102-1326) -------
    def SearchBestInWindow(self, KF, queries, query_desc, inv_level_sigma2=None):
        """Independent best match per query; chi-square gate enabled when inv_level_sigma2 is given (Fuse).
        Returns (best_idx[nq], best_dist[nq])."""
        q = np.ascontiguousarray(queries, QUERY_DTYPE)
        qd = np.ascontiguousarray(query_desc, np.uint8).reshape(-1, 32)
        bi = np.full(max(len(q), 1), -1, np.int32)
        bd = np.full(max(len(q), 1), 256, np.int32)
        sig = None if inv_level_sigma2 is None else np.ascontiguousarray(inv_level_sigma2, np.float32)
        v = KF.c_view()
        check(self._lib.orbhip_search_best_in_window(self._h, C.byref(v), ptr(q), ptr(qd), len(q), int(sig is not None),
                                                     ptr(sig), ptr(bi), ptr(bd)), "orbhip_search_best_in_window")
        return bi[:len(q)].copy(), bd[:len(q)].copy()

    # -- vocabulary-guided searches ---------------------------------------------
    def SearchByBoW(self, F1, node1, valid1, F2, node2, blocked2=None, max_dist=50):
        """ORBmatcher::SearchByBoW (src/ORBmatcher.cc:159-288 with max_dist=50; :522-655 with max_dist=49 and
        blocked2 = "pKF2 has no good map point").  node1/node2: vocabulary node id per keypoint (capi.NO_NODE = absent
        from the FeatureVector).  Returns (nmatches, matches12[n1])."""
        v1, v2 = F1.c_view(), F2.c_view()
        n1a = np.ascontiguousarray(node1, np.uint32)
        n2a = np.ascontiguousarray(node2, np.uint32)
        if len(n1a) != F1.N or len(n2a) != F2.N:
            raise ValueError("node id arrays must have one entry per keypoint")
        va = None if valid1 is None else np.ascontiguousarray(valid1, np.uint8)
        bl = None if blocked2 is None else np.ascontiguousarray(blocked2, np.uint8)
        m12 = np.full(max(F1.N, 1), -1, np.int32)
        nm = C.c_int(0)
        check(self._lib.orbhip_search_by_bow(self._h, C.byref(v1), ptr(n1a), ptr(va), C.byref(v2), ptr(n2a), ptr(bl),
                                             int(max_dist), self.mfNNratio, int(self.mbCheckOrientation), ptr(m12),
                                             C.byref(nm)), "orbhip_search_by_bow")
        return nm.value, m12[:F1.N].copy()

    def SearchByBoWDevice(self, pairs, cap, side1, f1_first, f1_step, side2, f2_first, f2_step, d_matches12, d_nmatches,
                          max_dist=50, d_valid1=0, d_blocked2=0):
        """Device-resident, batched SearchByBoW.  side1 / side2 = (d_kps, d_desc, d_n, d_node) device pointers (ints)
        in the extractor / vocabulary batch layout; asynchronous on the matcher's stream."""
        check(self._lib.orbhip_search_by_bow_device(self._h, pairs, cap, side1[0], side1[1], side1[2], side1[3], d_valid1,
                                                    f1_first, f1_step, side2[0], side2[1], side2[2], side2[3], d_blocked2,
                                                    f2_first, f2_step, int(max_dist), self.mfNNratio,
                                                    int(self.mbCheckOrientation), d_matches12, d_nmatches),
              "orbhip_search_by_bow_device")

    def SearchForTriangulation(self, F1, node1, valid1, F2, node2, valid2, F12, epipole, level_sigma2, bOnlyStereo=False):
        """ORBmatcher::SearchForTriangulation (src/ORBmatcher.cc:657-823).  Returns (nmatches, matches12[n1]); the
        reference's vMatchedPairs are the (i, matches12[i]) with matches12[i] >= 0."""
        v1, v2 = F1.c_view(), F2.c_view()
        n1a = np.ascontiguousarray(node1, np.uint32)
        n2a = np.ascontiguousarray(node2, np.uint32)
        if len(n1a) != F1.N or len(n2a) != F2.N:
            raise ValueError("node id arrays must have one entry per keypoint")
        va = None if valid1 is None else np.ascontiguousarray(valid1, np.uint8)
        vb = None if valid2 is None else np.ascontiguousarray(valid2, np.uint8)
        F = np.ascontiguousarray(F12, np.float32).reshape(9)
        sg = np.ascontiguousarray(level_sigma2, np.float32)
        m12 = np.full(max(F1.N, 1), -1, np.int32)
        nm = C.c_int(0)
        check(self._lib.orbhip_search_for_triangulation(self._h, C.byref(v1), ptr(n1a), ptr(va), C.byref(v2), ptr(n2a),
                                                        ptr(vb), ptr(F), float(epipole[0]), float(epipole[1]), ptr(sg),
                                                        int(bOnlyStereo), int(self.mbCheckOrientation), ptr(m12),
                                                        C.byref(nm)), "orbhip_search_for_triangulation")
        return nm.value, m12[:F1.N].copy()

    def ComputeDistinctiveDescriptors(self, descriptor_lists):
        """MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:242-307) for a batch of map points.
        descriptor_lists: sequence of (N_p x 32) uint8 arrays; returns best index per map point (-1 if empty)."""
        lens = [len(np.asarray(d).reshape(-1, 32)) for d in descriptor_lists]
        off = np.zeros(len(lens) + 1, np.int32)
        off[1:] = np.cumsum(lens)
        flat = (np.concatenate([np.asarray(d, np.uint8).reshape(-1, 32) for d in descriptor_lists])
                if off[-1] > 0 else np.zeros((1, 32), np.uint8))
        flat = np.ascontiguousarray(flat, np.uint8)
        best = np.full(max(len(lens), 1), -1, np.int32)
        check(self._lib.orbhip_distinctive_descriptors(self._h, ptr(flat), ptr(off), len(lens), ptr(best)),
              "orbhip_distinctive_descriptors")
        return best[:len(lens)].copy()

    # -- Frame constructor glue --------------------------------------------------
    def UndistortKeyPoints(self, keys, fx, fy, cx, cy, dist):
        """Frame::UndistortKeyPoints (src/Frame.cc:404-434); dist = (k1, k2, p1, p2[, k3]).  Returns mvKeysUn."""
        k = np.ascontiguousarray(keys, KP_DTYPE)
        d = np.zeros(5, np.float32)
        d[:len(dist)] = dist
        out = np.zeros(max(len(k), 1), KP_DTYPE)
        check(self._lib.orbhip_undistort_keypoints(self._h, ptr(k), len(k), fx, fy, cx, cy, ptr(d), ptr(out)),
              "orbhip_undistort_keypoints")
        return out[:len(k)].copy()

    def AssignFeaturesToGrid(self, F):
        """Frame::AssignFeaturesToGrid (src/Frame.cc:230-245).  Returns (cell_of[n], cell_start[3073], cell_items[m]):
        mGrid[x][y] = cell_items[cell_start[x*48+y] : cell_start[x*48+y+1]]."""
        v = F.c_view()
        cell_of = np.full(max(F.N, 1), -1, np.int32)
        start = np.zeros(FRAME_GRID_COLS * FRAME_GRID_ROWS + 1, np.int32)
        items = np.zeros(max(F.N, 1), np.int32)
        check(self._lib.orbhip_assign_features_to_grid(self._h, C.byref(v), ptr(cell_of), ptr(start), ptr(items)),
              "orbhip_assign_features_to_grid")
        return cell_of[:F.N].copy(), start, items[:start[-1]].copy()

    def ComputeStereoFromRGBD(self, keys, keys_un, imDepth, mbf):
        """Frame::ComputeStereoFromRGBD (src/Frame.cc:643-664); imDepth: 2-D float32.  Returns (mvuRight, mvDepth)."""
        k = np.ascontiguousarray(keys, KP_DTYPE)
        ku = k if keys_un is None else np.ascontiguousarray(keys_un, KP_DTYPE)
        d = np.ascontiguousarray(imDepth, np.float32)
        ur = np.full(max(len(k), 1), -1, np.float32)
        dp = np.full(max(len(k), 1), -1, np.float32)
        check(self._lib.orbhip_compute_stereo_from_rgbd(self._h, ptr(k), ptr(ku), len(k), ptr(d), d.shape[0], d.shape[1],
                                                        d.shape[1], float(mbf), ptr(ur), ptr(dp)),
              "orbhip_compute_stereo_from_rgbd")
        return ur[:len(k)].copy(), dp[:len(k)].copy()

    # -- device-resident, batched SearchByProjection ---------------------------
    def set_stream(self, stream):
        check(self._lib.orbhip_matcher_set_stream(self._h, stream), "orbhip_matcher_set_stream")

    def sync(self):
        check(self._lib.orbhip_matcher_sync(self._h), "orbhip_matcher_sync")

    def SearchByProjectionFrameDevice(self, pairs, d_kps, d_desc, d_n, cap, bounds, d_q, d_qdesc, d_nq, qcap, d_assign,
                                      d_nmatches, d_u_right=0, d_taken=0):
        """All d_* are device pointers (ints).  bounds = (mnMinX, mnMinY, mnMaxX, mnMaxY)."""
        b = [np.float32(v) for v in bounds]
        inv_w = np.float32(FRAME_GRID_COLS) / (b[2] - b[0])
        inv_h = np.float32(FRAME_GRID_ROWS) / (b[3] - b[1])
        check(self._lib.orbhip_search_by_projection_frame_device(
            self._h, pairs, d_kps, d_desc, d_n, cap, d_u_right, d_taken, b[0], b[1], inv_w, inv_h, d_q, d_qdesc, d_nq,
            qcap, int(self.mbCheckOrientation), d_assign, d_nmatches), "orbhip_search_by_projection_frame_device")

    def SearchForInitializationDevice(self, pairs, d_kps, d_desc, d_n, cap, f1_first, f1_step, f2_first, f2_step, bounds,
                                      keep_prev, d_prev_matched, windowSize, d_matches12, d_nmatches):
        """Batched, device-resident SearchForInitialization; keep_prev = 0 resets vbPrevMatched to F1's keypoints."""
        b = [np.float32(v) for v in bounds]
        inv_w = np.float32(FRAME_GRID_COLS) / (b[2] - b[0])
        inv_h = np.float32(FRAME_GRID_ROWS) / (b[3] - b[1])
        check(self._lib.orbhip_search_for_initialization_device(
            self._h, pairs, d_kps, d_desc, d_n, cap, f1_first, f1_step, f2_first, f2_step, b[0], b[1], inv_w, inv_h,
            int(not keep_prev), d_prev_matched, int(windowSize), self.mfNNratio, int(self.mbCheckOrientation), d_matches12,
            d_nmatches), "orbhip_search_for_initialization_device")

    # -- projection prologues on the device (src/ORBmatcher.cc:1339-1390; src/Frame.cc:269-325) ------------------
    def ProjectLastFrame(self, cam, Tcw, Tlw, world, flags, last_keys, th, bMono):
        """Prologue of SearchByProjection(CurrentFrame, LastFrame, th, bMono): QUERY_DTYPE[n] for
        SearchByProjectionFrame.  Tcw / Tlw: 4x4 or 3x4 float32; world [n,3]; flags [n] uint8 (POINT_* bits)."""
        Tc = np.ascontiguousarray(np.asarray(Tcw, np.float32)[:3, :4])
        Tl = np.ascontiguousarray(np.asarray(Tlw, np.float32)[:3, :4])
        world = np.ascontiguousarray(world, np.float32).reshape(-1, 3)
        flags = np.ascontiguousarray(flags, np.uint8)
        keys = np.ascontiguousarray(last_keys, KP_DTYPE)
        n = len(keys)
        q = np.zeros(n, QUERY_DTYPE)
        check(self._lib.orbhip_project_last_frame(self._h, C.byref(cam), ptr(Tc), ptr(Tl), n, ptr(world), ptr(flags),
                                                  ptr(keys), float(th), int(bMono), ptr(q)), "orbhip_project_last_frame")
        return q

    def FrustumQueries(self, cam, Tcw, world, normal, max_dist, min_dist, flags, viewingCosLimit, th):
        """Frame::isInFrustum for n map points + the window of SearchByProjection(F, vpMapPoints, th).
        Returns (QUERY_DTYPE[n], view_cos[n])."""
        Tc = np.ascontiguousarray(np.asarray(Tcw, np.float32)[:3, :4])
        world = np.ascontiguousarray(world, np.float32).reshape(-1, 3)
        normal = np.ascontiguousarray(normal, np.float32).reshape(-1, 3)
        mx, mn = np.ascontiguousarray(max_dist, np.float32), np.ascontiguousarray(min_dist, np.float32)
        flags = np.ascontiguousarray(flags, np.uint8)
        n = len(world)
        q, vc = np.zeros(n, QUERY_DTYPE), np.zeros(n, np.float32)
        check(self._lib.orbhip_frustum_queries(self._h, C.byref(cam), ptr(Tc), n, ptr(world), ptr(normal), ptr(mx), ptr(mn),
                                               ptr(flags), float(viewingCosLimit), float(th), ptr(q), ptr(vc)),
              "orbhip_frustum_queries")
        return q, vc

    # -- Fuse x2 (src/ORBmatcher.cc:825-1100) and SearchBySim3 (:1102-1326), prologue included ---------------------
    @staticmethod
    def _pts(world, normal, max_dist, min_dist, flags):
        w = np.ascontiguousarray(world, np.float32).reshape(-1, 3)
        nn = None if normal is None else np.ascontiguousarray(normal, np.float32).reshape(-1, 3)
        return (w, nn, np.ascontiguousarray(max_dist, np.float32), np.ascontiguousarray(min_dist, np.float32),
                np.ascontiguousarray(flags, np.uint8))

    def KeyFrameQueries(self, cam, mode, double_invz, T1, T2, world, normal, max_dist, min_dist, flags, th):
        T1 = np.ascontiguousarray(np.asarray(T1, np.float32)[:3, :4])
        T2 = None if T2 is None else np.ascontiguousarray(np.asarray(T2, np.float32)[:3, :4])
        w, nn, mx, mn, fg = self._pts(world, normal, max_dist, min_dist, flags)
        q = np.zeros(len(w), QUERY_DTYPE)
        check(self._lib.orbhip_keyframe_queries(self._h, C.byref(cam), int(mode), int(double_invz), ptr(T1), ptr(T2), len(w),
                                                ptr(w), ptr(nn), ptr(mx), ptr(mn), ptr(fg), float(th), ptr(q)),
              "orbhip_keyframe_queries")
        return q

    def Fuse(self, KF, cam, Tcw, world, normal, max_dist, min_dist, flags, point_desc, th, inv_level_sigma2, sim3_form=False):
        """Fuse up to the decision: (best_idx[n], best_dist[n]); the caller applies bestDist <= TH_LOW and the
        replace-or-add side effects in order."""
        Tc = np.ascontiguousarray(np.asarray(Tcw, np.float32)[:3, :4])
        w, nn, mx, mn, fg = self._pts(world, normal, max_dist, min_dist, flags)
        pd = np.ascontiguousarray(point_desc, np.uint8).reshape(-1, 32)
        sig = np.ascontiguousarray(inv_level_sigma2, np.float32)
        bi, bd = np.full(max(len(w), 1), -1, np.int32), np.full(max(len(w), 1), 256, np.int32)
        v = KF.c_view()
        check(self._lib.orbhip_fuse(self._h, C.byref(v), C.byref(cam), ptr(Tc), int(sim3_form), len(w), ptr(w), ptr(nn), ptr(mx),
                                    ptr(mn), ptr(fg), ptr(pd), float(th), ptr(sig), ptr(bi), ptr(bd)), "orbhip_fuse")
        return bi[:len(w)].copy(), bd[:len(w)].copy()

    def SearchBySim3(self, KF1, KF2, cam, T1w, T2w, S21, S12, pts1, pts2, th):
        """pts = (world, max_dist, min_dist, flags, desc) per key-frame slot.  Returns (nFound, matches12[N1])."""
        mats = [np.ascontiguousarray(np.asarray(T, np.float32)[:3, :4]) for T in (T1w, T2w, S21, S12)]
        a = [self._pts(p[0], None, p[1], p[2], p[3]) + (np.ascontiguousarray(p[4], np.uint8).reshape(-1, 32),) for p in (pts1, pts2)]
        m12 = np.full(max(KF1.N, 1), -1, np.int32)
        n = C.c_int()
        v1, v2 = KF1.c_view(), KF2.c_view()
        check(self._lib.orbhip_search_by_sim3(self._h, C.byref(v1), C.byref(v2), C.byref(cam), ptr(mats[0]), ptr(mats[1]),
                                              ptr(mats[2]), ptr(mats[3]), ptr(a[0][0]), ptr(a[0][2]), ptr(a[0][3]), ptr(a[0][4]),
                                              ptr(a[0][5]), ptr(a[1][0]), ptr(a[1][2]), ptr(a[1][3]), ptr(a[1][4]), ptr(a[1][5]),
                                              float(th), ptr(m12), C.byref(n)), "orbhip_search_by_sim3")
        return n.value, m12[:KF1.N].copy()

    def ProjectLastFrameDevice(self, pairs, cam, d_Tcw, d_Tlw, d_kps, d_n, cap, last_first, last_step, d_world, d_flags, th,
                               bMono, d_q, d_nq):
        check(self._lib.orbhip_project_last_frame_device(self._h, pairs, C.byref(cam), d_Tcw, d_Tlw, d_kps, d_n, cap,
                                                         last_first, last_step, d_world, d_flags, float(th), int(bMono),
                                                         d_q, d_nq), "orbhip_project_last_frame_device")

    def TrackLastFrameDevice(self, pairs, cam, d_Tcw, d_Tlw, d_kps, d_desc, d_n, cap, cur_first, cur_step, last_first,
                             last_step, d_world, d_flags, th, bMono, d_assign, d_nmatches, d_u_right=0, d_taken=0):
        """Prologue + search + resolve + rotation cull of SearchByProjection(CurrentFrame, LastFrame, th, bMono) for
        `pairs` (current, last) frame pairs of the extractor output arrays, all on the device, one stream."""
        check(self._lib.orbhip_track_last_frame_device(self._h, pairs, C.byref(cam), d_Tcw, d_Tlw, d_kps, d_desc, d_n, cap,
                                                       cur_first, cur_step, last_first, last_step, d_world, d_flags,
                                                       d_u_right, d_taken, float(th), int(bMono),
                                                       int(self.mbCheckOrientation), d_assign, d_nmatches),
              "orbhip_track_last_frame_device")

    def FrustumQueriesDevice(self, frames, cam, d_Tcw, pcap, d_np, d_world, d_normal, d_max_dist, d_min_dist, d_flags,
                             viewingCosLimit, th, d_q, d_view_cos=0):
        check(self._lib.orbhip_frustum_queries_device(self._h, frames, C.byref(cam), d_Tcw, pcap, d_np, d_world, d_normal,
                                                      d_max_dist, d_min_dist, d_flags, float(viewingCosLimit), float(th),
                                                      d_q, d_view_cos), "orbhip_frustum_queries_device")

    def SearchByProjectionPointsDevice(self, pairs, d_kps, d_desc, d_n, cap, bounds, d_q, d_qdesc, d_nq, qcap, d_assign,
                                       d_nmatches, d_u_right=0, d_taken=0):
        b = [np.float32(v) for v in bounds]
        inv_w = np.float32(FRAME_GRID_COLS) / (b[2] - b[0])
        inv_h = np.float32(FRAME_GRID_ROWS) / (b[3] - b[1])
        check(self._lib.orbhip_search_by_projection_points_device(
            self._h, pairs, d_kps, d_desc, d_n, cap, d_u_right, d_taken, b[0], b[1], inv_w, inv_h, d_q, d_qdesc, d_nq,
            qcap, self.mfNNratio, d_assign, d_nmatches), "orbhip_search_by_projection_points_device")

    def ComputeStereoMatchesDevice(self, ext_left, l0, ls, ext_right, r0, rs, pairs, d_kps_l, d_desc_l, d_n_l, d_kps_r,
                                   d_desc_r, d_n_r, cap, mbf, mb, d_u_right, d_depth, d_nmatches):
        """Batched, device-resident ComputeStereoMatches (all d_* are device pointers)."""
        check(self._lib.orbhip_compute_stereo_matches_device(self._h, ext_left._h, l0, ls, ext_right._h, r0, rs, pairs,
                                                             d_kps_l, d_desc_l, d_n_l, d_kps_r, d_desc_r, d_n_r, cap, mbf,
                                                             mb, d_u_right, d_depth, d_nmatches),
              "orbhip_compute_stereo_matches_device")

    # -- Frame::ComputeStereoMatches (src/Frame.cc:466-640) --------------------
    def ComputeStereoMatches(self, extractor_left, extractor_right, keys_l, desc_l, keys_r, desc_r, mbf, mb,
                             frame_l=0, frame_r=0):
        """Pyramids are those held by the two extractor handles after their last call.
        Returns (nmatches, mvuRight, mvDepth)."""
        kl = np.ascontiguousarray(keys_l, KP_DTYPE)
        kr = np.ascontiguousarray(keys_r, KP_DTYPE)
        dl = np.ascontiguousarray(desc_l, np.uint8)
        dr = np.ascontiguousarray(desc_r, np.uint8)
        ur = np.full(max(len(kl), 1), -1, np.float32)
        dp = np.full(max(len(kl), 1), -1, np.float32)
        n = C.c_int()
        check(self._lib.orbhip_compute_stereo_matches(self._h, extractor_left._h, frame_l, extractor_right._h,
                                                      frame_r, ptr(kl), ptr(dl), len(kl), ptr(kr), ptr(dr),
                                                      len(kr), mbf, mb, ptr(ur), ptr(dp), C.byref(n)),
              "orbhip_compute_stereo_matches")
        return n.value, ur[:len(kl)].copy(), dp[:len(kl)].copy()


def RadiusByViewingCos(viewCos):
    """src/ORBmatcher.cc:131-137"""
    return 2.5 if viewCos > 0.998 else 4.0


def make_camera(fx, fy, cx, cy, bounds, scale_factors, mbf=0.0, mb=0.0):
    """orbhip_camera from plain numbers.  bounds = (mnMinX, mnMinY, mnMaxX, mnMaxY); mfLogScaleFactor =
    log(mfScaleFactor) as the Frame constructor computes it (src/Frame.cc:71: float log of the float scale factor)."""
    cam = capi.Camera()
    cam.fx, cam.fy, cam.cx, cam.cy, cam.mbf, cam.mb = fx, fy, cx, cy, mbf, mb
    cam.min_x, cam.min_y, cam.max_x, cam.max_y = bounds
    sf = np.asarray(scale_factors, np.float32)
    cam.n_levels = len(sf)
    cam.log_scale_factor = float(np.log(np.float32(sf[1] if len(sf) > 1 else 1.2), dtype=np.float32))
    for i, v in enumerate(sf):
        cam.scale_factors[i] = float(v)
    return cam


def project_last_frame(Tcw, K, bounds, world_pts, last_octaves, last_angles, valid, observed, scale_factors,
                       th, mbf=0.0, bForward=False, bBackward=False):
    """Projection prologue of SearchByProjection(CurrentFrame, LastFrame, th, bMono)
    (src/ORBmatcher.cc:1360-1390) in float32, one operation at a time.
    K = (fx, fy, cx, cy); Tcw 4x4; world_pts [n,3].  Returns QUERY_DTYPE[n]."""
    f32 = np.float32
    Tcw = np.asarray(Tcw, f32)
    P = np.asarray(world_pts, f32)
    fx, fy, cx, cy = (f32(v) for v in K)
    n = len(P)
    q = np.zeros(n, QUERY_DTYPE)
    R, t = Tcw[:3, :3], Tcw[:3, 3]
    for i in range(n):
        if not valid[i]:
            continue
        xc = f32(f32(f32(R[0, 0] * P[i, 0]) + f32(R[0, 1] * P[i, 1])) + f32(R[0, 2] * P[i, 2])) + t[0]
        yc = f32(f32(f32(R[1, 0] * P[i, 0]) + f32(R[1, 1] * P[i, 1])) + f32(R[1, 2] * P[i, 2])) + t[1]
        zc = f32(f32(f32(R[2, 0] * P[i, 0]) + f32(R[2, 1] * P[i, 1])) + f32(R[2, 2] * P[i, 2])) + t[2]
        if zc == 0:
            continue
        invzc = f32(1.0 / np.float64(zc))
        if invzc < 0:
            continue
        u = f32(f32(f32(fx * xc) * invzc) + cx)
        v = f32(f32(f32(fy * yc) * invzc) + cy)
        if u < bounds[0] or u > bounds[2] or v < bounds[1] or v > bounds[3]:
            continue
        o = int(last_octaves[i])
        q[i]["valid"] = 1
        q[i]["u"], q[i]["v"] = u, v
        q[i]["radius"] = f32(f32(th) * f32(scale_factors[o]))
        if bForward:
            q[i]["min_level"], q[i]["max_level"] = o, -1
        elif bBackward:
            q[i]["min_level"], q[i]["max_level"] = 0, o
        else:
            q[i]["min_level"], q[i]["max_level"] = o - 1, o + 1
        q[i]["ur"] = f32(u - f32(f32(mbf) * invzc))
        q[i]["level_aux"] = o
        q[i]["angle"] = last_angles[i]
        q[i]["observed"] = int(observed[i])
    return q
