"""Host-side mirror of class ORBextractor (include/ORBextractor.h:45-110) over the C ABI.

`ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST)` keeps the
reference's constructor arguments; calling the object on an image returns
(keypoints, descriptors) like `operator()` fills its two output arguments
(src/ORBextractor.cc:1043-1105).  The `mask` argument is accepted and ignored,
as in the reference (include/ORBextractor.h:58).
"""
import ctypes as C

import numpy as np

from . import capi
from .capi import KP_DTYPE, check, lib, ptr


class ORBextractor:
    HARRIS_SCORE = 0
    FAST_SCORE = 1

    def __init__(self, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, device=0):
        self._lib = lib()
        h = C.c_void_p()
        check(self._lib.orbhip_extractor_create(nfeatures, scaleFactor, nlevels, iniThFAST,
                                                minThFAST, device, C.byref(h)),
              "orbhip_extractor_create")
        self._h = h
        self.nfeatures, self.nlevels, self.device = nfeatures, nlevels, device
        self._shape = None

    # -- lifetime ---------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._lib.orbhip_extractor_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- getters (include/ORBextractor.h:63-83) ----------------------------
    def _tables(self):
        n = self.nlevels
        sf, isf, s2, is2 = (np.zeros(n, np.float32) for _ in range(4))
        nf = np.zeros(n, np.int32)
        check(self._lib.orbhip_extractor_tables(self._h, ptr(sf), ptr(isf), ptr(s2), ptr(is2), ptr(nf)),
              "orbhip_extractor_tables")
        return sf, isf, s2, is2, nf

    def GetLevels(self):
        return self._lib.orbhip_extractor_levels(self._h)

    def GetScaleFactor(self):
        return float(self._tables()[0][1]) if self.nlevels > 1 else 1.0

    def GetScaleFactors(self):
        return self._tables()[0]

    def GetInverseScaleFactors(self):
        return self._tables()[1]

    def GetScaleSigmaSquares(self):
        return self._tables()[2]

    def GetInverseScaleSigmaSquares(self):
        return self._tables()[3]

    def features_per_level(self):
        return self._tables()[4]

    def capacity(self, rows, cols):
        cap = C.c_int()
        check(self._lib.orbhip_extractor_capacity(self._h, rows, cols, C.byref(cap)),
              "orbhip_extractor_capacity")
        return cap.value

    def set_blur_kernel(self, w7):
        w = np.ascontiguousarray(w7, np.int32)
        assert w.shape == (7,)
        check(self._lib.orbhip_extractor_set_blur_kernel(self._h, ptr(w)), "orbhip_extractor_set_blur_kernel")

    def set_lazy_level0(self, on=True):
        """mvImagePyramid[0] on demand: extractions stop writing the padded level-0 plane (a monocular caller never reads
        it); the first accessor that needs it builds it from the image buffer of the last extraction."""
        check(self._lib.orbhip_extractor_set_lazy_level0(self._h, int(on)), "orbhip_extractor_set_lazy_level0")

    def set_stage_gate(self, stage, wait_event=0, record_event=0):
        """Before launching stage 0 pyramid / 1 FAST / 2 octree / 3 descriptors wait for `wait_event`, after it record
        `record_event` (hipEvent_t handles as int, 0 = none): pins how several pipelines' kernels meet on the GPU."""
        check(self._lib.orbhip_extractor_set_stage_gate(self._h, stage, wait_event, record_event), "orbhip_extractor_set_stage_gate")

    def set_stage_gate_wait(self, stage, wait_event):
        check(self._lib.orbhip_extractor_set_stage_gate(self._h, stage, wait_event, C.c_void_p(-1)), "orbhip_extractor_set_stage_gate")

    def set_stage_gate_record(self, stage, record_event):
        check(self._lib.orbhip_extractor_set_stage_gate(self._h, stage, C.c_void_p(-1), record_event), "orbhip_extractor_set_stage_gate")

    # -- operator() ---------------------------------------------------------
    def __call__(self, image, mask=None):
        """image: uint8 [rows, cols] (CV_8UC1).  Returns (keypoints[KP_DTYPE], descriptors[n,32])."""
        if image is None or image.size == 0:
            return np.zeros(0, KP_DTYPE), np.zeros((0, 32), np.uint8)
        if image.dtype != np.uint8 or image.ndim != 2:
            raise TypeError("ORBextractor expects a CV_8UC1 image")  # assert at :1050
        if image.strides[1] != 1:
            image = np.ascontiguousarray(image)
        rows, cols = image.shape
        cap = self.capacity(rows, cols)
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int()
        check(self._lib.orbhip_extract(self._h, ptr(image), rows, cols, image.strides[0], ptr(kps),
                                       ptr(desc), cap, C.byref(n)), "orbhip_extract")
        self._shape = (rows, cols)
        return kps[:n.value].copy(), desc[:n.value].copy()

    def extract_batch(self, images):
        """images: uint8 [B, rows, cols] on the host.  Returns list of (keypoints, descriptors)."""
        images = np.ascontiguousarray(images, np.uint8)
        B, rows, cols = images.shape
        cap = self.capacity(rows, cols)
        kps = np.zeros((B, cap), KP_DTYPE)
        desc = np.zeros((B, cap, 32), np.uint8)
        n = np.zeros(B, np.int32)
        check(self._lib.orbhip_extract_batch(self._h, ptr(images), B, rows, cols, cols, rows * cols,
                                             ptr(kps), ptr(desc), cap, ptr(n)), "orbhip_extract_batch")
        self._shape = (rows, cols)
        return [(kps[b, :n[b]].copy(), desc[b, :n[b]].copy()) for b in range(B)]

    def extract_batch_device(self, d_images, batch, rows, cols, d_kps, d_desc, cap, d_n, d_status=0,
                             stride=None, frame_stride=None):
        """Device-pointer entry (ints from torch .data_ptr()); asynchronous on the handle's stream."""
        stride = cols if stride is None else stride
        frame_stride = rows * stride if frame_stride is None else frame_stride
        check(self._lib.orbhip_extract_batch_device(self._h, d_images, batch, rows, cols, stride,
                                                    frame_stride, d_kps, d_desc, cap, d_n, d_status),
              "orbhip_extract_batch_device")
        self._shape = (rows, cols)

    def sync(self):
        check(self._lib.orbhip_extractor_sync(self._h), "orbhip_extractor_sync")

    def stream(self):
        return self._lib.orbhip_extractor_stream(self._h)

    def set_stream(self, stream):
        """stream: hipStream_t as int (e.g. torch.cuda.current_stream().cuda_stream), 0 = handle's own."""
        check(self._lib.orbhip_extractor_set_stream(self._h, stream), "orbhip_extractor_set_stream")

    def set_profiling(self, on=True):
        check(self._lib.orbhip_extractor_set_profiling(self._h, int(on)), "orbhip_extractor_set_profiling")

    def stage_times_us(self):
        t = np.zeros(6, np.float32)
        check(self._lib.orbhip_extractor_stage_times(self._h, ptr(t)), "orbhip_extractor_stage_times")
        return dict(zip(["pyramid", "fast", "octree", "blur", "describe", "total"], t.tolist()))

    # -- mvImagePyramid (include/ORBextractor.h:85) -------------------------
    def level_info(self, level, frame=0):
        r, c, s, p = C.c_int(), C.c_int(), C.c_int(), C.c_void_p()
        check(self._lib.orbhip_pyramid_level(self._h, frame, level, C.byref(r), C.byref(c), C.byref(s),
                                             C.byref(p)), "orbhip_pyramid_level")
        return r.value, c.value, s.value, p.value

    def image_pyramid(self, level, frame=0, with_border=False):
        r, c, _, _ = self.level_info(level, frame)
        b = 19 if with_border else 0
        out = np.zeros((r + 2 * b, c + 2 * b), np.uint8)
        check(self._lib.orbhip_pyramid_level_download(self._h, frame, level, int(with_border), ptr(out),
                                                      out.strides[0]), "orbhip_pyramid_level_download")
        return out

    def blurred_level(self, level, frame=0):
        r, c, _, _ = self.level_info(level, frame)
        out = np.zeros((r, c), np.uint8)
        check(self._lib.orbhip_blurred_level_download(self._h, frame, level, ptr(out), out.strides[0]),
              "orbhip_blurred_level_download")
        return out

    def level_candidates(self, level, frame=0, cap=1 << 20):
        x, y, s = (np.zeros(cap, np.int32) for _ in range(3))
        n = C.c_int()
        check(self._lib.orbhip_level_candidates(self._h, frame, level, ptr(x), ptr(y), ptr(s), cap,
                                                C.byref(n)), "orbhip_level_candidates")
        return x[:n.value].copy(), y[:n.value].copy(), s[:n.value].copy()
