"""CPU-side checks of the C-ABI boundary: the library loads, exports every symbol the header
declares, and fails loudly (no CPU fallback) when there is no GPU.  No compute calls."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from helpers import has_gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from orb_slam2_comment_amd import capi
    hdr = open(os.path.join(ROOT, "include", "orbhip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(orbhip_[a-z_0-9]+)\s*\(", hdr))
    assert len(declared) >= 25
    L = capi.lib()
    for name in declared:
        assert hasattr(L, name), name
    assert declared == {s[0] for s in capi.SYMBOLS}


def test_struct_layouts_match_header():
    from orb_slam2_comment_amd import capi
    assert capi.KP_DTYPE.itemsize == 28          # cv::KeyPoint
    assert capi.QUERY_DTYPE.itemsize == 40
    assert C.sizeof(capi.FrameView) == 72


def test_product_does_not_touch_the_oracle():
    """The product package must never import/link anything under oracle/."""
    pkg = os.path.join(ROOT, "orb_slam2_comment_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "oracle_py" not in txt and "orb_oracle" not in txt and "liborb_oracle" not in txt, f


@pytest.mark.skipif(has_gpu(), reason="checks the no-GPU failure mode")
def test_fails_loudly_without_gpu():
    import orb_slam2_comment_amd as pkg
    from orb_slam2_comment_amd import capi
    with pytest.raises(pkg.OrbHipError) as ei:
        pkg.ORBextractor(1000, 1.2, 8, 20, 7)
    assert ei.value.code == capi.E_NODEVICE
    with pytest.raises(pkg.OrbHipError):
        pkg.ORBmatcher()


def test_bad_arguments_are_rejected_before_any_device_work():
    from orb_slam2_comment_amd import capi
    L = capi.lib()
    h = C.c_void_p()
    assert L.orbhip_extractor_create(1000, 1.2, 0, 20, 7, 0, C.byref(h)) == capi.E_ARG     # nlevels 0
    assert L.orbhip_extractor_create(1000, 1.0, 8, 20, 7, 0, C.byref(h)) == capi.E_ARG     # scale 1.0
    assert L.orbhip_extractor_create(1000, 1.2, 8, 20, 7, 0, None) == capi.E_ARG
    assert L.orbhip_matcher_create(0, None) == capi.E_ARG
    assert L.orbhip_last_error() is not None


def test_synth_is_deterministic():
    from orb_slam2_comment_amd.synth import synth_frame, synth_stereo
    a, b = synth_frame(1, 320, 240), synth_frame(1, 320, 240)
    assert a.dtype == np.uint8 and a.shape == (240, 320) and np.array_equal(a, b)
    assert not np.array_equal(a, synth_frame(2, 320, 240))
    l, r = synth_stereo(1, 320, 240)
    assert l.shape == r.shape and not np.array_equal(l, r)
    assert zlib_crc(a) == zlib_crc(b)


def zlib_crc(a):
    import zlib
    return zlib.crc32(a.tobytes())
