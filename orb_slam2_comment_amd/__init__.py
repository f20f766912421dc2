"""orb_slam2_comment_amd -- MI355X-native ORB front-end + descriptor matching for ORB-SLAM2.

Only the hot path of SURVEY.md section 8 lives here: `csrc/` (HIP kernels + the C ABI of
include/orbhip.h, built into liborbhip.so) and the host-side mirrors of the
reference interfaces (`ORBextractor`, `ORBmatcher`, `ORBVocabulary`).  There is no CPU fallback:
the compute entry points raise if the HIP library is missing.
"""
from . import capi  # noqa: F401
from .capi import KP_DTYPE, QUERY_DTYPE, OrbHipError  # noqa: F401
from .extractor import ORBextractor  # noqa: F401
from .matcher import FrameView, ORBmatcher  # noqa: F401
from .vocabulary import ORBVocabulary  # noqa: F401
