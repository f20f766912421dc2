"""tools/replay_kitti.py on a three-frame synthetic sequence laid out like KITTI odometry (mono_kitti.cc:127-157)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import synth_frame
from test_settings_cpu import YAML, _png

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_replay_tool(tmp_path):
    import orb_slam2_comment_amd as pkg
    from orb_slam2_comment_amd import settings as S
    seq = tmp_path / "00"
    (seq / "image_0").mkdir(parents=True)
    frames = [synth_frame(7, 640, 360, shift_xy=(2 * i, 0)) for i in range(3)]
    (seq / "times.txt").write_text("".join("%e\n" % (0.1 * i) for i in range(3)))
    for i, f in enumerate(frames):
        _png(str(seq / "image_0" / ("%06d.png" % i)), f, [4, 1, 2, 0, 3])
    yaml = tmp_path / "KITTI.yaml"
    yaml.write_text(YAML.replace("nFeatures: 2000", "nFeatures: 600"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "replay_kitti.py"), str(yaml), str(seq), "--match"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = r.stdout
    assert "Images in the sequence: 3" in out and "median extraction + matching time" in out
    # the tool's per-frame keypoint counts: first frame through the 2*nFeatures extractor (src/Tracking.cc:258-260)
    ex = S.make_extractors(S.load_settings(yaml), S.MONOCULAR)
    want = [len(ex["ini"](frames[0])[0])] + [len(ex["left"](f)[0]) for f in frames[1:]]
    got = float(out.split("mean keypoints per frame:")[1].split()[0])
    assert abs(got - sum(want) / 3.0) < 0.06
    assert float(out.split("mean matches to the previous frame:")[1].split()[0]) > 100


def test_replay_tool_stereo(tmp_path):
    """stereo_kitti.cc layout: image_0 / image_1 pairs, two extractors, ComputeStereoMatches; the tool's stereo-match
    count per pair equals a direct call of the mirror on the same pair."""
    import orb_slam2_comment_amd as pkg
    from orb_slam2_comment_amd import settings as S
    from helpers import synth_stereo
    seq = tmp_path / "00"
    (seq / "image_0").mkdir(parents=True)
    (seq / "image_1").mkdir(parents=True)
    pairs = [synth_stereo(3 + i, 640, 360) for i in range(2)]
    (seq / "times.txt").write_text("".join("%e\n" % (0.1 * i) for i in range(2)))
    for i, (l, r) in enumerate(pairs):
        _png(str(seq / "image_0" / ("%06d.png" % i)), l, [0, 1, 2])
        _png(str(seq / "image_1" / ("%06d.png" % i)), r, [2, 0, 1])
    yaml = tmp_path / "KITTI.yaml"
    yaml.write_text(YAML.replace("nFeatures: 2000", "nFeatures: 800"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "replay_kitti.py"), str(yaml), str(seq), "--stereo", "--match"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = r.stdout
    assert "Images in the sequence: 2" in out and "median extraction + stereo + matching time" in out
    st = S.load_settings(yaml)
    ex = S.make_extractors(st, S.STEREO)
    m = pkg.ORBmatcher(0.9, True)
    want = []
    for l, rr in pairs:
        kl, dl = ex["left"](l)
        kr, dr = ex["right"](rr)
        want.append(m.ComputeStereoMatches(ex["left"], ex["right"], kl, dl, kr, dr, float(st["Camera.bf"]),
                                           float(st["Camera.bf"]) / float(st["Camera.fx"]))[0])
    got = float(out.split("mean stereo matches per pair:")[1].split()[0])
    assert abs(got - sum(want) / 2.0) < 0.06 and got > 30
