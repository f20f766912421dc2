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
    # the tool's per-frame keypoint counts against the ORACLE: first frame through the 2*nFeatures extractor
    # (src/Tracking.cc:258-260)
    from oracle import oracle_py as O
    want = [len(O.OracleExtractor(1200, 1.2, 8, 20, 7).extract(frames[0])[0])] + \
           [len(O.OracleExtractor(600, 1.2, 8, 20, 7).extract(f)[0]) for f in frames[1:]]
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
    from oracle import oracle_py as O      # the stereo-match counts against the ORACLE on the same pairs
    want = []
    mbf = np.float32(st["Camera.bf"]); mb = np.float32(mbf / np.float32(st["Camera.fx"]))
    for l, rr in pairs:
        oL, oR = O.OracleExtractor(800, 1.2, 8, 20, 7), O.OracleExtractor(800, 1.2, 8, 20, 7)
        kl, dl = oL.extract(l)
        kr, dr = oR.extract(rr)
        lv_l = [np.ascontiguousarray(oL.level_padded(k))[19:-19, 19:-19] for k in range(8)]
        lv_r = [np.ascontiguousarray(oR.level_padded(k))[19:-19, 19:-19] for k in range(8)]
        t = oL.tables()
        want.append(O.compute_stereo_matches(kl, dl, kr, dr, lv_l, lv_r, t["scale"], t["inv_scale"], float(mbf), float(mb))[0])
    got = float(out.split("mean stereo matches per pair:")[1].split()[0])
    assert abs(got - sum(want) / 2.0) < 0.06 and got > 30


def _build_cpp_example(tmp_path):
    exe = str(tmp_path / "replay_kitti")
    libdir = os.path.join(ROOT, "orb_slam2_comment_amd")
    subprocess.run(["g++", "-O2", "-std=c++11", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "replay_kitti.cpp"),
                    "-o", exe, "-L", libdir, "-lorbhip", "-lz", "-lpthread", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return exe


def _records(path):
    buf, out, off = open(path, "rb").read(), [], 0
    while off < len(buf):
        nb = int(np.frombuffer(buf[off:off + 4], np.int32)[0])
        out.append(buf[off + 4:off + 4 + nb])
        off += 4 + nb
    return out


def test_cpp_replay_example_against_the_oracle(tmp_path, oracle):
    """examples/replay_kitti.cpp (the C++ counterpart of mono_kitti.cc / stereo_kitti.cc for the front-end) on synthetic
    KITTI-layout sequences: every frame's keypoints / descriptors, the stereo matches and the frame-to-frame matches it
    dumps are the ORACLE's results for the same images, not just the mirror's."""
    import orb_slam2_comment_amd as pkg
    from helpers import synth_stereo
    O = oracle
    exe = _build_cpp_example(tmp_path)
    yaml = tmp_path / "KITTI.yaml"
    yaml.write_text(YAML.replace("nFeatures: 2000", "nFeatures: 500"))
    nf, Wd, Hd = 500, 640, 360
    sf = np.cumprod(np.concatenate([[np.float32(1)], np.full(7, np.float32(1.2))])).astype(np.float32)
    b = (0.0, 0.0, float(Wd), float(Hd))
    # ---- monocular, PNG + one PGM frame ----
    seq = tmp_path / "00"
    (seq / "image_0").mkdir(parents=True)
    frames = [synth_frame(9, Wd, Hd, shift_xy=(2 * i, 0)) for i in range(3)]
    (seq / "times.txt").write_text("".join("%e\n" % (0.1 * i) for i in range(3)))
    for i, f in enumerate(frames[:2]):
        _png(str(seq / "image_0" / ("%06d.png" % i)), f, [4, 1, 2, 0, 3])
    with open(str(seq / "image_0" / "000002.pgm"), "wb") as f:
        f.write(b"P5\n# synthetic\n%d %d\n255\n" % (Wd, Hd) + frames[2].tobytes())
    out = str(tmp_path / "mono.bin")
    r = subprocess.run([exe, str(yaml), str(seq), "--match", "--dump", out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "Images in the sequence: 3" in r.stdout and "median extraction + matching time" in r.stdout
    rec = _records(out)      # frame 0: k, d; frames 1, 2: assign, k, d
    assert len(rec) == 2 + 3 + 3
    oras = [O.OracleExtractor(2 * nf, 1.2, 8, 20, 7), O.OracleExtractor(nf, 1.2, 8, 20, 7), O.OracleExtractor(nf, 1.2, 8, 20, 7)]
    want = [oras[i].extract(frames[i]) for i in range(3)]
    got = [(rec[0], rec[1]), (rec[3], rec[4]), (rec[6], rec[7])]
    for (gk, gd), (wk, wd) in zip(got, want):
        assert np.frombuffer(gk, pkg.KP_DTYPE).tobytes() == np.ascontiguousarray(wk, pkg.KP_DTYPE).tobytes()
        assert np.array_equal(np.frombuffer(gd, np.uint8).reshape(-1, 32), wd)
    for t, assign_rec in ((1, rec[2]), (2, rec[5])):
        lk, ld = want[t - 1]
        ck, cd = want[t]
        q = np.zeros(len(lk), pkg.QUERY_DTYPE)
        q["valid"] = 1; q["u"] = lk["x"]; q["v"] = lk["y"]; q["radius"] = np.float32(15) * sf[lk["octave"]]
        q["min_level"] = lk["octave"] - 1; q["max_level"] = lk["octave"] + 1; q["angle"] = lk["angle"]; q["observed"] = 1; q["ur"] = -1
        keep = []
        on, oassign = O.search_by_projection_frame(O.make_frame(ck, cd, None, b, sf, keep), q, ld, None, True)
        assert np.array_equal(np.frombuffer(assign_rec, np.int32), oassign) and on > 100
    # ---- stereo ----
    seq2 = tmp_path / "01"
    (seq2 / "image_0").mkdir(parents=True); (seq2 / "image_1").mkdir(parents=True)
    pairs = [synth_stereo(4, Wd, Hd), synth_stereo(4, Wd, Hd, shift_xy=(3, 0))]
    (seq2 / "times.txt").write_text("0.0\n0.1\n")
    for i, (l, rr) in enumerate(pairs):
        _png(str(seq2 / "image_0" / ("%06d.png" % i)), l, [1, 3])
        _png(str(seq2 / "image_1" / ("%06d.png" % i)), rr, [2, 4])
    out2 = str(tmp_path / "stereo.bin")
    r = subprocess.run([exe, str(yaml), str(seq2), "--stereo", "--match", "--dump", out2], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rec = _records(out2)     # frame 0: k, d, ur, depth; frame 1: assign, k, d, ur, depth
    assert len(rec) == 4 + 5
    mbf = np.float32(386.1448); mb = np.float32(mbf / np.float32(718.856))
    res = []
    for l, rr in pairs:
        oL, oR = O.OracleExtractor(nf, 1.2, 8, 20, 7), O.OracleExtractor(nf, 1.2, 8, 20, 7)
        kl, dl = oL.extract(l); kr, dr = oR.extract(rr)
        lv_l = [np.ascontiguousarray(oL.level_padded(k))[19:-19, 19:-19] for k in range(8)]
        lv_r = [np.ascontiguousarray(oR.level_padded(k))[19:-19, 19:-19] for k in range(8)]
        t = oL.tables()
        n, ur, dp = O.compute_stereo_matches(kl, dl, kr, dr, lv_l, lv_r, t["scale"], t["inv_scale"], float(mbf), float(mb))
        res.append((kl, dl, ur, dp, n))
    for (gk, gd, gu, gp), (wk, wd, wu, wp, n) in zip(((rec[0], rec[1], rec[2], rec[3]), (rec[5], rec[6], rec[7], rec[8])), res):
        assert np.frombuffer(gk, pkg.KP_DTYPE).tobytes() == np.ascontiguousarray(wk, pkg.KP_DTYPE).tobytes()
        assert np.array_equal(np.frombuffer(gd, np.uint8).reshape(-1, 32), wd)
        assert np.array_equal(np.frombuffer(gu, np.float32), wu) and np.array_equal(np.frombuffer(gp, np.float32), wp) and n > 30
    lk, ld, lur = res[0][0], res[0][1], res[0][2]
    ck, cd, cur_ur = res[1][0], res[1][1], res[1][2]
    q = np.zeros(len(lk), pkg.QUERY_DTYPE)
    q["valid"] = 1; q["u"] = lk["x"]; q["v"] = lk["y"]; q["radius"] = np.float32(7) * sf[lk["octave"]]
    q["min_level"] = lk["octave"] - 1; q["max_level"] = lk["octave"] + 1; q["angle"] = lk["angle"]; q["observed"] = 1
    q["ur"] = np.where(lur > 0, lur, lk["x"])
    keep = []
    on, oassign = O.search_by_projection_frame(O.make_frame(ck, cd, cur_ur, b, sf, keep), q, ld, None, True)
    assert np.array_equal(np.frombuffer(rec[4], np.int32), oassign) and on > 50
