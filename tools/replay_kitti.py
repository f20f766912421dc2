#!/usr/bin/env python3
"""Front-end replay of a KITTI-style sequence, the counterpart of Examples/Monocular/mono_kitti.cc for the part of
ORB-SLAM2 this repository replaces: reads the settings file (ORBextractor.* keys, src/Tracking.cc:112-125) and the
sequence (<seq>/times.txt, <seq>/image_0/%06d.png, mono_kitti.cc:127-157), runs ORBextractor::operator() on every
frame through the host API (one frame in, keypoints + descriptors out, like Frame::ExtractORB) and prints the same
statistics the example prints for tracking (median / mean per-frame time, mono_kitti.cc:110-119).  With --match it
also runs SearchByProjection(frame t, frame t-1) with identity motion (window th = 15) as a tracking stand-in.

  python tools/replay_kitti.py path/to/KITTI00-02.yaml path/to/sequence [--max-frames N] [--match]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from orb_slam2_comment_amd import FrameView, ORBmatcher, QUERY_DTYPE  # noqa: E402
from orb_slam2_comment_amd.settings import (MONOCULAR, load_kitti_sequence, load_settings, make_extractors,  # noqa: E402
                                            read_gray_image)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("settings")
    ap.add_argument("sequence")
    ap.add_argument("--max-frames", type=int, default=0)
    ap.add_argument("--match", action="store_true")
    args = ap.parse_args()
    st = load_settings(args.settings)
    names, stamps = load_kitti_sequence(args.sequence)
    if args.max_frames:
        names, stamps = names[:args.max_frames], stamps[:args.max_frames]
    ex = make_extractors(st, MONOCULAR)
    print("ORB Extractor Parameters:\n- Number of Features: %d\n- Scale Levels: %d\n- Scale Factor: %g\n"
          "- Initial Fast Threshold: %d\n- Minimum Fast Threshold: %d" %
          (st["ORBextractor.nFeatures"], st["ORBextractor.nLevels"], st["ORBextractor.scaleFactor"],
           st["ORBextractor.iniThFAST"], st["ORBextractor.minThFAST"]))
    print("Images in the sequence: %d" % len(names))
    matcher = ORBmatcher(0.9, True) if args.match else None
    times, counts, matches = [], [], []
    last = None
    for ni, name in enumerate(names):
        im = read_gray_image(name)
        if im.size == 0:
            print("Failed to load image at: %s" % name, file=sys.stderr)
            return 1
        t1 = time.perf_counter()
        e = ex["ini"] if ni == 0 else ex["left"]          # the first frame goes through mpIniORBextractor (src/Tracking.cc:258-260)
        kps, desc = e(im)
        if matcher is not None and last is not None and len(kps) and len(last[0]):
            sf = e.GetScaleFactors()
            cur = FrameView(kps, desc, sf, (0.0, 0.0, float(im.shape[1]), float(im.shape[0])))
            lk, ld = last
            q = np.zeros(len(lk), QUERY_DTYPE)
            q["valid"] = 1; q["u"] = lk["x"]; q["v"] = lk["y"]; q["radius"] = 15 * sf[lk["octave"]]
            q["min_level"] = lk["octave"] - 1; q["max_level"] = lk["octave"] + 1; q["angle"] = lk["angle"]; q["observed"] = 1
            matches.append(matcher.SearchByProjectionFrame(cur, q, ld)[0])
        times.append(time.perf_counter() - t1)
        counts.append(len(kps))
        last = (kps, desc)
    times.sort()
    n = len(times)
    print("-------\n")
    print("median extraction%s time: %.6f" % (" + matching" if args.match else "", times[n // 2]))
    print("mean extraction%s time: %.6f" % (" + matching" if args.match else "", sum(times) / n))
    print("mean keypoints per frame: %.1f" % (sum(counts) / n))
    if matches:
        print("mean matches to the previous frame: %.1f" % (sum(matches) / len(matches)))
    return 0


if __name__ == "__main__":
    sys.exit(main())
