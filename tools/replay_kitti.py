#!/usr/bin/env python3
"""Front-end replay of a KITTI-style sequence, the counterpart of Examples/Monocular/mono_kitti.cc and (with --stereo)
Examples/Stereo/stereo_kitti.cc for the part of ORB-SLAM2 this repository replaces: reads the settings file (ORBextractor.* keys, src/Tracking.cc:112-125) and the
sequence (<seq>/times.txt, <seq>/image_0/%06d.png, mono_kitti.cc:127-157), runs ORBextractor::operator() on every
frame through the host API (one frame in, keypoints + descriptors out, like Frame::ExtractORB) and prints the same
statistics the example prints for tracking (median / mean per-frame time, mono_kitti.cc:110-119).  With --match it
also runs SearchByProjection(frame t, frame t-1) with identity motion (window th = 15) as a tracking stand-in.

With --stereo the sequence is read like stereo_kitti.cc:130-157 (image_0 = left, image_1 = right), every pair goes
through the two extractors the Tracking constructor creates for a stereo sensor (src/Tracking.cc:119-122; the reference
runs them on two threads, src/Frame.cc:78-81) and Frame::ComputeStereoMatches (Camera.fx / Camera.bf of the settings
file, src/Tracking.cc:86 and src/Frame.cc:108-112); --match then tracks with the stereo window th = 7
(src/Tracking.cc:880) and the stereo consistency gate.

  python tools/replay_kitti.py path/to/KITTI00-02.yaml path/to/sequence [--max-frames N] [--match] [--stereo]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from orb_slam2_comment_amd import FrameView, ORBmatcher, QUERY_DTYPE  # noqa: E402
from orb_slam2_comment_amd.settings import (MONOCULAR, STEREO, load_kitti_sequence, load_settings, make_extractors,  # noqa: E402
                                            read_gray_image)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("settings")
    ap.add_argument("sequence")
    ap.add_argument("--max-frames", type=int, default=0)
    ap.add_argument("--match", action="store_true")
    ap.add_argument("--stereo", action="store_true", help="stereo_kitti.cc: image_0 / image_1 pairs + ComputeStereoMatches")
    args = ap.parse_args()
    st = load_settings(args.settings)
    names, stamps = load_kitti_sequence(args.sequence)
    if args.max_frames:
        names, stamps = names[:args.max_frames], stamps[:args.max_frames]
    if args.stereo:
        return stereo(args, st, names, stamps)
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


def stereo(args, st, names_left, stamps):
    """Examples/Stereo/stereo_kitti.cc: left / right image lists, two extractors, ComputeStereoMatches per pair."""
    import threading
    names_right, _ = load_kitti_sequence(args.sequence, camera="image_1")
    ex = make_extractors(st, STEREO)
    fx, bf = float(st["Camera.fx"]), float(st["Camera.bf"])       # mbf = Camera.bf; mb = mbf / fx (src/Frame.cc:112)
    print("Images in the sequence: %d" % len(names_left))
    matcher = ORBmatcher(0.9, True)
    times, counts, stereo_counts, matches = [], [], [], []
    last = None
    for ni, (nl, nr) in enumerate(zip(names_left, names_right)):
        iml, imr = read_gray_image(nl), read_gray_image(nr)
        if iml.size == 0:
            print("Failed to load image at: %s" % nl, file=sys.stderr)
            return 1
        t1 = time.perf_counter()
        res = {}
        tl = threading.Thread(target=lambda: res.__setitem__("l", ex["left"](iml)))      # src/Frame.cc:78-81
        tr = threading.Thread(target=lambda: res.__setitem__("r", ex["right"](imr)))
        tl.start(); tr.start(); tl.join(); tr.join()
        (kl, dl), (kr, dr) = res["l"], res["r"]
        ns, ur, depth = matcher.ComputeStereoMatches(ex["left"], ex["right"], kl, dl, kr, dr, bf, bf / fx) if len(kl) and len(kr) \
            else (0, np.full(len(kl), -1, np.float32), np.full(len(kl), -1, np.float32))
        if args.match and last is not None and len(kl) and len(last[0]):
            sf = ex["left"].GetScaleFactors()
            cur = FrameView(kl, dl, sf, (0.0, 0.0, float(iml.shape[1]), float(iml.shape[0])), ur)
            lk, ld, lur = last
            q = np.zeros(len(lk), QUERY_DTYPE)
            q["valid"] = 1; q["u"] = lk["x"]; q["v"] = lk["y"]; q["radius"] = 7 * sf[lk["octave"]]
            q["min_level"] = lk["octave"] - 1; q["max_level"] = lk["octave"] + 1; q["angle"] = lk["angle"]; q["observed"] = 1
            q["ur"] = np.where(lur > 0, lur, lk["x"])      # identity motion: the point keeps its right-image coordinate
            matches.append(matcher.SearchByProjectionFrame(cur, q, ld)[0])
        times.append(time.perf_counter() - t1)
        counts.append(len(kl)); stereo_counts.append(ns)
        last = (kl, dl, ur)
    times.sort()
    n = len(times)
    print("-------\n")
    print("median extraction + stereo%s time: %.6f" % (" + matching" if args.match else "", times[n // 2]))
    print("mean extraction + stereo%s time: %.6f" % (" + matching" if args.match else "", sum(times) / n))
    print("mean keypoints per left frame: %.1f" % (sum(counts) / n))
    print("mean stereo matches per pair: %.1f" % (sum(stereo_counts) / n))
    if matches:
        print("mean matches to the previous frame: %.1f" % (sum(matches) / len(matches)))
    return 0


if __name__ == "__main__":
    sys.exit(main())
