#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer entry a Tracking thread calls (orbhip_extract_batch: host images in,
host keypoints / descriptors out), straight through the C ABI with preallocated buffers.

  python tools/bench_host_path.py [--frames 64] [--reps 20] [--pinned]

A batch of 32 frames or more runs as the library's chunk pipeline (copy in / kernels / copy out on three streams); a
single frame replays the captured hipGraph.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--pinned", action="store_true", help="input frames in page-locked host memory (torch pin_memory)")
    args = ap.parse_args()
    import torch
    from orb_slam2_comment_amd import ORBextractor
    from orb_slam2_comment_amd.capi import KP_DTYPE, check, lib, ptr
    from orb_slam2_comment_amd.synth import synth_frame
    B = args.frames
    uniq = [synth_frame(1 + i) for i in range(8)]
    frames = np.stack([uniq[i % 8] for i in range(B)])
    keep = None
    if args.pinned:
        keep = torch.from_numpy(frames).pin_memory()
        frames = keep.numpy()
    ext = ORBextractor(1000, 1.2, 8, 20, 7)
    rows, cols = frames.shape[1:]
    cap = ext.capacity(rows, cols)
    kps = np.zeros((B, cap), KP_DTYPE)
    desc = np.zeros((B, cap, 32), np.uint8)
    n = np.zeros(B, np.int32)
    L = lib()

    def call():
        check(L.orbhip_extract_batch(ext._h, ptr(frames), B, rows, cols, cols, rows * cols, ptr(kps), ptr(desc), cap, ptr(n)),
              "orbhip_extract_batch")
    for _ in range(3):
        call()
    ts = []
    for _ in range(args.reps):
        t0 = time.perf_counter()
        call()
        ts.append(time.perf_counter() - t0)
    med = float(np.median(ts))
    print(json.dumps({"frames": B, "pinned_input": bool(args.pinned),
                      "ms_median": round(med * 1e3, 3), "ms_min": round(min(ts) * 1e3, 3), "frames_per_s": round(B / med, 1),
                      "mean_keypoints": round(float(n.mean()), 1)}))


if __name__ == "__main__":
    main()
