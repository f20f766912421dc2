#!/usr/bin/env python3
"""Development tool: how well do two stages of the front-end share the GPU?  Two 64-frame pipelines (extractor +
matcher each, own stream); every stage X of pipeline A is repeated alone, every stage Y of pipeline B alone, then
both loops run concurrently.  overlap = (t_X + t_Y) / t_both: 1.0 = the two stages just take turns, 2.0 = they fit
beside each other for free.  Stages: P pyramid, F FAST, O octree, B blur, D orientation + descriptors, M matching
(TrackLastFrameDevice).  Uses orbhip_dev_set_stage_mask of the -DORBHIP_DEVTOOLS build (tools/_dev/liborbhip_dev.so); the product library has no such export.

  python tools/coexec.py [--frames 64] [--reps 20]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

STAGES = {"P": 1, "F": 2, "O": 4, "B": 8, "D": 16}


def use_dev_build():
    """These tools drive development exports (orbhip_dev_*) that only the -DORBHIP_DEVTOOLS build of the library has:
    `make -C orb_slam2_comment_amd/csrc dev` writes it to tools/_dev/liborbhip_dev.so; the product library is untouched."""
    import subprocess
    from orb_slam2_comment_amd import capi
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "orb_slam2_comment_amd", "csrc"), "dev"], check=True)
    capi.use_library(os.path.join(ROOT, "tools", "_dev", "liborbhip_dev.so"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--groups", default="P,F,O,B,D,M", help="comma-separated stage groups, e.g. PB,F,ODM")
    args = ap.parse_args()
    use_dev_build()
    import torch
    from orb_slam2_comment_amd import ORBextractor, ORBmatcher
    from orb_slam2_comment_amd import matcher as M
    from orb_slam2_comment_amd.capi import lib, POINT_OBSERVED, POINT_PRESENT
    from orb_slam2_comment_amd.synth import synth_frame
    W, H, B = 1241, 376, args.frames
    dev = torch.device("cuda", 0)
    L = lib()
    frames = np.stack([synth_frame(1 + (i // 2) % 8, W, H, shift_xy=(3 * (i % 2), 0)) for i in range(min(B, 16))])
    frames = np.stack([frames[i % len(frames)] for i in range(B)])
    d_img = torch.from_numpy(frames).to(dev)
    pipes = []
    for _ in range(2):
        e = ORBextractor(1000, 1.2, 8, 20, 7, device=0)
        m = ORBmatcher(0.9, True, device=0)
        st = torch.cuda.Stream(dev)
        e.set_stream(st.cuda_stream); m.set_stream(st.cuda_stream)
        cap = e.capacity(H, W)
        o = {"k": torch.zeros((B, cap, 7), dtype=torch.int32, device=dev), "d": torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev),
             "n": torch.zeros(B, dtype=torch.int32, device=dev), "s": torch.zeros(B, dtype=torch.int32, device=dev),
             "a": torch.zeros((B // 2, cap), dtype=torch.int32, device=dev), "m": torch.zeros(B // 2, dtype=torch.int32, device=dev)}
        pipes.append((e, m, st, o, cap))
    sf = pipes[0][0].GetScaleFactors()
    cam = M.make_camera(718.856, 718.856, 607.1928, 185.2157, (0.0, 0.0, float(W), float(H)), sf, mbf=386.1448, mb=386.1448 / 718.856)
    Tlw = torch.eye(4)[:3, :].reshape(1, 12).repeat(B // 2, 1).contiguous().to(dev)
    Tc = torch.eye(4); Tc[0, 3] = 3.0 * 12.0 / 718.856
    Tcw = Tc[:3, :].reshape(1, 12).repeat(B // 2, 1).contiguous().to(dev)
    d_world = torch.zeros((B, pipes[0][4], 3), dtype=torch.float32, device=dev)
    d_flags = torch.full((B, pipes[0][4]), POINT_PRESENT | POINT_OBSERVED, dtype=torch.uint8, device=dev)

    def run(p, group):
        e, m, st, o, cap = pipes[p]
        mask = sum(STAGES[c] for c in group if c in STAGES)
        if mask:
            L.orbhip_dev_set_stage_mask(e._h, mask)
            e.extract_batch_device(d_img.data_ptr(), B, H, W, o["k"].data_ptr(), o["d"].data_ptr(), cap, o["n"].data_ptr(), o["s"].data_ptr())
        if "M" in group:
            m.TrackLastFrameDevice(B // 2, cam, Tcw.data_ptr(), Tlw.data_ptr(), o["k"].data_ptr(), o["d"].data_ptr(), o["n"].data_ptr(),
                                   cap, 1, 2, 0, 2, d_world.data_ptr(), d_flags.data_ptr(), 15.0, True, o["a"].data_ptr(), o["m"].data_ptr())

    for p in range(2):                      # full runs leave every intermediate buffer filled
        run(p, "PFOBD")
    torch.cuda.synchronize()
    kf = pipes[0][3]["k"][0::2].view(torch.float32)
    d_world[0::2, :, 0] = (kf[..., 0] - 607.1928) * (12.0 / 718.856)
    d_world[0::2, :, 1] = (kf[..., 1] - 185.2157) * (12.0 / 718.856)
    d_world[0::2, :, 2] = 12.0
    torch.cuda.synchronize()

    def timed(jobs, reps):
        # jobs: list of (pipe, group, count per rep)
        for _ in range(2):
            for p, g, c in jobs:
                for _ in range(c):
                    run(p, g)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            for p, g, c in jobs:
                for _ in range(c):
                    run(p, g)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e6

    groups = args.groups.split(",")
    alone = {}
    for g in groups:
        alone[g] = [timed([(p, g, 1)], args.reps * 3) for p in range(2)]
    print("alone (us per %d frames, pipeline A / B):" % B, {g: [round(v, 1) for v in alone[g]] for g in groups}, flush=True)
    res = {}
    print("pair   t_X    t_Y    both   overlap")
    for i, x in enumerate(groups):
        for y in groups[i:]:
            # balance the two loops: repeat the shorter stage so that both sides carry about the same stand-alone time
            tx, ty = alone[x][0], alone[y][1]
            cx = max(1, int(round(ty / tx))) if tx < ty else 1
            cy = max(1, int(round(tx / ty))) if ty < tx else 1
            both = timed([(0, x, cx), (1, y, cy)], args.reps)
            ov = (cx * tx + cy * ty) / both
            res[x + "|" + y] = {"cx": cx, "cy": cy, "t_x": round(cx * tx, 1), "t_y": round(cy * ty, 1), "both": round(both, 1), "overlap": round(ov, 3)}
            print("%-6s %6.1f %6.1f %6.1f  %.2f   (x%d, x%d)" % (x + "|" + y, cx * tx, cy * ty, both, ov, cx, cy), flush=True)
    print(json.dumps({"alone": alone, "pairs": res}))


if __name__ == "__main__":
    main()
