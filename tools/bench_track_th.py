#!/usr/bin/env python3
"""Secondary measurement: device time of orbhip_track_last_frame_device (prologue + window search + resolve) for 32
(last, cur) KITTI-shape pairs as a function of the search radius `th` (src/Tracking.cc:880-892 uses 7 / 15 and 2 x th on
its retry; wider windows put more than 64 candidates on a query's list).  HIP events on the launch stream.

  python tools/bench_track_th.py [--pairs 32] [--ths 7,15,30,60,100]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=32)
    ap.add_argument("--ths", default="7,15,30,60,100")
    args = ap.parse_args()
    import torch
    import bench as Bn
    from orb_slam2_comment_amd import ORBextractor, ORBmatcher
    from orb_slam2_comment_amd import matcher as M
    from orb_slam2_comment_amd.capi import POINT_OBSERVED, POINT_PRESENT
    from orb_slam2_comment_amd.synth import synth_frame
    dev = torch.device("cuda", 0)
    W, H, P = Bn.W, Bn.H, args.pairs
    B = 2 * P
    frames = np.stack([synth_frame(1 + (i // 2) % 8, W, H, shift_xy=(3 * (i % 2), 0)) for i in range(min(B, 16))])
    d_img = torch.from_numpy(np.stack([frames[i % len(frames)] for i in range(B)])).to(dev)
    st = torch.cuda.Stream(dev)
    ext = ORBextractor(Bn.NFEAT, 1.2, Bn.NLEVELS, 20, 7)
    mt = ORBmatcher(0.9, True)
    ext.set_stream(st.cuda_stream); mt.set_stream(st.cuda_stream)
    cap = ext.capacity(H, W)
    k = torch.zeros((B, cap, 7), dtype=torch.int32, device=dev); d = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    n = torch.zeros(B, dtype=torch.int32, device=dev); s = torch.zeros(B, dtype=torch.int32, device=dev)
    ext.extract_batch_device(d_img.data_ptr(), B, H, W, k.data_ptr(), d.data_ptr(), cap, n.data_ptr(), s.data_ptr())
    torch.cuda.synchronize(dev)
    cam = M.make_camera(Bn.KITTI_FX, Bn.KITTI_FY, Bn.KITTI_CX, Bn.KITTI_CY, (0.0, 0.0, float(W), float(H)), ext.GetScaleFactors(),
                        mbf=Bn.KITTI_BF, mb=Bn.KITTI_BF / Bn.KITTI_FX)
    Tlw = torch.eye(4)[:3, :].reshape(1, 12).repeat(P, 1).contiguous().to(dev)
    Tc = torch.eye(4); Tc[0, 3] = float(np.float32(3.0) * np.float32(Bn.DEPTH) / np.float32(Bn.KITTI_FX))
    Tcw = Tc[:3, :].reshape(1, 12).repeat(P, 1).contiguous().to(dev)
    world = torch.zeros((B, cap, 3), dtype=torch.float32, device=dev)
    kf = k[0::2].view(torch.float32)
    world[0::2, :, 0] = (kf[..., 0] - Bn.KITTI_CX) * (Bn.DEPTH / Bn.KITTI_FX)
    world[0::2, :, 1] = (kf[..., 1] - Bn.KITTI_CY) * (Bn.DEPTH / Bn.KITTI_FY)
    world[0::2, :, 2] = Bn.DEPTH
    flags = torch.full((B, cap), POINT_PRESENT | POINT_OBSERVED, dtype=torch.uint8, device=dev)
    a = torch.zeros((P, cap), dtype=torch.int32, device=dev); m = torch.zeros(P, dtype=torch.int32, device=dev)
    out = {}
    for th in [float(v) for v in args.ths.split(",")]:
        def run():
            mt.TrackLastFrameDevice(P, cam, Tcw.data_ptr(), Tlw.data_ptr(), k.data_ptr(), d.data_ptr(), n.data_ptr(), cap, 1, 2, 0, 2,
                                    world.data_ptr(), flags.data_ptr(), th, True, a.data_ptr(), m.data_ptr())
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(20):
            run()
        e1.record(st)
        torch.cuda.synchronize(dev)
        out["th_%g" % th] = {"us_per_call": round(e0.elapsed_time(e1) / 20 * 1e3, 1), "mean_matches": round(float(m.float().mean().item()), 1)}
    print(json.dumps({"pairs": P, "track_last_frame_device": out}))


if __name__ == "__main__":
    main()
