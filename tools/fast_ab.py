#!/usr/bin/env python3
"""Development tool: per-stage HIP-event times of one 64-frame pipeline for kernel variants, interleaved
rounds in ONE process (cdna_hip_programming.md section 5.4 rule 24).  Variants are the development switches
exported only by the -DORBHIP_DEVTOOLS build (tools/_dev/liborbhip_dev.so, orbhip_dev_*), never part of the product library.

  python tools/fast_ab.py [--rounds 8] [--frames 64] [--variants 1,0]
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def use_dev_build():
    """These tools drive development exports (orbhip_dev_*) that only the -DORBHIP_DEVTOOLS build of the library has:
    `make -C orb_slam2_comment_amd/csrc dev` writes it to tools/_dev/liborbhip_dev.so; the product library is untouched."""
    import subprocess
    from orb_slam2_comment_amd import capi
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "orb_slam2_comment_amd", "csrc"), "dev"], check=True)
    capi.use_library(os.path.join(ROOT, "tools", "_dev", "liborbhip_dev.so"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=8)
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--calls", type=int, default=20)
    ap.add_argument("--variants", default="0,3,4", help="FAST development variants; 1000 + T selects the octree workgroup size T "
                                                       "(256 / 512 / 1024) with the product FAST kernel")
    ap.add_argument("--size", default="1241x376")
    ap.add_argument("--nfeatures", type=int, default=1000)
    args = ap.parse_args()
    use_dev_build()
    import torch
    from orb_slam2_comment_amd import ORBextractor
    from orb_slam2_comment_amd.capi import lib
    from orb_slam2_comment_amd.synth import synth_frame
    W, H = (int(v) for v in args.size.split("x"))
    dev = torch.device("cuda", 0)
    B = args.frames
    frames = np.stack([synth_frame(1 + (i // 2) % 8, W, H, shift_xy=(3 * (i % 2), 0)) for i in range(min(B, 16))])
    frames = np.stack([frames[i % len(frames)] for i in range(B)])
    d_img = torch.from_numpy(frames).to(dev)
    ext = ORBextractor(args.nfeatures, 1.2, 8, 20, 7, device=0)
    ext.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    cap = ext.capacity(H, W)
    d_k = torch.zeros((B, cap, 7), dtype=torch.int32, device=dev)
    d_d = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    d_n = torch.zeros(B, dtype=torch.int32, device=dev)
    d_s = torch.zeros(B, dtype=torch.int32, device=dev)
    L = lib()
    variants = [int(v) for v in args.variants.split(",")]

    def select(v):
        if v >= 1000:
            L.orbhip_dev_set_fast_variant(ext._h, 0)
            L.orbhip_dev_set_octree_threads(ext._h, v - 1000)
        else:
            L.orbhip_dev_set_octree_threads(ext._h, 256)
            L.orbhip_dev_set_fast_variant(ext._h, v)

    def run():
        ext.extract_batch_device(d_img.data_ptr(), B, H, W, d_k.data_ptr(), d_d.data_ptr(), cap, d_n.data_ptr(), d_s.data_ptr())

    ref = None
    res = {v: [] for v in variants}
    for v in variants:                         # results must not depend on the variant
        select(v)
        run()
        torch.cuda.synchronize()
        sig = (d_n.cpu().numpy().copy(), d_d.cpu().numpy().copy(), d_k.cpu().numpy().copy())
        if ref is None:
            ref = sig
        elif v < 3 or v >= 1000:
            n = ref[0]
            same = np.array_equal(n, sig[0]) and all(np.array_equal(ref[1][b, :n[b]], sig[1][b, :n[b]]) and
                                                      np.array_equal(ref[2][b, :n[b]], sig[2][b, :n[b]]) for b in range(B))
            print("variant %d output identical to variant %d: %s" % (v, variants[0], same), file=sys.stderr)
    for r in range(args.rounds):
        for v in variants:
            select(v)
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            ext.set_profiling(True)
            for _ in range(args.calls):
                run()
            torch.cuda.synchronize()
            res[v].append(ext.stage_times_us())
            ext.set_profiling(False)
    out = {}
    if 2 in variants:                          # stamped diagnostic build: shares of the wave lifetime per phase
        L.orbhip_dev_set_fast_variant(ext._h, 2)
        z = (C.c_ulonglong * 8)()
        torch.cuda.synchronize()
        L.orbhip_dev_fast_stamps(z)
        run()
        torch.cuda.synchronize()
        L.orbhip_dev_fast_stamps(z)
        waves = max(1, z[7])
        names = ["prologue", "staging", "dense", "score", "nms_emit"]
        tot = float(sum(z[i] for i in range(5)))
        print(json.dumps({"stamps_cycles_per_wave": {names[i]: round(z[i] / waves, 1) for i in range(5)},
                          "shares": {names[i]: round(z[i] / tot, 3) for i in range(5)}, "waves": int(waves)}), file=sys.stderr)
    for v in variants:
        out[str(v)] = {k: {"median": round(float(np.median([r[k] for r in res[v]])), 2),
                           "min": round(float(np.min([r[k] for r in res[v]])), 2)} for k in res[v][0]}
    print(json.dumps({"frames": B, "size": args.size, "stage_us": out}))


if __name__ == "__main__":
    main()
