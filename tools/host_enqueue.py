#!/usr/bin/env python3
"""Development tool: is a step bound by the host's launch rate?  Host time to ENQUEUE the extractions of a step (no sync)
against the time the GPU needs for them, for 64 / 512 frames as 1 / 3 pipelines, and the host cost of one device-entry call
(10 launches) on an idle stream.  One MI355X: 64 frames as 3 pipelines 116 us enqueue / 218 us total per step, one call
37 us = 3.7 us per launch -- the small-batch steps are GPU-bound (under-filled launches), not launch-bound.

  python tools/host_enqueue.py
"""
import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from orb_slam2_comment_amd import ORBextractor
from orb_slam2_comment_amd.synth import synth_frame
dev = torch.device('cuda', 0)
W, H = 1241, 376
for B, Hn in ((64, 3), (64, 1), (512, 3)):
    per = B // Hn
    frames = np.stack([synth_frame(1 + i % 8, W, H) for i in range(per)])
    d_img = torch.from_numpy(frames).to(dev)
    exts, streams, outs = [], [], []
    for h in range(Hn):
        e = ORBextractor(1000, 1.2, 8, 20, 7, device=0); e.set_lazy_level0(True)
        st = torch.cuda.Stream(dev); e.set_stream(st.cuda_stream)
        cap = e.capacity(H, W)
        outs.append((torch.zeros((per, cap, 7), dtype=torch.int32, device=dev), torch.zeros((per, cap, 32), dtype=torch.uint8, device=dev),
                     torch.zeros(per, dtype=torch.int32, device=dev), torch.zeros(per, dtype=torch.int32, device=dev)))
        exts.append(e); streams.append(st)
    def step():
        for h, e in enumerate(exts):
            k, d, n, s = outs[h]
            e.extract_batch_device(d_img.data_ptr(), per, H, W, k.data_ptr(), d.data_ptr(), cap, n.data_ptr(), s.data_ptr())
    for _ in range(20): step()
    torch.cuda.synchronize()
    K = 300
    t0 = time.perf_counter()
    for _ in range(K): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("B %d pipelines %d: host enqueue %.1f us/step, total %.1f us/step" % (B, Hn, (t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6))
# host cost of one device-entry call when the GPU work is tiny (3 frames): pure launch overhead
e = ORBextractor(1000, 1.2, 8, 20, 7, device=0); e.set_lazy_level0(True)
st = torch.cuda.Stream(dev); e.set_stream(st.cuda_stream)
per = 3
frames = np.stack([synth_frame(1 + i % 8, W, H) for i in range(per)])
d_img = torch.from_numpy(frames).to(dev)
cap = e.capacity(H, W)
k = torch.zeros((per, cap, 7), dtype=torch.int32, device=dev); d = torch.zeros((per, cap, 32), dtype=torch.uint8, device=dev)
n = torch.zeros(per, dtype=torch.int32, device=dev); s = torch.zeros(per, dtype=torch.int32, device=dev)
for _ in range(20): e.extract_batch_device(d_img.data_ptr(), per, H, W, k.data_ptr(), d.data_ptr(), cap, n.data_ptr(), s.data_ptr())
torch.cuda.synchronize()
ts = []
for _ in range(200):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e.extract_batch_device(d_img.data_ptr(), per, H, W, k.data_ptr(), d.data_ptr(), cap, n.data_ptr(), s.data_ptr())
    ts.append(time.perf_counter() - t0)
print("one extract_batch_device call (10 launches) on an idle stream: host %.1f us median" % (np.median(ts) * 1e6))
