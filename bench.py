#!/usr/bin/env python3
"""Headline benchmark: frames/s of the ORB front-end + frame-to-frame matching on synthetic KITTI-shape frames.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--global-batch 512 | --frames-per-gpu B]

Headline workload (BASELINE.json metric "ORB extract+match, KITTI 1241x376 @1000 feat"; configs[1] + the
SearchByProjection half of configs[2]): a global batch of 512 synthetic 1241x376 uint8 frames (BASELINE.json configs[3]:
"batch=512 ... sharded one-per-GPU"), B = 512 / N of them per GPU, already resident in HBM as B/2 (LastFrame, CurrentFrame)
pairs.  One "step" = ORBextractor::operator() over the B frames (nFeatures=1000,
scaleFactor 1.2, 8 levels, iniThFAST 20 / minThFAST 7) + ORBmatcher::SearchByProjection(CurrentFrame, LastFrame,
th=15, bMono) for the B/2 pairs (src/Tracking.cc:880-885), projection prologue included, everything device resident.
`value` = frames through that step per second.

For N > 1 the driver launches one rank per GPU with torch.distributed.run; `python bench.py --gpus N` without that
environment spawns the N ranks itself (before anything touches a GPU).  Pair p of the global batch goes to rank
p mod N; every step ends with the RCCL gather of the fixed-capacity result slots to rank 0 (configs[3]).

Named secondary objects in the same JSON line (never `value`): extract_only (configs[1]), stereo (configs[2] as one leg:
extract L+R on two handles + Frame::ComputeStereoMatches + SearchByProjection at th = 7 with the `ur` gate), euroc_init (configs[4]: 752x480 @2000 + SearchForInitialization),
bow (extract + ComputeBoW + SearchByBoW).  Rank 0 prints ONE JSON line; README/DESIGN.md describe `roofline` and
`cpu_baseline`.
"""
import argparse
import json
import os
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H, NFEAT, NLEVELS = 1241, 376, 1000, 8
LEVEL_PX = [1241 * 376, 1034 * 313, 862 * 261, 718 * 218, 598 * 181, 499 * 151, 416 * 126, 346 * 105]
SUM_P = sum(LEVEL_PX)                                   # 1,444,097 px
# algorithmic bytes per frame of each stage (stage-materialised model)
ALGO_BYTES = {
    "pyramid": (SUM_P - LEVEL_PX[7]) + (SUM_P - LEVEL_PX[0]),  # K1 as executed by the headline's monocular handles (lazy
                                                              # mvImagePyramid[0]): reads SumP-P7, writes SumP-P0; with the
                                                              # level-0 copy it is P0 + (SumP-P7) + SumP (ALGO_PYRAMID_FULL)
    "fast": SUM_P,                                            # K2+K3: reads SumP
    "blur": 0,                                                # K6 is fused into the descriptor kernel (no blurred plane)
    "describe": NFEAT * (749 + 512) + NFEAT * 60,             # K5+K7 of SURVEY 8d: N*(749+512) read + N*(32+28) written; K6's
                                                              # 2*SumP is NOT counted: no blurred plane is written or read
    "octree": 0,
}
ALGO_PYRAMID_FULL = LEVEL_PX[0] + (SUM_P - LEVEL_PX[7]) + SUM_P
KERNEL_NAME = {"pyramid": "k_pyr_base(level 1)+k_pyr_rows(x6)", "fast": "k_fast_cells", "blur": "(fused into k_describe_fused)",
               "describe": "k_describe_fused", "octree": "k_octree"}
FRAME_BYTES_MODEL = 8971771                             # BASELINE.md section 3, whole path
HBM_PEAK_GBPS = 8000.0                                  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
KITTI_FX, KITTI_FY, KITTI_CX, KITTI_CY, KITTI_BF = 718.856, 718.856, 607.1928, 185.2157, 386.1448   # KITTI00-02.yaml
TRACK_TH, DEPTH = 15.0, 12.0                            # src/Tracking.cc:880 (mono th = 15); synthetic scene depth [m]
STEREO_TH = 7.0                                         # src/Tracking.cc:882-885 (stereo th = 7)
GLOBAL_BATCH = 512                                      # BASELINE.json configs[3]
MIN_TIMED_S = 1.0                                       # the timed region of every leg lasts at least this long


# ---------------------------------------------------------------------------------------------------------------
# CPU baseline: the C oracle (test infrastructure) built -O3 -march=native ON THIS HOST, all cores
# ---------------------------------------------------------------------------------------------------------------
def cpu_baseline(frames, budget_s=20.0):
    """Oracle (plain-C port of the reference algorithm, oracle/orb_oracle.c) on the host cores, same workload as the
    headline: per (last, cur) pair extract both frames, project the last frame's keypoints, SearchByProjection.
    One extractor per thread on a bounded sample.  The library is rebuilt for this host with the reference's own
    flags (-O3 -march=native, CMakeLists.txt:10-18; contraction stays off: the oracle defines the arithmetic)."""
    from oracle import oracle_py as O
    native = os.path.join(ROOT, "oracle", "_native")
    os.makedirs(native, exist_ok=True)
    so = os.path.join(native, "liborb_oracle_native.so")
    flags = "-O3 -march=native"
    try:
        subprocess.run(["gcc"] + flags.split() + ["-fPIC", "-std=c99", "-ffp-contract=off", "-fno-fast-math", "-shared", "-o", so,
                        os.path.join(ROOT, "oracle", "orb_oracle.c"), "-lm"], check=True, capture_output=True)
        O.use_library(so)
    except Exception:
        flags = "-O2 (prebuilt; native rebuild failed)"
    from orb_slam2_comment_amd import matcher as M
    cores = host_cores()
    O.lib()
    sf = np.cumprod(np.concatenate([[np.float32(1)], np.full(NLEVELS - 1, np.float32(1.2))])).astype(np.float32)
    cam = M.make_camera(KITTI_FX, KITTI_FY, KITTI_CX, KITTI_CY, (0.0, 0.0, float(W), float(H)), sf, mbf=KITTI_BF,
                        mb=KITTI_BF / KITTI_FX)
    Tlw = np.eye(4, dtype=np.float32)
    Tcw = np.eye(4, dtype=np.float32)
    Tcw[0, 3] = np.float32(3.0) * np.float32(DEPTH) / np.float32(KITTI_FX)
    npairs = len(frames) // 2

    def one_pair(e, p):
        kl, dl = e.extract(frames[2 * p])
        kc, dc = e.extract(frames[2 * p + 1])
        z = np.float32(DEPTH)
        X = np.stack([(kl["x"] - np.float32(KITTI_CX)) * z / np.float32(KITTI_FX),
                      (kl["y"] - np.float32(KITTI_CY)) * z / np.float32(KITTI_FY), np.full(len(kl), z, np.float32)], 1)
        q = O.project_last_frame(cam, Tcw, Tlw, X, np.full(len(kl), 3, np.uint8), kl, TRACK_TH, True)
        keep = []
        fv = O.make_frame(kc, dc, None, (0.0, 0.0, float(W), float(H)), sf, keep)
        return O.search_by_projection_frame(fv, q, dl, None, True)[0]

    warm = O.OracleExtractor(NFEAT, 1.2, NLEVELS, 20, 7)
    one_pair(warm, 0)
    t0 = time.perf_counter()
    nm = one_pair(warm, 0)                                              # single-thread estimate
    one = time.perf_counter() - t0
    done = [0] * cores
    t_end = [0.0]

    def work(t):                                                        # time-bounded: every thread stops at the budget
        e = O.OracleExtractor(NFEAT, 1.2, NLEVELS, 20, 7)
        i = 0
        while time.perf_counter() < t_end[0]:
            one_pair(e, (t * 7 + i) % npairs)
            done[t] += 2
            i += 1
    th = [threading.Thread(target=work, args=(t,)) for t in range(cores)]
    t0 = time.perf_counter()
    t_end[0] = t0 + budget_s
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    total = sum(done)
    return {"value": round(total / dt, 2), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "%d frames 1241x376 = %d (last, cur) pairs on %d threads (every core this process may use) for %.1f s: "
                      "extract + projection + SearchByProjection per pair (%d matches on pair 0); scalar C "
                      "port built %s -ffp-contract=off on this host, not OpenCV SIMD" %
                      (total, total // 2, cores, dt, nm, flags),
            "single_thread_frames_per_s": round(2.0 / one, 2)}


def host_cores():
    """Cores this process may really use: the affinity mask, capped by the cgroup CPU quota when there is one."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // per))
            break
        except Exception:
            continue
    return max(1, n)


# ---------------------------------------------------------------------------------------------------------------
def self_launch(args):
    """`python bench.py --gpus N` outside torch.distributed.run: bring up N ranks as child processes (this process has
    not touched a GPU and never will) and exit with their status."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def timed(fn, sync, steps, min_s, allreduce_max=None):
    """Run `steps` calls of fn R times so that the timed region lasts >= min_s; returns (seconds, R)."""
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    sync()
    probe = time.perf_counter() - t0
    if allreduce_max is not None:
        probe = allreduce_max(probe)
    reps = max(1, int(np.ceil(min_s / max(probe, 1e-6))))
    sync()
    t0 = time.perf_counter()
    for _ in range(reps * steps):
        fn()
    sync()
    dt = time.perf_counter() - t0
    if allreduce_max is not None:
        dt = allreduce_max(dt)
    return dt, reps


_REAL_STDOUT = None


def emit_json(obj):
    """The one JSON line of the run, on the process's original stdout."""
    line = (json.dumps(obj) + "\n").encode()
    if _REAL_STDOUT is None:
        sys.stdout.write(line.decode()); sys.stdout.flush()
    else:
        os.write(_REAL_STDOUT, line)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--global-batch", type=int, default=GLOBAL_BATCH,
                    help="frames per step over all GPUs (BASELINE.json configs[3]: 512); every GPU gets global/N of them, "
                         "split over --handles pipelines")
    ap.add_argument("--frames-per-gpu", type=int, default=None,
                    help="fix the frames per GPU and step instead (weak scaling: the global batch grows with N)")
    ap.add_argument("--min-time", type=float, default=MIN_TIMED_S,
                    help="minimum length in seconds of every timed region (profiler runs pass 0: one repeat of --steps)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--trace-run", action="store_true",
                    help="for kernel-trace tools: warm-up + K plain steps of the headline (no HIP events between the stages, no "
                         "secondary legs), one short JSON line")
    ap.add_argument("--no-secondary", "--no-match", dest="no_secondary", action="store_true",
                    help="skip the secondary legs (extract-only, stereo, EuRoC initialisation, BoW)")
    ap.add_argument("--dist-backend", default="nccl",
                    help="nccl (= RCCL; default) or gloo (rehearsal of the multi-rank control flow: ranks share cuda:0 "
                         "when there is one, and the gather goes through host memory)")
    ap.add_argument("--dry-run", action="store_true",
                    help="control-flow rehearsal without a GPU (gloo only): spawn, rendezvous, pair sharding and the "
                         "gather run on fabricated result slots; no extraction, no throughput claim")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearsal: run the multi-rank code path (RCCL init, gather, all_reduce) even with one rank")
    ap.add_argument("--out-buffers", type=int, default=4,
                    help="multi-GPU: output sets per rank that rotate between the pipelines and the gather stream (>= 2)")
    ap.add_argument("--gates", default="auto",
                    help="comma-separated extractor stages (0 pyramid, 1 FAST, 2 octree, 3 descriptors) whose launches are chained "
                         "in a ring over the pipelines (orbhip_extractor_set_stage_gate): pipeline h's stage waits for pipeline "
                         "h-1's.  Default \"auto\": the pyramid ring (\"0\") from 384 frames per GPU up, none below (measured on one MI355X: "
                         "512 frames 290 k with the ring / 287 k without; 256: 273 / 276 k; 128: 245 / 261 k; 64: 220 / 227 k frames/s)")
    ap.add_argument("--handles", type=int, default=0,
                    help="pipelines per GPU; the per-GPU batch is split evenly between them and they run concurrently "
                         "on separate HIP streams (extractor + matcher handle each).  Default: 3 from 64 frames per GPU up (64 frames per GPU, "
                         "configs[3] on 8 GPUs, without gates: 227 k frames/s per GPU against 223 k with 2 and 196 k with 1), 2 below")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args))

    # stdout carries the ONE JSON line and nothing else: RCCL prints a version banner and Gloo its rendezvous messages
    # to fd 1 from native code, so fd 1 is pointed at stderr for the lifetime of the run and the line goes to a duplicate
    # of the original descriptor
    global _REAL_STDOUT
    sys.stdout.flush()
    _REAL_STDOUT = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    import __graft_entry__ as g
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    rehearsal = world > 1 and args.dist_backend == "gloo"
    multi = world > 1 or args.force_dist          # take the distributed code path
    if args.dry_run and not (rehearsal or world == 1):
        raise SystemExit("--dry-run needs --dist-backend gloo")
    if not args.dry_run and not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no GPU visible); there is no CPU fallback")
    if rehearsal:
        local_rank = 0
    if args.force_dist and world == 1:
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal or args.dry_run:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    # the library is built by rank 0 only; everybody else waits for it before importing the package
    so_path = os.path.join(ROOT, "orb_slam2_comment_amd", "liborbhip.so")
    if rank == 0 and not os.path.exists(so_path):
        g.build()
    if multi:
        dist.barrier()
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus

    from orb_slam2_comment_amd.sharding import gather_into, gather_to_rank0, shard_indices
    if args.frames_per_gpu is not None:
        B, scaling = args.frames_per_gpu, "weak"          # per-GPU work fixed, the global batch grows with N
    else:
        B, scaling = (args.global_batch // world) & ~1, "strong"   # total work fixed (configs[3]: 512 frames over N GPUs)
    assert B % 2 == 0 and B >= 2, "frames per GPU must be even and >= 2: frames come as (last, cur) pairs"

    if args.dry_run:
        return dry_run(args, dist, torch, world, rank, multi, shard_indices((B // 2) * world, rank, world), gather_to_rank0, B)

    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    ctx = {"args": args, "torch": torch, "dist": dist, "dev": dev, "world": world, "rank": rank, "local_rank": local_rank,
           "multi": multi, "rehearsal": rehearsal, "gather_into": gather_into, "gather_to_rank0": gather_to_rank0}
    head = Headline(ctx, B)
    if args.trace_run:
        res = head.measure(stages=False)
        if rank == 0:
            emit_json({"metric": "trace run (headline steps only)", "value": round(res["fps"], 1), "unit": "frames/s", "n_gpus": world,
                       "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(res["dt_long"] / res["steps_long"] * 1e3, 4)})
        return
    res = head.measure()
    if rank == 0:
        print("[bench] headline done: %.1f frames/s; secondary legs ..." % res["fps"], file=sys.stderr, flush=True)
    secondary = None
    if not args.no_secondary and world == 1 and not multi:
        secondary = secondary_legs(args, torch, dev, local_rank, head)
    weak = None
    if world > 1 and scaling == "strong" and not args.no_secondary:
        # the round-2 workload beside it: 128 frames per GPU whatever N is (weak scaling), never `value`
        head.close()
        w = Headline(ctx, 128)
        wr = w.measure(stages=False)
        weak = {"value": round(wr["fps"], 1), "unit": "frames/s", "scaling": "weak",
                "what": "the same step with 128 frames per GPU (global batch %d)" % (128 * world),
                "ms_per_step": round(wr["dt_long"] / wr["steps_long"] * 1e3, 4), "timed_s": round(wr["dt_long"], 4)}
        w.close()

    if rank == 0:
        stage, stage_alone, Bm = res["stage"], res["stage_alone"], res["frames_per_launch"]
        fps, steps_long, dt_long, dt_k, reps = res["fps"], res["steps_long"], res["dt_long"], res["dt_k"], res["reps"]
        fps_k = B * world * args.steps / dt_k                   # exactly K steps
        # dominant single kernel: the largest launch among the one-launch stages when a pipeline runs alone (a property of
        # the kernel: rocprofv3 --stats of a single pipeline has k_fast_cells at 30 % of the GPU time, k_describe_fused at
        # 21 %; how much a launch is stretched while other pipelines share the GPU is a property of the schedule).  Its
        # launch time in the timed region is what `achieved` uses.
        kern = max(("fast", "describe"), key=lambda k: stage_alone[k])
        # frames per launch: the pipelines get 172 / 170 / 170 frames of a 512-frame step; stage times are means over
        # the pipelines, so the bytes are those of the MEAN launch
        algo = ALGO_BYTES[kern] * Bm
        achieved = algo / (stage[kern] * 1e-6) / 1e9
        traffic, valu = None, None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                pm = json.load(open(tpath))[kern]
                traffic = round(pm["hbm_bytes_per_frame"] * Bm)
                if "valu_issue_us_per_frame" in pm:
                    # the kernel is priced against HBM as the contract asks, but what bounds it is VALU issue: PMC
                    # instruction count x the measured issue cost of its instruction mix (tools/micro)
                    vi = pm["valu_issue_us_per_frame"] * Bm
                    valu = {"insts_per_launch": pm["valu_insts_per_frame"] * Bm, "issue_us_per_launch": round(vi, 1),
                            "frac_of_launch": round(vi / stage[kern], 3), "model": pm.get("valu_issue_model")}
            except Exception:
                traffic, valu = None, None
        try:
            metric_name = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
        except Exception:
            metric_name = "frames/sec ORB extract+match, KITTI 1241x376 @1000 feat; HBM GB/s vs peak"
        other_denoms = {"pyramid": "SURVEY 8d K1 without the level-0 copy, which these monocular handles do not execute: (SumP - P7) read "
                                   "+ (SumP - P0) written (the full K1 figure, P0 + (SumP - P7) + SumP, applies to the stereo leg)",
                        "describe": "SURVEY 8d K5+K7: N*(749+512) read + N*60 written; K6's 2*SumP is not moved (the blur is fused) "
                                    "and not counted"}
        out = {
            "metric": metric_name,
            "value": round(fps, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt_long / steps_long * 1e3, 4),
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": "configs[1]+[2]+[3]: a global batch of %d synthetic 1241x376 u8 frames per step, %d per GPU, resident in "
                                   "HBM as (last, cur) pairs; per step ORB extract (nFeatures=1000, 8-level pyramid) of every frame + "
                                   "device-resident SearchByProjection(CurrentFrame, LastFrame, th=15, mono; projection prologue "
                                   "included) for every pair; monocular handles: mvImagePyramid[0] on demand "
                                   "(orbhip_extractor_set_lazy_level0), never asked for here" % (B * world, B) +
                                   ("; pairs round-robin over the GPUs + RCCL gather of the result slots to rank 0" if world > 1 else ""),
                       "frames_per_gpu_per_step": B, "pairs_per_gpu_per_step": B // 2, "handles_per_gpu": head.Hn, "stage_gates": head.gates,
                       "frames_per_launch": head.splits, "mean_frames_per_launch": Bm, "global_batch": B * world, "nfeatures": NFEAT,
                       "levels": NLEVELS, "mean_keypoints_per_frame": res["mean_kp"],
                       "mean_matches_per_pair": res["mean_matches"],
                       "parallelism": "pair-sharded x%d" % world,
                       "gather_bytes_per_rank_per_step": head.flat_bytes if multi else 0},
            "timing": {"timed_steps": steps_long, "timed_s": round(dt_long, 4), "timed_repeats_of_steps": reps,
                       "exactly_k_steps": {"steps": args.steps, "s": round(dt_k, 5), "ms_per_step": round(dt_k / args.steps * 1e3, 4),
                                           "frames_per_s": round(fps_k, 1)}},
            "roofline": {"bound": "hbm", "kernel": KERNEL_NAME[kern],
                         "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic, "valu_issue": valu,
                         "algorithmic_bytes_per_launch": round(algo),
                         "avg_launch_us": round(stage[kern], 2),
                         "alone": {"what": "a launch of pipeline 0 (%d frames) with no other pipeline on the GPU" % head.splits[0],
                                   "avg_launch_us": round(stage_alone[kern], 2),
                                   "achieved": round(ALGO_BYTES[kern] * head.splits[0] / (stage_alone[kern] * 1e-6) / 1e9, 2),
                                   "frac": round(ALGO_BYTES[kern] * head.splits[0] / (stage_alone[kern] * 1e-6) / 1e9 / HBM_PEAK_GBPS, 5),
                                   "stage_us": {k: round(v, 2) for k, v in stage_alone.items()}},
                         "other_stages": {k: {"kernel": KERNEL_NAME[k], "algorithmic_bytes_per_launch": round(ALGO_BYTES[k] * Bm),
                                              "denominator": other_denoms[k],
                                              "frac": round(ALGO_BYTES[k] * Bm / (stage[k] * 1e-6) / 1e9 / HBM_PEAK_GBPS, 5),
                                              "frac_alone": round(ALGO_BYTES[k] * head.splits[0] / (stage_alone[k] * 1e-6) / 1e9 / HBM_PEAK_GBPS, 5)}
                                          for k in ("pyramid", "describe") if stage[k] > 0 and stage_alone[k] > 0},
                         "stage_us": {k: round(v, 2) for k, v in stage.items()},
                         "whole_path_GBps_model": round(FRAME_BYTES_MODEL * fps / world / 1e9, 2)},
        }
        if weak:
            out["weak_128_per_gpu"] = weak
        if secondary:
            out.update(secondary)
        if not args.no_cpu_baseline and world == 1:
            print("[bench] GPU legs done; cpu_baseline (~25 s) ...", file=sys.stderr, flush=True)
            out["cpu_baseline"] = cpu_baseline(head.frames[:16])
        else:
            out["cpu_baseline"] = None
        emit_json(out)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------------------------
class Headline:
    """The headline step for B frames per GPU: pipelines (extractor + matcher + stream each), resident frames, the
    synthetic map, rotating output sets and, with several ranks, the gather of step k beside the work of step k+1."""

    def __init__(self, ctx, B):
        self.ctx, self.B = ctx, B
        args, torch, dev, world = ctx["args"], ctx["torch"], ctx["dev"], ctx["world"]
        local_rank, multi, rehearsal = ctx["local_rank"], ctx["multi"], ctx["rehearsal"]
        from orb_slam2_comment_amd import ORBextractor, ORBmatcher
        from orb_slam2_comment_amd import matcher as M
        from orb_slam2_comment_amd.capi import POINT_OBSERVED, POINT_PRESENT
        from orb_slam2_comment_amd.synth import synth_frame
        pairs_local = B // 2
        # local frames: pair k = (scene s, scene s translated by 3 px): 8 scenes -> 16 distinct images per rank
        uniq = {}
        for li in range(min(B, 16)):
            key = (1 + (li // 2) % 8, li % 2)
            uniq[key] = synth_frame(key[0], W, H, shift_xy=(3 * key[1], 0))
        self.frames = frames = np.stack([uniq[(1 + (li // 2) % 8, li % 2)] for li in range(B)])
        self.d_img = d_img = torch.from_numpy(frames).to(dev)

        want = args.handles if args.handles > 0 else (3 if B >= 64 else 2)   # 64 / 128 frames: 3 pipelines 227 / 261 k, 2: 223 / 259 k
        self.Hn = Hn = max(1, min(want, pairs_local))
        self.psplit = psplit = [pairs_local // Hn + (1 if h < pairs_local % Hn else 0) for h in range(Hn)]   # pairs per pipeline
        self.splits = splits = [2 * p for p in psplit]
        self.offs = offs = [sum(splits[:h]) for h in range(Hn)]
        self.exts, self.mts, self.streams = [], [], []
        for h in range(Hn):
            e = ORBextractor(NFEAT, 1.2, NLEVELS, 20, 7, device=local_rank)
            e.set_lazy_level0(True)          # a monocular Tracking thread never reads mvImagePyramid[0] (src/Frame.cc:174-228)
            mt = ORBmatcher(0.9, True, device=local_rank)
            st = torch.cuda.Stream(dev)              # one real (non-null) stream per pipeline: a null handle would mean
            e.set_stream(st.cuda_stream)             # "the handle's own stream", and extractor and matcher must share one
            mt.set_stream(st.cuda_stream)
            self.exts.append(e); self.mts.append(mt); self.streams.append(st)
        exts, mts, streams = self.exts, self.mts, self.streams
        # stage gates: a ring of events per gated stage (torch creates an event's handle at its first record)
        self.gate_events = []
        gates = ("0" if B >= 384 else "") if args.gates == "auto" else args.gates
        self.gates = gates
        for spec in [v.strip() for v in gates.split(",") if v.strip() != ""] if Hn > 1 else []:
            # "a" = stage a of pipeline h waits for stage a of pipeline h-1; "a:b" = ... for stage b of pipeline h-1
            a, b = (int(v) for v in (spec.split(":") if ":" in spec else (spec, spec)))
            evs = [torch.cuda.Event() for _ in range(Hn)]
            for h in range(Hn):
                evs[h].record(streams[h])
            torch.cuda.synchronize(dev)
            for h in range(Hn):
                exts[h].set_stage_gate_wait(a, evs[(h - 1) % Hn].cuda_event)
                exts[h].set_stage_gate_record(b, evs[h].cuda_event)
            self.gate_events.append(evs)
        self.cap = cap = exts[0].capacity(H, W)
        sf = exts[0].GetScaleFactors()
        cam = M.make_camera(KITTI_FX, KITTI_FY, KITTI_CX, KITTI_CY, (0.0, 0.0, float(W), float(H)), sf, mbf=KITTI_BF,
                            mb=KITTI_BF / KITTI_FX)
        # one flat allocation per output set: {KeyPoint[B][cap] | desc[B][cap][32] | count[B] | assign[B/2][cap] |
        # nmatches[B/2]} are views of it, so the multi-GPU exchange is ONE gather of one contiguous buffer per step
        nb_k, nb_d, nb_a = B * cap * 28, B * cap * 32, pairs_local * cap * 4
        off_d = (nb_k + 255) & ~255                      # every view starts 256-byte aligned
        off_n = (off_d + nb_d + 255) & ~255
        off_a = (off_n + B * 4 + 255) & ~255
        off_m = (off_a + nb_a + 255) & ~255
        self.flat_bytes = flat_bytes = off_m + pairs_local * 4

        def out_set():
            flat = torch.zeros(flat_bytes, dtype=torch.uint8, device=dev)
            return {"k": flat[:nb_k].view(torch.int32).view(B, cap, 7), "d": flat[off_d:off_d + nb_d].view(B, cap, 32),
                    "n": flat[off_n:off_n + B * 4].view(torch.int32), "a": flat[off_a:off_a + nb_a].view(torch.int32).view(pairs_local, cap),
                    "m": flat[off_m:off_m + pairs_local * 4].view(torch.int32), "st": torch.zeros(B, dtype=torch.int32, device=dev),
                    "flat": flat}

        # the synthetic map: every last-frame keypoint carries a map point at depth DEPTH in the last camera's frame
        # (Tlw = I); the current camera is translated so that such a point moves by 3 px -- the shift between the two
        # rendered views.  World positions are rebuilt from the keypoints of each extraction ON THE DEVICE by a tiny
        # torch expression (input synthesis for the benchmark, the stand-in for the map that Tracking would hold).
        Tlw = torch.eye(4, dtype=torch.float32)[:3, :].reshape(1, 12).repeat(pairs_local, 1).contiguous().to(dev)
        Tc = torch.eye(4, dtype=torch.float32)
        Tc[0, 3] = float(np.float32(3.0) * np.float32(DEPTH) / np.float32(KITTI_FX))
        Tcw = Tc[:3, :].reshape(1, 12).repeat(pairs_local, 1).contiguous().to(dev)
        d_world = torch.zeros((B, cap, 3), dtype=torch.float32, device=dev)
        d_flags = torch.full((B, cap), POINT_PRESENT | POINT_OBSERVED, dtype=torch.uint8, device=dev)

        def build_map(o):
            kf = o["k"][0::2].view(torch.float32)                       # last frames of the pairs
            z = float(DEPTH)
            d_world[0::2, :, 0] = (kf[..., 0] - KITTI_CX) * (z / KITTI_FX)
            d_world[0::2, :, 1] = (kf[..., 1] - KITTI_CY) * (z / KITTI_FY)
            d_world[0::2, :, 2] = z

        self.obuf = obuf = [out_set()]
        nbuf = args.out_buffers if multi and not rehearsal else 1
        for _ in range(nbuf - 1):
            obuf.append(out_set())
        gstream = torch.cuda.Stream(dev) if nbuf >= 2 else None
        gdone = [None] * nbuf
        gout = None
        if nbuf >= 2 and ctx["rank"] == 0:
            gout = [torch.zeros((world, flat_bytes), dtype=torch.uint8, device=dev)]   # rank-major, same carve-up per rank
        stepno = [0]
        # torch events around the matching of pipeline 0 (recorded on the stream its launches go to): a ring of 256 pairs
        self.match_ev = match_ev = [[torch.cuda.Event(enable_timing=True) for _ in range(256)] for _ in range(2)]
        self.record_match = record_match = [False]
        self.match_calls = match_calls = [0]
        gather_into, gather_to_rank0 = ctx["gather_into"], ctx["gather_to_rank0"]

        def extract_all(o, which=None):
            for h, e in enumerate(exts):
                if which is not None and h != which:
                    continue
                sl = slice(offs[h], offs[h] + splits[h])
                e.extract_batch_device(d_img[sl].data_ptr(), splits[h], H, W, o["k"][sl].data_ptr(), o["d"][sl].data_ptr(),
                                       cap, o["n"][sl].data_ptr(), o["st"][sl].data_ptr())

        def match_all(o):
            for h, mt in enumerate(mts):
                f0, p0 = offs[h], offs[h] // 2
                if record_match[0] and h == 0:
                    match_ev[0][match_calls[0] % 256].record(streams[0])
                # pair j of this pipeline: LastFrame = frame f0 + 2j, CurrentFrame = f0 + 2j + 1 of the shared arrays
                mt.TrackLastFrameDevice(psplit[h], cam, Tcw[p0:].data_ptr(), Tlw[p0:].data_ptr(), o["k"].data_ptr(),
                                        o["d"].data_ptr(), o["n"].data_ptr(), cap, f0 + 1, 2, f0, 2, d_world.data_ptr(),
                                        d_flags.data_ptr(), TRACK_TH, True, o["a"][p0:].data_ptr(), o["m"][p0:].data_ptr())
                if record_match[0] and h == 0:
                    match_ev[1][match_calls[0] % 256].record(streams[0])
                    match_calls[0] += 1

        def step():
            k = stepno[0] % nbuf
            stepno[0] += 1
            o = obuf[k]
            if gdone[k] is not None:                 # the gather that last read this buffer must be finished
                for st in streams:
                    st.wait_event(gdone[k])
            extract_all(o)
            match_all(o)
            if multi:
                if rehearsal:
                    torch.cuda.synchronize(dev)
                    return gather_to_rank0(o["k"].cpu(), o["d"].cpu(), o["n"].cpu())
                for st in streams:
                    gstream.wait_stream(st)
                with torch.cuda.stream(gstream):
                    out = gather_into(gout, (o["flat"],))
                    gdone[k] = gstream.record_event()
                return out
            return None

        self.extract_all, self.step = extract_all, step
        # the map of every output set is built once from a first extraction (keypoints are deterministic per image)
        for o in obuf:
            extract_all(o)
            torch.cuda.synchronize(dev)
            build_map(o)
        torch.cuda.synchronize(dev)

    def barrier(self):
        torch, dev = self.ctx["torch"], self.ctx["dev"]
        torch.cuda.synchronize(dev)              # drains the pipeline streams and the gather stream
        if self.ctx["multi"]:
            self.ctx["dist"].barrier()
        torch.cuda.synchronize(dev)

    def allreduce_max(self, v):
        if not self.ctx["multi"]:
            return v
        torch = self.ctx["torch"]
        t = torch.tensor([v], dtype=torch.float64, device="cpu" if self.ctx["rehearsal"] else self.ctx["dev"])
        self.ctx["dist"].all_reduce(t, op=self.ctx["dist"].ReduceOp.MAX)
        return float(t.item())

    def measure(self, stages=True):
        args, torch, dev, world = self.ctx["args"], self.ctx["torch"], self.ctx["dev"], self.ctx["world"]
        exts, step, barrier, allreduce_max = self.exts, self.step, self.barrier, self.allreduce_max
        for _ in range(args.warmup):
            step()
        barrier()
        # ---- the measurement the contract asks for: EXACTLY K steps between barriers -> ms_per_step, value ----------
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        dt_k = allreduce_max(time.perf_counter() - t0)
        # ---- the same steps again, R x K of them (>= 1 s), with the per-stage HIP events on: stage_us / roofline -----
        if stages:
            for e in exts:
                e.set_profiling(True)                   # HIP events around every stage, on the launch stream
            self.record_match[0] = True
        reps = max(1, int(np.ceil(args.min_time / max(dt_k, 1e-6))))
        reps = int(allreduce_max(float(reps)))
        barrier()
        t0 = time.perf_counter()
        for _ in range(reps * args.steps):
            step()
        barrier()
        dt_long = allreduce_max(time.perf_counter() - t0)
        steps_long = reps * args.steps
        res = {"dt_k": dt_k, "dt_long": dt_long, "reps": reps, "steps_long": steps_long,
               "fps": self.B * world * steps_long / dt_long}                      # the >= 1 s region; `value`
        o = self.obuf[0]
        n_host, nm_host = o["n"].cpu().numpy(), o["m"].cpu().numpy()
        assert int(o["st"].abs().sum().item()) == 0 and n_host.min() > 0, "extraction failed"
        assert nm_host.min() > 100, "SearchByProjection found too few matches: the synthetic pairs are broken"
        res["mean_kp"], res["mean_matches"] = round(float(n_host.mean()), 1), round(float(nm_host.mean()), 1)
        if not stages:
            return res
        self.record_match[0] = False
        st_all = [e.stage_times_us() for e in exts]
        stage = {k: sum(st[k] for st in st_all) / len(st_all) for k in st_all[0]}
        for e in exts:
            e.set_profiling(False)
        nev = min(self.match_calls[0], 256)
        stage["match"] = float(np.mean([self.match_ev[0][i].elapsed_time(self.match_ev[1][i]) for i in range(nev)])) * 1e3 if nev else 0.0
        # pipeline 0 once more, ALONE on the GPU (the other pipelines idle): the stand-alone launch times of its kernels, which
        # is what rocprofv3 shows for a single-pipeline run; reported beside the contended times of the timed region above
        torch.cuda.synchronize(dev)
        exts[0].set_profiling(True)
        for _ in range(40):
            self.extract_all(self.obuf[0], which=0)
        torch.cuda.synchronize(dev)
        res["stage_alone"] = exts[0].stage_times_us()
        exts[0].set_profiling(False)
        res["stage"] = stage
        res["frames_per_launch"] = sum(self.splits) / len(self.splits)
        return res

    def close(self):
        self.ctx["torch"].cuda.synchronize(self.ctx["dev"])
        for e in self.exts:
            e.close()
        self.exts, self.mts, self.obuf, self.d_img = [], [], [], None
        self.ctx["torch"].cuda.empty_cache()


# ---------------------------------------------------------------------------------------------------------------
def secondary_legs(args, torch, dev, local_rank, head):
    """Named secondary measurements (single GPU).  Each leg: warm-up, then a timed region of >= MIN_TIMED_S."""
    from orb_slam2_comment_amd import ORBextractor, ORBmatcher, ORBVocabulary
    from orb_slam2_comment_amd import matcher as M
    from orb_slam2_comment_amd.capi import POINT_OBSERVED, POINT_PRESENT
    from orb_slam2_comment_amd.synth import synth_frame, synth_stereo, synth_vocabulary
    exts, o, d_img, cap, extract_all = head.exts, head.obuf[0], head.d_img, head.cap, head.extract_all
    B = min(d_img.shape[0], 128)              # frames per step of the secondary legs
    out = {}

    def sync():
        torch.cuda.synchronize(dev)

    def leg(fn, steps):
        for _ in range(3):
            fn()
        return timed(fn, sync, steps, args.min_time)

    # ---- configs[1]: extract-only ------------------------------------------------------------------------------
    for e in exts:
        e.set_profiling(True)
    dt, reps = leg(lambda: extract_all(o), args.steps)
    stages = [e.stage_times_us() for e in exts]
    st = {k: round(sum(s[k] for s in stages) / len(stages), 2) for k in stages[0]}
    for e in exts:
        e.set_profiling(False)
    out["extract_only"] = {"value": round(head.B * reps * args.steps / dt, 1), "unit": "frames/s",
                           "what": "configs[1]: ORBextractor::operator() over %d resident 1241x376 frames per step "
                                   "(%d pipelines, monocular handles: mvImagePyramid[0] on demand), nothing else" % (head.B, len(exts)),
                           "ms_per_step": round(dt / (reps * args.steps) * 1e3, 4), "timed_s": round(dt, 3), "stage_us": st}

    cur = torch.cuda.Stream(dev)               # every handle of the legs below launches on this one stream
    mt = ORBmatcher(0.9, True, device=local_rank)
    mt.set_stream(cur.cuda_stream)
    # ---- configs[2] as ONE leg: the stereo Tracking step.  Per step Bs stereo frames: the two extractors of a stereo
    # sensor on two streams (src/Frame.cc:78-81 runs them on two threads), Frame::ComputeStereoMatches for every frame,
    # then SearchByProjection(CurrentFrame, LastFrame, th = 7, bMono = false) with the `ur` gate for every (t, t+1) pair
    # (src/Tracking.cc:880-885, src/ORBmatcher.cc:1407-1413).  The map: every last-frame keypoint with a stereo depth
    # carries a map point at that depth; the current camera is the last one rotated about y by 3 px / fx, so that
    # projections land on the scene's 3-px shift whatever the depth and the projected `ur` keeps the measured disparity.
    Bs = B // 2
    lefts, rights = [], []
    for p in range(4):
        for sh in (0, 3):
            l, r = synth_stereo(1 + p, W, H, shift_xy=(sh, 0))
            lefts.append(l); rights.append(r)
    d_left = torch.from_numpy(np.stack([lefts[i % 8] for i in range(Bs)])).to(dev)
    d_right = torch.from_numpy(np.stack([rights[i % 8] for i in range(Bs)])).to(dev)
    sL, sR = cur, torch.cuda.Stream(dev)
    eL = ORBextractor(NFEAT, 1.2, NLEVELS, 20, 7, device=local_rank)
    eR = ORBextractor(NFEAT, 1.2, NLEVELS, 20, 7, device=local_rank)
    eL.set_stream(sL.cuda_stream); eR.set_stream(sR.cuda_stream)
    so = {}
    for side in "LR":
        so["k" + side] = torch.zeros((Bs, cap, 7), dtype=torch.int32, device=dev)
        so["d" + side] = torch.zeros((Bs, cap, 32), dtype=torch.uint8, device=dev)
        so["n" + side] = torch.zeros(Bs, dtype=torch.int32, device=dev)
        so["s" + side] = torch.zeros(Bs, dtype=torch.int32, device=dev)
    s_ur = torch.full((Bs, cap), -1.0, dtype=torch.float32, device=dev)
    s_dp = torch.full((Bs, cap), -1.0, dtype=torch.float32, device=dev)
    s_nm = torch.zeros(Bs, dtype=torch.int32, device=dev)
    t_a = torch.zeros((Bs // 2, cap), dtype=torch.int32, device=dev)
    t_m = torch.zeros(Bs // 2, dtype=torch.int32, device=dev)
    mbf = KITTI_BF
    mb = mbf / KITTI_FX                        # Examples/Stereo/KITTI00-02.yaml:8,25
    scam = M.make_camera(KITTI_FX, KITTI_FY, KITTI_CX, KITTI_CY, (0.0, 0.0, float(W), float(H)), eL.GetScaleFactors(),
                         mbf=mbf, mb=mb)
    th_y = 3.0 / KITTI_FX
    Tc = torch.eye(4, dtype=torch.float32)
    Tc[0, 0] = Tc[2, 2] = float(np.cos(th_y)); Tc[0, 2] = float(np.sin(th_y)); Tc[2, 0] = -float(np.sin(th_y))
    sTcw = Tc[:3, :].reshape(1, 12).repeat(Bs // 2, 1).contiguous().to(dev)
    sTlw = torch.eye(4, dtype=torch.float32)[:3, :].reshape(1, 12).repeat(Bs // 2, 1).contiguous().to(dev)
    s_world = torch.zeros((Bs, cap, 3), dtype=torch.float32, device=dev)
    s_flags = torch.zeros((Bs, cap), dtype=torch.uint8, device=dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]

    def stereo_extract_and_match():
        sR.wait_stream(sL)                     # the right extraction of step k+1 must not overtake the stereo search of step k
        eL.extract_batch_device(d_left.data_ptr(), Bs, H, W, so["kL"].data_ptr(), so["dL"].data_ptr(), cap, so["nL"].data_ptr(),
                                so["sL"].data_ptr())
        eR.extract_batch_device(d_right.data_ptr(), Bs, H, W, so["kR"].data_ptr(), so["dR"].data_ptr(), cap, so["nR"].data_ptr(),
                                so["sR"].data_ptr())
        sL.wait_stream(sR)
        ev[1].record(sL)
        mt.ComputeStereoMatchesDevice(eL, 0, 1, eR, 0, 1, Bs, so["kL"].data_ptr(), so["dL"].data_ptr(), so["nL"].data_ptr(),
                                      so["kR"].data_ptr(), so["dR"].data_ptr(), so["nR"].data_ptr(), cap, mbf, mb,
                                      s_ur.data_ptr(), s_dp.data_ptr(), s_nm.data_ptr())
        ev[2].record(sL)

    def stereo_step():
        ev[0].record(sL)
        stereo_extract_and_match()
        # pair j: LastFrame = left frame 2j, CurrentFrame = left frame 2j + 1; mvuRight of the current frame gates `ur`
        mt.TrackLastFrameDevice(Bs // 2, scam, sTcw.data_ptr(), sTlw.data_ptr(), so["kL"].data_ptr(), so["dL"].data_ptr(),
                                so["nL"].data_ptr(), cap, 1, 2, 0, 2, s_world.data_ptr(), s_flags.data_ptr(), STEREO_TH, False,
                                t_a.data_ptr(), t_m.data_ptr(), d_u_right=s_ur.data_ptr())
        ev[3].record(sL)

    stereo_extract_and_match()                 # the map, once (input synthesis): unproject the last frames' keypoints
    sync()
    kf = so["kL"].view(torch.float32)
    valid = (s_dp > 0) & (torch.arange(cap, device=dev)[None, :] < so["nL"][:, None])
    z = torch.where(valid, s_dp, torch.ones_like(s_dp))
    s_world[..., 0] = (kf[..., 0] - KITTI_CX) * z / KITTI_FX
    s_world[..., 1] = (kf[..., 1] - KITTI_CY) * z / KITTI_FY
    s_world[..., 2] = z
    s_flags[:] = torch.where(valid, POINT_PRESENT | POINT_OBSERVED, 0).to(torch.uint8)
    dts, reps = leg(stereo_step, args.steps)
    sync()
    assert int(so["sL"].abs().sum().item()) == 0 and int(so["sR"].abs().sum().item()) == 0, "stereo extraction failed"
    tm = float(t_m.float().mean().item())
    assert tm > 50, "stereo SearchByProjection found too few matches (%.1f): the synthetic stereo sequence is broken" % tm
    out["stereo"] = {"value": round(Bs * reps * args.steps / dts, 1), "unit": "stereo frames/s",
                     "what": "configs[2] as one leg: per step %d stereo frames 1241x376: extract left and right (two handles, two "
                             "streams) + device-resident Frame::ComputeStereoMatches (fx 718.856, bf 386.1448) + device-resident "
                             "SearchByProjection(CurrentFrame, LastFrame, th=7, bMono=false, `ur` gate on mvuRight) for the %d "
                             "(t, t+1) pairs; map points at the stereo depth of the last frame" % (Bs, Bs // 2),
                     "ms_per_step": round(dts / (reps * args.steps) * 1e3, 4), "timed_s": round(dts, 3),
                     "stage_us": {"extract_left_and_right": round(ev[0].elapsed_time(ev[1]) * 1e3, 1),
                                  "stereo_match": round(ev[1].elapsed_time(ev[2]) * 1e3, 1),
                                  "search_by_projection": round(ev[2].elapsed_time(ev[3]) * 1e3, 1)},
                     "mean_stereo_matches_per_frame": round(float(s_nm.float().mean().item()), 1),
                     "mean_projection_matches_per_pair": round(tm, 1)}
    eR.set_stream(0)
    sext, d_simg, Bb = eL, d_img, B            # the BoW leg below extracts the monocular frames with this handle (stream `cur`)

    # ---- configs[4]: EuRoC 752x480 @2000 + SearchForInitialization(windowSize 100, nnratio 0.9) ------------------
    EW, EH, ENF = 752, 480, 2000
    Be = 64
    eframes = np.stack([synth_frame(20 + (i // 2) % 8, EW, EH, shift_xy=(5 * (i % 2), 0)) for i in range(Be)])
    d_eimg = torch.from_numpy(eframes).to(dev)
    eext = ORBextractor(ENF, 1.2, NLEVELS, 20, 7, device=local_rank)       # mpIniORBextractor, src/Tracking.cc:125
    eext.set_stream(cur.cuda_stream)
    ecap = eext.capacity(EH, EW)
    e_k = torch.zeros((Be, ecap, 7), dtype=torch.int32, device=dev)
    e_d = torch.zeros((Be, ecap, 32), dtype=torch.uint8, device=dev)
    e_n = torch.zeros(Be, dtype=torch.int32, device=dev)
    e_st = torch.zeros(Be, dtype=torch.int32, device=dev)
    e_prev = torch.zeros((Be // 2, ecap, 2), dtype=torch.float32, device=dev)
    e_m12 = torch.zeros((Be // 2, ecap), dtype=torch.int32, device=dev)
    e_nm = torch.zeros(Be // 2, dtype=torch.int32, device=dev)
    im = ORBmatcher(0.9, True, device=local_rank)                            # src/Tracking.cc:599
    im.set_stream(cur.cuda_stream)
    eev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]

    def euroc_step():
        eev[0].record(cur)
        eext.extract_batch_device(d_eimg.data_ptr(), Be, EH, EW, e_k.data_ptr(), e_d.data_ptr(), ecap, e_n.data_ptr(),
                                  e_st.data_ptr())
        eev[1].record(cur)
        # vbPrevMatched = the reference frame's keypoint positions (src/Tracking.cc:578-580), set on the device
        im.SearchForInitializationDevice(Be // 2, e_k.data_ptr(), e_d.data_ptr(), e_n.data_ptr(), ecap, 0, 2, 1, 2,
                                         (0.0, 0.0, float(EW), float(EH)), 0, e_prev.data_ptr(), 100, e_m12.data_ptr(),
                                         e_nm.data_ptr())
        eev[2].record(cur)
    dte, reps = leg(euroc_step, args.steps)
    sync()
    out["euroc_init"] = {"value": round(Be * reps * args.steps / dte, 1), "unit": "frames/s",
                         "what": "configs[4]: extract %d resident 752x480 frames at nFeatures=2000 (one pipeline) + device-resident "
                                 "SearchForInitialization(windowSize=100, nnratio 0.9) for the %d (reference, current) pairs "
                                 "(src/Tracking.cc:599-600)" % (Be, Be // 2),
                         "ms_per_step": round(dte / (reps * args.steps) * 1e3, 4), "timed_s": round(dte, 3),
                         "stage_us": {"extract": round(eev[0].elapsed_time(eev[1]) * 1e3, 1),
                                      "search_for_initialization": round(eev[1].elapsed_time(eev[2]) * 1e3, 1)},
                         "mean_keypoints_per_frame": round(float(e_n.float().mean().item()), 1),
                         "mean_matches_per_pair": round(float(e_nm.float().mean().item()), 1)}

    # ---- relocalisation / reference-key-frame chain: extract + Frame::ComputeBoW + SearchByBoW --------------------
    voc = ORBVocabulary.from_arrays(10, 6, 0, 0, *synth_vocabulary(10, 6, 1), device=local_rank)
    voc.set_stream(cur.cuda_stream)
    b_word = torch.zeros((B, cap), dtype=torch.int32, device=dev)
    b_node = torch.zeros((B, cap), dtype=torch.int32, device=dev)
    b_ids = torch.zeros((B, cap), dtype=torch.int32, device=dev)
    b_w = torch.zeros((B, cap), dtype=torch.float64, device=dev)
    b_vals = torch.zeros((B, cap), dtype=torch.float64, device=dev)
    b_n = torch.zeros(B, dtype=torch.int32, device=dev)
    bm = ORBmatcher(0.7, True, device=local_rank)      # Tracking::TrackReferenceKeyFrame, src/Tracking.cc:759
    bm.set_stream(cur.cuda_stream)
    b_m12 = torch.zeros((B // 2, cap), dtype=torch.int32, device=dev)
    b_nm = torch.zeros(B // 2, dtype=torch.int32, device=dev)

    def bow_step():
        sext.extract_batch_device(d_simg.data_ptr(), Bb, H, W, o["k"].data_ptr(), o["d"].data_ptr(), cap,
                                  o["n"].data_ptr(), o["st"].data_ptr())
        voc.transform_device(Bb, o["d"].data_ptr(), o["n"].data_ptr(), cap, 4, b_word.data_ptr(), b_w.data_ptr(),
                             b_node.data_ptr(), b_ids.data_ptr(), b_vals.data_ptr(), b_n.data_ptr())
        side = (o["k"].data_ptr(), o["d"].data_ptr(), o["n"].data_ptr(), b_node.data_ptr())
        bm.SearchByBoWDevice(B // 2, cap, side, 0, 2, side, 1, 2, b_m12.data_ptr(), b_nm.data_ptr(), 50)
    dtb, reps = leg(bow_step, args.steps)
    sync()
    out["bow"] = {"value": round(B * reps * args.steps / dtb, 1), "unit": "frames/s",
                  "what": "extract (%d frames, one pipeline) + device-resident Frame::ComputeBoW (ORBVocabulary::transform, "
                          "synthetic k=10 L=6 tree, levelsup 4) + device-resident SearchByBoW for the %d (2k, 2k+1) pairs" % (B, B // 2),
                  "mean_bow_matches_per_pair": round(float(b_nm.float().mean().item()), 1),
                  "ms_per_step": round(dtb / (reps * args.steps) * 1e3, 4), "timed_s": round(dtb, 3),
                  "mean_bow_words_per_frame": round(float(b_n.float().mean().item()), 1)}
    voc.set_stream(0)
    return out


# ---------------------------------------------------------------------------------------------------------------
def dry_run(args, dist, torch, world, rank, multi, my_pairs, gather_to_rank0, B):
    """GPU-less rehearsal of the multi-rank control flow: pair sharding + gather of fabricated result slots."""
    cap = 64

    def fake(gframe):
        rng = np.random.default_rng(1000 + gframe)
        n = int(rng.integers(cap // 2, cap))
        k = np.zeros((cap, 7), np.int32); d = np.zeros((cap, 32), np.uint8)
        k[:n] = rng.integers(0, 1 << 20, (n, 7)); d[:n] = rng.integers(0, 256, (n, 32))
        return k, d, n
    gframes = [2 * p + j for p in my_pairs for j in (0, 1)]      # global frame ids of this rank, pair-major
    res = [fake(gf) for gf in gframes]
    kps = torch.from_numpy(np.stack([r[0] for r in res]))
    desc = torch.from_numpy(np.stack([r[1] for r in res]))
    cnt = torch.tensor([r[2] for r in res], dtype=torch.int32)
    ok = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = gather_to_rank0(kps, desc, cnt) if multi else (kps, desc, cnt)
        if rank == 0:
            K, D, N = out
            # gather_to_rank0 interleaves ranks per local index: local frame j of rank r sits at j*world + r
            for j, _ in enumerate(gframes):
                for r in range(world):
                    pj = (j // 2) * world + r                    # global pair of (rank r, local frame j)
                    k, d, n = fake(2 * pj + (j % 2))
                    ok &= int(N[j * world + r]) == n and np.array_equal(K[j * world + r].numpy(), k)
    dt = time.perf_counter() - t0
    if multi:
        dist.barrier()
    if rank == 0:
        emit_json({"metric": "dry run: control flow only, no extraction", "value": None, "unit": "frames/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / max(args.steps, 1) * 1e3, 4),
                          "higher_is_better": True, "scaling": "weak" if args.frames_per_gpu is not None else "strong", "vs_baseline": None,
                          "dtype": "u8", "data": "synthetic",
                          "config": {"workload": "dry run (gloo, no GPU): pair sharding + gather of fabricated slots",
                                     "pairs_per_rank": len(my_pairs), "frames_per_gpu_per_step": B, "global_batch": B * world,
                                     "parallelism": "pair-sharded x%d" % world},
                          "gather_verified": bool(ok)})
    if multi:
        dist.destroy_process_group()
    if rank == 0 and not ok:
        raise SystemExit(1)


if __name__ == "__main__":
    main()
