"""world_size-2 gloo test of the frame sharding + gather-to-rank-0 path (CPU tensors)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_result(g, cap):
    """Deterministic stand-in for the extraction result of global frame g."""
    rng = np.random.default_rng(1000 + g)
    n = int(rng.integers(cap // 2, cap))
    k = np.zeros((cap, 7), np.int32)
    d = np.zeros((cap, 32), np.uint8)
    k[:n] = rng.integers(0, 1 << 20, (n, 7))
    d[:n] = rng.integers(0, 256, (n, 32))
    return k, d, n


def _worker(rank, world, port, nframes, cap, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from orb_slam2_comment_amd.sharding import gather_into, gather_to_rank0, shard_indices
    mine = shard_indices(nframes, rank, world)
    res = [_fake_result(g, cap) for g in mine]
    kps = torch.from_numpy(np.stack([r[0] for r in res]))
    desc = torch.from_numpy(np.stack([r[1] for r in res]))
    cnt = torch.tensor([r[2] for r in res], dtype=torch.int32)
    out = gather_to_rank0(kps, desc, cnt)
    pre = None
    if rank == 0:
        pre = [torch.zeros((world,) + tuple(t.shape), dtype=t.dtype) for t in (kps, desc, cnt)]
    out2 = gather_into(pre, (kps, desc, cnt))
    if rank == 0:
        ok = True
        K, D, N = out
        K2, D2, N2 = out2
        nb = len(mine)
        for g in range(nframes):      # rank-major layout of gather_into: frame g = [g % W][g // W]
            ok &= np.array_equal(K2[g % world][g // world].numpy(), K[g].numpy())
            ok &= np.array_equal(D2[g % world][g // world].numpy(), D[g].numpy()) and int(N2[g % world][g // world]) == int(N[g])
        for g in range(nframes):
            k, d, n = _fake_result(g, cap)
            ok &= int(N[g]) == n and np.array_equal(K[g].numpy(), k) and np.array_equal(D[g].numpy(), d)
        q.put(ok)
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


def test_shard_indices_round_robin():
    from orb_slam2_comment_amd.sharding import shard_indices
    assert shard_indices(10, 0, 4) == [0, 4, 8]
    assert shard_indices(10, 3, 4) == [3, 7]
    allf = sorted(i for r in range(8) for i in shard_indices(512, r, 8))
    assert allf == list(range(512)) and len(shard_indices(512, 5, 8)) == 64


def test_gather_to_rank0_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 8, 40, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert ok


def test_bench_self_launches_two_ranks_gloo_dry_run():
    """`python bench.py --gpus 2` without a launcher environment brings up its own two ranks (before anything could
    touch a GPU); with --dist-backend gloo --dry-run the pair sharding and the gather run on fabricated slots."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--dry-run",
                        "--steps", "2", "--warmup", "0", "--frames-per-gpu", "8"], capture_output=True, text=True, env=env,
                       timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "stdout must carry the one JSON line only (Gloo / RCCL banners belong on stderr): %r" % lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["gather_verified"] is True
    assert out["config"]["pairs_per_rank"] == 4 and out["config"]["parallelism"] == "pair-sharded x2"


def test_bench_default_is_configs3_global_batch_512_at_8_ranks():
    """`python bench.py --gpus 8` runs BASELINE.json configs[3] as written: a global batch of 512 frames, 64 frames = 32
    (last, cur) pairs per GPU (strong scaling over N); --frames-per-gpu fixes the per-GPU batch instead (weak)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--dist-backend", "gloo", "--dry-run",
                        "--steps", "1", "--warmup", "0"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.strip()][-1])
    assert out["n_gpus"] == 8 and out["gather_verified"] is True and out["scaling"] == "strong"
    assert out["config"]["global_batch"] == 512 and out["config"]["frames_per_gpu_per_step"] == 64
    assert out["config"]["pairs_per_rank"] == 32
