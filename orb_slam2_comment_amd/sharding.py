"""Frame-level sharding across GPUs: frames are independent, so frame i
goes to rank i mod world_size, every rank runs the batch kernels on its own shard, and the only
exchange is one gather of fixed-capacity result slots {count, KeyPoint[cap], desc[cap x 32]}
to rank 0 (torch.distributed: backend "nccl" is RCCL over xGMI on MI355X, "gloo" on CPU)."""
import torch
import torch.distributed as dist


def shard_indices(num_frames, rank, world_size):
    """Global frame indices owned by `rank` (round-robin, frame i -> GPU i mod world)."""
    return list(range(rank, num_frames, world_size))


def gather_into(outs, tensors, dst=0, group=None):
    """Allocation-free form for steady-state loops: `outs` (rank `dst` only) are preallocated tensors of shape
    [W, *t.shape] per input tensor; rank w's shard lands in outs[i][w] (rank-major: global frame g = local
    index g // W of rank g % W).  Returns `outs` on `dst`, None elsewhere."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    for i, t in enumerate(tensors):
        t = t.contiguous()
        if rank == dst:
            dist.gather(t, gather_list=[outs[i][w] for w in range(world)], dst=dst, group=group)
        else:
            dist.gather(t, gather_list=None, dst=dst, group=group)
    return outs if rank == dst else None


def gather_to_rank0(kps, desc, counts, dst=0, group=None):
    """kps: int32 [b, cap, 7] (the 28-byte KeyPoint records viewed as 7 dwords),
    desc: uint8 [b, cap, 32], counts: int32 [b]; same shapes on every rank.
    Returns on rank `dst` (kps[W*b,...], desc[...], counts[...]) in GLOBAL frame order
    (frame g = local index g // W on rank g % W), and None elsewhere."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    outs = []
    for t in (kps, desc, counts):
        t = t.contiguous()
        if rank == dst:
            bufs = [torch.empty_like(t) for _ in range(world)]
            dist.gather(t, gather_list=bufs, dst=dst, group=group)
            # interleave: global frame g lives at bufs[g % W][g // W]
            stacked = torch.stack(bufs, dim=1)                    # [b, W, ...]
            outs.append(stacked.reshape((-1,) + tuple(t.shape[1:])))
        else:
            dist.gather(t, gather_list=None, dst=dst, group=group)
    return tuple(outs) if rank == dst else None
