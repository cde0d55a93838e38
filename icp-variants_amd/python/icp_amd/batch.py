"""Multi-GPU batch alignment: independent scan pairs sharded one per rank, ONE pose gather per batch.

The reference aligns ETH pairs in a plain loop with no state carried between indices (main.cpp:411-498,
experiment.cpp:319-396), so pairs shard with no data-path collective: pair p -> rank p mod G.  The only exchange is
a single all_gather of ceil(P/G) x 16 fp32 poses per rank at the end of the batch (RCCL over xGMI when the backend
is "nccl"; "gloo" on CPU for tests).  torch.distributed is plumbing here -- the solve runs in libicp_hip.so.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_pairs(n_pairs, rank, world):
    """Pair indices owned by `rank` (round-robin, 44 pairs on 8 GPUs -> 6,6,6,6,5,5,5,5)."""
    return list(range(rank, n_pairs, world))


def pairs_per_rank(n_pairs, world):
    return (n_pairs + world - 1) // world


def gather_poses(local_poses, n_pairs, device="cpu"):
    """local_poses: (k,16) float32 array of this rank's poses in shard order.  Returns (n_pairs,16) on every rank."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    cap = pairs_per_rank(n_pairs, world)
    buf = torch.zeros(cap, 16, dtype=torch.float32, device=device)
    lp = np.asarray(local_poses, np.float32).reshape(-1, 16)
    if len(lp):
        buf[:len(lp)] = torch.from_numpy(lp).to(device)
    if world == 1:
        return buf[:n_pairs].cpu().numpy()
    out = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)                       # the single collective of the batch
    allp = torch.stack(out).cpu().numpy()           # (world, cap, 16)
    res = np.zeros((n_pairs, 16), np.float32)
    for r in range(world):
        own = shard_pairs(n_pairs, r, world)
        res[own] = allp[r, :len(own)]
    return res


def align_batch(n_pairs, solve_pair, device="cpu"):
    """Runs solve_pair(p) -> 16 floats (column-major pose) for this rank's pairs, then gathers all poses."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    mine = shard_pairs(n_pairs, rank, world)
    local = np.stack([np.asarray(solve_pair(p), np.float32).reshape(16) for p in mine]) if mine else np.zeros((0, 16), np.float32)
    return gather_poses(local, n_pairs, device)
