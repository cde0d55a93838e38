"""world_size-2 gloo test of the N>1 path: pair sharding + the single pose all_gather (icp_amd/batch.py)."""
import os
import socket
import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _fake_pose(p):
    T = np.eye(4, dtype=np.float32); T[:3, 3] = [p, 2 * p, -p]; T[0, 1] = 0.001 * p
    return np.ascontiguousarray(T.T).reshape(16)


def _worker(rank, world, port, n_pairs, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    sys.path.insert(0, os.path.join(root, "icp-variants_amd", "python"))
    from icp_amd import batch
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = []

    def solve(p):
        calls.append(p)
        return _fake_pose(p)
    res = batch.align_batch(n_pairs, solve, device="cpu")
    q.put((rank, calls, res))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def test_shard_layout():
    import sys
    from icp_amd import batch
    assert [len(batch.shard_pairs(44, r, 8)) for r in range(8)] == [6, 6, 6, 6, 5, 5, 5, 5]
    assert sorted(sum((batch.shard_pairs(44, r, 8) for r in range(8)), [])) == list(range(44))
    assert batch.pairs_per_rank(44, 8) == 6


def test_two_rank_gloo_gather():
    world, n_pairs = 2, 7
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_pairs, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    exp = np.stack([_fake_pose(p) for p in range(n_pairs)])
    for rank, calls, res in got:
        assert calls == list(range(rank, n_pairs, world))          # each rank solved only its own pairs
        assert np.array_equal(res, exp)                            # every rank holds all poses, in pair order


def test_single_process_path():
    from icp_amd import batch
    res = batch.align_batch(3, _fake_pose)
    assert np.array_equal(res, np.stack([_fake_pose(p) for p in range(3)]))
