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


def test_bench_gpus_flag_starts_the_ranks_itself():
    """`python bench.py --gpus 2` with no launcher around it must start 2 ranks (here: --dry-run, gloo, no device) and report
    n_gpus = 2; a WORLD_SIZE that disagrees with --gpus is refused."""
    import json, subprocess, sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "2"], env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, timeout=300)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    lines = [ln for ln in out.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1                                              # exactly ONE JSON line on stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["dry_run"] is True and rec["gather_ok"] is True
    env["WORLD_SIZE"] = "1"
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert bad.returncode != 0 and b"refusing" in bad.stderr


def test_cabi_shard_rule_matches_python():
    from icp_amd import batch, binding
    for n_pairs, world in ((44, 8), (7, 2), (3, 4), (0, 2)):
        for r in range(world):
            assert binding.pairs_of_rank(n_pairs, r, world) == len(batch.shard_pairs(n_pairs, r, world))
        assert [binding.pair_owner(p, world) for p in range(n_pairs)] == [p % world for p in range(n_pairs)]


def test_gather_fallback_is_loud_and_taken_by_all_ranks_together():
    """bench.py's pose-gather decision with the C-ABI communicator unavailable (no GPU here; and rank 0 cannot even create the RCCL id:
    ICP_HIP_RCCL_LIB points nowhere): without --allow-gather-fallback EVERY rank stops with the reason (exit != 0, no JSON line);
    with it all ranks take torch.distributed.all_gather together and the line says so."""
    import json, subprocess, sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["ICP_HIP_RCCL_LIB"] = "/nonexistent/librccl.so.1"
    base = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run", "--dry-run-cabi", "--steps", "1"]
    bad = subprocess.run(base, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert bad.returncode != 0
    assert not [ln for ln in bad.stdout.decode().splitlines() if ln.lstrip().startswith("{")]
    err = bad.stderr.decode()
    assert "--allow-gather-fallback" in err and "rank 0" in err and "rank 1" in err          # both ranks stopped, each saying why
    ok = subprocess.run(base + ["--allow-gather-fallback"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert ok.returncode == 0, ok.stderr.decode()[-2000:]
    rec = json.loads([ln for ln in ok.stdout.decode().splitlines() if ln.lstrip().startswith("{")][0])
    assert rec["pose_gather"] == "torch.distributed.all_gather" and rec["gather_ok"] is True and rec["pairs_per_rank"] == [1, 1]
