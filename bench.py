#!/usr/bin/env python3
"""bench.py -- ICP iterations/s and correspondences/s on 370k-point ETH-Apartment-class pairs.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`.  For N > 1 the driver starts it through
torch.distributed.run (one rank per GPU, WORLD_SIZE set); when WORLD_SIZE is NOT set and N > 1, bench.py starts the N ranks
itself (a child `python -m torch.distributed.run ...` launched BEFORE anything touches the GPU) and relays rank 0's JSON line.
A run whose WORLD_SIZE disagrees with --gpus exits non-zero: a single rank can never be reported as N GPUs.

A "step" is one LinearICPOptimizer::estimatePose on one scan pair: BASELINE.json configs[1] -- synthetic
ETH-Apartment-like pair (344 x 1077 = 370 488 points each), exact k-NN matching,
point-to-plane linear solve, maxDist^2 = 10 (main.cpp:361), 50 iterations (main.cpp:366), rejection ON, constant
weights, no multi-resolution.  Clouds are uploaded (icp_set_target / icp_set_source) BEFORE the timed region;
the timed region is icp_run only (matching + weighting + rejection + system build + solve, 50 x).
N GPUs: every rank aligns its own pair (pair index = rank, weak scaling, independent pairs as in main.cpp:411),
then ONE gather of the 16-float poses per step (ncclAllGather issued from the C ABI, icp_gather_poses; RCCL over xGMI).
value = ICP iterations/s summed over all ranks; correspondences/s = value x 370 488.
`--eth-dir DIR` runs the same settings on real ETH scans (PCD + pose CSV in the reference's layout) instead.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "icp-variants_amd", "python")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np   # noqa: E402

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_VALU_PEAK_TFLOPS = 157.3    # MI355X_MICROARCH.md: peak FP32 vector


def cpu_baseline(pair, n_iter=100, threads=None):
    """Oracle (CPU restatement) timed on the host cores: exact kd-tree matcher (stand-in for the FLANN kd-tree the
    reference instantiates, NearestNeighbor.h:122-207) + reference-shaped fp32 dense solve.  Bounded sample: 100 iterations
    (two steps' worth of the workload, about 10 s of wall time on the GPU box's host cores)."""
    from oracle import oracle as orc
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = threads or max(1, min(16, ncpu))
    orc.set_num_threads(threads)
    prm = orc.make_params(metric=1, n_iterations=n_iter, max_distance=10.0, solver_mode=0, knn_kdtree=1)
    t0 = time.perf_counter()
    kd = orc.KdTree(pair["tgt_pts"])                       # buildIndex, once per pair (ICPOptimizer.h:532-535)
    t_build = time.perf_counter() - t0
    prm.kdtree = kd.h
    t0 = time.perf_counter()
    pose, recs = orc.estimate_pose(prm, pair["src_pts"], pair["src_nrm"], None, pair["tgt_pts"], pair["tgt_nrm"], None, np.eye(4, dtype=np.float32))
    dt = time.perf_counter() - t0
    its = len(recs) / dt
    return {"value": its, "unit": "ICP iterations/s", "cores": threads, "kind": "port",
            "sample": "%d full ICP iterations of the same %d-point pair (CPU oracle: exact kd-tree k-NN on %d OpenMP threads, "
                      "weighting/rejection/compaction + fp32 4n x 6 QR/Jacobi-SVD solve on 1 thread); kd-tree build %.3f s excluded, "
                      "match %.3f s/iter, rest %.3f s/iter" % (len(recs), len(pair["src_pts"]), threads, t_build,
                                                                 float(np.mean([r["seconds_match"] for r in recs])),
                                                                 float(np.mean([r["seconds_rest"] for r in recs])))}


def cpu_baseline_detail(pair, threads):
    """SURVEY.md 8d variants, each on a bounded sample: (i) exact brute-force matcher on all threads and on 1 thread
    (2048-query sample, extrapolated to the full query count), (ii) exact kd-tree matcher on 1 thread and on all threads (full
    cloud), (iii) the rest of an iteration (weighting, rejection, compaction, fp32 4n x 6 dense solve) on 1 thread."""
    from oracle import oracle as orc
    out = {}
    q = pair["src_pts"]; t = pair["tgt_pts"]
    sub = q[:: max(1, len(q) // 2048)][:2048]
    for name, th in (("bruteforce_all_threads", threads), ("bruteforce_1_thread", 1)):
        orc.set_num_threads(th)
        t0 = time.perf_counter(); orc.knn3(sub, t, 10.0); dt = time.perf_counter() - t0
        out[name + "_s_per_iter"] = dt * len(q) / len(sub)
    kd = orc.KdTree(t)
    for name, th in (("kdtree_1_thread", 1), ("kdtree_all_threads", threads)):
        orc.set_num_threads(th)
        t0 = time.perf_counter(); kd.query(q, 10.0); out[name + "_s_per_iter"] = time.perf_counter() - t0
    orc.set_num_threads(threads)
    prm = orc.make_params(metric=1, n_iterations=1, max_distance=10.0, solver_mode=0, knn_kdtree=1); prm.kdtree = kd.h
    _, _, _, tm, tr = orc.iterate(prm, pair["src_pts"], pair["src_nrm"], None, pair["tgt_pts"], pair["tgt_nrm"], None, np.eye(4, dtype=np.float32))
    out["weight_reject_compact_dense_solve_1_thread_s_per_iter"] = tr
    out["threads"] = threads
    out["note"] = "SURVEY.md 8d (i) brute force extrapolated from a 2048-query sample, (ii) exact kd-tree over all queries, (iii) rest of one iteration"
    return out


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def spawn_ranks(n):
    """--gpus N without a launcher: start the N ranks as a child torch.distributed.run job and relay its output.
    Called before torch is imported or any GPU call is made (never exec from a process that initialised the GPU)."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout.splitlines():                              # rank 0's JSON line on stdout, launcher / backend chatter on stderr
        (sys.stdout if line.lstrip().startswith("{") else sys.stderr).write(line + "\n")
    sys.stdout.flush()
    raise SystemExit(proc.returncode)


class PoseGather:
    """The single collective of a batch.  Default: ncclAllGather issued from the C ABI (icp_comm_* / icp_gather_poses); the
    128-byte unique id travels through torch.distributed, which is plumbing here.  If the C-ABI communicator cannot be created on
    EVERY rank the run stops (all ranks, with the reason) -- a line measured with another collective is not the design BASELINE.json
    names -- unless --allow-gather-fallback asks for torch.distributed.all_gather instead; the JSON line says which it was
    (`pose_gather`).  device "cpu" (dry run, gloo): the same decision logic without a GPU."""

    def __init__(self, world, rank, local_rank, mode, device, allow_fallback=False):
        import torch
        import torch.distributed as dist
        self.world, self.rank, self.torch, self.dist, self.device = world, rank, torch, dist, device
        self.comm = None; self.kind = "none (1 rank)"; self.error = None
        if world == 1:
            return
        self.kind = "torch.distributed.all_gather"
        if mode == "cabi":
            from icp_amd import binding
            ok = 1; idb = bytes(binding.COMM_ID_BYTES)
            if rank == 0:
                try:
                    idb = binding.Comm.unique_id()
                except Exception as e:                                 # noqa: BLE001 -- recorded in the JSON line
                    self.error = str(e)
            idt = torch.frombuffer(bytearray(idb), dtype=torch.uint8).clone().to(device)
            dist.broadcast(idt, src=0)                                 # every rank takes part, whatever happened on rank 0
            idb = bytes(idt.cpu().numpy().tobytes())
            if not any(idb):
                ok = 0                                                 # rank 0 could not create an id: nobody calls ncclCommInitRank
                self.error = self.error or "rank 0 could not create the RCCL unique id"
            else:
                try:
                    self.comm = binding.Comm(local_rank, world, rank, idb)
                except Exception as e:                                 # noqa: BLE001
                    self.error = str(e); ok = 0
            flag = torch.tensor([ok], dtype=torch.int32, device=device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)                # all ranks use the same path
            if int(flag.item()) == 1:
                self.kind = "icp_gather_poses (ncclAllGather from the C ABI)"
            else:
                if self.comm is not None:
                    self.comm.close()
                self.comm = None
                if not allow_fallback:                                 # every rank gets here together (the MIN above)
                    raise SystemExit("bench.py: rank %d: the C-ABI pose gather (icp_comm_create / ncclAllGather) is not available on every rank%s -- "
                                     "pass --allow-gather-fallback to measure with torch.distributed.all_gather instead (the line will say so)"
                                     % (rank, ": " + self.error if self.error else ""))

    def gather(self, local_poses, n_pairs):
        from icp_amd import batch
        if self.world == 1:
            return np.asarray(local_poses, np.float32).reshape(-1, 16)[:n_pairs]
        if self.comm is not None:
            return self.comm.gather_poses(local_poses, n_pairs)
        return batch.gather_poses(local_poses, n_pairs, self.device)


def dry_run(args, world, rank):
    """CPU rehearsal of the N-rank path (gloo, no device): sharding + the pose gather with synthetic poses.  For tests."""
    import torch.distributed as dist
    from icp_amd import batch
    if world > 1:
        dist.init_process_group(backend="gloo")
    n_pairs = args.pairs if args.pairs > 0 else world
    mine = batch.shard_pairs(n_pairs, rank, world)
    gather_kind = None
    if args.dry_run_cabi:                                              # the decision logic of the real run, on CPU tensors
        g = PoseGather(world, rank, int(os.environ.get("LOCAL_RANK", "0")), "cabi", "cpu", args.allow_gather_fallback)
        gather_kind = g.kind

    def fake(p):
        T = np.eye(4, dtype=np.float32); T[:3, 3] = [p, 2 * p, -p]
        return np.ascontiguousarray(T.T).reshape(16)
    t0 = time.perf_counter()
    for _ in range(max(1, args.steps)):
        poses = batch.align_batch(n_pairs, fake, device="cpu")
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    ok = bool(np.array_equal(poses, np.stack([fake(p) for p in range(n_pairs)])))
    if rank == 0:
        print(json.dumps({"metric": "dry run (no device): pair sharding + pose gather only", "value": 0.0, "unit": "ICP iterations/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / max(1, args.steps) * 1e3, "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "f32", "data": "synthetic", "dry_run": True, "config": {"workload": "dry run", "pairs": n_pairs},
                          "gather_ok": ok, "pairs_of_rank0": mine, "pose_gather": gather_kind,
                          "pairs_per_rank": [len(batch.shard_pairs(n_pairs, r, world)) for r in range(world)]}), flush=True)
    if world > 1:
        dist.destroy_process_group()
    if not ok:
        raise SystemExit(3)


def make_optimizer(binding, local_rank, args, incremental=True):
    opt = binding.LinearICPOptimizer(local_rank)
    opt.setMatchingMethod(0); opt.setMatchingMaxDistance(10.0)          # main.cpp:360-361
    opt.setMetric(1); opt.setNbOfIterations(args.iterations)             # main.cpp:364-366
    opt.setWeightingMethod(0); opt.setRejectionMethod(1)
    opt.setKnnBackend(1 if args.knn == "lbvh" else 0)
    opt.ctx.params.knn_incremental = 1 if incremental else 0
    opt.ctx.push_params()
    return opt


def batch_mode(args, world, rank, local_rank):
    """configs[3]: P consecutive pairs, pair p -> rank p mod N, ONE pose gather per step.  Unlike the default mode the
    timed region contains everything a real batch pays per pair: host->device upload (AoS->SoA on the device), BVH
    index build, query ordering and the 50 iterations.  Reported next to, never instead of, the resident-input number.
    The pairs of a rank run through icp_batch_run: one host thread per context inside the library."""
    import torch
    import torch.distributed as dist
    from icp_amd import binding, synth, batch
    mine = batch.shard_pairs(args.pairs, rank, world)
    if args.shared_scans and world == 1:
        # the ETH loop's shape (main.cpp:411-498): pair p aligns scan p + 1 to scan p, the perturbation is the INITIAL POSE (main.cpp:420-429);
        # scan p + 1 is handed over as the SAME arrays to pair p (source) and pair p + 1 (target): icp_batch_run uploads it once per run of pairs
        raw = [synth.laser_scan(synth.scan_pose(k % 45, 0xE7A0), 0xE7A0 + k % 45, args.n_tilt, args.n_beam, 0.01) for k in range(args.pairs + 1)]
        scans = []; init = []
        for p in mine:
            Tp = synth.perturbation(0xE7A0 + 100003 * (p + 1))
            scans.append(dict(src_pts=raw[p + 1][0], src_nrm=raw[p + 1][1], tgt_pts=raw[p][0], tgt_nrm=raw[p][1], gt=np.eye(4)))
            init.append(Tp)
    else:
        scans = [synth.eth_like_pair(p % 44, n_tilt=args.n_tilt, n_beam=args.n_beam) for p in mine]
        init = None
    n_ctx = max(1, args.contexts)
    opts = [make_optimizer(binding, local_rank, args) for _ in range(n_ctx)]
    for o in opts:
        o.ctx.set_stage_timing(0)                                  # no per-stage breakdown is reported in this mode
    ctxs = [o.ctx for o in opts]
    gather = PoseGather(world, rank, local_rank, args.gather, "cuda", args.allow_gather_fallback)

    def step():
        local, status, rc = binding.batch_run(ctxs, scans, init)
        if rc != 0:
            raise SystemExit("icp_batch_run failed: %s" % status.tolist())
        return gather.gather(local, args.pairs)

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        poses = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    n_src = len(scans[0]["src_pts"]) if scans else 0
    errs = []
    for p, d in zip(mine, scans):
        P = binding.pose_from_c(poses[p]).astype(np.float64)
        errs.append(float(np.linalg.norm(P[:3, 3] - d["gt"][:3, 3])))
    value = args.pairs * args.iterations * args.steps / elapsed
    out = {"metric": "ICP iterations/s (batch of scan pairs, uploads + index builds included)", "value": value, "unit": "ICP iterations/s",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
           "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "configs[3]: %d consecutive synthetic ETH-like pairs, pair p -> rank p mod N, exact %s k-NN + point-to-plane, "
                                  "%d iterations per pair, host->device uploads and index builds inside the timed region, one pose gather per step"
                                  % (args.pairs, args.knn, args.iterations), "pairs": args.pairs, "contexts_per_gpu": n_ctx, "shared_scans": bool(args.shared_scans and world == 1)},
           "pairs_per_s": args.pairs * args.steps / elapsed, "correspondences_per_s": value * n_src, "pose_gather": gather.kind,
           "pairs_per_rank": [len(batch.shard_pairs(args.pairs, r, world)) for r in range(world)], "max_over_ranks_step_ms": elapsed / args.steps * 1e3,
           "max_trans_err_vs_gt_m_rank0": max(errs) if errs else None}
    if gather.error:
        out["pose_gather_error"] = gather.error
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def load_workload(args, binding, synth, rank, ctx):
    """The pair this rank aligns: synthetic configs[1] scans, or real ETH files in the reference's layout (--eth-dir)."""
    if args.eth_dir:
        from icp_amd import eth
        rows = eth.load_rows(args.eth_dir, args.eth_csv)
        row = rows[(args.eth_index + rank) % len(rows)]
        src, tgt = eth.load_scans(args.eth_dir, args.eth_csv, row)
        pair = eth.prepare_pair(ctx, src, tgt, row["pose"])           # k = 5 normals on the device, pose_scaling 0.1 (main.cpp:420-429)
        return pair, "eth", "ETH pair %s (%s -> %s) from %s/%s" % (row["id"], row["source"], row["target"], args.eth_dir, args.eth_csv)
    pair = synth.eth_like_pair(rank % 44, n_tilt=args.n_tilt, n_beam=args.n_beam)
    return pair, "synthetic", "configs[1]: synthetic ETH-Apartment-like pair (rank, rank+1)"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--iterations", type=int, default=50, help="ICP iterations per step (main.cpp:366)")
    ap.add_argument("--knn", choices=["brute", "lbvh"], default=os.environ.get("ICP_BENCH_KNN", "lbvh"))
    ap.add_argument("--n-tilt", type=int, default=344)
    ap.add_argument("--n-beam", type=int, default=1077)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--per-iteration", action="store_true", help="add the matcher's average device ms per iteration index to per_regime_ms")
    ap.add_argument("--stage-timing", type=int, default=17, help="bracket every Nth ICP iteration of the timed region with HIP events (offset rotates "
                    "from step to step: after N steps every iteration index has been timed once); 1 = every iteration, as the reference's "
                    "TimeMeasure does (costs ~10 %% of an iteration); an event bracket costs ~3 us of stream time")
    ap.add_argument("--no-incremental", action="store_true", help="always walk the BVH (disable the exact verify-and-skip of converged queries)")
    ap.add_argument("--no-cpu-baseline-detail", action="store_true", help="skip the SURVEY 8d CPU variants (i)-(iii) (a few seconds)")
    ap.add_argument("--resident-pairs", type=int, default=1, help="default mode: this many independent resident pairs per GPU, aligned concurrently "
                    "on their own HIP streams in every step (the matcher is latency-bound: concurrent pairs fill the idle issue slots); "
                    "1 = the configs[1] workload as BASELINE.json states it")
    ap.add_argument("--contexts", type=int, default=4, help="batch mode: contexts (= HIP streams, host threads) per rank")
    ap.add_argument("--pairs", type=int, default=0, help="batch mode (configs[3]): align this many consecutive scan pairs per step, "
                    "sharded pair p -> rank p mod N, uploads and index builds INSIDE the timed region, one pose gather per step")
    ap.add_argument("--shared-scans", action="store_true", help="batch mode on one rank: the batch is a scan SEQUENCE (pair p = scans p, p + 1 as the same arrays; the perturbation is "
                    "the initial pose, as in the reference's ETH loop): icp_batch_run uploads a shared scan once per run of pairs")
    ap.add_argument("--gather", choices=["cabi", "torch"], default="cabi", help="pose gather: ncclAllGather from the C ABI (default) or torch.distributed")
    ap.add_argument("--allow-gather-fallback", action="store_true", help="if the C-ABI communicator cannot be created on every rank, gather with "
                    "torch.distributed.all_gather instead of stopping (the line then says so in `pose_gather`)")
    ap.add_argument("--dry-run-cabi", action="store_true", help="with --dry-run: also go through the C-ABI communicator set-up (no GPU: it cannot succeed; tests the "
                    "all-ranks-take-the-same-path logic of the fallback)")
    ap.add_argument("--no-extras", action="store_true", help="skip the legs outside the timed region (per-iteration records, the always-walk comparison run): "
                    "for profiling, so that a kernel trace holds the launches of the warm-up and timed steps only")
    ap.add_argument("--dry-run", action="store_true", help="no device: rehearse the N-rank sharding + pose gather on CPU (gloo)")
    ap.add_argument("--eth-dir", default=None, help="directory with <name>_global.csv and <name>/<scan>.pcd (the reference's Data/eth layout): "
                    "run the configs[1] settings on real ETH scans instead of the synthetic pair")
    ap.add_argument("--eth-csv", default="apartment_global.csv")
    ap.add_argument("--eth-index", type=int, default=0, help="first CSV row (rank r uses row index + r)")
    args = ap.parse_args()

    # ---- rank bookkeeping BEFORE anything touches the GPU ------------------------------------------------------------
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        spawn_ranks(args.gpus)                                          # does not return
    world = int(env_world or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d -- refusing to report a %d-rank run as %d GPUs" % (args.gpus, world, world, args.gpus))
    if args.dry_run:
        return dry_run(args, world, rank)

    import torch
    import torch.distributed as dist
    from icp_amd import binding, synth

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the ICP hot path has no CPU fallback")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d needs GPU %d but only %d visible" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    if args.pairs > 0:
        return batch_mode(args, world, rank, local_rank)

    # ---- workload: pair (rank, rank+1) of the synthetic 45-scan sequence, or real ETH files -------------------------
    opt = make_optimizer(binding, local_rank, args, incremental=not args.no_incremental)
    ctx = opt.ctx
    pair, data_kind, workload = load_workload(args, binding, synth, rank, ctx)
    n_src, n_tgt = len(pair["src_pts"]), len(pair["tgt_pts"])
    ctx.set_stage_timing(args.stage_timing)
    ctx.set_target(pair["tgt_pts"], pair["tgt_nrm"], None)              # resident in HBM before the timed region
    ctx.set_source(pair["src_pts"], pair["src_nrm"], None)
    eye = binding.pose_to_c(np.eye(4, dtype=np.float32))
    gather = PoseGather(world, rank, local_rank, args.gather, "cuda", args.allow_gather_fallback)

    # optional extra resident pairs on their own contexts / streams / host threads (ctypes releases the GIL)
    R = max(1, args.resident_pairs)
    extra = []
    for r in range(1, R):
        pr = synth.eth_like_pair((rank * R + r) % 44, n_tilt=args.n_tilt, n_beam=args.n_beam)
        o = make_optimizer(binding, local_rank, args, incremental=not args.no_incremental)
        o.ctx.set_stage_timing(0)
        o.ctx.set_target(pr["tgt_pts"], pr["tgt_nrm"], None); o.ctx.set_source(pr["src_pts"], pr["src_nrm"], None)
        extra.append(o.ctx)
    from concurrent.futures import ThreadPoolExecutor
    pools = [ThreadPoolExecutor(1) for _ in extra]

    def run_extra(cx):
        q = eye.copy(); cx.run_raw(q)
        return q

    def step():
        futs = [pl.submit(run_extra, cx) for pl, cx in zip(pools, extra)]
        pose = eye.copy()
        ctx.run_raw(pose)                                               # 50 ICP iterations, no host round trip inside
        for f in futs:
            f.result()
        if world > 1:
            gather.gather(pose.reshape(1, 16), world)                   # the single pose gather of the batch (one pair per rank)
        return pose

    for _ in range(args.warmup):
        step()
    acc = dict(match_ms=0.0, weight_reject_build_ms=0.0, solve_ms=0.0, total_ms=0.0, iterations=0, sampled_iterations=0)
    it_sum = np.zeros((2, args.iterations)); it_cnt = np.zeros(args.iterations)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pose = step()
        tm = ctx.timing()
        for k in acc:
            acc[k] += tm[k]
        if args.stage_timing > 0:
            a, _, d = ctx.iteration_times(args.iterations)              # host bookkeeping of the events already recorded
            m = a >= 0
            it_sum[0, :len(a)][m] += a[m]; it_sum[1, :len(a)][m] += d[m]; it_cnt[:len(a)][m] += 1
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    recs = []
    if not args.no_extras:
        _, recs, _ = ctx.run(binding.pose_from_c(eye), check=False)     # after the timed region: per-iteration records for the line

    iters_total = world * R * args.steps * args.iterations
    value = iters_total / elapsed
    # ---- roofline of the dominant kernel, HIP events on the context's own stream ----
    launches = max(acc["iterations"], 1)
    knn_ms = acc["match_ms"] / launches
    fused = args.knn == "lbvh"                                          # k_knn_bvh_post: search + weight / reject / accumulate in ONE launch
    match_bytes = 12 * n_src + 12 * n_tgt + 8 * n_src                   # read src xyz + tgt xyz, write Match
    post_bytes = 56 * n_src                                             # src xyz + normal, Match, gathered tgt xyz + normal
    alg_bytes = match_bytes + post_bytes if fused else match_bytes      # what ONE launch of the named kernel does
    have_stage = knn_ms > 0.0                                           # --stage-timing 0: no per-stage events, no kernel duration
    if not have_stage:
        knn_ms = float("inf")
    achieved = alg_bytes / (knn_ms * 1e-3) / 1e9
    pairs = float(n_src) * float(n_tgt)
    flops = pairs * 8.0                                                 # 3 sub + 3 mul + 2 add per pair (no FMA: bit-exact contract)
    traffic = None; traffic_source = None
    tfile = os.path.join(ROOT, "profiles", "traffic_latest.json")       # HBM bytes/launch from a separate rocprofv3 --pmc pass
    if os.path.exists(tfile):
        try:
            tj = json.load(open(tfile))
            if tj.get("knn") == args.knn and tj.get("n_src") == n_src:
                traffic = tj.get("hbm_bytes_per_launch")
                traffic_source = "static: profiles/traffic_latest.json (%s), NOT measured in this run" % tj.get("source", "separate rocprofv3 --pmc passes")
        except Exception:
            traffic = None
    gt = pair["gt"]
    P = binding.pose_from_c(pose).astype(np.float64)
    dR = P[:3, :3] @ gt[:3, :3].T
    rot_err = float(np.arctan2(0.5 * np.linalg.norm([dR[2, 1] - dR[1, 2], dR[0, 2] - dR[2, 0], dR[1, 0] - dR[0, 1]]), (np.trace(dR) - 1) / 2))
    trans_err = float(np.linalg.norm(P[:3, 3] - gt[:3, 3]))

    def regime(lo, hi, row):
        m = it_cnt[lo:hi] > 0
        return float(np.mean(it_sum[row, lo:hi][m] / it_cnt[lo:hi][m])) if m.any() else None
    per_regime = {"note": "average device ms per launch of the matcher / of the reduce+solve by iteration index (HIP events)",
                  "match_iteration_0": regime(0, 1, 0), "match_iterations_1_9": regime(1, 10, 0), "match_iterations_10_16": regime(10, 17, 0),
                  "match_iterations_17_plus": regime(17, args.iterations, 0), "solve_all": regime(0, args.iterations, 1)}

    if args.per_iteration:
        per_regime["match_by_iteration"] = [round(float(it_sum[0, i] / it_cnt[i]), 5) if it_cnt[i] > 0 else None for i in range(args.iterations)]
    out = {
        "metric": "ICP iterations/s (k-NN + point-to-plane linear, 370k-point ETH-Apartment-like pair)",
        "value": value, "unit": "ICP iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": data_kind,
        "config": {"workload": "%s, %d x %d pts, exact %s k-NN, point-to-plane linear, maxDist^2=10, %d iterations/step, rejection on"
                               % (workload, n_src, n_tgt, args.knn, args.iterations),
                   "pairs_per_step": world * R, "resident_pairs_per_gpu": R, "iterations_per_step": args.iterations, "knn_backend": args.knn,
                   "knn_incremental": (not args.no_incremental) and args.knn == "lbvh"},
        "correspondences_per_s": value * n_src,
        "ms_per_iteration": elapsed / (args.steps * args.iterations) * 1e3,
        "stage_ms_per_iteration": {"match": acc["match_ms"] / launches, "weight_reject_build": acc["weight_reject_build_ms"] / launches,
                                   "solve": acc["solve_ms"] / launches},
        "per_regime_ms": per_regime,
        "n_valid_first": recs[0]["n_valid"] if recs else None, "n_valid_last": recs[-1]["n_valid"] if recs else None,
        "roofline": {"kernel": ("k_knn_bvh_post_ring<3, false> (one launch per iteration: exact BVH 1-NN + weight/reject/accumulate epilogue; its first 34 blocks fold the previous "
                                "iteration's partials and solve -- the matcher blocks wait for that pose after issuing their loads)" if os.environ.get("ICP_HIP_MERGE", "1") != "0" else
                                "k_knn_bvh_post<3, false> (exact BVH 1-NN + weight/reject/accumulate epilogue, one launch per iteration; k_reduce_solve separate)") if fused else "k_knn_brute<3>",
                     "bound": "hbm", "achieved": achieved if have_stage else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS if have_stage else None, "traffic": traffic, "traffic_source": traffic_source,
                     "algorithmic_bytes_per_launch": alg_bytes, "match_only_bytes_per_launch": match_bytes,
                     "frac_match_only": (match_bytes / (knn_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if have_stage else None,
                     "avg_launch_ms": knn_ms if have_stage else None,
                     "timed_launches": acc["sampled_iterations"], "launches": acc["iterations"],
                     "note": "algorithmic bytes = SURVEY.md 8d per-iteration figure for what ONE launch of the named kernel does (fused: match 32 B/query "
                             "+ 12 B/target and weight/reject/accumulate 56 B/query); duration = HIP events on the context's stream around the launch of "
                             "every %s iteration of the timed region (offset rotating per step); in the merged form that launch also holds the reduce + solve of "
                             "the iteration before (no separate solve stage: stage_ms_per_iteration.solve is the closing launch of a run only)"
                             % ("" if args.stage_timing == 1 else "%d-th" % args.stage_timing)},
        "pose_error_vs_gt": {"rot_rad": rot_err, "trans_m": trans_err},
        "pose_gather": gather.kind,
        "pairs_per_rank": [R] * world, "max_over_ranks_step_ms": elapsed / args.steps * 1e3,
    }
    import ctypes as _C
    _a, _b = _C.c_int32(0), _C.c_int32(0)
    if ctx.lib.icp_debug_counters(ctx.h, _C.byref(_a), _C.byref(_b)) == 0:
        out["loop_form"] = {"runs_in_one_launch_per_level_or_merged_form": _a.value, "of_those_repeated_with_separate_launches": _b.value,
                            "note": "icp_run keeps the whole loop of a resolution level in ONE launch (k_icp_loop) when its grid fits the device, else rides the reducer "
                                    "in front of the next matcher launch; a run whose 6x6 system needs the eigen fallback is repeated with separate k_reduce_solve launches"}
    if gather.error:
        out["pose_gather_error"] = gather.error
    if args.knn == "brute":
        out["valu_roofline"] = {"pair_evals_per_s": pairs / (knn_ms * 1e-3), "achieved_tflops": flops / (knn_ms * 1e-3) / 1e12,
                                "peak_tflops": FP32_VALU_PEAK_TFLOPS,
                                "note": "brute force is FP32-VALU-bound, not HBM-bound (SURVEY.md 8d); 8 non-fused flop per pair"}
    if rank == 0 and world == 1 and fused and not args.no_incremental and not args.no_extras:
        # the same workload with the verify-and-skip test off (every query walks the tree in every iteration): outside the timed region
        o2 = make_optimizer(binding, local_rank, args, incremental=False)
        o2.ctx.set_stage_timing(0)
        o2.ctx.set_target(pair["tgt_pts"], pair["tgt_nrm"], None); o2.ctx.set_source(pair["src_pts"], pair["src_nrm"], None)
        q = eye.copy(); o2.ctx.run_raw(q)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        for _ in range(3):
            q = eye.copy(); o2.ctx.run_raw(q)
        torch.cuda.synchronize()
        out["no_incremental_value"] = 3 * args.iterations / (time.perf_counter() - t1)
        o2.ctx.close()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(pair)
        if not args.no_cpu_baseline_detail:
            out["cpu_baseline"]["detail"] = cpu_baseline_detail(pair, out["cpu_baseline"]["cores"])
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
