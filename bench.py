#!/usr/bin/env python3
"""bench.py -- ICP iterations/s and correspondences/s on 370k-point ETH-Apartment-class pairs.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 launched through
torch.distributed.run, one rank per GPU (RCCL).  Prints ONE JSON line on rank 0.

A "step" is one LinearICPOptimizer::estimatePose on one scan pair: BASELINE.json configs[1] -- synthetic
ETH-Apartment-like pair (344 x 1077 = 370 488 points each), exact k-NN matching,
point-to-plane linear solve, maxDist^2 = 10 (main.cpp:361), 50 iterations (main.cpp:366), rejection ON, constant
weights, no multi-resolution.  Clouds are uploaded (icp_set_target / icp_set_source) BEFORE the timed region;
the timed region is icp_run only (matching + weighting + rejection + system build + solve, 50 x).
N GPUs: every rank aligns its own pair (pair index = rank, weak scaling, independent pairs as in main.cpp:411),
then ONE all_gather of the 16-float poses per step (RCCL over xGMI).
value = ICP iterations/s summed over all ranks; correspondences/s = value x 370 488.
"""
import argparse
import json
import os
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
            "sample": "%d full ICP iterations of the same 370488-point pair (CPU oracle: exact kd-tree k-NN on %d OpenMP threads, "
                      "weighting/rejection/compaction + fp32 4n x 6 QR/Jacobi-SVD solve on 1 thread); kd-tree build %.3f s excluded, "
                      "match %.3f s/iter, rest %.3f s/iter" % (len(recs), threads, t_build,
                                                                 float(np.mean([r["seconds_match"] for r in recs])),
                                                                 float(np.mean([r["seconds_rest"] for r in recs])))}


def cpu_baseline_detail(pair, threads):
    """SURVEY.md 8d variants, each on a bounded sample: (i) exact brute-force matcher on all threads and on 1 thread
    (4096-query sample, extrapolated to 370 488 queries), (ii) exact kd-tree matcher on 1 thread and on all threads (full
    cloud), (iii) the rest of an iteration (weighting, rejection, compaction, fp32 4n x 6 dense solve) on 1 thread."""
    from oracle import oracle as orc
    out = {}
    q = pair["src_pts"]; t = pair["tgt_pts"]
    sub = q[:: max(1, len(q) // 4096)][:4096]
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
    return out


def batch_mode(args, world, rank, local_rank):
    """configs[3]: P consecutive pairs, pair p -> rank p mod N, ONE pose gather per step.  Unlike the default mode the
    timed region contains everything a real batch pays per pair: host->device upload (AoS->SoA on the device), BVH
    index build, query ordering and the 50 iterations.  Reported next to, never instead of, the resident-input number."""
    import torch
    import torch.distributed as dist
    from icp_amd import binding, synth, batch
    mine = batch.shard_pairs(args.pairs, rank, world)
    scans = {}
    for p in mine:
        scans[p] = synth.eth_like_pair(p % 44, n_tilt=args.n_tilt, n_beam=args.n_beam)
    # Several contexts (= HIP streams) per rank, each driven by its own host thread: while one pair iterates, other pairs'
    # host->device uploads, AoS->SoA conversions, index builds and iterations run on the other streams (ctypes releases
    # the GIL).  The matcher is latency-bound, so concurrent pairs fill issue slots a single pair leaves idle:
    # 1 / 2 / 3 / 4 contexts measured 165 / 250 / 288 / 340 pairs/s on one MI355X.
    from concurrent.futures import ThreadPoolExecutor
    n_ctx = max(1, args.contexts)
    ctxs = []
    for _ in range(n_ctx):
        opt = binding.LinearICPOptimizer(local_rank)
        opt.setMatchingMethod(0); opt.setMatchingMaxDistance(10.0); opt.setMetric(1); opt.setNbOfIterations(args.iterations)
        opt.setKnnBackend(1 if args.knn == "lbvh" else 0)
        opt.ctx.push_params()
        opt.ctx.set_stage_timing(0)                                  # no per-stage breakdown is reported in this mode
        ctxs.append(opt.ctx)
    eye = binding.pose_to_c(np.eye(4, dtype=np.float32))
    pools = [ThreadPoolExecutor(1) for _ in range(n_ctx)]          # one worker per context: a context is single-threaded

    def solve_on(ctx, p):
        d = scans[p]
        ctx.set_target(d["tgt_pts"], d["tgt_nrm"], None)
        ctx.set_source(d["src_pts"], d["src_nrm"], None)
        pose = eye.copy(); ctx.run_raw(pose)
        return pose

    def step():
        futs = {p: pools[i % n_ctx].submit(solve_on, ctxs[i % n_ctx], p) for i, p in enumerate(mine)}
        return batch.align_batch(args.pairs, lambda p: futs[p].result(), device="cuda")

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
    n_src = len(next(iter(scans.values()))["src_pts"]) if scans else 0
    errs = []
    for p in mine:
        P = binding.pose_from_c(poses[p]).astype(np.float64); gt = scans[p]["gt"]
        errs.append(float(np.linalg.norm(P[:3, 3] - gt[:3, 3])))
    value = args.pairs * args.iterations * args.steps / elapsed
    out = {"metric": "ICP iterations/s (batch of scan pairs, uploads + index builds included)", "value": value, "unit": "ICP iterations/s",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
           "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "configs[3]: %d consecutive synthetic ETH-like pairs, pair p -> rank p mod N, exact %s k-NN + point-to-plane, "
                                  "%d iterations per pair, host->device uploads and index builds inside the timed region, one pose all_gather per step"
                                  % (args.pairs, args.knn, args.iterations), "pairs": args.pairs},
           "pairs_per_s": args.pairs * args.steps / elapsed, "correspondences_per_s": value * n_src,
           "max_trans_err_vs_gt_m_rank0": max(errs) if errs else None}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--iterations", type=int, default=50, help="ICP iterations per step (main.cpp:366)")
    ap.add_argument("--knn", choices=["brute", "lbvh"], default=os.environ.get("ICP_BENCH_KNN", "lbvh"))
    ap.add_argument("--n-tilt", type=int, default=344)
    ap.add_argument("--n-beam", type=int, default=1077)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--stage-timing", type=int, default=5, help="bracket every Nth ICP iteration of the timed region with HIP events (offset rotates "
                    "from step to step); 1 = every iteration, as the reference's TimeMeasure does (costs ~10 %% of an iteration)")
    ap.add_argument("--no-incremental", action="store_true", help="always walk the BVH (disable the exact verify-and-skip of converged queries)")
    ap.add_argument("--cpu-baseline-detail", action="store_true", help="add the SURVEY 8d CPU variants (takes ~30 s more)")
    ap.add_argument("--resident-pairs", type=int, default=1, help="default mode: this many independent resident pairs per GPU, aligned concurrently "
                    "on their own HIP streams in every step (the matcher is latency-bound: concurrent pairs fill the idle issue slots); "
                    "1 = the configs[1] workload as BASELINE.json states it")
    ap.add_argument("--contexts", type=int, default=4, help="batch mode: contexts (= HIP streams, host threads) per rank")
    ap.add_argument("--pairs", type=int, default=0, help="batch mode (configs[3]): align this many consecutive scan pairs per step, "
                    "sharded pair p -> rank p mod N, uploads and index builds INSIDE the timed region, one pose gather per step")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from icp_amd import binding, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the ICP hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    if args.pairs > 0:
        return batch_mode(args, world, rank, local_rank)

    # ---- workload: pair (rank, rank+1) of the synthetic 45-scan sequence -------------------------
    pair = synth.eth_like_pair(rank % 44, n_tilt=args.n_tilt, n_beam=args.n_beam)
    n_src, n_tgt = len(pair["src_pts"]), len(pair["tgt_pts"])
    opt = binding.LinearICPOptimizer(local_rank)
    opt.setMatchingMethod(0); opt.setMatchingMaxDistance(10.0)          # main.cpp:360-361
    opt.setMetric(1); opt.setNbOfIterations(args.iterations)             # main.cpp:364-366
    opt.setWeightingMethod(0); opt.setRejectionMethod(1)
    opt.setKnnBackend(1 if args.knn == "lbvh" else 0)
    ctx = opt.ctx
    ctx.params.knn_incremental = 0 if args.no_incremental else 1
    ctx.push_params()
    ctx.set_stage_timing(args.stage_timing)
    ctx.set_target(pair["tgt_pts"], pair["tgt_nrm"], None)              # resident in HBM before the timed region
    ctx.set_source(pair["src_pts"], pair["src_nrm"], None)
    eye = binding.pose_to_c(np.eye(4, dtype=np.float32))
    poses_dev = torch.zeros(16, dtype=torch.float32, device="cuda")
    gathered = [torch.zeros(16, dtype=torch.float32, device="cuda") for _ in range(world)] if world > 1 else None

    # optional extra resident pairs on their own contexts / streams / host threads (ctypes releases the GIL)
    R = max(1, args.resident_pairs)
    extra = []
    for r in range(1, R):
        pr = synth.eth_like_pair((rank * R + r) % 44, n_tilt=args.n_tilt, n_beam=args.n_beam)
        o = binding.LinearICPOptimizer(local_rank)
        o.setMatchingMethod(0); o.setMatchingMaxDistance(10.0); o.setMetric(1); o.setNbOfIterations(args.iterations)
        o.setWeightingMethod(0); o.setRejectionMethod(1); o.setKnnBackend(1 if args.knn == "lbvh" else 0)
        o.ctx.params.knn_incremental = 0 if args.no_incremental else 1
        o.ctx.push_params(); o.ctx.set_stage_timing(0)
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
            poses_dev.copy_(torch.from_numpy(pose))
            dist.all_gather(gathered, poses_dev)                        # the single pose gather of the batch
        return pose

    for _ in range(args.warmup):
        step()
    acc = dict(match_ms=0.0, weight_reject_build_ms=0.0, solve_ms=0.0, total_ms=0.0, iterations=0, sampled_iterations=0)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pose = step()
        tm = ctx.timing()
        for k in acc:
            acc[k] += tm[k]
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    iters_total = world * R * args.steps * args.iterations
    value = iters_total / elapsed
    # ---- roofline of the dominant kernel (k-NN matcher), HIP events on the context's own stream ----
    launches = max(acc["iterations"], 1)
    knn_ms = acc["match_ms"] / launches
    alg_bytes = 12 * n_src + 12 * n_tgt + 8 * n_src                    # read src xyz + tgt xyz, write Match
    have_stage = knn_ms > 0.0                                           # --stage-timing 0: no per-stage events, no kernel duration
    if not have_stage:
        knn_ms = float("inf")
    achieved = alg_bytes / (knn_ms * 1e-3) / 1e9
    pairs = float(n_src) * float(n_tgt)
    flops = pairs * 8.0                                                 # 3 sub + 3 mul + 2 add per pair (no FMA: bit-exact contract)
    traffic = None
    tfile = os.path.join(ROOT, "profiles", "traffic_latest.json")       # HBM bytes/launch from a separate rocprofv3 --pmc pass
    if os.path.exists(tfile):
        try:
            tj = json.load(open(tfile))
            if tj.get("knn") == args.knn and tj.get("n_src") == n_src:
                traffic = tj.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    gt = pair["gt"]
    P = binding.pose_from_c(pose).astype(np.float64)
    dR = P[:3, :3] @ gt[:3, :3].T
    rot_err = float(np.arctan2(0.5 * np.linalg.norm([dR[2, 1] - dR[1, 2], dR[0, 2] - dR[2, 0], dR[1, 0] - dR[0, 1]]), (np.trace(dR) - 1) / 2))
    trans_err = float(np.linalg.norm(P[:3, 3] - gt[:3, 3]))

    out = {
        "metric": "ICP iterations/s (k-NN + point-to-plane linear, 370k-point ETH-Apartment-like pair)",
        "value": value, "unit": "ICP iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "configs[1]: synthetic ETH-Apartment-like pair (rank, rank+1), %d x %d pts, exact %s k-NN, "
                               "point-to-plane linear, maxDist^2=10, %d iterations/step, rejection on" % (n_src, n_tgt, args.knn, args.iterations),
                   "pairs_per_step": world * R, "resident_pairs_per_gpu": R, "iterations_per_step": args.iterations, "knn_backend": args.knn,
                   "knn_incremental": (not args.no_incremental) and args.knn == "lbvh"},
        "correspondences_per_s": value * n_src,
        "ms_per_iteration": elapsed / (args.steps * args.iterations) * 1e3,
        "stage_ms_per_iteration": {"match": acc["match_ms"] / launches, "weight_reject_build": acc["weight_reject_build_ms"] / launches,
                                   "solve": acc["solve_ms"] / launches},
        "roofline": {"kernel": "k_knn_bvh_post<3> (exact BVH 1-NN + weight/reject/accumulate epilogue)" if args.knn == "lbvh" else "k_knn_brute<3>", "bound": "hbm", "achieved": achieved if have_stage else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS if have_stage else None, "traffic": traffic,
                     "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": knn_ms if have_stage else None,
                     "timed_launches": acc["sampled_iterations"], "launches": acc["iterations"],
                     "note": "HIP events on the context's stream around the matcher of every %s iteration of the timed region "
                             "(offset rotating per step)" % ("" if args.stage_timing == 1 else "%d-th" % args.stage_timing)},
        "valu_roofline": {"pair_evals_per_s": pairs / (knn_ms * 1e-3) if args.knn == "brute" else None,
                          "achieved_tflops": flops / (knn_ms * 1e-3) / 1e12 if args.knn == "brute" else None,
                          "peak_tflops": FP32_VALU_PEAK_TFLOPS,
                          "note": "brute force is FP32-VALU-bound, not HBM-bound (SURVEY.md 8d); 8 non-fused flop per pair"},
        "pose_error_vs_gt": {"rot_rad": rot_err, "trans_m": trans_err},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(pair)
        if args.cpu_baseline_detail:
            out["cpu_baseline"]["detail"] = cpu_baseline_detail(pair, out["cpu_baseline"]["cores"])
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
