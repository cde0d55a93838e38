"""GPU tests of the BASELINE.json configurations that round 1 left untested, plus the round-2 entry points:
  * configs[2] at its stated organised geometry 1077 x 344 (projective + normal-angle rejection + symmetric linear),
  * configs[3] on one GPU: a batch of pairs through icp_batch_run (several contexts) + the pose gather, per-pair oracle check,
  * real-data path: PCD + pose CSV on disk -> icp_estimate_normals(k = 5) -> icp_run (ETHDataLoader.h:40-101, main.cpp:401-457),
  * icp_backproject_depth without colours, the record of an empty first iteration, per-iteration stage times.
Every comparison is HIP path vs the CPU oracle (our restatement of the reference -- parity unpinned)."""
import json
import os
import subprocess
import sys
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
f32 = np.float32
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


# ------------------------------------------------------------------------------------------------ configs[2]
def test_config2_projective_symmetric_1077x344(gpu_ctx_factory, orc):
    """SURVEY.md 8d config 3 / BASELINE configs[2]: organised pinhole image 1077 x 344 = 370 488 px, fx = fy = 540, cx = 538,
    cy = 171.5; projective matcher (window 12, NearestNeighbor.h:333-421), rejection, symmetric linear (ICPOptimizer.h:784-898),
    maxDist^2 = 0.1 (main.cpp:245), 35 iterations (main.cpp:226-228)."""
    from conftest import pose_error
    from icp_amd import synth
    W, H = 1077, 344
    K = np.array([[540.0, 0, 538.0], [0, 540.0, 171.5], [0, 0, 1]])
    r = synth.rgbd_pair(0, width=W, height=H, K=K)
    assert len(r["src_pts"]) == 370488 and len(r["tgt_pts"]) == W * H
    c = gpu_ctx_factory()
    c.params.matching = 1; c.params.metric = 2; c.params.max_distance = 0.1; c.params.n_iterations = 35
    c.params.fx, c.params.fy, c.params.cx, c.params.cy, c.params.width, c.params.height = 540.0, 540.0, 538.0, 171.5, W, H
    c.push_params(); c.set_target(r["tgt_pts"], r["tgt_nrm"]); c.set_source(r["src_pts"], r["src_nrm"])
    # matches and squared distances, all 370 488 queries, at identity and at the ground-truth motion: bit-exact
    for T in (np.eye(4, dtype=f32), r["gt"].astype(f32)):
        m, d2 = c.match(T)
        mo, do = orc.projective(orc.transform_points(r["src_pts"], T), r["tgt_pts"], W, H, K, 0.1)
        assert np.array_equal(m["idx"], mo["idx"]) and np.array_equal(m["weight"], mo["weight"])
        assert np.array_equal(d2.view(np.uint32), do.view(np.uint32))
    # the 35-iteration symmetric run, teacher-forced: from the oracle's pose at iteration k the device lands within 1e-5 of the
    # oracle's next pose (exact-flavour oracle: same fp32 rows, fp64 normal equations)
    prm = orc.make_params(metric=2, matching=1, n_iterations=1, max_distance=0.1, solver_mode=1, K=K, width=W, height=H)
    pose = np.eye(4, dtype=f32)
    for k in range(35):
        po, mo, nvo, _, _ = orc.iterate(prm, r["src_pts"], r["src_nrm"], None, r["tgt_pts"], r["tgt_nrm"], None, pose)
        pg, st = c.iterate(pose)
        assert st["n_valid"] == nvo, k
        ang, tr = pose_error(pg, po)
        assert ang < 1e-5 and tr < 1e-5, (k, ang, tr)
        pose = po
    # free-running: the loop registers the frames and is deterministic
    a, ra, _ = c.run(np.eye(4))
    b, rb, _ = c.run(np.eye(4))
    assert len(ra) == 35 and np.array_equal(a, b)
    ang, tr = pose_error(a, r["gt"])
    assert ang < 1e-3 and tr < 2e-3
    ang, tr = pose_error(a, pose)                                       # and ends where the oracle's chain ended
    assert ang < 1e-4 and tr < 1e-4


# ------------------------------------------------------------------------------------------------ configs[3]
@pytest.fixture(scope="module")
def small_batch():
    from icp_amd import synth
    return [synth.eth_like_pair(k, n_tilt=36, n_beam=110) for k in range(7)]


def _oracle_pose(orc, d, iters):
    kd = orc.KdTree(d["tgt_pts"])
    prm = orc.make_params(metric=1, n_iterations=iters, max_distance=10.0, solver_mode=1, knn_kdtree=1); prm.kdtree = kd.h
    pose, recs = orc.estimate_pose(prm, d["src_pts"], d["src_nrm"], None, d["tgt_pts"], d["tgt_nrm"], None, np.eye(4, dtype=f32))
    return pose


@pytest.mark.parametrize("knn", [1, 0])
def test_config3_batch_on_one_gpu(gpu_ctx_factory, orc, small_batch, knn):
    """The loop over independent pairs (main.cpp:411-498) as ONE icp_batch_run call with 3 contexts (3 host threads, 3 HIP streams):
    every pose within 1e-5 of the oracle's estimatePose for THAT pair and returned in pair order; then the gather (batch.align_batch,
    and icp_gather_poses on a 1-rank RCCL communicator) keeps that order."""
    from conftest import pose_error
    from icp_amd import binding, batch
    iters = 30
    ctxs = []
    for _ in range(3):
        c = gpu_ctx_factory()
        c.params.max_distance = 10.0; c.params.metric = 1; c.params.n_iterations = iters; c.params.knn_backend = knn
        c.push_params(); ctxs.append(c)
    poses, status, rc = binding.batch_run(ctxs, small_batch)
    assert rc == 0 and status.tolist() == [0] * len(small_batch)
    for p, d in enumerate(small_batch):
        po = _oracle_pose(orc, d, iters)
        ang, tr = pose_error(binding.pose_from_c(poses[p]), po)
        assert ang < 1e-5 and tr < 1e-5, (p, ang, tr)
        ang, tr = pose_error(binding.pose_from_c(poses[p]), d["gt"])
        assert tr < 0.05
    # same result one pair at a time on one context (the batch is order- and thread-independent, bit for bit)
    one, _, rc1 = binding.batch_run(ctxs[:1], small_batch)
    assert rc1 == 0 and np.array_equal(one, poses)
    # gather: pair order is kept
    calls = []

    def solve(p):
        calls.append(p)
        return poses[p]
    assert np.array_equal(batch.align_batch(len(small_batch), solve, device="cpu"), poses) and calls == list(range(len(small_batch)))
    comm = binding.Comm(0, 1, 0, binding.Comm.unique_id())              # RCCL loaded at run time; ncclAllGather on one rank
    assert np.array_equal(comm.gather_poses(poses, len(small_batch)), poses)
    comm.close()


def test_config3_batch_of_a_scan_sequence_shares_the_scans(gpu_ctx_factory, orc):
    """The ETH loop aligns scan k + 1 to scan k for consecutive k (main.cpp:411-498): scan k + 1 is the source of pair k and the target of
    pair k + 1.  Handed over as the SAME arrays, icp_batch_run uploads such a scan once per run of pairs and promotes it from source to
    target on the device.  Same poses, bit for bit, as the batch built from copies (every target uploaded), with 1 or 3 contexts; and
    every pose within 1e-5 of the oracle started from the same perturbed pose (main.cpp:420-429: the perturbation is the initial pose)."""
    from conftest import pose_error
    from icp_amd import binding, synth
    iters = 30
    scans = [synth.laser_scan(synth.scan_pose(k, 0xE7A0), 0xE7A0 + k, 36, 110, 0.01) for k in range(8)]
    init = [synth.perturbation(0xE7A0 + 100003 * (k + 1)) for k in range(7)]
    chain = [dict(src_pts=scans[k + 1][0], src_nrm=scans[k + 1][1], tgt_pts=scans[k][0], tgt_nrm=scans[k][1]) for k in range(7)]
    copies = [{key: np.array(v, copy=True) for key, v in d.items()} for d in chain]
    assert chain[1]["tgt_pts"] is chain[0]["src_pts"] and copies[1]["tgt_pts"] is not copies[0]["src_pts"]
    ctxs = []
    for _ in range(3):
        c = gpu_ctx_factory()
        c.params.max_distance = 10.0; c.params.metric = 1; c.params.n_iterations = iters; c.params.knn_backend = 1
        c.push_params(); ctxs.append(c)
    shared3, st, rc = binding.batch_run(ctxs, chain, init)
    assert rc == 0 and st.tolist() == [0] * 7
    shared1, _, rc1 = binding.batch_run(ctxs[:1], chain, init)
    plain3, _, rc2 = binding.batch_run(ctxs, copies, init)
    assert rc1 == 0 and rc2 == 0
    assert np.array_equal(shared3, shared1) and np.array_equal(shared3, plain3)
    for k, d in enumerate(chain):
        kd = orc.KdTree(d["tgt_pts"])
        prm = orc.make_params(metric=1, n_iterations=iters, max_distance=10.0, solver_mode=1, knn_kdtree=1); prm.kdtree = kd.h
        po, _ = orc.estimate_pose(prm, d["src_pts"], d["src_nrm"], None, d["tgt_pts"], d["tgt_nrm"], None, init[k].astype(f32))
        ang, tr = pose_error(binding.pose_from_c(shared3[k]), po)
        assert ang < 1e-5 and tr < 1e-5, (k, ang, tr)


def test_batch_run_reports_per_pair_errors(gpu_ctx_factory, small_batch):
    from icp_amd import binding
    c = gpu_ctx_factory()
    c.params.max_distance = 1e-12; c.params.metric = 1; c.params.n_iterations = 3; c.params.knn_backend = 1; c.push_params()
    poses, status, rc = binding.batch_run([c], small_batch[:2])
    assert rc == binding.ERR_NO_CORRESPONDENCES and status.tolist() == [binding.ERR_NO_CORRESPONDENCES] * 2
    assert np.array_equal(poses[0], binding.pose_to_c(np.eye(4)))        # nothing matched: the pose never moved


# ------------------------------------------------------------------------------------------------ real-data path
def _write_eth_dataset(tmp_path, n_pairs=2, n_tilt=48, n_beam=150):
    from icp_amd import synth, eth
    pairs = []
    for k in range(n_pairs):
        d = synth.eth_like_pair(k, n_tilt=n_tilt, n_beam=n_beam)
        d["pose"] = synth.perturbation(11 + k, scale=1.0)               # benchmark-style perturbation; the driver scales it by 0.1
        pairs.append(d)
    eth.write_synthetic_dataset(str(tmp_path), "apartment_global.csv", pairs)
    return pairs


def test_eth_files_to_pose_end_to_end(gpu_ctx_factory, orc, tmp_path):
    """alignETH on files (main.cpp:401-457): pose CSV row -> two PCD scans -> k = 5 normals on the device -> pose_scaling 0.1
    perturbation -> 50 point-to-plane iterations; checked against the oracle on the same prepared clouds."""
    from conftest import pose_error
    from icp_amd import eth, binding
    written = _write_eth_dataset(tmp_path)
    rows = eth.load_rows(str(tmp_path), "apartment_global.csv")
    assert len(rows) == 2 and eth.dataset_name("apartment_global.csv") == "apartment" and eth.dataset_name("eth/plain_global.csv") == "eth/plain"
    c = gpu_ctx_factory()
    for row, w in zip(rows, written):
        src, tgt = eth.load_scans(str(tmp_path), "apartment_global.csv", row)
        assert np.array_equal(src, w["src_unperturbed"]) and np.array_equal(tgt, w["tgt_pts"])      # binary PCD round trip is exact
        pair = eth.prepare_pair(c, src, tgt, row["pose"])
        assert np.isfinite(pair["src_nrm"]).all() and np.allclose(np.linalg.norm(pair["tgt_nrm"], axis=1), 1.0, atol=1e-4)
        c.params.max_distance = 10.0; c.params.metric = 1; c.params.n_iterations = 50; c.params.knn_backend = 1; c.params.record_rmse = 3
        c.push_params()
        c.set_convergence_reference(pair["src_pts"], pair["src_unperturbed"])        # ConvergenceMeasure(source, original_source, true)
        c.set_target(pair["tgt_pts"], pair["tgt_nrm"]); c.set_source(pair["src_pts"], pair["src_nrm"])
        pose, recs, _ = c.run(np.eye(4))
        po = _oracle_pose(orc, pair, 50)
        ang, tr = pose_error(pose, po)
        assert ang < 1e-5 and tr < 1e-5
        e0 = orc.benchmark_error(pair["src_pts"], pair["src_unperturbed"], np.eye(4))
        assert recs[-1]["benchmark_error"] < 0.5 * e0 and abs(recs[-1]["benchmark_error"] - orc.benchmark_error(pair["src_pts"], pair["src_unperturbed"], pose)) < 2e-6


def test_bench_eth_dir_is_one_command(tmp_path):
    """`python bench.py --eth-dir DIR` runs the files end to end and says so in its JSON line."""
    _write_eth_dataset(tmp_path, n_pairs=1)
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--eth-dir", str(tmp_path), "--steps", "2", "--warmup", "1",
                                   "--stage-timing", "1", "--no-cpu-baseline"], timeout=600).decode()
    line = json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1])
    assert line["data"] == "eth" and line["n_gpus"] == 1 and line["value"] > 0
    assert "apartment_global.csv" in line["config"]["workload"] and line["n_valid_last"] > 1000
    assert line["roofline"]["algorithmic_bytes_per_launch"] == (12 + 12 + 8 + 56) * 48 * 150
    assert line["per_regime_ms"]["match_iteration_0"] > 0 and line["no_incremental_value"] > 0


# ------------------------------------------------------------------------------------------------ advisor findings of round 1
def test_backproject_depth_without_colours(gpu_ctx_factory, orc):
    """rgbx == NULL (the header allows it): the staging layout must still hold the valid mask; compared with the oracle."""
    from icp_amd import synth
    W, H = 640, 480
    K = np.array([[525.0, 0, 319.5], [0, 525.0, 239.5], [0, 0, 1]], f32)
    r = synth.rgbd_pair(0)
    depth = r["tgt_pts"][:, 2].reshape(H, W).copy()
    c = gpu_ctx_factory()
    c.set_target(r["tgt_pts"], r["tgt_nrm"])                            # neighbouring device buffers that an overrun would hit
    E = synth.make_pose((0.02, -0.01, 0.03), (0.1, 0.2, -0.1))
    g = c.backproject_depth(depth, None, K, extrinsics=E)
    o = orc.backproject(depth, None, K, extrinsics=E)
    assert g[2] is None and o[2] is None
    assert np.array_equal(g[0].view(np.uint32), o[0].view(np.uint32)) and np.array_equal(g[1].view(np.uint32), o[1].view(np.uint32))
    assert np.array_equal(g[3], o[3]) and 0.7 < g[3].mean() < 1.0
    g2 = c.backproject_depth(depth, r["tgt_rgba"], K, extrinsics=E)      # and with colours right after, same context
    assert np.array_equal(g2[0].view(np.uint32), g[0].view(np.uint32)) and np.array_equal(g2[3], g[3])


def test_record_of_an_empty_first_iteration_keeps_the_incoming_pose(gpu_ctx_factory, bunny):
    """RANDOM_SAMPLING with a tiny probability: a seed for which iteration 0 selects nothing and later iterations do.  The record
    of iteration 0 must carry the incoming pose, not the final one."""
    from icp_amd import binding, synth
    n = len(bunny["src_pts"]); proba = f32(0.003)
    th = int(float(proba) * 4294967296.0)
    seed = None
    for s in range(1, 4000):
        if any(binding.select_hash(s, 0, i) < th for i in range(n)):
            continue                                                    # iteration 0 selects something (about 96 % of the seeds)
        if sum(1 for i in range(n) if binding.select_hash(s, 1, i) < th) >= 3:
            seed = s; break
    assert seed is not None
    c = gpu_ctx_factory()
    c.params.max_distance = 10.0; c.params.metric = 0; c.params.n_iterations = 4; c.params.selection = 1
    c.params.selection_proba = float(proba); c.params.selection_seed = seed; c.push_params()
    c.set_target(bunny["tgt_pts"], bunny["tgt_nrm"]); c.set_source(bunny["src_pts"], bunny["src_nrm"])
    T0 = synth.make_pose((0.01, -0.02, 0.015), (0.001, 0.002, -0.001)).astype(f32)
    pose, recs, rc = c.run(T0, check=False)
    assert rc == binding.ERR_NO_CORRESPONDENCES and recs[0]["n_src"] == 0 and recs[0]["status"] == binding.ERR_NO_CORRESPONDENCES
    assert recs[1]["n_src"] >= 3
    assert np.array_equal(recs[0]["pose"], T0)
    assert not np.array_equal(pose, T0) and np.array_equal(recs[-1]["pose"], pose)


def test_iteration_times_and_params_validation(gpu_ctx_factory, bunny):
    from icp_amd import binding
    c = gpu_ctx_factory()
    c.params.max_distance = 0.0003; c.params.metric = 1; c.params.n_iterations = 6; c.push_params()
    c.set_target(bunny["tgt_pts"], bunny["tgt_nrm"]); c.set_source(bunny["src_pts"], bunny["src_nrm"])
    c.set_stage_timing(1); c.run(np.eye(4))
    a, b, d = c.iteration_times()
    assert len(a) == 6 and (a > 0).all() and (d > 0).all() and abs(a.sum() - c.timing()["match_ms"]) < 1e-3
    c.set_stage_timing(3); c.run(np.eye(4))
    a, _, _ = c.iteration_times()
    assert (a > 0).sum() == 2 and (a < 0).sum() == 4
    c.set_stage_timing(1)
    for field, bad in (("knn_backend", 7), ("width", -1), ("height", -5), ("max_distance", float("nan")), ("selection_proba", float("nan"))):
        keep = getattr(c.params, field)
        setattr(c.params, field, bad)
        with pytest.raises(binding.IcpError):
            c.push_params()
        setattr(c.params, field, keep)
    c.push_params()


@pytest.mark.parametrize("order", ["binding_first", "torch_first", "binding_only"])
def test_comm_finds_the_rccl_of_its_own_hip_runtime(order):
    """PyTorch wheels ship a second ROCm stack (libamdhip64, libhsa-runtime64, librccl).  Whichever of torch / this library a process
    imports first, icp_comm_create has to load the RCCL that belongs to the HIP runtime libicp_hip.so is bound to (a fresh process
    per order: the loader state is the thing under test)."""
    import subprocess, sys
    imports = {"binding_first": "from icp_amd import binding; import torch", "torch_first": "import torch; from icp_amd import binding",
               "binding_only": "from icp_amd import binding"}[order]
    code = ("import sys, numpy as np; sys.path.insert(0, %r); %s\n"
            "c = binding.Context(0)\n"
            "comm = binding.Comm(0, 1, 0, binding.Comm.unique_id())\n"
            "p = np.arange(32, dtype=np.float32).reshape(2, 16)\n"
            "assert np.array_equal(comm.gather_poses(p, 2), p)\n"
            "comm.close(); print('gathered')\n") % (os.path.join(ROOT, "icp-variants_amd", "python"), imports)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "gathered" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
