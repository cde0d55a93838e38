"""The search as the LOOP runs it -- seeded with the previous iteration's neighbours, verify-and-skip tiers, shared walks, spread start
(dev_fused.hpp / dev_bvh.hpp knn_walk_shared) -- in front of the oracle, match for match.  icp_match_seeded drives the fused matcher
launch by launch with poses the test dictates (launch 0 unseeded, launch j seeded exactly as iteration j of icp_run) and returns the
last launch's Match records and squared distances; they must equal the oracle's exact search at that pose bit for bit
(NearestNeighbor.h:81-97 semantics: squared L2 in FLANN order, first = lowest-index minimum, threshold on the squared distance).
Plus: the free-running 50-iteration configs[1] run at full size against orc.estimate_pose, every iteration."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
f32 = np.float32
LBVH = 1


@pytest.fixture(scope="module")
def eth_pair():
    from icp_amd import synth
    return synth.eth_like_pair(0)


@pytest.fixture(scope="module")
def eth_oracle_run(eth_pair, orc):
    """configs[1] on the CPU oracle: exact kd-tree matcher, fp64 normal equations ('exact' flavour), 50 iterations."""
    p = eth_pair
    kd = orc.KdTree(p["tgt_pts"])
    prm = orc.make_params(metric=1, n_iterations=50, max_distance=10.0, solver_mode=1, knn_kdtree=1); prm.kdtree = kd.h
    pose, recs = orc.estimate_pose(prm, p["src_pts"], p["src_nrm"], None, p["tgt_pts"], p["tgt_nrm"], None, np.eye(4, dtype=f32))
    assert len(recs) == 50
    poses = [np.eye(4, dtype=f32)] + [r["pose"] for r in recs]          # poses[i] = the pose iteration i searches at
    return kd, poses, recs


def make_ctx(factory, pair, rejection, loop=False, **kw):
    import os
    old = os.environ.get("ICP_HIP_PERSIST")
    os.environ["ICP_HIP_PERSIST"] = "1" if loop else "0"     # icp_match_seeded then drives k_icp_loop (all launches' worth in ONE launch) with the poses dictated
    try:
        c = factory()
    finally:
        if old is None:
            del os.environ["ICP_HIP_PERSIST"]
        else:
            os.environ["ICP_HIP_PERSIST"] = old
    c.params.max_distance = 10.0; c.params.metric = 1; c.params.n_iterations = 50; c.params.knn_backend = LBVH; c.params.rejection = rejection
    for k, v in kw.items():
        setattr(c.params, k, v)
    c.push_params()
    c.set_target(pair["tgt_pts"], pair["tgt_nrm"]); c.set_source(pair["src_pts"], pair["src_nrm"])
    return c


@pytest.mark.parametrize("upto,loop", [(1, False), (5, False), (12, False), (30, False), (5, True), (30, True)])
def test_seeded_search_fullsize_bit_exact_vs_kdtree_oracle(gpu_ctx_factory, eth_pair, eth_oracle_run, orc, upto, loop):
    """370 488 x 370 488, the oracle's own pose sequence of iterations 0..upto replayed through the fused matcher: the records of launch
    `upto` -- reached through `upto` seeded, incremental launches (loop: iterations of ONE k_icp_loop launch, the waves resident, their
    queries' data parked from iteration to iteration) -- equal the oracle's kd-tree search at that pose, idx and d2 bits."""
    kd, poses, _ = eth_oracle_run
    c = make_ctx(gpu_ctx_factory, eth_pair, rejection=0, loop=loop)
    m, d2 = c.match_seeded(poses[: upto + 1])
    mo, do = kd.query(orc.transform_points(eth_pair["src_pts"], poses[upto]), 10.0)
    assert np.array_equal(m["idx"], mo["idx"]), int((m["idx"] != mo["idx"]).sum())
    assert np.array_equal(d2.view(np.uint32), do.view(np.uint32))
    assert np.array_equal(m["weight"], mo["weight"])
    c.close()


def test_seeded_search_pair_of_launches_and_rejection_records(gpu_ctx_factory, eth_pair, eth_oracle_run, orc):
    """(pose_prev, pose) = consecutive oracle poses (0,1), (4,5), (11,12), (29,30): one unseeded launch, then ONE seeded launch with a large
    step between anchor and query (the first pair) or a tiny one (the last); and with rejection on, the records the fused epilogue writes
    equal the oracle's applyWeights + pruneCorrespondences of the exact matches."""
    kd, poses, _ = eth_oracle_run
    p = eth_pair
    c = make_ctx(gpu_ctx_factory, p, rejection=1)
    for a in (0, 4, 11, 29):
        m, d2 = c.match_seeded([poses[a], poses[a + 1]])
        q = orc.transform_points(p["src_pts"], poses[a + 1])
        mo, do = kd.query(q, 10.0)
        assert np.array_equal(d2.view(np.uint32), do.view(np.uint32)), a
        sn = orc.transform_normals(p["src_nrm"], poses[a + 1])
        mw = orc.apply_weights(0, 10.0, q, p["tgt_pts"], sn, p["tgt_nrm"], None, None, mo)
        mp = orc.prune(sn, p["tgt_nrm"], mw)
        assert np.array_equal(m["idx"], mp["idx"]) and np.array_equal(m["weight"], mp["weight"]), a
    c.close()


def test_free_run_fullsize_every_iteration_vs_oracle(gpu_ctx_factory, eth_pair, eth_oracle_run):
    """configs[1] free-running at full size (ICPOptimizer.h:540-656): all 50 per-iteration valid counts equal the oracle's, all 50 poses
    within 1e-5 rad / 1e-5 m of it (the oracle's 'exact' flavour: same fp32 rows, fp64 sums and factorisation)."""
    from conftest import pose_error
    _, _, recs_o = eth_oracle_run
    c = make_ctx(gpu_ctx_factory, eth_pair, rejection=1)
    pose, recs, rc = c.run(np.eye(4))
    assert rc == 0 and len(recs) == 50
    worst = (0.0, 0.0)
    for k, (a, b) in enumerate(zip(recs, recs_o)):
        assert a["n_valid"] == b["n_valid"], (k, a["n_valid"], b["n_valid"])
        ang, tr = pose_error(a["pose"], b["pose"])
        assert ang < 1e-5 and tr < 1e-5, (k, ang, tr)
        worst = (max(worst[0], ang), max(worst[1], tr))
    print("free run vs oracle: worst rotation %.3g rad, worst translation %.3g m over 50 iterations" % worst)
    c.close()


def stress_cloud(rng, n, kind):
    if kind == 0:                                           # exact duplicates of a small set: lowest index must win every tie
        base = rng.uniform(-1, 1, (max(1, n // 4), 3)); p = base[rng.integers(0, len(base), n)]
    elif kind == 1:                                         # coarse grid: ties at every level of the tree
        p = rng.integers(-6, 7, (n, 3)) * 0.125
    else:                                                   # thin noisy plane + a far cluster
        p = np.c_[rng.uniform(-1, 1, (n, 2)), rng.normal(0, 1e-3, n)]
        m = max(1, n // 10); p[:m] = rng.normal(0, 0.01, (m, 3)) + np.array([5.0, 5.0, 5.0])
    return p.astype(f32)


def small_motion(rng, scale):
    w = rng.normal(size=3) * scale; t = rng.normal(size=3) * scale
    th = np.linalg.norm(w); k = w / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    T = np.eye(4); T[:3, :3] = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K; T[:3, 3] = t
    return T


@pytest.mark.parametrize("seed", range(9))
def test_seeded_search_on_tie_clouds_vs_brute_force_oracle(gpu_ctx_factory, orc, seed):
    """Clouds built to provoke a lowest-index violation in the shortcuts (exact duplicates, a coarse grid, queries on the bisector of
    two targets): a chain of 6 seeded launches along a shrinking random motion, the last launch's idx / d2 against the oracle's
    brute-force scan (strict <, first minimum: NearestNeighbor.h:87)."""
    rng = np.random.default_rng(7000 + seed)
    nt = int(rng.integers(200, 20000)); ns = int(rng.integers(200, 6000))
    tgt = stress_cloud(rng, nt, seed % 3)
    if seed % 2 == 0:                                       # on the bisector of two targets: the runner-up is exactly as close as the neighbour
        a = tgt[rng.integers(0, nt, ns)]; b = tgt[rng.integers(0, nt, ns)]
        src = (0.5 * (a.astype(np.float64) + b.astype(np.float64))).astype(f32)
    else:
        src = (tgt[rng.integers(0, nt, ns)] + rng.normal(0, 0.02, (ns, 3))).astype(f32)
    v = rng.normal(size=(nt, 3)); tn = (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(f32)
    v = rng.normal(size=(ns, 3)); sn = (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(f32)
    c = gpu_ctx_factory()
    c.params.max_distance = 4.0; c.params.metric = 1; c.params.rejection = 0; c.params.knn_backend = LBVH; c.push_params()
    c.set_target(tgt, tn); c.set_source(src, sn)
    poses = [np.eye(4, dtype=f32)]
    for j in range(6):                                      # ICP-like: the steps shrink; the last ones are exactly zero (converged: every query verifies)
        step = small_motion(rng, 0.05 * 0.3 ** j) if j < 4 else np.eye(4)
        poses.append((step @ poses[-1].astype(np.float64)).astype(f32))
    for upto in (1, 3, 6):
        m, d2 = c.match_seeded(poses[: upto + 1])
        mo, do = orc.knn3(orc.transform_points(src, poses[upto]), tgt, 4.0)
        assert np.array_equal(m["idx"], mo["idx"]), (upto, int((m["idx"] != mo["idx"]).sum()))
        assert np.array_equal(d2.view(np.uint32), do.view(np.uint32)), upto
    c.close()


@pytest.mark.parametrize("seed", range(3))
def test_seeded_search_lone_walkers_small_and_huge_radius(gpu_ctx_factory, orc, seed):
    """Waves with exactly ONE walker take the level-synchronous search (knn_walk_shared, ICP_LONE_WALK): most queries sit on the target
    surface and verify from the second launch on; a sprinkling of queries on the bisector of two targets keeps walking with a small
    radius (the frontier stays small: the lone search completes), a few far outliers keep walking with a larger one, and one query sits at the centre
    of a spherical shell of 4 000 targets (every box of the shell survives: the frontier overflows and the wave must start over on the general
    path -- tools/dev_lone_counts.py counts both outcomes on this very cloud).  The motion shrinks but never stops, so both kinds walk
    in every launch.  idx / d2 of several launches of the chain against the oracle's brute-force scan."""
    rng = np.random.default_rng(9100 + seed)
    nt = 30000 + 5000 * seed
    tgt = np.c_[rng.uniform(-2, 2, (nt, 2)), rng.normal(0, 2e-3, nt)].astype(f32)
    v = rng.normal(size=(4000, 3)); shell = np.array([10.0, 10.0, 10.0]) + v / np.linalg.norm(v, axis=1, keepdims=True) * (1.0 + rng.normal(0, 1e-4, (4000, 1)))
    n_close, n_bis, n_far = 6000, 60, 12
    close = tgt[rng.integers(0, nt, n_close)] + rng.normal(0, 1e-4, (n_close, 3))
    a = tgt[rng.integers(0, nt, n_bis)]
    d = np.linalg.norm(tgt[None, :, :2] - a[:, None, :2], axis=2); d[d == 0] = np.inf
    b = tgt[np.argmin(d, axis=1)]                           # a near neighbour on the surface: the query between them has two candidates
    bis = 0.5 * (a.astype(np.float64) + b.astype(np.float64))
    far = np.c_[rng.uniform(-2, 2, (n_far, 2)), rng.uniform(1.5, 3.0, n_far)]
    tgt = np.r_[tgt, shell.astype(f32)]; nt = len(tgt)       # ... and ONE query at the centre of a shell of 4 000 targets: every box of the shell survives its radius
    src = np.r_[close, bis, far, np.array([[10.0, 10.0, 10.0]])]
    src = src[rng.permutation(len(src))].astype(f32)
    v = rng.normal(size=(nt, 3)); tn = (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(f32)
    v = rng.normal(size=(len(src), 3)); sn = (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(f32)
    c = gpu_ctx_factory()
    c.params.max_distance = 100.0; c.params.metric = 1; c.params.rejection = 0; c.params.knn_backend = LBVH; c.push_params()
    c.set_target(tgt, tn); c.set_source(src, sn)
    poses = [np.eye(4, dtype=f32)]
    for j in range(7):
        poses.append((small_motion(rng, max(0.01 * 0.3 ** j, 2e-5)) @ poses[-1].astype(np.float64)).astype(f32))
    for upto in (2, 4, 5, 7):
        m, d2 = c.match_seeded(poses[: upto + 1])
        mo, do = orc.knn3(orc.transform_points(src, poses[upto]), tgt, 100.0)
        assert np.array_equal(m["idx"], mo["idx"]), (upto, int((m["idx"] != mo["idx"]).sum()))
        assert np.array_equal(d2.view(np.uint32), do.view(np.uint32)), upto
    c.close()


def test_seeded_search_colour_6d_vs_oracle(gpu_ctx_factory, orc):
    """The 6-D instantiation of the same fused matcher (colour ICP), seeded: bit-exact against the oracle's 6-D scan."""
    from icp_amd import synth
    K = np.array([[131.25, 0, 79.5], [0, 131.25, 59.5], [0, 0, 1]], f32)
    r = synth.rgbd_pair(0, width=160, height=120, K=K)
    tp, tn, tc = synth.compact_valid(r["tgt_pts"], r["tgt_nrm"], r["tgt_rgba"]); sp, sn, sc = synth.compact_valid(r["src_pts"], r["src_nrm"], r["src_rgba"])
    c = gpu_ctx_factory()
    c.params.max_distance = 0.1; c.params.metric = 1; c.params.rejection = 0; c.params.color_icp = 1; c.params.knn_backend = LBVH; c.push_params()
    c.set_target(tp, tn, tc); c.set_source(sp, sn, sc)
    rng = np.random.default_rng(5)
    poses = [np.eye(4, dtype=f32)]
    for j in range(4):
        poses.append((small_motion(rng, 0.01 * 0.4 ** j) @ poses[-1].astype(np.float64)).astype(f32))
    for upto in (1, 4):
        m, d2 = c.match_seeded(poses[: upto + 1])
        mo, do = orc.knn6(orc.transform_points(sp, poses[upto]), sc, tp, tc, 0.1)
        assert np.array_equal(m["idx"], mo["idx"]) and np.array_equal(d2.view(np.uint32), do.view(np.uint32)), upto
    c.close()


def test_match_seeded_argument_errors(gpu_ctx_factory, bunny):
    from icp_amd import binding
    c = gpu_ctx_factory()
    c.params.metric = 1; c.params.knn_backend = 0; c.push_params()                # brute-force backend: no fused matcher
    c.set_target(bunny["tgt_pts"], bunny["tgt_nrm"]); c.set_source(bunny["src_pts"], bunny["src_nrm"])
    with pytest.raises(binding.IcpError) as e:
        c.match_seeded([np.eye(4)])
    assert e.value.code == 1
    c.params.knn_backend = LBVH; c.params.metric = 2; c.push_params()             # symmetric: two passes, not the fused matcher
    with pytest.raises(binding.IcpError):
        c.match_seeded([np.eye(4)])
    c.params.metric = 1; c.params.max_distance = 0.0003; c.push_params()
    m, d2 = c.match_seeded([np.eye(4)])                                             # one pose = the unseeded launch = icp_match
    m1, d1 = c.match(np.eye(4))
    assert np.array_equal(d2.view(np.uint32), d1.view(np.uint32))
    c.close()
