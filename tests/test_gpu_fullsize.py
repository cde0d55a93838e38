"""GPU tests at BASELINE.json's full size (370 488 x 370 488, configs[1]) -- size-independent properties plus a
full bit-exact check against the oracle's exact kd-tree (bit-identical to its brute-force scan, test_oracle.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
f32 = np.float32


@pytest.fixture(scope="module")
def eth_pair():
    from icp_amd import synth
    return synth.eth_like_pair(0)


@pytest.fixture(scope="module")
def eth_ctx(gpu_ctx_factory, eth_pair):
    c = gpu_ctx_factory()
    c.params.max_distance = 10.0; c.params.metric = 1; c.params.n_iterations = 50          # main.cpp:361,364-366
    c.push_params()
    c.set_target(eth_pair["tgt_pts"], eth_pair["tgt_nrm"]); c.set_source(eth_pair["src_pts"], eth_pair["src_nrm"])
    return c


def test_fullsize_match_bit_exact_vs_kdtree_oracle(eth_ctx, eth_pair, orc):
    assert len(eth_pair["src_pts"]) == 370488
    m, d2 = eth_ctx.match(np.eye(4))
    kd = orc.KdTree(eth_pair["tgt_pts"])
    mo, do = kd.query(orc.transform_points(eth_pair["src_pts"], np.eye(4)), 10.0)
    assert np.array_equal(m["idx"], mo["idx"])
    assert np.array_equal(d2.view(np.uint32), do.view(np.uint32))
    # property: the reported d2 is the fp32 distance to the reported target, and a 512-query subsample agrees with the plain scan
    t = eth_pair["tgt_pts"][m["idx"]]; q = eth_pair["src_pts"]
    dx, dy, dz = q[:, 0] - t[:, 0], q[:, 1] - t[:, 1], q[:, 2] - t[:, 2]
    assert np.array_equal(((dx * dx + dy * dy) + dz * dz).view(np.uint32), d2.view(np.uint32))
    sub = np.random.default_rng(0).choice(len(q), 512, replace=False)
    ms, ds = orc.knn3(q[sub], eth_pair["tgt_pts"], 10.0)
    assert np.array_equal(ms["idx"], m["idx"][sub])


def test_fullsize_teacher_forced_iterations(eth_ctx, eth_pair, orc):
    """From the oracle's pose at iteration k the device iteration lands within 1e-5 of the oracle's next pose."""
    from conftest import pose_error
    p = eth_pair
    kd = orc.KdTree(p["tgt_pts"])
    prm = orc.make_params(metric=1, n_iterations=1, max_distance=10.0, solver_mode=1, knn_kdtree=1); prm.kdtree = kd.h
    pose = np.eye(4, dtype=f32)
    for k in range(4):
        po, mo, nvo, _, _ = orc.iterate(prm, p["src_pts"], p["src_nrm"], None, p["tgt_pts"], p["tgt_nrm"], None, pose)
        pg, st = eth_ctx.iterate(pose)
        assert st["n_valid"] == nvo
        ang, tr = pose_error(pg, po)
        assert ang < 1e-5 and tr < 1e-5, (k, ang, tr)
        pose = po
    # reference-shaped fp32 solve of the same system (1.48M x 6 in fp32): its own rounding noise, for the record
    prm0 = orc.make_params(metric=1, n_iterations=1, max_distance=10.0, solver_mode=0, knn_kdtree=1); prm0.kdtree = kd.h
    pf, _, _, _, _ = orc.iterate(prm0, p["src_pts"], p["src_nrm"], None, p["tgt_pts"], p["tgt_nrm"], None, np.eye(4))
    pg, _ = eth_ctx.iterate(np.eye(4))
    ang, tr = pose_error(pg, pf)
    assert ang < 1e-4 and tr < 1e-4


def test_fullsize_run_converges_and_is_deterministic(eth_ctx, eth_pair):
    from conftest import pose_error
    a, ra, _ = eth_ctx.run(np.eye(4))
    b, rb, _ = eth_ctx.run(np.eye(4))
    assert np.array_equal(a, b) and [r["n_valid"] for r in ra] == [r["n_valid"] for r in rb]     # fixed-order reductions
    assert len(ra) == 50
    ang, tr = pose_error(a, eth_pair["gt"])
    assert ang < 2e-3 and tr < 5e-3
    t = eth_ctx.timing()
    assert t["iterations"] == 50 and t["match_ms"] > 0


def test_fullsize_projective_tum_geometry(gpu_ctx_factory, orc):
    """configs[2] at native TUM geometry (640 x 480, K from VirtualSensor.h:44-46): projective matches bit-exact vs the oracle
    over all 307 200 queries, then the symmetric ICP run registers the frames."""
    from conftest import pose_error
    from icp_amd import synth
    r = synth.rgbd_pair(0)
    W, H, K = r["width"], r["height"], r["K"]
    c = gpu_ctx_factory()
    c.params.matching = 1; c.params.metric = 2; c.params.max_distance = 0.1; c.params.n_iterations = 35            # main.cpp:226-228,245
    c.params.fx, c.params.fy, c.params.cx, c.params.cy, c.params.width, c.params.height = float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]), W, H
    c.push_params(); c.set_target(r["tgt_pts"], r["tgt_nrm"]); c.set_source(r["src_pts"], r["src_nrm"])
    assert len(r["src_pts"]) == 307200
    for T in (np.eye(4, dtype=f32), r["gt"].astype(f32)):
        m, d2 = c.match(T)
        mo, do = orc.projective(orc.transform_points(r["src_pts"], T), r["tgt_pts"], W, H, K, 0.1)
        assert np.array_equal(m["idx"], mo["idx"]) and np.array_equal(d2.view(np.uint32), do.view(np.uint32))
    pose, recs, _ = c.run(np.eye(4))
    ang, tr = pose_error(pose, r["gt"])
    assert ang < 1e-3 and tr < 2e-3 and len(recs) == 35


def test_fullsize_colour_knn_tum_geometry(gpu_ctx_factory, orc):
    """configs[4] shape: 6-D k-NN over ~290k valid target pixels; a 16k-query subsample is checked bit-exact against the
    oracle's 6-D scan, the multires colour-ICP run against the ground-truth motion."""
    from conftest import pose_error
    from icp_amd import synth
    r = synth.rgbd_pair(0)
    tp, tn, tc = synth.compact_valid(r["tgt_pts"], r["tgt_nrm"], r["tgt_rgba"])
    sp, sn, sc = r["src_pts"], r["src_nrm"], r["src_rgba"]
    c = gpu_ctx_factory()
    c.params.color_icp = 1; c.params.weighting = 3; c.params.multires = 1; c.params.metric = 1; c.params.max_distance = 0.1
    c.params.n_iterations = 35; c.params.knn_backend = 1
    c.push_params(); c.set_target(tp, tn, tc); c.set_source(sp, sn, sc)
    m, d2 = c.match(np.eye(4))
    sub = np.random.default_rng(1).choice(len(sp), 16384, replace=False)
    mo, do = orc.knn6(sp[sub], sc[sub], tp, tc, 0.1)
    assert np.array_equal(m["idx"][sub], mo["idx"]) and np.array_equal(d2[sub].view(np.uint32), do.view(np.uint32))
    pose, recs, _ = c.run(np.eye(4))
    assert [x["n_src"] for x in recs][0] < 300 and recs[-1]["n_src"] > 250000          # coarse-to-fine schedule ran
    ang, tr = pose_error(pose, r["gt"])
    assert ang < 2e-3 and tr < 5e-3


def test_fullsize_incremental_search_bit_identical(gpu_ctx_factory, eth_pair):
    """50 iterations at 370k with and without the verify-and-skip k-NN: identical poses, bit for bit, every iteration;
    and the converged iterations really are cheaper."""
    res = []
    for inc in (1, 0):
        c = gpu_ctx_factory()
        c.params.max_distance = 10.0; c.params.metric = 1; c.params.n_iterations = 50; c.params.knn_backend = 1; c.params.knn_incremental = inc
        c.push_params(); c.set_target(eth_pair["tgt_pts"], eth_pair["tgt_nrm"]); c.set_source(eth_pair["src_pts"], eth_pair["src_nrm"])
        c.run(np.eye(4))
        pose, recs, _ = c.run(np.eye(4))
        res.append((recs, c.timing()["match_ms"]))
    for a, b in zip(res[0][0], res[1][0]):
        assert a["n_valid"] == b["n_valid"] and np.array_equal(a["pose"], b["pose"])
    assert res[0][1] < res[1][1]


def test_fullsize_incremental_run_is_bit_identical_to_always_walk(gpu_ctx_factory, eth_pair):
    """configs[1] at full size, 50 iterations: verify-and-skip (both tiers), shared walks and the spread start must not change a single
    match -- every iteration's pose and valid count equal the run in which every query walks the tree in every iteration."""
    out = []
    for inc in (1, 0):
        c = gpu_ctx_factory()
        c.params.max_distance = 10.0; c.params.metric = 1; c.params.n_iterations = 50; c.params.knn_backend = 1; c.params.knn_incremental = inc
        c.push_params()
        c.set_target(eth_pair["tgt_pts"], eth_pair["tgt_nrm"]); c.set_source(eth_pair["src_pts"], eth_pair["src_nrm"])
        _, recs, rc = c.run(np.eye(4))
        assert rc == 0 and len(recs) == 50
        out.append(recs)
    for k, (a, b) in enumerate(zip(*out)):
        assert a["n_valid"] == b["n_valid"] and np.array_equal(a["pose"], b["pose"]), k


def test_deep_tree_over_524288_targets(gpu_ctx_factory, orc):
    """A target of 603 120 points needs 9 levels of 4-wide nodes: the 64-bit pending mask and the <3, true> instance of the fused
    matcher.  Matches bit-exact against the oracle's kd-tree, teacher-forced iterations within 1e-5, and the incremental run
    bit-identical to the always-walk run."""
    from conftest import pose_error
    from icp_amd import synth
    p = synth.eth_like_pair(1, n_tilt=560, n_beam=1077)
    assert len(p["tgt_pts"]) == 603120
    src, srn = p["src_pts"][::9], p["src_nrm"][::9]                     # 67 014 queries keep the oracle quick
    c = gpu_ctx_factory()
    c.params.max_distance = 10.0; c.params.metric = 1; c.params.n_iterations = 30; c.params.knn_backend = 1; c.push_params()
    c.set_target(p["tgt_pts"], p["tgt_nrm"]); c.set_source(src, srn)
    kd = orc.KdTree(p["tgt_pts"])
    m, d2 = c.match(np.eye(4))
    mo, do = kd.query(orc.transform_points(src, np.eye(4)), 10.0)
    assert np.array_equal(m["idx"], mo["idx"]) and np.array_equal(d2.view(np.uint32), do.view(np.uint32))
    prm = orc.make_params(metric=1, n_iterations=1, max_distance=10.0, solver_mode=1, knn_kdtree=1); prm.kdtree = kd.h
    pose = np.eye(4, dtype=f32)
    for k in range(3):
        po, mo, nvo, _, _ = orc.iterate(prm, src, srn, None, p["tgt_pts"], p["tgt_nrm"], None, pose)
        pg, st = c.iterate(pose)
        assert st["n_valid"] == nvo
        ang, tr = pose_error(pg, po)
        assert ang < 1e-5 and tr < 1e-5, (k, ang, tr)
        pose = po
    out = []
    for inc in (1, 0):
        c.params.knn_incremental = inc; c.push_params()
        _, recs, rc = c.run(np.eye(4))
        assert rc == 0
        out.append(recs)
    for k, (a, b) in enumerate(zip(*out)):
        assert a["n_valid"] == b["n_valid"] and np.array_equal(a["pose"], b["pose"]), k
