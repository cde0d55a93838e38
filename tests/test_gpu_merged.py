"""The three forms of the point-to-plane loop -- "loop": k_icp_loop, all iterations of a level in ONE launch beside its reducer kernel
(dev_persist.hpp; ICP_HIP_PERSIST=1); "merged": one launch per iteration with the reducer of iteration i - 1 riding in front of the
matcher of iteration i (dev_solve.hpp "the ring form"; the default); "separate": k_reduce_solve as a launch of its own (ICP_HIP_MERGE=0).
In all of them the pose and the sums travel through self-validating granules or kernel boundaries, the fold order and the solve are
the same operations, so every iteration's pose and valid count must be equal bit for bit -- any stale or torn hand-over shows up as a
different pose.  Plus the routes out of the first two (rank-deficient system -> repeated with the separate launches; an iteration
without correspondences) and the re-arming of k_reduce_solve's own hand-over slots."""
import ctypes as C
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
f32 = np.float32
LBVH = 1


FORMS = ("loop", "hybrid", "merged", "separate")      # hybrid: merged launches for the first 7 iterations, k_icp_loop from there (ICP_HIP_LOOP_FROM)


def make_ctx(factory, form, **params):
    if form is True:
        form = "merged"
    elif form is False:
        form = "separate"
    env = {"ICP_HIP_MERGE": "0" if form == "separate" else "1", "ICP_HIP_PERSIST": "1" if form in ("loop", "hybrid") else "0",
           "ICP_HIP_LOOP_FROM": "7" if form == "hybrid" else "0"}                                              # read once, at icp_ctx_create
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        c = factory()
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v
    c.params.knn_backend = LBVH; c.params.metric = 1
    for k, v in params.items():
        setattr(c.params, k, v)
    c.push_params()
    return c


def counters(c):
    a, b = C.c_int32(0), C.c_int32(0)
    assert c.lib.icp_debug_counters(c.h, C.byref(a), C.byref(b)) == 0
    return a.value, b.value


def assert_same_run(ra, rb):
    assert len(ra) == len(rb)
    for k, (a, b) in enumerate(zip(ra, rb)):
        assert a["n_valid"] == b["n_valid"] and a["status"] == b["status"] and a["n_src"] == b["n_src"], k
        assert np.array_equal(a["pose"], b["pose"]), k


def test_merged_loop_fullsize_is_bit_identical_to_separate_launches(gpu_ctx_factory):
    from icp_amd import synth
    p = synth.eth_like_pair(0)
    out = []
    for form in FORMS:
        c = make_ctx(gpu_ctx_factory, form, max_distance=10.0, n_iterations=50)
        c.set_target(p["tgt_pts"], p["tgt_nrm"]); c.set_source(p["src_pts"], p["src_nrm"])
        for stage_timing in (0, 1, 7):                          # the event brackets sit between the launches: they must not matter
            c.set_stage_timing(stage_timing)
            pose, recs, rc = c.run(np.eye(4))
            assert rc == 0 and len(recs) == 50
            out.append((pose, recs))
        runs, fallbacks = counters(c)
        assert (runs, fallbacks) == ((0, 0) if form == "separate" else (3, 0))
        if form in ("loop", "hybrid"):                          # its iteration times come from the device's own clock, every iteration
            a, _, _ = c.iteration_times()
            lf = 7 if form == "hybrid" else 0                   # (iterations in front of it run one launch each: event-sampled)
            assert len(a) == 50 and (a[lf:] > 0).all() and a[lf] > a[-1]
        t = c.timing()
        assert t["iterations"] == 50 and t["match_ms"] > 0
        c.close()
    for a in out[1:]:
        assert np.array_equal(out[0][0], a[0])
        assert_same_run(out[0][1], a[1])


@pytest.mark.parametrize("case", ["bunny", "bunny_multires", "colour6d_weights", "two_iterations"])
def test_merged_loop_small_cases(gpu_ctx_factory, bunny, case):
    from icp_amd import synth
    kw = dict(max_distance=0.0003, n_iterations=20)
    tgt = (bunny["tgt_pts"], bunny["tgt_nrm"], None); src = (bunny["src_pts"], bunny["src_nrm"], None)
    if case == "bunny_multires":
        kw.update(multires=1, max_distance=0.001)
    elif case == "two_iterations":
        kw.update(n_iterations=2)
    elif case == "colour6d_weights":
        K = np.array([[131.25, 0, 79.5], [0, 131.25, 59.5], [0, 0, 1]], f32)
        r = synth.rgbd_pair(0, width=160, height=120, K=K)
        tgt = synth.compact_valid(r["tgt_pts"], r["tgt_nrm"], r["tgt_rgba"]); src = synth.compact_valid(r["src_pts"], r["src_nrm"], r["src_rgba"])
        kw.update(max_distance=0.1, color_icp=1, weighting=3, n_iterations=25)
    res = []
    for form in FORMS:
        c = make_ctx(gpu_ctx_factory, form, **kw)
        c.set_target(*tgt); c.set_source(*src)
        pose, recs, rc = c.run(np.eye(4), check=False)
        res.append((rc, pose, recs, counters(c)))
        c.close()
    for r in res[:-1]:
        assert r[0] == res[-1][0]
        assert np.array_equal(r[1], res[-1][1])
        assert_same_run(r[2], res[-1][2])
        assert r[3] == (1, 0)
    assert res[-1][3] == (0, 0)


def test_rank_deficient_system_leaves_the_merged_loop(gpu_ctx_factory, bunny):
    """ONE valid correspondence: four rows, rank <= 4 -- a pivot of the lane-parallel LDL^T fails, the merged chain passes the fault on
    and the run is repeated with the separate launches (whose k_reduce_solve carries the eigen fallback): same result as ICP_HIP_MERGE=0."""
    tp, tn = bunny["tgt_pts"], bunny["tgt_nrm"]
    sp = (tp[:40] + f32(1.0)).copy(); sn = tn[:40].copy()          # 39 sources a metre away from everything ...
    sp[7] = tp[7] + f32(1e-4)                                      # ... and one on the surface
    res = []
    for merge in ("loop", "separate", "merged", "separate"):      # (results 0 and 2 against 1)
        c = make_ctx(gpu_ctx_factory, merge, max_distance=0.0003, n_iterations=6, rejection=0)
        c.set_target(tp, tn); c.set_source(sp, sn)
        pose, recs, rc = c.run(np.eye(4), check=False)
        res.append((rc, pose, recs, counters(c)))
        pose2, recs2, rc2 = c.run(np.eye(4), check=False)          # the context is as good as new afterwards
        assert rc2 == rc and np.array_equal(pose2, pose)
        c.close()
    assert res[0][2][0]["n_valid"] == 1
    for a in (0, 2):
        assert res[a][0] == res[1][0] and np.array_equal(res[a][1], res[1][1])
        assert_same_run(res[a][2], res[1][2])
        assert res[a][3] == (1, 1)                              # one run in that form, one fallback (counted before the second run)


def test_iteration_without_correspondences_in_the_merged_loop(gpu_ctx_factory, bunny):
    """No valid pair at all: every iteration reports ICP_ERR_NO_CORRESPONDENCES and keeps the pose (the reference hangs in ASSERT,
    ICPOptimizer.h:680) -- inside the merged loop, without a fallback."""
    from icp_amd import binding
    res = []
    for merge in ("loop", "separate", "merged"):
        c = make_ctx(gpu_ctx_factory, merge, max_distance=1e-12, n_iterations=4)
        c.set_target(bunny["tgt_pts"], bunny["tgt_nrm"]); c.set_source(bunny["src_pts"] + f32(3.0), bunny["src_nrm"])
        pose, recs, rc = c.run(np.eye(4), check=False)
        assert rc == binding.ERR_NO_CORRESPONDENCES and np.array_equal(pose, np.eye(4, dtype=f32))
        assert all(r["status"] == binding.ERR_NO_CORRESPONDENCES and r["n_valid"] == 0 for r in recs) and len(recs) == 4
        res.append((recs, counters(c)))
        c.close()
    assert_same_run(res[0][0], res[1][0]); assert_same_run(res[2][0], res[1][0])
    assert res[0][1] == (1, 0) and res[2][1] == (1, 0)


def test_stale_total_in_the_handover_slots_is_never_consumed(gpu_ctx_factory, bunny):
    """k_reduce_solve's own hand-over (separate launches): a stale, valid-looking total left in a slot -- what a run cut short between a
    block's publish and block 0's re-arm leaves behind -- must not reach the next call: every entry point re-arms the slots first."""
    c = make_ctx(gpu_ctx_factory, False, max_distance=0.0003, n_iterations=8)
    c.set_target(bunny["tgt_pts"], bunny["tgt_nrm"]); c.set_source(bunny["src_pts"], bunny["src_nrm"])
    pose0, recs0, _ = c.run(np.eye(4))
    m0, sums0, nv0 = c.correspond(np.eye(4))
    c.lib.icp_debug_poison_handover.argtypes = [C.c_void_p, C.c_int32, C.c_double]
    for slot in (0, 7, 33):
        assert c.lib.icp_debug_poison_handover(c.h, slot, 12345.678) == 0
        pose1, recs1, _ = c.run(np.eye(4))
        assert np.array_equal(pose0, pose1)
        assert_same_run(recs0, recs1)
        assert c.lib.icp_debug_poison_handover(c.h, slot, -1.0) == 0
        m1, sums1, nv1 = c.correspond(np.eye(4))
        assert nv1 == nv0 and np.array_equal(sums0, sums1)
    c.close()
