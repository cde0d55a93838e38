"""CPU tests of the ORACLE (oracle/icp_oracle.cpp) -- the checker every GPU parity test relies on.

PARITY UNPINNED: the reference ships no tests / golden vectors for this path.  What pins the
oracle here: (1) independent numpy / scipy restatements of each stage, written from the reference source
(file:line cited per test), (2) the committed regression vectors in tests/golden/ (make_golden.py), (3) the only
numeric anchor the reference holds -- the 4 hand-picked bunny ground-truth correspondences (main.cpp:110-120).
"""
import numpy as np
import pytest
from scipy.spatial import cKDTree

f32 = np.float32


def rand_pose(rng, ang=0.2, tr=0.5):
    from icp_amd import synth
    return synth.make_pose(rng.uniform(-ang, ang, 3), rng.uniform(-tr, tr, 3)).astype(np.float32)


def np_knn3(q, t, max_dist):
    """NearestNeighbor.h:81-97 with squared distances (:181-185), FLANN L2 summation order, numpy float32."""
    idx = np.empty(len(q), np.int32); d2 = np.empty(len(q), np.float32)
    for i in range(len(q)):
        dx = q[i, 0] - t[:, 0]; dy = q[i, 1] - t[:, 1]; dz = q[i, 2] - t[:, 2]
        d = (dx * dx + dy * dy) + dz * dz
        d = np.where(np.isnan(d), np.inf, d)
        j = int(np.argmin(d))                       # first minimum
        if d[j] < np.finfo(np.float32).max:
            idx[i], d2[i] = j, d[j]
        else:
            idx[i], d2[i] = -1, np.finfo(np.float32).max
    keep = d2 <= max_dist
    return np.where(keep, idx, -1).astype(np.int32), d2


def test_transform_points_order(orc):
    """utils.h:106-118: ((R0*x + R1*y) + R2*z) + t in fp32, one rounding per op."""
    rng = np.random.default_rng(1)
    pts = rng.uniform(-8, 8, (500, 3)).astype(f32)
    T = rand_pose(rng)
    out = orc.transform_points(pts, T)
    exp = np.empty_like(pts)
    for r in range(3):
        exp[:, r] = ((T[r, 0] * pts[:, 0] + T[r, 1] * pts[:, 1]) + T[r, 2] * pts[:, 2]) + T[r, 3]
    assert np.array_equal(out.view(np.uint32), exp.view(np.uint32))


def test_transform_normals_is_inverse_transpose(orc):
    """utils.h:122-133: (R^-1)^T n; for a rigid R equal to R n up to fp32 rounding."""
    rng = np.random.default_rng(2)
    n = rng.normal(size=(200, 3)).astype(f32)
    T = rand_pose(rng)
    out = orc.transform_normals(n, T)
    N = np.linalg.inv(T[:3, :3].astype(np.float64)).T
    assert np.allclose(out, n.astype(np.float64) @ N.T, atol=2e-6)
    assert np.allclose(orc.normal_matrix(T), N, atol=1e-7)
    # non-rigid (scaled) pose: must still be the inverse transpose, not R
    S = T.copy(); S[:3, :3] *= f32(1.5)
    assert np.allclose(orc.normal_matrix(S), np.linalg.inv(S[:3, :3].astype(np.float64)).T, atol=1e-7)


def test_knn3_matches_numpy_restatement(orc):
    rng = np.random.default_rng(3)
    q = rng.uniform(-2, 2, (300, 3)).astype(f32)
    t = rng.uniform(-2, 2, (1000, 3)).astype(f32)
    t[10] = np.nan; t[20] = -np.inf; t[31] = t[30]; t[500] = t[30]      # invalid targets and exact duplicates
    q[0] = t[30]; q[1] = np.nan; q[2, 0] = np.inf
    m, d2 = orc.knn3(q, t, 0.05)
    ei, ed = np_knn3(q, t, 0.05)
    assert np.array_equal(m["idx"], ei)
    assert np.array_equal(d2.view(np.uint32), ed.view(np.uint32))
    assert m["idx"][0] == 30                                            # tie -> lowest index
    assert m["idx"][1] == -1 and m["idx"][2] == -1
    assert np.all(m["weight"][m["idx"] >= 0] == 1.0) and np.all(m["weight"][m["idx"] < 0] == 0.0)


def test_knn3_agrees_with_scipy_ckdtree(orc):
    """Independent exact NN in float64: same index wherever the NN is unambiguous beyond fp32 rounding."""
    rng = np.random.default_rng(4)
    q = rng.uniform(-8, 8, (2000, 3)).astype(f32)
    t = rng.uniform(-8, 8, (5000, 3)).astype(f32)
    m, d2 = orc.knn3(q, t, 1e9)
    dd, ii = cKDTree(t.astype(np.float64)).query(q.astype(np.float64), k=2)
    clear = (dd[:, 1] - dd[:, 0]) > 1e-4
    assert clear.sum() > 1900
    assert np.array_equal(m["idx"][clear], ii[clear, 0])
    assert np.allclose(d2, dd[:, 0] ** 2, rtol=1e-5)


def test_kdtree_is_bit_identical_to_bruteforce(orc):
    rng = np.random.default_rng(5)
    t = rng.uniform(-3, 3, (20000, 3)).astype(f32)
    t[5] = np.nan; t[77] = -np.inf; t[100] = t[99]; t[19999] = t[0]
    q = np.concatenate([rng.uniform(-4, 4, (3000, 3)).astype(f32), t[95:105], t[:3]])
    a, da = orc.knn3(q, t, 0.3)
    b, db = orc.KdTree(t).query(q, 0.3)
    assert np.array_equal(a["idx"], b["idx"])
    assert np.array_equal(da.view(np.uint32), db.view(np.uint32))


def test_threshold_is_squared_and_inclusive(orc):
    """NearestNeighbor.h:182 `*distances[i] <= m_maxDistance` on SQUARED distances."""
    t = np.array([[0, 0, 0]], f32)
    q = np.array([[0.5, 0, 0], [0.5000001, 0, 0]], f32)
    m, d2 = orc.knn3(q, t, float(f32(0.25)))
    assert m["idx"].tolist() == [0, -1]


def test_knn6_matches_numpy_restatement(orc):
    """NearestNeighbor.h:209-303: feature = [xyz, rgb/255], alpha ignored, sequential 6-term L2."""
    rng = np.random.default_rng(6)
    q = rng.uniform(-1, 1, (200, 3)).astype(f32); t = rng.uniform(-1, 1, (700, 3)).astype(f32)
    qc = rng.integers(0, 256, (200, 4)).astype(np.uint8); tc = rng.integers(0, 256, (700, 4)).astype(np.uint8)
    m, d2 = orc.knn6(q, qc, t, tc, 0.5)
    cn = f32(1) / f32(255)
    qf = (f32(1) * cn) * qc[:, :3].astype(f32); tf = (f32(1) * cn) * tc[:, :3].astype(f32)
    assert np.array_equal(orc.color_features(qc).view(np.uint32), qf.view(np.uint32))
    for i in range(len(q)):
        d = [q[i, k] - t[:, k] for k in range(3)] + [qf[i, k] - tf[:, k] for k in range(3)]
        s = d[0] * d[0] + d[1] * d[1]
        for k in range(2, 6):
            s = s + d[k] * d[k]
        j = int(np.argmin(s))
        assert d2[i].view(np.uint32) == s[j].view(np.uint32)
        assert m["idx"][i] == (j if s[j] <= f32(0.5) else -1)


def py_projective(q, tgt, w, h, K, max_dist, win=12):
    """Literal restatement of NearestNeighbor.h:351-421 with Python ints emulating the unsigned loop."""
    fx, fy, mx, my = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    out = []
    for p in q:
        if p[0] == -np.inf:
            out.append((0, 0.0)); continue
        with np.errstate(all="ignore"):
            uf = np.round(f32(f32(f32(p[0] * fx) / p[2]) + mx)); vf = np.round(f32(f32(f32(p[1] * fy) / p[2]) + my))
            # np.round is half-to-even; std::round is half-away-from-zero
            uf = f32(np.floor(np.abs(f32(f32(f32(p[0] * fx) / p[2]) + mx)) + f32(0.5)) * np.sign(f32(f32(f32(p[0] * fx) / p[2]) + mx)))
            vf = f32(np.floor(np.abs(f32(f32(f32(p[1] * fy) / p[2]) + my)) + f32(0.5)) * np.sign(f32(f32(f32(p[1] * fy) / p[2]) + my)))
        best, bi = np.finfo(np.float32).max, -1
        if np.isfinite(uf) and np.isfinite(vf) and uf >= 0 and vf >= 0 and uf < 2 ** 31 and vf < 2 ** 31:
            U, V = int(uf), int(vf)
            v = (V - win) % 2 ** 32
            while v < h and v <= V + win:
                u = (U - win) % 2 ** 32
                while u < w and u <= U + win:
                    j = w * v + u
                    if tgt[j, 0] != -np.inf:
                        d = p - tgt[j]
                        dist = f32(d[0] * d[0]) + (f32(d[1] * d[1]) + f32(d[2] * d[2]))
                        if best > dist:
                            best, bi = dist, j
                    u += 1
                v += 1
        out.append((bi, 1.0) if best <= max_dist else (-1, 0.0))
    return out


def test_projective_matches_literal_restatement(orc):
    from icp_amd import synth
    W, H = 64, 48
    K = np.array([[50.0, 0, 31.5], [0, 50.0, 23.5], [0, 0, 1]], f32)
    tp, tn, tc = synth.depth_frame(synth.camera_pose(0), K, W, H, seed=11, hole_frac=0.1)
    sp, sn, sc = synth.depth_frame(synth.camera_pose(1), K, W, H, seed=12, hole_frac=0.1)
    q = sp.copy()
    q[5] = [-np.inf, 0.1, 1.0]                        # x == MINF  -> Match{0, 0.f}
    q[6] = [np.nan, 0.1, 1.0]
    q[7] = [0.1, 0.1, -1.0]                           # behind the camera -> negative pixel
    q[8] = [0.1, 0.1, 0.0]                            # division by zero
    m, d2 = orc.projective(q, tp, W, H, K, 0.05)
    exp = py_projective(q, tp, W, H, K, f32(0.05))
    assert m["idx"].tolist() == [e[0] for e in exp]
    assert m["weight"].tolist() == [e[1] for e in exp]
    assert (m["idx"][5], m["weight"][5]) == (0, 0.0)
    assert (m["idx"] >= 0).sum() > 200
    # border quirk (:385-386): queries projecting to u < 12 or v < 12 never match
    u = np.round(q[:, 0] * K[0, 0] / q[:, 2] + K[0, 2]); v = np.round(q[:, 1] * K[1, 1] / q[:, 2] + K[1, 2])
    border = np.isfinite(u) & np.isfinite(v) & ((u < 12) | (v < 12)) & (q[:, 0] != -np.inf)
    assert border.sum() > 50 and np.all(m["idx"][border] == -1)


def test_weights_match_numpy_restatement(orc):
    """weighting.h:16-30,44-90: double subtraction, uint8 wrap-around of the colour difference."""
    rng = np.random.default_rng(7)
    n, mtg = 400, 300
    sp = rng.uniform(-1, 1, (n, 3)).astype(f32); tp = rng.uniform(-1, 1, (mtg, 3)).astype(f32)
    sn = rng.normal(size=(n, 3)).astype(f32); tn = rng.normal(size=(mtg, 3)).astype(f32)
    sc = rng.integers(0, 256, (n, 4)).astype(np.uint8); tc = rng.integers(0, 256, (mtg, 4)).astype(np.uint8)
    sp[3] = np.nan; sn[4] = np.inf; tn[0] = np.nan
    matches = np.zeros(n, orc.MATCH_DTYPE)
    matches["idx"] = rng.integers(-1, mtg, n); matches["weight"] = np.where(matches["idx"] >= 0, 1.0, 0.0)
    matches["idx"][4] = 5; matches["idx"][9] = 0
    maxd = f32(0.7)
    for method in (0, 1, 2, 3):
        out = orc.apply_weights(method, float(maxd), sp, tp, sn, tn, sc, tc, matches)
        exp = matches["weight"].copy()
        if method != 0:
            for i in range(n):
                j = matches["idx"][i]
                if j < 0:
                    continue
                w = f32(0)
                if method in (1, 3) and np.isfinite(sp[i]).all() and np.isfinite(tp[j]).all():
                    d = sp[i] - tp[j]
                    w = f32(w + f32(1.0 - float(f32(f32(f32(d[0] * d[0]) + f32(d[1] * d[1])) + f32(d[2] * d[2])) / maxd)))
                if method == 2 and np.isfinite(sn[i]).all() and np.isfinite(tn[j]).all():
                    w = f32(w + (f32(sn[i, 0] * tn[j, 0]) + (f32(sn[i, 1] * tn[j, 1]) + f32(sn[i, 2] * tn[j, 2]))))
                if method == 3:
                    e = (sc[i].astype(np.int32) - tc[j].astype(np.int32)) % 256
                    w = f32(w * f32(1.0 - float(f32(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]) / f32(195075))))
                exp[i] = w
        assert np.array_equal(out["weight"].view(np.uint32), exp.view(np.uint32)), method
        assert np.array_equal(out["idx"], matches["idx"])


def test_prune_60_degrees(orc):
    """ICPOptimizer.h:157-174: reject iff acos(cos) > 60 deg; NaN angle => kept; weight untouched."""
    ang = np.deg2rad(np.array([0, 30, 59.9, 60.1, 90, 180, 59.9999, 60.0001]))
    tn = np.tile(np.array([[0, 0, 1]], f32), (len(ang) + 2, 1))
    sn = np.stack([np.sin(ang), 0 * ang, np.cos(ang)], 1).astype(f32) * f32(2.5)      # un-normalised on purpose
    sn = np.concatenate([sn, np.array([[np.nan, 0, 1], [0, 0, 0]], f32)])
    m = np.zeros(len(sn), orc.MATCH_DTYPE); m["idx"] = np.arange(len(sn)); m["weight"] = 0.5
    out = orc.prune(sn, tn, m)
    assert out["idx"].tolist() == [0, 1, 2, -1, -1, -1, 6, -1, 8, 9]
    assert np.all(out["weight"] == 0.5)
    # the predicate is monotone in the cosine around 0.5 (the device derives its constant from this)
    cs = np.arange(np.float32(0.49).view(np.uint32), np.float32(0.51).view(np.uint32), 997, dtype=np.uint32).view(np.float32)
    pred = np.array([orc.prune_predicate(float(c)) for c in cs])
    assert np.all(pred[:-1] >= pred[1:]) and pred[0] and not pred[-1]


def _compacted(orc, rng, n=600):
    s = rng.uniform(-3, 3, (n, 3)).astype(f32)
    T = rand_pose(rng, 0.05, 0.05)
    d = orc.transform_points(s, T) + rng.normal(0, 0.002, (n, 3)).astype(f32)
    nt = rng.normal(size=(n, 3)); nt = (nt / np.linalg.norm(nt, axis=1, keepdims=True)).astype(f32)
    ns = (nt + rng.normal(0, 0.05, (n, 3))).astype(f32)
    w = rng.uniform(0.2, 1.0, n).astype(f32)
    return s, d.astype(f32), ns, nt, w, T


def test_p2plane_solve_vs_numpy_lstsq(orc):
    """ICPOptimizer.h:676-782: rows, lambda = 1 / 0.1, x = argmin |Ax-b|, R = Rx Ry Rz."""
    rng = np.random.default_rng(8)
    s, d, ns, nt, w, T = _compacted(orc, rng)
    S, D, N, W = [a.astype(np.float64) for a in (s, d, nt, w)]
    rows, rhs = [], []
    for i in range(len(s)):
        rows.append(np.concatenate([np.cross(S[i], N[i]), N[i]]) * W[i]); rhs.append((N[i] @ D[i] - N[i] @ S[i]) * W[i])
        for k, r in enumerate(([0, S[i, 2], -S[i, 1], 1, 0, 0], [-S[i, 2], 0, S[i, 0], 0, 1, 0], [S[i, 1], -S[i, 0], 0, 0, 0, 1])):
            rows.append(np.array(r, float) * 0.1 * W[i]); rhs.append((D[i, k] - S[i, k]) * 0.1 * W[i])
    x = np.linalg.lstsq(np.array(rows), np.array(rhs), rcond=None)[0]
    for mode, tol in ((1, 2e-6), (0, 2e-5)):
        pose, xo = orc.solve_p2plane(s, d, nt, w, mode)
        assert np.allclose(xo, x, atol=tol), mode
        from icp_amd import synth
        assert np.allclose(pose[:3, :3], synth.rot_xyz(*x[:3]), atol=tol) and np.allclose(pose[:3, 3], x[3:], atol=tol)


def test_p2p_solve_vs_numpy_kabsch(orc):
    """ProcrustesAligner.h:6-72: unweighted means, weights on the source side only, det fix."""
    rng = np.random.default_rng(9)
    s, d, ns, nt, w, T = _compacted(orc, rng)
    S, D, W = s.astype(np.float64), d.astype(np.float64), w.astype(np.float64)
    sm, dm = S.mean(0), D.mean(0)
    A = (D - dm).T @ (W[:, None] * (S - sm))
    U, _, Vt = np.linalg.svd(A)
    R = U @ np.diag([1, 1, np.linalg.det(U @ Vt)]) @ Vt
    t = R @ (dm - sm) - R @ dm + dm
    for mode, tol in ((1, 2e-6), (0, 2e-5)):
        pose = orc.solve_p2p(s, d, w, mode)
        assert np.allclose(pose[:3, :3], R, atol=tol) and np.allclose(pose[:3, 3], t, atol=tol), mode
    # reflection case: mirrored target forces det(UV^T) = -1
    d2 = d.copy(); d2[:, 2] *= -1
    pose = orc.solve_p2p(s, d2, w, 1)
    assert abs(np.linalg.det(pose[:3, :3].astype(np.float64)) - 1) < 1e-5


def test_symmetric_solve_vs_numpy(orc):
    """ICPOptimizer.h:784-898: centred rows with n_t + n_s, (A^T A + 1e-8 I) x = A^T b, Rodrigues composition."""
    rng = np.random.default_rng(10)
    s, d, ns, nt, w, T = _compacted(orc, rng)
    S, D, NS, NT, W = [a.astype(np.float64) for a in (s, d, ns, nt, w)]
    sm, dm = S.mean(0), D.mean(0)
    rows, rhs = [], []
    for i in range(len(s)):
        sc, dc, n = S[i] - sm, D[i] - dm, NT[i] + NS[i]
        rows.append(np.concatenate([np.cross(sc + dc, n), n]) * W[i]); rhs.append(((dc - sc) @ n) * W[i])
        for k, r in enumerate(([0, sc[2], -sc[1], 1, 0, 0], [-sc[2], 0, sc[0], 0, 1, 0], [sc[1], -sc[0], 0, 0, 0, 1])):
            rows.append(np.array(r, float) * 0.1 * W[i]); rhs.append((dc[k] - sc[k]) * 0.1 * W[i])
    A, b = np.array(rows), np.array(rhs)
    x = np.linalg.solve(A.T @ A + 1e-8 * np.eye(6), A.T @ b)
    at, tt = x[:3], x[3:]
    tan = np.linalg.norm(at); a = at / tan; sin = tan / np.sqrt(1 + tan * tan); cos = sin / tan
    Km = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    Rod = np.eye(4); Rod[:3, :3] = np.eye(3) + sin * Km + (1 - cos) * Km @ Km

    def Tr(v):
        M = np.eye(4); M[:3, 3] = v; return M
    exp = Tr(dm) @ Rod @ Tr(tt * cos) @ Rod @ Tr(-sm)
    for mode, tol in ((1, 3e-6), (0, 5e-5)):
        pose, xo = orc.solve_symmetric(s, d, ns, nt, w, mode)
        assert np.allclose(xo, x, atol=tol), mode
        assert np.allclose(pose, exp, atol=tol), mode


def test_compaction_skips_nonfinite_points_only(orc):
    """ICPOptimizer.h:594-610: src/tgt POINTS must be finite; normals are not checked on the linear path."""
    sp = np.array([[0, 0, 0], [np.nan, 0, 0], [1, 1, 1], [2, 2, 2]], f32); sn = np.array([[0, 0, 1]] * 4, f32)
    tp = np.array([[0, 0, 0.1], [np.inf, 0, 0], [1, 1, 1.1]], f32); tn = np.array([[0, 0, 1], [0, 0, 1], [np.nan, 0, 1]], f32)
    m = np.zeros(4, orc.MATCH_DTYPE); m["idx"] = [0, 0, 1, 2]; m["weight"] = [1, 1, 1, 0.5]
    cs, cd, cw, cnt, cns = orc.compact(sp, sn, tp, tn, m)
    assert len(cs) == 2 and np.array_equal(cs, sp[[0, 3]]) and np.array_equal(cd, tp[[0, 2]])
    assert cw.tolist() == [1.0, 0.5] and np.isnan(cnt[1, 0])


def test_coarse_resolution(orc):
    """PointCloud.h:325-343."""
    pts = np.arange(30, dtype=f32).reshape(10, 3); nrm = np.ones((10, 3), f32)
    pts[4] = np.nan; nrm[6] = np.inf
    op, on, oc, oi = orc.coarse(pts, nrm, None, 2)
    assert oi.tolist() == [0, 2, 8]
    op, on, oc, oi = orc.coarse(pts, nrm, None, 1)
    assert oi.tolist() == [0, 1, 2, 3, 5, 7, 8, 9]


def test_rmse(orc):
    """ConvergenceMeasure.h:50-66."""
    rng = np.random.default_rng(11)
    s = rng.uniform(-1, 1, (100, 3)).astype(f32); T = rand_pose(rng)
    r = orc.transform_points(s, T) + f32(0.01)
    r[3] = np.nan
    v = orc.rmse(s, r, T)
    assert abs(v - np.sqrt(3) * 0.01) < 1e-5


def test_random_selection_contract(orc, bunny):
    """RANDOM_SAMPLING (selection.h:88-106) with the reproducible hash predicate: library == oracle, Bernoulli(p) statistics,
    resampled every iteration, and ICP still converges on a 50 % sample (Data/bunny_experiments.csv rows bunny103-105)."""
    from icp_amd import binding
    idx = np.arange(20000)
    h = np.array([orc.select_hash(7, 3, int(i)) for i in idx[:2000]], np.uint64)
    assert all(binding.select_hash(7, 3, int(i)) == int(h[i]) for i in range(0, 2000, 97))
    frac = (h < 0.3 * 2 ** 32).mean()
    assert abs(frac - 0.3) < 0.04
    assert orc.select_hash(7, 3, 5) != orc.select_hash(7, 4, 5) and orc.select_hash(7, 3, 5) != orc.select_hash(8, 3, 5)
    sp, sn, tp, tn = bunny["src_pts"], bunny["src_nrm"], bunny["tgt_pts"], bunny["tgt_nrm"]
    prm = orc.make_params(metric=1, n_iterations=20, max_distance=0.0003, selection=1, selection_proba=0.5, selection_seed=11)
    pose, recs = orc.estimate_pose(prm, sp, sn, None, tp, tn, None, np.eye(4))
    ns = [r["n_src"] for r in recs]
    assert len(set(ns)) > 5 and all(400 < v < 660 for v in ns)            # a fresh ~50 % sample per iteration
    gs = sp[bunny["gt_src_idx"]]; gt = tp[bunny["gt_tgt_idx"]]
    assert orc.rmse(gs, gt, pose) < 2e-3
    prm1 = orc.make_params(metric=1, n_iterations=5, max_distance=0.0003, selection=1, selection_proba=1.0)
    prm0 = orc.make_params(metric=1, n_iterations=5, max_distance=0.0003)
    assert np.array_equal(orc.estimate_pose(prm1, sp, sn, None, tp, tn, None, np.eye(4))[0], orc.estimate_pose(prm0, sp, sn, None, tp, tn, None, np.eye(4))[0])


def test_backproject_depth(orc):
    """PointCloud(depthMap, ...) (PointCloud.h:78-165): numpy restatement incl. MINF holes, border normals and the colour-index quirk."""
    rng = np.random.default_rng(13)
    W, H = 40, 30
    K = np.array([[52.5, 0, 19.5], [0, 52.5, 14.5], [0, 0, 1]], f32)
    depth = (1.5 + 0.3 * np.sin(np.arange(W)[None, :] * 0.2) + 0.2 * np.cos(np.arange(H)[:, None] * 0.3)).astype(f32)
    depth[rng.random((H, W)) < 0.05] = -np.inf
    depth[10, 10:14] += f32(0.5)                                        # a depth jump: |du| > maxDistance/2 -> invalid normal
    rgbx = rng.integers(0, 256, (H * W, 4)).astype(np.uint8)
    xyz, nrm, rgba, valid = orc.backproject(depth, rgbx, K)
    u, v = np.meshgrid(np.arange(W, dtype=f32), np.arange(H, dtype=f32))
    d = depth
    with np.errstate(invalid="ignore"):
        ex = np.stack([((u - K[0, 2]) / K[0, 0] * d).astype(f32), ((v - K[1, 2]) / K[1, 1] * d).astype(f32), d], -1).reshape(-1, 3)
    hole = (d == -np.inf).reshape(-1)
    assert np.array_equal(xyz[~hole].view(np.uint32), ex[~hole].astype(f32).view(np.uint32)) and np.all(xyz[hole] == -np.inf)
    n2 = nrm.reshape(H, W, 3)
    assert np.all(n2[0] == -np.inf) and np.all(n2[-1] == -np.inf) and np.all(n2[:, 0] == -np.inf) and np.all(n2[:, -1] == -np.inf)
    with np.errstate(invalid="ignore"):
        du = f32(0.5) * (d[1:-1, 2:] - d[1:-1, :-2]); dv = f32(0.5) * (d[2:, 1:-1] - d[:-2, 1:-1])
    okn = np.isfinite(du) & np.isfinite(dv) & ~(np.abs(du) > 0.05) & ~(np.abs(dv) > 0.05)
    inner = n2[1:-1, 1:-1]
    assert np.all(inner[~okn] == -np.inf)
    expn = np.stack([-du, -dv, np.ones_like(du)], -1); expn = expn / np.sqrt(du * du + (dv * dv + f32(1)))[..., None]
    assert np.allclose(inner[okn], expn[okn], atol=1e-7) and np.allclose(np.linalg.norm(inner[okn], axis=1), 1, atol=1e-6)
    assert (~okn).sum() > 20 and not okn[9, 9]                          # pixel (10, 10) sits on the depth jump: |du| = 0.25 > 0.05
    assert np.array_equal(valid, np.isfinite(xyz).all(1) & np.isfinite(nrm).all(1))
    flat = rgbx.reshape(-1)
    assert np.array_equal(rgba[5], flat[5:9]) and np.array_equal(rgba[:-1, 0], flat[:H * W - 1])          # PointCloud.h:156-157 quirk
    assert np.array_equal(orc.backproject(depth, rgbx, K, fix_color_index=True)[2], rgbx)
    # non-identity extrinsics: points go through the inverse, normals stay in the camera frame (:128-129)
    from icp_amd import synth
    E = synth.make_pose((0.1, -0.2, 0.3), (0.5, -0.1, 0.2))
    x2, n_2, _, _ = orc.backproject(depth, None, K, extrinsics=E)
    Ei = np.linalg.inv(E)
    assert np.allclose(x2[~hole], ex[~hole].astype(np.float64) @ Ei[:3, :3].T + Ei[:3, 3], atol=2e-6) and np.array_equal(n_2, nrm)


def test_benchmark_error(orc):
    """ConvergenceMeasure.h:104-151: mean |T s - r| / |T s - centroid(T s)|."""
    rng = np.random.default_rng(12)
    s = rng.uniform(-3, 3, (500, 3)).astype(f32); T = rand_pose(rng)
    r = (s + rng.normal(0, 0.01, s.shape)).astype(f32)
    t = orc.transform_points(s, T).astype(np.float64)
    exp = np.mean(np.linalg.norm(t - r, axis=1) / np.linalg.norm(t - t.mean(0), axis=1))
    assert abs(orc.benchmark_error(s, r, T) - exp) < 1e-6 * exp


def test_golden_regression(orc, bunny, bunny_oracle):
    """The committed vectors (tests/golden/bunny_oracle.npz) are reproduced by the oracle as built here."""
    sp, sn, sc, tp, tn, tc = [bunny[k] for k in ("src_pts", "src_nrm", "src_rgba", "tgt_pts", "tgt_nrm", "tgt_rgba")]
    m, d2 = orc.knn3(sp, tp, 0.0003)
    assert np.array_equal(m["idx"], bunny_oracle["knn3_identity_idx"])
    assert np.array_equal(d2.view(np.uint32), bunny_oracle["knn3_identity_d2"].view(np.uint32))
    assert int((m["idx"] >= 0).sum()) == 576                          # 576/1054 at thr 0.0003
    for metric in (0, 1, 2):
        for weighting, multires in ((0, 0), (1, 0), (2, 0), (0, 1)):
            for mode in (0, 1):
                prm = orc.make_params(metric=metric, weighting=weighting, multires=multires, n_iterations=20, max_distance=0.0003, solver_mode=mode)
                pose, recs = orc.estimate_pose(prm, sp, sn, sc, tp, tn, tc, np.eye(4))
                key = "m%d_w%d_r%d_mode%d" % (metric, weighting, multires, mode)
                assert np.array_equal(np.array([r["n_valid"] for r in recs]), bunny_oracle[key + "_nvalid"]), key
                assert np.array_equal(np.array([r["n_src"] for r in recs]), bunny_oracle[key + "_nsrc"]), key
                assert np.allclose(np.stack([r["pose"] for r in recs]), bunny_oracle[key + "_poses"], atol=1e-7), key


def test_faithful_and_exact_modes_agree_on_bunny(bunny_oracle):
    """fp32-sequential sums (reference-shaped) vs fp64 sums: the reference's own rounding noise stays below the 1e-5
    parity tolerance at the converged pose.  (Mid-trajectory a single flipped match can separate the two runs by
    ~2e-4 for one iteration -- symmetric ICP, iteration 1 -- before both fall into the same fixed point.)"""
    for metric in (0, 1, 2):
        a = bunny_oracle["m%d_w0_r0_mode0_poses" % metric]; b = bunny_oracle["m%d_w0_r0_mode1_poses" % metric]
        assert np.abs(a[-1] - b[-1]).max() < 1e-5
        assert np.abs(a - b).max() < 1e-3


def test_bunny_ground_truth_anchor(orc, bunny, bunny_oracle):
    """main.cpp:110-120: source 215,424,640,1023 <-> target 294,258,1238,1310.  The only numbers the reference
    holds for this path: linear ICP must pull these pairs together (p2plane/symmetric below 1 mm, p2p below 5 mm)."""
    gs = bunny["src_pts"][bunny["gt_src_idx"]]; gt = bunny["tgt_pts"][bunny["gt_tgt_idx"]]
    initial = orc.rmse(gs, gt, np.eye(4))
    assert initial > 0.02
    bounds = {0: 5e-3, 1: 1e-3, 2: 1e-3}
    for metric in (0, 1, 2):
        final = orc.rmse(gs, gt, bunny_oracle["m%d_w0_r0_mode0_poses" % metric][-1])
        assert final < bounds[metric], (metric, final)
        assert final < initial / 5


def test_multires_schedule_matches_library(orc, bunny):
    """icp_schedule (host logic of libicp_hip.so) == the oracle's loop control (ICPOptimizer.h:503-516,634-655)."""
    from icp_amd import binding
    sp, sn, tp, tn = bunny["src_pts"], bunny["src_nrm"], bunny["tgt_pts"], bunny["tgt_nrm"]
    for n_iter in (1, 3, 20):
        prm = orc.make_params(metric=1, multires=1, n_iterations=n_iter, max_distance=0.0003)
        _, recs = orc.estimate_pose(prm, sp, sn, None, tp, tn, None, np.eye(4))
        p = binding.default_params(); p.multires = 1; p.n_iterations = n_iter
        fac = binding.schedule(p, len(sp))
        assert len(fac) == len(recs)
        assert [len(range(0, len(sp), f)) for f in fac] == [r["n_src"] for r in recs]
    p = binding.default_params(); p.multires = 1; p.n_iterations = 5
    assert binding.schedule(p, 370488) == [2048, 1024, 512, 256, 128, 64, 32, 16, 8, 4, 2, 1]
    p.multires = 0
    assert binding.schedule(p, 370488) == [0] * 5
    p.multires = 1; p.n_iterations = 0
    with pytest.raises(binding.IcpError):
        binding.schedule(p, 1000)
