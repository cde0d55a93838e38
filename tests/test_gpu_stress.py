"""Randomised differential tests of the BVH matcher's shortcuts (verify-and-skip tiers, shared walks, spread start) on clouds built to
provoke them: exact duplicates, points on a coarse grid (ties at every level), thin planes, clusters far from the queries, queries
placed exactly between two targets.  Two comparisons per case: the one-launch matches against the oracle's brute-force scan (bit
for bit), and a 25-iteration incremental run against the same run with every query walking the tree in every iteration (poses and
valid counts bit for bit)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
f32 = np.float32
LBVH = 1


def make_cloud(rng, n, kind):
    if kind == 0:                                           # uniform cube
        p = rng.uniform(-1, 1, (n, 3))
    elif kind == 1:                                         # coarse grid: many exact ties
        p = rng.integers(-6, 7, (n, 3)) * 0.125
    elif kind == 2:                                         # thin noisy plane + a far cluster
        p = np.c_[rng.uniform(-1, 1, (n, 2)), rng.normal(0, 1e-3, n)]
        m = max(1, n // 10); p[:m] = rng.normal(0, 0.01, (m, 3)) + np.array([5.0, 5.0, 5.0])
    else:                                                   # exact duplicates of a small set
        base = rng.uniform(-1, 1, (max(1, n // 4), 3)); p = base[rng.integers(0, len(base), n)]
    return p.astype(f32)


def unit_normals(rng, n):
    v = rng.normal(size=(n, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
    return v.astype(f32)


@pytest.mark.parametrize("seed", range(12))
def test_random_clouds_match_and_incremental(gpu_ctx_factory, orc, seed):
    rng = np.random.default_rng(1000 + seed)
    nt = int(rng.integers(1, 30000)); ns = int(rng.integers(1, 30000))
    tgt = make_cloud(rng, nt, seed % 4)
    if seed % 3 == 0 and nt >= 2:                           # queries exactly between two targets: the runner-up is as close as the neighbour
        a = tgt[rng.integers(0, nt, ns)]; b = tgt[rng.integers(0, nt, ns)]
        src = (0.5 * (a.astype(np.float64) + b.astype(np.float64))).astype(f32)
    else:
        src = (tgt[rng.integers(0, nt, ns)] + rng.normal(0, 0.02, (ns, 3))).astype(f32) if seed % 2 else make_cloud(rng, ns, (seed + 1) % 4)
    tn, sn = unit_normals(rng, nt), unit_normals(rng, ns)
    c = gpu_ctx_factory()
    c.params.max_distance = 4.0; c.params.metric = 1; c.params.rejection = 0; c.params.n_iterations = 25; c.params.knn_backend = LBVH; c.push_params()
    c.set_target(tgt, tn); c.set_source(src, sn)
    m, d2 = c.match(np.eye(4))
    sub = rng.choice(ns, min(ns, 2000), replace=False)
    mo, do = orc.knn3(src[sub], tgt, 4.0)
    assert np.array_equal(m["idx"][sub], mo["idx"]) and np.array_equal(d2[sub].view(np.uint32), do.view(np.uint32))
    out = []
    for inc in (1, 0):
        c.params.knn_incremental = inc; c.push_params()
        _, recs, rc = c.run(np.eye(4), check=False)
        out.append((rc, recs))
    assert out[0][0] == out[1][0] and len(out[0][1]) == len(out[1][1])
    for k, (a, b) in enumerate(zip(out[0][1], out[1][1])):
        assert a["n_valid"] == b["n_valid"] and np.array_equal(a["pose"], b["pose"]), k
