"""Dev check at 4x the bench size (1.48 M x 1.48 M points): match indices bit-exact against the oracle's kd-tree, incremental
run equal to the non-incremental one, timings.  usage (GPU box, not collected by pytest): python tests/manual_scale_check.py"""
import sys, os, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "icp-variants_amd", "python"))
import numpy as np
from icp_amd import binding, synth
from oracle import oracle as orc
if len(sys.argv) > 1: binding.LIB_PATH = os.path.join(binding.PKG_ROOT, "lib", sys.argv[1])
orc.build()
p = synth.eth_like_pair(0, n_tilt=688, n_beam=2154)
print("points", len(p["src_pts"]), len(p["tgt_pts"]), flush=True)
c = binding.Context(0)
c.params.max_distance = 10.0; c.params.metric = 1; c.params.n_iterations = 30; c.params.knn_backend = 1; c.params.rejection = 1; c.push_params()
t0 = time.perf_counter(); c.set_target(p["tgt_pts"], p["tgt_nrm"]); c.set_source(p["src_pts"], p["src_nrm"]); print("set_target+source %.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
res = []
for inc in (1, 0):
    c.params.knn_incremental = inc; c.push_params()
    c.run(np.eye(4))
    t0 = time.perf_counter(); pose, recs, rc = c.run(np.eye(4)); dt = time.perf_counter() - t0
    res.append((pose, [r["n_valid"] for r in recs]))
    print("incremental=%d: %.3f ms per iteration, rc %d" % (inc, dt / 30 * 1e3, rc), flush=True)
T = synth.make_pose((0.01, -0.02, 0.015), (0.05, -0.03, 0.02)).astype(np.float32)
m, d2 = c.match(T)
kd = orc.KdTree(p["tgt_pts"])
q = orc.transform_points(p["src_pts"], T)
mo, do = kd.query(q, 10.0)
# (after the timings: the oracle's OpenMP workers spin for a while and would slow the enqueueing host thread)
print("match idx equal:", bool(np.array_equal(m["idx"], mo["idx"])), " d2 bits equal:", bool(np.array_equal(d2.view(np.uint32), do.view(np.uint32))), flush=True)
print("incremental == full search:", bool(np.array_equal(res[0][0], res[1][0]) and res[0][1] == res[1][1]))
gt = p["gt"]; P = res[0][0].astype(np.float64)
print("translation error vs gt %.5f m" % float(np.linalg.norm(P[:3, 3] - gt[:3, 3])))
