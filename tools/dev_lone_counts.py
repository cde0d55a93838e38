"""Dev tool (times build): how often does a wave with ONE walker complete the level-synchronous search, how often does its frontier overflow and
the wave start over on the general path?  (a) the cloud of tests/test_gpu_seeded.py::test_seeded_search_lone_walkers_small_and_huge_radius,
(b) a 50-iteration run of configs[1].  usage: ICP_HIP_LIB=.../libicp_hip_times.so python tools/dev_lone_counts.py"""
import sys, os, ctypes as C
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "icp-variants_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from icp_amd import binding, synth
f32 = np.float32
def counters(c, reset=1):
    buf = np.zeros(16, np.uint32)
    assert c.lib.icp_debug_gx_counters(c.h, buf.ctypes.data_as(C.c_void_p), C.c_int32(reset)) == 0
    return buf.astype(np.int64)
def small_motion(rng, scale):
    w = rng.normal(size=3) * scale; t = rng.normal(size=3) * scale
    th = np.linalg.norm(w); k = w / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    T = np.eye(4); T[:3, :3] = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K; T[:3, 3] = t
    return T
rng = np.random.default_rng(9100)
nt = 30000
tgt = np.c_[rng.uniform(-2, 2, (nt, 2)), rng.normal(0, 2e-3, nt)].astype(f32)
v = rng.normal(size=(4000, 3)); shell = np.array([10.0, 10.0, 10.0]) + v / np.linalg.norm(v, axis=1, keepdims=True) * (1.0 + rng.normal(0, 1e-4, (4000, 1)))
n_close, n_bis, n_far = 6000, 60, 12
close = tgt[rng.integers(0, nt, n_close)] + rng.normal(0, 1e-4, (n_close, 3))
a = tgt[rng.integers(0, nt, n_bis)]
d = np.linalg.norm(tgt[None, :, :2] - a[:, None, :2], axis=2); d[d == 0] = np.inf
b = tgt[np.argmin(d, axis=1)]
far = np.c_[rng.uniform(-2, 2, (n_far, 2)), rng.uniform(1.5, 3.0, n_far)]
tgt = np.r_[tgt, shell.astype(f32)]; nt = len(tgt)
src = np.r_[close, 0.5 * (a.astype(np.float64) + b.astype(np.float64)), far, np.array([[10.0, 10.0, 10.0]])]
src = src[rng.permutation(len(src))].astype(f32)
v = rng.normal(size=(nt, 3)); tn = (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(f32)
v = rng.normal(size=(len(src), 3)); sn = (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(f32)
c = binding.Context(0)
c.params.max_distance = 100.0; c.params.metric = 1; c.params.rejection = 0; c.params.knn_backend = 1; c.push_params()
c.set_target(tgt, tn); c.set_source(src, sn)
poses = [np.eye(4, dtype=f32)]
for j in range(7):
    poses.append((small_motion(rng, max(0.01 * 0.3 ** j, 2e-5)) @ poses[-1].astype(np.float64)).astype(f32))
counters(c)
c.match_seeded(poses)
k = counters(c)
print("test cloud, chain of 7 seeded launches: lone searches completed %d, started over on the general path %d" % (k[11], k[12]))
c.close()
p = synth.eth_like_pair(0)
c = binding.Context(0)
c.params.max_distance = 10.0; c.params.metric = 1; c.params.knn_backend = 1; c.params.n_iterations = 50
c.set_stage_timing(0); c.push_params(); c.set_target(p["tgt_pts"], p["tgt_nrm"]); c.set_source(p["src_pts"], p["src_nrm"])
counters(c)
c.run(np.eye(4))
k = counters(c)
print("configs[1], 50 iterations: lone searches completed %d, started over on the general path %d" % (k[11], k[12]))
