"""Dev timing helper: configs[0] (bunny pair, 1054 x 1359, point-to-point, 20 iterations) on the GPU, both matchers."""
import sys, os, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "icp-variants_amd", "python"))
import numpy as np
from icp_amd import binding
d = np.load(os.path.join(ROOT, "tests", "golden", "bunny_pair.npz"))
for backend in (0, 1):
    for metric in (0, 1):
        c = binding.Context(0)
        c.params.max_distance = 0.0003; c.params.metric = metric; c.params.n_iterations = 20; c.params.knn_backend = backend; c.push_params()
        c.set_stage_timing(0)
        c.set_target(d["tgt_pts"], d["tgt_nrm"]); c.set_source(d["src_pts"], d["src_nrm"])
        eye = binding.pose_to_c(np.eye(4, dtype=np.float32))
        for _ in range(3):
            q = eye.copy(); c.run_raw(q)
        t0 = time.perf_counter()
        for _ in range(20):
            q = eye.copy(); c.run_raw(q)
        dt = (time.perf_counter() - t0) / 20
        print("bunny backend=%s metric=%d: %.3f ms per 20-iteration run = %.1f us/iteration = %.0f iterations/s" % ("lbvh" if backend else "brute", metric, dt * 1e3, dt / 20 * 1e6, 20 / dt))
