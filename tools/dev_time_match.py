import sys, os, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "icp-variants_amd", "python"))
import numpy as np
from icp_amd import binding, synth
p = synth.eth_like_pair(0)
for backend in (1, 0):
    c = binding.Context(0)
    c.params.max_distance = 10.0; c.params.metric = 1; c.params.n_iterations = 50; c.params.knn_backend = backend; c.push_params()
    t0 = time.time(); c.set_target(p["tgt_pts"], p["tgt_nrm"]); c.set_source(p["src_pts"], p["src_nrm"]); print("backend", backend, "upload+build s", time.time() - t0)
    for rep in range(2):
        pose, recs, rc = c.run(np.eye(4))
        t = c.timing(); print(" run", rep, {k: (round(v / t["iterations"], 4) if k != "iterations" else v) for k, v in t.items()})
    T = synth.make_pose((0.05, -0.04, 0.08), (0.3, -0.2, 0.1)).astype(np.float32)
    c.params.n_iterations = 1; c.push_params()
    pose, recs, rc = c.run(T); print(" perturbed single iter", c.timing())
