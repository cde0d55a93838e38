"""Small driver for rocprofv3 passes: config 2 (370k ETH-like pair, k-NN + point-to-plane), one icp_run of N iterations.
usage: python tools/prof_run.py [lbvh|brute] [iterations]"""
import sys, os
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "icp-variants_amd", "python"))
import numpy as np
from icp_amd import binding, synth
backend = sys.argv[1] if len(sys.argv) > 1 else "lbvh"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
p = synth.eth_like_pair(0)
c = binding.Context(0)
c.params.max_distance = 10.0; c.params.metric = 1; c.params.n_iterations = iters; c.params.knn_backend = 1 if backend == "lbvh" else 0
c.push_params()
c.set_target(p["tgt_pts"], p["tgt_nrm"]); c.set_source(p["src_pts"], p["src_nrm"])
pose, recs, rc = c.run(np.eye(4))
t = c.timing()
print(backend, {k: (v / t["iterations"] if k != "iterations" else v) for k, v in t.items()})
