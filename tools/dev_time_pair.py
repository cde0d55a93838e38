"""Dev timing helper: end-to-end wall clock of one pair -- icp_set_target (upload + index build), icp_set_source, first icp_run
(builds the Morton-sorted source level), second icp_run.  usage: python tools/dev_time_pair.py"""
import sys, os, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "icp-variants_amd", "python"))
import numpy as np
from icp_amd import binding, synth
p = synth.eth_like_pair(0)
c = binding.Context(0)
c.params.max_distance = 10.0; c.params.metric = 1; c.params.n_iterations = 50; c.params.knn_backend = 1; c.params.rejection = 1; c.push_params()
c.set_stage_timing(0)
for rep in range(3):
    t0 = time.perf_counter(); c.set_target(p["tgt_pts"], p["tgt_nrm"]); t1 = time.perf_counter()
    c.set_source(p["src_pts"], p["src_nrm"]); t2 = time.perf_counter()
    pose, recs, rc = c.run(np.eye(4)); t3 = time.perf_counter()
    pose, recs, rc = c.run(np.eye(4)); t4 = time.perf_counter()
    print("rep %d: set_target %.3f ms, set_source %.3f ms, first run %.3f ms, second run %.3f ms" % (rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3))
