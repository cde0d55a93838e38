"""Dev timing helper: 50 full-size point-to-plane iterations; prints the HIP-event stage split and the wall clock per iteration.
usage: python tools/dev_variants.py <library file under icp-variants_amd/lib>   (knobs: ICP_HIP_FUSE_POST, ICP_HIP_STAGE_EVENTS)"""
import sys, os, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "icp-variants_amd", "python"))
import numpy as np
from icp_amd import binding, synth
binding.LIB_PATH = os.path.join(binding.PKG_ROOT, "lib", sys.argv[1])
p = synth.eth_like_pair(0)
c = binding.Context(0)
c.params.max_distance = 10.0; c.params.metric = int(os.environ.get("ICP_DEV_METRIC", "1")); c.params.n_iterations = 50; c.params.knn_backend = 1; c.params.rejection = 1; c.push_params()
c.set_target(p["tgt_pts"], p["tgt_nrm"]); c.set_source(p["src_pts"], p["src_nrm"])
for rep in range(3):
    pose, recs, rc = c.run(np.eye(4))
eye = binding.pose_to_c(np.eye(4, dtype=np.float32))
t0 = time.perf_counter()
for rep in range(5):
    q = eye.copy(); c.run_raw(q)
wall = (time.perf_counter() - t0) / 250 * 1e3
t = c.timing()
print(sys.argv[1], "fuse=%s events=%s" % (os.environ.get("ICP_HIP_FUSE_POST", "1"), os.environ.get("ICP_HIP_STAGE_EVENTS", "1")),
      {k: round(v / t["iterations"], 4) for k, v in t.items() if k != "iterations"}, "wall_ms/iter %.4f" % wall, "pose checksum %.9f" % float(np.abs(pose).sum()))
