import sys, os
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "icp-variants_amd", "python"))
import numpy as np
from icp_amd import binding, synth
binding.LIB_PATH = os.path.join(binding.PKG_ROOT, "lib", "libicp_hip_stats.so")
p = synth.eth_like_pair(0)
c = binding.Context(0)
c.params.max_distance = 1e30; c.params.knn_backend = 1; c.push_params()
c.set_target(p["tgt_pts"], p["tgt_nrm"]); c.set_source(p["src_pts"], p["src_nrm"])
for name, T in (("identity", np.eye(4)), ("gt", p["gt"])):
    m, st = c.match(T)
    st = st.astype(np.int64); nodes = st % 4096; leaves = st // 4096
    print(name, "PACKET per-wave nodes mean/med/p99/max", nodes.mean(), np.median(nodes), np.percentile(nodes, 99), nodes.max(), "leaves", leaves.mean(), np.median(leaves), np.percentile(leaves, 99), leaves.max())
