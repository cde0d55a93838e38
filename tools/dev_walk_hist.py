"""Dev tool (needs a build with ICP_DEBUG_STEPS=1 ICP_SHARE_WALKS=0 -- a shared walk has no per-query length --: ICP_HIP_LIB=.../libicp_hip_dbg.so): distribution of the tree-walk length per query
and per wave for chosen ICP iterations of configs[1].  usage: ICP_HIP_LIB=... python tools/dev_walk_hist.py"""
import sys, os, ctypes as C
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "icp-variants_amd", "python"))
import numpy as np
from icp_amd import binding, synth
p = synth.eth_like_pair(0)
n = len(p["src_pts"])
c = binding.Context(0)
c.params.max_distance = 10.0; c.params.metric = 1; c.params.knn_backend = 1
c.set_stage_timing(0)
c.push_params(); c.set_target(p["tgt_pts"], p["tgt_nrm"]); c.set_source(p["src_pts"], p["src_nrm"])
for iters in (1, 2, 3, 5, 9, 13, 20, 40):
    c.params.n_iterations = iters; c.push_params()
    c.run(np.eye(4))
    buf = np.zeros(n, np.int32)
    rc = c.lib.icp_debug_steps(c.h, buf.ctypes.data_as(C.c_void_p), C.c_int32(n))
    assert rc == 0, rc
    tier2 = buf == -2; ver = buf == 0       # -2: second verification tier (two leaves); 0: verified
    nodes = np.where(buf > 0, buf & 0xFFFF, 0); leaves = np.where(buf > 0, buf >> 16, 0); tot = nodes + leaves
    w = tot[: (n // 64) * 64].reshape(-1, 64)
    wmax = w.max(1); wsum = w.sum(1)
    walk = tot[tot > 0]
    pct = lambda a, q: [int(x) for x in np.percentile(a, q)] if len(a) else []
    print("iteration %2d: verified %6d two-leaf tier %6d walked %6d | per query nodes mean %.1f leaves mean %.1f total pct[50,90,99,99.9,100] %s | "
          "per wave: max-lane pct[50,90,99,100] %s, lane utilisation %.2f, waves with max > 40: %d of %d"
          % (iters - 1, ver.sum(), tier2.sum(), len(walk), nodes[tot > 0].mean() if len(walk) else 0, leaves[tot > 0].mean() if len(walk) else 0,
             pct(walk, [50, 90, 99, 99.9, 100]), pct(wmax, [50, 90, 99, 100]), wsum.sum() / max(1, (wmax * 64).sum()), (wmax > 40).sum(), len(wmax)), flush=True)
