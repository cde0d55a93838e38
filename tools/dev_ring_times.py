"""Dev tool (needs a build with ICP_DEBUG_STEPS=1 ICP_DEBUG_TIMES=1: ICP_HIP_LIB=.../libicp_hip_times.so): the time line of ONE merged launch --
the reducer blocks of the iteration before (start, folded, total published; block 0: totals received, solved, pose published) beside
the matcher waves (start, front = pose received + verify, walks, post, reduce, end).  All times in us after the first stamp of the launch.
usage: ICP_HIP_LIB=... python tools/dev_ring_times.py [iterations ...]   (the last launch of a run of that many iterations is shown)"""
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
BT = int(os.environ.get("ICP_DEV_BVH_THREADS", "256")); nw = ((n + BT - 1) // BT) * (BT // 64)
NR = 34
for iters in [int(a) for a in sys.argv[1:]] or [3, 12, 40]:
    c.params.n_iterations = iters; c.push_params()
    for _ in range(2): c.run(np.eye(4))
    buf = np.zeros(n, np.int32)
    assert c.lib.icp_debug_steps(c.h, buf.ctypes.data_as(C.c_void_p), C.c_int32(n)) == 0
    rb = np.zeros((NR + 1) * 8, np.int32)
    assert c.lib.icp_debug_ring_times(c.h, rb.ctypes.data_as(C.c_void_p), C.c_int32(len(rb))) == 0
    w = buf[: nw * 8].reshape(nw, 8)[:, :6].astype(np.uint32).astype(np.int64)
    r = rb.reshape(NR + 1, 8).astype(np.uint32).astype(np.int64)
    t0 = min(w[:, 0].min(), r[:NR, 0].min())
    us = lambda x: (x - t0) * 0.01
    print("launch of iteration %d (reducer of iteration %d in its first %d blocks); spans %.2f us" % (iters - 1, iters - 2, NR, us(w[:, 5].max())))
    for j, name in enumerate(["reducer: block start", "reducer: folded", "reducer: total published"]):
        x = us(r[:NR, j]); print("  %-26s min %6.2f  mean %6.2f  max %6.2f" % (name, x.min(), x.mean(), x.max()))
    print("  block 0: totals received %.2f, solved %.2f, pose published %.2f" % tuple(us(r[NR, :3])))
    for j, name in enumerate(["matcher: start", "matcher: pose + verify", "matcher: walks", "matcher: post", "matcher: wave sums", "matcher: end"]):
        x = us(w[:, j]); print("  %-26s min %6.2f  mean %6.2f  p99 %6.2f  max %6.2f" % (name, x.min(), x.mean(), np.percentile(x, 99), x.max()))
