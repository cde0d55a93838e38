"""Dev tool (needs a build with ICP_DEBUG_STEPS=1 ICP_DEBUG_TIMES=1: ICP_HIP_LIB=.../libicp_hip_times.so): where ONE iteration of k_icp_loop goes.
Lane 0 of every matcher wave stamps the 100 MHz clock at: iteration top, pose received, pair ready, partial stored, barrier passed; every reducer
wave at: polling starts, fold complete, total published; block 0 also at: totals received, pose published.  All times relative to the moment
the pose of that iteration went out.  usage: ICP_HIP_LIB=... ICP_HIP_DBG_ITER=30 python tools/dev_loop_times.py"""
import sys, os, ctypes as C
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "icp-variants_amd", "python"))
import numpy as np
from icp_amd import binding, synth
p = synth.eth_like_pair(0)
n = len(p["src_pts"])
c = binding.Context(0)
c.params.max_distance = 10.0; c.params.metric = 1; c.params.knn_backend = 1; c.params.n_iterations = 50
c.set_stage_timing(0)
c.push_params(); c.set_target(p["tgt_pts"], p["tgt_nrm"]); c.set_source(p["src_pts"], p["src_nrm"])
for _ in range(3):
    c.run(np.eye(4))
a, _, _ = c.iteration_times()
print("iteration times (us):", np.round(a * 1000, 1).tolist())
buf = np.zeros(n, np.int32)
assert c.lib.icp_debug_steps(c.h, buf.ctypes.data_as(C.c_void_p), C.c_int32(n)) == 0
BT = int(os.environ.get("ICP_DEV_BVH_THREADS", "256")); nb = (n + BT - 1) // BT; nw = nb * (BT // 64); nred = 68      # as the library under test was built
wraw = buf[: nw * 8].reshape(nw, 8)
w = wraw[:, :5].astype(np.uint32).astype(np.int64)
r = buf[nw * 8: (nw + 2 * nred) * 8].reshape(2 * nred, 8)[:, :5].astype(np.uint32).astype(np.int64)
sv = buf[(nw + 2 * nred) * 8: (nw + 2 * nred) * 8 + 8].astype(np.uint32).astype(np.int64)
t0 = int(np.uint32(buf[(nw + 2 * nred + 2) * 8]))
it = int(os.environ.get("ICP_HIP_DBG_ITER", "-1"))
print("iteration %d; all times in us after its pose went out" % it)
def row(name, x):
    x = (x - t0) * 0.01
    print("  %-28s min %7.2f  mean %7.2f  p50 %7.2f  p99 %7.2f  max %7.2f" % (name, x.min(), x.mean(), np.percentile(x, 50), np.percentile(x, 99), x.max()))
for j, name in enumerate(["matcher: iteration top", "matcher: pose received", "matcher: pair ready", "matcher: partial stored", "matcher: barrier passed"]):
    row(name, w[:, j])
for j, name in enumerate(["reducer: polling starts", "reducer: fold complete", "reducer: total published"]):
    row(name, r[:, j])
print("  solver: totals received %.2f, pose published %.2f" % ((sv[3] - t0) * 0.01, (sv[4] - t0) * 0.01))

flags = wraw[:, 5]
searched = flags & 0xFF; renewed = (flags >> 8) & 0xFF; had = (flags >> 16) & 1
print("  waves: %d with lanes that searched (walk / not handed back), %d with lanes of the two-leaf tier handed back, %d came in with parked data" % ((searched > 0).sum(), (renewed > 0).sum(), had.sum()))
late = np.argsort(w[:, 3])[-8:]
for i in late:
    print("   late wave %5d: searched %2d renewed %2d had %d  stamps %s" % (i, searched[i], renewed[i], had[i], np.round((w[i] - t0) * 0.01, 2)))
for label, m in (("waves that came in with parked data", had == 1), ("waves that loaded", had == 0)):
    if m.any():
        x = (w[m][:, 3] - t0) * 0.01
        print("  %-36s %5d  partial stored: mean %.2f p99 %.2f max %.2f" % (label, m.sum(), x.mean(), np.percentile(x, 99), x.max()))
