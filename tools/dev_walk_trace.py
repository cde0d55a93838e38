"""Dev tool (a build with ICP_DEBUG_STEPS=1 ICP_DEBUG_TIMES=1 ICP_DEBUG_WALK_TRACE=1): the time line of the last SPARSE walk (a wave with <= 3
walkers, spread start) of the last launch of a run: clock at entry, after the spread start, at the top of every pass of the hand-over loop, at exit.
usage: ICP_HIP_LIB=.../libicp_hip_trace.so python tools/dev_walk_trace.py [iterations ...]"""
import sys, os, ctypes as C
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "icp-variants_amd", "python"))
import numpy as np
from icp_amd import binding, synth
p = synth.eth_like_pair(0)
c = binding.Context(0)
c.params.max_distance = 10.0; c.params.metric = 1; c.params.knn_backend = 1
c.set_stage_timing(0)
c.push_params(); c.set_target(p["tgt_pts"], p["tgt_nrm"]); c.set_source(p["src_pts"], p["src_nrm"])
for iters in [int(a) for a in sys.argv[1:]] or [31, 34, 37, 40]:
    c.params.n_iterations = iters; c.push_params()
    c.run(np.eye(4))
    buf = np.zeros(64, np.uint32)
    assert c.lib.icp_debug_gx_counters(c.h, buf.ctypes.data_as(C.c_void_p), C.c_int32(2)) == 0
    t = buf.astype(np.int64)
    trips = int(t[4])
    us = lambda x: ((x - t[2]) & 0xFFFFFFFF) * 0.01
    print("run of %d iterations: last sparse walk: %d walker(s), paths %d, spread done at %.2f us, %d passes, %d polls, exit at %.2f us" % (iters, t[0], t[1] & 0xFF, us(t[3]), trips, t[5], us(t[6])))
    print("   pass tops (us after entry): " + " ".join("%.2f" % us(t[7 + j]) for j in range(1, min(trips, 55) + 1)))
