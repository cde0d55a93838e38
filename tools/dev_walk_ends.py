"""Dev tool (a build with ICP_DEBUG_STEPS=1 ICP_DEBUG_TIMES=1 ICP_DEBUG_WALK_ENDS=1): where do the seeded walks of the last launch of a run end -- in the old neighbour's leaf, in the old runner-up's leaf
(the second leaf of the two-leaf tier), or elsewhere?  usage: ICP_HIP_LIB=.../libicp_hip_times.so python tools/dev_walk_ends.py [iterations ...]"""
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
prev = np.zeros(16, np.int64); last = 0
for iters in [int(a) for a in sys.argv[1:]] or [2, 3, 4, 6, 8, 10, 12, 14, 16, 20, 30]:
    c.params.n_iterations = iters; c.push_params()
    buf = np.zeros(16, np.uint32)
    c.lib.icp_debug_gx_counters(c.h, buf.ctypes.data_as(C.c_void_p), C.c_int32(1))
    c.run(np.eye(4))
    assert c.lib.icp_debug_gx_counters(c.h, buf.ctypes.data_as(C.c_void_p), C.c_int32(1)) == 0
    cur = buf.astype(np.int64); d = cur - prev                # a run of `iters` iterations minus the run before = iterations last .. iters - 1
    print("iterations %2d..%2d: seeded walks %7d, ending in the old neighbour's leaf %5.1f %%, in the old runner-up's leaf %5.1f %%"
          % (last, iters - 1, d[8], 100.0 * d[9] / max(d[8], 1), 100.0 * d[10] / max(d[8], 1)))
    prev = cur; last = iters
