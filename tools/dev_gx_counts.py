"""Dev tool (times build): events of the hand-over between blocks (GX), per launch of a run.  usage: ICP_HIP_LIB=.../libicp_hip_times.so python tools/dev_gx_counts.py [iterations ...]"""
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
prev = np.zeros(16, np.uint32)
names = ["posted", "claimed by helpers", "taken back", "results folded", "helper waves", "helper rounds", "rounds with a claim"]
for iters in [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4, 6, 8, 12]:
    c.params.n_iterations = iters; c.push_params()
    buf = np.zeros(16, np.uint32)
    c.lib.icp_debug_gx_counters(c.h, buf.ctypes.data_as(C.c_void_p), C.c_int32(1))
    c.run(np.eye(4))
    assert c.lib.icp_debug_gx_counters(c.h, buf.ctypes.data_as(C.c_void_p), C.c_int32(1)) == 0
    d = buf.astype(np.int64) - prev                       # a run of `iters` iterations minus the run before = the last launches
    print("run of %2d iterations: " % iters + "  ".join("%s %d" % (n, v) for n, v in zip(names, buf[:7])) + "   | added by the last launches: " + " ".join(str(int(x)) for x in d[:7]))
    prev = buf.astype(np.int64)
