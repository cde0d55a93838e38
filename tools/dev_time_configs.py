"""Times the other BASELINE configs at full size on the GPU (dev tool): config 3 (projective + symmetric, 1077x344 and 640x480)
and config 5 (6-D colour k-NN, multires, 640x480)."""
import sys, os, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "icp-variants_amd", "python"))
import numpy as np
from icp_amd import binding, synth

def run(name, ctx, iters_note=""):
    for rep in range(2):
        pose, recs, rc = ctx.run(np.eye(4), check=False)
    t = ctx.timing()
    print(name, "rc", rc, "iters", t["iterations"], {k: round(v / max(t["iterations"], 1), 4) for k, v in t.items() if k != "iterations"}, "n_valid last", recs[-1]["n_valid"], iters_note, flush=True)
    return pose

# config 3: organised 1077 x 344 pinhole, projective + rejection + symmetric, maxDist^2 = 0.1, 35 iterations
for (W, H, K) in ((1077, 344, np.array([[540.0, 0, 538.0], [0, 540.0, 171.5], [0, 0, 1]])), (640, 480, None)):
    p = synth.rgbd_pair(0, width=W, height=H, K=K)
    c = binding.Context(0)
    K = p["K"]
    c.params.matching = 1; c.params.metric = 2; c.params.max_distance = 0.1; c.params.n_iterations = 35
    c.params.fx, c.params.fy, c.params.cx, c.params.cy, c.params.width, c.params.height = float(K[0,0]), float(K[1,1]), float(K[0,2]), float(K[1,2]), W, H
    c.push_params(); c.set_target(p["tgt_pts"], p["tgt_nrm"]); c.set_source(p["src_pts"], p["src_nrm"])
    pose = run("config3 projective+symmetric %dx%d" % (W, H), c)
    print("   pose err vs gt", np.abs(pose - p["gt"]).max())
    c.close()
# config 5: colour ICP 6-D k-NN, colour weighting, multires, TUM-like 640x480
p = synth.rgbd_pair(0)
tp, tn, tc = synth.compact_valid(p["tgt_pts"], p["tgt_nrm"], p["tgt_rgba"])
c = binding.Context(0)
c.params.color_icp = 1; c.params.weighting = 3; c.params.multires = 1; c.params.metric = 1; c.params.max_distance = 0.1; c.params.n_iterations = 35; c.params.knn_backend = 1
c.push_params(); c.set_target(tp, tn, tc); c.set_source(p["src_pts"], p["src_nrm"], p["src_rgba"])
pose = run("config5 colour 6-D multires 640x480 (N_tgt=%d)" % len(tp), c)
print("   pose err vs gt", np.abs(pose - p["gt"]).max())
