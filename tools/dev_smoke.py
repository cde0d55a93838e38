import sys, os, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "icp-variants_amd", "python"))
import numpy as np
from icp_amd import binding, synth
from oracle import oracle as orc
d = np.load(os.path.join(ROOT, "tests/golden/bunny_pair.npz")); g = np.load(os.path.join(ROOT, "tests/golden/bunny_oracle.npz"))
sp, sn, sc, tp, tn, tc = [d[k] for k in ("src_pts","src_nrm","src_rgba","tgt_pts","tgt_nrm","tgt_rgba")]
ctx = binding.Context(0)
print(binding.load_library().icp_version())
ctx.params.max_distance = 0.0003; ctx.push_params()
ctx.set_target(tp, tn, tc); ctx.set_source(sp, sn, sc)
I = np.eye(4, dtype=np.float32)
m, d2 = ctx.match(I)
print("knn3 identity idx equal:", np.array_equal(m["idx"], g["knn3_identity_idx"]), "d2 bits equal:", np.array_equal(d2.view(np.uint32), g["knn3_identity_d2"].view(np.uint32)))
for metric in (0,1,2):
  for w in (0,1,2):
    ctx.params.metric = metric; ctx.params.weighting = w; ctx.params.n_iterations = 20; ctx.push_params()
    mm, sums, nv = ctx.correspond(I)
    if metric == 1:
        print(" w", w, "post idx eq:", np.array_equal(mm["idx"], g["iter0_w%d_idx"%w]), "weights bits eq:", np.array_equal(mm["weight"].view(np.uint32), g["iter0_w%d_weight"%w].view(np.uint32)), "nvalid", nv)
    pose, recs, rc = ctx.run(I)
    gp = g["m%d_w%d_r0_mode1_poses"%(metric,w)]
    err = np.abs(pose - gp[-1]).max()
    errs = [np.abs(r["pose"]-gp[i]).max() for i,r in enumerate(recs)]
    print("metric", metric, "w", w, "rc", rc, "iters", len(recs), "final max|dpose| vs oracle-exact:", err, "max over iters", max(errs), "nvalid", recs[-1]["n_valid"], g["m%d_w%d_r0_mode1_nvalid"%(metric,w)][-1], ctx.timing())
# multires
for metric in (0,1,2):
    ctx.params.metric = metric; ctx.params.weighting = 0; ctx.params.multires = 1; ctx.push_params()
    pose, recs, rc = ctx.run(I)
    gp = g["m%d_w0_r1_mode1_poses"%metric]
    print("multires metric", metric, "iters", len(recs), len(gp), "err", np.abs(pose-gp[-1]).max(), [r["n_src"] for r in recs][:8], list(g["m%d_w0_r1_mode1_nsrc"%metric][:8]))
ctx.params.multires = 0
# synthetic mid-size
pr = synth.eth_like_pair(0, n_tilt=86, n_beam=270)
print("synthetic", pr["src_pts"].shape)
ctx2 = binding.Context(0)
ctx2.params.max_distance = 10.0; ctx2.params.metric = 1; ctx2.params.n_iterations = 10; ctx2.push_params()
ctx2.set_target(pr["tgt_pts"], pr["tgt_nrm"], pr["tgt_rgba"]); ctx2.set_source(pr["src_pts"], pr["src_nrm"], pr["src_rgba"])
t=time.time(); m, d2 = ctx2.match(I); print("gpu match s", time.time()-t)
q = orc.transform_points(pr["src_pts"], I)
t=time.time(); mo, d2o = orc.knn3(q, pr["tgt_pts"], 10.0); print("cpu match s", time.time()-t, orc.num_threads())
print("mid idx eq", np.array_equal(m["idx"], mo["idx"]), "d2 eq", np.array_equal(d2.view(np.uint32), d2o.view(np.uint32)))
pose, recs, rc = ctx2.run(I)
prm = orc.make_params(metric=1, n_iterations=10, max_distance=10.0, solver_mode=1)
po, ro = orc.estimate_pose(prm, pr["src_pts"], pr["src_nrm"], None, pr["tgt_pts"], pr["tgt_nrm"], None, I)
print("synthetic run err vs oracle", np.abs(pose-po).max(), "vs gt", np.abs(pose-pr["gt"]).max(), ctx2.timing())
