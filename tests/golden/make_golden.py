"""Generates the committed fixtures under tests/golden/ (run in the build container only).

Inputs  : the bunny pair shipped with the reference (Data/bunny_part1.off = target, Data/bunny_part2_trans.off
          = source, BunnyDataLoader.h:10-11), parsed with icp_amd.meshio (SimpleMesh::loadMesh semantics) and turned
          into point clouds with PointCloud(SimpleMesh) semantics.  The .off files are DATA of the reference,
          stored here as float32 arrays (bunny_pair.npz) because /root/reference does not exist on the GPU box.
Outputs : what the CPU oracle (oracle/icp_oracle.cpp, "faithful" fp32 mode and "exact" fp64 mode) produces for the
          linear rows of Data/bunny_experiments.csv (bunny003-005, 203-205, 303-305) plus normals weighting:
          per-iteration poses / valid counts, iteration-0 matches, weights after pruning.
PARITY UNPINNED: the reference holds no expected outputs for this path; these vectors pin the ORACLE against
regressions and give the GPU tests a fixed target, they are not outputs of the reference binary.
"""
import os
import sys
import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "icp-variants_amd", "python"))
from icp_amd import meshio          # noqa: E402
from oracle import oracle as orc     # noqa: E402

REF_DATA = "/root/reference/Data"
OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    tv, _, tt = meshio.load_off(os.path.join(REF_DATA, "bunny_part1.off"))
    sv, _, st = meshio.load_off(os.path.join(REF_DATA, "bunny_part2_trans.off"))
    tp, tn, tc = meshio.mesh_to_cloud(tv, tt)
    sp, sn, sc = meshio.mesh_to_cloud(sv, st)
    np.savez_compressed(os.path.join(OUT, "bunny_pair.npz"), src_pts=sp, src_nrm=sn, src_rgba=sc, tgt_pts=tp, tgt_nrm=tn, tgt_rgba=tc,
                        src_tris=st, tgt_tris=tt,
                        gt_src_idx=np.array([215, 424, 640, 1023], np.int32), gt_tgt_idx=np.array([294, 258, 1238, 1310], np.int32))   # main.cpp:110-120
    out = {}
    I = np.eye(4, dtype=np.float32)
    # iteration-0 stage outputs at identity
    m0, d20 = orc.knn3(sp, tp, 0.0003)
    out["knn3_identity_idx"] = m0["idx"]; out["knn3_identity_d2"] = d20
    for metric in (0, 1, 2):
        for weighting in (0, 1, 2):
            for multires in (0, 1):
                if multires and weighting != 0:
                    continue
                for mode in (0, 1):
                    prm = orc.make_params(metric=metric, weighting=weighting, multires=multires, n_iterations=20, max_distance=0.0003, solver_mode=mode)
                    pose, recs = orc.estimate_pose(prm, sp, sn, sc, tp, tn, tc, I)
                    key = "m%d_w%d_r%d_mode%d" % (metric, weighting, multires, mode)
                    out[key + "_poses"] = np.stack([r["pose"] for r in recs])
                    out[key + "_nvalid"] = np.array([r["n_valid"] for r in recs], np.int32)
                    out[key + "_nsrc"] = np.array([r["n_src"] for r in recs], np.int32)
                    print(key, len(recs), recs[-1]["n_valid"], orc.rmse(sp[[215, 424, 640, 1023]], tp[[294, 258, 1238, 1310]], pose))
    # one full single-iteration record per weighting at identity (matches after weighting + pruning)
    for weighting in (0, 1, 2):
        prm = orc.make_params(metric=1, weighting=weighting, n_iterations=1, max_distance=0.0003)
        pose, m, nv, _, _ = orc.iterate(prm, sp, sn, sc, tp, tn, tc, I)
        out["iter0_w%d_idx" % weighting] = m["idx"]; out["iter0_w%d_weight" % weighting] = m["weight"]
    np.savez_compressed(os.path.join(OUT, "bunny_oracle.npz"), **out)
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
