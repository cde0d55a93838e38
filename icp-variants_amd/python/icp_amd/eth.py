"""Host side of the ETH benchmark runner (reference: ETHDataLoader.h:11-107 and alignETH, main.cpp:343-514).

Pure host logic over `formats.py` plus the two device steps in front of the loop (k = 5 normals, `icp_estimate_normals`):
  rows      = load_rows(eth_dir, csv_name)                 # CSVReader + ETHDataLoader ctor
  src, tgt  = load_scans(eth_dir, csv_name, row)           # loadPCDFile<PointXYZ> x 2          (ETHDataLoader.h:66-98)
  pair      = prepare_pair(ctx, src, tgt, row["pose"])     # PointCloud(pcl) normals k = 5       (PointCloud.h:41-76)
                                                           # + pose_scaling 0.1 and change_pose  (main.cpp:420-429, PointCloud.h:277-282)
`pair` has the keys of `synth.eth_like_pair` (src_pts, src_nrm, tgt_pts, tgt_nrm, src_unperturbed, gt), so the bench and the
tests drive real files and synthetic scans through the same code.
Layout on disk, as the reference expects it under Data/: <eth_dir>/<name>_global.csv and <eth_dir>/<name>/<scan>.pcd.
PCL is absent here: the normals are this repo's exact k-NN + fp64 PCA (parity with pcl::NormalEstimation unpinned).
"""
import os
import numpy as np

from . import formats


def dataset_name(csv_name):
    """ETHDataLoader.h:20-23: three find_last_not_of / erase calls on the file name (character SETS, not suffixes)."""
    name = csv_name.rstrip(".csv")
    name = name.rstrip("_local")
    name = name.rstrip("_global")
    return name


def load_rows(eth_dir, csv_name):
    return formats.read_pose_csv(os.path.join(eth_dir, csv_name))


def load_scans(eth_dir, csv_name, row):
    base = os.path.join(eth_dir, dataset_name(csv_name))
    src = formats.read_pcd(os.path.join(base, row["source"]))
    tgt = formats.read_pcd(os.path.join(base, row["target"]))
    return src, tgt


def change_pose(pose, pts, nrm):
    """PointCloud::change_pose (PointCloud.h:277-282): p <- (T [p,1]).head(3), n <- (T [n,0]).head(3), fp32."""
    T = np.asarray(pose, np.float32)
    R, t = T[:3, :3], T[:3, 3]
    return (pts @ R.T + t).astype(np.float32), (nrm @ R.T).astype(np.float32)


def prepare_pair(ctx, src_xyz, tgt_xyz, benchmark_pose, pose_scaling=0.1, k=5):
    """What alignETH does between getItem and estimatePose (main.cpp:413-429), with the normals on the device."""
    src_xyz = np.ascontiguousarray(src_xyz, np.float32); tgt_xyz = np.ascontiguousarray(tgt_xyz, np.float32)
    src_nrm, _ = ctx.estimate_normals(src_xyz, k=k)                     # viewpoint (0,0,0): PCL's default
    tgt_nrm, _ = ctx.estimate_normals(tgt_xyz, k=k)
    S = formats.scaled_initial_pose(benchmark_pose, pose_scaling)
    moved, moved_nrm = change_pose(S, src_xyz, src_nrm)
    rgba = np.tile(np.array([255, 255, 255, 1], np.uint8), (len(src_xyz), 1))        # PointCloud.h:73
    return dict(src_pts=moved, src_nrm=moved_nrm, src_rgba=rgba, src_unperturbed=src_xyz,
                tgt_pts=tgt_xyz, tgt_nrm=tgt_nrm, tgt_rgba=np.tile(np.array([255, 255, 255, 1], np.uint8), (len(tgt_xyz), 1)),
                gt=np.linalg.inv(np.asarray(S, np.float64)), initial=S)


def write_synthetic_dataset(eth_dir, csv_name, pairs, binary=True):
    """Writes synthetic scans in the reference's on-disk layout (used by the tests and for rehearsing --eth-dir offline):
    `pairs` = list of dicts with src_unperturbed, tgt_pts and a benchmark `pose` (unscaled perturbation)."""
    base = os.path.join(eth_dir, dataset_name(csv_name))
    os.makedirs(base, exist_ok=True)
    rows = []
    for i, p in enumerate(pairs):
        s, t = "Hokuyo_%d_src.pcd" % i, "Hokuyo_%d_tgt.pcd" % i
        formats.write_pcd(os.path.join(base, s), p["src_unperturbed"], binary=binary)
        formats.write_pcd(os.path.join(base, t), p["tgt_pts"], binary=binary)
        rows.append(dict(id=str(i), source=s, target=t, pose=p["pose"]))
    formats.write_pose_csv(os.path.join(eth_dir, csv_name), rows)
    return rows
