"""Host I/O either side of the hot path (icp_amd/formats.py): round trips and the reference loaders' conventions."""
import os
import numpy as np
import pytest

f32 = np.float32


def test_pcd_round_trip_ascii_and_binary(tmp_path):
    from icp_amd import formats
    rng = np.random.default_rng(0)
    xyz = rng.uniform(-8, 8, (1000, 3)).astype(f32); xyz[7] = np.nan
    for binary in (True, False):
        p = str(tmp_path / ("c%d.pcd" % binary))
        formats.write_pcd(p, xyz, binary=binary)
        back = formats.read_pcd(p)
        assert back.dtype == np.float32 and back.shape == xyz.shape
        assert np.array_equal(back[~np.isnan(xyz).any(1)], xyz[~np.isnan(xyz).any(1)]) and np.isnan(back[7]).all()
    # extra fields (intensity, normals) are skipped, as loadPCDFile<PointXYZ> does
    p = str(tmp_path / "extra.pcd")
    with open(p, "w") as f:
        f.write("# .PCD v0.7\nVERSION 0.7\nFIELDS x y z intensity\nSIZE 4 4 4 4\nTYPE F F F F\nCOUNT 1 1 1 1\nWIDTH 2\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS 2\nDATA ascii\n1 2 3 9\n4 5 6 9\n")
    assert np.array_equal(formats.read_pcd(p), np.array([[1, 2, 3], [4, 5, 6]], f32))


def test_pose_csv_follows_eth_loader(tmp_path):
    """ETHDataLoader.h:40-61: row index+1 (header skipped), pose from columns 4..15 row-major 3x4."""
    from icp_amd import formats, synth
    T = synth.make_pose((0.1, -0.2, 0.3), (1.0, -2.0, 0.5))
    p = str(tmp_path / "plain_global.csv")
    formats.write_pose_csv(p, [dict(id="0", source="Hokuyo_1.pcd", target="Hokuyo_0.pcd", pose=T), dict(id="1", source="Hokuyo_2.pcd", target="Hokuyo_1.pcd", pose=np.eye(4))])
    rows = formats.read_pose_csv(p)
    assert len(rows) == 2 and rows[0]["source"] == "Hokuyo_1.pcd" and rows[1]["target"] == "Hokuyo_1.pcd"
    assert np.allclose(rows[0]["pose"], T, atol=1e-6) and rows[0]["pose"].dtype == np.float32
    # main.cpp:420-429: angles and translation scaled by 0.1
    S = formats.scaled_initial_pose(rows[0]["pose"], 0.1)
    assert np.allclose(S, synth.make_pose((0.01, -0.02, 0.03), (0.1, -0.2, 0.05)), atol=1e-6)


def test_tum_lists_and_trajectory(tmp_path):
    """VirtualSensor.h:196-250: 3 header lines; trajectory poses are inverted; nearest timestamp wins (first minimum)."""
    from icp_amd import formats, synth
    d = tmp_path
    (d / "depth.txt").write_text("# depth maps\n# file: x\n# timestamp filename\n1.00 depth/1.00.png\n1.10 depth/1.10.png\n")
    ts, names = formats.read_tum_file_list(str(d / "depth.txt"))
    assert ts.tolist() == [1.0, 1.1] and names == ["depth/1.00.png", "depth/1.10.png"]
    T = synth.make_pose((0.0, 0.0, np.pi / 2), (1, 2, 3))
    qz, qw = np.sin(np.pi / 4), np.cos(np.pi / 4)
    (d / "groundtruth.txt").write_text("# gt\n# file\n# t tx ty tz qx qy qz qw\n0.95 1 2 3 0 0 %.12f %.12f\n1.20 0 0 0 0 0 0 1\n" % (qz, qw))
    tts, poses = formats.read_tum_trajectory(str(d / "groundtruth.txt"))
    assert np.allclose(poses[0], np.linalg.inv(T), atol=1e-6) and np.allclose(poses[1], np.eye(4))
    assert np.allclose(formats.nearest_pose(tts, poses, 1.0), poses[0]) and np.allclose(formats.nearest_pose(tts, poses, 1.15), poses[1])
    raw = np.array([[0, 5000], [2500, 65535]], np.uint16)
    dep = formats.decode_tum_depth(raw)
    assert dep[0, 0] == -np.inf and dep[0, 1] == 1.0 and dep[1, 0] == 0.5 and dep.dtype == np.float32


def test_off_writer_reader_round_trip(tmp_path, bunny):
    from icp_amd import formats, meshio
    p = str(tmp_path / "b.off")
    v = bunny["src_pts"].copy(); v[3] = np.nan
    formats.write_off(p, v, None, bunny["src_tris"])
    vv, cc, tt = meshio.load_off(p)
    assert np.array_equal(tt, bunny["src_tris"]) and np.allclose(vv[4:], v[4:], rtol=1e-5) and np.all(vv[3] == 0)
    formats.write_ply(str(tmp_path / "b.ply"), bunny["src_pts"][:5], bunny["src_nrm"][:5])
    assert open(str(tmp_path / "b.ply")).read().startswith("ply\nformat ascii 1.0\nelement vertex 5\n")


def test_end_to_end_eth_layout_on_synthetic_files(tmp_path, orc):
    """A miniature ETH-style data set on disk (PCD scans + pose CSV) driven like alignETH (main.cpp:401-457) through the oracle."""
    from icp_amd import formats, synth
    pr = synth.eth_like_pair(0, n_tilt=30, n_beam=90)
    formats.write_pcd(str(tmp_path / "Hokuyo_0.pcd"), pr["tgt_pts"]); formats.write_pcd(str(tmp_path / "Hokuyo_1.pcd"), pr["src_unperturbed"], binary=False)
    big = synth.perturbation(5, scale=1.0)                      # benchmark-style perturbation; the driver scales it by 0.1
    formats.write_pose_csv(str(tmp_path / "plain_global.csv"), [dict(id="0", source="Hokuyo_1.pcd", target="Hokuyo_0.pcd", pose=big)])
    row = formats.read_pose_csv(str(tmp_path / "plain_global.csv"))[0]
    src = formats.read_pcd(str(tmp_path / row["source"])); tgt = formats.read_pcd(str(tmp_path / row["target"]))
    assert np.allclose(src, pr["src_unperturbed"], atol=1e-6) and np.array_equal(tgt, pr["tgt_pts"])
    S = formats.scaled_initial_pose(row["pose"], 0.1)
    moved = synth.apply_pose(S, src)
    prm = orc.make_params(metric=0, n_iterations=15, max_distance=10.0, knn_kdtree=1)
    pose, recs = orc.estimate_pose(prm, moved, np.tile(f32([0, 0, 1]), (len(src), 1)), None, tgt, np.tile(f32([0, 0, 1]), (len(tgt), 1)), None, np.eye(4))
    assert orc.rmse(moved, src, pose) < 0.5 * orc.rmse(moved, src, np.eye(4))
