"""On-disk formats either side of the ICP hot path -- pure host I/O, numpy only.

  * PCD (ASCII / binary, x y z [+ other float fields])   what pcl::io::loadPCDFile<PointXYZ> reads for the ETH scans
                                                          (reference ETHDataLoader.h:66-98)
  * Fontana / ETH pose CSV                                id,source,target,overlap,T00..T23  (ETHDataLoader.h:40-61,
                                                          CSVReader.h:29-45: plain split on ',' , first row = header)
  * TUM RGB-D lists and trajectory                        depth.txt / rgb.txt / groundtruth.txt (VirtualSensor.h:196-250):
                                                          3 header lines, "timestamp filename" rows, trajectory rows
                                                          "t tx ty tz qx qy qz qw" stored INVERTED (world->camera), nearest
                                                          timestamp lookup (VirtualSensor.h:126-137)
  * TUM depth decoding                                    uint16 / 5000, 0 -> MINF (VirtualSensor.h:119-124)
  * OFF / PLY writers                                     SimpleMesh::writeMesh (SimpleMesh.h:231-259), PointCloud::writeToFile
                                                          (PointCloud.h:229-247; ASCII PLY with x y z intensity normals)
Image decoding itself (FreeImage in the reference) is left to PIL when available.
"""
import os
import struct
import numpy as np

MINF = np.float32(-np.inf)


# ---------------------------------------------------------------------------------------------- PCD
def read_pcd(path):
    """Returns an (N,3) float32 array of x, y, z (NaN rows preserved, like loadPCDFile on a non-dense cloud)."""
    with open(path, "rb") as f:
        header = {}
        while True:
            line = f.readline()
            if not line:
                raise ValueError("PCD header without DATA line")
            s = line.decode("ascii", "replace").strip()
            if not s or s.startswith("#"):
                continue
            key, _, val = s.partition(" ")
            header[key.upper()] = val.split()
            if key.upper() == "DATA":
                break
        fields = header["FIELDS"]
        sizes = [int(v) for v in header["SIZE"]]
        types = header["TYPE"]
        counts = [int(v) for v in header.get("COUNT", ["1"] * len(fields))]
        npts = int(header["POINTS"][0]) if "POINTS" in header else int(header["WIDTH"][0]) * int(header["HEIGHT"][0])
        kind = header["DATA"][0].lower()
        if not all(k in fields for k in ("x", "y", "z")):
            raise ValueError("PCD file has no x/y/z fields")
        if kind == "ascii":
            data = np.loadtxt(f, dtype=np.float64, ndmin=2)
            cols = np.cumsum([0] + counts)
            idx = [int(cols[fields.index(k)]) for k in ("x", "y", "z")]
            return np.ascontiguousarray(data[:npts][:, idx], dtype=np.float32)
        if kind == "binary":
            np_types = {("F", 4): "<f4", ("F", 8): "<f8", ("U", 1): "u1", ("U", 2): "<u2", ("U", 4): "<u4", ("I", 1): "i1", ("I", 2): "<i2", ("I", 4): "<i4"}
            dt = np.dtype([(name, np_types[(t, s)], (c,)) if c > 1 else (name, np_types[(t, s)]) for name, t, s, c in zip(fields, types, sizes, counts)])
            rec = np.frombuffer(f.read(dt.itemsize * npts), dtype=dt, count=npts)
            return np.ascontiguousarray(np.stack([rec["x"], rec["y"], rec["z"]], 1), dtype=np.float32)
        raise ValueError("unsupported PCD DATA kind: %s (binary_compressed is not used by the ETH sets)" % kind)


def write_pcd(path, xyz, binary=True):
    xyz = np.ascontiguousarray(xyz, dtype=np.float32)
    head = ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\n"
            "WIDTH %d\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %d\nDATA %s\n" % (len(xyz), len(xyz), "binary" if binary else "ascii"))
    with open(path, "wb") as f:
        f.write(head.encode("ascii"))
        if binary:
            f.write(xyz.tobytes())
        else:
            for p in xyz:
                f.write(("%.9g %.9g %.9g\n" % (p[0], p[1], p[2])).encode("ascii"))


# ------------------------------------------------------------------------------------- ETH pose CSV
def read_pose_csv(path):
    """ETHDataLoader: list of dicts {id, source, target, pose (4x4 float32 from columns 4..15)}; first row is the header."""
    rows = []
    with open(path, "r") as f:
        lines = [ln.rstrip("\n").rstrip("\r") for ln in f]
    for ln in lines[1:]:
        if not ln:
            continue
        v = ln.split(",")                                          # CSVReader: split on the delimiter, no quoting
        T = np.eye(4, dtype=np.float32)
        T[:3, :] = np.array([np.float32(float(x)) for x in v[4:16]], dtype=np.float32).reshape(3, 4)   # ETHDataLoader.h:57-61
        rows.append(dict(id=v[0], source=v[1].strip(), target=v[2].strip(), pose=T))
    return rows


def write_pose_csv(path, rows):
    with open(path, "w") as f:
        f.write("id,source,target,overlap," + ",".join("T%d%d" % (r, c) for r in range(3) for c in range(4)) + "\n")
        for r in rows:
            T = np.asarray(r["pose"], np.float64)
            f.write("%s,%s,%s,%s,%s\n" % (r["id"], r["source"], r["target"], r.get("overlap", 1.0), ",".join("%.9g" % T[i, j] for i in range(3) for j in range(4))))


def scaled_initial_pose(pose, scale=0.1):
    """main.cpp:420-429: Euler angles (X, Y, Z) and translation of the benchmark perturbation scaled by pose_scaling."""
    R = np.asarray(pose, np.float64)[:3, :3]
    # Eigen eulerAngles(0,1,2): R = Rx(a) Ry(b) Rz(c)
    b = np.arcsin(np.clip(R[0, 2], -1, 1))
    a = np.arctan2(-R[1, 2], R[2, 2])
    c = np.arctan2(-R[0, 1], R[0, 0])
    from . import synth
    return synth.make_pose((scale * a, scale * b, scale * c), scale * np.asarray(pose, np.float64)[:3, 3])


# ------------------------------------------------------------------------------------------ TUM RGB-D
def read_tum_file_list(path):
    """readFileList (VirtualSensor.h:196-215): skips 3 header lines; returns (timestamps float64, filenames)."""
    ts, names = [], []
    with open(path, "r") as f:
        lines = f.read().split("\n")[3:]
    for ln in lines:
        tok = ln.split()
        if len(tok) < 2:
            continue
        ts.append(float(tok[0])); names.append(tok[1])
    return np.array(ts, np.float64), names


def quat_to_rot(qx, qy, qz, qw):
    """Eigen::Quaternionf::toRotationMatrix (no normalisation, as Eigen)."""
    tx, ty, tz = 2 * qx, 2 * qy, 2 * qz
    twx, twy, twz = tx * qw, ty * qw, tz * qw
    txx, txy, txz = tx * qx, ty * qx, tz * qx
    tyy, tyz, tzz = ty * qy, tz * qy, tz * qz
    return np.array([[1 - (tyy + tzz), txy - twz, txz + twy], [txy + twz, 1 - (txx + tzz), tyz - twx], [txz - twy, tyz + twx, 1 - (txx + tyy)]])


def read_tum_trajectory(path):
    """readTrajectoryFile (VirtualSensor.h:217-250): rows 't tx ty tz qx qy qz qw'; each pose is stored INVERTED."""
    ts, poses = [], []
    with open(path, "r") as f:
        lines = f.read().split("\n")[3:]
    for ln in lines:
        tok = ln.split()
        if len(tok) < 8:
            continue
        t, tx, ty, tz, qx, qy, qz, qw = [float(x) for x in tok[:8]]
        if qx == 0 and qy == 0 and qz == 0 and qw == 0:           # rot.norm() == 0 ends the file
            break
        T = np.eye(4)
        T[:3, :3] = quat_to_rot(qx, qy, qz, qw); T[:3, 3] = [tx, ty, tz]
        ts.append(t); poses.append(np.linalg.inv(T).astype(np.float32))
    return np.array(ts, np.float64), poses


def nearest_pose(trajectory_ts, poses, timestamp):
    """Nearest-timestamp lookup, first minimum (VirtualSensor.h:126-137)."""
    return poses[int(np.argmin(np.abs(np.asarray(trajectory_ts) - timestamp)))]


def decode_tum_depth(raw_u16):
    """VirtualSensor.h:119-124: metres = value / 5000, 0 -> MINF."""
    raw = np.asarray(raw_u16)
    out = raw.astype(np.float32) * np.float32(1.0) / np.float32(5000.0)
    out[raw == 0] = MINF
    return out


def load_tum_frame(base_dir, depth_name, rgb_name):
    """Depth (float32 metres, MINF holes) and RGBX bytes of one frame; needs PIL for the PNG decode."""
    from PIL import Image
    depth = decode_tum_depth(np.array(Image.open(os.path.join(base_dir, depth_name))))
    rgb = np.array(Image.open(os.path.join(base_dir, rgb_name)).convert("RGBA"), dtype=np.uint8)
    return depth, rgb.reshape(-1, 4)


# ------------------------------------------------------------------------------------------ OFF / PLY
def write_off(path, verts, colors=None, tris=None):
    """SimpleMesh::writeMesh (SimpleMesh.h:231-259): COFF, non-finite vertices become '0.0 0.0 0.0 0 0 0 0'."""
    verts = np.asarray(verts, np.float32)
    colors = np.zeros((len(verts), 4), np.uint8) if colors is None else np.asarray(colors, np.uint8)
    tris = np.zeros((0, 3), np.int32) if tris is None else np.asarray(tris, np.int32)
    with open(path, "w") as f:
        f.write("COFF\n%d %d 0\n" % (len(verts), len(tris)))
        for p, c in zip(verts, colors):
            if np.isfinite(p).all():
                f.write("%g %g %g %d %d %d %d\n" % (p[0], p[1], p[2], c[0], c[1], c[2], c[3]))
            else:
                f.write("0.0 0.0 0.0 0 0 0 0\n")
        for t in tris:
            f.write("3 %d %d %d\n" % (t[0], t[1], t[2]))


def write_ply(path, xyz, normals):
    """PointCloud::writeToFile (PointCloud.h:229-247): pcl::PointXYZINormal cloud, intensity 1, saved as ASCII PLY."""
    xyz = np.asarray(xyz, np.float32); normals = np.asarray(normals, np.float32)
    with open(path, "w") as f:
        f.write("ply\nformat ascii 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\nproperty float intensity\n"
                "property float normal_x\nproperty float normal_y\nproperty float normal_z\nproperty float curvature\nend_header\n" % len(xyz))
        for p, n in zip(xyz, normals):
            f.write("%g %g %g 1 %g %g %g 0\n" % (p[0], p[1], p[2], n[0], n[1], n[2]))
