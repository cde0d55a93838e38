"""OFF/COFF mesh reader and mesh -> point-cloud conversion (host-side data model).

Follows SimpleMesh::loadMesh (reference SimpleMesh.h:161-229) and PointCloud(const SimpleMesh&)
(PointCloud.h:12-39): vertex normals are the normalised fp32 sum of un-normalised face normals,
accumulated in file order; colours of mesh-derived clouds are all zero (PointCloud.h:26).
Pre-processing outside the timed ICP loop -- plain numpy, fp32.
"""
import numpy as np


def load_off(path):
    """Returns (vertices (V,3) f32, colors (V,4) u8, triangles (T,3) i32)."""
    with open(path, "r") as f:
        tok = f.read().split()
    kind = tok[0]
    if kind not in ("OFF", "COFF"):
        raise ValueError("Incorrect mesh file type.")            # SimpleMesh.h:210-212
    nv, nt = int(tok[1]), int(tok[2])
    pos = 4
    stride = 7 if kind == "COFF" else 3
    body = tok[pos:pos + nv * stride]
    verts = np.array([np.float32(float(body[i * stride + k])) for i in range(nv) for k in range(3)], dtype=np.float32).reshape(nv, 3)
    if kind == "COFF":
        cols = np.array([int(body[i * stride + 3 + k]) & 0xFF for i in range(nv) for k in range(4)], dtype=np.uint8).reshape(nv, 4)
    else:
        cols = np.tile(np.array([0, 0, 0, 255], np.uint8), (nv, 1))
    pos += nv * stride
    tris = np.empty((nt, 3), np.int32)
    for i in range(nt):
        if int(tok[pos]) != 3:
            raise ValueError("We can only read triangular mesh.")   # SimpleMesh.h:220
        tris[i] = [int(tok[pos + 1]), int(tok[pos + 2]), int(tok[pos + 3])]
        pos += 4
    return verts, cols, tris


def mesh_to_cloud(verts, tris):
    """PointCloud(const SimpleMesh&): returns (points, normals, colors) with colours all zero."""
    f32 = np.float32
    pts = verts.astype(np.float32).copy()
    nrm = np.zeros_like(pts)
    for t in tris:
        a = pts[t[1]] - pts[t[0]]
        b = pts[t[2]] - pts[t[0]]
        fn = np.array([f32(a[1] * b[2]) - f32(a[2] * b[1]), f32(a[2] * b[0]) - f32(a[0] * b[2]), f32(a[0] * b[1]) - f32(a[1] * b[0])], dtype=np.float32)
        nrm[t[0]] += fn; nrm[t[1]] += fn; nrm[t[2]] += fn
    for i in range(len(nrm)):
        v = nrm[i]
        z = f32(v[0] * v[0]) + (f32(v[1] * v[1]) + f32(v[2] * v[2]))     # Eigen squaredNorm tree
        if z > 0:
            nrm[i] = v / np.sqrt(z, dtype=np.float32)
    cols = np.zeros((len(pts), 4), np.uint8)
    return pts, nrm, cols
