"""ctypes front-end of the CPU ORACLE (oracle/icp_oracle.cpp).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never from the product path (icp-variants_amd/).
PARITY UNPINNED: see the header of icp_oracle.cpp.

Array conventions mirror the reference containers: points/normals are (N,3) float32 C-contiguous
(== std::vector<Eigen::Vector3f>), colours (N,4) uint8 (== std::vector<Vector4uc>), pose is a
(4,4) float32 numpy array in the usual row/col indexing; it is handed to C column-major, i.e.
exactly Eigen::Matrix4f::data().
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "icp_oracle.cpp")
_LIB = os.path.join(_HERE, "_build", "libicp_oracle.so")
_LIB_SAN = os.path.join(_HERE, "_build", "libicp_oracle_san.so")
# ICP_ORACLE_SANITIZE=1: load the -fsanitize=address,undefined build ("race detection / sanitizers").  The process
# must then run with LD_PRELOAD=<libasan.so> -- tests/test_sanitize.py starts such a child for tests/test_oracle.py.
SANITIZE = os.environ.get("ICP_ORACLE_SANITIZE", "0") == "1"

MATCH_DTYPE = np.dtype([("idx", np.int32), ("weight", np.float32)])


def build(force=False, sanitize=None):
    """Compile the oracle with the flags its arithmetic contract requires.  sanitize=True: the AddressSanitizer +
    UndefinedBehaviourSanitizer build (same arithmetic flags, -O1 -g, every UB report fatal)."""
    sanitize = SANITIZE if sanitize is None else sanitize
    out = _LIB_SAN if sanitize else _LIB
    if not force and os.path.exists(out) and (not os.path.exists(_SRC) or os.path.getmtime(out) >= os.path.getmtime(_SRC)):
        return out
    os.makedirs(os.path.dirname(out), exist_ok=True)
    opt = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"] if sanitize else ["-O2"]
    cmd = ["g++"] + opt + ["-std=c++14", "-ffp-contract=off", "-fopenmp", "-fPIC", "-shared", "-o", out, _SRC]
    subprocess.check_call(cmd)
    return out


def sanitizer_preload():
    """LD_PRELOAD value a (non-instrumented) Python needs before it can load the sanitized oracle."""
    return subprocess.check_output(["g++", "-print-file-name=libasan.so"]).decode().strip()


class Params(C.Structure):
    _fields_ = [("metric", C.c_int), ("matching", C.c_int), ("weighting", C.c_int), ("rejection", C.c_int),
                ("color_icp", C.c_int), ("multires", C.c_int), ("n_iterations", C.c_int), ("solver_mode", C.c_int),
                ("max_distance", C.c_float), ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("width", C.c_int), ("height", C.c_int), ("window", C.c_int), ("knn_kdtree", C.c_int), ("kdtree", C.c_void_p),
                ("selection", C.c_int), ("selection_proba", C.c_float), ("selection_seed", C.c_uint)]


class IterRecord(C.Structure):
    _fields_ = [("n_src", C.c_int), ("n_valid", C.c_int), ("pose", C.c_float * 16),
                ("seconds_match", C.c_double), ("seconds_rest", C.c_double)]


def make_params(metric=0, matching=0, weighting=0, rejection=1, color_icp=0, multires=0, n_iterations=20,
                solver_mode=0, max_distance=0.0003, K=None, width=0, height=0, window=12, knn_kdtree=0, selection=0, selection_proba=1.0, selection_seed=0):
    p = Params()
    p.metric, p.matching, p.weighting, p.rejection = metric, matching, weighting, rejection
    p.color_icp, p.multires, p.n_iterations, p.solver_mode = int(color_icp), int(multires), n_iterations, solver_mode
    p.max_distance = max_distance
    if K is not None:
        K = np.asarray(K, dtype=np.float32)
        p.fx, p.fy, p.cx, p.cy = float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2])
    p.width, p.height, p.window = width, height, window
    p.knn_kdtree = int(knn_kdtree); p.kdtree = None
    p.selection, p.selection_proba, p.selection_seed = int(selection), float(selection_proba), int(selection_seed)
    return p


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.orc_rmse.restype = C.c_float
        _lib.orc_select_hash.restype = C.c_uint
        _lib.orc_benchmark_error.restype = C.c_double
        _lib.orc_kdtree_build.restype = C.c_void_p
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _u8(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.uint8)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _pose_c(pose):
    """(4,4) row/col-indexed numpy -> 16 floats column-major (Eigen layout)."""
    return np.ascontiguousarray(np.asarray(pose, dtype=np.float32).T).reshape(16)


def _pose_np(buf16):
    return np.array(buf16, dtype=np.float32).reshape(4, 4).T.copy()


def normal_matrix(pose):
    out = np.empty(9, np.float32)
    lib().orc_normal_matrix(_p(_pose_c(pose)), _p(out))
    return out.reshape(3, 3)


def transform_points(pts, pose):
    pts = _f32(pts); out = np.empty_like(pts)
    lib().orc_transform_points(_p(pts), C.c_int(len(pts)), _p(_pose_c(pose)), _p(out))
    return out


def transform_normals(nrm, pose):
    nrm = _f32(nrm); out = np.empty_like(nrm)
    lib().orc_transform_normals(_p(nrm), C.c_int(len(nrm)), _p(_pose_c(pose)), _p(out))
    return out


def knn3(q, tgt, max_dist):
    q, tgt = _f32(q), _f32(tgt)
    out = np.empty(len(q), MATCH_DTYPE); d2 = np.empty(len(q), np.float32)
    lib().orc_knn3(_p(q), C.c_int(len(q)), _p(tgt), C.c_int(len(tgt)), C.c_float(max_dist), _p(out), _p(d2))
    return out, d2


class KdTree:
    """Exact kd-tree over a target cloud; query() is bit-identical to knn3() (see icp_oracle.cpp)."""

    def __init__(self, tgt):
        self.tgt = _f32(tgt)
        self.h = C.c_void_p(lib().orc_kdtree_build(_p(self.tgt), C.c_int(len(self.tgt))))

    def query(self, q, max_dist):
        q = _f32(q)
        out = np.empty(len(q), MATCH_DTYPE); d2 = np.empty(len(q), np.float32)
        lib().orc_kdtree_query(self.h, _p(q), C.c_int(len(q)), C.c_float(max_dist), _p(out), _p(d2))
        return out, d2

    def __del__(self):
        try:
            if self.h:
                lib().orc_kdtree_free(self.h); self.h = None
        except Exception:
            pass


def knn6(q, qrgba, tgt, trgba, max_dist):
    q, tgt, qrgba, trgba = _f32(q), _f32(tgt), _u8(qrgba), _u8(trgba)
    out = np.empty(len(q), MATCH_DTYPE); d2 = np.empty(len(q), np.float32)
    lib().orc_knn6(_p(q), _p(qrgba), C.c_int(len(q)), _p(tgt), _p(trgba), C.c_int(len(tgt)), C.c_float(max_dist), _p(out), _p(d2))
    return out, d2


def color_features(rgba):
    rgba = _u8(rgba); out = np.empty((len(rgba), 3), np.float32)
    lib().orc_color_features(_p(rgba), C.c_int(len(rgba)), _p(out))
    return out


def projective(q, tgt, width, height, K, max_dist, window=12):
    q, tgt = _f32(q), _f32(tgt)
    K = np.asarray(K, dtype=np.float32)
    out = np.empty(len(q), MATCH_DTYPE); d2 = np.empty(len(q), np.float32)
    lib().orc_projective(_p(q), C.c_int(len(q)), _p(tgt), C.c_int(width), C.c_int(height),
                         C.c_float(K[0, 0]), C.c_float(K[1, 1]), C.c_float(K[0, 2]), C.c_float(K[1, 2]),
                         C.c_float(max_dist), C.c_int(window), _p(out), _p(d2))
    return out, d2


def apply_weights(method, max_distance, sp, tp, sn, tn, sc, tc, matches):
    m = matches.copy()
    sp, tp, sn, tn = _f32(sp), _f32(tp), _f32(sn), _f32(tn)
    sc = _u8(sc) if sc is not None else np.zeros((len(sp), 4), np.uint8)
    tc = _u8(tc) if tc is not None else np.zeros((len(tp), 4), np.uint8)
    lib().orc_apply_weights(C.c_int(method), C.c_float(max_distance), _p(sp), _p(tp), _p(sn), _p(tn), _p(sc), _p(tc), _p(m), C.c_int(len(m)))
    return m


def prune(sn, tn, matches):
    m = matches.copy()
    sn, tn = _f32(sn), _f32(tn)
    lib().orc_prune(_p(sn), _p(tn), _p(m), C.c_int(len(m)))
    return m


def prune_predicate(c):
    return bool(lib().orc_prune_predicate(C.c_float(c)))


def compact(sp, sn, tp, tn, matches):
    sp, sn, tp, tn = _f32(sp), _f32(sn), _f32(tp), _f32(tn)
    n = len(sp)
    cs = np.empty((n, 3), np.float32); cd = np.empty((n, 3), np.float32); cw = np.empty(n, np.float32)
    cnt = np.empty((n, 3), np.float32); cns = np.empty((n, 3), np.float32)
    k = lib().orc_compact(_p(sp), _p(sn), _p(tp), _p(tn), _p(matches), C.c_int(n), _p(cs), _p(cd), _p(cw), _p(cnt), _p(cns))
    return cs[:k].copy(), cd[:k].copy(), cw[:k].copy(), cnt[:k].copy(), cns[:k].copy()


def solve_p2p(s, d, w, mode=0):
    s, d, w = _f32(s), _f32(d), _f32(w); pose = np.empty(16, np.float32)
    rc = lib().orc_solve_p2p(_p(s), _p(d), _p(w), C.c_int(len(s)), C.c_int(mode), _p(pose))
    if rc: raise RuntimeError("no correspondences")
    return _pose_np(pose)


def solve_p2plane(s, d, nt, w, mode=0):
    s, d, nt, w = _f32(s), _f32(d), _f32(nt), _f32(w); pose = np.empty(16, np.float32); x = np.empty(6, np.float64)
    rc = lib().orc_solve_p2plane(_p(s), _p(d), _p(nt), _p(w), C.c_int(len(s)), C.c_int(mode), _p(pose), _p(x))
    if rc: raise RuntimeError("no correspondences")
    return _pose_np(pose), x


def solve_symmetric(s, d, ns, nt, w, mode=0):
    s, d, ns, nt, w = _f32(s), _f32(d), _f32(ns), _f32(nt), _f32(w); pose = np.empty(16, np.float32); x = np.empty(6, np.float64)
    rc = lib().orc_solve_symmetric(_p(s), _p(d), _p(ns), _p(nt), _p(w), C.c_int(len(s)), C.c_int(mode), _p(pose), _p(x))
    if rc: raise RuntimeError("no correspondences")
    return _pose_np(pose), x


def rmse(src, ref, pose):
    src, ref = _f32(src), _f32(ref)
    return float(lib().orc_rmse(_p(src), _p(ref), C.c_int(len(src)), _p(_pose_c(pose))))


def benchmark_error(src, ref, pose):
    src, ref = _f32(src), _f32(ref)
    return float(lib().orc_benchmark_error(_p(src), _p(ref), C.c_int(len(src)), _p(_pose_c(pose))))


def select_hash(seed, iteration, index):
    return int(lib().orc_select_hash(C.c_uint(seed), C.c_uint(iteration), C.c_uint(index)))


def backproject(depth, rgbx, K, extrinsics=None, max_distance=0.1, fix_color_index=False):
    depth = np.ascontiguousarray(depth, dtype=np.float32); h, w = depth.shape
    K = np.asarray(K, dtype=np.float32)
    inv = np.empty(12, np.float32)
    lib().orc_invert_extrinsics(_p(_pose_c(np.eye(4) if extrinsics is None else extrinsics)), _p(inv))
    rgbx = _u8(rgbx)
    xyz = np.empty((h * w, 3), np.float32); nrm = np.empty((h * w, 3), np.float32)
    rgba = np.empty((h * w, 4), np.uint8) if rgbx is not None else None; valid = np.empty(h * w, np.uint8)
    lib().orc_backproject(_p(depth), _p(rgbx), C.c_int(w), C.c_int(h), C.c_float(K[0, 0]), C.c_float(K[1, 1]), C.c_float(K[0, 2]), C.c_float(K[1, 2]),
                          _p(inv), C.c_float(max_distance), C.c_int(int(fix_color_index)), _p(xyz), _p(nrm), _p(rgba), _p(valid))
    return xyz, nrm, rgba, valid.astype(bool)


def coarse(pts, nrm, rgba, factor):
    pts, nrm = _f32(pts), _f32(nrm); n = len(pts)
    op = np.empty((n, 3), np.float32); on = np.empty((n, 3), np.float32); oi = np.empty(n, np.int32)
    oc = np.empty((n, 4), np.uint8) if rgba is not None else None
    k = lib().orc_coarse(_p(pts), _p(nrm), _p(_u8(rgba)), C.c_int(n), C.c_int(factor), _p(op), _p(on), _p(oc), _p(oi))
    return op[:k].copy(), on[:k].copy(), (oc[:k].copy() if oc is not None else None), oi[:k].copy()


def iterate(prm, sp, sn, sc, tp, tn, tc, pose):
    """One ICP iteration (ICPOptimizer.h:553-621). Returns (new_pose, matches_after_prune, n_valid, t_match, t_rest)."""
    sp, sn, tp, tn = _f32(sp), _f32(sn), _f32(tp), _f32(tn)
    sc = _u8(sc) if sc is not None else np.zeros((len(sp), 4), np.uint8)
    tc = _u8(tc) if tc is not None else np.zeros((len(tp), 4), np.uint8)
    pc = _pose_c(pose).copy(); m = np.empty(len(sp), MATCH_DTYPE); nv = C.c_int(0); tm = C.c_double(0); tr = C.c_double(0)
    rc = lib().orc_iterate(C.byref(prm), _p(sp), _p(sn), _p(sc), C.c_int(len(sp)), _p(tp), _p(tn), _p(tc), C.c_int(len(tp)),
                           _p(pc), _p(m), C.byref(nv), C.byref(tm), C.byref(tr))
    if rc: raise RuntimeError("no correspondences")
    return _pose_np(pc), m, nv.value, tm.value, tr.value


def estimate_pose(prm, sp, sn, sc, tp, tn, tc, pose, max_records=256):
    """LinearICPOptimizer::estimatePose (ICPOptimizer.h:493-663). Returns (pose, list of per-iteration dicts)."""
    sp, sn, tp, tn = _f32(sp), _f32(sn), _f32(tp), _f32(tn)
    sc = _u8(sc) if sc is not None else np.zeros((len(sp), 4), np.uint8)
    tc = _u8(tc) if tc is not None else np.zeros((len(tp), 4), np.uint8)
    pc = _pose_c(pose).copy(); recs = (IterRecord * max_records)()
    n = lib().orc_estimate_pose(C.byref(prm), _p(sp), _p(sn), _p(sc), C.c_int(len(sp)), _p(tp), _p(tn), _p(tc), C.c_int(len(tp)),
                                _p(pc), recs, C.c_int(max_records))
    if n < 0: raise RuntimeError("no correspondences")
    out = [dict(n_src=recs[i].n_src, n_valid=recs[i].n_valid, pose=_pose_np(recs[i].pose),
                seconds_match=recs[i].seconds_match, seconds_rest=recs[i].seconds_rest) for i in range(min(n, max_records))]
    return _pose_np(pc), out


def num_threads():
    return int(lib().orc_num_threads())


def set_num_threads(t):
    lib().orc_set_num_threads(C.c_int(t))
