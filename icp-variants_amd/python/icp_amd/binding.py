"""ctypes binding of the C ABI in include/icp_hip.h (libicp_hip.so) plus a thin optimizer facade.

This is harness code for tests/ and bench.py; the product is the shared library.  It fails loudly
(ImportError / IcpError) when the HIP library is missing or no GPU is usable -- there is no CPU path.

`LinearICPOptimizer` keeps the setter names of the reference's ICPOptimizer (ICPOptimizer.h:41-95)
so the parity tests read like the reference's drivers (main.cpp:43-181, 343-514).
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
PKG_ROOT = os.path.abspath(os.path.join(_HERE, "..", ".."))            # icp-variants_amd/
LIB_PATH = os.environ.get("ICP_HIP_LIB") or os.path.join(PKG_ROOT, "lib", "libicp_hip.so")     # ICP_HIP_LIB: development A/B builds

MATCH_DTYPE = np.dtype([("idx", np.int32), ("weight", np.float32)])

ICP_OK = 0
ERR_NAMES = {1: "INVALID_ARG", 2: "HIP", 3: "NO_TARGET", 4: "NO_SOURCE", 5: "NO_CAMERA", 6: "TARGET_SIZE",
             7: "COLOR_MISMATCH", 8: "NO_CORRESPONDENCES", 9: "NO_DEVICE", 10: "COMM"}
ERR_NO_CORRESPONDENCES = 8


class IcpError(RuntimeError):
    def __init__(self, code, msg=""):
        super().__init__("icp_hip error %d (%s): %s" % (code, ERR_NAMES.get(code, "?"), msg))
        self.code = code


class IcpParams(C.Structure):
    _fields_ = [("metric", C.c_int32), ("matching", C.c_int32), ("weighting", C.c_int32), ("rejection", C.c_int32),
                ("color_icp", C.c_int32), ("multires", C.c_int32), ("n_iterations", C.c_int32), ("max_distance", C.c_float),
                ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("width", C.c_int32), ("height", C.c_int32), ("knn_backend", C.c_int32),
                ("selection", C.c_int32), ("selection_proba", C.c_float), ("selection_seed", C.c_uint32),
                ("knn_incremental", C.c_int32), ("record_rmse", C.c_int32)]


class IcpIterStats(C.Structure):
    _fields_ = [("n_src", C.c_int32), ("n_valid", C.c_int32), ("pose", C.c_float * 16), ("rmse", C.c_float), ("benchmark_error", C.c_float),
                ("status", C.c_int32)]


class IcpTiming(C.Structure):
    _fields_ = [("match_ms", C.c_double), ("weight_reject_build_ms", C.c_double), ("solve_ms", C.c_double),
                ("total_ms", C.c_double), ("iterations", C.c_int32), ("sampled_iterations", C.c_int32)]


class IcpPair(C.Structure):
    _fields_ = [("src_xyz", C.c_void_p), ("src_normals", C.c_void_p), ("src_rgba", C.c_void_p), ("n_src", C.c_int32),
                ("tgt_xyz", C.c_void_p), ("tgt_normals", C.c_void_p), ("tgt_rgba", C.c_void_p), ("n_tgt", C.c_int32),
                ("initial_pose", C.c_float * 16)]


COMM_ID_BYTES = 128

# every symbol include/icp_hip.h declares (tests check the library exports all of them)
EXPORTS = ["icp_ctx_create", "icp_ctx_create_on_stream", "icp_ctx_destroy", "icp_last_error", "icp_params_default",
           "icp_set_params", "icp_get_params", "icp_set_target", "icp_set_source", "icp_query_matches", "icp_match", "icp_match_seeded",
           "icp_correspond", "icp_iterate", "icp_run", "icp_get_timing", "icp_get_iteration_times", "icp_set_stage_timing", "icp_set_convergence_reference", "icp_rmse", "icp_benchmark_error",
           "icp_transform_points", "icp_transform_normals", "icp_version", "icp_schedule", "icp_select_hash", "icp_backproject_depth", "icp_estimate_normals",
           "icp_batch_run", "icp_pair_owner", "icp_pairs_of_rank", "icp_comm_unique_id", "icp_comm_create", "icp_comm_destroy", "icp_gather_poses",
           "icp_comm_last_error"]

_lib = None


def load_library():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libicp_hip.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` (%s)" % LIB_PATH)
        _lib = C.CDLL(LIB_PATH)
        _lib.icp_last_error.restype = C.c_char_p
        _lib.icp_version.restype = C.c_char_p
        _lib.icp_comm_last_error.restype = C.c_char_p
        _lib.icp_pair_owner.restype = C.c_int32
        _lib.icp_pairs_of_rank.restype = C.c_int32
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def pose_to_c(pose):
    """(4,4) numpy (row, col) -> 16 floats column-major == Eigen::Matrix4f::data()."""
    return np.ascontiguousarray(np.asarray(pose, dtype=np.float32).T).reshape(16).copy()


def pose_from_c(buf):
    return np.array(buf, dtype=np.float32).reshape(4, 4).T.copy()


def schedule(params, n_src, max_out=4096):
    """icp_schedule: decimation factor per iteration (host logic only, needs no GPU)."""
    lib = load_library()
    buf = (C.c_int32 * max_out)(); cnt = C.c_int32(0)
    rc = lib.icp_schedule(C.byref(params), C.c_int32(n_src), buf, C.c_int32(max_out), C.byref(cnt))
    if rc != ICP_OK:
        raise IcpError(rc, "icp_schedule")
    return [buf[i] for i in range(min(cnt.value, max_out))]


def select_hash(seed, iteration, index):
    lib = load_library(); lib.icp_select_hash.restype = C.c_uint32
    return int(lib.icp_select_hash(C.c_uint32(seed), C.c_uint32(iteration), C.c_uint32(index)))


def default_params():
    p = IcpParams()
    load_library().icp_params_default(C.byref(p))
    return p


class Context:
    """RAII wrapper of icp_ctx."""

    def __init__(self, device=0, stream=None):
        self.lib = load_library()
        self.h = C.c_void_p()
        rc = self.lib.icp_ctx_create_on_stream(C.c_int(device), C.c_void_p(stream) if stream else None, C.byref(self.h))
        if rc != ICP_OK:
            self.h = None
            raise IcpError(rc, "icp_ctx_create failed (is a HIP device visible?)")
        self.params = IcpParams()
        self.lib.icp_params_default(C.byref(self.params))

    def close(self):
        if getattr(self, "h", None):
            self.lib.icp_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != ICP_OK:
            raise IcpError(rc, self.lib.icp_last_error(self.h).decode())

    def push_params(self):
        self._ck(self.lib.icp_set_params(self.h, C.byref(self.params)))

    def set_target(self, xyz, normals=None, rgba=None):
        xyz = _f32(xyz); normals = None if normals is None else _f32(normals)
        rgba = None if rgba is None else np.ascontiguousarray(rgba, dtype=np.uint8)
        self._ck(self.lib.icp_set_target(self.h, _ptr(xyz), _ptr(normals), _ptr(rgba), C.c_int32(len(xyz))))
        self.n_tgt = len(xyz)

    def set_source(self, xyz, normals=None, rgba=None):
        xyz = _f32(xyz); normals = None if normals is None else _f32(normals)
        rgba = None if rgba is None else np.ascontiguousarray(rgba, dtype=np.uint8)
        self._ck(self.lib.icp_set_source(self.h, _ptr(xyz), _ptr(normals), _ptr(rgba), C.c_int32(len(xyz))))
        self.n_src = len(xyz)

    def query_matches(self, transformed_xyz, rgba=None):
        q = _f32(transformed_xyz); rgba = None if rgba is None else np.ascontiguousarray(rgba, dtype=np.uint8)
        out = np.empty(len(q), MATCH_DTYPE)
        self._ck(self.lib.icp_query_matches(self.h, _ptr(q), _ptr(rgba), C.c_int32(len(q)), _ptr(out)))
        return out

    def match(self, pose):
        out = np.empty(self.n_src, MATCH_DTYPE); d2 = np.empty(self.n_src, np.float32)
        self._ck(self.lib.icp_match(self.h, _ptr(pose_to_c(pose)), _ptr(out), _ptr(d2)))
        return out, d2

    def match_seeded(self, poses):
        """icp_match_seeded: the fused matcher launched once per pose (first unseeded, then seeded + incremental as in the loop);
        returns the last launch's (Match records after weighting / rejection, squared distances) in source order."""
        ps = np.ascontiguousarray(np.stack([pose_to_c(p) for p in poses]), dtype=np.float32)
        out = np.empty(self.n_src, MATCH_DTYPE); d2 = np.empty(self.n_src, np.float32)
        self._ck(self.lib.icp_match_seeded(self.h, _ptr(ps), C.c_int32(len(ps)), _ptr(out), _ptr(d2)))
        return out, d2

    def correspond(self, pose):
        out = np.empty(self.n_src, MATCH_DTYPE); sums = np.zeros(64, np.float64); nv = C.c_int32(0)
        self._ck(self.lib.icp_correspond(self.h, _ptr(pose_to_c(pose)), _ptr(out), _ptr(sums), C.byref(nv)))
        return out, sums, nv.value

    def iterate(self, pose):
        p = pose_to_c(pose); st = IcpIterStats()
        self._ck(self.lib.icp_iterate(self.h, _ptr(p), C.byref(st)))
        return pose_from_c(p), dict(n_src=st.n_src, n_valid=st.n_valid, pose=pose_from_c(st.pose), rmse=st.rmse, benchmark_error=st.benchmark_error, status=st.status)

    def run(self, pose, max_stats=512, check=True):
        p = pose_to_c(pose); st = (IcpIterStats * max_stats)(); n = C.c_int32(0)
        rc = self.lib.icp_run(self.h, _ptr(p), st, C.c_int32(max_stats), C.byref(n))
        if check:
            self._ck(rc)
        recs = [dict(n_src=st[i].n_src, n_valid=st[i].n_valid, pose=pose_from_c(st[i].pose), rmse=st[i].rmse, benchmark_error=st[i].benchmark_error, status=st[i].status)
                for i in range(min(n.value, max_stats))]
        return pose_from_c(p), recs, rc

    def run_raw(self, pose_c16):
        """Timed-loop entry for bench.py: pose buffer in/out (column-major float32[16]), no record marshalling."""
        n = C.c_int32(0)
        self._ck(self.lib.icp_run(self.h, _ptr(pose_c16), None, C.c_int32(0), C.byref(n)))
        return n.value

    def timing(self):
        t = IcpTiming()
        self._ck(self.lib.icp_get_timing(self.h, C.byref(t)))
        return dict(match_ms=t.match_ms, weight_reject_build_ms=t.weight_reject_build_ms, solve_ms=t.solve_ms,
                    total_ms=t.total_ms, iterations=t.iterations, sampled_iterations=t.sampled_iterations)

    def iteration_times(self, max_out=4096):
        """Per-iteration (match, weight/reject/build, solve) device milliseconds of the last run; -1 = iteration not bracketed."""
        a = np.empty(max_out, np.float32); b = np.empty(max_out, np.float32); d = np.empty(max_out, np.float32); n = C.c_int32(0)
        self._ck(self.lib.icp_get_iteration_times(self.h, _ptr(a), _ptr(b), _ptr(d), C.c_int32(max_out), C.byref(n)))
        k = min(n.value, max_out)
        return a[:k].copy(), b[:k].copy(), d[:k].copy()

    def set_stage_timing(self, every_nth):
        """0: whole-run time only; 1: HIP events around every iteration's stages (default); N > 1: every Nth iteration, scaled."""
        self._ck(self.lib.icp_set_stage_timing(self.h, C.c_int32(int(every_nth))))

    def set_convergence_reference(self, src_xyz, ref_xyz):
        s, r = _f32(src_xyz), _f32(ref_xyz)
        self._ck(self.lib.icp_set_convergence_reference(self.h, _ptr(s), _ptr(r), C.c_int32(len(s))))

    def rmse(self, pose):
        out = C.c_float(0)
        self._ck(self.lib.icp_rmse(self.h, _ptr(pose_to_c(pose)), C.byref(out)))
        return out.value

    def benchmark_error(self, pose):
        out = C.c_float(0)
        self._ck(self.lib.icp_benchmark_error(self.h, _ptr(pose_to_c(pose)), C.byref(out)))
        return out.value

    def backproject_depth(self, depth, rgbx, K, extrinsics=None, max_distance=0.1, fix_color_index=False):
        """PointCloud(depthMap, colorFrame, K, extrinsics, w, h, keepOriginalSize=true) on the device (PointCloud.h:78-165)."""
        depth = np.ascontiguousarray(depth, dtype=np.float32); h, w = depth.shape
        K = np.asarray(K, dtype=np.float32); E = pose_to_c(np.eye(4) if extrinsics is None else extrinsics)
        rgbx = None if rgbx is None else np.ascontiguousarray(rgbx, dtype=np.uint8)
        xyz = np.empty((h * w, 3), np.float32); nrm = np.empty((h * w, 3), np.float32)
        rgba = np.empty((h * w, 4), np.uint8) if rgbx is not None else None; valid = np.empty(h * w, np.uint8)
        self._ck(self.lib.icp_backproject_depth(self.h, _ptr(depth), _ptr(rgbx), C.c_float(K[0, 0]), C.c_float(K[1, 1]), C.c_float(K[0, 2]), C.c_float(K[1, 2]),
                                                _ptr(E), C.c_int32(w), C.c_int32(h), C.c_float(max_distance), C.c_int32(int(fix_color_index)),
                                                _ptr(xyz), _ptr(nrm), _ptr(rgba), _ptr(valid)))
        return xyz, nrm, rgba, valid.astype(bool)

    def estimate_normals(self, xyz, k=5, viewpoint=(0.0, 0.0, 0.0)):
        """PointCloud(pcl cloud): k-NN PCA normals flipped towards the viewpoint (PointCloud.h:41-76)."""
        x = _f32(xyz); vp = np.asarray(viewpoint, np.float32)
        nrm = np.empty_like(x); curv = np.empty(len(x), np.float32)
        self._ck(self.lib.icp_estimate_normals(self.h, _ptr(x), C.c_int32(len(x)), C.c_int32(k), _ptr(vp), _ptr(nrm), _ptr(curv)))
        return nrm, curv

    def transform_points(self, xyz, pose):
        x = _f32(xyz); out = np.empty_like(x)
        self._ck(self.lib.icp_transform_points(self.h, _ptr(x), C.c_int32(len(x)), _ptr(pose_to_c(pose)), _ptr(out)))
        return out

    def transform_normals(self, nrm, pose):
        x = _f32(nrm); out = np.empty_like(x)
        self._ck(self.lib.icp_transform_normals(self.h, _ptr(x), C.c_int32(len(x)), _ptr(pose_to_c(pose)), _ptr(out)))
        return out


def pair_owner(pair, n_ranks):
    return int(load_library().icp_pair_owner(C.c_int32(pair), C.c_int32(n_ranks)))


def pairs_of_rank(n_pairs, rank, n_ranks):
    return int(load_library().icp_pairs_of_rank(C.c_int32(n_pairs), C.c_int32(rank), C.c_int32(n_ranks)))


def batch_run(contexts, pairs, initial_poses=None):
    """icp_batch_run: aligns `pairs` (dicts with src_pts/src_nrm[/src_rgba]/tgt_pts/tgt_nrm[/tgt_rgba]) on the given contexts of
    one device, one host thread per context inside the library.  Returns ((n,16) float32 column-major poses in pair order,
    per-pair status codes, overall status)."""
    lib = load_library()
    n = len(pairs)
    keep = []                                                            # keeps the converted arrays alive for the call
    arr = (IcpPair * max(n, 1))()
    for i, d in enumerate(pairs):
        def cv(key, dt):
            a = d.get(key)
            if a is None:
                return None
            a = np.ascontiguousarray(a, dtype=dt); keep.append(a)
            return a.ctypes.data
        arr[i].src_xyz = cv("src_pts", np.float32); arr[i].src_normals = cv("src_nrm", np.float32); arr[i].src_rgba = cv("src_rgba", np.uint8)
        arr[i].tgt_xyz = cv("tgt_pts", np.float32); arr[i].tgt_normals = cv("tgt_nrm", np.float32); arr[i].tgt_rgba = cv("tgt_rgba", np.uint8)
        arr[i].n_src = len(d["src_pts"]); arr[i].n_tgt = len(d["tgt_pts"])
        p0 = pose_to_c(np.eye(4) if initial_poses is None else initial_poses[i])
        for k in range(16):
            arr[i].initial_pose[k] = float(p0[k])
    hs = (C.c_void_p * len(contexts))(*[c.h for c in contexts])
    poses = np.zeros((n, 16), np.float32); status = np.zeros(n, np.int32)
    rc = lib.icp_batch_run(hs, C.c_int32(len(contexts)), arr, C.c_int32(n), _ptr(poses), _ptr(status))
    return poses, status, rc


class Comm:
    """icp_comm: one RCCL communicator per process / GPU; the unique id travels through the host application."""

    def __init__(self, device, n_ranks, rank, unique_id):
        self.lib = load_library()
        self.h = C.c_void_p()
        self.n_ranks, self.rank = n_ranks, rank
        idb = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(bytes(unique_id))
        rc = self.lib.icp_comm_create(C.c_int(device), C.c_int32(n_ranks), C.c_int32(rank), idb, C.byref(self.h))
        if rc != ICP_OK:
            self.h = None
            raise IcpError(rc, self.lib.icp_comm_last_error().decode())

    @staticmethod
    def unique_id():
        lib = load_library()
        buf = (C.c_uint8 * COMM_ID_BYTES)()
        rc = lib.icp_comm_unique_id(buf)
        if rc != ICP_OK:
            raise IcpError(rc, lib.icp_comm_last_error().decode())
        return bytes(buf)

    def gather_poses(self, local_poses, n_pairs):
        lp = np.ascontiguousarray(np.asarray(local_poses, np.float32).reshape(-1, 16))
        out = np.zeros((n_pairs, 16), np.float32)
        rc = self.lib.icp_gather_poses(self.h, _ptr(lp) if len(lp) else None, C.c_int32(len(lp)), C.c_int32(n_pairs), _ptr(out))
        if rc != ICP_OK:
            raise IcpError(rc, self.lib.icp_comm_last_error().decode())
        return out

    def close(self):
        if getattr(self, "h", None):
            self.lib.icp_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class LinearICPOptimizer:
    """Python mirror of the reference's LinearICPOptimizer setter surface (ICPOptimizer.h:41-95, 489-663)."""

    def __init__(self, device=0, stream=None):
        self.ctx = Context(device, stream)

    # -- setters, same names / meaning as the reference --
    def setMatchingMaxDistance(self, d): self.ctx.params.max_distance = d                      # ICPOptimizer.h:41-44
    def setMetric(self, m): self.ctx.params.metric = m                                         # :46-48
    def enableMultiResolution(self, on): self.ctx.params.multires = int(bool(on))              # :50-52
    def enableColorICP(self, on): self.ctx.params.color_icp = int(bool(on))                    # :54-56
    def setRejectionMethod(self, r): self.ctx.params.rejection = r                             # :63-65
    def setWeightingMethod(self, w): self.ctx.params.weighting = w                             # :67-69
    def setMatchingMethod(self, m): self.ctx.params.matching = m                               # :71-78
    def setNbOfIterations(self, n): self.ctx.params.n_iterations = n                           # :84-86
    def setKnnBackend(self, b): self.ctx.params.knn_backend = b

    def setSelectionMethod(self, method, proba=1.0, seed=0):                                   # :58-61 (+ explicit seed)
        self.ctx.params.selection = int(method); self.ctx.params.selection_proba = float(proba); self.ctx.params.selection_seed = int(seed)

    def setCameraParamsMatchingMethod(self, K, width, height):                                 # :80-82
        K = np.asarray(K, dtype=np.float32)
        p = self.ctx.params
        p.fx, p.fy, p.cx, p.cy, p.width, p.height = float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]), int(width), int(height)

    def setConvergenceMeasure(self, src_xyz, ref_xyz, runBenchmark=False):                     # :93-95, ConvergenceMeasure.h:35-44
        self.ctx.set_convergence_reference(src_xyz, ref_xyz)
        self.ctx.params.record_rmse = 3 if runBenchmark else 1

    def estimatePose(self, source, target, initialPose, check=True):
        """source/target: dicts with 'pts', 'nrm', optional 'rgba'.  Returns (pose, per-iteration records)."""
        self.ctx.push_params()
        self.ctx.set_target(target["pts"], target["nrm"], target.get("rgba"))
        self.ctx.set_source(source["pts"], source["nrm"], source.get("rgba"))
        pose, recs, rc = self.ctx.run(initialPose, check=check)
        self.last_status = rc
        return pose, recs
