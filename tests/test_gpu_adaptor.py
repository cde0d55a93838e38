"""C++14 adaptor classes (include/icp_hip_adaptor.hpp) driven like the reference's main.cpp, on the GPU."""
import os
import subprocess
import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
SRC = os.path.join(ROOT, "tests", "cpp", "bunny_adaptor.cpp")
LIBDIR = os.path.join(ROOT, "icp-variants_amd", "lib")


def build_driver(tmp):
    exe = os.path.join(tmp, "bunny_adaptor")
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), SRC, "-o", exe,
                           "-L", LIBDIR, "-licp_hip", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_adaptor_header_compiles_as_cxx14(tmp_path):
    """CPU-only: the header is valid C++14 and links against the C ABI (no GPU call is made)."""
    exe = build_driver(str(tmp_path))
    assert os.path.exists(exe)


@pytest.mark.gpu
@pytest.mark.parametrize("metric,multires", [(0, 0), (1, 0), (2, 0), (1, 1)])
def test_bunny_through_cxx_adaptor(tmp_path, bunny, bunny_oracle, metric, multires):
    from conftest import pose_error
    exe = build_driver(str(tmp_path))
    dump = os.path.join(str(tmp_path), "bunny.bin")
    with open(dump, "wb") as f:
        for k in ("src", "tgt"):
            n = len(bunny[k + "_pts"])
            f.write(np.int32(n).tobytes()); f.write(bunny[k + "_pts"].astype(np.float32).tobytes())
            f.write(bunny[k + "_nrm"].astype(np.float32).tobytes()); f.write(bunny[k + "_rgba"].astype(np.uint8).tobytes())
    out = subprocess.check_output([exe, dump, str(metric), str(multires)]).decode().splitlines()
    status = dict(zip(out[-2].split()[0::2], out[-2].split()[1::2]))
    assert status["status"] == "0" and status["iterations"] == "20" and status["recorded"] == "20"
    assert status["valid_at_identity"] == "576" and status["mismatch_status"] == "7" and status["empty"] == "1" and status["time_ok"] == "1"
    pose = np.array([float(v) for v in out[-1].split()[1:]], np.float64).reshape(4, 4)
    gp = bunny_oracle["m%d_w0_r%d_mode1_poses" % (metric, multires)][-1]
    ang, tr = pose_error(pose, gp)
    assert ang < 1e-5 and tr < 1e-5


BATCH_SRC = os.path.join(ROOT, "tests", "cpp", "batch_driver.cpp")


def build_batch_driver(tmp):
    exe = os.path.join(tmp, "batch_driver")
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), BATCH_SRC, "-o", exe,
                           "-L", LIBDIR, "-licp_hip", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_batch_driver_compiles_as_cxx14(tmp_path):
    assert os.path.exists(build_batch_driver(str(tmp_path)))


@pytest.mark.gpu
def test_batch_and_gather_from_a_cxx14_host(tmp_path, orc):
    """configs[3] without Python in the loop: icp_batch_run with 3 contexts + icp_gather_poses (RCCL, one rank) from C++14;
    every pose within 1e-5 of the oracle's estimatePose for that pair, in pair order (main.cpp:411-498)."""
    from conftest import pose_error
    from icp_amd import synth
    pairs = [synth.eth_like_pair(k, n_tilt=30, n_beam=100) for k in range(5)]
    dump = os.path.join(str(tmp_path), "pairs.bin")
    with open(dump, "wb") as f:
        f.write(np.int32(len(pairs)).tobytes())
        for d in pairs:
            for k in ("src", "tgt"):
                f.write(np.int32(len(d[k + "_pts"])).tobytes()); f.write(d[k + "_pts"].astype(np.float32).tobytes()); f.write(d[k + "_nrm"].astype(np.float32).tobytes())
    out = subprocess.check_output([build_batch_driver(str(tmp_path)), dump, "3", "25"]).decode().splitlines()
    out = [ln for ln in out if ln.startswith(("batch rc", "gather rc", "pose "))]        # (RCCL prints its version banner on stdout)
    assert out[0] == "batch rc 0" and out[1].startswith("gather rc 0")
    for p, d in enumerate(pairs):
        tok = out[2 + p].split()
        assert tok[:4] == ["pose", str(p), "status", "0"]
        P = np.array([float(v) for v in tok[4:]], np.float64).reshape(4, 4).T          # column-major on the wire
        kd = orc.KdTree(d["tgt_pts"])
        prm = orc.make_params(metric=1, n_iterations=25, max_distance=10.0, solver_mode=1, knn_kdtree=1); prm.kdtree = kd.h
        po, _ = orc.estimate_pose(prm, d["src_pts"], d["src_nrm"], None, d["tgt_pts"], d["tgt_nrm"], None, np.eye(4, dtype=np.float32))
        ang, tr = pose_error(P, po)
        assert ang < 1e-5 and tr < 1e-5, (p, ang, tr)
