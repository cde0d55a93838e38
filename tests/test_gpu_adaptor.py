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
