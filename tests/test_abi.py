"""The C-ABI library loads on a CPU-only box and exports every symbol include/icp_hip.h declares (no compute calls)."""
import ctypes
import os
import re
import subprocess
import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def header_functions():
    txt = open(os.path.join(ROOT, "include", "icp_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"^\s*(?:int|int32_t|uint32_t|const char\s*\*)\s+(icp_\w+)\s*\(", txt, flags=re.M)))


def test_header_and_binding_agree():
    from icp_amd import binding
    assert header_functions() == sorted(binding.EXPORTS)


def test_library_exports_every_declared_symbol():
    from icp_amd import binding
    lib = binding.load_library()
    for name in header_functions():
        assert hasattr(lib, name), name
    out = subprocess.check_output(["nm", "-D", "--defined-only", binding.LIB_PATH]).decode()
    exported = set(re.findall(r" T (icp_\w+)", out))
    assert set(header_functions()) <= exported
    assert lib.icp_version().decode().startswith("icp_hip gfx950")


def test_library_contains_gfx950_code_object():
    from icp_amd import binding
    blob = open(binding.LIB_PATH, "rb").read()
    assert b"gfx950" in blob and b"k_knn_brute" in blob


def test_params_default_match_reference_ctor():
    """ICPOptimizer ctor defaults, ICPOptimizer.h:29-31: metric 0, rejection 1, weighting 0, 20 iterations, 0.0003."""
    from icp_amd import binding
    p = binding.default_params()
    assert (p.metric, p.matching, p.weighting, p.rejection, p.color_icp, p.multires, p.n_iterations) == (0, 0, 0, 1, 0, 0, 20)
    assert np.float32(p.max_distance) == np.float32(0.0003)


def test_struct_layouts():
    from icp_amd import binding
    assert ctypes.sizeof(binding.IcpParams) == 20 * 4
    assert ctypes.sizeof(binding.IcpIterStats) == 4 + 4 + 64 + 4 + 4 + 4
    assert binding.MATCH_DTYPE.itemsize == 8            # struct Match, NearestNeighbor.h:7-10


def test_no_cpu_fallback_without_device():
    """Without a usable HIP device context creation fails loudly (ICP_ERR_NO_DEVICE) -- there is no CPU path."""
    from icp_amd import binding
    lib = binding.load_library()
    h = ctypes.c_void_p()
    rc = lib.icp_ctx_create(0, ctypes.byref(h))
    if rc == 0:                                         # running on a GPU box: fine, just clean up
        lib.icp_ctx_destroy(h)
    else:
        assert rc in (2, 9) and not h.value
        with pytest.raises(binding.IcpError):
            binding.Context(0)


def test_product_never_imports_oracle():
    """The shipped path (icp-variants_amd/, include/) must not reference the oracle."""
    pkg = os.path.join(ROOT, "icp-variants_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(base, f), errors="ignore").read()
                assert "import oracle" not in txt and "from oracle" not in txt and "libicp_oracle" not in txt, f


def test_missing_rccl_is_an_error_code_not_a_crash():
    """A host without librccl: icp_comm_unique_id / icp_comm_create return ICP_ERR_COMM with a message (the header promises a single-GPU
    host needs no RCCL at all).  Forced here through ICP_HIP_RCCL_LIB, in a child process (the library caches what it loaded)."""
    import sys
    code = (
        "import ctypes, os, sys\n"
        "sys.path.insert(0, %r)\n"
        "from icp_amd import binding\n"
        "lib = binding.load_library()\n"
        "buf = (ctypes.c_uint8 * 128)()\n"
        "rc = lib.icp_comm_unique_id(buf)\n"
        "msg = lib.icp_comm_last_error().decode()\n"
        "h = ctypes.c_void_p()\n"
        "rc2 = lib.icp_comm_create(0, 1, 0, buf, ctypes.byref(h))\n"
        "print(rc, rc2, msg)\n" % os.path.join(ROOT, "icp-variants_amd", "python"))
    env = dict(os.environ, ICP_HIP_RCCL_LIB="/nonexistent/librccl.so.1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    rc, rc2, msg = out.stdout.strip().split(" ", 2)
    assert int(rc) == 10                                    # ICP_ERR_COMM
    assert int(rc2) in (9, 10)                              # no HIP device on a CPU box (ICP_ERR_NO_DEVICE), ICP_ERR_COMM on a GPU box
    assert "ICP_HIP_RCCL_LIB" in msg and "/nonexistent/librccl.so.1" in msg


def test_fused_matcher_wave_mapping_is_a_permutation():
    """fused_wave_slot (dev_fused.hpp): the waves of a block come from several places of the query order (the cross-wave hand-over pairs
    hard and easy regions).  Whatever the stride and the grid, every stretch of 64 queries must belong to exactly one (block, wave) --
    a hole would silently drop queries, an overlap would count them twice.  Evaluated on the host through a debug hook: no GPU."""
    from icp_amd import binding
    lib = binding.load_library()
    lib.icp_debug_wave_slot.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(ctypes.c_int32)]
    lib.icp_debug_wave_slot.restype = ctypes.c_int32
    nw = ctypes.c_int32(0)
    assert lib.icp_debug_wave_slot(0, 0, 1, ctypes.byref(nw)) == 0 and nw.value >= 1
    for mgrid in (1, 2, 3, 7, 127, 128, 129, 255, 256, 257, 1447, 1448, 2895, 4099):
        seen = np.zeros(mgrid * nw.value, np.int32)
        for lb in range(mgrid):
            for w in range(nw.value):
                s = lib.icp_debug_wave_slot(lb, w, mgrid, None)
                assert 0 <= s < mgrid * nw.value, (mgrid, lb, w, s)
                seen[s] += 1
        assert (seen == 1).all(), mgrid
    assert lib.icp_debug_wave_slot(5, 0, 5, None) == -1 and lib.icp_debug_wave_slot(0, nw.value, 5, None) == -1
