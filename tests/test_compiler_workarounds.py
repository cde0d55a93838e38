"""Regression guard for the two compiler work-arounds in k_bvh_block_levels (icp-variants_amd/csrc/dev_bvh.hpp): the natural
spellings of two expressions crash the ROCm 7.2 gfx950 instruction selector.  This test compiles the device code once more with
ICP_ISEL_NATURAL=1 and records what the installed compiler does:
  * it still crashes / errors  -> the work-arounds are still load-bearing (expected today);
  * it compiles                -> the work-arounds can go; the test still passes but says so loudly.
Either way the product spelling must compile (that is the build itself, __graft_entry__.build()); the RESULTS of the work-around
spelling are covered by the GPU parity tests of the BVH (tests/test_gpu_lbvh*.py: bit-identical to brute force and to the oracle)."""
import os
import subprocess
import warnings

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_natural_spellings_status(tmp_path):
    src = os.path.join(ROOT, "icp-variants_amd", "csrc", "icp_hip.hip")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-std=c++17", "-fPIC", "--cuda-device-only", "-DICP_ISEL_NATURAL=1",
           "-c", src, "-o", str(tmp_path / "natural.o")]
    try:
        out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    except subprocess.TimeoutExpired:
        warnings.warn("compiling the natural spellings did not finish in 900 s: work-arounds stay")
        return
    text = out.stdout.decode(errors="replace")
    if out.returncode == 0:
        warnings.warn("the gfx950 instruction-selector crash is gone with this compiler: the two work-arounds in k_bvh_block_levels "
                      "(ICP_ISEL_NATURAL) can be dropped")
    else:
        assert "k_bvh_block_levels" in text or "LLVM ERROR" in text or "Cannot select" in text or "error" in text.lower(), text[-2000:]
