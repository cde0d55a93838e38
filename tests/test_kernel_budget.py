"""Register budget of the fused matcher (icp-variants_amd/csrc/dev_fused.hpp).  gfx950 allocates VGPRs in steps of 8 out of 512 per
SIMD lane: 80 registers are 6 resident waves per SIMD, 81 are 5 -- and an innocent-looking edit moves the count by two or three
(measured in round 2: 82 registers cost 10 % in the iterations where every query walks the tree).  Compiles the device code to
assembly and reads the counts the compiler reports; no GPU needed."""
import os
import re
import subprocess

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_fused_matcher_keeps_six_waves_per_simd_and_no_scratch(tmp_path):
    src = os.path.join(ROOT, "icp-variants_amd", "csrc", "icp_hip.hip")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    out = str(tmp_path / "icp_hip.s")
    import sys
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    flags = [f for f in g.HIPCC_FLAGS if f not in ("-shared", "-Wall")]      # the product's own flags
    subprocess.check_call([hipcc] + flags + ["--cuda-device-only", "-w", "-I", os.path.join(ROOT, "include"), "-S", src, "-o", out], timeout=900)
    text = open(out).read()
    seen = {}
    for name, field, val in re.findall(r"\.set (_ZN6icpdev\S*?)\.(num_vgpr|private_seg_size), (\d+)", text):
        seen.setdefault(name, {})[field] = int(val)

    def kernels(prefix):
        return {n: f for n, f in seen.items() if n.startswith("_ZN6icpdev" + prefix)}
    # the 3-D fused matchers: <3, false> (trees up to 8 four-wide levels) and <3, true>, separate launches and the merged loop's form
    # (which also carries the reducer blocks of the previous iteration: their fold must fit the same budget)
    for prefix in ("14k_knn_bvh_postILi3ELb", "19k_knn_bvh_post_ringILi3ELb"):
        ks = kernels(prefix)
        assert len(ks) == 2, (prefix, list(ks))
        for name, f in ks.items():
            assert f["num_vgpr"] <= 80, (name, f)              # 6 waves per SIMD
            assert f["private_seg_size"] == 0, (name, f)       # nothing spills
    # colour ICP (6-D boxes, QueryPt<6>, six shuffles per hand-over) and the stage-level matchers: 5 / 4 waves per SIMD, and -- what the
    # comments in dev_solve.hpp warn about -- no scratch: a dispatch with a scratch demand stalls the queue
    for prefix, cap in (("14k_knn_bvh_postILi6ELb", 96), ("19k_knn_bvh_post_ringILi6ELb", 96), ("9k_knn_bvhILi3E", 80), ("9k_knn_bvhILi6E", 112),
                        ("19k_ring_reduce_solve", 64), ("18k_icp_loop_reducer", 112)):      # (the loop's reducer sits in 112-register holes: dev_persist.hpp)
        ks = kernels(prefix)
        assert ks, prefix
        for name, f in ks.items():
            assert f["num_vgpr"] <= cap, (name, f)
            assert f["private_seg_size"] == 0, (name, f)
    # k_icp_loop (experimental: ICP_HIP_PERSIST=1): the whole grid must be resident -> 6 waves per SIMD by decree (__launch_bounds__), which the
    # allocator currently pays for with a few spilled dwords
    ks = kernels("10k_icp_loopILi")
    assert len(ks) == 4, list(ks)
    for name, f in ks.items():
        assert f["num_vgpr"] <= 80 and f["private_seg_size"] <= 64, (name, f)
