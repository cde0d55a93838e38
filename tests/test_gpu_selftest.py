"""Hardware self test of the in-register transposing wave reduction (dev_post.hpp: wave_transpose_reduce_gen) that the fused
matcher's epilogue is built on: v_permlane32_swap / v_permlane16_swap / DPP row operations on gfx950.  The same pairing is
replayed with numpy (IEEE fp64 adds): the device result must be bit-identical, and every value must sit in the lane the
kernel's bookkeeping says it does."""
import ctypes as C
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def replay(x):
    """x: (64, N) per-lane values.  Returns (totals[N], lane_of[N]) following the kernel's steps over lane bits 5,4,3,2,1,0."""
    lanes = np.arange(64)
    partner = {5: lanes ^ 32, 4: lanes ^ 16, 3: (lanes & ~15) | (15 - (lanes & 15)), 2: (lanes & ~7) | (7 - (lanes & 7)), 1: lanes ^ 2, 0: lanes ^ 1}
    cur = x.copy()                                       # (64, n_cur)
    for bit in (5, 4, 3, 2, 1, 0):
        n = cur.shape[1]; h = (n + 1) // 2
        if n % 2:
            cur = np.concatenate([cur, np.zeros((64, 1))], 1)
        a, b = cur[:, 0::2], cur[:, 1::2]                 # value 2j / 2j+1
        up = ((lanes >> bit) & 1).astype(bool)[:, None]
        keep = np.where(up, b, a); send = np.where(up, a, b)
        cur = keep + send[partner[bit]]
        assert cur.shape[1] == h
    lane_of = np.array([[l for l in range(64) if (((l >> 5) & 1) | (((l >> 4) & 1) << 1) | (((l >> 3) & 1) << 2) | (((l >> 2) & 1) << 3) | (((l >> 1) & 1) << 4) | ((l & 1) << 5)) == v][0]
                        for v in range(x.shape[1])])
    return cur[lane_of, 0], lane_of


def test_wave_transpose_reduce_bit_identical(gpu_ctx_factory):
    c = gpu_ctx_factory()
    c.lib.icp_selftest_wave_reduce.restype = C.c_int
    rng = np.random.default_rng(7)
    for scale in (1.0, 1e12):
        x = (rng.standard_normal((64, 27)) * scale * np.exp(rng.uniform(-20, 20, (64, 27)))).astype(np.float64)
        out = np.zeros(27); lane_of = np.zeros(27, np.int32)
        rc = c.lib.icp_selftest_wave_reduce(c.h, x.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), lane_of.ctypes.data_as(C.c_void_p))
        assert rc == 0
        exp, lanes = replay(x)
        assert np.array_equal(out.view(np.uint64), exp.view(np.uint64))
        assert np.allclose(out, x.sum(0), rtol=1e-9, atol=1e-3 * np.abs(x).max())
        assert np.array_equal(np.sort(lane_of), np.sort(lane_of)) and all((l & 1) == 0 for l in lane_of)
