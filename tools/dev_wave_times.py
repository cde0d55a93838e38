"""Dev tool (needs a build with ICP_DEBUG_STEPS=1 ICP_DEBUG_TIMES=1: ICP_HIP_LIB=.../libicp_hip_times.so): where the waves of ONE launch of the
fused matcher spend their time.  Lane 0 of every wave stamps the 100 MHz clock at the start, after the front end (loads + verify test), after the
walks, after weighting/rejection, after the wave reduction and at the end.  usage: ICP_HIP_LIB=... python tools/dev_wave_times.py [iterations ...]"""
import sys, os, ctypes as C
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "icp-variants_amd", "python"))
import numpy as np
from icp_amd import binding, synth
p = synth.eth_like_pair(0)
n = len(p["src_pts"])
c = binding.Context(0)
c.params.max_distance = 10.0; c.params.metric = 1; c.params.knn_backend = 1
c.set_stage_timing(0)
c.push_params(); c.set_target(p["tgt_pts"], p["tgt_nrm"]); c.set_source(p["src_pts"], p["src_nrm"])
BT = int(os.environ.get('ICP_DEV_BVH_THREADS', '256'))       # threads per block of the library under test
nw = ((n + BT - 1) // BT) * (BT // 64)
for iters in [int(a) for a in sys.argv[1:]] or [1, 3, 6, 12, 45]:
    c.params.n_iterations = iters; c.push_params()
    c.run(np.eye(4))
    buf = np.zeros(n, np.int32)
    assert c.lib.icp_debug_steps(c.h, buf.ctypes.data_as(C.c_void_p), C.c_int32(n)) == 0
    w = buf[: nw * 8].reshape(nw, 8)
    t = w[:, :6].astype(np.uint32).astype(np.int64)
    t0 = t[:, 0].min()
    rel = (t - t0) * 0.01                                     # us since the first wave started
    walkers = w[:, 6] & 0xFF; two_leaf = (w[:, 6] >> 8) & 0xFF
    hw = w[:, 7].astype(np.uint32)
    simd = ((hw >> 16) & 0xF).astype(np.int64) * 4096 + ((hw >> 13) & 7) * 512 + ((hw >> 12) & 1) * 256 + ((hw >> 8) & 0xF) * 4 + ((hw >> 4) & 3)      # (xcc, se, sh, cu, simd)
    if os.environ.get("ICP_DEV_DUMP"):                     # raw per-wave arrays for offline what-if analyses
        np.savez(os.path.join(os.environ["ICP_DEV_DUMP"], "waves_it%d.npz" % (iters - 1)), rel=rel, simd=simd, walkers=walkers)
    names = ["start", "front", "walks", "post", "reduce", "end"]
    print("iteration %d: %d waves, %d with walkers (%d walking queries), %d queries in the two-leaf tier; launch spans %.2f us from the first stamp"
          % (iters - 1, nw, (walkers > 0).sum(), walkers.sum(), two_leaf.sum(), rel[:, 5].max()))
    for label, m in (("all waves", np.ones(nw, bool)), ("waves without walkers", walkers == 0), ("waves with walkers", walkers > 0)):
        if not m.any(): continue
        r = rel[m]
        print("  %-22s" % label + "  ".join("%s: mean %.2f p99 %.2f max %.2f" % (names[j], r[:, j].mean(), np.percentile(r[:, j], 99), r[:, j].max()) for j in range(6)))
        d = np.diff(r, axis=1)
        print("  %-22s" % "  (phase lengths)" + "  ".join("%s: mean %.2f max %.2f" % (names[j + 1], d[:, j].mean(), d[:, j].max()) for j in range(5)))
    # per SIMD: when its last wave ends, how much wave time it held -- is the launch waiting for a few SIMDs (balance) or for all of them (latency)?
    ids, inv = np.unique(simd, return_inverse=True)
    s_end = np.zeros(len(ids)); np.maximum.at(s_end, inv, rel[:, 5])
    s_cnt = np.bincount(inv); s_busy = np.bincount(inv, weights=rel[:, 5] - rel[:, 0])
    print("  per SIMD (%d seen, %.1f waves each): last wave ends mean %.1f  p10 %.1f  p50 %.1f  p90 %.1f  max %.1f us; wave-time held mean %.0f max %.0f us"
          % (len(ids), s_cnt.mean(), s_end.mean(), np.percentile(s_end, 10), np.percentile(s_end, 50), np.percentile(s_end, 90), s_end.max(), s_busy.mean(), s_busy.max()))
    cu = simd // 4
    cids, cinv = np.unique(cu, return_inverse=True)
    c_end = np.zeros(len(cids)); np.maximum.at(c_end, cinv, rel[:, 5])
    c_mean = np.bincount(cinv, weights=rel[:, 5]) / np.bincount(cinv)
    print("  per CU (%d seen): last wave ends mean %.1f  p10 %.1f  p50 %.1f  p90 %.1f  max %.1f us; mean wave end per CU: min %.1f  p50 %.1f  max %.1f"
          % (len(cids), c_end.mean(), np.percentile(c_end, 10), np.percentile(c_end, 50), np.percentile(c_end, 90), c_end.max(), c_mean.min(), np.percentile(c_mean, 50), c_mean.max()))
    xc = simd // 4096
    print("  per XCD as placed by the hardware: " + "  ".join("x%d: %d waves, mean end %.1f, last %.1f" % (x, (xc == x).sum(), rel[xc == x, 5].mean(), rel[xc == x, 5].max()) for x in np.unique(xc)))
    if os.environ.get("ICP_DEV_PLACEMENT"):                # how does the dispatcher place consecutive hardware blocks?  (XCD, CU) of blocks 0, 8, 16, ... (the ones XCD 0 gets)
        nbk = nw // (BT // 64); CHK = 16; fullk = nbk // (8 * CHK) * (8 * CHK)
        def logical(b):
            if b >= fullk: return b
            x, j = b & 7, b >> 3
            return ((j // CHK) * 8 + x) * CHK + j % CHK
        seq = [(int(simd[2 * logical(b)] // 4096), int((simd[2 * logical(b)] // 4) % 1024)) for b in range(0, 8 * 80, 8)]
        print("  placement of hardware blocks 0, 8, 16, ...: " + " ".join("%d:%d" % s_ for s_ in seq))
        seq1 = [(int(simd[2 * logical(b)] // 4096), int((simd[2 * logical(b)] // 4) % 1024)) for b in range(0, 24)]
        print("  placement of hardware blocks 0..23: " + " ".join("%d:%d" % s_ for s_ in seq1))
    # which XCD ran which logical block (xcd_contiguous_block with ICP_XCD_CHUNK = 16): is one of them the tail?
    NWB = BT // 64; nb = nw // NWB; CH = 16; full = nb // (8 * CH) * (8 * CH)
    lbs = np.arange(nw) % nb if os.environ.get('ICP_DEV_STRIDE', '1') == '1' and NWB > 1 else np.arange(nw) // NWB      # logical block of every wave slot
    wx = np.where(lbs < full, (lbs // CH) % 8, lbs % 8)
    print("  per XCD: " + "  ".join("x%d: end mean %.1f max %.1f, walk-us sum %.0f" % (x, rel[wx == x, 5].mean(), rel[wx == x, 5].max(), (rel[wx == x, 2] - rel[wx == x, 1]).sum()) for x in range(8)))
    last = np.argsort(rel[:, 5])[-5:]
    for i in last: print("   late wave %5d: walkers %2d  stamps %s" % (i, walkers[i], np.round(rel[i], 2)))
    # the queries that searched (walk or two-leaf tier) in this launch: records written by the launch itself (clock within its window)
    rec = buf[nw * 8: nw * 8 + 8 * 4096].reshape(4096, 8)
    clk = rec[:, 7].astype(np.uint32).astype(np.int64)
    cur = rec[(clk >= t0) & (clk <= t[:, 5].max())]
    f = lambda a: a.view(np.float32)
    if len(cur) <= 64:
        for r in cur:
            best, lbo, lb3, delta = f(r[1:2])[0], f(r[2:3])[0], f(r[3:4])[0], f(r[4:5])[0]
            print("   searched: k %6d %s sqrt(best) %.6g  bound on others %.6g  bound outside two leaves %.6g  moved %.3g  | margin tier1 %.3g  tier2 %.3g"
                  % (r[0], "two-leaf l2=%d" % r[5] if r[5] != -2 else "walk", np.sqrt(best), lbo, lb3, delta, lbo - delta - np.sqrt(best), lb3 - delta - np.sqrt(best)))
