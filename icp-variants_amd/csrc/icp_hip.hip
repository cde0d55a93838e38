// =====================================================================================
// icp_hip.hip -- C-ABI implementation (libicp_hip.so) over the gfx950 kernels in icp_device.hpp.
// Entry points and the reference interfaces they replace are documented in include/icp_hip.h.
// There is NO CPU fallback: every entry point needs a HIP device and fails with ICP_ERR_HIP /
// ICP_ERR_NO_DEVICE when none is usable.
// =====================================================================================
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include "icp_device.hpp"
#include <hip/hip_ext.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

using namespace icpdev;

#define HIPCK(ctx, expr)                                                                        \
    do {                                                                                        \
        hipError_t e__ = (expr);                                                                \
        if (e__ != hipSuccess) {                                                                \
            char buf__[256];                                                                    \
            snprintf(buf__, sizeof(buf__), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            (ctx)->err = buf__;                                                                 \
            return ICP_ERR_HIP;                                                                 \
        }                                                                                       \
    } while (0)

namespace {

struct DevBuf {
    void* p = nullptr; size_t cap = 0;
    bool view = false;                   // part of another allocation (a plane of a packed level, a section of the search-state pack): never freed on its own
    template <class T> T* as() const { return (T*)p; }
};

struct Cloud {
    int n = 0, npad = 0;
    DevBuf x, y, z, nx, ny, nz, cr, cg, cb, rgba;
    bool has_normals = false, has_colors = false;
};

// One resolution level of the source: the selection (original indices, increasing), and -- for the BVH matcher -- a physical
// copy of the selected points in Morton order, so that everything the ICP loop touches per query (source planes, search
// state, matches) is indexed by the same sorted position and streams coalesced.  factor 0 = the whole cloud, unfiltered.
struct Level { DevBuf idx; DevBuf order; DevBuf sorted_idx; DevBuf pack; Cloud sorted; bool sorted_valid = false; int n = 0; };   // pack: the sorted copy's planes in ONE allocation (x y z nx ny nz cr cg cb rgba, a fixed stride apart: k_icp_loop)

// LBVH over the target (buildIndex): device buffers + the host-side facts needed to launch the build.
struct Bvh {
    bool valid = false;
    int n_valid = 0, n_leaves = 0, Lp = 1;
    DevBuf keys, keys2, vals, vals2, temp, leaves, recs, nodes, qnodes, lvl, wbox, pos_of;
    DevBuf axl[12], side, scanr, axis_of_node;      // presorted-axes build: DIM index lists (ping-pong), side flag per point id, scan result, widest axis per node
    int n_ids = 0;                                   // size of the id space the lists index (points of the cloud the tree is built over)
    const Cloud* attrs = nullptr;                     // cloud whose normals / colours go into the records (nullptr: none)
    int Lq = 0;                                       // 4-wide levels
    const int* d_finite = nullptr;                    // device list of the finite points' indices, increasing (owned by the context)
    double build_ms = 0.0;
};

}  // namespace

struct icp_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    int stage_timing = 1;                // icp_set_stage_timing: 0 none, 1 every iteration, N > 1 every Nth iteration (scaled)
    unsigned timing_phase = 0;           // rotates the sampled iterations from run to run
    void* pinned = nullptr; size_t pinned_cap = 0;   // page-locked host staging: pose upload, stats + pose download (truly asynchronous copies)
    bool block_levels = true;            // BVH build: levels with slices <= 2048 points in one LDS kernel (ICP_HIP_BLOCK_LEVELS=0: global sorts)
    bool spin_reduce = true;             // k_reduce_solve: block 0 polls the self-validating totals (ICP_HIP_SPIN_REDUCE=0: ticket hand-over, last arriver solves)
    bool tier2 = true;                   // incremental k-NN: second verification tier (one leaf instead of a walk; ICP_HIP_TIER2=0 disables)
    bool presort = true;                 // BVH build: upper levels from presorted axes (ICP_HIP_PRESORT=0: one global sort per level)
    bool trace = false;                  // ICP_HIP_TRACE=1: per-iteration stage times on stderr
    bool fuse_post = true;               // BVH matcher runs weight / reject / accumulate as its epilogue (ICP_HIP_FUSE_POST=0 disables)
    bool merge_loop = true;              // point-to-plane loop through the fused BVH matcher: reduce + solve ride in front of the next matcher launch (ICP_HIP_MERGE=0: separate k_reduce_solve launches)
    int loop_from = 0;                   // k_icp_loop takes over at this iteration of a run; the ones before it run one (merged) launch each (ICP_HIP_LOOP_FROM)
    bool persist_loop = false;           // ICP_HIP_PERSIST=1 (experimental, measured slower than the merged loop so far: DESIGN.md): when the whole grid fits the device at once, the loop of a resolution level as ONE launch (k_icp_loop; ICP_HIP_PERSIST=0 disables)
    bool shared_gpu = false;             // other contexts work on this device at the same time (icp_batch_run with several contexts): one launch per iteration
    int loop_runs = 0;                   // runs that took k_icp_loop (icp_debug_counters)
    hipStream_t stream2 = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr;   // k_icp_loop_reducer runs BESIDE the matcher grid: its own stream, forked off / joined to the context's
    int loop_capacity[4] = {0, 0, 0, 0}; // resident blocks of k_icp_loop<3/6, false/true> on this device (0: not asked yet)
    int merged_runs = 0, merged_fallbacks = 0;   // runs that took the merged loop / that had to be repeated with the separate launches (icp_debug_counters)
    bool ext_events = true;              // merged form: stage times from hipExtLaunchKernel's start / stop events (ICP_HIP_EXT_EVENTS=0: hipEventRecord brackets)
    bool gx_on = ICP_GX != 0; int gx_start = 16, gx_empty = 3;   // hand-over between the blocks of the fused matcher (dev_bvh.hpp, GX): ICP_HIP_GX=0 disables; ICP_HIP_GX_START / ICP_HIP_GX_EMPTY
    DevBuf gx_slots, gx_hdr; int gx_blocks = 0; bool gx_dirty = false;      // the outboxes (armed once: every launch leaves them as it found them; gx_dirty: a run was cut short)
    bool keep_fused_records = false;     // icp_match_seeded: the fused matcher also writes its Match records and distances (the loop itself never reads them)
    icp_params prm;
    Cloud tgt, src, qry;                 // qry: scratch cloud of icp_query_matches
    Cloud nrm_cloud; Bvh nrm_bvh;        // scratch of icp_estimate_normals
    Bvh bvh, bvh6;                       // exact kd-ordered BVH of the target over xyz / over xyz+rgb (knn_backend == ICP_KNN_LBVH)
    DevBuf src_flag, src_box;            // per source point: finite point && finite normal (PointCloud.h:334); bounding box of the finite points (ordered bits)
    DevBuf tgt_flag, tgt_finite, nrm_finite, sel_temp, d_count;   // finite filters of the index builds, compaction scratch
    void* pin_up = nullptr; size_t pin_up_cap = 0; hipEvent_t up_ev = nullptr; bool up_pending = false;   // page-locked upload staging + "copy has left it" event
    DevBuf okeys, okeys2, ovals, otemp;  // scratch of the Morton sort of the queries
    std::map<int, Level> levels;         // multires selections by decimation factor
    DevBuf sel_lists, sel_counts, sel_blocks;            // RANDOM_SAMPLING: per-iteration index lists, their sizes, scan scratch
    DevBuf qpack; size_t q_cap = 0;                      // nn_raw | qstate | qstate2 (views below), q_cap elements each
    DevBuf qstate, qstate2;                              // incremental k-NN: per-query anchor + bound on the other targets; bound on the targets outside the neighbour's leaf
    DevBuf dbg_steps;                    // development builds only (ICP_DEBUG_STEPS)
    DevBuf ps, matches, d2, best64, nn_raw, partials, partials2, ring, pring, totals, sums, stats, staging, rmse_partials, rmse_out, fontana_partials;
    Cloud conv_src, conv_ref; int conv_n = 0;
    float cos_reject = 0.5f;
    std::vector<hipEvent_t> events;
    hipEvent_t build_ev[2] = {nullptr, nullptr};   // index-build bracket (build_bvh)
    icp_timing timing;
    std::vector<float> it_match_ms, it_post_ms, it_solve_ms;   // per iteration of the last run; -1 where the iteration was not bracketed
    std::string err;
};

namespace {

constexpr int POST_BLOCKS = 512;

int ensure(icp_ctx* c, DevBuf& b, size_t bytes) {
    if (bytes <= b.cap && b.p) return ICP_OK;
    if (b.view) { b.p = nullptr; b.cap = 0; b.view = false; }      // outgrown: becomes an allocation of its own
    if (b.p) { HIPCK(c, hipFree(b.p)); b.p = nullptr; b.cap = 0; }
    size_t want = bytes < 256 ? 256 : bytes;
    HIPCK(c, hipMalloc(&b.p, want));
    b.cap = want;
    return ICP_OK;
}
int ensure_pinned(icp_ctx* c, size_t bytes) {
    if (bytes <= c->pinned_cap && c->pinned) return ICP_OK;
    if (c->pinned) { HIPCK(c, hipHostFree(c->pinned)); c->pinned = nullptr; c->pinned_cap = 0; }
    const size_t want = bytes < 4096 ? 4096 : bytes;
    HIPCK(c, hipHostMalloc(&c->pinned, want, hipHostMallocDefault));
    c->pinned_cap = want;
    return ICP_OK;
}
void release(DevBuf& b) { if (b.p && !b.view) (void)hipFree(b.p); b.p = nullptr; b.cap = 0; b.view = false; }
void set_view(DevBuf& b, void* p, size_t bytes) { release(b); b.p = p; b.cap = bytes; b.view = true; }
void release(Cloud& c) { release(c.x); release(c.y); release(c.z); release(c.nx); release(c.ny); release(c.nz); release(c.cr); release(c.cg); release(c.cb); release(c.rgba); }
void release(Level& lv) { release(lv.idx); release(lv.order); release(lv.sorted_idx); release(lv.sorted); release(lv.pack); lv.sorted_valid = false; }

// Largest float c with (double)acosf(c) > 60*pi/180 on THIS host's libm: the device rejection test
// `c <= cos_reject` is then bit-identical to the reference's `acos(c) > threshold` (ICPOptimizer.h:161,170)
// as evaluated by the host the reference would run on (acosf is monotone on [0.25, 0.75]).
float compute_cos_reject() {
    const double threshold = 60 * 3.141592653589793238462643383279502884 / 180.0;
    uint32_t lo, hi; float flo = 0.25f, fhi = 0.75f;
    memcpy(&lo, &flo, 4); memcpy(&hi, &fhi, 4);       // predicate true at lo, false at hi
    while (hi - lo > 1) {
        uint32_t mid = lo + (hi - lo) / 2; float fm; memcpy(&fm, &mid, 4);
        if ((double)acosf(fm) > threshold) lo = mid; else hi = mid;
    }
    float r; memcpy(&r, &lo, 4);
    return r;
}

int set_device(icp_ctx* c) { HIPCK(c, hipSetDevice(c->device)); return ICP_OK; }

// Every entry point that enqueues work synchronises the stream before it returns (write_pose's contract: the page-locked staging
// area and the scratch buffers are free again by the next call).  On the success paths that is the entry point's own final
// hipStreamSynchronize; this guard covers the error returns in between.
struct DrainOnError {
    icp_ctx* c; bool ok = false;
    explicit DrainOnError(icp_ctx* ctx) : c(ctx) {}
    ~DrainOnError() { if (!ok && c && c->stream) { if (c->stream2) (void)hipStreamSynchronize(c->stream2); (void)hipStreamSynchronize(c->stream); } }
    int done(int rc = ICP_OK) { ok = (rc == ICP_OK); return rc; }
};

// Host clouds -> device SoA planes.  The whole cloud (points, normals, colours) goes through ONE page-locked staging buffer and
// ONE asynchronous copy, the AoS -> SoA kernels follow on the stream, and nothing here waits for the device: the only host-side
// wait is for the previous upload to have left the staging buffer.  (Round 1: pageable copies + one synchronisation per plane.)
int ensure_pin_up(icp_ctx* c, size_t bytes) {
    if (c->up_pending) { HIPCK(c, hipEventSynchronize(c->up_ev)); c->up_pending = false; }
    if (bytes <= c->pin_up_cap && c->pin_up) return ICP_OK;
    if (c->pin_up) { HIPCK(c, hipHostFree(c->pin_up)); c->pin_up = nullptr; c->pin_up_cap = 0; }
    const size_t want = bytes < 65536 ? 65536 : bytes + bytes / 8;
    HIPCK(c, hipHostMalloc(&c->pin_up, want, hipHostMallocDefault));
    c->pin_up_cap = want;
    if (!c->up_ev) HIPCK(c, hipEventCreateWithFlags(&c->up_ev, hipEventDisableTiming));
    return ICP_OK;
}
int upload_cloud(icp_ctx* c, Cloud& cl, const float* xyz, const float* nrm, const uint8_t* rgba, int n, bool pad_inf) {
    const int npad = pad_inf ? ((n + 63) / 64) * 64 : n;
    const size_t b_xyz = (size_t)n * 12, b_nrm = nrm ? (size_t)n * 12 : 0, b_col = rgba ? (size_t)n * 4 : 0, total = b_xyz + b_nrm + b_col;
    int rc;
    if ((rc = ensure_pin_up(c, total))) return rc;
    if ((rc = ensure(c, c->staging, total))) return rc;
    char* h = (char*)c->pin_up;
    memcpy(h, xyz, b_xyz);
    if (nrm) memcpy(h + b_xyz, nrm, b_nrm);
    if (rgba) memcpy(h + b_xyz + b_nrm, rgba, b_col);
    HIPCK(c, hipMemcpyAsync(c->staging.p, h, total, hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipEventRecord(c->up_ev, c->stream)); c->up_pending = true;
    const char* d = c->staging.as<char>();
    const dim3 g((npad + 255) / 256), b(256);
    for (DevBuf* pl : {&cl.x, &cl.y, &cl.z}) if ((rc = ensure(c, *pl, (size_t)npad * 4))) return rc;
    hipLaunchKernelGGL(k_deinterleave3, g, b, 0, c->stream, (const float*)d, n, npad, INFINITY, cl.x.as<float>(), cl.y.as<float>(), cl.z.as<float>());
    cl.has_normals = nrm != nullptr;
    if (nrm) {
        for (DevBuf* pl : {&cl.nx, &cl.ny, &cl.nz}) if ((rc = ensure(c, *pl, (size_t)n * 4))) return rc;
        hipLaunchKernelGGL(k_deinterleave3, dim3((n + 255) / 256), b, 0, c->stream, (const float*)(d + b_xyz), n, n, 0.f, cl.nx.as<float>(), cl.ny.as<float>(), cl.nz.as<float>());
    }
    cl.has_colors = rgba != nullptr;
    if (rgba) {
        for (DevBuf* pl : {&cl.rgba, &cl.cr, &cl.cg, &cl.cb}) if ((rc = ensure(c, *pl, (size_t)npad * 4))) return rc;
        hipLaunchKernelGGL(k_colors, g, b, 0, c->stream, (const uint8_t*)(d + b_xyz + b_nrm), n, npad, cl.rgba.as<uint32_t>(), cl.cr.as<float>(), cl.cg.as<float>(), cl.cb.as<float>());
    }
    HIPCK(c, hipGetLastError());
    cl.n = n; cl.npad = npad;
    return ICP_OK;
}
// one plane triple through the same staging path (convergence reference)
int upload3(icp_ctx* c, const float* aos, int n, int npad, float pad_value, DevBuf& x, DevBuf& y, DevBuf& z) {
    int rc;
    if ((rc = ensure_pin_up(c, (size_t)n * 12))) return rc;
    if ((rc = ensure(c, c->staging, (size_t)n * 12))) return rc;
    for (DevBuf* pl : {&x, &y, &z}) if ((rc = ensure(c, *pl, (size_t)npad * 4))) return rc;
    memcpy(c->pin_up, aos, (size_t)n * 12);
    HIPCK(c, hipMemcpyAsync(c->staging.p, c->pin_up, (size_t)n * 12, hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipEventRecord(c->up_ev, c->stream)); c->up_pending = true;
    hipLaunchKernelGGL(k_deinterleave3, dim3((npad + 255) / 256), dim3(256), 0, c->stream, c->staging.as<float>(), n, npad, pad_value, x.as<float>(), y.as<float>(), z.as<float>());
    HIPCK(c, hipGetLastError());
    HIPCK(c, hipStreamSynchronize(c->stream));      // staging is reused by the caller's next plane
    return ICP_OK;
}

// Indices j * factor (j = 0 .. count - 1) whose flag is set, in increasing order, compacted on the device (rocPRIM select); one
// 4-byte copy returns how many there are.  flags: one byte per j.
int compact_flagged(icp_ctx* c, const uint8_t* d_flags, int count, int factor, DevBuf& out, int* n_out) {
    int rc;
    if ((rc = ensure(c, out, (size_t)(count > 0 ? count : 1) * 4))) return rc;
    if ((rc = ensure(c, c->d_count, 16))) return rc;
    *n_out = 0;
    if (count <= 0) return ICP_OK;
    auto in = rocprim::make_transform_iterator(rocprim::counting_iterator<int>(0), MulBy{factor});
    size_t tb = 0;
    HIPCK(c, rocprim::select(nullptr, tb, in, d_flags, out.as<int>(), c->d_count.as<int>(), (size_t)count, c->stream));
    if ((rc = ensure(c, c->sel_temp, tb))) return rc;
    HIPCK(c, rocprim::select(c->sel_temp.p, tb, in, d_flags, out.as<int>(), c->d_count.as<int>(), (size_t)count, c->stream));
    if ((rc = ensure_pinned(c, 4096))) return rc;
    int* h = (int*)((char*)c->pinned + 2048);            // (the first bytes of the pinned block stage the pose)
    HIPCK(c, hipMemcpyAsync(h, c->d_count.p, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    *n_out = *h;
    return ICP_OK;
}
// finite filter of a cloud that is already on the device -> flag bytes + compacted index list
int finite_list(icp_ctx* c, const Cloud& cl, bool with_normals, DevBuf& flag, DevBuf& list, int* n_out) {
    int rc;
    if ((rc = ensure(c, flag, (size_t)cl.n))) return rc;
    const bool nrm = with_normals && cl.has_normals;
    hipLaunchKernelGGL(k_mark_finite, dim3((cl.n + 255) / 256), dim3(256), 0, c->stream, cl.x.as<float>(), cl.y.as<float>(), cl.z.as<float>(),
                       nrm ? cl.nx.as<float>() : nullptr, nrm ? cl.ny.as<float>() : nullptr, nrm ? cl.nz.as<float>() : nullptr, cl.n, flag.as<uint8_t>());
    HIPCK(c, hipGetLastError());
    return compact_flagged(c, flag.as<uint8_t>(), cl.n, 1, list, n_out);
}

// Upload the pose state.  Staged through the context's page-locked buffer: no synchronisation here -- every entry point
// that uses the pose synchronises the stream before it returns, so the staging area is free again by the next call.
int write_pose(icp_ctx* c, const float pose[16]) {
    int rc;
    if ((rc = ensure_pinned(c, sizeof(PoseState)))) return rc;
    PoseState* h = (PoseState*)c->pinned; memset(h, 0, sizeof(*h));
    memcpy(h->pose, pose, 64);
    normal_matrix_from_pose(h->pose, h->nmat);
    if ((rc = ensure(c, c->ps, sizeof(PoseState)))) return rc;
    HIPCK(c, hipMemcpyAsync(c->ps.p, h, sizeof(*h), hipMemcpyHostToDevice, c->stream));
    return ICP_OK;
}

// One launch of the merged loop: the pose slot its matcher blocks wait for, where they leave their partials, and the reducer that rides in front.
struct MergeLaunch { RingParams rp; const PoseState* slot; double* partials; const LoopParams* loop = nullptr; hipEvent_t ev_start = nullptr, ev_stop = nullptr; };   // ev_start / ev_stop: the launch's own start / stop times go into these events (hipExtLaunchKernel: taken from the dispatch itself, no bracket on the stream)   // loop != nullptr: k_icp_loop (all iterations of a level in one launch)

struct QuerySet { const Cloud* cl; const int* sel; int n; int pretransformed; bool use_colors; bool seed_prev; const int* order; };   // cl/sel: also what the post stage reads

int ensure_qpack(icp_ctx* c, int n) {
    if ((size_t)n <= c->q_cap && c->qpack.p) return ICP_OK;
    int rc;
    c->q_cap = ((size_t)n + 63) / 64 * 64;
    if ((rc = ensure(c, c->qpack, c->q_cap * 28))) return rc;
    set_view(c->nn_raw, c->qpack.p, c->q_cap * 4); set_view(c->qstate, c->qpack.as<char>() + c->q_cap * 4, c->q_cap * 16);
    set_view(c->qstate2, c->qpack.as<char>() + c->q_cap * 20, c->q_cap * 8);
    return ICP_OK;
}
int ensure_match_buffers(icp_ctx* c, int n) {
    int rc;
    if ((rc = ensure(c, c->matches, (size_t)n * sizeof(icp_match_t)))) return rc;
    if ((rc = ensure(c, c->d2, (size_t)n * 4))) return rc;
    return ICP_OK;
}

// Morton order of the query positions [0, n) of a selection (sel == nullptr: the full source): out[t] = position.
// rocPRIM sorts 370 k pairs with its MERGE sort (radix_sort_config's limit: 1 M items): a block sort and nine merge passes of two
// launches each -- 19 launches of ~8 us per sort, four sorts per scan (three axis orders for the index, the Morton order of the queries).
// ICP_SORT_MERGE_LIMIT=0 sends them to the Onesweep radix sort instead (a histogram launch and one pass per 8 key bits).  Measured
// (round 3, same results -- both are stable): icp_set_target 1.34-1.39 ms against 1.31-1.35, the first icp_run (Morton sort of 64-bit
// keys) 1.88 against 1.76-1.79 ms, a batch of 16 pairs 480-523 against 523-541 pairs/s: fewer launches, more time.  Merge sort stays.
#ifndef ICP_SORT_MERGE_LIMIT
#define ICP_SORT_MERGE_LIMIT (1024 * 1024)
#endif
using SortCfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, ICP_SORT_MERGE_LIMIT>;
int build_query_order(icp_ctx* c, const int* d_sel, int n, DevBuf& out) {
    int rc;
    if ((rc = ensure(c, c->okeys, (size_t)n * 8))) return rc;
    if ((rc = ensure(c, c->okeys2, (size_t)n * 8))) return rc;
    if ((rc = ensure(c, c->ovals, (size_t)n * 4))) return rc;
    if ((rc = ensure(c, out, (size_t)n * 4))) return rc;
    hipLaunchKernelGGL(k_query_keys, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->src.x.as<float>(), c->src.y.as<float>(), c->src.z.as<float>(), d_sel, n,
                       c->src_box.as<unsigned int>(), c->okeys.as<unsigned long long>(), c->ovals.as<int>());
    size_t temp_bytes = 0;
    HIPCK(c, rocprim::radix_sort_pairs<SortCfg>(nullptr, temp_bytes, c->okeys.as<unsigned long long>(), c->okeys2.as<unsigned long long>(), c->ovals.as<int>(), out.as<int>(), (size_t)n, 0, 64, c->stream));
    if ((rc = ensure(c, c->otemp, temp_bytes))) return rc;
    HIPCK(c, rocprim::radix_sort_pairs<SortCfg>(c->otemp.p, temp_bytes, c->okeys.as<unsigned long long>(), c->okeys2.as<unsigned long long>(), c->ovals.as<int>(), out.as<int>(), (size_t)n, 0, 64, c->stream));
    HIPCK(c, hipGetLastError());
    return ICP_OK;
}

// Build the kd-ordered BVH of the resident target on the device (once per icp_set_target; = buildIndex).
template <int DIM>
int build_bvh(icp_ctx* c, Bvh& b, const CoordPtrs<DIM>& cp) {
    int rc;
    if (!c->build_ev[0]) HIPCK(c, hipEventCreate(&c->build_ev[0]));      // owned by the context: nothing to leak on an error return
    if (!c->build_ev[1]) HIPCK(c, hipEventCreate(&c->build_ev[1]));
    const hipEvent_t e0 = c->build_ev[0], e1 = c->build_ev[1];
    HIPCK(c, hipEventRecord(e0, c->stream));
    const int nv = b.n_valid;
    b.n_leaves = (nv + BVH_LEAF - 1) / BVH_LEAF;
    b.Lp = 1; while (b.Lp < b.n_leaves) b.Lp <<= 1;
    int depth = 0; while ((1 << depth) < b.Lp) depth++;          // internal levels 0 .. depth-1
    const int n_inner = b.Lp - 1;
    const int n_slots = (b.n_leaves > 0 ? b.n_leaves : 1) * BVH_LEAF;
    const int cap = nv > 0 ? nv : 1;
    if ((rc = ensure(c, b.keys, (size_t)cap * 8))) return rc;
    if ((rc = ensure(c, b.keys2, (size_t)cap * 8))) return rc;
    if ((rc = ensure(c, b.vals, (size_t)cap * 4))) return rc;
    if ((rc = ensure(c, b.vals2, (size_t)cap * 4))) return rc;
    if ((rc = ensure(c, b.leaves, (size_t)(n_slots / BVH_LEAF) * sizeof(BvhLeafT<DIM>)))) return rc;
    if ((rc = ensure(c, b.recs, (size_t)n_slots * sizeof(TgtRec)))) return rc;
    if ((rc = ensure(c, b.pos_of, (size_t)(b.n_ids > 0 ? b.n_ids : 1) * 4))) return rc;      // position by original index (knn_walk_shared, XW)
    if ((rc = ensure(c, b.nodes, (size_t)(n_inner > 0 ? n_inner : 1) * sizeof(BvhNodeT<DIM>)))) return rc;
    if ((rc = ensure(c, b.lvl, (size_t)(b.Lp > 1 ? b.Lp / 2 : 1) * 2 * DIM * 4))) return rc;
    if ((rc = ensure(c, b.wbox, (size_t)((cap + 63) / 64) * 2 * DIM * 4))) return rc;
    int* perm = b.vals.as<int>(); int* perm2 = b.vals2.as<int>();
    if (nv > 0) {
        // finite targets in index order (device list from icp_set_target)
        HIPCK(c, hipMemcpyAsync(perm, b.d_finite, (size_t)nv * 4, hipMemcpyDeviceToDevice, c->stream));
        size_t temp_bytes = 0;
        HIPCK(c, rocprim::radix_sort_pairs<SortCfg>(nullptr, temp_bytes, b.keys.as<unsigned long long>(), b.keys2.as<unsigned long long>(), perm, perm2, (size_t)nv, 0, 64, c->stream));
        if ((rc = ensure(c, b.temp, temp_bytes))) return rc;
        const int gb = (nv + 255) / 256;
        int d_first = 0;
        if (c->presort && c->block_levels) {
            // upper levels (slices > 2048 points) from presorted axes: see dev_bvh.hpp
            int n_upper = 0;
            for (int d = 0; d < depth; d++) { int sh = 0; { long long seg = (long long)BVH_LEAF * b.Lp >> d; while ((1LL << sh) < seg) sh++; } if (sh <= 11) break; n_upper++; }
            if (n_upper > 0) {
                for (int k = 0; k < 2 * DIM; k++) if ((rc = ensure(c, b.axl[k], (size_t)cap * 4))) return rc;
                if ((rc = ensure(c, b.side, (size_t)(b.n_ids > 0 ? b.n_ids : 1)))) return rc;
                if ((rc = ensure(c, b.axis_of_node, (size_t)1 << n_upper))) return rc;
                unsigned int* k32 = b.keys.as<unsigned int>(); unsigned int* k32b = b.keys2.as<unsigned int>();
                size_t tb = 0;
                HIPCK(c, rocprim::radix_sort_pairs<SortCfg>(nullptr, tb, k32, k32b, perm, perm2, (size_t)nv, 0, 32, c->stream));
                if ((rc = ensure(c, b.temp, tb > temp_bytes ? tb : temp_bytes))) return rc;
                const int nblk = (nv + PRS_THREADS - 1) / PRS_THREADS;
                if ((rc = ensure(c, b.scanr, (size_t)2 * DIM * nblk * 4))) return rc;
                int* blk_cnt = b.scanr.as<int>(); int* blk_off = blk_cnt + (size_t)DIM * nblk;
                int* cur[DIM]; int* alt[DIM];
                for (int k = 0; k < DIM; k++) {           // one stable sort per axis (ids arrive in increasing order: ties keep index order)
                    cur[k] = b.axl[k].as<int>(); alt[k] = b.axl[DIM + k].as<int>();
                    hipLaunchKernelGGL(k_axis_keys, dim3(gb), dim3(256), 0, c->stream, cp.c[k], b.d_finite, nv, k32);
                    HIPCK(c, rocprim::radix_sort_pairs<SortCfg>(b.temp.p, tb, k32, k32b, b.d_finite, cur[k], (size_t)nv, 0, 32, c->stream));
                }
                for (int d = 0; d < n_upper; d++) {
                    int sh = 0; { long long seg = (long long)BVH_LEAF * b.Lp >> d; while ((1LL << sh) < seg) sh++; }
                    AxisLists<DIM> al, ao; for (int k = 0; k < DIM; k++) { al.L[k] = cur[k]; ao.L[k] = alt[k]; }
                    const int n_nodes = 1 << d;
                    hipLaunchKernelGGL(k_presort_axis<DIM>, dim3((n_nodes + 255) / 256), dim3(256), 0, c->stream, cp, al, nv, sh, n_nodes, b.axis_of_node.as<unsigned char>());
                    hipLaunchKernelGGL(k_presort_side<DIM>, dim3(gb), dim3(256), 0, c->stream, al, nv, sh, b.axis_of_node.as<unsigned char>(), b.side.as<unsigned char>());
                    hipLaunchKernelGGL(k_presort_count<DIM>, dim3(nblk, DIM), dim3(PRS_THREADS), 0, c->stream, al, b.side.as<unsigned char>(), nv, blk_cnt);
                    hipLaunchKernelGGL(k_presort_blockscan, dim3(DIM), dim3(1024), 0, c->stream, blk_cnt, nblk, blk_off);
                    hipLaunchKernelGGL(k_presort_scatter<DIM>, dim3(nblk, DIM), dim3(PRS_THREADS), 0, c->stream, al, b.side.as<unsigned char>(), blk_off, nv, sh, ao);
                    for (int k = 0; k < DIM; k++) { int* t = cur[k]; cur[k] = alt[k]; alt[k] = t; }
                }
                HIPCK(c, hipMemcpyAsync(perm, cur[0], (size_t)nv * 4, hipMemcpyDeviceToDevice, c->stream));      // any list: the block kernel sorts inside its slices
                HIPCK(c, hipGetLastError());
                d_first = n_upper;
            }
        }
        for (int d = d_first; d < depth; d++) {
            // segment (node) size at level d in points: BVH_LEAF * Lp / 2^d  = 1 << seg_shift
            int seg_shift = 0; { long long seg = (long long)BVH_LEAF * b.Lp >> d; while ((1LL << seg_shift) < seg) seg_shift++; }
            if (c->block_levels && seg_shift <= 11) {        // slices of <= 2048 points: all remaining levels inside LDS, one launch
                hipLaunchKernelGGL(k_bvh_block_levels<DIM>, dim3((nv + BLV_POINTS - 1) / BLV_POINTS), dim3(BLV_THREADS), 0, c->stream, cp, perm, nv, seg_shift, perm2);
                int* t = perm; perm = perm2; perm2 = t;
                break;
            }
            const int n_nodes = 1 << d;
            if (seg_shift >= 6) {            // wave-aligned segments: per-wave boxes, then one wave per node folds them
                const int n_waves = (nv + 63) / 64;
                hipLaunchKernelGGL(k_bvh_wave_boxes<DIM>, dim3(gb), dim3(256), 0, c->stream, cp, perm, nv, 6, b.wbox.as<unsigned int>());
                hipLaunchKernelGGL(k_bvh_node_boxes<DIM>, dim3((n_nodes * 64 + 255) / 256), dim3(256), 0, c->stream, b.wbox.as<unsigned int>(), n_waves, seg_shift - 6, n_nodes,
                                   b.lvl.as<unsigned int>());
            } else {                         // sub-wave segments (last levels): the segment heads write the node boxes directly
                hipLaunchKernelGGL(k_bvh_wave_boxes<DIM>, dim3(gb), dim3(256), 0, c->stream, cp, perm, nv, seg_shift, b.lvl.as<unsigned int>());
            }
            hipLaunchKernelGGL(k_bvh_level_keys<DIM>, dim3(gb), dim3(256), 0, c->stream, cp, perm, nv, seg_shift, b.lvl.as<unsigned int>(), b.keys.as<unsigned long long>());
            HIPCK(c, rocprim::radix_sort_pairs<SortCfg>(b.temp.p, temp_bytes, b.keys.as<unsigned long long>(), b.keys2.as<unsigned long long>(), perm, perm2, (size_t)nv, 0, 32 + d, c->stream));
            int* t = perm; perm = perm2; perm2 = t;
        }
    }
    {
        const bool nrm = b.attrs && b.attrs->has_normals, col = b.attrs && b.attrs->has_colors;
        hipLaunchKernelGGL(k_bvh_gather<DIM>, dim3((n_slots + 255) / 256), dim3(256), 0, c->stream, cp,
                           nrm ? b.attrs->nx.as<float>() : nullptr, nrm ? b.attrs->ny.as<float>() : nullptr, nrm ? b.attrs->nz.as<float>() : nullptr,
                           col ? b.attrs->rgba.as<uint32_t>() : nullptr, perm, nv, n_slots, b.leaves.as<BvhLeafT<DIM>>(), b.recs.as<TgtRec>(), b.pos_of.as<int>());
    }
    for (int d = depth - 1; d >= 0; d--) {
        const int count = 1 << d, first = count - 1;
        hipLaunchKernelGGL(k_bvh_nodes<DIM>, dim3((count + 255) / 256), dim3(256), 0, c->stream, b.leaves.as<BvhLeafT<DIM>>(), b.n_leaves, b.Lp, first, count,
                           d == depth - 1 ? 1 : 0, b.nodes.as<BvhNodeT<DIM>>());
    }
    {   // 4-wide view of the same tree (two binary levels per step) for the 1-NN walk
        const int pad = depth & 1;
        b.Lq = (depth + pad) / 2;
        const long long nq = ((1ll << (2 * b.Lq)) - 1) / 3;
        if ((rc = ensure(c, b.qnodes, (size_t)(nq > 0 ? nq : 1) * sizeof(BvhQuadT<DIM>)))) return rc;
        if (nq > 0) hipLaunchKernelGGL(k_bvh_quad_nodes<DIM>, dim3((unsigned)((nq * 4 + 255) / 256)), dim3(256), 0, c->stream, b.nodes.as<BvhNodeT<DIM>>(), pad, b.Lq, b.qnodes.as<BvhQuadT<DIM>>());
    }
    HIPCK(c, hipGetLastError());
    HIPCK(c, hipEventRecord(e1, c->stream));
    HIPCK(c, hipEventSynchronize(e1));
    float ms = 0; HIPCK(c, hipEventElapsedTime(&ms, e0, e1)); b.build_ms = ms;
    b.valid = true;
    return ICP_OK;
}

CoordPtrs<3> target_coords3(const icp_ctx* c) { CoordPtrs<3> cp; cp.c[0] = c->tgt.x.as<float>(); cp.c[1] = c->tgt.y.as<float>(); cp.c[2] = c->tgt.z.as<float>(); return cp; }
CoordPtrs<6> target_coords6(const icp_ctx* c) {
    CoordPtrs<6> cp; cp.c[0] = c->tgt.x.as<float>(); cp.c[1] = c->tgt.y.as<float>(); cp.c[2] = c->tgt.z.as<float>();
    cp.c[3] = c->tgt.cr.as<float>(); cp.c[4] = c->tgt.cg.as<float>(); cp.c[5] = c->tgt.cb.as<float>(); return cp;
}

PostParams make_post_params(icp_ctx* c, const Cloud& src, const int* sel, int n) {
    const icp_params& p = c->prm;
    PostParams pp;
    pp.sx = src.x.as<float>(); pp.sy = src.y.as<float>(); pp.sz = src.z.as<float>();
    pp.snx = src.nx.as<float>(); pp.sny = src.ny.as<float>(); pp.snz = src.nz.as<float>();
    pp.srgba = src.rgba.as<uint32_t>(); pp.sel = sel; pp.n = n;
    pp.tx = c->tgt.x.as<float>(); pp.ty = c->tgt.y.as<float>(); pp.tz = c->tgt.z.as<float>();
    pp.tnx = c->tgt.nx.as<float>(); pp.tny = c->tgt.ny.as<float>(); pp.tnz = c->tgt.nz.as<float>(); pp.trgba = c->tgt.rgba.as<uint32_t>();
    pp.ps = c->ps.as<PoseState>(); pp.matches = c->matches.as<icp_match_t>();
    pp.metric = p.metric; pp.weighting = p.weighting; pp.rejection = p.rejection;
    pp.max_dist = p.max_distance; pp.cos_reject = c->cos_reject; pp.partials = c->partials.as<double>();
    return pp;
}

// The outboxes of the hand-over between blocks (GX): armed for `blocks` blocks -- granules all-ones, claim words and headers zero.
__global__ void k_gx_init(unsigned long long* slots, size_t n_granules, unsigned int* hdr, size_t n_hdr) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_granules) slots[i] = (i % GX_GRANULES == 4) ? 0ull : GX_EMPTY;
    if (i < n_hdr) hdr[i] = 0u;
}
static int ensure_gx(icp_ctx* c, int blocks) {
    if (!c->gx_on || c->shared_gpu) return ICP_OK;
    int rc;
    if (blocks > c->gx_blocks || c->gx_dirty) {
        const int nbk = blocks > c->gx_blocks ? blocks : c->gx_blocks;
        const size_t ng = (size_t)nbk * GX_SLOTS * GX_GRANULES;
        if ((rc = ensure(c, c->gx_slots, ng * 8))) return rc;
        if ((rc = ensure(c, c->gx_hdr, (size_t)nbk * 8))) return rc;
        hipLaunchKernelGGL(k_gx_init, dim3((unsigned int)((ng + 255) / 256)), dim3(256), 0, c->stream, c->gx_slots.as<unsigned long long>(), ng, c->gx_hdr.as<unsigned int>(), (size_t)nbk * 2);
        HIPCK(c, hipGetLastError());
        c->gx_blocks = nbk; c->gx_dirty = false;
    }
    return ICP_OK;
}

// fuse != nullptr: run the post stage (weight / reject / accumulate) as the epilogue of the search; *fused_blocks receives the
// number of block partials written.
template <int DIM>
int launch_bvh_query(icp_ctx* c, Bvh& b, const CoordPtrs<DIM>& cp, const KnnParams& kp, const int* order, int n, const Cloud* fuse, int* fused_blocks, const MergeLaunch* ml = nullptr) {
    int rc;
    if (!b.valid && (rc = build_bvh<DIM>(c, b, cp))) return rc;
    BvhViewT<DIM> bv; bv.leaves = b.leaves.as<BvhLeafT<DIM>>(); bv.nodes = b.nodes.as<BvhNodeT<DIM>>(); bv.n_valid = b.n_valid; bv.Lp = b.Lp; bv.tgt = cp;
    bv.qnodes = b.qnodes.as<BvhQuadT<DIM>>(); bv.Lq = b.Lq; bv.recs = b.recs.as<TgtRec>(); bv.pos_of = b.pos_of.as<int>();
    const int nb = fuse ? fused_nblocks(n) : (n + BVH_THREADS - 1) / BVH_THREADS;
    const size_t stack_bytes = (size_t)(ICP_SHARE_WALKS ? ICP_SHARE_ROWS : 1) * BVH_THREADS * 8;          // the shared walk's records in LDS
    if (fuse) {
        if ((rc = ensure(c, c->partials, (size_t)(nb > POST_BLOCKS ? nb : POST_BLOCKS) * NSUM * 8))) return rc;
        PostParams pp = make_post_params(c, *fuse, kp.sel, n);
        KnnParams kf = kp; kf.out = nullptr;
        if (DIM == 3 && c->gx_on && !c->shared_gpu && !(ml && ml->loop)) {
            if ((rc = ensure_gx(c, nb))) return rc;
            kf.gx = GxParams{c->gx_slots.as<unsigned long long>(), c->gx_hdr.as<unsigned int>(), c->gx_start, c->gx_empty};
        }
        if (!c->keep_fused_records) { pp.matches = nullptr; kf.d2_out = nullptr; }     // the loop never reads the records of a fused iteration, nor the distances
        const size_t red_bytes = (size_t)(BVH_THREADS / WAVE) * 33 * 8;           // the reduction reuses the (dead) traversal stacks
        static const size_t lds_pad = getenv("ICP_HIP_LDS_PAD") ? (size_t)atoi(getenv("ICP_HIP_LDS_PAD")) : 0;      // development: fewer resident blocks per CU
        const size_t lds = (stack_bytes > red_bytes ? stack_bytes : red_bytes) + xw_lds_bytes<DIM, BVH_THREADS>() + lds_pad;      // + the board of the cross-wave hand-over
        if (ml && ml->loop) {                                                      // the level's whole loop in one launch
            LoopK<DIM> K; memset(&K, 0, sizeof(K));
            // (the level's planes and the search state are packed: get_sorted_level / launch_match; anything else cannot take this path)
            const float* x0 = fuse->x.as<float>(); const long long S = fuse->y.as<float>() - x0;
            const bool packed_src = S > 0 && fuse->z.as<float>() == x0 + 2 * S && fuse->nx.as<float>() == x0 + 3 * S && fuse->ny.as<float>() == x0 + 4 * S && fuse->nz.as<float>() == x0 + 5 * S &&
                                    (!fuse->cr.p || (fuse->cr.as<float>() == x0 + 6 * S && fuse->cg.as<float>() == x0 + 7 * S && fuse->cb.as<float>() == x0 + 8 * S && fuse->rgba.as<float>() == x0 + 9 * S));
            const bool packed_q = kf.nn_raw == c->qpack.as<int>() && (!kf.qstate || (char*)kf.qstate == c->qpack.as<char>() + c->q_cap * 4) && (!kf.qstate2 || (char*)kf.qstate2 == c->qpack.as<char>() + c->q_cap * 20);
            if (!packed_src || !packed_q || kf.sel || order) { c->err = "k_icp_loop: level or search state not in the packed layout"; return ICP_ERR_INVALID_ARG; }
            K.src = x0; K.src_stride = (int)S; K.qpack = c->qpack.as<char>(); K.q_cap = (int)c->q_cap;
            K.leaves = bv.leaves; K.qnodes = bv.qnodes; K.recs = bv.recs; K.Lq = bv.Lq; K.n_valid = bv.n_valid;
            K.matches = pp.matches; K.d2_out = kf.d2_out; K.dbg_steps = kf.dbg_steps;
            K.n = kf.n; K.max_dist = kf.max_dist; K.incremental = kf.incremental; K.tier2 = kf.qstate2 ? 1 : 0;
            K.metric = pp.metric; K.weighting = pp.weighting; K.rejection = pp.rejection; K.cos_reject = pp.cos_reject;
            K.L = *ml->loop; K.L.nb = nb;
            const size_t lds_loop = (size_t)LOOP_LDS_ROWS * BVH_THREADS * 8;
            if (b.Lq <= 8) hipLaunchKernelGGL((k_icp_loop<DIM, false>), dim3(nb), dim3(BVH_THREADS), lds_loop, c->stream, K);
            else hipLaunchKernelGGL((k_icp_loop<DIM, true>), dim3(nb), dim3(BVH_THREADS), lds_loop, c->stream, K);
        }
        else if (ml) {                                                             // merged loop: reducer blocks in front, pose through the ring
            kf.ps = ml->slot; pp.ps = ml->slot; pp.partials = ml->partials; kf.fault = ml->rp.run_fault;
            if (ml->ev_start) {
                if (b.Lq <= 8) hipExtLaunchKernelGGL((k_knn_bvh_post_ring<DIM, false>), dim3(nb + ml->rp.n_red), dim3(BVH_THREADS), (uint32_t)lds, c->stream, ml->ev_start, ml->ev_stop, 0, kf, bv, order, pp, ml->rp);
                else hipExtLaunchKernelGGL((k_knn_bvh_post_ring<DIM, true>), dim3(nb + ml->rp.n_red), dim3(BVH_THREADS), (uint32_t)lds, c->stream, ml->ev_start, ml->ev_stop, 0, kf, bv, order, pp, ml->rp);
            }
            else if (b.Lq <= 8) hipLaunchKernelGGL((k_knn_bvh_post_ring<DIM, false>), dim3(nb + ml->rp.n_red), dim3(BVH_THREADS), lds, c->stream, kf, bv, order, pp, ml->rp);
            else hipLaunchKernelGGL((k_knn_bvh_post_ring<DIM, true>), dim3(nb + ml->rp.n_red), dim3(BVH_THREADS), lds, c->stream, kf, bv, order, pp, ml->rp);
        }
        else if (b.Lq <= 8) hipLaunchKernelGGL((k_knn_bvh_post<DIM, false>), dim3(nb), dim3(BVH_THREADS), lds, c->stream, kf, bv, order, pp);
        else hipLaunchKernelGGL((k_knn_bvh_post<DIM, true>), dim3(nb), dim3(BVH_THREADS), lds, c->stream, kf, bv, order, pp);
        *fused_blocks = nb;
    } else {
        hipLaunchKernelGGL(k_knn_bvh<DIM>, dim3(nb), dim3(BVH_THREADS), stack_bytes, c->stream, kp, bv, order);
    }
    HIPCK(c, hipGetLastError());
    return ICP_OK;
}

// Enqueue the matching stage (no sync).  fused_blocks != nullptr allows the BVH matcher to run the post stage as its epilogue;
// it is set to the number of block partials written, or left 0 when the matcher in use does not fuse.
int launch_match(icp_ctx* c, const QuerySet& q, int* fused_blocks = nullptr, const MergeLaunch* ml = nullptr) {
    const icp_params& p = c->prm;
    int rc;
    if (fused_blocks) *fused_blocks = 0;
    if ((rc = ensure_match_buffers(c, q.n))) return rc;
    if (p.matching == ICP_MATCH_PROJECTIVE) {
        ProjParams pp;
        pp.sx = q.cl->x.as<float>(); pp.sy = q.cl->y.as<float>(); pp.sz = q.cl->z.as<float>(); pp.sel = q.sel; pp.n = q.n;
        pp.tx = c->tgt.x.as<float>(); pp.ty = c->tgt.y.as<float>(); pp.tz = c->tgt.z.as<float>();
        pp.width = p.width; pp.height = p.height; pp.fx = p.fx; pp.fy = p.fy; pp.mx = p.cx; pp.my = p.cy; pp.window = 12;   // NearestNeighbor.h:319
        pp.ps = c->ps.as<PoseState>(); pp.pretransformed = q.pretransformed; pp.max_dist = p.max_distance;
        pp.out = c->matches.as<icp_match_t>(); pp.d2_out = c->d2.as<float>();
        hipLaunchKernelGGL(k_projective, dim3((q.n + 255) / 256), dim3(256), 0, c->stream, pp);
        HIPCK(c, hipGetLastError());
        return ICP_OK;
    }
    KnnParams kp;
    kp.sx = q.cl->x.as<float>(); kp.sy = q.cl->y.as<float>(); kp.sz = q.cl->z.as<float>();
    kp.scr = q.cl->cr.as<float>(); kp.scg = q.cl->cg.as<float>(); kp.scb = q.cl->cb.as<float>();
    kp.sel = q.sel; kp.n = q.n;
    kp.tx = c->tgt.x.as<float>(); kp.ty = c->tgt.y.as<float>(); kp.tz = c->tgt.z.as<float>();
    kp.tcr = c->tgt.cr.as<float>(); kp.tcg = c->tgt.cg.as<float>(); kp.tcb = c->tgt.cb.as<float>();
    kp.mpad = c->tgt.npad; kp.ps = c->ps.as<PoseState>(); kp.pretransformed = q.pretransformed; kp.max_dist = p.max_distance;
    kp.out = c->matches.as<icp_match_t>(); kp.d2_out = c->d2.as<float>(); kp.best64 = nullptr; kp.nn_raw = nullptr; kp.use_prev = 0; kp.qstate = nullptr; kp.qstate2 = nullptr; kp.incremental = 0; kp.dbg_steps = nullptr; kp.dbg_waves = 0; kp.fault = &c->ps.as<PoseState>()->fault; kp.gx = GxParams{nullptr, nullptr, 0, 0};
    if (p.knn_backend == ICP_KNN_LBVH) {
        kp.nseg = 1;
        // neighbour positions and the incremental search's state in ONE allocation, sections a fixed number of elements apart
        // (int nn_raw[q_cap] | float4 qstate[q_cap] | float2 qstate2[q_cap]): k_icp_loop is handed a base and a stride
        if ((rc = ensure_qpack(c, q.n))) return rc;
        kp.nn_raw = c->nn_raw.as<int>(); kp.use_prev = q.seed_prev ? 1 : 0;
#if ICP_DEBUG_STEPS
        if ((rc = ensure(c, c->dbg_steps, (size_t)q.n * 4))) return rc;
        kp.dbg_steps = c->dbg_steps.as<int>(); kp.dbg_waves = fused_nblocks(q.n) * (BVH_THREADS / WAVE);
#endif
        if (p.knn_incremental && !q.pretransformed) {
            kp.qstate = c->qstate.as<float4>(); kp.qstate2 = c->tier2 ? c->qstate2.as<float2>() : nullptr; kp.incremental = 1;
        }
        const Cloud* fuse = (fused_blocks != nullptr && p.metric != ICP_METRIC_SYMMETRIC && !q.pretransformed) ? q.cl : nullptr;
        if (q.use_colors) return launch_bvh_query<6>(c, c->bvh6, target_coords6(c), kp, q.order, q.n, fuse, fused_blocks, fuse ? ml : nullptr);
        return launch_bvh_query<3>(c, c->bvh, target_coords3(c), kp, q.order, q.n, fuse, fused_blocks, fuse ? ml : nullptr);
    }
    const int bx = (q.n + WAVE - 1) / WAVE;
    const int nch = kp.mpad / KNN_CH;
    int nseg = 1;
    if (bx < 1024) { nseg = (2048 + bx - 1) / bx; if (nseg > nch / 4) nseg = nch / 4; if (nseg < 1) nseg = 1; }
    kp.nseg = nseg;
    if (nseg > 1) {
        if ((rc = ensure(c, c->best64, (size_t)q.n * 8))) return rc;
        kp.best64 = c->best64.as<unsigned long long>();
        const unsigned long long init = ((unsigned long long)0x7F7FFFFFu << 32) | 0xFFFFFFFFull;   // (FLT_MAX, idx -1)
        hipLaunchKernelGGL(k_fill_u64, dim3((q.n + 255) / 256), dim3(256), 0, c->stream, kp.best64, q.n, init);
    }
    if (q.use_colors) hipLaunchKernelGGL(k_knn_brute<6>, dim3(bx, nseg), dim3(256), 0, c->stream, kp);
    else              hipLaunchKernelGGL(k_knn_brute<3>, dim3(bx, nseg), dim3(256), 0, c->stream, kp);
    if (nseg > 1)
        hipLaunchKernelGGL(k_knn_finalize, dim3((q.n + 255) / 256), dim3(256), 0, c->stream, kp.best64, q.n, p.max_distance,
                           c->matches.as<icp_match_t>(), c->d2.as<float>());
    HIPCK(c, hipGetLastError());
    return ICP_OK;
}

// The hand-over slots of k_reduce_solve (NSUM self-validating totals + the ticket) back to "nothing written": enqueued at the start of
// every entry point that launches it, so that whatever an earlier call left behind -- a run cut short by a HIP error between a
// block's publish and block 0's re-arm, a total that arrived after block 0 had given up -- can never be taken for a result.
int rearm_handover(icp_ctx* c) {
    int rc;
    if ((rc = ensure(c, c->totals, NSUM * 8 + 8))) return rc;
    HIPCK(c, hipMemsetAsync(c->totals.p, 0, NSUM * 8 + 8, c->stream));
    if (c->spin_reduce) hipLaunchKernelGGL(k_fill_u64, dim3(1), dim3(64), 0, c->stream, c->totals.as<unsigned long long>(), NSUM, TOTAL_SENTINEL);
    HIPCK(c, hipGetLastError());
    return ICP_OK;
}

// Enqueue weight + reject + accumulate (+ symmetric second pass) + reduce/solve (no sync).
int launch_post_and_solve(icp_ctx* c, const Cloud& src, const int* sel, int n, icp_iter_stats* d_stats, double* d_sums_out, int update_pose,
                          hipEvent_t ev_after_post, int fused_blocks = 0) {
    const icp_params& p = c->prm;
    int rc;
    if (!fused_blocks && (rc = ensure(c, c->partials, (size_t)POST_BLOCKS * NSUM * 8))) return rc;
    if (!c->totals.p && (rc = rearm_handover(c))) return rc;      // (the entry points re-arm before their first launch; this covers a first use)
    const PostParams pp = make_post_params(c, src, sel, n);
    int nb = (n + POST_THREADS - 1) / POST_THREADS; if (nb > POST_BLOCKS) nb = POST_BLOCKS; if (nb < 1) nb = 1;
    if (fused_blocks) nb = fused_blocks;                    // the matcher already wrote the block partials
    else hipLaunchKernelGGL(k_post, dim3(nb), dim3(POST_THREADS), 0, c->stream, pp);
    SolveParams sp; memset(&sp, 0, sizeof(sp));
    sp.partials = c->partials.as<double>(); sp.nblocks = nb; sp.ps = c->ps.as<PoseState>(); sp.metric = p.metric;
    sp.totals = c->totals.as<double>(); sp.ticket = (unsigned*)(c->totals.as<double>() + NSUM);
    sp.n_src = n; sp.update_pose = update_pose; sp.spin = c->spin_reduce ? 1 : 0;
    auto reduce_solve = [&]() { hipLaunchKernelGGL(k_reduce_solve, dim3(NSUM_USED), dim3(SOLVE_THREADS), 0, c->stream, sp); };
    if (p.metric == ICP_METRIC_SYMMETRIC) {
        sp.phase = 0; sp.stats = nullptr; sp.sums_out = nullptr;
        reduce_solve();                                                                              // means
        hipLaunchKernelGGL(k_sym_accumulate, dim3(nb), dim3(POST_THREADS), 0, c->stream, pp);
        if (ev_after_post) HIPCK(c, hipEventRecord(ev_after_post, c->stream));
        sp.phase = 1; sp.stats = d_stats; sp.sums_out = d_sums_out;
        reduce_solve();
    } else {
        if (ev_after_post) HIPCK(c, hipEventRecord(ev_after_post, c->stream));
        sp.phase = 0; sp.stats = d_stats; sp.sums_out = d_sums_out;
        reduce_solve();
    }
    HIPCK(c, hipGetLastError());
    return ICP_OK;
}

// How many blocks of k_icp_loop the device holds at once (its waiters wait for blocks of the same grid: the whole grid must be resident).
template <int DIM, bool WIDE>
int loop_capacity_of(icp_ctx* c, int* out) {
    int& cap = c->loop_capacity[(DIM == 6 ? 2 : 0) + (WIDE ? 1 : 0)];
    if (cap == 0) {
        int per_cu = 0, cus = 0;
        const size_t lds = (size_t)LOOP_LDS_ROWS * BVH_THREADS * 8;
        HIPCK(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)&k_icp_loop<DIM, WIDE>, BVH_THREADS, lds));
        HIPCK(c, hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device));
        cap = per_cu * cus > 0 ? per_cu * cus : -1;
    }
    *out = cap;
    return ICP_OK;
}
// One k_icp_loop at a time per process: two resident grids that each wait for blocks of their own that the other keeps from being
// dispatched would wait for each other (bounded, but seconds).  A context that does not get the token runs one launch per iteration.
std::atomic<int> g_loop_token{0};
struct LoopToken {
    bool held = false;
    bool try_take() { int expect = 0; held = g_loop_token.compare_exchange_strong(expect, 1); return held; }
    ~LoopToken() { if (held) g_loop_token.store(0); }
};

int check_ready(icp_ctx* c, bool need_source, bool full_pipeline) {
    const icp_params& p = c->prm;
    if (c->tgt.n <= 0) { c->err = "target index needs to be built before querying (icp_set_target)"; return ICP_ERR_NO_TARGET; }
    if (need_source && c->src.n <= 0) { c->err = "no source cloud (icp_set_source)"; return ICP_ERR_NO_SOURCE; }
    if (p.matching == ICP_MATCH_PROJECTIVE) {
        if (p.height <= 0 || p.width <= 0) { c->err = "set camera params before querying any matches"; return ICP_ERR_NO_CAMERA; }
        if ((long long)p.width * p.height != c->tgt.n) { c->err = "invalid size of target points (must be width*height)"; return ICP_ERR_TARGET_SIZE; }
    } else if (p.color_icp) {
        if (!c->tgt.has_colors || (need_source && !c->src.has_colors)) { c->err = "colour ICP needs colours on target and source"; return ICP_ERR_COLOR_MISMATCH; }
    }
    if (full_pipeline) {
        if (!c->tgt.has_normals || !c->src.has_normals) { c->err = "normals required on source and target"; return ICP_ERR_INVALID_ARG; }
        if (p.weighting == ICP_WEIGHT_COLORS && (!c->tgt.has_colors || !c->src.has_colors)) { c->err = "colour weighting needs colours"; return ICP_ERR_COLOR_MISMATCH; }
    }
    return ICP_OK;
}

// Selection for a decimation factor: PointCloud::getCoarseResolution (PointCloud.h:325-343).
int get_level(icp_ctx* c, int factor, const int** d_idx, int* n_out, const int** d_order) {
    auto it = c->levels.find(factor);
    if (it == c->levels.end()) {
        Level lv;
        int rc;
        if (factor > 0) {
            const int count = (c->src.n + factor - 1) / factor;              // candidates i = 0, factor, 2 factor, ... (PointCloud.h:331)
            if ((rc = ensure(c, c->staging, (size_t)count))) return rc;
            hipLaunchKernelGGL(k_stride_flags, dim3((count + 255) / 256), dim3(256), 0, c->stream, c->src_flag.as<uint8_t>(), c->src.n, factor, count, c->staging.as<uint8_t>());
            HIPCK(c, hipGetLastError());
            if ((rc = compact_flagged(c, c->staging.as<uint8_t>(), count, factor, lv.idx, &lv.n))) return rc;
        } else lv.n = c->src.n;                                   // factor 0: every point, no index list
        it = c->levels.emplace(factor, lv).first;
    }
    *d_idx = it->second.idx.as<int>(); *n_out = it->second.n;
    if (d_order) {
        *d_order = nullptr;
        if (it->second.n > 0) {
            if (!it->second.order.p) { int rc; if ((rc = build_query_order(c, it->second.idx.as<int>(), it->second.n, it->second.order))) return rc; }
            *d_order = it->second.order.as<int>();
        }
    }
    return ICP_OK;
}

// Morton order of the whole resident source for the stage-level entry points (results stay in source order); nullptr
// when the BVH matcher is not in use.
int get_full_order(icp_ctx* c, const int** out) {
    *out = nullptr;
    const icp_params& p = c->prm;
    if (!(p.matching == ICP_MATCH_KNN && p.knn_backend == ICP_KNN_LBVH) || c->src.n <= 0) return ICP_OK;
    const int* idx; int n;
    return get_level(c, 0, &idx, &n, out);
}

// The level's points physically permuted into Morton order (built once per icp_set_source and level).
int get_sorted_level(icp_ctx* c, int factor, const Cloud** cloud, int* n_out) {
    const int* d_idx; const int* d_order; int n, rc;
    if ((rc = get_level(c, factor, &d_idx, &n, &d_order))) return rc;
    Level& lv = c->levels[factor];
    *n_out = n;
    if (!lv.sorted_valid && n > 0) {
        if ((rc = ensure(c, lv.sorted_idx, (size_t)n * 4))) return rc;
        const dim3 g((n + 255) / 256), b(256);
        hipLaunchKernelGGL(k_compose_idx, g, b, 0, c->stream, d_idx, d_order, n, lv.sorted_idx.as<int>());
        const int* si = lv.sorted_idx.as<int>();
        Cloud& d = lv.sorted; const Cloud& s = c->src;
        d.n = n; d.npad = n; d.has_normals = s.has_normals; d.has_colors = s.has_colors;
        DevBuf* dst[9] = {&d.x, &d.y, &d.z, &d.nx, &d.ny, &d.nz, &d.cr, &d.cg, &d.cb};
        const DevBuf* srcp[9] = {&s.x, &s.y, &s.z, &s.nx, &s.ny, &s.nz, &s.cr, &s.cg, &s.cb};
        const size_t stride = ((size_t)n + 63) / 64 * 64;                         // elements between two planes
        if ((rc = ensure(c, lv.pack, 10 * stride * 4))) return rc;
        for (int k = 0; k < 9; k++) set_view(*dst[k], lv.pack.as<float>() + (size_t)k * stride, stride * 4);
        set_view(d.rgba, lv.pack.as<float>() + 9 * stride, stride * 4);
        for (int k = 0; k < 9; k++) {
            if (!srcp[k]->p) continue;
            hipLaunchKernelGGL(k_gather_f32, g, b, 0, c->stream, srcp[k]->as<float>(), si, n, dst[k]->as<float>());
        }
        if (s.rgba.p) hipLaunchKernelGGL(k_gather_u32, g, b, 0, c->stream, s.rgba.as<uint32_t>(), si, n, d.rgba.as<uint32_t>());
        HIPCK(c, hipGetLastError());
        lv.sorted_valid = true;
    }
    *cloud = &lv.sorted;
    return ICP_OK;
}

int ensure_events(icp_ctx* c, size_t count) {
    while (c->events.size() < count) { hipEvent_t e; HIPCK(c, hipEventCreate(&e)); c->events.push_back(e); }
    return ICP_OK;
}

}  // namespace

extern "C" {

const char* icp_version(void) { return "icp_hip gfx950 r2"; }

uint32_t icp_select_hash(uint32_t seed, uint32_t iteration, uint32_t index) { return select_hash(seed, iteration, index); }

int icp_params_default(icp_params* p) {
    if (!p) return ICP_ERR_INVALID_ARG;
    memset(p, 0, sizeof(*p));
    p->metric = 0; p->matching = 0; p->weighting = 0; p->rejection = 1; p->color_icp = 0; p->multires = 0;   // ICPOptimizer.h:29-31
    p->n_iterations = 20; p->max_distance = 0.0003f;
    p->knn_backend = ICP_KNN_BRUTE_FORCE; p->record_rmse = 0;
    p->knn_incremental = 1;
    p->selection = 0; p->selection_proba = 1.0f; p->selection_seed = 0u;      // setSelectionMethod(SELECT_ALL), proba default 1.0 (ICPOptimizer.h:58)
    return ICP_OK;
}

int icp_ctx_create_on_stream(int device, void* hip_stream, icp_ctx** out) {
    if (!out) return ICP_ERR_INVALID_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return ICP_ERR_NO_DEVICE;
    if (device < 0 || device >= ndev) return ICP_ERR_INVALID_ARG;
    icp_ctx* c = new icp_ctx();
    c->device = device;
    icp_params_default(&c->prm);
    memset(&c->timing, 0, sizeof(c->timing));
    if (hipSetDevice(device) != hipSuccess) { delete c; return ICP_ERR_HIP; }
    { const char* e = getenv("ICP_HIP_FUSE_POST"); if (e && e[0] == '0') c->fuse_post = false; }
    { const char* e = getenv("ICP_HIP_SPIN_REDUCE"); if (e) c->spin_reduce = e[0] == '1'; }
    { const char* e = getenv("ICP_HIP_TIER2"); if (e && e[0] == '0') c->tier2 = false; }
    { const char* e = getenv("ICP_HIP_MERGE"); if (e && e[0] == '0') c->merge_loop = false; }
    { const char* e = getenv("ICP_HIP_GX"); if (e && e[0] == '0') c->gx_on = false; }
    { const char* e = getenv("ICP_HIP_GX_START"); if (e && atoi(e) > 1) c->gx_start = atoi(e); }
    { const char* e = getenv("ICP_HIP_GX_EMPTY"); if (e && atoi(e) > 0) c->gx_empty = atoi(e); }
    { const char* e = getenv("ICP_HIP_PERSIST"); if (e) c->persist_loop = e[0] == '1'; }
    { const char* e = getenv("ICP_HIP_EXT_EVENTS"); if (e && e[0] == '0') c->ext_events = false; }
    { const char* e = getenv("ICP_HIP_LOOP_FROM"); if (e) c->loop_from = atoi(e); }
    { const char* e = getenv("ICP_HIP_PRESORT"); if (e && e[0] == '0') c->presort = false; }
    { const char* e = getenv("ICP_HIP_BLOCK_LEVELS"); if (e && e[0] == '0') c->block_levels = false; }
    { const char* e = getenv("ICP_HIP_TRACE"); if (e && e[0] == '1') c->trace = true; }
    { const char* e = getenv("ICP_HIP_STAGE_EVENTS"); if (e && e[0] >= '0' && e[0] <= '9') c->stage_timing = atoi(e); }
    if (hip_stream) { c->stream = (hipStream_t)hip_stream; c->owns_stream = false; }
    else {
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return ICP_ERR_HIP; }
        c->owns_stream = true;
    }
    c->cos_reject = compute_cos_reject();
    float ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    int rc = write_pose(c, ident);
    if (!rc && hipStreamSynchronize(c->stream) != hipSuccess) rc = ICP_ERR_HIP;
    if (rc) { icp_ctx_destroy(c); return rc; }
    *out = c;
    return ICP_OK;
}
int icp_ctx_create(int device, icp_ctx** out) { return icp_ctx_create_on_stream(device, nullptr, out); }

int icp_ctx_destroy(icp_ctx* c) {
    if (!c) return ICP_ERR_INVALID_ARG;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    release(c->tgt); release(c->src); release(c->qry); release(c->conv_src); release(c->conv_ref);
    release(c->nrm_cloud);
    for (Bvh* b : {&c->bvh, &c->bvh6, &c->nrm_bvh}) { release(b->qnodes); release(b->recs); for (DevBuf& d : b->axl) release(d); release(b->side); release(b->scanr); release(b->axis_of_node); }
    for (Bvh* b : {&c->bvh6, &c->nrm_bvh}) { release(b->keys); release(b->keys2); release(b->vals); release(b->vals2); release(b->temp); release(b->leaves); release(b->nodes); release(b->lvl); release(b->wbox); }
    release(c->bvh.keys); release(c->bvh.keys2); release(c->bvh.vals); release(c->bvh.vals2); release(c->bvh.temp); release(c->bvh.leaves); release(c->okeys); release(c->okeys2); release(c->ovals); release(c->otemp); release(c->bvh.nodes); release(c->bvh.lvl); release(c->bvh.wbox);
    for (auto& kv : c->levels) release(kv.second);
    release(c->ps); release(c->matches); release(c->d2); release(c->best64); release(c->nn_raw); release(c->qstate); release(c->qstate2); release(c->qpack); release(c->sel_lists); release(c->sel_counts); release(c->sel_blocks); release(c->partials); release(c->partials2); release(c->ring); release(c->pring); release(c->totals); release(c->dbg_steps); release(c->sums); release(c->gx_slots); release(c->gx_hdr);
    release(c->stats); release(c->staging); release(c->rmse_partials); release(c->rmse_out); release(c->fontana_partials);
    for (DevBuf* d : {&c->src_flag, &c->src_box, &c->tgt_flag, &c->tgt_finite, &c->nrm_finite, &c->sel_temp, &c->d_count}) release(*d);
    if (c->pin_up) (void)hipHostFree(c->pin_up);
    if (c->up_ev) (void)hipEventDestroy(c->up_ev);
    if (c->pinned) (void)hipHostFree(c->pinned);
    for (hipEvent_t e : c->events) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->build_ev) if (e) (void)hipEventDestroy(e);
    if (c->stream2) { (void)hipStreamSynchronize(c->stream2); (void)hipStreamDestroy(c->stream2); }
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->owns_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return ICP_OK;
}

const char* icp_last_error(const icp_ctx* c) { return c ? c->err.c_str() : "null context"; }

int icp_set_params(icp_ctx* c, const icp_params* p) {
    if (!c || !p) return ICP_ERR_INVALID_ARG;
    if (p->metric < 0 || p->metric > 2 || p->matching < 0 || p->matching > 1 || p->weighting < 0 || p->weighting > 3 || p->n_iterations < 0 || p->selection < 0 || p->selection > 1 ||
        (p->knn_backend != ICP_KNN_BRUTE_FORCE && p->knn_backend != ICP_KNN_LBVH) || p->width < 0 || p->height < 0 || (long long)p->width * p->height > 0x7FFFFFFFll ||
        std::isnan(p->max_distance) || std::isnan(p->selection_proba) || !std::isfinite(p->fx) || !std::isfinite(p->fy) || !std::isfinite(p->cx) || !std::isfinite(p->cy)) {
        c->err = "icp_set_params: value out of range"; return ICP_ERR_INVALID_ARG;
    }
    c->prm = *p;
    return ICP_OK;
}
int icp_get_params(const icp_ctx* c, icp_params* p) { if (!c || !p) return ICP_ERR_INVALID_ARG; *p = c->prm; return ICP_OK; }

int icp_set_target(icp_ctx* c, const float* xyz, const float* normals, const uint8_t* rgba, int32_t n) {
    if (!c || !xyz || n <= 0) { if (c) c->err = "icp_set_target: null points or n <= 0"; return ICP_ERR_INVALID_ARG; }
    int rc;
    DrainOnError guard(c);
    if ((rc = set_device(c))) return rc;
    if ((rc = upload_cloud(c, c->tgt, xyz, normals, rgba, n, true))) return rc;
    Bvh& b = c->bvh;
    c->bvh6.valid = false;
    b.valid = false; b.n_valid = 0;
    // non-finite targets can never win the strict-< argmin: they stay out of the tree (filter + compaction on the device)
    if ((rc = finite_list(c, c->tgt, false, c->tgt_flag, c->tgt_finite, &b.n_valid))) return rc;
    b.d_finite = c->tgt_finite.as<int>(); b.n_ids = n;
    c->bvh6.d_finite = b.d_finite; c->bvh6.n_valid = b.n_valid; c->bvh6.n_ids = n;
    b.attrs = &c->tgt; c->bvh6.attrs = &c->tgt;
    if (c->prm.knn_backend == ICP_KNN_LBVH && c->prm.matching == ICP_MATCH_KNN) {                         // buildIndex; otherwise built on first use
        if (c->prm.color_icp && rgba) return guard.done(build_bvh<6>(c, c->bvh6, target_coords6(c)));
        return guard.done(build_bvh<3>(c, b, target_coords3(c)));
    }
    return guard.done();
}

// Not part of icp_hip.h (icp_batch_run's own): the resident SOURCE becomes the target -- what icp_set_target(the same arrays) would leave,
// without the trip through the host: consecutive scan pairs (k, k + 1) share scan k + 1, the source of pair k and the target of pair
// k + 1 (main.cpp:411-498 loads it twice).  Plane copies on the device (+inf padding as upload_cloud does), then the same finite filter
// and index build.  Colours are not carried (ICP_ERR_INVALID_ARG when colour ICP is on).
int icp_internal_promote_source_to_target(icp_ctx* c) {
    if (!c) return ICP_ERR_INVALID_ARG;
    if (c->src.n <= 0) { c->err = "promote: no source cloud"; return ICP_ERR_NO_SOURCE; }
    if (c->prm.color_icp || c->prm.weighting == ICP_WEIGHT_COLORS) { c->err = "promote: colours are not carried"; return ICP_ERR_INVALID_ARG; }
    int rc;
    DrainOnError guard(c);
    if ((rc = set_device(c))) return rc;
    const Cloud& sc = c->src; Cloud& tg = c->tgt;
    const int n = sc.n, npad = ((n + 63) / 64) * 64;
    for (DevBuf* pl : {&tg.x, &tg.y, &tg.z}) if ((rc = ensure(c, *pl, (size_t)npad * 4))) return rc;
    if (sc.has_normals) for (DevBuf* pl : {&tg.nx, &tg.ny, &tg.nz}) if ((rc = ensure(c, *pl, (size_t)n * 4))) return rc;
    Planes6 pl;
    pl.s[0] = sc.x.as<float>(); pl.s[1] = sc.y.as<float>(); pl.s[2] = sc.z.as<float>();
    pl.d[0] = tg.x.as<float>(); pl.d[1] = tg.y.as<float>(); pl.d[2] = tg.z.as<float>();
    pl.s[3] = sc.has_normals ? sc.nx.as<float>() : nullptr; pl.s[4] = sc.has_normals ? sc.ny.as<float>() : nullptr; pl.s[5] = sc.has_normals ? sc.nz.as<float>() : nullptr;
    pl.d[3] = sc.has_normals ? tg.nx.as<float>() : nullptr; pl.d[4] = sc.has_normals ? tg.ny.as<float>() : nullptr; pl.d[5] = sc.has_normals ? tg.nz.as<float>() : nullptr;
    hipLaunchKernelGGL(k_copy_planes_pad, dim3((npad + 255) / 256, 6), dim3(256), 0, c->stream, pl, n, npad);
    HIPCK(c, hipGetLastError());
    tg.n = n; tg.npad = npad; tg.has_normals = sc.has_normals; tg.has_colors = false;
    Bvh& b = c->bvh;
    c->bvh6.valid = false;
    b.valid = false; b.n_valid = 0;
    if ((rc = finite_list(c, c->tgt, false, c->tgt_flag, c->tgt_finite, &b.n_valid))) return rc;
    b.d_finite = c->tgt_finite.as<int>(); b.n_ids = n;
    c->bvh6.d_finite = b.d_finite; c->bvh6.n_valid = b.n_valid; c->bvh6.n_ids = n;
    b.attrs = &c->tgt; c->bvh6.attrs = &c->tgt;
    if (c->prm.knn_backend == ICP_KNN_LBVH && c->prm.matching == ICP_MATCH_KNN) return guard.done(build_bvh<3>(c, b, target_coords3(c)));
    return guard.done();
}

int icp_set_source(icp_ctx* c, const float* xyz, const float* normals, const uint8_t* rgba, int32_t n) {
    if (!c || !xyz || n <= 0) { if (c) c->err = "icp_set_source: null points or n <= 0"; return ICP_ERR_INVALID_ARG; }
    int rc;
    DrainOnError guard(c);
    if ((rc = set_device(c))) return rc;
    if ((rc = upload_cloud(c, c->src, xyz, normals, rgba, n, false))) return rc;
    // validity of a source point for the multi-resolution selections (finite point && finite normal, PointCloud.h:334) and the
    // bounding box of the finite points (Morton order of the queries): both on the device, nothing waits for them here
    if ((rc = ensure(c, c->src_flag, (size_t)n))) return rc;
    if ((rc = ensure(c, c->src_box, 32))) return rc;
    hipLaunchKernelGGL(k_mark_finite, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->src.x.as<float>(), c->src.y.as<float>(), c->src.z.as<float>(),
                       normals ? c->src.nx.as<float>() : nullptr, normals ? c->src.ny.as<float>() : nullptr, normals ? c->src.nz.as<float>() : nullptr, n, c->src_flag.as<uint8_t>());
    HIPCK(c, hipMemsetAsync(c->src_box.p, 0xFF, 12, c->stream));
    HIPCK(c, hipMemsetAsync((char*)c->src_box.p + 12, 0x00, 12, c->stream));
    hipLaunchKernelGGL(k_bbox, dim3(256), dim3(256), 0, c->stream, c->src.x.as<float>(), c->src.y.as<float>(), c->src.z.as<float>(), n, c->src_box.as<unsigned int>());
    HIPCK(c, hipGetLastError());
    HIPCK(c, hipStreamSynchronize(c->stream));           // entry-point contract: the caller's arrays are free, the stream is idle
    for (auto& kv : c->levels) release(kv.second);
    c->levels.clear();
    return guard.done();
}

int icp_query_matches(icp_ctx* c, const float* transformed_xyz, const uint8_t* rgba, int32_t n, icp_match_t* out) {
    if (!c || !transformed_xyz || !out || n <= 0) { if (c) c->err = "icp_query_matches: bad argument"; return ICP_ERR_INVALID_ARG; }
    int rc;
    DrainOnError guard(c);
    if ((rc = set_device(c))) return rc;
    if ((rc = check_ready(c, false, false))) return rc;
    const bool colors = rgba != nullptr;
    if (c->prm.matching == ICP_MATCH_KNN && colors && !c->tgt.has_colors) {      // NearestNeighbor.h:240-243
        c->err = "index built without colours: call queryMatches without colours";
        return ICP_ERR_COLOR_MISMATCH;
    }
    if ((rc = upload_cloud(c, c->qry, transformed_xyz, nullptr, rgba, n, false))) return rc;
    QuerySet q{&c->qry, nullptr, n, 1, colors && c->prm.matching == ICP_MATCH_KNN, false, nullptr};
    if ((rc = launch_match(c, q))) return rc;
    HIPCK(c, hipMemcpyAsync(out, c->matches.p, (size_t)n * sizeof(icp_match_t), hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return guard.done();
}

int icp_match(icp_ctx* c, const float pose[16], icp_match_t* out, float* d2_out) {
    if (!c || !pose || !out) { if (c) c->err = "icp_match_t: bad argument"; return ICP_ERR_INVALID_ARG; }
    int rc;
    DrainOnError guard(c);
    if ((rc = set_device(c))) return rc;
    if ((rc = check_ready(c, true, false))) return rc;
    if ((rc = write_pose(c, pose))) return rc;
    const int* full_order = nullptr;
    if ((rc = get_full_order(c, &full_order))) return rc;
    QuerySet q{&c->src, nullptr, c->src.n, 0, c->prm.color_icp != 0 && c->prm.matching == ICP_MATCH_KNN, false, full_order};
    if ((rc = launch_match(c, q))) return rc;
    HIPCK(c, hipMemcpyAsync(out, c->matches.p, (size_t)q.n * sizeof(icp_match_t), hipMemcpyDeviceToHost, c->stream));
    if (d2_out) HIPCK(c, hipMemcpyAsync(d2_out, c->d2.p, (size_t)q.n * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return guard.done();
}

int icp_correspond(icp_ctx* c, const float pose[16], icp_match_t* out, double* sums_out, int32_t* n_valid_out) {
    if (!c || !pose) { if (c) c->err = "icp_correspond: bad argument"; return ICP_ERR_INVALID_ARG; }
    int rc;
    DrainOnError guard(c);
    if ((rc = set_device(c))) return rc;
    if ((rc = check_ready(c, true, true))) return rc;
    if ((rc = write_pose(c, pose))) return rc;
    const int* full_order = nullptr;
    if ((rc = get_full_order(c, &full_order))) return rc;
    QuerySet q{&c->src, nullptr, c->src.n, 0, c->prm.color_icp != 0 && c->prm.matching == ICP_MATCH_KNN, false, full_order};
    if ((rc = launch_match(c, q))) return rc;
    if ((rc = ensure(c, c->sums, NSUM * 8))) return rc;
    if ((rc = rearm_handover(c))) return rc;
    if ((rc = launch_post_and_solve(c, c->src, nullptr, q.n, nullptr, c->sums.as<double>(), 0, nullptr))) return rc;
    double hs[NSUM]; int fault = 0;
    if (out) HIPCK(c, hipMemcpyAsync(out, c->matches.p, (size_t)q.n * sizeof(icp_match_t), hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipMemcpyAsync(hs, c->sums.p, NSUM * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipMemcpyAsync(&fault, &c->ps.as<PoseState>()->fault, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    if (fault) { c->err = "reduction hand-over timed out on the device (k_reduce_solve)"; return ICP_ERR_HIP; }
    if (sums_out) { memset(sums_out, 0, 64 * 8); memcpy(sums_out, hs, NSUM * 8); }
    if (n_valid_out) *n_valid_out = (int32_t)hs[SUM_N];
    return guard.done();
}

// The fused matcher driven launch by launch with caller-dictated poses: launch 0 unseeded, launch j > 0 seeded + incremental exactly
// as iteration j of run_loop (same kernel, same buffers, same launch parameters); the last launch's records come back in source order.
int icp_match_seeded(icp_ctx* c, const float* poses, int32_t n_poses, icp_match_t* out, float* d2_out) {
    if (!c || !poses || n_poses <= 0) { if (c) c->err = "icp_match_seeded: bad argument"; return ICP_ERR_INVALID_ARG; }
    const icp_params& p = c->prm;
    if (p.matching != ICP_MATCH_KNN || p.knn_backend != ICP_KNN_LBVH || p.metric == ICP_METRIC_SYMMETRIC || !c->fuse_post) {
        c->err = "icp_match_seeded: needs k-NN matching on the LBVH backend with the fused point-to-point / point-to-plane matcher"; return ICP_ERR_INVALID_ARG;
    }
    int rc;
    DrainOnError guard(c);
    if ((rc = set_device(c))) return rc;
    if ((rc = check_ready(c, true, true))) return rc;
    const Cloud* cloud = nullptr; int n = 0;
    if ((rc = get_sorted_level(c, 0, &cloud, &n))) return rc;
    struct Keep { icp_ctx* c; ~Keep() { c->keep_fused_records = false; } } keep{c};
    if (c->persist_loop && c->merge_loop && p.metric == ICP_METRIC_POINT_TO_PLANE) {
        // what icp_run launches for this configuration: k_icp_loop, all the launches' worth of iterations in ONE launch -- here with every pose
        // slot filled in up front (replica 0 of each; nobody reduces, nobody solves), the last iteration writing its records
        const int nb = fused_nblocks(n);
        const size_t slot_bytes = (size_t)(n_poses + 1) * POSE_REPLICAS * POSE_REPLICA_STRIDE;
        if ((rc = ensure(c, c->ring, slot_bytes + 64))) return rc;
        if ((rc = ensure(c, c->pring, (size_t)PRING_DEPTH * NSUM_USED * nb * 8))) return rc;
        PoseState* slots = c->ring.as<PoseState>(); int* fault = (int*)(c->ring.as<char>() + slot_bytes);
        HIPCK(c, hipMemsetAsync(fault, 0, 64, c->stream));
        std::vector<PoseState> hp((size_t)n_poses);
        for (int j = 0; j < n_poses; j++) {
            memset(&hp[(size_t)j], 0, sizeof(PoseState)); memcpy(hp[(size_t)j].pose, poses + (size_t)16 * j, 64); normal_matrix_from_pose(hp[(size_t)j].pose, hp[(size_t)j].nmat);
            HIPCK(c, hipMemcpyAsync(loop_slot(slots, j, 0), &hp[(size_t)j], sizeof(PoseState), hipMemcpyHostToDevice, c->stream));
        }
        LoopParams L; memset(&L, 0, sizeof(L));
        L.iters = n_poses; L.first = 0; L.seed_first = 0; L.slots = slots; L.totals = nullptr; L.pring = c->pring.as<unsigned long long>(); L.nb = nb;
        L.dictated = 1; L.stats = nullptr; L.n_src = n; L.abort_word = fault; L.record_last = 1; L.clocks = nullptr;
        c->keep_fused_records = true;
        QuerySet q{cloud, nullptr, n, 0, p.color_icp != 0, false, nullptr};
        MergeLaunch ml; ml.loop = &L; ml.slot = nullptr; ml.partials = nullptr; memset(&ml.rp, 0, sizeof(ml.rp));
        int fused = 0;
        if ((rc = launch_match(c, q, &fused, &ml))) return rc;
        if (!fused) { c->err = "icp_match_seeded: the matcher did not take the fused path"; return ICP_ERR_INVALID_ARG; }
        int hf = 0;
        HIPCK(c, hipMemcpyAsync(&hf, fault, 4, hipMemcpyDeviceToHost, c->stream));
        HIPCK(c, hipStreamSynchronize(c->stream));           // (hp is read by the copies above)
        if (hf) { c->err = "icp_match_seeded: k_icp_loop gave up waiting"; return ICP_ERR_HIP; }
    } else
    for (int j = 0; j < n_poses; j++) {
        if ((rc = write_pose(c, poses + (size_t)16 * j))) return rc;
        c->keep_fused_records = (j == n_poses - 1);
        QuerySet q{cloud, nullptr, n, 0, p.color_icp != 0, j > 0, nullptr};
        int fused = 0;
        if ((rc = launch_match(c, q, &fused))) return rc;
        if (!fused) { c->err = "icp_match_seeded: the matcher did not take the fused path"; return ICP_ERR_INVALID_ARG; }
        HIPCK(c, hipStreamSynchronize(c->stream));           // the pose staging area is reused by the next launch
    }
    std::vector<int> pos((size_t)n); std::vector<icp_match_t> m((size_t)n); std::vector<float> d((size_t)n);
    HIPCK(c, hipMemcpyAsync(pos.data(), c->levels[0].sorted_idx.p, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipMemcpyAsync(m.data(), c->matches.p, (size_t)n * sizeof(icp_match_t), hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipMemcpyAsync(d.data(), c->d2.p, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    for (int t = 0; t < n; t++) {                            // sorted position -> source index
        if (out) out[pos[(size_t)t]] = m[(size_t)t];
        if (d2_out) d2_out[pos[(size_t)t]] = d[(size_t)t];
    }
    return guard.done();
}

// Iteration schedule of LinearICPOptimizer::estimatePose: ICPOptimizer.h:503-516 (coarsest level),
// :540 (loop condition `i < nIter || multires`) and :634-655 (refinement).  Pure host logic.
int icp_schedule(const icp_params* p, int32_t n_src, int32_t* factors_out, int32_t max_out, int32_t* count_out) {
    if (!p || !count_out || n_src < 0) return ICP_ERR_INVALID_ARG;
    int cnt = 0;
    if (!p->multires) {
        for (int i = 0; i < p->n_iterations; i++) { if (factors_out && cnt < max_out) factors_out[cnt] = 0; cnt++; }
    } else {
        if (p->n_iterations < 1) return ICP_ERR_INVALID_ARG;     // `i >= m_nIterations - 1` is unsigned in the reference: never true
        float res = 1.0f; int osz = n_src;
        while (1) { osz = (int)(osz / 2.0); if (osz < 100) break; res *= 2.0f; }      // MULTI_RESOLUTION_MINIMUM_POINTS :21
        for (int i = 0;; ++i) {
            if (factors_out && cnt < max_out) factors_out[cnt] = (int)res;
            cnt++;
            if (res == 1.0f && i >= p->n_iterations - 1) break;
            if (res == 1.0f) continue;
            res /= 2.0f; if (res < 1.0f) res = 1.0f;
        }
    }
    *count_out = cnt;
    return ICP_OK;
}

static int enqueue_fontana(icp_ctx* c, float* d_out);

static int run_loop(icp_ctx* c, float pose_inout[16], icp_iter_stats* stats, int32_t max_stats, int32_t* n_run, bool single) {
    const icp_params& p = c->prm;
    int rc;
    DrainOnError guard(c);
    if ((rc = set_device(c))) return rc;
    if ((rc = check_ready(c, true, true))) return rc;
    std::vector<int> factors;          // decimation factor per iteration; 0 = no selection (full cloud)
    if (single) factors.push_back(0);
    else {
        int32_t cnt = 0;
        if ((rc = icp_schedule(&p, c->src.n, nullptr, 0, &cnt))) { c->err = "multires with n_iterations < 1 never terminates in the reference"; return rc; }
        factors.resize((size_t)cnt);
        if (cnt > 0) icp_schedule(&p, c->src.n, factors.data(), cnt, &cnt);
    }
    const int iters = (int)factors.size();
    if (n_run) *n_run = 0;
    if (iters == 0) return guard.done();
    // page-locked staging for the whole run up front: [pose state up | per-iteration records down | pose state down]
    const size_t pin_stats = 256, pin_pose = pin_stats + (((size_t)iters * sizeof(icp_iter_stats) + 255) & ~(size_t)255);
    if ((rc = ensure_pinned(c, pin_pose + 512 + (size_t)(iters + 1) * 8))) return rc;
    float pose_in[16]; memcpy(pose_in, pose_inout, 64);        // the record of an empty iteration 0 carries the incoming pose
    if ((rc = write_pose(c, pose_inout))) return rc;
    // the records of the run, and behind them (merged / one-launch loops) the final pose state, the fault word and the device clocks:
    // ONE block, ONE copy back -- laid out like the page-locked block it lands in (pin_stats .. pin_pose .. + 128 .. + 192)
    const size_t stats_pad = pin_pose - pin_stats;
    if ((rc = ensure(c, c->stats, stats_pad + 192 + (size_t)(iters + 1) * 8))) return rc;
    // (every record of an iteration with work is written in full by k_reduce_solve; empty iterations are filled in on the host)
    if ((rc = ensure_events(c, (size_t)iters * 4 + 2))) return rc;
    // resolve selections up front (uploads) so the loop itself is launch-only
    std::vector<const int*> sels(iters, nullptr); std::vector<int> ns(iters, c->src.n);
    std::vector<const int*> orders(iters, nullptr);
    std::vector<const Cloud*> clouds(iters, &c->src);
    // BVH matcher without resampling: every level is a physical, Morton-sorted copy -> no index lists in the loop at all
    const bool sorted_levels = p.matching == ICP_MATCH_KNN && p.knn_backend == ICP_KNN_LBVH && !(!single && p.selection == 1);
    for (int i = 0; i < iters; i++) {
        if (sorted_levels) { if ((rc = get_sorted_level(c, factors[i], &clouds[i], &ns[i]))) return rc; }
        else if (factors[i] > 0) { if ((rc = get_level(c, factors[i], &sels[i], &ns[i], nullptr))) return rc; }
    }
    if (!single && p.selection == 1) {
        // RANDOM_SAMPLING (ICPOptimizer.h:549-550: resample at the start of every iteration, over the current level's cloud).
        // All resamples are drawn up front on the device; one small copy returns their sizes so the loop stays launch-only.
        double th = (double)p.selection_proba * 4294967296.0;
        const int take_all = th >= 4294967296.0 ? 1 : 0;
        const uint32_t threshold = th <= 0.0 ? 0u : (take_all ? 0xFFFFFFFFu : (uint32_t)th);
        const size_t cap = (size_t)c->src.n;
        if ((rc = ensure(c, c->sel_lists, (size_t)iters * cap * 4))) return rc;
        if ((rc = ensure(c, c->sel_counts, (size_t)iters * 4))) return rc;
        if ((rc = ensure(c, c->sel_blocks, (size_t)((cap + 255) / 256 + 1) * 4))) return rc;
        for (int i = 0; i < iters; i++) {
            const int nb = (ns[i] + 255) / 256;
            int* out = c->sel_lists.as<int>() + (size_t)i * cap;
            if (ns[i] > 0) {
                hipLaunchKernelGGL(k_select_count, dim3(nb), dim3(256), 0, c->stream, sels[i], ns[i], p.selection_seed, (uint32_t)i, threshold, take_all, c->sel_blocks.as<int>());
                hipLaunchKernelGGL(k_select_scan, dim3(1), dim3(1024), 0, c->stream, c->sel_blocks.as<int>(), nb, c->sel_counts.as<int>() + i);
                hipLaunchKernelGGL(k_select_scatter, dim3(nb), dim3(256), 0, c->stream, sels[i], ns[i], p.selection_seed, (uint32_t)i, threshold, take_all, c->sel_blocks.as<int>(), out);
            } else HIPCK(c, hipMemsetAsync(c->sel_counts.as<int>() + i, 0, 4, c->stream));
            sels[i] = out; orders[i] = nullptr;
        }
        HIPCK(c, hipGetLastError());
        std::vector<int> counts((size_t)iters);
        HIPCK(c, hipMemcpyAsync(counts.data(), c->sel_counts.p, (size_t)iters * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCK(c, hipStreamSynchronize(c->stream));
        for (int i = 0; i < iters; i++) ns[i] = counts[i];
    }
    const bool rmse = (p.record_rmse & 1) && c->conv_n > 0;
    const bool fontana = (p.record_rmse & 2) && c->conv_n > 0;
    if (rmse) { if ((rc = ensure(c, c->rmse_partials, 256 * 2 * 8))) return rc; }
    // The merged loop (dev_solve.hpp, "the ring form"): point-to-plane through the fused BVH matcher on sorted levels, nothing else on
    // the stream between two iterations.  Launch i = [reducer of iteration i - 1 | matcher of iteration i]; one reducer-only launch closes
    // the run.  Pose slots and totals rows are written once per run; both rings are reset here, so nothing survives an aborted run.
    bool merged = c->merge_loop && !single && iters >= 2 && sorted_levels && c->fuse_post && p.metric == ICP_METRIC_POINT_TO_PLANE && !rmse && !fontana;
    for (int i = 0; merged && i < iters; i++) if (ns[i] <= 0) merged = false;
    PoseState* slots = nullptr; unsigned long long* trows = nullptr; int* run_fault = nullptr;
    // k_icp_loop (dev_persist.hpp): all iterations of a resolution level in ONE launch, the waves resident from iteration to iteration.
    // Needs the whole grid on the device at once (its blocks wait for each other) and the device to itself: checked against the kernel's
    // occupancy; a context marked shared, or one that finds another context's loop in flight, runs one launch per iteration instead.
    struct Seg { int i0, i1; };
    std::vector<Seg> segs;
    LoopToken token;
    bool persist = merged && c->persist_loop && !c->shared_gpu;
    long long* d_clocks = nullptr;
    // The first loop_from iterations -- every query walks, a launch lasts 50-150 us, and the walk is ~5 % slower in k_icp_loop (it pays for
    // its residency with a few spilled registers) -- run one launch per iteration in the merged form; k_icp_loop takes over from there.
    int loop_from = c->loop_from < 0 ? 0 : c->loop_from;
    if (persist && loop_from >= iters) persist = false;
    if (persist) {
        Bvh& tb = p.color_icp ? c->bvh6 : c->bvh;
        int cap = -1;
        if (!tb.valid) persist = false;                    // (built on first use: the first run of such a context takes the per-launch loop)
        else if (p.color_icp) { if ((rc = tb.Lq <= 8 ? loop_capacity_of<6, false>(c, &cap) : loop_capacity_of<6, true>(c, &cap))) return rc; }
        else { if ((rc = tb.Lq <= 8 ? loop_capacity_of<3, false>(c, &cap) : loop_capacity_of<3, true>(c, &cap))) return rc; }
        int nbmax = 1;
        for (int i = 0; i < iters; i++) { const int nb = fused_nblocks(ns[i]); if (nb > nbmax) nbmax = nb; }
        for (int i = loop_from; persist && i < iters; ) {
            int j = i + 1;
            while (j < iters && clouds[j] == clouds[i] && ns[j] == ns[i] && factors[j] == factors[i]) j++;
            const int nb = fused_nblocks(ns[i]);
            if (nb + LOOP_RED + 1 > cap) persist = false;      // (the reducer's two-wave blocks sit in the holes the matcher grid leaves: dev_persist.hpp)
            if (nb > nbmax) nbmax = nb;
            segs.push_back(Seg{i, j});
            i = j;
        }
        if (persist && !token.try_take()) persist = false;
        if (persist) {
            // every buffer the launches below touch is sized for the largest level NOW: an allocation that grows between two launches frees its
            // old block, and hipFree waits for the device -- for a reducer kernel that is itself waiting for the matcher launch still to come
            { int nmax = 1; for (int i = 0; i < iters; i++) if (ns[i] > nmax) nmax = ns[i];
              if ((rc = ensure_match_buffers(c, nmax))) return rc;
              if ((rc = ensure_qpack(c, nmax))) return rc;
              if ((rc = ensure(c, c->partials, (size_t)(nbmax > POST_BLOCKS ? nbmax : POST_BLOCKS) * NSUM * 8))) return rc;
              if ((rc = ensure(c, c->partials2, (size_t)(nbmax > POST_BLOCKS ? nbmax : POST_BLOCKS) * NSUM * 8))) return rc; }
            size_t pring_granules = 0;
            for (const Seg& sg : segs) pring_granules += (size_t)PRING_DEPTH * NSUM_USED * (fused_nblocks(ns[sg.i0]));
            if ((rc = ensure(c, c->pring, pring_granules * 8))) return rc;
            HIPCK(c, hipMemsetAsync(c->pring.p, 0xFF, pring_granules * 8, c->stream));
            if (!c->stream2) {
                HIPCK(c, hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
                HIPCK(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
                HIPCK(c, hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
            }
            // its rings: [pose slots x POSE_REPLICAS | totals]; fault word and clocks sit behind the records (see above)
            const size_t slot_bytes = (size_t)(iters + 1) * POSE_REPLICAS * POSE_REPLICA_STRIDE, tot_bytes = (size_t)iters * TOTALS_ROW * 8;
            if ((rc = ensure(c, c->ring, slot_bytes + tot_bytes))) return rc;
            slots = c->ring.as<PoseState>(); trows = (unsigned long long*)(c->ring.as<char>() + slot_bytes);
            run_fault = (int*)(c->stats.as<char>() + stats_pad + 128); d_clocks = (long long*)(c->stats.as<char>() + stats_pad + 192);
            const int n_init = (iters + 1) * POSE_REPLICAS * 16 + iters * TOTALS_ROW + 16 + (iters + 1) * 2;
            hipLaunchKernelGGL(k_run_init, dim3((n_init + 255) / 256), dim3(256), 0, c->stream, c->ps.as<PoseState>(), slots, iters + 1, trows, iters * TOTALS_ROW, run_fault, 16 + (iters + 1) * 2);
        }
    }
    if (merged && !persist) {
        static_assert(sizeof(PoseState) == 128, "a pose slot is 16 granules");
        const size_t slot_bytes = (size_t)(iters + 1) * POSE_REPLICAS * POSE_REPLICA_STRIDE, tot_bytes = (size_t)iters * NSUM * 8;
        int nbmax = POST_BLOCKS;
        for (int i = 0; i < iters; i++) { const int nb = fused_nblocks(ns[i]); if (nb > nbmax) nbmax = nb; }
        if ((rc = ensure(c, c->ring, slot_bytes + tot_bytes))) return rc;
        if ((rc = ensure(c, c->partials, (size_t)nbmax * NSUM * 8))) return rc;
        if ((rc = ensure(c, c->partials2, (size_t)nbmax * NSUM * 8))) return rc;
        if (!p.color_icp && (rc = ensure_gx(c, nbmax))) return rc;
        slots = c->ring.as<PoseState>(); trows = (unsigned long long*)(c->ring.as<char>() + slot_bytes); run_fault = (int*)(c->stats.as<char>() + stats_pad + 128);
        const int n_init = (iters + 1) * POSE_REPLICAS * 16 + iters * NSUM + 16;
        hipLaunchKernelGGL(k_run_init, dim3((n_init + 255) / 256), dim3(256), 0, c->stream, c->ps.as<PoseState>(), slots, iters + 1, trows, iters * NSUM, run_fault, 16);      // (both rings are reset: nothing survives an aborted run)
    }
    // Stage timing (TimeMeasure.h:20-26).  A HIP event costs ~4 us of stream time, two to three per iteration are ~10 % of a
    // 0.07 ms iteration: mode N > 1 brackets only every Nth iteration (offset rotating from run to run) and scales the sums.
    // (k_icp_loop: no events inside a launch -- its reducer block 0 leaves the 100 MHz clock at every pose it publishes instead:
    //  "match" of iteration i = the time from pose i to pose i + 1, every iteration, at no cost.)
    // Event slots: 4 per iteration (start, after match, after post, end) + run start / run end.  In the merged loop an iteration is ONE
    // launch (its reduce + solve happen inside the next one): "match" is that launch, "solve" only the closing reducer-only launch.
    const int tmode = c->stage_timing;
    std::vector<char> sampled((size_t)iters, 0), post_event((size_t)iters, 0);
    for (int i = 0; i < iters; i++) sampled[i] = tmode == 1 || (tmode > 1 && (i + (int)(c->timing_phase % (unsigned)tmode)) % tmode == 0);
    c->timing_phase++;
    auto E = [&](int i, int k) { return c->events[(size_t)2 + 4 * i + k]; };
    auto end_event = [&](int i) { return (merged && i < (persist ? loop_from : iters) - 1) ? E(i, 1) : E(i, 3); };
    auto start_event = [&](int i) { return (i > 0 && sampled[i - 1] && !(merged && c->ext_events)) ? end_event(i - 1) : E(i, 0); };
    auto ring_params = [&](int i) {                      // the reducer of iteration i - 1, riding in launch i (i = iters: the closing launch)
        RingParams rp; memset(&rp, 0, sizeof(rp));
        rp.run_fault = run_fault;
        if (i > 0) {
            rp.n_red = NSUM_USED;
            rp.red_partials = ((i - 1) & 1) ? c->partials2.as<double>() : c->partials.as<double>(); rp.red_nblocks = fused_nblocks(ns[i - 1]);
            rp.totals_row = trows + (size_t)(i - 1) * NSUM; rp.ps_in = loop_slot(slots, i - 1, 0); rp.ps_out = loop_slot(slots, i, 0);
            rp.stats = c->stats.as<icp_iter_stats>() + (i - 1); rp.n_src = ns[i - 1];
            if (i == iters) rp.final_out = (PoseState*)(c->stats.as<char>() + stats_pad);
        }
        return rp;
    };
    if (!merged && (rc = rearm_handover(c))) return rc;
    HIPCK(c, hipEventRecord(c->events[0], c->stream));
    const int m_end = persist ? loop_from : iters;       // iterations [0, m_end) run one launch each
    for (int i = 0; i < m_end; i++) {
        icp_iter_stats* d_st = c->stats.as<icp_iter_stats>() + i;
        const bool ev = sampled[i] != 0;
        const bool ext_ev = ev && merged && ns[i] > 0 && c->ext_events;      // merged form: the launch's own start / stop times, no bracket on the stream
        if (ev && !ext_ev && !(i > 0 && sampled[i - 1] && !c->ext_events)) HIPCK(c, hipEventRecord(E(i, 0), c->stream));
        if (ns[i] > 0) {
            // seed the search with the previous iteration's neighbours when it matched the same queries (same level)
            const bool seed = i > 0 && factors[i] == factors[i - 1] && ns[i - 1] > 0 && p.selection == 0;
            QuerySet q{clouds[i], sels[i], ns[i], 0, p.color_icp != 0 && p.matching == ICP_MATCH_KNN, seed, orders[i]};
            int fused = 0;
            MergeLaunch ml;
            if (merged) { ml.rp = ring_params(i); ml.slot = loop_slot(slots, i, 0); ml.partials = (i & 1) ? c->partials2.as<double>() : c->partials.as<double>(); }
            if (ext_ev) { ml.ev_start = E(i, 0); ml.ev_stop = E(i, 1); }
            if ((rc = launch_match(c, q, c->fuse_post ? &fused : nullptr, merged ? &ml : nullptr))) return rc;
            if (merged && !fused) { c->err = "merged loop: the matcher did not take the fused path"; return ICP_ERR_HIP; }
            if (ev && !ext_ev) HIPCK(c, hipEventRecord(E(i, 1), c->stream));
            // fused epilogue: there is no separate post stage to bracket
            if (!merged) {
                if ((rc = launch_post_and_solve(c, *clouds[i], sels[i], ns[i], d_st, nullptr, 1, (ev && !fused) ? E(i, 2) : nullptr, fused))) return rc;
                post_event[i] = ev && !fused;
            }
        } else if (ev) {
            HIPCK(c, hipEventRecord(E(i, 1), c->stream));
        }
        if (rmse) {
            hipLaunchKernelGGL(k_rmse_partial, dim3(256), dim3(256), 0, c->stream, c->conv_src.x.as<float>(), c->conv_src.y.as<float>(), c->conv_src.z.as<float>(),
                               c->conv_ref.x.as<float>(), c->conv_ref.y.as<float>(), c->conv_ref.z.as<float>(), c->conv_n, c->ps.as<PoseState>(), c->rmse_partials.as<double>());
            hipLaunchKernelGGL(k_rmse_finish, dim3(1), dim3(64), 0, c->stream, c->rmse_partials.as<double>(), 256, &d_st->rmse);
        }
        if (fontana && (rc = enqueue_fontana(c, &d_st->benchmark_error))) return rc;
        if (merged && i == m_end - 1) {                  // the closing launch: reducer of the last of these iterations, nothing behind it to ride in
            hipLaunchKernelGGL(k_ring_reduce_solve, dim3(NSUM_USED), dim3(RING_THREADS), 0, c->stream, ring_params(m_end));
            HIPCK(c, hipGetLastError());
        }
        if (ev && !(merged && i < m_end - 1)) HIPCK(c, hipEventRecord(E(i, 3), c->stream));
    }
    if (persist) {
        // every level: the reducer on the second stream (forked off here, joined below), the matcher grid on the context's; each level has
        // its own section of the partial ring, so that a level's matcher never writes where the level before is still being re-armed
        HIPCK(c, hipEventRecord(c->ev_fork, c->stream));
        HIPCK(c, hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
        size_t pring_off = 0;
        for (const Seg& sg : segs) {
            const int i = sg.i0, nb = fused_nblocks(ns[i]);
            LoopParams L; memset(&L, 0, sizeof(L));
            L.iters = sg.i1 - sg.i0; L.first = i; L.seed_first = (i > 0 && i == loop_from && factors[i] == factors[i - 1] && ns[i - 1] == ns[i] && clouds[i] == clouds[i - 1]) ? 1 : 0; L.slots = slots; L.totals = trows; L.pring = c->pring.as<unsigned long long>() + pring_off; L.nb = nb;
            { const char* e = getenv("ICP_HIP_LOOP_PRESLEEP"); L.presleep_eighths = e ? atoi(e) : 5; }
            { const char* e = getenv("ICP_HIP_LOOP_WAVESLEEP"); L.wave_presleep_eighths = e ? atoi(e) : 0; }
            L.dictated = 0; L.stats = c->stats.as<icp_iter_stats>(); L.n_src = ns[i]; L.abort_word = run_fault; L.record_last = 0; L.clocks = d_clocks;
            L.final_out = (PoseState*)(c->stats.as<char>() + stats_pad); L.final_g = iters;
#if ICP_DEBUG_TIMES
            if ((rc = ensure(c, c->dbg_steps, (size_t)(ns[i] > 65536 ? ns[i] : 65536) * 4))) return rc;
            L.dbg = c->dbg_steps.as<int>(); L.dbg_waves = nb * (BVH_THREADS / WAVE); { const char* e = getenv("ICP_HIP_DBG_ITER"); L.dbg_iter = e ? atoi(e) : -1; }
#endif
            pring_off += (size_t)PRING_DEPTH * NSUM_USED * nb;
            QuerySet q{clouds[i], sels[i], ns[i], 0, p.color_icp != 0, false, orders[i]};
            MergeLaunch ml; ml.loop = &L; ml.slot = nullptr; ml.partials = nullptr; memset(&ml.rp, 0, sizeof(ml.rp));
            int fused = 0;
            if ((rc = launch_match(c, q, &fused, &ml))) return rc;      // (first: nothing on the host may block between the two launches of a level)
            if (!fused) { c->err = "k_icp_loop: the matcher did not take the fused path"; return ICP_ERR_HIP; }
            hipLaunchKernelGGL(k_icp_loop_reducer, dim3(LOOP_RED + 1), dim3(RING_THREADS), 0, c->stream2, L);      // 2 x 34 fold blocks + the solver
            HIPCK(c, hipGetLastError());
        }
        HIPCK(c, hipEventRecord(c->ev_join, c->stream2));
        HIPCK(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));
    }
    HIPCK(c, hipEventRecord(c->events[1], c->stream));
    std::vector<icp_iter_stats> hs((size_t)iters);
    if (merged) HIPCK(c, hipMemcpyAsync((char*)c->pinned + pin_stats, c->stats.p, stats_pad + 192 + (persist ? (size_t)(iters + 1) * 8 : 0), hipMemcpyDeviceToHost, c->stream));      // records | final pose state | fault | clocks
    else {
        HIPCK(c, hipMemcpyAsync((char*)c->pinned + pin_stats, c->stats.p, (size_t)iters * sizeof(icp_iter_stats), hipMemcpyDeviceToHost, c->stream));
        HIPCK(c, hipMemcpyAsync((char*)c->pinned + pin_pose, c->ps.p, sizeof(PoseState), hipMemcpyDeviceToHost, c->stream));
    }
    HIPCK(c, hipStreamSynchronize(c->stream));
    if (merged) {
        if (persist) c->loop_runs++; else c->merged_runs++;
        const PoseState* hp = (const PoseState*)((char*)c->pinned + pin_pose);
        const int rf = *(const int*)((char*)c->pinned + pin_pose + 128);
        if (hp->fault || rf) {
            // a pivot of the 6 x 6 system failed the rank test (the eigen fallback lives in k_reduce_solve only), or a bounded wait ran out:
            // the same run again, from the incoming pose, with the separate launches
            c->merged_fallbacks++; c->gx_dirty = true;     // (a run cut short may have left claimed slots in the outboxes)
            if (c->trace) fprintf(stderr, "[icp_hip] %s gave up: slot fault %d, abort word %d -> the run again with separate launches\n", persist ? "k_icp_loop" : "merged loop", hp->fault, rf);
            guard.ok = true;                                 // synchronised
            memcpy(pose_inout, pose_in, 64);
            const bool m0 = c->merge_loop;
            c->merge_loop = false;
            token.~LoopToken(); token.held = false;
            const int rc2 = run_loop(c, pose_inout, stats, max_stats, n_run, single);
            c->merge_loop = m0;
            return rc2;
        }
    }
    memcpy(hs.data(), (char*)c->pinned + pin_stats, (size_t)iters * sizeof(icp_iter_stats));
    memcpy(pose_inout, ((const PoseState*)((char*)c->pinned + pin_pose))->pose, 64);
    if (((const PoseState*)((char*)c->pinned + pin_pose))->fault) { c->gx_dirty = true; c->err = "reduction hand-over timed out on the device (k_reduce_solve)"; return ICP_ERR_HIP; }
    int status = ICP_OK;
    for (int i = 0; i < iters; i++) {
        if (ns[i] <= 0) { hs[i].n_src = 0; hs[i].n_valid = 0; hs[i].status = ICP_ERR_NO_CORRESPONDENCES; memcpy(hs[i].pose, i ? hs[i - 1].pose : pose_in, 64); hs[i].rmse = -1.f; hs[i].benchmark_error = -1.f; }
        if (!rmse) hs[i].rmse = -1.f;
        if (!fontana) hs[i].benchmark_error = -1.f;
        if (hs[i].status != ICP_OK && status == ICP_OK) status = hs[i].status;
        if (stats && i < max_stats) stats[i] = hs[i];
    }
    if (n_run) *n_run = iters;
    icp_timing& t = c->timing; memset(&t, 0, sizeof(t)); t.iterations = iters;
    int n_sampled = 0;
    c->it_match_ms.assign((size_t)iters, -1.f); c->it_post_ms.assign((size_t)iters, -1.f); c->it_solve_ms.assign((size_t)iters, -1.f);
    double ev_match = 0, ev_post = 0, ev_solve = 0; int n_ev = 0;
    for (int i = 0; i < m_end; i++) {
        if (!sampled[i]) continue;
        n_ev++;
        float a = 0, b = 0, d = 0;
        HIPCK(c, hipEventElapsedTime(&a, start_event(i), E(i, 1)));
        if (post_event[i]) HIPCK(c, hipEventElapsedTime(&b, E(i, 1), E(i, 2)));
        if (!(merged && i < m_end - 1)) HIPCK(c, hipEventElapsedTime(&d, post_event[i] ? E(i, 2) : E(i, 1), E(i, 3)));
        ev_match += a; ev_post += b; ev_solve += d;
        c->it_match_ms[(size_t)i] = a; c->it_post_ms[(size_t)i] = b; c->it_solve_ms[(size_t)i] = d;
        if (c->trace) fprintf(stderr, "[icp_hip] it %2d  n %d  match %.4f  post %.4f  solve %.4f ms\n", i, ns[i], a, b, d);
    }
    if (n_ev > 0) {                                       // sampled: scale to all the iterations that ran one launch each
        const double f = (double)m_end / n_ev;
        t.match_ms += ev_match * f; t.weight_reject_build_ms += ev_post * f; t.solve_ms += ev_solve * f;
    }
    n_sampled = n_ev;
    if (persist) {                                        // k_icp_loop: the device's own clock at every published pose, 100 MHz ticks, every iteration
        const long long* clk = (const long long*)((char*)c->pinned + pin_pose + 192);
        for (int i = loop_from; i < iters; i++) {
            const float a = (float)((double)(clk[i + 1] - clk[i]) * 1e-5);
            t.match_ms += a; c->it_match_ms[(size_t)i] = a; c->it_post_ms[(size_t)i] = 0.f; c->it_solve_ms[(size_t)i] = 0.f;
            if (c->trace) fprintf(stderr, "[icp_hip] it %2d  n %d  iteration %.4f ms (k_icp_loop)\n", i, ns[i], a);
        }
        n_sampled += iters - loop_from;
        if (n_ev == 0 && m_end > 0) { const double f = (double)iters / (iters - loop_from); t.match_ms *= f; }      // nothing timed in front: the loop's iterations stand for all
    }
    t.sampled_iterations = n_sampled;
    float tot = 0; HIPCK(c, hipEventElapsedTime(&tot, c->events[0], c->events[1])); t.total_ms = tot;
    if (status != ICP_OK) c->err = "no valid correspondences in at least one iteration (reference would hang in ASSERT)";
    guard.ok = true;                                     // synchronised above; `status` reports empty iterations, not a HIP failure
    return status;
}

int icp_iterate(icp_ctx* c, float pose_inout[16], icp_iter_stats* stats) {
    if (!c || !pose_inout) { if (c) c->err = "icp_iterate: bad argument"; return ICP_ERR_INVALID_ARG; }
    int32_t n = 0;
    return run_loop(c, pose_inout, stats, stats ? 1 : 0, &n, true);
}

int icp_run(icp_ctx* c, float pose_inout[16], icp_iter_stats* stats, int32_t max_stats, int32_t* n_iterations_run) {
    if (!c || !pose_inout) { if (c) c->err = "icp_run: bad argument"; return ICP_ERR_INVALID_ARG; }
    return run_loop(c, pose_inout, stats, stats ? max_stats : 0, n_iterations_run, false);
}

int icp_set_stage_timing(icp_ctx* c, int32_t every_nth) {
    if (!c || every_nth < 0) { if (c) c->err = "icp_set_stage_timing: bad argument"; return ICP_ERR_INVALID_ARG; }
    c->stage_timing = every_nth;
    return ICP_OK;
}

int icp_get_timing(const icp_ctx* c, icp_timing* out) { if (!c || !out) return ICP_ERR_INVALID_ARG; *out = c->timing; return ICP_OK; }

int icp_get_iteration_times(const icp_ctx* c, float* match_ms, float* weight_reject_build_ms, float* solve_ms, int32_t max_out, int32_t* count_out) {
    if (!c || !count_out || max_out < 0) return ICP_ERR_INVALID_ARG;
    const int n = (int)c->it_match_ms.size();
    for (int i = 0; i < n && i < max_out; i++) {
        if (match_ms) match_ms[i] = c->it_match_ms[(size_t)i];
        if (weight_reject_build_ms) weight_reject_build_ms[i] = c->it_post_ms[(size_t)i];
        if (solve_ms) solve_ms[i] = c->it_solve_ms[(size_t)i];
    }
    *count_out = n;
    return ICP_OK;
}

int icp_set_convergence_reference(icp_ctx* c, const float* src_xyz, const float* ref_xyz, int32_t n) {
    if (!c || !src_xyz || !ref_xyz || n <= 0) { if (c) c->err = "icp_set_convergence_reference: bad argument"; return ICP_ERR_INVALID_ARG; }
    int rc;
    DrainOnError guard(c);
    if ((rc = set_device(c))) return rc;
    if ((rc = upload3(c, src_xyz, n, n, 0.f, c->conv_src.x, c->conv_src.y, c->conv_src.z))) return rc;
    if ((rc = upload3(c, ref_xyz, n, n, 0.f, c->conv_ref.x, c->conv_ref.y, c->conv_ref.z))) return rc;
    c->conv_n = n;
    return guard.done();
}

static int enqueue_fontana(icp_ctx* c, float* d_out) {
    int rc;
    if ((rc = ensure(c, c->fontana_partials, (size_t)256 * 5 * 8))) return rc;
    double* cpart = c->fontana_partials.as<double>(); double* epart = cpart + 256 * 4;
    hipLaunchKernelGGL(k_fontana_centroid, dim3(256), dim3(256), 0, c->stream, c->conv_src.x.as<float>(), c->conv_src.y.as<float>(), c->conv_src.z.as<float>(),
                       c->conv_n, c->ps.as<PoseState>(), cpart);
    hipLaunchKernelGGL(k_fontana_error, dim3(256), dim3(256), 0, c->stream, c->conv_src.x.as<float>(), c->conv_src.y.as<float>(), c->conv_src.z.as<float>(),
                       c->conv_ref.x.as<float>(), c->conv_ref.y.as<float>(), c->conv_ref.z.as<float>(), c->conv_n, c->ps.as<PoseState>(), cpart, 256, epart);
    hipLaunchKernelGGL(k_fontana_finish, dim3(1), dim3(64), 0, c->stream, epart, 256, c->conv_n, d_out);
    HIPCK(c, hipGetLastError());
    return ICP_OK;
}

int icp_benchmark_error(icp_ctx* c, const float pose[16], float* error_out) {
    if (!c || !pose || !error_out) return ICP_ERR_INVALID_ARG;
    if (c->conv_n <= 0) { c->err = "icp_benchmark_error: no convergence reference set"; return ICP_ERR_INVALID_ARG; }
    int rc;
    DrainOnError guard(c);
    if ((rc = set_device(c))) return rc;
    if ((rc = write_pose(c, pose))) return rc;
    if ((rc = ensure(c, c->rmse_out, 4))) return rc;
    if ((rc = enqueue_fontana(c, c->rmse_out.as<float>()))) return rc;
    HIPCK(c, hipMemcpyAsync(error_out, c->rmse_out.p, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return guard.done();
}

int icp_rmse(icp_ctx* c, const float pose[16], float* rmse_out) {
    if (!c || !pose || !rmse_out) return ICP_ERR_INVALID_ARG;
    if (c->conv_n <= 0) { c->err = "icp_rmse: no convergence reference set"; return ICP_ERR_INVALID_ARG; }
    int rc;
    DrainOnError guard(c);
    if ((rc = set_device(c))) return rc;
    if ((rc = write_pose(c, pose))) return rc;
    if ((rc = ensure(c, c->rmse_partials, 256 * 2 * 8))) return rc;
    if ((rc = ensure(c, c->rmse_out, 4))) return rc;
    hipLaunchKernelGGL(k_rmse_partial, dim3(256), dim3(256), 0, c->stream, c->conv_src.x.as<float>(), c->conv_src.y.as<float>(), c->conv_src.z.as<float>(),
                       c->conv_ref.x.as<float>(), c->conv_ref.y.as<float>(), c->conv_ref.z.as<float>(), c->conv_n, c->ps.as<PoseState>(), c->rmse_partials.as<double>());
    hipLaunchKernelGGL(k_rmse_finish, dim3(1), dim3(64), 0, c->stream, c->rmse_partials.as<double>(), 256, c->rmse_out.as<float>());
    HIPCK(c, hipGetLastError());
    HIPCK(c, hipMemcpyAsync(rmse_out, c->rmse_out.p, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return guard.done();
}

int icp_backproject_depth(icp_ctx* c, const float* depth, const uint8_t* rgbx, float fx, float fy, float cx, float cy,
                          const float extrinsics[16], int32_t width, int32_t height, float max_distance, int32_t fix_color_index,
                          float* xyz_out, float* normals_out, uint8_t* rgba_out, uint8_t* valid_out) {
    if (!c || !depth || !extrinsics || !xyz_out || !normals_out || width <= 0 || height <= 0) { if (c) c->err = "icp_backproject_depth: bad argument"; return ICP_ERR_INVALID_ARG; }
    int rc;
    DrainOnError guard(c);
    if ((rc = set_device(c))) return rc;
    const size_t n = (size_t)width * height;
    // depthExtrinsics.inverse() (PointCloud.h:88-90): rigid/affine 4x4, inverted in fp64 and rounded once
    double R[9], t[3];
    for (int r = 0; r < 3; r++) { for (int k = 0; k < 3; k++) R[r * 3 + k] = extrinsics[k * 4 + r]; t[r] = extrinsics[12 + r]; }
    const double det = R[0] * (R[4] * R[8] - R[5] * R[7]) - R[1] * (R[3] * R[8] - R[5] * R[6]) + R[2] * (R[3] * R[7] - R[4] * R[6]);
    double Ri[9] = {(R[4] * R[8] - R[5] * R[7]) / det, (R[2] * R[7] - R[1] * R[8]) / det, (R[1] * R[5] - R[2] * R[4]) / det,
                    (R[5] * R[6] - R[3] * R[8]) / det, (R[0] * R[8] - R[2] * R[6]) / det, (R[2] * R[3] - R[0] * R[5]) / det,
                    (R[3] * R[7] - R[4] * R[6]) / det, (R[1] * R[6] - R[0] * R[7]) / det, (R[0] * R[4] - R[1] * R[3]) / det};
    float inv[12];
    for (int i = 0; i < 9; i++) inv[i] = (float)Ri[i];
    for (int r = 0; r < 3; r++) inv[9 + r] = (float)(-(Ri[r * 3] * t[0] + Ri[r * 3 + 1] * t[1] + Ri[r * 3 + 2] * t[2]));
    // fixed layout, colour slots always reserved: [depth 4n | rgbx 4n | inverse 64 | xyz 12n | normals 12n | rgba 4n | valid n]
    const size_t bytes = n * 4 + n * 4 + 64 + n * 12 * 2 + n * 4 + n;
    if ((rc = ensure(c, c->staging, bytes + 256))) return rc;
    char* base = c->staging.as<char>();
    float* d_depth = (float*)base; uint8_t* d_rgbx = (uint8_t*)(base + n * 4); float* d_inv = (float*)(base + n * 8);
    float* d_xyz = (float*)(base + n * 8 + 64); float* d_nrm = d_xyz + n * 3; uint8_t* d_rgba = (uint8_t*)(d_nrm + n * 3); uint8_t* d_valid = d_rgba + n * 4;
    HIPCK(c, hipMemcpyAsync(d_depth, depth, n * 4, hipMemcpyHostToDevice, c->stream));
    if (rgbx) HIPCK(c, hipMemcpyAsync(d_rgbx, rgbx, n * 4, hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipMemcpyAsync(d_inv, inv, sizeof(inv), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_backproject, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, d_depth, rgbx ? d_rgbx : nullptr, width, height, fx, fy, cx, cy, d_inv,
                       max_distance / 2.f, fix_color_index, d_xyz, d_nrm, (rgbx && rgba_out) ? d_rgba : nullptr, valid_out ? d_valid : nullptr);
    HIPCK(c, hipGetLastError());
    HIPCK(c, hipMemcpyAsync(xyz_out, d_xyz, n * 12, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipMemcpyAsync(normals_out, d_nrm, n * 12, hipMemcpyDeviceToHost, c->stream));
    if (rgbx && rgba_out) HIPCK(c, hipMemcpyAsync(rgba_out, d_rgba, n * 4, hipMemcpyDeviceToHost, c->stream));
    if (valid_out) HIPCK(c, hipMemcpyAsync(valid_out, d_valid, n, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return guard.done();
}

int icp_estimate_normals(icp_ctx* c, const float* xyz, int32_t n, int32_t k, const float viewpoint[3], float* normals_out, float* curvature_out) {
    if (!c || !xyz || !normals_out || n <= 0 || k < 3 || k > 8) { if (c) c->err = "icp_estimate_normals: bad argument (k must be 3..8)"; return ICP_ERR_INVALID_ARG; }
    int rc;
    DrainOnError guard(c);
    if ((rc = set_device(c))) return rc;
    Cloud& cl = c->nrm_cloud; Bvh& b = c->nrm_bvh;
    if ((rc = upload_cloud(c, cl, xyz, nullptr, nullptr, n, false))) return rc;
    b.valid = false;
    if ((rc = finite_list(c, cl, false, c->tgt_flag, c->nrm_finite, &b.n_valid))) return rc;      // (tgt_flag is scratch here: only its list is kept)
    b.d_finite = c->nrm_finite.as<int>(); b.n_ids = n;
    CoordPtrs<3> cp; cp.c[0] = cl.x.as<float>(); cp.c[1] = cl.y.as<float>(); cp.c[2] = cl.z.as<float>();
    if ((rc = build_bvh<3>(c, b, cp))) return rc;
    BvhViewT<3> bv; bv.leaves = b.leaves.as<BvhLeafT<3>>(); bv.nodes = b.nodes.as<BvhNodeT<3>>(); bv.n_valid = b.n_valid; bv.Lp = b.Lp; bv.tgt = cp;
    bv.qnodes = b.qnodes.as<BvhQuadT<3>>(); bv.Lq = b.Lq; bv.recs = b.recs.as<TgtRec>(); bv.pos_of = b.pos_of.as<int>();
    int depth = 0; while ((1 << depth) < b.Lp) depth++;
    if ((rc = ensure(c, c->staging, (size_t)n * 16))) return rc;
    float* d_n = c->staging.as<float>(); float* d_c = d_n + (size_t)n * 3;
    const float vx = viewpoint ? viewpoint[0] : 0.f, vy = viewpoint ? viewpoint[1] : 0.f, vz = viewpoint ? viewpoint[2] : 0.f;
    const dim3 grid((n + BVH_THREADS - 1) / BVH_THREADS), block(BVH_THREADS); const size_t lds = (size_t)(depth + 1) * BVH_THREADS * 2;
    switch (k) {
        case 3: hipLaunchKernelGGL(k_normals_knn<3>, grid, block, lds, c->stream, bv, n, depth, vx, vy, vz, d_n, d_c); break;
        case 4: hipLaunchKernelGGL(k_normals_knn<4>, grid, block, lds, c->stream, bv, n, depth, vx, vy, vz, d_n, d_c); break;
        case 5: hipLaunchKernelGGL(k_normals_knn<5>, grid, block, lds, c->stream, bv, n, depth, vx, vy, vz, d_n, d_c); break;
        case 6: hipLaunchKernelGGL(k_normals_knn<6>, grid, block, lds, c->stream, bv, n, depth, vx, vy, vz, d_n, d_c); break;
        case 7: hipLaunchKernelGGL(k_normals_knn<7>, grid, block, lds, c->stream, bv, n, depth, vx, vy, vz, d_n, d_c); break;
        default: hipLaunchKernelGGL(k_normals_knn<8>, grid, block, lds, c->stream, bv, n, depth, vx, vy, vz, d_n, d_c); break;
    }
    HIPCK(c, hipGetLastError());
    HIPCK(c, hipMemcpyAsync(normals_out, d_n, (size_t)n * 12, hipMemcpyDeviceToHost, c->stream));
    if (curvature_out) HIPCK(c, hipMemcpyAsync(curvature_out, d_c, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return guard.done();
}

static int transform_common(icp_ctx* c, const float* in, int32_t n, const float pose[16], float* out, int normals) {
    if (!c || !in || !out || !pose || n <= 0) { if (c) c->err = "icp_transform: bad argument"; return ICP_ERR_INVALID_ARG; }
    int rc;
    DrainOnError guard(c);
    if ((rc = set_device(c))) return rc;
    if ((rc = write_pose(c, pose))) return rc;
    if ((rc = ensure(c, c->staging, (size_t)n * 24))) return rc;
    float* din = c->staging.as<float>(); float* dout = din + (size_t)n * 3;
    HIPCK(c, hipMemcpyAsync(din, in, (size_t)n * 12, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_transform_aos, dim3((n + 255) / 256), dim3(256), 0, c->stream, din, n, c->ps.as<PoseState>(), normals, dout);
    HIPCK(c, hipGetLastError());
    HIPCK(c, hipMemcpyAsync(out, dout, (size_t)n * 12, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return guard.done();
}
// Development builds (ICP_DEBUG_STEPS=1): nodes | leaves << 16 visited by the walk of each query of the LAST matcher launch, in the
// order the launch indexed its queries (Morton order for a run); 0 = verified without a walk, -2 = second tier (two leaves), -1 = a walk whose length was not recorded.  Per-query lengths need
// ICP_SHARE_WALKS=0 (a shared walk has no per-query length); with ICP_DEBUG_TIMES=1 the buffer holds per-wave phase stamps instead (tools/dev_wave_times.py).
int icp_debug_steps(icp_ctx* c, int32_t* out, int32_t n) {
    if (!c || !out || n <= 0 || !c->dbg_steps.p || (size_t)n * 4 > c->dbg_steps.cap) return ICP_ERR_INVALID_ARG;
    int rc;
    if ((rc = set_device(c))) return rc;
    HIPCK(c, hipMemcpy(out, c->dbg_steps.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return ICP_OK;
}

// Development / test hooks (not part of icp_hip.h; called by tests/ through ctypes).
//   icp_debug_counters        : how many runs of this context took the merged loop, and how many of those had to be repeated with the
//                               separate k_reduce_solve launches (rank-deficient system, or a bounded wait that ran out).
//   icp_debug_poison_handover : leaves a stale, valid-looking total in slot `slot` of k_reduce_solve's hand-over area -- what a run cut
//                               short between a block's publish and block 0's re-arm would leave behind.  The next call must not see it.
int icp_debug_ring_times(icp_ctx* c, int32_t* out, int32_t n) {     // development builds (ICP_DEBUG_TIMES): the reducer blocks' clock stamps of the last merged launch
#if ICP_DEBUG_TIMES
    if (!c || !out || n < (NSUM_USED + 1) * 8) return ICP_ERR_INVALID_ARG;
    int rc;
    if ((rc = set_device(c))) return rc;
    HIPCK(c, hipMemcpyFromSymbol(out, HIP_SYMBOL(icpdev::g_ring_dbg), (size_t)(NSUM_USED + 1) * 8 * 4));
    return ICP_OK;
#else
    (void)c; (void)out; (void)n;
    return ICP_ERR_INVALID_ARG;
#endif
}
int icp_debug_gx_counters(icp_ctx* c, uint32_t* out16, int32_t reset) {     // development builds (ICP_DEBUG_TIMES): events of the hand-over between blocks since the last reset
#if ICP_DEBUG_TIMES
    if (!c || !out16) return ICP_ERR_INVALID_ARG;
    int rc;
    if ((rc = set_device(c))) return rc;
    HIPCK(c, hipStreamSynchronize(c->stream));
    HIPCK(c, hipMemcpyFromSymbol(out16, HIP_SYMBOL(icpdev::g_gx_dbg), 64));
    if (reset == 2) HIPCK(c, hipMemcpyFromSymbol(out16, HIP_SYMBOL(icpdev::g_walk_trace), 256));      // (reset == 2: the caller's buffer has 64 words and wants the trace of the last sparse walk instead, tools/dev_walk_trace.py)
    if (reset) { uint32_t z[16] = {0}; HIPCK(c, hipMemcpyToSymbol(HIP_SYMBOL(icpdev::g_gx_dbg), z, 64)); }
    return ICP_OK;
#else
    (void)c; (void)out16; (void)reset;
    return ICP_ERR_INVALID_ARG;
#endif
}
//   icp_debug_wave_slot       : host evaluation of the fused matcher's block -> wave mapping (fused_wave_slot): which stretch of 64 queries
//                               wave w of logical block lb takes in a grid of mgrid blocks; *waves_per_block receives BVH_THREADS / 64.  No GPU needed.
//   icp_debug_pos_of_mismatches: entries of the resident target's position-by-index map that do not point back at their record (must be 0).
int icp_debug_wave_slot(int32_t lb, int32_t w, int32_t mgrid, int32_t* waves_per_block) {
    if (waves_per_block) *waves_per_block = BVH_THREADS / WAVE;
    if (lb < 0 || lb >= mgrid || w < 0 || w >= BVH_THREADS / WAVE) return -1;
    return icpdev::fused_wave_slot(lb, w, mgrid);
}
__global__ void k_debug_pos_of(const icpdev::TgtRec* recs, const int* pos_of, int n_slots, int* bad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_slots && recs[i].idx >= 0 && pos_of[recs[i].idx] != i) atomicAdd(bad, 1);
}
int icp_debug_pos_of_mismatches(icp_ctx* c, int32_t* n_bad, int32_t* n_checked) {
    if (!c || !n_bad || !c->bvh.valid) return ICP_ERR_INVALID_ARG;
    int rc;
    if ((rc = set_device(c))) return rc;
    const int n_slots = (c->bvh.n_leaves > 0 ? c->bvh.n_leaves : 1) * BVH_LEAF;
    if ((rc = ensure(c, c->d_count, 4))) return rc;
    HIPCK(c, hipMemsetAsync(c->d_count.p, 0, 4, c->stream));
    hipLaunchKernelGGL(k_debug_pos_of, dim3((n_slots + 255) / 256), dim3(256), 0, c->stream, c->bvh.recs.as<icpdev::TgtRec>(), c->bvh.pos_of.as<int>(), n_slots, c->d_count.as<int>());
    HIPCK(c, hipMemcpyAsync(n_bad, c->d_count.p, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    if (n_checked) *n_checked = c->bvh.n_valid;
    return ICP_OK;
}
int icp_debug_counters(icp_ctx* c, int32_t* merged_runs, int32_t* merged_fallbacks) {
    if (!c) return ICP_ERR_INVALID_ARG;
    if (merged_runs) *merged_runs = c->merged_runs + c->loop_runs;
    if (merged_fallbacks) *merged_fallbacks = c->merged_fallbacks;
    return ICP_OK;
}
int icp_debug_poison_handover(icp_ctx* c, int32_t slot, double value) {
    if (!c || slot < 0 || slot >= NSUM) return ICP_ERR_INVALID_ARG;
    int rc;
    if ((rc = set_device(c))) return rc;
    if (!c->totals.p && (rc = rearm_handover(c))) return rc;
    HIPCK(c, hipMemcpyAsync(c->totals.as<double>() + slot, &value, 8, hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return ICP_OK;
}

// ---- hardware self test (not part of icp_hip.h; called by tests/ through ctypes) ------------------------------------------
// One wave folds n_values (<= 32) doubles per lane with wave_transpose_reduce_gen; out[v] = the wave total of value v read from
// the lane wave_value_of_lane says holds it.  tests/test_gpu_selftest.py replays the same pairing with numpy: bit-identical.
int icp_selftest_wave_reduce(icp_ctx* c, const double* in, double* out, int32_t* lane_of) {
    if (!c || !in || !out || !lane_of) return ICP_ERR_INVALID_ARG;
    int rc;
    if ((rc = set_device(c))) return rc;
    DrainOnError guard(c);
    if ((rc = ensure(c, c->staging, 64 * 27 * 8 + 27 * 8 + 27 * 4 + 64))) return rc;
    double* d_in = c->staging.as<double>(); double* d_out = d_in + 64 * 27; int* d_lane = (int*)(d_out + 27);
    HIPCK(c, hipMemcpyAsync(d_in, in, 64 * 27 * 8, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_selftest_wave_reduce, dim3(1), dim3(64), 0, c->stream, d_in, d_out, d_lane);
    HIPCK(c, hipGetLastError());
    HIPCK(c, hipMemcpyAsync(out, d_out, 27 * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipMemcpyAsync(lane_of, d_lane, 27 * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return guard.done();
}

int icp_transform_points(icp_ctx* c, const float* xyz, int32_t n, const float pose[16], float* out) { return transform_common(c, xyz, n, pose, out, 0); }
int icp_transform_normals(icp_ctx* c, const float* nrm, int32_t n, const float pose[16], float* out) { return transform_common(c, nrm, n, pose, out, 1); }

}  // extern "C"
