// =====================================================================================
// icp_batch.hip -- batches of independent scan pairs and the pose gather (libicp_hip.so).
//
// The reference aligns the ETH pairs in a plain loop and carries no state from one index to the next
// (main.cpp:411-498, experiment.cpp:319-396).  A batch therefore shards with no data-path exchange:
//   * inside one GPU  : icp_batch_run -- one host thread per context (= HIP stream), every thread takes the next pair
//                       that has not been started; uploads, index builds and iterations of different pairs overlap;
//   * across the GPUs : pair p -> rank p % n_ranks, ONE ncclAllGather of ceil(P / n_ranks) x 16 floats per rank at the
//                       end of the batch (icp_gather_poses).  RCCL is loaded with dlopen on first use, so a single-GPU
//                       host needs no librccl at all.
// Only the public C ABI of icp_hip.h is used here (a context is driven exactly as a C++14 host would drive it).
// =====================================================================================
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>          // types only: every RCCL entry point is resolved with dlsym
#include "../../include/icp_hip.h"

extern "C" {

int32_t icp_pair_owner(int32_t pair, int32_t n_ranks) { return (n_ranks > 0 && pair >= 0) ? pair % n_ranks : -1; }

int32_t icp_pairs_of_rank(int32_t n_pairs, int32_t rank, int32_t n_ranks) {
    if (n_ranks <= 0 || rank < 0 || rank >= n_ranks || n_pairs <= rank) return 0;
    return (n_pairs - rank + n_ranks - 1) / n_ranks;
}

int icp_internal_promote_source_to_target(icp_ctx* ctx);      // icp_hip.hip: the resident source becomes the target (no trip through the host)

// A pair whose target IS the source of the pair before it (same arrays, same size, no colours): consecutive scan pairs (k, k + 1) share
// scan k + 1 -- the reference loads it twice (main.cpp:411-498).
static bool target_is_previous_source(const icp_pair& prev, const icp_pair& cur) {
    return cur.tgt_xyz == prev.src_xyz && cur.tgt_normals == prev.src_normals && cur.n_tgt == prev.n_src && !cur.tgt_rgba && !prev.src_rgba && cur.tgt_xyz != nullptr;
}

int icp_batch_run(icp_ctx* const* ctxs, int32_t n_ctx, const icp_pair* pairs, int32_t n_pairs, float* poses_out, int32_t* status_out) {
    if (!ctxs || n_ctx <= 0 || n_pairs < 0 || (n_pairs > 0 && (!pairs || !poses_out))) return ICP_ERR_INVALID_ARG;
    for (int i = 0; i < n_ctx; i++) if (!ctxs[i]) return ICP_ERR_INVALID_ARG;
    std::vector<int32_t> st((size_t)n_pairs, ICP_OK);
    const int nt = n_ctx < n_pairs ? n_ctx : n_pairs;
    // A batch in which every pair's target is the source of the pair before it (a scan sequence) is dealt out in CONTIGUOUS runs, one per
    // context: inside a run the shared scan is uploaded once -- as the source of pair k; it is then promoted to the target of pair k + 1 on
    // the device.  Any other batch: every thread takes the next pair that has not been started.
    bool chain = n_pairs > 1;
    for (int32_t p = 1; p < n_pairs && chain; p++) chain = target_is_previous_source(pairs[p - 1], pairs[p]);
    std::atomic<int32_t> next(0);
    // ICP_HIP_BATCH_TIMES=1: where the host threads' time goes (wall time inside the three calls of a pair, summed over the pairs) -> stderr
    const bool times = getenv("ICP_HIP_BATCH_TIMES") != nullptr;
    std::atomic<long long> t_tgt(0), t_src(0), t_run(0);
    auto now = [] { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    auto align = [&](icp_ctx* c, int32_t p, bool promote) {
        const icp_pair& q = pairs[p];
        float* pose = poses_out + (size_t)p * 16;
        memcpy(pose, q.initial_pose, 64);
        const long long a0 = times ? now() : 0;
        int rc = promote ? icp_internal_promote_source_to_target(c)
                         : icp_set_target(c, q.tgt_xyz, q.tgt_normals, q.tgt_rgba, q.n_tgt);                // buildIndex, ICPOptimizer.h:532-535
        const long long a1 = times ? now() : 0;
        if (!rc) rc = icp_set_source(c, q.src_xyz, q.src_normals, q.src_rgba, q.n_src);
        const long long a2 = times ? now() : 0;
        if (!rc) { int32_t n = 0; rc = icp_run(c, pose, nullptr, 0, &n); }                           // estimatePose, main.cpp:457
        if (times) { const long long a3 = now(); t_tgt += a1 - a0; t_src += a2 - a1; t_run += a3 - a2; }
        st[(size_t)p] = rc;
        return rc;
    };
    const long long b0 = times ? now() : 0;
    auto worker = [&](icp_ctx* c, int t) {
        if (chain) {
            const int32_t lo = (int32_t)((long long)n_pairs * t / nt), hi = (int32_t)((long long)n_pairs * (t + 1) / nt);
            bool src_resident = false;                      // the context holds pairs[p - 1]'s source, untouched since its upload
            for (int32_t p = lo; p < hi; p++) src_resident = align(c, p, src_resident) == ICP_OK;
            return;
        }
        for (;;) {
            const int32_t p = next.fetch_add(1);
            if (p >= n_pairs) return;
            align(c, p, false);
        }
    };
    if (nt <= 1) { if (n_pairs > 0) worker(ctxs[0], 0); }
    else {
        std::vector<std::thread> th;
        th.reserve((size_t)nt);
        for (int i = 0; i < nt; i++) th.emplace_back(worker, ctxs[i], i);
        for (auto& t : th) t.join();
    }
    if (times && n_pairs > 0)
        fprintf(stderr, "icp_batch_run: %d pairs, %d contexts, %.3f ms; per pair inside target %.3f ms, source %.3f ms, run %.3f ms (sum over threads / pairs)\n", n_pairs, nt,
                (now() - b0) * 1e-6, t_tgt.load() * 1e-6 / n_pairs, t_src.load() * 1e-6 / n_pairs, t_run.load() * 1e-6 / n_pairs);
    int first = ICP_OK;
    for (int32_t p = 0; p < n_pairs; p++) {
        if (status_out) status_out[p] = st[(size_t)p];
        if (st[(size_t)p] != ICP_OK && first == ICP_OK) first = st[(size_t)p];
    }
    return first;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------------
// RCCL, resolved at run time
// ---------------------------------------------------------------------------------------------------------------------
namespace {

struct Rccl {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

std::mutex g_mu;
Rccl g_rccl;
thread_local std::string g_err;

bool load_rccl() {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_rccl.h) return true;
    // The RCCL to use is the one that sits next to the HIP runtime THIS library is bound to: its communicators must see the streams
    // and buffers created here.  A process can hold two ROCm stacks -- PyTorch wheels ship their own libamdhip64 / libhsa-runtime64 /
    // librccl: imported first, its libamdhip64 (soname libamdhip64.so.7) also serves this library; imported second, it comes on top of
    // the system stack this library already pulled in -- and an RCCL on the other stack finds "no ROCm-capable device".
    void* h = nullptr;
    Dl_info di;
    const char* forced = getenv("ICP_HIP_RCCL_LIB");      // tests / unusual installations: this file and nothing else
    if (forced && forced[0]) h = dlopen(forced, RTLD_NOW | RTLD_GLOBAL);
    else {
        if (dladdr((const void*)&hipGetDeviceCount, &di) && di.dli_fname) {
            std::string dir(di.dli_fname);
            const size_t slash = dir.rfind('/');
            if (slash != std::string::npos) {
                dir.resize(slash + 1);
                h = dlopen((dir + "librccl.so.1").c_str(), RTLD_NOW | RTLD_GLOBAL);
                if (!h) h = dlopen((dir + "librccl.so").c_str(), RTLD_NOW | RTLD_GLOBAL);
            }
        }
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    }
    if (!h) {
        const char* e = dlerror();                          // (one call: it returns the message AND clears it)
        g_err = std::string(forced && forced[0] ? "ICP_HIP_RCCL_LIB could not be loaded: " : "librccl.so.1 not found: ") + (e ? e : "no further detail from the loader");
        return false;
    }
    Rccl r; r.h = h;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
    r.AllGather = (decltype(r.AllGather))dlsym(h, "ncclAllGather");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.GetErrorString) { g_err = "librccl lacks an expected symbol"; return false; }
    g_rccl = r;
    return true;
}

int nccl_fail(const char* what, ncclResult_t e) {
    g_err = std::string(what) + " failed: " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "?");
    return ICP_ERR_COMM;
}
int hip_fail(const char* what, hipError_t e) {
    g_err = std::string(what) + " failed: " + hipGetErrorString(e);
    return ICP_ERR_HIP;
}

}  // namespace

struct icp_comm {
    int device = 0, n_ranks = 1, rank = 0;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    float* d_send = nullptr; float* d_recv = nullptr; size_t cap_pairs = 0;      // device staging: [cap x 16] and [n_ranks x cap x 16]
    std::vector<float> h_recv;
};

extern "C" {

const char* icp_comm_last_error(void) { return g_err.c_str(); }

int icp_comm_unique_id(uint8_t id_out[ICP_COMM_ID_BYTES]) {
    static_assert(ICP_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    if (!id_out) return ICP_ERR_INVALID_ARG;
    if (!load_rccl()) return ICP_ERR_COMM;
    ncclUniqueId id;
    const ncclResult_t e = g_rccl.GetUniqueId(&id);
    if (e != ncclSuccess) return nccl_fail("ncclGetUniqueId", e);
    memcpy(id_out, id.internal, ICP_COMM_ID_BYTES);
    return ICP_OK;
}

int icp_comm_destroy(icp_comm* c) {
    if (!c) return ICP_ERR_INVALID_ARG;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    if (c->d_send) (void)hipFree(c->d_send);
    if (c->d_recv) (void)hipFree(c->d_recv);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return ICP_OK;
}

int icp_comm_create(int device, int32_t n_ranks, int32_t rank, const uint8_t id[ICP_COMM_ID_BYTES], icp_comm** out) {
    if (!out) return ICP_ERR_INVALID_ARG;
    *out = nullptr;
    if (!id || n_ranks <= 0 || rank < 0 || rank >= n_ranks) { g_err = "icp_comm_create: bad argument"; return ICP_ERR_INVALID_ARG; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { g_err = "no HIP device"; return ICP_ERR_NO_DEVICE; }
    if (device < 0 || device >= ndev) { g_err = "icp_comm_create: no such device"; return ICP_ERR_INVALID_ARG; }
    if (!load_rccl()) return ICP_ERR_COMM;
    hipError_t he = hipSetDevice(device);
    if (he != hipSuccess) return hip_fail("hipSetDevice", he);
    icp_comm* c = new icp_comm();
    c->device = device; c->n_ranks = n_ranks; c->rank = rank;
    he = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (he != hipSuccess) { delete c; return hip_fail("hipStreamCreateWithFlags", he); }
    ncclUniqueId nid; memcpy(nid.internal, id, ICP_COMM_ID_BYTES);
    const ncclResult_t e = g_rccl.CommInitRank(&c->comm, n_ranks, nid, rank);
    if (e != ncclSuccess) { c->comm = nullptr; icp_comm_destroy(c); return nccl_fail("ncclCommInitRank", e); }
    *out = c;
    return ICP_OK;
}

int icp_gather_poses(icp_comm* c, const float* local_poses, int32_t n_local, int32_t n_pairs, float* all_poses_out) {
    if (!c || n_pairs < 0 || n_local < 0 || (n_local > 0 && !local_poses) || (n_pairs > 0 && !all_poses_out)) { g_err = "icp_gather_poses: bad argument"; return ICP_ERR_INVALID_ARG; }
    if (n_local != icp_pairs_of_rank(n_pairs, c->rank, c->n_ranks)) { g_err = "icp_gather_poses: n_local does not match the round-robin share of this rank"; return ICP_ERR_INVALID_ARG; }
    if (n_pairs == 0) return ICP_OK;
    hipError_t he = hipSetDevice(c->device);
    if (he != hipSuccess) return hip_fail("hipSetDevice", he);
    const size_t cap = (size_t)((n_pairs + c->n_ranks - 1) / c->n_ranks);       // every rank contributes the same count (ncclAllGather)
    if (cap > c->cap_pairs) {
        if (c->d_send) (void)hipFree(c->d_send);
        if (c->d_recv) (void)hipFree(c->d_recv);
        c->d_send = c->d_recv = nullptr; c->cap_pairs = 0;
        if ((he = hipMalloc((void**)&c->d_send, cap * 64)) != hipSuccess) return hip_fail("hipMalloc", he);
        if ((he = hipMalloc((void**)&c->d_recv, cap * 64 * (size_t)c->n_ranks)) != hipSuccess) return hip_fail("hipMalloc", he);
        c->cap_pairs = cap;
    }
    c->h_recv.resize(cap * 16 * (size_t)c->n_ranks);
    if ((he = hipMemsetAsync(c->d_send, 0, cap * 64, c->stream)) != hipSuccess) return hip_fail("hipMemsetAsync", he);
    if (n_local > 0 && (he = hipMemcpyAsync(c->d_send, local_poses, (size_t)n_local * 64, hipMemcpyHostToDevice, c->stream)) != hipSuccess) return hip_fail("hipMemcpyAsync", he);
    const ncclResult_t e = g_rccl.AllGather(c->d_send, c->d_recv, cap * 16, ncclFloat, c->comm, c->stream);      // the single collective of the batch
    if (e != ncclSuccess) { (void)hipStreamSynchronize(c->stream); return nccl_fail("ncclAllGather", e); }
    if ((he = hipMemcpyAsync(c->h_recv.data(), c->d_recv, c->h_recv.size() * 4, hipMemcpyDeviceToHost, c->stream)) != hipSuccess) return hip_fail("hipMemcpyAsync", he);
    if ((he = hipStreamSynchronize(c->stream)) != hipSuccess) return hip_fail("hipStreamSynchronize", he);
    for (int32_t p = 0; p < n_pairs; p++) {                                     // rank-major blocks -> pair order
        const int32_t r = p % c->n_ranks, k = p / c->n_ranks;
        memcpy(all_poses_out + (size_t)p * 16, c->h_recv.data() + ((size_t)r * cap + (size_t)k) * 16, 64);
    }
    return ICP_OK;
}

}  // extern "C"
