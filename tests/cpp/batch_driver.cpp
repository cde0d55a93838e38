// C++14 host for configs[3] (the loop over independent ETH pairs, main.cpp:411-498) on the plain C ABI -- no Python, no torch:
// K contexts on one device, ONE icp_batch_run for all pairs, then the pose gather on a one-rank RCCL communicator
// (icp_comm_unique_id / icp_comm_create / icp_gather_poses; with more ranks the id travels by whatever the host uses, e.g. MPI).
//   usage: batch_driver <dump.bin> <contexts> <iterations>
//   dump : int32 n_pairs, then per pair { int32 n_src, src xyz, src normals, int32 n_tgt, tgt xyz, tgt normals } (fp32)
#include "icp_hip.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

struct Cloud { std::vector<float> xyz, nrm; int32_t n = 0; };
static bool read_cloud(FILE* f, Cloud& c) {
    if (fread(&c.n, 4, 1, f) != 1 || c.n <= 0) return false;
    c.xyz.resize((size_t)c.n * 3); c.nrm.resize((size_t)c.n * 3);
    return fread(c.xyz.data(), 12, (size_t)c.n, f) == (size_t)c.n && fread(c.nrm.data(), 12, (size_t)c.n, f) == (size_t)c.n;
}

int main(int argc, char** argv) {
    if (argc < 4) { std::fprintf(stderr, "usage: %s dump.bin contexts iterations\n", argv[0]); return 2; }
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    int32_t n_pairs = 0;
    if (fread(&n_pairs, 4, 1, f) != 1 || n_pairs <= 0) return 2;
    std::vector<Cloud> src((size_t)n_pairs), tgt((size_t)n_pairs);
    for (int p = 0; p < n_pairs; p++) if (!read_cloud(f, src[(size_t)p]) || !read_cloud(f, tgt[(size_t)p])) return 2;
    std::fclose(f);

    const int n_ctx = std::atoi(argv[2]);
    std::vector<icp_ctx*> ctxs((size_t)n_ctx, nullptr);
    icp_params prm; icp_params_default(&prm);
    prm.metric = ICP_METRIC_POINT_TO_PLANE; prm.max_distance = 10.f; prm.n_iterations = std::atoi(argv[3]); prm.knn_backend = ICP_KNN_LBVH;   // main.cpp:360-366
    for (int i = 0; i < n_ctx; i++) {
        if (icp_ctx_create(0, &ctxs[(size_t)i]) != ICP_OK) { std::fprintf(stderr, "no device\n"); return 3; }
        if (icp_set_params(ctxs[(size_t)i], &prm) != ICP_OK) return 3;
    }
    std::vector<icp_pair> pairs((size_t)n_pairs);
    for (int p = 0; p < n_pairs; p++) {
        icp_pair& q = pairs[(size_t)p]; std::memset(&q, 0, sizeof(q));
        q.src_xyz = src[(size_t)p].xyz.data(); q.src_normals = src[(size_t)p].nrm.data(); q.n_src = src[(size_t)p].n;
        q.tgt_xyz = tgt[(size_t)p].xyz.data(); q.tgt_normals = tgt[(size_t)p].nrm.data(); q.n_tgt = tgt[(size_t)p].n;
        for (int k = 0; k < 4; k++) q.initial_pose[k * 5] = 1.f;                  // Matrix4f::Identity(), main.cpp:416
    }
    std::vector<float> poses((size_t)n_pairs * 16), gathered((size_t)n_pairs * 16);
    std::vector<int32_t> status((size_t)n_pairs);
    const int rc = icp_batch_run(ctxs.data(), n_ctx, pairs.data(), n_pairs, poses.data(), status.data());
    std::printf("batch rc %d\n", rc);

    uint8_t id[ICP_COMM_ID_BYTES]; icp_comm* comm = nullptr;
    int grc = icp_comm_unique_id(id);
    if (grc == ICP_OK) grc = icp_comm_create(0, 1, 0, id, &comm);
    if (grc == ICP_OK) grc = icp_gather_poses(comm, poses.data(), icp_pairs_of_rank(n_pairs, 0, 1), n_pairs, gathered.data());
    std::printf("gather rc %d %s\n", grc, grc == ICP_OK ? "" : icp_comm_last_error());
    if (comm) icp_comm_destroy(comm);
    for (int p = 0; p < n_pairs; p++) {
        std::printf("pose %d status %d", p, status[(size_t)p]);
        for (int k = 0; k < 16; k++) std::printf(" %.9g", (grc == ICP_OK ? gathered : poses)[(size_t)p * 16 + k]);
        std::printf("\n");
    }
    for (icp_ctx* c : ctxs) icp_ctx_destroy(c);
    return rc == ICP_OK && grc == ICP_OK ? 0 : 1;
}
