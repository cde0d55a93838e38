// dev_normals.hpp -- K-NN surface normals, gathers, Morton keys of the queries.
// Part of icp_device.hpp (included from there, inside namespace icpdev); see that file for the build contract.
// ------------------------------------------------------------------------------------------------
// Surface normals from the K nearest neighbours (the PCL NormalEstimation the reference runs on the ETH scans,
// PointCloud.h:41-76: setKSearch(5), viewpoint (0,0,0)): K-NN over the cloud's own kd-ordered BVH (the point itself is its
// first neighbour, as with pcl::search::KdTree), fp64 covariance of the K points, eigenvector of the smallest eigenvalue
// (fp64 Jacobi), flipped towards the viewpoint (pcl::flipNormalTowardsViewpoint), curvature = l0 / (l0 + l1 + l2).
// Neighbour sets are the exact K smallest (d2, index) pairs.  PCL itself is absent here: parity unpinned, checked against numpy.
template <int n> __device__ inline void jacobi_eig_sym(double* A, double* V, double* ev);     // defined with the solvers below

template <int K>
__device__ __forceinline__ void knn_insert(float (&bd)[K], int (&bj)[K], float d, int j) {
    // keep (bd, bj) sorted ascending by (d, j); called only when (d, j) beats the current worst
    bd[K - 1] = d; bj[K - 1] = j;
#pragma unroll
    for (int q = K - 1; q > 0; q--) {
        const bool sw = (bd[q] < bd[q - 1]) | ((bd[q] == bd[q - 1]) & (bj[q] < bj[q - 1]));
        const float td = bd[q]; const int tj = bj[q];
        bd[q] = sw ? bd[q - 1] : bd[q]; bj[q] = sw ? bj[q - 1] : bj[q];
        bd[q - 1] = sw ? td : bd[q - 1]; bj[q - 1] = sw ? tj : bj[q - 1];
    }
}

template <int K>
__global__ __launch_bounds__(BVH_THREADS) void k_normals_knn(const BvhViewT<3> bv, int n, int tree_depth, float vpx, float vpy, float vpz,
                                                             float* __restrict__ nrm_out /* AoS n x 3 */, float* __restrict__ curv_out) {
    extern __shared__ unsigned short bvh_lb16[];
    const int tid = threadIdx.x;
    const int i = blockIdx.x * BVH_THREADS + tid;
    if (i >= n) return;
    const float px = bv.tgt.c[0][i], py = bv.tgt.c[1][i], pz = bv.tgt.c[2][i];
    float nx = NAN, ny = NAN, nz = NAN, curv = NAN;
    if (finite3(px, py, pz) && bv.n_valid >= 3) {
        float bd[K]; int bj[K];
#pragma unroll
        for (int q = 0; q < K; q++) { bd[q] = FLT_MAX; bj[q] = 0x7fffffff; }
        f2 p2[3] = {{px, px}, {py, py}, {pz, pz}};
        TravState st; st.depth = 0; st.idx = 0; st.pending = 0u; st.alive = true;
        float unused_minlb = FLT_MAX;
        while (st.alive) {
            while (st.alive && st.depth < tree_depth) {
                const f2 l = pair_lb<3>(bv.nodes + ((1 << st.depth) - 1 + st.idx), p2);
                const bool swap = l.y < l.x;
                const float ln = swap ? l.y : l.x, lf = swap ? l.x : l.y;
                const float worst = bd[K - 1];
                const bool take_near = !(ln * 0.99999f > worst), take_far = !(lf * 0.99999f > worst);
                if (take_near) {
                    if (take_far) { bvh_lb16[st.depth * BVH_THREADS + tid] = (unsigned short)(__float_as_uint(lf) >> 16); st.pending |= 1u << st.depth; }
                    st.idx = 2 * st.idx + (swap ? 1 : 0); st.depth++;
                } else st.alive = false;
                trav_pop(st, bvh_lb16, tid, BVH_THREADS, bd[K - 1], unused_minlb);
            }
            if (st.alive) {
                const BvhLeafT<3>* __restrict__ lf = bv.leaves + st.idx;
#pragma unroll
                for (int t = 0; t < BVH_LEAF; t++) {
                    const float dx = px - lf->c[0][t], dy = py - lf->c[1][t], dz = pz - lf->c[2][t];
                    const float d = (dx * dx + dy * dy) + dz * dz;
                    const int j = lf->idx[t];
                    if (j >= 0 && ((d < bd[K - 1]) | ((d == bd[K - 1]) & (j < bj[K - 1])))) knn_insert<K>(bd, bj, d, j);
                }
                st.alive = false;
                trav_pop(st, bvh_lb16, tid, BVH_THREADS, bd[K - 1], unused_minlb);
            }
        }
        int cnt = 0;
        double m[3] = {0, 0, 0}, cxx = 0, cxy = 0, cxz = 0, cyy = 0, cyz = 0, czz = 0;
#pragma unroll
        for (int q = 0; q < K; q++) if (bd[q] < FLT_MAX) { const int j = bj[q]; m[0] += bv.tgt.c[0][j]; m[1] += bv.tgt.c[1][j]; m[2] += bv.tgt.c[2][j]; cnt++; }
        if (cnt >= 3) {
            m[0] /= cnt; m[1] /= cnt; m[2] /= cnt;
#pragma unroll
            for (int q = 0; q < K; q++) if (bd[q] < FLT_MAX) {
                const int j = bj[q];
                const double a = bv.tgt.c[0][j] - m[0], b = bv.tgt.c[1][j] - m[1], c = bv.tgt.c[2][j] - m[2];
                cxx += a * a; cxy += a * b; cxz += a * c; cyy += b * b; cyz += b * c; czz += c * c;
            }
            double A[9] = {cxx / cnt, cxy / cnt, cxz / cnt, cxy / cnt, cyy / cnt, cyz / cnt, cxz / cnt, cyz / cnt, czz / cnt}, V[9], ev[3];
            jacobi_eig_sym<3>(A, V, ev);
            int s0 = 0; if (ev[1] < ev[s0]) s0 = 1; if (ev[2] < ev[s0]) s0 = 2;
            double vx = V[0 * 3 + s0], vy = V[1 * 3 + s0], vz = V[2 * 3 + s0];
            const double len = sqrt(vx * vx + vy * vy + vz * vz);
            vx /= len; vy /= len; vz /= len;
            if ((vpx - px) * vx + (vpy - py) * vy + (vpz - pz) * vz < 0) { vx = -vx; vy = -vy; vz = -vz; }   // flipNormalTowardsViewpoint
            nx = (float)vx; ny = (float)vy; nz = (float)vz;
            const double tr = ev[0] + ev[1] + ev[2];
            curv = tr > 0 ? (float)(fabs(ev[s0]) / tr) : 0.f;
        }
    }
    nrm_out[(size_t)i * 3] = nx; nrm_out[(size_t)i * 3 + 1] = ny; nrm_out[(size_t)i * 3 + 2] = nz;
    if (curv_out) curv_out[i] = curv;
}

// out[t] = in[idx[t]] (one plane of a cloud) / out[t] = sel[order[t]]: the one-off physical permutation of the source into Morton order
__global__ void k_gather_f32(const float* __restrict__ in, const int* __restrict__ idx, int n, float* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x; if (t < n) out[t] = in[idx[t]];
}
__global__ void k_gather_u32(const uint32_t* __restrict__ in, const int* __restrict__ idx, int n, uint32_t* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x; if (t < n) out[t] = in[idx[t]];
}
__global__ void k_compose_idx(const int* __restrict__ sel, const int* __restrict__ order, int n, int* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x; if (t < n) out[t] = sel ? sel[order[t]] : order[t];
}
__global__ void k_iota(int* p, int n) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = i; }

// Morton key of the (untransformed) query points -> spatially coherent waves for k_knn_bvh.
__global__ void k_query_keys(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z, const int* __restrict__ sel, int n,
                             const unsigned int* __restrict__ box /* k_bbox of the cloud */, unsigned long long* __restrict__ keys, int* __restrict__ vals) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int i = sel ? sel[t] : t;
    const float a = x[i], b = y[i], c = z[i];
    unsigned long long key = ~0ull;
    if (finite3(a, b, c)) {
        float lo[3], sc[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            lo[k] = from_ordered_bits_u(box[k]);
            const float ext = from_ordered_bits_u(box[3 + k]) - lo[k];
            sc[k] = (ext > 0.f && isfinite(ext)) ? 2097151.f / ext : 0.f;
        }
        const float fa = fminf(fmaxf((a - lo[0]) * sc[0], 0.f), 2097151.f), fb = fminf(fmaxf((b - lo[1]) * sc[1], 0.f), 2097151.f), fc = fminf(fmaxf((c - lo[2]) * sc[2], 0.f), 2097151.f);
        key = spread21((unsigned int)fa) | (spread21((unsigned int)fb) << 1) | (spread21((unsigned int)fc) << 2);
    }
    keys[t] = key; vals[t] = t;
}
