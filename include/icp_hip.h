/* =====================================================================================
 * icp_hip.h -- C ABI of the MI355X-native ICP hot path (libicp_hip.so)
 *
 * Drop-in boundary for the linear-ICP inner loop of PetropoulakisPanagiotis/ICP-Variants.
 * The reference has no FFI; its "plugin surface" is three C++ interfaces resolved at compile
 * time.  Each entry point below names the reference interface it replaces (file:line relative
 * to icp-variants/ in the reference).  The C++14 adaptor classes that keep the reference's
 * method names on top of this ABI are in include/icp_hip_adaptor.hpp; INTEGRATION.md shows the
 * few lines a maintainer of the reference adds to switch over.
 *
 * Conventions (identical to the reference's containers, so no conversion is needed):
 *   points / normals : N x 3 fp32, row-major, 12 B per element  == std::vector<Eigen::Vector3f>::data()
 *   colours          : N x 4 uint8 RGBA                          == std::vector<Vector4uc>::data()   (Eigen.h:36)
 *   pose             : 16 fp32, COLUMN-major 4x4                 == Eigen::Matrix4f::data()
 *   intrinsics       : fx, fy, cx, cy of the depth camera        (Matrix3f K: K(0,0),K(1,1),K(0,2),K(1,2))
 *   max_distance     : SQUARED metres                            (NearestNeighbor.h:16-18, ICPOptimizer.h:154)
 * All functions return ICP_OK (0) or an error code; nothing ever spins (the reference's ASSERT
 * hangs in while(1), Eigen.h:9).  Host pointers only; the context owns every device buffer.
 * One host thread per context; contexts are independent (one per GPU / HIP stream).
 * ===================================================================================== */
#ifndef ICP_HIP_H
#define ICP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct icp_ctx icp_ctx;

/* struct Match, NearestNeighbor.h:7-10.  idx = -1 => no match. */
typedef struct icp_match_t { int32_t idx; float weight; } icp_match_t;

enum icp_status {
    ICP_OK = 0,
    ICP_ERR_INVALID_ARG = 1,
    ICP_ERR_HIP = 2,                 /* a HIP runtime call failed; see icp_last_error() */
    ICP_ERR_NO_TARGET = 3,           /* "index needs to be build before querying" NearestNeighbor.h:144-147,335-338 */
    ICP_ERR_NO_SOURCE = 4,
    ICP_ERR_NO_CAMERA = 5,           /* "Set camera params before querying"        NearestNeighbor.h:341-344 */
    ICP_ERR_TARGET_SIZE = 6,         /* "Invalid size of target points"            NearestNeighbor.h:346-349 */
    ICP_ERR_COLOR_MISMATCH = 7,      /* 3-D index queried with colours or vice versa NearestNeighbor.h:149-152,240-243 */
    ICP_ERR_NO_CORRESPONDENCES = 8,  /* reference ASSERT (ICPOptimizer.h:668,680,788) -- reported, never a hang */
    ICP_ERR_NO_DEVICE = 9,
    ICP_ERR_COMM = 10                /* RCCL missing or an RCCL call failed; see icp_comm_last_error() */
};

/* enum values mirror the reference */
enum { ICP_METRIC_POINT_TO_POINT = 0, ICP_METRIC_POINT_TO_PLANE = 1, ICP_METRIC_SYMMETRIC = 2 };   /* ICPOptimizer.h:46-48,131-136 */
enum { ICP_MATCH_KNN = 0, ICP_MATCH_PROJECTIVE = 1 };                                               /* ICPOptimizer.h:71-78 */
enum { ICP_WEIGHT_CONSTANT = 0, ICP_WEIGHT_DISTANCES = 1, ICP_WEIGHT_NORMALS = 2, ICP_WEIGHT_COLORS = 3 };   /* weighting.h:8 */
enum { ICP_KNN_BRUTE_FORCE = 0, ICP_KNN_LBVH = 1 };   /* both exact: identical (d2, lowest-index) argmin, bit for bit */

/* The setter surface of ICPOptimizer (ICPOptimizer.h:41-95) as one POD. */
typedef struct icp_params {
    int32_t metric;          /* setMetric                     default 0           */
    int32_t matching;        /* setMatchingMethod             default 0 (k-NN)    */
    int32_t weighting;       /* setWeightingMethod            default 0           */
    int32_t rejection;       /* setRejectionMethod            default 1 (60 deg)  */
    int32_t color_icp;       /* enableColorICP                default 0           */
    int32_t multires;        /* enableMultiResolution         default 0           */
    int32_t n_iterations;    /* setNbOfIterations             default 20          */
    float   max_distance;    /* setMatchingMaxDistance        default 0.0003f, squared metres */
    float   fx, fy, cx, cy;  /* setCameraParamsMatchingMethod (ICPOptimizer.h:80-82) */
    int32_t width, height;
    int32_t knn_backend;     /* ICP_KNN_* (extension; the reference's own index is an approximate, randomised FLANN kd-tree) */
    int32_t selection;       /* setSelectionMethod: 0 SELECT_ALL, 1 RANDOM_SAMPLING (selection.h:9)                          */
    float   selection_proba; /* Bernoulli probability per point and per iteration (selection.h:88-106)                     */
    uint32_t selection_seed; /* the reference seeds std::mt19937 from random_device (selection.h:76-79: not reproducible);
                                here a counter-based hash of (seed, iteration, point index) decides -- see icp_select_hash */
    int32_t knn_incremental; /* 1 (default): BVH k-NN verifies the previous neighbour with an exact distance bound and skips the
                                tree walk when it provably cannot change (bit-identical results); 0: always walk the tree */
    int32_t record_rmse;     /* bit 0: per-iteration RMSE against the convergence reference (ConvergenceMeasure.h:50-66);
                                bit 1: also the benchmark error (m_runBenchmark, ConvergenceMeasure.h:74-78,104-151) */
} icp_params;

/* Per-iteration record (what the reference prints / records each iteration, ICPOptimizer.h:541-631). */
typedef struct icp_iter_stats {
    int32_t n_src;           /* source points matched this iteration (multires level size) */
    int32_t n_valid;         /* correspondences that entered the solve (ICPOptimizer.h:594-610) */
    float   pose[16];        /* estimatedPose after the iteration, column-major */
    float   rmse;            /* RMSE vs convergence reference, or -1 */
    float   benchmark_error; /* Fontana-style benchmark error (ConvergenceMeasure.h:104-151) when record_rmse & 2, or -1 */
    int32_t status;          /* ICP_OK or ICP_ERR_NO_CORRESPONDENCES for this iteration */
} icp_iter_stats;

/* Stage times of the last icp_run, device milliseconds from HIP events: the TimeMeasure breakdown
 * (TimeMeasure.h:20-26).  Weighting, rejection and system build are one fused kernel here. */
typedef struct icp_timing {
    double match_ms;         /* matchingTime                                   */
    double weight_reject_build_ms;   /* weighingTime + rejectionTime + system build */
    double solve_ms;         /* reduction + linear solve + pose composition    */
    double total_ms;         /* convergenceTime                                 */
    int32_t iterations;
    int32_t sampled_iterations;   /* iterations that were bracketed by events (== iterations unless icp_set_stage_timing(N > 1)) */
} icp_timing;

/* -------- context -------- */
int icp_ctx_create(int device, icp_ctx** out);
/* Same, but all work is enqueued on an existing hipStream_t (e.g. torch.cuda.current_stream().cuda_stream). */
int icp_ctx_create_on_stream(int device, void* hip_stream, icp_ctx** out);
int icp_ctx_destroy(icp_ctx* ctx);
const char* icp_last_error(const icp_ctx* ctx);

/* -------- configuration: ICPOptimizer setters, ICPOptimizer.h:41-95 -------- */
int icp_params_default(icp_params* p);                       /* ICPOptimizer ctor defaults, ICPOptimizer.h:29-37 */
int icp_set_params(icp_ctx* ctx, const icp_params* p);
int icp_get_params(const icp_ctx* ctx, icp_params* p);

/* -------- NearestNeighborSearch::buildIndex, NearestNeighbor.h:24,27 (122-141, 209-232, 324-331) --------
 * Uploads the target once per pair (AoS -> SoA on the device).  normals may be NULL when only
 * icp_query_matches is used; rgba may be NULL unless color_icp / colour weighting is on. */
int icp_set_target(icp_ctx* ctx, const float* xyz, const float* normals, const uint8_t* rgba, int32_t n);
/* Source cloud of estimatePose (ICPOptimizer.h:140); resident across iterations. */
int icp_set_source(icp_ctx* ctx, const float* xyz, const float* normals, const uint8_t* rgba, int32_t n);

/* -------- NearestNeighborSearch::queryMatches, NearestNeighbor.h:25,28 --------
 * transformed_xyz are ALREADY transformed query points (exactly the argument the reference passes);
 * rgba != NULL selects the 6-D colour search (the target must have been set with colours). */
int icp_query_matches(icp_ctx* ctx, const float* transformed_xyz, const uint8_t* rgba, int32_t n, icp_match_t* out);

/* -------- single stages on the resident source (parity-test entry points) --------
 * icp_match      : transformPoints (utils.h:106-118) fused with queryMatches; d2_out (optional) = winning
 *                  squared distance (FLT_MAX when no candidate).
 * icp_correspond : + applyWeights (weighting.h:39-99) + pruneCorrespondences (ICPOptimizer.h:157-174) +
 *                  validity filter (ICPOptimizer.h:594-610).  sums_out (optional, 64 doubles) receives the
 *                  reduced accumulators the solver consumes (layout in DESIGN.md), n_valid_out the count. */
int icp_match(icp_ctx* ctx, const float pose[16], icp_match_t* out, float* d2_out);
/* The search exactly as the loop of icp_run runs it (parity-test entry point for the seeded / incremental path, NearestNeighbor.h:81-97
 * semantics): the fused matcher of the k-NN BVH backend is launched once per pose of poses[0 .. n_poses) (column-major, 16 floats
 * each) on the Morton-sorted resident source -- the first launch unseeded, every further one seeded with the previous launch's
 * neighbours and search state (verify-and-skip tiers, shared walks, spread start: whatever knn_incremental and the build enable),
 * i.e. launch j is iteration j of a run whose poses are dictated by the caller.  out / d2_out (optional) receive the LAST launch's
 * records in source order: the Match after weighting + rejection (= what icp_correspond returns; with rejection 0 and constant
 * weights the raw {idx, 1} / {-1, 0} of queryMatches) and the winning squared distance (FLT_MAX when there was no candidate).
 * Needs matching = k-NN, knn_backend = LBVH, a metric other than symmetric, normals on both clouds; ICP_ERR_INVALID_ARG otherwise. */
int icp_match_seeded(icp_ctx* ctx, const float* poses, int32_t n_poses, icp_match_t* out, float* d2_out);
int icp_correspond(icp_ctx* ctx, const float pose[16], icp_match_t* out, double* sums_out, int32_t* n_valid_out);

/* -------- one iteration / the whole loop: LinearICPOptimizer::estimatePose, ICPOptimizer.h:493-663 --------
 * pose_inout is the caller-owned in/out initialPose (ICPOptimizer.h:140,538,659).
 * icp_iterate runs stages 2-5 once on the full-resolution source (no multires bookkeeping).
 * icp_run runs n_iterations (or the multi-resolution schedule, ICPOptimizer.h:503-525,634-655) without
 * any host round trip inside the loop; stats (optional) receives up to max_stats records. */
int icp_iterate(icp_ctx* ctx, float pose_inout[16], icp_iter_stats* stats);
int icp_run(icp_ctx* ctx, float pose_inout[16], icp_iter_stats* stats, int32_t max_stats, int32_t* n_iterations_run);
int icp_get_timing(const icp_ctx* ctx, icp_timing* out);
/* The same breakdown iteration by iteration for the last icp_run (what TimeMeasure accumulates, before the sum): entry i is the
 * device time of iteration i in milliseconds, or -1 when that iteration was not bracketed (icp_set_stage_timing(N != 1)).  Any of
 * the three arrays may be NULL; *count_out = iterations of the last run. */
int icp_get_iteration_times(const icp_ctx* ctx, float* match_ms, float* weight_reject_build_ms, float* solve_ms, int32_t max_out, int32_t* count_out);
/* How icp_run fills the stage breakdown: 0 = whole-run time only, 1 = every iteration bracketed by HIP events (default;
 * what TimeMeasure does), N > 1 = every Nth iteration (rotating offset), stage sums scaled by iterations / sampled.
 * A HIP event costs about 4 us of stream time, i.e. mode 1 is ~10 % of a 0.07 ms iteration. */
int icp_set_stage_timing(icp_ctx* ctx, int32_t every_nth);
/* The iteration schedule icp_run will execute (pure host logic, no device needed): one decimation factor per
 * iteration, 0 = full cloud without selection.  ICPOptimizer.h:503-516,540,634-655 / PointCloud.h:325-343. */
int icp_schedule(const icp_params* p, int32_t n_src, int32_t* factors_out, int32_t max_out, int32_t* count_out);

/* -------- ConvergenceMeasure (ConvergenceMeasure.h:30-66): known-correspondence RMSE --------
 * src_xyz[i] (moved by the estimated pose) is compared with ref_xyz[i]. */
int icp_set_convergence_reference(icp_ctx* ctx, const float* src_xyz, const float* ref_xyz, int32_t n);
int icp_rmse(icp_ctx* ctx, const float pose[16], float* rmse_out);
/* ConvergenceMeasure::benchmarkError (ConvergenceMeasure.h:104-151): mean |T s_i - r_i| / |T s_i - centroid(T s)|. */
int icp_benchmark_error(icp_ctx* ctx, const float pose[16], float* error_out);

/* -------- utils.h:106-133 on the device (used by the adaptor for transformPoints/Normals) -------- */
int icp_transform_points(icp_ctx* ctx, const float* xyz, int32_t n, const float pose[16], float* out);
int icp_transform_normals(icp_ctx* ctx, const float* normals, int32_t n, const float pose[16], float* out);

/* -------- pre-processing in front of the loop --------
 * PointCloud(float* depthMap, BYTE* colorFrame, depthIntrinsics, depthExtrinsics, width, height, keepOriginalSize, ...)
 * (PointCloud.h:78-165): back-projects a depth image (MINF = no measurement, VirtualSensor.h:119-124) and computes
 * central-difference normals on the device.  Outputs are organised, width*height entries, invalid entries MINF
 * (= keepOriginalSize true, downsampleFactor 1); valid_out (optional) marks the entries the keepOriginalSize = false
 * filter keeps.  rgbx / rgba_out may be NULL.  fix_color_index = 0 reproduces the reference's colour indexing
 * (bytes i..i+3 of the RGBX frame for pixel i, PointCloud.h:156-157), 1 reads pixel i's own 4 bytes. */
int icp_backproject_depth(icp_ctx* ctx, const float* depth, const uint8_t* rgbx, float fx, float fy, float cx, float cy,
                          const float extrinsics[16], int32_t width, int32_t height, float max_distance, int32_t fix_color_index,
                          float* xyz_out, float* normals_out, uint8_t* rgba_out, uint8_t* valid_out);

/* PointCloud(pcl::PointCloud<PointXYZ>::Ptr) (PointCloud.h:41-76): normals of an unorganised scan from its k nearest
 * neighbours (pcl::NormalEstimation, setKSearch(5), viewpoint (0,0,0)): exact k-NN on the device, fp64 PCA, normal flipped
 * towards the viewpoint.  k in {3..8}.  Non-finite points get NaN normals.  curvature_out may be NULL.  Uses scratch
 * buffers of the context only (target / source stay untouched). */
int icp_estimate_normals(icp_ctx* ctx, const float* xyz, int32_t n, int32_t k, const float viewpoint[3], float* normals_out, float* curvature_out);

/* The selection predicate of RANDOM_SAMPLING: point `index` is kept in resample number `iteration` iff the returned
 * 32-bit hash is < proba * 2^32.  Exposed so host code (and the test oracle) can reproduce the device's choice exactly. */
uint32_t icp_select_hash(uint32_t seed, uint32_t iteration, uint32_t index);

/* -------- batches of independent scan pairs: the loop over ETH indices, main.cpp:411-498 / experiment.cpp:319-396 --------
 * The reference aligns the pairs one after the other and carries no state between them, so a batch shards with no
 * data-path exchange: pair p belongs to rank p % n_ranks (icp_pair_owner), and the only collective is ONE gather of the
 * 16-float poses at the end of the batch. */
typedef struct icp_pair {
    const float* src_xyz; const float* src_normals; const uint8_t* src_rgba; int32_t n_src;   /* input.source, borrowed for the call */
    const float* tgt_xyz; const float* tgt_normals; const uint8_t* tgt_rgba; int32_t n_tgt;   /* input.target */
    float initial_pose[16];                                                                  /* estimatedPose on entry (main.cpp:416) */
} icp_pair;

/* Aligns pairs[0..n_pairs) with the contexts ctxs[0..n_ctx) of ONE device: one host thread per context (a context is
 * single-threaded, contexts are independent), each taking the next pair not yet started -- while one pair iterates, the
 * uploads, index builds and iterations of the others overlap on their own HIP streams.  Every context runs with its own
 * icp_params (set them beforehand).  poses_out: n_pairs x 16 floats, column-major, in pair order; status_out (optional):
 * per-pair ICP_OK / error code.  Returns ICP_OK or the first error in pair order. */
int icp_batch_run(icp_ctx* const* ctxs, int32_t n_ctx, const icp_pair* pairs, int32_t n_pairs, float* poses_out, int32_t* status_out);

/* Round-robin ownership of the batch: rank of pair p, and the number of pairs a rank owns. */
int32_t icp_pair_owner(int32_t pair, int32_t n_ranks);
int32_t icp_pairs_of_rank(int32_t n_pairs, int32_t rank, int32_t n_ranks);

/* Pose gather across the GPUs of a node: one RCCL communicator (one rank per process / GPU, xGMI), ONE ncclAllGather of
 * ceil(n_pairs / n_ranks) x 16 floats per rank and batch.  RCCL (librccl.so.1) is loaded on first use, the library has no
 * link-time dependency on it.  The unique id is created on one rank and shipped to the others by the host application
 * (MPI, a file, torch.distributed, ...), exactly like ncclGetUniqueId / ncclCommInitRank. */
typedef struct icp_comm icp_comm;
enum { ICP_COMM_ID_BYTES = 128 };
int icp_comm_unique_id(uint8_t id_out[ICP_COMM_ID_BYTES]);
int icp_comm_create(int device, int32_t n_ranks, int32_t rank, const uint8_t id[ICP_COMM_ID_BYTES], icp_comm** out);
int icp_comm_destroy(icp_comm* comm);
/* local_poses: this rank's poses in the order of its pairs (rank, rank + n_ranks, ...), n_local = icp_pairs_of_rank(...);
 * all_poses_out: n_pairs x 16 floats in pair order, identical on every rank. */
int icp_gather_poses(icp_comm* comm, const float* local_poses, int32_t n_local, int32_t n_pairs, float* all_poses_out);
const char* icp_comm_last_error(void);

/* Library identification: returns e.g. "icp_hip gfx950 <build id>". */
const char* icp_version(void);

#ifdef __cplusplus
}
#endif
#endif /* ICP_HIP_H */
