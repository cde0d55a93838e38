// =====================================================================================
// icp_device.hpp -- hand-written HIP kernels of the ICP hot path for gfx950 (MI355X, wave64).
//
// Build contract: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (no fast-math): every fp32
// result is one IEEE rounding per operation, in the operation order of the CPU restatement
// (oracle/icp_oracle.cpp), so match indices / distances / weights are bit-identical to it.
// fp32 sqrt and divide are correctly rounded (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt).
//
// Files: dev_common.hpp, dev_knn_brute.hpp, dev_bvh.hpp, dev_normals.hpp, dev_projective.hpp, dev_post.hpp, dev_solve.hpp,
// dev_fused.hpp, dev_persist.hpp, dev_measures.hpp (included below, in this order, inside namespace icpdev).
//
// Kernel map (reference file:line relative to icp-variants/ of the reference):
//   k_deinterleave      AoS -> SoA upload conversion (+ colour features NearestNeighbor.h:212-221)
//   k_knn_brute<DIM>    transformPoints (utils.h:106-118) fused with exact 1-NN, first-minimum argmin
//                       (NearestNeighbor.h:81-97 semantics, squared L2 + squared threshold :181-185);
//                       DIM=6 adds rgb/255 (NearestNeighbor.h:209-303)
//   k_knn_finalize      merges target-split partial results (packed u64 atomicMin) into Match records
//   k_bvh_* / k_knn_bvh exact LBVH index (build once per pair = buildIndex, NearestNeighbor.h:122-141) and its query:
//                       bit-identical argmin to k_knn_brute at O(log M) per query
//   k_projective        NearestNeighborSearchProjective::queryMatches (NearestNeighbor.h:333-421)
//   k_post              transformNormals (utils.h:122-133) + applyWeights (weighting.h:39-99) +
//                       pruneCorrespondences (ICPOptimizer.h:157-174) + validity filter (:594-610) +
//                       normal-equation / moment accumulation (ICPOptimizer.h:676-751, ProcrustesAligner.h:43-55)
//   k_sym_accumulate    second pass of the symmetric objective with the means (ICPOptimizer.h:797-853)
//   k_icp_loop          the whole loop of one resolution level as ONE launch (point-to-plane, fused BVH matcher): resident waves keep their
//                       queries in registers, reducer blocks fold / solve / publish the pose through self-validating granules (dev_persist.hpp)
//   k_reduce_solve      fixed-order reduction of block partials + fp64 solve + pose composition
//                       (ICPOptimizer.h:614-620,753-781,855-897; ProcrustesAligner.h:56-66)
//   k_rmse_partial      ConvergenceMeasure::rmseAlignmentError (ConvergenceMeasure.h:50-66)
// =====================================================================================
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>
#include "../../include/icp_hip.h"

namespace icpdev {

#include "dev_common.hpp"
#include "dev_knn_brute.hpp"
#include "dev_bvh.hpp"
#include "dev_normals.hpp"
#include "dev_projective.hpp"
#include "dev_post.hpp"
#include "dev_solve.hpp"
#include "dev_fused.hpp"
#include "dev_persist.hpp"
#include "dev_measures.hpp"

}  // namespace icpdev
