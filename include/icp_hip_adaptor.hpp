// =====================================================================================
// icp_hip_adaptor.hpp -- C++14 host-side mirror of the reference's plugin surface on top of the C ABI
// (include/icp_hip.h).  Header-only.  Two classes:
//
//   NearestNeighborSearchHip : NearestNeighborSearch      (reference NearestNeighbor.h:12-36)
//       buildIndex / queryMatches / setCameraParams / setMatchingMaxDistance with the reference's
//       signatures; exact k-NN (3-D, 6-D colour) or projective search on the GPU.
//   HipLinearICPOptimizer    : ICPOptimizer               (reference ICPOptimizer.h:27-175, 489-663)
//       estimatePose(const PointCloud&, const PointCloud&, Matrix4f&, bool) running the whole
//       linear ICP loop on the device; fills TimeMeasure / ConvergenceMeasure like the reference.
//
// Two ways to compile it:
//   * inside the reference tree: #define ICP_HIP_WITH_REFERENCE_HEADERS and include this header AFTER
//     ICPOptimizer.h -- the classes then derive from the reference's own base classes and use Eigen types
//     (see INTEGRATION.md for the two-line switch in main.cpp / ICPOptimizer.h);
//   * standalone (this repository, no Eigen/FLANN/Ceres installed): the tiny interface types below
//     (namespace icp_hip_compat) re-declare the reference's containers with identical memory layout
//     (Vector3f = 3 packed floats, Vector4uc = 4 bytes, Matrix4f column-major) and identical method names.
// Errors: the reference prints and returns {} (NearestNeighbor.h:144-152,335-349) or spins in ASSERT
// (Eigen.h:9); here a failing call prints the library's message to std::cerr, returns {} / leaves the pose
// untouched, and lastStatus() reports the icp_status code.
// =====================================================================================
#ifndef ICP_HIP_ADAPTOR_HPP
#define ICP_HIP_ADAPTOR_HPP

#include "icp_hip.h"
#include <cstdint>
#include <cstring>
#include <iostream>
#include <memory>
#include <random>
#include <stdexcept>
#include <vector>

#ifndef ICP_HIP_WITH_REFERENCE_HEADERS
namespace icp_hip_compat {

struct Vector3f {                                   // layout == Eigen::Vector3f (12 B, no padding)
    float v[3];
    Vector3f() : v{0, 0, 0} {}
    Vector3f(float x, float y, float z) : v{x, y, z} {}
    float& operator[](int i) { return v[i]; }
    const float& operator[](int i) const { return v[i]; }
    float x() const { return v[0]; } float y() const { return v[1]; } float z() const { return v[2]; }
    const float* data() const { return v; }
};
struct Vector4uc {                                  // layout == Eigen::Matrix<unsigned char,4,1> (Eigen.h:36)
    unsigned char v[4];
    Vector4uc() : v{0, 0, 0, 0} {}
    Vector4uc(unsigned char r, unsigned char g, unsigned char b, unsigned char a) : v{r, g, b, a} {}
    unsigned char& operator[](int i) { return v[i]; }
    const unsigned char& operator[](int i) const { return v[i]; }
};
template <int N> struct MatrixNf {                  // column-major like Eigen's default
    float m[N * N];
    MatrixNf() { std::memset(m, 0, sizeof(m)); }
    static MatrixNf Identity() { MatrixNf r; for (int i = 0; i < N; i++) r.m[i * N + i] = 1.f; return r; }
    float& operator()(int r, int c) { return m[c * N + r]; }
    const float& operator()(int r, int c) const { return m[c * N + r]; }
    float* data() { return m; }
    const float* data() const { return m; }
};
typedef MatrixNf<3> Matrix3f;
typedef MatrixNf<4> Matrix4f;

struct Match { int idx; float weight; };            // NearestNeighbor.h:7-10

class NearestNeighborSearch {                       // NearestNeighbor.h:12-36
public:
    virtual ~NearestNeighborSearch() {}
    virtual void setMatchingMaxDistance(float maxDistance) { m_maxDistance = maxDistance; }   // squared
    virtual void buildIndex(const std::vector<Vector3f>& targetPoints) = 0;
    virtual std::vector<Match> queryMatches(const std::vector<Vector3f>& transformedPoints) = 0;
    virtual void buildIndex(const std::vector<Vector3f>& targetPoints, const std::vector<Vector4uc>& targetColors) = 0;
    virtual std::vector<Match> queryMatches(const std::vector<Vector3f>& transformedPoints, const std::vector<Vector4uc>& transformedColors) = 0;
    virtual void setCameraParams(const Matrix3f& depthIntrinsics, const unsigned width, const unsigned height) = 0;
protected:
    float m_maxDistance;
    NearestNeighborSearch() : m_maxDistance{0.005f} {}                                           // MAX_DISTANCE, NearestNeighbor.h:5
};

class PointCloud {                                  // the accessor surface of PointCloud.h:284-306
public:
    std::vector<Vector3f>& getPoints() { return m_points; }
    const std::vector<Vector3f>& getPoints() const { return m_points; }
    std::vector<Vector3f>& getNormals() { return m_normals; }
    const std::vector<Vector3f>& getNormals() const { return m_normals; }
    std::vector<Vector4uc>& getColors() { return m_colors; }
    const std::vector<Vector4uc>& getColors() const { return m_colors; }
private:
    std::vector<Vector3f> m_points, m_normals;
    std::vector<Vector4uc> m_colors;
};

struct TimeMeasure {                                // TimeMeasure.h:20-27
    double selectionTime = 0, matchingTime = 0, weighingTime = 0, rejectionTime = 0, solverTime = 0, convergenceTime = 0;
    unsigned int* nIterations = nullptr;
};
struct ConvergenceMeasure {                         // ConvergenceMeasure.h:69-78 (recording side only)
    std::vector<Matrix4f> recordedPoses;
    void recordAlignmentError(const Matrix4f& pose) { recordedPoses.push_back(pose); }
};

enum selection_methods { SELECT_ALL = 0, RANDOM_SAMPLING };                                      // selection.h:9
enum weighting_methods { CONSTANT_WEIGHTING = 0, DISTANCES_WEIGHTING, NORMALS_WEIGHTING, COLORS_WEIGHTING };   // weighting.h:8

class ICPOptimizer {                                // ICPOptimizer.h:27-175 (setter surface + protected state)
public:
    ICPOptimizer() : metric{0}, colorICP{false}, multiResolutionICP{false}, selectionMethod{0}, proba{1.0}, rejectionMethod{1},
                     weightingMethod{0}, matchingMethod{0}, m_nIterations{20}, m_timeMeasure{nullptr}, m_convergenceMeasure{nullptr},
                     maxDistance{0.0003f} {}
    virtual ~ICPOptimizer() = default;
    virtual void setMatchingMaxDistance(float d) { maxDistance = d; }
    void setMetric(unsigned int m) { metric = m; }
    void enableMultiResolution(bool e) { multiResolutionICP = e; }
    void enableColorICP(bool e) { colorICP = e; }
    void setSelectionMethod(unsigned int s, double p = 1.0) { selectionMethod = s; proba = p; }
    void setRejectionMethod(unsigned int r) { rejectionMethod = r; }
    void setWeightingMethod(unsigned int w) { weightingMethod = w; }
    virtual void setMatchingMethod(unsigned int m) { matchingMethod = m; }
    virtual void setCameraParamsMatchingMethod(const Matrix3f& K, const unsigned w, const unsigned h) = 0;
    void setNbOfIterations(unsigned n) { m_nIterations = n; }
    void setTimeMeasure(TimeMeasure& t) { t.nIterations = &m_nIterations; m_timeMeasure = &t; }
    void setConvergenceMeasure(ConvergenceMeasure& c) { m_convergenceMeasure = &c; }
    virtual void estimatePose(const PointCloud& source, const PointCloud& target, Matrix4f& initialPose, bool calculateRMSE = true) = 0;
protected:
    unsigned int metric; bool colorICP; bool multiResolutionICP; unsigned int selectionMethod; double proba;
    unsigned int rejectionMethod; unsigned int weightingMethod; unsigned int matchingMethod; unsigned m_nIterations;
    TimeMeasure* m_timeMeasure; ConvergenceMeasure* m_convergenceMeasure; float maxDistance;
};

}  // namespace icp_hip_compat
using namespace icp_hip_compat;
#endif  // !ICP_HIP_WITH_REFERENCE_HEADERS

namespace icp_hip_detail {
struct CtxDeleter { void operator()(icp_ctx* c) const { if (c) icp_ctx_destroy(c); } };
typedef std::unique_ptr<icp_ctx, CtxDeleter> CtxPtr;
inline CtxPtr make_ctx(int device) {
    icp_ctx* c = nullptr;
    const int rc = icp_ctx_create(device, &c);
    if (rc != ICP_OK) throw std::runtime_error("icp_ctx_create failed (no usable HIP device): status " + std::to_string(rc));
    return CtxPtr(c);
}
static_assert(sizeof(Vector3f) == 12, "Vector3f must be 3 packed floats");
static_assert(sizeof(Vector4uc) == 4, "Vector4uc must be 4 bytes");
static_assert(sizeof(Match) == sizeof(icp_match_t), "Match layout");
}  // namespace icp_hip_detail

// -------------------------------------------------------------------------------------------------
class NearestNeighborSearchHip : public NearestNeighborSearch {
public:
    explicit NearestNeighborSearchHip(unsigned matchingMethod = 0, int device = 0)
        : NearestNeighborSearch(), m_ctx(icp_hip_detail::make_ctx(device)), m_status(ICP_OK), m_built(false), m_withColors(false) {
        icp_params_default(&m_prm);
        m_prm.matching = (int32_t)matchingMethod; m_prm.knn_backend = ICP_KNN_LBVH; m_prm.max_distance = m_maxDistance;
    }
    void setMatchingMaxDistance(float maxDistance) override { m_maxDistance = maxDistance; m_prm.max_distance = maxDistance; }
    void buildIndex(const std::vector<Vector3f>& targetPoints) override { build(targetPoints, nullptr); }
    void buildIndex(const std::vector<Vector3f>& targetPoints, const std::vector<Vector4uc>& targetColors) override {
        if (targetColors.size() != targetPoints.size()) { fail(ICP_ERR_INVALID_ARG, "colours/points size mismatch"); return; }
        build(targetPoints, reinterpret_cast<const uint8_t*>(targetColors.data()));
    }
    std::vector<Match> queryMatches(const std::vector<Vector3f>& transformedPoints) override {
        if (m_built && m_withColors && m_prm.matching == ICP_MATCH_KNN) { fail(ICP_ERR_COLOR_MISMATCH, "Please call queryMatches with colors."); return {}; }   // NearestNeighbor.h:149-152
        return query(transformedPoints, nullptr);
    }
    std::vector<Match> queryMatches(const std::vector<Vector3f>& transformedPoints, const std::vector<Vector4uc>& transformedColors) override {
        if (m_built && !m_withColors) { fail(ICP_ERR_COLOR_MISMATCH, "Please call queryMatches without colors."); return {}; }                                  // NearestNeighbor.h:240-243
        if (transformedColors.size() != transformedPoints.size()) { fail(ICP_ERR_INVALID_ARG, "colours/points size mismatch"); return {}; }
        return query(transformedPoints, reinterpret_cast<const uint8_t*>(transformedColors.data()));
    }
    void setCameraParams(const Matrix3f& K, const unsigned width, const unsigned height) override {
        m_prm.fx = K(0, 0); m_prm.fy = K(1, 1); m_prm.cx = K(0, 2); m_prm.cy = K(1, 2); m_prm.width = (int32_t)width; m_prm.height = (int32_t)height;
    }
    int lastStatus() const { return m_status; }
    const icp_params& params() const { return m_prm; }
private:
    void build(const std::vector<Vector3f>& pts, const uint8_t* rgba) {
        m_status = icp_set_params(m_ctx.get(), &m_prm);
        if (m_status == ICP_OK) m_status = icp_set_target(m_ctx.get(), reinterpret_cast<const float*>(pts.data()), nullptr, rgba, (int32_t)pts.size());
        if (m_status != ICP_OK) { report(); return; }
        m_built = true; m_withColors = rgba != nullptr;
    }
    std::vector<Match> query(const std::vector<Vector3f>& q, const uint8_t* rgba) {
        std::vector<Match> out(q.size());
        if (q.empty()) return out;
        m_status = icp_set_params(m_ctx.get(), &m_prm);
        if (m_status == ICP_OK) m_status = icp_query_matches(m_ctx.get(), reinterpret_cast<const float*>(q.data()), rgba, (int32_t)q.size(), reinterpret_cast<icp_match_t*>(out.data()));
        if (m_status != ICP_OK) { report(); return {}; }
        return out;
    }
    void fail(int code, const char* msg) { m_status = code; std::cout << msg << std::endl; }
    void report() { std::cout << icp_last_error(m_ctx.get()) << std::endl; }
    icp_hip_detail::CtxPtr m_ctx; icp_params m_prm; int m_status; bool m_built, m_withColors;
};

// -------------------------------------------------------------------------------------------------
class HipLinearICPOptimizer : public ICPOptimizer {
public:
    explicit HipLinearICPOptimizer(int device = 0) : ICPOptimizer(), m_ctx(icp_hip_detail::make_ctx(device)), m_status(ICP_OK), m_hasCamera(false) {
        std::memset(m_cam, 0, sizeof(m_cam)); m_camW = m_camH = 0;
    }
    // setCameraParamsMatchingMethod, ICPOptimizer.h:80-82
    void setCameraParamsMatchingMethod(const Matrix3f& K, const unsigned width, const unsigned height)
#ifndef ICP_HIP_WITH_REFERENCE_HEADERS
        override
#endif
    {
        m_cam[0] = K(0, 0); m_cam[1] = K(1, 1); m_cam[2] = K(0, 2); m_cam[3] = K(1, 2); m_camW = width; m_camH = height; m_hasCamera = true;
    }
    // LinearICPOptimizer::estimatePose, ICPOptimizer.h:493-663
    void estimatePose(const PointCloud& source, const PointCloud& target, Matrix4f& initialPose, bool calculateRMSE = true) override {
        icp_params p; icp_params_default(&p);
        // RANDOM_SAMPLING: the reference seeds a fresh mt19937 from random_device per estimatePose (selection.h:76-79,
        // ICPOptimizer.h:519-525); same here unless setSelectionSeed() pinned a seed for reproducible runs.
        p.selection = (int32_t)selectionMethod; p.selection_proba = (float)proba;
        p.selection_seed = m_seedPinned ? m_seed : (uint32_t)std::random_device{}();
        p.metric = (int32_t)metric; p.matching = (int32_t)matchingMethod; p.weighting = (int32_t)weightingMethod; p.rejection = (int32_t)rejectionMethod;
        p.color_icp = colorICP ? 1 : 0; p.multires = multiResolutionICP ? 1 : 0; p.n_iterations = (int32_t)m_nIterations; p.max_distance = maxDistance;
        p.knn_backend = ICP_KNN_LBVH;
        if (m_hasCamera) { p.fx = m_cam[0]; p.fy = m_cam[1]; p.cx = m_cam[2]; p.cy = m_cam[3]; p.width = (int32_t)m_camW; p.height = (int32_t)m_camH; }
#ifdef ICP_HIP_WITH_REFERENCE_HEADERS
        // In the reference tree the base class forwards setCameraParamsMatchingMethod (non-virtual, ICPOptimizer.h:80-82) to
        // m_nearestNeighborSearch; with the INTEGRATION.md switch that object is a NearestNeighborSearchHip holding them.
        if (auto* nn = dynamic_cast<NearestNeighborSearchHip*>(m_nearestNeighborSearch.get())) {
            const icp_params& q = nn->params();
            if (q.width > 0 && q.height > 0) { p.fx = q.fx; p.fy = q.fy; p.cx = q.cx; p.cy = q.cy; p.width = q.width; p.height = q.height; }
        }
#endif
        icp_ctx* c = m_ctx.get();
        const auto& sp = source.getPoints(); const auto& tp = target.getPoints();
        const uint8_t* sc = source.getColors().size() == sp.size() ? reinterpret_cast<const uint8_t*>(source.getColors().data()) : nullptr;
        const uint8_t* tc = target.getColors().size() == tp.size() ? reinterpret_cast<const uint8_t*>(target.getColors().data()) : nullptr;
        m_status = icp_set_params(c, &p);
        if (m_status == ICP_OK) m_status = icp_set_target(c, reinterpret_cast<const float*>(tp.data()), reinterpret_cast<const float*>(target.getNormals().data()), tc, (int32_t)tp.size());   // buildIndex :532-535
        if (m_status == ICP_OK) m_status = icp_set_source(c, reinterpret_cast<const float*>(sp.data()), reinterpret_cast<const float*>(source.getNormals().data()), sc, (int32_t)sp.size());
        if (m_status != ICP_OK) { std::cerr << icp_last_error(c) << std::endl; return; }
        int32_t cap = 0;
        icp_schedule(&p, (int32_t)sp.size(), nullptr, 0, &cap);
        std::vector<icp_iter_stats> stats((size_t)(cap > 0 ? cap : 1));
        int32_t n = 0;
        float pose[16]; std::memcpy(pose, initialPose.data(), sizeof(pose));
        m_status = icp_run(c, pose, stats.data(), (int32_t)stats.size(), &n);
        if (m_status != ICP_OK) std::cerr << icp_last_error(c) << std::endl;
        if (m_status != ICP_OK && m_status != ICP_ERR_NO_CORRESPONDENCES) return;
        std::memcpy(initialPose.data(), pose, sizeof(pose));                                  // :659
        m_iterations.assign(stats.begin(), stats.begin() + n);
        if (m_timeMeasure) {                                                                  // :551-624,662
            icp_timing t; icp_get_timing(c, &t);
            m_timeMeasure->matchingTime += t.match_ms * 1e-3;
            m_timeMeasure->weighingTime += t.weight_reject_build_ms * 1e-3;                   // weighting + rejection + system build are one kernel
            m_timeMeasure->solverTime += t.solve_ms * 1e-3;
            m_timeMeasure->convergenceTime += t.total_ms * 1e-3;
        }
        if (calculateRMSE && m_convergenceMeasure) {                                          // :629-631, one record per iteration
            for (int32_t i = 0; i < n; i++) { Matrix4f P; std::memcpy(P.data(), stats[(size_t)i].pose, sizeof(pose)); m_convergenceMeasure->recordAlignmentError(P); }
        }
    }
    void setSelectionSeed(uint32_t seed) { m_seed = seed; m_seedPinned = true; }
    int lastStatus() const { return m_status; }
    const std::vector<icp_iter_stats>& iterations() const { return m_iterations; }
private:
    icp_hip_detail::CtxPtr m_ctx; int m_status; bool m_hasCamera; float m_cam[4]; unsigned m_camW, m_camH;
    uint32_t m_seed = 0; bool m_seedPinned = false;
    std::vector<icp_iter_stats> m_iterations;
};

#endif  // ICP_HIP_ADAPTOR_HPP
