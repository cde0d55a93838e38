// C++14 driver for the adaptor classes (include/icp_hip_adaptor.hpp), written like the reference's
// alignBunnyWithICP (main.cpp:43-181): build the optimizer, set the canonical bunny parameters, call
// estimatePose, print the pose.  Reads the clouds from a raw dump written by tests/test_gpu_adaptor.py.
//   usage: bunny_adaptor <dump.bin> <metric> <multires>
#include "icp_hip_adaptor.hpp"
#include <cstdio>
#include <cstdlib>

static bool read_cloud(FILE* f, PointCloud& pc) {
    int32_t n = 0;
    if (fread(&n, 4, 1, f) != 1) return false;
    pc.getPoints().resize(n); pc.getNormals().resize(n); pc.getColors().resize(n);
    if (fread(pc.getPoints().data(), 12, n, f) != (size_t)n) return false;
    if (fread(pc.getNormals().data(), 12, n, f) != (size_t)n) return false;
    if (fread(pc.getColors().data(), 4, n, f) != (size_t)n) return false;
    return true;
}

int main(int argc, char** argv) {
    if (argc < 4) { std::fprintf(stderr, "usage: %s dump.bin metric multires\n", argv[0]); return 2; }
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    PointCloud source, target;
    if (!read_cloud(f, source) || !read_cloud(f, target)) return 2;
    std::fclose(f);

    ICPOptimizer* optimizer = new HipLinearICPOptimizer();          // main.cpp:51-56 `new LinearICPOptimizer()`
    optimizer->setMetric((unsigned)std::atoi(argv[2]));             // main.cpp:59-70
    optimizer->setNbOfIterations(20);
    optimizer->setMatchingMethod(0);                                // main.cpp:74-75
    optimizer->setMatchingMaxDistance(0.0003f);
    optimizer->setSelectionMethod(SELECT_ALL);                      // main.cpp:78-81
    optimizer->setWeightingMethod(CONSTANT_WEIGHTING);              // main.cpp:84-95
    optimizer->enableMultiResolution(std::atoi(argv[3]) != 0);      // main.cpp:97-98
    ConvergenceMeasure convergenMearsure;
    optimizer->setConvergenceMeasure(convergenMearsure);            // main.cpp:123-124
    TimeMeasure timeMeasure;
    optimizer->setTimeMeasure(timeMeasure);                         // main.cpp:127-128
    Matrix4f estimatedPose = Matrix4f::Identity();
    optimizer->estimatePose(source, target, estimatedPose);         // main.cpp:133

    // NearestNeighborSearch plugin used directly, as the reference's optimizer does (ICPOptimizer.h:535,565)
    NearestNeighborSearchHip nn(0);
    nn.setMatchingMaxDistance(0.0003f);
    nn.buildIndex(target.getPoints());
    std::vector<Match> matches = nn.queryMatches(source.getPoints());
    int valid = 0; for (const Match& m : matches) valid += m.idx >= 0;
    std::vector<Match> bad = nn.queryMatches(source.getPoints(), source.getColors());   // colour query on a 3-D index: error path

    std::printf("status %d iterations %zu recorded %zu valid_at_identity %d mismatch_status %d empty %d time_ok %d\n",
                static_cast<HipLinearICPOptimizer*>(optimizer)->lastStatus(), static_cast<HipLinearICPOptimizer*>(optimizer)->iterations().size(),
                convergenMearsure.recordedPoses.size(), valid, nn.lastStatus(), (int)bad.empty(),
                (int)(timeMeasure.matchingTime > 0 && timeMeasure.convergenceTime >= timeMeasure.matchingTime));
    std::printf("pose");
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) std::printf(" %.9g", estimatedPose(r, c));
    std::printf("\n");
    delete optimizer;
    return 0;
}
