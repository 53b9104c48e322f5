// TEST INFRASTRUCTURE ONLY (part of oracle/_ref/libalva_ref.so) -- never linked into or called from alvaar_amd/.
//
// ref_find_plane_patched: the reference's OWN System::processPlane (src/slam/src/system.cpp:177-342) with its four defects repaired,
// as the second implementation alva_find_plane (f3) is compared with.  As shipped the function has no defined behaviour:
// cv::eigen2cv turns each point into a 3x1 CV_64F matrix which every later at<float>() reads as halves of doubles (:199-201, :235);
// the assignments "submatrix = expression" (:213, :217, :274, :286, :333) therefore see a type mismatch, re-allocate the temporary
// header and leave the design matrices unfilled; the three sample indices come from a generator that is re-created from
// std::random_device in every iteration (:210); and the k-th smallest distance is found with std::nth_element in place, after which the
// PERMUTED array is kept as the best hypothesis' per-point distances, so the inlier test pairs point i with another point's distance
// (:238-259).
//
// build_ref_shim.sh makes the repaired translation unit WITHOUT copying any reference text into this repository: it copies
// system.cpp into oracle/_ref/build/patched/ (git-ignored), applies the edits of oracle/ref_plane_patch.sed to the copy -- each edit
// is listed there with the defect it repairs -- and compiles the copy with -DSystem=SystemPlanePatched, so that the repaired class
// lives beside the untouched one (ref_system_* keeps running the reference exactly as shipped).  The sampler edit calls
// alva_ref_plane_pick below: the caller supplies the three indices of every iteration (ascending, as std::sample returns them), the
// same ones alva_find_plane gets through h_samples3.
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>
#include <Eigen/Core>
#include <sophus/se3.hpp>
#include <opencv2/core.hpp>
#define System SystemPlanePatched
#define private public  // only the reference's own headers are included under this
#include "system.hpp"
#undef private
#include "utils.hpp"

static const int *g_samples = nullptr;
static int g_iterations = 0;

// called by the repaired processPlane instead of std::sample(..., std::mt19937{std::random_device{}()})
void alva_ref_plane_pick(int iteration, std::vector<int> &picked) {
    for (int k = 0; k < 3; k++) picked[k] = g_samples[3 * (iteration < g_iterations ? iteration : g_iterations - 1) + k];
}

extern "C" int ref_find_plane_patched(const double *points_xyz, int n, const double *pose7_twc, const int *samples3, int num_iterations, float *pose16) {
    std::vector<Eigen::Vector3d> pts((size_t) n);
    for (int i = 0; i < n; i++) pts[(size_t) i] = Eigen::Vector3d(points_xyz[3 * i], points_xyz[3 * i + 1], points_xyz[3 * i + 2]);
    const Eigen::Quaterniond q(pose7_twc[6], pose7_twc[3], pose7_twc[4], pose7_twc[5]);   // pose7 = (t, qx qy qz qw)
    const Sophus::SE3d Twc(q.normalized(), Eigen::Vector3d(pose7_twc[0], pose7_twc[1], pose7_twc[2]));
    SystemPlanePatched s;
    s.state_ = std::make_shared<State>(640, 480, 40);
    s.state_->debug_ = false;
    g_samples = samples3;
    g_iterations = num_iterations;
    cv::Mat m = s.processPlane(pts, Twc, num_iterations);
    g_samples = nullptr;
    if (m.empty()) return 0;
    Utils::toPoseArray(m, pose16);
    return 1;
}
