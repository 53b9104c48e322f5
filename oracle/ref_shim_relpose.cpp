// TEST INFRASTRUCTURE ONLY (part of oracle/_ref/libalva_ref.so).  Marshalling layer around the reference's
// two-view initialisation: MultiViewGeometry::compute5ptEssentialMatrix
// (src/slam/src/multi_view_geometry.cpp:225-320) and the OpenGV pieces underneath it, exposed one by one so that every
// stage of the restatement in alva_oracle.c can be pinned separately.
#include <sstream>
#include <string>
#include <cstdint>
#include <cstring>
#include <vector>
#include <memory>
#include <unordered_map>
#include "frame.hpp"
#include "multi_view_geometry.hpp"
#include <opengv/relative_pose/methods.hpp>
#include <opengv/relative_pose/CentralRelativeAdapter.hpp>
#include <opengv/sac/Ransac.hpp>
#include <opengv/sac_problems/relative_pose/CentralRelativePoseSacProblem.hpp>
#include <opengv/math/Sturm.hpp>
#include <opengv/relative_pose/modules/fivept_nister/modules.hpp>
#include <Eigen/SVD>

namespace {
typedef opengv::sac_problems::relative_pose::CentralRelativePoseSacProblem Problem;

void fill(const double *bv, int n, opengv::bearingVectors_t &v) {
    v.resize(n);
    for (int i = 0; i < n; i++) v[i] = Eigen::Vector3d(bv[3 * i], bv[3 * i + 1], bv[3 * i + 2]);
}
void put_model(const opengv::transformation_t &T, double *m12) {  // R row-major (9) then t (3)
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) m12[3 * r + c] = T(r, c);
    for (int r = 0; r < 3; r++) m12[9 + r] = T(r, 3);
}
opengv::transformation_t get_model(const double *m12) {
    opengv::transformation_t T;
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) T(r, c) = m12[3 * r + c];
    for (int r = 0; r < 3; r++) T(r, 3) = m12[9 + r];
    return T;
}
}  // namespace

extern "C" {

// opengv::math::Sturm(p).findRoots() for a polynomial given highest power first.
int ref_sturm_roots(const double *coeffs, int ncoef, double *roots) {
    Eigen::MatrixXd p(1, ncoef);
    for (int i = 0; i < ncoef; i++) p(0, i) = coeffs[i];
    opengv::math::Sturm s(p);
    std::vector<double> r = s.findRoots();
    for (size_t i = 0; i < r.size(); i++) roots[i] = r[i];
    return (int) r.size();
}

// The null-space basis exactly as methods.cpp:246-263 forms it (JacobiSVD of the 5 x 9 constraint matrix); EE row-major [9][4].
void ref_nister_nullspace(const double *bv1, const double *bv2, double *EEout) {
    Eigen::MatrixXd Q(5, 9);
    for (int i = 0; i < 5; i++) {
        Eigen::Vector3d f(bv2[3 * i], bv2[3 * i + 1], bv2[3 * i + 2]), fp(bv1[3 * i], bv1[3 * i + 1], bv1[3 * i + 2]);
        Eigen::Matrix<double, 1, 9> row;
        row << f[0] * fp[0], f[1] * fp[0], f[2] * fp[0], f[0] * fp[1], f[1] * fp[1], f[2] * fp[1], f[0] * fp[2], f[1] * fp[2], f[2] * fp[2];
        Q.row(i) = row;
    }
    Eigen::JacobiSVD<Eigen::MatrixXd> SVD(Q, Eigen::ComputeFullV);
    Eigen::Matrix<double, 9, 4> EE = SVD.matrixV().block(0, 5, 9, 4);
    for (int r = 0; r < 9; r++)
        for (int c = 0; c < 4; c++) EEout[4 * r + c] = EE(r, c);
}

// fivept_nister::composeA (modules.cpp:38-368); EE row-major [9][4] in, A row-major [10][20] out.
void ref_nister_compose_a(const double *EEin, double *Aout) {
    Eigen::Matrix<double, 9, 4> EE;
    for (int r = 0; r < 9; r++)
        for (int c = 0; c < 4; c++) EE(r, c) = EEin[4 * r + c];
    Eigen::Matrix<double, 10, 20> A;
    opengv::relative_pose::modules::fivept_nister::composeA(EE, A);
    for (int r = 0; r < 10; r++)
        for (int c = 0; c < 20; c++) Aout[20 * r + c] = A(r, c);
}

// opengv::relative_pose::fivept_nister on the first five correspondences; E[k] row-major 3x3, returns the count (<= 10).
int ref_fivept_nister(const double *bv1, const double *bv2, double *E) {
    opengv::bearingVectors_t a, b;
    fill(bv1, 5, a);
    fill(bv2, 5, b);
    opengv::relative_pose::CentralRelativeAdapter adapter(a, b);
    std::vector<int> idx = {0, 1, 2, 3, 4};
    opengv::essentials_t es = opengv::relative_pose::fivept_nister(adapter, idx);
    for (size_t k = 0; k < es.size(); k++)
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) E[9 * k + 3 * r + c] = es[k](r, c);
    return (int) es.size();
}

// CentralRelativePoseSacProblem(NISTER)::computeModelCoefficients on an 8-index sample.
int ref_relpose_model(const double *bv1, const double *bv2, int n, const int *idx8, double *model12) {
    opengv::bearingVectors_t a, b;
    fill(bv1, n, a);
    fill(bv2, n, b);
    opengv::relative_pose::CentralRelativeAdapter adapter(a, b);
    Problem prob(adapter, Problem::NISTER, false);
    std::vector<int> idx(idx8, idx8 + 8);
    Problem::model_t m;
    if (!prob.computeModelCoefficients(idx, m)) return 0;
    put_model(m, model12);
    return 1;
}

// getDistancesToModel for every correspondence.
void ref_relpose_scores(const double *bv1, const double *bv2, int n, const double *model12, double *scores) {
    opengv::bearingVectors_t a, b;
    fill(bv1, n, a);
    fill(bv2, n, b);
    opengv::relative_pose::CentralRelativeAdapter adapter(a, b);
    Problem prob(adapter, Problem::NISTER, false);
    std::vector<double> d;
    prob.getDistancesToModel(get_model(model12), d);
    for (int i = 0; i < n; i++) scores[i] = d[i];
}

// optimizeModelCoefficients (= opengv::relative_pose::optimize_nonlinear on the inliers, starting from model12).
void ref_relpose_optimize(const double *bv1, const double *bv2, int n, const int *inliers, int nIn, const double *model12, double *out12) {
    opengv::bearingVectors_t a, b;
    fill(bv1, n, a);
    fill(bv2, n, b);
    opengv::relative_pose::CentralRelativeAdapter adapter(a, b);
    Problem prob(adapter, Problem::NISTER, false);
    std::vector<int> in(inliers, inliers + nIn);
    Problem::model_t o;
    prob.optimizeModelCoefficients(in, get_model(model12), o);
    put_model(o, out12);
}

// The same sequence of OpenGV calls as compute5ptEssentialMatrix (:256-289), returning the RANSAC internals as well:
// info[0] = iterations_, info[1] = number of inliers; ransacModel12 = model before the non-linear refinement;
// inlierMask[n].  Returns 0 when fewer than 10 inliers.
int ref_relpose_ransac(const double *bv1, const double *bv2, int n, int maxIterations, float errorThreshold, float fx, float fy,
                       double *ransacModel12, uint8_t *inlierMask, int *info) {
    opengv::bearingVectors_t a, b;
    fill(bv1, n, a);
    fill(bv2, n, b);
    opengv::relative_pose::CentralRelativeAdapter adapter(a, b);
    opengv::sac::Ransac<Problem> ransac;
    std::shared_ptr<Problem> prob(new Problem(adapter, Problem::NISTER, false));
    float focal = fx + fy;
    focal /= 2.;
    ransac.sac_model_ = prob;
    // multi_view_geometry.cpp:276 calls the C library's ::atan / ::cos (double) on the float quotient; in THIS translation unit the
    // unqualified names would pick libstdc++'s float overloads, so the double path is spelled out (checked against
    // ref_compute_5pt, which runs the reference's own line)
    ransac.threshold_ = 2.0 * (1.0 - ::cos((double) ::atan((double) (errorThreshold / focal))));
    ransac.max_iterations_ = maxIterations;
    ransac.computeModel(0);
    info[0] = ransac.iterations_;
    info[1] = (int) ransac.inliers_.size();
    memset(inlierMask, 0, (size_t) n);
    for (int i : ransac.inliers_) inlierMask[i] = 1;
    if (ransac.model_.empty()) return 0;
    put_model(ransac.model_coefficients_, ransacModel12);
    return ransac.inliers_.size() < 10 ? 0 : 1;
}

// MultiViewGeometry::compute5ptEssentialMatrix itself (doRandom = false => seed 12345u).
int ref_compute_5pt(const double *bv1, const double *bv2, int n, int maxIterations, float errorThreshold, int optimize, float fx,
                    float fy, double *R_out, double *t_out, int *outliers, int *nOutliers) {
    std::vector<Eigen::Vector3d, Eigen::aligned_allocator<Eigen::Vector3d>> a(n), b(n);
    for (int i = 0; i < n; i++) {
        a[i] = Eigen::Vector3d(bv1[3 * i], bv1[3 * i + 1], bv1[3 * i + 2]);
        b[i] = Eigen::Vector3d(bv2[3 * i], bv2[3 * i + 1], bv2[3 * i + 2]);
    }
    Eigen::Matrix3d R = Eigen::Matrix3d::Identity();
    Eigen::Vector3d t = Eigen::Vector3d::Zero();
    std::vector<int> out;
    bool ok = MultiViewGeometry::compute5ptEssentialMatrix(a, b, maxIterations, errorThreshold, optimize != 0, false, fx, fy, R, t, out);
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) R_out[3 * r + c] = R(r, c);
    for (int r = 0; r < 3; r++) t_out[r] = t(r);
    *nOutliers = (int) out.size();
    for (size_t i = 0; i < out.size(); i++) outliers[i] = out[i];
    return ok ? 1 : 0;
}

}  // extern "C"
