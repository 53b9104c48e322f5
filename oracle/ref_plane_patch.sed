# TEST INFRASTRUCTURE ONLY.  The repairs build_ref_shim.sh applies to a COPY of /root/reference/src/slam/src/system.cpp (the copy lives
# under oracle/_ref/build/patched/, git-ignored) before compiling it as class SystemPlanePatched -- see oracle/ref_shim_plane.cpp.
# Nothing else of the file changes; line numbers are those of the reference's system.cpp.
#
# (0) the sampler hook's declaration, in front of the first function
1i\
#include <vector>\
void alva_ref_plane_pick(int iteration, std::vector<int> &picked);
# (1) :199-201  points as 3x1 CV_32F (cv::eigen2cv re-creates its destination as 3x1 CV_64F; everything after reads float)
s|points\[i\] = pointWorldPos;|pointWorldPos.convertTo(points[i], CV_32F);|
# (2) :210      the three sample indices from the caller instead of a generator re-seeded from std::random_device every iteration
s|std::sample(indices.begin(), indices.end(), indicesPicked.begin(), 3, std::mt19937{std::random_device{}()});|alva_ref_plane_pick(n, indicesPicked);|
# (3) :213, :217, :274, :286, :333  "submatrix = expression" written as a copy INTO the submatrix (the assignment re-points the
#     temporary header whenever the expression's type or shape differs from it)
s|A.col(3) = cv::Mat::ones(3, 1, CV_32F);|A.col(3).setTo(1.0f);|
s|A.row(i).colRange(0, 3) = points\[indicesPicked\[i\]\].t();|cv::Mat(points[indicesPicked[i]].t()).copyTo(A.row(i).colRange(0, 3));|
s|planeCoefficientsMatrix.col(3) = cv::Mat::ones(numInliers, 1, CV_32F);|planeCoefficientsMatrix.col(3).setTo(1.0f);|
s|planeCoefficientsMatrix.row(i).colRange(0, 3) = worldPoint.t();|cv::Mat(worldPoint.t()).copyTo(planeCoefficientsMatrix.row(i).colRange(0, 3));|
s|planePose.rowRange(0, 3).colRange(0, 3) = R1 \* R2;|cv::Mat(R1 * R2).copyTo(planePose.rowRange(0, 3).colRange(0, 3));|
# (4) :238-246  the score is the k-th smallest distance, found with std::nth_element IN PLACE -- which permutes dists -- and the permuted
#     array is then kept as the per-point distances of the best hypothesis ("distances = dists"), so the inlier test at :253-259 pairs
#     point i with some other point's distance.  The selection runs on a copy; dists keeps the point order.
s|std::nth_element(dists.begin(), dists.begin() + std::max((int) (0.2 \* numMapPoints), 20), dists.end());|std::vector<float> distsSorted = dists; std::nth_element(distsSorted.begin(), distsSorted.begin() + std::max((int) (0.2 * numMapPoints), 20), distsSorted.end());|
s|const float medianDist = dists\[std::max((int) (0.2 \* numMapPoints), 20)\];|const float medianDist = distsSorted[std::max((int) (0.2 * numMapPoints), 20)];|
