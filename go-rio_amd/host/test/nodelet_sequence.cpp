// Drives the drop-in fast_gicp::FastAPDGICP exactly the way Go-RIO's front end does (scan_matching_odometry_nodelet.cpp):
// factory setters of registrations.cpp:38-51, then per frame setInputTarget (first frame / new keyframe, SMO:430, 588),
// setInputSource (SMO:442), align(*aligned, guess) (SMO:465), hasConverged / getFinalTransformation (SMO:473-479),
// getFitnessScore (SMO:675) -- through a pcl::Registration base pointer, never the concrete class.
// Input: a binary file [int32 n_frames] then per frame [int32 n][n x (x,y,z,label) float32].  Output: one JSON line per frame pair.
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>

#include <fast_gicp/gicp/fast_apdgicp.hpp>

using PointT = pcl::PointXYZINormal;

static pcl::Registration<PointT, PointT>::Ptr select_registration_method() {  // registrations.cpp:38-51 with launch/ntu_loop3.launch:85-96
  std::shared_ptr<fast_gicp::FastAPDGICP<PointT, PointT>> apdgicp(new fast_gicp::FastAPDGICP<PointT, PointT>());
  apdgicp->setNumThreads(0);
  apdgicp->setTransformationEpsilon(0.1);
  apdgicp->setMaximumIterations(64);
  apdgicp->setMaxCorrespondenceDistance(2.0);
  apdgicp->setCorrespondenceRandomness(20);
  apdgicp->setDistVar(0.86);
  apdgicp->setAzimuthVar(0.5);
  apdgicp->setElevationVar(1.0);
  return apdgicp;
}

int main(int argc, char** argv) {
  if (argc < 2) {
    std::fprintf(stderr, "usage: %s frames.bin\n", argv[0]);
    return 2;
  }
  std::FILE* f = std::fopen(argv[1], "rb");
  if (!f) return 2;
  int n_frames = 0;
  if (std::fread(&n_frames, 4, 1, f) != 1) return 2;
  std::vector<pcl::PointCloud<PointT>::Ptr> frames;
  for (int k = 0; k < n_frames; ++k) {
    int n = 0;
    if (std::fread(&n, 4, 1, f) != 1) return 2;
    std::vector<float> buf((size_t)n * 4);
    if (std::fread(buf.data(), 4, buf.size(), f) != buf.size()) return 2;
    pcl::PointCloud<PointT>::Ptr c(new pcl::PointCloud<PointT>());
    c->resize(n);
    for (int i = 0; i < n; ++i) {
      PointT& p = c->points[i];
      p.x = buf[4 * i];
      p.y = buf[4 * i + 1];
      p.z = buf[4 * i + 2];
      p.normal_x = buf[4 * i + 3];
    }
    frames.push_back(c);
  }
  std::fclose(f);

  pcl::Registration<PointT, PointT>::Ptr registration;
  try {
    registration = select_registration_method();
  } catch (const std::exception& e) {
    std::fprintf(stderr, "%s\n", e.what());
    return 3;  // no GPU: the drop-in refuses instead of falling back to a CPU path
  }
  Eigen::Matrix4f prev_trans = Eigen::Matrix4f::Identity();
  registration->setInputTarget(frames[0]);  // first frame becomes the keyframe (SMO:430)
  for (int k = 1; k < n_frames; ++k) {
    registration->setInputSource(frames[k]);  // SMO:442
    pcl::PointCloud<PointT>::Ptr aligned(new pcl::PointCloud<PointT>());
    registration->align(*aligned, prev_trans);  // SMO:465
    const Eigen::Matrix4f T = registration->getFinalTransformation();
    const double fitness = registration->getFitnessScore();
    std::printf("{\"frame\": %d, \"converged\": %d, \"fitness\": %.17g, \"aligned0\": [%.9g, %.9g, %.9g], \"label0\": %.9g, \"T\": [", k, registration->hasConverged() ? 1 : 0, fitness,
                aligned->points[0].x, aligned->points[0].y, aligned->points[0].z, aligned->points[0].normal_x);
    for (int r = 0; r < 4; ++r)
      for (int c = 0; c < 4; ++c) std::printf("%.9g%s", T(r, c), (r == 3 && c == 3) ? "" : ", ");
    std::printf("]}\n");
    if (registration->hasConverged()) prev_trans = T;
    if (k % 2 == 0) {  // keyframe update (SMO:588): the new target's covariances must be recomputed, the source ones not reused
      registration->setInputTarget(frames[k]);
      prev_trans = Eigen::Matrix4f::Identity();
    }
  }
  // The CPU kd-tree of pcl::Registration (rebuilt by initCompute() for every new target in a stock PCL registration) is built by the
  // drop-in only when a caller really searches it, as the inlier loop of scan_matching_odometry_nodelet.cpp:679-689 does.
  const int builds_before = pcl::search::compat_tree_builds();
  std::vector<int> k_indices;
  std::vector<float> k_sq_dists;
  registration->getSearchMethodTarget()->nearestKSearch(frames[n_frames - 1]->points[0], 1, k_indices, k_sq_dists);
  std::printf("{\"kdtree_builds_during_sequence\": %d, \"kdtree_builds_after_use\": %d, \"nn_found\": %d}\n", builds_before, pcl::search::compat_tree_builds(), (int)k_indices.size());
  return 0;
}
