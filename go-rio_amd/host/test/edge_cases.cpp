// Edge cases of the drop-in fast_gicp::FastAPDGICP the reference's callers can produce (APD:115-155): an EMPTY cloud set on one side
// must not leave the previous cloud resident on the device; a covariance vector of the wrong size is stored but ignored (recomputed
// at align, APD:149-154), it does not throw; re-setting valid clouds afterwards gives the original result again.
// Input: the same frames file as nodelet_sequence.  Output: one JSON object.
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <string>
#include <vector>

#include <fast_gicp/gicp/fast_apdgicp.hpp>

using PointT = pcl::PointXYZINormal;
using Reg = fast_gicp::FastAPDGICP<PointT, PointT>;

static pcl::PointCloud<PointT>::Ptr read_frame(std::FILE* f) {
  int n = 0;
  if (std::fread(&n, 4, 1, f) != 1) return nullptr;
  std::vector<float> buf((size_t)n * 4);
  if (std::fread(buf.data(), 4, buf.size(), f) != buf.size()) return nullptr;
  pcl::PointCloud<PointT>::Ptr c(new pcl::PointCloud<PointT>());
  c->resize(n);
  for (int i = 0; i < n; ++i) {
    PointT& p = c->points[i];
    p.x = buf[4 * i]; p.y = buf[4 * i + 1]; p.z = buf[4 * i + 2]; p.normal_x = buf[4 * i + 3];
  }
  return c;
}

static void print_T(const char* key, const Eigen::Matrix4f& T) {
  std::printf("\"%s\": [", key);
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) std::printf("%.9g%s", T(r, c), (r == 3 && c == 3) ? "" : ", ");
  std::printf("]");
}

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  std::FILE* f = std::fopen(argv[1], "rb");
  if (!f) return 2;
  int n_frames = 0;
  if (std::fread(&n_frames, 4, 1, f) != 1 || n_frames < 2) return 2;
  auto a = read_frame(f), b = read_frame(f);
  std::fclose(f);
  if (!a || !b) return 2;
  std::unique_ptr<Reg> reg;
  try {
    reg.reset(new Reg());
  } catch (const std::exception& e) {
    std::fprintf(stderr, "%s\n", e.what());
    return 3;
  }
  reg->setTransformationEpsilon(0.1);
  reg->setMaxCorrespondenceDistance(2.0);
  pcl::PointCloud<PointT> aligned;
  reg->setInputTarget(a);
  reg->setInputSource(b);
  reg->align(aligned);
  const Eigen::Matrix4f T0 = reg->getFinalTransformation();
  const bool conv0 = reg->hasConverged();

  // 1. an empty source: the align must fail loudly, not register the stale cloud
  pcl::PointCloud<PointT>::Ptr empty(new pcl::PointCloud<PointT>());
  reg->setInputSource(empty);
  bool threw = false;
  try {
    reg->align(aligned);
  } catch (const std::exception&) {
    threw = true;
  }
  // 2. valid source again (a NEW pointer with the same content): same answer as the first time
  pcl::PointCloud<PointT>::Ptr b2(new pcl::PointCloud<PointT>(*b));
  reg->setInputSource(b2);
  reg->align(aligned);
  const Eigen::Matrix4f T1 = reg->getFinalTransformation();
  // 3. covariances of the wrong size: stored, ignored, no exception; the result does not change
  Reg::CovarianceVector wrong(7, Eigen::Matrix4d::Identity());
  bool cov_threw = false;
  try {
    reg->setSourceCovariances(wrong);
    reg->setTargetCovariances(wrong);
  } catch (const std::exception&) {
    cov_threw = true;
  }
  reg->align(aligned);
  const Eigen::Matrix4f T2 = reg->getFinalTransformation();
  const float inl = reg->getInlierFraction();
  const double fit = reg->getFitnessScore();
  // 3b. a second object that references the first one's device-resident target: same answer
  Reg other;
  other.setTransformationEpsilon(0.1);
  other.setMaxCorrespondenceDistance(2.0);
  other.setInputTargetShared(*reg);
  other.setInputSource(b2);
  pcl::PointCloud<PointT> aligned2;
  other.align(aligned2);
  const Eigen::Matrix4f T3 = other.getFinalTransformation();
  // 4. empty target
  reg->setInputTarget(empty);
  bool threw_t = false;
  try {
    reg->align(aligned);
  } catch (const std::exception&) {
    threw_t = true;
  }
  std::printf("{\"converged\": %d, \"empty_source_throws\": %d, \"empty_target_throws\": %d, \"cov_mismatch_throws\": %d, \"inlier_fraction\": %.9g, \"fitness\": %.17g, ",
              conv0 ? 1 : 0, threw ? 1 : 0, threw_t ? 1 : 0, cov_threw ? 1 : 0, inl, fit);
  print_T("T0", T0);
  std::printf(", ");
  print_T("T1", T1);
  std::printf(", ");
  print_T("T2", T2);
  std::printf(", ");
  print_T("T3", T3);
  std::printf("}\n");
  return 0;
}
