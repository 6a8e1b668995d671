// Minimal stand-in for pcl::search::KdTree (PCL 1.10 signatures of the members the drop-in and its callers touch).  The real class
// builds a FLANN kd-tree in setInputCloud -- tens of milliseconds for a 100 k-point map, on the CPU; this stand-in only COUNTS those
// builds (pcl::search::compat_tree_builds) so that host/test/nodelet_sequence.cpp can show when pcl::Registration would have paid for
// one, and answers nearestKSearch by brute force.
#pragma once
#include <limits>
#include <memory>
#include <vector>
#include "../point_cloud.h"

namespace pcl {
namespace search {
inline int& compat_tree_builds() {
  static int n = 0;
  return n;
}
template <typename PointT>
class KdTree {
 public:
  using Ptr = std::shared_ptr<KdTree<PointT>>;
  using ConstPtr = std::shared_ptr<const KdTree<PointT>>;
  using PointCloudConstPtr = typename pcl::PointCloud<PointT>::ConstPtr;
  virtual ~KdTree() {}
  virtual void setInputCloud(const PointCloudConstPtr& cloud) {
    input_ = cloud;
    ++compat_tree_builds();
  }
  PointCloudConstPtr getInputCloud() const { return input_; }
  virtual int nearestKSearch(const PointT& point, int k, std::vector<int>& k_indices, std::vector<float>& k_sqr_distances) const {
    k_indices.clear();
    k_sqr_distances.clear();
    if (!input_ || k != 1) return 0;  // the callers of this path ask for one neighbour (scan_matching_odometry_nodelet.cpp:679-689)
    int best = -1;
    float bd = std::numeric_limits<float>::max();
    for (std::size_t i = 0; i < input_->size(); ++i) {
      const float dx = point.x - input_->points[i].x, dy = point.y - input_->points[i].y, dz = point.z - input_->points[i].z;
      const float d = dx * dx + dy * dy + dz * dz;
      if (d < bd) {
        bd = d;
        best = static_cast<int>(i);
      }
    }
    if (best < 0) return 0;
    k_indices.push_back(best);
    k_sqr_distances.push_back(bd);
    return 1;
  }
  virtual int radiusSearch(const PointT& point, double radius, std::vector<int>& k_indices, std::vector<float>& k_sqr_distances, unsigned int max_nn = 0) const {
    k_indices.clear();
    k_sqr_distances.clear();
    if (!input_) return 0;
    const float r2 = static_cast<float>(radius * radius);
    for (std::size_t i = 0; i < input_->size(); ++i) {
      const float dx = point.x - input_->points[i].x, dy = point.y - input_->points[i].y, dz = point.z - input_->points[i].z;
      const float d = dx * dx + dy * dy + dz * dz;
      if (d <= r2 && (max_nn == 0 || k_indices.size() < max_nn)) {
        k_indices.push_back(static_cast<int>(i));
        k_sqr_distances.push_back(d);
      }
    }
    return static_cast<int>(k_indices.size());
  }

 protected:
  PointCloudConstPtr input_;
};
}  // namespace search
}  // namespace pcl
