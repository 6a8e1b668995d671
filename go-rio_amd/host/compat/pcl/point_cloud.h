#pragma once
#include <memory>
#include <vector>
namespace pcl {
template <typename PointT>
class PointCloud {
 public:
  using Ptr = std::shared_ptr<PointCloud<PointT>>;
  using ConstPtr = std::shared_ptr<const PointCloud<PointT>>;
  std::vector<PointT> points;
  std::size_t size() const { return points.size(); }
  bool empty() const { return points.empty(); }
  void resize(std::size_t n) { points.resize(n); }
  PointT& at(std::size_t i) { return points.at(i); }
  const PointT& at(std::size_t i) const { return points.at(i); }
  PointT& operator[](std::size_t i) { return points[i]; }
  const PointT& operator[](std::size_t i) const { return points[i]; }
};
}  // namespace pcl
