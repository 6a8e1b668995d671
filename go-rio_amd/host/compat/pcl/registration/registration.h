// Minimal stand-in for pcl::Registration (PCL 1.10 semantics of align(): SURVEY.md appendix D): only what the drop-in class and
// its callers (scan_matching_odometry_nodelet.cpp:430-479, loop_detector.cpp:391-422) use.
#pragma once
#include <cfloat>
#include <cmath>
#include <memory>
#include <string>
#include "../point_cloud.h"
#include "../point_types.h"
#include "../search/kdtree.h"
#include <Eigen/Core>

namespace pcl {
template <typename PointSource, typename PointTarget, typename Scalar = float>
class Registration {
 public:
  using Matrix4 = Eigen::Matrix<Scalar, 4, 4>;
  using PointCloudSource = pcl::PointCloud<PointSource>;
  using PointCloudSourcePtr = typename PointCloudSource::Ptr;
  using PointCloudSourceConstPtr = typename PointCloudSource::ConstPtr;
  using PointCloudTarget = pcl::PointCloud<PointTarget>;
  using PointCloudTargetPtr = typename PointCloudTarget::Ptr;
  using PointCloudTargetConstPtr = typename PointCloudTarget::ConstPtr;
  using Ptr = std::shared_ptr<Registration<PointSource, PointTarget, Scalar>>;
  using KdTree = pcl::search::KdTree<PointTarget>;
  using KdTreePtr = typename KdTree::Ptr;

  Registration()
      : nr_iterations_(0), max_iterations_(10), converged_(false), transformation_epsilon_(0.0), corr_dist_threshold_(std::sqrt(DBL_MAX)), tree_(new KdTree),
        target_cloud_updated_(true), force_no_recompute_(false) {
    final_transformation_.setIdentity();
  }
  virtual ~Registration() {}
  virtual void setInputSource(const PointCloudSourceConstPtr& cloud) { input_ = cloud; }
  virtual void setInputTarget(const PointCloudTargetConstPtr& cloud) {  // PCL 1.10 registration.hpp: also flags the search tree as stale
    target_ = cloud;
    target_cloud_updated_ = true;
  }
  // PCL 1.10 registration.h: a caller-provided tree; with force_no_recompute the tree is NOT rebuilt when the target changes
  void setSearchMethodTarget(const KdTreePtr& tree, bool force_no_recompute = false) {
    tree_ = tree;
    force_no_recompute_ = force_no_recompute;
    target_cloud_updated_ = true;
  }
  KdTreePtr getSearchMethodTarget() const { return tree_; }
  void setMaximumIterations(int n) { max_iterations_ = n; }
  void setTransformationEpsilon(double e) { transformation_epsilon_ = e; }
  void setMaxCorrespondenceDistance(double d) { corr_dist_threshold_ = d; }
  double getMaxCorrespondenceDistance() const { return corr_dist_threshold_; }
  bool hasConverged() const { return converged_; }
  Matrix4 getFinalTransformation() const { return final_transformation_; }
  void align(PointCloudSource& output) { align(output, Matrix4::Identity()); }
  void align(PointCloudSource& output, const Matrix4& guess) {
    if (!input_ || !target_) return;  // initCompute() failure
    if (target_cloud_updated_ && !force_no_recompute_) {  // PCL 1.10 Registration::initCompute(): the CPU kd-tree of every new target
      tree_->setInputCloud(target_);
      target_cloud_updated_ = false;
    }
    output.points = input_->points;
    converged_ = false;
    final_transformation_.setIdentity();
    for (auto& p : output.points) p.data[3] = 1.0f;
    computeTransformation(output, guess);
  }
  virtual double getFitnessScore(double max_range = DBL_MAX) = 0;  // PCL implements this on its own kd-tree; the shim defers

 protected:
  virtual void computeTransformation(PointCloudSource& output, const Matrix4& guess) = 0;
  std::string reg_name_;
  PointCloudSourceConstPtr input_;
  PointCloudTargetConstPtr target_;
  int nr_iterations_;
  int max_iterations_;
  Matrix4 final_transformation_;
  bool converged_;
  double transformation_epsilon_;
  double corr_dist_threshold_;
  KdTreePtr tree_;
  bool target_cloud_updated_;
  bool force_no_recompute_;
};
}  // namespace pcl
