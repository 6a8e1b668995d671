// Drop-in for the reference's fast_gicp/gicp/fast_apdgicp.hpp (APDH:19-122) backed by the MI355X library libgorio_amd.so.
//
// Same namespace, class name, template parameters, public methods and protected virtuals as the reference, so
// 4DRadarSLAM/src/radar_graph_slam/registrations.cpp:38-51 compiles unchanged against this header and the nodelets keep driving
// it through pcl::Registration<PointXYZINormal,PointXYZINormal>::Ptr (scan_matching_odometry_nodelet.cpp:430-479).
// Differences: header-only (no impl/*.hpp, no explicit instantiation unit); setNumThreads is accepted and ignored; there is no
// CPU fallback -- if no HIP device is usable the constructor throws std::runtime_error.
//
// Data flow: setInputSource / setInputTarget copy x, y, z and normal_x (the cluster label written at
// preprocessing_nodelet_ntu.cpp:561-568) to the GPU as SoA; computeTransformation() runs k-NN covariances + the whole LM / GN
// loop on the device through gorio_apd_align(); covariances and correspondences are fetched lazily when asked for.
#ifndef FAST_GICP_FAST_APDGICP_HPP
#define FAST_GICP_FAST_APDGICP_HPP

#include <cfloat>
#include <cmath>
#include <iostream>
#include <limits>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include <Eigen/Core>
#include <Eigen/Geometry>

#include <pcl/point_types.h>
#include <pcl/point_cloud.h>
#include <pcl/registration/registration.h>

#include <fast_gicp/gicp/lsq_registration.hpp>
#include <fast_gicp/gicp/gicp_settings.hpp>

#include <gorio_apd.h>

namespace fast_gicp {

template <typename PointSource, typename PointTarget>
class FastAPDGICP : public LsqRegistration<PointSource, PointTarget> {
public:
  using Scalar = float;
  using Matrix4 = typename pcl::Registration<PointSource, PointTarget, Scalar>::Matrix4;
  using PointCloudSource = typename pcl::Registration<PointSource, PointTarget, Scalar>::PointCloudSource;
  using PointCloudSourcePtr = typename PointCloudSource::Ptr;
  using PointCloudSourceConstPtr = typename PointCloudSource::ConstPtr;
  using PointCloudTarget = typename pcl::Registration<PointSource, PointTarget, Scalar>::PointCloudTarget;
  using PointCloudTargetPtr = typename PointCloudTarget::Ptr;
  using PointCloudTargetConstPtr = typename PointCloudTarget::ConstPtr;
  using CovarianceVector = std::vector<Eigen::Matrix4d, Eigen::aligned_allocator<Eigen::Matrix4d>>;
#if PCL_VERSION >= PCL_VERSION_CALC(1, 10, 0)
  using Ptr = pcl::shared_ptr<FastAPDGICP<PointSource, PointTarget>>;
  using ConstPtr = pcl::shared_ptr<const FastAPDGICP<PointSource, PointTarget>>;
#else
  using Ptr = boost::shared_ptr<FastAPDGICP<PointSource, PointTarget>>;
  using ConstPtr = boost::shared_ptr<const FastAPDGICP<PointSource, PointTarget>>;
#endif

protected:
  using pcl::Registration<PointSource, PointTarget, Scalar>::reg_name_;
  using pcl::Registration<PointSource, PointTarget, Scalar>::input_;
  using pcl::Registration<PointSource, PointTarget, Scalar>::target_;
  using pcl::Registration<PointSource, PointTarget, Scalar>::corr_dist_threshold_;
  using pcl::Registration<PointSource, PointTarget, Scalar>::nr_iterations_;
  using pcl::Registration<PointSource, PointTarget, Scalar>::max_iterations_;
  using pcl::Registration<PointSource, PointTarget, Scalar>::final_transformation_;
  using pcl::Registration<PointSource, PointTarget, Scalar>::transformation_epsilon_;
  using pcl::Registration<PointSource, PointTarget, Scalar>::converged_;

public:
  explicit FastAPDGICP(int device = 0) {  // APD:14-28
    num_threads_ = 0;
    k_correspondences_ = 20;
    reg_name_ = "FastAPDGICP";
    corr_dist_threshold_ = std::numeric_limits<float>::max();
    regularization_method_ = RegularizationMethod::PLANE;
    const int rc = gorio_apd_create(&handle_, device);
    if (rc != GORIO_OK) throw std::runtime_error("FastAPDGICP: no usable HIP device (gorio_apd_create failed, code " + std::to_string(rc) + "); there is no CPU fallback");
    // pcl::Registration::align() -> initCompute() rebuilds its CPU kd-tree for every new target (PCL 1.10 registration.hpp:
    // `if (target_cloud_updated_ && !force_no_recompute_) tree_->setInputCloud(target_)`): tens of milliseconds for a 100 k-point map,
    // more than the whole GPU registration, for a tree this class never searches.  So the base class gets a tree that is built only if
    // somebody actually uses getSearchMethodTarget() (the inlier loop of scan_matching_odometry_nodelet.cpp:679-689 does; prefer
    // getInlierFraction(), which runs on the GPU), and is told not to rebuild it.
    typename pcl::Registration<PointSource, PointTarget, Scalar>::KdTreePtr tree(new LazyTargetTree());
    lazy_tree_ = static_cast<LazyTargetTree*>(tree.get());  // owned by the base class' tree_ from here on
    this->setSearchMethodTarget(tree, /*force_no_recompute=*/true);
  }
  virtual ~FastAPDGICP() override { gorio_apd_destroy(handle_); }
  FastAPDGICP(const FastAPDGICP&) = delete;
  FastAPDGICP& operator=(const FastAPDGICP&) = delete;

  void setNumThreads(int n) { num_threads_ = n; }  // APD:34-42: OpenMP team size; meaningless on the GPU
  void setCorrespondenceRandomness(int k) { k_correspondences_ = k; }
  void setRegularizationMethod(RegularizationMethod method) { regularization_method_ = method; }
  void setAzimuthVar(double var) { azimuth_variance_ = var; }
  void setElevationVar(double var) { elevation_variance_ = var; }
  void setDistVar(double var) { distance_variance_ = var; }

  virtual void swapSourceAndTarget() override {  // APD:89-98
    input_.swap(target_);
    lazy_tree_->defer(target_);
    check(gorio_apd_swap_source_and_target(handle_));
    source_covs_.swap(target_covs_);
    std::swap(source_covs_fresh_, target_covs_fresh_);
  }
  virtual void clearSource() override {  // APD:101-105
    input_.reset();
    source_covs_.clear();
    source_covs_fresh_ = false;
    check(gorio_apd_clear_source(handle_));
  }
  virtual void clearTarget() override {  // APD:107-112
    target_.reset();
    lazy_tree_->defer(PointCloudTargetConstPtr());
    target_covs_.clear();
    target_covs_fresh_ = false;
    check(gorio_apd_clear_target(handle_));
  }

  virtual void setInputSource(const PointCloudSourceConstPtr& cloud) override {  // APD:115-124
    if (input_ == cloud) return;
    pcl::Registration<PointSource, PointTarget, Scalar>::setInputSource(cloud);
    upload(cloud->points.data(), static_cast<int>(cloud->size()), sizeof(PointSource), true);
    source_covs_.clear();
    source_covs_fresh_ = false;
  }
  virtual void setInputTarget(const PointCloudTargetConstPtr& cloud) override {  // APD:127-135
    if (target_ == cloud) return;
    pcl::Registration<PointSource, PointTarget, Scalar>::setInputTarget(cloud);
    lazy_tree_->defer(cloud);
    upload(cloud->points.data(), static_cast<int>(cloud->size()), sizeof(PointTarget), false);
    target_covs_.clear();
    target_covs_fresh_ = false;
  }
  // APD:138-145 store the vector whatever its size; APD:149-154 recompute at align when the size does not match the cloud.  The ABI
  // call does the same: a mismatching set leaves the device without covariances for that cloud (no exception).
  virtual void setSourceCovariances(const CovarianceVector& covs) {  // APD:138-140
    source_covs_ = covs;
    source_covs_fresh_ = true;
    check(gorio_apd_set_source_covariances(handle_, covs.empty() ? nullptr : to_row_major_covs(covs).data(), static_cast<int>(covs.size())));
  }
  virtual void setTargetCovariances(const CovarianceVector& covs) {  // APD:143-145
    target_covs_ = covs;
    target_covs_fresh_ = true;
    check(gorio_apd_set_target_covariances(handle_, covs.empty() ? nullptr : to_row_major_covs(covs).data(), static_cast<int>(covs.size())));
  }
  const CovarianceVector& getSourceCovariances() const {  // APDH:73-75
    fetch_covs(true);
    return source_covs_;
  }
  const CovarianceVector& getTargetCovariances() const {  // APDH:77-79
    fetch_covs(false);
    return target_covs_;
  }

  // pcl::Registration::getFitnessScore as the nodelets call it (SMO:675, loop_detector.cpp:411), computed on the GPU from the
  // device-resident clouds.  Through a pcl::Registration base pointer real PCL still runs its own CPU version.
  double getFitnessScore(double max_range = std::numeric_limits<double>::max()) {
    float T[16];
    to_row_major(final_transformation_, T);
    double score = 0.0;
    check(gorio_apd_fitness_score(handle_, T, max_range, 0.0, &score, nullptr));
    return score;
  }
  // the inlier loop of publish_scan_matching_status (SMO:677-689) on the GPU: share of source points whose squared NN distance in
  // the target, after final_transformation_, is below max_correspondence_dist^2 (the nodelet hard-codes 0.5 m); float like SMO:689
  float getInlierFraction(double max_correspondence_dist = 0.5) {
    float T[16];
    to_row_major(final_transformation_, T);
    double score = 0.0, frac = 0.0;
    check(gorio_apd_fitness_score(handle_, T, 0.0, max_correspondence_dist, &score, &frac));
    const std::size_t n = input_ ? input_->size() : 0;
    return n ? static_cast<float>(std::llround(frac * static_cast<double>(n))) / n : 0.0f;
  }
  // Scan-to-submap target assembly on the GPU (extra; replaces the CPU loop of scan_matching_odometry_nodelet.cpp:602-612): keyframe
  // clouds moved by their relative poses, concatenated, downsampled (voxel_leaf <= 0: the launch files' NONE), set as the target.  The
  // assembled cloud is also returned as a pcl cloud and kept as target_, because pcl::Registration::align() insists on one.
  PointCloudTargetConstPtr setInputTargetSubmap(const std::vector<PointCloudTargetConstPtr>& clouds,
                                                const std::vector<Eigen::Matrix4d, Eigen::aligned_allocator<Eigen::Matrix4d>>& rel_poses, double voxel_leaf = 0.0) {
    if (clouds.empty() || clouds.size() != rel_poses.size()) throw std::invalid_argument("setInputTargetSubmap: one relative pose per keyframe cloud");
    std::vector<gorio_apd_keyframe> fr(clouds.size());
    std::vector<double> poses(clouds.size() * 16);
    for (std::size_t k = 0; k < clouds.size(); ++k) {
      for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) poses[k * 16 + r * 4 + c] = rel_poses[k](r, c);
      const bool any = clouds[k] && !clouds[k]->points.empty();
      fr[k].xyz = any ? clouds[k]->points[0].data : nullptr;
      fr[k].label = any ? &clouds[k]->points[0].normal_x : nullptr;
      fr[k].n = any ? static_cast<int>(clouds[k]->size()) : 0;
      fr[k].point_stride_bytes = static_cast<int>(sizeof(PointTarget));
      fr[k].rel_pose = &poses[k * 16];
    }
    int n = 0;
    check(gorio_apd_set_target_submap(handle_, fr.data(), static_cast<int>(fr.size()), voxel_leaf, &n));
    PointCloudTargetPtr out(new PointCloudTarget());
    out->resize(n);
    if (n > 0) check(gorio_apd_get_target_points(handle_, out->points[0].data, &out->points[0].normal_x, n, static_cast<int>(sizeof(PointTarget))));
    for (auto& p : out->points) p.data[3] = 1.0f;
    pcl::Registration<PointSource, PointTarget, Scalar>::setInputTarget(out);  // bookkeeping only: the device already holds it
    lazy_tree_->defer(out);
    target_covs_.clear();
    target_covs_fresh_ = false;
    return out;
  }
  // One map for many registrations (extra; C5 of the benchmark plan): reference the target another object already holds on the
  // device -- its points, search index and covariances -- instead of uploading and indexing a private copy.
  void setInputTargetShared(FastAPDGICP& owner) {
    pcl::Registration<PointSource, PointTarget, Scalar>::setInputTarget(owner.target_);
    lazy_tree_->defer(owner.target_);
    check(gorio_apd_set_target_shared(handle_, owner.handle_));
    target_covs_.clear();
    target_covs_fresh_ = false;
  }
  // extras (not in the reference): correspondences of the last linearisation, device handle
  void getCorrespondences(std::vector<int>& corr, std::vector<float>& sq_dist) {
    const int n = static_cast<int>(input_->size());
    corr.resize(n);
    sq_dist.resize(n);
    check(gorio_apd_get_correspondences(handle_, corr.data(), sq_dist.data(), n));
  }
  gorio_apd_t* handle() { return handle_; }

protected:
  virtual void computeTransformation(PointCloudSource& output, const Matrix4& guess) override {  // APD:148-157 + LSQ:55-80
    push_params();
    float g[16], T[16];
    to_row_major(guess, g);
    double H[36];
    int conv = 0, nit = 0;
    check(gorio_apd_align(handle_, g, T, H, &conv, &nit, nullptr));
    for (int r = 0; r < 4; ++r)
      for (int c = 0; c < 4; ++c) final_transformation_(r, c) = T[r * 4 + c];
    for (int r = 0; r < 6; ++r)
      for (int c = 0; c < 6; ++c) this->final_hessian_(r, c) = H[r * 6 + c];
    converged_ = conv != 0;
    nr_iterations_ = nit;
    if (!converged_ && std::string(gorio_apd_last_error(handle_)) == "lm not converged!!") std::cerr << "lm not converged!!" << std::endl;  // LSQ:72
    source_covs_fresh_ = target_covs_fresh_ = false;  // device now holds (possibly newly computed) covariances
    source_covs_.clear();
    target_covs_.clear();
    // pcl::transformPointCloud(*input_, output, final_transformation_), LSQ:79 (xyz only, labels untouched)
    const int n = static_cast<int>(input_->size());
    if (static_cast<int>(output.size()) != n) output.points = input_->points;
    if (n > 0) check(gorio_apd_transform_source(handle_, T, output.points[0].data, n, sizeof(PointSource)));
  }

  virtual void update_correspondences(const Eigen::Isometry3d& trans) { linearize(trans, nullptr, nullptr); }  // APD:160-220

  virtual double linearize(const Eigen::Isometry3d& trans, Eigen::Matrix<double, 6, 6>* H, Eigen::Matrix<double, 6, 1>* b) override {  // APD:224-307
    push_params();
    double T[16], Hr[36], br[6], err = 0.0;
    to_row_major(trans.matrix(), T);
    check(gorio_apd_linearize(handle_, T, Hr, br, &err));
    if (H && b) {
      for (int r = 0; r < 6; ++r) {
        for (int c = 0; c < 6; ++c) (*H)(r, c) = Hr[r * 6 + c];
        (*b)(r, 0) = br[r];
      }
    }
    return err;
  }

  virtual double compute_error(const Eigen::Isometry3d& trans) override {  // APD:310-346
    double T[16], err = 0.0;
    to_row_major(trans.matrix(), T);
    check(gorio_apd_compute_error(handle_, T, &err));
    return err;
  }

private:
  template <typename M, typename S>
  static void to_row_major(const M& m, S* out) {
    for (int r = 0; r < 4; ++r)
      for (int c = 0; c < 4; ++c) out[r * 4 + c] = static_cast<S>(m(r, c));
  }
  static std::vector<double> to_row_major_covs(const CovarianceVector& covs) {  // Eigen::Matrix4d is column-major, the ABI row-major
    std::vector<double> rm(covs.size() * 16);
    for (std::size_t i = 0; i < covs.size(); ++i)
      for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) rm[i * 16 + r * 4 + c] = covs[i](r, c);
    return rm;
  }
  void check(int rc) const {
    if (rc < 0) throw std::runtime_error(std::string("FastAPDGICP (gorio_amd): ") + gorio_apd_last_error(handle_) + " [code " + std::to_string(rc) + "]");
  }
  template <typename P>
  void upload(const P* pts, int n, std::size_t stride, bool source) {
    if (n <= 0) {  // an empty cloud replaces the previous one: nothing must stay resident on the device for this side (align then fails)
      check(source ? gorio_apd_clear_source(handle_) : gorio_apd_clear_target(handle_));
      return;
    }
    const float* xyz = pts[0].data;
    const float* label = &pts[0].normal_x;  // cluster label, APD:272
    check(source ? gorio_apd_set_source(handle_, xyz, label, n, static_cast<int>(stride)) : gorio_apd_set_target(handle_, xyz, label, n, static_cast<int>(stride)));
  }
  void push_params() {  // REG:41-48 setters + LSQ / APD members -> one struct
    gorio_apd_params p;
    gorio_apd_default_params(&p);
    p.k_correspondences = k_correspondences_;
    p.regularization = static_cast<int>(regularization_method_);
    p.dist_var = distance_variance_;
    p.azimuth_var = azimuth_variance_;
    p.elevation_var = elevation_variance_;
    p.corr_dist_threshold = corr_dist_threshold_;
    p.max_iterations = max_iterations_;
    p.rotation_epsilon = this->rotation_epsilon_;
    p.transformation_epsilon = transformation_epsilon_;
    p.optimizer = this->lsq_optimizer_type_ == LSQ_OPTIMIZER_TYPE::GaussNewton ? GORIO_OPT_GAUSS_NEWTON : GORIO_OPT_LEVENBERG_MARQUARDT;
    p.lm_max_iterations = this->lm_max_iterations_;
    p.lm_init_lambda_factor = this->lm_init_lambda_factor_;
    check(gorio_apd_set_params(handle_, &p));
  }
  void fetch_covs(bool source) const {
    CovarianceVector& v = source ? source_covs_ : target_covs_;
    bool& fresh = source ? source_covs_fresh_ : target_covs_fresh_;
    if (fresh) return;
    auto getter = source ? gorio_apd_get_source_covariances : gorio_apd_get_target_covariances;
    const int cnt = getter(handle_, nullptr, 0);
    if (cnt < 0) check(cnt);
    v.resize(cnt);
    if (cnt > 0) {
      std::vector<double> rm(static_cast<std::size_t>(cnt) * 16);
      check(getter(handle_, rm.data(), cnt));
      for (int i = 0; i < cnt; ++i)
        for (int r = 0; r < 4; ++r)
          for (int c = 0; c < 4; ++c) v[i](r, c) = rm[static_cast<std::size_t>(i) * 16 + r * 4 + c];
    }
    fresh = true;
  }

protected:
  // pcl::search::KdTree whose index is built on FIRST USE instead of at every setInputTarget (see the constructor): the registration
  // itself never searches it
  class LazyTargetTree : public pcl::search::KdTree<PointTarget> {
    using Base = pcl::search::KdTree<PointTarget>;

  public:
    void defer(const PointCloudTargetConstPtr& cloud) {
      pending_ = cloud;
      dirty_ = true;
    }
    int nearestKSearch(const PointTarget& point, int k, std::vector<int>& k_indices, std::vector<float>& k_sqr_distances) const override {
      build();
      return Base::nearestKSearch(point, k, k_indices, k_sqr_distances);
    }
    int radiusSearch(const PointTarget& point, double radius, std::vector<int>& k_indices, std::vector<float>& k_sqr_distances, unsigned int max_nn = 0) const override {
      build();
      return Base::radiusSearch(point, radius, k_indices, k_sqr_distances, max_nn);
    }

  private:
    void build() const {
      if (!dirty_) return;
      dirty_ = false;
      if (pending_) const_cast<LazyTargetTree*>(this)->Base::setInputCloud(pending_);
    }
    mutable PointCloudTargetConstPtr pending_;
    mutable bool dirty_ = false;
  };
  LazyTargetTree* lazy_tree_ = nullptr;

  int num_threads_;
  int k_correspondences_;
  RegularizationMethod regularization_method_;
  double azimuth_variance_ = 0.5;
  double elevation_variance_ = 1.0;
  double distance_variance_ = 0.86;
  gorio_apd_t* handle_ = nullptr;
  mutable CovarianceVector source_covs_, target_covs_;
  mutable bool source_covs_fresh_ = false, target_covs_fresh_ = false;
};
}  // namespace fast_gicp

#endif
