// fast_gicp::LsqRegistration for the MI355X back end.
//
// The reference keeps the Gauss-Newton / Levenberg-Marquardt loop in this class (lsq_registration_impl.hpp:55-173); here that loop runs
// on the GPU inside libgorio_amd.so (lm_solve_kernel), so what is left is the part of the class that callers and subclasses see:
// the optimiser settings, the final Hessian, evaluateCost(), and the three hooks a registration implements.  Names, namespace,
// template parameters and member names are the reference's (lsq_registration.hpp:15-85), so that code written against it --
// registrations.cpp:38-51, FastAPDGICP itself -- compiles unchanged.
#ifndef FAST_GICP_LSQ_REGISTRATION_HPP
#define FAST_GICP_LSQ_REGISTRATION_HPP

#include <Eigen/Core>
#include <Eigen/Geometry>
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#include <pcl/registration/registration.h>

namespace fast_gicp {

// which normal-equation step the device loop takes (reference default: LevenbergMarquardt)
enum class LSQ_OPTIMIZER_TYPE { GaussNewton, LevenbergMarquardt };

template <typename PointSource, typename PointTarget>
class LsqRegistration : public pcl::Registration<PointSource, PointTarget, float> {
  typedef pcl::Registration<PointSource, PointTarget, float> PclBase;
  typedef LsqRegistration<PointSource, PointTarget> Self;

public:
  EIGEN_MAKE_ALIGNED_OPERATOR_NEW

  // ---- types callers name (same spellings as the reference)
  typedef float Scalar;
  typedef typename PclBase::Matrix4 Matrix4;
  typedef typename PclBase::PointCloudSource PointCloudSource;
  typedef typename PclBase::PointCloudTarget PointCloudTarget;
  typedef typename PointCloudSource::Ptr PointCloudSourcePtr;
  typedef typename PointCloudSource::ConstPtr PointCloudSourceConstPtr;
  typedef typename PointCloudTarget::Ptr PointCloudTargetPtr;
  typedef typename PointCloudTarget::ConstPtr PointCloudTargetConstPtr;
  // the smart pointer flavour follows PCL's own: pcl::shared_ptr from 1.10 on, boost::shared_ptr before
#if PCL_VERSION >= PCL_VERSION_CALC(1, 10, 0)
  typedef pcl::shared_ptr<Self> Ptr;
  typedef pcl::shared_ptr<const Self> ConstPtr;
#else
  typedef boost::shared_ptr<Self> Ptr;
  typedef boost::shared_ptr<const Self> ConstPtr;
#endif

  // ---- construction: the defaults of lsq_registration_impl.hpp:10-24
  LsqRegistration()
  : rotation_epsilon_(2e-3),
    lsq_optimizer_type_(LSQ_OPTIMIZER_TYPE::LevenbergMarquardt),
    lm_max_iterations_(10),
    lm_init_lambda_factor_(1e-9),
    lm_lambda_(-1.0),
    lm_debug_print_(false) {
    this->reg_name_ = "LsqRegistration";
    this->max_iterations_ = 64;
    this->transformation_epsilon_ = 5e-4;
    final_hessian_.setIdentity();
  }
  virtual ~LsqRegistration() {}

  // ---- settings and results
  void setRotationEpsilon(double eps) { rotation_epsilon_ = eps; }                                    // convergence: max |R - I| / eps < 1
  void setInitialLambdaFactor(double init_lambda_factor) { lm_init_lambda_factor_ = init_lambda_factor; }  // lambda_0 = factor * max diag(H)
  void setDebugPrint(bool lm_debug_print) { lm_debug_print_ = lm_debug_print; }                       // kept for source compatibility
  const Eigen::Matrix<double, 6, 6>& getFinalHessian() const { return final_hessian_; }

  // error (and optionally H, b) of the current correspondence model at a float pose: one linearisation (lsq_registration_impl.hpp:50-52)
  double evaluateCost(const Eigen::Matrix4f& relative_pose, Eigen::Matrix<double, 6, 6>* H = nullptr, Eigen::Matrix<double, 6, 1>* b = nullptr) {
    const Eigen::Isometry3d pose(relative_pose.template cast<double>());
    return linearize(pose, H, b);
  }

  // registrations that cache per-cloud data override these
  virtual void clearSource() {}
  virtual void clearTarget() {}
  virtual void swapSourceAndTarget() {}

protected:
  // members of pcl::Registration this family of classes touches directly
  using PclBase::converged_;
  using PclBase::final_transformation_;
  using PclBase::input_;
  using PclBase::max_iterations_;
  using PclBase::nr_iterations_;
  using PclBase::transformation_epsilon_;

  // what a concrete registration supplies; computeTransformation() is where the device loop is entered
  virtual void computeTransformation(PointCloudSource& output, const Matrix4& guess) override = 0;
  virtual double linearize(const Eigen::Isometry3d& trans, Eigen::Matrix<double, 6, 6>* H = nullptr, Eigen::Matrix<double, 6, 1>* b = nullptr) = 0;
  virtual double compute_error(const Eigen::Isometry3d& trans) = 0;

  double rotation_epsilon_;                    // 2e-3
  LSQ_OPTIMIZER_TYPE lsq_optimizer_type_;      // LevenbergMarquardt
  int lm_max_iterations_;                      // error trials per outer iteration (10)
  double lm_init_lambda_factor_;               // 1e-9
  double lm_lambda_;                           // < 0: not initialised yet (reset by every align)
  bool lm_debug_print_;
  Eigen::Matrix<double, 6, 6> final_hessian_;  // H of the last accepted step
};

}  // namespace fast_gicp
#endif
