/*
 * ugpm_oracle.cpp -- CPU restatement of Go-RIO's UGPM GP pre-integration (TEST INFRASTRUCTURE ONLY).
 *
 * Parity oracle for the HIP path; never linked into or called by the product.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it.
 *
 * Restates, in plain C++17 (no Eigen / Ceres: both absent from the build image), the non-chunked UGPM branch of
 *   PRE   = /root/reference/4DRadarSLAM/include/VelInt/preint.h
 *   MATH  = /root/reference/4DRadarSLAM/include/VelInt/math_utils.h
 *   COST  = /root/reference/4DRadarSLAM/include/VelInt/cost_functions.h
 *   TYPES = /root/reference/4DRadarSLAM/include/VelInt/types.h
 * as called by the back end (radar_graph_slam_nodelet.cpp:465-530).  Each function cites the lines it follows.
 *
 * PARITY UNPINNED: the reference holds no test, example or recorded vector for VelInt and cannot be
 * compiled here.  The oracle is pinned only by analytic cases (constant angular rate => delta_R = Exp(w T); zero rotation +
 * constant velocity => delta_p = v T; seKernelIntegral against numerical quadrature of seKernel; kssInt against double
 * quadrature) and by a SciPy least-squares cross-check of the two GP fits (tests/test_oracle_ugpm.py).
 *
 * Third-party behaviour restated from the published algorithm of the pinned versions (docker/Dockerfile:24,36):
 *  - Ceres Solver 2.1.0 Solve() with the options of PRE:943-948 (1 thread, <= 50 iterations, DENSE_NORMAL_CHOLESKY,
 *    function_tolerance 1e-10, all else default): trust-region Levenberg-Marquardt, Jacobi scaling 1/(1+||J_col||) fixed at
 *    the initial point, step = -(J^T J + D^2/mu)^-1 J^T r with D^2 = clamp(diag(J^T J), 1e-6, 1e32), mu0 = 1e4, accept when
 *    rho > 1e-3, mu /= max(1/3, 1-(2 rho-1)^3) on accept, mu /= 2,4,8.. on reject, termination on |dcost| <= 1e-10 cost,
 *    |step| <= 1e-8 (|x| + 1e-8), max|grad| <= 1e-10, 50 iterations.  Constant parameter blocks are removed together with the
 *    residual blocks that depend only on them (PRE:962-964).
 *  - Eigen 3.3.7: MatrixXd::inverse() = partial-pivot LU (PRE:837), LLT + triangular solves (PRE:1482-1484), AngleAxisd <->
 *    matrix through a quaternion (MATH:48-58), array erf() = std::erf (MATH:120-123).
 *  - std::sort on indices (TYPES:324-325) is unstable; equal stamps occur by construction and the results do not depend on
 *    their order (zero-length integration steps), see SURVEY appendix C UGPM-6.
 *
 * Chunked mode (opt.quantum > 0, PRE:1584-1702) is not restated: Go-RIO never uses it and it indexes a 9-vector up to 11
 * (TYPES:36 vs MATH:545).
 */
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <numeric>
#include <stdexcept>
#include <thread>
#include <vector>

namespace ugpmo {

// ------------------------------------------------------------------------------------------------ small dense algebra
using V3 = std::array<double, 3>;
struct M3 {
  double m[9];
  double& operator()(int r, int c) { return m[r * 3 + c]; }
  double operator()(int r, int c) const { return m[r * 3 + c]; }
};
static M3 I3() { return M3{{1, 0, 0, 0, 1, 0, 0, 0, 1}}; }
static M3 mul(const M3& a, const M3& b) {
  M3 r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r(i, j) = a(i, 0) * b(0, j) + a(i, 1) * b(1, j) + a(i, 2) * b(2, j);
  return r;
}
static M3 tr(const M3& a) {
  M3 r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r(i, j) = a(j, i);
  return r;
}
static V3 mul(const M3& a, const V3& v) { return {a(0, 0) * v[0] + a(0, 1) * v[1] + a(0, 2) * v[2], a(1, 0) * v[0] + a(1, 1) * v[1] + a(1, 2) * v[2], a(2, 0) * v[0] + a(2, 1) * v[1] + a(2, 2) * v[2]}; }
static V3 operator-(const V3& a, const V3& b) { return {a[0] - b[0], a[1] - b[1], a[2] - b[2]}; }
static V3 operator+(const V3& a, const V3& b) { return {a[0] + b[0], a[1] + b[1], a[2] + b[2]}; }
static V3 operator*(double s, const V3& a) { return {s * a[0], s * a[1], s * a[2]}; }
static double norm(const V3& a) { return std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }
static M3 skew(const V3& v) { return M3{{0.0, -v[2], v[1], v[2], 0.0, -v[0], -v[1], v[0], 0.0}}; }  // MATH:187-195
static M3 add(const M3& a, const M3& b, double sb = 1.0) {
  M3 r;
  for (int i = 0; i < 9; i++) r.m[i] = a.m[i] + sb * b.m[i];
  return r;
}
static M3 scale(const M3& a, double s) {
  M3 r;
  for (int i = 0; i < 9; i++) r.m[i] = a.m[i] * s;
  return r;
}

struct MatX {  // row-major dynamic matrix
  int r = 0, c = 0;
  std::vector<double> d;
  MatX() {}
  MatX(int r_, int c_, double v = 0.0) : r(r_), c(c_), d((size_t)r_ * c_, v) {}
  double& operator()(int i, int j) { return d[(size_t)i * c + j]; }
  double operator()(int i, int j) const { return d[(size_t)i * c + j]; }
  double* row(int i) { return d.data() + (size_t)i * c; }
  const double* row(int i) const { return d.data() + (size_t)i * c; }
};
using VecX = std::vector<double>;

static MatX matmul(const MatX& A, const MatX& B) {
  MatX C(A.r, B.c);
  for (int i = 0; i < A.r; i++) {
    double* ci = C.row(i);
    for (int k = 0; k < A.c; k++) {
      const double a = A(i, k);
      const double* bk = B.row(k);
      for (int j = 0; j < B.c; j++) ci[j] += a * bk[j];
    }
  }
  return C;
}
static VecX matvec(const MatX& A, const VecX& x) {
  VecX y(A.r, 0.0);
  for (int i = 0; i < A.r; i++) {
    const double* ai = A.row(i);
    double s = 0.0;
    for (int j = 0; j < A.c; j++) s += ai[j] * x[j];
    y[i] = s;
  }
  return y;
}
// A^T A (symmetric), row-major accumulation over rows of A
static MatX gram(const MatX& A) {
  const int n = A.c;
  MatX G(n, n);
  for (int k = 0; k < A.r; k++) {
    const double* a = A.row(k);
    for (int i = 0; i < n; i++) {
      const double ai = a[i];
      if (ai == 0.0) continue;
      double* gi = G.row(i);
      for (int j = i; j < n; j++) gi[j] += ai * a[j];
    }
  }
  for (int i = 0; i < n; i++)
    for (int j = 0; j < i; j++) G(i, j) = G(j, i);
  return G;
}
// general inverse by partial-pivot LU (Eigen MatrixXd::inverse(), PRE:837)
static MatX inverse_lu(const MatX& Ain) {
  const int n = Ain.r;
  MatX A = Ain, Inv(n, n);
  std::vector<int> perm(n);
  std::iota(perm.begin(), perm.end(), 0);
  for (int k = 0; k < n; k++) {
    int piv = k;
    double best = std::fabs(A(k, k));
    for (int i = k + 1; i < n; i++)
      if (std::fabs(A(i, k)) > best) {
        best = std::fabs(A(i, k));
        piv = i;
      }
    if (piv != k) {
      for (int j = 0; j < n; j++) std::swap(A(k, j), A(piv, j));
      std::swap(perm[k], perm[piv]);
    }
    const double d = A(k, k);
    for (int i = k + 1; i < n; i++) {
      const double l = A(i, k) / d;
      A(i, k) = l;
      if (l != 0.0) {
        double* ai = A.row(i);
        const double* ak = A.row(k);
        for (int j = k + 1; j < n; j++) ai[j] -= l * ak[j];
      }
    }
  }
  // solve A X = P I column by column (row-major friendly: solve for X^T rows)
  std::vector<double> y(n);
  for (int col = 0; col < n; col++) {
    for (int i = 0; i < n; i++) {
      double s = (perm[i] == col) ? 1.0 : 0.0;
      const double* ai = A.row(i);
      for (int j = 0; j < i; j++) s -= ai[j] * y[j];
      y[i] = s;
    }
    for (int i = n - 1; i >= 0; i--) {
      double s = y[i];
      const double* ai = A.row(i);
      for (int j = i + 1; j < n; j++) s -= ai[j] * y[j];
      y[i] = s / ai[i];
    }
    for (int i = 0; i < n; i++) Inv(i, col) = y[i];
  }
  return Inv;
}
// Cholesky A = L L^T (lower), returns false when not positive definite
static bool cholesky(const MatX& A, MatX& L) {
  const int n = A.r;
  L = MatX(n, n);
  for (int j = 0; j < n; j++) {
    double s = A(j, j);
    const double* lj = L.row(j);
    for (int k = 0; k < j; k++) s -= lj[k] * lj[k];
    if (!(s > 0.0)) return false;
    const double d = std::sqrt(s);
    L(j, j) = d;
    for (int i = j + 1; i < n; i++) {
      double t = A(i, j);
      const double* li = L.row(i);
      for (int k = 0; k < j; k++) t -= li[k] * lj[k];
      L(i, j) = t / d;
    }
  }
  return true;
}
static VecX chol_solve(const MatX& L, const VecX& b) {
  const int n = L.r;
  VecX y(n);
  for (int i = 0; i < n; i++) {
    double s = b[i];
    const double* li = L.row(i);
    for (int k = 0; k < i; k++) s -= li[k] * y[k];
    y[i] = s / li[i];
  }
  for (int i = n - 1; i >= 0; i--) {
    double s = y[i];
    for (int k = i + 1; k < n; k++) s -= L(k, i) * y[k];
    y[i] = s / L(i, i);
  }
  return y;
}

// ------------------------------------------------------------------------------------------------ SO(3)  (MATH:11-99)
static const double kExpNormTolerance = 1e-14;
static const double kNumDtJacobianDelta = 0.01;
static const double kNumGyrBiasJacobianDelta = 0.0001;

// expMap, MATH:55-58: AngleAxisd(|v|, v.normalized()).toRotationMatrix()
static M3 expMap(const V3& v) {
  const double n2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
  const double angle = std::sqrt(n2);
  V3 ax = v;
  if (n2 > 0.0) ax = (1.0 / angle) * v;  // Eigen normalized(): zero vector stays zero
  const double s = std::sin(angle), c = std::cos(angle);
  const V3 sin_axis = s * ax, cos1_axis = (1.0 - c) * ax;
  M3 R;
  double tmp;
  tmp = cos1_axis[0] * ax[1];
  R(0, 1) = tmp - sin_axis[2];
  R(1, 0) = tmp + sin_axis[2];
  tmp = cos1_axis[0] * ax[2];
  R(0, 2) = tmp + sin_axis[1];
  R(2, 0) = tmp - sin_axis[1];
  tmp = cos1_axis[1] * ax[2];
  R(1, 2) = tmp - sin_axis[0];
  R(2, 1) = tmp + sin_axis[0];
  R(0, 0) = cos1_axis[0] * ax[0] + c;
  R(1, 1) = cos1_axis[1] * ax[1] + c;
  R(2, 2) = cos1_axis[2] * ax[2] + c;
  return R;
}

// logMap, MATH:48-51: AngleAxisd(R) (matrix -> quaternion -> angle axis), angle in [0, pi]
static V3 logMap(const M3& m) {
  double q[4];  // x y z w
  double t = m(0, 0) + m(1, 1) + m(2, 2);
  if (t > 0.0) {
    t = std::sqrt(t + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (m(2, 1) - m(1, 2)) * t;
    q[1] = (m(0, 2) - m(2, 0)) * t;
    q[2] = (m(1, 0) - m(0, 1)) * t;
  } else {
    int i = 0;
    if (m(1, 1) > m(0, 0)) i = 1;
    if (m(2, 2) > m(i, i)) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(m(i, i) - m(j, j) - m(k, k) + 1.0);
    q[i] = 0.5 * t;
    t = 0.5 / t;
    q[3] = (m(k, j) - m(j, k)) * t;
    q[j] = (m(j, i) + m(i, j)) * t;
    q[k] = (m(k, i) + m(i, k)) * t;
  }
  double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
  if (n != 0.0) {
    const double angle = 2.0 * std::atan2(n, std::fabs(q[3]));
    if (q[3] < 0) n = -n;
    return {angle * q[0] / n, angle * q[1] / n, angle * q[2] / n};
  }
  return {0.0, 0.0, 0.0};
}

// jacobianRighthandSO3, MATH:63-80
static M3 jacobianRighthandSO3(const V3& v) {
  M3 out = I3();
  const double n = norm(v);
  if (n > kExpNormTolerance) {
    const M3 S = skew(v);
    out = add(add(out, mul(S, S), (n - std::sin(n)) / (n * n * n)), S, -((1.0 - std::cos(n)) / (n * n)));
  }
  return out;
}
// inverseJacobianRighthandSO3, MATH:83-99
static M3 inverseJacobianRighthandSO3(const V3& v) {
  M3 out = I3();
  const double n = norm(v);
  if (n > kExpNormTolerance) {
    const M3 S = skew(v);
    out = add(add(out, S, 0.5), mul(S, S), (1.0 / (n * n)) - ((1 + std::cos(n)) / (2.0 * n * std::sin(n))));
  }
  return out;
}
// addN2Pi / getClosest, MATH:385-412
static V3 addN2Pi(const V3& r, int n) {
  const double nr = norm(r);
  if (nr != 0) return (2.0 * M_PI * n + nr) * ((1.0 / nr) * r);
  return r;
}

// ------------------------------------------------------------------------------------------------ SE kernels (MATH:102-141, 378-382)
static const double kSqrt2 = std::sqrt(2.0);
static const double kSqrtPi = std::sqrt(M_PI);

static MatX seKernel(const VecX& x1, const VecX& x2, double l2, double sf2) {  // MATH:102-110
  MatX K((int)x1.size(), (int)x2.size());
  for (size_t i = 0; i < x1.size(); i++)
    for (size_t j = 0; j < x2.size(); j++) {
      const double d = x1[i] - x2[j];
      K((int)i, (int)j) = std::exp((d * d) * (-0.5 / l2)) * sf2;
    }
  return K;
}
static MatX seKernelIntegral(double a, const VecX& b, const VecX& x2, double l2, double sf2) {  // MATH:114-126
  const double sqrt_inv_l2 = std::sqrt(1.0 / l2);
  const double alpha = kSqrt2 * sf2 * kSqrtPi / (2.0 * sqrt_inv_l2);
  MatX A((int)b.size(), (int)x2.size());
  for (size_t j = 0; j < x2.size(); j++) {
    const double c = std::erf(kSqrt2 * (-x2[j] + a) * sqrt_inv_l2 / 2.0);
    for (size_t i = 0; i < b.size(); i++) A((int)i, (int)j) = alpha * (std::erf(kSqrt2 * (b[i] - x2[j]) * sqrt_inv_l2 / 2.0) - c);
  }
  return A;
}
static MatX seKernelIntegralDt(double a, const VecX& b, const VecX& x2, double l2, double sf2) {  // MATH:130-141
  MatX A((int)b.size(), (int)x2.size());
  for (size_t j = 0; j < x2.size(); j++) {
    const double c = sf2 * std::exp(((x2[j] - a) * (x2[j] - a)) / (-2.0 * l2));
    for (size_t i = 0; i < b.size(); i++) A((int)i, (int)j) = sf2 * std::exp(std::pow(b[i] - x2[j], 2) / (-2.0 * l2)) - c;
  }
  return A;
}
static double kssInt(double a, double b, double l2, double sf2) {  // MATH:378-382
  return 2.0 * l2 * sf2 * std::exp(-std::pow(a - b, 2) / (2.0 * l2)) - 2.0 * l2 * sf2 +
         (std::sqrt(2.0) * sf2 * std::sqrt(M_PI) * std::erf((std::sqrt(2.0) * (a - b) * std::sqrt(1.0 / l2)) / 2.0) * (a - b)) / std::sqrt(1.0 / l2);
}

// ------------------------------------------------------------------------------------------------ types (TYPES:67-298)
struct DataSample {
  double t;
  double data[3];
};
struct GyroVelData {
  double t_offset = 0.0;
  std::vector<DataSample> vel, gyr;
  double vel_var = 0, gyr_var = 0;
  // TYPES:141-223: samples with from < t < to
  static std::vector<DataSample> slice(const std::vector<DataSample>& s, double from, double to) {
    std::vector<DataSample> out;
    if (from >= to || s.empty()) return out;
    size_t i = 0;
    bool loop = true;
    while (loop) {
      if (s[i].t > from) {
        if (s[i].t < to)
          out.push_back(s[i]);
        else
          loop = false;
      }
      if (i < s.size() - 1)
        i++;
      else
        loop = false;
    }
    return out;
  }
  GyroVelData get(double from, double to) const {
    if (!(from <= to)) throw std::invalid_argument("The argument of GyroVelData::Get are not consistent");
    GyroVelData o;
    o.t_offset = t_offset;
    o.vel_var = vel_var;
    o.gyr_var = gyr_var;
    o.vel = slice(vel, from, to);
    o.gyr = slice(gyr, from, to);
    return o;
  }
};
struct PreintMeas {  // TYPES:236-281
  M3 delta_R = I3();
  V3 delta_p = {0, 0, 0};
  double dt = 0, dt_sq_half = 0;
  double cov[36] = {0};
  M3 d_delta_R_d_bw = M3{{0}};
  V3 d_delta_R_d_t = {0, 0, 0};
  M3 d_delta_p_d_bw = M3{{0}};
  M3 d_delta_p_d_bv = M3{{0}};
  V3 d_delta_p_d_t = {0, 0, 0};
};
struct PreintPrior {
  double vel_bias[3] = {0, 0, 0};
  double gyr_bias[3] = {0, 0, 0};
};
struct GPSeHyper {
  double l2, sf2, sz2, mean;
};

// SortIndexTracker2, TYPES:332-458
struct Tracker {
  std::vector<std::vector<double>> data;
  std::vector<std::pair<int, int>> index_map;
  explicit Tracker(const std::vector<std::vector<double>>& d) : data(d) {
    std::vector<double> flat;
    std::vector<std::pair<int, int>> tmp;
    for (size_t i = 0; i < data.size(); i++)
      for (size_t j = 0; j < data[i].size(); j++) {
        flat.push_back(data[i][j]);
        tmp.emplace_back((int)i, (int)j);
      }
    std::vector<int> idx(flat.size());
    std::iota(idx.begin(), idx.end(), 0);
    std::sort(idx.begin(), idx.end(), [&flat](int a, int b) { return flat[a] < flat[b]; });
    for (int i : idx) index_map.push_back(tmp[i]);
  }
  int size() const { return (int)index_map.size(); }
  double get(int i) const { return data[index_map[i].first][index_map[i].second]; }
  double back() const { return get(size() - 1); }
  int getIndex(int a, int b) const {
    for (int k = 0; k < size(); k++)
      if (index_map[k].first == a && index_map[k].second == b) return k;
    return -1;
  }
  double getSmallestGap() const {  // TYPES:442-450: returns the LAST gap
    double diff = get(1) - get(0);
    for (int i = 1; i < size() - 1; i++) diff = get(i + 1) - get(i);
    return diff;
  }
};

// linearInterpolation, MATH:487-532 (value only; the variance is the constant `var`)
static VecX linearInterpolation(const VecX& data, const VecX& time, const Tracker& infer_t) {
  VecX out(infer_t.size());
  if (time.size() < 2) throw std::range_error("InterpolateLinear: this function need at least 2 data points to interpolate");
  int ptr = 0;
  double alpha = (data[1] - data[0]) / (time[1] - time[0]);
  double beta = data[0] - (alpha * time[0]);
  const int nt = (int)time.size();
  for (int i = 0; i < infer_t.size(); ++i) {
    const double ti = infer_t.get(i);
    if (ti > time[0]) {
      bool loop = true;
      while (loop) {
        if (ptr != (nt - 2)) {
          if ((ti <= time[ptr + 1]) && (ti > time[ptr])) {
            loop = false;
          } else {
            ptr++;
            alpha = (data[ptr + 1] - data[ptr]) / (time[ptr + 1] - time[ptr]);
            beta = data[ptr] - (alpha * time[ptr]);
          }
        } else {
          loop = false;
        }
      }
    }
    out[i] = alpha * ti + beta;
  }
  return out;
}

// ------------------------------------------------------------------------------------------------ LPM: IterativeIntegrator (PRE:170-742)
struct IterativeIntegrator {
  double start_t_;
  bool bare_;
  int start_index_ = 0;
  int nb_gyr_, nb_vel_;
  double gyr_var_, vel_var_;
  std::vector<VecX> gyr_data_;  // [3][nb_gyr]
  std::vector<VecX> vel_data_;  // [3][nb_vel]
  VecX gyr_time_, vel_time_;
  std::vector<std::vector<PreintMeas>> preint_;

  IterativeIntegrator(const GyroVelData& imu, double start_time, const PreintPrior& prior, const std::vector<std::vector<double>>& time, double min_freq, bool bare, bool rot_only) {
    start_t_ = start_time;
    bare_ = bare;
    const int nb_infer_vec = (int)time.size();
    nb_gyr_ = (int)imu.gyr.size();
    nb_vel_ = (int)imu.vel.size();
    gyr_var_ = imu.gyr_var;
    vel_var_ = imu.vel_var;
    gyr_data_.assign(3, VecX(nb_gyr_));
    vel_data_.assign(3, VecX(nb_vel_));
    gyr_time_.resize(nb_gyr_);
    vel_time_.resize(nb_vel_);
    for (int i = 0; i < nb_gyr_; i++) {  // PRE:196-202
      for (int a = 0; a < 3; a++) gyr_data_[a][i] = imu.gyr[i].data[a] - prior.gyr_bias[a];
      gyr_time_[i] = imu.gyr[i].t;
    }
    for (int i = 0; i < nb_vel_; i++) {  // PRE:203-209
      for (int a = 0; a < 3; a++) vel_data_[a][i] = imu.vel[i].data[a] - prior.vel_bias[a];
      vel_time_[i] = imu.vel[i].t;
    }
    // PRE:214-225
    std::vector<std::vector<double>> infer_t = time;
    infer_t.push_back({start_t_, start_t_ + kNumDtJacobianDelta});
    infer_t.push_back(vel_time_);
    Tracker t(infer_t);
    if (t.getSmallestGap() > (1.0 / min_freq)) {  // PRE:228-237
      std::vector<double> fake;
      const int nb_fake = (int)std::floor((t.back() - t.get(0)) * min_freq);
      const double offset = t.get(0);
      const double quantum = (t.back() - t.get(0)) / ((double)nb_fake);
      for (int i = 0; i < nb_fake; i++) fake.push_back(offset + (i * quantum));
      infer_t.push_back(fake);
      t = Tracker(infer_t);
    }
    start_index_ = t.getIndex(nb_infer_vec, 0);  // PRE:239
    std::vector<char> interest(t.size(), 1);     // PRE:242-250: everything but the fake stamps
    if ((int)infer_t.size() > nb_infer_vec + 2)
      for (int k = 0; k < t.size(); k++)
        if (t.index_map[k].first == nb_infer_vec + 2) interest[k] = 0;

    std::vector<PreintMeas> preint = rotPreint(t, interest);  // PRE:254
    for (int i = 0; i < nb_infer_vec; i++) {                  // PRE:257-261
      std::vector<PreintMeas> v;
      for (int k = 0; k < t.size(); k++)
        if (t.index_map[k].first == i) v.push_back(preint[k]);
      preint_.push_back(v);
    }
    if (rot_only) return;  // PRE:263-266

    const M3 delta_R_dt_start = preint[t.getIndex(nb_infer_vec, 1)].delta_R;  // PRE:268
    std::vector<PreintMeas> vel_time_preint;                                     // PRE:272
    for (int k = 0; k < t.size(); k++)
      if (t.index_map[k].first == nb_infer_vec + 1) vel_time_preint.push_back(preint[k]);
    std::vector<std::vector<double>> query(infer_t.begin(), infer_t.begin() + nb_infer_vec);  // PRE:279-282
    if (!bare_) {
      std::vector<MatX> d_vel_d_bv, d_vel_d_bw;
      std::vector<VecX> d_vel_d_dt;
      reprojectVelDataFull(vel_time_preint, delta_R_dt_start, d_vel_d_bv, d_vel_d_bw, d_vel_d_dt);  // PRE:278
      posePreintLPM(query, d_vel_d_bv, d_vel_d_bw, d_vel_d_dt);                                     // PRE:283
    } else {
      for (int i = 0; i < nb_vel_; i++) {  // MATH:415-426
        const V3 v = mul(vel_time_preint[i].delta_R, V3{vel_data_[0][i], vel_data_[1][i], vel_data_[2][i]});
        for (int a = 0; a < 3; a++) vel_data_[a][i] = v[a];
      }
      Tracker tq(query);
      posePreintLPMPartial(tq);  // PRE:292-293
    }
  }

  const PreintMeas& get(int a, int b) const { return preint_[a][b]; }

  // rotIterativeIntegration with covariance, PRE:407-487
  void rotIntegrateCov(const std::vector<VecX>& w, const Tracker& t, std::vector<PreintMeas>& out) const {
    M3 rot = I3();
    double cov[36] = {0};
    auto store_cov = [&](PreintMeas& o) {  // minCovDiag, PRE:393-405
      std::memcpy(o.cov, cov, sizeof(cov));
      for (int i = 0; i < 6; i++)
        if (o.cov[i * 6 + i] < 1e-6) o.cov[i * 6 + i] = 1e-6;
    };
    out[0].delta_R = rot;
    store_cov(out[0]);
    out[0].dt = t.get(0) - start_t_;
    out[0].dt_sq_half = 0.5 * out[0].dt * out[0].dt;
    for (int i = 0; i < t.size() - 1; i++) {
      const double dt = t.get(i + 1) - t.get(i);
      const V3 g = {w[0][i] * dt, w[1][i] * dt, w[2][i] * dt};
      const double gn = norm(g);
      M3 e_R = I3(), j_r = I3();
      if (gn > 0.0000000001) {
        const M3 S = skew(g);
        const double s = std::sin(gn), gn2 = gn * gn, sc2 = (1 - std::cos(gn)) / gn2;
        const M3 S2 = mul(S, S);
        e_R = add(add(e_R, S, s / gn), S2, sc2);
        j_r = add(add(j_r, S, -sc2), S2, (gn - s) / (gn2 * gn));
      }
      if ((i + 1) > start_index_) {  // PRE:456-466
        const M3 A = tr(e_R), B = scale(j_r, dt);
        M3 C;
        for (int a = 0; a < 3; a++)
          for (int b = 0; b < 3; b++) C(a, b) = cov[a * 6 + b];
        M3 imu = M3{{gyr_var_, 0, 0, 0, gyr_var_, 0, 0, 0, gyr_var_}};
        const M3 N = add(mul(mul(A, C), tr(A)), mul(mul(B, imu), tr(B)));
        for (int a = 0; a < 3; a++)
          for (int b = 0; b < 3; b++) cov[a * 6 + b] = N(a, b);
      }
      rot = mul(rot, e_R);
      out[i + 1].delta_R = rot;
      store_cov(out[i + 1]);
      out[i + 1].dt = t.get(i + 1) - start_t_;
      out[i + 1].dt_sq_half = out[i + 1].dt * out[i + 1].dt * 0.5;
      if ((i + 1) == start_index_) {  // PRE:477-485
        const M3 rt = tr(rot);
        for (int j = 0; j < i + 1; j++) out[j].delta_R = mul(rt, out[j].delta_R);
        rot = I3();
        out[start_index_].delta_R = rot;
      }
    }
  }
  // rotIterativeIntegration without covariance, PRE:489-519
  void rotIntegrate(const std::vector<VecX>& w, const Tracker& t, std::vector<M3>& out) const {
    M3 rot = I3();
    out[0] = rot;
    for (int i = 0; i < t.size() - 1; i++) {
      const double dt = t.get(i + 1) - t.get(i);
      const M3 e_R = expMap(V3{w[0][i] * dt, w[1][i] * dt, w[2][i] * dt});
      rot = mul(rot, e_R);
      out[i + 1] = rot;
      if ((i + 1) == start_index_) {
        const M3 rt = tr(rot);
        for (int j = 0; j < i + 1; j++) out[j] = mul(rt, out[j]);
        rot = I3();
        out[start_index_] = rot;
      }
    }
  }

  std::vector<PreintMeas> rotPreint(const Tracker& t, const std::vector<char>& interest) const {  // PRE:321-391
    std::vector<PreintMeas> out(t.size());
    std::vector<VecX> w(3), w_shift(3);
    VecX gyr_time_shifted;
    if (!bare_) {
      gyr_time_shifted = gyr_time_;
      for (auto& v : gyr_time_shifted) v -= kNumDtJacobianDelta;
    }
    for (int a = 0; a < 3; a++) {
      w[a] = linearInterpolation(gyr_data_[a], gyr_time_, t);
      if (!bare_) w_shift[a] = linearInterpolation(gyr_data_[a], gyr_time_shifted, t);
    }
    if (!bare_) {
      rotIntegrateCov(w, t, out);
      std::vector<M3> d_R_dt(t.size());
      rotIntegrate(w_shift, t, d_R_dt);
      for (int j = 0; j < t.size(); j++)
        if (interest[j]) out[j].d_delta_R_d_t = (1.0 / kNumDtJacobianDelta) * logMap(mul(tr(out[j].delta_R), d_R_dt[j]));  // PRE:361
      for (int a = 0; a < 3; a++) {  // PRE:365-379
        std::vector<VecX> wb = w;
        for (auto& v : wb[a]) v += kNumGyrBiasJacobianDelta;
        std::vector<M3> d_R(t.size());
        rotIntegrate(wb, t, d_R);
        for (int j = 0; j < t.size(); j++)
          if (interest[j]) {
            const V3 c = (1.0 / kNumGyrBiasJacobianDelta) * logMap(mul(tr(out[j].delta_R), d_R[j]));
            for (int r = 0; r < 3; r++) out[j].d_delta_R_d_bw(r, a) = c[r];
          }
      }
    } else {
      std::vector<M3> R(t.size());
      rotIntegrate(w, t, R);
      for (int i = 0; i < t.size(); i++) out[i].delta_R = R[i];
    }
    return out;
  }

  // jacobianExpMapZeroM, MATH:212-225 (9x3)
  static MatX jacobianExpMapZeroM(const M3& M) {
    MatX o(9, 3);
    const double rows[9][3] = {{0, 0, 0},
                               {M(2, 0), M(2, 1), M(2, 2)},
                               {-M(1, 0), -M(1, 1), -M(1, 2)},
                               {-M(2, 0), -M(2, 1), -M(2, 2)},
                               {0, 0, 0},
                               {M(0, 0), M(0, 1), M(0, 2)},
                               {M(1, 0), M(1, 1), M(1, 2)},
                               {-M(0, 0), -M(0, 1), -M(0, 2)},
                               {0, 0, 0}};
    for (int i = 0; i < 9; i++)
      for (int j = 0; j < 3; j++) o(i, j) = rows[i][j];
    return o;
  }

  // reprojectVelData with Jacobians, MATH:428-483
  void reprojectVelDataFull(const std::vector<PreintMeas>& pre, const M3& delta_R_dt_start, std::vector<MatX>& d_bv, std::vector<MatX>& d_bw, std::vector<VecX>& d_dt) {
    for (int a = 0; a < 3; a++) {
      d_bv.push_back(MatX(nb_vel_, 3));
      d_bw.push_back(MatX(nb_vel_, 3));
      d_dt.push_back(VecX(nb_vel_));
    }
    for (int i = 0; i < nb_vel_; i++) {
      V3 v = {vel_data_[0][i], vel_data_[1][i], vel_data_[2][i]};
      const M3& R = pre[i].delta_R;
      for (int a = 0; a < 3; a++)
        for (int c = 0; c < 3; c++) d_bv[a](i, c) = R(a, c);
      const MatX dRdbw = jacobianExpMapZeroM(pre[i].d_delta_R_d_bw);
      for (int a = 0; a < 3; a++) {
        double tmp[9];  // MATH:457-468
        for (int q = 0; q < 3; q++)
          for (int c = 0; c < 3; c++) tmp[q * 3 + c] = R(a, c) * v[q];
        for (int c = 0; c < 3; c++) {
          double s = 0.0;
          for (int k = 0; k < 9; k++) s += tmp[k] * dRdbw(k, c);
          d_bw[a](i, c) = s;
        }
      }
      v = mul(R, v);
      const V3 vel_rot_dt = mul(tr(delta_R_dt_start), v);
      const V3 dvdt = (1.0 / kNumDtJacobianDelta) * (vel_rot_dt - v);
      for (int a = 0; a < 3; a++) {
        d_dt[a][i] = dvdt[a];
        vel_data_[a][i] = v[a];
      }
    }
  }

  // posePreintLPMPartial, PRE:669-741
  void posePreintLPMPartial(const Tracker& time) {
    int data_ptr = 0, start_index = 0;
    while (time.get(start_index) < start_t_) {
      start_index++;
      if (start_index == time.size()) throw std::range_error("LPM Partial: the start_time is not in the query domain");
    }
    while (vel_time_[data_ptr + 1] < start_t_) {
      data_ptr++;
      if (data_ptr == (nb_vel_ - 1)) throw std::range_error("LPM Partial: the start_time is not in the data domain");
    }
    for (int axis = 0; axis < 3; ++axis) {
      const VecX& vd = vel_data_[axis];
      int ptr = data_ptr;
      double alpha = (vd[ptr + 1] - vd[ptr]) / (vel_time_[ptr + 1] - vel_time_[ptr]);
      double beta = vd[ptr] - alpha * vel_time_[ptr];
      double t_0 = start_t_, t_1 = vel_time_[ptr + 1];
      double d_0 = alpha * vel_time_[ptr] + beta, d_1 = vd[ptr + 1];
      double d_p_backup = 0;
      for (int i = start_index; i < time.size(); ++i) {
        const double ti = time.get(i);
        if (ti > vel_time_[0]) {
          bool loop = true;
          while (loop) {
            if ((ti >= vel_time_[ptr]) && (ti <= vel_time_[ptr + 1])) {
              loop = false;
            } else if (ptr < (nb_vel_ - 2)) {
              d_p_backup = d_p_backup + ((t_1 - t_0) * (d_0 + d_1) / 2.0);
              ptr++;
              t_0 = vel_time_[ptr];
              t_1 = vel_time_[ptr + 1];
              d_0 = vd[ptr];
              d_1 = vd[ptr + 1];
              alpha = (d_1 - d_0) / (t_1 - t_0);
              beta = d_0 - alpha * t_0;
            } else {
              loop = false;
            }
          }
        }
        const double temp_d_1 = alpha * ti + beta;
        const double temp_d_p = d_p_backup + ((ti - t_0) * (d_0 + temp_d_1) / 2.0);
        const auto idx = time.index_map[i];
        preint_[idx.first][idx.second].delta_p[axis] = temp_d_p;
      }
    }
  }

  // posePreintLPM, PRE:524-667
  void posePreintLPM(const std::vector<std::vector<double>>& tq, const std::vector<MatX>& d_bv, const std::vector<MatX>& d_bw, const std::vector<VecX>& d_dt) {
    Tracker time(tq);
    const std::vector<VecX> save_vel = vel_data_;
    const VecX save_time = vel_time_;
    for (int i = 0; i < nb_vel_; ++i) {  // PRE:537-545
      for (int a = 0; a < 3; a++) vel_data_[a][i] += kNumDtJacobianDelta * d_dt[a][i];
      vel_time_[i] -= kNumDtJacobianDelta;
    }
    posePreintLPMPartial(time);
    vel_data_ = save_vel;
    vel_time_ = save_time;

    int data_ptr = 0, start_index = 0;
    while (time.get(start_index) < start_t_) {
      start_index++;
      if (start_index == time.size()) throw std::range_error("FullLPM: the start_time is not in the query domain");
    }
    while (vel_time_[data_ptr + 1] < start_t_) {
      data_ptr++;
      if (data_ptr == (nb_vel_ - 1)) throw std::range_error("FullLPM: the start_time is not in the data domain");
    }
    for (int axis = 0; axis < 3; ++axis) {
      const VecX& vd = vel_data_[axis];
      auto rowv = [](const MatX& M, int r) { return V3{M(r, 0), M(r, 1), M(r, 2)}; };
      int ptr = data_ptr;
      double alpha = (vd[ptr + 1] - vd[ptr]) / (vel_time_[ptr + 1] - vel_time_[ptr]);
      double beta = vd[ptr] - alpha * vel_time_[ptr];
      double t_0 = start_t_, t_1 = vel_time_[ptr + 1];
      double d_0 = alpha * vel_time_[ptr] + beta, d_1 = vd[ptr + 1];
      double d_p_backup = 0;
      double ratio = (start_t_ - vel_time_[ptr]) / (vel_time_[ptr + 1] - vel_time_[ptr]);
      V3 d_d_0_d_bw = ratio * rowv(d_bw[axis], ptr + 1) + (1 - ratio) * rowv(d_bw[axis], ptr);
      V3 d_d_0_d_bv = ratio * rowv(d_bv[axis], ptr + 1) + (1 - ratio) * rowv(d_bv[axis], ptr);
      V3 d_p_d_bv_backup = {0, 0, 0}, d_p_d_bw_backup = {0, 0, 0};
      for (int i = start_index; i < time.size(); ++i) {
        const double ti = time.get(i);
        if (ti > vel_time_[0]) {
          bool loop = true;
          while (loop) {
            if ((ti >= vel_time_[ptr]) && (ti <= vel_time_[ptr + 1])) {
              loop = false;
            } else if (ptr < (nb_vel_ - 2)) {
              d_p_backup = d_p_backup + ((t_1 - t_0) * (d_0 + d_1) / 2.0);
              const double dt = t_1 - t_0;
              const V3 d_d_1_d_bv = rowv(d_bv[axis], ptr + 1), d_d_1_d_bw = rowv(d_bw[axis], ptr + 1);
              d_p_d_bv_backup = d_p_d_bv_backup + (dt / 2.0) * (d_d_0_d_bv + d_d_1_d_bv);
              d_p_d_bw_backup = d_p_d_bw_backup + (dt / 2.0) * (d_d_0_d_bw + d_d_1_d_bw);
              ptr++;
              t_0 = vel_time_[ptr];
              t_1 = vel_time_[ptr + 1];
              d_0 = vd[ptr];
              d_1 = vd[ptr + 1];
              alpha = (d_1 - d_0) / (t_1 - t_0);
              beta = d_0 - alpha * t_0;
              d_d_0_d_bv = rowv(d_bv[axis], ptr);
              d_d_0_d_bw = rowv(d_bw[axis], ptr);
            } else {
              loop = false;
            }
          }
        }
        const double temp_d_1 = alpha * ti + beta;
        const double temp_d_p = d_p_backup + ((ti - t_0) * (d_0 + temp_d_1) / 2.0);
        const double temp_d_p_var = (ti - start_t_) * vel_var_;
        const auto idx = time.index_map[i];
        PreintMeas& pm = preint_[idx.first][idx.second];
        pm.d_delta_p_d_t[axis] = (pm.delta_p[axis] - temp_d_p) / kNumDtJacobianDelta;  // PRE:646
        pm.delta_p[axis] = temp_d_p;
        pm.cov[(3 + axis) * 6 + 3 + axis] = temp_d_p_var;
        ratio = (ti - vel_time_[ptr]) / (vel_time_[ptr + 1] - vel_time_[ptr]);
        const V3 d_d_1_d_bw = ratio * rowv(d_bw[axis], ptr + 1) + (1 - ratio) * rowv(d_bw[axis], ptr);
        const V3 d_d_1_d_bv = ratio * rowv(d_bv[axis], ptr + 1) + (1 - ratio) * rowv(d_bv[axis], ptr);
        const double dt = ti - t_0;
        const V3 a = d_p_d_bv_backup + (dt / 2.0) * (d_d_0_d_bv + d_d_1_d_bv);
        const V3 b = d_p_d_bw_backup + (dt / 2.0) * (d_d_0_d_bw + d_d_1_d_bw);
        for (int c = 0; c < 3; c++) {
          pm.d_delta_p_d_bv(axis, c) = a[c];
          pm.d_delta_p_d_bw(axis, c) = b[c];
        }
      }
    }
  }
};

// ------------------------------------------------------------------------------------------------ cost functions (COST)
// JacobianRes, COST:73-145: d[ J_r(r) dr ] / d[r, dr]  (3 x 6).  The reference's expression is a symbolic-toolbox dump of the
// derivative of J_r(r) dr with J_r = I - (1-cos n)/n^2 S + (n - sin n)/n^3 S^2; restated here in closed form:
//   d/dr_k = -dA/dr_k (r x dr) - A (e_k x dr) + dB/dr_k (r x (r x dr)) + B (e_k x (r x dr) + r x (e_k x dr)),  d/d(dr) = J_r(r)
// with A = (1-cos n)/n^2, B = (n - sin n)/n^3.
static void JacobianRes(const V3& r, const V3& dr, double out[3][6]) {
  const double n2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
  const double n = std::sqrt(n2);
  auto cross = [](const V3& a, const V3& b) { return V3{a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]}; };
  if (n > kExpNormTolerance) {
    const double s = std::sin(n), c = std::cos(n);
    const double A = (1.0 - c) / n2, B = (n - s) / (n2 * n);
    const double dA = (n * s - 2.0 * (1.0 - c)) / (n2 * n);          // dA/dn
    const double dB = ((1.0 - c) * n - 3.0 * (n - s)) / (n2 * n2);  // dB/dn
    const V3 rxd = cross(r, dr), rxrxd = cross(r, rxd);
    for (int k = 0; k < 3; k++) {
      V3 e = {0, 0, 0};
      e[k] = 1.0;
      const double dn = r[k] / n;
      const V3 exd = cross(e, dr);
      const V3 t = (-dA * dn) * rxd + (-A) * exd + (dB * dn) * rxrxd + B * (cross(e, rxd) + cross(r, exd));
      for (int i = 0; i < 3; i++) out[i][k] = t[i];
    }
    const M3 Jr = jacobianRighthandSO3(r);
    for (int i = 0; i < 3; i++)
      for (int k = 0; k < 3; k++) out[i][3 + k] = Jr(i, k);
  } else {  // COST:137-141
    const M3 S = skew(dr);
    for (int i = 0; i < 3; i++)
      for (int k = 0; k < 3; k++) {
        out[i][k] = 0.5 * S(i, k);
        out[i][3 + k] = (i == k) ? 1.0 : 0.0;
      }
  }
}

// A least-squares problem in the shape Ceres sees it: n_param unknowns, evaluate(x, r, J or null)
struct LsqProblem {
  int n_param = 0, n_res = 0;
  virtual void evaluate(const VecX& x, VecX& r, MatX* J) const = 0;
  virtual ~LsqProblem() {}
};

struct SolveSummary {
  int iterations = 0, successful = 0;
  double initial_cost = 0, final_cost = 0;
  int termination = 0;  // 1 function tol, 2 parameter tol, 3 gradient tol, 4 max iterations, 5 radius
};

// Ceres 2.1 TrustRegionMinimizer + LevenbergMarquardtStrategy + DENSE_NORMAL_CHOLESKY, defaults + PRE:943-948
static SolveSummary ceres_like_solve(const LsqProblem& prob, VecX& x, int max_iter = 50, double function_tolerance = 1e-10) {
  const double gradient_tolerance = 1e-10, parameter_tolerance = 1e-8, min_relative_decrease = 1e-3;
  const double min_diag = 1e-6, max_diag = 1e32, max_radius = 1e16, min_radius = 1e-32;
  const int n = prob.n_param;
  SolveSummary sum;
  VecX r(prob.n_res), r_new(prob.n_res);
  MatX J(prob.n_res, n);
  prob.evaluate(x, r, &J);
  auto half_sq = [](const VecX& v) {
    double s = 0;
    for (double a : v) s += a * a;
    return 0.5 * s;
  };
  double cost = half_sq(r);
  sum.initial_cost = cost;
  const bool lm_trace = std::getenv("UGPMO_LMTRACE") != nullptr;  // per-iteration solver trace on stderr (debugging aid)
  if (lm_trace) std::fprintf(stderr, "[oracle lm] initial cost %.17g\n", cost);
  // Jacobi scaling fixed at the initial point
  VecX scale(n);
  for (int j = 0; j < n; j++) {
    double s = 0;
    for (int i = 0; i < J.r; i++) s += J(i, j) * J(i, j);
    scale[j] = 1.0 / (1.0 + std::sqrt(s));
  }
  auto apply_scale = [&](MatX& Jm) {
    for (int i = 0; i < Jm.r; i++) {
      double* row = Jm.row(i);
      for (int j = 0; j < n; j++) row[j] *= scale[j];
    }
  };
  auto gradient_max = [&](const MatX& Jm, const VecX& res) {  // J here is UNscaled
    double g = 0;
    for (int j = 0; j < n; j++) {
      double s = 0;
      for (int i = 0; i < Jm.r; i++) s += Jm(i, j) * res[i];
      g = std::max(g, std::fabs(s));
    }
    return g;
  };
  if (gradient_max(J, r) <= gradient_tolerance) {
    sum.termination = 3;
    sum.final_cost = cost;
    return sum;
  }
  apply_scale(J);
  double radius = 1e4, decrease_factor = 2.0;
  bool reuse_diagonal = false;
  VecX diag(n);
  double x_norm = 0;
  for (double a : x) x_norm += a * a;
  x_norm = std::sqrt(x_norm);
  MatX JtJ = gram(J);
  VecX Jtr(n);
  auto compute_Jtr = [&]() {
    std::fill(Jtr.begin(), Jtr.end(), 0.0);
    for (int i = 0; i < J.r; i++) {
      const double ri = r[i];
      const double* row = J.row(i);
      for (int j = 0; j < n; j++) Jtr[j] += row[j] * ri;
    }
  };
  compute_Jtr();
  int iter = 0;
  while (true) {
    if (iter >= max_iter) {
      sum.termination = 4;
      break;
    }
    if (radius < min_radius) {
      sum.termination = 5;
      break;
    }
    iter++;
    if (!reuse_diagonal)
      for (int j = 0; j < n; j++) diag[j] = std::min(std::max(JtJ(j, j), min_diag), max_diag);
    MatX lhs = JtJ;
    for (int j = 0; j < n; j++) lhs(j, j) += diag[j] / radius;
    MatX L;
    bool valid = cholesky(lhs, L);
    VecX step;
    double model_cost_change = 0;
    if (valid) {
      step = chol_solve(L, Jtr);
      for (double& s : step) s = -s;
      for (double s : step)
        if (!std::isfinite(s)) valid = false;
    }
    if (valid) {
      VecX mr(J.r, 0.0);
      for (int i = 0; i < J.r; i++) {
        const double* row = J.row(i);
        double s = 0;
        for (int j = 0; j < n; j++) s += row[j] * step[j];
        mr[i] = s;
      }
      for (int i = 0; i < J.r; i++) model_cost_change -= mr[i] * (r[i] + mr[i] / 2.0);
      if (!(model_cost_change > 0.0)) valid = false;
    }
    if (!valid) {  // StepIsInvalid
      radius /= decrease_factor;
      decrease_factor *= 2.0;
      reuse_diagonal = true;
      continue;
    }
    VecX x_new(n);
    double step_norm = 0;
    for (int j = 0; j < n; j++) {
      const double dj = step[j] * scale[j];
      x_new[j] = x[j] + dj;
      step_norm += dj * dj;
    }
    step_norm = std::sqrt(step_norm);
    prob.evaluate(x_new, r_new, nullptr);
    const double cost_new = half_sq(r_new);
    if (step_norm <= parameter_tolerance * (x_norm + parameter_tolerance)) {
      sum.termination = 2;
      break;
    }
    const double cost_change = cost - cost_new;
    if (std::fabs(cost_change) <= function_tolerance * cost) {
      sum.termination = 1;
      // Ceres declares convergence without taking the step when it is not successful; a successful one is not applied either
      break;
    }
    const double rho = cost_change / model_cost_change;
    if (lm_trace) std::fprintf(stderr, "[oracle lm] iter %d cost %.17g cost_new %.17g radius %.6g mcc %.17g step_norm2 %.6g x_norm %.17g rho %.6g\n", iter, cost, cost_new, radius, model_cost_change, step_norm * step_norm, x_norm, rho);
    if (rho > min_relative_decrease) {
      x = x_new;
      x_norm = 0;
      for (double a : x) x_norm += a * a;
      x_norm = std::sqrt(x_norm);
      cost = cost_new;
      prob.evaluate(x, r, &J);
      sum.successful++;
      if (gradient_max(J, r) <= gradient_tolerance) {
        sum.termination = 3;
        break;
      }
      apply_scale(J);
      JtJ = gram(J);
      compute_Jtr();
      radius = std::min(max_radius, radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * rho - 1.0, 3)));
      decrease_factor = 2.0;
      reuse_diagonal = false;
    } else {
      radius /= decrease_factor;
      decrease_factor *= 2.0;
      reuse_diagonal = true;
    }
  }
  sum.iterations = iter;
  sum.final_cost = cost;
  return sum;
}

// ------------------------------------------------------------------------------------------------ Se3Integrator (PRE:747-1494)
struct Se3Integrator {
  std::vector<GPSeHyper> hyper_;
  bool correlate_;
  double state_freq_;
  int nb_overlap_;
  double start_t_;
  int nb_gyr_ = 0, nb_vel_ = 0, nb_state_ = 0;
  std::vector<VecX> gyr_data_, vel_data_;  // [3][n]
  VecX gyr_time_, vel_time_, state_time_;
  std::vector<VecX> state_d_r_, state_vel_;  // [3][S] (columns of the reference's S x 3 matrices)
  std::vector<MatX> K_inv_, KK_inv_, K_int_K_inv_;
  std::vector<VecX> alpha_;
  VecX state_var_;
  MatX state_cor_;
  // Jacobian bookkeeping (PRE:1165-1175)
  std::vector<MatX> d_state_bw_;  // [3] S x 3
  std::vector<VecX> d_d_r_dt_;    // [3] S
  std::vector<V3> d_r_dt_local_, d_r_dt_local_shift_, delta_r_time_, state_r_temp_;
  std::vector<std::vector<V3>> delta_r_bw_, d_r_bw_local_shift_;  // [axis][S]
  std::vector<MatX> d_vel_bv_, d_vel_bw_;
  std::vector<VecX> d_vel_dt_;
  SolveSummary sum_rot, sum_vel;

  // GpNormCostFunction residual/Jacobian helper: r = ((KKinv - I) s) .* w   (COST:14-70)
  struct GpNorm {
    // What Ceres sees as the Jacobian: the reference builds jacobian_ = (KKinv - I) with COLUMN i scaled by w[i] (COST:36-42)
    // and hands it over through a column-major Eigen::Map (COST:63-64) while Ceres reads row-major, i.e. it sees the
    // transpose: J(r, c) = (KKinv - I)(c, r) * w[r]  (== diag(w) (KKinv - I) up to the rounding asymmetry of K K^-1).
    MatX Jm;
    VecX w;
    MatX KKinv;
    void init(const MatX& KK, const VecX& var) {
      const int n = KK.c;
      KKinv = KK;
      w.resize(n);
      for (int i = 0; i < n; i++) {
        w[i] = std::sqrt(1.0 / var[i]);
        if (std::isnan(w[i])) w[i] = 1.0;  // COST:40
      }
      Jm = MatX(n, n);
      for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) Jm(i, j) = (KK(j, i) - (i == j ? 1.0 : 0.0)) * w[i];
    }
    void residual(const VecX& s, double* r) const {  // COST:55-57: ((KKinv s - s) .* w) -- row-wise weight
      const int n = KKinv.c;
      for (int i = 0; i < n; i++) {
        const double* row = KKinv.row(i);
        double t = 0;
        for (int j = 0; j < n; j++) t += row[j] * s[j];
        r[i] = (t - s[i]) * w[i];
      }
    }
  };

  // RotCostFunction, COST:149-254
  struct RotCost {
    const Se3Integrator* p;
    std::vector<MatX> KsKinv, KsIntKinv;
    VecX d_time;
    V3 mean;
    int nd, ns;
    void init(const Se3Integrator* parent) {
      p = parent;
      nd = p->nb_gyr_;
      ns = p->nb_state_;
      d_time.resize(nd);
      for (int i = 0; i < nd; i++) d_time[i] = p->gyr_time_[i] - p->start_t_;
      KsKinv.resize(3);
      KsIntKinv.resize(3);
      for (int i = 0; i < 3; i++) {
        const MatX ks_int = seKernelIntegral(p->start_t_, p->gyr_time_, p->state_time_, p->hyper_[i].l2, p->hyper_[i].sf2);
        const MatX ks = seKernel(p->gyr_time_, p->state_time_, p->hyper_[i].l2, p->hyper_[i].sf2);
        KsKinv[i] = matmul(ks, p->K_inv_[i]);
        KsIntKinv[i] = matmul(ks_int, p->K_inv_[i]);
        mean[i] = p->hyper_[i].mean;
      }
    }
    // residuals r (3*nd, sample-major) and optional Jacobian blocks Jc[c] (3*nd x ns)
    void evaluate(const VecX s[3], double* r, MatX* Jc[3]) const {
      std::vector<VecX> dr(3), rot(3);
      for (int c = 0; c < 3; c++) {
        dr[c] = matvec(KsKinv[c], s[c]);
        rot[c] = matvec(KsIntKinv[c], s[c]);
      }
      for (int i = 0; i < nd; i++) {
        const V3 rv = {rot[0][i] + d_time[i] * mean[0], rot[1][i] + d_time[i] * mean[1], rot[2][i] + d_time[i] * mean[2]};
        const V3 dv = {dr[0][i] + mean[0], dr[1][i] + mean[1], dr[2][i] + mean[2]};
        const V3 t = mul(jacobianRighthandSO3(rv), dv);
        for (int a = 0; a < 3; a++) r[i * 3 + a] = t[a] - p->gyr_data_[a][i];  // NOT weighted (COST:250)
        if (Jc) {
          double D[3][6];
          JacobianRes(rv, dv, D);
          for (int c = 0; c < 3; c++) {
            if (!Jc[c]) continue;
            const double* ki = KsIntKinv[c].row(i);
            const double* ks = KsKinv[c].row(i);
            for (int a = 0; a < 3; a++) {
              double* out = Jc[c]->row(i * 3 + a);
              const double d0 = D[a][c], d1 = D[a][c + 3];
              for (int j = 0; j < ns; j++) out[j] = d0 * ki[j] + d1 * ks[j];
            }
          }
        }
      }
    }
  };

  // VelCostFunction, COST:257-385
  struct VelCost {
    const Se3Integrator* p;
    std::vector<MatX> KvelKinv, KgyrIntKinv;
    VecX d_time;
    V3 mean_vel, mean_dr;
    int nd, ns;
    double wgt;
    void init(const Se3Integrator* parent, double vel_var) {
      p = parent;
      nd = p->nb_vel_;
      ns = p->nb_state_;
      wgt = std::sqrt(1.0 / vel_var);
      d_time.resize(nd);
      for (int i = 0; i < nd; i++) d_time[i] = p->vel_time_[i] - p->start_t_;
      KvelKinv.resize(3);
      KgyrIntKinv.resize(3);
      for (int i = 0; i < 3; i++) {
        KgyrIntKinv[i] = matmul(seKernelIntegral(p->start_t_, p->vel_time_, p->state_time_, p->hyper_[i].l2, p->hyper_[i].sf2), p->K_inv_[i]);
        mean_dr[i] = p->hyper_[i].mean;
        KvelKinv[i] = matmul(seKernel(p->vel_time_, p->state_time_, p->hyper_[i + 3].l2, p->hyper_[i + 3].sf2), p->K_inv_[i + 3]);
        mean_vel[i] = p->hyper_[i + 3].mean;
      }
    }
    // Jr[c]: d/d(rot state c), Jv[c]: d/d(vel state c)  (3*nd x ns each), any may be null
    void evaluate(const VecX sr[3], const VecX sv[3], double* r, MatX* Jr[3], MatX* Jv[3]) const {
      std::vector<VecX> rot(3), vel(3);
      for (int c = 0; c < 3; c++) {
        rot[c] = matvec(KgyrIntKinv[c], sr[c]);
        vel[c] = matvec(KvelKinv[c], sv[c]);
      }
      for (int i = 0; i < nd; i++) {
        const V3 rv = {rot[0][i] + d_time[i] * mean_dr[0], rot[1][i] + d_time[i] * mean_dr[1], rot[2][i] + d_time[i] * mean_dr[2]};
        const M3 R_T = expMap(V3{-rv[0], -rv[1], -rv[2]});
        const V3 vv = {vel[0][i] + mean_vel[0], vel[1][i] + mean_vel[1], vel[2][i] + mean_vel[2]};
        const V3 t = mul(R_T, vv);
        for (int a = 0; a < 3; a++) r[i * 3 + a] = (t[a] - p->vel_data_[a][i]) * wgt;  // COST:381
        if (Jr || Jv) {
          M3 d_res_d_r = M3{{0}};
          if (Jr) d_res_d_r = mul(skew(t), jacobianRighthandSO3(rv));  // COST:362
          for (int c = 0; c < 3; c++) {
            if (Jr && Jr[c]) {
              const double* k = KgyrIntKinv[c].row(i);
              for (int a = 0; a < 3; a++) {
                double* out = Jr[c]->row(i * 3 + a);
                const double f = wgt * d_res_d_r(a, c);
                for (int j = 0; j < ns; j++) out[j] = f * k[j];
              }
            }
            if (Jv && Jv[c]) {
              const double* k = KvelKinv[c].row(i);
              for (int a = 0; a < 3; a++) {
                double* out = Jv[c]->row(i * 3 + a);
                const double f = wgt * R_T(a, c);
                for (int j = 0; j < ns; j++) out[j] = f * k[j];
              }
            }
          }
        }
      }
    }
  };

  GpNorm gp_[6];
  RotCost rot_cost_;
  VelCost vel_cost_;

  // problem #1 (PRE:872-952): unknowns = 3 rot channels; residuals = 3 GpNorm + RotCost
  struct RotProblem : LsqProblem {
    const Se3Integrator* p;
    void evaluate(const VecX& x, VecX& r, MatX* J) const override {
      const int S = p->nb_state_, nd = p->nb_gyr_;
      VecX s[3];
      for (int c = 0; c < 3; c++) s[c].assign(x.begin() + c * S, x.begin() + (c + 1) * S);
      if (J) std::fill(J->d.begin(), J->d.end(), 0.0);
      for (int c = 0; c < 3; c++) {
        p->gp_[c].residual(s[c], r.data() + c * S);
        if (J)
          for (int i = 0; i < S; i++) std::memcpy(J->row(c * S + i) + c * S, p->gp_[c].Jm.row(i), sizeof(double) * S);
      }
      if (J) {
        MatX Jc0(3 * nd, S), Jc1(3 * nd, S), Jc2(3 * nd, S);
        MatX* Jc[3] = {&Jc0, &Jc1, &Jc2};
        p->rot_cost_.evaluate(s, r.data() + 3 * S, Jc);
        for (int c = 0; c < 3; c++)
          for (int i = 0; i < 3 * nd; i++) std::memcpy(J->row(3 * S + i) + c * S, Jc[c]->row(i), sizeof(double) * S);
      } else {
        p->rot_cost_.evaluate(s, r.data() + 3 * S, nullptr);
      }
    }
  };
  // problem #2 (PRE:954-967): rot states constant; unknowns = 3 vel channels; residuals = VelCost + 3 GpNorm (vel)
  struct VelProblem : LsqProblem {
    const Se3Integrator* p;
    void evaluate(const VecX& x, VecX& r, MatX* J) const override {
      const int S = p->nb_state_, nd = p->nb_vel_;
      VecX sv[3];
      for (int c = 0; c < 3; c++) sv[c].assign(x.begin() + c * S, x.begin() + (c + 1) * S);
      if (J) std::fill(J->d.begin(), J->d.end(), 0.0);
      if (J) {
        MatX Jc0(3 * nd, S), Jc1(3 * nd, S), Jc2(3 * nd, S);
        MatX* Jv[3] = {&Jc0, &Jc1, &Jc2};
        p->vel_cost_.evaluate(p->state_d_r_.data(), sv, r.data(), nullptr, Jv);
        for (int c = 0; c < 3; c++)
          for (int i = 0; i < 3 * nd; i++) std::memcpy(J->row(i) + c * S, Jv[c]->row(i), sizeof(double) * S);
      } else {
        p->vel_cost_.evaluate(p->state_d_r_.data(), sv, r.data(), nullptr, nullptr);
      }
      for (int c = 0; c < 3; c++) {
        p->gp_[3 + c].residual(sv[c], r.data() + 3 * nd + c * S);
        if (J)
          for (int i = 0; i < S; i++) std::memcpy(J->row(3 * nd + c * S + i) + c * S, p->gp_[3 + c].Jm.row(i), sizeof(double) * S);
      }
    }
  };

  Se3Integrator(const GyroVelData& imu_data, double start_time, const PreintPrior& prior, double window_duration, double state_freq, int nb_overlap, bool correlate)
      : hyper_(6), correlate_(correlate), state_freq_(state_freq), nb_overlap_(nb_overlap), start_t_(start_time) {
    // PRE:766-771
    const double vel_freq = (imu_data.vel.size() - 1) / (imu_data.vel.back().t - imu_data.vel[0].t);
    const double gyr_freq = (imu_data.gyr.size() - 1) / (imu_data.gyr.back().t - imu_data.gyr[0].t);
    const double imu_freq = std::min(vel_freq, gyr_freq);
    state_freq_ = std::max(state_freq, 5.0 / window_duration);
    state_freq_ = std::min(state_freq_, imu_freq);
    // PRE:775-786
    nb_state_ = (int)(std::ceil(window_duration * state_freq_) + (2 * nb_overlap_));
    state_time_.resize(nb_state_);
    const double t0 = start_t_ - (((double)nb_overlap_) / state_freq_);
    std::vector<double> t_vect(nb_state_), t_vect_dt(nb_state_);
    for (int i = 0; i < nb_state_; i++) {
      state_time_[i] = t0 + ((double)i) / state_freq_;
      t_vect[i] = state_time_[i];
      t_vect_dt[i] = state_time_[i] + kNumDtJacobianDelta;
    }
    GyroVelData data = imu_data.get(t_vect[0], t_vect.back());  // PRE:789
    nb_gyr_ = (int)data.gyr.size();
    nb_vel_ = (int)data.vel.size();
    gyr_data_.assign(3, VecX(nb_gyr_));
    vel_data_.assign(3, VecX(nb_vel_));
    gyr_time_.resize(nb_gyr_);
    vel_time_.resize(nb_vel_);
    for (int i = 0; i < nb_gyr_; i++) {
      for (int a = 0; a < 3; a++) gyr_data_[a][i] = data.gyr[i].data[a] - prior.gyr_bias[a];
      gyr_time_[i] = data.gyr[i].t;
    }
    for (int i = 0; i < nb_vel_; i++) {
      for (int a = 0; a < 3; a++) vel_data_[a][i] = data.vel[i].data[a] - prior.vel_bias[a];
      vel_time_[i] = data.vel[i].t;
    }
    initialiseStateWithLPM(data, t_vect, t_vect_dt, prior);  // PRE:815
    initialiseStateDiff(data, t_vect, t_vect_dt);             // PRE:818
    initialiseHyperParam(data);                               // PRE:821

    // PRE:825-866
    const int S = nb_state_;
    VecX state_std(6 * S, 0.0);
    state_var_.assign(6 * S, 0.0);
    K_inv_.resize(6);
    KK_inv_.resize(6);
    K_int_K_inv_.resize(3);
    std::vector<VecX> var1000(6, VecX(S));
    for (int i = 0; i < 6; i++) {
      const MatX K = seKernel(state_time_, state_time_, hyper_[i].l2, hyper_[i].sf2);
      MatX to_inv = K;
      for (int j = 0; j < S; j++) to_inv(j, j) += hyper_[i].sz2;
      K_inv_[i] = inverse_lu(to_inv);
      KK_inv_[i] = matmul(K, K_inv_[i]);
      if (i < 3) K_int_K_inv_[i] = matmul(seKernelIntegral(start_t_, state_time_, state_time_, hyper_[i].l2, hyper_[i].sf2), K_inv_[i]);
      const MatX KKK = matmul(KK_inv_[i], K);
      for (int j = 0; j < S; j++) {
        double v = -KKK(j, j) + hyper_[i].sf2 + hyper_[i].sz2;
        if (v <= 0) v = hyper_[i].sz2;
        if (correlate) state_std[i * S + j] = std::sqrt(v);
        state_var_[i * S + j] = v;
        var1000[i][j] = 1000.0 * v;
      }
    }
    for (int i = 0; i < 6; i++) gp_[i].init(KK_inv_[i], var1000[i]);
    rot_cost_.init(this);
    vel_cost_.init(this, imu_data.vel_var);

    // PRE:887-940: state correlation from the Jacobians at the LPM-initialised state (a side thread in the reference)
    std::thread cor_thread;
    MatX state_J;
    if (correlate) {
      state_J = MatX(3 * nb_gyr_ + 3 * nb_vel_, 6 * S);
      {
        MatX J0(3 * nb_gyr_, S), J1(3 * nb_gyr_, S), J2(3 * nb_gyr_, S);
        MatX* Jc[3] = {&J0, &J1, &J2};
        VecX res(3 * nb_gyr_);
        rot_cost_.evaluate(state_d_r_.data(), res.data(), Jc);
        for (int c = 0; c < 3; c++)
          for (int i = 0; i < 3 * nb_gyr_; i++) std::memcpy(state_J.row(i) + c * S, Jc[c]->row(i), sizeof(double) * S);
      }
      {
        MatX J0(3 * nb_vel_, S), J1(3 * nb_vel_, S), J2(3 * nb_vel_, S), J3(3 * nb_vel_, S), J4(3 * nb_vel_, S), J5(3 * nb_vel_, S);
        MatX* Jr[3] = {&J0, &J1, &J2};
        MatX* Jv[3] = {&J3, &J4, &J5};
        VecX res(3 * nb_vel_);
        vel_cost_.evaluate(state_d_r_.data(), state_vel_.data(), res.data(), Jr, Jv);
        for (int c = 0; c < 3; c++)
          for (int i = 0; i < 3 * nb_vel_; i++) {
            std::memcpy(state_J.row(3 * nb_gyr_ + i) + c * S, Jr[c]->row(i), sizeof(double) * S);
            std::memcpy(state_J.row(3 * nb_gyr_ + i) + (3 + c) * S, Jv[c]->row(i), sizeof(double) * S);
          }
      }
      cor_thread = std::thread([this, &state_J, state_std]() { computeStateCorr(state_J, state_std); });  // PRE:939
    }

    {  // ceres::Solve #1, PRE:943-952
      RotProblem prob;
      prob.p = this;
      prob.n_param = 3 * S;
      prob.n_res = 3 * S + 3 * nb_gyr_;
      VecX x(3 * S);
      for (int c = 0; c < 3; c++) std::copy(state_d_r_[c].begin(), state_d_r_[c].end(), x.begin() + c * S);
      sum_rot = ceres_like_solve(prob, x);
      for (int c = 0; c < 3; c++) state_d_r_[c].assign(x.begin() + c * S, x.begin() + (c + 1) * S);
    }
    {  // ceres::Solve #2, PRE:954-967
      VelProblem prob;
      prob.p = this;
      prob.n_param = 3 * S;
      prob.n_res = 3 * nb_vel_ + 3 * S;
      VecX x(3 * S);
      for (int c = 0; c < 3; c++) std::copy(state_vel_[c].begin(), state_vel_[c].end(), x.begin() + c * S);
      sum_vel = ceres_like_solve(prob, x);
      for (int c = 0; c < 3; c++) state_vel_[c].assign(x.begin() + c * S, x.begin() + (c + 1) * S);
    }
    finishStateDiff();  // PRE:970

    alpha_.resize(6);  // PRE:978-989
    for (int i = 0; i < 6; i++) alpha_[i] = matvec(K_inv_[i], i < 3 ? state_d_r_[i] : state_vel_[i - 3]);

    // PRE:995-1060
    d_vel_bv_.assign(3, MatX(S, 3));
    d_vel_bw_.assign(3, MatX(S, 3));
    d_vel_dt_.assign(3, VecX(S));
    std::vector<MatX> d_state_r_bw(3);
    std::vector<VecX> state_r(3);
    for (int a = 0; a < 3; a++) {
      state_r[a] = matvec(K_int_K_inv_[a], state_d_r_[a]);
      for (int i = 0; i < S; i++) state_r[a][i] += (state_time_[i] - start_t_) * hyper_[a].mean;
      d_state_r_bw[a] = matmul(K_int_K_inv_[a], d_state_bw_[a]);  // net effect of the 12-step loop PRE:1001-1019
    }
    V3 start_r_dt;
    for (int i = 0; i < 3; i++) {
      const MatX k = seKernelIntegral(start_t_, VecX{start_t_ + kNumDtJacobianDelta}, state_time_, hyper_[i].l2, hyper_[i].sf2);
      double s = 0;
      for (int j = 0; j < S; j++) s += k(0, j) * alpha_[i][j];
      start_r_dt[i] = s + kNumDtJacobianDelta * hyper_[i].mean;
    }
    const M3 delta_R_dt_start = expMap(start_r_dt);
    const V3 mean_vel = {hyper_[3].mean, hyper_[4].mean, hyper_[5].mean};
    for (int i = 0; i < S; i++) {
      const V3 ri = {state_r[0][i], state_r[1][i], state_r[2][i]};
      const M3 R = expMap(ri);
      for (int a = 0; a < 3; a++)
        for (int c = 0; c < 3; c++) d_vel_bv_[a](i, c) = R(a, c);
      const V3 sv = V3{state_vel_[0][i], state_vel_[1][i], state_vel_[2][i]} + mean_vel;
      M3 d_r_bw;
      for (int a = 0; a < 3; a++)
        for (int c = 0; c < 3; c++) d_r_bw(a, c) = d_state_r_bw[a](i, c);
      const M3 d_vel_bw = scale(mul(mul(skew(sv), jacobianRighthandSO3(V3{-ri[0], -ri[1], -ri[2]})), d_r_bw), -1.0);  // PRE:1048
      for (int a = 0; a < 3; a++)
        for (int c = 0; c < 3; c++) d_vel_bw_[a](i, c) = d_vel_bw(a, c);
      const V3 vel_rot_dt = mul(tr(delta_R_dt_start), sv);
      const V3 dvdt = (1.0 / kNumDtJacobianDelta) * (vel_rot_dt - sv);
      for (int a = 0; a < 3; a++) d_vel_dt_[a][i] = dvdt[a];
    }
    if (correlate) cor_thread.join();  // PRE:1062-1065
  }

  // unwrapped rotation vectors / numeric state derivative shared by PRE:1198-1263 and PRE:1265-1397
  template <class GetR>
  void unwrap_states(const M3& start_R, GetR getR, std::vector<V3>& r0, std::vector<V3>& r1) const {
    r0.assign(nb_state_, V3{0, 0, 0});
    r1.assign(nb_state_, V3{0, 0, 0});
    for (int pass = 0; pass < 2; pass++) {
      double revolution[2] = {0.0, 0.0};
      V3 prev[2] = {V3{0, 0, 0}, V3{0, 0, 0}};
      const int from = pass == 0 ? nb_overlap_ : nb_overlap_ - 1, to = pass == 0 ? nb_state_ : -1, step = pass == 0 ? 1 : -1;
      for (int i = from; i != to; i += step) {
        for (int j = 0; j < 2; j++) {
          const V3 temp_r = logMap(mul(tr(start_R), getR(j, i)));
          const V3 cand[3] = {addN2Pi(temp_r, (int)revolution[j] - 1), addN2Pi(temp_r, (int)revolution[j]), addN2Pi(temp_r, (int)revolution[j] + 1)};
          int id = 0;
          double best = std::numeric_limits<double>::max();
          for (int q = 0; q < 3; q++) {
            const double d = norm(prev[j] - cand[q]);
            if (d < best) {
              best = d;
              id = q;
            }
          }
          prev[j] = cand[id];
          revolution[j] += (id - 1);
        }
        r0[i] = prev[0];
        r1[i] = prev[1];
      }
    }
  }

  static std::vector<std::vector<double>> lpm_queries(const std::vector<double>& t_vect, const std::vector<double>& t_vect_dt, double start_t) { return {t_vect, t_vect_dt, std::vector<double>(1, start_t)}; }

  void initialiseStateWithLPM(const GyroVelData& data, const std::vector<double>& t_vect, const std::vector<double>& t_vect_dt, const PreintPrior& prior) {  // PRE:1198-1264
    IterativeIntegrator lpm(data, t_vect[0], prior, lpm_queries(t_vect, t_vect_dt, start_t_), 500, false, false);  // VelPreintegration(type=LPM), PRE:1207 -> PRE:1569
    const M3 start_R = lpm.get(2, 0).delta_R;
    std::vector<V3> r0, r1;
    unwrap_states(start_R, [&](int j, int i) { return lpm.get(j, i).delta_R; }, r0, r1);
    state_d_r_.assign(3, VecX(nb_state_));
    state_vel_.assign(3, VecX(nb_state_));
    d_r_dt_local_.assign(nb_state_, V3{0, 0, 0});
    state_r_temp_ = r0;
    for (int i = 0; i < nb_state_; i++) {
      const V3 d = (1.0 / kNumDtJacobianDelta) * (r1[i] - r0[i]);
      const V3 v = mul(tr(start_R), (1.0 / kNumDtJacobianDelta) * (lpm.get(1, i).delta_p - lpm.get(0, i).delta_p));
      for (int a = 0; a < 3; a++) {
        state_d_r_[a][i] = d[a];
        state_vel_[a][i] = v[a];
      }
      d_r_dt_local_[i] = mul(jacobianRighthandSO3(r0[i]), d);
    }
  }

  void initialiseStateDiff(const GyroVelData& data, const std::vector<double>& t_vect, const std::vector<double>& t_vect_dt) {  // PRE:1265-1399
    {
      GyroVelData shifted = data;
      for (auto& s : shifted.gyr) s.t -= kNumDtJacobianDelta;
      for (auto& s : shifted.vel) s.t -= kNumDtJacobianDelta;
      PreintPrior zero;
      IterativeIntegrator lpm(shifted, t_vect[0], zero, lpm_queries(t_vect, t_vect_dt, start_t_), 500, false, false);
      const M3 start_R = lpm.get(2, 0).delta_R;
      std::vector<V3> r0, r1;
      unwrap_states(start_R, [&](int j, int i) { return lpm.get(j, i).delta_R; }, r0, r1);
      d_r_dt_local_shift_.assign(nb_state_, V3{0, 0, 0});
      delta_r_time_.assign(nb_state_, V3{0, 0, 0});
      for (int i = 0; i < nb_state_; i++) {
        const M3 Jr = jacobianRighthandSO3(r0[i]);
        d_r_dt_local_shift_[i] = mul(Jr, (1.0 / kNumDtJacobianDelta) * (r1[i] - r0[i]));
        delta_r_time_[i] = mul(Jr, r0[i] - state_r_temp_[i]);
      }
    }
    d_r_bw_local_shift_.assign(3, std::vector<V3>(nb_state_, V3{0, 0, 0}));
    delta_r_bw_.assign(3, std::vector<V3>(nb_state_, V3{0, 0, 0}));
    for (int axis = 0; axis < 3; axis++) {
      GyroVelData biased = data;
      for (auto& s : biased.gyr) s.data[axis] += kNumGyrBiasJacobianDelta;
      PreintPrior zero;
      IterativeIntegrator lpm(biased, t_vect[0], zero, lpm_queries(t_vect, t_vect_dt, start_t_), 500.0, true, true);  // PRE:1350
      const M3 start_R = lpm.get(2, 0).delta_R;
      std::vector<V3> r0, r1;
      unwrap_states(start_R, [&](int j, int i) { return lpm.get(j, i).delta_R; }, r0, r1);
      for (int i = 0; i < nb_state_; i++) {
        const M3 Jr = jacobianRighthandSO3(r0[i]);
        d_r_bw_local_shift_[axis][i] = mul(Jr, (1.0 / kNumDtJacobianDelta) * (r1[i] - r0[i]));
        delta_r_bw_[axis][i] = mul(Jr, r0[i] - state_r_temp_[i]);
      }
    }
  }

  void initialiseHyperParam(const GyroVelData& data) {  // PRE:1444-1476
    for (int i = 0; i < 6; i++) {
      VecX& s = i < 3 ? state_d_r_[i] : state_vel_[i - 3];
      double m = 0;
      for (double v : s) m += v;
      m /= (double)s.size();
      double var = 0;
      for (double v : s) var += (v - m) * (v - m);
      var /= (double)s.size();
      hyper_[i].mean = m;
      hyper_[i].sf2 = std::max(var, i < 3 ? data.gyr_var : data.vel_var);
      for (double& v : s) v -= m;
    }
    const double l2 = std::pow(3.0 / state_freq_, 2);
    for (int i = 0; i < 3; i++) {
      hyper_[i].l2 = l2;
      hyper_[i + 3].l2 = l2;
      hyper_[i].sz2 = data.gyr_var;
      hyper_[i + 3].sz2 = data.vel_var;
    }
  }

  void finishStateDiff() {  // PRE:1401-1441
    const int S = nb_state_;
    std::vector<VecX> state_r(3);
    for (int a = 0; a < 3; a++) {
      state_r[a] = matvec(K_int_K_inv_[a], state_d_r_[a]);
      for (int i = 0; i < S; i++) state_r[a][i] += (state_time_[i] - start_t_) * hyper_[a].mean;
    }
    d_d_r_dt_.assign(3, VecX(S));
    d_state_bw_.assign(3, MatX(S, 3));
    for (int i = 0; i < S; i++) {
      const V3 ri = {state_r[0][i], state_r[1][i], state_r[2][i]};
      const M3 Jinv = inverseJacobianRighthandSO3(ri);
      const V3 d_r = mul(Jinv, d_r_dt_local_[i]);
      const V3 temp_r = ri + mul(Jinv, delta_r_time_[i]);
      const V3 d_r_dt = mul(inverseJacobianRighthandSO3(temp_r), d_r_dt_local_shift_[i]);
      const V3 dd = (1.0 / kNumDtJacobianDelta) * (d_r_dt - d_r);
      for (int a = 0; a < 3; a++) d_d_r_dt_[a][i] = dd[a];
      for (int axis = 0; axis < 3; axis++) {
        const V3 temp_r_w = ri + mul(Jinv, delta_r_bw_[axis][i]);
        const V3 d2 = mul(inverseJacobianRighthandSO3(temp_r_w), d_r_bw_local_shift_[axis][i]);
        const V3 t = (1.0 / kNumGyrBiasJacobianDelta) * (d2 - d_r);
        for (int a = 0; a < 3; a++) d_state_bw_[a](i, axis) = t[a];
      }
    }
  }

  void computeStateCorr(const MatX& state_J, const VecX& state_std) {  // PRE:1478-1492
    const int n = 6 * nb_state_;
    MatX A = gram(state_J);
    for (int i = 0; i < n; i++) A(i, i) += 0.00001;
    MatX L;
    if (!cholesky(A, L)) throw std::runtime_error("state correlation: LLT failed");
    // (L L^T)^-1 = L^-T L^-1 : invert L (lower) then form the product
    MatX Li(n, n);
    for (int col = 0; col < n; col++) {
      Li(col, col) = 1.0 / L(col, col);
      for (int i = col + 1; i < n; i++) {
        double s = 0;
        const double* li = L.row(i);
        for (int k = col; k < i; k++) s -= li[k] * Li(k, col);
        Li(i, col) = s / L(i, i);
      }
    }
    MatX C(n, n);
    for (int k = 0; k < n; k++) {
      const double* lk = Li.row(k);
      for (int i = 0; i <= k; i++) {
        const double a = lk[i];
        if (a == 0.0) continue;
        double* ci = C.row(i);
        for (int j = 0; j <= k; j++) ci[j] += a * lk[j];
      }
    }
    VecX dsc(n);
    for (int i = 0; i < n; i++) dsc[i] = state_std[i] * (1.0 / std::sqrt(C(i, i)));
    for (int i = 0; i < n; i++)
      for (int j = 0; j < n; j++) C(i, j) *= dsc[i] * dsc[j];
    state_cor_ = C;
  }

  PreintMeas get(double t) const {  // PRE:1069-1153
    const int S = nb_state_;
    PreintMeas pre;
    V3 r, d_r_dt, p, d_p_dt;
    M3 d_r_dw, d_p_dw, d_p_dv;
    const double dt = t - start_t_;
    MatX state_ks(6, 6 * S);
    double var_vec[6];
    for (int i = 0; i < 6; i++) {
      const MatX ks = seKernelIntegral(start_t_, VecX{t}, state_time_, hyper_[i].l2, hyper_[i].sf2);
      VecX ks_K_inv(S, 0.0);
      for (int k = 0; k < S; k++) {
        const double a = ks(0, k);
        const double* row = K_inv_[i].row(k);
        for (int j = 0; j < S; j++) ks_K_inv[j] += a * row[j];
      }
      double ka = 0, kKk = 0;
      for (int j = 0; j < S; j++) {
        ka += ks(0, j) * alpha_[i][j];
        kKk += ks_K_inv[j] * ks(0, j);
        state_ks(i, i * S + j) = ks_K_inv[j];
      }
      var_vec[i] = kssInt(start_t_, t, hyper_[i].l2, hyper_[i].sf2) - kKk;
      if (i < 3) {
        r[i] = ka + (t - start_t_) * hyper_[i].mean;
        double s = 0;
        for (int c = 0; c < 3; c++) {
          double q = 0;
          for (int j = 0; j < S; j++) q += ks_K_inv[j] * d_state_bw_[i](j, c);
          d_r_dw(i, c) = q;
        }
        for (int j = 0; j < S; j++) s += ks_K_inv[j] * d_d_r_dt_[i][j];
        d_r_dt[i] = s;
        if (var_vec[i] <= 0) var_vec[i] = dt * dt * hyper_[i].sz2;
      } else {
        const MatX ks_dt = seKernelIntegralDt(start_t_, VecX{t}, state_time_, hyper_[i].l2, hyper_[i].sf2);
        p[i - 3] = ka + (t - start_t_) * hyper_[i].mean;
        double kd = 0, kv = 0;
        for (int j = 0; j < S; j++) {
          kd += ks_dt(0, j) * alpha_[i][j];
          kv += ks_K_inv[j] * d_vel_dt_[i - 3][j];
        }
        d_p_dt[i - 3] = kd + kv;
        for (int c = 0; c < 3; c++) {
          double qw = 0, qv = 0;
          for (int j = 0; j < S; j++) {
            qw += ks_K_inv[j] * d_vel_bw_[i - 3](j, c);
            qv += ks_K_inv[j] * d_vel_bv_[i - 3](j, c);
          }
          d_p_dw(i - 3, c) = qw;
          d_p_dv(i - 3, c) = qv;
        }
        if (var_vec[i] <= 0) var_vec[i] = std::pow(dt, 2.0) * hyper_[i].sz2;
      }
    }
    const M3 j_right = jacobianRighthandSO3(r);
    pre.dt = dt;
    pre.dt_sq_half = 0.5 * std::pow(dt, 2);
    pre.delta_R = expMap(r);
    pre.d_delta_R_d_t = mul(j_right, d_r_dt);
    pre.d_delta_R_d_bw = mul(j_right, d_r_dw);
    pre.delta_p = p;
    pre.d_delta_p_d_t = d_p_dt;
    pre.d_delta_p_d_bw = d_p_dw;
    pre.d_delta_p_d_bv = d_p_dv;
    double cov[36] = {0};
    if (correlate_) {  // PRE:1131-1136: state_ks C state_ks^T; row i of state_ks lives in block i only
      for (int a = 0; a < 6; a++)
        for (int b = 0; b < 6; b++) {
          double s = 0;
          for (int j = 0; j < S; j++) {
            const double ka = state_ks(a, a * S + j);
            if (ka == 0.0) continue;
            const double* crow = state_cor_.row(a * S + j) + b * S;
            double q = 0;
            for (int k = 0; k < S; k++) q += crow[k] * state_ks(b, b * S + k);
            s += ka * q;
          }
          cov[a * 6 + b] = s;
        }
    } else {
      for (int a = 0; a < 6; a++) {
        double s = 0;
        for (int j = 0; j < S; j++) s += state_ks(a, a * S + j) * state_var_[a * S + j] * state_ks(a, a * S + j);
        cov[a * 6 + a] = s;
      }
    }
    double tdv[6];  // PRE:1141-1145
    for (int a = 0; a < 6; a++) tdv[a] = std::sqrt(var_vec[a]) * (1.0 / std::sqrt(cov[a * 6 + a]));
    for (int a = 0; a < 6; a++)
      for (int b = 0; b < 6; b++) cov[a * 6 + b] *= tdv[a] * tdv[b];
    // PRE:1148-1150
    M3 c00, c03;
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) {
        c00(a, b) = cov[a * 6 + b];
        c03(a, b) = cov[a * 6 + 3 + b];
      }
    const M3 n00 = mul(mul(j_right, c00), tr(j_right)), n03 = mul(j_right, c03);
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) {
        cov[a * 6 + b] = n00(a, b);
        cov[a * 6 + 3 + b] = n03(a, b);
        cov[(3 + b) * 6 + a] = n03(a, b);
      }
    std::memcpy(pre.cov, cov, sizeof(cov));
    return pre;
  }
};

// VelPreintegration::get cov inflation, PRE:1744-1757
static void inflate_cov(PreintMeas& out, double vel_bias_std, double gyr_bias_std) {
  if (!(vel_bias_std > 0.0 || gyr_bias_std > 0.0)) return;
  double J[36] = {0}, bc[6] = {gyr_bias_std * gyr_bias_std, gyr_bias_std * gyr_bias_std, gyr_bias_std * gyr_bias_std, vel_bias_std * vel_bias_std, vel_bias_std * vel_bias_std, vel_bias_std * vel_bias_std};
  for (int a = 0; a < 3; a++) {
    J[a * 6 + a] = 1.0;  // inverseJacobianRighthandSO3(0) = I
    for (int b = 0; b < 3; b++) {
      J[(3 + a) * 6 + b] = out.d_delta_p_d_bw(a, b);
      J[(3 + a) * 6 + 3 + b] = out.d_delta_p_d_bv(a, b);
    }
  }
  for (int a = 0; a < 6; a++)
    for (int b = 0; b < 6; b++) {
      double s = 0;
      for (int k = 0; k < 6; k++) s += J[a * 6 + k] * bc[k] * J[b * 6 + k];
      out.cov[a * 6 + b] += s;
    }
}


// ------------------------------------------------------------------------------------------------ chunked mode (PRE:1584-1702, MATH:206-314, 540-726)
// The reference splits [start_t, last inference time] into chunks of opt.quantum seconds, pre-integrates every chunk on its own
// (a non-chunked VelPreintegration over the samples of the chunk +- overlap periods) and chains the chunks with combinePreints.
// ONE deliberate deviation, stated where it happens: TYPES:36 declares `Vec12` as a 9-vector, so the reference's covariance
// propagation (MATH:540-574) reads and writes past the end of `eps` -- undefined behaviour, no defined result to follow.  The
// restatement uses the 12 components the code plainly intends.  Everything else (delta_R, delta_p, dt, the five Jacobians) is
// well defined in the reference and is followed operation by operation.
static const double kLogTraceTolerance = 3.0 - kExpNormTolerance;  // MATH:12

// jacobianLogMap, MATH:227-313: d log(R) / d vec(R) (3 x 9, vec column-major).  The reference spells every entry out (symbolic
// toolbox output); they are three distinct expressions of c = R00/2 + R11/2 + R22/2 - 0.5, evaluated here in the reference's order.
struct M39 {
  double m[27];
  double& operator()(int r, int c) { return m[r * 9 + c]; }
  double operator()(int r, int c) const { return m[r * 9 + c]; }
};
static M39 jacobianLogMap(const M3& R) {
  M39 o;
  for (double& v : o.m) v = 0.0;
  const double trace = R(0, 0) + R(1, 1) + R(2, 2);
  if (trace < kLogTraceTolerance) {
    const double c = R(0, 0) / 2.0 + R(1, 1) / 2.0 + R(2, 2) / 2.0 - 0.5;
    const double half = std::acos(c) / (2 * std::pow(1 - std::pow(c, 2), 0.5));                  // the +-A entries
    auto diag_term = [&](double u) {                                                             // u / (4 (c^2 - 1)) + acos(c) u c / (4 (1 - c^2)^1.5)
      return u / (4 * (std::pow(c, 2) - 1)) + (std::acos(c) * u * c) / (4 * std::pow(1 - std::pow(c, 2), 1.5));
    };
    const double d0 = -diag_term(R(1, 2) - R(2, 1));  // MATH:236-258: - u/(..) - (..)
    const double d1 = diag_term(R(0, 2) - R(2, 0));   // MATH:259-281
    const double d2 = -diag_term(R(0, 1) - R(1, 0));  // MATH:282-304
    o(0, 0) = d0; o(0, 4) = d0; o(0, 8) = d0; o(0, 5) = half; o(0, 7) = -half;
    o(1, 0) = d1; o(1, 4) = d1; o(1, 8) = d1; o(1, 2) = -half; o(1, 6) = half;
    o(2, 0) = d2; o(2, 4) = d2; o(2, 8) = d2; o(2, 1) = half; o(2, 3) = -half;
  } else {  // MATH:306-309
    o(0, 5) = 0.5; o(0, 7) = -0.5;
    o(1, 2) = -0.5; o(1, 6) = 0.5;
    o(2, 1) = 0.5; o(2, 3) = -0.5;
  }
  return o;
}
// jacobianYX(R) * jacobianExpMapZeroM(M)  (MATH:212-225, 342-349): 9 x 3, block b (rows 3b..3b+2) = R * E_b with
// E_0 = [0; M row 2; -M row 1], E_1 = [-M row 2; 0; M row 0], E_2 = [M row 1; -M row 0; 0]
struct M93 {
  double m[27];
  double& operator()(int r, int c) { return m[r * 3 + c]; }
  double operator()(int r, int c) const { return m[r * 3 + c]; }
};
static M93 rot_times_exp_zero(const M3& R, const M3& M) {
  double E[9][3];
  for (int k = 0; k < 3; k++) {
    E[0][k] = 0;        E[1][k] = M(2, k);  E[2][k] = -M(1, k);
    E[3][k] = -M(2, k); E[4][k] = 0;        E[5][k] = M(0, k);
    E[6][k] = M(1, k);  E[7][k] = -M(0, k); E[8][k] = 0;
  }
  M93 o;
  for (int b = 0; b < 3; b++)
    for (int i = 0; i < 3; i++)
      for (int k = 0; k < 3; k++) {  // a 9 x 9 block-diagonal product in Eigen: the zero blocks contribute exact zeros, the order inside the block is j = 0, 1, 2
        double acc = 0;
        for (int j = 0; j < 3; j++) acc += R(i, j) * E[3 * b + j][k];
        o(3 * b + i, k) = acc;
      }
  return o;
}
// propagateJacobianRp (matrix form MATH:577-593, vector form MATH:594-610: the vector form is the matrix form with one column)
static void propagate_rp(const M3& R, const double* d_r, const V3& p, const double* d_p, int ncol, double* out /* 3 x ncol row-major */) {
  M3 Mr = M3{{0}};
  for (int i = 0; i < 3; i++)
    for (int k = 0; k < ncol; k++) Mr(i, k) = d_r[i * ncol + k];
  const M93 dR = rot_times_exp_zero(R, Mr);
  for (int i = 0; i < 3; i++)
    for (int k = 0; k < ncol; k++) {
      double rd = 0;
      for (int j = 0; j < 3; j++) rd += R(i, j) * d_p[j * ncol + k];
      out[i * ncol + k] = ((rd + dR(i, k) * p[0]) + dR(3 + i, k) * p[1]) + dR(6 + i, k) * p[2];
    }
}
// propagateJacobianRR (MATH:612-648 / 649-686): jacobianLogMap(R1 R2) * d vec(R1 R2)
static void propagate_rr(const M3& R1, const double* d_r1, const M3& R2, const double* d_r2, int ncol, double* out /* 3 x ncol */) {
  M3 M1 = M3{{0}}, M2 = M3{{0}};
  for (int i = 0; i < 3; i++)
    for (int k = 0; k < ncol; k++) {
      M1(i, k) = d_r1[i * ncol + k];
      M2(i, k) = d_r2[i * ncol + k];
    }
  const M93 dR1 = rot_times_exp_zero(R1, M1), dR2 = rot_times_exp_zero(R2, M2);
  double dRR[9][3];
  for (int c = 0; c < 3; c++)      // column of R1 R2
    for (int i = 0; i < 3; i++)    // row
      for (int k = 0; k < ncol; k++) {
        double rd = 0;
        for (int j = 0; j < 3; j++) rd += R1(i, j) * dR2(3 * c + j, k);
        dRR[3 * c + i][k] = ((rd + dR1(i, k) * R2(0, c)) + dR1(3 + i, k) * R2(1, c)) + dR1(6 + i, k) * R2(2, c);
      }
  const M39 JL = jacobianLogMap(mul(R1, R2));
  for (int i = 0; i < 3; i++)
    for (int k = 0; k < ncol; k++) {
      double acc = 0;
      for (int q = 0; q < 9; q++) acc += JL(i, q) * dRR[q][k];
      out[i * ncol + k] = acc;
    }
}
// perturbationPropagation, MATH:540-553 (eps has the 12 components the code addresses; see the note above)
static void perturbation_propagation(const double eps[12], const PreintMeas& prev, const PreintMeas& curr, double out[6]) {
  const V3 e_r1 = {eps[0], eps[1], eps[2]}, e_p1 = {eps[3], eps[4], eps[5]}, e_r2 = {eps[6], eps[7], eps[8]}, e_p2 = {eps[9], eps[10], eps[11]};
  const M3 exp_r1 = expMap(e_r1);
  const M3 R_exp = mul(prev.delta_R, exp_r1);
  const V3 r = logMap(mul(mul(mul(tr(curr.delta_R), exp_r1), curr.delta_R), expMap(e_r2)));
  const V3 p = e_p1 + mul(R_exp, curr.delta_p + e_p2);
  for (int i = 0; i < 3; i++) {
    out[i] = r[i];
    out[3 + i] = p[i];
  }
}
// propagatePreintCov, MATH:556-574: forward differences with step 1e-5, then J blkdiag(prev.cov, curr.cov) J^T
static void propagate_cov(const PreintMeas& prev, const PreintMeas& curr, double cov_out[36]) {
  const double quantum = 1e-5;
  double J[6][12], eps[12] = {0}, base[6], pert[6];
  perturbation_propagation(eps, prev, curr, base);
  for (int i = 0; i < 12; i++) {
    eps[i] = quantum;
    perturbation_propagation(eps, prev, curr, pert);
    for (int a = 0; a < 6; a++) J[a][i] = (pert[a] - base[a]) / quantum;
    eps[i] = 0;
  }
  double C[12][12] = {{0}};
  for (int a = 0; a < 6; a++)
    for (int b = 0; b < 6; b++) {
      C[a][b] = prev.cov[a * 6 + b];
      C[6 + a][6 + b] = curr.cov[a * 6 + b];
    }
  double T[6][12];
  for (int a = 0; a < 6; a++)
    for (int j = 0; j < 12; j++) {
      double acc = 0;
      for (int k = 0; k < 12; k++) acc += J[a][k] * C[k][j];
      T[a][j] = acc;
    }
  for (int a = 0; a < 6; a++)
    for (int b = 0; b < 6; b++) {
      double acc = 0;
      for (int k = 0; k < 12; k++) acc += T[a][k] * J[b][k];
      cov_out[a * 6 + b] = acc;
    }
}
// combinePreints, MATH:689-726
static PreintMeas combinePreints(const PreintMeas& prev, const PreintMeas& preint) {
  if (preint.dt == 0.0) return prev;
  PreintMeas t = preint;
  propagate_cov(prev, preint, t.cov);
  M3 tmp;
  // velocity-bias and gyro-bias Jacobians of the position (MATH:704-707)
  t.d_delta_p_d_bv = add(prev.d_delta_p_d_bv, mul(prev.delta_R, t.d_delta_p_d_bv));
  propagate_rp(prev.delta_R, prev.d_delta_R_d_bw.m, t.delta_p, t.d_delta_p_d_bw.m, 3, tmp.m);
  t.d_delta_p_d_bw = add(prev.d_delta_p_d_bw, tmp);
  propagate_rr(tr(t.delta_R), prev.d_delta_R_d_bw.m, t.delta_R, t.d_delta_R_d_bw.m, 3, tmp.m);  // MATH:708
  t.d_delta_R_d_bw = tmp;
  // time-shift Jacobians (MATH:711-713)
  V3 v;
  propagate_rp(prev.delta_R, prev.d_delta_R_d_t.data(), t.delta_p, t.d_delta_p_d_t.data(), 1, v.data());
  t.d_delta_p_d_t = prev.d_delta_p_d_t + v;
  propagate_rr(tr(t.delta_R), prev.d_delta_R_d_t.data(), t.delta_R, t.d_delta_R_d_t.data(), 1, v.data());
  t.d_delta_R_d_t = v;
  // chunk combination (MATH:717-722)
  t.delta_p = prev.delta_p + mul(prev.delta_R, t.delta_p);
  t.delta_R = mul(prev.delta_R, t.delta_R);
  t.dt = prev.dt + t.dt;
  t.dt_sq_half = 0.5 * t.dt * t.dt;
  return t;
}

// A non-chunked VelPreintegration (PRE:1532-1581) reduced to what the chunk loop needs: its records by (group, index), un-inflated
struct PlainPreint {
  std::vector<std::vector<PreintMeas>> rec;
  int nb_state = 0, nb_gyr = 0, nb_vel = 0, it_rot = 0, it_vel = 0;
  double cost_rot = 0, cost_vel = 0, state_freq = 0;
  PlainPreint(const GyroVelData& data, double start_t, const std::vector<std::vector<double>>& infer_t, int type, double min_freq, double state_freq_opt, bool correlate, int overlap,
              const PreintPrior& prior) {
    rec.resize(infer_t.size());
    if (type == 1) {
      std::vector<double> mx;
      for (const auto& g : infer_t)
        if (!g.empty()) mx.push_back(*std::max_element(g.begin(), g.end()));
      if (mx.empty()) throw std::invalid_argument("chunk without inference times");
      const double duration = *std::max_element(mx.begin(), mx.end()) - start_t;
      Se3Integrator se3(data, start_t, prior, duration, state_freq_opt, overlap, correlate);
      for (size_t i = 0; i < infer_t.size(); i++)
        for (double t : infer_t[i]) rec[i].push_back(se3.get(t));
      nb_state = se3.nb_state_; nb_gyr = se3.nb_gyr_; nb_vel = se3.nb_vel_;
      it_rot = se3.sum_rot.iterations; it_vel = se3.sum_vel.iterations;
      cost_rot = se3.sum_rot.final_cost; cost_vel = se3.sum_vel.final_cost;
      state_freq = se3.state_freq_;
    } else {
      IterativeIntegrator lpm(data, start_t, prior, infer_t, min_freq, false, false);
      for (size_t i = 0; i < infer_t.size(); i++)
        for (size_t j = 0; j < infer_t[i].size(); j++) rec[i].push_back(lpm.get((int)i, (int)j));
    }
  }
};

// VelPreintegration with opt.quantum >= 0, PRE:1584-1702.  Returns the records by (group, index), un-inflated.
static std::vector<std::vector<PreintMeas>> chunked_preint(const GyroVelData& imu, double start_t, const std::vector<std::vector<double>>& infer_t, int type, double min_freq,
                                                          double state_freq, bool correlate, int overlap, double quantum, const PreintPrior& prior, double* diag) {
  std::vector<double> temp_t;
  for (const auto& t : infer_t)
    if (!t.empty()) temp_t.push_back(t.back());
  if (infer_t.empty() || infer_t[0].empty()) throw std::invalid_argument("chunked mode needs a non-empty first vector of inference times");
  double last_t = infer_t[0].back();
  if (temp_t.size() > 1) last_t = *std::max_element(temp_t.begin(), temp_t.end());
  const double vel_period = (imu.vel.back().t - imu.vel[0].t) / (imu.vel.size() - 1);
  const double gyr_period = (imu.gyr.back().t - imu.gyr[0].t) / (imu.gyr.size() - 1);
  const double t_overlap = std::max(vel_period, gyr_period) * overlap;
  int nb_chunks = (int)std::ceil((last_t - start_t) / quantum);
  if (nb_chunks == 0) nb_chunks = 1;
  std::vector<size_t> pointers(infer_t.size(), 0);
  PreintMeas prev;
  std::vector<std::vector<PreintMeas>> out(infer_t.size());
  for (int i = 0; i < nb_chunks; i++) {
    const double cs = start_t + (i * quantum);
    double ce = start_t + ((i + 1) * quantum);
    if (i == nb_chunks - 1) ce = std::numeric_limits<double>::infinity();
    std::vector<std::vector<double>> sub(infer_t.size());
    if (i != nb_chunks - 1) sub.push_back({ce});
    for (size_t j = 0; j < infer_t.size(); j++)
      while (pointers[j] < infer_t[j].size() && infer_t[j][pointers[j]] < ce) sub[j].push_back(infer_t[j][pointers[j]++]);
    const GyroVelData data = imu.get(cs - t_overlap, ce + t_overlap);
    PlainPreint pre(data, cs, sub, type, min_freq, state_freq, correlate, overlap, prior);
    if (diag) {
      diag[0] = pre.nb_state; diag[1] = pre.nb_gyr; diag[2] = pre.nb_vel;
      diag[3] += pre.it_rot; diag[4] += pre.it_vel; diag[5] += pre.cost_rot; diag[6] += pre.cost_vel;
      diag[7] = pre.state_freq;
    }
    const size_t n = infer_t.size();
    if (i == 0) {
      if (nb_chunks > 1) prev = pre.rec[n][0];
      for (size_t j = 0; j < n; j++)
        for (const PreintMeas& m : pre.rec[j]) out[j].push_back(m);
    } else {
      for (size_t j = 0; j < n; j++)
        for (const PreintMeas& m : pre.rec[j]) out[j].push_back(combinePreints(prev, m));
      if (i != nb_chunks - 1) prev = combinePreints(prev, pre.rec[n][0]);
    }
  }
  return out;
}

}  // namespace ugpmo

// ================================================================================================ C interface for ctypes
extern "C" {

// flat output record: 76 doubles per query time
//  [0..8] delta_R row-major, [9..11] delta_p, [12] dt, [13] dt_sq_half, [14..49] cov 6x6 row-major,
//  [50..58] d_delta_R_d_bw, [59..61] d_delta_R_d_t, [62..70] d_delta_p_d_bw ... see pack()
static void pack(const ugpmo::PreintMeas& m, double* o) {
  for (int i = 0; i < 9; i++) o[i] = m.delta_R.m[i];
  for (int i = 0; i < 3; i++) o[9 + i] = m.delta_p[i];
  o[12] = m.dt;
  o[13] = m.dt_sq_half;
  for (int i = 0; i < 36; i++) o[14 + i] = m.cov[i];
  for (int i = 0; i < 9; i++) o[50 + i] = m.d_delta_R_d_bw.m[i];
  for (int i = 0; i < 3; i++) o[59 + i] = m.d_delta_R_d_t[i];
  for (int i = 0; i < 9; i++) o[62 + i] = m.d_delta_p_d_bw.m[i];
  for (int i = 0; i < 9; i++) o[71 + i] = m.d_delta_p_d_bv.m[i];
  for (int i = 0; i < 3; i++) o[80 + i] = m.d_delta_p_d_t[i];
}

static ugpmo::GyroVelData make_data(const double* gyr_t, const double* gyr, int n_g, const double* vel_t, const double* vel, int n_v, double gyr_var, double vel_var) {
  ugpmo::GyroVelData d;
  d.gyr_var = gyr_var;
  d.vel_var = vel_var;
  d.gyr.resize(n_g);
  d.vel.resize(n_v);
  for (int i = 0; i < n_g; i++) d.gyr[i] = {gyr_t[i], {gyr[3 * i], gyr[3 * i + 1], gyr[3 * i + 2]}};
  for (int i = 0; i < n_v; i++) d.vel[i] = {vel_t[i], {vel[3 * i], vel[3 * i + 1], vel[3 * i + 2]}};
  return d;
}

/*
 * ugpm::VelPreintegration(imu, start_t, infer_t[n_infer], opt{type}, prior) + get(0, j, vel_bias_std, gyr_bias_std)  (PRE:1517-1581, 1734-1765).
 * type: 0 = LPM, 1 = UGPM.  out: n_infer * 83 doubles (see pack()).  diag (may be null): [0] nb_state, [1] nb_gyr, [2] nb_vel,
 * [3..4] LM iterations of the two solves, [5..6] their final costs, [7] state_freq.
 * Returns 0, or -1 with the exception text in err (cap err_cap).
 */
int ugpmo_preintegrate(const double* gyr_t, const double* gyr, int n_g, const double* vel_t, const double* vel, int n_v, double gyr_var, double vel_var, double start_t,
                       const double* infer_t, int n_infer, int type, double min_freq, double state_freq, int correlate, int overlap, const double* gyr_bias, const double* vel_bias,
                       double vel_bias_std, double gyr_bias_std, double* out, double* diag, char* err, int err_cap) {
  try {
    using namespace ugpmo;
    const GyroVelData data = make_data(gyr_t, gyr, n_g, vel_t, vel, n_v, gyr_var, vel_var);
    PreintPrior prior;
    for (int i = 0; i < 3; i++) {
      prior.gyr_bias[i] = gyr_bias ? gyr_bias[i] : 0.0;
      prior.vel_bias[i] = vel_bias ? vel_bias[i] : 0.0;
    }
    std::vector<double> q(infer_t, infer_t + n_infer);
    if (type == 1) {
      const double duration = *std::max_element(q.begin(), q.end()) - start_t;  // PRE:1544-1552
      Se3Integrator se3(data, start_t, prior, duration, state_freq, overlap, correlate != 0);
      for (int j = 0; j < n_infer; j++) {
        PreintMeas m = se3.get(q[j]);
        inflate_cov(m, vel_bias_std, gyr_bias_std);
        pack(m, out + (size_t)j * 83);
      }
      if (diag) {
        diag[0] = se3.nb_state_;
        diag[1] = se3.nb_gyr_;
        diag[2] = se3.nb_vel_;
        diag[3] = se3.sum_rot.iterations;
        diag[4] = se3.sum_vel.iterations;
        diag[5] = se3.sum_rot.final_cost;
        diag[6] = se3.sum_vel.final_cost;
        diag[7] = se3.state_freq_;
      }
    } else {
      IterativeIntegrator lpm(data, start_t, prior, {q}, min_freq, false, false);  // PRE:1569
      for (int j = 0; j < n_infer; j++) {
        PreintMeas m = lpm.get(0, j);
        inflate_cov(m, vel_bias_std, gyr_bias_std);
        pack(m, out + (size_t)j * 83);
      }
    }
    return 0;
  } catch (const std::exception& e) {
    if (err && err_cap > 0) {
      std::strncpy(err, e.what(), err_cap - 1);
      err[err_cap - 1] = 0;
    }
    return -1;
  }
}

/*
 * ugpm::VelPreintegration with opt.quantum >= 0 (chunked mode, PRE:1584-1702) + get(i, j, vel_bias_std, gyr_bias_std) for every stamp.
 * infer_t: the inner vectors laid end to end, group_sizes[n_groups] their lengths.  out: n_infer * 83 doubles, group-major.
 * diag (may be null): as ugpmo_preintegrate, with [0..2], [7] of the LAST chunk and [3..6] summed over the chunks.
 */
int ugpmo_preintegrate_chunked(const double* gyr_t, const double* gyr, int n_g, const double* vel_t, const double* vel, int n_v, double gyr_var, double vel_var, double start_t,
                               const double* infer_t, const int* group_sizes, int n_groups, int type, double min_freq, double state_freq, int correlate, int overlap, double quantum,
                               const double* gyr_bias, const double* vel_bias, double vel_bias_std, double gyr_bias_std, double* out, double* diag, char* err, int err_cap) {
  try {
    using namespace ugpmo;
    const GyroVelData data = make_data(gyr_t, gyr, n_g, vel_t, vel, n_v, gyr_var, vel_var);
    PreintPrior prior;
    for (int i = 0; i < 3; i++) {
      prior.gyr_bias[i] = gyr_bias ? gyr_bias[i] : 0.0;
      prior.vel_bias[i] = vel_bias ? vel_bias[i] : 0.0;
    }
    std::vector<std::vector<double>> groups(n_groups);
    size_t o = 0;
    for (int g = 0; g < n_groups; g++)
      for (int k = 0; k < group_sizes[g]; k++) groups[g].push_back(infer_t[o++]);
    if (diag)
      for (int i = 0; i < 8; i++) diag[i] = 0;
    const auto rec = chunked_preint(data, start_t, groups, type, min_freq, state_freq, correlate != 0, overlap, quantum, prior, diag);
    o = 0;
    for (int g = 0; g < n_groups; g++) {
      if ((int)rec[g].size() != group_sizes[g]) throw std::range_error("VelPreintegration::get: Trying to get precomputed preintegrated measurements (wrong index query?)");
      for (PreintMeas m : rec[g]) {
        inflate_cov(m, vel_bias_std, gyr_bias_std);
        pack(m, out + 83 * o++);
      }
    }
    return 0;
  } catch (const std::exception& e) {
    if (err && err_cap > 0) {
      std::strncpy(err, e.what(), err_cap - 1);
      err[err_cap - 1] = 0;
    }
    return -1;
  }
}

/* combinePreints (MATH:689-726) on two packed records (83 doubles each, see pack()) */
void ugpmo_combine_preints(const double* prev83, const double* cur83, double* out83) {
  auto unpack = [](const double* o) {
    ugpmo::PreintMeas m;
    for (int i = 0; i < 9; i++) m.delta_R.m[i] = o[i];
    for (int i = 0; i < 3; i++) m.delta_p[i] = o[9 + i];
    m.dt = o[12];
    m.dt_sq_half = o[13];
    for (int i = 0; i < 36; i++) m.cov[i] = o[14 + i];
    for (int i = 0; i < 9; i++) m.d_delta_R_d_bw.m[i] = o[50 + i];
    for (int i = 0; i < 3; i++) m.d_delta_R_d_t[i] = o[59 + i];
    for (int i = 0; i < 9; i++) m.d_delta_p_d_bw.m[i] = o[62 + i];
    for (int i = 0; i < 9; i++) m.d_delta_p_d_bv.m[i] = o[71 + i];
    for (int i = 0; i < 3; i++) m.d_delta_p_d_t[i] = o[80 + i];
    return m;
  };
  pack(ugpmo::combinePreints(unpack(prev83), unpack(cur83)), out83);
}

/*
 * The optimised GP states and hyper-parameters of one UGPM window (for the cross-check against oracle/ugpm_scipy.py): states_out
 * receives 6 * S doubles (channel-major: d_r x, y, z, vel x, y, z; mean-subtracted as the reference keeps them, PRE:1464-1465),
 * hyper_out 6 * 4 doubles (l2, sf2, sz2, mean per channel).  Returns S, or -1 (exception text in err) / -2 (cap too small).
 */
int ugpmo_states(const double* gyr_t, const double* gyr, int n_g, const double* vel_t, const double* vel, int n_v, double gyr_var, double vel_var, double start_t,
                 double end_t, double state_freq, int correlate, int overlap, const double* gyr_bias, const double* vel_bias, double* states_out, int states_cap,
                 double* hyper_out, char* err, int err_cap) {
  try {
    using namespace ugpmo;
    const GyroVelData data = make_data(gyr_t, gyr, n_g, vel_t, vel, n_v, gyr_var, vel_var);
    PreintPrior prior;
    for (int i = 0; i < 3; i++) {
      prior.gyr_bias[i] = gyr_bias ? gyr_bias[i] : 0.0;
      prior.vel_bias[i] = vel_bias ? vel_bias[i] : 0.0;
    }
    Se3Integrator se3(data, start_t, prior, end_t - start_t, state_freq, overlap, correlate != 0);
    const int S = se3.nb_state_;
    if (6 * S > states_cap) return -2;
    for (int c = 0; c < 6; c++) {
      const VecX& s = c < 3 ? se3.state_d_r_[c] : se3.state_vel_[c - 3];
      for (int i = 0; i < S; i++) states_out[(size_t)c * S + i] = s[i];
      if (hyper_out) {
        hyper_out[c * 4 + 0] = se3.hyper_[c].l2;
        hyper_out[c * 4 + 1] = se3.hyper_[c].sf2;
        hyper_out[c * 4 + 2] = se3.hyper_[c].sz2;
        hyper_out[c * 4 + 3] = se3.hyper_[c].mean;
      }
    }
    return S;
  } catch (const std::exception& e) {
    if (err && err_cap > 0) {
      std::strncpy(err, e.what(), err_cap - 1);
      err[err_cap - 1] = 0;
    }
    return -1;
  }
}

// kernel helpers exported for the analytic pinning tests
void ugpmo_se_kernel(const double* x1, int n1, const double* x2, int n2, double l2, double sf2, double* out) {
  ugpmo::MatX K = ugpmo::seKernel(ugpmo::VecX(x1, x1 + n1), ugpmo::VecX(x2, x2 + n2), l2, sf2);
  std::memcpy(out, K.d.data(), sizeof(double) * K.d.size());
}
void ugpmo_se_kernel_integral(double a, const double* b, int nb, const double* x2, int n2, double l2, double sf2, double* out) {
  ugpmo::MatX K = ugpmo::seKernelIntegral(a, ugpmo::VecX(b, b + nb), ugpmo::VecX(x2, x2 + n2), l2, sf2);
  std::memcpy(out, K.d.data(), sizeof(double) * K.d.size());
}
double ugpmo_kss_int(double a, double b, double l2, double sf2) { return ugpmo::kssInt(a, b, l2, sf2); }
void ugpmo_exp_map(const double* v, double* R) {
  ugpmo::M3 m = ugpmo::expMap({v[0], v[1], v[2]});
  std::memcpy(R, m.m, sizeof(m.m));
}
void ugpmo_log_map(const double* R, double* v) {
  ugpmo::M3 m;
  std::memcpy(m.m, R, sizeof(m.m));
  ugpmo::V3 r = ugpmo::logMap(m);
  v[0] = r[0];
  v[1] = r[1];
  v[2] = r[2];
}
void ugpmo_jacobian_res(const double* r, const double* dr, double* out18) {
  double D[3][6];
  ugpmo::JacobianRes({r[0], r[1], r[2]}, {dr[0], dr[1], dr[2]}, D);
  std::memcpy(out18, D, sizeof(D));
}
void ugpmo_jr(const double* r, double* out9) {
  ugpmo::M3 m = ugpmo::jacobianRighthandSO3({r[0], r[1], r[2]});
  std::memcpy(out9, m.m, sizeof(m.m));
}
}  // extern "C"
