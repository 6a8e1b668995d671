// Drop-in for the reference's VelInt/preint.h (PRE:22-82, 1517-1781): ugpm::VelPreintegration backed by libgorio_amd.so.
//
// Same class name, namespace, three constructors and three get() overloads, so radar_graph_slam_nodelet.cpp:497-513 compiles
// unchanged.  The constructor ships the IMU window to the GPU through gorio_ugpm_preint_batch() (include/gorio_ugpm.h) with zero
// bias standard deviations; get() adds the bias-prior inflation of PRE:1744-1757 on the host from the returned Jacobians (a 6x6
// product), exactly as the reference applies it per call.  opt.quantum > 0 (chunked mode, PRE:1584-1702) is handled by the library:
// every chunk is a window of the same device batch.  There is no CPU fallback: without a HIP device the constructor throws.
#ifndef UGPM_2_H
#define UGPM_2_H

#include <stdexcept>
#include <string>
#include <vector>

#include "types.h"

#include <gorio_ugpm.h>

namespace ugpm {

enum QueryType { kVecVec, kVec, kSingle };
const int kOverlap = 8;

class VelPreintegration {
public:
  VelPreintegration(const GyroVelData& imu_data, const double start_t, const std::vector<std::vector<double> >& infer_t, const PreintOption opt,
                    const PreintPrior prior, const bool rot_only = false, const int overlap = kOverlap, const int device = 0)
      : imu_data_(imu_data), start_t_(start_t), opt_(opt), prior_(prior) {
    (void)rot_only;  // ignored on the UGPM branch of the reference too (PRE:1540-1566)
    std::vector<double> gt(imu_data.gyr.size()), g(3 * imu_data.gyr.size()), vt(imu_data.vel.size()), v(3 * imu_data.vel.size());
    for (size_t i = 0; i < imu_data.gyr.size(); ++i) {
      gt[i] = imu_data.gyr[i].t;
      for (int a = 0; a < 3; ++a) g[3 * i + a] = imu_data.gyr[i].data[a];
    }
    for (size_t i = 0; i < imu_data.vel.size(); ++i) {
      vt[i] = imu_data.vel[i].t;
      for (int a = 0; a < 3; ++a) v[3 * i + a] = imu_data.vel[i].data[a];
    }
    std::vector<double> flat;
    std::vector<int> group_sizes;
    for (const auto& grp : infer_t) {
      flat.insert(flat.end(), grp.begin(), grp.end());
      group_sizes.push_back(static_cast<int>(grp.size()));
    }
    gorio_ugpm_window w;
    gorio_ugpm_default_window(&w);
    w.gyr_t = gt.data(); w.gyr = g.data(); w.n_gyr = static_cast<int>(gt.size());
    w.vel_t = vt.data(); w.vel = v.data(); w.n_vel = static_cast<int>(vt.size());
    w.gyr_var = imu_data.gyr_var; w.vel_var = imu_data.vel_var;
    w.start_t = start_t;
    w.infer_t = flat.data(); w.n_infer = static_cast<int>(flat.size());
    w.type = opt.type == UGPM ? GORIO_UGPM_TYPE_UGPM : GORIO_UGPM_TYPE_LPM;
    w.min_freq = opt.min_freq; w.quantum = opt.quantum; w.state_freq = opt.state_freq;
    w.correlate = opt.correlate ? 1 : 0; w.overlap = overlap;
    for (int a = 0; a < 3; ++a) { w.gyr_bias[a] = prior.gyr_bias[a]; w.vel_bias[a] = prior.vel_bias[a]; }
    w.vel_bias_std = 0.0; w.gyr_bias_std = 0.0;
    w.group_sizes = group_sizes.data(); w.n_groups = static_cast<int>(group_sizes.size());
    std::vector<gorio_ugpm_meas> out(flat.size());
    const int rc = gorio_ugpm_preint_batch(&w, 1, out.data(), nullptr, device);
    if (rc != GORIO_UGPM_OK) {
      const std::string msg = gorio_ugpm_last_error();
      if (rc == GORIO_UGPM_ERR_ARGUMENT) throw std::invalid_argument(msg);  // TYPES:160
      if (rc == GORIO_UGPM_ERR_RANGE) throw std::range_error(msg);          // MATH:493, PRE:680-686
      throw std::runtime_error("VelPreintegration (gorio_amd): " + msg + " [code " + std::to_string(rc) + "]");
    }
    preint_.resize(infer_t.size());
    size_t k = 0;
    for (size_t i = 0; i < infer_t.size(); ++i)
      for (size_t j = 0; j < infer_t[i].size(); ++j) preint_[i].push_back(unpack(out[k++]));
  }
  VelPreintegration(const GyroVelData& imu_data, const double start_t, const std::vector<double>& infer_t, const PreintOption opt, const PreintPrior prior,
                    const bool rot_only = false, const int overlap = kOverlap, const int device = 0)
      : VelPreintegration(imu_data, start_t, std::vector<std::vector<double> >(1, infer_t), opt, prior, rot_only, overlap, device) {
    query_type_ = kVec;  // PRE:1716
  }
  VelPreintegration(const GyroVelData& imu_data, const double start_t, const double infer_t, const PreintOption opt, const PreintPrior prior,
                    const bool rot_only = false, const int overlap = kOverlap, const int device = 0)
      : VelPreintegration(imu_data, start_t, std::vector<std::vector<double> >(1, std::vector<double>(1, infer_t)), opt, prior, rot_only, overlap, device) {
    query_type_ = kSingle;  // PRE:1729
  }

  PreintMeas get(const int index_1, const int index_2, double vel_bias_std = 0.3, double gyr_bias_std = 0.03) {  // PRE:1734-1765
    if ((index_1 >= 0) && (index_2 >= 0) && (index_1 < static_cast<int>(preint_.size())) && (index_2 < static_cast<int>(preint_[index_1].size()))) {
      PreintMeas out = preint_[index_1][index_2];
      if (vel_bias_std > 0.0 || gyr_bias_std > 0.0) {  // cov += J diag(b) J^T, J = [I 0; d_p_d_bw d_p_d_bv]
        double J[6][6] = {{0}};
        const double bc[6] = {gyr_bias_std * gyr_bias_std, gyr_bias_std * gyr_bias_std, gyr_bias_std * gyr_bias_std,
                              vel_bias_std * vel_bias_std, vel_bias_std * vel_bias_std, vel_bias_std * vel_bias_std};
        for (int a = 0; a < 3; ++a) {
          J[a][a] = 1.0;
          for (int b = 0; b < 3; ++b) {
            J[3 + a][b] = out.d_delta_p_d_bw(a, b);
            J[3 + a][3 + b] = out.d_delta_p_d_bv(a, b);
          }
        }
        for (int a = 0; a < 6; ++a)
          for (int b = 0; b < 6; ++b) {
            double s = 0.0;
            for (int q = 0; q < 6; ++q) s += J[a][q] * bc[q] * J[b][q];
            out.cov(a, b) += s;
          }
      }
      return out;
    }
    throw std::range_error("VelPreintegration::get: Trying to get precomputed preintegrated measurements (wrong index query?)");
  }
  PreintMeas get(const int index_1, double vel_bias_std = 0.3, double gyr_bias_std = 0.03) {  // PRE:1769-1773
    if (query_type_ == kVec) return get(0, index_1, vel_bias_std, gyr_bias_std);
    throw std::range_error("VelPreintegration::get: The type of query does not math the type of constructor");
  }
  PreintMeas get(double vel_bias_std = 0.3, double gyr_bias_std = 0.03) {  // PRE:1777-1781
    if (query_type_ == kSingle) return get(0, 0, vel_bias_std, gyr_bias_std);
    throw std::range_error("VelPreintegration::get: The type of query does not math the type of constructor");
  }
  PreintPrior getPrior() { return prior_; }

  // record <-> PreintMeas (public: combinePreints below uses them)
  static gorio_ugpm_meas pack(const PreintMeas& o) {
    gorio_ugpm_meas m;
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) {
        m.delta_R[r * 3 + c] = o.delta_R(r, c);
        m.d_delta_R_d_bw[r * 3 + c] = o.d_delta_R_d_bw(r, c);
        m.d_delta_p_d_bw[r * 3 + c] = o.d_delta_p_d_bw(r, c);
        m.d_delta_p_d_bv[r * 3 + c] = o.d_delta_p_d_bv(r, c);
      }
      m.delta_p[r] = o.delta_p(r, 0);
      m.d_delta_R_d_t[r] = o.d_delta_R_d_t(r, 0);
      m.d_delta_p_d_t[r] = o.d_delta_p_d_t(r, 0);
    }
    for (int r = 0; r < 6; ++r)
      for (int c = 0; c < 6; ++c) m.cov[r * 6 + c] = o.cov(r, c);
    m.dt = o.dt;
    m.dt_sq_half = o.dt_sq_half;
    return m;
  }
  static PreintMeas unpack(const gorio_ugpm_meas& m) {
    PreintMeas o;
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) {
        o.delta_R(r, c) = m.delta_R[r * 3 + c];
        o.d_delta_R_d_bw(r, c) = m.d_delta_R_d_bw[r * 3 + c];
        o.d_delta_p_d_bw(r, c) = m.d_delta_p_d_bw[r * 3 + c];
        o.d_delta_p_d_bv(r, c) = m.d_delta_p_d_bv[r * 3 + c];
      }
      o.delta_p(r, 0) = m.delta_p[r];
      o.d_delta_R_d_t(r, 0) = m.d_delta_R_d_t[r];
      o.d_delta_p_d_t(r, 0) = m.d_delta_p_d_t[r];
    }
    for (int r = 0; r < 6; ++r)
      for (int c = 0; c < 6; ++c) o.cov(r, c) = m.cov[r * 6 + c];
    o.dt = m.dt;
    o.dt_sq_half = m.dt_sq_half;
    return o;
  }

private:
  GyroVelData imu_data_;
  double start_t_;
  PreintOption opt_;
  PreintPrior prior_;
  std::vector<std::vector<PreintMeas> > preint_;
  QueryType query_type_ = kVecVec;
};

// combinePreints (math_utils.h:689-726): the measurement of two consecutive intervals.  The chunked constructor (opt.quantum > 0,
// PRE:1584-1702) chains its chunks with it inside the library; exported because the reference's header does.
inline PreintMeas combinePreints(const PreintMeas& prev_preint, const PreintMeas& preint) {
  const gorio_ugpm_meas a = VelPreintegration::pack(prev_preint), b = VelPreintegration::pack(preint);
  gorio_ugpm_meas o;
  if (gorio_ugpm_combine_preints(&a, &b, &o) != GORIO_UGPM_OK) throw std::runtime_error(std::string("combinePreints (gorio_amd): ") + gorio_ugpm_last_error());
  return VelPreintegration::unpack(o);
}

}  // namespace ugpm
#endif
