// Drop-in for the reference's VelInt/types.h (TYPES:11-298): the data types ugpm::VelPreintegration exchanges with its caller
// (radar_graph_slam_nodelet.cpp:465-530).  Same names, members and defaults.
#ifndef PREINT_TYPES_H
#define PREINT_TYPES_H

#include <algorithm>
#include <iostream>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include <Eigen/Core>

namespace ugpm {

enum PreintType { LPM, UGPM };  // TYPES:15

inline PreintType strToPreintType(std::string type) {  // TYPES:17-28
  std::transform(type.begin(), type.end(), type.begin(), [](unsigned char c) { return std::tolower(c); });
  if (type == "lpm") return LPM;
  if (type == "ugpm") return UGPM;
  throw std::range_error("The type of preintegration method is unknown, program stopping now");
}

typedef Eigen::Matrix<double, 3, 1> Vec3;
typedef Eigen::Matrix<double, 3, 3> Mat3;
typedef Eigen::Matrix<double, 6, 6> Mat6;

struct DataSample {  // TYPES:67-71
  double t;
  double data[3];
};

struct GyroVelData {  // TYPES:74-224
  double t_offset = 0.0;
  std::vector<DataSample> vel;
  std::vector<DataSample> gyr;
  double vel_var;
  double gyr_var;

  GyroVelData() {}
  // TYPES:98-138: both streams must ascend in time (the sampling-regularity part is commented out in the reference too)
  bool checkFrequency() const {
    bool sorted = true;
    for (std::size_t i = 0; i + 1 < vel.size(); ++i)
      if (vel[i + 1].t - vel[i].t < 0) sorted = false;
    if (!sorted) std::cout << "WARNING: Velocity data is not sorted in time" << std::endl;
    bool gsorted = true;
    for (std::size_t i = 0; i + 1 < gyr.size(); ++i)
      if (gyr[i + 1].t - gyr[i].t < 0) gsorted = false;
    if (!gsorted) std::cout << "WARNING: Gyroscope data is not sorted in time" << std::endl;
    return sorted && gsorted;
  }
  GyroVelData get(double from, double to) const {  // TYPES:141-162, 187-223: from < t < to
    if (!(from <= to)) throw std::invalid_argument("The argument of GyroVelData::Get are not consistent");
    GyroVelData out;
    out.t_offset = t_offset;
    out.vel_var = vel_var;
    out.gyr_var = gyr_var;
    for (const auto& s : vel)
      if (s.t > from && s.t < to) out.vel.push_back(s);
    for (const auto& s : gyr)
      if (s.t > from && s.t < to) out.gyr.push_back(s);
    return out;
  }
};
typedef std::shared_ptr<GyroVelData> GyroVelDataPtr;

struct PreintMeasBasic {  // TYPES:236-257
  EIGEN_MAKE_ALIGNED_OPERATOR_NEW
  Mat3 delta_R;
  Vec3 delta_p;
  double dt;
  double dt_sq_half;
};

struct PreintMeas : PreintMeasBasic {  // TYPES:259-281
  EIGEN_MAKE_ALIGNED_OPERATOR_NEW
  Mat6 cov;
  Mat3 d_delta_R_d_bw;
  Vec3 d_delta_R_d_t;
  Mat3 d_delta_p_d_bw;
  Mat3 d_delta_p_d_bv;
  Vec3 d_delta_p_d_t;
  PreintMeas() {}
  PreintMeas(Mat3 d_R, Vec3 d_p, double dt_, double dt_sq_half_, Mat6 cov_mat) : PreintMeasBasic{d_R, d_p, dt_, dt_sq_half_}, cov(cov_mat) {}  // TYPES:271-276
};
typedef std::shared_ptr<PreintMeas> PreintMeasPtr;

struct PreintOption {  // TYPES:285-292
  double min_freq = 500;
  PreintType type = UGPM;
  double quantum = -1;
  double state_freq = 50.0;
  bool correlate = true;
};

struct PreintPrior {  // TYPES:294-298
  std::vector<double> vel_bias = {0, 0, 0};
  std::vector<double> gyr_bias = {0, 0, 0};
};

}  // namespace ugpm
#endif
