// Minimal stand-in for pcl/point_types.h (see compat/Eigen/Core for the rationale).  PointXYZINormal has PCL's 48-byte layout:
// data[4] = {x,y,z,1}, data_n[4] = {normal_x,normal_y,normal_z,0}, {intensity, curvature, pad, pad}.
#pragma once
#define PCL_VERSION_CALC(MAJ, MIN, PATCH) (MAJ * 100000 + MIN * 100 + PATCH)
#define PCL_VERSION PCL_VERSION_CALC(1, 10, 0)
#include <memory>
namespace pcl {
template <typename T>
using shared_ptr = std::shared_ptr<T>;
struct alignas(16) PointXYZINormal {
  union {
    float data[4];
    struct {
      float x, y, z;
    };
  };
  union {
    float data_n[4];
    struct {
      float normal_x, normal_y, normal_z;
    };
  };
  union {
    struct {
      float intensity, curvature;
    };
    float data_c[4];
  };
  PointXYZINormal() : data{0, 0, 0, 1.f}, data_n{0, 0, 0, 0}, data_c{0, 0, 0, 0} {}
};
static_assert(sizeof(PointXYZINormal) == 48, "PointXYZINormal must be 48 bytes");
}  // namespace pcl
