// Enumerations of fast_gicp that cross into this back end.  Enumerator names and ORDER are the reference's
// (fast_gicp/gicp/gicp_settings.hpp:6-10): the values travel through the C ABI as plain ints (gorio_regularization).
#ifndef FAST_GICP_GICP_SETTINGS_HPP
#define FAST_GICP_GICP_SETTINGS_HPP

namespace fast_gicp {

// how the 3 x 3 neighbourhood covariance is conditioned before it is used (fast_apdgicp_impl.hpp:374-405)
enum class RegularizationMethod {
  NONE,                // raw covariance
  MIN_EIG,             // eigenvalues clamped from below
  NORMALIZED_MIN_EIG,  // ... after normalising by the largest
  PLANE,               // (1, 1, 1e-3): the default of FastAPDGICP
  FROBENIUS            // lambda I added, then normalised by the Frobenius norm of the inverse
};

// only used by the voxelised registrations of fast_gicp (not part of this back end); kept so that headers including this file compile
enum class NeighborSearchMethod { DIRECT27, DIRECT7, DIRECT1, DIRECT_RADIUS };
enum class VoxelAccumulationMode { ADDITIVE, ADDITIVE_WEIGHTED, MULTIPLICATIVE };

}  // namespace fast_gicp
#endif
