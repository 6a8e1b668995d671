// apd_device.h -- device-side data layout shared by apd_kernels.hip and apd_api.hip (not part of the public ABI).
//
// HBM layout of one cloud (FastAPDGICP::input_ / target_ + source_covs_ / target_covs_, APDH:109-110):
//   x[n_pad], y[n_pad], z[n_pad], label[n_pad]  float SoA; n_pad = n rounded up to 16 (+16), padding coordinates = 1e30 so a
//                                                padded candidate has distance +inf and can never win a search
//   p4[n_pad]    float4 (x, y, z, label): the gather copy (the SoA arrays feed the wave-uniform / coalesced streams)
//   cov6[n][6]   double, upper triangle (c00 c01 c02 c11 c12 c22) of the 3x3 block of the 4x4 covariance (row/col 3 are 0)
//   geo_w[n]     double, sigma3/sigma1 of the regularised covariance (APD:266-269; a pure function of cov6)
// Per source point of a pair (FastAPDGICP::correspondences_, sq_distances_, mahalanobis_, APDH:111-114):
//   best_key[n]  u64 (float bits of d) << 32 | target index, ~0 = armed / nothing found
//   corr[n] int32, sqd[n] float, omega6[n][6] double (3x3 block of the Mahalanobis matrix; row/col 3 are 0, APD:218)
#pragma once
#include <stdint.h>

namespace gorio {

// Exact search accelerator of one cloud (the role pcl::search::KdTree plays in the reference, APDH:106-107): a Morton-ordered COPY
// of the coordinates with the original index of every point, axis-aligned boxes of 32-point tiles and of 16-tile super tiles.
// Searches visit tiles in an order that finds near candidates first and skip every tile whose box is farther than the current
// bound, so they return exactly what the exhaustive search returns (ties included: candidates compare as (distance, original index)).
struct SearchIndex {
  float* sx;     // [n_spad] Morton-ordered coordinates, padded with 1e30
  float* sy;
  float* sz;
  int* orig;     // [n_spad] original index, 0x7fffffff for padding
  float4* s4;    // [n_spad] the same once more as (x, y, z, original index bits): ONE 16-byte load where a kernel stages a tile or reads a query
  float* tbox;   // [n_tiles][8]  lo.x lo.y lo.z - hi.x hi.y hi.z -   (tile = 32 consecutive sorted points; empty: lo = +inf, hi = -inf)
  float* sbox;   // [n_super][8]  (super tile = 16 tiles = 512 points)
  float* bbox;   // [ceil(n_super / 64)][8]  (block = 64 super tiles = 32 768 points)
  int n;
  int n_spad;    // multiple of 512
  int n_tiles;
  int n_super;
};

struct CloudView {
  float* x;
  float* y;
  float* z;
  float* label;
  float4* p4;       // [n_pad] the same points once more as (x, y, z, label): ONE 16-byte access where a kernel gathers a point by index
  double* cov6;
  double* geo_w;
  int n;
  int n_pad;
  SearchIndex idx;  // valid only in GORIO_SEARCH_PRUNED mode
};

// device-resident optimiser state of one scan pair (the members of LsqRegistration, LSQH:75-84, plus loop bookkeeping)
struct PairState {
  double x0[16];    // current pose, row-major (x0_isom, LSQ:56)
  double xi[16];    // trial pose of compute_error API calls
  double H[36];     // last linearisation
  double b[6];
  double y0;        // error at x0 (LSQ:130)
  double yi;        // error of the last compute_error API call
  double lambda;    // lm_lambda_ (LSQ:58: reset to -1 per align)
  double Hfin[36];  // final_hessian_
  float Tf[12];     // float cast of x0 (APD:164) used by the next correspondence search
  int iter;         // outer iterations executed so far
  int done;         // loop finished (converged / LM failure / max_iterations)
  int converged;    // converged_
  int nr_iterations;  // nr_iterations_ (LSQ:68)
  int lm_failed;    // "lm not converged!!" (LSQ:71-74)
  int n_linearize;  // linearize() calls
  int n_error;      // compute_error() trials
  unsigned int arrive;  // linearize_kernel workgroups of this pair that have stored their partial (fused optimiser step)
  // sharded-source mode (gorio_apd_comm_init): the LM trial loop is cut into launches around the all-reduces, so its locals live here
  double sd[6];      // last solved step d (LSQ:138)
  double sdelta[16]; // its delta transform (LSQ:140-142)
  double nu;         // LSQ:135
  int trial_active;  // an LM trial pose is waiting for its (all-reduced) error
  int ok;            // this outer iteration produced a step (LSQ:71)
  int trial;         // trials consumed in this outer iteration
  int pad2_;
};

struct PairDesc {
  CloudView src;
  CloudView tgt;
  unsigned long long* best_key;
  int* corr;
  float* sqd;
  double* omega6;
  double* partials;  // [nblk][28]
  int* seed;         // [src n_spad] by SORTED source position: original target index of the last search's winner (warm start of the next
                     // pruned search of the same align); read only when state->n_linearize > 0, any in-range value is valid
  unsigned int* nn_work;  // [2][nn_wcap] cycles every query wave (64 sorted source points) spent in the pruned search: the launches of an align
                          // alternate between the two halves, so that a plan can be made from a finished launch while the next one runs
  unsigned int* nn_plan;  // [1 + 16 nn_wcap] work plan of the next pruned searches (nn_plan_kernel): [0] = number of entries, then one entry per
                          // workgroup, heaviest first: wave << 8 | part << 4 | log2(parts)
  PairState* state;
  int nn_wcap;
  int pad0_;
  int nblk;          // ceil(src.n / 256)
  int nn_splits;     // target range split count for nn_search_kernel
  int nn_chunk;      // candidates per split (multiple of 16)
  int cl_points;     // N of cl_weight = 1/N (APD:273); 0 = src.n
  int shard_lo;      // sharded-source mode: this rank searches only query positions [shard_lo, shard_hi) -- sorted positions for the
  int shard_hi;      // pruned search, original indices for the exhaustive one (multiples of 256 or INT_MAX); points outside keep an
                     // armed key, so linearize / compute_error see them as "no correspondence" and add nothing
  int write_omega;   // store the Mahalanobis matrices (mahalanobis_, APDH:111).  Only compute_error reads them back (LM trials, the
                     // parity hooks); a Gauss-Newton align never does, so it skips the 48 B per point per linearisation
  int pad_;
};

// XCD-aware placement of a (x, y, units) launch whose third grid dimension counts independent units (scan pairs, clouds): the hardware
// deals consecutive workgroups round-robin to the 8 XCDs, each with an L2 of its own, so with the plain (blockIdx.x, blockIdx.z) numbering
// every XCD sees every unit's clouds.  Here unit u runs entirely on XCD u mod 8 -- its source, target, boxes and covariances stay in ONE
// L2 -- by renumbering: workgroup L (linear launch order) -> xcd = L mod 8, slot = L / 8, unit = (slot / inner) * 8 + xcd, inner index
// = slot mod inner.  Only when there are at least 16 units in multiples of 8 (a lone pair must keep the whole chip).
#ifndef GORIO_XCD_UNITS
#define GORIO_XCD_UNITS 1
#endif
struct GridPos {
  unsigned int x, y, z;
};
#ifdef __HIPCC__
__device__ __forceinline__ GridPos xcd_grid_pos() {
  GridPos g{blockIdx.x, blockIdx.y, blockIdx.z};
#if GORIO_XCD_UNITS
  if (gridDim.z >= 16u && (gridDim.z & 7u) == 0u) {
    const unsigned int inner = gridDim.x * gridDim.y;
    const unsigned int L = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned int slot = L >> 3, in = slot % inner;
    g.z = (slot / inner) * 8u + (L & 7u);
    g.x = in % gridDim.x;
    g.y = in / gridDim.x;
  }
#endif
  return g;
}
#endif

struct KnnJob {
  CloudView cloud;
  float* part_d;  // [splits][K][n]
  int* part_i;
  int* knn_out;   // [n][k] or null (parity hook: only kept when gorio_apd_params.keep_knn_indices is set)
  int* redo;      // [ceil(n_spad / 64)] per query wave: 1 = knn_collect_kernel gave up (more ties than its buffer holds), the
                  // insertion kernel knn_pruned_kernel redoes that wave; null = knn_pruned_kernel does every wave
  float* kth;     // [n_spad] k-th smallest distance of every query by SORTED position (knn_kth_kernel -> knn_collect_kernel)
  int k;
  int regularization;
  int splits;
  int chunk_len;  // multiple of 16
  int qpw;        // queries per wave of knn_kth_kernel / knn_collect_kernel: 64 when the call fills the chip, 32 / 16 / 8 when it does not
  int pad0_;
};

struct ApdConsts {
  double thr2;     // corr_dist_threshold_^2 (APD:183)
  double dist_var; // distance_variance_
  double sin_az;   // sin(azimuth_variance_ / 180 * pi)  (APD:196, evaluated on the host in double)
  double sin_el;   // sin(elevation_variance_ / 180 * pi) (APD:197)
  double rot_eps;
  double trans_eps;
  double lm_init_lambda_factor;
  double inv_n_scale;  // 1.0 (numerator of cl_weight = 1.0 / correspondences_.size(), APD:273)
  int optimizer;
  int lm_max_iterations;
  int max_iterations;
  int pad_;
};

// one cloud whose search index is being built (batched over blockIdx.y)
struct IndexJob {
  const float* x;
  const float* y;
  const float* z;
  int n;
  int npow2;                  // sort size (power of two >= n, >= 4096)
  unsigned long long* keys;   // [npow2] (33-bit Morton code) << 31 | index
  unsigned int* bb;           // [6] order-preserving encodings of min x,y,z / max x,y,z
  SearchIndex idx;
};

struct TfArg {
  float m[12];
};

}  // namespace gorio
