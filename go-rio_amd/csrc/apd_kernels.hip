// apd_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the APD-GICP hot path.
//
// What each kernel computes, and the reference lines it replaces (paths relative to /root/reference):
//   APD = fast_apdgicp/include/fast_gicp/gicp/impl/fast_apdgicp_impl.hpp
//   LSQ = fast_apdgicp/include/fast_gicp/gicp/impl/lsq_registration_impl.hpp
//   SO3 = fast_apdgicp/include/fast_gicp/so3/so3.hpp
//
//   knn_partial_kernel   exact self k-NN (k <= 32) of a cloud under FLANN's float L2_Simple metric      APD:364
//   cov_finalize_kernel  merge partial lists, 3x3 covariance in fp64, regularisation, geo weight        APD:366-407, 266-269
//   nn_search_kernel     float transform + exact 1-NN of every source point in the target                APD:164-180
//   linearize_kernel     gate, sensor covariance, Mahalanobis matrix, residual, J^T O J / J^T O e, error  APD:183-218, 247-295
//   lm_solve_kernel      reduce partials, LM / GN step (6x6 LDLT, so3_exp), error trials, convergence     LSQ:67-76, 107-173; APD:310-346
//
// Design notes (MI355X):
//  * Point clouds are SoA float arrays in HBM.  The all-pairs searches are FP32-VALU bound, not HBM bound (a 16k x 16k
//    search is 2.7e8 distance evaluations over 200 KB of coordinates): one lane owns one query point, and the candidate
//    coordinates are read with WAVE-UNIFORM addresses, so they travel through the scalar data cache into SGPRs
//    (s_load_dwordx8/x16) and enter the VALU as free scalar operands -- no LDS round trip, no vector-memory traffic in
//    the inner loop.  The candidate range is split over blockIdx.y so that a single 16k-point scan still yields
//    thousands of waves for 256 CUs; partial results meet in a packed (distance bits, index) 64-bit atomicMin, which is
//    also what makes ties resolve to the LOWEST index (the definition the parity tests check).
//  * The distance expression is kept un-fused (-ffp-contract=off): d = ((dx*dx) + dy*dy) + dz*dz, float, so the indices
//    are bit-identical to FLANN's L2_Simple on the host.
//  * Everything after the search is fp64 (the reference accumulates H, b and the error in double): per-lane 28
//    accumulators (21 upper-triangular H + 6 b + 1 error), wave reduction with DPP-class shuffles, one LDS hop per
//    block, block partials to HBM and a deterministic tree in lm_solve_kernel (no float atomics: results are
//    run-to-run reproducible).
//  * The whole Gauss-Newton / Levenberg-Marquardt loop state lives on the device (PairState); the host only enqueues
//    kernels and polls a flag every few iterations, so a batch of scan pairs advances in lock-step with no per-iteration
//    host round trip.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "apd_device.h"

namespace gorio {

// ----------------------------------------------------------------------------------------------- helpers

// Candidate coordinates are read through the CONSTANT address space: with a wave-uniform index the compiler then emits
// s_load_dwordx8 into SGPRs (scalar data cache) and the VALU takes them as scalar operands.  Legal because no kernel
// writes the coordinate arrays it searches; the scalar cache is invalidated at every kernel boundary.
typedef const float __attribute__((address_space(4)))* scalar_fp;
__device__ __forceinline__ scalar_fp as_scalar(const float* p) { return (scalar_fp)p; }

__device__ __forceinline__ float sqdist3(float qx, float qy, float qz, float tx, float ty, float tz) {
  // FLANN L2_Simple<float>: result += diff*diff over 3 dims, no FMA (this file is built with -ffp-contract=off)
  float dx = qx - tx, dy = qy - ty, dz = qz - tz;
  float r = dx * dx;
  r = r + dy * dy;
  r = r + dz * dz;
  return r;
}

// Eigen Isometry3f * Vector4f (APD:176): ((m0*x + m1*y) + m2*z) + m3, float, no FMA
__device__ __forceinline__ void transform_f(const float* __restrict__ Tf, float x, float y, float z, float& qx, float& qy, float& qz) {
  float a = Tf[0] * x;
  a = a + Tf[1] * y;
  a = a + Tf[2] * z;
  qx = a + Tf[3];
  a = Tf[4] * x;
  a = a + Tf[5] * y;
  a = a + Tf[6] * z;
  qy = a + Tf[7];
  a = Tf[8] * x;
  a = a + Tf[9] * y;
  a = a + Tf[10] * z;
  qz = a + Tf[11];
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// Cross-lane plumbing for the 28-value reduction of linearize_kernel.  gfx950 has v_permlane32_swap / v_permlane16_swap: ONE
// instruction exchanges the upper half of one register with the lower half of another, so after "swap, add" a register holds value
// A summed over lane pairs in one half of the wave and value B in the other -- two values are folded per add and nothing is
// selected.  The pairing (l, l+32), (l, l+16), then rotations by 8, 4, 2, 1 inside a row is the tree of wave_sum, so the sums are
// bit-identical to 28 separate wave_sum calls; they cost 49 adds + 98 register moves instead of 168 adds + 336 ds_bpermute.
__device__ __forceinline__ void lane_swap32(double& a, double& b) {
  const unsigned long long ua = (unsigned long long)__double_as_longlong(a), ub = (unsigned long long)__double_as_longlong(b);
  const auto lo = __builtin_amdgcn_permlane32_swap((unsigned int)ua, (unsigned int)ub, false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap((unsigned int)(ua >> 32), (unsigned int)(ub >> 32), false, false);
  a = __longlong_as_double((long long)(((unsigned long long)hi[0] << 32) | lo[0]));
  b = __longlong_as_double((long long)(((unsigned long long)hi[1] << 32) | lo[1]));
}
__device__ __forceinline__ void lane_swap16(double& a, double& b) {
  const unsigned long long ua = (unsigned long long)__double_as_longlong(a), ub = (unsigned long long)__double_as_longlong(b);
  const auto lo = __builtin_amdgcn_permlane16_swap((unsigned int)ua, (unsigned int)ub, false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap((unsigned int)(ua >> 32), (unsigned int)(ub >> 32), false, false);
  a = __longlong_as_double((long long)(((unsigned long long)hi[0] << 32) | lo[0]));
  b = __longlong_as_double((long long)(((unsigned long long)hi[1] << 32) | lo[1]));
}
template <int CTRL>  // 0x120 + n: row_ror:n (rotation inside a row of 16 lanes)
__device__ __forceinline__ double row_rotate(double v) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  const unsigned int lo = __builtin_amdgcn_update_dpp(0u, (unsigned int)u, CTRL, 0xf, 0xf, false);
  const unsigned int hi = __builtin_amdgcn_update_dpp(0u, (unsigned int)(u >> 32), CTRL, 0xf, 0xf, false);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// 28 per-lane values -> 28 wave sums: afterwards out[k] (k < 7) of every lane in row r (lanes 16 r .. 16 r + 15) is the sum of acc[7 r + k]
__device__ __forceinline__ void wave_sum28(const double (&acc)[28], double (&out)[7]) {
  double c1[14];
#pragma unroll
  for (int k = 0; k < 14; ++k) {
    double a = acc[k], b = acc[14 + k];
    lane_swap32(a, b);
    c1[k] = a + b;  // lanes 0-31: acc[k] over (l, l+32); lanes 32-63: acc[14+k]
  }
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    double a = c1[k], b = c1[7 + k];
    lane_swap16(a, b);
    double v = a + b;  // rows 0..3: acc[k], acc[7+k], acc[14+k], acc[21+k] over the four lanes (l mod 16) + 16 j
    v += row_rotate<0x128>(v);
    v += row_rotate<0x124>(v);
    v += row_rotate<0x122>(v);
    v += row_rotate<0x121>(v);
    out[k] = v;
  }
}

// 1 / sqrt(x) and sqrt(x) for x > 0 in fp64 from v_rsq_f64 (about 2^-26) and two Newton steps (full precision up to an ulp or two);
// used where the reference takes a double sqrt whose result only enters 1e-9-tolerance arithmetic
__device__ __forceinline__ double rsqrt_newton(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = y * (1.5 - 0.5 * x * y * y);
  y = y * (1.5 - 0.5 * x * y * y);
  return y;
}

// ----------------------------------------------------------------------------------------------- k-NN (self)

// Sorted insertion of (cd, ci) into an ascending list held in registers (static indexing only -> stays in VGPRs).
// Strict '<' keeps earlier (== lower index) candidates ahead of equal distances.
template <int K>
__device__ __forceinline__ void topk_insert(float (&bd)[K], int (&bi)[K], float cd, int ci) {
  bool ins = false;
#pragma unroll
  for (int t = 0; t < K; ++t) {
    ins = ins || (cd < bd[t]);
    float td = bd[t];
    int ti = bi[t];
    bd[t] = ins ? cd : td;
    bi[t] = ins ? ci : ti;
    cd = ins ? td : cd;
    ci = ins ? ti : ci;
  }
}

// grid: (ceil(n/256), splits, clouds).  Each lane scans candidates [j0, j1) of its own cloud and writes its K best
// (sorted by distance, ties by index) to part_d / part_i laid out [split][t][i] (coalesced in i).
template <int K>
__global__ __launch_bounds__(256) void knn_partial_kernel(const KnnJob* __restrict__ jobs) {
  const KnnJob job = jobs[blockIdx.z];
  const int n = job.cloud.n;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (blockIdx.x * 256 >= n) return;
  const int split = blockIdx.y;
  if (split >= job.splits) return;
  const int j0 = split * job.chunk_len;
  int j1 = j0 + job.chunk_len;
  if (j1 > job.cloud.n_pad) j1 = job.cloud.n_pad;
  const scalar_fp cx = as_scalar(job.cloud.x);
  const scalar_fp cy = as_scalar(job.cloud.y);
  const scalar_fp cz = as_scalar(job.cloud.z);
  const int iq = i < n ? i : n - 1;
  const float qx = job.cloud.x[iq], qy = job.cloud.y[iq], qz = job.cloud.z[iq];

  float bd[K];
  int bi[K];
#pragma unroll
  for (int t = 0; t < K; ++t) {
    bd[t] = INFINITY;
    bi[t] = 0x7fffffff;
  }
  for (int j = j0; j < j1; j += 8) {  // chunk_len and n_pad are multiples of 16; padding points sit at 1e30 -> d = inf
    float d[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) d[u] = sqdist3(qx, qy, qz, cx[j + u], cy[j + u], cz[j + u]);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (d[u] < bd[K - 1]) topk_insert<K>(bd, bi, d[u], j + u);
    }
  }
  if (i < n) {
    const size_t base = (size_t)split * K * n + i;
#pragma unroll
    for (int t = 0; t < K; ++t) {
      job.part_d[base + (size_t)t * n] = bd[t];
      job.part_i[base + (size_t)t * n] = bi[t];
    }
  }
}

// One Jacobi rotation of a symmetric 3x3 in the (p,q) plane; r is the remaining index.  Classical closed form
// (a_pp -= t a_pq, a_qq += t a_pq, a_pq = 0), eigenvectors accumulated in the columns p, q of v.
__device__ __forceinline__ void jacobi_rot(double& app, double& aqq, double& apq, double& apr, double& aqr, double& v0p, double& v1p, double& v2p, double& v0q, double& v1q, double& v2q) {
  if (apq == 0.0) return;
  const double g = 100.0 * fabs(apq);
  if (fabs(app) + g == fabs(app) && fabs(aqq) + g == fabs(aqq)) {
    apq = 0.0;
    return;
  }
  const double theta = (aqq - app) / (2.0 * apq);
  double t = 1.0 / (fabs(theta) + sqrt(theta * theta + 1.0));
  if (theta < 0.0) t = -t;
  const double c = 1.0 / sqrt(t * t + 1.0);
  const double s = t * c;
  app = app - t * apq;
  aqq = aqq + t * apq;
  apq = 0.0;
  const double npr = c * apr - s * aqr;
  const double nqr = s * apr + c * aqr;
  apr = npr;
  aqr = nqr;
  double a, b;
  a = v0p; b = v0q; v0p = c * a - s * b; v0q = s * a + c * b;
  a = v1p; b = v1q; v1p = c * a - s * b; v1q = s * a + c * b;
  a = v2p; b = v2q; v2p = c * a - s * b; v2q = s * a + c * b;
}

// eigen-decomposition of the symmetric matrix (a00 a01 a02; . a11 a12; . . a22): w[] descending, V columns = vectors
struct Eig3 {
  double w0, w1, w2;
  double v00, v10, v20;  // column 0
  double v01, v11, v21;  // column 1
  double v02, v12, v22;  // column 2
};

__device__ __forceinline__ void swap_cols(double& wa, double& wb, double& a0, double& a1, double& a2, double& b0, double& b1, double& b2) {
  double t;
  t = wa; wa = wb; wb = t;
  t = a0; a0 = b0; b0 = t;
  t = a1; a1 = b1; b1 = t;
  t = a2; a2 = b2; b2 = t;
}

__device__ __forceinline__ Eig3 sym3_eigen(double a00, double a01, double a02, double a11, double a12, double a22) {
  Eig3 e;
  e.v00 = 1; e.v10 = 0; e.v20 = 0;
  e.v01 = 0; e.v11 = 1; e.v21 = 0;
  e.v02 = 0; e.v12 = 0; e.v22 = 1;
#pragma unroll 1
  for (int sweep = 0; sweep < 12; ++sweep) {
    if (a01 == 0.0 && a02 == 0.0 && a12 == 0.0) break;
    jacobi_rot(a00, a11, a01, a02, a12, e.v00, e.v10, e.v20, e.v01, e.v11, e.v21);  // (0,1), r = 2
    jacobi_rot(a00, a22, a02, a01, a12, e.v00, e.v10, e.v20, e.v02, e.v12, e.v22);  // (0,2), r = 1
    jacobi_rot(a11, a22, a12, a01, a02, e.v01, e.v11, e.v21, e.v02, e.v12, e.v22);  // (1,2), r = 0
  }
  e.w0 = a00; e.w1 = a11; e.w2 = a22;
  if (e.w0 < e.w1) swap_cols(e.w0, e.w1, e.v00, e.v10, e.v20, e.v01, e.v11, e.v21);
  if (e.w1 < e.w2) swap_cols(e.w1, e.w2, e.v01, e.v11, e.v21, e.v02, e.v12, e.v22);
  if (e.w0 < e.w1) swap_cols(e.w0, e.w1, e.v00, e.v10, e.v20, e.v01, e.v11, e.v21);
  return e;
}

// sigma_3 / sigma_1 of a symmetric 3x3 (APD:266-269): singular values of a symmetric matrix are |eigenvalues|
__device__ __forceinline__ double geo_weight(double c00, double c01, double c02, double c11, double c12, double c22) {
  Eig3 e = sym3_eigen(c00, c01, c02, c11, c12, c22);
  const double s0 = fabs(e.w0), s1 = fabs(e.w1), s2 = fabs(e.w2);
  const double smax = fmax(s0, fmax(s1, s2)), smin = fmin(s0, fmin(s1, s2));
  return smin / smax;
}

__device__ __forceinline__ void inv_sym3(double a00, double a01, double a02, double a11, double a12, double a22, double& i00, double& i01, double& i02, double& i11, double& i12, double& i22) {
  const double c00 = a11 * a22 - a12 * a12;
  const double c01 = a02 * a12 - a01 * a22;
  const double c02 = a01 * a12 - a02 * a11;
  const double det = a00 * c00 + a01 * c01 + a02 * c02;
  const double r = 1.0 / det;
  i00 = c00 * r;
  i01 = c01 * r;
  i02 = c02 * r;
  i11 = (a00 * a22 - a02 * a02) * r;
  i12 = (a01 * a02 - a00 * a12) * r;
  i22 = (a00 * a11 - a01 * a01) * r;
}

// APD:366-407 for one point from its final neighbour list (bd, bi sorted by distance then index): 3x3 covariance in fp64 in the
// list order (bit-identical sums to the host restatement), regularisation, geo weight; results stored at original index i.
template <int K>
__device__ __forceinline__ void covariance_from_list(const KnnJob& job, int i, const float (&bd)[K], const int (&bi)[K]) {
  const int k = job.k;
  (void)bd;
  if (job.knn_out) {
#pragma unroll
    for (int t = 0; t < K; ++t)
      if (t < k) job.knn_out[(size_t)i * k + t] = bi[t];
  }
  // APD:366-372 in the neighbour order of the list (same order as the oracle => bit-identical sums).  The k neighbours are gathered
  // ONCE, one 16-byte access each from the (x, y, z, label) copy of the cloud, and kept in registers for both passes.
  const float4* __restrict__ pts = job.cloud.p4;
  float px[K], py[K], pz[K];
#pragma unroll
  for (int t = 0; t < K; ++t) {
    const float4 v = pts[t < k ? bi[t] : bi[0]];
    px[t] = v.x; py[t] = v.y; pz[t] = v.z;
  }
  double mx = 0.0, my = 0.0, mz = 0.0;
#pragma unroll
  for (int t = 0; t < K; ++t)
    if (t < k) {
      mx += (double)px[t];
      my += (double)py[t];
      mz += (double)pz[t];
    }
  const double kd = (double)k;
  mx /= kd;
  my /= kd;
  mz /= kd;
  double c00 = 0, c01 = 0, c02 = 0, c11 = 0, c12 = 0, c22 = 0;
#pragma unroll
  for (int t = 0; t < K; ++t)
    if (t < k) {
      const double dx = (double)px[t] - mx, dy = (double)py[t] - my, dz = (double)pz[t] - mz;
      c00 += dx * dx;
      c01 += dx * dy;
      c02 += dx * dz;
      c11 += dy * dy;
      c12 += dy * dz;
      c22 += dz * dz;
    }
  c00 /= kd; c01 /= kd; c02 /= kd; c11 /= kd; c12 /= kd; c22 /= kd;

  double r00, r01, r02, r11, r12, r22;
  double geo = 0.0;
  bool have_geo = false;
  if (job.regularization == 0) {  // NONE, APD:374-376
    r00 = c00; r01 = c01; r02 = c02; r11 = c11; r12 = c12; r22 = c22;
  } else if (job.regularization == 4) {  // FROBENIUS, APD:377-383: (C_inv / ||C_inv||_F)^-1 with C = cov + 1e-3 I
    double i00, i01, i02, i11, i12, i22;
    inv_sym3(c00 + 1e-3, c01, c02, c11 + 1e-3, c12, c22 + 1e-3, i00, i01, i02, i11, i12, i22);
    const double fro = sqrt(i00 * i00 + i11 * i11 + i22 * i22 + 2.0 * (i01 * i01 + i02 * i02 + i12 * i12));
    inv_sym3(i00 / fro, i01 / fro, i02 / fro, i11 / fro, i12 / fro, i22 / fro, r00, r01, r02, r11, r12, r22);
  } else {  // SVD based, APD:384-406 (symmetric PSD: U == V == eigenvectors)
    Eig3 e = sym3_eigen(c00, c01, c02, c11, c12, c22);
    double s0 = fabs(e.w0), s1 = fabs(e.w1), s2 = fabs(e.w2);
    double l0, l1, l2;
    if (job.regularization == 3) {  // PLANE
      l0 = 1.0; l1 = 1.0; l2 = 1e-3;
    } else if (job.regularization == 1) {  // MIN_EIG
      l0 = fmax(s0, 1e-3); l1 = fmax(s1, 1e-3); l2 = fmax(s2, 1e-3);
    } else {  // NORMALIZED_MIN_EIG
      const double smax = fmax(s0, fmax(s1, s2));
      l0 = fmax(s0 / smax, 1e-3); l1 = fmax(s1 / smax, 1e-3); l2 = fmax(s2 / smax, 1e-3);
    }
    r00 = e.v00 * l0 * e.v00 + e.v01 * l1 * e.v01 + e.v02 * l2 * e.v02;
    r01 = e.v00 * l0 * e.v10 + e.v01 * l1 * e.v11 + e.v02 * l2 * e.v12;
    r02 = e.v00 * l0 * e.v20 + e.v01 * l1 * e.v21 + e.v02 * l2 * e.v22;
    r11 = e.v10 * l0 * e.v10 + e.v11 * l1 * e.v11 + e.v12 * l2 * e.v12;
    r12 = e.v10 * l0 * e.v20 + e.v11 * l1 * e.v21 + e.v12 * l2 * e.v22;
    r22 = e.v20 * l0 * e.v20 + e.v21 * l1 * e.v21 + e.v22 * l2 * e.v22;
    // APD:266-269 take sigma_3 / sigma_1 of THIS matrix; it was just assembled as V diag(l) V^T with orthonormal V, so its singular
    // values are l0, l1, l2 (to rounding: the second decomposition the reference runs per point per call can differ by ~1e-16 relative,
    // in a weight that only scales the LM acceptance error)
    geo = fmin(l0, fmin(l1, l2)) / fmax(l0, fmax(l1, l2));
    have_geo = true;
  }
  double* o = job.cloud.cov6 + (size_t)i * 6;
  o[0] = r00; o[1] = r01; o[2] = r02; o[3] = r11; o[4] = r12; o[5] = r22;
  job.cloud.geo_w[i] = have_geo ? geo : geo_weight(r00, r01, r02, r11, r12, r22);
}

// grid: (ceil(n/256), 1, clouds).  Merges the per-split lists, then APD:366-407 per point.
template <int K>
__global__ __launch_bounds__(256) void cov_finalize_kernel(const KnnJob* __restrict__ jobs) {
  const KnnJob job = jobs[blockIdx.z];
  const int n = job.cloud.n;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float bd[K];
  int bi[K];
#pragma unroll
  for (int t = 0; t < K; ++t) {
    bd[t] = INFINITY;
    bi[t] = 0x7fffffff;
  }
  // splits ascend in index range and each list is sorted by (distance, index): strict '<' insertion keeps (d, idx) order
  for (int s = 0; s < job.splits; ++s) {
    const size_t base = (size_t)s * K * n + i;
    for (int t = 0; t < K; ++t) {
      const float d = job.part_d[base + (size_t)t * n];
      if (!(d < bd[K - 1])) break;  // the rest of this list is no better
      topk_insert<K>(bd, bi, d, job.part_i[base + (size_t)t * n]);
    }
  }
  covariance_from_list<K>(job, i, bd, bi);
}

// geo weights for covariances supplied through setSourceCovariances / setTargetCovariances
__global__ __launch_bounds__(256) void geo_weight_kernel(const double* __restrict__ cov6, double* __restrict__ geo_w, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double* c = cov6 + (size_t)i * 6;
  geo_w[i] = geo_weight(c[0], c[1], c[2], c[3], c[4], c[5]);
}

// ----------------------------------------------------------------------------------------------- 1-NN correspondences

// grid: (ceil(max_n/256), splits, pairs).  One lane = one source point; candidates [j0, j1) of the target arrive through
// the scalar cache.  Result: atomicMin of (float bits of d) << 32 | j  -- d >= 0 so the bit pattern orders like the value,
// and equal distances fall through to the lower index.
__global__ __launch_bounds__(256) void nn_search_kernel(const PairDesc* __restrict__ descs) {
  const PairDesc& pd = descs[blockIdx.z];
  const PairState* __restrict__ st = pd.state;
  if (st->done) return;
  const int n = pd.src.n;
  if (blockIdx.x * 256 >= n) return;
  if ((int)(blockIdx.x * 256) < pd.shard_lo || (int)(blockIdx.x * 256) >= pd.shard_hi) return;  // another rank's part of the source
  const int split = blockIdx.y;
  if (split >= pd.nn_splits) return;
  const int j0 = split * pd.nn_chunk;
  int j1 = j0 + pd.nn_chunk;
  if (j1 > pd.tgt.n_pad) j1 = pd.tgt.n_pad;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int iq = i < n ? i : n - 1;
  float qx, qy, qz;
  transform_f(st->Tf, pd.src.x[iq], pd.src.y[iq], pd.src.z[iq], qx, qy, qz);
  const scalar_fp tx = as_scalar(pd.tgt.x);
  const scalar_fp ty = as_scalar(pd.tgt.y);
  const scalar_fp tz = as_scalar(pd.tgt.z);
  // v_cmp / v_cndmask / v_min issue at HALF the rate of v_sub / v_mul / v_add on gfx950 (tools/valu_rate.hip): a compare and two selects
  // per candidate cost as much as six of the eight distance instructions.  So the loop only tracks the minimum of every 16-candidate
  // chunk (a v_min3 tree: half an instruction per candidate) and which chunk held the best one; the index inside that chunk is
  // recovered once at the end.  Strict '<' between chunks and "first equal" inside the chunk keep ties on the LOWEST index.
  float best = INFINITY;
  int bc = 0x7fffffff;  // first candidate of the chunk that holds the best distance
  for (int j = j0; j < j1; j += 16) {
    float d[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) d[u] = sqdist3(qx, qy, qz, tx[j + u], ty[j + u], tz[j + u]);
    const float m0 = fminf(fminf(d[0], d[1]), d[2]), m1 = fminf(fminf(d[3], d[4]), d[5]), m2 = fminf(fminf(d[6], d[7]), d[8]);
    const float m3 = fminf(fminf(d[9], d[10]), d[11]), m4 = fminf(fminf(d[12], d[13]), d[14]);
    const float m = fminf(fminf(fminf(m0, m1), m2), fminf(fminf(m3, m4), d[15]));
    const bool lt = m < best;
    best = lt ? m : best;
    bc = lt ? j : bc;
  }
  int bj = 0x7fffffff;
  if (bc != 0x7fffffff) {
#pragma unroll
    for (int u = 15; u >= 0; --u) {  // descending, so the lowest matching index is the one that stays
      const float d = sqdist3(qx, qy, qz, pd.tgt.x[bc + u], pd.tgt.y[bc + u], pd.tgt.z[bc + u]);
      bj = d == best ? bc + u : bj;
    }
  }
  if (i < n && bj != 0x7fffffff) {
    const unsigned long long key = ((unsigned long long)__float_as_uint(best) << 32) | (unsigned int)bj;
    atomicMin(pd.best_key + i, key);
  }
}

// ----------------------------------------------------------------------------------------------- linearize

struct PointTerms {
  double e0, e1, e2;     // residual b - T a
  double a0, a1, a2;     // T a
  double o00, o01, o02, o11, o12, o22;  // Mahalanobis 3x3 block
};

// error / residual at pose T (row-major 3x4 part) for source point a and target point b with Mahalanobis block o
__device__ __forceinline__ double residual_terms(const double* __restrict__ T, float ax, float ay, float az, float bx, float by, float bz, PointTerms& p) {
#pragma clang fp contract(fast)
  const double x = (double)ax, y = (double)ay, z = (double)az;
  p.a0 = T[0] * x + T[1] * y + T[2] * z + T[3];
  p.a1 = T[4] * x + T[5] * y + T[6] * z + T[7];
  p.a2 = T[8] * x + T[9] * y + T[10] * z + T[11];
  p.e0 = (double)bx - p.a0;
  p.e1 = (double)by - p.a1;
  p.e2 = (double)bz - p.a2;
  const double m0 = p.o00 * p.e0 + p.o01 * p.e1 + p.o02 * p.e2;
  const double m1 = p.o01 * p.e0 + p.o11 * p.e1 + p.o12 * p.e2;
  const double m2 = p.o02 * p.e0 + p.o12 * p.e1 + p.o22 * p.e2;
  return p.e0 * m0 + p.e1 * m1 + p.e2 * m2;
}

__device__ void gn_step_tail(const PairDesc& pd, const ApdConsts& cst);

__device__ __forceinline__ float sumsq2_f(float a, float b) {  // x^2 + y^2 in float, un-fused (APD:198 squares floats)
  float r = a * a;
  r = r + b * b;
  return r;
}

// grid: (ceil(max_n/256), 1, pairs).  Consumes (and re-arms) best_key, writes corr / sqd / omega6 and one 28-double
// partial per block: [0..20] upper triangle of H row-major, [21..26] b, [27] weighted error.
// fuse != 0 (Gauss-Newton aligns): the workgroup that stores the LAST partial of its pair goes on to run the optimiser step
// (gn_step_tail) -- the partials are summed in block order whoever sums them, so the result is the one lm_solve_kernel gives, one launch earlier.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 6))) void linearize_kernel(const PairDesc* __restrict__ descs, ApdConsts cst, int fuse) {
  const GridPos gp = xcd_grid_pos();  // pair -> XCD (apd_device.h)
  const PairDesc& pd = descs[gp.z];
  PairState* __restrict__ st = pd.state;
  // the optimiser state is constant until the LAST workgroup of the launch rewrites it (gn_step_tail, after every other workgroup has
  // arrived): read it through the scalar cache.  Through the generic pointer these would be flat loads, whose completion the compiler
  // can only wait for together with every other outstanding load.
  const __attribute__((address_space(4))) PairState* sst = (const __attribute__((address_space(4))) PairState*)pd.state;
  if (sst->done) return;
  const int n = pd.src.n;
  if (gp.x * 256 >= (unsigned int)n) return;
  const int i = gp.x * 256 + threadIdx.x;

  double acc[28];
#pragma unroll
  for (int t = 0; t < 28; ++t) acc[t] = 0.0;

  if (i < n) {
    // every load of this point is issued up front through the global address space -- the key first (the target gathers depend on
    // it), then the source side, which does not -- and unconditionally (indices clamped): a load inside a branch makes the compiler
    // wait for all outstanding loads at the join.
    typedef const __attribute__((address_space(1))) unsigned long long* g_u64p;
    typedef const __attribute__((address_space(1))) double* g_f64p;
    typedef const __attribute__((address_space(1))) float* g_f32p;
    const unsigned long long key = ((g_u64p)pd.best_key)[i];
    const float ax = ((g_f32p)pd.src.x)[i], ay = ((g_f32p)pd.src.y)[i], az = ((g_f32p)pd.src.z)[i];
    const float src_label = ((g_f32p)pd.src.label)[i];
    const double src_geo_w = ((g_f64p)pd.src.geo_w)[i];
    double cA[6];
#pragma unroll
    for (int t = 0; t < 6; ++t) cA[t] = ((g_f64p)pd.src.cov6)[(size_t)i * 6 + t];
    pd.best_key[i] = ~0ull;  // re-arm for the next search
    const float d = __uint_as_float((unsigned int)(key >> 32));
    int j = (int)(unsigned int)(key & 0xffffffffu);
    const bool found = key != ~0ull;
    pd.sqd[i] = found ? d : INFINITY;                       // APD:180
    if (!found || !((double)d < cst.thr2)) j = -1;          // APD:183
    pd.corr[i] = j;
    const int jj = j >= 0 ? j : 0;
    double cB[6];
#pragma unroll
    for (int t = 0; t < 6; ++t) cB[t] = ((g_f64p)pd.tgt.cov6)[(size_t)jj * 6 + t];
    float4 tb;  // the matched target point and its cluster label in one 16-byte gather
    {
      typedef float v4f_lin __attribute__((ext_vector_type(4)));
      const v4f_lin t4 = ((const __attribute__((address_space(1))) v4f_lin*)pd.tgt.p4)[jj];
      tb = make_float4(t4.x, t4.y, t4.z, t4.w);
    }
    double* om = pd.omega6 + (size_t)i * 6;
    if (j >= 0) {
      // fp64 from here on is tolerance arithmetic (H, b, error agree with the host restatement to 1e-9): let the compiler fuse
      // multiply-adds in THIS block (the build is -ffp-contract=off for the float search arithmetic, which lives in functions of its own)
#pragma clang fp contract(fast)
      double T[12];
      float Tf[12];
#pragma unroll
      for (int t = 0; t < 12; ++t) {
        T[t] = sst->x0[t];
        Tf[t] = sst->Tf[t];
      }
      float qx, qy, qz;
      transform_f(Tf, ax, ay, az, qx, qy, qz);
      // sensor covariance at the transformed point, APD:194-210
      const double px = (double)qx, py = (double)qy, pz = (double)qz;
      const double pxy2 = px * px + py * py;
      const double dist = sqrt(pxy2 + pz * pz);
      const double s_x = dist * cst.dist_var / 400;
      const double s_y = dist * cst.sin_az;
      const double s_z = dist * cst.sin_el;
      const float rxy = (float)sqrt((double)sumsq2_f(qx, qy));  // sqrt(float) overload, correctly rounded
      // The reference's angles are FLOATS: elevation = atan2(float, float), azimuth likewise (APD:198-199), fed to double sin / cos.
      // sin and cos of the un-rounded angle are ratios of the operands (no trigonometry); the float rounding moves the angle by
      // dr = (double)(float)angle - angle, |dr| < 2^-24 * pi, and sin(a + dr), cos(a + dr) follow from the addition theorems with a
      // two-term series in dr (next terms < 1e-28).  Saves both fp64 sincos evaluations; agrees with them to an ulp or two.
      double ce, se, ca, sa;
      {
        const double rd = (double)rxy;
        const double el = atan2(rd, pz), az = atan2(py, px);
        const double del = (double)(float)el - el, daz = (double)(float)az - az;
        const double h2 = rd * rd + pz * pz;
        const double ih = h2 > 0.0 ? rsqrt_newton(h2) : 0.0, ir = pxy2 > 0.0 ? rsqrt_newton(pxy2) : 0.0;
        const double se0 = rd * ih, ce0 = h2 > 0.0 ? pz * ih : 1.0;      // atan2(0, 0) = 0
        const double sa0 = py * ir, ca0 = pxy2 > 0.0 ? px * ir : 1.0;
        const double sde = del - del * del * del * (1.0 / 6.0), cde = 1.0 - 0.5 * del * del;
        const double sda = daz - daz * daz * daz * (1.0 / 6.0), cda = 1.0 - 0.5 * daz * daz;
        se = se0 * cde + ce0 * sde;
        ce = ce0 * cde - se0 * sde;
        sa = sa0 * cda + ca0 * sda;
        ca = ca0 * cda - sa0 * sda;
      }
      // A = Rz(az) Ry(el) diag(s): columns of R scaled
      const double A00 = ca * ce * s_x, A01 = -sa * s_y, A02 = ca * se * s_z;
      const double A10 = sa * ce * s_x, A11 = ca * s_y, A12 = sa * se * s_z;
      const double A20 = -se * s_x, A22 = ce * s_z;  // A21 = 0
      const double r00 = A00 * A00 + A01 * A01 + A02 * A02;
      const double r01 = A00 * A10 + A01 * A11 + A02 * A12;
      const double r02 = A00 * A20 + A02 * A22;
      const double r11 = A10 * A10 + A11 * A11 + A12 * A12;
      const double r12 = A10 * A20 + A12 * A22;
      const double r22 = A20 * A20 + A22 * A22;
      // RCR = (C_B + cov_r) + R (C_A + cov_r) R^T, APD:213-214 (3x3 block; row/col 3 of the 4x4 decouple, APD:215-218)
      const double a00 = cA[0] + r00, a01 = cA[1] + r01, a02 = cA[2] + r02, a11 = cA[3] + r11, a12 = cA[4] + r12, a22 = cA[5] + r22;
      // M = R * Asym
      const double R00 = T[0], R01 = T[1], R02 = T[2], R10 = T[4], R11 = T[5], R12 = T[6], R20 = T[8], R21 = T[9], R22 = T[10];
      const double M00 = R00 * a00 + R01 * a01 + R02 * a02, M01 = R00 * a01 + R01 * a11 + R02 * a12, M02 = R00 * a02 + R01 * a12 + R02 * a22;
      const double M10 = R10 * a00 + R11 * a01 + R12 * a02, M11 = R10 * a01 + R11 * a11 + R12 * a12, M12 = R10 * a02 + R11 * a12 + R12 * a22;
      const double M20 = R20 * a00 + R21 * a01 + R22 * a02, M21 = R20 * a01 + R21 * a11 + R22 * a12, M22 = R20 * a02 + R21 * a12 + R22 * a22;
      const double q00 = (cB[0] + r00) + (M00 * R00 + M01 * R01 + M02 * R02);
      const double q01 = (cB[1] + r01) + (M00 * R10 + M01 * R11 + M02 * R12);
      const double q02 = (cB[2] + r02) + (M00 * R20 + M01 * R21 + M02 * R22);
      const double q11 = (cB[3] + r11) + (M10 * R10 + M11 * R11 + M12 * R12);
      const double q12 = (cB[4] + r12) + (M10 * R20 + M11 * R21 + M12 * R22);
      const double q22 = (cB[5] + r22) + (M20 * R20 + M21 * R21 + M22 * R22);
      PointTerms p;
      inv_sym3(q00, q01, q02, q11, q12, q22, p.o00, p.o01, p.o02, p.o11, p.o12, p.o22);  // APD:217
      if (pd.write_omega) {
        om[0] = p.o00; om[1] = p.o01; om[2] = p.o02; om[3] = p.o11; om[4] = p.o12; om[5] = p.o22;
      }

      const double quad = residual_terms(T, ax, ay, az, tb.x, tb.y, tb.z, p);  // APD:255-263
      const double w = 1.0 + src_geo_w + ((tb.w == src_label) ? 1.0 / (double)(pd.cl_points > 0 ? pd.cl_points : n) : 0.0);  // APD:266-276
      acc[27] = w * quad;

      // J = [skew(Ta) | -I], APD:284-287.  With S = skew(Ta): H_rr = S^T O S, H_rt = -S^T O, H_tt = O, b_r = S^T O e, b_t = -O e.
      // G = S^T O (3x3): rows of S^T are (0, a2, -a1), (-a2, 0, a0), (a1, -a0, 0)
      const double G00 = p.a2 * p.o01 - p.a1 * p.o02, G01 = p.a2 * p.o11 - p.a1 * p.o12, G02 = p.a2 * p.o12 - p.a1 * p.o22;
      const double G10 = -p.a2 * p.o00 + p.a0 * p.o02, G11 = -p.a2 * p.o01 + p.a0 * p.o12, G12 = -p.a2 * p.o02 + p.a0 * p.o22;
      const double G20 = p.a1 * p.o00 - p.a0 * p.o01, G21 = p.a1 * p.o01 - p.a0 * p.o11, G22 = p.a1 * p.o02 - p.a0 * p.o12;
      // H_rr = G S : columns of S are (0, a2, -a1), (-a2, 0, a0), (a1, -a0, 0)
      const double Hrr00 = G01 * p.a2 - G02 * p.a1, Hrr01 = -G00 * p.a2 + G02 * p.a0, Hrr02 = G00 * p.a1 - G01 * p.a0;
      const double Hrr11 = -G10 * p.a2 + G12 * p.a0, Hrr12 = G10 * p.a1 - G11 * p.a0;
      const double Hrr22 = G20 * p.a1 - G21 * p.a0;
      // upper triangle, row-major: (0,0..5) (1,1..5) (2,2..5) (3,3..5) (4,4..5) (5,5)
      acc[0] = Hrr00; acc[1] = Hrr01; acc[2] = Hrr02; acc[3] = -G00; acc[4] = -G01; acc[5] = -G02;
      acc[6] = Hrr11; acc[7] = Hrr12; acc[8] = -G10; acc[9] = -G11; acc[10] = -G12;
      acc[11] = Hrr22; acc[12] = -G20; acc[13] = -G21; acc[14] = -G22;
      acc[15] = p.o00; acc[16] = p.o01; acc[17] = p.o02;
      acc[18] = p.o11; acc[19] = p.o12;
      acc[20] = p.o22;
      const double oe0 = p.o00 * p.e0 + p.o01 * p.e1 + p.o02 * p.e2;
      const double oe1 = p.o01 * p.e0 + p.o11 * p.e1 + p.o12 * p.e2;
      const double oe2 = p.o02 * p.e0 + p.o12 * p.e1 + p.o22 * p.e2;
      acc[21] = G00 * p.e0 + G01 * p.e1 + G02 * p.e2;
      acc[22] = G10 * p.e0 + G11 * p.e1 + G12 * p.e2;
      acc[23] = G20 * p.e0 + G21 * p.e1 + G22 * p.e2;
      acc[24] = -oe0; acc[25] = -oe1; acc[26] = -oe2;
    } else {
      if (pd.write_omega) {
        om[0] = 0; om[1] = 0; om[2] = 0; om[3] = 0; om[4] = 0; om[5] = 0;
      }
    }
  }

  __shared__ double red[4][28];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  {
    double ws[7];
    wave_sum28(acc, ws);
    if ((lane & 15) == 0) {
#pragma unroll
      for (int k = 0; k < 7; ++k) red[wv][7 * (lane >> 4) + k] = ws[k];
    }
  }
  __syncthreads();
  if (threadIdx.x < 28) {
    const double s = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    if (!fuse) {
      pd.partials[(size_t)gp.x * 28 + threadIdx.x] = s;
    } else {
      // Hand-over to the last workgroup of the pair WITHOUT a device-scope fence: on this part a release fence writes back the
      // whole L2 of the XCD (the eight XCDs have private L2s), and 4096 workgroups doing that cost 0.6 ms per launch.  The partial
      // is stored write-through (agent-scope store, `sc1`), the wave waits for the store to be acknowledged, and only then is the
      // arrival counted; the consumer reads the partials with agent-scope loads.
      __hip_atomic_store(pd.partials + (size_t)gp.x * 28 + threadIdx.x, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (!fuse) return;
  __shared__ int s_last;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // the 28 stores above were issued by this wave (threads 0..27 and thread 0 share wave 0)
  if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(&st->arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned int)pd.nblk - 1u;
  __syncthreads();
  if (!s_last) return;
  if (threadIdx.x == 0) __hip_atomic_store(&st->arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-armed for the next launch
  gn_step_tail(pd, cst);
}

// ----------------------------------------------------------------------------------------------- LM / GN step

// Eigen::LDLT<6x6>(A).solve(rhs): LDL^T with symmetric diagonal pivoting.  Runs on one lane.  Pivoting indexes the matrix
// dynamically; the work area `ws` (>= 56 doubles, LDS) keeps that out of private memory -- a kernel with a scratch segment pays for it
// on every wave it launches, and linearize_kernel (which may end in this solve) launches sixteen thousand.
__device__ void ldlt6_solve(const double* __restrict__ A_in, const double* __restrict__ rhs, double* __restrict__ x, double* __restrict__ ws) {
  double* A = ws;             // [36]
  double* y = ws + 36;        // [6]
  double* col = ws + 42;      // [6]
  int* perm = reinterpret_cast<int*>(ws + 48);  // [6]
  for (int i = 0; i < 36; ++i) A[i] = A_in[i];
  for (int i = 0; i < 6; ++i) perm[i] = i;
  for (int k = 0; k < 6; ++k) {
    int piv = k;
    double best = fabs(A[k * 6 + k]);
    for (int i = k + 1; i < 6; ++i)
      if (fabs(A[i * 6 + i]) > best) {
        best = fabs(A[i * 6 + i]);
        piv = i;
      }
    if (piv != k) {
      for (int c = 0; c < 6; ++c) {
        double t = A[k * 6 + c];
        A[k * 6 + c] = A[piv * 6 + c];
        A[piv * 6 + c] = t;
      }
      for (int r = 0; r < 6; ++r) {
        double t = A[r * 6 + k];
        A[r * 6 + k] = A[r * 6 + piv];
        A[r * 6 + piv] = t;
      }
      int t = perm[k];
      perm[k] = perm[piv];
      perm[piv] = t;
    }
    const double d = A[k * 6 + k];
    if (d == 0.0) continue;
    for (int i = k + 1; i < 6; ++i) col[i] = A[i * 6 + k];
    for (int i = k + 1; i < 6; ++i) {
      const double l = col[i] / d;
      for (int j = k + 1; j <= i; ++j) A[i * 6 + j] -= l * col[j];
      A[i * 6 + k] = l;
    }
    for (int i = k + 1; i < 6; ++i)
      for (int j = i + 1; j < 6; ++j) A[i * 6 + j] = A[j * 6 + i];
  }
  for (int i = 0; i < 6; ++i) y[i] = rhs[perm[i]];
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < i; ++j) y[i] -= A[i * 6 + j] * y[j];
  for (int i = 0; i < 6; ++i) y[i] = (A[i * 6 + i] != 0.0) ? y[i] / A[i * 6 + i] : 0.0;
  for (int i = 5; i >= 0; --i)
    for (int j = i + 1; j < 6; ++j) y[i] -= A[j * 6 + i] * y[j];
  for (int i = 0; i < 6; ++i) x[perm[i]] = y[i];
}

// delta = [so3_exp(d[0:3]).toRotationMatrix() | d[3:6]] (SO3:59-78, LSQ:117-119 / 140-142); row-major 4x4
__device__ void delta_from_d(const double* __restrict__ d, double* __restrict__ delta) {
  const double theta_sq = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
  double imag, real;
  if (theta_sq < 1e-10) {
    const double theta_quad = theta_sq * theta_sq;
    imag = 0.5 - 1.0 / 48.0 * theta_sq + 1.0 / 3840.0 * theta_quad;
    real = 1.0 - 1.0 / 8.0 * theta_sq + 1.0 / 384.0 * theta_quad;
  } else {
    const double theta = sqrt(theta_sq);
    const double half = 0.5 * theta;
    imag = sin(half) / theta;
    real = cos(half);
  }
  const double w = real, x = imag * d[0], y = imag * d[1], z = imag * d[2];
  const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
  delta[0] = 1.0 - (tyy + tzz); delta[1] = txy - twz; delta[2] = txz + twy; delta[3] = d[3];
  delta[4] = txy + twz; delta[5] = 1.0 - (txx + tzz); delta[6] = tyz - twx; delta[7] = d[4];
  delta[8] = txz - twy; delta[9] = tyz + twx; delta[10] = 1.0 - (txx + tyy); delta[11] = d[5];
  delta[12] = 0; delta[13] = 0; delta[14] = 0; delta[15] = 1;
}

__device__ void isom_mul(const double* __restrict__ A, const double* __restrict__ B, double* __restrict__ C) {
  double t[16];  // static indices only (fully unrolled): stays in registers
#pragma unroll
  for (int r = 0; r < 3; ++r) {
#pragma unroll
    for (int c = 0; c < 3; ++c) t[r * 4 + c] = A[r * 4 + 0] * B[c] + A[r * 4 + 1] * B[4 + c] + A[r * 4 + 2] * B[8 + c];
    t[r * 4 + 3] = A[r * 4 + 0] * B[3] + A[r * 4 + 1] * B[7] + A[r * 4 + 2] * B[11] + A[r * 4 + 3];
  }
  t[12] = 0; t[13] = 0; t[14] = 0; t[15] = 1;
#pragma unroll
  for (int i = 0; i < 16; ++i) C[i] = t[i];
}

// is_converged, LSQ:83-92
__device__ bool is_converged(const double* __restrict__ delta, double inv_rot_eps, double inv_trans_eps) {
  double rmax = 0.0, tmax = 0.0;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const double v = inv_rot_eps * fabs(delta[r * 4 + c] - (r == c ? 1.0 : 0.0));
      rmax = v > rmax ? v : rmax;  // Eigen maxCoeff semantics for finite values
    }
    const double v = inv_trans_eps * fabs(delta[r * 4 + 3]);
    tmax = v > tmax ? v : tmax;
  }
  return fmax(rmax, tmax) < 1.0;
}

// compute_error (APD:310-346) over all source points of one pair by one workgroup; result valid on every thread
__device__ double block_error(const PairDesc& pd, const double* __restrict__ T, const ApdConsts& cst, double* __restrict__ sred) {
  const int n = pd.src.n;
  double sum = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int j = pd.corr[i];
    if (j < 0) continue;
    const double* om = pd.omega6 + (size_t)i * 6;
    PointTerms p;
    p.o00 = om[0]; p.o01 = om[1]; p.o02 = om[2]; p.o11 = om[3]; p.o12 = om[4]; p.o22 = om[5];
    const float4 tb = pd.tgt.p4[j];
    const double quad = residual_terms(T, pd.src.x[i], pd.src.y[i], pd.src.z[i], tb.x, tb.y, tb.z, p);
    const double w = 1.0 + pd.src.geo_w[i] + ((tb.w == pd.src.label[i]) ? 1.0 / (double)(pd.cl_points > 0 ? pd.cl_points : n) : 0.0);
    sum += w * quad;
  }
  sum = wave_sum(sum);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  __syncthreads();  // sred may still be read by a previous call
  if (lane == 0) sred[wv] = sum;
  __syncthreads();
  double tot = 0.0;
  const int nw = blockDim.x >> 6;
  for (int w = 0; w < nw; ++w) tot += sred[w];
  return tot;
}

// grid: (pairs), block 1024.  mode 0: full optimiser step (LSQ:67-76 body).  mode 1: only publish H, b, y0 (linearize API).
// mode 2: only evaluate the error at st->xi with the stored correspondences (compute_error API).
__device__ void lm_solve_body(const PairDesc& pd, const ApdConsts& cst, int mode) {
  PairState* __restrict__ st = pd.state;
  __shared__ double sH[36], sb[6], sxi[16], sdelta[16], sd[6], sws[56], sHl[36], snb[6];
  __shared__ double sy0, sred[16];
  __shared__ int sflag;
  if (mode == 2) {
    const double yi = block_error(pd, st->xi, cst, sred);
    if (threadIdx.x == 0) st->yi = yi;
    return;
  }
  if (st->done) return;

  // deterministic reduction of the block partials (APD:297-304)
  if (threadIdx.x < 28) {
    double s = 0.0;
    for (int bk0 = 0; bk0 < pd.nblk; bk0 += 16) {  // same ascending order as a plain loop, but sixteen loads in flight at a time
      double v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = (bk0 + u < pd.nblk) ? pd.partials[(size_t)(bk0 + u) * 28 + threadIdx.x] : 0.0;
#pragma unroll
      for (int u = 0; u < 16; ++u) s += v[u];
    }
    const int t = threadIdx.x;
    if (t < 21) {
      int r = 0, c = t;  // unpack upper-triangular index
      int rowlen = 6;
      while (c >= rowlen) {
        c -= rowlen;
        ++r;
        --rowlen;
      }
      c += r;
      sH[r * 6 + c] = s;
      sH[c * 6 + r] = s;
    } else if (t < 27) {
      sb[t - 21] = s;
    } else {
      sy0 = s;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int a = 0; a < 36; ++a) st->H[a] = sH[a];
    for (int a = 0; a < 6; ++a) st->b[a] = sb[a];
    st->y0 = sy0;
    st->n_linearize += 1;
  }
  if (mode == 1) return;

  const double inv_re = 1.0 / cst.rot_eps, inv_te = 1.0 / cst.trans_eps;
  int ok = 0;
  if (cst.optimizer == 0) {  // step_gn, LSQ:107-123
    if (threadIdx.x == 0) {
      for (int a = 0; a < 6; ++a) snb[a] = -sb[a];
      ldlt6_solve(sH, snb, sd, sws);
      delta_from_d(sd, sdelta);
      isom_mul(sdelta, st->x0, sxi);
      for (int a = 0; a < 16; ++a) st->x0[a] = sxi[a];
      for (int a = 0; a < 36; ++a) st->Hfin[a] = sH[a];
    }
    ok = 1;
  } else {  // step_lm, LSQ:127-173
    double lambda = st->lambda;
    if (lambda < 0.0) {  // LSQ:131-133
      double mx = 0.0;
      for (int a = 0; a < 6; ++a) mx = fmax(mx, fabs(sH[a * 6 + a]));
      lambda = cst.lm_init_lambda_factor * mx;
    }
    double nu = 2.0;
    for (int trial = 0; trial < cst.lm_max_iterations; ++trial) {
      if (threadIdx.x == 0) {
        for (int a = 0; a < 36; ++a) sHl[a] = sH[a];
        for (int a = 0; a < 6; ++a) {
          sHl[a * 6 + a] += lambda;
          snb[a] = -sb[a];
        }
        ldlt6_solve(sHl, snb, sd, sws);     // LSQ:137-138
        delta_from_d(sd, sdelta);           // LSQ:140-142
        isom_mul(sdelta, st->x0, sxi);      // LSQ:144
      }
      __syncthreads();
      const double yi = block_error(pd, sxi, cst, sred);  // LSQ:145
      if (threadIdx.x == 0) {
        st->n_error += 1;
        double den = 0.0;
        for (int a = 0; a < 6; ++a) den += sd[a] * (lambda * sd[a] - sb[a]);
        const double rho = (sy0 - yi) / den;  // LSQ:146
        if (rho < 0) {                        // LSQ:156-164
          sflag = is_converged(sdelta, inv_re, inv_te) ? 2 : 0;
        } else {
          sflag = 1;
          for (int a = 0; a < 16; ++a) st->x0[a] = sxi[a];                  // LSQ:166
          const double f = 1 - pow(2 * rho - 1, 3);
          st->lambda = lambda * fmax(1.0 / 3.0, f);                        // LSQ:167
          for (int a = 0; a < 36; ++a) st->Hfin[a] = sH[a];                 // LSQ:168
        }
      }
      __syncthreads();
      const int flag = sflag;
      if (flag == 1) {
        ok = 1;
        break;
      }
      if (flag == 2) {
        ok = 1;
        if (threadIdx.x == 0) st->lambda = lambda;
        break;
      }
      lambda = nu * lambda;  // LSQ:161-162 (every thread tracks the same scalars)
      nu = 2 * nu;
      if (threadIdx.x == 0) st->lambda = lambda;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int it = st->iter;
    st->nr_iterations = it;  // LSQ:68
    int done = 0, conv = 0;
    if (!ok) {
      done = 1;  // "lm not converged!!" LSQ:71-74
      st->lm_failed = 1;
    } else {
      conv = is_converged(sdelta, inv_re, inv_te) ? 1 : 0;  // LSQ:75
      if (conv) done = 1;
    }
    st->iter = it + 1;
    if (it + 1 >= cst.max_iterations) done = 1;
    st->converged = conv;
    for (int a = 0; a < 12; ++a) st->Tf[a] = (float)st->x0[a];  // APD:164 for the next search; LSQ:78 at the end
    __threadfence();
    st->done = done;
  }
}

__global__ __launch_bounds__(1024) void lm_solve_kernel(const PairDesc* __restrict__ descs, ApdConsts cst, int mode) {
  lm_solve_body(descs[blockIdx.x], cst, mode);
}

// The Gauss-Newton case of lm_solve_body (mode 0, optimizer 0) for the last workgroup of linearize_kernel: the same sums in the same
// order, the same step_gn (LSQ:107-123) and bookkeeping (LSQ:67-76), written to need few registers and no private memory (it is
// inlined into a kernel that launches a wave per 64 source points).
__device__ void gn_step_tail(const PairDesc& pd, const ApdConsts& cst) {
  PairState* __restrict__ st = pd.state;
  __shared__ double tH[36], tb[8], tws[56], td[6], tdelta[16], txi[16];
  if (threadIdx.x < 28) {
    double s = 0.0;
    for (int bk0 = 0; bk0 < pd.nblk; bk0 += 16) {
      double v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u)
        v[u] = (bk0 + u < pd.nblk) ? __hip_atomic_load(pd.partials + (size_t)(bk0 + u) * 28 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
#pragma unroll
      for (int u = 0; u < 16; ++u) s += v[u];
    }
    const int t = threadIdx.x;
    if (t < 21) {
      int r = 0, c = t, rowlen = 6;
      while (c >= rowlen) {
        c -= rowlen;
        ++r;
        --rowlen;
      }
      c += r;
      tH[r * 6 + c] = s;
      tH[c * 6 + r] = s;
    } else {
      tb[t - 21] = s;  // b[0..5], then the error
    }
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  for (int a = 0; a < 36; ++a) {
    st->H[a] = tH[a];
    st->Hfin[a] = tH[a];  // LSQ:120
  }
  for (int a = 0; a < 6; ++a) {
    st->b[a] = tb[a];
    tb[a] = -tb[a];
  }
  st->y0 = tb[6];
  st->n_linearize += 1;
  ldlt6_solve(tH, tb, td, tws);      // LSQ:112
  delta_from_d(td, tdelta);          // LSQ:117-118
  isom_mul(tdelta, st->x0, txi);     // LSQ:119
  for (int a = 0; a < 16; ++a) st->x0[a] = txi[a];
  const int it = st->iter;
  st->nr_iterations = it;            // LSQ:68
  const int conv = is_converged(tdelta, 1.0 / cst.rot_eps, 1.0 / cst.trans_eps) ? 1 : 0;  // LSQ:75
  st->iter = it + 1;
  st->converged = conv;
  for (int a = 0; a < 12; ++a) st->Tf[a] = (float)txi[a];  // APD:164 for the next search; LSQ:78 at the end
  st->done = (conv || it + 1 >= cst.max_iterations) ? 1 : 0;  // read by the NEXT launch only: the kernel boundary orders it
}

// ----------------------------------------------------------------------------------------------- sharded-source optimiser
//
// "One large co-registration" (SURVEY 8e): the source points are split over the ranks of an RCCL communicator, the target (map) is
// replicated.  Every rank runs the same launch sequence on its own stream; the ONLY exchange is an in-place ncclAllReduce of 28
// doubles (upper triangle of H, b, error) per linearisation and of 1 double per Levenberg-Marquardt trial.  lm_solve_kernel's body is
// cut at those two points: its locals (d, delta, lambda, nu) live in PairState, every rank takes the same decisions from the same
// reduced numbers, so the poses stay bit-identical on all ranks without a broadcast.

// sum of this rank's block partials in block order -> red[28] (then all-reduced in place).  grid 1, block 64
__global__ __launch_bounds__(64) void shard_reduce_partials_kernel(const PairDesc* __restrict__ descs, double* __restrict__ red) {
  const PairDesc& pd = descs[0];
  if (pd.state->done) return;
  if (threadIdx.x < 28) {
    double s = 0.0;
    for (int bk = 0; bk < pd.nblk; ++bk) s += pd.partials[(size_t)bk * 28 + threadIdx.x];
    red[threadIdx.x] = s;
  }
}

__device__ void shard_prepare_trial(PairState* __restrict__ st, double lambda) {
  __shared__ double ws[56];
  double Hl[36], nb[6];
  for (int a = 0; a < 36; ++a) Hl[a] = st->H[a];
  for (int a = 0; a < 6; ++a) {
    Hl[a * 6 + a] += lambda;
    nb[a] = -st->b[a];
  }
  double d[6], delta[16], xi[16];
  ldlt6_solve(Hl, nb, d, ws);           // LSQ:137-138
  delta_from_d(d, delta);               // LSQ:140-142
  isom_mul(delta, st->x0, xi);          // LSQ:144
  for (int a = 0; a < 6; ++a) st->sd[a] = d[a];
  for (int a = 0; a < 16; ++a) {
    st->sdelta[a] = delta[a];
    st->xi[a] = xi[a];
  }
  st->trial_active = 1;
}

__device__ void shard_end_iteration(PairState* __restrict__ st, const ApdConsts& cst) {  // the tail of lm_solve_kernel mode 0
  const double inv_re = 1.0 / cst.rot_eps, inv_te = 1.0 / cst.trans_eps;
  const int it = st->iter;
  st->nr_iterations = it;  // LSQ:68
  int done = 0, conv = 0;
  if (!st->ok) {
    done = 1;  // "lm not converged!!" LSQ:71-74
    st->lm_failed = 1;
  } else {
    double delta[16];
    for (int a = 0; a < 16; ++a) delta[a] = st->sdelta[a];
    conv = is_converged(delta, inv_re, inv_te) ? 1 : 0;  // LSQ:75
    if (conv) done = 1;
  }
  st->iter = it + 1;
  if (it + 1 >= cst.max_iterations) done = 1;
  st->converged = conv;
  for (int a = 0; a < 12; ++a) st->Tf[a] = (float)st->x0[a];
  st->trial_active = 0;
  __threadfence();
  st->done = done;
}

// after the all-reduce of red[28]: publish H, b, y0; Gauss-Newton: the whole step; LM: lambda initialisation and the first trial
// pose.  mode 1 = linearize API (publish only).  grid 1, block 64 (one lane works: 6 x 6 algebra)
__global__ __launch_bounds__(64) void shard_begin_kernel(const PairDesc* __restrict__ descs, const double* __restrict__ red, ApdConsts cst, int mode) {
  const PairDesc& pd = descs[0];
  PairState* __restrict__ st = pd.state;
  if (st->done || threadIdx.x != 0) return;
  int q = 0;
  for (int r = 0; r < 6; ++r)
    for (int c = r; c < 6; ++c) {
      st->H[r * 6 + c] = red[q];
      st->H[c * 6 + r] = red[q];
      ++q;
    }
  for (int a = 0; a < 6; ++a) st->b[a] = red[21 + a];
  st->y0 = red[27];
  st->n_linearize += 1;
  if (mode == 1) return;
  st->ok = 0;
  st->trial = 0;
  st->trial_active = 0;
  if (cst.optimizer == 0) {  // step_gn, LSQ:107-123
    __shared__ double ws[56];
    double Hl[36], nb[6], d[6], delta[16], xn[16], x0[16];
    for (int a = 0; a < 36; ++a) Hl[a] = st->H[a];
    for (int a = 0; a < 6; ++a) nb[a] = -st->b[a];
    for (int a = 0; a < 16; ++a) x0[a] = st->x0[a];
    ldlt6_solve(Hl, nb, d, ws);
    delta_from_d(d, delta);
    isom_mul(delta, x0, xn);
    for (int a = 0; a < 16; ++a) {
      st->x0[a] = xn[a];
      st->sdelta[a] = delta[a];
    }
    for (int a = 0; a < 36; ++a) st->Hfin[a] = st->H[a];
    st->ok = 1;
    shard_end_iteration(st, cst);
    return;
  }
  if (st->lambda < 0.0) {  // LSQ:131-133
    double mx = 0.0;
    for (int a = 0; a < 6; ++a) mx = fmax(mx, fabs(st->H[a * 6 + a]));
    st->lambda = cst.lm_init_lambda_factor * mx;
  }
  st->nu = 2.0;
  shard_prepare_trial(st, st->lambda);
}

// this rank's part of compute_error at the trial pose (APD:310-346) -> ered[0] (then all-reduced in place).  mode 2: the
// compute_error API (always evaluates at st->xi).  grid 1, block 1024
__global__ __launch_bounds__(1024) void shard_trial_error_kernel(const PairDesc* __restrict__ descs, double* __restrict__ ered, ApdConsts cst, int mode) {
  const PairDesc& pd = descs[0];
  PairState* __restrict__ st = pd.state;
  __shared__ double sred[16];
  __shared__ double sxi[16];
  if (mode != 2 && (st->done || !st->trial_active)) {  // wave-uniform: state is only written between launches
    if (threadIdx.x == 0) ered[0] = 0.0;
    return;
  }
  if (threadIdx.x < 16) sxi[threadIdx.x] = st->xi[threadIdx.x];
  __syncthreads();
  const double yi = block_error(pd, sxi, cst, sred);
  if (threadIdx.x == 0) ered[0] = yi;
}

// after the all-reduce of the trial error: accept / reject (LSQ:146-170); on a rejection the next trial pose; with `last` (or once the
// iteration is decided and no trial is pending) nothing is left but the end-of-iteration bookkeeping.  grid 1, block 64
__global__ __launch_bounds__(64) void shard_trial_decide_kernel(const PairDesc* __restrict__ descs, const double* __restrict__ ered, ApdConsts cst, int last) {
  const PairDesc& pd = descs[0];
  PairState* __restrict__ st = pd.state;
  if (st->done || threadIdx.x != 0) return;
  if (st->trial_active) {
    const double inv_re = 1.0 / cst.rot_eps, inv_te = 1.0 / cst.trans_eps;
    const double yi = ered[0], lambda = st->lambda;
    st->n_error += 1;
    st->trial += 1;
    double den = 0.0;
    for (int a = 0; a < 6; ++a) den += st->sd[a] * (lambda * st->sd[a] - st->b[a]);
    const double rho = (st->y0 - yi) / den;  // LSQ:146
    if (rho < 0) {                             // LSQ:156-164
      double delta[16];
      for (int a = 0; a < 16; ++a) delta[a] = st->sdelta[a];
      if (is_converged(delta, inv_re, inv_te)) {
        st->ok = 1;
        st->trial_active = 0;
      } else {
        st->lambda = st->nu * lambda;
        st->nu = 2 * st->nu;
        if (st->trial < cst.lm_max_iterations) shard_prepare_trial(st, st->lambda);
        else st->trial_active = 0;
      }
    } else {
      for (int a = 0; a < 16; ++a) st->x0[a] = st->xi[a];                   // LSQ:166
      const double f = 1 - pow(2 * rho - 1, 3);
      st->lambda = lambda * fmax(1.0 / 3.0, f);                            // LSQ:167
      for (int a = 0; a < 36; ++a) st->Hfin[a] = st->H[a];                  // LSQ:168
      st->ok = 1;
      st->trial_active = 0;
    }
  }
  if (last) shard_end_iteration(st, cst);
}

// final_transformation applied to the source (LSQ:79, pcl::transformPointCloud with a Matrix4f): float, Eigen product order
__global__ __launch_bounds__(256) void transform_cloud_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z, int n, TfArg tf, float* __restrict__ out_xyz) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float qx, qy, qz;
  transform_f(tf.m, x[i], y[i], z[i], qx, qy, qz);
  out_xyz[3 * (size_t)i + 0] = qx;
  out_xyz[3 * (size_t)i + 1] = qy;
  out_xyz[3 * (size_t)i + 2] = qz;
}

// getFitnessScore / inlier fraction from the packed NN keys of a search at the final transformation
__global__ __launch_bounds__(256) void fitness_kernel(unsigned long long* __restrict__ best_key, int n, double max_range_sq, double inlier_sq, double* __restrict__ out /* [blocks][3]: sum, count, inliers */) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  double s = 0.0, c = 0.0, inl = 0.0;
  if (i < n) {
    const unsigned long long key = best_key[i];
    best_key[i] = ~0ull;
    if (key != ~0ull) {
      const double d = (double)__uint_as_float((unsigned int)(key >> 32));
      if (d <= max_range_sq) {
        s = d;
        c = 1.0;
      }
      if (d < inlier_sq) inl = 1.0;
    }
  }
  s = wave_sum(s);
  c = wave_sum(c);
  inl = wave_sum(inl);
  __shared__ double sw[4][3];
  if ((threadIdx.x & 63) == 0) {
    sw[threadIdx.x >> 6][0] = s;
    sw[threadIdx.x >> 6][1] = c;
    sw[threadIdx.x >> 6][2] = inl;
  }
  __syncthreads();
  if (threadIdx.x < 3)  // one partial per block, added in block order on the host: the score is reproducible run to run
    out[(size_t)blockIdx.x * 3 + threadIdx.x] = (sw[0][threadIdx.x] + sw[1][threadIdx.x]) + (sw[2][threadIdx.x] + sw[3][threadIdx.x]);
}

}  // namespace gorio
