// ugpm_kernels.hip -- gfx950 kernels of the UGPM GP pre-integration (gyro + radar ego-velocity), fp64 throughout.
//
// Reference lines replaced (paths relative to /root/reference/4DRadarSLAM/include/VelInt):
//   lpm_init_kernel      5 LPM rotation integrations + 1 position integration, angle unwrapping, GP state / hyper-parameter
//                        initialisation                                            preint.h:170-742, 1198-1399, 1444-1476
//   gram_kernel          SE Gram matrix, (K + sz2 I)^-1 by Cholesky, K K^-1, K_int K^-1, posterior variances    preint.h:832-866
//   cross_kernel         K_s K^-1 / K_s_int K^-1 tables of the two cost functions           cost_functions.h:183-190, 293-308
//   rot_eval / vel_eval  residuals and analytic Jacobians of GpNorm + Rot / Vel cost functions         cost_functions.h:14-385
//   corr_jac_kernel      stacked Jacobian of the state-correlation step                                   preint.h:887-937
//   ata_kernel           J^T J (+ J^T r) on the fp64 matrix cores (v_mfma_f64_16x16x4_f64), symmetric tiles only
//   lm_step / lm_decide  Ceres-style trust-region Levenberg-Marquardt (normal equations, Cholesky in HBM/L2)  preint.h:943-967
//   corr_factor_kernel   (J^T J + 1e-5 I) = L L^T; corr_diag_kernel: diag(A^-1), correlation scaling   preint.h:1478-1492
//   finish_kernel        alpha = K^-1 s, post-integration Jacobian tables                        preint.h:978-1060, 1401-1441
//   infer_kernel         Se3Integrator::get(t) + the bias-prior inflation of VelPreintegration::get    preint.h:1069-1153, 1744-1757
//
// One workgroup (set) per window; windows are independent so a batch fills the chip without any inter-workgroup traffic.
// The work is small dense fp64 algebra (S = 66 states, ~260 samples per sensor): matrices live in HBM but are L2 resident
// (11 MB per window), tiles are staged through LDS in ata_kernel, and every reduction is a fixed-order tree, so results are
// run-to-run reproducible.
#include <hip/hip_runtime.h>
#include <math.h>

#include "ugpm_device.h"

namespace gorio {
namespace ug {

// The window descriptors are written by the host before the launch and never change during it: reading one through the
// CONSTANT address space puts its members in SGPRs (s_load) and, more importantly, lets the compiler treat every pointer member
// as a GLOBAL pointer (a generic pointer loaded from constant memory cannot point to LDS / scratch), so all accesses below are
// global_load / global_store instead of flat_* -- flat operations count against both the LDS and the vector-memory wait
// counters and serialise the two.
__device__ __forceinline__ UgpmWin load_win(const UgpmWin* __restrict__ wins, int i) {
  UgpmWin w;
  __builtin_memcpy(&w, (const __attribute__((address_space(4))) void*)(wins + i), sizeof(UgpmWin));
  return w;
}


// XCD-aware placement of a (parts, windows, z) launch: consecutive workgroups go round-robin to the 8 XCDs, so with the plain numbering
// the workgroups of one window -- which share its matrices -- land on different L2s.  Renumbered, window w runs entirely on XCD w mod 8
// (workgroup L in launch order -> xcd = L mod 8, slot = L / 8, window = (slot / inner) * 8 + xcd, inner index = slot mod inner), the
// XCD the (windows, parts) launches of the LM kernels put it on anyway when the window count is a multiple of 8.  Only with at least 16
// windows in multiples of 8; otherwise the plain numbering.
#ifndef GORIO_UGPM_XCD
#define GORIO_UGPM_XCD 1
#endif
struct WinPart {
  int win, part, z;
};
__device__ __forceinline__ WinPart xcd_win_part() {
  WinPart r{(int)blockIdx.y, (int)blockIdx.x, (int)blockIdx.z};
#if GORIO_UGPM_XCD
  if (gridDim.y >= 16u && (gridDim.y & 7u) == 0u) {
    const unsigned int inner = gridDim.x * gridDim.z;
    const unsigned int L = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned int slot = L >> 3, in = slot % inner;
    r.win = (int)((slot / inner) * 8u + (L & 7u));
    r.part = (int)(in % gridDim.x);
    r.z = (int)(in / gridDim.x);
  }
#endif
  return r;
}

constexpr double kDt = 0.01;        // kNumDtJacobianDelta, math_utils.h:15
constexpr double kBw = 0.0001;      // kNumGyrBiasJacobianDelta, math_utils.h:17
constexpr double kExpTol = 1e-14;   // kExpNormTolerance, math_utils.h:11
constexpr double kPi = 3.14159265358979323846;

struct V3 {
  double x, y, z;
};
struct M3 {
  double m[9];
};
__device__ __forceinline__ V3 v3(double x, double y, double z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(double s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ double vnorm(V3 a) { return sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ M3 eye3() { return M3{{1, 0, 0, 0, 1, 0, 0, 0, 1}}; }
__device__ __forceinline__ M3 mmul(const M3& a, const M3& b) {
  M3 r;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) r.m[i * 3 + j] = a.m[i * 3] * b.m[j] + a.m[i * 3 + 1] * b.m[3 + j] + a.m[i * 3 + 2] * b.m[6 + j];
  return r;
}
__device__ __forceinline__ M3 mtr(const M3& a) { return M3{{a.m[0], a.m[3], a.m[6], a.m[1], a.m[4], a.m[7], a.m[2], a.m[5], a.m[8]}}; }
__device__ __forceinline__ V3 mvec(const M3& a, V3 v) { return V3{a.m[0] * v.x + a.m[1] * v.y + a.m[2] * v.z, a.m[3] * v.x + a.m[4] * v.y + a.m[5] * v.z, a.m[6] * v.x + a.m[7] * v.y + a.m[8] * v.z}; }
__device__ __forceinline__ M3 skew(V3 v) { return M3{{0.0, -v.z, v.y, v.z, 0.0, -v.x, -v.y, v.x, 0.0}}; }
__device__ __forceinline__ void store3(double* p, V3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
__device__ __forceinline__ V3 load3(const double* p) { return V3{p[0], p[1], p[2]}; }
__device__ __forceinline__ void storeM(double* p, const M3& a) {
#pragma unroll
  for (int i = 0; i < 9; ++i) p[i] = a.m[i];
}
__device__ __forceinline__ M3 loadM(const double* p) {
  M3 a;
#pragma unroll
  for (int i = 0; i < 9; ++i) a.m[i] = p[i];
  return a;
}

// expMap (math_utils.h:55-58): AngleAxis(|v|, v / |v|) -> c I + (1 - c) a a^T + s [a]x ; a zero vector gives I
__device__ M3 expMap(V3 v) {
  const double n2 = v.x * v.x + v.y * v.y + v.z * v.z;
  const double ang = sqrt(n2);
  V3 a = v;
  if (n2 > 0.0) a = (1.0 / ang) * v;
  const double s = sin(ang), c = cos(ang);
  const V3 sa = s * a, ca = (1.0 - c) * a;
  M3 R;
  double t;
  t = ca.x * a.y; R.m[1] = t - sa.z; R.m[3] = t + sa.z;
  t = ca.x * a.z; R.m[2] = t + sa.y; R.m[6] = t - sa.y;
  t = ca.y * a.z; R.m[5] = t - sa.x; R.m[7] = t + sa.x;
  R.m[0] = ca.x * a.x + c; R.m[4] = ca.y * a.y + c; R.m[8] = ca.z * a.z + c;
  return R;
}

// logMap (math_utils.h:48-51): rotation matrix -> quaternion -> angle * axis, angle in [0, pi]
__device__ V3 logMap(const M3& R) {
  double q[4];
  double t = R.m[0] + R.m[4] + R.m[8];
  if (t > 0.0) {
    t = sqrt(t + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (R.m[7] - R.m[5]) * t;
    q[1] = (R.m[2] - R.m[6]) * t;
    q[2] = (R.m[3] - R.m[1]) * t;
  } else {
    int i = 0;
    if (R.m[4] > R.m[0]) i = 1;
    if (R.m[8] > R.m[i * 4]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = sqrt(R.m[i * 4] - R.m[j * 4] - R.m[k * 4] + 1.0);
    q[i] = 0.5 * t;
    t = 0.5 / t;
    q[3] = (R.m[k * 3 + j] - R.m[j * 3 + k]) * t;
    q[j] = (R.m[j * 3 + i] + R.m[i * 3 + j]) * t;
    q[k] = (R.m[k * 3 + i] + R.m[i * 3 + k]) * t;
  }
  double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
  if (n != 0.0) {
    const double ang = 2.0 * atan2(n, fabs(q[3]));
    if (q[3] < 0) n = -n;
    return V3{ang * q[0] / n, ang * q[1] / n, ang * q[2] / n};
  }
  return V3{0, 0, 0};
}

// jacobianRighthandSO3 (math_utils.h:63-80) and its inverse (math_utils.h:83-99)
__device__ M3 Jr(V3 v) {
  M3 out = eye3();
  const double n = vnorm(v);
  if (n > kExpTol) {
    const M3 S = skew(v), S2 = mmul(S, S);
    const double a = (n - sin(n)) / (n * n * n), b = (1.0 - cos(n)) / (n * n);
#pragma unroll
    for (int i = 0; i < 9; ++i) out.m[i] += a * S2.m[i] - b * S.m[i];
  }
  return out;
}
__device__ M3 JrInv(V3 v) {
  M3 out = eye3();
  const double n = vnorm(v);
  if (n > kExpTol) {
    const M3 S = skew(v), S2 = mmul(S, S);
    const double a = (1.0 / (n * n)) - ((1 + cos(n)) / (2.0 * n * sin(n)));
#pragma unroll
    for (int i = 0; i < 9; ++i) out.m[i] += 0.5 * S.m[i] + a * S2.m[i];
  }
  return out;
}
__device__ V3 addN2Pi(V3 r, int n) {  // math_utils.h:385-397
  const double nr = vnorm(r);
  if (nr != 0) return (2.0 * kPi * n + nr) * ((1.0 / nr) * r);
  return r;
}

// SE kernel and its integral (math_utils.h:102-126); sqrt_inv_l2 = sqrt(1 / l2)
__device__ __forceinline__ double se_k(double x1, double x2, double l2, double sf2) {
  const double d = x1 - x2;
  return exp((d * d) * (-0.5 / l2)) * sf2;
}
__device__ __forceinline__ double se_kint(double a, double b, double x2, double l2, double sf2) {
  const double sq = sqrt(1.0 / l2);
  const double alpha = sqrt(2.0) * sf2 * sqrt(kPi) / (2.0 * sq);
  return alpha * (erf(sqrt(2.0) * (b - x2) * sq / 2.0) - erf(sqrt(2.0) * (-x2 + a) * sq / 2.0));
}
__device__ __forceinline__ double se_kint_dt(double a, double b, double x2, double l2, double sf2) {  // math_utils.h:130-141
  return sf2 * exp(((b - x2) * (b - x2)) / (-2.0 * l2)) - sf2 * exp(((x2 - a) * (x2 - a)) / (-2.0 * l2));
}
__device__ __forceinline__ double kss_int(double a, double b, double l2, double sf2) {  // math_utils.h:378-382
  return 2.0 * l2 * sf2 * exp(-((a - b) * (a - b)) / (2.0 * l2)) - 2.0 * l2 * sf2 + (sqrt(2.0) * sf2 * sqrt(kPi) * erf((sqrt(2.0) * (a - b) * sqrt(1.0 / l2)) / 2.0) * (a - b)) / sqrt(1.0 / l2);
}

// =============================================================================================== LPM initialisation

// One LPM rotation integration (IterativeIntegrator::rotPreint, preint.h:321-391, 407-519) on ONE lane, streaming over the merged
// time line instead of materialising it.  variant: 0 = data minus the bias prior; 1 = all sample times shifted by -0.01, no prior
// (preint.h:1270-1285); 2..4 = gyro axis (variant - 2) + 1e-4, no prior, bare (preint.h:1338-1350).
__device__ void lpm_rotation(const UgpmWin& w, int variant) {
  const int S = w.S, G = w.G, V = w.V;
  const double lpm_start = w.state_t[0];
  const double tshift = variant == 1 ? kDt : 0.0;
  // the time line is the union of 6 ascending lists (preint.h:214-237): 0 t_vect, 1 t_vect + 0.01, 2 {start_t}, 3 {lpm_start,
  // lpm_start + 0.01}, 4 velocity stamps, 5 500 Hz filler stamps
  int nl[6] = {S, S, 1, 2, V, 0};
  double fake_first = 0.0, fake_q = 0.0;
  auto val = [&](int l, int i) -> double {
    switch (l) {
      case 0: return w.state_t[i];
      case 1: return w.state_t[i] + kDt;
      case 2: return w.start_t;
      case 3: return lpm_start + (i ? kDt : 0.0);
      case 4: return w.vel_t[i] - tshift;
      default: return fake_first + i * fake_q;
    }
  };
  {  // filler decision, preint.h:228-237 with getSmallestGap() == LAST gap of the merged line (types.h:442-450)
    double first = val(0, 0), top1 = -1e300, top2 = -1e300;
    for (int l = 0; l < 5; ++l) {
      if (nl[l] == 0) continue;
      const double f = val(l, 0);
      if (f < first) first = f;
      for (int q = nl[l] - 1; q >= 0 && q >= nl[l] - 2; --q) {
        const double c = val(l, q);
        if (c > top1) {
          top2 = top1;
          top1 = c;
        } else if (c > top2) {
          top2 = c;
        }
      }
    }
    if ((top1 - top2) > (1.0 / 500.0)) {
      const int nb = (int)floor((top1 - first) * 500.0);
      if (nb > 0) {
        nl[5] = nb;
        fake_first = first;
        fake_q = (top1 - first) / ((double)nb);
      }
    }
  }
  // gyro accessors for this variant
  auto gt = [&](int i) -> double { return w.gyr_t[i] - tshift; };
  auto gd = [&](int a, int i) -> double {
    double d = w.gyr[a * G + i];
    if (variant == 0) d -= w.gyr_bias[a];
    if (variant - 2 == a) d += kBw;
    return d;
  };
  // linearInterpolation state per axis (math_utils.h:487-532); the bracketing samples are cached in registers and refreshed only
  // when the segment pointer advances
  int ptr[3] = {0, 0, 0};
  double al[3], be[3], tlo[3], thi[3], dhi[3];
  for (int a = 0; a < 3; ++a) {
    tlo[a] = gt(0);
    thi[a] = gt(1);
    const double d0 = gd(a, 0);
    dhi[a] = gd(a, 1);
    al[a] = (dhi[a] - d0) / (thi[a] - tlo[a]);
    be[a] = d0 - (al[a] * tlo[a]);
  }
  const double gt0 = gt(0);
  auto interp = [&](double t, double* out) {
    for (int a = 0; a < 3; ++a) {
      if (t > gt0) {
        while (ptr[a] != (G - 2)) {
          if ((t <= thi[a]) && (t > tlo[a])) break;
          ptr[a]++;
          const double d0 = dhi[a];
          tlo[a] = thi[a];
          thi[a] = gt(ptr[a] + 1);
          dhi[a] = gd(a, ptr[a] + 1);
          al[a] = (dhi[a] - d0) / (thi[a] - tlo[a]);
          be[a] = d0 - (al[a] * tlo[a]);
        }
      }
      out[a] = al[a] * t + be[a];
    }
  };
  // streaming merge; ties are emitted in the list order 3, 0, 1, 2, 4, 5 so that nothing captured precedes the LPM start stamp.
  // The head of every list is cached in a register and refreshed only when that list advances (one load per step).
  int p0 = 0, p1 = 0, p2 = 0, p3 = 0, p4 = 0, p5 = 0;
  const double kInf = 1.7976931348623157e308;
  double h0 = nl[0] > 0 ? val(0, 0) : kInf, h1 = nl[1] > 0 ? val(1, 0) : kInf, h2 = val(2, 0), h3 = val(3, 0);
  double h4 = nl[4] > 0 ? val(4, 0) : kInf, h5 = nl[5] > 0 ? val(5, 0) : kInf;
  int total = 0;
  for (int l = 0; l < 6; ++l) total += nl[l];
  M3 R = eye3();
  double t_prev = 0.0, w_prev[3] = {0, 0, 0};
  double* Rq = w.Rq + (size_t)variant * 2 * S * 9;
  for (int step = 0; step < total; ++step) {
    int bl = 3;
    double bt = h3;
    if (h0 < bt) { bl = 0; bt = h0; }
    if (h1 < bt) { bl = 1; bt = h1; }
    if (h2 < bt) { bl = 2; bt = h2; }
    if (h4 < bt) { bl = 4; bt = h4; }
    if (h5 < bt) { bl = 5; bt = h5; }
    int idx;
    switch (bl) {
      case 0: idx = p0++; h0 = p0 < nl[0] ? val(0, p0) : kInf; break;
      case 1: idx = p1++; h1 = p1 < nl[1] ? val(1, p1) : kInf; break;
      case 2: idx = p2++; h2 = p2 < nl[2] ? val(2, p2) : kInf; break;
      case 3: idx = p3++; h3 = p3 < nl[3] ? val(3, p3) : kInf; break;
      case 4: idx = p4++; h4 = p4 < nl[4] ? val(4, p4) : kInf; break;
      default: idx = p5++; h5 = p5 < nl[5] ? val(5, p5) : kInf; break;
    }
    if (step > 0) {  // rotIterativeIntegration, preint.h:421-453 / 496-506
      const double dt = bt - t_prev;
      R = mmul(R, expMap(v3(w_prev[0] * dt, w_prev[1] * dt, w_prev[2] * dt)));
    }
    if (bl == 3 && idx == 0) R = eye3();  // start_index_: re-reference to the LPM start (preint.h:477-485, 509-517)
    if (bl == 0) storeM(Rq + (size_t)idx * 9, R);
    else if (bl == 1) storeM(Rq + (size_t)(S + idx) * 9, R);
    else if (bl == 2) storeM(w.Rstart + variant * 9, R);
    else if (bl == 4 && variant == 0) {  // reprojectVelData, math_utils.h:415-426
      const V3 vr = mvec(R, v3(w.vel[idx] - w.vel_bias[0], w.vel[V + idx] - w.vel_bias[1], w.vel[2 * V + idx] - w.vel_bias[2]));
      w.velr[idx] = vr.x;
      w.velr[V + idx] = vr.y;
      w.velr[2 * V + idx] = vr.z;
    }
    interp(bt, w_prev);
    t_prev = bt;
  }
}

// ---- workgroup-parallel form of lpm_rotation (same result up to the association order of the rotation products) ----
// The merged time line is materialised in LDS by RANK (every stamp finds its position with binary searches in the other
// lists), the interpolated rates and the per-step rotations are evaluated by all lanes, and the running product -- the only
// sequential part of the reference loop -- becomes a three-phase scan: per-lane chunk products, a Hillis-Steele scan of the
// chunk totals in LDS, and a final sweep.  kLpmMaxT bounds the line length; longer lines use the one-lane routine above.
constexpr int kLpmMaxT = 2048;
struct LpmLds {
  double tl[kLpmMaxT];             // merged stamps
  unsigned short kind[kLpmMaxT];   // list id of the stamp
  unsigned short kidx[kLpmMaxT];   // its index in that list
  double tot[2][256][9];           // chunk totals, ping-pong
  double ps[9];                    // P(start index)^T
  int start_rank;
};

__device__ __forceinline__ int lower_bound_f(int n, double v, const double* __restrict__ a, double shift) {  // first i with a[i] - shift >= v
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (a[mid] - shift < v) lo = mid + 1; else hi = mid;
  }
  return lo;
}
__device__ __forceinline__ int upper_bound_f(int n, double v, const double* __restrict__ a, double shift) {  // first i with a[i] - shift > v
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (a[mid] - shift <= v) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// returns false (nothing written) when the time line does not fit kLpmMaxT.  All threads of the workgroup must call it.
__device__ bool lpm_rotation_parallel(const UgpmWin& w, int variant, LpmLds& L, double* __restrict__ Rseq /* [T][9] global scratch */, size_t rseq_cap) {
  const int S = w.S, G = w.G, V = w.V;
  const int nthr = blockDim.x, tid = threadIdx.x;
  const double lpm_start = w.state_t[0];
  const double tshift = variant == 1 ? kDt : 0.0;
  // list sizes and the 500 Hz filler (same rule as lpm_rotation)
  int nfake = 0;
  double fake_first = 0.0, fake_q = 0.0;
  {
    double first = w.state_t[0], top1 = -1e300, top2 = -1e300;
    auto consider = [&](double c) {
      if (c > top1) { top2 = top1; top1 = c; } else if (c > top2) { top2 = c; }
    };
    consider(w.state_t[S - 1]); if (S > 1) consider(w.state_t[S - 2]);
    consider(w.state_t[S - 1] + kDt); if (S > 1) consider(w.state_t[S - 2] + kDt);
    consider(w.start_t);
    consider(lpm_start + kDt); consider(lpm_start);
    consider(w.vel_t[V - 1] - tshift); if (V > 1) consider(w.vel_t[V - 2] - tshift);
    first = fmin(first, fmin(w.start_t, w.vel_t[0] - tshift));
    if ((top1 - top2) > (1.0 / 500.0)) {
      const int nb = (int)floor((top1 - first) * 500.0);
      if (nb > 0) { nfake = nb; fake_first = first; fake_q = (top1 - first) / ((double)nb); }
    }
  }
  const int T = 2 * S + 3 + V + nfake;
  if (T > kLpmMaxT || (size_t)T * 9 > rseq_cap) return false;
  // ---- 1. rank of every stamp.  tie order of the lists: 3, 0, 1, 2, 4, 5  (prio[l] below)
  auto count_in = [&](int l, double v, bool upper) -> int {  // number of stamps of list l that precede a stamp of value v
    switch (l) {
      case 0: return upper ? upper_bound_f(S, v, w.state_t, 0.0) : lower_bound_f(S, v, w.state_t, 0.0);
      case 1: return upper ? upper_bound_f(S, v, w.state_t, -kDt) : lower_bound_f(S, v, w.state_t, -kDt);
      case 2: return upper ? (w.start_t <= v ? 1 : 0) : (w.start_t < v ? 1 : 0);
      case 3: { const double a0 = lpm_start, a1 = lpm_start + kDt; return upper ? ((a0 <= v) + (a1 <= v)) : ((a0 < v) + (a1 < v)); }
      case 4: return upper ? upper_bound_f(V, v, w.vel_t, tshift) : lower_bound_f(V, v, w.vel_t, tshift);
      default: {
        int lo = 0, hi = nfake;
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          const double c = fake_first + mid * fake_q;
          if (upper ? (c <= v) : (c < v)) lo = mid + 1; else hi = mid;
        }
        return lo;
      }
    }
  };
  const int prio[6] = {1, 2, 3, 0, 4, 5};
  const int nl[6] = {S, S, 1, 2, V, nfake};
  for (int e = tid; e < T; e += nthr) {
    int l = 0, i = e;
    while (i >= nl[l]) { i -= nl[l]; ++l; }
    double v;
    switch (l) {
      case 0: v = w.state_t[i]; break;
      case 1: v = w.state_t[i] + kDt; break;
      case 2: v = w.start_t; break;
      case 3: v = lpm_start + (i ? kDt : 0.0); break;
      case 4: v = w.vel_t[i] - tshift; break;
      default: v = fake_first + i * fake_q; break;
    }
    int rank = i;
    for (int b = 0; b < 6; ++b)
      if (b != l && nl[b] > 0) rank += count_in(b, v, prio[b] < prio[l]);
    L.tl[rank] = v;
    L.kind[rank] = (unsigned short)l;
    L.kidx[rank] = (unsigned short)i;
    if (l == 3 && i == 0) L.start_rank = rank;
  }
  __syncthreads();
  // ---- 2./3. interpolated rate at every stamp and the step rotation E_i = Exp(w_i (t_{i+1} - t_i)), i < T - 1, chunked per lane
  auto gt = [&](int i) -> double { return w.gyr_t[i] - tshift; };
  auto gd = [&](int a, int i) -> double {
    double d = w.gyr[a * G + i];
    if (variant == 0) d -= w.gyr_bias[a];
    if (variant - 2 == a) d += kBw;
    return d;
  };
  const double gt0 = gt(0);
  auto step_rot = [&](int i) -> M3 {
    const double t = L.tl[i];
    int p = 0;
    if (t > gt0) {
      p = lower_bound_f(G, t, w.gyr_t, tshift) - 1;  // gt[p] < t <= gt[p + 1]
      if (p > G - 2) p = G - 2;
      if (p < 0) p = 0;
    }
    const double t0 = gt(p), t1 = gt(p + 1);
    double wv[3];
    for (int a = 0; a < 3; ++a) {
      const double d0 = gd(a, p), d1 = gd(a, p + 1);
      const double al = (d1 - d0) / (t1 - t0);
      const double be = d0 - (al * t0);
      wv[a] = al * t + be;
    }
    const double dt = L.tl[i + 1] - t;
    return expMap(v3(wv[0] * dt, wv[1] * dt, wv[2] * dt));
  };
  // P(i) = E_0 E_1 ... E_{i-1} (P(0) = I).  Lane c owns stamps [c*ch, (c+1)*ch): phase A = local products
  const int nch = nthr < 256 ? nthr : 256;
  const int ch = (T + nch - 1) / nch;
  if (tid < nch) {
    M3 acc = eye3();
    const int i0 = tid * ch, i1 = min(T, i0 + ch);
    for (int i = i0; i < i1; ++i) {
      storeM(Rseq + (size_t)i * 9, acc);            // local prefix (exclusive)
      if (i < T - 1) acc = mmul(acc, step_rot(i));
    }
    storeM(&L.tot[0][tid][0], acc);  // product of the chunk's steps
  }
  __syncthreads();
  // phase B: inclusive Hillis-Steele scan of the chunk totals
  int cur = 0;
  for (int off = 1; off < nch; off <<= 1) {
    if (tid < nch) {
      M3 m = loadM(&L.tot[cur][tid][0]);
      if (tid >= off) m = mmul(loadM(&L.tot[cur][tid - off][0]), m);
      storeM(&L.tot[cur ^ 1][tid][0], m);
    }
    __syncthreads();
    cur ^= 1;
  }
  // phase C: P(i) = (product of all earlier chunks) * local prefix; P(start) for the re-referencing of preint.h:477-485
  if (tid < nch) {
    const M3 pre = tid > 0 ? loadM(&L.tot[cur][tid - 1][0]) : eye3();
    const int i0 = tid * ch, i1 = min(T, i0 + ch);
    for (int i = i0; i < i1; ++i) {
      const M3 Pi = mmul(pre, loadM(Rseq + (size_t)i * 9));
      storeM(Rseq + (size_t)i * 9, Pi);
      if (i == L.start_rank) storeM(L.ps, mtr(Pi));
    }
  }
  __syncthreads();
  // ---- 4./5. re-reference to the LPM start and capture the stamps UGPM reads
  const M3 PsT = loadM(L.ps);
  double* Rq = w.Rq + (size_t)variant * 2 * S * 9;
  for (int i = tid; i < T; i += nthr) {
    const int l = L.kind[i], idx = L.kidx[i];
    if (l == 5 || l == 3) continue;
    if (l == 4 && variant != 0) continue;
    // stamps ahead of the LPM start keep the un-referenced product, as the sequential loop leaves them
    const M3 Pi = loadM(Rseq + (size_t)i * 9);
    const M3 R = i >= L.start_rank ? mmul(PsT, Pi) : Pi;
    if (l == 0) storeM(Rq + (size_t)idx * 9, R);
    else if (l == 1) storeM(Rq + (size_t)(S + idx) * 9, R);
    else if (l == 2) storeM(w.Rstart + variant * 9, R);
    else {  // reprojectVelData, math_utils.h:415-426
      const V3 vr = mvec(R, v3(w.vel[idx] - w.vel_bias[0], w.vel[V + idx] - w.vel_bias[1], w.vel[2 * V + idx] - w.vel_bias[2]));
      w.velr[idx] = vr.x;
      w.velr[V + idx] = vr.y;
      w.velr[2 * V + idx] = vr.z;
    }
  }
  __syncthreads();
  return true;
}

// Trapezoid position integration of the rotated velocity (posePreintLPM main loop, preint.h:552-665, values only) for one axis.
__device__ void lpm_position(const UgpmWin& w, int axis) {
  const int S = w.S, V = w.V;
  const double start = w.state_t[0];
  const double* vt = w.vel_t;
  const double* vd = w.velr + (size_t)axis * V;
  int data_ptr = 0;
  while (vt[data_ptr + 1] < start) {
    data_ptr++;
    if (data_ptr == V - 1) {
      *w.status = -3;
      return;
    }
  }
  int ptr = data_ptr;
  double alpha = (vd[ptr + 1] - vd[ptr]) / (vt[ptr + 1] - vt[ptr]);
  double beta = vd[ptr] - alpha * vt[ptr];
  double t_0 = start, t_1 = vt[ptr + 1];
  double d_0 = alpha * vt[ptr] + beta, d_1 = vd[ptr + 1];
  double backup = 0.0;
  int p0 = 0, p1 = 0, p2 = 0;  // query lists: t_vect, t_vect + 0.01, {start_t}
  for (int step = 0; step < 2 * S + 1; ++step) {
    int bl = -1;
    double t = 0.0;
    if (p0 < S) { bl = 0; t = w.state_t[p0]; }
    if (p1 < S && (bl < 0 || w.state_t[p1] + kDt < t)) { bl = 1; t = w.state_t[p1] + kDt; }
    if (p2 < 1 && (bl < 0 || w.start_t < t)) { bl = 2; t = w.start_t; }
    int idx = 0;
    if (bl == 0) idx = p0++;
    else if (bl == 1) idx = p1++;
    else p2++;
    if (t > vt[0]) {
      while (true) {
        if ((t >= vt[ptr]) && (t <= vt[ptr + 1])) break;
        if (ptr < (V - 2)) {
          backup = backup + ((t_1 - t_0) * (d_0 + d_1) / 2.0);
          ptr++;
          t_0 = vt[ptr];
          t_1 = vt[ptr + 1];
          d_0 = vd[ptr];
          d_1 = vd[ptr + 1];
          alpha = (d_1 - d_0) / (t_1 - t_0);
          beta = d_0 - alpha * t_0;
        } else {
          break;
        }
      }
    }
    const double temp_d_1 = alpha * t + beta;
    const double temp_d_p = backup + ((t - t_0) * (d_0 + temp_d_1) / 2.0);
    if (bl < 2) w.dp[((size_t)bl * S + idx) * 3 + axis] = temp_d_p;
  }
}

// 2 pi unwrapping of the rotation vectors (preint.h:1214-1263 and the three copies in 1288-1395)
__device__ void unwrap_variant(const UgpmWin& w, int variant) {
  const int S = w.S;
  const M3 sRt = mtr(loadM(w.Rstart + variant * 9));
  const double* Rq = w.Rq + (size_t)variant * 2 * S * 9;
  double* r0 = w.r0 + (size_t)variant * S * 3;
  double* r1 = w.r1 + (size_t)variant * S * 3;
  for (int pass = 0; pass < 2; ++pass) {
    int rev[2] = {0, 0};
    V3 prev[2] = {v3(0, 0, 0), v3(0, 0, 0)};
    const int from = pass == 0 ? w.overlap : w.overlap - 1, to = pass == 0 ? S : -1, stp = pass == 0 ? 1 : -1;
    for (int i = from; i != to; i += stp) {
      for (int j = 0; j < 2; ++j) {
        const V3 tr = logMap(mmul(sRt, loadM(Rq + ((size_t)j * S + i) * 9)));
        int id = 0;
        double best = 1.7976931348623157e308;
        V3 bc = tr;
        for (int q = 0; q < 3; ++q) {
          const V3 c = addN2Pi(tr, rev[j] + q - 1);
          const double d = vnorm(prev[j] - c);
          if (d < best) {
            best = d;
            id = q;
            bc = c;
          }
        }
        prev[j] = bc;
        rev[j] += (id - 1);
      }
      store3(r0 + (size_t)i * 3, prev[0]);
      store3(r1 + (size_t)i * 3, prev[1]);
    }
  }
}

// grid: (windows), block 320 = 5 waves; wave v's first lane integrates LPM variant v.
// LPM initialisation, part 1: one workgroup per (window, integration).  The five rotation integrations (nominal, time-shifted, three
// gyro-bias perturbations) are independent, so they run side by side; each is followed by its own 2 pi unwrapping, and the nominal
// one by the trapezoid position integration that needs its re-projected velocities.  grid: (windows, 5), block 320.
__global__ __launch_bounds__(320) void lpm_rot_kernel(const UgpmWin* __restrict__ wins) {
  const UgpmWin w = load_win(wins, blockIdx.x);
  if (*w.status != 0) return;
  const int var = blockIdx.y, lane = threadIdx.x & 63;
  __shared__ LpmLds lds;
  {
    // scratch for the P(i) sequence: a fifth of the (still unused) rotation-problem Jacobian buffer per integration
    const size_t cap = ((size_t)(3 * w.S + 3 * w.G) * 3 * w.S) / 5;
    double* Rseq = w.Jrot + (size_t)var * cap;
    if (!lpm_rotation_parallel(w, var, lds, Rseq, cap)) {  // time line too long for the LDS form: one lane integrates
      if (threadIdx.x == 0) lpm_rotation(w, var);
    }
  }
  __syncthreads();
  if (var == 0 && threadIdx.x < 3) lpm_position(w, threadIdx.x);
  if (threadIdx.x == 64) unwrap_variant(w, var);
  (void)lane;
}

// LPM initialisation, part 2 (after all five integrations): GP state seeds, their finite-difference Jacobians, hyper-parameters.
// grid: (windows), block 320.
__global__ __launch_bounds__(320) void lpm_init_kernel(const UgpmWin* __restrict__ wins) {
  const UgpmWin w = load_win(wins, blockIdx.x);
  if (*w.status != 0) return;
  const int S = w.S;
  const M3 sRt0 = mtr(loadM(w.Rstart));
  for (int i = threadIdx.x; i < S; i += blockDim.x) {  // preint.h:1231-1236, 1304-1307, 1369-1372
    const V3 a0 = load3(w.r0 + (size_t)i * 3), a1 = load3(w.r1 + (size_t)i * 3);
    const V3 d = (1.0 / kDt) * (a1 - a0);
    w.s_dr[i] = d.x; w.s_dr[S + i] = d.y; w.s_dr[2 * S + i] = d.z;
    const V3 vel = mvec(sRt0, (1.0 / kDt) * (load3(w.dp + ((size_t)S + i) * 3) - load3(w.dp + (size_t)i * 3)));
    w.s_vel[i] = vel.x; w.s_vel[S + i] = vel.y; w.s_vel[2 * S + i] = vel.z;
    store3(w.d_r_dt_local + (size_t)i * 3, mvec(Jr(a0), d));
    {
      const V3 b0 = load3(w.r0 + ((size_t)S + i) * 3), b1 = load3(w.r1 + ((size_t)S + i) * 3);
      const M3 J = Jr(b0);
      store3(w.d_r_dt_local_shift + (size_t)i * 3, mvec(J, (1.0 / kDt) * (b1 - b0)));
      store3(w.delta_r_time + (size_t)i * 3, mvec(J, b0 - a0));
    }
    for (int ax = 0; ax < 3; ++ax) {
      const V3 b0 = load3(w.r0 + ((size_t)(2 + ax) * S + i) * 3), b1 = load3(w.r1 + ((size_t)(2 + ax) * S + i) * 3);
      const M3 J = Jr(b0);
      store3(w.d_r_bw_local_shift + ((size_t)ax * S + i) * 3, mvec(J, (1.0 / kDt) * (b1 - b0)));
      store3(w.delta_r_bw + ((size_t)ax * S + i) * 3, mvec(J, b0 - a0));
    }
  }
  __syncthreads();
  if (threadIdx.x < 6) {  // initialiseHyperParam, preint.h:1444-1476
    const int c = threadIdx.x;
    double* s = (c < 3 ? w.s_dr + (size_t)c * S : w.s_vel + (size_t)(c - 3) * S);
    double m = 0.0;
    for (int i = 0; i < S; ++i) m += s[i];
    m /= (double)S;
    double var = 0.0;
    for (int i = 0; i < S; ++i) var += (s[i] - m) * (s[i] - m);
    var /= (double)S;
    const double noise = c < 3 ? w.gyr_var : w.vel_var;
    w.hyper[c * 4 + 0] = (3.0 / w.state_freq) * (3.0 / w.state_freq);
    w.hyper[c * 4 + 1] = fmax(var, noise);
    w.hyper[c * 4 + 2] = noise;
    w.hyper[c * 4 + 3] = m;
    for (int i = 0; i < S; ++i) s[i] -= m;
  }
}

// =============================================================================================== block-wide dense helpers

// In-place lower Cholesky of the n x n matrix A (row-major, leading dimension lda, resident in HBM / L2) by one workgroup.
// Right-looking, blocked by 16 columns:
//  - the 16 x 16 diagonal block is factored by one wave entirely in REGISTERS (lane r owns row r; the pivot column is exchanged
//    with v_readlane, whose lane operand is a compile-time constant after unrolling -- no LDS round trip inside the 16 steps);
//  - the panel below is solved against it one row per lane (its 16 entries are requested from L2 before the factor finishes) and
//    kept in LDS;
//  - the trailing update A22 -= P P^T runs on the fp64 matrix cores, one 16 x 16 tile per MFMA chain, operands read from the LDS
//    panel; the only global traffic is one read-modify-write sweep of the trailing triangle per block column.
// Returns false on a non-positive pivot.  The upper triangle is never referenced.  n <= kCholMaxN.
// With `rhs` (n doubles in GLOBAL memory like A; needs n < 384) the right-hand side rides along as one more row of the matrix being factored, so on
// return it holds L^-1 rhs: the forward substitution costs nothing beyond one extra row in every panel.
constexpr int kCholNB = 16;
constexpr int kCholMaxN = 6 * 160;
#ifdef GORIO_CHOL_TIMING  // tools/chol_bench.hip: cycles per phase, accumulated by thread 0 of workgroup 0
__device__ long long g_chol_t[12];
#define CHOL_T(k)                                                         \
  do {                                                                    \
    if (blockIdx.x == 0 && threadIdx.x == 0) {                            \
      const long long now_ = (long long)__builtin_readcyclecounter();     \
      g_chol_t[k] += now_ - chol_t_last_;                                 \
      chol_t_last_ = now_;                                                \
    }                                                                     \
  } while (0)
#define CHOL_T_INIT long long chol_t_last_ = (long long)__builtin_readcyclecounter()
#define CHOL_T_VMWAIT asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define CHOL_T(k) do { } while (0)
#define CHOL_T_INIT do { } while (0)
#define CHOL_T_VMWAIT do { } while (0)
#endif
typedef double f64x4 __attribute__((ext_vector_type(4)));
struct CholLds {
  double D[2][kCholNB][kCholNB + 1];   // factored diagonal block (double buffered for the look-ahead)
  double Dinv[2][kCholNB];
  double Dt[kCholNB][kCholNB + 1];      // next diagonal block after its trailing update, on its way from MFMA layout to one row per lane
  double dinv_n[384];        // 1 / L[i][i] of every row, kept for block_backward (filled when a right-hand side is carried)
  unsigned short tile[328];  // q -> (tile row << 8 | tile column) of the lower-triangular 16 x 16 tiling, q = tr (tr + 1) / 2 + tc, tr < 25
  double P[kCholMaxN > 384 ? 384 : kCholMaxN][kCholNB + 1];  // panel rows of the current block column (n - kb - NB <= 384 rows handled per pass)
};

__device__ __forceinline__ double readlane_f64(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// 1 / sqrt(a) to about 1 ulp: hardware estimate + two Newton steps (each y += y (1/2 - a/2 y^2)).  The pivot of the block
// factorisation needs sqrt(a) and 1 / sqrt(a) on its critical path; this chain is a quarter of the length of sqrt + divide.
__device__ __forceinline__ double rsqrt_newton(double a) {
  double y = __builtin_amdgcn_rsq(a);
  const double h = 0.5 * a;
#pragma unroll
  for (int it = 0; it < 2; ++it) {  // v_rsq_f64 is good to 2^-23: two quadratic steps reach double precision
    const double e = __builtin_fma(-h * y, y, 0.5);
    y = __builtin_fma(y, e, y);
  }
  return y;
}

__device__ __forceinline__ bool block_cholesky(double* __restrict__ A, int n, int lda, CholLds& L, int* __restrict__ sflag, double* __restrict__ rhs = nullptr) {
  constexpr int NB = kCholNB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwave = (int)(blockDim.x >> 6);
  if (tid == 0) *sflag = 0;
  for (int q = tid; q < 325; q += blockDim.x) {
    int tr = (int)((sqrtf(8.0f * (float)q + 1.0f) - 1.0f) * 0.5f);
    while (tr * (tr + 1) / 2 > q) --tr;
    while ((tr + 1) * (tr + 2) / 2 <= q) ++tr;
    L.tile[q] = (unsigned short)((tr << 8) | (q - tr * (tr + 1) / 2));
  }
  __syncthreads();
  CHOL_T_INIT;
  // LOOK-AHEAD (matrices that fit one panel pass): the serial factorisation of the NEXT diagonal block runs on wave 0 while the
  // other waves do the trailing update of the current panel -- wave 0 first updates that one 16 x 16 tile, stages it in LDS,
  // re-reads it one row per lane and factors it in registers.  The diagonal block buffers are double buffered.
  const bool la = n - min(NB, n) + (rhs != nullptr ? 1 : 0) <= 384 && nwave >= 2;  // the first panel (the longest) fits one pass
  int cur = 0;
  // factor the 16 x 16 block held one row per lane in a[] (lane r = row r, identity rows pad a short block), publish it in
  // L.D[buf] / L.Dinv[buf] and in A
  auto factor_block = [&](double (&a)[NB], int kbd, int nbd, int buf) {
    const int r = lane;
    bool ok = true;
    double yinv = 1.0;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const double ajj = readlane_f64(a[j], j);
      if (!(ajj > 0.0)) ok = false;
      const double y = rsqrt_newton(ajj);
      const double lrj = a[j] * y;  // row j: a_jj / sqrt(a_jj) = sqrt(a_jj)
      if (r == j) yinv = y;
      a[j] = lrj;
#pragma unroll
      for (int c = j + 1; c < NB; ++c) {
        const double lcj = readlane_f64(lrj, c);
        a[c] = __builtin_fma(-lrj, lcj, a[c]);  // meaningful for r >= c only; the upper part of a[] is never read
      }
    }
    if (r < NB) {
#pragma unroll
      for (int c = 0; c < NB; ++c)
        if (c <= r) {
          L.D[buf][r][c] = a[c];
          if (r < nbd) A[(size_t)(kbd + r) * lda + kbd + c] = a[c];
        }
      L.Dinv[buf][r] = yinv;
      if (rhs != nullptr && r < nbd) L.dinv_n[kbd + r] = yinv;
    }
    if (!ok && lane == 0) *sflag = 1;
  };
  for (int kb = 0; kb < n; kb += NB) {
    const int nb = min(NB, n - kb);
    const int m = n - kb - nb;  // rows below the diagonal block
    const int me = m + (rhs != nullptr ? 1 : 0);
    auto panel_row = [&](int i) -> double* { return (rhs != nullptr && i == m) ? rhs + kb : A + (size_t)(kb + nb + i) * lda + kb; };
    double x[NB];
    const bool own = tid < min(me, 384);
    auto load_row = [&](const double* arow) {
#pragma unroll
      for (int c = 0; c < NB; ++c) x[c] = c < nb ? arow[c] : 0.0;
    };
    if (!la || kb == 0) {
      // waves that do not factor request their first panel row now: it does not depend on the factor
      if (wave != 0 && own) load_row(panel_row(tid));
      if (wave == 0) {  // diagonal block from memory
        const int r = lane;
        double a[NB];
#pragma unroll
        for (int c = 0; c < NB; ++c) a[c] = (r < nb && c <= r) ? A[(size_t)(kb + r) * lda + kb + c] : ((r == c) ? 1.0 : 0.0);
        factor_block(a, kb, nb, cur);
        if (own) load_row(panel_row(tid));
      }
      CHOL_T(0);
      __syncthreads();
      CHOL_T(1);
    } else if (own) {
      load_row(panel_row(tid));  // the block was factored during the previous trailing update (barrier at its end)
    }
    if (*sflag) return false;
    if (me <= 0) break;
    for (int p0 = 0; p0 < me; p0 += 384) {  // panel passes (one pass unless n > 400)
      const int mp = min(384, me - p0);
      for (int i = tid; i < mp; i += blockDim.x) {  // panel solve: row i of L21 = A21 L11^-T (all 16 columns; padding solves to 0)
        double* arow = panel_row(p0 + i);
        if (!(p0 == 0 && i == tid)) load_row(arow);
        // keep the 136 reads of the diagonal block inside the loop body: hoisted out of it they do not fit the register file
        asm volatile("" ::: "memory");
#pragma unroll
        for (int c = 0; c < NB; ++c) {
          double v = x[c];
#pragma unroll
          for (int q = 0; q < c; ++q) v = __builtin_fma(-x[q], L.D[cur][c][q], v);
          v *= L.Dinv[cur][c];
          x[c] = v;
          L.P[i][c] = v;
        }
#pragma unroll
        for (int c = 0; c < NB; ++c)
          if (c < nb) arow[c] = x[c];
      }
      for (int i = mp + tid; i < (((rhs != nullptr ? m : mp) + 15) & ~15); i += blockDim.x) {  // rows that only pad the last 16-row tile (the rhs row, index m, multiplies results that are discarded)
#pragma unroll
        for (int c = 0; c < NB; ++c) L.P[i][c] = 0.0;
      }
      CHOL_T(2);
      __syncthreads();
      CHOL_T(3);
      // trailing update restricted to rows of this pass: A[i][k] -= P[i] . P[k] for kb+nb <= k <= i, 16 x 16 tiles on the matrix
      // cores (operand element (row l & 15, k = l >> 4) for both factors; result row = (l >> 4) + 4 reg, column = l & 15).
      // Columns k that belong to an earlier pass need their panel rows too, so with more than one pass we fall back to reading
      // L21 from A (below).
      if (p0 == 0) {
        // The phase is instruction-issue bound (two waves per SIMD), so the tile loop is kept lean: the tile coordinates come
        // from a table and are made scalar, each access is one instruction (uniform tile base + per-lane offset computed once per
        // panel), interior tiles carry no predicates, and diagonal tiles are updated in full -- their strictly upper part lands in
        // the (never referenced) upper triangle of A.  Four tiles per wave are in flight so that the read half of their
        // read-modify-writes overlaps the matrix-core work.  The right-hand-side row is not part of the tiling (below).
        const int mt = rhs != nullptr ? m : mp;  // matrix rows of this pass (the right-hand-side row is handled separately)
        const int T = (mt + 15) / 16;
        const int ntile = T * (T + 1) / 2;
        const int lr = lane & 15, lk = lane >> 4;
        const int wave_u = __builtin_amdgcn_readfirstlane(wave);
        const int vofs = lk * lda + lr, rstep = 4 * lda;
        const double* Pl = &L.P[0][0] + (lr * (NB + 1) + lk);
        double* Abase = A + (size_t)(kb + nb) * lda + kb + nb;
        constexpr int TU = 4;
        const bool ahead = la && m > 0;           // wave 0: tile 0 (the next diagonal block) and its factorisation
        const int q_first = ahead ? 1 : 0;        // tiles left to the tiling loop
        const int w_first = ahead ? 1 : 0, w_cnt = nwave - w_first;
        if (ahead && wave_u == 0) {
          double val[4];
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) val[rg] = (lk + 4 * rg < mt && lr < mt) ? Abase[vofs + rg * rstep] : 0.0;
          f64x4 acc = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int k0 = 0; k0 < NB; k0 += 4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Pl[k0], Pl[k0], acc, 0, 0, 0);
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) L.Dt[lk + 4 * rg][lr] = val[rg] - acc[rg];
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          const int nbn = min(NB, m), r = lane;
          double a[NB];
#pragma unroll
          for (int c = 0; c < NB; ++c) a[c] = (r < nbn && c <= r) ? L.Dt[r][c] : ((r == c) ? 1.0 : 0.0);
          factor_block(a, kb + nb, nbn, cur ^ 1);
        }
        if (wave_u >= w_first) {
          for (int q0 = q_first + (wave_u - w_first) * TU; q0 < ntile; q0 += w_cnt * TU) {
            int trv[TU], tcv[TU];
            double cv[TU][4];
#pragma unroll
            for (int u = 0; u < TU; ++u) {
              const int t = __builtin_amdgcn_readfirstlane((int)L.tile[min(q0 + u, ntile - 1)]);
              trv[u] = t >> 8;
              tcv[u] = t & 255;
              const double* tb = Abase + (size_t)(trv[u] * 16) * lda + tcv[u] * 16;
              if (q0 + u < ntile) {
                if (trv[u] * 16 + 16 <= mt) {
#pragma unroll
                  for (int rg = 0; rg < 4; ++rg) cv[u][rg] = tb[vofs + rg * rstep];
                } else {
#pragma unroll
                  for (int rg = 0; rg < 4; ++rg)
                    cv[u][rg] = (trv[u] * 16 + lk + 4 * rg < mt && tcv[u] * 16 + lr < mt) ? tb[vofs + rg * rstep] : 0.0;
                }
              }
            }
#pragma unroll
            for (int u = 0; u < TU; ++u) {
              if (q0 + u >= ntile) break;
              const double* pa = Pl + trv[u] * 16 * (NB + 1);
              const double* pb = Pl + tcv[u] * 16 * (NB + 1);
              f64x4 acc = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
              for (int k0 = 0; k0 < NB; k0 += 4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[k0], pb[k0], acc, 0, 0, 0);
              double* tb = Abase + (size_t)(trv[u] * 16) * lda + tcv[u] * 16;
              if (trv[u] * 16 + 16 <= mt) {
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) tb[vofs + rg * rstep] = cv[u][rg] - acc[rg];
              } else {
#pragma unroll
                for (int rg = 0; rg < 4; ++rg)
                  if (trv[u] * 16 + lk + 4 * rg < mt && tcv[u] * 16 + lr < mt) tb[vofs + rg * rstep] = cv[u][rg] - acc[rg];
              }
            }
          }
          if (rhs != nullptr) {  // the right-hand-side row: rhs[k] -= P[m] . P[k]
            for (int k = tid - w_first * 64; k < m; k += w_cnt * 64) {
              double acc = 0.0;
#pragma unroll
              for (int c = 0; c < NB; ++c) acc = __builtin_fma(L.P[m][c], L.P[k][c], acc);
              rhs[kb + nb + k] -= acc;
            }
          }
        }
      }
      if (p0 != 0 || me > 384) {  // generic (slow) path for matrices larger than one panel pass: element-wise from global memory
        for (long q = tid; q < (long)mp * m; q += blockDim.x) {
          const int i = p0 + (int)(q / m), k = (int)(q % m);
          if (k > i || (p0 == 0 && k < mp)) continue;
          double sacc = 0.0;
          for (int c = 0; c < nb; ++c) sacc += A[(size_t)(kb + nb + i) * lda + kb + c] * A[(size_t)(kb + nb + k) * lda + kb + c];
          A[(size_t)(kb + nb + i) * lda + kb + nb + k] -= sacc;
        }
      }
      CHOL_T(4);
      __syncthreads();
      CHOL_T(5);
    }
    if (la) cur ^= 1;
  }
  __syncthreads();
  return true;
}

// Forward substitution L X = B for NC right-hand sides by ONE wave, X / B held in LDS as X[row * ldx + col].  Lanes split the
// dot product of a row (coalesced reads of the L row from HBM / L2), a shuffle tree reduces it, lane 0 finishes the row.
// No workgroup barrier is involved, so several waves can solve different column groups of the same system concurrently.
// Rows < row0 of B must be zero (they are skipped).
template <int NC>
__device__ __forceinline__ void wave_forward(const double* __restrict__ L, int n, int lda, double* __restrict__ X, int ldx, int row0) {
  const int lane = threadIdx.x & 63;
  for (int i = row0; i < n; ++i) {
    const double* Lr = L + (size_t)i * lda;
    double acc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] = 0.0;
    for (int k = row0 + lane; k < i; k += 64) {
      const double l = Lr[k];
#pragma unroll
      for (int c = 0; c < NC; ++c) acc[c] += l * X[(size_t)k * ldx + c];
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) acc[c] += __shfl_down(acc[c], off, 64);
    }
    if (lane == 0) {
      const double d = Lr[i];
#pragma unroll
      for (int c = 0; c < NC; ++c) X[(size_t)i * ldx + c] = (X[(size_t)i * ldx + c] - acc[c]) / d;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

// Backward substitution L^T x = y (one right-hand side, in place in LDS) by one wave
__device__ __forceinline__ void wave_backward(const double* __restrict__ L, int n, int lda, double* __restrict__ x) {
  const int lane = threadIdx.x & 63;
  for (int i = n - 1; i >= 0; --i) {
    double acc = 0.0;
    for (int k = i + 1 + lane; k < n; k += 64) acc += L[(size_t)k * lda + i] * x[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (lane == 0) x[i] = (x[i] - acc) / L[(size_t)i * lda + i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

// Backward substitution L^T x = y for one right-hand side in LDS by the whole workgroup (n <= blockDim.x, n < 384), after a
// block_cholesky that carried a right-hand side (C.dinv_n holds the inverse diagonal).  Thread i keeps x[i] in a register and
// column i of the current 16-row block of L in registers (coalesced loads; the next block is requested before the current one is
// used).  Per block the 16 owners publish their entries and the 16 x 16 diagonal block in LDS (double buffered: ONE barrier per
// block), every wave then solves the small transposed system redundantly in registers (v_readlane broadcasts) and each thread
// to the left removes the 16 new unknowns from its own entry.
__device__ __forceinline__ void block_backward(const double* __restrict__ Lm, int n, int lda, double* __restrict__ x, CholLds& C) {
  constexpr int NB = kCholNB;
  const int tid = threadIdx.x, lane = tid & 63;
  double cur[NB], nxt[NB];
  const int kb_last = ((n - 1) / NB) * NB;
  auto fetch = [&](int kb, double* dst) {
    const int nb = min(NB, n - kb);
#pragma unroll
    for (int c = 0; c < NB; ++c) dst[c] = (c < nb && tid <= kb + c) ? Lm[(size_t)(kb + c) * lda + tid] : 0.0;
  };
  fetch(kb_last, cur);
  double xv = tid < n ? x[tid] : 0.0;
  // staging: buffer b = rows [b * 17, b * 17 + 16) of the panel array hold the diagonal block, row b * 17 + 16 the 16 entries of x
  int buf = 0;
  const int r = lane & 15;
  for (int kb = kb_last; kb >= 0; kb -= NB) {
    const int nb = min(NB, n - kb);
    if (kb >= NB) fetch(kb - NB, nxt);
    double (*Db)[NB + 1] = &C.P[buf * (NB + 1)];
    if (tid >= kb && tid < kb + nb) {
#pragma unroll
      for (int c = 0; c < NB; ++c) Db[c][tid - kb] = cur[c];  // Db[c][q] = L[kb + c][kb + q], q <= c
      Db[NB][tid - kb] = xv;
    }
    __syncthreads();
    double v = r < nb ? Db[NB][r] : 0.0;
    const double dinv = r < nb ? C.dinv_n[kb + r] : 1.0;
    double dcol[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) dcol[j] = (j > r && j < nb) ? Db[j][r] : 0.0;
#pragma unroll
    for (int j = NB - 1; j >= 0; --j) {
      const double xj = readlane_f64(v, j) * readlane_f64(dinv, j);
      if (r == j) v = xj;
      v = __builtin_fma(-dcol[j], xj, v);  // dcol[j] is zero for j <= r
    }
    if (tid < nb) x[kb + tid] = v;
    if (tid < kb) {
#pragma unroll
      for (int c = 0; c < NB; ++c) xv = __builtin_fma(-cur[c], readlane_f64(v, c), xv);
    }
#pragma unroll
    for (int c = 0; c < NB; ++c) cur[c] = nxt[c];
    buf ^= 1;
  }
  __syncthreads();
}

__device__ __forceinline__ double block_sum(double v, double* sred) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0.0;
  for (int q = 0; q < (int)(blockDim.x >> 6); ++q) t += sred[q];
  return t;
}
__device__ __forceinline__ double block_max(double v, double* sred) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = sred[0];
  for (int q = 1; q < (int)(blockDim.x >> 6); ++q) t = fmax(t, sred[q]);
  return t;
}

// Blocked forward substitution L X = B for 16 right-hand-side columns by one workgroup of 256 threads, block rows jb .. nblk-1
// (B is zero above block row jb).  For block row i the update B_i - sum_k L_ik X_k runs on the fp64 matrix cores (the k range is
// dealt to the 4 waves; operands: L straight from L2 -- the next block row is requested while the triangular solve of the current
// one runs -- and the X_k tiles from LDS), the 4 partial tiles are added in LDS in wave order, and one wave solves the 16 x 16
// triangular system for all 16 columns at once (lane = column, L_ii entries broadcast from LDS).
// X_k lands in Xc + (k - jb) * 16 * 17 as [r][c] at r * 17 + c; rows past n are zero.  On entry Xc holds B in the same layout
// (the caller fills it with all lanes; a barrier is taken here before it is read).
struct Solve16Lds {
  double Pt[4][16][17];  // partial update tiles of the 4 waves
  double Dl[16][17];     // L_ii
};
__device__ __forceinline__ void blocked_lower_solve16(const double* __restrict__ Lm, int n, int jb, double* __restrict__ Xc, Solve16Lds& sh) {
  const int nblk = (n + 15) / 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lk = lane >> 4;
  // operand prefetch: for block row i, wave v owns k = jb + v, jb + v + 4, ...; element (row i*16 + lr, col k*16 + k0 + lk), k0 = 0,4,8,12
  constexpr int KMAX = 16;  // blocks per wave per row: supports nblk - jb <= 64 (n <= 1024)
  double la[KMAX][4];       // one buffer: the next block row is requested right after the matrix cores consumed the current one
  auto fetch = [&](int i) {
#pragma unroll
    for (int t = 0; t < KMAX; ++t) {
      const int k = jb + wave + 4 * t;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = i * 16 + lr, col = k * 16 + q * 4 + lk;
        la[t][q] = (k < i && row < n) ? Lm[(size_t)row * n + col] : 0.0;
      }
    }
  };
  for (int i = jb; i < nblk; ++i) {
    const int nb = min(16, n - i * 16);
    {  // diagonal block of L to LDS (lower part; rows past n padded with the identity)
      const int r = tid >> 4, c = tid & 15;
      const int row = i * 16 + r, col = i * 16 + c;
      sh.Dl[r][c] = (c <= r && row < n) ? Lm[(size_t)row * n + col] : ((r == c) ? 1.0 : 0.0);
    }
    f64x4 acc = f64x4{0.0, 0.0, 0.0, 0.0};
    if (i > jb) {
#pragma unroll
      for (int t = 0; t < KMAX; ++t) {
        const int k = jb + wave + 4 * t;
        if (k < i) {
          const double* xk = Xc + (size_t)(k - jb) * 16 * 17;
#pragma unroll
          for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(la[t][q], xk[(q * 4 + lk) * 17 + lr], acc, 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) sh.Pt[wave][lk + 4 * rg][lr] = acc[rg];
    if (i + 1 < nblk) fetch(i + 1);  // lands while the triangular solve below runs
    __syncthreads();
    if (wave == 0 && lane < 16) {  // B_i - sum of partials (wave order), then L_ii X_i = ., column `lane`
      const int c = lane;
      double x[16];
      double* xi = Xc + (size_t)(i - jb) * 16 * 17;
      const double dinv = 1.0 / sh.Dl[c][c];  // lane c: reciprocal of diagonal entry c, one division off the 16-step chain
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const double bsum = ((sh.Pt[0][r][c] + sh.Pt[1][r][c]) + sh.Pt[2][r][c]) + sh.Pt[3][r][c];
        double v = xi[r * 17 + c] - bsum;
#pragma unroll
        for (int q = 0; q < 16; ++q)
          if (q < r) v = __builtin_fma(-sh.Dl[r][q], x[q], v);
        v = v * readlane_f64(dinv, r);
        x[r] = (r < nb) ? v : 0.0;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) xi[r * 17 + c] = x[r];
    }
    __syncthreads();
  }
}

// =============================================================================================== Gram stage

// Small dense products on the fp64 matrix cores for S x S matrices resident in L2 (S <= 160), 16 x 16 output tiles dealt to
// the waves of the workgroup, operands read straight from global memory (for every k-step of 4 a lane supplies one element of
// each operand; the loop is unrolled so several steps of loads are in flight).  Result layout of v_mfma_f64_16x16x4_f64:
// col = lane & 15, row = (lane >> 4) + 4 reg.
//   small_gemm_tn_sym: C = B^T B for a LOWER-triangular B (B[k][i] = 0 for k < i): only k >= the tile's row block contributes;
//                      lower tiles are computed and mirrored.
//   small_gemm_nn:     C = A B.
__device__ __forceinline__ void small_gemm_tn_sym(const double* __restrict__ B, double* __restrict__ C, int S) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = (int)(blockDim.x >> 6);
  const int lr = lane & 15, lk = lane >> 4;
  const int T = (S + 15) / 16;
  for (int q = wave; q < T * (T + 1) / 2; q += nwave) {
    int tr = (int)((sqrtf(8.0f * (float)q + 1.0f) - 1.0f) * 0.5f);
    while (tr * (tr + 1) / 2 > q) --tr;
    while ((tr + 1) * (tr + 2) / 2 <= q) ++tr;
    const int tc = q - tr * (tr + 1) / 2;
    const int i = tr * 16 + lr, j = tc * 16 + lr;
    f64x4 acc = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
    for (int k0 = tr * 16; k0 < S; k0 += 4) {
      const int k = k0 + lk;
      const double av = (k < S && i < S) ? B[(size_t)k * S + i] : 0.0;
      const double bv = (k < S && j < S) ? B[(size_t)k * S + j] : 0.0;
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
    }
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const int r = tr * 16 + lk + 4 * rg, c = tc * 16 + lr;
      if (r < S && c < S) {
        C[(size_t)r * S + c] = acc[rg];
        C[(size_t)c * S + r] = acc[rg];
      }
    }
  }
}
__device__ __forceinline__ void small_gemm_nn(const double* __restrict__ A, const double* __restrict__ B, double* __restrict__ C, int S) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = (int)(blockDim.x >> 6);
  const int lr = lane & 15, lk = lane >> 4;
  const int T = (S + 15) / 16;
  for (int q = wave; q < T * T; q += nwave) {
    const int ti = q / T, tj = q - ti * T;
    const int i = ti * 16 + lr, j = tj * 16 + lr;
    f64x4 acc = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
    for (int k0 = 0; k0 < S; k0 += 4) {
      const int k = k0 + lk;
      const double av = (k < S && i < S) ? A[(size_t)i * S + k] : 0.0;
      const double bv = (k < S && j < S) ? B[(size_t)k * S + j] : 0.0;
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
    }
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const int r = ti * 16 + lk + 4 * rg, c = tj * 16 + lr;
      if (r < S && c < S) C[(size_t)r * S + c] = acc[rg];
    }
  }
}

// grid: (6 channels, windows), block 256.  preint.h:832-866 for one channel: K + sz2 I = L L^T (block_cholesky), L^-1 by the
// blocked 16-column forward substitution, K^-1 = L^-T L^-1, K K^-1 and K_int K^-1 on the matrix cores.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gram_kernel(const UgpmWin* __restrict__ wins) {
  const WinPart wp = xcd_win_part();
  const UgpmWin w = load_win(wins, wp.win);
  if (*w.status != 0) return;
  const int c = wp.part, S = w.S;
  const double l2 = w.hyper[c * 4 + 0], sf2 = w.hyper[c * 4 + 1], sz2 = w.hyper[c * 4 + 2];
  double* A = w.Kinv + (size_t)c * S * S;    // K + sz2 I -> L -> finally K^-1
  double* B = w.KKinv + (size_t)c * S * S;   // L^-1 scratch -> finally K K^-1
  double* Kc = w.JtJ + (size_t)c * S * S;    // scratch: the Gram matrix itself (JtJ / lhs are idle until the LM stage, 9 S^2 each)
  const double* st = w.state_t;
  __shared__ int sflag;
  __shared__ CholLds chol;
  __shared__ Solve16Lds sh;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwave = (int)(blockDim.x >> 6);
  for (int i = wave; i < S; i += nwave)
    for (int j = lane; j < S; j += 64) {
      const double k = se_k(st[i], st[j], l2, sf2);
      Kc[(size_t)i * S + j] = k;
      A[(size_t)i * S + j] = k + (i == j ? sz2 : 0.0);
    }
  __syncthreads();
  if (!block_cholesky(A, S, S, chol, &sflag)) {
    if (threadIdx.x == 0) *w.status = -6;
    return;
  }
  // L^-1, 16 columns at a time (the panel buffer of the factorisation is free: S + 16 <= 384 rows of 17)
  {
    double* Xc = &chol.P[0][0];
    const int nblk = (S + 15) / 16;
    for (int jb = 0; jb < nblk; ++jb) {
      const int rows = (nblk - jb) * 16;
      for (int q = threadIdx.x; q < rows * 16; q += blockDim.x) Xc[(size_t)(q >> 4) * 17 + (q & 15)] = (q >> 4) == (q & 15) ? 1.0 : 0.0;
      blocked_lower_solve16(A, S, jb, Xc, sh);
      for (int i = wave; i < S; i += nwave) {  // column block jb of L^-1: zeros above the diagonal block
        if (lane < 16 && jb * 16 + lane < S) B[(size_t)i * S + jb * 16 + lane] = i >= jb * 16 ? Xc[(size_t)(i - jb * 16) * 17 + lane] : 0.0;
      }
      __syncthreads();
    }
  }
  __threadfence_block();
  __syncthreads();
  small_gemm_tn_sym(B, A, S);  // K^-1 = L^-T L^-1
  __syncthreads();
  small_gemm_nn(Kc, A, B, S);  // K K^-1, preint.h:838
  __syncthreads();
  for (int j = wave; j < S; j += nwave) {  // preint.h:846-864: diag(K K^-1 K) by rows (K is symmetric)
    double s = 0.0;
    for (int k = lane; k < S; k += 64) s += B[(size_t)j * S + k] * Kc[(size_t)j * S + k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) {
      double v = -s + sf2 + sz2;
      if (v <= 0) v = sz2;
      w.var[c * S + j] = v;
      w.sstd[c * S + j] = sqrt(v);
      double wt = sqrt(1.0 / (1000.0 * v));  // cost_functions.h:31 with the 1000 x variance of preint.h:853, 864
      if (isnan(wt)) wt = 1.0;
      w.wgp[c * S + j] = wt;
    }
  }
  if (c < 3) {  // K_int K^-1, preint.h:842-844
    double* Cq = w.KintKinv + (size_t)c * S * S;
    double* Ki = w.lhs + (size_t)c * S * S;  // scratch: K_int
    for (int i = wave; i < S; i += nwave)
      for (int j = lane; j < S; j += 64) Ki[(size_t)i * S + j] = se_kint(w.start_t, st[i], st[j], l2, sf2);
    __syncthreads();
    small_gemm_nn(Ki, A, Cq, S);
  }
}

// grid: (12 tables, windows, row tiles of kCrossRows), block 256.  Tables 0-2 K_s K^-1 (gyro stamps), 3-5 K_s_int K^-1 (gyro stamps),
// 6-8 K_s_int K^-1 (velocity stamps, rotation channels), 9-11 K_s K^-1 (velocity stamps, velocity channels).
constexpr int kCrossRows = 32;  // table rows per workgroup (8 made 25 k tiny workgroups per batch: dispatch bound)
__global__ __launch_bounds__(256) void cross_kernel(const UgpmWin* __restrict__ wins) {
  const WinPart wp = xcd_win_part();
  const UgpmWin w = load_win(wins, wp.win);
  if (*w.status != 0) return;
  const int tab = wp.part, S = w.S;
  const int c = tab % 3, kind = tab / 3;
  const int ch = kind == 3 ? 3 + c : c;
  const int N = kind < 2 ? w.G : w.V;
  const double* tt = kind < 2 ? w.gyr_t : w.vel_t;
  double* out = (kind == 0 ? w.KsKinv : kind == 1 ? w.KsIntKinv : kind == 2 ? w.KgyrIntKinv : w.KvelKinv) + (size_t)c * N * S;
  const double* Ki = w.Kinv + (size_t)ch * S * S;
  const double l2 = w.hyper[ch * 4 + 0], sf2 = w.hyper[ch * 4 + 1];
  const bool integral = (kind == 1 || kind == 2);
  // row stride of the staged kernel rows: 4 doubles past a multiple of 32 banks that holds S columns (dynamic LDS: 25 KB at S = 66 instead of
  // the 42 KB a stride for the largest S would take -- these workgroups share their CUs with the scan matcher's kernels)
  const int LDK = ((S + 31) / 32) * 32 + 4;
  extern __shared__ double ks_dyn[];
  double* ks = ks_dyn;  // [kCrossRows][LDK]
  const int row0 = wp.z * kCrossRows;
  if (row0 >= N) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int r = wave; r < kCrossRows; r += 4) {  // kernel rows k(t_n, state_t[.]) of this tile; rows past N and columns past S are zero
    const int n = row0 + r;
    for (int k = lane; k < LDK; k += 64) {
      double v = 0.0;
      if (n < N && k < S) v = integral ? se_kint(w.start_t, tt[n], w.state_t[k], l2, sf2) : se_k(tt[n], w.state_t[k], l2, sf2);
      ks[(size_t)r * LDK + k] = v;
    }
  }
  __syncthreads();
  // out tile = ks (32 x S) . K^-1 (S x S) on the matrix cores: 2 x ceil(S / 16) tiles dealt to the 4 waves
  const int lr = lane & 15, lk = lane >> 4;
  const int TJ = (S + 15) / 16;
  for (int q = wave; q < 2 * TJ; q += 4) {
    const int ti = q / TJ, tj = q - ti * TJ;
    const int j = tj * 16 + lr;
    f64x4 acc = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
    for (int k0 = 0; k0 < S; k0 += 4) {
      const int k = k0 + lk;
      const double av = ks[(size_t)(ti * 16 + lr) * LDK + k];  // zero for k >= S
      const double bv = (k < S && j < S) ? Ki[(size_t)k * S + j] : 0.0;
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
    }
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const int n = row0 + ti * 16 + lk + 4 * rg;
      if (n < N && j < S) out[(size_t)n * S + j] = acc[rg];
    }
  }
}

// =============================================================================================== cost functions

// d[J_r(r) dr] / d[r, dr] (cost_functions.h:73-145; the reference's symbolic dump is this derivative written out):
//   d/dr_k = -A' (r_k/n) (r x dr) - A (e_k x dr) + B' (r_k/n) (r x (r x dr)) + B (e_k x (r x dr) + r x (e_k x dr)),  d/d(dr) = J_r(r)
__device__ void jacobian_res(V3 r, V3 dr, double D[3][6]) {
  const double n2 = r.x * r.x + r.y * r.y + r.z * r.z;
  const double n = sqrt(n2);
  if (n > kExpTol) {
    const double s = sin(n), c = cos(n);
    const double A = (1.0 - c) / n2, B = (n - s) / (n2 * n);
    const double dA = (n * s - 2.0 * (1.0 - c)) / (n2 * n);
    const double dB = ((1.0 - c) * n - 3.0 * (n - s)) / (n2 * n2);
    const V3 rxd = cross(r, dr), rxrxd = cross(r, rxd);
    const double rk[3] = {r.x, r.y, r.z};
    for (int k = 0; k < 3; ++k) {
      const V3 e = v3(k == 0, k == 1, k == 2);
      const double dn = rk[k] / n;
      const V3 exd = cross(e, dr);
      const V3 t = (-dA * dn) * rxd + (-A) * exd + (dB * dn) * rxrxd + B * (cross(e, rxd) + cross(r, exd));
      D[0][k] = t.x; D[1][k] = t.y; D[2][k] = t.z;
    }
    const M3 J = Jr(r);
    for (int i = 0; i < 3; ++i)
      for (int k = 0; k < 3; ++k) D[i][3 + k] = J.m[i * 3 + k];
  } else {
    const M3 S = skew(dr);
    for (int i = 0; i < 3; ++i)
      for (int k = 0; k < 3; ++k) {
        D[i][k] = 0.5 * S.m[i * 3 + k];
        D[i][3 + k] = (i == k) ? 1.0 : 0.0;
      }
  }
}

__device__ __forceinline__ double row_dot(const double* __restrict__ row, const double* __restrict__ s, int S) {
  double t = 0.0;
  for (int j = 0; j < S; ++j) t += row[j] * s[j];
  return t;
}

// Six table-row dot products of one sample by one wave (lanes along the rows: coalesced; all twelve loads of an iteration are in
// flight together), results in every lane.  rows a[c] (c = 0..2) meet va + c S, rows b[c] meet vb + c S.
__device__ __forceinline__ void wave_row_dots6(const double* __restrict__ a, size_t a_stride, const double* __restrict__ va,
                                               const double* __restrict__ b, size_t b_stride, const double* __restrict__ vb, int S, double (&out)[6]) {
  double t[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  for (int j = threadIdx.x & 63; j < S; j += 64) {
    double ra[3], rb[3], xa[3], xb[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      ra[c] = a[c * a_stride + j];
      rb[c] = b[c * b_stride + j];
      xa[c] = va[(size_t)c * S + j];
      xb[c] = vb[(size_t)c * S + j];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      t[c] += ra[c] * xa[c];
      t[3 + c] += rb[c] * xb[c];
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
#pragma unroll
    for (int q = 0; q < 6; ++q) t[q] += __shfl_xor(t[q], off, 64);
#pragma unroll
  for (int q = 0; q < 6; ++q) out[q] = t[q];
}

// three table rows (stride apart) against v + c S, by one wave; results in every lane
__device__ __forceinline__ void wave_row_dots3(const double* __restrict__ a, size_t a_stride, const double* __restrict__ va, int S, double (&out)[3]) {
  double t[3] = {0.0, 0.0, 0.0};
  for (int j = threadIdx.x & 63; j < S; j += 64) {
    double ra[3], xa[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      ra[c] = a[c * a_stride + j];
      xa[c] = va[(size_t)c * S + j];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) t[c] += ra[c] * xa[c];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
#pragma unroll
    for (int q = 0; q < 3; ++q) t[q] += __shfl_xor(t[q], off, 64);
#pragma unroll
  for (int q = 0; q < 3; ++q) out[q] = t[q];
}

// lm_step_kernel may run as several workgroups per window, all of which read the solver's control words; what the step consumes
// ("a fresh linearisation arrived", "first evaluation") and counts (iterations) is therefore written here, by ONE thread of the
// candidate evaluation that follows every step in stream order -- nothing in between reads these words.
__device__ __forceinline__ void lm_step_bookkeeping(const UgpmWin& w) {
  w.lmi[3] = 0;
  w.lmi[8] = 0;
  w.lmi[0] += 1;
}

// Acceptance test of a trust-region step (Ceres 2.1 TrustRegionMinimizer: tolerances of preint.h:943-948, rho > 1e-3, radius
// update), evaluated from the candidate residuals res_new.  It is fused into the kernels that re-linearise after the step, whose
// grid has several workgroups per window: EVERY workgroup evaluates the (deterministic) decision for itself from inputs none of
// them writes (res_new, lmc[0], lmc[4], lmc[8..13], lmi[9..11]), and only the workgroup with commit == true stores the new solver state.
// Returns 1 accepted, 0 rejected / invalid step, -1 terminated.  All threads of the workgroup must call it.
__device__ int lm_decide_block(const UgpmWin& w, int m, int n, bool commit, double* sred /* [8] LDS */) {
  // the step came from lm_step_kernel in one piece (problem #1) or as kVelBlocks independent diagonal blocks (problem #2)
  const int nblk = w.lmi[7] == 1 ? kVelBlocks : 1;
  bool step_ok = true;
  double mcc = 0.0, sn2 = 0.0;
  for (int q = 0; q < nblk; ++q) {
    step_ok = step_ok && w.lmi[9 + q] != 0;
    mcc += w.lmc[8 + q];
    sn2 += w.lmc[11 + q];
  }
  if (!(mcc > 0.0)) step_ok = false;
  const double sn = sqrt(sn2);
  if (!step_ok) {  // StepIsInvalid
    if (commit && threadIdx.x == 0) {
      w.lmc[2] = w.lmc[2] / w.lmc[3];
      w.lmc[3] *= 2.0;
      w.lmi[2] = 1;
    }
    return 0;
  }
  double c = 0.0;
  const int nacc = min((int)blockDim.x, 256);  // 256 accumulators in every caller: the sum does not depend on the caller's block size
  if ((int)threadIdx.x < nacc)
    for (int k = threadIdx.x; k < m; k += nacc) c += w.res_new[k] * w.res_new[k];
  const double cost_new = 0.5 * block_sum(c, sred);
  const double cost = w.lmc[0];
  const double cost_change = cost - cost_new;
  int verdict;
  double rho = 0.0;
  if (sn <= 1e-8 * (w.lmc[4] + 1e-8)) verdict = -2;                     // parameter_tolerance
  else if (fabs(cost_change) <= 1e-10 * cost) verdict = -1;             // function_tolerance, preint.h:948
  else {
    rho = cost_change / mcc;
    verdict = rho > 1e-3 ? 1 : 0;                                       // min_relative_decrease
  }
  __syncthreads();  // every thread has read the solver state before the committing workgroup's thread 0 rewrites it
  if (commit) {
    if (threadIdx.x == 0) {
      w.lmc[1] = cost_new;
      if (verdict < 0) {
        w.lmi[1] = 1;
        w.lmi[5] = verdict == -2 ? 2 : 1;
      } else if (verdict == 1) {
        w.lmi[6] += 1;
        w.lmi[3] = 1;
        w.lmc[2] = fmin(1e16, w.lmc[2] / fmax(1.0 / 3.0, 1.0 - pow(2.0 * rho - 1.0, 3)));
        w.lmc[3] = 2.0;
        w.lmi[2] = 0;
      } else {
        w.lmc[2] = w.lmc[2] / w.lmc[3];
        w.lmc[3] *= 2.0;
        w.lmi[2] = 1;
      }
    }
    if (verdict == 1)
      for (int j = threadIdx.x; j < n; j += blockDim.x) w.lmv[5 * (size_t)n + j] = w.lmv[6 * (size_t)n + j];
  }
  return verdict < 0 ? -1 : verdict;
}

// GpNorm residual rows [0, S) of channel block `blk` (cost_functions.h:47-57) and, once, its constant Jacobian block
// J(r, c) = (KKinv - I)(c, r) w[r]  (cost_functions.h:36-42 as seen by the solver through the column-major map at :63-64)
__device__ void gpnorm_rows(const UgpmWin& w, int ch, const double* __restrict__ s, double* __restrict__ res, double* __restrict__ J, int ldj, int col0, bool writeJ) {
  const int S = w.S;
  const double* KK = w.KKinv + (size_t)ch * S * S;
  const double* wt = w.wgp + (size_t)ch * S;
  if ((int)blockIdx.y != ch % (int)gridDim.y) return;  // one of the workgroups of this window takes the channel
  for (int i = threadIdx.x; i < S; i += blockDim.x) res[i] = (row_dot(KK + (size_t)i * S, s, S) - s[i]) * wt[i];
  if (writeJ)
    for (int q = threadIdx.x; q < S * S; q += blockDim.x) {
      const int i = q / S, j = q % S;
      J[(size_t)i * ldj + col0 + j] = (KK[(size_t)j * S + i] - (i == j ? 1.0 : 0.0)) * wt[i];
    }
}

// Problem #1 (preint.h:872-952): unknowns x = [s_dr0 | s_dr1 | s_dr2] (3S); rows = 3 GpNorm blocks (3S) then RotCost (3G).
// grid: (windows, splits), block 256: the workgroups of one window share its samples.  mode: 0 residual at x_new -> res_new; 1 (after
// the acceptance test) residual + Jacobian at x_new -> res, Jrot; 2 as 1 at x and also (re)writes the constant GpNorm blocks and zeroes
// the rest (first evaluation); 3 = 0 and the Jacobian rows of 1 in ONE launch, BEFORE the acceptance test: residual at x_new ->
// res_new, Jacobian at x_new -> Jrot.  Nothing but the J^T J launch that follows reads Jrot, and that launch takes the acceptance
// test itself (ata_kernel, decide) and leaves J^T J and g alone after a rejected step, so the speculative rows cost one write of
// Jrot on the rare rejection and save a launch -- and the second pass over the kernel tables -- on every iteration.
__global__ __launch_bounds__(256) void rot_eval_kernel(const UgpmWin* __restrict__ wins, int mode) {
  const UgpmWin w = load_win(wins, blockIdx.x);
  if (*w.status != 0 || w.lmi[1]) return;
  const int S = w.S, G = w.G, n = 3 * S;
  __shared__ double sred[8];
  if ((mode == 0 || mode == 3) && blockIdx.y == 0 && threadIdx.x == 0) lm_step_bookkeeping(w);
  if (mode == 1) {  // step acceptance (see lm_decide_block), then the Jacobian only after an accepted step -- at x_new, which is
                    // what the committing workgroup is copying into x meanwhile
    if (lm_decide_block(w, 3 * S + 3 * G, n, blockIdx.y == 0, sred) != 1) return;
  }
  const double* x = mode == 2 ? w.lmv + 5 * (size_t)n : w.lmv + 6 * (size_t)n;
  double* res = (mode == 0 || mode == 3) ? w.res_new : w.res;
  double* J = w.Jrot;
  const int i_lo = (int)(((long)G * blockIdx.y) / gridDim.y), i_hi = (int)(((long)G * (blockIdx.y + 1)) / gridDim.y);  // this workgroup's samples
  if (mode == 2) {  // zero the rows this workgroup owns: its GpNorm channel block and its sample rows
    for (int c = 0; c < 3; ++c)
      if ((int)blockIdx.y == c % (int)gridDim.y)
        for (size_t q = threadIdx.x; q < (size_t)S * n; q += blockDim.x) J[(size_t)c * S * n + q] = 0.0;
  }
  __syncthreads();
  for (int c = 0; c < 3; ++c) gpnorm_rows(w, c, x + (size_t)c * S, res + (size_t)c * S, J + (size_t)c * S * n, n, c * S, mode == 2);
  // cost_functions.h:201-253.  The six table-row dot products of a sample are taken by a whole wave (coalesced rows) into LDS,
  // then one lane per sample does the SO(3) part.
  __shared__ double sdot[256][6];
  for (int ib = i_lo; ib < i_hi; ib += 256) {
    const int cnt = min(256, i_hi - ib);
    for (int q = (int)(threadIdx.x >> 6); q < cnt; q += (int)(blockDim.x >> 6)) {
      const int i = ib + q;
      double d[6];
      wave_row_dots6(w.KsKinv + (size_t)i * S, (size_t)G * S, x, w.KsIntKinv + (size_t)i * S, (size_t)G * S, x, S, d);
      if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int q2 = 0; q2 < 6; ++q2) sdot[q][q2] = d[q2];
      }
    }
    __syncthreads();
    for (int q = threadIdx.x; q < cnt; q += blockDim.x) {
      const int i = ib + q;
      const double drv[3] = {sdot[q][0], sdot[q][1], sdot[q][2]}, rot[3] = {sdot[q][3], sdot[q][4], sdot[q][5]};
      const double dtm = w.gyr_t[i] - w.start_t;
      const V3 rv = v3(rot[0] + dtm * w.hyper[3], rot[1] + dtm * w.hyper[7], rot[2] + dtm * w.hyper[11]);
      const V3 dv = v3(drv[0] + w.hyper[3], drv[1] + w.hyper[7], drv[2] + w.hyper[11]);
      const V3 t = mvec(Jr(rv), dv);
      double* r = res + 3 * S + 3 * i;
      r[0] = t.x - (w.gyr[i] - w.gyr_bias[0]);  // un-weighted on purpose (cost_functions.h:250)
      r[1] = t.y - (w.gyr[G + i] - w.gyr_bias[1]);
      r[2] = t.z - (w.gyr[2 * G + i] - w.gyr_bias[2]);
      if (mode != 0) {
        double D[3][6];
        jacobian_res(rv, dv, D);
        double* st = w.sample_tmp + (size_t)i * 24;
        for (int a = 0; a < 3; ++a)
          for (int k = 0; k < 6; ++k) st[a * 6 + k] = D[a][k];
      }
    }
    __syncthreads();
  }
  if (mode == 0) return;
  __syncthreads();
  {  // cost_functions.h:229-246: one wave per (sample, axis) row, lanes along the S columns of a channel (coalesced, no index division)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwave = (int)(blockDim.x >> 6);
    for (int rr = wave; rr < (i_hi - i_lo) * 3; rr += nwave) {
      const int i = i_lo + rr / 3, a = rr % 3;
      const double* st = w.sample_tmp + (size_t)i * 24;
      double* row = J + (size_t)(3 * S + 3 * i + a) * n;
      for (int c = 0; c < 3; ++c) {
        const double f0 = st[a * 6 + c], f1 = st[a * 6 + c + 3];
        const double* k0 = w.KsIntKinv + ((size_t)c * G + i) * S;
        const double* k1 = w.KsKinv + ((size_t)c * G + i) * S;
        for (int j = lane; j < S; j += 64) row[c * S + j] = f0 * k0[j] + f1 * k1[j];
      }
    }
  }
}

// rotation vector / R^T at a velocity stamp from the (now constant) rotation states (cost_functions.h:333-353)
__device__ __forceinline__ V3 vel_rot_vec(const UgpmWin& w, int i) {
  const int S = w.S, V = w.V;
  double rot[3];
  for (int c = 0; c < 3; ++c) rot[c] = row_dot(w.KgyrIntKinv + ((size_t)c * V + i) * S, w.s_dr + (size_t)c * S, S);
  const double dtm = w.vel_t[i] - w.start_t;
  return v3(rot[0] + dtm * w.hyper[3], rot[1] + dtm * w.hyper[7], rot[2] + dtm * w.hyper[11]);
}

// Problem #2 (preint.h:954-967): rotation states constant; x = [s_vel0 | s_vel1 | s_vel2]; rows = VelCost (3V) then 3 GpNorm (3S).
__global__ __launch_bounds__(256) void vel_eval_kernel(const UgpmWin* __restrict__ wins, int mode) {
  const UgpmWin w = load_win(wins, blockIdx.x);
  if (*w.status != 0 || w.lmi[1]) return;
  if (mode == 1 && !w.lmi[3]) return;
  if (mode == 0 && blockIdx.y == 0 && threadIdx.x == 0) lm_step_bookkeeping(w);
  const int S = w.S, V = w.V, n = 3 * S;
  const double* x = mode == 0 ? w.lmv + 6 * (size_t)n : w.lmv + 5 * (size_t)n;
  double* res = mode == 0 ? w.res_new : w.res;
  double* J = w.Jvel;
  const int i_lo = (int)(((long)V * blockIdx.y) / gridDim.y), i_hi = (int)(((long)V * (blockIdx.y + 1)) / gridDim.y);
  if (mode == 2) {
    for (int c = 0; c < 3; ++c)
      if ((int)blockIdx.y == (3 + c) % (int)gridDim.y)
        for (size_t q = threadIdx.x; q < (size_t)S * n; q += blockDim.x) J[((size_t)3 * V + (size_t)c * S) * n + q] = 0.0;
    for (size_t q = threadIdx.x; q < (size_t)(i_hi - i_lo) * 3 * n; q += blockDim.x) J[(size_t)i_lo * 3 * n + q] = 0.0;
  }
  __syncthreads();
  for (int c = 0; c < 3; ++c) gpnorm_rows(w, 3 + c, x + (size_t)c * S, res + 3 * V + (size_t)c * S, J + ((size_t)3 * V + (size_t)c * S) * n, n, c * S, mode == 2);
  const double wgt = sqrt(1.0 / w.vel_var);
  // cost_functions.h:323-381; dot products by whole waves as in rot_eval_kernel
  __shared__ double sdot[256][6];
  for (int ib = i_lo; ib < i_hi; ib += 256) {
    const int cnt = min(256, i_hi - ib);
    for (int q = (int)(threadIdx.x >> 6); q < cnt; q += (int)(blockDim.x >> 6)) {
      const int i = ib + q;
      // 0..2 vel_rot_vec, 3..5 velocity.  The rotation states are constants of this problem, so R(t)^T of every sample is computed
      // once (mode 2, kept in sample_tmp) and candidate evaluations (mode 0) only need the three velocity dot products.
      if (mode == 0) {
        double d3[3];
        wave_row_dots3(w.KvelKinv + (size_t)i * S, (size_t)V * S, x, S, d3);
        if ((threadIdx.x & 63) == 0) {
#pragma unroll
          for (int q2 = 0; q2 < 3; ++q2) sdot[q][3 + q2] = d3[q2];
        }
      } else {
        double d[6];
        wave_row_dots6(w.KgyrIntKinv + (size_t)i * S, (size_t)V * S, w.s_dr, w.KvelKinv + (size_t)i * S, (size_t)V * S, x, S, d);
        if ((threadIdx.x & 63) == 0) {
#pragma unroll
          for (int q2 = 0; q2 < 6; ++q2) sdot[q][q2] = d[q2];
        }
      }
    }
    __syncthreads();
    for (int q = threadIdx.x; q < cnt; q += blockDim.x) {
      const int i = ib + q;
      M3 RT;
      if (mode == 0) {
        RT = loadM(w.sample_tmp + (size_t)i * 24);
      } else {
        const double dtm = w.vel_t[i] - w.start_t;
        const V3 rv = v3(sdot[q][0] + dtm * w.hyper[3], sdot[q][1] + dtm * w.hyper[7], sdot[q][2] + dtm * w.hyper[11]);
        RT = expMap(v3(-rv.x, -rv.y, -rv.z));
      }
      const V3 vv = v3(sdot[q][3] + w.hyper[15], sdot[q][4] + w.hyper[19], sdot[q][5] + w.hyper[23]);
      const V3 t = mvec(RT, vv);
      double* r = res + 3 * i;
      r[0] = (t.x - (w.vel[i] - w.vel_bias[0])) * wgt;
      r[1] = (t.y - (w.vel[V + i] - w.vel_bias[1])) * wgt;
      r[2] = (t.z - (w.vel[2 * V + i] - w.vel_bias[2])) * wgt;
      if (mode != 0) storeM(w.sample_tmp + (size_t)i * 24, RT);
    }
    __syncthreads();
  }
  if (mode == 0) return;
  __syncthreads();
  {  // cost_functions.h:372-376, one wave per (sample, axis) row
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwave = (int)(blockDim.x >> 6);
    for (int rr = wave; rr < (i_hi - i_lo) * 3; rr += nwave) {
      const int i = i_lo + rr / 3, a = rr % 3;
      double* row = J + (size_t)(3 * i + a) * n;
      for (int c = 0; c < 3; ++c) {
        const double f = wgt * w.sample_tmp[(size_t)i * 24 + a * 3 + c];
        const double* k0 = w.KvelKinv + ((size_t)c * V + i) * S;
        for (int j = lane; j < S; j += 64) row[c * S + j] = f * k0[j];
      }
    }
  }
}

// Stacked Jacobian of the correlation step at the LPM-initialised state (preint.h:887-937): rows 3G (RotCost) + 3V (VelCost),
// columns 6S = [rot channels | velocity channels].  grid: (kCorrJacParts, windows), block 256: workgroup p owns every
// kCorrJacParts-th gyro sample and velocity sample -- first the 3 x 6 / 3 x 3 factors of its samples (one lane per sample), then
// the rows of those samples, one wave per (sample, axis) row with the lanes running along the S columns of a channel (coalesced
// 512-byte stores, no index arithmetic beyond adds).
constexpr int kCorrJacParts = 16;
__global__ __launch_bounds__(256) void corr_jac_kernel(const UgpmWin* __restrict__ wins) {
  const WinPart wp = xcd_win_part();
  const UgpmWin w = load_win(wins, wp.win);
  if (*w.status != 0 || !w.correlate) return;
  const int S = w.S, G = w.G, V = w.V, n = 6 * S;
  const int part = wp.part, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double* J = w.Jc;
  const double wgt = sqrt(1.0 / w.vel_var);
  // ---- gyro samples i = part, part + kCorrJacParts, ...: table dot products by whole waves (wave_row_dots6), then one lane per sample
  __shared__ double sdot[256][6];
  {
    const int nmine = (G - part + kCorrJacParts - 1) / kCorrJacParts;
    for (int qb = 0; qb < nmine; qb += 256) {
      const int cnt = min(256, nmine - qb);
      for (int q = wave; q < cnt; q += 4) {
        const int i = part + kCorrJacParts * (qb + q);
        double d[6];
        wave_row_dots6(w.KsKinv + (size_t)i * S, (size_t)G * S, w.s_dr, w.KsIntKinv + (size_t)i * S, (size_t)G * S, w.s_dr, S, d);
        if (lane == 0) {
#pragma unroll
          for (int q2 = 0; q2 < 6; ++q2) sdot[q][q2] = d[q2];
        }
      }
      __syncthreads();
      for (int q = threadIdx.x; q < cnt; q += blockDim.x) {
        const int i = part + kCorrJacParts * (qb + q);
        const double dtm = w.gyr_t[i] - w.start_t;
        double D[3][6];
        jacobian_res(v3(sdot[q][3] + dtm * w.hyper[3], sdot[q][4] + dtm * w.hyper[7], sdot[q][5] + dtm * w.hyper[11]),
                     v3(sdot[q][0] + w.hyper[3], sdot[q][1] + w.hyper[7], sdot[q][2] + w.hyper[11]), D);
        double* st = w.sample_tmp_c + (size_t)i * 24;
        for (int a = 0; a < 3; ++a)
          for (int k = 0; k < 6; ++k) st[a * 6 + k] = D[a][k];
      }
      __syncthreads();
    }
  }
  __syncthreads();
  {
    const int nmine = (G - part + kCorrJacParts - 1) / kCorrJacParts;  // samples of this part
    for (int rr = wave; rr < nmine * 3; rr += 4) {                      // row = (sample, axis a)
      const int i = part + kCorrJacParts * (rr / 3), a = rr % 3;
      const double* st = w.sample_tmp_c + (size_t)i * 24;
      double* row = J + (size_t)(3 * i + a) * n;
      for (int c = 0; c < 3; ++c) {
        const double f0 = st[a * 6 + c], f1 = st[a * 6 + c + 3];
        const double* k0 = w.KsIntKinv + ((size_t)c * G + i) * S;
        const double* k1 = w.KsKinv + ((size_t)c * G + i) * S;
        for (int j = lane; j < S; j += 64) row[c * S + j] = f0 * k0[j] + f1 * k1[j];
      }
      for (int j = lane; j < 3 * S; j += 64) row[3 * S + j] = 0.0;  // a rotation residual does not see the velocity channels
    }
  }
  __syncthreads();
  // ---- velocity samples (cost_functions.h:350-377 with all six blocks); sample_tmp is reused per sample index, and the gyro rows
  // of THIS workgroup that read it are complete (barrier above); other workgroups touch other sample indices only when
  // G and V samples with the same index belong to the same part, which they do (same i -> same part)
  {
    const int nmine = (V - part + kCorrJacParts - 1) / kCorrJacParts;
    for (int qb = 0; qb < nmine; qb += 256) {
      const int cnt = min(256, nmine - qb);
      for (int q = wave; q < cnt; q += 4) {
        const int i = part + kCorrJacParts * (qb + q);
        double d[6];  // 0..2 vel_rot_vec, 3..5 velocity
        wave_row_dots6(w.KgyrIntKinv + (size_t)i * S, (size_t)V * S, w.s_dr, w.KvelKinv + (size_t)i * S, (size_t)V * S, w.s_vel, S, d);
        if (lane == 0) {
#pragma unroll
          for (int q2 = 0; q2 < 6; ++q2) sdot[q][q2] = d[q2];
        }
      }
      __syncthreads();
      for (int q = threadIdx.x; q < cnt; q += blockDim.x) {
        const int i = part + kCorrJacParts * (qb + q);
        const double dtm = w.vel_t[i] - w.start_t;
        const V3 rv = v3(sdot[q][0] + dtm * w.hyper[3], sdot[q][1] + dtm * w.hyper[7], sdot[q][2] + dtm * w.hyper[11]);
        const M3 RT = expMap(v3(-rv.x, -rv.y, -rv.z));
        const V3 t = mvec(RT, v3(sdot[q][3] + w.hyper[15], sdot[q][4] + w.hyper[19], sdot[q][5] + w.hyper[23]));
        const M3 dres = mmul(skew(t), Jr(rv));
        double* st = w.sample_tmp_c + (size_t)i * 24;
        storeM(st, dres);
        storeM(st + 9, RT);
      }
      __syncthreads();
    }
  }
  __syncthreads();
  {
    const int nmine = (V - part + kCorrJacParts - 1) / kCorrJacParts;
    for (int rr = wave; rr < nmine * 3; rr += 4) {
      const int i = part + kCorrJacParts * (rr / 3), a = rr % 3;
      const double* st = w.sample_tmp_c + (size_t)i * 24;
      double* row = J + (size_t)(3 * G + 3 * i + a) * n;
      for (int c = 0; c < 3; ++c) {
        const double f0 = wgt * st[a * 3 + c], f1 = wgt * st[9 + a * 3 + c];
        const double* k0 = w.KgyrIntKinv + ((size_t)c * V + i) * S;
        const double* k1 = w.KvelKinv + ((size_t)c * V + i) * S;
        for (int j = lane; j < S; j += 64) {
          row[c * S + j] = f0 * k0[j];
          row[(3 + c) * S + j] = f1 * k1[j];
        }
      }
    }
  }
}

// =============================================================================================== J^T J

// C = A^T A (and g = A^T r) for A (m x n, row-major) on the fp64 matrix cores.
// v_mfma_f64_16x16x4_f64 takes A^T as its 16 x 4 operand and A as its 4 x 16 operand; BOTH are row segments of A (lane l supplies
// A[k0 + (l >> 4)][c0 + (l & 15)]), so no transpose is ever formed.  Result layout: col = lane & 15, row = (lane >> 4) + 4 reg.
// Work split: the lower-triangular 16 x 16 tiling of C is cut into groups of TPG tiles (TPG / 8 accumulators per wave, 8 waves); one
// workgroup owns a group over ALL rows of A.  It streams the rows through LDS in chunks of 16 (8 for n > 496) -- every element of
// A is read from L2 once per group instead of once per output tile -- and writes C symmetrically.  (Splitting the rows over several
// workgroups with partial tiles, an arrival counter and an ordered reduction was measured in round 1 and dropped.)
// grid: 1-D, ceil(windows / 8) * groups * 8 workgroups (window and group from the workgroup number, see the kernel).
// which: 0 rot problem, 1 vel problem, 2 correlation.  Dynamic LDS: 2 * KC * (npad + 1) doubles.  n <= 512 when g is formed.
// decide (LM problems): take the step acceptance test first, see below.
__device__ __forceinline__ int ata_npad(int n) { return ((n + 15) / 32) * 32 + 16; }  // >= round_up(n, 16); rows 32 banks apart

// CMAX = 64-column groups of a staged row (npad <= 64 CMAX), KC = rows per staged chunk, TPG = output tiles per workgroup (8 waves x
// TPG / 8 accumulators): 32 for the LM problems (n = 3S: more, shorter workgroups), 48 for the correlation (n = 6S).
template <int CMAX, int KC, int TPG>
__global__ __launch_bounds__(512) void ata_kernel(const UgpmWin* __restrict__ wins, int which, int n_windows, int groups_max, int decide) {
  // Consecutive workgroups of a launch go round-robin to the 8 XCDs, each with an L2 of its own: ALL tile groups of a window are given
  // workgroup numbers of one residue mod 8, 8 apart, so they start together on ONE XCD and stream the window's J through that L2 side
  // by side (one fetch from HBM / MALL serves the groups) instead of once per XCD.
  const int L = blockIdx.x;
  const int xcd = L & 7, slot = L >> 3;
#ifdef GORIO_ATA_NO_XCD
  const int win = (slot * 8 + xcd) / groups_max, grp = (slot * 8 + xcd) % groups_max;
#else
  const int win = (slot / groups_max) * 8 + xcd, grp = slot % groups_max;
#endif
  if (win >= n_windows) return;
  const UgpmWin w = load_win(wins, win);
  if (*w.status != 0) return;
  int m, n;
  const double* A;
  double* C;
  const double* r = nullptr;
  double* g = nullptr;
  if (which == 2) {
    if (!w.correlate) return;
    m = 3 * w.G + 3 * w.V; n = 6 * w.S; A = w.Jc; C = w.Ac;
  } else {
    if (w.lmi[1]) return;
    n = 3 * w.S;
    m = which == 0 ? 3 * w.S + 3 * w.G : 3 * w.V + 3 * w.S;
    A = which == 0 ? w.Jrot : w.Jvel;
    C = w.JtJ; r = w.res; g = w.lmv;
    if (decide) {
      // The Jacobian in A was written at the CANDIDATE point (rot_eval_kernel mode 3).  Every workgroup of the window takes the
      // acceptance test for itself (lm_decide_block; group 0 stores the new solver state) and only an accepted step is linearised:
      // J^T J and g keep the values of the current point otherwise.  The residual of the new point is the candidate residual.
      __shared__ double sred[8];
      if (lm_decide_block(w, m, n, grp == 0, sred) != 1) return;
      r = w.res_new;
      if (grp == 0)
        for (int k = threadIdx.x; k < m; k += blockDim.x) w.res[k] = r[k];
    } else if (!w.lmi[3]) {
      return;
    }
  }
  const int T = (n + 15) / 16, ntile = T * (T + 1) / 2;
  const int ng = (ntile + TPG - 1) / TPG;
  if (grp >= ng) return;
  const int q_lo = (int)((long)grp * ntile / ng), q_hi = (int)((long)(grp + 1) * ntile / ng);
  const int npad = ata_npad(n);
  const int nck = (m + KC - 1) / KC;
  const int c_lo = 0, c_hi = nck;
  extern __shared__ double ata_lds[];  // [2][KC][npad] staged rows of A, then [2][KC] staged entries of r
  double* rl = ata_lds + (size_t)2 * KC * npad;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lk = lane >> 4;
  constexpr int NT = TPG / 8;
  // this wave's tiles: q = q_lo + wave + 8 t.  Slots past the end of the group repeat the group's last tile (computed, not stored):
  // the loop below then has no branches and the compiler can keep all operand reads of a k-step in flight ahead of the MFMAs.
  int trofs[NT], tcofs[NT], qv[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int q = min(q_lo + wave + 8 * t, q_hi - 1);
    int tr = (int)((sqrtf(8.0f * (float)q + 1.0f) - 1.0f) * 0.5f);
    while (tr * (tr + 1) / 2 > q) --tr;
    while ((tr + 1) * (tr + 2) / 2 <= q) ++tr;
    qv[t] = __builtin_amdgcn_readfirstlane(q);
    trofs[t] = __builtin_amdgcn_readfirstlane(tr) * 16;
    tcofs[t] = __builtin_amdgcn_readfirstlane(q - tr * (tr + 1) / 2) * 16;
  }
  const int nt = max(0, min(NT, (q_hi - q_lo - wave + 7) / 8));
  f64x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f64x4{0.0, 0.0, 0.0, 0.0};
  double gacc = 0.0;
  // g = A^T r rides along.  One thread per column in one workgroup made a 16-step dependent LDS chain per chunk the longest phase
  // of that workgroup (a quarter of its time); instead every group owns a slice of the columns and all 512 threads share it: thread
  // (gpart, column) takes every gparts-th row of a chunk, the partial sums meet in LDS at the end in a fixed order.
  const int gj0 = (int)((long)grp * n / ng), gj1 = (int)((long)(grp + 1) * n / ng);
  const int gcolw = ((gj1 - gj0 + 63) / 64) * 64;
  const int gparts = gcolw <= 64 ? 8 : (gcolw <= 128 ? 4 : (gcolw <= 256 ? 2 : 1));
  const int gpart = (int)threadIdx.x / gcolw, gj = gj0 + (int)threadIdx.x % gcolw;
  const bool gcol = g != nullptr && gpart < gparts && gj < gj1;
  // chunk loader: wave v brings rows v, v + 8 (KC = 16) of the chunk, 64 columns per instruction
  constexpr int RMAX = KC / 8;
  double pre[RMAX][CMAX];
  double rpre = 0.0;
  auto fetch = [&](int ck) {
#pragma unroll
    for (int rr = 0; rr < RMAX; ++rr) {
      const int k = ck * KC + wave + 8 * rr;
#pragma unroll
      for (int cc = 0; cc < CMAX; ++cc) {
        const int col = lane + 64 * cc;
        pre[rr][cc] = (k < m && col < n) ? A[(size_t)k * n + col] : 0.0;
      }
    }
    if (r != nullptr && tid < KC) rpre = (ck * KC + tid < m) ? r[ck * KC + tid] : 0.0;
  };
  auto stash = [&](int buf) {
    double* dst = ata_lds + (size_t)buf * KC * npad;
#pragma unroll
    for (int rr = 0; rr < RMAX; ++rr) {
      const int row = wave + 8 * rr;
#pragma unroll
      for (int cc = 0; cc < CMAX; ++cc) {
        const int col = lane + 64 * cc;
        if (col < npad) dst[row * npad + col] = pre[rr][cc];
      }
    }
    if (tid < KC) rl[buf * KC + tid] = rpre;
  };
  CHOL_T_INIT;
  if (c_lo < c_hi) {
    fetch(c_lo);
    stash(0);
  }
  __syncthreads();
  CHOL_T(0);
  for (int ck = c_lo; ck < c_hi; ++ck) {
    const int buf = (ck - c_lo) & 1;
    if (ck + 1 < c_hi) fetch(ck + 1);
    const double* src = ata_lds + (size_t)buf * KC * npad;
#pragma unroll
    for (int kk = 0; kk < KC; kk += 4) {
      const double* rowp = src + (kk + lk) * npad + lr;
      double av[NT], bv[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        av[t] = rowp[trofs[t]];
        bv[t] = rowp[tcofs[t]];
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[t], bv[t], acc[t], 0, 0, 0);
    }
    CHOL_T(1);
    if (gcol) {  // this thread's share of g = A^T r: column gj, rows gpart, gpart + gparts, ... of the chunk
      for (int kk = gpart; kk < KC; kk += gparts) gacc += src[kk * npad + gj] * rl[buf * KC + kk];
    }
    CHOL_T(2);
    if (ck + 1 < c_hi) stash(buf ^ 1);
    CHOL_T(3);
    __syncthreads();
    CHOL_T(4);
  }
  {  // this workgroup saw every row: its accumulators ARE the result
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (t < nt) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int i = trofs[t] + lk + 4 * rg, j = tcofs[t] + lr;
          if (i < n && j < n) {
            C[(size_t)i * n + j] = acc[t][rg];
            C[(size_t)j * n + i] = acc[t][rg];
          }
        }
      }
    }
    if (g != nullptr) {  // uniform per workgroup.  The staged chunks are dead: their LDS holds the partial sums [gparts][gcolw]
      __syncthreads();
      if (gpart < gparts) ata_lds[gpart * gcolw + (int)threadIdx.x % gcolw] = gcol ? gacc : 0.0;
      __syncthreads();
      if (tid < gj1 - gj0) {
        double v = 0.0;
        for (int q = 0; q < gparts; ++q) v += ata_lds[q * gcolw + tid];
        g[gj0 + tid] = v;
      }
    }
  }
  CHOL_T(7);
}

// Problem #2 is LINEAR in its unknowns (the rotation states are constants there, so VelCost's Jacobian R(t)^T K_vel K^-1 and
// the GpNorm blocks do not depend on x): after an accepted step the Jacobian and J^T J of the first evaluation are still exact,
// the residual at the new point is the candidate residual already computed, and only the gradient g = J^T r changes.  This
// kernel replaces the re-linearisation (vel_eval mode 1 + ata) for that problem with exactly those values.
// grid: (windows, ceil(3 max_S / 64)), block 256 = 4 row slices x 64 columns.
__global__ __launch_bounds__(256) void lm_relinearize_linear_kernel(const UgpmWin* __restrict__ wins) {
  const UgpmWin w = load_win(wins, blockIdx.x);
  if (*w.status != 0 || w.lmi[1]) return;
  const int n = 3 * w.S, m = 3 * w.V + 3 * w.S;
  const int j0 = blockIdx.y * 64;
  if (j0 >= n) return;
  __shared__ double sred[8];
  if (lm_decide_block(w, m, n, blockIdx.y == 0, sred) != 1) return;  // step acceptance, fused (see lm_decide_block)
  const double* J = w.Jvel;
  const double* rn = w.res_new;
  __shared__ double sg[4][64];
  if (blockIdx.y == 0)
    for (int k = threadIdx.x; k < m; k += blockDim.x) w.res[k] = rn[k];
  const int slice = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int j = j0 + lane;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  if (j < n) {
    int k = slice;
    for (; k + 12 < m; k += 16) {  // four independent chains keep four loads in flight
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[u] += J[(size_t)(k + 4 * u) * n + j] * rn[k + 4 * u];
    }
    for (; k < m; k += 4) acc[0] += J[(size_t)k * n + j] * rn[k];
  }
  sg[slice][lane] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  __syncthreads();
  if (slice == 0 && j < n) w.lmv[j] = (sg[0][lane] + sg[1][lane]) + (sg[2][lane] + sg[3][lane]);
}

// =============================================================================================== Levenberg-Marquardt (Ceres 2.1 defaults + preint.h:943-948)
// lmv layout (vectors of n = 3S): 0 g = J^T r (unscaled), 1 scale, 2 diag, 3 step (scaled), 4 delta (unscaled), 5 x, 6 x_new, 7 rhs
// lmc: 0 cost, 1 cost_new, 2 radius, 3 decrease_factor, 4 x_norm, 5 model_cost_change, 6 step_norm, 7 initial cost
// lmi: 0 iter, 1 done, 2 reuse_diag, 3 need_J, 4 step_valid, 5 termination, 6 successful, 7 problem, 8 first

// start a problem: load x, reset the control block.  grid: (windows), block 256.
__global__ __launch_bounds__(256) void lm_begin_kernel(const UgpmWin* __restrict__ wins, int problem) {
  const UgpmWin w = load_win(wins, blockIdx.x);
  if (*w.status != 0) return;
  const int n = 3 * w.S;
  const double* src = problem == 0 ? w.s_dr : w.s_vel;
  for (int j = threadIdx.x; j < n; j += blockDim.x) w.lmv[5 * (size_t)n + j] = src[j];
  if (threadIdx.x == 0) {
    for (int q = 0; q < 16; ++q) w.lmi[q] = 0;
    w.lmi[3] = 1;  // need_J
    w.lmi[7] = problem;
    w.lmi[8] = 1;  // first evaluation: scaling and the gradient check of iteration zero are pending
    w.lmc[2] = 1e4;
    w.lmc[3] = 2.0;
  }
}

// One trust-region step: (on fresh J^T J) cost / gradient test / Jacobi scaling, then solve (D J^T J D + diag / radius) y = D g,
// step = -y, delta = D step, model cost change, candidate x_new.  grid: (windows, blocks), block 512.
// blocks = 1: the whole 3S x 3S system (problem #1).  blocks = kVelBlocks (problem #2): VelCost's Jacobian is R(t)^T applied to
// per-channel rows (cost_functions.h:375) and R R^T = I, the GpNorm blocks are per channel, so J^T J is block diagonal with one
// S x S block per velocity channel (its off-diagonal blocks are rounding noise, 1e-16 of the diagonal ones): workgroup b factors
// and solves block b alone -- a 66 x 66 factorisation instead of 198 x 198 on the critical path of every velocity iteration.
// Every workgroup evaluates the (cheap, deterministic) global quantities for itself -- cost, gradient maximum, |x| -- and writes
// identical values; the words a step consumes or counts are updated by the evaluation kernel that follows (lm_step_bookkeeping), so no
// workgroup reads a control word another one writes in the same launch.
__global__ __launch_bounds__(512) void lm_step_kernel(const UgpmWin* __restrict__ wins) {
  const UgpmWin w = load_win(wins, blockIdx.x);
  if (*w.status != 0 || w.lmi[1]) return;
  const int n = 3 * w.S;
  const int problem = w.lmi[7];
  const int m = problem == 0 ? 3 * w.S + 3 * w.G : 3 * w.V + 3 * w.S;
  const int nblk = (int)gridDim.y, blk = (int)blockIdx.y;
  const int j0 = (int)((long)n * blk / nblk), j1 = (int)((long)n * (blk + 1) / nblk), ns = j1 - j0;  // this workgroup's unknowns
  double* g = w.lmv;
  double* scale = w.lmv + (size_t)n;
  double* diag = w.lmv + 2 * (size_t)n;
  double* step = w.lmv + 3 * (size_t)n;
  double* delta = w.lmv + 4 * (size_t)n;
  double* x = w.lmv + 5 * (size_t)n;
  double* xn = w.lmv + 6 * (size_t)n;
  __shared__ double sred[8];
  __shared__ int sflag;
  __shared__ CholLds chol;
  if (w.lmi[3]) {  // a fresh linearisation arrived
    double c = 0.0;
    for (int k = threadIdx.x; k < m; k += blockDim.x) c += w.res[k] * w.res[k];
    c = 0.5 * block_sum(c, sred);
    double gm = 0.0, xn2 = 0.0;
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
      gm = fmax(gm, fabs(g[j]));
      xn2 += x[j] * x[j];
    }
    gm = block_max(gm, sred);
    xn2 = block_sum(xn2, sred);
    if (w.lmi[8])
      for (int j = j0 + (int)threadIdx.x; j < j1; j += blockDim.x) scale[j] = 1.0 / (1.0 + sqrt(w.JtJ[(size_t)j * n + j]));  // Jacobi scaling, fixed
    __syncthreads();
    if (threadIdx.x == 0) {  // the same values from every workgroup of the window
      w.lmc[0] = c;
      w.lmc[4] = sqrt(xn2);
      if (w.lmi[8]) w.lmc[7] = c;
      if (gm <= 1e-10) {  // gradient_tolerance
        w.lmi[1] = 1;
        w.lmi[5] = 3;
      }
    }
    if (gm <= 1e-10) return;
  }
  if (w.lmi[0] >= 50) {  // max_num_iterations, preint.h:945
    if (threadIdx.x == 0) { w.lmi[1] = 1; w.lmi[5] = 4; }
    return;
  }
  const double radius = w.lmc[2];
  if (radius < 1e-32) {
    if (threadIdx.x == 0) { w.lmi[1] = 1; w.lmi[5] = 5; }
    return;
  }
  if (!w.lmi[2])
    for (int j = j0 + (int)threadIdx.x; j < j1; j += blockDim.x) diag[j] = fmin(fmax(w.JtJ[(size_t)j * n + j] * scale[j] * scale[j], 1e-6), 1e32);
  __syncthreads();
  double* L = w.lhs + (size_t)j0 * n + j0;  // this workgroup's diagonal block inside the n x n buffer (row stride n)
  __shared__ double xs[kCholMaxN / 2];
  const bool fused = ns < 384 && ns <= (int)blockDim.x && ns <= kCholMaxN / 2;  // rhs rides through the factorisation, blocked back-substitution
  for (int i = threadIdx.x >> 6; i < ns; i += (int)(blockDim.x >> 6)) {  // lower triangle of D J^T J D + diag / radius, one row per wave
    const double si = scale[j0 + i];
    for (int j = threadIdx.x & 63; j <= i; j += 64)
      L[(size_t)i * n + j] = w.JtJ[(size_t)(j0 + i) * n + j0 + j] * si * scale[j0 + j] + (i == j ? diag[j0 + i] / radius : 0.0);
  }
  double* rhs = w.lmv + 7 * (size_t)n + j0;  // right-hand side row carried through the factorisation (global, like the matrix)
  for (int j = threadIdx.x; j < ns; j += blockDim.x) {
    step[j0 + j] = g[j0 + j] * scale[j0 + j];
    if (fused) rhs[j] = g[j0 + j] * scale[j0 + j];
  }
  __syncthreads();
  bool valid = block_cholesky(L, ns, n, chol, &sflag, fused ? rhs : nullptr);
  if (valid) {
    if (fused) {
      for (int j = threadIdx.x; j < ns; j += blockDim.x) xs[j] = rhs[j];
      __syncthreads();
      block_backward(L, ns, n, xs, chol);
      for (int j = threadIdx.x; j < ns; j += blockDim.x) step[j0 + j] = xs[j];
    } else {
      double* xp = &chol.P[0][0];  // the panel buffer is free again: solve in LDS
      for (int j = threadIdx.x; j < ns; j += blockDim.x) xp[j] = step[j0 + j];
      __syncthreads();
      if (threadIdx.x < 64) {
        wave_forward<1>(L, ns, n, xp, 1, 0);
        wave_backward(L, ns, n, xp);
      }
      __syncthreads();
      for (int j = threadIdx.x; j < ns; j += blockDim.x) step[j0 + j] = xp[j];
    }
    __syncthreads();
    double bad = 0.0;
    for (int j = j0 + (int)threadIdx.x; j < j1; j += blockDim.x) {
      step[j] = -step[j];
      if (!isfinite(step[j])) bad = 1.0;
      delta[j] = step[j] * scale[j];
    }
    bad = block_max(bad, sred);
    valid = bad == 0.0;
  }
  double mcc = 0.0, sn2 = 0.0;
  if (valid) {  // this block's share of the model cost change -(J d)^T (r + J d / 2) = -(d.g + d^T (J^T J) d / 2)
    double acc = 0.0;
    for (int i = j0 + (int)(threadIdx.x >> 6); i < j1; i += (int)(blockDim.x >> 6)) {  // 0.5 d^T (J^T J) d, one matrix row per wave (coalesced)
      const double hi = 0.5 * delta[i];
      for (int j = j0 + (int)(threadIdx.x & 63); j < j1; j += 64) acc += hi * w.JtJ[(size_t)i * n + j] * delta[j];
    }
    for (int i = j0 + (int)threadIdx.x; i < j1; i += blockDim.x) {
      acc += delta[i] * g[i];
      sn2 += delta[i] * delta[i];
      xn[i] = x[i] + delta[i];
    }
    mcc = -block_sum(acc, sred);
    sn2 = block_sum(sn2, sred);
  }
  if (threadIdx.x == 0) {  // lm_decide_block adds the blocks up (the step is invalid unless every block is valid and the sum positive)
    w.lmi[9 + blk] = valid ? 1 : 0;
    w.lmc[8 + blk] = mcc;
    w.lmc[11 + blk] = sn2;
  }
}


// write the solution back into the state.  grid: (windows), block 256.
__global__ __launch_bounds__(256) void lm_end_kernel(const UgpmWin* __restrict__ wins, int problem, double* __restrict__ diag_out /* [windows][4] */) {
  const UgpmWin w = load_win(wins, blockIdx.x);
  if (*w.status != 0) return;
  const int n = 3 * w.S;
  double* dst = problem == 0 ? w.s_dr : w.s_vel;
  for (int j = threadIdx.x; j < n; j += blockDim.x) dst[j] = w.lmv[5 * (size_t)n + j];
  if (threadIdx.x == 0) {
    diag_out[blockIdx.x * 4 + problem * 2 + 0] = (double)w.lmi[0];
    diag_out[blockIdx.x * 4 + problem * 2 + 1] = w.lmc[0];
  }
}

// =============================================================================================== state correlation

// A = J^T J + 1e-5 I = L L^T, L^-1, dsc = state_std / sqrt(diag(A^-1)) (preint.h:1478-1492).  grid: (windows), block 1024.
__global__ __launch_bounds__(512) void corr_factor_kernel(const UgpmWin* __restrict__ wins) {
  const UgpmWin w = load_win(wins, blockIdx.x);
  if (*w.status != 0 || !w.correlate) return;
  const int n = 6 * w.S;
  __shared__ int sflag;
  __shared__ CholLds chol;
  for (int i = threadIdx.x; i < n; i += blockDim.x) w.Ac[(size_t)i * n + i] += 0.00001;
  __syncthreads();
  if (!block_cholesky(w.Ac, n, n, chol, &sflag)) {
    if (threadIdx.x == 0) *w.status = -6;
    return;
  }
}

// diag(A^-1) = squared column norms of L^-1, 16 columns per workgroup; then dsc = state_std / sqrt(diag(A^-1)) (preint.h:1487-1489).
// grid: (ceil(6S / 16), windows), block 256.  Dynamic LDS: (rows + 16) * 17 doubles for X, rows = 6 max_S.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void corr_diag_kernel(const UgpmWin* __restrict__ wins) {
  const WinPart wp = xcd_win_part();
  const UgpmWin w = load_win(wins, wp.win);
  if (*w.status != 0 || !w.correlate) return;
  const int n = 6 * w.S;
  const int jb = wp.part, j0 = jb * 16;
  if (j0 >= n) return;
  extern __shared__ double Xc[];
  __shared__ Solve16Lds sh;
  {
    const int rows = ((n + 15) / 16 - jb) * 16;
    for (int q = threadIdx.x; q < rows * 16; q += blockDim.x) Xc[(size_t)(q >> 4) * 17 + (q & 15)] = (q >> 4) == (q & 15) ? 1.0 : 0.0;  // E_j
  }
  blocked_lower_solve16(w.Ac, n, jb, Xc, sh);
  if (threadIdx.x < 16) {
    const int c = threadIdx.x, col = j0 + c;
    const int rows = ((n + 15) / 16 - jb) * 16;
    double ssq = 0.0;
    for (int r = 0; r < rows; ++r) ssq = __builtin_fma(Xc[(size_t)r * 17 + c], Xc[(size_t)r * 17 + c], ssq);
    if (col < n) w.dsc[col] = w.sstd[col] * (1.0 / sqrt(ssq));
  }
}

// =============================================================================================== inference tables + get(t)

// preint.h:978-1060 and finishStateDiff (preint.h:1401-1441).  grid: (windows), block 256.
__global__ __launch_bounds__(256) void finish_kernel(const UgpmWin* __restrict__ wins) {
  const UgpmWin w = load_win(wins, blockIdx.x);
  if (*w.status != 0) return;
  const int S = w.S;
  for (int q = threadIdx.x; q < 6 * S; q += blockDim.x) {  // alpha = K^-1 s
    const int c = q / S, i = q % S;
    const double* s = c < 3 ? w.s_dr + (size_t)c * S : w.s_vel + (size_t)(c - 3) * S;
    w.alpha[q] = row_dot(w.Kinv + ((size_t)c * S + i) * S, s, S);
  }
  for (int q = threadIdx.x; q < 3 * S; q += blockDim.x) {  // state_r = K_int K^-1 s + dt mean (preint.h:1005, 1410)
    const int a = q / S, i = q % S;
    w.state_r[q] = row_dot(w.KintKinv + ((size_t)a * S + i) * S, w.s_dr + (size_t)a * S, S) + (w.state_t[i] - w.start_t) * w.hyper[a * 4 + 3];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < S; i += blockDim.x) {  // finishStateDiff
    const V3 ri = v3(w.state_r[i], w.state_r[S + i], w.state_r[2 * S + i]);
    const M3 Ji = JrInv(ri);
    const V3 d_r = mvec(Ji, load3(w.d_r_dt_local + (size_t)i * 3));
    const V3 temp_r = ri + mvec(Ji, load3(w.delta_r_time + (size_t)i * 3));
    const V3 dd = (1.0 / kDt) * (mvec(JrInv(temp_r), load3(w.d_r_dt_local_shift + (size_t)i * 3)) - d_r);
    w.d_d_r_dt[i] = dd.x; w.d_d_r_dt[S + i] = dd.y; w.d_d_r_dt[2 * S + i] = dd.z;
    for (int ax = 0; ax < 3; ++ax) {
      const V3 trw = ri + mvec(Ji, load3(w.delta_r_bw + ((size_t)ax * S + i) * 3));
      const V3 t = (1.0 / kBw) * (mvec(JrInv(trw), load3(w.d_r_bw_local_shift + ((size_t)ax * S + i) * 3)) - d_r);
      w.d_state_bw[((size_t)0 * S + i) * 3 + ax] = t.x;
      w.d_state_bw[((size_t)1 * S + i) * 3 + ax] = t.y;
      w.d_state_bw[((size_t)2 * S + i) * 3 + ax] = t.z;
    }
  }
  __syncthreads();
  // delta_R_dt_start (preint.h:1024-1031)
  __shared__ double srd[3];
  if (threadIdx.x < 3) {
    const int c = threadIdx.x;
    double s = 0.0;
    for (int j = 0; j < S; ++j) s += se_kint(w.start_t, w.start_t + kDt, w.state_t[j], w.hyper[c * 4], w.hyper[c * 4 + 1]) * w.alpha[c * S + j];
    srd[c] = s + kDt * w.hyper[c * 4 + 3];
  }
  __syncthreads();
  const M3 dRt = mtr(expMap(v3(srd[0], srd[1], srd[2])));
  const V3 mean_vel = v3(w.hyper[15], w.hyper[19], w.hyper[23]);
  for (int i = threadIdx.x; i < S; i += blockDim.x) {  // preint.h:1035-1060
    const V3 ri = v3(w.state_r[i], w.state_r[S + i], w.state_r[2 * S + i]);
    const M3 R = expMap(ri);
    M3 drbw;  // rows = K_int K^-1 d_state_bw (preint.h:1009-1017), 3 x 3 per state
    for (int a = 0; a < 3; ++a)
      for (int c = 0; c < 3; ++c) {
        double s = 0.0;
        const double* row = w.KintKinv + ((size_t)a * S + i) * S;
        for (int j = 0; j < S; ++j) s += row[j] * w.d_state_bw[((size_t)a * S + j) * 3 + c];
        drbw.m[a * 3 + c] = s;
      }
    const V3 sv = v3(w.s_vel[i], w.s_vel[S + i], w.s_vel[2 * S + i]) + mean_vel;
    const M3 dvbw = mmul(mmul(skew(sv), Jr(v3(-ri.x, -ri.y, -ri.z))), drbw);  // negated below (preint.h:1048)
    const V3 dvdt = (1.0 / kDt) * (mvec(dRt, sv) - sv);
    const double dv[3] = {dvdt.x, dvdt.y, dvdt.z};
    for (int a = 0; a < 3; ++a) {
      for (int c = 0; c < 3; ++c) {
        w.d_vel_bv[((size_t)a * S + i) * 3 + c] = R.m[a * 3 + c];
        w.d_vel_bw[((size_t)a * S + i) * 3 + c] = -dvbw.m[a * 3 + c];
      }
      w.d_vel_dt[a * S + i] = dv[a];
    }
  }
}

// Se3Integrator::get(t) (preint.h:1069-1153) + cov inflation of VelPreintegration::get (preint.h:1744-1757).
// grid: (max n_infer, windows), block 256.
__global__ __launch_bounds__(256) void infer_kernel(const UgpmWin* __restrict__ wins) {
  const WinPart wp = xcd_win_part();
  const UgpmWin w = load_win(wins, wp.win);
  const int qi = wp.part;
  if (qi >= w.n_infer) return;
  double* out = w.out + (size_t)qi * 83;
  if (*w.status != 0) {
    for (int k = threadIdx.x; k < 83; k += blockDim.x) out[k] = __longlong_as_double(0x7ff8000000000000LL);
    return;
  }
  const int S = w.S, n = 6 * S;
  const double t = w.infer_t[qi], dt = t - w.start_t;
  __shared__ double ksv[6][160];  // ks = k_int(start, t, state_t) per channel (each value is used S times below: evaluated once)
  __shared__ double ksd[3][160];  // d ks / dt of the velocity channels
  __shared__ double ksK[6][160];  // ks K^-1 per channel
  __shared__ double sval[6][8];   // per channel: 0 ks.alpha, 1 ks K^-1 ks^T, 2 d_r_dt / (ks_dt.alpha + ksK.d_vel_dt), 3..5 d/d bw
  __shared__ double sval2[6][4];  // d/d bv
  __shared__ double sred[8];
  __shared__ double scov[36];
  for (int q = threadIdx.x; q < 6 * S; q += blockDim.x) {
    const int c = q / S, j = q - c * S;
    ksv[c][j] = se_kint(w.start_t, t, w.state_t[j], w.hyper[c * 4], w.hyper[c * 4 + 1]);
    if (c >= 3) ksd[c - 3][j] = se_kint_dt(w.start_t, t, w.state_t[j], w.hyper[c * 4], w.hyper[c * 4 + 1]);
  }
  __syncthreads();
  for (int q = threadIdx.x; q < 6 * S; q += blockDim.x) {
    const int c = q / S, j = q - c * S;
    const double* Kc = w.Kinv + (size_t)c * S * S + j;
    double s = 0.0;
    for (int k = 0; k < S; ++k) s += ksv[c][k] * Kc[(size_t)k * S];
    ksK[c][j] = s;
  }
  __syncthreads();
  {  // per-channel contractions over the S states: one wave per channel (lanes along j), nine sums each
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int c = wv; c < 6; c += (int)(blockDim.x >> 6)) {
      double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // ka, kKk, e2, bw[3], bv[3]
      for (int j = lane; j < S; j += 64) {
        const double ks = ksv[c][j], kk = ksK[c][j];
        acc[0] += ks * w.alpha[c * S + j];
        acc[1] += kk * ks;
        if (c < 3) {
          acc[2] += kk * w.d_d_r_dt[c * S + j];
          for (int q = 0; q < 3; ++q) acc[3 + q] += kk * w.d_state_bw[((size_t)c * S + j) * 3 + q];
        } else {
          acc[2] += ksd[c - 3][j] * w.alpha[c * S + j] + kk * w.d_vel_dt[(c - 3) * S + j];
          for (int q = 0; q < 3; ++q) {
            acc[3 + q] += kk * w.d_vel_bw[((size_t)(c - 3) * S + j) * 3 + q];
            acc[6 + q] += kk * w.d_vel_bv[((size_t)(c - 3) * S + j) * 3 + q];
          }
        }
      }
#pragma unroll
      for (int v = 0; v < 9; ++v)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc[v] += __shfl_down(acc[v], off, 64);
      if (lane == 0) {
        const double l2 = w.hyper[c * 4], sf2 = w.hyper[c * 4 + 1];
        double var = kss_int(w.start_t, t, l2, sf2) - acc[1];
        if (var <= 0) var = dt * dt * w.hyper[c * 4 + 2];
        sval[c][0] = acc[0] + dt * w.hyper[c * 4 + 3];
        sval[c][1] = var;
        sval[c][2] = acc[2];
        for (int q = 0; q < 3; ++q) {
          sval[c][3 + q] = acc[3 + q];
          sval2[c][q] = acc[6 + q];
        }
      }
    }
  }
  __syncthreads();
  // covariance: cov_ab = ks_a C_ab ks_b^T with C = D A^-1 D  =>  (L^-1 u_a) . (L^-1 u_b), u_a = dsc .* ksK_a placed in block a:
  // one blocked forward substitution with the six u_a as right-hand-side columns (blocked_lower_solve16)
  if (w.correlate) {
    extern __shared__ double Xc[];  // [ceil16(n)][17]: columns 0..5 = L^-1 u_a
    __shared__ Solve16Lds sh;
    const int rows = ((n + 15) / 16) * 16;
    for (int q = threadIdx.x; q < rows * 16; q += blockDim.x) Xc[(size_t)(q >> 4) * 17 + (q & 15)] = 0.0;
    __syncthreads();
    for (int row = threadIdx.x; row < n; row += blockDim.x) {
      const int c = row / S;
      Xc[(size_t)row * 17 + c] = w.dsc[row] * ksK[c][row - c * S];
    }
    blocked_lower_solve16(w.Ac, n, 0, Xc, sh);
    for (int pair = 0; pair < 21; ++pair) {
      int a2 = 0, b2 = pair;
      while (b2 >= 6 - a2) { b2 -= 6 - a2; ++a2; }
      b2 += a2;
      double acc = 0.0;
      for (int k = threadIdx.x; k < rows; k += blockDim.x) acc += Xc[(size_t)k * 17 + a2] * Xc[(size_t)k * 17 + b2];
      acc = block_sum(acc, sred);
      if (threadIdx.x == 0) {
        scov[a2 * 6 + b2] = acc;
        scov[b2 * 6 + a2] = acc;
      }
    }
  } else {
    if (threadIdx.x < 36) scov[threadIdx.x] = 0.0;
    __syncthreads();
    if (threadIdx.x < 6) {
      const int a2 = threadIdx.x;
      double s2 = 0.0;
      for (int j = 0; j < S; ++j) s2 += ksK[a2][j] * w.var[a2 * S + j] * ksK[a2][j];
      scov[a2 * 6 + a2] = s2;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const V3 r = v3(sval[0][0], sval[1][0], sval[2][0]);
    const M3 jr = Jr(r);
    const M3 R = expMap(r);
    M3 d_r_dw, d_p_dw, d_p_dv;
    for (int a = 0; a < 3; ++a)
      for (int q = 0; q < 3; ++q) {
        d_r_dw.m[a * 3 + q] = sval[a][3 + q];
        d_p_dw.m[a * 3 + q] = sval[3 + a][3 + q];
        d_p_dv.m[a * 3 + q] = sval2[3 + a][q];
      }
    storeM(out, R);
    out[9] = sval[3][0]; out[10] = sval[4][0]; out[11] = sval[5][0];
    out[12] = dt;
    out[13] = 0.5 * dt * dt;
    double cov[36], td[6];
    for (int a = 0; a < 6; ++a) td[a] = sqrt(sval[a][1]) * (1.0 / sqrt(scov[a * 6 + a]));  // preint.h:1141-1145
    for (int a = 0; a < 6; ++a)
      for (int b = 0; b < 6; ++b) cov[a * 6 + b] = scov[a * 6 + b] * td[a] * td[b];
    M3 c00, c03;  // preint.h:1148-1150
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) {
        c00.m[a * 3 + b] = cov[a * 6 + b];
        c03.m[a * 3 + b] = cov[a * 6 + 3 + b];
      }
    const M3 n00 = mmul(mmul(jr, c00), mtr(jr)), n03 = mmul(jr, c03);
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) {
        cov[a * 6 + b] = n00.m[a * 3 + b];
        cov[a * 6 + 3 + b] = n03.m[a * 3 + b];
        cov[(3 + b) * 6 + a] = n03.m[a * 3 + b];
      }
    if (w.vel_bias_std > 0.0 || w.gyr_bias_std > 0.0) {  // preint.h:1744-1757
      double J[36];
      for (int q = 0; q < 36; ++q) J[q] = 0.0;
      const double g2 = w.gyr_bias_std * w.gyr_bias_std, v2 = w.vel_bias_std * w.vel_bias_std;
      const double bc[6] = {g2, g2, g2, v2, v2, v2};
      for (int a = 0; a < 3; ++a) {
        J[a * 6 + a] = 1.0;
        for (int b = 0; b < 3; ++b) {
          J[(3 + a) * 6 + b] = d_p_dw.m[a * 3 + b];
          J[(3 + a) * 6 + 3 + b] = d_p_dv.m[a * 3 + b];
        }
      }
      for (int a = 0; a < 6; ++a)
        for (int b = 0; b < 6; ++b) {
          double s = 0.0;
          for (int k = 0; k < 6; ++k) s += J[a * 6 + k] * bc[k] * J[b * 6 + k];
          cov[a * 6 + b] += s;
        }
    }
    for (int q = 0; q < 36; ++q) out[14 + q] = cov[q];
    storeM(out + 50, mmul(jr, d_r_dw));
    store3(out + 59, mvec(jr, v3(sval[0][2], sval[1][2], sval[2][2])));
    storeM(out + 62, d_p_dw);
    storeM(out + 71, d_p_dv);
    out[80] = sval[3][2]; out[81] = sval[4][2]; out[82] = sval[5][2];
  }
}

}  // namespace ug
}  // namespace gorio
