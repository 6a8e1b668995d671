// ugpm_chunks.h -- chunked pre-integration (PreintOption::quantum >= 0): the plan that cuts one request into chunk windows and the
// chaining of the chunk results.  Host code: the chunk windows themselves are ordinary windows and run in the SAME device batch as
// every other window of the call (the reference integrates them one after the other, preint.h:1613-1699; they only depend on the
// data, so here they are all in flight together), what is left is a few 3 x 3 products per record.
//
// Reference lines replaced (paths relative to /root/reference/4DRadarSLAM/include/VelInt):
//   plan_chunks        preint.h:1584-1662   chunk bounds, the stamps of a chunk (plus its end stamp), the sample range of a chunk
//   chain_chunks       preint.h:1664-1699   first chunk as is, later chunks through combinePreints with the running chunk product
//   combine_preints    math_utils.h:689-726 (+ propagateJacobianRp / RR :577-686, jacobianLogMap :227-313, jacobianExpMapZeroM / V
//                      :206-225, jacobianYX :342-349, perturbationPropagation :540-553, propagatePreintCov :556-574)
// types.h:36 declares Vec12 with NINE rows, so perturbationPropagation / propagatePreintCov of the reference run off the end of
// their perturbation vector (undefined behaviour, no result to reproduce); the twelve components they address are used here.
#pragma once

#include <algorithm>
#include <cmath>
#include <limits>
#include <string>
#include <vector>

#include "../../include/gorio_ugpm.h"

namespace gorio {
namespace chunks {

// ------------------------------------------------------------------------------------------------ 3 x 3 helpers (row-major double[9])
inline void mat_mul(const double* a, const double* b, double* r) {
  double t[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) t[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
  std::copy(t, t + 9, r);
}
inline void mat_t(const double* a, double* r) {
  const double t[9] = {a[0], a[3], a[6], a[1], a[4], a[7], a[2], a[5], a[8]};
  std::copy(t, t + 9, r);
}
inline void mat_vec(const double* a, const double* v, double* r) {
  const double t[3] = {a[0] * v[0] + a[1] * v[1] + a[2] * v[2], a[3] * v[0] + a[4] * v[1] + a[5] * v[2], a[6] * v[0] + a[7] * v[1] + a[8] * v[2]};
  std::copy(t, t + 3, r);
}
// expMap (math_utils.h:55-58): Eigen AngleAxis(|v|, v / |v|).toRotationMatrix()
inline void exp_map(const double* v, double* R) {
  const double n2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
  const double ang = std::sqrt(n2);
  double a[3] = {v[0], v[1], v[2]};
  if (n2 > 0.0)
    for (double& q : a) q *= 1.0 / ang;
  const double s = std::sin(ang), c = std::cos(ang);
  const double sa[3] = {s * a[0], s * a[1], s * a[2]}, ca[3] = {(1.0 - c) * a[0], (1.0 - c) * a[1], (1.0 - c) * a[2]};
  double t;
  t = ca[0] * a[1]; R[1] = t - sa[2]; R[3] = t + sa[2];
  t = ca[0] * a[2]; R[2] = t + sa[1]; R[6] = t - sa[1];
  t = ca[1] * a[2]; R[5] = t - sa[0]; R[7] = t + sa[0];
  R[0] = ca[0] * a[0] + c; R[4] = ca[1] * a[1] + c; R[8] = ca[2] * a[2] + c;
}
// logMap (math_utils.h:48-51): Eigen AngleAxis(R) = matrix -> quaternion -> angle * axis
inline void log_map(const double* R, double* r) {
  double q[4];
  double t = R[0] + R[4] + R[8];
  if (t > 0.0) {
    t = std::sqrt(t + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (R[7] - R[5]) * t;
    q[1] = (R[2] - R[6]) * t;
    q[2] = (R[3] - R[1]) * t;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[i * 4]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(R[i * 4] - R[j * 4] - R[k * 4] + 1.0);
    q[i] = 0.5 * t;
    t = 0.5 / t;
    q[3] = (R[k * 3 + j] - R[j * 3 + k]) * t;
    q[j] = (R[j * 3 + i] + R[i * 3 + j]) * t;
    q[k] = (R[k * 3 + i] + R[i * 3 + k]) * t;
  }
  double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
  if (n != 0.0) {
    const double ang = 2.0 * std::atan2(n, std::fabs(q[3]));
    if (q[3] < 0) n = -n;
    for (int a = 0; a < 3; ++a) r[a] = ang * q[a] / n;
  } else {
    r[0] = r[1] = r[2] = 0.0;
  }
}

// d vec(R Exp(M b)) / d b at b = 0 for a 3 x ncol matrix M (vec column-major, 9 x ncol): jacobianYX(R) * jacobianExpMapZeroM(M)
inline void rot_perturbation(const double* R, const double* M, int ncol, double* out /* [9][ncol] */) {
  // [M b]x by columns of the skew matrix: column 0 = (0, m2, -m1), column 1 = (-m2, 0, m0), column 2 = (m1, -m0, 0)
  static const int src[9] = {-1, 2, 1, 2, -1, 0, 1, 0, -1};
  static const double sgn[9] = {0, 1, -1, -1, 0, 1, 1, -1, 0};
  for (int col = 0; col < 3; ++col)
    for (int i = 0; i < 3; ++i)
      for (int k = 0; k < ncol; ++k) {
        double acc = 0.0;
        for (int j = 0; j < 3; ++j) {
          const int q = 3 * col + j;
          const double e = src[q] < 0 ? 0.0 : sgn[q] * M[src[q] * ncol + k];
          acc += R[3 * i + j] * e;
        }
        out[(3 * col + i) * ncol + k] = acc;
      }
}
// jacobianLogMap (math_utils.h:227-313): d log(R) / d vec(R), 3 x 9
inline void log_jacobian(const double* R, double* J /* [3][9] */) {
  std::fill(J, J + 27, 0.0);
  const double trace = R[0] + R[4] + R[8];
  double half = 0.5, d[3] = {0.0, 0.0, 0.0};
  if (trace < 3.0 - 1e-14) {  // kLogTraceTolerance, math_utils.h:12
    const double c = R[0] / 2.0 + R[4] / 2.0 + R[8] / 2.0 - 0.5;
    const double th = std::acos(c);
    half = th / (2 * std::pow(1 - std::pow(c, 2), 0.5));
    const double u[3] = {R[5] - R[7], R[2] - R[6], R[1] - R[3]};
    const double sg[3] = {-1.0, 1.0, -1.0};
    for (int a = 0; a < 3; ++a) d[a] = sg[a] * (u[a] / (4 * (std::pow(c, 2) - 1)) + (th * u[a] * c) / (4 * std::pow(1 - std::pow(c, 2), 1.5)));
  }
  for (int a = 0; a < 3; ++a) J[9 * a + 0] = J[9 * a + 4] = J[9 * a + 8] = d[a];
  J[5] = half; J[7] = -half;
  J[9 + 2] = -half; J[9 + 6] = half;
  J[18 + 1] = half; J[18 + 3] = -half;
}
// propagateJacobianRp (math_utils.h:577-610): d (R Exp(d_r b) (p + d_p b)) / d b, 3 x ncol
inline void jac_rot_point(const double* R, const double* d_r, const double* p, const double* d_p, int ncol, double* out) {
  double dR[27];
  rot_perturbation(R, d_r, ncol, dR);
  for (int i = 0; i < 3; ++i)
    for (int k = 0; k < ncol; ++k) {
      double rd = 0.0;
      for (int j = 0; j < 3; ++j) rd += R[3 * i + j] * d_p[j * ncol + k];
      out[i * ncol + k] = ((rd + dR[i * ncol + k] * p[0]) + dR[(3 + i) * ncol + k] * p[1]) + dR[(6 + i) * ncol + k] * p[2];
    }
}
// propagateJacobianRR (math_utils.h:612-686): d log(R1 Exp(d_r1 b) R2 Exp(d_r2 b)) / d b, 3 x ncol
inline void jac_rot_rot(const double* R1, const double* d_r1, const double* R2, const double* d_r2, int ncol, double* out) {
  double dR1[27], dR2[27], dRR[27], R12[9], JL[27];
  rot_perturbation(R1, d_r1, ncol, dR1);
  rot_perturbation(R2, d_r2, ncol, dR2);
  for (int col = 0; col < 3; ++col)
    for (int i = 0; i < 3; ++i)
      for (int k = 0; k < ncol; ++k) {
        double rd = 0.0;
        for (int j = 0; j < 3; ++j) rd += R1[3 * i + j] * dR2[(3 * col + j) * ncol + k];
        dRR[(3 * col + i) * ncol + k] = ((rd + dR1[i * ncol + k] * R2[col]) + dR1[(3 + i) * ncol + k] * R2[3 + col]) + dR1[(6 + i) * ncol + k] * R2[6 + col];
      }
  mat_mul(R1, R2, R12);
  log_jacobian(R12, JL);
  for (int i = 0; i < 3; ++i)
    for (int k = 0; k < ncol; ++k) {
      double acc = 0.0;
      for (int q = 0; q < 9; ++q) acc += JL[9 * i + q] * dRR[q * ncol + k];
      out[i * ncol + k] = acc;
    }
}
// perturbationPropagation (math_utils.h:540-553) with the twelve components it addresses
inline void perturb(const double* eps, const gorio_ugpm_meas& prev, const gorio_ugpm_meas& cur, double* out6) {
  double E1[9], E2[9], RE[9], A[9], Rt[9], v[3];
  exp_map(eps, E1);
  exp_map(eps + 6, E2);
  mat_mul(prev.delta_R, E1, RE);
  mat_t(cur.delta_R, Rt);
  mat_mul(Rt, E1, A);
  mat_mul(A, cur.delta_R, A);
  mat_mul(A, E2, A);
  log_map(A, out6);
  for (int a = 0; a < 3; ++a) v[a] = cur.delta_p[a] + eps[9 + a];
  mat_vec(RE, v, v);
  for (int a = 0; a < 3; ++a) out6[3 + a] = eps[3 + a] + v[a];
}
// propagatePreintCov (math_utils.h:556-574)
inline void chain_cov(const gorio_ugpm_meas& prev, const gorio_ugpm_meas& cur, double* cov36) {
  const double h = 1e-5;
  double eps[12] = {0}, base[6], pert[6], J[6][12], T[6][12];
  perturb(eps, prev, cur, base);
  for (int i = 0; i < 12; ++i) {
    eps[i] = h;
    perturb(eps, prev, cur, pert);
    for (int a = 0; a < 6; ++a) J[a][i] = (pert[a] - base[a]) / h;
    eps[i] = 0.0;
  }
  for (int a = 0; a < 6; ++a)
    for (int j = 0; j < 12; ++j) {
      const double* C = j < 6 ? prev.cov : cur.cov;  // blkdiag(prev.cov, cur.cov): the off-diagonal blocks are zero
      const int k0 = j < 6 ? 0 : 6;
      double acc = 0.0;
      for (int k = 0; k < 6; ++k) acc += J[a][k0 + k] * C[6 * k + (j - k0)];
      T[a][j] = acc;
    }
  for (int a = 0; a < 6; ++a)
    for (int b = 0; b < 6; ++b) {
      double acc = 0.0;
      for (int k = 0; k < 12; ++k) acc += T[a][k] * J[b][k];
      cov36[6 * a + b] = acc;
    }
}

// combinePreints (math_utils.h:689-726)
inline gorio_ugpm_meas combine_preints(const gorio_ugpm_meas& prev, const gorio_ugpm_meas& cur) {
  if (cur.dt == 0.0) return prev;
  gorio_ugpm_meas o = cur;
  chain_cov(prev, cur, o.cov);
  double t9[9], t3[3], Rt[9];
  mat_mul(prev.delta_R, cur.d_delta_p_d_bv, t9);
  for (int i = 0; i < 9; ++i) o.d_delta_p_d_bv[i] = prev.d_delta_p_d_bv[i] + t9[i];
  jac_rot_point(prev.delta_R, prev.d_delta_R_d_bw, cur.delta_p, cur.d_delta_p_d_bw, 3, t9);
  for (int i = 0; i < 9; ++i) o.d_delta_p_d_bw[i] = prev.d_delta_p_d_bw[i] + t9[i];
  mat_t(cur.delta_R, Rt);
  jac_rot_rot(Rt, prev.d_delta_R_d_bw, cur.delta_R, cur.d_delta_R_d_bw, 3, o.d_delta_R_d_bw);
  jac_rot_point(prev.delta_R, prev.d_delta_R_d_t, cur.delta_p, cur.d_delta_p_d_t, 1, t3);
  for (int i = 0; i < 3; ++i) o.d_delta_p_d_t[i] = prev.d_delta_p_d_t[i] + t3[i];
  jac_rot_rot(Rt, prev.d_delta_R_d_t, cur.delta_R, cur.d_delta_R_d_t, 1, o.d_delta_R_d_t);
  mat_vec(prev.delta_R, cur.delta_p, t3);
  for (int i = 0; i < 3; ++i) o.delta_p[i] = prev.delta_p[i] + t3[i];
  mat_mul(prev.delta_R, cur.delta_R, o.delta_R);
  o.dt = prev.dt + cur.dt;
  o.dt_sq_half = 0.5 * o.dt * o.dt;
  return o;
}

// VelPreintegration::get's bias-prior inflation (preint.h:1744-1757) of a chained record
inline void inflate(gorio_ugpm_meas& m, double vel_bias_std, double gyr_bias_std) {
  if (!(vel_bias_std > 0.0 || gyr_bias_std > 0.0)) return;
  double J[36] = {0};
  const double bc[6] = {gyr_bias_std * gyr_bias_std, gyr_bias_std * gyr_bias_std, gyr_bias_std * gyr_bias_std, vel_bias_std * vel_bias_std, vel_bias_std * vel_bias_std,
                        vel_bias_std * vel_bias_std};
  for (int a = 0; a < 3; ++a) {
    J[6 * a + a] = 1.0;  // inverseJacobianRighthandSO3(0)
    for (int b = 0; b < 3; ++b) {
      J[6 * (3 + a) + b] = m.d_delta_p_d_bw[3 * a + b];
      J[6 * (3 + a) + 3 + b] = m.d_delta_p_d_bv[3 * a + b];
    }
  }
  for (int a = 0; a < 6; ++a)
    for (int b = 0; b < 6; ++b) {
      double s = 0.0;
      for (int k = 0; k < 6; ++k) s += J[6 * a + k] * bc[k] * J[6 * b + k];
      m.cov[6 * a + b] += s;
    }
}

// ------------------------------------------------------------------------------------------------ plan
struct Chunk {
  double start_t;
  int g0, ng, v0, nv;            // sample ranges of the request's arrays (from < t < to, types.h:187-222)
  std::vector<double> infer_t;   // the request's stamps that fall into the chunk, group-major, then the chunk's end stamp (not for the last chunk)
  std::vector<int> group_sizes;  // n_groups entries (+ 1 for the end stamp)
  std::vector<int> dest;         // per stamp: index of the request's output record (-1 for the end stamp)
};
struct Plan {
  std::vector<Chunk> chunks;
};

// samples with from < t < to of an ascending array: [first, first + count)
inline void sample_range(const double* t, int n, double from, double to, int& first, int& count) {
  first = 0;
  count = 0;
  if (!(from < to)) return;
  int a = 0;
  while (a < n && !(t[a] > from)) ++a;
  int b = a;
  while (b < n && t[b] < to) ++b;
  first = a;
  count = b - a;
}

// preint.h:1586-1662.  Returns 0 or a gorio_ugpm_status with the text in `err`.
inline int plan_chunks(const gorio_ugpm_window& w, Plan& plan, std::string& err) {
  if (!(w.quantum > 0.0)) { err = "opt.quantum = 0 divides by zero in the reference (preint.h:1609)"; return GORIO_UGPM_ERR_INVALID; }
  if (w.n_gyr < 2 || w.n_vel < 2) { err = "InterpolateLinear: this function need at least 2 data points to interpolate"; return GORIO_UGPM_ERR_RANGE; }
  std::vector<int> sizes;
  if (w.group_sizes && w.n_groups > 0) sizes.assign(w.group_sizes, w.group_sizes + w.n_groups);
  else sizes.assign(1, w.n_infer);
  std::vector<int> first(sizes.size(), 0);
  long tot = 0;
  for (size_t g = 0; g < sizes.size(); ++g) {
    if (sizes[g] < 0) { err = "negative group size"; return GORIO_UGPM_ERR_INVALID; }
    first[g] = (int)tot;
    tot += sizes[g];
  }
  if (tot != w.n_infer) { err = "group_sizes do not add up to n_infer"; return GORIO_UGPM_ERR_INVALID; }
  if (sizes[0] == 0) { err = "chunked mode reads the last stamp of the first vector of inference times (preint.h:1594)"; return GORIO_UGPM_ERR_INVALID; }
  // last inference stamp: the largest of the vectors' LAST elements (preint.h:1588-1598)
  int nonempty = 0;
  double last_of_lasts = -std::numeric_limits<double>::infinity();
  for (size_t g = 0; g < sizes.size(); ++g)
    if (sizes[g] > 0) {
      ++nonempty;
      last_of_lasts = std::max(last_of_lasts, w.infer_t[first[g] + sizes[g] - 1]);
    }
  double last_t = w.infer_t[first[0] + sizes[0] - 1];
  if (nonempty > 1) last_t = last_of_lasts;
  const double vel_period = (w.vel_t[w.n_vel - 1] - w.vel_t[0]) / (w.n_vel - 1);
  const double gyr_period = (w.gyr_t[w.n_gyr - 1] - w.gyr_t[0]) / (w.n_gyr - 1);
  const double t_overlap = std::max(vel_period, gyr_period) * w.overlap;
  const double nbf = std::ceil((last_t - w.start_t) / w.quantum);
  if (!(nbf >= 0.0) || nbf > 4096.0) { err = "chunk count outside [0, 4096] (last inference time before start_t, or a tiny quantum)"; return GORIO_UGPM_ERR_RANGE; }
  int nb = (int)nbf;
  if (nb == 0) nb = 1;
  std::vector<int> ptr(sizes.size(), 0);
  plan.chunks.resize(nb);
  for (int i = 0; i < nb; ++i) {
    Chunk& c = plan.chunks[i];
    c.start_t = w.start_t + (i * w.quantum);
    double end_t = w.start_t + ((i + 1) * w.quantum);
    const bool last = i == nb - 1;
    if (last) end_t = std::numeric_limits<double>::infinity();
    for (size_t g = 0; g < sizes.size(); ++g) {
      int cnt = 0;
      while (ptr[g] < sizes[g] && w.infer_t[first[g] + ptr[g]] < end_t) {
        c.infer_t.push_back(w.infer_t[first[g] + ptr[g]]);
        c.dest.push_back(first[g] + ptr[g]);
        ++ptr[g];
        ++cnt;
      }
      c.group_sizes.push_back(cnt);
    }
    if (!last) {
      c.infer_t.push_back(end_t);
      c.dest.push_back(-1);
      c.group_sizes.push_back(1);
    }
    if (c.infer_t.empty()) { err = "a chunk without inference times (max_element of an empty range in the reference, preint.h:1552)"; return GORIO_UGPM_ERR_INVALID; }
    const double from = c.start_t - t_overlap, to = end_t + t_overlap;
    if (!(from <= to)) { err = "The argument of GyroVelData::Get are not consistent"; return GORIO_UGPM_ERR_ARGUMENT; }
    sample_range(w.gyr_t, w.n_gyr, from, to, c.g0, c.ng);
    sample_range(w.vel_t, w.n_vel, from, to, c.v0, c.nv);
  }
  return 0;
}

// preint.h:1664-1699: `recs[i]` are the records of chunk i in the order of plan.chunks[i].infer_t; out: the request's n_infer records
inline void chain_chunks(const Plan& plan, const std::vector<const gorio_ugpm_meas*>& recs, double vel_bias_std, double gyr_bias_std, gorio_ugpm_meas* out) {
  gorio_ugpm_meas running{};
  const int nb = (int)plan.chunks.size();
  for (int i = 0; i < nb; ++i) {
    const Chunk& c = plan.chunks[i];
    const int n = (int)c.infer_t.size();
    for (int k = 0; k < n; ++k) {
      if (c.dest[k] < 0) continue;
      gorio_ugpm_meas m = i == 0 ? recs[i][k] : combine_preints(running, recs[i][k]);
      inflate(m, vel_bias_std, gyr_bias_std);
      out[c.dest[k]] = m;
    }
    if (i != nb - 1) running = i == 0 ? recs[i][n - 1] : combine_preints(running, recs[i][n - 1]);
  }
}

}  // namespace chunks
}  // namespace gorio
