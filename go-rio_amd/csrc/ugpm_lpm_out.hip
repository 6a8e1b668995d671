// ugpm_lpm_out.hip -- opt.type = LPM as the OUTPUT method of ugpm::VelPreintegration (preint.h:1567-1580): the complete
// IterativeIntegrator (preint.h:170-742) with its rotation covariance, numeric post-integration Jacobians, velocity re-projection
// with Jacobians and the trapezoid position integration.  Included by ugpm_api.hip after ugpm_kernels.hip (shares its SO(3) helpers).
//
// Paths relative to /root/reference/4DRadarSLAM/include/VelInt: PRE = preint.h, MATH = math_utils.h, TYPES = types.h.
//
// The merged, sorted time line (SortIndexTracker2, TYPES:332-458 -- queries, {start, start + 0.01}, ego-velocity stamps and the
// min_freq filler stamps of PRE:228-237) is bookkeeping and is laid out by the host; everything numeric runs here:
//   lpm_out_steps_kernel   interpolated rates at every stamp (MATH:487-532) and the step rotations E_i = Exp(w_i dt_i) of the five
//                          integrations (nominal, gyro stamps shifted by -0.01, gyro axis a + 1e-4), plus J_r(w_i dt_i) dt_i for
//                          the covariance recursion -- all stamps in parallel
//   lpm_out_scan_kernel    the running products P_{i+1} = P_i E_i (PRE:468, 505) and the covariance recursion (PRE:456-466): the
//                          only sequential part, one lane per integration
//   lpm_out_finish_kernel  re-referencing to the start stamp (PRE:477-485), d_delta_R_d_t / d_delta_R_d_bw (PRE:352-379),
//                          reprojectVelData with Jacobians (MATH:428-483), posePreintLPM (PRE:524-667) and the 83-double records
// This path is not on Go-RIO's launch configuration (the nodelet uses UGPM); it is written for parity, not for speed.
#include <hip/hip_runtime.h>

namespace gorio {
namespace ug {

struct LpmOutWin {
  const double* gyr_t;   // [G]
  const double* gyr;     // [3][G] raw (bias not removed)
  const double* vel_t;   // [V]
  const double* vel;     // [3][V]
  const double* infer_t; // [n_infer]
  const double* tl;      // [T] merged stamps, ascending
  const int* kind;       // [T] 0 query, 1 {start, start + 0.01}, 2 ego-velocity stamp, 3 filler
  const int* kidx;       // [T] index inside its list
  const int* qpos;       // [n_infer] rank of query j in the merged line
  const int* qorder;     // [n_infer] query indices by ascending time (stable)
  const int* qrot;       // [n_infer] rank whose ROTATION part goes to record j: the j-th smallest stamp of j's inner vector (PRE:259)
  int G, V, T, n_infer;
  int start_index;       // rank of `start` (PRE:239)
  int dt_index;          // rank of `start + 0.01` (PRE:268)
  double start_t, gyr_var, vel_var;
  double gyr_bias[3], vel_bias[3];
  double vel_bias_std, gyr_bias_std;
  double* E;      // [5][T][9]  step rotations, then running products, then re-referenced delta_R
  double* B;      // [T][9]     J_r(w dt) dt of the nominal integration
  double* cov3;   // [T][9]     rotation covariance after every stamp
  double* dRdt;   // [T][3]
  double* dRdbw;  // [T][9]
  double* velr;   // [3][V]  re-projected velocities
  double* d_bw;   // [2][V][9]  first half: row a = d velr_a / d b_w; second half: d velr / d b_v = delta_R at the stamp
  double* d_dt;   // [3][V]
  double* dp_shift;  // [n_infer][3]
  double* out;    // [n_infer][83]
  int* status;
};

__device__ __forceinline__ LpmOutWin load_lpm_win(const LpmOutWin* __restrict__ wins, int i) {
  LpmOutWin w;
  __builtin_memcpy(&w, (const __attribute__((address_space(4))) void*)(wins + i), sizeof(LpmOutWin));
  return w;
}

// linearInterpolation (MATH:487-532) of one gyro axis at time t: the segment p with time[p] < t <= time[p + 1] (what the reference's
// forward-moving pointer reaches for ascending queries), clamped to the first / last segment outside the data span
__device__ __forceinline__ double lpm_interp(const LpmOutWin& w, int axis, double t, double tshift, double bias, double bump) {
  const int G = w.G;
  int p = 0;
  if (t > w.gyr_t[0] - tshift) {
    p = lower_bound_f(G, t, w.gyr_t, tshift) - 1;
    if (p > G - 2) p = G - 2;
    if (p < 0) p = 0;
  }
  const double t0 = w.gyr_t[p] - tshift, t1 = w.gyr_t[p + 1] - tshift;
  const double d0 = w.gyr[(size_t)axis * G + p] - bias, d1 = w.gyr[(size_t)axis * G + p + 1] - bias;
  const double al = (d1 - d0) / (t1 - t0);
  const double be = d0 - (al * t0);
  return (al * t + be) + bump;  // the bias-Jacobian runs add 1e-4 to the INTERPOLATED rate (PRE:367-370)
}

// grid: (ceil(max T / 256), 5, windows), block 256
__global__ __launch_bounds__(256) void lpm_out_steps_kernel(const LpmOutWin* __restrict__ wins) {
  const LpmOutWin w = load_lpm_win(wins, blockIdx.z);
  if (*w.status != 0) return;
  const int i = blockIdx.x * 256 + threadIdx.x, var = blockIdx.y;
  if (i >= w.T - 1) return;
  const double t = w.tl[i], dt = w.tl[i + 1] - t;
  const double tshift = var == 1 ? kDt : 0.0;
  V3 g;
  g.x = lpm_interp(w, 0, t, tshift, w.gyr_bias[0], var == 2 ? kBw : 0.0) * dt;
  g.y = lpm_interp(w, 1, t, tshift, w.gyr_bias[1], var == 3 ? kBw : 0.0) * dt;
  g.z = lpm_interp(w, 2, t, tshift, w.gyr_bias[2], var == 4 ? kBw : 0.0) * dt;
  double* E = w.E + ((size_t)var * w.T + i) * 9;
  if (var != 0) {
    storeM(E, expMap(g));  // PRE:503
    return;
  }
  // nominal integration, PRE:424-453: explicit Rodrigues form with its own 1e-10 threshold, and J_r for the covariance
  M3 e_R = eye3(), j_r = eye3();
  const double gn = vnorm(g);
  if (gn > 0.0000000001) {
    const M3 S = skew(g), S2 = mmul(S, S);
    const double s = sin(gn), gn2 = gn * gn, sc2 = (1 - cos(gn)) / gn2;
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      e_R.m[q] = e_R.m[q] + ((s / gn) * S.m[q]) + (sc2 * S2.m[q]);
      j_r.m[q] = j_r.m[q] - (sc2 * S.m[q]) + (((gn - s) / (gn2 * gn)) * S2.m[q]);
    }
  }
  storeM(E, e_R);
#pragma unroll
  for (int q = 0; q < 9; ++q) j_r.m[q] *= dt;
  storeM(w.B + (size_t)i * 9, j_r);
}

// grid: (5, windows), block 64: lane 0 multiplies the steps up (in place: E[i] becomes P_i = E_0 ... E_{i-1}); the nominal one also
// runs the covariance recursion cov <- E^T cov E + B diag(var) B^T for steps beyond the start stamp (PRE:456-466)
__global__ __launch_bounds__(64) void lpm_out_scan_kernel(const LpmOutWin* __restrict__ wins) {
  const LpmOutWin w = load_lpm_win(wins, blockIdx.y);
  if (*w.status != 0 || threadIdx.x != 0) return;
  const int var = blockIdx.x, T = w.T;
  double* E = w.E + (size_t)var * T * 9;
  M3 P = eye3(), C = M3{{0, 0, 0, 0, 0, 0, 0, 0, 0}};
  if (var == 0) storeM(w.cov3, C);
  M3 next = T > 1 ? loadM(E) : eye3();
  for (int i = 0; i < T - 1; ++i) {
    const M3 Ei = next;
    if (i + 2 < T) next = loadM(E + (size_t)(i + 1) * 9);  // the next step's load is in flight during this step's products
    storeM(E + (size_t)i * 9, P);
    if (var == 0) {
      if ((i + 1) > w.start_index) {
        const M3 A = mtr(Ei), Bm = loadM(w.B + (size_t)i * 9);
        M3 Bv = Bm;
#pragma unroll
        for (int q = 0; q < 9; ++q) Bv.m[q] *= w.gyr_var;  // B * diag(var, var, var)
        const M3 t1 = mmul(mmul(A, C), mtr(A)), t2 = mmul(Bv, mtr(Bm));
#pragma unroll
        for (int q = 0; q < 9; ++q) C.m[q] = t1.m[q] + t2.m[q];
      }
      storeM(w.cov3 + (size_t)(i + 1) * 9, C);
    }
    P = mmul(P, Ei);
  }
  storeM(E + (size_t)(T - 1) * 9, P);
}

// Trapezoid integral of the piecewise-linear data d(t) from `start` to every query >= start (posePreintLPMPartial PRE:669-741 and
// the main loop of posePreintLPM PRE:552-665) for ONE axis on one lane.  tsh / bump express the shifted copy of the data
// (vel_time - 0.01, vel_data + 0.01 d_vel_d_dt, PRE:537-545) without materialising it.  With `full`, it also accumulates the
// Jacobians against b_v and b_w and writes delta_p, d_delta_p_d_t, the variance and the two Jacobian rows into the records.
__device__ void lpm_out_position(const LpmOutWin& w, int axis, bool full) {
  const int V = w.V;
  const double start = w.start_t;
  const double tsh = full ? 0.0 : kDt;
  auto vt = [&](int i) -> double { return w.vel_t[i] - tsh; };
  auto vd = [&](int i) -> double { return full ? w.velr[(size_t)axis * V + i] : w.velr[(size_t)axis * V + i] + kDt * w.d_dt[(size_t)axis * V + i]; };
  int data_ptr = 0;
  while (vt(data_ptr + 1) < start) {
    data_ptr++;
    if (data_ptr == V - 1) {
      *w.status = -3;  // "the start_time is not in the data domain", PRE:686 / 566
      return;
    }
  }
  int ptr = data_ptr;
  double alpha = (vd(ptr + 1) - vd(ptr)) / (vt(ptr + 1) - vt(ptr));
  double beta = vd(ptr) - alpha * vt(ptr);
  double t_0 = start, t_1 = vt(ptr + 1);
  double d_0 = alpha * vt(ptr) + beta, d_1 = vd(ptr + 1);
  double backup = 0.0;
  // Jacobian rows of the data nodes: d_vel_d_bw[axis].row(i) in the first half of the table, d_vel_d_bv[axis].row(i) (= row `axis` of
  // delta_R at velocity stamp i, MATH:447-449) in the second half
  auto row_bv = [&](int i) -> V3 { return load3(w.d_bw + ((size_t)V + i) * 9 + axis * 3); };
  auto row_bw = [&](int i) -> V3 { return load3(w.d_bw + (size_t)i * 9 + axis * 3); };
  double ratio = (start - vt(ptr)) / (vt(ptr + 1) - vt(ptr));
  V3 g0w = v3(0, 0, 0), g0v = v3(0, 0, 0), accv = v3(0, 0, 0), accw = v3(0, 0, 0);
  if (full) {
    g0w = ratio * row_bw(ptr + 1) + (1 - ratio) * row_bw(ptr);
    g0v = ratio * row_bv(ptr + 1) + (1 - ratio) * row_bv(ptr);
  }
  for (int k = 0; k < w.n_infer; ++k) {
    const int q = w.qorder[k];
    const double ti = w.infer_t[q];
    if (ti < start) continue;  // PRE:553-557: queries ahead of the start are not integrated
    if (ti > vt(0)) {
      while (true) {
        if ((ti >= vt(ptr)) && (ti <= vt(ptr + 1))) break;
        if (ptr < (V - 2)) {
          backup = backup + ((t_1 - t_0) * (d_0 + d_1) / 2.0);
          if (full) {
            const double dt = t_1 - t_0;
            accv = accv + (dt / 2.0) * (g0v + row_bv(ptr + 1));
            accw = accw + (dt / 2.0) * (g0w + row_bw(ptr + 1));
          }
          ptr++;
          t_0 = vt(ptr);
          t_1 = vt(ptr + 1);
          d_0 = vd(ptr);
          d_1 = vd(ptr + 1);
          alpha = (d_1 - d_0) / (t_1 - t_0);
          beta = d_0 - alpha * t_0;
          if (full) {
            g0v = row_bv(ptr);
            g0w = row_bw(ptr);
          }
        } else {
          break;
        }
      }
    }
    const double temp_d_1 = alpha * ti + beta;
    const double temp_d_p = backup + ((ti - t_0) * (d_0 + temp_d_1) / 2.0);
    if (!full) {
      w.dp_shift[(size_t)q * 3 + axis] = temp_d_p;
      continue;
    }
    double* o = w.out + (size_t)q * 83;
    o[80 + axis] = (w.dp_shift[(size_t)q * 3 + axis] - temp_d_p) / kDt;  // d_delta_p_d_t, PRE:646
    o[9 + axis] = temp_d_p;
    o[14 + (3 + axis) * 6 + 3 + axis] = (ti - start) * w.vel_var;         // PRE:648
    ratio = (ti - vt(ptr)) / (vt(ptr + 1) - vt(ptr));
    const V3 g1w = ratio * row_bw(ptr + 1) + (1 - ratio) * row_bw(ptr);
    const V3 g1v = ratio * row_bv(ptr + 1) + (1 - ratio) * row_bv(ptr);
    const double dt = ti - t_0;
    store3(o + 71 + axis * 3, accv + (dt / 2.0) * (g0v + g1v));  // d_delta_p_d_bv row
    store3(o + 62 + axis * 3, accw + (dt / 2.0) * (g0w + g1w));  // d_delta_p_d_bw row
  }
}

// grid: (windows), block 256
__global__ __launch_bounds__(256) void lpm_out_finish_kernel(const LpmOutWin* __restrict__ wins) {
  const LpmOutWin w = load_lpm_win(wins, blockIdx.x);
  if (*w.status != 0) return;
  const int T = w.T, V = w.V, tid = threadIdx.x;
  __shared__ double sPsT[5][9];
  if (tid < 5) storeM(sPsT[tid], mtr(loadM(w.E + ((size_t)tid * T + w.start_index) * 9)));
  __syncthreads();
  // ---- re-reference every integration to the start stamp: the net effect of PRE:477-485 / 508-517 on every j is P_start^T P_j
  for (int e = tid; e < 5 * T; e += 256) {
    const int var = e / T;
    double* p = w.E + (size_t)e * 9;
    storeM(p, mmul(loadM(sPsT[var]), loadM(p)));
  }
  __syncthreads();
  // ---- numeric post-integration Jacobians at the stamps of interest (everything but the filler), PRE:352-379
  for (int j = tid; j < T; j += 256) {
    if (w.kind[j] == 3) continue;
    const M3 Rt = mtr(loadM(w.E + (size_t)j * 9));
    store3(w.dRdt + (size_t)j * 3, (1.0 / kDt) * logMap(mmul(Rt, loadM(w.E + ((size_t)T + j) * 9))));
    double* M = w.dRdbw + (size_t)j * 9;
    for (int a = 0; a < 3; ++a) {
      const V3 c = (1.0 / kBw) * logMap(mmul(Rt, loadM(w.E + ((size_t)(2 + a) * T + j) * 9)));
      M[0 * 3 + a] = c.x; M[1 * 3 + a] = c.y; M[2 * 3 + a] = c.z;  // column a
    }
  }
  __syncthreads();
  // ---- reprojectVelData with Jacobians, MATH:428-483
  const M3 RdtT = mtr(loadM(w.E + (size_t)w.dt_index * 9));  // delta_R_dt_start^T, PRE:268
  for (int j = tid; j < T; j += 256) {
    if (w.kind[j] != 2) continue;
    const int i = w.kidx[j];
    const M3 R = loadM(w.E + (size_t)j * 9);
    const V3 v = v3(w.vel[i] - w.vel_bias[0], w.vel[(size_t)V + i] - w.vel_bias[1], w.vel[(size_t)2 * V + i] - w.vel_bias[2]);
    const M3 M = loadM(w.dRdbw + (size_t)j * 9);
    storeM(w.d_bw + ((size_t)V + i) * 9, R);  // d_vel_d_bv[a].row(i) = R.row(a), MATH:447-449
    for (int a = 0; a < 3; ++a) {
      // MATH:453-468 multiply a 1 x 9 row (R(a, col) v(q)) with the 9 x 3 jacobianExpMapZeroM(d_delta_R_d_bw) (MATH:212-225), whose
      // rows are  sum_m eps(q, col, m) M.row(m):  the product collapses to (v x R.row(a)) M
      const V3 c = cross(v, v3(R.m[a * 3], R.m[a * 3 + 1], R.m[a * 3 + 2]));
      double* o = w.d_bw + (size_t)i * 9 + a * 3;
      o[0] = c.x * M.m[0] + c.y * M.m[3] + c.z * M.m[6];
      o[1] = c.x * M.m[1] + c.y * M.m[4] + c.z * M.m[7];
      o[2] = c.x * M.m[2] + c.y * M.m[5] + c.z * M.m[8];
    }
    const V3 vr = mvec(R, v);
    const V3 ddt = (1.0 / kDt) * (mvec(RdtT, vr) - vr);
    w.velr[i] = vr.x; w.velr[(size_t)V + i] = vr.y; w.velr[(size_t)2 * V + i] = vr.z;
    w.d_dt[i] = ddt.x; w.d_dt[(size_t)V + i] = ddt.y; w.d_dt[(size_t)2 * V + i] = ddt.z;
  }
  // ---- records: rotation part of every query (rotPreint's output, PRE:414-418, 469-472), zero elsewhere
  for (int q = tid; q < w.n_infer; q += 256) {
    const int j = w.qrot[q];
    double* o = w.out + (size_t)q * 83;
    for (int e = 0; e < 83; ++e) o[e] = 0.0;
    storeM(o, loadM(w.E + (size_t)j * 9));
    const double dt = w.tl[j] - w.start_t;
    o[12] = dt;
    o[13] = dt * dt * 0.5;
    const double* c3 = w.cov3 + (size_t)j * 9;
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) o[14 + r * 6 + c] = c3[r * 3 + c];
    for (int d = 0; d < 6; ++d)
      if (o[14 + d * 6 + d] < 1e-6) o[14 + d * 6 + d] = 1e-6;  // minCovDiag, PRE:393-405
    for (int e = 0; e < 9; ++e) o[50 + e] = w.dRdbw[(size_t)j * 9 + e];
    for (int e = 0; e < 3; ++e) o[59 + e] = w.dRdt[(size_t)j * 3 + e];
    for (int e = 0; e < 3; ++e) w.dp_shift[(size_t)q * 3 + e] = 0.0;
  }
  __syncthreads();
  // ---- posePreintLPM, PRE:524-667: a partial pass on the time-shifted data, then the full pass (one lane per axis)
  if (tid < 3) {
    lpm_out_position(w, tid, false);
    lpm_out_position(w, tid, true);
  }
  __syncthreads();
  // ---- bias-prior inflation of VelPreintegration::get, PRE:1744-1757: cov += J diag(b) J^T, J = [I 0; d_p_d_bw d_p_d_bv]
  if (w.vel_bias_std > 0.0 || w.gyr_bias_std > 0.0) {
    for (int q = tid; q < w.n_infer; q += 256) {
      double* o = w.out + (size_t)q * 83;
      double J[6][6];
      for (int a = 0; a < 6; ++a)
        for (int b = 0; b < 6; ++b) J[a][b] = 0.0;
      for (int a = 0; a < 3; ++a) {
        J[a][a] = 1.0;
        for (int b = 0; b < 3; ++b) {
          J[3 + a][b] = o[62 + a * 3 + b];
          J[3 + a][3 + b] = o[71 + a * 3 + b];
        }
      }
      const double g2 = w.gyr_bias_std * w.gyr_bias_std, v2 = w.vel_bias_std * w.vel_bias_std;
      for (int a = 0; a < 6; ++a)
        for (int b = 0; b < 6; ++b) {
          double s = 0.0;
          for (int k = 0; k < 6; ++k) s += J[a][k] * (k < 3 ? g2 : v2) * J[b][k];
          o[14 + a * 6 + b] += s;
        }
    }
  }
}

}  // namespace ug
}  // namespace gorio
