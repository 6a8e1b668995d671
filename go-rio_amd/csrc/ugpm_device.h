// ugpm_device.h -- device-side layout of one GP pre-integration window (shared by ugpm_kernels.hip / ugpm_api.hip).
//
// All matrices are fp64, row-major, resident in HBM for the whole batch (one slab per window, ~11 MB at S = 66,
// n_g = n_v = 259; hundreds of windows fit 288 GB trivially).  Notation follows the reference (VelInt/preint.h):
//   S = nb_state_, G = nb_gyr_, V = nb_vel_ (samples strictly inside the padded state window, PRE:789-792).
#pragma once
#include <stdint.h>

namespace gorio {

constexpr int kAtaTilesLm = 24, kAtaTilesCorr = 48;  // 16 x 16 output tiles per workgroup (8 waves x 4 / 6 accumulators)
constexpr int kWinInts = 48;        // ints per window: lmi[16], status at 16
constexpr int kVelBlocks = 3;       // problem #2's normal matrix is block diagonal (one block per velocity channel): lm_step_kernel runs one workgroup per block

struct UgpmWin {
  // ---- inputs
  const double* gyr_t;  // [G]
  const double* gyr;    // [3][G]  raw samples (bias NOT removed)
  const double* vel_t;  // [V]
  const double* vel;    // [3][V]
  const double* infer_t;  // [n_infer]
  double* state_t;        // [S]
  int G, V, S, n_infer;
  int correlate, overlap;
  double start_t, state_freq, gyr_var, vel_var;
  double gyr_bias[3], vel_bias[3];
  double vel_bias_std, gyr_bias_std;
  // ---- LPM stage (5 rotation integrations: 0 base, 1 time-shifted data, 2..4 gyro-bias perturbed)
  double* Rq;      // [5][2][S][9]  delta_R at t_vect (0) and t_vect + 0.01 (1)
  double* Rstart;  // [5][9]        delta_R at start_t
  double* velr;    // [3][V]        velocities rotated into the LPM start frame (variant 0)
  double* dp;      // [2][S][3]     delta_p at t_vect / t_vect + 0.01 (variant 0)
  double* r0;      // [5][S][3]     unwrapped rotation vectors at t_vect
  double* r1;      // [5][S][3]     ... at t_vect + 0.01
  // ---- GP state (PRE:1161-1175)
  double* s_dr;    // [3][S]  state_d_r_ (mean removed)
  double* s_vel;   // [3][S]  state_vel_
  double* hyper;   // [6][4]  l2, sf2, sz2, mean
  double* d_r_dt_local;        // [S][3]
  double* d_r_dt_local_shift;  // [S][3]
  double* delta_r_time;        // [S][3]
  double* delta_r_bw;          // [3][S][3]
  double* d_r_bw_local_shift;  // [3][S][3]
  // ---- Gram stage (PRE:832-866)
  double* Kinv;      // [6][S][S]
  double* KKinv;     // [6][S][S]
  double* KintKinv;  // [3][S][S]
  double* var;       // [6][S]  state_var_
  double* wgp;       // [6][S]  GpNorm weights sqrt(1 / (1000 var)) (COST:31, 40)
  double* sstd;      // [6][S]  state_std
  // ---- cross-kernel products (COST:183-190, 293-308)
  double* KsKinv;       // [3][G][S]
  double* KsIntKinv;    // [3][G][S]
  double* KgyrIntKinv;  // [3][V][S]
  double* KvelKinv;     // [3][V][S]
  // ---- LM problems (Ceres restatement)
  double* Jrot;   // [(3S+3G)][3S]
  double* Jvel;   // [(3V+3S)][3S]
  double* res;    // [max rows]
  double* res_new;
  double* JtJ;    // [3S][3S]
  double* lhs;    // [3S][3S]
  double* lmv;    // 8 vectors of 3S: g, scale, diag, step, delta, x, x_new, tmp
  double* sample_tmp;  // [max(G,V)][24] per-sample scratch of the evaluators
  double* sample_tmp_c;  // the same for corr_jac_kernel, which runs beside them on the second stream
  // ---- state correlation (PRE:886-940, 1478-1492)
  double* Jc;      // [(3G+3V)][6S]
  double* Ac;      // [6S][6S]  J^T J + 1e-5 I  -> Cholesky factor L
  double* Linv;    // [6S][6S]
  double* dsc;     // [6S]  state_std / sqrt(diag(A^-1))
  // ---- inference tables (PRE:978-1060, 1401-1441)
  double* alpha;      // [6][S]
  double* state_r;    // [3][S]
  double* d_state_bw; // [3][S][3]
  double* d_d_r_dt;   // [3][S]
  double* d_vel_bv;   // [3][S][3]
  double* d_vel_bw;   // [3][S][3]
  double* d_vel_dt;   // [3][S]
  double* out;        // [n_infer][83]
  // ---- LM control block (device resident)
  double* lmc;  // [16]: 0 cost, 1 cost_new, 2 radius, 3 decrease_factor, 4 x_norm, 5 / 6 unused, 7 cost0, 8..10 model cost change and
                // 11..13 squared step norm of the (up to three) diagonal blocks lm_step_kernel solved
  int* lmi;     // [16]: 0 iter, 1 done, 2 reuse_diag, 3 need_J, 4 step_valid, 5 termination, 6 successful, 7 problem (0 rot, 1 vel)
  int* status;  // [1] per-window gorio_ugpm_status
};

}  // namespace gorio
