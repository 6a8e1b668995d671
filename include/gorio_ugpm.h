/*
 * gorio_ugpm.h -- C ABI of the MI355X-native UGPM GP pre-integration back end (libgorio_amd.so).
 *
 * The reference's pre-integration is a header-only C++ class, ugpm::VelPreintegration (4DRadarSLAM/include/VelInt/preint.h:22-82),
 * constructed once per keyframe by the back end (4DRadarSLAM/apps/radar_graph_slam_nodelet.cpp:497-513).  There is no FFI in the
 * reference; this header is the boundary the drop-in class go-rio_amd/host/VelInt/preint.h uses underneath the same class surface.
 * Paths below are relative to /root/reference/4DRadarSLAM:
 *   PRE   = include/VelInt/preint.h        TYPES = include/VelInt/types.h
 *   MATH  = include/VelInt/math_utils.h    COST  = include/VelInt/cost_functions.h    RGS = apps/radar_graph_slam_nodelet.cpp
 *
 * One call pre-integrates a BATCH of independent windows (one workgroup set per window, all windows in one launch set);
 * plain pointers and sizes only; host pointers are caller-owned and only read during the call; results are written to `out`.
 * Return 0 or a negative gorio_ugpm_status; gorio_ugpm_last_error() gives the text (thread-local).  No CPU fallback.
 */
#ifndef GORIO_UGPM_H
#define GORIO_UGPM_H

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  GORIO_UGPM_OK = 0,
  GORIO_UGPM_ERR_INVALID = -1,     /* null pointers, empty batch ... */
  GORIO_UGPM_ERR_NO_DEVICE = -2,   /* no usable HIP device / HIP runtime error */
  GORIO_UGPM_ERR_RANGE = -3,       /* what the reference reports with std::range_error (MATH:493, PRE:680-686, TYPES:415) */
  GORIO_UGPM_ERR_ARGUMENT = -4,    /* std::invalid_argument of GyroVelData::get (TYPES:160) */
  GORIO_UGPM_ERR_UNSUPPORTED = -5, /* more than 160 GP states in one (chunk) window */
  GORIO_UGPM_ERR_NUMERIC = -6      /* a Cholesky factorisation met a non-positive pivot */
} gorio_ugpm_status;

/* ugpm::PreintType (TYPES:15) */
typedef enum { GORIO_UGPM_TYPE_LPM = 0, GORIO_UGPM_TYPE_UGPM = 1 } gorio_ugpm_type;

/*
 * One pre-integration request == one ugpm::VelPreintegration construction (PRE:1517-1581):
 *   GyroVelData{gyr, vel, gyr_var, vel_var} (TYPES:74-224), start_t, infer_t, PreintOption (TYPES:285-292), PreintPrior (TYPES:294-298)
 * plus the two bias standard deviations of VelPreintegration::get (PRE:55; the back end passes 0, 0 at RGS:513).
 * gyr / vel: n x 3 doubles, sample-major (DataSample::data, TYPES:67-71); times ascending.
 */
typedef struct {
  const double* gyr_t;
  const double* gyr;
  int n_gyr;
  const double* vel_t;
  const double* vel;
  int n_vel;
  double gyr_var;     /* RGS:476: 1.74532925e-3 */
  double vel_var;     /* RGS:493: 1e-6 */
  double start_t;
  const double* infer_t; /* query times; the nodelet passes exactly one (RGS:503-508) */
  int n_infer;
  int type;           /* gorio_ugpm_type; default UGPM (TYPES:288).  LPM = IterativeIntegrator as the output method (PRE:1567-1580) over ALL samples given */
  double min_freq;    /* PreintOption::min_freq, default 500 (TYPES:287); the internal LPM passes always use 500 (PRE:1201) */
  double quantum;     /* PreintOption::quantum, default -1 (no chunks).  > 0: chunked mode (PRE:1584-1702): the request is cut into
                         chunks of `quantum` seconds, every chunk is pre-integrated as a window of its own (all chunks of all requests
                         in the same device batch) and the chunk results are chained with combinePreints (MATH:689-726).  0 is refused
                         (the reference divides by it).  TYPES:36 declares Vec12 with nine rows, so the reference's covariance
                         propagation (MATH:540-574) is undefined behaviour; the twelve components it addresses are used here */
  double state_freq;  /* PreintOption::state_freq, default 50 (TYPES:290) */
  int correlate;      /* PreintOption::correlate, default true (TYPES:291) */
  int overlap;        /* kOverlap = 8 (PRE:19) */
  double gyr_bias[3]; /* PreintPrior (TYPES:294-298) */
  double vel_bias[3];
  double vel_bias_std; /* arguments of get(): 0.3 / 0.03 by default (PRE:55), 0 / 0 from the nodelet */
  double gyr_bias_std;
  /* The reference's first constructor takes infer_t as vector<vector<double>> (PRE:1517-1523); here the inner vectors are laid end to
   * end in infer_t and group_sizes[n_groups] gives their lengths (NULL / 0 = one vector of n_infer stamps).  UGPM evaluates every
   * stamp on its own, so grouping changes nothing there.  For type = LPM it reproduces a detail of the reference: the rotation part
   * of record j of a group is that of the group's j-th SMALLEST stamp (SortIndexTracker2::getVector, TYPES:378-387, PRE:259) while
   * the position part is written by original index (PRE:640-664) -- identical whenever each inner vector is ascending. */
  const int* group_sizes;
  int n_groups;
} gorio_ugpm_window;

/* ugpm::PreintMeas (TYPES:236-281); matrices ROW-major.  83 doubles. */
typedef struct {
  double delta_R[9];
  double delta_p[3];
  double dt;
  double dt_sq_half;
  double cov[36];
  double d_delta_R_d_bw[9];
  double d_delta_R_d_t[3];
  double d_delta_p_d_bw[9];
  double d_delta_p_d_bv[9];
  double d_delta_p_d_t[3];
} gorio_ugpm_meas;

/* optional per-window diagnostics */
typedef struct {
  int nb_state;   /* S (PRE:775) */
  int nb_gyr;     /* gyro samples inside the padded window (PRE:789-792) */
  int nb_vel;
  int iters_rot;  /* LM iterations of the two GP fits (PRE:952, 967) */
  int iters_vel;
  int status;     /* per-window gorio_ugpm_status */
  double cost_rot;
  double cost_vel;
  double state_freq; /* effective state frequency (PRE:766-771) */
} gorio_ugpm_diag;

void gorio_ugpm_default_window(gorio_ugpm_window* w); /* PreintOption / PreintPrior defaults, kOverlap */

/*
 * n_windows constructions + get(0, j, vel_bias_std, gyr_bias_std) for every j < n_infer of every window.
 * out: sum of n_infer records, window-major in the order given.  diag: n_windows records or NULL.
 * A per-window failure (bad data) makes the call return that window's error code after all other windows were processed;
 * its records are filled with NaN.  diag of a chunked request: sizes and state frequency of its last chunk, iterations and costs
 * summed over its chunks.
 */
int gorio_ugpm_preint_batch(const gorio_ugpm_window* windows, int n_windows, gorio_ugpm_meas* out, gorio_ugpm_diag* diag, int device);

/* combinePreints(prev, cur) (MATH:689-726): the pre-integrated measurement of two consecutive intervals; cur.dt == 0 returns prev. */
int gorio_ugpm_combine_preints(const gorio_ugpm_meas* prev, const gorio_ugpm_meas* cur, gorio_ugpm_meas* out);

const char* gorio_ugpm_last_error(void);

/* seconds spent in device kernels of the last batch on this thread, by stage: [0] LPM initialisation (or the whole LPM output path),
 * [1] Gram / inverse / cross-kernel products, [2] state correlation, [3] LM fits, [4] inference, [5] the J^T J launches of the LM fits
 * (a part of [3]), [6] the J^T J launch of the correlation (a part of [2]), [7] reserved; and launch-set counts. Either may be NULL. */
int gorio_ugpm_get_stage_times(double seconds[8], int counts[8]);

/* Test hook, process-wide: speculative_rot = 0 runs the rotation fit (PRE:943-952) as four launches per iteration (step, candidate
 * residual, acceptance + Jacobian, J^T J) instead of three (step, candidate residual + Jacobian, acceptance + J^T J).  The two
 * schedules take the same steps; tests compare them.  Default 1. */
void gorio_ugpm_debug_set_schedule(int speculative_rot);

#ifdef __cplusplus
}
#endif
#endif /* GORIO_UGPM_H */
