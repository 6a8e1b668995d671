/*
 * gorio_apd.h -- C ABI of the MI355X-native APD-GICP scan-matching back end (libgorio_amd.so).
 *
 * The reference has no FFI for this path: the seam is C++ virtual dispatch through
 * pcl::Registration<PointXYZINormal,PointXYZINormal>.  This header is the boundary a drop-in
 * fast_gicp::FastAPDGICP uses underneath that class surface (go-rio_amd/host/fast_gicp/gicp/fast_apdgicp.hpp in this
 * repository); every entry point cites the reference member it replaces.  Paths are relative to /root/reference:
 *   APDH = fast_apdgicp/include/fast_gicp/gicp/fast_apdgicp.hpp
 *   APD  = fast_apdgicp/include/fast_gicp/gicp/impl/fast_apdgicp_impl.hpp
 *   LSQH = fast_apdgicp/include/fast_gicp/gicp/lsq_registration.hpp
 *   LSQ  = fast_apdgicp/include/fast_gicp/gicp/impl/lsq_registration_impl.hpp
 *   REG  = 4DRadarSLAM/src/radar_graph_slam/registrations.cpp
 *
 * Conventions
 *  - plain pointers and sizes only; all host pointers are caller-owned and only read / written during the call.
 *  - 4x4 matrices are ROW-major (T[r*4+c]); pcl / Eigen matrices are column-major, the C++ shim transposes.
 *  - covariances cross the ABI as n * 16 doubles (Eigen::Matrix4d per point, symmetric, row/col 3 zero), exactly the
 *    element type of FastAPDGICP::source_covs_ / target_covs_ (APDH:109-110).
 *  - return 0 on success, negative gorio_status otherwise; gorio_apd_last_error() gives the text.
 *  - a handle is one FastAPDGICP object: not thread-safe, distinct handles are independent (SURVEY 8b "Threading").
 *  - there is no CPU fallback: every compute entry point fails with GORIO_ERR_NO_DEVICE when no HIP device is usable.
 */
#ifndef GORIO_APD_H
#define GORIO_APD_H

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  GORIO_OK = 0,
  GORIO_ERR_INVALID = -1,    /* bad argument (null pointer, n < k, ...) */
  GORIO_ERR_NO_DEVICE = -2,  /* no usable HIP device / HIP runtime error */
  GORIO_ERR_STATE = -3,      /* call order violated (align before setInputTarget, ...) */
  GORIO_ERR_ALLOC = -4,
  GORIO_ERR_UNSUPPORTED = -5 /* e.g. k_correspondences > 32, unknown regularisation (APD:389-391 aborts there) */
} gorio_status;

/* fast_gicp::RegularizationMethod, gicp_settings.hpp:6 (same order) */
typedef enum { GORIO_REG_NONE = 0, GORIO_REG_MIN_EIG = 1, GORIO_REG_NORMALIZED_MIN_EIG = 2, GORIO_REG_PLANE = 3, GORIO_REG_FROBENIUS = 4 } gorio_regularization;
/* fast_gicp::LSQ_OPTIMIZER_TYPE, lsq_registration.hpp:13 (same order) */
typedef enum { GORIO_OPT_GAUSS_NEWTON = 0, GORIO_OPT_LEVENBERG_MARQUARDT = 1 } gorio_optimizer;
/* correspondence search strategy; every mode returns the SAME indices (exact search, ties -> lowest index) */
typedef enum { GORIO_SEARCH_BRUTE_FORCE = 0, GORIO_SEARCH_PRUNED = 1 } gorio_search;

typedef struct {
  int k_correspondences;         /* setCorrespondenceRandomness, APD:45; default 20 (APD:21); <= 32 */
  int regularization;            /* setRegularizationMethod, APD:50; default PLANE (APD:25) */
  double dist_var;               /* setDistVar APD:63; default 0.86 (APDH:118) */
  double azimuth_var;            /* setAzimuthVar APD:55; default 0.5 deg (APDH:116) */
  double elevation_var;          /* setElevationVar APD:59; default 1.0 deg (APDH:117) */
  double corr_dist_threshold;    /* pcl setMaxCorrespondenceDistance (REG:44); default FLT_MAX (APD:23) */
  int max_iterations;            /* pcl setMaximumIterations (REG:43); default 64 (LSQ:13) */
  double rotation_epsilon;       /* setRotationEpsilon LSQ:30; default 2e-3 (LSQ:14) */
  double transformation_epsilon; /* pcl setTransformationEpsilon (REG:42); default 5e-4 (LSQ:15) */
  int optimizer;                 /* lsq_optimizer_type_, default LM (LSQ:17) */
  int lm_max_iterations;         /* default 10 (LSQ:19) */
  double lm_init_lambda_factor;  /* setInitialLambdaFactor LSQ:35; default 1e-9 (LSQ:20) */
  int search;                    /* gorio_search; default brute force */
  int cl_weight_points;          /* N in cl_weight = 1/N (APD:273: correspondences_.size()); 0 = this handle's source size.  Set it to
                                    the global source size when the source cloud is sharded over several handles / GPUs */
  int keep_knn_indices;          /* parity hook: keep the k neighbour indices of every point for gorio_apd_get_knn_indices (80 B per point
                                    of extra stores in the covariance kernel); default 0 */
} gorio_apd_params;

typedef struct gorio_apd gorio_apd_t;

/* FastAPDGICP::FastAPDGICP() APD:14-28 / ~FastAPDGICP APD:31.  device = HIP device ordinal. */
int gorio_apd_create(gorio_apd_t** out, int device);
void gorio_apd_destroy(gorio_apd_t* h);
const char* gorio_apd_last_error(const gorio_apd_t* h);

/* library defaults == constructor defaults of the reference (APD:14-28, LSQ:10-24, APDH:116-118) */
void gorio_apd_default_params(gorio_apd_params* p);
/* all setters of APD:34-65, LSQ:30-42 and the pcl::Registration setters used at REG:42-44, in one call */
int gorio_apd_set_params(gorio_apd_t* h, const gorio_apd_params* p);
int gorio_apd_get_params(const gorio_apd_t* h, gorio_apd_params* p);

/*
 * setInputSource APD:115-124 / setInputTarget APD:127-135.  xyz points at the first x; consecutive points are
 * `point_stride_bytes` apart (48 for pcl::PointXYZINormal, 12 for packed xyz); label points at the first cluster label
 * (PointXYZINormal::normal_x, written by preprocessing_nodelet_ntu.cpp:561-568) with the same stride, or NULL (all 0).
 * Copies the cloud to the device as SoA and invalidates that cloud's covariances (APD:122, 133).  The pointer-equality
 * early-out of APD:116-118 / 128-130 is the C++ shim's job (it owns the shared_ptrs).
 */
int gorio_apd_set_source(gorio_apd_t* h, const float* xyz, const float* label, int n, int point_stride_bytes);
int gorio_apd_set_target(gorio_apd_t* h, const float* xyz, const float* label, int n, int point_stride_bytes);
/* Same, from DEVICE-resident SoA buffers (x, y, z, label each n floats; label may be NULL): device-to-device copy. */
int gorio_apd_set_source_device(gorio_apd_t* h, const float* d_x, const float* d_y, const float* d_z, const float* d_label, int n);
int gorio_apd_set_target_device(gorio_apd_t* h, const float* d_x, const float* d_y, const float* d_z, const float* d_label, int n);

/* Batched form of the two calls above for `count` handles on one device: ONE copy launch for all clouds.  source / target: arrays
 * of `count` descriptors, or NULL to leave that side of every handle untouched.  Same semantics as the single calls: the clouds are
 * copied (device-to-device, on the library's stream) and the covariances of every touched cloud are invalidated. */
typedef struct {
  const float* x;
  const float* y;
  const float* z;
  const float* label; /* may be NULL */
  int n;
} gorio_apd_device_cloud;
int gorio_apd_set_clouds_device_batch(gorio_apd_t** handles, int count, const gorio_apd_device_cloud* source, const gorio_apd_device_cloud* target);

/* Batch extension (no counterpart in the reference, where every FastAPDGICP object builds its own kd-tree and covariances even when
 * two objects are handed the same cloud pointer): h's target becomes THE SAME device-resident cloud as owner's current target --
 * points, covariances and search index exist once per GPU however many handles register against them (BASELINE config C5: 512 scan
 * pairs against one 1 M-point map: one upload, one index build and one k-NN pass instead of 512).  Both handles must live on one
 * device.  Results are identical to giving h a private copy PROVIDED every sharer estimates covariances with the same k_correspondences
 * and regularization (the reference estimates them per object with its own settings, APD:149-154; here the shared cloud carries one set):
 * sharing with, or aligning / linearizing on, a handle whose two parameters differ from the ones the shared covariances were estimated with
 * fails with GORIO_ERR_INVALID (covariances supplied through gorio_apd_set_target_covariances carry no parameters and suit every
 * sharer).  The link is by value of the moment: a later setInputTarget / clearTarget on either handle detaches that handle only.
 * gorio_apd_set_target_covariances on a shared target is seen by every sharer. */
int gorio_apd_set_target_shared(gorio_apd_t* h, gorio_apd_t* owner);

/*
 * Scan-to-submap target assembly (scan_matching_odometry_nodelet.cpp:602-618, "SMO"): what the nodelet does on the CPU before
 * registration_s2m->setInputTarget -- every keyframe cloud moved by rel_pose = odom_i^-1 * odom_newest (pcl::transformPointCloud with
 * a double matrix, SMO:606-608), concatenated in the order given (SMO:609), passed through downsample() (SMO:611, 405-415) -- here on
 * the GPU, straight into the handle's device-resident target (the covariances of the new target are stale, as after setInputTarget).
 *   voxel_leaf <= 0   downsample_method NONE of the shipped launch files: pcl::PassThrough, i.e. only non-finite points are dropped
 *   voxel_leaf  > 0   pcl::VoxelGrid with that leaf size, downsample_all_data (SMO:145-149): one centroid per occupied voxel, in
 *                     ascending voxel index; the label of a voxel is the NORMALISED sum of its points' labels (sign), as
 *                     AccumulatorNormal leaves normal_x
 * n_target (may be NULL) receives the number of points of the assembled target; gorio_apd_get_target_points reads them back (xyz and
 * label strided like set_target's input; label_out may be NULL).
 */
typedef struct {
  const float* xyz;        /* first x of the keyframe cloud (host) */
  const float* label;      /* first normal_x, same stride, or NULL */
  int n;
  int point_stride_bytes;  /* 48 for pcl::PointXYZINormal */
  const double* rel_pose;  /* 16 doubles, ROW-major 4x4 */
} gorio_apd_keyframe;
int gorio_apd_set_target_submap(gorio_apd_t* h, const gorio_apd_keyframe* frames, int count, double voxel_leaf, int* n_target);
int gorio_apd_get_target_points(gorio_apd_t* h, float* xyz_out, float* label_out, int n, int point_stride_bytes);

/* clearSource APD:101-105, clearTarget APD:107-112, swapSourceAndTarget APD:89-98 */
int gorio_apd_clear_source(gorio_apd_t* h);
int gorio_apd_clear_target(gorio_apd_t* h);
int gorio_apd_swap_source_and_target(gorio_apd_t* h);

/* setSourceCovariances APD:138-140 / setTargetCovariances APD:143-145: n * 16 doubles.  When n differs from the size of the cloud
 * currently set (or no cloud is set) the call leaves that cloud WITHOUT covariances and returns GORIO_OK: the reference keeps such a
 * vector but recomputes the covariances at the next align because the sizes differ (APD:149-154). */
int gorio_apd_set_source_covariances(gorio_apd_t* h, const double* cov4x4, int n);
int gorio_apd_set_target_covariances(gorio_apd_t* h, const double* cov4x4, int n);
/* getSourceCovariances APDH:73-75 / getTargetCovariances APDH:77-79.  Returns the number of covariances currently held
 * (0 when stale, as source_covs_.size() would); copies min(count, n) of them when cov4x4 != NULL. */
int gorio_apd_get_source_covariances(gorio_apd_t* h, double* cov4x4, int n);
int gorio_apd_get_target_covariances(gorio_apd_t* h, double* cov4x4, int n);
/* calculate_covariances APD:351-411 for whichever cloud is stale (what computeTransformation does first, APD:149-154) */
int gorio_apd_calculate_covariances(gorio_apd_t* h);
/* parity hook: the k neighbour indices (sorted by distance, ties by index) used for cloud `which` (0 source, 1 target);
 * idx holds n*k ints.  Only valid after the covariances were computed by this library with params.keep_knn_indices set. */
int gorio_apd_get_knn_indices(gorio_apd_t* h, int which, int* idx, int n_times_k);

/*
 * computeTransformation APD:148-157 + LsqRegistration::computeTransformation LSQ:55-80 (what pcl::Registration::align
 * dispatches to).  guess / T_out: row-major float 4x4 (final_transformation_, LSQ:78); H_out: 36 doubles
 * (final_hessian_, LSQ:120/168; may be NULL); converged = hasConverged(); nr_iterations = nr_iterations_ (LSQ:68).
 * n_linearize (may be NULL) = number of linearize() calls executed = the unit of the throughput metric.
 */
int gorio_apd_align(gorio_apd_t* h, const float guess[16], float T_out[16], double* H_out, int* converged, int* nr_iterations, int* n_linearize);

/* same for `count` independent handles advanced in lock-step on one device (one launch set per iteration).
 * guesses / T_out: count * 16 floats; H_out: count * 36 doubles or NULL; the int outputs: count entries or NULL.
 * All handles must be distinct, live on one device and carry the same parameters (cl_weight_points is per handle); otherwise
 * GORIO_ERR_INVALID.  Targets may be shared between the handles (gorio_apd_set_target_shared). */
int gorio_apd_align_batch(gorio_apd_t** handles, int count, const float* guesses, float* T_out, double* H_out, int* converged, int* nr_iterations, int* n_linearize);

/* linearize APD:224-307 (== evaluateCost LSQ:50-52 with a double pose): updates correspondences + Mahalanobis matrices at
 * T (row-major double 4x4) and returns H (36, may be NULL together with b), b (6) and the weighted error. */
int gorio_apd_linearize(gorio_apd_t* h, const double T[16], double* H, double* b, double* error);
/* compute_error APD:310-346: error at T with the correspondences / Mahalanobis matrices of the last linearize */
int gorio_apd_compute_error(gorio_apd_t* h, const double T[16], double* error);
/* parity hooks: correspondences_ / sq_distances_ (APDH:113-114) and mahalanobis_ (APDH:111; n*16 doubles, row/col 3 zero;
 * entries of rejected points are zero) of the last linearize.  The matrices are materialised by gorio_apd_linearize and by a
 * Levenberg-Marquardt align (whose error trials read them); a Gauss-Newton align skips the store, after it get_mahalanobis and
 * compute_error return GORIO_ERR_STATE until the next gorio_apd_linearize. */
int gorio_apd_get_correspondences(gorio_apd_t* h, int* corr, float* sq_dist, int n);
int gorio_apd_get_mahalanobis(gorio_apd_t* h, double* maha4x4, int n);

/* pcl::transformPointCloud(*input_, output, final_transformation_) of LSQ:79 on the device-resident source:
 * xyz_out strided like set_source's input (only x, y, z are written). */
int gorio_apd_transform_source(gorio_apd_t* h, const float T[16], float* xyz_out, int n, int point_stride_bytes);

/* pcl::Registration::getFitnessScore(max_range) as the callers use it (scan_matching_odometry_nodelet.cpp:675,
 * loop_detector.cpp:411): mean squared NN distance of the source moved by T over the points whose SQUARED distance is <= max_range
 * (PCL compares the squared distance with max_range), DBL_MAX when no point qualifies.  When inlier_fraction != NULL it also
 * receives the inlier fraction of publish_scan_matching_status (scan_matching_odometry_nodelet.cpp:677-689): the share of source
 * points whose squared NN distance is < inlier_dist * inlier_dist.  The nodelet hard-codes max_correspondence_dist = 0.5 m there
 * (SMO:677); pass inlier_dist <= 0 to get exactly that.  Neither statistic depends on corr_dist_threshold.  A handle that searches only
 * its rank's share of the source (gorio_apd_comm_init / gorio_apd_debug_set_shard with more than one rank) returns GORIO_ERR_STATE: score
 * the pose on an unsharded handle. */
int gorio_apd_fitness_score(gorio_apd_t* h, const float T[16], double max_range, double inlier_dist, double* score, double* inlier_fraction);

/*
 * Sharded-source mode -- "one large co-registration" (no counterpart in the reference, whose parallelism is one OpenMP
 * team): several processes, one per GPU, each with a handle holding the SAME source and target clouds, form an RCCL communicator
 * (RCCL = the "nccl" of ROCm, over xGMI inside a node).  From then on gorio_apd_align / _linearize / _compute_error on those handles
 * are COLLECTIVE calls -- every rank makes them, in the same order with the same arguments.  Each rank searches and linearises only
 * its contiguous part of the source (a spatially compact run of the search order); the covariances are estimated on the whole
 * clouds by every rank, so they are exactly those of an unsharded run.  The only exchange is one in-place ncclAllReduce of 28
 * doubles (H upper triangle, b, error) per linearisation and of 1 double per Levenberg-Marquardt trial, enqueued on the launch stream
 * between the kernels; every rank then takes the same 6 x 6 step, so all ranks return the same pose without a broadcast and the
 * result equals the unsharded one up to the rounding of the reordered fp64 sums.
 *   rank 0:   gorio_comm_get_unique_id(id);  hand the 128 bytes to the other ranks by any means (MPI, torch.distributed, a file)
 *   all:      gorio_apd_comm_init(h, world_size, rank, id);  ...  gorio_apd_comm_destroy(h);
 * A handle with a communicator cannot be part of a gorio_apd_align_batch.  librccl is loaded on first use.
 */
int gorio_comm_get_unique_id(char id[128]);
int gorio_apd_comm_init(gorio_apd_t* h, int world_size, int rank, const char id[128]);
int gorio_apd_comm_destroy(gorio_apd_t* h);
/* What RCCL reports about the handle's communicator (ncclCommCount, ncclCommUserRank) and how many ncclAllReduce calls the handle has
 * enqueued on it since comm_init: evidence for benchmarks and tests that the collective path ran across that many ranks.  Any output
 * pointer may be NULL.  GORIO_ERR_STATE without a communicator. */
int gorio_apd_comm_info(gorio_apd_t* h, int* world_size, int* rank, long long* allreduce_count);
/* Test hook: the source partition of rank `rank` of `world_size` WITHOUT a communicator -- gorio_apd_linearize / _compute_error then
 * return this rank's partial sums (a test adds them up itself; two such handles can live on one GPU, which two RCCL ranks cannot).
 * world_size = 1 switches it off. */
int gorio_apd_debug_set_shard(gorio_apd_t* h, int world_size, int rank);
/* Test hooks of the two schedule optimisations of a Gauss-Newton align, so that a regression can be localised (both default to on; neither
 * changes a result):
 *   fuse_step   != 0: the optimiser step of LSQ:107-123 runs in the LAST workgroup of the linearisation launch (a fence-free hand-over
 *                     validated on gfx950); 0: it runs as its own launch after the linearisation.
 *   plan_search != 0: from the third correspondence search of an align on, query waves that were slow in the second one are cut into
 *                     parts and dispatched heaviest first; 0: every search uses the natural schedule. */
int gorio_apd_debug_set_schedule(gorio_apd_t* h, int fuse_step, int plan_search);

/* seconds spent inside device kernels of the last align / align_batch, by stage (HIP events on the launch stream):
 * [0] k-NN + covariance estimation, [1] correspondence search, [2] linearize, [3] LM/GN solve + error trials, [4] search-index build
 * (Morton sort, kd refinement, boxes; pruned mode only), [5..7] reserved (0); plus launch-set counts in counts[0..7] (either may be
 * NULL).  Filled only after gorio_apd_set_profiling(h, 1). */
int gorio_apd_set_profiling(gorio_apd_t* h, int enable);
int gorio_apd_get_stage_times(gorio_apd_t* h, double seconds[8], int counts[8]);

#ifdef __cplusplus
}
#endif
#endif /* GORIO_APD_H */
