/*
 * gorio_prep.h -- C ABI of the preprocessing steps that FEED the hot path, on the MI355X (libgorio_amd.so).
 *
 * Paths relative to /root/reference/4DRadarSLAM:
 *   PREP = apps/preprocessing_nodelet_ntu.cpp      DBS = include/dbscan/DBSCAN_simple.h, DBSCAN_kdtree.h
 *
 * Plain pointers and sizes; host pointers are caller-owned and only read / written during the call; 0 on success or a negative
 * gorio_status (include/gorio_apd.h); gorio_prep_last_error() gives the text (thread-local).  No CPU fallback: without a HIP device
 * the calls fail with GORIO_ERR_NO_DEVICE.
 */
#ifndef GORIO_PREP_H
#define GORIO_PREP_H

#ifdef __cplusplus
extern "C" {
#endif

/*
 * The cluster labels the preprocessing nodelet writes into PointXYZINormal::normal_x (PREP:518-568) and APD-GICP later compares
 * (fast_apdgicp_impl.hpp:271-273): DBSCANKdtreeCluster over the whole scan -- setCorePointMinPts(10), setClusterTolerance(0.9),
 * setMinClusterSize(20), setMaxClusterSize(25000) in the nodelet (PREP:523-526) -- then the clusters ranked by the distance of their
 * centroid from the sensor and label = rank + 1 (PREP:533-568); 0 for points in no cluster.
 *   xyz / point_stride_bytes          first x of the scan, bytes between points (48 for pcl::PointXYZINormal)
 *   label_out / label_stride_bytes    first normal_x to write, bytes between labels (the same 48 when written into the cloud itself)
 * Every radius search of DBS:28-100 runs on the GPU (all points at once, exact); the order-dependent queue of DBS is replayed over the
 * resulting adjacency on the host, so the clusters are those of the reference's sequential algorithm.
 */
int gorio_prep_dbscan_labels(int device, const float* xyz, int n, int point_stride_bytes, double eps, int core_min_pts, int min_cluster_size, int max_cluster_size,
                             float* label_out, int label_stride_bytes, int* n_clusters);

/*
 * pcl::RadiusOutlierRemoval as preprocessing_nodelet_ntu.cpp:163-171, 626-634 configures it (launch files: radius_radius 2,
 * radius_min_neighbors 1 - 5): keep[i] = 1 when MORE than min_neighbors points of the cloud, the point itself included, lie within
 * `radius` of point i (float squared distance compared with radius^2 in double, as PCL 1.10 does on its nearestKSearch results).
 * The caller compacts the cloud with the mask (pcl::Filter::filter keeps the surviving points in their original order).
 */
int gorio_prep_radius_outlier_mask(int device, const float* xyz, int n, int point_stride_bytes, double radius, int min_neighbors, unsigned char* keep, int* n_kept);

/*
 * pcl::StatisticalOutlierRemoval as preprocessing_nodelet_ntu.cpp:153-162, 626-634 configures it -- the nodelet's DEFAULT outlier filter
 * (statistical_mean_k 20, statistical_stddev 1.0; the launch files carry 30 / 1.2 beside their RADIUS choice): for every point the mean
 * distance to its mean_k nearest neighbours (PCL 1.10 filters/impl/statistical_outlier_removal.hpp: nearestKSearch with mean_k + 1, the
 * first result -- the point itself -- left out, double sum of the square roots of the float squared distances, divided by mean_k, kept
 * as float), then keep[i] = 1 when that value is at most mean + stddev_mul * stddev over all points (double sums in point order,
 * variance with n - 1).  mean_k in [1, 31], n > mean_k.  mean_dist_out (n floats, may be NULL) receives the per-point values.
 * PCL / FLANN are not in the image: the restatement this is tested against follows the source as recalled (parity unpinned).
 */
int gorio_prep_statistical_outlier_mask(int device, const float* xyz, int n, int point_stride_bytes, int mean_k, double stddev_mul, unsigned char* keep, int* n_kept,
                                        float* mean_dist_out);

/*
 * pcl::VoxelGrid as the preprocessing nodelet applies it to every scan (preprocessing_nodelet_ntu.cpp:137-139, 608-622; launch files:
 * downsample_method VOXELGRID, downsample_resolution 0.1): one output point per occupied voxel = the centroid of its points, output
 * ordered by voxel index as PCL's sorted index vector yields it.  Only the coordinates are produced: they are all the hot path reads
 * at this stage (normal_x is written later, by the DBSCAN step).  Runs the device voxel grid of gorio_apd_set_target_submap.
 * n_out receives the number of voxels even when out_capacity is too small.
 */
int gorio_prep_voxel_downsample(int device, const float* xyz, int n, int point_stride_bytes, double leaf, float* xyz_out, int out_stride_bytes, int out_capacity, int* n_out);

/*
 * REVE Doppler ego-velocity (REVE = src/radar_ego_velocity_estimator.cpp, REVEH = include/radar_ego_velocity_estimator.h):
 * RadarEgoVelocityEstimator::estimate (REVE:60-170) -- per-target gates, zero-velocity test, 3-D least squares with RANSAC
 * (REVE:172-250, 252-303).  The velocities it produces are the `vel` samples of the GP pre-integration windows
 * (apps/radar_graph_slam_nodelet.cpp:274-280, 481-495).
 * The reference draws its RANSAC samples with std::shuffle seeded from std::random_device (REVE:186-193); the caller of this ABI
 * draws them (any RNG) and passes what idx[0 .. N_ransac_points) holds after each shuffle: sample_idx[n_iter][n_ransac_points],
 * indices into the list of VALID targets (those that pass the gates of REVE:83-85, in input order).  A caller that needs the count of
 * valid targets first can call once with n_iter = 0 and read *n_valid.
 *   xyz / intensity / doppler   first x, first intensity, first doppler of the scan; stride_bytes between targets
 *   inlier_mask / outlier_mask  n bytes each (may be NULL): the clouds estimate() publishes (REVE:130-137)
 */
typedef struct {
  float min_dist, max_dist, min_db, elevation_thresh_deg, azimuth_thresh_deg, doppler_velocity_correction_factor;        /* REVEH:32-37 */
  float thresh_zero_velocity, allowed_outlier_percentage, sigma_zero_velocity_x, sigma_zero_velocity_y, sigma_zero_velocity_z;  /* :39-43 */
  float sigma_offset_radar_x, sigma_offset_radar_y, sigma_offset_radar_z, max_sigma_x, max_sigma_y, max_sigma_z;         /* :45-51 */
  float inlier_thresh;           /* :59 */
  int use_ransac;                /* :55 */
  int n_ransac_points;           /* :58 */
  float outlier_prob, success_prob;  /* :56-57, only used by gorio_prep_reve_ransac_iterations */
} gorio_reve_config;
void gorio_prep_reve_default_config(gorio_reve_config* c);
int gorio_prep_reve_ransac_iterations(const gorio_reve_config* c); /* setRansacIter, REVEH:138-141 (3 with the defaults) */
int gorio_prep_ego_velocity(int device, const float* xyz, const float* intensity, const float* doppler, int n, int stride_bytes, const gorio_reve_config* cfg,
                            const unsigned int* sample_idx, int n_iter, double v_r[3], double sigma_v_r[3], unsigned char* inlier_mask, unsigned char* outlier_mask,
                            int* n_valid, int* zero_velocity, int* success);

const char* gorio_prep_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* GORIO_PREP_H */
