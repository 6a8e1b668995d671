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

const char* gorio_prep_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* GORIO_PREP_H */
