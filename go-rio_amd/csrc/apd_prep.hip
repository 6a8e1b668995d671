// apd_prep.hip -- preprocessing that feeds the hot path: the radius searches of the DBSCAN cluster labelling
// (preprocessing_nodelet_ntu.cpp:518-568, DBSCAN_simple.h:28-100).  Included by apd_api.hip after apd_index.hip.
//
// DBSCAN_simple.h is an order-dependent queue (points visited in index order, first cluster to reach a point keeps it as a member,
// seed neighbours re-queued whatever their state) whose cost is entirely in its radius searches -- one per visited point, each a
// kd-tree query in the reference.  Here EVERY point's neighbourhood is found at once on the GPU, through the same exact tile search
// the registration uses, for the larger of the two radii the queue can ask for:
//   seed radius       |norm - 1| / 50 + eps    (DBSCAN_simple.h:36-40)
//   expansion radius  (norm - 1) / 100 + eps   (DBSCAN_simple.h:65-68; never larger than the seed radius)
// as a CSR adjacency (neighbour index, one flag bit for "also inside the expansion radius"); the queue itself is then replayed on
// the host over that adjacency, statement for statement, so the clusters are exactly the reference's.
// A neighbour is a point whose float squared distance (FLANN L2_Simple, un-fused) is < (float)(radius * radius), the query included.
#include <hip/hip_runtime.h>

namespace gorio {

struct RadiusArgs {
  double eps;
  int* cnt;               // [n] by ORIGINAL index: neighbours inside the seed radius
  const long long* offs;  // [n] by original index (fill pass)
  int* adj;               // CSR payload: neighbour original index | 0x80000000 when also inside the expansion radius
};

// mode 0: count; mode 1: fill.  grid: ceil(n_spad / 256), block 256.  One lane = one query in sorted order.
template <int MODE>
__global__ __launch_bounds__(256) void radius_neighbours_kernel(CloudView cloud, RadiusArgs a) {
  const SearchIndex& si = cloud.idx;
  const int n = si.n;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (blockIdx.x * 256 >= n) return;
  const int lane = threadIdx.x & 63;
  const int pq = p < n ? p : n - 1;
  const float qx = si.sx[pq], qy = si.sy[pq], qz = si.sz[pq];
  // DBSCAN_simple.h:36-39: the seed radius stores std::sqrt(float expression) in a double and continues in double; DBS:65-67: the
  // expansion radius evaluates (std::sqrt(float) - 1) / 100 entirely in FLOAT and only adds the double eps_ in double
  float n2 = qx * qx;
  n2 = n2 + qy * qy;
  n2 = n2 + qz * qz;
  const double norm = (double)sqrtf(n2);
  const float ef = (sqrtf(n2) - 1.0f) / 100.0f;
  const double r_seed = fabs(norm - 1) / 50 + a.eps, r_exp = (double)ef + a.eps;
  const float r2s = p < n ? (float)(r_seed * r_seed) : 0.0f, r2e = (float)(r_exp * r_exp);
  const float qlo[3] = {wave_min(qx), wave_min(qy), wave_min(qz)};
  const float qhi[3] = {wave_max(qx), wave_max(qy), wave_max(qz)};
  const scalar_fp tx = as_scalar(si.sx);
  const scalar_fp ty = as_scalar(si.sy);
  const scalar_fp tz = as_scalar(si.sz);
  const scalar_ip to = (scalar_ip)si.orig;
  const float4* __restrict__ tb4 = reinterpret_cast<const float4*>(si.tbox);
  const int ng = (si.n_tiles + 63) / 64;
  const float wb = wave_max(r2s);
  const int me = p < n ? si.orig[p] : 0;
  int cnt = 0;
  int* out = nullptr;
  if (MODE == 1 && p < n) out = a.adj + a.offs[me];
  for (int g = 0; g < ng; ++g) {
    const int tl = g * 64 + lane;
    float4 lo = make_float4(INFINITY, INFINITY, INFINITY, 0.f), hi = make_float4(-INFINITY, -INFINITY, -INFINITY, 0.f);
    if (tl < si.n_tiles) {
      lo = tb4[2 * (size_t)tl];
      hi = tb4[2 * (size_t)tl + 1];
    }
    unsigned long long mask = __ballot(box_box_bound(qlo, qhi, lo, hi) < wb);
    while (mask) {
      const int tlane = __builtin_ctzll(mask);
      mask &= mask - 1;
      const float bx[8] = {lane_f(lo.x, tlane), lane_f(lo.y, tlane), lane_f(lo.z, tlane), 0.f, lane_f(hi.x, tlane), lane_f(hi.y, tlane), lane_f(hi.z, tlane), 0.f};
      if (__ballot(box_bound(qx, qy, qz, bx) < r2s) == 0) continue;
      const int j0 = (g * 64 + tlane) * 32;
#pragma unroll 4
      for (int u = 0; u < 32; ++u) {
        const float d = sqdist3(qx, qy, qz, tx[j0 + u], ty[j0 + u], tz[j0 + u]);
        if (d < r2s) {  // padding points sit at 1e30: d = inf
          if (MODE == 1) out[cnt] = to[j0 + u] | (d < r2e ? (int)0x80000000 : 0);
          ++cnt;
        }
      }
    }
  }
  if (MODE == 0 && p < n) a.cnt[me] = cnt;
}

// pcl::RadiusOutlierRemoval as the preprocessing nodelet configures it (preprocessing_nodelet_ntu.cpp:163-171, 626-634; launch files:
// radius 2 m, 1 - 5 neighbours): number of points of the SAME cloud, the query included, whose float squared distance is at most the
// squared radius.  r2 = largest float whose double value is <= radius * radius (the comparison PCL makes is in double on float
// distances).  grid: ceil(n_spad / 256), block 256; cnt[] by ORIGINAL index.
__global__ __launch_bounds__(256) void radius_count_kernel(CloudView cloud, float r2, int* __restrict__ cnt_out) {
  const SearchIndex& si = cloud.idx;
  const int n = si.n;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (blockIdx.x * 256 >= n) return;
  const int lane = threadIdx.x & 63;
  const int pq = p < n ? p : n - 1;
  const float qx = si.sx[pq], qy = si.sy[pq], qz = si.sz[pq];
  const float qlo[3] = {wave_min(qx), wave_min(qy), wave_min(qz)};
  const float qhi[3] = {wave_max(qx), wave_max(qy), wave_max(qz)};
  const scalar_fp tx = as_scalar(si.sx);
  const scalar_fp ty = as_scalar(si.sy);
  const scalar_fp tz = as_scalar(si.sz);
  const float4* __restrict__ tb4 = reinterpret_cast<const float4*>(si.tbox);
  const int ng = (si.n_tiles + 63) / 64;
  int cnt = 0;
  for (int g = 0; g < ng; ++g) {
    const int tl = g * 64 + lane;
    float4 lo = make_float4(INFINITY, INFINITY, INFINITY, 0.f), hi = make_float4(-INFINITY, -INFINITY, -INFINITY, 0.f);
    if (tl < si.n_tiles) {
      lo = tb4[2 * (size_t)tl];
      hi = tb4[2 * (size_t)tl + 1];
    }
    unsigned long long mask = __ballot(box_box_bound(qlo, qhi, lo, hi) <= r2);
    while (mask) {
      const int tlane = __builtin_ctzll(mask);
      mask &= mask - 1;
      const float bx[8] = {lane_f(lo.x, tlane), lane_f(lo.y, tlane), lane_f(lo.z, tlane), 0.f, lane_f(hi.x, tlane), lane_f(hi.y, tlane), lane_f(hi.z, tlane), 0.f};
      if (__ballot(box_bound(qx, qy, qz, bx) <= r2) == 0) continue;
      const int j0 = (g * 64 + tlane) * 32;
#pragma unroll 4
      for (int u = 0; u < 32; ++u) cnt += sqdist3(qx, qy, qz, tx[j0 + u], ty[j0 + u], tz[j0 + u]) <= r2 ? 1 : 0;  // padding points: d = inf
    }
  }
  if (p < n) cnt_out[si.orig[p]] = cnt;
}


// pcl::StatisticalOutlierRemoval, first half (PCL 1.10 filters/impl/statistical_outlier_removal.hpp, as preprocessing_nodelet_ntu.cpp:
// 153-162 configures it -- the nodelet's DEFAULT outlier filter, mean_k 20 / 30): for every point the mean of the distances to its
// mean_k nearest neighbours, the point itself (the first of the mean_k + 1 results of nearestKSearch) left out.  The k = mean_k + 1
// smallest float squared distances are kept as in knn_kth_kernel -- an ascending register list, a candidate enters by
// D[t] = med3(D[t-1], c, D[t]) -- with ONE list length for every k <= 32: the 32 - k slots below the list proper hold -inf, which no
// candidate moves, so D[31] is the k-th smallest throughout.  The sum runs over the sorted distances 1 .. k-1 in double, on double
// square roots of the float values, and is divided by mean_k and rounded to float as PCL does.
// grid: ceil(n_spad / 256), block 256; mean_out[] by ORIGINAL index.
__global__ __launch_bounds__(256) void sor_mean_distance_kernel(CloudView cloud, int k, float* __restrict__ mean_out) {
  constexpr int K = 32;
  const SearchIndex& si = cloud.idx;
  const int n = si.n;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (blockIdx.x * 256 >= n) return;
  const int lane = threadIdx.x & 63;
  const int pq = p < n ? p : n - 1;
  const float qx = si.sx[pq], qy = si.sy[pq], qz = si.sz[pq];
  const float qlo[3] = {wave_min(qx), wave_min(qy), wave_min(qz)};
  const float qhi[3] = {wave_max(qx), wave_max(qy), wave_max(qz)};
  const scalar_fp tx = as_scalar(si.sx);
  const scalar_fp ty = as_scalar(si.sy);
  const scalar_fp tz = as_scalar(si.sz);
  const float4* __restrict__ tb4 = reinterpret_cast<const float4*>(si.tbox);
  const int ng = (si.n_tiles + 63) / 64;
  const int g0 = (blockIdx.x * 256 + (threadIdx.x & ~63)) / 32 / 64;  // the group of this wave's own tiles: visited first, it tightens the bounds
  float D[K];
#pragma unroll
  for (int t = 0; t < K; ++t) D[t] = t < K - k ? -INFINITY : INFINITY;
  for (int v = 0; v < 2 * ng; ++v) {
    const int off = (v + 1) >> 1;
    const int g = (v & 1) ? g0 - off : g0 + off;
    if (g < 0 || g >= ng) continue;
    const int tl = g * 64 + lane;
    float4 lo = make_float4(INFINITY, INFINITY, INFINITY, 0.f), hi = make_float4(-INFINITY, -INFINITY, -INFINITY, 0.f);
    if (tl < si.n_tiles) {
      lo = tb4[2 * (size_t)tl];
      hi = tb4[2 * (size_t)tl + 1];
    }
    const float wb = wave_max(D[K - 1]);
    unsigned long long mask = __ballot(box_box_bound(qlo, qhi, lo, hi) < wb);  // strictly below: an equal distance changes no distance list
    while (mask) {
      const int tlane = __builtin_ctzll(mask);
      mask &= mask - 1;
      const float bx[8] = {lane_f(lo.x, tlane), lane_f(lo.y, tlane), lane_f(lo.z, tlane), 0.f, lane_f(hi.x, tlane), lane_f(hi.y, tlane), lane_f(hi.z, tlane), 0.f};
      if (__ballot(box_bound(qx, qy, qz, bx) < D[K - 1]) == 0) continue;
      const int j0 = (g * 64 + tlane) * 32;
#pragma unroll 2
      for (int u = 0; u < 32; ++u) {
        const float c = sqdist3(qx, qy, qz, tx[j0 + u], ty[j0 + u], tz[j0 + u]);  // padding points sit at 1e30: c = inf
        if (__ballot(c < D[K - 1]) == 0) continue;
#pragma unroll
        for (int t = K - 1; t > 0; --t) D[t] = __builtin_amdgcn_fmed3f(D[t - 1], c, D[t]);
        D[0] = __builtin_amdgcn_fmed3f(D[0], c, -INFINITY);
      }
    }
  }
  if (p < n) {
    double sum = 0.0;
#pragma unroll
    for (int t = 1; t < K; ++t)  // slot K - k is the query itself (distance 0, or the nearest of its duplicates)
      if (t > K - k) sum += sqrt((double)D[t]);
    mean_out[si.orig[p]] = (float)(sum / (double)(k - 1));
  }
}

}  // namespace gorio

// ----------------------------------------------------------------------------------------------- REVE Doppler ego-velocity
// REVE = /root/reference/4DRadarSLAM/src/radar_ego_velocity_estimator.cpp.  The per-target work of estimate() (REVE:75-90: range,
// azimuth / elevation gates, unit direction, corrected Doppler), of every RANSAC hypothesis (REVE:203-214: |y - H v| against the
// inlier threshold for ALL targets) and of the final least squares (REVE:252-290: H^T H, H^T y, e^T e) are data parallel and run
// here; the 3 x 3 solves, the order statistic and the bookkeeping of the best hypothesis are a few hundred flops on the host.
namespace gorio {

struct ReveCfg {
  double min_dist, max_dist, min_db, az_lim, el_lim, doppler_factor_unused;
  float doppler_factor;
  float pad_;
};

// grid: ceil(n / 256).  f[i][4] = x/r, y/r, z/r, corrected doppler; valid[i]
__global__ __launch_bounds__(256) void reve_features_kernel(const float* __restrict__ xyz, const float* __restrict__ inten, const float* __restrict__ dop, int stride, int n, ReveCfg c,
                                                            double* __restrict__ f, unsigned char* __restrict__ valid) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float x = xyz[(size_t)i * stride], y = xyz[(size_t)i * stride + 1], z = xyz[(size_t)i * stride + 2];
  const double r = sqrt((double)x * (double)x + (double)y * (double)y + (double)z * (double)z);  // Vector3(x, y, z).norm(), REVE:78
  const double azimuth = (double)(float)atan2((double)y, (double)x);                           // atan2(float, float), REVE:80
  float rxy2 = x * x;
  rxy2 = rxy2 + y * y;
  const float rxy = (float)sqrt((double)rxy2);
  const double elevation = (double)(float)atan2((double)rxy, (double)z) - 1.57079632679489661923;  // REVE:81
  const bool ok = r > c.min_dist && r < c.max_dist && (double)inten[(size_t)i * stride] > c.min_db && fabs(azimuth) < c.az_lim && fabs(elevation) < c.el_lim;
  valid[i] = ok ? 1 : 0;
  const float d = -dop[(size_t)i * stride] * c.doppler_factor;  // float product, REVE:87
  f[4 * (size_t)i] = x / r;
  f[4 * (size_t)i + 1] = y / r;
  f[4 * (size_t)i + 2] = z / r;
  f[4 * (size_t)i + 3] = (double)d;
}

// grid: (ceil(m / 256), hypotheses).  flags[k][j] = |y_j - h_j . v_k| < thresh (REVE:203-214) over the VALID targets
__global__ __launch_bounds__(256) void reve_eval_kernel(const double* __restrict__ f, int m, const double* __restrict__ v /* [K][3] */, double thresh, unsigned char* __restrict__ flags) {
  const int j = blockIdx.x * 256 + threadIdx.x, k = blockIdx.y;
  if (j >= m) return;
  const double* r = f + 4 * (size_t)j;
  const double err = fabs(r[3] - (r[0] * v[3 * k] + r[1] * v[3 * k + 1] + r[2] * v[3 * k + 2]));
  flags[(size_t)k * m + j] = err < thresh ? 1 : 0;
}

// grid: ceil(m / 256).  per block: 10 sums over the selected rows -- H^T H (6 unique), H^T y (3), and, with v, e^T e of e = H v - y
__global__ __launch_bounds__(256) void reve_sums_kernel(const double* __restrict__ f, int m, const unsigned char* __restrict__ sel, const double* __restrict__ v, double* __restrict__ out) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  double a[10];
#pragma unroll
  for (int q = 0; q < 10; ++q) a[q] = 0.0;
  if (j < m && sel[j]) {
    const double* r = f + 4 * (size_t)j;
    a[0] = r[0] * r[0]; a[1] = r[0] * r[1]; a[2] = r[0] * r[2]; a[3] = r[1] * r[1]; a[4] = r[1] * r[2]; a[5] = r[2] * r[2];
    a[6] = r[0] * r[3]; a[7] = r[1] * r[3]; a[8] = r[2] * r[3];
    if (v) {
      const double e = (r[0] * v[0] + r[1] * v[1] + r[2] * v[2]) - r[3];
      a[9] = e * e;
    }
  }
  __shared__ double red[4][10];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < 10; ++q) {
    const double s = wave_sum(a[q]);
    if (lane == 0) red[wv][q] = s;
  }
  __syncthreads();
  if (threadIdx.x < 10) out[(size_t)blockIdx.x * 10 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

}  // namespace gorio
