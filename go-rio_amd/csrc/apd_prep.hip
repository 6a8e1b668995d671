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
  // DBSCAN_simple.h:36-39, 65-67: std::sqrt of a float expression (float), then double arithmetic
  float n2 = qx * qx;
  n2 = n2 + qy * qy;
  n2 = n2 + qz * qz;
  const double norm = (double)sqrtf(n2);
  const double r_seed = fabs(norm - 1) / 50 + a.eps, r_exp = (norm - 1) / 100 + a.eps;
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

}  // namespace gorio
