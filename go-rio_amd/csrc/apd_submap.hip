// apd_submap.hip -- scan-to-submap target assembly on the device.  Included by apd_api.hip after apd_index.hip
// (it reuses the LDS bitonic sort kernels of the search-index build).
//
// SMO = /root/reference/4DRadarSLAM/apps/scan_matching_odometry_nodelet.cpp.  SMO:602-618: the last keyframe clouds are moved into the
// newest keyframe's frame (pcl::transformPointCloud with a double 4x4), concatenated, passed through downsample() (SMO:405-415: NONE =
// PassThrough in every shipped launch file, VOXELGRID = pcl::VoxelGrid) and become the registration target.
//   submap_transform_kernel   out = (float)(T x) in double, left to right (pcl::detail::Transformer<double>::se3), label carried along
//   submap_store_kernel       the assembled points into the target cloud's SoA + float4 buffers (+ padding)
//   vox_*                     pcl::VoxelGrid with downsample_all_data: voxel index per point, sort by (voxel, input order), one
//                             centroid per occupied voxel in ascending voxel order, float sums in input order, label = sign of the
//                             summed labels (the normalised "normal" of AccumulatorNormal; normal_y = normal_z = 0 in this pipeline)
// PCL is not under /root/reference: this follows the published sources of PCL 1.10 (the test-side CPU restatement follows the same lines).
#include <hip/hip_runtime.h>

namespace gorio {

struct SubmapFrame {
  double T[12];  // rows 0..2 of rel_pose, row-major
  int begin, end;  // this frame's points in the packed staging array
};

// grid: (ceil(max frame size / 256), frames), block 256
__global__ __launch_bounds__(256) void submap_transform_kernel(const float4* __restrict__ in, const SubmapFrame* __restrict__ frames, float4* __restrict__ out) {
  const SubmapFrame f = frames[blockIdx.y];
  const int i = f.begin + blockIdx.x * 256 + threadIdx.x;
  if (i >= f.end) return;
  const float4 p = in[i];
  const double x = (double)p.x, y = (double)p.y, z = (double)p.z;
  float4 q;
  q.x = (float)(f.T[0] * x + f.T[1] * y + f.T[2] * z + f.T[3]);
  q.y = (float)(f.T[4] * x + f.T[5] * y + f.T[6] * z + f.T[7]);
  q.z = (float)(f.T[8] * x + f.T[9] * y + f.T[10] * z + f.T[11]);
  q.w = p.w;
  out[i] = q;
}

// grid: ceil(n_pad / 256)
__global__ __launch_bounds__(256) void submap_store_kernel(const float4* __restrict__ pts, int n, int n_pad, float* __restrict__ x, float* __restrict__ y, float* __restrict__ z,
                                                           float* __restrict__ label, float4* __restrict__ p4) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_pad) return;
  const float4 v = i < n ? pts[i] : make_float4(1e30f, 1e30f, 1e30f, 0.0f);
  x[i] = v.x; y[i] = v.y; z[i] = v.z; label[i] = v.w;
  p4[i] = v;
}

__global__ void vox_bbox_init_kernel(unsigned int* __restrict__ bb) {
  if (threadIdx.x < 3) bb[threadIdx.x] = 0xffffffffu;
  else if (threadIdx.x < 6) bb[threadIdx.x] = 0u;
}
__global__ __launch_bounds__(256) void vox_bbox_kernel(const float4* __restrict__ pts, int n, unsigned int* __restrict__ bb) {
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const float4 p = pts[i];
    const float v[3] = {p.x, p.y, p.z};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      lo[a] = fminf(lo[a], v[a]);
      hi[a] = fmaxf(hi[a], v[a]);
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      lo[a] = fminf(lo[a], __shfl_down(lo[a], off, 64));
      hi[a] = fmaxf(hi[a], __shfl_down(hi[a], off, 64));
    }
    if ((threadIdx.x & 63) == 0) {
      atomicMin(bb + a, f2ord(lo[a]));
      atomicMax(bb + 3 + a, f2ord(hi[a]));
    }
  }
}

struct VoxGrid {
  float inv;       // 1 / leaf, float like pcl::VoxelGrid::inverse_leaf_size_
  int min_b[3];
  int div0, div01; // divb_mul = (1, div_b.x, div_b.x div_b.y)
};

// keys[i] = voxel index << 31 | i  (the voxel count fits int32 or the host falls back to no downsampling, like PCL)
__global__ __launch_bounds__(256) void vox_key_kernel(const float4* __restrict__ pts, int n, int npow2, VoxGrid g, unsigned long long* __restrict__ keys) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= npow2) return;
  unsigned long long key = ~0ull;
  if (i < n) {
    const float4 p = pts[i];
    const int i0 = (int)floorf(p.x * g.inv) - g.min_b[0];
    const int i1 = (int)floorf(p.y * g.inv) - g.min_b[1];
    const int i2 = (int)floorf(p.z * g.inv) - g.min_b[2];
    const unsigned long long idx = (unsigned long long)((long long)i0 + (long long)i1 * g.div0 + (long long)i2 * g.div01);
    key = (idx << 31) | (unsigned long long)i;
  }
  keys[i] = key;
}

// number of voxel starts per 256-key block -> counts[block]
__global__ __launch_bounds__(256) void vox_count_kernel(const unsigned long long* __restrict__ keys, int n, int* __restrict__ counts) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  const bool start = p < n && (p == 0 || (keys[p] >> 31) != (keys[p - 1] >> 31));
  const unsigned long long m = __ballot(start);
  __shared__ int sw[4];
  if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = __builtin_popcountll(m);
  __syncthreads();
  if (threadIdx.x == 0) counts[blockIdx.x] = sw[0] + sw[1] + sw[2] + sw[3];
}
// exclusive scan of the block counts by ONE workgroup (<= a few thousand blocks); total -> counts[nblocks]
__global__ __launch_bounds__(1024) void vox_scan_kernel(int* __restrict__ counts, int nblocks) {
  __shared__ int s[1024];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < nblocks; base += 1024) {
    const int i = base + threadIdx.x;
    const int v = i < nblocks ? counts[i] : 0;
    s[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      const int t = threadIdx.x >= off ? s[threadIdx.x - off] : 0;
      __syncthreads();
      s[threadIdx.x] += t;
      __syncthreads();
    }
    if (i < nblocks) counts[i] = carry + s[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry += s[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) counts[nblocks] = carry;
}
// one lane per voxel start: the centroid of its run of keys, written at its rank in ascending voxel order
__global__ __launch_bounds__(256) void vox_centroid_kernel(const unsigned long long* __restrict__ keys, const float4* __restrict__ pts, int n, const int* __restrict__ offsets,
                                                           float4* __restrict__ out) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  const bool start = p < n && (p == 0 || (keys[p] >> 31) != (keys[p - 1] >> 31));
  const unsigned long long m = __ballot(start);
  __shared__ int sw[4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) sw[wv] = __builtin_popcountll(m);
  __syncthreads();
  if (!start) return;
  int rank = offsets[blockIdx.x] + __builtin_popcountll(m & ((1ull << lane) - 1ull));
  for (int q = 0; q < wv; ++q) rank += sw[q];
  const unsigned long long vox = keys[p] >> 31;
  float sx = 0.f, sy = 0.f, sz = 0.f, sn = 0.f;
  int cnt = 0;
  for (int j = p; j < n && (keys[j] >> 31) == vox; ++j) {  // input order inside the voxel (the key's low bits ascend)
    const float4 v = pts[(int)(keys[j] & 0x7fffffffull)];
    sx += v.x; sy += v.y; sz += v.z; sn += v.w;
    ++cnt;
  }
  const float c = (float)cnt;
  const float z2 = sn * sn;
  out[rank] = make_float4(sx / c, sy / c, sz / c, z2 > 0.f ? sn / sqrtf(z2) : sn);
}

}  // namespace gorio
