// apd_index.hip -- building the exact search accelerator (SearchIndex, apd_device.h) and the two branch-and-bound searches that
// use it.  Included by apd_api.hip after apd_kernels.hip.
//
//   bbox_kernel / morton_kernel      bounding box, 3 x 11-bit Morton code per point
//   bitonic_*_kernel                 LDS-tiled bitonic sort of (code << 31 | index) keys
//   gather_sorted_kernel / box_kernel   Morton-ordered copy + tile / super-tile boxes
//   nn_search_pruned_kernel          1-NN correspondences (APD:164-180), same packed-key output as nn_search_kernel
//   knn_pruned_kernel                self k-NN (APD:364), same list layout as knn_partial_kernel with splits = 1
//
// Why the pruning is exact: the lower bound of a box is evaluated with the SAME float expression as a point distance
// (dx*dx, + dy*dy, + dz*dz on the clamped coordinate differences).  Every IEEE operation in it is monotone in |dx|, |dy|, |dz|,
// and a point inside the box has component differences at least as large as the clamped ones, so bound <= distance holds in
// float arithmetic, not just in exact arithmetic.  A tile is skipped only when bound > current best, hence no skipped point can
// beat or tie the final answer; ties are broken on the ORIGINAL index, so the result equals the exhaustive search bit for bit.
#include <hip/hip_runtime.h>

namespace gorio {

__device__ __forceinline__ unsigned int f2ord(float f) {  // order-preserving float -> uint
  const unsigned int u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned int o) {
  const unsigned int u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
  return __uint_as_float(u);
}

__global__ void bbox_init_kernel(const IndexJob* __restrict__ jobs, int nj) {
  const int q = blockIdx.x * 64 + threadIdx.x;
  if (q >= nj) return;
  for (int a = 0; a < 3; ++a) {
    jobs[q].bb[a] = 0xffffffffu;
    jobs[q].bb[3 + a] = 0u;
  }
}

// grid: (blocks, jobs), block 256
__global__ __launch_bounds__(256) void bbox_kernel(const IndexJob* __restrict__ jobs) {
  const IndexJob& jb = jobs[blockIdx.y];
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = blockIdx.x * 256 + threadIdx.x; i < jb.n; i += gridDim.x * 256) {
    const float v[3] = {jb.x[i], jb.y[i], jb.z[i]};
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
      atomicMin(jb.bb + a, f2ord(lo[a]));
      atomicMax(jb.bb + 3 + a, f2ord(hi[a]));
    }
  }
}

__device__ __forceinline__ unsigned long long spread11(unsigned int v) {  // 11 bits -> every third bit
  unsigned long long x = v & 0x7ffu;
  x = (x | (x << 32)) & 0x1f00000000ffffull;
  x = (x | (x << 16)) & 0x1f0000ff0000ffull;
  x = (x | (x << 8)) & 0x100f00f00f00f00full;
  x = (x | (x << 4)) & 0x10c30c30c30c30c3ull;
  x = (x | (x << 2)) & 0x1249249249249249ull;
  return x;
}

// grid: (blocks over npow2, jobs)
__global__ __launch_bounds__(256) void morton_kernel(const IndexJob* __restrict__ jobs) {
  const IndexJob& jb = jobs[blockIdx.y];
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= jb.npow2) return;
  unsigned long long key = ~0ull;
  if (i < jb.n) {
    unsigned int q[3];
    const float v[3] = {jb.x[i], jb.y[i], jb.z[i]};
    // ONE cell size for the three axes (cubic cells): tiles of consecutive codes are then compact in metres, not in bbox fractions
    float ext = 1e-6f;
#pragma unroll
    for (int a = 0; a < 3; ++a) ext = fmaxf(ext, ord2f(jb.bb[3 + a]) - ord2f(jb.bb[a]));
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float lo = ord2f(jb.bb[a]);
      float t = (v[a] - lo) / ext * 2047.0f;
      t = fminf(fmaxf(t, 0.0f), 2047.0f);
      q[a] = (unsigned int)t;
    }
    const unsigned long long code = spread11(q[0]) | (spread11(q[1]) << 1) | (spread11(q[2]) << 2);
    key = (code << 31) | (unsigned long long)i;
  }
  jb.keys[i] = key;
}

constexpr int kSortTile = 4096;  // u64 keys per LDS tile (32 KB), 1024 threads

__device__ __forceinline__ void cmpx(unsigned long long& a, unsigned long long& b, bool up) {
  if ((a > b) == up) {
    const unsigned long long t = a;
    a = b;
    b = t;
  }
}

// full bitonic sort of every 4096-key tile in LDS; direction alternates with the tile index so that tiles pair into bitonic
// sequences for the global stages.  grid: (max tiles, jobs), block 1024
__global__ __launch_bounds__(1024) void bitonic_tile_sort_kernel(const IndexJob* __restrict__ jobs) {
  const IndexJob& jb = jobs[blockIdx.y];
  const int base = blockIdx.x * kSortTile;
  if (base >= jb.npow2) return;
  __shared__ unsigned long long s[kSortTile];
  for (int q = threadIdx.x; q < kSortTile; q += 1024) s[q] = jb.keys[base + q];
  __syncthreads();
  for (int k = 2; k <= kSortTile; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = threadIdx.x; t < kSortTile / 2; t += 1024) {
        const int lo = ((t / j) * 2 * j) + (t % j), hi = lo + j;
        const bool up = (((base + lo) & k) == 0);
        unsigned long long a = s[lo], b = s[hi];
        cmpx(a, b, up);
        s[lo] = a;
        s[hi] = b;
      }
      __syncthreads();
    }
  }
  for (int q = threadIdx.x; q < kSortTile; q += 1024) jb.keys[base + q] = s[q];
}

// one global compare-exchange stage (stride j >= kSortTile) of merge size k.  grid: (max npow2 / 2 / 256, jobs), block 256
__global__ __launch_bounds__(256) void bitonic_global_kernel(const IndexJob* __restrict__ jobs, int k, int j) {
  const IndexJob& jb = jobs[blockIdx.y];
  if (k > jb.npow2) return;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= jb.npow2 / 2) return;
  const int lo = ((t / j) * 2 * j) + (t % j), hi = lo + j;
  const bool up = ((lo & k) == 0);
  unsigned long long a = jb.keys[lo], b = jb.keys[hi];
  if ((a > b) == up) {
    jb.keys[lo] = b;
    jb.keys[hi] = a;
  }
}

// remaining stages (stride < kSortTile) of merge size k inside LDS.  grid: (max tiles, jobs), block 1024
__global__ __launch_bounds__(1024) void bitonic_tile_merge_kernel(const IndexJob* __restrict__ jobs, int k) {
  const IndexJob& jb = jobs[blockIdx.y];
  if (k > jb.npow2) return;
  const int base = blockIdx.x * kSortTile;
  if (base >= jb.npow2) return;
  __shared__ unsigned long long s[kSortTile];
  for (int q = threadIdx.x; q < kSortTile; q += 1024) s[q] = jb.keys[base + q];
  __syncthreads();
  for (int j = kSortTile >> 1; j > 0; j >>= 1) {
    for (int t = threadIdx.x; t < kSortTile / 2; t += 1024) {
      const int lo = ((t / j) * 2 * j) + (t % j), hi = lo + j;
      const bool up = (((base + lo) & k) == 0);
      unsigned long long a = s[lo], b = s[hi];
      cmpx(a, b, up);
      s[lo] = a;
      s[hi] = b;
    }
    __syncthreads();
  }
  for (int q = threadIdx.x; q < kSortTile; q += 1024) jb.keys[base + q] = s[q];
}

// grid: (blocks over n_spad, jobs)
__global__ __launch_bounds__(256) void gather_sorted_kernel(const IndexJob* __restrict__ jobs) {
  const IndexJob& jb = jobs[blockIdx.y];
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= jb.idx.n_spad) return;
  if (p < jb.n) {
    const int i = (int)(jb.keys[p] & 0x7fffffffull);
    jb.idx.sx[p] = jb.x[i];
    jb.idx.sy[p] = jb.y[i];
    jb.idx.sz[p] = jb.z[i];
    jb.idx.orig[p] = i;
    jb.idx.s4[p] = make_float4(jb.x[i], jb.y[i], jb.z[i], __int_as_float(i));
  } else {
    jb.idx.sx[p] = 1e30f;
    jb.idx.sy[p] = 1e30f;
    jb.idx.sz[p] = 1e30f;
    jb.idx.orig[p] = 0x7fffffff;
    jb.idx.s4[p] = make_float4(1e30f, 1e30f, 1e30f, __int_as_float(0x7fffffff));
  }
}

// kd refinement of the Morton order: inside every chunk of 4096 sorted points (a compact region of space after the Morton sort)
// the points are re-ordered by recursive MEDIAN SPLITS along the widest axis down to 32-point leaves, entirely in LDS.  Leaves of
// a median-split tree tile space without the long, overlapping boxes that runs of a space-filling curve produce: on the C4
// clouds a query needs 2.2 tiles instead of 6.0 and a wave of 64 queries 11 instead of 17.  Any permutation is a valid index
// (the searches are exact for every ordering), so this changes the work of the searches, never their results.
// Per level (segment sizes 4096 .. 64): segment bounding boxes (wave reductions + LDS atomics), widest axis, bitonic sort of
// every segment by that coordinate, payload permutation.  Padding points (1e30) are the largest on every axis and therefore stay
// at the tail of the chunk.  grid: (chunks over max n_spad, jobs), block 1024.
// Chunk size (a multiple of 1024, at most 4096: the index inside a chunk takes 12 key bits) by the clouds of the call: 2048 points when
// every cloud named in the call has at most kKdSmallCloud points (scans, and local maps of the C3 size), 4096 otherwise.  Measured in round 3: 16 k-point scans against each other
// build faster with the smaller chunk (0.71 -> 0.60 ms per C4 batch: 512-thread workgroups, one sort level less) AND search faster (20
// searches 1.99 -> 1.90 ms, k-NN 1.66 -> 1.59 ms, a lone 5 k x 5 k align 0.95 -> 0.79 ms); scans that meet a big map search it faster
// when THEY are ordered with 4096-point chunks too (64 scans x 1 M-point map: 20 searches 10.2 vs 10.9 ms), and the map's own tiles are
// worse with small chunks (11.2 ms with 1024).
constexpr int kKdSmallCloud = 131072;  // (65536 until the end of round 3: a 100 k-point local map is better off with 2048-point chunks too -- C3 step 2.72 -> 2.55 ms)
template <int kKdChunk>
__global__ __launch_bounds__(kKdChunk / 4) void kd_refine_kernel(const IndexJob* __restrict__ jobs) {
  constexpr int kKdThreads = kKdChunk / 4;  // four points, two compare-exchange pairs per thread and stage
  const IndexJob& jb = jobs[blockIdx.y];
  const int base = blockIdx.x * kKdChunk;
  if (base >= jb.idx.n_spad) return;
  __shared__ float px[kKdChunk], py[kKdChunk], pz[kKdChunk];
  __shared__ int po[kKdChunk];
  __shared__ unsigned long long keys[kKdChunk];
  __shared__ unsigned int sb[64][6];  // per segment: ordered-uint min x,y,z / max x,y,z
  const int tid = threadIdx.x, lane = tid & 63;
  for (int e = tid; e < kKdChunk; e += kKdThreads) {
    const int p = base + e;
    const bool in = p < jb.idx.n_spad;
    px[e] = in ? jb.idx.sx[p] : 1e30f;
    py[e] = in ? jb.idx.sy[p] : 1e30f;
    pz[e] = in ? jb.idx.sz[p] : 1e30f;
    po[e] = in ? jb.idx.orig[p] : 0x7fffffff;
  }
  __syncthreads();
  for (int seg = kKdChunk; seg >= 64; seg >>= 1) {
    const int nseg = kKdChunk / seg;
    for (int q = tid; q < nseg * 6; q += kKdThreads) sb[q / 6][q % 6] = (q % 6) < 3 ? 0xffffffffu : 0u;
    __syncthreads();
    for (int r = 0; r < 4; ++r) {
      const int e = tid + kKdThreads * r;
      const bool valid = po[e] != 0x7fffffff;
      const float v[3] = {px[e], py[e], pz[e]};
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        unsigned int lo = valid ? f2ord(v[a]) : 0xffffffffu, hi = valid ? f2ord(v[a]) : 0u;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
          lo = min(lo, (unsigned int)__shfl_xor((int)lo, off, 64));
          hi = max(hi, (unsigned int)__shfl_xor((int)hi, off, 64));
        }
        if (lane == 0) {
          atomicMin(&sb[e / seg][a], lo);
          atomicMax(&sb[e / seg][3 + a], hi);
        }
      }
    }
    __syncthreads();
    for (int r = 0; r < 4; ++r) {
      const int e = tid + kKdThreads * r;
      const unsigned int* b = sb[e / seg];
      int axis = 0;
      float best = -1.0f;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const float ext = b[3 + a] >= b[a] ? ord2f(b[3 + a]) - ord2f(b[a]) : 0.0f;
        if (ext > best) {
          best = ext;
          axis = a;
        }
      }
      const float c = axis == 0 ? px[e] : (axis == 1 ? py[e] : pz[e]);
      keys[e] = ((unsigned long long)f2ord(c) << 12) | (unsigned long long)e;
    }
    __syncthreads();
    // bitonic sort of every segment.  Strides >= 256 need the whole workgroup (barrier per stage); smaller strides stay inside
    // the 256-key window of one wave, where the in-order LDS pipe of the wave is the only synchronisation needed.
    for (int lk = 1; (1 << lk) <= seg; ++lk) {
      const int k = 1 << lk;
      int lj = lk - 1;  // stride j = 1 << lj; pair t -> lo = (t / j) * 2j + t % j, by shifts
      for (; lj >= 8; --lj) {
        const int j = 1 << lj;
        for (int r = 0; r < 2; ++r) {
          const int t = tid + kKdThreads * r;
          const int lo = ((t >> lj) << (lj + 1)) | (t & (j - 1)), hi = lo + j;
          const bool up = (k == seg) || ((lo & k) == 0);
          unsigned long long a = keys[lo], b = keys[hi];
          if ((a > b) == up) {
            keys[lo] = b;
            keys[hi] = a;
          }
        }
        __syncthreads();
      }
      const int wbase = (tid >> 6) * 256;
      for (; lj >= 0; --lj) {
        const int j = 1 << lj;
        for (int r = 0; r < 2; ++r) {
          const int t = lane + 64 * r;
          const int lo = wbase + (((t >> lj) << (lj + 1)) | (t & (j - 1))), hi = lo + j;
          const bool up = (k == seg) || ((lo & k) == 0);
          unsigned long long a = keys[lo], b = keys[hi];
          if ((a > b) == up) {
            keys[lo] = b;
            keys[hi] = a;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
      if (k >= 256 && 2 * k <= seg) __syncthreads();  // the next merge starts with a workgroup-wide stride
    }
    __syncthreads();
    float nx[4], ny[4], nz[4];
    int no[4];
    for (int r = 0; r < 4; ++r) {
      const int src = (int)(keys[tid + kKdThreads * r] & 4095ull);
      nx[r] = px[src]; ny[r] = py[src]; nz[r] = pz[src]; no[r] = po[src];
    }
    __syncthreads();
    for (int r = 0; r < 4; ++r) {
      const int e = tid + kKdThreads * r;
      px[e] = nx[r]; py[e] = ny[r]; pz[e] = nz[r]; po[e] = no[r];
    }
    __syncthreads();
  }
  for (int e = tid; e < kKdChunk; e += kKdThreads) {
    const int p = base + e;
    if (p < jb.idx.n_spad) {
      jb.idx.sx[p] = px[e];
      jb.idx.sy[p] = py[e];
      jb.idx.sz[p] = pz[e];
      jb.idx.orig[p] = po[e];
      jb.idx.s4[p] = make_float4(px[e], py[e], pz[e], __int_as_float(po[e]));
    }
  }
}

// boxes of 32-point tiles (one thread per tile) -- grid: (blocks over n_tiles, jobs); then super tiles in box_super_kernel
__global__ __launch_bounds__(256) void box_tile_kernel(const IndexJob* __restrict__ jobs) {
  const IndexJob& jb = jobs[blockIdx.y];
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= jb.idx.n_tiles) return;
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int q = 0; q < 32; ++q) {
    const int p = t * 32 + q;
    if (p < jb.n) {
      const float v[3] = {jb.idx.sx[p], jb.idx.sy[p], jb.idx.sz[p]};
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        lo[a] = fminf(lo[a], v[a]);
        hi[a] = fmaxf(hi[a], v[a]);
      }
    }
  }
  float* b = jb.idx.tbox + (size_t)t * 8;
  b[0] = lo[0]; b[1] = lo[1]; b[2] = lo[2]; b[3] = 0.f;
  b[4] = hi[0]; b[5] = hi[1]; b[6] = hi[2]; b[7] = 0.f;
}
__global__ __launch_bounds__(256) void box_super_kernel(const IndexJob* __restrict__ jobs) {
  const IndexJob& jb = jobs[blockIdx.y];
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s >= jb.idx.n_super) return;
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int q = 0; q < 16; ++q) {
    const int t = s * 16 + q;
    if (t < jb.idx.n_tiles) {
      const float* b = jb.idx.tbox + (size_t)t * 8;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        lo[a] = fminf(lo[a], b[a]);
        hi[a] = fmaxf(hi[a], b[4 + a]);
      }
    }
  }
  float* b = jb.idx.sbox + (size_t)s * 8;
  b[0] = lo[0]; b[1] = lo[1]; b[2] = lo[2]; b[3] = 0.f;
  b[4] = hi[0]; b[5] = hi[1]; b[6] = hi[2]; b[7] = 0.f;
}

// boxes of blocks (64 super tiles), one thread per block -- grid: (blocks over n_blk, jobs)
__global__ __launch_bounds__(64) void box_block_kernel(const IndexJob* __restrict__ jobs) {
  const IndexJob& jb = jobs[blockIdx.y];
  const int k = blockIdx.x * 64 + threadIdx.x;
  if (k >= (jb.idx.n_super + 63) / 64) return;
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int q = 0; q < 64; ++q) {
    const int t = k * 64 + q;
    if (t < jb.idx.n_super) {
      const float* b = jb.idx.sbox + (size_t)t * 8;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        lo[a] = fminf(lo[a], b[a]);
        hi[a] = fmaxf(hi[a], b[4 + a]);
      }
    }
  }
  float* b = jb.idx.bbox + (size_t)k * 8;
  b[0] = lo[0]; b[1] = lo[1]; b[2] = lo[2]; b[3] = 0.f;
  b[4] = hi[0]; b[5] = hi[1]; b[6] = hi[2]; b[7] = 0.f;
}

// ----------------------------------------------------------------------------------------------- branch-and-bound searches

// float lower bound of sqdist3(q, p) over all p in the box; same operation sequence as sqdist3 (see the header comment)
__device__ __forceinline__ float box_bound(float qx, float qy, float qz, const float* __restrict__ b) {
  const float cx = fminf(fmaxf(qx, b[0]), b[4]);
  const float cy = fminf(fmaxf(qy, b[1]), b[5]);
  const float cz = fminf(fmaxf(qz, b[2]), b[6]);
  const float dx = qx - cx, dy = qy - cy, dz = qz - cz;
  float r = dx * dx;
  r = r + dy * dy;
  r = r + dz * dz;
  return r;
}

typedef const int __attribute__((address_space(4)))* scalar_ip;

// wave-wide min / max (all lanes receive the result)
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off, 64));
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

// squared distance between two boxes (a lower bound of box_bound(q, tile) for every q inside the query box); the same un-fused
// expression on per-axis gaps, so it is again monotone and never exceeds a real point distance in float arithmetic
__device__ __forceinline__ float box_box_bound(const float (&qlo)[3], const float (&qhi)[3], float4 lo, float4 hi) {
  const float gx = fmaxf(fmaxf(lo.x - qhi[0], qlo[0] - hi.x), 0.0f);
  const float gy = fmaxf(fmaxf(lo.y - qhi[1], qlo[1] - hi.y), 0.0f);
  const float gz = fmaxf(fmaxf(lo.z - qhi[2], qlo[2] - hi.z), 0.0f);
  float r = gx * gx;
  r = r + gy * gy;
  r = r + gz * gz;
  return r;
}

__device__ __forceinline__ float lane_f(float v, int lane) { return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), lane)); }

// Two-level test shared by both searches.  Coarse: 64 tile boxes at a time, one per lane (coalesced 32-byte loads), against the
// box of the wave's 64 queries and the loosest per-lane bound -> ballot mask of candidate tiles.  Fine: for each candidate the
// tile box is broadcast from the lane that holds it (v_readlane, no memory access) and tested per lane; only if some lane still
// needs the tile are its 32 points fetched (wave-uniform scalar loads) and evaluated by all lanes.
// visit order: 64-tile groups outward from `g0` (where near neighbours are expected) so the bound tightens early.

#ifdef GORIO_STATS  // development statistics (never in the shipped build): work counters of the pruned searches
// [0..7] work counters, [8..15] cycles per phase of nn_search_pruned_kernel (s_memtime), summed over waves; 1024 copies (a wave adds to the
// copy of its number mod 1024: one shared set made the atomics the bottleneck of the instrumented kernel), summed by the host
__device__ unsigned long long g_search_stats[1024][24];
#define STAT_ROW() g_search_stats[((blockIdx.x + 977u * blockIdx.z) * 4u + (threadIdx.x >> 6)) & 1023u]
#define STAT_ADD(k, v) do { if ((threadIdx.x & 63) == 0) atomicAdd(&STAT_ROW()[k], (unsigned long long)(v)); } while (0)
#define STAT_MAX(k, v) do { if ((threadIdx.x & 63) == 0) atomicMax(&STAT_ROW()[k], (unsigned long long)(v)); } while (0)
#define STAT_DECL(name) int name = 0
#define STAT_INC(name) ++name
// wave-local counters and phase clocks of nn_search_pruned_kernel: accumulated in registers, ONE set of atomics per wave at the end
#define STAT_CLOCK_DECL() unsigned long long st_acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned int st_cnt_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long st_t_ = __builtin_amdgcn_s_memtime()
#define STAT_PHASE(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); st_acc_[k] += now_ - st_t_; st_t_ = now_; } while (0)
#define STAT_LOCAL(k, v) st_cnt_[k] += (unsigned int)(v)
#define STAT_CLOCK_FLUSH() do { if ((threadIdx.x & 63) == 0) { unsigned long long tot_ = 0; for (int k_ = 0; k_ < 8; ++k_) { tot_ += st_acc_[k_]; atomicAdd(&STAT_ROW()[8 + k_], st_acc_[k_]); atomicAdd(&STAT_ROW()[k_], (unsigned long long)st_cnt_[k_]); } int b_ = 0; while (b_ < 7 && tot_ >= (32768ull << b_)) ++b_; atomicAdd(&STAT_ROW()[16 + b_], 1ull); } } while (0)  /* [16..23]: histogram of a wave's cycles: < 32 k, < 64 k, ... */
#else
#define STAT_LOCAL(k, v) do { } while (0)
#define STAT_CLOCK_DECL() do { } while (0)
#define STAT_PHASE(k) do { } while (0)
#define STAT_CLOCK_FLUSH() do { } while (0)
#define STAT_ADD(k, v) do { } while (0)
#define STAT_MAX(k, v) do { } while (0)
#define STAT_DECL(name) do { } while (0)
#define STAT_INC(name) do { } while (0)
#endif

// ----------------------------------------------------------------------------------------------- 1-NN correspondences, pruned
//
// 1-NN of every source point in the target (APD:164-180), same packed-key output as nn_search_kernel.  One lane = one source point taken in
// the source's own sorted order (a wave's 64 queries are spatially compact).  Round-3 rewrite.  The round-2 kernel filled 52 % of the VALU
// issue time alone, 61 % of it with half-rate instructions, its waves were parked 40 % of their life, and -- what the counters did not
// show -- a launch lasted as long as its SLOWEST wave (5 - 25 x the mean).  What changed, in the order it paid:
//   * WORK PLAN (nn_plan_kernel below): every wave records the cycles it took; from the third search of an align on, waves that were slow
//     in the second one are cut into 2 .. 16 parts and the parts are dispatched heaviest first.
//   * ONE-WAVE WORKGROUPS: the waves of a workgroup never cooperate, but a workgroup keeps its LDS (the occupancy limit here) until its
//     slowest wave ends.
//   * FEWER DEPENDENT ROUND TRIPS (a wave is a chain of L2 / Infinity-Cache round trips of 1-2 k cycles each): everything that does not
//     depend on the bound is loaded at the start, ahead of the seed chain (query, seed, boxes of the first 64 blocks, of the home block's
//     super tiles and of the home group's tiles); the next group's tile boxes are in flight while this group's tiles are tested; loads are
//     unconditional with clamped indices, validity being applied to the ballots (a load inside a branch makes the compiler wait for every
//     outstanding load at the join); the optimiser state is read through the scalar cache, not by flat loads.
//   * a BLOCK level (32 768 points) above the super tiles: a wave against a 1 M-point map tests 31 block boxes and the super tiles of
//     the few surviving blocks instead of all 1954 super tiles; the group mask of a pass lives in registers, not in LDS.
//   * the (lane, tile) evaluations of a batch are COMPACTED: the round-2 walk took max-over-lanes rounds per batch (4.7 rounds of 64
//     lanes per wave for 2.2 tiles per lane); now every (query, slot) pair becomes an item of a list in LDS and the wave evaluates
//     64 items per round whatever query they belong to (the query is re-read from LDS): 3.6 rounds.
//   * an item keeps only the MINIMUM distance of its 32 candidates (v_min3_u32 on the float bits: 0.56 half-rate instruction per
//     candidate instead of a 64-bit compare and two selects) and which 8-candidate groups hold it; items meet in an LDS atomic
//     min on (distance bits, slot, groups); the owning lane then re-evaluates only the winning group to recover the ORIGINAL index
//     (lowest index among equal distances).  Equal minima from two different tiles -- the only case the packed key cannot order by
//     index -- raise a flag and that lane re-walks its slots with full (distance, index) keys.
//   * the previous correspondence (by sorted source position, pd.seed) seeds only the BOUND; its index is found again when its tile
//     is evaluated (the tile's box bound never exceeds the distance of a point inside it, so the tile is always taken).
//   * wave reductions by DPP (v_max_u32 row_ror / row_bcast on order-preserving integer keys) instead of six ds_bpermute round trips each;
//     the per-lane lower bound of a tile box is max3(lo - q, q - hi, 0) per axis (11 full-rate + 4 half-rate instructions, was 8 + 12);
//     tiles are staged from one float4 array (x, y, z, original index): one 16-byte load per point instead of four.
// Measured and dropped: the candidate tile's box by a prefetched scalar load instead of six v_readlane (every test then waits ~ 400 cycles
// for the scalar cache); 12 or 16 slots per batch (fewer flushes, but the LDS they take costs a wave per SIMD: 2.55 / 2.67 vs 2.50 ms).
#ifndef GORIO_NN_SLOTS
#define GORIO_NN_SLOTS 8
#endif
#ifndef GORIO_NN_WAVES
#define GORIO_NN_WAVES 5  // 96 registers; the 36 bytes of scratch this costs were measured: 2.00 vs 2.10 ms per 20 launches with 4
#endif
#ifndef GORIO_NN_BLOCK
#define GORIO_NN_BLOCK 64  // threads per workgroup: the waves of a workgroup never cooperate, and a workgroup holds its LDS until its SLOWEST wave ends
#endif
constexpr int kNnBlock = GORIO_NN_BLOCK;

// wave-wide max / min of an unsigned key by DPP; the result is wave-uniform (an SGPR after readlane)
__device__ __forceinline__ unsigned int wave_umax(unsigned int v) {
  int x = (int)v;
#define GORIO_DPP_MAX(ctrl, rmask) x = (int)max((unsigned int)x, (unsigned int)__builtin_amdgcn_update_dpp(0, x, ctrl, rmask, 0xf, true))
  GORIO_DPP_MAX(0x128, 0xf);  // row_ror:8
  GORIO_DPP_MAX(0x124, 0xf);  // row_ror:4
  GORIO_DPP_MAX(0x122, 0xf);  // row_ror:2
  GORIO_DPP_MAX(0x121, 0xf);  // row_ror:1  -> every lane holds its row's maximum
  GORIO_DPP_MAX(0x142, 0xa);  // row_bcast:15 into rows 1 and 3
  GORIO_DPP_MAX(0x143, 0xc);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's maximum
#undef GORIO_DPP_MAX
  return (unsigned int)__builtin_amdgcn_readlane(x, 63);
}
__device__ __forceinline__ float wave_fmax_u(float v) { return ord2f(wave_umax(f2ord(v))); }
__device__ __forceinline__ float wave_fmin_u(float v) { return ord2f(~wave_umax(~f2ord(v))); }

__device__ __forceinline__ unsigned int umin3(unsigned int a, unsigned int b, unsigned int c) { return min(min(a, b), c); }

struct TileBox {
  float lx, ly, lz, p0, hx, hy, hz, p1;
};
// per-lane lower bound of sqdist3(q, p) over the box: per axis max(lo - q, q - hi, 0) is |q - clamp(q)| computed without a clamp, and
// the squares are summed in the order of sqdist3, so the value is the one box_bound() gives
__device__ __forceinline__ float box_bound_s(float qx, float qy, float qz, const TileBox& b) {
  const float dx = fmaxf(fmaxf(b.lx - qx, qx - b.hx), 0.0f);
  const float dy = fmaxf(fmaxf(b.ly - qy, qy - b.hy), 0.0f);
  const float dz = fmaxf(fmaxf(b.lz - qz, qz - b.hz), 0.0f);
  float r = dx * dx;
  r = r + dy * dy;
  r = r + dz * dz;
  return r;
}

// grid: (ceil(n_spad_src / kNnBlock), splits, pairs), block kNnBlock.  Output as nn_search_kernel: best_key[orig source index].
typedef float v4f_t __attribute__((ext_vector_type(4)));
struct global_f4p {  // 16-byte loads through the GLOBAL address space (descriptor members are generic pointers: a plain load would be flat_load)
  const __attribute__((address_space(1))) v4f_t* p;
  __device__ __forceinline__ explicit global_f4p(const void* q) : p((const __attribute__((address_space(1))) v4f_t*)q) {}
  __device__ __forceinline__ float4 operator[](size_t i) const {
    const v4f_t v = p[i];
    return make_float4(v.x, v.y, v.z, v.w);
  }
};
// LDS byte address of a __shared__ object and eight 16-byte LDS reads issued back to back with ONE wait.  Written as one asm block
// because the compiler (i) narrows a float4 read whose .w is unused to ds_read_b96, which takes 8 LDS cycles instead of 4, and
// (ii) hoists every read of an unrolled loop to its top, which costs a wave per SIMD in registers here.
__device__ __forceinline__ unsigned int lds_addr(const void* p) { return (unsigned int)(size_t)(const __attribute__((address_space(3))) char*)p; }
__device__ __forceinline__ void lds_read8_b128(v4f_t (&c)[8], unsigned int addr) {
  asm volatile(
      "ds_read_b128 %0, %8\n\t"
      "ds_read_b128 %1, %8 offset:16\n\t"
      "ds_read_b128 %2, %8 offset:32\n\t"
      "ds_read_b128 %3, %8 offset:48\n\t"
      "ds_read_b128 %4, %8 offset:64\n\t"
      "ds_read_b128 %5, %8 offset:80\n\t"
      "ds_read_b128 %6, %8 offset:96\n\t"
      "ds_read_b128 %7, %8 offset:112\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(c[0]), "=&v"(c[1]), "=&v"(c[2]), "=&v"(c[3]), "=&v"(c[4]), "=&v"(c[5]), "=&v"(c[6]), "=&v"(c[7])
      : "v"(addr)
      : "memory");
}

__global__ __launch_bounds__(kNnBlock) __attribute__((amdgpu_waves_per_eu(GORIO_NN_WAVES, GORIO_NN_WAVES))) void nn_search_pruned_kernel(const PairDesc* __restrict__ descs, float bound_f, int mode) {
  // (No pair -> XCD renumbering here, unlike linearize_kernel and the k-NN kernels: measured, it changes nothing for 64 pairs with targets
  // of their own -- 1.89 ms per 20 launches either way -- and costs 12 % against a shared 1 M-point map, where it only takes away the
  // balance of the heaviest-first work list.)
  const PairDesc& pd = descs[blockIdx.z];
  // the optimiser state is constant while this kernel runs: read it through the scalar cache (a generic pointer would make these flat
  // loads, whose completion the compiler can only wait for together with every other outstanding load)
  const __attribute__((address_space(4))) PairState* st = (const __attribute__((address_space(4))) PairState*)pd.state;
  if (st->done) return;
  const SearchIndex& si = pd.src.idx;
  const SearchIndex& ti = pd.tgt.idx;
  // Which query wave, and which share of its tile groups, this wave of the grid works on.  Natural launches (mode bit 1 clear): wave =
  // position in the grid, the groups dealt over gridDim.y workgroups.  Planned launches: the entry of nn_plan_kernel's list -- waves that
  // were slow in the previous search come first and are cut into more parts (a launch lasts as long as its slowest wave).
  const unsigned long long t_start = __builtin_amdgcn_s_memtime();
  // (made wave-uniform for the compiler right here: everything the traversal state depends on must stay in scalar registers)
  int wq = __builtin_amdgcn_readfirstlane(blockIdx.x * (kNnBlock / 64) + (threadIdx.x >> 6)), part = blockIdx.y, nparts = gridDim.y;
  if (mode & 2) {
    const __attribute__((address_space(4))) unsigned int* plan = (const __attribute__((address_space(4))) unsigned int*)pd.nn_plan;
    if ((unsigned int)wq >= plan[0]) return;
    const unsigned int e = plan[1 + wq];
    wq = (int)(e >> 8);
    part = (int)((e >> 4) & 15u);
    nparts = 1 << (e & 7u);
  }
  const int p = wq * 64 + (threadIdx.x & 63);
  if (wq * 64 >= si.n) return;
  if (wq * 64 < pd.shard_lo || wq * 64 >= pd.shard_hi) return;  // another rank's part of the source (bounds are multiples of 256)
  constexpr int S = GORIO_NN_SLOTS;
  constexpr unsigned int kNone = 0xffffffffu;
  static_assert(S >= 2 && S <= 16, "slot index is packed into 4 bits");
  __shared__ float4 s_pts[kNnBlock / 64][S][33];  // 33: slots 528 B apart -> lanes on different slots hit different banks
  __shared__ float4 s_q[kNnBlock / 64][64];
  __shared__ unsigned long long s_win[kNnBlock / 64][64];
  __shared__ unsigned short s_items[kNnBlock / 64][64 * S];
  __shared__ unsigned long long s_tie[kNnBlock / 64];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  STAT_CLOCK_DECL();
  const int pq = p < si.n ? p : si.n - 1;
  const global_f4p tb4(ti.tbox);
  const global_f4p sb4(ti.sbox);
  const global_f4p kb4(ti.bbox);
  const global_f4p t4(ti.s4);
  const int ng = (ti.n_tiles + 63) / 64;
  const int n_blk = (ti.n_super + 63) >> 6;  // block = 64 super tiles = 16 groups = 32 768 points
  int g0 = (int)(((long)(wq * 64) * ng) / (si.n > 0 ? si.n : 1));
  g0 = __builtin_amdgcn_readfirstlane(g0 < ng ? g0 : ng - 1);
  // ---- every load that does not depend on the bound is issued NOW, ahead of the seed chain (the kernel is a chain of dependent
  // round trips to L2, 1-2 k cycles each under load): the query, its seed, the boxes of the first 64 blocks, of the super tiles of the
  // home block and of the tiles of the home group (where the wave's queries sit in the target's order: almost always the first group
  // visited).  The traversal below takes a box from these registers when the id matches and loads it otherwise.
  // (every index is clamped into its array and the condition applied to the loaded value: a load inside a branch ends the compiler's
  // tracking of outstanding loads at the join, and it then waits for ALL of them at the first use of any)
  // the seed first: the point it names is the one dependent load of this block, and the counter of outstanding loads is in order
  int sd = ((const __attribute__((address_space(1))) int*)pd.seed)[pq];  // garbage before the first search of an align: used only when seeded
  __builtin_amdgcn_sched_barrier(0);
  const float4 sp = global_f4p(si.s4)[pq];
  const bool seeded = st->n_linearize > 0;
  int pt_g = g0;  // group whose tile boxes sit in (pt_lo, pt_hi)
  float4 pt_lo, pt_hi;
  {
    const int t = g0 * 64 + lane, tc = t < ti.n_tiles ? t : ti.n_tiles - 1;
    pt_lo = tb4[2 * (size_t)tc];  // lanes past the last tile hold a copy of it: they are masked out of the ballot, not out of the data
    pt_hi = tb4[2 * (size_t)tc + 1];
  }
  const int ps_b = g0 >> 4;  // block whose super-tile boxes sit in (ps_lo, ps_hi)
  float4 ps_lo, ps_hi;
  {
    const int t = ps_b * 64 + lane, tc = t < ti.n_super ? t : ti.n_super - 1;
    ps_lo = sb4[2 * (size_t)tc];
    ps_hi = sb4[2 * (size_t)tc + 1];
  }
  float4 pk_lo, pk_hi;  // boxes of blocks 0 .. 63
  {
    const int tc = lane < n_blk ? lane : n_blk - 1;
    pk_lo = kb4[2 * (size_t)tc];
    pk_hi = kb4[2 * (size_t)tc + 1];
  }
  if (!seeded || p >= si.n || sd < 0 || sd >= pd.tgt.n) sd = -1;
  const float4 tp = global_f4p(pd.tgt.p4)[sd < 0 ? 0 : sd];
  float qx, qy, qz;
  {
    float Tf[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) Tf[i] = st->Tf[i];
    transform_f(Tf, sp.x, sp.y, sp.z, qx, qy, qz);
  }
  // candidates farther than the correspondence gate can never be accepted (APD:183): the gate is the first bound.  Warm start inside one
  // align: the correspondence of the previous linearisation, evaluated under the CURRENT pose with the same expression as any other
  // candidate, tightens the bound; the answer does not depend on it (any target point is a valid seed).
  float bestd = bound_f;
  {
    const float d = sqdist3(qx, qy, qz, tp.x, tp.y, tp.z);
    if (sd >= 0 && d <= bestd) bestd = d;
  }
  unsigned long long best = ((unsigned long long)__float_as_uint(bestd) << 32) | kNone;  // index unknown until a tile delivers it
  s_q[wave][lane] = make_float4(qx, qy, qz, 0.f);
  const float qlo[3] = {wave_fmin_u(qx), wave_fmin_u(qy), wave_fmin_u(qz)};
  const float qhi[3] = {wave_fmax_u(qx), wave_fmax_u(qy), wave_fmax_u(qz)};
  float wb = __uint_as_float(wave_umax(__float_as_uint(bestd)));  // loosest bound of the wave (distances are >= 0: the bits order like the values)
  STAT_LOCAL(5, 1);
  // workgroups of one query wave (gridDim.y > 1 only when a launch has too few query waves to fill the chip, or for big targets) share the
  // groups round-robin: bit pattern of "my" groups inside any 64-group word (gridDim.y is a power of two <= 16)
  const int nsplit = nparts;
  unsigned long long mine = 1ull << (part & 63);
  for (int sft = nsplit; sft < 64; sft <<= 1) mine |= mine << sft;
  // The first, unseeded launch of an align has only the gate as its bound: it flushes its first batch after two tiles so that every lane
  // owns a real bound before the remaining tiles are tested.
  int flush_at = seeded ? S : 2;
  unsigned int mneed = 0u;  // slots of the current batch this lane must visit
  int slot_tile = 0;        // lane s: tile held by slot s
  int nslots = 0;           // wave-uniform
  // The traversal is ONE loop over a wave-uniform state (so the batch evaluation below exists once in the code).  FOUR levels of boxes:
  // blocks (32 768 points) 64 per instruction against the box of the wave's queries and its loosest bound; the 64 super tiles (512 points,
  // four per group of 64 tiles) of every surviving block -> bit mask of the groups that can matter at all (a pass covers 2048 groups; the
  // words of the mask live in lane w of a register pair); those groups, nearest (in index order, which follows space) first, get the
  // tile-level test one tile per lane, the next group's boxes being loaded while this group's tiles are tested; the surviving tiles are
  // tested per lane against boxes broadcast with v_readlane.
  constexpr int kPassGroups = 2048;
  int c0 = -kPassGroups, cg = 0, nwords = 0, gl = 0, w0 = 0, v = 0, wcur = 0, pivot = 0, g = 0;
  unsigned long long gm = 0ull, mask = 0ull;
  unsigned int gmw_lo = 0u, gmw_hi = 0u;  // lane w: word w of the pass's group mask
  float4 lo = pt_lo, hi = pt_hi;          // tile boxes of the current group, one tile per lane
  bool done = false;
  auto nearest_bit = [](unsigned long long m, int pv) -> int {  // the set bit nearest to pv (ties: upward)
    const unsigned long long up = m & (~0ull << pv), dn = m & ~(~0ull << pv);
    const int bu = up ? __builtin_ctzll(up) : 1000, bd = dn ? 63 - __builtin_clzll(dn) : -1000;
    return (bu - pv) <= (pv - bd) ? bu : bd;
  };
  STAT_PHASE(0);
  for (;;) {
    while (!mask && !done) {  // advance to the next group that has candidate tiles
      if (gm) {
        // groups are visited outward from where the queries sit, so the bound tightens early
        const int bit = nearest_bit(gm, pivot);
        gm &= ~(1ull << bit);
        g = c0 + wcur * 64 + bit;
        if (pt_g == g) {
          lo = pt_lo;
          hi = pt_hi;
        } else {
          const int t = g * 64 + lane, tc = t < ti.n_tiles ? t : ti.n_tiles - 1;
          lo = tb4[2 * (size_t)tc];
          hi = tb4[2 * (size_t)tc + 1];
        }
        mask = __ballot(g * 64 + lane < ti.n_tiles && box_box_bound(qlo, qhi, lo, hi) <= wb);
        if (gm) {  // the boxes of the group that will be visited next are in flight while this group's tiles are tested and evaluated (issued
                   // AFTER the last use of this group's loads, so that no wait for those covers these)
          pt_g = c0 + wcur * 64 + nearest_bit(gm, pivot);
          const int t = pt_g * 64 + lane, tc = t < ti.n_tiles ? t : ti.n_tiles - 1;
          pt_lo = tb4[2 * (size_t)tc];
          pt_hi = tb4[2 * (size_t)tc + 1];
        }
        STAT_LOCAL(7, __builtin_popcountll(mask));
        STAT_LOCAL(1, 1);
        continue;
      }
      if (v < 2 * nwords) {
        const int woff = (v + 1) >> 1;
        wcur = (v & 1) ? w0 - woff : w0 + woff;
        ++v;
        if (wcur < 0 || wcur >= nwords) continue;
        gm = (((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)gmw_hi, wcur) << 32) | (unsigned int)__builtin_amdgcn_readlane((int)gmw_lo, wcur)) & mine;
        pivot = wcur == w0 ? (gl & 63) : (wcur > w0 ? 0 : 63);
        continue;
      }
      c0 += kPassGroups;  // next pass of up to 2048 groups = 128 blocks
      if (c0 >= ng) {
        done = true;
        break;
      }
      cg = ng - c0 < kPassGroups ? ng - c0 : kPassGroups;
      nwords = (cg + 63) >> 6;
      gmw_lo = 0u;
      gmw_hi = 0u;
      for (int bw = 0; bw < 2; ++bw) {  // the pass's blocks, 64 per ballot
        const int blk_base = (c0 >> 4) + bw * 64;
        if (blk_base >= n_blk) break;
        unsigned long long bm = 1ull;  // a target of one block: nothing to test
        if (n_blk > 1) {
          float4 klo = pk_lo, khi = pk_hi;
          if (blk_base != 0) {
            const int tc = blk_base + lane < n_blk ? blk_base + lane : n_blk - 1;
            klo = kb4[2 * (size_t)tc];
            khi = kb4[2 * (size_t)tc + 1];
          }
          bm = __ballot(blk_base + lane < n_blk && box_box_bound(qlo, qhi, klo, khi) <= wb);
        }
        STAT_LOCAL(2, __builtin_popcountll(bm));
        while (bm) {
          const int bb = __builtin_ctzll(bm);
          bm &= bm - 1;
          const int blk = blk_base + bb;
          float4 slo = ps_lo, shi = ps_hi;
          if (ps_b != blk) {
            const int t = blk * 64 + lane, tc = t < ti.n_super ? t : ti.n_super - 1;
            slo = sb4[2 * (size_t)tc];
            shi = sb4[2 * (size_t)tc + 1];
          }
          unsigned long long m = __ballot(blk * 64 + lane < ti.n_super && box_box_bound(qlo, qhi, slo, shi) <= wb);
          m |= m >> 1;
          m |= m >> 2;  // bit 4 j: any of the four super tiles of group j
          m &= 0x1111111111111111ull;
          m = (m | (m >> 3)) & 0x0303030303030303ull;
          m = (m | (m >> 6)) & 0x000f000f000f000full;
          m = (m | (m >> 12)) & 0x000000ff000000ffull;
          m = (m | (m >> 24)) & 0xffffull;  // 16 group bits of this block
          const int rel = blk - (c0 >> 4);  // block inside the pass: word rel / 4, bits (rel % 4) * 16 ..
          const unsigned long long sh = m << ((rel & 3) * 16);
          if (lane == (rel >> 2)) {
            gmw_lo |= (unsigned int)sh;
            gmw_hi |= (unsigned int)(sh >> 32);
          }
        }
      }
      gl = g0 - c0;  // where near neighbours are expected, relative to this pass
      gl = gl < 0 ? 0 : (gl >= cg ? cg - 1 : gl);
      w0 = gl >> 6;
      v = 0;
    }
    STAT_PHASE(1);
    if (mask) {  // one candidate tile: its box is broadcast from the lane that holds it
      const int cur = __builtin_ctzll(mask);
      mask &= mask - 1;
      const float bx0 = lane_f(lo.x, cur), bx1 = lane_f(lo.y, cur), bx2 = lane_f(lo.z, cur);
      const float bx4 = lane_f(hi.x, cur), bx5 = lane_f(hi.y, cur), bx6 = lane_f(hi.z, cur);
      const TileBox bcur = TileBox{bx0, bx1, bx2, 0.f, bx4, bx5, bx6, 0.f};
      const bool need = box_bound_s(qx, qy, qz, bcur) <= bestd;
      if (__ballot(need)) {
        if (need) mneed |= 1u << nslots;
        if (lane == nslots) slot_tile = g * 64 + cur;
        ++nslots;
        STAT_LOCAL(0, 1);
      }
    }
    STAT_PHASE(2);
    if (nslots >= flush_at || (done && nslots > 0)) {
      flush_at = S;
      // ---- batch evaluation
      // 1. stage the tiles: lanes 0-31 / 32-63 bring one point each of two slots per pass; all loads are issued before the first store
      //    (slots beyond nslots re-load an older tile and are not stored)
      float4 stage[(S + 1) / 2];
#pragma unroll
      for (int h = 0; h < (S + 1) / 2; ++h) {
        const int ta = __builtin_amdgcn_readlane(slot_tile, 2 * h);
        const int tb = __builtin_amdgcn_readlane(slot_tile, 2 * h + 1 < S ? 2 * h + 1 : 2 * h);
        const int tile = lane < 32 ? ta : tb;
        stage[h] = t4[(size_t)tile * 32 + (lane & 31)];
      }
#pragma unroll
      for (int h = 0; h < (S + 1) / 2; ++h) {
        const int my = 2 * h + (lane >> 5);
        if (my < nslots) s_pts[wave][my][lane & 31] = stage[h];
      }
      STAT_PHASE(3);
      // 2. item list, slot-major (consecutive items mostly share a slot: their candidate reads are LDS broadcasts)
      int nitems = 0;
      for (int s = 0; s < nslots; ++s) {
        const bool mine = (mneed >> s) & 1u;
        const unsigned long long bal = __ballot(mine);
        if (mine) s_items[wave][nitems + __builtin_amdgcn_mbcnt_hi((unsigned int)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)bal, 0u))] = (unsigned short)((lane << 4) | s);
        nitems += __builtin_popcountll(bal);
      }
      s_win[wave][lane] = ((unsigned long long)__float_as_uint(bestd) << 32) | kNone;
      if (lane == 0) s_tie[wave] = 0ull;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      STAT_LOCAL(3, nitems);
      STAT_LOCAL(6, 1);
      STAT_PHASE(4);
      // 3. rounds of 64 items: minimum of the tile's 32 distances (float bits, v_min3_u32) and the 8-candidate groups that hold it
      for (int r0 = 0; r0 < nitems; r0 += 64) {
        STAT_LOCAL(4, 1);
        const int i = r0 + lane;
        const bool valid = i < nitems;
        const unsigned int it = valid ? (unsigned int)s_items[wave][i] : 0u;
        const int Q = (int)(it >> 4), sl = (int)(it & 15u);
        const float4 q4 = s_q[wave][Q];
        unsigned int mg[4];
        const unsigned int cp_addr = lds_addr(&s_pts[wave][sl][0]);
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) {
          v4f_t c[8];
          lds_read8_b128(c, cp_addr + gg * 128);  // eight 16-byte reads in flight, one wait (the compiler would hoist all 32: 128 registers)
          unsigned int d[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) d[k] = __float_as_uint(sqdist3(q4.x, q4.y, q4.z, c[k].x, c[k].y, c[k].z));
          mg[gg] = umin3(umin3(d[0], d[1], d[2]), umin3(d[3], d[4], d[5]), min(d[6], d[7]));
          asm volatile("" : "+v"(mg[gg]));  // pins this group's arithmetic between its reads and the next group's (volatile asm keeps its order)
        }
        const unsigned int m = min(umin3(mg[0], mg[1], mg[2]), mg[3]);
        const unsigned int gmk = (mg[0] == m ? 1u : 0u) | (mg[1] == m ? 2u : 0u) | (mg[2] == m ? 4u : 0u) | (mg[3] == m ? 8u : 0u);
        if (valid) {
          const unsigned long long key = ((unsigned long long)m << 32) | (unsigned long long)(((unsigned int)sl << 4) | gmk);
          const unsigned long long old = atomicMin(&s_win[wave][Q], key);
          // equal minima in two tiles: the packed key cannot say which holds the lower original index
          if ((unsigned int)(old >> 32) == m && (unsigned int)old != kNone) atomicOr(&s_tie[wave], 1ull << Q);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      STAT_PHASE(5);
      // 4. the owner of a query recovers the original index of its winner from the winning group(s)
      const unsigned long long w = s_win[wave][lane];
      const unsigned long long ties = s_tie[wave];
      if ((unsigned int)w != kNone) {
        const unsigned int m = (unsigned int)(w >> 32);
        const int sl = (int)(((unsigned int)w >> 4) & 15u);
        unsigned int gmk = (unsigned int)w & 15u;
        unsigned int idx = kNone;
        const float4* __restrict__ cp = s_pts[wave][sl];
        while (gmk) {
          const int gg = __builtin_ctz(gmk);
          gmk &= gmk - 1u;
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const float4 c = cp[gg * 8 + k];
            const unsigned int d = __float_as_uint(sqdist3(qx, qy, qz, c.x, c.y, c.z));
            const unsigned int ci = __float_as_uint(c.w);
            idx = d == m ? min(idx, ci) : idx;
          }
        }
        const unsigned long long key = ((unsigned long long)m << 32) | (unsigned long long)idx;
        best = key < best ? key : best;
      }
      if (ties) {  // rare: lanes whose minimum occurred in more than one tile walk all their slots with full keys
        if ((ties >> lane) & 1ull) {
          unsigned int mm = mneed;
          while (mm) {
            const int sl = __builtin_ctz(mm);
            mm &= mm - 1u;
#pragma unroll 1
            for (int k = 0; k < 32; ++k) {
              const float4 c = s_pts[wave][sl][k];
              const float d = sqdist3(qx, qy, qz, c.x, c.y, c.z);
              const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned long long)__float_as_uint(c.w);
              best = key < best ? key : best;
            }
          }
        }
      }
      bestd = __uint_as_float((unsigned int)(best >> 32));
      wb = __uint_as_float(wave_umax(__float_as_uint(bestd)));
      nslots = 0;
      mneed = 0u;
      __builtin_amdgcn_wave_barrier();
      STAT_PHASE(6);
    }
    if (done) break;
  }
  if (p < si.n && (unsigned int)best != kNone) {
    const int oi = __float_as_int(sp.w);  // original index of this query (the .w of its sorted record)
    if (nsplit > 1) atomicMin(pd.best_key + oi, best);
    else pd.best_key[oi] = best;
    pd.seed[p] = (int)(unsigned int)best;  // next launch's warm start (with nsplit > 1 whichever workgroup writes last: any target point is valid)
  }
#ifndef GORIO_NN_NOWORK
  if ((threadIdx.x & 63) == 0) {  // what this wave cost, for the plan of the next launches
    const unsigned int dt = (unsigned int)(__builtin_amdgcn_s_memtime() - t_start);
    unsigned int* wk = pd.nn_work + (size_t)(mode & 1) * pd.nn_wcap + wq;
    if (nparts == 1) *wk = dt;  // (this half was zeroed by the previous launch, or by arm_keys_kernel)
    else atomicAdd(wk, dt);
    if (part == 0) pd.nn_work[(size_t)((mode & 1) ^ 1) * pd.nn_wcap + wq] = 0u;
  }
#endif
  STAT_PHASE(7);
  STAT_CLOCK_FLUSH();
}

// Work plan of the next pruned searches of every pair from the cycles its query waves took in a finished one (nn_work[buf]).  A launch lasts
// as long as its slowest wave, and a few waves (far returns whose bound ball is full of target points) take 5 - 25 x the mean: each wave is
// cut into 1, 2, 4, 8 or 16 parts (the parts share the wave's tile groups round-robin) so that no part exceeds about half the time the
// launch would take if its work were spread perfectly, and the parts are listed heaviest first (dispatch follows the list).  The list
// holds at most max_entries parts (the grid of the planned launches: twice the query waves for a batch, up to sixteen times for a lone
// pair whose 256 waves cannot fill the chip otherwise).  The plan changes the schedule, never a result.  grid: pairs, block 256.
__global__ __launch_bounds__(256) void nn_plan_kernel(const PairDesc* __restrict__ descs, int buf, int npairs, int max_entries, int budget_div) {
  const PairDesc& pd = descs[blockIdx.x];
  const int nw = (pd.src.idx.n + 63) / 64;
  const unsigned int* __restrict__ w = pd.nn_work + (size_t)buf * pd.nn_wcap;
  unsigned int* __restrict__ plan = pd.nn_plan;
  __shared__ unsigned long long s_tot;
  __shared__ unsigned int s_cnt[32], s_off[32], s_entries;
  const int tid = threadIdx.x;
  if (tid == 0) s_tot = 0ull;
  if (tid < 32) s_cnt[tid] = 0u;
  __syncthreads();
  unsigned long long loc = 0ull;
  for (int i = tid; i < nw; i += 256) loc += w[i];
  atomicAdd(&s_tot, loc);
  __syncthreads();
  // budget of one part: half of (all work of the batch / wave slots of the chip), assuming the pairs of a batch are alike
  unsigned long long T = s_tot * (unsigned long long)npairs / (unsigned long long)budget_div;
  if (T < 4096ull) T = 4096ull;
  int max_lg = 4;
  auto parts_of = [&](unsigned int wi) -> int {
    int lg = 0;
    while (lg < max_lg && (unsigned long long)wi > (T << lg)) ++lg;
    return lg;  // log2(parts)
  };
  for (int attempt = 0; attempt < 13; ++attempt) {  // at most max_entries (>= nw) entries: the grid of a planned launch covers exactly that many
    if (attempt == 12) max_lg = 0;  // never reached with sane work figures: one part per wave always fits
    if (tid == 0) s_entries = 0u;
    __syncthreads();
    unsigned int mine = 0u;
    for (int i = tid; i < nw; i += 256) mine += 1u << parts_of(w[i]);
    atomicAdd(&s_entries, mine);
    __syncthreads();
    const bool ok = s_entries <= (unsigned int)max_entries;
    __syncthreads();
    if (ok) break;
    T *= 2ull;
  }
  // counting sort by the work of a part, heaviest bucket first (32 buckets of log2)
  for (int i = tid; i < nw; i += 256) {
    const int lg = parts_of(w[i]);
    const unsigned int pw = w[i] >> lg;
    const int bkt = 31 - (pw ? 31 - __builtin_clz(pw) : 0);
    atomicAdd(&s_cnt[bkt], 1u << lg);
  }
  __syncthreads();
  if (tid == 0) {
    unsigned int o = 0u;
    for (int q = 0; q < 32; ++q) {
      s_off[q] = o;
      o += s_cnt[q];
    }
    plan[0] = o;
  }
  __syncthreads();
  for (int i = tid; i < nw; i += 256) {
    const int lg = parts_of(w[i]);
    const unsigned int pw = w[i] >> lg;
    const int bkt = 31 - (pw ? 31 - __builtin_clz(pw) : 0);
    const unsigned int at = atomicAdd(&s_off[bkt], 1u << lg);
    for (int q = 0; q < (1 << lg); ++q) plan[1 + at + q] = ((unsigned int)i << 8) | ((unsigned int)q << 4) | (unsigned int)lg;
  }
}


// Insertion of a packed key ((float bits of d) << 32 | original index; d >= 0, so the integer order IS the lexicographic
// (distance, index) order) into an ascending register list: a compare and two 64-bit selects per slot.
template <int K>
__device__ __forceinline__ void topk_insert_key(unsigned long long (&L)[K], unsigned long long c) {
#pragma unroll
  for (int t = 0; t < K; ++t) {
    const bool lt = c < L[t];
    const unsigned long long lo = lt ? c : L[t];
    c = lt ? L[t] : c;
    L[t] = lo;
  }
}

// self k-NN, pruned, fused with the covariance estimation of APD:366-407 (covariance_from_list).
// grid: (ceil(n_spad / 256), 1, clouds), block 256.
template <int K>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) void knn_pruned_kernel(const KnnJob* __restrict__ jobs) {
  const KnnJob& job = jobs[blockIdx.z];
  const SearchIndex& si = job.cloud.idx;
  const int n = si.n;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (blockIdx.x * 256 >= n) return;
  const int lane = threadIdx.x & 63;
  if (job.redo && job.redo[(blockIdx.x * 256 + threadIdx.x) >> 6] == 0) return;  // this wave was done by knn_select_kernel
  const int pq = p < n ? p : n - 1;
  const float qx = si.sx[pq], qy = si.sy[pq], qz = si.sz[pq];
  const float qlo[3] = {wave_min(qx), wave_min(qy), wave_min(qz)};
  const float qhi[3] = {wave_max(qx), wave_max(qy), wave_max(qz)};
  const scalar_fp tx = as_scalar(si.sx);
  const scalar_fp ty = as_scalar(si.sy);
  const scalar_fp tz = as_scalar(si.sz);
  const scalar_ip to = (scalar_ip)si.orig;
  const float4* __restrict__ tb4 = reinterpret_cast<const float4*>(si.tbox);
  __shared__ float s_d[32][256];  // distances of the current tile, one column per lane
  STAT_ADD(0, 1);
  const __attribute__((address_space(1))) int* og = (const __attribute__((address_space(1))) int*)si.orig;
  unsigned long long L[K];  // ascending (distance, original index) keys
#pragma unroll
  for (int t = 0; t < K; ++t) L[t] = 0x7f8000007fffffffull;  // (+inf, int max)
  auto kth = [&]() -> float { return __uint_as_float((unsigned int)(L[K - 1] >> 32)); };
  const int ng = (si.n_tiles + 63) / 64;
  const int own_tile = __builtin_amdgcn_readfirstlane((blockIdx.x * 256 + (threadIdx.x & ~63)) / 32);
  const int g0 = own_tile / 64;
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
    // in the own group take the own tiles first (they fill the list with true neighbours), then the rest
    for (int phase = (g == g0 ? 0 : 1); phase < 2; ++phase) {
      const float wb = wave_max(kth());
      unsigned long long mask = __ballot(box_box_bound(qlo, qhi, lo, hi) <= wb);
      STAT_ADD(1, __builtin_popcountll(mask));
      if (g == g0) {
        const int ol = own_tile - g * 64;
        const unsigned long long near = (((ol + 3) >= 64) ? ~0ull : ((1ull << (ol + 3)) - 1ull)) & ~((ol >= 1) ? ((1ull << (ol - 1)) - 1ull) : 0ull);  // tiles ol-1 .. ol+2
        mask = phase == 0 ? (mask & near) : (mask & ~near);
      }
      while (mask) {
        const int tlane = __builtin_ctzll(mask);
        mask &= mask - 1;
        const float bx[8] = {lane_f(lo.x, tlane), lane_f(lo.y, tlane), lane_f(lo.z, tlane), 0.f, lane_f(hi.x, tlane), lane_f(hi.y, tlane), lane_f(hi.z, tlane), 0.f};
        if (__ballot(box_bound(qx, qy, qz, bx) <= kth()) == 0) continue;
        const int j0 = (g * 64 + tlane) * 32;
        // Pass 1 (every lane, no divergence): the 32 distances go to a lane-private LDS column and each lane marks the candidates
        // that beat its current k-th key.  Pass 2: the wave inserts marked candidates one per lane per round -- max-over-lanes
        // rounds instead of one (divergent, 20-step) insertion per candidate that ANY lane accepts.  The k smallest
        // (distance, index) keys do not depend on the order of insertion, so the lists are unchanged.
        STAT_ADD(2, 1);
        const unsigned long long kkey = L[K - 1];
        unsigned int marks = 0u;
#pragma unroll
        for (int gg = 0; gg < 32; gg += 8) {
          float d[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) d[u] = sqdist3(qx, qy, qz, tx[j0 + gg + u], ty[j0 + gg + u], tz[j0 + gg + u]);
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const unsigned long long key = ((unsigned long long)__float_as_uint(d[u]) << 32) | (unsigned int)to[j0 + gg + u];
            s_d[gg + u][threadIdx.x] = d[u];
            if (key < kkey) marks |= 1u << (gg + u);
          }
        }
        while (__ballot(marks != 0u)) {
          STAT_ADD(3, 1);
          if (marks != 0u) {
            const int u = __builtin_ctz(marks);
            marks &= marks - 1u;
            const float cd = s_d[u][threadIdx.x];
            const int ci = og[j0 + u];  // 32 consecutive ints of the tile: one cache line for the whole wave
            const unsigned long long key = ((unsigned long long)__float_as_uint(cd) << 32) | (unsigned int)ci;
            if (key < L[K - 1]) topk_insert_key<K>(L, key);
          }
        }
      }
    }
  }
  if (p < n) {
    float bd[K];
    int bi[K];
#pragma unroll
    for (int t = 0; t < K; ++t) {
      bd[t] = __uint_as_float((unsigned int)(L[t] >> 32));
      bi[t] = (int)(unsigned int)(L[t] & 0xffffffffull);
    }
    covariance_from_list<K>(job, si.orig[p], bd, bi);
  }  // lists never leave the registers: no partial lists, no second kernel
}

// ----------------------------------------------------------------------------------------------- self k-NN by SELECTION
//
// knn_pruned_kernel above keeps a sorted (distance, index) list per lane and pays a 20-slot 64-bit insertion (~105 VALU instructions)
// for every candidate ANY lane of the wave accepts, tile by tile: it runs at a quarter of the VALU issue rate (divergent rounds,
// LDS round trips) and 78 % of what it does issue is insertion.  The two kernels below find the same lists by selection:
//   knn_kth_kernel      the k-th smallest DISTANCE of every query, exactly: a sorted list of K floats per lane; a candidate c of a needed
//                       tile is inserted from the back, slot t becoming med3(D[t-1], c, D[t]) of the OLD neighbours: one instruction per
//                       slot, no carry chain, no compare, no divergence, no LDS; candidates no lane can use are skipped for the whole
//                       wave.  The multiset of the K smallest distances does not need the indices.  (v_med3 / v_min / v_max / v_cmp /
//                       v_cndmask all issue at HALF rate on gfx950 -- tools/valu_rate.hip -- so a 64-bit (distance, index) insertion
//                       costs 5 half-rate instructions per slot where this costs one.)
//   knn_collect_kernel  walks the tiles again with the now exact bound d_k; every candidate with d <= d_k (at most K - 1 below d_k plus
//                       the ties at d_k) is appended to a lane-private LDS column as a packed (distance, original index) key; the
//                       <= CAP buffered keys are then sorted into the register list with the 64-bit insertion (dense: every lane has
//                       ~K of them), which also resolves ties on the original index, and the covariance is formed from the list
//                       (covariance_from_list) as before.
// A wave in which some lane meets more than CAP candidates at or below its d_k (only with many exactly equal distances: lattices,
// duplicated points) flags itself in job.redo and is redone by knn_pruned_kernel, so the result is exact for every input.
// grid of both: (ceil(n_spad / qpw), 1, clouds), block kKnnBlock (one wave: the waves never cooperate, see nn_search_pruned_kernel).
// qpw = KnnJob::qpw queries per wave (64, 32, 16 or 8; the other lanes idle).  A wave walks the UNION of the tiles its queries need, and
// its life grows with that union; when the call has fewer 64-query waves than the chip has SIMDs (a lone pair of scans) smaller groups
// give more, shorter waves: a lone 5 k x 5 k align 0.79 -> 0.73 ms.  (It does not shorten the slow waves of a big map -- 44 of the 1564
// waves of a 100 k-point map take 2.5 x the average with 3 x the tiles and insertions, whatever the group size: profiles/r03/experiments.md.)
#ifndef GORIO_KNN_BLOCK
#define GORIO_KNN_BLOCK 64
#endif
constexpr int kKnnBlock = GORIO_KNN_BLOCK;

template <int K>
__global__ __launch_bounds__(kKnnBlock) void knn_kth_kernel(const KnnJob* __restrict__ jobs) {
  const GridPos gp = xcd_grid_pos();  // cloud -> XCD (apd_device.h)
  const KnnJob& job = jobs[gp.z];
  const SearchIndex& si = job.cloud.idx;
  const int n = si.n;
  const int qpw = job.qpw;  // queries per wave (see KnnJob): lanes qpw .. 63 idle
  const int base = gp.x * qpw;
  if (base >= n) return;
  const int lane = threadIdx.x & 63;
  const bool live = lane < qpw;
  const int p = base + (live ? lane : 0);
  const int pq = p < n ? p : n - 1;
  const float qx = si.sx[pq], qy = si.sy[pq], qz = si.sz[pq];
  const float qlo[3] = {wave_fmin_u(qx), wave_fmin_u(qy), wave_fmin_u(qz)};  // DPP reductions, wave-uniform results
  const float qhi[3] = {wave_fmax_u(qx), wave_fmax_u(qy), wave_fmax_u(qz)};
  const scalar_fp tx = as_scalar(si.sx);
  const scalar_fp ty = as_scalar(si.sy);
  const scalar_fp tz = as_scalar(si.sz);
  const float4* __restrict__ tb4 = reinterpret_cast<const float4*>(si.tbox);
  const int ng = (si.n_tiles + 63) / 64;
  const int own_tile = __builtin_amdgcn_readfirstlane(base / 32);
  const int g0 = own_tile / 64;
  if (lane == 0) job.redo[base >> 6] = 0;  // knn_collect_kernel raises it; with qpw < 64 several waves share one flag
  float D[K];  // the K smallest distances met so far, ascending.  An idle lane holds -inf throughout: it needs no tile, accepts no candidate
#pragma unroll
  for (int t = 0; t < K; ++t) D[t] = live ? INFINITY : -INFINITY;
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
    for (int phase = (g == g0 ? 0 : 1); phase < 2; ++phase) {  // own tiles first: they fill the list with near neighbours
      const float wb = wave_fmax_u(D[K - 1]);
      unsigned long long mask = __ballot(box_box_bound(qlo, qhi, lo, hi) < wb);  // strictly below: an equal distance changes no distance list
      if (g == g0) {
        const int ol = own_tile - g * 64;
        const unsigned long long near = (((ol + 3) >= 64) ? ~0ull : ((1ull << (ol + 3)) - 1ull)) & ~((ol >= 1) ? ((1ull << (ol - 1)) - 1ull) : 0ull);  // tiles ol-1 .. ol+2
        mask = phase == 0 ? (mask & near) : (mask & ~near);
      }
      while (mask) {
        const int tlane = __builtin_ctzll(mask);
        mask &= mask - 1;
        const TileBox bx = TileBox{lane_f(lo.x, tlane), lane_f(lo.y, tlane), lane_f(lo.z, tlane), 0.f, lane_f(hi.x, tlane), lane_f(hi.y, tlane), lane_f(hi.z, tlane), 0.f};
        if (__ballot(box_bound_s(qx, qy, qz, bx) < D[K - 1]) == 0) continue;
        const int j0 = (g * 64 + tlane) * 32;
#pragma unroll 2
        for (int gg = 0; gg < 32; gg += 8) {
          float d[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) d[u] = sqdist3(qx, qy, qz, tx[j0 + gg + u], ty[j0 + gg + u], tz[j0 + gg + u]);
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const float c = d[u];
            if (__ballot(c < D[K - 1]) == 0) continue;  // no lane's list changes: skip the insertion for the whole wave
            // insertion from the back: the new slot t is the median of (old D[t-1], c, old D[t]) -- every slot from OLD values, so the
            // K operations are independent (no carry chain); descending t keeps D[t-1] old when slot t is written
#pragma unroll
            for (int t = K - 1; t > 0; --t) D[t] = __builtin_amdgcn_fmed3f(D[t - 1], c, D[t]);
            D[0] = __builtin_amdgcn_fmed3f(D[0], c, -INFINITY);  // min(D[0], c) without the canonicalisation fminf would add
          }
        }
      }
    }
  }
  if (live) job.kth[p] = D[K - 1];  // indexed by SORTED position (n_spad entries)
}

template <int K>
__global__ __launch_bounds__(kKnnBlock) void knn_collect_kernel(const KnnJob* __restrict__ jobs) {
  const GridPos gp = xcd_grid_pos();  // cloud -> XCD (apd_device.h)
  const KnnJob& job = jobs[gp.z];
  const SearchIndex& si = job.cloud.idx;
  const int n = si.n;
  const int qpw = job.qpw;
  const int base = gp.x * qpw;
  if (base >= n) return;
  const int lane = threadIdx.x & 63;
  const bool live = lane < qpw;
  const int p = base + (live ? lane : 0);
  const int pq = p < n ? p : n - 1;
  const float qx = si.sx[pq], qy = si.sy[pq], qz = si.sz[pq];
  const float qlo[3] = {wave_fmin_u(qx), wave_fmin_u(qy), wave_fmin_u(qz)};  // DPP reductions, wave-uniform results
  const float qhi[3] = {wave_fmax_u(qx), wave_fmax_u(qy), wave_fmax_u(qz)};
  const scalar_fp tx = as_scalar(si.sx);
  const scalar_fp ty = as_scalar(si.sy);
  const scalar_fp tz = as_scalar(si.sz);
  const scalar_ip to = (scalar_ip)si.orig;
  const float4* __restrict__ tb4 = reinterpret_cast<const float4*>(si.tbox);
  constexpr int CAP = K + 4;
  __shared__ unsigned long long s_buf[CAP][kKnnBlock];  // keys at or below d_k, one column per lane (bank = lane: conflict free)
  const int ng = (si.n_tiles + 63) / 64;
  const float dk = live ? job.kth[p] : -INFINITY;  // an idle lane needs no tile and keeps no candidate
  STAT_ADD(0, 1);
  int cnt = 0;
  {
    const float wb = wave_fmax_u(dk);
    for (int g = 0; g < ng; ++g) {
      const int tl = g * 64 + lane;
      float4 lo = make_float4(INFINITY, INFINITY, INFINITY, 0.f), hi = make_float4(-INFINITY, -INFINITY, -INFINITY, 0.f);
      if (tl < si.n_tiles) {
        lo = tb4[2 * (size_t)tl];
        hi = tb4[2 * (size_t)tl + 1];
      }
      unsigned long long mask = __ballot(box_box_bound(qlo, qhi, lo, hi) <= wb);
      while (mask) {
        const int tlane = __builtin_ctzll(mask);
        mask &= mask - 1;
        const TileBox bx = TileBox{lane_f(lo.x, tlane), lane_f(lo.y, tlane), lane_f(lo.z, tlane), 0.f, lane_f(hi.x, tlane), lane_f(hi.y, tlane), lane_f(hi.z, tlane), 0.f};
        const unsigned long long needb = __ballot(box_bound_s(qx, qy, qz, bx) <= dk);
        if (needb == 0) continue;
        STAT_ADD(1, 1);
        STAT_ADD(2, __builtin_popcountll(needb));
        const int j0 = (g * 64 + tlane) * 32;
#pragma unroll
        for (int gg = 0; gg < 32; gg += 8) {
          float d[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) d[u] = sqdist3(qx, qy, qz, tx[j0 + gg + u], ty[j0 + gg + u], tz[j0 + gg + u]);
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            if (d[u] <= dk) {
              if (cnt < CAP) s_buf[cnt][threadIdx.x] = ((unsigned long long)__float_as_uint(d[u]) << 32) | (unsigned int)to[j0 + gg + u];
              ++cnt;
            }
          }
        }
      }
    }
  }
  const bool redo = __ballot(cnt > CAP) != 0ull;
  if (redo) {
    if (lane == 0) job.redo[base >> 6] = 1;  // cleared by knn_kth_kernel; knn_pruned_kernel redoes the 64 sorted positions of the flag
    return;
  }
  // sort the buffered keys (ties fall to the lower original index, surplus ties beyond K drop off the end)
  unsigned long long L[K];
#pragma unroll
  for (int t = 0; t < K; ++t) L[t] = 0x7f8000007fffffffull;  // (+inf, int max)
  int rounds = cnt;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) rounds = max(rounds, __shfl_xor(rounds, o, 64));
  for (int r = 0; r < rounds; ++r) {
    if (r < cnt) topk_insert_key<K>(L, s_buf[r][threadIdx.x]);
  }
  if (live && p < n) {
    float bd[K];
    int bi[K];
#pragma unroll
    for (int t = 0; t < K; ++t) {
      bd[t] = __uint_as_float((unsigned int)(L[t] >> 32));
      bi[t] = (int)(unsigned int)(L[t] & 0xffffffffull);
    }
    covariance_from_list<K>(job, si.orig[p], bd, bi);
  }
}

}  // namespace gorio
