// apd_api.hip -- host side of the APD-GICP C ABI declared in include/gorio_apd.h.
//
// One gorio_apd handle == one fast_gicp::FastAPDGICP object (APDH:20-122): it owns the device copies of the two clouds,
// their covariances, the per-source-point correspondence state and the device-resident optimiser state, and it drives
// the kernels of apd_kernels.hip.  No CPU compute path exists here: without a HIP device every entry point fails.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_set>
#include <vector>

#include "../../include/gorio_apd.h"
#include "apd_kernels.hip"
#include "apd_index.hip"
#include "apd_submap.hip"
#include "apd_prep.hip"
#include "../../include/gorio_prep.h"

using namespace gorio;

namespace {

constexpr int kPad = 16;
constexpr float kFar = 1e30f;

struct DevCloud {
  float *x = nullptr, *y = nullptr, *z = nullptr, *label = nullptr;
  float4* p4 = nullptr;
  double *cov6 = nullptr, *geo_w = nullptr;
  int* knn = nullptr;      // [n][k] neighbour indices of the last covariance computation (parity hook, params.keep_knn_indices)
  int* redo = nullptr;     // per query wave: redo flags of knn_collect_kernel
  float* kth = nullptr;    // per sorted position: k-th distance (knn_kth_kernel)
  int redo_cap = 0;
  float* part_d = nullptr; // k-NN partial lists
  int* part_i = nullptr;
  size_t part_cap = 0;     // elements
  int knn_k = 0;
  bool knn_valid = false;  // knn holds the lists of the current covariances
  int n = 0, n_pad = 0, cap = 0;
  bool present = false;
  int cov_count = 0;       // == source_covs_.size(): n when valid, 0 when stale
  int cov_k = -1, cov_reg = -1;  // k_correspondences / regularization the covariances were ESTIMATED with; -1: supplied by set_*_covariances
  // exact search accelerator (GORIO_SEARCH_PRUNED)
  SearchIndex idx = SearchIndex{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0};
  unsigned long long* keys = nullptr;
  unsigned int* bb = nullptr;
  int idx_cap = 0, keys_cap = 0;
  bool idx_valid = false;
  int device = 0;
  CloudView view() const { return CloudView{x, y, z, label, p4, cov6, geo_w, n, n_pad, idx}; }
  DevCloud() = default;
  DevCloud(const DevCloud&) = delete;
  DevCloud& operator=(const DevCloud&) = delete;
  ~DevCloud();  // frees the device buffers: a cloud may be shared by several handles (gorio_apd_set_target_shared) and lives until the last one lets go
};

}  // namespace

struct gorio_apd {
  int device = 0;
  hipStream_t stream = nullptr;
  gorio_apd_params params;
  std::shared_ptr<DevCloud> src, tgt;  // never null; tgt may be shared with other handles of the device
  // per-source-point state
  unsigned long long* best_key = nullptr;
  int* seed = nullptr;     // warm start of the pruned search, by sorted source position (PairDesc::seed)
  unsigned int* nn_work = nullptr;  // PairDesc::nn_work / nn_plan (measured work of the query waves, plan of the next searches)
  unsigned int* nn_plan = nullptr;
  int nn_wcap = 0;
  int align_budget = 0;    // loop iterations the previous batch led by this handle needed (0 = none yet): enqueued before the first look at the done flags
  int* corr = nullptr;
  float* sqd = nullptr;
  double* omega6 = nullptr;
  double* partials = nullptr;
  int pt_cap = 0;
  bool corr_valid = false;
  bool omega_valid = false;  // omega6 holds the Mahalanobis matrices of the last linearisation (not after a Gauss-Newton align)
  PairState* d_state = nullptr;
  PairDesc* d_desc = nullptr;   // batch descriptor array (owned by the handle that leads a batch)
  int desc_cap = 0;
  PairState* d_states_batch = nullptr;
  KnnJob* d_jobs = nullptr;
  size_t fit_cap = 0;           // doubles in d_fit
  void* d_copy_jobs = nullptr;  // gorio_apd_set_clouds_device_batch
  size_t copy_jobs_cap = 0;
  int jobs_cap = 0;
  IndexJob* d_ijobs = nullptr;
  int ijobs_cap = 0;
  double* d_fit = nullptr;
  // scan-to-submap assembly scratch (gorio_apd_set_target_submap)
  float4* d_sub_in = nullptr;
  float4* d_sub_out = nullptr;
  float4* d_sub_vox = nullptr;
  size_t sub_cap = 0;
  unsigned long long* d_sub_keys = nullptr;
  size_t sub_keys_cap = 0;
  int* d_sub_counts = nullptr;
  size_t sub_counts_cap = 0;
  SubmapFrame* d_sub_frames = nullptr;
  int sub_frames_cap = 0;
  unsigned int* d_sub_bb = nullptr;
  IndexJob* d_sub_job = nullptr;
  // sharded-source mode (gorio_apd_comm_init): RCCL communicator over the ranks that share one source cloud
  ncclComm_t comm = nullptr;
  int comm_world = 1, comm_rank = 0;
  bool shard_only = false;  // gorio_apd_debug_set_shard: the source partition of a rank without the collectives (test hook)
  bool fuse_step = true, plan_search = true;  // gorio_apd_debug_set_schedule
  double* d_red = nullptr;  // [32] all-reduce buffer: 28 sums of a linearisation, [28] trial error, [29] scratch
  long long allreduce_count = 0;  // ncclAllReduce calls enqueued through this handle's communicator (gorio_apd_comm_info)
  // pinned host staging of the small descriptor arrays a call uploads (index jobs, k-NN jobs, pair descriptors / states, copy jobs): a
  // copy from pageable memory needs a stream drain before the vector it came from may die; a pinned buffer of the handle's own needs none
  struct Pinned {
    void* p = nullptr;
    size_t cap = 0;
    hipEvent_t ev = nullptr;  // recorded behind the last upload from this buffer: the next writer waits for it
    bool pending = false;
  };
  Pinned pin_ijobs, pin_jobs, pin_desc, pin_states, pin_copy;
  std::string err;
  // profiling
  bool profiling = false;
  double stage_s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int stage_n[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  struct EvPair { hipEvent_t start, stop; int stage; int prev; };  // prev >= 0: the span starts at ev_pool[prev].stop (StageChain)
  std::vector<EvPair> ev_pool;
  size_t ev_used = 0;
};

namespace {

// All handles of one device share ONE launch stream: setInput* copies, index builds and align batches of different handles are
// then ordered by the stream itself and a batch needs no cross-stream synchronisation.  (UGPM uses its own stream and overlaps.)
hipStream_t device_stream(int device) {
  static std::mutex mu;
  static std::vector<hipStream_t> streams;
  std::lock_guard<std::mutex> lock(mu);
  if ((int)streams.size() <= device) streams.resize(device + 1, nullptr);
  if (!streams[device]) {
    if (hipStreamCreateWithFlags(&streams[device], hipStreamNonBlocking) != hipSuccess) return nullptr;
  }
  return streams[device];
}

int fail(gorio_apd* h, int code, const std::string& msg) {
  if (h) h->err = msg;
  return code;
}

#define HIP_TRY(h, expr)                                                                                        \
  do {                                                                                                          \
    hipError_t e_ = (expr);                                                                                     \
    if (e_ != hipSuccess) return fail(h, GORIO_ERR_NO_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

int roundup(int v, int m) { return (v + m - 1) / m * m; }

// Upload `bytes` of a short-lived host array through the handle's pinned staging buffer `b`: no stream drain, the caller's array may die
// right away.  A buffer that still feeds an earlier upload is waited for first (its event), never overwritten under it.
int upload_staged(gorio_apd* h, gorio_apd::Pinned& b, void* dst, const void* src, size_t bytes) {
  if (b.pending) {
    HIP_TRY(h, hipEventSynchronize(b.ev));
    b.pending = false;
  }
  if (bytes > b.cap) {
    if (b.p) hipHostFree(b.p);
    b.p = nullptr;
    b.cap = 0;
    HIP_TRY(h, hipHostMalloc(&b.p, bytes + bytes / 2 + 256, hipHostMallocDefault));
    b.cap = bytes + bytes / 2 + 256;
  }
  if (!b.ev) HIP_TRY(h, hipEventCreateWithFlags(&b.ev, hipEventDisableTiming));
  std::memcpy(b.p, src, bytes);
  HIP_TRY(h, hipMemcpyAsync(dst, b.p, bytes, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipEventRecord(b.ev, h->stream));
  b.pending = true;
  return GORIO_OK;
}

// RCCL is loaded on first use (dlopen): a process that never shards a source never needs librccl, and one that already has a copy
// mapped (PyTorch ships its own) keeps using that copy.
struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};
Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (r.lib) break;
    }
    if (!r.lib) return;
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.lib, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.lib, "ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.lib, "ncclCommDestroy"));
    r.CommCount = reinterpret_cast<decltype(r.CommCount)>(dlsym(r.lib, "ncclCommCount"));
    r.CommUserRank = reinterpret_cast<decltype(r.CommUserRank)>(dlsym(r.lib, "ncclCommUserRank"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(r.lib, "ncclAllReduce"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.lib, "ncclGetErrorString"));
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce && r.GetErrorString;
  });
  return r;
}

#define NCCL_TRY(h, expr)                                                                                              \
  do {                                                                                                                 \
    ncclResult_t e_ = (expr);                                                                                          \
    if (e_ != ncclSuccess) return fail(h, GORIO_ERR_NO_DEVICE, std::string(#expr) + ": " + rccl().GetErrorString(e_)); \
  } while (0)

}  // namespace
DevCloud::~DevCloud() {
  DevCloud& c = *this;
  hipSetDevice(c.device);
  hipFree(c.idx.sx); hipFree(c.idx.sy); hipFree(c.idx.sz); hipFree(c.idx.orig); hipFree(c.idx.s4); hipFree(c.idx.tbox); hipFree(c.idx.sbox); hipFree(c.idx.bbox); hipFree(c.keys); hipFree(c.bb);
  hipFree(c.x); hipFree(c.y); hipFree(c.z); hipFree(c.label); hipFree(c.p4); hipFree(c.cov6); hipFree(c.geo_w); hipFree(c.knn); hipFree(c.part_d); hipFree(c.part_i); hipFree(c.redo); hipFree(c.kth);
}
namespace {

// a cloud about to be overwritten must not be one other handles still look at (gorio_apd_set_target_shared): detach first
void make_private(gorio_apd* h, std::shared_ptr<DevCloud>& c) {
  if (c.use_count() > 1) {
    c = std::make_shared<DevCloud>();
    c->device = h->device;
  }
}

int ensure_cloud(gorio_apd* h, DevCloud& c, int n) {
  const int n_pad = roundup(n, kPad) + kPad;
  if (n_pad > c.cap) {
    hipFree(c.x); hipFree(c.y); hipFree(c.z); hipFree(c.label); hipFree(c.p4); hipFree(c.cov6); hipFree(c.geo_w); hipFree(c.knn);
    c.x = c.y = c.z = c.label = nullptr; c.p4 = nullptr; c.cov6 = c.geo_w = nullptr; c.knn = nullptr; c.knn_k = 0;
    const int cap = n_pad + n_pad / 8;
    HIP_TRY(h, hipMalloc(&c.x, sizeof(float) * cap));
    HIP_TRY(h, hipMalloc(&c.y, sizeof(float) * cap));
    HIP_TRY(h, hipMalloc(&c.z, sizeof(float) * cap));
    HIP_TRY(h, hipMalloc(&c.label, sizeof(float) * cap));
    HIP_TRY(h, hipMalloc(&c.p4, sizeof(float4) * cap));
    HIP_TRY(h, hipMalloc(&c.cov6, sizeof(double) * 6 * cap));
    HIP_TRY(h, hipMalloc(&c.geo_w, sizeof(double) * cap));
    c.cap = cap;
  }
  c.n = n;
  c.n_pad = n_pad;
  return GORIO_OK;
}

int ensure_points(gorio_apd* h, int n) {
  if (n > h->pt_cap) {
    hipFree(h->best_key); hipFree(h->seed); hipFree(h->nn_work); hipFree(h->nn_plan); hipFree(h->corr); hipFree(h->sqd); hipFree(h->omega6); hipFree(h->partials);
    h->best_key = nullptr; h->seed = nullptr; h->nn_work = nullptr; h->nn_plan = nullptr; h->corr = nullptr; h->sqd = nullptr; h->omega6 = nullptr; h->partials = nullptr;
    const int cap = n + n / 8 + 256;
    HIP_TRY(h, hipMalloc(&h->best_key, sizeof(unsigned long long) * cap));
    HIP_TRY(h, hipMalloc(&h->seed, sizeof(int) * (cap + 512)));  // indexed by sorted position < roundup(n, 512)
    h->nn_wcap = (cap + 512) / 64 + 1;
    HIP_TRY(h, hipMalloc(&h->nn_work, sizeof(unsigned int) * 2 * h->nn_wcap));
    HIP_TRY(h, hipMalloc(&h->nn_plan, sizeof(unsigned int) * (1 + 16 * (size_t)h->nn_wcap)));
    HIP_TRY(h, hipMemsetAsync(h->nn_work, 0, sizeof(unsigned int) * 2 * h->nn_wcap, h->stream));
    HIP_TRY(h, hipMemsetAsync(h->nn_plan, 0, sizeof(unsigned int) * (1 + 16 * (size_t)h->nn_wcap), h->stream));
    HIP_TRY(h, hipMalloc(&h->corr, sizeof(int) * cap));
    HIP_TRY(h, hipMalloc(&h->sqd, sizeof(float) * cap));
    HIP_TRY(h, hipMalloc(&h->omega6, sizeof(double) * 6 * cap));
    HIP_TRY(h, hipMalloc(&h->partials, sizeof(double) * 28 * (cap / 256 + 2)));
    h->pt_cap = cap;
  }
  return GORIO_OK;
}

// host strided AoS -> device SoA (+ padding)
int upload_cloud(gorio_apd* h, DevCloud& c, const float* xyz, const float* label, int n, int stride_bytes) {
  if (!xyz || n <= 0 || stride_bytes < 12 || (stride_bytes % 4) != 0) return fail(h, GORIO_ERR_INVALID, "set_input: bad cloud arguments");
  HIP_TRY(h, hipSetDevice(h->device));
  int rc = ensure_cloud(h, c, n);
  if (rc) return rc;
  const int np = c.n_pad;
  std::vector<float> buf((size_t)np * 8);
  float *bx = buf.data(), *by = bx + np, *bz = by + np, *bl = bz + np, *b4 = bl + np;
  const int st = stride_bytes / 4;
  for (int i = 0; i < n; ++i) {
    const float* p = xyz + (size_t)i * st;
    bx[i] = p[0]; by[i] = p[1]; bz[i] = p[2];
    bl[i] = label ? label[(size_t)i * st] : 0.0f;
  }
  for (int i = n; i < np; ++i) { bx[i] = kFar; by[i] = kFar; bz[i] = kFar; bl[i] = 0.0f; }
  for (int i = 0; i < np; ++i) { b4[4 * (size_t)i] = bx[i]; b4[4 * (size_t)i + 1] = by[i]; b4[4 * (size_t)i + 2] = bz[i]; b4[4 * (size_t)i + 3] = bl[i]; }
  HIP_TRY(h, hipMemcpyAsync(c.x, bx, sizeof(float) * np, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipMemcpyAsync(c.y, by, sizeof(float) * np, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipMemcpyAsync(c.z, bz, sizeof(float) * np, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipMemcpyAsync(c.label, bl, sizeof(float) * np, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipMemcpyAsync(c.p4, b4, sizeof(float4) * np, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  c.present = true;
  c.cov_count = 0;
  c.idx_valid = false;
  return GORIO_OK;
}

// setInput* from device-resident SoA buffers: one launch copies the four arrays and writes the padding
__global__ __launch_bounds__(256) void copy_cloud_kernel(const float* __restrict__ sx, const float* __restrict__ sy, const float* __restrict__ sz, const float* __restrict__ sl,
                                                         float* __restrict__ x, float* __restrict__ y, float* __restrict__ z, float* __restrict__ label, float4* __restrict__ p4, int n, int n_pad) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_pad) return;
  const bool in = i < n;
  const float4 v = make_float4(in ? sx[i] : 1e30f, in ? sy[i] : 1e30f, in ? sz[i] : 1e30f, (in && sl) ? sl[i] : 0.0f);
  x[i] = v.x;
  y[i] = v.y;
  z[i] = v.z;
  label[i] = v.w;
  p4[i] = v;
}

int upload_cloud_device(gorio_apd* h, DevCloud& c, const float* dx, const float* dy, const float* dz, const float* dl, int n) {
  if (!dx || !dy || !dz || n <= 0) return fail(h, GORIO_ERR_INVALID, "set_input_device: bad cloud arguments");
  HIP_TRY(h, hipSetDevice(h->device));
  int rc = ensure_cloud(h, c, n);
  if (rc) return rc;
  copy_cloud_kernel<<<(c.n_pad + 255) / 256, 256, 0, h->stream>>>(dx, dy, dz, dl, c.x, c.y, c.z, c.label, c.p4, n, c.n_pad);
  HIP_TRY(h, hipGetLastError());
  c.present = true;
  c.cov_count = 0;
  c.idx_valid = false;
  return GORIO_OK;
}

// the same for many clouds in ONE launch (grid.y = cloud): a step of the batched pipeline re-targets 2 x pairs clouds, and 128
// five-microsecond launches in a row are 0.7 ms of stream time
struct CopyJob {
  const float* sx; const float* sy; const float* sz; const float* sl;
  float* x; float* y; float* z; float* label;
  float4* p4;
  int n, n_pad;
};
__global__ __launch_bounds__(256) void copy_clouds_kernel(const CopyJob* __restrict__ jobs) {
  const CopyJob jb = jobs[blockIdx.y];
  for (int i = blockIdx.x * 256 + threadIdx.x; i < jb.n_pad; i += gridDim.x * 256) {
    const bool in = i < jb.n;
    const float4 v = make_float4(in ? jb.sx[i] : 1e30f, in ? jb.sy[i] : 1e30f, in ? jb.sz[i] : 1e30f, (in && jb.sl) ? jb.sl[i] : 0.0f);
    jb.x[i] = v.x;
    jb.y[i] = v.y;
    jb.z[i] = v.z;
    jb.label[i] = v.w;
    jb.p4[i] = v;
  }
}

ApdConsts make_consts(const gorio_apd_params& p) {  // inv_n_scale is patched per launch set by cl_scale()
  ApdConsts c;
  c.thr2 = p.corr_dist_threshold * p.corr_dist_threshold;
  c.dist_var = p.dist_var;
  c.sin_az = std::sin(p.azimuth_var / 180 * M_PI);
  c.sin_el = std::sin(p.elevation_var / 180 * M_PI);
  c.rot_eps = p.rotation_epsilon;
  c.trans_eps = p.transformation_epsilon;
  c.lm_init_lambda_factor = p.lm_init_lambda_factor;
  c.inv_n_scale = 1.0;
  c.optimizer = p.optimizer;
  c.lm_max_iterations = p.lm_max_iterations;
  c.max_iterations = p.max_iterations;
  c.pad_ = 0;
  return c;
}

// Stage timing with HIP events on the launch stream.  Events are only RECORDED while kernels are being enqueued (no host
// synchronisation inside the loop); resolve_stage_events() turns them into seconds after the stream has drained.
struct StageTimer {
  gorio_apd* h;
  bool on;
  size_t slot;
  StageTimer(gorio_apd* h_, int s) : h(h_), on(h_->profiling), slot(0) {
    if (!on) return;
    if (h->ev_used == h->ev_pool.size()) {
      hipEvent_t a = nullptr, b = nullptr;
      if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { on = false; return; }
      h->ev_pool.push_back({a, b, s, -1});
    }
    slot = h->ev_used++;
    h->ev_pool[slot].stage = s;
    h->ev_pool[slot].prev = -1;
    hipEventRecord(h->ev_pool[slot].start, h->stream);
  }
  ~StageTimer() {
    if (on) hipEventRecord(h->ev_pool[slot].stop, h->stream);
  }
};

// Back-to-back kernels of the optimiser loop: ONE event between two kernels closes the span of the first and opens the span of the
// second (two records per kernel cost the loop about 10 % in launch gaps).  A span then includes the launch gap before its kernel.
struct StageChain {
  gorio_apd* h;
  bool on;
  int last = -1;
  int take(int stage) {
    if (h->ev_used == h->ev_pool.size()) {
      hipEvent_t a = nullptr, b = nullptr;
      if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { on = false; return -1; }
      h->ev_pool.push_back({a, b, stage, -1});
    }
    const int q = (int)h->ev_used++;
    h->ev_pool[q].stage = stage;
    h->ev_pool[q].prev = -1;
    return q;
  }
  explicit StageChain(gorio_apd* h_) : h(h_), on(h_->profiling) {
    if (!on) return;
    last = take(-1);  // boundary only, no span
    if (on) hipEventRecord(h->ev_pool[last].stop, h->stream);
  }
  void mark(int stage) {  // call right after the launch(es) of `stage`
    if (!on) return;
    const int q = take(stage);
    if (!on) return;
    h->ev_pool[q].prev = last;
    hipEventRecord(h->ev_pool[q].stop, h->stream);
    last = q;
  }
};

void resolve_stage_events(gorio_apd* h) {
  if (!h->profiling) return;
  for (size_t q = 0; q < h->ev_used; ++q) {
    float ms = 0.f;
    if (h->ev_pool[q].stage < 0) continue;
    const hipEvent_t from = h->ev_pool[q].prev >= 0 ? h->ev_pool[h->ev_pool[q].prev].stop : h->ev_pool[q].start;
    if (hipEventSynchronize(h->ev_pool[q].stop) == hipSuccess && hipEventElapsedTime(&ms, from, h->ev_pool[q].stop) == hipSuccess) {
      h->stage_s[h->ev_pool[q].stage] += ms * 1e-3;
      h->stage_n[h->ev_pool[q].stage] += 1;
    }
  }
  h->ev_used = 0;
}

// Build the search accelerator of every listed cloud that lacks one (batched: one launch per sort stage for all clouds).
int run_index_build(gorio_apd* lead, std::vector<std::pair<gorio_apd*, DevCloud*>>& clouds) {
  std::vector<std::pair<gorio_apd*, DevCloud*>> todo;
  bool small_call = true;  // every cloud named in the call (built now or not) is a scan-sized one: kd chunks of 2048 points (kd_refine_kernel)
  {
    std::unordered_set<DevCloud*> seen;
    for (auto& c : clouds) {
      if (c.second->n > kKdSmallCloud) small_call = false;
      if (!c.second->idx_valid && seen.insert(c.second).second) todo.push_back(c);
    }
  }
  if (todo.empty()) return GORIO_OK;
  const int nj = (int)todo.size();
  std::vector<IndexJob> jobs(nj);
  int max_pow2 = kSortTile, max_spad = 512, max_n = 1;
  for (int q = 0; q < nj; ++q) {
    gorio_apd* h = todo[q].first;
    DevCloud& c = *todo[q].second;
    const int n_spad = roundup(c.n, 512);
    int npow2 = kSortTile;
    while (npow2 < c.n) npow2 <<= 1;
    if (n_spad > c.idx_cap) {
      hipFree(c.idx.sx); hipFree(c.idx.sy); hipFree(c.idx.sz); hipFree(c.idx.orig); hipFree(c.idx.s4); hipFree(c.idx.tbox); hipFree(c.idx.sbox); hipFree(c.idx.bbox);
      c.idx.sx = c.idx.sy = c.idx.sz = nullptr; c.idx.orig = nullptr; c.idx.s4 = nullptr; c.idx.tbox = c.idx.sbox = c.idx.bbox = nullptr;
      const int cap = n_spad + roundup(n_spad / 8, 512);
      HIP_TRY(h, hipMalloc(&c.idx.sx, sizeof(float) * cap));
      HIP_TRY(h, hipMalloc(&c.idx.sy, sizeof(float) * cap));
      HIP_TRY(h, hipMalloc(&c.idx.sz, sizeof(float) * cap));
      HIP_TRY(h, hipMalloc(&c.idx.orig, sizeof(int) * cap));
      HIP_TRY(h, hipMalloc(&c.idx.s4, sizeof(float4) * cap));
      HIP_TRY(h, hipMalloc(&c.idx.tbox, sizeof(float) * 8 * (cap / 32)));
      HIP_TRY(h, hipMalloc(&c.idx.sbox, sizeof(float) * 8 * (cap / 512)));
      HIP_TRY(h, hipMalloc(&c.idx.bbox, sizeof(float) * 8 * (cap / 32768 + 1)));
      c.idx_cap = cap;
    }
    if (npow2 > c.keys_cap) {
      hipFree(c.keys);
      c.keys = nullptr;
      HIP_TRY(h, hipMalloc(&c.keys, sizeof(unsigned long long) * npow2));
      c.keys_cap = npow2;
    }
    if (!c.bb) HIP_TRY(h, hipMalloc(&c.bb, sizeof(unsigned int) * 6));
    c.idx.n = c.n;
    c.idx.n_spad = n_spad;
    c.idx.n_tiles = n_spad / 32;
    c.idx.n_super = n_spad / 512;
    IndexJob& j = jobs[q];
    j.x = c.x; j.y = c.y; j.z = c.z; j.n = c.n; j.npow2 = npow2; j.keys = c.keys; j.bb = c.bb; j.idx = c.idx;
    max_pow2 = std::max(max_pow2, npow2);
    max_spad = std::max(max_spad, n_spad);
    max_n = std::max(max_n, c.n);
  }
  if (nj > lead->ijobs_cap) {
    hipFree(lead->d_ijobs);
    lead->d_ijobs = nullptr;
    HIP_TRY(lead, hipMalloc(&lead->d_ijobs, sizeof(IndexJob) * nj));
    lead->ijobs_cap = nj;
  }
  if (int rc = upload_staged(lead, lead->pin_ijobs, lead->d_ijobs, jobs.data(), sizeof(IndexJob) * nj)) return rc;
  {
    StageTimer t(lead, 4);
    const IndexJob* dj = lead->d_ijobs;
    bbox_init_kernel<<<(nj + 63) / 64, 64, 0, lead->stream>>>(dj, nj);
    bbox_kernel<<<dim3(std::min(64, (max_n + 255) / 256), nj), 256, 0, lead->stream>>>(dj);
    morton_kernel<<<dim3((max_pow2 + 255) / 256, nj), 256, 0, lead->stream>>>(dj);
    bitonic_tile_sort_kernel<<<dim3(max_pow2 / kSortTile, nj), 1024, 0, lead->stream>>>(dj);
    for (int k = 2 * kSortTile; k <= max_pow2; k <<= 1) {
      for (int j = k >> 1; j >= kSortTile; j >>= 1) bitonic_global_kernel<<<dim3((max_pow2 / 2 + 255) / 256, nj), 256, 0, lead->stream>>>(dj, k, j);
      bitonic_tile_merge_kernel<<<dim3(max_pow2 / kSortTile, nj), 1024, 0, lead->stream>>>(dj, k);
    }
    gather_sorted_kernel<<<dim3((max_spad + 255) / 256, nj), 256, 0, lead->stream>>>(dj);
    if (small_call) kd_refine_kernel<2048><<<dim3((max_spad + 2047) / 2048, nj), 512, 0, lead->stream>>>(dj);
    else kd_refine_kernel<4096><<<dim3((max_spad + 4095) / 4096, nj), 1024, 0, lead->stream>>>(dj);
    box_tile_kernel<<<dim3((max_spad / 32 + 255) / 256, nj), 256, 0, lead->stream>>>(dj);
    box_super_kernel<<<dim3((max_spad / 512 + 255) / 256, nj), 256, 0, lead->stream>>>(dj);
    box_block_kernel<<<dim3((max_spad / 32768 + 64) / 64, nj), 64, 0, lead->stream>>>(dj);
  }
  HIP_TRY(lead, hipGetLastError());
  for (auto& t : todo) t.second->idx_valid = true;
  return GORIO_OK;
}

// covariance estimation for a list of clouds on lead's stream
int run_covariances(gorio_apd* lead, std::vector<std::pair<gorio_apd*, DevCloud*>>& todo) {
  if (todo.empty()) return GORIO_OK;
  const int njobs = (int)todo.size();
  const int k = lead->params.k_correspondences;
  const int K = k <= 20 ? 20 : 32;
  const bool pruned = lead->params.search == GORIO_SEARCH_PRUNED;
  const bool select = pruned && K == 20;  // knn_kth_kernel + knn_collect_kernel (LDS buffer sized for K = 20); K = 32 keeps the insertion kernel
  if (pruned) {
    int rc = run_index_build(lead, todo);
    if (rc) return rc;
  }
  long total_waves = 0;
  int max_n = 0;
  for (auto& t : todo) {
    total_waves += (t.second->n + 63) / 64;
    if (t.second->n > max_n) max_n = t.second->n;
  }
  int splits = (int)((8192 + total_waves - 1) / total_waves);
  if (splits < 1) splits = 1;
  if (splits > 32) splits = 32;
  if (pruned) splits = 1;
  std::vector<KnnJob> jobs(njobs);
  int max_splits = 1;
  for (int q = 0; q < njobs; ++q) {
    gorio_apd* h = todo[q].first;
    DevCloud& c = *todo[q].second;
    int chunk = roundup((c.n_pad + splits - 1) / splits, kPad);
    if (chunk < 256) chunk = 256;
    const int s = pruned ? 1 : (c.n_pad + chunk - 1) / chunk;
    if (s > max_splits) max_splits = s;
    const size_t need = pruned ? 0 : (size_t)s * K * c.n;
    if (need > c.part_cap) {
      hipFree(c.part_d); hipFree(c.part_i);
      c.part_d = nullptr; c.part_i = nullptr;
      HIP_TRY(h, hipMalloc(&c.part_d, sizeof(float) * need));
      HIP_TRY(h, hipMalloc(&c.part_i, sizeof(int) * need));
      c.part_cap = need;
    }
    const bool keep = h->params.keep_knn_indices != 0;
    if (keep && (c.knn_k != k || !c.knn)) {
      hipFree(c.knn);
      c.knn = nullptr;
      HIP_TRY(h, hipMalloc(&c.knn, sizeof(int) * (size_t)c.cap * k));
      c.knn_k = k;
    }
    c.knn_valid = keep;
    if (select) {
      const int nw = roundup(c.n, 512) / 64;
      if (nw > c.redo_cap) {
        hipFree(c.redo); hipFree(c.kth);
        c.redo = nullptr; c.kth = nullptr;
        HIP_TRY(h, hipMalloc(&c.redo, sizeof(int) * (nw + nw / 8)));
        HIP_TRY(h, hipMalloc(&c.kth, sizeof(float) * 64 * (nw + nw / 8)));
        c.redo_cap = nw + nw / 8;
      }
    }
    KnnJob& j = jobs[q];
    j.cloud = c.view();
    j.part_d = c.part_d;
    j.part_i = c.part_i;
    j.knn_out = keep ? c.knn : nullptr;
    j.redo = select ? c.redo : nullptr;
    j.kth = select ? c.kth : nullptr;
    j.k = k;
    j.regularization = h->params.regularization;
    j.splits = s;
    j.chunk_len = chunk;
    j.qpw = 64;
    j.pad0_ = 0;
  }
  // queries per wave of the selection kernels: halved while the call has fewer waves than the chip has SIMDs (1024)
  int qpw = 64;
  {
    long w64 = 0;
    for (int q = 0; q < njobs; ++q) w64 += (jobs[q].cloud.idx.n + 63) / 64;
    while (qpw > 8 && w64 * (64 / qpw) < 1024) qpw /= 2;
    if (const char* e = std::getenv("GORIO_KNN_QPW")) {  // experiments (profiles/r03/experiments.md)
      const int v = std::atoi(e);
      if (v == 8 || v == 16 || v == 32 || v == 64) qpw = v;
    }
    for (int q = 0; q < njobs; ++q) jobs[q].qpw = qpw;
  }
  if (njobs > lead->jobs_cap) {
    hipFree(lead->d_jobs);
    lead->d_jobs = nullptr;
    HIP_TRY(lead, hipMalloc(&lead->d_jobs, sizeof(KnnJob) * njobs));
    lead->jobs_cap = njobs;
  }
  if (int rc = upload_staged(lead, lead->pin_jobs, lead->d_jobs, jobs.data(), sizeof(KnnJob) * njobs)) return rc;
  {
    StageTimer t(lead, 0);
    dim3 g1((max_n + 255) / 256, max_splits, njobs), g2((max_n + 255) / 256, 1, njobs);
    dim3 gp((roundup(max_n, 512) + 255) / 256, 1, njobs);
    if (K == 20) {
      if (pruned) {
        const dim3 gs(gp.x * (256 / qpw), 1, njobs);
        knn_kth_kernel<20><<<gs, kKnnBlock, 0, lead->stream>>>(lead->d_jobs);
        knn_collect_kernel<20><<<gs, kKnnBlock, 0, lead->stream>>>(lead->d_jobs);
        knn_pruned_kernel<20><<<gp, 256, 0, lead->stream>>>(lead->d_jobs);  // only the waves knn_collect_kernel flagged (massive ties) do anything
      } else {
        knn_partial_kernel<20><<<g1, 256, 0, lead->stream>>>(lead->d_jobs);
        cov_finalize_kernel<20><<<g2, 256, 0, lead->stream>>>(lead->d_jobs);
      }
    } else {
      if (pruned) knn_pruned_kernel<32><<<gp, 256, 0, lead->stream>>>(lead->d_jobs);
      else {
        knn_partial_kernel<32><<<g1, 256, 0, lead->stream>>>(lead->d_jobs);
        cov_finalize_kernel<32><<<g2, 256, 0, lead->stream>>>(lead->d_jobs);
      }
    }
  }
  HIP_TRY(lead, hipGetLastError());
  for (auto& t : todo) {
    t.second->cov_count = t.second->n;
    t.second->cov_k = k;
    t.second->cov_reg = lead->params.regularization;
  }
  return GORIO_OK;
}

// largest float f with (double)f < thr2: candidates beyond it can never pass the gate of APD:183
float gate_bound(double thr2) {
  if (!(thr2 < (double)FLT_MAX)) return FLT_MAX;
  float f = (float)thr2;
  while (!((double)f < thr2)) f = std::nextafterf(f, 0.0f);
  return f;
}

// mode: bit 0 = which half of nn_work this launch accumulates into, bit 1 = take the work list of nn_plan_kernel instead of the grid position
void launch_pruned(dim3 grid, hipStream_t stream, const PairDesc* d_desc, float bound, int mode) {
  nn_search_pruned_kernel<<<dim3(grid.x * (256 / kNnBlock), grid.y, grid.z), kNnBlock, 0, stream>>>(d_desc, bound, mode);
}

// launch_index: position of this search inside its align (0 = the unseeded one), or -1 for a search outside an align loop.  From the third
// search of an align on, the work measured in the second one (the first seeded one) decides which query waves are cut into parts and
// which run first (nn_plan_kernel); the first two use the grid position, with the groups of every wave dealt over `splits` workgroups.
void launch_nn(gorio_apd* lead, const PairDesc* d_desc, dim3 g_nn, int max_src_spad, int count, int max_tgt_n, int launch_index = -1) {
  if (lead->params.search == GORIO_SEARCH_PRUNED) {
    const double thr = lead->params.corr_dist_threshold;
    const long waves = (long)count * ((max_src_spad + 63) / 64);
    const bool no_plan = !lead->plan_search;  // gorio_apd_debug_set_schedule
    const bool planned = launch_index >= 2 && !lead->comm && !lead->shard_only && !no_plan;
    // entries of the work list == workgroups of a planned launch: twice the query waves for a batch that fills the chip anyway, more for
    // a few pairs (a lone 16k scan has 256 query waves: 16 parts each are 4096 workgroups), never more than 16 parts per wave
    const int nw_max = (max_src_spad + 63) / 64;
    // part budget = all work / plan_div (10240 = two parts per wave slot of the chip) and at most capmul parts per query wave on average.
    // Finer cutting pays only where the work has a heavy tail -- the far returns of a scan against a big, dense map: measured on 64 scans
    // x 1 M-point map 11.5 -> 10.3 ms per 20 searches with (4, 20480), on 64 pairs of 16 k points 2.01 -> 2.14 ms.
    const bool big = max_tgt_n >= 131072;
    const int capmul = big ? 4 : 2, plan_div = big ? 20480 : 10240;
    const int plan_cap = std::min(16 * nw_max, std::max(capmul * nw_max, 8192 / std::max(1, count)));
    if (planned) {
      launch_pruned(dim3((plan_cap + 3) / 4, 1, count), lead->stream, d_desc, gate_bound(thr * thr), 2 | (launch_index & 1));
      return;
    }
    int splits = (int)(4096 / (waves > 0 ? waves : 1));  // a lone 16k scan has 256 query waves: deal the tile groups over more workgroups
    if (splits < 1) splits = 1;
    if (splits > 16) splits = 16;
    // A big, dense target has query waves that need hundreds of tiles (a far radar return whose nearest map point is a metre away
    // sits in a ball full of map points) next to waves that need two: dealing the tile groups of every wave over several workgroups
    // shortens that tail until the plan takes over.
    if (max_tgt_n >= 131072 && splits < 8) splits = 8;
    while (splits & (splits - 1)) splits &= splits - 1;  // the kernel deals groups by their low bits: a power of two
    launch_pruned(dim3((max_src_spad + 255) / 256, splits, count), lead->stream, d_desc, gate_bound(thr * thr), launch_index >= 0 ? (launch_index & 1) : 0);
    if (launch_index == 1 && !lead->comm && !lead->shard_only && !no_plan) nn_plan_kernel<<<count, 256, 0, lead->stream>>>(d_desc, 1, count, plan_cap, plan_div);
  } else {
    nn_search_kernel<<<g_nn, 256, 0, lead->stream>>>(d_desc);
  }
}

// a target shared with other handles carries ONE set of covariances: a sharer whose k_correspondences / regularization differ from the ones
// they were estimated with would silently register with another object's covariances (the reference estimates them per object, APD:149-154)
bool shared_cov_mismatch(const gorio_apd* h) {
  const DevCloud& t = *h->tgt;
  return h->tgt.use_count() > 1 && t.cov_count == t.n && t.cov_k >= 0 && (t.cov_k != h->params.k_correspondences || t.cov_reg != h->params.regularization);
}

int check_ready(gorio_apd* h) {
  if (!h->src->present) return fail(h, GORIO_ERR_STATE, "no input source set (setInputSource)");
  if (!h->tgt->present) return fail(h, GORIO_ERR_STATE, "no input target set (setInputTarget)");
  const gorio_apd_params& p = h->params;
  if (shared_cov_mismatch(h)) return fail(h, GORIO_ERR_INVALID, "the shared target's covariances were estimated with another k_correspondences / regularization: give this handle a target of its own (setInputTarget)");
  if (p.k_correspondences < 1 || p.k_correspondences > 32) return fail(h, GORIO_ERR_UNSUPPORTED, "k_correspondences must be in [1, 32]");
  if (p.regularization < 0 || p.regularization > 4) return fail(h, GORIO_ERR_UNSUPPORTED, "unknown regularization method (the reference aborts here, APD:389-391)");
  if (h->src->cov_count != h->src->n && h->src->n < p.k_correspondences) return fail(h, GORIO_ERR_INVALID, "source cloud has fewer points than k_correspondences (undefined in the reference, APD:366-369)");
  if (h->tgt->cov_count != h->tgt->n && h->tgt->n < p.k_correspondences) return fail(h, GORIO_ERR_INVALID, "target cloud has fewer points than k_correspondences (undefined in the reference, APD:366-369)");
  return GORIO_OK;
}

void fill_desc(gorio_apd* h, PairDesc& d, PairState* state, long total_src_waves) {
  d.src = h->src->view();
  d.tgt = h->tgt->view();
  d.best_key = h->best_key;
  d.seed = h->seed;
  d.nn_work = h->nn_work;
  d.nn_plan = h->nn_plan;
  d.nn_wcap = h->nn_wcap;
  d.pad0_ = 0;
  d.corr = h->corr;
  d.sqd = h->sqd;
  d.omega6 = h->omega6;
  d.partials = h->partials;
  d.state = state;
  d.nblk = (h->src->n + 255) / 256;
  int splits = (int)((8192 + total_src_waves - 1) / total_src_waves);
  if (splits < 1) splits = 1;
  if (splits > 64) splits = 64;
  int chunk = roundup((h->tgt->n_pad + splits - 1) / splits, kPad);
  if (chunk < 512) chunk = 512;
  d.nn_chunk = chunk;
  d.nn_splits = (h->tgt->n_pad + chunk - 1) / chunk;
  d.cl_points = h->params.cl_weight_points;
  d.write_omega = 1;
  d.shard_lo = 0;
  d.shard_hi = INT_MAX;
  if ((h->comm || h->shard_only) && h->comm_world > 1) {
    // this rank's contiguous part of the query order, in units of 256 (one workgroup): sorted positions for the pruned search (a
    // spatially compact part of the scan), original indices for the exhaustive one.  Which points a rank owns changes the order of
    // the fp64 sums, never their value beyond rounding.
    const int nq = h->params.search == GORIO_SEARCH_PRUNED ? roundup(h->src->n, 512) : h->src->n;
    const long nb = (nq + 255) / 256;
    d.shard_lo = (int)(nb * h->comm_rank / h->comm_world) * 256;
    d.shard_hi = h->comm_rank == h->comm_world - 1 ? INT_MAX : (int)(nb * (h->comm_rank + 1) / h->comm_world) * 256;
  }
}

void init_state(PairState& s, const double* T16) {
  std::memset(&s, 0, sizeof(s));
  for (int i = 0; i < 16; ++i) s.x0[i] = T16[i];
  s.x0[12] = 0; s.x0[13] = 0; s.x0[14] = 0; s.x0[15] = 1;
  for (int i = 0; i < 16; ++i) s.xi[i] = s.x0[i];
  for (int i = 0; i < 12; ++i) s.Tf[i] = (float)s.x0[i];
  s.lambda = -1.0;
  for (int i = 0; i < 6; ++i) s.Hfin[i * 6 + i] = 1.0;  // final_hessian_.setIdentity(), LSQ:23
}

// batched helpers driven by the descriptor array
__global__ __launch_bounds__(256) void arm_keys_kernel(const PairDesc* __restrict__ descs) {
  const PairDesc& pd = descs[blockIdx.y];
  for (int i = blockIdx.x * 256 + threadIdx.x; i < pd.src.n; i += gridDim.x * 256) pd.best_key[i] = ~0ull;
  if (pd.nn_work)
    for (int i = blockIdx.x * 256 + threadIdx.x; i < pd.nn_wcap; i += gridDim.x * 256) {
      pd.nn_work[i] = 0u;
      pd.nn_work[pd.nn_wcap + i] = 0u;
    }
}
__global__ __launch_bounds__(64) void gather_states_kernel(const PairDesc* __restrict__ descs, PairState* __restrict__ out) {
  const unsigned int* s = reinterpret_cast<const unsigned int*>(descs[blockIdx.x].state);
  unsigned int* d = reinterpret_cast<unsigned int*>(out + blockIdx.x);
  for (int q = threadIdx.x; q < (int)(sizeof(PairState) / 4); q += 64) d[q] = s[q];
}
__global__ __launch_bounds__(64) void scatter_states_kernel(const PairDesc* __restrict__ descs, const PairState* __restrict__ in) {
  unsigned int* d = reinterpret_cast<unsigned int*>(descs[blockIdx.x].state);
  const unsigned int* s = reinterpret_cast<const unsigned int*>(in + blockIdx.x);
  for (int q = threadIdx.x; q < (int)(sizeof(PairState) / 4); q += 64) d[q] = s[q];
}

int ensure_batch(gorio_apd* lead, int count) {
  if (count > lead->desc_cap) {
    hipFree(lead->d_desc); hipFree(lead->d_states_batch);
    lead->d_desc = nullptr; lead->d_states_batch = nullptr;
    HIP_TRY(lead, hipMalloc(&lead->d_desc, sizeof(PairDesc) * count));
    HIP_TRY(lead, hipMalloc(&lead->d_states_batch, sizeof(PairState) * count));
    lead->desc_cap = count;
  }
  return GORIO_OK;
}

// Shared body of align / align_batch.
int align_impl(gorio_apd** hs, int count, const float* guesses, float* T_out, double* H_out, int* converged, int* nr_iterations, int* n_linearize) {
  if (!hs || count <= 0 || !guesses || !T_out) return GORIO_ERR_INVALID;
  gorio_apd* lead = hs[0];
  if (!lead) return GORIO_ERR_INVALID;
  HIP_TRY(lead, hipSetDevice(lead->device));
  for (int q = 0; q < count; ++q)
    if (hs[q] && hs[q]->comm && count != 1) return fail(lead, GORIO_ERR_INVALID, "a handle with a communicator (sharded source) cannot be part of a batch");
  std::unordered_set<gorio_apd*> distinct;
  for (int q = 0; q < count; ++q) {
    gorio_apd* h = hs[q];
    if (!h) return fail(lead, GORIO_ERR_INVALID, "null handle in batch");
    if (h->device != lead->device) return fail(lead, GORIO_ERR_INVALID, "all handles of a batch must live on one device");
    if (!distinct.insert(h).second) return fail(lead, GORIO_ERR_INVALID, "the same handle appears twice in a batch (its buffers would be raced on)");
    {  // one launch set advances all pairs: everything but cl_weight_points (per handle) must agree with the first handle
      gorio_apd_params a = h->params, b = lead->params;
      a.cl_weight_points = b.cl_weight_points = 0;
      if (std::memcmp(&a, &b, sizeof(a)) != 0) return fail(lead, GORIO_ERR_INVALID, "all handles of a batch must carry the same parameters (cl_weight_points excepted)");
    }
    int rc = check_ready(h);
    if (rc) {
      if (h != lead) lead->err = h->err;
      return rc;
    }
    rc = ensure_points(h, h->src->n);
    if (rc) return rc;
  }
  // APD:149-154: covariances of whichever cloud is stale
  std::vector<std::pair<gorio_apd*, DevCloud*>> todo;
  {
    std::unordered_set<DevCloud*> seen;  // a target shared by many handles of the batch is estimated once
    for (int q = 0; q < count; ++q) {
      gorio_apd* h = hs[q];
      if (h->src->cov_count != h->src->n && seen.insert(h->src.get()).second) todo.emplace_back(h, h->src.get());
      if (h->tgt->cov_count != h->tgt->n && seen.insert(h->tgt.get()).second) todo.emplace_back(h, h->tgt.get());
    }
  }
  int rc = GORIO_OK;
  if (lead->params.search == GORIO_SEARCH_PRUNED) {  // the search indices first, for ALL clouds of the batch in one call: the kd chunk size is chosen by the whole set (run_index_build)
    std::vector<std::pair<gorio_apd*, DevCloud*>> all;
    for (int q = 0; q < count; ++q) {
      all.emplace_back(hs[q], hs[q]->src.get());
      all.emplace_back(hs[q], hs[q]->tgt.get());
    }
    rc = run_index_build(lead, all);
    if (rc) return rc;
  }
  rc = run_covariances(lead, todo);
  if (rc) return rc;

  rc = ensure_batch(lead, count);
  if (rc) return rc;
  long total_src_waves = 0;
  int max_n = 0, max_m = 0;
  for (int q = 0; q < count; ++q) {
    total_src_waves += (hs[q]->src->n + 63) / 64;
    if (hs[q]->src->n > max_n) max_n = hs[q]->src->n;
    if (hs[q]->tgt->n > max_m) max_m = hs[q]->tgt->n;
  }
  std::vector<PairDesc> descs(count);
  std::vector<PairState> states(count);
  int max_splits = 1;
  for (int q = 0; q < count; ++q) {
    double T[16];
    for (int i = 0; i < 16; ++i) T[i] = (double)guesses[(size_t)q * 16 + i];  // LSQ:56
    init_state(states[q], T);
    fill_desc(hs[q], descs[q], hs[q]->d_state, total_src_waves);
    // Gauss-Newton never reads the Mahalanobis matrices back (only LM error trials and the parity hooks do): do not store them
    descs[q].write_omega = lead->params.optimizer == GORIO_OPT_LEVENBERG_MARQUARDT ? 1 : 0;
    hs[q]->omega_valid = descs[q].write_omega != 0;
    if (descs[q].nn_splits > max_splits) max_splits = descs[q].nn_splits;
    hs[q]->corr_valid = true;
  }
  if (int rc2 = upload_staged(lead, lead->pin_desc, lead->d_desc, descs.data(), sizeof(PairDesc) * count)) return rc2;
  if (int rc2 = upload_staged(lead, lead->pin_states, lead->d_states_batch, states.data(), sizeof(PairState) * count)) return rc2;
  scatter_states_kernel<<<count, 64, 0, lead->stream>>>(lead->d_desc, lead->d_states_batch);
  arm_keys_kernel<<<dim3(std::min(64, (max_n + 255) / 256), count), 256, 0, lead->stream>>>(lead->d_desc);

  const ApdConsts cst = make_consts(lead->params);
  const dim3 g_nn((max_n + 255) / 256, max_splits, count), g_lin((max_n + 255) / 256, 1, count), g_lm(count);
  if (lead->comm) {  // sharded source: the same loop, cut at the two all-reduces (count == 1, checked above)
    Rccl& R = rccl();
    const int max_it = lead->params.max_iterations;
    const bool lm = lead->params.optimizer == GORIO_OPT_LEVENBERG_MARQUARDT;
    for (int it = 0; it < max_it; ++it) {
      launch_nn(lead, lead->d_desc, g_nn, roundup(max_n, 512), 1, max_m);
      linearize_kernel<<<g_lin, 256, 0, lead->stream>>>(lead->d_desc, cst, 0);
      shard_reduce_partials_kernel<<<1, 64, 0, lead->stream>>>(lead->d_desc, lead->d_red);
      NCCL_TRY(lead, R.AllReduce(lead->d_red, lead->d_red, 28, ncclDouble, ncclSum, lead->comm, lead->stream));  // THE collective: H, b, error
      ++lead->allreduce_count;
      shard_begin_kernel<<<1, 64, 0, lead->stream>>>(lead->d_desc, lead->d_red, cst, 0);
      int trials = 0;
      bool active = lm;
      while (active) {
        // two trial slots per round trip: a Levenberg-Marquardt step is almost always decided by the first or second trial; every rank
        // reads the same flags, so every rank enqueues the same sequence of collectives
        for (int t = 0; t < 2 && trials < lead->params.lm_max_iterations; ++t, ++trials) {
          shard_trial_error_kernel<<<1, 1024, 0, lead->stream>>>(lead->d_desc, lead->d_red + 28, cst, 0);
          NCCL_TRY(lead, R.AllReduce(lead->d_red + 28, lead->d_red + 28, 1, ncclDouble, ncclSum, lead->comm, lead->stream));
          ++lead->allreduce_count;
          shard_trial_decide_kernel<<<1, 64, 0, lead->stream>>>(lead->d_desc, lead->d_red + 28, cst, 0);
        }
        HIP_TRY(lead, hipMemcpyAsync(&states[0], lead->d_state, sizeof(PairState), hipMemcpyDeviceToHost, lead->stream));
        HIP_TRY(lead, hipStreamSynchronize(lead->stream));
        active = states[0].trial_active != 0 && trials < lead->params.lm_max_iterations;
      }
      if (lm) shard_trial_decide_kernel<<<1, 64, 0, lead->stream>>>(lead->d_desc, lead->d_red + 28, cst, 1);  // end-of-iteration bookkeeping
      HIP_TRY(lead, hipGetLastError());
      HIP_TRY(lead, hipMemcpyAsync(&states[0], lead->d_state, sizeof(PairState), hipMemcpyDeviceToHost, lead->stream));
      HIP_TRY(lead, hipStreamSynchronize(lead->stream));
      if (states[0].done) break;
    }
    const PairState& s0 = states[0];
    for (int i = 0; i < 12; ++i) T_out[i] = (float)s0.x0[i];
    T_out[12] = 0.f; T_out[13] = 0.f; T_out[14] = 0.f; T_out[15] = 1.f;
    if (H_out) std::memcpy(H_out, s0.Hfin, sizeof(double) * 36);
    if (converged) converged[0] = s0.converged;
    if (nr_iterations) nr_iterations[0] = s0.nr_iterations;
    if (n_linearize) n_linearize[0] = s0.n_linearize;
    if (s0.lm_failed) lead->err = "lm not converged!!";
    resolve_stage_events(lead);
    return GORIO_OK;
  }
  int launched = 0;
  // Iterations between looks at the done flags (a look drains the stream): 4, 8, 16, 16, ... for the first batch of a handle; later
  // batches first enqueue as many iterations as the previous batch needed -- on like data that look is the only one.  A finished pair's
  // kernels return at once, so the schedule of the looks never changes a result.
  int chunk_iters = lead->align_budget > 0 ? lead->align_budget : 4;
  const bool fuse_gn = lead->params.optimizer != GORIO_OPT_LEVENBERG_MARQUARDT && lead->fuse_step;
  const int max_it = lead->params.max_iterations;
  while (launched < max_it) {
    const int todo_it = std::min(chunk_iters, max_it - launched);
    StageChain chain(lead);
    for (int it = 0; it < todo_it; ++it) {
      launch_nn(lead, lead->d_desc, g_nn, roundup(max_n, 512), count, max_m, launched + it);
      chain.mark(1);
      // Gauss-Newton needs no error trials: the optimiser step rides on the linearisation launch (its last workgroup per pair).
      // Levenberg-Marquardt keeps its own launch: an error trial wants the 1024 threads of lm_solve_kernel.
      linearize_kernel<<<g_lin, 256, 0, lead->stream>>>(lead->d_desc, cst, fuse_gn ? 1 : 0);
      chain.mark(2);
      if (!fuse_gn) {
        lm_solve_kernel<<<g_lm, 1024, 0, lead->stream>>>(lead->d_desc, cst, 0);
        chain.mark(3);
      }
    }
    launched += todo_it;
    chunk_iters = launched == todo_it && lead->align_budget > 0 ? 4 : std::min(16, chunk_iters * 2);  // after a budgeted first chunk: 4, 8, 16, ...
    HIP_TRY(lead, hipGetLastError());
    gather_states_kernel<<<count, 64, 0, lead->stream>>>(lead->d_desc, lead->d_states_batch);
    HIP_TRY(lead, hipMemcpyAsync(states.data(), lead->d_states_batch, sizeof(PairState) * count, hipMemcpyDeviceToHost, lead->stream));
    HIP_TRY(lead, hipStreamSynchronize(lead->stream));
    bool all_done = true;
    for (int q = 0; q < count; ++q) all_done = all_done && states[q].done;
    if (all_done) break;
  }
  {
    int need = 1;  // loop iterations the slowest pair used = its linearisations
    for (int q = 0; q < count; ++q) need = std::max(need, states[q].n_linearize);
    lead->align_budget = std::min(need, max_it);
  }
  for (int q = 0; q < count; ++q) {
    const PairState& s = states[q];
    for (int i = 0; i < 12; ++i) T_out[(size_t)q * 16 + i] = (float)s.x0[i];  // LSQ:78
    T_out[(size_t)q * 16 + 12] = 0.f; T_out[(size_t)q * 16 + 13] = 0.f; T_out[(size_t)q * 16 + 14] = 0.f; T_out[(size_t)q * 16 + 15] = 1.f;
    if (H_out) std::memcpy(H_out + (size_t)q * 36, s.Hfin, sizeof(double) * 36);
    if (converged) converged[q] = s.converged;
    if (nr_iterations) nr_iterations[q] = s.nr_iterations;
    if (n_linearize) n_linearize[q] = s.n_linearize;
    if (s.lm_failed) hs[q]->err = "lm not converged!!";  // LSQ:72 prints this to stderr
  }
  resolve_stage_events(lead);
  return GORIO_OK;
}

int single_desc(gorio_apd* h) {
  int rc = ensure_batch(h, 1);
  if (rc) return rc;
  PairDesc d;
  fill_desc(h, d, h->d_state, (h->src->n + 63) / 64);
  HIP_TRY(h, hipMemcpyAsync(h->d_desc, &d, sizeof(PairDesc), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return GORIO_OK;
}

}  // namespace

// =============================================================================================== C ABI

extern "C" {

void gorio_apd_default_params(gorio_apd_params* p) {
  if (!p) return;
  std::memset(p, 0, sizeof(*p));
  p->k_correspondences = 20;
  p->regularization = GORIO_REG_PLANE;
  p->dist_var = 0.86;
  p->azimuth_var = 0.5;
  p->elevation_var = 1.0;
  p->corr_dist_threshold = (double)FLT_MAX;
  p->max_iterations = 64;
  p->rotation_epsilon = 2e-3;
  p->transformation_epsilon = 5e-4;
  p->optimizer = GORIO_OPT_LEVENBERG_MARQUARDT;
  p->lm_max_iterations = 10;
  p->lm_init_lambda_factor = 1e-9;
  p->search = GORIO_SEARCH_BRUTE_FORCE;
  p->cl_weight_points = 0;
  p->keep_knn_indices = 0;
}

int gorio_apd_create(gorio_apd_t** out, int device) {
  if (!out) return GORIO_ERR_INVALID;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return GORIO_ERR_NO_DEVICE;
  if (device < 0 || device >= ndev) return GORIO_ERR_INVALID;
  if (hipSetDevice(device) != hipSuccess) return GORIO_ERR_NO_DEVICE;
  gorio_apd* h = new (std::nothrow) gorio_apd();
  if (!h) return GORIO_ERR_ALLOC;
  h->device = device;
  h->src = std::make_shared<DevCloud>();
  h->tgt = std::make_shared<DevCloud>();
  h->src->device = h->tgt->device = device;
  gorio_apd_default_params(&h->params);
  h->stream = device_stream(device);
  {  // the fence-free "last workgroup runs the optimiser step" hand-over relies on how gfx950 writes through and acknowledges sc1 stores across
     // its XCD L2s: it was validated there and is switched off on anything else (the step then takes its own launch)
    hipDeviceProp_t prop;
    h->fuse_step = hipGetDeviceProperties(&prop, device) == hipSuccess && std::strncmp(prop.gcnArchName, "gfx950", 6) == 0;
  }
  if (!h->stream || hipMalloc(&h->d_state, sizeof(PairState)) != hipSuccess ||
      hipMalloc(&h->d_fit, sizeof(double) * 4) != hipSuccess || hipMalloc(&h->d_red, sizeof(double) * 32) != hipSuccess) {
    delete h;
    return GORIO_ERR_NO_DEVICE;
  }
  *out = h;
  return GORIO_OK;
}

void gorio_apd_destroy(gorio_apd_t* h) {
  if (!h) return;
  hipSetDevice(h->device);
  if (h->stream) hipStreamSynchronize(h->stream);
  if (h->comm && rccl().ok) rccl().CommDestroy(h->comm);
  hipFree(h->d_red);
  hipFree(h->d_sub_in); hipFree(h->d_sub_out); hipFree(h->d_sub_vox); hipFree(h->d_sub_keys); hipFree(h->d_sub_counts); hipFree(h->d_sub_frames); hipFree(h->d_sub_bb); hipFree(h->d_sub_job);
  h->src.reset();
  h->tgt.reset();
  hipFree(h->best_key); hipFree(h->seed); hipFree(h->nn_work); hipFree(h->nn_plan); hipFree(h->corr); hipFree(h->sqd); hipFree(h->omega6); hipFree(h->partials);
  hipFree(h->d_state); hipFree(h->d_desc); hipFree(h->d_states_batch); hipFree(h->d_jobs); hipFree(h->d_ijobs); hipFree(h->d_fit); hipFree(h->d_copy_jobs);
  for (auto& e : h->ev_pool) { hipEventDestroy(e.start); hipEventDestroy(e.stop); }
  for (gorio_apd::Pinned* b : {&h->pin_ijobs, &h->pin_jobs, &h->pin_desc, &h->pin_states, &h->pin_copy}) {
    if (b->p) hipHostFree(b->p);
    if (b->ev) hipEventDestroy(b->ev);
  }
  delete h;
}

const char* gorio_apd_last_error(const gorio_apd_t* h) { return h ? h->err.c_str() : "null handle"; }

int gorio_apd_set_params(gorio_apd_t* h, const gorio_apd_params* p) {
  if (!h || !p) return GORIO_ERR_INVALID;
  if (p->k_correspondences < 1 || p->k_correspondences > 32) return fail(h, GORIO_ERR_UNSUPPORTED, "k_correspondences must be in [1, 32]");
  if (p->regularization < 0 || p->regularization > 4) return fail(h, GORIO_ERR_UNSUPPORTED, "unknown regularization method");
  if (p->optimizer != GORIO_OPT_GAUSS_NEWTON && p->optimizer != GORIO_OPT_LEVENBERG_MARQUARDT) return fail(h, GORIO_ERR_INVALID, "unknown optimizer");
  std::memset(&h->params, 0, sizeof(h->params));  // padding bytes stay zero so whole-struct comparisons (batch validation) are meaningful
  h->params.k_correspondences = p->k_correspondences; h->params.regularization = p->regularization;
  h->params.dist_var = p->dist_var; h->params.azimuth_var = p->azimuth_var; h->params.elevation_var = p->elevation_var;
  h->params.corr_dist_threshold = p->corr_dist_threshold; h->params.max_iterations = p->max_iterations;
  h->params.rotation_epsilon = p->rotation_epsilon; h->params.transformation_epsilon = p->transformation_epsilon;
  h->params.optimizer = p->optimizer; h->params.lm_max_iterations = p->lm_max_iterations; h->params.lm_init_lambda_factor = p->lm_init_lambda_factor;
  h->params.search = p->search; h->params.cl_weight_points = p->cl_weight_points; h->params.keep_knn_indices = p->keep_knn_indices;
  return GORIO_OK;
}

int gorio_apd_get_params(const gorio_apd_t* h, gorio_apd_params* p) {
  if (!h || !p) return GORIO_ERR_INVALID;
  *p = h->params;
  return GORIO_OK;
}

int gorio_apd_set_source(gorio_apd_t* h, const float* xyz, const float* label, int n, int stride) {
  if (!h) return GORIO_ERR_INVALID;
  h->corr_valid = false;
  make_private(h, h->src);
  return upload_cloud(h, *h->src, xyz, label, n, stride);
}
int gorio_apd_set_target(gorio_apd_t* h, const float* xyz, const float* label, int n, int stride) {
  if (!h) return GORIO_ERR_INVALID;
  h->corr_valid = false;
  make_private(h, h->tgt);
  return upload_cloud(h, *h->tgt, xyz, label, n, stride);
}
int gorio_apd_set_source_device(gorio_apd_t* h, const float* dx, const float* dy, const float* dz, const float* dl, int n) {
  if (!h) return GORIO_ERR_INVALID;
  h->corr_valid = false;
  make_private(h, h->src);
  return upload_cloud_device(h, *h->src, dx, dy, dz, dl, n);
}
int gorio_apd_set_target_device(gorio_apd_t* h, const float* dx, const float* dy, const float* dz, const float* dl, int n) {
  if (!h) return GORIO_ERR_INVALID;
  h->corr_valid = false;
  make_private(h, h->tgt);
  return upload_cloud_device(h, *h->tgt, dx, dy, dz, dl, n);
}

int gorio_apd_set_clouds_device_batch(gorio_apd_t** handles, int count, const gorio_apd_device_cloud* source, const gorio_apd_device_cloud* target) {
  if (!handles || count <= 0 || (!source && !target)) return GORIO_ERR_INVALID;
  gorio_apd* lead = handles[0];
  if (!lead) return GORIO_ERR_INVALID;
  HIP_TRY(lead, hipSetDevice(lead->device));
  std::vector<CopyJob> jobs;
  jobs.reserve(2 * (size_t)count);
  int max_pad = 0;
  for (int q = 0; q < count; ++q) {
    gorio_apd* h = handles[q];
    if (!h) return fail(lead, GORIO_ERR_INVALID, "null handle in batch");
    if (h->device != lead->device) return fail(lead, GORIO_ERR_INVALID, "all handles of a batch must live on one device");
    for (int side = 0; side < 2; ++side) {
      const gorio_apd_device_cloud* arr = side == 0 ? source : target;
      if (!arr) continue;
      const gorio_apd_device_cloud& in = arr[q];
      if (!in.x || !in.y || !in.z || in.n <= 0) return fail(lead, GORIO_ERR_INVALID, "set_clouds_device_batch: bad cloud arguments");
      make_private(h, side == 0 ? h->src : h->tgt);
      DevCloud& c = side == 0 ? *h->src : *h->tgt;
      int rc = ensure_cloud(h, c, in.n);
      if (rc) {
        if (h != lead) lead->err = h->err;
        return rc;
      }
      jobs.push_back(CopyJob{in.x, in.y, in.z, in.label, c.x, c.y, c.z, c.label, c.p4, in.n, c.n_pad});
      max_pad = std::max(max_pad, c.n_pad);
      c.present = true;
      c.cov_count = 0;
      c.idx_valid = false;
      h->corr_valid = false;
    }
  }
  const size_t bytes = sizeof(CopyJob) * jobs.size();
  if (bytes > lead->copy_jobs_cap) {
    hipFree(lead->d_copy_jobs);
    lead->d_copy_jobs = nullptr;
    lead->copy_jobs_cap = 0;
    HIP_TRY(lead, hipMalloc(&lead->d_copy_jobs, bytes));
    lead->copy_jobs_cap = bytes;
  }
  if (int rc = upload_staged(lead, lead->pin_copy, lead->d_copy_jobs, jobs.data(), bytes)) return rc;
  copy_clouds_kernel<<<dim3(std::min(16, (max_pad + 255) / 256), (unsigned)jobs.size()), 256, 0, lead->stream>>>(static_cast<const CopyJob*>(lead->d_copy_jobs));
  HIP_TRY(lead, hipGetLastError());
  return GORIO_OK;
}

int gorio_apd_clear_source(gorio_apd_t* h) {
  if (!h) return GORIO_ERR_INVALID;
  make_private(h, h->src);
  h->src->present = false; h->src->n = 0; h->src->cov_count = 0; h->corr_valid = false;  // APD:101-105
  return GORIO_OK;
}
int gorio_apd_clear_target(gorio_apd_t* h) {
  if (!h) return GORIO_ERR_INVALID;
  make_private(h, h->tgt);
  h->tgt->present = false; h->tgt->n = 0; h->tgt->cov_count = 0; h->corr_valid = false;  // APD:107-112
  return GORIO_OK;
}
int gorio_apd_set_target_shared(gorio_apd_t* h, gorio_apd_t* owner) {
  if (!h || !owner) return GORIO_ERR_INVALID;
  if (h->device != owner->device) return fail(h, GORIO_ERR_INVALID, "set_target_shared: both handles must live on one device");
  if (!owner->tgt->present) return fail(h, GORIO_ERR_STATE, "set_target_shared: the owner has no input target");
  {
    const DevCloud& t = *owner->tgt;
    if (t.cov_count == t.n && t.cov_k >= 0 && (t.cov_k != h->params.k_correspondences || t.cov_reg != h->params.regularization))
      return fail(h, GORIO_ERR_INVALID, "set_target_shared: the owner's covariances were estimated with another k_correspondences / regularization than this handle's");
  }
  h->tgt = owner->tgt;  // points, covariances, search index: one copy on the device, alive until the last handle lets go of it
  h->corr_valid = false;
  return GORIO_OK;
}

int gorio_apd_set_target_submap(gorio_apd_t* h, const gorio_apd_keyframe* frames, int count, double voxel_leaf, int* n_target) {
  if (!h || !frames || count <= 0) return GORIO_ERR_INVALID;
  HIP_TRY(h, hipSetDevice(h->device));
  // ---- host staging: finite points of all frames, packed (x, y, z, label); pcl::PassThrough and pcl::VoxelGrid both skip non-finite points
  std::vector<float4> stage;
  std::vector<SubmapFrame> fr(count);
  int max_frame = 0;
  for (int k = 0; k < count; ++k) {
    const gorio_apd_keyframe& f = frames[k];
    if (f.n < 0 || (f.n > 0 && !f.xyz) || f.point_stride_bytes < 12 || (f.point_stride_bytes % 4) != 0 || !f.rel_pose) return fail(h, GORIO_ERR_INVALID, "set_target_submap: bad keyframe arguments");
    fr[k].begin = (int)stage.size();
    const int st = f.point_stride_bytes / 4;
    for (int i = 0; i < f.n; ++i) {
      const float* p = f.xyz + (size_t)i * st;
      if (!std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2])) continue;
      stage.push_back(make_float4(p[0], p[1], p[2], f.label ? f.label[(size_t)i * st] : 0.0f));
    }
    fr[k].end = (int)stage.size();
    for (int q = 0; q < 12; ++q) fr[k].T[q] = f.rel_pose[q];
    max_frame = std::max(max_frame, fr[k].end - fr[k].begin);
  }
  const int m = (int)stage.size();
  if (m <= 0) return fail(h, GORIO_ERR_INVALID, "set_target_submap: no finite point in any keyframe");
  if ((size_t)m > h->sub_cap) {
    hipFree(h->d_sub_in); hipFree(h->d_sub_out); hipFree(h->d_sub_vox);
    h->d_sub_in = h->d_sub_out = h->d_sub_vox = nullptr;
    h->sub_cap = 0;
    const size_t cap = (size_t)m + (size_t)m / 8;
    HIP_TRY(h, hipMalloc(&h->d_sub_in, sizeof(float4) * cap));
    HIP_TRY(h, hipMalloc(&h->d_sub_out, sizeof(float4) * cap));
    HIP_TRY(h, hipMalloc(&h->d_sub_vox, sizeof(float4) * cap));
    h->sub_cap = cap;
  }
  if (count > h->sub_frames_cap) {
    hipFree(h->d_sub_frames);
    h->d_sub_frames = nullptr;
    HIP_TRY(h, hipMalloc(&h->d_sub_frames, sizeof(SubmapFrame) * count));
    h->sub_frames_cap = count;
  }
  if (!h->d_sub_bb) HIP_TRY(h, hipMalloc(&h->d_sub_bb, sizeof(unsigned int) * 8));
  if (!h->d_sub_job) HIP_TRY(h, hipMalloc(&h->d_sub_job, sizeof(IndexJob)));
  HIP_TRY(h, hipMemcpyAsync(h->d_sub_in, stage.data(), sizeof(float4) * m, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipMemcpyAsync(h->d_sub_frames, fr.data(), sizeof(SubmapFrame) * count, hipMemcpyHostToDevice, h->stream));
  submap_transform_kernel<<<dim3((max_frame + 255) / 256, count), 256, 0, h->stream>>>(h->d_sub_in, h->d_sub_frames, h->d_sub_out);
  HIP_TRY(h, hipGetLastError());
  const float4* result = h->d_sub_out;
  int n_out = m;
  if (voxel_leaf > 0.0) {  // pcl::VoxelGrid (SMO:145-149)
    unsigned int bb[6];
    vox_bbox_init_kernel<<<1, 64, 0, h->stream>>>(h->d_sub_bb);
    vox_bbox_kernel<<<std::min(256, (m + 255) / 256), 256, 0, h->stream>>>(h->d_sub_out, m, h->d_sub_bb);
    HIP_TRY(h, hipMemcpyAsync(bb, h->d_sub_bb, sizeof(bb), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));  // also covers the pageable staging vectors above
    auto ord2f_host = [](unsigned int o) {
      const unsigned int u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
      float f;
      std::memcpy(&f, &u, 4);
      return f;
    };
    float mn[3], mx[3];
    for (int a = 0; a < 3; ++a) { mn[a] = ord2f_host(bb[a]); mx[a] = ord2f_host(bb[3 + a]); }
    VoxGrid g;
    g.inv = 1.0f / (float)voxel_leaf;
    const long long dx = (long long)((mx[0] - mn[0]) * g.inv) + 1, dy = (long long)((mx[1] - mn[1]) * g.inv) + 1, dz = (long long)((mx[2] - mn[2]) * g.inv) + 1;
    if (dx * dy * dz <= (long long)INT_MAX) {  // otherwise: "Leaf size is too small for the input dataset" -> PCL returns the input unchanged
      int div_b[3];
      for (int a = 0; a < 3; ++a) {
        g.min_b[a] = (int)std::floor(mn[a] * g.inv);
        div_b[a] = (int)std::floor(mx[a] * g.inv) - g.min_b[a] + 1;
      }
      g.div0 = div_b[0];
      g.div01 = div_b[0] * div_b[1];
      int npow2 = kSortTile;
      while (npow2 < m) npow2 <<= 1;
      if ((size_t)npow2 > h->sub_keys_cap) {
        hipFree(h->d_sub_keys);
        h->d_sub_keys = nullptr;
        h->sub_keys_cap = 0;
        HIP_TRY(h, hipMalloc(&h->d_sub_keys, sizeof(unsigned long long) * npow2));
        h->sub_keys_cap = npow2;
      }
      const int nblocks = (m + 255) / 256;
      if ((size_t)nblocks + 1 > h->sub_counts_cap) {
        hipFree(h->d_sub_counts);
        h->d_sub_counts = nullptr;
        h->sub_counts_cap = 0;
        HIP_TRY(h, hipMalloc(&h->d_sub_counts, sizeof(int) * (nblocks + 1 + nblocks / 8)));
        h->sub_counts_cap = nblocks + 1 + nblocks / 8;
      }
      IndexJob job;
      std::memset(&job, 0, sizeof(job));
      job.n = m;
      job.npow2 = npow2;
      job.keys = h->d_sub_keys;
      HIP_TRY(h, hipMemcpyAsync(h->d_sub_job, &job, sizeof(job), hipMemcpyHostToDevice, h->stream));
      HIP_TRY(h, hipStreamSynchronize(h->stream));  // `job` is a stack object
      vox_key_kernel<<<(npow2 + 255) / 256, 256, 0, h->stream>>>(h->d_sub_out, m, npow2, g, h->d_sub_keys);
      const IndexJob* dj = h->d_sub_job;
      bitonic_tile_sort_kernel<<<dim3(npow2 / kSortTile, 1), 1024, 0, h->stream>>>(dj);
      for (int k = 2 * kSortTile; k <= npow2; k <<= 1) {
        for (int j = k >> 1; j >= kSortTile; j >>= 1) bitonic_global_kernel<<<dim3((npow2 / 2 + 255) / 256, 1), 256, 0, h->stream>>>(dj, k, j);
        bitonic_tile_merge_kernel<<<dim3(npow2 / kSortTile, 1), 1024, 0, h->stream>>>(dj, k);
      }
      vox_count_kernel<<<nblocks, 256, 0, h->stream>>>(h->d_sub_keys, m, h->d_sub_counts);
      vox_scan_kernel<<<1, 1024, 0, h->stream>>>(h->d_sub_counts, nblocks);
      vox_centroid_kernel<<<nblocks, 256, 0, h->stream>>>(h->d_sub_keys, h->d_sub_out, m, h->d_sub_counts, h->d_sub_vox);
      HIP_TRY(h, hipGetLastError());
      HIP_TRY(h, hipMemcpyAsync(&n_out, h->d_sub_counts + nblocks, sizeof(int), hipMemcpyDeviceToHost, h->stream));
      HIP_TRY(h, hipStreamSynchronize(h->stream));
      result = h->d_sub_vox;
    }
  }
  make_private(h, h->tgt);
  DevCloud& c = *h->tgt;
  int rc = ensure_cloud(h, c, n_out);
  if (rc) return rc;
  submap_store_kernel<<<(c.n_pad + 255) / 256, 256, 0, h->stream>>>(result, n_out, c.n_pad, c.x, c.y, c.z, c.label, c.p4);
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipStreamSynchronize(h->stream));  // the staging vectors die with this call
  c.present = true;
  c.cov_count = 0;
  c.idx_valid = false;
  c.knn_valid = false;
  h->corr_valid = false;
  if (n_target) *n_target = n_out;
  return GORIO_OK;
}

int gorio_apd_get_target_points(gorio_apd_t* h, float* xyz_out, float* label_out, int n, int point_stride_bytes) {
  if (!h || !xyz_out) return GORIO_ERR_INVALID;
  if (!h->tgt->present || n != h->tgt->n) return fail(h, GORIO_ERR_STATE, "get_target_points: no matching target cloud");
  if (point_stride_bytes < 12 || point_stride_bytes % 4) return fail(h, GORIO_ERR_INVALID, "get_target_points: bad stride");
  HIP_TRY(h, hipSetDevice(h->device));
  std::vector<float4> tmp((size_t)n);
  HIP_TRY(h, hipMemcpyAsync(tmp.data(), h->tgt->p4, sizeof(float4) * (size_t)n, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  const int st = point_stride_bytes / 4;
  for (int i = 0; i < n; ++i) {
    float* o = xyz_out + (size_t)i * st;
    o[0] = tmp[i].x; o[1] = tmp[i].y; o[2] = tmp[i].z;
    if (label_out) label_out[(size_t)i * st] = tmp[i].w;
  }
  return GORIO_OK;
}

int gorio_apd_swap_source_and_target(gorio_apd_t* h) {
  if (!h) return GORIO_ERR_INVALID;
  std::swap(h->src, h->tgt);  // APD:90-92: clouds, trees and covariances change sides
  h->corr_valid = false;      // APD:96-97
  return GORIO_OK;
}

static int set_covs(gorio_apd* h, DevCloud& c, const double* cov, int n) {
  if (n < 0 || (n > 0 && !cov)) return fail(h, GORIO_ERR_INVALID, "set_covariances: bad arguments");
  if (!c.present || n != c.n) {
    // the reference stores the vector whatever its size (APD:138-145) and recomputes the covariances in computeTransformation when the
    // size does not match the cloud (APD:149-154): a mismatching set is therefore the same as none
    c.cov_count = 0;
    c.knn_valid = false;
    return GORIO_OK;
  }
  HIP_TRY(h, hipSetDevice(h->device));
  std::vector<double> c6((size_t)n * 6);
  for (int i = 0; i < n; ++i) {
    const double* m = cov + (size_t)i * 16;
    double* o = c6.data() + (size_t)i * 6;
    o[0] = m[0]; o[1] = m[1]; o[2] = m[2]; o[3] = m[5]; o[4] = m[6]; o[5] = m[10];
  }
  HIP_TRY(h, hipMemcpyAsync(c.cov6, c6.data(), sizeof(double) * 6 * n, hipMemcpyHostToDevice, h->stream));
  geo_weight_kernel<<<(n + 255) / 256, 256, 0, h->stream>>>(c.cov6, c.geo_w, n);
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  c.cov_count = n;
  c.cov_k = c.cov_reg = -1;
  c.knn_valid = false;  // these covariances did not come from a k-NN search of this library
  return GORIO_OK;
}

int gorio_apd_set_source_covariances(gorio_apd_t* h, const double* cov, int n) { return h ? set_covs(h, *h->src, cov, n) : GORIO_ERR_INVALID; }
int gorio_apd_set_target_covariances(gorio_apd_t* h, const double* cov, int n) { return h ? set_covs(h, *h->tgt, cov, n) : GORIO_ERR_INVALID; }

static int get_covs(gorio_apd* h, DevCloud& c, double* cov, int n) {
  const int cnt = c.present ? c.cov_count : 0;
  if (!cov || cnt == 0) return cnt;
  const int m = n < cnt ? n : cnt;
  if (hipSetDevice(h->device) != hipSuccess) return fail(h, GORIO_ERR_NO_DEVICE, "hipSetDevice failed");
  std::vector<double> c6((size_t)m * 6);
  if (hipMemcpyAsync(c6.data(), c.cov6, sizeof(double) * 6 * m, hipMemcpyDeviceToHost, h->stream) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess)
    return fail(h, GORIO_ERR_NO_DEVICE, "covariance download failed");
  for (int i = 0; i < m; ++i) {
    const double* s = c6.data() + (size_t)i * 6;
    double* o = cov + (size_t)i * 16;
    o[0] = s[0]; o[1] = s[1]; o[2] = s[2]; o[3] = 0;
    o[4] = s[1]; o[5] = s[3]; o[6] = s[4]; o[7] = 0;
    o[8] = s[2]; o[9] = s[4]; o[10] = s[5]; o[11] = 0;
    o[12] = 0; o[13] = 0; o[14] = 0; o[15] = 0;
  }
  return cnt;
}
int gorio_apd_get_source_covariances(gorio_apd_t* h, double* cov, int n) { return h ? get_covs(h, *h->src, cov, n) : GORIO_ERR_INVALID; }
int gorio_apd_get_target_covariances(gorio_apd_t* h, double* cov, int n) { return h ? get_covs(h, *h->tgt, cov, n) : GORIO_ERR_INVALID; }

int gorio_apd_calculate_covariances(gorio_apd_t* h) {
  if (!h) return GORIO_ERR_INVALID;
  HIP_TRY(h, hipSetDevice(h->device));
  const gorio_apd_params& p = h->params;
  std::vector<std::pair<gorio_apd*, DevCloud*>> todo;
  for (DevCloud* c : {h->src.get(), h->tgt.get()}) {
    if (c->present && c->cov_count != c->n) {
      if (c->n < p.k_correspondences) return fail(h, GORIO_ERR_INVALID, "cloud has fewer points than k_correspondences (undefined in the reference, APD:366-369)");
      todo.emplace_back(h, c);
    }
  }
  int rc = GORIO_OK;
  if (p.search == GORIO_SEARCH_PRUNED && !todo.empty()) {  // indices of both clouds in one call (kd chunk size by the pair, run_index_build)
    std::vector<std::pair<gorio_apd*, DevCloud*>> both;
    for (DevCloud* c : {h->src.get(), h->tgt.get()})
      if (c->present) both.emplace_back(h, c);
    rc = run_index_build(h, both);
    if (rc) return rc;
  }
  rc = run_covariances(h, todo);
  if (rc) return rc;
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  resolve_stage_events(h);
  return GORIO_OK;
}

int gorio_apd_get_knn_indices(gorio_apd_t* h, int which, int* idx, int n_times_k) {
  if (!h || !idx) return GORIO_ERR_INVALID;
  DevCloud& c = which == 0 ? *h->src : *h->tgt;
  if (!c.present || !c.knn || !c.knn_valid || c.cov_count != c.n) return fail(h, GORIO_ERR_STATE, "no k-NN result held for this cloud (set params.keep_knn_indices before the covariances are computed)");
  if (n_times_k != c.n * c.knn_k) return fail(h, GORIO_ERR_INVALID, "get_knn_indices: size mismatch");
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, hipMemcpyAsync(idx, c.knn, sizeof(int) * (size_t)n_times_k, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return GORIO_OK;
}

int gorio_apd_align(gorio_apd_t* h, const float guess[16], float T_out[16], double* H_out, int* converged, int* nr_iterations, int* n_linearize) {
  if (!h) return GORIO_ERR_INVALID;
  gorio_apd* hs[1] = {h};
  return align_impl(hs, 1, guess, T_out, H_out, converged, nr_iterations, n_linearize);
}

int gorio_apd_align_batch(gorio_apd_t** handles, int count, const float* guesses, float* T_out, double* H_out, int* converged, int* nr_iterations, int* n_linearize) {
  return align_impl(handles, count, guesses, T_out, H_out, converged, nr_iterations, n_linearize);
}

int gorio_apd_linearize(gorio_apd_t* h, const double T[16], double* H, double* b, double* error) {
  if (!h || !T) return GORIO_ERR_INVALID;
  HIP_TRY(h, hipSetDevice(h->device));
  int rc = check_ready(h);
  if (rc) return rc;
  rc = ensure_points(h, h->src->n);
  if (rc) return rc;
  rc = gorio_apd_calculate_covariances(h);
  if (rc) return rc;
  if (h->params.search == GORIO_SEARCH_PRUNED) {
    std::vector<std::pair<gorio_apd*, DevCloud*>> all = {{h, h->src.get()}, {h, h->tgt.get()}};
    rc = run_index_build(h, all);
    if (rc) return rc;
  }
  rc = single_desc(h);
  if (rc) return rc;
  PairState s;
  init_state(s, T);
  HIP_TRY(h, hipMemcpyAsync(h->d_state, &s, sizeof(s), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipMemsetAsync(h->best_key, 0xff, sizeof(unsigned long long) * h->src->n, h->stream));
  const ApdConsts cst = make_consts(h->params);
  PairDesc d;
  fill_desc(h, d, h->d_state, (h->src->n + 63) / 64);
  const int nbx = (h->src->n + 255) / 256;
  launch_nn(h, h->d_desc, dim3(nbx, d.nn_splits, 1), roundup(h->src->n, 512), 1, h->tgt->n);
  linearize_kernel<<<dim3(nbx, 1, 1), 256, 0, h->stream>>>(h->d_desc, cst, 0);
  if (h->comm) {  // every rank of the communicator makes this call; H, b and the error come back summed over all of them
    shard_reduce_partials_kernel<<<1, 64, 0, h->stream>>>(h->d_desc, h->d_red);
    NCCL_TRY(h, rccl().AllReduce(h->d_red, h->d_red, 28, ncclDouble, ncclSum, h->comm, h->stream));
    shard_begin_kernel<<<1, 64, 0, h->stream>>>(h->d_desc, h->d_red, cst, 1);
  } else {
    lm_solve_kernel<<<1, 1024, 0, h->stream>>>(h->d_desc, cst, 1);
  }
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipMemcpyAsync(&s, h->d_state, sizeof(s), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  h->corr_valid = true;
  h->omega_valid = true;
  if (H && b) {
    std::memcpy(H, s.H, sizeof(double) * 36);
    std::memcpy(b, s.b, sizeof(double) * 6);
  }
  if (error) *error = s.y0;
  return GORIO_OK;
}

int gorio_apd_compute_error(gorio_apd_t* h, const double T[16], double* error) {
  if (!h || !T || !error) return GORIO_ERR_INVALID;
  HIP_TRY(h, hipSetDevice(h->device));
  if (!h->corr_valid || !h->omega_valid) return fail(h, GORIO_ERR_STATE, "compute_error needs the correspondences and Mahalanobis matrices of a previous linearize (or LM align)");
  int rc = single_desc(h);
  if (rc) return rc;
  double xi[16];
  for (int i = 0; i < 16; ++i) xi[i] = T[i];
  HIP_TRY(h, hipMemcpyAsync(reinterpret_cast<char*>(h->d_state) + offsetof(PairState, xi), xi, sizeof(xi), hipMemcpyHostToDevice, h->stream));
  const ApdConsts cst = make_consts(h->params);
  double yi = 0.0;
  if (h->comm) {
    shard_trial_error_kernel<<<1, 1024, 0, h->stream>>>(h->d_desc, h->d_red + 28, cst, 2);
    NCCL_TRY(h, rccl().AllReduce(h->d_red + 28, h->d_red + 28, 1, ncclDouble, ncclSum, h->comm, h->stream));
    HIP_TRY(h, hipMemcpyAsync(&yi, h->d_red + 28, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    *error = yi;
    return GORIO_OK;
  }
  lm_solve_kernel<<<1, 1024, 0, h->stream>>>(h->d_desc, cst, 2);
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipMemcpyAsync(&yi, reinterpret_cast<char*>(h->d_state) + offsetof(PairState, yi), sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  *error = yi;
  return GORIO_OK;
}

int gorio_apd_get_correspondences(gorio_apd_t* h, int* corr, float* sq_dist, int n) {
  if (!h) return GORIO_ERR_INVALID;
  if (!h->corr_valid) return fail(h, GORIO_ERR_STATE, "no correspondences held");
  if (n != h->src->n) return fail(h, GORIO_ERR_INVALID, "get_correspondences: size mismatch");
  HIP_TRY(h, hipSetDevice(h->device));
  if (corr) HIP_TRY(h, hipMemcpyAsync(corr, h->corr, sizeof(int) * n, hipMemcpyDeviceToHost, h->stream));
  if (sq_dist) HIP_TRY(h, hipMemcpyAsync(sq_dist, h->sqd, sizeof(float) * n, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return GORIO_OK;
}

int gorio_apd_get_mahalanobis(gorio_apd_t* h, double* maha, int n) {
  if (!h || !maha) return GORIO_ERR_INVALID;
  if (!h->corr_valid || !h->omega_valid) return fail(h, GORIO_ERR_STATE, "no Mahalanobis matrices held (a Gauss-Newton align does not materialise them: call gorio_apd_linearize)");
  if (n != h->src->n) return fail(h, GORIO_ERR_INVALID, "get_mahalanobis: size mismatch");
  HIP_TRY(h, hipSetDevice(h->device));
  std::vector<double> o6((size_t)n * 6);
  HIP_TRY(h, hipMemcpyAsync(o6.data(), h->omega6, sizeof(double) * 6 * n, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  for (int i = 0; i < n; ++i) {
    const double* s = o6.data() + (size_t)i * 6;
    double* o = maha + (size_t)i * 16;
    o[0] = s[0]; o[1] = s[1]; o[2] = s[2]; o[3] = 0;
    o[4] = s[1]; o[5] = s[3]; o[6] = s[4]; o[7] = 0;
    o[8] = s[2]; o[9] = s[4]; o[10] = s[5]; o[11] = 0;
    o[12] = 0; o[13] = 0; o[14] = 0; o[15] = 0;
  }
  return GORIO_OK;
}

int gorio_apd_transform_source(gorio_apd_t* h, const float T[16], float* xyz_out, int n, int stride) {
  if (!h || !T || !xyz_out) return GORIO_ERR_INVALID;
  if (!h->src->present || n != h->src->n) return fail(h, GORIO_ERR_STATE, "transform_source: no matching source cloud");
  if (stride < 12 || stride % 4) return fail(h, GORIO_ERR_INVALID, "transform_source: bad stride");
  HIP_TRY(h, hipSetDevice(h->device));
  float* d_out = nullptr;
  HIP_TRY(h, hipMalloc(&d_out, sizeof(float) * 3 * (size_t)n));
  TfArg tf;
  for (int i = 0; i < 12; ++i) tf.m[i] = T[i];
  transform_cloud_kernel<<<(n + 255) / 256, 256, 0, h->stream>>>(h->src->x, h->src->y, h->src->z, n, tf, d_out);
  std::vector<float> tmp((size_t)n * 3);
  hipError_t e = hipMemcpyAsync(tmp.data(), d_out, sizeof(float) * 3 * (size_t)n, hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  hipFree(d_out);
  if (e != hipSuccess) return fail(h, GORIO_ERR_NO_DEVICE, hipGetErrorString(e));
  const int st = stride / 4;
  for (int i = 0; i < n; ++i) {
    float* o = xyz_out + (size_t)i * st;
    o[0] = tmp[3 * (size_t)i]; o[1] = tmp[3 * (size_t)i + 1]; o[2] = tmp[3 * (size_t)i + 2];
  }
  return GORIO_OK;
}

int gorio_apd_fitness_score(gorio_apd_t* h, const float T[16], double max_range, double inlier_dist, double* score, double* inlier_fraction) {
  if (!h || !T || !score) return GORIO_ERR_INVALID;
  if (!h->src->present || !h->tgt->present) return fail(h, GORIO_ERR_STATE, "fitness_score: clouds not set");
  if ((h->comm || h->shard_only) && h->comm_world > 1)
    return fail(h, GORIO_ERR_STATE, "fitness_score: this handle searches only its rank's share of the source (gorio_apd_comm_init); score the pose on an unsharded handle");
  HIP_TRY(h, hipSetDevice(h->device));
  int rc = ensure_points(h, h->src->n);
  if (rc) return rc;
  const bool pruned = h->params.search == GORIO_SEARCH_PRUNED;
  if (pruned) {  // the same exact branch-and-bound search as the registration (matters for 100 k-point maps)
    std::vector<std::pair<gorio_apd*, DevCloud*>> both = {{h, h->src.get()}, {h, h->tgt.get()}};
    rc = run_index_build(h, both);
    if (rc) return rc;
  }
  rc = single_desc(h);
  if (rc) return rc;
  PairState s;
  double Td[16];
  for (int i = 0; i < 16; ++i) Td[i] = (double)T[i];
  init_state(s, Td);
  for (int i = 0; i < 12; ++i) s.Tf[i] = T[i];
  const int nbx = (h->src->n + 255) / 256;
  if ((size_t)nbx * 3 > h->fit_cap) {
    hipFree(h->d_fit);
    h->d_fit = nullptr;
    h->fit_cap = 0;
    HIP_TRY(h, hipMalloc(&h->d_fit, sizeof(double) * 3 * nbx));
    h->fit_cap = (size_t)nbx * 3;
  }
  HIP_TRY(h, hipMemcpyAsync(h->d_state, &s, sizeof(s), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipMemsetAsync(h->best_key, 0xff, sizeof(unsigned long long) * h->src->n, h->stream));
  PairDesc d;
  fill_desc(h, d, h->d_state, (h->src->n + 63) / 64);
  if (!(inlier_dist > 0.0)) inlier_dist = 0.5;  // const double max_correspondence_dist = 0.5, SMO:677
  const double inlier_sq = inlier_dist * inlier_dist;
  if (pruned) {
    // nothing farther than max(max_range, inlier_dist^2) is counted by either statistic: that is the search bound (rounded up to a float)
    const double lim = std::max(max_range, inlier_sq);
    float bf = FLT_MAX;
    if (lim < (double)FLT_MAX) {
      bf = (float)lim;
      if ((double)bf < lim) bf = std::nextafterf(bf, FLT_MAX);
    }
    const int waves = (h->src->n + 63) / 64;
    int splits = 4096 / (waves > 0 ? waves : 1);
    splits = std::min(16, std::max(1, splits));
    while (splits & (splits - 1)) splits &= splits - 1;
    launch_pruned(dim3((roundup(h->src->n, 512) + 255) / 256, splits, 1), h->stream, h->d_desc, bf, 0);
  } else {
    nn_search_kernel<<<dim3(nbx, d.nn_splits, 1), 256, 0, h->stream>>>(h->d_desc);
  }
  fitness_kernel<<<nbx, 256, 0, h->stream>>>(h->best_key, h->src->n, max_range, inlier_sq, h->d_fit);
  HIP_TRY(h, hipGetLastError());
  std::vector<double> part((size_t)nbx * 3);
  HIP_TRY(h, hipMemcpyAsync(part.data(), h->d_fit, sizeof(double) * part.size(), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  h->corr_valid = false;
  double out[3] = {0.0, 0.0, 0.0};
  for (int bk = 0; bk < nbx; ++bk)
    for (int q = 0; q < 3; ++q) out[q] += part[(size_t)bk * 3 + q];
  *score = out[1] > 0 ? out[0] / out[1] : DBL_MAX;  // pcl: returns max double when no correspondence is in range
  if (inlier_fraction) *inlier_fraction = out[2] / (double)h->src->n;
  return GORIO_OK;
}

int gorio_comm_get_unique_id(char id[128]) {
  if (!id) return GORIO_ERR_INVALID;
  Rccl& R = rccl();
  if (!R.ok) return GORIO_ERR_NO_DEVICE;
  ncclUniqueId u;
  if (R.GetUniqueId(&u) != ncclSuccess) return GORIO_ERR_NO_DEVICE;
  static_assert(sizeof(u) == 128, "ncclUniqueId is 128 bytes");
  std::memcpy(id, &u, 128);
  return GORIO_OK;
}

int gorio_apd_comm_init(gorio_apd_t* h, int world_size, int rank, const char id[128]) {
  if (!h || !id || world_size < 1 || rank < 0 || rank >= world_size) return GORIO_ERR_INVALID;
  Rccl& R = rccl();
  if (!R.ok) return fail(h, GORIO_ERR_NO_DEVICE, "librccl could not be loaded");
  HIP_TRY(h, hipSetDevice(h->device));
  if (h->comm) {
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    R.CommDestroy(h->comm);
    h->comm = nullptr;
  }
  ncclUniqueId u;
  std::memcpy(&u, id, 128);
  NCCL_TRY(h, R.CommInitRank(&h->comm, world_size, u, rank));
  h->comm_world = world_size;
  h->comm_rank = rank;
  h->shard_only = false;
  h->corr_valid = false;
  return GORIO_OK;
}

int gorio_apd_comm_info(gorio_apd_t* h, int* world_size, int* rank, long long* allreduce_count) {
  if (!h) return GORIO_ERR_INVALID;
  if (!h->comm) return fail(h, GORIO_ERR_STATE, "comm_info: the handle has no communicator");
  Rccl& R = rccl();
  int w = 0, r = 0;
  if (!R.CommCount || !R.CommUserRank) return fail(h, GORIO_ERR_NO_DEVICE, "librccl lacks ncclCommCount / ncclCommUserRank");
  NCCL_TRY(h, R.CommCount(h->comm, &w));  // what RCCL itself says, not what the caller passed to comm_init
  NCCL_TRY(h, R.CommUserRank(h->comm, &r));
  if (world_size) *world_size = w;
  if (rank) *rank = r;
  if (allreduce_count) *allreduce_count = h->allreduce_count;
  return GORIO_OK;
}

int gorio_apd_debug_set_shard(gorio_apd_t* h, int world_size, int rank) {
  if (!h || world_size < 1 || rank < 0 || rank >= world_size) return GORIO_ERR_INVALID;
  if (h->comm) return fail(h, GORIO_ERR_STATE, "debug_set_shard: the handle has a communicator");
  h->comm_world = world_size;
  h->comm_rank = rank;
  h->shard_only = world_size > 1;
  h->corr_valid = false;
  return GORIO_OK;
}

int gorio_apd_debug_set_schedule(gorio_apd_t* h, int fuse_step, int plan_search) {
  if (!h) return GORIO_ERR_INVALID;
  h->fuse_step = fuse_step != 0;
  h->plan_search = plan_search != 0;
  return GORIO_OK;
}

int gorio_apd_comm_destroy(gorio_apd_t* h) {
  if (!h) return GORIO_ERR_INVALID;
  if (h->comm) {
    hipSetDevice(h->device);
    hipStreamSynchronize(h->stream);
    if (rccl().ok) rccl().CommDestroy(h->comm);
    h->comm = nullptr;
  }
  h->comm_world = 1;
  h->comm_rank = 0;
  h->shard_only = false;
  h->corr_valid = false;
  return GORIO_OK;
}

int gorio_apd_set_profiling(gorio_apd_t* h, int enable) {
  if (!h) return GORIO_ERR_INVALID;
  h->profiling = enable != 0;
  for (int i = 0; i < 8; ++i) { h->stage_s[i] = 0; h->stage_n[i] = 0; }
  return GORIO_OK;
}

int gorio_apd_get_stage_times(gorio_apd_t* h, double seconds[8], int counts[8]) {
  if (!h) return GORIO_ERR_INVALID;
  for (int i = 0; i < 8; ++i) {
    if (seconds) seconds[i] = h->stage_s[i];
    if (counts) counts[i] = h->stage_n[i];
  }
  return GORIO_OK;
}

}  // extern "C"

// =============================================================================================== preprocessing (include/gorio_prep.h)

namespace {
thread_local std::string g_prep_err;
int prep_fail(int code, const std::string& m) {
  g_prep_err = m;
  return code;
}
struct PrepCtx {  // per thread: a private registration handle serves as the device-side cloud + search-index holder
  gorio_apd* h = nullptr;
  int device = -1;
  int* d_cnt = nullptr;
  long long* d_offs = nullptr;
  int* d_adj = nullptr;
  size_t pts_cap = 0, adj_cap = 0;
  ~PrepCtx() {
    if (h) {
      hipSetDevice(device);
      hipFree(d_cnt); hipFree(d_offs); hipFree(d_adj);
      gorio_apd_destroy(h);
    }
  }
};
thread_local PrepCtx g_prep;
}  // namespace

extern "C" {

const char* gorio_prep_last_error(void) { return g_prep_err.c_str(); }

int gorio_prep_dbscan_labels(int device, const float* xyz, int n, int point_stride_bytes, double eps, int core_min_pts, int min_cluster_size, int max_cluster_size,
                             float* label_out, int label_stride_bytes, int* n_clusters) {
  if (!xyz || !label_out || n <= 0 || point_stride_bytes < 12 || (point_stride_bytes % 4) || label_stride_bytes < 4 || (label_stride_bytes % 4))
    return prep_fail(GORIO_ERR_INVALID, "dbscan_labels: bad arguments");
  PrepCtx& c = g_prep;
  if (!c.h || c.device != device) {
    if (c.h) {
      hipSetDevice(c.device);
      hipFree(c.d_cnt); hipFree(c.d_offs); hipFree(c.d_adj);
      gorio_apd_destroy(c.h);
      c = PrepCtx();
    }
    const int rc = gorio_apd_create(&c.h, device);
    if (rc) return prep_fail(rc, "dbscan_labels: no usable HIP device (there is no CPU fallback)");
    c.device = device;
  }
  gorio_apd* h = c.h;
  h->params.search = GORIO_SEARCH_PRUNED;
  int rc = gorio_apd_set_source(h, xyz, nullptr, n, point_stride_bytes);
  if (rc) return prep_fail(rc, h->err);
#define PREP_HIP(expr)                                                                                   \
  do {                                                                                                   \
    hipError_t e_ = (expr);                                                                              \
    if (e_ != hipSuccess) return prep_fail(GORIO_ERR_NO_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)
  {
    std::vector<std::pair<gorio_apd*, DevCloud*>> one = {{h, h->src.get()}};
    rc = run_index_build(h, one);
    if (rc) return prep_fail(rc, h->err);
  }
  if ((size_t)n > c.pts_cap) {
    hipFree(c.d_cnt); hipFree(c.d_offs);
    c.d_cnt = nullptr; c.d_offs = nullptr;
    c.pts_cap = 0;
    PREP_HIP(hipMalloc(&c.d_cnt, sizeof(int) * ((size_t)n + n / 8)));
    PREP_HIP(hipMalloc(&c.d_offs, sizeof(long long) * ((size_t)n + n / 8)));
    c.pts_cap = (size_t)n + n / 8;
  }
  const CloudView cv = h->src->view();
  const int grid = (roundup(n, 512) + 255) / 256;
  RadiusArgs ra{eps, c.d_cnt, c.d_offs, nullptr};
  radius_neighbours_kernel<0><<<grid, 256, 0, h->stream>>>(cv, ra);
  PREP_HIP(hipGetLastError());
  std::vector<int> cnt((size_t)n);
  PREP_HIP(hipMemcpyAsync(cnt.data(), c.d_cnt, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, h->stream));
  PREP_HIP(hipStreamSynchronize(h->stream));
  std::vector<long long> offs((size_t)n + 1);
  offs[0] = 0;
  for (int i = 0; i < n; ++i) offs[i + 1] = offs[i] + cnt[i];
  const long long E = offs[n];
  if ((size_t)E > c.adj_cap) {
    hipFree(c.d_adj);
    c.d_adj = nullptr;
    c.adj_cap = 0;
    PREP_HIP(hipMalloc(&c.d_adj, sizeof(int) * ((size_t)E + (size_t)E / 8 + 16)));
    c.adj_cap = (size_t)E + (size_t)E / 8 + 16;
  }
  PREP_HIP(hipMemcpyAsync(c.d_offs, offs.data(), sizeof(long long) * (size_t)n, hipMemcpyHostToDevice, h->stream));
  ra.adj = c.d_adj;
  radius_neighbours_kernel<1><<<grid, 256, 0, h->stream>>>(cv, ra);
  PREP_HIP(hipGetLastError());
  std::vector<int> adj((size_t)E);
  if (E > 0) PREP_HIP(hipMemcpyAsync(adj.data(), c.d_adj, sizeof(int) * (size_t)E, hipMemcpyDeviceToHost, h->stream));
  PREP_HIP(hipStreamSynchronize(h->stream));
#undef PREP_HIP

  // ---- the queue of DBSCAN_simple.h:28-100, statement for statement, over the adjacency (the radius searches are done)
  enum : unsigned char { UN = 0, PROCESSING = 1, PROCESSED = 2 };
  std::vector<unsigned char> types((size_t)n, UN), noise((size_t)n, 0);
  std::vector<int> queue;
  std::vector<std::vector<int>> clusters;
  auto seed_count = [&](int i) { return cnt[i]; };  // |N(i, seed radius)|, the point itself included
  auto exp_count = [&](int i) {
    int k = 0;
    for (long long e = offs[i]; e < offs[i + 1]; ++e) k += (adj[(size_t)e] < 0);
    return k;
  };
  for (int i = 0; i < n; ++i) {
    if (types[i] == PROCESSED) continue;
    if (seed_count(i) < core_min_pts) {
      noise[i] = 1;
      continue;
    }
    queue.clear();
    queue.push_back(i);
    types[i] = PROCESSED;
    for (long long e = offs[i]; e < offs[i + 1]; ++e) {
      const int j = adj[(size_t)e] & 0x7fffffff;
      if (j != i) {
        queue.push_back(j);  // DBS:50-54: whatever its state
        types[j] = PROCESSING;
      }
    }
    size_t sq = 1;
    while (sq < queue.size()) {
      const int q = queue[sq];
      if (noise[q] || types[q] == PROCESSED) {
        types[q] = PROCESSED;
        sq++;
        continue;
      }
      if (exp_count(q) >= core_min_pts) {
        for (long long e = offs[q]; e < offs[q + 1]; ++e) {
          if (adj[(size_t)e] >= 0) continue;  // outside the expansion radius
          const int j = adj[(size_t)e] & 0x7fffffff;
          if (types[j] == UN) {
            queue.push_back(j);
            types[j] = PROCESSING;
          }
        }
      }
      types[q] = PROCESSED;
      sq++;
    }
    if ((int)queue.size() >= min_cluster_size && (int)queue.size() <= max_cluster_size) clusters.push_back(queue);  // DBS:83-95
  }
  // ---- preprocessing_nodelet_ntu.cpp:533-568: rank the clusters by the distance of their centroid, write rank + 1
  const int st = point_stride_bytes / 4, lst = label_stride_bytes / 4;
  for (int i = 0; i < n; ++i) label_out[(size_t)i * lst] = 0.0f;
  const int nc = (int)clusters.size();
  std::vector<float> dist((size_t)nc);
  std::vector<int> order((size_t)nc);
  for (int cidx = 0; cidx < nc; ++cidx) {
    std::vector<int>& m = clusters[cidx];
    std::sort(m.begin(), m.end());  // DBS:91
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int idx : m) {
      const float* p = xyz + (size_t)idx * st;
      sx += p[0]; sy += p[1]; sz += p[2];
    }
    const int num = (int)m.size();
    const float cx = sx / num, cy = sy / num, cz = sz / num;
    dist[cidx] = (float)std::sqrt((double)cx * cx + (double)cy * cy + (double)cz * cz);  // std::hypot(float, float, float)
    order[cidx] = cidx;
  }
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return dist[a] < dist[b]; });
  for (int r = 0; r < nc; ++r)
    for (int idx : clusters[order[r]]) label_out[(size_t)idx * lst] = (float)(r + 1);
  if (n_clusters) *n_clusters = nc;
  return GORIO_OK;
}

int gorio_prep_radius_outlier_mask(int device, const float* xyz, int n, int point_stride_bytes, double radius, int min_neighbors, unsigned char* keep, int* n_kept) {
  if (!xyz || !keep || n <= 0 || point_stride_bytes < 12 || (point_stride_bytes % 4) || !(radius > 0.0) || min_neighbors < 0)
    return prep_fail(GORIO_ERR_INVALID, "radius_outlier_mask: bad arguments");
  PrepCtx& c = g_prep;
  if (!c.h || c.device != device) {
    if (c.h) {
      hipSetDevice(c.device);
      hipFree(c.d_cnt); hipFree(c.d_offs); hipFree(c.d_adj);
      gorio_apd_destroy(c.h);
      c = PrepCtx();
    }
    const int rc = gorio_apd_create(&c.h, device);
    if (rc) return prep_fail(rc, "radius_outlier_mask: no usable HIP device (there is no CPU fallback)");
    c.device = device;
  }
  gorio_apd* h = c.h;
  h->params.search = GORIO_SEARCH_PRUNED;
  int rc = gorio_apd_set_source(h, xyz, nullptr, n, point_stride_bytes);
  if (rc) return prep_fail(rc, h->err);
  {
    std::vector<std::pair<gorio_apd*, DevCloud*>> one = {{h, h->src.get()}};
    rc = run_index_build(h, one);
    if (rc) return prep_fail(rc, h->err);
  }
  auto hip_fail = [&](const char* what, hipError_t e) { return prep_fail(GORIO_ERR_NO_DEVICE, std::string(what) + ": " + hipGetErrorString(e)); };
  if ((size_t)n > c.pts_cap) {
    hipFree(c.d_cnt); hipFree(c.d_offs);
    c.d_cnt = nullptr; c.d_offs = nullptr;
    c.pts_cap = 0;
    hipError_t e = hipMalloc(&c.d_cnt, sizeof(int) * ((size_t)n + n / 8));
    if (e == hipSuccess) e = hipMalloc(&c.d_offs, sizeof(long long) * ((size_t)n + n / 8));
    if (e != hipSuccess) return hip_fail("hipMalloc", e);
    c.pts_cap = (size_t)n + n / 8;
  }
  const double r2d = radius * radius;
  float r2 = FLT_MAX;
  if (r2d < (double)FLT_MAX) {  // largest float whose double value is <= r^2: (double)d <= r^2  <=>  d <= r2 for every float d
    r2 = (float)r2d;
    while ((double)r2 > r2d) r2 = std::nextafterf(r2, 0.0f);
  }
  radius_count_kernel<<<(roundup(n, 512) + 255) / 256, 256, 0, h->stream>>>(h->src->view(), r2, c.d_cnt);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail("radius_count_kernel", e);
  std::vector<int> cnt((size_t)n);
  e = hipMemcpyAsync(cnt.data(), c.d_cnt, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  if (e != hipSuccess) return hip_fail("download", e);
  int kept = 0;
  for (int i = 0; i < n; ++i) {
    keep[i] = cnt[i] > min_neighbors ? 1 : 0;  // the query itself is one of the counted points
    kept += keep[i];
  }
  if (n_kept) *n_kept = kept;
  return GORIO_OK;
}

int gorio_prep_statistical_outlier_mask(int device, const float* xyz, int n, int point_stride_bytes, int mean_k, double stddev_mul, unsigned char* keep, int* n_kept,
                                        float* mean_dist_out) {
  if (!xyz || !keep || n <= 0 || point_stride_bytes < 12 || (point_stride_bytes % 4) || mean_k < 1 || mean_k > 31)
    return prep_fail(GORIO_ERR_INVALID, "statistical_outlier_mask: bad arguments (mean_k must lie in [1, 31])");
  if (n < mean_k + 1) return prep_fail(GORIO_ERR_INVALID, "statistical_outlier_mask: fewer points than mean_k + 1 (PCL then sums distances nearestKSearch never set)");
  PrepCtx& c = g_prep;
  if (!c.h || c.device != device) {
    if (c.h) {
      hipSetDevice(c.device);
      hipFree(c.d_cnt); hipFree(c.d_offs); hipFree(c.d_adj);
      gorio_apd_destroy(c.h);
      c = PrepCtx();
    }
    const int rc = gorio_apd_create(&c.h, device);
    if (rc) return prep_fail(rc, "statistical_outlier_mask: no usable HIP device (there is no CPU fallback)");
    c.device = device;
  }
  gorio_apd* h = c.h;
  h->params.search = GORIO_SEARCH_PRUNED;
  int rc = gorio_apd_set_source(h, xyz, nullptr, n, point_stride_bytes);
  if (rc) return prep_fail(rc, h->err);
  {
    std::vector<std::pair<gorio_apd*, DevCloud*>> one = {{h, h->src.get()}};
    rc = run_index_build(h, one);
    if (rc) return prep_fail(rc, h->err);
  }
  auto hip_fail = [&](const char* what, hipError_t e) { return prep_fail(GORIO_ERR_NO_DEVICE, std::string(what) + ": " + hipGetErrorString(e)); };
  if ((size_t)n > c.pts_cap) {
    hipFree(c.d_cnt); hipFree(c.d_offs);
    c.d_cnt = nullptr; c.d_offs = nullptr;
    c.pts_cap = 0;
    hipError_t e = hipMalloc(&c.d_cnt, sizeof(int) * ((size_t)n + n / 8));
    if (e == hipSuccess) e = hipMalloc(&c.d_offs, sizeof(long long) * ((size_t)n + n / 8));
    if (e != hipSuccess) return hip_fail("hipMalloc", e);
    c.pts_cap = (size_t)n + n / 8;
  }
  float* d_mean = reinterpret_cast<float*>(c.d_cnt);  // one 4-byte word per point, like the neighbour counts
  sor_mean_distance_kernel<<<(roundup(n, 512) + 255) / 256, 256, 0, h->stream>>>(h->src->view(), mean_k + 1, d_mean);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail("sor_mean_distance_kernel", e);
  std::vector<float> dist((size_t)n);
  e = hipMemcpyAsync(dist.data(), d_mean, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  if (e != hipSuccess) return hip_fail("download", e);
  // mean and standard deviation of the per-point values and the threshold, in PCL's order and types (statistical_outlier_removal.hpp:
  // double sums over the float distances in point order, the n - 1 form of the variance); N numbers: done on the host
  double sum = 0.0, sq_sum = 0.0;
  for (int i = 0; i < n; ++i) {
    sum += dist[i];
    sq_sum += dist[i] * dist[i];
  }
  const double mean = sum / static_cast<double>(n);
  const double variance = (sq_sum - sum * sum / static_cast<double>(n)) / (static_cast<double>(n) - 1);
  const double stddev = std::sqrt(variance);
  const double threshold = mean + stddev_mul * stddev;
  int kept = 0;
  for (int i = 0; i < n; ++i) {
    keep[i] = dist[i] <= threshold ? 1 : 0;  // negative_ = false: the inliers stay
    kept += keep[i];
    if (mean_dist_out) mean_dist_out[i] = dist[i];
  }
  if (n_kept) *n_kept = kept;
  return GORIO_OK;
}

int gorio_prep_voxel_downsample(int device, const float* xyz, int n, int point_stride_bytes, double leaf, float* xyz_out, int out_stride_bytes, int out_capacity, int* n_out) {
  if (!xyz || !xyz_out || !n_out || n <= 0 || point_stride_bytes < 12 || (point_stride_bytes % 4) || out_stride_bytes < 12 || (out_stride_bytes % 4) || !(leaf > 0.0))
    return prep_fail(GORIO_ERR_INVALID, "voxel_downsample: bad arguments");
  PrepCtx& c = g_prep;
  if (!c.h || c.device != device) {
    if (c.h) {
      hipSetDevice(c.device);
      hipFree(c.d_cnt); hipFree(c.d_offs); hipFree(c.d_adj);
      gorio_apd_destroy(c.h);
      c = PrepCtx();
    }
    const int rc = gorio_apd_create(&c.h, device);
    if (rc) return prep_fail(rc, "voxel_downsample: no usable HIP device (there is no CPU fallback)");
    c.device = device;
  }
  // the scan-to-submap assembly with ONE frame and the identity pose is exactly pcl::VoxelGrid on that cloud (the float transform by
  // the identity returns every coordinate unchanged)
  const double eye[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  gorio_apd_keyframe fr;
  fr.xyz = xyz;
  fr.label = nullptr;
  fr.n = n;
  fr.point_stride_bytes = point_stride_bytes;
  fr.rel_pose = eye;
  int m = 0;
  int rc = gorio_apd_set_target_submap(c.h, &fr, 1, leaf, &m);
  if (rc) return prep_fail(rc, c.h->err);
  *n_out = m;
  if (m > out_capacity) return prep_fail(GORIO_ERR_INVALID, "voxel_downsample: output capacity too small (n_out holds the size needed)");
  rc = gorio_apd_get_target_points(c.h, xyz_out, nullptr, m, out_stride_bytes);
  if (rc) return prep_fail(rc, c.h->err);
  return GORIO_OK;
}

}  // extern "C"

// ----------------------------------------------------------------------------------------------- REVE (include/gorio_prep.h)
namespace {

void host_ldlt3_solve(const double* A_in, const double* rhs, double* x) {  // Eigen::LDLT<3x3>: the 6 x 6 routine of the kernels, n = 3
  double A[9];
  int perm[3] = {0, 1, 2};
  std::memcpy(A, A_in, sizeof(A));
  for (int k = 0; k < 3; ++k) {
    int piv = k;
    double best = std::fabs(A[k * 3 + k]);
    for (int i = k + 1; i < 3; ++i)
      if (std::fabs(A[i * 3 + i]) > best) {
        best = std::fabs(A[i * 3 + i]);
        piv = i;
      }
    if (piv != k) {
      for (int c = 0; c < 3; ++c) std::swap(A[k * 3 + c], A[piv * 3 + c]);
      for (int r = 0; r < 3; ++r) std::swap(A[r * 3 + k], A[r * 3 + piv]);
      std::swap(perm[k], perm[piv]);
    }
    const double d = A[k * 3 + k];
    if (d == 0.0) continue;
    double col[3];
    for (int i = k + 1; i < 3; ++i) col[i] = A[i * 3 + k];
    for (int i = k + 1; i < 3; ++i) {
      const double l = col[i] / d;
      for (int j = k + 1; j <= i; ++j) A[i * 3 + j] -= l * col[j];
      A[i * 3 + k] = l;
    }
    for (int i = k + 1; i < 3; ++i)
      for (int j = i + 1; j < 3; ++j) A[i * 3 + j] = A[j * 3 + i];
  }
  double y[3];
  for (int i = 0; i < 3; ++i) y[i] = rhs[perm[i]];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < i; ++j) y[i] -= A[i * 3 + j] * y[j];
  for (int i = 0; i < 3; ++i) y[i] = (A[i * 3 + i] != 0.0) ? y[i] / A[i * 3 + i] : 0.0;
  for (int i = 2; i >= 0; --i)
    for (int j = i + 1; j < 3; ++j) y[i] -= A[j * 3 + i] * y[j];
  for (int i = 0; i < 3; ++i) x[perm[i]] = y[i];
}

struct ReveCtx {
  int device = -1;
  hipStream_t stream = nullptr;
  float* d_in = nullptr;
  double* d_f = nullptr;
  double* d_fv = nullptr;
  unsigned char* d_valid = nullptr;
  unsigned char* d_flags = nullptr;
  double* d_v = nullptr;
  double* d_out = nullptr;
  size_t cap = 0, flags_cap = 0;
};
thread_local ReveCtx g_reve;

}  // namespace

extern "C" {

void gorio_prep_reve_default_config(gorio_reve_config* c) {  // radar_ego_velocity_estimator.h:30-60
  if (!c) return;
  std::memset(c, 0, sizeof(*c));
  c->min_dist = 1; c->max_dist = 400; c->min_db = 0; c->elevation_thresh_deg = 22.5f; c->azimuth_thresh_deg = 56.5f; c->doppler_velocity_correction_factor = 1;
  c->thresh_zero_velocity = 0.05f; c->allowed_outlier_percentage = 0.30f; c->sigma_zero_velocity_x = 1.0e-03f; c->sigma_zero_velocity_y = 3.2e-03f; c->sigma_zero_velocity_z = 1.0e-02f;
  c->max_sigma_x = 0.2f; c->max_sigma_y = 0.2f; c->max_sigma_z = 0.2f; c->inlier_thresh = 0.5f; c->use_ransac = 1; c->n_ransac_points = 5;
  c->outlier_prob = 0.05f; c->success_prob = 0.995f;
}

int gorio_prep_reve_ransac_iterations(const gorio_reve_config* c) {  // setRansacIter, radar_ego_velocity_estimator.h:138-141
  if (!c) return 0;
  return (int)(unsigned int)((std::log(1.0 - c->success_prob)) / std::log(1.0 - std::pow(1.0 - c->outlier_prob, (float)c->n_ransac_points)));
}

int gorio_prep_ego_velocity(int device, const float* xyz, const float* intensity, const float* doppler, int n, int stride_bytes, const gorio_reve_config* cfg,
                            const unsigned int* sample_idx, int n_iter, double v_r[3], double sigma_v_r[3], unsigned char* inlier_mask, unsigned char* outlier_mask,
                            int* n_valid, int* zero_velocity, int* success) {
  if (!xyz || !intensity || !doppler || !cfg || !v_r || !sigma_v_r || n <= 0 || stride_bytes < 4 || (stride_bytes % 4) || n_iter < 0 || (n_iter > 0 && !sample_idx) || cfg->n_ransac_points < 3 || cfg->n_ransac_points > 64)
    return prep_fail(GORIO_ERR_INVALID, "ego_velocity: bad arguments");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return prep_fail(GORIO_ERR_NO_DEVICE, "ego_velocity: no usable HIP device (there is no CPU fallback)");
  if (device < 0 || device >= ndev) return prep_fail(GORIO_ERR_INVALID, "ego_velocity: bad device ordinal");
#define REVE_HIP(expr)                                                                                   \
  do {                                                                                                   \
    hipError_t e_ = (expr);                                                                              \
    if (e_ != hipSuccess) return prep_fail(GORIO_ERR_NO_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)
  REVE_HIP(hipSetDevice(device));
  ReveCtx& c = g_reve;
  if (c.device != device) {
    c = ReveCtx();
    c.device = device;
    c.stream = device_stream(device);
    if (!c.stream) return prep_fail(GORIO_ERR_NO_DEVICE, "ego_velocity: no stream");
    REVE_HIP(hipMalloc(&c.d_v, sizeof(double) * 3 * 64));
  }
  if ((size_t)n > c.cap) {
    hipFree(c.d_in); hipFree(c.d_f); hipFree(c.d_fv); hipFree(c.d_valid); hipFree(c.d_out);
    const size_t cap = (size_t)n + n / 8;
    REVE_HIP(hipMalloc(&c.d_in, sizeof(float) * 5 * cap));
    REVE_HIP(hipMalloc(&c.d_f, sizeof(double) * 4 * cap));
    REVE_HIP(hipMalloc(&c.d_fv, sizeof(double) * 4 * cap));
    REVE_HIP(hipMalloc(&c.d_valid, cap));
    REVE_HIP(hipMalloc(&c.d_out, sizeof(double) * 10 * (cap / 256 + 2)));
    c.cap = cap;
  }
  // ---- per-target features on the device
  const int st = stride_bytes / 4;
  std::vector<float> in((size_t)n * 5);
  for (int i = 0; i < n; ++i) {
    in[5 * (size_t)i] = xyz[(size_t)i * st]; in[5 * (size_t)i + 1] = xyz[(size_t)i * st + 1]; in[5 * (size_t)i + 2] = xyz[(size_t)i * st + 2];
    in[5 * (size_t)i + 3] = intensity[(size_t)i * st]; in[5 * (size_t)i + 4] = doppler[(size_t)i * st];
  }
  REVE_HIP(hipMemcpyAsync(c.d_in, in.data(), sizeof(float) * in.size(), hipMemcpyHostToDevice, c.stream));
  ReveCfg rc;
  rc.min_dist = cfg->min_dist; rc.max_dist = cfg->max_dist; rc.min_db = cfg->min_db;
  rc.az_lim = (double)cfg->azimuth_thresh_deg * M_PI / 180.0; rc.el_lim = (double)cfg->elevation_thresh_deg * M_PI / 180.0;  // angles::from_degrees
  rc.doppler_factor_unused = 0; rc.doppler_factor = cfg->doppler_velocity_correction_factor; rc.pad_ = 0;
  reve_features_kernel<<<(n + 255) / 256, 256, 0, c.stream>>>(c.d_in, c.d_in + 3, c.d_in + 4, 5, n, rc, c.d_f, c.d_valid);
  REVE_HIP(hipGetLastError());
  std::vector<double> f((size_t)n * 4);
  std::vector<unsigned char> valid((size_t)n);
  REVE_HIP(hipMemcpyAsync(f.data(), c.d_f, sizeof(double) * f.size(), hipMemcpyDeviceToHost, c.stream));
  REVE_HIP(hipMemcpyAsync(valid.data(), c.d_valid, (size_t)n, hipMemcpyDeviceToHost, c.stream));
  REVE_HIP(hipStreamSynchronize(c.stream));
  std::vector<int> vidx;
  std::vector<double> fv;
  for (int i = 0; i < n; ++i)
    if (valid[i]) {
      vidx.push_back(i);
      fv.insert(fv.end(), f.begin() + 4 * (size_t)i, f.begin() + 4 * (size_t)i + 4);
    }
  const int m = (int)vidx.size();
  if (inlier_mask) std::memset(inlier_mask, 0, (size_t)n);
  if (outlier_mask) std::memset(outlier_mask, 0, (size_t)n);
  v_r[0] = v_r[1] = v_r[2] = 0.0;
  sigma_v_r[0] = sigma_v_r[1] = sigma_v_r[2] = 0.0;
  if (n_valid) *n_valid = m;
  if (zero_velocity) *zero_velocity = 0;
  int ok = 0;
  // solve3DFull (REVE:252-303) over the valid rows selected by `sel`: sums on the device, 3 x 3 algebra here
  auto solve = [&](const std::vector<unsigned char>& sel, int rows, bool estimate_sigma, double* v, double* sigma) -> int {
    const int nb = (m + 255) / 256;
    if (hipMemcpyAsync(c.d_valid, sel.data(), (size_t)m, hipMemcpyHostToDevice, c.stream) != hipSuccess) return -1;
    std::vector<double> part((size_t)nb * 10);
    double s[10];
    for (int pass = 0; pass < (estimate_sigma ? 2 : 1); ++pass) {
      if (pass == 1 && hipMemcpyAsync(c.d_v, v, sizeof(double) * 3, hipMemcpyHostToDevice, c.stream) != hipSuccess) return -1;
      reve_sums_kernel<<<nb, 256, 0, c.stream>>>(c.d_fv, m, c.d_valid, pass == 1 ? c.d_v : nullptr, c.d_out);
      if (hipMemcpyAsync(part.data(), c.d_out, sizeof(double) * part.size(), hipMemcpyDeviceToHost, c.stream) != hipSuccess || hipStreamSynchronize(c.stream) != hipSuccess) return -1;
      for (int q = 0; q < 10; ++q) s[q] = 0.0;
      for (int b = 0; b < nb; ++b)
        for (int q = 0; q < 10; ++q) s[q] += part[(size_t)b * 10 + q];
      if (pass == 0) {
        const double HTH[9] = {s[0], s[1], s[2], s[1], s[3], s[4], s[2], s[4], s[5]}, HTy[3] = {s[6], s[7], s[8]};
        host_ldlt3_solve(HTH, HTy, v);  // use_cholesky_instead_of_bdcsvd = true: (HTH).ldlt().solve(H^T y), REVE:272
      }
    }
    if (estimate_sigma) {  // REVE:278-290
      const double H0 = s[0], H1 = s[1], H2 = s[2], H4 = s[3], H5 = s[4], H8 = s[5];
      const double c00 = H4 * H8 - H5 * H5, c01 = H5 * H2 - H1 * H8, c02 = H1 * H5 - H4 * H2;
      const double det = H0 * c00 + H1 * c01 + H2 * c02;
      const double sc = s[9] / (double)(rows - 3);
      double sg[3] = {sc * (c00 / det), sc * ((H0 * H8 - H2 * H2) / det), sc * ((H0 * H4 - H1 * H1) / det)};
      sigma[0] = sg[0]; sigma[1] = sg[1]; sigma[2] = sg[2];
      if (sg[0] >= 0.0 && sg[1] >= 0.0 && sg[2] >= 0.0) {
        sigma[0] = std::sqrt(sg[0]) + cfg->sigma_offset_radar_x;
        sigma[1] = std::sqrt(sg[1]) + cfg->sigma_offset_radar_y;
        sigma[2] = std::sqrt(sg[2]) + cfg->sigma_offset_radar_z;
      }
    }
    return 0;
  };
  if (m > 2) {
    REVE_HIP(hipMemcpyAsync(c.d_fv, fv.data(), sizeof(double) * fv.size(), hipMemcpyHostToDevice, c.stream));
    std::vector<double> vd((size_t)m);
    for (int k = 0; k < m; ++k) vd[k] = std::fabs(fv[4 * (size_t)k + 3]);
    const size_t nth = std::min((size_t)((double)m * (1.0 - (double)cfg->allowed_outlier_percentage)), (size_t)m - 1);
    std::nth_element(vd.begin(), vd.begin() + nth, vd.end());  // REVE:105-108
    if (vd[nth] < cfg->thresh_zero_velocity) {                 // REVE:110-121
      if (zero_velocity) *zero_velocity = 1;
      sigma_v_r[0] = cfg->sigma_zero_velocity_x; sigma_v_r[1] = cfg->sigma_zero_velocity_y; sigma_v_r[2] = cfg->sigma_zero_velocity_z;
      if (inlier_mask)
        for (int k = 0; k < m; ++k)
          if (std::fabs(fv[4 * (size_t)k + 3]) < cfg->thresh_zero_velocity) inlier_mask[vidx[k]] = 1;
      ok = 1;
    } else if (!cfg->use_ransac) {
      std::vector<unsigned char> all((size_t)m, 1);
      if (solve(all, m, true, v_r, sigma_v_r)) return prep_fail(GORIO_ERR_NO_DEVICE, "ego_velocity: device error");
      if (inlier_mask) for (int k = 0; k < m; ++k) inlier_mask[vidx[k]] = 1;
      ok = 1;
    } else {  // solve3DFullRansac, REVE:172-250
      std::vector<unsigned char> best_in, best_out;
      size_t nbi = 0, nbo = 0;
      const int K = m >= cfg->n_ransac_points ? n_iter : 0;
      if (K > 0) {
        if (K > 64) return prep_fail(GORIO_ERR_INVALID, "ego_velocity: more than 64 RANSAC iterations");
        std::vector<double> vs((size_t)K * 3);
        for (int k = 0; k < K; ++k) {  // the sample systems are N_ransac_points rows: solved here (3 x 3)
          double HTH[9] = {0}, HTy[3] = {0};
          for (int q = 0; q < cfg->n_ransac_points; ++q) {
            const unsigned int row = sample_idx[(size_t)k * cfg->n_ransac_points + q];
            if (row >= (unsigned int)m) return prep_fail(GORIO_ERR_INVALID, "ego_velocity: sample index outside the valid targets");
            const double* r = fv.data() + 4 * (size_t)row;
            for (int a = 0; a < 3; ++a) {
              for (int b = 0; b < 3; ++b) HTH[a * 3 + b] += r[a] * r[b];
              HTy[a] += r[a] * r[3];
            }
          }
          host_ldlt3_solve(HTH, HTy, vs.data() + 3 * (size_t)k);
        }
        if ((size_t)K * m > c.flags_cap) {
          hipFree(c.d_flags);
          c.d_flags = nullptr;
          c.flags_cap = 0;
          REVE_HIP(hipMalloc(&c.d_flags, (size_t)K * m + 1024));
          c.flags_cap = (size_t)K * m + 1024;
        }
        REVE_HIP(hipMemcpyAsync(c.d_v, vs.data(), sizeof(double) * vs.size(), hipMemcpyHostToDevice, c.stream));
        reve_eval_kernel<<<dim3((m + 255) / 256, K), 256, 0, c.stream>>>(c.d_fv, m, c.d_v, (double)cfg->inlier_thresh, c.d_flags);
        REVE_HIP(hipGetLastError());
        std::vector<unsigned char> flags((size_t)K * m);
        REVE_HIP(hipMemcpyAsync(flags.data(), c.d_flags, flags.size(), hipMemcpyDeviceToHost, c.stream));
        REVE_HIP(hipStreamSynchronize(c.stream));
        for (int k = 0; k < K; ++k) {
          const unsigned char* fl = flags.data() + (size_t)k * m;
          size_t ni = 0;
          for (int j = 0; j < m; ++j) ni += fl[j];
          size_t no = (size_t)m - ni;
          std::vector<unsigned char> cur_in(fl, fl + m), cur_out((size_t)m);
          for (int j = 0; j < m; ++j) cur_out[j] = !fl[j];
          if ((float)no / (float)(ni + no) > 0.05) {  // REVE:215-220
            std::fill(cur_in.begin(), cur_in.end(), 1);
            std::fill(cur_out.begin(), cur_out.end(), 0);
            ni = (size_t)m;
            no = 0;
          }
          if (ni > nbi) { best_in = cur_in; nbi = ni; }
          if (no > nbo) { best_out = cur_out; nbo = no; }
          v_r[0] = vs[3 * (size_t)k]; v_r[1] = vs[3 * (size_t)k + 1]; v_r[2] = vs[3 * (size_t)k + 2];
        }
      }
      if (nbi > 0) {
        if (solve(best_in, (int)nbi, true, v_r, sigma_v_r)) return prep_fail(GORIO_ERR_NO_DEVICE, "ego_velocity: device error");
        ok = 1;  // REVE:301: true whatever the sigma test said
        if (inlier_mask) for (int j = 0; j < m; ++j) if (best_in[j]) inlier_mask[vidx[j]] = 1;
      }
      if (outlier_mask && nbo > 0) for (int j = 0; j < m; ++j) if (best_out[j]) outlier_mask[vidx[j]] = 1;
    }
  }
#undef REVE_HIP
  if (success) *success = ok;
  return GORIO_OK;
}

}  // extern "C"

#ifdef GORIO_STATS
extern "C" int gorio_debug_search_stats(unsigned long long out[24], int reset) {
  static std::vector<unsigned long long> all(1024 * 24);
  if (hipMemcpyFromSymbol(all.data(), HIP_SYMBOL(gorio::g_search_stats), sizeof(unsigned long long) * 1024 * 24) != hipSuccess) return -1;
  for (int k = 0; k < 24; ++k) {
    out[k] = 0;
    for (int r = 0; r < 1024; ++r) out[k] = (k == 2 && false) ? std::max(out[k], all[(size_t)r * 24 + k]) : out[k] + all[(size_t)r * 24 + k];
  }
  if (reset) {
    std::fill(all.begin(), all.end(), 0ull);
    if (hipMemcpyToSymbol(HIP_SYMBOL(gorio::g_search_stats), all.data(), sizeof(unsigned long long) * 1024 * 24) != hipSuccess) return -1;
  }
  return 0;
}
#endif
