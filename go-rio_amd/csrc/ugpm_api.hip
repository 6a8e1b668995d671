// ugpm_api.hip -- host side of the UGPM C ABI (include/gorio_ugpm.h): window bookkeeping (state time line, sample slicing:
// preint.h:766-811), device workspace, kernel sequencing of ugpm_kernels.hip.  No numerics happen on the host and there is no
// CPU fallback.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <limits>
#include <memory>
#include <string>
#include <atomic>
#include <vector>

#include "../../include/gorio_ugpm.h"
#include "ugpm_kernels.hip"
#include "ugpm_lpm_out.hip"
#include "ugpm_chunks.h"

using namespace gorio;

namespace {

#ifndef GORIO_EVAL_SPLIT
#define GORIO_EVAL_SPLIT 12  // measured in round 3 (6 / 12 / 24): LM fits of a C4 batch alone 3.10 / 2.98 / 3.00 ms
#endif
constexpr int kEvalSplit = GORIO_EVAL_SPLIT;  // workgroups per window in the residual / Jacobian evaluators

thread_local std::string g_err;
thread_local double g_stage_s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
thread_local int g_stage_n[8] = {0, 0, 0, 0, 0, 0, 0, 0};

struct Ctx {  // per-thread, per-device cached buffers
  int device = -1;
  hipStream_t stream = nullptr;
  hipStream_t stream2 = nullptr;  // the state-correlation chain runs here, beside the two GP fits (a helper thread in the reference, preint.h:939-1064)
  hipEvent_t ev_jac = nullptr, ev_corr = nullptr, ev_up = nullptr;
  struct Group { hipStream_t s = nullptr, s2 = nullptr; hipEvent_t ev_jac = nullptr, ev_corr = nullptr, ev_done = nullptr, ev_tab = nullptr; };
  std::vector<Group> groups;  // group 0 = (stream, stream2)
  double* ws = nullptr;
  size_t ws_cap = 0;
  UgpmWin* d_wins = nullptr;
  int wins_cap = 0;
  int* d_ints = nullptr;  // per window: kWinInts ints (lmi[16], status)
  int lm_budget[2] = {0, 0};  // iterations the two fits of the PREVIOUS batch needed: that many are enqueued before the first look at the done flags
  double* d_diag = nullptr;
  double* pin_in = nullptr;  // pinned staging of the batch's input arrays (the H2D copy is then a DMA the call does not wait for)
  size_t pin_in_cap = 0;
  // opt.type = LPM windows (ugpm_lpm_out.hip)
  double* lpm_ws = nullptr;
  size_t lpm_ws_cap = 0;
  int* lpm_ints = nullptr;
  size_t lpm_ints_cap = 0;
  ug::LpmOutWin* d_lpm_wins = nullptr;
  int lpm_wins_cap = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
  std::vector<int> ev_stage;
};
thread_local Ctx g_ctx;
std::atomic<int> g_speculative_rot{1};  // gorio_ugpm_debug_set_schedule

int ufail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

#define UHIP(expr)                                                                                              \
  do {                                                                                                          \
    hipError_t e_ = (expr);                                                                                     \
    if (e_ != hipSuccess) return ufail(GORIO_UGPM_ERR_NO_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

struct HostWin {
  int g0 = 0, G = 0, v0 = 0, V = 0, S = 0;
  double state_freq = 0;
  std::vector<double> state_t;
  int status = 0;
  size_t ws_doubles = 0;
  bool is_lpm = false;  // opt.type = LPM: handled by ugpm_lpm_out.hip, skipped by every UGPM kernel
};

// GyroVelData::get(from, to): samples with from < t < to, scanning until the first t >= to (types.h:187-223)
void slice(const double* t, int n, double from, double to, int& i0, int& cnt) {
  i0 = 0;
  cnt = 0;
  if (from >= to || n <= 0) return;
  bool started = false;
  for (int i = 0; i < n; ++i) {
    if (t[i] > from) {
      if (t[i] < to) {
        if (!started) {
          i0 = i;
          started = true;
        }
        cnt++;
      } else {
        break;
      }
    }
  }
}

// carve one window's slab; returns the number of doubles used (called once with base == nullptr for sizing)
size_t input_doubles(const gorio_ugpm_window& w, const HostWin& h) { return (size_t)h.G * 4 + (size_t)h.V * 4 + (size_t)w.n_infer + (size_t)h.S; }

// `in` = this window's slice of the batch-wide contiguous input region (one upload for the whole batch), `outp` = its slice of
// the batch-wide output region (one download)
size_t carve(const gorio_ugpm_window& w, const HostWin& h, UgpmWin& u, double* base, double* in, double* outp) {
  double* p = base;
  auto take = [&](size_t cnt) { double* r = p; p += cnt; return r; };
  auto take_in = [&](size_t cnt) { double* r = in; in += cnt; return r; };
  const size_t S = h.S, G = h.G, V = h.V, n = 3 * S, mrot = 3 * S + 3 * G, mvel = 3 * V + 3 * S, mc = 3 * G + 3 * V, nc = 6 * S;
  u.gyr_t = take_in(G); u.gyr = take_in(3 * G); u.vel_t = take_in(V); u.vel = take_in(3 * V); u.infer_t = take_in(w.n_infer); u.state_t = take_in(S);
  u.Rq = take(5 * 2 * S * 9); u.Rstart = take(5 * 9); u.velr = take(3 * V); u.dp = take(2 * S * 3); u.r0 = take(5 * S * 3); u.r1 = take(5 * S * 3);
  u.s_dr = take(3 * S); u.s_vel = take(3 * S); u.hyper = take(24);
  u.d_r_dt_local = take(S * 3); u.d_r_dt_local_shift = take(S * 3); u.delta_r_time = take(S * 3); u.delta_r_bw = take(3 * S * 3); u.d_r_bw_local_shift = take(3 * S * 3);
  u.Kinv = take(6 * S * S); u.KKinv = take(6 * S * S); u.KintKinv = take(3 * S * S); u.var = take(6 * S); u.wgp = take(6 * S); u.sstd = take(6 * S);
  u.KsKinv = take(3 * G * S); u.KsIntKinv = take(3 * G * S); u.KgyrIntKinv = take(3 * V * S); u.KvelKinv = take(3 * V * S);
  u.Jrot = take(mrot * n); u.Jvel = take(mvel * n); u.res = take(std::max(mrot, mvel)); u.res_new = take(std::max(mrot, mvel));
  u.JtJ = take(n * n); u.lhs = take(n * n); u.lmv = take(8 * n); u.sample_tmp = take(std::max(G, V) * 24); u.sample_tmp_c = take(std::max(G, V) * 24);
  if (w.correlate) { u.Jc = take(mc * nc); u.Ac = take(nc * nc); }
  u.dsc = take(nc);
  u.alpha = take(6 * S); u.state_r = take(3 * S); u.d_state_bw = take(3 * S * 3); u.d_d_r_dt = take(3 * S); u.d_vel_bv = take(3 * S * 3); u.d_vel_bw = take(3 * S * 3);
  u.d_vel_dt = take(3 * S); u.out = outp; u.lmc = take(16);
  return ((size_t)(p - base) + 31) / 32 * 32;
}

struct Stage {  // HIP events around a group of launches on `stream` (default: the context's main stream); may nest
  Ctx& c;
  hipStream_t stream;
  size_t slot = 0;
  bool on;
  Stage(Ctx& c_, int s, hipStream_t st = nullptr) : c(c_), stream(st ? st : c_.stream), on(true) {
    hipEvent_t a = nullptr, b = nullptr;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) {
      on = false;
      return;
    }
    slot = c.ev.size();
    c.ev.emplace_back(a, b);
    c.ev_stage.push_back(s);
    hipEventRecord(a, stream);
  }
  ~Stage() {
    if (on) hipEventRecord(c.ev[slot].second, stream);
  }
};

// ---- opt.type = LPM (preint.h:1567-1580): host bookkeeping of one IterativeIntegrator = the merged, sorted time line
// (SortIndexTracker2, types.h:332-458) and the filler stamps of preint.h:228-237.  No numerics.
struct LpmHost {
  std::vector<double> tl;
  std::vector<int> kind, kidx, qpos, qorder, qrot;
  int start_index = 0, dt_index = 0;
};

void build_lpm_timeline(const gorio_ugpm_window& w, LpmHost& L) {
  struct Stamp { double t; int kind, idx; };
  std::vector<Stamp> st;
  st.reserve((size_t)w.n_infer + 2 + w.n_vel);
  for (int j = 0; j < w.n_infer; ++j) st.push_back({w.infer_t[j], 0, j});
  st.push_back({w.start_t, 1, 0});
  st.push_back({w.start_t + 0.01, 1, 1});  // kNumDtJacobianDelta, preint.h:216-219
  for (int i = 0; i < w.n_vel; ++i) st.push_back({w.vel_t[i], 2, i});
  auto by_time = [](const Stamp& a, const Stamp& b) { return a.t < b.t; };
  std::stable_sort(st.begin(), st.end(), by_time);
  // getSmallestGap() returns the LAST gap of the sorted line (types.h:442-450), preint.h:228
  if (st.size() >= 2 && (st.back().t - st[st.size() - 2].t) > (1.0 / w.min_freq)) {
    const double first = st.front().t, last = st.back().t;
    const int nb = (int)std::floor((last - first) * w.min_freq);
    if (nb > 0) {
      const double quantum = (last - first) / ((double)nb);
      for (int i = 0; i < nb; ++i) st.push_back({first + (i * quantum), 3, i});
      std::stable_sort(st.begin(), st.end(), by_time);
    }
  }
  const size_t T = st.size();
  L.tl.resize(T); L.kind.resize(T); L.kidx.resize(T);
  L.qpos.assign(w.n_infer, 0);
  L.qorder.clear();
  for (size_t r = 0; r < T; ++r) {
    L.tl[r] = st[r].t; L.kind[r] = st[r].kind; L.kidx[r] = st[r].idx;
    if (st[r].kind == 0) { L.qpos[st[r].idx] = (int)r; L.qorder.push_back(st[r].idx); }
    if (st[r].kind == 1 && st[r].idx == 0) L.start_index = (int)r;
    if (st[r].kind == 1 && st[r].idx == 1) L.dt_index = (int)r;
  }
  // preint_[g] = t.getVector(preint, g) (preint.h:259, types.h:378-387): the rotation part of record k of inner vector g is that of the
  // vector's k-th stamp IN SORTED ORDER; the position part is written by original index later (preint.h:640-664)
  L.qrot.assign(w.n_infer, 0);
  std::vector<int> group_of(w.n_infer, 0), first_of_group(1, 0);
  if (w.group_sizes && w.n_groups > 0) {
    int o = 0;
    first_of_group.clear();
    for (int g = 0; g < w.n_groups; ++g) {
      first_of_group.push_back(o);
      for (int k = 0; k < w.group_sizes[g] && o < w.n_infer; ++k) group_of[o++] = g;
    }
  }
  std::vector<int> filled(first_of_group.size(), 0);
  for (size_t r = 0; r < T; ++r)
    if (st[r].kind == 0) {
      const int g = group_of[st[r].idx];
      L.qrot[first_of_group[g] + filled[g]++] = (int)r;
    }
}

}  // namespace

extern "C" {

void gorio_ugpm_default_window(gorio_ugpm_window* w) {
  if (!w) return;
  std::memset(w, 0, sizeof(*w));
  w->type = GORIO_UGPM_TYPE_UGPM;  // types.h:288
  w->min_freq = 500;                // types.h:287
  w->quantum = -1;                  // types.h:289
  w->state_freq = 50.0;             // types.h:290
  w->correlate = 1;                 // types.h:291
  w->overlap = 8;                   // preint.h:19
  w->vel_bias_std = 0.3;            // preint.h:55
  w->gyr_bias_std = 0.03;
}

const char* gorio_ugpm_last_error(void) { return g_err.c_str(); }

void gorio_ugpm_debug_set_schedule(int speculative_rot) { g_speculative_rot.store(speculative_rot ? 1 : 0); }

int gorio_ugpm_get_stage_times(double seconds[8], int counts[8]) {
  for (int i = 0; i < 8; ++i) {
    if (seconds) seconds[i] = g_stage_s[i];
    if (counts) counts[i] = g_stage_n[i];
  }
  return 0;
}

// every window non-chunked (quantum < 0): the device path
static int preint_batch_flat(const gorio_ugpm_window* windows, int n_windows, gorio_ugpm_meas* out, gorio_ugpm_diag* diag, int device) {
  if (!windows || n_windows <= 0 || !out) return ufail(GORIO_UGPM_ERR_INVALID, "gorio_ugpm_preint_batch: bad arguments");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return ufail(GORIO_UGPM_ERR_NO_DEVICE, "no usable HIP device (no CPU fallback exists)");
  if (device < 0 || device >= ndev) return ufail(GORIO_UGPM_ERR_INVALID, "bad device ordinal");
  UHIP(hipSetDevice(device));
  Ctx& c = g_ctx;
  if (c.device != device) {
    for (size_t g = 1; g < c.groups.size(); ++g) {
      hipStreamDestroy(c.groups[g].s); hipStreamDestroy(c.groups[g].s2);
      hipEventDestroy(c.groups[g].ev_jac); hipEventDestroy(c.groups[g].ev_corr);
    }
    for (auto& g : c.groups) { hipEventDestroy(g.ev_done); hipEventDestroy(g.ev_tab); }
    if (c.ev_up) hipEventDestroy(c.ev_up);
    if (c.stream) hipStreamDestroy(c.stream);
    if (c.stream2) hipStreamDestroy(c.stream2);
    if (c.ev_jac) hipEventDestroy(c.ev_jac);
    if (c.ev_corr) hipEventDestroy(c.ev_corr);
    hipFree(c.ws); hipFree(c.d_wins); hipFree(c.d_ints); hipFree(c.d_diag); hipFree(c.lpm_ws); hipFree(c.lpm_ints); hipFree(c.d_lpm_wins);
    if (c.pin_in) hipHostFree(c.pin_in);
    c = Ctx();
    c.device = device;
    bool made = false;
    if (!made) {  // highest priority: these are many small latency-bound launches that should not queue behind the scan matcher's large grids
      int lo = 0, hi = 0;
      if (hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hipStreamCreateWithPriority(&c.stream, hipStreamNonBlocking, hi) == hipSuccess) made = true;
    }
    if (!made) UHIP(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
    UHIP(hipStreamCreateWithFlags(&c.stream2, hipStreamNonBlocking));
    UHIP(hipEventCreateWithFlags(&c.ev_jac, hipEventDisableTiming));
    UHIP(hipEventCreateWithFlags(&c.ev_corr, hipEventDisableTiming));
    // ata_kernel stages J through up to ~128 KB of dynamic LDS (the default limit is 64 KB)
    UHIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&ug::corr_diag_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
    UHIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&ug::infer_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 17 * (6 * 160 + 16) * 8) /* S = 160: with the 31 KB of static LDS this is just inside the 160 KB of a CU */);
    {
      const void* fns[] = {reinterpret_cast<const void*>(&ug::ata_kernel<4, 16, kAtaTilesCorr>), reinterpret_cast<const void*>(&ug::ata_kernel<8, 16, kAtaTilesCorr>),
                           reinterpret_cast<const void*>(&ug::ata_kernel<16, 8, kAtaTilesCorr>), reinterpret_cast<const void*>(&ug::ata_kernel<4, 16, kAtaTilesLm>),
                           reinterpret_cast<const void*>(&ug::ata_kernel<8, 16, kAtaTilesLm>)};
      for (const void* f : fns) UHIP(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
    }
  }
  for (int i = 0; i < 8; ++i) { g_stage_s[i] = 0; g_stage_n[i] = 0; }
  const bool trace = std::getenv("GORIO_UGPM_TRACE") != nullptr;
  const bool lmtrace = std::getenv("GORIO_UGPM_LMTRACE") != nullptr;
  auto tnow = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double tt0 = tnow();
  double tt1 = 0, tt2 = 0, tt3 = 0;

  // ---- host bookkeeping per window (preint.h:1532-1556, 766-811): no numerics beyond the state time line
  std::vector<HostWin> hw(n_windows);
  size_t total_doubles = 0, total_in = 0;
  int total_infer = 0, max_infer = 0, max_S = 0;
  int first_error = 0, n_lpm = 0;
  std::string first_error_msg;
  auto win_fail = [&](int i, int code, const std::string& m) {
    hw[i].status = code;
    if (!first_error) {
      first_error = code;
      first_error_msg = "window " + std::to_string(i) + ": " + m;
    }
  };
  for (int i = 0; i < n_windows; ++i) {
    const gorio_ugpm_window& w = windows[i];
    HostWin& h = hw[i];
    total_infer += std::max(0, w.n_infer);
    max_infer = std::max(max_infer, w.n_infer);
    if (!w.gyr_t || !w.gyr || !w.vel_t || !w.vel || !w.infer_t || w.n_infer <= 0) { win_fail(i, GORIO_UGPM_ERR_INVALID, "null pointers or no inference time"); continue; }
    if (w.quantum >= 0) { win_fail(i, GORIO_UGPM_ERR_INVALID, "a chunked request reached the device path"); continue; }  // gorio_ugpm_preint_batch expands them
    if (w.type != GORIO_UGPM_TYPE_UGPM && w.type != GORIO_UGPM_TYPE_LPM) { win_fail(i, GORIO_UGPM_ERR_INVALID, "unknown pre-integration type"); continue; }
    if (w.n_gyr < 2 || w.n_vel < 2) { win_fail(i, GORIO_UGPM_ERR_RANGE, "InterpolateLinear: this function need at least 2 data points to interpolate"); continue; }
    if (w.group_sizes && w.n_groups > 0) {
      long tot = 0;
      for (int g = 0; g < w.n_groups; ++g) tot += w.group_sizes[g] < 0 ? -(1L << 40) : w.group_sizes[g];
      if (tot != w.n_infer) { win_fail(i, GORIO_UGPM_ERR_INVALID, "group_sizes do not add up to n_infer"); continue; }
    }
    if (w.type == GORIO_UGPM_TYPE_LPM) {  // preint.h:1567-1580: IterativeIntegrator over the WHOLE data set, no state window
      if (!(w.min_freq > 0.0)) { win_fail(i, GORIO_UGPM_ERR_INVALID, "min_freq must be positive"); continue; }
      bool any = false;
      for (int j = 0; j < w.n_infer; ++j) any = any || (w.infer_t[j] >= w.start_t);
      if (!any) { win_fail(i, GORIO_UGPM_ERR_RANGE, "FullLPM: the start_time is not in the query domain"); continue; }  // preint.h:559
      h.is_lpm = true;
      h.G = w.n_gyr;
      h.V = w.n_vel;
      n_lpm++;
      continue;
    }
    const double vel_freq = (w.n_vel - 1) / (w.vel_t[w.n_vel - 1] - w.vel_t[0]);
    const double gyr_freq = (w.n_gyr - 1) / (w.gyr_t[w.n_gyr - 1] - w.gyr_t[0]);
    const double duration = *std::max_element(w.infer_t, w.infer_t + w.n_infer) - w.start_t;  // preint.h:1544-1552
    if (!(duration > 0.0) || !std::isfinite(duration)) { win_fail(i, GORIO_UGPM_ERR_ARGUMENT, "inference time is not after start_t"); continue; }
    double sf = std::max(w.state_freq, 5.0 / duration);  // preint.h:770-771
    sf = std::min(sf, std::min(vel_freq, gyr_freq));
    h.state_freq = sf;
    h.S = (int)(std::ceil(duration * sf) + (2 * w.overlap));  // preint.h:775
    if (h.S < 2 * w.overlap + 1 || h.S > 160) { win_fail(i, GORIO_UGPM_ERR_UNSUPPORTED, "number of GP states outside [2 overlap + 1, 160]"); continue; }
    h.state_t.resize(h.S);
    const double t0 = w.start_t - (((double)w.overlap) / sf);
    for (int k = 0; k < h.S; ++k) h.state_t[k] = t0 + ((double)k) / sf;  // preint.h:777-783
    if (!(h.state_t[0] <= h.state_t.back())) { win_fail(i, GORIO_UGPM_ERR_ARGUMENT, "The argument of GyroVelData::Get are not consistent"); continue; }
    slice(w.gyr_t, w.n_gyr, h.state_t[0], h.state_t.back(), h.g0, h.G);  // preint.h:789
    slice(w.vel_t, w.n_vel, h.state_t[0], h.state_t.back(), h.v0, h.V);
    if (h.G < 2 || h.V < 2) { win_fail(i, GORIO_UGPM_ERR_RANGE, "fewer than 2 gyro / velocity samples inside the state window"); continue; }
    max_S = std::max(max_S, h.S);
    UgpmWin dummy;
    h.ws_doubles = carve(w, h, dummy, nullptr, nullptr, nullptr);
    total_in += (input_doubles(w, h) + 3) / 4 * 4;
    total_doubles += h.ws_doubles;
  }
  const size_t total_out = (size_t)total_infer * 83;
  total_doubles += total_in + total_out + 64;
  if (total_doubles > c.ws_cap) {
    hipFree(c.ws);
    c.ws = nullptr;
    c.ws_cap = 0;
    UHIP(hipMalloc(&c.ws, sizeof(double) * total_doubles));
    c.ws_cap = total_doubles;
  }
  if (n_windows > c.wins_cap) {
    hipFree(c.d_wins); hipFree(c.d_ints); hipFree(c.d_diag);
    c.d_wins = nullptr; c.d_ints = nullptr; c.d_diag = nullptr;
    UHIP(hipMalloc(&c.d_wins, sizeof(UgpmWin) * n_windows));
    UHIP(hipMalloc(&c.d_ints, sizeof(int) * kWinInts * n_windows));
    UHIP(hipMalloc(&c.d_diag, sizeof(double) * 4 * n_windows));
    c.wins_cap = n_windows;
  }

  // ---- carve the workspace, stage the inputs
  std::vector<UgpmWin> dw(n_windows);
  std::vector<int> ints(kWinInts * (size_t)n_windows, 0);
  double* in_region = c.ws;
  double* out_region = c.ws + total_in;
  double* base = out_region + (total_out + 31) / 32 * 32;
  if (total_in > c.pin_in_cap) {
    if (c.pin_in) hipHostFree(c.pin_in);
    c.pin_in = nullptr;
    c.pin_in_cap = 0;
    UHIP(hipHostMalloc(reinterpret_cast<void**>(&c.pin_in), sizeof(double) * (total_in + total_in / 4 + 64), hipHostMallocDefault));
    c.pin_in_cap = total_in + total_in / 4 + 64;
  }
  double* const stage_in = c.pin_in;  // every use of it ends before this call returns (the call ends with a stream synchronisation)
  size_t in_off = 0, out_off = 0;
  std::vector<size_t> out_offs(n_windows, 0);
  for (int i = 0; i < n_windows; ++i) {
    const gorio_ugpm_window& w = windows[i];
    HostWin& h = hw[i];
    UgpmWin& u = dw[i];
    std::memset(&u, 0, sizeof(u));
    u.lmi = c.d_ints + kWinInts * (size_t)i;
    u.status = c.d_ints + kWinInts * (size_t)i + 16;
    ints[kWinInts * (size_t)i + 16] = h.status;
    u.n_infer = std::max(0, w.n_infer);
    out_offs[i] = out_off;
    u.out = out_region + out_off;
    out_off += (size_t)u.n_infer * 83;
    if (h.status != 0) continue;
    if (h.is_lpm) {
      ints[kWinInts * (size_t)i + 16] = 1;  // every UGPM kernel skips this window; its own status word is slot 17
      u.n_infer = 0;                        // and infer_kernel writes nothing for it
      continue;
    }
    carve(w, h, u, base, in_region + in_off, u.out);
    const size_t S = h.S, G = h.G, V = h.V;
    u.G = h.G; u.V = h.V; u.S = h.S;
    u.correlate = w.correlate ? 1 : 0; u.overlap = w.overlap;
    u.start_t = w.start_t; u.state_freq = h.state_freq; u.gyr_var = w.gyr_var; u.vel_var = w.vel_var;
    for (int a = 0; a < 3; ++a) { u.gyr_bias[a] = w.gyr_bias[a]; u.vel_bias[a] = w.vel_bias[a]; }
    u.vel_bias_std = w.vel_bias_std; u.gyr_bias_std = w.gyr_bias_std;
    base += h.ws_doubles;
    double* s = stage_in + in_off;
    const size_t padded = (input_doubles(w, h) + 3) / 4 * 4;
    for (size_t k = padded - 4; k < padded; ++k) s[k] = 0.0;  // the padding of this window's slot (filled below up to input_doubles)
    for (size_t k = 0; k < G; ++k) s[k] = w.gyr_t[h.g0 + k];
    s += G;
    for (int a = 0; a < 3; ++a)
      for (size_t k = 0; k < G; ++k) s[a * G + k] = w.gyr[3 * (size_t)(h.g0 + k) + a];
    s += 3 * G;
    for (size_t k = 0; k < V; ++k) s[k] = w.vel_t[h.v0 + k];
    s += V;
    for (int a = 0; a < 3; ++a)
      for (size_t k = 0; k < V; ++k) s[a * V + k] = w.vel[3 * (size_t)(h.v0 + k) + a];
    s += 3 * V;
    for (int k = 0; k < w.n_infer; ++k) s[k] = w.infer_t[k];
    s += w.n_infer;
    for (size_t k = 0; k < S; ++k) s[k] = h.state_t[k];
    in_off += (input_doubles(w, h) + 3) / 4 * 4;
  }
  if (total_in) UHIP(hipMemcpyAsync(in_region, stage_in, sizeof(double) * total_in, hipMemcpyHostToDevice, c.stream));
  UHIP(hipMemcpyAsync(c.d_wins, dw.data(), sizeof(UgpmWin) * n_windows, hipMemcpyHostToDevice, c.stream));
  UHIP(hipMemcpyAsync(c.d_ints, ints.data(), sizeof(int) * ints.size(), hipMemcpyHostToDevice, c.stream));
  UHIP(hipMemsetAsync(c.d_diag, 0, sizeof(double) * 4 * n_windows, c.stream));  // windows no solver touches report zero iterations
  tt1 = tnow();
  // no synchronisation here: the three staging vectors outlive every use of them (they are locals of this call, which ends with a
  // stream synchronisation), and the kernels below are ordered behind the copies on the same stream
  tt2 = tnow();

  // ---- opt.type = LPM windows: their own workspace and three launches (ugpm_lpm_out.hip)
  std::vector<ug::LpmOutWin> lw;
  std::vector<double> lpm_in;
  std::vector<int> lpm_ints_h;
  if (n_lpm > 0) {
    std::vector<LpmHost> lh(n_lpm);
    std::vector<int> widx;
    size_t dbl = 0, in_dbl = 0, nint = 0;
    int max_T = 2;
    for (int i = 0; i < n_windows; ++i) {
      if (!hw[i].is_lpm || hw[i].status != 0) continue;
      LpmHost& L = lh[widx.size()];
      build_lpm_timeline(windows[i], L);
      widx.push_back(i);
      const size_t T = L.tl.size(), G = windows[i].n_gyr, V = windows[i].n_vel, Q = windows[i].n_infer;
      max_T = std::max(max_T, (int)T);
      in_dbl += 4 * G + 4 * V + Q + T;
      dbl += 45 * T + 9 * T + 9 * T + 3 * T + 9 * T + 3 * V + 18 * V + 3 * V + 3 * Q + 8;
      nint += 2 * T + 3 * Q;
    }
    if (in_dbl + dbl > c.lpm_ws_cap) {
      hipFree(c.lpm_ws);
      c.lpm_ws = nullptr;
      c.lpm_ws_cap = 0;
      UHIP(hipMalloc(&c.lpm_ws, sizeof(double) * (in_dbl + dbl)));
      c.lpm_ws_cap = in_dbl + dbl;
    }
    if (nint > c.lpm_ints_cap) {
      hipFree(c.lpm_ints);
      c.lpm_ints = nullptr;
      c.lpm_ints_cap = 0;
      UHIP(hipMalloc(&c.lpm_ints, sizeof(int) * nint));
      c.lpm_ints_cap = nint;
    }
    if (n_lpm > c.lpm_wins_cap) {
      hipFree(c.d_lpm_wins);
      c.d_lpm_wins = nullptr;
      UHIP(hipMalloc(&c.d_lpm_wins, sizeof(ug::LpmOutWin) * n_lpm));
      c.lpm_wins_cap = n_lpm;
    }
    lw.resize(widx.size());
    lpm_in.assign(in_dbl, 0.0);
    lpm_ints_h.assign(nint, 0);
    double* din = c.lpm_ws;           // inputs of all LPM windows, one upload
    double* dsc = c.lpm_ws + in_dbl;  // scratch
    size_t io = 0, so = 0, no = 0;
    for (size_t k = 0; k < widx.size(); ++k) {
      const int i = widx[k];
      const gorio_ugpm_window& w = windows[i];
      const LpmHost& L = lh[k];
      const size_t T = L.tl.size(), G = w.n_gyr, V = w.n_vel, Q = w.n_infer;
      ug::LpmOutWin& u = lw[k];
      std::memset(&u, 0, sizeof(u));
      double* hin = lpm_in.data() + io;
      auto take_in = [&](size_t cnt) { const double* r = din + io; io += cnt; return r; };
      auto take = [&](size_t cnt) { double* r = dsc + so; so += cnt; return r; };
      u.gyr_t = take_in(G); u.gyr = take_in(3 * G); u.vel_t = take_in(V); u.vel = take_in(3 * V); u.infer_t = take_in(Q); u.tl = take_in(T);
      for (size_t q = 0; q < G; ++q) hin[q] = w.gyr_t[q];
      hin += G;
      for (int a = 0; a < 3; ++a)
        for (size_t q = 0; q < G; ++q) hin[a * G + q] = w.gyr[3 * q + a];
      hin += 3 * G;
      for (size_t q = 0; q < V; ++q) hin[q] = w.vel_t[q];
      hin += V;
      for (int a = 0; a < 3; ++a)
        for (size_t q = 0; q < V; ++q) hin[a * V + q] = w.vel[3 * q + a];
      hin += 3 * V;
      for (size_t q = 0; q < Q; ++q) hin[q] = w.infer_t[q];
      hin += Q;
      for (size_t q = 0; q < T; ++q) hin[q] = L.tl[q];
      int* hi = lpm_ints_h.data() + no;
      u.kind = c.lpm_ints + no; u.kidx = c.lpm_ints + no + T; u.qpos = c.lpm_ints + no + 2 * T; u.qorder = c.lpm_ints + no + 2 * T + Q; u.qrot = c.lpm_ints + no + 2 * T + 2 * Q;
      for (size_t q = 0; q < T; ++q) { hi[q] = L.kind[q]; hi[T + q] = L.kidx[q]; }
      for (size_t q = 0; q < Q; ++q) { hi[2 * T + q] = L.qpos[q]; hi[2 * T + Q + q] = L.qorder[q]; hi[2 * T + 2 * Q + q] = L.qrot[q]; }
      no += 2 * T + 3 * Q;
      u.G = (int)G; u.V = (int)V; u.T = (int)T; u.n_infer = (int)Q;
      u.start_index = L.start_index; u.dt_index = L.dt_index;
      u.start_t = w.start_t; u.gyr_var = w.gyr_var; u.vel_var = w.vel_var;
      for (int a = 0; a < 3; ++a) { u.gyr_bias[a] = w.gyr_bias[a]; u.vel_bias[a] = w.vel_bias[a]; }
      u.vel_bias_std = w.vel_bias_std; u.gyr_bias_std = w.gyr_bias_std;
      u.E = take(45 * T); u.B = take(9 * T); u.cov3 = take(9 * T); u.dRdt = take(3 * T); u.dRdbw = take(9 * T);
      u.velr = take(3 * V); u.d_bw = take(18 * V); u.d_dt = take(3 * V); u.dp_shift = take(3 * Q);
      u.out = dw[i].out;
      u.status = c.d_ints + kWinInts * (size_t)i + 17;
    }
    UHIP(hipMemcpyAsync(c.lpm_ws, lpm_in.data(), sizeof(double) * in_dbl, hipMemcpyHostToDevice, c.stream));
    UHIP(hipMemcpyAsync(c.lpm_ints, lpm_ints_h.data(), sizeof(int) * nint, hipMemcpyHostToDevice, c.stream));
    UHIP(hipMemcpyAsync(c.d_lpm_wins, lw.data(), sizeof(ug::LpmOutWin) * lw.size(), hipMemcpyHostToDevice, c.stream));
    const int nl = (int)lw.size();
    if (nl > 0) {
      Stage st(c, 0);
      ug::lpm_out_steps_kernel<<<dim3((max_T + 255) / 256, 5, nl), 256, 0, c.stream>>>(c.d_lpm_wins);
      ug::lpm_out_scan_kernel<<<dim3(5, nl), 64, 0, c.stream>>>(c.d_lpm_wins);
      ug::lpm_out_finish_kernel<<<nl, 256, 0, c.stream>>>(c.d_lpm_wins);
      UHIP(hipGetLastError());
    }
  }

  const int nw = n_windows;
  const int max_G = [&] { int m = 2; for (auto& h : hw) m = std::max(m, h.is_lpm ? 2 : h.G); return m; }();
  const int max_V = [&] { int m = 2; for (auto& h : hw) m = std::max(m, h.is_lpm ? 2 : h.V); return m; }();
  if (max_S > 0) {
    // The windows are independent and nearly every kernel below is a chain of short, latency-bound launches with one (or a few)
    // workgroups per window, so the batch CAN be cut into groups that advance on their own pairs of streams (main + correlation
    // chain).  Measured on the C4 batch (64 windows, profiles/r02/ugpm_groups.txt): 1 group 5.54 ms, 2 groups 5.70 ms, 4 groups
    // 7.36 ms alone, and 10.5 / 10.8 / 11.2 ms per overlapped step -- twice the launches cost more host and queue time than the
    // concurrency returns, so the default stays ONE group; GORIO_UGPM_GROUPS overrides it for experiments.  The arithmetic of a
    // window does not depend on the grouping.
    int n_groups = 1;
    if (const char* e = std::getenv("GORIO_UGPM_GROUPS")) n_groups = std::atoi(e);
    n_groups = std::max(1, std::min(std::min(n_groups, 8), nw));
    while ((int)c.groups.size() < n_groups) {
      Ctx::Group gnew;
      if (c.groups.empty()) {
        gnew.s = c.stream;
        gnew.s2 = c.stream2;
        gnew.ev_jac = c.ev_jac;
        gnew.ev_corr = c.ev_corr;
      } else {
        int lo = 0, hi = 0;
        if (!(hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hipStreamCreateWithPriority(&gnew.s, hipStreamNonBlocking, hi) == hipSuccess))
          UHIP(hipStreamCreateWithFlags(&gnew.s, hipStreamNonBlocking));
        UHIP(hipStreamCreateWithFlags(&gnew.s2, hipStreamNonBlocking));
        UHIP(hipEventCreateWithFlags(&gnew.ev_jac, hipEventDisableTiming));
        UHIP(hipEventCreateWithFlags(&gnew.ev_corr, hipEventDisableTiming));
      }
      UHIP(hipEventCreateWithFlags(&gnew.ev_done, hipEventDisableTiming));
      UHIP(hipEventCreateWithFlags(&gnew.ev_tab, hipEventDisableTiming));
      c.groups.push_back(gnew);
    }
    if (!c.ev_up) UHIP(hipEventCreateWithFlags(&c.ev_up, hipEventDisableTiming));
    UHIP(hipEventRecord(c.ev_up, c.stream));  // inputs, window descriptors and control words are on the device once this fires
    struct Run { int g0, nw; hipStream_t s, s2; hipEvent_t ev_jac, ev_corr, ev_done, ev_tab; bool active; };
    std::vector<Run> runs(n_groups);
    for (int g = 0; g < n_groups; ++g) {
      const int a = (int)((long)nw * g / n_groups), b = (int)((long)nw * (g + 1) / n_groups);
      runs[g] = Run{a, b - a, c.groups[g].s, c.groups[g].s2, c.groups[g].ev_jac, c.groups[g].ev_corr, c.groups[g].ev_done, c.groups[g].ev_tab, true};
      if (g > 0) UHIP(hipStreamWaitEvent(runs[g].s, c.ev_up, 0));
    }
    // J^T J launches: one workgroup per (row slice, tile group, window), see ata_kernel
    auto launch_ata = [&](const Run& r, int which, int decide = 0) {
      hipStream_t sq = which == 2 ? r.s2 : r.s;
      Stage st_ata(c, which == 2 ? 6 : 5, sq);
      const int n = (which == 2 ? 6 : 3) * max_S, T = (n + 15) / 16, ntile = T * (T + 1) / 2;
      const int tpg = which == 2 ? kAtaTilesCorr : kAtaTilesLm;
      const int ng = (ntile + tpg - 1) / tpg;
      const int npad = ((n + 15) / 32) * 32 + 16;
      const int grid = ((r.nw + 7) / 8) * ng * 8;  // 8 windows (one per XCD) x ng tile groups per slice of the grid
      const UgpmWin* dw_ = c.d_wins + r.g0;
      // LDS as small as the staging needs (53 KB at n = 198): the scan matcher's kernels share the CUs with these workgroups
      auto lds = [&](int kc) { return sizeof(double) * 2 * kc * (npad + 1); };
      if (which == 2) {
        if (npad <= 256) ug::ata_kernel<4, 16, kAtaTilesCorr><<<grid, 512, lds(16), sq>>>(dw_, which, r.nw, ng, decide);
        else if (npad <= 512) ug::ata_kernel<8, 16, kAtaTilesCorr><<<grid, 512, lds(16), sq>>>(dw_, which, r.nw, ng, decide);
        else ug::ata_kernel<16, 8, kAtaTilesCorr><<<grid, 512, lds(8), sq>>>(dw_, which, r.nw, ng, decide);
      } else {
        if (npad <= 256) ug::ata_kernel<4, 16, kAtaTilesLm><<<grid, 512, lds(16), sq>>>(dw_, which, r.nw, ng, decide);
        else ug::ata_kernel<8, 16, kAtaTilesLm><<<grid, 512, lds(16), sq>>>(dw_, which, r.nw, ng, decide);  // n = 3S <= 480
      }
    };
    for (const Run& r : runs) {
      const UgpmWin* dw_ = c.d_wins + r.g0;
      {
        Stage st(c, 0, r.s);
        ug::lpm_rot_kernel<<<dim3(r.nw, 5), 320, 0, r.s>>>(dw_);
        ug::lpm_init_kernel<<<r.nw, 320, 0, r.s>>>(dw_);
      }
      {
        Stage st(c, 1, r.s);
        ug::gram_kernel<<<dim3(6, r.nw), 256, 0, r.s>>>(dw_);
        ug::cross_kernel<<<dim3(12, r.nw, (std::max(max_G, max_V) + ug::kCrossRows - 1) / ug::kCrossRows), 256, sizeof(double) * ug::kCrossRows * (((max_S + 31) / 32) * 32 + 4), r.s>>>(dw_);
      }
      // State correlation at the LPM-initialised state.  The reference assembles the Jacobian synchronously (preint.h:887-937) and
      // hands J^T J, its factorisation and the inverse diagonal to a helper thread that runs beside the two ceres::Solve calls and is
      // joined before the first get() (preint.h:939, 1062-1065).  Here the whole chain, Jacobian included, runs on a second stream
      // as soon as the kernel tables exist; the Jacobian reads the LPM-initialised states, so the main stream waits for it (ev_jac)
      // before the first fit writes its solution back (lm_end_kernel), and the inference waits for the end of the chain (ev_corr).
      UHIP(hipEventRecord(r.ev_tab, r.s));
      UHIP(hipStreamWaitEvent(r.s2, r.ev_tab, 0));
      {
        Stage st(c, 2, r.s2);
        ug::corr_jac_kernel<<<dim3(ug::kCorrJacParts, r.nw), 256, 0, r.s2>>>(dw_);
        UHIP(hipEventRecord(r.ev_jac, r.s2));
        launch_ata(r, 2);
        ug::corr_factor_kernel<<<r.nw, 512, 0, r.s2>>>(dw_);
        ug::corr_diag_kernel<<<dim3((6 * max_S + 15) / 16, r.nw), 256, sizeof(double) * 17 * (6 * max_S + 16), r.s2>>>(dw_);
      }
      UHIP(hipEventRecord(r.ev_corr, r.s2));
    }
    std::vector<int> flags(kWinInts * (size_t)nw);
    const bool speculative = g_speculative_rot.load() != 0;
    for (int problem = 0; problem < 2; ++problem) {  // ceres::Solve #1 (rotation) and #2 (velocity), preint.h:943-967
      std::vector<std::unique_ptr<Stage>> st_lm;
      for (Run& r : runs) {
        const UgpmWin* dw_ = c.d_wins + r.g0;
        st_lm.emplace_back(new Stage(c, 3, r.s));
        ug::lm_begin_kernel<<<r.nw, 256, 0, r.s>>>(dw_, problem);
        if (problem == 0) ug::rot_eval_kernel<<<dim3(r.nw, kEvalSplit), 256, 0, r.s>>>(dw_, 2);
        else ug::vel_eval_kernel<<<dim3(r.nw, kEvalSplit), 256, 0, r.s>>>(dw_, 2);
        launch_ata(r, problem);
        r.active = true;
      }
      for (int it = 0; it <= 51; ++it) {
        bool any = false;
        for (Run& r : runs) {
          if (!r.active) continue;
          any = true;
          const UgpmWin* dw_ = c.d_wins + r.g0;
          ug::lm_step_kernel<<<dim3(r.nw, problem == 1 ? kVelBlocks : 1), 512, 0, r.s>>>(dw_);
          if (problem == 0 && speculative) {
            // three launches per iteration: the candidate residual AND the Jacobian at the candidate in one evaluation, the
            // acceptance test inside the J^T J launch (rot_eval_kernel mode 3, ata_kernel decide)
            ug::rot_eval_kernel<<<dim3(r.nw, kEvalSplit), 256, 0, r.s>>>(dw_, 3);
            launch_ata(r, problem, 1);
            continue;
          }
          if (problem == 0) ug::rot_eval_kernel<<<dim3(r.nw, kEvalSplit), 256, 0, r.s>>>(dw_, 0);
          else ug::vel_eval_kernel<<<dim3(r.nw, kEvalSplit), 256, 0, r.s>>>(dw_, 0);
          if (problem == 0) {
            ug::rot_eval_kernel<<<dim3(r.nw, kEvalSplit), 256, 0, r.s>>>(dw_, 1);
            launch_ata(r, problem);
          } else {
            ug::lm_relinearize_linear_kernel<<<dim3(r.nw, (3 * max_S + 63) / 64), 256, 0, r.s>>>(dw_);  // linear problem: J and J^T J stay exact
          }
        }
        if (!any) break;
        if (lmtrace) {  // GORIO_UGPM_LMTRACE: the solver's control words of window 0 after every iteration (a debugging aid: drains the stream)
          double lc[16];
          int li[16];
          UHIP(hipStreamSynchronize(runs[0].s));
          UHIP(hipMemcpy(lc, dw[0].lmc, sizeof(lc), hipMemcpyDeviceToHost));
          UHIP(hipMemcpy(li, dw[0].lmi, sizeof(li), hipMemcpyDeviceToHost));
          std::fprintf(stderr, "[ugpm lm] problem %d it %d: iter %d done %d term %d succ %d cost %.17g cost_new %.17g radius %.6g mcc %.17g step_norm2 %.6g x_norm %.17g initial %.17g\n", problem, it,
                       li[0], li[1], li[5], li[6], lc[0], lc[1], lc[2], lc[8], lc[11], lc[4], lc[7]);
        }
        // When to look at the done flags (a look drains the stream: copy, synchronise, ~30 us of idle GPU).  A finished window's kernels
        // return at once, so iterations enqueued beyond the need cost three empty launches each, far less than a look.  The first batch of a
        // context looks every other iteration from the fourth on (no window of the C2 shape finishes in fewer than four); later batches
        // enqueue as many iterations as the previous batch needed before the first look -- on like data that look is the only one.
        const int budget = c.lm_budget[problem];
        const bool look = budget > 0 ? (it + 1 >= budget && ((it + 1 - budget) & 1) == 0) : (it >= 3 && (it & 1) == 1);
        if (look) {
          for (Run& r : runs)
            if (r.active) UHIP(hipMemcpyAsync(flags.data() + kWinInts * (size_t)r.g0, c.d_ints + kWinInts * (size_t)r.g0, sizeof(int) * kWinInts * (size_t)r.nw, hipMemcpyDeviceToHost, r.s));
          for (Run& r : runs) {
            if (!r.active) continue;
            UHIP(hipStreamSynchronize(r.s));
            bool all = true;
            for (int i = r.g0; i < r.g0 + r.nw; ++i) all = all && (flags[kWinInts * (size_t)i + 1] || flags[kWinInts * (size_t)i + 16] != 0);
            if (all) r.active = false;
          }
          bool every = true;
          for (Run& r : runs) every = every && !r.active;
          if (every) {  // iterations the slowest window really needed: its step count, plus the step kernel that noticed a termination of its own (gradient, iteration cap, radius)
            int need = 1;
            for (int i = 0; i < nw; ++i) {
              const int* f = flags.data() + kWinInts * (size_t)i;
              if (f[16] != 0) continue;
              need = std::max(need, f[0] + (f[5] >= 3 ? 1 : 0));
            }
            c.lm_budget[problem] = std::min(need, it + 1);
          }
        }
      }
      for (size_t g = 0; g < runs.size(); ++g) {
        if (problem == 0) UHIP(hipStreamWaitEvent(runs[g].s, runs[g].ev_jac, 0));  // corr_jac_kernel has read the initial states
        ug::lm_end_kernel<<<runs[g].nw, 256, 0, runs[g].s>>>(c.d_wins + runs[g].g0, problem, c.d_diag + 4 * (size_t)runs[g].g0);
        st_lm[g].reset();  // stage stop event behind the group's last launch of this problem
      }
    }
    for (Run& r : runs) {
      const UgpmWin* dw_ = c.d_wins + r.g0;
      UHIP(hipStreamWaitEvent(r.s, r.ev_corr, 0));  // join of the correlation chain (preint.h:1062-1065)
      {
        Stage st(c, 4, r.s);
        ug::finish_kernel<<<r.nw, 256, 0, r.s>>>(dw_);
        ug::infer_kernel<<<dim3(std::max(1, max_infer), r.nw), 256, sizeof(double) * 17 * (6 * max_S + 16), r.s>>>(dw_);
      }
      UHIP(hipEventRecord(r.ev_done, r.s));
    }
    for (size_t g = 1; g < runs.size(); ++g) UHIP(hipStreamWaitEvent(c.stream, runs[g].ev_done, 0));  // the downloads below follow every group
    UHIP(hipGetLastError());
  }
  tt3 = tnow();
  // ---- results
  std::vector<int> fin(kWinInts * (size_t)nw);
  std::vector<double> dg(4 * (size_t)nw);
  UHIP(hipMemcpyAsync(fin.data(), c.d_ints, sizeof(int) * fin.size(), hipMemcpyDeviceToHost, c.stream));
  UHIP(hipMemcpyAsync(dg.data(), c.d_diag, sizeof(double) * dg.size(), hipMemcpyDeviceToHost, c.stream));
  if (total_out) UHIP(hipMemcpyAsync(reinterpret_cast<double*>(out), out_region, sizeof(double) * total_out, hipMemcpyDeviceToHost, c.stream));
  UHIP(hipStreamSynchronize(c.stream));
  for (size_t q = 0; q < c.ev.size(); ++q) {
    float ms = 0.f;
    if (hipEventSynchronize(c.ev[q].second) == hipSuccess && hipEventElapsedTime(&ms, c.ev[q].first, c.ev[q].second) == hipSuccess) {
      g_stage_s[c.ev_stage[q]] += ms * 1e-3;
      g_stage_n[c.ev_stage[q]] += 1;
    }
    hipEventDestroy(c.ev[q].first);
    hipEventDestroy(c.ev[q].second);
  }
  c.ev.clear();
  c.ev_stage.clear();
  for (int i = 0; i < nw; ++i) {
    if (hw[i].status != 0) {  // rejected on the host: its records were never written
      double* o = reinterpret_cast<double*>(out) + out_offs[i];
      for (size_t k = 0; k < (size_t)std::max(0, windows[i].n_infer) * 83; ++k) o[k] = std::numeric_limits<double>::quiet_NaN();
    }
    int st = hw[i].status != 0 ? hw[i].status : fin[kWinInts * (size_t)i + (hw[i].is_lpm ? 17 : 16)];
    if (hw[i].is_lpm && hw[i].status == 0 && st != 0) {  // the LPM kernels stop mid-way on a data-domain error: no partial records
      double* o = reinterpret_cast<double*>(out) + out_offs[i];
      for (size_t k = 0; k < (size_t)std::max(0, windows[i].n_infer) * 83; ++k) o[k] = std::numeric_limits<double>::quiet_NaN();
    }
    if (st != 0 && hw[i].status == 0 && !first_error) {
      first_error = st;
      first_error_msg = "window " + std::to_string(i) + (st == GORIO_UGPM_ERR_NUMERIC ? ": Cholesky factorisation met a non-positive pivot" : ": LPM Partial: the start_time is not in the data domain");
    }
    if (diag) {
      gorio_ugpm_diag& d = diag[i];
      d.nb_state = hw[i].S; d.nb_gyr = hw[i].G; d.nb_vel = hw[i].V;
      d.iters_rot = (int)dg[4 * (size_t)i + 0]; d.cost_rot = dg[4 * (size_t)i + 1];
      d.iters_vel = (int)dg[4 * (size_t)i + 2]; d.cost_vel = dg[4 * (size_t)i + 3];
      d.status = st; d.state_freq = hw[i].state_freq;
    }
  }
  if (trace) std::fprintf(stderr, "[ugpm trace] prep %.3f ms, upload sync %.3f ms, kernels+polls %.3f ms, results %.3f ms\n", (tt1 - tt0) * 1e3, (tt2 - tt1) * 1e3, (tt3 - tt2) * 1e3, (tnow() - tt3) * 1e3);
  if (first_error) return ufail(first_error, first_error_msg);
  return GORIO_UGPM_OK;
}

// Requests with opt.quantum >= 0 (preint.h:1584-1702) are cut into chunk windows (ugpm_chunks.h); the chunk windows of ALL such
// requests join the other windows of the call in one device batch, and their records are chained on the host afterwards.
int gorio_ugpm_preint_batch(const gorio_ugpm_window* windows, int n_windows, gorio_ugpm_meas* out, gorio_ugpm_diag* diag, int device) {
  if (!windows || n_windows <= 0 || !out) return ufail(GORIO_UGPM_ERR_INVALID, "gorio_ugpm_preint_batch: bad arguments");
  bool any_chunked = false;
  for (int i = 0; i < n_windows; ++i) any_chunked = any_chunked || windows[i].quantum >= 0;
  if (!any_chunked) return preint_batch_flat(windows, n_windows, out, diag, device);
  struct Req { int first = 0, count = 0, status = 0; chunks::Plan plan; size_t out0 = 0; };  // first/count: its windows in the expanded batch
  std::vector<Req> reqs(n_windows);
  std::vector<gorio_ugpm_window> flat;
  std::vector<size_t> flat_out0;
  int first_error = 0;
  std::string first_error_msg;
  size_t n_flat_out = 0, out_off = 0;
  for (int i = 0; i < n_windows; ++i) {
    const gorio_ugpm_window& w = windows[i];
    Req& r = reqs[i];
    r.out0 = out_off;
    out_off += (size_t)std::max(0, w.n_infer);
    r.first = (int)flat.size();
    if (w.quantum < 0) {
      r.count = 1;
      flat.push_back(w);
      flat_out0.push_back(n_flat_out);
      n_flat_out += (size_t)std::max(0, w.n_infer);
      continue;
    }
    std::string err;
    if (!w.gyr_t || !w.gyr || !w.vel_t || !w.vel || !w.infer_t || w.n_infer <= 0) {
      r.status = GORIO_UGPM_ERR_INVALID;
      err = "null pointers or no inference time";
    } else {
      r.status = chunks::plan_chunks(w, r.plan, err);
    }
    if (r.status != 0) {
      if (!first_error) {
        first_error = r.status;
        first_error_msg = "window " + std::to_string(i) + ": " + err;
      }
      continue;
    }
    for (chunks::Chunk& c : r.plan.chunks) {  // one ordinary window per chunk: VelPreintegration(data of the chunk, chunk start, stamps, quantum = -1), get(.., 0, 0)
      gorio_ugpm_window cw = w;
      cw.gyr_t = w.gyr_t + c.g0; cw.gyr = w.gyr + 3 * (size_t)c.g0; cw.n_gyr = c.ng;
      cw.vel_t = w.vel_t + c.v0; cw.vel = w.vel + 3 * (size_t)c.v0; cw.n_vel = c.nv;
      cw.start_t = c.start_t;
      cw.infer_t = c.infer_t.data(); cw.n_infer = (int)c.infer_t.size();
      cw.group_sizes = c.group_sizes.data(); cw.n_groups = (int)c.group_sizes.size();
      cw.quantum = -1;
      cw.vel_bias_std = 0.0; cw.gyr_bias_std = 0.0;
      flat.push_back(cw);
      flat_out0.push_back(n_flat_out);
      n_flat_out += c.infer_t.size();
    }
    r.count = (int)r.plan.chunks.size();
  }
  const gorio_ugpm_meas nan_meas = [] {
    gorio_ugpm_meas m;
    double* p = reinterpret_cast<double*>(&m);
    for (size_t k = 0; k < sizeof(m) / sizeof(double); ++k) p[k] = std::numeric_limits<double>::quiet_NaN();
    return m;
  }();
  std::vector<gorio_ugpm_meas> flat_out(std::max<size_t>(1, n_flat_out), nan_meas);
  std::vector<gorio_ugpm_diag> flat_diag(std::max<size_t>(1, flat.size()));
  int rc = 0;
  if (!flat.empty()) {
    rc = preint_batch_flat(flat.data(), (int)flat.size(), flat_out.data(), flat_diag.data(), device);
    bool per_window = false;  // a per-window failure leaves the other windows valid and is reported through the diagnostics; anything else fails the call
    for (const gorio_ugpm_diag& d : flat_diag) per_window = per_window || d.status != 0;
    if (rc != 0 && !per_window) return rc;
  }
  const std::string flat_msg = rc != 0 ? g_err : std::string();
  for (int i = 0; i < n_windows; ++i) {
    const gorio_ugpm_window& w = windows[i];
    Req& r = reqs[i];
    gorio_ugpm_meas* o = out + r.out0;
    gorio_ugpm_diag d{};
    if (r.status == 0 && w.quantum < 0) {
      for (int k = 0; k < w.n_infer; ++k) o[k] = flat_out[flat_out0[r.first] + k];
      d = flat_diag[r.first];
      r.status = d.status;
    } else if (r.status == 0) {
      std::vector<const gorio_ugpm_meas*> recs;
      for (int q = 0; q < r.count; ++q) {
        const gorio_ugpm_diag& cd = flat_diag[r.first + q];
        if (cd.status != 0 && r.status == 0) r.status = cd.status;
        recs.push_back(flat_out.data() + flat_out0[r.first + q]);
        // diagnostics of a chunked request: sizes and state frequency of its LAST chunk, iterations and costs summed over the chunks
        d.nb_state = cd.nb_state; d.nb_gyr = cd.nb_gyr; d.nb_vel = cd.nb_vel; d.state_freq = cd.state_freq;
        d.iters_rot += cd.iters_rot; d.iters_vel += cd.iters_vel; d.cost_rot += cd.cost_rot; d.cost_vel += cd.cost_vel;
      }
      if (r.status == 0) chunks::chain_chunks(r.plan, recs, w.vel_bias_std, w.gyr_bias_std, o);
    }
    if (r.status != 0) {
      for (int k = 0; k < std::max(0, w.n_infer); ++k) o[k] = nan_meas;
      if (!first_error) {
        first_error = r.status;
        first_error_msg = "window " + std::to_string(i) + (flat_msg.empty() ? std::string(": failed") : ": (expanded batch) " + flat_msg);
      }
    }
    d.status = r.status;
    if (diag) diag[i] = d;
  }
  if (first_error) return ufail(first_error, first_error_msg);
  return 0;
}

/* combinePreints (math_utils.h:689-726) */
int gorio_ugpm_combine_preints(const gorio_ugpm_meas* prev, const gorio_ugpm_meas* cur, gorio_ugpm_meas* out) {
  if (!prev || !cur || !out) return ufail(GORIO_UGPM_ERR_INVALID, "gorio_ugpm_combine_preints: null pointer");
  *out = chunks::combine_preints(*prev, *cur);
  return 0;
}

}  // extern "C"
