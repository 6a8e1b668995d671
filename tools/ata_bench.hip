// Micro-benchmark of ata_kernel (development tool).  Phase cycles are those of thread 0 of workgroup 0.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -DGORIO_CHOL_TIMING -I include -o tools/ata_bench tools/ata_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../go-rio_amd/csrc/ugpm_kernels.hip"
using namespace gorio;
using namespace gorio::ug;

int main(int argc, char** argv) {
  const int S = argc > 1 ? atoi(argv[1]) : 66, G = argc > 2 ? atoi(argv[2]) : 260, nw = argc > 3 ? atoi(argv[3]) : 64, reps = 10;
  const int n = 3 * S, m = 3 * S + 3 * G;
  const int T = (n + 15) / 16, ntile = T * (T + 1) / 2, ng = (ntile + kAtaTilesLm - 1) / kAtaTilesLm;
  std::vector<UgpmWin> hw(nw);
  std::vector<double> J((size_t)m * n), rr(m);
  srand(3);
  for (auto& v : J) v = rand() / (double)RAND_MAX - 0.5;
  for (auto& v : rr) v = rand() / (double)RAND_MAX - 0.5;
  int* ints;
  hipMalloc(&ints, sizeof(int) * kWinInts * nw);
  hipMemset(ints, 0, sizeof(int) * kWinInts * nw);
  std::vector<int> hi(kWinInts * (size_t)nw, 0);
  for (int i = 0; i < nw; ++i) hi[kWinInts * (size_t)i + 3] = 1;  // need_J
  hipMemcpy(ints, hi.data(), sizeof(int) * hi.size(), hipMemcpyHostToDevice);
  double *dJ, *dr;
  hipMalloc(&dJ, J.size() * 8 * nw);
  hipMalloc(&dr, rr.size() * 8);
  for (int i = 0; i < nw; ++i) hipMemcpy(dJ + (size_t)i * J.size(), J.data(), J.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(dr, rr.data(), rr.size() * 8, hipMemcpyHostToDevice);
  for (int i = 0; i < nw; ++i) {
    UgpmWin& u = hw[i];
    memset(&u, 0, sizeof u);
    u.S = S; u.G = G; u.V = G;
    u.lmi = ints + kWinInts * (size_t)i; u.status = u.lmi + 16;
    u.Jrot = dJ + (size_t)i * J.size(); u.res = dr;
    hipMalloc(&u.JtJ, (size_t)n * n * 8);
    hipMalloc(&u.lmv, (size_t)8 * n * 8);
  }
  UgpmWin* dw;
  hipMalloc(&dw, sizeof(UgpmWin) * nw);
  hipMemcpy(dw, hw.data(), sizeof(UgpmWin) * nw, hipMemcpyHostToDevice);
  const int npad = ((n + 15) / 32) * 32 + 16;
  const int units = nw * ng, grid = ((units + 7) / 8) * 8;
  const size_t lds = sizeof(double) * 2 * 16 * (npad + 1);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&ata_kernel<4, 16, kAtaTilesLm>), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&ata_kernel<8, 16, kAtaTilesLm>), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
  auto launch = [&] {
    if (npad <= 256) ata_kernel<4, 16, kAtaTilesLm><<<grid, 512, lds>>>(dw, 0, nw, ng);
    else ata_kernel<8, 16, kAtaTilesLm><<<grid, 512, lds>>>(dw, 0, nw, ng);
  };
  launch();
  hipDeviceSynchronize();
  long long z[12] = {0};
  hipMemcpyToSymbol(HIP_SYMBOL(g_chol_t), z, sizeof z);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) launch();
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  long long t[12];
  hipMemcpyFromSymbol(t, HIP_SYMBOL(g_chol_t), sizeof t);
  // check against the host
  std::vector<double> C((size_t)n * n), gv(n);
  hipMemcpy(C.data(), hw[nw - 1].JtJ, C.size() * 8, hipMemcpyDeviceToHost);
  hipMemcpy(gv.data(), hw[nw - 1].lmv, n * 8, hipMemcpyDeviceToHost);
  double emax = 0, gmax = 0;
  for (int i = 0; i < n; i += 7)
    for (int j = 0; j < n; j += 5) {
      double sref = 0;
      for (int k = 0; k < m; ++k) sref += J[(size_t)k * n + i] * J[(size_t)k * n + j];
      emax = fmax(emax, fabs(sref - C[(size_t)i * n + j]));
    }
  for (int j = 0; j < n; ++j) {
    double sref = 0;
    for (int k = 0; k < m; ++k) sref += J[(size_t)k * n + j] * rr[k];
    gmax = fmax(gmax, fabs(sref - gv[j]));
  }
  printf("S=%d m=%d n=%d windows=%d grid=%d: %.1f us per launch; max |C - ref| %.2e, max |g - ref| %.2e\n", S, m, n, nw, grid, ms * 1e3 / reps, emax, gmax);
  const char* names[8] = {"prologue", "mfma (sum over chunks)", "g", "stash", "barrier", "partial stores", "fence+barrier", "reduce (if last)"};
  for (int k = 0; k < 8; ++k) printf("  %-26s %10.0f cycles/launch\n", names[k], (double)t[k] / reps);
  return emax < 1e-9 ? 0 : 1;
}
