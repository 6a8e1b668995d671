// Micro-benchmark of the workgroup Cholesky / triangular solves of go-rio_amd/csrc/ugpm_kernels.hip (development tool, not shipped).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -DGORIO_CHOL_TIMING -I include -o /tmp/chol_bench tools/chol_bench.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../go-rio_amd/csrc/ugpm_kernels.hip"

using namespace gorio;
using namespace gorio::ug;

__global__ __launch_bounds__(512) void chol_bench_kernel(double* mats, int n, int reps, int with_rhs, double* xout, double* rhs_all) {
  __shared__ CholLds chol;
  __shared__ int sflag;
  __shared__ double xs[480];
  double* orig = mats + (size_t)blockIdx.x * 2 * n * n;
  double* work = orig + (size_t)n * n;
  double* rhs = rhs_all + (size_t)blockIdx.x * n;
  for (int rep = 0; rep < reps; ++rep) {
    for (int q = threadIdx.x; q < n * n; q += blockDim.x) work[q] = orig[q];
    for (int j = threadIdx.x; j < n; j += blockDim.x) rhs[j] = 1.0 + 0.01 * j;
    __syncthreads();
    const long long t0 = (long long)__builtin_readcyclecounter();
    const bool ok = block_cholesky(work, n, n, chol, &sflag, with_rhs ? rhs : nullptr);
    const long long t1 = (long long)__builtin_readcyclecounter();
    for (int j = threadIdx.x; j < n; j += blockDim.x) xs[j] = rhs[j];
    __syncthreads();
    if (ok && with_rhs) block_backward(work, n, n, xs, chol);
    const long long t2 = (long long)__builtin_readcyclecounter();
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      g_chol_t[6] += t1 - t0;
      g_chol_t[7] += t2 - t1;
    }
    __syncthreads();
  }
  for (int j = threadIdx.x; j < n; j += blockDim.x) xout[(size_t)blockIdx.x * n + j] = xs[j];
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 198, nmat = argc > 2 ? atoi(argv[2]) : 64, reps = argc > 3 ? atoi(argv[3]) : 10;
  std::vector<double> h((size_t)nmat * 2 * n * n);
  srand(1);
  std::vector<double> B((size_t)n * n);
  for (int b = 0; b < nmat; ++b) {
    for (auto& v : B) v = rand() / (double)RAND_MAX - 0.5;
    double* A = h.data() + (size_t)b * 2 * n * n;
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) {
        double s = i == j ? 1.0 : 0.0;
        for (int k = 0; k < n; ++k) s += B[(size_t)i * n + k] * B[(size_t)j * n + k];
        A[(size_t)i * n + j] = s;
      }
  }
  double *d, *x, *rh;
  hipMalloc(&d, h.size() * 8);
  hipMalloc(&x, (size_t)nmat * n * 8);
  hipMalloc(&rh, (size_t)nmat * n * 8);
  hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  chol_bench_kernel<<<nmat, 512>>>(d, n, 1, 1, x, rh);
  hipDeviceSynchronize();
  long long z[12] = {0};
  hipMemcpyToSymbol(HIP_SYMBOL(g_chol_t), z, sizeof z);
  hipEventRecord(e0);
  chol_bench_kernel<<<nmat, 512>>>(d, n, reps, 1, x, rh);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  long long t[12];
  hipMemcpyFromSymbol(t, HIP_SYMBOL(g_chol_t), sizeof t);
  // check: residual of A x = b for matrix 0
  std::vector<double> xs(n);
  hipMemcpy(xs.data(), x, n * 8, hipMemcpyDeviceToHost);
  double rmax = 0;
  for (int i = 0; i < n; ++i) {
    double s = 0;
    for (int j = 0; j < n; ++j) s += h[(size_t)i * n + j] * xs[j];
    rmax = fmax(rmax, fabs(s - (1.0 + 0.01 * i)));
  }
  printf("n=%d mats=%d reps=%d: %.1f us per factor+solve (kernel %.3f ms), residual %.3e\n", n, nmat, reps, ms * 1e3 / reps, ms, rmax);
  const char* names[12] = {"diag rest", "wait after diag", "panel solve", "wait after solve", "trailing", "wait after trailing", "cholesky total", "backward total", "diag: load block", "diag: factor + stores", "diag: load own row", "-"};
  for (int k = 0; k < 11; ++k) printf("  %-28s %10.0f cycles/rep\n", names[k], (double)t[k] / reps);
  return rmax < 1e-6 ? 0 : 1;
}
