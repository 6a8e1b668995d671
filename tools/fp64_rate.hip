// Issue-rate probe for fp64 on gfx950 (development tool): cycles per v_mfma_f64_16x16x4_f64 and per v_fma_f64, for 1..4 waves per
// SIMD.   hipcc --offload-arch=gfx950 -O3 -o tools/fp64_rate tools/fp64_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));

__global__ void mfma_probe(double* out, long long* cyc, int iters) {
  f64x4 acc[8];
  for (int t = 0; t < 8; ++t) acc[t] = f64x4{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  __syncthreads();
  const long long t0 = (long long)__builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
  }
  const long long t1 = (long long)__builtin_readcyclecounter();
  double s = 0;
  for (int t = 0; t < 8; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { cyc[2 * (threadIdx.x >> 6)] = t0; cyc[2 * (threadIdx.x >> 6) + 1] = t1; }
}

__global__ void fma_probe(double* out, long long* cyc, int iters) {
  double acc[16];
  for (int t = 0; t < 16; ++t) acc[t] = t;
  double a = 1.0 + threadIdx.x * 1e-9, b = threadIdx.x * 1e-6;
  __syncthreads();
  const long long t0 = (long long)__builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[t] = __builtin_fma(acc[t], a, b);
  }
  const long long t1 = (long long)__builtin_readcyclecounter();
  double s = 0;
  for (int t = 0; t < 16; ++t) s += acc[t];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { cyc[2 * (threadIdx.x >> 6)] = t0; cyc[2 * (threadIdx.x >> 6) + 1] = t1; }
}

__global__ void fma_chain_probe(double* out, long long* cyc, int iters) {
  double acc = threadIdx.x;
  double a = 1.0 + threadIdx.x * 1e-9, b = threadIdx.x * 1e-6;
  const long long t0 = (long long)__builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int t = 0; t < 16; ++t) acc = __builtin_fma(acc, a, b);
  }
  const long long t1 = (long long)__builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { cyc[2 * (threadIdx.x >> 6)] = t0; cyc[2 * (threadIdx.x >> 6) + 1] = t1; }
}

int main() {
  double* out;
  long long* cyc;
  hipMalloc(&out, 8 << 20);
  hipMalloc(&cyc, 1024);
  const int iters = 2000;
  auto span = [&](int threads) {
    long long h[64];
    hipMemcpy(h, cyc, 16 * (threads / 64), hipMemcpyDeviceToHost);
    long long lo = h[0], hi = h[1];
    for (int w = 0; w < threads / 64; ++w) { if (h[2 * w] < lo) lo = h[2 * w]; if (h[2 * w + 1] > hi) hi = h[2 * w + 1]; }
    return hi - lo;
  };
  for (int threads : {64, 256, 512, 1024}) {
    long long c;
    for (int grid : {1, 1024}) {
      mfma_probe<<<grid, threads>>>(out, cyc, iters);
      c = span(threads);
      printf("mfma_f64_16x16x4: block %4d grid %4d: %.1f cycles per MFMA per wave (%.1f per SIMD-MFMA)\n", threads, grid, (double)c / (iters * 8),
             (double)c / (iters * 8) / ((threads + 255) / 256));
      fma_probe<<<grid, threads>>>(out, cyc, iters);
      c = span(threads);
      printf("v_fma_f64 (16 chains): block %4d grid %4d: %.1f cycles per FMA per wave\n", threads, grid, (double)c / (iters * 16));
    }
  }
  long long c;
  fma_chain_probe<<<1, 64>>>(out, cyc, iters);
  c = span(64);
  printf("v_fma_f64 dependent chain: %.1f cycles per FMA\n", (double)c / (iters * 16));
  return 0;
}
