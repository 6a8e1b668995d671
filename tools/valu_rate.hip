// tools/valu_rate.hip -- issue rate of the VALU instructions the selection k-NN is made of (development micro-benchmark).
// Every lane runs 8 independent dependency chains of one instruction kind; cycles per wave-instruction per SIMD are reported for
// 1, 2, 4 and 8 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 -o tools/valu_rate tools/valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int OP>
__global__ void rate_kernel(float* out, int iters, float seed) {
  float a[8], b = seed * 1.0001f, c = seed * 0.5f;
  for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * 0.001f + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        if (OP == 1) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if (OP == 2) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if (OP == 3) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        if (OP == 4) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if (OP == 5) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));
        if (OP == 6) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        if (OP == 7) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(double*)&a[i & 6]) : "v"(*(double*)&a[(i + 2) & 6]));
        if (OP == 8) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if (OP == 9) asm volatile("v_max_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if (OP == 10) asm volatile("v_med3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        if (OP == 11) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        if (OP == 12) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
        if (OP == 13) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(double*)&a[i & 6]) : "v"(*(double*)&a[(i + 2) & 6]));
        if (OP == 14) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(double*)&a[i & 6]) : "v"(*(double*)&a[(i + 2) & 6]));
        if (OP == 15) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(*(double*)&a[i & 6]) : "v"(*(double*)&a[(i + 2) & 6]));
        if (OP == 16) asm volatile("v_add_f64 %0, %0, %1" : "+v"(*(double*)&a[i & 6]) : "v"(*(double*)&a[(i + 2) & 6]));
        if (OP == 17) asm volatile("v_cmp_lt_u64 vcc, %0, %1" : : "v"(*(double*)&a[i & 6]), "v"(*(double*)&a[(i + 2) & 6]) : "vcc");
        if (OP == 18) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if (OP == 19) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if (OP == 20) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if (OP == 21) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "s"(1ull));
        if (OP == 22) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
void run(const char* name) {
  float* d;
  hipMalloc(&d, sizeof(float) * 256 * 8 * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 2000;
  std::printf("%-14s", name);
  for (int wps : {1, 2, 4, 8}) {
    const int blocks = 256 * wps;  // 256-thread blocks = 4 waves = one per SIMD of a CU
    rate_kernel<OP><<<blocks, 256>>>(d, 10, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    rate_kernel<OP><<<blocks, 256>>>(d, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double inst_per_simd = (double)iters * 64 * wps;  // wave-instructions issued on one SIMD
    std::printf("  %d w/SIMD: %.2f cyc/inst @2.4GHz", wps, ms * 1e-3 * 2.4e9 / inst_per_simd);
  }
  std::printf("\n");
  hipFree(d);
}

int main() {
  run<0>("v_fma_f32");
  run<4>("v_add_f32");
  run<1>("v_min_f32");
  run<2>("v_max_f32");
  run<3>("v_med3_f32");
  run<6>("v_min3_f32");
  run<5>("v_cndmask_b32");
  run<7>("v_pk_mul_f32");
  run<13>("v_pk_add_f32");
  run<14>("v_pk_fma_f32");
  run<18>("v_sub_f32");
  run<19>("v_mul_f32");
  run<20>("v_and_b32");
  run<8>("v_min_u32");
  run<9>("v_max_u32");
  run<10>("v_med3_u32");
  run<11>("v_min3_u32");
  run<12>("cmp+cndmask");
  run<21>("v_cndmask sgpr");
  run<22>("v_cmp_lt_u32");
  run<17>("v_cmp_lt_u64");
  run<15>("v_fma_f64");
  run<16>("v_add_f64");
  return 0;
}
