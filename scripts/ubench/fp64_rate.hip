// FP64 FMA issue-rate microbenchmark: how many waves per SIMD does v_fma_f64 need to reach peak?
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NACC>
__global__ void k(double *out, int iters, double a, double b) {
  double acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; i++) acc[i] = threadIdx.x * 1e-3 + i;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = fma(acc[i], a, b);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < NACC; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int wg_per_cu, int threads) {
  int ncu = 256, iters = 4000;
  int blocks = ncu * wg_per_cu;
  double *out; hipMalloc(&out, sizeof(double) * blocks * threads);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<NACC><<<blocks, threads>>>(out, 10, 1.0000001, 1e-9);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<NACC><<<blocks, threads>>>(out, iters, 1.0000001, 1e-9);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = 2.0 * NACC * (double)iters * blocks * threads;
  printf("NACC=%2d  blocks/CU=%d threads=%4d (waves/SIMD=%.2f): %.3f ms  %.1f TFLOP/s\n", NACC, wg_per_cu, threads,
         wg_per_cu * threads / 256.0, ms, flops / ms / 1e9);
  hipFree(out);
}
int main() {
  run<16>(1, 256); run<16>(2, 256); run<16>(4, 256); run<16>(8, 256);
  run<4>(1, 256); run<4>(2, 256); run<4>(4, 256);
  run<64>(1, 256); run<64>(2, 256);
  return 0;
}
