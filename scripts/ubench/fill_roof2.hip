// Follow-up of fill_roof: which store pattern reaches the runtime fill's 6.5 TB/s?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef double d2v __attribute__((ext_vector_type(2)));
template <bool NT> __device__ inline void st(d2v *p, d2v x) { if (NT) __builtin_nontemporal_store(x, p); else *p = x; }
// grid-stride, U stores per thread per iteration (independent addresses one grid apart)
template <bool NT, int U>
__global__ __launch_bounds__(256) void k_stride_u(double *out, int64_t n2) {
  d2v *o = reinterpret_cast<d2v *>(out);
  const d2v x = {1.0, 2.0};
  const int64_t G = (int64_t)gridDim.x * 256;
  int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (; t + (U - 1) * G < n2; t += U * G) {
#pragma unroll
    for (int u = 0; u < U; u++) st<NT>(&o[t + u * G], x);
  }
  for (; t < n2; t += G) st<NT>(&o[t], x);
}
// each thread writes U consecutive 16-B words (a wave covers U KB contiguous per iteration, lanes U*16 B apart)
template <bool NT, int U>
__global__ __launch_bounds__(256) void k_thread_contig(double *out, int64_t n2) {
  d2v *o = reinterpret_cast<d2v *>(out);
  const d2v x = {1.0, 2.0};
  const int64_t G = (int64_t)gridDim.x * 256 * U;
  for (int64_t t = ((int64_t)blockIdx.x * 256 + threadIdx.x) * U; t + U <= n2; t += G) {
#pragma unroll
    for (int u = 0; u < U; u++) st<NT>(&o[t + u], x);
  }
}
// one workgroup = one contiguous block of B KB, written once (no loop): many small workgroups
template <bool NT, int U>
__global__ __launch_bounds__(256) void k_block(double *out, int64_t n2) {
  d2v *o = reinterpret_cast<d2v *>(out) + (int64_t)blockIdx.x * 256 * U;
  const d2v x = {1.0, 2.0};
#pragma unroll
  for (int u = 0; u < U; u++)
    if ((int64_t)blockIdx.x * 256 * U + u * 256 + threadIdx.x < n2) st<NT>(&o[u * 256 + threadIdx.x], x);
}
int main() {
  const int64_t n = (int64_t)(4.25e9 / 8) & ~((1LL << 16) - 1);
  double *out;
  if (hipMalloc(&out, n * 8) != hipSuccess) return 1;
  (void)hipMemset(out, 0, n * 8);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto timeit = [&](const char *name, auto fn) {
    fn(); (void)hipDeviceSynchronize();
    float best = 1e9;
    for (int r = 0; r < 5; r++) {
      (void)hipEventRecord(e0, 0);
      fn();
      (void)hipEventRecord(e1, 0);
      (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    printf("%-60s %.3f ms  %.2f TB/s\n", name, best, n * 8.0 / best / 1e9);
  };
  timeit("hipMemsetAsync", [&] { (void)hipMemsetAsync(out, 0, n * 8, 0); });
  char nm[96];
  for (int g : {2048, 16384, 65536}) {
    snprintf(nm, sizeof nm, "grid-stride plain x1, %d wg", g); timeit(nm, [&] { k_stride_u<false, 1><<<g, 256>>>(out, n / 2); });
    snprintf(nm, sizeof nm, "grid-stride plain x4, %d wg", g); timeit(nm, [&] { k_stride_u<false, 4><<<g, 256>>>(out, n / 2); });
    snprintf(nm, sizeof nm, "grid-stride nt x4, %d wg", g); timeit(nm, [&] { k_stride_u<true, 4><<<g, 256>>>(out, n / 2); });
    snprintf(nm, sizeof nm, "thread-contiguous 64 B plain, %d wg", g); timeit(nm, [&] { k_thread_contig<false, 4><<<g, 256>>>(out, n / 2); });
  }
  {
    const int64_t nb4 = (n / 2 + 256 * 4 - 1) / (256 * 4), nb1 = (n / 2 + 255) / 256, nb16 = (n / 2 + 256 * 16 - 1) / (256 * 16);
    snprintf(nm, sizeof nm, "one 4 KB block per workgroup plain (%lld wg)", (long long)nb1); timeit(nm, [&] { k_block<false, 1><<<(unsigned)nb1, 256>>>(out, n / 2); });
    snprintf(nm, sizeof nm, "one 16 KB block per workgroup plain (%lld wg)", (long long)nb4); timeit(nm, [&] { k_block<false, 4><<<(unsigned)nb4, 256>>>(out, n / 2); });
    snprintf(nm, sizeof nm, "one 16 KB block per workgroup nt (%lld wg)", (long long)nb4); timeit(nm, [&] { k_block<true, 4><<<(unsigned)nb4, 256>>>(out, n / 2); });
    snprintf(nm, sizeof nm, "one 64 KB block per workgroup plain (%lld wg)", (long long)nb16); timeit(nm, [&] { k_block<false, 16><<<(unsigned)nb16, 256>>>(out, n / 2); });
  }
  return 0;
}
