// What is the store roof of this part?  4.25 GB (the elastic K kernel's output) written by different patterns:
//   hipMemsetAsync (the runtime's own fill kernel), grid-stride 16-B stores (plain / non-temporal), one contiguous
//   chunk per workgroup (each wave a contiguous sub-chunk), 8-B stores, different grid sizes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef double d2v __attribute__((ext_vector_type(2)));
template <bool NT>
__global__ __launch_bounds__(256) void k_stride(double *out, int64_t n2) {
  d2v *o = reinterpret_cast<d2v *>(out);
  const d2v x = {1.0, 2.0};
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < n2; t += (int64_t)gridDim.x * 256) {
    if (NT) __builtin_nontemporal_store(x, &o[t]); else o[t] = x;
  }
}
// workgroup b owns [b*chunk, (b+1)*chunk); its 4 waves walk it together, 4 KB per workgroup step
template <bool NT>
__global__ __launch_bounds__(256) void k_chunk(double *out, int64_t n2, int64_t chunk2) {
  d2v *o = reinterpret_cast<d2v *>(out) + (int64_t)blockIdx.x * chunk2;
  const d2v x = {1.0, 2.0};
  int64_t end = n2 - (int64_t)blockIdx.x * chunk2;
  if (end > chunk2) end = chunk2;
  for (int64_t t = threadIdx.x; t < end; t += 256) {
    if (NT) __builtin_nontemporal_store(x, &o[t]); else o[t] = x;
  }
}
// each WAVE owns a contiguous chunk (the pattern of the autonomous-wave kernels)
template <bool NT>
__global__ __launch_bounds__(64) void k_wavechunk(double *out, int64_t n2, int64_t chunk2) {
  d2v *o = reinterpret_cast<d2v *>(out) + (int64_t)blockIdx.x * chunk2;
  const d2v x = {1.0, 2.0};
  int64_t end = n2 - (int64_t)blockIdx.x * chunk2;
  if (end > chunk2) end = chunk2;
  for (int64_t t = threadIdx.x; t < end; t += 64) {
    if (NT) __builtin_nontemporal_store(x, &o[t]); else o[t] = x;
  }
}
__global__ __launch_bounds__(256) void k_stride8(double *out, int64_t n) {
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (int64_t)gridDim.x * 256) __builtin_nontemporal_store(1.0, &out[t]);
}
int main() {
  const int64_t n = (int64_t)(4.25e9 / 8) & ~1023LL;
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
    printf("%-56s %.3f ms  %.2f TB/s\n", name, best, n * 8.0 / best / 1e9);
  };
  timeit("hipMemsetAsync", [&] { (void)hipMemsetAsync(out, 0, n * 8, 0); });
  for (int g : {512, 1024, 2048, 4096, 8192, 16384}) {
    char nm[96];
    snprintf(nm, sizeof nm, "grid-stride 16 B nt, %d workgroups", g);
    timeit(nm, [&] { k_stride<true><<<g, 256>>>(out, n / 2); });
  }
  timeit("grid-stride 16 B plain, 2048 workgroups", [&] { k_stride<false><<<2048, 256>>>(out, n / 2); });
  timeit("grid-stride 8 B nt, 2048 workgroups", [&] { k_stride8<<<2048, 256>>>(out, n); });
  for (int g : {1024, 2048, 8192, 32768, 131072}) {
    char nm[96];
    const int64_t chunk2 = (n / 2 + g - 1) / g;
    snprintf(nm, sizeof nm, "chunk per workgroup nt, %d workgroups (%lld KB each)", g, (long long)(chunk2 * 16 / 1024));
    timeit(nm, [&] { k_chunk<true><<<g, 256>>>(out, n / 2, chunk2); });
  }
  for (int g : {2048, 4096, 16384, 65536}) {
    char nm[96];
    const int64_t chunk2 = (n / 2 + g - 1) / g;
    snprintf(nm, sizeof nm, "chunk per WAVE nt, %d waves (%lld KB each)", g, (long long)(chunk2 * 16 / 1024));
    timeit(nm, [&] { k_wavechunk<true><<<g, 64>>>(out, n / 2, chunk2); });
    snprintf(nm, sizeof nm, "chunk per WAVE plain, %d waves", g);
    timeit(nm, [&] { k_wavechunk<false><<<g, 64>>>(out, n / 2, chunk2); });
  }
  return 0;
}
