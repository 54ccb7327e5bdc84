// Store rate of K-like output units: one-shot workgroups that each write R KB contiguous (a tile row of the elastic K
// kernel is a 31 KB run), dispatched in memory order; W waves per workgroup, each wave its own run.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef double d2v __attribute__((ext_vector_type(2)));
template <bool NT>
__global__ void k_runs(double *out, int64_t n2, int run2) {  // run2: 16-B words per wave
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, W = blockDim.x >> 6;
  const int64_t base = ((int64_t)blockIdx.x * W + w) * run2;
  d2v *o = reinterpret_cast<d2v *>(out) + base;
  const d2v x = {1.0, 2.0};
  int64_t end = n2 - base;
  if (end > run2) end = run2;
  for (int t = lane; t < end; t += 64) { if (NT) __builtin_nontemporal_store(x, &o[t]); else o[t] = x; }
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
      (void)hipEventRecord(e0, 0); fn(); (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    printf("%-64s %.3f ms  %.2f TB/s\n", name, best, n * 8.0 / best / 1e9);
  };
  char nm[128];
  for (int kb : {4, 8, 16, 31, 62, 124}) {
    for (int W : {1, 4}) {
      const int run2 = kb * 1024 / 16;
      const int64_t nwg = (n / 2 + (int64_t)run2 * W - 1) / ((int64_t)run2 * W);
      snprintf(nm, sizeof nm, "one-shot, %d wave(s)/wg, %d KB run per wave, plain (%lld wg)", W, kb, (long long)nwg);
      timeit(nm, [&] { k_runs<false><<<(unsigned)nwg, 64 * W>>>(out, n / 2, run2); });
      snprintf(nm, sizeof nm, "one-shot, %d wave(s)/wg, %d KB run per wave, nt", W, kb);
      timeit(nm, [&] { k_runs<true><<<(unsigned)nwg, 64 * W>>>(out, n / 2, run2); });
    }
  }
  return 0;
}
