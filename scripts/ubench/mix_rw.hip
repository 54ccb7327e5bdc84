// Mixed read/write microbenchmark: can a 4.8 GB streaming read and a 4.25 GB streaming write overlap on MI355X's HBM?
// (the headline step is a store-bound K kernel followed by a read-bound RHS kernel, each at 0.57-0.66 of 8 TB/s)
//   A  fill alone, B  read alone, C  A then B on one stream, D  A and B on two streams,
//   E  one kernel whose workgroups alternate between the two roles (R of every 8 workgroups read)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef double d2v __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k_fill(double *out, int64_t n2, double v) {
  d2v *o = reinterpret_cast<d2v *>(out);
  const d2v x = {v, v};
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < n2; t += (int64_t)gridDim.x * 256) __builtin_nontemporal_store(x, &o[t]);
}
__global__ __launch_bounds__(256) void k_read(const double *in, int64_t n2, double *sink) {
  const d2v *s = reinterpret_cast<const d2v *>(in);
  d2v acc = {0, 0};
  int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t st = (int64_t)gridDim.x * 256;
  for (; t + 3 * st < n2; t += 4 * st) {
    const d2v a = __builtin_nontemporal_load(&s[t]), b = __builtin_nontemporal_load(&s[t + st]);
    const d2v c = __builtin_nontemporal_load(&s[t + 2 * st]), d = __builtin_nontemporal_load(&s[t + 3 * st]);
    acc += a + b + c + d;
  }
  for (; t < n2; t += st) acc += __builtin_nontemporal_load(&s[t]);
  if (acc.x == 123.456) sink[0] = acc.y;
}
// role by workgroup: (blockIdx % 8) < R reads, the others fill; each role strides over its own array
__global__ __launch_bounds__(256) void k_mix(const double *in, int64_t nr2, double *out, int64_t nw2, double *sink, int R) {
  const int g = blockIdx.x & 7, grp = blockIdx.x >> 3;
  const int64_t ngrp = gridDim.x >> 3;
  if (g < R) {
    const d2v *s = reinterpret_cast<const d2v *>(in);
    d2v acc = {0, 0};
    const int64_t st = ngrp * R * 256;
    for (int64_t t = ((int64_t)grp * R + g) * 256 + threadIdx.x; t < nr2; t += st) acc += __builtin_nontemporal_load(&s[t]);
    if (acc.x == 123.456) sink[0] = acc.y;
  } else {
    d2v *o = reinterpret_cast<d2v *>(out);
    const d2v x = {1.0, 2.0};
    const int W = 8 - R;
    const int64_t st = ngrp * W * 256;
    for (int64_t t = ((int64_t)grp * W + (g - R)) * 256 + threadIdx.x; t < nw2; t += st) __builtin_nontemporal_store(x, &o[t]);
  }
}
int main() {
  const int64_t nr = (int64_t)(4.83e9 / 8), nw = (int64_t)(4.25e9 / 8);
  double *in, *out, *sink;
  hipMalloc(&in, nr * 8); hipMalloc(&out, nw * 8); hipMalloc(&sink, 64);
  hipMemset(in, 0, nr * 8); hipMemset(out, 0, nw * 8);
  hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
  hipEvent_t e0, e1, e2; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
  const int G = 256 * 8;
  auto timeit = [&](const char *name, auto fn, double bytes) {
    fn(); hipDeviceSynchronize();
    float best = 1e9;
    for (int r = 0; r < 5; r++) {
      hipDeviceSynchronize();
      hipEventRecord(e0, s1);
      fn();
      hipEventRecord(e1, s1);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    printf("%-44s %.3f ms  %.2f TB/s\n", name, best, bytes / best / 1e9);
  };
  timeit("A fill 4.25 GB", [&] { k_fill<<<G, 256, 0, s1>>>(out, nw / 2, 1.0); }, nw * 8.0);
  timeit("B read 4.83 GB", [&] { k_read<<<G, 256, 0, s1>>>(in, nr / 2, sink); }, nr * 8.0);
  timeit("C fill then read, one stream", [&] { k_fill<<<G, 256, 0, s1>>>(out, nw / 2, 1.0); k_read<<<G, 256, 0, s1>>>(in, nr / 2, sink); }, (nw + nr) * 8.0);
  timeit("D fill and read, two streams", [&] {
    hipEventRecord(e2, s1); hipStreamWaitEvent(s2, e2, 0);
    k_fill<<<G / 2, 256, 0, s1>>>(out, nw / 2, 1.0);
    k_read<<<G / 2, 256, 0, s2>>>(in, nr / 2, sink);
    hipEventRecord(e2, s2); hipStreamWaitEvent(s1, e2, 0);
  }, (nw + nr) * 8.0);
  for (int R = 2; R <= 6; R++) {
    char nm[64]; snprintf(nm, sizeof nm, "E one kernel, %d of 8 workgroups read", R);
    timeit(nm, [&] { k_mix<<<G, 256, 0, s1>>>(in, nr / 2, out, nw / 2, sink, R); }, (nw + nr) * 8.0);
  }
  return 0;
}
