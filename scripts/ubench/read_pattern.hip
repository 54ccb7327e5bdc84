// Read-pattern microbenchmark: the S3 RHS reads doubles [112,400) of every 400-double element record.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
// each 32-lane group reads one element's 144 double2 (q + 32k), `ahead` elements in flight per group
template <int AHEAD>
__global__ __launch_bounds__(256) void rd(const double *gp, int64_t nel, double *sink, int skip) {
  const int tid = threadIdx.x, el = tid >> 5, q = tid & 31;
  double2 acc = make_double2(0, 0);
  const int64_t stride = (int64_t)gridDim.x * 8;
  for (int64_t e0 = (int64_t)blockIdx.x * 8 + el; e0 < nel; e0 += stride * AHEAD) {
    double2 v[AHEAD][5];
#pragma unroll
    for (int a = 0; a < AHEAD; a++) {
      const int64_t e = e0 + a * stride;
      if (e < nel) {
        const double2 *src = reinterpret_cast<const double2 *>(gp + e * 400 + skip);
#pragma unroll
        for (int k = 0; k < 5; k++) v[a][k] = (q + 32 * k < (400 - skip) / 2) ? src[q + 32 * k] : make_double2(0, 0);
      } else {
#pragma unroll
        for (int k = 0; k < 5; k++) v[a][k] = make_double2(0, 0);
      }
    }
#pragma unroll
    for (int a = 0; a < AHEAD; a++)
#pragma unroll
      for (int k = 0; k < 5; k++) { acc.x += v[a][k].x; acc.y += v[a][k].y; }
  }
  if (acc.x == 123.456) sink[0] = acc.y;
}
// plain linear read
__global__ void lin(const double2 *p, int64_t n2, double *sink) {
  double2 acc = make_double2(0, 0);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n2; i += (int64_t)gridDim.x * blockDim.x) {
    const double2 v = p[i]; acc.x += v.x; acc.y += v.y;
  }
  if (acc.x == 123.456) sink[0] = acc.y;
}
template <typename F>
void timeit(const char *name, double bytes, F launch) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; r++) launch();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  printf("%-50s %.3f ms  %.2f TB/s\n", name, ms, bytes / ms / 1e9);
}
int main() {
  const int64_t nel = 2097152;
  double *gp, *sink; hipMalloc(&gp, nel * 400 * 8); hipMalloc(&sink, 64);
  hipMemset(gp, 0x3f, nel * 400 * 8);  // non-zero bytes
  const double b288 = nel * 288.0 * 8, b400 = nel * 400.0 * 8;
  for (int wg : {4, 8}) {
    printf("--- %d WG/CU ---\n", wg);
    timeit("linear 16B/lane, whole array", b400, [&] { lin<<<256 * wg, 256>>>((const double2 *)gp, nel * 200, sink); });
    timeit("records [112,400), 1 element ahead", b288, [&] { rd<1><<<256 * wg, 256>>>(gp, nel, sink, 112); });
    timeit("records [112,400), 2 ahead", b288, [&] { rd<2><<<256 * wg, 256>>>(gp, nel, sink, 112); });
    timeit("records [112,400), 4 ahead", b288, [&] { rd<4><<<256 * wg, 256>>>(gp, nel, sink, 112); });
    timeit("records [0,400) whole, 2 ahead (bytes=400)", b400, [&] { rd<2><<<256 * wg, 256>>>(gp, nel, sink, 0); });
  }
  return 0;
}
