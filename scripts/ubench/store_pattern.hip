// Store-pattern microbenchmark: which write shapes of the CSR value array reach the HBM fill rate?
// All variants write the same 2,097,152 node blocks of 243 doubles (4.08 GB), 64 nodes per wave-item.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
constexpr int NB = 243, NR = 81;
struct __attribute__((packed, aligned(8))) d2 { double x, y; };

// P0: linear fill, 16 B per lane
__global__ void p0(double *out, int64_t n2) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n2; i += (int64_t)gridDim.x * blockDim.x)
    reinterpret_cast<double2 *>(out)[i] = make_double2(1.0, 2.0);
}
// P1: wave w<3 writes row c=w of its 64 nodes, lane = node, 40 x 16 B + 8 B, lane stride 1944 B
template <int WIDE>
__global__ __launch_bounds__(256) void p1(double *out, int64_t ntile, int iters) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (w == 3) return;
  for (int it = 0; it < iters; it++) {
    const int64_t tile = blockIdx.x + (int64_t)it * gridDim.x;
    if (tile >= ntile) break;
    double *p = out + (tile * 64 + lane) * NB + w * NR;
    const double v = lane + it;
    if (WIDE) {
#pragma unroll
      for (int k = 0; k < 40; k++) *reinterpret_cast<d2 *>(p + 2 * k) = d2{v, v};
      p[80] = v;
    } else {
#pragma unroll
      for (int k = 0; k < 81; k++) p[k] = v;
    }
  }
}
// P1nt: as P1 (8-byte stores) with non-temporal stores
__global__ __launch_bounds__(256) void p1nt(double *out, int64_t ntile, int iters) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (w == 3) return;
  for (int it = 0; it < iters; it++) {
    const int64_t tile = blockIdx.x + (int64_t)it * gridDim.x;
    if (tile >= ntile) break;
    double *p = out + (tile * 64 + lane) * NB + w * NR;
    const double v = lane + it;
#pragma unroll
    for (int k = 0; k < 81; k++) __builtin_nontemporal_store(v, p + k);
  }
}
// P3: transposed: wave w<3 writes row c=w of 64 nodes as 81 instructions of 64 consecutive values of [node][81]
__global__ __launch_bounds__(256) void p3(double *out, int64_t ntile, int iters) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (w == 3) return;
  for (int it = 0; it < iters; it++) {
    const int64_t tile = blockIdx.x + (int64_t)it * gridDim.x;
    if (tile >= ntile) break;
    double *p = out + tile * 64 * NB + w * NR;
    const double v = lane + it;
#pragma unroll 9
    for (int k = 0; k < 81; k++) {
      const int idx = lane + 64 * k, node = idx / NR, r = idx - node * NR;
      p[node * NB + r] = v;
    }
  }
}
// P4: whole 64-node block contiguous, 3 waves, 16 B per lane
__global__ __launch_bounds__(256) void p4(double *out, int64_t ntile, int iters) {
  extern __shared__ double lds[];
  const int tid = threadIdx.x;
  if (tid >= 192) return;
  for (int it = 0; it < iters; it++) {
    const int64_t tile = blockIdx.x + (int64_t)it * gridDim.x;
    if (tile >= ntile) break;
    d2 *p = reinterpret_cast<d2 *>(out + tile * 64 * NB);
    const double v = tid + it;
    for (int k = tid; k < 64 * NB / 2; k += 192) p[k] = d2{v, v};
  }
}
// P5: thirds: wave w<3 writes third dz=w (27 values of each of the 3 rows) of 64 nodes, transposed 64-value instrs
__global__ __launch_bounds__(256) void p5(double *out, int64_t ntile, int iters) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (w == 3) return;
  for (int it = 0; it < iters; it++) {
    const int64_t tile = blockIdx.x + (int64_t)it * gridDim.x;
    if (tile >= ntile) break;
    const double v = lane + it;
    for (int c = 0; c < 3; c++) {
      double *p = out + tile * 64 * NB + c * NR + w * 27;
#pragma unroll 9
      for (int k = 0; k < 27; k++) {
        const int idx = lane + 64 * k, node = idx / 27, r = idx - node * 27;
        p[node * NB + r] = v;
      }
    }
  }
}
// P6: like P1 wide but lanes in groups: each instruction covers 8 nodes x 8 lanes x 16 B = 128 B runs
__global__ __launch_bounds__(256) void p6(double *out, int64_t ntile, int iters) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (w == 3) return;
  for (int it = 0; it < iters; it++) {
    const int64_t tile = blockIdx.x + (int64_t)it * gridDim.x;
    if (tile >= ntile) break;
    const double v = lane + it;
    for (int g = 0; g < 8; g++) {  // 8 nodes per group pass
      double *p = out + (tile * 64 + g * 8 + (lane >> 3)) * NB + w * NR;
#pragma unroll
      for (int k = 0; k < 5; k++) *reinterpret_cast<d2 *>(p + 16 * k + 2 * (lane & 7)) = d2{v, v};
      if ((lane & 7) == 0) p[80] = v;
    }
  }
}

// P7: contiguous runs of `run` doubles at 8-byte alignment offset `mis`, 192 threads, 16 B per lane
__global__ __launch_bounds__(256) void p7(double *out, int64_t ntile, int iters, int run, int mis, int64_t gap) {
  extern __shared__ double lds[];
  const int tid = threadIdx.x;
  if (tid >= 192) return;
  const int nrun = 64 * NB / run;
  for (int it = 0; it < iters; it++) {
    const int64_t tile = blockIdx.x + (int64_t)it * gridDim.x;
    if (tile >= ntile) break;
    const double v = tid + it;
    for (int r = 0; r < nrun; r++) {
      d2 *p = reinterpret_cast<d2 *>(out + mis + tile * 64 * NB + (int64_t)r * (run + gap));
      for (int k = tid; k < run / 2; k += 192) p[k] = d2{v, v};
    }
  }
}
// P8: whole block contiguous, 8 B per lane
__global__ __launch_bounds__(256) void p8(double *out, int64_t ntile, int iters) {
  extern __shared__ double lds[];
  const int tid = threadIdx.x;
  if (tid >= 192) return;
  for (int it = 0; it < iters; it++) {
    const int64_t tile = blockIdx.x + (int64_t)it * gridDim.x;
    if (tile >= ntile) break;
    double *p = out + tile * 64 * NB;
    const double v = tid + it;
    for (int k = tid; k < 64 * NB; k += 192) p[k] = v;
  }
}

// P9: the autonomous-wave form for dof 3 (what k_assemble_p3_rect / k_assemble_e2_rect do for dof 1 / 2): one wave
// per workgroup owns a run of 21 nodes x 243 doubles (40.8 KB stage, 3 waves per CU), lane = (node, dof row) with
// NFMA dependent-free FMAs on its 81 accumulators per step, deposit, then 16 B per lane from the line boundary below
// the run.  Skeleton only: is a wave-private run worth building for the elastic kernel?
template <int NFMA>
__global__ __launch_bounds__(64) void p9(double *out, int64_t nrun) {
  constexpr int RL = 21, RUN = RL * NB;
  __shared__ double stage[RUN];
  const int lane = threadIdx.x, n = lane % RL, c = lane / RL;
  double acc[81];
  for (int q = 0; q < 81; q++) acc[q] = lane + q;
  for (int64_t r = blockIdx.x; r < nrun; r += gridDim.x) {
    const double f = 1.0 + 1e-9 * (double)r;
#pragma unroll
    for (int k = 0; k < NFMA / 81; k++)
#pragma unroll
      for (int q = 0; q < 81; q++) acc[q] = fma(acc[q], f, 0.5);
    __builtin_amdgcn_wave_barrier();
    if (c < 3)
#pragma unroll
      for (int q = 0; q < 81; q++) stage[n * NB + c * 81 + q] = acc[q];
    __builtin_amdgcn_wave_barrier();
    double *runp = out + r * RUN;
    const int head = (int)((reinterpret_cast<uintptr_t>(runp) & 127) >> 3);
    const int nchunk = (head + RUN + 1) >> 1;
    constexpr int NIT = (RUN / 2 + 8 + 63) / 64;
#pragma unroll 8
    for (int it = 0; it < NIT; it++) {
      const int q = lane + 64 * it, s0 = 2 * q - head;
      const int sa = s0 < 0 ? 0 : (s0 < RUN ? s0 : RUN - 1), sb = s0 + 1 < 0 ? 0 : (s0 + 1 < RUN ? s0 + 1 : RUN - 1);
      const double2 v = make_double2(stage[sa], stage[sb]);
      if (q < nchunk) {
        if (s0 >= 0 && s0 + 1 < RUN) *reinterpret_cast<double2 *>(runp + s0) = v;
        else if (s0 == -1) runp[0] = v.y;
        else if (s0 == RUN - 1) runp[s0] = v.x;
      }
    }
  }
}

template <typename F>
void timeit(const char *name, double bytes, F launch) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; r++) launch();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  printf("%-44s %.3f ms  %.2f TB/s\n", name, ms, bytes / ms / 1e9);
}
int main() {
  const int64_t nnode = 2097152, ntile = nnode / 64;
  const double bytes = 8.0 * NB * nnode;
  double *out; hipMalloc(&out, (size_t)bytes + 4096);
  {
    const int64_t nrun = nnode / 21;
    printf("--- autonomous waves, 3 per CU ---\n");
    timeit("P9 wave-private 21-node runs, no FMAs", 8.0 * NB * 21 * nrun, [&] { p9<0><<<256 * 3, 64>>>(out, nrun); });
    timeit("P9 wave-private 21-node runs, 486 FMAs/step", 8.0 * NB * 21 * nrun, [&] { p9<486><<<256 * 3, 64>>>(out, nrun); });
    timeit("P9 wave-private 21-node runs, 1458 FMAs/step", 8.0 * NB * 21 * nrun, [&] { p9<1458><<<256 * 3, 64>>>(out, nrun); });
  }
  for (int wgcu : {2, 4}) {
    const int lds = wgcu == 2 ? 70000 : 36000;
    const int grid = 256 * wgcu, iters = (int)((ntile + grid - 1) / grid);
    printf("--- %d WG/CU ---\n", wgcu);
    hipFuncSetAttribute((const void *)p1<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipFuncSetAttribute((const void *)p1<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipFuncSetAttribute((const void *)p3, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipFuncSetAttribute((const void *)p4, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipFuncSetAttribute((const void *)p5, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipFuncSetAttribute((const void *)p6, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    timeit("P0 linear fill 16B/lane", bytes, [&] { p0<<<256 * 8, 256>>>(out, (int64_t)(bytes / 16)); });
    timeit("P1 row per lane, 16B stores (stride 1944B)", bytes, [&] { p1<1><<<grid, 256, lds>>>(out, ntile, iters); });
    timeit("P1 row per lane, 8B stores", bytes, [&] { p1<0><<<grid, 256, lds>>>(out, ntile, iters); });
    hipFuncSetAttribute((const void *)p1nt, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    timeit("P1 row per lane, 8B non-temporal stores", bytes, [&] { p1nt<<<grid, 256, lds>>>(out, ntile, iters); });
    timeit("P3 row transposed, 648B runs, 8B/lane", bytes, [&] { p3<<<grid, 256, lds>>>(out, ntile, iters); });
    timeit("P4 whole block contiguous 16B/lane", bytes, [&] { p4<<<grid, 256, lds>>>(out, ntile, iters); });
    timeit("P5 thirds transposed, 216B runs (current)", bytes, [&] { p5<<<grid, 256, lds>>>(out, ntile, iters); });
    timeit("P6 row, 8 lanes per node, 128B runs", bytes, [&] { p6<<<grid, 256, lds>>>(out, ntile, iters); });
    hipFuncSetAttribute((const void *)p7, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipFuncSetAttribute((const void *)p8, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    timeit("P7 block contiguous, misaligned by 8 B", bytes, [&] { p7<<<grid, 256, lds>>>(out, ntile, iters, 64 * NB, 1, 0); });
    timeit("P7 4 runs of 31 KB (tile rows), aligned", bytes, [&] { p7<<<grid, 256, lds>>>(out, ntile, iters, 16 * NB, 0, 0); });
    timeit("P7 4 runs of 31 KB, misaligned 8 B", bytes, [&] { p7<<<grid, 256, lds>>>(out, ntile, iters, 16 * NB, 1, 0); });
    timeit("P7 64 runs of 1944 B (node blocks)", bytes, [&] { p7<<<grid, 256, lds>>>(out, ntile, iters, NB - 1, 0, 1); });
    timeit("P7 block contiguous, misaligned by 16 B", bytes, [&] { p7<<<grid, 256, lds>>>(out, ntile, iters, 64 * NB, 2, 0); });
    timeit("P7 block contiguous, misaligned by 32 B", bytes, [&] { p7<<<grid, 256, lds>>>(out, ntile, iters, 64 * NB, 4, 0); });
    timeit("P7 block contiguous, misaligned by 64 B", bytes, [&] { p7<<<grid, 256, lds>>>(out, ntile, iters, 64 * NB, 8, 0); });
    timeit("P8 block contiguous, 8 B per lane", bytes, [&] { p8<<<grid, 256, lds>>>(out, ntile, iters); });
  }
  return 0;
}
